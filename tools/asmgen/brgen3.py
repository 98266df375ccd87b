"""Generator of the COOPERATIVE blind-rotation loop: THREE waves work on ONE unit, wave c owns polynomial c of the
accumulator (k_blind_rotate6, launches of at most one unit per SIMD -- e.g. the 8-blocks-per-GPU share of the medium
instance on an 8-GPU node, or a single AES block).

Why: a CMux chain is 768 dependent steps, and with one unit per SIMD the step time IS the launch time.  One wave per
unit spends a step on three rotations, three forward transforms, nine multiply-accumulates, three inverses and three
updates, all in one in-order instruction stream (k_blind_rotate: 24-26 k cycles per step).  Here every wave does ONE of
each -- rotate and decompose its own polynomial, transform it, publish the transform through its (idle) transpose tile,
multiply-accumulate column c against the three published transforms, invert, update its own polynomial -- so the
dependent path of a step is a third as long and the accumulator never moves: wave c's polynomial lives in its
registers from the first step to the last.  What crosses between waves is 8 KiB per wave and step (one digit
transform), read twice by the neighbours; two s_barrier per step: one after the transforms are published, one before
the tiles are reused for the inverse's exchanges.  Three waves per SIMD (168 registers each) when the launch fills
the chip: they cover each other's LDS and key round trips as the two waves of k_blind_rotate5 do.

Arithmetic, operation order and results are those of k_blind_rotate / brgen.py / the oracle's txo_blind_rotate():
every column accumulates q = 0, 1, 2 in that order, transforms and conversions are the same instruction sequences.
Bit-identical; tests/asmgen_emu_blind_rotate3.py runs three emulated waves (shared LDS, barriers honoured, both
scheduling orders) against the oracle's primitives.

    python3 tools/asmgen/brgen3.py --out temp-fhe-transciphering_amd/csrc      (tx_br3_asm.inc + clobbers)

LDS image of a workgroup (192 threads = one unit): compact twiddles at 0 (9 088 B), then three 8 320-byte tiles (pitch
65 complex), one per wave: transpose tile of that wave's transforms, staging area of its rotation, and between the two
barriers the published transform [8 registers][64 lanes] complex.  34 048 B: four workgroups per CU.

Register map (VGPR, 168): 0..15 as brgen.py (9: tile 0 + lane * 16, the address the published transforms are read
from), A 16..47 (this wave's polynomial, 16 x u64), D 48..79 (digits -> their transform; landing ring of the
published transforms in the multiply-accumulate), F 80..111 (rotation addresses and sign masks, then the Fourier
accumulator), X 112..159 (key ring of 12 quads; twiddles + spare in the transforms), E 160..167 (two more landing quads).
"""
import os
import sys
from gcn import V, S, Lit, VCC, Prog, M0
import brgen
from brgen import (Gen, SG, quad, qre, qim, f64bits, T0, T1, LANE, L8T, L16T, L16, OFF1, TWBA, XHW, XLW, EREG, L8, R10, R11,
                   CARRY, PITCH, N, H, KP, NSTAMP, ST_ACC, ST_DBG)

NV = 168                                 # registers per wave: three waves per SIMD
A3, D3, F3, X3, E3 = 16, 48, 80, 112, 160
RA3 = [V(F3 + m) for m in range(16)]     # rotation addresses (before the multiply-accumulate)
SGN3 = [V(F3 + 16 + m) for m in range(16)]
L16T0 = V(9)                             # tile of wave 0 + lane * 16
TILE_STRIDE = 8 * PITCH                  # 8 320 bytes from one wave's tile to the next
TILE0 = S(68)                            # LDS address of wave 0's tile
ABL3 = os.environ.get("BR3_ABL", "")     # timing-only ablations (B no barriers, P no publish/read of transforms): outputs invalid
# fair arbitration between the workgroups of a CU (below): their step counters live in one 32-byte entry of a table in global memory
FTAB = S(66, 2)                          # this CU's table entry: four progress words
FSLOT = S(69)                            # bits 3:0: byte offset of this workgroup's word in it; bits 31:4: PFOFF (key prefetch, below)
PF_ROWS = int(os.environ.get("BR3_PF_ROWS", 0))       # the key row touched this many steps ahead of its use (0: no prefetch)
PF_LDS = 3 * TILE_STRIDE                 # from wave 0's tile: 1 KiB behind the three tiles, where the touched bytes are dropped
FEPOCH = S(59)                           # launch number << 16: a word of another launch is not a competitor
FAIR_LEADS = [int(x) for x in os.environ.get("BR3_FAIR_LEADS", "2,4,8").replace("-", ",").split(",")]   # lead (steps) below which the priority is 3, 2, 1
# SOFT mode (round 5, k_blind_rotate8): several units share ONE workgroup (placement by construction, as k_blind_rotate7) but the
# three waves of a unit meet at a SOFTWARE barrier of their own -- per-wave arrival words in LDS, polled -- instead of s_barrier,
# which would span every unit of the workgroup (lockstep); the fair arbitration reads the units' step counters from LDS too.
# The registers of the global fairness table are free in this mode and carry the LDS addresses.
SFLAGS = S(66)                           # LDS address of this unit's arrival words (three used, 16 bytes)
SMYFLAG = S(67)                          # ... of this wave's word
SPROG = S(69)                            # LDS address of the workgroup's four step counters (16 bytes)
SMYPROG = S(70)                          # ... of this unit's counter
SBIDX = S(59)                            # barriers passed so far
SSPIN = S(71)                            # polls left before a wave gives up waiting (a broken launch must still END)
SFAULT = S(72, 2)                        # global address of the context's fault word: a wave that gave up stores 1 there (the call then fails)
SOFT_SPIN = int(os.environ.get("BR3_SOFT_SPIN", 1 << 18))      # (the emulator test of the give-up path sets 3)
SOFT_SLEEP = int(os.environ.get("BR3_SOFT_SLEEP", 4))      # s_sleep units (64 cycles) between two polls (1, 2, 4 measured: 5.88, 5.86, 5.85 ms per 1024 units)


class CoopGen(Gen):
    def __init__(self, stamps=False, prio="", xhi_regs=None, xlo_regs=None, cu_fair=None, soft=None):
        super().__init__(stamps=stamps, prio=prio, fair=False)
        self.soft = (os.environ.get("BR3_SOFT", "0") == "1") if soft is None else soft
        if self.soft and os.environ.get("BR3_SOFT_NOFAIR") == "1": cu_fair = False      # debugging
        assert not (self.soft and stamps), "the soft-barrier form uses the stamp accumulators' registers"
        # Fair arbitration between the (up to four) workgroups of a CU.  The SIMD arbiter serves the oldest wave first, so the
        # workgroups of a CU finish one after the other (1024 units: at 3.9, 5.0, 6.5 and 8.1 ms) and the last one runs with ever
        # fewer neighbours to fill its stalls; the vector pipe, 78 % busy while all four are there (SQ_ACTIVE_INST_VALU x 3 waves
        # / SQ_WAVE_CYCLES, profiles/r4d_*), idles in the tail.  Each workgroup therefore publishes its step counter in its CU's
        # entry of a global table (one word per workgroup, tagged with the launch number), reads the entry one step ahead of its
        # use, and sets its priority by its lead over the slowest workgroup of the entry: the slowest runs at s_setprio 3, a
        # lead of FAIR_LEADS steps or more at 2, 1, 0.  All advance at the same rate and finish together.  Priorities only:
        # results cannot depend on it.
        self.cu_fair = (os.environ.get("BR3_FAIR", "1") == "1") if cu_fair is None else cu_fair
        # BR3_XHI=regs: exchange hi of both transforms in registers (v_permlane32_swap / v_permlane16_swap / DPP) instead of
        # through the tile: 32 of a step's 174 LDS instructions less per wave for 128 more vector instructions.  Measured
        # (profiles/r4d_*): 2 % faster at one workgroup per CU, equal at three, slower at four -- there the vector pipe is
        # the bound, not the LDS pipeline -- so the default stays the LDS form; both exchanges in registers lose everywhere
        self.xhi_regs = (os.environ.get("BR3_XHI", "lds") == "regs") if xhi_regs is None else xhi_regs
        # both exchanges in registers: the tile then only stages the rotation and carries the published transform, and the second
        # barrier of a step (nobody still reads my tile) moves from the middle of the inverse to the next step's staging stores
        self.xlo_regs = (os.environ.get("BR3_XLO", "lds") == "regs") if xlo_regs is None else xlo_regs
        assert self.xhi_regs or not self.xlo_regs

    def barrier(self, phase):
        """every LDS operation of this wave has completed, then the workgroup's three waves meet (diagnostic build: the time
        up to the barrier goes to `phase`, the wait at the barrier to accumulator 5)"""
        self.lg_wait_all()
        self.stamp(phase)
        if self.soft: self.soft_barrier()
        elif "B" not in ABL3: self.p.emit("s_barrier")
        self.stamp(5)

    SOFTQ = quad(E3, 0)                      # landing quad of the polls: free outside the multiply-accumulate

    def soft_barrier(self):
        """The unit's three waves meet without s_barrier: each writes the number of this barrier to its own word (a wave's LDS
        operations execute in order, so whoever reads the word also finds what the wave stored before it), then polls the
        unit's three words until none is behind.  s_sleep between polls leaves the SIMD to the other units' waves -- that is the
        point: the units of a workgroup drift apart and fill each other's stalls.  A wave gives up after SOFT_SPIN polls: wrong
        results (the tests would show them) instead of a launch that never ends."""
        e = self.e
        if "B" in ABL3: return
        self.nlab += 1
        poll, done = "L_sp%d_%%=" % self.nlab, "L_sd%d_%%=" % self.nlab
        e("s_add_u32", SBIDX, SBIDX, Lit(1))
        e("v_mov_b32", T0.lo(), SMYFLAG)
        e("v_mov_b32", T0.hi(), SBIDX)
        self.ds_write("ds_write_b32", T0.lo(), T0.hi())
        e("v_mov_b32", T0.lo(), SFLAGS)
        e("s_mov_b32", SSPIN, Lit(SOFT_SPIN))
        self.p.label(poll)
        self.ds_read("ds_read_b128", self.SOFTQ, T0.lo())
        self.lg_wait_all()
        e("v_min3_u32", T0.hi(), V(self.SOFTQ.i), V(self.SOFTQ.i + 1), V(self.SOFTQ.i + 2))
        e("v_readfirstlane_b32", SG.TMP, T0.hi())
        e("s_nop", Lit(3))
        e("s_cmp_ge_u32", SG.TMP, SBIDX)
        e("s_cbranch_scc1", done)
        e("s_sub_u32", SSPIN, SSPIN, Lit(1))
        e("s_cmp_eq_u32", SSPIN, Lit(0))
        nap = "L_sn%d_%%=" % self.nlab
        e("s_cbranch_scc0", nap)
        # gave up: the results of this launch are invalid.  Say so -- the host fails the call -- and go on to the end.  (Emitted
        # outside the wait-count model: the store is retired right here, with whatever else was in flight.)
        self.p.emit("v_mov_b32", T0.lo(), Lit(0))
        self.p.emit("v_mov_b32", T0.hi(), Lit(1))
        self.p.emit("global_store_dword", T0.lo(), T0.hi(), SFAULT, offset=0)
        self.p.emit("s_waitcnt", vm=0)
        e("s_branch", done)
        self.p.label(nap)
        e("s_sleep", Lit(SOFT_SLEEP))
        e("s_branch", poll)
        self.p.label(done)
        if os.environ.get("BR3_SOFT_HW") == "1": self.p.emit("s_barrier")      # debugging: the hardware barrier behind the soft one

    def soft_fair_publish(self):
        """loop head: this unit's step counter to LDS (its three waves store the same word), the four counters requested"""
        e = self.e
        e("v_mov_b32", T0.lo(), SMYPROG)
        e("v_mov_b32", T0.hi(), SG.I)
        self.ds_write("ds_write_b32", T0.lo(), T0.hi())
        e("v_mov_b32", T0.lo(), SPROG)
        self.ds_read("ds_read_b128", self.SOFTQ, T0.lo())

    def soft_fair_apply(self):
        """priority by this unit's lead over the slowest unit of the workgroup (a finished unit's counter is 0xFFFFFFF0)"""
        e = self.e
        self.c("---- fair arbitration between the units of the workgroup (LDS step counters) -> s_setprio")
        words = [SG.CTV.lo(), SG.CTV.hi(), SG.TMP, SG.TMP2]
        for k in range(4): e("v_readfirstlane_b32", words[k], V(self.SOFTQ.i + k))
        e("s_nop", Lit(3))
        e("s_min_u32", SG.TMP, SG.TMP, SG.TMP2)
        e("s_min_u32", SG.CTV.lo(), SG.CTV.lo(), SG.CTV.hi())
        e("s_min_u32", SG.TMP2, SG.TMP, SG.CTV.lo())
        e("s_min_u32", SG.TMP2, SG.TMP2, SG.I)
        e("s_sub_u32", SG.TMP, SG.I, SG.TMP2)
        self.nlab += 1
        done = "L_fp%d_%%=" % self.nlab
        for prio, lead in zip((3, 2, 1), FAIR_LEADS):
            nxt = "L_fq%d_%d_%%=" % (self.nlab, prio)
            e("s_cmp_lt_u32", SG.TMP, Lit(lead))
            e("s_cbranch_scc0", nxt)
            e("s_setprio", Lit(prio))
            e("s_branch", done)
            self.p.label(nxt)
        e("s_setprio", Lit(0))
        self.p.label(done)

    def soft_fair_finish(self):
        e = self.e
        e("s_setprio", Lit(0))
        e("v_mov_b32", T0.lo(), SMYPROG)
        e("v_mov_b32", T0.hi(), Lit(0xFFFFFFF0))
        self.ds_write("ds_write_b32", T0.lo(), T0.hi())
        self.lg_wait_all()

    # ---------------------------------------------------------------- one polynomial forward (fft_fwd of tx_device.hpp)
    def fwd1(self, v, xq, before_pass3=None):
        tw, spare = xq[:8], xq[8]
        home_lo = min(q.i for q in v)
        self.c("forward: twist constants")
        self.load_tw(tw, 0, 8)
        self.twist_consts([v], None, conj=False)
        self.c("forward: pass 1 (dft8, pass-A twiddles with the lane part of the twist, exchange hi)")
        v, spare = self.dft8(v, spare, +1)
        self.cmul_all([(v[r], tw[r]) for r in range(8)])
        v, spare = self.xchg_hi(v, spare)
        self.c("forward: pass 2 (dft8, pass-B twiddles, exchange lo)")
        self.load_tw(tw, 0, 7, compact=True)
        v, spare = self.dft8(v, spare, +1)
        self.cmul_all([(v[r], tw[r - 1]) for r in range(1, 8)])
        v, spare = self.xchg_lo(v, spare)
        self.c("forward: pass 3 (dft8)")
        if before_pass3: before_pass3()            # the twiddle quads xq[0..7] are free from here on
        v, spare = self.dft8(v, spare, +1, home=(home_lo, home_lo + 32))
        v, spare = self.home(v, spare, home_lo, home_lo + 32)
        return v, spare

    def inv1_hooked(self, v, xq, hook):
        """Gen.inv1 with `hook` in front of the first exchange store (the tile is still being read by the neighbours
        until then)"""
        tw, spare = xq[:8], xq[8]
        self.c("inverse: pass 3' (dft8, exchange lo)")
        self.load_tw(tw, 0, 7, compact=True)
        v, spare = self.dft8(v, spare, -1)
        hook()
        v, spare = self.xchg_lo(v, spare)
        self.c("inverse: pass 2' (pass-B twiddles conj, dft8, exchange hi)")
        self.cmul_all([(v[r], tw[r - 1]) for r in range(1, 8)], spare, conj=True)
        self.load_tw(tw, 0, 8)
        v, spare = self.dft8(v, spare, -1)
        v, spare = self.xchg_hi(v, spare)
        self.c("inverse: pass 1' (pass-A twiddles conj, dft8, untwist constants)")
        self.cmul_all([(v[r], tw[r]) for r in range(8)], spare, conj=True)
        home_lo = min(q.i for q in v)
        v, spare = self.dft8(v, spare, -1, home=(home_lo, home_lo + 32))
        v, spare = self.home(v, spare, home_lo, home_lo + 32)
        self.twist_consts([v], spare, conj=True)
        return v, spare

    # ---------------------------------------------------------------- the kernel
    def prologue(self):
        e = self.e
        self.c("---- prologue: lane-dependent addresses, constants, this wave's polynomial from its tile")
        self.p.emit("s_waitcnt", vm=0, lgkm=0)
        e("v_mbcnt_lo_u32_b32", LANE, Lit(-1), Lit(0))
        e("v_mbcnt_hi_u32_b32", LANE, Lit(-1), LANE)
        e("v_lshl_add_u32", L8T, LANE, Lit(3), SG.TILE)
        e("v_lshl_add_u32", L16T, LANE, Lit(4), SG.TILE)
        e("v_lshl_add_u32", L16T0, LANE, Lit(4), TILE0)
        e("v_lshlrev_b32", L16, Lit(4), LANE)
        if self.e_reg is None: e("v_add_u32", OFF1, Lit(4096), L16)
        e("v_and_b32", T0.lo(), Lit(7), LANE)                   # b
        e("v_lshlrev_b32", TWBA, Lit(4), T0.lo())
        e("v_add_u32", TWBA, Lit(brgen.TWB), TWBA)
        e("v_lshrrev_b32", T0.hi(), Lit(3), LANE)               # a
        e("v_lshl_add_u32", T1.lo(), T0.hi(), Lit(6), T0.hi())  # a*65
        e("v_add_u32", T0.hi(), T1.lo(), T0.lo())               # a*65 + b
        e("v_lshl_add_u32", XHW, T0.hi(), Lit(4), SG.TILE)
        e("v_and_b32", T1.lo(), Lit(0x38), LANE)                # g8
        e("v_lshl_add_u32", T1.hi(), T0.lo(), Lit(6), T0.lo())  # b*65
        e("v_add_u32", T1.lo(), T1.hi(), T1.lo())               # b*65 + g8
        e("v_lshl_add_u32", XLW, T1.lo(), Lit(4), SG.TILE)
        self.tie_lane(T0.lo())                                   # lane & 1 (balanced rule) or 0 (tfhe-rs' rule)
        e("v_lshlrev_b32", EREG, Lit(9), T0.lo())                # 0x100 - (e << 9), e = 1 - (lane & 1)   (digit_pair)
        e("v_add_u32", EREG, Lit(-0x100), EREG)
        if self.e_reg is not None: e("v_xor_b32", self.e_reg, Lit(1), T0.lo())
        self.even_lanes_to_vcc()
        for name, val in (("C_2M64", 2.0 ** -64), ("C_M2P64", -(2.0 ** 64)), ("C_84", 1.5 * 2.0 ** 84), ("C_52", 1.5 * 2.0 ** 52),
                          ("C_SQH", 0.70710678118654752440)):
            s, b = getattr(SG, name), f64bits(val)
            e("s_mov_b32", s.lo(), Lit(b & 0xFFFFFFFF)); e("s_mov_b32", s.hi(), Lit(b >> 32))
        e("s_mov_b32", SG.C_M4338, Lit((-0x43380000) & 0xFFFFFFFF))
        # the wrapper left this wave's polynomial (zero for the mask polynomials, the rotated test vector for the body)
        # in its tile, natural order
        for m in range(16): self.ds_read("ds_read_b64", V(A3 + 2 * m, 2), L8T, offset=512 * m)
        if self.neg_acc:         # the loop works on -acc (brgen.Gen)
            for m in range(16): self.negate64(V(A3 + 2 * m, 2), CARRY[m % 4])
        e("v_mov_b32", R10, Lit(0))
        self.gload("global_load_dwordx2", V(10, 2), R10, SG.CT, offset=0)     # mask element 0
        e("s_mov_b32", SG.I, Lit(0))
        if self.soft: e("s_mov_b32", SBIDX, Lit(0))
        elif self.cu_fair: self.fair_request()
        self.lg_wait_all()
        if self.stamps:
            for k in range(NSTAMP):
                e("s_mov_b32", ST_ACC[k].lo(), Lit(0)); e("s_mov_b32", ST_ACC[k].hi(), Lit(0))
            self.stamp(None)

    # ---------------------------------------------------------------- fair arbitration between the workgroups of a CU
    FAIRQ = quad(E3, 0)                      # landing quad of the table entry: free outside the multiply-accumulate

    def fair_request(self):
        """the CU's table entry, a step ahead of its use (sc0 sc1: from L2, where the other workgroups' stores are).  (Also issued
        under launch number 0, where nothing reads it: the loop head's full wait expects one outstanding load on either path.)"""
        self.e("v_mov_b32", T0.lo(), Lit(0))
        self.e("global_load_dwordx4", self.FAIRQ, T0.lo(), FTAB, offset=0, cache="sc0 sc1")

    def fair_step(self):
        """loop head (every vector-memory operation has retired): publish my step counter, take the entry requested a step
        ago, priority by my lead over the slowest workgroup of this launch on this CU"""
        e = self.e
        self.c("---- fair arbitration: publish step, lead over the slowest workgroup of the CU -> s_setprio")
        self.nlab += 1
        off = "L_fo%d_%%=" % self.nlab
        e("s_cmp_eq_u32", FEPOCH, Lit(0))                        # launch number 0: a launch that has nobody to be fair to
        e("s_cbranch_scc1", off)
        e("s_or_b32", SG.TMP, FEPOCH, SG.I)
        e("s_and_b32", SG.TMP2, FSLOT, Lit(15))
        e("v_mov_b32", T0.lo(), SG.TMP2)
        e("v_mov_b32", T0.hi(), SG.TMP)
        e("global_store_dword", T0.lo(), T0.hi(), FTAB, offset=0)
        words = [SG.CTV.lo(), SG.CTV.hi(), SG.TMP, SG.TMP2]       # (all dead at the loop head)
        for k in range(4): e("v_readfirstlane_b32", words[k], V(self.FAIRQ.i + k))
        e("s_nop", Lit(3))
        for k in range(4): e("s_xor_b32", words[k], words[k], FEPOCH)       # a word of this launch: its step count; of another: >= 2^16
        e("s_min_u32", SG.TMP, SG.TMP, SG.TMP2)
        e("s_min_u32", SG.CTV.lo(), SG.CTV.lo(), SG.CTV.hi())
        e("s_min_u32", SG.TMP2, SG.TMP, SG.CTV.lo())
        e("s_min_u32", SG.TMP2, SG.TMP2, SG.I)                   # (my own word may be a step old)
        e("s_sub_u32", SG.TMP, SG.I, SG.TMP2)                    # my lead, >= 0
        self.nlab += 1
        done = "L_fp%d_%%=" % self.nlab
        for prio, lead in zip((3, 2, 1), FAIR_LEADS):
            nxt = "L_fq%d_%d_%%=" % (self.nlab, prio)
            e("s_cmp_lt_u32", SG.TMP, Lit(lead))
            e("s_cbranch_scc0", nxt)
            e("s_setprio", Lit(prio))
            e("s_branch", done)
            self.p.label(nxt)
        e("s_setprio", Lit(0))
        self.p.label(done)
        self.p.label(off)

    def key_prefetch(self):
        """EXPERIMENT, off by default (BR3_PF_ROWS=0): touch 1 KiB of the key row PF_ROWS steps ahead (LDS-DMA into a dump area: no
        registers) so that it is in L2 when the multiply-accumulate asks for it.  Built on the suspicion that the round pipeline
        (whose other kernels stream gigabytes between two blind rotations) costs the cooperative kernel its key in the caches;
        measured, it does not: 8.5-9.1 ms with the prefetch against 8.2-8.9 ms without for 1024 units in the pipeline
        (gpurun_out/r4g_sustained.log) -- what the pipeline costs there is workgroup PLACEMENT (k_blind_rotate7, tx_br2.hpp).
        The row has 72 pieces of 1 KiB; the wrapper gives every wave one of them (PFOFF = PF_ROWS rows + piece - this wave's
        column, relative to ROW).  Skipped for the last PF_ROWS steps (nothing behind the key)."""
        e = self.e
        if PF_ROWS == 0: return
        self.nlab += 1
        skip = "L_pf%d_%%=" % self.nlab
        e("s_add_u32", SG.TMP, SG.I, Lit(PF_ROWS))
        e("s_cmp_lt_u32", SG.TMP, SG.NSTEP)
        e("s_cbranch_scc0", skip)
        e("s_lshr_b32", SG.TMP, FSLOT, Lit(4))
        e("s_add_u32", SG.CTV.lo(), SG.ROW.lo(), SG.TMP)
        e("s_addc_u32", SG.CTV.hi(), SG.ROW.hi(), Lit(0))
        e("s_add_u32", M0, TILE0, Lit(PF_LDS))
        e("s_nop", Lit(0))
        e("global_load_lds_dwordx4", L16, SG.CTV, offset=0)
        self.vmq.append(set())
        self.p.label(skip)

    def fair_finish(self):
        """this workgroup no longer competes (a word of 0xFFFF steps is never the minimum)"""
        e = self.e
        e("s_setprio", Lit(0))
        e("s_or_b32", SG.TMP, FEPOCH, Lit(0xFFFF))
        e("s_and_b32", SG.TMP2, FSLOT, Lit(15))
        e("v_mov_b32", T0.lo(), SG.TMP2)
        e("v_mov_b32", T0.hi(), SG.TMP)
        e("global_store_dword", T0.lo(), T0.hi(), FTAB, offset=0)

    def step(self):
        e = self.e
        self.p.label("L_step%=")
        self.setprio("R")
        self.c("---- modswitch(ct[i]) -> rot; base = (lane - rot) mod 2N")
        self.p.emit("s_waitcnt", vm=0); self.vmq.clear()
        if self.soft:
            if self.cu_fair: self.soft_fair_publish()
        elif self.cu_fair: self.fair_step()
        self.key_prefetch()
        e("v_readfirstlane_b32", SG.CTV.hi(), R11)
        e("s_nop", Lit(3))
        e("s_lshr_b32", SG.TMP, SG.CTV.hi(), Lit(23))           # x >> 55
        e("s_and_b32", SG.TMP2, SG.TMP, Lit(1))
        e("s_add_u32", SG.TMP, SG.TMP, SG.TMP2)
        e("s_lshr_b32", SG.TMP, SG.TMP, Lit(1))
        e("s_lshl_b32", SG.ROT, SG.TMP, Lit(3))
        e("v_subrev_u32", T0.lo(), SG.ROT, LANE)
        e("v_and_b32", T0.lo(), Lit(2 * N - 1), T0.lo())        # base
        e("v_lshrrev_b32", R10, Lit(6), T0.lo())                # hb
        e("v_and_b32", R11, Lit(63), T0.lo())
        e("v_lshl_add_u32", R11, R11, Lit(3), SG.TILE)          # tile + (base & 63) * 8
        for m in range(16):
            t = T0.hi()
            e("v_add_u32", t, Lit(m + (16 if self.neg_acc else 0)), R10)      # (+ 16: the mask inverted, for the negated accumulator)
            e("v_bfe_i32", SGN3[m], t, Lit(4), Lit(1))          # 0 / -1: coefficient comes from the negated half
            e("v_and_b32", t, Lit(15), t)
            e("v_lshl_add_u32", RA3[m], t, Lit(9), R11)
        if self.soft and self.cu_fair: self.soft_fair_apply()       # (the counters requested at the loop head have landed)
        # next mask element (index min(i+1, n-1)), a step ahead of its use
        e("s_add_u32", SG.TMP, SG.I, Lit(1))
        e("s_cmp_lt_u32", SG.TMP, SG.NSTEP)
        e("s_cbranch_scc1", "L_nx%=")
        e("s_mov_b32", SG.TMP, SG.I)
        self.p.label("L_nx%=")
        e("s_lshl_b32", SG.TMP, SG.TMP, Lit(3))
        e("v_mov_b32", R10, SG.TMP)
        self.gload("global_load_dwordx2", V(10, 2), R10, SG.CT, offset=0)
        # ---- key ring: this wave's column, 24 key quads (q = 0, 1, 2 x 8 registers) through 12 register quads.  Eleven are
        # issued ahead of the multiply-accumulate into the quads that are free at the time: X9..11 at the start of the step,
        # X0..7 before the last forward pass (twiddles done), X8 (the spare) at the multiply-accumulate.
        NG, EARLY = 12, 11
        stream = [(q, r) for q in range(3) for r in range(8)]
        kps = (SG.KP0, SG.KP1, SG.KP2)
        ring = [quad(X3, r) for r in [9, 10, 11, 0, 1, 2, 3, 4, 5, 6, 7, 8]]

        def issue(k):
            q, r = stream[k]
            self.key_quad(ring[k % NG], r, kps[q])

        def early(lo, hi):
            self.c("key ring: elements %d..%d, issued ahead of their use" % (lo, hi - 1))
            if lo == 0:
                for q in range(3):       # ROW points at column c of row i: key[q][c] is q * 3 polynomials further
                    e("s_add_u32", kps[q].lo(), SG.ROW.lo(), Lit(q * 3 * 8192 + self.key_bias()))
                    e("s_addc_u32", kps[q].hi(), SG.ROW.hi(), Lit(0))
            for k in range(lo, hi): issue(k)
        early(0, 3)
        # ---- rotation: stage the polynomial in the tile (natural order), gather the rotated words
        self.c("---- rotation: stage this wave's polynomial in its tile, gather rotated coefficients")
        if self.xlo_regs: self.barrier(3)         # the neighbours have read the transform published in this tile a step ago
        for m in range(16): self.ds_write("ds_write_b64", L8T, V(A3 + 2 * m, 2), offset=512 * m)
        for m in range(8):
            self.ds_read("ds_read_b64", V(D3 + 4 * m, 2), RA3[m])
            self.ds_read("ds_read_b64", V(D3 + 4 * m + 2, 2), RA3[m + 8])
        self.c("---- digits: ct1 = acc X^rot - acc, one level (B = 23), as f64")
        temps = [T0, T1, V(X3, 2), V(X3 + 2, 2)] if os.environ.get("BR_DIGIT", "short") == "short" else [T0.lo(), T0.hi(), T1.lo(), T1.hi()]   # the ring's quad 0 is idle here
        chains = []
        for m in range(8):
            for half in range(2):
                j = m + 8 * half
                slot = V(D3 + 4 * m + 2 * half, 2)
                chains.append(self.chain(self.digit_pair, slot, V(A3 + 2 * j), V(A3 + 2 * j + 1), SGN3[j], temps[len(chains) % 4], CARRY[len(chains) % 4]))
                if len(chains) == 4:
                    self.merge(chains); chains = []
        self.stamp(0)
        # ---- forward transform of this wave's digit polynomial
        self.setprio("F")
        xq = [quad(X3, r) for r in range(12)]
        v, spare = self.fwd1([quad(D3, r) for r in range(8)], xq, before_pass3=lambda: early(3, EARLY))
        self.stamp(1)
        # ---- publish it: [register r][lane] complex in this wave's tile (idle until the inverse)
        self.c("---- publish the transform; barrier; F = sum_q D_q * key[q][c] with D_q from the three tiles")
        if "P" not in ABL3:
            for r in range(8): self.ds_write("ds_write_b128", L16T, v[r], offset=1024 * r)
        self.barrier(2)
        self.setprio("M")
        for k in range(EARLY, NG): issue(k)
        # the published transforms land in a ring of ten quads (D is dead: the stores above captured their data)
        land = [quad(D3, r) for r in range(8)] + [quad(E3, 0), quad(E3, 1)]
        NL = len(land)

        def fetch(k):
            q, r = stream[k]
            if "P" in ABL3: return
            self.ds_read("ds_read_b128", land[k % NL], L16T0, offset=q * TILE_STRIDE + 1024 * r)
        for k in range(NL): fetch(k)
        Fq = [quad(F3, r) for r in range(8)]

        def mac(k):
            q, r = stream[k]
            g, d = ring[k % NG], land[k % NL]
            re0 = Lit(0, f64=0.0) if q == 0 else qre(Fq[r])
            im0 = Lit(0, f64=0.0) if q == 0 else qim(Fq[r])
            self.e("v_fma_f64", qre(Fq[r]), qre(d), qre(g), re0)
            self.e("v_fma_f64", qim(Fq[r]), qre(d), qim(g), im0)
            self.e("v_fma_f64", qre(Fq[r]), -qim(d), qim(g), qre(Fq[r]))
            self.e("v_fma_f64", qim(Fq[r]), qim(d), qre(g), qim(Fq[r]))
        for k in range(0, len(stream), 2):
            self.merge([self.chain(mac, k), self.chain(mac, k + 1)])
            for kk in (k, k + 1):
                if kk + NG < len(stream): issue(kk + NG)
                if kk + NL < len(stream): fetch(kk + NL)
        if self.cu_fair and not self.soft: self.fair_request()  # (the landing quads of the published transforms are free again)
        self.stamp(2)
        # ---- inverse; the tiles are free for its exchanges once every wave has read what it needs
        self.setprio("I")
        v, sp2 = self.inv1_hooked(Fq, xq, (lambda: None) if self.xlo_regs else (lambda: self.barrier(3)))
        self.stamp(3)
        self.setprio("U")
        self.c("---- this wave's polynomial += round(inverse)")
        tmp = [(V(X3 + 4 * k, 2), V(X3 + 4 * k + 2, 2)) for k in range(4)]

        def upd(m, half, k):
            j = m + 8 * half
            x = V(v[m].i + 2 * half, 2)
            al, ah = V(A3 + 2 * j), V(A3 + 2 * j + 1)
            self.torus_add(x, al, ah, al, ah, tmp[k][0], tmp[k][1], CARRY[k], neg=self.neg_acc)
        chains = []
        for m in range(8):
            for half in range(2):
                chains.append(self.chain(upd, m, half, len(chains)))
                if len(chains) == 4:
                    self.merge(chains); chains = []
        self.stamp(4)
        # ---- loop control
        self.lg_wait_all()
        e("s_add_u32", SG.ROW.lo(), SG.ROW.lo(), Lit(KP * KP * H * 16))
        e("s_addc_u32", SG.ROW.hi(), SG.ROW.hi(), Lit(0))
        e("s_add_u32", SG.I, SG.I, Lit(1))
        e("s_cmp_lt_u32", SG.I, SG.NSTEP)
        if "K" in brgen.ABL: self.vm_wait_all()
        # same state as at the loop head: nothing in flight but (fair mode) the table entry requested for the next step, which
        # the full wait at the loop head -- or the epilogue's -- retires
        assert self.lgq == [] and all(q == self.regs(self.FAIRQ) for q in self.vmq) and len(self.vmq) <= 1, (self.vmq, self.lgq)
        e("s_cbranch_scc1", "L_step%=")

    def epilogue(self):
        e = self.e
        self.c("---- epilogue: store this wave's polynomial (OUT points at it)")
        self.vm_wait_all()
        if self.soft:
            if self.cu_fair: self.soft_fair_finish()
        elif self.cu_fair: self.fair_finish()
        e("v_lshlrev_b32", L8, Lit(3), LANE)                    # (register 9: the published-transform address is dead)
        e("v_add_u32", T0.lo(), Lit(4096), L8)
        for m in range(16):
            if self.neg_acc: self.negate64(V(A3 + 2 * m, 2), CARRY[m % 4])       # back to acc
            self.gstore("global_store_dwordx2", L8 if m < 8 else T0.lo(), V(A3 + 2 * m, 2), SG.OUT, offset=512 * (m & 7))
        self.vm_wait_all()
        if self.stamps:
            e("v_mov_b32", T0.hi(), Lit(0))
            for k in range(NSTAMP):
                e("v_mov_b32", T1.lo(), ST_ACC[k].lo()); e("v_mov_b32", T1.hi(), ST_ACC[k].hi())
                self.gstore("global_store_dwordx2", T0.hi(), T1, ST_DBG, offset=8 * k)
            self.vm_wait_all()


def render_inc_soft(prog):
    """soft-barrier form: %0 ct (64), %1 column c of bsk row 0 (64), %2 out + c N (64), %3 n, %4 this wave's tile, %5 the unit's first
    tile, %6 LDS address of the unit's arrival words, %7 of this wave's word, %8 of the workgroup's step counters, %9 of this unit's
    counter, %10 the tie mode, %11 the context's fault word (64)"""
    head = ["s_mov_b64 %s, %%0" % SG.CT, "s_mov_b64 %s, %%1" % SG.ROW, "s_mov_b64 %s, %%2" % SG.OUT,
            "s_mov_b32 %s, %%3" % SG.NSTEP, "s_mov_b32 %s, %%4" % SG.TILE, "s_mov_b32 %s, %%5" % TILE0,
            "s_mov_b32 %s, %%6" % SFLAGS, "s_mov_b32 %s, %%7" % SMYFLAG, "s_mov_b32 %s, %%8" % SPROG, "s_mov_b32 %s, %%9" % SMYPROG,
            "s_mov_b32 %s, %%10" % SG.TIEMODE, "s_mov_b64 %s, %%11" % SFAULT]
    lines = ["  " + h for h in head] + prog.render()
    out = ["// generated by tools/asmgen/brgen3.py (soft-barrier form) -- do not edit"]
    for l in lines:
        out.append('"%s\\n"' % l.replace('"', '\\"'))
    return "\n".join(out) + "\n"


def render_inc(prog, stamps=False):
    """operands: %0 ct (64), %1 column c of bsk row 0 (64), %2 out + c N (64), %3 n, %4 this wave's tile, %5 wave 0's tile,
    %6 this CU's entry of the fairness table (64), %7 byte offset of this workgroup's word in it | key prefetch offset << 4,
    %8 launch number << 16, %9 the tie mode (1 = balanced top-level tie, 0 = tfhe-rs' rule);
    stamps: %10 the debug buffer (64)"""
    head = ["s_mov_b64 %s, %%0" % SG.CT, "s_mov_b64 %s, %%1" % SG.ROW, "s_mov_b64 %s, %%2" % SG.OUT,
            "s_mov_b32 %s, %%3" % SG.NSTEP, "s_mov_b32 %s, %%4" % SG.TILE, "s_mov_b32 %s, %%5" % TILE0,
            "s_mov_b64 %s, %%6" % FTAB, "s_mov_b32 %s, %%7" % FSLOT, "s_mov_b32 %s, %%8" % FEPOCH, "s_mov_b32 %s, %%9" % SG.TIEMODE]
    if stamps: head.append("s_mov_b64 %s, %%10" % ST_DBG)
    lines = ["  " + h for h in head] + prog.render()
    out = ["// generated by tools/asmgen/brgen3.py -- do not edit"]
    for l in lines:
        out.append('"%s\\n"' % l.replace('"', '\\"'))
    return "\n".join(out) + "\n"


def render_clobbers():
    regs = ["v%d" % i for i in range(NV)] + ["s%d" % i for i in range(34, 96)] + ["vcc", "scc"] + (["m0"] if PF_ROWS else []) + ["memory"]
    return "// generated by tools/asmgen/brgen3.py -- do not edit\n" + ", ".join('"%s"' % r for r in regs) + "\n"


if __name__ == "__main__":
    prog = CoopGen(prio=os.environ.get("BR3_PRIO", "")).build()
    if len(sys.argv) > 2 and sys.argv[1] == "--out":
        open(os.path.join(sys.argv[2], "tx_br3_asm.inc"), "w").write(render_inc(prog))
        open(os.path.join(sys.argv[2], "tx_br3s_asm.inc"), "w").write(render_inc_soft(CoopGen(soft=True, prio=os.environ.get("BR3_PRIO", "")).build()))
        open(os.path.join(sys.argv[2], "tx_br3_asm_stamps.inc"), "w").write(render_inc(CoopGen(stamps=True).build(), stamps=True))
        open(os.path.join(sys.argv[2], "tx_br3_asm_clobbers.inc"), "w").write(render_clobbers())
        open(os.path.join(sys.argv[2], "tx_br3_asm_consts.inc"), "w").write(
            "// generated by tools/asmgen/brgen3.py -- do not edit\n"
            "constexpr int BR6_PF_ROWS = %d;      // the loop touches the key row this many steps ahead (and skips the last ones)\n" % PF_ROWS)
    elif len(sys.argv) > 1 and sys.argv[1] == "--count":
        c = prog.count()
        for k in sorted(c, key=lambda k: -c[k]): print("%6d %s" % (c[k], k))
        print("%6d total" % sum(c.values()))
    else:
        sys.stdout.write(render_inc(prog))
