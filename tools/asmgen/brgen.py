"""Generator of the blind-rotation CMux loop as gfx950 assembly (two waves per SIMD, 256 VGPRs per wave).

hipcc cannot hold this kernel's working set in 256 registers (it spills 50-100 registers to scratch and the
scratch traffic shares the vector-memory counter with the key stream), so the loop is emitted with a fixed
register map, explicit software pipelining and counted waits.  The arithmetic is, operation for operation, the
one of k_blind_rotate / the CPU oracle (oracle/txo_fft.c, txo_core.c): same butterfly network, same fma
placement, same summation order -- outputs are bit-identical.  tests/asmgen_emu_blind_rotate.py runs the emitted
stream in a lane-exact emulator (gcn.py) against the oracle's primitives; tests/test_gpu_parity.py compares the
kernel with the oracle on the GPU.

    python3 tools/asmgen/brgen.py > temp-fhe-transciphering_amd/csrc/tx_br_asm.inc

Register map (VGPR):
    0 lane          1 tile + lane*8      2 tile + lane*16     3 lane*16 (twiddle row / key offset)
    4 lane*16+4096  5 pass-B twiddle     6 xchg_hi store      7 xchg_lo store
    8 e = 1-(lane&1)                     9 lane*8 (epilogue)  10,11 next mask element / hb, lb8
    12:13, 14:15 temporaries T0, T1
    16..47 A0   48..79 A1   (mask polynomials of the accumulator, 16 x u64 each)
    80..111 D0  112..143 D1  144..175 D2   (digit polynomials -> their transforms, 8 complex each)
    176..207 F  (Fourier accumulator of one column; rotation addresses + sign masks before the MAC)
    208..255 X  (key ring in the MAC; twiddles + spare quad in the transforms; body words in the rotation)
SGPR: see `SG` below.  LDS image: tx_br2.hpp (twiddles at 0, per wave: 8 320-byte tile of pitch 65 complex + 8 KiB body).
"""
import os
import struct
import sys
from gcn import V, S, Lit, VCC, Prog, VCC_LO, VCC_HI, VCC64
from fftconst import C16

N, LOGN, H, KP = 1024, 10, 512, 3
TWB = 8 * 64 * 16                       # byte offset of the compact pass-B twiddles (after the eight pass-A rows)
PITCH = 65 * 16                         # bytes per row of the transpose tile: 64 complex + one of padding (XPITCH of tx_device.hpp)
BODY = 8 * PITCH                        # the body polynomial (blind rotation) / the GLWE (trace) follows the 8 320-byte tile

# ---- register map
LANE, L8T, L16T, L16, OFF1, TWBA, XHW, XLW, EREG, L8, R10, R11 = [V(i) for i in range(12)]
PARTNER = V(9)      # fair mode: the partner wave's step counter, requested a step ahead (lane*8 lives here only in the epilogue)
T0, T1 = V(12, 2), V(14, 2)
A = [16, 48]
D = [80, 112, 144]
F = 176
X = 208
RA = [V(176 + m) for m in range(16)]        # rotation addresses (phase R only)
SGN = [V(192 + m) for m in range(16)]       # sign masks (phase R only)


class SG:
    CT = S(34, 2); ROW = S(36, 2); OUT = S(38, 2)
    NSTEP = S(40); I = S(41); TILE = S(42); ROT = S(43); CTV = S(44, 2); TMP = S(46); TMP2 = S(47)
    C_2M64 = S(48, 2); C_M2P64 = S(50, 2); C_84 = S(52, 2); C_52 = S(54, 2); C_SQH = S(56, 2); C_M4338 = S(58)
    KP0 = S(60, 2); KP1 = S(62, 2); KP2 = S(64, 2); KPN = S(66, 2)
    CS = S(44, 2); CC = S(46, 2)       # twist constants (sin, cos), set where they are used: they share CTV / TMP:TMP2, which are dead inside a transform
    FAIR = S(68); PROGA = S(69); PROGB = S(59); PBASE = S(67)   # fair mode: flag, LDS addresses of my / the partner's step counter, steps done before this unit
    TIEMODE = S(46)      # operand, read by the prologue only (= TMP, free until then): 1 = balanced top-level tie, 0 = tfhe-rs' rule


def f64bits(x): return struct.unpack("<Q", struct.pack("<d", x))[0]


def quad(base, r): return V(base + 4 * r, 4)
def qre(q): return V(q.i, 2)
def qim(q): return V(q.i + 2, 2)


class Cx:
    """view of a complex value: each component is a 64-bit register pair with an optional negation"""
    def __init__(self, re, im, q=None): self.re, self.im, self.q = re, im, q
    @staticmethod
    def of(q): return Cx(qre(q), qim(q), q)
    def mul_i(self): return Cx(-self.im, self.re, self.q)         # * i   (a view: same quad)
    def mul_mi(self): return Cx(self.im, -self.re, self.q)        # * -i


CARRY = [S(88, 2), S(90, 2), S(92, 2), S(94, 2)]      # one carry mask per interleaved integer chain (VOP3 carry operands)
ABL = os.environ.get("BR_ABL", "")      # timing-only ablations (K keys, X exchanges, T twiddle loads, M MAC arithmetic): outputs invalid
NSTAMP = 6
ST_ACC = [S(70 + 2 * k, 2) for k in range(NSTAMP)]
ST_PREV, ST_NOW, ST_DBG = S(82, 2), S(84, 2), S(86, 2)


class Gen:
    CC, CS = SG.CC, SG.CS

    def __init__(self, stamps=False, prio="I", deep_prefetch=True, phase_stamps=True, fair=False):
        self.stamps = stamps
        # fair: the two waves of a SIMD (waves w and w + 4 of a 512-thread workgroup) exchange their step counters
        # through LDS every step; the one that is behind runs at a higher priority.  Without it the older wave wins
        # every arbitration (25 k against 40 k cycles per step), finishes far ahead, and the younger one ends alone.
        self.fair = fair
        self.fair_h = int(os.environ.get("BR_FAIR_H", 8))
        self.nlab = 0
        self.deep_prefetch = deep_prefetch      # False: the 8-of-12 key prefetch of the first two-wave versions (experiments)
        self.phase_stamps = phase_stamps        # False with stamps=True: only the clock marks at the start and the end of a unit
        # phases that run at s_setprio 1 ("R", "F", "M", "I", "U").  The single-polynomial inverse is the phase whose
        # exchange round trips nothing covers: a wave in it goes first when both waves of a SIMD have an instruction
        # ready (8192 units: 47.4-47.6 against 48.4-49.4 ms; MI the same, F / RU less)
        self.prio = prio
        self.p = Prog()
        self.buf = None
        self.vmq = []          # in-flight vector-memory ops, oldest first: sets of destination registers
        self.lgq = []
        self.valu_w = [set(), set()]   # VGPRs written by VALU instructions one / two wait states ago (cross-lane hazards)
        self.xhi_regs = False  # True: exchange hi (register index <-> lane bits 5:3) in registers instead of through the tile
        self.xlo_regs = False  # True: exchange lo (register index <-> lane bits 2:0) in registers as well
        # The accumulator is kept NEGATED inside the loop (round 4).  A CMux step needs d = +-acc[j + rot] - acc[j] and
        # acc[j] += round(x): with acc' = -acc these are d = -+acc'[j + rot] + acc'[j] and acc'[j] += round(-x) -- both plain 64-bit
        # ADDS (one v_lshl_add_u64 each: 4 cycles, where the v_sub_co_u32 / v_subb_co_u32 and v_add_co_u32 / v_addc_co_u32 pairs
        # cost 8), the first with the gather's sign mask inverted (free: the literal of its index add) and the second with the
        # negation on the source modifiers of the rounding's first two operations.  The polynomials are negated once on the way in
        # and once on the way out (per unit, outside the loop).  Same integers mod 2^64 at every step: bit-identical.
        self.neg_acc = True
        # e_reg: VGPR 4 holds e = 1 on even lanes (digit_pair's "+ e" is then a plain v_add_u32: 2 cycles where the carry-in form
        # took 4).  The register was lane * 16 + 4096, the offset of key rows 4..7: the key pointers are biased by 4096 instead and
        # all eight rows use lane * 16 with immediate offsets -4096 .. 3072
        self.e_reg = V(4) if os.environ.get("BR_EREG", "1") == "1" else None

    # ---------------------------------------------------------------- emission + counters
    # Every emitted instruction goes through e(): registers it touches that are the destination of a load still in
    # flight get the (counted) wait in front of it, and loads / stores are entered in the two in-order queues.  So
    # waits always sit as late as the program order allows, and reordering the emission can never lose one.
    def e(self, op, *a, **kw):
        if self.buf is not None:
            self.buf.append((op, a, kw)); return
        if op not in ("s_waitcnt", "label", "comment"):
            touched = set()
            for x in a:
                if isinstance(x, V): touched |= self.regs(x)
            if touched:
                k = self._wait(self.vmq, touched, 63)
                if k is not None: self.p.emit("s_waitcnt", vm=k)
                k = self._wait(self.lgq, touched, 15)
                if k is not None: self.p.emit("s_waitcnt", lgkm=k)
        # cross-lane instructions (v_permlane*_swap, DPP): 2 wait states after a VALU write of a VGPR they read; nothing inside an
        # asm string is padded by the assembler, so the s_nop is emitted here
        if "_swap_" in op or op.endswith("_dpp"):
            rd = set()
            for x in (a if "_swap_" in op else a[1:2]):
                if isinstance(x, V): rd |= self.regs(x)
            if rd & self.valu_w[0]: self.p.emit("s_nop", Lit(1)); self.valu_w = [set(), set()]
            elif rd & self.valu_w[1]: self.p.emit("s_nop", Lit(0)); self.valu_w = [set(), self.valu_w[0]]
        # v_readfirstlane / v_readlane of a VGPR the PREVIOUS instruction wrote: 1 wait state on gfx940+ (it would read the old
        # value -- round 5, found on the hardware: a poll that compared a register with itself)
        if op in ("v_readfirstlane_b32", "v_readlane_b32") and isinstance(a[1], V) and self.regs(a[1]) & self.valu_w[0]:
            self.p.emit("s_nop", Lit(0)); self.valu_w = [set(), self.valu_w[0]]
        wrote = set()
        if op.startswith("v_") and a and isinstance(a[0], V):
            wrote = self.regs(a[0])
            if "_swap_" in op: wrote |= self.regs(a[1])
        if op not in ("label", "comment"):
            nws = (a[0].v + 1) if op == "s_nop" else 1
            for _ in range(min(nws, 2)): self.valu_w = [wrote, self.valu_w[0]]; wrote = set()
        self.p.emit(op, *a, **kw)
        if op.startswith("ds_read"): self.lgq.append(self.regs(a[0]))
        elif op.startswith("ds_write"): self.lgq.append(set())
        elif op in ("s_memtime", "s_memrealtime"): self.lgq.append(set())
        elif op.startswith("global_load_lds"): pass          # LDS-DMA: no destination register; the caller tags it
        elif op.startswith("global_load"): self.vmq.append(self.regs(a[0]))
        elif op.startswith("global_store"): self.vmq.append(set())

    def c(self, t):
        if self.buf is None: self.p.comment(t)

    def chain(self, fn, *args):
        """record the instructions fn(*args) emits, without processing them (re-entrant: a recorded function may
        itself record and merge chains; what it merges lands in the outer recording)"""
        saved, self.buf = self.buf, []
        fn(*args)
        b, self.buf = self.buf, saved
        return b

    def merge_spaced(self, main, sparse):
        """emit chain `main` with the instructions of `sparse` spread evenly through it"""
        main, sparse = list(main), list(sparse)
        k = max(1, len(main) // (len(sparse) + 1))
        n = 0
        for op, a, kw in main:
            self.e(op, *a, **kw)
            n += 1
            if sparse and n % k == 0:
                o2, a2, k2 = sparse.pop(0)
                self.e(o2, *a2, **k2)
        for o2, a2, k2 in sparse: self.e(o2, *a2, **k2)

    def merge(self, chains):
        """emit independent instruction chains round-robin (in-order issue: the neighbours cover each other's latency)"""
        chains = [list(c) for c in chains if c]
        while chains:
            for c in list(chains):
                op, a, kw = c.pop(0)
                self.e(op, *a, **kw)
                if not c: chains.remove(c)

    def _wait(self, q, regs, limit):
        regs = set(regs)
        idx = -1
        for k, r in enumerate(q):
            if r & regs: idx = k
        if idx < 0: return None
        keep = min(len(q) - (idx + 1), limit)
        del q[:len(q) - keep]
        return keep

    def negate64(self, pair, cy):
        """pair <- -pair (two's complement), carry mask cy"""
        self.e("v_sub_co_u32", pair.lo(), cy, Lit(0), pair.lo())
        self.e("v_subb_co_u32", pair.hi(), cy, Lit(0), pair.hi(), cy)

    def even_lanes_to_vcc(self):
        """vcc = the lanes with e = 1: the carry-in mask of digit_pair's "+ e" (nothing else in the loops uses vcc) -- the even
        lanes under the balanced rule, every lane under tfhe-rs' rule (TIEMODE = 0)"""
        self.e("s_cmp_eq_u32", SG.TIEMODE, Lit(0))
        self.e("s_cselect_b32", VCC_LO, Lit(-1), Lit(0x55555555)); self.e("s_cselect_b32", VCC_HI, Lit(-1), Lit(0x55555555))      # (-1 is an inline constant: one literal per instruction)

    def tie_lane(self, dst, mode=None):
        """dst = lane & TIEMODE: the parity bit that sends a top-level tie to -B/2 (balanced rule: odd lanes = odd coefficient
        indices; tfhe-rs' rule: nobody).  TIEMODE arrives as an operand in a scratch register: nothing may have written it yet."""
        mode = mode or SG.TIEMODE
        for ins in self.p.ins:
            assert not (ins.op.startswith("s_") and ins.a and isinstance(ins.a[0], S) and ins.a[0].i <= mode.i < ins.a[0].i + ins.a[0].n), \
                "the tie-mode operand register was overwritten before the prologue read it: " + ins.text()
        self.e("v_and_b32", dst, mode, LANE)

    def vm_wait(self, regs): pass        # kept for call sites: e() places every wait
    def lg_wait(self, regs): pass

    def lg_wait_all(self):
        if self.lgq:
            self.p.emit("s_waitcnt", lgkm=0); self.lgq.clear()

    def vm_wait_all(self):
        if self.vmq:
            self.p.emit("s_waitcnt", vm=0); self.vmq.clear()

    @staticmethod
    def regs(v): return set(range(v.i, v.i + v.n))

    def ds_read(self, op, d, addr, offset=0): self.e(op, d, addr, offset=offset)
    def ds_write(self, op, addr, src, offset=0): self.e(op, addr, src, offset=offset)
    def gload(self, op, d, voff, sbase, offset=0):
        if "K" in ABL and op == "global_load_dwordx4": return          # timing experiment: results are garbage
        self.e(op, d, voff, sbase, offset=offset)
    def gstore(self, op, voff, src, sbase, offset=0, cache=None):
        if cache: self.e(op, voff, src, sbase, offset=offset, cache=cache)
        else: self.e(op, voff, src, sbase, offset=offset)

    def setprio(self, phase):
        hi = 1 if phase in self.prio else 0
        if not self.fair:
            if self.prio: self.e("s_setprio", Lit(hi))
            return
        # behind (FAIR != 0): priority 2 + phase bit; level or ahead: 0 + phase bit
        self.nlab += 1
        la, lb = "L_pa%d_%%=" % self.nlab, "L_pb%d_%%=" % self.nlab
        self.e("s_cmp_lg_u32", SG.FAIR, Lit(0))
        self.e("s_cbranch_scc1", la)
        self.e("s_setprio", Lit(hi))
        self.e("s_branch", lb)
        self.p.label(la)
        self.e("s_setprio", Lit(2 + hi))
        self.p.label(lb)

    def fair_exchange(self):
        """step start: take the partner's counter requested a step ago, publish mine, request the next one"""
        e = self.e
        e("v_readfirstlane_b32", SG.TMP, PARTNER)                # e() waits for the pending ds_read
        e("s_nop", Lit(3))
        # hysteresis: priority flips rarely -- a wave that has fallen FAIR_H steps behind goes first until it has caught
        # up; in between the hardware's own rule (oldest first) runs undisturbed, which is the efficient regime for a pair
        e("s_add_u32", SG.TMP2, SG.PBASE, SG.I)                  # steps this wave has done since the kernel started
        e("s_sub_u32", SG.TMP, SG.TMP, SG.TMP2)                  # partner - me (as a signed number)
        self.nlab += 1
        l_set, l_end = "L_fs%d_%%=" % self.nlab, "L_fe%d_%%=" % self.nlab
        e("s_cmp_lg_u32", SG.FAIR, Lit(0))
        e("s_cbranch_scc0", l_set)
        e("s_cmp_gt_i32", SG.TMP, Lit(0))                        # still behind? keep the priority
        e("s_cbranch_scc1", l_end)
        e("s_mov_b32", SG.FAIR, Lit(0))
        e("s_branch", l_end)
        self.p.label(l_set)
        e("s_cmp_gt_i32", SG.TMP, Lit(self.fair_h))
        e("s_cbranch_scc0", l_end)
        e("s_mov_b32", SG.FAIR, Lit(1))
        self.p.label(l_end)
        e("v_mov_b32", T0.lo(), SG.TMP2)
        e("v_mov_b32", T0.hi(), SG.PROGA)
        e("ds_write_b32", T0.hi(), T0.lo(), offset=0)
        e("v_mov_b32", T1.lo(), SG.PROGB)
        e("ds_read_b32", PARTNER, T1.lo(), offset=0)

    def clock_mark(self, slot):
        """diagnostic build: s_memtime and s_memrealtime (100 MHz) to dbg[slot], dbg[slot + 1]: the host derives the shader clock"""
        e = self.e
        self.lg_wait_all()
        for k, op in enumerate(("s_memtime", "s_memrealtime")):
            e(op, ST_NOW)
            self.lg_wait_all()
            e("v_mov_b32", T0.hi(), Lit(0))
            e("v_mov_b32", T1.lo(), ST_NOW.lo()); e("v_mov_b32", T1.hi(), ST_NOW.hi())
            e("global_store_dwordx2", T0.hi(), T1, ST_DBG, offset=8 * (slot + k))

    SUB = os.environ.get("GEN_SUBSTAMPS", "")     # "inv3" / "fwd3" / "inv1": the stamps go inside that transform (accumulators 0..3), everything else to 5

    def substamp(self, where, k):
        if self.SUB == where: self.stamp(k, raw=True)

    def stamp(self, k, raw=False):
        """diagnostic build only: cycles since the previous stamp are added to accumulator k (s_memtime returns
        through lgkmcnt, so the LDS queue is drained first: the stamps perturb what they measure)"""
        if not self.stamps or not self.phase_stamps: return
        if self.SUB and not raw and k is not None: k = 5
        self.lg_wait_all()
        self.e("s_memtime", ST_NOW)
        self.lg_wait_all()
        if k is not None:
            self.e("s_sub_u32", SG.TMP, ST_NOW.lo(), ST_PREV.lo())
            self.e("s_subb_u32", SG.TMP2, ST_NOW.hi(), ST_PREV.hi())
            self.e("s_add_u32", ST_ACC[k].lo(), ST_ACC[k].lo(), SG.TMP)
            self.e("s_addc_u32", ST_ACC[k].hi(), ST_ACC[k].hi(), SG.TMP2)
        self.e("s_mov_b32", ST_PREV.lo(), ST_NOW.lo())
        self.e("s_mov_b32", ST_PREV.hi(), ST_NOW.hi())

    # ---------------------------------------------------------------- complex helpers (bit-exact op order)
    def cadd_to(self, dst, x, y):      # dst = x + y   (dst quad; x, y views)
        self.e("v_add_f64", qre(dst), x.re, y.re)
        self.e("v_add_f64", qim(dst), x.im, y.im)

    def csub_to(self, dst, x, y):
        self.e("v_add_f64", qre(dst), x.re, -y.re)
        self.e("v_add_f64", qim(dst), x.im, -y.im)

    def bfly(self, x, y, spare):
        """(x, y) -> (x + y in x's quad, x - y in `spare`); returns (sum, diff, freed quad of y or None).
        x must be a plain view of its quad; y may be any view."""
        assert x.q is not None and not x.re.neg and not x.im.neg
        self.csub_to(spare, x, y)
        self.cadd_to(x.q, x, y)
        return Cx.of(x.q), Cx.of(spare), y.q

    def cmul(self, q, w, tmp):         # q *= w        cmul(): t = y*wy; u = y*wx; re = fma(x, wx, -t); im = fma(x, wy, u)
        self.cmul_w(q, qre(w), qim(w), tmp)

    def cmulc(self, q, w, tmp, tmp2):  # q *= conj(w)  cmulc(): t = y*wy; u = x*wy; re = fma(x, wx, t); im = fma(y, wx, -u)
        self.cmulc_w(q, qre(w), qim(w), tmp, tmp2)

    def cmul_w(self, q, wx, wy, tmp):  # the factor's components are any 64-bit operands (register pairs or scalar constants)
        x, y = qre(q), qim(q)
        self.e("v_mul_f64", tmp, y, wy)
        self.e("v_mul_f64", y, y, wx)
        self.e("v_fma_f64", y, x, wy, y)
        self.e("v_fma_f64", x, x, wx, -tmp)

    def cmulc_w(self, q, wx, wy, tmp, tmp2):
        x, y = qre(q), qim(q)
        self.e("v_mul_f64", tmp, y, wy)
        self.e("v_mul_f64", tmp2, x, wy)
        self.e("v_fma_f64", y, y, wx, -tmp2)
        self.e("v_fma_f64", x, x, wx, tmp)

    def cmul_all(self, pairs, spare=None, conj=False):
        """q *= w (or conj(w)) for a list of (q, w); two chains in flight (temporaries: T0, T1; for the conjugate
        form, which needs two each, also the two halves of the free quad `spare`)"""
        tA = (T0, T1) if conj else (T0,)
        tB = (qre(spare), qim(spare)) if conj else (T1,)
        fn = self.cmulc if conj else self.cmul
        for k in range(0, len(pairs), 2):
            chains = [self.chain(fn, pairs[k][0], pairs[k][1], *tA)]
            if k + 1 < len(pairs): chains.append(self.chain(fn, pairs[k + 1][0], pairs[k + 1][1], *tB))
            self.merge(chains)

    def twist_consts(self, polys, spare, conj):
        """register part of the twist (c16() of tx_device.hpp): v[m] *= exp(+-i pi m/16), m = 1..7, for each list of 8
        quads in polys.  exp(i pi m/16) = (c, s) and exp(i pi (8-m)/16) = (s, c): one pair of scalar constants serves two
        registers; m = 4 is (sqrt 1/2, sqrt 1/2), which the butterflies keep anyway.  Two chains in flight."""
        e = self.e
        for mm in (1, 2, 3, 4):
            if mm < 4:
                for sreg, val in ((self.CC, C16[mm][0]), (self.CS, C16[mm][1])):
                    b = f64bits(val)
                    e("s_mov_b32", sreg.lo(), Lit(b & 0xFFFFFFFF)); e("s_mov_b32", sreg.hi(), Lit(b >> 32))
                jobs = []
                for v in polys: jobs += [(v[mm], self.CC, self.CS), (v[8 - mm], self.CS, self.CC)]
            else:
                assert C16[4][0] == C16[4][1] == 0.70710678118654752440
                jobs = [(v[4], SG.C_SQH, SG.C_SQH) for v in polys]
            for k in range(0, len(jobs), 2):
                if conj:
                    chains = [self.chain(self.cmulc_w, *jobs[k], T0, T1)]
                    if k + 1 < len(jobs): chains.append(self.chain(self.cmulc_w, *jobs[k + 1], qre(spare), qim(spare)))
                else:
                    chains = [self.chain(self.cmul_w, *jobs[k], T0)]
                    if k + 1 < len(jobs): chains.append(self.chain(self.cmul_w, *jobs[k + 1], T1))
                self.merge(chains)

    def dft8(self, v, spare, sign, home=None):
        """8-point DFT of the complex values in quads v[0..7] (dft8<sign> of tx_device.hpp, same pairing).
        Returns (list of quads holding outputs 0..7, the new spare quad).  home = (lo, hi): a pass that no exchange follows --
        the quad outside registers lo..hi-1 (the spare on entry) must be the free one on exit: the last butterfly is then the
        one whose first operand sits in it, written as (sum -> spare, difference -> the second operand's quad), which frees
        it (round 4: this took two v_mov_b64 per transform before; same operations, same operands, another destination)."""
        tmp = T0
        a, b = [None] * 4, [None] * 4
        for i in range(4):
            a[i], b[i], fr = self.bfly(Cx.of(v[i]), Cx.of(v[i + 4]), spare)
            spare = fr
        c = SG.C_SQH
        b1, b3 = b[1].q, b[3].q
        # odd branch: u1 = sqrt2 w8 b1, u3 = sqrt2 w8^3 b3 (the factor sqrt(1/2) enters the last butterflies as an fma);
        # each keeps one component in a temporary pair (T0 / T1) until the butterfly that consumes it
        if sign > 0:
            self.e("v_add_f64", T0, qre(b1), -qim(b1))             # u1 = (b1.x - b1.y, b1.x + b1.y)
            self.e("v_add_f64", qim(b1), qre(b1), qim(b1))
            u1 = Cx(T0, qim(b1), b1)
            t2 = b[2].mul_i()                                      # (-b2.y, b2.x)
            self.e("v_add_f64", T1, qre(b3), qim(b3))              # u3 = (-(b3.x + b3.y), b3.x - b3.y)
            self.e("v_add_f64", qim(b3), qre(b3), -qim(b3))
            u3 = Cx(-T1, qim(b3), b3)
        else:
            self.e("v_add_f64", T0, qre(b1), qim(b1))              # u1 = (b1.x + b1.y, b1.y - b1.x)
            self.e("v_add_f64", qim(b1), qim(b1), -qre(b1))
            u1 = Cx(T0, qim(b1), b1)
            t2 = b[2].mul_mi()                                     # (b2.y, -b2.x)
            self.e("v_add_f64", T1, qim(b3), -qre(b3))             # u3 = (b3.y - b3.x, -(b3.x + b3.y))
            self.e("v_add_f64", qim(b3), qre(b3), qim(b3))
            u3 = Cx(T1, -qim(b3), b3)
        out = [None] * 8

        def bfly_scaled(x, y, spare):
            """(x, y) -> (x + c y in x's quad, x - c y in `spare`), c = sqrt(1/2), one fma per component"""
            assert x.q is not None and not x.re.neg and not x.im.neg
            self.e("v_fma_f64", qre(spare), -y.re, c, x.re)
            self.e("v_fma_f64", qim(spare), -y.im, c, x.im)
            self.e("v_fma_f64", qre(x.q), y.re, c, x.re)
            self.e("v_fma_f64", qim(x.q), y.im, c, x.im)
            return Cx.of(x.q), Cx.of(spare), y.q

        def foreign(q): return home is not None and not (home[0] <= q.i < home[1])

        def last_free_x(x, y, spare, scaled):
            """(x, y) -> (x + [c] y in `spare`, x - [c] y in y's quad); frees x's quad.  y must be a plain view of its quad."""
            assert y.q is not None and not y.re.neg and not y.im.neg and y.re.i == qre(y.q).i and y.im.i == qim(y.q).i
            if scaled:
                self.e("v_fma_f64", qre(spare), y.re, c, x.re)
                self.e("v_fma_f64", qim(spare), y.im, c, x.im)
                self.e("v_fma_f64", qre(y.q), -y.re, c, x.re)
                self.e("v_fma_f64", qim(y.q), -y.im, c, x.im)
            else:
                self.cadd_to(spare, x, y)
                self.csub_to(y.q, x, y)
            return Cx.of(spare), Cx.of(y.q), x.q

        def four(x0, x1, x2, x3, o0, o1, o2, o3, spare, scaled=False, final=False):
            # p0 = x0 + x2; q0 = x0 - x2; p1 = x1 + x3; d = x1 - x3; q1 = +-i d; o0 = p0 + p1; o1 = p0 - p1; o2 = q0 + q1; o3 = q0 - q1
            # (scaled: the last four use c p1 and c q1)
            p0, q0, fr = self.bfly(x0, x2, spare); spare = fr
            p1, d, fr = self.bfly(x1, x3, spare); spare = fr
            q1 = d.mul_i() if sign > 0 else d.mul_mi()
            last = bfly_scaled if scaled else self.bfly
            if final and foreign(p0.q):          # the foreign quad holds p0: (q0, q1) first, then (p0, p1) with p0's quad freed
                r2, r3, fr = last(q0, q1, spare); spare = fr
                r0, r1, fr = last_free_x(p0, p1, spare, scaled); spare = fr
            else:
                r0, r1, fr = last(p0, p1, spare); spare = fr
                r2, r3, fr = last(q0, q1, spare); spare = fr
            out[o0], out[o1], out[o2], out[o3] = r0.q, r1.q, r2.q, r3.q
            return spare
        if foreign(a[0].q):           # (a layout whose spare lies below the home region: the foreign quad came in as v[0])
            spare = four(b[0], u1, t2, u3, 1, 5, 3, 7, spare, scaled=True)
            spare = four(a[0], a[1], a[2], a[3], 0, 4, 2, 6, spare, final=True)
        else:
            spare = four(a[0], a[1], a[2], a[3], 0, 4, 2, 6, spare)
            spare = four(b[0], u1, t2, u3, 1, 5, 3, 7, spare, scaled=True, final=True)
        return out, spare

    def home(self, v, spare, lo, hi):
        """after a dft8 that is not followed by an exchange: if an output lives in the foreign spare quad (outside
        registers lo..hi-1), move it into the free quad of the home region so that the foreign quad is free again"""
        for k in range(8):
            if not (lo <= v[k].i < hi):
                assert lo <= spare.i < hi
                self.e("v_mov_b64", qre(spare), qre(v[k]))
                self.e("v_mov_b64", qim(spare), qim(v[k]))
                v[k], spare = spare, v[k]
        return v, spare

    # ---------------------------------------------------------------- LDS tile exchanges (tile of pitch 65 complex, tx_br2.hpp)
    def xchg_hi_write(self, v):
        if "X" in ABL or "w" in ABL: return
        for r in range(8): self.ds_write("ds_write_b128", XHW, v[r], offset=128 * r)

    def xchg_hi(self, v, spare):
        """exchange hi of the eight quads v (element (lane 8a + b, register r) -> (lane 8r + b, register a)); returns the
        eight quads that hold the result and the new spare quad"""
        if self.xhi_regs and not ("X" in ABL): return self.xchg_hi_regs(v, spare)
        self.xchg_hi_write(v)
        free = sorted(v + [spare], key=lambda q: q.i)
        v, spare = free[:8], free[8]
        self.xchg_read(v, swz=False)
        return v, spare

    def swap_reg_bit_with_lane_bit(self, v, spare, step, from_lower, from_upper, mask_clear):
        """pairs (r, r + step) of the eight quads v, r with that register-index bit clear: the lanes with the lane bit SET of the
        lower register trade places with the lanes with it CLEAR of the upper one.  No swap instruction exists for lane bits 3..0,
        so each register of a pair is rebuilt by one v_cndmask_b32 whose src0 is moved by DPP (`from_lower`: lane i reads lane
        i - 2^bit, `from_upper`: lane i reads lane i + 2^bit) -- the first into the spare quad (its partner still needs the old
        value), the second in place; the freed quad is the next pair's spare, and the pairs alternate which side goes first so
        that vcc (mask_clear = the lanes with the bit clear, per 32) flips once per pair.  Returns (v, spare)."""
        e = self.e
        v = list(v)
        e("s_mov_b32", VCC_LO, Lit(mask_clear)); e("s_mov_b32", VCC_HI, Lit(mask_clear))
        low_first = True
        tail = " row_mask:0xf bank_mask:0xf"
        fl = from_lower + tail + (" bound_ctrl:0" if "sh" in from_lower else "")
        fu = from_upper + tail + (" bound_ctrl:0" if "sh" in from_upper else "")
        for r in [r for r in range(8) if not (r & step)]:
            lo, hi = v[r], v[r + step]
            if low_first:        # vcc = bit clear: new lo = own lanes where clear, the upper register's (moved up) where set
                for k in range(4): e("v_cndmask_b32_dpp", V(spare.i + k), V(hi.i + k), V(lo.i + k), VCC, dpp=fl)
                e("s_not_b64", VCC64, VCC64)
                for k in range(4): e("v_cndmask_b32_dpp", V(hi.i + k), V(lo.i + k), V(hi.i + k), VCC, dpp=fu)
                v[r], spare = spare, lo
            else:                # vcc = bit set: new hi first
                for k in range(4): e("v_cndmask_b32_dpp", V(spare.i + k), V(lo.i + k), V(hi.i + k), VCC, dpp=fu)
                e("s_not_b64", VCC64, VCC64)
                for k in range(4): e("v_cndmask_b32_dpp", V(lo.i + k), V(hi.i + k), V(lo.i + k), VCC, dpp=fl)
                v[r + step], spare = spare, hi
            low_first = not low_first
        self.even_lanes_to_vcc()                 # (digit_pair's carry mask)
        return v, spare

    def xchg_hi_regs(self, v, spare):
        """Exchange hi without LDS: the 3-bit register index is swapped with lane bits 5, 4, 3 one bit at a time.  Bit 5:
        v_permlane32_swap (upper half of the first operand <-> lower half of the second); bit 4: v_permlane16_swap (odd rows of
        the first <-> even rows of the second); bit 3: DPP row rotation by 8 (swap_reg_bit_with_lane_bit).  64 vector + 6 scalar
        instructions for what 8 ds_write_b128 + 8 ds_read_b128 and a dependent LDS round trip did.  Same values in the same
        registers of the same lanes: bit-identical by construction (tests/asmgen_emu_blind_rotate3.py runs both forms)."""
        e = self.e
        v = list(v)
        for r in range(4):                                   # register bit 2 <-> lane bit 5
            for k in range(4): e("v_permlane32_swap_b32", V(v[r].i + k), V(v[r + 4].i + k))
        for r in (0, 1, 4, 5):                               # register bit 1 <-> lane bit 4
            for k in range(4): e("v_permlane16_swap_b32", V(v[r].i + k), V(v[r + 2].i + k))
        return self.swap_reg_bit_with_lane_bit(v, spare, 1, "row_ror:8", "row_ror:8", 0x00FF00FF)     # register bit 0 <-> lane bit 3

    def xchg_lo(self, v, spare):
        """exchange lo of the eight quads v (element (lane 8g + b, register r) -> (lane 8g + r, register b)); returns (v, spare)"""
        if self.xlo_regs and not ("X" in ABL): return self.xchg_lo_regs(v, spare)
        self.xchg_lo_write(v)
        free = sorted(v + [spare], key=lambda q: q.i)
        v, spare = free[:8], free[8]
        self.xchg_read(v, swz=True)
        return v, spare

    def xchg_lo_regs(self, v, spare):
        """Exchange lo without LDS: register-index bit j <-> lane bit j, j = 2, 1, 0: row shifts by 4, then the quad permutations
        [2,3,0,1] and [1,0,3,2].  96 vector + 15 scalar instructions."""
        v, spare = self.swap_reg_bit_with_lane_bit(v, spare, 4, "row_shr:4", "row_shl:4", 0x0F0F0F0F)
        v, spare = self.swap_reg_bit_with_lane_bit(v, spare, 2, "quad_perm:[2,3,0,1]", "quad_perm:[2,3,0,1]", 0x33333333)
        return self.swap_reg_bit_with_lane_bit(v, spare, 1, "quad_perm:[1,0,3,2]", "quad_perm:[1,0,3,2]", 0x55555555)

    def xchg_read(self, dst, swz):
        """row r of the tile, this lane's column (both exchanges read the same way; swz is kept for the call sites)"""
        if "X" in ABL or "r" in ABL: return
        for r in range(8): self.ds_read("ds_read_b128", dst[r], L16T, offset=PITCH * r)

    def xchg_lo_write(self, v):
        if "X" in ABL or "w" in ABL: return
        for r in range(8): self.ds_write("ds_write_b128", XLW, v[r], offset=16 * r)

    def load_tw(self, dst, row0, nrows, compact=False):
        """twiddle rows row0.. into quads dst[0..nrows-1]"""
        if "T" in ABL: return
        for k in range(nrows):
            if compact: self.ds_read("ds_read_b128", dst[k], TWBA, offset=128 * k)
            else: self.ds_read("ds_read_b128", dst[k], L16, offset=1024 * (row0 + k))

    # ---------------------------------------------------------------- forward transforms, three polynomials together
    def fwd3(self, polys, xq, before_pass3=None):
        """polys: three lists of 8 quads (digits as doubles); xq: 12 free quads of region X.
        On return polys[p][r] is the quad holding F-layout register r of polynomial p."""
        tw, spare = xq[:8], xq[8]
        home_lo = [min(q.i for q in polys[p]) for p in range(3)]
        self.c("forward x3: twist constants")
        self.substamp("fwd3", 5)
        self.load_tw(tw, 0, 8)                                 # in flight during the constant multiplications
        self.twist_consts(polys, None, conj=False)
        self.substamp("fwd3", 0)
        self.c("forward x3: pass 1 (dft8, pass-A twiddles with the lane part of the twist, exchange hi)")
        for p in range(3):
            polys[p], spare = self.dft8(polys[p], spare, +1)
            self.cmul_all([(polys[p][r], tw[r]) for r in range(8)])
            self.xchg_hi_write(polys[p])
            # the stores captured their data: the eight quads + the spare are free; read back into a canonical set
            free = sorted(polys[p] + [spare], key=lambda q: q.i)
            polys[p], spare = free[:8], free[8]
            self.xchg_read(polys[p], swz=False)
        self.c("forward x3: pass 2 (dft8, pass-B twiddles, exchange lo)")
        self.substamp("fwd3", 1)
        self.load_tw(tw, 0, 7, compact=True)
        for p in range(3):
            self.lg_wait(set().union(*[self.regs(q) for q in polys[p]]))
            polys[p], spare = self.dft8(polys[p], spare, +1)
            self.cmul_all([(polys[p][r], tw[r - 1]) for r in range(1, 8)])
            self.xchg_lo_write(polys[p])
            free = sorted(polys[p] + [spare], key=lambda q: q.i)
            polys[p], spare = free[:8], free[8]
            self.xchg_read(polys[p], swz=True)
        self.c("forward x3: pass 3 (dft8)")
        self.substamp("fwd3", 2)
        if before_pass3: before_pass3()            # the twiddle quads xq[0..7] are free from here on
        for p in range(3):
            self.lg_wait(set().union(*[self.regs(q) for q in polys[p]]))
            polys[p], spare = self.dft8(polys[p], spare, +1, home=(home_lo[p], home_lo[p] + 32))
            polys[p], spare = self.home(polys[p], spare, home_lo[p], home_lo[p] + 32)
        self.substamp("fwd3", 3)
        return polys, spare

    def inv1(self, v, xq):
        """unnormalised inverse of one polynomial (fft_inv); v: 8 quads, xq: free quads of X (>= 9)."""
        tw, spare = xq[:8], xq[8]
        self.c("inverse: pass 3' (dft8, exchange lo)")
        self.load_tw(tw, 0, 7, compact=True)                   # in flight during the butterflies
        v, spare = self.dft8(v, spare, -1)
        self.xchg_lo_write(v)
        free = sorted(v + [spare], key=lambda q: q.i)
        v, spare = free[:8], free[8]
        self.xchg_read(v, swz=True)
        self.lg_wait(set().union(*[self.regs(q) for q in v]))
        self.c("inverse: pass 2' (pass-B twiddles conj, dft8, exchange hi)")
        self.cmul_all([(v[r], tw[r - 1]) for r in range(1, 8)], spare, conj=True)
        self.load_tw(tw, 0, 8)
        v, spare = self.dft8(v, spare, -1)
        self.xchg_hi_write(v)
        free = sorted(v + [spare], key=lambda q: q.i)
        v, spare = free[:8], free[8]
        self.xchg_read(v, swz=False)
        self.lg_wait(set().union(*[self.regs(q) for q in v]))
        self.c("inverse: pass 1' (pass-A twiddles conj, dft8, untwist constants)")
        self.cmul_all([(v[r], tw[r]) for r in range(8)], spare, conj=True)
        home_lo = min(q.i for q in v)
        v, spare = self.dft8(v, spare, -1, home=(home_lo, home_lo + 32))
        v, spare = self.home(v, spare, home_lo, home_lo + 32)
        self.twist_consts([v], spare, conj=True)
        return v, spare

    def inv3(self, polys, xq):
        """unnormalised inverses of three polynomials together (fft_inv_n<3>): polynomial p's tile round trip is
        in flight while the others' butterflies issue.  polys: three lists of 8 quads; xq: >= 9 free quads."""
        tw, spare = xq[:8], xq[8]
        home_lo = [min(q.i for q in polys[p]) for p in range(3)]
        self.c("inverse x3: pass 3' (dft8, exchange lo)")
        self.substamp("inv3", 5)
        self.load_tw(tw, 0, 7, compact=True)
        for p in range(3):
            polys[p], spare = self.dft8(polys[p], spare, -1)
            self.xchg_lo_write(polys[p])
            free = sorted(polys[p] + [spare], key=lambda q: q.i)
            polys[p], spare = free[:8], free[8]
            self.xchg_read(polys[p], swz=True)
        self.c("inverse x3: pass 2' (pass-B twiddles conj, dft8, exchange hi)")
        self.substamp("inv3", 0)
        for p in range(3):
            self.cmul_all([(polys[p][r], tw[r - 1]) for r in range(1, 8)], spare, conj=True)
            if p == 2: self.load_tw(tw, 0, 8)                # the pass-A rows replace the pass-B rows after their last use
            polys[p], spare = self.dft8(polys[p], spare, -1)
            self.xchg_hi_write(polys[p])
            free = sorted(polys[p] + [spare], key=lambda q: q.i)
            polys[p], spare = free[:8], free[8]
            self.xchg_read(polys[p], swz=False)
        self.c("inverse x3: pass 1' (pass-A twiddles conj, dft8, untwist constants)")
        self.substamp("inv3", 1)
        for p in range(3):
            self.cmul_all([(polys[p][r], tw[r]) for r in range(8)], spare, conj=True)
            polys[p], spare = self.dft8(polys[p], spare, -1, home=(home_lo[p], home_lo[p] + 32))
            polys[p], spare = self.home(polys[p], spare, home_lo[p], home_lo[p] + 32)
        self.substamp("inv3", 2)
        self.twist_consts(polys, spare, conj=True)
        self.substamp("inv3", 3)
        return polys, spare

    # ---------------------------------------------------------------- integer <-> double
    def key_bias(self): return 4096 if self.e_reg is not None else 0

    def key_quad(self, dst, r, ptr):
        """load row r (0..7: 64 complex = 1 KiB) of the key polynomial at ptr (biased by key_bias()) into the quad dst"""
        if self.e_reg is not None: self.gload("global_load_dwordx4", dst, L16, ptr, offset=(r - 4) * 1024)
        else: self.gload("global_load_dwordx4", dst, L16 if r < 4 else OFF1, ptr, offset=(r & 3) * 1024)

    def digit_pair(self, q, own_lo, own_hi, sgn, t, cy=VCC):
        """q holds a rotated torus word in (q.i, q.i+1): replace it by the f64 digit of (+-word - own).
        pbs_digit of tx_kernels.hpp (B = 23, one level, balanced top tie).  t: a temporary -- an aligned register PAIR selects the
        short form below (negated accumulator only), a single register the five-instruction difference of rounds 2-3."""
        lo, hi = V(q.i), V(q.i + 1)
        short = self.neg_acc and t.n == 2
        if short:
            # The words are those of -acc and sgn is the inverted mask: d = acc'[j] +- acc'[j + rot], + where sgn = 0, - where
            # sgn = -1.  x - y = ~(~x + y), so d = ((own ^ sgn) + word) ^ sgn: two xors into the temporary pair, ONE 64-bit add --
            # and only the high dword of d is ever read (B = 23, one level), so the closing xor is that dword's alone and rides in
            # the rounding add (v_xad_u32).  Seven instructions per word where the conditional negation (xor, xor, v_sub_co_u32,
            # v_subb_co_u32: the carry forms cost 4 cycles each) + add + rounding add took nine: 8 cycles x 48 words per step.
            assert own_hi.i == own_lo.i + 1 and own_lo.i % 2 == 0 and t.i % 2 == 0
            self.e("v_xor_b32", t.lo(), own_lo, sgn)
            self.e("v_xor_b32", t.hi(), own_hi, sgn)
            self.e("v_lshl_add_u64", V(q.i, 2), t, Lit(0), V(q.i, 2))
            t = t.lo()
        else:
            if t.n == 2: t = t.lo()
            self.e("v_xor_b32", lo, lo, sgn)
            self.e("v_xor_b32", hi, hi, sgn)
            self.e("v_sub_co_u32", lo, cy, lo, sgn)
            self.e("v_subb_co_u32", hi, cy, hi, sgn, cy)
            if self.neg_acc:         # the words are those of -acc and sgn is the inverted mask: -+acc'[j + rot] + acc'[j]
                assert own_hi.i == own_lo.i + 1 and own_lo.i % 2 == 0
                self.e("v_lshl_add_u64", V(q.i, 2), V(q.i, 2), Lit(0), V(own_lo.i, 2))
            else:
                self.e("v_sub_co_u32", lo, cy, lo, own_lo)
                self.e("v_subb_co_u32", hi, cy, hi, own_hi, cy)
        # digit = sext23(((hi + 0x100) >> 9) - e) + e with e = 1 on even coefficient indices (= even lanes), 0 on odd ones: the tie
        # 2^22 goes to +B/2 there and to -B/2 here.  Three instructions (round 4; five before): EREG holds 0x100 - (e << 9), so one
        # add rounds and shifts the tie, an arithmetic shift right by 9 of the 32-bit word IS the sign-extended 23-bit field, and
        # "+ e" is the carry-in of a v_addc_co_u32 whose carry mask is vcc = the even lanes (set once in the prologue)
        if short: self.e("v_xad_u32", t, hi, sgn, EREG)
        else: self.e("v_add_u32", t, hi, EREG)
        self.e("v_ashrrev_i32", t, Lit(9), t)
        if self.e_reg is not None: self.e("v_add_u32", t, t, self.e_reg)
        else: self.e("v_addc_co_u32", t, cy, t, Lit(0), VCC)
        self.e("v_cvt_f64_i32", V(q.i, 2), t)

    def torus_add(self, x, cur_lo, cur_hi, dst_lo, dst_hi, tA, tB, cy=VCC, neg=False):
        """dst = cur + round-half-even(x) mod 2^64 (torus_add_f64 of tx_device.hpp); x: f64 pair (destroyed); tA, tB: temporary
        pairs.  neg: dst = cur - round(x) instead -- rounding to nearest even is symmetric, so that is cur + round(-x), and the
        negation rides on the source modifiers of the first two operations.  The rounded value is assembled in x itself (low
        dword: the 2^52 magic's mantissa; high dword: bits(t1).lo + bits(t2).hi - 0x43380000) and added by ONE v_lshl_add_u64
        where the operands are aligned pairs (round 4: v_add_co_u32 + v_addc_co_u32 cost 4 cycles each, the 64-bit add 4 in all,
        tools/microbench/insn_rate.hip)."""
        xs = -x if neg else x
        self.e("v_mul_f64", tA, xs, SG.C_2M64)
        self.e("v_rndne_f64", tA, tA)
        self.e("v_fma_f64", x, tA, SG.C_M2P64, xs)             # ys
        self.e("v_add_f64", tA, x, SG.C_84)                    # t1
        self.e("v_add_f64", tB, tA, -SG.C_84)                  # h
        self.e("v_add_f64", x, x, -tB)                         # lo
        self.e("v_add_f64", x, x, SG.C_52)                     # t2
        pair = lambda lo, hi: isinstance(lo, V) and isinstance(hi, V) and hi.i == lo.i + 1 and lo.i % 2 == 0
        if pair(cur_lo, cur_hi) and pair(dst_lo, dst_hi):
            self.e("v_add3_u32", x.hi(), tA.lo(), x.hi(), SG.C_M4338)      # rhi = bits(t1).lo + bits(t2).hi - 0x43380000, next to the low dword
            self.e("v_lshl_add_u64", V(dst_lo.i, 2), x, Lit(0), V(cur_lo.i, 2))
        else:
            self.e("v_add3_u32", tA.lo(), tA.lo(), x.hi(), SG.C_M4338)
            self.e("v_add_co_u32", dst_lo, cy, cur_lo, x.lo())
            self.e("v_addc_co_u32", dst_hi, cy, cur_hi, tA.lo(), cy)

    # ---------------------------------------------------------------- the kernel
    def prologue(self):
        e = self.e
        self.c("---- prologue: lane-dependent addresses, constants, accumulator mask = 0")
        self.p.emit("s_waitcnt", vm=0, lgkm=0)
        e("v_mbcnt_lo_u32_b32", LANE, Lit(-1), Lit(0))
        e("v_mbcnt_hi_u32_b32", LANE, Lit(-1), LANE)
        e("v_lshl_add_u32", L8T, LANE, Lit(3), SG.TILE)
        e("v_lshl_add_u32", L16T, LANE, Lit(4), SG.TILE)
        e("v_lshlrev_b32", L16, Lit(4), LANE)
        if self.e_reg is None: e("v_add_u32", OFF1, Lit(4096), L16)
        e("v_and_b32", T0.lo(), Lit(7), LANE)                   # b
        e("v_lshlrev_b32", TWBA, Lit(4), T0.lo())
        e("v_add_u32", TWBA, Lit(TWB), TWBA)
        e("v_lshrrev_b32", T0.hi(), Lit(3), LANE)               # a
        # tile rows have a pitch of 65 complex (tx_device.hpp): exchange hi stores element r at row a, column 8 r + b;
        # exchange lo at row b, column g8 + r; both read row r, column lane -- every b128 access conflict-free, no swizzle
        assert PITCH == 65 * 16
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
        if self.fair:
            e("v_mov_b32", PARTNER, Lit(0))
            e("s_mov_b32", SG.FAIR, Lit(0))
        else: e("v_lshlrev_b32", L8, Lit(3), LANE)
        for name, val in (("C_2M64", 2.0 ** -64), ("C_M2P64", -(2.0 ** 64)), ("C_84", 1.5 * 2.0 ** 84), ("C_52", 1.5 * 2.0 ** 52),
                          ("C_SQH", 0.70710678118654752440)):
            s, b = getattr(SG, name), f64bits(val)
            e("s_mov_b32", s.lo(), Lit(b & 0xFFFFFFFF)); e("s_mov_b32", s.hi(), Lit(b >> 32))
        e("s_mov_b32", SG.C_M4338, Lit((-0x43380000) & 0xFFFFFFFF))
        for q in range(2):
            for m in range(32): e("v_mov_b32", V(A[q] + m), Lit(0))
        if self.neg_acc:         # the loop works on -acc: the mask polynomials are 0 either way, the body is negated in place
            W = [V(X + 2 * j, 2) for j in range(16)]
            for j in range(16): self.ds_read("ds_read_b64", W[j], L8T, offset=BODY + 512 * j)
            for j in range(16):
                self.negate64(W[j], CARRY[j % 4])
                self.ds_write("ds_write_b64", L8T, W[j], offset=BODY + 512 * j)
            self.lg_wait_all()
        e("v_mov_b32", R10, Lit(0))
        self.gload("global_load_dwordx2", V(10, 2), R10, SG.CT, offset=0)     # mask element 0 (voffset register = its own dst: read at issue)
        e("s_mov_b32", SG.I, Lit(0))
        if self.stamps:
            for k in range(NSTAMP):
                e("s_mov_b32", ST_ACC[k].lo(), Lit(0)); e("s_mov_b32", ST_ACC[k].hi(), Lit(0))
            self.clock_mark(8)
            self.stamp(None)

    def step(self):
        e = self.e
        self.p.label("L_step%=")
        # ---- rotation amount of this step (wave-uniform), prefetch of the next mask element
        if self.fair: self.fair_exchange()
        self.setprio("R")
        self.c("---- modswitch(ct[i]) -> rot; base = (lane - rot) mod 2N")
        # loop head: the only vector-memory operation that can be in flight is the load of this step's mask element
        # (from the prologue on entry; on the back edge it retired behind the key loads).  A full wait makes both
        # entries the same state.
        self.p.emit("s_waitcnt", vm=0); self.vmq.clear()
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
            e("v_bfe_i32", SGN[m], t, Lit(4), Lit(1))           # 0 / -1: coefficient comes from the negated half
            e("v_and_b32", t, Lit(15), t)
            e("v_lshl_add_u32", RA[m], t, Lit(9), R11)
        # next mask element (index min(i+1, n-1)), a step ahead of its use
        e("s_add_u32", SG.TMP, SG.I, Lit(1))
        e("s_cmp_lt_u32", SG.TMP, SG.NSTEP)
        e("s_cbranch_scc1", "L_nx%=")
        e("s_mov_b32", SG.TMP, SG.I)
        self.p.label("L_nx%=")
        e("s_lshl_b32", SG.TMP, SG.TMP, Lit(3))
        e("v_mov_b32", R10, SG.TMP)
        self.gload("global_load_dwordx2", V(10, 2), R10, SG.CT, offset=0)
        NG = 12
        # Key ring: 12 register quads (X).  Eleven of a column's first twelve elements are issued ahead of the MAC, into
        # the quads that are free at the time: column 0 -- X9..11 at the start of the step (the rotation uses X0..7 only),
        # X0..7 before the last forward pass (twiddles done), X8 (the spare) at the MAC; columns 1, 2 -- X1..11 right
        # after the previous inverse (the conversions then run two chains on X0, T0, T1), X0 at the MAC.
        EARLY = 11 if self.deep_prefetch else 8
        stream = [(q, r) for q in range(3) for r in range(8)]
        kps = (SG.KP0, SG.KP1, SG.KP2)

        def ring_of(col):      # slot order = consumption order = issue order
            if self.deep_prefetch: order = [9, 10, 11, 0, 1, 2, 3, 4, 5, 6, 7, 8] if col == 0 else [4, 5, 6, 7, 8, 9, 10, 11, 1, 2, 3, 0]
            else: order = list(range(12)) if col == 0 else list(range(4, 12)) + list(range(4))
            return [quad(X, r) for r in order]

        def key_ptrs(col):
            for q in range(3):
                e("s_add_u32", kps[q].lo(), SG.ROW.lo(), Lit((q * 3 + col) * 8192 + self.key_bias()))
                e("s_addc_u32", kps[q].hi(), SG.ROW.hi(), Lit(0))

        def issue(col, k):
            q, r = stream[k]
            self.key_quad(ring_of(col)[k % NG], r, kps[q])

        def early(col, lo=0, hi=EARLY):
            self.c("key ring: elements %d..%d of column %d, issued ahead of their use" % (lo, hi - 1, col))
            if lo == 0: key_ptrs(col)
            for k in range(lo, hi): issue(col, k)
        if self.deep_prefetch: early(0, 0, 3)                  # X9..11: free for the whole rotation / forward phase
        # ---- stage the mask polynomials through the tile, gather the rotated words (in-order LDS: no waits in between)
        self.c("---- rotation: stage mask polynomials in the tile, gather rotated coefficients")
        xq = [quad(X, r) for r in range(12)]
        OWN = [V(X + 2 * j, 2) for j in range(16)]              # body words of this lane
        temps = [T0, T1, V(X + 32, 2), V(X + 34, 2)] if os.environ.get("BR_DIGIT", "short") == "short" else [T0.lo(), T0.hi(), T1.lo(), T1.hi()]   # X8, the spare quad, is idle here

        def lds_traffic(q):
            if q < 2:
                for m in range(16): self.ds_write("ds_write_b64", L8T, V(A[q] + 2 * m, 2), offset=512 * m)
            else:
                for j in range(16): self.ds_read("ds_read_b64", OWN[j], L8T, offset=BODY + 512 * j)
            for m in range(8):
                self.ds_read("ds_read_b64", V(D[q] + 4 * m, 2), RA[m], offset=BODY if q == 2 else 0)
                self.ds_read("ds_read_b64", V(D[q] + 4 * m + 2, 2), RA[m + 8], offset=BODY if q == 2 else 0)

        def digits(q):
            chains = []
            for m in range(8):
                for half in range(2):
                    j = m + 8 * half
                    slot = V(D[q] + 4 * m + 2 * half, 2)
                    if q < 2: ol, oh = V(A[q] + 2 * j), V(A[q] + 2 * j + 1)
                    else: ol, oh = OWN[j].lo(), OWN[j].hi()
                    chains.append(self.chain(self.digit_pair, slot, ol, oh, SGN[j], temps[len(chains) % 4], CARRY[len(chains) % 4]))
                    if len(chains) == 4:
                        self.merge(chains); chains = []
        lds_traffic(0)
        lds_traffic(1)                  # in-order LDS: polynomial 1's stores queue behind polynomial 0's gathers
        self.c("---- digits: ct1 = acc X^rot - acc, one level (B = 23), as f64")
        digits(0)
        lds_traffic(2)
        digits(1)
        digits(2)
        self.stamp(0)
        # ---- forward transforms
        polys = [[quad(D[p], r) for r in range(8)] for p in range(3)]
        self.setprio("F")
        polys, spare = self.fwd3(polys, xq, before_pass3=lambda: early(0, 3 if self.deep_prefetch else 0, EARLY))
        self.stamp(1)
        # ---- per column: multiply-accumulate against the key, inverse transform, update
        for col in range(3):
            self.c("---- column %d: F = sum_q D_q * key[q][col]" % col)
            self.setprio("M")
            ring = ring_of(col)
            for k in range(EARLY, NG): issue(col, k)
            Fq = [quad(F, r) for r in range(8)]

            def mac(k):
                if "M" in ABL: return
                q, r = stream[k]
                g, d = ring[k % NG], polys[q][r]
                re0 = Lit(0, f64=0.0) if q == 0 else qre(Fq[r])
                im0 = Lit(0, f64=0.0) if q == 0 else qim(Fq[r])
                self.e("v_fma_f64", qre(Fq[r]), qre(d), qre(g), re0)
                self.e("v_fma_f64", qim(Fq[r]), qre(d), qim(g), im0)
                self.e("v_fma_f64", qre(Fq[r]), -qim(d), qim(g), qre(Fq[r]))
                self.e("v_fma_f64", qim(Fq[r]), qim(d), qre(g), qim(Fq[r]))
            for k in range(0, len(stream), 2):
                self.merge([self.chain(mac, k), self.chain(mac, k + 1)])
                for kk in (k, k + 1):
                    if kk + NG < len(stream): issue(col, kk + NG)
            self.stamp(2)
            self.setprio("I")
            v, sp2 = self.inv1(Fq, [quad(X, r) for r in range(12)])
            self.stamp(3)
            self.setprio("U")
            if col < 2: early(col + 1)              # lands during the conversions below (they use X[0..3] only)
            self.c("---- column %d: accumulator += round(inverse)" % col)
            # conversions: columns 0, 1 run two chains (temporaries X0, T0, T1: X1..11 already hold the next column's keys),
            # column 2 four (X0..3; nothing is prefetched behind it)
            nch = 4 if (col == 2 or not self.deep_prefetch) else 2
            tmp = [(V(X + 4 * k, 2), V(X + 4 * k + 2, 2)) for k in range(4)] if nch == 4 else [(V(X, 2), V(X + 2, 2)), (T0, T1)]
            CUR = [V(X + 16 + 2 * j, 2) for j in range(16)]                      # body words (column 2): X[4..11]
            if col == 2:
                for j in range(16): self.ds_read("ds_read_b64", CUR[j], L8T, offset=BODY + 512 * j)

            def upd(m, half, k):
                j = m + 8 * half
                x = V(v[m].i + 2 * half, 2)
                if col < 2:
                    al, ah = V(A[col] + 2 * j), V(A[col] + 2 * j + 1)
                    self.torus_add(x, al, ah, al, ah, tmp[k][0], tmp[k][1], CARRY[k], neg=self.neg_acc)
                else:
                    self.torus_add(x, CUR[j].lo(), CUR[j].hi(), CUR[j].lo(), CUR[j].hi(), tmp[k][0], tmp[k][1], CARRY[k], neg=self.neg_acc)
                    self.ds_write("ds_write_b64", L8T, CUR[j], offset=BODY + 512 * j)
            chains = []
            for m in range(8):
                for half in range(2):
                    chains.append(self.chain(upd, m, half, len(chains)))
                    if len(chains) == nch:
                        self.merge(chains); chains = []
            self.stamp(4)
        # ---- loop control
        self.lg_wait_all()
        e("s_add_u32", SG.ROW.lo(), SG.ROW.lo(), Lit(KP * KP * H * 16))
        e("s_addc_u32", SG.ROW.hi(), SG.ROW.hi(), Lit(0))
        e("s_add_u32", SG.I, SG.I, Lit(1))
        e("s_cmp_lt_u32", SG.I, SG.NSTEP)
        if "K" in ABL: self.vm_wait_all()
        assert self.vmq == [] and self.lgq == [], (self.vmq, self.lgq)      # same state as at the loop head
        e("s_cbranch_scc1", "L_step%=")

    def epilogue(self):
        e = self.e
        self.c("---- epilogue: store the accumulator")
        self.vm_wait_all()
        if self.fair:
            self.lg_wait_all()
            e("s_setprio", Lit(0))
            e("s_add_u32", SG.TMP2, SG.PBASE, SG.NSTEP)          # this unit is done (the wrapper publishes "finished" after its last one)
            e("v_mov_b32", T0.lo(), SG.TMP2)
            e("v_mov_b32", T0.hi(), SG.PROGA)
            e("ds_write_b32", T0.hi(), T0.lo(), offset=0)
            e("v_lshlrev_b32", L8, Lit(3), LANE)
        e("v_add_u32", T0.lo(), Lit(4096), L8)
        for q in range(3):
            e("s_add_u32", SG.KP0.lo(), SG.OUT.lo(), Lit(q * 8192))
            e("s_addc_u32", SG.KP0.hi(), SG.OUT.hi(), Lit(0))
            for m in range(16):
                if q < 2: src = V(A[q] + 2 * m, 2)
                else:
                    src = V(X + 2 * m, 2)
                    self.ds_read("ds_read_b64", src, L8T, offset=BODY + 512 * m)
                    self.lg_wait(self.regs(src))
                if self.neg_acc: self.negate64(src, CARRY[m % 4])           # back to acc
                self.gstore("global_store_dwordx2", L8 if m < 8 else T0.lo(), src, SG.KP0, offset=512 * (m & 7))
        self.vm_wait_all()
        if self.stamps:
            e("v_mov_b32", T0.hi(), Lit(0))
            for k in range(NSTAMP):
                e("v_mov_b32", T1.lo(), ST_ACC[k].lo()); e("v_mov_b32", T1.hi(), ST_ACC[k].hi())
                self.gstore("global_store_dwordx2", T0.hi(), T1, ST_DBG, offset=8 * k)
            self.clock_mark(10)
            self.vm_wait_all()

    def build(self):
        self.prologue()
        self.step()
        self.epilogue()
        return self.p


INPUTS = ["ct", "row", "out", "n", "tile"]     # asm operands %0..%4 ("s" constraints)


def render_inc(prog, stamps=False, fair=False):
    """C string literal pieces of ONE asm block.  Operand copies first: %0 ct (64), %1 bsk row 0 (64), %2 out (64), %3 n, %4 tile base,
    fair: %5 / %6 LDS addresses of this wave's / its SIMD partner's step counter, %7 steps done before this unit; then the tie mode
    (1 = balanced top-level tie, 0 = tfhe-rs' rule); stamps: then the debug buffer (64)"""
    head = ["s_mov_b64 %s, %%0" % SG.CT, "s_mov_b64 %s, %%1" % SG.ROW, "s_mov_b64 %s, %%2" % SG.OUT,
            "s_mov_b32 %s, %%3" % SG.NSTEP, "s_mov_b32 %s, %%4" % SG.TILE]
    nop = 5
    if fair:
        head += ["s_mov_b32 %s, %%5" % SG.PROGA, "s_mov_b32 %s, %%6" % SG.PROGB, "s_mov_b32 %s, %%7" % SG.PBASE]; nop = 8
    head.append("s_mov_b32 %s, %%%d" % (SG.TIEMODE, nop)); nop += 1
    if stamps: head.append("s_mov_b64 %s, %%%d" % (ST_DBG, nop))
    lines = ["  " + h for h in head] + prog.render()
    out = ["// generated by tools/asmgen/brgen.py -- do not edit"]
    for l in lines:
        out.append('"%s\\n"' % l.replace('"', '\\"'))
    return "\n".join(out) + "\n"


def render_clobbers():
    regs = ["v%d" % i for i in range(256)] + ["s%d" % i for i in range(34, 96)] + ["vcc", "scc", "memory"]
    return "// generated by tools/asmgen/brgen.py -- do not edit\n" + ", ".join('"%s"' % r for r in regs) + "\n"


if __name__ == "__main__":
    import os as _os
    g = Gen(prio=_os.environ.get("BR_PRIO", "I"), deep_prefetch=_os.environ.get("BR_EARLY", "11") != "8")
    prog = g.build()
    if len(sys.argv) > 2 and sys.argv[1] == "--out":
        import os
        open(os.path.join(sys.argv[2], "tx_br_asm.inc"), "w").write(render_inc(prog))
        open(os.path.join(sys.argv[2], "tx_br_asm_fair.inc"), "w").write(render_inc(Gen(prio=_os.environ.get("BR_PRIO", "I"), fair=True).build(), fair=True))
        open(os.path.join(sys.argv[2], "tx_br_asm_fair_stamps.inc"), "w").write(render_inc(
            Gen(stamps=True, fair=True, phase_stamps=not _os.environ.get("BR_CLOCK_ONLY")).build(), stamps=True, fair=True))
        open(os.path.join(sys.argv[2], "tx_br_asm_stamps.inc"), "w").write(render_inc(Gen(stamps=True, deep_prefetch=_os.environ.get("BR_EARLY", "11") != "8",
                                                                                           phase_stamps=not _os.environ.get("BR_CLOCK_ONLY")).build(), stamps=True))
        open(os.path.join(sys.argv[2], "tx_br_asm_clobbers.inc"), "w").write(render_clobbers())
    elif len(sys.argv) > 1 and sys.argv[1] == "--count":
        c = prog.count()
        for k in sorted(c, key=lambda k: -c[k]): print("%6d %s" % (c[k], k))
        print("%6d total" % sum(c.values()))
    else:
        sys.stdout.write(render_inc(prog))
