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
from gcn import V, S, Lit, VCC, Prog
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


class CoopGen(Gen):
    def __init__(self, stamps=False, prio=""):
        super().__init__(stamps=stamps, prio=prio, fair=False)

    def barrier(self, phase):
        """every LDS operation of this wave has completed, then the workgroup's three waves meet (diagnostic build: the time
        up to the barrier goes to `phase`, the wait at the barrier to accumulator 5)"""
        self.lg_wait_all()
        self.stamp(phase)
        if "B" not in ABL3: self.p.emit("s_barrier")
        self.stamp(5)

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
        self.xchg_hi_write(v)
        free = sorted(v + [spare], key=lambda q: q.i)
        v, spare = free[:8], free[8]
        self.xchg_read(v, swz=False)
        self.c("forward: pass 2 (dft8, pass-B twiddles, exchange lo)")
        self.load_tw(tw, 0, 7, compact=True)
        v, spare = self.dft8(v, spare, +1)
        self.cmul_all([(v[r], tw[r - 1]) for r in range(1, 8)])
        self.xchg_lo_write(v)
        free = sorted(v + [spare], key=lambda q: q.i)
        v, spare = free[:8], free[8]
        self.xchg_read(v, swz=True)
        self.c("forward: pass 3 (dft8)")
        if before_pass3: before_pass3()            # the twiddle quads xq[0..7] are free from here on
        v, spare = self.dft8(v, spare, +1)
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
        self.xchg_lo_write(v)
        free = sorted(v + [spare], key=lambda q: q.i)
        v, spare = free[:8], free[8]
        self.xchg_read(v, swz=True)
        self.c("inverse: pass 2' (pass-B twiddles conj, dft8, exchange hi)")
        self.cmul_all([(v[r], tw[r - 1]) for r in range(1, 8)], spare, conj=True)
        self.load_tw(tw, 0, 8)
        v, spare = self.dft8(v, spare, -1)
        self.xchg_hi_write(v)
        free = sorted(v + [spare], key=lambda q: q.i)
        v, spare = free[:8], free[8]
        self.xchg_read(v, swz=False)
        self.c("inverse: pass 1' (pass-A twiddles conj, dft8, untwist constants)")
        self.cmul_all([(v[r], tw[r]) for r in range(8)], spare, conj=True)
        home_lo = min(q.i for q in v)
        v, spare = self.dft8(v, spare, -1)
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
        e("v_add_u32", OFF1, Lit(4096), L16)
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
        e("v_and_b32", T0.lo(), Lit(1), LANE)
        e("v_sub_u32", EREG, Lit(1), T0.lo())
        for name, val in (("C_2M64", 2.0 ** -64), ("C_M2P64", -(2.0 ** 64)), ("C_84", 1.5 * 2.0 ** 84), ("C_52", 1.5 * 2.0 ** 52),
                          ("C_SQH", 0.70710678118654752440)):
            s, b = getattr(SG, name), f64bits(val)
            e("s_mov_b32", s.lo(), Lit(b & 0xFFFFFFFF)); e("s_mov_b32", s.hi(), Lit(b >> 32))
        e("s_mov_b32", SG.C_M4338, Lit((-0x43380000) & 0xFFFFFFFF))
        # the wrapper left this wave's polynomial (zero for the mask polynomials, the rotated test vector for the body)
        # in its tile, natural order
        for m in range(16): self.ds_read("ds_read_b64", V(A3 + 2 * m, 2), L8T, offset=512 * m)
        e("v_mov_b32", R10, Lit(0))
        self.gload("global_load_dwordx2", V(10, 2), R10, SG.CT, offset=0)     # mask element 0
        e("s_mov_b32", SG.I, Lit(0))
        self.lg_wait_all()
        if self.stamps:
            for k in range(NSTAMP):
                e("s_mov_b32", ST_ACC[k].lo(), Lit(0)); e("s_mov_b32", ST_ACC[k].hi(), Lit(0))
            self.stamp(None)

    def step(self):
        e = self.e
        self.p.label("L_step%=")
        self.setprio("R")
        self.c("---- modswitch(ct[i]) -> rot; base = (lane - rot) mod 2N")
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
            e("v_add_u32", t, Lit(m), R10)
            e("v_bfe_i32", SGN3[m], t, Lit(4), Lit(1))          # 0 / -1: coefficient comes from the negated half
            e("v_and_b32", t, Lit(15), t)
            e("v_lshl_add_u32", RA3[m], t, Lit(9), R11)
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
            self.gload("global_load_dwordx4", ring[k % NG], L16 if r < 4 else OFF1, kps[q], offset=(r & 3) * 1024)

        def early(lo, hi):
            self.c("key ring: elements %d..%d, issued ahead of their use" % (lo, hi - 1))
            if lo == 0:
                for q in range(3):       # ROW points at column c of row i: key[q][c] is q * 3 polynomials further
                    e("s_add_u32", kps[q].lo(), SG.ROW.lo(), Lit(q * 3 * 8192))
                    e("s_addc_u32", kps[q].hi(), SG.ROW.hi(), Lit(0))
            for k in range(lo, hi): issue(k)
        early(0, 3)
        # ---- rotation: stage the polynomial in the tile (natural order), gather the rotated words
        self.c("---- rotation: stage this wave's polynomial in its tile, gather rotated coefficients")
        for m in range(16): self.ds_write("ds_write_b64", L8T, V(A3 + 2 * m, 2), offset=512 * m)
        for m in range(8):
            self.ds_read("ds_read_b64", V(D3 + 4 * m, 2), RA3[m])
            self.ds_read("ds_read_b64", V(D3 + 4 * m + 2, 2), RA3[m + 8])
        self.c("---- digits: ct1 = acc X^rot - acc, one level (B = 23), as f64")
        temps = [T0.lo(), T0.hi(), T1.lo(), T1.hi()]
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
        self.stamp(2)
        # ---- inverse; the tiles are free for its exchanges once every wave has read what it needs
        self.setprio("I")
        v, sp2 = self.inv1_hooked(Fq, xq, lambda: self.barrier(3))
        self.stamp(3)
        self.setprio("U")
        self.c("---- this wave's polynomial += round(inverse)")
        tmp = [(V(X3 + 4 * k, 2), V(X3 + 4 * k + 2, 2)) for k in range(4)]

        def upd(m, half, k):
            j = m + 8 * half
            x = V(v[m].i + 2 * half, 2)
            al, ah = V(A3 + 2 * j), V(A3 + 2 * j + 1)
            self.torus_add(x, al, ah, al, ah, tmp[k][0], tmp[k][1], CARRY[k])
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
        assert self.vmq == [] and self.lgq == [], (self.vmq, self.lgq)      # same state as at the loop head
        e("s_cbranch_scc1", "L_step%=")

    def epilogue(self):
        e = self.e
        self.c("---- epilogue: store this wave's polynomial (OUT points at it)")
        self.vm_wait_all()
        e("v_lshlrev_b32", L8, Lit(3), LANE)                    # (register 9: the published-transform address is dead)
        e("v_add_u32", T0.lo(), Lit(4096), L8)
        for m in range(16):
            self.gstore("global_store_dwordx2", L8 if m < 8 else T0.lo(), V(A3 + 2 * m, 2), SG.OUT, offset=512 * (m & 7))
        self.vm_wait_all()
        if self.stamps:
            e("v_mov_b32", T0.hi(), Lit(0))
            for k in range(NSTAMP):
                e("v_mov_b32", T1.lo(), ST_ACC[k].lo()); e("v_mov_b32", T1.hi(), ST_ACC[k].hi())
                self.gstore("global_store_dwordx2", T0.hi(), T1, ST_DBG, offset=8 * k)
            self.vm_wait_all()


def render_inc(prog, stamps=False):
    """operands: %0 ct (64), %1 column c of bsk row 0 (64), %2 out + c N (64), %3 n, %4 this wave's tile, %5 wave 0's tile;
    stamps: %6 the debug buffer (64)"""
    head = ["s_mov_b64 %s, %%0" % SG.CT, "s_mov_b64 %s, %%1" % SG.ROW, "s_mov_b64 %s, %%2" % SG.OUT,
            "s_mov_b32 %s, %%3" % SG.NSTEP, "s_mov_b32 %s, %%4" % SG.TILE, "s_mov_b32 %s, %%5" % TILE0]
    if stamps: head.append("s_mov_b64 %s, %%6" % ST_DBG)
    lines = ["  " + h for h in head] + prog.render()
    out = ["// generated by tools/asmgen/brgen3.py -- do not edit"]
    for l in lines:
        out.append('"%s\\n"' % l.replace('"', '\\"'))
    return "\n".join(out) + "\n"


def render_clobbers():
    regs = ["v%d" % i for i in range(NV)] + ["s%d" % i for i in range(34, 96)] + ["vcc", "scc", "memory"]
    return "// generated by tools/asmgen/brgen3.py -- do not edit\n" + ", ".join('"%s"' % r for r in regs) + "\n"


if __name__ == "__main__":
    prog = CoopGen(prio=os.environ.get("BR3_PRIO", "")).build()
    if len(sys.argv) > 2 and sys.argv[1] == "--out":
        open(os.path.join(sys.argv[2], "tx_br3_asm.inc"), "w").write(render_inc(prog))
        open(os.path.join(sys.argv[2], "tx_br3_asm_stamps.inc"), "w").write(render_inc(CoopGen(stamps=True).build(), stamps=True))
        open(os.path.join(sys.argv[2], "tx_br3_asm_clobbers.inc"), "w").write(render_clobbers())
    elif len(sys.argv) > 1 and sys.argv[1] == "--count":
        c = prog.count()
        for k in sorted(c, key=lambda k: -c[k]): print("%6d %s" % (c[k], k))
        print("%6d total" % sum(c.values()))
    else:
        sys.stdout.write(render_inc(prog))
