"""Generator of the PAIRED blind-rotation loop: TWO waves work on ONE unit, two to one (an experiment of round 3 that is not in the library -- its parked wrapper is tools/asmgen/k_blind_rotate7.hip.txt, not the k_blind_rotate7 of tx_br2.hpp --, launches of three to
four units per CU: the 8-blocks-per-GPU share of the medium instance on an 8-GPU node).

The cooperative loop of brgen3.py (three waves per unit, one per polynomial) shortens a CMux chain by 2.4x when the chip is
nearly empty, but at four units per CU its twelve waves per CU -- three per SIMD at 168 registers, 25 % more LDS operations
than the one-wave kernel, two three-wave barriers per step across SIMDs -- only tie with four one-wave units.  At that
occupancy the chip has room for exactly TWO waves per SIMD at 256 registers, so here a unit is split two to one:
    wave A owns the two MASK polynomials of the accumulator and columns 0 and 1: per step two rotations + decompositions,
           two forward transforms (interleaved), publishes both, multiply-accumulates two columns (its own two transforms
           from registers, wave B's from LDS), two inverses (interleaved), two updates;
    wave B owns the BODY polynomial and column 2: one of each, reads both of A's transforms from LDS.
A is the critical path (two thirds of the one-wave loop's work); B does its half of that in A's shadow.  Workgroup = the
two waves of one unit; the two waves that share a SIMD then come from different workgroups and drift in phase (pairing
every A with its own B on one SIMD, in workgroups of four units, measured 5 % slower: the barriers lock them in phase and
they stall together).  Two s_barrier per step (transforms published / tiles free again), as brgen3.py.

Arithmetic, operation order and results are those of k_blind_rotate / brgen.py / the oracle's txo_blind_rotate(): every
column accumulates q = 0, 1, 2 in that order.  Bit-identical; tests/asmgen_emu_blind_rotate2.py runs the two programs as two
emulated waves (shared LDS, barriers honoured, both scheduling orders) against the oracle's primitives.

    python3 tools/asmgen/brgen2.py --out temp-fhe-transciphering_amd/csrc      (tx_br2a_asm.inc, tx_br2b_asm.inc + clobbers)

LDS image of a workgroup: compact twiddles at 0 (9 088 B), tile of wave A (8 320 B, pitch 65 complex), tile of wave B, an
8 KiB buffer for A's second transform: 33 920 B, four workgroups per CU.

Register maps (VGPR).  A: 0..15 as brgen.py, A0 16..47, A1 48..79 (mask polynomials), D0 80..111, D1 112..143 (digits ->
transforms; D0 lands B's transform in the last third of the multiply-accumulate), F0 144..175, F1 176..207 (rotation tables,
then the two Fourier accumulators), X 208..255 (key ring / twiddles + spare).  B: 9 tile A + lane * 16, A2 16..47 (body), D2
48..79, F2 80..111, X 112..159, L 160..191 (landing ring of A's transforms).
"""
import os
import sys
from gcn import V, S, Lit, VCC, Prog
import brgen
import brgen3
from brgen import (Gen, SG, quad, qre, qim, f64bits, T0, T1, LANE, L8T, L16T, L16, OFF1, TWBA, XHW, XLW, EREG, L8, R10, R11,
                   CARRY, PITCH, N, H, KP)

TILE_BYTES = 8 * PITCH                   # 8 320
PBUF = 2 * TILE_BYTES                    # A's second transform, from tile A
UNIT_LDS = 2 * TILE_BYTES + 8192         # 24 832 bytes per unit
L16A = V(9)                              # (wave B) tile A + lane * 16
KP6 = [S(60, 2), S(62, 2), S(64, 2), S(66, 2), S(70, 2), S(72, 2)]     # key pointers (q, column)
ABL2 = os.environ.get("BR2_ABL", "")     # timing-only ablations (B no barriers, P no publish / reads of transforms): outputs invalid


class PairGen(brgen3.CoopGen):
    def __init__(self, role):
        super().__init__(stamps=False, prio="")
        self.e_reg = None                     # (this experiment keeps lane * 16 + 4096 in v4 and the carry-in form of the digit)
        self.neg_acc = False          # (this experiment keeps the accumulator as it was built: its own prologues / epilogues)
        assert role in "AB"
        self.role = role
        if role == "A":
            self.AP, self.DP, self.FP, self.X = [16, 48], [80, 112], [144, 176], 208
        else:
            self.AP, self.DP, self.FP, self.X, self.LAND = [16], [48], [80], 112, 160
        self.NP = len(self.AP)
        self.RA = [V(self.FP[0] + m) for m in range(16)]
        self.SGN = [V(self.FP[0] + 16 + m) for m in range(16)]

    def barrier(self, phase=None):
        self.lg_wait_all()
        if "B" not in ABL2: self.p.emit("s_barrier")

    # ---------------------------------------------------------------- P polynomials forward / inverse (fwd3 / inv3 of brgen.py for any P)
    def fwdP(self, polys, xq, before_pass3=None):
        P = len(polys)
        tw, spare = xq[:8], xq[8]
        home_lo = [min(q.i for q in polys[p]) for p in range(P)]
        self.load_tw(tw, 0, 8)
        self.twist_consts(polys, None, conj=False)
        for p in range(P):
            polys[p], spare = self.dft8(polys[p], spare, +1)
            self.cmul_all([(polys[p][r], tw[r]) for r in range(8)])
            self.xchg_hi_write(polys[p])
            free = sorted(polys[p] + [spare], key=lambda q: q.i)
            polys[p], spare = free[:8], free[8]
            self.xchg_read(polys[p], swz=False)
        self.load_tw(tw, 0, 7, compact=True)
        for p in range(P):
            polys[p], spare = self.dft8(polys[p], spare, +1)
            self.cmul_all([(polys[p][r], tw[r - 1]) for r in range(1, 8)])
            self.xchg_lo_write(polys[p])
            free = sorted(polys[p] + [spare], key=lambda q: q.i)
            polys[p], spare = free[:8], free[8]
            self.xchg_read(polys[p], swz=True)
        if before_pass3: before_pass3()
        for p in range(P):
            polys[p], spare = self.dft8(polys[p], spare, +1)
            polys[p], spare = self.home(polys[p], spare, home_lo[p], home_lo[p] + 32)
        return polys, spare

    def invP(self, polys, xq, hook):
        """hook() runs in front of the first exchange store (the tiles are being read by the partner until then)"""
        P = len(polys)
        tw, spare = xq[:8], xq[8]
        home_lo = [min(q.i for q in polys[p]) for p in range(P)]
        self.load_tw(tw, 0, 7, compact=True)
        for p in range(P):
            polys[p], spare = self.dft8(polys[p], spare, -1)
            if p == 0: hook()
            self.xchg_lo_write(polys[p])
            free = sorted(polys[p] + [spare], key=lambda q: q.i)
            polys[p], spare = free[:8], free[8]
            self.xchg_read(polys[p], swz=True)
        for p in range(P):
            self.cmul_all([(polys[p][r], tw[r - 1]) for r in range(1, 8)], spare, conj=True)
            if p == P - 1: self.load_tw(tw, 0, 8)            # the pass-A rows replace the pass-B rows after their last use
            polys[p], spare = self.dft8(polys[p], spare, -1)
            self.xchg_hi_write(polys[p])
            free = sorted(polys[p] + [spare], key=lambda q: q.i)
            polys[p], spare = free[:8], free[8]
            self.xchg_read(polys[p], swz=False)
        for p in range(P):
            self.cmul_all([(polys[p][r], tw[r]) for r in range(8)], spare, conj=True)
            polys[p], spare = self.dft8(polys[p], spare, -1)
            polys[p], spare = self.home(polys[p], spare, home_lo[p], home_lo[p] + 32)
        self.twist_consts(polys, spare, conj=True)
        return polys, spare

    # ---------------------------------------------------------------- the kernel
    def prologue(self):
        e = self.e
        self.c("---- prologue (wave %s): addresses, constants, this wave's polynomial(s)" % self.role)
        self.p.emit("s_waitcnt", vm=0, lgkm=0)
        prio = os.environ.get("BR2_PRIO_" + self.role)       # experiment (A above B measured no different)
        if prio: e("s_setprio", Lit(int(prio)))
        e("v_mbcnt_lo_u32_b32", LANE, Lit(-1), Lit(0))
        e("v_mbcnt_hi_u32_b32", LANE, Lit(-1), LANE)
        e("v_lshl_add_u32", L8T, LANE, Lit(3), SG.TILE)
        e("v_lshl_add_u32", L16T, LANE, Lit(4), SG.TILE)
        if self.role == "B":
            e("s_sub_u32", SG.TMP, SG.TILE, Lit(TILE_BYTES))    # tile A precedes tile B
            e("v_lshl_add_u32", L16A, LANE, Lit(4), SG.TMP)
        e("v_lshlrev_b32", L16, Lit(4), LANE)
        e("v_add_u32", OFF1, Lit(4096), L16)
        e("v_and_b32", T0.lo(), Lit(7), LANE)
        e("v_lshlrev_b32", TWBA, Lit(4), T0.lo())
        e("v_add_u32", TWBA, Lit(brgen.TWB), TWBA)
        e("v_lshrrev_b32", T0.hi(), Lit(3), LANE)
        e("v_lshl_add_u32", T1.lo(), T0.hi(), Lit(6), T0.hi())
        e("v_add_u32", T0.hi(), T1.lo(), T0.lo())
        e("v_lshl_add_u32", XHW, T0.hi(), Lit(4), SG.TILE)
        e("v_and_b32", T1.lo(), Lit(0x38), LANE)
        e("v_lshl_add_u32", T1.hi(), T0.lo(), Lit(6), T0.lo())
        e("v_add_u32", T1.lo(), T1.hi(), T1.lo())
        e("v_lshl_add_u32", XLW, T1.lo(), Lit(4), SG.TILE)
        e("v_and_b32", T0.lo(), Lit(1), LANE)
        e("v_lshlrev_b32", EREG, Lit(9), T0.lo())                # 0x100 - (e << 9), e = 1 - (lane & 1)   (digit_pair)
        e("v_add_u32", EREG, Lit(-0x100), EREG)
        self.even_lanes_to_vcc()
        for name, val in (("C_2M64", 2.0 ** -64), ("C_M2P64", -(2.0 ** 64)), ("C_84", 1.5 * 2.0 ** 84), ("C_52", 1.5 * 2.0 ** 52),
                          ("C_SQH", 0.70710678118654752440)):
            s, b = getattr(SG, name), f64bits(val)
            e("s_mov_b32", s.lo(), Lit(b & 0xFFFFFFFF)); e("s_mov_b32", s.hi(), Lit(b >> 32))
        e("s_mov_b32", SG.C_M4338, Lit((-0x43380000) & 0xFFFFFFFF))
        if self.role == "A":
            for q in range(2):
                for m in range(32): e("v_mov_b32", V(self.AP[q] + m), Lit(0))      # acc = (0, 0, TV X^(-b~)): the mask polynomials start at zero
        else:
            for m in range(16): self.ds_read("ds_read_b64", V(self.AP[0] + 2 * m, 2), L8T, offset=512 * m)    # the wrapper left the rotated test vector in tile B
        e("v_mov_b32", R10, Lit(0))
        self.gload("global_load_dwordx2", V(10, 2), R10, SG.CT, offset=0)     # mask element 0
        e("s_mov_b32", SG.I, Lit(0))
        self.lg_wait_all()

    def step(self):
        e = self.e
        A_ = self.role == "A"
        NP, AP, DP, FP, X = self.NP, self.AP, self.DP, self.FP, self.X
        RA, SGN = self.RA, self.SGN
        self.p.label("L_step%=")
        self.c("---- modswitch(ct[i]) -> rot; base = (lane - rot) mod 2N")
        self.p.emit("s_waitcnt", vm=0); self.vmq.clear()
        e("v_readfirstlane_b32", SG.CTV.hi(), R11)
        e("s_nop", Lit(3))
        e("s_lshr_b32", SG.TMP, SG.CTV.hi(), Lit(23))
        e("s_and_b32", SG.TMP2, SG.TMP, Lit(1))
        e("s_add_u32", SG.TMP, SG.TMP, SG.TMP2)
        e("s_lshr_b32", SG.TMP, SG.TMP, Lit(1))
        e("s_lshl_b32", SG.ROT, SG.TMP, Lit(3))
        e("v_subrev_u32", T0.lo(), SG.ROT, LANE)
        e("v_and_b32", T0.lo(), Lit(2 * N - 1), T0.lo())
        e("v_lshrrev_b32", R10, Lit(6), T0.lo())
        e("v_and_b32", R11, Lit(63), T0.lo())
        e("v_lshl_add_u32", R11, R11, Lit(3), SG.TILE)
        for m in range(16):
            t = T0.hi()
            e("v_add_u32", t, Lit(m), R10)
            e("v_bfe_i32", SGN[m], t, Lit(4), Lit(1))
            e("v_and_b32", t, Lit(15), t)
            e("v_lshl_add_u32", RA[m], t, Lit(9), R11)
        e("s_add_u32", SG.TMP, SG.I, Lit(1))
        e("s_cmp_lt_u32", SG.TMP, SG.NSTEP)
        e("s_cbranch_scc1", "L_nx%=")
        e("s_mov_b32", SG.TMP, SG.I)
        self.p.label("L_nx%=")
        e("s_lshl_b32", SG.TMP, SG.TMP, Lit(3))
        e("v_mov_b32", R10, SG.TMP)
        self.gload("global_load_dwordx2", V(10, 2), R10, SG.CT, offset=0)
        # ---- key stream of this wave: A columns 0 and 1 (48 quads: q outer, then column, then register), B column 2 (24 quads);
        # ROW points at (row i, q = 0, this wave's first column)
        NG = 12
        cols = [0, 1] if A_ else [0]
        stream = [(q, c, r) for q in range(3) for c in cols for r in range(8)]
        ring = [quad(X, r) for r in [9, 10, 11, 0, 1, 2, 3, 4, 5, 6, 7, 8]]
        EARLY = 11

        def kp(q, c): return KP6[q * len(cols) + c]

        def issue(k):
            q, c, r = stream[k]
            self.gload("global_load_dwordx4", ring[k % NG], L16 if r < 4 else OFF1, kp(q, c), offset=(r & 3) * 1024)

        def early(lo, hi):
            if lo == 0:
                for q in range(3):
                    for c in cols:
                        e("s_add_u32", kp(q, c).lo(), SG.ROW.lo(), Lit((q * 3 + c) * 8192))
                        e("s_addc_u32", kp(q, c).hi(), SG.ROW.hi(), Lit(0))
            for k in range(lo, hi): issue(k)
        early(0, 3)
        # ---- rotation + digits of this wave's polynomial(s)
        self.c("---- rotation: stage the polynomial(s) in the tile, gather rotated coefficients; digits")
        temps = [T0.lo(), T0.hi(), T1.lo(), T1.hi()]

        def lds_traffic(q):
            for m in range(16): self.ds_write("ds_write_b64", L8T, V(AP[q] + 2 * m, 2), offset=512 * m)
            for m in range(8):
                self.ds_read("ds_read_b64", V(DP[q] + 4 * m, 2), RA[m])
                self.ds_read("ds_read_b64", V(DP[q] + 4 * m + 2, 2), RA[m + 8])

        def digits(q):
            chains = []
            for m in range(8):
                for half in range(2):
                    j = m + 8 * half
                    slot = V(DP[q] + 4 * m + 2 * half, 2)
                    chains.append(self.chain(self.digit_pair, slot, V(AP[q] + 2 * j), V(AP[q] + 2 * j + 1), SGN[j], temps[len(chains) % 4], CARRY[len(chains) % 4]))
                    if len(chains) == 4:
                        self.merge(chains); chains = []
        for q in range(NP): lds_traffic(q)       # in-order LDS: polynomial 1's stores queue behind polynomial 0's gathers
        for q in range(NP): digits(q)
        # ---- forward transform(s)
        xq = [quad(X, r) for r in range(12)]
        polys = [[quad(DP[p], r) for r in range(8)] for p in range(NP)]
        polys, spare = self.fwdP(polys, xq, before_pass3=lambda: early(3, EARLY))
        # ---- publish: own tile (and A's second transform behind both tiles), [register][lane] complex
        self.c("---- publish the transform(s); barrier; multiply-accumulate; first inverse pass; barrier")
        if "P" not in ABL2:
            for p in range(NP):
                for r in range(8): self.ds_write("ds_write_b128", L16T, polys[p][r], offset=(PBUF if p == 1 else 0) + 1024 * r)
        self.barrier()
        for k in range(EARLY, NG): issue(k)
        Fq = [[quad(FP[c], r) for r in range(8)] for c in range(len(cols))]

        def mac(k, d):
            q, c, r = stream[k]
            g, acc = ring[k % NG], Fq[c][r]
            re0 = Lit(0, f64=0.0) if q == 0 else qre(acc)
            im0 = Lit(0, f64=0.0) if q == 0 else qim(acc)
            self.e("v_fma_f64", qre(acc), qre(d), qre(g), re0)
            self.e("v_fma_f64", qim(acc), qre(d), qim(g), im0)
            self.e("v_fma_f64", qre(acc), -qim(d), qim(g), qre(acc))
            self.e("v_fma_f64", qim(acc), qim(d), qre(g), qim(acc))

        if A_:
            land = [quad(DP[0], r) for r in range(8)]           # B's transform lands where A's first one was (dead after q = 0)

            def operand(k):
                q, c, r = stream[k]
                return polys[q][r] if q < 2 else land[r]
            for k in range(0, len(stream), 2):
                self.merge([self.chain(mac, k, operand(k)), self.chain(mac, k + 1, operand(k + 1))])
                for kk in (k, k + 1):
                    if kk + NG < len(stream): issue(kk + NG)
                if k + 2 == 16 and "P" not in ABL2:             # q = 0 is done for both columns: fetch B's transform (tile B)
                    for r in range(8): self.ds_read("ds_read_b128", land[r], L16T, offset=TILE_BYTES + 1024 * r)
        else:
            land = [quad(self.LAND, r) for r in range(8)]
            dstream = [(q, r) for q in range(2) for r in range(8)]             # A's two transforms: tile A, then the buffer

            def fetch(i):
                q, r = dstream[i]
                if "P" in ABL2: return
                self.ds_read("ds_read_b128", land[i % 8], L16A, offset=(PBUF if q == 1 else 0) + 1024 * r)
            for i in range(8): fetch(i)

            def operand(k):
                q, c, r = stream[k]
                return land[r] if q < 2 else polys[0][r]        # (q * 8 + r) % 8 = r
            for k in range(0, len(stream), 2):
                self.merge([self.chain(mac, k, operand(k)), self.chain(mac, k + 1, operand(k + 1))])
                for kk in (k, k + 1):
                    if kk + NG < len(stream): issue(kk + NG)
                    if kk + 8 < len(dstream): fetch(kk + 8)
        # ---- inverse(s); the tiles are free for the exchanges once both waves have read what they need
        v, sp2 = self.invP(Fq, xq, self.barrier)
        self.c("---- polynomial(s) += round(inverse)")
        tmp = [(V(X + 4 * k, 2), V(X + 4 * k + 2, 2)) for k in range(4)]

        def upd(c, m, half, k):
            j = m + 8 * half
            x = V(v[c][m].i + 2 * half, 2)
            al, ah = V(AP[c] + 2 * j), V(AP[c] + 2 * j + 1)
            self.torus_add(x, al, ah, al, ah, tmp[k][0], tmp[k][1], CARRY[k])
        for c in range(NP):
            chains = []
            for m in range(8):
                for half in range(2):
                    chains.append(self.chain(upd, c, m, half, len(chains)))
                    if len(chains) == 4:
                        self.merge(chains); chains = []
        # ---- loop control
        self.lg_wait_all()
        e("s_add_u32", SG.ROW.lo(), SG.ROW.lo(), Lit(KP * KP * H * 16))
        e("s_addc_u32", SG.ROW.hi(), SG.ROW.hi(), Lit(0))
        e("s_add_u32", SG.I, SG.I, Lit(1))
        e("s_cmp_lt_u32", SG.I, SG.NSTEP)
        if "K" in brgen.ABL: self.vm_wait_all()
        assert self.vmq == [] and self.lgq == [], (self.vmq, self.lgq)
        e("s_cbranch_scc1", "L_step%=")

    def epilogue(self):
        e = self.e
        self.c("---- epilogue: store this wave's polynomial(s) (OUT points at the first)")
        self.vm_wait_all()
        e("v_lshlrev_b32", L8, Lit(3), LANE)
        e("v_add_u32", T0.lo(), Lit(4096), L8)
        for q in range(self.NP):
            if q:
                e("s_add_u32", SG.OUT.lo(), SG.OUT.lo(), Lit(8192))
                e("s_addc_u32", SG.OUT.hi(), SG.OUT.hi(), Lit(0))
            for m in range(16):
                self.gstore("global_store_dwordx2", L8 if m < 8 else T0.lo(), V(self.AP[q] + 2 * m, 2), SG.OUT, offset=512 * (m & 7))
        self.vm_wait_all()


def render_inc(prog, role):
    """operands: %0 ct (64), %1 first column of this wave in bsk row 0 (64), %2 out: this wave's first polynomial (64), %3 n, %4 this wave's tile"""
    head = ["s_mov_b64 %s, %%0" % SG.CT, "s_mov_b64 %s, %%1" % SG.ROW, "s_mov_b64 %s, %%2" % SG.OUT,
            "s_mov_b32 %s, %%3" % SG.NSTEP, "s_mov_b32 %s, %%4" % SG.TILE]
    lines = ["  " + h for h in head] + prog.render()
    out = ["// generated by tools/asmgen/brgen2.py (wave %s) -- do not edit" % role]
    for l in lines:
        out.append('"%s\\n"' % l.replace('"', '\\"').replace("%=", "%=" ))
    return "\n".join(out) + "\n"


def render_clobbers():
    regs = ["v%d" % i for i in range(256)] + ["s%d" % i for i in range(34, 96)] + ["vcc", "scc", "memory"]
    return "// generated by tools/asmgen/brgen2.py -- do not edit\n" + ", ".join('"%s"' % r for r in regs) + "\n"


if __name__ == "__main__":
    progs = {r: PairGen(r).build() for r in "AB"}
    if len(sys.argv) > 2 and sys.argv[1] == "--out":
        for r in "AB": open(os.path.join(sys.argv[2], "tx_br2%s_asm.inc" % r.lower()), "w").write(render_inc(progs[r], r))
        open(os.path.join(sys.argv[2], "tx_br2_asm_clobbers.inc"), "w").write(render_clobbers())
    else:
        for r in "AB":
            c = progs[r].count()
            print("wave %s: %d instructions; FP64 %d" % (r, sum(c.values()), sum(v for k, v in c.items() if k.endswith("_f64") or k == "v_cvt_f64_i32")))
