"""Generator of the scheme-switching kernel body (a8 + a9: GLev element -> three GGSW rows in the Fourier domain) as
gfx950 assembly: k_scheme_switch of csrc/tx_kernels.hpp with the explicit schedule of the trace loop (trgen.py) -- one
wave per SIMD (six digit transforms live across the two passes over the key), the digit transforms parked in
accumulation registers, two key register sets behind the workgroup's LDS ring, transforms whose exchange stores are
spread through the next polynomial's butterflies.  Same products in the same order as k_scheme_switch and the oracle
(txo_scheme_switch + txo_ggsw_to_fourier): bit-identical.

    python3 tools/asmgen/ssgen.py --out temp-fhe-transciphering_amd/csrc      (tx_ss_asm.inc; clobbers: tx_trace_asm_clobbers.inc)

Unit = (ciphertext, level): the wrapper (k_scheme_switch4, tx_br2.hpp) puts the GLev element (3 x 1024 torus words)
behind the wave's tile, as the trace kernel does with its GLWE.  Output [row 3][col 3][512] complex at OUT:
rows 0, 1 = F(round(F^-1(sum_j D[j / 3] ss_key_i(j)))) 2^-9, row 2 = F(input) 2^-9.
Digit polynomial index r * 2 + i (r: GLev polynomial, i = 0 the finest level) = accumulation-register block.
"""
import os
import sys
from gcn import V, S, A, Lit
import brgen
import trgen
from brgen import SG, quad, qre, qim, T0, T1, L8T, L16, CARRY, f64bits
from trgen import TS, TraceGen, P, DC, K, GA, GS, TMP, GL, KP, N

SS_B, SS_L = 17, 2
OUT = S(36, 2)          # output pointer (the trace loop's SPLIT / a free register)
OFF1 = V(9)             # lane * 16 + 4096: rows 4..7 of an output polynomial (the trace loop's gather index register)
H = 512


class SsGen(TraceGen):
    DIG_B, DIG_L = SS_B, SS_L

    def __init__(self, stamps=False):
        super().__init__(stamps=stamps)
        self.sel = 0

    def key_off(self, j):
        """SsKeyFn of tx_kernels.hpp: ss_f [i 2][lev 2][row 3][col 3][512]; digit polynomial j // 3 = (GLev polynomial r,
        level finest first), column j % 3"""
        r, i, c = j // (SS_L * KP), (j // KP) % SS_L, j % KP
        return ((((self.sel * SS_L + (SS_L - 1 - i)) * KP + r) * KP + c) * H) * 16

    # ------------------------------------------------------------------ pieces
    def load_words(self, r, X):
        for j in range(16): self.ds_read("ds_read_b64", X[j], L8T, offset=GL + r * 8192 + 512 * j)

    def decompose(self, X, homes):
        """16 words X[j] (pairs, destroyed) -> SS_L digit polynomials as f64 in the 32-register regions homes[level]"""
        chains = []
        for m in range(8):
            for half in range(2):
                j = m + 8 * half
                outs = [V(homes[lev] + 4 * m + 2 * half, 2) for lev in range(SS_L)]
                k = len(chains)
                chains.append(self.chain(self.digits_bl, X[j], None, outs, TMP[3 * k:3 * k + 3], CARRY[k], SS_B, SS_L))
                if len(chains) == 4:
                    self.merge(chains); chains = []

    def park(self, polys, blocks):
        for p, g in enumerate(blocks):
            for r in range(8):
                for k in range(4):
                    self.e("v_accvgpr_write_b32", A((g * 8 + r) * 4 + k), V(polys[p][r].i + k))

    def to_signed_f64(self, x, tA, tB):
        """x <- (double)(int64)(round-half-even(x) mod 2^64): f64_to_torus then torus_to_f64 of tx_device.hpp"""
        e = self.e
        e("v_mul_f64", tA, x, SG.C_2M64)
        e("v_rndne_f64", tA, tA)
        e("v_fma_f64", x, tA, SG.C_M2P64, x)
        e("v_add_f64", tA, x, SG.C_84)
        e("v_add_f64", tB, tA, -SG.C_84)
        e("v_add_f64", x, x, -tB)
        e("v_add_f64", x, x, SG.C_52)
        e("v_add3_u32", tA.lo(), tA.lo(), x.hi(), SG.C_M4338)      # high dword of the torus value; the low one is x.lo
        e("v_cvt_f64_i32", tB, tA.lo())
        e("v_cvt_f64_u32", x, x.lo())
        e("v_fma_f64", x, tB, -TS.C_M2P32, x)                       # hi 2^32 + lo, one rounding

    def words_to_f64(self, w, tB):
        """w (torus word pair) <- (double)(int64)w"""
        e = self.e
        e("v_cvt_f64_i32", tB, w.hi())
        e("v_cvt_f64_u32", w, w.lo())
        e("v_fma_f64", w, tB, -TS.C_M2P32, w)

    # SS_NT=1 (default since round 5): the Fourier GGSW rows leave with the non-temporal hint.  A launch writes 3.9 GB of them
    # (72 KiB per unit) through the same L2 that serves the 3.2 MB of automorphism / scheme-switching keys every wave streams:
    # counters of the round-4 form showed 2.2 GB of fabric reads per launch for 0.2 GB of input -- the keys, evicted by the
    # output stream and re-fetched.  The rows are read next by another kernel from HBM either way (4.2 GB > any cache).
    NT = os.environ.get("SS_NT", "1") != "0"

    def scale_store(self, polys, row):
        """three transforms -> x 2^-9 -> OUT[row][c][r][lane]"""
        e = self.e
        for c in range(KP):
            e("s_add_u32", TS.KP_.lo(), OUT.lo(), Lit((row * KP + c) * 8192))
            e("s_addc_u32", TS.KP_.hi(), OUT.hi(), Lit(0))
            for r in range(8):
                q = polys[c][r]
                e("v_mul_f64", qre(q), qre(q), TS.C_2M32)          # this loop keeps 2^-9 in that register pair (prologue)
                e("v_mul_f64", qim(q), qim(q), TS.C_2M32)
                self.gstore("global_store_dwordx4", L16 if r < 4 else OFF1, q, TS.KP_, offset=(r & 3) * 1024, cache="nt" if self.NT else None)

    # ------------------------------------------------------------------ the kernel body
    def prologue(self):
        super().prologue()
        e = self.e
        b = f64bits(2.0 ** -9)
        e("s_mov_b32", TS.C_2M32.lo(), Lit(b & 0xFFFFFFFF)); e("s_mov_b32", TS.C_2M32.hi(), Lit(b >> 32))   # this loop's scale
        e("v_add_u32", OFF1, Lit(4096), L16)

    def body(self):
        e = self.e
        X = [V(K[0] + 2 * j, 2) for j in range(16)]
        xq16 = [quad(K[0], r) for r in range(8)] + [quad(K[1], r) for r in range(8)]
        EXTRA = GA[0].i                                          # 224..255: the digit polynomial carried from batch 0 to batch 1
        regions = lambda homes: [[quad(h, r) for r in range(8)] for h in homes]
        self.c("---- digits of the three GLev polynomials (2 levels each), transformed three at a time, parked")
        self.load_words(0, X); self.decompose(X, [P[0], P[1]])
        self.load_words(1, X); self.decompose(X, [P[2], EXTRA])
        polys, _ = self.fwd3(regions([P[0], P[1], P[2]]), xq16)
        self.park(polys, [0, 1, 2])
        self.load_words(2, X); self.decompose(X, [P[1], P[2]])
        polys, _ = self.fwd3(regions([EXTRA, P[1], P[2]]), xq16)
        self.park(polys, [3, 4, 5])
        self.stamp(0)
        for i in range(2):
            self.sel = i
            self.c("---- mask row %d: 18 key polynomials through the ring, inverse, round, forward, store" % i)
            e("s_mov_b32", TS.PB.lo(), TS.AUTO.lo()); e("s_mov_b32", TS.PB.hi(), TS.AUTO.hi())
            # The stores of rows 0 and 2 may still be in flight under the second pass: they are OLDER than every ring load,
            # so a counted wait for a ring load can only be satisfied late by them, never early (loads return in order).
            Fq = self.key_pass()
            self.stamp(1)
            xq = xq16 + [quad(DC, r) for r in range(8)]
            Fq, spare = self.inv3(Fq, xq)
            self.stamp(2)
            tmp = [(V(K[0] + 4 * k, 2), V(K[0] + 4 * k + 2, 2)) for k in range(4)]
            chains = []
            for c in range(KP):
                for m in range(8):
                    for half in range(2):
                        k = len(chains)
                        chains.append(self.chain(self.to_signed_f64, V(Fq[c][m].i + 2 * half, 2), tmp[k][0], tmp[k][1]))
                        if len(chains) == 4:
                            self.merge(chains); chains = []
            polys, _ = self.fwd3(Fq, xq16)
            self.scale_store(polys, i)
            self.stamp(3)
            if i == 0:
                self.c("---- body row: the GLev element itself, transformed")
                homes = [P[0], P[1], P[2]]
                tB = [V(K[0] + 2 * k, 2) for k in range(4)]
                for r in range(KP):
                    W = [V(homes[r] + 4 * m + 2 * half, 2) for half in range(2) for m in range(8)]        # W[j], j = m + 8 half
                    for j in range(16): self.ds_read("ds_read_b64", W[j], L8T, offset=GL + r * 8192 + 512 * j)
                    for j0 in range(0, 16, 4):
                        self.merge([self.chain(self.words_to_f64, W[j0 + k], tB[k]) for k in range(4)])
                polys, _ = self.fwd3(regions(homes), xq16)
                self.scale_store(polys, 2)
                self.stamp(4)

    def build(self):
        self.prologue()
        self.skew()
        self.body()
        self.lg_wait_all()
        self.vm_wait_all()
        if self.stamps:
            e = self.e
            e("v_mov_b32", T0.hi(), Lit(0))
            for k in range(brgen.NSTAMP):
                e("v_mov_b32", T1.lo(), brgen.ST_ACC[k].lo()); e("v_mov_b32", T1.hi(), brgen.ST_ACC[k].hi())
                e("global_store_dwordx2", T0.hi(), T1, brgen.ST_DBG, offset=8 * k)
            self.vm_wait_all()
        return self.p


def render_inc(prog, stamps=False):
    """operands: %0 ss_f (64), %1 output of this unit (64), %2 wave index, %3 LDS address of slot 0, %4 this wave's tile,
    %5 the tie mode (1 = balanced top-level tie, 0 = tfhe-rs' rule); stamps: %6 the debug buffer"""
    head = ["s_mov_b64 %s, %%0" % TS.AUTO, "s_mov_b64 %s, %%1" % OUT, "s_mov_b32 %s, %%2" % TS.WAVE,
            "s_mov_b32 %s, %%3" % TS.SLOT0, "s_mov_b32 %s, %%4" % TS.TILE, "s_mov_b32 %s, %%5" % TS.TIEMODE]
    if stamps: head.append("s_mov_b64 %s, %%6" % brgen.ST_DBG)
    lines = ["  " + h for h in head] + ["  s_mov_b32 %s, m0" % TS.M0SAVE] + prog.render() + ["  s_mov_b32 m0, %s" % TS.M0SAVE]
    out = ["// generated by tools/asmgen/ssgen.py -- do not edit"]
    for l in lines: out.append('"%s\\n"' % l.replace('"', '\\"'))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    prog = SsGen().build()
    if len(sys.argv) > 2 and sys.argv[1] == "--out":
        open(os.path.join(sys.argv[2], "tx_ss_asm.inc"), "w").write(render_inc(prog))
        open(os.path.join(sys.argv[2], "tx_ss_asm_stamps.inc"), "w").write(render_inc(SsGen(stamps=True).build(), stamps=True))
    else:
        c = prog.count()
        for k in sorted(c, key=lambda k: -c[k]): print("%6d %s" % (c[k], k))
        print("%6d total" % sum(c.values()))
