"""Generator of the LUT CMux tree (a10: evaluate_8_to_8_cipher_lut, server_encrypted_aes_decryption.rs:550-588) as gfx950
assembly: the eight CMux steps of k_lut_eval (csrc/tx_kernels.hpp) on one accumulator GLWE that lives in the wave's LDS
slice, one wave per SIMD, with the explicit schedule of the trace loop (trgen.py): three-way interleaved transforms whose
exchange stores are spread through the next polynomial's butterflies, and the selector GGSW streamed from L2 through a ring of
twelve register quads whose loads run eleven quads ahead of the multiply-accumulate (the hipcc kernel keeps one polynomial
in flight and waits for the next behind every 32 fma: 27 % of its cycles at s_waitcnt, profiles/r3a_pmc_sq_lut_eval_*).

    python3 tools/asmgen/lutgen.py --out temp-fhe-transciphering_amd/csrc      (tx_lut_asm.inc; clobbers: tx_trace_asm_clobbers.inc)

One CMux step, selector bit i (LSB first), acc <- acc + GGSW_i (x) (acc X^(-2^i) - acc), B = 2^2, l = 7:
  for row r = 0, 1, 2:  d = acc_r X^(-2^i) - acc_r  (gather through the LDS image, as the trace's automorphism);
      signed decomposition Decomp<2, 7>, finest level first, in groups of 3, 3, 1 levels (the order of extprod_row: the
      products enter every Fourier accumulator row by row, finest level first -- the oracle's order); the decomposition state
      of the sixteen coefficients waits in accumulation registers between the groups; each group: digits -> f64 -> forward
      transforms (three together, or the last level alone) -> F[c] += D * GGSW_i[level][r][c], c = 0, 1, 2;
  three inverse transforms together, acc_c += round(.) in LDS.
Arithmetic, operation order and results are those of k_lut_eval / the oracle's txo_lut_eval: bit-identical
(tests/asmgen_emu_lut.py runs the emitted stream against a reference built from the oracle's primitives and against
txo_lut_eval itself).

LDS image (k_lut_eval4, tx_br2.hpp): that of the trace kernels -- compact twiddles at 0, per wave the 8 320-byte tile and the
accumulator GLWE (3 x 1024 torus words, natural order) behind it.

Register map (VGPR): 0..15 as trgen.py (4: lane * 16 + 4096, 8: top-level tie mask); 16..31 chain temporaries of the
decomposition / four quads of the key ring / the transforms' spare quads; P0 32..63, P1 64..95, P2 96..127: words, then the digit
polynomials of a group and their transforms; F 128..223: the three Fourier accumulators of a CMux step; 224..255: gather
addresses and sign masks / twiddles / eight quads of the key ring.  AGPR 0..15: decomposition states of a row.
"""
import os
import sys
from gcn import V, S, A, Lit
import brgen
import brgen3
import trgen
from brgen import SG, quad, qre, qim, T0, T1, LANE, L8T, L16, CARRY
from trgen import TS, TraceGen, P, GA, GS, GL, KP, N, DMAOFF

CBS_B, CBS_L = 2, 7
H = 512
F = 128                                   # three accumulators of 32 registers
XQ8 = [quad(224, r) for r in range(8)]    # twiddles in the transforms
SP3 = [V(20, 4), V(24, 4), V(28, 4)]      # spare quads of the three-way transforms
RING = XQ8 + SP3 + [V(16, 4)]             # key ring: twelve quads, in slot order
CH = [(V(16 + 4 * k), V(17 + 4 * k), V(18 + 4 * k), V(19 + 4 * k)) for k in range(4)]     # (state, res, u, carry) of four chains
GROUPS = [(0, 3), (3, 3), (6, 1)]         # (first digit index (0 = finest), levels) of the groups of a row
GGSW_BYTES = CBS_L * KP * KP * H * 16     # one selector GGSW in the Fourier domain: 516 096 B


class LutGen(TraceGen):
    DIG_B, DIG_L = CBS_B, CBS_L

    def __init__(self, bits=8, stamps=False):
        super().__init__(stamps=stamps)
        self.bits = bits

    # ------------------------------------------------------------------ pieces
    def gather_table(self):
        """coefficient j of p X^(-k), k = 2^bit, is p[j + k] for j + k < N and -p[j + k - N] otherwise (k <= 128: one wrap)"""
        e = self.e
        e("s_lshl_b32", TS.T, Lit(1), TS.IDX)
        cur = CH[0][0]
        e("v_add_u32", cur, TS.T, LANE)
        for m in range(16):
            if m: e("v_add_u32", cur, Lit(64), cur)
            e("v_bfe_i32", GS[m], cur, Lit(trgen.LOGN), Lit(1))
            e("v_and_b32", CH[0][1], Lit(N - 1), cur)
            e("v_lshl_add_u32", GA[m], CH[0][1], Lit(3), TS.GLB)

    def row_states(self, r):
        """d = +-p[j + k] - p[j] for this lane's sixteen coefficients of row r; the decomposition needs only the high dword:
        state = ((d >> 49) + 1) >> 1 = (d.hi + 2^17) >> 18 (the carry out of bit 63 is dropped either way); parked in a0..a15"""
        X = [V(P[0] + 2 * j, 2) for j in range(16)]
        OWN = [V(P[1] + 2 * j, 2) for j in range(16)]
        for m in range(8):
            self.ds_read("ds_read_b64", X[m], GA[m], offset=r * 8192)
            self.ds_read("ds_read_b64", X[m + 8], GA[m + 8], offset=r * 8192)
        for j in range(16): self.ds_read("ds_read_b64", OWN[j], L8T, offset=GL + r * 8192 + 512 * j)

        def one(j, k):
            e = self.e
            lo, hi, cy = X[j].lo(), X[j].hi(), CARRY[k]
            e("v_xor_b32", lo, lo, GS[j]); e("v_xor_b32", hi, hi, GS[j])
            e("v_sub_co_u32", lo, cy, lo, GS[j]); e("v_subb_co_u32", hi, cy, hi, GS[j], cy)
            e("v_sub_co_u32", lo, cy, lo, OWN[j].lo()); e("v_subb_co_u32", hi, cy, hi, OWN[j].hi(), cy)
            e("v_add_u32", hi, Lit(1 << 17), hi)
            e("v_lshrrev_b32", hi, Lit(64 - CBS_B * CBS_L - 32), hi)
            e("v_accvgpr_write_b32", A(j), hi)
        for j0 in range(0, 16, 4):
            self.merge([self.chain(one, j0 + k, k) for k in range(4)])

    def group_digits(self, first, nlev):
        """levels first .. first + nlev - 1 (0 = finest; CBS_L - 1 = the top level with the balanced tie) of the row in a0..a15
        -> f64 digit polynomials in P[0 .. nlev - 1]; Decomp<B, L>::next of tx_kernels.hpp / txo_decompose_idx"""
        B = CBS_B

        def one(j, k):
            e = self.e
            st, res, u, car = CH[k]
            m, half = j % 8, j // 8
            e("v_accvgpr_read_b32", st, A(j))
            for i in range(nlev):
                lev = first + i
                e("v_and_b32", res, Lit((1 << B) - 1), st)
                e("v_lshrrev_b32", st, Lit(B), st)
                e("v_add_u32", u, Lit(-1), res)
                e("v_or_b32", u, u, st)
                if lev == CBS_L - 1: e("v_or_b32", u, u, trgen.TIE)
                e("v_and_b32", u, u, res)
                e("v_bfe_i32", car, u, Lit(B - 1), Lit(1))          # -carry (0 or -1)
                if lev < CBS_L - 1: e("v_sub_u32", st, st, car)     # state += carry
                e("v_lshl_add_u32", res, car, Lit(B), res)          # res -= carry << B
                e("v_cvt_f64_i32", V(P[i] + 4 * m + 2 * half, 2), res)
            if first + nlev < CBS_L: e("v_accvgpr_write_b32", A(j), st)
        for j0 in range(0, 16, 4):
            self.merge([self.chain(one, j0 + k, k) for k in range(4)])

    def key_off(self, lev, r, c):
        """GGSW [level 1 (coarsest) first][row][column][512] complex; digit index lev counts from the finest level"""
        return (((CBS_L - 1 - lev) * KP + r) * KP + c) * H * 16

    def mac_group(self, polys, first, r, fresh):
        """F[c] += polys[i] * GGSW[first + i][r][c] for i in order, c = 0, 1, 2: 24 key quads per digit polynomial through the
        ring; `fresh`: the first products of a CMux step start the accumulators"""
        e = self.e
        stream = [(i, c, rr) for i in range(len(polys)) for c in range(KP) for rr in range(8)]
        NG = len(RING)
        Fq = [[quad(F + 32 * c, rr) for rr in range(8)] for c in range(KP)]

        def issue(k):
            i, c, rr = stream[k]
            if rr == 0:
                e("s_add_u32", TS.KP_.lo(), TS.PB.lo(), Lit(self.key_off(first + i, r, c)))
                e("s_addc_u32", TS.KP_.hi(), TS.PB.hi(), Lit(0))
            self.gload("global_load_dwordx4", RING[k % NG], L16 if rr < 4 else DMAOFF, TS.KP_, offset=(rr & 3) * 1024)

        def mac(k):
            i, c, rr = stream[k]
            g, d, f = RING[k % NG], polys[i][rr], Fq[c][rr]
            zero = fresh and i == 0
            re0 = Lit(0, f64=0.0) if zero else qre(f)
            im0 = Lit(0, f64=0.0) if zero else qim(f)
            self.e("v_fma_f64", qre(f), qre(d), qre(g), re0)
            self.e("v_fma_f64", qim(f), qre(d), qim(g), im0)
            self.e("v_fma_f64", qre(f), -qim(d), qim(g), qre(f))
            self.e("v_fma_f64", qim(f), qim(d), qre(g), qim(f))
        for k in range(min(NG, len(stream))): issue(k)              # the ring full: twelve quads = one and a half key polynomials ahead
        for k in range(0, len(stream), 2):
            self.merge([self.chain(mac, k), self.chain(mac, k + 1)])
            for kk in (k, k + 1):                                   # a slot is refilled as soon as its quad has been multiplied
                if kk + NG < len(stream): issue(kk + NG)

    # ------------------------------------------------------------------ the kernel body
    def prologue(self):
        super().prologue()
        self.e("v_add_u32", DMAOFF, Lit(4096), L16)                 # rows 4..7 of a key polynomial
        self.e("s_mov_b32", TS.PB.lo(), TS.AUTO.lo()); self.e("s_mov_b32", TS.PB.hi(), TS.AUTO.hi())

    def step(self):
        e = self.e
        self.p.label("L_lut%=")
        self.c("---- CMux step: selector bit IDX, rotation by 2^IDX")
        regions = lambda n: [[quad(P[i], rr) for rr in range(8)] for i in range(n)]
        xq3 = XQ8 + SP3
        for r in range(KP):
            self.c("---- row %d: difference, decomposition state" % r)
            self.gather_table()                                  # (its registers carry twiddles and key quads in between)
            self.row_states(r)
            for gi, (first, nlev) in enumerate(GROUPS):
                self.c("---- row %d, levels %d..%d: digits, forward transform%s, multiply-accumulate" % (r, first, first + nlev - 1, "s" if nlev > 1 else ""))
                self.group_digits(first, nlev)
                self.stamp(0)
                if nlev == 3:
                    polys, _ = self.fwd3(regions(3), xq3)
                else:
                    v, _ = brgen3.CoopGen.fwd1(self, regions(1)[0], XQ8 + [SP3[0]])
                    polys = [v]
                self.stamp(1)
                self.mac_group(polys, first, r, fresh=(r == 0 and gi == 0))
                self.stamp(2)
        Fq = [[quad(F + 32 * c, rr) for rr in range(8)] for c in range(KP)]
        xq = [quad(P[0], rr) for rr in range(8)] + [quad(P[1], rr) for rr in range(3)]
        Fq, spare = self.inv3(Fq, xq)
        self.stamp(3)
        self.c("---- acc[c] += round(inverse)")
        for c in range(KP):
            CUR = [V(P[2] + 2 * j, 2) for j in range(16)]
            for j in range(16): self.ds_read("ds_read_b64", CUR[j], L8T, offset=GL + c * 8192 + 512 * j)
            tmp = [(V(P[1] + 16 + 4 * k, 2), V(P[1] + 16 + 4 * k + 2, 2)) for k in range(4)]

            def upd(m, half, k, c=c, CUR=CUR, tmp=tmp):
                j = m + 8 * half
                x = V(Fq[c][m].i + 2 * half, 2)
                self.torus_add(x, CUR[j].lo(), CUR[j].hi(), CUR[j].lo(), CUR[j].hi(), tmp[k][0], tmp[k][1], CARRY[k])
                self.ds_write("ds_write_b64", L8T, CUR[j], offset=GL + c * 8192 + 512 * j)
            chains = []
            for m in range(8):
                for half in range(2):
                    chains.append(self.chain(upd, m, half, len(chains)))
                    if len(chains) == 4:
                        self.merge(chains); chains = []
        self.stamp(4)
        # ---- loop control: the next selector GGSW
        self.lg_wait_all()
        e("s_add_u32", TS.PB.lo(), TS.PB.lo(), Lit(GGSW_BYTES))
        e("s_addc_u32", TS.PB.hi(), TS.PB.hi(), Lit(0))
        e("s_add_u32", TS.IDX, TS.IDX, Lit(1))
        e("s_cmp_lt_u32", TS.IDX, Lit(self.bits))
        assert self.vmq == [] and self.lgq == [], (self.vmq, self.lgq)
        e("s_cbranch_scc1", "L_lut%=")

    def build(self):
        self.prologue()
        self.step()
        self.lg_wait_all()
        if self.stamps:
            e = self.e
            e("v_mov_b32", T0.hi(), Lit(0))
            for k in range(brgen.NSTAMP):
                e("v_mov_b32", T1.lo(), brgen.ST_ACC[k].lo()); e("v_mov_b32", T1.hi(), brgen.ST_ACC[k].hi())
                e("global_store_dwordx2", T0.hi(), T1, brgen.ST_DBG, offset=8 * k)
            self.vm_wait_all()
        return self.p


def render_inc(prog, stamps=False):
    """operands: %0 the byte's eight selector GGSWs, bit 0 first (64), %1 this wave's tile; stamps: %2 the debug buffer (64)"""
    # (the tie mode is the constant 1 here: this experiment kernel has the balanced rule only)
    head = ["s_mov_b64 %s, %%0" % TS.AUTO, "s_mov_b32 %s, %%1" % TS.TILE, "s_mov_b32 %s, 0" % TS.WAVE, "s_mov_b32 %s, %%1" % TS.SLOT0,
            "s_mov_b32 %s, 1" % TS.TIEMODE]
    if stamps: head.append("s_mov_b64 %s, %%2" % brgen.ST_DBG)
    lines = ["  " + h for h in head] + ["  s_mov_b32 %s, m0" % TS.M0SAVE] + prog.render() + ["  s_mov_b32 m0, %s" % TS.M0SAVE]
    out = ["// generated by tools/asmgen/lutgen.py -- do not edit"]
    for l in lines: out.append('"%s\\n"' % l.replace('"', '\\"'))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    prog = LutGen().build()
    if len(sys.argv) > 2 and sys.argv[1] == "--out":
        open(os.path.join(sys.argv[2], "tx_lut_asm.inc"), "w").write(render_inc(prog))
        open(os.path.join(sys.argv[2], "tx_lut_asm_stamps.inc"), "w").write(render_inc(LutGen(stamps=True).build(), stamps=True))
    else:
        c = prog.count()
        for k in sorted(c, key=lambda k: -c[k]): print("%6d %s" % (c[k], k))
        print("%6d total" % sum(c.values()))
