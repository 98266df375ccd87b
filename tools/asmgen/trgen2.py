"""Generator of the COOPERATIVE trace loop (a7): TWO waves per unit (k_extract_trace5), so that the trace kernel, too, runs
at two waves per SIMD.

Why it needs two waves on ONE unit: a trace step multiplies six digit transforms with both halves of the split key, and
the one-wave loop (trgen.py) must keep all six live across its two passes (192 registers, next to 24 KiB of GLWE in LDS):
one unit per SIMD is all that fits, the vector pipe is busy half of the time (PMC: 50.5 %), and a second unit per SIMD has
nowhere to live.  Two waves that split ONE unit's step do fit, because together they need the six transforms only one at a
time: wave A accumulates the low-split pass, wave B the high-split pass, each into its own three Fourier accumulators, and a
digit transform is multiplied into both as soon as it exists.  Per step and wave: the full decomposition of both gathered
mask polynomials (cheap integer work, done twice so that nothing but transforms crosses), THREE forward transforms (A takes
digit polynomials 0, 2, 4 of the six, B takes 1, 3, 5; a wave publishes each transform through its transpose tile, idle at
that point), 18 multiply-accumulates against its half of the key (streamed from L2 into a register ring as the blind
rotation does: no LDS ring, no per-polynomial barrier), three inverse transforms, and the update.  The GLWE's mask
polynomial q lives in the registers of wave q for all ten steps (staged through the tile for the automorphism's gather);
only the body polynomial stays in LDS.  What crosses between the waves: 3 x 8 KiB of transforms each way and one polynomial
of rounded results each way per step, ten s_barrier (of two waves).

Same arithmetic as trgen.py / trace_lds() / the oracle's txo_trace(): every accumulator takes the six digit transforms in
the order 0..5, transforms and conversions are the same instruction sequences, integer additions commute.  Bit-identical;
tests/asmgen_emu_trace2.py runs the two emulated waves (shared LDS, barriers honoured, both scheduling orders) against
txo_trace() with real keys.

    python3 tools/asmgen/trgen2.py --out temp-fhe-transciphering_amd/csrc      (tx_trace2_asm.inc + clobbers)

LDS image of a workgroup (128 threads = one unit): compact twiddles at 0 (9 088 B), tile of wave A, tile of wave B (8 320 B
each, pitch 65 complex), body polynomial (8 KiB): 33 920 B, four workgroups per CU.

Register map (VGPR, 256): 0..7 as brgen.py, 8 top-level tie mask, 9 partner tile + lane * 8, 10 tile A + lane * 16, 11 tile A
+ lane * 8, 12..15 temporaries; OWN 16..47 this wave's mask polynomial; PD 48..79 its three digit polynomials, packed
(two 16-bit fields + one dword per coefficient); F0 80..111, F1 112..143, F2 144..175 the Fourier accumulators of this
wave's own column, the partner's column and the body column (gather tables and gathered words before the first
multiply-accumulate); D 176..207 the digit polynomial being transformed / landing ring of published transforms; X 208..255
key ring of 12 quads (twiddles + spare in the transforms).
"""
import os
import sys
from gcn import V, S, Lit, VCC, Prog
import brgen
import trgen
from brgen import Gen, SG, quad, qre, qim, f64bits, T0, T1, LANE, L8T, L16T, L16, OFF1, TWBA, XHW, XLW, CARRY, PITCH
from trgen import TraceGen, TS, AUTO_B, AUTO_L, AUTO_SPLIT, N, LOGN, KP

TIE, L8P, L16A, L8A = V(8), V(9), V(10), V(11)
OWN, PD, F, D, X = 16, 48, [80, 112, 144], 176, 208
TILE_BYTES = 8 * PITCH                  # 8 320
BODY_OFF = 2 * TILE_BYTES               # the body polynomial, from tile A
GA = [V(F[0] + m) for m in range(16)]          # gather addresses relative to tile A (before the first multiply-accumulate)
GS = [V(F[0] + 16 + m) for m in range(16)]     # gather sign masks
ABL2 = os.environ.get("TR2_ABL", "")          # timing-only ablations (B no barriers, K no key loads, P no publish / landing reads): outputs invalid


class T2:      # scalar registers (constants at 48..59 and carries at 88..95 as brgen; 64..69 as trgen)
    AUTO = S(34, 2); SPLIT = S(36); W = S(37); TILEP = S(38); TILEA = S(39); IDX = S(41); TILE = SG.TILE; TINV = S(43); C64 = S(44)
    T = S(45); Tb = S(46); SELB = S(60, 2); OFFO = S(62); OFFT = S(63)
    C_M2P32 = TS.C_M2P32; C_1P24 = TS.C_1P24; C_2M32 = TS.C_2M32
    PB = S(70, 2); KP6 = [S(72 + 2 * k, 2) for k in range(6)]; CC = S(84, 2); CS = S(86, 2)


class Trace2Gen(TraceGen):
    CC, CS = T2.CC, T2.CS

    def __init__(self, steps=LOGN):
        super().__init__(steps=steps, stamps=False)

    def barrier(self):
        self.lg_wait_all()
        if "B" not in ABL2: self.p.emit("s_barrier")

    def drain(self):
        """both memory counters at zero: the state in front of and behind a section that only one of the two waves runs"""
        self.lg_wait_all(); self.vm_wait_all()

    def only(self, wave, fn):
        """fn() runs on wave `wave` (0 = A, 1 = B) only; the in-order queues are empty on both sides of the branch"""
        self.drain()
        self.nlab += 1
        lab = "L_only%d_%%=" % self.nlab
        self.e("s_cmp_lg_u32", T2.W, Lit(0))
        self.e("s_cbranch_scc1" if wave == 0 else "s_cbranch_scc0", lab)
        fn()
        self.drain()
        self.p.label(lab)

    fwd1 = None     # set below (CoopGen's single-polynomial forward transform)

    # ------------------------------------------------------------------ decomposition to integers (digits_bl without the conversion)
    def digits_int(self, x, sgn, res3, u, car, cy):
        lo, hi = x.lo(), x.hi()
        e = self.e
        B, L = AUTO_B, AUTO_L
        total = B * L
        e("v_xor_b32", lo, lo, sgn)
        e("v_xor_b32", hi, hi, sgn)
        e("v_sub_co_u32", lo, cy, lo, sgn)
        e("v_subb_co_u32", hi, cy, hi, sgn, cy)
        e("v_add_co_u32", lo, cy, T2.C_1P24, lo)
        e("v_addc_co_u32", hi, cy, Lit(0), hi, cy)
        e("v_alignbit_b32", lo, hi, lo, Lit(64 - total))
        e("v_lshrrev_b32", hi, Lit(64 - total), hi)
        for lev in range(L):
            res = res3[lev]
            e("v_and_b32", res, Lit((1 << B) - 1), lo)
            if lev == 0: e("v_alignbit_b32", lo, hi, lo, Lit(B))
            else: e("v_lshrrev_b32", lo, Lit(B), lo)
            e("v_add_u32", u, Lit(-1), res)
            e("v_or_b32", u, u, lo)
            if lev == L - 1: e("v_or_b32", u, u, TIE)
            e("v_and_b32", u, u, res)
            e("v_bfe_i32", car, u, Lit(B - 1), Lit(1))
            if lev < L - 1: e("v_sub_u32", lo, lo, car)
            e("v_lshl_add_u32", res, car, Lit(B), res)

    # ------------------------------------------------------------------ the kernel
    def prologue(self):
        e = self.e
        self.c("---- prologue (cooperative trace): addresses, constants, this wave's mask polynomial from its tile")
        self.p.emit("s_waitcnt", vm=0, lgkm=0)
        e("v_mbcnt_lo_u32_b32", LANE, Lit(-1), Lit(0))
        e("v_mbcnt_hi_u32_b32", LANE, Lit(-1), LANE)
        e("v_lshl_add_u32", L8T, LANE, Lit(3), T2.TILE)
        e("v_lshl_add_u32", L16T, LANE, Lit(4), T2.TILE)
        e("v_lshlrev_b32", L16, Lit(4), LANE)
        e("v_add_u32", OFF1, Lit(4096), L16)
        # partner's tile: tile A + (1 - w) * 8320
        e("s_sub_u32", T2.T, Lit(1), T2.W)
        e("s_mul_i32", T2.T, T2.T, Lit(TILE_BYTES))
        e("s_add_u32", T2.TILEP, T2.TILEA, T2.T)
        e("v_lshl_add_u32", L8P, LANE, Lit(3), T2.TILEP)
        e("v_lshl_add_u32", L16A, LANE, Lit(4), T2.TILEA)
        e("v_lshl_add_u32", L8A, LANE, Lit(3), T2.TILEA)
        e("s_sub_u32", T2.SELB.lo(), Lit(0), T2.W)              # all ones on wave B
        e("s_sub_u32", T2.SELB.hi(), Lit(0), T2.W)
        e("s_lshl_b32", T2.OFFO, T2.W, Lit(13))                 # this wave's own column: w * 8192 into a key row
        e("s_sub_u32", T2.T, Lit(1), T2.W)
        e("s_lshl_b32", T2.OFFT, T2.T, Lit(13))                 # the partner's column
        e("v_and_b32", T0.lo(), Lit(7), LANE)
        e("v_lshlrev_b32", TWBA, Lit(4), T0.lo())
        e("v_add_u32", TWBA, Lit(brgen.TWB), TWBA)
        e("v_lshrrev_b32", T0.hi(), Lit(3), LANE)
        e("v_lshl_add_u32", T1.lo(), T0.hi(), Lit(6), T0.hi())  # a*65
        e("v_add_u32", T0.hi(), T1.lo(), T0.lo())               # a*65 + b
        e("v_lshl_add_u32", XHW, T0.hi(), Lit(4), T2.TILE)
        e("v_and_b32", T1.lo(), Lit(0x38), LANE)
        e("v_lshl_add_u32", T1.hi(), T0.lo(), Lit(6), T0.lo())  # b*65
        e("v_add_u32", T1.lo(), T1.hi(), T1.lo())               # b*65 + g8
        e("v_lshl_add_u32", XLW, T1.lo(), Lit(4), T2.TILE)
        e("v_and_b32", T0.lo(), Lit(1), LANE)
        e("v_lshlrev_b32", TIE, Lit(AUTO_B - 1), T0.lo())
        for name, val in (("C_2M64", 2.0 ** -64), ("C_M2P64", -(2.0 ** 64)), ("C_84", 1.5 * 2.0 ** 84), ("C_52", 1.5 * 2.0 ** 52),
                          ("C_SQH", 0.70710678118654752440)):
            s_, b = getattr(SG, name), f64bits(val)
            e("s_mov_b32", s_.lo(), Lit(b & 0xFFFFFFFF)); e("s_mov_b32", s_.hi(), Lit(b >> 32))
        for s_, val in ((T2.C_2M32, 2.0 ** -32), (T2.C_M2P32, -(2.0 ** 32))):
            b = f64bits(val)
            e("s_mov_b32", s_.lo(), Lit(b & 0xFFFFFFFF)); e("s_mov_b32", s_.hi(), Lit(b >> 32))
        e("s_mov_b32", SG.C_M4338, Lit((-0x43380000) & 0xFFFFFFFF))
        e("s_mov_b32", T2.C_1P24, Lit(1 << (63 - AUTO_B * AUTO_L)))
        e("s_mov_b32", T2.IDX, Lit(int(os.environ.get("TR_START", 0))))
        for m in range(16): self.ds_read("ds_read_b64", V(OWN + 2 * m, 2), L8T, offset=512 * m)
        self.lg_wait_all()

    def step(self):
        e = self.e
        self.p.label("L_tr%=")
        self.c("---- t = N/2^idx + 1, t^-1 mod 2N, gather addresses relative to tile A")
        e("s_lshr_b32", T2.T, Lit(N), T2.IDX)
        e("s_add_u32", T2.T, T2.T, Lit(1))                      # t
        e("s_mov_b32", T2.TINV, T2.T)
        for _ in range(2):
            e("s_mul_i32", T2.Tb, T2.T, T2.TINV)
            e("s_sub_u32", T2.Tb, Lit(2), T2.Tb)
            e("s_mul_i32", T2.TINV, T2.TINV, T2.Tb)
        e("s_lshl_b32", T2.C64, T2.TINV, Lit(6))
        e("s_and_b32", T2.C64, T2.C64, Lit(2 * N - 1))
        e("v_mul_lo_u32", T0.lo(), LANE, T2.TINV)
        e("v_and_b32", T0.lo(), Lit(2 * N - 1), T0.lo())
        for m in range(16):
            if m:
                e("v_add_u32", T0.lo(), T2.C64, T0.lo())
                e("v_and_b32", T0.lo(), Lit(2 * N - 1), T0.lo())
            e("v_bfe_i32", GS[m], T0.lo(), Lit(LOGN), Lit(1))
            e("v_and_b32", T0.hi(), Lit(N - 1), T0.lo())
            e("v_lshl_add_u32", GA[m], T0.hi(), Lit(3), T2.TILEA)
        # key base of this step and this wave's split: auto_f + w * split_stride + idx * (2 * 3 * 3 * 512 * 16)
        e("s_mul_i32", T2.T, T2.IDX, Lit(2 * AUTO_L * KP * 512 * 16))
        e("s_mul_i32", T2.Tb, T2.W, T2.SPLIT)
        e("s_add_u32", T2.T, T2.T, T2.Tb)
        e("s_add_u32", T2.PB.lo(), T2.AUTO.lo(), T2.T)
        e("s_addc_u32", T2.PB.hi(), T2.AUTO.hi(), Lit(0))
        # ---- stage this wave's mask polynomial (natural order) for the automorphism's gather
        self.c("---- stage the mask polynomial in the tile; barrier; gather both polynomials' images p(X^t)")
        for m in range(16): self.ds_write("ds_write_b64", L8T, V(OWN + 2 * m, 2), offset=512 * m)
        self.barrier()
        X0 = [V(F[1] + 2 * j, 2) for j in range(16)]
        X1 = [V(F[2] + 2 * j, 2) for j in range(16)]
        for m in range(8):
            self.ds_read("ds_read_b64", X0[m], GA[m], offset=0)
            self.ds_read("ds_read_b64", X0[m + 8], GA[m + 8], offset=0)
        for m in range(8):
            self.ds_read("ds_read_b64", X1[m], GA[m], offset=TILE_BYTES)
            self.ds_read("ds_read_b64", X1[m + 8], GA[m + 8], offset=TILE_BYTES)

        # ---- body polynomial += its automorphism image (wave B; every word is read before the first write)
        def body_auto():
            self.c("---- (wave B) body polynomial += its automorphism image")
            BO = [V(D + 2 * j, 2) for j in range(16)]
            BG = [V(X + 2 * j, 2) for j in range(16)]
            for j in range(16): self.ds_read("ds_read_b64", BO[j], L8A, offset=BODY_OFF + 512 * j)
            for j in range(16): self.ds_read("ds_read_b64", BG[j], GA[j], offset=BODY_OFF)

            def body(j, k):
                lo, hi = BG[j].lo(), BG[j].hi()
                e("v_xor_b32", lo, lo, GS[j]); e("v_xor_b32", hi, hi, GS[j])
                e("v_sub_co_u32", lo, CARRY[k], lo, GS[j]); e("v_subb_co_u32", hi, CARRY[k], hi, GS[j], CARRY[k])
                e("v_add_co_u32", lo, CARRY[k], BO[j].lo(), lo); e("v_addc_co_u32", hi, CARRY[k], BO[j].hi(), hi, CARRY[k])
                self.ds_write("ds_write_b64", L8A, BG[j], offset=BODY_OFF + 512 * j)
            for j0 in range(0, 16, 4):
                self.merge([self.chain(body, j0 + k, k) for k in range(4)])
        self.only(1, body_auto)          # (drains the gathers above on both waves)

        # ---- decomposition of both polynomials; this wave keeps its three digit polynomials, packed
        self.c("---- 3-level decomposition of both images; keep digit polynomials (A: 0, 2, 4; B: 1, 3, 5), packed")
        def decomp(j, k):
            tb = X + 12 * k                          # 12 temporaries per chain
            r = [[V(tb + 3 * q + l) for l in range(3)] for q in range(2)]
            u, car, s = V(tb + 6), V(tb + 7), [V(tb + 8), V(tb + 9)]
            self.digits_int(X0[j], GS[j], r[0], u, car, CARRY[k])
            self.digits_int(X1[j], GS[j], r[1], u, car, CARRY[k])
            e("v_cndmask_b32", s[0], r[0][0], r[0][1], T2.SELB)             # round 0: A digit polynomial 0 = (q0, level 0), B 1 = (q0, level 1)
            e("v_cndmask_b32", s[1], r[0][2], r[1][0], T2.SELB)             # round 1: A 2 = (q0, level 2), B 3 = (q1, level 0)
            e("v_cndmask_b32", V(PD + 2 * j + 1), r[1][1], r[1][2], T2.SELB)   # round 2: A 4 = (q1, level 1), B 5 = (q1, level 2)
            e("v_and_b32", s[0], Lit(0xFFFF), s[0])
            e("v_lshl_or_b32", V(PD + 2 * j), s[1], Lit(16), s[0])
        for j0 in range(0, 16, 4):
            self.merge([self.chain(decomp, j0 + k, k) for k in range(4)])
        self.barrier()                               # both waves have gathered: the tiles are free for the transforms

        xq = [quad(X, r) for r in range(12)]
        ring = [quad(X, r) for r in [9, 10, 11, 0, 1, 2, 3, 4, 5, 6, 7, 8]]
        NG = 12
        Fq = [[quad(F[role], r) for r in range(8)] for role in range(3)]
        land = [quad(D, r) for r in range(8)]
        for k in range(3):
            ge, go = 2 * k, 2 * k + 1
            self.c("---- round %d: transform this wave's digit polynomial (A: %d, B: %d), publish, multiply-accumulate %d then %d" % (k, ge, go, ge, go))
            # key pointers of the round: (digit polynomial, role) -> row of the split key; role 0 = own column, 1 = partner's, 2 = body
            for gi, g in enumerate((ge, go)):
                for role, off in ((0, T2.OFFO), (1, T2.OFFT), (2, Lit(2 * 8192))):
                    kp = T2.KP6[gi * 3 + role]
                    e("s_add_u32", kp.lo(), T2.PB.lo(), Lit(self.key_off(3 * g)))
                    e("s_addc_u32", kp.hi(), T2.PB.hi(), Lit(0))
                    e("s_add_u32", kp.lo(), kp.lo(), off)
                    e("s_addc_u32", kp.hi(), kp.hi(), Lit(0))
            stream = [(gi, r, role) for gi in range(2) for r in range(8) for role in range(3)]      # 48 key quads

            def issue(i):
                gi, r, role = stream[i]
                if "K" in ABL2: return
                self.gload("global_load_dwordx4", ring[i % NG], L16 if r < 4 else OFF1, T2.KP6[gi * 3 + role], offset=(r & 3) * 1024)
            for i in range(3): issue(i)                              # X9..11 are free during the transform
            # unpack the digit polynomial of this round
            for m in range(8):
                for half in range(2):
                    j = m + 8 * half
                    dst = V(D + 4 * m + 2 * half, 2)
                    t = T0.lo() if half == 0 else T0.hi()
                    if k == 0:
                        e("v_bfe_i32", t, V(PD + 2 * j), Lit(0), Lit(16)); e("v_cvt_f64_i32", dst, t)
                    elif k == 1:
                        e("v_ashrrev_i32", t, Lit(16), V(PD + 2 * j)); e("v_cvt_f64_i32", dst, t)
                    else:
                        e("v_cvt_f64_i32", dst, V(PD + 2 * j + 1))
            v, spare = self.fwd1([quad(D, r) for r in range(8)], xq, before_pass3=lambda: [issue(i) for i in range(3, 11)])
            if "P" not in ABL2:
                for r in range(8): self.ds_write("ds_write_b128", L16T, v[r], offset=1024 * r)
            self.barrier()
            issue(11)
            # the published transforms (tile A: the even one, tile B: the odd one) through the landing ring D
            dstream = [(gi, r) for gi in range(2) for r in range(8)]

            def fetch(i):
                gi, r = dstream[i]
                if "P" in ABL2: return
                self.ds_read("ds_read_b128", land[i % 8], L16A, offset=gi * TILE_BYTES + 1024 * r)
            for i in range(8): fetch(i)

            def mac(i):
                gi, r, role = stream[i]
                g = (ge, go)[gi]
                kq, d, acc = ring[i % NG], land[(gi * 8 + r) % 8], Fq[role][r]
                re0 = Lit(0, f64=0.0) if g == 0 else qre(acc)
                im0 = Lit(0, f64=0.0) if g == 0 else qim(acc)
                self.e("v_fma_f64", qre(acc), qre(d), qre(kq), re0)
                self.e("v_fma_f64", qim(acc), qre(d), qim(kq), im0)
                self.e("v_fma_f64", qre(acc), -qim(d), qim(kq), qre(acc))
                self.e("v_fma_f64", qim(acc), qim(d), qre(kq), qim(acc))
            for i0 in range(0, len(stream), 3):                      # one published quad against its three key quads
                self.merge([self.chain(mac, i0), self.chain(mac, i0 + 1), self.chain(mac, i0 + 2)])
                for i in (i0, i0 + 1, i0 + 2):
                    if i + NG < len(stream): issue(i + NG)
                di = i0 // 3
                if di + 8 < len(dstream): fetch(di + 8)
            self.barrier()                           # both waves have read both tiles
        # ---- three inverse transforms together
        polys, spare = self.inv3(Fq, xq)
        # ---- updates.  A (low split): full rounding; B (high split): (round(x) mod 2^32) << 41, i.e. only high dwords change.
        # own column: into this wave's polynomial; partner's column: as 64-bit increments through the tile; body: in LDS,
        # A in front of the barrier, B behind it (integer additions commute)
        tmp4 = [(V(X + 4 * k_, 2), V(X + 4 * k_ + 2, 2)) for k_ in range(4)]
        INC = [V(D + 2 * j, 2) for j in range(16)]
        CUR = [V(X + 16 + 2 * j, 2) for j in range(16)]              # X4..11: body words (wave A)

        def lo_updates():
            self.c("---- (wave A) own column and increments of the partner's column: round(inverse); body += round(inverse)")
            def upd_own(m, half, k_):
                j = m + 8 * half
                x = V(polys[0][m].i + 2 * half, 2)
                al, ah = V(OWN + 2 * j), V(OWN + 2 * j + 1)
                self.torus_add(x, al, ah, al, ah, tmp4[k_][0], tmp4[k_][1], CARRY[k_])
            def upd_oth(m, half, k_):
                j = m + 8 * half
                x = V(polys[1][m].i + 2 * half, 2)
                self.torus_add(x, Lit(0), Lit(0), INC[j].lo(), INC[j].hi(), tmp4[k_][0], tmp4[k_][1], CARRY[k_])
                self.ds_write("ds_write_b64", L8T, INC[j], offset=512 * j)
            for fn in (upd_own, upd_oth):
                chains = []
                for m in range(8):
                    for half in range(2):
                        chains.append(self.chain(fn, m, half, len(chains)))
                        if len(chains) == 4:
                            self.merge(chains); chains = []
            for j in range(16): self.ds_read("ds_read_b64", CUR[j], L8A, offset=BODY_OFF + 512 * j)
            def upd_body(m, half, k_):
                j = m + 8 * half
                x = V(polys[2][m].i + 2 * half, 2)
                self.torus_add(x, CUR[j].lo(), CUR[j].hi(), CUR[j].lo(), CUR[j].hi(), tmp4[k_][0], tmp4[k_][1], CARRY[k_])
                self.ds_write("ds_write_b64", L8A, CUR[j], offset=BODY_OFF + 512 * j)
            chains = []
            for m in range(8):
                for half in range(2):
                    chains.append(self.chain(upd_body, m, half, len(chains)))
                    if len(chains) == 4:
                        self.merge(chains); chains = []

        def hi_part(x, t):
            """x (f64 pair, destroyed) -> x.lo = round(x) mod 2^32 (low dword of the rounded value, as trgen.py)"""
            e("v_mul_f64", t, x, T2.C_2M32)
            e("v_rndne_f64", t, t)
            e("v_fma_f64", x, t, T2.C_M2P32, x)
            e("v_add_f64", x, x, SG.C_52)

        def hi_updates():
            self.c("---- (wave B) own column and increments of the partner's column: (round(inverse) mod 2^32) << 41; body column prepared")
            t4 = [V(X + 2 * k_, 2) for k_ in range(4)]
            def upd_own(m, half, k_):
                j = m + 8 * half
                x = V(polys[0][m].i + 2 * half, 2)
                hi_part(x, t4[k_])
                e("v_lshl_add_u32", V(OWN + 2 * j + 1), x.lo(), Lit(AUTO_SPLIT - 32), V(OWN + 2 * j + 1))
            def upd_oth(m, half, k_):
                j = m + 8 * half
                x = V(polys[1][m].i + 2 * half, 2)
                hi_part(x, t4[k_])
                e("v_mov_b32", INC[j].lo(), Lit(0))
                e("v_lshlrev_b32", INC[j].hi(), Lit(AUTO_SPLIT - 32), x.lo())
                self.ds_write("ds_write_b64", L8T, INC[j], offset=512 * j)
            def prep_body(m, half, k_):
                x = V(polys[2][m].i + 2 * half, 2)
                hi_part(x, t4[k_])                    # x.lo: what is added (<< 9) to the body's high dword behind the barrier
            for fn in (upd_own, upd_oth, prep_body):
                chains = []
                for m in range(8):
                    for half in range(2):
                        chains.append(self.chain(fn, m, half, len(chains)))
                        if len(chains) == 4:
                            self.merge(chains); chains = []
        self.only(0, lo_updates)
        self.only(1, hi_updates)
        self.barrier()
        self.c("---- this wave's polynomial += the partner's increments")
        for j in range(16): self.ds_read("ds_read_b64", INC[j], L8P, offset=512 * j)
        for j in range(16):
            k_ = j % 4
            e("v_add_co_u32", V(OWN + 2 * j), CARRY[k_], V(OWN + 2 * j), INC[j].lo())
            e("v_addc_co_u32", V(OWN + 2 * j + 1), CARRY[k_], V(OWN + 2 * j + 1), INC[j].hi(), CARRY[k_])

        def hi_body():
            self.c("---- (wave B) body high dwords += (round(inverse) mod 2^32) << 9")
            CURH = [V(X + 16 + j) for j in range(16)]
            for j in range(16): self.ds_read("ds_read_b32", CURH[j], L8A, offset=BODY_OFF + 512 * j + 4)
            for m in range(8):
                for half in range(2):
                    j = m + 8 * half
                    x = V(polys[2][m].i + 2 * half, 2)
                    e("v_lshl_add_u32", CURH[j], x.lo(), Lit(AUTO_SPLIT - 32), CURH[j])
                    self.ds_write("ds_write_b32", L8A, CURH[j], offset=BODY_OFF + 512 * j + 4)
        self.only(1, hi_body)
        self.barrier()                               # the tiles and the body are consistent: next step (or the end)
        # ---- loop control
        e("s_add_u32", T2.IDX, T2.IDX, Lit(1))
        e("s_cmp_lt_u32", T2.IDX, Lit(self.steps))
        assert self.vmq == [] and self.lgq == [], (self.vmq, self.lgq)
        e("s_cbranch_scc1", "L_tr%=")

    def epilogue(self):
        self.c("---- epilogue: this wave's mask polynomial back to its tile (natural order); the body is in LDS")
        for m in range(16): self.ds_write("ds_write_b64", L8T, V(OWN + 2 * m, 2), offset=512 * m)
        self.lg_wait_all()

    def build(self):
        self.prologue()
        self.step()
        self.epilogue()
        return self.p


import brgen3
Trace2Gen.fwd1 = brgen3.CoopGen.fwd1


def render_inc(prog):
    """operands: %0 auto_f (64), %1 split stride in bytes, %2 wave of the pair (0 = A, 1 = B), %3 this wave's tile, %4 tile A"""
    head = ["s_mov_b64 %s, %%0" % T2.AUTO, "s_mov_b32 %s, %%1" % T2.SPLIT, "s_mov_b32 %s, %%2" % T2.W,
            "s_mov_b32 %s, %%3" % T2.TILE, "s_mov_b32 %s, %%4" % T2.TILEA]
    lines = ["  " + h for h in head] + prog.render()
    out = ["// generated by tools/asmgen/trgen2.py -- do not edit"]
    for l in lines: out.append('"%s\\n"' % l.replace('"', '\\"'))
    return "\n".join(out) + "\n"


def render_clobbers():
    regs = ["v%d" % i for i in range(256)] + ["s%d" % i for i in range(34, 96)] + ["vcc", "scc", "memory"]
    return "// generated by tools/asmgen/trgen2.py -- do not edit\n" + ", ".join('"%s"' % r for r in regs) + "\n"


if __name__ == "__main__":
    prog = Trace2Gen(steps=int(os.environ.get("TR_STEPS", LOGN))).build()
    if len(sys.argv) > 2 and sys.argv[1] == "--out":
        open(os.path.join(sys.argv[2], "tx_trace2_asm.inc"), "w").write(render_inc(prog))
        open(os.path.join(sys.argv[2], "tx_trace2_asm_clobbers.inc"), "w").write(render_clobbers())
    else:
        c = prog.count()
        for k in sorted(c, key=lambda k: -c[k]): print("%6d %s" % (c[k], k))
        print("%6d total" % sum(c.values()))
