"""Generator of the homomorphic-trace loop (a7: 10 x automorphism + GLWE key switch with the 41/23-bit split key) as
gfx950 assembly: the body of trace_lds() of csrc/tx_kernels.hpp, one wave per SIMD (a trace step keeps six digit
transforms live across both passes over the key: 192 registers next to the 24 KiB GLWE -- one unit per SIMD is all
that fits), four waves per workgroup sharing every key polynomial through the LDS ring.

What the explicit schedule buys over the hipcc version (whose key pass is 33 % of the step: one register set, so
every slot read is exposed behind its barrier): the six digit transforms are parked in accumulation registers, which
leaves room for TWO key register sets -- the LDS read of key polynomial j+1 is in flight while polynomial j is
multiplied -- and for three-way interleaved transforms in both directions; the high-split pass converts with the
4-operation form (only the low dword of the rounded value survives the << 41) and touches only the high dwords of
the GLWE in LDS.  Arithmetic, operation order and results are those of trace_lds() / oracle txo_trace(): bit-identical.

    python3 tools/asmgen/trgen.py --out temp-fhe-transciphering_amd/csrc      (tx_trace_asm.inc + clobbers)

LDS image (k_extract_trace4 / k_trace4 in tx_br2.hpp): compact twiddles at 0 (9 088 B), then per wave 32 896 B:
8 320-byte transpose tile of pitch 65 (= ring slot of that wave) + the GLWE, 3 x 1024 torus words in natural order.

Register map (VGPR): 0..15 as brgen.py (4: DMA source offset, 8: top-level tie mask, 9: gather index of m = 0),
16..19 ring slot read addresses, 20..31 temporaries, P0 32..63 / P1 64..95 / P2 96..127 (digit polynomials of three
levels -> their transforms; the three Fourier accumulators in the key pass), DC 128..159 (the digit transform being
multiplied), K0 160..191 / K1 192..223 (key register sets; twiddles + spare in the transforms), 224..239 gather
addresses, 240..255 gather sign masks.  AGPR 0..191: the six digit transforms [q*3 + level][32].
"""
import sys
import os
from gcn import V, S, A, Lit, VCC, Prog, M0
import brgen
from brgen import Gen, SG, quad, qre, qim, T0, T1, LANE, L8T, L16T, L16, TWBA, XHW, XLW, CARRY, f64bits

N, LOGN, KP = 1024, 10, 3
AUTO_B, AUTO_L, AUTO_SPLIT = 13, 3, 41
DMAOFF, TIE, IDX0, DMAOFF2 = V(4), V(8), V(9), V(10)
SL = [V(16 + s) for s in range(4)]
TMP = [V(20 + k) for k in range(12)]
P = [32, 64, 96]
DC = 128
K = [160, 192]
GA = [V(224 + m) for m in range(16)]
GS = [V(240 + m) for m in range(16)]
WAVE_LDS = brgen.BODY + 3 * 8192           # per wave: padded tile (8 320 B, = its ring slot) + the GLWE
ABL = os.environ.get("TR_ABL", "")        # timing-only ablations of the key pass (D no DMA, B no barriers, R no slot reads, A no AGPR reads, M no MAC): outputs invalid
GL = brgen.BODY                            # the GLWE follows the tile


class TS:      # scalar registers of this loop (constants at 48..58 and carries at 88..95 as brgen)
    AUTO = S(34, 2); SPLIT = S(36); WAVE = S(38); SLOT0 = S(39); IDX = S(41); TILE = SG.TILE; TINV = S(43); C64 = S(44)
    T = S(45); T2 = S(46)
    TIEMODE = S(46)       # operand, read by the prologue only (= T2, free until then): 1 = balanced top-level tie, 0 = tfhe-rs' rule
    KP_ = S(60, 2); PB = S(62, 2); GLB = S(40); C_M2P32 = S(64, 2); M0A = S(66); C_1P24 = S(67); C_2M32 = S(68, 2); M0SAVE = S(47)


class TraceGen(Gen):
    CC, CS = TS.KP_, S(44, 2)         # twist constants: the DMA pointer and C64:T are dead inside a transform
    DIG_B, DIG_L = AUTO_B, AUTO_L     # the decomposition this loop uses (prologue constants)

    def __init__(self, steps=LOGN, stop_after=None, stamps=False):
        super().__init__(stamps=stamps, prio="")
        self.steps = steps                # debugging: fewer trace steps
        # TR_XHI=regs: exchange hi of every transform in registers (Gen.xchg_hi_regs) instead of through the tile -- a lone wave
        # pays 53 cycles per ds_write_b128 when the four waves of the workgroup store together
        self.xhi_regs = os.environ.get("TR_XHI", "lds") == "regs"
        self.stop_after = stop_after      # debugging: leave the step after phase "G" (digits parked), "B" (body), "M0" ...

    # ------------------------------------------------------------------ transforms of a wave that is alone on its SIMD
    # A ds_write_b128 holds the CU's LDS store path for 13 cycles, and a wave waits at a store until the path takes it:
    # with the four waves of the workgroup in the same pass, eight stores in a row cost a wave ~420 cycles (stamps
    # inside the transforms, tools/substamps_trace.sh: stores 53 cycles each, loads 16) in which it issues nothing.
    # Here the stores of polynomial p's exchange are spread through the butterflies of polynomial p + 1 (the next
    # pass's first polynomial after the last one), so each store finds the path as free as four waves leave it; the
    # loads follow the last store.  Every polynomial has its own spare quad.  Same arithmetic, same order per polynomial.
    PIPE = os.environ.get("TR_PIPE", "1") != "0"

    def _pipeline(self, items):
        """items: (valu function, store function or None, load function or None) in program order"""
        pend = None
        for valu, store, load in items:
            ch = self.chain(valu)
            if pend:
                self.merge_spaced(ch, pend[0])
                pend[1]()                   # the loads in a row behind the last store (spreading them too measured the same)
            else:
                self.merge([ch])
            pend = (self.chain(store), load) if store else None
        assert pend is None

    def inv3(self, polys, xq):
        if not self.PIPE: return super().inv3(polys, xq)
        tw, spares = xq[:8], [xq[8], xq[9], xq[10]]
        home_lo = [min(q.i for q in polys[p]) for p in range(3)]
        self.c("inverse x3 (stores of an exchange spread through the next polynomial's butterflies)")
        self.substamp("inv3", 5)
        self.load_tw(tw, 0, 7, compact=True)

        def reread(p):
            def f():
                free = sorted(polys[p] + [spares[p]], key=lambda q: q.i)
                polys[p], spares[p] = free[:8], free[8]
                self.xchg_read(polys[p], swz=False)
            return f

        def valu(pas, p):
            def f():
                if pas == 2:
                    self.cmul_all([(polys[p][r], tw[r - 1]) for r in range(1, 8)], spares[p], conj=True)
                    if p == 2: self.load_tw(tw, 0, 8)               # the pass-A rows replace the pass-B rows after their last use
                if pas == 1: self.cmul_all([(polys[p][r], tw[r]) for r in range(8)], spares[p], conj=True)
                polys[p], spares[p] = self.dft8(polys[p], spares[p], -1, home=(home_lo[p], home_lo[p] + 32) if pas == 1 else None)
                if pas == 1: polys[p], spares[p] = self.home(polys[p], spares[p], home_lo[p], home_lo[p] + 32)
                if pas == 2 and self.xhi_regs: polys[p], spares[p] = self.xchg_hi_regs(polys[p], spares[p])
            return f
        items = []
        for pas in (3, 2, 1):
            for p in range(3):
                st = None
                if pas == 3: st = (lambda p=p: self.xchg_lo_write(polys[p]))
                if pas == 2 and not self.xhi_regs: st = (lambda p=p: self.xchg_hi_write(polys[p]))
                items.append((valu(pas, p), st, reread(p) if st else None))
        self._pipeline(items)
        self.substamp("inv3", 2)
        self.twist_consts(polys, spares[0], conj=True)
        self.substamp("inv3", 3)
        return polys, spares[0]

    def fwd3(self, polys, xq, before_pass3=None):
        if not self.PIPE: return super().fwd3(polys, xq, before_pass3)
        assert before_pass3 is None
        tw, spares = xq[:8], [xq[8], xq[9], xq[10]]
        home_lo = [min(q.i for q in polys[p]) for p in range(3)]
        self.c("forward x3 (stores of an exchange spread through the next polynomial's butterflies)")
        self.substamp("fwd3", 5)
        self.load_tw(tw, 0, 8)
        self.twist_consts(polys, None, conj=False)
        self.substamp("fwd3", 0)

        def reread(p):
            def f():
                free = sorted(polys[p] + [spares[p]], key=lambda q: q.i)
                polys[p], spares[p] = free[:8], free[8]
                self.xchg_read(polys[p], swz=False)
            return f

        def valu(pas, p):
            def f():
                polys[p], spares[p] = self.dft8(polys[p], spares[p], +1, home=(home_lo[p], home_lo[p] + 32) if pas == 3 else None)
                if pas == 1:
                    self.cmul_all([(polys[p][r], tw[r]) for r in range(8)])
                    if p == 2: self.load_tw(tw, 0, 7, compact=True)
                    if self.xhi_regs: polys[p], spares[p] = self.xchg_hi_regs(polys[p], spares[p])
                if pas == 2: self.cmul_all([(polys[p][r], tw[r - 1]) for r in range(1, 8)])
                if pas == 3: polys[p], spares[p] = self.home(polys[p], spares[p], home_lo[p], home_lo[p] + 32)
            return f
        items = []
        for pas in (1, 2, 3):
            for p in range(3):
                st = None
                if pas == 1 and not self.xhi_regs: st = (lambda p=p: self.xchg_hi_write(polys[p]))
                if pas == 2: st = (lambda p=p: self.xchg_lo_write(polys[p]))
                items.append((valu(pas, p), st, reread(p) if st else None))
        self._pipeline(items)
        self.substamp("fwd3", 3)
        return polys, spares[0]

    # ------------------------------------------------------------------ LDS-DMA ring (see KeyRing in tx_kernels.hpp)
    NSLOT = int(os.environ.get("TR_SLOTS", 4))       # ring mode: slots 0..3 are the waves' tiles, further ones follow the last wave's GLWE

    @staticmethod
    def slot_off(s_):
        return s_ * WAVE_LDS if s_ < 4 else 4 * WAVE_LDS + (s_ - 4) * 8192

    @staticmethod
    def key_off(j):
        """byte offset of key polynomial j of a pass from the pass base: digit polynomial j // 3 = (mask polynomial q, level
        i finest first), column j % 3; the key stores level 1 (coarsest) first"""
        q, i, c = j // 9, (j // 3) % 3, j % 3
        return ((q * AUTO_L + (AUTO_L - 1 - i)) * KP + c) * 8192

    def dma(self, j, slot=None):
        """this wave's quarter of key polynomial j of the current pass -> ring slot (default j % NSLOT), two 1 KiB loads"""
        off = self.key_off(j)
        e = self.e
        if slot is None: slot = j % self.NSLOT
        if "D" in ABL: return
        e("s_add_u32", TS.KP_.lo(), TS.PB.lo(), Lit(off))
        e("s_addc_u32", TS.KP_.hi(), TS.PB.hi(), Lit(0))
        # destination: slot0 + (j % 4) * 32 KiB + wave * 2 KiB (wave-uniform, in M0)
        e("s_add_u32", M0, TS.M0A, Lit(self.slot_off(slot)))
        e("s_nop", Lit(0))
        e("global_load_lds_dwordx4", DMAOFF, TS.KP_, offset=0)
        self.vmq.append({1000 + j})
        e("s_add_u32", M0, TS.M0A, Lit(self.slot_off(slot) + 1024))
        e("s_nop", Lit(0))
        e("global_load_lds_dwordx4", DMAOFF2, TS.KP_, offset=0)      # no instruction offset on an LDS-DMA load: it would also move the LDS address
        self.vmq.append({1000 + j})

    SKEW = int(os.environ.get("TR_SKEW", 2))     # s_sleep units (64 cycles) of phase offset per wave index

    def skew(self):
        """Wave w waits w * SKEW * 64 cycles.  The four waves of a workgroup leave a barrier together and run the same
        code: without an offset all four reach every tile exchange at the same moment and queue on the CU's LDS store
        path (a ds_write_b128 holds it for 13 cycles, 26 when only one SIMD of a half is storing), each blocked behind
        the others' stores instead of running its butterflies."""
        if not self.SKEW: return
        e = self.e
        self.nlab += 1
        lab = "L_sk%d_%%=" % self.nlab
        for w in range(1, 4):
            e("s_cmp_lt_u32", TS.WAVE, Lit(w))
            e("s_cbranch_scc1", lab)
            e("s_sleep", Lit(self.SKEW))
        self.p.label(lab)

    # "global" (default since round 4): every wave streams the key polynomials into registers itself, three register sets;
    # "ring": the workgroup's shared LDS ring (LDS-DMA fill, one s_barrier per polynomial).  Rounds 2-3 measured the two equal
    # (14.1 ms standalone, any ring depth from 4 to 6 slots): streaming asks the CU's vector-memory path for 32 KiB per polynomial
    # time, four times what the ring does, the ring pays its per-polynomial barrier and the LDS-DMA's landing rate.  With the
    # rest of the loop faster (spread exchange stores, 64-bit adds) the streamed form is now ahead: 12.70 against 13.15 ms per
    # launch in the pipeline, same box, alternating (profiles/r4ae_*); timing-only ablations of the ring form there: without the
    # DMA 12.53, without the slot reads 12.25, without the barriers 11.95, without DMA and reads 10.93 ms.  (Alternating the two
    # - even polynomials through the ring, odd ones streamed - was slower than either: 14.7 ms.)
    KEYS = os.environ.get("TR_KEYS", "global")

    def key_load(self, j, dst):
        """key polynomial j of the current pass -> the eight quads dst (F-layout rows of 1 KiB, as the bootstrap key in brgen)"""
        off = self.key_off(j)
        e = self.e
        e("s_add_u32", TS.KP_.lo(), TS.PB.lo(), Lit(off))
        e("s_addc_u32", TS.KP_.hi(), TS.PB.hi(), Lit(0))
        for r in range(8):
            self.gload("global_load_dwordx4", dst[r], L16 if r < 4 else DMAOFF, TS.KP_, offset=(r & 3) * 1024)

    def ring_barrier(self, need=None):
        """this wave's LDS traffic done and (need: key polynomial) its DMA landed, then the workgroup barrier"""
        if need is not None:
            k = self._wait(self.vmq, {1000 + need}, 63)
            if k is not None: self.p.emit("s_waitcnt", vm=k)
        self.lg_wait_all()
        if "B" in ABL and need is not None: return
        self.p.emit("s_barrier")

    # ------------------------------------------------------------------ decomposition (Decomp<13, 3>, finest level first)
    def digits3(self, x, sgn, outs, t, cy):
        """x: rotated torus word (pair, destroyed); outs: three f64 pairs (levels finest..top); t: 3 temporaries"""
        self.digits_bl(x, sgn, outs, t, cy, AUTO_B, AUTO_L)

    def digits_bl(self, x, sgn, outs, t, cy, B, L):
        """signed decomposition Decomp<B, L> of tx_kernels.hpp (tfhe-rs SignedDecomposer), finest level first: x (pair,
        destroyed; negated first where sgn is all ones) -> L f64 digits.  TS.C_1P24 holds the rounding constant
        2^(63 - B L), TIE the top-level tie mask (lane & 1) << (B - 1).  B L > 32: the state spans both dwords at first."""
        lo, hi = x.lo(), x.hi()
        res, u, car = t
        e = self.e
        total = B * L
        assert 32 < total < 64
        if sgn is not None:
            e("v_xor_b32", lo, lo, sgn)
            e("v_xor_b32", hi, hi, sgn)
            e("v_sub_co_u32", lo, cy, lo, sgn)
            e("v_subb_co_u32", hi, cy, hi, sgn, cy)
        # init: ((x >> (63 - B L)) + 1) >> 1 & (2^(B L) - 1)  ==  (x + 2^(63 - B L)) >> (64 - B L) (the carry out of bit 63 is dropped either way)
        e("v_add_co_u32", lo, cy, TS.C_1P24, lo)             # VOP3 (own carry register): no literal, the constant sits in an SGPR
        e("v_addc_co_u32", hi, cy, Lit(0), hi, cy)
        e("v_alignbit_b32", lo, hi, lo, Lit(64 - total))    # state bits 0..31 (hi >> (64 - B L) holds the rest)
        e("v_lshrrev_b32", hi, Lit(64 - total), hi)
        for lev in range(L):
            e("v_and_b32", res, Lit((1 << B) - 1), lo)
            if lev == 0: e("v_alignbit_b32", lo, hi, lo, Lit(B))       # B L - B <= 32 bits: fits one dword from here on
            else: e("v_lshrrev_b32", lo, Lit(B), lo)
            e("v_add_u32", u, Lit(-1), res)
            e("v_or_b32", u, u, lo)
            if lev == L - 1: e("v_or_b32", u, u, TIE)
            e("v_and_b32", u, u, res)
            e("v_bfe_i32", car, u, Lit(B - 1), Lit(1))          # -carry (0 or -1)
            if lev < L - 1: e("v_sub_u32", lo, lo, car)         # state += carry
            e("v_lshl_add_u32", res, car, Lit(B), res)          # res -= carry << B
            e("v_cvt_f64_i32", outs[lev], res)

    # ------------------------------------------------------------------ the kernel
    def prologue(self):
        e = self.e
        self.c("---- prologue (trace): addresses, constants")
        self.p.emit("s_waitcnt", vm=0, lgkm=0)
        e("v_mbcnt_lo_u32_b32", LANE, Lit(-1), Lit(0))
        e("v_mbcnt_hi_u32_b32", LANE, Lit(-1), LANE)
        e("v_lshl_add_u32", L8T, LANE, Lit(3), TS.TILE)
        e("v_lshl_add_u32", L16T, LANE, Lit(4), TS.TILE)
        e("v_lshlrev_b32", L16, Lit(4), LANE)
        e("s_lshl_b32", TS.T, TS.WAVE, Lit(11))                 # wave * 2048
        if self.KEYS == "global":
            e("v_add_u32", DMAOFF, Lit(4096), L16)              # rows 4..7 of a key polynomial
        else:
            e("v_add_u32", DMAOFF, TS.T, L16)                   # (wave * 128 + lane) * 16
            e("v_add_u32", DMAOFF2, Lit(1024), DMAOFF)          # second half: 64 complex further
        e("s_add_u32", TS.M0A, TS.SLOT0, TS.T)                  # slot 0 + wave * 2 KiB
        e("v_and_b32", T0.lo(), Lit(7), LANE)
        e("v_lshlrev_b32", TWBA, Lit(4), T0.lo())
        e("v_add_u32", TWBA, Lit(brgen.TWB), TWBA)
        e("v_lshrrev_b32", T0.hi(), Lit(3), LANE)
        e("v_lshl_add_u32", T1.lo(), T0.hi(), Lit(6), T0.hi())  # a*65 (tile pitch: brgen.PITCH)
        e("v_add_u32", T0.hi(), T1.lo(), T0.lo())               # a*65 + b
        e("v_lshl_add_u32", XHW, T0.hi(), Lit(4), TS.TILE)
        e("v_and_b32", T1.lo(), Lit(0x38), LANE)
        e("v_lshl_add_u32", T1.hi(), T0.lo(), Lit(6), T0.lo())  # b*65
        e("v_add_u32", T1.lo(), T1.hi(), T1.lo())               # b*65 + g8
        e("v_lshl_add_u32", XLW, T1.lo(), Lit(4), TS.TILE)
        self.tie_lane(T0.lo(), TS.TIEMODE)                      # lane & 1 (balanced rule) or 0 (tfhe-rs' rule)
        e("v_lshlrev_b32", TIE, Lit(self.DIG_B - 1), T0.lo())   # tie_mask: (lane & 1) << (B - 1)
        for s_ in range(4):
            e("s_add_u32", TS.T, TS.SLOT0, Lit(s_ * WAVE_LDS))
            e("v_add_u32", SL[s_], TS.T, L16)
        e("s_add_u32", TS.GLB, TS.TILE, Lit(GL))
        for name, val in (("C_2M64", 2.0 ** -64), ("C_M2P64", -(2.0 ** 64)), ("C_84", 1.5 * 2.0 ** 84), ("C_52", 1.5 * 2.0 ** 52),
                          ("C_SQH", 0.70710678118654752440)):
            s_, b = getattr(SG, name), f64bits(val)
            e("s_mov_b32", s_.lo(), Lit(b & 0xFFFFFFFF)); e("s_mov_b32", s_.hi(), Lit(b >> 32))
        for s_, val in ((TS.C_2M32, 2.0 ** -32), (TS.C_M2P32, -(2.0 ** 32))):
            b = f64bits(val)
            e("s_mov_b32", s_.lo(), Lit(b & 0xFFFFFFFF)); e("s_mov_b32", s_.hi(), Lit(b >> 32))
        e("s_mov_b32", SG.C_M4338, Lit((-0x43380000) & 0xFFFFFFFF))
        e("s_mov_b32", TS.C_1P24, Lit(1 << (63 - self.DIG_B * self.DIG_L)))      # rounding constant of the decomposition
        e("s_mov_b32", TS.IDX, Lit(int(os.environ.get("TR_START", 0))))
        if self.stamps:
            for k in range(brgen.NSTAMP):
                e("s_mov_b32", brgen.ST_ACC[k].lo(), Lit(0)); e("s_mov_b32", brgen.ST_ACC[k].hi(), Lit(0))
            self.stamp(None)

    # TR_EARLY=1 (experiment, off): the first two key polynomials of a pass requested before the pass starts (under the body phase /
    # under the last conversions).  Bit-identical; measured 12.72-12.87 against 12.58-12.76 ms per launch (same box, alternating):
    # what a pass waits for at its start is not worth the vector-memory slots the early requests take from the other waves
    EARLY = os.environ.get("TR_EARLY", "0") != "0"

    def early_sets(self):
        """Register sets in which the first two key polynomials of a pass can land before the pass starts: the accumulators of
        columns 1 and 2 (P[1], P[2]) are not written before polynomials 1 and 2 are multiplied -- polynomial 0 (column 0) is read
        from P[1] while F[0] = P[0] is started, polynomial 1 (column 1) from P[2] while F[1] = P[1] is started, and P[2] is
        started by polynomial 2, which arrives through the ordinary sets."""
        return [[quad(P[1], r) for r in range(8)], [quad(P[2], r) for r in range(8)]]

    def key_prefetch(self, j):
        """issue early polynomial j (0 or 1) of the pass whose base is in PB (streamed keys only)"""
        if self.KEYS == "global" and self.EARLY: self.key_load(j, self.early_sets()[j])

    def key_pass(self, early=False):
        """F[c] += D[j / 3] * key(j), j = 0..17, for the three columns c; D from the accumulation registers, the key
        polynomials at PB + key_off(j).  Returns the quads of the three Fourier accumulators.  early: polynomials 0 and 1 were
        requested by key_prefetch before the pass."""
        e = self.e
        Fq = [[quad(P[c], r) for r in range(8)] for c in range(KP)]
        KS = [[quad(K[b], r) for r in range(8)] for b in range(2)]
        DCq = [quad(DC, r) for r in range(8)]
        NK = 2 * AUTO_L * KP
        def load_dc(g):
            if "A" in ABL: return
            for r in range(8):
                for k in range(4):
                    self.e("v_accvgpr_read_b32", V(DCq[r].i + k), A((g * 8 + r) * 4 + k))

        def mac(j, r, kset):
            if "M" in ABL: return
            c, g, d, first = j % KP, kset[r], DCq[r], j < KP
            re0 = Lit(0, f64=0.0) if first else qre(Fq[c][r])
            im0 = Lit(0, f64=0.0) if first else qim(Fq[c][r])
            self.e("v_fma_f64", qre(Fq[c][r]), qre(d), qre(g), re0)
            self.e("v_fma_f64", qim(Fq[c][r]), qre(d), qim(g), im0)
            self.e("v_fma_f64", qre(Fq[c][r]), -qim(d), qim(g), qre(Fq[c][r]))
            self.e("v_fma_f64", qim(Fq[c][r]), qim(d), qre(g), qim(Fq[c][r]))

        def macs(j, kset):
            if j % KP == 0: load_dc(j // KP)
            for r in range(0, 8, 2): self.merge([self.chain(mac, j, r, kset), self.chain(mac, j, r + 1, kset)])
        K2 = [V(GA[0].i + 4 * r, 4) for r in range(4)] + [V(GS[0].i + 4 * r, 4) for r in range(4)]     # the gather tables' registers, free here
        if self.KEYS == "global":
            # Every wave streams the 18 key polynomials of the pass into registers itself, three register sets
            # (the two key sets + the gather tables' registers, free here), three polynomials in flight: no ring, no
            # barrier, no LDS traffic in the pass (418 cycles per polynomial against 485 through the ring and 175 of
            # arithmetic; the kernel as a whole takes the same time).
            KS = KS + [K2]
            if early and self.EARLY:
                # polynomials 0 and 1 are (nearly) there; 2, 3, 4 start now and land while 0 and 1 are multiplied
                ES = self.early_sets()
                for j in range(2, 5): self.key_load(j, KS[(j - 2) % 3])
                for j in range(NK):
                    macs(j, ES[j] if j < 2 else KS[(j - 2) % 3])
                    if j >= 2 and j + 3 < NK: self.key_load(j + 3, KS[(j - 2) % 3])
            else:
                for j in range(3): self.key_load(j, KS[j % 3])
                for j in range(NK):
                    macs(j, KS[j % 3])                       # e() waits for the polynomial's loads at their first use
                    if j + 3 < NK: self.key_load(j + 3, KS[j % 3])
        else:
            # Ring of four slots, one key polynomial per slot, DMA three polynomials ahead (a pair-wise ring with one
            # barrier per two polynomials was measured slower: one pair of lookahead does not cover the DMA latency).
            self.ring_barrier()                              # every wave is done with its tile
            NS = self.NSLOT
            SLR = list(SL)
            for k in range(4, NS):
                e("s_add_u32", TS.T, TS.SLOT0, Lit(self.slot_off(k)))
                e("v_add_u32", TMP[k - 4], TS.T, L16)
                SLR.append(TMP[k - 4])
            for j in range(NS - 1): self.dma(j)
            for j in range(NK):
                self.ring_barrier(need=j)                    # polynomial j landed everywhere; slot (j - 1) % NS is free
                if "R" not in ABL:
                    for r in range(8): self.ds_read("ds_read_b128", KS[j % 2][r], SLR[j % NS], offset=1024 * r)
                if j + NS - 1 < NK: self.dma(j + NS - 1)
                if j >= 1: macs(j - 1, KS[(j - 1) % 2])       # covers the slot read above
            macs(NK - 1, KS[(NK - 1) % 2])
            self.ring_barrier()                              # every wave has read the last slot: tiles free again
            self.skew()
        return Fq

    def step(self):
        e = self.e
        self.p.label("L_tr%=")
        self.c("---- t = N/2^idx + 1, t^-1 mod 2N (two Newton steps), gather addresses: coefficient i of p(X^t) is +-p[i t^-1 mod 2N]")
        e("s_lshr_b32", TS.T, Lit(N), TS.IDX)
        e("s_add_u32", TS.T, TS.T, Lit(1))                      # t
        e("s_mov_b32", TS.TINV, TS.T)
        for _ in range(2):
            e("s_mul_i32", TS.T2, TS.T, TS.TINV)
            e("s_sub_u32", TS.T2, Lit(2), TS.T2)
            e("s_mul_i32", TS.TINV, TS.TINV, TS.T2)
        e("s_lshl_b32", TS.C64, TS.TINV, Lit(6))
        e("s_and_b32", TS.C64, TS.C64, Lit(2 * N - 1))
        e("v_mul_lo_u32", IDX0, LANE, TS.TINV)
        e("v_and_b32", IDX0, Lit(2 * N - 1), IDX0)
        cur = TMP[0]
        e("v_mov_b32", cur, IDX0)
        for m in range(16):
            if m:
                e("v_add_u32", cur, TS.C64, cur)
                e("v_and_b32", cur, Lit(2 * N - 1), cur)
            e("v_bfe_i32", GS[m], cur, Lit(LOGN), Lit(1))
            e("v_and_b32", TMP[1], Lit(N - 1), cur)
            e("v_lshl_add_u32", GA[m], TMP[1], Lit(3), TS.GLB)
        # ---- the two mask polynomials: gather, decompose (3 levels), transform, park in AGPRs
        for q in range(2):
            self.c("---- mask polynomial %d: gather p(X^t), 3-level decomposition, three forward transforms" % q)
            X = [V(K[0] + 2 * j, 2) for j in range(16)]
            for m in range(8):
                self.ds_read("ds_read_b64", X[m], GA[m], offset=q * 8192)
                self.ds_read("ds_read_b64", X[m + 8], GA[m + 8], offset=q * 8192)
            chains = []
            for m in range(8):
                for half in range(2):
                    j = m + 8 * half
                    outs = [V(P[lev] + 4 * m + 2 * half, 2) for lev in range(AUTO_L)]
                    k = len(chains)
                    chains.append(self.chain(self.digits3, X[j], GS[j], outs, TMP[3 * k:3 * k + 3], CARRY[k]))
                    if len(chains) == 4:
                        self.merge(chains); chains = []
            polys = [[quad(P[lev], r) for r in range(8)] for lev in range(AUTO_L)]
            xq = [quad(K[0], r) for r in range(8)] + [quad(K[1], r) for r in range(8)]
            polys, spare = self.fwd3(polys, xq)
            for lev in range(AUTO_L):
                for r in range(8):
                    for k in range(4):
                        e("v_accvgpr_write_b32", A(((q * AUTO_L + lev) * 8 + r) * 4 + k), V(polys[lev][r].i + k))
        self.stamp(0)
        early0 = self.KEYS == "global" and self.EARLY
        if early0:               # the first pass's first two key polynomials land under the body phase (P[1], P[2] are free: the transforms are parked)
            e("s_mul_i32", TS.T, TS.IDX, Lit(2 * AUTO_L * KP * 512 * 16))
            e("s_add_u32", TS.PB.lo(), TS.AUTO.lo(), TS.T)
            e("s_addc_u32", TS.PB.hi(), TS.AUTO.hi(), Lit(0))
            self.key_prefetch(0); self.key_prefetch(1)
        # ---- body: g[2] += g[2](X^t)   (reads of every word before the first write)
        self.c("---- body polynomial += its automorphism image")
        OWN = [V(K[0] + 2 * j, 2) for j in range(16)]
        PB_ = [V(K[1] + 2 * j, 2) for j in range(16)]
        for j in range(16): self.ds_read("ds_read_b64", OWN[j], L8T, offset=GL + 2 * 8192 + 512 * j)
        for j in range(16): self.ds_read("ds_read_b64", PB_[j], GA[j], offset=2 * 8192)

        def body(j, k):
            lo, hi = PB_[j].lo(), PB_[j].hi()
            e_ = self.e
            e_("v_xor_b32", lo, lo, GS[j]); e_("v_xor_b32", hi, hi, GS[j])
            e_("v_sub_co_u32", lo, CARRY[k], lo, GS[j]); e_("v_subb_co_u32", hi, CARRY[k], hi, GS[j], CARRY[k])
            e_("v_add_co_u32", lo, CARRY[k], OWN[j].lo(), lo); e_("v_addc_co_u32", hi, CARRY[k], OWN[j].hi(), hi, CARRY[k])
            self.ds_write("ds_write_b64", L8T, PB_[j], offset=GL + 2 * 8192 + 512 * j)
        for j0 in range(0, 16, 4):
            self.merge([self.chain(body, j0 + k, k) for k in range(4)])
        self.stamp(1)
        # ---- two passes over the split key
        def pass_base(s_):
            # pass base pointer: auto_f + s * split_stride + idx * (2 * 3 * 3 * 512 * 16)
            e("s_mul_i32", TS.T, TS.IDX, Lit(2 * AUTO_L * KP * 512 * 16))
            if s_: e("s_add_u32", TS.T, TS.T, TS.SPLIT)
            e("s_add_u32", TS.PB.lo(), TS.AUTO.lo(), TS.T)
            e("s_addc_u32", TS.PB.hi(), TS.AUTO.hi(), Lit(0))
        for s_ in range(2):
            self.c("---- key pass %d (split %s): F[c] += D[j / 3] * key(j), j = 0..17" % (s_, "low 41 bits" if s_ == 0 else "high 23 bits"))
            if not (self.KEYS == "global" and self.EARLY): pass_base(s_)       # (early mode: set where the pass's first two polynomials are requested)
            Fq = self.key_pass(early=True)
            self.stamp(2)
            # ---- three inverse transforms together, then the accumulation into the GLWE
            xq = [quad(K[0], r) for r in range(8)] + [quad(K[1], r) for r in range(8)] + [quad(DC, r) for r in range(8)]
            Fq, spare = self.inv3(Fq, xq)
            self.stamp(3)
            self.c("---- g[c] += round(inverse)%s" % (" << 41" if s_ else ""))
            pre = s_ == 0 and self.KEYS == "global" and self.EARLY
            if pre: pass_base(1)
            for c in ((1, 2, 0) if pre else range(KP)):      # (columns 1 and 2 first: their registers take the next pass's first two key polynomials)
                if s_ == 0:
                    CUR = [V(K[0] + 2 * j, 2) for j in range(16)]            # full words
                    for j in range(16): self.ds_read("ds_read_b64", CUR[j], L8T, offset=GL + c * 8192 + 512 * j)
                    tmp = [(V(K[1] + 4 * k, 2), V(K[1] + 4 * k + 2, 2)) for k in range(4)]

                    def upd(m, half, k, c=c, CUR=CUR, tmp=tmp):
                        j = m + 8 * half
                        x = V(Fq[c][m].i + 2 * half, 2)
                        self.torus_add(x, CUR[j].lo(), CUR[j].hi(), CUR[j].lo(), CUR[j].hi(), tmp[k][0], tmp[k][1], CARRY[k])
                        self.ds_write("ds_write_b64", L8T, CUR[j], offset=GL + c * 8192 + 512 * j)
                else:
                    CURH = [V(K[0] + j) for j in range(16)]                  # only the high dwords change
                    for j in range(16): self.ds_read("ds_read_b32", CURH[j], L8T, offset=GL + c * 8192 + 512 * j + 4)
                    tmp = [V(K[1] + 2 * k, 2) for k in range(4)]

                    def upd(m, half, k, c=c, CURH=CURH, tmp=tmp):
                        j = m + 8 * half
                        x = V(Fq[c][m].i + 2 * half, 2)
                        e_ = self.e
                        # low dword of round(x): x mod 2^32 exactly, then the 2^52 magic; (word << 41) adds (low << 9) to the high dword
                        e_("v_mul_f64", tmp[k], x, TS.C_2M32)
                        e_("v_rndne_f64", tmp[k], tmp[k])
                        e_("v_fma_f64", x, tmp[k], TS.C_M2P32, x)
                        e_("v_add_f64", x, x, SG.C_52)
                        e_("v_lshl_add_u32", CURH[j], x.lo(), Lit(AUTO_SPLIT - 32), CURH[j])
                        self.ds_write("ds_write_b32", L8T, CURH[j], offset=GL + c * 8192 + 512 * j + 4)
                chains = []
                for m in range(8):
                    for half in range(2):
                        chains.append(self.chain(upd, m, half, len(chains)))
                        if len(chains) == 4:
                            self.merge(chains); chains = []
                if pre and c == 1: self.key_prefetch(0)
                if pre and c == 2: self.key_prefetch(1)
            self.stamp(4)
        # ---- loop control
        self.lg_wait_all()
        e("s_add_u32", TS.IDX, TS.IDX, Lit(1))
        e("s_cmp_lt_u32", TS.IDX, Lit(self.steps))
        assert self.vmq == [] and self.lgq == [], (self.vmq, self.lgq)
        e("s_cbranch_scc1", "L_tr%=")

    def build(self):
        self.prologue()
        self.skew()
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
    """operands: %0 auto_f (64), %1 split stride in bytes, %2 wave index, %3 LDS address of slot 0 (= wave 0's tile), %4 this wave's tile,
    %5 the tie mode (1 = balanced top-level tie, 0 = tfhe-rs' rule); stamps: %6 the debug buffer"""
    head = ["s_mov_b64 %s, %%0" % TS.AUTO, "s_mov_b32 %s, %%1" % TS.SPLIT, "s_mov_b32 %s, %%2" % TS.WAVE,
            "s_mov_b32 %s, %%3" % TS.SLOT0, "s_mov_b32 %s, %%4" % TS.TILE, "s_mov_b32 %s, %%5" % TS.TIEMODE]
    if stamps: head.append("s_mov_b64 %s, %%6" % brgen.ST_DBG)
    # M0 carries the LDS-DMA destinations: saved and restored around the loop (it is not on the clobber list)
    lines = ["  " + h for h in head] + ["  s_mov_b32 %s, m0" % TS.M0SAVE] + prog.render() + ["  s_mov_b32 m0, %s" % TS.M0SAVE]
    out = ["// generated by tools/asmgen/trgen.py -- do not edit"]
    for l in lines: out.append('"%s\\n"' % l.replace('"', '\\"'))
    return "\n".join(out) + "\n"


def render_clobbers():
    regs = ["v%d" % i for i in range(256)] + ["a%d" % i for i in range(192)] + ["s%d" % i for i in range(34, 96)] + ["vcc", "scc", "memory"]
    return "// generated by tools/asmgen/trgen.py -- do not edit\n" + ", ".join('"%s"' % r for r in regs) + "\n"


if __name__ == "__main__":
    prog = TraceGen(steps=int(os.environ.get("TR_STEPS", LOGN))).build()
    if len(sys.argv) > 2 and sys.argv[1] == "--out":
        open(os.path.join(sys.argv[2], "tx_trace_asm.inc"), "w").write(render_inc(prog))
        open(os.path.join(sys.argv[2], "tx_trace_asm_stamps.inc"), "w").write(render_inc(TraceGen(stamps=True).build(), stamps=True))
        open(os.path.join(sys.argv[2], "tx_trace_asm_clobbers.inc"), "w").write(render_clobbers())
    else:
        c = prog.count()
        for k in sorted(c, key=lambda k: -c[k]): print("%6d %s" % (c[k], k))
        print("%6d total" % sum(c.values()))
