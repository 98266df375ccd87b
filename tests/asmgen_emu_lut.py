"""Runs the generated LUT CMux tree (tools/asmgen/lutgen.py: eight CMux steps on an accumulator GLWE in the wave's LDS slice, the
selector GGSWs streamed through a register ring) in the lane-exact emulator against (1) a reference built from the CPU oracle's
primitives (txo_decompose_idx, txo_fft_forward_int, txo_fft_inverse_add, products in the oracle's order) on the WHOLE GLWE
after every requested number of steps, and (2) txo_lut_eval itself (the four LWEs extracted at 0, 256, 512, 768).  Random
Fourier GGSWs of realistic magnitude.  Must match bit for bit; a register read before its load was waited for raises.

    python3 tests/asmgen_emu_lut.py [bits=8] [seed=1]
"""
import ctypes as C
import os
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools", "asmgen"))
sys.path.insert(0, HERE)
import gcn
import lutgen
import oracle_lib as O
from asmgen_emu_blind_rotate import vfma, rotate

N, H, KP, LV = 1024, 512, 3, 7
TW_BYTES = (8 * 64 + 7 * 8) * 16
WAVE_LDS = 8320 + 3 * 8192             # padded tile + the accumulator GLWE
LDS_BYTES = TW_BYTES + WAVE_LDS


def reference(L, H_, acc, gg, bits):
    """acc [3][N] torus words; gg [bits][LV][3][3][H] complex (level 1 = coarsest first)"""
    acc = acc.copy()
    one = (C.c_int64 * LV)()
    for i in range(bits):
        k = 1 << i
        re = [None] * KP; im = [None] * KP
        for r in range(KP):
            diff = rotate(acc[r], 2 * N - k) - acc[r]
            dig = np.zeros((LV, N), np.int64)
            for j in range(N):
                L.txo_decompose_idx(C.c_uint64(int(diff[j])), 2, LV, j & 1, one)
                dig[:, j] = one[:]
            for lev in range(LV):                       # finest level first; the key stores the coarsest first
                out = np.zeros(N, np.float64)
                L.txo_fft_forward_int(N, O.p_i64(np.ascontiguousarray(dig[lev])), O.p_f64(out))
                d = out.view(np.complex128)
                for c in range(KP):
                    g = gg[i, LV - 1 - lev, r, c]
                    if re[c] is None: re[c] = np.zeros(H); im[c] = np.zeros(H)
                    re[c] = vfma(H_, d.real.copy(), g.real.copy(), re[c])
                    im[c] = vfma(H_, d.real.copy(), g.imag.copy(), im[c])
                    re[c] = vfma(H_, -d.imag.copy(), g.imag.copy(), re[c])
                    im[c] = vfma(H_, d.imag.copy(), g.real.copy(), im[c])
        for c in range(KP):
            Fc = np.empty(N, np.float64); Fc[0::2] = re[c]; Fc[1::2] = im[c]
            a = acc[c].copy()
            L.txo_fft_inverse_add(N, O.p_f64(Fc), O.p_u64(a))
            acc[c] = a
    return acc


def main():
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    L = O.lib()
    H_ = gcn._helpers()
    L.txo_decompose_idx.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int64)]
    L.txo_fft_forward_int.argtypes = [C.c_uint32, O.i64p, O.f64p]
    L.txo_fft_inverse_add.argtypes = [C.c_uint32, O.f64p, O.u64p]
    L.txo_fft_forward_torus_scaled.argtypes = [C.c_uint32, O.u64p, O.f64p]
    tw = np.ctypeslib.as_array(L.txo_twiddles_1024(), shape=(15, 64, 2)).copy().view(np.complex128).reshape(15, 64)
    gg = np.zeros((8, LV, KP, KP, H), np.complex128)        # all eight selectors (txo_lut_eval walks them all)
    for idx in np.ndindex(8, LV, KP, KP):
        pol = rng.integers(0, 1 << 63, N, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, N, dtype=np.uint64)
        out = np.zeros(N, np.float64)
        L.txo_fft_forward_torus_scaled(N, O.p_u64(pol), O.p_f64(out))
        gg[idx] = out.view(np.complex128)
    acc0 = rng.integers(0, 1 << 63, (KP, N), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (KP, N), dtype=np.uint64)
    acc0[2, :8] = [0, 1, 2**64 - 1, 2**63, 2**63 - 1, 3 << 49, (1 << 50) - 1, 1 << 49]      # rounding boundaries of the decomposition
    ref = reference(L, H_, acc0, gg, bits)

    lds = np.zeros(LDS_BYTES, np.uint8)
    lw = lds.view(np.complex128)
    lw[:8 * 64] = tw[:8].reshape(-1)
    for r in range(7): lw[8 * 64 + r * 8: 8 * 64 + r * 8 + 8] = tw[8 + r, :8]
    glwe = lds[TW_BYTES + 8320: TW_BYTES + 8320 + 3 * 8192].view(np.uint64)
    glwe[:] = acc0.reshape(-1)
    GG = 0x5000_0000
    key_buf = gg.reshape(-1).view(np.uint8).copy()
    TS = lutgen.TS
    prog = lutgen.LutGen(bits=bits, stamps=bool(os.environ.get("BR_STAMPS"))).build()
    e = gcn.Emu(lds, [(GG, key_buf)])
    e.s[TS.AUTO.i] = GG & 0xFFFFFFFF; e.s[TS.AUTO.i + 1] = GG >> 32
    e.s[TS.WAVE.i] = 0; e.s[TS.SLOT0.i] = TW_BYTES; e.s[TS.TILE.i] = TW_BYTES
    e.s[TS.TIEMODE.i] = 1
    if os.environ.get("BR_STAMPS"):
        import brgen
        dbg = np.zeros(128, np.uint8); e.g.append((0x4000_0000, dbg)); e.s[brgen.ST_DBG.i] = 0x4000_0000; e.s[brgen.ST_DBG.i + 1] = 0
    t = time.time()
    e.run(prog)
    got = glwe.reshape(KP, N)
    ok = np.array_equal(got, ref)
    top = max(max((x.i + x.n) for x in i.a if isinstance(x, gcn.V)) for i in prog.ins if any(isinstance(x, gcn.V) for x in i.a))
    print("LUT tree, %d CMux steps, seed %d: %d instructions, top VGPR %d, %.1f s -> %s" % (bits, seed, e.nexec, top, time.time() - t, "BIT-IDENTICAL" if ok else "MISMATCH"))
    if not ok:
        for q in range(KP):
            bad = np.nonzero(got[q] != ref[q])[0]
            print("  poly %d: %d mismatches" % (q, bad.size), bad[:8], [hex(int(x)) for x in got[q][bad[:3]]], [hex(int(x)) for x in ref[q][bad[:3]]])
        sys.exit(1)
    if bits == 8:
        # the oracle's own entry point on the same selectors and table: LWEs extracted at 0, 256, 512, 768
        p = O.aes_tight(n=256)
        keys = O.Keys(p, seed=seed + 20)
        ctx = O.Ctx(keys, threads=2)
        luts = np.zeros((2, KP * N), np.uint64); luts[0] = acc0.reshape(-1)
        ggsw_f = np.ascontiguousarray(gg.reshape(-1).view(np.float64))
        out = ctx.lut_eval(ggsw_f, luts.reshape(-1), 1, 2 * KP * N, 2 * KP * N)          # [1][8][2049]
        for i in range(4):
            nth = 256 * i
            lwe = np.zeros(2 * N + 1, np.uint64)
            for q in range(2):
                a = got[q]
                lwe[q * N: q * N + nth + 1] = a[nth::-1]
                lwe[q * N + nth + 1: (q + 1) * N] = (np.uint64(0) - a[N - 1: nth: -1])
            lwe[2 * N] = got[2][nth]
            if not np.array_equal(lwe, out[0, i]):
                print("  sample extraction at %d differs from txo_lut_eval" % nth); ok = False
        ctx.close()
        print("txo_lut_eval (extractions at 0, 256, 512, 768):", "BIT-IDENTICAL" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
