"""Runs the generated blind-rotation loop in the lane-exact emulator against a reference built from the CPU
oracle's primitives (txo_fft_forward_int, txo_fft_inverse_add, txo_decompose_idx, txo_modswitch), for a few
CMux steps with a realistic random Fourier key.  Everything must match bit for bit, and no register may be
read before its load was waited for (gcn.Poison).

    python3 tests/asmgen_emu_blind_rotate.py [steps=2] [seed=1]
"""
import ctypes as C
import os
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools", "asmgen"))
import gcn
import brgen

N, H, KP = 1024, 512, 3
TW_BYTES = (8 * 64 + 7 * 8) * 16
TILE = TW_BYTES                      # wave 0's tile
WAVE_BYTES = 8320 + 8192               # padded tile (pitch 65 complex) + body polynomial
LDS_BYTES = TW_BYTES + WAVE_BYTES + 64   # + the two step counters of the fair variant


def oracle():
    L = C.CDLL(os.path.join(ROOT, "oracle", "libtxoracle.so"))
    L.txo_twiddles_1024.restype = C.POINTER(C.c_double)
    L.txo_modswitch.restype = C.c_uint32
    L.txo_modswitch.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
    return L


def p(a, t): return a.ctypes.data_as(C.POINTER(t))


def reference(L, H_, lwe, key, steps):
    """acc after `steps` CMux steps; key: [steps][3][3][512] complex128 (F-layout)"""
    acc = np.zeros((KP, N), np.uint64)
    rot0 = 2 * N - int(L.txo_modswitch(C.c_uint64(int(lwe[-1])), 10, 3))
    tv = np.zeros(N, np.uint64)
    for i in range(N):
        v = np.uint64(1) << np.uint64(63 - 2 * ((i & 7) + 1))
        tv[i] = (np.uint64(0) - v) if i < N // 2 else v
    acc[2] = rotate(tv, rot0)
    for i in range(steps):
        rot = int(L.txo_modswitch(C.c_uint64(int(lwe[i])), 10, 3))
        dF = np.zeros((KP, H), np.complex128)
        for q in range(KP):
            ct1 = rotate(acc[q], rot) - acc[q]
            dig = np.zeros(N, np.int64)
            one = (C.c_int64 * 1)()
            for j in range(N):
                L.txo_decompose_idx(C.c_uint64(int(ct1[j])), 23, 1, j & 1, one)
                dig[j] = one[0]
            out = np.zeros(N, np.float64)
            L.txo_fft_forward_int(N, p(dig, C.c_int64), p(out, C.c_double))
            dF[q] = out.view(np.complex128)
        for c in range(KP):
            re = np.zeros(H); im = np.zeros(H)
            for q in range(KP):
                d, g = dF[q], key[i, q, c]
                re = vfma(H_, d.real.copy(), g.real.copy(), re)
                re = vfma(H_, -d.imag.copy(), g.imag.copy(), re)
                im = vfma(H_, d.real.copy(), g.imag.copy(), im)
                im = vfma(H_, d.imag.copy(), g.real.copy(), im)
            Fc = np.empty(N, np.float64); Fc[0::2] = re; Fc[1::2] = im
            a = acc[c].copy()
            L.txo_fft_inverse_add(N, p(Fc, C.c_double), p(a, C.c_uint64))
            acc[c] = a
    return acc


def vfma(H_, a, b, c):
    o = np.empty_like(a)
    dp = C.POINTER(C.c_double)
    H_.vfma(a.size, a.ctypes.data_as(dp), b.ctypes.data_as(dp), np.ascontiguousarray(c).ctypes.data_as(dp), o.ctypes.data_as(dp))
    return o


def rotate(pol, rot):
    """coefficients of pol(X) * X^rot, rot in [0, 2N]"""
    out = np.zeros(N, np.uint64)
    for j in range(N):
        src = j - rot
        neg = False
        if src < 0: src += N; neg = not neg
        if src < 0: src += N; neg = not neg
        out[j] = (np.uint64(0) - pol[src]) if neg else pol[src]
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    L = oracle()
    H_ = gcn._helpers()
    tw = np.ctypeslib.as_array(L.txo_twiddles_1024(), shape=(15, 64, 2)).copy().view(np.complex128).reshape(15, 64)
    # realistic key: transform of uniformly random torus polynomials, scaled by 2/N as the library stores it
    key = np.zeros((steps, KP, KP, H), np.complex128)
    L.txo_fft_forward_torus_scaled.argtypes = [C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    for i in range(steps):
        for q in range(KP):
            for c in range(KP):
                pol = rng.integers(0, 1 << 63, N, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, N, dtype=np.uint64)
                out = np.zeros(N, np.float64)
                L.txo_fft_forward_torus_scaled(N, p(pol, C.c_uint64), p(out, C.c_double))
                key[i, q, c] = out.view(np.complex128)
    lwe = rng.integers(0, 1 << 63, steps + 1, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, steps + 1, dtype=np.uint64)
    if steps >= 2: lwe[1] = 0                                        # a zero mask element: rotation by 0
    ref = reference(L, H_, lwe, key, steps)

    # ---- the emulated wave
    lds = np.zeros(LDS_BYTES, np.uint8)
    lw = lds.view(np.complex128)
    lw[:8 * 64] = tw[:8].reshape(-1)
    for r in range(7): lw[8 * 64 + r * 8: 8 * 64 + r * 8 + 8] = tw[8 + r, :8]
    body = lds[TILE + 8320: TILE + WAVE_BYTES].view(np.uint64)
    rot0 = 2 * N - int(L.txo_modswitch(C.c_uint64(int(lwe[-1])), 10, 3))
    tv = np.zeros(N, np.uint64)
    for i in range(N):
        v = np.uint64(1) << np.uint64(63 - 2 * ((i & 7) + 1))
        tv[i] = (np.uint64(0) - v) if i < N // 2 else v
    body[:] = rotate(tv, rot0)
    CT, ROW, OUT = 0x1000_0000, 0x2000_0000, 0x3000_0000
    ct_buf = lwe.copy().view(np.uint8)
    key_buf = key.reshape(-1).view(np.uint8).copy()
    out_buf = np.zeros(KP * N * 8, np.uint8)
    emu = gcn.Emu(lds, [(CT, ct_buf), (ROW, key_buf), (OUT, out_buf)])
    SG = brgen.SG
    for s, val in ((SG.CT, CT), (SG.ROW, ROW), (SG.OUT, OUT)):
        emu.s[s.i] = val & 0xFFFFFFFF; emu.s[s.i + 1] = val >> 32
    emu.s[SG.NSTEP.i] = steps
    emu.s[SG.TILE.i] = TILE
    emu.s[SG.TIEMODE.i] = 0 if os.environ.get('EMU_TIE') == 'tfhe' else 1      # tie-mode operand: 1 = balanced, 0 = tfhe-rs' rule
    fair = os.environ.get('BR_FAIR')            # '0': partner never ahead, '1': partner always ahead (both priority paths)
    prog = brgen.Gen(stamps=bool(os.environ.get('BR_STAMPS')), fair=fair is not None).build()
    if fair is not None:
        emu.s[SG.PROGA.i] = TW_BYTES + WAVE_BYTES; emu.s[SG.PROGB.i] = TW_BYTES + WAVE_BYTES + 4; emu.s[SG.PBASE.i] = 1536
        lds[TW_BYTES + WAVE_BYTES + 4: TW_BYTES + WAVE_BYTES + 8].view(np.uint32)[0] = 0xFFFFFFFF if fair == '1' else 0
    if os.environ.get('BR_STAMPS'):
        dbg = np.zeros(128, np.uint8); emu.g.append((0x4000_0000, dbg)); emu.s[brgen.ST_DBG.i] = 0x4000_0000; emu.s[brgen.ST_DBG.i + 1] = 0
    t = time.time()
    emu.run(prog)
    got = out_buf.view(np.uint64).reshape(KP, N)
    ok = np.array_equal(got, ref)
    print("steps %d seed %d: %d instructions executed in %.1f s -> %s" % (steps, seed, emu.nexec, time.time() - t, "BIT-IDENTICAL" if ok else "MISMATCH"))
    if not ok:
        for q in range(KP):
            bad = np.nonzero(got[q] != ref[q])[0]
            print("  poly %d: %d mismatches" % (q, bad.size), bad[:8], [hex(int(x)) for x in got[q][bad[:3]]], [hex(int(x)) for x in ref[q][bad[:3]]])
        sys.exit(1)


if __name__ == "__main__":
    main()
