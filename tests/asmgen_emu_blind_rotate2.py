"""Runs the generated PAIRED blind-rotation loops (tools/asmgen/brgen2.py: wave A = the two mask polynomials and columns 0, 1;
wave B = the body polynomial and column 2) as two emulated waves of one workgroup -- shared LDS image, every s_barrier
honoured, scheduled between barriers in both orders -- against the same oracle-built reference as
tests/asmgen_emu_blind_rotate.py.  Must match bit for bit.

    python3 tests/asmgen_emu_blind_rotate2.py [steps=2] [seed=1]
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
import brgen
import brgen2
import asmgen_emu_blind_rotate as one

N, H, KP = one.N, one.H, one.KP
TW_BYTES = one.TW_BYTES
LDS_BYTES = TW_BYTES + brgen2.UNIT_LDS


def run_pair(emus, progs, order):
    gens = [e.run_gen(p) for e, p in zip(emus, progs)]
    alive = [True] * len(gens)
    while any(alive):
        states = []
        for k in order:
            if not alive[k]: continue
            try:
                states.append(next(gens[k]))
            except StopIteration:
                alive[k] = False
                states.append("end")
        if len(set(states)) != 1: raise RuntimeError("waves disagree at a barrier: %s" % states)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    L = one.oracle()
    H_ = gcn._helpers()
    p = one.p
    tw = np.ctypeslib.as_array(L.txo_twiddles_1024(), shape=(15, 64, 2)).copy().view(np.complex128).reshape(15, 64)
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
    if steps >= 2: lwe[1] = 0
    ref = one.reference(L, H_, lwe, key, steps)
    rot0 = 2 * N - int(L.txo_modswitch(C.c_uint64(int(lwe[-1])), 10, 3))
    tv = np.zeros(N, np.uint64)
    for i in range(N):
        v = np.uint64(1) << np.uint64(63 - 2 * ((i & 7) + 1))
        tv[i] = (np.uint64(0) - v) if i < N // 2 else v
    progs = [brgen2.PairGen("A").build(), brgen2.PairGen("B").build()]
    ok = True
    for order in ([0, 1], [1, 0]):
        lds = np.zeros(LDS_BYTES, np.uint8)
        lw = lds.view(np.complex128)
        lw[:8 * 64] = tw[:8].reshape(-1)
        for r in range(7): lw[8 * 64 + r * 8: 8 * 64 + r * 8 + 8] = tw[8 + r, :8]
        lds[TW_BYTES + brgen2.TILE_BYTES: TW_BYTES + brgen2.TILE_BYTES + 8192].view(np.uint64)[:] = one.rotate(tv, rot0)    # tile B: the body
        CT, ROW, OUT = 0x1000_0000, 0x2000_0000, 0x3000_0000
        ct_buf = lwe.copy().view(np.uint8)
        key_buf = key.reshape(-1).view(np.uint8).copy()
        out_buf = np.zeros(KP * N * 8, np.uint8)
        SG = brgen.SG
        emus = []
        for w in range(2):
            e = gcn.Emu(lds, [(CT, ct_buf), (ROW, key_buf), (OUT, out_buf)])
            for s, val in ((SG.CT, CT), (SG.ROW, ROW + (2 * H * 16 if w else 0)), (SG.OUT, OUT + (2 * N * 8 if w else 0))):
                e.s[s.i] = val & 0xFFFFFFFF; e.s[s.i + 1] = val >> 32
            e.s[SG.NSTEP.i] = steps
            e.s[SG.TILE.i] = TW_BYTES + w * brgen2.TILE_BYTES
            e.s[SG.TIEMODE.i] = 1
            emus.append(e)
        t = time.time()
        run_pair(emus, progs, order)
        got = out_buf.view(np.uint64).reshape(KP, N)
        same = np.array_equal(got, ref)
        ok = ok and same
        print("steps %d seed %d order %s: %d + %d instructions, %.1f s -> %s" %
              (steps, seed, order, emus[0].nexec, emus[1].nexec, time.time() - t, "BIT-IDENTICAL" if same else "MISMATCH"))
        if not same:
            for q in range(KP):
                bad = np.nonzero(got[q] != ref[q])[0]
                print("  poly %d: %d mismatches" % (q, bad.size), bad[:8], [hex(int(x)) for x in got[q][bad[:3]]], [hex(int(x)) for x in ref[q][bad[:3]]])
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
