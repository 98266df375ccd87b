"""Runs the generated COOPERATIVE trace loop (tools/asmgen/trgen2.py: two waves per unit) as a workgroup of two emulated
waves -- one shared LDS image (twiddles, tile A, tile B, body polynomial), every s_barrier honoured, scheduled between
barriers in both orders -- against the CPU oracle's txo_trace() on the same GLWE with real automorphism keys (oracle keygen
at n = 256).  Must match bit for bit.

    python3 tests/asmgen_emu_trace2.py [seed=1]          (TR_STEPS=k: the first k trace steps only, compared with nothing)
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
import trgen2
import oracle_lib as O
from asmgen_emu_blind_rotate3 import run_group_ordered

N, H, KP = 1024, 512, 3
TW_BYTES = (8 * 64 + 7 * 8) * 16
TILE = 8320
LDS_BYTES = TW_BYTES + 2 * TILE + 8192


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rng = np.random.default_rng(seed)
    L = O.lib()
    p = O.aes_tight(n=256)
    keys = O.Keys(p, seed=seed + 10)
    ctx = O.Ctx(keys, threads=4)
    glwe = rng.integers(0, 1 << 63, (1, KP, N), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (1, KP, N), dtype=np.uint64)
    ref = ctx.trace(glwe)[0]
    L.txo_fft_forward_torus_scaled.argtypes = [C.c_uint32, O.u64p, O.f64p]
    std = keys.auto.reshape(-1, N)
    auto_f = np.zeros((2, std.shape[0], H), np.complex128)
    for i in range(std.shape[0]):
        for s, part in ((0, std[i] & np.uint64((1 << 41) - 1)), (1, std[i] >> np.uint64(41))):
            out = np.zeros(N, np.float64)
            L.txo_fft_forward_torus_scaled(N, O.p_u64(np.ascontiguousarray(part)), O.p_f64(out))
            auto_f[s, i] = out.view(np.complex128)
    tw = np.ctypeslib.as_array(L.txo_twiddles_1024(), shape=(15, 64, 2)).copy().view(np.complex128).reshape(15, 64)
    AUTO = 0x5000_0000
    key_buf = auto_f.reshape(-1).view(np.uint8).copy()
    T2 = trgen2.T2
    prog = trgen2.Trace2Gen(steps=int(os.environ.get("TR_STEPS", 10))).build()
    ok = True
    for order in ([0, 1], [1, 0]):
        lds = np.zeros(LDS_BYTES, np.uint8)
        lw = lds.view(np.complex128)
        lw[:8 * 64] = tw[:8].reshape(-1)
        for r in range(7): lw[8 * 64 + r * 8: 8 * 64 + r * 8 + 8] = tw[8 + r, :8]
        for q in range(2): lds[TW_BYTES + q * TILE: TW_BYTES + q * TILE + 8192].view(np.uint64)[:] = glwe[0, q]
        lds[TW_BYTES + 2 * TILE: TW_BYTES + 2 * TILE + 8192].view(np.uint64)[:] = glwe[0, 2]
        emus = []
        for w in range(2):
            e = gcn.Emu(lds, [(AUTO, key_buf)])
            e.s[T2.AUTO.i] = AUTO & 0xFFFFFFFF; e.s[T2.AUTO.i + 1] = AUTO >> 32
            e.s[T2.SPLIT.i] = auto_f.shape[1] * H * 16
            e.s[T2.W.i] = w
            e.s[T2.TILE.i] = TW_BYTES + w * TILE
            e.s[T2.TILEA.i] = TW_BYTES
            emus.append(e)
        t = time.time()
        run_group_ordered(emus, prog, order)
        got = np.stack([lds[TW_BYTES + q * TILE: TW_BYTES + q * TILE + 8192].view(np.uint64) for q in range(2)] +
                       [lds[TW_BYTES + 2 * TILE: TW_BYTES + 2 * TILE + 8192].view(np.uint64)])
        same = np.array_equal(got, ref)
        ok = ok and same
        print("cooperative trace, 2 waves, seed %d order %s: %d instructions per wave in %.1f s -> %s" %
              (seed, order, emus[0].nexec, time.time() - t, "BIT-IDENTICAL" if same else "MISMATCH"))
        if not same:
            for q in range(KP):
                bad = np.nonzero(got[q] != ref[q])[0]
                print("  poly %d: %d mismatches" % (q, bad.size), bad[:6], [hex(int(x)) for x in got[q][bad[:2]]], [hex(int(x)) for x in ref[q][bad[:2]]])
    ctx.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
