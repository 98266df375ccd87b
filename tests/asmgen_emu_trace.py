"""Runs the generated trace loop (trgen.py) for a whole workgroup -- four emulated waves sharing one LDS image, the
key ring filled by their LDS-DMA quarters, every s_barrier honoured -- against the CPU oracle's txo_trace() on the
same four GLWEs with real automorphism keys (oracle keygen at n = 256).  Must match bit for bit.

    python3 tests/asmgen_emu_trace.py [seed=1]
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
import trgen
import oracle_lib as O

N, H, KP = 1024, 512, 3
TW_BYTES = (8 * 64 + 7 * 8) * 16
WAVE_LDS = 8320 + 3 * 8192             # padded tile + GLWE
LDS_BYTES = TW_BYTES + 4 * WAVE_LDS + 2 * 8192      # + two more ring slots (TR_SLOTS experiments)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rng = np.random.default_rng(seed)
    L = O.lib()
    p = O.aes_tight(n=256)
    keys = O.Keys(p, seed=seed + 10)
    ctx = O.Ctx(keys, threads=4)
    glwe = rng.integers(0, 1 << 63, (4, KP, N), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (4, KP, N), dtype=np.uint64)
    ref = ctx.trace(glwe)

    # Fourier automorphism keys as tx_load_eval_keys builds them: [split 2][poly][512] complex, scaled by 2/N
    L.txo_fft_forward_torus_scaled.argtypes = [C.c_uint32, O.u64p, O.f64p]
    std = keys.auto.reshape(-1, N)
    auto_f = np.zeros((2, std.shape[0], H), np.complex128)
    for i in range(std.shape[0]):
        for s, part in ((0, std[i] & np.uint64((1 << 41) - 1)), (1, std[i] >> np.uint64(41))):
            out = np.zeros(N, np.float64)
            L.txo_fft_forward_torus_scaled(N, O.p_u64(np.ascontiguousarray(part)), O.p_f64(out))
            auto_f[s, i] = out.view(np.complex128)
    tw = np.ctypeslib.as_array(L.txo_twiddles_1024(), shape=(15, 64, 2)).copy().view(np.complex128).reshape(15, 64)

    lds = np.zeros(LDS_BYTES, np.uint8)
    lw = lds.view(np.complex128)
    lw[:8 * 64] = tw[:8].reshape(-1)
    for r in range(7): lw[8 * 64 + r * 8: 8 * 64 + r * 8 + 8] = tw[8 + r, :8]
    for w in range(4):
        base = TW_BYTES + w * WAVE_LDS + 8320
        lds[base: base + 3 * 8192].view(np.uint64)[:] = glwe[w].reshape(-1)
    AUTO = 0x5000_0000
    key_buf = auto_f.reshape(-1).view(np.uint8).copy()
    TS = trgen.TS
    prog = trgen.TraceGen(steps=int(os.environ.get('TR_STEPS', 10))).build()
    emus = []
    for w in range(4):
        e = gcn.Emu(lds, [(AUTO, key_buf)])
        e.s[TS.AUTO.i] = AUTO & 0xFFFFFFFF; e.s[TS.AUTO.i + 1] = AUTO >> 32
        e.s[TS.SPLIT.i] = auto_f.shape[1] * H * 16
        e.s[TS.WAVE.i] = w
        e.s[TS.SLOT0.i] = TW_BYTES
        e.s[TS.TILE.i] = TW_BYTES + w * WAVE_LDS
        e.s[TS.TIEMODE.i] = 0 if os.environ.get('EMU_TIE') == 'tfhe' else 1
        emus.append(e)
    t = time.time()
    gcn.run_group(emus, prog)
    ok = True
    for w in range(4):
        base = TW_BYTES + w * WAVE_LDS + 8320
        got = lds[base: base + 3 * 8192].view(np.uint64).reshape(KP, N)
        if not np.array_equal(got, ref[w]):
            ok = False
            for q in range(KP):
                bad = np.nonzero(got[q] != ref[w][q])[0]
                print("  wave %d poly %d: %d mismatches" % (w, q, bad.size), bad[:6], [hex(int(x)) for x in got[q][bad[:2]]], [hex(int(x)) for x in ref[w][q][bad[:2]]])
    print("trace, 4 waves, seed %d: %d instructions per wave in %.1f s -> %s" % (seed, emus[0].nexec, time.time() - t, "BIT-IDENTICAL" if ok else "MISMATCH"))
    ctx.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
