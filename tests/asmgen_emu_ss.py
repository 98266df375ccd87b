"""Runs the generated scheme-switching body (ssgen.py) for a whole workgroup -- four emulated waves sharing one LDS
image and the key ring -- against the CPU oracle's txo_scheme_switch() + txo_ggsw_to_fourier() on four GLev elements
with a real scheme-switching key (oracle keygen at n = 256).  Must match bit for bit.

    python3 tests/asmgen_emu_ss.py [seed=1]
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
import ssgen
import oracle_lib as O

N, H, KP = 1024, 512, 3
TW_BYTES = (8 * 64 + 7 * 8) * 16
WAVE_LDS = 8320 + 3 * 8192             # padded tile + the GLev element
LDS_BYTES = TW_BYTES + 4 * WAVE_LDS + 2 * 8192


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rng = np.random.default_rng(seed)
    L = O.lib()
    p = O.aes_tight(n=256)
    keys = O.Keys(p, seed=seed + 20)
    ctx = O.Ctx(keys, threads=4)
    lv = p.cbs_level
    # one "ciphertext" = cbs_level GLev elements; the four waves take elements 0..3 of it
    glev = rng.integers(0, 1 << 63, (1, lv, KP, N), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (1, lv, KP, N), dtype=np.uint64)
    ref = ctx.ggsw_to_fourier(ctx.scheme_switch(glev))           # [1][lv][row 3][col 3][N] doubles

    L.txo_fft_forward_torus_scaled.argtypes = [C.c_uint32, O.u64p, O.f64p]
    std = keys.ss.reshape(-1, N)
    ss_f = np.zeros((std.shape[0], H), np.complex128)
    for i in range(std.shape[0]):
        out = np.zeros(N, np.float64)
        L.txo_fft_forward_torus_scaled(N, O.p_u64(np.ascontiguousarray(std[i])), O.p_f64(out))
        ss_f[i] = out.view(np.complex128)
    tw = np.ctypeslib.as_array(L.txo_twiddles_1024(), shape=(15, 64, 2)).copy().view(np.complex128).reshape(15, 64)

    lds = np.zeros(LDS_BYTES, np.uint8)
    lw = lds.view(np.complex128)
    lw[:8 * 64] = tw[:8].reshape(-1)
    for r in range(7): lw[8 * 64 + r * 8: 8 * 64 + r * 8 + 8] = tw[8 + r, :8]
    for w in range(4):
        base = TW_BYTES + w * WAVE_LDS + 8320
        lds[base: base + 3 * 8192].view(np.uint64)[:] = glev[0, w].reshape(-1)
    SSF, OUTB = 0x5000_0000, 0x6000_0000
    key_buf = ss_f.reshape(-1).view(np.uint8).copy()
    out_buf = np.zeros(4 * KP * KP * H * 16, np.uint8)
    TS = ssgen.TS
    prog = ssgen.SsGen().build()
    emus = []
    for w in range(4):
        e = gcn.Emu(lds, [(SSF, key_buf), (OUTB, out_buf)])
        e.s[TS.AUTO.i] = SSF & 0xFFFFFFFF; e.s[TS.AUTO.i + 1] = SSF >> 32
        o = OUTB + w * KP * KP * H * 16
        e.s[ssgen.OUT.i] = o & 0xFFFFFFFF; e.s[ssgen.OUT.i + 1] = o >> 32
        e.s[TS.WAVE.i] = w
        e.s[TS.SLOT0.i] = TW_BYTES
        e.s[TS.TILE.i] = TW_BYTES + w * WAVE_LDS
        e.s[TS.TIEMODE.i] = 0 if os.environ.get('EMU_TIE') == 'tfhe' else 1
        emus.append(e)
    t = time.time()
    gcn.run_group(emus, prog)
    got = out_buf.view(np.float64).reshape(4, KP, KP, N)
    ok = True
    for w in range(4):
        for row in range(KP):
            for c in range(KP):
                a, b = got[w, row, c].view(np.uint64), ref[0, w, row, c].view(np.uint64)
                if not np.array_equal(a, b):
                    ok = False
                    bad = np.nonzero(a != b)[0]
                    print("  wave %d row %d col %d: %d mismatches" % (w, row, c, bad.size), bad[:6], got[w, row, c][bad[:2]], ref[0, w, row, c][bad[:2]])
    print("scheme switch, 4 waves, seed %d: %d instructions per wave in %.1f s -> %s" % (seed, emus[0].nexec, time.time() - t, "BIT-IDENTICAL" if ok else "MISMATCH"))
    ctx.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
