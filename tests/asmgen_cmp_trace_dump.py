"""exploratory: compare a GPU dump of tools/debug_trace.py (gpurun_out/dbg_trace) with the emulator run of the same
inputs for TR_STEPS steps"""
import ctypes as C, os, sys, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools", "asmgen")); sys.path.insert(0, HERE)
import gcn, trgen, oracle_lib as O
N, H, KP = 1024, 512, 3
TW_BYTES = (8 * 64 + 7 * 8) * 16; WAVE_LDS = 8320 + 3 * 8192
d = os.path.join(ROOT, "gpurun_out", "dbg_trace")
acc, auto, got = np.load(d + "/acc.npy"), np.load(d + "/auto.npy"), np.load(d + "/out_v4.npy")
steps = int(os.environ.get("TR_STEPS", 10))
L = O.lib()
L.txo_fft_forward_torus_scaled.argtypes = [C.c_uint32, O.u64p, O.f64p]
std = auto.reshape(-1, N)
auto_f = np.zeros((2, std.shape[0], H), np.complex128)
for i in range(std.shape[0]):
    for s, part in ((0, std[i] & np.uint64((1 << 41) - 1)), (1, std[i] >> np.uint64(41))):
        o = np.zeros(N, np.float64); L.txo_fft_forward_torus_scaled(N, O.p_u64(np.ascontiguousarray(part)), O.p_f64(o)); auto_f[s, i] = o.view(np.complex128)
tw = np.ctypeslib.as_array(L.txo_twiddles_1024(), shape=(15, 64, 2)).copy().view(np.complex128).reshape(15, 64)
def prep(a, lvl):
    g = np.zeros((3, N), np.uint64)
    for q in range(2):
        for j in range(N):
            s = j + lvl
            x = a[q, s] if s < N else np.uint64(0) - a[q, s - N]
            g[q, j] = (x >> np.uint64(10)) if j == 0 else np.uint64(0) - ((np.uint64(0) - x) >> np.uint64(10))
    g[2, 0] = (a[2, lvl] + (np.uint64(1) << np.uint64(63 - 2 * (lvl + 1)))) >> np.uint64(10)
    return g
lds = np.zeros(TW_BYTES + 4 * WAVE_LDS, np.uint8)
lw = lds.view(np.complex128); lw[:8 * 64] = tw[:8].reshape(-1)
for r in range(7): lw[8 * 64 + r * 8: 8 * 64 + r * 8 + 8] = tw[8 + r, :8]
with np.errstate(over="ignore"):
    for w in range(4): lds[TW_BYTES + w * WAVE_LDS + 8320: TW_BYTES + w * WAVE_LDS + 8320 + 24576].view(np.uint64)[:] = prep(acc[0], w).reshape(-1)
AUTO = 0x5000_0000; key_buf = auto_f.reshape(-1).view(np.uint8).copy(); TS = trgen.TS
prog = trgen.TraceGen(steps=steps).build(); emus = []
for w in range(4):
    e = gcn.Emu(lds, [(AUTO, key_buf)]); e.s[TS.AUTO.i] = AUTO; e.s[TS.AUTO.i + 1] = 0; e.s[TS.SPLIT.i] = auto_f.shape[1] * H * 16
    e.s[TS.WAVE.i] = w; e.s[TS.SLOT0.i] = TW_BYTES; e.s[TS.TILE.i] = TW_BYTES + w * WAVE_LDS; emus.append(e)
gcn.run_group(emus, prog)
for w in range(4):
    exp = lds[TW_BYTES + w * WAVE_LDS + 8320: TW_BYTES + w * WAVE_LDS + 8320 + 24576].view(np.uint64).reshape(3, N)
    for q in range(3):
        bad = np.nonzero(exp[q] != got[w, q])[0]
        print("unit", w, "poly", q, "mismatches", bad.size, bad[:10], [hex(int(x)) for x in got[w, q][bad[:2]]], [hex(int(x)) for x in exp[q][bad[:2]]])
