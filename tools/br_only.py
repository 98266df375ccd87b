"""exploratory: time / profile the blind-rotation kernel alone (random keys: timing is data-independent)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
import numpy as np
import txhip, txclient
if os.environ.get('TX_LIB'): txhip.LIB_PATH = os.environ['TX_LIB']
import ctypes as C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
p = txhip.aes_tight(n=n)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(words): return rng.integers(0, 1 << 63, words, dtype=np.uint64)
eng = txhip.Engine(p, 0)
eng.load_eval_keys(rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p))))
lwe = rnd(count * (n + 1)).reshape(count, n + 1)
for r in range(reps):
    eng.blind_rotate(lwe)
    print("blind_rotate", count, "units:", round(eng.timing()["ms_blind_rotate"], 3), "ms", flush=True)
if hasattr(txhip.lib(), "tx_debug_stamps"):        # -DTXD_STAMPS build: cycles per phase and step of one wave
    st = (C.c_ulonglong * 16)()
    txhip.lib().tx_debug_stamps(C.c_void_p(eng.h), st)
    names = ["rotate+decompose", "forward x3", "MAC", "inverse x3", "convert+update"]
    steps = max(1, st[6])
    print("stamps (cycles/step):", {n: round(st[i] / steps) for i, n in enumerate(names)}, "sum", round(sum(st[:5]) / steps),
          "| waiting at barriers (k_blind_rotate6):", round(st[5] / steps))
    print("raw stamps 8..11:", [int(st[k]) for k in range(8, 12)])
    if st[10] > st[8] and st[11] > st[9]:      # generated loop: s_memtime / s_memrealtime (100 MHz) at the start and the end of the unit
        print("shader clock of that wave: %.3f GHz over %.2f ms" % ((st[10] - st[8]) / (st[11] - st[9]) * 0.1, (st[11] - st[9]) * 1e-5))
    print("inside forward x3:", {n: round(st[8 + i] / steps) for i, n in enumerate(["twist", "dft8+tw+xchg_hi", "dft8+tw+xchg_lo", "dft8"])})
if len(sys.argv) > 4:
    acc = rnd(count // 8 * 3072).reshape(-1, 3, 1024)
    for r in range(reps):
        eng.extract_trace(acc)
        print("extract_trace", acc.shape[0] * 7, "units:", round(eng.timing()["ms_trace"], 3), "ms", flush=True)
if hasattr(txhip.lib(), "tx_debug_wave_times") and os.environ.get("TX_BR_VARIANT", "4") in ("4", "6"):
    nw = min(4096, count // 4 if os.environ.get("TX_BR_VARIANT", "4") == "4" else count)
    buf = (C.c_ulonglong * (2 * nw))()
    txhip.lib().tx_debug_wave_times(C.c_void_p(eng.h), buf, 2 * nw)
    a = np.array(buf[:], np.uint64).reshape(-1, 2).astype(np.int64)
    a = a[a[:, 1] > a[:, 0]]
    t0 = a[:, 0].min()
    life = (a[:, 1] - a[:, 0]) * 1e-5; end = (a[:, 1] - t0) * 1e-5; start = (a[:, 0] - t0) * 1e-5
    print("workgroups %d: lifetime min/median/max %.2f / %.2f / %.2f ms; last start %.2f ms, last end %.2f ms" % (a.shape[0], life.min(), np.median(life), life.max(), start.max(), end.max()))
    h, edges = np.histogram(start, bins=np.arange(0, end.max() + 2, 2.0))
    print("starts per 2 ms:", h.tolist())
    h, edges = np.histogram(end, bins=np.arange(0, end.max() + 2, 2.0))
    print("ends per 2 ms:  ", h.tolist())
    print("lifetime by start time (2 ms bins):", [round(float(np.median(life[(start >= lo) & (start < lo + 2)])), 2) if ((start >= lo) & (start < lo + 2)).any() else None for lo in np.arange(0, start.max() + 2, 2.0)])
if hasattr(txhip.lib(), "tx_debug_wave_times") and os.environ.get("TX_BR_VARIANT") == "5":
    nw = min(2048, (count + 0))
    buf = (C.c_ulonglong * (2 * nw))()
    txhip.lib().tx_debug_wave_times(C.c_void_p(eng.h), buf, 2 * nw)
    a = np.array(buf[:], np.uint64).reshape(-1, 2).astype(np.int64)
    a = a[a[:, 1] > a[:, 0]]
    t0 = a[:, 0].min()
    life = (a[:, 1] - a[:, 0]) * 1e-5; end = (a[:, 1] - t0) * 1e-5; start = (a[:, 0] - t0) * 1e-5
    print("waves %d: start max %.3f ms; lifetime min/median/max %.2f / %.2f / %.2f ms; end min/median/max %.2f / %.2f / %.2f ms" %
          (a.shape[0], start.max(), life.min(), np.median(life), life.max(), end.min(), np.median(end), end.max()))
    # by CU (8 consecutive wave slots) and by SIMD pair
    cu = end.reshape(-1, 8) if end.size % 8 == 0 else None
    if cu is not None:
        print("per workgroup: max-min of its 8 waves' end times: median %.2f ms, max %.2f ms; workgroup end time min/median/max %.2f / %.2f / %.2f" %
              (np.median(cu.max(1) - cu.min(1)), (cu.max(1) - cu.min(1)).max(), cu.max(1).min(), np.median(cu.max(1)), cu.max(1).max()))
