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
    print("stamps (cycles/step):", {n: round(st[i] / steps) for i, n in enumerate(names)}, "sum", round(sum(st[:5]) / steps))
    print("inside forward x3:", {n: round(st[8 + i] / steps) for i, n in enumerate(["twist", "dft8+tw+xchg_hi", "dft8+tw+xchg_lo", "dft8"])})
if len(sys.argv) > 4:
    acc = rnd(count // 8 * 3072).reshape(-1, 3, 1024)
    for r in range(reps):
        eng.extract_trace(acc)
        print("extract_trace", acc.shape[0] * 7, "units:", round(eng.timing()["ms_trace"], 3), "ms", flush=True)
