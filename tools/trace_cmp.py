"""exploratory: extract + trace kernels against each other (TX_TRACE_VARIANT 4 = one wave per unit, 5 = two waves per unit): same bits, time"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
import numpy as np
import txhip, txclient
if os.environ.get('TX_LIB'): txhip.LIB_PATH = os.environ['TX_LIB']
import ctypes as C
count = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = txhip.aes_tight(n=768)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(words): return rng.integers(0, 1 << 63, words, dtype=np.uint64)
keys = [rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p)))]
acc = rnd(count * 3072).reshape(-1, 3, 1024)
outs = {}
for v in sys.argv[3:] or ["4", "5"]:
    os.environ["TX_TRACE_VARIANT"] = v
    eng = txhip.Engine(p, 0)
    eng.load_eval_keys(*keys)
    for r in range(reps):
        o = eng.extract_trace(acc)
        print("variant", v, "extract_trace", acc.shape[0] * 7, "units:", round(eng.timing()["ms_trace"], 3), "ms", flush=True)
    outs[v] = o
    eng.close()
vs = list(outs)
for a in vs[1:]:
    print("variant", a, "==", vs[0], ":", bool(np.array_equal(outs[a], outs[vs[0]])))
