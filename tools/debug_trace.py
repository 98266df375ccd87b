"""exploratory: run k_extract_trace4 (possibly a TR_STEPS-limited debug build given by TX_LIB) and k_extract_trace on the
same random inputs; save inputs and outputs under gpurun_out/dbg_trace/ for an offline comparison with the emulator"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
import numpy as np, ctypes as C
import txhip, txclient
if os.environ.get('TX_LIB'): txhip.LIB_PATH = os.environ['TX_LIB']
count = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p = txhip.aes_tight(n=256)
rng = np.random.default_rng(3)
L = txclient.lib()
def rnd(words): return rng.integers(0, 1 << 63, words, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, words, dtype=np.uint64)
keys = [rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p)))]
acc = rnd(count * 3072).reshape(-1, 3, 1024)
out = os.path.join(ROOT, "gpurun_out", "dbg_trace"); os.makedirs(out, exist_ok=True)
np.save(os.path.join(out, "acc.npy"), acc); np.save(os.path.join(out, "auto.npy"), keys[2])
for v in ("0", "4"):
    os.environ["TX_TRACE_VARIANT"] = v
    eng = txhip.Engine(p, 0)
    eng.load_eval_keys(*keys)
    o = eng.extract_trace(acc).reshape(-1, 3, 1024)
    np.save(os.path.join(out, "out_v%s.npy" % v), o)
    eng.close()
print("saved", out)
