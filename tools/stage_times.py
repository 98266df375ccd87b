"""exploratory: per-kernel-family average launch time of one medium CTR transcipher with random keys (timing is
data-independent); TX_LIB selects an ablation build, e.g. -DTXD_ABL_NOKEY"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "temp-fhe-transciphering_amd", "python"))
import numpy as np, txhip, txclient, ctypes as C
if os.environ.get('TX_LIB'): txhip.LIB_PATH = os.environ['TX_LIB']
p = txhip.aes_tight(n=768)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(words): return rng.integers(0, 1 << 63, words, dtype=np.uint64)
eng = txhip.Engine(p, 0)
eng.load_eval_keys(rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p))))
K, N = 3, 1024
eng.load_ctr_luts(rnd(10 * 3 * 16 * 2 * K * N))
ct = rng.integers(0, 256, 16 * 64, dtype=np.uint8).tobytes()
for r in range(2):
    eng.transcipher(ct, iv=bytes(16))
    t = eng.timing()
    print({k: round(v / max(1, t["n_" + k[3:]]), 3) for k, v in t.items() if k.startswith("ms_") and k != "ms_total"}, flush=True)
