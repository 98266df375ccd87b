"""exploratory: per-family kernel times of tx_transcipher on 8 CTR blocks (the per-GPU share of an 8-GPU run), TX_LIB selects a build"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "temp-fhe-transciphering_amd", "python"))
import numpy as np, txhip, txclient, ctypes as C
if os.environ.get('TX_LIB'): txhip.LIB_PATH = os.environ['TX_LIB']
p = txhip.aes_tight(n=768)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(words): return rng.integers(0, 1 << 63, words, dtype=np.uint64)
eng = txhip.Engine(p, 0)
eng.load_eval_keys(rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p))))
eng.load_ctr_luts(rnd(10 * 3 * 16 * 2 * 3 * 1024))
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ct = rng.integers(0, 256, 16 * nb, dtype=np.uint8).tobytes()
for rep in range(6):
    eng.transcipher(ct, bytes(16))
    t = eng.timing()
    print(eng.last_blind_rotate_kernel(), "total %.2f ms; blind rotation main kernel %.3f ms per launch (%d launches)" % (t["ms_total"], t["ms_br_main"] / max(1, t["n_br_main"]), t["n_br_main"]), flush=True)
