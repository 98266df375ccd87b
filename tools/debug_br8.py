"""exploratory: k_blind_rotate8 (soft barriers) against k_blind_rotate7 on random inputs: which units / polynomials differ, and is it stable"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
import numpy as np, txhip, txclient, ctypes as C
if os.environ.get('TX_LIB'): txhip.EXP_LIB_PATH = os.environ['TX_LIB']
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = txhip.aes_tight(n=n)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(words): return rng.integers(0, 1 << 63, words, dtype=np.uint64)
keys = [rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p)))]
engs = {}
for v in ("7", "8"):
    os.environ["TX_BR_VARIANT"] = v
    e = txhip.Engine(p, 0, experiments=True)
    e.load_eval_keys(*keys)
    engs[v] = e
for count in (1, 13, 1024):
    lwe = rnd(count * (n + 1)).reshape(count, n + 1)
    a7 = engs["7"].blind_rotate(lwe)
    for rep in range(2):
        a8 = engs["8"].blind_rotate(lwe)
        bad = (a7 != a8)
        units = np.nonzero(bad.any(axis=(1, 2)))[0]
        print("count", count, "rep", rep, engs["8"].last_blind_rotate_kernel(), "units that differ:", len(units), units[:12].tolist(),
              "polys of the first:", bad[units[0]].any(axis=1).tolist() if len(units) else None,
              "coefficients:", int(bad[units[0]].sum()) if len(units) else 0, "ms", round(engs["8"].timing()["ms_blind_rotate"], 3), flush=True)
# one CMux step only
p1 = txhip.aes_tight(n=256)
