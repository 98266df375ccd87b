"""exploratory: time the scheme-switching kernel alone; with a -DTXD_STAMPS -DTXD_STAMPS_SS build (TX_LIB) also cycles per phase"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
import numpy as np
import txhip, txclient
if os.environ.get('TX_LIB'): txhip.LIB_PATH = os.environ['TX_LIB']
import ctypes as C
count = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
p = txhip.aes_tight(n=768)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(words): return rng.integers(0, 1 << 63, words, dtype=np.uint64)
eng = txhip.Engine(p, 0)
eng.load_eval_keys(rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p))))
glev = rnd(count * 7 * 3072)
for r in range(reps):
    eng.scheme_switch_fourier(glev)
    print("scheme_switch", count * 7, "units:", round(eng.timing()["ms_scheme_switch"], 3), "ms", flush=True)
if hasattr(txhip.lib(), "tx_debug_stamps"):
    st = (C.c_ulonglong * 16)()
    txhip.lib().tx_debug_stamps(C.c_void_p(eng.h), st)
    print("raw stamps:", [int(st[i]) for i in range(6)], "(digits+2 fwd3+park | key passes | inverses | convert+fwd3+store rows 0,1 | body row)")
