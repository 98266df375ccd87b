"""exploratory: time the LUT tree alone on the medium instance's shape (1024 bytes, CTR tables: 3 tables x 2 accumulators per byte); TX_LIB selects a build"""
import sys, os
sys.path.insert(0, "temp-fhe-transciphering_amd/python")
import numpy as np, txhip, txclient, ctypes as C
if os.environ.get('TX_LIB'): txhip.LIB_PATH = os.environ['TX_LIB']
p = txhip.aes_tight(n=768)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(words): return rng.integers(0, 1 << 63, words, dtype=np.uint64)
eng = txhip.Engine(p, 0)
eng.load_eval_keys(rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p))))
eng.load_ctr_luts(rnd(10 * 3 * 16 * 2 * 3 * 1024))
nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nlut = int(sys.argv[2]) if len(sys.argv) > 2 else 3
gg = rng.standard_normal(nbytes * 8 * 7 * 3 * 3 * 1024)
for rep in range(4):
    eng.lut_eval(gg, 1, 3, nlut)
    print("lut_eval", nbytes, "bytes x", nlut, "tables:", round(eng.timing()["ms_lut_eval"], 3), "ms", flush=True)
if hasattr(txhip.lib(), "tx_debug_stamps"):        # -DTXD_STAMPS -DTXD_STAMPS_LUT build: cycles per phase and CMux step of one wave
    st = (C.c_ulonglong * 16)()
    txhip.lib().tx_debug_stamps(C.c_void_p(eng.h), st)
    names = ["difference + digits", "forward", "multiply-accumulate", "inverse x3", "round + update"]
    steps = max(1, st[6])
    print("stamps (cycles per CMux step):", {n: round(st[i] / steps) for i, n in enumerate(names)}, "sum", round(sum(st[:5]) / steps))
