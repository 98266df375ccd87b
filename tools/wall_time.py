"""exploratory: wall time of tx_transcipher on the medium CTR instance (random keys; results stay on the device), TX_LIB selects a build"""
import sys, os, time
sys.path.insert(0, "temp-fhe-transciphering_amd/python")
import numpy as np, txhip, txclient, ctypes as C, torch
if os.environ.get('TX_LIB'): txhip.LIB_PATH = os.environ['TX_LIB']
p = txhip.aes_tight(n=768)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(words): return rng.integers(0, 1 << 63, words, dtype=np.uint64)
eng = txhip.Engine(p, 0)
eng.load_eval_keys(rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p))))
eng.load_ctr_luts(rnd(10 * 3 * 16 * 2 * 3 * 1024))
ct = rng.integers(0, 256, 16 * 64, dtype=np.uint8).tobytes()
out = torch.empty((64 * 128, 2049), dtype=torch.int64, device="cuda:0")
eng.transcipher(ct, bytes(16), out_device_ptr=out.data_ptr())
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(3): eng.transcipher(ct, bytes(16), out_device_ptr=out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("wall ms per call: %.2f   sum of kernel events: %.2f" % (dt * 1e3, eng.timing()["ms_total"]), flush=True)
