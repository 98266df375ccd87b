"""exploratory (stamps build): which context makes some workgroups of a 1024-unit k_blind_rotate6 launch start late?"""
import sys, os
sys.path.insert(0, "temp-fhe-transciphering_amd/python")
import numpy as np, txhip, txclient, ctypes as C
txhip.LIB_PATH = os.environ["TX_LIB"]
os.environ["TX_CTR_SHARE"] = "0"
p = txhip.aes_tight(n=768)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(words): return rng.integers(0, 1 << 63, words, dtype=np.uint64)
eng = txhip.Engine(p, 0)
eng.load_eval_keys(rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p))))
eng.load_ctr_luts(rnd(10 * 3 * 16 * 2 * 3 * 1024))
prev = [np.zeros((4096, 8), np.uint32)]
def placement():
    tab = np.zeros((4096, 8), np.uint32)
    txhip.lib().tx_debug_fair_table(C.c_void_p(eng.h), tab.ctypes.data_as(C.c_void_p))
    d = (tab[:, 4] - prev[0][:, 4]).astype(np.int64)
    prev[0] = tab
    used = np.nonzero(d)[0]
    per_xcc = [int(d[x * 256:(x + 1) * 256].sum()) for x in range(16) if d[x * 256:(x + 1) * 256].sum()]
    cus_xcc = [int((d[x * 256:(x + 1) * 256] > 0).sum()) for x in range(16) if d[x * 256:(x + 1) * 256].sum()]
    return "CUs used %d, workgroups per CU histogram %s, per XCC %s on %s CUs" % (used.size, np.bincount(d[used]).tolist(), per_xcc, cus_xcc)
def report(what):
    nw = 1024
    buf = (C.c_ulonglong * (2 * nw))()
    txhip.lib().tx_debug_wave_times(C.c_void_p(eng.h), buf, 2 * nw)
    a = np.array(buf[:], np.uint64).reshape(-1, 2).astype(np.int64)
    a = a[a[:, 1] > a[:, 0]]
    t0 = a[:, 0].min()
    start = (a[:, 0] - t0) * 1e-5; end = (a[:, 1] - t0) * 1e-5
    late = np.nonzero(start > 1.0)[0]
    print("%-46s BR %.3f ms; %d workgroups, %d start late (units %s), last end %.2f ms" % (what, eng.timing()["ms_blind_rotate"] / max(1, eng.timing()["n_blind_rotate"]), a.shape[0], late.size, late[:16].tolist(), end.max()), flush=True)
    print("      ", placement(), flush=True)
lwe = rnd(1024 * 769).reshape(1024, 769)
big = rnd(1024 * 2049).reshape(1024, 2049)
for rep in range(2):
    eng.blind_rotate(lwe); report("stage blind_rotate (fresh)")
eng.circuit_bootstrap(big); report("stage circuit_bootstrap: KS + BR + trace/SS")
eng.blind_rotate(lwe); report("stage blind_rotate after circuit_bootstrap")
ct = rng.integers(0, 256, 16 * 8, dtype=np.uint8).tobytes()
eng.transcipher(ct, bytes(16)); report("transcipher 8 blocks (last BR of 9)")
eng.blind_rotate(lwe); report("stage blind_rotate after transcipher")
eng.blind_rotate(lwe); report("stage blind_rotate again")
eng.lwe_keyswitch(big); eng.blind_rotate(lwe); report("stage blind_rotate after stage keyswitch")
acc = rnd(146 * 3072).reshape(146, 3, 1024)
eng.extract_trace(acc); eng.blind_rotate(lwe); report("stage blind_rotate after stage extract_trace")
