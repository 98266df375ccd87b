"""exploratory (stamps build, TX_LIB=build_exp/libtx_br3<tag>st.so): workgroup start / end times and the shader clock of k_blind_rotate6
INSIDE the round pipeline (the last blind rotation of a transcipher call over NB blocks, every round at full size)"""
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
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ct = rng.integers(0, 256, 16 * nb, dtype=np.uint8).tobytes()
for rep in range(3):
    eng.transcipher(ct, bytes(16))
    t = eng.timing()
    print(eng.last_blind_rotate_kernel(), "total %.2f ms; blind rotation %.3f ms per launch" % (t["ms_total"], t["ms_br_main"] / max(1, t["n_br_main"])), flush=True)
    cnt = nb * 128
    nw = min(4096, cnt)
    buf = (C.c_ulonglong * (2 * nw))()
    txhip.lib().tx_debug_wave_times(C.c_void_p(eng.h), buf, 2 * nw)
    a = np.array(buf[:], np.uint64).reshape(-1, 2).astype(np.int64)
    a = a[a[:, 1] > a[:, 0]]
    t0 = a[:, 0].min()
    life = (a[:, 1] - a[:, 0]) * 1e-5; end = (a[:, 1] - t0) * 1e-5; start = (a[:, 0] - t0) * 1e-5
    print("  workgroups %d: lifetime min/median/max %.2f / %.2f / %.2f ms; last start %.2f ms, last end %.2f ms" % (a.shape[0], life.min(), np.median(life), life.max(), start.max(), end.max()))
    print("  starts per ms:", np.histogram(start, bins=np.arange(0, end.max() + 1, 1.0))[0].tolist())
    print("  ends per ms:  ", np.histogram(end, bins=np.arange(0, end.max() + 1, 1.0))[0].tolist())
    st = (C.c_ulonglong * 16)()
    txhip.lib().tx_debug_stamps(C.c_void_p(eng.h), st)
    steps = max(1, st[6])
    cyc = sum(st[:6])
    print("  stamped wave: %d cycles over its unit (%d per step, %d of them waiting at barriers); with the median lifetime: %.2f GHz" % (cyc, cyc // steps, st[5] // steps, cyc / (np.median(life) * 1e-3) * 1e-9))
