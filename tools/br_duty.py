"""exploratory: k_blind_rotate5 (8192 units) launched with a host-side pause between launches: does the kernel's time depend on the duty cycle
(power averaging)?  tools/br_duty.py [pause_ms ...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
import numpy as np, ctypes as C
import txhip, txclient
p = txhip.aes_tight(n=768)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(words): return rng.integers(0, 1 << 63, words, dtype=np.uint64)
eng = txhip.Engine(p, 0)
eng.load_eval_keys(rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p))))
lwe = rnd(8192 * 769).reshape(8192, 769)
for pause in [float(x) for x in sys.argv[1:]] or [0.0, 5.0, 15.0, 40.0, 0.0]:
    ts = []
    t0 = time.time()
    for r in range(40):
        eng.blind_rotate(lwe)
        ts.append(eng.timing()["ms_blind_rotate"])
        if pause: time.sleep(pause / 1e3)
    wall = (time.time() - t0) / 40 * 1e3
    print("pause %5.1f ms: kernel %.2f ms (last 20 launches: min %.2f max %.2f), wall per launch %.1f ms -> duty %.0f %%" %
          (pause, sum(ts[20:]) / 20, min(ts[20:]), max(ts[20:]), wall, 100 * sum(ts[20:]) / 20 / wall), flush=True)
