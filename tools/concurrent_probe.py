"""exploratory: does the power-bound blind rotation run faster when part of the chip runs the (less power-hungry) trace kernel
beside it?  Two contexts (two streams) on one device, two Python threads: k_blind_rotate5 capped to BR_WGS persistent workgroups
(TX_BR5_WGS) and the non-persistent k_extract_trace4 filling the CUs it leaves.
    python tools/concurrent_probe.py <br_wgs> <br_units> <trace_accs>"""
import sys, os, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
import numpy as np
import ctypes as C
br_wgs, br_units, tr_accs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
import txhip, txclient
p = txhip.aes_tight(n=768)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(w): return rng.integers(0, 1 << 63, w, dtype=np.uint64)
keys = [rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p)))]
os.environ["TX_BR5_WGS"] = str(br_wgs); os.environ["TX_BR_VARIANT"] = "5"
e_br = txhip.Engine(p, 0); e_br.load_eval_keys(*keys)
del os.environ["TX_BR5_WGS"]; del os.environ["TX_BR_VARIANT"]
e_tr = txhip.Engine(p, 0); e_tr.load_eval_keys(*keys)
lwe = rnd(br_units * 769).reshape(br_units, 769)
acc = rnd(tr_accs * 3072).reshape(tr_accs, 3, 1024)
def run_br(out): e_br.blind_rotate(lwe); out["br"] = e_br.timing()["ms_blind_rotate"]
def run_tr(out): e_tr.extract_trace(acc); out["tr"] = e_tr.timing()["ms_trace"]
for rep in range(3):
    o = {}
    run_br(o); run_tr(o)
    print("alone:      blind rotation %d units on %d CUs %.2f ms; trace %d units %.2f ms" % (br_units, br_wgs, o["br"], tr_accs * 7, o["tr"]), flush=True)
    o = {}
    t0 = time.perf_counter()
    th = [threading.Thread(target=run_br, args=(o,)), threading.Thread(target=run_tr, args=(o,))]
    for t in th: t.start()
    for t in th: t.join()
    print("concurrent: blind rotation %.2f ms; trace %.2f ms; wall %.2f ms" % (o["br"], o["tr"], (time.perf_counter() - t0) * 1e3), flush=True)
