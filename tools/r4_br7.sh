#!/bin/bash
# tools/r4_br7.sh <tag>: k_blind_rotate7 (four units per twelve-wave workgroup) -- parity against the one-wave kernel on full and ragged
# launches, launch times alone, twelve back to back, and inside the round pipeline with every round at 1024 units
set -e
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python - > $O/${tag}_parity.log 2>&1 <<'PY' || { tail -20 $O/${tag}_parity.log; exit 1; }
import sys, os
sys.path.insert(0, "temp-fhe-transciphering_amd/python")
import numpy as np, ctypes as C
import txhip, txclient
p = txhip.aes_tight(n=768)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(w): return rng.integers(0, 1 << 63, w, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, w, dtype=np.uint64)
keys = [rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p)))]
lwes = {cnt: np.random.default_rng(cnt).integers(0, 1 << 63, (cnt, 769), dtype=np.uint64) * np.uint64(2) for cnt in (1, 6, 13, 1021, 1024, 1500)}
out = {}
for v in ("0", "7", "6"):
    os.environ["TX_BR_VARIANT"] = v
    e = txhip.Engine(p, 0); e.load_eval_keys(*keys)
    for cnt, l in lwes.items(): out[(v, cnt)] = e.blind_rotate(l); assert e.last_blind_rotate_kernel() == {"0": "k_blind_rotate", "6": "k_blind_rotate6", "7": "k_blind_rotate7"}[v]
    e.close()
for cnt in lwes:
    s7, s6 = bool(np.array_equal(out[("0", cnt)], out[("7", cnt)])), bool(np.array_equal(out[("0", cnt)], out[("6", cnt)]))
    print("count", cnt, "k_blind_rotate7 == k_blind_rotate:", s7, " k_blind_rotate6 == k_blind_rotate:", s6)
    assert s7 and s6
PY
cat $O/${tag}_parity.log
for v in 7 6 0; do for c in 1024 768 512; do echo "variant $v, $c units: $(TX_BR_VARIANT=$v timeout -k 10 120 python tools/br_only.py 768 $c 8 2>&1 | grep '^blind_rotate' | awk '{printf "%s ", $4}')"; done; done > $O/${tag}_small.log 2>&1
cat $O/${tag}_small.log
for v in 7 6 0; do echo "== variant $v: pipeline, 8 blocks, every round 1024 units"; TX_CTR_SHARE=0 TX_BR_VARIANT=$v timeout -k 10 200 python tools/wall_time8.py 8 2>&1 | tail -4; done > $O/${tag}_wall8.log 2>&1
cat $O/${tag}_wall8.log
