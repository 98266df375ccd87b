#!/bin/bash
# tools/r4_br6_variants.sh <tag>: the cooperative blind rotation with its tile exchanges through LDS / in registers (build_exp/libtx_br3{ll,rl,rr}.so
# from tools/abl_br3.sh: exchange hi, exchange lo = l(ds) / r(egisters)): parity against the one-wave kernel, launch times alone and
# in the round pipeline, phase stamps, and SQ / LDS counter passes over the 1024-unit launch of each
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
lwes = {cnt: np.random.default_rng(cnt).integers(0, 1 << 63, (cnt, 769), dtype=np.uint64) * np.uint64(2) for cnt in (13, 1024)}
os.environ["TX_BR_VARIANT"] = "0"
e = txhip.Engine(p, 0); e.load_eval_keys(*keys)
ref = {cnt: e.blind_rotate(l) for cnt, l in lwes.items()}
e.close()
print("reference: k_blind_rotate (one wave per unit), library", txhip.LIB_PATH)
PY
for t in ll rl rr; do
TX_LIB=$R/build_exp/libtx_br3$t.so timeout -k 10 300 python - $t >> $O/${tag}_parity.log 2>&1 <<'PY' || { tail -20 $O/${tag}_parity.log; exit 1; }
import sys, os
sys.path.insert(0, "temp-fhe-transciphering_amd/python")
import numpy as np, ctypes as C
import txhip, txclient
txhip.LIB_PATH = os.environ["TX_LIB"]
p = txhip.aes_tight(n=768)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(w): return rng.integers(0, 1 << 63, w, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, w, dtype=np.uint64)
keys = [rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p)))]
lwes = {cnt: np.random.default_rng(cnt).integers(0, 1 << 63, (cnt, 769), dtype=np.uint64) * np.uint64(2) for cnt in (13, 1024)}
out = {}
for v in ("0", "6"):
    os.environ["TX_BR_VARIANT"] = v
    e = txhip.Engine(p, 0); e.load_eval_keys(*keys)
    for cnt, l in lwes.items(): out[(v, cnt)] = e.blind_rotate(l)
    e.close()
for cnt in lwes:
    same = bool(np.array_equal(out[("0", cnt)], out[("6", cnt)]))
    print("variant", sys.argv[1], "count", cnt, "k_blind_rotate6 == k_blind_rotate:", same)
    assert same
PY
done
cat $O/${tag}_parity.log
for t in ll rl rr; do for c in 1024 768 512 128; do TX_LIB=$R/build_exp/libtx_br3$t.so TX_BR_VARIANT=6 timeout -k 10 120 python tools/br_only.py 768 $c 4 2>&1 | tail -2 | sed "s/^/$t: /"; done; done > $O/${tag}_small.log 2>&1
cat $O/${tag}_small.log
for t in ll rl rr; do for nb in 8 6; do echo "== $t, $nb blocks"; TX_LIB=$R/build_exp/libtx_br3$t.so TX_BR_VARIANT=6 timeout -k 10 200 python tools/wall_time8.py $nb 2>&1 | tail -2; done; done > $O/${tag}_wall8.log 2>&1
echo "== one-wave kernel, 8 blocks" >> $O/${tag}_wall8.log; TX_BR_VARIANT=0 timeout -k 10 200 python tools/wall_time8.py 8 2>&1 | tail -2 >> $O/${tag}_wall8.log
cat $O/${tag}_wall8.log
for t in llst rlst rrst; do for c in 1024 128; do TX_LIB=$R/build_exp/libtx_br3$t.so TX_BR_VARIANT=6 timeout -k 10 120 python tools/br_only.py 768 $c 3 2>&1 | grep -v "^raw\|inside forward" | tail -6 | sed "s/^/$t $c: /"; done; done > $O/${tag}_stamps.log 2>&1
cat $O/${tag}_stamps.log
for t in ll rl rr; do
  TX_LIB=$R/build_exp/libtx_br3$t.so TX_BR_VARIANT=6 bash tools/pmc_kernel.sh gpurun_out/${tag}_pmc_$t blind_rotate6 tools/br_only.py 768 1024 2 > $O/${tag}_pmc_$t.txt 2>&1
  cat $O/${tag}_pmc_$t.txt
done
