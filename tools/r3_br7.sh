#!/bin/bash
# tools/r3_br7.sh <tag>: the paired blind-rotation EXPERIMENT (TX_BR_VARIANT=7; needs a library with tools/asmgen/k_blind_rotate7.hip.txt pasted in, see there) -- parity against the one-wave kernel, then timings
set -e
tag=$1
# This tree does not build the experiment: the library validates TX_BR_VARIANT to 0..6 and has no variant-7 dispatch.  To rerun it,
# paste tools/asmgen/k_blind_rotate7.hip.txt into csrc/tx_br2.hpp, add the launch_br case and widen the knob range in tx_create
# (three edits, listed at the top of the .hip.txt), rebuild, and set R3_BR7_BUILT=1.  profiles/r3p_* / r3v_* / r3y_* came from such a build.
if [ "${R3_BR7_BUILT:-0}" != "1" ]; then
  echo "tools/r3_br7.sh: the paired blind rotation (variant 7) is not in the library of this tree; see the comment at the top of this script" >&2
  exit 3
fi
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python - > $O/${tag}_br7_parity.log 2>&1 <<'PY' || { tail -20 $O/${tag}_br7_parity.log; exit 1; }
import sys, os
sys.path.insert(0, "temp-fhe-transciphering_amd/python")
import numpy as np, txhip, txclient, ctypes as C
p = txhip.aes_tight(n=768)
rng = np.random.default_rng(0)
L = txclient.lib()
def rnd(w): return rng.integers(0, 1 << 63, w, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, w, dtype=np.uint64)
keys = [rnd(L.txc_bsk_words(C.byref(p))), rnd(L.txc_ksk_words(C.byref(p))), rnd(L.txc_auto_words(C.byref(p))), rnd(L.txc_ss_words(C.byref(p)))]
outs = {}
for v in ("0", "7"):
    os.environ["TX_BR_VARIANT"] = v
    e = txhip.Engine(p, 0); e.load_eval_keys(*keys)
    for cnt in (1, 13, 1024):
        lwe = np.random.default_rng(cnt).integers(0, 1 << 63, (cnt, 769), dtype=np.uint64) * np.uint64(2)
        outs[(v, cnt)] = e.blind_rotate(lwe)
        print(v, cnt, e.last_blind_rotate_kernel(), round(e.timing()["ms_blind_rotate"], 3), "ms", flush=True)
    e.close()
for cnt in (1, 13, 1024):
    print("count", cnt, "variant 7 == 0:", bool(np.array_equal(outs[("7", cnt)], outs[("0", cnt)])))
    assert np.array_equal(outs[("7", cnt)], outs[("0", cnt)])
PY
cat $O/${tag}_br7_parity.log
for v in 7 6 0; do for c in 1024 896 768 512; do TX_BR_VARIANT=$v timeout -k 10 120 python tools/br_only.py 768 $c 4 2>&1 | tail -2 | sed "s/^/variant $v: /"; done; done > $O/${tag}_br7_small.log 2>&1
cat $O/${tag}_br7_small.log
