#!/bin/bash
# tools/r4_br6.sh <tag>: the cooperative blind rotation (k_blind_rotate6) with exchange hi in registers -- GPU parity tests of
# the blind rotation, then launch times alone (tools/br_only.py) and inside the round pipeline (tools/wall_time8.py), against the
# one-wave kernel (TX_BR_VARIANT=0) on the same box
set -e
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blind_rot or cooperative or dispatch or smoke or partial" > $O/${tag}_pytest_br.log 2>&1 || { tail -30 $O/${tag}_pytest_br.log; exit 1; }
tail -2 $O/${tag}_pytest_br.log
for v in 6 0; do for c in 1024 768 512 128; do TX_BR_VARIANT=$v timeout -k 10 120 python tools/br_only.py 768 $c 4 2>&1 | tail -2 | sed "s/^/variant $v: /"; done; done > $O/${tag}_br_small.log 2>&1
cat $O/${tag}_br_small.log
for v in 6 0; do for nb in 8 6 4 1; do echo "== variant $v, $nb blocks"; TX_BR_VARIANT=$v timeout -k 10 200 python tools/wall_time8.py $nb 2>&1 | tail -3; done; done > $O/${tag}_wall8.log 2>&1
cat $O/${tag}_wall8.log
