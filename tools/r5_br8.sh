#!/bin/bash
# k_blind_rotate8 (soft barriers per unit inside the twelve-wave workgroup) against k_blind_rotate7 (lockstep): parity, then 1024-unit
# launches alone and inside the round pipeline (8 blocks, every round 1024 units), alternating, same box
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "every_blind_rotation_kernel" > $O/${tag}_pytest_br8.log 2>&1 || { tail -40 $O/${tag}_pytest_br8.log; exit 1; }
tail -2 $O/${tag}_pytest_br8.log
X=$R/temp-fhe-transciphering_amd/libtfhe_mi355x_exp.so
{
echo "## launches alone, eight in a row (ms), alternating variants"
for rep in 1 2; do
  for v in 7 8; do
    echo -n "variant $v, 1024 units: "; TX_LIB=$X TX_BR_VARIANT=$v timeout -k 10 120 python tools/br_only.py 768 1024 8 | grep "^blind_rotate" | awk '{printf "%s ", $4} END {print ""}'
  done
done
for v in 7 8; do
  echo -n "variant $v, 1000 units (ragged): "; TX_LIB=$X TX_BR_VARIANT=$v timeout -k 10 120 python tools/br_only.py 768 1000 4 | grep "^blind_rotate" | awk '{printf "%s ", $4} END {print ""}'
done
echo "## inside the round pipeline, 8 blocks, every round 1024 units (TX_CTR_SHARE=0)"
for rep in 1 2; do
  for v in 7 8; do
    echo "== variant $v"; TX_CTR_SHARE=0 TX_LIB=$X TX_BR_VARIANT=$v timeout -k 10 200 python tools/wall_time8.py 8 | tail -4
  done
done
} 2>&1 | tee $O/${tag}_blind_rotate8_soft_barriers_vs_lockstep.txt
