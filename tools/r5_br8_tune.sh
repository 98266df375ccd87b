#!/bin/bash
# k_blind_rotate8 tuning: poll period, fairness thresholds; inside the round pipeline (8 blocks, every round 1024 units), alternating
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
{
for rep in 1 2; do
  for v in s1 s2 s4 nf l124 l4816; do
    echo -n "$v: "; TX_CTR_SHARE=0 TX_LIB=$R/build_exp/libtx_br3$v.so TX_BR_VARIANT=8 timeout -k 10 200 python tools/wall_time8.py 8 | tail -3 | awk '{printf "%s ", $9} END {print ""}'
  done
done
echo "## 768 and 896 units per launch (6 and 7 blocks), variant 8 against the dispatch's choice"
for nb in 6 7; do
  for v in -1 8; do
    echo -n "blocks $nb variant $v: "; TX_CTR_SHARE=0 TX_LIB=$R/build_exp/libtx_br3s1.so TX_BR_VARIANT=$v timeout -k 10 200 python tools/wall_time8.py $nb | tail -2 | awk '{printf "%s %s ", $1, $9} END {print ""}'
  done
done
} 2>&1 | tee $O/${tag}_blind_rotate8_tuning.txt
