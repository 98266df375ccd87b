#!/bin/bash
# tools/r4_br6_sustained.sh <tag> lib-tag ...: 1024-unit launches of the cooperative blind rotation under sustained load: twelve
# launches back to back (tools/br_only.py) and inside the round pipeline with every launch at 1024 units (TX_CTR_SHARE=0)
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for t in "$@"; do
  echo "== $t: twelve 1024-unit launches"; TX_LIB=$R/build_exp/libtx_br3$t.so TX_BR_VARIANT=6 timeout -k 10 200 python tools/br_only.py 768 1024 12 2>&1 | grep "^blind_rotate" | awk '{printf "%s ", $4} END {print ""}'
  echo "== $t: pipeline, 8 blocks, every round 1024 units"; TX_CTR_SHARE=0 TX_LIB=$R/build_exp/libtx_br3$t.so TX_BR_VARIANT=6 timeout -k 10 200 python tools/wall_time8.py 8 2>&1 | tail -4
done > $O/${tag}_sustained.log 2>&1
echo "== one-wave kernel: pipeline, 8 blocks, every round 1024 units" >> $O/${tag}_sustained.log
TX_CTR_SHARE=0 TX_BR_VARIANT=0 timeout -k 10 200 python tools/wall_time8.py 8 2>&1 | tail -3 >> $O/${tag}_sustained.log
cat $O/${tag}_sustained.log
