#!/bin/bash
# tools/r3_br6_abl.sh <tag> <count> lib-tag ...: time the cooperative blind rotation from diagnostic / ablation builds
tag=$1; cnt=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for t in "$@"; do
  TX_LIB=$R/build_exp/libtx_br3$t.so TX_BR_VARIANT=6 timeout -k 10 120 python tools/br_only.py 768 $cnt 3 2>&1 | grep -v "^raw\|inside forward" | sed "s/^/$t $cnt: /"
done > $O/${tag}_abl_$cnt.log 2>&1
cat $O/${tag}_abl_$cnt.log
