#!/bin/bash
# soak under both decomposer tie rules: 40 key seeds x (ECB + CTR) medium instances each (5.9 M bootstraps per rule)
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 500 python tools/soak.py 100 140 64 > $O/${tag}_soak_40_instances_balanced_rule.txt 2>&1 || true
tail -2 $O/${tag}_soak_40_instances_balanced_rule.txt
timeout -k 10 500 python tools/soak.py 100 140 64 tfhe > $O/${tag}_soak_40_instances_tfhe_rs_rule.txt 2>&1 || true
tail -2 $O/${tag}_soak_40_instances_tfhe_rs_rule.txt
grep -c "False" $O/${tag}_soak_40_instances_tfhe_rs_rule.txt || true
