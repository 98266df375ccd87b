#!/bin/bash
# one GPU-box session of round 3:  tools/r3_session.sh <tag>   -> gpurun_out/<tag>_*
# GPU tests, the default bench line, the 8-blocks-per-GPU share, the small-launch blind rotation alone, and PMC passes over
# the fused trace + scheme-switching kernel and the LUT tree as they run in the pipeline
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > $O/${tag}_pytest.log 2>&1 || { tail -30 $O/${tag}_pytest.log; exit 1; }
tail -3 $O/${tag}_pytest.log
python bench.py > $O/${tag}_bench.log 2>&1; grep "^{" $O/${tag}_bench.log | tail -1 > $O/${tag}_bench_line.json; cat $O/${tag}_bench_line.json
python bench.py --blocks 8 --no-cpu-baseline --steps 5 > $O/${tag}_bench8.log 2>&1; grep "^{" $O/${tag}_bench8.log | tail -1 > $O/${tag}_bench8_line.json; cat $O/${tag}_bench8_line.json
python tools/br_only.py 768 1024 3 > $O/${tag}_br1024.log 2>&1; cat $O/${tag}_br1024.log
