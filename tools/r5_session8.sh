#!/bin/bash
# the 8-blocks-per-GPU share under rocprofv3: kernel stats and SQ / LDS counters of k_blind_rotate8 inside the pipeline
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
bash tools/pmc_kernel.sh gpurun_out/${tag}_pmc_blind_rotate8 blind_rotate8 bench.py --blocks 8 --steps 1 --warmup 1 --no-cpu-baseline --no-equal-work-legs --no-stage-times > $O/${tag}_pmc_blind_rotate8.txt 2>&1 || { tail -20 $O/${tag}_pmc_blind_rotate8.txt; exit 1; }
cat $O/${tag}_pmc_blind_rotate8.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${tag}_stats8 -o bench8 --output-format csv -- python3 $R/bench.py --blocks 8 --steps 5 --warmup 2 --no-cpu-baseline --no-equal-work-legs --no-stage-times > $O/${tag}_bench8_profiled.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/${tag}_stats8/*kernel_stats.csv") + glob.glob("$O/${tag}_stats8/*/*kernel_stats.csv"):
    for i, r in enumerate(csv.reader(open(f))):
        if i < 8: print(r[0].split("(")[0], r[1:5])
PY
