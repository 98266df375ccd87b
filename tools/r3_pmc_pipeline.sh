#!/bin/bash
# PMC passes over kernels AS THEY RUN IN THE PIPELINE (bench.py, one step): tools/r3_pmc_pipeline.sh <tag> <kernel substring> ...
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
for pat in "$@"; do
  bash $R/tools/pmc_kernel.sh gpurun_out/${tag}_pmc_$pat $pat bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-equal-work-legs --no-stage-times > $R/gpurun_out/${tag}_pmc_$pat.txt 2>&1
  cat $R/gpurun_out/${tag}_pmc_$pat.txt
done
