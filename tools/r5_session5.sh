#!/bin/bash
# counters of the tree as it is (the generated streams changed: tie-mode operand): SQ / LDS passes of the three kernels inside the
# pipeline, kernel stats, FETCH / WRITE passes -> tools/roofline_inputs.py <tag>
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
bash tools/r3_pmc_pipeline.sh $tag blind_rotate5 trace_scheme_switch4 lut_eval > $O/${tag}_pmc_pipeline.txt 2>&1 || { tail -20 $O/${tag}_pmc_pipeline.txt; exit 1; }
bash tools/profile_round.sh $tag > $O/${tag}_profile_round.txt 2>&1 || { tail -20 $O/${tag}_profile_round.txt; exit 1; }
tail -12 $O/${tag}_profile_round.txt
timeout -k 10 300 python tools/soak.py 0 6 > $O/${tag}_soak_medium.txt 2>&1 || { tail -5 $O/${tag}_soak_medium.txt; exit 1; }
tail -3 $O/${tag}_soak_medium.txt
timeout -k 10 200 python tools/soak.py 0 6 8 > $O/${tag}_soak_8_blocks.txt 2>&1 || { tail -5 $O/${tag}_soak_8_blocks.txt; exit 1; }
tail -2 $O/${tag}_soak_8_blocks.txt
