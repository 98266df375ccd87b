#!/bin/bash
# Round-5 first GPU session: the GPU suite at the new tree, where the wall time of the two server stages goes
# (TX_STAGE_TIMING), counters of kernels #2 / #3 inside the pipeline, one bench line.
#   tools/r5_session1.sh <tag>
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${tag}_pytest.log 2>&1 || { tail -40 $O/${tag}_pytest.log; exit 1; }
tail -3 $O/${tag}_pytest.log
for size in 2 3; do
  for dv in 0 0,0; do
    W=$(mktemp -d /tmp/r5stageXXXX)
    TX_DEVICES=$dv TX_STAGE_TIMING=$O/${tag}_stage_timing_size${size}_dev${dv/,/_}.jsonl timeout -k 10 300 python temp-fhe-transciphering_amd/python/run_submission.py $size --root $W --seed 123 --fhe_seed 99 > $O/${tag}_stages_size${size}_dev${dv/,/_}.log 2>&1
    cp $W/measurements/*/results.json $O/${tag}_size${size}_dev${dv/,/_}_stage_executables_results.json
    rm -rf $W
  done
done
cat $O/${tag}_stage_timing_*.jsonl
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $O/${tag}_bench.log 2>&1; tail -1 $O/${tag}_bench.log > $O/${tag}_bench_line.json
bash tools/r3_pmc_pipeline.sh $tag trace_scheme_switch4 lut_eval
bash tools/profile_round.sh $tag > $O/${tag}_profile_round.txt 2>&1; tail -30 $O/${tag}_profile_round.txt
