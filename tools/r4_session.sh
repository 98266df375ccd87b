#!/bin/bash
# one GPU-box session of round 4:  tools/r4_session.sh <tag> [tests|notests]   -> gpurun_out/<tag>_*  and gpurun_out/prof_<tag>/
# GPU tests, the default bench line (with the equal-work legs), the mini-workload leg, the 8-blocks-per-GPU share, then the profile
# set of tools/profile_round.sh and SQ / LDS counter passes over k_blind_rotate5 as it runs in `bench.py --steps 1`.
# tools/roofline_inputs.py <tag> (run in the repo afterwards) turns the counter CSVs into profiles/roofline_inputs.json.
set -e
tag=$1; mode=${2:-tests}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$mode" = "tests" ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/${tag}_pytest.log 2>&1 || { tail -40 $O/${tag}_pytest.log; exit 1; }
  tail -3 $O/${tag}_pytest.log
fi
python bench.py > $O/${tag}_bench.log 2>&1 || { tail -20 $O/${tag}_bench.log; exit 1; }
grep "^{" $O/${tag}_bench.log | tail -1 > $O/${tag}_bench_line.json; cat $O/${tag}_bench_line.json
python bench.py --with-mini-workloads --no-cpu-baseline --no-equal-work-legs > $O/${tag}_bench_mini.log 2>&1 || { tail -20 $O/${tag}_bench_mini.log; exit 1; }
grep "^{" $O/${tag}_bench_mini.log | tail -1 > $O/${tag}_bench_mini_line.json; cat $O/${tag}_bench_mini_line.json
python bench.py --blocks 8 --no-cpu-baseline --no-equal-work-legs --steps 5 > $O/${tag}_bench8.log 2>&1; grep "^{" $O/${tag}_bench8.log | tail -1 > $O/${tag}_bench8_line.json; cat $O/${tag}_bench8_line.json
bash tools/profile_round.sh $tag > $O/${tag}_profile_round.txt 2>&1 || { tail -20 $O/${tag}_profile_round.txt; exit 1; }
tail -30 $O/${tag}_profile_round.txt
bash tools/r3_pmc_pipeline.sh $tag blind_rotate5
