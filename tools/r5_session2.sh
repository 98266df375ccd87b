#!/bin/bash
# Round-5 second GPU session: GPU suite at the new tree; stage wall times (progressive pinning, reserve, fast exit);
# A/B of the non-temporal GGSW stores (build_exp/libtx_trnt0.so = SS_NT=0, libtx_trnt1.so = SS_NT=1, built by
# tools/abl_trace.sh before the call) with L2 hit counters; then the counter set of the tree as it is.
#   tools/r5_session2.sh <tag>
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${tag}_pytest.log 2>&1 || { tail -40 $O/${tag}_pytest.log; exit 1; }
tail -3 $O/${tag}_pytest.log
for size in 2 3; do
  W=$(mktemp -d /tmp/r5stageXXXX)
  TX_DEVICES=0 TX_STAGE_TIMING=$O/${tag}_stage_timing_size${size}.jsonl timeout -k 10 300 python temp-fhe-transciphering_amd/python/run_submission.py $size --root $W --seed 123 --fhe_seed 99 > $O/${tag}_stages_size${size}.log 2>&1
  cp $W/measurements/*/results.json $O/${tag}_size${size}_stage_executables_results.json
  rm -rf $W
done
cat $O/${tag}_stage_timing_*.jsonl
python - <<PY
import json
for s in (2, 3):
    d = json.load(open("$O/${tag}_size%d_stage_executables_results.json" % s))["per_stage"]
    print("size", s, {k: v for k, v in d.items() if k[0] in "78"})
PY
echo "== A/B: GGSW rows stored plain (nt0) / non-temporal (nt1); per-family average launch ms of one medium CTR transcipher"
for rep in 1 2 3; do
  for v in nt0 nt1; do
    echo -n "$v: "; TX_LIB=$R/build_exp/libtx_tr$v.so timeout -k 10 120 python tools/stage_times.py | tail -1
  done
done 2>&1 | tee $O/${tag}_ab_nontemporal_ggsw_stores.txt
cd /tmp && export TMPDIR=/tmp
for v in nt0 nt1; do
  mkdir -p $O/${tag}_tcc_$v
  TX_LIB=$R/build_exp/libtx_tr$v.so rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/${tag}_tcc_$v -o tcc --output-format csv -- python3 $R/tools/stage_times.py > $O/${tag}_tcc_$v.log 2>&1
  TX_LIB=$R/build_exp/libtx_tr$v.so rocprofv3 --pmc FETCH_SIZE -d $O/${tag}_fetch_$v -o fetch --output-format csv -- python3 $R/tools/stage_times.py > $O/${tag}_fetch_$v.log 2>&1
done
python3 - <<PY | tee -a $O/${tag}_ab_nontemporal_ggsw_stores.txt
import csv, glob
for v in ("nt0", "nt1"):
    for kind in ("tcc", "fetch"):
        for f in glob.glob("$O/${tag}_%s_%s/*counter_collection.csv" % (kind, v)) + glob.glob("$O/${tag}_%s_%s/*/*counter_collection.csv" % (kind, v)):
            acc = {}
            for r in csv.DictReader(open(f)):
                k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
                a = acc.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
            for (k, c), (s, n) in sorted(acc.items()):
                if "trace_scheme" in k or "lut_eval" in k or "blind_rotate5" in k: print(v, k, c, "per launch %.4g over %d launches" % (s / n, n))
PY
cd $R
bash tools/r3_pmc_pipeline.sh $tag blind_rotate5 trace_scheme_switch4 lut_eval > $O/${tag}_pmc_pipeline.txt 2>&1
bash tools/profile_round.sh $tag > $O/${tag}_profile_round.txt 2>&1; tail -12 $O/${tag}_profile_round.txt
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $O/${tag}_bench.log 2>&1; tail -1 $O/${tag}_bench.log > $O/${tag}_bench_line.json; tail -c 1500 $O/${tag}_bench_line.json
