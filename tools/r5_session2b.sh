#!/bin/bash
# the part of tools/r5_session2.sh behind the A/B timings: cache counters of the A/B builds, the counter set of the tree, a bench line
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in nt0 nt1; do
  TX_LIB=$R/build_exp/libtx_tr$v.so rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/${tag}_tcc_$v -o tcc --output-format csv -- python3 $R/tools/stage_times.py > $O/${tag}_tcc_$v.log 2>&1 || tail -5 $O/${tag}_tcc_$v.log
  TX_LIB=$R/build_exp/libtx_tr$v.so rocprofv3 --pmc FETCH_SIZE -d $O/${tag}_fetch_$v -o fetch --output-format csv -- python3 $R/tools/stage_times.py > $O/${tag}_fetch_$v.log 2>&1 || tail -5 $O/${tag}_fetch_$v.log
done
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -d $O/${tag}_tcp -o tcp --output-format csv -- python3 $R/tools/stage_times.py > $O/${tag}_tcp.log 2>&1 || tail -5 $O/${tag}_tcp.log
python3 - <<PY | tee $O/${tag}_cache_counters_trace_lut.txt
import csv, glob
for v in ("tcc_nt0", "fetch_nt0", "tcc_nt1", "fetch_nt1", "tcp"):
    for f in glob.glob("$O/${tag}_%s/*counter_collection.csv" % v) + glob.glob("$O/${tag}_%s/*/*counter_collection.csv" % v):
        acc = {}
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            a = acc.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
        for (k, c), (s, n) in sorted(acc.items()):
            if "trace_scheme" in k or "lut_eval" in k or "blind_rotate5" in k: print(v, k, c, "per launch %.5g over %d launches" % (s / n, n))
PY
cd $R
bash tools/r3_pmc_pipeline.sh $tag blind_rotate5 trace_scheme_switch4 lut_eval > $O/${tag}_pmc_pipeline.txt 2>&1 || { tail -20 $O/${tag}_pmc_pipeline.txt; exit 1; }
bash tools/profile_round.sh $tag > $O/${tag}_profile_round.txt 2>&1 || { tail -20 $O/${tag}_profile_round.txt; exit 1; }
tail -12 $O/${tag}_profile_round.txt
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $O/${tag}_bench.log 2>&1 || { tail -20 $O/${tag}_bench.log; exit 1; }
tail -1 $O/${tag}_bench.log > $O/${tag}_bench_line.json; tail -c 2500 $O/${tag}_bench_line.json
