#!/bin/bash
# Round profile set, run on the GPU box:  tools/profile_round.sh <tag>   (writes gpurun_out/prof_<tag>/)
#   1. rocprofv3 --kernel-trace --stats over the default bench command (per-kernel average launch times)
#   2. separate --pmc passes FETCH_SIZE / WRITE_SIZE over the blind rotation alone (8192 units) and over the bench command
#      itself (one step): the average per k_blind_rotate5 launch of the command's own launch mix (CTR: eight rounds of 8192
#      units and the shared second round's 2048) is bench.py's `traffic`
# Counters are collected with --kernel-trace only (no sys/hip/hsa trace domains), as the pool requires.
set -e
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/stats -o bench --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-equal-work-legs --no-stage-times > $out/bench_profiled.log 2>&1
grep "^{" $out/bench_profiled.log | tail -1 > $out/bench_line_profiled.json
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o fetch --output-format csv -- python3 $R/tools/br_only.py 768 8192 1 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -o write --output-format csv -- python3 $R/tools/br_only.py 768 8192 1 > $out/write.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/fetch_bench -o fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-equal-work-legs --no-stage-times > $out/fetch_bench.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write_bench -o write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-equal-work-legs --no-stage-times > $out/write_bench.log 2>&1
python3 - <<PY
import csv, glob
for name in ("fetch_bench", "write_bench"):
    for f in glob.glob("$out/%s/*counter_collection.csv" % name) + glob.glob("$out/%s/*/*counter_collection.csv" % name):
        acc = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            a = acc.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in acc.items():
            if "blind_rotate" in k or "trace_scheme" in k or "lut_eval" in k: print(name, k, "launches", n, "sum", v, "average per launch", v / n)
for f in sorted(glob.glob("$out/stats/*kernel_stats.csv") + glob.glob("$out/stats/*/*kernel_stats.csv")):
    print(f)
    for i, r in enumerate(csv.reader(open(f))):
        if i < 12: print("  ", r[:6])
for name in ("fetch", "write"):
    for f in glob.glob("$out/%s/*counter_collection.csv" % name) + glob.glob("$out/%s/*/*counter_collection.csv" % name):
        for r in csv.DictReader(open(f)):
            if "blind_rotate" in r["Kernel_Name"]:
                print(name, r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Counter_Value"])
PY
