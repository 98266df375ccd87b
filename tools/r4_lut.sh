#!/bin/bash
# tools/r4_lut.sh <tag>: LUT tree parity tests, its time in the pipeline and its fabric traffic (FETCH_SIZE pass over bench.py --steps 1)
set -e
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lut or ctr or schedule or stages or smoke or toy" > $O/${tag}_pytest_lut.log 2>&1 || { tail -30 $O/${tag}_pytest_lut.log; exit 1; }
tail -2 $O/${tag}_pytest_lut.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-equal-work-legs 2>&1 | grep "^{" | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('medium:', round(l['value'],2), 'blocks/s', l['kernel_ms_last_step'])"
python bench.py --mode ecb --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('medium ecb:', round(l['value'],2), 'blocks/s', l['kernel_ms_last_step'])"
python bench.py --blocks 8 --steps 5 --warmup 2 --no-cpu-baseline --no-equal-work-legs 2>&1 | grep "^{" | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('8 blocks:', round(l['value'],2), 'blocks/s', l['kernel_ms_last_step'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $O/${tag}_fetch -o fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-equal-work-legs > $O/${tag}_fetch.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/${tag}_fetch/*counter_collection.csv") + glob.glob("$O/${tag}_fetch/*/*counter_collection.csv"):
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        a = acc.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in acc.items():
        if "lut_eval" in k or "trace_scheme" in k: print(k, "launches", n, "FETCH_SIZE average per launch (KiB)", v / n, "-> x2 =", round(2 * v / n * 1024 / 1e9, 2), "GB")
PY
