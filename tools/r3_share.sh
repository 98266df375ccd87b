#!/bin/bash
# tools/r3_share.sh <tag>: CTR column sharing -- parity tests, then bench with and without
set -e
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ctr or blind_rotation or cooperative or medium or large or chunked" > $O/${tag}_pytest_ctr.log 2>&1 || { tail -30 $O/${tag}_pytest_ctr.log; exit 1; }
tail -2 $O/${tag}_pytest_ctr.log
for f in "" "--no-ctr-share"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 $f > $O/${tag}_bench$f.log 2>&1; grep "^{" $O/${tag}_bench$f.log | tail -1 > $O/${tag}_bench${f}_line.json
  python - <<PY
import json
l = json.load(open("$O/${tag}_bench${f}_line.json"))
r = l["roofline"]
print("bench $f:", round(l["value"], 2), "blocks/s", r["kernel"], "avg", round(r["avg_launch_ms"], 3), "ms over", r["launches"], "launches of", r["avg_units_per_launch"], "units; frac", round(r["frac"], 6), l["kernel_ms_last_step"])
PY
done
timeout -k 10 300 python bench.py --blocks 8 --no-cpu-baseline --steps 5 > $O/${tag}_bench8.log 2>&1; grep "^{" $O/${tag}_bench8.log | tail -1 > $O/${tag}_bench8_line.json
python - <<PY
import json
l = json.load(open("$O/${tag}_bench8_line.json"))
print("blocks 8:", round(l["value"], 2), "blocks/s", l["roofline"]["kernel"], round(l["roofline"]["avg_launch_ms"], 3), "ms", l["kernel_ms_last_step"])
PY
