#!/bin/bash
# a short GPU-box check of a kernel change:  tools/r4_quick.sh <tag>  -> GPU tests, the default bench line twice, the 8-block share
set -e
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${tag}_pytest.log 2>&1 || { tail -40 $O/${tag}_pytest.log; exit 1; }
tail -2 $O/${tag}_pytest.log
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-equal-work-legs > $O/${tag}_bench$i.log 2>&1 || { tail -20 $O/${tag}_bench$i.log; exit 1; }
  python - $O/${tag}_bench$i.log <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("blocks/s %.2f" % d["value"], d["roofline"]["kernel"], "%.2f ms" % d["roofline"]["avg_launch_ms"], d["kernel_ms_last_step"])
PY
done
python bench.py --blocks 8 --no-cpu-baseline --no-equal-work-legs --steps 5 > $O/${tag}_bench8.log 2>&1
python - $O/${tag}_bench8.log <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("8 blocks: blocks/s %.2f" % d["value"], d["roofline"]["kernel"], "%.2f ms" % d["roofline"]["avg_launch_ms"], d["kernel_ms_last_step"])
PY
