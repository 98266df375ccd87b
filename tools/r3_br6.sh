#!/bin/bash
# tools/r3_br6.sh <tag>: the cooperative blind rotation on the GPU -- parity tests first, then timings
set -e
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "blind_rotation or cooperative or ecb_schedule or stages_bit" > $O/${tag}_pytest_br.log 2>&1 || { tail -30 $O/${tag}_pytest_br.log; exit 1; }
tail -2 $O/${tag}_pytest_br.log
for v in 6 0; do for c in 1024 512 128; do TX_BR_VARIANT=$v timeout -k 10 120 python tools/br_only.py 768 $c 3 2>&1 | sed "s/^/variant $v: /"; done; done > $O/${tag}_br_small.log 2>&1
cat $O/${tag}_br_small.log
timeout -k 10 300 python bench.py --blocks 8 --no-cpu-baseline --steps 5 > $O/${tag}_bench8.log 2>&1; grep "^{" $O/${tag}_bench8.log | tail -1 > $O/${tag}_bench8_line.json
python - <<PY
import json
l = json.load(open("$O/${tag}_bench8_line.json"))
print("blocks 8:", round(l["value"], 2), "blocks/s", l["roofline"]["kernel"], round(l["roofline"]["avg_launch_ms"], 3), "ms", l["kernel_ms_last_step"])
PY
