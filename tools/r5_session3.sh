#!/bin/bash
# Round-5 third GPU session: GPU suite, the driver's bench command shape, the 8-blocks-per-GPU share
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/${tag}_pytest.log 2>&1 || { tail -40 $O/${tag}_pytest.log; exit 1; }
tail -3 $O/${tag}_pytest.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/${tag}_bench.log 2>&1 || { tail -20 $O/${tag}_bench.log; exit 1; }
tail -1 $O/${tag}_bench.log > $O/${tag}_bench_line_driver_shape.json
python - <<PY
import json
l = json.load(open("$O/${tag}_bench_line_driver_shape.json"))
print("value", l["value"], "ms_per_step", l["ms_per_step"], "stage_s", l["stage_s"], "cpu", l["cpu_baseline"]["value"], l["cpu_baseline"]["sample"])
print([(k["kernel"], k["ms_per_step"], k["vector_pipe_busy"], k["traffic"]) for k in l["roofline"]["kernels"]])
PY
timeout -k 10 300 python bench.py --blocks 8 --steps 10 --warmup 3 --no-cpu-baseline --no-stage-times > $O/${tag}_bench8.log 2>&1 || { tail -20 $O/${tag}_bench8.log; exit 1; }
tail -1 $O/${tag}_bench8.log > $O/${tag}_bench_line_8_blocks.json
python -c "import json; l=json.load(open('$O/${tag}_bench_line_8_blocks.json')); print('8 blocks:', l['value'], l['roofline']['kernel'], l['roofline']['avg_launch_ms'])"
