#!/bin/bash
# final GPU session of round 5: the suite, smoke, the driver's bench shape, the 8-block share, soaks of the small-launch kernels
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/${tag}_pytest.log 2>&1 || { tail -40 $O/${tag}_pytest.log; exit 1; }
tail -2 $O/${tag}_pytest.log
timeout -k 10 300 python __graft_entry__.py smoke > $O/${tag}_smoke.log 2>&1 || { tail -20 $O/${tag}_smoke.log; exit 1; }
tail -1 $O/${tag}_smoke.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/${tag}_bench.log 2>&1 || { tail -20 $O/${tag}_bench.log; exit 1; }
tail -1 $O/${tag}_bench.log > $O/${tag}_bench_line_driver_shape.json
python - <<PY
import json
l = json.load(open("$O/${tag}_bench_line_driver_shape.json"))
print("value", l["value"], "ms_per_step", l["ms_per_step"], "host", l["value_host_result"], "stage_s", {k: l["stage_s"][k] for k in ("aes_decryption", "compute", "first_run", "verified")})
PY
timeout -k 10 300 python bench.py --blocks 8 --steps 10 --warmup 3 --no-cpu-baseline --no-stage-times > $O/${tag}_bench8.log 2>&1 || { tail -20 $O/${tag}_bench8.log; exit 1; }
tail -1 $O/${tag}_bench8.log > $O/${tag}_bench_line_8_blocks.json
python -c "import json; l=json.load(open('$O/${tag}_bench_line_8_blocks.json')); print('8 blocks:', l['value'], l['roofline']['kernel'], l['roofline']['avg_launch_ms'])"
for nb in 8 7 6 3; do timeout -k 10 200 python tools/soak.py 200 206 $nb > $O/${tag}_soak_${nb}_blocks.txt 2>&1 || { tail -5 $O/${tag}_soak_${nb}_blocks.txt; exit 1; }; tail -1 $O/${tag}_soak_${nb}_blocks.txt; done
