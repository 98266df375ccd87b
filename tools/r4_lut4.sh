#!/bin/bash
# tools/r4_lut4.sh <tag>: the generated LUT tree (k_lut_eval4) -- parity tests, then its time alone and in the pipeline against the hipcc kernel
set -e
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lut or ctr or schedule or stages or smoke or toy or max or inner" > $O/${tag}_pytest_lut.log 2>&1 || { tail -30 $O/${tag}_pytest_lut.log; exit 1; }
tail -2 $O/${tag}_pytest_lut.log
for v in 4 0; do echo "== TX_LUT_VARIANT=$v"; TX_LUT_VARIANT=$v python tools/lut_only.py 1024 3 | tail -2; TX_LUT_VARIANT=$v python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-equal-work-legs 2>&1 | grep "^{" | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('medium:', round(l['value'],2), 'blocks/s', l['kernel_ms_last_step'])"; TX_LUT_VARIANT=$v python bench.py --blocks 8 --steps 5 --warmup 2 --no-cpu-baseline --no-equal-work-legs 2>&1 | grep "^{" | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('8 blocks:', round(l['value'],2), 'blocks/s', l['kernel_ms_last_step'])"; done
