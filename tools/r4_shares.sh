#!/bin/bash
# tools/r4_shares.sh <tag>: the per-GPU shares of the medium instance on ONE GPU (64 / 32 / 16 / 8 blocks = the share at 1 / 2 / 4 / 8 GPUs):
# the compute side of strong scaling before the gather, projected from one box.  Also the large instance's share at 8 GPUs (64 blocks) = the first line.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for nb in 64 32 16 8 64; do
  timeout -k 10 300 python bench.py --blocks $nb --no-cpu-baseline --no-equal-work-legs --steps 4 --warmup 2 2>/dev/null | grep "^{" | python -c "
import sys, json
l = json.loads(sys.stdin.read()); r = l['roofline']
print('blocks', l['config']['blocks'], round(l['value'], 2), 'blocks/s', r['kernel'], round(r['avg_launch_ms'], 2), 'ms avg over', r['launches'], 'launches;', {k: v for k, v in l['kernel_ms_last_step'].items() if v})"
done > $O/${tag:-$1}_shares.txt 2>&1
cat $O/${tag:-$1}_shares.txt
