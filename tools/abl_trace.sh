#!/bin/bash
# exploratory: build variants of the generated trace loop into build_exp/libtx_tr<tag>.so
#   tools/abl_trace.sh tag[:ENV=val,ENV=val...[:-Dflag,-Dflag]] ...   (TR_ABL / BR_ABL timing-only ablations: outputs invalid; TR_SKEW etc.)
set -e
cd "$(dirname "$0")/.."
C=temp-fhe-transciphering_amd/csrc
for spec in "$@"; do
  IFS=: read tag envs flags <<< "$spec"
  rm -rf build_exp/src_$tag && cp -r $C build_exp/src_$tag
  env $(echo $envs | tr ',' ' ') python tools/asmgen/trgen.py --out build_exp/src_$tag
  env $(echo $envs | tr ',' ' ') python tools/asmgen/ssgen.py --out build_exp/src_$tag
done
for spec in "$@"; do
  IFS=: read tag envs flags <<< "$spec"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value $(echo $flags | tr ',' ' ') -Iinclude -o build_exp/libtx_tr$tag.so build_exp/src_$tag/tx_api.hip &
done
wait
