#!/bin/bash
# exploratory: build variants of the generated blind-rotation loops into build_exp/libtx_br<tag>.so
#   tools/abl_br.sh tag[:ENV=val,ENV=val...] ...   (BR_ABL timing-only ablations: outputs invalid; BR_FAIR_H, BR_PRIO, ...)
set -e
cd "$(dirname "$0")/.."
C=temp-fhe-transciphering_amd/csrc
for spec in "$@"; do
  IFS=: read tag envs <<< "$spec"
  rm -rf build_exp/src_$tag && cp -r $C build_exp/src_$tag
  env $(echo $envs | tr ',' ' ') python tools/asmgen/brgen.py --out build_exp/src_$tag
done
for spec in "$@"; do
  IFS=: read tag envs <<< "$spec"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -Iinclude -o build_exp/libtx_br$tag.so build_exp/src_$tag/tx_api.hip &
done
wait
