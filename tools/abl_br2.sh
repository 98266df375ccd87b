#!/bin/bash
# exploratory: build variants of the paired blind-rotation loops into build_exp/libtx_br2<tag>.so
#   tools/abl_br2.sh tag[:ENV=val,ENV=val...][:extra hipcc flags] ...   (BR2_ABL / BR_ABL timing-only ablations: outputs invalid)
set -e
cd "$(dirname "$0")/.."
C=temp-fhe-transciphering_amd/csrc
for spec in "$@"; do
  IFS=: read tag envs flags <<< "$spec"
  rm -rf build_exp/src2_$tag && mkdir -p build_exp && cp -r $C build_exp/src2_$tag
  env $(echo $envs | tr ',' ' ') python tools/asmgen/brgen2.py --out build_exp/src2_$tag
done
for spec in "$@"; do
  IFS=: read tag envs flags <<< "$spec"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value $(echo $flags | tr ',' ' ') -Iinclude -o build_exp/libtx_br2$tag.so build_exp/src2_$tag/tx_api.hip 2> build_exp/br2$tag.err &
done
wait
ls build_exp/libtx_br2*.so
