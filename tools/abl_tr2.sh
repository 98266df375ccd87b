#!/bin/bash
# exploratory: build variants of the cooperative trace loop into build_exp/libtx_tr2<tag>.so
#   tools/abl_tr2.sh tag[:ENV=val,ENV=val...] ...     (TR2_ABL / BR_ABL timing-only ablations: outputs invalid)
set -e
cd "$(dirname "$0")/.."
C=temp-fhe-transciphering_amd/csrc
for spec in "$@"; do
  IFS=: read tag envs <<< "$spec"
  rm -rf build_exp/srct2_$tag && mkdir -p build_exp && cp -r $C build_exp/srct2_$tag
  env $(echo $envs | tr ',' ' ') python tools/asmgen/trgen2.py --out build_exp/srct2_$tag
done
for spec in "$@"; do
  IFS=: read tag envs <<< "$spec"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -Iinclude -o build_exp/libtx_tr2$tag.so build_exp/srct2_$tag/tx_api.hip 2> build_exp/tr2$tag.err &
done
wait
ls build_exp/libtx_tr2*.so
