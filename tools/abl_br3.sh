#!/bin/bash
# exploratory: build variants of the cooperative blind-rotation loop into build_exp/libtx_br3<tag>.so
#   tools/abl_br3.sh tag[:ENV=val,ENV=val...][:extra hipcc flags] ...
#   (BR3_ABL / BR_ABL timing-only ablations: outputs invalid; tag "st" with flags -DTXD_STAMPS,-DTXD_STAMPS_BR6 = phase stamps)
set -e
cd "$(dirname "$0")/.."
C=temp-fhe-transciphering_amd/csrc
for g in brgen trgen ssgen; do python tools/asmgen/$g.py --out $C > /dev/null; done     # the *_stamps.inc variants (untracked)
for spec in "$@"; do
  IFS=: read tag envs flags <<< "$spec"
  rm -rf build_exp/src3_$tag && mkdir -p build_exp && cp -r $C build_exp/src3_$tag
  env $(echo $envs | tr ',' ' ') python tools/asmgen/brgen3.py --out build_exp/src3_$tag
done
for spec in "$@"; do
  IFS=: read tag envs flags <<< "$spec"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value $(echo $flags | tr ',' ' ') -Iinclude -o build_exp/libtx_br3$tag.so build_exp/src3_$tag/tx_api.hip 2> build_exp/br3$tag.err &
done
wait
ls -la build_exp/libtx_br3*.so
