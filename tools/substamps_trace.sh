#!/bin/bash
# exploratory: stamped builds of the trace loop with the stamps moved inside a transform
#   tools/substamps_trace.sh tag[:ENV=val,...] ...     (GEN_SUBSTAMPS = inv3 / fwd3; BR_ABL letters w / r / X skip exchange stores / loads / both)
set -e
cd "$(dirname "$0")/.."
C=temp-fhe-transciphering_amd/csrc
for spec in "$@"; do
  IFS=: read tag envs <<< "$spec"
  rm -rf build_exp/src_s$tag && cp -r $C build_exp/src_s$tag
  env $(echo $envs | tr ',' ' ') python tools/asmgen/trgen.py --out build_exp/src_s$tag
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -DTXD_STAMPS -DTXD_STAMPS_TRACE -Iinclude -o build_exp/libtx_st_$tag.so build_exp/src_s$tag/tx_api.hip &
done
wait
