#!/bin/bash
# exploratory: stamped builds of the trace loop with the stamps moved inside a transform (GEN_SUBSTAMPS = inv3 / fwd3)
set -e
cd "$(dirname "$0")/.."
C=temp-fhe-transciphering_amd/csrc
for m in "$@"; do
  x=$m; [ "$m" = "phase" ] && x=""
  rm -rf build_exp/src_s$m && cp -r $C build_exp/src_s$m
  GEN_SUBSTAMPS=$x python tools/asmgen/trgen.py --out build_exp/src_s$m
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -DTXD_STAMPS -DTXD_STAMPS_TRACE -Iinclude -o build_exp/libtx_st_$m.so build_exp/src_s$m/tx_api.hip &
done
wait
