#!/bin/bash
# Builds libbscnav.so (gfx950 only) next to the Python package.  -ffp-contract=off: every fma in the
# geometry chain is explicit (__fma_rn); nothing else may be fused (bit-exact parity with the reference).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
# BSC_OUT / BSC_OBJ: an A/B build (other BSC_EXTRA_FLAGS) beside the product; python picks it up through BSC_LIB_PATH
OUT="${BSC_OUT:-$HERE/../libbscnav.so}"
OBJ="${BSC_OBJ:-$HERE/_obj}"
mkdir -p "$OBJ"
FLAGS="${BSC_EXTRA_FLAGS} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
pids=()
# host_rng.cpp: host-only C++ (NumPy's MT19937 shuffle restated, AVX2 paths behind a runtime check) — plain g++
if [ ! -f "$OBJ/host_rng.o" ] || [ "$HERE/host_rng.cpp" -nt "$OBJ/host_rng.o" ] || [ "$HERE/../../include/bscnav.h" -nt "$OBJ/host_rng.o" ]; then
  ( g++ -O3 -std=c++17 -fPIC -Wall -I"$HERE/../../include" -c "$HERE/host_rng.cpp" -o "$OBJ/host_rng.o" ) &
  pids+=($!)
fi
for f in prims radix ingest dense flush localize cluster frontier encoder_ops encoder_gemm capi; do
  src="$HERE/$f.hip"; obj="$OBJ/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/bsc_internal.h" -nt "$obj" ] || [ "$HERE/geometry_dev.h" -nt "$obj" ] || [ "$HERE/../../include/bscnav.h" -nt "$obj" ]; then
    ( hipcc $FLAGS -c "$src" -o "$obj" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/prims.o "$OBJ"/radix.o "$OBJ"/ingest.o "$OBJ"/dense.o "$OBJ"/flush.o "$OBJ"/localize.o "$OBJ"/cluster.o "$OBJ"/frontier.o "$OBJ"/encoder_ops.o "$OBJ"/encoder_gemm.o "$OBJ"/host_rng.o "$OBJ"/capi.o
echo "built $OUT"
