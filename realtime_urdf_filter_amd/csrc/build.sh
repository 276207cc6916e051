#!/bin/bash
# Builds the product library (HIP kernels + C ABI) for gfx950, in-tree.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
mkdir -p "$here/../lib"
exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
  -I"$root/include" -I"$here" \
  -Wno-unused-value -Wno-unused-result "$here/rtuf_kernels.hip" "$here/rtuf_api.cpp" \
  -o "$here/../lib/librtuf.so" "$@"
