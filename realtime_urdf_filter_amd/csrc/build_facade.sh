#!/bin/bash
# Compiles the C++ examples against librtuf.so: the header-only facade (example_filter) and the multi-device host over
# RCCL (multi_gpu_filter: needs the HIP + RCCL headers, so hipcc).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
mkdir -p "$root/examples/bin"
g++ -std=c++17 -O2 -Wall -Wextra -Wno-reorder -I"$root/include" "$root/examples/example_filter.cpp" \
  -L"$here/../lib" -lrtuf -Wl,-rpath,'$ORIGIN/../../realtime_urdf_filter_amd/lib' -o "$root/examples/bin/example_filter"
hipcc -std=c++17 -O2 -Wall -Wno-unused-result -I"$root/include" "$root/examples/multi_gpu_filter.cpp" \
  -L"$here/../lib" -lrtuf -lrccl -Wl,-rpath,'$ORIGIN/../../realtime_urdf_filter_amd/lib' -o "$root/examples/bin/multi_gpu_filter"
