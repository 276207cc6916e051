#!/bin/bash
# Compiles the C++ facade example against librtuf.so (checks that the header-only facade builds).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
mkdir -p "$root/examples/bin"
exec g++ -std=c++17 -O2 -Wall -Wextra -Wno-reorder -I"$root/include" "$root/examples/example_filter.cpp" \
  -L"$here/../lib" -lrtuf -Wl,-rpath,'$ORIGIN/../../realtime_urdf_filter_amd/lib' -o "$root/examples/bin/example_filter"
