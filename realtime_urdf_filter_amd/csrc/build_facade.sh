#!/bin/bash
# Compiles the C++ examples against librtuf.so: the header-only facade (example_filter) and the multi-device host over
# RCCL (multi_gpu_filter: needs the HIP + RCCL headers, so hipcc).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
mkdir -p "$root/examples/bin"
g++ -std=c++17 -O2 -Wall -Wextra -I"$root/include" "$root/examples/example_filter.cpp" \
  -L"$here/../lib" -lrtuf -Wl,-rpath,'$ORIGIN/../../realtime_urdf_filter_amd/lib' -o "$root/examples/bin/example_filter"
hipcc -std=c++17 -O2 -Wall -Wno-unused-result -I"$root/include" "$root/examples/multi_gpu_filter.cpp" \
  -L"$here/../lib" -lrtuf -lrccl -Wl,-rpath,'$ORIGIN/../../realtime_urdf_filter_amd/lib' -o "$root/examples/bin/multi_gpu_filter"
# the ROS adapter's sources against tests/ros_mock (a mock of the ROS 1 types they touch, NOT ROS): compile check of node and
# nodelet, and the harness that drives the camera callback (tests/test_ros_adapter.py)
mock="$root/tests/ros_mock"
for f in rtuf_node rtuf_nodelet; do
  g++ -std=c++17 -fsyntax-only -Wall -Wextra -I"$mock/include" -I"$root/ros/include" -I"$root/include" "$root/ros/src/$f.cpp"
done
g++ -std=c++17 -O2 -Wall -Wextra -I"$mock/include" -I"$root/ros/include" -I"$root/include" "$mock/ros_adapter_harness.cpp" \
  -L"$here/../lib" -lrtuf -Wl,-rpath,'$ORIGIN/../../realtime_urdf_filter_amd/lib' -o "$root/examples/bin/ros_adapter_harness"
