#!/bin/bash
# Builds the timing-experiment variant of the library (-DRTUF_ABLATE): the only build that contains the
# "skip this part of the kernel" branches.  Its images are WRONG by design; it is never the product library.
# The scripts/ablate_*.sh experiments select it with RTUF_LIB=realtime_urdf_filter_amd/lib/variants/librtuf_ablate.so.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/realtime_urdf_filter_amd/lib/variants"
"$root/realtime_urdf_filter_amd/csrc/build.sh" -DRTUF_ABLATE -o "$root/realtime_urdf_filter_amd/lib/variants/librtuf_ablate.so"
