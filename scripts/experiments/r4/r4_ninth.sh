#!/bin/bash
# round 4: set-up phase 2 on per-vertex cells (packed 16-bit) against the snapped-coordinate form (RTUF_CELLS=0 variant)
out=gpurun_out/r4_ninth; mkdir -p $out
timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $out/parity.txt 2>&1; tail -3 $out/parity.txt
V=realtime_urdf_filter_amd/lib/variants/librtuf_nocells.so
bash scripts/r4_ab.sh r4_ab_cells "" -- $V ""
bash scripts/r4_ab.sh r4_ab_cells_near "--near-arm --steps 40" -- $V ""
bash scripts/r4_ab.sh r4_ab_cells_c4 "--workload c4 --shard-of 8 --steps 50" -- $V ""
bash scripts/r4_ab.sh r4_ab_cells_c5 "--workload c5 --shard-of 8 --steps 30" -- $V ""
timeout 900 python scripts/fuzz_parity.py 6000 $((99 + RANDOM)) 2>&1 | tail -1; timeout 900 python scripts/fuzz_features.py 2000 $((98 + RANDOM)) 2>&1 | tail -1; FUZZ_BIG=1 timeout 600 python scripts/fuzz_parity.py 100 $((97 + RANDOM)) 2>&1 | tail -1
timeout 1200 python -m pytest tests -x -q -m gpu --deselect tests/test_parity_gpu.py 2>&1 | tail -2
