for a in "--width 1280 --height 720 --streams 64" "--streams 1024" "--triangles 20000" "--streams 64 --triangles 1000000" "--no-mask" "--streams 256 --pipelines 2 --u16"; do
 echo -n "$a : "; python bench.py --steps ${STEPS:-60} --warmup 3 --cpu-seconds 0 --check-frames 3 $a 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['rasteriser']['regrowths'], d['rasteriser']['max_bin_fill'], d['rasteriser']['bin_capacity'], d['parity'])"; done
