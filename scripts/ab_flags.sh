for f in ${FLAGS:-0 0x8000 0 0x8000}; do echo -n "flags=$f "; python bench.py --steps 30 --warmup 3 --cpu-seconds 0 --check-frames 2 --debug-flags $f 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['kernel_ms_per_step']['ms_setup'], d['kernel_ms_per_step']['ms_raster'], d['parity'])"; done
