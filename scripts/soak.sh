one() { python bench.py --cpu-seconds 0 --host-copy-seconds 0 --isolated-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'timed_steps', d['timed_steps'], 'ms/step', round(d['ms_per_step'],4), 'mism', d['parity']['mismatching_values'], 'frames checked', d['parity'].get('frames_checked'), 'regrowths', d['rasteriser']['regrowths'])"; }
echo -n "c3 120 s: "; one --min-seconds 120
echo -n "c3 host poses, 60 s: "; one --min-seconds 60 --host-poses
echo -n "arm in front of the lens, 60 s: "; one --min-seconds 60 --near-arm --steps 40
echo -n "c5 share 60 s: "; one --min-seconds 60 --workload c5 --shard-of 8 --steps 30
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
