#!/bin/bash
# frames/s, per-kernel times and memory against the bins' direct capacity (one box)
out=gpurun_out/${1:-r4_capsweep}; mkdir -p $out
Q="--cpu-seconds 0 --host-copy-seconds 0 --min-seconds 2 --isolated-seconds 1.5 --check-frames 8"
run() { # lib, args
  RTUF_LIB=$1 timeout 600 python bench.py $Q $2 > $out/b.json 2> $out/b.err
  python - $out/b.json "$(basename ${1:-product} .so) $2" <<'PY' | tee -a $out/summary.txt
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; ks={e["kernel"]:e for e in [r]+r["all_kernels"]}; c=d["config"]
    print("%-46s" % sys.argv[2], round(d["value"]), "frames/s parity", d["parity"]["mismatching_values"], "GB %.2f" % (d.get("device_memory_bytes", c.get("device_memory_bytes", 0))/1e9),
          {k.split("_")[0]:(round(v["avg_launch_ms"]*1e3,1), round((v.get("in_headline_run") or {}).get("avg_launch_ms",0)*1e3,1)) for k,v in ks.items()}, "one-lane", round((r.get("one_lane_leg") or {}).get("frames_per_s",0)))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
shift
for rep in 1 2; do
  run realtime_urdf_filter_amd/lib/variants/librtuf_prev.so ""
  for a in "$@"; do run "" "$a"; done
done
