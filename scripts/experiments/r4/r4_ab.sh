#!/bin/bash
# A/B of library variants on one box: usage scripts/r4_ab.sh <outdir> <bench args> -- lib1 lib2 ...   ("" = the product library)
out=gpurun_out/$1; mkdir -p $out; shift
args="$1"; shift; shift
for rep in 1 2; do
for lib in "$@"; do
  tag=$(basename ${lib:-product} .so)
  RTUF_LIB=$lib timeout 600 python bench.py --cpu-seconds 0 --host-copy-seconds 0 --min-seconds 2 --isolated-seconds 1.5 --check-frames 8 $args > $out/bench_${tag}_$rep.json 2> $out/bench_${tag}_$rep.err
  python - $out/bench_${tag}_$rep.json "$tag" <<'PY' | tee -a $out/summary.txt
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; ks={e["kernel"]:e for e in [r]+r["all_kernels"]}
    print("%-22s" % sys.argv[2], round(d["value"]), "frames/s", "parity", d["parity"]["mismatching_values"],
          {k.split("_")[0]:(round(v["avg_launch_ms"]*1e3,1), round((v.get("in_headline_run") or {}).get("avg_launch_ms",0)*1e3,1)) for k,v in ks.items()}, "one-lane", round((r.get("one_lane_leg") or {}).get("frames_per_s",0)))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
done; done
