#!/bin/bash
# Run on the GPU box (gpurun): regenerates everything under profiles/ for the current build.
#   bash scripts/refresh_profiles.sh <tag>       e.g. r01  -> gpurun_out/profiles/<tag>_*
# rocprofv3 passes are separate: --kernel-trace --stats, then --pmc FETCH_SIZE, then --pmc WRITE_SIZE.
tag=${1:-r01}
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/profiles
rm -rf $out; mkdir -p $out
B="python $root/bench.py"
$B --cpu-seconds 20 > $out/${tag}_bench.json 2> $out/bench.err
$B --cpu-seconds 0 --two-kernel > $out/${tag}_bench_two_kernel.json 2>> $out/bench.err
$B --cpu-seconds 0 --u16 > $out/${tag}_bench_u16.json 2>> $out/bench.err
$B --steps 200 --cpu-seconds 0 --streams 1 > $out/${tag}_bench_batch1.json 2>> $out/bench.err
$B --cpu-seconds 0 --host-poses > $out/${tag}_bench_host_poses.json 2>> $out/bench.err
$B --cpu-seconds 0 --pipelines 2 > $out/${tag}_bench_pipelines2.json 2>> $out/bench.err
$B --cpu-seconds 0 --pipelines 3 > $out/${tag}_bench_pipelines3.json 2>> $out/bench.err
cd /tmp
for mode in "" "--two-kernel"; do
  suffix=${mode:+_two_kernel}
  rm -rf /tmp/rp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o t -- $B --steps 20 --cpu-seconds 0 --check-frames 0 $mode > /dev/null 2>&1
  cp $(find /tmp/rp -name '*kernel_stats.csv' | head -1) $out/${tag}_kernel_stats${suffix}.csv
done
{
  echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), command: python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --check-frames 0"
  echo "# values are KiB per launch, averaged over the launches of the run (MI355X, gfx950, ROCm 7.2)"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/rp; rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/rp -o t -- $B --steps 20 --cpu-seconds 0 --check-frames 0 > /dev/null 2>&1
    echo "## $ctr"
    python $root/scripts/pmc_summary.py $(find /tmp/rp -name '*counter_collection.csv' | head -1)
  done
} > $out/${tag}_pmc_hbm_traffic.txt
cd $root
python - "$out/${tag}_pmc_hbm_traffic.txt" "$out/hbm_traffic.json" "$tag" <<'PY'
import json, re, sys
txt = open(sys.argv[1]).read()
def grab(section):
    part = txt.split("## " + section)[1]
    m = re.search(r"tile_kernel<false, false>[^\n]*\n\s+%s\s+avg ([0-9.e+]+)" % section, part)
    return float(m.group(1))
f, w = grab("FETCH_SIZE"), grab("WRITE_SIZE")
json.dump({"kernel": "tile_kernel<fused>", "mode": "fused", "streams": 256, "width": 640, "height": 480, "triangles": 250388,
           "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
           "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): fetch bytes = 2 x FETCH_SIZE x 1024; write bytes = WRITE_SIZE x 1024",
           "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024),
           "source": "profiles/%s_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, python bench.py --steps 20 --warmup 3)" % sys.argv[3]},
          open(sys.argv[2], "w"), indent=1)
PY
bash scripts/pmc_kernels.sh SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES > $out/pmc_sq.body 2>&1
{ echo "# rocprofv3 --kernel-trace --pmc <4 SQ counters per pass>, command: python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --check-frames 0 (scripts/pmc_kernels.sh)"
  echo "# values are per launch, averaged over the launches of the run (MI355X, gfx950, ROCm 7.2); SQ_INSTS_* count wave64 instructions"
  cat $out/pmc_sq.body; } > $out/${tag}_pmc_sq.txt; rm -f $out/pmc_sq.body
python - "$out/${tag}_pmc_sq.txt" "$out/valu_counts.json" "$tag" <<'PY'
import json, re, sys
txt = open(sys.argv[1]).read()
def grab(kernel):
    m = re.search(re.escape(kernel) + r"[^\n]*\n(?:\s+SQ_\w+\s+avg [0-9.e+]+[^\n]*\n)*?\s+SQ_INSTS_VALU\s+avg ([0-9.e+]+)", txt)
    return float(m.group(1)) if m else None
json.dump({"streams": 256, "width": 640, "height": 480, "triangles": 250388,
           "wave64_valu_instructions_per_launch": {"tile_kernel<fused>": grab("tile_kernel<false, false>"), "setup_kernel": grab("setup_kernel<false>")},
           "peak_G_per_s": 614.4, "peak_note": "256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction",
           "source": "profiles/%s_pmc_sq.txt (rocprofv3 --pmc SQ_INSTS_VALU ..., python bench.py --steps 10 --warmup 2)" % sys.argv[3]},
          open(sys.argv[2], "w"), indent=1)
PY
ls -la $out; tail -c 600 $out/${tag}_bench.json; cat $out/${tag}_kernel_stats.csv | head -12; cat $out/hbm_traffic.json
