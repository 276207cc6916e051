#!/bin/bash
# Run on the GPU box (gpurun): regenerates everything under profiles/ for the current build in one go.
#   bash scripts/refresh_profiles.sh <tag>       e.g. r02  -> gpurun_out/profiles/<tag>_* (+ the three un-tagged json files
#   bench.py reads: pmc_counters.json, valu_peak.json; llvmpipe_baseline.json comes from the development container)
# rocprofv3 passes are separate runs: --kernel-trace --stats, then one --pmc pass per counter group (never together
# with other trace domains).
tag=${1:-r06}
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/profiles
rm -rf $out; mkdir -p $out
B="python $root/bench.py"
# The counters and the per-kernel trace are taken on the ONE-LANE launch shape (--lanes 1: every kernel alone on the GPU, the
# whole batch per launch): that is what bench.py's `roofline` describes (its one-lane leg); the default context overlaps the
# kernels of several launch groups, whose per-launch times describe no single kernel (traced too, as *_default_lanes).
PROF_ARGS="--steps 20 --min-seconds 0 --cpu-seconds 0 --check-frames 0 --lanes 1 --isolated-seconds 0 --host-copy-seconds 0"

# ---- 1. bench lines ---------------------------------------------------------------------------------------------
Q="--cpu-seconds 0 --host-copy-seconds 0 --min-seconds 3"       # the side lines: no CPU legs, no host-plane legs, 3 s timed
$B $Q > /dev/null 2> $out/bench.err      # first run on a fresh box is the slowest: warm-up
t0=$(date +%s.%N)
$B > $out/${tag}_bench.json 2>> $out/bench.err                   # THE line: defaults, exactly what the driver runs
echo "python bench.py (defaults, second run on this box): $(python -c "print(round($(date +%s.%N) - $t0, 1))") s wall" > $out/${tag}_bench_wall_time.txt
$B $Q --lanes 1 > $out/${tag}_bench_one_lane.json 2>> $out/bench.err
$B $Q --lanes 2 > $out/${tag}_bench_two_lanes.json 2>> $out/bench.err
$B $Q --launch-group 43 > $out/${tag}_bench_groups_of_43.json 2>> $out/bench.err
$B $Q --lanes 2 --launch-group 64 > $out/${tag}_bench_two_lanes_groups_of_64.json 2>> $out/bench.err
$B $Q --launch-group 22 > $out/${tag}_bench_groups_of_22.json 2>> $out/bench.err
$B $Q --two-kernel > $out/${tag}_bench_two_kernel.json 2>> $out/bench.err
$B $Q --two-kernel --streams 1024 --steps 30 > $out/${tag}_bench_two_kernel_1024.json 2>> $out/bench.err
$B $Q --streams 1024 --steps 30 > $out/${tag}_bench_1024.json 2>> $out/bench.err
$B $Q --u16 > $out/${tag}_bench_u16.json 2>> $out/bench.err
$B $Q --steps 500 --streams 1 > $out/${tag}_bench_batch1.json 2>> $out/bench.err
$B $Q --steps 500 --streams 1 --lanes 1 > $out/${tag}_bench_batch1_one_lane.json 2>> $out/bench.err
$B $Q --steps 500 --streams 1 --pipelines 3 > $out/${tag}_bench_batch1_pipelines3.json 2>> $out/bench.err
$B $Q --host-poses > $out/${tag}_bench_host_poses.json 2>> $out/bench.err
$B $Q --workload c4 --shard-of 8 --steps 50 > $out/${tag}_bench_c4_share.json 2>> $out/bench.err
$B $Q --workload c5 --shard-of 8 --steps 30 > $out/${tag}_bench_c5_share.json 2>> $out/bench.err
$B $Q --near-arm --steps 40 > $out/${tag}_bench_near_arm.json 2>> $out/bench.err      # every stream: forearm 0.1-0.35 m in front of the lens
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 $root/bench.py --gpus 1 $Q 2>> $out/bench.err | grep '^{' | tail -1 > $out/${tag}_bench_rccl_world1.json      # (RCCL prints its banner on stdout)
python $root/scripts/clip_stress.py > $out/${tag}_clip_stress.txt 2>> $out/bench.err
bash $root/scripts/overdraw.sh > $out/overdraw.json 2>> $out/bench.err
python $root/scripts/host_planes_rate.py > $out/${tag}_host_planes.json 2>> $out/bench.err
python $root/scripts/clip_stress.py > /dev/null 2>&1

# ---- 2. kernel traces -------------------------------------------------------------------------------------------
cd /tmp
trace() {   # name, bench args: the bench's own default step counts (only the CPU legs and the second context are left out)
  rm -rf /tmp/rp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o t -- $B --cpu-seconds 0 --check-frames 0 --host-copy-seconds 0 --isolated-seconds 0 --min-seconds 1 $2 > /dev/null 2>&1
  cp $(find /tmp/rp -name '*kernel_stats.csv' | head -1) $out/${tag}_kernel_stats$1.csv
}
trace "" "--lanes 1"                       # the roofline's launch shape: one lane, 256 streams per launch
trace "_default_lanes" ""                  # the headline context: three lanes, 86 streams per launch, kernels of the three groups overlap
trace "_two_kernel" "--two-kernel --lanes 1"
trace "_two_kernel_1024" "--two-kernel --streams 1024 --steps 30 --lanes 1"
trace "_c4_share" "--workload c4 --shard-of 8 --steps 50 --lanes 1"
trace "_near_arm" "--near-arm --steps 40 --lanes 1"

# ---- 3. counters (one pass per group) ---------------------------------------------------------------------------
pmc() {     # output file, bench args, counters...
  local f=$1 args=$2; shift 2
  rm -rf /tmp/rp; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/rp -o t -- $B $PROF_ARGS $args > /dev/null 2>&1
  { echo "## $*"; python $root/scripts/pmc_summary.py $(find /tmp/rp -name '*counter_collection.csv' | head -1); } >> $f
}
hdr="# rocprofv3 --kernel-trace --pmc <group> (one pass per '##' group), command: python bench.py $PROF_ARGS"
for mode in "" "--two-kernel" "--two-kernel --streams 1024" "--workload c4 --shard-of 8" "--near-arm"; do
  suffix=$(echo "$mode" | sed 's/--//g; s/ /_/g; s/-/_/g'); suffix=${suffix:+_$suffix}
  f=$out/${tag}_pmc${suffix}.txt
  echo "$hdr $mode   (values per launch, averaged over the launches of the run; FETCH_SIZE / WRITE_SIZE in KiB)" > $f
  pmc $f "$mode" FETCH_SIZE
  pmc $f "$mode" WRITE_SIZE
  if [ "$mode" = "--workload c4 --shard-of 8" ] || [ "$mode" = "--near-arm" ]; then
    pmc $f "$mode" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
    pmc $f "$mode" SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
  fi
  if [ -z "$mode" ]; then
    pmc $f "$mode" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
    pmc $f "$mode" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES
    pmc $f "$mode" SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM
    pmc $f "$mode" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY
    pmc $f "$mode" SQ_THREAD_CYCLES_VALU
  fi
done

# ---- 4. VALU issue micro-benchmark + the counters' calibration on it ----------------------------------------------
$root/scripts/bin/valu_peak 8 > $out/valu_peak_raw.json 2> $out/valu_peak.err
rm -rf /tmp/rp; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/rp -o t -- $root/scripts/bin/valu_peak 8 > /dev/null 2>&1
python - $(find /tmp/rp -name '*counter_collection.csv' | head -1) > $out/${tag}_pmc_valu_peak.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -- scripts/bin/valu_peak 8   (last launch of each kernel)")
for k, v in agg.items():
    if not k.startswith("k_"):
        continue
    i, a = v["SQ_INSTS_VALU"][-1], v["SQ_ACTIVE_INST_VALU"][-1]
    print("%-34s SQ_INSTS_VALU %.4g  SQ_ACTIVE_INST_VALU %.4g  ratio %.3f" % (k.split("(")[0], i, a, a / i if i else 0))
PY
cd $root
# the compare threshold's division core against the IEEE division, every float z the kernels can hand it (scripts/fdiv_check.hip)
$root/scripts/bin/fdiv_check 24 > $out/${tag}_fdiv_check.txt 2>&1
bash scripts/lane_util.sh > $out/${tag}_lanes.json 2> $out/lanes.err
python scripts/pmc_to_json.py $out $tag > $out/pmc_to_json.log 2>&1
python scripts/valu_mix.py $out/valu_peak.json > $out/valu_mix.log 2>&1 || true
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5) > $out/${tag}_gpu_tests.txt
{ python scripts/fuzz_parity.py 6000 $((20260928 + RANDOM)) 2>&1 | tail -2; python scripts/fuzz_features.py 3000 $((20270000 + RANDOM)) 2>&1 | tail -2; FUZZ_BIG=1 python scripts/fuzz_parity.py 150 $((333 + RANDOM)) 2>&1 | tail -2; } > $out/${tag}_fuzz.txt
ls -la $out; cat $out/pmc_to_json.log; tail -c 400 $out/bench.err
for f in $out/${tag}_bench*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-44s %9.0f frames/s  %.4f ms/step  %s %.1f us frac %.3f  parity %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"] * 1e3, r["frac"], d["parity"]["mismatching_values"]))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
