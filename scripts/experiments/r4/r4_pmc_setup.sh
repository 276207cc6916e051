#!/bin/bash
# where the set-up kernel's instructions are: SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS under the ablation flags (RTUF_ABLATE build), one lane
export TMPDIR=/tmp; root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r4_pmc_setup; mkdir -p $out
cd /tmp
for f in 0 0x10000 0x20000 0x40000 0x60000; do
  args="--steps 12 --min-seconds 0 --cpu-seconds 0 --check-frames 0 --debug-flags $f --lanes 1 --isolated-seconds 0 --host-copy-seconds 0"
  rm -rf /tmp/rp; (cd $root && RTUF_LIB=$root/realtime_urdf_filter_amd/lib/variants/librtuf_ablate.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d /tmp/rp -o t -- python $root/bench.py $args > /dev/null 2>&1)
  v=$(python $root/scripts/pmc_summary.py $(find /tmp/rp -name '*counter_collection.csv' | head -1) | grep -A4 "setup_kernel<false>" | tr '\n' ' ')
  echo "flags=$f $v" | tee -a $out/summary.txt
done
