# instruction / wave counters of the big kernels for several library variants (one rocprofv3 --pmc pass per group)
for v in "$@"; do
  if [ "$v" = default ]; then unset RTUF_LIB; else export RTUF_LIB=$PWD/realtime_urdf_filter_amd/lib/variants/librtuf_$v.so; fi
  echo "=== $v"
  bash scripts/pmc_kernels.sh SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD 2>&1 | grep -A5 "setup_kernel<false>\|tile_kernel<false, false>" | grep -v "^--"
  bash scripts/pmc_kernels.sh SQ_WAVES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS 2>&1 | grep -A5 "setup_kernel<false>\|tile_kernel<false, false>" | grep -v "^--"
done
