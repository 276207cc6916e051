#!/bin/bash
# tile / set-up kernel time with parts of the work skipped (RTUF_ABLATE build: images are wrong), one lane, C3 and near-arm
export RTUF_LIB=realtime_urdf_filter_amd/lib/variants/librtuf_ablate.so
out=gpurun_out/r4_ablate; mkdir -p $out
for wl in "" "--near-arm"; do
for f in 0 0x100 0xc00 0x800 0x400 0x200 0x1000 0x2000 0x3000 0x10000 0x20000 0x40000 0x60000; do
 echo -n "[$wl] flags=$f " | tee -a $out/summary.txt
 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --check-frames 0 --lanes 1 --isolated-seconds 0 --host-copy-seconds 0 --min-seconds 1 --debug-flags $f $wl 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; ks={e['kernel'].split('_')[0]:round(e['avg_launch_ms']*1e3,1) for e in [r]+r['all_kernels']}; print(round(d['value']), ks)" | tee -a $out/summary.txt
done; done
