#!/bin/bash
# round-4 GPU call 14: fused Mlp forward keeps the raw x fragments as the epilogue's residual (LN, C = 96); phase map of a replayed step
O=$PWD/gpurun_out/r04; mkdir -p $O; R=$PWD
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_wblock.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "mlp or Mlp" > $O/pytest14.log 2>&1; echo "pytest rc=$?" >> $O/pytest14.log
tail -3 $O/pytest14.log
for v in base ""; do
  L=""; [ -n "$v" ] && L=$PWD/facialmmt_amd/libfmmt_hip_$v.so
  PROBE_LIB=$L timeout 200 python tools/probes/mlp_fwd_only.py 2>&1 | grep "mlp_ln"
  PROBE_LIB=$L timeout 300 python tools/probes/time_swin.py 640 2>&1 | grep "fwd"
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/proftl -o r --output-format csv -- python $R/bench.py --no-cpu-baseline --host-input-leg 0 --other-configs 0 > $O/proftl.log 2>&1
cd $R
python tools/timeline.py $O/proftl > $O/timeline14.txt 2>&1
rm -rf $O/proftl
grep -A80 "phase map" $O/timeline14.txt | head -90
