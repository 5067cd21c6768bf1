#!/bin/bash
# round-4 GPU call 2: parity tests of the generic (fp32 / bf16) instantiations + start-stagger A/B of the persistent NT kernel
O=gpurun_out/r04; mkdir -p $O
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1200 python -m pytest tests/test_gpu_wblock.py tests/test_gpu_swin.py tests/test_gpu_ops.py tests/test_gpu_torch_ops.py -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest2.log 2>&1; echo "pytest rc=$?" >> $O/pytest2.log
for v in new stag1 stag2 stag3; do
  L=""; [ $v != new ] && L=$PWD/facialmmt_amd/libfmmt_hip_$v.so
  PROBE_LIB=$L timeout 300 python tools/probes/nt_probe.py > $O/nt2_$v.txt 2>&1
done
for v in new stag1 stag2 stag3; do
  L=""; [ $v != new ] && L=$PWD/facialmmt_amd/libfmmt_hip_$v.so
  PROBE_LIB=$L timeout 300 python tools/probes/time_swin.py 640 > $O/swin2_$v.txt 2>&1
done
tail -3 $O/pytest2.log
