#!/bin/bash
# round-4 GPU call 3: Mlp generic-kernel parity tests; base-2 backward softmax; GELU' through the LDS table on the persistent kernel (gbw build)
O=gpurun_out/r04; mkdir -p $O
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1500 python -m pytest tests/test_gpu_wblock.py tests/test_gpu_swin.py tests/test_gpu_ops.py -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest3.log 2>&1; echo "pytest rc=$?" >> $O/pytest3.log
G=$PWD/facialmmt_amd/libfmmt_hip_gbw.so
timeout 300 python tools/probes/wattn_bench.py > $O/wattn3_new.txt 2>&1
for v in new gbw; do
  L=""; [ $v = gbw ] && L=$G
  PROBE_LIB=$L timeout 300 python tools/probes/nt_probe.py > $O/nt3_$v.txt 2>&1
  PROBE_LIB=$L timeout 300 python tools/probes/time_swin.py 640 > $O/swin3_$v.txt 2>&1
done
PROBE_LIB=$G timeout 600 python -m pytest tests/test_gpu_swin.py -m gpu -q -k "full_size or bf16_gradients" -p no:cacheprovider > $O/pytest3_gbw.log 2>&1
tail -3 $O/pytest3.log
