#!/bin/bash
# round-4 GPU call 8: fused Mlp forward (C = 96) as a software pipeline (reads a phase ahead, MFMAs interleaved with the epilogue's VALU)
O=$PWD/gpurun_out/r04; mkdir -p $O
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for v in base ""; do
  L=""; [ -n "$v" ] && L=$PWD/facialmmt_amd/libfmmt_hip_$v.so
  PROBE_LIB=$L timeout 200 python tools/probes/mlp_fwd_only.py 2>&1 | grep -v amdgpu.ids
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_wblock.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "mlp or Mlp" > $O/pytest8.log 2>&1; echo "pytest rc=$?" >> $O/pytest8.log
tail -5 $O/pytest8.log
