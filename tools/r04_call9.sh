#!/bin/bash
# round-4 GPU call 9: the library built with -fno-slp-vectorize (no compiler-formed v_pk_* beside the MFMAs) against the default build
O=$PWD/gpurun_out/r04; mkdir -p $O
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for v in "" noslp; do
  L=""; [ -n "$v" ] && L=$PWD/facialmmt_amd/libfmmt_hip_$v.so
  echo "== build: ${v:-default}"
  PROBE_LIB=$L timeout 200 python tools/probes/mlp_fwd_only.py 2>&1 | grep -v amdgpu.ids
  PROBE_LIB=$L timeout 300 python tools/probes/time_swin.py 640 2>&1 | grep "fwd"
  PROBE_LIB=$L timeout 300 python tools/probes/nt_probe.py 2>&1 | grep -v amdgpu.ids | tail -16
done
