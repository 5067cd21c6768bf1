#!/bin/bash
# round-4 GPU call 12: whole steps, base build against the working tree's (polynomial GELU 8 / 10 terms, pipelined fused Mlp forward), alternating
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for v in base "" base ""; do
  L=""; [ -n "$v" ] && L=$PWD/facialmmt_amd/libfmmt_hip_$v.so
  PROBE_LIB=$L timeout 600 python tools/probes/bench_with_lib.py --no-cpu-baseline --other-configs 0 --host-input-leg 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib=${v:-new}', d['ms_per_step'], d['value'])"
done
for v in base ""; do
  L=""; [ -n "$v" ] && L=$PWD/facialmmt_amd/libfmmt_hip_$v.so
  PROBE_LIB=$L timeout 200 python tools/probes/mlp_fwd_only.py 2>&1 | grep -v amdgpu.ids
  PROBE_LIB=$L timeout 300 python tools/probes/time_swin.py 640 2>&1 | grep "fwd"
done
