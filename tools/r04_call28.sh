#!/bin/bash
# round-4 GPU call 28: fmmt_cast_batch with 16-byte loads / 8-byte stores on whole aligned fp32 tiles -- bit-exactness test + whole-step A/B
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 600 python -m pytest tests/test_gpu_glue.py -m gpu -q -p no:cacheprovider -k "cast_batch" 2>&1 | tail -2
for v in prev "" prev "" prev ""; do
  L=""; [ -n "$v" ] && L=$PWD/facialmmt_amd/libfmmt_hip_$v.so
  PROBE_LIB=$L timeout 600 python tools/probes/bench_with_lib.py --no-cpu-baseline --other-configs 0 --host-input-leg 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib=${v:-new}', d['ms_per_step'], d['value'])"
done
