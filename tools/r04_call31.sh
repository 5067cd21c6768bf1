#!/bin/bash
# round-4 GPU call 31: gradient hand-over + clip norm in one pass (fmmt_grad_handover) -- parity tests, then whole-step A/B (FUSED_HANDOVER patched off)
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
O=$PWD/gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_ddp.py tests/test_gpu_cabi.py -m gpu -q --maxfail=3 -p no:cacheprovider > $O/pytest31.log 2>&1; tail -3 $O/pytest31.log; grep -n "Error" $O/pytest31.log | head -5
for v in 0 1 0 1 0 1; do
  timeout 600 python - --no-cpu-baseline --other-configs 0 --host-input-leg 0 --steps 10 <<PY 2>$O/bench31.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused_handover=$v', d['ms_per_step'], d['value'])" || tail -5 $O/bench31.err
import sys, runpy
from facialmmt_amd import train_step
train_step.FUSED_HANDOVER = bool($v)
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
PY
done
