#!/bin/bash
# round-4 GPU call 17: fmmt_linear_ln_bwd (d(LN out) GEMM + LayerNorm' + residual gradient in one launch) -- tests, same-call A/B
O=$PWD/gpurun_out/r04; mkdir -p $O
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1200 python -m pytest tests/test_gpu_wblock.py tests/test_gpu_swin.py -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest17.log 2>&1; echo "pytest rc=$?" >> $O/pytest17.log
tail -4 $O/pytest17.log
timeout 300 python tests/support_wblock_cases.py --speed 2>&1 | grep stage0
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu
import torch, time
from facialmmt_amd import ops
import tests.support_wblock_cases as S
PY
for v in 0 1 0 1; do
  timeout 300 python - <<PY 2>&1 | grep "fwd+bwd"
import sys, runpy
from facialmmt_amd import ops
ops._WBLOCK_LNBWD = bool($v)
sys.argv = ["time_swin.py", "640"]
print("lnbwd=$v", end=" ")
runpy.run_path("tools/probes/time_swin.py", run_name="__main__")
PY
done
