#!/bin/bash
# round-4 GPU call 6: GELU as a packed odd polynomial (no table, no transcendental) -- suite + same-call A/B against the round's baseline build
O=$PWD/gpurun_out/r04; mkdir -p $O
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
B=$PWD/facialmmt_amd/libfmmt_hip_base.so
PROBE_LIB=$B timeout 300 python tools/probes/mlp_bench.py > $O/mlp6_base.txt 2>&1
timeout 300 python tools/probes/mlp_bench.py > $O/mlp6_new.txt 2>&1
PROBE_LIB=$B timeout 300 python tools/probes/time_swin.py 640 > $O/swin6_base.txt 2>&1
timeout 300 python tools/probes/time_swin.py 640 > $O/swin6_new.txt 2>&1
timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_wblock.py tests/test_gpu_swin.py tests/test_gpu_torch_ops.py tests/test_gpu_cabi.py -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest6.log 2>&1; echo "pytest rc=$?" >> $O/pytest6.log
tail -3 $O/pytest6.log; cat $O/mlp6_base.txt $O/mlp6_new.txt | grep -v amdgpu.ids; grep "fwd" $O/swin6_*.txt
