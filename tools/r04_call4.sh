#!/bin/bash
# round-4 GPU call 4: window-attention backward with own tiles from registers (own-first tile walk)
O=gpurun_out/r04; mkdir -p $O
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1500 python -m pytest tests/test_gpu_wblock.py tests/test_gpu_swin.py tests/test_gpu_ops.py tests/test_gpu_torch_ops.py -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest4.log 2>&1; echo "pytest rc=$?" >> $O/pytest4.log
B=$PWD/facialmmt_amd/libfmmt_hip_base.so
PROBE_LIB=$B timeout 300 python tools/probes/wattn_bench.py > $O/wattn4_base.txt 2>&1
timeout 300 python tools/probes/wattn_bench.py > $O/wattn4_new.txt 2>&1
PROBE_LIB=$B timeout 300 python tools/probes/time_swin.py 640 > $O/swin4_base.txt 2>&1
timeout 300 python tools/probes/time_swin.py 640 > $O/swin4_new.txt 2>&1
tail -3 $O/pytest4.log; grep "per step" $O/wattn4_*.txt; grep "fwd+bwd" $O/swin4_*.txt
