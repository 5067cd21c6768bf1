#!/bin/bash
# round-4 GPU call 33 (final binary and defaults of the session: sequential fusion halves, vectorised cast_batch, fused gradient hand-over): the whole GPU suite on the working tree, the issue-rate probe, then the round's profile pass (tools/profile_round.sh r04)
O=$PWD/gpurun_out/r04; mkdir -p $O
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1800 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest33.log 2>&1; echo "pytest rc=$?" >> $O/pytest33.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/probes/valu_rate.hip > /dev/null 2>&1 && /tmp/valu_rate > $O/issue_rates.txt 2>&1
bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1
tail -3 $O/pytest33.log; cut -c1-300 gpurun_out/summ_r04/bench.json
