#!/bin/bash
# round-4 GPU call 5 (session re-entry): whole GPU suite, the bench line, per-kernel stats (with / without the text branch overlapped),
# Swin fwd+bwd alone with its kernel table, fused stage-0 launches one by one
O=$PWD/gpurun_out/r04; mkdir -p $O; R=$PWD
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1800 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest5.log 2>&1; echo "pytest rc=$?" >> $O/pytest5.log
timeout 900 python bench.py > $O/bench5.json 2> $O/bench5.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/profiso5 -o r --output-format csv -- python $R/bench.py --no-cpu-baseline --host-input-leg 0 --overlap-text 0 > $O/profiso5.log 2>&1
cd $R
PROF_ROWS=60 python tools/summarize_prof.py stats $O/profiso5 14 > $O/kernel_stats_no_overlap5.md
rm -rf $O/profiso5
timeout 300 python tools/probes/time_swin.py 640 > $O/swin_time5.txt 2>&1
bash tools/prof_swin.sh > /dev/null 2>&1; cp gpurun_out/swin_kernel_stats.md $O/swin_kernel_stats5.md
timeout 300 python tests/support_wblock_cases.py --speed > $O/wblock_speed5.txt 2>&1
tail -3 $O/pytest5.log; tail -1 $O/bench5.json | cut -c1-400; grep "fwd+bwd" $O/swin_time5.txt
