#!/bin/bash
# Collect the evidence committed under profiles/ (run on the GPU box through gpurun):
#   tools/profile_round.sh TAG        -> gpurun_out/summ_TAG/{bench.json, bench_config3.json, bench_config4.json, kernel_stats*.md,
#                                        launch_census.md, pmc_*.md, timeline.txt, gemm_shapes.txt, nt_shapes.txt, tn_shapes.txt, few_shapes.txt}
# then, here: bash tools/copy_profiles.sh TAG   (-> profiles/TAG_*, profiles/traffic.json)
# PMC passes run WITHOUT HIP graphs (counter collection under graph replay crashes rocprofv3 on this image) and
# each under its own timeout.
TAG=${1:-x}
R=/root/repo
O=$R/gpurun_out
S=$O/summ_$TAG
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err
tail -1 $O/bench_$TAG.json > $S/bench.json
for c in 3 4; do
  timeout 600 python $R/bench.py --config $c --no-cpu-baseline 2> $O/bench_c${c}_$TAG.err | tail -1 > $S/bench_config$c.json
done
# per-kernel GPU time of the default command minus the CPU leg: 2 eager warm-up passes inside the graph capture + 2 warm-up
# + 6 timed + 1 idle-queue step (all graph replays) + 3 eager probe passes = 14 passes
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o r --output-format csv -- python $R/bench.py --no-cpu-baseline --host-input-leg 0 > $O/prof_$TAG.log 2>&1
# the same without the text branch forked onto its own stream: per-kernel durations undisturbed by concurrent text-encoder work
timeout 600 rocprofv3 --kernel-trace --stats -d $O/profiso_$TAG -o r --output-format csv -- python $R/bench.py --no-cpu-baseline --host-input-leg 0 --overlap-text 0 > $O/profiso_$TAG.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${c}_$TAG -o r --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --graphs 0 > $O/pmc_${c}_$TAG.log 2>&1
done
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_MFMA_$TAG -o r --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --graphs 0 > $O/pmc_MFMA_$TAG.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $O/pmc_LDS_$TAG -o r --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --graphs 0 > $O/pmc_LDS_$TAG.log 2>&1
# summarise on the box and drop the raw traces (gpurun copies back at most 64 MiB)
cd $R
PROF_ROWS=60 python tools/summarize_prof.py stats $O/prof_$TAG 14 > $S/kernel_stats.md
PROF_ROWS=60 python tools/summarize_prof.py stats $O/profiso_$TAG 14 > $S/kernel_stats_no_overlap.md
python tools/count_launches.py $O/prof_$TAG 14 50 > $S/launch_census.md
python tools/summarize_prof.py pmc $O/pmc_FETCH_SIZE_$TAG > $S/pmc_fetch.md
python tools/summarize_prof.py pmc $O/pmc_WRITE_SIZE_$TAG > $S/pmc_write.md
python tools/summarize_prof.py pmc $O/pmc_MFMA_$TAG > $S/pmc_mfma.md
python tools/summarize_prof.py pmc $O/pmc_LDS_$TAG > $S/pmc_lds.md
python tools/timeline.py $O/prof_$TAG > $S/timeline.txt 2>&1
rm -rf $O/prof_$TAG $O/profiso_$TAG $O/pmc_FETCH_SIZE_$TAG $O/pmc_WRITE_SIZE_$TAG $O/pmc_MFMA_$TAG $O/pmc_LDS_$TAG
GEMM_BENCH_VENDOR=1 timeout 600 python tools/probes/gemm_bench.py > $S/gemm_shapes.txt 2>&1
timeout 300 python tools/probes/nt_probe.py --vendor > $S/nt_shapes.txt 2>&1
timeout 300 python tools/probes/tn_probe.py > $S/tn_shapes.txt 2>&1
timeout 200 python tools/probes/few_probe.py > $S/few_shapes.txt 2>&1
# Swin forward + backward alone (640 frames) with its own per-kernel table, and the fused stage-0 launches one by one
timeout 300 python tools/probes/time_swin.py 640 > $S/swin_time.txt 2>&1
bash tools/prof_swin.sh > /dev/null 2>&1; cp $O/swin_kernel_stats.md $S/swin_kernel_stats.md
timeout 300 python tests/support_wblock_cases.py --speed > $S/wblock_speed.txt 2>&1
python tools/make_traffic_json.py $S $TAG > $O/traffic_$TAG.json 2>/dev/null
cut -c1-300 $S/bench.json
