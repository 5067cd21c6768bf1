#!/bin/bash
# Collect the evidence committed under profiles/ (run on the GPU box through gpurun):
#   tools/profile_round.sh TAG        -> gpurun_out/{bench_TAG.json, prof_TAG/, pmc_*_TAG/}
# PMC passes run WITHOUT HIP graphs (counter collection under graph replay crashes rocprofv3 on this image) and
# each under its own timeout.
TAG=${1:-x}
R=/root/repo
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o r --output-format csv -- python $R/bench.py --no-cpu-baseline > $O/prof_$TAG.log 2>&1
# the same without the second-stream overlap (8 steps): per-kernel durations undisturbed by concurrent text-encoder work
timeout 600 rocprofv3 --kernel-trace --stats -d $O/profiso_$TAG -o r --output-format csv -- python $R/bench.py --no-cpu-baseline --overlap-text 0 > $O/profiso_$TAG.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${c}_$TAG -o r --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --graphs 0 > $O/pmc_${c}_$TAG.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_MFMA_$TAG -o r --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --graphs 0 > $O/pmc_MFMA_$TAG.log 2>&1
# summarise on the box and drop the raw traces (gpurun copies back at most 64 MiB)
S=$O/summ_$TAG
mkdir -p $S
cp $O/bench_$TAG.json $S/bench.json
python $R/tools/summarize_prof.py stats $O/prof_$TAG 11 > $S/kernel_stats.md     # 2 warm-up + 6 timed + 3 post-region steps without overlap
python $R/tools/summarize_prof.py stats $O/profiso_$TAG 8 > $S/kernel_stats_no_overlap.md
python $R/tools/summarize_prof.py pmc $O/pmc_FETCH_SIZE_$TAG > $S/pmc_fetch.md
python $R/tools/summarize_prof.py pmc $O/pmc_WRITE_SIZE_$TAG > $S/pmc_write.md
python $R/tools/summarize_prof.py pmc $O/pmc_MFMA_$TAG > $S/pmc_mfma.md
rm -rf $O/prof_$TAG $O/profiso_$TAG $O/pmc_FETCH_SIZE_$TAG $O/pmc_WRITE_SIZE_$TAG $O/pmc_MFMA_$TAG
tail -1 $O/bench_$TAG.json | cut -c1-200
