#!/bin/bash
# Counter passes over ONE command (run on the GPU box through gpurun):  tools/pmc_kernel.sh TAG -- <command ...>
# Writes gpurun_out/pmc_TAG_{sq1,sq2,mem}.md (per-kernel sums / per-dispatch averages) and drops the raw traces.
TAG=$1; shift; shift
R=/root/repo; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmcraw_${TAG}_$name -o r --output-format csv -- "${CMD[@]}" > $O/pmc_${TAG}_$name.log 2>&1
  python $R/tools/summarize_prof.py pmc $O/pmcraw_${TAG}_$name > $O/pmc_${TAG}_$name.md 2>&1
  rm -rf $O/pmcraw_${TAG}_$name
}
CMD=("$@")
cd $R
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
run mem1 FETCH_SIZE GRBM_GUI_ACTIVE
run mem2 WRITE_SIZE
