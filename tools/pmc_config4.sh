cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc4_$c
  timeout 500 rocprofv3 --kernel-trace --pmc $c -d $O/pmc4_$c -o r --output-format csv -- python $R/bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --graphs 0 --other-configs 0 --host-input-leg 0 > $O/pmc4_$c.log 2>&1
  cd $R; PMC_ROWS=25 python tools/summarize_prof.py pmc $O/pmc4_$c > $O/pmc4_$c.md; cd /tmp
  rm -rf $O/pmc4_$c
done
grep -h "linear_nt_kernelIDF16bLi128ELi128ELi64ELi2ELb1" $O/pmc4_FETCH_SIZE.md $O/pmc4_WRITE_SIZE.md
