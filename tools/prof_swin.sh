cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
rm -rf $O/prof_swin
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_swin -o r --output-format csv -- python $R/tools/probes/time_swin.py 640 > $O/prof_swin.log 2>&1
cd $R
PROF_ROWS=70 python tools/summarize_prof.py stats $O/prof_swin 5 > $O/swin_kernel_stats.md
rm -rf $O/prof_swin
head -80 $O/swin_kernel_stats.md
