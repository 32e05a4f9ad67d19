export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ovtrace; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $R/bench.py --mode minibatch --capture --steps 12 --warmup 4 --scale 0.25 > $OUT/stdout.log 2>&1)
python $R/scripts/overlap_trace.py $(find $OUT -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/overlap_trace.txt 2>&1
head -80 $R/gpurun_out/overlap_trace.txt
find $OUT -name "*.csv" -size +4M -delete
