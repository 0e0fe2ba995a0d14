cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/train_trace4 -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py --mc 320 --side 64 --n 16 --iters 2 > $GRAFT_REPO_ROOT/gpurun_out/train_trace4.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/train_trace4 -name "*kernel_stats.csv" -exec cp {} gpurun_out/train_kernel_stats4.csv \;
find gpurun_out/train_trace4 -name "*kernel_trace.csv" -delete
head -40 gpurun_out/train_kernel_stats4.csv | cut -c1-150
