set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/trace3; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-repeats --no-side-kernels > $O/bench.json 2> $O/err.txt
f=$(find $O/t -name "*kernel_trace.csv" | head -1); cp $f $O/kernel_trace.csv; wc -l $O/kernel_trace.csv
