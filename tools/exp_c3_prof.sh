# usage (GPU box): bash tools/exp_c3_prof.sh -- rocprofv3 kernel stats of the configs[2] step (development loop of the encoder kernels)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/exp_c3; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --config c3 --no-cpu-baseline > $O/b.json 2> $O/err
grep "conv_\|softattn\|maxpath" $(find $O/st -name "*kernel_stats.csv" | head -1) | cut -c1-200 > gpurun_out/exp_c3.txt; rm -rf $O/st
