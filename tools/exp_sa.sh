# usage (GPU box): bash tools/exp_sa.sh -- stamps + rocprofv3 kernel stats of the one-batch-at-a-time step (development loop of the similarity kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 python3 tools/sa_rt_stamps.py > gpurun_out/sa_rt_stamps.txt 2>&1
O=gpurun_out/exp_sa; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --steps 3000 --warmup 300 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels > $O/b.json 2> $O/err
head -4 $(find $O/st -name "*kernel_stats.csv" | head -1) | cut -c1-150 > gpurun_out/exp_sa.txt; rm -rf $O/st
