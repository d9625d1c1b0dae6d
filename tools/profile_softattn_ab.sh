# usage (GPU box): bash tools/profile_softattn_ab.sh TAG -- rocprofv3 kernel stats of the one-batch-at-a-time step with the
# similarity kernel in its row-tile form and (ALIGNER_SOFTATTN_STRIPS=1) in its strip-per-wave form, same box
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r05}; O=gpurun_out/$T; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_new -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels > $O/bench_streams1.json 2> $O/stats_new.err
cp $(find $O/stats_new -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; rm -rf $O/stats_new
export ALIGNER_SOFTATTN_STRIPS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_old -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels > $O/bench_streams1_strips.json 2> $O/stats_old.err
cp $(find $O/stats_old -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_strips.csv; rm -rf $O/stats_old
head -5 $O/bench_kernel_stats.csv | cut -c1-200; head -5 $O/bench_kernel_stats_strips.csv | cut -c1-200
