# usage (GPU box): bash tools/profile_c3_c5.sh OUTDIR -- bench lines + rocprofv3 kernel stats of BASELINE configs[2] and configs[4]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/c3c5}; mkdir -p $O
for c in c3 c5; do
  python3 bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err
  tail -1 $O/bench_$c.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -- python3 bench.py --config $c --no-cpu-baseline > /dev/null 2> $O/prof_$c.err
  f=$(find $O/stats_$c -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$c.csv
  head -12 $O/kernel_stats_$c.csv | cut -c1-200
done
