cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in 0 1 2 4 3 5; do
export ALIGNER_SA_EXP=$e
O=gpurun_out/exp_sa$e; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels > $O/b.json 2> $O/err
echo "EXP $e: $(grep softattn_rt $(find $O/st -name '*kernel_stats.csv' | head -1) | cut -d, -f2-7)"; rm -rf $O/st
done
