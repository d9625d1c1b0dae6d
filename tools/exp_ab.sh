# usage (GPU box): bash tools/exp_ab.sh -- the similarity kernel of two builds of the library on the SAME box, alternating
# (aligner_amd/lib/libaligner_amd.so against aligner_amd/lib/prev.so; boxes differ by +-3 %: never compare across calls)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in ${ROUNDS:-1 2}; do for L in prev.so libaligner_amd.so; do
export ALIGNER_AMD_LIB=$GRAFT_REPO_ROOT/aligner_amd/lib/$L
O=gpurun_out/ab_$L; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --steps 3000 --warmup 300 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels > $O/b.json 2> $O/err
echo "$L: $(grep 'softattn' $(find $O/st -name '*kernel_stats.csv' | head -1) | head -1 | sed 's/.*SoftAttnParams)",//' | cut -d, -f1-6)"; rm -rf $O/st
done; done
