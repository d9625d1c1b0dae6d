# usage (GPU box): bash tools/exp_env.sh VAR v1 v2 ... -- the similarity kernel under rocprofv3 with VAR set to each value in turn, twice (same box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
VAR=$1; shift
for r in 1 2; do for v in "$@"; do
export $VAR=$v
O=gpurun_out/env_$v; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --steps 3000 --warmup 300 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels > $O/b.json 2> $O/err
echo "$VAR=$v: $(grep 'softattn' $(find $O/st -name '*kernel_stats.csv' | head -1) | head -1 | sed 's/.*SoftAttnParams)",//' | cut -d, -f1-6)"; rm -rf $O/st
done; done
