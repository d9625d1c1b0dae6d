# usage (on the GPU box): bash tools/profile_mobo.sh OUTDIR -- kernel stats of the boundary search at [8,500,4000]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/mobo}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/mobo_time.py > $O/mobo_time.txt 2> $O/mobo_time.err
cat $O/mobo_time.txt
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; cat $O/kernel_stats.csv
