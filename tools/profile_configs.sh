# usage (on the GPU box): bash tools/profile_configs.sh OUTDIR -- kernel stats of every configuration + in-kernel stamps
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/configs}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/config_times.py > $O/config_times.txt 2> $O/config_times.err
for dt in f32 bf16; do for fl in 0 256; do echo "== [8,500,4000] $dt flags $fl"; STAMP_DTYPE=$dt STAMP_FLAGS=$fl python tools/stamps.py 8 500 4000 2>&1 | grep -v amdgpu.ids; done; done > $O/dp_stamps_c5.txt
python tools/stamps.py 2>&1 | grep -v amdgpu.ids > $O/dp_stamps_c2.txt
cat $O/config_times.txt; find $O/stats -name "*kernel_stats.csv" | head -2
