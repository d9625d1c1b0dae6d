# usage (GPU box): bash tools/profile_encoders.sh TAG -- per-kernel times of the two encoder stacks of BASELINE configs[2]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r04}; O=gpurun_out/$T/enc; mkdir -p $O
python3 tools/encoder_times.py > $O/encoder_times.txt 2>&1; grep -v amdgpu.ids $O/encoder_times.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/encoder_times.py > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/encoders_kernel_stats.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/encoders_kernel_stats.csv")):
    n = r["Name"]
    if "conv" in n: print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>4}  {n[:110]}')
PY
rm -rf $O/stats
