# usage (GPU box): bash tools/exp_c3_ab.sh "opts A" "opts B" ... -- configs[2] under rocprofv3 with ALIGNER_DEBUG_OPTIONS set to each argument in turn
# ("-" = none), twice: the conv kernels' average times and the step, same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "$@"; do
if [ "$v" = "-" ]; then unset ALIGNER_DEBUG_OPTIONS; else export ALIGNER_DEBUG_OPTIONS=$v; fi
O=gpurun_out/c3ab; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --config c3 --no-cpu-baseline > $O/b.json 2> $O/err
python3 - "$v" $O/b.json $(find $O/st -name '*kernel_stats.csv' | head -1) <<'PY'
import csv, json, sys
tag, bj, st = sys.argv[1:4]
try: ms = json.load(open(bj))["ms_per_step"]
except Exception: ms = None
rows = [r for r in csv.DictReader(open(st)) if "conv_" in r["Name"] or "softattn" in r["Name"] or "maxpath" in r["Name"]]
print(f"[{tag}] ms/step {ms} | " + " | ".join(f'{r["Name"].split("(")[0].replace("void aligner::", "")[:44]} x{r["Calls"]} {float(r["AverageNs"]) / 1e3:.1f} us' for r in rows[:9]))
PY
rm -rf $O/st
done; done
