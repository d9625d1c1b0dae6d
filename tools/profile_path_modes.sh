# usage (GPU box): bash tools/profile_path_modes.sh OUTDIR -- one batch at a time, the three ways of writing the dense path
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/pathmodes}; mkdir -p $O
for m in expand branch inline; do
  python3 bench.py --streams 1 --path-mode $m --no-repeats --no-cpu-baseline --steps 100 --warmup 10 > $O/bench_$m.json 2> $O/bench_$m.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -- python3 bench.py --streams 1 --path-mode $m --no-repeats --no-cpu-baseline --steps 100 --warmup 10 > /dev/null 2> $O/prof_$m.err
  f=$(find $O/stats_$m -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$m.csv
  python3 - <<PY
import json,csv
d=json.loads(open("$O/bench_$m.json").read().strip().splitlines()[-1])
print("$m: ms_per_step", d["ms_per_step"])
for r in csv.DictReader(open("$O/kernel_stats_$m.csv")):
    n=r["Name"]
    if any(k in n for k in ("softattn","maxpath","expand","zero_path","scatter")): print("   %-60s calls %5s avg %8.2f us min %8.2f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
