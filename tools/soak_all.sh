# usage (GPU box): bash tools/soak_all.sh [cases] -- every randomized soak against its oracle, one after the other
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/soak
N=${1:-60}
rc=0
for t in maxpath two_cus softattn conv objective ctc mobo; do
  echo "== soak_$t"; timeout -k 10 600 python3 tools/soak_$t.py $N 7 > gpurun_out/soak/$t.log 2>&1 || rc=1
  tail -2 gpurun_out/soak/$t.log
done
exit $rc
