# usage (GPU box): bash tools/soak_long.sh SEED -- longer soaks of the code this round touched
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/soak_long
S=${1:-101}; rc=0
for t in conv objective ctc; do
  echo "== soak_$t 400 cases seed $S"; timeout -k 10 900 python3 tools/soak_$t.py 400 $S > gpurun_out/soak_long/$t.log 2>&1 || rc=1
  tail -2 gpurun_out/soak_long/$t.log
done
for t in maxpath two_cus softattn mobo; do
  echo "== soak_$t 150 cases seed $S"; timeout -k 10 600 python3 tools/soak_$t.py 150 $S > gpurun_out/soak_long/$t.log 2>&1 || rc=1
  tail -2 gpurun_out/soak_long/$t.log
done
exit $rc
