# usage (GPU box): bash tools/exp_pitch_ab.sh -- the headline step with its log-probs pitched (default) against contiguous, same box:
# rocprofv3 kernel averages (one batch at a time) and the three-batch step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2; do for f in "" "--contiguous-logp"; do
O=gpurun_out/pitch_ab; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --steps 3000 --warmup 300 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels $f > $O/b.json 2> $O/err
echo "[${f:-pitched}] $(grep 'softattn\|maxpath_pipe\|expand' $(find $O/st -name '*kernel_stats.csv' | head -1) | sed 's/(aligner::[A-Za-z]*Params)//; s/"void aligner:://; s/<[^>]*>//' | awk -F'",' '{split($2,a,","); printf "%s %.2f us | ", substr($1,1,28), a[3]/1000}')"; rm -rf $O/st
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $f 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   driver-shaped: ms/step', d['ms_per_step'], 'serial', d.get('serial_ms_per_step'), 'value', d['value'], 'guards', all(d['guards'].values()) if isinstance(d.get('guards'), dict) else d.get('guards'))"
done; done
