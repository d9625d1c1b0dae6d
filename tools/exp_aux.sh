set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in 0 17 16 1; do
export ALIGNER_SA_AUX=$a
O=gpurun_out/aux$a; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels > $O/b.json 2> $O/err
echo aux $a; head -4 $(find $O/st -name "*kernel_stats.csv" | head -1) | cut -c1-140; rm -rf $O/st
python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('3 streams: ms/step', d['ms_per_step'], 'repeats', d.get('ms_per_step_repeats'), 'serial', d.get('serial_ms_per_step'))"
python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-shaped: ms/step', d['ms_per_step'], 'serial', d.get('serial_ms_per_step'))"
done
