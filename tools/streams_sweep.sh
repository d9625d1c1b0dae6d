# usage (GPU box): bash tools/streams_sweep.sh -- batches in flight x hardware queues (GPU_MAX_HW_QUEUES), ms per step
cd $GRAFT_REPO_ROOT
for q in default 2 4 8; do
  for s in 2 3 4 5 6; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    python3 bench.py --streams $s --no-cpu-baseline --no-side-kernels 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $q streams', d['config']['batches_in_flight'], 'ms/step', d['ms_per_step'], 'repeats', d.get('ms_per_step_repeats'), 'value', d['value'])"
  done
done
