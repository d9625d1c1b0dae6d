set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h/round
for s in 2 3 4; do python bench.py --streams $s --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams', d['config']['batches_in_flight'], 'ms/step', d['ms_per_step'], 'serial', d.get('serial_ms_per_step'), 'value', d['value'])"; done > gpurun_out/r02h/round/streams.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02h/round/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --streams 1 --no-repeats > gpurun_out/r02h/round/bench_streams1.json 2> gpurun_out/r02h/round/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02h/round/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-repeats > /dev/null 2> gpurun_out/r02h/round/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02h/round/pmc_write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-repeats > /dev/null 2> gpurun_out/r02h/round/pmc_write.err
python tools/pmc_traffic.py gpurun_out/r02h/round/pmc_fetch gpurun_out/r02h/round/pmc_write gpurun_out/r02h/round/pmc_hbm_traffic.json "round 2, third session, final build" > gpurun_out/r02h/round/pmc.txt 2>&1
python bench.py > gpurun_out/r02h/round/bench.json 2> gpurun_out/r02h/round/bench.err
cat gpurun_out/r02h/round/streams.txt; cat gpurun_out/r02h/round/pmc.txt | head -30
find gpurun_out/r02h/round/stats -name "*kernel_stats.csv" | head
