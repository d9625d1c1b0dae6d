set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
for s in 2 3 4; do python bench.py --streams $s --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams', d['config']['batches_in_flight'], 'ms/step', d['ms_per_step'], 'serial', d.get('serial_ms_per_step'), 'value', d['value'])"; done > gpurun_out/r02e/streams.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02e/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --streams 1 --no-repeats > gpurun_out/r02e/bench_streams1.json 2> gpurun_out/r02e/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02e/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-repeats > /dev/null 2> gpurun_out/r02e/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02e/pmc_write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-repeats > /dev/null 2> gpurun_out/r02e/pmc_write.err
python tools/pmc_traffic.py gpurun_out/r02e/pmc_fetch gpurun_out/r02e/pmc_write gpurun_out/r02e/pmc_hbm_traffic.json "round 2, second session, final build" > gpurun_out/r02e/pmc.txt 2>&1
python bench.py > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err
cat gpurun_out/r02e/streams.txt; cat gpurun_out/r02e/pmc.txt | head -30
find gpurun_out/r02e/stats -name "*kernel_stats.csv" | head
