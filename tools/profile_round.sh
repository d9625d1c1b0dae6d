# usage (GPU box): bash tools/profile_round.sh TAG -- the round's measurement set into gpurun_out/TAG/round
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r03}; O=gpurun_out/$T/round; mkdir -p $O
for s in 2 3 4; do python3 bench.py --streams $s --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams', d['config']['batches_in_flight'], 'ms/step', d['ms_per_step'], 'serial', d.get('serial_ms_per_step'), 'serial with the expand kernel', d.get('serial_ms_per_step_expand_kernel'), 'value', d['value'])"; done > $O/streams.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels > $O/bench_streams1.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fused -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels --path-mode fused > $O/bench_streams1_fused.json 2> $O/stats_fused.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels > /dev/null 2> $O/pmc_write.err
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json "round 3 ($T)" > $O/pmc.txt 2>&1
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_shaped.json 2> $O/bench_driver_shaped.err
python3 bench.py --config c4 --gpus 1 > $O/bench_c4_n1.json 2> $O/bench_c4.err
cat $O/streams.txt; head -30 $O/pmc.txt
for d in stats stats_fused; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; head -6 $O/${d}_kernel_stats.csv | cut -c1-160; done
