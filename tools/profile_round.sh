# usage (GPU box): bash tools/profile_round.sh TAG -- the round's measurement set into gpurun_out/TAG/round
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r05}; O=gpurun_out/$T/round; mkdir -p $O
stats() { f=$(find $O/$1 -name "*kernel_stats.csv" | head -1); cp $f $O/$2; rm -rf $O/$1; head -7 $O/$2 | cut -c1-170; }
# ---- headline (configs[1]) ----
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_shaped.json 2> $O/bench_driver_shaped.err
for s in 2 3 4; do python3 bench.py --streams $s --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams', d['config']['batches_in_flight'], 'ms/step', d['ms_per_step'], 'serial', d.get('serial_ms_per_step'), 'value', d['value'])"; done > $O/streams_sweep.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 3000 --warmup 300 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels > $O/bench_streams1.json 2> $O/stats.err
stats stats bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels > /dev/null 2> $O/pmc_write.err
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json "($T)" > $O/pmc.txt 2>&1; rm -rf $O/pmc_fetch $O/pmc_write
# ---- configs[2]: the full pipeline ----
python3 bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 bench.py --config c3 --no-cpu-baseline > /dev/null 2> $O/stats_c3.err
stats stats_c3 bench_c3_kernel_stats.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmcA -- python3 bench.py --config c3 --no-cpu-baseline --steps 10 > /dev/null 2> $O/pmcA.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d $O/pmcB -- python3 bench.py --config c3 --no-cpu-baseline --steps 10 > /dev/null 2> $O/pmcB.err
python3 tools/pmc_issue.py $O/bench_c3_pmc_issue.json "rocprofv3 --pmc (two passes) of: python3 bench.py --config c3 --no-cpu-baseline --steps 10" $O/pmcA $O/pmcB > $O/pmc_c3.txt 2>&1; rm -rf $O/pmcA $O/pmcB
# ---- configs[3] on one GPU, configs[4] ----
python3 bench.py --config c4 --gpus 1 > $O/bench_c4_n1.json 2> $O/bench_c4.err
python3 bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python3 bench.py --config c5 --no-cpu-baseline > /dev/null 2> $O/stats_c5.err
stats stats_c5 bench_c5_kernel_stats.csv
python3 tools/c5_stamps.py > $O/dp_stamps_c5.txt 2>&1 || true
python3 tools/sa_rt_stamps.py > $O/softattn_stamps.txt 2>&1 || true
python3 tools/dropin_time.py > $O/dropin_times.txt 2>&1 || true
python3 tools/fwdsum_serial_time.py > $O/fwdsum_serial_times.txt 2>&1 || true
python3 tools/c5_zero_blocks.py > $O/c5_zero_workgroups.txt 2>&1 || true
python3 tools/conv_fused_stamps.py > $O/conv_fused_stamps.txt 2>&1 || true
python3 tools/config_times.py > $O/config_times.txt 2>&1 || true
cat $O/streams_sweep.txt; head -30 $O/pmc.txt; python3 -c "
import json
for f in ('bench','bench_driver_shaped','bench_c3','bench_c4_n1','bench_c5'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['ms_per_step'], d['value'], d.get('serial_ms_per_step'), (d.get('guards') or ''))
"
