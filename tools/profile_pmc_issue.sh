# usage (GPU box): bash tools/profile_pmc_issue.sh TAG -- SQ issue counters of the headline step's kernels (bench.py, one batch in
# flight, eager-equivalent: --streams 1 --no-repeats --no-side-kernels) -> gpurun_out/TAG/pmc_issue_counters.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r05}; O=gpurun_out/$T; mkdir -p $O
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --no-repeats --no-side-kernels"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $O/pmcA -- $B > /dev/null 2> $O/pmcA.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmcB -- $B > /dev/null 2> $O/pmcB.err
python3 tools/pmc_issue.py $O/pmc_issue_counters.json "rocprofv3 --pmc (two passes) of: $B" $O/pmcA $O/pmcB > $O/pmc_issue.txt 2>&1
python3 - <<PY
import json
d = json.load(open("$O/pmc_issue_counters.json"))["kernels"]
for k in ("maxpath_pipelined_kernel", "softattn_rt_kernel", "softattn_kernel", "expand_kernel"):
    v = d.get(k, {})
    print(k, {x: v.get(x) for x in ("SQ_WAVES", "kernel_cycles", "resident_waves", "wave_occupancy", "wait_any_share", "issue_stall_share", "valu_per_wave", "mfma_busy_share")})
PY
rm -rf $O/pmcA $O/pmcB
