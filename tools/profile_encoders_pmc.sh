# usage (GPU box): bash tools/profile_encoders_pmc.sh TAG -- SQ counters of the encoder stacks' kernels (tools/encoder_times.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r04}; O=gpurun_out/$T/enc; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc1 -- python3 tools/encoder_times.py > $O/pmc1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc2 -- python3 tools/encoder_times.py > $O/pmc2.log 2>&1
python3 tools/pmc_issue.py $O/encoders_pmc.json "tools/encoder_times.py: BASELINE configs[2]'s encoder stacks" $O/pmc1 $O/pmc2 > $O/pmc.txt 2>&1; cat $O/pmc.txt
rm -rf $O/pmc1 $O/pmc2
