# usage (GPU box): bash tools/profile_conv.sh TAG [conv_one.py args] -- kernel stats + SQ counters of one wide conv layer
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r04}; shift
O=gpurun_out/$T/conv; mkdir -p $O
python3 tools/conv_one.py "$@" > $O/conv_one.txt 2>&1; cat $O/conv_one.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/conv_one.py "$@" > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/conv_one_kernel_stats.csv; cut -c1-170 $O/conv_one_kernel_stats.csv | head -8
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc1 -- python3 tools/conv_one.py "$@" > $O/pmc1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $O/pmc2 -- python3 tools/conv_one.py "$@" > $O/pmc2.log 2>&1
python3 tools/pmc_issue.py $O/conv_one_pmc.json "tools/conv_one.py $*" $O/pmc1 $O/pmc2 > $O/pmc.txt 2>&1; head -90 $O/pmc.txt
rm -rf $O/stats $O/pmc1 $O/pmc2
