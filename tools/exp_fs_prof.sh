cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/fs_prof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 tools/fwdsum_serial_time.py > $O/out.txt 2> $O/err
head -12 $(find $O/st -name "*kernel_stats.csv" | head -1) | cut -c1-170 > gpurun_out/fs_prof.txt; rm -rf $O/st
