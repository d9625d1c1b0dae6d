# usage (GPU box): bash tools/exp_c3_ab.sh "opts A" "opts B" ... -- configs[2] under rocprofv3 with ALIGNER_DEBUG_OPTIONS set to each argument in turn
# ("-" = none), twice: the conv kernels' average times and the step, same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "$@"; do
if [ "$v" = "-" ]; then unset ALIGNER_DEBUG_OPTIONS; else export ALIGNER_DEBUG_OPTIONS=$v; fi
O=gpurun_out/c3ab; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --config c3 --no-cpu-baseline > $O/b.json 2> $O/err
echo "[$v] $(grep -o 'ms_per_step": [0-9.]*' $O/b.json) $(grep 'conv_narrow_ring\|conv_narrow_kernel<1, 2, 1, false>\|conv_narrow_fused\|conv_gemm_kernel<3, 13, true>' $(find $O/st -name '*kernel_stats.csv' | head -1) | sed 's/(aligner::[A-Za-z]*Params)//; s/"void aligner:://' | cut -d, -f1-6 | awk -F'",' '{split($2,a,","); printf "%s %s x %.1f us | ", $1, a[1], a[3]/1000}')"; rm -rf $O/st
done; done
