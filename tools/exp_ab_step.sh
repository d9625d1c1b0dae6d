cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2 3 4 5; do for L in prev.so libaligner_amd.so; do
export ALIGNER_AMD_LIB=$GRAFT_REPO_ROOT/aligner_amd/lib/$L
python3 bench.py --no-cpu-baseline --no-repeats --no-side-kernels --steps 10000 --warmup 1000 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'], d['roofline']['all_kernels']['maxpath_pipelined_kernel']['us'])"
done; done
