cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r --output-format csv -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/prof.log 2>&1
python tools/timeline.py gpurun_out/prof/r_kernel_trace.csv 40 k_gather_mean_multi | tail -16
