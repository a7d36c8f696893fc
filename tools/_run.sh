cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
python bench.py --no-cpu-baseline 2>&1 | grep metric | cut -c1-200
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r --output-format csv -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/prof.log 2>&1
python tools/timeline.py gpurun_out/prof/r_kernel_trace.csv 40 k_gather_multi_adam | tail -6
