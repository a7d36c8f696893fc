cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_mp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mp -o r --output-format csv -- python bench.py --aggregator max_pool --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_mp.log 2>&1
find gpurun_out/prof_mp -name "*kernel_trace*.csv" -size +20M -delete
