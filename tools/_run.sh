cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python bench.py --no-cpu-baseline --aggregator max_pool --steps 50 --warmup 5 > gpurun_out/bench_maxpool.log 2>&1; tail -1 gpurun_out/bench_maxpool.log | cut -c1-200; tail -1 gpurun_out/bench_maxpool.log | python -c "import sys,json; print(json.loads(sys.stdin.read())['roofline'])"
rm -rf gpurun_out/prof_mp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mp -o r --output-format csv -- python bench.py --aggregator max_pool --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_mp.log 2>&1
python tools/timeline.py gpurun_out/prof_mp/r_kernel_trace.csv 60 k_gather_multi_adam > gpurun_out/prof_mp_timeline.txt
find gpurun_out/prof_mp -name "*kernel_trace*.csv" -size +20M -delete
