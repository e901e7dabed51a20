cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6_build.log 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_resident_queue.py -x -q -m gpu -k "huge or getters" > gpurun_out/r6_t9.log 2>&1; tail -5 gpurun_out/r6_t9.log
( time timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_3.json 2> gpurun_out/r6_bench_3.err ) 2> gpurun_out/r6_bench_3.time
echo "bench exit $?"; cat gpurun_out/r6_bench_3.time
