cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6_build.log 2>&1
( time timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_final.json 2> gpurun_out/r6_bench_final.err ) 2> gpurun_out/r6_bench_final.time
echo "bench exit $?"; cat gpurun_out/r6_bench_final.time
timeout -k 10 900 python tests/fuzz_parity.py > gpurun_out/r6_fuzz_parity.txt 2>&1; echo "fuzz exit $?"; tail -2 gpurun_out/r6_fuzz_parity.txt
