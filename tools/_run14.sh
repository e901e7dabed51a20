cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6_build.log 2>&1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=25 > gpurun_out/r6_gpu_suite_3.log 2>&1
echo "exit $?" >> gpurun_out/r6_gpu_suite_3.log
grep -A30 "slowest" gpurun_out/r6_gpu_suite_3.log | head -40
