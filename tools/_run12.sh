cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6_build.log 2>&1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r6_gpu_suite_1.log 2>&1
echo "exit $?" >> gpurun_out/r6_gpu_suite_1.log
tail -8 gpurun_out/r6_gpu_suite_1.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
