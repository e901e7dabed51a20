cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6_build.log 2>&1
sleep 12
timeout -k 10 300 python tools/cold_start_trace.py > gpurun_out/r6_cold_trace_1.txt 2>&1
echo "trace exit $?"
