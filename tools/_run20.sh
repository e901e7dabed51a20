cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6_build.log 2>&1
timeout -k 10 1150 python tests/parity_full.py --workload cfg4_share --out gpurun_out/r6_parity_cfg4_share.json --procs 16 > gpurun_out/r6_parity_cfg4_share.log 2>&1; echo "exit $?"; tail -2 gpurun_out/r6_parity_cfg4_share.log | cut -c1-600
