cd $GRAFT_REPO_ROOT
for c in 4 16 1; do echo "=== chunk $c GiB"; timeout -k 10 200 ./tools/ubench/vmm_probe 112 $c; done > gpurun_out/r6_vmm_probe.txt 2>&1
echo "exit $?" >> gpurun_out/r6_vmm_probe.txt
