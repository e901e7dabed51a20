#!/usr/bin/env python
"""Rebuild the library with extra -D flags (kernel experiments) and print bench.py's kernel times.

    python tools/variant_bench.py base "" sel "-DDYN_EXP_BITS_SELECT" ...

Meant for a GPU box (gpurun): the rebuilt .so replaces the snapshot's copy there only.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(argv):
    pairs = list(zip(argv[0::2], argv[1::2]))
    for name, flags in pairs:
        env = dict(os.environ, DYN_HIPCC_EXTRA=flags)
        subprocess.run([sys.executable, "-c", "import dynamont_amd._native as n; n.build(force=True)"], cwd=ROOT, env=env, check=True)
        r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + os.environ.get("DYN_VARIANT_BENCH_ARGS", "").split(),
                           cwd=ROOT, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode or not line:
            print(name, "FAILED", r.stderr[-500:], flush=True)
            continue
        j = json.loads(line[-1])
        print(f"{name:>14s} [{flags}]: ms/step {j['ms_per_step']:.3f}  {j['kernel_ms_per_step']}  resident {j.get('kernel_resident_Msamp_s')}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
