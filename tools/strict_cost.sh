#!/bin/bash
# GPU box: what strict mode costs. cfg2 (1 024 reads = one read per wave: a launch lasts as long as its slowest read) and
# cfg4's share (4 096 reads per batch: the queue balances). Writes gpurun_out/$1/strict_cost_*.json
out=gpurun_out/${1:-strict}
mkdir -p $out
for w in cfg2 cfg4_share; do
  for m in off ties all; do
    python bench.py --steps 6 --warmup 2 --no-cpu-baseline --workload $w --strict $m > $out/strict_cost_${w}_$m.json 2> $out/strict_cost_${w}_$m.err || exit 1
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/strict_cost_*.json")):
    d = json.load(open(f))
    print(f.split("strict_cost_")[1][:-5], d["value"], "Msamp/s", d["ms_per_step"], "ms/step", "strict reads/step:", d.get("strict_reads_per_step"))
PY
