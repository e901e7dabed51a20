#!/bin/bash
# GPU box: what strict mode costs. cfg2 (1 024 reads per batch) and cfg4's share (4 096 reads per batch), each in the plain
# arithmetic (off), the default (ties: reads with identical neighbouring columns run the certified sweeps) and all (every
# read certified). Writes gpurun_out/$1/strict_cost_*.json and the summary gpurun_out/$1/strict_mode_cost.json
out=gpurun_out/${1:-strict}
mkdir -p $out
for w in cfg2 cfg4_share; do
  for m in off ties all; do
    python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-plain --no-polya --no-scale-ref --no-e2e --workload $w --strict $m \
      > $out/strict_cost_${w}_$m.json 2> $out/strict_cost_${w}_$m.err || exit 1
  done
done
python - <<PY
import json, glob
res = {}
for f in sorted(glob.glob("$out/strict_cost_*_*.json")):
    d = json.load(open(f))
    name = f.split("strict_cost_")[1][:-5]
    res[name] = {"value_Msamp_s": d["value"], "ms_per_step": d["ms_per_step"], "kernel_ms_per_step": d["kernel_ms_per_step"],
                 "strict_reads_per_step": d.get("strict_reads_per_step"), "certified_rows_per_step": d.get("certified_rows_per_step"),
                 "certificate_fallbacks_per_row": d.get("certificate_fallbacks_per_row"), "wave_occupancy": d["roofline"].get("wave_occupancy"),
                 "reads_per_batch": d["config"]["reads_per_batch"]}
    print(name, d["value"], "Msamp/s", d["ms_per_step"], "ms/step", "strict reads/step:", d.get("strict_reads_per_step"))
# per strict read, from the wave-time shares of the mixed run (mode ties): time a flagged read spends in a sweep over the time
# an unflagged read of the same batch spends there
for w in ("cfg2", "cfg4_share"):
    r = res[w + "_ties"]; k = r["kernel_ms_per_step"]; ns = r["strict_reads_per_step"]; n = r["reads_per_batch"]
    per = {}
    for s in ("backward", "forward"):
        strict = k["ms_%s_strict" % s] / ns
        plain = (k["ms_" + s] - k["ms_%s_strict" % s]) / (n - ns)
        per[s] = round(strict / plain, 3)
    tr = k["ms_trace"] / n
    strict = (k["ms_backward_strict"] + k["ms_forward_strict"]) / ns + tr
    plain = (k["ms_backward"] + k["ms_forward"] - k["ms_backward_strict"] - k["ms_forward_strict"]) / (n - ns) + tr
    per["read"] = round(strict / plain, 3)
    res[w + "_per_strict_read_in_mode_ties"] = per
    res[w + "_all_over_off"] = round(res[w + "_off"]["value_Msamp_s"] / res[w + "_all"]["value_Msamp_s"], 3)
    print(w, "per strict read:", per, " all/off:", res[w + "_all_over_off"])
res["_note"] = ("tools/strict_cost.sh on one MI355X box, 12 steps each, batches in the resident read queue. *_per_strict_read_in_mode_ties: wave time of a "
                "flagged read over an unflagged read of the same batches, per sweep and per read (sweeps + traceback); *_all_over_off: throughput of the "
                "plain arithmetic over the one with every read certified")
json.dump(res, open("$out/strict_mode_cost.json", "w"), indent=1)
PY
