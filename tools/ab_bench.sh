#!/bin/bash
# Runs ON THE GPU BOX: A/B of the library in build/prev (a git worktree of an earlier commit, built in the authoring
# container) against the current tree, interleaved on the same GPU (box-to-box spread is ~3 %).
#   bash tools/ab_bench.sh <outdir> [bench args...]
OUT=$1; shift
mkdir -p $OUT
for rep in 1 2; do
  (cd build/prev && timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > ../../$OUT/prev_$rep.json 2> ../../$OUT/prev_$rep.err)
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $OUT/new_$rep.json 2> $OUT/new_$rep.err
done
python - "$OUT" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value', j['value'], 'resident', j.get('kernel_resident_Msamp_s'), j.get('kernel_ms_per_step'))
    except Exception as e:
        print(f, 'ERR', e)
PY
