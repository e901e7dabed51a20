#!/bin/bash
# Runs ON THE GPU BOX: package power and shader clock (rocm-smi once per second) while a bench.py run is in flight.
#   bash tools/power_clock.sh <out.txt> [bench args...]     e.g.  --mode train --steps 300
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
echo "# package power and shader clock while: bench.py --no-cpu-baseline $*" > "$OUT"
timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > "$OUT.bench.json" 2> "$OUT.bench.err" &
BP=$!
t=0
while kill -0 $BP 2>/dev/null; do
  t=$((t + 1))
  line=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power|Average Graphics Package Power" | sed -E 's/.*sclk clock level: [0-9]+: //; s/.*Power \(W\): //' | tr '\n' ' ')
  echo "t=$t $line" >> "$OUT"
  sleep 1
done
wait $BP
tail -c 400 "$OUT.bench.json" >> "$OUT"
