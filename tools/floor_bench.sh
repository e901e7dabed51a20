#!/bin/bash
# GPU box: the align kernel's floor. Same-box A/B of the product build against -DDYN_EXP_MAXPLUS (logPlus -> max: one
# operation instead of 19, no table lookup; loads, stores, ballots, hand-overs and traceback unchanged), strict mode off,
# cfg2. Prints kernel times; the rebuilt library lives on the box only.
export DYN_VARIANT_BENCH_ARGS="--steps 8 --warmup 2 --no-e2e --no-plain --strict off"
timeout -k 10 900 python tools/variant_bench.py product "" maxplus "-DDYN_EXP_MAXPLUS" product2 "" maxplus2 "-DDYN_EXP_MAXPLUS"
