#!/bin/bash
# Runs ON THE GPU BOX (gpurun): rocprofv3 evidence for the bench workloads. Kernel trace/stats and each PMC group are
# separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with other traces).
#   bash tools/profile_round.sh r02 [regex of run names, e.g. train]
set -u
R=${1:-r02}
ONLY=${2:-.}
OUT=/root/repo/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# one handle per process here: its CU-masked session stream is destroyed with it instead of parked (rocprofv3's exit handler
# faults on a queue that is still alive -- after the output files are written, but the exit code is 139)
export DYN_DESTROY_SESSION_STREAM=1
B=/root/repo/bench.py
run() { name=$1; shift; [[ $name =~ $ONLY ]] || return 0; echo "== $name"; timeout -k 10 280 "$@" > $OUT/$name.log 2>&1 || echo "   rc=$?"; }
# (round 4) the bench's default is strict mode "ties": launches of the MIXED kernel k_read_queue<JOB_ALIGN, true>. The sub-records
# (plain_arithmetic, e2e_cli) are switched off so that every launch in a profile belongs to the headline region; the plain
# kernel k_read_queue<JOB_ALIGN, false> is profiled by its own runs (--strict off).
X="--no-cpu-baseline --no-plain --no-e2e --no-short --no-train --no-cfg3 --no-cold"
# stats runs: no warm-up, no resident leg -- every k_read_queue launch of the process belongs to the timed region, so that
# rocprofv3's Calls / TotalDurationNs are the bench line's roofline.launches / kernel_ms_total (the engine merges tickets
# that wait into one launch: launches differ in size, the totals are what has to agree)
S="$X --no-resident --no-polya --no-scale-ref --warmup 0"
X="$X --no-polya --no-scale-ref"
# (round 5) align(calc=true) batches of >= 512 reads run in the RESIDENT read queue: ONE k_session launch per session (here:
# per run) instead of a k_read_queue launch per two or three batches. stats_align_nosess is the same command with one launch
# per batch (--no-sessions), the A/B of the resident queue under the profiler.
# The counter passes (--pmc) run WITHOUT sessions: rocprofv3 serialises kernels while it collects counters, and a resident
# kernel waits for kernels of other streams (it would sit there until its idle watchdog ends it). Bytes and instructions per
# lattice cell are those of the same sweeps either way.
NS="--no-sessions"
run stats_align   rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_align   -- python3 $B $S --steps 24
run stats_align_nosess rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_align_nosess -- python3 $B $S --steps 24 $NS
run stats_align_plain rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_align_plain -- python3 $B $S --steps 24 --strict off
run stats_train   rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_train   -- python3 $B $S --steps 12 --mode train
run stats_cfg3    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg3    -- python3 $B $S --workload cfg3 --steps 4 --batches 1
run pmc_fetch_align rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch_align -- python3 $B $X --steps 2 --warmup 0 --batches 1 --depth 1 $NS
run pmc_write_align rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_align -- python3 $B $X --steps 2 --warmup 0 --batches 1 --depth 1 $NS
run pmc_sq_align  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq_align -- python3 $B $X --steps 2 --warmup 0 --batches 1 --depth 1 $NS
run pmc_fetch_align_plain rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch_align_plain -- python3 $B $X --steps 2 --warmup 0 --batches 1 --depth 1 $NS --strict off
run pmc_write_align_plain rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_align_plain -- python3 $B $X --steps 2 --warmup 0 --batches 1 --depth 1 $NS --strict off
run pmc_sq_align_plain  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq_align_plain -- python3 $B $X --steps 2 --warmup 0 --batches 1 --depth 1 $NS --strict off
run pmc_fetch_train rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch_train -- python3 $B $X --steps 2 --warmup 0 --batches 1 --depth 1 --mode train
run pmc_write_train rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_train -- python3 $B $X --steps 2 --warmup 0 --batches 1 --depth 1 --mode train
run pmc_sq_train  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq_train -- python3 $B $X --steps 2 --warmup 0 --batches 1 --depth 1 --mode train
# keep only what is small enough to travel back: stats + counter csv files
find $OUT -name "*.csv" -size +8M -delete
find $OUT -name "*kernel_trace.csv" -delete
du -sh $OUT
