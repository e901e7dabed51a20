#!/bin/bash
# The host side of the library under the sanitizers, on the CPU (no GPU needed; ~15 minutes on 8 cores):
#   1. ASan + UBSan and TSan builds of every .cpp translation unit (Makefile beside this script)
#   2. the CPU tests of the file readers, the formatter and the sink against those builds
#      (DYN_LIB_PATH selects the library dynamont_amd/_native.py loads; the sanitizer runtime is preloaded into python)
#   3. the byte-level fuzz of BGZF / BAM records / VBZ chunks / model TSV / CSV rows (fuzz_host.cpp), >= 1e5 mutations
# Output: $OUT/sanitizers.txt (default profiles/r06) (summary) + fuzz_asan_ubsan.txt / fuzz_tsan.txt. Exit code 0 = no report anywhere.
set -u
cd "$(dirname "$0")/../.."
OUT=${OUT:-profiles/r06}
mkdir -p "$OUT"
N_ASAN=${N_ASAN:-40000}
N_TSAN=${N_TSAN:-4000}
TESTS="tests/test_bam_reader.py tests/test_pod5_native.py tests/test_format_pinning.py tests/test_harness.py tests/test_abi_host.py"
fail=0
make -s -C tools/sanitize -j"$(nproc)" all || exit 2
# (no pipeline around the block: a `{ ...; } | tee` runs it in a subshell and `fail` would stay 0 out here)
exec > >(tee "$OUT/sanitizers.txt") 2>&1
{
  echo "# host-side sanitizer runs ($(date -u +%Y-%m-%dT%H:%MZ), $(g++ --version | head -1))"
  # The Python tests run on the ASan + UBSan build only: CPython under a preloaded TSan runtime deadlocks in its own start-up
  # on this image (tried, round 5); the thread sanitizer sees the library through fuzz_host_tsan -- the BAM reader's inflate
  # threads on every mutated file, the VBZ decoder, and the sink's sink / compress / writer threads fed by four producers.
  # (libstdc++ is preloaded too: python does not link it, and ASan resolves its __cxa_throw interceptor when it starts.)
  rt="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so)"
  echo "## asan+ubsan: pytest -m 'not gpu' $TESTS"
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD="$rt" \
    DYN_LIB_PATH="$PWD/build/sanitize/libdynamont_mi_asan.so" timeout 1500 python -m pytest $TESTS -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -4
  rc=${PIPESTATUS[0]}
  echo "exit code $rc"
  [ "$rc" = 0 ] || fail=1
  echo "## asan+ubsan: fuzz_host tests/fuzz_corpus $N_ASAN (seed 5)"
  ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 timeout 3000 build/sanitize/fuzz_host_asan tests/fuzz_corpus "$N_ASAN" 5 > "$OUT/fuzz_asan_ubsan.txt" 2>&1
  rc=$?
  tail -6 "$OUT/fuzz_asan_ubsan.txt"
  echo "exit code $rc"
  [ "$rc" = 0 ] || fail=1
  echo "## tsan: fuzz_host tests/fuzz_corpus $N_TSAN (seed 6)"
  TSAN_OPTIONS="halt_on_error=1" timeout 3000 build/sanitize/fuzz_host_tsan tests/fuzz_corpus "$N_TSAN" 6 > "$OUT/fuzz_tsan.txt" 2>&1
  rc=$?
  tail -6 "$OUT/fuzz_tsan.txt"
  echo "exit code $rc"
  [ "$rc" = 0 ] || fail=1
  echo "## result: $([ $fail = 0 ] && echo 'no sanitizer report, every run exit code 0' || echo 'FAILURES above')"
}
exit $fail
