#!/bin/bash
# performance_sweep.sh — the reference's scripts/performance.ps1 (loop 2^25 .. 2^1 over
# `--num-elements $p --perf-csv-to-stdout`) for this engine, extended to 2^30, plus what
# Performance/perfToOverallCSV.py did: every run's CSV rows folded into ONE file with a single header.
#   tools/performance_sweep.sh [max_log2=28] [min_log2=1] [out=performance.csv] [extra rsx_tests flags...]
# CPU referees are skipped above 2^CPU_MAX_LOG2 (default 24: tens of seconds per task beyond).
MAX=${1:-28}; MIN=${2:-1}; OUT=${3:-performance.csv}; shift 3 2>/dev/null
BIN=$(dirname "$0")/../radix-sort_amd/host/bin/rsx_tests
HDR=""
: > "$OUT"
for ((p=MAX; p>=MIN; p--)); do
  n=$((1 << p)); extra=""; [ $p -gt ${CPU_MAX_LOG2:-24} ] && extra="--skip-cpu"
  "$BIN" --num-elements $n --perf-csv-to-stdout $extra "$@" 2>/dev/null | awk -v out="$OUT" -v first="$([ -z "$HDR" ] && echo 1 || echo 0)" '
     /^NumElements,/ { if (first == 1 && !seen) { print >> out; seen = 1 } next_is_row = 1; next }
     next_is_row    { print >> out; next_is_row = 0 }'
  HDR=done
  echo "2^$p done: $(wc -l < "$OUT") lines" >&2
done
