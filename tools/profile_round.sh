#!/bin/bash
# tools/profile_round.sh OUTDIR: kernel-trace statistics of the default bench command, then the
# hardware-counter passes (tools/pmc_collect.sh).  Every step has its own time limit; stops at the
# first failure.
set -u
out=$1
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$out" && out=$(cd "$out" && pwd)
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o vm --output-format csv -- python3 "$root/bench.py" --steps 20 --warmup 5 > "$out/bench_traced.json" 2> "$out/bench_traced.err" || { echo "kernel trace failed"; tail -n 5 "$out/bench_traced.err"; exit 1; }
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
rm -rf "$out/trace"
"$root/tools/pmc_collect.sh" "$out/pmc" 30 4 || exit 1
echo "profile_round done"
