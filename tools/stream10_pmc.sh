#!/bin/bash
# tools/stream10_pmc.sh OUTDIR: FETCH_SIZE and WRITE_SIZE per launch of the kernels of tools/bin/stream_bench10
# (two launches against one launch in ticket order), one counter per run
set -u
out=$1
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$out" && out=$(cd "$out" && pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c -d "$out/$c" -o vm --output-format csv -- "$root/tools/bin/stream_bench10" 30 > "$out/$c.log" 2>&1 || { echo "pass $c failed"; tail -n 5 "$out/$c.log"; exit 1; }
  python3 "$root/tools/pmc_summary.py" "$out/$c" > "$out/$c.json" || exit 1
  rm -rf "$out/$c"
done
echo "stream10 passes done"
