#!/bin/bash
# tools/bench_block_sizes.sh: iteration time against the number of 1080p frames in the block of one GPU (what a
# caller sizing blocks for several GPUs, or batches for one, wants to know)
root=$(cd "$(dirname "$0")/.." && pwd)
for L in 4 8 15 30 60 120 240; do
  timeout -k 10 300 python3 $root/bench.py --steps 60 --warmup 10 --no-cpu-baseline --extra-flows "" --frames-per-gpu $L 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames $L', round(d['ms_per_step'],4), 'ms,', round(d['ms_per_step']/$L*1e3,2), 'us per frame,', round(d['value']), 'Mpx*it/s,', round(d['iteration_roofline']['frac_of_8TBps'],4))" || exit 1
done
