#!/bin/bash
# tools/bench_small_frames.sh: iteration and sweep times of the quick-start frame size (150 x 150 low-res) with
# 2, 3, 6, 12, 24 frames: what part of a sweep is a fixed latency and what part grows with the workgroups.
root=$(cd "$(dirname "$0")/.." && pwd)
for L in 2 3 6 12 24; do
  timeout -k 10 200 python3 $root/bench.py --steps 400 --warmup 20 --no-cpu-baseline --extra-flows "" --lr 150 150 --frames-per-gpu $L 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames $L', round(d['ms_per_step'],4), round(d['iteration_roofline']['frac_of_8TBps'],4), {k: round(x,4) for k,x in d['roofline']['all_kernels_ms'].items()})" || exit 1
done
