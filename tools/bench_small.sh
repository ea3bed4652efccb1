#!/bin/bash
# tools/bench_small.sh [LIB]: iteration time of the small workloads (the reference's quick-start size and the
# 'calendar' size) with the given build of the library (default: the in-tree one)
root=$(cd "$(dirname "$0")/.." && pwd)
lib=${1:-$root/superresolution_amd/lib/libmmc_hip.so}
run() { MMC_HIP_LIB=$lib timeout -k 10 200 python3 $root/bench.py --steps 400 --warmup 20 --no-cpu-baseline --extra-flows "" "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $lib) $*', round(d['ms_per_step'],4), round(d['iteration_roofline']['frac_of_8TBps'],4), {k: round(x,4) for k,x in d['roofline']['all_kernels_ms'].items()})" || exit 1; }
run --lr 150 150 --frames-per-gpu 3
run --lr 144 180 --frames-per-gpu 13
run --lr 150 150 --frames-per-gpu 3 --kappa nan
