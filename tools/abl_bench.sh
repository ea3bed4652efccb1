#!/bin/bash
# ablation runs: MMC_DEBUG bit masks over the ablation build
root=$(cd "$(dirname "$0")/.." && pwd)
for d in "$@"; do
  MMC_DEBUG=$d MMC_HIP_LIB=$root/superresolution_amd/lib/variants/libmmc_abl.so timeout -k 10 120 python3 $root/bench.py --steps 40 --warmup 5 --no-cpu-baseline --extra-flows "" 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg=$d', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['all_kernels_ms'].items()})" || exit 1
done
