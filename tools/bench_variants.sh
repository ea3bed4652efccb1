#!/bin/bash
# tools/bench_variants.sh NAME ...: short bench of each experimental library build (see build_variant.sh)
root=$(cd "$(dirname "$0")/.." && pwd)
for v in "$@"; do
  MMC_HIP_LIB=$root/superresolution_amd/lib/variants/libmmc_$v.so timeout -k 10 300 python3 $root/bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['all_kernels_ms'].items()})" || exit 1
done
