#!/bin/bash
# tools/abl_small.sh MASK ...: the ablation build (tools/build_variant.sh abl -DMMC_ABLATION) on the quick-start
# size (150 x 150 low-res, 3 frames), one MMC_DEBUG bit mask per run: what each phase of a sweep costs when all
# workgroups are resident at once.  ABLLIB=<name> picks another variant build than 'abl'.
root=$(cd "$(dirname "$0")/.." && pwd)
for d in "$@"; do
  MMC_DEBUG=$d MMC_HIP_LIB=$root/superresolution_amd/lib/variants/libmmc_${ABLLIB:-abl}.so timeout -k 10 120 python3 $root/bench.py --steps 400 --warmup 20 --no-cpu-baseline --extra-flows "" --lr 150 150 --frames-per-gpu 3 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg=$d', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['all_kernels_ms'].items()})" || exit 1
done
