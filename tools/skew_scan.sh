# tools/skew_scan.sh [bench args]: bench of the -DMMC_ABLATION build (tools/build_variant.sh abl -DMMC_ABLATION)
# with the shift between consecutive per-pixel planes forced to each of a list of values (MMC_PLANE_SKEW)
root=$(cd "$(dirname "$0")/.." && pwd)
for sk in 0 45056 86016 151552 200704 327680 466944 688128 1003520 0; do
  MMC_PLANE_SKEW=$sk MMC_HIP_LIB=$root/superresolution_amd/lib/variants/libmmc_abl.so timeout -k 10 200 python3 $root/bench.py --steps 40 --warmup 5 --no-cpu-baseline --extra-flows "" "$@" 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skew', $sk, round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['all_kernels_ms'].items()})" || exit 1
done
