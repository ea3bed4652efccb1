root=$(pwd)
for k in $(seq 0 31); do sk=$((k*16384+4096))
  MMC_PLANE_SKEW=$sk MMC_HIP_LIB=$root/superresolution_amd/lib/variants/libmmc_abl.so timeout -k 10 200 python3 $root/bench.py --steps 30 --warmup 5 --no-cpu-baseline --extra-flows "" "$@" 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skew', $sk, round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['all_kernels_ms'].items()})" || exit 1
done
