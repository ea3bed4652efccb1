# tools/skew_scan.sh "SKEWS" [bench args]: bench with the shift between consecutive per-pixel planes forced
# to each of the listed byte values (MMC_PLANE_SKEW), then three runs with the probe choosing (timings traced)
root=$(cd "$(dirname "$0")/.." && pwd)
skews=$1; shift
for sk in $skews; do
  MMC_PLANE_SKEW=$sk timeout -k 10 200 python3 $root/bench.py --steps 40 --warmup 5 --no-cpu-baseline --extra-flows "" "$@" 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skew', $sk, round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['all_kernels_ms'].items()})" || exit 1
done
for i in 1 2 3; do
  MMC_PROBE_TRACE=1 timeout -k 10 200 python3 $root/bench.py --steps 40 --warmup 5 --no-cpu-baseline --extra-flows "" "$@" 2> /tmp/probe.err \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('probe', d['config']['plane_skew_bytes'], round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['all_kernels_ms'].items()})" || exit 1
  grep "plane shift" /tmp/probe.err
done
