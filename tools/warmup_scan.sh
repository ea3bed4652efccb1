# tools/warmup_scan.sh: the 20-step timed region after warm-ups of different length (how long the GPU takes to settle)
root=$(cd "$(dirname "$0")/.." && pwd)
for w in 5 20 50 200 5; do
  timeout -k 10 200 python3 $root/bench.py --steps 20 --warmup $w --no-cpu-baseline --extra-flows "" "$@" 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup', $w, round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['all_kernels_ms'].items()})" || exit 1
done
