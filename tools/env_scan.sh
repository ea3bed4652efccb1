# tools/env_scan.sh VAR "VALUES" [bench args]: short bench with the environment variable VAR set to each value
root=$(cd "$(dirname "$0")/.." && pwd)
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v timeout -k 10 200 python3 $root/bench.py --steps 40 --warmup 5 --no-cpu-baseline --extra-flows "" "$@" 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var', '$v', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['all_kernels_ms'].items()})" || exit 1
done
