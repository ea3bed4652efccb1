mkdir -p gpurun_out/r2c
run() { timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --extra-flows "" "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],4), round(d['iteration_roofline']['frac_of_8TBps'],4), {k: round(x,4) for k,x in d['roofline']['all_kernels_ms'].items()}, d['config']['plane_skew_bytes'])" || exit 1; }
run --lr 540 960 --frames-per-gpu 15
run --lr 144 180 --frames-per-gpu 13
run --lr 150 150 --frames-per-gpu 3
run --kappa 1
run --kappa nan
run --flow smooth
run --flow outliers
run --direction forward
run --direction forward-backward
