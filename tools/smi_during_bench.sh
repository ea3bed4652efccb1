mkdir -p gpurun_out/r2c
(timeout -k 5 120 python3 bench.py --steps 4000 --warmup 20 --no-cpu-baseline --extra-flows "" > gpurun_out/r2c/long.json 2>/dev/null) &
BP=$!
sleep 12
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|socclk|Temperature \(Sensor (junction|memory)" ; echo --; sleep 0.5; done > gpurun_out/r2c/smi.log 2>&1
wait $BP
cat gpurun_out/r2c/smi.log | head -60
python3 -c "
import json; d=json.load(open('gpurun_out/r2c/long.json')); print(d['ms_per_step'], d['roofline']['all_kernels_ms'])"
