python -m pytest tests/test_gpu_knn.py -m gpu -x -q 2>&1 | tail -3
python3 bench.py --no-cpu-baseline --no-sub-results --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/exp_pipe.json
python3 -c "
import json
d=json.load(open('gpurun_out/exp_pipe.json')); print(d['ms_per_step'], d['serial_ms_per_cloud'], d['roofline']['avg_launch_ms'])
print({s['name']:s['ms_per_step'] for s in d['stages']})
"
