python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --mode train --batch 1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4_bench_line_train_b1_rccl_world1.json
python3 -c "
import json
d=json.load(open('gpurun_out/r4_bench_line_train_b1_rccl_world1.json'))
print({k:d[k] for k in ('ms_per_step','collectives_per_step','collective_bytes_per_step','collective_ms','collective_host_ms','launches_per_step')})
print(d['sections'])"
