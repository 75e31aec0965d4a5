#!/bin/bash
# The training-step counter passes of profiles/regen_r5.sh on their own (FETCH_SIZE / WRITE_SIZE of the batch-8 fp32 / bf16-MLP and the
# batch-1 step): gpurun --timeout 1500 -- 'bash profiles/pmc_train_r5.sh <commit>'
set -u
export PS_PROFILE_ROUND=r5
COMMIT=${1:-unknown}
bash profiles/run_pmc_train.sh tr_f32_fetch FETCH_SIZE --batch 8
bash profiles/run_pmc_train.sh tr_f32_write WRITE_SIZE --batch 8
bash profiles/run_pmc_train.sh tr_bf16_fetch FETCH_SIZE --batch 8 --bf16-mlp
bash profiles/run_pmc_train.sh tr_bf16_write WRITE_SIZE --batch 8 --bf16-mlp
bash profiles/run_pmc_train.sh tr1_f32_fetch FETCH_SIZE --batch 1
bash profiles/run_pmc_train.sh tr1_f32_write WRITE_SIZE --batch 1
T="--steps 2 --warmup 1 --no-cpu-baseline --no-stage-timing"
python3 profiles/make_pmc_train_summary.py b8_f32 gpurun_out/pmc_tr_f32_fetch gpurun_out/pmc_tr_f32_write "$COMMIT" "python3 bench.py --mode train --batch 8 $T" \
    b8_bf16 gpurun_out/pmc_tr_bf16_fetch gpurun_out/pmc_tr_bf16_write "$COMMIT" "python3 bench.py --mode train --batch 8 --bf16-mlp $T" \
    b1_f32 gpurun_out/pmc_tr1_f32_fetch gpurun_out/pmc_tr1_f32_write "$COMMIT" "python3 bench.py --mode train --batch 1 $T" > gpurun_out/pmc_train_summary.txt
cp profiles/r5_pmc_traffic_train.json profiles/r5_pmc_train_per_kernel_*.json gpurun_out/
cat gpurun_out/pmc_train_summary.txt
