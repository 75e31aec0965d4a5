#!/bin/bash
# Regenerates every round-6 profile artifact in one gpurun call (run from the repo root on the GPU box):
#   gpurun --timeout 2400 -- 'bash profiles/regen_r6.sh <commit>'
# then, back in the container:  python profiles/collect_r6.py     (copies the summaries from gpurun_out/ into profiles/)
# Passes: kernel-trace stats of the default (pipelined) and the serial bench and of the batch-8 / batch-1 training steps; FETCH_SIZE /
# WRITE_SIZE / SQ counter passes of the serial forward and FETCH_SIZE / WRITE_SIZE passes of the training steps (each --pmc pass on its
# own, no trace domains, the program directly after `--`); then the bench lines with the fresh PMC traffic in place.
set -u
export PS_PROFILE_ROUND=r6
COMMIT=${1:-unknown}
TOPN=6 bash profiles/run_kernel_stats.sh pipe --steps 200 --warmup 20 --no-sub-results
TOPN=6 bash profiles/run_kernel_stats.sh serial --no-pipeline --steps 30 --warmup 3
TOPN=6 bash profiles/run_kernel_stats.sh train_b8 --mode train --batch 8 --steps 2 --warmup 1
TOPN=6 bash profiles/run_kernel_stats.sh train_b8_bf16 --mode train --batch 8 --bf16-mlp --steps 2 --warmup 1
TOPN=6 bash profiles/run_kernel_stats.sh train_b1 --mode train --batch 1 --steps 4 --warmup 2
bash profiles/run_pmc.sh fetch FETCH_SIZE --steps 5 --warmup 1
bash profiles/run_pmc.sh write WRITE_SIZE --steps 5 --warmup 1
bash profiles/run_pmc.sh sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --steps 3 --warmup 1
python3 profiles/make_pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write r6 "$COMMIT" "python bench.py --no-pipeline --no-cpu-baseline --no-stage-timing --steps 5 --warmup 1" > gpurun_out/pmc_summary.txt
cp profiles/r6_pmc_per_kernel.json profiles/r6_pmc_traffic.json gpurun_out/
python3 profiles/summarize_sq.py gpurun_out/pmc_sq > gpurun_out/r6_sq_counters.txt
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
cp profiles/r6_pmc_traffic_train.json profiles/r6_pmc_train_per_kernel_*.json gpurun_out/
python3 bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/r6_bench_line.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r6_bench_line_driver_form.json
python3 bench.py --steps 30 --warmup 3 --no-pipeline --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_bench_line_serial.json
python3 bench.py --workload config5 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_bench_line_config5.json
python3 bench.py --batch 2 --steps 100 --warmup 20 --no-cpu-baseline --no-sub-results 2>/dev/null | tail -1 > gpurun_out/r6_bench_line_batch2.json
python3 bench.py --mode train --batch 8 --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r6_bench_line_train_b8.json
python3 bench.py --mode train --batch 8 --bf16-mlp --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_bench_line_train_b8_bf16.json
python3 bench.py --mode train --batch 8 --atomic-scatter --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_bench_line_train_b8_atomic_scatter.json
python3 bench.py --mode train --batch 1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_bench_line_train_b1.json
python3 bench.py --gpus 2 --share-gpu --dist-backend gloo --steps 20 --warmup 5 --no-cpu-baseline --no-sub-results 2>/dev/null | tail -1 > gpurun_out/r6_bench_line_2ranks_one_gpu_gloo.json
python3 bench.py --gpus 8 --share-gpu --dist-backend gloo --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_bench_line_8ranks_one_gpu_gloo.json
ls -la gpurun_out | head -80
