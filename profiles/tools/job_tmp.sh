bash profiles/run_pmc_train.sh tr_f32_fetch FETCH_SIZE --batch 8
bash profiles/run_pmc_train.sh tr_f32_write WRITE_SIZE --batch 8
bash profiles/run_pmc_train.sh tr_bf16_fetch FETCH_SIZE --batch 8 --bf16-mlp
bash profiles/run_pmc_train.sh tr_bf16_write WRITE_SIZE --batch 8 --bf16-mlp
bash profiles/run_pmc_train.sh tr1_f32_fetch FETCH_SIZE --batch 1
bash profiles/run_pmc_train.sh tr1_f32_write WRITE_SIZE --batch 1
python3 bench.py > gpurun_out/r4_line2.json 2> gpurun_out/r4_line2.err; tail -c 1800 gpurun_out/r4_line2.json
python3 bench.py --mode train --batch 1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4_train_b1.json
