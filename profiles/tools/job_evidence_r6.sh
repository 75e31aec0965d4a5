#!/bin/bash
# Round-6 evidence passes (one gpurun call): counters of the deep dense layers, the K-NN write side A/B (item 1d), the
# 8-ranks-on-one-GPU gloo bench line (item 5b), the default bench line with its new diagnostics (items 7, 8).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash profiles/run_pmc.sh g32_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" --steps 5 --warmup 1
bash profiles/run_pmc.sh g32_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" --steps 3 --warmup 1
bash profiles/run_pmc.sh g32_ta "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum FETCH_SIZE" --steps 3 --warmup 1
for t in g32_tcc g32_sq g32_ta; do echo "== $t"; python3 profiles/summarize_sq.py gpurun_out/pmc_$t gemm32b; python3 profiles/summarize_sq.py gpurun_out/pmc_$t knn_pair; done > gpurun_out/r6_gemm32b_counters.txt 2>&1
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_knnw_$c -o pmc -- python3 profiles/tools/knn_write_ab.py 5 > gpurun_out/pmc_knnw_$c.log 2>&1
  echo "== $c"; python3 profiles/summarize_sq.py gpurun_out/pmc_knnw_$c knn_pair
done > gpurun_out/r6_knn_write_ab.txt 2>&1
python3 bench.py --gpus 8 --share-gpu --dist-backend gloo --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/r6_8ranks.err | tail -1 > gpurun_out/r6_bench_line_8ranks_one_gpu_gloo.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r6_driver.err | tail -1 > gpurun_out/r6_bench_line_driver_form_a.json
cat gpurun_out/r6_gemm32b_counters.txt gpurun_out/r6_knn_write_ab.txt
python3 - <<PY
import json
for f in ("r6_bench_line_8ranks_one_gpu_gloo", "r6_bench_line_driver_form_a"):
    try:
        j = json.load(open("gpurun_out/%s.json" % f))
        print(f, j["ms_per_step"], j.get("ranks_seen"), j.get("rank0_pinned_cpus"), json.dumps(j.get("summary"))[:1500])
        if "include_pcie" in json.dumps(j):
            sr = j.get("sub", j.get("sub_results", {}))
            print(json.dumps({k: v for k, v in sr.get("include_pcie", {}).items() if k != "what"}))
        cb = j.get("cpu_baseline")
        if cb: print(json.dumps({k: v for k, v in cb.items() if k not in ("sample",)})[:1200])
    except Exception as e:
        print(f, "unreadable", e)
PY
