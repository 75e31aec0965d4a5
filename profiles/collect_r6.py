#!/usr/bin/env python3
"""Copies the summaries profiles/regen_r6.sh left under gpurun_out/ into profiles/ (tracked)."""
import glob
import json
import os
import shutil

here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(os.path.dirname(here), "gpurun_out")
pairs = {"prof_pipe/pipe_kernel_stats.csv": "r6_bench_kernel_stats.csv", "prof_serial/serial_kernel_stats.csv": "r6_serial_kernel_stats.csv",
         "prof_train_b8/train_b8_kernel_stats.csv": "r6_train_b8_kernel_stats.csv",
         "prof_train_b8_bf16/train_b8_bf16_kernel_stats.csv": "r6_train_b8_bf16_kernel_stats.csv",
         "prof_train_b1/train_b1_kernel_stats.csv": "r6_train_b1_kernel_stats.csv",
         "r6_pmc_per_kernel.json": "r6_pmc_per_kernel.json", "r6_pmc_traffic.json": "r6_pmc_traffic.json", "r6_sq_counters.txt": "r6_sq_counters.txt",
         "r6_pmc_traffic_train.json": "r6_pmc_traffic_train.json", "r6_pmc_train_per_kernel_b8_f32.json": "r6_pmc_train_per_kernel_b8_f32.json",
         "r6_pmc_train_per_kernel_b8_bf16.json": "r6_pmc_train_per_kernel_b8_bf16.json",
         "r6_pmc_train_per_kernel_b1_f32.json": "r6_pmc_train_per_kernel_b1_f32.json", "pmc_train_summary.txt": "r6_pmc_train_summary.txt"}
for n in ("", "_driver_form", "_serial", "_config5", "_batch2", "_train_b8", "_train_b8_bf16", "_train_b8_atomic_scatter", "_train_b1", "_2ranks_one_gpu_gloo", "_8ranks_one_gpu_gloo"):
    pairs["r6_bench_line%s.json" % n] = "r6_bench_line%s.json" % n
for s, d in pairs.items():
    cand = glob.glob(os.path.join(src, "**", os.path.basename(s)), recursive=True) if not os.path.exists(os.path.join(src, s)) else [os.path.join(src, s)]
    if not cand:
        print("missing", s)
        continue
    shutil.copyfile(cand[0], os.path.join(here, d))
    if d.startswith("r6_bench_line"):
        try:
            j = json.load(open(os.path.join(here, d)))
            print("%-44s %8.3f ms/step  %10.1f M %s" % (d, j["ms_per_step"], j["value"] / 1e6, j["unit"]))
        except Exception as e:  # an empty line = that bench failed on the box
            print("%-44s unreadable: %s" % (d, e))
