#!/usr/bin/env python3
"""Copies the summaries profiles/regen_r1.sh left under gpurun_out/ into profiles/ (tracked)."""
import glob
import json
import os
import shutil

here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(os.path.dirname(here), "gpurun_out")
pairs = {"prof_pipe/pipe_kernel_stats.csv": "r1_bench_kernel_stats.csv", "prof_serial/serial_kernel_stats.csv": "r1_serial_kernel_stats.csv",
         "r1_pmc_per_kernel.json": "r1_pmc_per_kernel.json", "r1_pmc_traffic.json": "r1_pmc_traffic.json", "r1_sq_counters.txt": "r1_sq_counters.txt",
         "r1_bench_line.json": "r1_bench_line.json", "r1_bench_line_serial.json": "r1_bench_line_serial.json",
         "r1_bench_line_config5.json": "r1_bench_line_config5.json", "r1_bench_line_train_b8.json": "r1_bench_line_train_b8.json",
         "r1_bench_line_train_b1.json": "r1_bench_line_train_b1.json", "r1_bench_line_train_b8_bf16.json": "r1_bench_line_train_b8_bf16.json"}
for s, d in pairs.items():
    cand = glob.glob(os.path.join(src, "**", os.path.basename(s)), recursive=True) if not os.path.exists(os.path.join(src, s)) else [os.path.join(src, s)]
    if not cand:
        print("missing", s)
        continue
    shutil.copyfile(cand[0], os.path.join(here, d))
    if d.startswith("r1_bench_line"):
        j = json.load(open(os.path.join(here, d)))
        print("%-32s %8.3f ms/step  %10.1f M %s" % (d, j["ms_per_step"], j["value"] / 1e6, j["unit"]))
