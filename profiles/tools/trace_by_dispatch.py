"""Per-dispatch kernel durations from a rocprofv3 --kernel-trace CSV, grouped into consecutive runs of the same kernel name + grid (one run = one
shape of exp_gemm32b.py).  usage: python3 profiles/tools/trace_by_dispatch.py <dir> [name filter]"""
import csv, glob, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "gemm32"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
runs = []
for r in rows:
    if flt not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"))
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if runs and runs[-1][0] == key:
        runs[-1][1].append(dur)
    else:
        runs.append((key, [dur]))
for key, ds in runs:
    ds2 = sorted(ds)
    print("%-42s grid %-8s lds %-6s vgpr %-4s n %2d  min %6.1f  med %6.1f us" % (key[0], key[1], key[2], key[3], len(ds), ds2[0], ds2[len(ds2) // 2]))
