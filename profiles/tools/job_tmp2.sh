cd /root/repo
PS_BN_SLICE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_b1s -o b1 -- python3 bench.py --mode train --batch 1 --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timing > gpurun_out/prof_b1s.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_b1s/b1_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'knn_pair' in r['Kernel_Name']]
step=rows[idx[-1]:]
prev_end=None
for r in step:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    gap=(int(r['Start_Timestamp'])-prev_end)/1e3 if prev_end else 0
    prev_end=int(r['End_Timestamp'])
    if 'bn_slice' in r['Kernel_Name'] or 'wgrad_reduce' in r['Kernel_Name'] or 'pack_batch' in r['Kernel_Name']:
        print("%-50s g %8s wg %5s dur %7.1f gap_before %6.1f"%(r['Kernel_Name'][:50], r['Grid_Size_X'], r['Workgroup_Size_X'], d, gap))
PY
