export PS_EXP_G32B_TWICE=1
TOPN=1 bash profiles/run_kernel_stats.sh g32b2 --no-pipeline --steps 10 --warmup 3 --no-sub-results > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_g32b2/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r for r in rows if 'gemm32b' in r['Kernel_Name']]
last=names[-14:]
for a,b in zip(last[0::2],last[1::2]):
    print(a['Kernel_Name'][40:70], a['Grid_Size'], 'first', int(a['End_Timestamp'])-int(a['Start_Timestamp']), 'second', int(b['End_Timestamp'])-int(b['Start_Timestamp']), 'gap', int(b['Start_Timestamp'])-int(a['End_Timestamp']))
PY
