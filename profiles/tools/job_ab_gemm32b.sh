#!/bin/bash
# How much of the serial time of the deep dense layers shows in the PIPELINED headline?  A/B: every dense layer on the fp32 MFMA (gemm32.hip)
# against the split-bf16 form (gemm32b.hip) -- serial one-lane and 4-lane figures of both.  (tuning flavour: the default build ignores PS_*)
cd "$GRAFT_REPO_ROOT"
for v in 3e8 1e30; do
  PS_GEMM32B_MIN_FLOPS=$v python3 profiles/tools/with_variant.py tuning bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-sub-results 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=l.get('sub_results',{}).get('serial',{})
print('PS_GEMM32B_MIN_FLOPS=$v  pipelined %.4f ms/step  serial %.4f  one-lane %.4f' % (l['ms_per_step'], s.get('ms_per_cloud',0), s.get('ms_per_cloud_one_lane',0)))
st = {r['stage']: r['ms'] for r in l.get('stages', [])} if isinstance(l.get('stages'), list) else {}
print({k: round(v,4) for k,v in st.items() if 'dense' in k or 'dec' in k})
"
done
