cd "$GRAFT_REPO_ROOT"
for l in 3 4 5 6; do for rep in 1 2; do
GPU_MAX_HW_QUEUES=$((l+2)) python3 bench.py --gpus 1 --lanes $l --steps 20 --warmup 5 --no-cpu-baseline --no-sub-results 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('lanes $l (queues $((l+2))): coalesced %.4f  one-per-launch %.4f ms/step' % (j['ms_per_step'], j['one_cloud_per_launch']['ms_per_step']))"
done; done
