cd $GRAFT_REPO_ROOT
cp point-unet_amd/libpointseg_hip.so /tmp/orig.so
for v in abl1 abl2; do
  cp point-unet_amd/libpointseg_$v.so point-unet_amd/libpointseg_hip.so
  echo "== $v"
  TOPN=60 bash profiles/run_kernel_stats.sh $v --steps 20 --warmup 3 --no-sub-results --no-pipeline 2>&1 | grep -i "att32b"
done
cp /tmp/orig.so point-unet_amd/libpointseg_hip.so
