cd ${GRAFT_REPO_ROOT:-/root/repo}
cp cvpr2020_manet_amd/libmanet_hip.so /tmp/ab_new.so; cp cvpr2020_manet_amd/libmanet_hip_old.so /tmp/ab_old.so
for v in new old; do
  cp /tmp/ab_$v.so cvpr2020_manet_amd/libmanet_hip.so
  echo "== $v"; bash tools/e2e_launch_count.sh ab_$v 2>&1 | grep -E "conv1x1|dwconv|global_match|local_fused|head_layer1|total|frames/s"
done
cp /tmp/ab_new.so cvpr2020_manet_amd/libmanet_hip.so
