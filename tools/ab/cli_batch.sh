cd /tmp && mkdir -p clib && cd clib
for args in "--batch 1" "--batch 1 --inflight 16" "--batch 4" "--batch 32" "--batch 1 --inflight 1" "--batch 1 --gpu-ids 0,0" "--batch 4 --gpu-ids 0,0"; do
  $GRAFT_REPO_ROOT/hanamaru-renderer_amd/hanamaru-hip -w 1920 -h 1080 -s 768 -t 100000 -i 100000 $args --assets $GRAFT_REPO_ROOT/assets 2>&1 | grep "gpu:" | sed "s/^/$args : /"
done
