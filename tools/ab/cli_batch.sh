# GPU box: throughput of the hanamaru-hip CLI by report / launch granularity (profiles/rNN_cli_report_granularity.txt)
cd /tmp && mkdir -p clib && cd clib
for args in "" "--launch 1" "--launch 1 --inflight 16" "--launch 2" "--launch 8" "--batch 4" "--batch 32" "--launch 1 --inflight 1" "--gpu-ids 0,0" "--precise"; do
  $GRAFT_REPO_ROOT/hanamaru-renderer_amd/hanamaru-hip -w 1920 -h 1080 -s 1024 -t 100000 -i 100000 $args --assets $GRAFT_REPO_ROOT/assets 2>&1 | grep "gpu:\|^launches" | tr '\n' ' ' | sed "s/^/[default${args:+: }$args] /"; echo
done
