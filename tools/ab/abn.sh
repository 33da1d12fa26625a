#!/bin/bash
# tools/ab/abn.sh libA libB ... : two rounds over all the libraries on one box
cp hanamaru-renderer_amd/libhanamaru_hip.so /tmp/lib_cur.so
for rep in 1 2; do for L in "$@"; do
  cp $L hanamaru-renderer_amd/libhanamaru_hip.so
  python bench.py --steps 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-40s ->  %.1f Mpaths/s   trace %.2f ms   seed %.2f ms' % ('$L', d['value'], r['avg_launch_ms'], r['seed_kernel_avg_ms']))"
done; done
cp /tmp/lib_cur.so hanamaru-renderer_amd/libhanamaru_hip.so
