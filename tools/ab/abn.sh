#!/bin/bash
# tools/ab/abn.sh libA libB ... : two rounds over all the libraries on one box (boxes differ by ~1 %, so builds are only compared
# within a call); BENCH_ARGS adds bench.py flags
cp hanamaru-renderer_amd/libhanamaru_hip.so /tmp/lib_cur.so
for rep in 1 2; do for L in "$@"; do
  cp $L hanamaru-renderer_amd/libhanamaru_hip.so
  python bench.py --steps 16 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-28s -> %7.1f Mpaths/s   trace %.2f ms (alone %.2f)   seed %.2f ms   lanes/box %.1f  nodes/ray %.2f' % ('$L', d['value'], r['avg_launch_ms'], r.get('avg_launch_ms_alone', 0), r['seed_kernel_avg_ms'], r.get('lanes_per_box_pass', 0), r.get('node_tests_per_ray', 0)))"
done; done
cp /tmp/lib_cur.so hanamaru-renderer_amd/libhanamaru_hip.so
