#!/bin/bash
# tools/ab/cull_ab.sh [scenes...] : nee_setup's shortcuts on / off on one box (bit-identical images; the rate and the kernels' times differ)
for sc in "${@:-rtcamp6_v3_1 rtcamp6_v2 rtcamp6_v1 tbf3}"; do for cull in 7 0 7 0; do
  python bench.py --scene $sc --steps 12 --no-cpu-baseline --debug nee_cull=$cull 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-14s nee_cull %s -> %7.1f Mpaths/s   trace %.2f ms (alone %.2f)   seed %.2f ms   rays/path %.3f  lanes/box %.1f  nodes/ray %.2f' % ('$sc', '$cull', d['value'], r['avg_launch_ms'], r.get('avg_launch_ms_alone', 0), r['seed_kernel_avg_ms'], r.get('rays_per_path', 0), r.get('lanes_per_box_pass', 0), r.get('node_tests_per_ray', 0)))"
done; done
