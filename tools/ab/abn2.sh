#!/bin/bash
# like abn.sh, plus the governor's level / moves and the seed kernel alone
cp hanamaru-renderer_amd/libhanamaru_hip.so /tmp/lib_cur.so
for rep in 1 2; do for L in "$@"; do
  cp $L hanamaru-renderer_amd/libhanamaru_hip.so
  python bench.py --steps 16 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; g = r['priority_governor']
print('%-24s -> %7.1f Mpaths/s   trace %.2f ms (alone %.2f)   seed %.2f ms   gov level %d judged %d moves %d  wgs %s (%d moves)  rays/path %.3f' % ('$L', d['value'], r['avg_launch_ms'], r.get('avg_launch_ms_alone', 0), r['seed_kernel_avg_ms'], g['level'], g['launches_judged'], g['moves'], g['trace_workgroups'], g['trace_workgroup_moves'], r.get('rays_per_path', 0)))"
done; done
cp /tmp/lib_cur.so hanamaru-renderer_amd/libhanamaru_hip.so
