#!/bin/bash
# usage: tools/bench_sweep.sh "<common bench flags>" "<flag name>" v1 v2 ...   -> one short line per value
common="$1"; flag="$2"; shift 2
for v in "$@"; do
  python bench.py --steps 16 --no-cpu-baseline $common $flag $v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$flag $v  ->  %.1f Mpaths/s   trace %.2f ms   seed %.2f ms' % (d['value'], r['avg_launch_ms'], r['seed_kernel_avg_ms']))"
done
