#!/bin/bash
# How many lanes of a trace wave hold a path when the traversal starts (NOTES.md J: "the lanes that sit out a box pass hold parked or finished
# paths, not nothing").  Needs a PROBE build of the library, not kept in the tree: in trace_kernel.h's counters instantiation replace the leaf
# counters (`if (n) { ws.ph[4]++; ws.ph[5] += n; }` -> nothing) by `if (CNT) { ws.ph[4]++; ws.ph[5] += n_active; }` in front of "---- C: traversal",
# `tools/ab/mkvariant.sh idleprobe`, restore the source.  The bench line's lanes_per_leaf_call is then that average.
# Round 5 (profiles/r05_lanes_holding_a_path.txt): 59.8 - 62.4 of 64.
cp hanamaru-renderer_amd/libhanamaru_hip.so /tmp/lib_cur.so
cp tools/ab/lib_idleprobe.so hanamaru-renderer_amd/libhanamaru_hip.so
for sc in rtcamp6_v3_1 rtcamp6_v2 rtcamp6_v1 tbf3; do
python bench.py --scene $sc --steps 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$sc: lanes that hold a path when the traversal starts (probe build: leaf counters reused): %.1f of 64; lanes per box pass %.1f, per shade call %.1f' % (r['lanes_per_leaf_call'], r['lanes_per_box_pass'], r['lanes_per_shade_call']))"
done
cp /tmp/lib_cur.so hanamaru-renderer_amd/libhanamaru_hip.so
