#!/usr/bin/env python3
"""tools/collect_profiles.py rNN: copy the summaries of gpurun_out/final_rNN (tools/final_profiles.sh) into profiles/ and print the
table of DESIGN.md §5.1."""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "final_" + tag), os.path.join(root, "profiles")
for f in sorted(glob.glob(os.path.join(src, tag + "_*"))):
    shutil.copy(f, dst)
shutil.copy(os.path.join(src, "pytest_gpu.log"), os.path.join(dst, tag + "_pytest_gpu_tail.txt"))


def line(name):
    return json.loads(open(os.path.join(src, "%s_%s.json.log" % (tag, name))).read().strip().splitlines()[-1])


for name in ("bench_full_unprofiled", "bench_full", "bench_c2_spheres", "bench_c5_4k_dodeca", "bench_lbvh", "bench_ploc", "bench_gpus2_one_device", "bench_russian_roulette_nonparity"):
    d = line(name)
    r = d["roofline"]
    print("%-34s %8.1f Mpaths/s  step %.2f ms  trace %.2f (alone %s) seed %.2f  frac %.3f (hbm-normalised alone %s)  l2 %.3f  8d %.3f/%s  nodes/ray %.2f tris/ray %.2f  lanes box %.1f  build %.2f ms" % (
        name, d["value"], d["ms_per_step"], r["avg_launch_ms"], r.get("avg_launch_ms_alone"), r["seed_kernel_avg_ms"], r["frac"] or 0, r.get("hbm_normalised_alone"),
        (r.get("l2") or {}).get("frac", 0), (r.get("survey_8d") or {}).get("frac", 0), (r.get("survey_8d") or {}).get("normalised_alone"), r.get("node_tests_per_ray", 0),
        r.get("tri_tests_per_ray", 0), r.get("lanes_per_box_pass", 0), r["bvh_build_ms"]))
d = line("bench_full_unprofiled")
r = d["roofline"]
for k in ("algorithmic_bytes_per_path", "traversal_section", "traversal_only", "physical", "traffic", "traffic_write", "traffic_stale", "phase_share_of_wave_cycles", "lanes_per_shade_call", "lanes_per_leaf_call", "rays_per_path"):
    print(k, json.dumps(r.get(k)))
print("issue", json.dumps(r.get("issue"))[:600])
print("seed_kernel", d.get("seed_kernel"))
print("cpu_baseline", {k: v for k, v in d.get("cpu_baseline", {}).items() if k != "sample"})
print(open(os.path.join(src, tag + "_bench_kernel_stats.md")).read())
