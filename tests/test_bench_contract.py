"""CPU tier: the bench line committed with this round's profiles carries every field the driver's contract names, and its roofline
arithmetic is self-consistent (the figures are re-derived from the line's own counts).  Guards bench.py's JSON against drifting away
from the contract between GPU runs — the line itself is produced on the GPU box (profiles/rNN_bench_full_unprofiled.json.log)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_full_unprofiled.json.log")))
    assert files, "no committed bench line"
    return [json.loads(ln) for ln in open(files[-1]) if ln.startswith("{")][-1], files[-1]


def test_committed_bench_line_follows_the_contract():
    j, path = _latest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, (k, path)
    assert j["unit"] == "Mpaths/s" and j["higher_is_better"] is True and j["vs_baseline"] is None and j["dtype"] == "f32" and j["n_gpus"] == 1
    assert "workload" in j["config"] and "model" not in j["config"] and "rtcamp6_v3_1 1920x1080" in j["config"]["workload"]
    # value = paths / time: steps x samplings per step x W x H x 4 over steps x ms_per_step
    assert abs(j["config"]["paths_total"] / (j["steps"] * j["ms_per_step"] * 1e-3) / 1e6 - j["value"]) <= 2e-3 * j["value"]
    assert j["config"]["paths_total"] == 1920 * 1080 * 4 * j["config"]["samplings_total"]
    r = j["roofline"]
    for k in ("bound", "bound_contract", "achieved", "peak", "unit", "frac", "frac_survey_8d", "traffic", "kernel", "pair_bound", "algorithmic_bytes_per_path", "avg_launch_ms"):
        assert k in r, k
    assert r["bound_contract"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["kernel"] == "trace_kernel" and r["frac"] == r["frac_survey_8d"]
    # 32 B per node test + 36 B per triangle test (+ 16 / 24 per sphere / cuboid test: ~24 B per path in this scene)
    per_path = r["rays_per_path"] * (32 * r["node_tests_per_ray"] + 36 * r["tri_tests_per_ray"])
    assert per_path <= r["algorithmic_bytes_per_path"] <= per_path + 60
    achieved = r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9
    assert abs(achieved - r["achieved"]) <= 2e-3 * achieved and abs(r["achieved"] / r["peak"] - r["frac"]) <= 1e-3
    assert 0 < r["loaded_bytes"]["frac"] <= r["frac"] and isinstance(r["traffic"], int) and r["traffic"] > 0 and r["traffic_stale"] is False
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mpaths/s" and "sample" in c
    m = j["multi_gpu"]
    assert m["accumulator_bytes"] == 1920 * 1080 * 3 * 4 and len(m["per_rank"]) == 1 and m["per_rank"][0]["paths"] == j["config"]["paths_total"]


def test_committed_rocprof_summary_agrees_with_the_line():
    """profiles/rNN_bench_kernel_stats.md (rocprofv3 --kernel-trace --stats of the same command) and the line's HIP-event duration of the
    dominant kernel agree (the contract asks for that): within 3 %."""
    j, path = _latest_line()
    md = open(path.replace("_bench_full_unprofiled.json.log", "_bench_kernel_stats.md")).read()
    row = [ln for ln in md.splitlines() if "trace_kernel<false, 5, true" in ln][0].split("|")
    avg_ms = float(row[4])
    assert abs(avg_ms - j["roofline"]["avg_launch_ms"]) <= 0.03 * avg_ms, (avg_ms, j["roofline"]["avg_launch_ms"])
    seed = [ln for ln in md.splitlines() if "seed_seg_kernel" in ln][0].split("|")
    assert abs(float(seed[4]) - j["roofline"]["seed_kernel_avg_ms"]) <= 0.03 * float(seed[4])


def _latest_scene_line(scene):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_%s.json.log" % scene)))
    assert files, "no committed bench line for %s" % scene
    return [json.loads(ln) for ln in open(files[-1]) if ln.startswith("{")][-1], files[-1]


def test_committed_lines_of_the_trace_bound_scenes():
    """The scenes whose trace kernel is the slower kernel of the pair (or level with it) — rtcamp6_v2, rtcamp6_v1, tbf3 — are tracked like the
    headline: a bench line with its roofline (re-derivable from its own counts), PMC traffic taken on these kernel sources for THAT scene, and a
    rocprofv3 kernel-trace summary of the same command that agrees with the line's HIP-event durations.  These are the only lines on which
    trace-kernel work shows (every BASELINE configuration runs at the seed kernel's pace)."""
    for scene in ("rtcamp6_v2", "rtcamp6_v1", "tbf3"):
        j, path = _latest_scene_line(scene)
        assert j["unit"] == "Mpaths/s" and j["n_gpus"] == 1 and j["value"] > 0 and scene + " 1920x1080" in j["config"]["workload"]
        r = j["roofline"]
        assert r["kernel"] == "trace_kernel" and r["pair_bound"] in ("trace_kernel", "seed_seg_kernel") and r["bound_contract"] == "hbm"
        slower = "trace_kernel" if r["avg_launch_ms"] > r["seed_kernel_avg_ms"] else "seed_seg_kernel"
        assert r["pair_bound"] == slower, (scene, r["avg_launch_ms"], r["seed_kernel_avg_ms"])
        per_path = r["rays_per_path"] * (32 * r["node_tests_per_ray"] + 36 * r["tri_tests_per_ray"])
        assert per_path <= r["algorithmic_bytes_per_path"] <= per_path + 150, (scene, per_path, r["algorithmic_bytes_per_path"])   # + 16 / 24 B per sphere / cuboid test
        achieved = r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9
        assert abs(achieved - r["achieved"]) <= 2e-3 * achieved and abs(r["achieved"] / r["peak"] - r["frac"]) <= 1e-3
        assert abs(r["reference_rays_per_path"] - r["rays_per_path"] - r["nee_shadow_rays_culled_per_path"]) <= 2e-3 and r["nee_shadow_rays_culled_per_path"] > 0
        assert isinstance(r["traffic"], int) and r["traffic"] > 0 and r["traffic_stale"] is False and scene in r["traffic_source"]
        assert 30 <= r["lanes_per_box_pass"] <= 64 and j["multi_gpu"]["exchange_verified"] is True
        # the physical roofline: L1 lane accesses (PMC) against 1.41 per clock per CU (tools/taprobe.hip) — a fraction, also with the chip to itself
        lk = r["physical"]["l1_lookup"]
        assert abs(lk["peak_per_s"] - 256 * 2.4e9 * 1.41) < 1e6 and abs(lk["lane_accesses_per_launch"] / (r["avg_launch_ms"] * 1e-3) / lk["peak_per_s"] - lk["frac"]) <= 1e-3
        assert 0.2 < lk["frac"] < lk["frac_alone"] < 1.0, (scene, lk)
        # the same roofline from the run's OWN counts (one load per node and sphere test, two per cuboid test, three per triangle test): what the
        # PMC pass counted is that figure to within 15 % (neighbouring lanes that sit on the same node merge; shading and the hand-off records add a little)
        live = r["l1_lookup"]
        lanes = r["rays_per_path"] * (r["node_tests_per_ray"] + 3 * r["tri_tests_per_ray"])
        assert lanes <= live["lane_loads_per_path"] <= lanes + 0.01 * lanes + 20, (scene, lanes, live)
        assert abs(live["lane_loads_per_path"] * r["algorithmic_bytes_per_launch"] / r["algorithmic_bytes_per_path"] / (r["avg_launch_ms"] * 1e-3) / 1e9 - live["achieved"]) <= 2e-3 * live["achieved"]
        assert abs(live["achieved"] / live["peak"] - live["frac"]) <= 1e-3 and 0.2 < live["frac"] < live["frac_alone"] < 1.0
        assert 0.85 * live["lane_loads_per_path"] <= r["physical"]["l1_line_accesses_per_path"] <= 1.15 * live["lane_loads_per_path"], (scene, live, r["physical"]["l1_line_accesses_per_path"])
        md = open(path.replace("_bench_%s.json.log" % scene, "_bench_%s_kernel_stats.md" % scene)).read()
        row = [ln for ln in md.splitlines() if "trace_kernel<false, 5, true" in ln][0].split("|")
        assert abs(float(row[4]) - r["avg_launch_ms"]) <= 0.03 * float(row[4]), (scene, row[4], r["avg_launch_ms"])
        seed = [ln for ln in md.splitlines() if "seed_seg_kernel" in ln][0].split("|")
        assert abs(float(seed[4]) - r["seed_kernel_avg_ms"]) <= 0.03 * float(seed[4])


def test_committed_headline_line_proves_its_exchange_and_its_shortcuts():
    """Round 5 fields of the headline line: the multi-GPU evidence (one rank: no communicator; checksum of the parts against the total), the
    NEE shadow rays that were not traced, the wave budget the governor settled on, and which figures are replayed from profiles/."""
    j, _ = _latest_line()
    m = j["multi_gpu"]
    assert m["rccl"]["nranks"] == 1 and m["rccl"]["path"] == "none" and m["exchange_verified"] is True and m["checksum"]["rel_err"] <= 1e-6
    r = j["roofline"]
    assert 0.3 < r["nee_shadow_rays_culled_per_path"] < 0.9 and 2.9 < r["reference_rays_per_path"] < 3.2      # 3.05 scene.intersect calls per path
    g = r["priority_governor"]
    assert g["level"] == 0 and (g["trace_workgroups"] == "all" or 512 <= g["trace_workgroups"] <= 1024)
    assert r["replayed_from_profiles"]["fields"] == ["traffic", "traffic_write", "physical", "issue"] and "schema" in r
