#!/usr/bin/env python3
"""bench.py — Mpaths/s of the HIP path-tracing hot path on BASELINE.json's headline configuration.

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): the reference's live scene `rtcamp6_v3_1` at 1920x1080.  One STEP = one
hr_render() call of `--spp-per-step` samplings (default 16, launched 4 at a time) of the whole image on every GPU
= 16 x 1920 x 1080 x 4 camera paths per GPU.  The default K = 64 steps is exactly BASELINE's 1920x1080 x 1024 samplings on one GPU.
With N GPUs the sampling indices are sharded round-robin ((s-1) mod N == rank, one process per GPU), every
GPU still renders `spp-per-step` samplings per step (weak scaling), and the fp32 radiance accumulators are
summed with ONE all-reduce (RCCL) inside the timed region.  Inputs (scene, textures) are resident in HBM
before the timed region; nothing is skipped inside it (seed kernel + trace kernel + accumulation).

Extra objects on the JSON line:
  roofline     — trace kernel: algorithmic bytes per launch (32 B/node test, 36 B/triangle test,
                 16 B/sphere, 24 B/cuboid, counted by an instrumented run of the same kernel on the same seeds)
                 divided by the kernel's mean launch duration (HIP events on its stream), vs 8 TB/s HBM peak.
  cpu_baseline — the CPU oracle (f64 restatement of the reference path, oracle/) timed on this box's host cores
                 on a bounded sample; reported, not optimised.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBS = 34500.0  # same guide: aggregate L2 bandwidth


def usable_cpus():
    """Host threads this process can really run at once: hardware threads, clipped by the scheduler affinity and by the cgroup CPU
    quota (a container on a 256-thread host with cpu.max = 16 CPUs runs 256 threads no faster than 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, (os.cpu_count() or 1), quota


def latest_pmc_traffic():
    """profiles/rNN_pmc_traffic.json of the newest round (written by tools/prof_pmc.sh), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    for f in reversed(files):
        try:
            j = json.load(open(f))
            if "kernels" in j:
                return j
        except (OSError, ValueError):
            pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-step", type=int, default=16)
    ap.add_argument("--batch", type=int, default=0, help="samplings per kernel launch (0 = the library's automatic choice: 4 at 1080p)")
    ap.add_argument("--scene", default="rtcamp6_v3_1")
    ap.add_argument("--adv-den", type=int, default=0, help="trace kernel early-exit denominator (0 = library default)")
    ap.add_argument("--leaf-den", type=int, default=0)
    ap.add_argument("--min-waves", type=int, default=0)
    ap.add_argument("--max-leaf", type=int, default=0)
    ap.add_argument("--bvh-builder", type=int, default=0, help="0 = host SAH (default), 1 = device LBVH, 2 = device PLOC")
    ap.add_argument("--split-ratio", type=float, default=None, help="early split clipping: -1 automatic (library default), 0 off, > 0 ratio")
    ap.add_argument("--seed-mode", type=int, default=-1)
    ap.add_argument("--seed-prio", type=int, default=-1)
    ap.add_argument("--trace-boost", type=int, default=-2, help="-1 = governed (library default), 0 / 1 / 2 = fixed level")
    ap.add_argument("--init-prio", type=int, default=-1)
    ap.add_argument("--seed-split", type=int, default=-1)
    ap.add_argument("--max-tail-gib", type=float, default=0.0)
    ap.add_argument("--trace-wgs", type=int, default=0, help="trace-kernel workgroups per CU (0 = library default)")
    ap.add_argument("--quant-nodes", type=int, default=-1)
    ap.add_argument("--kchunk", type=int, default=0)
    ap.add_argument("--node-unroll", type=int, default=0)
    ap.add_argument("--debug-skip", type=int, default=0, help="timing experiments: skip seeding kernels after the warm-up (image is garbage)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counters", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import hanamaru_amd as ha
    from hanamaru_amd.sharding import step_range

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    # HR_BENCH_ONE_DEVICE=1 is a debugging aid for boxes with a single GPU: every rank uses cuda:0 and the
    # collective runs over gloo on a host copy (RCCL refuses two ranks on one device).  Never set by the driver.
    one_device = os.environ.get("HR_BENCH_ONE_DEVICE") == "1"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if one_device else "nccl", rank=rank, world_size=world)
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def all_reduce_(t, op):
        if one_device:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    # The accumulators are summed by the LIBRARY (hr_allreduce_accumulator: ncclAllReduce on the device accumulator, RCCL loaded
    # by libhanamaru_hip.so); torch.distributed only carries the ncclUniqueId to the ranks, the barriers and the max over ranks.
    lib_rccl = dist is not None and not one_device

    W, H, SPS = args.width, args.height, args.spp_per_step
    scene = ha.Scene(args.scene)
    r = ha.Renderer(local_rank)
    if args.max_leaf:
        r.set_option("max_leaf", args.max_leaf)
    if args.bvh_builder:
        r.set_option("bvh_builder", args.bvh_builder)
    if args.quant_nodes >= 0:
        r.set_option("quant_nodes", args.quant_nodes)
    if args.split_ratio is not None:
        r.set_option("split_ratio", args.split_ratio)
    r.upload_scene(scene)
    r.set_resolution(W, H)
    r.set_option("batch", args.batch)
    if args.adv_den:
        r.set_option("adv_den", args.adv_den)
    if args.leaf_den:
        r.set_option("leaf_den", args.leaf_den)
    if args.min_waves:
        r.set_option("min_waves", args.min_waves)
    if args.trace_wgs:
        r.set_option("trace_wgs", args.trace_wgs)
    if args.kchunk:
        r.set_option("kchunk", args.kchunk)
    if args.node_unroll:
        r.set_option("node_unroll", args.node_unroll)
    if args.seed_mode >= 0:
        r.set_option("seed_mode", args.seed_mode)
    if args.trace_boost >= -1:
        r.set_option("trace_boost", args.trace_boost)
    if args.seed_prio >= 0:
        r.set_option("seed_prio", args.seed_prio)
    if args.max_tail_gib:
        r.set_option("max_tail_gib", args.max_tail_gib)
    if args.seed_split >= 0:
        r.set_option("seed_split", args.seed_split)
    if args.init_prio >= 0:
        r.set_option("init_prio", args.init_prio)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    r.bind_accumulator(acc.data_ptr())
    if lib_rccl:
        uid = [ha.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        r.comm_init_rank(uid[0], world, rank)
        r.allreduce_accumulator()      # warm-up: the first collective sets up the rings
        r.synchronize()
    paths_per_step_gpu = W * H * 4 * SPS

    def run_step(i):
        # step i covers samplings [i*SPS*world + 1, (i+1)*SPS*world]; this rank takes (s-1) % world == rank
        r.render(*step_range(i, SPS, world, rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- algorithmic bytes per path: instrumented run of the same kernel (outside the timed region)
    bytes_per_path = None
    counters = None
    if rank == 0 and not args.no_counters:
        r.set_option("counters", 1)
        r.clear()
        r.render(1, 3)
        r.synchronize()
        counters = r.stats()
        r.set_option("counters", 0)
        alg = 32 * counters["node_tests"] + 36 * counters["tri_tests"] + 16 * counters["sphere_tests"] + 24 * counters["cuboid_tests"]
        bytes_per_path = alg / max(1, counters["paths"])
    r.clear()

    for i in range(args.warmup):
        run_step(i)
    r.synchronize()
    r.clear()
    if args.debug_skip:
        r.set_option("debug_skip", args.debug_skip)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(i)
    if lib_rccl:
        r.allreduce_accumulator()      # enqueued on the render stream behind the last step
    r.synchronize()
    if dist is not None and not lib_rccl:
        all_reduce_(acc, dist.ReduceOp.SUM)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce_(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = r.stats()
    acc_mean = float(r.read_accumulator().mean()) if lib_rccl else float(acc.mean().item())   # after the all-reduce: the total
    # ---- the trace kernel with the chip to itself (seed kernel skipped: the hand-off buffers still hold the previous
    #      batches' draws, so the workload is the same) — outside the timed region, reported next to the concurrent figure
    alone_ms = None
    if rank == 0 and not args.debug_skip and not args.no_counters:
        if os.environ.get("HR_BENCH_CHECKSUM") == "1":
            sys.stderr.write("accumulator mean after all-reduce: %.9g\n" % acc_mean)
        r.set_option("debug_skip", 2)
        r.render(*step_range(0, SPS, world, rank))
        r.synchronize()
        st2 = r.stats()
        r.set_option("debug_skip", 0)
        alone_ms = (st2["trace_kernel_ms"] - st["trace_kernel_ms"]) / max(1, st2["trace_launches"] - st["trace_launches"])

    if os.environ.get("HR_BENCH_CHECKSUM") == "1" and rank == 0 and alone_ms is None:
        sys.stderr.write("accumulator mean after all-reduce: %.9g\n" % acc_mean)
    if rank == 0:
        total_paths = paths_per_step_gpu * world * args.steps
        value = total_paths / elapsed / 1e6
        out = {
            "metric": "Mpaths/sec at 1920x1080x1024spp (rtcamp6 scene); achieved GB/s in BVH traversal",
            "value": round(value, 3), "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "reference scene assets shipped in-repo (assets/), per-path ISAAC-64 seeds as in renderer.rs:165-168",
            "config": {"workload": "%s %dx%d, %d samplings (x4 sub-samples) per step per GPU; K=64 steps = 1024 samplings" % (args.scene, W, H, SPS),
                       "samplings_per_step_per_gpu": SPS, "samplings_per_launch_requested": args.batch, "paths_per_step": paths_per_step_gpu * world,
                       "parallelism": "spp-sharded x%d, one all-reduce (%s)" % (world, "ncclAllReduce inside libhanamaru_hip.so" if lib_rccl else
                                                                                  ("gloo on a host copy: HR_BENCH_ONE_DEVICE debugging aid" if dist is not None else "single rank"))},
            "rays_per_s_M": None,
        }
        launches = max(1, st["trace_launches"])
        avg_ms = st["trace_kernel_ms"] / launches
        # the library may cap the samplings per launch (hand-off buffer size): use what was actually launched
        paths_per_launch = st["paths"] / launches if st["paths"] else W * H * 4 * min(args.batch or 4, SPS)
        roof = {"bound": "hbm", "kernel": "trace_kernel", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "avg_launch_ms": round(avg_ms, 4), "launches": int(st["trace_launches"]),
                "seed_kernel_avg_ms": round(st["seed_kernel_ms"] / max(1, st["seed_launches"]), 4),
                "bvh_builder": "device-lbvh" if args.bvh_builder else "host-sah", "bvh_build_ms": round(st["bvh_build_ms"], 4)}
        if bytes_per_path is not None and avg_ms > 0:
            gbs = bytes_per_path * paths_per_launch / (avg_ms * 1e-3) / 1e9
            roof.update({"achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_path": round(bytes_per_path, 1),
                         "algorithmic_bytes_per_launch": int(bytes_per_path * paths_per_launch),
                         "node_tests_per_ray": round(counters["node_tests"] / max(1, counters["rays"]), 2),
                         "tri_tests_per_ray": round(counters["tri_tests"] / max(1, counters["rays"]), 2),
                         "rays_per_path": round(counters["rays"] / max(1, counters["paths"]), 3),
                         "lanes_per_shade_call": round(counters["shade_lanes"] / max(1, counters["shade_calls"]), 1),
                         "lanes_per_box_pass": round(counters["box_lanes"] / max(1, counters["box_passes"]), 1),
                         "lanes_per_leaf_call": round(counters["leaf_lanes"] / max(1, counters["leaf_calls"]), 1),
                         "wave_passes_per_path": {k: round(counters[k] / max(1, counters["paths"]), 3) for k in ("shade_calls", "box_passes", "leaf_calls")},
                         "phase_share_of_wave_cycles": dict(zip(("shade", "refill", "box", "leaf"),
                                                                [round(float(v) / max(1.0, float(sum(counters["phase_cycles"]))), 3) for v in counters["phase_cycles"]]))})
            out["rays_per_s_M"] = round(value * counters["rays"] / max(1, counters["paths"]), 1)
            if alone_ms:
                gbs_alone = bytes_per_path * paths_per_launch / (alone_ms * 1e-3) / 1e9
                roof.update({"avg_launch_ms_alone": round(alone_ms, 4), "achieved_alone": round(gbs_alone, 1), "frac_alone": round(gbs_alone / HBM_PEAK_GBS, 4),
                             "note": "achieved / frac: trace kernel running concurrently with the seed kernel of the next batch (the production schedule); "
                                     "*_alone: the same kernel on the same workload with the chip to itself"})
        # L2: the BVH working set is L2-resident, so the same algorithmic bytes are also quoted against the aggregate L2 bandwidth
        if roof.get("achieved"):
            roof["l2"] = {"achieved": roof["achieved"], "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(roof["achieved"] / L2_PEAK_GBS, 4),
                          "note": "algorithmic traversal bytes per second vs the aggregate L2 bandwidth of MI355X_MICROARCH.md (34.5 TB/s)"}
        # HBM / fabric traffic and instruction counts are PMC measurements (separate rocprofv3 --pmc passes of this same command,
        # tools/prof_pmc.sh, which also writes the JSON read here); per launch like `achieved`
        pmc = latest_pmc_traffic()
        if pmc and (W, H, args.scene) == (1920, 1080, "rtcamp6_v3_1"):
            tk = pmc["kernels"].get("trace_kernel", {})
            if "fetch_bytes_per_path" in tk:
                roof["traffic"] = int(tk["fetch_bytes_per_path"] * paths_per_launch)
                roof["traffic_source"] = pmc["source"]
                roof["traffic_write"] = int(tk.get("write_bytes_per_path", 0) * paths_per_launch)
            # issue: wave-level instructions per path of BOTH kernels vs what the chip can issue (one instruction per wave per
            # ~4.5 cycles is the per-wave rate; the per-SIMD VALU rate is one wave64 instruction per 2 cycles)
            per_path = {k: {f: v[f] for f in ("valu_per_path", "salu_per_path", "vmem_per_path", "lds_per_path") if f in v} for k, v in pmc["kernels"].items()}
            valu = sum(v.get("valu_per_path", 0.0) for v in per_path.values())
            if valu > 0:
                chip_valu_per_s = 256 * 4 * 2.4e9 / 2.0     # 1024 SIMD-32 units, a wave64 VALU instruction every 2 cycles
                roof["issue"] = {"wave_instructions_per_path": per_path, "valu_wave_instructions_per_path_both_kernels": round(valu, 1),
                                 "valu_ceiling_Mpaths_per_s": round(chip_valu_per_s / valu / 1e6, 1), "frac": round(value * 1e6 * valu / chip_valu_per_s, 4),
                                 "source": pmc["source"]}
        out["roofline"] = roof
        # The other kernel of the pair: per-path ISAAC-64 seeding.  Bound neither by HBM nor by MFMA but by how fast ONE wave can
        # issue: the LDS holds 80 generator states per CU (2 KiB each), two consumer waves of 40 lanes run their rounds, and a
        # lone wave issues one instruction per ~4.5 cycles (LDS reads ~8, LDS writes ~16: tools/issueprobe.hip).  Ceiling =
        # 80 states / (256 round steps x ~101 cycles of issue + 11 init blocks x ~510 cycles at 2.4 GHz = 13.2 us) x 256 CUs.
        seed_ms = st["seed_kernel_ms"] / max(1, st["seed_launches"])
        if seed_ms > 0:
            seed_rate = paths_per_launch / (seed_ms * 1e-3) / 1e6
            ceiling = 80.0 / 13.2e-6 * 256 / 1e6
            out["seed_kernel"] = {"kernel": "seed_seg_kernel", "bound": "lds_capacity_x_single_wave_issue_rate", "avg_launch_ms": round(seed_ms, 4),
                                  "achieved": round(seed_rate, 1), "peak": round(ceiling, 1), "unit": "Mpaths/s", "frac": round(seed_rate / ceiling, 4)}

        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py as orc
            o = orc.OracleScene(scene.desc_ptr)
            cores, hw_threads, quota = usable_cpus()
            # one thread first (bounded: a 240x135 image), then every host core at BASELINE's 1920x1080: a
            # 480x270 probe sizes the sample to roughly 12 s of CPU work (at least one full 1080p sampling)
            tc0 = time.perf_counter()
            o.render(240, 135, 1, 2, threads=1)
            one_rate = 240 * 135 * 4 / (time.perf_counter() - tc0) / 1e6
            tc0 = time.perf_counter()
            o.render(480, 270, 1, 2, threads=cores)
            probe_rate = 480 * 270 * 4 / max(time.perf_counter() - tc0, 1e-4) / 1e6
            cw, ch = 1920, 1080
            mult = int(max(1, min(64, 12.0 * probe_rate * 1e6 / (cw * ch * 4))))
            tc0 = time.perf_counter()
            o.render(cw, ch, 2, 2 + mult, threads=cores)
            dt = time.perf_counter() - tc0
            rate = cw * ch * 4 * mult / dt / 1e6
            out["cpu_baseline"] = {"value": round(rate, 4), "unit": "Mpaths/s", "cores": cores, "kind": "port",
                                   "one_thread_value": round(one_rate, 4), "parallel_efficiency": round(rate / (one_rate * cores), 3),
                                   "host_hardware_threads": hw_threads, "cgroup_cpu_quota": quota,
                                   "sample": "%s %dx%d x %d samplings (%d paths) on %d threads = every CPU this container may use (%d hardware threads, cgroup quota %s; "
                                             "persistent pool, 16-pixel chunks, one barrier per sampling like renderer.rs:32-43); f64 oracle with the reference-order "
                                             "BVHs; one thread: 240x135 x 1 sampling" %
                                             (args.scene, cw, ch, mult, cw * ch * 4 * mult, cores, hw_threads, ("%.0f CPUs" % quota) if quota else "none")}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
