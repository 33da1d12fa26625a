#!/usr/bin/env python3
"""bench.py — Mpaths/s of the HIP path-tracing hot path on BASELINE.json's headline configuration.

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): the reference's live scene `rtcamp6_v3_1` at 1920x1080.  One STEP = one
hr_render() call of `--spp-per-step` samplings (default 16, launched 4 at a time) of the whole image on every GPU
= 16 x 1920 x 1080 x 4 camera paths per GPU.  The default K = 64 steps is exactly BASELINE's 1920x1080 x 1024 samplings on one GPU.
With N GPUs the sampling indices are sharded round-robin ((s-1) mod N == rank), every GPU still renders `spp-per-step`
samplings per step (weak scaling), and the fp32 radiance accumulators are summed with ONE all-reduce inside the timed region.
Inputs (scene, textures) are resident in HBM before the timed region; nothing is skipped inside it (seed kernel + trace
kernel + accumulation + the all-reduce).

Two ways to run N > 1, the same library calls underneath:
  * under a launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, WORLD_SIZE set): one process
    per GPU, hr_comm_init_rank + hr_allreduce_accumulator = ncclAllReduce (RCCL) inside libhanamaru_hip.so; torch.distributed
    only carries the ncclUniqueId, the barriers and the max over ranks;
  * without a launcher (`python bench.py --gpus N ...`): ONE process drives N contexts (hr_render only enqueues; one host thread
    keeps N GPUs busy), hr_comm_init_local + hr_allreduce_accumulators = one RCCL group all-reduce over the N devices.  On a box
    with fewer than N GPUs all contexts share device 0 and the library sums them with a kernel on that device (RCCL wants one
    rank per device) — labelled in config.parallelism; that is the path the 1-GPU test tier exercises.

`--total-samplings S` switches to strong scaling: exactly the samplings 1..S in total, sharded over the N GPUs and spread over
--steps steps ("scaling": "strong").  BASELINE config 4 is `--gpus 8 --total-samplings 4096`, config 5
`--scene rtcamp6_dodeca --width 3840 --height 2160 --gpus 8 --total-samplings 1024`.

Extra objects on the JSON line:
  roofline     — trace kernel, `achieved` / `frac` (= `frac_survey_8d`): SURVEY.md §8(d)'s byte booking (32 B per node test, 36 B per
                 triangle, 16 B per sphere, 24 B per cuboid; test counts from an instrumented run of the same kernel on the same
                 seeds) per launch, divided by the kernel's mean launch duration (HIP events on its stream), vs the 8 TB/s HBM peak —
                 the BASELINE-mandated normalisation.  `bound` names what the PMC passes say limits the kernel physically
                 (`l1_ta_issue`: L1 tag lookups and lane divergence; the tree is L2-resident), `bound_contract` the roof the metric
                 is priced against (`hbm`), `pair_bound` the slower kernel of the concurrent pair (the seed kernel).  Next to them:
                 `loaded_bytes` (what the lanes really request with this build's 16-byte node records), the traversal section's
                 share, the traversal-only workload (hr_render_debug, Depth mode), `traffic` (PMC) and `physical`.
  multi_gpu    — per rank: wall time until its samplings were done, the all-reduce as it saw it, its kernels' summed durations;
                 the accumulator's size; max / min over ranks.  `rccl`: what the communicators report about themselves after the timed
                 region (ranks, devices, RCCL version, which code path: rccl-rank | rccl-group | same-device-fallback); `checksum`: the
                 f64 sums of the ranks' own accumulators against the all-reduced total.  A run whose exchange is not verified by both
                 exits with code 3.
  cpu_baseline — the CPU oracle (f64 restatement of the reference path, oracle/) timed on this box's host cores
                 on a bounded sample; reported, not optimised.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# multi-process GPU work on this pool: the host driver only supports dmabuf IPC (without this RCCL fails with `hipIpcGetMemHandle: invalid
# argument`); exported by the image already — kept here for whoever launches bench.py from a bare environment.  Before HIP initialises.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBS = 34500.0  # same guide: aggregate L2 bandwidth
SHADER_CLOCK_HZ = 2.4e9   # MI355X peak engine clock; measured steady 2.39 GHz under this workload (profiles/r02_power_clocks.txt)
L1_LANES_PER_CLOCK = 1.41  # tools/taprobe.hip: 64 lanes' 16-byte loads from distinct lines take a CU 45.5 cycles (profiles/r05_taprobe.txt)
# ceiling of the seed kernel: 80 generator states per CU, each resident for one window of 11 init blocks
# (~510 cycles each) plus 256 round steps whose serial chain is one dependent LDS gather (72 cycles) + four dependent issue
# slots of a lone wave (~101 cycles per step) = 31.5 k cycles = 13.1 us at 2.4 GHz
SEED_CEILING_US_PER_GROUP = 13.2


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


def kernel_source_sha():
    """sha256 over the kernel sources: a PMC summary taken on other kernels is reported as stale."""
    import glob
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "hanamaru-renderer_amd", "csrc", "*"))):
        if f.endswith((".h", ".hip", ".cpp")):
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def latest_pmc_traffic(scene="rtcamp6_v3_1"):
    """profiles/rNN_pmc_traffic[_<scene>].json of the newest round for this scene (written by tools/prof_pmc.sh), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")))
    for f in reversed(files):
        try:
            j = json.load(open(f))
            if "kernels" in j and j.get("scene", "rtcamp6_v3_1") == scene:
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
    ap.add_argument("--spp-per-step", type=int, default=0,
                    help="samplings per step per GPU; 0 (default) = BASELINE's headline: exactly 1,024 samplings per GPU spread over --steps steps, whatever --steps is")
    ap.add_argument("--headline-samplings", type=int, default=1024, help="samplings per GPU of the default plan (BASELINE: 1024)")
    ap.add_argument("--precise", action="store_true", help="option precise_shading 1: the bounce geometry in f64 (DESIGN.md §4.5); default: the library's automatic choice (on for scenes without meshes)")
    ap.add_argument("--no-precise", action="store_true", help="option precise_shading 0: fp32 shading whatever the scene")
    ap.add_argument("--total-samplings", type=int, default=0,
                    help="strong scaling: render exactly samplings 1..S in total, sharded over the GPUs and spread over --steps steps (BASELINE config 4: --gpus 8 --total-samplings 4096)")
    ap.add_argument("--batch", type=int, default=0, help="samplings per kernel launch (0 = the library's automatic choice: 4 at 1080p)")
    ap.add_argument("--scene", default="rtcamp6_v3_1")
    ap.add_argument("--max-leaf", type=int, default=0)
    ap.add_argument("--bvh-builder", type=int, default=-1, help="-1 = by scene size (library default: host SAH below 200,000 primitives), 0 = host SAH, 1 = device LBVH, 2 = device PLOC")
    ap.add_argument("--split-ratio", type=float, default=None, help="early split clipping: -1 automatic (library default), 0 off, > 0 ratio")
    ap.add_argument("--trace-boost", type=int, default=-2, help="-1 = governed (library default), 0 .. 5 = fixed level")
    ap.add_argument("--max-tail-gib", type=float, default=0.0)
    ap.add_argument("--quant-nodes", type=int, default=-1)
    ap.add_argument("--russian-roulette", type=int, default=0, help="NOT the reference's estimator: first iteration that plays (0 = off, default)")
    ap.add_argument("--debug", action="append", default=[], metavar="KEY=VALUE",
                    help="measurement knob of hr_set_debug_option (adv_den, leaf_den, min_waves, kchunk, node_unroll, trace_wgs, seed_mode, seed_split, seed_prio, init_prio, ploc_top)")
    ap.add_argument("--debug-skip", type=int, default=0, help="timing experiments: drop parts of the pipeline after the warm-up (image is garbage)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counters", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import hanamaru_amd as ha
    from hanamaru_amd.sharding import headline_step_range, step_range, strong_plan, strong_step_range

    exit_code = 0
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    launcher = env_world > 1
    rank = int(os.environ.get("RANK", "0")) if launcher else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launcher else 0
    world = env_world if launcher else max(1, args.gpus)
    # HR_BENCH_ONE_DEVICE=1 is a debugging aid for the launcher path on boxes with a single GPU: every rank uses cuda:0 and the
    # collective runs over gloo on a host copy (RCCL refuses two ranks on one device).  Never set by the driver.
    one_device = launcher and os.environ.get("HR_BENCH_ONE_DEVICE") == "1"
    dist = None
    if launcher:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not one_device and torch.cuda.device_count() > local_rank:
            torch.cuda.set_device(local_rank)      # before the process group exists: its barriers run on this rank's own GPU
        dist.init_process_group(backend="gloo" if one_device else "nccl", rank=rank, world_size=world)
    if one_device:
        local_rank = 0
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # the contexts this process drives: (global rank, device)
    if launcher:
        # (a launcher that gives every rank its own device visibility — one visible GPU per process — leaves nothing but device 0 to choose)
        mine = [(rank, local_rank if ndev > local_rank else 0)]
        how = "gloo on a host copy: HR_BENCH_ONE_DEVICE debugging aid" if one_device else "one process per GPU, ncclAllReduce (RCCL) inside libhanamaru_hip.so"
    elif world == 1:
        mine = [(0, 0)]
        how = "single rank"
    elif ndev >= world:
        mine = [(r, r) for r in range(world)]
        how = "one process drives %d GPUs, one RCCL group all-reduce (hr_allreduce_accumulators) inside libhanamaru_hip.so" % world
    else:
        mine = [(r, 0) for r in range(world)]
        how = "FALLBACK: %d contexts share device 0 (the box has %d GPU), summed by the library's kernel on that device instead of RCCL" % (world, ndev)
    torch.cuda.set_device(mine[0][1])
    dev = torch.device("cuda", mine[0][1])

    W, H, SPS = args.width, args.height, args.spp_per_step
    S_TOTAL = args.total_samplings
    # the default plan (neither --spp-per-step nor --total-samplings): the literal BASELINE configuration — samplings 1 .. 1024 per GPU
    # in exactly --steps steps (steps differ by at most one sampling); more steps than samplings: one sampling per step
    HEADLINE = 0
    # the library's launch size (hr_api.hip: --batch, or automatic = launches of about 33 M paths, 4 samplings at 1920x1080): the plan's steps are
    # whole launches wherever --steps allows it, so that no step ends in a short launch (20 steps: 16 x 52 + 4 x 48 samplings, not 4 x 52 + 16 x 51)
    LAUNCH = args.batch if args.batch else min(64, max(4, -(-33177600 // (W * H * 4))))
    if not SPS and not S_TOTAL:
        if args.steps <= args.headline_samplings:
            HEADLINE = args.headline_samplings
            if HEADLINE % LAUNCH or args.steps > HEADLINE // LAUNCH:
                LAUNCH = 1
            SPS = -(-(HEADLINE // LAUNCH) // args.steps) * LAUNCH      # the largest step (for the record; steps hold SPS or SPS - LAUNCH samplings)
        else:
            SPS = 1
    elif not SPS:
        SPS = 16
    if S_TOTAL:
        # strong scaling: a fixed total of samplings 1..S_TOTAL, whatever N is.  A step covers SPS * world consecutive sampling
        # indices as always; SPS is chosen so that --steps steps cover S_TOTAL, and the last step is clipped to it.
        SPS, args.steps = strong_plan(S_TOTAL, args.steps, world)
    scene = ha.Scene(args.scene)
    rs = []
    for _, d in mine:
        r = ha.Renderer(d)
        if args.max_leaf:
            r.set_option("max_leaf", args.max_leaf)
        if args.bvh_builder >= 0:
            r.set_option("bvh_builder", args.bvh_builder)
        if args.quant_nodes >= 0:
            r.set_option("quant_nodes", args.quant_nodes)
        if args.split_ratio is not None:
            r.set_option("split_ratio", args.split_ratio)
        for kv in args.debug:          # builder knobs act at the upload
            k, v = kv.split("=")
            if k == "ploc_top":
                r.set_debug_option(k, float(v))
        r.upload_scene(scene)
        r.set_resolution(W, H)
        r.set_option("batch", args.batch)
        if args.trace_boost >= -1:
            r.set_option("trace_boost", args.trace_boost)
        if args.max_tail_gib:
            r.set_option("max_tail_gib", args.max_tail_gib)
        if args.russian_roulette:
            r.set_option("russian_roulette", args.russian_roulette)
        if args.precise or args.no_precise:
            r.set_option("precise_shading", 1 if args.precise else 0)
        for kv in args.debug:
            k, v = kv.split("=")
            r.set_debug_option(k, float(v))
        rs.append(r)
    r0 = rs[0]
    acc = None
    if one_device:     # the gloo aid sums a torch tensor the accumulator is bound to
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        r0.bind_accumulator(acc.data_ptr())
    lib_rccl = launcher and not one_device
    if lib_rccl:
        uid = [ha.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        r0.comm_init_rank(uid[0], world, rank)
    elif len(rs) > 1:
        ha.comm_init_local(rs)

    def all_reduce():
        """ONE all-reduce of the accumulators, enqueued behind the render work."""
        if lib_rccl:
            r0.allreduce_accumulator()
        elif len(rs) > 1:
            ha.allreduce_accumulators(rs)

    def sync_all():
        for r in rs:
            r.synchronize()

    if lib_rccl or len(rs) > 1:
        all_reduce()      # warm-up: the first collective sets up the rings
        sync_all()
    paths_per_step_gpu = W * H * 4 * SPS

    def run_step(i):
        # step i covers samplings [i*SPS*world + 1, (i+1)*SPS*world]; rank g takes (s-1) % world == g
        for r, (g, _) in zip(rs, mine):
            if HEADLINE:
                b, e, stride = headline_step_range(i, args.steps, HEADLINE, world, g, LAUNCH)
            else:
                b, e, stride = strong_step_range(i, SPS, world, g, S_TOTAL) if S_TOTAL else step_range(i, SPS, world, g)   # --total-samplings: the last step ends at sampling S_TOTAL
            r.render(b, e, stride)

    def barrier():
        if dist is not None:
            dist.barrier()
        sync_all()
        torch.cuda.synchronize()

    # ---- test counts per path: instrumented run of the same kernel on the same seeds (outside the timed region)
    counters = None
    if rank == 0 and not args.no_counters:
        r0.set_option("counters", 1)
        r0.clear()
        r0.render(1, 3)
        r0.synchronize()
        counters = r0.stats()
        r0.set_option("counters", 0)
    for r in rs:
        r.clear()

    for i in range(args.warmup):
        run_step(i)
    sync_all()
    for r in rs:
        r.clear()
    if args.debug_skip:
        for r in rs:
            r.set_debug_option("debug_skip", args.debug_skip)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(i)
    # every context's render work is awaited before the collective is enqueued: costs one host round trip per context (~20 us), and
    # buys the split of the timed region into "rendering" and "the all-reduce" on the line (per rank)
    render_done = []
    for r in rs:
        r.synchronize()
        render_done.append(time.perf_counter() - t0)
    t_ar0 = time.perf_counter()
    all_reduce()
    sync_all()
    if one_device:
        h = acc.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        acc.copy_(h)
    t_ar1 = time.perf_counter()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    # per rank: wall time until its own samplings were done, the all-reduce as it saw it (a rank that arrives early waits in it for
    # the slowest one: the MIN over ranks is the collective's own duration), the kernels' summed HIP-event durations
    per_rank = []
    for r, (g, d), rd in zip(rs, mine, render_done):
        s_ = r.stats()
        per_rank.append({"rank": g, "device": d, "render_ms": round(rd * 1e3, 3), "allreduce_ms": round((t_ar1 - t_ar0) * 1e3, 3),
                         "seed_kernel_ms": round(s_["seed_kernel_ms"], 3), "trace_kernel_ms": round(s_["trace_kernel_ms"], 3),
                         "launches": int(s_["trace_launches"]), "paths": int(s_["paths"])})
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank)
        per_rank = [x for part in gathered for x in part]
    st = r0.stats()
    acc_mean = float(acc.mean().item()) if one_device else float(r0.read_accumulator().mean())   # after the all-reduce: the total
    if os.environ.get("HR_BENCH_CHECKSUM") == "1" and rank == 0:
        sys.stderr.write("accumulator mean after all-reduce: %.9g\n" % acc_mean)
    # ---- outside the timed region: the evidence that the exchange happened.  Every context reports what its communicator says about
    # itself (hr_comm_info: ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion, asked of RCCL now) and the f64 sum of
    # its OWN accumulator (untouched by the all-reduce, which writes the total to a buffer of its own) and of the total it received.
    # The parts must add up to the total: multi_gpu.checksum; a mismatch, or a communicator smaller than --gpus on a box that has the
    # devices, makes the run fail (exit code 3) instead of printing a rate for an exchange that did not take place.
    exchange = []
    for r, (g, d) in zip(rs, mine):
        ci = r.comm_info()
        own = None if one_device else r.accumulator_sum(False)
        tot = None
        if not one_device:
            tot = r.accumulator_sum(True) if (lib_rccl or len(rs) > 1) else own
        exchange.append({"rank": g, "device": d, "comm": ci, "own_sum": own, "total_sum": tot})
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, exchange)
        exchange = [x for part in gathered for x in part]
    exchange.sort(key=lambda x: x["rank"])

    # ---- outside the timed region, rank 0: the post chain (renderer.rs:64-90 as two HIP kernels) on the all-reduced accumulator
    post = None
    if rank == 0 and not args.debug_skip:
        n_s = S_TOTAL if S_TOTAL else HEADLINE * world if HEADLINE else SPS * world * args.steps
        p0 = r0.stats()["post_kernel_ms"]
        tp0 = time.perf_counter()
        img = r0.resolve(max(1, n_s))
        tp1 = time.perf_counter()
        post = {"kernels": "tonemap_gamma_kernel + bilateral_quantise_kernel", "resolution": [W, H], "post_kernel_ms": round(r0.stats()["post_kernel_ms"] - p0, 4),
                "hr_resolve_wall_ms": round((tp1 - tp0) * 1e3, 3), "image_mean_u8": round(float(img.mean()), 3),
                "note": "hr_resolve on the all-reduced accumulator, once, outside the timed region: HIP-event time of the two kernels, and the call's wall time incl. the W x H x 3 byte copy to the host"}
    # ---- outside the timed region, rank 0: the trace kernel with the chip to itself, and the traversal-only workload
    alone_ms = None
    trav = None
    if rank == 0 and not args.debug_skip and not args.no_counters:
        # seed kernel skipped: the hand-off buffers still hold the previous batches' draws, so the workload is the same
        r0.set_debug_option("debug_skip", 2)
        r0.render(*step_range(0, SPS, world, mine[0][0]))
        r0.synchronize()
        st2 = r0.stats()
        r0.set_debug_option("debug_skip", 0)
        alone_ms = (st2["trace_kernel_ms"] - st["trace_kernel_ms"]) / max(1, st2["trace_launches"] - st["trace_launches"])
        # traversal only: DebugRenderer's Depth mode through the render kernel's traversal — camera rays,
        # closest hits, nothing shaded, no seed kernel beside it.  Counts from one instrumented launch, time from ten plain ones.
        r0.clear()
        r0.set_option("counters", 1)
        r0.render_debug(2)
        tc = r0.stats()
        r0.set_option("counters", 0)
        r0.clear()
        for _ in range(10):
            r0.render_debug(2)
        ts = r0.stats()
        r0.clear()
        trav = (tc, ts["debug_kernel_ms"] / max(1, ts["debug_launches"]))

    if rank == 0:
        samplings_run = S_TOTAL if S_TOTAL else HEADLINE * world if HEADLINE else SPS * world * args.steps      # sampling indices 1..samplings_run, over all GPUs
        total_paths = W * H * 4 * samplings_run
        assert sum(x["paths"] for x in per_rank) == total_paths, (per_rank, total_paths)
        value = total_paths / elapsed / 1e6
        accum_bytes = W * H * 3 * 4
        render_ms = [x["render_ms"] for x in per_rank]
        ar_ms = [x["allreduce_ms"] for x in per_rank]
        out = {
            "metric": "Mpaths/sec at 1920x1080x1024spp (rtcamp6 scene); achieved GB/s in BVH traversal",
            "value": round(value, 3), "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if S_TOTAL else "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "reference scene assets shipped in-repo (assets/), per-path ISAAC-64 seeds as in renderer.rs:165-168",
            "config": {"workload": "%s %dx%d x %d samplings (x4 sub-samples = %d paths) in this run: %d steps x %s samplings per step per GPU x %d GPU(s)%s%s"
                                   % (args.scene, W, H, samplings_run, total_paths, args.steps, ("%d or %d" % (SPS - LAUNCH, SPS)) if HEADLINE and (HEADLINE // LAUNCH) % args.steps else str(SPS), world,
                                      (", last step clipped to --total-samplings %d" % S_TOTAL) if S_TOTAL else "",
                                      ("; the default plan: BASELINE's %d samplings per GPU whatever --steps is, steps of whole launches (%d samplings)" % (HEADLINE, LAUNCH)) if HEADLINE else ""),
                       "shading": ["fp32 shading (megakernel)", "precise shading: bounce geometry in f64 from the f64 draws, in the megakernel", "precise shading: bounce geometry in f64 from the f64 draws, in the split pipeline",
                                   "fp32 shading in the split pipeline (debug)"][int(st.get("shading_in_force", 0))] + (" [option precise_shading %d]" % (1 if args.precise else 0) if args.precise or args.no_precise else " [automatic]"),
                       "samplings_total": samplings_run, "paths_total": total_paths,
                       "samplings_per_step_per_gpu": SPS, "samplings_per_launch_requested": args.batch, "paths_per_step": round(total_paths / args.steps) if HEADLINE else paths_per_step_gpu * world,
                       "parallelism": "spp-sharded x%d, one all-reduce (%s)" % (world, how), "devices": sorted(set(d for _, d in mine)) if not launcher else [local_rank],
                       "estimator": "reference (no Russian roulette)" if not args.russian_roulette else "NON-PARITY: Russian roulette from iteration %d" % args.russian_roulette},
            "rays_per_s_M": None,
            # the timed region, split: every rank's wall time until its own samplings were done, then ONE all-reduce of the accumulators
            # (fp32 W x H x 3).  A rank that is done early waits inside the collective for the slowest one, so the collective's own
            # duration is the MIN over ranks; a wide max - min spread of render_ms means one GPU was slower, not the exchange.
            "multi_gpu": {"accumulator_bytes": accum_bytes,
                          "render_ms": {"max": max(render_ms), "min": min(render_ms)},
                          "allreduce_ms": {"max": max(ar_ms), "min": min(ar_ms)},
                          "allreduce_share_of_timed_region": round(min(ar_ms) * 1e-3 / elapsed, 5),
                          "per_rank": sorted(per_rank, key=lambda x: x["rank"]),
                          "rccl": None, "checksum": None,
                          "note": ("one process drives all contexts: render_ms of rank k is observed after ranks 0..k-1 were awaited (a lower bound for the "
                                   "later ranks' own time is their kernels' summed durations)" if (not launcher and world > 1) else "one process per rank")},
        }
        # what the communicators said about themselves, and the checksum of the exchange
        comms = [x["comm"] for x in exchange]
        paths_seen = sorted(set(c["path"] for c in comms))
        rccl = {"path": paths_seen[0] if len(paths_seen) == 1 else paths_seen,
                "nranks": max(1, min(c["nranks"] for c in comms)) if world > 1 else 1,
                "ranks_seen": sorted(c["rank"] for c in comms) if world > 1 else [0],
                "devices_seen": sorted(set(c["device"] for c in comms)),
                "version": max(c["rccl_version"] for c in comms),
                "allreduces_per_rank": sorted(set(c["allreduces"] for c in comms)),
                "source": "hr_comm_info() of every context after the timed region: ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion as the communicator answers them"
                          if world > 1 and not one_device else ("single rank: no communicator, no collective" if world == 1 else "HR_BENCH_ONE_DEVICE debugging aid: torch.distributed gloo on a host copy, no RCCL")}
        if world > 1 and not one_device and any(c["path"].startswith("rccl") for c in comms):
            try:      # which RCCL build the collective ran on: ONE per process — the one the host (torch) had mapped, if any
                lp, reused = ha.comm_library()
                n_mapped = len({line.split()[-1] for line in open("/proc/self/maps") if "librccl" in line})
                rccl["library"] = {"path": lp, "reused_the_hosts": reused, "librccl_objects_mapped_in_this_process": n_mapped}
            except (ha.HipError, OSError) as ex:
                rccl["library"] = {"error": str(ex)}
        out["multi_gpu"]["rccl"] = rccl
        failures = []
        if not one_device:
            parts = [sum(x["own_sum"][k] for x in exchange) for k in range(3)]
            totals = [x["total_sum"] for x in exchange]
            total = totals[0]
            rel = max(abs(parts[k] - total[k]) / max(abs(total[k]), 1e-30) for k in range(3))
            same = all(t == total for t in totals)
            out["multi_gpu"]["checksum"] = {"sum_of_parts": [float("%.12g" % v) for v in parts], "total": [float("%.12g" % v) for v in total], "rel_err": float("%.3g" % rel),
                                            "totals_identical_on_all_ranks": same,
                                            "definition": "per-channel f64 device sums (hr_accumulator_sum): every rank's OWN accumulator, added over the ranks, against the all-reduced total "
                                                          "rank 0 holds; rel_err = max over channels of |parts - total| / |total| (fp32 all-reduce rounding: ~1e-8); a run with rel_err > 1e-6 exits with code 3"}
            if rel > 1e-6 or not (total[0] > 0.0):
                failures.append("checksum: the ranks' accumulators do not add up to the all-reduced total (rel_err %.3g)" % rel)
            if not same:
                failures.append("checksum: the ranks hold different totals")
        if world > 1 and not one_device:
            if rccl["nranks"] != world or rccl["ranks_seen"] != list(range(world)):
                failures.append("communicator reports %s ranks %s, --gpus %d" % (rccl["nranks"], rccl["ranks_seen"], world))
            have_devices = launcher or ndev >= world
            if have_devices and paths_seen not in (["rccl-rank"], ["rccl-group"]):
                failures.append("the box has the devices, but the exchange did not run over RCCL (path %s)" % paths_seen)
            # distinct devices: ncclCommCuDevice answers with the process's own ordinal, so the check only means something where this process
            # sees all of the job's devices (a launcher that restricts every rank to one visible GPU makes them all "device 0")
            if ndev >= world and have_devices and len(rccl["devices_seen"]) != world:
                failures.append("the exchange did not run on %d distinct devices (devices %s)" % (world, rccl["devices_seen"]))
        out["multi_gpu"]["exchange_verified"] = not failures
        launches = max(1, st["trace_launches"])
        avg_ms = st["trace_kernel_ms"] / launches
        # the library may cap the samplings per launch (hand-off buffer size): use what was actually launched
        paths_per_launch = st["paths"] / launches if st["paths"] else W * H * 4 * min(args.batch or 4, SPS)
        quant = args.quant_nodes != 0
        node_b = 16 if quant else 32
        builder = {0: "host-sah", 1: "device-lbvh", 2: "device-ploc"}.get(int(st["bvh_builder_used"]), "?")
        # `bound`: what the PMC passes say limits the kernel (the CUs' L1 tag lookups / texture-addresser issue and lane divergence — the
        # tree is L2-resident, physical HBM traffic is ~6 % of peak); `bound_contract` / `peak`: BASELINE's metric prices the traversal
        # against the HBM peak, and that normalisation is what `achieved` / `frac` are (SURVEY.md 8(d)'s byte booking).
        roof = {"schema": "r04+: achieved / frac = SURVEY 8(d) byte booking (until r03 these two fields were what is now loaded_bytes.achieved / .frac)",
                "bound": "l1_ta_issue", "bound_contract": "hbm", "kernel": "trace_kernel" if int(st.get("shading_in_force", 0)) < 2 else "split pipeline: wf_start_kernel + 10 x (wf_traverse_kernel, wf_shade_kernel) — avg_launch_ms is the whole sequence of a launch",
                "pair_bound": "seed_seg_kernel" if st["seed_kernel_ms"] / max(1, st["seed_launches"]) >= avg_ms else "trace_kernel",   # the slower kernel of the concurrent pair
                "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "frac_survey_8d": None, "traffic": None,
                "avg_launch_ms": round(avg_ms, 4), "launches": int(st["trace_launches"]),
                "seed_kernel_avg_ms": round(st["seed_kernel_ms"] / max(1, st["seed_launches"]), 4),
                "bvh_builder": builder, "bvh_build_ms": round(st["bvh_build_ms"], 4),
                "priority_governor": {"level": int(st["governor_level"]), "launches_judged": int(st["governor_decisions"]), "moves": int(st["governor_moves"]),
                                      "trace_workgroups": int(st["governor_budget"]) or "all", "trace_workgroup_moves": int(st["governor_budget_moves"]),
                                      "note": "which kernel's waves come first (0 = the seed kernel's producer waves .. 4 = the trace kernel's box and leaf phases), decided on the device from the kernels' own time stamps, launch by launch"}}

        def loaded_bytes(c):   # what the lanes request for the tests they perform (device_scene.h record sizes)
            return node_b * c["node_tests"] + 48 * c["tri_tests"] + 16 * c["sphere_tests"] + 32 * c["cuboid_tests"]

        def survey_bytes(c):   # the information content of a test, whatever the layout
            return 32 * c["node_tests"] + 36 * c["tri_tests"] + 16 * c["sphere_tests"] + 24 * c["cuboid_tests"]

        if counters is not None and avg_ms > 0:
            npaths = max(1, counters["paths"])
            lb, sb = loaded_bytes(counters) / npaths, survey_bytes(counters) / npaths
            gbs = lb * paths_per_launch / (avg_ms * 1e-3) / 1e9
            sgbs = sb * paths_per_launch / (avg_ms * 1e-3) / 1e9
            pc = [float(v) for v in counters["phase_cycles"]]
            share = dict(zip(("shade", "refill", "box", "leaf"), [round(v / max(1.0, sum(pc)), 3) for v in pc]))
            trav_share = (pc[2] + pc[3]) / max(1.0, sum(pc))
            roof.update({
                "achieved": round(sgbs, 1), "frac": round(sgbs / HBM_PEAK_GBS, 4), "frac_survey_8d": round(sgbs / HBM_PEAK_GBS, 4),
                "definition": "SURVEY.md 8(d): 32 B per node test + 36 B per triangle test + 16 B per sphere test + 24 B per cuboid test that the kernel performs "
                              "(test counts from the instrumented build of the same kernel on the same seeds) per launch / the kernel's mean launch duration (HIP events on "
                              "its stream), over the HBM3E peak.  It is the BASELINE-mandated normalisation, comparable across rounds and record layouts; it is NOT a "
                              "physical share of HBM bandwidth (the tree is L2-resident: see `traffic`, `l2`, `physical`), and with the chip to itself it passes 1",
                "frac_across_rounds": "r04: 0.83 (4,354 B/path, every scene.intersect call of the reference traced, every trace workgroup kept).  r05 lowers the ratio twice while raising "
                                      "Mpaths/s: NEE shadow rays known to add nothing are not traced (fewer tests = fewer algorithmic bytes per path: nee_shadow_rays_culled_per_path), and "
                                      "where the trace kernel is the faster kernel of the pair the governor keeps part of its workgroups out (priority_governor.trace_workgroups), so the same "
                                      "bytes take longer on purpose — the seed kernel beside it gains more than that (profiles/r05_bench_full_wave_budget_off.json.log: all workgroups kept)",
                "algorithmic_bytes_per_path": round(sb, 1), "algorithmic_bytes_per_launch": int(sb * paths_per_launch),
                "loaded_bytes": {"bytes_per_path": round(lb, 1), "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                                 "note": "the bytes the lanes really request for those tests with this build's records (%d B per node visit on the %s records, 48 B per triangle, "
                                         "16 B per sphere, 32 B per cuboid): the conservative figure, below the 8(d) booking because a node visit is a 16-byte load here"
                                         % (node_b, "16-byte quantised" if quant else "32-byte fp32")},
                "node_tests_per_ray": round(counters["node_tests"] / max(1, counters["rays"]), 2),
                "tri_tests_per_ray": round(counters["tri_tests"] / max(1, counters["rays"]), 2),
                "rays_per_path": round(counters["rays"] / npaths, 3),
                # scene.intersect calls of the reference that the kernel does not make: NEE shadow rays known to contribute exactly nothing before
                # they are traced (sample on the emitter's far side; GGX with the emitter below the horizon) — bit-identical image (pt_core.h nee_setup)
                "nee_shadow_rays_culled_per_path": round(counters["shadow_culled"] / npaths, 3),
                "reference_rays_per_path": round((counters["rays"] + counters["shadow_culled"]) / npaths, 3),
                "lanes_per_shade_call": round(counters["shade_lanes"] / max(1, counters["shade_calls"]), 1),
                "lanes_per_box_pass": round(counters["box_lanes"] / max(1, counters["box_passes"]), 1),
                "lanes_per_leaf_call": round(counters["leaf_lanes"] / max(1, counters["leaf_calls"]), 1),
                "wave_passes_per_path": {k: round(counters[k] / npaths, 3) for k in ("shade_calls", "box_passes", "leaf_calls")},
                "phase_share_of_wave_cycles": share,
                # north_star's "traversal section": the box + leaf phases' share of the wave cycles applied to the kernel's time
                "traversal_section": {"share_of_wave_cycles": round(trav_share, 3), "ms_per_launch": round(avg_ms * trav_share, 4),
                                      "achieved": round(gbs / max(trav_share, 1e-9), 1), "frac": round(gbs / max(trav_share, 1e-9) / HBM_PEAK_GBS, 4),
                                      "survey_8d_normalised": round(sgbs / max(trav_share, 1e-9) / HBM_PEAK_GBS, 4),
                                      "note": "the bytes the lanes load for the traversal (loaded_bytes) over the time the kernel's waves spend in the box and leaf phases (shade and "
                                              "refill excluded), against the HBM peak; survey_8d_normalised: the same with SURVEY 8(d)'s byte booking — not a physical fraction, it may pass 1"}})
            out["rays_per_s_M"] = round(value * counters["rays"] / npaths, 1)
            if alone_ms:
                roof.update({"avg_launch_ms_alone": round(alone_ms, 4), "achieved_alone": round(sgbs * avg_ms / alone_ms, 1),
                             "hbm_normalised_alone": round(sgbs * avg_ms / alone_ms / HBM_PEAK_GBS, 4),
                             "note": "achieved / frac: trace kernel running concurrently with the seed kernel of the next batch (the production schedule); "
                                     "*_alone: the same kernel on the same workload with the chip to itself.  The bytes are served by the CUs' L1 and "
                                     "the L2 (physical HBM traffic: `traffic`), so with the chip to itself their rate can pass the HBM "
                                     "peak: hbm_normalised_alone is NOT a fraction of anything physical — l2.frac_alone and physical.ta_busy_frac are"})
                roof["loaded_bytes"]["normalised_alone"] = round(gbs * avg_ms / alone_ms / HBM_PEAK_GBS, 4)
            # The kernel's PHYSICAL roofline, from this run's own counts: every test is lane-loads the L1 has to look up (one 16-byte load per node
            # test and per sphere test, two per cuboid test, three per triangle test), and a CU's L1 serves 1.41 of them per clock whatever the
            # wave's locality (tools/taprobe.hip, profiles/r05_taprobe.txt).  The PMC pass's TCP_TOTAL_CACHE_ACCESSES (physical.l1_lookup, replayed
            # from profiles/) counts the same thing on the hardware side: within 10 % of this figure.
            ll = (counters["node_tests"] + 3 * counters["tri_tests"] + counters["sphere_tests"] + 2 * counters["cuboid_tests"]) / npaths
            l1_peak = 256 * SHADER_CLOCK_HZ * L1_LANES_PER_CLOCK
            roof["l1_lookup"] = {"lane_loads_per_path": round(ll, 1), "peak": round(l1_peak / 1e9, 1), "unit": "G lane-loads/s",
                                 "achieved": round(ll * paths_per_launch / (avg_ms * 1e-3) / 1e9, 1), "frac": round(ll * paths_per_launch / (avg_ms * 1e-3) / l1_peak, 4),
                                 "frac_alone": round(ll * paths_per_launch / (alone_ms * 1e-3) / l1_peak, 4) if alone_ms else None,
                                 "note": "traversal loads only (shading, textures and the hand-off records add a few per cent), counted by the instrumented kernel in THIS run; "
                                         "peak = 256 CUs x 2.4 GHz x 1.41 lane-loads per clock (measured); frac beside the seed kernel, frac_alone with the chip to itself"}
            # L2 is the level that serves the tree: the same bytes against its aggregate bandwidth
            roof["l2"] = {"achieved": round(gbs, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / L2_PEAK_GBS, 4),
                          "frac_alone": round(gbs * avg_ms / alone_ms / L2_PEAK_GBS, 4) if alone_ms else None,
                          "note": "the loaded bytes per second vs the aggregate L2 bandwidth of MI355X_MICROARCH.md (34.5 TB/s); part of them is served by the CUs' L1"}
        if trav is not None:
            tc, tms = trav
            rays = max(1, tc["rays"])
            tb = loaded_bytes(tc)
            roof["traversal_only"] = {
                "workload": "hr_render_debug Depth mode: %dx%dx4 pinhole camera rays through the render kernel's traversal, closest hit, nothing shaded, chip to itself" % (W, H),
                "ms_per_launch": round(tms, 4), "Mrays_per_s": round(rays / (tms * 1e-3) / 1e6, 1),
                "node_tests_per_ray": round(tc["node_tests"] / rays, 2), "tri_tests_per_ray": round(tc["tri_tests"] / rays, 2),
                "lanes_per_box_pass": round(tc["box_lanes"] / max(1, tc["box_passes"]), 1),
                "achieved": round(tb / (tms * 1e-3) / 1e9, 1), "bound": "l2", "peak": L2_PEAK_GBS, "unit": "GB/s",
                "frac": round(tb / (tms * 1e-3) / 1e9 / L2_PEAK_GBS, 4),
                "hbm_normalised": round(tb / (tms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "survey_8d_normalised": round(survey_bytes(tc) / (tms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "coherent camera rays: the CUs' L1 and the L2 serve nearly all of the loaded bytes, so the rate is priced against the aggregate L2 bandwidth; "
                        "the *_normalised figures divide the same bytes (and SURVEY 8(d)'s) by the HBM peak for comparison with the render kernel's line and are NOT fractions of anything physical (they exceed 1)"}
        # HBM / fabric traffic and instruction counts are PMC measurements (separate rocprofv3 --pmc passes of this same command,
        # tools/prof_pmc.sh, which also writes the JSON read here); per launch like `achieved`
        pmc = latest_pmc_traffic(args.scene)
        if pmc and (W, H) == (1920, 1080):
            stale = pmc.get("csrc_sha") != kernel_source_sha()
            tk = pmc["kernels"].get("trace_kernel", {})
            if "fetch_bytes_per_path" in tk:
                roof["traffic"] = int(tk["fetch_bytes_per_path"] * paths_per_launch)
                roof["traffic_source"] = pmc["source"]
                roof["traffic_stale"] = stale   # true: the PMC passes were taken on other kernel sources than the ones running now
                roof["traffic_write"] = int(tk.get("write_bytes_per_path", 0) * paths_per_launch)
            phys = {k: tk[k] for k in ("l2_hit_rate", "valu_lane_utilisation", "ta_busy_frac", "l1_line_accesses_per_path", "l1_to_l2_requests_per_path") if k in tk}
            roof["replayed_from_profiles"] = {"fields": ["traffic", "traffic_write", "physical", "issue"], "file": pmc["source"],
                                              "note": "PMC figures are NOT measured in this run: separate rocprofv3 --pmc passes of this same command (tools/prof_pmc.sh), read back from the "
                                                      "newest profiles/rNN_pmc_traffic.json; traffic_stale says whether the kernel sources have changed since"}
            if phys:
                phys["note"] = "what bounds the kernel physically (PMC, kernel alone on the chip): the L1's tag lookups — one cache line per clock per CU — and SIMD lane divergence, not bytes"
                if "l1_line_accesses_per_path" in phys and avg_ms > 0:
                    # The kernel's physical roofline.  tools/taprobe.hip (profiles/r05_taprobe.txt): a vector load costs the CU's L1 about 0.71 cycles per
                    # ACTIVE LANE whatever its width (8 or 16 bytes) and whatever the lanes' locality (45.5 CU cycles per 64-lane load with the 16-byte
                    # slots spread over their lines, as a tree's records are; floor 16.4 cycles per instruction), unless 4 NEIGHBOURING lanes read one
                    # aligned 64-byte block — a traversal's lanes rarely do.  So the L1 serves 256 CUs x shader clock x 1.41 lane accesses per second (the
                    # probe's own pointer chase: 864 - 890 G/s), and TCP_TOTAL_CACHE_ACCESSES counts what the kernel asks of it.
                    peak = 256 * SHADER_CLOCK_HZ * L1_LANES_PER_CLOCK
                    acc = phys["l1_line_accesses_per_path"] * paths_per_launch
                    phys["l1_lookup"] = {"lane_accesses_per_launch": int(acc), "peak_per_s": peak, "unit": "L1 lane accesses/s",
                                         "achieved_per_s": round(acc / (avg_ms * 1e-3), 0), "frac": round(acc / (avg_ms * 1e-3) / peak, 4),
                                         "frac_alone": round(acc / (alone_ms * 1e-3) / peak, 4) if alone_ms else None,
                                         "note": "1.41 lane accesses per clock per CU (measured law: profiles/r05_taprobe.txt); frac = beside the seed kernel (which asks almost nothing of the L1 "
                                                 "but shares the SIMDs' issue slots), frac_alone = the same kernel with the chip to itself"}
                roof["physical"] = phys
            # issue: wave-level instructions per path of BOTH kernels vs what the chip can issue (one instruction per wave per
            # ~4.5 cycles is the per-wave rate; the per-SIMD VALU rate is one wave64 instruction per 2 cycles)
            per_path = {k: {f: v[f] for f in ("valu_per_path", "salu_per_path", "vmem_per_path", "lds_per_path") if f in v} for k, v in pmc["kernels"].items()}
            valu = sum(v.get("valu_per_path", 0.0) for v in per_path.values())
            if valu > 0:
                chip_valu_per_s = 256 * 4 * 2.4e9 / 2.0     # 1024 SIMD-32 units, a wave64 VALU instruction every 2 cycles
                roof["issue"] = {"wave_instructions_per_path": per_path, "valu_wave_instructions_per_path_both_kernels": round(valu, 1),
                                 "valu_ceiling_Mpaths_per_s": round(chip_valu_per_s / valu / 1e6, 1), "frac": round(value / world * 1e6 * valu / chip_valu_per_s, 4),
                                 "source": pmc["source"], "stale": stale}
        out["roofline"] = roof
        if post:
            out["post_chain"] = post
        # The other kernel of the pair: per-path ISAAC-64 seeding.  Bound neither by HBM nor by MFMA but by LDS capacity x the
        # serial chain of ONE wave (SEED_CEILING_US_PER_GROUP above).
        seed_ms = st["seed_kernel_ms"] / max(1, st["seed_launches"])
        if seed_ms > 0:
            seed_rate = paths_per_launch / (seed_ms * 1e-3) / 1e6
            ceiling = 80.0 / (SEED_CEILING_US_PER_GROUP * 1e-6) * 256 / 1e6
            out["seed_kernel"] = {"kernel": "seed_seg_kernel", "bound": "lds_capacity_x_single_wave_serial_chain", "avg_launch_ms": round(seed_ms, 4),
                                  "achieved": round(seed_rate, 1), "peak": round(ceiling, 1), "unit": "Mpaths/s", "frac": round(seed_rate / ceiling, 4)}
            # the headline against the ceiling of the kernel that bounds it, and where the difference goes: the consumer
            # wave's cycles per group of 40 paths by phase, from the phase-timing build of the seed kernel run alone, outside the timed region
            head = {"ceiling": round(ceiling, 1), "unit": "Mpaths/s", "frac_of_ceiling": round(value / world / ceiling, 4),
                    "note": "whole-pipeline rate per GPU over the seed kernel's model ceiling (80 LDS-resident generator states per CU x a window of 11 x 510 cycles + a round of 256 x 101 cycles)"}
            if world == 1 and not args.no_counters and not args.debug_skip:
                try:
                    r0.set_debug_option("seed_prof", 1)
                    r0.set_debug_option("debug_skip", 16)      # no trace kernel beside it
                    s_a = r0.stats()
                    r0.render(1, 9)
                    r0.synchronize()
                    s_b = r0.stats()
                    ph = [b - a for a, b in zip(s_a["seed_phase_cycles"], s_b["seed_phase_cycles"])]
                    groups = max(1, ph[7])
                    alone_seed_ms = (s_b["seed_kernel_ms"] - s_a["seed_kernel_ms"]) / max(1, s_b["seed_launches"] - s_a["seed_launches"])
                    per = {k: ph[i] / groups for i, k in enumerate(["issue_reg_loads", "wait_regs", "window_11_init_blocks", "barrier_b", "round_256_steps_and_record", "fixup_note", "barrier_a"])}
                    total = sum(per.values())
                    model = {"window": 11 * 510.0, "round": 256 * 101.0}
                    probe = {"window": 11 * 577.0, "round": 256 * 107.6}     # NOTES.md A: what the isolated loops measure (a block of the three-run window 577 cycles, a round step 107.6: profiles/r03_roundprobe4.txt)
                    ticks_to_cycles = SHADER_CLOCK_HZ / 1e8                 # s_memtime counts the 100 MHz constant clock
                    cyc = {k: v * ticks_to_cycles for k, v in per.items()} if total < 5000 else dict(per)
                    tot_c = sum(cyc.values())
                    groups_per_wave = paths_per_launch / 40.0 / (2 * 256)
                    in_phases_ms = tot_c * groups_per_wave / SHADER_CLOCK_HZ * 1e3
                    head["account"] = {
                        "cycles_per_group": {k: round(v, 0) for k, v in cyc.items()}, "cycles_per_group_total": round(tot_c, 0),
                        "model_cycles_per_group": model["window"] + model["round"], "probe_floor_cycles_per_group": probe["window"] + probe["round"],
                        "terms": {
                            "1_phases_vs_model": round((model["window"] + model["round"]) / tot_c, 4),
                            "  window_over_model": round(cyc["window_11_init_blocks"] / model["window"], 4), "  window_over_probe_floor": round(cyc["window_11_init_blocks"] / probe["window"], 4),
                            "  round_over_model": round(cyc["round_256_steps_and_record"] / model["round"], 4), "  round_over_probe_floor": round(cyc["round_256_steps_and_record"] / probe["round"], 4),
                            "  bookkeeping_share": round((tot_c - cyc["window_11_init_blocks"] - cyc["round_256_steps_and_record"]) / tot_c, 4),
                            "2_kernel_ramp_tail_fixup": round(in_phases_ms / alone_seed_ms, 4) if alone_seed_ms > 0 else None,
                            "3_beside_the_trace_side": round(alone_seed_ms / seed_ms, 4) if seed_ms > 0 else None,
                            "4_pair_over_seed_kernel": round(value / seed_rate, 4)},
                        "seed_kernel_alone_ms": round(alone_seed_ms, 4),
                        "note": "frac_of_ceiling = term 1 x term 2 x term 3 x term 4: (1) the consumer wave's measured cycles per group against the model's window + round, "
                                "(2) the time its groups account for against the kernel's own duration alone (launch ramp, the last groups, the fix-up pass), (3) alone against beside the "
                                "trace side, (4) the pipeline's rate against the seed kernel's.  window / round _over_probe_floor: against what the isolated probes measure for the same loops "
                                "(NOTES.md A) — at ~1.0 nothing is left in the kernel that the probes do not have too"}
                except Exception as ex:      # a measurement aid: never the reason a bench line is missing
                    head["account"] = {"error": str(ex)}
                finally:
                    r0.set_debug_option("seed_prof", 0)
                    r0.set_debug_option("debug_skip", 0)
            out["headline"] = head

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
        if failures:
            sys.stderr.write("bench.py: the multi-GPU exchange is not verified:\n  " + "\n  ".join(failures) + "\n")
            exit_code = 3
    for r in rs:
        r.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
