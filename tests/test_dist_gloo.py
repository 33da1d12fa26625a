"""N > 1 path on CPU: two processes over gloo shard the sampling indices exactly as bench.py does on GPUs
(hanamaru_amd.sharding), each renders its shard, one all-reduce sums the fp32 accumulators.  The renderer used
here is the host emulation of the kernels (no GPU in this tier); what is under test is the sharding + collective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPS, STEPS = 40, 24, 2, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    for p in (os.path.join(ROOT, "hanamaru-renderer_amd", "python"), os.path.join(ROOT, "tests", "emu")):
        sys.path.insert(0, p)
    import emu_py
    import hanamaru_amd as ha
    from hanamaru_amd.sharding import step_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = ha.Scene("cornell_mini")
    e = emu_py.EmuScene(sc.desc_ptr)
    acc = np.zeros((H, W, 3), dtype=np.float32)
    for step in range(STEPS):
        b, en, st = step_range(step, SPS, world, rank)
        e.render(W, H, b, en, st, threads=2, acc=acc)
    t = torch.from_numpy(acc)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_covers_every_sampling_once():
    from hanamaru_amd.sharding import samplings_of
    for world in (1, 2, 4, 8):
        for sps in (1, 3, 16):
            seen = []
            for step in range(3):
                for rank in range(world):
                    got = samplings_of(step, sps, world, rank)
                    assert len(got) == sps and all((s - 1) % world == rank for s in got)
                    seen += got
            assert sorted(seen) == list(range(1, 3 * sps * world + 1))


def test_strong_scaling_plan_covers_exactly_the_total():
    """bench.py --total-samplings S: whatever S, step count and world size, the ranks' clipped ranges over all steps are the samplings
    1..S, each exactly once, rank r holding those with (s - 1) mod world == r (BASELINE config 4: S = 4096, world 8; config 5: 1024)."""
    from hanamaru_amd.sharding import strong_plan, strong_step_range
    for total, steps, world in [(4096, 32, 8), (1024, 16, 8), (21, 2, 8), (21, 3, 1), (1, 64, 8), (7, 64, 2), (1000, 7, 4), (64, 2, 8), (5, 1, 8)]:
        sps, nsteps = strong_plan(total, steps, world)
        assert 1 <= nsteps <= max(steps, 1) and sps * world * nsteps >= total > sps * world * (nsteps - 1)
        seen = []
        for step in range(nsteps):
            for rank in range(world):
                b, e, st = strong_step_range(step, sps, world, rank, total)
                got = list(range(b, e, st))
                assert all((s - 1) % world == rank for s in got) and len(got) <= sps
                seen += got
        assert sorted(seen) == list(range(1, total + 1)), (total, steps, world)
    assert strong_plan(4096, 32, 8) == (16, 32)          # the command of DESIGN.md 7: 16 samplings per step per GPU, as the weak-scaling default


@pytest.mark.timeout(300)
def test_headline_plan_is_exactly_1024_samplings_per_gpu_for_any_step_count():
    """bench.py's default plan (no --spp-per-step, no --total-samplings): BASELINE's 1,024 samplings per GPU in exactly --steps steps,
    whatever --steps is — the driver's `--steps 20` renders samplings 1..1024, not 320 (main.rs:1249-1251: `-s` is the sampling count) —, in
    steps of whole kernel launches where the step count allows it (20 steps at 1080p: 16 x 52 + 4 x 48 samplings)."""
    sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))
    from hanamaru_amd.sharding import headline_step_range
    for world, steps, unit in [(w, k, u) for w in (1, 2, 3, 8) for k in (1, 7, 20, 64, 256, 333, 1000, 1024) for u in (1, 4)]:
        if True:
            seen, per_rank = [], [0] * world
            sizes = set()
            for i in range(steps):
                n_step = 0
                for r in range(world):
                    b, e, st = headline_step_range(i, steps, 1024, world, r, unit)
                    mine = list(range(b, e, st))
                    assert all((s - 1) % world == r for s in mine)
                    seen += mine
                    per_rank[r] += len(mine)
                    n_step += len(mine)
                assert n_step % world == 0 and n_step >= world     # every rank the same count, at least one sampling per step
                sizes.add(n_step // world)
            assert sorted(seen) == list(range(1, 1024 * world + 1))
            assert per_rank == [1024] * world
            if unit > 1 and steps <= 1024 // unit:      # whole launches of `unit` samplings per GPU: no step ends in a short launch
                assert all(n % unit == 0 for n in sizes) and max(sizes) - min(sizes) <= unit, (world, steps, sizes)
            else:
                assert max(sizes) - min(sizes) <= 1


def test_two_ranks_equal_one(tmp_path, emu, ha):
    world = 2
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    sc = ha.Scene("cornell_mini")
    e = emu.EmuScene(sc.desc_ptr)
    ref, _ = e.render(W, H, 1, STEPS * SPS * world + 1, threads=0)
    # same samplings, different fp32 summation order (per-rank partial sums, then the all-reduce)
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    assert got.sum() > 0
