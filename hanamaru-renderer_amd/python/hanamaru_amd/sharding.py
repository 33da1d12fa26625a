"""Sampling-index sharding across GPUs: units (pixel, sub-sample, sampling) are independent
and seeded by index (renderer.rs:165-167), so rank r of `world` renders samplings s with (s-1) % world == r
and the fp32 accumulators are summed with one all-reduce.  Pure index arithmetic — shared by bench.py and the
gloo test."""


def step_range(step, samplings_per_rank, world, rank):
    """hr_render(begin, end, stride) arguments for `rank` in bench step `step` (0-origin).

    Step `step` covers the 1-origin sampling indices [step*S*world + 1, (step+1)*S*world]; every rank gets
    exactly S = samplings_per_rank of them."""
    base = step * samplings_per_rank * world + 1
    return base + rank, base + samplings_per_rank * world, world


def samplings_of(step, samplings_per_rank, world, rank):
    b, e, s = step_range(step, samplings_per_rank, world, rank)
    return list(range(b, e, s))


def strong_plan(total_samplings, steps, world):
    """bench.py --total-samplings S (strong scaling: BASELINE config 4 is S = 4096 over 8 GPUs): (samplings per step per rank, steps)
    such that `steps` steps of samplings_per_rank * world consecutive sampling indices cover 1..S — the last step is clipped to S."""
    sps = max(1, -(-total_samplings // (steps * world)))
    return sps, -(-total_samplings // (sps * world))


def strong_step_range(step, samplings_per_rank, world, rank, total_samplings):
    """step_range with the end clipped to the total: hr_render(begin, end, stride) arguments (an empty range when begin >= end)."""
    b, e, s = step_range(step, samplings_per_rank, world, rank)
    return b, min(e, total_samplings + 1), s


def headline_step_range(step, steps, per_gpu_total, world, rank, unit=1):
    """bench.py's default plan: exactly `per_gpu_total` samplings per GPU (BASELINE's 1,024) whatever --steps is.  Step i of K covers
    the per-GPU sampling counts [lo_i, lo_(i+1)) with lo_i = floor(i*U/K) * unit, U = per_gpu_total / unit — whole kernel launches (`unit`
    samplings per GPU, the library's launch size: 4 at 1920x1080), so that no step ends in a short launch; steps differ by at most one launch.
    With unit = 1, with a total that is no multiple of `unit`, or with more steps than launches: lo_i = floor(i*T/K), steps differ by at most
    one sampling.  The 1-origin sampling indices of step i are lo_i*world + 1 .. lo_(i+1)*world over all GPUs, of which rank r takes those
    with (s-1) % world == r.  hr_render(begin, end, stride) arguments; requires steps <= per_gpu_total."""
    if unit > 1 and per_gpu_total % unit == 0 and steps <= per_gpu_total // unit:
        units = per_gpu_total // unit
        lo, hi = (step * units) // steps * unit, ((step + 1) * units) // steps * unit
    else:
        lo, hi = (step * per_gpu_total) // steps, ((step + 1) * per_gpu_total) // steps
    return lo * world + 1 + rank, hi * world + 1, world
