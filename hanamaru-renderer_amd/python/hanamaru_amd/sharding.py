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
