"""ISAAC-64 (rand 0.4.3 StdRng) — pins the oracle's, the host layer's and the kernels' generator.

The crate is a Cargo.lock dependency that is not vendored under /root/reference; the known-answer vectors
are the ones rand ships in its own test-suite."""
import ctypes as C
import os

import numpy as np

KAT1_SEED = [1, 23, 456, 7890, 12345]
KAT1 = [547121783600835980, 14377643087320773276, 17351601304698403469, 1238879483818134882, 11952566807690396487,
        13970131091560099343, 4469761996653280935, 15552757044682284409, 6860251611068737823, 13722198873481261842]
KAT2_SEED = [12345, 67890, 54321, 9876]
KAT2 = [18143823860592706164, 8491801882678285927, 2699425367717515619, 17196852593171130876, 2606123525235546165,
        15790932315217671084, 596345674630742204, 9947027391921273664, 11788097613744130851, 10391409374914919106]


def test_oracle_known_answers(orc):
    assert orc.isaac64(KAT1_SEED, 10).tolist() == KAT1
    assert orc.isaac64(KAT2_SEED, 10, skip=10000).tolist() == KAT2


def test_host_layer_known_answers(ha):
    L = ha.host_lib()
    for seed, skip, exp in [(KAT1_SEED, 0, KAT1), (KAT2_SEED, 10000, KAT2)]:
        s = (C.c_uint64 * len(seed))(*seed)
        out = (C.c_uint64 * 10)()
        assert L.hh_debug_isaac64(s, len(seed), skip, out, 10) == 0
        assert list(out) == exp


def test_second_round_is_reached(orc):
    # more than 256 outputs forces a second isaac64() round (cnt wraps): compare 300 sequential vs skip
    a = orc.isaac64(KAT2_SEED, 300)
    b = orc.isaac64(KAT2_SEED, 44, skip=256)
    assert np.array_equal(a[256:], b)


def test_path_seed_probe_vector(orc):
    # seed [8700304, 1, 223781, 501148] = pixel (0,0) sub (0,0) of 480x270, sampling 1
    d = orc.path_draws(480, 270, 0, 0, 0, 0, 1, 4)
    assert np.array_equal(d, orc.isaac64([8700304, 1, 223781, 501148], 4))
    f = [orc.u64_to_f64(v) for v in d]
    assert f == [0.2285200597432051, 0.5250368802542618, 0.6681100348683542, 0.41564599708796934]


def test_seed_words_saturate_on_strips(orc, emu):
    """renderer.rs:165-166: `((4.0 + nc) * k) as usize`.  nc is divided by min(w, h) (renderer.rs:53-54), so on an image more than four
    times as wide as high (or as high as wide) 4 + nc goes negative on the far side — and Rust's float-to-integer `as` saturates: the
    seed word is 0 there (a C cast would be undefined: it wraps on x86 and clamps on the GPU).  Oracle and kernel code both follow the
    Rust rule, explicitly."""
    # 64x1: frag.x = 0 .. 63, nc.x = (x - 0.5) * 2 - 64 (sub-sample 0) -> negative 4 + nc.x for x <= 30
    for x, s_word in [(0, 0), (30, 0), (31, int((4.0 + ((31 - 0.5) * 2 - 64)) * 100870.0)), (63, int((4.0 + ((63 - 0.5) * 2 - 64)) * 100870.0))]:
        ncy = ((1 - 0.5) * 2 - 1) / 1.0
        t_word = int((4.0 + ncy) * 100304.0)
        ref = orc.isaac64([8700304, 9, s_word, t_word], 64)
        assert np.array_equal(orc.path_draws(64, 1, x, 0, 0, 0, 9, 64), ref), x
        assert np.array_equal(emu.raw_draws(64, 1, x, 0, 0, 9, 64), ref), x
    # 1x64: frag.y = 64 - y; the bottom rows have 4 + nc.y < 0
    for y, neg in [(0, False), (33, False), (34, True), (63, True)]:
        ncy = ((64 - y - 0.5) * 2 - 64) / 1.0
        t_word = 0 if neg else int((4.0 + ncy) * 100304.0)
        assert (4.0 + ncy < 0) == neg
        s_word = int((4.0 + ((0 - 0.5) * 2 - 1) / 1.0) * 100870.0)
        ref = orc.isaac64([8700304, 3, s_word, t_word], 64)
        assert np.array_equal(orc.path_draws(1, 64, 0, y, 0, 0, 3, 64), ref), y
        assert np.array_equal(emu.raw_draws(1, 64, 0, y, 0, 3, 64), ref), y


def test_u64_to_f64_range(orc):
    assert orc.u64_to_f64(0) == 0.0
    assert orc.u64_to_f64(2**64 - 1) == 1.0 - 2.0**-52
    assert orc.u64_to_f64(1 << 52) == 0.0          # only the low 52 bits are used


def test_golden_path_draws(orc):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "path_draws.npz"))
    for case, exp in zip(g["cases"], g["draws"]):
        assert np.array_equal(orc.path_draws(*[int(v) for v in case], 8), exp)


def test_kernel_generator_core_matches_oracle(emu, orc):
    """The seed kernel's per-lane code (isaac_core.h, compiled for the host) against the oracle, bit-exact,
    over the whole stored window of 64 outputs."""
    rng = np.random.default_rng(3)
    for w, h in [(480, 270), (1920, 1080), (7, 5)]:
        for _ in range(12):
            x, y, sub, s = int(rng.integers(w)), int(rng.integers(h)), int(rng.integers(4)), int(rng.integers(1, 5000))
            ref = orc.path_draws(w, h, x, y, sub & 1, sub >> 1, s, 64)
            assert np.array_equal(emu.raw_draws(w, h, x, y, sub, s, 64), ref)
            # split form: scratch-free init (pass 1 regenerated next to pass 2) + round on the loaded state
            assert np.array_equal(emu.raw_draws_split(w, h, x, y, sub, s, 64), ref)
            # producer / consumer form: the producer ships blocks >= HEAD + the pass-1 end state, the consumer redoes the rest
            for head in (0, 8, 16, 24, 32):
                assert np.array_equal(emu.raw_draws_pc(w, h, x, y, sub, s, head, 64), ref)
            # three-run form (seed_seg_kernel): the sweep's registers at blocks 0 / 11 / 21 computed ahead, three runs of 11 blocks
            assert np.array_equal(emu.raw_draws_seg(w, h, x, y, sub, s, 64), ref)


def test_kernel_lens_rejection_matches_oracle(emu, orc):
    """20 fp32 draws per path after the lens rejection loop (camera.rs:66-81) == the oracle's f64 draws rounded once."""
    rng = np.random.default_rng(4)
    deep = 0
    for _ in range(400):
        w, h = 320, 200
        x, y, sub, s = int(rng.integers(w)), int(rng.integers(h)), int(rng.integers(4)), int(rng.integers(1, 99))
        got, ok = emu.path_draws(w, h, x, y, sub, s, 1)
        assert ok
        f = [orc.u64_to_f64(v) for v in orc.path_draws(w, h, x, y, sub & 1, sub >> 1, s, 64)]
        j = 0
        while not ((2 * f[2 * j] - 1) ** 2 + (2 * f[2 * j + 1] - 1) ** 2 < 1.0):
            j += 1
        deep = max(deep, j)
        exp = np.asarray([2 * f[2 * j] - 1, 2 * f[2 * j + 1] - 1] + f[2 * j + 2:2 * j + 20], dtype=np.float64).astype(np.float32)
        assert np.array_equal(got, exp)
        # the record's twin (precise shading): fp32 draw + residual = the reference's f64 draw, to 2^-49
        lo, ok = emu.path_draw_residuals(w, h, x, y, sub, s, 1)
        exact = np.asarray(f[2 * j + 2:2 * j + 20], dtype=np.float64)
        assert ok and np.abs(got[2:].astype(np.float64) + lo[2:].astype(np.float64) - exact).max() <= 2.0 ** -49
        assert np.array_equal(lo[2:], (exact - got[2:].astype(np.float64)).astype(np.float32))
    assert deep >= 2                       # the sample exercised repeated rejections
    # paths that reject the first LENS_FAST = 5 attempts leave the 28-draw hand-off record and go through the fix-up
    # path (record_from_window, isaac_core.h): probability 4.6e-4 each, so search for some
    fixed = 0
    for i in range(30000):
        w, h = 640, 360
        x, y, sub, s = i % w, (i // w) % h, i % 4, 7 + i // (w * h)
        f = [orc.u64_to_f64(v) for v in orc.path_draws(w, h, x, y, sub & 1, sub >> 1, s, 64)]
        j = 0
        while not ((2 * f[2 * j] - 1) ** 2 + (2 * f[2 * j + 1] - 1) ** 2 < 1.0):
            j += 1
        if j < 5:
            continue
        fixed += 1
        got, ok = emu.path_draws(w, h, x, y, sub, s, 1)
        exp = np.asarray([2 * f[2 * j] - 1, 2 * f[2 * j + 1] - 1] + f[2 * j + 2:2 * j + 20], dtype=np.float64).astype(np.float32)
        assert ok and np.array_equal(got, exp)
        lo, ok = emu.path_draw_residuals(w, h, x, y, sub, s, 1)      # the fix-up path writes the twin too, rebased like the record
        exact = np.asarray(f[2 * j + 2:2 * j + 20], dtype=np.float64)
        assert ok and np.array_equal(lo[2:], (exact - got[2:].astype(np.float64)).astype(np.float32))
    assert fixed >= 3
    sq, ok = emu.path_draws(64, 64, 1, 2, 3, 4, 0)   # square lens: first pair always accepted
    f = [orc.u64_to_f64(v) for v in orc.path_draws(64, 64, 1, 2, 1, 1, 4, 4)]
    assert ok and sq[0] == np.float32(2 * f[0] - 1) and sq[2] == np.float32(f[2])
