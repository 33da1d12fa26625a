/*
 * hanamaru_hip_debug.h — the part of libhanamaru_hip.so that is NOT the product ABI: measurement knobs that change the kernels' schedule
 * (one of them, "debug_skip", produces a garbage image on purpose) and the unit-level entry points the parity tests and the profiling
 * tools call.  Same library, same hr_ctx; kept in a header of its own so that a product host cannot reach them by including
 * hanamaru_hip.h (round 5; until ABI 5 they were declared there).  Users: tests/, tools/, bench.py's --debug flags.
 */
#ifndef HANAMARU_HIP_DEBUG_H
#define HANAMARU_HIP_DEBUG_H

#include "hanamaru_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement / experiment knobs, kept out of hr_set_option so that a host cannot change the kernels' schedule — or produce a
 * garbage image — by a key string meant for a product option: "adv_den" / "leaf_den" (trace-kernel phase thresholds), "min_waves"
 * (4..6, occupancy variant of the trace kernel; only with quant_nodes = 1), "kchunk", "node_unroll" (1 | 2), "trace_wgs",
 * "seed_mode" (2 = three-run seed kernel, default; 3 = its phase-shifted four-run form and 4 = its five-wave four-run form, both
 * slower, kept as measured experiments; 1 = producer / consumer kernel with a
 * ring of generator words; 0 = fused),
 * "seed_split" (seed_mode 1), "seed_prio" / "init_prio" (s_setprio of the seed kernel's consumer / producer waves), "seed_prof"
 * (phase timing build of the seed kernel -> hr_stats.seed_phase_cycles; seed_mode 3: 1 | 2 | 3 = consumer 0, consumer 1, producer 0), "ploc_top" (bvh_builder 2: clusters the bottom-up merges
 * leave for the top-down build over them; 1 = merge to the root; takes effect at the next hr_upload_scene), "debug_skip" (bit mask that drops parts of the pipeline
 * for timing experiments: THE IMAGE IS GARBAGE), "nee_cull" (mask of pt_core.h nee_setup's shortcuts in force: 1 = far side of the emitter, 2 = GGX below the
 * horizon; bit 2 is reserved — a third shortcut was measured and dropped, DESIGN.md §4.2 —; default 7; 0 = trace every NEE shadow ray — the bit-identical A/B of the shortcuts). */
int hr_set_debug_option(hr_ctx *ctx, const char *key, double value);

/* ---- unit-level entry points used by the parity tests (same kernels' device functions) ---- */

/* Raw ISAAC-64 outputs of the per-path generators exactly as the seed kernel stores them:
 * for path p (= pixel-major, sub-sample minor: ((y*W + x)*4 + sy*2 + sx)) out[p*window + k] =
 * k-th next_u64() of StdRng::from_seed([8700304, sampling, s, t]) (renderer.rs:165-168). */
int hr_debug_draws(hr_ctx *ctx, uint32_t sampling, uint32_t first_path, uint32_t num_paths,
                   uint32_t window, uint64_t *host_out);

/* The DRAWS_PER_PATH (=20) fp32 draws the seed kernel hands to the trace kernel for every path of one
 * sampling: out[((y*W + x)*4 + sub)*20 + d]; d=0,1 = accepted lens sample (2u-1, 2v-1) after the rejection
 * loop of camera.rs:66-81, d=2.. = the (f64,f64) pairs of renderer.rs:175 in order. */
int hr_debug_path_draws(hr_ctx *ctx, uint32_t sampling, float *host_out);
/* The same slots of the records' twin (precise shading in force; option "draw_residuals"): what rounding each of those draws to fp32 took away —
 * f64 draw d of the path = (double)draw d + (double)residual d, to 2^-49; d = 0, 1 belong to the RAW lens draws u, v (not to 2u-1, 2v-1).
 * HR_ERR_UNSUPPORTED when the launch would carry no residuals. */
int hr_debug_path_draw_residuals(hr_ctx *ctx, uint32_t sampling, float *host_out);

/* Per-path accounting of ONE sampling through the production pipeline (seed kernel + the render kernel's LOG instantiation: the same
 * traversal and the same path state machine as hr_render; the accumulator is not touched).  out: W*H*4 records of eight 32-bit words,
 * record ((y*W + x)*4 + sy*2 + sx) = { radiance r, g, b (float bits) of calc_pixel (renderer.rs:163-203), scene.intersect calls of the path
 * (main + shadow rays), event log bytes 0-3, 4-7, 8 (one byte per iteration of renderer.rs:174, see pt_core.h PathLog: miss / surface type
 * hit / sample returned None, reflected or transmitted, which emitters' shadow rays were visible), hash of the element indices hit }.
 * The oracle keeps the same log (orc_path_log): tests/test_gpu_parity.py compares path by path. */
int hr_debug_path_log(hr_ctx *ctx, uint32_t sampling, uint32_t *host_out);

/* "draw_residuals" (default 1): with precise shading in force the default seed kernel (seed_mode 2) also writes what rounding each draw to fp32
 * took away into the records' twin (device_scene.h RenderParams::rec_lo_off) and shading computes with fp32 draw + residual = the reference's
 * f64 draw; 0 = the fp32 draws alone (the A/B: profiles/r06_exact_draws_ablation.txt).  Other seed kernels (seed_mode != 2) never write residuals. */
/* The split pipeline (debug option "trace_mode" 1: the loop body of renderer.rs:174-200 cut at scene.intersect into a traversal kernel and a
 * shading kernel per path iteration) kernel by kernel: ONE launch of num_k samplings from `sampling`, alone on the chip, an event between
 * every two kernels.  ms_out[21]: [0] = camera rays, [2 s - 1] / [2 s] = traversal / shading kernel of step s = 1 .. 10;
 * counts_out[22]: [2 s] / [2 s + 1] = rays / live paths of step s.  The accumulator is not touched. */
int hr_debug_wf_profile(hr_ctx *ctx, uint32_t sampling, uint32_t num_k, double *ms_out, uint32_t *counts_out);

/* Closest-hit query for n rays (scene.rs:385-401 minus the material fetch).
 * rays: n * 6 floats (origin, direction).  out per ray: 8 floats
 * { hit(0/1), distance, pos.x, pos.y, pos.z, n.x, n.y, n.z }, plus element index in out_element. */
int hr_debug_intersect(hr_ctx *ctx, uint32_t n, const float *rays, float *out, int32_t *out_element);

/* The same query through the PRODUCTION traversal of hr_render (scalar walk above: one node + its leaf per step on the 32-byte
 * records): 64 rays per wave through the render kernel's box / leaf phases on the record format it walks for this scene, with
 * parked leaves and closest-hit culling.  shadow_len == NULL or shadow_len[i] <= 0: closest hit (bvh.rs:213-290 + scene.rs:385-401).
 * shadow_len[i] > 0: ray i is a shadow ray towards a light sample at that distance (renderer.rs:276-282) — the search is limited to
 * the sample distance + 0.03 and stops at the first hit more than 0.02 in front of the sample, exactly as in the render kernel;
 * the visibility verdict of renderer.rs:280 is then  hit && (distance - shadow_len)^2 < 4e-4. */
int hr_debug_trace(hr_ctx *ctx, uint32_t n, const float *rays, const float *shadow_len, float *out, int32_t *out_element);

#ifdef __cplusplus
}
#endif
#endif
