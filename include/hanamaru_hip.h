/*
 * hanamaru_hip.h — C ABI of the MI355X back end for hanamaru-renderer's render loop.
 *
 * What this replaces in the reference (all citations into /root/reference/src):
 *   - `Renderer::render`  (renderer.rs:25-46): the per-sampling, per-pixel, 2x2 sub-sample loop that
 *     accumulates `calc_pixel` into `accumulation_buf`          -> hr_render()
 *   - `PathTracingRenderer::calc_pixel` + `next_event_estimation` (renderer.rs:163-203, 269-296)
 *     incl. everything below it (scene.rs / bvh.rs / material.rs / texture.rs / camera.rs:66-96)
 *                                                               -> the HIP kernels behind hr_render()
 *   - `Renderer::update_imgbuf` (renderer.rs:64-90): scale, tonemap.rs Reinhard, gamma, filter.rs
 *     bilateral, color.rs quantise                              -> hr_resolve()
 *
 * The reference has no FFI; the seam is the `Renderer` trait (renderer.rs:20-25).  A Rust host keeps
 * its Scene/Camera/Material builders, loader and PNG writer, fills an `hr_scene_desc` with pointers
 * into its own `Vec<Vector3>` (Vector3 is #[repr(C)] {x,y,z: f64}, vector.rs:6-12) and calls these
 * entry points from `Renderer::render`.  See INTEGRATION.md for the Rust `extern "C"` block.
 *
 * Conventions: every function returns 0 (HR_OK) or a negative hr_status; hr_last_error() gives text.
 * No exceptions cross the boundary.  A context is bound to ONE GPU and is not thread-safe; different contexts may be driven by
 * different host threads at the same time (hr_last_error is per thread).  Multi-GPU is one context per GPU — one process each, or one
 * process driving them all —, sharded by sampling index (hr_render's `stride`).
 * Host buffers passed in are copied — the caller keeps ownership.  No torch / STL types in signatures.
 */
#ifndef HANAMARU_HIP_H
#define HANAMARU_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HR_ABI_VERSION 7

typedef enum hr_status {
    HR_OK = 0,
    HR_ERR_INVALID = -1,      /* bad argument / call order */
    HR_ERR_DEVICE = -2,       /* HIP runtime error (text in hr_last_error) */
    HR_ERR_NO_SCENE = -3,
    HR_ERR_NO_TARGET = -4,    /* hr_set_resolution not called */
    HR_ERR_RNG_WINDOW = -5,   /* a path consumed more ISAAC-64 outputs than the stored window */
    HR_ERR_UNSUPPORTED = -6
} hr_status;

/* vector.rs:6-12 — #[repr(C)] f64 triple */
typedef struct hr_vec3 { double x, y, z; } hr_vec3;

/* material.rs:9-15 SurfaceType.  `param` = f0 (GGX) or refractive_index (Refraction / GGXRefraction). */
enum { HR_DIFFUSE = 0, HR_SPECULAR = 1, HR_REFRACTION = 2, HR_GGX = 3, HR_GGX_REFRACTION = 4 };

/* texture.rs:72-75 Texture { image_texture: Option<ImageTexture>, color } */
typedef struct hr_texture {
    hr_vec3 color;            /* tint, multiplied with the bilinear sample (texture.rs:108-114) */
    int32_t image;            /* index into hr_scene_desc.images, or -1 = constant colour */
    int32_t _pad;
} hr_texture;

/* material.rs:17-23 Material */
typedef struct hr_material {
    int32_t surface;          /* HR_DIFFUSE ... */
    int32_t _pad;
    double param;
    hr_texture albedo, emission, roughness;
} hr_material;

/* decoded image, RGBA8, row 0 = top row (image crate order, texture.rs:59-63 flips y itself) */
typedef struct hr_image {
    const uint8_t *rgba;
    uint32_t width, height;
} hr_image;

/* scene.rs Intersectable implementors used by live scenes: Sphere :51, Cuboid :146, BvhMesh :236 */
enum { HR_SPHERE = 0, HR_CUBOID = 1, HR_MESH = 2 };

typedef struct hr_element {
    int32_t kind;
    int32_t _pad;
    hr_material material;
    hr_vec3 center; double radius;          /* HR_SPHERE  (scene.rs:51-55) */
    hr_vec3 aabb_min, aabb_max;             /* HR_CUBOID  (scene.rs:146-149, bvh.rs:8-11) */
    const hr_vec3 *vertexes;                /* HR_MESH: world-space (loader.rs:31) */
    uint64_t num_vertexes;
    const uint64_t *faces;                  /* 3 vertex indices per face (scene.rs:196-200, usize) */
    uint64_t num_faces;
} hr_element;

/* camera.rs:7-29 Camera (already built by Camera::new, camera.rs:45-64) */
typedef struct hr_camera {
    hr_vec3 eye, right, up, forward, plane_half_right, plane_half_up;
    double lens_radius, focus_distance;
    int32_t lens_shape;                     /* 0 = Square, 1 = Circle (camera.rs:31-36) */
    int32_t _pad;
} hr_camera;

/* scene.rs:268-276 Skybox: px, nx, py, ny, pz, nz */
typedef struct hr_skybox {
    int32_t face_image[6];
    hr_vec3 intensity;
} hr_skybox;

typedef struct hr_scene_desc {
    const hr_element *elements; uint32_t num_elements;   /* order = Scene.elements order (scene.rs:327-330) */
    const hr_image *images;     uint32_t num_images;
    hr_skybox skybox;
    hr_camera camera;
} hr_scene_desc;

/* Work / timing counters.  Counter fields are only filled when option "counters" = 1. */
typedef struct hr_stats {
    uint64_t paths;            /* calc_pixel-equivalents rendered since hr_clear */
    uint64_t rays;             /* scene.intersect-equivalents (primary + bounce + shadow) */
    uint64_t node_tests;       /* AABB tests performed by the traversal kernel */
    uint64_t tri_tests, sphere_tests, cuboid_tests;
    uint64_t rng_overflow;     /* paths that ran past the stored ISAAC window */
    double seed_kernel_ms;     /* sum of HIP-event durations of the seed kernel launches */
    double trace_kernel_ms;    /* ... of the path-trace megakernel launches */
    double post_kernel_ms;
    uint64_t seed_launches, trace_launches;
    uint64_t bvh_nodes, triangles, spheres, cuboids;   /* triangles: of the scene (early split clipping may store one as several references) */
    /* counters build only: wave-level phase statistics of the trace kernel (invocations, lanes served) */
    uint64_t shade_calls, shade_lanes, box_passes, box_lanes, leaf_calls, leaf_lanes, outer_iters;
    uint64_t phase_cycles[4];  /* counters build: wave-cycles in A shade, B refill, C box phase, C leaf phase */
    double bvh_build_ms;       /* device BVH build of the last hr_upload_scene (option bvh_builder = 1 | 2), else 0 */
    uint64_t seed_phase_cycles[8]; /* debug option seed_prof: consumer-wave cycles per phase of the seed kernel, [7] = groups */
    double debug_kernel_ms;    /* sum of HIP-event durations of the hr_render_debug launches (the traversal-only workload) */
    uint64_t debug_launches;
    /* the priority governor (option trace_boost): the level kernels start at now (0 = the seed kernel's producer waves first .. 4 = the
     * trace kernel's box and leaf phases first), launches it has judged since the last scene / resolution / option change, level changes */
    uint64_t governor_level, governor_decisions, governor_moves;
    /* counters build: NEE shadow rays the reference traces and discards, which the kernel knows to add nothing before it traces them
     * (sample on the emitter's far side; GGX with the emitter below the horizon) — not in `rays` */
    uint64_t shadow_culled;
    /* the governor's wave budget: how many of the trace kernel's persistent workgroups stay (0 = all of them) — fewer where the trace
     * kernel is the faster kernel of the pair and its surplus waves only slow the seed kernel beside it —, and how often it changed */
    uint64_t governor_budget, governor_budget_moves;
    uint64_t bvh_builder_used;  /* the builder the last hr_upload_scene used (0 host SAH, 1 device LBVH, 2 device PLOC): what option bvh_builder = -1 chose */
    uint64_t shading_in_force;  /* what option precise_shading means for the scene in place: 0 = fp32 shading (megakernel), 1 = precise shading in the
                                 * megakernel, 2 = precise shading in the split pipeline (same bits as 1), 3 = fp32 shading in the split pipeline (debug) */
} hr_stats;

typedef struct hr_ctx hr_ctx;

const char *hr_last_error(void);
int hr_abi_version(void);

int hr_create(int device_id, hr_ctx **out);
int hr_destroy(hr_ctx *ctx);

/* Scene: copies + converts to fp32 SoA, builds the device BVH, uploads.  HR_ERR_INVALID for a description the reference could not render
 * either (no elements, a mesh without data, an image index out of range, non-finite geometry or camera — the reference panics in its BVH
 * build on a NaN): the scene uploaded before stays in place. */
int hr_upload_scene(hr_ctx *ctx, const hr_scene_desc *scene);

/* Output target (ImageBuffer dims, renderer.rs:26-28).  Allocates + zeroes the fp32 RGB accumulator. */
int hr_set_resolution(hr_ctx *ctx, uint32_t width, uint32_t height);
/* Optional: accumulate into caller-owned DEVICE memory (W*H*3 floats), e.g. a torch tensor that
 * torch.distributed all-reduces over RCCL.  Pass NULL to return to the internal buffer.
 * EXCLUSIVE: one context per buffer, and the caller must not touch the buffer on another stream between hr_render and the next
 * hr_synchronize / read — a launch's radiance is added with plain loads and stores (no atomics), in a fixed order (bit-reproducible
 * renders).  Binding a buffer another context of this process holds returns HR_ERR_INVALID, and so does a pointer that is not device memory of
 * this context's device, is not float-aligned, or has fewer than W*H*3 floats between it and the end of its allocation.  hr_set_resolution unbinds. */
int hr_bind_accumulator(hr_ctx *ctx, float *device_rgb);
void *hr_accumulator_device_ptr(hr_ctx *ctx);
/* Optional: run on a caller-owned hipStream_t (opaque).  NULL = the context's own stream. */
int hr_set_stream(hr_ctx *ctx, void *hip_stream);

int hr_clear(hr_ctx *ctx);                 /* zero accumulator + stats */

/* Accumulate samplings s = begin, begin+stride, ... (s < end) — 1-origin like renderer.rs:31-32.
 * Each sampling adds the 2x2 sub-sample sum of calc_pixel to every pixel (renderer.rs:33-38,48-60).
 * Asynchronous with respect to the host; hr_synchronize() or any read waits. */
int hr_render(hr_ctx *ctx, uint32_t sampling_begin, uint32_t sampling_end, uint32_t stride);
int hr_synchronize(hr_ctx *ctx);
/* hr_render only enqueues.  hr_mark records a marker behind everything enqueued so far; hr_wait blocks until that marker is
 * reached while later work keeps running (a host loop can keep one chunk of samplings in flight while it reports on the
 * previous one — the reference's report_progress cadence, renderer.rs:205-251, without draining the GPU).  hr_synchronize
 * waits for everything and reports kernel-side errors. */
int hr_mark(hr_ctx *ctx, uint64_t *ticket);
int hr_wait(hr_ctx *ctx, uint64_t ticket);

/* DebugRenderer (renderer.rs:101-146, max_sampling = 1): adds ONE sampling of the chosen visualiser to the
 * accumulator — pinhole rays, no RNG.  mode: 0 Shading, 1 Normal, 2 Depth, 3 FocalPlane (renderer.rs:102-107;
 * the reference's -d flag selects FocalPlane, main.rs:1280).  Resolve with samplings_done = 1.
 * The rays go through the render kernel's traversal (same records, same box / leaf phases): Depth mode doubles as the
 * traversal-only workload of bench.py (hr_stats.debug_kernel_ms; with option "counters" the node / primitive test counts). */
int hr_render_debug(hr_ctx *ctx, int mode);

int hr_read_accumulator(hr_ctx *ctx, float *host_rgb);        /* W*H*3, row-major, top row first */
int hr_write_accumulator(hr_ctx *ctx, const float *host_rgb); /* resume / post-chain tests */

/* renderer.rs:64-90: scale by 1/(samplings*4) -> Reinhard -> gamma -> bilateral 3x3 -> u8 RGB. */
int hr_resolve(hr_ctx *ctx, uint32_t samplings_done, uint8_t *host_rgb8);

/* ---- multi-GPU (one node, RCCL over xGMI) --------------------------------------------------------------------------
 * The loop being sharded is renderer.rs:32-43: samplings are independent and seeded by index, so rank r of N renders
 * s = begin + r, begin + r + N, ... (hr_render's stride) into its own accumulator.  Before the resolve of renderer.rs:64-90
 * the accumulators are summed with ONE ncclAllReduce (fp32, W*H*3 values: 24.9 MB at 1080p).  The sum goes to a separate
 * buffer: the rank's own accumulator is untouched, so rendering can continue after a progress image.  After the call
 * hr_resolve / hr_read_accumulator return the TOTAL, until the next hr_render / hr_render_debug / hr_clear /
 * hr_write_accumulator on that context.  The collective is enqueued on the context's stream behind its render work.
 *   one process per GPU:   rank 0: hr_comm_get_unique_id -> share the HR_COMM_ID_BYTES with the other ranks (any transport)
 *                          every rank: hr_comm_init_rank, ..., hr_allreduce_accumulator
 *   one process, N GPUs:   hr_comm_init_local(ctxs, N), ..., hr_allreduce_accumulators(ctxs, N)   (one RCCL group call)
 * RCCL is loaded on first use; without it these return HR_ERR_UNSUPPORTED (the library has no host-side sum; a host that must run
 * on such a box sums hr_read_accumulator results itself, as the hanamaru-hip CLI does).  RCCL wants one
 * rank per device: hr_comm_init_local over contexts that all share ONE device (a single-GPU box) sums them with a kernel on
 * that device instead; a mix of shared and distinct devices is rejected. */
#define HR_COMM_ID_BYTES 128
int hr_comm_get_unique_id(void *id_out);
int hr_comm_init_rank(hr_ctx *ctx, const void *id, int world_size, int rank);
int hr_comm_init_local(hr_ctx **ctxs, int n);
int hr_comm_destroy(hr_ctx *ctx);
int hr_allreduce_accumulator(hr_ctx *ctx);
int hr_allreduce_accumulators(hr_ctx **ctxs, int n);
void *hr_total_device_ptr(hr_ctx *ctx);   /* device pointer of the all-reduced accumulator, NULL when not valid */

/* What the context's communicator says about ITSELF (asked of RCCL at the time of the call: ncclCommCount, ncclCommUserRank,
 * ncclCommCuDevice, ncclGetVersion) — so that a host can print evidence that its all-reduce really ran over N ranks:
 *   path  HR_COMM_NONE            no communicator
 *         HR_COMM_RCCL_RANK       hr_comm_init_rank: this process is one rank of an RCCL communicator
 *         HR_COMM_RCCL_GROUP      hr_comm_init_local over distinct devices: one process, one RCCL communicator per device
 *         HR_COMM_SAME_DEVICE_SUM hr_comm_init_local over contexts that share one device: NOT RCCL, a sum kernel on that device
 *   nranks / rank / device: RCCL's answers (same-device sum: the group's size, this context's index, the shared device)
 *   rccl_version: ncclGetVersion's code (0 without RCCL);  allreduces: collectives this context has enqueued on that communicator */
enum { HR_COMM_NONE = 0, HR_COMM_RCCL_RANK = 1, HR_COMM_RCCL_GROUP = 2, HR_COMM_SAME_DEVICE_SUM = 3 };
typedef struct hr_comm_info_t {
    int32_t path, nranks, rank, device;
    int32_t rccl_version, _pad;
    uint64_t allreduces;
} hr_comm_info_t;
int hr_comm_info(hr_ctx *ctx, hr_comm_info_t *out);
/* Which RCCL the library runs its collective on: the path of the shared object ncclAllReduce was resolved from (NUL-terminated into
 * path_out, truncated to cap), *reused_out = 1 when an RCCL already mapped into the process was taken instead of loading another one —
 * the library never puts a second RCCL build beside the host's (PyTorch maps its own torch/lib/librccl.so).  Loads RCCL if nothing has
 * yet; HR_ERR_UNSUPPORTED (and the loader's message) when there is none. */
int hr_comm_library(char *path_out, size_t cap, int *reused_out);
/* Per-channel sum of an accumulator, in f64 on the device: which = 0 this context's own accumulator, 1 = the all-reduced total.
 * The checksum of the exchange: the ranks' own sums add up to the total's sum (to fp32 rounding of the all-reduce: ~1e-7 relative). */
int hr_accumulator_sum(hr_ctx *ctx, int which, double out_rgb[3]);

int hr_get_stats(hr_ctx *ctx, hr_stats *out);
/* Options that leave the image as the reference computes it (the summation order of the accumulator aside):
 *   "counters"      0 / 1: instrumented build of the trace kernel (fills the counter fields of hr_stats)
 *   "batch"         samplings per launch, 1..64; 0 = automatic (about 33 M paths per launch: 4 at 1080p, up to 64 for small images)
 *   "trace_boost"   -1 = the two kernels are balanced from their own time stamps, on the device, launch by launch (default): five
 *                   levels from "the seed kernel's producer waves above the trace kernel" (0) over "alternating" (1) and "equal" (2)
 *                   to "the trace kernel's box phase (3) and leaf phase (4) above the producer waves" — and, where the trace kernel is
 *                   the faster kernel of the pair by a margin, how many of its persistent workgroups stay (its surplus waves only slow
 *                   the seed kernel beside it); 0 .. 4 = fixed level, every workgroup kept
 *                   (hr_stats.governor_level / governor_budget say where it stands)
 *   "max_tail_gib"  cap of each seed -> trace hand-off buffer, 1..128 GiB (default 20)
 *   "rng_window"    fixed: 64
 *   "precise_shading"  the GEOMETRY of every bounce in the reference's own f64: hit distance again from the f64 ray and the f64 primitive, hit
 *                   point, normal, mirror / Snell / Fresnel (material.rs:154-199) and the sampled lobe directions, FROM THE REFERENCE'S f64 DRAWS
 *                   (the seed kernel hands over what rounding a draw to fp32 took away as well: the hand-off record doubles); roughness maps are
 *                   read at f64 texture coordinates; the ray is carried as fp32 + residual, the walk stays fp32.  Same estimator, closer to the
 *                   reference: the paths that take the reference's branches and still differ by more than 1e-3 — refraction chains through
 *                   faceted glass, bounces off small spheres, GGX lobes driven by a roughness map — fall from 90 - 990 per million to 0 - 6
 *                   (BASELINE config 2: none in 10^6 paths, worst path 4e-5).  Two implementations that render the same bits:
 *                   in the megakernel at 128 VGPRs (2 - 4 % slower on scenes without meshes) and in the split pipeline's shading kernel
 *                   (7 - 30 % on mesh scenes); the library takes the faster one for the scene.
 *                   -1 (default) = automatic: ON for scenes without triangle meshes (BASELINE config 2: small spheres are what multiplies an
 *                   fp32 ray's error, and there it costs little), OFF for the others; 0 = off; 1 = on.  hr_stats.shading_in_force says
 *                   what runs.  (1 excludes "russian_roulette"; -1 stands back when the roulette is on.)
 *   next hr_upload_scene:
 *   "bvh_builder"   -1 = by scene size (default): the host's binned-SAH build below 200,000 primitives (the best tree; one host thread,
 *                   < 1 s), the device PLOC build from there on (0.97 - 0.99 of that tree's quality; 4 x 10^6 triangles in 38 ms instead of
 *                   25 s); 0 = host build, 1 = LBVH, 2 = PLOC on the device — replaces bvh.rs:107-211 (hr_stats.bvh_builder_used)
 *   "max_leaf"      BVH leaf size, 1..15 (default 4)
 *   "split_ratio"   early split clipping of long thin triangles in the host builder: -1 = automatic (kept when it cuts the SAH
 *                   cost by more than 7 %, default), 0 = off, > 0 = always, with that box / triangle area ratio
 *   "quant_nodes"   1 = the trace kernel walks the 16-byte quantised node records (default), 0 = the 32-byte fp32 records of the
 *                   same tree (identical hits)
 * One option that does NOT preserve the image (off by default; every parity test runs with it off):
 *   "russian_roulette"  0 = off.  k in 2..9: from path iteration k on a path survives with probability q = min(1, max(reflectance))
 *                   and its reflectance is divided by q.  The reference has no Russian roulette (renderer.rs:174-200 runs every
 *                   path to the bounce limit); the estimator stays unbiased (the decisions come from a hash of the path's
 *                   indices, not from its ISAAC-64 stream, whose draws the reference estimator has all spoken for); its noise
 *                   changes, it traces ~10 % fewer rays. */
int hr_set_option(hr_ctx *ctx, const char *key, double value);
/* The measurement knobs (hr_set_debug_option) and the unit-level entry points of the parity tests (hr_debug_*) live in
 * hanamaru_hip_debug.h: same library, but a product host — the Rust shell of INTEGRATION.md, the hanamaru-hip CLI — includes and binds
 * this header only (tests/test_abi.py checks that the CLI binary imports no hr_debug_* symbol). */

#ifdef __cplusplus
}
#endif
#endif
