// UNTESTED (no Rust toolchain in the build environment).  Mirror of include/hanamaru_hip.h.
#![allow(dead_code)]
use std::os::raw::{c_char, c_int, c_void};
use vector::Vector3; // #[repr(C)] {x, y, z: f64} (vector.rs:6-12) == hr_vec3

pub const HR_SPHERE: i32 = 0;
pub const HR_CUBOID: i32 = 1;
pub const HR_MESH: i32 = 2;

#[repr(C)] #[derive(Clone, Copy)] pub struct HrTexture { pub color: Vector3, pub image: i32, pub _pad: i32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct HrMaterial { pub surface: i32, pub _pad: i32, pub param: f64,
                                                           pub albedo: HrTexture, pub emission: HrTexture, pub roughness: HrTexture }
#[repr(C)] #[derive(Clone, Copy)] pub struct HrImage { pub rgba: *const u8, pub width: u32, pub height: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct HrElement { pub kind: i32, pub _pad: i32, pub material: HrMaterial,
                                                          pub center: Vector3, pub radius: f64,
                                                          pub aabb_min: Vector3, pub aabb_max: Vector3,
                                                          pub vertexes: *const Vector3, pub num_vertexes: u64,
                                                          pub faces: *const u64, pub num_faces: u64 }
#[repr(C)] #[derive(Clone, Copy)] pub struct HrCamera { pub eye: Vector3, pub right: Vector3, pub up: Vector3, pub forward: Vector3,
                                                         pub plane_half_right: Vector3, pub plane_half_up: Vector3,
                                                         pub lens_radius: f64, pub focus_distance: f64, pub lens_shape: i32, pub _pad: i32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct HrSkybox { pub face_image: [i32; 6], pub intensity: Vector3 }
#[repr(C)] pub struct HrSceneDesc { pub elements: *const HrElement, pub num_elements: u32,
                                    pub images: *const HrImage, pub num_images: u32,
                                    pub skybox: HrSkybox, pub camera: HrCamera }
pub enum HrCtx {}

extern "C" {
    pub fn hr_last_error() -> *const c_char;
    pub fn hr_create(device_id: c_int, out: *mut *mut HrCtx) -> c_int;
    pub fn hr_destroy(ctx: *mut HrCtx) -> c_int;
    pub fn hr_upload_scene(ctx: *mut HrCtx, scene: *const HrSceneDesc) -> c_int;
    pub fn hr_set_resolution(ctx: *mut HrCtx, width: u32, height: u32) -> c_int;
    pub fn hr_clear(ctx: *mut HrCtx) -> c_int;
    pub fn hr_render(ctx: *mut HrCtx, sampling_begin: u32, sampling_end: u32, stride: u32) -> c_int;
    pub fn hr_render_debug(ctx: *mut HrCtx, mode: c_int) -> c_int;
    pub fn hr_synchronize(ctx: *mut HrCtx) -> c_int;
    /// marker behind everything enqueued so far / wait for it while later work keeps running
    pub fn hr_mark(ctx: *mut HrCtx, ticket: *mut u64) -> c_int;
    pub fn hr_wait(ctx: *mut HrCtx, ticket: u64) -> c_int;
    pub fn hr_read_accumulator(ctx: *mut HrCtx, host_rgb: *mut f32) -> c_int;
    pub fn hr_write_accumulator(ctx: *mut HrCtx, host_rgb: *const f32) -> c_int;
    pub fn hr_resolve(ctx: *mut HrCtx, samplings_done: u32, host_rgb8: *mut u8) -> c_int;
    pub fn hr_bind_accumulator(ctx: *mut HrCtx, device_rgb: *mut f32) -> c_int;
    pub fn hr_accumulator_device_ptr(ctx: *mut HrCtx) -> *mut c_void;
    /// e.g. ("bvh_builder", 1.0) = build the BVH on the GPU, ("batch", 4.0) = samplings per launch; see hanamaru_hip.h.  Every key this
    /// call accepts leaves the image as the reference computes it, except the documented opt-in "russian_roulette" (off by default).
    /// (The library's measurement knobs sit behind hr_set_debug_option, deliberately not bound here.)
    pub fn hr_set_option(ctx: *mut HrCtx, key: *const c_char, value: f64) -> c_int;
    // multi-GPU: one ncclAllReduce of the accumulators, issued by the library (include/hanamaru_hip.h; RCCL is dlopen'ed on first use)
    pub fn hr_comm_get_unique_id(id_out: *mut u8 /* HR_COMM_ID_BYTES = 128 */) -> c_int;
    pub fn hr_comm_init_rank(ctx: *mut HrCtx, id: *const u8, world_size: c_int, rank: c_int) -> c_int;
    pub fn hr_comm_init_local(ctxs: *mut *mut HrCtx, n: c_int) -> c_int;
    pub fn hr_allreduce_accumulator(ctx: *mut HrCtx) -> c_int;
    pub fn hr_allreduce_accumulators(ctxs: *mut *mut HrCtx, n: c_int) -> c_int;
    pub fn hr_comm_destroy(ctx: *mut HrCtx) -> c_int;
}
