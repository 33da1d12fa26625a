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
/// hr_comm_info_t: what the communicator reports about itself (path: 0 none, 1 RCCL rank of hr_comm_init_rank, 2 RCCL group of one
/// process, 3 same-device sum — not RCCL)
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct HrCommInfo { pub path: i32, pub nranks: i32, pub rank: i32, pub device: i32,
                                                                    pub rccl_version: i32, pub _pad: i32, pub allreduces: u64 }
pub enum HrCtx {}

extern "C" {
    pub fn hr_last_error() -> *const c_char;
    /// must equal HR_ABI_VERSION below (checked by HipRenderer::new)
    pub fn hr_abi_version() -> c_int;
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
    pub fn hr_comm_info(ctx: *mut HrCtx, out: *mut HrCommInfo) -> c_int;
    /// the RCCL the library's collective runs on (path of the shared object; reused = 1: one that was already mapped, e.g. the host's own)
    pub fn hr_comm_library(path_out: *mut c_char, cap: usize, reused_out: *mut c_int) -> c_int;
    /// which: 0 = this context's own accumulator, 1 = the all-reduced total; per-channel f64 sums (the checksum of the exchange)
    pub fn hr_accumulator_sum(ctx: *mut HrCtx, which: c_int, out_rgb: *mut f64) -> c_int;
}

// ---- GENERATED by tools/gen_rust_layout.py from include/hanamaru_hip.h: do not edit ----
/// the ABI these mirrors were checked against (hr_abi_version() of the library must return it)
pub const HR_ABI_VERSION: i32 = 7;
const _: () = assert!(std::mem::size_of::<HrTexture>() == 32 && std::mem::align_of::<HrTexture>() == 8);   // hr_texture
const _: () = assert!(std::mem::size_of::<HrMaterial>() == 112 && std::mem::align_of::<HrMaterial>() == 8);   // hr_material
const _: () = assert!(std::mem::size_of::<HrImage>() == 16 && std::mem::align_of::<HrImage>() == 8);   // hr_image
const _: () = assert!(std::mem::size_of::<HrElement>() == 232 && std::mem::align_of::<HrElement>() == 8);   // hr_element
const _: () = assert!(std::mem::size_of::<HrCamera>() == 168 && std::mem::align_of::<HrCamera>() == 8);   // hr_camera
const _: () = assert!(std::mem::size_of::<HrSkybox>() == 48 && std::mem::align_of::<HrSkybox>() == 8);   // hr_skybox
const _: () = assert!(std::mem::size_of::<HrSceneDesc>() == 248 && std::mem::align_of::<HrSceneDesc>() == 8);   // hr_scene_desc
const _: () = assert!(std::mem::size_of::<HrCommInfo>() == 32 && std::mem::align_of::<HrCommInfo>() == 8);   // hr_comm_info_t
const _: () = assert!(std::mem::size_of::<Vector3>() == 24);   // hr_vec3
#[cfg(test)]
mod layout {
    use super::*;
    macro_rules! off { ($t:ty, $f:ident) => {{ let u = std::mem::MaybeUninit::<$t>::uninit(); let b = u.as_ptr();
        unsafe { (std::ptr::addr_of!((*b).$f) as usize) - (b as usize) } }} }
    #[test]
    fn field_offsets_match_the_c_header() {
        assert_eq!(off!(HrTexture, color), 0);
        assert_eq!(off!(HrTexture, image), 24);
        assert_eq!(off!(HrMaterial, surface), 0);
        assert_eq!(off!(HrMaterial, param), 8);
        assert_eq!(off!(HrMaterial, albedo), 16);
        assert_eq!(off!(HrMaterial, emission), 48);
        assert_eq!(off!(HrMaterial, roughness), 80);
        assert_eq!(off!(HrImage, rgba), 0);
        assert_eq!(off!(HrImage, width), 8);
        assert_eq!(off!(HrImage, height), 12);
        assert_eq!(off!(HrElement, kind), 0);
        assert_eq!(off!(HrElement, material), 8);
        assert_eq!(off!(HrElement, center), 120);
        assert_eq!(off!(HrElement, radius), 144);
        assert_eq!(off!(HrElement, aabb_min), 152);
        assert_eq!(off!(HrElement, aabb_max), 176);
        assert_eq!(off!(HrElement, vertexes), 200);
        assert_eq!(off!(HrElement, num_vertexes), 208);
        assert_eq!(off!(HrElement, faces), 216);
        assert_eq!(off!(HrElement, num_faces), 224);
        assert_eq!(off!(HrCamera, eye), 0);
        assert_eq!(off!(HrCamera, right), 24);
        assert_eq!(off!(HrCamera, up), 48);
        assert_eq!(off!(HrCamera, forward), 72);
        assert_eq!(off!(HrCamera, plane_half_right), 96);
        assert_eq!(off!(HrCamera, plane_half_up), 120);
        assert_eq!(off!(HrCamera, lens_radius), 144);
        assert_eq!(off!(HrCamera, focus_distance), 152);
        assert_eq!(off!(HrCamera, lens_shape), 160);
        assert_eq!(off!(HrSkybox, face_image), 0);
        assert_eq!(off!(HrSkybox, intensity), 24);
        assert_eq!(off!(HrSceneDesc, elements), 0);
        assert_eq!(off!(HrSceneDesc, num_elements), 8);
        assert_eq!(off!(HrSceneDesc, images), 16);
        assert_eq!(off!(HrSceneDesc, num_images), 24);
        assert_eq!(off!(HrSceneDesc, skybox), 32);
        assert_eq!(off!(HrSceneDesc, camera), 80);
        assert_eq!(off!(HrCommInfo, path), 0);
        assert_eq!(off!(HrCommInfo, nranks), 4);
        assert_eq!(off!(HrCommInfo, rank), 8);
        assert_eq!(off!(HrCommInfo, device), 12);
        assert_eq!(off!(HrCommInfo, rccl_version), 16);
        assert_eq!(off!(HrCommInfo, allreduces), 24);
    }
}
// ---- END GENERATED ----
