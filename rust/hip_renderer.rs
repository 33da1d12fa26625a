// UNTESTED (no Rust toolchain in the build environment).
// `Renderer` implementation that replaces the rayon pixel loop (renderer.rs:32-43) by the HIP back end and keeps the
// reference's progress / time-limit policy (renderer.rs:205-251) on the host clock.
extern crate image;
extern crate time;

use std::ffi::CStr;
use std::ptr;
use image::{ImageBuffer, Rgb, GenericImage};

use camera::{Camera, LensShape};
use color::Color;
use config;
use hip_ffi::*;
use material::{Material, SurfaceType};
use renderer::Renderer;
use scene::{Intersectable, SceneTrait, BvhScene};
use texture::Texture;
use vector::{Vector2, Vector3};

/// What `Intersectable::describe()` (added to scene.rs, see README.md) hands back.
pub enum ElementRef<'a> {
    Sphere { center: &'a Vector3, radius: f64, material: &'a Material },
    Cuboid { min: &'a Vector3, max: &'a Vector3, material: &'a Material },
    Mesh { vertexes: &'a Vec<Vector3>, faces: Vec<[u64; 3]>, material: &'a Material },
}

const IN_FLIGHT: usize = 8; // samplings enqueued ahead of the one being reported (hr_render only enqueues: the GPU never drains between reports)

pub struct HipRenderer {
    ctx: *mut HrCtx,
    sampling: u32,
    time_limit_sec: f64,
    report_interval_sec: f64,
    begin: time::Tm,
    last_report_progress: time::Tm,
    last_report_image: time::Tm,
    report_image_counter: u32,
}
unsafe impl Sync for HipRenderer {} // the context is only touched from render(), on the calling thread

fn check(rc: i32) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(hr_last_error()) }.to_string_lossy().into_owned();
        panic!("hanamaru_hip error {}: {}", rc, msg); // the reference panics on every error too
    }
}

struct Flat { // keeps everything the hr_scene_desc points into alive during hr_upload_scene
    elements: Vec<HrElement>, faces: Vec<Vec<[u64; 3]>>, images: Vec<HrImage>, pixels: Vec<Vec<u8>>,
}
impl Flat {
    fn texture(&mut self, t: &Texture) -> HrTexture {
        let image = match t.image_texture {
            Some(ref it) => { // texture.rs:10-12: DynamicImage -> RGBA8, row 0 = top
                let rgba = it.image.to_rgba();
                let (w, h) = (rgba.width(), rgba.height());
                self.pixels.push(rgba.into_raw());
                self.images.push(HrImage { rgba: self.pixels.last().unwrap().as_ptr(), width: w, height: h });
                (self.images.len() - 1) as i32
            }
            None => -1,
        };
        HrTexture { color: t.color, image, _pad: 0 }
    }
    fn material(&mut self, m: &Material) -> HrMaterial {
        let (surface, param) = match m.surface { // material.rs:9-15
            SurfaceType::Diffuse => (0, 0.0),
            SurfaceType::Specular => (1, 0.0),
            SurfaceType::Refraction { refractive_index } => (2, refractive_index),
            SurfaceType::GGX { f0 } => (3, f0),
            SurfaceType::GGXRefraction { refractive_index } => (4, refractive_index),
        };
        HrMaterial { surface, _pad: 0, param, albedo: self.texture(&m.albedo), emission: self.texture(&m.emission),
                     roughness: self.texture(&m.roughness) }
    }
}

impl HipRenderer {
    fn secs(t: time::Tm, since: time::Tm) -> f64 { (t - since).num_milliseconds() as f64 * 0.001 }
    /// renderer.rs:222-251 once nothing is in flight: writes the image the rule asks for.  `why`: Some(..) = final image (the render ends).
    fn write_image(&mut self, why: Option<&str>, sampling: u32, used: f64, imgbuf: &mut ImageBuffer<Rgb<u8>, Vec<u8>>) {
        let path = format!("{:>03}.png", self.report_image_counter);
        match why {
            Some(w) => { println!("{}", w); println!("output final image: {}", path); println!("remain: {:.3} sec.", self.time_limit_sec - used); }
            None => println!("output progress image: {}", path),
        }
        check(unsafe { hr_synchronize(self.ctx) });
        check(unsafe { hr_resolve(self.ctx, sampling, imgbuf.as_mut_ptr()) }); // RGB8, row-major, top row first
        let _ = image::ImageRgb8(imgbuf.clone()).save(&path);
        if why.is_none() { self.report_image_counter += 1; }
    }
    pub fn new(sampling: u32, time_limit_sec: f64, report_interval_sec: f64) -> HipRenderer {
        // the #[repr(C)] mirrors of hip_ffi.rs were generated for this ABI (their sizes are compile-time assertions there)
        assert_eq!(unsafe { hr_abi_version() }, HR_ABI_VERSION, "libhanamaru_hip.so and hip_ffi.rs disagree about the ABI version");
        let mut ctx: *mut HrCtx = ptr::null_mut();
        check(unsafe { hr_create(0, &mut ctx) });
        let now = time::now();
        HipRenderer { ctx, sampling, time_limit_sec, report_interval_sec, begin: now, last_report_progress: now,
                      last_report_image: now, report_image_counter: 0 }
    }

    fn upload(&mut self, scene: &BvhScene, camera: &Camera) {
        let zero = Vector3::zero();
        let mut flat = Flat { elements: vec![], faces: vec![], images: vec![], pixels: vec![] };
        for e in &scene.scene.elements {
            let mut h = HrElement { kind: 0, _pad: 0, material: flat.material(e.material()), center: zero, radius: 0.0,
                                    aabb_min: zero, aabb_max: zero, vertexes: ptr::null(), num_vertexes: 0, faces: ptr::null(), num_faces: 0 };
            match e.describe() {
                ElementRef::Sphere { center, radius, .. } => { h.kind = HR_SPHERE; h.center = *center; h.radius = radius; }
                ElementRef::Cuboid { min, max, .. } => { h.kind = HR_CUBOID; h.aabb_min = *min; h.aabb_max = *max; }
                ElementRef::Mesh { vertexes, faces, .. } => {
                    h.kind = HR_MESH;
                    h.vertexes = vertexes.as_ptr(); h.num_vertexes = vertexes.len() as u64;
                    flat.faces.push(faces);
                    let f = flat.faces.last().unwrap();
                    h.faces = f.as_ptr() as *const u64; h.num_faces = f.len() as u64;
                }
            }
            flat.elements.push(h);
        }
        let sky = &scene.scene.skybox; // scene.rs:268-276: px, nx, py, ny, pz, nz
        let mut face_image = [0i32; 6];
        for (i, t) in [&sky.px_texture, &sky.nx_texture, &sky.py_texture, &sky.ny_texture, &sky.pz_texture, &sky.nz_texture].iter().enumerate() {
            let rgba = t.image.to_rgba();
            let (w, h) = (rgba.width(), rgba.height());
            flat.pixels.push(rgba.into_raw());
            flat.images.push(HrImage { rgba: flat.pixels.last().unwrap().as_ptr(), width: w, height: h });
            face_image[i] = (flat.images.len() - 1) as i32;
        }
        let desc = HrSceneDesc {
            elements: flat.elements.as_ptr(), num_elements: flat.elements.len() as u32,
            images: flat.images.as_ptr(), num_images: flat.images.len() as u32,
            skybox: HrSkybox { face_image, intensity: sky.intensity },
            camera: HrCamera { eye: camera.eye, right: camera.right, up: camera.up, forward: camera.forward,
                               plane_half_right: camera.plane_half_right, plane_half_up: camera.plane_half_up,
                               lens_radius: camera.lens_radius, focus_distance: camera.focus_distance,
                               lens_shape: match camera.lens_shape { LensShape::Square => 0, LensShape::Circle => 1 }, _pad: 0 },
        };
        check(unsafe { hr_upload_scene(self.ctx, &desc) }); // copies everything: `flat` may be dropped afterwards
    }
}

impl Renderer for HipRenderer {
    fn max_sampling(&self) -> u32 { self.sampling }

    // never called: the per-path work runs on the GPU
    fn calc_pixel(&self, _: &SceneTrait, _: &Camera, _: &Vec<&Box<Intersectable>>, _: &Vector2, _: u32) -> Color { unreachable!() }

    fn render(&mut self, scene: &SceneTrait, camera: &Camera, imgbuf: &mut ImageBuffer<Rgb<u8>, Vec<u8>>) -> u32 {
        // main.rs:1216 always passes a BvhScene; the trait object needs `fn as_bvh_scene(&self) -> &BvhScene` (one line in scene.rs)
        self.upload(scene.as_bvh_scene(), camera);
        check(unsafe { hr_set_resolution(self.ctx, imgbuf.width(), imgbuf.height()) });
        // renderer.rs:32-43 with report_progress (renderer.rs:205-251) — the loop of hanamaru-hip's cli_main.cpp (compiled and tested there),
        // statement for statement.  Every sampling gets its own "rendering:" line; the GPU is fed LAUNCHES of `lrep` samplings (what fills the
        // chip: 4 at 1920x1080) and up to IN_FLIGHT launches are enqueued ahead (hr_mark behind each, hr_wait for the oldest).  A launch's lines
        // are printed when it is done, its wall time split evenly over them.  The time-limit rule (renderer.rs:222-231) is asked when samplings
        // are ISSUED, for the moment they would finish: n are issued only if used + 1.1 x last x (in flight + n) <= limit — with one sampling
        // in flight (report_interval_sec <= 0: an image is due after every report) the reference's rule to the letter.  A progress image is
        // written at a launch boundary, after the launches in flight have been reported: it holds exactly the samplings of the line before it.
        let per_sampling = ((imgbuf.width() as u64 + 3) / 4) * ((imgbuf.height() as u64 + 3) / 4) * 64;
        let lrep: u32 = if self.report_interval_sec <= 0.0 { 1 } else { ((33_177_600 + per_sampling - 1) / per_sampling).max(4).min(64) as u32 };
        let depth = if self.report_interval_sec <= 0.0 { 1 } else { IN_FLIGHT };
        if lrep > 1 { println!("launches of {} reports ({} samplings): a launch's time is split evenly over its reports' lines.", lrep, lrep); }
        struct Launch { begin: u32, end: u32, issued: time::Tm, ticket: u64 }
        let mut q: std::collections::VecDeque<Launch> = std::collections::VecDeque::new();
        let (mut next, mut done, mut in_flight) = (1u32, 0u32, 0u32);
        let (mut measured, mut last, mut used) = (false, 0.0f64, 0.0f64);
        let spp = config::SUPERSAMPLING * config::SUPERSAMPLING;
        loop {
            // issue: samplings left, room in the pipeline, and the time-limit rule asked for the moment they would finish
            loop {
                if next > self.sampling || q.len() >= depth { break; }
                let mut n = lrep.min(self.sampling + 1 - next);
                if measured {
                    let room = self.time_limit_sec - Self::secs(time::now(), self.begin);
                    let fit = if last > 0.0 { room / (1.1 * last) - in_flight as f64 } else if room >= 0.0 { n as f64 } else { 0.0 };
                    if fit < 1.0 { break; }
                    if fit < n as f64 { n = fit as u32; }
                }
                let mut t = 0u64;
                check(unsafe { hr_render(self.ctx, next, next + n, 1) });
                check(unsafe { hr_mark(self.ctx, &mut t) });
                q.push_back(Launch { begin: next, end: next + n, issued: time::now(), ticket: t });
                next += n;
                in_flight += n;
            }
            // nothing in flight and nothing may follow: the render ends here (renderer.rs:222-241, the time limit asked first)
            let c = match q.pop_front() {
                Some(c) => c,
                None => {
                    let why = if next <= self.sampling || used + 1.1 * last > self.time_limit_sec { "reached time limit" } else { "reached max sampling" };
                    self.write_image(Some(why), done, used, imgbuf);
                    return done;
                }
            };
            // report: wait for the oldest launch and print its samplings' lines (renderer.rs:206-214)
            let mut report = |c: Launch, me: &mut HipRenderer, done: &mut u32, used: &mut f64, last: &mut f64, in_flight: &mut u32| {
                check(unsafe { hr_wait(me.ctx, c.ticket) });
                let now = time::now();
                let n = c.end - c.begin;
                let t0 = if c.issued > me.last_report_progress { c.issued } else { me.last_report_progress };
                *last = Self::secs(now, t0) / n as f64;
                for j in 0..n {
                    *done = c.begin + j;
                    *used = Self::secs(t0, me.begin) + *last * (j + 1) as f64;
                    println!("rendering: {}x{} sampled (last {:.3} sec). total: {:.3} sec ({:.2} %).", *done, spp, *last, *used, *used / me.time_limit_sec * 100.0);
                }
                *in_flight -= n;
                me.last_report_progress = now;
                *used = Self::secs(now, me.begin);
            };
            report(c, self, &mut done, &mut used, &mut last, &mut in_flight);
            measured = true;
            if Self::secs(self.last_report_progress, self.last_report_image) >= self.report_interval_sec {   // renderer.rs:243-251, with the `now` of the report
                while let Some(c2) = q.pop_front() { report(c2, self, &mut done, &mut used, &mut last, &mut in_flight); }
                // nothing is in flight now: the reference's own rules apply as they stand, in their order (renderer.rs:222-241)
                if used + 1.1 * last > self.time_limit_sec { self.write_image(Some("reached time limit"), done, used, imgbuf); return done; }
                if done >= self.sampling { self.write_image(Some("reached max sampling"), done, used, imgbuf); return done; }
                self.write_image(None, done, used, imgbuf);
                self.last_report_image = self.last_report_progress;     // `now` of the report that triggered it (renderer.rs:250)
            }
        }
    }

    // renderer.rs:205-251: the trait asks for it; the loop above applies its three rules itself (it has to ask the time-limit rule at issue
    // time and to report a launch's samplings together), so nothing calls this.  Kept for a caller that drives samplings one by one.
    fn report_progress(&mut self, _acc: &Vec<Vector3>, sampling: u32, imgbuf: &mut ImageBuffer<Rgb<u8>, Vec<u8>>) -> bool {
        let now = time::now();
        let used = (now - self.begin).num_milliseconds() as f64 * 0.001;
        let last = (now - self.last_report_progress).num_milliseconds() as f64 * 0.001;
        println!("rendering: {}x{} sampled (last {:.3} sec). total: {:.3} sec ({:.2} %).", sampling,
                 config::SUPERSAMPLING * config::SUPERSAMPLING, last, used, used / self.time_limit_sec * 100.0);
        let stop_time = used + last * 1.1 > self.time_limit_sec;
        let stop_max = sampling >= self.sampling;
        let interval = (now - self.last_report_image).num_milliseconds() as f64 * 0.001 >= self.report_interval_sec;
        if stop_time || stop_max || interval {
            let path = format!("{:>03}.png", self.report_image_counter);
            if stop_time { println!("reached time limit"); } else if stop_max { println!("reached max sampling"); }
            if stop_time || stop_max { println!("output final image: {}", path); println!("remain: {:.3} sec.", self.time_limit_sec - used); }
            else { println!("output progress image: {}", path); }
            check(unsafe { hr_resolve(self.ctx, sampling, imgbuf.as_mut_ptr()) }); // RGB8, row-major, top row first
            let _ = image::ImageRgb8(imgbuf.clone()).save(&path);
            if stop_time || stop_max { return true; }
            self.report_image_counter += 1;
            self.last_report_image = now;
        }
        self.last_report_progress = now;
        false
    }
}

impl Drop for HipRenderer {
    fn drop(&mut self) { unsafe { hr_destroy(self.ctx); } }
}
