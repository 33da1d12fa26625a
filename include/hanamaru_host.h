/*
 * hanamaru_host.h — host-side scene authoring / asset IO (the part of the reference that stays on the
 * host: loader.rs, camera.rs, matrix.rs, the init_scene_* builders of main.rs, image decode/encode).
 *
 * The reference host is Rust; there is no Rust toolchain in this environment, so this layer is C++17
 * behind a C ABI.  It produces the `hr_scene_desc` that include/hanamaru_hip.h consumes.  Nothing in
 * here touches the GPU.
 */
#ifndef HANAMARU_HOST_H
#define HANAMARU_HOST_H

#include "hanamaru_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hh_scene hh_scene;

const char *hh_last_error(void);

/* Scenes (asset paths are resolved relative to `asset_root`, which must contain models/ and textures/):
 *   "rtcamp6_v3_1"  — main.rs:1020-1153, the reference's live scene (BASELINE configs 1, 3, 4)
 *   "spheres"       — BASELINE config 2: the 100+5 sphere generator of main.rs:862-905 (ISAAC-64 seed
 *                     [870,2000,304,2], gen_range + AABB-collision rejection), camera and skybox of main.rs:808-858,
 *                     GGX replaced by alternating Diffuse/Specular, no mesh
 *   "rtcamp6_v2"    — main.rs:804-926 as written: the same generator with GGX f0 0.9 spheres, five emitters, the refractive
 *                     fractal dodecahedron, Ryfjallet skybox
 *   "rtcamp6_v1"    — main.rs:725-802: emissive sphere inside the refractive houdini_boss mesh, checkered floor
 *   "rtcamp5"       — main.rs:252-500: bunnies, an earth-textured emissive sphere, image roughness on a sphere, TIFF marble floor,
 *                     1 + 12 + 30 diamonds placed by gen_range draws with AABB-collision rejection (matches the reference's rtcamp5.png)
 *   "tbf3"          — main.rs:502-724: KLab logo mesh, four earth-textured emitters, 8 + 20 generated spheres / diamonds
 *   "material_examples" — main.rs:139-250: one sphere per surface type (incl. GGXRefraction) under a spherical light
 *   "rtcamp6_dodeca"— BASELINE config 5: rtcamp6_v3_1 + models/fractal_dodecahedron.obj with the
 *                     Refraction-1.5 material of main.rs:910-915
 *   "rtcamp6_v3"    — main.rs:928-1017: two emissive spheres (one of radius 1 mm next to the camera), aperture 0.2
 *   "simple"        — main.rs:54-136: GGX floor with image albedo + image roughness, two coloured emitters, skybox intensity 0
 *   "cornell_mini"  — tiny build-defined scene touching all five surface types + textured sphere (tests)
 */
int hh_scene_create(const char *name, const char *asset_root, hh_scene **out);
const hr_scene_desc *hh_scene_desc(const hh_scene *scene);
void hh_scene_destroy(hh_scene *scene);

/* loader.rs:12-59 semantics.  `matrix` = 16 doubles, row-major Matrix44 (matrix.rs:5-7). */
int hh_load_obj(const char *path, const double *matrix, hr_vec3 **vertexes, uint64_t *num_vertexes,
                uint64_t **faces, uint64_t *num_faces);
/* PNG (8-bit, colour types 0/2/3/4/6, non-interlaced), baseline JPEG and baseline TIFF (8-bit strips, raw or LZW, optional
 * horizontal predictor) -> RGBA8, row 0 = top. */
int hh_decode_image(const char *path, uint8_t **rgba, uint32_t *width, uint32_t *height);
int hh_write_png_rgb8(const char *path, const uint8_t *rgb, uint32_t width, uint32_t height);
void hh_free(void *p);

/* camera.rs:45-64 Camera::new */
void hh_camera_new(hr_vec3 eye, hr_vec3 target, hr_vec3 y_up, double v_fov_deg, int32_t lens_shape,
                   double aperture, double focus_distance, hr_camera *out);

#ifdef __cplusplus
}
#endif
#endif
