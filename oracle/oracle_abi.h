/* TEST INFRASTRUCTURE ONLY.
 *
 * One C ABI, implemented twice:
 *   - oracle/tutu_oracle.cpp  -> oracle/libtutu_oracle.so : our own CPU restatement of the reference's path-tracing
 *                                hot path (kind "port"), every function citing the reference file:line it follows;
 *   - oracle/ref_harness.cpp  -> oracle/_ref/libtutu_ref.so : a thin harness around the reference's OWN code,
 *                                compiled from /root/reference/include where the sources lie (kind "reference").
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load either library.  The product
 * (tuturenderer_amd/, include/tutu_hip.h) never links, imports or calls anything declared here.
 *
 * Conventions: all vectors are packed float32 xyz triples; arrays are row-major; functions return 0 on success,
 * a negative number on error; nothing throws across the boundary.
 */
#ifndef TUTU_ORACLE_ABI_H
#define TUTU_ORACLE_ABI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Same field order and size (56 B) as the reference's Material (Material.hpp:19-30). type = MaterialType enum
 * order (Material.hpp:9-16): 0 LAMBERTIAN 1 PERFECT_REFLECTIVE 2 PERFECT_REFRACTIVE 3 MICROFACET_R 4 MICROFACET_T
 * 5 UNLIT. */
typedef struct TorMaterial {
	float diffuse[3];
	float specular[3];
	float emission[3];
	int32_t type;
	float alpha;
	float eta;
	float roughness;
	float metallic;
} TorMaterial;

typedef struct TorTexture {
	int32_t width, height;
	const float* rgb; /* width*height*3, row-major (Texture::rgb) */
} TorTexture;

typedef struct TorSceneDesc {
	int32_t n_tris;
	const float* verts;    /* n_tris*9 : v0 v1 v2 */
	const float* normals;  /* n_tris*9 : n0 n1 n2 (un-normalised allowed, as the reference stores them) */
	const int32_t* mat_id; /* n_tris */
	int32_t n_mats;
	const TorMaterial* mats;
	float eta;    /* scene index of refraction (4th number of `bkgcolor`) */
	float bkg[3]; /* background colour */
	int32_t width, height;
	float eye[3], viewdir[3], updir[3];
	int32_t hfov; /* integer degrees, like the config keyword */
	/* optional textures: NULL / 0 = none.  Four map lists like PPMGenerator's diffuseMaps,
	 * normalMaps, roughnessMaps, metallicMaps; normal-map texels already mapped to [-1,1] (c*2-1), as the `bump`
	 * keyword does at load time (PPMGenerator.hpp:713-722). */
	const float* uvs;       /* n_tris*6: uv0 uv1 uv2 */
	const int32_t* tex_ids; /* n_tris*4: diffuse, normal, roughness, metallic map index, -1 = none */
	int32_t n_textures[4];
	const struct TorTexture* textures[4];
	/* optional spheres: Scene::objList is the triangles with the spheres inserted at the given
	 * object-list positions (ascending; NULL = all spheres after the triangles).  With spheres present every
	 * "triangle index" this ABI reports is an index into that combined object list. */
	int32_t n_spheres;
	const float* spheres;            /* n_spheres*4: centre xyz, radius */
	const int32_t* sphere_mat_id;    /* n_spheres */
	const int32_t* sphere_tex_ids;   /* n_spheres*4 or NULL */
	const int32_t* sphere_pos;       /* n_spheres or NULL */
} TorSceneDesc;

const char* tor_kind(void); /* "port" or "reference" */

/* ---- RNG ---- */
int tor_philox4x32_10(int n, const uint32_t* ctr4, uint32_t key0, uint32_t key1, uint32_t* out4);
/* xi stream exactly as a sample consumes it: draw k of (pix,smp) under key (k0,k1) */
int tor_rng_stream(uint32_t pix, uint32_t smp, uint32_t key0, uint32_t key1, int n, float* xi);

/* ---- pure functions ---- */
int tor_bbox_intersect(int n, const float* pmin, const float* pmax, const float* o, const float* d, uint8_t* hit);
int tor_tri_intersect(int n, const float* verts9, const float* normals9, const float* o, const float* d,
                      uint8_t* hit, float* t, float* pos, float* Ns, float* Ng);
int tor_tri_area(int n, const float* verts9, float* area);
int tor_math_normalized(int n, const float* v, float* out);
int tor_math_fresnel(int n, const float* I, const float* N, const float* eta_i, const float* eta_t, float* out);
int tor_math_fresnel_schlick(int n, const float* cos_theta, const float* F0, float* out3);
int tor_math_reflect(int n, const float* I, const float* N, float* out);
int tor_math_refract(int n, const float* I, const float* N, const float* eta_i, const float* eta_t, float* out);
int tor_math_D(int n, const float* h, const float* nrm, const float* rough, float* out);
int tor_math_G(int n, const float* wi, const float* wo, const float* nrm, const float* rough, const float* h,
               float* out);
int tor_math_mis(int n, const float* a, const float* b, float* out);
int tor_math_local2world(int n, const float* N, const float* dir, float* out);
int tor_write_pixel(int n, const float* c, int32_t* out);
/* Texture::getRGBat (Texture.hpp:18-39) on one texture */
int tor_texture_lookup(const struct TorTexture* t, int n, const float* u, const float* v, float* rgb);

/* ---- post-processing: Postprocessor.hpp with HDR_BLOOM (global.hpp:32) ----
 * stage 0 = the whole performPostProcess (bloom then exposure tone map), 1 = getEmmisiveTexture, 2 = getGaussianBlurTexture
 * (KERNELSIZE 10, STDDEV 30), 3 = getHDRtexture alone.  in / out: width*height*3 floats. */
int tor_postprocess(int stage, int width, int height, const float* in, float* out);

/* ---- material (one material, n evaluations) ---- */
int tor_mat_bxdf(int n, const TorMaterial* m, const float* wi, const float* wo, const float* Ng, const float* Ns,
                 float eta_scene, const uint8_t* tir, float* out3);
int tor_mat_pdf(int n, const TorMaterial* m, const float* wi, const float* wo, const float* N, float eta_i,
                float eta_t, float* out);
/* xi: n*3 injected draws per call (consumed in order); ndraws = how many were consumed */
int tor_mat_sample(int n, const TorMaterial* m, const float* wo, const float* N, float eta_i, const float* xi3,
                   float* wi, uint8_t* ok, uint8_t* special, int32_t* ndraws);

/* ---- scene ---- */
int tor_scene_create(const TorSceneDesc* d, void** out);
int tor_scene_destroy(void* h);
/* pre-order dump of the BVH: bounds6 = pMin,pMax ; leaf_tri = triangle index for a leaf, -1 for an inner node */
int tor_scene_bvh_dump(void* h, int cap, int32_t* n_nodes, float* bounds6, int32_t* leaf_tri);
int tor_scene_closest(void* h, int n, const float* o, const float* d, uint8_t* hit, float* t, int32_t* tri,
                      float* pos, float* Ns, float* Ng);
int tor_scene_any(void* h, int n, const float* orig, const float* target, uint8_t* blocked);
int tor_scene_lights(void* h, int cap, int32_t* n_lights, int32_t* tri);
int tor_scene_sample_light(void* h, int n, const float* xi3, int32_t* tri, float* pos, float* nrm, float* pdf);
int tor_scene_light_pdf(void* h, int n, const int32_t* tri, float* pdf);
/* out18 = ul, delta_h, delta_v, c_off_h, c_off_v, eye  (PathTracing.hpp:357-391) */
int tor_camera(void* h, float* out18);
int tor_camera_raydir(void* h, int n, const int32_t* px, const int32_t* py, float* d);
/* out22 = world2Raster (row-major 4x4), imagePlaneDist, filmPlaneAreaInv, lensAreaInv, fwdDir  (Camera.hpp:12-49) */
int tor_camera_raster(void* h, float* out22);

/* ---- estimator ---- */
/* radiance of individual samples (pixel index = y*W+x, sample index) under key (k0,k1); ndraws/nclosest optional */
int tor_trace_samples(void* h, int n, const uint32_t* pix, const uint32_t* smp, uint32_t key0, uint32_t key1,
                      float* L3, int32_t* ndraws, int32_t* nclosest);
/* render rect [x0,x1) x [y0,y1) at spp into rgb (full H*W*3 buffer, only the rect is written) */
int tor_render(void* h, int spp, uint32_t key0, uint32_t key1, int x0, int y0, int x1, int y1, int nthreads,
               float* rgb);

/* ---- the other integrators behind the IIntegrator seam (Renderer.hpp:41-49): type 1 LightTracing,
 * 2 NaivePT, 3 BDPT.  A whole frame in the reference's own loop order, single-threaded, with ONE sequential random
 * stream (Philox counter (0xFFFFFFFF, type, draw >> 2, 0)); rgb = H*W*3, starts as the background colour like
 * Camera::initialize leaves the framebuffer.  Implemented by both libraries: the reference harness calls the reference's
 * own LightTracing::integrate / NaivePT::integrate / sub_render_bdpt. */
int tor_render_integrator(void* h, int type, int spp, uint32_t key0, uint32_t key1, float* rgb);
/* port only: single (pixel, sample) units with their own stream, and a frame assembled from such units */
int tor_integrator_samples(void* h, int type, int spp, int n, const uint32_t* pix, const uint32_t* smp, uint32_t key0, uint32_t key1,
                           float* own3, uint8_t* alive, int max_ev, int32_t* n_ev, int32_t* ev_op, int32_t* ev_index, float* ev_rgb);
int tor_render_integrator_units(void* h, int type, int spp, uint32_t key0, uint32_t key1, int nthreads, float* rgb);

#ifdef __cplusplus
}
#endif
#endif
