/* tutu_hip.h -- C ABI of the MI355X (gfx950) path-tracing integrator.
 *
 * This library is a drop-in for ONE path of bobhansky/TutuRenderer: what happens inside
 *     IIntegrator::integrate(PPMGenerator* g)            (reference include/IIntegrator.hpp:17-24)
 * when the integrator is PathTracing (include/PathTracing.hpp:352-516), i.e. the per-pixel / per-sample loop
 * sub_render_pt -> PathTracing::traceRay -> BVH closest-hit / any-hit -> Material BxDF / sample / pdf.
 * The reference has no process or device boundary anywhere; this header inserts one at that seam.  A host-side
 * C++ mirror of the reference's Renderer / IIntegrator / Scene / PPMGenerator surface that calls these entry points
 * lives in tuturenderer_amd/host/ (see INTEGRATION.md for the binding a maintainer of the reference would add).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.  Every function returns 0 (TUTU_OK) or a negative
 *     TUTU_E_* code; nothing throws across the boundary; tutu_hip_error_string() names a code.
 *   - The caller owns every host buffer it passes; the library copies what it needs during the call.  The library
 *     owns all device memory behind the opaque TutuCtx.
 *   - One host thread per context at a time (the reference's integrate() is not re-entrant either: globals SPP,
 *     records -- global.hpp:19, IIntegrator.hpp:15).  A multi-GPU driver creates one context per device.
 *   - All arithmetic is IEEE fp32 without FMA contraction, in the reference's expression order.
 */
#ifndef TUTU_HIP_H
#define TUTU_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TUTU_OK 0
#define TUTU_E_INVALID -1      /* bad argument (null pointer, negative size, index out of range) */
#define TUTU_E_NO_DEVICE -2    /* no HIP device / device index out of range */
#define TUTU_E_HIP -3          /* a HIP runtime call failed; see tutu_hip_last_error() */
#define TUTU_E_OOM -4          /* device or host allocation failed */
#define TUTU_E_BVH_DEPTH -5    /* BVH deeper than the traversal stack (TUTU_MAX_BVH_DEPTH) */
#define TUTU_E_UNSUPPORTED -6  /* feature outside the PathTracing hot path */

#define TUTU_MAX_BVH_DEPTH 30

/* MaterialType, same order as reference Material.hpp:9-16 */
enum TutuMaterialType {
	TUTU_LAMBERTIAN = 0,
	TUTU_PERFECT_REFLECTIVE = 1,
	TUTU_PERFECT_REFRACTIVE = 2,
	TUTU_MICROFACET_R = 3,
	TUTU_MICROFACET_T = 4,
	TUTU_UNLIT = 5
};

/* Same fields, order and size (56 B) as the reference's `class Material` data members (Material.hpp:21-30). */
typedef struct TutuMaterial {
	float diffuse[3];
	float specular[3];
	float emission[3];
	int32_t type; /* TutuMaterialType */
	float alpha;  /* opacity */
	float eta;    /* index of refraction */
	float roughness;
	float metallic;
} TutuMaterial;

/* What Renderer::render() sees of the scene when it calls integrate(g): the triangle list of g->scene.objList in
 * load order (PPMGenerator::loadObj, PPMGenerator.hpp:164-208), each triangle's Material, the scene IOR g->eta and
 * g->bkgcolor.  Lights are not passed: like PPMGenerator::initializeLights (PPMGenerator.hpp:317-324) the library
 * takes every triangle whose material has a non-zero emission, in list order. */
/* One image of PPMGenerator's diffuseMaps / normalMaps / roughnessMaps / metallicMaps (PPMGenerator.hpp:36-39), as
 * loadTexture leaves it (Texture.hpp:10-16; PPMGenerator.hpp:1027-1084): width*height float RGB triples, row-major,
 * channel/255.  Normal maps are passed AFTER the `bump` keyword's c*2-1 mapping (PPMGenerator.hpp:713-722). */
typedef struct TutuTexture {
	int32_t width, height;
	const float* rgb; /* width*height*3 */
} TutuTexture;

/* Optional per-triangle texture data: what PPMGenerator::loadObj stores on each Triangle when a texture keyword is
 * active (uv0..uv2, textureIndex, normalMapIndex, roughnessMapIndex, metallicMapIndex; PPMGenerator.hpp:182-201,
 * Object.hpp:31-35).  A triangle is "texture activated" when any of its four indices is not -1. */
typedef struct TutuTextureSet {
	const float* uvs;       /* n_tris*6: uv0 uv1 uv2 */
	const int32_t* tex_ids; /* n_tris*4: diffuse, normal, roughness, metallic map index; -1 = none */
	uint32_t n_maps[4];     /* list sizes, in the same order */
	const TutuTexture* maps[4];
} TutuTextureSet;

/* Optional spheres: the `sphere cx cy cz r` objects of config.txt (Sphere.hpp:6-21; PPMGenerator.hpp:358-402).
 * Scene::objList is then the triangles with the spheres inserted at the given object-list positions (ascending;
 * NULL = all spheres after the triangles) -- the order matters because BVHAccel::recursiveBuild sorts that list and
 * because the light list is taken in that order.  With spheres present, every "triangle index" this ABI reports
 * (TutuHit::tri, eval_sample_light's tri) is an index into the combined object list.
 * Reference quirks kept: the quadratic assumes a unit direction (A = 1) even for the un-normalised mirror/refraction
 * directions; Sphere::getArea is pi*r*r; Sphere::samplePoint draws the two angles uniformly. */
typedef struct TutuSphereSet {
	uint32_t n_spheres;
	const float* spheres;    /* n_spheres*4: centre xyz, radius */
	const int32_t* mat_id;   /* n_spheres: index into TutuSceneDesc::mats */
	const int32_t* tex_ids;  /* n_spheres*4 (diffuse, normal, roughness, metallic map; -1 = none) or NULL */
	const int32_t* pos;      /* n_spheres object-list positions or NULL */
} TutuSphereSet;

typedef struct TutuSceneDesc {
	uint32_t n_tris;
	const float* verts;    /* n_tris*9: v0 v1 v2 (Triangle.hpp:11-14) */
	const float* normals;  /* n_tris*9: n0 n1 n2, un-normalised allowed (Triangle.hpp:16) */
	const int32_t* mat_id; /* n_tris: index into mats */
	uint32_t n_mats;
	const TutuMaterial* mats;
	float eta;    /* g->eta: 4th number of the `bkgcolor` keyword */
	float bkg[3]; /* g->bkgcolor */
	const TutuTextureSet* textures; /* NULL: no textured triangles (textureModify, IIntegrator.hpp:89-127, never runs) */
	const TutuSphereSet* spheres;   /* NULL: triangles only */
} TutuSceneDesc;

/* The camera keywords of config.txt (imsize / eye / viewdir / hfov / updir; PPMGenerator.hpp:492-540). */
typedef struct TutuCameraDesc {
	int32_t width, height;
	float eye[3], viewdir[3], updir[3];
	int32_t hfov; /* integer degrees */
} TutuCameraDesc;

/* The six vectors PathTracing::integrate derives from the camera (PathTracing.hpp:357-391) and hands to its
 * worker threads (Thread_arg_pt, PathTracing.hpp:13-23). */
typedef struct TutuCameraFrame {
	int32_t width, height;
	float ul[3], delta_h[3], delta_v[3], c_off_h[3], c_off_v[3], eye[3];
} TutuCameraFrame;

typedef struct TutuRenderParams {
	int32_t spp;         /* the reference's global `SPP` (global.hpp:19) */
	uint32_t key0, key1; /* Philox4x32-10 key; draw k of sample s of pixel p = word k&3 of philox((p, s, k>>2, 0), key) */
	/* Work list: either an explicit list of pixel indices (y*width+x), or, when pixels == NULL, the rectangle
	 * [x0,x1) x [y0,y1) in row-major order.  Output i of the call belongs to work item i. */
	const int32_t* pixels;
	int32_t n_pixels;
	int32_t x0, y0, x1, y1;
	int32_t spp_per_pass; /* samples per pixel traced per wavefront pass; 0 = choose from max_paths */
	int64_t max_paths;    /* cap on paths in flight, summed over the (up to 4) work sets whose passes overlap (device memory ~ 400 B each); 0 = default (option "paths_mi" TUTU_PATHS_MI [4,4096], 168 Mi = 67 GB), less if the device has less free.
	                       * With the default (both 0), a context's FIRST render allocates only knob "cold_paths_mi" (12 Mi slots, 4.8 GB) itself and a host thread
	                       * allocates the rest meanwhile (tutu_hip_work_ready); a caller that names a size gets exactly that, allocated before the call renders. */
} TutuRenderParams;

typedef struct TutuStats {
	uint64_t samples;      /* work items * spp */
	uint64_t closest_rays; /* closest-hit rays traced (primary rays counted once per pixel, as traced) */
	uint64_t shadow_rays;  /* any-hit rays traced */
	uint64_t segments;     /* path vertices shaded */
	uint32_t passes;
	uint32_t trace_launches; /* launches of the closest-hit traversal kernel */
	float ms_total;          /* device time of the whole call's kernels, HIP events on the library's stream.  The ms_* fields and
	                          * shade_material_launches are filled only with the knob "kernel_events" = 1 (TUTU_KERNEL_EVENTS; default 0) */
	float ms_trace_closest;  /* summed over launches */
	float ms_trace_any;
	float ms_shade;          /* all shade launches (= ms_shade_first + ms_shade_material + ms_shade_terminal) */
	float ms_other;          /* primary rays, list building, resolve */
	float ms_shade_first;    /* depth-0 launches (k_shade<SHADE_FIRST>) */
	float ms_shade_material; /* the shade launches of depths 1..MAX_DEPTH (one per stage, or one per material class with TUTU_CLASS_SORT) */
	float ms_shade_terminal; /* launches that only connect (last depth; scenes without scattering materials) */
	uint32_t shade_material_launches;
	/* traversal work measured on the device, summed over the call's queue-stage rays (primary rays excluded):
	 * nodes entered and leaf (triangle / sphere) tests, for closest-hit and any-hit rays */
	uint64_t nodes_closest, leaves_closest, nodes_any, leaves_any;
	uint32_t spp_per_pass; /* samples per pixel of a full wavefront pass of this call (the last pass may be shorter) */
	uint32_t n_sets;       /* passes that were in flight at once */
	/* how often a WAVE of the traversal kernels executed its inner-node step / its leaf step: nodes_* (without the leaves)
	 * / (64 * wave_node_steps_*) = the fraction of lanes that had a node to test in such a step */
	uint64_t wave_node_steps_closest, wave_leaf_steps_closest, wave_node_steps_any, wave_leaf_steps_any;
} TutuStats;

typedef struct TutuHit {
	float t;     /* FLT_MAX on a miss */
	float b1;    /* barycentric weight of v1 */
	float b2;    /* barycentric weight of v2 */
	int32_t tri; /* index into the caller's triangle list, -1 on a miss */
} TutuHit;

/* Flattened BVH as the traversal kernels read it (host-side builder exposed for parity tests). */
typedef struct TutuBvhInfo {
	uint32_t n_tris;
	uint32_t n_inner;    /* inner nodes (64 B each) */
	uint32_t depth;      /* longest root-to-leaf path, in edges */
	float root_bounds[6];
} TutuBvhInfo;

typedef struct TutuCtx TutuCtx;

/* ------------------------------------------------------------------------------------------------------------
 * Host-only helpers (no GPU needed) */

/* Camera::initialize (Camera.hpp:12-17,43-44) + the camera-frame lines of PathTracing::integrate
 * (PathTracing.hpp:357-391). */
int tutu_camera_frame(const TutuCameraDesc* cam, TutuCameraFrame* out);

/* What LightTracing / NaivePT / BDPT read from g->cam (Camera.hpp:12-79): position, fwdDir, the three scalars of
 * Camera::initialize (:43-47) and world2Raster = scale(w/2, h/2, 0) * translate(1, 1, 0) * perspective(hfov, 0.1, 10000,
 * w/h) * world2Cam (:29-41; Vector.hpp:338-373), row-major -- the matrix behind Camera::worldPos2PixelIndex (:61-79). */
typedef struct TutuCameraRaster {
	float position[3], fwdDir[3];
	int32_t width, height;
	float imagePlaneDist, filmPlaneAreaInv, lensAreaInv;
	float world2raster[16];
} TutuCameraRaster;
int tutu_camera_raster(const TutuCameraDesc* cam, TutuCameraRaster* out);

/* BVHAccel::recursiveBuild (BVH.hpp:47-123) over the triangle list: median split on the centroid along the
 * longest axis, one triangle per leaf, same std::sort, same tie order.  Writes the tree in pre-order like the
 * parity oracles do: bounds6[i] = pMin,pMax ; leaf_tri[i] = triangle index or -1 for an inner node.
 * cap = capacity of the two arrays in nodes (2*n_tris-1 are needed). */
int tutu_bvh_build_preorder(uint32_t n_tris, const float* verts, uint32_t cap, uint32_t* n_nodes, float* bounds6,
                            int32_t* leaf_tri, TutuBvhInfo* info);

/* Host-only (no GPU), for tests of the builder: the EIGHT-WIDE tree tutu_hip_create builds for memory-resident scenes (round 5;
 * csrc/host_scene.hpp: GpuWide8Node, 128 B per node id).  It is one of the trees the traversal kernels may WALK instead of the
 * reference's BVHAccel tree (BVH.hpp:47-123) -- never what decides a hit: every candidate is validated against the reference's leaf
 * box (csrc/device_trace.h).  nodes128: cap_ids records of 128 B or NULL; *n_ids = ids in use (the ids of slots that hold no inner
 * child are zero-filled holes), *n_nodes = nodes that exist, *depth = levels; leaf_boxes8: the reference's leaf box of every object in
 * leaf order, [cap_objects][8] floats (min xyz, -, max xyz, -) or NULL; *margin = the quantisation margin of the boxes.
 * TUTU_E_INVALID when the scene gets no such tree (fewer than three objects, coordinates the 8-bit frames cannot serve). */
int tutu_host_wide8(const TutuSceneDesc* scene, uint32_t cap_ids, void* nodes128, uint32_t* n_ids, uint32_t* n_nodes, uint32_t* depth,
                    uint32_t cap_objects, float* leaf_boxes8, uint32_t* n_objects, double* margin);

const char* tutu_hip_error_string(int code);
const char* tutu_hip_last_error(void); /* text of the last HIP failure on this thread */
const char* tutu_hip_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Device entry points */

int tutu_hip_device_count(int* count);

/* Renderer::Renderer's scene set-up for this path (Renderer.hpp:35-54: Scene::initializeBVH) plus
 * Renderer::render's g->initializeLights() (Renderer.hpp:64): builds the BVH on the host, flattens scene, lights
 * and materials to linear arrays and uploads them to `device`. */
int tutu_hip_create(const TutuSceneDesc* scene, int device, TutuCtx** out);
int tutu_hip_destroy(TutuCtx* ctx);

/* IIntegrator::integrate for PathTracing: out_rgb[i*3..] = linear radiance of work item i, exactly what
 * sub_render_pt stores into g->cam.FrameBuffer.rgb (PathTracing.hpp:501-513): (1/spp) * sum over samples whose
 * radiance has no NaN component.  out_rgb is a HOST pointer to n_items*3 floats. */
int tutu_hip_render(TutuCtx* ctx, const TutuCameraFrame* cam, const TutuRenderParams* params, float* out_rgb,
                    TutuStats* stats);

/* Cold start.  The reference's scene programs render once per process (src/main_cornellBox.cpp:75-79): what the FIRST render of
 * a context costs is what a drop-in user waits for.  Device allocations cost time in proportion to their size whenever the
 * driver has to clear memory another process left behind (0.0 - 1.7 s for the default 67 GB of work sets), so with the default
 * sizing the first render allocates a small set of work buffers itself (knob "cold_paths_mi"), renders with smaller passes,
 * and a host thread owned by the context allocates the full-size sets meanwhile; the first later render that finds them ready
 * switches over (same frame, bit for bit: only the grouping of samples into passes changes).
 * tutu_hip_work_ready: returns 1 while that allocation is still running, 0 when the context has its full-size sets (or never
 * needed more), negative on error; wait != 0 blocks until it is done.  A caller that renders many frames and wants the
 * steady state from the first one sets "cold_paths_mi" to 0 before its first render (bench.py's timed context does). */
int tutu_hip_work_ready(TutuCtx* ctx, int wait);

/* Same, but out_rgb is a DEVICE pointer on ctx's device (e.g. a torch tensor's data_ptr).  The frame is ORDERED with `stream` (a
 * hipStream_t of that device; NULL = no ordering with anything of the caller's): its first kernel waits for what is enqueued on
 * `stream` at the time of the call, and `stream` waits for its last kernel -- the kernels themselves run on the context's own four
 * streams (one hardware queue each; a pass on a fifth stream of the caller's would share a queue with another pass).  The call returns after the work is complete (it ends with a hipStreamSynchronize of that stream: the ray counters of TutuStats
 * are read back then; stats == NULL skips the read-back, not the wait). */
int tutu_hip_render_device(TutuCtx* ctx, const TutuCameraFrame* cam, const TutuRenderParams* params,
                           float* d_out_rgb, void* stream, TutuStats* stats);

/* The N-device form of tutu_hip_render: `n` contexts of the SAME scene (normally one per device; several on one device
 * work too), one host thread per context inside the call.  Replaces the static row split over std::threads of
 * PathTracing::integrate (PathTracing.hpp:393-429): work items are dealt to the contexts in 32x32 pixel tiles,
 * round-robin; every context renders its items and writes them to out_rgb (host pointer, item order as in
 * tutu_hip_render).  The frame is bit-identical for any n.  stats: n entries or NULL. */
int tutu_hip_render_multi(TutuCtx* const* ctxs, int32_t n, const TutuCameraFrame* cam, const TutuRenderParams* params,
                          float* out_rgb, TutuStats* stats);
/* Same, with the frame left in DEVICE memory of ctxs[0]'s device (d_out_rgb: n_items*3 floats there).  The gather is on the
 * device side -- what replaces the shared frame buffer the reference's threads write into (PathTracing.hpp:393-429): every
 * context renders its piece into its own device's memory, the pieces travel to ctxs[0]'s device into one buffer in context
 * order, and one kernel un-tiles it into d_out_rgb on `stream` (a hipStream_t of that device, NULL = ctxs[0]'s own); the call
 * returns after that kernel.  How the pieces travel (option "gather_rccl" of ctxs[0]: 0 / 1 / 2; read-only "gather_path",
 * "rccl_available", "peer_access"): contexts on SEVERAL devices -- ONE grouped RCCL exchange over xGMI (ncclCommInitAll over
 * the distinct devices, ncclGroupStart ... ncclSend / ncclRecv ... ncclGroupEnd; RCCL is looked up at run time, the
 * communicators are kept for the next frame); contexts that share a device, RCCL absent or switched off -- peer / local copies,
 * with peer access enabled explicitly.  tutu_hip_render_multi is this + one frame-sized D2H copy.  (bench.py's
 * process-per-GPU form does the same exchange through torch.distributed: tuturenderer_amd/dist.py.) */
int tutu_hip_render_multi_device(TutuCtx* const* ctxs, int32_t n, const TutuCameraFrame* cam, const TutuRenderParams* params,
                                 float* d_out_rgb, void* stream, TutuStats* stats);

/* Kernel-level entry points for parity tests.
 * closest: BVHStrategy::UpdateInter -> getIntersection (BVHStrategy.hpp:8-11, BVH.hpp:145-167)
 * any:     isShadowRayBlocked -> hasIntersection (IIntegrator.hpp:135-153, BVH.hpp:170-194): orig and the target
 *          point (light position), blocked[i] = 1 when some triangle is hit with t < dist and |t-dist| >= 1e-4. */
int tutu_hip_trace_closest(TutuCtx* ctx, uint32_t n, const float* orig, const float* dir, TutuHit* hits);
int tutu_hip_trace_any(TutuCtx* ctx, uint32_t n, const float* orig, const float* target, uint8_t* blocked);

/* Radiance of individual samples (pixel index, sample index) -- one PathTracing::traceRay(eye, dir, 0, 1) each
 * (PathTracing.hpp:509), NaN samples returned as they are.  Runs the same wavefront pipeline as tutu_hip_render. */
int tutu_hip_trace_samples(TutuCtx* ctx, const TutuCameraFrame* cam, uint32_t n, const uint32_t* pix,
                           const uint32_t* smp, uint32_t key0, uint32_t key1, float* L3);

/* The reference's OTHER integrators behind the same seam (Renderer.hpp:41-49 picks by PPMGenerator::integrateType;
 * SURVEY.md 8f-4) -- IIntegrator::integrate of
 *   TUTU_INTEGRATOR_LIGHT    LightTracing::integrate  (LightTracing.hpp:23-207: one light path per (pixel, sample) splatted
 *                            onto the pixel it projects to; the directly visible light is a setRGB [sic])
 *   TUTU_INTEGRATOR_NAIVEPT  NaivePT::integrate       (NaivePT.hpp:24-164: emission of the first hit times the camera terms)
 *   TUTU_INTEGRATOR_BDPT     BDPT::integrate          (BDPT.hpp:387-900: sub_render_bdpt, every (s, t) strategy up to path
 *                            length 7, MISweight :70-230)
 * over the whole frame: out_rgb (HOST, width*height*3) = g->cam.FrameBuffer.rgb after integrate(), starting from the
 * background colour (Camera.hpp:26-27).  The reference draws from one global rand() stream shared by its threads, so its
 * own picture is not reproducible; here unit (pixel p, sample s) has the Philox stream (p, s) of tutu_hip_render, and the
 * frame is what ONE thread running the reference's loops in (y, x, sample) order would leave behind with those streams:
 * setRGB / addRGB are applied per target pixel in that order (a later setRGB replaces what earlier units added, as there).
 * TUTU_INTEGRATOR_PATH is refused here (tutu_hip_render is that integrator).  stats: samples, ms_total; may be NULL. */
enum TutuIntegrator { TUTU_INTEGRATOR_PATH = 0, TUTU_INTEGRATOR_LIGHT = 1, TUTU_INTEGRATOR_NAIVEPT = 2, TUTU_INTEGRATOR_BDPT = 3 };
int tutu_hip_render_integrator(TutuCtx* ctx, int32_t type, const TutuCameraDesc* cam, int32_t spp, uint32_t key0,
                               uint32_t key1, float* out_rgb, TutuStats* stats);
/* Individual units of those integrators, for parity tests: unit i = (pix[i], smp[i]) of an spp-sample frame.
 * own3[i] = what the unit adds to its own pixel's estimate (before the 1/spp scaling; zero for LightTracing),
 * alive[i] = 0 when the primary ray missed (NaivePT `continue`, BDPT `break`), and the unit's frame-buffer events in
 * program order: n_ev[i] of them (at most TUTU_MAX_UNIT_EVENTS), event k at [i * max_ev + k]: ev_op 0 = setRGB /
 * 1 = addRGB, ev_index = target pixel, ev_rgb = the value (already scaled by 1/spp, as the reference passes it). */
#define TUTU_MAX_UNIT_EVENTS 8
int tutu_hip_integrator_samples(TutuCtx* ctx, int32_t type, const TutuCameraDesc* cam, int32_t spp, uint32_t n,
                                const uint32_t* pix, const uint32_t* smp, uint32_t key0, uint32_t key1, float* own3,
                                uint8_t* alive, int32_t max_ev, int32_t* n_ev, int32_t* ev_op, int32_t* ev_index,
                                float* ev_rgb);

/* Device evaluation of the material functions on arrays (one material, n evaluations), for function-level parity:
 * Material::BxDF (Material.hpp:62-191), Material::pdf (:350-439), Material::sampleDirection (:200-343) with the
 * three random numbers of each evaluation supplied by the caller. */
int tutu_hip_eval_bxdf(TutuCtx* ctx, uint32_t n, const TutuMaterial* m, const float* wi, const float* wo,
                       const float* Ng, const float* Ns, float eta_scene, const uint8_t* tir, float* out3);
int tutu_hip_eval_pdf(TutuCtx* ctx, uint32_t n, const TutuMaterial* m, const float* wi, const float* wo,
                      const float* N, float eta_i, float eta_t, float* out);
int tutu_hip_eval_sample(TutuCtx* ctx, uint32_t n, const TutuMaterial* m, const float* wo, const float* N,
                         float eta_i, const float* xi3, float* wi, uint8_t* ok, uint8_t* special, int32_t* ndraws);
/* Texture::getRGBat (Texture.hpp:18-39) on map `index` of list `list` (0 diffuse, 1 normal, 2 roughness, 3 metallic)
 * of the context's scene; TUTU_E_INVALID when the scene has no such map. */
int tutu_hip_eval_texture(TutuCtx* ctx, int32_t list, int32_t index, uint32_t n, const float* u, const float* v,
                          float* rgb);
/* sampleLight (IIntegrator.hpp:173-192) + Triangle::samplePoint (Triangle.hpp:119-142) with supplied xi */
int tutu_hip_eval_sample_light(TutuCtx* ctx, uint32_t n, const float* xi3, int32_t* tri, float* pos, float* nrm,
                               float* pdf);

/* Device evaluation of the remaining hot-path functions on arrays, for function-level parity (SURVEY.md 8a rows a10, a11,
 * a16, a17).  in[k] = the k-th input array, n rows; widths (floats per row) in -> out:
 *   BBOX            BoundBox::IntersectRay (BoundBox.hpp:55-92)        pmin3 pmax3 o3 d3 -> hit (0/1)
 *   TRI             Triangle::intersect (Triangle.hpp:23-74)           verts9 normals9 o3 d3 -> hit, t, pos3, Ns3, Ng3 (zeros on a miss)
 *   NORMALIZED      Vector3f::normalized (Vector.hpp:213-220)          v3 -> 3
 *   FRESNEL         global.hpp:242-261                                 I3 N3 eta_i eta_t -> 1
 *   FRESNEL_SCHLICK global.hpp:236-239                                 cos F0_3 -> 3
 *   REFLECT / REFRACT  getReflectionDir / getRefractionDir (:264-301)  I3 N3 [eta_i eta_t] -> 3
 *   D / G           D_ndf (:311-324) / G_smf (:334-346)                h3 n3 rough -> 1 / wi3 wo3 n3 rough h3 -> 1
 *   MIS             getMisWeight (:374-380)                            a b -> 1
 *   LOCAL2WORLD     SphereLocal2world (:387-410)                       N3 dir3 -> 3
 *   RNG             the xi stream that replaces getRandomFloat         (pix, smp, key0, key1, first draw) as uint32 -> 8 xi
 *   PHILOX          Philox4x32-10 block                                (ctr0, ctr1, ctr2, key0, key1) as uint32 -> 4 words as uint32
 *   LIBM            the C library calls of the path (global.hpp:238-341, IIntegrator.hpp:196-206, Material.hpp:217-245: sinf cosf
 *                   acosf tanf powf; Sphere.hpp:64: atan2f), restated from glibc 2.35 in csrc/device_libm.h
 *                                                                      x y -> sinf(x) cosf(x) acosf(x) tanf(x) powf(x, y) powf(x, 5) atan2f(x, y) atanf(x) */
enum TutuFn {
	TUTU_FN_BBOX = 0, TUTU_FN_TRI, TUTU_FN_NORMALIZED, TUTU_FN_FRESNEL, TUTU_FN_FRESNEL_SCHLICK, TUTU_FN_REFLECT, TUTU_FN_REFRACT,
	TUTU_FN_D, TUTU_FN_G, TUTU_FN_MIS, TUTU_FN_LOCAL2WORLD, TUTU_FN_RNG, TUTU_FN_PHILOX, TUTU_FN_LIBM, TUTU_FN_COUNT
};
int tutu_hip_eval_fn(TutuCtx* ctx, int32_t fn, uint32_t n, const float* const* in, float* out);

/* scene facts */
int tutu_hip_scene_info(TutuCtx* ctx, TutuBvhInfo* bvh, uint32_t* n_lights);

/* Post-processing of a finished linear-radiance frame (width*height*3 floats, row-major) on the device:
 * Postprocessor.hpp under HDR_BLOOM (global.hpp:32).  stage 0 = Postprocessor::performPostProcess (:29-57: emissive
 * extraction, Gaussian blur twice, + original, exposure tone map), 1 = getEmmisiveTexture (:128-155),
 * 2 = getGaussianBlurTexture(KERNELSIZE 10, STDDEV 30) (:62-125), 3 = getHDRtexture (:178-201).  Bit-exact up to exp(). */
int tutu_hip_postprocess(TutuCtx* ctx, int32_t stage, int32_t width, int32_t height, const float* rgb_in, float* rgb_out);
/* PPMGenerator::writePixel's channel mapping with GAMMA_COORECTION (PPMGenerator.hpp:825-843):
 * level = (int)(255 * pow(clamp(0, 1, c), 0.78f)) for each of the n values. */
int tutu_hip_quantise(TutuCtx* ctx, uint32_t n, const float* values, int32_t* levels);

/* Tuning / measuring knobs that do not change any result.
 *   "sets"  number of wavefront passes in flight (1..4, 0 = default).  With 1, kernels run one at a time and the
 *           per-kernel times of TutuStats are exclusive; with more, stages of different passes overlap and every
 *           launch's duration includes the time it shares the device with others.
 * The remaining knobs take their start values from TUTU_* environment variables ONCE, at tutu_hip_create, where a
 * value outside its range fails the create with TUTU_E_INVALID (it never reaches a kernel):
 *   "sets_default" TUTU_SETS [1,4] | "one_set" TUTU_ONE_SET {0,1} | "shade_bpc" TUTU_SHADE_BPC [0,16] (0 = one per CU for the one-class kernel while passes overlap, else all that fit) |
 *   "trace_bpc" TUTU_TRACE_BPC [0,8] (0 = from the LDS footprint) | "refill_min" TUTU_REFILL_MIN [1,64] |
 *   "inner_steps" TUTU_INNER_STEPS [1,64] / "inner_steps_any" TUTU_INNER_STEPS_ANY [1,64] (node visits per round of the closest-hit /
 *   any-hit kernel) | "leaf_again" TUTU_LEAF_AGAIN [1,65] (lanes still holding a leaf that trigger a second leaf step in a round;
 *   65 = never) | "trace_xcd" TUTU_TRACE_XCD {0,1} (default 1: the traversal blocks of one XCD take adjacent ranges of the ray
 *   list, so that each XCD's L2 serves one part of the picture) | "any_near_first" TUTU_ANY_NEAR_FIRST {0,1} | "kernel_events" TUTU_KERNEL_EVENTS {0,1} (default 0; 1 = a HIP
 *   event pair around every launch, which is what fills TutuStats' ms_* fields: ~2400 events, 2.7 % of a 512-spp Cornell frame;
 *   the ray / node counters of TutuStats do not need it) |
 *   "exact" TUTU_EXACT {0,1} (1: EVERY ray takes the exact walk -- the reference's own tree, the reference's own slab test, no distance
 *   pruning, BVH.hpp:145-194 visit for visit: the strict form, several times slower on big scenes; DESIGN.md section 4 says what the default
 *   walk assumes instead) | "cold_paths_mi" TUTU_COLD_PATHS_MI [0,4096] (see tutu_hip_work_ready; 0 = allocate everything in the first render) |
 *   "flat" TUTU_FLAT {0,1} (create-only; 1: a scene of at most 24 leaves is scanned flat instead of walked, device_shade.h: k_trace_flat; read-only
 *   fact "flat_leaves") | "flat_share" TUTU_FLAT_SHARE {0,1} (1: the closest-hit scan deals its (ray, leaf) pairs to the wave's lanes through LDS) |
 *   "util_stats" TUTU_UTIL_STATS {0,1} | "bidir_units" TUTU_BIDIR_UNITS [64, 2^24] ((pixel, sample) units per batch of
 *   tutu_hip_render_integrator; batches are whole pixels) |
 *   "wide" TUTU_WIDE [0,2] (memory-resident scenes: 0 walk the binary SAH tree, 1 the four-wide quantised tree when the binary
 *   nodes take at least "wide_min_mb" TUTU_WIDE_MIN_MB [0,65536] megabytes (default 0: always), 2 always) |
 *   "wide_inner_steps" TUTU_WIDE_INNER_STEPS [1,64] / "wide_inner_steps_any" TUTU_WIDE_INNER_STEPS_ANY [1,64] | "wide_lds_stack" TUTU_WIDE_LDS_STACK [4,64] and "lds_stack_max"
 *   TUTU_LDS_STACK_MAX [0,64]: entries of the traversal stack kept in LDS (wide / binary tree); deeper ones live in HBM.
 *   "wide8" TUTU_WIDE8 [0,2] (create-only; round 5: the EIGHT-wide quantised tree -- one stack entry per set of pending children, visiting
 *   order from per-node tables consumed with __ffs, leaves on a stack of their own; csrc/device_shade.h: trace_persistent8 -- 0 never, 1 for
 *   trees of fewer than "wide_early_max_mb" megabytes of four-wide nodes (default: it gains 0.5-2 % there and loses 10 % on the broom
 *   stand-in), 2 wherever the scene has one) with "wide8_inner_steps" / "wide8_inner_steps_any" TUTU_WIDE8_INNER_STEPS(_ANY) [1,64],
 *   "wide8_leaf_steps" TUTU_WIDE8_LEAF_STEPS [1,8], "wide8_leaf_again" TUTU_WIDE8_LEAF_AGAIN [1,65], "wide8_leaf_room" TUTU_WIDE8_LEAF_ROOM
 *   [1,32] (create-only); environment only (host tree build): TUTU_WIDE8_SLOTS = 2 kd slots + tables (default), 1 octant slots, 0 tree order.
 *   "gather_rccl" TUTU_GATHER_RCCL [0,2] (tutu_hip_render_multi(_device), option of the FIRST context: the pieces travel by one grouped
 *   RCCL send / recv: 0 never, 1 when the contexts sit on several devices, 2 always -- a context on the root's device by a self send / recv).
 *   Round 5, later: "exact_sum" TUTU_EXACT_SUM {0,1} (PathTracing: every path's radiance is folded from its deepest vertex back, the order in
 *   which the reference's recursive traceRay returns it (PathTracing.hpp:275-277, :133), instead of summed forward: frames and per-sample
 *   radiances are then the reference build's bit for bit; 224 B more per path slot, allocated on first use; 4-13 % of the frame rate) |
 *   "trace_deal" TUTU_TRACE_DEAL {-1, 0, 6..16} (persistent walks: log2 of the chunk of list positions dealt round-robin to the waves, 0 =
 *   one contiguous range per wave, -1 = auto: 6 for a pass that runs by itself, 0 when passes overlap) | "wide8_top" TUTU_WIDE8_TOP [0,592]
 *   (create-only: node ids of the eight-wide tree staged in LDS at most, default 40; read-only fact "wide8_top_nodes") | read-only
 *   "last_trace_us" (duration of the kernel behind the last tutu_hip_trace_closest / tutu_hip_trace_any call: measuring tools).
 *   Read-only facts: "wide8_tree", "wide8_depth", "wide8_nodes", "wide8_entries", "gather_path" (how the last N-context frame was gathered:
 *   0 copies, 1 RCCL, -1 none yet), "peer_access", "rccl_available",
 *   "wide_tree", "wide_depth", "fast_depth", "stack_entries", "stack_entries_hbm", "trace_blocks_per_cu",
 *   "trace_lds_bytes", "work_paths_mi", "growing", "grow_ms".
 *   The knobs tutu_hip_create consumes (tree choice, LDS carve-up: "wide", "wide_min_mb", "wide_early", "wide_early_max_mb", "lds_stack_max",
 *   "wide_lds_stack", "trace_bpc", "wide8", "wide8_leaf_room") are refused by tutu_hip_set_option afterwards (TUTU_E_INVALID) unless the value is the one in effect.
 *   Environment only (host tree build): TUTU_WIDE_COLLAPSE = 1 collapses the host-built wide tree greedily by surface area, 0 takes the
 *   grandchildren of every second binary level; unset: greedy when the walked tree holds more than 1.5 references per object, i.e. was built over
 *   clipped references (read-only fact "wide_greedy"; DESIGN.md section 6: +3 % on the broom stand-in, +2 % on the veach room, -3 % on the bunny
 *   stand-in, same hits).
 *   Testing aid, read at every device allocation rather than at create: TUTU_DEBUG_FILL=<0..255> fills every fresh allocation
 *   with that byte (no result may depend on what hipMalloc hands out; tests/test_hip_wide.py).
 * tutu_hip_get_option reports the effective value of any of them, plus the read-only facts "sah_tree", "lds_scene"
 * and "shade_tab" -- a benchmark line should echo them (bench.py does). */
int tutu_hip_set_option(TutuCtx* ctx, const char* name, int value);
int tutu_hip_get_option(TutuCtx* ctx, const char* name, int* value);

#ifdef __cplusplus
}
#endif
#endif /* TUTU_HIP_H */
