// TEST INFRASTRUCTURE ONLY -- harness around the REFERENCE'S OWN CODE.
//
// Compiled by oracle/Makefile as
//   g++ -std=c++17 -O2 -ffp-contract=off -fopenmp -pthread -include oracle/ref_shim.h -I/root/reference/include ...
// into oracle/_ref/libtutu_ref.so (git-ignored; travels to the GPU box as a prebuilt binary).  Every tor_*
// function below calls the reference's own function of the same meaning; the harness adds no algorithm of its own
// except (a) the camera-frame lines of PathTracing::integrate (PathTracing.hpp:357-391), which cannot be called
// because the stock thread launcher hands a dangling pointer to its workers (PathTracing.hpp:397-409), and (b) the
// per-pixel/per-sample loop of sub_render_pt (PathTracing.hpp:499-515), re-driven here so that each sample gets its
// own counter-based RNG stream (see ref_shim.h).
//
// Nothing in this file is shipped or measured as the product.
#include "Renderer.hpp" // the reference's umbrella header (pulls PathTracing.hpp, BVH.hpp, Material.hpp, ...)
#include "OBJ_Loader.h"
#include "Sphere.hpp"
#include "Postprocessor.hpp"

#include "oracle_abi.h"

#include <atomic>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <thread>
#include <unistd.h>
#include <unordered_map>

// ------------------------------------------------------------------------------------------------- RNG plumbing
namespace tutu_ref {
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
	const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
	const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
	const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
	const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
	const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
	c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
static inline void philox4x32_10(const uint32_t ctr[4], uint32_t k0, uint32_t k1, uint32_t out[4]) {
	uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
	uint32_t k[2] = {k0, k1};
	for (int r = 0; r < 10; r++) {
		philox_round(c, k);
		k[0] += 0x9E3779B9u;
		k[1] += 0xBB67AE85u;
	}
	out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
RngState& rng_state() {
	thread_local RngState s;
	return s;
}
uint32_t next_u32() {
	RngState& s = rng_state();
	if (s.mode == 1) {
		float xi = 0.f;
		if (s.inj && s.inj_i < s.inj_n) xi = s.inj[s.inj_i];
		s.inj_i++;
		return ((uint32_t)(xi * 16777216.0f)) << 8;
	}
	uint32_t ctr[4] = {s.pix, s.smp, s.draw >> 2, 0u}, out[4];
	philox4x32_10(ctr, s.key0, s.key1, out);
	const uint32_t u = out[s.draw & 3u];
	s.draw++;
	return (u >> 8) << 8;
}
} // namespace tutu_ref

static inline Vector3f V(const float* p) { return Vector3f(p[0], p[1], p[2]); }
static inline void S(float* p, const Vector3f& v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

static Material to_material(const TorMaterial* m) {
	Material r;
	r.diffuse = V(m->diffuse);
	r.specular = V(m->specular);
	r.emission = V(m->emission);
	r.mType = (MaterialType)m->type;
	r.alpha = m->alpha;
	r.eta = m->eta;
	r.roughness = m->roughness;
	r.metallic = m->metallic;
	return r;
}

struct CoutSilencer {
	std::streambuf* old;
	std::ostringstream sink;
	CoutSilencer() { old = std::cout.rdbuf(sink.rdbuf()); }
	~CoutSilencer() { std::cout.rdbuf(old); }
};

// counts closest-hit queries per thread; delegates to the reference's BVHStrategy
static thread_local int g_nclosest = 0;
class CountingStrategy : public IIntersectStrategy {
public:
	BVHStrategy inner;
	void UpdateInter(Intersection& inter, Scene& sce, const Vector3f& o, const Vector3f& d) override {
		g_nclosest++;
		static_cast<IIntersectStrategy&>(inner).UpdateInter(inter, sce, o, d);
	}
	float getShadowCoeffi(Scene& sce, Intersection& p, Vector3f& lightPos) override {
		return static_cast<IIntersectStrategy&>(inner).getShadowCoeffi(sce, p, lightPos);
	}
};

static Texture* make_texture(const TorTexture* t, int list, int index) {
	Texture* tex = new Texture;  // owned by the PPMGenerator (deleteMaps, PPMGenerator.hpp:110-136)
	tex->name = "tex" + std::to_string(list) + "_" + std::to_string(index);
	tex->width = t->width;
	tex->height = t->height;
	tex->rgb.resize((size_t)t->width * t->height);
	for (size_t i = 0; i < tex->rgb.size(); i++) tex->rgb[i] = V(t->rgb + 3 * i);
	return tex;
}

struct RefScene {
	PPMGenerator* g = nullptr;
	PathTracing* pt = nullptr;
	CountingStrategy* strat = nullptr;
	std::unordered_map<Object*, int> index;
	Vector3f ul, delta_h, delta_v, c_off_h, c_off_v, eyePos;
};

static std::string write_tmp_config(const TorSceneDesc* d) {
	char path[] = "/tmp/tutu_ref_cfg_XXXXXX.txt";
	int fd = mkstemps(path, 4);
	if (fd < 0) return std::string();
	FILE* f = fdopen(fd, "w");
	// same keyword grammar as configs/config_cornellBox.txt; exact float values are patched in afterwards
	fprintf(f, "imsize %d %d\neye 0 0 0\nviewdir 0 0 1\nhfov %d\nupdir 0 1 0\nbkgcolor 0 0 0 1.0\nintegrator path",
	        d->width, d->height, d->hfov);
	fclose(f);
	return std::string(path);
}

static PPMGenerator* make_generator(const TorSceneDesc* d) {
	std::string cfg = write_tmp_config(d);
	if (cfg.empty()) return nullptr;
	PPMGenerator* g = new PPMGenerator(cfg.c_str());
	unlink(cfg.c_str());
	g->inputName = "tutu_ref_harness.txt";
	// patch in the exact float values, then redo the camera set-up lines of PPMGenerator::initialize
	// (PPMGenerator.hpp:296-304)
	g->eyePos = V(d->eye);
	g->viewdir = V(d->viewdir);
	g->updir = V(d->updir);
	g->bkgcolor = V(d->bkg);
	g->eta = d->eta;
	g->cam.width = g->width;
	g->cam.height = g->height;
	g->cam.hfov = g->hfov;
	g->cam.position = g->eyePos;
	g->cam.fwdDir = g->viewdir;
	g->cam.upDir = g->updir;
	g->cam.initialize(g->bkgcolor);
	return g;
}

// the camera-frame lines of PathTracing::integrate (PathTracing.hpp:357-391)
static void camera_frame(RefScene* s) {
	PPMGenerator* g = s->g;
	Camera& cam = g->cam;
	Vector3f u = crossProduct(cam.fwdDir, cam.upDir);
	u = normalized(u);
	Vector3f v = crossProduct(u, cam.fwdDir);
	v = normalized(v);
	float d = cam.imagePlaneDist;
	if (g->parallel_projection) d = 4.f;
	float width_half = fabs(tan(degree2Radians(cam.hfov / 2.f)) * d);
	float aspect_ratio = cam.width / (float)cam.height;
	float height_half = width_half / aspect_ratio;
	Vector3f n = normalized(g->viewdir);
	Vector3f eyePos = cam.position;
	Vector3f ul = eyePos + d * n - width_half * u + height_half * v;
	Vector3f ur = eyePos + d * n + width_half * u + height_half * v;
	Vector3f ll = eyePos + d * n - width_half * u - height_half * v;
	Vector3f delta_h = Vector3f(0, 0, 0);
	if (g->width != 1) delta_h = (ur - ul) / (g->width - 1);
	Vector3f delta_v = Vector3f(0, 0, 0);
	if (g->height != 1) delta_v = (ll - ul) / (g->height - 1);
	Vector3f c_off_h = (ur - ul) / (float)(g->width * 2);
	Vector3f c_off_v = (ll - ul) / (float)(g->height * 2);
	s->ul = ul; s->delta_h = delta_h; s->delta_v = delta_v;
	s->c_off_h = c_off_h; s->c_off_v = c_off_v; s->eyePos = eyePos;
}

// the pixel -> primary ray lines of sub_render_pt (PathTracing.hpp:502-504; note c_off_v twice, as there)
static inline Vector3f pixel_raydir(const RefScene* s, int x, int y) {
	Vector3f pixelPos = s->ul + x * s->delta_h + y * s->delta_v + s->c_off_v + s->c_off_v;
	return normalized((pixelPos - s->eyePos));
}

extern "C" {

const char* tor_kind(void) { return "reference"; }

int tor_philox4x32_10(int n, const uint32_t* ctr4, uint32_t key0, uint32_t key1, uint32_t* out4) {
	for (int i = 0; i < n; i++) tutu_ref::philox4x32_10(ctr4 + 4 * i, key0, key1, out4 + 4 * i);
	return 0;
}

int tor_rng_stream(uint32_t pix, uint32_t smp, uint32_t key0, uint32_t key1, int n, float* xi) {
	tutu_ref::RngState& s = tutu_ref::rng_state();
	s = tutu_ref::RngState();
	s.pix = pix; s.smp = smp; s.key0 = key0; s.key1 = key1;
	for (int i = 0; i < n; i++) xi[i] = getRandomFloat(); // global.hpp:182-199 through the swapped engine
	return 0;
}

int tor_bbox_intersect(int n, const float* pmin, const float* pmax, const float* o, const float* d, uint8_t* hit) {
	for (int i = 0; i < n; i++) {
		Vector3f a = V(pmin + 3 * i), b = V(pmax + 3 * i);
		BoundBox bb;
		bb.pMin = a;
		bb.pMax = b;
		hit[i] = bb.IntersectRay(V(o + 3 * i), V(d + 3 * i)) ? 1 : 0;
	}
	return 0;
}

static void fill_triangle(Triangle& t, const float* v9, const float* n9) {
	t.objectType = OBJTYPE::TRIANGLE;
	t.v0 = V(v9); t.v1 = V(v9 + 3); t.v2 = V(v9 + 6);
	t.n0 = V(n9); t.n1 = V(n9 + 3); t.n2 = V(n9 + 6);
}

int tor_tri_intersect(int n, const float* verts9, const float* normals9, const float* o, const float* d,
                      uint8_t* hit, float* t, float* pos, float* Ns, float* Ng) {
	for (int i = 0; i < n; i++) {
		Triangle tri;
		fill_triangle(tri, verts9 + 9 * i, normals9 + 9 * i);
		Intersection inter;
		bool h = tri.intersect(V(o + 3 * i), V(d + 3 * i), inter);
		hit[i] = h ? 1 : 0;
		t[i] = inter.t;
		S(pos + 3 * i, inter.pos);
		S(Ns + 3 * i, inter.Ns);
		S(Ng + 3 * i, inter.Ng);
	}
	return 0;
}

int tor_tri_area(int n, const float* verts9, float* area) {
	float zeros[9] = {0};
	for (int i = 0; i < n; i++) {
		Triangle tri;
		fill_triangle(tri, verts9 + 9 * i, zeros);
		area[i] = tri.getArea();
	}
	return 0;
}

int tor_math_normalized(int n, const float* v, float* out) {
	for (int i = 0; i < n; i++) S(out + 3 * i, normalized(V(v + 3 * i)));
	return 0;
}
int tor_math_fresnel(int n, const float* I, const float* N, const float* eta_i, const float* eta_t, float* out) {
	for (int i = 0; i < n; i++) out[i] = fresnel(V(I + 3 * i), V(N + 3 * i), eta_i[i], eta_t[i]);
	return 0;
}
int tor_math_fresnel_schlick(int n, const float* cos_theta, const float* F0, float* out3) {
	for (int i = 0; i < n; i++) S(out3 + 3 * i, fresnelSchlick(cos_theta[i], V(F0 + 3 * i)));
	return 0;
}
int tor_math_reflect(int n, const float* I, const float* N, float* out) {
	for (int i = 0; i < n; i++) S(out + 3 * i, getReflectionDir(V(I + 3 * i), V(N + 3 * i)));
	return 0;
}
int tor_math_refract(int n, const float* I, const float* N, const float* eta_i, const float* eta_t, float* out) {
	for (int i = 0; i < n; i++) S(out + 3 * i, getRefractionDir(V(I + 3 * i), V(N + 3 * i), eta_i[i], eta_t[i]));
	return 0;
}
int tor_math_D(int n, const float* h, const float* nrm, const float* rough, float* out) {
	for (int i = 0; i < n; i++) out[i] = D_ndf(V(h + 3 * i), V(nrm + 3 * i), rough[i]);
	return 0;
}
int tor_math_G(int n, const float* wi, const float* wo, const float* nrm, const float* rough, const float* h,
               float* out) {
	for (int i = 0; i < n; i++) out[i] = G_smf(V(wi + 3 * i), V(wo + 3 * i), V(nrm + 3 * i), rough[i], V(h + 3 * i));
	return 0;
}
int tor_math_mis(int n, const float* a, const float* b, float* out) {
	for (int i = 0; i < n; i++) out[i] = getMisWeight(a[i], b[i]);
	return 0;
}
int tor_math_local2world(int n, const float* N, const float* dir, float* out) {
	for (int i = 0; i < n; i++) S(out + 3 * i, SphereLocal2world(V(N + 3 * i), V(dir + 3 * i)));
	return 0;
}

// PPMGenerator::writePixel's per-channel mapping (PPMGenerator.hpp:825-843), GAMMA_COORECTION defined
int tor_write_pixel(int n, const float* c, int32_t* out) {
	for (int i = 0; i < n; i++) {
		Vector3f color(c[i], 0, 0);
		color.x = 255 * pow(clamp(0, 1, color.x), GAMMA_VAL);
		out[i] = (int)color.x;
	}
	return 0;
}

// Postprocessor.hpp, the functions performPostProcess calls under HDR_BLOOM (global.hpp:32)
int tor_postprocess(int stage, int width, int height, const float* in, float* out) {
	Texture src;
	src.width = width;
	src.height = height;
	src.rgb.resize((size_t)width * height);
	for (size_t i = 0; i < src.rgb.size(); i++) src.rgb[i] = V(in + 3 * i);
	Postprocessor p(&src);
	Texture res;
	switch (stage) {
	case 0: res = p.performPostProcess(); break;
	case 1: res = p.getEmmisiveTexture(&src); break;
	case 2: res = p.getGaussianBlurTexture(&src, KERNELSIZE, STDDEV); break;
	case 3: res = p.getHDRtexture(&src); break;
	default: return -1;
	}
	if ((size_t)res.width * res.height != src.rgb.size()) return -2;
	for (size_t i = 0; i < res.rgb.size(); i++) S(out + 3 * i, res.rgb[i]);
	return 0;
}

int tor_mat_bxdf(int n, const TorMaterial* m, const float* wi, const float* wo, const float* Ng, const float* Ns,
                 float eta_scene, const uint8_t* tir, float* out3) {
	for (int i = 0; i < n; i++) {
		Material mat = to_material(m);
		Vector3f r = mat.BxDF(V(wi + 3 * i), V(wo + 3 * i), V(Ng + 3 * i), V(Ns + 3 * i), eta_scene, false,
		                      tir ? tir[i] != 0 : false);
		S(out3 + 3 * i, r);
	}
	return 0;
}

int tor_mat_pdf(int n, const TorMaterial* m, const float* wi, const float* wo, const float* N, float eta_i,
                float eta_t, float* out) {
	Material mat = to_material(m);
	for (int i = 0; i < n; i++) out[i] = mat.pdf(V(wi + 3 * i), V(wo + 3 * i), V(N + 3 * i), eta_i, eta_t);
	return 0;
}

int tor_mat_sample(int n, const TorMaterial* m, const float* wo, const float* N, float eta_i, const float* xi3,
                   float* wi, uint8_t* ok, uint8_t* special, int32_t* ndraws) {
	tutu_ref::RngState& s = tutu_ref::rng_state();
	for (int i = 0; i < n; i++) {
		Material mat = to_material(m); // fresh copy: sampleDirection writes to `alpha` (Material.hpp:213)
		s = tutu_ref::RngState();
		s.mode = 1; s.inj = xi3 + 3 * i; s.inj_n = 3; s.inj_i = 0;
		Vector3f res(0.f);
		auto [success, sp] = mat.sampleDirection(V(wo + 3 * i), V(N + 3 * i), res, eta_i);
		S(wi + 3 * i, res);
		ok[i] = success ? 1 : 0;
		special[i] = sp ? 1 : 0;
		ndraws[i] = s.inj_i;
	}
	s = tutu_ref::RngState();
	return 0;
}

int tor_scene_create(const TorSceneDesc* d, void** out) {
	if (!d || !out) return -1;
	CoutSilencer quiet;
	RefScene* s = new RefScene();
	s->g = make_generator(d);
	if (!s->g) { delete s; return -2; }
	PPMGenerator* g = s->g;
	{
		std::vector<Texture*>* lists[4] = {&g->diffuseMaps, &g->normalMaps, &g->roughnessMaps, &g->metallicMaps};
		for (int k = 0; k < 4; k++)
			for (int i = 0; i < d->n_textures[k]; i++) lists[k]->push_back(make_texture(&d->textures[k][i], k, i));
	}
	// Scene::objList = the triangles with the spheres inserted at their object-list positions
	const int n_obj = d->n_tris + d->n_spheres;
	std::vector<int> sphere_at(n_obj, -1);
	for (int j = 0; j < d->n_spheres; j++) {
		const int pos = d->sphere_pos ? d->sphere_pos[j] : d->n_tris + j;
		if (pos < 0 || pos >= n_obj || sphere_at[pos] != -1) { delete s->g; delete s; return -1; }
		sphere_at[pos] = j;
	}
	int i = -1;  // running triangle index
	for (int slot = 0; slot < n_obj; slot++) {
		if (sphere_at[slot] >= 0) {
			// as PPMGenerator::readObject does for the `sphere` keyword (PPMGenerator.hpp:358-402)
			const int j = sphere_at[slot];
			std::unique_ptr<Sphere> sp = std::make_unique<Sphere>();
			sp->mtlcolor = to_material(&d->mats[d->sphere_mat_id[j]]);
			sp->centerPos = Vector3f(d->spheres[4 * j + 0], d->spheres[4 * j + 1], d->spheres[4 * j + 2]);
			sp->radius = d->spheres[4 * j + 3];
			sp->objectType = OBJTYPE::SPEHRE;
			if (d->sphere_tex_ids) {
				sp->textureIndex = d->sphere_tex_ids[4 * j + 0];
				sp->normalMapIndex = d->sphere_tex_ids[4 * j + 1];
				sp->roughnessMapIndex = d->sphere_tex_ids[4 * j + 2];
				sp->metallicMapIndex = d->sphere_tex_ids[4 * j + 3];
				if (sp->textureIndex != -1 || sp->normalMapIndex != -1 || sp->metallicMapIndex != -1 || sp->roughnessMapIndex != -1)
					sp->isTextureActivated = true;
			}
			sp->initializeBound();
			s->index[sp.get()] = slot;
			g->scene.add(std::move(sp));
			continue;
		}
		i++;
		// as PPMGenerator::loadObj does per triangle (PPMGenerator.hpp:170-203)
		std::unique_ptr<Triangle> t = std::make_unique<Triangle>();
		fill_triangle(*t, d->verts + 9 * i, d->normals + 9 * i);
		Material mat = to_material(&d->mats[d->mat_id[i]]);
		t->mtlcolor = mat;
		if (d->uvs) {
			t->uv0 = Vector2f(d->uvs[6 * i + 0], d->uvs[6 * i + 1]);
			t->uv1 = Vector2f(d->uvs[6 * i + 2], d->uvs[6 * i + 3]);
			t->uv2 = Vector2f(d->uvs[6 * i + 4], d->uvs[6 * i + 5]);
		}
		if (d->tex_ids) {  // as PPMGenerator::loadObj sets them (PPMGenerator.hpp:195-201)
			t->textureIndex = d->tex_ids[4 * i + 0];
			t->normalMapIndex = d->tex_ids[4 * i + 1];
			t->roughnessMapIndex = d->tex_ids[4 * i + 2];
			t->metallicMapIndex = d->tex_ids[4 * i + 3];
			if (t->textureIndex != -1 || t->normalMapIndex != -1 || t->metallicMapIndex != -1 || t->roughnessMapIndex != -1)
				t->isTextureActivated = true;
		}
		t->initializeBound();
		s->index[t.get()] = slot;
		g->scene.add(std::move(t));
	}
	g->scene.initializeBVH(); // Renderer.hpp:53
	g->initializeLights();    // Renderer.hpp:64
	s->strat = new CountingStrategy();
	s->pt = new PathTracing(g, s->strat);
	camera_frame(s);
	*out = s;
	return 0;
}

int tor_scene_destroy(void* h) {
	RefScene* s = (RefScene*)h;
	if (!s) return 0;
	delete s->pt;
	delete s->strat;
	delete s->g;
	delete s;
	return 0;
}

static void dump_node(RefScene* s, BVHNode* n, int cap, int32_t* count, float* bounds6, int32_t* leaf_tri) {
	if (!n) return;
	int i = (*count)++;
	bool leaf = !n->left && !n->right;
	if (i < cap) {
		S(bounds6 + 6 * i, n->bound.pMin);
		S(bounds6 + 6 * i + 3, n->bound.pMax);
		leaf_tri[i] = leaf ? s->index[n->obj] : -1;
	}
	dump_node(s, n->left, cap, count, bounds6, leaf_tri);
	dump_node(s, n->right, cap, count, bounds6, leaf_tri);
}

int tor_scene_bvh_dump(void* h, int cap, int32_t* n_nodes, float* bounds6, int32_t* leaf_tri) {
	RefScene* s = (RefScene*)h;
	*n_nodes = 0;
	dump_node(s, s->g->scene.BVHaccelerator->getNode(), cap, n_nodes, bounds6, leaf_tri);
	return 0;
}

int tor_scene_closest(void* h, int n, const float* o, const float* d, uint8_t* hit, float* t, int32_t* tri,
                      float* pos, float* Ns, float* Ng) {
	RefScene* s = (RefScene*)h;
	BVHNode* root = s->g->scene.BVHaccelerator->getNode();
	for (int i = 0; i < n; i++) {
		Intersection inter = getIntersection(root, V(o + 3 * i), V(d + 3 * i));
		hit[i] = inter.intersected ? 1 : 0;
		t[i] = inter.t;
		tri[i] = inter.intersected ? s->index[inter.obj] : -1;
		S(pos + 3 * i, inter.pos);
		S(Ns + 3 * i, inter.Ns);
		S(Ng + 3 * i, inter.Ng);
	}
	return 0;
}

int tor_scene_any(void* h, int n, const float* orig, const float* target, uint8_t* blocked) {
	RefScene* s = (RefScene*)h;
	for (int i = 0; i < n; i++) {
		Vector3f lp = V(target + 3 * i);
		blocked[i] = isShadowRayBlocked(V(orig + 3 * i), lp, s->g) ? 1 : 0;
	}
	return 0;
}

int tor_scene_lights(void* h, int cap, int32_t* n_lights, int32_t* tri) {
	RefScene* s = (RefScene*)h;
	*n_lights = (int)s->g->lightlist.size();
	for (int i = 0; i < *n_lights && i < cap; i++) tri[i] = s->index[s->g->lightlist[i]];
	return 0;
}

int tor_scene_sample_light(void* h, int n, const float* xi3, int32_t* tri, float* pos, float* nrm, float* pdf) {
	RefScene* s = (RefScene*)h;
	tutu_ref::RngState& r = tutu_ref::rng_state();
	for (int i = 0; i < n; i++) {
		r = tutu_ref::RngState();
		r.mode = 1; r.inj = xi3 + 3 * i; r.inj_n = 3; r.inj_i = 0;
		Intersection li;
		float p = 0.f;
		sampleLight(li, p, s->g);
		tri[i] = li.intersected ? s->index[li.obj] : -1;
		S(pos + 3 * i, li.pos);
		S(nrm + 3 * i, li.Ns);
		pdf[i] = p;
	}
	r = tutu_ref::RngState();
	return 0;
}

int tor_scene_light_pdf(void* h, int n, const int32_t* tri, float* pdf) {
	RefScene* s = (RefScene*)h;
	for (int i = 0; i < n; i++) {
		Intersection inter;
		inter.intersected = true;
		inter.obj = s->g->scene.objList[tri[i]].get();
		pdf[i] = getLightPdf(inter, s->g);
	}
	return 0;
}

int tor_camera(void* h, float* out18) {
	RefScene* s = (RefScene*)h;
	S(out18, s->ul); S(out18 + 3, s->delta_h); S(out18 + 6, s->delta_v);
	S(out18 + 9, s->c_off_h); S(out18 + 12, s->c_off_v); S(out18 + 15, s->eyePos);
	return 0;
}

int tor_camera_raster(void* h, float* out22) {  // the reference's own Camera after initialize()
	RefScene* s = (RefScene*)h;
	Camera& cam = s->g->cam;
	for (int i = 0; i < 16; i++) out22[i] = cam.world2Raster.ele[i];
	out22[16] = cam.imagePlaneDist; out22[17] = cam.filmPlaneAreaInv; out22[18] = cam.lensAreaInv;
	S(out22 + 19, cam.fwdDir);
	return 0;
}

int tor_camera_raydir(void* h, int n, const int32_t* px, const int32_t* py, float* d) {
	RefScene* s = (RefScene*)h;
	for (int i = 0; i < n; i++) S(d + 3 * i, pixel_raydir(s, px[i], py[i]));
	return 0;
}

int tor_trace_samples(void* h, int n, const uint32_t* pix, const uint32_t* smp, uint32_t key0, uint32_t key1,
                      float* L3, int32_t* ndraws, int32_t* nclosest) {
	RefScene* s = (RefScene*)h;
	tutu_ref::RngState& r = tutu_ref::rng_state();
	const int W = s->g->width;
	for (int i = 0; i < n; i++) {
		r = tutu_ref::RngState();
		r.pix = pix[i]; r.smp = smp[i]; r.key0 = key0; r.key1 = key1;
		g_nclosest = 0;
		Vector3f rayDir = pixel_raydir(s, (int)(pix[i] % W), (int)(pix[i] / W));
		Vector3f res = s->pt->traceRay(s->eyePos, rayDir, 0, Vector3f(1), nullptr, 0);
		S(L3 + 3 * i, res);
		if (ndraws) ndraws[i] = (int)r.draw;
		if (nclosest) nclosest[i] = g_nclosest;
	}
	return 0;
}

int tor_render(void* h, int spp, uint32_t key0, uint32_t key1, int x0, int y0, int x1, int y1, int nthreads,
               float* rgb) {
	RefScene* s = (RefScene*)h;
	if (spp <= 0 || nthreads <= 0) return -1;
	SPP = spp;             // global.hpp:19
	SPP_inv = 1.f / SPP;   // global.hpp:20
	const int W = s->g->width;
	std::atomic<int> next_row(y0);
	auto worker = [&](int tid) {
		tutu_ref::RngState& r = tutu_ref::rng_state();
		for (;;) {
			int y = next_row.fetch_add(1);
			if (y >= y1) break;
			for (int x = x0; x < x1; x++) {
				// PathTracing.hpp:501-513
				Vector3f rayDir = pixel_raydir(s, x, y);
				Vector3f estimate;
				for (int i = 0; i < SPP; i++) {
					r = tutu_ref::RngState();
					r.pix = (uint32_t)(y * W + x); r.smp = (uint32_t)i; r.key0 = key0; r.key1 = key1;
					Vector3f res = s->pt->traceRay(s->eyePos, rayDir, 0, Vector3f(1), nullptr, tid);
					if (!isnan(res.x) && !isnan(res.y) && !isnan(res.z)) estimate = estimate + res;
				}
				S(rgb + 3 * ((size_t)y * W + x), estimate * SPP_inv);
			}
		}
	};
	std::vector<std::thread> th;
	for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t);
	for (auto& t : th) t.join();
	return 0;
}

// ---- the other integrators behind the seam (Renderer.hpp:41-49), driven through the reference's OWN code:
// LightTracing::integrate and NaivePT::integrate are single-threaded and self-contained and are called as they are; for
// BDPT the worker sub_render_bdpt (BDPT.hpp:634-899, the live MULTITHREAD == 1 body) is called directly on all rows with a
// stable argument -- BDPT::integrate's thread launcher has the same dangling `&arg` as PathTracing's (BDPT.hpp:583-611) --
// after the camera-frame lines of BDPT::integrate (:390-418).  ONE sequential random stream for the whole frame.
int tor_render_integrator(void* h, int type, int spp, uint32_t key0, uint32_t key1, float* rgb) {
	RefScene* s = (RefScene*)h;
	if (type < 1 || type > 3 || spp <= 0) return -1;
	CoutSilencer quiet;
	PPMGenerator* g = s->g;
	SPP = spp;
	SPP_inv = 1.f / SPP;
	Camera& cam = g->cam;
	cam.FrameBuffer.rgb.assign((size_t)cam.width * cam.height, Vector3f(g->bkgcolor.x, g->bkgcolor.y, g->bkgcolor.z));  // Camera.hpp:26-27
	tutu_ref::RngState& r = tutu_ref::rng_state();
	r = tutu_ref::RngState();
	r.pix = 0xFFFFFFFFu; r.smp = (uint32_t)type; r.key0 = key0; r.key1 = key1;
	if (type == 1) {
		LightTracing lt(g, s->strat);
		lt.integrate(g);
	} else if (type == 2) {
		NaivePT np(g, s->strat);
		np.integrate(g);
	} else {
		BDPT bd(g, s->strat);
		Vector3f u = normalized(crossProduct(cam.fwdDir, cam.upDir));
		Vector3f v = normalized(crossProduct(u, cam.fwdDir));
		float d = cam.imagePlaneDist;
		float width_half = fabs(tan(degree2Radians(cam.hfov / 2.f)) * d);
		float aspect_ratio = cam.width / (float)cam.height;
		float height_half = width_half / aspect_ratio;
		Vector3f n = normalized(g->viewdir);
		Vector3f eyePos = cam.position;
		Vector3f ul = eyePos + d * n - width_half * u + height_half * v;
		Vector3f ur = eyePos + d * n + width_half * u + height_half * v;
		Vector3f ll = eyePos + d * n - width_half * u - height_half * v;
		Vector3f delta_h = Vector3f(0, 0, 0);
		if (g->width != 1) delta_h = (ur - ul) / (g->width - 1);
		Vector3f delta_v = Vector3f(0, 0, 0);
		if (g->height != 1) delta_v = (ll - ul) / (g->height - 1);
		Vector3f c_off_h = (ur - ul) / (float)(g->width * 2);
		Vector3f c_off_v = (ll - ul) / (float)(g->height * 2);
		Thread_arg_bdpt arg{&ul, &delta_v, &delta_h, &c_off_h, &c_off_v, &eyePos, g, &bd};
		sub_render_bdpt(&arg, 0, 0, g->height);
	}
	for (size_t i = 0; i < cam.FrameBuffer.rgb.size(); i++) S(rgb + 3 * i, cam.FrameBuffer.rgb[i]);
	return 0;
}

int tor_texture_lookup(const struct TorTexture* t, int n, const float* u, const float* v, float* rgb) {
	Texture* tex = make_texture(t, 9, 0);
	for (int i = 0; i < n; i++) S(rgb + 3 * i, tex->getRGBat(u[i], v[i]));
	delete tex;
	return 0;
}

// reference OBJ path: objl::Loader::LoadFile + PPMGenerator::loadObj, then read the triangles back
int tor_ref_load_obj(const char* path, int cap, int32_t* n_tris, float* verts9, float* normals9) {
	CoutSilencer quiet;
	TorSceneDesc d;
	memset(&d, 0, sizeof(d));
	d.width = 4; d.height = 4; d.hfov = 40;
	d.viewdir[2] = 1.f; d.updir[1] = 1.f; d.eta = 1.f;
	PPMGenerator* g = make_generator(&d);
	if (!g) return -2;
	objl::Loader loader;
	*n_tris = 0;
	int rc = 0;
	if (loader.LoadFile(path)) {
		Material m;
		g->loadObj(loader, m);
		*n_tris = (int)g->scene.objList.size();
		for (int i = 0; i < *n_tris && i < cap; i++) {
			Triangle* t = static_cast<Triangle*>(g->scene.objList[i].get());
			S(verts9 + 9 * i, t->v0); S(verts9 + 9 * i + 3, t->v1); S(verts9 + 9 * i + 6, t->v2);
			S(normals9 + 9 * i, t->n0); S(normals9 + 9 * i + 3, t->n1); S(normals9 + 9 * i + 6, t->n2);
		}
	} else {
		rc = -3;
	}
	g->scene.initializeBVH(); // Scene::~Scene deletes an otherwise uninitialised pointer (Scene.hpp:37-39)
	delete g;
	return rc;
}

} // extern "C"
