// TEST INFRASTRUCTURE ONLY -- boundary proof (VERDICT r1 #4/#6a).
//
// The reference's OWN front-end (PPMGenerator config parser, objl::Loader, PPMGenerator::loadObj, the P3 writer;
// compiled from /root/reference/include where the sources lie) with tuturenderer_amd/integration/HipPathTracing.hpp --
// the file a maintainer adds to the reference -- as its integrator.  oracle/Makefile builds it into
// oracle/_ref/ref_binding (git-ignored; it travels to the GPU box like the other _ref binaries).
//
//   ref_binding <config.txt> [--dry] [--spp N] [--key1 K] <mesh spec> ...
//   mesh spec = path|type|dr,dg,db|er,eg,eb|eta|roughness|metallic      (type = MaterialType number)
// --dry stops before the first device call and prints what the binding would hand to tutu_hip_create (counts, FNV hash
// of the flattened scene, the camera frame): a GPU-less check that the binding compiles against the reference's real
// members and that the reference's loader and ours produce the same bytes.  Without --dry the frame is rendered on the
// GPU and written by the reference's own PPMGenerator::generate() next to the config.
#include <cstdio>
#include <cstdlib>
#include <string>

#ifdef TUTU_BUNDLED_FRONTEND
// the SAME program over the bundled front-end (oracle/bundled_binding): what the reference build's outputs are compared with
#include "../tuturenderer_amd/host/tutu_renderer.hpp"
#else
#include "PPMGenerator.hpp"
#include "IIntegrator.hpp"
#include "OBJ_Loader.h"
// the binding under test
#include "HipPathTracing.hpp"
#endif

static std::vector<std::string> split(const std::string& s, char sep) {
	std::vector<std::string> out;
	size_t a = 0;
	for (;;) {
		const size_t b = s.find(sep, a);
		out.push_back(s.substr(a, b == std::string::npos ? b : b - a));
		if (b == std::string::npos) break;
		a = b + 1;
	}
	return out;
}

int main(int argc, char* argv[]) {
	if (argc < 2) return 2;
	PPMGenerator g(argv[1]);
	bool dry = false;
	int spp = 16;
	for (int i = 2; i < argc; i++) {
		const std::string a = argv[i];
		if (a == "--dry") dry = true;
		else if (a == "--spp" && i + 1 < argc) spp = atoi(argv[++i]);
		else if (a == "--key1" && i + 1 < argc) TUTU_SEED1 = (uint32_t)atoi(argv[++i]);
		else {
			const std::vector<std::string> p = split(a, '|');
			if (p.size() != 7) {
				printf("bad mesh spec %s\n", a.c_str());
				return 2;
			}
			Material m;
			m.mType = (MaterialType)atoi(p[1].c_str());
			const std::vector<std::string> d = split(p[2], ','), e = split(p[3], ',');
			m.diffuse = Vector3f((float)atof(d[0].c_str()), (float)atof(d[1].c_str()), (float)atof(d[2].c_str()));
			m.emission = Vector3f((float)atof(e[0].c_str()), (float)atof(e[1].c_str()), (float)atof(e[2].c_str()));
			m.eta = (float)atof(p[4].c_str());
			m.roughness = (float)atof(p[5].c_str());
			m.metallic = (float)atof(p[6].c_str());
			objl::Loader l;
			if (l.LoadFile(p[0])) g.loadObj(l, m, -1, -1);  // as src/main_cornellBox.cpp:27-30
		}
	}
	SPP = spp;
	SPP_inv = 1.f / SPP;
	g.scene.initializeBVH();  // Renderer::Renderer, Renderer.hpp:53 (also: Scene::~Scene deletes the accel pointer, Scene.hpp:37-39)
	g.initializeLights();     // Renderer::render, Renderer.hpp:64
	HipPathTracing integ(&g, nullptr, g.integrateType);  // the `integrator` keyword of the config (Renderer.hpp:41-49)
	if (dry) {
		HipPathTracing::FlatScene f;
		HipPathTracing::flatten(&g, f);
		TutuCameraFrame cf;
		if (tutu_camera_frame(&f.cd, &cf) != TUTU_OK) return 3;
		printf("tris %u mats %u spheres %u lights %zu hash %016llx\n", f.sd.n_tris, f.sd.n_mats, f.sps.n_spheres, g.lightlist.size(),
		       (unsigned long long)f.hash);
		printf("frame %d %d", cf.width, cf.height);
		const float* v[6] = {cf.ul, cf.delta_h, cf.delta_v, cf.c_off_h, cf.c_off_v, cf.eye};
		for (int k = 0; k < 6; k++)
			for (int j = 0; j < 3; j++) printf(" %a", v[k][j]);
		printf("\n");
		return 0;
	}
	IIntegrator* integrator = &integ;  // through the reference's own interface, as Renderer::render calls it (Renderer.hpp:65)
	integrator->integrate(&g);
	g.generate();  // the reference's own P3 writer (PPMGenerator.hpp:140-160)
	return 0;
}
