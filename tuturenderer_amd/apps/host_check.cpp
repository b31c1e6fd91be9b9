// Host-only check tool (no GPU call): parses a config with the front-end, loads the OBJ files named on the command
// line with the front-end's OBJ reader + loadObj, and dumps what the integrator would hand to the GPU library, so
// tests can compare it with the golden dumps of the reference's own loader.
//   host_check <config.txt> <out.bin> [obj ...]
// out.bin: int32 width,height,hfov,integrateType,n_tris ; float eye[3],viewdir[3],updir[3],bkg[3],eta ;
//          n_tris*9 float verts ; n_tris*9 float normals ; 256 x int32 quantise() of a 0..1 ramp ;
//          per triangle: 6 float uv, int32 textureIndex, normalMapIndex, roughnessMapIndex, metallicMapIndex, activated ;
//          for each of the 4 map lists: int32 count, then per map int32 width, height + width*height*3 float ;
//          int32 n_spheres, then per sphere: 4 float centre+radius, int32 object-list slot, 4 int32 map indices,
//          int32 activated, int32 material type, 3 float diffuse, float eta
// (n_tris and all per-triangle sections count the triangles only, in object-list order)
#include <cstdio>

#include "../host/tutu_renderer.hpp"

int main(int argc, char* argv[]) {
	if (argc < 3) return 2;
	PPMGenerator g(argv[1]);
	Material m;
	for (int i = 3; i < argc; i++) {
		objl::Loader l;
		if (l.LoadFile(argv[i])) g.loadObj(l, m);
	}
	FILE* f = fopen(argv[2], "wb");
	if (!f) return 3;
	int32_t n_tris = 0;
	for (auto& o : g.scene.objList) n_tris += o->objectType == TRIANGLE ? 1 : 0;
	const int32_t head[5] = {g.width, g.height, g.hfov, g.integrateType, n_tris};
	fwrite(head, sizeof(head), 1, f);
	const float cam[13] = {g.eyePos.x, g.eyePos.y, g.eyePos.z, g.viewdir.x, g.viewdir.y, g.viewdir.z, g.updir.x, g.updir.y, g.updir.z,
	                       g.bkgcolor.x, g.bkgcolor.y, g.bkgcolor.z, g.eta};
	fwrite(cam, sizeof(cam), 1, f);
	for (int pass = 0; pass < 2; pass++)
		for (auto& o : g.scene.objList) {
			if (o->objectType != TRIANGLE) continue;
			const Triangle* t = static_cast<const Triangle*>(o.get());
			const Vector3f* p[3] = {pass == 0 ? &t->v0 : &t->n0, pass == 0 ? &t->v1 : &t->n1, pass == 0 ? &t->v2 : &t->n2};
			for (int k = 0; k < 3; k++) fwrite(&p[k]->x, sizeof(float), 3, f);
		}
	for (int i = 0; i < 256; i++) {
		const int32_t q = (int32_t)PPMGenerator::quantise(-0.1f + 1.3f * (float)i / 255.f);
		fwrite(&q, sizeof(q), 1, f);
	}
	for (auto& o : g.scene.objList) {
		if (o->objectType != TRIANGLE) continue;
		const Triangle* t = static_cast<const Triangle*>(o.get());
		const float uv[6] = {t->uv0.x, t->uv0.y, t->uv1.x, t->uv1.y, t->uv2.x, t->uv2.y};
		fwrite(uv, sizeof(uv), 1, f);
		const int32_t ids[5] = {t->textureIndex, t->normalMapIndex, t->roughnessMapIndex, t->metallicMapIndex, t->isTextureActivated ? 1 : 0};
		fwrite(ids, sizeof(ids), 1, f);
	}
	const std::vector<Texture*>* lists[4] = {&g.diffuseMaps, &g.normalMaps, &g.roughnessMaps, &g.metallicMaps};
	for (int k = 0; k < 4; k++) {
		const int32_t cnt = (int32_t)lists[k]->size();
		fwrite(&cnt, sizeof(cnt), 1, f);
		for (const Texture* t : *lists[k]) {
			const int32_t wh[2] = {t->width, t->height};
			fwrite(wh, sizeof(wh), 1, f);
			for (const Vector3f& c : t->rgb) fwrite(&c.x, sizeof(float), 3, f);
		}
	}
	const int32_t n_sph = (int32_t)g.scene.objList.size() - n_tris;
	fwrite(&n_sph, sizeof(n_sph), 1, f);
	for (size_t slot = 0; slot < g.scene.objList.size(); slot++) {
		const Object* o = g.scene.objList[slot].get();
		if (o->objectType != SPEHRE) continue;
		const Sphere* sp = static_cast<const Sphere*>(o);
		const float c4[4] = {sp->centerPos.x, sp->centerPos.y, sp->centerPos.z, sp->radius};
		fwrite(c4, sizeof(c4), 1, f);
		const int32_t ids[7] = {(int32_t)slot, o->textureIndex, o->normalMapIndex, o->roughnessMapIndex, o->metallicMapIndex,
		                        o->isTextureActivated ? 1 : 0, (int32_t)o->mtlcolor.mType};
		fwrite(ids, sizeof(ids), 1, f);
		const float m4[4] = {o->mtlcolor.diffuse.x, o->mtlcolor.diffuse.y, o->mtlcolor.diffuse.z, o->mtlcolor.eta};
		fwrite(m4, sizeof(m4), 1, f);
	}
	fclose(f);
	return 0;
}
