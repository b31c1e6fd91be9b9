// Scene program: the Veach-BDPT room -- the statements of the reference's src/main_veach_bdpt.cpp:14-107 against
// tuturenderer_amd/host/tutu_renderer.hpp.  With `integrator path` in the config it is BASELINE config 5; `integrator bdpt`
// (what the reference runs it with), `light` and `naivept` select the device versions of those integrators.
// usage: main_veach <config.txt> [spp] [model dir]
// The reference asks for "veach_slight.obj" while the file is veach_sLight.obj, so on a case-sensitive file system
// the small light is silently skipped (src/main_veach_bdpt.cpp:49); the same file name is asked for here.
#include <chrono>
#include <string>

#include "../host/tutu_renderer.hpp"

int main(int argc, char* argv[]) {
	if (argc < 2) {
		std::cout << "ERROR: lack of the input configuration file, please provide its path as the first argument.\n";
		return 0;
	}
	if (argc > 2) {
		SPP = std::atoi(argv[2]);
		SPP_inv = 1.f / SPP;
	}
	const std::string dir = argc > 3 ? std::string(argv[3]) : std::string("../model/veach_bdpt");

	PPMGenerator g(argv[1]);

	Material roomMtl;
	roomMtl.mType = LAMBERTIAN;
	roomMtl.diffuse = {0.725f, 0.71f, 0.68f};
	objl::Loader room;
	if (room.LoadFile(dir + "/veach_room.obj")) g.loadObj(room, roomMtl);

	Material LlightMtl;
	LlightMtl.diffuse = {0.725f, 0.71f, 0.68f};
	LlightMtl.emission = {500.0f, 500.0f, 500.0f};
	LlightMtl.emission = LlightMtl.emission * 0.5f;
	objl::Loader Llight;
	if (Llight.LoadFile(dir + "/veach_Llight.obj")) g.loadObj(Llight, LlightMtl);

	Material sLlightMtl;
	sLlightMtl.diffuse = {0.725f, 0.71f, 0.68f};
	sLlightMtl.emission = {6999.999881f, 5450.000167f, 3630.000055f};
	sLlightMtl.emission = sLlightMtl.emission * 0.5f;
	objl::Loader slight;
	if (slight.LoadFile(dir + "/veach_slight.obj")) g.loadObj(slight, sLlightMtl);

	Material tableMtl;
	tableMtl.mType = LAMBERTIAN;
	tableMtl.diffuse = {0.32962962985f, 0.257976263762f, 0.150291711092f};
	objl::Loader table;
	if (table.LoadFile(dir + "/veach_table.obj")) g.loadObj(table, tableMtl);

	Material glassMtl;
	glassMtl.mType = PERFECT_REFRACTIVE;
	glassMtl.eta = 1.5f;
	objl::Loader glass;
	if (glass.LoadFile(dir + "/veach_glass.obj")) g.loadObj(glass, glassMtl);

	Material tallLampMtl;
	tallLampMtl.mType = MICROFACET_R;
	tallLampMtl.roughness = 0.2775146484375f;
	tallLampMtl.metallic = 0.5f;
	tallLampMtl.diffuse = {0.32962962985f, 0.257976263762f, 0.150291711092f};
	objl::Loader tallLamp;
	if (tallLamp.LoadFile(dir + "/veach_tallLamp.obj")) g.loadObj(tallLamp, tallLampMtl);

	objl::Loader wallLamp;
	if (wallLamp.LoadFile(dir + "/veach_wallLamp.obj")) g.loadObj(wallLamp, roomMtl);

	Renderer r(&g);
	auto start = std::chrono::steady_clock::now();
	r.render();
	auto end = std::chrono::steady_clock::now();
	const double sec = std::chrono::duration<double>(end - start).count();
	std::cout << "\nRendering Time consumed: \n" << sec << " seconds ("
	          << (double)g.width * g.height * SPP / sec / 1e6 << " Msamples/s, " << g.scene.objList.size() << " triangles, " << SPP
	          << " spp)\n";
	Postprocessor p(&g.cam.FrameBuffer);
	std::cout << "output to img...\n";
	g.generate();
	return 0;
}
