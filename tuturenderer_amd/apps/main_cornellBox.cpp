// Scene program: Cornell box through the GPU path tracer -- the same statements as the reference's
// src/main_cornellBox.cpp:13-92, compiled against tuturenderer_amd/host/tutu_renderer.hpp instead of the
// reference's headers.   usage: main_cornellBox <config.txt> [spp] [model dir]
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
	const std::string dir = argc > 3 ? std::string(argv[3]) : std::string("../model/cornellBox");

	PPMGenerator g(argv[1]);

	Material floorMtl;
	floorMtl.mType = LAMBERTIAN;
	floorMtl.diffuse = {0.725f, 0.71f, 0.68f};
	objl::Loader floor;
	if (floor.LoadFile(dir + "/floor.obj")) g.loadObj(floor, floorMtl, -1, -1);

	Material lightMtl;
	lightMtl.diffuse = {0.725f, 0.71f, 0.68f};
	lightMtl.emission = {47.8348007f, 38.5663986f, 31.0807991f};
	objl::Loader light;
	if (light.LoadFile(dir + "/light.obj")) g.loadObj(light, lightMtl, -1, -1);

	Material green;
	green.mType = LAMBERTIAN;
	green.diffuse = {0.14f, 0.45f, 0.091f};
	objl::Loader right;
	if (right.LoadFile(dir + "/right.obj")) g.loadObj(right, green, -1, -1);

	Material red;
	red.mType = LAMBERTIAN;
	red.diffuse = {0.63f, 0.065f, 0.05f};
	objl::Loader left;
	if (left.LoadFile(dir + "/left.obj")) g.loadObj(left, red, -1, -1);

	Material white;
	white.mType = LAMBERTIAN;
	white.diffuse = {0.725f, 0.71f, 0.68f};
	objl::Loader tall;
	if (tall.LoadFile(dir + "/tallbox.obj")) g.loadObj(tall, white, -1, -1);
	objl::Loader shortb;
	if (shortb.LoadFile(dir + "/shortbox.obj")) g.loadObj(shortb, white, -1, -1);

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
