// Host-side front-end of the MI355X path tracer: the caller-visible API surface of bobhansky/TutuRenderer that the
// PathTracing path goes through, written from scratch over the C ABI of include/tutu_hip.h.
//
// A scene program written against the reference (src/main_cornellBox.cpp:20-90, src/main_veach_bdpt.cpp:21-105)
// compiles against this header with the same statements:
//     PPMGenerator g(argv[1]);  Material m; m.mType = LAMBERTIAN; ...  objl::Loader l; l.LoadFile(path);
//     g.loadObj(l, m);  Renderer r(&g);  r.render();  g.generate();
// What is kept: class and member names, argument meaning, defaults, error behaviour (exit(-1) on a bad config, like
// PPMGenerator.hpp:80-86,309-314).  What is different: there is no CPU integrator behind it.  Renderer installs
// HipPathTracing, whose integrate() hands the scene to the GPU library; without that library or without a GPU the
// program stops with an error -- it never falls back to a host path.
//
// Reference map (file:line in /root/reference/include):
//   Vector3f/Vector2f   Vector.hpp:49-225        Material          Material.hpp:9-56
//   Object/Triangle     Object.hpp:15-44, Triangle.hpp:8-17,104-116
//   Texture (framebuffer) Texture.hpp:10-16      Camera            Camera.hpp:12-49,81-96
//   Scene               Scene.hpp:11-41          PPMGenerator      PPMGenerator.hpp:32-160,164-324,488-868
//   IIntegrator         IIntegrator.hpp:17-24    Renderer          Renderer.hpp:32-72
//   globals SPP/SPP_inv global.hpp:19-20         Postprocessor     Postprocessor.hpp:16-28 (constructor only)
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tutu_hip.h"
#include "obj_loader.hpp"

// ---------------------------------------------------------------------------------------------- globals
// global.hpp:19-20.  `inline` so that several translation units may include this header.
inline int SPP = 64;
inline float SPP_inv = 1.f / SPP;
// Philox key of the render (new: the reference seeds mt19937 from random_device, global.hpp:188-195)
inline uint32_t TUTU_SEED0 = 0x5EED0001u;
inline uint32_t TUTU_SEED1 = 0u;
// samples per pixel traced per wavefront pass (0 = let the library choose)
inline int TUTU_SPP_PER_PASS = 0;

// ---------------------------------------------------------------------------------------------- vectors
struct Vector2f {
	float x = -1.f, y = -1.f;  // (-1,-1) = "no texture coordinate"
	Vector2f() = default;
	Vector2f(float a, float b) : x(a), y(b) {}
};

struct Vector3f {
	float x = 0.f, y = 0.f, z = 0.f;
	Vector3f() = default;
	Vector3f(float a, float b, float c) : x(a), y(b), z(c) {}
	Vector3f(float s) : x(s), y(s), z(s) {}  // implicit, as in the reference: `Vector3f a = 1;`
	Vector3f operator+(const Vector3f& o) const { return {x + o.x, y + o.y, z + o.z}; }
	Vector3f operator-(const Vector3f& o) const { return {x - o.x, y - o.y, z - o.z}; }
	Vector3f operator-() const { return {-x, -y, -z}; }
	Vector3f operator*(float c) const { return {x * c, y * c, z * c}; }
	Vector3f operator*(const Vector3f& o) const { return {x * o.x, y * o.y, z * o.z}; }
	Vector3f operator/(float c) const { return {x / c, y / c, z / c}; }
	float dot(const Vector3f& o) const { return x * o.x + y * o.y + z * o.z; }
	float norm() const { return std::sqrt(x * x + y * y + z * z); }
	float norm2() const { return x * x + y * y + z * z; }
};
inline Vector3f operator*(float c, const Vector3f& v) { return {v.x * c, v.y * c, v.z * c}; }
inline Vector3f normalized(const Vector3f& v) {
	const float len = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
	if (len > 0) {
		const float inv = 1 / len;
		return {v.x * inv, v.y * inv, v.z * inv};
	}
	return v;
}
inline Vector3f crossProduct(const Vector3f& a, const Vector3f& b) {
	return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// ---------------------------------------------------------------------------------------------- material
enum MaterialType { LAMBERTIAN, PERFECT_REFLECTIVE, PERFECT_REFRACTIVE, MICROFACET_R, MICROFACET_T, UNLIT };

// Plain data here: BxDF / sampleDirection / pdf run on the GPU (tuturenderer_amd/csrc/device_math.h).
class Material {
public:
	Vector3f diffuse = Vector3f(0.9f, 0.9f, 0.9f);
	Vector3f specular = Vector3f(1.f);
	Vector3f emission = Vector3f(0.f);
	MaterialType mType = LAMBERTIAN;
	float alpha = 1;
	float eta = 1;
	float roughness = 1;
	float metallic = 0;
	bool hasEmission() const { return emission.x || emission.y || emission.z; }
};

// ---------------------------------------------------------------------------------------------- primitives
enum OBJTYPE { TRIANGLE, SPEHRE };

struct BoundBox {
	Vector3f pMin, pMax;
};

class Object {
public:
	virtual ~Object() {}
	OBJTYPE objectType = TRIANGLE;
	Material mtlcolor;
	bool isTextureActivated = false;
	int textureIndex = -1, normalMapIndex = -1, roughnessMapIndex = -1, metallicMapIndex = -1;
	BoundBox bound;
	virtual void initializeBound() = 0;
	virtual float getArea() = 0;
};

class Triangle : public Object {
public:
	Vector3f v0, v1, v2;
	Vector3f n0, n1, n2;
	Vector2f uv0, uv1, uv2;
	void initializeBound() override {
		bound.pMin = {std::fmin(std::fmin(v0.x, v1.x), v2.x), std::fmin(std::fmin(v0.y, v1.y), v2.y), std::fmin(std::fmin(v0.z, v1.z), v2.z)};
		bound.pMax = {std::fmax(std::fmax(v0.x, v1.x), v2.x), std::fmax(std::fmax(v0.y, v1.y), v2.y), std::fmax(std::fmax(v0.z, v1.z), v2.z)};
	}
	float getArea() override { return crossProduct(v1 - v0, v2 - v0).norm() * 0.5f; }
};

class Sphere : public Object {  // Sphere.hpp:6-21, 133-142
public:
	Vector3f centerPos;
	float radius = 1.f;
	Sphere() { objectType = SPEHRE; }
	void initializeBound() override {
		bound.pMin = {centerPos.x - radius, centerPos.y - radius, centerPos.z - radius};
		bound.pMax = {centerPos.x + radius, centerPos.y + radius, centerPos.z + radius};
	}
	float getArea() override { return radius * radius * 3.1415926535897f; }  // [sic]
};

// ---------------------------------------------------------------------------------------------- framebuffer / camera
class Texture {
public:
	std::string name;
	int width = 0, height = 0;
	std::vector<Vector3f> rgb;
};

class Camera {
public:
	Vector3f fwdDir, upDir, rightDir, position;
	Texture FrameBuffer;
	int width = 0, height = 0, hfov = 0;
	float imagePlaneDist = 0.f;
	void initialize(Vector3f bkgcolor) {
		fwdDir = normalized(fwdDir);
		rightDir = normalized(crossProduct(fwdDir, upDir));
		upDir = normalized(crossProduct(rightDir, fwdDir));
		FrameBuffer.rgb.assign((size_t)width * height, bkgcolor);
		FrameBuffer.width = width;
		FrameBuffer.height = height;
		const float kPi = 3.1415926535897f;
		imagePlaneDist = width / (2.f * std::tan((hfov * 0.5f) * kPi / 180.f));
	}
};

// ---------------------------------------------------------------------------------------------- scene
class Scene {
public:
	std::vector<std::unique_ptr<Object>> objList;
	void add(std::unique_ptr<Object> obj) { objList.emplace_back(std::move(obj)); }
	// The acceleration structure is built by the GPU library when the integrator creates its context
	// (tutu_hip_create); kept so that Renderer's call sequence is the reference's (Renderer.hpp:53).
	void initializeBVH() { bvhRequested = true; }
	bool bvhRequested = false;
};

// ---------------------------------------------------------------------------------------------- front-end
class PPMGenerator {
public:
	std::ifstream fin;
	std::ofstream fout;
	std::string inputName;
	std::vector<Object*> lightlist;

	int width = -1, height = -1;
	Vector3f eyePos = Vector3f(FLT_MAX, 0, 0);
	Vector3f viewdir = Vector3f(FLT_MAX, 0, 0);
	int hfov = -1;
	Vector3f updir = Vector3f(FLT_MAX, 0, 0);
	Vector3f bkgcolor = Vector3f(FLT_MAX, 0, 0);
	float eta = 1.f;
	Scene scene;
	Camera cam;
	int integrateType = -1;  // 0 path, 1 light, 2 naivept, 3 bdpt
	int parallel_projection = 0;
	Material mtlcolor;  // current material of the inline `v/f` geometry
	std::vector<Vector3f> vertices, normals;
	std::vector<Vector2f> textCoords;
	// texture state of the config parser (PPMGenerator.hpp:36-39, 59-64)
	std::vector<Texture*> diffuseMaps, normalMaps, roughnessMaps, metallicMaps;
	bool isTextureOn = false;
	int textIndex = -1, bumpIndex = -1, roughnessIndex = -1, metallicIndex = -1;

	~PPMGenerator() {
		for (auto* lst : {&diffuseMaps, &normalMaps, &roughnessMaps, &metallicMaps})
			for (Texture* t : *lst) delete t;
	}
	PPMGenerator(const PPMGenerator&) = delete;
	PPMGenerator& operator=(const PPMGenerator&) = delete;

	// ASCII PPM ("P3 w h max r g b ...") -> Texture with channel/max floats; a name already in the list is not
	// loaded again (PPMGenerator.hpp:1027-1084)
	void loadTexture(const char* name, std::vector<Texture*>& textList) {
		for (Texture* t : textList)
			if (t->name == name) return;
		std::ifstream input(name, std::ios_base::in);
		if (!input.is_open()) {
			std::cout << "ERROR:: texture file does not exits, program terminates.\n";
			exit(-1);
		}
		std::string b0, b1, b2;
		input >> b0 >> b1 >> b2;
		if (b0 != "P3") {
			std::cout << "ERROR:: Need P3 keyword, program terminates.\n";
			exit(-1);
		}
		needPosInt(b1);
		needPosInt(b2);
		Texture* t = new Texture;
		t->name = name;
		t->width = std::stoi(b1);
		t->height = std::stoi(b2);
		input >> b0;
		const float max = (float)std::stoi(b0);
		t->rgb.reserve((size_t)t->width * t->height);
		for (int j = 0; j < t->height; j++)
			for (int i = 0; i < t->width; i++) {
				input >> b0 >> b1 >> b2;
				needPosInt(b0);
				needPosInt(b1);
				needPosInt(b2);
				t->rgb.emplace_back(Vector3f(std::stoi(b0) / max, std::stoi(b1) / max, std::stoi(b2) / max));
			}
		textList.emplace_back(t);
	}

	explicit PPMGenerator(const char* path) {
		fin.open(path, std::ios_base::in);
		if (!fin.is_open()) {
			std::cout << "ERROR:: inputfile does not exits, program terminates.\n";
			exit(-1);
		}
		inputName = path;
		initialize();
	}

	// OBJ_Loader meshes -> triangles, three consecutive vertices per triangle; indices are ignored, exactly like
	// the reference (PPMGenerator.hpp:164-208), so only pre-triangulated OBJ files work.
	void loadObj(objl::Loader& loader, Material& mtl, int textureIndex = -1, int bumpMapIndex = -1, int roughnessIndex = -1,
	             int metallicIndex = -1) {
		for (auto& m : loader.LoadedMeshes) {
			for (size_t i = 0; i + 2 < m.Vertices.size(); i += 3) {
				auto t = std::make_unique<Triangle>();
				const objl::Vertex &a = m.Vertices[i], &b = m.Vertices[i + 1], &c = m.Vertices[i + 2];
				t->v0 = {a.Position.X, a.Position.Y, a.Position.Z};
				t->v1 = {b.Position.X, b.Position.Y, b.Position.Z};
				t->v2 = {c.Position.X, c.Position.Y, c.Position.Z};
				t->n0 = {a.Normal.X, a.Normal.Y, a.Normal.Z};
				t->n1 = {b.Normal.X, b.Normal.Y, b.Normal.Z};
				t->n2 = {c.Normal.X, c.Normal.Y, c.Normal.Z};
				t->uv0 = {a.TextureCoordinate.X, a.TextureCoordinate.Y};
				t->uv1 = {b.TextureCoordinate.X, b.TextureCoordinate.Y};
				t->uv2 = {c.TextureCoordinate.X, c.TextureCoordinate.Y};
				t->mtlcolor = mtl;
				t->textureIndex = textureIndex;
				t->normalMapIndex = bumpMapIndex;
				t->roughnessMapIndex = roughnessIndex;
				t->metallicMapIndex = metallicIndex;
				if (textureIndex != -1 || bumpMapIndex != -1 || roughnessIndex != -1 || metallicIndex != -1) t->isTextureActivated = true;
				t->initializeBound();
				scene.add(std::move(t));
			}
			std::cout << "object loaded sucessfully\n";
		}
	}

	void transObj(objl::Loader& loader, float dx, float dy, float dz) {
		for (auto& m : loader.LoadedMeshes)
			for (auto& v : m.Vertices) {
				v.Position.X += dx;
				v.Position.Y += dy;
				v.Position.Z += dz;
			}
	}
	void scaleObj(objl::Loader& loader, float sx, float sy, float sz) {
		for (auto& m : loader.LoadedMeshes)
			for (auto& v : m.Vertices) {
				v.Position.X *= sx;
				v.Position.Y *= sy;
				v.Position.Z *= sz;
			}
	}
	// axis 0/1/2 = world x/y/z, angle in degrees (PPMGenerator.hpp:232-270)
	void rotateObj(objl::Loader& loader, int axis, float degree) {
		if (degree == 0) return;
		const float rad = degree * 3.1415926535897f / 180.f;
		const float c = std::cos(rad), s = std::sin(rad);
		auto rot = [&](float& X, float& Y, float& Z) {
			const float x = X, y = Y, z = Z;
			if (axis == 0) {
				Y = c * y - s * z;
				Z = s * y + c * z;
			} else if (axis == 1) {
				X = c * x + s * z;
				Z = -s * x + c * z;
			} else if (axis == 2) {
				X = c * x - s * y;
				Y = s * x + c * y;
			}
		};
		for (auto& m : loader.LoadedMeshes)
			for (auto& v : m.Vertices) {
				rot(v.Position.X, v.Position.Y, v.Position.Z);
				rot(v.Normal.X, v.Normal.Y, v.Normal.Z);
			}
	}

	void initializeLights() {
		for (auto& o : scene.objList)
			if (o->mtlcolor.hasEmission()) lightlist.push_back(o.get());
		std::cout << "light initialization complete \n";
	}

	size_t getIndex(int x, int y) const { return (size_t)y * width + x; }

	// ASCII P3, `<config minus .txt>.ppm`, 255*pow(clamp01(c), 0.78) truncated (PPMGenerator.hpp:140-160, 804-845)
	void generate() {
		std::string outName;
		const std::size_t pos = inputName.find(".txt");
		if (pos == std::string::npos) outName = inputName + ".ppm";
		else if (pos == 0) outName = ".ppm";
		else outName = inputName.substr(0, pos) + ".ppm";
		fout.open(outName);
		fout << "P3\n" << width << "\n" << height << "\n" << 255 << "\n";
		for (int y = 0; y < height; y++)
			for (int x = 0; x < width; x++) {
				Vector3f& c = cam.FrameBuffer.rgb[getIndex(x, y)];
				if (std::isinf(c.x)) std::cout << x << ", " << y << "is inf\n";
				else if (std::isnan(c.x)) std::cout << x << ", " << y << "is nan\n";
				c.x = quantise(c.x);
				c.y = quantise(c.y);
				c.z = quantise(c.z);
				fout << (int)c.x << " " << (int)c.y << " " << (int)c.z << "\n";
			}
		std::cout << "Generating image successfully.\n";
		fout.close();
	}
	static float quantise(float v) {
		const float lo = 0.f, hi = 1.f;
		const float cl = std::max(lo, std::min(hi, v));
		return 255 * std::pow(cl, 0.78f);
	}

private:
	std::string next() {
		if (fin.eof()) throw std::runtime_error("Insufficient or invalid data as input, check your config file\n");
		std::string s;
		fin >> s;
		return s;
	}
	// the reference's strict number grammar (global.hpp:74-126): digits with at most one inner '.', optional '-'
	static void needPosInt(const std::string& s) {
		for (char c : s)
			if (c < '0' || c > '9') throw std::runtime_error(s + ": expect a positive number");
	}
	static void needFloat(const std::string& s) {
		size_t i = 0;
		if (!s.empty() && s[0] == '-') i = 1;
		if (i >= s.size()) throw std::runtime_error(s + ": not a valid float number");
		bool dot = false;
		for (size_t k = i; k < s.size(); k++) {
			const char c = s[k];
			if (c == '.' && !dot && k != i && k != s.size() - 1) dot = true;
			else if (c < '0' || c > '9') throw std::runtime_error(s + ": not a valid float number");
		}
	}
	float number() {
		const std::string s = next();
		needFloat(s);
		return std::stof(s);
	}
	Vector3f triple() {
		const float a = number(), b = number(), c = number();
		return {a, b, c};
	}
	static int faceIndex(const std::string& tok, size_t from, size_t* end) {
		size_t k = from;
		while (k < tok.size() && tok[k] >= '0' && tok[k] <= '9') k++;
		if (k == from) throw std::runtime_error("f face information is not valid");
		*end = k;
		return std::stoi(tok.substr(from, k - from));
	}

	void keyword(const std::string& key) {
		if (key == "imsize") {
			const std::string a = next(), b = next();
			needPosInt(a);
			needPosInt(b);
			width = std::stoi(a);
			height = std::stoi(b);
		} else if (key == "eye") eyePos = triple();
		else if (key == "viewdir") viewdir = triple();
		else if (key == "hfov") {
			const std::string a = next();
			needPosInt(a);
			hfov = std::stoi(a);
		} else if (key == "updir") updir = triple();
		else if (key == "bkgcolor") {
			bkgcolor = triple();
			eta = number();  // 4th number = index of refraction of the scene
		} else if (key == "projection") {
			if (next() == "parallel") parallel_projection = 1;
		} else if (key == "light") {
			for (int i = 0; i < 7; i++) number();  // parsed and ignored, as in the reference (:559-568)
		} else if (key == "attlight") {
			for (int i = 0; i < 10; i++) number();
		} else if (key == "mtlcolor") {
			mtlcolor.diffuse = triple();
			mtlcolor.specular = triple();
			mtlcolor.alpha = number();
			mtlcolor.eta = number();
			isTextureOn = false;  // :608
		} else if (key == "MICROFACET_R" || key == "MICROFACET_T") {
			mtlcolor.mType = key == "MICROFACET_R" ? MICROFACET_R : MICROFACET_T;
			mtlcolor.diffuse = triple();
			mtlcolor.alpha = number();
			mtlcolor.eta = number();
			mtlcolor.roughness = number();
			mtlcolor.metallic = number();
		} else if (key == "PERFECT_REFLECTIVE") mtlcolor.mType = PERFECT_REFLECTIVE;
		else if (key == "PERFECT_REFRACTIVE") {
			mtlcolor.mType = PERFECT_REFRACTIVE;
			mtlcolor.eta = std::stof(next());
		} else if (key == "depthcueing") {
			for (int i = 0; i < 7; i++) number();
		} else if (key == "integrator") {
			const std::string a = next();
			if (a == "path") integrateType = 0;
			else if (a == "light") integrateType = 1;
			else if (a == "naivept") integrateType = 2;
			else if (a == "bdpt") integrateType = 3;
			else throw std::runtime_error("unknown integrator\n");
		} else if (key == "v") {
			const float a = std::stof(next()), b = std::stof(next()), c = std::stof(next());
			vertices.push_back({a, b, c});
		} else if (key == "vn") normals.push_back(normalized(triple()));
		else if (key == "vt") {
			const float a = number(), b = number();
			textCoords.push_back({a, b});
		} else if (key == "f") face();
		else if (key == "texture") textIndex = textureKeyword(diffuseMaps, false);
		else if (key == "bump") bumpIndex = textureKeyword(normalMaps, true);
		else if (key == "roughnessTexture") roughnessIndex = textureKeyword(roughnessMaps, false);
		else if (key == "metallicTexture") metallicIndex = textureKeyword(metallicMaps, false);
		else if (key == "sphere") sphere();
		else throw std::runtime_error("extraneous string in the input file\n");
	}

	// `texture` / `bump` / `roughnessTexture` / `metallicTexture` <file>  (PPMGenerator.hpp:669-765): load the map unless
	// a map of that name is already in the list, make it the active one; a freshly loaded normal map is mapped from
	// [0,1] to [-1,1] (:713-722)
	int textureKeyword(std::vector<Texture*>& list, bool isNormalMap) {
		const size_t size0 = list.size();
		const std::string a = next();
		loadTexture(a.c_str(), list);
		isTextureOn = true;
		if (list.size() == size0) {
			for (size_t i = 0; i < list.size(); i++)
				if (list[i]->name == a) return (int)i;
			return -1;
		}
		const int idx = (int)list.size() - 1;
		if (isNormalMap)
			for (Vector3f& c : list[(size_t)idx]->rgb) {
				c = c * 2.f;
				c.x = c.x - 1.f;
				c.y = c.y - 1.f;
				c.z = c.z - 1.f;
			}
		return idx;
	}

	// sphere cx cy cz r  (PPMGenerator.hpp:358-402): takes the current material, the active albedo map and, once each, the
	// pending normal / roughness / metallic maps
	void sphere() {
		auto s = std::make_unique<Sphere>();
		s->mtlcolor = mtlcolor;
		s->centerPos = triple();
		s->radius = number();
		s->objectType = SPEHRE;
		if (isTextureOn) {
			s->isTextureActivated = true;
			s->textureIndex = textIndex;
			if (bumpIndex != -1) {
				s->normalMapIndex = bumpIndex;
				bumpIndex = -1;
			}
			if (roughnessIndex != -1) {
				s->roughnessMapIndex = roughnessIndex;
				roughnessIndex = -1;
			}
			if (metallicIndex != -1) {
				s->metallicMapIndex = metallicIndex;
				metallicIndex = -1;
			}
		}
		s->initializeBound();
		scene.add(std::move(s));
	}

	// f a b c | a//n .. | a/t .. | a/t/n ..   (PPMGenerator.hpp:404-452, 879-1023)
	void face() {
		const std::string tok[3] = {next(), next(), next()};
		Triangle t;
		t.mtlcolor = mtlcolor;
		Vector3f* pv[3] = {&t.v0, &t.v1, &t.v2};
		Vector3f* pn[3] = {&t.n0, &t.n1, &t.n2};
		Vector2f* pt[3] = {&t.uv0, &t.uv1, &t.uv2};
		bool haveN = false;
		for (int k = 0; k < 3; k++) {
			size_t e = 0;
			const int iv = faceIndex(tok[k], 0, &e);
			*pv[k] = at(vertices, iv - 1);
			if (e == tok[k].size()) continue;
			if (tok[k][e] != '/') throw std::runtime_error("f face information is not valid");
			if (e + 1 < tok[k].size() && tok[k][e + 1] == '/') {
				size_t e2 = 0;
				*pn[k] = at(normals, faceIndex(tok[k], e + 2, &e2) - 1);
				haveN = true;
			} else {
				size_t e2 = 0;
				*pt[k] = at(textCoords, faceIndex(tok[k], e + 1, &e2) - 1);
				if (e2 < tok[k].size()) {
					size_t e3 = 0;
					*pn[k] = at(normals, faceIndex(tok[k], e2 + 1, &e3) - 1);
					haveN = true;
				}
			}
		}
		if (!haveN) {
			const Vector3f n = normalized(crossProduct(t.v1 - t.v0, t.v2 - t.v0));
			t.n0 = t.n1 = t.n2 = n;
		}
		auto s = std::make_unique<Triangle>(t);
		if (isTextureOn) {  // :439-448 -- a face takes the active albedo map and, once, the pending normal map
			s->isTextureActivated = true;
			s->textureIndex = textIndex;
			if (bumpIndex != -1) {
				s->normalMapIndex = bumpIndex;
				bumpIndex = -1;
			}
		}
		s->objectType = TRIANGLE;
		s->initializeBound();
		scene.add(std::move(s));
	}
	template <typename T>
	static T at(const std::vector<T>& v, int i) {
		if (i < 0 || (size_t)i >= v.size()) throw std::runtime_error("index is out of bound\n");
		return v[(size_t)i];
	}

	static bool nearlyEqual(float a, float b) { return std::fabs(a - b) < 0.0001f; }

	void initialize() {
		try {
			// same loop shape as PPMGenerator::initialize (PPMGenerator.hpp:282-286): a keyword that is the very
			// last token of a file without trailing newline is not processed there either
			std::string key;
			fin >> key;
			while (!fin.eof()) {
				keyword(key);
				fin >> key;
			}
			if (width == -1 || height == -1 || eyePos.x == FLT_MAX || viewdir.x == FLT_MAX || hfov == -1 || updir.x == FLT_MAX ||
			    bkgcolor.x == FLT_MAX || integrateType == -1)
				throw std::runtime_error("insufficient input data: unable to start the program\n");
			if (nearlyEqual(viewdir.x, updir.x) && nearlyEqual(viewdir.y, updir.y) && nearlyEqual(viewdir.z, updir.z))
				throw std::runtime_error("invalid viewPlane infomation: updir and view dir can't be the same");
			cam.width = width;
			cam.height = height;
			cam.hfov = hfov;
			cam.position = eyePos;
			cam.fwdDir = viewdir;
			cam.upDir = updir;
			cam.initialize(bkgcolor);
		} catch (const std::exception& e) {
			std::cout << "ERROR: " << e.what();
			exit(-1);
		}
	}
};

// ---------------------------------------------------------------------------------------------- integrators
class IIntersectStrategy {
public:
	virtual ~IIntersectStrategy() {}
};

class IIntegrator {
public:
	virtual ~IIntegrator() {}
	virtual void integrate(PPMGenerator* g) = 0;
	IIntersectStrategy* interStrategy = nullptr;
	PPMGenerator* g = nullptr;
};

// IIntegrator for all four `integrator` keywords: PathTracing::integrate (PathTracing.hpp:352-516) -- or LightTracing / NaivePT / BDPT -- as one call
// into the GPU library -- the same binding file a maintainer adds to the reference itself.
#define TUTU_BINDING_KNOBS  // TUTU_SEED0 / TUTU_SEED1 / TUTU_SPP_PER_PASS are declared at the top of this header
inline int TUTU_GPUS = 0;
#include "../integration/HipPathTracing.hpp"

// ---------------------------------------------------------------------------------------------- Renderer
class Renderer {
public:
	explicit Renderer(PPMGenerator* ppmg) : g(ppmg) {
		interStrategy = new IIntersectStrategy();  // traversal lives on the device; kept for the member's sake
		// Renderer.hpp:41-49: path / light / naivept / bdpt -- all four run on the device
		integrator = new HipPathTracing(g, interStrategy, g->integrateType);
		g->scene.initializeBVH();
	}
	~Renderer() {
		delete integrator;
		delete interStrategy;
	}
	void render() {
		g->initializeLights();
		integrator->integrate(g);
	}
	PPMGenerator* g;
	IIntersectStrategy* interStrategy = nullptr;
	IIntegrator* integrator = nullptr;
};

// Constructed by every reference main; the performPostProcess() call is commented out there
// (src/main_cornellBox.cpp:84).  Under HDR_BLOOM (global.hpp:32) it is bloom + exposure tone mapping
// (Postprocessor.hpp:29-57); here the four passes run on the GPU (tutu_hip_postprocess).
class Postprocessor {
public:
	Texture* source;
	explicit Postprocessor(Texture* src) : source(src) {}
	Texture performPostProcess() {
		Texture out;
		out.width = source->width;
		out.height = source->height;
		out.rgb.resize(source->rgb.size());
		if (out.rgb.empty()) return out;
		TutuSceneDesc sd;
		std::memset(&sd, 0, sizeof(sd));  // an empty scene: the context only provides the device and its stream
		TutuCtx* ctx = nullptr;
		int rc = tutu_hip_create(&sd, 0, &ctx);
		if (rc == TUTU_OK) rc = tutu_hip_postprocess(ctx, 0, source->width, source->height, &source->rgb[0].x, &out.rgb[0].x);
		if (rc != TUTU_OK) {
			std::cout << "ERROR: post-processing: " << tutu_hip_error_string(rc) << " " << tutu_hip_last_error() << "\n";
			exit(-1);
		}
		tutu_hip_destroy(ctx);
		return out;
	}
};
