// Minimal Wavefront OBJ reader with the interface the scene programs use from the reference's vendored
// OBJ_Loader.h (objl::Loader::LoadFile, LoadedMeshes[i].Vertices[j].{Position,Normal,TextureCoordinate}).
// Written from scratch; it reproduces the two behaviours of that loader the renderer depends on:
//   * vertex order: every `f` line appends its corner vertices, in order, to the current mesh
//     (OBJ_Loader.h:580-590) -- PPMGenerator::loadObj then takes them three at a time;
//   * faces without `vn`: all corners get the UN-normalised face normal (v1-v0) x (v2-v1), the author-modified
//     orientation of OBJ_Loader.h:818-836.
// Meshes are split at `o` / `g` / `usemtl` lines once they have faces (OBJ_Loader.h:484-527, 612-641);
// .mtl files are not read (the renderer ignores them: materials come from the scene program).
#pragma once
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace objl {

struct Vector2 {
	float X = 0.f, Y = 0.f;
};
struct Vector3 {
	float X = 0.f, Y = 0.f, Z = 0.f;
};
struct Vertex {
	Vector3 Position;
	Vector3 Normal;
	Vector2 TextureCoordinate;
};
struct Mesh {
	std::string MeshName;
	std::vector<Vertex> Vertices;
	std::vector<unsigned int> Indices;
};

class Loader {
public:
	std::vector<Mesh> LoadedMeshes;
	std::vector<Vertex> LoadedVertices;
	std::vector<unsigned int> LoadedIndices;

	bool LoadFile(std::string Path) {
		if (Path.size() < 4 || Path.substr(Path.size() - 4) != ".obj") return false;
		std::ifstream file(Path);
		if (!file.is_open()) return false;
		LoadedMeshes.clear();
		LoadedVertices.clear();
		LoadedIndices.clear();
		std::vector<Vector3> P, N;
		std::vector<Vector2> T;
		Mesh cur;
		auto flush = [&](const std::string& nextName) {
			if (!cur.Vertices.empty()) LoadedMeshes.push_back(cur);
			cur = Mesh();
			cur.MeshName = nextName;
		};
		std::string line;
		while (std::getline(file, line)) {
			std::istringstream ls(line);
			std::string tag;
			if (!(ls >> tag)) continue;
			if (tag == "v") {
				Vector3 p;
				ls >> p.X >> p.Y >> p.Z;
				P.push_back(p);
			} else if (tag == "vt") {
				Vector2 t;
				ls >> t.X >> t.Y;
				T.push_back(t);
			} else if (tag == "vn") {
				Vector3 n;
				ls >> n.X >> n.Y >> n.Z;
				N.push_back(n);
			} else if (tag == "o" || tag == "g" || tag == "usemtl") {
				std::string name;
				std::getline(ls, name);
				flush(name);
			} else if (tag == "f") {
				std::vector<Vertex> corners;
				bool noNormal = false;
				std::string tok;
				while (ls >> tok) {
					Vertex v;
					int idx[3] = {0, 0, 0};
					bool has[3] = {false, false, false};
					size_t start = 0;
					for (int k = 0; k < 3 && start <= tok.size(); k++) {
						size_t slash = tok.find('/', start);
						std::string part = tok.substr(start, slash == std::string::npos ? std::string::npos : slash - start);
						if (!part.empty()) {
							idx[k] = std::stoi(part);
							has[k] = true;
						}
						if (slash == std::string::npos) break;
						start = slash + 1;
					}
					if (!has[0]) continue;
					v.Position = pick(P, idx[0]);
					if (has[1]) v.TextureCoordinate = pick(T, idx[1]);
					if (has[2]) v.Normal = pick(N, idx[2]);
					else noNormal = true;
					corners.push_back(v);
				}
				if (noNormal && corners.size() >= 3) {
					const Vector3 a = sub(corners[1].Position, corners[0].Position);
					const Vector3 b = sub(corners[2].Position, corners[1].Position);
					Vector3 n;
					n.X = a.Y * b.Z - a.Z * b.Y;
					n.Y = a.Z * b.X - a.X * b.Z;
					n.Z = a.X * b.Y - a.Y * b.X;
					for (auto& c : corners) c.Normal = n;
				}
				for (auto& c : corners) {
					cur.Indices.push_back((unsigned int)cur.Vertices.size());
					cur.Vertices.push_back(c);
					LoadedIndices.push_back((unsigned int)LoadedVertices.size());
					LoadedVertices.push_back(c);
				}
			}
		}
		if (!cur.Vertices.empty()) LoadedMeshes.push_back(cur);
		return !(LoadedMeshes.empty() && LoadedVertices.empty());
	}

private:
	template <typename V>
	static V pick(const std::vector<V>& arr, int i) {  // 1-based, negative = relative to the end
		if (i < 0) i = (int)arr.size() + i;
		else i = i - 1;
		if (i < 0 || (size_t)i >= arr.size()) return V();
		return arr[(size_t)i];
	}
	static Vector3 sub(const Vector3& a, const Vector3& b) {
		Vector3 r;
		r.X = a.X - b.X;
		r.Y = a.Y - b.Y;
		r.Z = a.Z - b.Z;
		return r;
	}
};

}  // namespace objl
