// pt_scene_io.hpp -- scene files for the plugin.  The reference has no loader: its scene
// is a literal in the constructor (PathTracer.cu:48-69) and the triangle count is hard-coded at the
// launch (:133).  This is the on-disk format next to the path: a line-oriented text file that is a
// superset of the Wavefront OBJ subset {v, f, usemtl} plus two records for what OBJ cannot say:
//
//   # comment
//   v  x y z [w]                      vertex (w defaults to 1, like the reference's vertices)
//   m  name type r g b a [refl]       material; type = diffuse | reflective | emissive (Structures.cuh:57-59)
//   usemtl name                       material for the faces that follow (default: material 0)
//   f  i j k [l ...]                  OBJ face, 1-based (negative = relative), '/'-suffixes ignored, fan-triangulated
//   t  i j k mat                      raw reference Triangle record, 0-based indices (Structures.cuh:48-54)
//
// Host-only, no device work.  Validation (index ranges) is done again by pt_hip_scene_pack.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "pt_hip.h"

namespace ptio {

struct Scene {
	std::vector<pt_vertex> vertices;
	std::vector<pt_triangle> triangles;
	std::vector<pt_material> materials;
};

inline bool parse_material_type(const std::string& s, uint32_t& type) {
	if (s == "diffuse" || s == "0") { type = PT_MAT_DIFFUSE; return true; }
	if (s == "reflective" || s == "1") { type = PT_MAT_REFLECTIVE; return true; }
	if (s == "emissive" || s == "2") { type = PT_MAT_EMISSIVE; return true; }
	return false;
}

// returns "" on success, else a message naming the line
inline std::string read(const char* path, Scene& out) {
	std::ifstream in(path);
	if (!in) return std::string("cannot open '") + path + "'";
	std::map<std::string, uint32_t> names;
	uint32_t current = 0;
	std::string line;
	for (int lineno = 1; std::getline(in, line); lineno++) {
		std::istringstream ss(line);
		std::string tag;
		if (!(ss >> tag) || tag[0] == '#') continue;
		auto bad = [&](const char* what) { return std::string(path) + ":" + std::to_string(lineno) + ": " + what; };
		if (tag == "v") {
			pt_vertex v{};
			v.position.w = 1.0f;
			if (!(ss >> v.position.x >> v.position.y >> v.position.z)) return bad("vertex needs x y z");
			float w;
			if (ss >> w) v.position.w = w;
			out.vertices.push_back(v);
		} else if (tag == "m") {
			std::string name, type;
			pt_material m{};
			if (!(ss >> name >> type >> m.color.x >> m.color.y >> m.color.z >> m.color.w)) return bad("material needs name type r g b a");
			if (!parse_material_type(type, m.type)) return bad("material type must be diffuse, reflective or emissive");
			float refl;
			if (ss >> refl) m.reflectivity = refl;
			names[name] = (uint32_t)out.materials.size();
			out.materials.push_back(m);
		} else if (tag == "usemtl") {
			std::string name;
			ss >> name;
			auto it = names.find(name);
			if (it == names.end()) return bad("usemtl names an undeclared material");
			current = it->second;
		} else if (tag == "f") {
			std::vector<uint32_t> idx;
			std::string tok;
			while (ss >> tok) {
				long v = std::strtol(tok.c_str(), nullptr, 10);          // stops at '/'
				if (v == 0) return bad("face index 0 (OBJ indices are 1-based)");
				long resolved = v > 0 ? v - 1 : (long)out.vertices.size() + v;
				if (resolved < 0 || resolved >= (long)out.vertices.size()) return bad("face index out of range");
				idx.push_back((uint32_t)resolved);
			}
			if (idx.size() < 3) return bad("face needs at least 3 vertices");
			for (size_t k = 1; k + 1 < idx.size(); k++) out.triangles.push_back(pt_triangle{ idx[0], idx[k], idx[k + 1], current });
		} else if (tag == "t") {
			pt_triangle t{};
			if (!(ss >> t.vertexIdx0 >> t.vertexIdx1 >> t.vertexIdx2 >> t.materialIdx)) return bad("triangle needs i j k material");
			out.triangles.push_back(t);
		} else if (tag == "vn" || tag == "vt" || tag == "o" || tag == "g" || tag == "s" || tag == "mtllib") {
			continue;                                                    // OBJ records that carry nothing the tracer uses
		} else {
			return bad("unknown record");
		}
	}
	if (out.materials.empty() && !out.triangles.empty()) return std::string(path) + ": triangles but no material ('m' record)";
	return "";
}

inline std::string write(const char* path, const pt_vertex* v, uint32_t nv, const pt_triangle* t, uint32_t nt,
                         const pt_material* m, uint32_t nm) {
	FILE* f = std::fopen(path, "w");
	if (!f) return std::string("cannot write '") + path + "'";
	std::fprintf(f, "# pathtracer_amd scene: %u vertices, %u triangles, %u materials\n", nv, nt, nm);
	static const char* types[3] = { "diffuse", "reflective", "emissive" };
	for (uint32_t k = 0; k < nm; k++)
		std::fprintf(f, "m m%u %s %.9g %.9g %.9g %.9g %.9g\n", k, m[k].type < 3 ? types[m[k].type] : "reflective",
		             m[k].color.x, m[k].color.y, m[k].color.z, m[k].color.w, m[k].reflectivity);
	for (uint32_t k = 0; k < nv; k++)
		std::fprintf(f, "v %.9g %.9g %.9g %.9g\n", v[k].position.x, v[k].position.y, v[k].position.z, v[k].position.w);
	for (uint32_t k = 0; k < nt; k++)
		std::fprintf(f, "t %u %u %u %u\n", t[k].vertexIdx0, t[k].vertexIdx1, t[k].vertexIdx2, t[k].materialIdx);
	std::fclose(f);
	return "";
}

} // namespace ptio
