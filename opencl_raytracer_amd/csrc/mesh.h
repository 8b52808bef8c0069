// mesh.h -- triangle mesh + OFF reader (host side of the scene build).
//
// Mirrors the interface of reference include/mesh.h:14-23: a Mesh is three flat
// arrays, faces hold 3 vertex ids per triangle, and the two free functions keep
// the reference's names, argument meaning and error behaviour.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "vec3.h"

struct Mesh {
	std::vector<Vec3f> vertices;
	std::vector<uint32_t> faces;
	std::vector<Vec3f> vnormals;
};

// Reads an OFF file ("OFF", "V F E", V x "x y z", F x "3 a b c").
// Throws std::invalid_argument("No filename given") / std::runtime_error(
// "Cannot read file" | "File not recognized as OFF model" |
// "Invalid face with != 3 vertices"); a face with an out-of-range vertex id is
// skipped with a warning on stdout (reference src/mesh.cc:7-67).
void load_off_mesh(const std::string &filename, Mesh *mesh);

// Area-weighted vertex normals: un-normalised face normals accumulated in file
// order, then normalised; zero-length normals stay zero (reference src/mesh.cc:95-139).
void compute_vertex_normals(Mesh *mesh);
