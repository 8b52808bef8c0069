#include "mesh.h"

#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <stdexcept>

namespace {

// Whitespace tokenizer over the whole file.  Numbers go through strtof /
// strtoul, which round exactly like the iostream extraction the reference
// uses, so every float has the same bits.
class Tokens {
	public:
		explicit Tokens(std::string text) : buf(std::move(text)), pos(0) {}
		bool next(const char **begin, size_t *len) {
			while (pos < buf.size() && is_space(buf[pos]))
				++pos;
			if (pos >= buf.size())
				return false;
			const size_t start = pos;
			while (pos < buf.size() && !is_space(buf[pos]))
				++pos;
			*begin = buf.data() + start;
			*len = pos - start;
			// strtof needs a terminator; tokens are separated by at least one
			// whitespace byte (or end of buffer, which std::string terminates).
			if (pos < buf.size())
				buf[pos++] = '\0';
			return true;
		}
		float next_float() {
			const char *p;
			size_t n;
			if (!next(&p, &n))
				throw std::runtime_error("Unexpected end of OFF file");
			return std::strtof(p, nullptr);
		}
		size_t bytes_left() const { return buf.size() - pos; }
		bool next_uint(unsigned long *out) {
			const char *p;
			size_t n;
			if (!next(&p, &n))
				return false;
			char *end = nullptr;
			*out = std::strtoul(p, &end, 10);
			return end != p;
		}
	private:
		static bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }
		std::string buf;
		size_t pos;
};

std::string slurp(const std::string &filename) {
	std::FILE *f = std::fopen(filename.c_str(), "rb");
	if (!f)
		throw std::runtime_error("Cannot read file");
	std::string text;
	char chunk[1 << 16];
	size_t got;
	while ((got = std::fread(chunk, 1, sizeof chunk, f)) > 0)
		text.append(chunk, got);
	std::fclose(f);
	return text;
}

}  // namespace

void load_off_mesh(const std::string &filename, Mesh *mesh) {
	if (filename.empty())
		throw std::invalid_argument("No filename given");
	Tokens in(slurp(filename));
	const char *tok;
	size_t len;
	if (!in.next(&tok, &len) || std::string(tok, len) != "OFF")
		throw std::runtime_error("File not recognized as OFF model");
	unsigned long num_vertices = 0, num_faces = 0, num_edges = 0;
	in.next_uint(&num_vertices);
	in.next_uint(&num_faces);
	in.next_uint(&num_edges);
	// A vertex takes at least 6 bytes of text ("0 0 0\n"), a face at least 8 ("3 0 0 0\n"): a header that
	// promises more than the file can hold is a truncated or hostile file, not a reason to reserve memory.
	if (num_vertices > (in.bytes_left() + 1) / 6 || num_faces > (in.bytes_left() + 1) / 8)
		throw std::runtime_error("OFF header counts exceed the file size");
	mesh->vertices.reserve(num_vertices);
	mesh->faces.reserve(num_faces * 3 + 3);
	for (unsigned long i = 0; i < num_vertices; ++i) {
		const float x = in.next_float();
		const float y = in.next_float();
		const float z = in.next_float();
		mesh->vertices.push_back(Vec3f(x, y, z));
	}
	for (unsigned long i = 0; i < num_faces; ++i) {
		unsigned long corners = 0;
		if (!in.next_uint(&corners))
			throw std::runtime_error("Unexpected end of OFF file");
		if (corners != 3)
			throw std::runtime_error("Invalid face with != 3 vertices");
		unsigned long vidx[3] = { 0, 0, 0 };
		bool indices_good = true;
		for (int j = 0; j < 3; ++j) {
			if (!in.next_uint(&vidx[j]))
				throw std::runtime_error("Unexpected end of OFF file");
			if (vidx[j] >= num_vertices) {
				std::cout << "OFF Loader: Warning: Face " << i << " has invalid vertex " << vidx[j]
				          << ", skipping face." << std::endl;
				indices_good = false;
			}
		}
		if (indices_good)
			for (int j = 0; j < 3; ++j)
				mesh->faces.push_back((uint32_t) vidx[j]);
	}
}

void compute_vertex_normals(Mesh *mesh) {
	mesh->vnormals.assign(mesh->vertices.size(), Vec3f(0, 0, 0));
	unsigned zero_face_normals = 0;
	for (size_t f = 0; f + 2 < mesh->faces.size(); f += 3) {
		const uint32_t ia = mesh->faces[f], ib = mesh->faces[f + 1], ic = mesh->faces[f + 2];
		const Vec3f &a = mesh->vertices[ia];
		const Vec3f face_normal = (mesh->vertices[ib] - a).cross(mesh->vertices[ic] - a);
		if (face_normal.length() == 0) {
			++zero_face_normals;
			continue;
		}
		mesh->vnormals[ia] += face_normal;
		mesh->vnormals[ib] += face_normal;
		mesh->vnormals[ic] += face_normal;
	}
	unsigned zero_vertex_normals = 0;
	for (Vec3f &n : mesh->vnormals) {
		const float l = n.length();
		if (l > 0)
			n /= l;
		else
			++zero_vertex_normals;
	}
	if (zero_face_normals > 0 || zero_vertex_normals > 0)
		std::cout << "Warning: Zero-length normals: " << zero_face_normals << " face normals, "
		          << zero_vertex_normals << " vertex normals" << std::endl;
}
