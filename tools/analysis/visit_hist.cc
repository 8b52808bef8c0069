// Analysis tool (not product, not oracle): per-depth histogram of BVH node visits
// for the primary and AO rays of a frame, using the product's host code for the
// scene and a plain skip-list walk (AO rays stop at the first accepted triangle,
// like the GPU kernel).  Used to size the LDS-resident top of the tree.
//   g++ -O2 -fopenmp -I opencl_raytracer_amd/csrc tools/analysis/visit_hist.cc \
//       opencl_raytracer_amd/csrc/{mesh,bvh,scene_pack,walk_tree,ray_tracer}.cc -o /tmp/visit_hist
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bvh.h"
#include "mesh.h"
#include "scene_pack.h"
using namespace ocrt;

struct R { float o[3], d[3], inv[3]; };
static bool slab(const NodeRec &n, const R &r, float maxd) {
	float tn = 1e-45f, tf = maxd;
	for (int k = 0; k < 3; ++k) {
		float a = (n.lo[k] - r.o[k]) * r.inv[k], b = (n.hi[k] - r.o[k]) * r.inv[k];
		tn = fmaxf(tn, fminf(a, b));
		tf = fminf(tf, fmaxf(a, b));
	}
	return tn <= tf && tn < maxd;
}
static bool tri(const TriRec &t, const R &r, float *dist, float *s_, float *t_, float p[3]) {
	float w0[3] = { r.o[0] - t.ta[0], r.o[1] - t.ta[1], r.o[2] - t.ta[2] };
	float a = -((t.n[0] * w0[0] + t.n[1] * w0[1]) + t.n[2] * w0[2]);
	float b = (t.n[0] * r.d[0] + t.n[1] * r.d[1]) + t.n[2] * r.d[2];
	if (fabsf(b) < 1e-6f) return false;
	float rr = a / b;
	if (rr < 0) return false;
	float ip[3] = { r.o[0] + rr * r.d[0], r.o[1] + rr * r.d[1], r.o[2] + rr * r.d[2] };
	float w[3] = { ip[0] - t.ta[0], ip[1] - t.ta[1], ip[2] - t.ta[2] };
	float wu = (t.u[0] * w[0] + t.u[1] * w[1]) + t.u[2] * w[2];
	float wv = (w[0] * t.v[0] + w[1] * t.v[1]) + w[2] * t.v[2];
	float s = (t.uv * wv - t.vv * wu) / t.D;
	if (s < -1e-5f || (double) s > 1.00001) return false;
	float tt = (t.uv * wu - t.uu * wv) / t.D;
	if (tt < -1e-5f || (double) (s + tt) > 1.00001) return false;
	float e[3] = { ip[0] - r.o[0], ip[1] - r.o[1], ip[2] - r.o[2] };
	*dist = sqrtf((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
	*s_ = s; *t_ = tt; memcpy(p, ip, sizeof ip);
	return true;
}
int main(int argc, char **argv) {
	Mesh m; load_off_mesh(argv[1], &m); compute_vertex_normals(&m);
	BVH bvh(argc > 2 && !strcmp(argv[2], "sah") ? BVH::Method::SURFACE_AREA_HEURISTIC : BVH::Method::CUT_LONGEST_AXIS);
	bvh.buildBVH(m);
	auto sf = sort_faces_by_leaf_order(m, bvh);
	PackedScene P = pack_scene(sf, bvh.nodes, bvh.aabbs, m.vertices, m.vnormals);
	const size_t N = P.nodes.size();
	std::vector<int> depth(N, 0);
	{ std::vector<std::pair<size_t,int>> st; for (size_t i = 0; i < N; ++i) { while (!st.empty() && st.back().first <= i) st.pop_back(); depth[i] = (int) st.size(); if (P.nodes[i].skip > 1) st.push_back({ i + P.nodes[i].skip, 0 }); } }
	const int W = 1920, H = 1080;
	auto table = uniform_ao_table(3, 4, 90);
	const int ND = (int) table.size() / 4;
	std::vector<unsigned long long> hist_p(64, 0), hist_a(64, 0);
	unsigned long long rays_a = 0, rays_p = 0, leaf_p = 0, leaf_a = 0;
	const float a = 1.0f * 1920;
#pragma omp parallel
	{
		std::vector<unsigned long long> hp(64, 0), ha(64, 0);
		unsigned long long ra = 0, rp = 0, lp = 0, la = 0;
#pragma omp for schedule(dynamic, 4)
		for (int y = 0; y < H; y += 2) for (int x = 0; x < W; x += 2) {  // quarter sampling
			R r; r.o[0] = 0; r.o[1] = 0; r.o[2] = 2;
			float d[3] = { (x + 0.5f) / a - W / (2.0f * a), -((y + 0.5f) / a - H / (2.0f * a)), -1.0f };
			float l = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
			for (int k = 0; k < 3; ++k) { r.d[k] = d[k] / l; r.inv[k] = 1.0f / r.d[k]; }
			float best = INFINITY, bs = 0, bt = 0, bp[3] = { 0, 0, 0 }; unsigned bl = 0; bool hit = false;
			++rp;
			for (size_t i = 0; i < N;) {
				hp[depth[i]]++;
				if (slab(P.nodes[i], r, 100000.0f)) {
					if (P.nodes[i].skip == 1) { ++lp; float dd, s, t, p[3]; if (tri(P.tris[P.nodes[i].leaf], r, &dd, &s, &t, p)) { hit = true; if (best > dd) { best = dd; bs = s; bt = t; memcpy(bp, p, sizeof p); bl = P.nodes[i].leaf; } } }
					++i;
				} else i += P.nodes[i].skip;
			}
			if (!hit) continue;
			const ShadeRec &sh = P.shade[bl];
			float b0 = 1.0f - bs - bt, n[3];
			for (int k = 0; k < 3; ++k) n[k] = (sh.n0[k] * b0 + sh.n1[k] * bs) + sh.n2[k] * bt;
			float nl = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
			for (int k = 0; k < 3; ++k) n[k] /= nl;
			float h[3] = { n[0], n[1], n[2] };
			float ax = fabsf(n[0]), ay = fabsf(n[1]), az = fabsf(n[2]);
			if (ax <= ay && ax <= az) h[0] = 1; else if (ay <= ax && ay <= az) h[1] = 1; else h[2] = 1;
			float bx[3] = { h[1] * n[2] - h[2] * n[1], h[2] * n[0] - h[0] * n[2], h[0] * n[1] - h[1] * n[0] };
			float bl2 = sqrtf((bx[0] * bx[0] + bx[1] * bx[1]) + bx[2] * bx[2]); for (int k = 0; k < 3; ++k) bx[k] /= bl2;
			float bz[3] = { bx[1] * n[2] - bx[2] * n[1], bx[2] * n[0] - bx[0] * n[2], bx[0] * n[1] - bx[1] * n[0] };
			float bl3 = sqrtf((bz[0] * bz[0] + bz[1] * bz[1]) + bz[2] * bz[2]); for (int k = 0; k < 3; ++k) bz[k] /= bl3;
			for (int q = 0; q < ND; ++q) {
				R ar; for (int k = 0; k < 3; ++k) { ar.o[k] = bp[k] + n[k] * 1e-5f; ar.d[k] = (bx[k] * table[4 * q] + n[k] * table[4 * q + 1]) + bz[k] * table[4 * q + 2]; ar.inv[k] = 1.0f / ar.d[k]; }
				++ra;
				for (size_t i = 0; i < N;) {
					ha[depth[i]]++;
					if (slab(P.nodes[i], ar, 0.2f)) {
						if (P.nodes[i].skip == 1) { ++la; float dd, s, t, p[3]; if (tri(P.tris[P.nodes[i].leaf], ar, &dd, &s, &t, p)) break; }
						++i;
					} else i += P.nodes[i].skip;
				}
			}
		}
#pragma omp critical
		{ for (int k = 0; k < 64; ++k) { hist_p[k] += hp[k]; hist_a[k] += ha[k]; } rays_a += ra; rays_p += rp; leaf_p += lp; leaf_a += la; }
	}
	unsigned long long tp = 0, ta = 0; for (int k = 0; k < 64; ++k) { tp += hist_p[k]; ta += hist_a[k]; }
	printf("primary rays %llu visits/ray %.2f leaf tests/ray %.2f | AO rays %llu visits/ray %.2f leaf tests/ray %.2f\n", rays_p, (double) tp / rays_p, (double) leaf_p / rays_p, rays_a, (double) ta / rays_a, (double) leaf_a / rays_a);
	std::vector<size_t> per_depth(64, 0); for (size_t i = 0; i < N; ++i) per_depth[depth[i]]++;
	unsigned long long cp = 0, ca = 0; size_t cn = 0;
	printf("depth nodes cum_nodes | primary: visits/ray cum%% | AO: visits/ray cum%%\n");
	for (int k = 0; k < 40 && (hist_p[k] || hist_a[k]); ++k) {
		cp += hist_p[k]; ca += hist_a[k]; cn += per_depth[k];
		printf("%2d %7zu %7zu | %6.2f %5.1f | %6.2f %5.1f\n", k, per_depth[k], cn, (double) hist_p[k] / rays_p, 100.0 * cp / tp, (double) hist_a[k] / rays_a, 100.0 * ca / ta);
	}
}
