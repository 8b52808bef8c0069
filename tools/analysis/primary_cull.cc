#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "bvh.h"
#include "mesh.h"
#include "scene_pack.h"
using namespace ocrt;

struct R { float o[3], d[3], inv[3]; bool live; };
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

// Analysis tool (not product, not oracle): node tests of the shared PRIMARY walk of a tile as it is (every ray keeps
// max_distance = 100000 to the end, like the reference) and if a ray that has found a triangle only entered boxes
// nearer than its best hit so far (immediately / only after every 64 collected pairs, like the batches).
//   g++ -O2 -fopenmp -I opencl_raytracer_amd/csrc tools/analysis/primary_cull.cc \
//       opencl_raytracer_amd/csrc/{mesh,bvh,scene_pack,walk_tree,ray_tracer}.cc -o /tmp/primary_cull
int main(int argc, char **argv) {
	Mesh m; load_off_mesh(argv[1], &m); compute_vertex_normals(&m);
	BVH bvh(BVH::Method::CUT_LONGEST_AXIS);
	bvh.buildBVH(m);
	auto sf = sort_faces_by_leaf_order(m, bvh);
	PackedScene P = pack_scene(sf, bvh.nodes, bvh.aabbs, m.vertices, m.vnormals);
	const size_t N = P.nodes.size();
	const int W = argc > 2 ? atoi(argv[2]) : 1920, H = argc > 3 ? atoi(argv[3]) : 1080;
	const int stride = argc > 4 ? atoi(argv[4]) : 3;
	const float a = 1.0f * (W > H ? W : H);
	unsigned long long packets = 0, plain = 0, culled = 0, batched = 0, plain_pairs = 0, culled_pairs = 0, batched_pairs = 0, interval_visits = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : interval_visits, packets, plain, culled, batched, plain_pairs, culled_pairs, batched_pairs)
	for (int ty = 0; ty < H / 8; ty += stride) for (int tx = 0; tx < W / 8; tx += stride) {
		R rays[64];
		for (int l = 0; l < 64; ++l) {
			const int x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
			R &r = rays[l]; r.o[0] = 0; r.o[1] = 0; r.o[2] = 2; r.live = true;
			float d[3] = { (x + 0.5f) / a - W / (2.0f * a), -((y + 0.5f) / a - H / (2.0f * a)), -1.0f };
			float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
			for (int k = 0; k < 3; ++k) { r.d[k] = d[k] / len; r.inv[k] = 1.0f / r.d[k]; }
		}
		++packets;
		{
			// the tile's interval: segments from where a ray enters the root box to where it leaves it
			float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
			bool any = false;
			for (int l = 0; l < 64; ++l) {
				float tn = 0.0f, tf = 100000.0f;
				for (int k = 0; k < 3; ++k) {
					float a = (P.nodes[0].lo[k] - rays[l].o[k]) * rays[l].inv[k], b = (P.nodes[0].hi[k] - rays[l].o[k]) * rays[l].inv[k];
					tn = fmaxf(tn, fminf(a, b)); tf = fminf(tf, fmaxf(a, b));
				}
				if (!(tn <= tf)) continue;
				any = true;
				for (int k = 0; k < 3; ++k) {
					const float a = rays[l].o[k] + rays[l].d[k] * tn * 0.999f, b = rays[l].o[k] + rays[l].d[k] * tf * 1.001f;
					lo[k] = fminf(lo[k], fminf(a, b) - 1e-4f); hi[k] = fmaxf(hi[k], fmaxf(a, b) + 1e-4f);
				}
			}
			auto meets = [&](size_t c) {
				bool over = true;
				for (int k = 0; k < 3; ++k) over = over && !(P.nodes[c].lo[k] > hi[k] || P.nodes[c].hi[k] < lo[k]);
				return over;
			};
			size_t begin = 0, end = any ? P.nodes[0].skip : 0, n = 0;
			while (any && P.nodes[n].skip > 1) {
				size_t first = 0, index = 0, others = 0;
				for (size_t c = n + 1; c < n + P.nodes[n].skip; c += P.nodes[c].skip) {
					if (first) others += meets(c);
					else { ++index; if (meets(c)) first = c; }
				}
				if (!first) { n += P.nodes[n].skip; break; }
				if (index == 1 && others) break;
				n = first;
			}
			begin = n; n = 0;
			while (any && P.nodes[n].skip > 1) {
				size_t last = 0;
				for (size_t c = n + 1; c < n + P.nodes[n].skip; c += P.nodes[c].skip) if (meets(c)) last = c;
				if (!last) { end = n; break; }
				end = last + P.nodes[last].skip; n = last;
			}
			for (size_t i = begin; i < end;) {
				++interval_visits;
				int hits = 0;
				for (int l = 0; l < 64; ++l) hits += slab(P.nodes[i], rays[l], 100000.0f);
				if (hits) ++i; else i += P.nodes[i].skip;
			}
		}
		for (int mode = 0; mode < 3; ++mode) {
			float best[64], limit[64];
			for (int l = 0; l < 64; ++l) { best[l] = INFINITY; limit[l] = 100000.0f; }
			unsigned long long visits = 0, pairs = 0, waiting = 0;
			for (size_t i = 0; i < N;) {
				++visits;
				int hits = 0;
				for (int l = 0; l < 64; ++l) if (slab(P.nodes[i], rays[l], limit[l])) {
					++hits;
					if (P.nodes[i].skip == 1) {
						++pairs; ++waiting;
						float dd, s, t, p[3];
						if (tri(P.tris[P.nodes[i].leaf], rays[l], &dd, &s, &t, p) && dd < best[l]) best[l] = dd;
						if (mode == 1 && best[l] < INFINITY) limit[l] = best[l] * 1.0001f;
					}
				}
				if (mode == 2 && waiting >= 64) { waiting = 0; for (int l = 0; l < 64; ++l) if (best[l] < INFINITY) limit[l] = best[l] * 1.0001f; }
				if (hits) ++i; else i += P.nodes[i].skip;
			}
			if (mode == 0) { plain += visits; plain_pairs += pairs; } else if (mode == 1) { culled += visits; culled_pairs += pairs; } else { batched += visits; batched_pairs += pairs; }
		}
	}
	printf("with a walk interval per tile (box around the rays' segments inside the root box): %.1f node tests per packet\n", (double) interval_visits / packets);
	printf("%llu primary packets: %.1f node tests and %.1f (lane, leaf) pairs per packet as it is; limit = best hit at once: %.1f and %.1f; after every 64 pairs: %.1f and %.1f\n",
	       packets, (double) plain / packets, (double) plain_pairs / packets, (double) culled / packets, (double) culled_pairs / packets, (double) batched / packets, (double) batched_pairs / packets);
}
