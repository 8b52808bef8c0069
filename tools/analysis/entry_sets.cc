// Analysis tool (not product, not oracle): how many node tests would the shared ambient-occlusion walk of a tile save
// if it did not start at the root but at a short per-tile list of subtrees -- those that overlap the box of the tile's
// hit points grown by AO_MAX_DISTANCE (every leaf a ray of the tile can reach overlaps it)?
//   g++ -O2 -fopenmp -I opencl_raytracer_amd/csrc tools/analysis/entry_sets.cc \
//       opencl_raytracer_amd/csrc/{mesh,bvh,scene_pack,walk_tree,ray_tracer}.cc -o /tmp/entry_sets
//   /tmp/entry_sets meshes/bunny.off 1920 1080 3 0.2
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

struct Range { size_t at, end; };
static bool overlaps(const NodeRec &n, const float qlo[3], const float qhi[3]) {
	for (int k = 0; k < 3; ++k) if (n.lo[k] > qhi[k] || n.hi[k] < qlo[k]) return false;
	return true;
}
// the frontier: split the largest entry into its overlapping children until `budget` entries exist
static std::vector<Range> entries(const std::vector<NodeRec> &nodes, const float qlo[3], const float qhi[3], size_t budget, int *tests) {
	std::vector<Range> list;
	*tests = 1;
	if (!overlaps(nodes[0], qlo, qhi)) return list;
	list.push_back({ 0, nodes.size() });
	for (;;) {
		// the largest splittable entry
		size_t pick = list.size(), size = 1;
		for (size_t e = 0; e < list.size(); ++e) if (list[e].end - list[e].at > size) { size = list[e].end - list[e].at; pick = e; }
		if (pick == list.size()) break;
		// its children (the walk tree is contracted: a node may have more than two)
		const size_t at = list[pick].at, end = list[pick].end;
		std::vector<Range> kids;
		for (size_t c = at + 1; c < end; c += nodes[c].skip) {
			++*tests;
			if (overlaps(nodes[c], qlo, qhi)) kids.push_back({ c, c + nodes[c].skip });
		}
		if (list.size() - 1 + kids.size() > budget && kids.size() > 1) break;
		list.erase(list.begin() + pick);
		list.insert(list.begin() + pick, kids.begin(), kids.end());
		if (list.empty()) break;
	}
	return list;
}
int main(int argc, char **argv) {
	Mesh m; load_off_mesh(argv[1], &m); compute_vertex_normals(&m);
	BVH bvh(BVH::Method::CUT_LONGEST_AXIS);
	bvh.buildBVH(m);
	auto sf = sort_faces_by_leaf_order(m, bvh);
	PackedScene P = pack_scene(sf, bvh.nodes, bvh.aabbs, m.vertices, m.vnormals);
	const size_t N = P.nodes.size();
	const int W = argc > 2 ? atoi(argv[2]) : 1920, H = argc > 3 ? atoi(argv[3]) : 1080;
	const int stride = argc > 4 ? atoi(argv[4]) : 3;
	const float D = argc > 5 ? (float) atof(argv[5]) : 0.2f;
	auto table = uniform_ao_table(3, 4, 90);
	const int ND = (int) table.size() / 4;
	const float a = 1.0f * (W > H ? W : H);
	const size_t budgets[5] = { 1, 2, 4, 8, 16 };
	unsigned long long open_visits = 0, coherent_packets = 0;  // one entry, the walk runs on to the end of the array (no range check in the loop)
	unsigned long long packets = 0, root_visits = 0, entry_visits[5] = { 0 }, entry_tests[5] = { 0 }, entry_count[5] = { 0 }, tiles = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : packets, root_visits, tiles, open_visits)
	for (int ty = 0; ty < H / 8; ty += stride) for (int tx = 0; tx < W / 8; tx += stride) {
		float hp[64][3], hn[64][3]; int nh = 0;
		for (int l = 0; l < 64; ++l) {
			const int x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
			R r; r.o[0] = 0; r.o[1] = 0; r.o[2] = 2; r.live = true;
			float d[3] = { (x + 0.5f) / a - W / (2.0f * a), -((y + 0.5f) / a - H / (2.0f * a)), -1.0f };
			float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
			for (int k = 0; k < 3; ++k) { r.d[k] = d[k] / len; r.inv[k] = 1.0f / r.d[k]; }
			float best = INFINITY, bs = 0, bt = 0, bp[3] = { 0, 0, 0 }; unsigned bl = 0; bool hit = false;
			for (size_t i = 0; i < N;) {
				if (slab(P.nodes[i], r, 100000.0f)) {
					if (P.nodes[i].skip == 1) { float dd, s, t, p[3]; if (tri(P.tris[P.nodes[i].leaf], r, &dd, &s, &t, p)) { hit = true; if (best > dd) { best = dd; bs = s; bt = t; memcpy(bp, p, sizeof p); bl = P.nodes[i].leaf; } } }
					++i;
				} else i += P.nodes[i].skip;
			}
			if (!hit) continue;
			const ShadeRec &sh = P.shade[bl];
			float b0 = 1.0f - bs - bt, n[3];
			for (int k = 0; k < 3; ++k) n[k] = (sh.n0[k] * b0 + sh.n1[k] * bs) + sh.n2[k] * bt;
			float nl = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
			for (int k = 0; k < 3; ++k) { hn[nh][k] = n[k] / nl; hp[nh][k] = bp[k]; }
			++nh;
		}
		if (!nh) continue;
		++tiles;
		float qlo[3] = { INFINITY, INFINITY, INFINITY }, qhi[3] = { -INFINITY, -INFINITY, -INFINITY };
		for (int l = 0; l < nh; ++l) for (int k = 0; k < 3; ++k) { qlo[k] = fminf(qlo[k], hp[l][k] - D * 1.002f - 1e-4f); qhi[k] = fmaxf(qhi[k], hp[l][k] + D * 1.002f + 1e-4f); }
		std::vector<Range> lists[5]; int tests[5];
		if (getenv("DEBUG_TILE") && tiles < 3) printf("tile %d,%d nh %d Q [%g %g %g] [%g %g %g] root [%g %g %g] [%g %g %g]\n", tx, ty, nh, qlo[0], qlo[1], qlo[2], qhi[0], qhi[1], qhi[2], P.nodes[0].lo[0], P.nodes[0].lo[1], P.nodes[0].lo[2], P.nodes[0].hi[0], P.nodes[0].hi[1], P.nodes[0].hi[2]);
		for (int b = 0; b < 5; ++b) lists[b] = entries(P.nodes, qlo, qhi, budgets[b], &tests[b]);
		unsigned long long ev[5] = { 0 }, rv = 0;
		for (int q = 0; q < ND; ++q) {
			R ar[64];
			for (int l = 0; l < nh; ++l) {
				const float *n = hn[l];
				float h[3] = { n[0], n[1], n[2] };
				float ax = fabsf(n[0]), ay = fabsf(n[1]), az = fabsf(n[2]);
				if (ax <= ay && ax <= az) h[0] = 1; else if (ay <= ax && ay <= az) h[1] = 1; else h[2] = 1;
				float bx[3] = { h[1] * n[2] - h[2] * n[1], h[2] * n[0] - h[0] * n[2], h[0] * n[1] - h[1] * n[0] };
				float l2 = sqrtf((bx[0] * bx[0] + bx[1] * bx[1]) + bx[2] * bx[2]); for (int k = 0; k < 3; ++k) bx[k] /= l2;
				float bz[3] = { bx[1] * n[2] - bx[2] * n[1], bx[2] * n[0] - bx[0] * n[2], bx[0] * n[1] - bx[1] * n[0] };
				float l3 = sqrtf((bz[0] * bz[0] + bz[1] * bz[1]) + bz[2] * bz[2]); for (int k = 0; k < 3; ++k) bz[k] /= l3;
				for (int k = 0; k < 3; ++k) { ar[l].o[k] = hp[l][k] + n[k] * 1e-5f; ar[l].d[k] = (bx[k] * table[4 * q] + n[k] * table[4 * q + 1]) + bz[k] * table[4 * q + 2]; ar[l].inv[k] = 1.0f / ar[l].d[k]; }
			}
			{  // sign coherence of the packet (kernels.hip, walk_variant): every ray agrees on every axis?
				bool coherent = true;
				for (int k = 0; k < 3; ++k) { int pos = 0; for (int l = 0; l < nh; ++l) pos += ar[l].inv[k] >= 0; coherent = coherent && (pos == 0 || pos == nh); }
#pragma omp atomic
				coherent_packets += coherent ? 1 : 0;
			}
			auto walk = [&](size_t from, size_t to, R *rays, int *live) {
				unsigned long long v = 0;
				for (size_t i = from; i < to && *live;) {
					++v;
					int hits = 0;
					for (int l = 0; l < nh; ++l) if (rays[l].live && slab(P.nodes[i], rays[l], D)) {
						++hits;
						if (P.nodes[i].skip == 1) { float dd, s, t, p[3]; if (tri(P.tris[P.nodes[i].leaf], rays[l], &dd, &s, &t, p)) { rays[l].live = false; --*live; } }
					}
					if (hits) ++i; else i += P.nodes[i].skip;
				}
				return v;
			};
			{ R c[64]; memcpy(c, ar, sizeof c); for (int l = 0; l < nh; ++l) c[l].live = true; int live = nh; rv += walk(0, N, c, &live); }
			if (!lists[0].empty()) { R c[64]; memcpy(c, ar, sizeof c); for (int l = 0; l < nh; ++l) c[l].live = true; int live = nh; open_visits += walk(lists[0][0].at, N, c, &live); }
			for (int b = 0; b < 5; ++b) {
				R c[64]; memcpy(c, ar, sizeof c); for (int l = 0; l < nh; ++l) c[l].live = true; int live = nh;
				for (const Range &e : lists[b]) ev[b] += walk(e.at, e.end, c, &live);
			}
		}
		packets += ND; root_visits += rv;
		if (getenv("DUMP_TILES")) {
			// cost class as primary_kernel computes it: leaves the tile's shared primary walk stopped at
			R pr[64];
			for (int l = 0; l < 64; ++l) {
				const int x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
				R &r = pr[l]; r.o[0] = 0; r.o[1] = 0; r.o[2] = 2; r.live = true;
				float d[3] = { (x + 0.5f) / a - W / (2.0f * a), -((y + 0.5f) / a - H / (2.0f * a)), -1.0f };
				float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
				for (int k = 0; k < 3; ++k) { r.d[k] = d[k] / len; r.inv[k] = 1.0f / r.d[k]; }
			}
			int leaf_stops = 0, node_tests = 0, leaf_pairs = 0;
			for (size_t i = 0; i < N;) {
				int hits = 0;
				++node_tests;
				for (int l = 0; l < 64; ++l) hits += slab(P.nodes[i], pr[l], 100000.0f);
				if (hits) { if (P.nodes[i].skip == 1) { ++leaf_stops; leaf_pairs += hits; } ++i; } else i += P.nodes[i].skip;
			}
#pragma omp critical
			printf("TILE %d %d hits %d class %d ao_node_tests %llu prim_nodes %d prim_pairs %d\n", tx, ty, nh, leaf_stops, rv, node_tests, leaf_pairs);
		}
#pragma omp critical
		for (int b = 0; b < 5; ++b) { entry_visits[b] += ev[b]; entry_tests[b] += tests[b]; entry_count[b] += lists[b].size(); }
	}
	printf("%llu tiles with hits, %llu AO packets: %.1f node tests per packet from the root\n", tiles, packets, (double) root_visits / packets);
	printf("  sign-coherent packets: %.1f %%\n", 100.0 * coherent_packets / packets);
	printf("  one entry, walking on to the end of the array: %.1f node tests per packet (%.1f%%)\n", (double) open_visits / packets, 100.0 * open_visits / root_visits);
	for (int b = 0; b < 5; ++b)
		printf("  up to %2zu entries per tile (%.1f on average, found with %.1f box tests per tile): %.1f node tests per packet (%.1f%%)\n", budgets[b],
		       (double) entry_count[b] / tiles, (double) entry_tests[b] / tiles, (double) entry_visits[b] / packets, 100.0 * entry_visits[b] / root_visits);
}
