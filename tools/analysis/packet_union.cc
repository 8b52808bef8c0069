// Analysis tool (not product, not oracle): how many nodes would a SHARED skip-list
// walk of a whole 8x8 tile visit (a node is entered when any of the tile's live rays
// hits its box) compared with the rays' individual walks?  Primary packets = the 64
// sub-pixels of a tile; AO packets = one table direction from all hit points of a tile.
//   g++ -O2 -fopenmp -I opencl_raytracer_amd/csrc tools/analysis/packet_union.cc \
//       opencl_raytracer_amd/csrc/{mesh,bvh,scene_pack,walk_tree,ray_tracer}.cc -o /tmp/packet_union
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
struct Stats { unsigned long long nodeath_visits = 0, max_tests_sum = 0, tests_hist[16] = {0}; unsigned long long packets = 0, rays = 0, packet_visits = 0, single_visits = 0, lane_box_hits = 0, lane_live_at_visit = 0, leaf_tests = 0, packet_leaf_visits = 0; };
int main(int argc, char **argv) {
	Mesh m; load_off_mesh(argv[1], &m); compute_vertex_normals(&m);
	BVH bvh(BVH::Method::CUT_LONGEST_AXIS);
	bvh.buildBVH(m);
	auto sf = sort_faces_by_leaf_order(m, bvh);
	PackedScene P = pack_scene(sf, bvh.nodes, bvh.aabbs, m.vertices, m.vnormals);
	const size_t N = P.nodes.size();
	const int W = argc > 2 ? atoi(argv[2]) : 1920, H = argc > 3 ? atoi(argv[3]) : 1080;
	const int stride = argc > 4 ? atoi(argv[4]) : 3;  // every stride-th tile in x and y
	auto table = uniform_ao_table(3, 4, 90);
	const int ND = (int) table.size() / 4;
	const float a = 1.0f * (W > H ? W : H);
	Stats prim, ao;
	std::vector<double> tile_prim, tile_prim_leaves, tile_ao, tile_hits_v;
#pragma omp parallel
	{
		Stats sp, sa;
#pragma omp for schedule(dynamic, 1)
		for (int ty = 0; ty < H / 8; ty += stride) for (int tx = 0; tx < W / 8; tx += stride) {
			R rays[64]; float hp[64][3], hn[64][3]; int nh = 0;
			for (int l = 0; l < 64; ++l) {
				const int x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
				R &r = rays[l]; r.o[0] = 0; r.o[1] = 0; r.o[2] = 2; r.live = true;
				float d[3] = { (x + 0.5f) / a - W / (2.0f * a), -((y + 0.5f) / a - H / (2.0f * a)), -1.0f };
				float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
				for (int k = 0; k < 3; ++k) { r.d[k] = d[k] / len; r.inv[k] = 1.0f / r.d[k]; }
			}
			// individual primary walks (also produce the hit points)
			for (int l = 0; l < 64; ++l) {
				const R &r = rays[l];
				float best = INFINITY, bs = 0, bt = 0, bp[3] = { 0, 0, 0 }; unsigned bl = 0; bool hit = false;
				for (size_t i = 0; i < N;) {
					sp.single_visits++;
					if (slab(P.nodes[i], r, 100000.0f)) {
						if (P.nodes[i].skip == 1) { sp.leaf_tests++; float dd, s, t, p[3]; if (tri(P.tris[P.nodes[i].leaf], r, &dd, &s, &t, p)) { hit = true; if (best > dd) { best = dd; bs = s; bt = t; memcpy(bp, p, sizeof p); bl = P.nodes[i].leaf; } } }
						++i;
					} else i += P.nodes[i].skip;
				}
				sp.rays++;
				if (!hit) continue;
				const ShadeRec &sh = P.shade[bl];
				float b0 = 1.0f - bs - bt, n[3];
				for (int k = 0; k < 3; ++k) n[k] = (sh.n0[k] * b0 + sh.n1[k] * bs) + sh.n2[k] * bt;
				float nl = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
				for (int k = 0; k < 3; ++k) { hn[nh][k] = n[k] / nl; hp[nh][k] = bp[k]; }
				++nh;
			}
			// shared primary walk
			sp.packets++;
			unsigned long long pv0 = sp.packet_visits, pl0 = sp.packet_leaf_visits, av0 = sa.packet_visits;
			for (size_t i = 0; i < N;) {
				sp.packet_visits++;
				int hits = 0;
				for (int l = 0; l < 64; ++l) hits += slab(P.nodes[i], rays[l], 100000.0f);
				sp.lane_box_hits += hits; sp.lane_live_at_visit += 64;
				if (hits) { if (P.nodes[i].skip == 1) sp.packet_leaf_visits++; ++i; } else i += P.nodes[i].skip;
			}
			const double my_prim = (double) (sp.packet_visits - pv0), my_leaves = (double) (sp.packet_leaf_visits - pl0);
			if (!nh) continue;
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
					ar[l].live = true;
				}
				for (int l = 0; l < nh; ++l) {
					sa.rays++;
					for (size_t i = 0; i < N;) {
						sa.single_visits++;
						if (slab(P.nodes[i], ar[l], 0.2f)) {
							if (P.nodes[i].skip == 1) { sa.leaf_tests++; float dd, s, t, p[3]; if (tri(P.tris[P.nodes[i].leaf], ar[l], &dd, &s, &t, p)) break; }
							++i;
						} else i += P.nodes[i].skip;
					}
				}
				// shared AO walk in which nobody leaves early (leaf tests deferred to the end)
				{
					int tests[64] = { 0 };
					for (size_t i = 0; i < N;) {
						sa.nodeath_visits++;
						int hits = 0;
						for (int l = 0; l < nh; ++l) if (slab(P.nodes[i], ar[l], 0.2f)) { ++hits; if (P.nodes[i].skip == 1) tests[l]++; }
						if (hits) ++i; else i += P.nodes[i].skip;
					}
					int mx = 0; for (int l = 0; l < nh; ++l) { mx = tests[l] > mx ? tests[l] : mx; sa.tests_hist[tests[l] > 15 ? 15 : tests[l]]++; }
					sa.max_tests_sum += mx;
				}
				// shared AO walk; a lane leaves the packet at its first accepted triangle
				sa.packets++;
				int live = nh;
				for (size_t i = 0; i < N && live;) {
					sa.packet_visits++;
					int hits = 0;
					sa.lane_live_at_visit += live;
					for (int l = 0; l < nh; ++l) if (ar[l].live && slab(P.nodes[i], ar[l], 0.2f)) {
						++hits;
						if (P.nodes[i].skip == 1) { float dd, s, t, p[3]; if (tri(P.tris[P.nodes[i].leaf], ar[l], &dd, &s, &t, p)) { ar[l].live = false; --live; } }
					}
					sa.lane_box_hits += hits;
					if (hits) { if (P.nodes[i].skip == 1) sa.packet_leaf_visits++; ++i; } else i += P.nodes[i].skip;
				}
			}
#pragma omp critical
			{ tile_prim.push_back(my_prim); tile_prim_leaves.push_back(my_leaves); tile_ao.push_back((double) (sa.packet_visits - av0)); tile_hits_v.push_back(nh); }
		}
#pragma omp critical
		{
			Stats *d[2] = { &prim, &ao }; Stats *s[2] = { &sp, &sa };
			for (int k = 0; k < 2; ++k) { d[k]->packets += s[k]->packets; d[k]->rays += s[k]->rays; d[k]->packet_visits += s[k]->packet_visits; d[k]->single_visits += s[k]->single_visits; d[k]->lane_box_hits += s[k]->lane_box_hits; d[k]->lane_live_at_visit += s[k]->lane_live_at_visit; d[k]->leaf_tests += s[k]->leaf_tests; d[k]->packet_leaf_visits += s[k]->packet_leaf_visits; d[k]->nodeath_visits += s[k]->nodeath_visits; d[k]->max_tests_sum += s[k]->max_tests_sum; for (int q = 0; q < 16; ++q) d[k]->tests_hist[q] += s[k]->tests_hist[q]; }
		}
	}
	auto corr = [](const std::vector<double> &x, const std::vector<double> &y) { double mx = 0, my = 0; size_t n = x.size(); for (size_t i = 0; i < n; ++i) { mx += x[i]; my += y[i]; } mx /= n; my /= n; double sxy = 0, sxx = 0, syy = 0; for (size_t i = 0; i < n; ++i) { sxy += (x[i] - mx) * (y[i] - my); sxx += (x[i] - mx) * (x[i] - mx); syy += (y[i] - my) * (y[i] - my); } return sxy / sqrt(sxx * syy); };
	if (!tile_ao.size()) return 0;
	{ std::vector<double> sorted = tile_ao; std::sort(sorted.begin(), sorted.end()); size_t n = sorted.size(); printf("per hit tile: AO union visits (28 packets) median %.0f, p90 %.0f, p99 %.0f, max %.0f; corr with primary union visits %.2f, with primary leaf visits %.2f, with hit count %.2f\n", sorted[n / 2], sorted[n * 9 / 10], sorted[n * 99 / 100], sorted[n - 1], corr(tile_prim, tile_ao), corr(tile_prim_leaves, tile_ao), corr(tile_hits_v, tile_ao)); }
	const char *name[2] = { "primary", "AO" }; Stats *st[2] = { &prim, &ao };
	for (int k = 0; k < 2; ++k) {
		Stats &s = *st[k];
		printf("%-8s packets %llu rays/packet %.1f | single walk: %.1f visits/ray, %.2f leaf tests/ray | shared walk: %.1f visits/packet (%.1f leaves), %.1f%% of live lanes hit the box\n",
		       name[k], s.packets, (double) s.rays / s.packets, (double) s.single_visits / s.rays, (double) s.leaf_tests / s.rays,
		       (double) s.packet_visits / s.packets, (double) s.packet_leaf_visits / s.packets, 100.0 * s.lane_box_hits / s.lane_live_at_visit);
		if (s.nodeath_visits) {
			printf("         nobody leaves early: %.1f visits/packet, max leaf tests of a lane %.2f on average; lanes by #tests:", (double) s.nodeath_visits / s.packets, (double) s.max_tests_sum / s.packets);
			for (int q = 0; q < 16; ++q) printf(" %.1f%%", 100.0 * s.tests_hist[q] / s.rays);
			printf("\n");
		}
	}
}
