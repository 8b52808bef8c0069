// Analysis tool (not product, not oracle): what the ambient-occlusion pass's shared packet walk does per packet, by
// tile class, and what alternatives would do -- counted on the CPU on the very tree the kernels walk (pack_scene's
// rebuilt, contracted walk tree).  Events are those of kernels.hip (walk_collect / shared_walk_any_hit): node tests of
// coherent and mixed packets, leaves tested on the spot (hit by >= batch_below lanes), (lane, leaf) pairs appended
// and the batches they are tested in; a VALU model weighs them (constants read off the device assembly).
//
// Variants:
//   packets   current  : the kernel's pieces (64 >> floor(log2(hits)) directions per piece, 64 rays per packet)
//             dense    : the tile's rays direction-major, 64 at a time across direction boundaries
//             octant   : the tile's rays sorted by the sign octant of their direction, then direction-major
//   PCULL     a LOWER BOUND for any table made per packet at upload: nodes clear of the box around the packet's segments are
//             skipped, inner nodes that hold that box whole are entered, both at no cost
//   PCUT K    at most K intervals per packet: the largest subtree of the list is replaced by its children that meet the box
//             around the packet's segments, neighbours merging, while the list stays within K (what a K-entry table could hold)
//   fat K     subtrees of at most K leaves are "fat leaves": a packet that reaches one with fewer than `batch_below`
//             lanes appends (lane, fat leaf) pairs and skips it; the pairs are expanded 64 at a time, every lane
//             testing the K leaf boxes of ITS pair (dense work instead of wave-wide tests for a few lanes)
//
//   g++ -O2 -fopenmp -I opencl_raytracer_amd/csrc tools/analysis/ao_packets.cc \
//       opencl_raytracer_amd/csrc/{mesh,bvh,scene_pack,walk_tree,ray_tracer}.cc -o /tmp/ao_packets
//   /tmp/ao_packets meshes/bunny.off 1920 1080 1 0.2
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bvh.h"
#include "mesh.h"
#include "scene_pack.h"
using namespace ocrt;

struct R {
	float o[3], d[3], inv[3];
	int h, k;
};
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

struct Events {
	unsigned long long packets = 0, rays = 0, coherent_packets = 0;
	unsigned long long nodes_coherent = 0, nodes_mixed = 0;   // wave-wide node tests
	unsigned long long lane_hits = 0, lane_alive = 0;         // lanes passing / lanes alive at the node tests
	unsigned long long spot_leaves = 0;                      // leaves tested on the spot (wave-wide gate + triangle)
	unsigned long long appends = 0;                          // leaf appends (wave-wide, 5 VALU)
	unsigned long long pairs = 0, batches = 0;               // (lane, leaf) pairs and the batches they ran in
	unsigned long long fat_appends = 0, fat_pairs = 0, fat_batches = 0, fat_box_tests = 0;  // fat-leaf expansion
	unsigned long long occluded = 0;
	unsigned long long own_nodes = 0;                        // node tests of the rays' individual walks (lane work)
	unsigned long long loads[5] = { 0, 0, 0, 0, 0 };        // dependent loads if one load fetched 2 (the kernel), 3, 4, 6, 8 consecutive nodes
	unsigned long long odd_loads = 0;                        // ... of the kernel's, those that begin at an odd node: two 64-byte lines
	unsigned long long line_loads = 0;                       // loads if every load fetched the ALIGNED 64-byte line of the node wanted
	void add(const Events &e) {
		const unsigned long long *s = &e.packets;
		unsigned long long *d = &packets;
		for (size_t i = 0; i < sizeof(Events) / sizeof(unsigned long long); ++i) d[i] += s[i];
	}
};

// VALU per event (wave64 instructions), read off build/kernels.s of round 3's kernels
struct Model {
	// round 3's kernels (the model total then reproduces the measured 620.7 M of the headline pass): set-up 140 incl. ~35
	// lane operations of SGPR spill code, node test 9 / 15, and ~10 lane operations per leaf stop or batch (in `spot`,
	// `batch`).  Round 4 (MODEL=r4): set-up 105, mixed test 12 (centre / half-extent form), spill code gone.
	double setup = 140, coherent = 9, mixed = 15, spot = 95, append = 5, batch = 125, fat_append = 5, fat_batch_fixed = 20, fat_box = 21;
	double tile_setup = 110;  // tangent frames, per job
	Model() {
		const char *m = getenv("MODEL");
		if (m && std::string(m) == "r4") {
			setup = 100; coherent = 12; mixed = 12; spot = 86; batch = 116;  // (one loop for every packet: 12 either way)
		}
	}
	double cost(const Events &e, unsigned long long jobs) const {
		return setup * e.packets + coherent * e.nodes_coherent + mixed * e.nodes_mixed + spot * e.spot_leaves + append * e.appends +
		       batch * e.batches + fat_append * e.fat_appends + fat_batch_fixed * e.fat_batches + fat_box * e.fat_box_tests + tile_setup * jobs;
	}
};

struct Tree {
	const std::vector<NodeRec> *nodes;
	const std::vector<TriRec> *tris;
	std::vector<uint32_t> leaves_under;  // per node
	std::vector<uint32_t> first_child_leaves;  // scratch
};

// One packet through the shared walk.  fat_k = 0: the kernel as it is.
static void walk_packet(const Tree &T, const R *rays, int n, float D, int batch_below, int fat_k, Events &ev, size_t from = 0, size_t to = (size_t) -1,
                        const float *region_lo = nullptr, const float *region_hi = nullptr,
                        const std::vector<std::pair<size_t, size_t>> *ranges = nullptr) {
	const std::vector<NodeRec> &N = *T.nodes;
	bool alive[64];
	int live = n;
	for (int l = 0; l < n; ++l) alive[l] = true;
	bool coherent = true;
	for (int k = 0; k < 3; ++k) {
		int pos = 0;
		for (int l = 0; l < n; ++l) pos += rays[l].inv[k] >= 0;
		coherent = coherent && (pos == 0 || pos == n);
	}
	++ev.packets;
	ev.rays += n;
	ev.coherent_packets += coherent;
	struct Pair { int lane; uint32_t what; };
	std::vector<Pair> waiting, fat_waiting;
	bool found[64] = { false };
	auto run_batch = [&](size_t count) {
		// every pair: gate (the leaf's box was hit, so it passes up to padding) + triangle
		++ev.batches;
		for (size_t p = 0; p < count; ++p) {
			const Pair &pr = waiting[p];
			float dd, s, t, pp[3];
			if (tri((*T.tris)[pr.what], rays[pr.lane], &dd, &s, &t, pp)) found[pr.lane] = true;
		}
		waiting.erase(waiting.begin(), waiting.begin() + count);
		for (int l = 0; l < n; ++l)
			if (alive[l] && found[l]) { alive[l] = false; --live; ++ev.occluded; }
	};
	auto append_leaf_hits = [&](uint32_t leaf, const bool *hit, int hits) {
		++ev.appends;
		for (int l = 0; l < n; ++l) if (hit[l]) waiting.push_back({ l, leaf });
		ev.pairs += hits;
	};
	auto run_fat_batch = [&](size_t count) {
		++ev.fat_batches;
		// every lane tests the leaf boxes of its pair's fat leaf, one child after the other; after each child the
		// pairs found go to the leaf list (which is run whenever it holds 64)
		uint32_t most = 0;
		for (size_t p = 0; p < count; ++p) most = std::max(most, T.leaves_under[fat_waiting[p].what]);
		for (uint32_t j = 0; j < most; ++j) {
			++ev.fat_box_tests;
			for (size_t p = 0; p < count; ++p) {
				const Pair &pr = fat_waiting[p];
				if (j >= T.leaves_under[pr.what]) continue;
				const NodeRec &child = N[pr.what + 1 + j];
				if (alive[pr.lane] && slab(child, rays[pr.lane], D)) { waiting.push_back({ pr.lane, child.leaf }); ++ev.pairs; }
			}
			while (waiting.size() >= 64) run_batch(64);
		}
		fat_waiting.erase(fat_waiting.begin(), fat_waiting.begin() + count);
	};
	if (to > N.size()) to = N.size();
	size_t window[5] = { (size_t) -100, (size_t) -100, (size_t) -100, (size_t) -100, (size_t) -100 }, line = (size_t) -1;
	const size_t n_ranges = ranges ? ranges->size() : 1;
	for (size_t range = 0; range < n_ranges && live; ++range) {
	if (ranges) { from = (*ranges)[range].first; to = (*ranges)[range].second; }
	for (size_t i = from; i < to && live;) {
		if (region_lo) {
			// PCULL, a lower bound for ANY table made per packet at upload: a node clear of the box around the packet's
			// segments is skipped at no cost, an inner node that holds that box whole is entered at no cost
			bool meets = true, holds = true;
			for (int k = 0; k < 3; ++k) {
				meets = meets && !(N[i].lo[k] > region_hi[k] || N[i].hi[k] < region_lo[k]);
				holds = holds && N[i].lo[k] <= region_lo[k] && N[i].hi[k] >= region_hi[k];
			}
			if (!meets) { i += N[i].skip; continue; }
			if (holds && N[i].skip > 1) { ++i; continue; }
		}
		for (int w = 0; w < 5; ++w) {
			static const size_t WIDTH[5] = { 2, 3, 4, 6, 8 };
			if (i < window[w] || i >= window[w] + WIDTH[w]) { ++ev.loads[w]; window[w] = i; if (w == 0) ev.odd_loads += i & 1; }
		}
		if ((i >> 1) != line) { ++ev.line_loads; line = i >> 1; }
		bool hit[64];
		int hits = 0;
		for (int l = 0; l < n; ++l) {
			hit[l] = alive[l] && slab(N[i], rays[l], D);
			hits += hit[l];
		}
		(coherent ? ev.nodes_coherent : ev.nodes_mixed) += 1;
		ev.lane_hits += hits;
		ev.lane_alive += live;
		if (!hits) { i += N[i].skip; continue; }
		if (N[i].skip == 1) {
			if (hits >= batch_below) {
				++ev.spot_leaves;
				for (int l = 0; l < n; ++l) if (hit[l]) {
					float dd, s, t, pp[3];
					if (tri((*T.tris)[N[i].leaf], rays[l], &dd, &s, &t, pp)) { alive[l] = false; --live; ++ev.occluded; }
				}
			} else {
				append_leaf_hits(N[i].leaf, hit, hits);
				while (waiting.size() >= 64) run_batch(64);
			}
			++i;
			continue;
		}
		if (fat_k && T.leaves_under[i] <= (uint32_t) fat_k && hits < batch_below) {
			// a fat leaf reached by few lanes: (lane, fat leaf) pairs, skip the subtree
			++ev.fat_appends;
			for (int l = 0; l < n; ++l) if (hit[l]) fat_waiting.push_back({ l, (uint32_t) i });
			ev.fat_pairs += hits;
			while (fat_waiting.size() >= 64) run_fat_batch(64);
			i += N[i].skip;
			continue;
		}
		++i;
	}
	}
	if (live) {
		if (!fat_waiting.empty()) run_fat_batch(fat_waiting.size());
		while (waiting.size() >= 64) run_batch(64);
		if (!waiting.empty()) run_batch(waiting.size());
	}
}

static unsigned long long own_walk(const Tree &T, const R &r, float D) {
	const std::vector<NodeRec> &N = *T.nodes;
	unsigned long long v = 0;
	for (size_t i = 0; i < N.size();) {
		++v;
		if (slab(N[i], r, D)) {
			if (N[i].skip == 1) { float dd, s, t, pp[3]; if (tri((*T.tris)[N[i].leaf], r, &dd, &s, &t, pp)) break; }
			++i;
		} else i += N[i].skip;
	}
	return v;
}

// Flattens every maximal subtree of at most K leaves: its inner nodes go, its leaves become the children of its root.
static std::vector<NodeRec> flatten_fat(const std::vector<NodeRec> &in, uint32_t K, std::vector<uint32_t> &leaves_under) {
	std::vector<uint32_t> lu(in.size(), 0);
	for (size_t i = in.size(); i-- > 0;) {
		if (in[i].skip == 1) { lu[i] = 1; continue; }
		uint32_t s = 0;
		for (size_t c = i + 1; c < i + in[i].skip; c += in[c].skip) s += lu[c];
		lu[i] = s;
	}
	std::vector<NodeRec> out;
	struct Open { size_t end, at; };
	std::vector<Open> parents;
	auto close = [&](size_t upto) {
		while (!parents.empty() && parents.back().end <= upto) {
			out[parents.back().at].skip = (uint32_t) (out.size() - parents.back().at);
			parents.pop_back();
		}
	};
	size_t fat_end = 0;  // inside a fat subtree up to here: inner nodes are dropped
	for (size_t i = 0; i < in.size(); ++i) {
		close(i);
		const NodeRec &n = in[i];
		if (i < fat_end && n.skip > 1) continue;
		out.push_back(n);
		leaves_under.push_back(lu[i]);
		if (n.skip > 1) {
			parents.push_back({ i + n.skip, out.size() - 1 });
			if (i >= fat_end && K && lu[i] <= K) fat_end = i + n.skip;
		}
	}
	close(in.size());
	return out;
}

int main(int argc, char **argv) {
	Mesh m;
	load_off_mesh(argv[1], &m);
	compute_vertex_normals(&m);
	BVH bvh(BVH::Method::CUT_LONGEST_AXIS);
	bvh.buildBVH(m);
	auto sf = sort_faces_by_leaf_order(m, bvh);
	PackedScene P = pack_scene(sf, bvh.nodes, bvh.aabbs, m.vertices, m.vnormals);
	const int W = argc > 2 ? atoi(argv[2]) : 1920, H = argc > 3 ? atoi(argv[3]) : 1080;
	const int stride = argc > 4 ? atoi(argv[4]) : 1;
	const float D = argc > 5 ? (float) atof(argv[5]) : 0.2f;
	const int batch_below = 48;
	auto table = uniform_ao_table(3, 4, 90);
	const int ND = (int) table.size() / 4;
	const float a = 1.0f * (W > H ? W : H);
	const size_t N = P.nodes.size();
	printf("walk tree: %zu nodes, %zu leaves, rebuilt %d\n", N, P.tris.size(), (int) P.rebuilt);

	const int FATS[] = { 0, 2, 4, 8, 16 };
	const int NF = 5;
	std::vector<NodeRec> fat_nodes[NF];
	Tree trees[NF];
	for (int f = 0; f < NF; ++f) {
		fat_nodes[f] = flatten_fat(P.nodes, (uint32_t) FATS[f], trees[f].leaves_under);
		trees[f].nodes = &fat_nodes[f];
		trees[f].tris = &P.tris;
		printf("fat %2d: %zu nodes\n", FATS[f], fat_nodes[f].size());
	}
	// classes: 0 = plane only (every hit on the largest leaf's plane), 1 = model (no hit on the plane), 2 = both
	// variants: [packets: current, dense, octant] x [fat: 0, 2, 4, 8, 16]
	const int NP = 3;
	static Events ev[3][NP][NF];
	static Events entry_ev[3], trim_ev[3], pdir_ev[3], pcull_ev[3], pcut_ev[3][3];
	static unsigned long long tiles_in[3], jobs_in[3][NP], hits_in[3], partial_tiles[3];
	// the ground plane: the two largest triangles
	float plane_y = -0.48f;
	{
		double best = -1;
		for (const TriRec &t : P.tris) {
			const double ar = std::sqrt((double) t.n[0] * t.n[0] + (double) t.n[1] * t.n[1] + (double) t.n[2] * t.n[2]);
			if (ar > best) { best = ar; plane_y = t.ta[1]; }
		}
	}
#pragma omp parallel
	{
		static thread_local Events lev[3][NP][NF];
		static thread_local Events lentry[3], ltrim[3], lpdir[3], lpcull[3], lpcut[3][3];
		unsigned long long ltiles[3] = { 0 }, ljobs[3][NP] = { { 0 } }, lhits[3] = { 0 }, lpartial[3] = { 0 };
#pragma omp for schedule(dynamic, 1)
		for (int ty = 0; ty < (H + 7) / 8; ty += stride)
			for (int tx = 0; tx < (W + 7) / 8; tx += stride) {
				float hp[64][3], hn[64][3];
				int nh = 0, on_plane = 0;
				for (int l = 0; l < 64; ++l) {
					const int x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
					if (x >= W || y >= H) continue;
					R r;
					r.o[0] = 0; r.o[1] = 0; r.o[2] = 2;
					float d[3] = { (x + 0.5f) / a - W / (2.0f * a), -((y + 0.5f) / a - H / (2.0f * a)), -1.0f };
					float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
					for (int k = 0; k < 3; ++k) { r.d[k] = d[k] / len; r.inv[k] = 1.0f / r.d[k]; }
					float best = INFINITY, bs = 0, bt = 0, bp[3] = { 0, 0, 0 };
					unsigned bl = 0;
					bool hit = false;
					for (size_t i = 0; i < N;) {
						if (slab(P.nodes[i], r, 100000.0f)) {
							if (P.nodes[i].skip == 1) {
								float dd, s, t, p[3];
								if (tri(P.tris[P.nodes[i].leaf], r, &dd, &s, &t, p)) {
									if (best > dd || (best == dd && P.nodes[i].leaf < bl)) { best = dd; bs = s; bt = t; memcpy(bp, p, sizeof p); bl = P.nodes[i].leaf; }
									hit = true;
								}
							}
							++i;
						} else i += P.nodes[i].skip;
					}
					if (!hit) continue;
					const ShadeRec &sh = P.shade[bl];
					float b0 = 1.0f - bs - bt, n[3];
					for (int k = 0; k < 3; ++k) n[k] = (sh.n0[k] * b0 + sh.n1[k] * bs) + sh.n2[k] * bt;
					float nl = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
					for (int k = 0; k < 3; ++k) { hn[nh][k] = n[k] / nl; hp[nh][k] = bp[k]; }
					on_plane += fabsf(bp[1] - plane_y) < 1e-4f;
					++nh;
				}
				if (!nh) continue;
				const int cls = on_plane == nh ? 0 : on_plane == 0 ? 1 : 2;
				++ltiles[cls];
				lhits[cls] += nh;
				lpartial[cls] += nh < 64;
				// all rays of the tile, direction-major
				std::vector<R> all((size_t) nh * ND);
				for (int l = 0; l < nh; ++l) {
					const float *n = hn[l];
					float h[3] = { n[0], n[1], n[2] };
					float ax = fabsf(n[0]), ay = fabsf(n[1]), az = fabsf(n[2]);
					if (ax <= ay && ax <= az) h[0] = 1; else if (ay <= ax && ay <= az) h[1] = 1; else h[2] = 1;
					float bx[3] = { h[1] * n[2] - h[2] * n[1], h[2] * n[0] - h[0] * n[2], h[0] * n[1] - h[1] * n[0] };
					float l2 = sqrtf((bx[0] * bx[0] + bx[1] * bx[1]) + bx[2] * bx[2]);
					for (int k = 0; k < 3; ++k) bx[k] /= l2;
					float bz[3] = { bx[1] * n[2] - bx[2] * n[1], bx[2] * n[0] - bx[0] * n[2], bx[0] * n[1] - bx[1] * n[0] };
					float l3 = sqrtf((bz[0] * bz[0] + bz[1] * bz[1]) + bz[2] * bz[2]);
					for (int k = 0; k < 3; ++k) bz[k] /= l3;
					for (int q = 0; q < ND; ++q) {
						R &r = all[(size_t) q * nh + l];
						r.h = l; r.k = q;
						for (int k = 0; k < 3; ++k) {
							r.o[k] = hp[l][k] + n[k] * 1e-5f;
							r.d[k] = (bx[k] * table[4 * q] + n[k] * table[4 * q + 1]) + bz[k] * table[4 * q + 2];
							r.inv[k] = 1.0f / r.d[k];
						}
					}
				}
				for (int f = 0; f < NF; ++f) {
					// current: pieces of `chunk` directions
					{
						int lg = 0;
						while ((2 << lg) <= nh) ++lg;
						const int chunk = 64 >> lg;
						for (int q0 = 0; q0 < ND; q0 += chunk) {
							const int dirs = std::min(chunk, ND - q0), total = dirs * nh;
							for (int base = 0; base < total; base += 64)
								walk_packet(trees[f], &all[(size_t) q0 * nh + base], std::min(64, total - base), D, batch_below, FATS[f], lev[cls][0][f]);
						}
						if (f == 0) ljobs[cls][0] += 4;  // (a tile's table is built by the four waves of the workgroup)
					}
					// dense
					for (size_t base = 0; base < all.size(); base += 64)
						walk_packet(trees[f], &all[base], (int) std::min<size_t>(64, all.size() - base), D, batch_below, FATS[f], lev[cls][1][f]);
					// octant-sorted
					{
						std::vector<R> sorted = all;
						std::stable_sort(sorted.begin(), sorted.end(), [](const R &x, const R &y) {
							const int ox = (x.inv[0] >= 0) | (x.inv[1] >= 0) << 1 | (x.inv[2] >= 0) << 2;
							const int oy = (y.inv[0] >= 0) | (y.inv[1] >= 0) << 1 | (y.inv[2] >= 0) << 2;
							return ox < oy;
						});
						for (size_t base = 0; base < sorted.size(); base += 64)
							walk_packet(trees[f], &sorted[base], (int) std::min<size_t>(64, sorted.size() - base), D, batch_below, FATS[f], lev[cls][2][f]);
					}
				}
				for (const R &r : all) lev[cls][0][0].own_nodes += own_walk(trees[0], r, D);
				{
					// ENTRY: the deepest node whose subtree holds every leaf a ray of this tile can reach (the box of the
					// tile's ray origins grown by D): descend from the root while exactly one child's box meets that region
					float qlo[3] = { INFINITY, INFINITY, INFINITY }, qhi[3] = { -INFINITY, -INFINITY, -INFINITY };
					for (int l = 0; l < nh; ++l) for (int k = 0; k < 3; ++k) { qlo[k] = fminf(qlo[k], hp[l][k] - D * 1.002f - 1e-4f); qhi[k] = fmaxf(qhi[k], hp[l][k] + D * 1.002f + 1e-4f); }
					const std::vector<NodeRec> &N0 = *trees[0].nodes;
					size_t at = 0;
					for (;;) {
						if (N0[at].skip == 1) break;
						size_t chosen = 0; int count = 0;
						for (size_t c = at + 1; c < at + N0[at].skip; c += N0[c].skip) {
							bool over = true;
							for (int k = 0; k < 3; ++k) over = over && !(N0[c].lo[k] > qhi[k] || N0[c].hi[k] < qlo[k]);
							if (over) { ++count; chosen = c; }
						}
						if (count != 1) break;
						at = chosen;
					}
					int lg = 0;
					while ((2 << lg) <= nh) ++lg;
					const int chunk = 64 >> lg;
					for (int q0 = 0; q0 < ND; q0 += chunk) {
						const int dirs = std::min(chunk, ND - q0), total = dirs * nh;
						for (int base = 0; base < total; base += 64)
							walk_packet(trees[0], &all[(size_t) q0 * nh + base], std::min(64, total - base), D, batch_below, 0, lentry[cls], at, at + N0[at].skip);
					}
					// TRIM: the walk goes through a pre-order INTERVAL, which need not be a subtree: from the entry on, drop a node
					// (and the ancestors' tests) while its first child is clear of the region, cut the tail while the last child is
					auto meets = [&](size_t c) {
						bool over = true;
						for (int k = 0; k < 3; ++k) over = over && !(N0[c].lo[k] > qhi[k] || N0[c].hi[k] < qlo[k]);
						return over;
					};
					size_t begin = at, end = at + N0[at].skip;
					if (N0[at].skip > 1) {
						size_t n = at;
						for (;;) {  // left
							if (N0[n].skip == 1) break;
							size_t first = 0, index = 0, others = 0;
							for (size_t c = n + 1; c < n + N0[n].skip; c += N0[c].skip) {
								if (first) others += meets(c);
								else { ++index; if (meets(c)) first = c; }
							}
							if (!first) { n = n + N0[n].skip; break; }
							if (index == 1 && others) break;
							n = first;
						}
						begin = n;
						n = at;
						for (;;) {  // right
							if (N0[n].skip == 1) break;
							size_t last = 0;
							for (size_t c = n + 1; c < n + N0[n].skip; c += N0[c].skip)
								if (meets(c)) last = c;
							if (!last) { end = n; break; }
							end = last + N0[last].skip;
							n = last;
						}
						if (begin > end) begin = end;
					}
					for (int q0 = 0; q0 < ND; q0 += chunk) {
						const int dirs = std::min(chunk, ND - q0), total = dirs * nh;
						for (int base = 0; base < total && begin < end; base += 64)
							walk_packet(trees[0], &all[(size_t) q0 * nh + base], std::min(64, total - base), D, batch_below, 0, ltrim[cls], begin, end);
					}
					// PDIR: the same interval, found per PACKET for the box of its rays' segments o ... o + D d (a table of
					// tiles x direction chunks made at upload)
					for (int q0 = 0; q0 < ND; q0 += chunk) {
						const int dirs = std::min(chunk, ND - q0), total = dirs * nh;
						for (int base = 0; base < total; base += 64) {
							const R *rays = &all[(size_t) q0 * nh + base];
							const int nr = std::min(64, total - base);
							float plo[3] = { INFINITY, INFINITY, INFINITY }, phi[3] = { -INFINITY, -INFINITY, -INFINITY };
							for (int l = 0; l < nr; ++l)
								for (int k = 0; k < 3; ++k) {
									const float a = rays[l].o[k], b = rays[l].o[k] + rays[l].d[k] * D * 1.002f, m = D * 0.002f + 1e-4f;
									plo[k] = fminf(plo[k], fminf(a, b) - m);
									phi[k] = fmaxf(phi[k], fmaxf(a, b) + m);
								}
							static const int pray = getenv("PRAY") ? atoi(getenv("PRAY")) : 0;  // experiment: boxes around groups of 64 / PRAY rays instead of one
							auto pmeets = [&](size_t c) {
								bool over = true;
								for (int k = 0; k < 3; ++k) over = over && !(N0[c].lo[k] > phi[k] || N0[c].hi[k] < plo[k]);
								if (!over || !pray) return over;
								const int group = 64 / pray;
								for (int g0 = 0; g0 < nr; g0 += group) {
									float glo[3] = { INFINITY, INFINITY, INFINITY }, ghi[3] = { -INFINITY, -INFINITY, -INFINITY };
									for (int l = g0; l < std::min(nr, g0 + group); ++l)
										for (int k = 0; k < 3; ++k) {
											const float a = rays[l].o[k], b = rays[l].o[k] + rays[l].d[k] * D * 1.002f, m = D * 0.002f + 1e-4f;
											glo[k] = fminf(glo[k], fminf(a, b) - m);
											ghi[k] = fmaxf(ghi[k], fmaxf(a, b) + m);
										}
									bool g = true;
									for (int k = 0; k < 3; ++k) g = g && !(N0[c].lo[k] > ghi[k] || N0[c].hi[k] < glo[k]);
									if (g) return true;
								}
								return false;
							};
							size_t pb = 0, pe = N0[0].skip, n = 0;
							for (;;) {
								if (N0[n].skip == 1) break;
								size_t first = 0, index = 0, others = 0;
								for (size_t c = n + 1; c < n + N0[n].skip; c += N0[c].skip) {
									if (first) others += pmeets(c);
									else { ++index; if (pmeets(c)) first = c; }
								}
								if (!first) { n = n + N0[n].skip; break; }
								if (index == 1 && others) break;
								n = first;
							}
							pb = n;
							n = 0;
							for (;;) {
								if (N0[n].skip == 1) break;
								size_t last = 0;
								for (size_t c = n + 1; c < n + N0[n].skip; c += N0[c].skip)
									if (pmeets(c)) last = c;
								if (!last) { pe = n; break; }
								pe = last + N0[last].skip;
								n = last;
							}
							if (pb < pe) walk_packet(trees[0], rays, nr, D, batch_below, 0, lpdir[cls], pb, pe);
							else { ++lpdir[cls].packets; lpdir[cls].rays += nr; }
							walk_packet(trees[0], rays, nr, D, batch_below, 0, lpcull[cls], 0, (size_t) -1, plo, phi);
							// PCUT K: at most K intervals per packet -- from [whole array] on, the largest subtree of the list is replaced by
							// its children that meet the box (neighbours in the array merge into one interval) while the list stays within K
							for (int kk = 0; kk < 3; ++kk) {
								const size_t K = (size_t) (2 << kk);
								std::vector<size_t> cut = { 0 };  // subtree roots, in array order
								auto intervals_of = [&](const std::vector<size_t> &c) {
									std::vector<std::pair<size_t, size_t>> out;
									for (size_t node : c) {
										if (!out.empty() && out.back().second == node) out.back().second = node + N0[node].skip;
										else out.push_back({ node, node + N0[node].skip });
									}
									return out;
								};
								for (;;) {
									size_t pick = cut.size(), size = 1;
									for (size_t j = 0; j < cut.size(); ++j)
										if (N0[cut[j]].skip > size) { size = N0[cut[j]].skip; pick = j; }
									if (pick == cut.size()) break;
									std::vector<size_t> next(cut.begin(), cut.begin() + (long) pick);
									const size_t node = cut[pick];
									for (size_t c = node + 1; c < node + N0[node].skip; c += N0[c].skip)
										if (pmeets(c)) next.push_back(c);
									next.insert(next.end(), cut.begin() + (long) pick + 1, cut.end());
									if (intervals_of(next).size() > K) break;
									cut.swap(next);
									if (cut.empty()) break;
								}
								const auto ranges = intervals_of(cut);
								if (!ranges.empty()) walk_packet(trees[0], rays, nr, D, batch_below, 0, lpcut[cls][kk], 0, (size_t) -1, nullptr, nullptr, &ranges);
								else { ++lpcut[cls][kk].packets; lpcut[cls][kk].rays += nr; }
							}
						}
					}
				}
			}
#pragma omp critical
		{
			for (int c = 0; c < 3; ++c) {
				tiles_in[c] += ltiles[c];
				hits_in[c] += lhits[c];
				partial_tiles[c] += lpartial[c];
				for (int p = 0; p < NP; ++p) {
					jobs_in[c][p] += ljobs[c][p];
					for (int f = 0; f < NF; ++f) ev[c][p][f].add(lev[c][p][f]);
				}
				entry_ev[c].add(lentry[c]);
				trim_ev[c].add(ltrim[c]);
				pdir_ev[c].add(lpdir[c]);
				pcull_ev[c].add(lpcull[c]);
				for (int kk = 0; kk < 3; ++kk) pcut_ev[c][kk].add(lpcut[c][kk]);
			}
		}
	}
	const char *cname[3] = { "plane", "model", "both" }, *pname[NP] = { "current", "dense", "octant" };
	Model M;
	const double scale = (double) stride * stride;
	double total_current = 0;
	for (int c = 0; c < 3; ++c) total_current += M.cost(ev[c][0][0], tiles_in[c] * 4);
	printf("\nVALU model: setup %g, node %g / %g, spot leaf %g, append %g, batch %g, tangent frames %g per wave and tile\n", M.setup, M.coherent, M.mixed, M.spot, M.append, M.batch, M.tile_setup);
	printf("model total for the kernel as it is: %.1f M wave64 VALU (measured on the headline frame: 620.7 M in round 3, 562.7 M in round 4)\n\n", total_current * scale / 1e6);
	for (int c = 0; c < 3; ++c) {
		printf("== %s tiles: %llu (%llu partial), %llu hit sub-pixels, %.2f M rays; a ray's own walk: %.1f node tests\n", cname[c], tiles_in[c] * (unsigned long long) scale,
		       partial_tiles[c] * (unsigned long long) scale, hits_in[c] * (unsigned long long) scale, hits_in[c] * scale * ND / 1e6,
		       (double) ev[c][0][0].own_nodes / (double) (hits_in[c] * ND ? hits_in[c] * ND : 1));
		printf("%-8s %-4s %9s %6s %6s %7s %7s %6s %7s %7s %7s %7s %7s %8s %8s\n", "packets", "fat", "packets", "rays/p", "coh%", "nodes/p", "lanes%", "spot/p", "app/p", "pairs/p", "bat/p", "fatp/p", "fatb/p", "VALU/p", "M VALU");
		if (entry_ev[c].packets) {
			const Events &e = entry_ev[c];
			printf("ENTRY    dependent loads per packet with 2 / 3 / 4 / 6 / 8 consecutive nodes per load: %.1f / %.1f / %.1f / %.1f / %.1f\n", e.loads[0] / (double) e.packets,
			       e.loads[1] / (double) e.packets, e.loads[2] / (double) e.packets, e.loads[3] / (double) e.packets, e.loads[4] / (double) e.packets);
			printf("ENTRY    of the kernel's loads %.1f %% begin at an odd node (two lines); fetching the aligned line of the node wanted instead: %.1f loads per packet\n",
			       100.0 * e.odd_loads / (double) e.loads[0], e.line_loads / (double) e.packets);
			const double pk = (double) e.packets, cost = M.cost(e, tiles_in[c] * 4);
			printf("%-8s %-4d %9.0f %6.1f %6.1f %7.1f %7.1f %6.2f %7.1f %7.1f %7.2f %7.1f %7.2f %8.0f %8.1f\n", "ENTRY", 0, pk * scale, e.rays / pk,
			       100.0 * e.coherent_packets / pk, (e.nodes_coherent + e.nodes_mixed) / pk, 100.0 * e.lane_hits / (double) (e.lane_alive ? e.lane_alive : 1), e.spot_leaves / pk,
			       e.appends / pk, e.pairs / pk, e.batches / pk, 0.0, 0.0, cost / pk, cost * scale / 1e6);
		}
		if (trim_ev[c].packets) {
			const Events &e = trim_ev[c];
			const double pk = (double) e.packets, cost = M.cost(e, tiles_in[c] * 4);
			printf("%-8s %-4d %9.0f %6.1f %6.1f %7.1f %7.1f %6.2f %7.1f %7.1f %7.2f %7.1f %7.2f %8.0f %8.1f\n", "TRIM", 0, pk * scale, e.rays / pk,
			       100.0 * e.coherent_packets / pk, (e.nodes_coherent + e.nodes_mixed) / pk, 100.0 * e.lane_hits / (double) (e.lane_alive ? e.lane_alive : 1), e.spot_leaves / pk,
			       e.appends / pk, e.pairs / pk, e.batches / pk, 0.0, 0.0, cost / pk, cost * scale / 1e6);
		}
		if (pdir_ev[c].packets) {
			const Events &e = pdir_ev[c];
			const double pk = (double) e.packets, cost = M.cost(e, tiles_in[c] * 4);
			printf("%-8s %-4d %9.0f %6.1f %6.1f %7.1f %7.1f %6.2f %7.1f %7.1f %7.2f %7.1f %7.2f %8.0f %8.1f\n", "PDIR", 0, pk * scale, e.rays / pk,
			       100.0 * e.coherent_packets / pk, (e.nodes_coherent + e.nodes_mixed) / pk, 100.0 * e.lane_hits / (double) (e.lane_alive ? e.lane_alive : 1), e.spot_leaves / pk,
			       e.appends / pk, e.pairs / pk, e.batches / pk, 0.0, 0.0, cost / pk, cost * scale / 1e6);
		}
		if (pcull_ev[c].packets) {
			const Events &e = pcull_ev[c];
			const double pk = (double) e.packets, cost = M.cost(e, tiles_in[c] * 4);
			printf("%-8s %-4d %9.0f %6.1f %6.1f %7.1f %7.1f %6.2f %7.1f %7.1f %7.2f %7.1f %7.2f %8.0f %8.1f\n", "PCULL", 0, pk * scale, e.rays / pk,
			       100.0 * e.coherent_packets / pk, (e.nodes_coherent + e.nodes_mixed) / pk, 100.0 * e.lane_hits / (double) (e.lane_alive ? e.lane_alive : 1), e.spot_leaves / pk,
			       e.appends / pk, e.pairs / pk, e.batches / pk, 0.0, 0.0, cost / pk, cost * scale / 1e6);
		}
		for (int kk = 0; kk < 3; ++kk)
			if (pcut_ev[c][kk].packets) {
				const Events &e = pcut_ev[c][kk];
				const double pk = (double) e.packets, cost = M.cost(e, tiles_in[c] * 4);
				printf("%-8s %-4d %9.0f %6.1f %6.1f %7.1f %7.1f %6.2f %7.1f %7.1f %7.2f %7.1f %7.2f %8.0f %8.1f\n", kk == 0 ? "PCUT2" : kk == 1 ? "PCUT4" : "PCUT8", 0, pk * scale, e.rays / pk,
				       100.0 * e.coherent_packets / pk, (e.nodes_coherent + e.nodes_mixed) / pk, 100.0 * e.lane_hits / (double) (e.lane_alive ? e.lane_alive : 1), e.spot_leaves / pk,
				       e.appends / pk, e.pairs / pk, e.batches / pk, 0.0, 0.0, cost / pk, cost * scale / 1e6);
			}
		for (int p = 0; p < NP; ++p)
			for (int f = 0; f < NF; ++f) {
				const Events &e = ev[c][p][f];
				if (!e.packets) continue;
				const double pk = (double) e.packets;
				const double cost = M.cost(e, tiles_in[c] * 4);
				printf("%-8s %-4d %9.0f %6.1f %6.1f %7.1f %7.1f %6.2f %7.1f %7.1f %7.2f %7.1f %7.2f %8.0f %8.1f\n", pname[p], FATS[f], pk * scale, e.rays / pk,
				       100.0 * e.coherent_packets / pk, (e.nodes_coherent + e.nodes_mixed) / pk, 100.0 * e.lane_hits / (double) (e.lane_alive ? e.lane_alive : 1), e.spot_leaves / pk,
				       e.appends / pk, e.pairs / pk, e.batches / pk, e.fat_pairs / pk, e.fat_batches / pk, cost / pk, cost * scale / 1e6);
			}
	}
	return 0;
}
