// kernels/walk.hip.h -- the shared walk: the hand-scheduled node loop, the exact leaf gate, the any-hit walk of a packet
// (part of the one translation unit kernels.hip; see its head for the passes and the arithmetic contract)
#pragma once
#include "common.hip.h"

namespace ocrt {

// ---------------------------------------------------------------------------
// Shared walk.  The 64 rays of a wave visit the union of their nodes together:
// one wave-uniform node index `at`, so a node (and a leaf's triangle) arrives by
// scalar loads and the box is tested out of SGPRs; no per-lane index, no gathers,
// no scheduling.  Needs sibling subtrees to tile their parent's index range (KernelParams::shared_walk;
// checked at upload for the uploaded binary tree and again for the rebuilt, possibly wider one).
//
// Fast form (`exact` false: regular scene with nested boxes, regular rays): every
// live lane tests every visited box.  A lane that missed an ancestor also misses
// each box nested in it -- (lo-o)*inv and (hi-o)*inv are monotone in lo and hi, so
// near can only grow and far only shrink -- hence a lane accepts exactly the
// triangles of its own walk, in the same ascending leaf order.
//
// Exact form: each lane also keeps `mine`, the next node of its OWN walk, tests a
// box only when at == mine, and uses the reference's select-based slab test: lane
// by lane that is the reference's walk (src/intersect_kernel.cl:184-213) of the node
// array the kernel was given, whatever the boxes and rays hold.  For damaged scene
// arrays that array is the uploaded one, so this IS the reference walk.  On a regular,
// nested scene it may be the rebuilt tree (scene_pack.cc); the form is then only taken by
// packets holding a ray that is not "selectable" -- a NaN direction or origin (zero-length
// vertex normals), all three reciprocals infinite -- and such a ray fails the slab test
// at the ROOT of any tree (`NaN < max_distance` is false, reference :60), so it tests no
// triangle in either tree, while its selectable neighbours get, lane by lane, the
// monotone slab test on nested boxes, for which the tree does not matter.
// tests/test_hip_parity.py::test_zero_normals_on_a_rebuilt_tree covers it.
// `at` never overtakes a live lane's `mine` because subtree ranges nest.
// ---------------------------------------------------------------------------
// One node of the exact form for a lane: the reference's slab test where the lane's own walk stands (`mine`).
__device__ __forceinline__ bool exact_box(const float4 lo, const float4 hi, const Ray &ray, float max_distance, bool alive,
                                          uint32_t at, uint32_t skip, uint32_t &mine) {
	const bool here = alive && mine == at;
	const bool box = here && slab_hit(lo, hi, ray, max_distance);
	mine = here ? (box ? at + 1u : at + skip) : mine;
	return box;
}

// Per packet: which lanes have a reciprocal direction >= 0 on each axis (the reference's
// `inv >= 0 ? lo : hi` choice of the near plane, made once instead of at every node).
struct SignMasks {
	unsigned long long x, y, z;
};
__device__ __forceinline__ SignMasks sign_masks(const Ray &r) {
	SignMasks m;
	m.x = wave_ballot(r.ix >= 0.0f);
	m.y = wave_ballot(r.iy >= 0.0f);
	m.z = wave_ballot(r.iz >= 0.0f);
	return m;
}

// A node record by one scalar load (asm for the reason given at walk_collect: a plain load through
// nodes_ptr next to that loop makes the compiler keep the pointer in VGPRs).
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32x8 scalar_load_node(const float4 *nodes_ptr, uint32_t at) {
	u32x8 r;
	const uint32_t offset = at * 32u;
	asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(nodes_ptr), "s"(offset));
	return r;
}

// The node steps of the fast form, hand-scheduled.  From byte offset `at` on it walks the packet through the
// PADDED copy of the tree (scene_pack.cc, pad_walk_boxes): per node a conservative slab test on values fetched by a
// scalar load -- t = fma(plane, inv, oi) with oi = -(o * inv) rounded once, near/far planes picked by the sign of
// inv, max(.., tiny = bit pattern 1), min(.., below), near <= far -- and a scalar decision: some live lane hit ->
// first child, nobody -> skip the subtree.  The outward margin of the boxes makes up for the fma's rounding
// (proof at padded_bound), so a lane passes every box the reference's own test (src/intersect_kernel.cl:21-61)
// would let it pass, and possibly a few more: the walk only finds CANDIDATE leaves, the exact test on the leaf's
// own box is the caller's (exact_leaf_gate).  Zero direction components enter with +-2^100 for the infinite
// reciprocal (WalkRay); should a NaN still arise, v_max3 / v_min3 drop it (IEEE maxNum / minNum, the kernel runs
// with IEEE mode on) -- one constraint fewer, conservative.
//
// Primary packets are sign-coherent -- every live lane's reciprocal direction has the same sign on each axis -- but on
// the image's centre lines: the near and far plane of each axis are then known when the loop is entered and the test is
// 6 v_fma + max + min + max3 + min3 + cmp = 11 vector instructions; the loop exists once per sign octant
// (OCRT_WALK_COHERENT).  Mixed packets select per lane with v_cndmask on the sign masks: 17 (the first generation of
// this loop computed (b - o) * inv exactly: 23).  The any-hit rays of the ambient-occlusion pass, whose max_distance is
// one number per frame, take the SCALED form of the test on centre / half-extent records: 12, one loop for every packet
// (OCRT_TEST_CE_SCALED below).  What an instruction costs here (tools/microbench/
// valu_rate_probe.hip, 8 waves per SIMD): ~2.3 cycles per SIMD for v_fma / v_mul / v_add / v_mov on registers, ~4.2
// for everything else (min / max / max3 / cmp / cndmask, v_pk_fma_f32, and alone also an fma with a scalar operand):
// the 11-instruction test runs at 35.7 cycles, the 9-instruction one at 28.3, a v_pk_fma_f32 version with 8 at 35.9.
//
// One 64-byte load fetches a node and its pre-order successor: after a hit on an inner node its first child is
// tested straight from s[56:63].  The array ends in two END records whose infinite box every live lane "hits" and
// whose leaf field says WALK_END, so the loop needs no bounds check.  Scalar instructions per node: load, wait,
// s_and (sets SCC), branch, add = 5 on a miss.
// At a leaf hit by fewer than `batch_below` lanes it does not stop but appends the (lane, leaf) pairs to the
// wave's list in LDS (entry = leaf | lane << 26 at index waiting + rank of the lane among the hitters) and walks
// on; `leaf_stops` counts the leaves some lane hit.
// Returns 0: walk over; 1: `leaf` is hit by many lanes (`hit`), test it now, `at` is on it; 2: 64 or more pairs
// are waiting, run a batch, `at` is on the leaf appended last.
// Scratch: s[42:63], v56-v62; only scalar outputs, so that the compiler knows the results to be wave-uniform.
// The pointer operand must not be dereferenced by plain loads elsewhere in the same kernel: the compiler then
// keeps it in VGPRs and cannot hand it to this operand.
// gfx950 hazards checked by hand (the assembler inserts nothing inside inline asm): v_cmp writes VCC -> s_and_b64
// reads it (SALU reads of VALU-written SGPRs are interlocked); s_mov_b64 exec -> ds_write_b32 / following VALU
// (EXEC writes by SALU are interlocked for vector and LDS instructions); s_load -> s_waitcnt lgkmcnt(0) before the
// first use (also drains the ds_write of an append, harmless); no v_readlane / v_div_fmas / VMEM-with-SGPR-address
// consumers of VALU-written SGPRs in here.  The build fails if the kernels using this loop spill vector registers
// or leave 8 waves per SIMD (tools/check_kernel_resources.py).
#define OCRT_TEST_COHERENT(NX, NY, NZ, FX, FY, FZ) \
	"\tv_fma_f32 v56, " NX ", %[ix], %[oix]\n"     \
	"\tv_fma_f32 v57, " NY ", %[iy], %[oiy]\n"     \
	"\tv_fma_f32 v58, " NZ ", %[iz], %[oiz]\n"     \
	"\tv_fma_f32 v59, " FX ", %[ix], %[oix]\n"     \
	"\tv_fma_f32 v60, " FY ", %[iy], %[oiy]\n"     \
	"\tv_fma_f32 v61, " FZ ", %[iz], %[oiz]\n"     \
	"\tv_max_f32 v58, 1, v58\n"                    \
	"\tv_min_f32 v61, %[below], v61\n"             \
	"\tv_max3_f32 v56, v56, v57, v58\n"            \
	"\tv_min3_f32 v59, v59, v60, v61\n"            \
	"\tv_cmp_le_f32 vcc, v56, v59\n"
#define OCRT_TEST_MIXED(LX, LY, LZ, HX, HY, HZ)    \
	"\tv_fma_f32 v56, " LX ", %[ix], %[oix]\n"     \
	"\tv_fma_f32 v57, " HX ", %[ix], %[oix]\n"     \
	"\tv_fma_f32 v58, " LY ", %[iy], %[oiy]\n"     \
	"\tv_fma_f32 v59, " HY ", %[iy], %[oiy]\n"     \
	"\tv_fma_f32 v60, " LZ ", %[iz], %[oiz]\n"     \
	"\tv_fma_f32 v61, " HZ ", %[iz], %[oiz]\n"     \
	"\tv_cndmask_b32 v62, v57, v56, %[px]\n"       \
	"\tv_cndmask_b32 v56, v56, v57, %[px]\n"       \
	"\tv_cndmask_b32 v57, v59, v58, %[py]\n"       \
	"\tv_cndmask_b32 v58, v58, v59, %[py]\n"       \
	"\tv_cndmask_b32 v59, v61, v60, %[pz]\n"       \
	"\tv_cndmask_b32 v60, v60, v61, %[pz]\n"       \
	"\tv_max_f32 v59, 1, v59\n"                    \
	"\tv_min_f32 v60, %[below], v60\n"             \
	"\tv_max3_f32 v62, v62, v57, v59\n"            \
	"\tv_min3_f32 v56, v56, v58, v60\n"            \
	"\tv_cmp_le_f32 vcc, v62, v56\n"
// The SCALED form (any-hit rays, whose max_distance is one number per frame): the reciprocals carry a factor
// ~ 1 / max_distance, so "t < max_distance" reads "t' <= 1" and both limits fit the CLAMP modifier of the z-axis fmas
// (clamp to [0, 1]): near = max3(x, y, clamp(z)), far = min3(x, y, clamp(z)), hit iff near < far.  The comparison is
// strict so that a box behind the origin on z (far clamped to 0, near >= 0) fails; why no pair the reference accepts is
// lost to that: scene_pack.cc, padded_bound ("The scaled form").
// It reads the CENTRE / HALF-EXTENT copy of the walk array (scene_pack.cc, ce_record: c in the lo fields, e in the hi
// fields; the copy lies behind the plane form's records and their END records):
// t_c = fma(c, inv, oi), near = fma(-e, |inv|, t_c), far = fma(e, |inv|, t_c) -- right for either sign of inv, so ONE loop
// serves every any-hit packet: 9 v_fma + max3 + min3 + cmp = 12 vector instructions, nine of them of the fast class.
// (Rounds 2-3 walked the plane-form records here too: a loop per sign octant, 6 fma + max3 + min3 + cmp = 9 per node, and
// a select form of 15 -- nine of them of the slow class -- for packets whose rays disagree on a sign, a third of the
// bunny's model packets.  The one loop measures 1.3 ... 4.4 % faster per frame on every workload, although coherent
// packets execute three instructions more per node: fast-class fmas, one array in the caches, a ninth of the code.)
// Conservative like the plane form (the half-extent carries the rounding of t_c: proof at ce_record).
#define OCRT_TEST_CE_SCALED(CX, CY, CZ, EX, EY, EZ)       \
	"\tv_fma_f32 v56, " CX ", %[ix], %[oix]\n"           \
	"\tv_fma_f32 v57, " CY ", %[iy], %[oiy]\n"           \
	"\tv_fma_f32 v58, " CZ ", %[iz], %[oiz]\n"           \
	"\tv_fma_f32 v59, -" EX ", |%[ix]|, v56\n"           \
	"\tv_fma_f32 v56, " EX ", |%[ix]|, v56\n"            \
	"\tv_fma_f32 v60, -" EY ", |%[iy]|, v57\n"           \
	"\tv_fma_f32 v57, " EY ", |%[iy]|, v57\n"            \
	"\tv_fma_f32 v61, -" EZ ", |%[iz]|, v58 clamp\n"     \
	"\tv_fma_f32 v58, " EZ ", |%[iz]|, v58 clamp\n"      \
	"\tv_max3_f32 v59, v59, v60, v61\n"                  \
	"\tv_min3_f32 v56, v56, v57, v58\n"                  \
	"\tv_cmp_lt_f32 vcc, v59, v56\n"
// (LEAF: the s-register holding the node's leaf field; NEXT: where the walk goes on after an append)
#define OCRT_WALK_LEAF(LEAF, NOW, NEXT)                 \
	"\ts_cmp_eq_u32 " LEAF ", -2\n"                     \
	"\ts_cbranch_scc1 .Lw_over_%=\n"                    \
	"\ts_bcnt1_i32_b64 s46, s[44:45]\n"                 \
	"\ts_add_u32 %[stops], %[stops], 1\n"               \
	"\ts_cmp_ge_u32 s46, %[batch_below]\n"              \
	"\ts_cbranch_scc1 " NOW "\n"                        \
	"\tv_mbcnt_lo_u32_b32 v56, s44, 0\n"                \
	"\tv_mbcnt_hi_u32_b32 v56, s45, v56\n"              \
	"\tv_add_u32 v56, %[waiting], v56\n"                \
	"\tv_lshl_add_u32 v56, v56, 2, %[list]\n"           \
	"\tv_or_b32 v57, " LEAF ", %[tag]\n"                \
	"\ts_mov_b64 s[42:43], exec\n"                      \
	"\ts_mov_b64 exec, s[44:45]\n"                      \
	"\tds_write_b32 v56, v57\n"                         \
	"\ts_mov_b64 exec, s[42:43]\n"                      \
	"\ts_add_u32 %[waiting], %[waiting], s46\n"         \
	"\ts_cmp_ge_u32 %[waiting], 64\n"                   \
	"\ts_cbranch_scc1 .Lw_full_%=\n"                    \
	"\ts_branch " NEXT "\n"
// PF_B: what the loop does between the tests of a pair, once `a` is known to be hit.  OCRT_PF_SUCCESSORS touches -- with
// one-dword scalar loads nobody reads -- both places the walk can go to after `b`: the line behind the pair and b's skip
// target.  One of the two is the next load, which then finds its line in the scalar cache or on its way instead of
// starting a round trip of its own: the walk is a chain of dependent loads, and this takes the test of `b` out of the
// chain.  It pays where packets mostly descend (the bunny's model tiles: ambient-occlusion pass -2 %, frames in flight
// -2.5 ... -3.7 %) and costs where they mostly miss (the interior scene: +1 ... 2 %; the useless one of the two loads is
// waited for by the next s_waitcnt all the same) -- so a render host can be told which form to launch
// (DeviceRenderer::setAoPrefetch; a frame ring measures both on its scene at upload).  Forms that touch the skip target of
// `a` before its test, or the next line alone, measured worse on one side or the other (profiles/r04_notes.md).
#define OCRT_PF_NONE ""
// Only where `b` is an inner node: then every path from here leads to the loop's next load and its s_waitcnt lgkmcnt(0),
// which also waits for these two.  Behind a leaf the loop may be LEFT (a leaf stop, a full list) -- with a load still on
// its way to s47, a register the compiler is free to use again the moment the asm block ends: it would be overwritten
// whenever the load lands.  (That was the first form of this; a 20 k-triangle height field showed it, the bunny did not.)
#define OCRT_PF_SUCCESSORS                                    \
	"\ts_cmp_lg_u32 s63, -1\n"                               \
	"\ts_cbranch_scc1 .Lw_no_pf_%=\n"                        \
	"\ts_add_u32 s46, %[at], 32\n"                           \
	"\ts_load_dword s47, %[base], s46\n"                     \
	"\ts_add_u32 s46, %[at], s59\n"                          \
	"\ts_load_dword s47, %[base], s46\n"                     \
	".Lw_no_pf_%=:\n"
// HEAD: what is checked while a pair is being fetched.  OCRT_HEAD_END: the walk is over once `at` has left the range it
// was given (the any-hit walks of a tile run through ONE subtree, its entry: ao_kernel) -- two scalar instructions per
// pair, in the shadow of the load (which is waited for on the way out too: its sixteen registers are the compiler's again
// once the block ends; the records behind any range are there to be read, END records at the latest).
#define OCRT_HEAD_NONE ""
#define OCRT_HEAD_END "\ts_cmp_ge_u32 %[at], %[end]\n\ts_cbranch_scc1 .Lw_over_wait_%=\n"
#define OCRT_WALK_ASM(HEAD, TEST_A, TEST_B, PF_B)           \
	"\ts_branch .Lw_node_%=\n"                              \
	".Lw_miss_a_%=:\n"                                      \
	"\ts_add_u32 %[at], %[at], s51\n"                       \
	".Lw_node_%=:\n"                                        \
	"\ts_load_dwordx16 s[48:63], %[base], %[at]\n"          \
	HEAD                                                    \
	"\ts_waitcnt lgkmcnt(0)\n"                              \
	TEST_A                                                  \
	"\ts_and_b64 s[44:45], vcc, %[alive]\n"                 \
	"\ts_cbranch_scc0 .Lw_miss_a_%=\n"                      \
	"\ts_cmp_lg_u32 s55, -1\n"                              \
	"\ts_cbranch_scc1 .Lw_leaf_a_%=\n"                      \
	".Lw_next_b_%=:\n"                                      \
	"\ts_add_u32 %[at], %[at], 32\n"                        \
	PF_B                                                    \
	TEST_B                                                  \
	"\ts_and_b64 s[44:45], vcc, %[alive]\n"                 \
	"\ts_cbranch_scc0 .Lw_miss_b_%=\n"                      \
	"\ts_cmp_lg_u32 s63, -1\n"                              \
	"\ts_cbranch_scc1 .Lw_leaf_b_%=\n"                      \
	".Lw_next_a_%=:\n"                                      \
	"\ts_add_u32 %[at], %[at], 32\n"                        \
	"\ts_branch .Lw_node_%=\n"                              \
	".Lw_miss_b_%=:\n"                                      \
	"\ts_add_u32 %[at], %[at], s59\n"                       \
	"\ts_branch .Lw_node_%=\n"                              \
	".Lw_leaf_a_%=:\n"                                      \
	OCRT_WALK_LEAF("s55", ".Lw_now_a_%=", ".Lw_next_b_%=")  \
	".Lw_now_a_%=:\n"                                       \
	"\ts_mov_b32 %[leaf], s55\n"                            \
	"\ts_branch .Lw_now_%=\n"                               \
	".Lw_leaf_b_%=:\n"                                      \
	OCRT_WALK_LEAF("s63", ".Lw_now_b_%=", ".Lw_next_a_%=")  \
	".Lw_now_b_%=:\n"                                       \
	"\ts_mov_b32 %[leaf], s63\n"                            \
	".Lw_now_%=:\n"                                         \
	"\ts_mov_b64 %[hit], s[44:45]\n"                        \
	"\ts_mov_b32 %[status], 1\n"                            \
	"\ts_branch .Lw_out_%=\n"                               \
	".Lw_full_%=:\n"                                        \
	"\ts_mov_b32 %[status], 2\n"                            \
	"\ts_branch .Lw_out_%=\n"                               \
	".Lw_over_wait_%=:\n"                                   \
	"\ts_waitcnt lgkmcnt(0)\n"                              \
	".Lw_over_%=:\n"                                        \
	"\ts_mov_b32 %[status], 0\n"                            \
	".Lw_out_%=:\n"
#define OCRT_WALK_CLOBBERS                                                                                              \
	"s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", \
	    "s59", "s60", "s61", "s62", "s63", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "vcc", "scc", "memory"
// node a = s[48:55] (lo.xyz, skip bytes, hi.xyz, leaf), node b = s[56:63]; X/Y/Z: "P" = reciprocal >= 0 on that axis
// (near plane lo), "N" = negative (near plane hi)
#define OCRT_NEAR_P(LO, HI) LO
#define OCRT_NEAR_N(LO, HI) HI
#define OCRT_FAR_P(LO, HI) HI
#define OCRT_FAR_N(LO, HI) LO
#define OCRT_WALK_COHERENT(TEST, X, Y, Z, PF)                                                                               \
	asm volatile(OCRT_WALK_ASM(OCRT_HEAD_NONE, TEST(OCRT_NEAR_##X("s48", "s52"), OCRT_NEAR_##Y("s49", "s53"),              \
	                                OCRT_NEAR_##Z("s50", "s54"), OCRT_FAR_##X("s48", "s52"),                               \
	                                OCRT_FAR_##Y("s49", "s53"), OCRT_FAR_##Z("s50", "s54")),                               \
	                           TEST(OCRT_NEAR_##X("s56", "s60"), OCRT_NEAR_##Y("s57", "s61"),                              \
	                                OCRT_NEAR_##Z("s58", "s62"), OCRT_FAR_##X("s56", "s60"),                               \
	                                OCRT_FAR_##Y("s57", "s61"), OCRT_FAR_##Z("s58", "s62")), PF)                           \
	             : [at] "+s"(at), [waiting] "+s"(waiting), [stops] "+s"(leaf_stops), [hit] "=&s"(hit_mask),               \
	               [leaf] "=&s"(leaf), [status] "=&s"(status)                                                            \
	             : [base] "s"(walk_ptr), [alive] "s"(alive_mask), [below] "v"(below), [batch_below] "s"(batch_below),     \
	               [list] "s"(list_lds_address), [tag] "v"(lane_tag), [ix] "v"(ray.ix), [iy] "v"(ray.iy), [iz] "v"(ray.iz), \
	               [oix] "v"(ray.oix), [oiy] "v"(ray.oiy), [oiz] "v"(ray.oiz)                                             \
	             : OCRT_WALK_CLOBBERS)

#define OCRT_WALK_MIXED(TEST, PF)                                                                                         \
	asm volatile(OCRT_WALK_ASM(OCRT_HEAD_NONE, TEST("s48", "s49", "s50", "s52", "s53", "s54"), TEST("s56", "s57", "s58", "s60", "s61", "s62"), PF) \
	             : [at] "+s"(at), [waiting] "+s"(waiting), [stops] "+s"(leaf_stops), [hit] "=&s"(hit_mask),               \
	               [leaf] "=&s"(leaf), [status] "=&s"(status)                                                            \
	             : [base] "s"(walk_ptr), [alive] "s"(alive_mask), [below] "v"(below), [batch_below] "s"(batch_below),     \
	               [px] "s"(sign.x), [py] "s"(sign.y), [pz] "s"(sign.z), [list] "s"(list_lds_address), [tag] "v"(lane_tag), \
	               [ix] "v"(ray.ix), [iy] "v"(ray.iy), [iz] "v"(ray.iz), [oix] "v"(ray.oix), [oiy] "v"(ray.oiy),          \
	               [oiz] "v"(ray.oiz)                                                                                    \
	             : OCRT_WALK_CLOBBERS)
#define OCRT_WALK_MIXED_CE(TEST, PF)                                                                                      \
	asm volatile(OCRT_WALK_ASM(OCRT_HEAD_END, TEST("s48", "s49", "s50", "s52", "s53", "s54"), TEST("s56", "s57", "s58", "s60", "s61", "s62"), PF) \
	             : [at] "+s"(at), [waiting] "+s"(waiting), [stops] "+s"(leaf_stops), [hit] "=&s"(hit_mask),               \
	               [leaf] "=&s"(leaf), [status] "=&s"(status)                                                            \
	             : [base] "s"(walk_ptr), [alive] "s"(alive_mask), [end] "s"(walk_end), [batch_below] "s"(batch_below),    \
	               [list] "s"(list_lds_address), [tag] "v"(lane_tag),                                                      \
	               [ix] "v"(ray.ix), [iy] "v"(ray.iy), [iz] "v"(ray.iz), [oix] "v"(ray.oix), [oiy] "v"(ray.oiy),          \
	               [oiz] "v"(ray.oiz)                                                                                    \
	             : OCRT_WALK_CLOBBERS)
#define OCRT_WALK_SWITCH(COHERENT_TEST, MIXED_TEST, PF)                 \
	switch (variant) {                                                  \
	case 0u: OCRT_WALK_COHERENT(COHERENT_TEST, N, N, N, PF); break;     \
	case 1u: OCRT_WALK_COHERENT(COHERENT_TEST, P, N, N, PF); break;     \
	case 2u: OCRT_WALK_COHERENT(COHERENT_TEST, N, P, N, PF); break;     \
	case 3u: OCRT_WALK_COHERENT(COHERENT_TEST, P, P, N, PF); break;     \
	case 4u: OCRT_WALK_COHERENT(COHERENT_TEST, N, N, P, PF); break;     \
	case 5u: OCRT_WALK_COHERENT(COHERENT_TEST, P, N, P, PF); break;     \
	case 6u: OCRT_WALK_COHERENT(COHERENT_TEST, N, P, P, PF); break;     \
	case 7u: OCRT_WALK_COHERENT(COHERENT_TEST, P, P, P, PF); break;     \
	default: MIXED_TEST; break;                                         \
	}

// `variant`: 0..7 = sign octant of a coherent packet (bit 0: x reciprocals >= 0, bit 1: y, bit 2: z), 8 = mixed.
// SCALED: `ray` was made with the frame's walk_scale and `below` is not looked at (the limit is 1.0).
constexpr uint32_t WALK_MIXED = 8u;
template <bool SCALED, bool PREFETCH = false>
__device__ __forceinline__ uint32_t walk_collect(uint32_t variant, const float4 *walk_ptr, uint32_t &at, const WalkRay &ray,
                                                 const SignMasks &sign, float below, unsigned long long alive_mask,
                                                 unsigned long long &hit_mask, uint32_t &leaf, uint32_t &waiting,
                                                 uint32_t &leaf_stops, uint32_t list_lds_address, uint32_t lane_tag,
                                                 uint32_t batch_below, uint32_t walk_end = 0xFFFFFFFFu) {
	(void) walk_end;  // (the any-hit loop's: byte offset behind the subtree it walks)
	uint32_t status;
	if (SCALED) {
		// one loop for every any-hit packet, whatever its rays' signs (the caller starts `at` in the centre / half-extent
		// copy of the array)
		(void) sign;
		(void) variant;
		if (PREFETCH) {
			OCRT_WALK_MIXED_CE(OCRT_TEST_CE_SCALED, OCRT_PF_SUCCESSORS);
		} else {
			OCRT_WALK_MIXED_CE(OCRT_TEST_CE_SCALED, OCRT_PF_NONE);
		}
	} else {
		// (the primary pass keeps the plane form and its loop per sign octant: its packets are coherent but for the
		// image's centre lines, and 11 instructions beat 14: 1-5 % of the pass)
		OCRT_WALK_SWITCH(OCRT_TEST_COHERENT, OCRT_WALK_MIXED(OCRT_TEST_MIXED, OCRT_PF_NONE), OCRT_PF_NONE)
	}
	return status;
}

// Which loop a packet takes: its sign octant if every live lane agrees on every axis, else WALK_MIXED.
__device__ __forceinline__ uint32_t walk_variant(const SignMasks &sign, unsigned long long alive_mask) {
	const unsigned long long x = sign.x & alive_mask, y = sign.y & alive_mask, z = sign.z & alive_mask;
	bool coherent = (x == 0ull || x == alive_mask) && (y == 0ull || y == alive_mask) && (z == 0ull || z == alive_mask);
#ifdef OCRT_ALWAYS_MIXED
	coherent = false;
#endif
	return coherent ? (x != 0ull ? 1u : 0u) | (y != 0ull ? 2u : 0u) | (z != 0ull ? 4u : 0u) : WALK_MIXED;
}

// The exact test on a candidate leaf's OWN box (uploaded, unpadded), the reference's gate of the triangle test
// (src/intersect_kernel.cl:189,195).  Only packets of the fast form get here -- regular boxes, selectable rays --,
// for which the reference's chain of comparisons folds into max(near, tiny) <= min(far, below) with the near / far
// plane picked by the sign of the reciprocal and IEEE maxNum / minNum dropping the NaN of 0 * inf (the
// first generation of walk_collect applied exactly this arithmetic to every node).  `below` is the largest float
// under the ray kind's max_distance.
__device__ __forceinline__ bool exact_leaf_gate(const float4 lo, const float4 hi, const Ray &r, float below) {
	const float x0 = (lo.x - r.ox) * r.ix, x1 = (hi.x - r.ox) * r.ix;
	const float y0 = (lo.y - r.oy) * r.iy, y1 = (hi.y - r.oy) * r.iy;
	const float z0 = (lo.z - r.oz) * r.iz, z1 = (hi.z - r.oz) * r.iz;
	const bool px = r.ix >= 0.0f, py = r.iy >= 0.0f, pz = r.iz >= 0.0f;
	const float tiny = __uint_as_float(1u);
	const float t_near = fmaxf(fmaxf(px ? x0 : x1, py ? y0 : y1), fmaxf(pz ? z0 : z1, tiny));
	const float t_far = fminf(fminf(px ? x1 : x0, py ? y1 : y0), fminf(pz ? z1 : z0, below));
	return t_near <= t_far;
}

// Triangle tests of an any-hit packet waiting to be run 64 at a time (LDS, one per wave).
struct LeafBatch {
	unsigned int entry[128];       // leaf | owning lane << 26; up to 63 waiting + 64 appended at one leaf
	unsigned int occluded_bits[2];  // lanes whose ray was found occluded by the batch just run
};

// The same for a closest-hit packet (primary rays).  The reference keeps the first hit in
// leaf order among the nearest (`best.distance > distance`, strict): that is the minimum of
// (distance, leaf) in lexicographic order, so the tests may run in any order and on any lane
// if each ray's minimum of key = distance bits << 32 | leaf is kept -- here by LDS atomics.
// The winner's barycentrics and hit point are recomputed by the ray's own lane at the end.
struct ClosestBatch {
	unsigned int entry[128];
	unsigned long long best_key[64];
	unsigned int hit_bits[2];  // rays with an accepted triangle, whatever its distance (reference :108-113)
	float prune_margin;        // KernelParams::prune_margin, kept HERE across the walk (a scalar register held across it would be spilled)
	unsigned int unpruned_bytes;  // KernelParams::unpruned_bytes, likewise
};
constexpr unsigned long long KEY_NONE = ~0ull;
constexpr uint32_t INF_BITS = 0x7F800000u;

// Any-hit shared walk of one packet (AO): a lane leaves at its first accepted
// triangle and bumps *occluded (reference :251 only uses the boolean).
// EXACT walks `nodes_ptr` (exact boxes); the fast form walks `walk_ptr` (padded boxes) and gates every candidate
// leaf with its own box, the head of its leaf record (scalar loads for a leaf tested on the spot; in a batch each lane
// loads its pair's record relative to the same scalar base: no buffer descriptor held across the walk).
template <bool EXACT, bool PREFETCH = false>
__device__ __forceinline__ void shared_walk_any_hit(const float4 *__restrict__ nodes_ptr, const float4 *__restrict__ walk_ptr,
                                                    const float4 *__restrict__ tris_ptr,
                                                    uint32_t count, const Ray &ray_in, const float (&frame)[12][64], uint32_t h,
                                                    float max_distance, float below, float walk_scale, bool alive, bool tame,
                                                    unsigned int *occluded, LeafBatch &batch, uint32_t batch_below,
                                                    unsigned long long *prof, uint32_t entry_begin = 0u, uint32_t entry_end = 0xFFFFFFFFu,
                                                    uint32_t copy = 0u) {  // (`copy`: KernelParams::walk_ce_bytes, the fast form only)
	(void) prof;  // (-DOCRT_STAMPS builds: time in the node loop / in batches, loop entries, batches, leaf stops)
	// the live lanes as a scalar mask: the node steps then need no per-lane bookkeeping at all
	unsigned long long alive_mask = wave_ballot(alive);
	// Registers held across the walk are scarce (64 per lane at 8 waves per SIMD, 7 of them the loop's own): the
	// ray's origin stays in the tile's LDS table (frame[0..2][h], where setup_ray took it from) and is read again
	// where a triangle or a leaf's own box is tested, and the lane number is recomputed where it is needed.
	auto with_origin = [&]() {
		Ray r = ray_in;
		r.ox = frame[0][h]; r.oy = frame[1][h]; r.oz = frame[2][h];
		return r;
	};
	if (!EXACT) {
		// The triangle tests are not run where the walk meets them -- a leaf is hit by 15 of the 64
		// rays on average -- but collected as (ray, leaf) pairs and run 64 at a time, each lane taking
		// ANY pair: it fetches that ray from its owner (cross-lane reads) and the triangle by a
		// gather.  An any-hit ray only needs "some accepted triangle", so neither the order of the
		// tests nor who computes them matters, and every test is the same arithmetic on the same
		// operands as before.  A ray found occluded leaves the walk after the batch instead of at
		// the leaf, which only lets it ride along a little longer.
		uint32_t waiting = 0u;  // pairs in batch.entry (wave-uniform)
		uint32_t leaf_stops = 0u;  // (not used by this pass)
		auto run_batch = [&](uint32_t n) {
#ifdef OCRT_STAMPS
			const unsigned long long tb0 = __builtin_amdgcn_s_memrealtime();
#endif
			wave_lds_sync();
			const uint32_t lane = fresh_lane();
			const uint32_t pair = batch.entry[lane < n ? lane : 0u];
			const int owner = (int) (pair >> 26);
			const Ray ray = with_origin();
			Ray theirs;
			theirs.ox = __shfl(ray.ox, owner); theirs.oy = __shfl(ray.oy, owner); theirs.oz = __shfl(ray.oz, owner);
			theirs.dx = __shfl(ray.dx, owner); theirs.dy = __shfl(ray.dy, owner); theirs.dz = __shfl(ray.dz, owner);
			theirs.ix = __shfl(ray.ix, owner); theirs.iy = __shfl(ray.iy, owner); theirs.iz = __shfl(ray.iz, owner);
			if (lane < n) {
				// the pair is a candidate of the padded walk: the leaf's own box decides whether the reference tests it
				const uint32_t pair_leaf = pair & 0x03FFFFFFu;
				const float4 *rec = (const float4 *) ((const char *) tris_ptr + pair_leaf * LEAF_BYTES);  // (scalar base + 32-bit lane offset)
				const float4 lo = rec[0], hi = rec[1];
				const bool gate = exact_leaf_gate(lo, hi, theirs, below);
#ifdef OCRT_STAMPS
				prof[5] += n;  // candidate pairs / pairs whose own box passes
				prof[6] += (unsigned long long) __popcll(wave_ballot(gate));
#endif
				if (gate) {
					if (tri_any_hit(rec[2], rec[3], rec[4], rec[5], hi.w, theirs))
						atomicOr(&batch.occluded_bits[owner >> 5], 1u << (owner & 31));
				}
			}
			wave_lds_sync();
			const uint32_t bits = batch.occluded_bits[lane >> 5];
			if (alive && ((bits >> (lane & 31u)) & 1u)) {
				atomicAdd(occluded, 1u);  // once per ray, however many of its pairs were accepted
				alive = false;
			}
			wave_lds_sync();
			if (lane < 2u)
				batch.occluded_bits[lane] = 0u;
			alive_mask = wave_ballot(alive);
#ifdef OCRT_STAMPS
			prof[1] += __builtin_amdgcn_s_memrealtime() - tb0;
			prof[3] += 1;
#endif
		};
		const uint32_t list_lds_address = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (uintptr_t) &batch.entry[0]);  // (low half of the flat address; scalar)
		const SignMasks sign{ 0ull, 0ull, 0ull };  // (not looked at by the any-hit loop)
		const uint32_t variant = WALK_MIXED;
		// byte offset of the node: the walk reads the centre / half-extent copy of the records, which lies behind the plane
		// form's and its two END records (scene_pack.cc, make_walk_array; KernelParams::walk_ce_bytes) -- and of that copy
		// only the tile's ENTRY subtree [entry_begin, entry_end): the deepest node under which every leaf lies that a ray
		// of this tile can reach (entry_kernel)
		uint32_t at = copy + entry_begin;
		const uint32_t whole = count * 32u;
		const uint32_t end = copy + (entry_end < whole ? entry_end : whole);
		const WalkRay walk_ray = tame ? make_walk_ray(with_origin(), walk_scale, true) : make_walk_ray(with_origin(), walk_scale);  // (wave-uniform)
		const uint32_t lane_tag = fresh_lane() << 26;
		while (alive_mask != 0ull && at < end) {
			uint32_t leaf = 0u;
			unsigned long long hit_mask = 0ull;
#ifdef OCRT_STAMPS
			const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
#endif
			const uint32_t status = walk_collect<true, PREFETCH>(variant, walk_ptr, at, walk_ray, sign, below, alive_mask, hit_mask, leaf,
			                                           waiting, leaf_stops, list_lds_address, lane_tag, batch_below, end);
#ifdef OCRT_STAMPS
			prof[0] += __builtin_amdgcn_s_memrealtime() - tw0;
			prof[2] += 1;
#endif
			if (status == 0u)
				break;
			if (status == 1u) {
				// enough of the packet is at this leaf: test it here, box and triangle out of SGPRs
				const float4 *rec = tris_ptr + LEAF_F4 * leaf;
				const float4 lo = rec[0], hi = rec[1], q0 = rec[2], q1 = rec[3], q2 = rec[4], q3 = rec[5];
				const Ray ray = with_origin();
				if (((hit_mask >> fresh_lane()) & 1ull) && exact_leaf_gate(lo, hi, ray, below)) {
					if (tri_any_hit(q0, q1, q2, q3, hi.w, ray)) {
						atomicAdd(occluded, 1u);
						alive = false;
					}
				}
				alive_mask = wave_ballot(alive);
			} else {
				run_batch(64u);
				waiting -= 64u;
				const uint32_t me = fresh_lane();
				if (me < waiting)  // the pairs beyond the batch move to the front
					batch.entry[me] = batch.entry[64u + me];
			}
			at += 32u;
		}
		if (waiting != 0u)
			run_batch(waiting);
#ifdef OCRT_STAMPS
		prof[4] += leaf_stops;
#endif
		return;
	}
	const Ray ray = ray_in;
	uint32_t mine = 0u;
	uint32_t at = 0u;
	while (at < count) {
		const u32x8 node = scalar_load_node(nodes_ptr, at);
		const float4 lo = make_float4(__uint_as_float(node[0]), __uint_as_float(node[1]), __uint_as_float(node[2]), 0.0f);
		const float4 hi = make_float4(__uint_as_float(node[4]), __uint_as_float(node[5]), __uint_as_float(node[6]), 0.0f);
		const uint32_t skip = node[3], leaf = node[7];
		const bool here = alive && mine == at;
		const bool box = here && slab_hit(lo, hi, ray, max_distance);
		mine = here ? (box ? at + 1u : at + skip) : mine;
		const unsigned long long hit_mask = wave_ballot(box);
		if (hit_mask != 0ull && leaf != NONE) {
			const float4 *tri = tris_ptr + LEAF_F4 * leaf + LEAF_TRI_F4;
			const float4 q0 = tri[0], q1 = tri[1], q2 = tri[2], q3 = tri[3];
			if (box) {
				const TriResult tr = tri_eval<false>(q0, q1, q2, q3, ray);
				if (tr.accepted) {
					atomicAdd(occluded, 1u);
					alive = false;
				}
			}
			if (wave_ballot(alive) == 0ull)
				break;
		}
		at = (uint32_t) __builtin_amdgcn_readfirstlane((int) (at + (hit_mask != 0ull ? 1u : skip)));
	}
}


}  // namespace ocrt
