// kernels/handoff.hip.h -- data handed from one workgroup to another INSIDE a launch (the fused frame kernel)
// (part of the one translation unit kernels.hip; see its head for the passes and the arithmetic contract)
#pragma once
#include "common.hip.h"

namespace ocrt {

// In the fused frame kernel (kernels/frame.hip.h) a tile's hit records are written by the workgroup that cast its
// primary rays and read, in the same launch, by whichever workgroup claims the tile's ambient-occlusion packets -- on
// another CU, possibly another XCD.  Per-XCD L2s are not coherent with each other and a CU's vector L1 is never
// refreshed by another CU's stores (MI355X_MICROARCH.md, "inter-workgroup visibility"), so the hand-over follows the
// form measured valid there without an agent-scope fence on either side:
//   producer  every byte handed over is stored device-coherently (`sc1`: written through, 4- and 16-byte stores), every
//             storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, then ONE lane
//             stores the flags -- `sc1` stores again: tile_ready[tile] = the frame's number;
//   consumer  a wave polls the flag with `sc1` loads (global_load, never flat_) and only after its poll has matched
//             loads the records, with `sc1` loads to registers.
// The same rule already carried the tile words into round 4's ordering step.  Staleness cannot be seen in a test that
// renders the same frame twice -- the records of frame f + 1 ARE those of frame f -- so tests poison the hit list
// between frames (rt_debug_poison_hit_list, tests/test_fused_frame.py).
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (s_nop 1: a vector-memory store of more than 64 bits reads its data registers after issue -- whoever writes them next
// must leave two wait states, and the compiler's hazard recogniser, which inserts them for its own instructions, does
// not look inside an asm block: without the nop the instruction scheduled right behind this one recycled the first two
// data registers and the records' x and y went out as garbage.)
__device__ __forceinline__ void store_f4_device_coherent(void *p, float4 v) {
	const f32x4 data = { v.x, v.y, v.z, v.w };
	asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(data) : "memory");
}
__device__ __forceinline__ void store_u32_device_coherent(uint32_t *p, uint32_t v) {
	asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
// two consecutive float4 (a hit record), both loads in flight before the wait
__device__ __forceinline__ void load_2f4_device_coherent(const float4 *p, float4 &a, float4 &b) {
	f32x4 x, y;
	asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
	             : "=&v"(x), "=&v"(y) : "v"(p) : "memory");
	a = make_float4(x.x, x.y, x.z, x.w);
	b = make_float4(y.x, y.y, y.z, y.w);
}
// one word through a scalar base and a 32-bit lane offset (wave-uniform use: every lane reads the same word)
__device__ __forceinline__ uint32_t load_u32_device_coherent(const uint32_t *base, uint32_t index) {
	uint32_t v;
	const uint32_t offset = index * 4u;
	asm volatile("global_load_dword %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(offset), "s"(base) : "memory");
	return (uint32_t) __builtin_amdgcn_readfirstlane((int) v);
}

// The number of the frame being rendered (FrameCounters::frame_seq + 1; the finishing kernel, which runs strictly
// after the frame kernel, counts it up -- a replayed graph cannot pass a new number in).  Never 0: a zeroed flag array
// says "no frame yet".
__device__ __forceinline__ uint32_t frame_number(const FrameCounters *counters) {
	uint32_t seq = 0u;
	asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(seq) : "s"(counters), "i"((int) offsetof(FrameCounters, frame_seq)) : "memory");
	return seq + 1u;
}

// Wave-uniform: waits until tile_ready[tile] holds this frame's number.  Whoever is waited for is RUNNING (the caller
// has seen every block of the tile's group claimed, primary_claims), so the wait is bounded by that workgroup's four
// tiles -- and bounded in any case: after ~50 ms a wave gives up, says so in the frame's counters (the host reports a
// device error) and stops waiting for good; a frame must never hang the GPU, whatever went wrong.
constexpr uint32_t READY_SPINS = 1u << 16;
__device__ __forceinline__ bool tile_is_ready(const uint32_t *tile_ready, uint32_t tile, FrameCounters *counters, bool gave_up) {
	const uint32_t want = frame_number(counters);
	for (uint32_t spin = 0u; spin < (gave_up ? 1u : READY_SPINS); ++spin) {
		if (load_u32_device_coherent(tile_ready, tile) == want)
			return true;
		__builtin_amdgcn_s_sleep(16);
	}
	if (!gave_up && fresh_lane() == 0u)
		atomicAdd(&counters->stalled, 1u);
	return false;
}

}  // namespace ocrt
