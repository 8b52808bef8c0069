// kernels/frame.hip.h -- the fused frame kernel: primary rays and ambient occlusion in ONE persistent launch
// (part of the one translation unit kernels.hip; see its head for the passes and the arithmetic contract)
#pragma once
#include "ao.hip.h"

namespace ocrt {

// Why: a frame rendered on its own -- the reference's blocking operator()(), src/opencl_host.cc:137-149 -- paid for the
// seam between its two ray passes: the primary pass is latency-bound (a model tile's wave is ~50 dependent scalar loads
// and ~30 leaf stops; the chip holds 3.6 waves per SIMD on average while it runs), the ambient-occlusion pass is bound by
// vector-instruction issue, and between them the device drained and filled again.  With frames in flight (the ring) the
// next frame's primary pass runs beside a falling AO pass; one frame alone had nothing to put there.
// Here the persistent workgroups of the ambient-occlusion pass do the primary pass themselves, a little AHEAD of the
// any-hit work all through the frame: a workgroup that has claimed ambient-occlusion tiles first takes 2 x 2 tile blocks of
// the group's primary work -- in the order the claims will want them -- while the cursor is below what its claim needs plus
// a margin (primary_top_up), then casts the claim's any-hit packets.  At any time a few of a CU's waves walk primary
// packets beside the any-hit packets of the others; the background's blocks, which nobody waits for, come last and fill
// the end of the frame, where the any-hit queues run dry.
// What replaces the launch boundary: a flag per tile (kernels/handoff.hip.h).  Deadlock cannot happen: a workgroup only
// waits for a flag after it has seen the cursor pass the blocks its claim needs, i.e. those blocks in the hands of running
// workgroups; and a wait is bounded anyway (tile_is_ready).
// The claim order of both kinds of work is per upload (DeviceRenderer::orderTiles); the finishing kernel, which follows
// in the stream, puts the cursors back and counts the frame (FrameCounters::frame_seq) for the next launch's flags.
// UNIFORM mode only (the RANDOM sampler re-reads hit records elsewhere; it keeps the two-kernel frame).
template <int MODE, bool PREFETCH>
__global__ __launch_bounds__(64 * AO_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void frame_kernel(FrameArgs A) {
	__shared__ TileShared shared_tiles[AO_WAVES];  // (a wave's slice also serves as its ClosestBatch during primary claims)
	__shared__ unsigned int wg_claim[8];
	__shared__ unsigned int wg_primary[2];
	static_assert(sizeof(ClosestBatch) <= sizeof(TileShared), "a wave's primary batch must fit its slice");
	ao_pass<MODE, true, PREFETCH, true>(A, shared_tiles, wg_claim, wg_primary);
}

}  // namespace ocrt
