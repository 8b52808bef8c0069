// kernels/finish.hip.h -- pass 3: AO factor + supersample box filter + quantisation; the occlusion total; a resize on its own
// (part of the one translation unit kernels.hip; see its head for the passes and the arithmetic contract)
#pragma once
#include "common.hip.h"
#include "primary.hip.h"

namespace ocrt {

// The frame's last kernel also closes the frame's books: the claim cursors of the ray passes go back to 0 and the frame
// is counted -- the fused frame kernel (kernels/frame.hip.h) compares its tiles' flags with that count, and a replayed
// graph cannot be handed a new number.  Everything that claimed from those cursors has ended: this kernel follows the ray
// passes in the stream.  One workgroup, eight lanes.
__device__ __forceinline__ void frame_is_over(FrameCounters *counters) {
	if (counters && blockIdx.x == 0u && blockIdx.y == 0u && threadIdx.x < XCD_GROUPS) {
		counters->queue[threadIdx.x].head = counters->queue[threadIdx.x].split_units;
		counters->queue[threadIdx.x].split_head = 0u;
		counters->queue[threadIdx.x].primary_head = 0u;
		if (threadIdx.x == 0u)
			counters->frame_seq += 1u;
	}
}

// Pass 3, the frame's last kernel: value *= 1 - hits / n (reference :256 and :305-307) for the sub-pixels that wait
// for it, and the supersample box filter + 8-bit quantisation (reference src/ray_tracer.cc:3-16) in the same sweep
// over the float image.  One thread per OUTPUT pixel of this rank's bands: it visits its n x n sub-pixels in the
// reference's order (ssY-major, ssX-minor), replaces every pending tag (primary_tile) by
// value * (1 - occluded / n_dirs) -- value and count from the tile's slot of the hit list --, WRITES THAT BACK (the float
// image is what `download` hands out, reference src/opencl_host.cc:150-153) and sums.  `out` may be null (a frame
// without the device resize).  (The frame's occlusion TOTAL is no business of the frame: until round 4 this kernel summed
// it -- per lane, wave, workgroup, then one atomic per workgroup on ONE address: 8 640 of them at 1080p, which took
// longer than the rest of the kernel, 48 us -> 12 us without, 0.29 -> 0.07 ms at 4K.  The counts stay in the hit list
// until the host's next frame, and whoever asks for the statistic has them summed then: occluded_sum_kernel.)
// Band layout as in resize_kernel below.
__global__ __launch_bounds__(256) void finish_kernel(float *__restrict__ image, const HitRec *__restrict__ hits,
                                                     const uint32_t *__restrict__ occluded_of,
                                                     const uint32_t *__restrict__ tile_base, unsigned char *__restrict__ out,
                                                     uint32_t width, uint32_t height, uint32_t total_width, uint32_t n,
                                                     uint32_t tiles_x, Partition part, uint32_t rows_per_band, uint32_t ao_divisor,
                                                     FrameCounters *__restrict__ counters) {
	frame_is_over(counters);
	const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t j = blockIdx.y;
	const uint32_t band_local = j / rows_per_band;
	const uint32_t y = (band_local * part.nranks + part.rank) * rows_per_band + (j - band_local * rows_per_band);
	if (x >= width)
		return;
	const float divisor = (float) ao_divisor;
	float total = 0.0f;
	for (uint32_t sy = 0; sy < n; ++sy) {
		const uint32_t row_index = j * n + sy;
		float *row = image + (size_t) row_index * total_width + (size_t) x * n;
		for (uint32_t sx = 0; sx < n; ++sx) {
			float v = row[sx];
			const uint32_t bits = __float_as_uint(v);
			if (is_pending(bits)) {
				const uint32_t column = x * n + sx;
				const size_t slot = (size_t) tile_base[(size_t) (row_index / TILE_H) * tiles_x + column / TILE_W] + (bits & 63u);
				v = hits[slot].value * (1.0f - ((float) occluded_of[slot] / divisor));
				row[sx] = v;
			}
			total += v;
		}
	}
	if (out)
		out[(size_t) j * width + x] = y < height ? (unsigned char) ((total / (float) (n * n)) * 255.0f) : (unsigned char) 0;
}

// The same pass for supersampled frames (n >= 2).  With one thread per output pixel a wave's 64 lanes read 64 places
// 4 n bytes apart -- and, through the tags, 64 tiles' parts of the hit list -- with every load: at `-s 64` (n = 8: a pixel is a
// whole tile) that is 64 cache lines per instruction and the pass runs at the L1's line rate, 3.85 ms for 3.8 GB
// (profiles/r04_notes.md, section 13).  Here a workgroup takes `pixels_per_block` neighbouring output pixels of one row:
// its threads sweep the n sub-pixel rows ALONG the rows (a wave reads 256 contiguous bytes of the image and the slots of
// eight neighbouring tiles), replace the tags as above, write back, and leave the values in LDS; then one thread per
// output pixel adds its n x n values up in the reference's order (ssY-major, ssX-minor: src/ray_tracer.cc:7-13) -- the
// same additions in the same order as finish_kernel's, so the same bits.  The cells of a pixel are n * n | 1 floats
// apart (odd: the adding threads do not meet in a bank).
// RESOLVE = false: the box filter alone, of an image that holds no tags any more (a resize on its own: launch_resize).
constexpr uint32_t FINISH_CELL_FLOATS = 4160u;  // 64 pixels of 8 x 8 sub-pixels and their padding
template <bool RESOLVE>
__global__ __launch_bounds__(256) void finish_wide_kernel(float *__restrict__ image, const HitRec *__restrict__ hits,
                                                          const uint32_t *__restrict__ occluded_of,
                                                          const uint32_t *__restrict__ tile_base, unsigned char *__restrict__ out,
                                                          uint32_t width, uint32_t height, uint32_t total_width, uint32_t n,
                                                          uint32_t tiles_x, Partition part, uint32_t rows_per_band, uint32_t ao_divisor,
                                                          uint32_t pixels_per_block, FrameCounters *__restrict__ counters) {
	__shared__ float cell[FINISH_CELL_FLOATS];
	if (RESOLVE)
		frame_is_over(counters);
	const uint32_t x0 = blockIdx.x * pixels_per_block;
	const uint32_t pixels = width - x0 < pixels_per_block ? width - x0 : pixels_per_block;
	const uint32_t columns = pixels * n;
	const uint32_t stride = (n * n) | 1u;
	const uint32_t j = blockIdx.y;
	const uint32_t band_local = j / rows_per_band;
	const uint32_t y = (band_local * part.nranks + part.rank) * rows_per_band + (j - band_local * rows_per_band);
	const float divisor = (float) ao_divisor;
	for (uint32_t sy = 0; sy < n; ++sy) {
		const uint32_t row_index = j * n + sy;
		float *row = image + (size_t) row_index * total_width + (size_t) x0 * n;
		const uint32_t *bases = tile_base + (size_t) (row_index / TILE_H) * tiles_x;
		for (uint32_t c = threadIdx.x; c < columns; c += 256u) {
			float v = row[c];
			const uint32_t bits = __float_as_uint(v);
			if (RESOLVE && is_pending(bits)) {
				const size_t slot = (size_t) bases[(x0 * n + c) / TILE_W] + (bits & 63u);
				v = hits[slot].value * (1.0f - ((float) occluded_of[slot] / divisor));
				row[c] = v;
			}
			const uint32_t p = c / n;
			cell[p * stride + sy * n + (c - p * n)] = v;
		}
	}
	__syncthreads();
	if (out && threadIdx.x < pixels) {
		const float *cells = cell + threadIdx.x * stride;
		float total = 0.0f;
		for (uint32_t i = 0; i < n * n; ++i)
			total += cells[i];
		out[(size_t) j * width + x0 + threadIdx.x] = y < height ? (unsigned char) ((total / (float) (n * n)) * 255.0f) : (unsigned char) 0;
	}
}

// The occlusion counts of a frame, summed: RenderStats::ao_occluded, on demand (DeviceRenderer::stats) -- the counts are
// in the hit list from the end of the ambient-occlusion pass until the host's next primary pass clears them slot by slot.
// Grid-stride, per lane / wave / workgroup, one atomic per workgroup (at most 256) onto a total the launcher has zeroed.
__global__ __launch_bounds__(256) void occluded_sum_kernel(const uint32_t *__restrict__ occluded_of, size_t slots,
                                                           FrameCounters *__restrict__ counters) {
	__shared__ unsigned long long block_total;
	if (threadIdx.x == 0)
		block_total = 0ull;
	__syncthreads();
	unsigned long long mine = 0ull;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t) gridDim.x * blockDim.x)
		mine += occluded_of[i];
	for (int offset = 32; offset > 0; offset >>= 1)
		mine += (unsigned long long) __shfl_down((long long) mine, offset);
	if ((threadIdx.x & 63u) == 0u && mine)
		atomicAdd(&block_total, mine);
	__syncthreads();
	if (threadIdx.x == 0 && block_total)
		atomicAdd(&counters->occluded, block_total);
}

// Supersample box filter + 8-bit quantisation on the device: one thread per
// output pixel, ssY-major / ssX-minor float summation and truncating store,
// exactly reference src/ray_tracer.cc:3-16.  Works on this rank's bands only: both the
// float image and the 8-bit buffer hold them back to back, so local output row j is
// the box filter of local sub-pixel rows j*n .. j*n+n-1; it is global row
// (band_local * nranks + rank) * rows_per_band + j % rows_per_band, and rows past the
// image's height are written as 0.
__global__ __launch_bounds__(256) void resize_kernel(const float *__restrict__ tmp, unsigned char *__restrict__ out,
                                                     uint32_t width, uint32_t height, uint32_t total_width, uint32_t n,
                                                     Partition part, uint32_t rows_per_band) {
	const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t j = blockIdx.y;
	const uint32_t band_local = j / rows_per_band;
	const uint32_t y = (band_local * part.nranks + part.rank) * rows_per_band + (j - band_local * rows_per_band);
	if (x >= width)
		return;
	unsigned char q = 0;
	if (y < height) {
		float total = 0.0f;
		for (uint32_t sy = 0; sy < n; ++sy) {
			const float *row = tmp + (size_t) (j * n + sy) * total_width + (size_t) x * n;
			for (uint32_t sx = 0; sx < n; ++sx)
				total += row[sx];
		}
		q = (unsigned char) ((total / (float) (n * n)) * 255.0f);
	}
	out[(size_t) j * width + x] = q;
}

}  // namespace ocrt
