// tri_predicate_check.hip -- GPU check of opencl_raytracer_amd/csrc/tri_predicate.h (test infrastructure): wherever
// tri_zone() says "accepted" or "rejected", the reference's own two divisions (tri_accepts_exact) say the same.
// 2^32 (X, Y, D) triples, most of them built so that s = X / D, t = Y / D or s + t lands within a few units in the
// last place -- up to 1e-4 -- of a threshold or of a zone edge; the rest wide-ranging, plus raw bit patterns
// (infinities, NaNs, subnormals) and triangles whose D makes inv_d a NaN.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -I opencl_raytracer_amd/csrc -o check tests/tri_predicate_check.hip
//   ./check  ->  one JSON line; exit status 1 on any disagreement
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "tri_predicate.h"

using namespace ocrt;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}
__device__ __forceinline__ float unit(uint32_t h) { return (float) (h >> 8) * (1.0f / 16777216.0f); }  // [0, 1)
__device__ __forceinline__ float nudge(float x, uint32_t h) {  // a few units in the last place either way
	const int steps = (int) (h % 9u) - 4;
	return __uint_as_float(__float_as_uint(x) + (uint32_t) steps);
}
// A value near one of the places where a decision changes.
__device__ __forceinline__ float near_edge(uint32_t h) {
	const float edges[8] = { TRI_LOW, 0.0f, TRI_OUT_LOW, 1.00001f, TRI_IN_HIGH, TRI_OUT_HIGH, 1.0f, 0.5f };
	const float edge = edges[h & 7u];
	const float span = exp2f(-13.0f - 20.0f * unit(mix(h ^ 0x9e3779b9u)));  // 1.2e-4 ... 1e-10
	return edge + ((h >> 3) & 1u ? span : -span) * unit(mix(h + 77u));
}

struct Tally {
	unsigned long long said_in_but_rejected, said_out_but_accepted, undecided, decided, nan_inverse_decided;
};

__global__ void sweep(Tally *tally, uint32_t per_thread) {
	unsigned long long bad_in = 0, bad_out = 0, undecided = 0, decided = 0, nan_decided = 0;
	const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
	for (uint32_t k = 0; k < per_thread; ++k) {
		const uint32_t h = mix(id * 2654435761u + k * 40503u + 1u);
		const uint32_t kind = h % 16u;
		float D, X, Y;
		// D: negative for a real triangle (uv^2 < uu vv); magnitudes over the whole range tri_inverse_d lets through
		const float magnitude = exp2f(-99.0f + 198.0f * unit(mix(h ^ 0x1234567u)));
		D = (mix(h + 5u) & 7u) ? -magnitude : magnitude;
		if (kind < 10u) {  // s and t near edges, or s near an edge and s + t near one
			const float s = near_edge(mix(h + 11u));
			const float t = (kind & 1u) ? near_edge(mix(h + 13u)) : near_edge(mix(h + 17u)) - s;
			X = nudge(s * D, mix(h + 19u));
			Y = nudge(t * D, mix(h + 23u));
		} else if (kind < 13u) {  // anywhere in [-4, 6)
			X = (10.0f * unit(mix(h + 29u)) - 4.0f) * D;
			Y = (10.0f * unit(mix(h + 31u)) - 4.0f) * D;
		} else if (kind == 13u) {  // magnitudes over the whole float range against this D
			X = exp2f(-140.0f + 268.0f * unit(mix(h + 37u))) * ((h >> 9) & 1u ? 1.0f : -1.0f);
			Y = exp2f(-140.0f + 268.0f * unit(mix(h + 41u))) * ((h >> 10) & 1u ? 1.0f : -1.0f);
		} else if (kind == 14u) {  // raw bit patterns: infinities, NaNs, subnormals, zeros
			X = __uint_as_float(mix(h + 43u));
			Y = __uint_as_float(mix(h + 47u));
			if ((h >> 11) & 1u)
				X = near_edge(mix(h + 53u)) * D;
		} else {  // a D that tri_inverse_d refuses
			const uint32_t pick = mix(h + 59u) % 5u;
			D = pick == 0u ? 0.0f : pick == 1u ? __uint_as_float(mix(h + 61u) & 0x007FFFFFu) : pick == 2u ? 3.0e33f
			    : pick == 3u ? __uint_as_float(0x7FC00000u) : -__uint_as_float(0x7F800000u);
			X = near_edge(mix(h + 67u));
			Y = near_edge(mix(h + 71u));
		}
		const float inv_d = tri_inverse_d(D);
		const unsigned int zone = tri_zone(X, Y, inv_d);
		const bool exact = tri_accepts_exact(X, Y, D);
		if (zone == 2u) {
			++undecided;
		} else {
			++decided;
			if (inv_d != inv_d)
				++nan_decided;
			if (zone == 1u && !exact)
				++bad_in;
			if (zone == 0u && exact)
				++bad_out;
		}
	}
	atomicAdd(&tally->said_in_but_rejected, bad_in);
	atomicAdd(&tally->said_out_but_accepted, bad_out);
	atomicAdd(&tally->undecided, undecided);
	atomicAdd(&tally->decided, decided);
	atomicAdd(&tally->nan_inverse_decided, nan_decided);
}

int main() {
	Tally *device = nullptr, host = {};
	if (hipMalloc(&device, sizeof(Tally)) != hipSuccess || hipMemset(device, 0, sizeof(Tally)) != hipSuccess) {
		fprintf(stderr, "tri_predicate_check: no device memory\n");
		return 2;
	}
	const uint32_t blocks = 16384, threads = 256, per_thread = 1024;  // 2^32 triples
	sweep<<<blocks, threads>>>(device, per_thread);
	if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&host, device, sizeof(Tally), hipMemcpyDeviceToHost) != hipSuccess) {
		fprintf(stderr, "tri_predicate_check: the sweep failed\n");
		return 2;
	}
	printf("{\"triples\": %llu, \"decided_without_division\": %llu, \"undecided\": %llu, \"said_in_but_rejected\": %llu, "
	       "\"said_out_but_accepted\": %llu, \"nan_inverse_decided\": %llu}\n",
	       host.decided + host.undecided, host.decided, host.undecided, host.said_in_but_rejected, host.said_out_but_accepted,
	       host.nan_inverse_decided);
	return host.said_in_but_rejected || host.said_out_but_accepted || host.nan_inverse_decided ? 1 : 0;
}
