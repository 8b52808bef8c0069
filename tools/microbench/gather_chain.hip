// Microbenchmark: dependent 16-byte gathers from an L2-resident table -- the
// memory pattern of a BVH walk (load node -> decide -> load next node).
// Prints ns and cycles per dependent step per wave for combinations of
//   loads per step (1 or 2 x 16 B), waves per SIMD, lane coherence (how many
//   consecutive lanes share an address), with buffer_load or LDS as the source.
// Build: hipcc -O3 --offload-arch=gfx950 -o gather_chain gather_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int LOADS>
__global__ __launch_bounds__(256) void chain(const uint4 *table, unsigned n_bytes, unsigned entries, int steps,
                                             int share, unsigned *sink) {
	__amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *) table, 0, (int) n_bytes, 0x00020000);
	const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned idx = ((gid / share) * 2654435761u) % entries;
	unsigned acc = 0;
	for (int s = 0; s < steps; ++s) {
		u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int) (idx * 32u), 0, 0);
		unsigned next = a.w;
		if (LOADS == 2) {
			u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int) (idx * 32u + 16u), 0, 0);
			next ^= b.x & 0u;
			acc += b.y;
		}
		acc += a.x;
		idx = next;
	}
	if (acc == 0x12345678u)
		sink[gid] = acc;
}

__global__ __launch_bounds__(256) void chain_lds(const uint4 *table, unsigned entries_lds, int steps, int share,
                                                 unsigned *sink) {
	extern __shared__ uint4 lds[];
	for (unsigned i = threadIdx.x; i < entries_lds; i += blockDim.x) {
		uint4 v = table[i * 2];
		v.w %= entries_lds;
		lds[i] = v;
	}
	__syncthreads();
	const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned idx = ((gid / share) * 2654435761u) % entries_lds;
	unsigned acc = 0;
	for (int s = 0; s < steps; ++s) {
		uint4 a = lds[idx];
		acc += a.x;
		idx = a.w;
	}
	if (acc == 0x12345678u)
		sink[gid] = acc;
}

int main() {
	const unsigned entries = 141139;  // bunny's node count, 32 B each = 4.5 MB
	std::vector<uint4> host(entries * 2);
	unsigned seed = 12345;
	for (unsigned i = 0; i < entries; ++i) {
		seed = seed * 1664525u + 1013904223u;
		host[2 * i] = make_uint4(i, 0, 0, (seed >> 8) % entries);
		host[2 * i + 1] = make_uint4(0, i, 0, 0);
	}
	uint4 *table;
	unsigned *sink;
	hipMalloc(&table, host.size() * sizeof(uint4));
	hipMalloc(&sink, 64u << 20);
	hipMemcpy(table, host.data(), host.size() * sizeof(uint4), hipMemcpyHostToDevice);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int steps = 2000;
	printf("source loads/step waves/SIMD share   ms     ns/step  cycles/step(2.4GHz)  Gsteps/s(lane)\n");
	for (int loads = 1; loads <= 2; ++loads)
		for (int wps : { 1, 2, 4, 8 })
			for (int share : { 64, 16, 4, 1 }) {
				const int blocks = 256 * wps;  // 256 CUs x wps blocks of 4 waves = wps waves per SIMD
				for (int rep = 0; rep < 2; ++rep) {
					hipEventRecord(e0);
					if (loads == 1)
						hipLaunchKernelGGL(chain<1>, dim3(blocks), dim3(256), 0, 0, table, entries * 32u, entries, steps, share, sink);
					else
						hipLaunchKernelGGL(chain<2>, dim3(blocks), dim3(256), 0, 0, table, entries * 32u, entries, steps, share, sink);
					hipEventRecord(e1);
					hipEventSynchronize(e1);
				}
				float ms;
				hipEventElapsedTime(&ms, e0, e1);
				const double ns = ms * 1e6 / steps;
				printf("L2     %d          %d          %2d     %7.3f %8.1f %8.0f            %8.2f\n", loads, wps, share, ms, ns,
				       ns * 2.4, (double) blocks * 256 * steps / (ms * 1e6));
			}
	for (int wps : { 1, 2, 4, 8 })
		for (int share : { 64, 16, 4, 1 }) {
			const int blocks = 256 * wps;
			const unsigned entries_lds = 1024;  // 16 KB per block
			for (int rep = 0; rep < 2; ++rep) {
				hipEventRecord(e0);
				hipLaunchKernelGGL(chain_lds, dim3(blocks), dim3(256), entries_lds * 16, 0, table, entries_lds, steps, share, sink);
				hipEventRecord(e1);
				hipEventSynchronize(e1);
			}
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			const double ns = ms * 1e6 / steps;
			printf("LDS    1          %d          %2d     %7.3f %8.1f %8.0f            %8.2f\n", wps, share, ms, ns, ns * 2.4,
			       (double) blocks * 256 * steps / (ms * 1e6));
		}
	return 0;
}
