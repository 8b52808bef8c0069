// Microbenchmark: what bounds a wave-uniform chain of dependent 64-byte loads (the shared BVH walk's node pairs) once the
// table is larger than the caches -- the scalar data cache's miss path, or the memory behind it?
// Every wave follows its own chain through a table of 64-byte records (word 0 = index of the next record, one random
// cycle over the whole table), 8 waves per SIMD on every CU, fetched
//   scalar:  s_load_dwordx16 (the walk's form: through the scalar data cache, shared by a few CUs),
//   vector:  one global_load_dword per lane, lane l reading word (l & 15) of the record (through the CU's vector L1),
//            the next index taken with v_readfirstlane,
// with 1 or 2 independent chains per wave.  Prints dependent steps per second over the chip, ns per step of one chain and
// the bytes per second that is in 64-byte lines.
// Build: hipcc -O3 --offload-arch=gfx950 -o chain_beyond_caches chain_beyond_caches.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                      \
	do {                                                                              \
		hipError_t e_ = (x);                                                          \
		if (e_ != hipSuccess) {                                                       \
			fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
			exit(1);                                                                  \
		}                                                                             \
	} while (0)

template <int CHAINS>
__global__ __launch_bounds__(256) void chain_scalar(const uint32_t *table, uint32_t entries, int steps, uint32_t *sink) {
	const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	uint32_t idx[CHAINS], acc = 0;
	for (int c = 0; c < CHAINS; ++c)
		idx[c] = (uint32_t) (((uint64_t) (wave * CHAINS + c) * 2654435761ull) % entries);
	for (int s = 0; s < steps; ++s) {
		u32x16 r[CHAINS];
		for (int c = 0; c < CHAINS; ++c) {
			const uint32_t *p = table + (size_t) idx[c] * 16u;
			asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(r[c]) : "s"(p) : "memory");
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		for (int c = 0; c < CHAINS; ++c) {
			// (tie the wait to the registers)
			asm volatile("" : "+s"(r[c]));
			idx[c] = r[c][0];
			acc += r[c][5] ^ r[c][15];
		}
	}
	if (acc == 0x12345678u)
		sink[wave] = acc;
}

// The walk's windows are 32-byte aligned: half of its 64-byte loads lie across two lines.  Records of 128 bytes here;
// BYTES (64 or 128) fetched from byte OFFSET (0 or 32) of the record: 1, 2, 2 or 3 lines per step.
template <int BYTES, int OFFSET>
__global__ __launch_bounds__(256) void chain_scalar_window(const uint32_t *table, uint32_t entries, int steps, uint32_t *sink) {
	const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	uint32_t idx = (uint32_t) (((uint64_t) wave * 2654435761ull) % entries), acc = 0;
	for (int s = 0; s < steps; ++s) {
		const uint32_t *p = table + (size_t) idx * 32u + OFFSET / 4;
		u32x16 a, b;
		asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(a) : "s"(p) : "memory");
		if (BYTES == 128)
			asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(b) : "s"(p) : "memory");
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		asm volatile("" : "+s"(a));
		if (BYTES == 128) {
			asm volatile("" : "+s"(b));
			acc += b[7];
		}
		idx = a[OFFSET ? 8 : 0];
		acc += a[5];
	}
	if (acc == 0x12345678u)
		sink[wave] = acc;
}

template <int CHAINS>
__global__ __launch_bounds__(256) void chain_vector(const uint32_t *table, uint32_t entries, int steps, uint32_t *sink) {
	const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const uint32_t word = threadIdx.x & 15u;
	uint32_t idx[CHAINS], acc = 0;
	for (int c = 0; c < CHAINS; ++c)
		idx[c] = (uint32_t) (((uint64_t) (wave * CHAINS + c) * 2654435761ull) % entries);
	for (int s = 0; s < steps; ++s) {
		uint32_t v[CHAINS];
		for (int c = 0; c < CHAINS; ++c)
			v[c] = table[(size_t) idx[c] * 16u + word];
		for (int c = 0; c < CHAINS; ++c) {
			idx[c] = __builtin_amdgcn_readfirstlane(v[c]);
			acc += v[c];
		}
	}
	if (acc == 0x12345678u)
		sink[wave] = acc;
}

int main(int argc, char **argv) {
	const double gib_max = argc > 1 ? atof(argv[1]) : 4.0;
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	uint32_t *sink;
	CHECK(hipMalloc(&sink, 1 << 20));
	const double sizes_mb[] = {4.5, 64.0, 580.0, 4096.0};
	for (double mb : sizes_mb) {
		if (mb / 1024.0 > gib_max)
			continue;
		const uint32_t entries = (uint32_t) (mb * 1024.0 * 1024.0 / 64.0);
		std::vector<uint32_t> next(entries);
		for (uint32_t i = 0; i < entries; ++i)
			next[i] = i;
		std::mt19937_64 rng(7);
		for (uint32_t i = entries - 1; i > 0; --i) {  // Sattolo: one cycle through every record
			const uint32_t j = (uint32_t) (rng() % i);
			std::swap(next[i], next[j]);
		}
		std::vector<uint32_t> host((size_t) entries * 16u);
		for (uint32_t i = 0; i < entries; ++i) {
			host[(size_t) i * 16u] = next[i];
			for (uint32_t w = 1; w < 16; ++w)
				host[(size_t) i * 16u + w] = i * 16u + w;
		}
		uint32_t *table;
		CHECK(hipMalloc(&table, host.size() * 4u));
		CHECK(hipMemcpy(table, host.data(), host.size() * 4u, hipMemcpyHostToDevice));
		for (int waves_per_simd : {8, 4, 2}) {
			const int blocks = cus * waves_per_simd;  // 4 waves per block, 4 SIMDs per CU
			const int steps = mb > 100.0 ? 1500 : 3000;
			for (int form = 0; form < 4; ++form) {
				hipEvent_t e0, e1;
				CHECK(hipEventCreate(&e0));
				CHECK(hipEventCreate(&e1));
				float best = 1e30f;
				for (int rep = 0; rep < 3; ++rep) {
					CHECK(hipEventRecord(e0));
					switch (form) {
					case 0: chain_scalar<1><<<blocks, 256>>>(table, entries, steps, sink); break;
					case 1: chain_scalar<2><<<blocks, 256>>>(table, entries, steps, sink); break;
					case 2: chain_vector<1><<<blocks, 256>>>(table, entries, steps, sink); break;
					default: chain_vector<2><<<blocks, 256>>>(table, entries, steps, sink); break;
					}
					CHECK(hipEventRecord(e1));
					CHECK(hipEventSynchronize(e1));
					float ms;
					CHECK(hipEventElapsedTime(&ms, e0, e1));
					best = ms < best ? ms : best;
				}
				const int chains = (form & 1) + 1;
				const double total = (double) blocks * 4.0 * chains * steps;
				printf("table %7.1f MB, %d waves/SIMD, %-6s x%d chains: %8.2f G steps/s over the chip, %7.1f ns per step of a chain, %7.1f GB/s in 64-byte lines\n",
				       mb, waves_per_simd, form < 2 ? "scalar" : "vector", chains, total / best * 1e-6, best * 1e6 / steps, total * 64.0 / best * 1e-6);
				fflush(stdout);
			}
		}
		CHECK(hipFree(table));
	}
	// windows across lines: 128-byte records, the next index in word 0 and in word 16 (= word 8 of a window from byte 32)
	for (double mb : {9.0, 128.0, 1160.0}) {
		const uint32_t entries = (uint32_t) (mb * 1024.0 * 1024.0 / 128.0);
		std::vector<uint32_t> next(entries);
		for (uint32_t i = 0; i < entries; ++i)
			next[i] = i;
		std::mt19937_64 rng(11);
		for (uint32_t i = entries - 1; i > 0; --i) {
			const uint32_t j = (uint32_t) (rng() % i);
			std::swap(next[i], next[j]);
		}
		std::vector<uint32_t> host((size_t) entries * 32u, 3u);
		for (uint32_t i = 0; i < entries; ++i)
			host[(size_t) i * 32u] = host[(size_t) i * 32u + 16u] = next[i];
		uint32_t *table;
		CHECK(hipMalloc(&table, host.size() * 4u + 256u));
		CHECK(hipMemcpy(table, host.data(), host.size() * 4u, hipMemcpyHostToDevice));
		const int blocks = cus * 8, steps = 1500;
		for (int form = 0; form < 4; ++form) {
			hipEvent_t e0, e1;
			CHECK(hipEventCreate(&e0));
			CHECK(hipEventCreate(&e1));
			float best = 1e30f;
			for (int rep = 0; rep < 3; ++rep) {
				CHECK(hipEventRecord(e0));
				switch (form) {
				case 0: chain_scalar_window<64, 0><<<blocks, 256>>>(table, entries, steps, sink); break;
				case 1: chain_scalar_window<64, 32><<<blocks, 256>>>(table, entries, steps, sink); break;
				case 2: chain_scalar_window<128, 0><<<blocks, 256>>>(table, entries, steps, sink); break;
				default: chain_scalar_window<128, 32><<<blocks, 256>>>(table, entries, steps, sink); break;
				}
				CHECK(hipEventRecord(e1));
				CHECK(hipEventSynchronize(e1));
				float ms;
				CHECK(hipEventElapsedTime(&ms, e0, e1));
				best = ms < best ? ms : best;
			}
			static const int lines[4] = {1, 2, 2, 3};
			const double total = (double) blocks * 4.0 * steps;
			printf("table %7.1f MB of 128-byte records, 8 waves/SIMD, scalar, %3d bytes from byte %2d (%d lines per step): %8.2f G steps/s, %7.1f ns per step, %6.2f G lines/s\n", mb,
			       form < 2 ? 64 : 128, (form & 1) * 32, lines[form], total / best * 1e-6, best * 1e6 / steps, total * lines[form] / best * 1e-6);
			fflush(stdout);
		}
		CHECK(hipFree(table));
	}
	return 0;
}
