// Microbenchmark: how many wave64 VALU instructions per cycle does one SIMD of an
// MI355X sustain on the instruction mix of the slab test (v_sub / v_mul / v_min /
// v_max, f32, no packed forms), as a function of waves per SIMD?  And with the
// scalar instructions of the walk loop in between?
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int SCALAR>
__global__ __launch_bounds__(256) void mix(float *sink, int iters, float seed) {
	float a = threadIdx.x, b = seed, c = 1.5f, d = 2.5f, e = 3.5f, f = 4.5f;
	unsigned s = blockIdx.x;
	for (int i = 0; i < iters; ++i) {
		// 24 vector instructions, chains of length 4 like the slab test
		asm volatile(
		    "v_sub_f32 %0, %6, %0\n\tv_sub_f32 %1, %6, %1\n\tv_sub_f32 %2, %6, %2\n\tv_sub_f32 %3, %6, %3\n\t"
		    "v_sub_f32 %4, %6, %4\n\tv_sub_f32 %5, %6, %5\n\t"
		    "v_mul_f32 %0, %7, %0\n\tv_mul_f32 %1, %7, %1\n\tv_mul_f32 %2, %7, %2\n\tv_mul_f32 %3, %7, %3\n\t"
		    "v_mul_f32 %4, %7, %4\n\tv_mul_f32 %5, %7, %5\n\t"
		    "v_min_f32 %0, %0, %1\n\tv_max_f32 %1, %1, %2\n\tv_min_f32 %2, %2, %3\n\tv_max_f32 %3, %3, %4\n\t"
		    "v_min_f32 %4, %4, %5\n\tv_max_f32 %5, %5, %0\n\t"
		    "v_max_f32 %0, 1, %0\n\tv_min_f32 %1, %6, %1\n\t"
		    "v_max3_f32 %2, %2, %3, %4\n\tv_min3_f32 %3, %3, %4, %5\n\t"
		    "v_cmp_le_f32 vcc, %2, %3\n\t"
		    "v_add_f32 %4, %4, %5\n\t"
		    : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f)
		    : "s"(seed), "v"(1.0001f)
		    : "vcc");
		if (SCALAR) {
			// the 9 scalar instructions of a node step
			asm volatile(
			    "s_and_b64 vcc, vcc, exec\n\ts_cmp_lg_u64 vcc, 0\n\ts_lshl_b32 %0, %0, 1\n\ts_add_u32 %0, %0, 3\n\t"
			    "s_cmp_lt_u32 %0, 0x7fffffff\n\ts_and_b32 %0, %0, 0xffff\n\ts_add_u32 %0, %0, 1\n\ts_lshr_b32 %0, %0, 1\n\t"
			    : "+s"(s)
			    :
			    : "vcc", "scc");
		}
	}
	sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + (float) s;
}

int main() {
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const double ghz = prop.clockRate * 1e-6;
	float *sink;
	hipMalloc(&sink, sizeof(float) * 256 * cus * 8);
	const int iters = 20000;
	for (int scalar = 0; scalar < 2; ++scalar)
		for (int blocks_per_cu = 1; blocks_per_cu <= 8; blocks_per_cu *= 2) {  // 256 threads = 1 wave per SIMD
			hipEvent_t t0, t1;
			hipEventCreate(&t0);
			hipEventCreate(&t1);
			for (int rep = 0; rep < 2; ++rep) {
				hipEventRecord(t0);
				if (scalar)
					hipLaunchKernelGGL(mix<1>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, sink, iters, 0.5f);
				else
					hipLaunchKernelGGL(mix<0>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, sink, iters, 0.5f);
				hipEventRecord(t1);
				hipEventSynchronize(t1);
			}
			float ms;
			hipEventElapsedTime(&ms, t0, t1);
			const double cycles = ms * 1e-3 * ghz * 1e9;
			const double valu_per_simd = 24.0 * iters * blocks_per_cu;
			printf("scalar=%d waves/SIMD=%d: %.3f ms, %.3f VALU instr/cycle/SIMD (%.2f cycles per instr), clock %.2f GHz\n", scalar,
			       blocks_per_cu, ms, valu_per_simd / cycles, cycles / valu_per_simd, ghz);
		}
	return 0;
}
