// Probe (gfx950): sustained issue rate of single vector instructions, 8 waves per SIMD (8 workgroups of 4 waves per CU),
// 32 instructions per loop turn on 8 destination registers.  Event-timed; the shader clock is read with s_memtime
// against the 100 MHz wall clock.  hipcc --offload-arch=gfx950 -O2 -w -o valu_rate_probe tools/microbench/valu_rate_probe.hip   (output of one run: profiles/r02_valu_rate_probe.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2v __attribute__((ext_vector_type(2)));

#define R4(S) S S S S
#define KERNEL(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                                       \
	__global__ __launch_bounds__(256) void NAME(unsigned long long *out, int iters, float a, float b) {                    \
		f2v p0 = {(float)threadIdx.x, 1.f}, p1 = {2.f, 3.f}, p2 = {4.f, 5.f}, p3 = {6.f, 7.f}, va = {a, a}, vb = {b, b};    \
		asm volatile("s_mov_b32 s20, %0\n s_mov_b32 s21, %1\n s_mov_b32 s22, %0\n s_mov_b32 s23, %1\n s_mov_b64 s[24:25], -1\n v_mov_b32 v48, %0\n v_mov_b32 v49, %1\n v_mov_b32 v50, %0\n v_mov_b32 v51, %1\n v_mov_b32 v52, %0\n v_mov_b32 v53, %1\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0" \
		             : : "s"(a), "s"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53");                                       \
		const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();                                   \
		for (int i = 0; i < iters; ++i)                                                                                    \
			asm volatile(R4(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n")                                \
			             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(va), "v"(vb) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53"); \
		const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();                                   \
		if (p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y == 12345.678f) out[0] = 1;                                \
		if ((threadIdx.x & 63) == 0) { atomicAdd(&out[1], t1 - t0); atomicAdd(&out[2], w1 - w0); }                          \
	}
// %0..%3 = register pairs (lo: %0 as v[n:n+1]); single registers are addressed through the pair's halves below
#define L(P) "%" #P   // the pair
// LLVM inline asm cannot name half of a pair, so single-register kinds use 8 scratch registers v40-v47 instead
#define S8(OP, SRC) OP " v40, " SRC, OP " v41, " SRC, OP " v42, " SRC, OP " v43, " SRC, OP " v44, " SRC, OP " v45, " SRC, OP " v46, " SRC, OP " v47, " SRC
#define KERNEL1(NAME, OP, SRC) KERNEL(NAME, OP " v40, " SRC, OP " v41, " SRC, OP " v42, " SRC, OP " v43, " SRC, OP " v44, " SRC, OP " v45, " SRC, OP " v46, " SRC, OP " v47, " SRC)

KERNEL1(fma_vvv, "v_fma_f32", "v48, v49, v50")
KERNEL(fma_self, "v_fma_f32 v40, v40, v48, v49", "v_fma_f32 v41, v41, v48, v49", "v_fma_f32 v42, v42, v48, v49", "v_fma_f32 v43, v43, v48, v49",
       "v_fma_f32 v44, v44, v48, v49", "v_fma_f32 v45, v45, v48, v49", "v_fma_f32 v46, v46, v48, v49", "v_fma_f32 v47, v47, v48, v49")
KERNEL1(fma_svv, "v_fma_f32", "s20, v49, v50")
KERNEL1(fma_vsv, "v_fma_f32", "v48, s20, v50")
KERNEL1(fma_vvs, "v_fma_f32", "v48, v49, s20")
KERNEL1(mul_vv, "v_mul_f32", "v48, v49")
KERNEL1(mul_sv, "v_mul_f32", "s20, v49")
KERNEL1(add_vv, "v_add_f32", "v48, v49")
KERNEL1(sub_sv, "v_sub_f32", "s20, v49")
KERNEL1(max_vv, "v_max_f32", "v48, v49")
KERNEL1(max_sv, "v_max_f32", "s20, v49")
KERNEL1(max_cv, "v_max_f32", "1, v49")
KERNEL1(max3_vvv, "v_max3_f32", "v48, v49, v50")
KERNEL1(min3_svv, "v_min3_f32", "s20, v49, v50")
KERNEL1(med3_vvv, "v_med3_f32", "v48, v49, v50")
KERNEL1(mov_v, "v_mov_b32", "v48")
KERNEL1(mov_s, "v_mov_b32", "s20")
KERNEL1(cndmask_vcc, "v_cndmask_b32", "v48, v49, vcc")
KERNEL1(cndmask_s, "v_cndmask_b32", "v48, v49, s[24:25]")
KERNEL1(and_vv, "v_and_b32", "v48, v49")
KERNEL1(add_u32, "v_add_u32", "v48, v49")
KERNEL1(lshl_add, "v_lshl_add_u32", "v48, 2, v50")
KERNEL(cmp_vcc, "v_cmp_le_f32 vcc, v48, v49", "v_cmp_le_f32 vcc, v48, v49", "v_cmp_le_f32 vcc, v48, v49", "v_cmp_le_f32 vcc, v48, v49",
       "v_cmp_le_f32 vcc, v48, v49", "v_cmp_le_f32 vcc, v48, v49", "v_cmp_le_f32 vcc, v48, v49", "v_cmp_le_f32 vcc, v48, v49")
KERNEL(cmp_sgpr, "v_cmp_le_f32 s[24:25], v48, v49", "v_cmp_le_f32 s[24:25], v48, v49", "v_cmp_le_f32 s[24:25], v48, v49", "v_cmp_le_f32 s[24:25], v48, v49",
       "v_cmp_le_f32 s[24:25], v48, v49", "v_cmp_le_f32 s[24:25], v48, v49", "v_cmp_le_f32 s[24:25], v48, v49", "v_cmp_le_f32 s[24:25], v48, v49")
#define PK8(OP, SRC) KERNEL(OP##_k, #OP " v[40:41], " SRC, #OP " v[42:43], " SRC, #OP " v[44:45], " SRC, #OP " v[46:47], " SRC, #OP " v[40:41], " SRC, #OP " v[42:43], " SRC, #OP " v[44:45], " SRC, #OP " v[46:47], " SRC)
KERNEL(pk_fma_vvv, "v_pk_fma_f32 v[40:41], v[48:49], v[50:51], v[52:53]", "v_pk_fma_f32 v[42:43], v[48:49], v[50:51], v[52:53]",
       "v_pk_fma_f32 v[44:45], v[48:49], v[50:51], v[52:53]", "v_pk_fma_f32 v[46:47], v[48:49], v[50:51], v[52:53]",
       "v_pk_fma_f32 v[40:41], v[48:49], v[50:51], v[52:53]", "v_pk_fma_f32 v[42:43], v[48:49], v[50:51], v[52:53]",
       "v_pk_fma_f32 v[44:45], v[48:49], v[50:51], v[52:53]", "v_pk_fma_f32 v[46:47], v[48:49], v[50:51], v[52:53]")
KERNEL(pk_fma_svv, "v_pk_fma_f32 v[40:41], s[20:21], v[50:51], v[52:53]", "v_pk_fma_f32 v[42:43], s[22:23], v[50:51], v[52:53]",
       "v_pk_fma_f32 v[44:45], s[20:21], v[50:51], v[52:53]", "v_pk_fma_f32 v[46:47], s[22:23], v[50:51], v[52:53]",
       "v_pk_fma_f32 v[40:41], s[20:21], v[50:51], v[52:53]", "v_pk_fma_f32 v[42:43], s[22:23], v[50:51], v[52:53]",
       "v_pk_fma_f32 v[44:45], s[20:21], v[50:51], v[52:53]", "v_pk_fma_f32 v[46:47], s[22:23], v[50:51], v[52:53]")
KERNEL(pk_fma_svv_sel, "v_pk_fma_f32 v[40:41], s[20:21], v[50:51], v[52:53] op_sel:[1,0,0] op_sel_hi:[0,1,1]", "v_pk_fma_f32 v[42:43], s[22:23], v[50:51], v[52:53] op_sel:[1,0,0] op_sel_hi:[0,1,1]",
       "v_pk_fma_f32 v[44:45], s[20:21], v[50:51], v[52:53] op_sel:[1,0,0] op_sel_hi:[0,1,1]", "v_pk_fma_f32 v[46:47], s[22:23], v[50:51], v[52:53] op_sel:[1,0,0] op_sel_hi:[0,1,1]",
       "v_pk_fma_f32 v[40:41], s[20:21], v[50:51], v[52:53] op_sel:[1,0,0] op_sel_hi:[0,1,1]", "v_pk_fma_f32 v[42:43], s[22:23], v[50:51], v[52:53] op_sel:[1,0,0] op_sel_hi:[0,1,1]",
       "v_pk_fma_f32 v[44:45], s[20:21], v[50:51], v[52:53] op_sel:[1,0,0] op_sel_hi:[0,1,1]", "v_pk_fma_f32 v[46:47], s[22:23], v[50:51], v[52:53] op_sel:[1,0,0] op_sel_hi:[0,1,1]")
KERNEL(pk_mul_vv, "v_pk_mul_f32 v[40:41], v[48:49], v[50:51]", "v_pk_mul_f32 v[42:43], v[48:49], v[50:51]", "v_pk_mul_f32 v[44:45], v[48:49], v[50:51]", "v_pk_mul_f32 v[46:47], v[48:49], v[50:51]",
       "v_pk_mul_f32 v[40:41], v[48:49], v[50:51]", "v_pk_mul_f32 v[42:43], v[48:49], v[50:51]", "v_pk_mul_f32 v[44:45], v[48:49], v[50:51]", "v_pk_mul_f32 v[46:47], v[48:49], v[50:51]")
// the node test itself, today's form and the packed form (s20..s23 stand for the planes)
KERNEL(node_test_11, "v_fma_f32 v40, s20, v48, v49\n v_fma_f32 v41, s21, v48, v49\n v_fma_f32 v42, s22, v48, v49", "v_fma_f32 v43, s23, v48, v49\n v_fma_f32 v44, s20, v48, v49\n v_fma_f32 v45, s21, v48, v49",
       "v_max_f32 v42, 1, v42", "v_min_f32 v45, s22, v45", "v_max3_f32 v40, v40, v41, v42", "v_min3_f32 v43, v43, v44, v45", "v_cmp_le_f32 vcc, v40, v43", "s_and_b64 s[24:25], vcc, exec")
KERNEL(node_test_8, "v_pk_fma_f32 v[40:41], s[20:21], v[50:51], v[52:53]", "v_pk_fma_f32 v[42:43], s[22:23], v[50:51], v[52:53]", "v_pk_fma_f32 v[44:45], s[20:21], v[50:51], v[52:53] op_sel:[1,0,0] op_sel_hi:[0,1,1]",
       "v_max_f32 v42, 1, v42", "v_min_f32 v45, s22, v45\n v_max3_f32 v40, v40, v41, v42", "v_min3_f32 v43, v43, v44, v45", "v_cmp_le_f32 vcc, v40, v43", "s_and_b64 s[24:25], vcc, exec")

// ... and with the two clamps folded into the z-axis fmas (t scaled so that the far limit is 1.0)
KERNEL(node_test_9, "v_fma_f32 v40, s20, v48, v49\n v_fma_f32 v41, s21, v48, v49\n v_fma_f32 v42, s22, v48, v49 clamp", "v_fma_f32 v43, s23, v48, v49\n v_fma_f32 v44, s20, v48, v49\n v_fma_f32 v45, s21, v48, v49 clamp",
       "v_max3_f32 v40, v40, v41, v42", "v_min3_f32 v43, v43, v44, v45", "v_cmp_le_f32 vcc, v40, v43", "s_and_b64 s[24:25], vcc, exec", "", "")
// ... and round 4's centre / half-extent test, the one loop of every any-hit packet: 9 v_fma + max3 + min3 + cmp
KERNEL(node_test_12, "v_fma_f32 v40, s20, v48, v49\n v_fma_f32 v41, s21, v48, v49\n v_fma_f32 v42, s22, v48, v49",
       "v_fma_f32 v43, -s23, |v48|, v40\n v_fma_f32 v40, s23, |v48|, v40", "v_fma_f32 v44, -s20, |v48|, v41\n v_fma_f32 v41, s20, |v48|, v41",
       "v_fma_f32 v45, -s21, |v48|, v42 clamp\n v_fma_f32 v42, s21, |v48|, v42 clamp", "v_max3_f32 v43, v43, v44, v45", "v_min3_f32 v40, v40, v41, v42",
       "v_cmp_lt_f32 vcc, v43, v40", "s_and_b64 s[24:25], vcc, exec")
KERNEL1(fma_svv_clamp, "v_fma_f32", "s20, v49, v50 clamp")
KERNEL(fma_svv_4regs, "v_fma_f32 v40, s20, v48, v49", "v_fma_f32 v41, s21, v48, v49", "v_fma_f32 v42, s22, v48, v49", "v_fma_f32 v43, s23, v48, v49",
       "v_fma_f32 v44, s20, v48, v49", "v_fma_f32 v45, s21, v48, v49", "v_fma_f32 v46, s22, v48, v49", "v_fma_f32 v47, s23, v48, v49")
struct Kind { const char *name; void (*k)(unsigned long long *, int, float, float); double vector_per_turn; };
int main() {
	setvbuf(stdout, nullptr, _IONBF, 0);
	unsigned long long *d, h[3]; hipMalloc(&d, 64);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define K(N) {#N, N, 32.0}
	const Kind kinds[] = {K(fma_vvv), K(fma_self), K(fma_svv), K(fma_vsv), K(fma_vvs), K(mul_vv), K(mul_sv), K(add_vv), K(sub_sv), K(max_vv), K(max_sv), K(max_cv),
	                      K(max3_vvv), K(min3_svv), K(med3_vvv), K(mov_v), K(mov_s), K(cndmask_vcc), K(cndmask_s), K(and_vv), K(add_u32), K(lshl_add), K(cmp_vcc), K(cmp_sgpr),
	                      K(pk_fma_vvv), K(pk_fma_svv), K(pk_fma_svv_sel), K(pk_mul_vv), {"node_test_11 (4 tests per turn)", node_test_11, 44.0}, {"node_test_8 (4 tests per turn)", node_test_8, 32.0},
	                      {"node_test_9 (4 tests per turn)", node_test_9, 36.0}, {"node_test_12 (4 tests per turn)", node_test_12, 48.0}, K(fma_svv_clamp),
	                      K(fma_svv_4regs)};
	const int iters = 10000;
	for (const Kind &kind : kinds) {
		double best = 1e30, ghz = 0;
		for (int rep = 0; rep < 2; ++rep) {
			hipMemset(d, 0, 24);
			hipEventRecord(e0);
			kind.k<<<256 * 8, 256>>>(d, iters, 1.0001f, 0.5f);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
			if (ms < best) { best = ms; ghz = (double)h[1] / (h[2] * 10.0); }
		}
		const double per_simd = kind.vector_per_turn * iters * 8;
		printf("%-32s %.3f ms  %.3f instr/ns/SIMD  clock %.2f GHz  -> %.2f cycles per vector instruction per SIMD\n", kind.name, best, per_simd / (best * 1e6), ghz,
		       best * 1e6 * ghz / per_simd);
	}
	return 0;
}
