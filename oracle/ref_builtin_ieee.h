/*
 * ref_builtin_ieee.h -- DIAGNOSTIC ONLY (test infrastructure).
 *
 * Force-included (-include) into *diagnostic* gfx950 builds of the reference kernel to
 * replace ONE OpenCL geometric builtin at a time by the IEEE definition of SURVEY.md
 * 8a-0.3, so that tests/test_refkernel_gpu.py can attribute the differences between the
 * reference kernel linked against ROCm's real builtin library (the `default` and `strict`
 * builds, which never see this file) and the oracle to individual builtins.
 * Select with -DIEEE_DOT, -DIEEE_CROSS, -DIEEE_NORMALIZE, -DIEEE_LENGTH, and -DIEEE_TRIG
 * (sin / cos / cospi / sinpi of the UNIFORM ring angles evaluated in double and rounded
 * once, the definition of SURVEY.md 8a-0.5).
 */
#pragma OPENCL FP_CONTRACT OFF
#ifdef IEEE_DOT
inline float __attribute__((overloadable)) ieee_dot(float4 a, float4 b) {
	return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w;
}
#else
#define ieee_dot dot
#endif
#ifdef IEEE_LENGTH
inline float __attribute__((overloadable)) ieee_length(float4 a) { return sqrt(ieee_dot(a, a)); }
#else
#define ieee_length length
#endif
#ifdef IEEE_NORMALIZE
inline float4 __attribute__((overloadable)) ieee_normalize(float4 a) {
	const float l = sqrt(ieee_dot(a, a));
	return (float4) (a.x / l, a.y / l, a.z / l, a.w / l);
}
#endif
#ifdef IEEE_CROSS
inline float4 __attribute__((overloadable)) ieee_cross(float4 a, float4 b) {
	return (float4) (a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.0f);
}
#endif
#ifdef IEEE_DOT
#define dot ieee_dot
#endif
#ifdef IEEE_LENGTH
#define length ieee_length
#endif
#ifdef IEEE_NORMALIZE
#define normalize ieee_normalize
#endif
#ifdef IEEE_CROSS
#define cross ieee_cross
#endif
#ifdef IEEE_TRIG
#pragma OPENCL EXTENSION cl_khr_fp64 : enable
inline float __attribute__((overloadable)) ieee_sin(float x) { return (float) sin((double) x); }
inline float __attribute__((overloadable)) ieee_cos(float x) { return (float) cos((double) x); }
inline float __attribute__((overloadable)) ieee_cospi(float x) { return (float) cos(M_PI * (double) x); }
inline float __attribute__((overloadable)) ieee_sinpi(float x) { return (float) sin(M_PI * (double) x); }
/* the kernel also calls cos() on a double argument (ray_count, reference :241): leave that one to the library */
inline double __attribute__((overloadable)) ieee_cos(double x) { return cos(x); }
#define sin ieee_sin
#define cos ieee_cos
#define cospi ieee_cospi
#define sinpi ieee_sinpi
#endif
