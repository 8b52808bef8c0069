// walk_margin_check.cc -- property test of the outward margin of the padded walk boxes (CPU, no GPU needed).
//
// The fast form of the shared walk tests a box with t = fma(b', inv, -(o * inv)) on padded planes b'
// (opencl_raytracer_amd/csrc/scene_pack.cc, padded_bound; kernels.hip, walk_collect) where the reference computes
// fl(fl(b - o) * inv) (src/intersect_kernel.cl:21-61).  Claim: whenever the reference's test passes for a ray and a
// box, the fma test on the padded box passes too.  This program throws random and adversarial (ray, box) pairs at
// both tests with the kernel's exact arithmetic (std::fmaf = one rounding; maxNum / minNum like v_max3 / v_min3)
// and reports every pair that the reference accepts and the conservative test rejects.  Exit code 0 = none.
// Ambient-occlusion pairs also go through the SCALED form of the test (reciprocals times walk_scale_for(max_distance),
// z-axis values clamped to [0, 1]; kernels.hip, OCRT_TEST_COHERENT_SCALED).  (Its extra margin term is there by
// proof: the base margin's slack hides its absence from random pairs, grazing ones included.)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>

namespace ocrt {
float padded_bound(float b, float origin_bound, bool upper, float scaled_reach);  // libocrt_hip.so
float walk_scale_for(float max_distance);
bool walk_scale_usable(float scale, float origin_limit);
void padded_centre_extent(float padded_lo, float padded_hi, float origin_bound, float *centre, float *half_extent);
}

namespace {

const float INF = std::numeric_limits<float>::infinity();

// reference src/intersect_kernel.cl:21-61, straight
bool reference_slab(const float lo[3], const float hi[3], const float o[3], const float d[3], float max_distance) {
	float t_min, t_max, ty_min, ty_max, tz_min, tz_max;
	float div = 1.0f / d[0];
	if (div >= 0) { t_min = (lo[0] - o[0]) * div; t_max = (hi[0] - o[0]) * div; }
	else { t_min = (hi[0] - o[0]) * div; t_max = (lo[0] - o[0]) * div; }
	div = 1 / d[1];
	if (div >= 0) { ty_min = (lo[1] - o[1]) * div; ty_max = (hi[1] - o[1]) * div; }
	else { ty_min = (hi[1] - o[1]) * div; ty_max = (lo[1] - o[1]) * div; }
	if (t_min > ty_max || ty_min > t_max) return false;
	t_min = std::fmax(t_min, ty_min);
	t_max = std::fmin(t_max, ty_max);
	div = 1 / d[2];
	if (div >= 0) { tz_min = (lo[2] - o[2]) * div; tz_max = (hi[2] - o[2]) * div; }
	else { tz_min = (hi[2] - o[2]) * div; tz_max = (lo[2] - o[2]) * div; }
	if (t_min > tz_max || tz_min > t_max) return false;
	t_min = std::fmax(t_min, tz_min);
	t_max = std::fmin(t_max, tz_max);
	return t_min < max_distance && t_max > 0;
}

float walk_reciprocal(float i) { return std::fabs(i) == INF ? std::copysign(0x1.0p+100f, i) : i; }

// kernels.hip, OCRT_TEST_MIXED / OCRT_TEST_COHERENT on the padded box
bool conservative_slab(const float plo[3], const float phi[3], const float o[3], const float d[3], float below) {
	float near[3], far[3];
	for (int k = 0; k < 3; ++k) {
		const float inv = 1.0f / d[k];
		const float wi = walk_reciprocal(inv);
		const float oi = -(o[k] * wi);
		const float a = std::fmaf(plo[k], wi, oi), b = std::fmaf(phi[k], wi, oi);
		near[k] = inv >= 0 ? a : b;
		far[k] = inv >= 0 ? b : a;
	}
	const float tiny = std::numeric_limits<float>::denorm_min();
	const float n = std::fmax(std::fmax(near[0], near[1]), std::fmax(near[2], tiny));  // fmax / fmin drop NaN like v_max3 / v_min3
	const float f = std::fmin(std::fmin(far[0], far[1]), std::fmin(far[2], below));
	return n <= f;
}

// kernels.hip, OCRT_TEST_MIXED_SCALED / OCRT_TEST_COHERENT_SCALED: reciprocals times `scale`, the z-axis fmas clamped
// to [0, 1] (v_fma_f32 ... clamp: NaN -> 0 with DX10_CLAMP, which is what amdhsa kernels run with)
float clamp01(float v) { return v != v ? 0.0f : std::fmin(std::fmax(v, 0.0f), 1.0f); }
bool scaled_slab(const float plo[3], const float phi[3], const float o[3], const float d[3], float scale) {
	float near[3], far[3];
	for (int k = 0; k < 3; ++k) {
		const float inv = 1.0f / d[k];
		const float wi = walk_reciprocal(inv) * scale;
		const float oi = -(o[k] * wi);
		float a = std::fmaf(plo[k], wi, oi), b = std::fmaf(phi[k], wi, oi);
		if (k == 2) { a = clamp01(a); b = clamp01(b); }
		near[k] = inv >= 0 ? a : b;
		far[k] = inv >= 0 ? b : a;
	}
	const float n = std::fmax(std::fmax(near[0], near[1]), near[2]);
	const float f = std::fmin(std::fmin(far[0], far[1]), far[2]);
	return n < f;  // strictly: a box behind the origin on z (far clamped to 0) must fail
}

// kernels.hip, OCRT_TEST_CE_SCALED (mixed packets): centre / half-extent records, t_c = fma(c, inv, oi), the planes
// fma(-e, |inv|, t_c) and fma(e, |inv|, t_c), z clamped, compared strictly
bool ce_slab(const float c[3], const float e[3], const float o[3], const float d[3], float scale) {
	float near[3], far[3];
	for (int k = 0; k < 3; ++k) {
		const float inv = 1.0f / d[k];
		const float wi = walk_reciprocal(inv) * scale;
		const float oi = -(o[k] * wi);
		const float tc = std::fmaf(c[k], wi, oi);
		near[k] = std::fmaf(-e[k], std::fabs(wi), tc);
		far[k] = std::fmaf(e[k], std::fabs(wi), tc);
		if (k == 2) { near[k] = clamp01(near[k]); far[k] = clamp01(far[k]); }
	}
	const float n = std::fmax(std::fmax(near[0], near[1]), near[2]);
	const float f = std::fmin(std::fmin(far[0], far[1]), far[2]);
	return n < f;
}

}  // namespace

int main(int argc, char **argv) {
	const long cases = argc > 1 ? std::atol(argv[1]) : 20000000L;
	const bool unpadded = argc > 2 && std::atoi(argv[2]) != 0;  // self-check: without the margin misses MUST show up
	// The corner the scaled form must refuse: a scene of extent 1e6 (origins up to 2e6 + 4) with a max_distance so small
	// that origin * 2^100 * scale overflows -- every fma on a zero-direction axis would be +-inf and real boxes rejected.
	if (ocrt::walk_scale_usable(ocrt::walk_scale_for(0.007f), 2.0e6f + 4.0f) ||
	    ocrt::walk_scale_usable(ocrt::walk_scale_for(1.0e-6f), 2.0e6f + 4.0f) ||
	    !ocrt::walk_scale_usable(ocrt::walk_scale_for(0.2f), 24.0f) || !ocrt::walk_scale_usable(ocrt::walk_scale_for(0.5f), 4100.0f)) {
		std::printf("walk_scale_usable: wrong answer in the overflow corner\n");
		return 1;
	}
	for (float scale : { ocrt::walk_scale_for(0.007f), ocrt::walk_scale_for(3.0f), ocrt::walk_scale_for(1.0e-6f) })
		for (float limit : { 24.0f, 4100.0f, 2.0e6f + 4.0f })
			if (ocrt::walk_scale_usable(scale, limit)) {
				const float worst = limit * (0x1.0p+100f * scale), finite = limit * (1.0e30f * scale);
				if (!(worst < INF) || !(finite < INF)) {
					std::printf("walk_scale_usable accepted an overflowing pair: scale %a limit %a\n", scale, limit);
					return 1;
				}
			}
	std::mt19937_64 rng(20261004);
	std::uniform_real_distribution<float> unit(-1.0f, 1.0f);
	std::uniform_int_distribution<int> pick(0, 15);
	long accepted = 0, failures = 0, scaled_cases = 0, scaled_failures = 0, ce_failures = 0;
	for (long c = 0; c < cases; ++c) {
		const float extent = std::ldexp(1.0f, pick(rng) - 4);         // scene extents 1/16 .. 2048
		const float max_distance = (c & 1) ? 100000.0f : extent * std::ldexp(1.0f, -(pick(rng) % 6));  // primary / AO rays
		float lo[3], hi[3], o[3], d[3];
		for (int k = 0; k < 3; ++k) {
			const float centre = unit(rng) * extent, half = std::fabs(unit(rng)) * extent * std::ldexp(1.0f, -(pick(rng) % 12));
			lo[k] = centre - half;
			hi[k] = centre + half;
			if (pick(rng) == 0)
				hi[k] = lo[k];  // flat box
			d[k] = unit(rng);
			o[k] = centre + unit(rng) * half * 1.5f;  // in or near the slab
			switch (pick(rng)) {  // adversarial placements and directions
			case 0: o[k] = lo[k]; break;
			case 1: o[k] = hi[k]; break;
			case 2: o[k] = std::nextafter(lo[k], -INF); break;
			case 3: o[k] = std::nextafter(hi[k], INF); break;
			case 4: d[k] = 0.0f; break;
			case 5: d[k] = -0.0f; break;
			case 6: d[k] = std::ldexp(unit(rng), -20 - pick(rng)); break;  // nearly parallel to the slab
			case 7: o[k] = unit(rng) * extent; break;                       // anywhere in the scene
			default: break;
			}
		}
		if ((c & 1) == 1) { o[0] = 0.0f; o[1] = 0.0f; o[2] = 2.0f; }  // primary rays start at the camera
		if ((c & 6) == 2) {
			// grazing: the ray leaves through the far plane of one axis where it enters through the near plane of
			// another (a box edge), at a distance below max_distance -- near and far t agree to a few ulps
			const int i = pick(rng) % 3, j = (i + 1 + pick(rng) % 2) % 3;
			float t = std::fabs(unit(rng)) * max_distance;
			if ((c & 8) == 8) {
				// ... far out along the ray, which runs mostly along j and starts near the coordinate origin: the
				// planes' own magnitude buys the least slack against the scaled form's per-axis rounding there
				t = (0.5f + 0.5f * std::fabs(unit(rng))) * max_distance;
				for (int k = 0; k < 3; ++k)
					o[k] = unit(rng) * max_distance * std::ldexp(1.0f, -8);
				d[j] = std::copysign(0.9f + 0.1f * std::fabs(unit(rng)), d[j]);
				d[i] = std::copysign(0.05f + 0.3f * std::fabs(unit(rng)), d[i] == 0.0f ? 1.0f : d[i]);
			}
			if (d[i] != 0.0f && d[j] != 0.0f) {
				const float pi = o[i] + d[i] * t, pj = o[j] + d[j] * t;
				float span = std::fabs(unit(rng)) * extent * std::ldexp(1.0f, -(pick(rng) % 12));
				if (d[i] > 0) { lo[i] = pi; hi[i] = pi + span; } else { hi[i] = pi; lo[i] = pi - span; }   // near plane of i at t
				span = std::fabs(unit(rng)) * extent * std::ldexp(1.0f, -(pick(rng) % 12));
				if (d[j] > 0) { hi[j] = pj; lo[j] = pj - span; } else { lo[j] = pj; hi[j] = pj + span; }   // far plane of j at t
			}
		}
		// reciprocals must be infinite or below 1e30, origins within the scene (ray_is_selectable)
		bool selectable = false;
		for (int k = 0; k < 3; ++k) {
			const float inv = std::fabs(1.0f / d[k]);
			if (!(inv <= 1.0e30f || inv == INF)) { selectable = false; break; }
			if (inv <= 1.0e30f) selectable = true;
		}
		if (!selectable)
			continue;
		if (!reference_slab(lo, hi, o, d, max_distance))
			continue;
		++accepted;
		// the margin of scene_pack.cc, make_walk_array: O_k = max(|camera_k|, B_k + reach), reach = D * 1.001 (AO) or
		// the scene itself (no usable bound), capped by origin_limit = 2 extent + 4
		const double camera[3] = { 0.0, 0.0, 2.0 };
		const float scene = 2.0f * extent;  // |coordinates| <= 2 extent here
		const bool ao_bounded = (c & 1) == 0 && max_distance <= scene;
		const double reach = ao_bounded ? (double) max_distance * 1.001 : 2.0 * (double) scene;
		// the scaled form serves the ambient-occlusion rays where walk_scale_for() allows it
		const float scale = (c & 1) == 0 ? ocrt::walk_scale_for(max_distance) : 0.0f;
		const float scaled_reach = scale > 0.0f ? max_distance * 1.001f : 0.0f;
		float plo[3], phi[3], centre[3], half[3];
		for (int k = 0; k < 3; ++k) {
			const double box = std::fmax(std::fabs((double) lo[k]), std::fabs((double) hi[k]));
			const double origin = std::fmin(2.0 * scene + 4.0, std::fmax(camera[k], box + reach));
			plo[k] = unpadded ? lo[k] : ocrt::padded_bound(lo[k], (float) origin, false, scaled_reach);
			phi[k] = unpadded ? hi[k] : ocrt::padded_bound(hi[k], (float) origin, true, scaled_reach);
			if (unpadded) {
				centre[k] = 0.5f * lo[k] + 0.5f * hi[k];
				half[k] = std::fmax(centre[k] - lo[k], hi[k] - centre[k]);
			} else {
				ocrt::padded_centre_extent(plo[k], phi[k], (float) origin, &centre[k], &half[k]);
			}
		}
		if (scale > 0.0f) {
			bool unit_like = true;  // (ray_is_selectable: finite reciprocals are those of a unit vector's components)
			for (int k = 0; k < 3; ++k)
				unit_like = unit_like && std::fabs(1.0f / d[k]) >= 0.5f;
			if (unit_like) {
				++scaled_cases;
				if (!scaled_slab(plo, phi, o, d, scale) && ++scaled_failures <= 10 && !unpadded)
					std::printf("MISSED (scaled): o %a %a %a d %a %a %a lo %a %a %a hi %a %a %a md %a\n", o[0], o[1], o[2], d[0], d[1],
					            d[2], lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], max_distance);
				if (!ce_slab(centre, half, o, d, scale) && ++ce_failures <= 10 && !unpadded)
					std::printf("MISSED (centre / half-extent): o %a %a %a d %a %a %a lo %a %a %a hi %a %a %a md %a\n", o[0], o[1], o[2], d[0],
					            d[1], d[2], lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], max_distance);
			}
		}
		const float below = std::nextafter(max_distance, -INF);
		if (!conservative_slab(plo, phi, o, d, below)) {
			if (++failures <= 10 && !unpadded)
				std::printf("MISSED: o %a %a %a d %a %a %a lo %a %a %a hi %a %a %a md %a\n", o[0], o[1], o[2], d[0], d[1], d[2], lo[0],
				            lo[1], lo[2], hi[0], hi[1], hi[2], max_distance);
		}
	}
	std::printf("%ld pairs accepted by the reference's box test, %ld of them missed by the conservative test\n", accepted, failures);
	std::printf("%ld of those through the scaled test as well, %ld of them missed; %ld missed by its centre / half-extent form\n",
	            scaled_cases, scaled_failures, ce_failures);
	if (unpadded)
		return failures > 0 && scaled_failures > 0 && ce_failures > 0 ? 0 : 1;
	return failures == 0 && scaled_failures == 0 && ce_failures == 0 && accepted > cases / 100 && scaled_cases > cases / 400 ? 0 : 1;
}
