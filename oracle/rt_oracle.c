/*
 * rt_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see rt_oracle.h).
 *
 * Plain-C, three-lane restatement of reference src/intersect_kernel.cl under
 * the arithmetic contract of SURVEY.md 8a-0:
 *   - IEEE binary32 + - * / sqrt in the reference's source order, no FMA
 *     contraction (build with -ffp-contract=off, never -ffast-math);
 *   - OpenCL builtins fixed as: dot = (x*x'+y*y')+z*z' (w lanes are all 0),
 *     length = sqrtf(dot(v,v)), normalize = v / length(v) lane by lane,
 *     max/min = IEEE maxNum/minNum, clamp = min(max(x,lo),hi);
 *   - unsuffixed literals are double (0.0, 1.00001): the float operand is
 *     promoted for the comparison.
 *
 * Every function names the reference lines it follows.
 */
#include "rt_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct v3 {
	float x, y, z;
} v3;

static inline v3 v3_make(float x, float y, float z) {
	v3 r = { x, y, z };
	return r;
}
static inline v3 v3_load4(const float *base, uint32_t idx) {
	const float *p = base + 4u * (size_t) idx;
	return v3_make(p[0], p[1], p[2]);
}
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b) {
	return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }
static inline v3 v3_normalize(v3 a) {
	const float l = v3_length(a);
	return v3_make(a.x / l, a.y / l, a.z / l);
}
/* IEEE-754 maxNum / minNum (what v_max_f32 / fmaxf compute; sign of zero is
 * never observable in the callers below). */
static inline float f_max(float a, float b) { return (a > b || b != b) ? a : b; }
static inline float f_min(float a, float b) { return (a < b || b != b) ? a : b; }

typedef struct hit_record {
	uint32_t face_id; /* 3 * leaf index, as in the reference */
	v3 barycentric;
	v3 position;
	float distance;
} hit_record;

typedef struct ray_counters {
	uint64_t node_visits;
	uint64_t tri_tests;
} ray_counters;

/* reference: src/intersect_kernel.cl:21-61 (aabb_intersect). */
static inline int box_hit(const float *bb, v3 o, v3 d, float max_distance) {
	float t_min, t_max, ty_min, ty_max, tz_min, tz_max;
	float div = 1.0f / d.x;
	if (div >= 0) {
		t_min = (bb[0] - o.x) * div;
		t_max = (bb[4] - o.x) * div;
	} else {
		t_min = (bb[4] - o.x) * div;
		t_max = (bb[0] - o.x) * div;
	}
	div = 1 / d.y;
	if (div >= 0) {
		ty_min = (bb[1] - o.y) * div;
		ty_max = (bb[5] - o.y) * div;
	} else {
		ty_min = (bb[5] - o.y) * div;
		ty_max = (bb[1] - o.y) * div;
	}
	if (t_min > ty_max || ty_min > t_max)
		return 0;
	t_min = f_max(t_min, ty_min);
	t_max = f_min(t_max, ty_max);
	div = 1 / d.z;
	if (div >= 0) {
		tz_min = (bb[2] - o.z) * div;
		tz_max = (bb[6] - o.z) * div;
	} else {
		tz_min = (bb[6] - o.z) * div;
		tz_max = (bb[2] - o.z) * div;
	}
	if (t_min > tz_max || tz_min > t_max)
		return 0;
	t_min = f_max(t_min, tz_min);
	t_max = f_min(t_max, tz_max);
	return t_min < max_distance && t_max > 0;
}

/* reference: src/intersect_kernel.cl:65-114 (triangle_intersect).  rec may be
 * NULL for any-hit rays (the reference passes an uninitialised record there
 * and only uses the boolean, :249-253). */
static inline int tri_hit(v3 ta, v3 tb, v3 tc, uint32_t face_id, v3 o, v3 d, hit_record *rec) {
	const v3 u = v3_sub(tb, ta);
	const v3 v = v3_sub(tc, ta);
	const v3 n = v3_cross(u, v);
	const v3 w0 = v3_sub(o, ta);
	const float a = -v3_dot(n, w0);
	const float b = v3_dot(n, d);
	if (fabsf(b) < 0.000001f)
		return 0;
	const float r = a / b;
	if ((double) r < 0.0)
		return 0;
	const v3 ip = v3_add(o, v3_scale(d, r));
	const float uu = v3_dot(u, u);
	const float uv = v3_dot(u, v);
	const float vv = v3_dot(v, v);
	const v3 w = v3_sub(ip, ta);
	const float wu = v3_dot(u, w);
	const float wv = v3_dot(w, v);
	const float D = uv * uv - uu * vv;
	const float s = (uv * wv - vv * wu) / D;
	if (s < -0.00001f || (double) s > 1.00001)
		return 0;
	const float t = (uv * wu - uu * wv) / D;
	if (t < -0.00001f || (double) (s + t) > 1.00001)
		return 0;
	if (rec) {
		const float distance = v3_length(v3_sub(ip, o));
		if (rec->distance > distance) {
			rec->face_id = face_id;
			rec->barycentric = v3_make(1.0f - s - t, s, t);
			rec->position = ip;
			rec->distance = distance;
		}
	}
	return 1;
}

/* reference: src/intersect_kernel.cl:184-213 (scene_intersect): stackless
 * pre-order walk, nodes[i] = subtree size, one triangle per leaf. */
static int scene_hit(const orc_scene *s, v3 o, v3 d, hit_record *rec, float max_distance,
                     ray_counters *rc) {
	int is_intersecting = 0;
	uint32_t triangle_index = 0;
	const uint32_t count = s->nodes[0];
	for (uint32_t i = 0; i < count;) {
		const uint32_t node_count = s->nodes[i];
		rc->node_visits++;
		if (!box_hit(s->aabbs + 8u * (size_t) i, o, d, max_distance)) {
			triangle_index += (node_count + 1) >> 1;
			i += node_count;
		} else {
			if (node_count == 1) {
				const uint32_t face_id = triangle_index * 3;
				rc->tri_tests++;
				is_intersecting |= tri_hit(v3_load4(s->vertices, s->faces[face_id + 0]),
				                           v3_load4(s->vertices, s->faces[face_id + 1]),
				                           v3_load4(s->vertices, s->faces[face_id + 2]),
				                           face_id, o, d, rec);
				++triangle_index;
			}
			++i;
		}
	}
	return is_intersecting;
}

/* Tangent frame from the shading normal: reference src/intersect_kernel.cl:224-236
 * (and :155-167 for the RANDOM sampler, which normalises the normal first). */
static inline void hemisphere_basis(v3 basis_y, v3 *basis_x, v3 *basis_z) {
	v3 h = basis_y;
	if (fabsf(h.x) <= fabsf(h.y) && fabsf(h.x) <= fabsf(h.z))
		h.x = 1.0f;
	else if (fabsf(h.y) <= fabsf(h.x) && fabsf(h.y) <= fabsf(h.z))
		h.y = 1.0f;
	else if (fabsf(h.z) <= fabsf(h.x) && fabsf(h.z) <= fabsf(h.y))
		h.z = 1.0f;
	*basis_x = v3_normalize(v3_cross(h, basis_y));
	*basis_z = v3_normalize(v3_cross(*basis_x, basis_y));
}

/* OpenCL cospi/sinpi as fixed by the contract: evaluate in double, round once. */
static inline float cl_cospi(float x) { return (float) cos(M_PI * (double) x); }
static inline float cl_sinpi(float x) { return (float) sin(M_PI * (double) x); }

/* reference: src/intersect_kernel.cl:219-246.  M_PI / M_PI_2 are the double
 * macros of OpenCL C (the #ifndef at :1-4 is inactive), 2.0f * M_PI is double. */
uint32_t orc_ao_table(const orc_params *p, float *xyz, uint32_t cap) {
	uint32_t n = 0;
	const uint32_t circle_count = p->ao_num_samples;
	const float degrees = (float) (M_PI / 180);
	const float alpha_min = (float) p->ao_alpha_min * degrees;
	const float alpha_max = (float) p->ao_alpha_max * degrees;
	(void) alpha_min;
	for (uint32_t c = 0; c < circle_count; ++c) {
		const float step = alpha_max / circle_count;
		const float angle = (step * c) + alpha_min;
		const uint32_t ray_count = (uint32_t) ((2.0f * M_PI * cosf(angle)) / step);
		const float theta = (float) (M_PI_2 - angle);
		for (uint32_t k = 0; k <= ray_count; ++k) {
			const float phi = (float) ((2.0f * M_PI * k) / ray_count);
			const float xs = sinf(theta) * cl_cospi(phi);
			const float ys = cosf(theta);
			const float zs = sinf(theta) * cl_sinpi(phi);
			if (xyz && n < cap) {
				xyz[3 * n + 0] = xs;
				xyz[3 * n + 1] = ys;
				xyz[3 * n + 2] = zs;
			}
			++n;
		}
	}
	return n;
}

/* xorshift128: reference src/intersect_kernel.cl:128-152. */
typedef struct rng128 {
	uint32_t x, y, z, w;
} rng128;
static inline uint32_t rng_next(rng128 *v) {
	uint32_t t = v->x ^ (v->x << 11u);
	v->x = v->y;
	v->y = v->z;
	v->z = v->w;
	return v->w = v->w ^ (v->w >> 19u) ^ (t ^ (t >> 8u));
}
static inline void rng_seed(rng128 *v, uint32_t seed) {
	v->x = (123456789u ^ seed) * 88675123u;
	v->y = (362436069u ^ seed) * 123456789u;
	v->z = (521288629u ^ seed) * 362436069u;
	v->w = (88675123u ^ seed) * 521288629u;
	rng_next(v);
}
static inline float rng_float(rng128 *v) { return 2.32830643653869629E-10f * rng_next(v); }

/* reference: src/intersect_kernel.cl:214-277 (ambient_occlusion). */
static float ambient_occlusion(const orc_params *p, const orc_scene *s, const float *table,
                               uint32_t table_n, v3 point, v3 normal, uint32_t index,
                               ray_counters *rc, uint64_t *rays, uint64_t *occluded) {
	const v3 origin = v3_add(point, v3_scale(normal, 1.0f / 100000.0f));
	uint32_t hits = 0;
	const float max_distance = p->ao_max_distance;
	if (p->ao_method == 0) {
		v3 basis_x, basis_z;
		const v3 basis_y = normal;
		hemisphere_basis(basis_y, &basis_x, &basis_z);
		for (uint32_t k = 0; k < table_n; ++k) {
			const float xs = table[3 * k + 0], ys = table[3 * k + 1], zs = table[3 * k + 2];
			const v3 dir = v3_add(v3_add(v3_scale(basis_x, xs), v3_scale(basis_y, ys)),
			                      v3_scale(basis_z, zs));
			if (scene_hit(s, origin, dir, NULL, max_distance, rc))
				++hits;
		}
		*rays += table_n;
		*occluded += hits;
		return 1.0f - ((float) hits / (float) table_n);
	}
	/* RANDOM (:257-276): libm-dependent, outside the bit-exact contract. */
	v3 basis_x, basis_z;
	const v3 basis_y = v3_normalize(normal);
	hemisphere_basis(basis_y, &basis_x, &basis_z);
	rng128 rng;
	rng_seed(&rng, 536870923u * index);
	uint32_t n = p->ao_num_samples;
	++n;
	if (scene_hit(s, origin, normal, NULL, max_distance, rc))
		++hits;
	for (uint32_t i = 0; i < n; ++i) {
		const float xi1 = rng_float(&rng);
		const float xi2 = rng_float(&rng);
		const float theta = acosf(sqrtf(1.0f - xi1));
		const float phi = (float) (2.0 * xi2);
		const float xs = sinf(theta) * cl_cospi(phi);
		const float ys = cosf(theta);
		const float zs = sinf(theta) * cl_sinpi(phi);
		const v3 dir = v3_normalize(v3_add(
		    v3_add(v3_scale(basis_x, xs), v3_scale(basis_y, ys)), v3_scale(basis_z, zs)));
		if (scene_hit(s, origin, dir, NULL, max_distance, rc))
			++hits;
	}
	*rays += n + 1;
	*occluded += hits;
	return 1.0f - ((float) hits / (float) n);
}

/* reference: src/intersect_kernel.cl:278-310 (__kernel intersect), one sub-pixel. */
static float shade_subpixel(const orc_params *p, const orc_scene *s, const float *table,
                            uint32_t table_n, uint32_t x, uint32_t y, orc_counters *c) {
	const uint32_t W = p->width, H = p->height;
	const uint32_t index = y * W + x;
	const v3 camera = v3_make(0.0f, 0.0f, 2.0f);
	const float a = p->focal_length * (float) (int32_t) (W > H ? W : H);
	const v3 ray_dir = v3_normalize(v3_make(
	    ((float) x + 0.5f) / a - (float) (int32_t) W / (2.0f * a),
	    -(((float) y + 0.5f) / a - (float) (int32_t) H / (2.0f * a)), -1.0f));
	hit_record rec;
	memset(&rec, 0, sizeof rec);
	rec.distance = INFINITY;
	ray_counters rc = { 0, 0 };
	const int hit = scene_hit(s, camera, ray_dir, &rec, 100000.0f, &rc);
	c->primary_rays++;
	c->primary_node_visits += rc.node_visits;
	c->primary_tri_tests += rc.tri_tests;
	if (!hit)
		return 0.0f;
	c->primary_hits++;
	/* get_smooth_normal, :118-127 */
	const uint32_t v0 = s->faces[rec.face_id + 0];
	const uint32_t v1 = s->faces[rec.face_id + 1];
	const uint32_t v2 = s->faces[rec.face_id + 2];
	const v3 normal = v3_normalize(v3_add(
	    v3_add(v3_scale(v3_load4(s->normals, v0), rec.barycentric.x),
	           v3_scale(v3_load4(s->normals, v1), rec.barycentric.y)),
	    v3_scale(v3_load4(s->normals, v2), rec.barycentric.z)));
	float value = 1.0f;
	if (p->shading_enable) /* shade, :115-117 */
		value = f_min(f_max(-v3_dot(normal, ray_dir), 0.f), 1.f);
	if (p->ao_enable && p->ao_num_samples > 0) {
		ray_counters arc = { 0, 0 };
		value *= ambient_occlusion(p, s, table, table_n, rec.position, normal, index, &arc,
		                           &c->ao_rays, &c->ao_occluded);
		c->ao_node_visits += arc.node_visits;
		c->ao_tri_tests += arc.tri_tests;
	}
	return value;
}

#define ORC_MAX_AO_DIRS 65536u

int orc_render_rows(const orc_params *p, const orc_scene *s, float *image, uint32_t y0,
                    uint32_t y1, orc_counters *counters, int nthreads) {
	static float table_storage[3 * ORC_MAX_AO_DIRS];
	float *table = table_storage;
	uint32_t table_n = 0;
	if (p->ao_enable && p->ao_num_samples > 0 && p->ao_method == 0) {
		table_n = orc_ao_table(p, table, ORC_MAX_AO_DIRS);
		if (table_n > ORC_MAX_AO_DIRS)
			return -1;
	}
	orc_counters total;
	memset(&total, 0, sizeof total);
	int used = 1;
#ifdef _OPENMP
	if (nthreads <= 0)
		nthreads = omp_get_max_threads();
	used = nthreads;
#pragma omp parallel num_threads(nthreads)
#endif
	{
		orc_counters local;
		memset(&local, 0, sizeof local);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
		for (int64_t y = (int64_t) y0; y < (int64_t) y1; ++y) {
			for (uint32_t x = 0; x < p->width; ++x)
				image[(size_t) y * p->width + x] =
				    shade_subpixel(p, s, table, table_n, x, (uint32_t) y, &local);
		}
#ifdef _OPENMP
#pragma omp critical
#endif
		{
			total.primary_rays += local.primary_rays;
			total.primary_hits += local.primary_hits;
			total.primary_node_visits += local.primary_node_visits;
			total.primary_tri_tests += local.primary_tri_tests;
			total.ao_rays += local.ao_rays;
			total.ao_occluded += local.ao_occluded;
			total.ao_node_visits += local.ao_node_visits;
			total.ao_tri_tests += local.ao_tri_tests;
		}
	}
	if (counters)
		*counters = total;
	return used;
}

int orc_render(const orc_params *p, const orc_scene *s, float *image, orc_counters *counters,
               int nthreads) {
	return orc_render_rows(p, s, image, 0, p->height, counters, nthreads);
}

uint32_t orc_ss_factor(uint32_t n_super_samples) {
	return (uint32_t) sqrt((double) n_super_samples);
}

/* reference: src/ray_tracer.cc:3-16. `total / (n * n)` divides by an unsigned
 * converted to float; `* 255` is a float multiply; the store truncates. */
void orc_resize(const float *tmp, uint8_t *image, uint32_t width, uint32_t height,
                uint32_t n_super_samples) {
	const uint32_t n = orc_ss_factor(n_super_samples);
	const uint32_t total_width = width * n;
	for (uint32_t y = 0; y < height; ++y) {
		for (uint32_t x = 0; x < width; ++x) {
			float total = 0;
			for (uint32_t ssy = 0; ssy < n; ++ssy)
				for (uint32_t ssx = 0; ssx < n; ++ssx)
					total += tmp[(size_t) (y * n + ssy) * total_width + (x * n + ssx)];
			image[(size_t) y * width + x] = (uint8_t) ((total / (float) (n * n)) * 255);
		}
	}
}
