// aabb.h -- axis-aligned box with exact min/max merging.
//
// Same surface as reference include/aabb.h:5-24 / src/aabb.cc: an inverted
// default box, merge by std::min/std::max (argument order kept, so even the
// sign of a zero bound matches), longest axis with x >= y >= z tie order, and
// the closed-interval inside() that the longest-axis split relies on.
#pragma once
#include <algorithm>
#include <limits>

#include "vec3.h"

struct AABB {
	Vec3f min;
	Vec3f max;

	AABB() : min(std::numeric_limits<float>::max()), max(-std::numeric_limits<float>::max()) {}
	AABB(const Vec3f &lo, const Vec3f &hi) : min(lo), max(hi) {}

	void merge(const AABB &b) {
		for (unsigned i = 0; i < 3; ++i) {
			min[i] = std::min(min[i], b.min[i]);
			max[i] = std::max(max[i], b.max[i]);
		}
	}
	void merge(const Vec3f &p) {
		for (unsigned i = 0; i < 3; ++i) {
			min[i] = std::min(min[i], p[i]);
			max[i] = std::max(max[i], p[i]);
		}
	}
	int getLongestAxis() const {
		const Vec3f d = max - min;
		if (d.x >= d.y && d.x >= d.z)
			return 0;
		if (d.y >= d.x && d.y >= d.z)
			return 1;
		return 2;
	}
	bool inside(const Vec3f &p) const {
		for (unsigned i = 0; i < 3; ++i)
			if (p[i] > max[i] || p[i] < min[i])
				return false;
		return true;
	}
};
