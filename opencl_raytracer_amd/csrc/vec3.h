// vec3.h -- 16-byte float vector used by every host<->device buffer.
//
// Layout contract (reference include/vec3.h:93-94): three floats plus one pad
// lane, so std::vector<Vec3f> can be uploaded as a float4 array.  Unlike the
// reference (whose non-default ctors leave the 4th lane indeterminate,
// include/vec3.h:12-22) the pad lane is ALWAYS zero here, because the kernel
// contract relies on w == 0.
#pragma once
#include <cmath>
#include <cstddef>

struct Vec3f {
	float x, y, z, w;

	constexpr Vec3f() : x(0.0f), y(0.0f), z(0.0f), w(0.0f) {}
	constexpr explicit Vec3f(float s) : x(s), y(s), z(s), w(0.0f) {}
	constexpr Vec3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_), w(0.0f) {}

	float operator[](unsigned i) const { return (&x)[i]; }
	float &operator[](unsigned i) { return (&x)[i]; }

	Vec3f operator+(const Vec3f &r) const { return Vec3f(x + r.x, y + r.y, z + r.z); }
	Vec3f operator-(const Vec3f &r) const { return Vec3f(x - r.x, y - r.y, z - r.z); }
	Vec3f operator*(float s) const { return Vec3f(x * s, y * s, z * s); }
	Vec3f operator/(float s) const { return Vec3f(x / s, y / s, z / s); }
	Vec3f &operator+=(const Vec3f &r) {
		x += r.x;
		y += r.y;
		z += r.z;
		return *this;
	}
	Vec3f &operator/=(float s) {
		x /= s;
		y /= s;
		z /= s;
		return *this;
	}
	// Evaluation order (x*x' + y*y') + z*z' is part of the arithmetic contract.
	float dot(const Vec3f &r) const { return x * r.x + y * r.y + z * r.z; }
	Vec3f cross(const Vec3f &r) const {
		return Vec3f(y * r.z - r.y * z, z * r.x - r.z * x, x * r.y - r.x * y);
	}
	float length() const { return std::sqrt(dot(*this)); }
};
static_assert(sizeof(Vec3f) == 16, "Vec3f must be float4-compatible");
