// rt_vec.h -- host-side float3 with the operation ORDER of GLM 0.9.8.5, which the reference computes with
// (glm/glm/detail/func_geometric.inl): dot = (x*x' + y*y') + z*z', cross as written there,
// normalize(v) = v * (1 / sqrt(dot(v,v))), length = sqrt(dot(v,v)).  Bit parity of the loader's derived
// data (triangle normals, light bases, camera basis) depends on keeping this order; build with
// -ffp-contract=off.
#pragma once
#include <cmath>

namespace rt {
struct vec3 {
	float x, y, z;
	vec3() : x(0), y(0), z(0) {}                     // glm zero-initialises (type_vec3.inl:34-40)
	vec3(float a, float b, float c) : x(a), y(b), z(c) {}
	float& operator[](int i) { return (&x)[i]; }
	const float& operator[](int i) const { return (&x)[i]; }
};
inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline float dot(vec3 a, vec3 b) { vec3 t = a * b; return t.x + t.y + t.z; }
inline vec3 cross(vec3 x, vec3 y) { return vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
}  // namespace rt
