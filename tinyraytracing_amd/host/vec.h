// vec.h — minimal fp32 vector types for the host side.
// The reference takes these from glm (ray.h:2, triangle.h:2, scene.h:7); glm is
// not available, and only a dozen operations are used on the path, so they are
// written out here with the evaluation order glm documents (dot = (x+y)+z,
// normalize = v * (1/sqrt(dot))), which the kernels and the oracle also use.
#pragma once
#include <cmath>

namespace trt {

struct vec2 {
    float x = 0.f, y = 0.f;
    vec2() {}
    vec2(float a, float b) : x(a), y(b) {}
};

struct vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    vec3() {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};

inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }

inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b)
{
    return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline vec3 vmin(vec3 a, vec3 b) { return vec3(b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z); }
inline vec3 vmax(vec3 a, vec3 b) { return vec3(a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z); }

}  // namespace trt
