// types.h — host-side primitive types with the reference's public names and
// members, so code written against the reference's loaders keeps compiling:
//   Ray      (ray.h:10-21, type constants ray.h:5-8)
//   Triangle (triangle.h:9-26)
//   Light    (light.h:9-18)
//   Material (material.h:11-33)
//   Camera   (camera.h:5-24)
// Differences, all forced by the missing third-party libraries (glm, Eigen,
// OpenCV): vectors are trt::vec3, barycentrics are closed-form instead of an
// Eigen QR solve (triangle.cpp:12-29), and the texture is an owned RGB byte
// array instead of a cv::Mat.  Triangles additionally carry an integer
// material id: the device path never sees a std::string.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "vec.h"

namespace trt {

constexpr int DIFFUSE = 0;
constexpr int SPECULAR = 1;
constexpr int TRANSMISSION = 2;
constexpr int INVALID = 3;

class Ray {
public:
    Ray() {}
    Ray(vec3 s, vec3 d) : startpoint(s), direction(d) {}
    Ray(vec3 s, vec3 d, int r) : startpoint(s), direction(d), ray_type(r) {}
    vec3 startpoint;
    vec3 direction;
    int ray_type = INVALID;
};

class Triangle {
public:
    // Area by the law of cosines in double, as the reference computes it for the
    // light CDF (triangle.cpp:3-10).  (The reference spells it calAera.)
    double calAera() const;
    // Barycentric coordinates (b0,b1,b2) of a point in the triangle's plane.
    // The reference solves a 4x3 least-squares system (triangle.cpp:12-29); for
    // an in-plane point that solution is the area-ratio form used here.
    vec3 findBaryCor(vec3 hitp) const;

    vec3 v[3];
    vec3 vn[3];
    vec2 vt[3];
    vec3 normal;
    vec3 center;
    double area = 0.0;       // cumulative light area at insertion (scene.cpp:203)
    std::string mtl_name;
    int mtl_id = -1;
    bool is_emissive = false;
};

class Light {
public:
    Light() {}
    Light(std::string m, vec3 r) : mtl_name(std::move(m)), radiance(r) {}
    std::string mtl_name;
    vec3 radiance;
};

class Material {
public:
    // Loads map_Kd into `img` (8-bit RGB, row 0 = first row of the file).
    // Returns false (and leaves img empty) when the file cannot be decoded; the
    // reference only prints a message in that case (material.cpp:7-9).
    bool readinMap();

    vec3 Kd, Ks, Tr;
    float Ns = 1.f;
    float Ni = 1.f;
    std::string map_Kd;
    bool is_emissive = false;
    vec3 radiance;
    double area = 0.0;
    std::vector<Triangle> triangles;  // emissive triangles for light sampling
    std::vector<uint8_t> img;         // RGB, map_height x map_width x 3
    int map_height = 0, map_width = 0;
    std::string name;
};

class Camera {
public:
    void setCamera();             // camera.cpp:3-17
    Ray getRay(float s, float t) const;  // camera.cpp:19-28
    void Print() const;

    double fovy = 90;
    vec3 eye = vec3(278.0f, 273.0f, -800.0f);
    vec3 lookat = vec3(278.0f, 273.0f, -799.0f);
    vec3 up = vec3(0.0f, 1.0f, 0.0f);
    double aspect_ratio = 1.0;
    vec3 lower_left_corner, horizontal, vertical;
};

}  // namespace trt
