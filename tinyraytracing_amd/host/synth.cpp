// synth.cpp — deterministic synthetic geometry for the large BASELINE.json
// configurations: "soup" = n random small triangles inside the
// Cornell box (config 3: deep-BVH stress), "blob" = a noise-displaced geodesic
// sphere with smooth normals (config 5: Stanford-style mesh).  Both are added to
// an already loaded base scene (scenes/back) whose walls and light close and
// light the scene.  Integer-hash based: no <random>, no files.
#include <cmath>
#include <stdexcept>

#include "scene.h"
#include "trt_prims.h"

namespace trt {
namespace {

struct Hash {
    uint32_t seed;
    // uniform in [0,1) from (index, lane)
    float u(uint64_t i, uint32_t lane) const
    {
        uint32_t x = trt_mix32((uint32_t)i ^ seed);
        x = trt_mix32(x + (uint32_t)(i >> 32) * 0x9E3779B9u + lane * 0x85EBCA6Bu);
        return (float)(x >> 8) * 5.9604644775390625e-8f;
    }
};

float lattice(uint32_t seed, int x, int y, int z)
{
    uint32_t h = trt_mix32(seed ^ (uint32_t)x * 0x8DA6B343u);
    h = trt_mix32(h ^ (uint32_t)y * 0xD8163841u);
    h = trt_mix32(h ^ (uint32_t)z * 0xCB1AB31Fu);
    return (float)(h >> 8) * 5.9604644775390625e-8f;
}

float valueNoise(uint32_t seed, vec3 p)
{
    const float fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
    const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    float tx = p.x - fx, ty = p.y - fy, tz = p.z - fz;
    tx = tx * tx * (3.f - 2.f * tx);
    ty = ty * ty * (3.f - 2.f * ty);
    tz = tz * tz * (3.f - 2.f * tz);
    float c[2][2][2];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int d = 0; d < 2; ++d) c[a][b][d] = lattice(seed, ix + a, iy + b, iz + d);
    auto lerp = [](float a, float b, float t) { return a + (b - a) * t; };
    const float x00 = lerp(c[0][0][0], c[1][0][0], tx), x10 = lerp(c[0][1][0], c[1][1][0], tx);
    const float x01 = lerp(c[0][0][1], c[1][0][1], tx), x11 = lerp(c[0][1][1], c[1][1][1], tx);
    return lerp(lerp(x00, x10, ty), lerp(x01, x11, ty), tz);
}

int requireMaterial(Scene& scene, const char* name)
{
    auto it = scene.material_ids.find(name);
    if (it == scene.material_ids.end()) throw std::runtime_error(std::string("synthetic scene: base scene lacks material ") + name);
    return it->second;
}

}  // namespace

void makeSoupScene(Scene& scene, uint32_t seed, uint64_t n_random, int width, int height)
{
    if (width > 0 && height > 0) scene.setResolution(width, height);
    const int mats[3] = {requireMaterial(scene, "back:DiffuseWhite"), requireMaterial(scene, "back:LeftWall"),
                         requireMaterial(scene, "back:RightWall")};
    const Hash h{seed};
    scene.triangles.reserve(scene.triangles.size() + n_random);
    for (uint64_t i = 0; i < n_random; ++i) {
        const vec3 c(h.u(i, 0) * 556.0f, h.u(i, 1) * 548.8f, h.u(i, 2) * 559.2f);
        const vec3 e1((h.u(i, 3) - 0.5f) * 6.0f, (h.u(i, 4) - 0.5f) * 6.0f, (h.u(i, 5) - 0.5f) * 6.0f);
        const vec3 e2((h.u(i, 6) - 0.5f) * 6.0f, (h.u(i, 7) - 0.5f) * 6.0f, (h.u(i, 8) - 0.5f) * 6.0f);
        Triangle t;
        t.v[0] = c - (e1 + e2) / 3.0f;
        t.v[1] = t.v[0] + e1;
        t.v[2] = t.v[0] + e2;
        const vec3 g = cross(t.v[1] - t.v[0], t.v[2] - t.v[0]);
        if (!(dot(g, g) > 1e-12f)) {  // degenerate draw: make it a tiny right triangle
            t.v[1] = t.v[0] + vec3(0.5f, 0.f, 0.f);
            t.v[2] = t.v[0] + vec3(0.f, 0.5f, 0.f);
        }
        t.normal = normalize(cross(t.v[1] - t.v[0], t.v[2] - t.v[0]));
        t.vn[0] = t.vn[1] = t.vn[2] = t.normal;
        t.center = (t.v[0] + t.v[1] + t.v[2]) / 3.0f;
        t.mtl_id = mats[i % 3];
        scene.triangles.push_back(std::move(t));
    }
}

void makeBlobScene(Scene& scene, uint32_t seed, uint64_t n_min, int width, int height)
{
    if (width > 0 && height > 0) scene.setResolution(width, height);
    const int mat = requireMaterial(scene, "back:DiffuseWhite");
    // geodesic frequency m: 20*m*m faces
    uint64_t m = 1;
    while (20ull * m * m < n_min) ++m;

    const float tphi = 1.61803398875f;
    const vec3 iv[12] = {normalize(vec3(-1, tphi, 0)), normalize(vec3(1, tphi, 0)), normalize(vec3(-1, -tphi, 0)), normalize(vec3(1, -tphi, 0)),
                         normalize(vec3(0, -1, tphi)), normalize(vec3(0, 1, tphi)), normalize(vec3(0, -1, -tphi)), normalize(vec3(0, 1, -tphi)),
                         normalize(vec3(tphi, 0, -1)), normalize(vec3(tphi, 0, 1)), normalize(vec3(-tphi, 0, -1)), normalize(vec3(-tphi, 0, 1))};
    static const int faces[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4}, {11, 10, 2}, {10, 7, 6}, {7, 1, 8},
                                     {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8}, {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
    const vec3 centre(278.0f, 200.0f, 280.0f);
    const float R = 200.0f, A = 25.0f;
    auto radius = [&](vec3 n) {
        const float nz = 0.5714f * valueNoise(seed, n * 2.0f + vec3(11.5f)) + 0.2857f * valueNoise(seed + 1, n * 4.0f + vec3(23.25f)) +
                         0.1429f * valueNoise(seed + 2, n * 8.0f + vec3(47.125f));
        return R + A * (2.0f * nz - 1.0f);
    };
    auto point = [&](vec3 n) { return centre + n * radius(n); };
    auto smoothNormal = [&](vec3 n) {
        // tangent frame + central differences of the displaced surface: a function
        // of the direction only, so shared vertices get identical normals
        vec3 t1 = std::fabs(n.x) > std::fabs(n.y) ? normalize(vec3(n.z, 0, -n.x)) : normalize(vec3(0, -n.z, n.y));
        vec3 t2 = cross(n, t1);
        const float eps = 2e-3f;
        const vec3 du = point(normalize(n + t1 * eps)) - point(normalize(n - t1 * eps));
        const vec3 dv = point(normalize(n + t2 * eps)) - point(normalize(n - t2 * eps));
        vec3 nn = normalize(cross(du, dv));
        if (dot(nn, n) < 0) nn = -nn;
        return nn;
    };
    auto corner = [&](const int* f, uint64_t i, uint64_t j) {
        const float a = (float)(m - i - j) / (float)m, b = (float)i / (float)m, c = (float)j / (float)m;
        return normalize(iv[f[0]] * a + iv[f[1]] * b + iv[f[2]] * c);
    };
    // face f, row i, column j -> triangle slot f*m*m + i*(2m-i) + 2j (+1 for the inverted one): every slot is
    // written exactly once, so the rows can be generated in parallel and the order never depends on the threads
    const size_t base = scene.triangles.size();
    scene.triangles.resize(base + 20ull * m * m);
    auto emit = [&](size_t slot, vec3 n0, vec3 n1, vec3 n2) {
        Triangle& t = scene.triangles[base + slot];
        t.v[0] = point(n0); t.v[1] = point(n1); t.v[2] = point(n2);
        t.vn[0] = smoothNormal(n0); t.vn[1] = smoothNormal(n1); t.vn[2] = smoothNormal(n2);
        t.normal = normalize(cross(t.v[1] - t.v[0], t.v[2] - t.v[0]));
        t.center = (t.v[0] + t.v[1] + t.v[2]) / 3.0f;
        t.mtl_id = mat;
    };
    const long long rows_total = 20ll * (long long)m;
#pragma omp parallel for schedule(dynamic, 16)
    for (long long fr = 0; fr < rows_total; ++fr) {
        const int f = (int)(fr / (long long)m);
        const uint64_t i = (uint64_t)(fr % (long long)m);
        const size_t row0 = (size_t)f * m * m + (size_t)(i * (2 * m - i));
        for (uint64_t j = 0; i + j < m; ++j) {
            emit(row0 + 2 * j, corner(faces[f], i, j), corner(faces[f], i + 1, j), corner(faces[f], i, j + 1));
            if (i + j + 1 < m) emit(row0 + 2 * j + 1, corner(faces[f], i + 1, j), corner(faces[f], i + 1, j + 1), corner(faces[f], i, j + 1));
        }
    }
}

}  // namespace trt
