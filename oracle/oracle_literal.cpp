/*
 * oracle_literal.cpp — the reference's OWN arithmetic, statement by statement (ORACLE_MODE_LITERAL).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  oracle.cpp restates the path in the formulation the HIP kernels
 * use (Moller-Trumbore, fp32 scalars, the polynomial sin/cos/pow of trt_prims.h, iterative beta form) so that
 * GPU and oracle agree bit for bit.  THIS file restates the same path the way /root/reference/RayTracingOnCPU
 * computes it, so that the distance between the two formulations can be measured and frozen as the stated
 * tolerance (tests/test_literal_tolerance.py):
 *
 *   bvh.cpp:177-209   interactTriangle: stored unit normal (scene.cpp:196), |N.d| < 1e-5 cut, plane distance
 *                     t = ((p1 - S).N) / (d.N), t < 0.0005 cut, P = S + d t, three edge cross products whose
 *                     signs are taken in double
 *   triangle.cpp:12-29 findBaryCor: least-squares solve of the 4x3 system [v0 v1 v2; 1 1 1] b = [P; 1] in
 *                     double by column-pivoted Householder QR (what Eigen's colPivHouseholderQr().solve does),
 *                     narrowed to float
 *   bvh.cpp:211-229   interactBVHNode (pn from those barycentrics), :146-175 traverseBVH, :231-245 interactAABB
 *   pathTracing.cpp:3-102 shade() as the RECURSION it is, with the double scalars where the reference has them
 *                     (:20-25 texel address and Kd, :39,:45-46 light draws, :62 pdf, :68-69 cos_alpha and
 *                     pow(double), :116-125 Sample's angles through libm sin/cos/asin/acos/pow, :157-174 Fresnel,
 *                     :191-194 lobe selection)
 *   triangle.cpp:3-10 calAera in double, accumulated as scene.cpp:201-203 does, for the light CDFs and 1/A
 *   main.cpp:88-108   pixel -> (s, t), jitter, color / SAMPLE, double accumulation
 *
 * What is NOT the reference's: the random numbers.  The reference draws from shared static engines with no
 * per-pixel stream; here every draw is the counter stream of trt_prims.h, widened to double
 * where the reference's distribution is double, consumed in the fast oracle's order (jitter x, y; per vertex:
 * per light {CDF; r1, r2, r3}; RR; [Fresnel]; [lobe]; [phi, theta]) — so both formulations follow the SAME
 * path until a rounding difference flips a branch.  Two reference statements have no effect on the image and
 * are skipped: shadow rays whose sample is back-facing (traced at pathTracing.cpp:54, discarded at :60) and
 * INVALID extension rays (traced at :81, discarded at :82).
 * libm (glibc here, the MSVC runtime there) stands in for the reference's libm: both are faithfully rounded
 * double routines whose results are narrowed to float right away (pathTracing.cpp:69,131).
 */
#include "oracle.h"

#include <omp.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <random>
#include <vector>

#include "trt_prims.h"

namespace {

// glm::vec3 arithmetic in glm's evaluation order
struct V3 {
    float x, y, z;
};
inline V3 mk(float x, float y, float z) { return V3{x, y, z}; }
inline V3 ld(const float* p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline float length2(V3 a) { return dot(a, a); }
inline V3 normalize(V3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
inline float gmin(float a, float b) { return (b < a) ? b : a; }
inline float gmax(float a, float b) { return (a < b) ? b : a; }
inline V3 reflect(V3 I, V3 N) { return I - (N * dot(N, I)) * 2.0f; }
inline V3 refract(V3 I, V3 N, float eta)
{
    const float dn = dot(N, I);
    const float k = 1.0f - eta * eta * (1.0f - dn * dn);
    if (k < 0.0f) return mk(0.f, 0.f, 0.f);
    return I * eta - N * (eta * dn + sqrtf(k));
}

// Triangle (triangle.h:9-26) as readobj leaves it (scene.cpp:196)
struct Tri {
    V3 v[3];
    V3 normal;
};

// triangle.cpp:12-29: min ||A b - B|| for A = [v0 v1 v2; 1 1 1] (4x3), B = [P; 1], in double, by Householder
// QR with column pivoting.
inline V3 findBaryCor(const Tri& T, V3 hitp)
{
    double A[4][3] = {{T.v[0].x, T.v[1].x, T.v[2].x}, {T.v[0].y, T.v[1].y, T.v[2].y}, {T.v[0].z, T.v[1].z, T.v[2].z}, {1.0, 1.0, 1.0}};
    double B[4] = {hitp.x, hitp.y, hitp.z, 1.0};
    int perm[3] = {0, 1, 2};
    int rank = 3;
    for (int k = 0; k < 3; ++k) {
        // pivot: remaining column of largest norm
        int best = k;
        double best_n = -1.0;
        for (int c = k; c < 3; ++c) {
            double n2 = 0.0;
            for (int r = k; r < 4; ++r) n2 += A[r][c] * A[r][c];
            if (n2 > best_n) { best_n = n2; best = c; }
        }
        if (best != k) {
            for (int r = 0; r < 4; ++r) std::swap(A[r][k], A[r][best]);
            std::swap(perm[k], perm[best]);
        }
        if (!(best_n > 0.0)) { rank = k; break; }
        // Householder vector for column k, rows k..3
        const double norm = std::sqrt(best_n);
        const double alpha = A[k][k] > 0.0 ? -norm : norm;
        double v[4] = {0, 0, 0, 0};
        for (int r = k; r < 4; ++r) v[r] = A[r][k];
        v[k] -= alpha;
        double vv = 0.0;
        for (int r = k; r < 4; ++r) vv += v[r] * v[r];
        if (vv > 0.0) {
            for (int c = k; c < 3; ++c) {
                double s = 0.0;
                for (int r = k; r < 4; ++r) s += v[r] * A[r][c];
                s = 2.0 * s / vv;
                for (int r = k; r < 4; ++r) A[r][c] -= s * v[r];
            }
            double s = 0.0;
            for (int r = k; r < 4; ++r) s += v[r] * B[r];
            s = 2.0 * s / vv;
            for (int r = k; r < 4; ++r) B[r] -= s * v[r];
        }
    }
    double y[3] = {0, 0, 0};
    for (int k = rank - 1; k >= 0; --k) {
        double s = B[k];
        for (int c = k + 1; c < rank; ++c) s -= A[k][c] * y[c];
        y[k] = s / A[k][k];
    }
    double b[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) b[perm[k]] = y[k];
    return mk((float)b[0], (float)b[1], (float)b[2]);
}

struct HitRecord {  // bvh.h:7-15
    bool is_hit = false;
    float distance = TRT_INF;
    V3 hitpoint = mk(0, 0, 0);
    V3 pn = mk(0, 0, 0);
    int32_t tri = -1;
};

struct Counters {
    uint64_t rays[3] = {0, 0, 0};
    uint64_t shaded = 0;
    uint32_t max_bounces = 0;
};

// Which of the reference's five engines a draw comes from (main.cpp:57, pathTracing.cpp:33,106,113,149).
enum { ENG_MAIN = 0, ENG_SHADE = 1, ENG_RR = 2, ENG_SAMPLE = 3, ENG_NEXTRAY = 4 };
// EXPERIMENT (oracle_render_literal_experiment): the reference's own random sources — five function-local static
// std::default_random_engine objects shared by every sample.  MSVC's default_random_engine is mt19937; main's is seeded with
// time(NULL) before the scene is loaded, shade's, Sample's and nextRay's with time(NULL) at their first call (the same second:
// IDENTICAL streams), RR's is default-seeded.  uniform_real_distribution<double> takes two 32-bit outputs per draw.
struct Engines {
    std::mt19937 e[5];
    std::uniform_real_distribution<double> u{0.0, 1.0};
    Engines(uint32_t seed_main, uint32_t seed_shade, uint32_t seed_sample, uint32_t seed_nextray)
    {
        e[ENG_MAIN].seed(seed_main); e[ENG_SHADE].seed(seed_shade); e[ENG_SAMPLE].seed(seed_sample); e[ENG_NEXTRAY].seed(seed_nextray);
        e[ENG_RR].seed(std::mt19937::default_seed);
    }
};
struct Stream {
    trt_rng_key key;
    uint32_t ctr;
    Engines* eng = nullptr;  // experiment only: draws come from the shared engines instead of the per-sample counter stream
    double next(int which) { return eng ? eng->u(eng->e[which]) : (double)trt_rng_uniform(key, ctr++); }  // the reference's distributions are double
};

struct Literal {
    const trt_scene* s;
    std::vector<Tri> tris;
    std::vector<double> light_area;               // Material::area per light (scene.cpp:202)
    std::vector<std::vector<double>> light_cum;   // Triangle::area of the light's copies (scene.cpp:203)
    bool ray_offset = false;  // TRT_FLAG_RAY_OFFSET (not the reference's: lets the tolerance be measured with Q6 out of the way)
    bool specular_ks = false; // TRT_FLAG_SPECULAR_KS: `case SPECULAR: L_indir += m.Ks * intensity` — what the revision behind the staircase snapshots had — instead of the committed Kd (pathTracing.cpp:91-93)
    // origin of a ray leaving the hit in direction w: the hit point, or eps off the surface on w's side (include/trt.h)
    V3 rayOrigin(const HitRecord& rec, V3 w) const
    {
        if (!ray_offset) return rec.hitpoint;
        const V3 P = rec.hitpoint;
        const float eps = TRT_OFFSET_EPS * fmaxf(1.0f, fmaxf(fabsf(P.x), fmaxf(fabsf(P.y), fabsf(P.z))));
        const V3 off = tris[rec.tri].normal * eps;
        return dot(off, w) >= 0.0f ? P + off : P - off;
    }

    explicit Literal(const trt_scene* sc) : s(sc)
    {
        tris.resize(sc->n_tris);
        for (uint32_t i = 0; i < sc->n_tris; ++i) {
            const float* p = sc->tri_v + (size_t)i * 9;
            Tri& t = tris[i];
            t.v[0] = ld(p); t.v[1] = ld(p + 3); t.v[2] = ld(p + 6);
            t.normal = normalize(cross(t.v[1] - t.v[0], t.v[2] - t.v[0]));  // scene.cpp:196
        }
        light_area.resize(sc->n_lights);
        light_cum.resize(sc->n_lights);
        for (uint32_t l = 0; l < sc->n_lights; ++l) {
            const trt_light& L = sc->lights[l];
            double run = 0.0;
            for (uint32_t k = 0; k < L.tri_count; ++k) {
                const trt_light_tri& lt = sc->light_tris[L.tri_first + k];
                // calAera, triangle.cpp:3-10
                const double a = length(ld(lt.v[1]) - ld(lt.v[0])), b = length(ld(lt.v[2]) - ld(lt.v[0])), c = length(ld(lt.v[2]) - ld(lt.v[1]));
                const double cos_c = (a * a + b * b - c * c) / (2 * a * b);
                const double sin_c = std::sqrt(1 - std::pow(cos_c, 2));
                run += a * b * sin_c / 2;
                light_cum[l].push_back(run);
            }
            light_area[l] = run;
        }
    }
    bool emissive(int32_t tri) const { return s->materials[s->tri_mat[tri]].is_emissive != 0; }

    // bvh.cpp:177-209
    bool interactTriangle(const Tri& T, V3 S, V3 d, float& t_out, V3& P_out) const
    {
        const V3 p1 = T.v[0], p2 = T.v[1], p3 = T.v[2], N = T.normal;
        if (fabsf(dot(N, d)) < 0.00001f) return false;
        const float t = dot(p1 - S, N) / dot(d, N);
        if (t < 0.0005f) return false;
        const V3 P = S + d * t;
        const V3 c1 = cross(p2 - p1, P - p1);
        const V3 c2 = cross(p3 - p2, P - p2);
        const V3 c3 = cross(p1 - p3, P - p3);
        const double dir1 = dot(c1, N), dir2 = dot(c2, N), dir3 = dot(c3, N);
        const bool r1 = dir1 > 0 && dir2 > 0 && dir3 > 0;
        const bool r2 = dir1 < 0 && dir2 < 0 && dir3 < 0;
        if (!(r1 || r2)) return false;
        t_out = t;
        P_out = P;
        return true;
    }

    // bvh.cpp:211-229
    HitRecord interactBVHNode(V3 S, V3 d, uint32_t first, uint32_t count) const
    {
        HitRecord res;
        for (uint32_t i = first; i < first + count; ++i) {
            float t;
            V3 P;
            if (!interactTriangle(tris[i], S, d, t, P)) continue;
            if ((t == res.distance && emissive((int32_t)i)) || t < res.distance) {
                res.is_hit = true;
                res.distance = t;
                res.hitpoint = P;
                res.tri = (int32_t)i;
                const V3 bc = findBaryCor(tris[i], P);
                const float* vn = s->tri_vn + (size_t)i * 9;
                res.pn = normalize((ld(vn) * bc.x + ld(vn + 3) * bc.y) + ld(vn + 6) * bc.z);
            }
        }
        return res;
    }

    // bvh.cpp:231-245 (1.0 / d in double narrowed to float is the correctly rounded float quotient)
    static float interactAABB(const float* AA, const float* BB, V3 o, V3 inv)
    {
        const V3 in = (ld(BB) - o) * inv, out = (ld(AA) - o) * inv;
        const V3 tmax = mk(gmax(in.x, out.x), gmax(in.y, out.y), gmax(in.z, out.z));
        const V3 tmin = mk(gmin(in.x, out.x), gmin(in.y, out.y), gmin(in.z, out.z));
        const float t1 = gmin(tmax.x, gmin(tmax.y, tmax.z));
        const float t0 = gmax(tmin.x, gmax(tmin.y, tmin.z));
        return (t1 >= t0) ? ((t0 > 0.0) ? t0 : t1) : -1.0f;
    }

    // bvh.cpp:146-175
    HitRecord traverse(uint32_t ref, V3 S, V3 d, V3 inv) const
    {
        if (ref & TRT_LEAF_BIT) return interactBVHNode(S, d, TRT_LEAF_FIRST(ref), TRT_LEAF_COUNT(ref));
        const trt_bvh_node& n = s->nodes[ref];
        const float d1 = interactAABB(n.lo0, n.hi0, S, inv);
        const float d2 = interactAABB(n.lo1, n.hi1, S, inv);
        HitRecord r1, r2;
        if (d1 > 0) r1 = traverse(n.child0, S, d, inv);
        if (d2 > 0) r2 = traverse(n.child1, S, d, inv);
        if (r1.distance == r2.distance) return (r1.tri >= 0 && emissive(r1.tri)) ? r1 : r2;
        return r1.distance < r2.distance ? r1 : r2;
    }
    HitRecord traverseBVH(V3 S, V3 d) const
    {
        const V3 inv = mk((float)(1.0 / d.x), (float)(1.0 / d.y), (float)(1.0 / d.z));
        return traverse(0u, S, d, inv);
    }

    // pathTracing.cpp:111-145
    static V3 Sample(V3 direction, int ray_type, double Ns, Stream& rng)
    {
        const double phi = rng.next(ENG_SAMPLE) * 2 * (double)TRT_PI;
        double theta;
        if (ray_type == TRT_RAY_DIFFUSE) theta = std::asin(std::sqrt(rng.next(ENG_SAMPLE)));
        else theta = std::acos(std::pow(rng.next(ENG_SAMPLE), (double)1 / (Ns + 1)));
        const V3 sample = mk((float)(std::sin(theta) * std::cos(phi)), (float)std::cos(theta), (float)(std::sin(theta) * std::sin(phi)));
        V3 front;
        if (fabsf(direction.x) > fabsf(direction.y)) front = normalize(mk(direction.z, 0, -direction.x));
        else front = normalize(mk(0, -direction.z, direction.y));
        const V3 right = cross(direction, front);
        return normalize((right * sample.x + direction * sample.y) + front * sample.z);
    }

    // pathTracing.cpp:147-209
    static int nextRay(const trt_material& m, V3 pn, V3 ray_direction, Stream& rng, V3& next_dir)
    {
        if (m.Ni > 1) {
            double n1, n2;
            const double cos_in = dot(ray_direction, pn);
            V3 normal;
            if (cos_in > 0) { normal = -pn; n1 = m.Ni; n2 = 1.0; }
            else { normal = pn; n1 = 1.0; n2 = m.Ni; }
            const double rf0 = std::pow((n1 - n2) / (n1 + n2), 2);
            const double fresnel = rf0 + (1.0f - rf0) * std::pow(1.0f - std::abs(cos_in), 5);
            if (fresnel < rng.next(ENG_NEXTRAY)) {
                next_dir = refract(ray_direction, normal, (float)(n1 / n2));
                if (next_dir.x != 0.0f || next_dir.y != 0.0f || next_dir.z != 0.0f) return TRT_RAY_TRANSMISSION;
                next_dir = reflect(ray_direction, normal);
                return TRT_RAY_SPECULAR;
            }
        }
        const double Kd_len = length(ld(m.Kd)), Ks_len = length(ld(m.Ks));
        const double kd = Kd_len / (Kd_len + Ks_len), ks = Ks_len / (Kd_len + Ks_len);
        const double p = rng.next(ENG_NEXTRAY);
        if (p < kd) {
            next_dir = Sample(pn, TRT_RAY_DIFFUSE, m.Ns, rng);
            return TRT_RAY_DIFFUSE;
        }
        if (m.Ns > 1 && p < kd + ks) {
            next_dir = Sample(reflect(ray_direction, pn), TRT_RAY_SPECULAR, m.Ns, rng);
            return TRT_RAY_SPECULAR;
        }
        next_dir = mk(0, 0, 0);
        return TRT_RAY_INVALID;
    }

    // pathTracing.cpp:3-102
    V3 shade(const HitRecord& rec, V3 wi, Stream& rng, Counters& cnt, uint32_t depth) const
    {
        if (depth > cnt.max_bounces) cnt.max_bounces = depth;
        const trt_material& m = s->materials[s->tri_mat[rec.tri]];
        if (m.is_emissive) return ld(m.radiance);
        cnt.shaded++;
        V3 L_dir = mk(0, 0, 0), L_indir = mk(0, 0, 0);
        V3 Kd;
        if (m.tex >= 0) {
            const trt_texture& tx = s->textures[m.tex];
            const V3 bc = findBaryCor(tris[rec.tri], rec.hitpoint);
            const float* vt = s->tri_vt + (size_t)rec.tri * 6;
            const double col = (vt[0] * bc.x + vt[2] * bc.y) + vt[4] * bc.z;  // float expression widened on assignment (:20-21)
            const double row = (vt[1] * bc.x + vt[3] * bc.y) + vt[5] * bc.z;
            const double irow = row - std::floor(row), icol = col - std::floor(col);
            int r = (int)(irow * tx.height), c = (int)(icol * tx.width);
            if (r > tx.height - 1) r = tx.height - 1;  // the reference would read out of bounds
            if (c > tx.width - 1) c = tx.width - 1;
            if (r < 0) r = 0;
            if (c < 0) c = 0;
            const uint8_t* px = tx.rgb + ((size_t)r * tx.width + c) * 3;
            Kd = mk((float)((double)px[0] / 255), (float)((double)px[1] / 255), (float)((double)px[2] / 255));
        } else {
            Kd = ld(m.Kd);
        }
        const double area0 = s->n_lights ? light_area[0] : 0.0;  // Q3: the static distribution of :38
        for (uint32_t li = 0; li < s->n_lights; ++li) {
            const trt_light& L = s->lights[li];
            const double total_area = light_area[li];
            const double rnd = rng.next(ENG_SHADE) * area0;
            for (uint32_t k = 0; k < L.tri_count; ++k) {
                if (!(rnd < light_cum[li][k])) continue;
                const trt_light_tri& lt = s->light_tris[L.tri_first + k];
                const double rnd1 = rng.next(ENG_SHADE), rnd2 = rng.next(ENG_SHADE), rnd3 = rng.next(ENG_SHADE);
                const float p1 = (float)(rnd1 / (rnd1 + rnd2 + rnd3)), p2 = (float)(rnd2 / (rnd1 + rnd2 + rnd3)), p3 = (float)(rnd3 / (rnd1 + rnd2 + rnd3));
                const V3 light_p = (ld(lt.v[0]) * p1 + ld(lt.v[1]) * p2) + ld(lt.v[2]) * p3;
                const V3 light_n = normalize((ld(lt.vn[0]) * p1 + ld(lt.vn[1]) * p2) + ld(lt.vn[2]) * p3);
                const V3 wo = normalize(light_p - rec.hitpoint);
                if (dot(wo, rec.pn) > 0) {  // :60 (tested before the trace here; the trace has no side effect)
                    cnt.rays[1]++;
                    const HitRecord rec_sample = traverseBVH(rayOrigin(rec, wo), wo);
                    const bool visibility = rec_sample.tri >= 0 && s->tri_mat[rec_sample.tri] == L.mat;  // :55 (a miss carries mtl_name "")
                    if (visibility) {
                        const float pdf_light = (float)(double(1) / total_area);
                        const float cos_theta_p = fabsf(dot(wo, light_n));
                        const float cos_theta = fabsf(dot(wo, rec.pn) / length(rec.pn));
                        const V3 radiance = ld(s->materials[L.mat].radiance);
                        const V3 intensity = (((radiance * cos_theta_p) * cos_theta) / length2(light_p - rec.hitpoint)) / pdf_light;
                        const V3 h = normalize((wi + wo) * 0.5f);
                        const double cos_alpha = std::fmax(dot(rec.pn, h), 0);
                        L_dir = L_dir + intensity * (Kd / TRT_PI + ((ld(m.Ks) * (m.Ns + 2.0f)) * (float)std::pow(cos_alpha, m.Ns)) / (2.0f * TRT_PI));
                    }
                }
                break;
            }
        }
        if (rng.next(ENG_RR) < (double)TRT_P_RR) {  // RR(P_RR), :104-109
            V3 nd;
            const int type = nextRay(m, rec.pn, -wi, rng, nd);
            if (type != TRT_RAY_INVALID) {
                cnt.rays[2]++;
                const HitRecord ret = traverseBVH(rayOrigin(rec, nd), nd);
                if (ret.is_hit) {
                    const V3 intensity = shade(ret, -nd, rng, cnt, depth + 1) / TRT_P_RR;
                    if (type == TRT_RAY_TRANSMISSION) L_indir = L_indir + ld(m.Tr) * intensity;
                    else if (!emissive(ret.tri)) L_indir = L_indir + ((specular_ks && type == TRT_RAY_SPECULAR) ? ld(m.Ks) : Kd) * intensity;
                }
            }
        }
        return L_dir + L_indir;
    }
};

inline bool rowSelected(const trt_params* p, int y)
{
    if (p->row_mod <= 1) return true;
    return ((y / p->row_block) % p->row_mod) == p->row_rem;
}

}  // namespace

extern "C" {

int oracle_render_literal(const trt_scene* scene, const trt_params* p, float* out_rgb, oracle_stats* stats, int threads)
{
    if (!scene || !p || !out_rgb) return TRT_EINVAL;
    if (p->width < 2 || p->height < 2 || p->spp < 1) return TRT_EINVAL;
    if (p->x0 < 0 || p->y0 < 0 || p->x1 > p->width || p->y1 > p->height || p->x0 >= p->x1 || p->y0 >= p->y1) return TRT_EINVAL;
    if (scene->n_nodes < 1 || !scene->nodes) return TRT_EINVAL;
    if (p->max_depth != 0 || (p->flags & (TRT_FLAG_FIXED_NEE | TRT_FLAG_FIXED_PIXELS))) return TRT_EINVAL;  // the reference has neither
    Literal lit(scene);
    lit.ray_offset = (p->flags & TRT_FLAG_RAY_OFFSET) != 0;
    lit.specular_ks = (p->flags & TRT_FLAG_SPECULAR_KS) != 0;
    std::vector<int> rows;
    for (int y = p->y0; y < p->y1; ++y)
        if (rowSelected(p, y)) rows.push_back(y);
    const int tw = p->x1 - p->x0;
    if (threads <= 0) threads = omp_get_num_procs();
    Counters total;
    const auto t_begin = std::chrono::steady_clock::now();
#pragma omp parallel num_threads(threads)
    {
        Counters cnt;
#pragma omp for schedule(dynamic, 1)
        for (long r = 0; r < (long)rows.size(); ++r) {
            const int i = rows[(size_t)r];
            for (int j = p->x0; j < p->x1; ++j) {
                double acc[3] = {0, 0, 0};
                const uint32_t pixel = (uint32_t)i * (uint32_t)p->width + (uint32_t)j;
                for (int k = 0; k < p->spp; ++k) {
                    Stream rng{trt_rng_make_key(p->seed, pixel, (uint32_t)k), 0};
                    double x = double(j) / double(p->width - 1.0);      // main.cpp:88-93
                    double y = double(p->height - i) / double(p->height - 1.0);
                    x += (rng.next(ENG_MAIN) - 0.5f) / double(p->width);
                    y += (rng.next(ENG_MAIN) - 0.5f) / double(p->height);
                    // camera.cpp:19-28
                    const trt_camera& cam = scene->camera;
                    const float sf = (float)x, tf = (float)y;
                    const V3 eye = ld(cam.eye);
                    const V3 dir = normalize(((ld(cam.lower_left_corner) + ld(cam.horizontal) * sf) + ld(cam.vertical) * tf) - eye);
                    cnt.rays[0]++;
                    const HitRecord rec = lit.traverseBVH(eye, dir);
                    V3 color = mk(0, 0, 0);
                    if (rec.is_hit) color = lit.shade(rec, -dir, rng, cnt, 0) / (float)p->spp;  // main.cpp:101
                    acc[0] += color.x;
                    acc[1] += color.y;
                    acc[2] += color.z;
                }
                float* px = out_rgb + ((size_t)r * tw + (size_t)(j - p->x0)) * 3;
                px[0] = (float)acc[0];
                px[1] = (float)acc[1];
                px[2] = (float)acc[2];
            }
        }
#pragma omp critical
        {
            for (int i = 0; i < 3; ++i) total.rays[i] += cnt.rays[i];
            total.shaded += cnt.shaded;
            total.max_bounces = std::max(total.max_bounces, cnt.max_bounces);
        }
    }
    const auto t_end = std::chrono::steady_clock::now();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays_camera = total.rays[0];
        stats->rays_shadow = total.rays[1];
        stats->rays_indirect = total.rays[2];
        stats->shaded_hits = total.shaded;
        stats->max_bounces = total.max_bounces;
        stats->threads = threads;
        stats->seconds = std::chrono::duration<double>(t_end - t_begin).count();
    }
    return TRT_OK;
}

/* EXPERIMENTS on the two things the parity path deliberately does NOT take from the reference (oracle.h): its accumulation and its
 * random sources.  The loop nest is main.cpp:80-113's: samples outermost, then rows, then columns, color / SAMPLE added to one shared
 * double image.
 *   ORACLE_EXP_RACY_ACCUM: `threads` OpenMP threads over the SAMPLE index, every one of them sweeping the whole image and adding into
 *     the same pixels WITHOUT synchronisation (main.cpp:79-81,103-108).  Draws stay the per-sample counter streams, so the race-free sum is
 *     oracle_render_literal's image exactly and the difference is the mass lost to overwritten updates.  A real data race, on purpose;
 *     never run under a sanitizer.
 *   ORACLE_EXP_SHARED_ENGINES: one thread; the draws come from the reference's five engines (struct Engines): main's seeded p->seed,
 *     shade's, Sample's and nextRay's ALL seeded p->seed + 1 (identical streams, as time(NULL) leaves them), RR's default-seeded.
 *   ORACLE_EXP_INDEPENDENT_ENGINES: the same with three different seeds (the control: what changing the generator alone does). */
#ifndef ORACLE_EXPERIMENTS
/* liboracle.so — the library every parity test loads — does not contain the experiments (one of them is a deliberate data race);
 * oracle/Makefile builds them into liboracle_exp.so (-DORACLE_EXPERIMENTS), which only tests/test_ref_png.py's experiment tests load. */
int oracle_render_literal_experiment(const trt_scene*, const trt_params*, float*, int, int) { return TRT_EINVAL; }
#else
int oracle_render_literal_experiment(const trt_scene* scene, const trt_params* p, float* out_rgb, int threads, int experiment)
{
    if (!scene || !p || !out_rgb) return TRT_EINVAL;
    if (p->width < 2 || p->height < 2 || p->spp < 1 || scene->n_nodes < 1 || !scene->nodes) return TRT_EINVAL;
    if (p->x0 != 0 || p->y0 != 0 || p->x1 != p->width || p->y1 != p->height || p->row_mod > 1 || p->max_depth != 0 || p->flags != 0) return TRT_EINVAL;
    const bool racy = experiment == ORACLE_EXP_RACY_ACCUM;
    if (!racy && experiment != ORACLE_EXP_SHARED_ENGINES && experiment != ORACLE_EXP_INDEPENDENT_ENGINES) return TRT_EINVAL;
    Literal lit(scene);
    const int W = p->width, H = p->height, SAMPLE = p->spp;
    std::vector<double> image((size_t)W * H * 3, 0.0);  // main.cpp:74-75
    Engines eng(p->seed, p->seed + 1u, experiment == ORACLE_EXP_SHARED_ENGINES ? p->seed + 1u : p->seed + 2u,
                experiment == ORACLE_EXP_SHARED_ENGINES ? p->seed + 1u : p->seed + 3u);
    if (threads <= 0) threads = omp_get_num_procs();
    if (!racy) threads = 1;
    double* const base = image.data();
#pragma omp parallel for num_threads(threads) schedule(static, 1)
    for (int k = 0; k < SAMPLE; k++) {
        Counters cnt;
        double* px = base;
        for (int i = 0; i < H; i++) {
            for (int j = 0; j < W; j++) {
                Stream rng{trt_rng_make_key(p->seed, (uint32_t)i * (uint32_t)W + (uint32_t)j, (uint32_t)k), 0, racy ? nullptr : &eng};
                double x = double(j) / double(W - 1.0);
                double y = double(H - i) / double(H - 1.0);
                x += (rng.next(ENG_MAIN) - 0.5f) / double(W);
                y += (rng.next(ENG_MAIN) - 0.5f) / double(H);
                const trt_camera& cam = scene->camera;
                const float sf = (float)x, tf = (float)y;
                const V3 eye = ld(cam.eye);
                const V3 dir = normalize(((ld(cam.lower_left_corner) + ld(cam.horizontal) * sf) + ld(cam.vertical) * tf) - eye);
                const HitRecord rec = lit.traverseBVH(eye, dir);
                V3 color = mk(0, 0, 0);
                if (rec.is_hit) color = lit.shade(rec, -dir, rng, cnt, 0) / (float)SAMPLE;
                // main.cpp:103-108: unsynchronised read-modify-write of shared memory (volatile: one load and one store per
                // statement, as the reference's compiled loop has them, instead of whatever the optimiser would merge)
                volatile double* q = px;
                q[0] = q[0] + color.x;
                q[1] = q[1] + color.y;
                q[2] = q[2] + color.z;
                px += 3;
            }
        }
    }
    for (size_t i = 0; i < image.size(); ++i) out_rgb[i] = (float)image[i];
    return TRT_OK;
}
#endif  /* ORACLE_EXPERIMENTS */

/* interactTriangle + findBaryCor of the reference on one triangle: returns 1 on hit, out = {t, b0, b1, b2}. */
int oracle_tri_test_literal(const float v[9], const float o[3], const float d[3], float out[4])
{
    trt_scene dummy;
    std::memset(&dummy, 0, sizeof(dummy));
    Literal lit(&dummy);
    Tri T;
    T.v[0] = ld(v); T.v[1] = ld(v + 3); T.v[2] = ld(v + 6);
    T.normal = normalize(cross(T.v[1] - T.v[0], T.v[2] - T.v[0]));
    float t;
    V3 P;
    if (!lit.interactTriangle(T, ld(o), ld(d), t, P)) return 0;
    const V3 bc = findBaryCor(T, P);
    out[0] = t; out[1] = bc.x; out[2] = bc.y; out[3] = bc.z;
    return 1;
}

/* traverseBVH of the reference (literal triangle test) on a ray batch: t, tri, and the hit's barycentric weights of v1, v2. */
int oracle_trace_literal(const trt_scene* scene, uint64_t n, const float* org, const float* dir, float* t, int32_t* tri, float* uv)
{
    if (!scene || !org || !dir || !t || !tri || scene->n_nodes < 1) return TRT_EINVAL;
    const Literal lit(scene);
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        const HitRecord h = lit.traverseBVH(ld(org + i * 3), ld(dir + i * 3));
        t[i] = h.distance;
        tri[i] = h.tri;
        if (uv) {
            uv[i * 2] = 0.f; uv[i * 2 + 1] = 0.f;
            if (h.tri >= 0) {
                const V3 bc = findBaryCor(lit.tris[h.tri], h.hitpoint);
                uv[i * 2] = bc.y; uv[i * 2 + 1] = bc.z;
            }
        }
    }
    return TRT_OK;
}

}  // extern "C"
