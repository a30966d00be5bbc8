// capi.cpp — extern "C" surface of libtrt_host.so (include/trt_host.h).
#include <cstring>
#include <exception>
#include <memory>
#include <string>

#include "image_out.h"
#include "scene.h"
#include "trt_host.h"

namespace trt {
bool decodeJPEG(const std::string& path, std::vector<uint8_t>& rgb, int& width, int& height);
bool decodePNG(const std::string& path, std::vector<uint8_t>& rgb, int& width, int& height);
}

struct trth_scene {
    trt::Scene scene;
    trt::FlatBVH bvh;
    std::unique_ptr<trt::FlatScene> flat;
};

namespace {
thread_local std::string g_err;
int fail(const std::exception& e) { g_err = e.what(); return 1; }
int fail(const char* msg) { g_err = msg; return 1; }
}  // namespace

extern "C" {

const char* trth_last_error(void) { return g_err.c_str(); }

trth_scene* trth_scene_load(const char* xml_path, const char* obj_path, const char* mtl_path, const char* basedir, int width, int height)
{
    return trth_scene_load_opts(xml_path, obj_path, mtl_path, basedir, width, height, 0);
}

trth_scene* trth_scene_load_opts(const char* xml_path, const char* obj_path, const char* mtl_path, const char* basedir, int width, int height, int triangulate_polygons)
{
    if (!xml_path || !obj_path || !mtl_path || !basedir) { fail("trth_scene_load: null path"); return nullptr; }
    std::unique_ptr<trth_scene> s(new trth_scene);
    s->scene.triangulate_polygons = triangulate_polygons != 0;
    try {
        s->scene.readxml(xml_path);
        if (width > 0 && height > 0) s->scene.setResolution(width, height);
        s->scene.readobj(obj_path);
        s->scene.readmtl(mtl_path, basedir);
    } catch (const std::exception& e) {
        fail(e);
        return nullptr;
    }
    return s.release();
}

int trth_scene_drop_tris(trth_scene* s, uint32_t first, uint32_t count)
{
    if (!s) return fail("null scene");
    if (s->flat) return fail("scene already built");
    auto& t = s->scene.triangles;
    if ((uint64_t)first + count > t.size()) return fail("trth_scene_drop_tris: range out of bounds");
    for (uint32_t i = first; i < first + count; ++i)
        if (t[i].is_emissive) return fail("trth_scene_drop_tris: cannot drop light triangles");
    t.erase(t.begin() + first, t.begin() + first + count);
    return 0;
}

int trth_scene_add_soup(trth_scene* s, uint32_t seed, uint64_t n)
{
    if (!s) return fail("null scene");
    if (s->flat) return fail("scene already built");
    try { trt::makeSoupScene(s->scene, seed, n, 0, 0); } catch (const std::exception& e) { return fail(e); }
    return 0;
}

int trth_scene_add_blob(trth_scene* s, uint32_t seed, uint64_t n)
{
    if (!s) return fail("null scene");
    if (s->flat) return fail("scene already built");
    try { trt::makeBlobScene(s->scene, seed, n, 0, 0); } catch (const std::exception& e) { return fail(e); }
    return 0;
}

int trth_scene_build(trth_scene* s, int leaf_num, int builder)
{
    if (!s) return fail("null scene");
    if (builder < 0 || builder > 2) return fail("trth_scene_build: unknown builder");
    try {
        s->bvh = trt::buildBVH(s->scene.triangles, leaf_num, (trt::BvhBuilder)builder);
        s->flat.reset(new trt::FlatScene);
        s->flat->build(s->scene, s->bvh);
    } catch (const std::exception& e) {
        s->flat.reset();
        return fail(e);
    }
    return 0;
}

int trth_scene_vertices(const trth_scene* s, float* out, uint64_t capacity_floats)
{
    if (!s || !out) return fail("null argument");
    const auto& t = s->scene.triangles;
    if (capacity_floats < (uint64_t)t.size() * 9) return fail("trth_scene_vertices: buffer too small");
    const size_t n = t.size();
#pragma omp parallel for schedule(static) num_threads(trt::hostThreads()) if (n >= 100000)
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            out[i * 9 + k * 3 + 0] = t[i].v[k].x;
            out[i * 9 + k * 3 + 1] = t[i].v[k].y;
            out[i * 9 + k * 3 + 2] = t[i].v[k].z;
        }
    return 0;
}

int trth_scene_adopt_bvh(trth_scene* s, const trt_bvh_node* nodes, uint32_t n_nodes, const uint32_t* order, uint32_t depth)
{
    if (!s || !nodes || !order || n_nodes < 1) return fail("trth_scene_adopt_bvh: null argument");
    const size_t n = s->scene.triangles.size();
    {   // `order` must be a permutation of the triangles
        std::vector<uint8_t> used(n, 0);
        for (size_t i = 0; i < n; ++i) {
            if (order[i] >= n || used[order[i]]) return fail("trth_scene_adopt_bvh: order is not a permutation of the triangles");
            used[order[i]] = 1;
        }
    }
    try {
        // The flat arrays are gathered through `order`; the scene's Triangle objects stay where they are (moving 10 M of them costs more
        // than the GPU build saved, and nothing reads their order once the scene is flat).
        s->bvh.nodes.assign(nodes, nodes + n_nodes);
        s->bvh.depth = depth;
        s->flat.reset(new trt::FlatScene);
        s->flat->build(s->scene, s->bvh, order);
    } catch (const std::exception& e) {
        s->flat.reset();
        return fail(e);
    }
    return 0;
}

const trt_scene* trth_scene_flat(const trth_scene* s)
{
    if (!s || !s->flat) { fail("scene not built"); return nullptr; }
    return s->flat->c_scene();
}

int trth_scene_info(const trth_scene* s, int64_t info[8])
{
    if (!s || !info) return fail("null argument");
    info[0] = s->scene.img_width;
    info[1] = s->scene.img_height;
    info[2] = s->scene.n_vertices;
    info[3] = s->scene.n_vn;
    info[4] = s->scene.n_vt;
    info[5] = (int64_t)s->scene.triangles.size();
    info[6] = (int64_t)s->scene.materials.size();
    info[7] = (int64_t)s->scene.lights.size();
    return 0;
}

double trth_scene_light_area(const trth_scene* s, uint32_t i)
{
    if (!s || i >= s->scene.lights.size()) { fail("light index out of range"); return -1.0; }
    auto it = s->scene.material_ids.find(s->scene.lights[i].mtl_name);
    return it == s->scene.material_ids.end() ? -1.0 : s->scene.materials[(size_t)it->second].area;
}

const char* trth_scene_material_name(const trth_scene* s, uint32_t i)
{
    if (!s || i >= s->scene.materials.size()) { fail("material index out of range"); return nullptr; }
    return s->scene.materials[i].name.c_str();
}

void trth_scene_free(trth_scene* s) { delete s; }

int trth_decode_jpeg(const char* path, int* width, int* height, uint8_t* rgb, uint64_t rgb_capacity)
{
    if (!path || !width || !height) return fail("trth_decode_jpeg: null argument");
    std::vector<uint8_t> px;
    int w = 0, h = 0;
    try {
        if (!trt::decodeJPEG(path, px, w, h)) return fail("trth_decode_jpeg: not a JPEG this decoder handles (8-bit Huffman, grey or YCbCr)");
    } catch (const std::exception& e) { return fail(e); }
    *width = w;
    *height = h;
    if (rgb) {
        if (rgb_capacity < px.size()) return fail("trth_decode_jpeg: buffer too small");
        std::memcpy(rgb, px.data(), px.size());
    }
    return 0;
}

int trth_decode_png(const char* path, int* width, int* height, uint8_t* rgb, uint64_t rgb_capacity)
{
    if (!path || !width || !height) return fail("trth_decode_png: null argument");
    std::vector<uint8_t> px;
    int w = 0, h = 0;
    try {
        if (!trt::decodePNG(path, px, w, h)) return fail("trth_decode_png: not a PNG this decoder handles");
    } catch (const std::exception& e) { return fail(e); }
    *width = w;
    *height = h;
    if (rgb) {
        if (rgb_capacity < px.size()) return fail("trth_decode_png: buffer too small");
        std::memcpy(rgb, px.data(), px.size());
    }
    return 0;
}

int trth_abi_sizes(int64_t out[12])
{
    if (!out) return fail("null argument");
    out[0] = sizeof(trt_bvh_node); out[1] = sizeof(trt_material); out[2] = sizeof(trt_light); out[3] = sizeof(trt_light_tri);
    out[4] = sizeof(trt_texture); out[5] = sizeof(trt_camera); out[6] = sizeof(trt_scene); out[7] = sizeof(trt_params);
    out[8] = sizeof(trt_stats); out[9] = TRT_ABI_VERSION; out[10] = 0; out[11] = 0;
    return 0;
}

int trth_tonemap(const float* linear_rgb, int width, int height, uint8_t* out)
{
    if (!linear_rgb || !out || width <= 0 || height <= 0) return fail("trth_tonemap: bad argument");
    std::vector<uint8_t> tmp;
    trt::tonemap(linear_rgb, width, height, tmp);
    std::memcpy(out, tmp.data(), tmp.size());
    return 0;
}

int trth_write_png(const char* path, int width, int height, const float* linear_rgb)
{
    if (!path || !linear_rgb || width <= 0 || height <= 0) return fail("trth_write_png: bad argument");
    std::vector<uint8_t> tmp;
    trt::tonemap(linear_rgb, width, height, tmp);
    return trt::writePNG(path, width, height, tmp.data()) ? 0 : fail("trth_write_png: I/O error");
}

int trth_write_png_bytes(const char* path, int width, int height, const uint8_t* rgb)
{
    if (!path || !rgb || width <= 0 || height <= 0) return fail("trth_write_png_bytes: bad argument");
    return trt::writePNG(path, width, height, rgb) ? 0 : fail("trth_write_png_bytes: I/O error");
}

}  // extern "C"
