#include "render.h"

#include "trt_build.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace trt {

namespace {
const char CKPT_MAGIC[8] = {'T', 'R', 'T', 'A', 'C', 'C', '2', 0};
struct CkptHead {
    char magic[8];
    int32_t width, height, spp, samples_done, max_depth;
    uint32_t seed;
    uint32_t flags, n_triangles, n_nodes, reserved;
    uint64_t scene_hash;
    uint64_t n_doubles;
};
uint64_t fnv1a(const void* data, size_t bytes, uint64_t h)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < bytes; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
// what the accumulated sums depend on besides the render parameters: geometry, materials, lights, camera
uint64_t sceneHash(const trt_scene* s)
{
    uint64_t h = 1469598103934665603ull;
    h = fnv1a(s->tri_v, (size_t)s->n_tris * 9 * sizeof(float), h);
    h = fnv1a(s->tri_vn, (size_t)s->n_tris * 9 * sizeof(float), h);
    h = fnv1a(s->tri_vt, (size_t)s->n_tris * 6 * sizeof(float), h);
    h = fnv1a(s->tri_mat, (size_t)s->n_tris * sizeof(int32_t), h);
    h = fnv1a(s->materials, (size_t)s->n_materials * sizeof(trt_material), h);
    h = fnv1a(s->lights, (size_t)s->n_lights * sizeof(trt_light), h);
    h = fnv1a(&s->camera, sizeof(trt_camera), h);
    return h;
}
}  // namespace

bool readCheckpoint(const std::string& path, Checkpoint& head, std::vector<double>& accum)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    CkptHead h;
    const bool ok = std::fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, CKPT_MAGIC, 8) == 0 && h.width > 0 && h.height > 0 &&
                    h.n_doubles == (uint64_t)h.width * (uint64_t)h.height * 3 && h.samples_done >= 0 && h.samples_done <= h.spp;
    if (ok) {
        accum.resize(h.n_doubles);
        if (std::fread(accum.data(), sizeof(double), accum.size(), f) != accum.size()) { std::fclose(f); throw std::runtime_error("checkpoint " + path + ": truncated"); }
    }
    std::fclose(f);
    if (!ok) throw std::runtime_error("checkpoint " + path + ": not an accumulator file of this program");
    head.width = h.width; head.height = h.height; head.spp = h.spp; head.samples_done = h.samples_done; head.max_depth = h.max_depth; head.seed = h.seed;
    head.flags = h.flags; head.n_triangles = h.n_triangles; head.n_nodes = h.n_nodes; head.scene_hash = h.scene_hash;
    return true;
}

void writeCheckpoint(const std::string& path, const Checkpoint& head, const std::vector<double>& accum)
{
    const std::string tmp = path + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + tmp);
    CkptHead h;
    std::memcpy(h.magic, CKPT_MAGIC, 8);
    h.width = head.width; h.height = head.height; h.spp = head.spp; h.samples_done = head.samples_done; h.max_depth = head.max_depth; h.seed = head.seed;
    h.flags = head.flags; h.n_triangles = head.n_triangles; h.n_nodes = head.n_nodes; h.reserved = 0; h.scene_hash = head.scene_hash;
    h.n_doubles = accum.size();
    const bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && std::fwrite(accum.data(), sizeof(double), accum.size(), f) == accum.size();
    if (std::fclose(f) != 0 || !ok) throw std::runtime_error("cannot write " + tmp);
    if (std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot rename " + tmp + " to " + path);
}

void render(Scene& scene, const RenderOpts& opts, double* image, trt_stats* stats)
{
    FlatBVH bvh;
    std::vector<uint32_t> gpu_order;
    if (opts.gpu_builder) {
        // buildBVH on the device (main.cpp:76): the nodes, and the permutation that is the reference's in-place sort of scene.triangles
        const size_t n = scene.triangles.size();
        std::vector<float> v(std::max<size_t>(n, 1) * 9);
        for (size_t i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) {
                v[i * 9 + k * 3 + 0] = scene.triangles[i].v[k].x;
                v[i * 9 + k * 3 + 1] = scene.triangles[i].v[k].y;
                v[i * 9 + k * 3 + 2] = scene.triangles[i].v[k].z;
            }
        bvh.nodes.resize(std::max<size_t>(n, 2) - 1);
        std::vector<uint32_t> order(std::max<size_t>(n, 1));
        uint32_t n_nodes = 0;
        const int dev = opts.devices.empty() ? opts.device : opts.devices[0];
        // libtrt_lbvh.so is loaded here, on request, and nowhere else: a render with the host builder neither needs nor maps it
        // (include/trt_build.h).  Looked up next to this executable / library first (the in-tree layout), then on the loader's path.
        using build_fn = decltype(&trt_build_lbvh);
        using err_fn = decltype(&trt_build_last_error);
        static void* lib = nullptr;
        if (!lib) {
            Dl_info me{};
            std::string dir;
            if (dladdr(reinterpret_cast<void*>(&sceneHash), &me) && me.dli_fname) {
                dir = me.dli_fname;
                const size_t slash = dir.find_last_of('/');
                dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
            }
            if (!dir.empty()) lib = dlopen((dir + "libtrt_lbvh.so").c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!lib) lib = dlopen("libtrt_lbvh.so", RTLD_NOW | RTLD_LOCAL);
            if (!lib) throw std::runtime_error(std::string("the GPU BVH builder was asked for but libtrt_lbvh.so does not load: ") + dlerror());
        }
        const build_fn build = reinterpret_cast<build_fn>(dlsym(lib, "trt_build_lbvh"));
        const err_fn last_error = reinterpret_cast<err_fn>(dlsym(lib, "trt_build_last_error"));
        if (!build || !last_error) throw std::runtime_error("libtrt_lbvh.so lacks trt_build_lbvh / trt_build_last_error");
        if (build(v.data(), (uint32_t)n, opts.leaf_num, dev, bvh.nodes.data(), (uint32_t)bvh.nodes.size(), &n_nodes, order.data(), &bvh.depth, nullptr) != TRT_OK)
            throw std::runtime_error(std::string("trt_build_lbvh: ") + last_error());
        bvh.nodes.resize(n_nodes);
        gpu_order.swap(order);  // the flat arrays are gathered through it below: position i = scene.triangles[order[i]]
    } else {
        bvh = buildBVH(scene.triangles, opts.leaf_num, opts.builder);
    }
    FlatScene flat;
    flat.build(scene, bvh, gpu_order.empty() ? nullptr : gpu_order.data());

    trt_params p{};
    p.width = scene.img_width;
    p.height = scene.img_height;
    p.spp = opts.spp;
    p.seed = opts.seed;
    p.x0 = 0; p.y0 = 0; p.x1 = p.width; p.y1 = p.height;
    p.row_block = 1; p.row_mod = 1; p.row_rem = 0;
    p.max_depth = opts.max_depth;
    p.flags = (opts.timing ? TRT_FLAG_TIMING : 0u) | (opts.overlap ? TRT_FLAG_OVERLAP : 0u) | (opts.fixed_nee ? TRT_FLAG_FIXED_NEE : 0u) | (opts.fixed_pixels ? TRT_FLAG_FIXED_PIXELS : 0u) | (opts.ray_offset ? TRT_FLAG_RAY_OFFSET : 0u) | (opts.specular_ks ? TRT_FLAG_SPECULAR_KS : 0u);
    p.mem_budget = opts.mem_budget;
    std::vector<float> out((size_t)p.width * p.height * 3);
    int rc = TRT_OK;
    std::string msg;
    const bool progressive = !(opts.every <= 0 && opts.checkpoint.empty() && !opts.on_progress && opts.stop_after <= 0);
    if (opts.devices.size() > 1) {
        // several GPUs: one host thread and one replica of the scene per device, one gather (trt_group_render)
        if (progressive) throw std::runtime_error("progressive / check-pointed renders run on one device");
        trt_group* g = nullptr;
        if (trt_group_create(flat.c_scene(), (int)opts.devices.size(), opts.devices.data(), &g) != TRT_OK) throw std::runtime_error(std::string("trt_group_create: ") + trt_last_error());
        p.row_block = opts.row_block > 0 ? opts.row_block : 8;
        rc = trt_group_render(g, &p, out.data(), stats, nullptr);
        if (rc) msg = trt_last_error();
        trt_group_destroy(g);
        if (rc != TRT_OK) throw std::runtime_error("trt_group_render: " + msg);
        for (size_t i = 0; i < out.size(); ++i) image[i] += (double)out[i];
        return;
    }
    trt_handle* h = nullptr;
    if (trt_create(flat.c_scene(), opts.devices.size() == 1 ? opts.devices[0] : opts.device, &h) != TRT_OK) throw std::runtime_error(std::string("trt_create: ") + trt_last_error());
    if (!progressive) {
        rc = trt_render(h, &p, out.data(), stats);
        if (rc) msg = trt_last_error();
    } else {
        // progressive: the accumulator lives on the host between calls (and in the checkpoint file)
        std::vector<double> accum(out.size(), 0.0);
        int done = 0;
        const uint32_t est_flags = p.flags & (TRT_FLAG_FIXED_NEE | TRT_FLAG_FIXED_PIXELS | TRT_FLAG_RAY_OFFSET | TRT_FLAG_SPECULAR_KS);
        const uint64_t scene_hash = opts.checkpoint.empty() ? 0ull : sceneHash(flat.c_scene());
        try {
            Checkpoint ck;
            std::vector<double> saved;
            if (!opts.checkpoint.empty() && readCheckpoint(opts.checkpoint, ck, saved)) {
                if (ck.width != p.width || ck.height != p.height || ck.spp != p.spp || ck.seed != p.seed || ck.max_depth != p.max_depth)
                    throw std::runtime_error("checkpoint " + opts.checkpoint + " belongs to another render (size, spp, seed or max depth differ)");
                if (ck.flags != est_flags) throw std::runtime_error("checkpoint " + opts.checkpoint + " was accumulated with another estimator (--fixed-nee / --fixed-pixels / --ray-offset differ)");
                if (ck.n_triangles != flat.c_scene()->n_tris || ck.n_nodes != flat.c_scene()->n_nodes || ck.scene_hash != scene_hash)
                    throw std::runtime_error("checkpoint " + opts.checkpoint + " belongs to another scene (triangles, BVH or scene hash differ)");
                accum.swap(saved);
                done = ck.samples_done;
            }
            if (stats) std::memset(stats, 0, sizeof(*stats));
            const int step = opts.every > 0 ? opts.every : p.spp;
            for (size_t i = 0; i < out.size(); ++i) out[i] = (float)accum[i];  // the picture the checkpoint holds (k_finalize's rounding)
            while (done < p.spp && rc == TRT_OK && !(opts.stop_after > 0 && done >= opts.stop_after)) {
                const int end = std::min(p.spp, done + step);
                trt_stats st;
                rc = trt_render_samples(h, &p, done, end, accum.data(), out.data(), &st);
                if (rc) { msg = trt_last_error(); break; }
                done = end;
                if (stats) {
                    stats->rays_camera += st.rays_camera; stats->rays_shadow += st.rays_shadow; stats->rays_indirect += st.rays_indirect;
                    stats->shaded_hits += st.shaded_hits; stats->render_ms += st.render_ms; stats->passes += st.passes;
                    stats->max_bounces = std::max(stats->max_bounces, st.max_bounces); stats->rows_rendered = st.rows_rendered;
                    stats->inner_node_bytes = st.inner_node_bytes;
                    for (int k = 0; k < TRT_MAX_KERNELS; ++k) { stats->launches[k] += st.launches[k]; stats->kernel_ms[k] += st.kernel_ms[k]; }
                }
                if (!opts.checkpoint.empty()) {
                    Checkpoint w;
                    w.width = p.width; w.height = p.height; w.spp = p.spp; w.samples_done = done; w.max_depth = p.max_depth; w.seed = p.seed;
                    w.flags = est_flags; w.n_triangles = flat.c_scene()->n_tris; w.n_nodes = flat.c_scene()->n_nodes; w.scene_hash = scene_hash;
                    writeCheckpoint(opts.checkpoint, w, accum);
                }
                if (opts.on_progress) opts.on_progress(done, out.data());
            }
        } catch (...) {
            trt_destroy(h);
            throw;
        }
        if (rc == TRT_OK && done < p.spp)
            std::fprintf(stderr, "warning: stopped after %d of %d samples: the image holds the sum of %d samples divided by %d (resume from the checkpoint to finish it)\n", done, p.spp, done, p.spp);
    }
    trt_destroy(h);
    if (rc != TRT_OK) throw std::runtime_error("trt_render: " + msg);
    for (size_t i = 0; i < out.size(); ++i) image[i] += (double)out[i];
}

}  // namespace trt
