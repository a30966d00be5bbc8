#include "render.h"

#include <stdexcept>
#include <vector>

namespace trt {

void render(Scene& scene, const RenderOpts& opts, double* image, trt_stats* stats)
{
    FlatBVH bvh = buildBVH(scene.triangles, opts.leaf_num, opts.builder);
    FlatScene flat;
    flat.build(scene, bvh);

    trt_handle* h = nullptr;
    if (trt_create(flat.c_scene(), opts.device, &h) != TRT_OK) throw std::runtime_error(std::string("trt_create: ") + trt_last_error());
    trt_params p{};
    p.width = scene.img_width;
    p.height = scene.img_height;
    p.spp = opts.spp;
    p.seed = opts.seed;
    p.x0 = 0; p.y0 = 0; p.x1 = p.width; p.y1 = p.height;
    p.row_block = 1; p.row_mod = 1; p.row_rem = 0;
    p.max_depth = opts.max_depth;
    p.flags = (opts.timing ? TRT_FLAG_TIMING : 0u) | (opts.overlap ? TRT_FLAG_OVERLAP : 0u);
    p.mem_budget = opts.mem_budget;
    std::vector<float> out((size_t)p.width * p.height * 3);
    const int rc = trt_render(h, &p, out.data(), stats);
    const std::string msg = rc ? trt_last_error() : "";
    trt_destroy(h);
    if (rc != TRT_OK) throw std::runtime_error("trt_render: " + msg);
    for (size_t i = 0; i < out.size(); ++i) image[i] += (double)out[i];
}

}  // namespace trt
