// render.h — host C++ entry that replaces the body of the reference's main()
// (main.cpp:74-113): given a loaded Scene it builds the BVH, flattens, and runs
// the sample/pixel loop on the GPU through the C-ABI (include/trt.h).
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "scene.h"
#include "trt.h"

namespace trt {

struct RenderOpts {
    int spp = 256;            // SAMPLE (main.cpp:13)
    uint32_t seed = 0x5EED0001u;
    int device = 0;
    std::vector<int> devices;  // more than one entry: the image is tiled over these GPUs of the node in interleaved row stripes and
                               // gathered with one ncclGather on devices[0] (trt_group_*, include/trt.h); `device` is then ignored
    int row_block = 8;        // stripe height of that tiling
    int leaf_num = 2;         // the reference calls buildBVH(..., 8) (main.cpp:76); 2 is fastest on the GPU
    BvhBuilder builder = BVH_AUTO;
    bool gpu_builder = false;  // build the BVH on the GPU instead (trt_build_lbvh, include/trt_build.h: LBVH with a SAH top; `builder` is then ignored)
    int max_depth = 0;
    uint64_t mem_budget = 0;
    bool timing = false;
    bool overlap = true;      // two sample passes in flight (TRT_FLAG_OVERLAP)
    bool fixed_pixels = false;  // TRT_FLAG_FIXED_PIXELS: pixel-centred sampling grid instead of quirks Q1/Q2
    bool ray_offset = false;  // TRT_FLAG_RAY_OFFSET: rays start eps off the surface they leave instead of on it (quirk Q6)
    bool specular_ks = false; // TRT_FLAG_SPECULAR_KS: SPECULAR bounces weighted by Ks, the look of the reference's own saved renders (include/trt.h)
    bool fixed_nee = false;   // TRT_FLAG_FIXED_NEE: unbiased light sampling + occlusion-test shadow rays instead of quirks Q3-Q5
    // Progressive output (trt_render_samples): render `every` samples per call (0 = all in one call), hand the
    // image so far to `on_progress`, and keep the accumulator in `checkpoint` (written after every call, atomically;
    // read at start: a file that matches width/height/spp/seed/max_depth resumes where it stopped).
    int every = 0;
    int stop_after = 0;       // > 0: return once this many samples are done (a time-boxed run; resume from the checkpoint later)
    std::string checkpoint;
    std::function<void(int samples_done, const float* rgb)> on_progress;
};

// Accumulator file of a progressive render: header + width*height*3 doubles (the sums of trt_render_samples).
struct Checkpoint {
    int32_t width = 0, height = 0, spp = 0, samples_done = 0, max_depth = 0;
    uint32_t seed = 0;
    uint32_t flags = 0;           // the estimator the sums belong to: TRT_FLAG_FIXED_NEE / TRT_FLAG_FIXED_PIXELS
    uint32_t n_triangles = 0, n_nodes = 0;
    uint64_t scene_hash = 0;      // FNV-1a over the flattened triangles (post-BVH order), materials, lights and the camera
};
bool readCheckpoint(const std::string& path, Checkpoint& head, std::vector<double>& accum);   // false: no such file
void writeCheckpoint(const std::string& path, const Checkpoint& head, const std::vector<double>& accum);

// image: img_width*img_height*3 doubles, zero-initialised by the caller like
// main.cpp:74-75; the averaged linear radiance is ADDED to it (main.cpp:103-108).
// Throws std::runtime_error with the library's message on failure.
void render(Scene& scene, const RenderOpts& opts, double* image, trt_stats* stats = nullptr);

}  // namespace trt
