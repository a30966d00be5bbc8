// tinyrt — command-line driver with the flow of the reference's main()
// (main.cpp:44-119): load xml -> obj -> mtl, render, write <basedir>/image<SPP>.png.
// The reference prompts on stdin for basedir / mtl / xml / obj / SPP
// (main.cpp:46-55); the same five values are taken from argv here, plus
// optional overrides.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "image_out.h"
#include "render.h"

static void usage()
{
    std::fprintf(stderr,
                 "usage: tinyrt <basedir> <mtl> <xml> <obj> <spp> [--width W --height H] [--seed S] [--device D | --gpus N | --devices a,b,..]\n"
                 "              [--leaf N] [--gpu-bvh] [--max-depth D] [--out file.png] [--fixed | --fixed-nee | --fixed-pixels] [--ray-offset] [--specular-ks] [--polygons]\n"
                 "              [--every N] [--checkpoint file.acc] [--stop-after M]\n"
                 "                                                    progressive: N samples per step, image rewritten after\n"
                 "                                                    every step, accumulator kept in file.acc (resumes from it)\n");
}

int main(int argc, char** argv)
{
    if (argc < 6) { usage(); return 2; }
    const std::string basedir = argv[1], mtl = argv[2], xml = argv[3], obj = argv[4];
    trt::RenderOpts opts;
    opts.spp = std::atoi(argv[5]);
    opts.timing = true;
    int width = 0, height = 0;
    bool polygons = false;  // fan-triangulate faces of more than three vertices (the reference keeps their first three only)
    std::string out_path;
    for (int i = 6; i < argc; ++i) {
        auto need = [&](const char* flag) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", flag); std::exit(2); }
            return argv[++i];
        };
        if (!std::strcmp(argv[i], "--width")) width = std::atoi(need("--width"));
        else if (!std::strcmp(argv[i], "--height")) height = std::atoi(need("--height"));
        else if (!std::strcmp(argv[i], "--seed")) opts.seed = (uint32_t)std::strtoul(need("--seed"), nullptr, 0);
        else if (!std::strcmp(argv[i], "--device")) opts.device = std::atoi(need("--device"));
        else if (!std::strcmp(argv[i], "--gpus")) {  // devices 0..N-1 of this node: image tiled in row stripes, one RCCL gather
            const int n = std::atoi(need("--gpus"));
            opts.devices.clear();
            for (int d = 0; d < n; ++d) opts.devices.push_back(d);
        } else if (!std::strcmp(argv[i], "--devices")) {  // explicit list; one device may appear several times (rehearsal on one GPU)
            opts.devices.clear();
            const std::string list = need("--devices");
            size_t pos = 0;
            while (pos < list.size()) {
                const size_t c = list.find(',', pos);
                opts.devices.push_back(std::atoi(list.substr(pos, c == std::string::npos ? std::string::npos : c - pos).c_str()));
                if (c == std::string::npos) break;
                pos = c + 1;
            }
        }
        else if (!std::strcmp(argv[i], "--row-block")) opts.row_block = std::atoi(need("--row-block"));
        else if (!std::strcmp(argv[i], "--leaf")) opts.leaf_num = std::atoi(need("--leaf"));
        else if (!std::strcmp(argv[i], "--gpu-bvh")) opts.gpu_builder = true;
        else if (!std::strcmp(argv[i], "--max-depth")) opts.max_depth = std::atoi(need("--max-depth"));
        else if (!std::strcmp(argv[i], "--out")) out_path = need("--out");
        else if (!std::strcmp(argv[i], "--fixed-nee")) opts.fixed_nee = true;
        else if (!std::strcmp(argv[i], "--fixed-pixels")) opts.fixed_pixels = true;
        else if (!std::strcmp(argv[i], "--ray-offset")) opts.ray_offset = true;
        else if (!std::strcmp(argv[i], "--specular-ks")) opts.specular_ks = true;  // the look of the reference's own saved renders (TRT_FLAG_SPECULAR_KS)
        else if (!std::strcmp(argv[i], "--polygons")) polygons = true;
        else if (!std::strcmp(argv[i], "--fixed")) opts.fixed_nee = opts.fixed_pixels = true;
        else if (!std::strcmp(argv[i], "--every")) opts.every = std::atoi(need("--every"));
        else if (!std::strcmp(argv[i], "--checkpoint")) opts.checkpoint = need("--checkpoint");
        else if (!std::strcmp(argv[i], "--stop-after")) opts.stop_after = std::atoi(need("--stop-after"));
        else { usage(); return 2; }
    }
    try {
        const auto t0 = std::chrono::steady_clock::now();
        trt::Scene scene;
        scene.triangulate_polygons = polygons;
        scene.readxml(xml);   // the order cannot be changed (main.cpp:66)
        if (width > 0 && height > 0) scene.setResolution(width, height);
        scene.readobj(obj);
        scene.readmtl(mtl, basedir);
        std::printf("image info:\nwidth: %d height: %d\n", scene.img_width, scene.img_height);
        std::printf("num of vertices: %d\nnum of vn: %d\nnum of vt: %d\nnum of triangles: %d\nnum of materials: %d\n", scene.n_vertices, scene.n_vn,
                    scene.n_vt, (int)scene.triangles.size(), (int)scene.materials.size());
        scene.camera.Print();
        std::vector<double> image((size_t)scene.img_width * scene.img_height * 3, 0.0);
        trt_stats st{};
        if (opts.every > 0 || !opts.checkpoint.empty()) {
            // the picture so far after every step (what a viewer would poll), same file as the final one
            const int w = scene.img_width, hgt = scene.img_height;
            const std::string prog_path = out_path.empty() ? basedir + "/image" + std::to_string(opts.spp) + ".png" : out_path;
            opts.on_progress = [w, hgt, prog_path, &opts](int done, const float* rgb) {
                std::vector<uint8_t> bytes;
                trt::tonemap(rgb, w, hgt, bytes);
                if (!trt::writePNG(prog_path, w, hgt, bytes.data())) std::fprintf(stderr, "cannot write %s\n", prog_path.c_str());
                std::fprintf(stderr, "\r%d / %d samples", done, opts.spp);
            };
        }
        trt::render(scene, opts, image.data(), &st);
        const uint64_t rays = st.rays_camera + st.rays_shadow + st.rays_indirect;
        std::printf("rays: %llu (camera %llu, shadow %llu, indirect %llu)  render %.3f ms  %.1f Mrays/s\n", (unsigned long long)rays,
                    (unsigned long long)st.rays_camera, (unsigned long long)st.rays_shadow, (unsigned long long)st.rays_indirect, st.render_ms,
                    st.render_ms > 0 ? rays / st.render_ms / 1e3 : 0.0);
        bool ok;
        if (out_path.empty()) ok = trt::imshow(image.data(), basedir, std::to_string(opts.spp), scene.img_width, scene.img_height);
        else {
            std::vector<uint8_t> bytes;
            trt::tonemap(image.data(), scene.img_width, scene.img_height, bytes);
            ok = trt::writePNG(out_path, scene.img_width, scene.img_height, bytes.data());
        }
        if (!ok) { std::fprintf(stderr, "cannot write the PNG\n"); return 1; }
        std::fprintf(stderr, "\nDone.\n");
        std::printf("%f\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
