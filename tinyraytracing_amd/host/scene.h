// scene.h — Scene loaders with the reference's interface (scene.h:19-36):
// readxml / readobj / readmtl, called in that order (main.cpp:66-69).
// Re-implemented without tinyxml2 / OpenCV / glm.  Errors throw
// std::runtime_error instead of exit() (scene.cpp:10,64,122).
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "trt.h"
#include "types.h"

namespace trt {

class Scene {
public:
    void readxml(const std::string& xml_path);                             // scene.cpp:3-55
    void readmtl(const std::string& mtl_path, const std::string& basedir); // scene.cpp:57-113
    void readobj(const std::string& obj_path);                             // scene.cpp:115-213

    // Overrides the XML resolution and recomputes the aspect ratio and the
    // camera basis the way readxml does (scene.cpp:13-15,24).
    void setResolution(int width, int height);

    // materials[name] with create-on-first-use semantics of the reference's
    // unordered_map::operator[] (scene.cpp:51,83,199).
    Material& material(const std::string& name);
    int materialId(const std::string& name);

    int img_width = 0;
    int img_height = 0;
    std::vector<Triangle> triangles;
    std::vector<Light> lights;
    std::vector<Material> materials;  // indexed by material id
    std::unordered_map<std::string, int> material_ids;
    Camera camera;

    // Faces with more than three vertices: the reference reads the first three tokens of an `f` line and silently drops the rest
    // (scene.cpp:162) — that stays the default.  With triangulate_polygons a face of n vertices becomes the fan
    // (v0,v1,v2), (v0,v2,v3), ... in file order (what an OBJ exporter means by a quad).  Set before readobj().
    bool triangulate_polygons = false;

    // loader statistics (the counts the reference prints, scene.cpp:209-212)
    int n_vertices = 0, n_vn = 0, n_vt = 0;
};

// OpenMP team size of the host-side loops over triangles: TRT_HOST_THREADS, else min(16, OpenMP's default) (bvh.cpp says why).
int hostThreads();

// ---- BVH ---------------------------------------------------------------------
enum BvhBuilder {
    BVH_SWEEP_SAH = 0,  // the reference's full-sweep SAH (bvh.cpp:16-144), O(n log^2 n)
    BVH_BINNED_SAH = 1, // 32-bin SAH for large inputs (1M-10M triangles)
    BVH_AUTO = 2        // sweep up to 64k triangles, binned above
};

struct FlatBVH {
    std::vector<trt_bvh_node> nodes;
    uint32_t depth = 0;
};

// Builds the BVH and reorders `triangles` into leaf order, like
// buildBVH(triangles, 0, n-1, leaf_num) at main.cpp:76.  Boxes are padded by
// +-0.001 (bvh.cpp:31-40).  leaf_num <= 15.
FlatBVH buildBVH(std::vector<Triangle>& triangles, int leaf_num, BvhBuilder builder = BVH_AUTO);

// ---- flattening to the C-ABI scene ------------------------------------------------
// Owns every array a trt_scene points to.
class FlatScene {
public:
    FlatScene() {}
    FlatScene(const FlatScene&) = delete;
    FlatScene& operator=(const FlatScene&) = delete;

    // scene.triangles must already be in BVH order (buildBVH ran) — or `order` says where position i's triangle stands in
    // scene.triangles (a tree from another builder, trt_build.h: the flat arrays are gathered through it and the 10 M Triangle
    // objects of a big scene stay where they are).
    void build(const Scene& scene, const FlatBVH& bvh, const uint32_t* order = nullptr);
    const trt_scene* c_scene() const { return &flat; }

    std::vector<float> tri_v, tri_vn, tri_vt;
    std::vector<int32_t> tri_mat;
    std::vector<trt_bvh_node> nodes;
    std::vector<trt_material> materials;
    std::vector<trt_light> lights;
    std::vector<trt_light_tri> light_tris;
    std::vector<trt_texture> textures;
    std::vector<std::vector<uint8_t>> texture_data;
    trt_scene flat{};
};

// ---- synthetic scenes (BASELINE.json configs 3 and 5) ------------------
// The `back` Cornell box (walls + both light quads, without its cube) filled with
// n uniformly placed random triangles.  Deterministic integer-hash generator.
void makeSoupScene(Scene& scene, uint32_t seed, uint64_t n_random, int width, int height);
// A noise-displaced icosphere with >= n_min faces and smooth vertex normals inside the
// `back` box.
void makeBlobScene(Scene& scene, uint32_t seed, uint64_t n_min, int width, int height);

}  // namespace trt
