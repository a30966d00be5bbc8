/*
 * trt_host.h — C-ABI of the host-side library (libtrt_host.so): the scene
 * loaders, BVH builder, flattening and PNG output that sit on either side of
 * the hot path.  These are the reference's Scene::readxml / readobj / readmtl
 * (scene.cpp:3-213), Camera::setCamera (camera.cpp:3-17), buildBVH
 * (bvh.cpp:16-144) and imshow + svpng (main.cpp:19-42, svpng.inc:77-108),
 * re-implemented without glm / Eigen / tinyxml2 / OpenCV.  No HIP in here; the
 * flat scene it produces is what trt_create() (trt.h) consumes.
 *
 * All functions returning int use 0 = ok; the message of the last failure on
 * the calling thread is trth_last_error().
 */
#ifndef TRT_HOST_H
#define TRT_HOST_H

#include "trt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trth_scene trth_scene;

/* builder ids for trth_scene_build */
#define TRTH_BVH_SWEEP_SAH 0
#define TRTH_BVH_BINNED_SAH 1
#define TRTH_BVH_AUTO 2

/* readxml -> readobj -> readmtl in the mandatory order (main.cpp:66-69).
 * width/height > 0 override the XML resolution (aspect recomputed as at
 * scene.cpp:15).  Returns NULL on failure. */
trth_scene* trth_scene_load(const char* xml_path, const char* obj_path, const char* mtl_path,
                            const char* basedir, int width, int height);

/* The same with loader options: triangulate_polygons != 0 turns an `f` line of n > 3 vertices into the fan (v0,v1,v2), (v0,v2,v3), ...
 * instead of keeping its first three vertices only (what the reference does, scene.cpp:162, and what trth_scene_load does). */
trth_scene* trth_scene_load_opts(const char* xml_path, const char* obj_path, const char* mtl_path,
                                 const char* basedir, int width, int height, int triangulate_polygons);

/* Removes triangles [first, first+count) in file order (before the BVH build). */
int trth_scene_drop_tris(trth_scene* s, uint32_t first, uint32_t count);
/* Synthetic geometry added to a loaded base scene (scenes/back): see host/synth.cpp. */
int trth_scene_add_soup(trth_scene* s, uint32_t seed, uint64_t n_random);
int trth_scene_add_blob(trth_scene* s, uint32_t seed, uint64_t n_min_faces);

/* buildBVH(scene.triangles, 0, n-1, leaf_num) (main.cpp:76) + flattening. */
int trth_scene_build(trth_scene* s, int leaf_num, int builder);

/* The other way to a built scene: a tree from another builder (the GPU builder of trt_build.h).  trth_scene_vertices copies the
 * vertices of the triangles in their current order (n_triangles * 9 floats: what that builder takes); trth_scene_adopt_bvh
 * installs the nodes and flattens with the triangles in the builder's order — position i of the flat arrays gets the triangle
 * that stands at order[i], the in-place sort of buildBVH (bvh.cpp:16-144) as one permutation (the scene's own triangle list
 * keeps its order: only the flat scene is what trt_create sees).  The tree is not checked here: trt_create validates every tree it is given. */
int trth_scene_vertices(const trth_scene* s, float* out, uint64_t capacity_floats);
int trth_scene_adopt_bvh(trth_scene* s, const trt_bvh_node* nodes, uint32_t n_nodes, const uint32_t* order, uint32_t depth);

/* Valid after trth_scene_build / trth_scene_adopt_bvh; owned by the scene. */
const trt_scene* trth_scene_flat(const trth_scene* s);

/* info[0..7] = width, height, n_vertices, n_vn, n_vt, n_triangles, n_materials, n_lights */
int trth_scene_info(const trth_scene* s, int64_t info[8]);
/* total area of light i (Material::area) in double */
double trth_scene_light_area(const trth_scene* s, uint32_t i);
const char* trth_scene_material_name(const trth_scene* s, uint32_t i);

void trth_scene_free(trth_scene* s);

/* imshow(): gamma 1/2.2f, clamp, truncate to 8 bit (main.cpp:34-36). out has w*h*3 bytes. */
int trth_tonemap(const float* linear_rgb, int width, int height, uint8_t* out);
/* tonemap + stored-deflate PNG. */
int trth_write_png(const char* path, int width, int height, const float* linear_rgb);
int trth_write_png_bytes(const char* path, int width, int height, const uint8_t* rgb);

/* Material::readinMap()'s JPEG path (material.cpp:3-11 uses cv::imread): baseline or progressive JPEG -> 8-bit RGB with libjpeg's
 * arithmetic.  Call with rgb = NULL to get the size first. */
int trth_decode_jpeg(const char* path, int* width, int* height, uint8_t* rgb, uint64_t rgb_capacity);
/* ... and its PNG path: any colour type and bit depth, interlaced or not -> 8-bit RGB as cv::imread's default flag yields it (palette expanded, grey
 * replicated, 16-bit samples stripped to the high byte, alpha dropped).  Same calling convention. */
int trth_decode_png(const char* path, int* width, int* height, uint8_t* rgb, uint64_t rgb_capacity);

/* sizeof() of the trt.h structs as the C compiler sees them, for binding self-checks:
 * bvh_node, material, light, light_tri, texture, camera, scene, params, stats, ABI version. */
int trth_abi_sizes(int64_t out[12]);

const char* trth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* TRT_HOST_H */
