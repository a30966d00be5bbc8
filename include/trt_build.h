/*
 * trt_build.h — C-ABI of the GPU BVH builder (libtrt_lbvh.so, MI355X / gfx950).
 *
 * Replaces, for scenes of millions of triangles, the call
 *     BVHNode* root = buildBVH(scene.triangles, 0, scene.triangles.size() - 1, leaf_num);     (main.cpp:76, bvh.cpp:16-144)
 * The reference's builder sorts `scene.triangles` in place while it recurses, so that a leaf is a contiguous range of the
 * array; this one returns the same two things in flat form: the node array trt_create() takes (trt.h, trt_bvh_node) and the
 * permutation that puts the caller's triangles into leaf order.  The tree is an LBVH (Lauterbach et al. 2009; hierarchy of
 * Karras 2012): 63-bit Morton codes of the centres of the triangles' boxes, one radix sort, every inner node from its
 * own index, boxes bottom-up.  Its top is SAH: the radix tree is cut into its maximal subtrees of <= 2048 triangles
 * (n / 64 for scenes below 131 k triangles, at least 256; TRT_LBVH_CLUSTER in the environment overrides; 0 = keep the radix tree as it is), an exact sweep-SAH tree over those clusters — a few
 * thousand boxes, built on the host in milliseconds — becomes the upper part of the BVH, and each cluster's radix subtree
 * hangs below its leaf (node visits per ray within 4-7 % of the host SAH builder's tree on scenes of 1-10 M triangles, against
 * 4-23 % for the radix tree alone).  Topology is free
 * (SURVEY.md §8a Q10): hits, tie rules and images depend on the tree only through the leaf order, which trt_create checks
 * (validateBvh) as for every caller's tree, and the oracle walks the very same nodes, so the parity tests hold unchanged.
 *
 * A separate library from libtrt_hip.so on purpose: the render path neither needs nor loads it.
 */
#ifndef TRT_BUILD_H
#define TRT_BUILD_H

#include <stdint.h>

#include "trt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tri_v: n_tris * 9 floats (host), the vertices as in trt_scene.tri_v, in the caller's current order.
 * leaf_num: 1..15 triangles per leaf at most (main.cpp:76 passes 8).
 * nodes_out: host array of node_capacity nodes; n_tris - 1 always suffices (max(n_tris, 2) - 1 inner nodes at most).  Root at [0].
 *            Child boxes are the exact bounds padded by -/+ 0.001f, as the reference pads every node (bvh.cpp:31-40).
 * order_out: n_tris entries; order_out[i] = index in the caller's arrays of the triangle that belongs at position i.
 *            The caller permutes tri_v / tri_vn / tri_vt / tri_mat accordingly before trt_create (trth_scene_adopt_bvh does).
 * depth_out: inner nodes on the longest root path (informational; trt_create measures it again).
 * ms_out:    optional, [0] = from the first kernel to the last (hipEvents; the host's SAH over the clusters lies in between), [1] = the whole call on the host clock.
 * Returns TRT_OK or a trt.h error code; message in trt_build_last_error(). */
int trt_build_lbvh(const float* tri_v, uint32_t n_tris, int leaf_num, int device, trt_bvh_node* nodes_out, uint32_t node_capacity,
                   uint32_t* n_nodes_out, uint32_t* order_out, uint32_t* depth_out, double ms_out[2]);

const char* trt_build_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* TRT_BUILD_H */
