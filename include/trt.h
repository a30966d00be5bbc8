/*
 * trt.h — C-ABI of the MI355X path-tracing hot path (libtrt_hip.so).
 *
 * This is the drop-in boundary for the per-pixel Monte-Carlo inner loop of
 * TinyRayTracing.  The reference has no FFI; its "render entry point" is the
 * body of main() (RayTracingOnCPU/main.cpp:79-113: the OpenMP sample/pixel
 * loop calling Camera::getRay camera.cpp:19-28, traverseBVH bvh.cpp:146-175
 * and shade pathTracing.cpp:3-102).  Every entry point below replaces a piece
 * of that loop; the citation says which.
 *
 * Plain C: pointers + sizes only, no C++/torch types.  All functions return 0
 * on success and a non-zero TRT_E* code on failure; the message is available
 * from trt_last_error() (thread-local).  The library never calls exit()
 * (the reference exit()s from its loaders, scene.cpp:10,64,122, and from
 * Sample(), pathTracing.cpp:129).
 */
#ifndef TRT_H
#define TRT_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRT_ABI_VERSION 4

/* error codes */
#define TRT_OK 0
#define TRT_EINVAL 1   /* bad argument / inconsistent scene */
#define TRT_EHIP 2     /* HIP runtime error */
#define TRT_ENOMEM 3   /* device or host allocation failed */
#define TRT_ENODEV 4   /* no usable gfx950 device */

/* Reference constants (pathtracing.h:11-12, bvh.h:5, ray.h:5-8, bvh.cpp:185,189). */
#define TRT_PI 3.1415926f
#define TRT_P_RR 0.8f
#define TRT_INF 114514.0f
#define TRT_T_MIN 0.0005f
#define TRT_PARALLEL_EPS 0.00001f
#define TRT_RAY_DIFFUSE 0
#define TRT_RAY_SPECULAR 1
#define TRT_RAY_TRANSMISSION 2
#define TRT_RAY_INVALID 3

/* ---- flat scene (what main.cpp hands the loop: scene.triangles in post-BVH
 *      order, scene.materials, scene.lights, scene.camera, root) ------------ */

/* BVH2 inner node, 64 B.  Replaces BVHNode (bvh.h:16-22: two pointers, index,
 * num, AA, BB).  The two children's padded boxes (bvh.cpp:31-40) live in the
 * parent so one 64-B fetch decides both descents (bvh.cpp:156-166).
 * child refs: bit31 = 0 -> index of an inner node;
 *             bit31 = 1 -> leaf: bits 30..27 = triangle count (0..15),
 *                                bits 26..0  = first triangle (post-BVH order).
 * Post-BVH order (what bvh.cpp's in-place sort leaves behind): every triangle under child0 has a lower index than
 * every triangle under child1; trt_create checks it (the tie rule of bvh.cpp:168-172 is applied by index).
 * Boxes: any finite coordinates or +-inf, nested or not (a subtree is entered iff the ray passes its stored box, bvh.cpp:162-166, whatever the boxes above
 * or below it say); a NaN coordinate is refused by trt_create (TRT_EINVAL): glm::min / glm::max pass a NaN on from their first operand only, so the reference's
 * own answer on such a box depends on the axis the NaN sits on.
 * Leaf size: any tree with leaves of 1..15 triangles is walked by the same kernels and gives the reference's hits on THAT tree.  As in the reference
 * (bvh.cpp:151-154) every triangle of a leaf is tested whenever the ray passes the leaf's box, so large leaves cost tests: the reference's own
 * buildBVH(..., 8) tree runs at about 0.8 of the speed of a tree built with 2 (every published number is on leaf 2). */
typedef struct trt_bvh_node {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    uint32_t child0, child1;
    uint32_t reserved[2];
} trt_bvh_node;

#define TRT_LEAF_BIT 0x80000000u
#define TRT_LEAF_COUNT(ref) (((ref) >> 27) & 15u)
#define TRT_LEAF_FIRST(ref) ((ref) & 0x07FFFFFFu)
#define TRT_MAKE_LEAF(first, count) (TRT_LEAF_BIT | ((uint32_t)(count) << 27) | (uint32_t)(first))
#define TRT_MAX_LEAF_TRIS 15u
#define TRT_MAX_TRIS 0x07FFFFFFu

/* Material (material.h:11-33) with the light's radiance folded in
 * (scene.cpp:50-52).  tex = -1 when map_Kd == "". */
typedef struct trt_material {
    float Kd[3], Ks[3], Tr[3];
    float Ns, Ni;
    float radiance[3];
    int32_t is_emissive;
    int32_t tex;
} trt_material;

/* One <light> element, XML order (scene.cpp:25-54).  area = Material::area
 * accumulated in readobj (scene.cpp:201-203).  `radiance` is kept for the caller's convenience only: both the emissive hit
 * (pathTracing.cpp:11) and the direct term (pathTracing.cpp:65) read materials[mat].radiance, and so does this library. */
typedef struct trt_light {
    int32_t mat;          /* material id of mtlname */
    float radiance[3];
    float area;           /* total area A_l of this light's triangles */
    uint32_t tri_first;   /* into light_tris */
    uint32_t tri_count;
} trt_light;

/* Copy of an emissive triangle kept per light material for sampling
 * (Material::triangles, scene.cpp:204); cum_area = Triangle::area, the
 * running total at the time it was appended (scene.cpp:203). */
typedef struct trt_light_tri {
    float v[3][3];
    float vn[3][3];
    float cum_area;
} trt_light_tri;

/* 8-bit RGB texture, row-major, row 0 first as stored in the file (the
 * reference indexes cv::Mat rows directly, pathTracing.cpp:22-25). */
typedef struct trt_texture {
    int32_t width, height;
    const uint8_t* rgb;
} trt_texture;

/* Camera after setCamera() (camera.cpp:3-17). */
typedef struct trt_camera {
    float eye[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
} trt_camera;

typedef struct trt_scene {
    uint32_t n_tris;
    const float* tri_v;      /* [n_tris][3][3]  Triangle::v  (triangle.h:14), post-BVH order */
    const float* tri_vn;     /* [n_tris][3][3]  Triangle::vn (triangle.h:15) */
    const float* tri_vt;     /* [n_tris][3][2]  Triangle::vt (triangle.h:16) */
    const int32_t* tri_mat;  /* [n_tris] material id (replaces Triangle::mtl_name) */
    uint32_t n_nodes;
    const trt_bvh_node* nodes; /* nodes[0] is the root; n_nodes >= 1 */
    uint32_t bvh_depth;      /* max number of inner nodes on a root->leaf path */
    uint32_t n_materials;
    const trt_material* materials;
    uint32_t n_lights;
    const trt_light* lights;
    uint32_t n_light_tris;
    const trt_light_tri* light_tris;
    uint32_t n_textures;
    const trt_texture* textures;
    trt_camera camera;
} trt_scene;

/* ---- render parameters ---------------------------------------------------- */

#define TRT_FLAG_TIMING 1u   /* per-kernel hipEvent timing into trt_stats */
#define TRT_FLAG_COUNT 2u    /* count inner-node visits / triangle tests (stats kernels) */
#define TRT_FLAG_OVERLAP 4u  /* keep two sample passes in flight on two streams (+4-5 % throughput; per-kernel
                              * timings then include the other pass's kernels, so profiling runs leave it off) */
#define TRT_FLAG_FIXED_NEE 8u /* opt out of the reference's next-event-estimation quirks: every light's CDF
                              * draw spans that light's own area, light points are uniform on the chosen triangle, and the
                              * shadow test is an occlusion test up to the light sample (any hit in [0.0005, 0.999 * distance)
                              * blocks; a miss is visible) instead of closest-hit + material comparison; as everywhere, nothing farther than 114514 is seen (Q7).  Off = parity mode. */
#define TRT_FLAG_FIXED_PIXELS 16u /* opt out of the pixel-grid quirks (Q1, Q2): pixel (i, j) samples its own cell
                                  * [j/W, (j+1)/W) x [(H-1-i)/H, (H-i)/H) of the image plane uniformly.  Off = parity mode. */

#define TRT_FLAG_RAY_OFFSET 32u /* opt out of Q6 (rays start ON the surface they leave; only t < 0.0005, bvh.cpp:189, keeps them from
                                  * hitting it again, and at the 1000-unit distances of the Cornell box the hit point's rounding error
                                  * beats that for grazing directions): shadow and continuation rays start at P + s * eps * Ng, Ng the
                                  * hit triangle's geometric normal, s the side the ray leaves to, eps = 1e-4 * max(1, |P.x|, |P.y|, |P.z|).
                                  * Off = parity mode. */
#define TRT_OFFSET_EPS 0.0001f
#define TRT_FLAG_SPECULAR_KS 64u /* the look of the reference's own saved renders: the light that returns along a SPECULAR bounce is weighted by the material's Ks —
                                  * what the revision that wrote example-scenes-cg22/staircase/image*.png did — instead of the texel Kd the committed source
                                  * multiplies by (pathTracing.cpp:91-93, quirk Q8).  With it the HIP path reproduces staircase/image256.png to its noise floor
                                  * (profiles/r04_staircase_residual.txt); scenes whose glossy materials have Kd = Ks (veach-mis) or none (back) do not change.
                                  * Off = parity with the committed source. */

typedef struct trt_params {
    int32_t width, height;   /* full image size (scene.img_width/height, scene.cpp:13-14) */
    int32_t spp;             /* SAMPLE (main.cpp:13,55) */
    uint32_t seed;           /* counter-RNG seed; stream = (seed, pixel y*width+x, sample) */
    /* tile rectangle [x0,x1) x [y0,y1) of the full image rendered by this call */
    int32_t x0, y0, x1, y1;
    /* row interleave for multi-GPU image tiling: only rows y (inside the tile)
     * with ((y / row_block) % row_mod) == row_rem are rendered; output rows are
     * packed in increasing y.  row_mod <= 1 renders every row. */
    int32_t row_block, row_mod, row_rem;
    int32_t max_depth;       /* 0 = unbounded like the reference (pathTracing.cpp:78-99) */
    uint32_t flags;
    uint64_t mem_budget;     /* bytes of HBM for path/queue state; 0 = three quarters of what is free (one pass when it fits) */
} trt_params;

#define TRT_MAX_KERNELS 8
enum {
    TRT_K_GEN_PRIMARY = 0,  /* unused since primary-ray generation is fused into bounce 0 (always 0 launches) */
    TRT_K_TRACE_CLOSEST = 1,
    TRT_K_SHADE = 2,
    TRT_K_TRACE_SHADOW = 3,
    TRT_K_RESOLVE = 4,
    TRT_K_TAIL = 5        /* the last, short-queue bounces of a pass fused into one launch */
};

typedef struct trt_stats {
    uint64_t rays_camera;       /* primary rays traced */
    uint64_t rays_shadow;       /* NEE closest-hit rays traced (pathTracing.cpp:54) */
    uint64_t rays_indirect;     /* valid extension rays traced (pathTracing.cpp:81, INVALID excluded) */
    uint64_t shaded_hits;       /* path vertices that reached shade() */
    uint64_t inner_visits[2];   /* [closest, shadow] inner nodes visited = all child boxes fetched and tested (TRT_FLAG_COUNT);
                                 * inner_node_bytes says how much one visit reads */
    uint64_t tri_tests[2];      /* [closest, shadow] triangle tests (TRT_FLAG_COUNT) */
    uint64_t wave_steps[2];     /* wave-level iterations of the traversal kernels' [inner-node, leaf] phases (TRT_FLAG_COUNT):
                                 * inner_visits / (64 * wave_steps[0]) is the SIMD utilisation of the inner phase */
    uint64_t launches[TRT_MAX_KERNELS];
    double kernel_ms[TRT_MAX_KERNELS]; /* summed launch durations (TRT_FLAG_TIMING) */
    double render_ms;           /* first gen_primary launch -> last resolve, device time */
    uint32_t passes;            /* sample chunks the render was split into */
    uint32_t max_bounces;       /* deepest path vertex index reached */
    uint64_t rows_rendered;     /* rows in the packed output */
    uint32_t inner_node_bytes;  /* bytes one inner-node visit fetches: 64 (the caller's BVH2 node: two boxes + refs, tiny scenes),
                                 * 128 (its exact 4-wide collapse: four boxes + refs) or 80 (its 8-wide collapse with quantised boxes) */
    uint32_t redo_rays;         /* rays whose traversal result failed the check made when it is stored (a hit in front of the box of its
                                 * own leaf, or — 8-wide nodes — on a leaf whose exact box the ray misses), and rays with a direction component
                                 * of exactly zero (for them the reference's slab test can meet 0 * inf; they are traced with its literal form where
                                 * their origin may lie on a box plane), traced again in the exact form: a handful per 10^7, respectively one or two
                                 * per 10^5, on padded trees; a large number says the slow path is carrying the render */
    uint64_t lane_census[4];    /* TRT_FLAG_COUNT, persistent traversal kernels: summed over every wave iteration, the lanes waiting for a node
                                 * step [0], for a triangle step [1], holding a finished ray that waits for the refill batch [2]; [3] = the
                                 * iterations (x 64 = lane slots; what is left held no ray) — where the idle SIMD lanes are */
} trt_stats;

typedef struct trt_handle trt_handle;

/* Number of image rows a trt_params selects (tile rows passing the interleave). */
int trt_rows_selected(const trt_params* p);

/* Upload the flat scene to HBM on `device` (HIP ordinal) and build the
 * device-side 48-B Moller-Trumbore triangle records and 64-B shading records.
 * Replaces nothing the reference times; it is the hand-over of scene.triangles /
 * root to the loop at main.cpp:76-81. */
int trt_create(const trt_scene* scene, int device, trt_handle** out);

/* render(): replaces main.cpp:79-113.  Renders p->spp samples of every
 * selected pixel and writes the averaged linear radiance, RGB interleaved,
 * row-major, rows packed (see row_block), as float into a HOST buffer of
 * rows_selected * (x1-x0) * 3 floats.  `stats` may be NULL. */
int trt_render(trt_handle* h, const trt_params* p, float* out_rgb_host, trt_stats* stats);

/* Same, but out_rgb is DEVICE memory on the handle's device and all work is
 * enqueued on `hip_stream` (a hipStream_t, or NULL for the default stream);
 * returns after the stream has been synchronised. */
int trt_render_device(trt_handle* h, const trt_params* p, float* out_rgb_dev,
                      void* hip_stream, trt_stats* stats);

/* Progressive / resumable render.  The reference keeps one `double* image` for the whole run and adds
 * color/SAMPLE of every sample to it (main.cpp:74-75,101), then shows it once at the end (main.cpp:114).
 * This entry renders samples [sample_begin, sample_end) of the p->spp samples and adds them, in sample
 * order, onto `accum` — HOST, rows_selected * (x1-x0) * 3 doubles, in/out: the running sums of
 * (double)(L_s / spp).  Calls that cover [0, spp) in increasing order, starting from zeros, leave exactly
 * the sums of one trt_render call (and in out_rgb_host — optional, may be NULL — exactly its image), so
 * a long render can be shown while it converges, check-pointed (save accum + sample_end) and resumed. */
int trt_render_samples(trt_handle* h, const trt_params* p, int32_t sample_begin, int32_t sample_end,
                       double* accum_host, float* out_rgb_host, trt_stats* stats);

/* traverseBVH (bvh.cpp:146-175) on a batch of n rays given as HOST arrays
 * org[n][3], dir[n][3].  Outputs (host): t[n] (TRT_INF on miss), tri[n]
 * (post-BVH triangle index, -1 on miss), uv[n][2] (barycentrics of v1,v2).
 * `stats` (optional) receives inner_visits[0]/tri_tests[0] and kernel_ms.
 * One rule beyond bvh.cpp's text ("Formulation"): a triangle hit whose distance lies IN FRONT of the box of the
 * leaf the triangle sits in — by more than a tolerance of 2^-16 relative plus 2^-17 of the scene's largest coordinate — does
 * not count: for a ray within ~1e-4 rad of a triangle's plane the computed distance can come out there; the reference rejects
 * such hits through its inside test on the computed point (bvh.cpp:191-198).  The tolerance keeps the rule off honest hits:
 * leaf boxes need NOT be padded (a triangle lying on a face of its leaf's box is found), at any coordinate magnitude. */
int trt_trace_closest(trt_handle* h, uint64_t n, const float* org, const float* dir,
                      float* t, int32_t* tri, float* uv, trt_stats* stats);

void trt_destroy(trt_handle* h);

/* ---- one node, several GPUs -------------------------------------------------------------------------------------
 * The sample/pixel loop has no cross-pixel dependency (main.cpp:84-108), so the image is tiled: the scene is replicated
 * on every device of the group, the rows of the tile are dealt to the devices in interleaved stripes of `row_block` rows
 * (device k renders the rows y with (y / row_block) % n == k), every device renders its stripes on its own host thread
 * (trt_render_device), and ONE ncclGather (RCCL over xGMI; /opt/rocm/include/rccl/rccl.h:745, communicators from
 * ncclCommInitAll :236) brings the packed stripes to devices[0], which un-interleaves them.  The random streams are keyed
 * by the global (pixel, sample), so the image is bit-identical to trt_render's for every n and every row_block.
 * RCCL is loaded (dlopen librccl.so.1) by trt_group_create only when the group spans more than one DISTINCT device;
 * a group whose entries name the same device several times (a rehearsal on a one-GPU box) gathers with device copies.
 * TRT_GROUP_FORCE_RCCL=1 in the environment at trt_group_create (a test switch) sends a group of ONE device through the
 * RCCL route as well — communicator of size 1, ncclGather to itself — so that the dlopen, the symbol bindings, the data type
 * constant and the stream ordering run on a one-GPU box.  Every device has a host thread of its own for the group's life. */
typedef struct trt_group trt_group;
int trt_group_create(const trt_scene* scene, int n_devices, const int* devices, trt_group** out);
/* p: as for trt_render (tile, spp, seed, flags); p->row_block (>= 1; 0 = 8) is the stripe height, row_mod / row_rem are
 * set by the library; the tile's first row p->y0 must be a multiple of row_block * group size (TRT_EINVAL otherwise: stripes are
 * counted from image row 0).  out_rgb_host: (y1-y0) * (x1-x0) * 3 floats.  stats (optional): rays / launches / kernel_ms summed
 * over the devices, render_ms = the slowest device's, plus gather_ms = gather + un-interleave on devices[0]. */
int trt_group_render(trt_group* g, const trt_params* p, float* out_rgb_host, trt_stats* stats, double* gather_ms);
/* The same with the image left in DEVICE memory of devices[0] (out_rgb_dev0: (y1-y0) * (x1-x0) * 3 floats there; nothing crosses
 * PCIe): what a caller that goes on working on the GPU links, and what bench.py --group times. */
int trt_group_render_device(trt_group* g, const trt_params* p, float* out_rgb_dev0, trt_stats* stats, double* gather_ms);
int trt_group_size(const trt_group* g);
void trt_group_destroy(trt_group* g);

const char* trt_last_error(void);

int trt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TRT_H */
