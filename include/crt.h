/*
 * include/crt.h -- C ABI of libcrt.so, the MI355X-native drop-in for the
 * reference's rendering hot path.
 *
 * The reference (guomc9/CudaRayTracing) has no FFI layer: the boundary of the
 * hot path is the C++ class `Render` (reference: include/Render.cuh:357-557)
 * constructed from a `Scene*` and driven by `run_view` (src/main.cu:282,372).
 * Every entry point below names the reference interface it replaces.  All
 * signatures are plain C (pointers, sizes, PODs); no torch / HIP types.
 * All functions return CRT_OK (0) or a negative crt_status; nothing prints and
 * continues (the reference printf's CUDA errors and carries on,
 * Render.cuh:393-397,441-473).
 *
 * Threading: one host thread per scene handle; handles are not shared across
 * threads (the reference is single-threaded and not re-entrant either).
 */
#ifndef CRT_H
#define CRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: CRT_TRAVERSAL_* renumbered (0 = EXACT, the default of a zeroed crt_params; FAST moved to 2), crt_intersect's `traversal`
 *    argument carries flag bits (CRT_INTERSECT_RAW_DIRECTIONS 0x100, _FORCE_EXACT 0x200, _VISIBILITY 0x400), progressive /
 *    preview / multi-device / accel-info entry points and structs added.  A client built against version 2 must be rebuilt:
 *    check crt_abi_version() == CRT_ABI_VERSION at load time. */
#define CRT_ABI_VERSION 5

typedef enum {
    CRT_OK = 0,
    CRT_ERR_INVALID_ARG = -1,
    CRT_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure at init (reference: src/main.cu:92-105) */
    CRT_ERR_HIP = -3,         /* a HIP call failed; see crt_last_error() */
    CRT_ERR_UNSUPPORTED = -4, /* valid for the reference but outside this build (e.g. a JPEG map_Kd texture) */
    CRT_ERR_IO = -5,          /* file missing / unreadable */
    CRT_ERR_PARSE = -6,       /* malformed OBJ / MTL / JSON */
    CRT_ERR_OOM = -7
} crt_status;

const char* crt_strerror(int status);
/* thread-local detail string of the last failure in this thread ("" if none) */
const char* crt_last_error(void);
int crt_abi_version(void);

/* ------------------------------------------------------------------------
 * Flat scene description handed to the device layer.
 * ---------------------------------------------------------------------- */

/* BVH node exactly as the reference uploads it (DeviceBVHNode,
 * include/DeviceBVH.cuh:9-15): post-order array, root = last; a node is a leaf
 * iff lc < 0 && rc < 0; leaf triangles are [it, it+n) of the BVH-ordered
 * triangle array. 40 bytes. */
typedef struct {
    int32_t lc, rc;
    uint32_t n;
    int32_t it;
    float aa[3];
    float bb[3];
} crt_bvh_node;

/* Triangle (reference DeviceTriangle, include/DeviceTriangle.cuh:12-37) with the
 * per-triangle material copy replaced by an index into crt_material[]. */
typedef struct {
    float v1[3], v2[3], v3[3];
    float normal[3];     /* geometric normal from winding (Triangle.h:27) */
    float area;          /* Triangle.h:39 */
    float area_of_obj;   /* Object.h:15-23 */
    int32_t material;
} crt_triangle;

/* reference DeviceMaterial (include/DeviceMaterial.cuh:5-37); ks/ka are never
 * read on the hot path and always zero (Loader.h:45,47,107). */
typedef struct {
    float kd[3];
    float ke[3];
    float ns;
    int32_t mode;      /* 0 = DIFFUSE, 1 = SPECULAR (Material.h:7-10) */
    int32_t has_emit;  /* Material.h:36-39 */
} crt_material;

/* One light object = a contiguous run of light_tris (reference DeviceLight,
 * include/DeviceLights.cuh:5-31: its own unsorted triangle copies). */
typedef struct {
    uint32_t first_tri;
    uint32_t count;
} crt_light;

typedef struct {
    const crt_bvh_node* nodes;     uint32_t n_nodes;  int32_t root;
    const crt_triangle* tris;      uint32_t n_tris;        /* BVH order */
    const crt_material* materials; uint32_t n_materials;
    const crt_triangle* light_tris; uint32_t n_light_tris;  /* shape order */
    const crt_light* lights;       uint32_t n_lights;
} crt_scene_desc;

typedef struct {
    float eye[3];
    float inv_view[9]; /* column-major 3x3 (Eigen::Matrix3f storage), from crt_inverse_view */
    float fov_y;       /* radians (src/main.cu:278) */
} crt_camera;

enum {
    CRT_TRAVERSAL_EXACT = 0,     /* THE DEFAULT (a zeroed crt_params): the 4-wide tree over the reference's leaves, nearest child first, any-hit
                                    visibility rays, zero-contribution samples answered without traversal -- every step of it provably
                                    result-neutral (docs/experiments.md 4.3), so the frame is REFERENCE's bit for bit (soak: 3.9e11 rays of
                                    full-size frames, 0 pixel slots differ) at 1/4 .. 1/6 of REFERENCE's time */
    CRT_TRAVERSAL_REFERENCE = 1, /* exhaustive traversal in the reference's visit order (DeviceBVH.cuh:128-170) */
    CRT_TRAVERSAL_FAST = 2       /* CRT_TRAVERSAL_EXACT plus distance pruning: skip a box entered beyond t_ref + 0.1 % + 1e-3 + 1e-4 x reach x steep
                                    (csrc/crt_trace.h; reach = max |origin coordinate| + t_ref, steep = max |1 / direction component|).  C2 -7 %,
                                    veach-mis -21 % frame time.  Exact unless a Moeller-Trumbore hit lies further in front of its own leaf box
                                    than that slack, which happens for rays lying in the plane of a triangle (det -> 0: unbounded error, and
                                    the reference has no determinant threshold).  MEASURED (tools/soak_fast_vs_reference.py,
                                    profiles/r02_soak_fast_vs_reference.jsonl): 2 rays in 3.66e11 on full 1920x1080x4096 veach-mis frames
                                    (tessellated spheres; each changes one next-event sample: the last bit of one pixel), 0 in 3.4e11 on full
                                    3840x2160x256 cornell-box frames; a larger slack only thins them out (docs/experiments.md 4.3).
                                    Bit-identical to REFERENCE on every frame and probe of the test-suite (two lost rays are kept as known
                                    answers in tests/test_adversarial_traversal.py); bench.py times it beside the default and re-checks a
                                    slice of the benchmark frame against REFERENCE in every run */
};
enum {
    CRT_FLAG_STATS = 1u,         /* fill the traversal counters of crt_stats (slower counting kernels) */
    CRT_FLAG_TILED_OUTPUT = 2u,  /* write this rank's pixels in compact 8x8-tile order instead of row-major */
    CRT_FLAG_TRACE_ALL = 8u,     /* CRT_TRAVERSAL_EXACT / _FAST trace every next-event sample, also those whose contribution is exactly
                                    zero (crt_stats.rays_untraced stays 0); same frame, for measuring the traversal alone */
    CRT_FLAG_FORCE_EXACT = 4u,   /* test hook: treat every ray of CRT_TRAVERSAL_FAST as one with non-finite operands (reference
                                    box arithmetic on the reference topology, still pruned / any-hit); results are unchanged */
    CRT_FLAG_BOUNDED_RADIANCE = 16u /* keep the radiance of a WINDOW of samples instead of one value per path: the frame's sum c += L_k / spp
                                    (Render.cuh:348) is made in sample order inside the launch ("commit ring", docs/experiments.md section 9), the
                                    whole sample range is one launch whatever its size, and the handle needs 16 B x pixels x 32 ... 64
                                    samples (245 MB for 800x600) instead of 16 B per path (3.9 GB for 800x600 spp 512; 17 GB per 2^30 paths).
                                    Same bits.  Costs time (800x600 spp 512: 167 ms instead of 93): off by default -- memory is what this
                                    device has plenty of.  Ignored with CRT_FLAG_STATS, by the fallback pipeline, and when the sample range
                                    is no longer than the window */
};

typedef struct {
    uint32_t width, height;      /* Scene::width/height (Scene.h:28-31) */
    uint32_t spp;                /* Render::spp (Render.cuh:362) */
    float p_rr;                  /* Render::P_RR */
    int32_t light_sample_n;      /* Render::light_sample_n */
    uint64_t seed;               /* replaces clock() (Render.cuh:341) */
    uint32_t rank, world;        /* pixel-tile shard: 8x8 tile t belongs to rank t % world; world >= 1 */
    uint32_t traversal;          /* CRT_TRAVERSAL_* */
    uint32_t flags;              /* CRT_FLAG_* */
} crt_params;

typedef struct {
    uint64_t paths;              /* W*H*spp of this shard */
    uint64_t rays;               /* closest-hit queries (= DeviceBVH::intersect calls) */
    uint64_t shadow_rays, probe_rays;
    uint64_t inner_pops, leaf_pops, tri_tests, hits; /* visit counters of the traversal that ran; CRT_FLAG_STATS only
                                                         (with CRT_TRAVERSAL_REFERENCE: the reference's visit set) */
    uint64_t stack_sum, stack_max;                   /* CRT_FLAG_STATS: sum / max over rays of the traversal stack high-water */
    uint64_t phase_cycles[24];                       /* reserved (zero): the per-phase cycle stamps of rounds 1-2 were replaced by the
                                                        basic-block profile of tools/bbprof */
    float kernel_ms;             /* sum of the HIP-event times of the render kernel's launches: k_mega3, one launch per chunk of at most
                                    2^30 work items (the wavefront fallback pipeline: the k_trace launches) */
    float logic_ms;              /* 0 for k_mega3 (the path logic is fused into it); the wavefront fallback: the k_logic launches */
    float total_ms;              /* HIP-event time of the whole device pipeline of this call (k_order_items, k_mega3, k_accumulate) */
    uint32_t kernel_launches;    /* number of k_mega3 (fallback: k_trace) launches */
    uint64_t rays_untraced;      /* of `shadow_rays`: next-event samples whose contribution is exactly zero (clamped cosine, black
                                    BSDF), answered without traversal by CRT_TRAVERSAL_EXACT / _FAST -- adding +0 cannot change L_dir;
                                    0 with CRT_FLAG_TRACE_ALL or CRT_TRAVERSAL_REFERENCE */
} crt_stats;

/* ------------------------------------------------------------------------
 * Device layer
 * ---------------------------------------------------------------------- */
typedef struct crt_scene crt_scene;

/* number of HIP devices visible (reference: config_CUDA, src/main.cu:92-105) */
int crt_device_count(int* count);

/* Upload a flat scene to `device` (replaces the DeviceBVH / DeviceLights /
 * DeviceTriangle / DeviceMaterial constructors, DeviceBVH.cuh:52-80,
 * DeviceLights.cuh:12-31,63-87, and the device half of Render's constructor,
 * Render.cuh:387-414).  The per-pixel stack allocations of Render.cuh:416-422
 * have no equivalent: traversal stacks live in LDS. */
int crt_scene_create(const crt_scene_desc* desc, int device, crt_scene** out);
/* How the acceleration trees of CRT_TRAVERSAL_FAST were built at crt_scene_create: a binned-SAH tree over the reference's leaves
 * (csrc/crt_accel.h; on the device by default, csrc/crt_accel_build.hip -- CRT_SAH_HOST=1 forces the host builder) collapsed to 4
 * children per node. */
typedef struct {
    uint32_t n_leaves, n_nodes2, n_nodes4, depth2, depth4;
    uint32_t sah_on_device;   /* 1: built by the device builder, 0: host builder (forced, or the device build could not run) */
    uint32_t index_splits;    /* ranges whose leaf centroids all coincide (duplicate leaves): split by index -- the only place where the
                                 device builder's tree may differ from the host builder's (equal leaves on different sides) */
    float sah_ms;             /* host clock: the whole SAH build (uploads and renumbering included) */
    float sah_device_ms;      /* HIP events around the level loop (0 for the host builder) */
    float runtime_init_ms;    /* host clock of crt_scene_create's first device calls (hipSetDevice, a 4-byte allocation and copy): the HIP
                                 runtime's one-off start in a process -- context creation, loading libcrt.so's code objects; about 150 ms
                                 for the first scene of a process, microseconds afterwards.  Not part of sah_ms. */
    uint32_t layout_caps;     /* which pool layouts of the render kernel the scene allows: bit 0 = node AND leaf refs fit 16-bit stack
                                 entries (coupled form, 8 LDS levels), bit 1 = the four-wide nodes alone do (decoupled leaves, 6 levels:
                                 what a scene of roughly 50 000 - 160 000 triangles renders with), bit 2 = leaf records fit a leaf-queue
                                 entry (decoupled leaves possible at all), bit 3 = the copy of the four-wide tree without its rows of refs
                                 exists (at most 32 768 nodes, leaves of one record -- bvh_thresh_n <= 2: six loads per inner visit instead of seven) */
} crt_accel_info;
int crt_scene_accel_info(crt_scene* scene, crt_accel_info* out);
/* replaces Render::free (Render.cuh:477-487) */
int crt_scene_destroy(crt_scene* scene);

/* Number of pixel slots a shard writes with CRT_FLAG_TILED_OUTPUT:
 * ceil(n_tiles_of_rank) * 64, n_tiles = ceil(W/8)*ceil(H/8). */
int crt_shard_slots(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, uint64_t* slots);

/* Render one frame and copy it to host memory: replaces Render::run_view
 * (Render.cuh:435-475: kernel launch, synchronize, D2H copy; the OpenGL PBO
 * copy is dropped).  out_rgb: 3 bytes per pixel, row-major, row 0 = image top
 * (W*H*3 bytes, or slots*3 with CRT_FLAG_TILED_OUTPUT).  out_mean (optional,
 * may be NULL): pre-tone-map mean radiance, 3 floats per pixel, same order.
 * stats optional. */
int crt_render(crt_scene* scene, const crt_camera* cam, const crt_params* params, uint8_t* out_rgb,
               float* out_mean, crt_stats* stats);

/* Same, but outputs stay in device memory (d_rgb / d_mean are device pointers
 * on the scene's device, d_mean may be NULL) and the work is enqueued on
 * `hip_stream` (a hipStream_t, NULL = default stream) without synchronizing.
 * If stats != NULL the call synchronizes the stream to read counters/timers. */
int crt_render_device(crt_scene* scene, const crt_camera* cam, const crt_params* params, void* d_rgb,
                      void* d_mean, void* hip_stream, crt_stats* stats);

/* Device time of the render kernel launches of the LAST frame submitted on the handle (first launch's start to last launch's end, HIP
 * events recorded on the frame's stream without synchronizing): lets a caller that pipelines frames with crt_render_device(stats = NULL)
 * read every frame's kernel time afterwards.  The frame's stream must have been synchronized (CRT_ERR_HIP otherwise). */
int crt_last_launch_ms(crt_scene* scene, float* ms, uint32_t* launches);

/* Bytes of per-path radiance storage the handle's last render used, and the size of its commit ring in samples (0 = one radiance per
 * path of a chunk; see CRT_FLAG_BOUNDED_RADIANCE). */
int crt_radiance_storage(crt_scene* scene, uint64_t* bytes, uint32_t* ring_samples);

/* Progressive rendering (the reference re-renders all spp on every click, src/main.cu:368-377):
 * renders samples [sample_begin, sample_begin + sample_count) of params->spp into the accumulator the scene handle
 * owns (temp_color += L_k / spp, in sample order as Render.cuh:348).  Ranges must be submitted in ascending order
 * starting at 0, with the same camera / params, and no other render call on the handle in between; the range that
 * ends at spp tone-maps and writes out_rgb / out_mean -- bit-identical to one crt_render call.  For the other ranges
 * out_rgb / out_mean are not written and may be NULL. */
int crt_render_range(crt_scene* scene, const crt_camera* cam, const crt_params* params, uint32_t sample_begin,
                     uint32_t sample_count, uint8_t* out_rgb, float* out_mean, crt_stats* stats);
int crt_render_range_device(crt_scene* scene, const crt_camera* cam, const crt_params* params, uint32_t sample_begin,
                            uint32_t sample_count, void* d_rgb, void* d_mean, void* hip_stream, crt_stats* stats);

/* The displayable frame of a progressive render in flight -- the viewer half of SURVEY 8(f) row 4 (the reference shows nothing
 * until all spp are done and re-renders from scratch on every click, src/main.cu:368-377).  After a range that ends at
 * `done` < spp the accumulator holds sum_{k < done} L_k / spp; the preview is tone-map(accumulator * spp / done), written in the
 * layout of the range calls (row-major, or compact tiles with CRT_FLAG_TILED_OUTPUT).  It only READS the accumulator: the bits
 * of the final frame do not depend on whether, or how often, previews were taken.  *samples_done (optional) receives `done`.
 * CRT_ERR_INVALID_ARG when no progressive render is in flight (before the first range, after the one that ends at spp). */
int crt_preview(crt_scene* scene, uint8_t* out_rgb, float* out_mean, uint32_t* samples_done);
int crt_preview_device(crt_scene* scene, void* d_rgb, void* d_mean, void* hip_stream, uint32_t* samples_done);

/* ------------------------------------------------------------------------
 * Multi-device rendering in ONE process.  The reference picks device 0 and stops there
 * (config_CUDA, src/main.cu:92-105); a crt_multi holds one device replica of the scene per entry of
 * `devices` (= rank), renders the interleaved 8x8-tile shards concurrently (one host thread and one HIP
 * stream per device), exchanges the compact tile buffers with ONE ncclAllGather over RCCL / xGMI and
 * de-interleaves them into the row-major frame on rank 0 -- the frame is identical to crt_render's on
 * one device, whatever the number of ranks.
 * ---------------------------------------------------------------------- */
typedef struct crt_multi crt_multi;
enum {
    CRT_GATHER_AUTO = 0,  /* RCCL when there are two or more DISTINCT devices, peer copies otherwise */
    CRT_GATHER_RCCL = 1,  /* ncclCommInitAll + one ncclAllGather per frame (librccl.so.1 is bound at run time; its absence is CRT_ERR_UNSUPPORTED) */
    CRT_GATHER_COPY = 2   /* every rank copies its block into rank 0's buffer (hipMemcpyPeerAsync); also the only mode that accepts
                             several ranks on ONE device (test configuration: RCCL refuses duplicate devices) */
};
typedef struct {
    uint32_t n_ranks;        /* device replicas that rendered */
    uint32_t gather;         /* CRT_GATHER_* that ran */
    uint32_t rccl_ranks;     /* ncclCommCount of the communicator the gather ran on; 0 with CRT_GATHER_COPY */
    int32_t rccl_version;    /* ncclGetVersion; 0 if RCCL was not used */
    float render_ms;         /* host clock: until the slowest rank's shard is complete */
    float gather_ms;         /* host clock: exchange + de-interleave + copy to the host */
    float frame_ms;          /* host clock: the whole call */
    float max_kernel_ms;     /* largest crt_stats.kernel_ms over the ranks */
    uint64_t bytes_per_rank; /* size of one rank's block in the exchange */
    uint64_t rays, paths, rays_untraced; /* sums over the ranks */
    char fallback_reason[160]; /* (ABI 5) CRT_GATHER_AUTO only: why the gather runs on peer copies although the devices are distinct -- librccl.so.1 not
                                  found, ncclCommInitAll refused ...; empty when no fallback happened.  An explicit CRT_GATHER_RCCL fails instead. */
} crt_multi_info;
/* gather: CRT_GATHER_*.  Uploads the scene to every device (crt_scene_create per rank). */
int crt_multi_create(const crt_scene_desc* desc, const int* devices, uint32_t n_devices, uint32_t gather, crt_multi** out);
int crt_multi_destroy(crt_multi* multi);
/* = Render::run_view over all ranks.  params->rank / world are ignored (rank r renders tiles t % n == r).  out_rgb: W*H*3 bytes,
 * row-major, row 0 = image top, may be NULL (the frame then stays on rank 0's device, crt_multi_frame_device); out_mean optional
 * (needs out_rgb); stats: NULL or n_devices entries, one per rank; info optional. */
int crt_multi_render(crt_multi* multi, const crt_camera* cam, const crt_params* params, uint8_t* out_rgb, float* out_mean,
                     crt_stats* stats, crt_multi_info* info);
/* device pointers of the last frame on rank 0's device (d_mean: NULL unless the last call asked for the mean) */
int crt_multi_frame_device(crt_multi* multi, void** d_rgb, void** d_mean, int* device);

/* Closest-hit query for n rays (device-side DeviceBVH::intersect,
 * DeviceBVH.cuh:128-170), host buffers. dirs are normalised as Ray's
 * constructor does (Ray.cuh:12-15). out_tri: BVH-order triangle index or -1. */
#define CRT_INTERSECT_RAW_DIRECTIONS 0x100u /* OR into `traversal`: take dirs as they are (already a Ray's direction), do not normalise again */
#define CRT_INTERSECT_FORCE_EXACT 0x200u     /* OR into `traversal`: as CRT_FLAG_FORCE_EXACT for the queries (test hook) */
/* OR into `traversal`: the rays are visibility rays -- blocked() of Render.cuh:19-27.  out_t[i] holds t_to_light on entry; on return
 * out_t[i] = 1.0f if the ray is blocked (t_to_light - closest.t > EPSILON) else 0.0f and out_tri[i] = a blocking triangle or -1
 * (REFERENCE: the closest hit; FAST: the first one the any-hit traversal met). */
#define CRT_INTERSECT_VISIBILITY 0x400u
int crt_intersect(crt_scene* scene, uint32_t n, const float* origins, const float* dirs, uint32_t traversal,
                  int32_t* out_tri, float* out_t);

/* Device-side evaluation of the deterministic math / RNG helpers, for parity
 * tests against the oracle.  fn in {"sin","cos","tan","acos","atan2","exp","log10","pow","uniform"}. */
int crt_device_math(int device, const char* fn, uint32_t n, const float* a, const float* b, float* out);
int crt_device_philox(int device, uint32_t n, const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4);
/* Exhaustive self-check of the short reciprocal the kernels use in place of the division 1.0f / x (Ray.cuh:14,
 * DeviceTriangle.cuh:47): evaluates both for all 2^32 bit patterns of x on the device and returns, in *mismatches, the
 * number of inputs INSIDE the guarded range (2^-126 <= |x| < 2^126) whose bits differ (must be 0), and in *outside the
 * number of inputs outside the range that differ (those take the division itself).  A few milliseconds. */
int crt_device_rcp_check(int device, uint64_t* mismatches, uint64_t* outside);

/* ------------------------------------------------------------------------
 * Host layer: scene ingestion and BVH build on the CPU (north star: "C++ host
 * code builds the BVH and triangle/material/light arrays as today").
 * Mirrors Scene / Loader / Object / BVH / Camera of the reference.
 * ---------------------------------------------------------------------- */
typedef struct crt_host_scene crt_host_scene;

/* Scene(width, height)  (Scene.h:28-31) */
int crt_host_scene_create(uint32_t width, uint32_t height, crt_host_scene** out);
int crt_host_scene_destroy(crt_host_scene* scene); /* Scene::free */
/* Loader::read_OBJ + load_object for every shape + Scene::add_normal_obj/add_light_obj
 * in shape order (src/main.cu:122-145) */
int crt_host_scene_add_obj(crt_host_scene* scene, const char* obj_path, const char* mtl_dir);
/* Scene::set_BVH(thresh_n) (Scene.h:50-54 -> BVH.h:30-84) */
int crt_host_scene_set_bvh(crt_host_scene* scene, uint32_t thresh_n);
/* The same BVH built on the GPU (csrc/crt_bvh_build.hip): level-synchronous median split; per level the
 * device replays the quicksort phase of libstdc++'s std::sort on every range (equal centroid coordinates are the norm on real
 * meshes, and an unstable sort's order of equal keys is its own) and finishes with one stable radix sort of the whole triangle
 * order.  Node and triangle arrays are BYTE-IDENTICAL to crt_host_scene_set_bvh's.  A range whose quicksort phase hits std::sort's
 * depth limit is sorted by the host between two levels (info->host_sorts); scenes with -0.0 or non-finite coordinates are built on
 * the host entirely (info->host_triangles = all). */
typedef struct {
    uint32_t n_triangles, n_nodes, levels;
    uint32_t host_ranges;      /* subtrees finished by the host builder (none unless a range's extents are NaN) */
    uint32_t host_triangles;   /* triangles in them */
    uint32_t host_sorts;       /* single range sorts done by the host's std::sort between two levels: ranges whose quicksort phase ran into
                                  std::sort's depth limit (heapsort), e.g. already-sorted keys full of ties */
    uint64_t host_sort_elements;
    float device_ms;           /* HIP events around the level loop */
    float total_ms;            /* host clock: uploads, level loop, downloads, host subtrees */
    float host_build_ms;       /* host clock: the host builder's part (host ranges, or everything on a fallback) */
} crt_bvh_build_info;
int crt_host_scene_set_bvh_device(crt_host_scene* scene, uint32_t thresh_n, int device, crt_bvh_build_info* info);
/* Flat view of the built scene; pointers stay valid until the host scene is
 * destroyed or modified. */
int crt_host_scene_desc(const crt_host_scene* scene, crt_scene_desc* out);
int crt_host_scene_num_objects(const crt_host_scene* scene, uint32_t* n);
/* Object area as printed by the reference (Object.h:25) and whether it is a light */
int crt_host_scene_object(const crt_host_scene* scene, uint32_t index, float* area, int32_t* is_light,
                          uint32_t* n_tris);

/* get_inverse_view_matrix (Camera.h:9-36); out = 9 floats column-major */
int crt_inverse_view(const float eye[3], const float lookat[3], const float up[3], float out[9]);

/* Task / config.json (src/main.cu:40-90) */
typedef struct {
    uint32_t n_objs;           /* entries of OBJ_paths in the file -- any number, as src/main.cu:74-78 loops over them.  The two arrays of this struct
                                  hold the first 8 only: a client that walks obj_path[i] / mtl_dir[i] must stop at min(n_objs, 8) and take the rest
                                  from crt_task_obj */
    char obj_path[8][512];     /* the first eight; crt_task_obj() returns any of them */
    char mtl_dir[8][512];
    float lookat[3], up[3], eye_pos[3];
    float fov_y;        /* degrees, as in the file */
    uint32_t width, height, bvh_thresh_n, light_sample_n, spp;
    float p_rr;
} crt_task;
int crt_task_load(const char* config_json_path, crt_task* out);
/* Entry `index` (< n_objs) of the file's OBJ_paths: the two strings, NUL-terminated, into buffers of `cap` bytes each
 * (CRT_ERR_INVALID_ARG if one does not fit or the index is out of range).  For configurations of more than eight OBJ files. */
int crt_task_obj(const char* config_json_path, uint32_t index, char* obj_path, char* mtl_dir, uint32_t cap);

/* What the reference's texture decoder returns for a map_Kd file -- stbi_load(path, &x, &y, &comp, 0) of Loader.h:58: 8-bit
 * samples, row 0 = top, the file's own channel count.  Decodes every format that decoder reads -- PNG (plain and Adam7), JPEG
 * (baseline and progressive), BMP, TGA, GIF (first frame), PSD, Softimage PIC, binary PNM and Radiance HDR (csrc/crt_image.h,
 * crt_png.h, crt_jpeg.h, crt_formats.h) -- with that decoder's own conventions, pinned sample for sample against the reference's
 * vendored stb_image by tests/golden/stb_decode.json.  A file that is none of these, or damaged beyond what the reference's
 * decoder accepts, returns CRT_ERR_UNSUPPORTED.  out may be NULL (size query); cap = bytes available at out (x * y * comp needed). */
int crt_image_load(const char* path, int32_t* x, int32_t* y, int32_t* comp, uint8_t* out, uint64_t cap);

/* stb-free PNG writer used by Render::save_frame_buffer's replacement (Render.cuh:489-493) */
int crt_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb);

#ifdef __cplusplus
}
#endif
#endif /* CRT_H */
