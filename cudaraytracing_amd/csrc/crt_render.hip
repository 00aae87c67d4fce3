// cudaraytracing_amd/csrc/crt_render.hip -- host side of the device layer of libcrt.so: scene upload (the flat HBM layouts of crt_device.h, the two trees,
// the 4-wide collapse), the launch logic of a frame, and the C ABI of include/crt.h (crt_scene_create, crt_render*, crt_preview*,
// crt_intersect, crt_device_*).  The kernels live in crt_mega3.hip, crt_wavefront.hip, crt_frame.hip.
#include "crt_internal.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <type_traits>
#include <array>
#include <atomic>
#include <vector>

using namespace crtdev;
using namespace crtk;

struct crt_scene {
    int device = 0;
    DevBuf<float4> nodes, tri_geo, mats, ltri, nodes3, leaf_geo, tri_nm, nodes4, nodes4i, leaf_geo_i;
    DevBuf<int32_t> rec_map;
    bool impl_ok = false;        // the implicit-refs copy of the 4-wide tree exists (nodes4i: leaves of one record, <= 32 768 nodes)
    int depth4 = 1; // depth of the 4-wide tree
    bool ref16_ok = false; // refs of the 4-wide tree and of the leaf records fit 16 bits (k_mega3's 16-bit stack layout)
    bool ref16_inner_ok = false; // refs of the 4-wide tree alone fit 16 bits (decoupled leaves: the stack holds inner nodes only)
    bool dec_ok = false;         // leaf records fit the 24 bits of a leaf-queue entry
    uint32_t max_leaf = 0; // triangles in the largest leaf
    DevBuf<int32_t> tri_mat, leaf_count;
    DevBuf<uint4> lights;
    // path pool + per-item radiance + cross-chunk accumulator
    DevBuf<float4> p_ro, p_rd, p_vx, p_la, p_cc, p_vn, p_rec_a, p_rec_b, L;
    DevBuf<uint4> p_id;
    uint32_t n_mats = 0;
    DevBuf<float2> p_res;
    DevBuf<float> accum;
    DevBuf<unsigned long long> counters;      // [CNT_SHARDS][CNT_STRIDE]
    DevBuf<unsigned int> item_next;           // [ITEM_SHARDS][ITEM_STRIDE]
    DevBuf<uint32_t> item_list;               // k_order_items: the order of the work items of a launch (small launches only)
    DevBuf<unsigned int> ring_done, ring_state; // commit ring: finished items per (shard, sample), shard words
    DevBuf<float4> ring_L;                      // commit ring: radiance of [ring samples][shards * slots per shard] (uncached memory)
    std::vector<unsigned int> ring_state_host;
    uint64_t last_radiance_bytes = 0;           // per-work-item (or ring) radiance storage the last render used
    uint32_t last_ring_samples = 0;             // its ring size in samples (0: one radiance per work item)
    DevBuf<unsigned int> order_cnt;           // [ITEM_SHARDS][2] counters, one 128 B line each
    DevBuf<unsigned int> slot_next[2];        // [SLOT_SHARDS][SLOT_STRIDE], one per pool half
    DevBuf<int2> spill[2];                    // traversal stack overflow, one per pool half
    hipStream_t aux_stream = nullptr;         // second pool half runs here so that k_logic overlaps k_trace
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr; // around the k_mega3 launches of the last frame, recorded without synchronizing (crt_last_launch_ms)
    uint32_t last_launches = 0;
    int n_cus = 0;
    unsigned long long* h_counters = nullptr; // pinned copy of counters
    DevScene dev{};
    int stack_cap = 0;
    uint32_t n_tris = 0;
    crt_accel_info accel{};
    // progressive render in flight: what the accumulator holds (crt_preview)
    struct { uint32_t samples = 0, spp = 0, width = 0, height = 0, rank = 0, world = 1, tiled = 0; } acc;
    std::vector<hipEvent_t> ev;
    ~crt_scene()
    {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        if (ev_k0) (void)hipEventDestroy(ev_k0);
        if (ev_k1) (void)hipEventDestroy(ev_k1);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (aux_stream) (void)hipStreamDestroy(aux_stream);
        if (h_counters) (void)hipHostFree(h_counters);
    }
};

// 16-bit stack entries: the scene allows it (crt_scene::ref16_ok) and CRT_REF16=0 does not forbid it
static bool use_ref16(const crt_scene* sc, int mode, bool dec = false)
{
    if (mode == 1 || !(dec ? sc->ref16_inner_ok : sc->ref16_ok)) return false;
    // CRT_REF16=0 ("the leaf refs do not fit 16 bits"): 32-bit entries for the coupled form -- the decoupled form keeps its 16-bit
    // entries; CRT_REF32=1: 32-bit stack entries in either form (tests, A/B)
    const char* e = std::getenv("CRT_REF16");
    const char* f = std::getenv("CRT_REF32");
    if (f && f[0] == '1') return false;
    return dec || !(e && e[0] == '0');
}

namespace {

int fail(int status, const std::string& msg)
{
    crt_set_last_error_(msg.c_str());
    return status;
}
int fail_hip(const HipFail& f)
{
    return fail(CRT_ERR_HIP, std::string(f.what) + ": " + hipGetErrorString(f.e));
}

float as_float(int32_t v) { float f; std::memcpy(&f, &v, 4); return f; }
int32_t as_int(float f) { int32_t v; std::memcpy(&v, &f, 4); return v; }

// Builds the device node array (layout: crt_device.h) holding TWO trees over the same leaves:
//   [0, A)      the SAH tree of crt_accel.h, used by CRT_TRAVERSAL_FAST for rays with finite inv_dir
//   [A, A + R)  the reference's own topology (post-order BVH re-laid breadth-first), used by
//               CRT_TRAVERSAL_REFERENCE and by FAST rays whose inv_dir is not finite.
// Returns the larger tree depth (root = 1).
struct AccelInfo {
    uint32_t n_leaves = 0, n_nodes2 = 0, on_device = 0, index_splits = 0;
    float sah_ms = 0.0f, sah_device_ms = 0.0f;
};
int convert_bvh(const crt_scene_desc& d, std::vector<float4>& nodes, std::vector<int32_t>& leaf_count, int32_t& root_fast, int32_t& root_exact, AccelInfo* ai = nullptr)
{
    auto is_leaf = [&](int32_t i) { return d.nodes[i].lc < 0 && d.nodes[i].rc < 0; };
    auto leaf_ref = [&](int32_t i) -> int32_t {
        const crt_bvh_node& n = d.nodes[i];
        uint32_t cnt = n.n <= 15 ? n.n : 0;
        if (n.n > 15) leaf_count[n.it] = (int32_t)n.n;
        return (int32_t)~(((uint32_t)n.it << 4) | cnt);
    };
    leaf_count.assign(d.n_tris ? d.n_tris : 1, 0);
    nodes.clear();
    if (is_leaf(d.root)) { root_fast = root_exact = leaf_ref(d.root); return 1; }

    // ---- SAH tree over the reference leaves ----
    std::vector<crtaccel::Prim> prims;
    for (uint32_t i = 0; i < d.n_nodes; i++) {
        if (!is_leaf((int32_t)i)) continue;
        crtaccel::Prim p;
        for (int a = 0; a < 3; a++) { p.box.lo[a] = d.nodes[i].aa[a]; p.box.hi[a] = d.nodes[i].bb[a]; }
        p.ref = leaf_ref((int32_t)i);
        prims.push_back(p);
    }
    std::vector<crtaccel::Node> acc;
    int32_t acc_root = 0;
    // the SAH tree over the reference leaves: on the device (crt_accel_build.hip; CRT_SAH_HOST=1 forces the host builder, which is
    // also the fallback); the 4-wide collapse below stays on the host (linear, a few hundred microseconds)
    const auto sah_t0 = std::chrono::steady_clock::now();
    int depth_fast = -1;
    float dev_ms = 0.0f;
    const bool want_device = !(std::getenv("CRT_SAH_HOST") && std::getenv("CRT_SAH_HOST")[0] == '1');
    uint32_t index_splits = 0;
    if (want_device) depth_fast = crtaccel::build_sah_device(prims, acc, acc_root, &dev_ms, &index_splits);
    const bool on_device = depth_fast >= 0;
    if (!on_device) depth_fast = crtaccel::build_sah(prims, acc, acc_root, &index_splits);
    // one pass of insertion-based optimisation over the built tree (crt_accel.h: optimize_sah; only moves that save half of what the
    // node costs where it is); CRT_SAH_OPT=<passes> / CRT_SAH_OPT_MARGIN=<fraction> override, CRT_SAH_OPT=0 leaves the tree as built
    {
        const char* opt_ = std::getenv("CRT_SAH_OPT");
        const int passes = opt_ ? std::atoi(opt_) : 1;
        const char* form_ = std::getenv("CRT_SAH_OPT_FORM"); // (A/B hook: "serial" = the pass of rounds 4 / 5, every search on the tree as the last move left it)
        if (!acc.empty() && passes > 0) depth_fast = (form_ && form_[0] == 's') ? crtaccel::optimize_sah_serial(acc, passes) : crtaccel::optimize_sah(acc, passes);
    }
    if (ai) {
        ai->n_leaves = (uint32_t)prims.size(); ai->n_nodes2 = (uint32_t)acc.size(); ai->on_device = on_device ? 1u : 0u;
        ai->sah_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - sah_t0).count();
        ai->sah_device_ms = dev_ms;
        ai->index_splits = index_splits;
    }
    const int32_t A = (int32_t)acc.size();
    nodes.resize((size_t)A * 4);
    for (int32_t q = 0; q < A; q++) {
        const crtaccel::Node& n = acc[q];
        nodes[q * 4 + 0] = make_float4(n.box[0].lo[0], n.box[0].lo[1], n.box[0].lo[2], as_float(n.child[0]));
        nodes[q * 4 + 1] = make_float4(n.box[0].hi[0], n.box[0].hi[1], n.box[0].hi[2], as_float(n.child[1]));
        nodes[q * 4 + 2] = make_float4(n.box[1].lo[0], n.box[1].lo[1], n.box[1].lo[2], 0.0f);
        nodes[q * 4 + 3] = make_float4(n.box[1].hi[0], n.box[1].hi[1], n.box[1].hi[2], 0.0f);
    }
    root_fast = acc_root; // 0 (two or more leaves here)

    // ---- the reference topology, breadth-first numbering of inner nodes ----
    std::vector<int32_t> order, index(d.n_nodes, -1), depth_of;
    order.push_back(d.root);
    depth_of.push_back(1);
    index[d.root] = 0;
    int max_depth = 1;
    for (size_t q = 0; q < order.size(); q++) {
        const crt_bvh_node& n = d.nodes[order[q]];
        int32_t ch[2] = {n.lc, n.rc};
        for (int c = 0; c < 2; c++) {
            max_depth = std::max(max_depth, depth_of[q] + 1);
            if (!is_leaf(ch[c])) {
                index[ch[c]] = (int32_t)order.size();
                order.push_back(ch[c]);
                depth_of.push_back(depth_of[q] + 1);
            }
        }
    }
    nodes.resize(((size_t)A + order.size()) * 4);
    for (size_t q = 0; q < order.size(); q++) {
        const crt_bvh_node& n = d.nodes[order[q]];
        const crt_bvh_node& l = d.nodes[n.lc];
        const crt_bvh_node& r = d.nodes[n.rc];
        int32_t lref = is_leaf(n.lc) ? leaf_ref(n.lc) : A + index[n.lc];
        int32_t rref = is_leaf(n.rc) ? leaf_ref(n.rc) : A + index[n.rc];
        const size_t o = ((size_t)A + q) * 4;
        nodes[o + 0] = make_float4(l.aa[0], l.aa[1], l.aa[2], as_float(lref));
        nodes[o + 1] = make_float4(l.bb[0], l.bb[1], l.bb[2], as_float(rref));
        nodes[o + 2] = make_float4(r.aa[0], r.aa[1], r.aa[2], 0.0f);
        nodes[o + 3] = make_float4(r.bb[0], r.bb[1], r.bb[2], 0.0f);
    }
    root_exact = A;
    return std::max(max_depth, depth_fast);
}

int validate_desc(const crt_scene_desc* d)
{
    if (!d || !d->nodes || !d->tris || !d->materials || d->n_nodes == 0 || d->n_tris == 0 || d->n_materials == 0)
        return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: empty scene description");
    if (d->root < 0 || (uint32_t)d->root >= d->n_nodes) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: root index out of range");
    if (d->n_tris >= (1u << 27)) return fail(CRT_ERR_UNSUPPORTED, "crt_scene_create: more than 2^27 triangles");
    if (d->n_lights && (!d->lights || !d->light_tris)) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: lights without triangles");
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        const crt_bvh_node& n = d->nodes[i];
        bool leaf = n.lc < 0 && n.rc < 0;
        if (leaf) {
            if (n.it < 0 || n.n == 0 || (uint64_t)n.it + n.n > d->n_tris) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: leaf range outside the triangle array");
        } else {
            // inner nodes of the reference builder always have two children (BVH.h:79-81) that precede them (post-order)
            if (n.lc < 0 || n.rc < 0 || (uint32_t)n.lc >= i || (uint32_t)n.rc >= i) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: inner node children must precede it (post-order)");
        }
    }
    {
        // every node must hang under the root exactly once (the reference builder emits a tree in post-order, BVH.h:37-84), and the
        // leaves must own disjoint triangle ranges: the FAST traversal builds its own tree over ALL leaves of the description, so a
        // leaf the reference topology cannot reach, or two leaves sharing triangles, would make the two modes disagree
        std::vector<uint8_t> seen(d->n_nodes, 0), owned(d->n_tris, 0);
        std::vector<int32_t> todo(1, d->root);
        uint32_t visited = 0;
        while (!todo.empty()) {
            const int32_t i = todo.back();
            todo.pop_back();
            if (seen[i]) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: a node is reachable from the root more than once");
            seen[i] = 1;
            visited++;
            const crt_bvh_node& n = d->nodes[i];
            if (n.lc < 0 && n.rc < 0) {
                for (uint32_t k = 0; k < n.n; k++) {
                    if (owned[(uint32_t)n.it + k]) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: two leaves share a triangle");
                    owned[(uint32_t)n.it + k] = 1;
                }
            } else { todo.push_back(n.lc); todo.push_back(n.rc); }
        }
        if (visited != d->n_nodes) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: nodes that the root does not reach");
    }
    if (d->n_materials >= (1u << 30)) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: more than 2^30 - 1 materials (the triangle rows keep two flag bits beside the index)");
    for (uint32_t i = 0; i < d->n_tris; i++)
        if (d->tris[i].material < 0 || (uint32_t)d->tris[i].material >= d->n_materials) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: triangle material index out of range");
    for (uint32_t i = 0; i < d->n_light_tris; i++)
        if (d->light_tris[i].material < 0 || (uint32_t)d->light_tris[i].material >= d->n_materials) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: light triangle material index out of range");
    for (uint32_t i = 0; i < d->n_lights; i++)
        if (d->lights[i].count == 0 || (uint64_t)d->lights[i].first_tri + d->lights[i].count > d->n_light_tris) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: light range outside the light triangle array");
    return CRT_OK;
}

struct Shard {
    uint32_t tiles_x, tiles_y, n_tiles, local_tiles, nslots;
};
struct RingPlan { uint32_t samples, spsh, shards; }; // commit ring of a launch: samples held (0 = one radiance per work item), pixel slots per cursor shard
Shard make_shard(uint32_t w, uint32_t h, uint32_t world)
{
    Shard s;
    s.tiles_x = (w + CRT_TILE - 1) / CRT_TILE;
    s.tiles_y = (h + CRT_TILE - 1) / CRT_TILE;
    s.n_tiles = s.tiles_x * s.tiles_y;
    s.local_tiles = (s.n_tiles + world - 1) / world; // padded so every rank writes the same number of slots
    s.nslots = s.local_tiles * 64u;
    return s;
}

// 1 / n for n a power of two (exactly representable), else 0
float inv_if_pow2(int32_t n) { return (n > 0 && (n & (n - 1)) == 0) ? 1.0f / (float)n : 0.0f; }

uint32_t env_u32(const char* name, uint32_t dflt)
{
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    long x = std::strtol(v, nullptr, 10);
    return x > 0 ? (uint32_t)x : dflt;
}

// The instantiation of k_mega3 for a traversal mode (0 FAST, 1 REFERENCE, 2 EXACT), with or without counters, every sample traced
// or not (FAST only), render or query form, 32- or 16-bit stack entries (never for REFERENCE)
// The decoupled-leaves form (Pool4LdsT) of a launch: CRT_TRAVERSAL_EXACT on a scene whose four-wide nodes fit the 16-bit entries of
// its stack (inner nodes only: up to about 160 000 triangles) and whose leaf records fit a queue entry.  Since the traversal steps
// alternate without the scheduler (crt_mega3.hip, CHAIN_MIN) it is the faster form on every scene measured (stand-in cornell-box
// - 4 %, veach-mis equal, the 102 412-triangle variant - 5 % against the coupled form with 32-bit entries), and the one whose
// layout does not change with the number of leaves.  CRT_DEC=1 / 0 forces / forbids it (tests, A/B).
static bool use_dec(const crt_scene* sc, int mode)
{
    if (mode != 2 || !sc->dec_ok) return false;
    const char* e = std::getenv("CRT_DEC");
    if (e && e[0] == '0') return false;
    // (round 6: also beyond 32 768 four-wide nodes, where its stack has three 32-bit levels in LDS -- a 348 172-triangle cornell-box
    // --detail 7,5, 88 230 nodes: 14.6 ms at spp 64 against 16.1 ms of the coupled form with 32-bit entries, tools/big_mesh_probe.py)
    return true;
}
// The copy of the 4-wide tree without its rows of refs (nodes4i, round 6): the decoupled-leaves kernels with 16-bit stack entries take it
// whenever the scene offers it (crt_scene::impl_ok: leaves of one record, <= 32 768 nodes); CRT_IMPL=0 keeps them on nodes4 (tests, A/B).
static bool use_impl(const crt_scene* sc, bool dec, bool r16)
{
    if (!dec || !r16 || !sc->impl_ok) return false;
    const char* e = std::getenv("CRT_IMPL");
    return !(e && e[0] == '0');
}
// Which pipeline renders: 4 = k_mega3 (the product), 2 = the wavefront pipeline (k_logic + k_trace).  k_mega3 keeps the best
// triangle's offset inside its leaf in 16 bits, addresses nodes and leaf records with 32-bit byte offsets and the traversal stack
// depth in 8 bits; scenes beyond any of these fall back to the wavefront pipeline, which has no such limits.  The CRT_TEST_*
// variables lower the limits so that the tests can force each fallback on a small scene.
uint32_t choose_pipeline(const crt_scene* sc);

const uint64_t kMaxChunkItems = 1ull << 30; // paths per chunk (17 GB of per-path radiance: sized for 288 GB of HBM, every launch ends with a 2 ms tail)

struct TraceSetup {
    TParams T;
    size_t lds;
    int mode_id;
    uint32_t blocks;
};
// Everything a k_trace launch over `pool` needs (grid sized to the device's residency: the kernel is persistent).
TraceSetup make_trace_setup(crt_scene* sc, const Pool& pool, uint32_t traversal, bool want_stats, int half = 0, int n_halves = 1)
{
    TraceSetup S;
    std::memset(&S.T, 0, sizeof(S.T));
    TParams& T = S.T;
    T.sc = sc->dev; T.pool = pool; T.counters = sc->counters.p;
    T.slot_next = sc->slot_next[half].p;
    T.refill_min = (int32_t)std::min<uint32_t>(64, env_u32("CRT_REFILL_MIN", REFILL_MIN));
    T.leaf_min = (int32_t)std::min<uint32_t>(64, env_u32("CRT_LEAF_MIN", LEAF_MIN));
    T.slots_per_shard = ((pool.n + SLOT_SHARDS - 1) / SLOT_SHARDS + 63u) & ~63u;
    // LDS holds the first levels of the traversal stack; the rest (rarely touched) spills to HBM/L2
    const int lds_cap = (int)std::min<uint32_t>((uint32_t)sc->stack_cap, std::max(2u, env_u32("CRT_STACK_LDS", 8)));
    T.stack_cap = lds_cap;
    S.lds = (size_t)lds_cap * 256 * sizeof(int2);
    S.mode_id = (traversal == CRT_TRAVERSAL_REFERENCE ? 2 : traversal == CRT_TRAVERSAL_EXACT ? 4 : 0) + (want_stats ? 1 : 0);
    int per_cu = trace_blocks_per_cu(S.mode_id, S.lds);
    // with two pool halves in flight leave room for the other half's k_logic blocks
    const uint32_t dflt_per_cu = n_halves > 1 ? 3u : 64u; // measured best on MI355X (C2): 3 trace blocks + logic blocks per CU
    per_cu = (int)std::min<uint32_t>((uint32_t)per_cu, env_u32("CRT_TRACE_BLOCKS_PER_CU", dflt_per_cu));
    S.blocks = std::min<uint32_t>((pool.n + 255) / 256, (uint32_t)(sc->n_cus * per_cu));
    const int spill_levels = std::max(1, sc->stack_cap - lds_cap);
    T.spill_stride = S.blocks * 256u;
    sc->spill[half].ensure((size_t)spill_levels * T.spill_stride);
    T.spill = sc->spill[half].p;
    return S;
}
void launch_trace_pass(crt_scene* sc, const TraceSetup& S, hipStream_t st)
{
    HIP_CHECK(hipMemsetAsync(S.T.slot_next, 0, (size_t)SLOT_SHARDS * SLOT_STRIDE * sizeof(unsigned int), st));
    launch_trace(S.mode_id, S.T, S.blocks, S.lds, st);
}

// Renders samples [s_begin, s_begin + s_count) of the prm->spp samples per pixel into the scene's accumulator
// (temp_color += L_k / spp in sample order, Render.cuh:348); the range that ends at spp also tone-maps and writes the frame.
int render_impl(crt_scene* sc, const crt_camera* cam, const crt_params* prm, void* d_rgb, void* d_mean, hipStream_t st, crt_stats* stats,
                uint32_t s_begin = 0, uint32_t s_count = 0xffffffffu)
{
    if (!sc || !cam || !prm) return fail(CRT_ERR_INVALID_ARG, "crt_render: null argument");
    if (s_count == 0xffffffffu) s_count = prm->spp > s_begin ? prm->spp - s_begin : 0;
    if (s_count == 0 || (uint64_t)s_begin + s_count > prm->spp) return fail(CRT_ERR_INVALID_ARG, "crt_render: sample range outside [0, spp)");
    const uint32_t s_end = s_begin + s_count;
    if (!d_rgb && s_end == prm->spp) return fail(CRT_ERR_INVALID_ARG, "crt_render: null frame buffer");
    if (prm->width == 0 || prm->height == 0 || prm->spp == 0) return fail(CRT_ERR_INVALID_ARG, "crt_render: width, height and spp must be positive");
    if (prm->world == 0 || prm->rank >= prm->world) return fail(CRT_ERR_INVALID_ARG, "crt_render: need rank < world");
    if (prm->light_sample_n < 0 || prm->light_sample_n > 4096) return fail(CRT_ERR_INVALID_ARG, "crt_render: light_sample_n must be in [0, 4096]");
    if ((uint64_t)prm->width * prm->height > 0xffffffffull) return fail(CRT_ERR_UNSUPPORTED, "crt_render: more than 2^32 pixels");
    if (prm->traversal != CRT_TRAVERSAL_FAST && prm->traversal != CRT_TRAVERSAL_REFERENCE && prm->traversal != CRT_TRAVERSAL_EXACT)
        return fail(CRT_ERR_INVALID_ARG, "crt_render: unknown traversal mode");
    const bool want_stats = (prm->flags & CRT_FLAG_STATS) != 0;
    const bool tiled = (prm->flags & CRT_FLAG_TILED_OUTPUT) != 0;
    if (prm->world > 1 && !tiled) return fail(CRT_ERR_INVALID_ARG, "crt_render: world > 1 needs CRT_FLAG_TILED_OUTPUT");
    if ((uint64_t)sc->dev.n_lights * (uint64_t)prm->light_sample_n > 0xffffu) return fail(CRT_ERR_UNSUPPORTED, "crt_render: more than 65535 next-event samples per vertex");
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        Shard sh = make_shard(prm->width, prm->height, prm->world);
        const uint64_t max_items = std::min<uint64_t>(kMaxChunkItems, 1ull << std::min(30u, env_u32("CRT_CHUNK_LOG2", 30))); // (test hook: small chunks)
        uint32_t chunk = (uint32_t)std::min<uint64_t>(s_count, std::max<uint64_t>(1, max_items / sh.nslots));
        uint64_t cap = (uint64_t)chunk * sh.nslots;
        const uint32_t pipeline = choose_pipeline(sc);
        // ---- commit ring (megakernel only, CRT_FLAG_BOUNDED_RADIANCE): radiance storage for a window of samples, the sum
        // c += L_k / spp made inside the launch; the whole sample range is then ONE launch.  ring samples = 4 x the depth of the work in
        // flight (pool slots / pixel slots), at least 32: a shard is held back only when one of its paths takes four times as long as
        // the rest of the pool.
        RingPlan ring;
        std::memset(&ring, 0, sizeof(ring));
        if (pipeline == 4 && !want_stats) {
            // cursor shards: the commits of a shard are a serial chain (one wave, a memory round trip per 256 pixel slots), so a ring
            // launch has more and smaller shards than the 64 of a launch without: about 1 024 pixel slots each, at most 1 024 shards
            uint32_t shards = ITEM_SHARDS;
            while (shards < 1024u && sh.nslots / (shards * 2u) >= 1024u) shards *= 2u;
            const uint32_t spsh = ((sh.nslots + shards - 1) / shards + 63u) & ~63u;
            const uint64_t pool_slots = (uint64_t)sc->n_cus * 16u * (uint64_t)POOL3_P;
            uint32_t rs = 32;
            while (rs < 65536u && (uint64_t)rs * sh.nslots < 4ull * pool_slots) rs <<= 1;
            const uint32_t forced = env_u32("CRT_COMMIT_RING_LOG2", 0); // (test hook: a ring of 2^n samples, with or without the flag)
            if (forced) rs = 1u << std::min(16u, forced);
            const uint64_t per_shard = (uint64_t)spsh * s_count;
            const bool fits32 = per_shard * shards < 0xffffffffull;
            if ((forced || (prm->flags & CRT_FLAG_BOUNDED_RADIANCE)) && rs < s_count && fits32) {
                ring.samples = rs; ring.spsh = spsh; ring.shards = shards;
                chunk = s_count;
                cap = (uint64_t)rs * spsh * shards;
            }
        }
        const uint32_t pool_log2 = std::min(26u, std::max(8u, env_u32("CRT_POOL_LOG2", 22)));
        const uint32_t pool_n = (uint32_t)std::min<uint64_t>((cap + 255) / 256 * 256, 1ull << pool_log2);
        const int batch_max = (int)std::min<uint32_t>(kMaxBatch, env_u32("CRT_ROUND_BATCH", 16));
        int batch = batch_max;

        if (ring.samples) sc->ring_L.ensure_uncached(cap);
        else sc->L.ensure(cap);
        sc->last_radiance_bytes = cap * sizeof(float4);
        sc->last_ring_samples = ring.samples;
        sc->accum.ensure_uncached((size_t)sh.nslots * 3); // (always uncached: a progressive render may switch between launches with and without the ring)
        const bool timing = stats != nullptr;
        if (timing && sc->ev.size() < (size_t)(4 * kMaxBatch + 4)) {
            while (sc->ev.size() < (size_t)(4 * kMaxBatch + 4)) {
                hipEvent_t e;
                HIP_CHECK(hipEventCreate(&e));
                sc->ev.push_back(e);
            }
        }
        const size_t counters_bytes = (size_t)CNT_SHARDS * CNT_STRIDE * sizeof(unsigned long long);
        HIP_CHECK(hipMemsetAsync(sc->counters.p, 0, counters_bytes, st));
        auto counter_sum = [&](int c) {
            unsigned long long v = 0;
            for (int s = 0; s < CNT_SHARDS; s++) v += sc->h_counters[s * CNT_STRIDE + c];
            return v;
        };
        unsigned long long alive_seen = 0;

        if (pipeline == 4) {
            // ---------- fused persistent megakernel: one launch per chunk ----------
            const bool reference = prm->traversal == CRT_TRAVERSAL_REFERENCE;
            const bool exact = prm->traversal == CRT_TRAVERSAL_EXACT;
            const int mode_id = (reference ? 2 : exact ? 4 : 0) + (want_stats ? 1 : 0);
            const int mode3 = reference ? 1 : exact ? 2 : 0;
            const bool dec = use_dec(sc, mode3);
            const bool r16 = use_ref16(sc, mode3, dec);
            const bool impl = use_impl(sc, dec, r16);
            const Mega3Kernel kern3 = mega3_kernel(mode3, want_stats, mode3 != 1 && (prm->flags & CRT_FLAG_TRACE_ALL) != 0, false, r16, ring.samples != 0, dec, impl);
            const uint32_t pool_p = mega3_pool_p(dec, ring.samples != 0);
            MParams M;
            std::memset(&M, 0, sizeof(M));
            int per_cu = 1;
            uint32_t blocks, lanes;
            {
                // one wave per workgroup, pool_p rays per wave
                auto q3 = [&](int* n) {
                    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(n, kern3, 64, 0);
                    if (e != hipSuccess || *n < 1) *n = 1;
                };
                q3(&per_cu);
                per_cu = (int)std::min<uint32_t>((uint32_t)per_cu, env_u32("CRT_MEGA_BLOCKS_PER_CU", 64));
                const uint64_t most_items = ring.samples ? (uint64_t)s_count * sh.nslots : cap;
                blocks = std::min<uint32_t>((uint32_t)std::min<uint64_t>((most_items + pool_p - 1) / pool_p, 0x7fffffffull), (uint32_t)(sc->n_cus * per_cu));
                lanes = blocks * pool_p; // pool slots
            }
            sc->p_vx.ensure(lanes); sc->p_la.ensure(lanes); sc->p_cc.ensure(lanes); sc->p_id.ensure(lanes);
#if !CRT_X_NOVN
            sc->p_vn.ensure(lanes); // (round 5: k_mega3 reads normal and material from tri_nm through the triangle in its 8-byte id plane, crt_path.h)
#endif
            sc->p_rec_a.ensure((size_t)lanes * CRT_BOUNCE_STACK_SIZE);
            sc->p_rec_b.ensure((size_t)lanes * CRT_BOUNCE_STACK_SIZE);
            // (16-bit layout: a ray on the reference-arithmetic path keeps its whole stack in the global area)
            const int lds_cap = mega3_lds_levels(dec, r16);
            const int spill_levels = r16 ? std::max(1, sc->stack_cap) : std::max(1, sc->stack_cap - lds_cap);
            sc->spill[0].ensure((size_t)spill_levels * lanes);
            Pool pool;
            std::memset(&pool, 0, sizeof(pool));
            pool.vx = sc->p_vx.p; pool.la = sc->p_la.p; pool.cc = sc->p_cc.p; pool.vn = sc->p_vn.p; pool.id = sc->p_id.p;
            pool.rec_a = sc->p_rec_a.p; pool.rec_b = sc->p_rec_b.p; pool.n = lanes;
            LParams P;
            std::memset(&P, 0, sizeof(P));
            P.sc = sc->dev; P.pool = pool;
            std::memcpy(P.eye, cam->eye, sizeof(P.eye));
            std::memcpy(P.inv_view, cam->inv_view, sizeof(P.inv_view));
            P.scale = det_tanf(cam->fov_y / 2);                       // Render.cuh:338
            P.ar = (float)prm->width / (float)prm->height;            // Render.cuh:339
            P.width = prm->width; P.height = prm->height;
            P.p_rr = prm->p_rr; P.lsn = prm->light_sample_n; P.seed = prm->seed;
            P.rank = prm->rank; P.world = prm->world; P.tiles_x = sh.tiles_x; P.n_tiles = sh.n_tiles;
            P.nslots = sh.nslots;
            P.inv_lsn_pow2 = inv_if_pow2(prm->light_sample_n); P.lsn_div = make_fastdiv((uint32_t)std::max(1, prm->light_sample_n)); P.nslots_div = make_fastdiv(sh.nslots); P.tiles_x_div = make_fastdiv(sh.tiles_x);
            P.L = sc->L.p; P.counters = sc->counters.p; P.item_next = sc->item_next.p; P.n_mats = sc->n_mats;
            M.sc = sc->dev; M.counters = sc->counters.p; M.spill = sc->spill[0].p; M.spill_stride = lanes; M.stack_cap = lds_cap;
            AParams A;
            std::memset(&A, 0, sizeof(A));
            A.width = prm->width; A.height = prm->height; A.spp = prm->spp;
            A.rank = prm->rank; A.world = prm->world; A.tiles_x = sh.tiles_x; A.n_tiles = sh.n_tiles;
            A.nslots = sh.nslots; A.tiled_output = tiled ? 1 : 0;
            A.L = sc->L.p; A.accum = sc->accum.p;
            A.out_rgb = (uint8_t*)d_rgb; A.out_mean = (float*)d_mean;
            hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
            if (timing) { e0 = sc->ev[0]; e1 = sc->ev[1]; e2 = sc->ev[2]; e3 = sc->ev[3]; HIP_CHECK(hipEventRecord(e0, st)); }
            double kernel_ms = 0.0;
            uint32_t launches = 0;
            for (uint32_t s0 = s_begin; s0 < s_end; s0 += chunk) {
                uint32_t ns = std::min(chunk, s_end - s0);
                P.sample_begin = s0;
                P.n_items = (uint32_t)((uint64_t)ns * sh.nslots);
                P.items_per_shard = ((P.n_items + ITEM_SHARDS - 1) / ITEM_SHARDS + 63u) & ~63u;
                if (ring.samples) { // cursor shard = ring.spsh pixel slots x ns samples
                    P.items_per_shard = ring.spsh * ns;
                    P.n_items = P.items_per_shard * ring.shards;
                    P.ring_mask = ring.samples - 1u; P.spsh = ring.spsh; P.spsh_div = make_fastdiv(ring.spsh); P.ring_shards = ring.shards;
                    P.ring_stride = ring.spsh * ring.shards; P.n_samples = ns; P.tail_first = P.items_per_shard; P.spp_f = (float)prm->spp;
                    sc->ring_done.ensure_uncached((size_t)ring.shards * ring.samples);
                    sc->ring_state.ensure_uncached((size_t)ring.shards * ITEM_STRIDE);
                    P.ring_done = sc->ring_done.p; P.ring_state = sc->ring_state.p; P.accum = sc->accum.p; P.L = sc->ring_L.p;
                    std::vector<unsigned int>& state = sc->ring_state_host; // (a member: the copy below may still read it after this scope)
                    state.assign((size_t)ring.shards * ITEM_STRIDE, 0u);
                    for (uint32_t slot = 0; slot < sh.nslots; slot++) { // word 1: the pixel slots of the shard that are pixels
                        const uint32_t tile = (slot >> 6) * prm->world + prm->rank, pix = slot & 63u;
                        if (tile >= sh.n_tiles) continue;
                        const uint32_t ty = tile / sh.tiles_x, tx = tile - ty * sh.tiles_x;
                        if (tx * CRT_TILE + (pix & 7u) < prm->width && ty * CRT_TILE + (pix >> 3) < prm->height) state[(size_t)(slot / ring.spsh) * ITEM_STRIDE + 1]++;
                    }
                    HIP_CHECK(hipMemcpyAsync(sc->ring_state.p, state.data(), state.size() * sizeof(unsigned int), hipMemcpyHostToDevice, st));
                    HIP_CHECK(hipMemsetAsync(sc->ring_done.p, 0, (size_t)ring.shards * ring.samples * sizeof(unsigned int), st));
                }
                // the paths that stop at their first vertex are handed out last (k_order_items): 1 % of a whole C2 frame on one GPU,
                // 8 % of a rank's share on eight.  CRT_ITEM_ORDER=0 switches it off.
                P.item_list = nullptr;
                {
                    const char* eo = std::getenv("CRT_ITEM_ORDER");
                    const bool order = !(eo && eo[0] == '0');
                    if (order && P.n_items > 0) {
                        // the window: the last 2^19 work items of every shard (measured on C2, wall time of a rank's share at 1 / 2 / 4 / 8 ranks: no
                        // order 107.3 / 54.9 / 28.4 / 15.9 ms; 2^17: 107.4 / 54.5 / 28.3 / 15.2; 2^19: 106.8 / 54.2 / 27.7 / 14.6; whole shards:
                        // 107.2 / 54.1 / 27.5 / 14.6 -- the pass itself costs 0.9 ms for the 245.8 M items of a whole frame)
                        P.order_window = std::min<uint32_t>(P.items_per_shard, env_u32("CRT_ORDER_WINDOW", 1u << 19));
                        if (ring.samples) { // the window may span half the ring: its items stand for the launch's last sample at the gate
                            P.order_window = std::min<uint32_t>(P.order_window, (ring.samples / 2u) * ring.spsh);
                            P.tail_first = P.items_per_shard - P.order_window;
                        }
                        P.items_per_shard_div = make_fastdiv(std::max(1u, P.items_per_shard));
                        const uint32_t n_sh = ring.samples ? ring.shards : (uint32_t)ITEM_SHARDS;
                        sc->item_list.ensure((size_t)n_sh * P.order_window);
                        sc->order_cnt.ensure((size_t)n_sh * 2 * 32);
                        HIP_CHECK(hipMemsetAsync(sc->order_cnt.p, 0, (size_t)n_sh * 2 * 32 * sizeof(unsigned int), st));
                        const uint32_t spans = (P.order_window + 1023u) / 1024u;
                        launch_order_items(ring.samples != 0, n_sh * spans, st, P, sc->item_list.p, sc->order_cnt.p);
                        HIP_CHECK(hipGetLastError());
                        P.item_list = sc->item_list.p;
                    }
                }
                P.items_per_shard_div = make_fastdiv(std::max(1u, P.items_per_shard));
                M.P = P;
                HIP_CHECK(hipMemsetAsync(sc->item_next.p, 0, (size_t)(ring.samples ? ring.shards : (uint32_t)ITEM_SHARDS) * ITEM_STRIDE * sizeof(unsigned int), st));
                if (timing) HIP_CHECK(hipEventRecord(e1, st));
                if (s0 == s_begin) HIP_CHECK(hipEventRecord(sc->ev_k0, st));
                {
                    MParams3 M3;
                    M3.M = M;
                    M3.spill = (int*)sc->spill[0].p; // (one word per entry; the buffer is sized for the two-word entries of k_trace)
                    M3.force_exact = (prm->flags & CRT_FLAG_FORCE_EXACT) ? 1u : 0u;
                    M3.dbg_loads = 0; M3.dbg_valu = 0;
                    if (!bbprof_launch(kern3, M3, blocks, st)) hipLaunchKernelGGL(kern3, dim3(blocks), dim3(64), 0, st, M3);
                }
                HIP_CHECK(hipGetLastError());
                if (s0 + ns >= s_end) HIP_CHECK(hipEventRecord(sc->ev_k1, st));
                if (timing) {
                    HIP_CHECK(hipEventRecord(e2, st));
                    HIP_CHECK(hipStreamSynchronize(st));
                    float ms = 0.0f;
                    HIP_CHECK(hipEventElapsedTime(&ms, e1, e2));
                    kernel_ms += ms;
                }
                launches++;
                sc->last_launches = launches;
                A.chunk_samples = ns;
                A.first_chunk = s0 == 0; A.last_chunk = s0 + ns >= prm->spp;
                if (ring.samples) { A.chunk_samples = 0; A.first_chunk = 0; } // the sum is in the accumulator already: tone mapping only
                if (!ring.samples || A.last_chunk) {
                    launch_accumulate(A, st);
                    HIP_CHECK(hipGetLastError());
                }
                sc->acc.samples = A.last_chunk ? 0u : s0 + ns; sc->acc.spp = prm->spp; sc->acc.width = prm->width; sc->acc.height = prm->height;
                sc->acc.rank = prm->rank; sc->acc.world = prm->world; sc->acc.tiled = tiled ? 1u : 0u;
            }
            if (stats) {
                HIP_CHECK(hipEventRecord(e3, st));
                HIP_CHECK(hipMemcpyAsync(sc->h_counters, sc->counters.p, counters_bytes, hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                std::memset(stats, 0, sizeof(*stats));
                stats->paths = counter_sum(C_PATHS); stats->rays = counter_sum(C_RAYS); stats->shadow_rays = counter_sum(C_SHADOW);
                stats->probe_rays = counter_sum(C_PROBE);
                stats->rays_untraced = counter_sum(C_UNTRACED);
                stats->inner_pops = counter_sum(C_INNER); stats->leaf_pops = counter_sum(C_LEAF); stats->tri_tests = counter_sum(C_TESTS);
                stats->hits = counter_sum(C_HITS);
                stats->stack_sum = counter_sum(C_SUMSP);
                for (int sh2 = 0; sh2 < CNT_SHARDS; sh2++) stats->stack_max = std::max<uint64_t>(stats->stack_max, sc->h_counters[sh2 * CNT_STRIDE + C_MAXSP]);
                float total = 0.0f;
                HIP_CHECK(hipEventElapsedTime(&total, e0, e3));
                stats->phase_cycles[0] = counter_sum(C_CYC_LOGIC); stats->phase_cycles[1] = counter_sum(C_CYC_LEAF);
                stats->phase_cycles[2] = counter_sum(C_CYC_INNER); stats->phase_cycles[3] = counter_sum(C_CYC_OTHER);
                for (int i = 0; i < 20; i++) stats->phase_cycles[4 + i] = counter_sum(C_DIAG + i);
                stats->kernel_ms = (float)kernel_ms;
                stats->logic_ms = 0.0f;
                stats->total_ms = total;
                stats->kernel_launches = launches;
            }
            return CRT_OK;
        }

        // The pool is split into halves that run on two streams: the HBM-bound k_logic of one half
        // overlaps the issue-bound k_trace of the other.
        const int n_halves = (pool_n >= 2 * 65536u && env_u32("CRT_STREAMS", 2) >= 2) ? 2 : 1;
        const uint32_t half_n = n_halves == 2 ? ((pool_n / 2 + 255) / 256 * 256) : pool_n;
        sc->p_ro.ensure((size_t)half_n * n_halves); sc->p_rd.ensure((size_t)half_n * n_halves); sc->p_vx.ensure((size_t)half_n * n_halves);
        sc->p_la.ensure((size_t)half_n * n_halves); sc->p_cc.ensure((size_t)half_n * n_halves); sc->p_res.ensure((size_t)half_n * n_halves);
        sc->p_vn.ensure((size_t)half_n * n_halves); sc->p_id.ensure((size_t)half_n * n_halves);
        sc->p_rec_a.ensure((size_t)half_n * n_halves * CRT_BOUNCE_STACK_SIZE);
        sc->p_rec_b.ensure((size_t)half_n * n_halves * CRT_BOUNCE_STACK_SIZE);
        Pool pools[2];
        for (int h = 0; h < n_halves; h++) {
            Pool& pool = pools[h];
            const size_t o = (size_t)h * half_n;
            pool.ro = sc->p_ro.p + o; pool.rd = sc->p_rd.p + o; pool.vx = sc->p_vx.p + o; pool.la = sc->p_la.p + o; pool.cc = sc->p_cc.p + o;
            pool.vn = sc->p_vn.p + o; pool.id = sc->p_id.p + o; pool.res = sc->p_res.p + o;
            pool.rec_a = sc->p_rec_a.p + o * CRT_BOUNCE_STACK_SIZE; pool.rec_b = sc->p_rec_b.p + o * CRT_BOUNCE_STACK_SIZE;
            pool.n = half_n;
        }
        hipStream_t streams[2] = {st, sc->aux_stream};

        LParams P;
        std::memset(&P, 0, sizeof(P));
        P.sc = sc->dev;
        std::memcpy(P.eye, cam->eye, sizeof(P.eye));
        std::memcpy(P.inv_view, cam->inv_view, sizeof(P.inv_view));
        P.scale = det_tanf(cam->fov_y / 2);                       // Render.cuh:338
        P.ar = (float)prm->width / (float)prm->height;            // Render.cuh:339
        P.width = prm->width; P.height = prm->height;
        P.p_rr = prm->p_rr; P.lsn = prm->light_sample_n; P.seed = prm->seed;
        P.rank = prm->rank; P.world = prm->world; P.tiles_x = sh.tiles_x; P.n_tiles = sh.n_tiles;
        P.nslots = sh.nslots;
        P.inv_lsn_pow2 = inv_if_pow2(prm->light_sample_n); P.lsn_div = make_fastdiv((uint32_t)std::max(1, prm->light_sample_n)); P.nslots_div = make_fastdiv(sh.nslots); P.tiles_x_div = make_fastdiv(sh.tiles_x);
        P.L = sc->L.p;
        P.counters = sc->counters.p;
        P.item_next = sc->item_next.p;
        P.n_mats = sc->n_mats;
        const bool lds_tables = sc->n_mats <= LOGIC_TABLE_MAX && (uint32_t)sc->dev.n_lights <= LOGIC_TABLE_MAX;

        TraceSetup TS[2];
        LParams PH[2];
        for (int h = 0; h < n_halves; h++) TS[h] = make_trace_setup(sc, pools[h], prm->traversal, want_stats, h, n_halves);

        AParams A;
        std::memset(&A, 0, sizeof(A));
        A.width = prm->width; A.height = prm->height; A.spp = prm->spp;
        A.rank = prm->rank; A.world = prm->world; A.tiles_x = sh.tiles_x; A.n_tiles = sh.n_tiles;
        A.nslots = sh.nslots; A.tiled_output = tiled ? 1 : 0;
        A.L = sc->L.p; A.accum = sc->accum.p;
        A.out_rgb = (uint8_t*)d_rgb; A.out_mean = (float*)d_mean;

        double trace_ms = 0.0, logic_ms = 0.0;
        uint32_t trace_launches = 0;
        hipEvent_t ev_begin = nullptr, ev_end = nullptr;
        if (timing) {
            ev_begin = sc->ev[4 * kMaxBatch + 2];
            ev_end = sc->ev[4 * kMaxBatch + 3];
            HIP_CHECK(hipEventRecord(ev_begin, st));
        }
        const dim3 pool_grid((half_n + 255) / 256);
        const int evs_per_half = 2 * kMaxBatch + 1;
        for (uint32_t s0 = s_begin; s0 < s_end; s0 += chunk) {
            uint32_t ns = std::min(chunk, s_end - s0);
            P.sample_begin = s0;
            P.n_items = (uint32_t)((uint64_t)ns * sh.nslots);
            P.items_per_shard = ((P.n_items + ITEM_SHARDS - 1) / ITEM_SHARDS + 63u) & ~63u;
            HIP_CHECK(hipMemsetAsync(sc->item_next.p, 0, (size_t)ITEM_SHARDS * ITEM_STRIDE * sizeof(unsigned int), st));
            for (int h = 0; h < n_halves; h++) {
                PH[h] = P;
                PH[h].pool = pools[h];
                launch_pool_init(pool_grid.x, st, pools[h]);
            }
            HIP_CHECK(hipGetLastError());
            for (;;) {
                if (n_halves == 2) { // fork: the second half's chain follows what is queued on st so far
                    HIP_CHECK(hipEventRecord(sc->ev_fork, st));
                    HIP_CHECK(hipStreamWaitEvent(sc->aux_stream, sc->ev_fork, 0));
                }
                for (int h = 0; h < n_halves; h++)
                    if (timing) HIP_CHECK(hipEventRecord(sc->ev[h * evs_per_half], streams[h]));
                for (int b = 0; b < batch; b++) {
                    for (int h = 0; h < n_halves; h++) {
                        hipStream_t hs = streams[h];
                        hipEvent_t* ev = sc->ev.data() + h * evs_per_half;
                        launch_logic(lds_tables, pool_grid.x, hs, PH[h]);
                        if (timing) HIP_CHECK(hipEventRecord(ev[2 * b + 1], hs));
                        launch_trace_pass(sc, TS[h], hs);
                        if (timing) HIP_CHECK(hipEventRecord(ev[2 * b + 2], hs));
                    }
                }
                HIP_CHECK(hipGetLastError());
                if (n_halves == 2) { // join
                    HIP_CHECK(hipEventRecord(sc->ev_join, sc->aux_stream));
                    HIP_CHECK(hipStreamWaitEvent(st, sc->ev_join, 0));
                }
                HIP_CHECK(hipMemcpyAsync(sc->h_counters, sc->counters.p, counters_bytes, hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                if (timing) {
                    double bl = 0.0, bt = 0.0;
                    for (int h = 0; h < n_halves; h++) {
                        hipEvent_t* ev = sc->ev.data() + h * evs_per_half;
                        for (int b = 0; b < batch; b++) {
                            float a = 0.0f, c = 0.0f;
                            HIP_CHECK(hipEventElapsedTime(&a, ev[2 * b], ev[2 * b + 1]));
                            HIP_CHECK(hipEventElapsedTime(&c, ev[2 * b + 1], ev[2 * b + 2]));
                            bl += a; bt += c;
                        }
                    }
                    logic_ms += bl; trace_ms += bt;
                    if (std::getenv("CRT_TRACE_LOG"))
                        fprintf(stderr, "[crt] rounds %u..%u: rays in batch %llu, logic %.3f ms, trace %.3f ms\n", trace_launches, trace_launches + batch - 1,
                                (unsigned long long)(counter_sum(C_ALIVE) - alive_seen), bl, bt);
                }
                trace_launches += (uint32_t)(batch * n_halves);
                unsigned long long alive_now = counter_sum(C_ALIVE);
                if (alive_now == alive_seen) break; // no slot emitted a ray during the whole batch: chunk done
                // once the pool runs dry (no more regeneration) check more often, so that few empty rounds are launched
                const unsigned long long per_round = (alive_now - alive_seen) / (unsigned long long)batch;
                batch = per_round * 8 < (unsigned long long)half_n * n_halves ? std::min(batch_max, 4) : batch_max;
                if (per_round * 512 < (unsigned long long)half_n * n_halves) batch = std::min(batch_max, 2);
                alive_seen = alive_now;
            }
            A.chunk_samples = ns;
            A.first_chunk = s0 == 0; A.last_chunk = s0 + ns >= prm->spp;
            launch_accumulate(A, st);
            HIP_CHECK(hipGetLastError());
            sc->acc.samples = A.last_chunk ? 0u : s0 + ns; sc->acc.spp = prm->spp; sc->acc.width = prm->width; sc->acc.height = prm->height;
            sc->acc.rank = prm->rank; sc->acc.world = prm->world; sc->acc.tiled = tiled ? 1u : 0u;
        }
        if (stats) {
            HIP_CHECK(hipEventRecord(ev_end, st));
            HIP_CHECK(hipStreamSynchronize(st));
            std::memset(stats, 0, sizeof(*stats));
            stats->paths = counter_sum(C_PATHS); stats->rays = counter_sum(C_RAYS); stats->shadow_rays = counter_sum(C_SHADOW);
            stats->probe_rays = counter_sum(C_PROBE);
            stats->inner_pops = counter_sum(C_INNER); stats->leaf_pops = counter_sum(C_LEAF); stats->tri_tests = counter_sum(C_TESTS);
            stats->hits = counter_sum(C_HITS);
            stats->stack_sum = counter_sum(C_SUMSP);
            for (int sh2 = 0; sh2 < CNT_SHARDS; sh2++) stats->stack_max = std::max<uint64_t>(stats->stack_max, sc->h_counters[sh2 * CNT_STRIDE + C_MAXSP]);
            float total = 0.0f;
            HIP_CHECK(hipEventElapsedTime(&total, ev_begin, ev_end));
            stats->kernel_ms = (float)trace_ms;
            stats->logic_ms = (float)logic_ms;
            stats->total_ms = total;
            stats->kernel_launches = trace_launches;
        }
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

uint32_t choose_pipeline(const crt_scene* sc)
{
    uint32_t pipeline = env_u32("CRT_PIPELINE", 4);
    if (pipeline != 2) pipeline = 4;
    const uint64_t max_leaf = env_u32("CRT_TEST_MAX_LEAF", CRT_MEGA3_MAX_LEAF);
    const uint64_t max_bytes = std::getenv("CRT_TEST_MAX_BYTES") ? (uint64_t)env_u32("CRT_TEST_MAX_BYTES", 0xffffffffu) : (1ull << 32);
    const uint32_t max_stack = env_u32("CRT_TEST_MAX_STACK", CRT_MEGA3_MAX_STACK);
    if (pipeline == 4 && sc->max_leaf > max_leaf) pipeline = 2;
    if (pipeline == 4 && (sc->nodes4.n * sizeof(float4) >= max_bytes || sc->nodes3.n * sizeof(float4) >= max_bytes || sc->leaf_geo.n * sizeof(float4) >= max_bytes))
        pipeline = 2; // (33 M nodes / 53 M records)
    if (pipeline == 4 && (uint32_t)sc->stack_cap > max_stack) pipeline = 2; // a deeper stack would spill into the flag bits of word D
    return pipeline;
}

} // namespace

extern "C" {

int crt_device_count(int* count)
{
    if (!count) return fail(CRT_ERR_INVALID_ARG, "crt_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(CRT_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    return CRT_OK;
}

int crt_shard_slots(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, uint64_t* slots)
{
    if (!slots || width == 0 || height == 0 || world == 0 || rank >= world) return fail(CRT_ERR_INVALID_ARG, "crt_shard_slots: bad arguments");
    *slots = make_shard(width, height, world).nslots;
    return CRT_OK;
}

int crt_scene_create(const crt_scene_desc* d, int device, crt_scene** out)
{
    if (!out) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: null output");
    *out = nullptr;
    int rc = validate_desc(d);
    if (rc != CRT_OK) return rc;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(CRT_ERR_NO_DEVICE, "crt_scene_create: no HIP device available");
    if (device < 0 || device >= n) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: device index out of range");
    crt_scene* sc = nullptr;
    try {
        sc = new crt_scene();
        sc->device = device;
        {
            // the HIP runtime starts with the first call that needs the device (context, the library's code objects): timed by itself
            // so that it is not booked on whatever happens to come first (it was the SAH build's first upload: "147 ms" of tree building)
            const auto t0 = std::chrono::steady_clock::now();
            HIP_CHECK(hipSetDevice(device));
            // (round 6: the first copy from / to pageable memory beyond the runtime's small-copy path sets up its staging -- 7.3 - 8.7 ms once
            // per process, 0.03 ms from then on, tools/copy_probe.cpp -- and was booked on the tree build's first upload and download; the 3 MB
            // download of the built tree paid another 8.6 ms after a 512 KB warm-up: the path beyond 1 MB.  Once per device and process.)
            static std::atomic<uint64_t> warmed{0};
            const uint64_t bit = 1ull << (device & 63);
            if (!(warmed.fetch_or(bit) & bit)) {
                std::vector<char> page(4u << 20, 0);
                DevBuf<char> warm;
                warm.ensure(page.size());
                HIP_CHECK(hipMemcpy(warm.p, page.data(), page.size(), hipMemcpyHostToDevice));
                HIP_CHECK(hipMemcpy(page.data(), warm.p, page.size(), hipMemcpyDeviceToHost));
            } else {
                HIP_CHECK(hipFree(nullptr)); // (the context, if this thread has none yet)
            }
            sc->accel.runtime_init_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        std::vector<float4> nodes, geo(d->n_tris * 3ull), mats(d->n_materials * 3ull), ltri(d->n_light_tris * 4ull);
        std::vector<int32_t> leaf_count, tri_mat(d->n_tris);
        int32_t root_fast = 0, root_exact = 0;
        AccelInfo ai;
        int depth = convert_bvh(*d, nodes, leaf_count, root_fast, root_exact, &ai);
        sc->accel.n_leaves = ai.n_leaves; sc->accel.n_nodes2 = ai.n_nodes2; sc->accel.sah_on_device = ai.on_device;
        sc->accel.sah_ms = ai.sah_ms; sc->accel.sah_device_ms = ai.sah_device_ms; sc->accel.index_splits = ai.index_splits;
        for (uint32_t i = 0; i < d->n_tris; i++) {
            const crt_triangle& t = d->tris[i];
            // e1 = v2 - v1, e2 = v3 - v1 as DeviceTriangle's constructor computes them (DeviceTriangle.cuh:27-28)
            float e1[3] = {t.v2[0] - t.v1[0], t.v2[1] - t.v1[1], t.v2[2] - t.v1[2]};
            float e2[3] = {t.v3[0] - t.v1[0], t.v3[1] - t.v1[1], t.v3[2] - t.v1[2]};
            geo[i * 3ull + 0] = make_float4(t.v1[0], t.v1[1], t.v1[2], e1[0]);
            geo[i * 3ull + 1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
            geo[i * 3ull + 2] = make_float4(e2[2], t.normal[0], t.normal[1], t.normal[2]);
            tri_mat[i] = t.material;
        }
        for (uint32_t i = 0; i < d->n_materials; i++) {
            const crt_material& m = d->materials[i];
            const float pi_f = (float)3.14159265358979323846;
            int32_t flags = (m.has_emit ? 1 : 0) | (m.mode == 1 ? 2 : 0);
            mats[i * 3ull + 0] = make_float4(m.kd[0] / pi_f, m.kd[1] / pi_f, m.kd[2] / pi_f, m.ns); // f_r = kd / float(M_PI) (Render.cuh:259)
            mats[i * 3ull + 1] = make_float4(m.kd[0], m.kd[1], m.kd[2], as_float(flags));
            mats[i * 3ull + 2] = make_float4(m.ke[0], m.ke[1], m.ke[2], 0.0f);
        }
        for (uint32_t i = 0; i < d->n_light_tris; i++) {
            const crt_triangle& t = d->light_tris[i];
            const crt_material& m = d->materials[t.material];
            ltri[i * 4ull + 0] = make_float4(t.v1[0], t.v1[1], t.v1[2], t.v2[0]);
            ltri[i * 4ull + 1] = make_float4(t.v2[1], t.v2[2], t.v3[0], t.v3[1]);
            ltri[i * 4ull + 2] = make_float4(t.v3[2], t.normal[0], t.normal[1], t.normal[2]);
            ltri[i * 4ull + 3] = make_float4(m.ke[0], m.ke[1], m.ke[2], t.area_of_obj);
        }
        // ---- k_mega3 layouts (crt_device.h): triangle-pair records per leaf, child boxes as (left, right) pairs ----
        std::vector<float4> leaf_geo, nodes3(nodes.size());
        std::vector<int32_t> rec_of_it(d->n_tris, -1);
        uint32_t max_leaf = 0;
        for (uint32_t i = 0; i < d->n_nodes; i++) {
            const crt_bvh_node& nn = d->nodes[i];
            if (!(nn.lc < 0 && nn.rc < 0)) continue;
            max_leaf = std::max(max_leaf, nn.n);
            rec_of_it[nn.it] = (int32_t)(leaf_geo.size() / 5);
            for (uint32_t k = 0; k < nn.n; k += 2) {
                const uint32_t ia = (uint32_t)nn.it + k, ib = k + 1 < nn.n ? ia + 1 : ia; // odd tail: the second lane repeats a and is masked
                const float4 a0 = geo[ia * 3ull], a1 = geo[ia * 3ull + 1], a2 = geo[ia * 3ull + 2];
                const float4 b0 = geo[ib * 3ull], b1 = geo[ib * 3ull + 1], b2 = geo[ib * 3ull + 2];
                leaf_geo.push_back(make_float4(a0.x, b0.x, a0.y, b0.y));  // v1.x, v1.y
                leaf_geo.push_back(make_float4(a0.z, b0.z, a0.w, b0.w));  // v1.z, e1.x
                leaf_geo.push_back(make_float4(a1.x, b1.x, a1.y, b1.y));  // e1.y, e1.z
                leaf_geo.push_back(make_float4(a1.z, b1.z, a1.w, b1.w));  // e2.x, e2.y
                leaf_geo.push_back(make_float4(a2.x, b2.x, as_float((int32_t)ia), as_float((int32_t)(nn.n - k)))); // e2.z, index, remaining
            }
        }
        auto ref3 = [&](int32_t r) -> int32_t { // old child ref -> k_mega3 child ref
            if (r >= 0) return r;
            return ~rec_of_it[(uint32_t)~r >> 4];
        };
        for (size_t q = 0; q * 4 < nodes.size(); q++) {
            const float4 a = nodes[q * 4], b = nodes[q * 4 + 1], c = nodes[q * 4 + 2], e = nodes[q * 4 + 3];
            int32_t lr, rr;
            std::memcpy(&lr, &a.w, 4); std::memcpy(&rr, &b.w, 4);
            nodes3[q * 4 + 0] = make_float4(a.x, c.x, a.y, c.y);
            nodes3[q * 4 + 1] = make_float4(a.z, c.z, b.x, e.x);
            nodes3[q * 4 + 2] = make_float4(b.y, e.y, b.z, e.z);
            nodes3[q * 4 + 3] = make_float4(as_float(ref3(lr)), as_float(ref3(rr)), 0.0f, 0.0f);
        }
        // ---- 4-wide tree for the rays with finite operands: the SAH tree collapsed (crt_device.h, nodes4) ----
        std::vector<float4> nodes4;
        struct Topo4 { bool used[4]; int32_t ref[4]; float lo[4][3], hi[4][3]; }; // the 4-wide tree as built below, by node: what the implicit-refs copy is made from
        std::vector<Topo4> topo;
        float coord_max = 0.0f; // largest |coordinate| of a box of the 4-wide tree (+inf if any is not finite): start_ray's overflow test
        int32_t root4 = ref3(root_fast);
        int depth4 = 1;
        if (root4 >= 0) {
            struct Child { float lo[3], hi[3]; int32_t ref; }; // ref: nodes3 index (>= 0) or leaf ref (< 0)
            auto children_of = [&](int32_t q, Child out[2]) {
                const float4 n0 = nodes3[q * 4ull], n1 = nodes3[q * 4ull + 1], n2 = nodes3[q * 4ull + 2], n3 = nodes3[q * 4ull + 3];
                out[0] = Child{{n0.x, n0.z, n1.x}, {n1.z, n2.x, n2.z}, 0};
                out[1] = Child{{n0.y, n0.w, n1.y}, {n1.w, n2.y, n2.w}, 0};
                std::memcpy(&out[0].ref, &n3.x, 4); std::memcpy(&out[1].ref, &n3.y, 4);
            };
            auto area = [](const Child& c) {
                const double dx = (double)c.hi[0] - c.lo[0], dy = (double)c.hi[1] - c.lo[1], dz = (double)c.hi[2] - c.lo[2];
                return dx * dy + dy * dz + dz * dx;
            };
            // ---- which binary nodes become 4-wide nodes: the collapse that minimises the summed area of the 4-wide nodes -- the SAH
            //      cost of the inner steps, every step costing the same whatever the number of children used (Ylitie et al. 2017,
            //      the leaves being fixed here): cost(q) = A(q) + min_k D(left, k) + D(right, 4 - k), D(c, j) = cheapest cover of
            //      subtree c by at most j children of the node above = min(D(c, j - 1), min_k D(c.left, k) + D(c.right, j - k)),
            //      D(c, 1) = cost(c), D(leaf, .) = 0.  Against the round-1 rule (CRT_COLLAPSE=greedy: open the child with the largest
            //      area until there are four): cornell-box 9 967 instead of 11 993 nodes, 4.62 instead of 5.02 inner steps per ray,
            //      C2 -3.7 %; veach-mis 842 / 974 nodes, 5.05 / 5.13 steps, same time.
            const char* collapse_env = std::getenv("CRT_COLLAPSE");
            const bool collapse_dp = !(collapse_env && std::strcmp(collapse_env, "greedy") == 0);
            struct DpNode { double D[3]; uint8_t kw, c2, c3; }; // D[j-1]; kw: left share of the node's own four; c2 / c3: choice for j = 2 / 3
            std::vector<DpNode> dpn;
            if (collapse_dp) {
                dpn.assign(nodes3.size() / 4, DpNode{{0, 0, 0}, 1, 0, 0});
                struct Fr { int32_t q; double area; int state; };
                std::vector<Fr> st;
                st.push_back(Fr{root4, 0.0, 0});
                auto Dof = [&](const Child& c, int j) { return c.ref < 0 ? 0.0 : dpn[(size_t)c.ref].D[j - 1]; };
                while (!st.empty()) {
                    Fr& f = st.back();
                    Child two[2];
                    children_of(f.q, two);
                    if (f.state == 0) {
                        f.state = 1;
                        const int32_t q = f.q; (void)q;
                        for (int i = 0; i < 2; i++)
                            if (two[i].ref >= 0) st.push_back(Fr{two[i].ref, area(two[i]), 0}); // (invalidates f: not used below)
                        continue;
                    }
                    DpNode& n = dpn[(size_t)f.q];
                    // the node as a 4-wide node: its own step + the cheapest forest of four under it
                    double best = 0.0; int bk = 1;
                    for (int k = 1; k <= 3; k++) {
                        const double v = Dof(two[0], k) + Dof(two[1], 4 - k);
                        if (k == 1 || v < best) { best = v; bk = k; }
                    }
                    n.kw = (uint8_t)bk;
                    n.D[0] = f.area + best;
                    const double open2 = Dof(two[0], 1) + Dof(two[1], 1);
                    n.c2 = open2 < n.D[0] ? 1 : 0;
                    n.D[1] = n.c2 ? open2 : n.D[0];
                    const double o12 = Dof(two[0], 1) + Dof(two[1], 2), o21 = Dof(two[0], 2) + Dof(two[1], 1);
                    n.c3 = 0; n.D[2] = n.D[1];
                    if (o12 < n.D[2]) { n.D[2] = o12; n.c3 = 1; }
                    if (o21 < n.D[2]) { n.D[2] = o21; n.c3 = 2; }
                    st.pop_back();
                }
            }
            // the (at most j) roots that cover the subtree of child c in the cheapest way
            std::vector<Child> cover;
            struct Ex { Child c; int j; };
            auto expand = [&](const Child& c0, int j0) {
                std::vector<Ex> ex;
                ex.push_back(Ex{c0, j0});
                while (!ex.empty()) {
                    Ex e = ex.back(); ex.pop_back();
                    if (e.c.ref < 0 || e.j == 1) { cover.push_back(e.c); continue; }
                    const DpNode& n = dpn[(size_t)e.c.ref];
                    const int choice = e.j == 2 ? (n.c2 ? 1 : 0) : n.c3;
                    if (choice == 0) { ex.push_back(Ex{e.c, e.j - 1}); continue; }
                    Child two[2];
                    children_of(e.c.ref, two);
                    ex.push_back(Ex{two[1], e.j - choice});
                    ex.push_back(Ex{two[0], choice});
                }
            };
            struct Todo { int32_t node2; int32_t slot; int depth; }; // slot: index of the BVH4 node to fill
            std::vector<Todo> todo;
            nodes4.resize(8);
            todo.push_back(Todo{root4, 0, 1});
            root4 = 0;
            const float qn_ = std::numeric_limits<float>::quiet_NaN();
            (void)qn_;
            for (size_t t = 0; t < todo.size(); t++) {
                const Todo cur = todo[t];
                depth4 = std::max(depth4, cur.depth + 1);
                std::vector<Child> ch(2);
                children_of(cur.node2, ch.data());
                if (collapse_dp) {
                    const Child l = ch[0], r = ch[1];
                    const int k = dpn[(size_t)cur.node2].kw;
                    cover.clear();
                    expand(l, k);
                    expand(r, 4 - k);
                    ch = cover;
                }
                while (!collapse_dp && ch.size() < 4) { // open the largest inner child
                    int best = -1;
                    double ba = -1.0;
                    for (size_t i = 0; i < ch.size(); i++)
                        if (ch[i].ref >= 0 && area(ch[i]) > ba) { ba = area(ch[i]); best = (int)i; }
                    if (best < 0) break;
                    Child two[2];
                    children_of(ch[best].ref, two);
                    ch[best] = two[0];
                    ch.push_back(two[1]);
                }
                int32_t refs[4];
                float lo[4][3], hi[4][3];
                if (topo.size() <= (size_t)cur.slot) topo.resize((size_t)cur.slot + 1);
                for (int i = 0; i < 4; i++) {
                    if (i < (int)ch.size()) {
                        for (int a = 0; a < 3; a++) { lo[i][a] = ch[i].lo[a]; hi[i][a] = ch[i].hi[a]; }
                        if (ch[i].ref >= 0) {
                            refs[i] = (int32_t)(nodes4.size() / 8);
                            nodes4.resize(nodes4.size() + 8);
                            todo.push_back(Todo{ch[i].ref, refs[i], cur.depth + 1});
                        } else refs[i] = ch[i].ref;
                    } else {
                        // empty slot: the inverted box (t_enter = +inf, t_exit = -inf whatever the signs of the direction)
                        for (int a = 0; a < 3; a++) { lo[i][a] = std::numeric_limits<float>::infinity(); hi[i][a] = -std::numeric_limits<float>::infinity(); }
                        refs[i] = ~0x7ffffff0; // (never followed)
                    }
                    Topo4& tp = topo[(size_t)cur.slot];
                    tp.used[i] = i < (int)ch.size();
                    tp.ref[i] = refs[i];
                    for (int a = 0; a < 3; a++) { tp.lo[i][a] = lo[i][a]; tp.hi[i][a] = hi[i][a]; }
                }
                for (int i = 0; i < (int)ch.size() && i < 4; i++)
                    for (int a = 0; a < 3; a++) {
                        const float m = std::max(std::fabs(lo[i][a]), std::fabs(hi[i][a]));
                        coord_max = (m <= FLT_MAX && coord_max <= FLT_MAX) ? std::max(coord_max, m) : std::numeric_limits<float>::infinity();
                    }
                float4* o = &nodes4[(size_t)cur.slot * 8];
                for (int a = 0; a < 3; a++) { // plane-major: [2a] = lo of axis a of the four children, [2a + 1] = hi
                    o[2 * a + 0] = make_float4(lo[0][a], lo[1][a], lo[2][a], lo[3][a]);
                    o[2 * a + 1] = make_float4(hi[0][a], hi[1][a], hi[2][a], hi[3][a]);
                }
                o[6] = make_float4(as_float(refs[0]), as_float(refs[1]), as_float(refs[2]), as_float(refs[3]));
                // the refs as the decoupled-leaves step wants them: a leaf as 0x80000000 | record << 8, ready to take the ray id
                // (records beyond 2^23 - 1 do not fit: crt_scene::dec_ok)
                auto dref = [](int32_t r) -> int32_t { return r >= 0 ? r : (int32_t)(0x80000000u | (((uint32_t)~r & 0x7fffffu) << 8)); };
                o[7] = make_float4(as_float(dref(refs[0])), as_float(dref(refs[1])), as_float(dref(refs[2])), as_float(dref(refs[3])));
            }
        }
        if (nodes4.empty()) nodes4.resize(8);
        const size_t n_nodes4 = nodes4.size() / 8;
        {
            // the empty node behind the tree (DevScene::empty4_off): four inverted boxes, refs that are never followed
            const float pinf_ = std::numeric_limits<float>::infinity();
            for (int a = 0; a < 3; a++) { nodes4.push_back(make_float4(pinf_, pinf_, pinf_, pinf_)); nodes4.push_back(make_float4(-pinf_, -pinf_, -pinf_, -pinf_)); }
            const float er = as_float(~0x7ffffff0);
            nodes4.push_back(make_float4(er, er, er, er)); nodes4.push_back(make_float4(er, er, er, er));
            sc->dev.empty4_off = (uint32_t)(n_nodes4 * 128);
        }
        // ---- the same tree WITHOUT its rows of refs (round 6): nodes4i, 6 x float4 (96 B) per node ----
        // What bounds k_mega3 is the number of divergent vector-memory instructions (one more 4-byte load per inner visit costs
        // the stand-in 2.1 % and veach-mis 6.7 %; 9 % fewer vector ALU instructions cost nothing), and a visit loads seven float4 -- six rows
        // of planes and the refs.  Here the refs are IMPLIED:
        //   * nodes are numbered breadth first in two ranges: [0, n_mixed) the nodes with an inner child, [n_mixed, n) the nodes whose
        //     children are all leaves ("fringe"); a node's children are ordered mixed, fringe, leaves, empty, so that its mixed children
        //     are fm, fm + 1, .. and its fringe children ff, ff + 1, ..;
        //   * leaf child k of node n is record 4 n + k of a SPARSE copy of the leaf records (leaf_geo_i; leaves of one record only:
        //     bvh_thresh_n <= 2, else the layout is not offered);
        //   * (fm, ff, number of mixed, number of fringe children) -- 36 bits -- live in the low 12 mantissa bits of the six planes of
        //     child 0 of a mixed node, which is an INNER child: the same 12 bits in the lo and in the hi plane of an axis (a ray reads them
        //     from the near plane whatever its direction), the planes moved OUTWARDS to the next value with those bits (a box grows by at
        //     most 2^-11 of its coordinates).  An inner box may be any superset (crt_trace.h); a leaf's box stays the reference's own, bit
        //     for bit -- which is why the bits can only live in an inner child, and why the fringe nodes are told apart by their number.
        // Offered when it applies (crt_scene::impl_ok); the decoupled-leaves kernels with 16-bit stack entries take it (CRT_IMPL=0: not).
        sc->impl_ok = false;
        if (root4 >= 0 && n_nodes4 <= 32768 && max_leaf <= 2 && topo.size() == n_nodes4) {
            const size_t n4 = n_nodes4;
            std::vector<uint8_t> mixed(n4, 0);
            for (size_t i = 0; i < n4; i++)
                for (int k = 0; k < 4; k++) if (topo[i].used[k] && topo[i].ref[k] >= 0) mixed[i] = 1;
            // The children of a class are ordered by "occupancy", descending: the summed area of the triangles below a child over its box's half
            // area (round 6).  The traversal takes the NEAREST hit inner child first whatever the slots; the slots decide
            // the order of the others (the loser of (0,1) is popped before the loser of the final, visit_front), the order of a visit's queue
            // entries, and the breadth-first numbers.  Against the order the collapse happens to leave: C2 75.86 -> 75.35 ms, veach-mis spp 256
            // 71.17 -> 70.43, inner visits per ray 4.496 -> 4.490 / 6.740 -> 6.685 (by box area instead: 75.61 / 70.41).  Visiting the any-hit
            // rays' children in this static order INSTEAD of nearest first was measured in both directions and loses (docs/experiments.md 6.10).
            // CRT_CHILD_ORDER=none|area: A/B hooks.
            std::vector<double> prio4(n4 * 4, 0.0);
            const char* co_ = std::getenv("CRT_CHILD_ORDER");
            if (!(co_ && co_[0] == 'n')) {
                const bool by_area = co_ && co_[0] == 'a';
                std::vector<double> tri_area_below(n4, -1.0);
                auto leaf_area = [&](int32_t ref) {
                    const size_t dense = (size_t)(~ref);
                    const int it = as_int(leaf_geo[dense * 5 + 4].z), cnt = as_int(leaf_geo[dense * 5 + 4].w);
                    double a = 0.0;
                    for (int q = 0; q < cnt; q++) {
                        const crt_triangle& tr = d->tris[(size_t)it + (size_t)q];
                        const double e1[3] = {(double)tr.v2[0] - tr.v1[0], (double)tr.v2[1] - tr.v1[1], (double)tr.v2[2] - tr.v1[2]};
                        const double e2[3] = {(double)tr.v3[0] - tr.v1[0], (double)tr.v3[1] - tr.v1[1], (double)tr.v3[2] - tr.v1[2]};
                        const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                        a += 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
                    }
                    return a;
                };
                std::vector<std::pair<int32_t, int>> st_(1, std::make_pair(0, 0)); // post-order over the 4-wide tree
                while (!st_.empty()) {
                    const int32_t i = st_.back().first;
                    int32_t next_ = -1;
                    for (int k = 0; k < 4 && next_ < 0; k++)
                        if (topo[(size_t)i].used[k] && topo[(size_t)i].ref[k] >= 0 && tri_area_below[(size_t)topo[(size_t)i].ref[k]] < 0.0) next_ = topo[(size_t)i].ref[k];
                    if (next_ >= 0) { st_.push_back(std::make_pair(next_, 0)); continue; }
                    double sum = 0.0;
                    for (int c = 0; c < 4; c++) {
                        const Topo4& t = topo[(size_t)i];
                        if (!t.used[c]) continue;
                        const double below = t.ref[c] < 0 ? leaf_area(t.ref[c]) : tri_area_below[(size_t)t.ref[c]];
                        sum += below;
                        const double ex = (double)t.hi[c][0] - t.lo[c][0], ey = (double)t.hi[c][1] - t.lo[c][1], ez = (double)t.hi[c][2] - t.lo[c][2];
                        const double ha = ex * ey + ey * ez + ez * ex;
                        prio4[(size_t)i * 4 + (size_t)c] = by_area ? ha : (ha > 0.0 ? below / ha : 1e30);
                    }
                    tri_area_below[(size_t)i] = sum;
                    st_.pop_back();
                }
            }
            // children order: mixed, fringe, leaves, empty; then breadth-first numbers in the two ranges
            std::vector<std::array<int, 4>> order(n4);
            for (size_t i = 0; i < n4; i++) {
                int o = 0;
                for (int pass = 0; pass < 4; pass++) {
                    const int o0 = o;
                    for (int k = 0; k < 4; k++) {
                        const Topo4& t = topo[i];
                        const int cls = !t.used[k] ? 3 : (t.ref[k] < 0 ? 2 : (mixed[(size_t)t.ref[k]] ? 0 : 1));
                        if (cls == pass) order[i][o++] = k;
                    }
                    std::stable_sort(order[i].begin() + o0, order[i].begin() + o, [&](int a, int b) { return prio4[i * 4 + (size_t)a] > prio4[i * 4 + (size_t)b]; });
                }
            }
            uint32_t n_mixed = 0;
            for (size_t i = 0; i < n4; i++) n_mixed += mixed[i];
            std::vector<int32_t> newid(n4, -1);
            {
                uint32_t cm_ = 0, cf_ = n_mixed;
                std::vector<int32_t> bfs;
                bfs.push_back(0);
                newid[0] = mixed[0] ? (int32_t)cm_++ : (int32_t)cf_++;
                for (size_t t = 0; t < bfs.size(); t++) {
                    const int32_t i = bfs[t];
                    for (int o = 0; o < 4; o++) {
                        const int k = order[(size_t)i][o];
                        if (!topo[(size_t)i].used[k] || topo[(size_t)i].ref[k] < 0) continue;
                        const int32_t c = topo[(size_t)i].ref[k];
                        newid[(size_t)c] = mixed[(size_t)c] ? (int32_t)cm_++ : (int32_t)cf_++;
                        bfs.push_back(c);
                    }
                }
            }
            auto raw = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
            auto unraw = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
            bool ok = true;
            // the nearest value <= f (down) / >= f (up) whose low 12 bits are `chunk`
            auto with_bits = [&](float f, uint32_t chunk, bool up) -> float {
                if (!(std::fabs(f) <= FLT_MAX)) { ok = false; return f; }
                const uint32_t r = raw(f);
                const bool neg = (r >> 31) != 0;
                uint32_t m = r & 0x7fffffffu; // magnitude: grows with |f|
                const bool grow = neg ? !up : up; // does the magnitude have to grow?
                uint32_t c = (m & ~0xfffu) | chunk;
                if (grow) { if (c < m) c += 0x1000u; }
                else if (c > m) {
                    if (c >= 0x1000u) c -= 0x1000u;
                    else { // |f| below the smallest magnitude with these bits: cross zero -- the smallest magnitude of the other sign
                        const uint32_t other = chunk | (neg ? 0u : 0x80000000u);
                        return unraw(other);
                    }
                }
                if (c >= 0x7f800000u) { ok = false; return f; }
                return unraw(c | (neg ? 0x80000000u : 0u));
            };
            std::vector<float4> n4i((n4 + 1) * (size_t)NODE4I_F4, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            const size_t n_rec_i = (n4 + 1) * 4;
            std::vector<float4> lgi(n_rec_i * 5, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            std::vector<int32_t> rec_map(leaf_geo.size() / 5, 0);
            float cmax_i = coord_max;
            for (size_t i = 0; i < n4 && ok; i++) {
                const Topo4& t = topo[i];
                const size_t ni = (size_t)newid[i];
                float lo[4][3], hi[4][3];
                uint32_t n_m = 0, n_f = 0, fm = 0, ff = 0;
                for (int o = 0; o < 4; o++) {
                    const int k = order[i][o];
                    for (int a = 0; a < 3; a++) { lo[o][a] = t.lo[k][a]; hi[o][a] = t.hi[k][a]; }
                    if (t.used[k] && t.ref[k] >= 0) {
                        const uint32_t id = (uint32_t)newid[(size_t)t.ref[k]];
                        if (mixed[(size_t)t.ref[k]]) { if (n_m++ == 0) fm = id; } else { if (n_f++ == 0) ff = id; }
                    } else if (t.used[k]) { // a leaf of one record: its copy at 4 n + o
                        const size_t dense = (size_t)(~t.ref[k]);
                        const size_t sparse = ni * 4 + (size_t)o;
                        for (int q = 0; q < 5; q++) lgi[sparse * 5 + (size_t)q] = leaf_geo[dense * 5 + (size_t)q];
                        rec_map[dense] = (int32_t)sparse;
                    }
                }
                if (mixed[i]) {
                    const uint32_t chunk[3] = {fm & 0xfffu, ((fm >> 12) & 7u) | ((ff & 0x1ffu) << 3), ((ff >> 9) & 63u) | (n_m << 6) | (n_f << 9)};
                    for (int a = 0; a < 3; a++) {
                        lo[0][a] = with_bits(lo[0][a], chunk[a], false);
                        hi[0][a] = with_bits(hi[0][a], chunk[a], true);
                        const float m = std::max(std::fabs(lo[0][a]), std::fabs(hi[0][a]));
                        cmax_i = std::max(cmax_i, m);
                    }
                }
                float4* o6 = &n4i[ni * (size_t)NODE4I_F4];
                for (int a = 0; a < 3; a++) {
                    o6[2 * a + 0] = make_float4(lo[0][a], lo[1][a], lo[2][a], lo[3][a]);
                    o6[2 * a + 1] = make_float4(hi[0][a], hi[1][a], hi[2][a], hi[3][a]);
                }
            }
            if (ok && cmax_i <= FLT_MAX) {
                const float pinf_ = std::numeric_limits<float>::infinity();
                for (int a = 0; a < 3; a++) { n4i[n4 * (size_t)NODE4I_F4 + 2 * (size_t)a] = make_float4(pinf_, pinf_, pinf_, pinf_); n4i[n4 * (size_t)NODE4I_F4 + 2 * (size_t)a + 1] = make_float4(-pinf_, -pinf_, -pinf_, -pinf_); }
                sc->nodes4i.upload(n4i); sc->leaf_geo_i.upload(lgi); sc->rec_map.upload(rec_map);
                sc->dev.nodes4i = sc->nodes4i.p; sc->dev.leaf_geo_i = sc->leaf_geo_i.p; sc->dev.rec_map = sc->rec_map.p;
                sc->dev.n_mixed4i = n_mixed;
                sc->dev.root4i = newid[0];
                sc->dev.empty4i_off = (uint32_t)(n4 * NODE4I_F4 * 16);
                coord_max = cmax_i; // (start_ray's overflow test covers both copies of the tree)
                sc->impl_ok = true;
            }
        }
        sc->nodes4.upload(nodes4);
        sc->dev.nodes4 = sc->nodes4.p;
        sc->dev.root4 = root4;
        sc->dev.coord_max = coord_max;
        sc->depth4 = depth4;
        sc->accel.n_nodes4 = (uint32_t)n_nodes4; sc->accel.depth2 = (uint32_t)depth; sc->accel.depth4 = (uint32_t)depth4;
        sc->ref16_ok = n_nodes4 <= 32768 && leaf_geo.size() / 5 <= 32768; // node refs 0 .. 32767, leaf refs ~0 .. ~32767
        sc->ref16_inner_ok = n_nodes4 <= 32768;
        sc->dec_ok = leaf_geo.size() / 5 <= (size_t)LEAF_REC_MAX + 1;
        sc->accel.layout_caps = (sc->ref16_ok ? 1u : 0u) | (sc->ref16_inner_ok ? 2u : 0u) | (sc->dec_ok ? 4u : 0u) | (sc->impl_ok ? 8u : 0u);
        std::vector<float4> tri_nm(d->n_tris);
        for (uint32_t i = 0; i < d->n_tris; i++) {
            const crt_material& m = d->materials[d->tris[i].material];
            const int32_t w = (int32_t)((uint32_t)d->tris[i].material | (m.has_emit ? 1u << 30 : 0u) | (m.mode == 1 ? 1u << 31 : 0u)); // TNM_* (crt_device.h)
            tri_nm[i] = make_float4(d->tris[i].normal[0], d->tris[i].normal[1], d->tris[i].normal[2], as_float(w));
        }
        sc->nodes3.upload(nodes3); sc->leaf_geo.upload(leaf_geo); sc->tri_nm.upload(tri_nm);
        sc->dev.tri_nm = sc->tri_nm.p;
        sc->max_leaf = max_leaf;
        sc->dev.nodes3 = sc->nodes3.p; sc->dev.leaf_geo = sc->leaf_geo.p;
        sc->dev.root3_fast = ref3(root_fast); sc->dev.root3_exact = ref3(root_exact);
        std::vector<uint4> lights(d->n_lights);
        for (uint32_t i = 0; i < d->n_lights; i++) {
            FastDiv fd = make_fastdiv(d->lights[i].count);
            lights[i] = make_uint4(d->lights[i].first_tri, d->lights[i].count, fd.m, fd.sh);
        }
        sc->nodes.upload(nodes); sc->tri_geo.upload(geo); sc->tri_mat.upload(tri_mat); sc->mats.upload(mats);
        sc->ltri.upload(ltri); sc->lights.upload(lights); sc->leaf_count.upload(leaf_count);
        sc->counters.alloc((size_t)CNT_SHARDS * CNT_STRIDE);
        sc->item_next.alloc((size_t)1024 * ITEM_STRIDE); // (a commit-ring launch has up to 1 024 cursor shards)
        sc->slot_next[0].alloc((size_t)SLOT_SHARDS * SLOT_STRIDE);
        sc->slot_next[1].alloc((size_t)SLOT_SHARDS * SLOT_STRIDE);
        HIP_CHECK(hipStreamCreateWithFlags(&sc->aux_stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&sc->ev_fork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&sc->ev_join, hipEventDisableTiming));
        HIP_CHECK(hipEventCreate(&sc->ev_k0));
        HIP_CHECK(hipEventCreate(&sc->ev_k1));
        {
            hipDeviceProp_t prop;
            HIP_CHECK(hipGetDeviceProperties(&prop, device));
            sc->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        HIP_CHECK(hipHostMalloc((void**)&sc->h_counters, (size_t)CNT_SHARDS * CNT_STRIDE * sizeof(unsigned long long), hipHostMallocDefault));
        sc->dev.nodes = sc->nodes.p; sc->dev.tri_geo = sc->tri_geo.p; sc->dev.tri_mat = sc->tri_mat.p; sc->dev.mats = sc->mats.p;
        sc->dev.ltri = sc->ltri.p; sc->dev.lights = sc->lights.p; sc->dev.leaf_count = sc->leaf_count.p;
        sc->dev.root_fast = root_fast; sc->dev.root_exact = root_exact; sc->dev.n_lights = (int32_t)d->n_lights;
        sc->n_tris = d->n_tris;
        sc->n_mats = d->n_materials;
        // Both traversal modes hold at most one pending sibling per tree level.
        sc->stack_cap = std::max(depth + 2, 3 * sc->depth4 + 2); // BVH2: one pending sibling per level; BVH4: up to three
        *out = sc;
        return CRT_OK;
    } catch (const HipFail& f) {
        delete sc;
        return fail_hip(f);
    } catch (const std::bad_alloc&) {
        delete sc;
        return fail(CRT_ERR_OOM, "crt_scene_create: out of host memory");
    }
}

int crt_scene_accel_info(crt_scene* sc, crt_accel_info* out)
{
    if (!sc || !out) return fail(CRT_ERR_INVALID_ARG, "crt_scene_accel_info: null argument");
    *out = sc->accel;
    return CRT_OK;
}

int crt_scene_destroy(crt_scene* sc)
{
    if (!sc) return CRT_OK;
    (void)hipSetDevice(sc->device);
    delete sc;
    return CRT_OK;
}

int crt_render_device(crt_scene* sc, const crt_camera* cam, const crt_params* prm, void* d_rgb, void* d_mean, void* stream, crt_stats* stats)
{
    return render_impl(sc, cam, prm, d_rgb, d_mean, (hipStream_t)stream, stats);
}

int crt_render(crt_scene* sc, const crt_camera* cam, const crt_params* prm, uint8_t* out_rgb, float* out_mean, crt_stats* stats)
{
    if (!sc || !prm || !out_rgb) return fail(CRT_ERR_INVALID_ARG, "crt_render: null argument");
    if (prm->world == 0 || prm->rank >= prm->world || prm->width == 0 || prm->height == 0) return fail(CRT_ERR_INVALID_ARG, "crt_render: bad shard or size");
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        const bool tiled = (prm->flags & CRT_FLAG_TILED_OUTPUT) != 0;
        uint64_t npix = tiled ? make_shard(prm->width, prm->height, prm->world).nslots : (uint64_t)prm->width * prm->height;
        DevBuf<uint8_t> d_rgb;
        DevBuf<float> d_mean;
        d_rgb.alloc(npix * 3);
        if (out_mean) d_mean.alloc(npix * 3);
        int rc = render_impl(sc, cam, prm, d_rgb.p, out_mean ? d_mean.p : nullptr, nullptr, stats);
        if (rc != CRT_OK) return rc;
        HIP_CHECK(hipDeviceSynchronize()); // Render.cuh:440
        HIP_CHECK(hipMemcpy(out_rgb, d_rgb.p, npix * 3, hipMemcpyDeviceToHost)); // Render.cuh:464
        if (out_mean) HIP_CHECK(hipMemcpy(out_mean, d_mean.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_render_range_device(crt_scene* sc, const crt_camera* cam, const crt_params* prm, uint32_t sample_begin, uint32_t sample_count,
                            void* d_rgb, void* d_mean, void* stream, crt_stats* stats)
{
    if (sample_count == 0xffffffffu) return fail(CRT_ERR_INVALID_ARG, "crt_render_range: bad sample count");
    return render_impl(sc, cam, prm, d_rgb, d_mean, (hipStream_t)stream, stats, sample_begin, sample_count);
}

int crt_render_range(crt_scene* sc, const crt_camera* cam, const crt_params* prm, uint32_t sample_begin, uint32_t sample_count,
                     uint8_t* out_rgb, float* out_mean, crt_stats* stats)
{
    if (!sc || !prm) return fail(CRT_ERR_INVALID_ARG, "crt_render_range: null argument");
    if (prm->world == 0 || prm->rank >= prm->world || prm->width == 0 || prm->height == 0) return fail(CRT_ERR_INVALID_ARG, "crt_render_range: bad shard or size");
    if (sample_count == 0xffffffffu || sample_count == 0 || (uint64_t)sample_begin + sample_count > prm->spp)
        return fail(CRT_ERR_INVALID_ARG, "crt_render_range: sample range outside [0, spp)");
    const bool last = sample_begin + sample_count == prm->spp;
    if (last && !out_rgb) return fail(CRT_ERR_INVALID_ARG, "crt_render_range: the range that ends at spp needs a frame buffer");
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        const bool tiled = (prm->flags & CRT_FLAG_TILED_OUTPUT) != 0;
        uint64_t npix = tiled ? make_shard(prm->width, prm->height, prm->world).nslots : (uint64_t)prm->width * prm->height;
        DevBuf<uint8_t> d_rgb;
        DevBuf<float> d_mean;
        if (last) d_rgb.alloc(npix * 3);
        if (last && out_mean) d_mean.alloc(npix * 3);
        int rc = render_impl(sc, cam, prm, last ? d_rgb.p : nullptr, last && out_mean ? d_mean.p : nullptr, nullptr, stats, sample_begin, sample_count);
        if (rc != CRT_OK) return rc;
        HIP_CHECK(hipDeviceSynchronize());
        if (last) {
            HIP_CHECK(hipMemcpy(out_rgb, d_rgb.p, npix * 3, hipMemcpyDeviceToHost));
            if (out_mean) HIP_CHECK(hipMemcpy(out_mean, d_mean.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
        }
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_last_launch_ms(crt_scene* sc, float* ms, uint32_t* launches)
{
    if (!sc || !ms) return fail(CRT_ERR_INVALID_ARG, "crt_last_launch_ms: null argument");
    if (sc->last_launches == 0) return fail(CRT_ERR_INVALID_ARG, "crt_last_launch_ms: no frame has been rendered by the megakernel on this handle");
    hipError_t e = hipEventElapsedTime(ms, sc->ev_k0, sc->ev_k1);
    if (e != hipSuccess) return fail(CRT_ERR_HIP, std::string("crt_last_launch_ms: hipEventElapsedTime: ") + hipGetErrorString(e) + " (synchronize the stream first)");
    if (launches) *launches = sc->last_launches;
    return CRT_OK;
}

int crt_radiance_storage(crt_scene* sc, uint64_t* bytes, uint32_t* ring_samples)
{
    if (!sc || !bytes) return fail(CRT_ERR_INVALID_ARG, "crt_radiance_storage: null argument");
    *bytes = sc->last_radiance_bytes;
    if (ring_samples) *ring_samples = sc->last_ring_samples;
    return CRT_OK;
}

int crt_preview_device(crt_scene* sc, void* d_rgb, void* d_mean, void* stream, uint32_t* samples_done)
{
    if (!sc || !d_rgb) return fail(CRT_ERR_INVALID_ARG, "crt_preview: null argument");
    if (sc->acc.samples == 0) return fail(CRT_ERR_INVALID_ARG, "crt_preview: no progressive render in flight (submit a range that ends before spp first)");
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        Shard sh = make_shard(sc->acc.width, sc->acc.height, sc->acc.world);
        AParams A;
        std::memset(&A, 0, sizeof(A));
        A.width = sc->acc.width; A.height = sc->acc.height; A.spp = sc->acc.spp;
        A.rank = sc->acc.rank; A.world = sc->acc.world; A.tiles_x = sh.tiles_x; A.n_tiles = sh.n_tiles;
        A.nslots = sh.nslots; A.tiled_output = sc->acc.tiled;
        A.accum = sc->accum.p;
        A.out_rgb = (uint8_t*)d_rgb; A.out_mean = (float*)d_mean;
        const float scale = (float)sc->acc.spp / (float)sc->acc.samples;
        launch_preview(A, scale, (hipStream_t)stream);
        HIP_CHECK(hipGetLastError());
        if (samples_done) *samples_done = sc->acc.samples;
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_preview(crt_scene* sc, uint8_t* out_rgb, float* out_mean, uint32_t* samples_done)
{
    if (!sc || !out_rgb) return fail(CRT_ERR_INVALID_ARG, "crt_preview: null argument");
    if (sc->acc.samples == 0) return fail(CRT_ERR_INVALID_ARG, "crt_preview: no progressive render in flight (submit a range that ends before spp first)");
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        const uint64_t npix = sc->acc.tiled ? make_shard(sc->acc.width, sc->acc.height, sc->acc.world).nslots : (uint64_t)sc->acc.width * sc->acc.height;
        DevBuf<uint8_t> d_rgb;
        DevBuf<float> d_mean;
        d_rgb.alloc(npix * 3);
        if (out_mean) d_mean.alloc(npix * 3);
        if (sc->acc.tiled) HIP_CHECK(hipMemset(d_rgb.p, 0, npix * 3));
        int rc = crt_preview_device(sc, d_rgb.p, out_mean ? d_mean.p : nullptr, nullptr, samples_done);
        if (rc != CRT_OK) return rc;
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(out_rgb, d_rgb.p, npix * 3, hipMemcpyDeviceToHost));
        if (out_mean) HIP_CHECK(hipMemcpy(out_mean, d_mean.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_intersect(crt_scene* sc, uint32_t n, const float* origins, const float* dirs, uint32_t traversal, int32_t* out_tri, float* out_t)
{
    if (!sc || !origins || !dirs || !out_tri || !out_t) return fail(CRT_ERR_INVALID_ARG, "crt_intersect: null argument");
    const bool raw_dir = (traversal & CRT_INTERSECT_RAW_DIRECTIONS) != 0;
    const bool force_exact = (traversal & CRT_INTERSECT_FORCE_EXACT) != 0;
    const bool any_hit = (traversal & CRT_INTERSECT_VISIBILITY) != 0;
    traversal &= ~(uint32_t)(CRT_INTERSECT_RAW_DIRECTIONS | CRT_INTERSECT_FORCE_EXACT | CRT_INTERSECT_VISIBILITY);
    if (traversal != CRT_TRAVERSAL_FAST && traversal != CRT_TRAVERSAL_REFERENCE && traversal != CRT_TRAVERSAL_EXACT)
        return fail(CRT_ERR_INVALID_ARG, "crt_intersect: unknown traversal mode");
    if (n == 0) return CRT_OK;
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        DevBuf<float> o, d, lim;
        o.alloc(n * 3ull); d.alloc(n * 3ull);
        HIP_CHECK(hipMemcpy(o.p, origins, n * 12ull, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d.p, dirs, n * 12ull, hipMemcpyHostToDevice));
        if (any_hit) { lim.alloc(n); HIP_CHECK(hipMemcpy(lim.p, out_t, n * 4ull, hipMemcpyHostToDevice)); }
        // blocked() of Render.cuh:19-27 from a finished visibility ray (limit = out_t[i] on entry): REFERENCE compares the closest
        // hit, FAST recorded a hit only if it passes the comparison (shadow_blocked)
        const bool reference_mode = traversal == CRT_TRAVERSAL_REFERENCE;
        auto answer = [&](uint32_t i, float T, int32_t tri) {
            if (!any_hit) { out_t[i] = T; out_tri[i] = tri; return; }
            const float tl = out_t[i];
            const bool blocked = reference_mode ? (tl - T > CRT_EPSILON) : (tri >= 0 || tl - FLT_MAX > CRT_EPSILON);
            out_t[i] = blocked ? 1.0f : 0.0f;
            out_tri[i] = blocked ? tri : -1;
        };
        sc->p_ro.ensure(n); sc->p_rd.ensure(n); sc->p_res.ensure(n);
        Pool pool;
        std::memset(&pool, 0, sizeof(pool));
        pool.ro = sc->p_ro.p; pool.rd = sc->p_rd.p; pool.res = sc->p_res.p; pool.n = n;
        launch_fill_rays(pool, n, o.p, d.p, raw_dir, any_hit ? lim.p : (const float*)nullptr);
        HIP_CHECK(hipGetLastError());
        if (choose_pipeline(sc) == 4) {
            // the rays walk the traversal phases of the render kernel itself (k_mega3 in query form: work item = ray)
            const bool reference = traversal == CRT_TRAVERSAL_REFERENCE;
            int per_cu = 1;
            const bool exact = traversal == CRT_TRAVERSAL_EXACT;
            const int mode3 = reference ? 1 : exact ? 2 : 0;
            const bool dec = use_dec(sc, mode3);
            const bool r16 = use_ref16(sc, mode3, dec);
            const Mega3Kernel kern3 = mega3_kernel(mode3, false, false, true, r16, false, dec, use_impl(sc, dec, r16));
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern3, 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
            const uint32_t pool_p = mega3_pool_p(dec, false);
            const uint32_t blocks = std::min<uint32_t>((n + pool_p - 1) / pool_p, (uint32_t)(sc->n_cus * per_cu));
            const uint32_t lanes = blocks * pool_p;
            sc->p_la.ensure(lanes); sc->p_id.ensure(lanes); sc->L.ensure(n);
            sc->spill[0].ensure((size_t)(r16 ? std::max(1, sc->stack_cap) : std::max(1, sc->stack_cap - mega3_lds_levels(dec, r16))) * lanes);
            MParams3 M3;
            std::memset(&M3, 0, sizeof(M3));
            LParams& P = M3.M.P;
            P.sc = sc->dev;
            P.pool.la = sc->p_la.p; P.pool.id = sc->p_id.p; P.pool.n = lanes;
            P.n_items = n;
            P.items_per_shard = ((n + ITEM_SHARDS - 1) / ITEM_SHARDS + 63u) & ~63u;
            P.item_next = sc->item_next.p; P.L = sc->L.p; P.counters = sc->counters.p;
            P.q_o = sc->p_ro.p; P.q_d = sc->p_rd.p;
            P.nslots = 1; P.nslots_div = make_fastdiv(1); P.tiles_x = 1; P.tiles_x_div = make_fastdiv(1); P.lsn_div = make_fastdiv(1);
            M3.M.sc = sc->dev; M3.M.counters = sc->counters.p; M3.M.spill_stride = lanes; M3.M.stack_cap = mega3_lds_levels(dec, r16);
            M3.spill = (int*)sc->spill[0].p;
            M3.force_exact = force_exact ? 1u : 0u;
            HIP_CHECK(hipMemsetAsync(sc->item_next.p, 0, (size_t)ITEM_SHARDS * ITEM_STRIDE * sizeof(unsigned int), nullptr));
            hipLaunchKernelGGL(kern3, dim3(blocks), dim3(64), 0, nullptr, M3);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipDeviceSynchronize());
            std::vector<float4> res(n);
            HIP_CHECK(hipMemcpy(res.data(), sc->L.p, n * sizeof(float4), hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < n; i++) {
                int32_t tri;
                std::memcpy(&tri, &res[i].y, 4);
                answer(i, res[i].x, tri);
            }
            return CRT_OK;
        }
        TraceSetup TS = make_trace_setup(sc, pool, traversal, false);
        launch_trace_pass(sc, TS, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        std::vector<float2> res(n);
        HIP_CHECK(hipMemcpy(res.data(), sc->p_res.p, n * sizeof(float2), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; i++) {
            int32_t tri;
            std::memcpy(&tri, &res[i].y, 4);
            answer(i, res[i].x, tri);
        }
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_device_math(int device, const char* fn, uint32_t n, const float* a, const float* b, float* out)
{
    if (!fn || !a || !out) return fail(CRT_ERR_INVALID_ARG, "crt_device_math: null argument");
    static const char* names[] = {"sin", "cos", "tan", "acos", "atan2", "exp", "log10", "pow", "uniform", "sincos_s", "sincos_c", "div_short", "div_short_bounded"};
    int id = -1;
    for (int i = 0; i < 13; i++)
        if (std::strcmp(fn, names[i]) == 0) id = i;
    if (id < 0) return fail(CRT_ERR_INVALID_ARG, std::string("crt_device_math: unknown function ") + fn);
    if (n == 0) return CRT_OK;
    try {
        HIP_CHECK(hipSetDevice(device));
        DevBuf<float> da, db, dout;
        da.alloc(n); dout.alloc(n);
        HIP_CHECK(hipMemcpy(da.p, a, n * 4ull, hipMemcpyHostToDevice));
        if (b) { db.alloc(n); HIP_CHECK(hipMemcpy(db.p, b, n * 4ull, hipMemcpyHostToDevice)); }
        launch_math(id, n, da.p, b ? db.p : nullptr, dout.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(out, dout.p, n * 4ull, hipMemcpyDeviceToHost));
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_device_philox(int device, uint32_t n, const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4)
{
    if (!ctr4 || !key2 || !out4) return fail(CRT_ERR_INVALID_ARG, "crt_device_philox: null argument");
    if (n == 0) return CRT_OK;
    try {
        HIP_CHECK(hipSetDevice(device));
        DevBuf<uint32_t> c, k, o;
        c.alloc(n * 4ull); k.alloc(n * 2ull); o.alloc(n * 4ull);
        HIP_CHECK(hipMemcpy(c.p, ctr4, n * 16ull, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(k.p, key2, n * 8ull, hipMemcpyHostToDevice));
        launch_philox(n, c.p, k.p, o.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(out4, o.p, n * 16ull, hipMemcpyDeviceToHost));
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_device_rcp_check(int device, uint64_t* mismatches, uint64_t* outside)
{
    if (!mismatches || !outside) return fail(CRT_ERR_INVALID_ARG, "crt_device_rcp_check: null argument");
    try {
        HIP_CHECK(hipSetDevice(device));
        DevBuf<unsigned long long> c;
        c.alloc(2);
        HIP_CHECK(hipMemset(c.p, 0, 2 * sizeof(unsigned long long)));
        launch_rcp_check(c.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        unsigned long long h[2];
        HIP_CHECK(hipMemcpy(h, c.p, sizeof(h), hipMemcpyDeviceToHost));
        *mismatches = h[0]; *outside = h[1];
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

} // extern "C"

