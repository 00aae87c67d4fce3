// tools/wide_probe.cpp -- what an 8-wide node would change in the walk of k_mega3's decoupled inner step, counted on the host.
//
// The production tree is the binned-SAH tree over the reference's leaves (crt_accel.h: build_sah + one pass of optimize_sah) collapsed to W = 4
// children per node by the dynamic programme of crt_render.hip (minimum summed area of the wide nodes).  This program builds the SAME binary tree
// from a scene file (host layer of libcrt.so, no GPU), collapses it with the same dynamic programme for W = 4 and W = 8, and walks both with the
// same rays -- camera rays, one uniform-hemisphere bounce ray per primary hit and one ray towards a light sample per hit (the kernel's three
// kinds) -- counting, per ray: wide nodes visited, child boxes tested (W per visit), leaf boxes hit.  CRT_TRAVERSAL_EXACT visits a node iff its
// box and all its ancestors' boxes pass hit_AABB, whatever the order, so for rays that are not any-hit rays these counts are the kernel's
// (and an upper bound for the any-hit ones); the leaf boxes hit are the same for every W -- a check.
// build: g++ -O2 -std=c++17 -Iinclude -Icudaraytracing_amd/csrc tools/wide_probe.cpp -Lcudaraytracing_amd/lib -lcrt -Wl,-rpath,$PWD/cudaraytracing_amd/lib -o /tmp/wide_probe
// usage: /tmp/wide_probe scenes/cornell-box/config.json [primary rays, default 20000]
#include "crt.h"
#include "crt_accel.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <random>
#include <string>

using crtaccel::Box;
struct V3 { float x, y, z; };
static V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 mul(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V3 unit(V3 a) { const float l = std::sqrt(dot(a, a)); return {a.x / l, a.y / l, a.z / l}; }

// hit_AABB (DeviceBVH.cuh:87-126) for finite operands
static bool hit_box(const Box& b, V3 o, V3 inv, V3 d)
{
    float tx0 = (b.lo[0] - o.x) * inv.x, tx1 = (b.hi[0] - o.x) * inv.x; if (d.x < 0) std::swap(tx0, tx1);
    float ty0 = (b.lo[1] - o.y) * inv.y, ty1 = (b.hi[1] - o.y) * inv.y; if (d.y < 0) std::swap(ty0, ty1);
    float tz0 = (b.lo[2] - o.z) * inv.z, tz1 = (b.hi[2] - o.z) * inv.z; if (d.z < 0) std::swap(tz0, tz1);
    const float e = std::max(std::max(tx0, ty0), tz0), x = std::min(std::min(tx1, ty1), tz1);
    return e <= x + 0.00001f && x >= 0;
}

struct Wide { std::vector<Box> box; std::vector<int32_t> ref; }; // up to W children: ref >= 0 wide node, < 0 leaf
struct WideTree { int W; std::vector<Wide> nodes; double summed_area = 0; };

// the dynamic programme of crt_render.hip, for any W: which binary nodes become wide nodes so that the summed area of the wide nodes is least
static WideTree collapse(const std::vector<crtaccel::Node>& bin, int W)
{
    const int A = (int)bin.size();
    auto area_of = [&](int q, int s) { return bin[q].box[s].half_area(); };
    // D[q][j-1], j = 1 .. W-1: cheapest cover of the subtree of binary node q by at most j wide-node children; choice[q][j-1]: 0 = as with j-1, k = split k | j-k
    std::vector<std::vector<double>> D(A, std::vector<double>(W - 1, 0.0));
    std::vector<std::vector<int>> choice(A, std::vector<int>(W - 1, 0));
    std::vector<int> kw(A, 1);
    std::vector<double> own_area(A, 0.0);
    own_area[0] = [&] { Box b = bin[0].box[0]; b.grow(bin[0].box[1]); return b.half_area(); }();
    for (int q = 0; q < A; q++) for (int s = 0; s < 2; s++) if (bin[q].child[s] >= 0) own_area[bin[q].child[s]] = area_of(q, s);
    auto Dof = [&](int32_t c, int j) { return c < 0 ? 0.0 : D[c][j - 1]; };
    for (int q = A - 1; q >= 0; q--) { // breadth-first numbering: children have larger indices
        const int32_t l = bin[q].child[0], r = bin[q].child[1];
        double best = 0; int bk = 1;
        for (int k = 1; k <= W - 1; k++) { const double v = Dof(l, k) + Dof(r, W - k); if (k == 1 || v < best) { best = v; bk = k; } }
        kw[q] = bk;
        D[q][0] = own_area[q] + best;
        for (int j = 2; j <= W - 1; j++) {
            D[q][j - 1] = D[q][j - 2]; choice[q][j - 1] = 0;
            for (int k = 1; k <= j - 1; k++) { const double v = Dof(l, k) + Dof(r, j - k); if (v < D[q][j - 1]) { D[q][j - 1] = v; choice[q][j - 1] = k; } }
        }
    }
    WideTree T; T.W = W;
    struct Todo { int q, slot; };
    std::vector<Todo> todo{{0, 0}};
    T.nodes.resize(1);
    struct Ch { Box b; int32_t ref; };
    std::function<void(const Ch&, int, std::vector<Ch>&)> expand = [&](const Ch& c, int j, std::vector<Ch>& out) {
        if (c.ref < 0 || j == 1) { out.push_back(c); return; }
        const int ch = choice[c.ref][j - 1];
        if (ch == 0) { expand(c, j - 1, out); return; }
        expand(Ch{bin[c.ref].box[0], bin[c.ref].child[0]}, ch, out);
        expand(Ch{bin[c.ref].box[1], bin[c.ref].child[1]}, j - ch, out);
    };
    for (size_t t = 0; t < todo.size(); t++) {
        const Todo cur = todo[t];
        std::vector<Ch> ch;
        expand(Ch{bin[cur.q].box[0], bin[cur.q].child[0]}, kw[cur.q], ch);
        expand(Ch{bin[cur.q].box[1], bin[cur.q].child[1]}, W - kw[cur.q], ch);
        T.summed_area += own_area[cur.q];
        Wide w;
        for (const Ch& c : ch) {
            w.box.push_back(c.b);
            if (c.ref >= 0) { w.ref.push_back((int32_t)T.nodes.size()); T.nodes.emplace_back(); todo.push_back(Todo{c.ref, (int)T.nodes.size() - 1}); }
            else w.ref.push_back(c.ref);
        }
        T.nodes[cur.slot] = w;
    }
    return T;
}

struct Counts { unsigned long long visits = 0, boxes = 0, slots = 0, leaves = 0, rays = 0, all_leaf = 0, no_leaf = 0; };
static void walk(const WideTree& T, V3 o, V3 d, Counts& c)
{
    const V3 inv = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    std::vector<int> st{0};
    c.rays++;
    while (!st.empty()) {
        const Wide& w = T.nodes[st.back()]; st.pop_back();
        c.visits++; c.slots += (unsigned long long)T.W; c.boxes += w.box.size();
        { size_t nl = 0; for (int32_t r : w.ref) nl += r < 0; c.all_leaf += nl == w.ref.size(); c.no_leaf += nl == 0; }
        for (size_t i = 0; i < w.box.size(); i++)
            if (hit_box(w.box[i], o, inv, d)) { if (w.ref[i] >= 0) st.push_back(w.ref[i]); else c.leaves++; }
    }
}

int main(int argc, char** argv)
{
    const char* cfg = argc > 1 ? argv[1] : "scenes/cornell-box/config.json";
    const int n_primary = argc > 2 ? std::atoi(argv[2]) : 20000;
    crt_task task;
    if (crt_task_load(cfg, &task) != 0) { std::printf("task: %s\n", crt_last_error()); return 1; }
    crt_host_scene* hs = nullptr;
    if (crt_host_scene_create(task.width, task.height, &hs) != 0) return 1;
    for (uint32_t i = 0; i < task.n_objs && i < 8; i++)
        if (crt_host_scene_add_obj(hs, task.obj_path[i], task.mtl_dir[i]) != 0) { std::printf("load: %s\n", crt_last_error()); return 1; }
    if (crt_host_scene_set_bvh(hs, task.bvh_thresh_n) != 0) return 1;
    crt_scene_desc d;
    if (crt_host_scene_desc(hs, &d) != 0) return 1;
    std::vector<crtaccel::Prim> prims;
    for (uint32_t i = 0; i < d.n_nodes; i++) {
        if (!(d.nodes[i].lc < 0 && d.nodes[i].rc < 0)) continue;
        crtaccel::Prim p;
        for (int a = 0; a < 3; a++) { p.box.lo[a] = d.nodes[i].aa[a]; p.box.hi[a] = d.nodes[i].bb[a]; }
        p.ref = ~(int32_t)i;
        prims.push_back(p);
    }
    std::vector<crtaccel::Node> bin;
    int32_t root = 0;
    crtaccel::build_sah(prims, bin, root);
    crtaccel::optimize_sah(bin, 1);
    // closest hit through the reference tree (for the secondary rays' origins)
    auto closest = [&](V3 o, V3 dir, float& t_out, int& tri_out) {
        const V3 inv = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
        std::vector<int> st{d.root};
        t_out = FLT_MAX; tri_out = -1;
        while (!st.empty()) {
            const crt_bvh_node& n = d.nodes[st.back()]; st.pop_back();
            Box b; for (int a = 0; a < 3; a++) { b.lo[a] = n.aa[a]; b.hi[a] = n.bb[a]; }
            if (!hit_box(b, o, inv, dir)) continue;
            if (n.lc < 0 && n.rc < 0) {
                for (int k = 0; k < (int)n.n; k++) {
                    const crt_triangle& T = d.tris[n.it + k];
                    const V3 v1 = {T.v1[0], T.v1[1], T.v1[2]}, e1 = sub({T.v2[0], T.v2[1], T.v2[2]}, v1), e2 = sub({T.v3[0], T.v3[1], T.v3[2]}, v1);
                    const V3 s = sub(o, v1), s1 = cross(dir, e2), s2 = cross(s, e1);
                    const float det = dot(s1, e1), r = 1.0f / det, be = dot(s1, s) * r, ga = dot(s2, dir) * r, t = dot(s2, e2) * r, al = 1 - be - ga;
                    if (al > 0 && be > 0 && ga > 0 && al < 1 && be < 1 && ga < 1 && t > 0.00001f && t < t_out) { t_out = t; tri_out = n.it + k; }
                }
            } else { if (n.lc >= 0) st.push_back(n.lc); if (n.rc >= 0) st.push_back(n.rc); }
        }
    };
    float iv[9];
    crt_inverse_view(task.eye_pos, task.lookat, task.up, iv);
    const float scale = std::tan(task.fov_y * 3.14159265358979f / 180.0f * 0.5f), ar = (float)task.width / task.height;
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    struct Ray { V3 o, d; int kind; };
    std::vector<Ray> rays;
    for (int i = 0; i < n_primary; i++) {
        const float x = (2 * U(rng) - 1) * scale * ar, y = (1 - 2 * U(rng)) * scale;
        const V3 cd = unit({-x, y, 1});
        const V3 wd = unit({iv[0] * cd.x + iv[3] * cd.y + iv[6] * cd.z, iv[1] * cd.x + iv[4] * cd.y + iv[7] * cd.z, iv[2] * cd.x + iv[5] * cd.y + iv[8] * cd.z});
        const V3 eye = {task.eye_pos[0], task.eye_pos[1], task.eye_pos[2]};
        rays.push_back({eye, wd, 0});
        float t; int tri;
        closest(eye, wd, t, tri);
        if (tri < 0) continue;
        const V3 p = add(eye, mul(wd, t));
        const crt_triangle& T = d.tris[tri];
        V3 n = {T.normal[0], T.normal[1], T.normal[2]};
        // a uniform direction in the hemisphere of n
        V3 h;
        do { h = {2 * U(rng) - 1, 2 * U(rng) - 1, 2 * U(rng) - 1}; } while (dot(h, h) > 1 || dot(h, h) < 1e-4f);
        h = unit(h);
        if (dot(h, n) < 0) h = mul(h, -1);
        rays.push_back({p, h, 1});
        if (d.n_light_tris) { // towards a uniform point of a light triangle chosen by count (DeviceLights.cuh:33-37)
            const crt_triangle& L = d.light_tris[(size_t)(U(rng) * d.n_light_tris) % d.n_light_tris];
            const float a = U(rng), b = U(rng) * (1 - a), g = 1 - a - b;
            const V3 lp = {a * L.v1[0] + b * L.v2[0] + g * L.v3[0], a * L.v1[1] + b * L.v2[1] + g * L.v3[1], a * L.v1[2] + b * L.v2[2] + g * L.v3[2]};
            rays.push_back({p, unit(sub(lp, p)), 2});
        }
    }
    std::printf("{\"scene\": \"%s\", \"leaves\": %zu, \"binary_nodes\": %zu, \"rays\": %zu}\n", cfg, prims.size(), bin.size(), rays.size());
    for (int W : {2, 4, 6, 8}) {
        const WideTree T = collapse(bin, W);
        Counts c[3];
        for (const Ray& r : rays) {
            if (r.d.x == 0 || r.d.y == 0 || r.d.z == 0) continue;
            walk(T, r.o, r.d, c[r.kind]);
        }
        unsigned long long children = 0, inner_children = 0;
        for (const Wide& w : T.nodes) { children += w.box.size(); for (int32_t r : w.ref) inner_children += r >= 0; }
        Counts a; for (int k = 0; k < 3; k++) { a.all_leaf += c[k].all_leaf; a.no_leaf += c[k].no_leaf; a.visits += c[k].visits; a.boxes += c[k].boxes; a.slots += c[k].slots; a.leaves += c[k].leaves; a.rays += c[k].rays; }
        std::printf("{\"W\": %d, \"wide_nodes\": %zu, \"children_per_node\": %.2f, \"summed_area\": %.5e, \"visits_per_ray\": %.3f, \"child_slots_per_ray\": %.2f, "
                    "\"real_children_per_ray\": %.2f, \"leaf_boxes_hit_per_ray\": %.3f, \"visits_at_nodes_whose_children_are_all_leaves\": %.3f, \"visits_at_nodes_without_leaf_children\": %.3f, \"by_kind_visits\": {\"camera\": %.2f, \"bounce\": %.2f, \"to_light\": %.2f}}\n",
                    W, T.nodes.size(), (double)children / T.nodes.size(), T.summed_area, (double)a.visits / a.rays, (double)a.slots / a.rays, (double)a.boxes / a.rays,
                    (double)a.leaves / a.rays, (double)a.all_leaf / a.visits, (double)a.no_leaf / a.visits, (double)c[0].visits / std::max(1ull, c[0].rays), (double)c[1].visits / std::max(1ull, c[1].rays), (double)c[2].visits / std::max(1ull, c[2].rays));
    }
    return 0;
}
