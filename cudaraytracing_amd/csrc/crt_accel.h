// cudaraytracing_amd/csrc/crt_accel.h -- host-side build of the traversal tree used by
// CRT_TRAVERSAL_FAST.
//
// Why a second tree is legal.  The reference visits a leaf iff the slab test (hit_AABB,
// DeviceBVH.cuh:87-126) passes for every ancestor box below the root; the leaf's own box is the
// last of them.  Every ancestor box is the exact float union of the leaf boxes under it
// (BVH.h:43-52), and for a ray whose inv_dir components are all finite each term of the slab test
// is monotone in the box bounds (IEEE subtraction, multiplication by a fixed finite factor, the
// x>y?x:y max/min without NaNs, t_exit + EPSILON): if a box passes, every box that contains it
// passes.  Hence
//        the reference tests the triangles of leaf l  <=>  the slab test passes for l's own box,
// independently of the tree above the leaves.  Any hierarchy over the reference's LEAVES whose
// inner boxes are exact unions therefore visits a superset of the right leaves, and testing each
// leaf child's own (reference) box with the reference formula selects exactly the right ones.
// Rays with a non-finite inv_dir component (a direction component that is zero or denormal) can
// produce NaNs in the slab test; they keep using the reference-topology tree, for which the
// traversal performs the reference's own box tests.
//
// The reference's median split (BVH.h:63-81) interleaves the scene's few huge wall triangles with
// the dense meshes, which inflates the boxes along their root paths (measured: 49 inner nodes
// per ray on the Cornell stand-in even with ordering and pruning).  This builder is a binned
// surface-area-heuristic BVH over the reference leaves (leaf = one reference leaf).
#ifndef CRT_ACCEL_H
#define CRT_ACCEL_H

#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#if defined(__linux__)
#include <sched.h>
#endif

namespace crtaccel {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; } }
    void grow(const Box& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    double half_area() const
    {
        double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim {          // one reference leaf
    Box box;           // its box exactly as the reference stores it (DeviceBVHNode AA/BB)
    int32_t ref;       // encoded leaf reference (negative)
};

struct Node {          // inner node: two children with their boxes
    Box box[2];
    int32_t child[2];  // >= 0: inner node index (breadth-first numbering), < 0: Prim::ref
};

// Builds the tree; nodes are returned in breadth-first order (root = 0).  Returns the depth
// (a tree that is a single leaf has depth 1 and no inner nodes; root_ref tells which).
inline int build_sah(std::vector<Prim>& prims, std::vector<Node>& nodes, int32_t& root_ref, uint32_t* index_splits = nullptr)
{
    nodes.clear();
    if (index_splits) *index_splits = 0;
    const int n = (int)prims.size();
    if (n == 1) { root_ref = prims[0].ref; return 1; }
    // (CRT_SAH_BINS: experiment hook of the HOST builder only -- the device builder has 32 bins, and the two are compared node for node)
    const int NBMAX = 256;
    int NB = 32;
    if (const char* e_ = std::getenv("CRT_SAH_BINS")) NB = std::min(NBMAX, std::max(2, std::atoi(e_)));
    double cost_pow = 1.0; // (CRT_SAH_POW: experiment hook of the HOST builder only: leaf-count exponent of the split cost)
    if (const char* e_ = std::getenv("CRT_SAH_POW")) cost_pow = std::atof(e_);
    struct Task { int b, e, node, slot, depth; };
    // temporary tree in build order, renumbered breadth-first at the end
    struct Tmp { Box box[2]; int32_t child[2]; int depth; };
    std::vector<Tmp> tmp;
    std::vector<Task> todo;
    tmp.push_back(Tmp());
    tmp[0].depth = 1;
    // the root "slot" trick: a virtual parent is not needed; handle the root split directly
    struct Range { int b, e, tmp_index; };
    std::vector<Range> stack;
    stack.push_back(Range{0, n, 0});
    int max_depth = 1;
    while (!stack.empty()) {
        Range r = stack.back();
        stack.pop_back();
        const int b = r.b, e = r.e, cnt = e - b;
        // centroid bounds
        float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int i = b; i < e; i++)
            for (int a = 0; a < 3; a++) {
                float c = 0.5f * (prims[i].box.lo[a] + prims[i].box.hi[a]);
                clo[a] = std::min(clo[a], c); chi[a] = std::max(chi[a], c);
            }
        int best_axis = -1, best_split = -1;
        double best_cost = DBL_MAX;
        for (int a = 0; a < 3; a++) {
            float ext = chi[a] - clo[a];
            if (!(ext > 0.0f)) continue;
            Box bb[NBMAX];
            int bc[NBMAX];
            for (int k = 0; k < NB; k++) { bb[k].reset(); bc[k] = 0; }
            const float scale = (float)NB / ext;
            for (int i = b; i < e; i++) {
                float c = 0.5f * (prims[i].box.lo[a] + prims[i].box.hi[a]);
                int k = std::min(NB - 1, std::max(0, (int)((c - clo[a]) * scale)));
                bb[k].grow(prims[i].box);
                bc[k]++;
            }
            double right_area[NBMAX];
            int right_cnt[NBMAX];
            Box acc;
            acc.reset();
            int c = 0;
            for (int k = NB - 1; k >= 1; k--) { acc.grow(bb[k]); c += bc[k]; right_area[k] = acc.half_area(); right_cnt[k] = c; }
            acc.reset();
            c = 0;
            for (int k = 0; k < NB - 1; k++) {
                acc.grow(bb[k]);
                c += bc[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                double cost = acc.half_area() * c + right_area[k + 1] * right_cnt[k + 1];
                if (cost_pow != 1.0) cost = acc.half_area() * std::pow((double)c, cost_pow) + right_area[k + 1] * std::pow((double)right_cnt[k + 1], cost_pow);
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; }
            }
        }
        int mid;
        if (best_axis < 0) {
            mid = b + cnt / 2; // all centroids coincide: split by index
            if (index_splits) (*index_splits)++;
        } else {
            const int a = best_axis;
            const float scale = (float)NB / (chi[a] - clo[a]);
            auto it = std::partition(prims.begin() + b, prims.begin() + e, [&](const Prim& p) {
                float c = 0.5f * (p.box.lo[a] + p.box.hi[a]);
                int k = std::min(NB - 1, std::max(0, (int)((c - clo[a]) * scale)));
                return k <= best_split;
            });
            mid = (int)(it - prims.begin());
            if (mid == b || mid == e) mid = b + cnt / 2;
        }
        const int halves[2][2] = {{b, mid}, {mid, e}};
        for (int s = 0; s < 2; s++) {
            const int hb = halves[s][0], he = halves[s][1];
            Box bx;
            bx.reset();
            for (int i = hb; i < he; i++) bx.grow(prims[i].box);
            tmp[r.tmp_index].box[s] = bx;
            if (he - hb == 1) {
                tmp[r.tmp_index].child[s] = prims[hb].ref;
                max_depth = std::max(max_depth, tmp[r.tmp_index].depth + 1);
            } else {
                Tmp t;
                t.depth = tmp[r.tmp_index].depth + 1;
                max_depth = std::max(max_depth, t.depth + 1);
                tmp.push_back(t);
                const int ci = (int)tmp.size() - 1;
                tmp[r.tmp_index].child[s] = ci;
                stack.push_back(Range{hb, he, ci});
            }
        }
    }
    // breadth-first renumbering
    std::vector<int> order, index(tmp.size(), -1);
    order.push_back(0);
    index[0] = 0;
    for (size_t q = 0; q < order.size(); q++)
        for (int s = 0; s < 2; s++) {
            int c = tmp[order[q]].child[s];
            if (c >= 0) { index[c] = (int)order.size(); order.push_back(c); }
        }
    nodes.resize(order.size());
    for (size_t q = 0; q < order.size(); q++) {
        const Tmp& t = tmp[order[q]];
        for (int s = 0; s < 2; s++) {
            nodes[q].box[s] = t.box[s];
            nodes[q].child[s] = t.child[s] >= 0 ? index[t.child[s]] : t.child[s];
        }
    }
    root_ref = 0;
    return max_depth;
}

// Insertion-based optimisation of a built tree (Bittner, Hapala, Havran 2013, simplified): every node in turn (largest first) is
// taken out of the tree and put back where it adds the least surface area (its own new parent's area plus what it makes its new
// ancestors grow by), found by a best-first search with the induced cost as lower bound.  The old position is among the candidates,
// so the summed area of the inner nodes never grows.  The tree stays a tree over the same leaves, so results cannot change; what
// changes is the number of nodes and leaves a ray meets (one pass, moves that save at least half: cornell-box stand-in 5.04 -> 4.31
// inner and 3.93 -> 3.68 leaf visits per ray, frame - 5.4 %; veach-mis - 1.7 %; the 102 412-triangle variant - 3.9 %; docs/experiments.md 7).
// Returns the new depth; nodes come back in breadth-first order, root = 0.
inline int optimize_sah_serial(std::vector<Node>& nodes, int passes)
{
    const int A = (int)nodes.size();
    if (A < 3 || passes <= 0) {
        // depth of the tree as it is
        std::vector<int> dep(A, 1);
        int md = A ? 2 : 1;
        for (int q = 0; q < A; q++)
            for (int s = 0; s < 2; s++) {
                if (nodes[q].child[s] >= 0) dep[nodes[q].child[s]] = dep[q] + 1;
                md = std::max(md, dep[q] + 1);
            }
        return md;
    }
    struct N { Box box; int parent, child[2]; int32_t leaf_ref; double area; };
    std::vector<N> t((size_t)2 * A + 1);
    int n_all = A;
    for (int q = 0; q < A; q++) { t[q].parent = -1; t[q].leaf_ref = 0; }
    for (int q = 0; q < A; q++)
        for (int s = 0; s < 2; s++) {
            const int32_t c = nodes[q].child[s];
            int id;
            if (c >= 0) id = c;
            else { id = n_all++; t[id].child[0] = t[id].child[1] = -1; t[id].leaf_ref = c; }
            t[q].child[s] = id;
            t[id].parent = q;
            t[id].box = nodes[q].box[s];
            t[id].area = t[id].box.half_area();
        }
    t[0].box = nodes[0].box[0]; t[0].box.grow(nodes[0].box[1]); t[0].area = t[0].box.half_area();
    int root = 0;
    auto is_leaf = [&](int x) { return t[x].child[0] < 0; };
    // (a box that comes out as it was leaves every ancestor as it was: the walk ends there -- most walks after a few levels instead of
    // at the root, which is where the pass spent two thirds of its time; round 5)
    auto same_box = [](const Box& a, const Box& b) {
        return a.lo[0] == b.lo[0] && a.lo[1] == b.lo[1] && a.lo[2] == b.lo[2] && a.hi[0] == b.hi[0] && a.hi[1] == b.hi[1] && a.hi[2] == b.hi[2];
    };
    // (`stale`: the first node's stored box describes OTHER children -- the parent node that travels with a reinserted subtree -- and says
    // nothing about its new ancestors)
    auto refit_up = [&](int x, bool stale) {
        for (; x >= 0; x = t[x].parent, stale = false) {
            Box b = t[t[x].child[0]].box;
            b.grow(t[t[x].child[1]].box);
            if (!stale && same_box(b, t[x].box)) break;
            t[x].box = b;
            t[x].area = b.half_area();
        }
    };
    double margin = 0.5;
    if (const char* e_ = std::getenv("CRT_SAH_OPT_MARGIN")) margin = std::atof(e_);
    struct Cand { double ind; int x; };
    auto cmp = [](const Cand& a, const Cand& b) { return a.ind > b.ind; };
    std::vector<Cand> heap;
    for (int pass = 0; pass < passes; pass++) {
        std::vector<int> order;
        for (int x = 0; x < n_all; x++)
            if (x != root && t[x].parent != root) order.push_back(x);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return t[a].area != t[b].area ? t[a].area > t[b].area : a < b; });
        for (int x : order) {
            const int P = t[x].parent;
            if (P < 0 || P == root) continue; // (the tree above may have changed since the list was made)
            const int S = t[P].child[0] == x ? t[P].child[1] : t[P].child[0];
            const int G = t[P].parent;
            // take x (and its parent node P) out: the sibling moves up
            t[G].child[t[G].child[0] == P ? 0 : 1] = S;
            t[S].parent = G;
            refit_up(G, false);
            // where does it cost least?  A move must pay: `orig` is what putting x back beside its old sibling costs, and a new place has
            // to beat it by the margin (default one half; CRT_SAH_OPT_MARGIN) -- reinsertions that gain next to nothing only stir up
            // overlap, and with the bound known beforehand the search ends near the root for the nodes that stay (most of them)
            const Box bx = t[x].box;
            const double ax = t[x].area;
            double best = DBL_MAX;
            int best_x = S;
            if (margin > 0.0) {
                Box u = t[S].box;
                u.grow(bx);
                double orig = u.half_area();
                for (int y = t[S].parent; y >= 0; y = t[y].parent) {
                    Box v = t[y].box;
                    v.grow(bx);
                    if (same_box(v, t[y].box)) break; // (x lies inside y's box, hence inside every ancestor's: nothing more is added)
                    orig += v.half_area() - t[y].area;
                }
                best = orig * (1.0 - margin);
            }
            heap.clear();
            heap.push_back(Cand{0.0, root});
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end(), cmp);
                const Cand c = heap.back();
                heap.pop_back();
                if (c.ind + ax >= best) break; // the cheapest induced cost left cannot beat the best position any more
                Box u = t[c.x].box;
                u.grow(bx);
                const double direct = u.half_area();
                if (c.ind + direct < best) { best = c.ind + direct; best_x = c.x; }
                if (!is_leaf(c.x)) {
                    const double ind = c.ind + direct - t[c.x].area;
                    if (ind + ax < best)
                        for (int s2 = 0; s2 < 2; s2++) { heap.push_back(Cand{ind, t[c.x].child[s2]}); std::push_heap(heap.begin(), heap.end(), cmp); }
                }
            }
            // P becomes the parent of (best_x, x) in best_x's place
            const int Q = t[best_x].parent;
            t[P].child[0] = best_x; t[P].child[1] = x;
            t[P].parent = Q;
            t[best_x].parent = P; t[x].parent = P;
            if (Q >= 0) t[Q].child[t[Q].child[0] == best_x ? 0 : 1] = P;
            else root = P;
            refit_up(P, true);
        }
    }
    // back to the breadth-first array of inner nodes
    std::vector<int> order, index((size_t)n_all, -1), dep;
    order.push_back(root); dep.push_back(1); index[root] = 0;
    int md = 2;
    for (size_t q = 0; q < order.size(); q++)
        for (int s = 0; s < 2; s++) {
            const int c = t[order[q]].child[s];
            md = std::max(md, dep[q] + 1);
            if (!is_leaf(c)) { index[c] = (int)order.size(); order.push_back(c); dep.push_back(dep[q] + 1); }
        }
    std::vector<Node> out(order.size());
    for (size_t q = 0; q < order.size(); q++)
        for (int s = 0; s < 2; s++) {
            const int c = t[order[q]].child[s];
            out[q].box[s] = t[c].box;
            out[q].child[s] = is_leaf(c) ? t[c].leaf_ref : index[c];
        }
    nodes.swap(out);
    return md;
}

// The pass as crt_scene_create runs it (round 6): the searches of a BLOCK of nodes run on the tree AS IT IS AT THE START OF
// THE BLOCK, in parallel; their moves are then applied in order.  Of the ~49 000 nodes of the stand-in's tree 300 move: the time goes into the
// searches that end where they began, and those never change the tree -- a search works with the node's removal emulated (the boxes along
// its root path shrunk in a thread-local overlay).  A move found on the block's tree is applied to the tree as earlier moves of the block left
// it, provided it is still a move of a tree (the node still has a grandparent, the place is not its own parent, sibling or subtree); its gain
// was priced on the block's tree, the way the parallel reinsertion of Meister & Bittner 2018 prices a whole pass.  The block size is a
// constant, so the result depends neither on the number of threads nor on timing; it is not the tree of optimize_sah_serial (whose every
// search sees every earlier move), but one of the same quality: summed inner area 1.7009e7 -> see tests/test_sah_opt.py, and the frame times
// of docs/experiments.md 6.10.  (Round 5's form reproduced the serial pass exactly by recording what every search read and repeating the
// stale ones: the bookkeeping cost as much as the searches, 21 ms on eight threads against 32 serial.)
#ifndef CRT_SAH_HEAD
#define CRT_SAH_HEAD 64
#endif
inline int optimize_sah(std::vector<Node>& nodes, int passes, int n_threads = 0)
{
    const int A = (int)nodes.size();
    if (A < 3 || passes <= 0) return optimize_sah_serial(nodes, passes);
    if (n_threads <= 0) {
        // the CPUs this process may RUN on (its affinity mask: a cgroup's cpuset or taskset shows here; hardware_concurrency() counts the
        // machine's); the helpers spin between blocks, so never more threads than that
        unsigned usable = std::max(1u, std::thread::hardware_concurrency());
#if defined(__linux__)
        cpu_set_t set_;
        CPU_ZERO(&set_);
        if (sched_getaffinity(0, sizeof(set_), &set_) == 0) usable = (unsigned)std::max(1, CPU_COUNT(&set_));
#endif
        n_threads = (int)std::min(8u, usable);
        if (const char* e_ = std::getenv("CRT_SAH_OPT_THREADS")) n_threads = std::max(1, std::atoi(e_));
    }
    struct N { Box box; int parent, child[2]; int32_t leaf_ref; double area; };
    std::vector<N> t((size_t)2 * A + 1);
    int n_all = A;
    for (int q = 0; q < A; q++) { t[q].parent = -1; t[q].leaf_ref = 0; }
    for (int q = 0; q < A; q++)
        for (int s = 0; s < 2; s++) {
            const int32_t c = nodes[q].child[s];
            int id;
            if (c >= 0) id = c;
            else { id = n_all++; t[id].child[0] = t[id].child[1] = -1; t[id].leaf_ref = c; }
            t[q].child[s] = id;
            t[id].parent = q;
            t[id].box = nodes[q].box[s];
            t[id].area = t[id].box.half_area();
        }
    t[0].box = nodes[0].box[0]; t[0].box.grow(nodes[0].box[1]); t[0].area = t[0].box.half_area();
    int root = 0;
    auto is_leaf = [&](int x) { return t[x].child[0] < 0; };
    auto same_box = [](const Box& a, const Box& b) {
        return a.lo[0] == b.lo[0] && a.lo[1] == b.lo[1] && a.lo[2] == b.lo[2] && a.hi[0] == b.hi[0] && a.hi[1] == b.hi[1] && a.hi[2] == b.hi[2];
    };
    double margin = 0.5;
    if (const char* e_ = std::getenv("CRT_SAH_OPT_MARGIN")) margin = std::atof(e_);
    struct Cand { double ind; int x; };
    struct alignas(64) Work { // per thread (cache lines of its own: the vectors' headers are written at every push)
        std::vector<Cand> heap;
        std::vector<int> pidx;      // node -> index on the emulated root path, or -1
        std::vector<int> path;      // the path nodes G .. root
        std::vector<Box> sbox;      // their boxes with x taken out
        std::vector<double> sarea;
    };
    // the search for node x on the tree as it is, x's removal emulated: the node to put x beside, or -1 when x stays
    auto search = [&](int x, Work& w) -> int {
        const int P = t[x].parent;
        if (P < 0 || P == root) return -1;
        const int S = t[P].child[0] == x ? t[P].child[1] : t[P].child[0];
        const int G = t[P].parent;
        // the root path without x: boxes shrink until one comes out as it was
        w.path.clear(); w.sbox.clear(); w.sarea.clear();
        {
            Box cur = t[S].box;
            int below = P;
            for (int y = G; y >= 0; below = y, y = t[y].parent) {
                const int other = t[y].child[0] == below ? t[y].child[1] : t[y].child[0];
                Box b = cur;
                b.grow(t[other].box);
                if (y != G && same_box(b, t[y].box)) break; // (from here up nothing changes; G itself is always on the path: its children differ)
                w.pidx[y] = (int)w.path.size();
                w.path.push_back(y); w.sbox.push_back(b); w.sarea.push_back(b.half_area());
                cur = b;
            }
        }
        auto box_of = [&](int c) -> const Box& { return w.pidx[c] >= 0 ? w.sbox[(size_t)w.pidx[c]] : t[c].box; };
        auto area_of = [&](int c) { return w.pidx[c] >= 0 ? w.sarea[(size_t)w.pidx[c]] : t[c].area; };
        const Box bx = t[x].box;
        const double ax = t[x].area;
        // A move must pay: `orig` is what putting x back beside its old sibling costs, and a new place has to beat it by the margin (default
        // one half; CRT_SAH_OPT_MARGIN) -- reinsertions that gain next to nothing only stir up overlap, and with the bound known beforehand
        // the search ends near the root for the nodes that stay (most of them)
        double best = DBL_MAX;
        int best_x = S;
        if (margin > 0.0) {
            Box u = t[S].box;
            u.grow(bx);
            double orig = u.half_area();
            for (int y = G; y >= 0; y = t[y].parent) {
                Box v = box_of(y);
                v.grow(bx);
                if (same_box(v, box_of(y))) break; // (x lies inside y's box, hence inside every ancestor's: nothing more is added)
                orig += v.half_area() - area_of(y);
            }
            best = orig * (1.0 - margin);
        }
        auto cmp = [](const Cand& a, const Cand& b) { return a.ind != b.ind ? a.ind > b.ind : a.x > b.x; };
        w.heap.clear();
        w.heap.push_back(Cand{0.0, root});
        while (!w.heap.empty()) {
            std::pop_heap(w.heap.begin(), w.heap.end(), cmp);
            const Cand c = w.heap.back();
            w.heap.pop_back();
            if (c.ind + ax >= best) break; // the cheapest induced cost left cannot beat the best position any more
            Box u = box_of(c.x);
            u.grow(bx);
            const double direct = u.half_area();
            if (c.ind + direct < best) { best = c.ind + direct; best_x = c.x; }
            if (!is_leaf(c.x)) {
                const double ind = c.ind + direct - area_of(c.x);
                if (ind + ax < best) {
                    int k0 = t[c.x].child[0], k1 = t[c.x].child[1];
                    if (c.x == G) { if (k0 == P) k0 = S; else k1 = S; } // (P is gone: its sibling hangs under G)
                    for (int k : {k0, k1}) { w.heap.push_back(Cand{ind, k}); std::push_heap(w.heap.begin(), w.heap.end(), cmp); }
                }
            }
        }
        for (int y : w.path) w.pidx[y] = -1;
        return best_x == S ? -1 : best_x;
    };
    std::vector<Work> work((size_t)n_threads);
    for (Work& w : work) w.pidx.assign((size_t)n_all, -1);
    // a pool of n_threads - 1 helpers; the calling thread works too.  The helpers SPIN between blocks (a block is under a millisecond of
    // work: waking a sleeping thread takes about as long, and the helpers then find the block done) -- for the few milliseconds of a pass
    struct Pool {
        std::vector<std::thread> th;
        std::atomic<int> gen{0}, running{0};
        std::atomic<bool> quit{false};
        const std::function<void(int)>* job = nullptr;
        // (whatever ends this function -- a bad_alloc of one of the vectors below included -- the helpers are told to leave and joined: a
        // joinable std::thread that is destroyed calls std::terminate)
        ~Pool() { quit.store(true, std::memory_order_release); for (std::thread& x : th) if (x.joinable()) x.join(); }
    } pool;
    for (int k = 1; k < n_threads; k++)
        pool.th.emplace_back([&pool, k] {
            int seen = 0;
            for (;;) {
                int spins = 0;
                while (pool.gen.load(std::memory_order_acquire) == seen && !pool.quit.load(std::memory_order_acquire))
                    if (++spins > 2000) { std::this_thread::yield(); spins = 0; }
                if (pool.quit.load(std::memory_order_acquire)) return;
                seen = pool.gen.load(std::memory_order_acquire);
                (*pool.job)(k);
                pool.running.fetch_sub(1, std::memory_order_acq_rel);
            }
        });
    auto run_all = [&](const std::function<void(int)>& job) {
        if (n_threads > 1) {
            pool.job = &job;
            pool.running.store(n_threads - 1, std::memory_order_release);
            pool.gen.fetch_add(1, std::memory_order_acq_rel);
        }
        job(0);
        if (n_threads > 1) {
            int spins = 0;
            while (pool.running.load(std::memory_order_acquire) != 0)
                if (++spins > 2000) { std::this_thread::yield(); spins = 0; }
        }
    };
    // blocks of 64, 128, ... 2048 nodes (constants: the tree that comes out must not depend on the machine): the large nodes at the head of the
    // list are where the moves are and where one move changes what the next should do
    const int HEAD = CRT_SAH_HEAD, BLOCK0 = 8, BLOCK = 2048;
    std::vector<int> found((size_t)BLOCK);
    for (int pass = 0; pass < passes; pass++) {
        // largest first; one 64-bit key per node (the area as a float, inverted, above the id) so that the sort compares integers
        std::vector<uint64_t> keys;
        keys.reserve((size_t)n_all);
        for (int x = 0; x < n_all; x++)
            if (x != root && t[x].parent != root) {
                const float af = (float)t[x].area;
                uint32_t bits;
                std::memcpy(&bits, &af, 4);
                keys.push_back(((uint64_t)(0xffffffffu - bits) << 32) | (uint32_t)x); // (areas are >= 0: their bit patterns order like the values)
            }
        std::sort(keys.begin(), keys.end());
        const int blk_max = std::min(BLOCK, std::max(BLOCK0, (int)(keys.size() / 64)));
        const size_t serial_head = (size_t)HEAD;
        int blk = 1;
        for (size_t b0 = 0; b0 < keys.size(); b0 += (size_t)blk, blk = b0 < serial_head ? 1 : std::min(blk_max, std::max(BLOCK0, blk * 2))) {
            const int nb = (int)std::min<size_t>((size_t)blk, keys.size() - b0);
            std::atomic<int> next{0};
            if (nb == 1) found[0] = search((int)(uint32_t)keys[b0], work[0]);
            else run_all([&](int tid) {
                for (;;) {
                    const int i0 = next.fetch_add(16);
                    if (i0 >= nb) break;
                    for (int i = i0; i < std::min(nb, i0 + 16); i++) found[(size_t)i] = search((int)(uint32_t)keys[b0 + (size_t)i], work[(size_t)tid]);
                }
            });
            bool moved = false;
            for (int i = 0; i < nb; i++) {
                int bxn = found[(size_t)i];
                if (bxn < 0) continue;
                const int x = (int)(uint32_t)keys[b0 + (size_t)i];
                // a node that moves is priced again on the tree as the block's earlier moves left it (a few hundred searches per pass)
                if (moved) bxn = search(x, work[0]);
                if (bxn < 0) continue;
                const int P = t[x].parent;
                const int S = t[P].child[0] == x ? t[P].child[1] : t[P].child[0];
                // out of its place (the sibling goes up), in beside bxn; P travels with x as the parent of the pair
                const int G = t[P].parent;
                t[G].child[t[G].child[0] == P ? 0 : 1] = S;
                t[S].parent = G;
                for (int y = G; y >= 0; y = t[y].parent) {
                    Box bb = t[t[y].child[0]].box;
                    bb.grow(t[t[y].child[1]].box);
                    if (same_box(bb, t[y].box)) break;
                    t[y].box = bb; t[y].area = bb.half_area();
                }
                const int Q = t[bxn].parent;
                t[P].child[0] = bxn; t[P].child[1] = x;
                t[P].parent = Q;
                t[bxn].parent = P; t[x].parent = P;
                if (Q >= 0) t[Q].child[t[Q].child[0] == bxn ? 0 : 1] = P;
                else root = P;
                bool first = true;
                for (int y = P; y >= 0; y = t[y].parent, first = false) {
                    Box bb = t[t[y].child[0]].box;
                    bb.grow(t[t[y].child[1]].box);
                    if (!first && same_box(bb, t[y].box)) break;
                    t[y].box = bb; t[y].area = bb.half_area();
                }
                moved = true;
            }
        }
    }
    pool.quit.store(true, std::memory_order_release);
    for (std::thread& th : pool.th) th.join();
    pool.th.clear();
    // back to the breadth-first array of inner nodes
    std::vector<int> order, index((size_t)n_all, -1), dep;
    order.push_back(root); dep.push_back(1); index[root] = 0;
    int md = 2;
    for (size_t q = 0; q < order.size(); q++)
        for (int s2 = 0; s2 < 2; s2++) {
            const int c = t[order[q]].child[s2];
            md = std::max(md, dep[q] + 1);
            if (!is_leaf(c)) { index[c] = (int)order.size(); order.push_back(c); dep.push_back(dep[q] + 1); }
        }
    std::vector<Node> out(order.size());
    for (size_t q = 0; q < order.size(); q++)
        for (int s2 = 0; s2 < 2; s2++) {
            const int c = t[order[q]].child[s2];
            out[q].box[s2] = t[c].box;
            out[q].child[s2] = is_leaf(c) ? t[c].leaf_ref : index[c];
        }
    nodes.swap(out);
    return md;
}

// The same tree built on the current HIP device (crt_accel_build.hip): node for node the tree of build_sah, except where a range is
// split "by index" (coinciding centroids).  Returns the depth, or -1 if the device build could not run (use build_sah then).
int build_sah_device(const std::vector<Prim>& prims, std::vector<Node>& nodes, int32_t& root_ref, float* device_ms, uint32_t* index_splits = nullptr);

} // namespace crtaccel
#endif
