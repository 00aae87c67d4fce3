// tools/sah_opt_bench.cpp -- host-side timing of the tree set-up of crt_scene_create on a scene file, without a GPU: the binned SAH build over
// the reference's leaves (crt_accel.h: build_sah -- on a GPU box the device builder does this part) and the insertion-based optimisation pass
// (optimize_sah), which is host code on every box and the larger part of `scene_setup.sah_tree.ms`.
// build: g++ -O2 -std=c++17 -Iinclude -Icudaraytracing_amd/csrc tools/sah_opt_bench.cpp -Lcudaraytracing_amd/lib -lcrt -Wl,-rpath,$PWD/cudaraytracing_amd/lib -o /tmp/sah_opt_bench
// usage: /tmp/sah_opt_bench scenes/cornell-box/config.json [reps]
#include "crt.h"
#include "crt_accel.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

int main(int argc, char** argv)
{
    const char* cfg = argc > 1 ? argv[1] : "scenes/cornell-box/config.json";
    const int reps = argc > 2 ? std::atoi(argv[2]) : 5;
    crt_task task;
    if (crt_task_load(cfg, &task) != 0) { std::printf("task: %s\n", crt_last_error()); return 1; }
    crt_host_scene* hs = nullptr;
    if (crt_host_scene_create(task.width, task.height, &hs) != 0) return 1;
    for (uint32_t i = 0; i < task.n_objs && i < 8; i++)
        if (crt_host_scene_add_obj(hs, task.obj_path[i], task.mtl_dir[i]) != 0) { std::printf("load: %s\n", crt_last_error()); return 1; }
    if (crt_host_scene_set_bvh(hs, task.bvh_thresh_n) != 0) { std::printf("bvh: %s\n", crt_last_error()); return 1; }
    crt_scene_desc d;
    if (crt_host_scene_desc(hs, &d) != 0) return 1;
    std::vector<crtaccel::Prim> prims0;
    for (uint32_t i = 0; i < d.n_nodes; i++) {
        if (!(d.nodes[i].lc < 0 && d.nodes[i].rc < 0)) continue;
        crtaccel::Prim p;
        for (int a = 0; a < 3; a++) { p.box.lo[a] = d.nodes[i].aa[a]; p.box.hi[a] = d.nodes[i].bb[a]; }
        p.ref = ~(int32_t)i;
        prims0.push_back(p);
    }
    using clk = std::chrono::steady_clock;
    double best_build = 1e9, best_opt = 1e9;
    size_t n_nodes = 0;
    int depth = 0;
    for (int r = 0; r < reps; r++) {
        std::vector<crtaccel::Prim> prims = prims0;
        std::vector<crtaccel::Node> nodes;
        int32_t root = 0;
        const auto t0 = clk::now();
        crtaccel::build_sah(prims, nodes, root);
        const auto t1 = clk::now();
        depth = crtaccel::optimize_sah(nodes, 1);
        const auto t2 = clk::now();
        if (r == 0) { // the same pass on one thread and by the serial form of round 4: same tree whatever the threads; a tree over the same leaves whose
                      // boxes are its children's; summed inner area beside the serial form's
            std::vector<crtaccel::Prim> p2 = prims0;
            std::vector<crtaccel::Node> n1, ns, n0;
            int32_t r2 = 0;
            crtaccel::build_sah(p2, n1, r2);
            ns = n1; n0 = n1;
            const auto u0 = clk::now();
            crtaccel::optimize_sah(n1, 1, 1);
            const auto u1 = clk::now();
            crtaccel::optimize_sah_serial(ns, 1);
            const auto u2 = clk::now();
            const bool same = n1.size() == nodes.size() && std::memcmp(n1.data(), nodes.data(), n1.size() * sizeof(crtaccel::Node)) == 0;
            auto area_of = [](const std::vector<crtaccel::Node>& v) { double a = 0; for (const auto& n : v) { crtaccel::Box b = n.box[0]; b.grow(n.box[1]); a += b.half_area(); } return a; };
            // validity of the tree the product uses: every inner node reached once from the root, every leaf ref of the built tree present once,
            // the box stored for an inner child equal to the union of that child's two boxes
            bool valid = nodes.size() == n0.size();
            std::vector<int32_t> refs_in, refs_out;
            for (const auto& n : n0) for (int s = 0; s < 2; s++) if (n.child[s] < 0) refs_in.push_back(n.child[s]);
            std::vector<char> seen(nodes.size(), 0);
            std::vector<int> todo(1, 0);
            size_t reached = 0;
            while (valid && !todo.empty()) {
                const int q = todo.back(); todo.pop_back();
                if (q < 0 || (size_t)q >= nodes.size() || seen[q]) { valid = false; break; }
                seen[q] = 1; reached++;
                for (int s = 0; s < 2; s++) {
                    const int32_t c = nodes[q].child[s];
                    if (c < 0) { refs_out.push_back(c); continue; }
                    if ((size_t)c >= nodes.size()) { valid = false; break; }
                    crtaccel::Box b = nodes[c].box[0]; b.grow(nodes[c].box[1]);
                    if (std::memcmp(&b, &nodes[q].box[s], sizeof(b)) != 0) valid = false;
                    todo.push_back(c);
                }
            }
            std::sort(refs_in.begin(), refs_in.end()); std::sort(refs_out.begin(), refs_out.end());
            valid = valid && reached == nodes.size() && refs_in == refs_out;
            std::printf("{\"one_thread_equals_many\": %s, \"valid_tree\": %s, \"one_thread_ms\": %.2f, \"serial_round4_ms\": %.2f, \"summed_area_built\": %.6e, \"summed_area_serial\": %.6e, \"summed_area_batched\": %.6e}\n",
                        same ? "true" : "false", valid ? "true" : "false", std::chrono::duration<double, std::milli>(u1 - u0).count(), std::chrono::duration<double, std::milli>(u2 - u1).count(),
                        area_of(n0), area_of(ns), area_of(n1));
        }
        best_build = std::min(best_build, std::chrono::duration<double, std::milli>(t1 - t0).count());
        best_opt = std::min(best_opt, std::chrono::duration<double, std::milli>(t2 - t1).count());
        n_nodes = nodes.size();
        double area = 0;
        for (const auto& n : nodes) { crtaccel::Box b = n.box[0]; b.grow(n.box[1]); area += b.half_area(); }
        if (r == 0) std::printf("{\"leaves\": %zu, \"nodes\": %zu, \"depth\": %d, \"summed_area\": %.6e}\n", prims0.size(), n_nodes, depth, area);
    }
    std::printf("{\"host_build_sah_ms\": %.2f, \"optimize_sah_ms\": %.2f}\n", best_build, best_opt);
    return 0;
}
