// crt_cli -- headless replacement of the reference's GUI shell for the hot path:
// reads a config.json (src/main.cu:67-90), ingests the OBJ/MTL files and builds the BVH exactly
// as render_view() does (src/main.cu:119-145,276), renders one frame with Render::run_view
// (src/main.cu:371-372) and saves it with Render::save_frame_buffer (src/main.cu:363).
// The GUI-settable knobs (spp, P_RR, light_sample_n, eye/lookat/up: Gui.h) are flags.
//
//   crt_cli <config.json> [-o out.png] [--spp N] [--p-rr X] [--lsn N] [--seed S] [--width W] [--height H]
//           [--eye x y z] [--lookat x y z] [--up x y z] [--reference | --exact | --fast] [--bounded-radiance] [--base-dir DIR] [--device N]
//           [--gpus N | --devices a,b,...] [--gather auto|rccl|copy]
// --gpus N renders on devices 0..N-1 of this node in one process (crt_multi: interleaved pixel tiles, one RCCL all-gather);
// --devices names the device of every rank explicitly (a repeated index puts two ranks on one GPU: --gather copy only).
#include "crt_host.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <config.json> [-o out.png] [--spp N] [--p-rr X] [--lsn N] [--seed S] [--width W] [--height H]\n"
                             "       [--eye x y z] [--lookat x y z] [--up x y z] [--reference | --exact | --fast] [--bounded-radiance] [--base-dir DIR] [--device N]\n"
                             "       [--gpus N | --devices a,b,...] [--gather auto|rccl|copy]\n", argv[0]);
        return 2;
    }
    try {
        crt::TaskObjs all_objs;
        crt_task task = crt::load_task(argv[1], &all_objs);
        std::string out = "out.png", base_dir = ".";
        uint64_t seed = 0;
        int device = 0;
        bool reference = false, exact = false, fast = false, bounded = false;
        std::vector<int> devices;
        uint32_t gather = CRT_GATHER_AUTO;
        auto need = [&](int i, int n) { if (i + n >= argc) throw crt::Error(CRT_ERR_INVALID_ARG, std::string("missing value after ") + argv[i]); };
        for (int i = 2; i < argc; i++) {
            std::string a = argv[i];
            if (a == "-o") { need(i, 1); out = argv[++i]; }
            else if (a == "--spp") { need(i, 1); task.spp = (uint32_t)std::atoi(argv[++i]); }
            else if (a == "--p-rr") { need(i, 1); task.p_rr = (float)std::atof(argv[++i]); }
            else if (a == "--lsn") { need(i, 1); task.light_sample_n = (uint32_t)std::atoi(argv[++i]); }
            else if (a == "--seed") { need(i, 1); seed = std::strtoull(argv[++i], nullptr, 10); }
            else if (a == "--width") { need(i, 1); task.width = (uint32_t)std::atoi(argv[++i]); }
            else if (a == "--height") { need(i, 1); task.height = (uint32_t)std::atoi(argv[++i]); }
            else if (a == "--device") { need(i, 1); device = std::atoi(argv[++i]); }
            else if (a == "--gpus") {
                need(i, 1);
                const int n = std::atoi(argv[++i]);
                if (n < 1) throw crt::Error(CRT_ERR_INVALID_ARG, "--gpus needs a positive count");
                devices.clear();
                for (int k = 0; k < n; k++) devices.push_back(k);
            } else if (a == "--devices") {
                need(i, 1);
                devices.clear();
                for (const char* q = argv[++i]; *q;) {
                    char* end = nullptr;
                    const long v = std::strtol(q, &end, 10);
                    if (end == q) throw crt::Error(CRT_ERR_INVALID_ARG, "--devices needs a comma-separated list of device indices");
                    devices.push_back((int)v);
                    q = *end == ',' ? end + 1 : end;
                }
            } else if (a == "--gather") {
                need(i, 1);
                const std::string g = argv[++i];
                if (g == "auto") gather = CRT_GATHER_AUTO; else if (g == "rccl") gather = CRT_GATHER_RCCL; else if (g == "copy") gather = CRT_GATHER_COPY;
                else throw crt::Error(CRT_ERR_INVALID_ARG, "--gather must be auto, rccl or copy");
            }
            else if (a == "--base-dir") { need(i, 1); base_dir = argv[++i]; }
            else if (a == "--reference") reference = true;
            else if (a == "--exact") exact = true;
            else if (a == "--fast") fast = true;
            else if (a == "--bounded-radiance") bounded = true; // CRT_FLAG_BOUNDED_RADIANCE: a ring of samples instead of one radiance per path
            else if (a == "--eye" || a == "--lookat" || a == "--up") {
                need(i, 3);
                float* dst = a == "--eye" ? task.eye_pos : (a == "--lookat" ? task.lookat : task.up);
                for (int k = 0; k < 3; k++) dst[k] = (float)std::atof(argv[++i]);
            } else throw crt::Error(CRT_ERR_INVALID_ARG, "unknown option " + a);
        }
        crt::Scene scene(task.width, task.height);
        crt::load_task_scene(task, scene, base_dir, &all_objs);
        scene.set_BVH(task.bvh_thresh_n);
        std::printf("triangles: %zu, BVH nodes: %zu, lights: %zu\n", scene.get_triangles().size(), scene.get_bvh().get_nodes_size(),
                    scene.get_light_objs().size());
        const bool multi = !devices.empty();
        crt::Render render_one_or_many = multi ? crt::Render(&scene, task.spp, task.p_rr, task.light_sample_n, devices, gather)
                                               : crt::Render(&scene, task.spp, task.p_rr, task.light_sample_n, device);
        crt::Render& render = render_one_or_many;
        render.set_seed(seed);
        render.set_traversal(reference ? CRT_TRAVERSAL_REFERENCE : (fast && !exact) ? CRT_TRAVERSAL_FAST : CRT_TRAVERSAL_EXACT);
        if (bounded) render.set_flags(CRT_FLAG_BOUNDED_RADIANCE);
        float inv_view[9];
        crt::get_inverse_view_matrix(task.eye_pos, task.lookat, task.up, inv_view);
        float fov_y = task.fov_y * (float)M_PI / 180; // src/main.cu:278
        auto t0 = std::chrono::high_resolution_clock::now();
        render.run_view(task.eye_pos, inv_view, fov_y);
        std::chrono::duration<double> dt = std::chrono::high_resolution_clock::now() - t0;
        const crt_stats& st = render.last_stats();
        std::printf("render cost: %.6f seconds (device %.3f ms, %llu rays, %.1f Mrays/s)\n", dt.count(), st.total_ms,
                    (unsigned long long)st.rays, st.total_ms > 0 ? st.rays / st.total_ms / 1e3 : 0.0);
        if (multi) {
            const crt_multi_info& mi = render.last_multi_info();
            std::printf("ranks: %u, gather: %s, rccl ranks: %u (rccl %d), %llu B per rank, render %.3f ms + gather %.3f ms\n", mi.n_ranks,
                        mi.gather == CRT_GATHER_RCCL ? "rccl" : "copy", mi.rccl_ranks, mi.rccl_version, (unsigned long long)mi.bytes_per_rank,
                        mi.render_ms, mi.gather_ms);
            if (mi.fallback_reason[0]) std::printf("gather fell back to peer copies: %s\n", mi.fallback_reason);
        }
        render.save_frame_buffer(out.c_str());
        std::printf("%s\n", out.c_str());
        render.free();
        return 0;
    } catch (const crt::Error& e) {
        std::fprintf(stderr, "crt_cli: %s (%s)\n", e.what(), crt_strerror(e.status));
        return 1;
    }
}
