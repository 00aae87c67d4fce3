#!/usr/bin/env python3
"""Per-wave times of k_mega3 launches (s_memrealtime at a wave's start, at the start of its last path, at its first exhausted cursor
and at its end): where the end of a launch goes.  Needs a DIAGNOSTIC build of the library: apply
tools/wave_times.patch (the kernel then writes four words per wave into a buffer whose address travels in MParams3::dbg_loads /
dbg_valu, the host dumps it to $CRT_WAVE_TIMES after every launch), build it with tools/ab_build.sh and point CRT_LIB_PATH at it:
    git apply tools/wave_times.patch && tools/ab_build.sh wt && git checkout cudaraytracing_amd/csrc/crt_kernels.hip
    CRT_LIB_PATH=$PWD/cudaraytracing_amd/lib/ab/wt.so CRT_WAVE_TIMES=/tmp/wt.bin python3 tools/wave_times.py"""
import os, sys, json, numpy as np
ROOT='/root/repo' if os.path.isdir('/root/repo/tools') else os.getcwd()
sys.path.insert(0, ROOT)
import torch, cudaraytracing_amd as crt
t = crt.Task(os.path.join(ROOT, "scenes", "cornell-box", "config.json"), base_dir=ROOT)
sc = crt.Scene.from_task(t, 800, 600)
iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up); fov = crt.fov_to_radians(t.fov_y)
dev = torch.device("cuda:0")
for world, spp in ((8, 512), (1, 64), (1, 512)):
    slots = crt.shard_slots(800, 600, 0, world)
    local = torch.empty((slots, 3), dtype=torch.uint8, device=dev)
    r = crt.Render(sc, spp, t.P_RR, t.light_sample_n)
    for rep in range(3):
        st = r.run_view_device(t.eye_pos, iv, fov, local.data_ptr(), None, None, rank=0, world=world, tiled=True, want_stats=True, width=800, height=600)
        torch.cuda.synchronize()
    a = np.fromfile(os.environ["CRT_WAVE_TIMES"], dtype=np.uint64).reshape(-1, 4).astype(np.float64)
    t0 = a[:, 0].min(); us = lambda x: (x - t0) / 100.0   # 100 MHz -> us
    start, dry, end, last = us(a[:, 0]), us(a[:, 1]), us(a[:, 2]), us(a[:, 3])
    dry = np.where(a[:, 1] == 0, end, dry)
    q = lambda v: [round(float(x), 1) for x in np.percentile(v, [0, 10, 50, 90, 99, 100])]
    print(json.dumps({"world": world, "spp": spp, "kernel_ms": round(st["kernel_ms"], 3), "waves": int(len(a)),
                      "start_us": q(start), "last_path_started_us": q(last), "first_dry_us": q(dry), "end_us": q(end),
                      "end_minus_last_start_us": q(end - last), "kernel_end_minus_wave_end_us": q(end.max() - end)}), flush=True)
    r.free()
