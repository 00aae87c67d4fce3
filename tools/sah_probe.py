import os, sys, time
sys.path.insert(0, "/root/repo")
import cudaraytracing_amd as crt
ROOT="/root/repo"
for name in ("cornell-box", "cornell-box", "veach-mis"):
    t = crt.Task(os.path.join(ROOT, "scenes", name, "config.json"), base_dir=ROOT)
    sc = crt.Scene.from_task(t, 800, 600)
    t0=time.perf_counter()
    r = crt.Render(sc, 1, t.P_RR, t.light_sample_n)
    print(name, "Render() %.1f ms" % ((time.perf_counter()-t0)*1e3), r.accel_info())
    r.free()
