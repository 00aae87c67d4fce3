import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
for scene in ("cornell-box", "veach-mis"):
    t = crt.Task(os.path.join(ROOT, "scenes", scene, "config.json"), base_dir=ROOT)
    sc = crt.Scene.from_task(t)
    r = crt.Render(sc, 8, t.P_RR, t.light_sample_n)
    iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up); fov = crt.fov_to_radians(t.fov_y)
    for mode in (crt.TRAVERSAL_EXACT, crt.TRAVERSAL_FAST, crt.TRAVERSAL_REFERENCE):
        r.traversal = mode
        r.run_view(t.eye_pos, iv, fov, stats=True, want_mean=False)
        s = r.stats
        print(scene, {crt.TRAVERSAL_EXACT: "exact", crt.TRAVERSAL_FAST: "fast", crt.TRAVERSAL_REFERENCE: "ref"}[mode], {k: round(s[k] / s["rays"], 2) for k in ("inner_pops", "leaf_pops", "tri_tests", "hits", "stack_sum")}, "stack_max", s["stack_max"], "rays", s["rays"])
