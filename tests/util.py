"""Shared helpers for the test-suite (scene setup for both the product and the oracle)."""
import functools
import os

import numpy as np

import cudaraytracing_amd as crt
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = {"cornell-box": os.path.join(ROOT, "scenes", "cornell-box", "config.json"),
          "veach-mis": os.path.join(ROOT, "scenes", "veach-mis", "config.json")}


@functools.lru_cache(maxsize=None)
def task(name):
    return crt.Task(SCENES[name], base_dir=ROOT)


@functools.lru_cache(maxsize=None)
def host_scene(name):
    return crt.Scene.from_task(task(name))


@functools.lru_cache(maxsize=None)
def oracle_scene(name):
    t = task(name)
    return O.OracleScene(t.OBJ_paths, t.bvh_thresh_n)


def camera(name):
    t = task(name)
    return t.eye_pos, crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up), crt.fov_to_radians(t.fov_y)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def random_rays(name, n, seed):
    """Rays from inside the scene bounds towards random directions plus camera-like rays."""
    rng = np.random.default_rng(seed)
    nodes = oracle_scene(name).nodes()
    root = nodes[oracle_scene(name).root]
    lo, hi = root["aa"], root["bb"]
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    eye = np.asarray(task(name).eye_pos, dtype=np.float32)
    half = n // 2
    o[:half] = eye
    tgt = (lo + (hi - lo) * rng.random((half, 3))).astype(np.float32)
    d[:half] = tgt - eye
    # axis-parallel directions and directions with one zero component (inv_dir = +-inf, DeviceBVH.cuh:97-121
    # relies on IEEE inf/NaN semantics), some starting exactly on a box plane (0 * inf = NaN in the slab test)
    d[half:half + 6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float32)
    k = min(n // 8, 256)
    z = d[half + 6:half + 6 + k]
    z[np.arange(k), rng.integers(0, 3, k)] = 0.0
    z[::3, rng.integers(0, 3)] = 0.0
    ob = o[half + 6:half + 6 + k]
    pick = nodes[rng.integers(0, len(nodes), k)]
    ax = rng.integers(0, 3, k)
    side = rng.integers(0, 2, k)
    ob[np.arange(k), ax] = np.where(side == 0, pick["aa"][np.arange(k), ax], pick["bb"][np.arange(k), ax])
    return o, d
