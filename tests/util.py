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
    # a few axis-parallel directions (inv_dir = +-inf, DeviceBVH.cuh:97-121 relies on IEEE semantics)
    d[half:half + 6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float32)
    return o, d
