"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (dd2360-raytracing_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None


def build_oracle(force=False):
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("rt_oracle_capi.cpp", "rt_oracle.hpp")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build_oracle())
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_create_custom.restype = C.c_void_p
        L.orc_scene_create_custom.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_scene_info.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_scene_spheres.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        L.orc_scene_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_scene_world_rng.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_scene_octree_nodes.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        L.orc_scene_octree_leaves.argtypes = [C.c_void_p] + [C.c_void_p] * 2
        L.orc_trace.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.orc_render_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_render_progressive.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_ppm.restype = C.c_int64
        L.orc_ppm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]
        L.orc_xorwow_init.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_xorwow_next.restype = C.c_uint32
        L.orc_xorwow_next.argtypes = [C.c_void_p]
        L.orc_uniform.restype = C.c_float
        L.orc_uniform.argtypes = [C.c_void_p]
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_pow5.restype = C.c_float
        L.orc_pow5.argtypes = [C.c_float]
        L.orc_hw_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    """create_world + buildOctree of the reference (main.cu:146-204, acceleration_structure.h:195) on the CPU."""

    def __init__(self, num_spheres, nx, ny, radius=0.1, fp16=False, use_octree=False, spl=30, custom=None):
        """custom = (geom Nx4, mat Nx4, kind N, camera 22 floats): a caller-supplied world instead of create_world"""
        self.L = lib()
        self.n, self.nx, self.ny, self.fp16, self.use_octree, self.spl = num_spheres, nx, ny, bool(fp16), bool(use_octree), spl
        if custom is None:
            self.h = C.c_void_p(self.L.orc_scene_create(num_spheres, radius, nx, ny, int(fp16), int(use_octree), spl))
        else:
            geom, mat, kind, cam = (np.ascontiguousarray(custom[0], np.float32), np.ascontiguousarray(custom[1], np.float32),
                                    np.ascontiguousarray(custom[2], np.int32), np.ascontiguousarray(custom[3], np.float32))
            assert geom.shape == (num_spheres, 4) and mat.shape == (num_spheres, 4) and kind.shape == (num_spheres,) and cam.size == 22
            self.h = C.c_void_p(self.L.orc_scene_create_custom(num_spheres, _p(geom), _p(mat), _p(kind), _p(cam), nx, ny, int(fp16), int(use_octree), spl))

    def close(self):
        if self.h:
            self.L.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        out = np.zeros(8, np.int64)
        self.L.orc_scene_info(self.h, _p(out))
        keys = ["slots", "real", "world_draws", "node_count", "leaf_count", "leaf_entries", "dropped_full", "dropped_outside"]
        return dict(zip(keys, out.tolist()))

    def spheres(self):
        geom = np.zeros((self.n, 4), np.float32)
        mat = np.zeros((self.n, 4), np.float32)
        kind = np.zeros(self.n, np.int32)
        self.L.orc_scene_spheres(self.h, _p(geom), _p(mat), _p(kind))
        return geom, mat, kind

    def camera(self):
        out = np.zeros(22, np.float32)
        self.L.orc_scene_camera(self.h, _p(out))
        return out

    def world_rng(self):
        out = np.zeros(12, np.uint32)
        self.L.orc_scene_world_rng(self.h, _p(out))
        return out

    def octree(self):
        level = np.zeros(585, np.int32)
        box = np.zeros((585, 6), np.float32)
        children = np.zeros((585, 8), np.int32)
        self.L.orc_scene_octree_nodes(self.h, _p(level), _p(box), _p(children))
        lc = self.info()["leaf_count"]
        counts = np.zeros(lc, np.int32)
        idx = np.zeros((lc, self.spl), np.int32)
        self.L.orc_scene_octree_leaves(self.h, _p(counts), _p(idx))
        return dict(level=level, box=box, children=children, counts=counts, indices=idx)

    def trace(self, rays, mode=0):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        hit = np.zeros(n, np.int32)
        sph = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        p = np.zeros((n, 3), np.float32)
        nrm = np.zeros((n, 3), np.float32)
        self.L.orc_trace(self.h, n, _p(rays), mode, _p(hit), _p(sph), _p(t), _p(p), _p(nrm))
        return dict(hit=hit, sphere=sph, t=t, p=p, normal=nrm)

    def render_init(self, row0=0, rows=None):
        rows = self.ny if rows is None else rows
        st = np.zeros((rows * self.nx, 12), np.uint32)
        self.L.orc_render_init(self.nx, self.ny, row0, rows, _p(st))
        return st

    def render(self, ns, states=None, row0=0, rows=None, nthreads=1, counters=False):
        rows = self.ny if rows is None else rows
        if states is None:
            states = self.render_init(row0, rows)
        fb = np.zeros((rows, self.nx, 3), np.float32)
        cnt = np.zeros(6, np.uint64) if counters else None
        self.L.orc_render(self.h, _p(fb), self.nx, self.ny, ns, _p(states), row0, rows, nthreads, _p(cnt) if counters else None)
        if counters:
            keys = ["rays", "sphere_tests", "slab_tests", "bucket_visits", "draws", "samples"]
            return fb, states, dict(zip(keys, cnt.tolist()))
        return fb, states

    def render_progressive(self, fb, current_sample, states, nthreads=1):
        self.L.orc_render_progressive(self.h, _p(fb), self.nx, self.ny, current_sample, _p(states), nthreads)
        return fb


def ppm_bytes(fb):
    """output_to_stream (main.cu:321-333) as bytes."""
    fb = np.ascontiguousarray(fb, np.float32)
    ny, nx = fb.shape[0], fb.shape[1]
    L = lib()
    n = L.orc_ppm(_p(fb), nx, ny, None, 0)
    buf = C.create_string_buffer(n)
    L.orc_ppm(_p(fb), nx, ny, buf, n)
    return buf.raw[:n]


def xorwow_stream(seed, count):
    L = lib()
    st = np.zeros(12, np.uint32)
    L.orc_xorwow_init(_p(st), seed)
    init = st.copy()
    out = np.array([L.orc_uniform(_p(st)) for _ in range(count)], np.float32)
    return init, out
