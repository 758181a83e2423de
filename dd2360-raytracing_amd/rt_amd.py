"""ctypes binding of librt_amd.so (include/rt_amd.h) — the thin Python layer used by tests/, bench.py and
__graft_entry__.py.  The product is the C-ABI library and the C++ host program (host/main.cpp); this module only
marshals arguments.  Device memory, streams and process groups come from PyTorch (plumbing only).

There is no fallback of any kind: if librt_amd.so cannot be loaded, or a call returns non-zero, RtError is raised.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB", os.path.join(_HERE, "librt_amd.so"))   # RT_AMD_LIB: tuning experiments only

FP32, FP16 = 0, 1
MAT_NONE, MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC = -1, 0, 1, 2
OCTREE_MAX_NODES = 585
TRAVERSAL_REFERENCE, TRAVERSAL_FAST = 0, 1
ARITH_IEEE, ARITH_CONTRACT = 0, 1
IMAGE_P3, IMAGE_P6, IMAGE_PFM = 0, 1, 2

# PODs of include/rt_amd.h
rand_state_dtype = np.dtype([("d", "<u4"), ("v", "<u4", 5), ("boxmuller_flag", "<i4"), ("boxmuller_flag_double", "<i4"),
                             ("boxmuller_extra", "<f4"), ("pad_", "<u4"), ("boxmuller_extra_double", "<f8")])
sphere_dtype = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("material", "<i4"), ("albedo", "<f4", 3), ("param", "<f4")])
camera_dtype = np.dtype([("origin", "<f4", 3), ("lower_left_corner", "<f4", 3), ("horizontal", "<f4", 3), ("vertical", "<f4", 3),
                         ("u", "<f4", 3), ("v", "<f4", 3), ("w", "<f4", 3), ("lens_radius", "<f4")])
octnode_dtype = np.dtype([("level", "<i4"), ("aabb", "<f4", 6), ("children", "<i4", 8)])
hit_record_dtype = np.dtype([("t", "<f4"), ("p", "<f4", 3), ("normal", "<f4", 3), ("sphere", "<i4")])
assert rand_state_dtype.itemsize == 48 and sphere_dtype.itemsize == 36 and camera_dtype.itemsize == 88
assert octnode_dtype.itemsize == 60 and hit_record_dtype.itemsize == 32


class RtError(RuntimeError):
    pass


class Partition(C.Structure):
    """rt_partition: part `part` of `nparts`; tile_end > tile_begin: the tile range [tile_begin, tile_end) (a band of rt_split_balanced),
    otherwise runs of PART_RUN tiles dealt round-robin"""
    _fields_ = [("part", C.c_int32), ("nparts", C.c_int32), ("tile_begin", C.c_int64), ("tile_end", C.c_int64)]


WHOLE = Partition(0, 1)
PART_RUN = 64          # rt_amd.h RT_PART_RUN: consecutive tiles per run of the tile split (tests restate the split with it)

# every symbol include/rt_amd.h declares: (restype, argtypes)
_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
SYMBOLS = {
    "rt_abi_version": (_i, []),
    "rt_device_check": (_i, [_vp]),
    "rt_error_string": (C.c_char_p, [_i]),
    "rt_rand_init": (_i, [_vp]),
    "rt_create_world": (_i, [_vp, _i, _f, _vp, _i, _i, _vp, _i, _vp]),
    "rt_camera_init": (_i, [_vp, _vp, _vp, _vp, _f, _f, _f, _f, _i]),
    "rt_world_create": (_i, [_vp, _i, _vp, _i, _vp]),
    "rt_world_upload": (_i, [_vp]),
    "rt_free_world": (_i, [_vp]),
    "rt_build_octree": (_i, [_vp, _i, _i, _i, _vp]),
    "rt_octree_upload": (_i, [_vp]),
    "rt_build_octree_gpu": (_i, [_vp, _i, _vp, _vp]),
    "rt_octree_debug_array": (_i, [_vp, _i, _vp, C.c_size_t, _vp]),
    "rt_free_octree": (_i, [_vp]),
    "rt_octree_info": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "rt_octree_flat_info": (_i, [_vp, _vp, _vp]),
    "rt_octree_set_traversal": (_i, [_vp, _i]),
    "rt_world_set_list_traversal": (_i, [_vp, _i]),
    "rt_world_set_arith": (_i, [_vp, _i]),
    "rt_world_list_accel_info": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "rt_octree_accel_info": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "rt_octree_nodes": (_i, [_vp, _vp]),
    "rt_octree_leaves": (_i, [_vp, _vp, _vp]),
    "rt_part_pixels": (_i64, [_i, _i, Partition]),
    "rt_render_init": (_i, [_i, _i, _vp, Partition, _vp]),
    "rt_render": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, Partition, _vp]),
    "rt_render_progressive": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, Partition, _vp]),
    "rt_world_render_times": (_i, [_vp, _vp, _i, _vp]),
    "rt_render_ctx_create": (_i, [_vp]),
    "rt_render_ctx_reserve": (_i, [_vp, _i, _i, Partition]),
    "rt_render_ctx_destroy": (_i, [_vp]),
    "rt_render_on": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, Partition, _vp]),
    "rt_render_progressive_on": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, Partition, _vp]),
    "rt_render_ctx_times": (_i, [_vp, _vp, _i, _vp]),
    "rt_render_ctx_counters": (_i, [_vp, _vp]),
    "rt_world_render_counters": (_i, [_vp, _vp]),
    "rt_render_kernel_name": (_i, [_vp, _vp, _i, _vp, _i]),
    "rt_multi_unique_id": (_i, [_vp]),
    "rt_multi_init": (_i, [_vp, _i, _i, _vp]),
    "rt_multi_probe": (_i, []),
    "rt_multi_init_custom": (_i, [_vp, _i, _i, _vp, _vp]),
    "rt_multi_destroy": (_i, [_vp]),
    "rt_multi_reserve": (_i, [_vp, _i, _i, _i, _i]),
    "rt_multi_render": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "rt_multi_last_render_ms": (_i, [_vp, _vp, _vp]),
    "rt_multi_selftest": (_i, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "rt_assemble": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rt_assemble_split": (_i, [_vp, _vp, _i, _i, _i, _vp, _i64, _i, _vp]),
    "rt_split_balanced": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rt_multi_set_split": (_i, [_vp, _i]),
    "rt_multi_last_split": (_i, [_vp, _vp]),
    "rt_trace_rays": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "rt_write_ppm": (_i, [C.c_char_p, _i, _i, _vp, _i]),
    "rt_format_ppm": (_i64, [_i, _i, _vp, _i, _vp, _i64]),
    "rt_write_image": (_i, [C.c_char_p, _i, _i, _vp, _i, _i]),
}

_LIB = None


def lib():
    """Load librt_amd.so.  Raises RtError when it is missing or does not export the whole ABI (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RtError("librt_amd.so is not built (%s): run __graft_entry__.build() / make -C dd2360-raytracing_amd" % LIB_PATH)
        # One HIP runtime per process: PyTorch bundles its own libamdhip64.so (same soname as /opt/rocm's).  Import it
        # first so that librt_amd.so's NEEDED libamdhip64.so.7 resolves to the copy torch already mapped; two runtimes
        # in one process leave the second one without devices (hipErrorNoDevice).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:
            raise RtError("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                raise RtError("librt_amd.so does not export %s" % name)
            fn.restype, fn.argtypes = res, args
        _LIB = L
    return _LIB


def check(rc, what):
    if rc != 0:
        msg = lib().rt_error_string(int(rc))
        raise RtError("%s failed: %d (%s)" % (what, rc, msg.decode() if msg else "?"))


def _np(a):
    return a.ctypes.data_as(C.c_void_p)


def _dev(t):
    """device pointer of a torch tensor (or a raw int)"""
    return C.c_void_p(t if isinstance(t, int) else t.data_ptr())


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def render_kernel_name(world, octree=None, mode=0):
    """the kernel rt_render (mode 0) / rt_render_progressive (mode 1) launches for this world and tree (the library's own rule)"""
    buf = C.create_string_buffer(64)
    check(lib().rt_render_kernel_name(world.h, octree.h if octree is not None else None, mode, buf, 64), "rt_render_kernel_name")
    return buf.value.decode()


class RenderCtx:
    """rt_render_ctx: per-launch state for frames rendered concurrently on one GPU (one context per stream)"""

    def __init__(self):
        h = C.c_void_p()
        check(lib().rt_render_ctx_create(C.byref(h)), "rt_render_ctx_create")
        self.h = h

    def reserve(self, max_x, max_y, part=None):
        check(lib().rt_render_ctx_reserve(self.h, max_x, max_y, part or WHOLE), "rt_render_ctx_reserve")
        return self

    def render(self, fb, max_x, max_y, ns, world, d_rand_state, octree=None, part=None, stream=None):
        check(lib().rt_render_on(self.h, _dev(fb), max_x, max_y, ns, world.h, _dev(d_rand_state), octree.h if octree is not None else None,
                                 part or WHOLE, C.c_void_p(stream) if stream is not None else _stream()), "rt_render_on")

    def times(self):
        out = np.zeros(64, np.float32)
        n = C.c_int(0)
        check(lib().rt_render_ctx_times(self.h, _np(out), 64, C.byref(n)), "rt_render_ctx_times")
        return out[: n.value].tolist()

    def counters(self):
        """scheduling counters of the latest launch: slots handed out, thin waves left, long chains pre-classified, handles taken"""
        out = np.zeros(4, np.uint32)
        check(lib().rt_render_ctx_counters(self.h, _np(out)), "rt_render_ctx_counters")
        return dict(zip(("slots", "thin_waves", "long_chains", "long_handles"), (int(v) for v in out)))

    def close(self):
        if getattr(self, "h", None):
            lib().rt_render_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


SPLIT_RUNS, SPLIT_BALANCED, SPLIT_BALANCED_CACHED = 0, 1, 2      # rt_multi_set_split
GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
MULTI_ID_BYTES = 128


def multi_unique_id():
    """rank 0: the RCCL unique id (bytes) to hand to the other ranks"""
    buf = C.create_string_buffer(MULTI_ID_BYTES)
    check(lib().rt_multi_unique_id(buf), "rt_multi_unique_id")
    return buf.raw


def multi_probe():
    """rt_multi_probe: 0 when this rank could enter rt_multi_init's collective part (RCCL bound, context on the current device)"""
    return int(lib().rt_multi_probe())


class Multi:
    """rt_multi: tile split over the ranks of a node + the single framebuffer exchange + rt_assemble on the root.
    unique_id (bytes) selects RCCL; gather (a Python callable with the rt_gather_fn arguments) a custom exchange."""

    def __init__(self, rank, nranks, unique_id=None, gather=None):
        h = C.c_void_p()
        if gather is not None:
            self._cb = GATHER_FN(gather)                          # keep the thunk alive
            check(lib().rt_multi_init_custom(C.byref(h), rank, nranks, C.cast(self._cb, C.c_void_p), None), "rt_multi_init_custom")
        else:
            assert unique_id is not None and len(unique_id) == MULTI_ID_BYTES
            self._id = C.create_string_buffer(unique_id, MULTI_ID_BYTES)
            check(lib().rt_multi_init(C.byref(h), rank, nranks, self._id), "rt_multi_init")
        self.h, self.rank, self.nranks = h, rank, nranks
        self.split_mode = SPLIT_RUNS                             # the library's default (rt_multi_set_split)

    def reserve(self, max_x, max_y, precision=FP32, root=0):
        check(lib().rt_multi_reserve(self.h, max_x, max_y, precision, root), "rt_multi_reserve")
        return self

    def render(self, fb_full, max_x, max_y, ns, world, octree=None, root=0, precision=None):
        check(lib().rt_multi_render(self.h, _dev(fb_full) if fb_full is not None else None, max_x, max_y, ns, world.h,
                                    octree.h if octree is not None else None, world.precision if precision is None else precision,
                                    root, _stream()), "rt_multi_render")

    def set_split(self, mode):
        check(lib().rt_multi_set_split(self.h, mode), "rt_multi_set_split")
        self.split_mode = mode
        return self

    def last_split(self):
        st = (C.c_int64 * (self.nranks + 1))()
        check(lib().rt_multi_last_split(self.h, st), "rt_multi_last_split")
        return list(st)

    def last_render_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        check(lib().rt_multi_last_render_ms(self.h, C.byref(a), C.byref(b)), "rt_multi_last_render_ms")
        return a.value, b.value

    def selftest(self, d_src, d_dst, nbytes):
        check(lib().rt_multi_selftest(self.h, _dev(d_src), _dev(d_dst), nbytes, _stream()), "rt_multi_selftest")

    def close(self):
        if getattr(self, "h", None):
            lib().rt_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_check():
    n = C.c_int(0)
    rc = lib().rt_device_check(C.byref(n))
    return rc, n.value


def split_balanced(world, octree, max_x, max_y, nparts, counts=False):
    """rt_split_balanced: starts[nparts + 1] of the bands of equal predicted cost (and, counts=True, the pilot's per-tile bounces, tests and columns)"""
    import numpy as np
    tiles = ((max_x + 7) // 8) * ((max_y + 7) // 8)
    st = (C.c_int64 * (nparts + 1))()
    b = np.zeros(tiles, np.int32) if counts else None
    t = np.zeros(tiles, np.int32) if counts else None
    c = np.zeros(tiles, np.int32) if counts else None
    check(lib().rt_split_balanced(None, world.h, octree.h if octree is not None else None, max_x, max_y, nparts, st,
                                  b.ctypes.data_as(C.c_void_p) if counts else None, t.ctypes.data_as(C.c_void_p) if counts else None,
                                  c.ctypes.data_as(C.c_void_p) if counts else None, _stream()), "rt_split_balanced")
    return (list(st), b, t, c) if counts else list(st)


def split_parts(starts):
    """the rt_partition of every band of a split"""
    n = len(starts) - 1
    return [Partition(p, n, starts[p], starts[p + 1]) for p in range(n)]


def assemble_split(fb_full, fb_parts, max_x, max_y, starts, part_stride_px, precision=FP32):
    n = len(starts) - 1
    st = (C.c_int64 * (n + 1))(*starts)
    check(lib().rt_assemble_split(_dev(fb_full), _dev(fb_parts), max_x, max_y, n, st, part_stride_px, precision, _stream()), "rt_assemble_split")


def part_pixels(max_x, max_y, part=WHOLE):
    n = lib().rt_part_pixels(max_x, max_y, part)
    if n < 0:
        raise RtError("rt_part_pixels: invalid argument")
    return n


class World:
    """rand_init + create_world of main.cu:388-401 (host side), and the device scene they feed."""

    def __init__(self, num_spheres, nx, ny, sphere_radius=0.1, precision=FP32, spheres=None, camera=None):
        L = lib()
        self.num_spheres, self.nx, self.ny, self.precision = num_spheres, nx, ny, precision
        self.rand_state = np.zeros(1, rand_state_dtype)
        if spheres is None:
            self.spheres = np.zeros(num_spheres, sphere_dtype)
            self.camera = np.zeros(1, camera_dtype)
            created = C.c_int(0)
            check(L.rt_rand_init(_np(self.rand_state)), "rt_rand_init")
            check(L.rt_create_world(_np(self.spheres), num_spheres, sphere_radius, _np(self.camera), nx, ny, _np(self.rand_state),
                                    precision, C.byref(created)), "rt_create_world")
            self.created = created.value
        else:
            self.spheres = np.ascontiguousarray(spheres, sphere_dtype)
            self.camera = np.ascontiguousarray(camera, camera_dtype).reshape(1)
            self.created = int((self.spheres["material"] != MAT_NONE).sum())
        h = C.c_void_p()
        check(L.rt_world_create(_np(self.spheres), num_spheres, _np(self.camera), precision, C.byref(h)), "rt_world_create")
        self.h = h

    def upload(self):
        check(lib().rt_world_upload(self.h), "rt_world_upload")
        return self

    def set_list_traversal(self, mode):
        """TRAVERSAL_REFERENCE (every sphere in list order) or TRAVERSAL_FAST (default: the candidate grid) for renders without an octree"""
        check(lib().rt_world_set_list_traversal(self.h, mode), "rt_world_set_list_traversal")
        return self

    def set_arith(self, mode):
        """ARITH_IEEE (default, the parity contract) or ARITH_CONTRACT (FMA contraction allowed: a tolerance mode)"""
        check(lib().rt_world_set_arith(self.h, mode), "rt_world_set_arith")
        return self

    def list_accel_info(self):
        e, g, h, n, l = C.c_int(0), C.c_int(0), C.c_float(0), C.c_int(0), C.c_int(0)
        check(lib().rt_world_list_accel_info(self.h, C.byref(e), C.byref(g), C.byref(h), C.byref(n), C.byref(l)), "rt_world_list_accel_info")
        return {"enabled": bool(e.value), "grid_dim": g.value, "cell_size": h.value, "grid_entries": n.value, "large_spheres": l.value}

    def render_times(self):
        """device times (ms) of the render kernel of the calls since the last query (HIP events on the launch stream)"""
        out = np.zeros(64, np.float32)
        n = C.c_int(0)
        check(lib().rt_world_render_times(self.h, _np(out), 64, C.byref(n)), "rt_world_render_times")
        return out[: n.value].tolist()

    def render_counters(self):
        """scheduling counters of the latest render() on this world (see RenderCtx.counters)"""
        out = np.zeros(4, np.uint32)
        check(lib().rt_world_render_counters(self.h, _np(out)), "rt_world_render_counters")
        return dict(zip(("slots", "thin_waves", "long_chains", "long_handles"), (int(v) for v in out)))

    def close(self):
        if getattr(self, "h", None):
            lib().rt_free_world(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Octree:
    """buildOctree (acceleration_structure.h:195) + upload (main.cu:413-417)."""

    def __init__(self, world, spheres_per_leaf=30, gpu=False):
        """gpu=True: rt_build_octree_gpu (built on the device from the world's device copy, ready to render)"""
        L = lib()
        self.world, self.spl = world, spheres_per_leaf
        h = C.c_void_p()
        if gpu:
            check(L.rt_build_octree_gpu(world.h, spheres_per_leaf, C.byref(h), _stream()), "rt_build_octree_gpu")
        else:
            check(L.rt_build_octree(_np(world.spheres), world.num_spheres, spheres_per_leaf, world.precision, C.byref(h)), "rt_build_octree")
        self.h = h

    def upload(self):
        check(lib().rt_octree_upload(self.h), "rt_octree_upload")
        return self

    def set_traversal(self, mode):
        """TRAVERSAL_REFERENCE (exact bucket scan) or TRAVERSAL_FAST (default, culling grid + fallback)"""
        check(lib().rt_octree_set_traversal(self.h, mode), "rt_octree_set_traversal")
        return self

    def device_array(self, which):
        """bytes of one device-resident array of the (uploaded) tree, see rt_octree_debug_array"""
        n = C.c_size_t(0)
        check(lib().rt_octree_debug_array(self.h, which, None, 0, C.byref(n)), "rt_octree_debug_array")
        buf = np.zeros(max(1, n.value), np.uint8)
        check(lib().rt_octree_debug_array(self.h, which, _np(buf), n.value, C.byref(n)), "rt_octree_debug_array")
        return buf[: n.value]

    def accel_info(self):
        g, n, l = C.c_int(0), C.c_int(0), C.c_int(0)
        h = C.c_float(0)
        check(lib().rt_octree_accel_info(self.h, C.byref(g), C.byref(h), C.byref(n), C.byref(l)), "rt_octree_accel_info")
        return dict(grid_dim=g.value, cell_size=h.value, grid_entries=n.value, large_spheres=l.value)

    def info(self):
        v = [C.c_int(0) for _ in range(5)]
        check(lib().rt_octree_info(self.h, *[C.byref(x) for x in v]), "rt_octree_info")
        fn, fe = C.c_int(0), C.c_int(0)
        check(lib().rt_octree_flat_info(self.h, C.byref(fn), C.byref(fe)), "rt_octree_flat_info")
        keys = ["node_count", "leaf_count", "spl", "dropped_full", "dropped_outside"]
        d = dict(zip(keys, [x.value for x in v]))
        d.update(flat_nodes=fn.value, flat_entries=fe.value)
        return d

    def nodes(self):
        out = np.zeros(OCTREE_MAX_NODES, octnode_dtype)
        check(lib().rt_octree_nodes(self.h, _np(out)), "rt_octree_nodes")
        return out

    def leaves(self):
        lc = self.info()["leaf_count"]
        counts = np.zeros(lc, np.int32)
        idx = np.zeros((lc, self.spl), np.int32)
        check(lib().rt_octree_leaves(self.h, _np(counts), _np(idx)), "rt_octree_leaves")
        return counts, idx

    def close(self):
        if getattr(self, "h", None):
            lib().rt_free_octree(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_init(lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist, precision=FP32):
    cam = np.zeros(1, camera_dtype)
    a = [np.asarray(x, np.float32) for x in (lookfrom, lookat, vup)]
    check(lib().rt_camera_init(_np(cam), _np(a[0]), _np(a[1]), _np(a[2]), vfov, aspect, aperture, focus_dist, precision), "rt_camera_init")
    return cam


# ---- device-side calls: buffers are torch CUDA tensors -------------------------------------------------------------
def alloc_rand_state(max_x, max_y, part=WHOLE, device="cuda"):
    import torch
    return torch.zeros(part_pixels(max_x, max_y, part) * 48, dtype=torch.uint8, device=device)


def alloc_fb(max_x, max_y, part=WHOLE, precision=FP32, device="cuda"):
    import torch
    return torch.zeros(part_pixels(max_x, max_y, part) * 3, dtype=torch.float32 if precision == FP32 else torch.float16, device=device)


def render_init(max_x, max_y, d_rand_state, part=WHOLE):
    check(lib().rt_render_init(max_x, max_y, _dev(d_rand_state), part, _stream()), "rt_render_init")


def render(fb, max_x, max_y, ns, world, d_rand_state, octree=None, part=WHOLE):
    check(lib().rt_render(_dev(fb), max_x, max_y, ns, world.h, _dev(d_rand_state), octree.h if octree is not None else None, part, _stream()),
          "rt_render")


def render_progressive(fb, max_x, max_y, current_sample, world, d_rand_state, octree=None, part=WHOLE):
    check(lib().rt_render_progressive(_dev(fb), max_x, max_y, current_sample, world.h, _dev(d_rand_state),
                                      octree.h if octree is not None else None, part, _stream()), "rt_render_progressive")


def assemble(fb_full, fb_parts, max_x, max_y, nparts, precision=FP32):
    check(lib().rt_assemble(_dev(fb_full), _dev(fb_parts), max_x, max_y, nparts, precision, _stream()), "rt_assemble")


def trace_rays(world, octree, d_rays, n, d_out):
    check(lib().rt_trace_rays(world.h, octree.h if octree is not None else None, _dev(d_rays), n, _dev(d_out), _stream()), "rt_trace_rays")


def write_image(path, fb_host, nx, ny, precision=FP32, fmt=IMAGE_P6):
    fb_host = np.ascontiguousarray(fb_host)
    check(lib().rt_write_image(str(path).encode(), nx, ny, _np(fb_host), precision, fmt), "rt_write_image")


def format_ppm(fb_host, nx, ny, precision=FP32):
    fb_host = np.ascontiguousarray(fb_host)
    cap = nx * ny * 36 + 64                                      # three ints of <= 11 characters per pixel: formatted once
    buf = C.create_string_buffer(cap)
    n = lib().rt_format_ppm(nx, ny, _np(fb_host), precision, buf, cap)
    if n < 0 or n > cap:
        raise RtError("rt_format_ppm failed: %d" % n)
    return buf.raw[:n]
