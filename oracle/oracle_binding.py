"""ctypes view of oracle/librtu_oracle.so — TEST INFRASTRUCTURE.

May be imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. The product package never imports this module."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "librtu_oracle.so")
if not os.path.exists(_PATH):
    raise ImportError("%s missing: run `make -C oracle` (or __graft_entry__.build())" % _PATH)
lib = ctypes.CDLL(_PATH)

STAT_FIELDS = ("primary_rays", "primary_hits", "secondary_rays", "shadow_rays", "node_tests", "mesh_entries",
               "inner_visits", "leaf_visits", "leaf_elems", "tri_tests", "tri_accepts")
ERR_ARG, ERR_STOCHASTIC, ERR_UNSUPPORTED = -1, -2, -3


class OracleStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in STAT_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in STAT_FIELDS}


lib.rtu_oracle_render_rows.restype = ctypes.c_int
lib.rtu_oracle_render_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.POINTER(OracleStats), ctypes.c_int]
lib.rtu_oracle_render.restype = ctypes.c_int
lib.rtu_oracle_render.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                  ctypes.POINTER(OracleStats), ctypes.c_int]
lib.rtu_oracle_render_scheduled.restype = ctypes.c_int
lib.rtu_oracle_render_scheduled.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
lib.rtu_oracle_camera_frame.restype = ctypes.c_int
lib.rtu_oracle_camera_frame.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.rtu_oracle_postprocess.restype = None
lib.rtu_oracle_postprocess.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p]


lib.rtu_oracle_render_samples.restype = ctypes.c_int
lib.rtu_oracle_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(OracleStats), ctypes.c_int]
lib.rtu_oracle_render_paths.restype = ctypes.c_int
lib.rtu_oracle_render_paths.argtypes = lib.rtu_oracle_render_samples.argtypes
lib.rtu_oracle_portable_acos.restype = None
lib.rtu_oracle_portable_acos.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
lib.rtu_oracle_portable_sincos.restype = None
lib.rtu_oracle_portable_sincos.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for _n, _k in (("rtu_oracle_rand31", 2), ("rtu_oracle_sample_key", 2), ("rtu_oracle_child_key", 2)):
    getattr(lib, _n).restype = ctypes.c_uint32
    getattr(lib, _n).argtypes = [ctypes.c_uint32] * _k
STREAM_KEYED, STREAM_SEQUENTIAL = 0, 1
TRIG_PORTABLE, TRIG_LIBM = 0, 1


class OracleError(RuntimeError):
    def __init__(self, code):
        self.code = code
        super().__init__("oracle error %d" % code)


def render(scene, width, height, threads=1, row0=0, nrows=None):
    """Recipe W. `scene` is a raytracer_utah_amd.Scene. Returns (rgbz [rows,W,4] float32, stats dict)."""
    if nrows is None:
        nrows = height - row0
    out = np.empty((nrows, width, 4), np.float32)
    st = OracleStats()
    rc = lib.rtu_oracle_render_rows(scene.desc_ptr, width, height, row0, nrows, out.ctypes.data, ctypes.byref(st), threads)
    if rc != 0:
        raise OracleError(rc)
    return out, st.as_dict()


def debug_all_triangles(on):
    """Test hook: the oracle tests every triangle of a mesh whatever its boxes say (not the reference's algorithm)."""
    lib.rtu_oracle_debug_all_triangles.argtypes = [ctypes.c_int]
    lib.rtu_oracle_debug_all_triangles.restype = None
    lib.rtu_oracle_debug_all_triangles(1 if on else 0)


def render_scheduled(scene, width, height, threads, per_pixel):
    """Recipe W, whole frame; per_pixel: the reference's PixelIterator schedule (one atomic fetch per pixel,
    PixelIterator.h:25-38) instead of chunks of rows. Same image, different scaling."""
    out = np.empty((height, width, 4), np.float32)
    st = OracleStats()
    rc = lib.rtu_oracle_render_scheduled(scene.desc_ptr, width, height, out.ctypes.data, ctypes.byref(st), threads, 1 if per_pixel else 0)
    if rc != 0:
        raise OracleError(rc)
    return out, st.as_dict()


def render_samples(scene, width, height, spp, stream=STREAM_KEYED, trig=TRIG_PORTABLE, threads=1, row0=0, nrows=None):
    """Recipe S (row f1): spp samples per pixel with soft shadows / glossy bounces / depth of field."""
    if nrows is None:
        nrows = height - row0
    out = np.empty((nrows, width, 4), np.float32)
    st = OracleStats()
    rc = lib.rtu_oracle_render_samples(scene.desc_ptr, width, height, row0, nrows, spp, stream, trig, out.ctypes.data,
                                       ctypes.byref(st), threads)
    if rc != 0:
        raise OracleError(rc)
    return out, st.as_dict()


def render_paths(scene, width, height, spp, stream=STREAM_KEYED, trig=TRIG_PORTABLE, threads=1, row0=0, nrows=None):
    """Recipe P (config 5): recipe S plus the 4-bounce Monte-Carlo gather."""
    if nrows is None:
        nrows = height - row0
    out = np.empty((nrows, width, 4), np.float32)
    st = OracleStats()
    rc = lib.rtu_oracle_render_paths(scene.desc_ptr, width, height, row0, nrows, spp, stream, trig, out.ctypes.data,
                                     ctypes.byref(st), threads)
    if rc != 0:
        raise OracleError(rc)
    return out, st.as_dict()


def portable_acos(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib.rtu_oracle_portable_acos(x.ctypes.data, x.size, out.ctypes.data)
    return out


def portable_sincos(t):
    t = np.ascontiguousarray(t, np.float32)
    s, c = np.empty_like(t), np.empty_like(t)
    lib.rtu_oracle_portable_sincos(t.ctypes.data, t.size, s.ctypes.data, c.ctypes.data)
    return s, c


def camera_frame(camera, width, height):
    out = np.empty(12, np.float32)
    rc = lib.rtu_oracle_camera_frame(ctypes.addressof(camera), width, height, out.ctypes.data)
    if rc != 0:
        raise OracleError(rc)
    return out.reshape(4, 3)  # pos, origin, u, v


def postprocess(rgbz):
    """gamma + Color24 + z-image: returns (rgb8 [H,W,3], z [H,W], zimg8 [H,W])."""
    a = np.ascontiguousarray(rgbz, np.float32)
    h, w = a.shape[:2]
    rgb = np.empty((h, w, 3), np.uint8)
    z = np.empty((h, w), np.float32)
    zi = np.empty((h, w), np.uint8)
    lib.rtu_oracle_postprocess(a.ctypes.data, w, h, rgb.ctypes.data, z.ctypes.data, zi.ctypes.data)
    return rgb, z, zi
