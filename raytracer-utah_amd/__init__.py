"""raytracer-utah_amd — MI355X-native render hot path for RayTracer-Utah scenes.

Python here is plumbing only (tests, bench.py, torch.distributed glue): a ctypes
view of the two product libraries

  lib/librtu_hip.so   C-ABI of include/rtu_render.h  (hand-written gfx950 HIP kernels)
  lib/librtu_host.so  C entry points of include/rtu_host.h (scene loader, flattener,
                      RenderImage mirror, PNG, BeginRender)

There is NO CPU fallback: if the HIP library is missing or cannot be loaded the
import fails loudly (build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C raytracer-utah_amd`). The CPU oracle lives under oracle/ and is test
infrastructure; nothing in this package touches it.

The directory name contains a hyphen, so it is imported by path as module
`raytracer_utah_amd` (see __graft_entry__.load_package()).
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_PKG, "lib")

RTU_BIGFLOAT = 1.0e30
RTU_BAND_ROWS = 8

RTU_OK = 0
RTU_ERR_ARG = -1
RTU_ERR_HIP = -2
RTU_ERR_UNSUPPORTED = -3
RTU_ERR_STOCHASTIC = -4
RTU_ERR_NO_SCENE = -5
RTU_ERR_NO_DEVICE = -6
RTU_ERR_CAPACITY = -7
RTU_ERR_CANCELLED = -8


class RtuError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__("rtu error %d: %s" % (code, msg))


class RtuCamera(ctypes.Structure):
    _fields_ = [("pos", ctypes.c_float * 3), ("dir", ctypes.c_float * 3), ("up", ctypes.c_float * 3),
                ("fov", ctypes.c_float), ("focaldist", ctypes.c_float), ("dof", ctypes.c_float),
                ("img_width", ctypes.c_int32), ("img_height", ctypes.c_int32)]


class RtuEnvColor(ctypes.Structure):
    _fields_ = [("color", ctypes.c_float * 3), ("has_map", ctypes.c_int32), ("map_is_null", ctypes.c_int32),
                ("reserved", ctypes.c_int32 * 3)]


class RtuTexMap(ctypes.Structure):
    _fields_ = [("present", ctypes.c_int32), ("texture", ctypes.c_int32), ("tm", ctypes.c_float * 9), ("itm", ctypes.c_float * 9),
                ("pos", ctypes.c_float * 3), ("reserved", ctypes.c_int32)]


class RtuMesh(ctypes.Structure):
    _fields_ = [("nv", ctypes.c_uint32), ("nf", ctypes.c_uint32), ("nvn", ctypes.c_uint32), ("nvt", ctypes.c_uint32),
                ("n_bvh_nodes", ctypes.c_uint32), ("n_elements", ctypes.c_uint32), ("bvh_depth", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32), ("bound_min", ctypes.c_float * 3), ("bound_max", ctypes.c_float * 3),
                ("v", ctypes.c_void_p), ("f", ctypes.c_void_p), ("vn", ctypes.c_void_p), ("fn", ctypes.c_void_p),
                ("vt", ctypes.c_void_p), ("ft", ctypes.c_void_p), ("bvh", ctypes.c_void_p), ("elements", ctypes.c_void_p)]


class RtuSceneDesc(ctypes.Structure):
    _fields_ = [("n_nodes", ctypes.c_uint32), ("n_materials", ctypes.c_uint32), ("n_lights", ctypes.c_uint32),
                ("n_meshes", ctypes.c_uint32), ("nodes", ctypes.c_void_p), ("materials", ctypes.c_void_p),
                ("lights", ctypes.c_void_p), ("meshes", ctypes.c_void_p), ("camera", RtuCamera),
                ("background", RtuEnvColor), ("environment", RtuEnvColor),
                ("n_textures", ctypes.c_uint32), ("reserved0", ctypes.c_uint32), ("textures", ctypes.c_void_p),
                ("material_maps", ctypes.c_void_p), ("background_map", RtuTexMap), ("environment_map", RtuTexMap)]


class RtuFrameDesc(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("shard_rank", ctypes.c_int32),
                ("shard_count", ctypes.c_int32), ("max_bounce", ctypes.c_int32), ("collect_stats", ctypes.c_int32),
                ("coop_threshold", ctypes.c_int32), ("samples", ctypes.c_int32), ("cam_pos", ctypes.c_float * 3), ("origin", ctypes.c_float * 3),
                ("u", ctypes.c_float * 3), ("v", ctypes.c_float * 3), ("lens_up", ctypes.c_float * 3), ("lens_right", ctypes.c_float * 3),
                ("dof", ctypes.c_float), ("gather_bounces", ctypes.c_int32)]


STAT_FIELDS = ("primary_rays", "primary_hits", "secondary_rays", "shadow_rays", "node_tests", "mesh_entries",
               "inner_visits", "leaf_visits", "leaf_elems", "tri_tests", "tri_accepts")


class RtuStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in STAT_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in STAT_FIELDS}


TOUCH_FIELDS = ["rays", "node_tests", "mesh_box_tests", "inner4", "inner8", "inner_ref", "tri_tests", "winners", "xform_levels", "record_bytes", "bound_tests", "inline_shadow_rays"]
KERNEL_SLOTS = 40


class RtuTouched(ctypes.Structure):
    """rtu_render.h: what one kernel launch of the fast variant touched (collect_stats == 2)."""
    _fields_ = [(n, ctypes.c_uint64) for n in TOUCH_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in TOUCH_FIELDS}


def algorithmic_bytes(stats, pixels):
    """cache-agnostic bytes the path touches for one frame."""
    s = stats if isinstance(stats, dict) else stats.as_dict()
    return (56 * s["inner_visits"] + 4 * s["leaf_elems"] + 48 * s["tri_tests"] + 96 * s["tri_accepts"]
            + 72 * s["node_tests"] + 16 * pixels)


def total_rays(stats):
    s = stats if isinstance(stats, dict) else stats.as_dict()
    return s["primary_rays"] + s["secondary_rays"] + s["shadow_rays"]


def _load(name):
    path = os.path.join(_LIB, name)
    if not os.path.exists(path):
        raise ImportError("%s is missing: build the HIP extension first (__graft_entry__.build() or "
                          "`make -C raytracer-utah_amd`); there is no CPU fallback" % path)
    return ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


hip = _load("librtu_hip.so")
host = _load("librtu_host.so")

_P = ctypes.c_void_p
_I = ctypes.c_int


def _sig(lib, name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


# ---- rtu_render.h ----------------------------------------------------------
HIP_SYMBOLS = ["rtu_device_count", "rtu_error_string", "rtu_create_context", "rtu_destroy_context", "rtu_last_error",
               "rtu_upload_scene", "rtu_validate_scene", "rtu_frame_setup", "rtu_shard_rows", "rtu_shard_max_rows", "rtu_shard_global_row",
               "rtu_render_frame_device", "rtu_render_frames_device", "rtu_pack_image_device", "rtu_minmax_z_device", "rtu_pack_output_device", "rtu_render_frame", "rtu_frame_status", "rtu_render_timeline", "rtu_frame_counts", "rtu_timeline_exits", "rtu_mesh_info", "rtu_light_list_info", "rtu_debug_light_list", "rtu_debug_light_list_free", "rtu_debug_walk_stack_limit", "rtu_debug_node_bounds", "rtu_debug_flags", "rtu_set_sequences_in_flight", "rtu_debug_tail_from", "rtu_get_stats", "rtu_get_touched", "rtu_get_touched_launches", "rtu_touched_bytes", "rtu_kernel_slot_name", "rtu_probe_kernel", "rtu_probe_read", "rtu_time_render", "rtu_selftest_division", "rtu_selftest_primitives", "rtu_context_stream", "rtu_context_device", "rtu_context_sync", "rtu_host_alloc_pinned", "rtu_host_free_pinned", "rtu_copy_to_host_async", "rtu_device_alloc",
               "rtu_device_free", "rtu_copy_to_host", "rtu_device_info", "rtu_set_cancel_flag", "rtu_create_context_multi", "rtu_destroy_context_multi",
               "rtu_multi_size", "rtu_multi_context", "rtu_multi_last_error", "rtu_multi_upload_scene", "rtu_multi_render_frame", "rtu_multi_gather_kind"]
_sig(hip, "rtu_device_count", _I)
_sig(hip, "rtu_error_string", ctypes.c_char_p, _I)
_sig(hip, "rtu_create_context", _P, _I, ctypes.POINTER(_I))
_sig(hip, "rtu_destroy_context", None, _P)
_sig(hip, "rtu_last_error", ctypes.c_char_p, _P)
_sig(hip, "rtu_upload_scene", _I, _P, _P)
_sig(hip, "rtu_validate_scene", _I, _P, ctypes.c_char_p, ctypes.c_size_t)
_sig(hip, "rtu_frame_setup", _I, ctypes.POINTER(RtuCamera), _I, _I, ctypes.POINTER(RtuFrameDesc))
_sig(hip, "rtu_shard_rows", _I, ctypes.POINTER(RtuFrameDesc))
_sig(hip, "rtu_shard_max_rows", _I, _I, _I)
_sig(hip, "rtu_shard_global_row", _I, ctypes.POINTER(RtuFrameDesc), _I)
_sig(hip, "rtu_render_frame_device", _I, _P, ctypes.POINTER(RtuFrameDesc), _P, _P)
_sig(hip, "rtu_render_frames_device", _I, _P, ctypes.POINTER(RtuFrameDesc), _I, _P, _P)
_sig(hip, "rtu_pack_image_device", _I, _P, _P, ctypes.c_size_t, _P, _P, _P)
_sig(hip, "rtu_minmax_z_device", _I, _P, _P, ctypes.c_size_t, _I, _P, _P)
_sig(hip, "rtu_pack_output_device", _I, _P, _P, ctypes.c_size_t, _I, _P, _P, _P)
_sig(hip, "rtu_render_frame", _I, _P, ctypes.POINTER(RtuFrameDesc), _P, ctypes.POINTER(RtuStats))
_sig(hip, "rtu_frame_status", _I, _P)
_sig(hip, "rtu_get_touched", _I, _P, ctypes.POINTER(RtuTouched), _I)
_sig(hip, "rtu_touched_bytes", ctypes.c_uint64, ctypes.POINTER(RtuTouched), _I)
_sig(hip, "rtu_get_touched_launches", _I, _P, ctypes.POINTER(ctypes.c_uint32), _I)
_sig(hip, "rtu_kernel_slot_name", ctypes.c_char_p, _I)
_sig(hip, "rtu_probe_kernel", _I, _P, _I)
_sig(hip, "rtu_probe_read", _I, _P, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(_I))
_sig(hip, "rtu_debug_walk_stack_limit", _I, _P, ctypes.c_uint32)
_sig(hip, "rtu_debug_node_bounds", _I, _P, _I)
_sig(hip, "rtu_debug_flags", _I, _P, ctypes.c_uint32)
_sig(hip, "rtu_set_sequences_in_flight", _I, _P, _I)
_sig(hip, "rtu_debug_tail_from", _I, _P, _I)
_sig(hip, "rtu_timeline_exits", _I, _P, _I, _I, ctypes.POINTER(ctypes.c_double))
_sig(hip, "rtu_mesh_info", _I, _P, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32))
_sig(hip, "rtu_light_list_info", _I, _P, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32))
_sig(hip, "rtu_frame_counts", _I, _P, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32))
_sig(hip, "rtu_render_timeline", _I, _P, ctypes.POINTER(RtuFrameDesc), _P, _I, ctypes.POINTER(_I), ctypes.POINTER(ctypes.c_double),
     ctypes.POINTER(ctypes.c_double))
_sig(hip, "rtu_get_stats", _I, _P, ctypes.POINTER(RtuStats))
_sig(hip, "rtu_time_render", _I, _P, ctypes.POINTER(RtuFrameDesc), _P, _P, _I, ctypes.POINTER(ctypes.c_float))
_sig(hip, "rtu_selftest_primitives", _I, _P, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.POINTER(ctypes.c_ulonglong))
_sig(hip, "rtu_selftest_division", _I, _P, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.POINTER(ctypes.c_ulonglong))
_sig(hip, "rtu_device_alloc", _P, _P, ctypes.c_size_t)
_sig(hip, "rtu_device_free", None, _P, _P)
_sig(hip, "rtu_copy_to_host", _I, _P, _P, _P, ctypes.c_size_t)


class RtuDeviceInfo(ctypes.Structure):
    _fields_ = [("compute_units", ctypes.c_int32), ("clock_khz", ctypes.c_int32), ("memory_clock_khz", ctypes.c_int32), ("memory_bus_bits", ctypes.c_int32),
                ("l2_bytes", ctypes.c_uint64), ("hbm_bytes", ctypes.c_uint64), ("name", ctypes.c_char * 64), ("arch", ctypes.c_char * 64)]


ROWS_DONE = ctypes.CFUNCTYPE(None, _P, ctypes.POINTER(ctypes.c_float), _I, _I)


class RtuProgress(ctypes.Structure):
    _fields_ = [("cancel", ctypes.POINTER(_I)), ("rows_done", ROWS_DONE), ("user", _P)]


_sig(hip, "rtu_device_info", _I, _I, ctypes.POINTER(RtuDeviceInfo))
_sig(hip, "rtu_set_cancel_flag", _I, _P, ctypes.POINTER(_I))
_sig(hip, "rtu_create_context_multi", _P, ctypes.POINTER(_I), _I, ctypes.POINTER(_I))
_sig(hip, "rtu_destroy_context_multi", None, _P)
_sig(hip, "rtu_multi_size", _I, _P)
_sig(hip, "rtu_multi_context", _P, _P, _I)
_sig(hip, "rtu_multi_last_error", ctypes.c_char_p, _P)
_sig(hip, "rtu_multi_upload_scene", _I, _P, _P)
_sig(hip, "rtu_multi_render_frame", _I, _P, ctypes.POINTER(RtuFrameDesc), _P, ctypes.POINTER(RtuProgress))
_sig(hip, "rtu_multi_gather_kind", _I, _P)


class RtuLightListDump(ctypes.Structure):
    _fields_ = [("usable", ctypes.c_int32), ("node", ctypes.c_int32), ("light", ctypes.c_int32), ("G", ctypes.c_uint32), ("point", ctypes.c_uint32),
                ("n_entries", ctypes.c_uint32), ("X", ctypes.c_float * 3), ("Y", ctypes.c_float * 3), ("Z", ctypes.c_float * 3), ("L", ctypes.c_float * 3),
                ("u0", ctypes.c_float), ("v0", ctypes.c_float), ("su", ctypes.c_float), ("sv", ctypes.c_float),
                ("cell_off", ctypes.POINTER(ctypes.c_uint32)), ("entry_face", ctypes.POINTER(ctypes.c_uint32)), ("entry_zmin", ctypes.POINTER(ctypes.c_float))]


_sig(hip, "rtu_debug_light_list", _I, _P, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(RtuLightListDump))
_sig(hip, "rtu_debug_light_list_free", None, ctypes.POINTER(RtuLightListDump))


def light_list(scene, light_slot, cover_slot):
    """Pure host code: the occluder list of (non-ambient light, mesh node) as numpy arrays, or None when no list is usable from there:
    dict(G, point, X, Y, Z, L, u0, v0, su, sv, node, light, cell_off [G*G+1], entry_face, entry_zmin)."""
    import numpy as np
    d = RtuLightListDump()
    rc = hip.rtu_debug_light_list(scene.desc_ptr, light_slot, cover_slot, ctypes.byref(d))
    if rc != RTU_OK:
        raise RtuError(rc, "rtu_debug_light_list")
    try:
        if not d.usable:
            return None
        n = d.G * d.G + 1
        return {"G": d.G, "point": bool(d.point), "X": np.array(list(d.X)), "Y": np.array(list(d.Y)), "Z": np.array(list(d.Z)), "L": np.array(list(d.L)),
                "u0": d.u0, "v0": d.v0, "su": d.su, "sv": d.sv, "node": d.node, "light": d.light,
                "cell_off": np.ctypeslib.as_array(d.cell_off, (n,)).copy(), "entry_face": np.ctypeslib.as_array(d.entry_face, (max(d.n_entries, 1),))[:d.n_entries].copy(),
                "entry_zmin": np.ctypeslib.as_array(d.entry_zmin, (max(d.n_entries, 1),))[:d.n_entries].copy()}
    finally:
        hip.rtu_debug_light_list_free(ctypes.byref(d))


def device_info(device_id=0):
    """What hipGetDeviceProperties says about a GPU (dict), e.g. for the HBM peak of the roofline."""
    o = RtuDeviceInfo()
    rc = hip.rtu_device_info(device_id, ctypes.byref(o))
    if rc != RTU_OK:
        raise RtuError(rc, "rtu_device_info")
    return {"compute_units": o.compute_units, "clock_khz": o.clock_khz, "memory_clock_khz": o.memory_clock_khz, "memory_bus_bits": o.memory_bus_bits,
            "l2_bytes": o.l2_bytes, "hbm_bytes": o.hbm_bytes, "name": o.name.decode(), "arch": o.arch.decode()}

# ---- rtu_host.h --------------------------------------------------------------
HOST_SYMBOLS = ["rtu_scene_load_xml", "rtu_scene_clone", "rtu_scene_load_blob", "rtu_scene_load_blob_file",
                "rtu_scene_to_blob", "rtu_scene_save_blob_file", "rtu_blob_free", "rtu_scene_desc",
                "rtu_scene_set_resolution", "rtu_scene_free", "rtu_host_last_error", "rtu_image_create",
                "rtu_image_free", "rtu_image_width", "rtu_image_height", "rtu_image_pixels", "rtu_image_zbuffer",
                "rtu_image_zimage", "rtu_image_num_rendered", "rtu_image_is_done", "rtu_image_from_rgbz",
                "rtu_image_compute_zimg", "rtu_image_save_png", "rtu_image_save_zpng", "rtu_write_png",
                "rtu_begin_render", "rtu_begin_render_sampled", "rtu_begin_render_paths", "rtu_stop_render", "rtu_render_wait", "rtu_render_gather_kind", "rtu_render_job_free"]
_sig(host, "rtu_scene_load_xml", _P, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p)
_sig(host, "rtu_scene_clone", _P, _P)
_sig(host, "rtu_scene_load_blob", _P, _P, ctypes.c_size_t)
_sig(host, "rtu_scene_load_blob_file", _P, ctypes.c_char_p)
_sig(host, "rtu_scene_to_blob", _P, _P, ctypes.POINTER(ctypes.c_size_t))
_sig(host, "rtu_scene_save_blob_file", _I, _P, ctypes.c_char_p)
_sig(host, "rtu_blob_free", None, _P)
_sig(host, "rtu_scene_desc", ctypes.POINTER(RtuSceneDesc), _P)
_sig(host, "rtu_scene_set_resolution", None, _P, _I, _I)
_sig(host, "rtu_scene_free", None, _P)
_sig(host, "rtu_host_last_error", ctypes.c_char_p)
_sig(host, "rtu_image_create", _P, _I, _I)
_sig(host, "rtu_image_free", None, _P)
_sig(host, "rtu_image_width", _I, _P)
_sig(host, "rtu_image_height", _I, _P)
_sig(host, "rtu_image_pixels", _P, _P)
_sig(host, "rtu_image_zbuffer", _P, _P)
_sig(host, "rtu_image_zimage", _P, _P)
_sig(host, "rtu_image_num_rendered", _I, _P)
_sig(host, "rtu_image_is_done", _I, _P)
_sig(host, "rtu_image_from_rgbz", None, _P, _P, _I, _I)
_sig(host, "rtu_image_compute_zimg", None, _P)
_sig(host, "rtu_image_save_png", _I, _P, ctypes.c_char_p)
_sig(host, "rtu_image_save_zpng", _I, _P, ctypes.c_char_p)
_sig(host, "rtu_write_png", _I, ctypes.c_char_p, _P, _I, _I, _I)
_sig(host, "rtu_begin_render", _P, _P, _P, ctypes.POINTER(_I), _I, ctypes.c_char_p, ctypes.c_char_p)
_sig(host, "rtu_begin_render_sampled", _P, _P, _P, ctypes.POINTER(_I), _I, _I, ctypes.c_char_p, ctypes.c_char_p)
_sig(host, "rtu_begin_render_paths", _P, _P, _P, ctypes.POINTER(_I), _I, _I, ctypes.c_char_p, ctypes.c_char_p)
_sig(host, "rtu_stop_render", None, _P)
_sig(host, "rtu_render_wait", _I, _P)
_sig(host, "rtu_render_gather_kind", _I, _P)
_sig(host, "rtu_render_job_free", None, _P)


class Scene:
    """Owned flattened scene (RtuScene*)."""

    def __init__(self, handle):
        if not handle:
            raise RtuError(RTU_ERR_ARG, host.rtu_host_last_error().decode())
        self._h = handle

    @classmethod
    def from_xml(cls, path, remap_from=None, remap_to=None):
        enc = lambda s: s.encode() if s is not None else None
        return cls(host.rtu_scene_load_xml(path.encode(), enc(remap_from), enc(remap_to)))

    @classmethod
    def from_blob_bytes(cls, data):
        buf = ctypes.create_string_buffer(data, len(data))
        return cls(host.rtu_scene_load_blob(ctypes.cast(buf, _P), len(data)))

    @classmethod
    def from_blob_file(cls, path):
        if path.endswith(".gz"):
            import gzip
            with gzip.open(path, "rb") as f:
                return cls.from_blob_bytes(f.read())
        return cls(host.rtu_scene_load_blob_file(path.encode()))

    def to_blob_bytes(self):
        n = ctypes.c_size_t(0)
        p = host.rtu_scene_to_blob(self.desc_ptr, ctypes.byref(n))
        if not p:
            raise RtuError(RTU_ERR_ARG, "serialisation failed")
        try:
            return ctypes.string_at(p, n.value)
        finally:
            host.rtu_blob_free(p)

    @property
    def desc(self):
        return host.rtu_scene_desc(self._h).contents

    @property
    def desc_ptr(self):
        return ctypes.cast(host.rtu_scene_desc(self._h), _P)

    def set_resolution(self, w, h):
        host.rtu_scene_set_resolution(self._h, w, h)

    def close(self):
        if self._h:
            host.rtu_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def frame_setup(camera, width, height, shard_rank=0, shard_count=1, collect_stats=False, max_bounce=5, samples=0, gather_bounces=0):
    f = RtuFrameDesc()
    rc = hip.rtu_frame_setup(ctypes.byref(camera), width, height, ctypes.byref(f))
    if rc != RTU_OK:
        raise RtuError(rc, "rtu_frame_setup")
    f.shard_rank, f.shard_count = shard_rank, shard_count
    f.collect_stats = int(collect_stats)  # False / True, or 2: touched-bytes mode of the fast variant
    f.max_bounce = max_bounce
    f.samples = samples  # 0: recipe W; S >= 1: recipe S (soft shadows, glossy bounces, depth of field)
    f.gather_bounces = gather_bounces  # 4 (with samples): recipe P, + the Monte-Carlo gather of config 5
    return f


class MultiContext:
    """Several GPUs behind one handle (RtuMultiContext*): the frame sharded by interleaved 8-row bands, gathered and de-interleaved
    under the C-ABI. device_ids may repeat (several contexts on one GPU)."""

    def __init__(self, device_ids):
        err = _I(0)
        arr = (_I * len(device_ids))(*device_ids)
        self._h = hip.rtu_create_context_multi(arr, len(device_ids), ctypes.byref(err))
        if not self._h:
            raise RtuError(err.value, hip.rtu_error_string(err.value).decode())
        self.n = len(device_ids)

    def _check(self, rc):
        if rc != RTU_OK:
            raise RtuError(rc, hip.rtu_multi_last_error(self._h).decode())

    def upload(self, scene):
        self._check(hip.rtu_multi_upload_scene(self._h, scene.desc_ptr))

    def context_handle(self, i):
        return hip.rtu_multi_context(self._h, i)

    def render(self, frame, on_rows=None, cancel=None):
        """The whole frame [H, W, 4] float32. on_rows(row0, nrows): called per band as the shards arrive; cancel: a ctypes.c_int the
        caller may set non-zero (the call then raises RtuError(RTU_ERR_CANCELLED))."""
        import numpy as np
        out = np.empty((frame.height, frame.width, 4), np.float32)
        prog = RtuProgress()
        cb = ROWS_DONE(lambda user, rows, row0, nrows: on_rows(row0, nrows)) if on_rows else ROWS_DONE()
        prog.rows_done = cb
        prog.cancel = ctypes.pointer(cancel) if cancel is not None else None
        self._check(hip.rtu_multi_render_frame(self._h, ctypes.byref(frame), out.ctypes.data, ctypes.byref(prog)))
        return out

    def gather_kind(self):
        return hip.rtu_multi_gather_kind(self._h)

    def close(self):
        if self._h:
            hip.rtu_destroy_context_multi(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One GPU (RtuContext*)."""

    def __init__(self, device_id=0):
        err = _I(0)
        self._h = hip.rtu_create_context(device_id, ctypes.byref(err))
        if not self._h:
            raise RtuError(err.value, hip.rtu_error_string(err.value).decode())

    def _check(self, rc):
        if rc != RTU_OK:
            raise RtuError(rc, hip.rtu_last_error(self._h).decode())

    def upload(self, scene):
        self._check(hip.rtu_upload_scene(self._h, scene.desc_ptr))

    def render(self, frame, stats=False):
        """Render this shard; returns (rgbz float32 [rows, W, 4], stats dict or None)."""
        import numpy as np
        rows = hip.rtu_shard_rows(ctypes.byref(frame))
        out = np.empty((rows, frame.width, 4), np.float32)
        st = RtuStats() if stats else None
        self._check(hip.rtu_render_frame(self._h, ctypes.byref(frame), out.ctypes.data,
                                         ctypes.byref(st) if stats else None))
        return out, (st.as_dict() if stats else None)

    def render_device(self, frame, d_ptr, stream=None):
        self._check(hip.rtu_render_frame_device(self._h, ctypes.byref(frame), d_ptr, stream))

    def render_frames_device(self, frames, d_ptr, stream=None):
        """Frames in flight: len(frames) frames of recipe W in one launch sequence, images consecutive at d_ptr."""
        arr = (RtuFrameDesc * len(frames))(*frames)
        self._check(hip.rtu_render_frames_device(self._h, arr, len(frames), d_ptr, stream))

    def pack_image_device(self, d_rgbz, n_pixels, d_z, d_rgb8, stream=None):
        """float4 image -> float z + Color24 pixels (the reference's RenderImage content), on the device."""
        self._check(hip.rtu_pack_image_device(self._h, d_rgbz, n_pixels, d_z, d_rgb8, stream))

    def minmax_z_device(self, d_rgbz, pixels_per_frame, n_frames, d_minmax, stream=None):
        """Per frame of a batch: keys of this shard's zmin / zmax as int64 pairs (all-reduce MIN over the shards gives the frame's)."""
        self._check(hip.rtu_minmax_z_device(self._h, d_rgbz, pixels_per_frame, n_frames, d_minmax, stream))

    def pack_output_device(self, d_rgbz, pixels_per_frame, n_frames, d_minmax, d_out4, stream=None):
        """float4 images -> 4 bytes per pixel {Color24, z-image byte}: the content of Result.png and ZBuffer.png."""
        self._check(hip.rtu_pack_output_device(self._h, d_rgbz, pixels_per_frame, n_frames, d_minmax, d_out4, stream))

    def frame_status(self):
        """Synchronise; raises RtuError(RTU_ERR_CAPACITY) if the frame must be rendered again."""
        self._check(hip.rtu_frame_status(self._h))

    TIMELINE_SLOTS = (["k_primary", "k_primary2c", "k_primary2"] +
                      ["%s(L%d)" % (k, L) for L in range(6) for k in ("k_trace", "k_trace2c", "k_trace2", "k_consume")] +
                      ["k_combine(L%d)" % L for L in range(6)])
    # when the tail kernel (k_tail) takes over from level Ls, its stamps are in the k_trace(L<Ls>) slot

    def render_timeline(self, frame, d_ptr):
        """One frame with in-kernel GPU-clock stamps: [(kernel, start_us, end_us)] in launch order."""
        n = 40
        slot, t0, t1 = (_I * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)()
        rc = hip.rtu_render_timeline(self._h, ctypes.byref(frame), d_ptr, n, slot, t0, t1)
        if rc < 0:
            self._check(rc)
        order = {name: i for i, name in enumerate(self.TIMELINE_SLOTS)}
        rows = [(self.TIMELINE_SLOTS[slot[i]], t0[i], t1[i]) for i in range(rc)]
        # launch order: combines run bottom-up after everything else
        return sorted(rows, key=lambda r: (r[0].startswith("k_combine"), -order[r[0]] if r[0].startswith("k_combine") else order[r[0]]))

    def timeline_exits(self, kernel):
        """Exit times (us after the kernel's first entry) of the wavefronts of `kernel` (a name of
        TIMELINE_SLOTS) in the frame last rendered by render_timeline."""
        import numpy as np
        buf = (ctypes.c_double * 8192)()
        rc = hip.rtu_timeline_exits(self._h, self.TIMELINE_SLOTS.index(kernel), 8192, buf)
        if rc < 0:
            self._check(rc)
        return np.array(buf[:rc])

    def mesh_info(self, mesh=0):
        """dict(faces, sah_depth, stack4, nodes4, nodes8) of an uploaded mesh."""
        o = (ctypes.c_uint32 * 5)()
        self._check(hip.rtu_mesh_info(self._h, mesh, o))
        return dict(zip(("faces", "sah_depth", "stack4", "nodes4", "nodes8"), list(o)))

    def light_lists(self):
        """[dict(light, cover, G, entries, longest)]: the occluder lists of shadow rays built at upload."""
        out, i = [], 0
        while True:
            o = (ctypes.c_uint32 * 5)()
            if hip.rtu_light_list_info(self._h, i, o) != 0:
                return out
            out.append(dict(zip(("light", "cover", "G", "entries", "longest"), list(o))))
            i += 1

    def frame_counts(self):
        """(frames per level [6], rays deferred to stage 2 per phase [7]) of the most recent frame."""
        fr, de = (ctypes.c_uint32 * 6)(), (ctypes.c_uint32 * 7)()
        self._check(hip.rtu_frame_counts(self._h, fr, de))
        return list(fr), list(de)

    def time_render(self, frame, d_ptr, stream, iters):
        ms = ctypes.c_float(0)
        self._check(hip.rtu_time_render(self._h, ctypes.byref(frame), d_ptr, stream, iters, ctypes.byref(ms)))
        return ms.value

    def touched(self, textured=False):
        """Touched-bytes mode (frame.collect_stats == 2): {kernel slot name: counters + 'bytes'} of the launches of the most
        recent launch sequence that touched anything."""
        arr = (RtuTouched * KERNEL_SLOTS)()
        n = hip.rtu_get_touched(self._h, arr, KERNEL_SLOTS)
        if n < 0:
            self._check(n)
        nl = (ctypes.c_uint32 * KERNEL_SLOTS)()
        hip.rtu_get_touched_launches(self._h, nl, KERNEL_SLOTS)
        out = {}
        for k in range(n):
            d = arr[k].as_dict()
            if any(d.values()):
                d["bytes"] = int(hip.rtu_touched_bytes(ctypes.byref(arr[k]), 1 if textured else 0))
                d["launches"] = int(nl[k])  # of this slot's kernel since the counters were zeroed (a sampled frame is many launch sequences)
                out[hip.rtu_kernel_slot_name(k).decode()] = d
        return out

    def probe_kernel(self, slot_name):
        """Bracket every launch of that kernel slot ('k_trace2(L0)', ...; None: stop) with HIP events on its stream."""
        slot = -1
        if slot_name is not None:
            names = [hip.rtu_kernel_slot_name(k).decode() for k in range(KERNEL_SLOTS)]
            slot = names.index(slot_name)
        self._check(hip.rtu_probe_kernel(self._h, slot))

    def probe_read(self):
        """(summed milliseconds, launches measured) since the last read; synchronises."""
        ms, n = ctypes.c_float(0), _I(0)
        self._check(hip.rtu_probe_read(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def stats(self):
        st = RtuStats()
        self._check(hip.rtu_get_stats(self._h, ctypes.byref(st)))
        return st.as_dict()

    def close(self):
        if self._h:
            hip.rtu_destroy_context(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_rows(frame):
    return hip.rtu_shard_rows(ctypes.byref(frame))


def shard_global_rows(frame):
    """Global row index of every local row of this shard."""
    import numpy as np
    n = shard_rows(frame)
    return np.array([hip.rtu_shard_global_row(ctypes.byref(frame), i) for i in range(n)], dtype=np.int64)


def assemble(shards, frames, height):
    """De-interleave per-shard compact buffers into one [H, W, 4] image."""
    import numpy as np
    w = frames[0].width
    out = np.empty((height, w, 4), np.float32)
    for buf, fr in zip(shards, frames):
        rows = shard_global_rows(fr)
        out[rows] = buf[:len(rows)]
    return out


class Image:
    """RenderImage mirror (RtuImage*): Color24 pixels, float z-buffer, z-image."""

    def __init__(self, width, height):
        self._h = host.rtu_image_create(width, height)
        self.width, self.height = width, height

    def fill(self, rgbz, row0=0):
        import numpy as np
        a = np.ascontiguousarray(rgbz, dtype=np.float32)
        host.rtu_image_from_rgbz(self._h, a.ctypes.data, row0, a.shape[0])

    def compute_zimage(self):
        host.rtu_image_compute_zimg(self._h)

    def pixels(self):
        import numpy as np
        p = host.rtu_image_pixels(self._h)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)),
                                     (self.height, self.width, 3)).copy()

    def zbuffer(self):
        import numpy as np
        p = host.rtu_image_zbuffer(self._h)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_float)), (self.height, self.width)).copy()

    def zimage(self):
        import numpy as np
        p = host.rtu_image_zimage(self._h)
        if not p:
            return None
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), (self.height, self.width)).copy()

    def save(self, result_png=None, zbuffer_png=None):
        if result_png and host.rtu_image_save_png(self._h, result_png.encode()) != 0:
            raise RtuError(RTU_ERR_ARG, "cannot write " + result_png)
        if zbuffer_png and host.rtu_image_save_zpng(self._h, zbuffer_png.encode()) != 0:
            raise RtuError(RTU_ERR_ARG, "cannot write " + zbuffer_png)

    def close(self):
        if self._h:
            host.rtu_image_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
