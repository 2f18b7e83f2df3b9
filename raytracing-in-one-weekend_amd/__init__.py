"""ctypes binding of librt_mi355x.so, the MI355X path-tracing library.

The names mirror the reference's procedures (ray-tracer/main.odin): ``render_region``
(:510), ``render_tiled`` (:635), ``calculate_camera_render_data`` (:450), ``lookat`` (:65),
``build_dev_scene`` (:687), ``build_final_scene`` (:739), ``serialize_ppm`` (:791).
Everything here goes through the C ABI of ``include/rt_mi355x.h``; there is no Python or
CPU implementation of the render path behind it, and loading fails loudly when the shared
library is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_MI355X_LIB") or os.path.join(_HERE, "librt_mi355x.so")  # env override: A/B builds of the library

# ---- constants of include/rt_mi355x.h ------------------------------------------------
RT_OK = 0
RT_ERR_INVALID_ARGUMENT = -1
RT_ERR_INVALID_SCENE = -2
RT_ERR_DEVICE = -3
RT_ERR_UNSUPPORTED = -4
RT_ERR_BUFFER_TOO_SMALL = -5
LAMBERTIAN, METALLIC, DIELECTRIC = 0, 1, 2
PRECISION_F32, PRECISION_F64, PRECISION_F32_LITERAL, PRECISION_MIXED, PRECISION_MIXED_FIXED = 0, 1, 2, 3, 4
FLAG_PRINT_PROGRESS = 1
#: bytes per accumulator element (three elements per pixel) of rt_render_accumulate, by precision
ACCUM_ELEMENT_BYTES = {PRECISION_F32: 8, PRECISION_F64: 8, PRECISION_F32_LITERAL: 4, PRECISION_MIXED: 8, PRECISION_MIXED_FIXED: 16}
FETCH_AUTO, FETCH_SCALAR, FETCH_LDS, FETCH_BVH, FETCH_BVH_GLOBAL = 0, 1, 2, 3, 4
RNG_STOCHASTIC, RNG_DETERMINISTIC = 0, 1
STAT_SEGMENTS, STAT_SPHERE_TESTS, STAT_KERNEL_NS, STAT_SAMPLES, STAT_WAVE_ITERATIONS, STAT_BVH_NODES, STAT_BVH_TESTS = 0, 1, 2, 3, 4, 5, 6
STATS_COUNT = 8

# numpy views of the 40-byte Odin structs (main.odin:487-491, :285-289)
SPHERE_DTYPE = np.dtype([("center", "<f8", (3,)), ("radius", "<f8"), ("material_index", "<i8")])
MATERIAL_DTYPE = np.dtype([("type", "<i8"), ("albedo", "<f8", (3,)), ("param1", "<f8")])
assert SPHERE_DTYPE.itemsize == 40 and MATERIAL_DTYPE.itemsize == 40


class V3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]

    def tuple(self):
        return (self.x, self.y, self.z)


def _v3(v) -> V3:
    return V3(float(v[0]), float(v[1]), float(v[2]))


class Rect(C.Structure):
    """rect (main.odin:503-508): half-open pixel rectangle."""
    _fields_ = [("min_x", C.c_int64), ("min_y", C.c_int64), ("one_past_max_x", C.c_int64), ("one_past_max_y", C.c_int64)]


class CameraRenderData(C.Structure):
    """camera_render_data (main.odin:440-448)."""
    _fields_ = [("position", V3), ("pixel_delta_u", V3), ("pixel_delta_v", V3), ("view_plane_top_left_pixel_center", V3),
                ("dof_disk_u", V3), ("dof_disk_v", V3), ("depth_of_field_angle", C.c_double),
                ("samples_per_pixel", C.c_int64), ("max_ray_bounces", C.c_int64)]


class CameraSettings(C.Structure):
    """camera_settings (main.odin:425-438)."""
    _fields_ = [("position", V3), ("right", V3), ("up", V3), ("forward", V3), ("aspect_ratio", C.c_double),
                ("image_size_x", C.c_double), ("image_size_y", C.c_double), ("focus_distance", C.c_double),
                ("vfov", C.c_double), ("depth_of_field_angle", C.c_double),
                ("samples_per_pixel", C.c_int64), ("max_ray_bounces", C.c_int64)]

    @property
    def image_width(self) -> int:  # cast(i64)camera_settings.image_size.x (main.odin:833)
        return int(self.image_size_x)

    @property
    def image_height(self) -> int:  # cast(i64)camera_settings.image_size.y (main.odin:834)
        return int(self.image_size_y)


class Options(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("rng_mode", C.c_int32), ("precision", C.c_int32), ("sphere_fetch", C.c_int32),
                ("device", C.c_int32), ("device_count", C.c_int32), ("sample_chunks", C.c_int32),
                ("flags", C.c_int32), ("reserved", C.c_int32),
                ("linear_out", C.POINTER(V3)), ("stats", C.POINTER(C.c_uint64))]


assert C.sizeof(CameraRenderData) == 168 and C.sizeof(CameraSettings) == 160 and C.sizeof(Rect) == 32


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rt_mi355x error {code}: {message}")
        self.code = code
        self.message = message


_lib = None


def lib() -> C.CDLL:
    """Load librt_mi355x.so (built in-tree by ``make -C raytracing-in-one-weekend_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no fallback implementation")
    # PyTorch-ROCm wheels carry their own libamdhip64.so.7.  A process must run ONE HIP runtime:
    # with torch imported first, this library's DT_NEEDED libamdhip64.so.7 binds to the copy
    # torch already mapped (same SONAME); loaded the other way round, torch would map a second
    # runtime next to the system one and find no devices.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint32
    P = C.POINTER
    L.rt_version.restype = C.c_char_p
    L.rt_last_error.restype = C.c_char_p
    L.rt_device_count.restype = i32
    L.rt_options_default.argtypes = [P(Options)]
    L.rt_options_default.restype = None
    scene = [vp, i64, vp, i64]
    L.rt_render_region.argtypes = [vp, i64, i64, Rect, P(CameraRenderData)] + scene + [P(Options)]
    L.rt_render_region.restype = i32
    L.rt_render_tiled.argtypes = [vp, i64, i64, P(CameraRenderData)] + scene + [P(Options)]
    L.rt_render_tiled.restype = i32
    L.rt_world_create.argtypes = scene + [i32, P(vp)]
    L.rt_world_create.restype = i32
    L.rt_world_destroy.argtypes = [vp]
    L.rt_world_destroy.restype = i32
    L.rt_render_accumulate.argtypes = [vp, P(CameraRenderData), i64, i64, Rect, i64, i32, i32, i64, i64, u64, i32, i32, i32, vp, vp, vp]
    L.rt_render_accumulate.restype = i32
    L.rt_render_accumulate_static.argtypes = L.rt_render_accumulate.argtypes
    L.rt_render_accumulate_static.restype = i32
    L.rt_static_scratch_bytes.argtypes = []
    L.rt_static_scratch_bytes.restype = i64
    L.rt_finalize.argtypes = [vp, vp, i32, i64, i64, Rect, i64, i32, i32, i64, vp, vp, vp]
    L.rt_finalize.restype = i32
    L.rt_finalize_ppm.argtypes = [vp, vp, i32, i64, i64, i64, i32, i32, i64, vp, vp]
    L.rt_finalize_ppm.restype = i32
    L.rt_render_ppm.argtypes = [i64, i64, P(CameraRenderData), vp, i64, vp, i64, P(Options), vp, i64]
    L.rt_render_ppm.restype = i64
    L.rt_bvh_validate.argtypes = [vp, i64, P(i64), P(i64)]
    L.rt_bvh_validate.restype = i32
    L.rt_world_last_kernel_ns.argtypes = [vp]
    L.rt_world_last_kernel_ns.restype = i64
    L.rt_lookat.argtypes = [V3, V3, V3, P(V3), P(V3), P(V3)]
    L.rt_lookat.restype = None
    L.rt_calculate_camera_render_data.argtypes = [P(CameraSettings), P(CameraRenderData)]
    L.rt_calculate_camera_render_data.restype = None
    build = [P(CameraSettings), vp, P(i64), vp, P(i64)]
    L.rt_build_dev_scene.argtypes = [i64] + build
    L.rt_build_two_sphere_scene.argtypes = [i64] + build
    L.rt_build_final_scene.argtypes = [u64, i64] + build
    L.rt_build_stress_scene.argtypes = [u64, i64, i64] + build
    for f in (L.rt_build_dev_scene, L.rt_build_two_sphere_scene, L.rt_build_final_scene, L.rt_build_stress_scene):
        f.restype = i32
    L.rt_serialize_ppm.argtypes = [vp, i64, i64, vp, i64]
    L.rt_serialize_ppm.restype = i64
    L.rt_quantize_rgb8.argtypes = [vp, i64, vp]
    L.rt_quantize_rgb8.restype = i32
    L.rt_release_cache.argtypes = []
    L.rt_release_cache.restype = i32
    L.rt_rng_u32.argtypes = [u64, u32, u32, u32, u32]
    L.rt_rng_u32.restype = u32
    _lib = L
    return L


#: every symbol include/rt_mi355x.h declares
EXPORTED_SYMBOLS = (
    "rt_version", "rt_last_error", "rt_device_count", "rt_options_default", "rt_render_region", "rt_render_tiled",
    "rt_world_create", "rt_world_destroy", "rt_render_accumulate", "rt_finalize", "rt_world_last_kernel_ns", "rt_lookat",
    "rt_calculate_camera_render_data", "rt_build_dev_scene", "rt_build_two_sphere_scene", "rt_build_final_scene",
    "rt_build_stress_scene", "rt_serialize_ppm", "rt_quantize_rgb8", "rt_rng_u32", "rt_finalize_ppm", "rt_render_ppm", "rt_bvh_validate",
    "rt_render_accumulate_static", "rt_static_scratch_bytes", "rt_release_cache",
)


def _check(rc: int) -> None:
    if rc != RT_OK:
        raise RtError(rc, lib().rt_last_error().decode())


def _scene_ptrs(spheres: np.ndarray, materials: np.ndarray):
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    materials = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
    return spheres, materials, spheres.ctypes.data, len(spheres), materials.ctypes.data, len(materials)


# ---- host-side helpers ---------------------------------------------------------------
def lookat(position=(0, 0, 0), target=(0, 0, -1), axis_up=(0, 1, 0)):
    """lookat (main.odin:65-71) -> (right, up, forward)."""
    r, u, f = V3(), V3(), V3()
    lib().rt_lookat(_v3(position), _v3(target), _v3(axis_up), C.byref(r), C.byref(u), C.byref(f))
    return r.tuple(), u.tuple(), f.tuple()


def calculate_camera_render_data(settings: CameraSettings) -> CameraRenderData:
    """calculate_camera_render_data (main.odin:450-485)."""
    out = CameraRenderData()
    lib().rt_calculate_camera_render_data(C.byref(settings), C.byref(out))
    return out


def _build(fn, *head) -> Tuple[CameraSettings, np.ndarray, np.ndarray]:
    settings = CameraSettings()
    ns, nm = C.c_int64(0), C.c_int64(0)
    rc = fn(*head, C.byref(settings), None, C.byref(ns), None, C.byref(nm))
    if rc not in (RT_OK, RT_ERR_BUFFER_TOO_SMALL):
        raise RtError(rc, "scene builder rejected its arguments")
    spheres = np.zeros(ns.value, dtype=SPHERE_DTYPE)
    materials = np.zeros(nm.value, dtype=MATERIAL_DTYPE)
    rc = fn(*head, C.byref(settings), spheres.ctypes.data, C.byref(ns), materials.ctypes.data, C.byref(nm))
    if rc != RT_OK:
        raise RtError(rc, "scene builder failed")
    return settings, spheres, materials


def build_dev_scene(image_width: int = 200):
    """build_dev_scene (main.odin:687-737) -> (camera_settings, spheres, materials)."""
    return _build(lib().rt_build_dev_scene, image_width)


def build_two_sphere_scene(image_width: int = 400):
    return _build(lib().rt_build_two_sphere_scene, image_width)


def build_final_scene(seed: int = 1, image_width: int = 1200):
    """build_final_scene (main.odin:739-789) with splitmix64(seed) draws."""
    return _build(lib().rt_build_final_scene, seed, image_width)


def build_stress_scene(seed: int = 5, total_spheres: int = 10000, image_width: int = 1920):
    return _build(lib().rt_build_stress_scene, seed, total_spheres, image_width)


def serialize_ppm(image: np.ndarray) -> bytes:
    """serialize_ppm (main.odin:791-824); image is (height, width, 3) float64, gamma space."""
    image = np.ascontiguousarray(image, dtype=np.float64)
    h, w, _ = image.shape
    need = lib().rt_serialize_ppm(image.ctypes.data, w, h, None, 0)
    buf = C.create_string_buffer(int(need))
    got = lib().rt_serialize_ppm(image.ctypes.data, w, h, buf, need)
    if got != need:
        raise RtError(int(got), "rt_serialize_ppm failed")
    return buf.raw[:need]


def render_ppm(width: int, height: int, camera_render_data: CameraRenderData, spheres: np.ndarray, materials: np.ndarray, *,
               seed: int = 0, rng_mode: int = RNG_DETERMINISTIC, precision: int = PRECISION_MIXED, sphere_fetch: int = FETCH_AUTO,
               device: int = -1, device_count: int = 1, sample_chunks: int = 0) -> bytes:
    """render_tiled followed by serialize_ppm (main.odin:840-848) with the quantiser and the text
    formatting on the GPU; the bytes equal serialize_ppm(render_tiled(...))."""
    spheres, materials, sp, ns, mp, nm = _scene_ptrs(spheres, materials)
    o = Options()
    lib().rt_options_default(C.byref(o))
    o.seed, o.rng_mode, o.precision, o.sphere_fetch = seed, rng_mode, precision, sphere_fetch
    o.device, o.device_count, o.sample_chunks = device, device_count, sample_chunks
    need = lib().rt_render_ppm(width, height, C.byref(camera_render_data), sp, ns, mp, nm, C.byref(o), None, 0)
    if need < 0:
        raise RtError(int(need), lib().rt_last_error().decode())
    buf = C.create_string_buffer(int(need))
    got = lib().rt_render_ppm(width, height, C.byref(camera_render_data), sp, ns, mp, nm, C.byref(o), buf, need)
    if got != need:
        raise RtError(int(got), lib().rt_last_error().decode())
    return buf.raw[:need]


def bvh_validate(spheres: np.ndarray):
    """build the hierarchy of FETCH_BVH on the host and check its invariants; returns (n_nodes, max_leaf)"""
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    n_nodes, max_leaf = C.c_int64(0), C.c_int64(0)
    _check(lib().rt_bvh_validate(spheres.ctypes.data, len(spheres), C.byref(n_nodes), C.byref(max_leaf)))
    return int(n_nodes.value), int(max_leaf.value)


def quantize_rgb8(image: np.ndarray) -> np.ndarray:
    image = np.ascontiguousarray(image, dtype=np.float64)
    out = np.empty(image.shape, dtype=np.uint8)
    _check(lib().rt_quantize_rgb8(image.ctypes.data, image.size // 3, out.ctypes.data))
    return out


def release_cache() -> None:
    """free what the one-shot entry points keep per GPU between calls (world, buffers, stream)"""
    _check(lib().rt_release_cache())


def rng_u32(seed: int, pixel: int, sample: int, bounce: int, draw: int) -> int:
    return int(lib().rt_rng_u32(seed, pixel, sample, bounce, draw))


# ---- the render path -----------------------------------------------------------------
def render_region(image: np.ndarray, region, camera_render_data: CameraRenderData, spheres: np.ndarray, materials: np.ndarray, *,
                  seed: int = 0, rng_mode: int = RNG_DETERMINISTIC, precision: int = PRECISION_MIXED, sphere_fetch: int = FETCH_AUTO,
                  device: int = -1, device_count: int = 1, sample_chunks: int = 0, flags: int = 0,
                  linear_out: Optional[np.ndarray] = None, want_stats: bool = False):
    """render_region (main.odin:510-591) on the GPU.

    ``image`` is a C-contiguous (height, width, 3) float64 array (``image.data`` of the
    reference, row-major, index u + v*width); pixels of ``region`` = (min_x, min_y,
    one_past_max_x, one_past_max_y) are overwritten with gamma-space colours.
    Returns the stats array (numpy uint64[STATS_COUNT]) when ``want_stats``.
    """
    if image.dtype != np.float64 or image.ndim != 3 or image.shape[2] != 3 or not image.flags.c_contiguous:
        raise ValueError("image must be a C-contiguous (height, width, 3) float64 array")
    h, w, _ = image.shape
    spheres, materials, sp, ns, mp, nm = _scene_ptrs(spheres, materials)
    o = Options()
    lib().rt_options_default(C.byref(o))
    o.seed, o.rng_mode, o.precision, o.sphere_fetch = seed, rng_mode, precision, sphere_fetch
    o.device, o.device_count, o.sample_chunks, o.flags = device, device_count, sample_chunks, flags
    if linear_out is not None:
        if linear_out.dtype != np.float64 or linear_out.shape != image.shape or not linear_out.flags.c_contiguous:
            raise ValueError("linear_out must match image")
        o.linear_out = C.cast(linear_out.ctypes.data, C.POINTER(V3))
    stats = np.zeros(STATS_COUNT, dtype=np.uint64)
    if want_stats:
        o.stats = C.cast(stats.ctypes.data, C.POINTER(C.c_uint64))
    _check(lib().rt_render_region(image.ctypes.data, w, h, Rect(*[int(v) for v in region]), C.byref(camera_render_data), sp, ns, mp, nm, C.byref(o)))
    return stats if want_stats else None


def render_tiled(image: np.ndarray, camera_render_data: CameraRenderData, spheres: np.ndarray, materials: np.ndarray, **kw):
    """render_tiled (main.odin:635-685): the whole image."""
    h, w, _ = image.shape
    return render_region(image, (0, 0, w, h), camera_render_data, spheres, materials, **kw)


class World:
    """The reference's ``world`` (main.odin:493-496) resident on one GPU (rt_world)."""

    def __init__(self, spheres: np.ndarray, materials: np.ndarray, device: int = -1):
        spheres, materials, sp, ns, mp, nm = _scene_ptrs(spheres, materials)
        self.n_spheres, self.n_materials = ns, nm
        handle = C.c_void_p()
        _check(lib().rt_world_create(sp, ns, mp, nm, device, C.byref(handle)))
        self._h = handle

    def close(self):
        if self._h:
            lib().rt_world_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_accumulate(self, cam: CameraRenderData, width: int, height: int, d_accum: int, *, region=None, strip_rows: int = 8,
                          strip_index: int = 0, strip_count: int = 1, sample_begin: int = 0, sample_count: Optional[int] = None,
                          seed: int = 0, precision: int = PRECISION_MIXED, sphere_fetch: int = FETCH_AUTO, sample_chunks: int = 0,
                          d_stats: int = 0, stream: int = 0, static_scratch: bool = False) -> None:
        """rt_render_accumulate: d_accum / d_stats / stream are raw device pointers / a hipStream_t.
        static_scratch: rt_render_accumulate_static, the entry point that allocates nothing."""
        region = (0, 0, width, height) if region is None else region
        sample_count = cam.samples_per_pixel if sample_count is None else sample_count
        fn = lib().rt_render_accumulate_static if static_scratch else lib().rt_render_accumulate
        _check(fn(self._h, C.byref(cam), width, height, Rect(*[int(v) for v in region]), strip_rows, strip_index,
                                          strip_count, sample_begin, sample_count, seed, precision, sphere_fetch, sample_chunks,
                                          d_accum, d_stats or None, stream or None))

    def finalize(self, d_accum: int, width: int, height: int, total_samples: int, d_image: int, *, region=None, strip_rows: int = 8,
                 strip_index: int = 0, strip_count: int = 1, precision: int = PRECISION_MIXED, d_linear: int = 0, stream: int = 0) -> None:
        region = (0, 0, width, height) if region is None else region
        _check(lib().rt_finalize(self._h, d_accum, precision, width, height, Rect(*[int(v) for v in region]), strip_rows, strip_index,
                                 strip_count, total_samples, d_image, d_linear or None, stream or None))

    def finalize_ppm(self, d_accum: int, width: int, height: int, total_samples: int, d_text: int, *, strip_rows: int = 8,
                     strip_index: int = 0, strip_count: int = 1, precision: int = PRECISION_MIXED, stream: int = 0) -> None:
        """the tail of the pixel loop and the body of serialize_ppm (main.odin:577-586, :806-820) on the device:
        12 bytes of text per pixel at d_text + 12 * (u + v*width)"""
        _check(lib().rt_finalize_ppm(self._h, d_accum, precision, width, height, strip_rows, strip_index, strip_count, total_samples,
                                     d_text, stream or None))

    def last_kernel_ns(self) -> int:
        ns = lib().rt_world_last_kernel_ns(self._h)
        if ns < 0:
            raise RtError(int(ns), lib().rt_last_error().decode())
        return int(ns)
