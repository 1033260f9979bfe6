"""ctypes binding of libraytracer_amd.so (include/rt_capi.h) -- plumbing only; all work happens in the library.

The library is built in-tree by `make lib` (or __graft_entry__.build()).  There is no Python or CPU fallback: if the
shared object is missing, load() raises, and if no GPU is present every device entry point returns RT_EHIP.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_LIB_PATH") or os.path.join(_HERE, "lib", "libraytracer_amd.so")   # RT_LIB_PATH: diagnostic builds

RT_OK, RT_EINVAL, RT_EIO, RT_EPARSE, RT_EHIP, RT_ENOMEM, RT_ELIMIT = 0, -1, -2, -3, -4, -5, -6
RT_FLAG_COUNTERS, RT_FLAG_TIME_KERNELS, RT_FLAG_WHOLE_FRAME = 1, 2, 4
MAX_REFLECTIONS = 16
RT_STATIC_MAX_RAYS = 65536

# every symbol include/rt_capi.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "rt_load_scene_json", "rt_load_scene_json_text", "rt_flat_scene_save", "rt_flat_scene_load", "rt_flat_scene_free",
    "rt_camera_setup", "rt_write_bmp", "rt_quantize_rgb8", "rt_device_count", "rt_scene_upload", "rt_scene_free",
    "rt_render", "rt_render_submit", "rt_render_wait", "rt_render_host", "rt_trace_batch", "rt_trace_pick", "rt_pick_narration", "rt_intersect_batch", "rt_intersect_batch_static", "rt_shadow_batch", "rt_tiles_merge",
    "rt_tiles_count", "rt_tiles_pack", "rt_tiles_unpack", "rt_tiles_pack_rgb8", "rt_tiles_unpack_rgb8", "rt_tiles_quantize_rgb8", "rt_view_create", "rt_view_free", "rt_view_key", "rt_view_reshape", "rt_view_camera", "rt_view_state", "rt_view_render", "rt_view_render_host", "rt_view_scanline", "rt_view_click",
    "rt_last_error", "rt_version",
]


class rt_pick_event(C.Structure):
    _fields_ = [("path", C.c_uint32), ("kind", C.c_int32), ("arg", C.c_int32), ("flags", C.c_int32), ("p", C.c_float * 3)]


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"raytracer_amd error {code}: {msg}")
        self.code = code


class rt_camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("facing", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3)]


class rt_stats(C.Structure):
    _fields_ = [("rays_nearest", C.c_uint64), ("rays_shadow", C.c_uint64), ("nodes_nearest", C.c_uint64), ("prims_nearest", C.c_uint64),
                ("nodes_shadow", C.c_uint64), ("prims_shadow", C.c_uint64),
                ("levels", C.c_uint64), ("batch_splits", C.c_uint64), ("gpu_ms", C.c_float), ("nearest_ms", C.c_float),
                ("shadow_ms", C.c_float), ("max_walk", C.c_float), ("steps_nearest", C.c_uint64), ("steps_shadow", C.c_uint64),
                ("state_cases", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad"}


class rt_flat_scene(C.Structure):
    _fields_ = ([("fov", C.c_double), ("background", C.c_float * 3), ("n_objects", C.c_int32), ("root_box", C.c_float * 6),
                 ("root_child", C.c_int32), ("root_count", C.c_int32)] +
                [(n, C.c_int32) for n in ("n_planes", "n_spheres", "n_tris", "n_boxes", "n_csgs", "n_csg_nodes", "n_meshes", "n_mesh_tris",
                                          "n_uvs", "n_materials", "n_lights", "n_textures", "n_bvh_nodes", "n_leaf_prims")] +
                [("n_texel_bytes", C.c_int64)] +
                [(n, C.c_void_p) for n in ("planes", "spheres", "tris", "boxes", "csgs", "csg_nodes", "meshes", "mesh_tris", "uvs", "materials",
                                           "lights", "textures", "texels", "bvh_nodes", "leaf_prims")])


_lib = None


def load():
    """Loads the in-tree shared library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make lib` (hipcc, gfx950). There is no fallback path.")
    # PyTorch ships its own libamdhip64.so.7 / libhsa-runtime64 (torch/lib).  Two HIP runtimes in one process do not
    # share a device, so torch's copy must be the one already loaded when this library's NEEDED libamdhip64.so.7 is
    # resolved (the dynamic linker then reuses it by SONAME).  torch is plumbing here: device buffers, streams, RCCL.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_version.restype = C.c_char_p
    lib.rt_tiles_count.restype = C.c_int64
    lib.rt_camera_setup.argtypes = [C.c_double, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(rt_camera)]
    lib.rt_load_scene_json.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.POINTER(rt_flat_scene))]
    lib.rt_load_scene_json_text.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.POINTER(rt_flat_scene))]
    lib.rt_flat_scene_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(rt_flat_scene))]
    lib.rt_flat_scene_save.argtypes = [C.POINTER(rt_flat_scene), C.c_char_p]
    lib.rt_flat_scene_free.argtypes = [C.POINTER(rt_flat_scene)]
    lib.rt_flat_scene_free.restype = None
    lib.rt_scene_upload.argtypes = [C.POINTER(rt_flat_scene), C.c_int, C.POINTER(C.c_void_p)]
    lib.rt_scene_free.argtypes = [C.c_void_p]
    lib.rt_scene_free.restype = None
    lib.rt_render.argtypes = [C.c_void_p, C.POINTER(rt_camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                              C.c_void_p, C.c_void_p, C.POINTER(rt_stats)]
    lib.rt_render_submit.argtypes = [C.c_void_p, C.POINTER(rt_camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.rt_render_wait.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(rt_stats)]
    lib.rt_render_host.argtypes = [C.c_void_p, C.POINTER(rt_camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                   C.c_void_p, C.POINTER(rt_stats)]
    lib.rt_trace_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_uint32,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(rt_stats)]
    lib.rt_trace_pick.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.rt_pick_narration.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    lib.rt_pick_narration.restype = C.c_size_t
    lib.rt_intersect_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(rt_stats)]
    lib.rt_intersect_batch_static.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rt_shadow_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(rt_stats)]
    for name in ("rt_tiles_pack", "rt_tiles_unpack", "rt_tiles_merge", "rt_tiles_pack_rgb8", "rt_tiles_unpack_rgb8", "rt_tiles_quantize_rgb8"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.rt_tiles_count.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.rt_view_create.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.rt_view_free.argtypes = [C.c_void_p]
    lib.rt_view_free.restype = None
    lib.rt_view_key.argtypes = [C.c_void_p, C.c_ubyte, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.rt_view_reshape.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.rt_view_camera.argtypes = [C.c_void_p, C.POINTER(rt_camera), C.POINTER(C.c_int)]
    lib.rt_view_state.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.rt_view_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(rt_stats)]
    lib.rt_view_render_host.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(rt_stats)]
    lib.rt_view_scanline.argtypes = [C.c_int, C.c_int]
    lib.rt_view_click.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.rt_write_bmp.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    lib.rt_quantize_rgb8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    _lib = lib
    return lib


def check(rc):
    if rc != RT_OK:
        raise RtError(rc, load().rt_last_error().decode(errors="replace"))


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(shape) if shape else a
