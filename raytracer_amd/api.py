"""Host-side mirror of the reference's Raytracer API over libraytracer_amd.so (no compute happens in Python)."""
import ctypes as C
import os

import numpy as np

from . import capi
from .capi import rt_camera, rt_flat_scene, rt_stats, check, ptr, f32


def _aa(aa):
    """aa: n (the reference's n x n grid: 1 or 2) or an (aa_x, aa_y) pair."""
    return (int(aa), int(aa)) if np.isscalar(aa) else (int(aa[0]), int(aa[1]))


def device_count():
    return int(capi.load().rt_device_count())


class FlatScene:
    """The flat scene IR (include/rt_scene.h) as the loader produces it; host memory, no GPU needed."""

    def __init__(self, handle):
        self._h = handle   # POINTER(rt_flat_scene), owned

    @classmethod
    def from_json(cls, path, asset_dir=None):
        """choose_scene()'s parsing and BVH build (raytracer.cpp:117-305) on a scene file."""
        h = C.POINTER(rt_flat_scene)()
        check(capi.load().rt_load_scene_json(os.fsencode(path), None if asset_dir is None else os.fsencode(asset_dir), C.byref(h)))
        return cls(h)

    @classmethod
    def from_json_text(cls, text, asset_dir=""):
        data = text.encode() if isinstance(text, str) else text
        h = C.POINTER(rt_flat_scene)()
        check(capi.load().rt_load_scene_json_text(data, len(data), os.fsencode(asset_dir), C.byref(h)))
        return cls(h)

    @classmethod
    def load(cls, path):
        h = C.POINTER(rt_flat_scene)()
        check(capi.load().rt_flat_scene_load(os.fsencode(path), C.byref(h)))
        return cls(h)

    def save(self, path):
        check(capi.load().rt_flat_scene_save(self._h, os.fsencode(path)))

    @property
    def info(self):
        return self._h.contents

    @property
    def fov(self):
        return float(self._h.contents.fov)

    @property
    def background(self):
        return np.array(self._h.contents.background[:], np.float32)

    def __del__(self):
        # (at interpreter shutdown the module globals may already be gone: the process is about to release the memory anyway)
        if getattr(self, "_h", None) and capi is not None and getattr(capi, "load", None) is not None:
            capi.load().rt_flat_scene_free(self._h)
            self._h = None


class Camera:
    """setFacing() of q1.cpp:30-44: eye + yaw/pitch -> facing/right/up scaled for the field of view."""

    def __init__(self, fov_deg, width, height, eye=(0, 0, 0), rot_x=0.0, rot_y=0.0):
        self.c = rt_camera()
        e = (C.c_float * 3)(*[float(v) for v in eye])
        check(capi.load().rt_camera_setup(float(fov_deg), int(width), int(height), e, float(rot_x), float(rot_y), C.byref(self.c)))
        self.width, self.height = int(width), int(height)


class Scene:
    """A scene resident in one GPU's HBM plus its work queues (rt_scene)."""

    def __init__(self, flat, device=0):
        self.flat = flat
        self.device = int(device)
        h = C.c_void_p()
        check(capi.load().rt_scene_upload(flat._h, self.device, C.byref(h)))
        self._h = h
        self.last_stats = None

    def close(self):
        if getattr(self, "_h", None):
            capi.load().rt_scene_free(self._h)
            self._h = None

    def __del__(self):
        if capi is not None and getattr(capi, "load", None) is not None:   # (interpreter shutdown: see FlatScene.__del__)
            self.close()

    def camera(self, width, height, eye=(0, 0, 0), rot_x=0.0, rot_y=0.0):
        return Camera(self.flat.fov, width, height, eye, rot_x, rot_y)

    def render(self, width, height, aa=1, max_depth=capi.MAX_REFLECTIONS, seed=1, flags=0, camera=None):
        """One frame into host memory: float32 [H, W, 3], row 0 = bottom row (q1.cpp's GL convention)."""
        cam = camera or self.camera(width, height)
        ax, ay = _aa(aa)
        out = np.empty((height, width, 3), np.float32)
        st = rt_stats()
        check(capi.load().rt_render_host(self._h, C.byref(cam.c), width, height, ax, ay, max_depth, seed, flags, ptr(out), C.byref(st)))
        self.last_stats = st.as_dict()
        return out

    def render_into(self, frame_dev_ptr, width, height, aa=1, max_depth=capi.MAX_REFLECTIONS, seed=1, flags=0, camera=None,
                    tile_rank=0, tile_ranks=1, stream=None):
        """One frame (or this rank's tiles of it) into a DEVICE buffer of width*height*3 floats; returns the stats."""
        cam = camera or self.camera(width, height)
        ax, ay = _aa(aa)
        st = rt_stats()
        check(capi.load().rt_render(self._h, C.byref(cam.c), width, height, ax, ay, max_depth, tile_rank, tile_ranks, seed, flags,
                                    C.c_void_p(frame_dev_ptr), C.c_void_p(stream or 0), C.byref(st)))
        self.last_stats = st.as_dict()
        return self.last_stats

    def render_submit(self, frame_dev_ptr, width, height, aa=1, max_depth=capi.MAX_REFLECTIONS, seed=1, flags=0, camera=None,
                      tile_rank=0, tile_ranks=1, stream=None):
        """Queues a frame (rt_render_submit) and returns its ticket; at most 8 may be outstanding, each into its own buffer."""
        cam = camera or self.camera(width, height)
        ax, ay = _aa(aa)
        ticket = C.c_uint64(0)
        check(capi.load().rt_render_submit(self._h, C.byref(cam.c), width, height, ax, ay, max_depth, tile_rank, tile_ranks, seed, flags,
                                           C.c_void_p(frame_dev_ptr), C.c_void_p(stream or 0), C.byref(ticket)))
        return int(ticket.value)

    def render_wait(self, ticket):
        """Blocks until the frame of `ticket` is complete (rt_render_wait); returns its stats."""
        st = rt_stats()
        check(capi.load().rt_render_wait(self._h, C.c_uint64(ticket), C.byref(st)))
        self.last_stats = st.as_dict()
        return self.last_stats

    def trace_batch(self, e, s, max_depth=capi.MAX_REFLECTIONS, seed=1, flags=0):
        """trace() for n rays (raytracer.cpp:307): returns dict(hit, id, t, rgb)."""
        e, s = f32(e, (-1, 3)), f32(s, (-1, 3))
        n = len(e)
        assert len(s) == n
        hit, oid, t, rgb = np.zeros(n, np.uint8), np.full(n, -1, np.int32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
        st = rt_stats()
        check(capi.load().rt_trace_batch(self._h, ptr(e), ptr(s), n, max_depth, seed, flags, ptr(hit), ptr(oid), ptr(t), ptr(rgb), C.byref(st)))
        self.last_stats = st.as_dict()
        return dict(hit=hit, id=oid, t=t, rgb=rgb)

    def trace_pick(self, e, s, max_depth=capi.MAX_REFLECTIONS, seed=1, cap=1 << 16):
        """trace(e, s, colour, pick=True) (raytracer.cpp:307-327): returns (hit, rgb, narration) -- the text the reference
        prints on std::cout for the picked ray, from the events the kernels record (rt_trace_pick / rt_pick_narration)."""
        e, s = f32(e, (3,)), f32(s, (3,))
        hit, rgb = np.zeros(1, np.uint8), np.zeros(3, np.float32)
        ev = (capi.rt_pick_event * cap)()
        n = C.c_size_t(0)
        lib = capi.load()
        check(lib.rt_trace_pick(self._h, ptr(e), ptr(s), max_depth, seed, ptr(hit), ptr(rgb), C.cast(ev, C.c_void_p), cap, C.byref(n)))
        m = min(n.value, cap)
        size = lib.rt_pick_narration(C.cast(ev, C.c_void_p), m, None, 0) + 1
        buf = C.create_string_buffer(size)
        lib.rt_pick_narration(C.cast(ev, C.c_void_p), m, buf, size)
        return bool(hit[0]), rgb, buf.value.decode()

    def intersect_batch(self, e, d, flags=0):
        """BVH::findNearest for n rays (bvh.cpp:72): returns dict(hit, id, t)."""
        e, d = f32(e, (-1, 3)), f32(d, (-1, 3))
        n = len(e)
        hit, oid, t = np.zeros(n, np.uint8), np.full(n, -1, np.int32), np.zeros(n, np.float32)
        st = rt_stats()
        check(capi.load().rt_intersect_batch(self._h, ptr(e), ptr(d), n, flags, ptr(hit), ptr(oid), ptr(t), C.byref(st)))
        self.last_stats = st.as_dict()
        return dict(hit=hit, id=oid, t=t)

    def intersect_batch_static(self, e_dev_ptr, d_dev_ptr, n, hit_dev_ptr, id_dev_ptr, t_dev_ptr, stream=None):
        """rt_intersect_batch_static: device buffers in and out, no allocation, enqueued on `stream` and not waited for."""
        check(capi.load().rt_intersect_batch_static(self._h, C.c_void_p(e_dev_ptr), C.c_void_p(d_dev_ptr), int(n), C.c_void_p(hit_dev_ptr or 0),
                                                    C.c_void_p(id_dev_ptr or 0), C.c_void_p(t_dev_ptr or 0), C.c_void_p(stream or 0)))

    def shadow_batch(self, point, light_pos, flags=0):
        """shadowRay for n queries (raytracer.cpp:111): returns dict(lit, shadow)."""
        p, l = f32(point, (-1, 3)), f32(light_pos, (-1, 3))
        n = len(p)
        lit, sh = np.zeros(n, np.uint8), np.zeros((n, 3), np.float32)
        st = rt_stats()
        check(capi.load().rt_shadow_batch(self._h, ptr(p), ptr(l), n, flags, ptr(lit), ptr(sh), C.byref(st)))
        self.last_stats = st.as_dict()
        return dict(lit=lit, shadow=sh)

    def tiles_count(self, width, height, tile_rank, tile_ranks):
        return int(capi.load().rt_tiles_count(width, height, tile_rank, tile_ranks))

    def tiles_pack(self, frame_ptr, packed_ptr, width, height, tile_rank, tile_ranks, stream=None):
        check(capi.load().rt_tiles_pack(self._h, C.c_void_p(frame_ptr), C.c_void_p(packed_ptr), width, height, tile_rank, tile_ranks, C.c_void_p(stream or 0)))

    def tiles_unpack(self, packed_ptr, frame_ptr, width, height, tile_rank, tile_ranks, stream=None):
        check(capi.load().rt_tiles_unpack(self._h, C.c_void_p(packed_ptr), C.c_void_p(frame_ptr), width, height, tile_rank, tile_ranks, C.c_void_p(stream or 0)))


class View:
    """The reference's interactive front end without its window (rt_view_*, q1.cpp:107-288): keys move the camera, render() is display()."""

    def __init__(self, scene=None, fov=None, width=512, height=512):
        self.scene = scene
        h = C.c_void_p()
        f = float(fov if fov is not None else (scene.flat.fov if scene is not None else 60.0))
        check(capi.load().rt_view_create(scene._h if scene is not None else None, f, int(width), int(height), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            capi.load().rt_view_free(self._h)
            self._h = None

    def __del__(self):
        if capi is not None and getattr(capi, "load", None) is not None:   # (interpreter shutdown: see FlatScene.__del__)
            self.close()

    def key(self, key):
        """keyboard(key): returns (result, text) -- 1 new frame, 0 none, -1 Escape; text = what the reference prints."""
        buf = C.create_string_buffer(512)
        n = C.c_size_t(0)
        k = key if isinstance(key, int) else ord(key)
        rc = capi.load().rt_view_key(self._h, k, buf, 512, C.byref(n))
        return rc, buf.value.decode()

    def keys(self, text):
        return [self.key(ch) for ch in text]

    def reshape(self, width, height):
        check(capi.load().rt_view_reshape(self._h, int(width), int(height)))

    def camera(self):
        cam = Camera.__new__(Camera)
        cam.c = rt_camera()
        aa = C.c_int(1)
        check(capi.load().rt_view_camera(self._h, C.byref(cam.c), C.byref(aa)))
        st = self.state()
        cam.width, cam.height = st["width"], st["height"]
        return cam, int(aa.value)

    def state(self):
        eye = (C.c_float * 3)()
        rx, ry, w, h, d = C.c_float(), C.c_float(), C.c_int(), C.c_int(), C.c_int()
        check(capi.load().rt_view_state(self._h, eye, C.byref(rx), C.byref(ry), C.byref(w), C.byref(h), C.byref(d)))
        return dict(eye=np.array(eye[:], np.float32), rotation_x=float(rx.value), rotation_y=float(ry.value), width=int(w.value), height=int(h.value),
                    needs_redraw=bool(d.value))

    def render_into(self, frame_dev_ptr, stream=None):
        st = rt_stats()
        check(capi.load().rt_view_render(self._h, C.c_void_p(frame_dev_ptr), C.c_void_p(stream or 0), C.byref(st)))
        return st.as_dict()

    def render(self):
        """display(): the frame as float32 [H, W, 3] on the host, row 0 = bottom."""
        st0 = self.state()
        out = np.empty((st0["height"], st0["width"], 3), np.float32)
        st = rt_stats()
        check(capi.load().rt_view_render_host(self._h, ptr(out), C.byref(st)))
        self.last_stats = st.as_dict()
        return out

    @staticmethod
    def scanline(height, k):
        return int(capi.load().rt_view_scanline(int(height), int(k)))

    def click(self, x, y_from_top):
        buf = C.create_string_buffer(1 << 16)
        n = C.c_size_t(0)
        check(capi.load().rt_view_click(self._h, int(x), int(y_from_top), buf, 1 << 16, C.byref(n)))
        return buf.value.decode()


def _tiles_rgb8(name):
    def call(self, src_ptr, dst_ptr, width, height, tile_rank, tile_ranks, stream=None):
        check(getattr(capi.load(), name)(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), width, height, tile_rank, tile_ranks, C.c_void_p(stream or 0)))
    call.__doc__ = name + " (include/rt_capi.h): the tile exchange with 8-bit pixels, quantised on the device"
    return call


Scene.tiles_pack_rgb8 = _tiles_rgb8("rt_tiles_pack_rgb8")
Scene.tiles_unpack_rgb8 = _tiles_rgb8("rt_tiles_unpack_rgb8")
Scene.tiles_quantize_rgb8 = _tiles_rgb8("rt_tiles_quantize_rgb8")


def quantize_rgb8(rgb):
    """[H, W, 3] float, row 0 = bottom  ->  [H, W, 3] uint8, row 0 = top (clamp, *255 + 0.5, NaN -> 0)."""
    rgb = f32(rgb)
    h, w = rgb.shape[:2]
    out = np.empty((h, w, 3), np.uint8)
    check(capi.load().rt_quantize_rgb8(ptr(rgb), w, h, ptr(out)))
    return out


def write_bmp(path, rgb):
    rgb = f32(rgb)
    h, w = rgb.shape[:2]
    check(capi.load().rt_write_bmp(os.fsencode(path), ptr(rgb), w, h))


# ---- the reference's process-global API (raytracer.h:12-18) ------------------------------------------------
_current = None
fov = 60.0
background_colour = np.zeros(3, np.float32)


def choose_scene(fn=None, device=0):
    """choose_scene(fn) (raytracer.cpp:117): a bare name resolves to scenes/<fn>.json, default "c"."""
    global _current, fov, background_colour
    if fn is None:
        fn = "c"
    path = fn if (fn.endswith(".json") or fn.endswith(".rtscene")) else os.path.join("scenes", fn + ".json")
    flat = FlatScene.load(path) if path.endswith(".rtscene") else FlatScene.from_json(path)
    _current = Scene(flat, device)
    fov, background_colour = flat.fov, flat.background
    return _current


def current_scene():
    if _current is None:
        raise RuntimeError("choose_scene() has not been called")
    return _current


def trace(e, s, pick=False, reflectionCount=0):
    """trace(e, s, colour, pick, reflectionCount) -> (hit, colour); colour is None on a miss (left untouched in the reference)."""
    if reflectionCount > capi.MAX_REFLECTIONS:
        if pick:
            print("Maximum number of reflections reached.")
        return False, np.zeros(3, np.float32)
    if pick:   # the reference's narration of the picked ray, on stdout
        hit, rgb, text = current_scene().trace_pick(e, s, max_depth=capi.MAX_REFLECTIONS - max(reflectionCount, 0))
        print(text, end="")
        return (True, rgb) if hit else (False, None)
    r = current_scene().trace_batch(np.asarray(e, np.float32)[None], np.asarray(s, np.float32)[None],
                                    max_depth=capi.MAX_REFLECTIONS - max(reflectionCount, 0))
    if not r["hit"][0]:
        return False, None
    return True, r["rgb"][0]


def shadowRay(point, lightPos):
    """shadowRay(point, lightPos, shadow) -> (lit, shadow attenuation starting from (1,1,1))."""
    r = current_scene().shadow_batch(np.asarray(point, np.float32)[None], np.asarray(lightPos, np.float32)[None])
    return bool(r["lit"][0]), r["shadow"][0]
