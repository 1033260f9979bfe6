"""ctypes binding of oracle/liboracle_restate.so -- the CHECKER.  Imported only by tests/, smoke() and bench.py's
cpu_baseline leg; the product package never imports it."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "liboracle_restate.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} missing: run `make oracle`")
        _lib = C.CDLL(LIB_PATH)
    return _lib


class own_texels:
    """Inside this context the restatement lights every hit on a textured triangle with its own texel -- what the device's level-by-level
    pipeline computes -- instead of reproducing the reference's texel state (oracle/restate.h: oracle_texel_state)."""
    def __enter__(self):
        self.prev = lib().oracle_texel_state(0)
    def __exit__(self, *exc):
        lib().oracle_texel_state(self.prev)


def texel_cases(reset=True):
    """Hits the restatement lit with another hit's texel since the last reset (oracle_texel_cases)."""
    f = lib().oracle_texel_cases
    f.restype = C.c_uint64
    return int(f(1 if reset else 0))


def state_cases(reset=True):
    """Exit searches of transmitted rays that found no exit since the last reset (oracle_state_cases)."""
    f = lib().oracle_state_cases
    f.restype = C.c_uint64
    return int(f(1 if reset else 0))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def trace_batch(flat, e, s, max_depth=16, seed=0, threads=1):
    e = np.ascontiguousarray(e, np.float32).reshape(-1, 3)
    s = np.ascontiguousarray(s, np.float32).reshape(-1, 3)
    n = len(e)
    hit, oid, t, rgb = np.zeros(n, np.uint8), np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    cnt = (C.c_uint64 * 2)(0, 0)
    rc = lib().oracle_trace_batch(flat._h, _p(e), _p(s), C.c_size_t(n), int(max_depth), C.c_uint32(seed), _p(hit), _p(oid), _p(t), _p(rgb), cnt, int(threads))
    assert rc == 0
    return dict(hit=hit, id=oid, t=t, rgb=rgb, rays_nearest=int(cnt[0]), rays_shadow=int(cnt[1]))


def intersect_batch(flat, e, d):
    e = np.ascontiguousarray(e, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
    n = len(e)
    hit, oid, t = np.zeros(n, np.uint8), np.zeros(n, np.int32), np.zeros(n, np.float32)
    assert lib().oracle_intersect_batch(flat._h, _p(e), _p(d), C.c_size_t(n), _p(hit), _p(oid), _p(t)) == 0
    return dict(hit=hit, id=oid, t=t)


def shadow_batch(flat, p, l):
    p = np.ascontiguousarray(p, np.float32).reshape(-1, 3)
    l = np.ascontiguousarray(l, np.float32).reshape(-1, 3)
    n = len(p)
    lit, sh = np.zeros(n, np.uint8), np.zeros((n, 3), np.float32)
    assert lib().oracle_shadow_batch(flat._h, _p(p), _p(l), C.c_size_t(n), _p(lit), _p(sh)) == 0
    return dict(lit=lit, shadow=sh)


def render(flat, cam, width, height, aa=1, max_depth=16, seed=0, threads=1):
    rgb = np.zeros((height, width, 3), np.float32)
    cnt = (C.c_uint64 * 2)(0, 0)
    ax, ay = (int(aa), int(aa)) if np.isscalar(aa) else (int(aa[0]), int(aa[1]))
    rc = lib().oracle_render(flat._h, C.byref(cam.c), int(width), int(height), ax, ay, int(max_depth), C.c_uint32(seed), _p(rgb), cnt, int(threads))
    assert rc == 0
    return rgb, int(cnt[0]), int(cnt[1])


def primary_targets(cam, width, height, aa=1):
    """s(x,y) for every pixel / sample, in ray-index order (y*W+x)*aa^2+k."""
    out = np.zeros((height * width * aa * aa, 3), np.float32)
    buf = (C.c_float * 3)()
    f = lib().oracle_primary_target
    i = 0
    for y in range(height):
        for x in range(width):
            for k in range(aa * aa):
                f(C.byref(cam.c), int(width), int(height), x, y, int(aa), int(aa), k, buf)
                out[i] = buf[:]
                i += 1
    return out
