"""Reader for the binary dumps written by oracle/ref_driver.cpp (format RTREF02)."""
import numpy as np


def read_ref_dump(path):
    buf = open(path, "rb").read()
    assert buf[:7] == b"RTREF02", "not a ref_driver dump"
    off = 8
    hdr = np.frombuffer(buf, np.int32, 8, off); off += 32
    W, H, aa, n_near, n_shad, n_ids = (int(v) for v in hdr[:6])
    counts = np.frombuffer(buf, np.uint64, 2, off); off += 16
    fov = float(np.frombuffer(buf, np.float64, 1, off)[0]); off += 8
    cam = np.frombuffer(buf, np.float32, 15, off).copy(); off += 60
    npx = W * H
    out = dict(W=W, H=H, aa=aa, n_ids=n_ids, rays_nearest=int(counts[0]), rays_shadow=int(counts[1]), fov=fov,
               eye=cam[0:3], facing=cam[3:6], right=cam[6:9], up=cam[9:12], background=cam[12:15])

    def take(dtype, n, shape=None):
        nonlocal off
        a = np.frombuffer(buf, dtype, n, off).copy()
        off += a.nbytes
        return a.reshape(shape) if shape else a
    out["hit"] = take(np.uint8, npx)
    out["id"] = take(np.int32, npx)
    out["t"] = take(np.float32, npx)
    out["rgb"] = take(np.float32, npx * 3, (npx, 3))
    out["s"] = take(np.float32, npx * 3, (npx, 3))
    out["cnt_near"] = take(np.uint32, npx)
    out["cnt_shadow"] = take(np.uint32, npx)
    near = take(np.dtype([("e", "<f4", 3), ("d", "<f4", 3), ("id", "<i4"), ("t", "<f4")]), n_near)
    shad = take(np.dtype([("p", "<f4", 3), ("l", "<f4", 3), ("lit", "<i4"), ("shadow", "<f4", 3)]), n_shad)
    out["sec_e"], out["sec_d"], out["sec_id"], out["sec_t"] = near["e"].copy(), near["d"].copy(), near["id"].copy(), near["t"].copy()
    out["shd_p"], out["shd_l"], out["shd_lit"], out["shd_shadow"] = shad["p"].copy(), shad["l"].copy(), shad["lit"].copy(), shad["shadow"].copy()
    assert off == len(buf), (off, len(buf))
    return out
