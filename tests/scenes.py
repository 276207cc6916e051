"""Seeded random test scenes shared by the CPU and GPU parity tests (numpy only)."""
import numpy as np


def rand_rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def gl(T):
    """4x4 row-major numpy -> OpenGL column-major 16-vector."""
    return np.ascontiguousarray(np.asarray(T, np.float64).T).reshape(16).copy()


def projection(fx, fy, cx, cy, W, H, n=0.1, f=8.0):
    """getProjectionMatrix (reference src/urdf_filter.cpp:485-500)."""
    P = np.zeros(16)
    P[0] = -2.0 * fx / W
    P[5] = 2.0 * fy / H
    P[8] = 2.0 * (0.5 - cx / W)
    P[9] = 2.0 * (cy / H - 0.5)
    P[10] = -(f + n) / (f - n)
    P[14] = -2.0 * f * n / (f - n)
    P[11] = -1
    return P


def soup_geometry(rng, n_links=6, tris_per_link=40, scale_lo=0.05, scale_hi=0.6):
    """Random triangle soups, one draw per link: list of (pre_op, op, verts, tris)."""
    geo = []
    for _ in range(n_links):
        v = rng.normal(scale=rng.uniform(scale_lo, scale_hi), size=(tris_per_link * 3, 3)).astype(np.float32)
        t = np.arange(tris_per_link * 3, dtype=np.uint32).reshape(-1, 3)
        pre = int(rng.integers(0, 3))
        if pre == 1:
            op = [float(np.float32(rng.uniform(0.5, 1.5))) for _ in range(3)]
        elif pre == 2:
            op = [float(np.float32(rng.uniform(-0.2, 0.2))) for _ in range(3)]
        else:
            op = [0.0, 0.0, 0.0]
        geo.append((pre, op, v, t))
    return geo


def random_link_poses(rng, n_links, near=False, far=False):
    tfs = []
    for k in range(n_links):
        zlo = 0.05 if near else 0.6
        c = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1, 1), rng.uniform(zlo, 5)])
        if far and k == 0:
            c = np.array([0.0, 0.0, 7.6])
        T = np.eye(4)
        T[:3, :3] = rand_rot(rng)
        T[:3, 3] = c
        tfs.append(gl(T))
    return tfs


def random_camera(rng, small=True):
    """(camera_offset_inv, camera_tf) as GL 16-vectors: small perturbations of identity."""
    Toff = np.eye(4)
    Toff[:3, :3] = rand_rot(rng) if not small else np.eye(3)
    Toff[:3, 3] = rng.uniform(-0.05, 0.05, 3)
    C = np.eye(4)
    a = rng.uniform(-0.15, 0.15)
    C[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    C[:3, 3] = rng.uniform(-0.2, 0.2, 3)
    if not small:
        Toff = np.eye(4)
        Toff[:3, 3] = rng.uniform(-0.05, 0.05, 3)
    return gl(np.linalg.inv(Toff)), gl(C)


def sensor_depth(W, H, phase=0.0, specials=True):
    """SURVEY C1 depth: smooth surface + NaN / zero / +inf pixel classes."""
    yy, xx = np.mgrid[0:H, 0:W]
    d = np.clip(2.5 + 1.5 * np.sin(0.013 * xx + phase) * np.cos(0.017 * yy), 0.4, 7.5).astype(np.float32)
    if specials:
        h = (xx.astype(np.uint64) * np.uint64(2654435761) + yy.astype(np.uint64) * np.uint64(40503) + np.uint64(int(phase * 1000) & 0xffff)) & np.uint64(0xffffffff)
        h = (h ^ (h >> np.uint64(15))) * np.uint64(2246822519) & np.uint64(0xffffffff)
        h ^= h >> np.uint64(13)
        d[(h & np.uint64(63)) == 1] = np.nan
        d[(h & np.uint64(63)) == 2] = 0.0
        d[(h & np.uint64(1023)) == 3] = np.inf
        d[(h & np.uint64(255)) == 4] = 7.9
    return d
