"""Seeded synthetic registration inputs: icospheres, smooth band-limited
features and a known warp.  Pure numpy; used by tests/ and bench.py to feed the product path and the
oracle with identical inputs.  Not part of the hot path.
"""
import numpy as np

RAD = 100.0


def smooth_feature(xyz, d=0, seed=1234):
    """Smooth feature of direction: low-order polynomials + sinusoids, amplitude O(1), phase from seed+d."""
    rng = np.random.default_rng(seed + d)
    u = np.asarray(xyz, dtype=np.float64) / RAD
    x, y, z = u[:, 0], u[:, 1], u[:, 2]
    c = rng.uniform(-1.0, 1.0, size=12)
    ph = rng.uniform(0.0, 2 * np.pi, size=6)
    k = rng.uniform(2.0, 7.0, size=(6, 3))
    f = c[0] * x + c[1] * y + c[2] * z + c[3] * x * y + c[4] * y * z + c[5] * x * z + c[6] * (x * x - y * y)
    for j in range(6):
        f = f + 0.5 * c[6 + j % 6] * np.sin(k[j, 0] * x + k[j, 1] * y + k[j, 2] * z + ph[j])
    return f


def features(xyz, D, seed=1234):
    """D x V feature matrix (the reference's pvalues layout)."""
    return np.stack([smooth_feature(xyz, d, seed) for d in range(D)])


def rotation(axis, angle_deg):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    t = np.deg2rad(angle_deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * (K @ K)


def known_warp(xyz, seed=7, rot_deg=3.0, amp=1.0):
    """Global rotation + smooth tangential displacement of at most ~amp (sphere units), back on radius 100."""
    rng = np.random.default_rng(seed)
    axis = rng.normal(size=3)
    R = rotation(axis, rot_deg)
    p = np.asarray(xyz, dtype=np.float64) @ R.T
    u = p / RAD
    k = rng.uniform(1.5, 4.0, size=(3, 3))
    ph = rng.uniform(0, 2 * np.pi, size=3)
    disp = np.stack([np.sin(u @ k[j] + ph[j]) for j in range(3)], axis=1) * amp / np.sqrt(3.0)
    disp = disp - u * np.sum(disp * u, axis=1, keepdims=True)  # tangential part
    q = p + disp
    return q / np.linalg.norm(q, axis=1, keepdims=True) * RAD


def random_sphere_points(n, seed=0, radius=RAD):
    rng = np.random.default_rng(seed)
    p = rng.normal(size=(n, 3))
    return p / np.linalg.norm(p, axis=1, keepdims=True) * radius


def anatomy(sphere_xyz, seed=0, base=60.0):
    """a smooth star-shaped "cortical" surface on the vertices of a sphere (V x 3): radius base + a few mm of smooth relief -- the anatomical
    surface (--inanat / --refanat) of a synthetic subject"""
    d = np.asarray(sphere_xyz, dtype=np.float64)
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    r = base + 6.0 * smooth_feature(d * RAD, 0, seed) + 3.0 * smooth_feature(d * RAD, 1, seed + 1) + 1.5 * smooth_feature(d * RAD, 2, seed + 2)
    return d * r[:, None]
