"""Seeded triangle groups of awkward kinds for the device-side tree build (tests/test_gpu_mesh.py, tools/fuzz_mesh.py): soups,
regular grids (ties between candidate planes), flat sheets (an axis of zero extent), duplicated triangles and shared centres,
slivers, wide ranges of scale, two far clusters.  kind = seed % 7."""
import numpy as np


def triangles(seed):
    rng = np.random.default_rng(seed)
    kind = seed % 7
    n = int(rng.choice([64, 65, 100, 257, 1000, 3000]))
    if kind == 0:    # soup
        c = rng.uniform(-1, 1, (n, 1, 3)); p = c + rng.uniform(-0.2, 0.2, (n, 3, 3))
    elif kind == 1:  # regular grid of equal triangles: equal costs, centres on the candidate planes
        k = int(np.ceil(n ** (1 / 3))); g = np.stack(np.meshgrid(*[np.arange(k)] * 3, indexing="ij"), -1).reshape(-1, 1, 3)[:n].astype(float)
        p = g * 0.25 + np.array([[0, 0, 0], [0.2, 0, 0], [0, 0.2, 0]])[None]
    elif kind == 2:  # a flat sheet (zero extent in y)
        c = rng.uniform(-1, 1, (n, 1, 3)); p = c + rng.uniform(-0.1, 0.1, (n, 3, 3)); p[:, :, 1] = 0.5
    elif kind == 3:  # duplicates and shared centres
        m = max(8, n // 4); c = rng.uniform(-1, 1, (m, 1, 3)); base = c + rng.uniform(-0.2, 0.2, (m, 3, 3)); p = base[rng.integers(0, m, n)]
    elif kind == 4:  # slivers along one axis
        c = rng.uniform(-1, 1, (n, 1, 3)); d = rng.uniform(-0.5, 0.5, (n, 3, 3)); d[:, :, 1:] *= 1e-6; p = c + d
    elif kind == 5:  # scales from 1e-4 to 1e2
        c = rng.uniform(-1, 1, (n, 1, 3)) * 10.0 ** rng.uniform(-4, 2, (n, 1, 1)); p = c + rng.uniform(-0.2, 0.2, (n, 3, 3)) * 10.0 ** rng.uniform(-4, 1, (n, 1, 1))
    else:            # two far clusters, one of them a single point's neighbourhood
        c = np.where(rng.random((n, 1, 1)) < 0.5, -5.0, 5.0) + rng.uniform(-1e-3, 1e-3, (n, 1, 3)); p = c + rng.uniform(-1e-3, 1e-3, (n, 3, 3))
    return p.reshape(n, 9)
