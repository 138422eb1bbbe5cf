"""Triangle sets for the BVH-builder parity tests (buildBVH, main.cu:20-233): ordinary geometry and
inputs chosen to drive every branch of the builder — the centroid-mean retry (:192-202), forced
oversize leaves (:213-221), the median fallback (:119-128), ties on the split plane, +/-0 bounds."""
import numpy as np


def arrays_of(tris):
    """tris float32 [n,3,3] -> (points float32 [3n,4], mesh int32 [n,20]) in the reference's layouts
    (float4 positions, 80-byte Triangle with aInd/bInd/cInd first, objects.cuh:159-172)."""
    tris = np.ascontiguousarray(tris, np.float32)
    n = tris.shape[0]
    pts = np.zeros((3 * n, 4), np.float32)
    pts[:, :3] = tris.reshape(-1, 3)
    mesh = np.zeros((n, 20), np.int32)
    mesh[:, 0] = 3 * np.arange(n); mesh[:, 1] = mesh[:, 0] + 1; mesh[:, 2] = mesh[:, 0] + 2
    mesh[:, 9] = 2
    mesh[:, 16] = -51
    mesh[:, 17] = np.arange(n)
    return pts, mesh


def shared_vertex_arrays(nx, ny, rng):
    """An indexed height-field (vertices shared between triangles, like an OBJ)."""
    xs, ys = np.meshgrid(np.linspace(-2, 2, nx + 1), np.linspace(-1, 1, ny + 1), indexing="ij")
    z = 0.3 * np.sin(3 * xs) * np.cos(2 * ys) + 0.02 * rng.standard_normal(xs.shape)
    pts = np.zeros(((nx + 1) * (ny + 1), 4), np.float32)
    pts[:, 0], pts[:, 1], pts[:, 2] = xs.ravel(), ys.ravel(), z.ravel()
    vid = lambda i, j: i * (ny + 1) + j
    tri = []
    for i in range(nx):
        for j in range(ny):
            tri.append((vid(i, j), vid(i + 1, j), vid(i + 1, j + 1)))
            tri.append((vid(i, j), vid(i + 1, j + 1), vid(i, j + 1)))
    tri = np.array(tri, np.int32)
    mesh = np.zeros((len(tri), 20), np.int32)
    mesh[:, 0:3] = tri
    mesh[:, 9] = 2; mesh[:, 16] = -51; mesh[:, 17] = np.arange(len(tri))
    return pts, mesh


def _soup(rng, n, size=0.1, box=2.0):
    c = (rng.random((n, 1, 3)) - 0.5) * 2 * box
    return (c + (rng.random((n, 3, 3)) - 0.5) * size).astype(np.float32)


def cases():
    rng = np.random.default_rng(20240)
    out = {}
    out["soup5k"] = arrays_of(_soup(rng, 5000))
    out["soup_flat"] = arrays_of(_soup(rng, 3000) * np.array([1.0, 1e-3, 0.2], np.float32))
    out["heightfield"] = shared_vertex_arrays(40, 30, rng)
    # quads on a lattice: many equal centroids per axis (ties on every split plane)
    g = np.stack(np.meshgrid(np.arange(24), np.arange(20), np.arange(3), indexing="ij"), -1).reshape(-1, 1, 3).astype(np.float32)
    out["lattice"] = arrays_of(g + np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32))
    # 700 copies of 3 triangles: all splits fail -> mean retry -> forced oversize leaves
    base = _soup(rng, 3, size=1.0)
    out["dupes"] = arrays_of(np.concatenate([base] * 700))
    # every triangle spans the whole extent on x: all centroids in one bucket -> median fallback
    s = np.zeros((600, 3, 3), np.float32)
    s[:, 0] = [-10, 0, 0]; s[:, 1] = [10, 0, 0]; s[:, 2] = [0, 0.01, 0]
    s[:, :, 1] += (rng.random((600, 1)) * 0.5).astype(np.float32)
    s[:, :, 2] += (rng.random((600, 1)) * 0.5).astype(np.float32)
    s[:, 2, 0] += (rng.random(600) * 0.4 - 0.2).astype(np.float32)
    out["slivers"] = arrays_of(s)
    # same, with exactly equal centroids in pairs (the index tie-break of the fallback's order)
    out["slivers_tied"] = arrays_of(np.concatenate([s[:200], s[:200]]))
    # enough of them that the device builder's large-node sort (bitonic network) runs, ties included
    sb = np.zeros((5000, 3, 3), np.float32)
    sb[:, 0] = [-10, 0, 0]; sb[:, 1] = [10, 0, 0]; sb[:, 2] = [0, 0.01, 0]
    sb[:, :, 1] += (rng.random((5000, 1)) * 0.5).astype(np.float32)
    sb[:, :, 2] += (rng.random((5000, 1)) * 0.5).astype(np.float32)
    sb[:, 2, 0] += np.round(rng.random(5000) * 40 - 20).astype(np.float32) / 100
    out["slivers_big"] = arrays_of(sb)
    # one huge triangle over many small ones: the best split isolates one primitive on the right
    big = np.array([[[-50, -50, -1], [50, -50, -1], [0, 80, -1]]], np.float32)
    out["big_small"] = arrays_of(np.concatenate([_soup(rng, 2000, box=1.0), big, _soup(rng, 50, box=30.0, size=5.0)]))
    # centroids on a line, in clusters
    t = np.repeat(rng.random(40), 50)[:, None, None].astype(np.float32)
    out["clusters"] = arrays_of(t * np.array([4.0, 0.5, 0.1], np.float32) + (rng.random((2000, 3, 3)).astype(np.float32) - 0.5) * 1e-3)
    # +/-0 coordinates and degenerate (point / segment) triangles
    z = _soup(rng, 400, box=1.0)
    z[::3, :, 0] = 0.0; z[1::3, :, 0] = -0.0; z[::5] = z[::5, :1]
    z[::7, :, 1] = np.float32(1e-6) - np.float32(1e-6)
    out["zeros"] = arrays_of(z)
    # median fallback on a LARGE node (> 1024 primitives: the device builder's bitonic sort) whose centroids on the split
    # axis mix -0.0 and +0.0, as Blender-exported OBJs do: the comparator `ca < cb` holds them equal (ties by index), an
    # order-preserving integer key does not unless the zero is canonicalised. a.x + b.x + c.x is -0 only if all three are.
    nz = 6000
    sz = np.zeros((nz, 3, 3), np.float32)
    yy = (rng.random((nz, 1)) * 0.5).astype(np.float32); zz = (rng.random((nz, 1)) * 0.5).astype(np.float32)
    sz[:, 0] = [-10, 0, 0]; sz[:, 1] = [10, 0, 0]; sz[:, 2] = [0, 0.01, 0]          # wide slivers: centroid.x = +0 ...
    sz[:, :, 1] += yy; sz[:, :, 2] += zz
    pos = np.arange(nz) % 5 >= 2                                                    # ... or a small positive value (60 %)
    sz[pos, 2, 0] = (np.round(rng.random(int(pos.sum())) * 20 + 1) / 100).astype(np.float32)
    neg = np.arange(nz) % 5 == 0                                                    # 20 %: triangles in the x = -0 plane
    sz[neg, :, 0] = np.float32(-0.0)
    sz[neg, 1, 1] += np.float32(0.3); sz[neg, 2, 2] += np.float32(0.3)
    out["signed_zero_median"] = arrays_of(sz)
    for n in (1, 2, 3, 5):
        out["tiny%d" % n] = arrays_of(_soup(rng, n, size=0.5, box=1.0))
    return out


LEAF_SIZES = (1, 2, 4, 8)
