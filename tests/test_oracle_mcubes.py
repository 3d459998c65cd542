"""Marching-cubes oracle (our variant; the reference has none): topological sanity of the
generated case table and of meshes it produces.  CPU only."""
import numpy as np
import pytest

from monoport_amd import synthetic as syn


def sphere_volume(r, radius=0.6, sharp=8.0):
    g = ((np.arange(r) + 0.5) / r) * 2 - 1
    z, y, x = np.meshgrid(g, g, g, indexing="ij")
    d = np.sqrt(x * x + y * y + z * z)
    return (1.0 / (1.0 + np.exp(-sharp * (radius - d) / radius))).astype(np.float32)


def edge_counts(faces):
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    und = np.sort(e, 1)
    _, inv, cnt = np.unique(und, axis=0, return_inverse=True, return_counts=True)
    return e, und, cnt


def test_table_uses_only_crossing_edges(oracle):
    t = oracle._mc_tables()
    assert t["tri"].shape[0] == 256 and t["count"][0] == 0 and t["count"][255] == 0
    for case in range(256):
        inside = [(case >> i) & 1 for i in range(8)]
        crossing = {e for e, (a, b) in enumerate(t["edges"]) if inside[a] != inside[b]}
        used = set(int(v) for v in t["tri"][case, :t["count"][case]].reshape(-1))
        assert used == crossing, case


def test_sphere_mesh_is_closed_oriented_manifold(oracle):
    vol = sphere_volume(33)
    v, f = oracle.marching_cubes(vol)
    assert v.dtype == np.float32 and f.dtype == np.int32 and f.min() == 0 and f.max() == len(v) - 1
    e, und, cnt = edge_counts(f)
    assert (cnt == 2).all()  # watertight: every edge shared by exactly two triangles
    # consistently oriented: each undirected edge appears once in each direction
    key = e[:, 0].astype(np.int64) * len(v) + e[:, 1]
    rev = e[:, 1].astype(np.int64) * len(v) + e[:, 0]
    assert np.array_equal(np.sort(key), np.sort(rev))
    n_edges = len(np.unique(und, axis=0))
    assert len(v) - n_edges + len(f) == 2  # Euler characteristic of a sphere
    # outward orientation: positive signed volume, close to the analytic sphere
    p = v[f].astype(np.float64)
    vol6 = np.einsum("ij,ij->i", p[:, 0], np.cross(p[:, 1], p[:, 2])).sum()
    assert vol6 > 0
    assert abs(vol6 / 6 - 4 / 3 * np.pi * 0.6 ** 3) < 0.05
    # vertices lie on lattice edges: inside the box, near radius 0.6
    rad = np.linalg.norm(v, axis=1)
    assert np.abs(rad - 0.6).max() < 0.08


def test_blob_volume_mesh_is_watertight(oracle):
    """Random blobs contain ambiguous faces: the face-consistent table must stay crack free."""
    vol = syn.blob_volume(33, 5)
    v, f = oracle.marching_cubes(vol)
    _, _, cnt = edge_counts(f)
    assert (cnt == 2).all()


def test_empty_and_full(oracle):
    v, f = oracle.marching_cubes(np.zeros((9, 9, 9), np.float32))
    assert v.shape == (0, 3) and f.shape == (0, 3)
    v, f = oracle.marching_cubes(np.ones((9, 9, 9), np.float32))
    assert v.shape == (0, 3) and f.shape == (0, 3)
