"""model/kpconv/ops/grid_subsample.py + radius_search.py on the GPU (cofii2p_amd/neighbors.py) against oracle/neighbors_oracle.py
(restated from the published C++ of the un-vendored extension: parity with the extension itself is unpinned).  Bit-exact: integer
index rows, float32 barycentres computed in the same operation order.  Needs a real MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import neighbors_oracle as NO  # noqa: E402

DEV = "cuda:0"


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def cloud(seed, n, extent=20.0):
    g = np.random.default_rng(seed)
    p = g.uniform(-extent, extent, (n, 3)).astype(np.float32)
    p[:, 2] *= 0.1
    return p


@pytest.mark.parametrize("lengths,voxel", [([5000], 0.3), ([20480, 9000, 1], 0.6), ([300, 0, 700], 2.5), ([40000], 0.15)])
def test_grid_subsample_against_oracle(lengths, voxel):
    from model.kpconv.ops import grid_subsample   # the reference's import path

    pts = np.concatenate([cloud(7 + i, n) + 3.0 * i for i, n in enumerate(lengths)]) if sum(lengths) else np.zeros((0, 3), np.float32)
    pts[:50] = pts[0]                               # duplicates
    pts[50:100] = np.round(pts[50:100] / voxel) * voxel   # points on cell faces
    want_p, want_l = NO.grid_subsample(pts, lengths, voxel)
    got_p, got_l = grid_subsample(G(pts), torch.tensor(lengths), voxel)
    assert got_l.tolist() == want_l.tolist() and got_l.dtype == torch.int64
    assert np.array_equal(got_p.cpu().numpy(), want_p)
    # size-independent properties: one barycentre per occupied cell (ascending cell order), mass preserved cloud by cloud
    gp, start, o = got_p.cpu().numpy().astype(np.float64), 0, 0
    for n, m in zip(lengths, want_l.tolist()):
        if n:
            seg = pts[start:start + n]
            keys, cnt = np.unique(_cell_keys(seg, voxel), return_counts=True)
            assert len(keys) == m
            assert (_cell_keys_of(gp[o:o + m].astype(np.float32), seg, voxel) == keys).mean() > 0.999   # a barycentre may round onto a face
            np.testing.assert_allclose((gp[o:o + m] * cnt[:, None]).sum(0) / n, seg.astype(np.float64).mean(0), atol=1e-3)
        start, o = start + n, o + m


def _cell_keys_of(bary, seg, voxel):
    dl = np.float32(voxel)
    origin = np.floor(seg.min(0) * (np.float32(1) / dl)) * dl
    c = np.floor((bary - origin) / dl).astype(np.int64)
    return (c[:, 2] << 26) | (c[:, 1] << 13) | c[:, 0]


def _cell_keys(p, voxel):
    dl = np.float32(voxel)
    origin = np.floor(p.min(0) * (np.float32(1) / dl)) * dl
    c = np.floor((p - origin) / dl).astype(np.int64)
    return (c[:, 2] << 26) | (c[:, 1] << 13) | c[:, 0]


@pytest.mark.parametrize("ql,sl,radius,limit", [([700], [3000], 2.0, 64), ([2000, 500], [5000, 1500], 1.2, 128), ([100, 0, 50], [40, 10, 900], 3.0, 32),
                                                 ([300], [300], 0.05, 16), ([64], [20], 100.0, 32)])
def test_radius_search_against_oracle(ql, sl, radius, limit):
    from model.kpconv.ops import radius_search   # the reference's import path

    s = np.concatenate([cloud(11 + i, n) for i, n in enumerate(sl)])
    q = np.concatenate([cloud(31 + i, n) for i, n in enumerate(ql)]) if sum(ql) else np.zeros((0, 3), np.float32)
    if ql[0] <= sl[0]:
        q[:ql[0] // 2] = s[:ql[0] // 2]            # queries that coincide with support points (distance 0)
    want = NO.radius_search(q, s, ql, sl, radius, limit)
    got = radius_search(G(q), G(s), torch.tensor(ql), torch.tensor(sl), radius, limit)
    assert got.dtype == torch.int64 and tuple(got.shape) == want.shape
    assert np.array_equal(got.cpu().numpy(), want)
    # properties: every kept index lies in the query's own cloud and inside the radius, rows are sorted by distance, fill = total support
    total = s.shape[0]
    g = got.cpu().numpy()
    q0 = s0 = 0
    for nq, ns in zip(ql, sl):
        rows = g[q0:q0 + nq]
        real = rows != total
        assert ((rows[real] >= s0) & (rows[real] < s0 + ns)).all()
        for r in range(0, nq, max(1, nq // 25)):
            ids = rows[r][real[r]]
            d = ((s[ids].astype(np.float64) - q[q0 + r].astype(np.float64)) ** 2).sum(1)
            assert (d < radius * radius * (1 + 1e-5) + 1e-6).all() and (np.diff(d) >= -1e-4).all()
            assert real[r].sum() == len(ids) and not real[r][len(ids):].any()   # the fill sits behind the neighbours
        q0, s0 = q0 + nq, s0 + ns


def test_radius_search_rejects_bad_arguments():
    from cofii2p_amd import _lib, neighbors

    q, s = G(cloud(1, 10)), G(cloud(2, 20))
    with pytest.raises(_lib.CofiError):
        neighbors.radius_search(q, s, torch.tensor([10]), torch.tensor([20]), 1.0, 0)      # the extension's unlimited mode
    with pytest.raises(_lib.CofiError):
        neighbors.radius_search(q, s, torch.tensor([9]), torch.tensor([20]), 1.0, 16)
    with pytest.raises(_lib.CofiError):
        neighbors.grid_subsample(q, torch.tensor([4, 4]), 0.5)
