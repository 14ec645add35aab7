"""CPU checks of oracle/proposals_oracle.py (the spec of the region-proposal kernels): labelling order against
scipy.ndimage.label, k-means determinism and fixed point, candidate generation on a synthetic thumbnail."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import proposals_oracle as PO


def blobs(seed, hw=(96, 128), n=9):
    rng = np.random.default_rng(seed)
    img = np.zeros(hw, np.uint8)
    yy, xx = np.mgrid[:hw[0], :hw[1]]
    for _ in range(n):
        cy, cx = rng.integers(5, hw[0] - 5), rng.integers(5, hw[1] - 5)
        ry, rx = rng.integers(3, 14), rng.integers(3, 18)
        img[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 1
    return img


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_cc_matches_scipy_raster_order(seed):
    rng = np.random.default_rng(seed)
    mask = (rng.random((40, 57)) < 0.42).astype(np.uint8)
    got = PO.connected_components(mask)
    ref, n = ndimage.label(mask, structure=np.ones((3, 3)))
    assert got.max() == n
    # same partition ...
    pairs = np.unique(np.stack((got.ravel(), ref.ravel())), axis=1)
    assert pairs.shape[1] == n + 1
    # ... and numbered by the raster position of each component's first pixel
    first = [np.flatnonzero(got.ravel() == l)[0] for l in range(1, n + 1)]
    assert first == sorted(first)


def test_cc_edge_cases():
    assert PO.connected_components(np.zeros((5, 7), np.uint8)).max() == 0
    assert (PO.connected_components(np.ones((5, 7), np.uint8)) == 1).all()
    diag = np.eye(6, dtype=np.uint8)
    assert PO.connected_components(diag).max() == 1                         # 8-connectivity
    checker = (np.indices((6, 6)).sum(0) % 2).astype(np.uint8)
    assert PO.connected_components(checker).max() == 1


def test_kmeans_fixed_point_and_determinism():
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.integers(0, 20, (60, 2)), rng.integers(40, 60, (50, 2)), rng.integers(80, 99, (70, 2))])
    pts = pts[np.lexsort((pts[:, 0], pts[:, 1]))]
    c1, l1 = PO.kmeans(pts, 3)
    c2, l2 = PO.kmeans(pts.copy(), 3)
    assert np.array_equal(c1, c2) and np.array_equal(l1, l2)
    for j in range(3):                                                      # centres are the means of their members
        assert np.allclose(c1[j], pts[l1 == j].mean(0))
    d = ((pts[:, None, :] - c1[None]) ** 2).sum(-1)
    assert np.array_equal(np.argmin(d, 1), l1)


def test_key_points_and_candidates():
    gt = blobs(5)
    n, cnt, out, fgi = PO.get_key_points(gt, 4, 3)
    assert n == 3 and cnt.shape == (3, 2) and out.shape == gt.shape
    assert set(np.unique(out)) <= {0, 1, 2, 3}
    assert PO.get_key_points(np.zeros((32, 32)), 4, 3) == (None, None, None, None)
    tissue = np.ones_like(gt)
    meta = PO.scannet_candidates(gt, tissue)
    assert len(meta) >= 1
    for k, r in meta.items():
        assert r['tile_id'] == k and r['cnt_xy'].shape[1] == 2 and r['perim_xy'].shape[1] == 2
        assert len(r['foreground_indices'][0]) > 0
