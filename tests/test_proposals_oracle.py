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


# ------------------------------------------------------------------------------ SLIC specification (slic.py:43)
def _thumb(seed, hw):
    """Smooth colour fields + blobs: an H&E-like thumbnail on which superpixels have something to follow."""
    rng = np.random.default_rng(seed)
    h, w = hw
    yy, xx = np.mgrid[:h, :w]
    img = np.empty((h, w, 3), np.float64)
    for c in range(3):
        img[..., c] = 200 + 30 * np.sin(yy / (7.0 + c) + rng.uniform(0, 6)) * np.cos(xx / (9.0 + 2 * c) + rng.uniform(0, 6))
    for _ in range(6):
        cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(6, 20)
        sel = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
        img[sel] = rng.uniform(60, 200, 3)
    img += rng.normal(0, 4, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def test_slic_building_blocks():
    from scipy import ndimage as ndi
    # rgb2lab on known colours (published values: white L = 100, sRGB red = (53.24, 80.09, 67.20))
    lab = PO.rgb2lab(np.array([[[1.0, 1.0, 1.0], [0.0, 0.0, 0.0], [1.0, 0.0, 0.0]]]))
    assert np.allclose(lab[0, 0], (100.0, 0.0, 0.0), atol=2e-2) and np.allclose(lab[0, 1], 0.0, atol=1e-9)
    assert np.allclose(lab[0, 2], (53.24, 80.09, 67.20), atol=5e-2)
    # the Gaussian weights are scipy's, and the separable symmetric pass reproduces scipy bit for bit
    fw, radius = PO.gaussian_weights(5.0)
    assert radius == 20 and len(fw) == 41 and abs(fw.sum() - 1) < 1e-15
    x = np.random.default_rng(1).random((1, 23, 31, 3))
    ref = ndi.gaussian_filter(x, [5.0, 5.0, 5.0, 0])

    def sym_pass(a, axis):
        n = a.shape[axis]
        idx = np.arange(-radius, n + radius)
        idx = np.mod(idx, 2 * n)
        idx = np.where(idx < n, idx, 2 * n - 1 - idx)                      # scipy 'reflect'
        ext = np.take(a, idx, axis)
        sl = lambda o: np.take(ext, np.arange(n) + radius + o, axis)
        out = sl(0) * fw[radius]
        for jj in range(-radius, 0):
            out = out + (sl(jj) + sl(-jj)) * fw[radius + jj]
        return out
    got = sym_pass(sym_pass(sym_pass(x, 0), 1), 2)
    assert np.array_equal(got, ref)                                        # the order of summation the device kernel uses
    # the regular grid: ~n_segments centres, steps as skimage computes them
    segs, sy, sx, step = PO.slic_setup(150, 200, 200)
    assert 150 <= len(segs) <= 260 and step == max(sy, sx) and segs[:, 5].all() and not segs[:, 2:5].any()


@pytest.mark.parametrize('seed,hw,nseg,sigma', [(0, (90, 120), 60, 3.0), (1, (64, 64), 30, 0.0)])
def test_slic_labels_properties(seed, hw, nseg, sigma):
    img = _thumb(seed, hw)
    lab = PO.slic_labels(img, nseg, 20.0, sigma)
    assert np.array_equal(lab, PO.slic_labels(img.copy(), nseg, 20.0, sigma))      # deterministic
    segs, sy, sx, _ = PO.slic_setup(hw[0], hw[1], nseg)
    assert lab.min() >= 0 and lab.max() < len(segs) and len(np.unique(lab)) >= 0.7 * len(segs)
    # every pixel lies inside the final window of some centre: no superpixel is wider than 4 steps + 1
    for k in np.unique(lab):
        ys, xs = np.nonzero(lab == k)
        assert ys.max() - ys.min() <= 4 * sy + 1 and xs.max() - xs.min() <= 4 * sx + 1
    labels, meta = PO.slic_candidates(img, (hw[0] * 2, hw[1] * 2), nseg, 20.0, sigma, us_kmeans=2, n_cnt=3)
    assert labels.shape == (hw[0] * 2, hw[1] * 2) and len(meta) >= 0.5 * len(np.unique(lab))
    for k, r in meta.items():
        assert r['cnt_xy'].shape == (3, 2) and r['perim_xy'].shape[1] == 2 and (labels[r['foreground_indices']] == k).all()


def test_find_nuclei_lab_and_fill_mask_oracle():
    from oracle import wsi_oracle as WO
    img = _thumb(4, (60, 80))
    img[5:25, 10:40] = (180, 60, 150)
    m = WO.find_nuclei_lab(img)
    assert m.dtype == np.uint8 and m[10, 20] == 1 and 0 < m.mean() < 0.9
    ring = np.zeros((40, 40), np.uint8)
    ring[8:30, 8] = ring[8:30, 29] = ring[8, 8:30] = ring[29, 8:30] = 1
    f = WO.fill_mask(ring)
    assert f[18, 18] == 1 and f[2, 2] == 0 and f.sum() >= 22 * 22
