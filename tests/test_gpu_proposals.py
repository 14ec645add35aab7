"""Region-proposal kernels against oracle/proposals_oracle.py: bit-exact masks, labels, k-means labels / centres and the
`metadata` of scannet_candidates (reference scannet.py:55-127, utils/regiontools.py:68-102, utils/preprocessing.py:94-98)."""
import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import proposals_oracle as PO
from oracle.wsi_oracle import find_nuclei_hsv
from tests.test_proposals_oracle import blobs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def P():
    from wsi_segmentation_pipeline_amd import proposals
    return proposals


def test_find_nuclei_hsv_exact(P):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (301, 517, 3), dtype=np.uint8)
    img[:40] = 255
    img[40:60] = 0
    img[60:80, :, :] = rng.integers(0, 256, (20, 517, 1), dtype=np.uint8)         # grey: delta == 0
    for mu in (0.1, 0.05, 0.5):
        got = P.find_nuclei(torch.from_numpy(img).cuda(), mu).cpu().numpy()
        assert np.array_equal(got, find_nuclei_hsv(img, mu))
    rgba = np.concatenate((img, np.full(img.shape[:2] + (1,), 255, np.uint8)), -1)
    assert np.array_equal(P.find_nuclei(torch.from_numpy(rgba).cuda()).cpu().numpy(), find_nuclei_hsv(img))
    from utils import preprocessing
    assert np.array_equal(preprocessing.find_nuclei(torch.from_numpy(img).cuda()).cpu().numpy(), preprocessing.find_nuclei(img))


@pytest.mark.parametrize('seed,hw,p', [(0, (40, 57), 0.42), (1, (64, 64), 0.6), (2, (33, 130), 0.3), (3, (1, 50), 0.5), (4, (50, 1), 0.5)])
def test_cc_exact_small(P, seed, hw, p):
    mask = (np.random.default_rng(seed).random(hw) < p).astype(np.uint8)
    lab, n = P.connected_components(torch.from_numpy(mask).cuda())
    ref = PO.connected_components(mask)
    assert n == ref.max()
    assert np.array_equal(lab.cpu().numpy(), ref)


def test_cc_edge_cases(P):
    for m in (np.zeros((9, 11), np.uint8), np.ones((9, 11), np.uint8), np.eye(16, dtype=np.uint8),
              (np.indices((31, 31)).sum(0) % 2).astype(np.uint8)):
        lab, n = P.connected_components(torch.from_numpy(m).cuda())
        assert np.array_equal(lab.cpu().numpy(), PO.connected_components(m))
    # a long serpentine: one component whose union chain crosses the whole image
    s = np.zeros((65, 200), np.uint8)
    s[::2] = 1
    s[1::4, -1] = 1
    s[3::4, 0] = 1
    lab, n = P.connected_components(torch.from_numpy(s).cuda())
    assert n == 1 and np.array_equal(lab.cpu().numpy(), s.astype(np.int32))


def test_cc_full_size_vs_scipy(P):
    """A 2500 x 2500 thumbnail (cfg5 scale): same partition as scipy.ndimage.label and labels in raster order of first pixels."""
    rng = np.random.default_rng(7)
    coarse = rng.random((125, 125)) < 0.45
    mask = np.kron(coarse, np.ones((20, 20), bool)) & (rng.random((2500, 2500)) < 0.97)
    lab, n = P.connected_components(torch.from_numpy(mask.astype(np.uint8)).cuda())
    lab = lab.cpu().numpy()
    ref, nref = ndimage.label(mask, structure=np.ones((3, 3)))
    assert n == nref
    assert ((lab == 0) == (ref == 0)).all()
    m = np.zeros(n + 1, np.int64)
    m[lab.ravel()] = ref.ravel()                                                   # last writer: any pixel of the component
    assert np.array_equal(m[lab], ref) and len(np.unique(m)) == n + 1
    firsts = np.full(n + 1, lab.size, np.int64)
    np.minimum.at(firsts, lab.ravel(), np.arange(lab.size))
    assert (np.diff(firsts[1:]) > 0).all()


@pytest.mark.parametrize('seed,n,k', [(0, 500, 3), (1, 4000, 7), (2, 64, 2), (3, 20000, 12)])
def test_kmeans_exact(P, seed, n, k):
    rng = np.random.default_rng(seed)
    pts = rng.integers(0, 600, (n, 2))
    pts = np.unique(pts, axis=0)
    pts = pts[np.lexsort((pts[:, 0], pts[:, 1]))].astype(np.int32)
    c, l = P.kmeans(torch.from_numpy(pts).cuda(), k)
    rc, rl = PO.kmeans(pts, k)
    assert np.array_equal(l.cpu().numpy(), rl)
    assert np.array_equal(c.cpu().numpy(), rc)


def test_key_points_exact(P):
    gt = blobs(5)
    n, cnt, out, fgi = P.get_key_points(torch.from_numpy(gt).cuda(), 4, 3)
    rn, rcnt, rout, rfgi = PO.get_key_points(gt, 4, 3)
    assert n == rn and np.array_equal(cnt, rcnt) and np.array_equal(out.cpu().numpy(), rout)
    assert all(np.array_equal(a, b) for a, b in zip(fgi, rfgi))
    assert P.get_key_points(torch.zeros((32, 32), dtype=torch.uint8).cuda(), 4, 3) == (None, None, None, None)
    from utils import regiontools
    n2, cnt2, out2, _ = regiontools.get_key_points(gt, 4, 3, 3)
    assert n2 == rn and np.array_equal(cnt2, rcnt) and np.array_equal(out2, rout)


@pytest.mark.parametrize('seed,hw,nb', [(5, (96, 128), 9), (6, (160, 200), 14), (8, (240, 240), 4)])
def test_scannet_candidates_exact(P, seed, hw, nb):
    gt = blobs(seed, hw, nb)
    tissue = blobs(seed + 100, hw, 30)
    got = P.scannet_candidates(torch.from_numpy(gt).cuda(), torch.from_numpy(tissue).cuda())
    ref = PO.scannet_candidates(gt, tissue)
    assert list(got) == list(ref)
    for key in ref:
        for f in ('cnt_xy', 'perim_xy'):
            assert np.array_equal(got[key][f], ref[key][f]), (key, f)
        assert all(np.array_equal(a, b) for a, b in zip(got[key]['foreground_indices'], ref[key]['foreground_indices']))
        assert got[key]['tile_id'] == ref[key]['tile_id'] and got[key]['scan_level'] == 2


@pytest.mark.parametrize('iw,ih,ph,pw,sh,sw,m,seed', [(904, 1112, 256, 256, 64, 64, 1.0, 0), (2000, 1500, 256, 256, 256, 256, 0.25, 1),
                                                       (700, 650, 128, 128, 96, 96, 0.0625, 2), (300, 300, 256, 256, 32, 32, 1.0, 3),
                                                       (250, 250, 256, 256, 32, 32, 1.0, 4), (5000, 4000, 256, 256, 32, 32, 0.25, 5)])
def test_tile_grid_device_equals_oracle(iw, ih, ph, pw, sh, sw, m, seed):
    """wsi_tile_grid == the reference loops (oracle.wsi_oracle.tile_grid) tile for tile, in order."""
    from oracle import wsi_oracle as WO
    from wsi_segmentation_pipeline_amd import slide as S
    rng = np.random.default_rng(seed)
    mh, mw = max(1, int(ih * m)), max(1, int(iw * m))
    coarse = rng.random((mh // 8 + 1, mw // 8 + 1)) < 0.5
    mask = (np.kron(coarse, np.ones((8, 8), bool))[:mh, :mw] & (rng.random((mh, mw)) < 0.12)).astype(np.uint8)
    for mk in (None, mask):
        got = S.tile_grid_device(iw, ih, ph, pw, sh, sw, None if mk is None else torch.from_numpy(mk).cuda(), m, device='cuda:0').cpu().numpy()
        want = np.array(WO.tile_grid(iw, ih, ph, pw, sh, sw, mk, m), np.int32).reshape(-1, 2)
        assert got.shape == want.shape and np.array_equal(got, want)
        assert np.array_equal(got, S.tile_grid(iw, ih, ph, pw, sh, sw, mk, m))


# ------------------------------------------------------------------------------ SLIC (slic.py:43-75)
@pytest.mark.parametrize('seed,hw,nseg,sigma', [(0, (180, 240), 200, 5.0), (1, (97, 131), 60, 3.0), (2, (128, 128), 100, 0.0),
                                               (3, (300, 210), 200, 5.0)])
def test_slic_labels_exact(P, seed, hw, nseg, sigma):
    """wsi_slic == oracle slic_labels on seeded thumbnails: identical labels (the Gaussian pass is scipy's summation order, the
    cluster sums are exact integers, ties go to the lower centre on both sides)."""
    from tests.test_proposals_oracle import _thumb
    img = _thumb(seed, hw)
    got = P.slic(torch.from_numpy(img).cuda(), nseg, 20.0, sigma).cpu().numpy()
    ref = PO.slic_labels(img, nseg, 20.0, sigma)
    nd = int((got != ref).sum())
    print('slic %s: %d superpixels, %d / %d pixels differ' % (hw, len(np.unique(ref)), nd, ref.size))
    assert nd == 0


def test_slic_candidates_metadata_exact(P):
    from tests.test_proposals_oracle import _thumb
    img = _thumb(7, (120, 160))
    labels, meta = P.slic_candidates(torch.from_numpy(img).cuda(), (480, 640), 80, 20.0, 5.0)
    rl, rm = PO.slic_candidates(img, (480, 640), 80, 20.0, 5.0)
    assert np.array_equal(labels.cpu().numpy(), rl) and sorted(meta) == sorted(rm) and len(meta) > 20
    for k in meta:
        assert np.array_equal(meta[k]['cnt_xy'], rm[k]['cnt_xy']) and np.array_equal(meta[k]['perim_xy'], rm[k]['perim_xy'])
        assert all(np.array_equal(a, b) for a, b in zip(meta[k]['foreground_indices'], rm[k]['foreground_indices']))


# ------------------------------------------------------------------------------ find_nuclei, the non-default modes
def test_find_nuclei_lab_and_fill_mask_exact(P):
    """preprocessing.find_nuclei(mode='lab') and fill_mask=True (/root/reference/utils/preprocessing.py:88-92,101-106) on the
    device against the oracle (rgb2lab restated, SciPy's binary_fill_holes itself, the cv2 close restated)."""
    from tests.test_proposals_oracle import _thumb
    from oracle import wsi_oracle as WO
    from utils import preprocessing
    for seed, hw in ((0, (150, 201)), (3, (97, 64))):
        img = _thumb(seed, hw)
        img[10:40, 20:60] = (180, 60, 150)                                          # purple block: high a
        g = torch.from_numpy(img).cuda()
        for mu in (0.1, 0.5):
            assert np.array_equal(P.find_nuclei_lab(g, mu).cpu().numpy(), WO.find_nuclei_lab(img, mu))
        assert np.array_equal(preprocessing.find_nuclei(g, mode='lab').cpu().numpy(), WO.find_nuclei_lab(img))
    rng = np.random.default_rng(5)
    for hw in ((120, 160), (64, 64), (33, 130)):
        m = (rng.random(hw) < 0.08).astype(np.uint8)
        yy, xx = np.mgrid[:hw[0], :hw[1]]
        ring = (np.abs(np.hypot(yy - hw[0] / 2, xx - hw[1] / 2) - min(hw) / 3) < 2.5).astype(np.uint8)   # a closed ring: its inside is a hole
        m |= ring
        m[:, 0] = 0
        got = P.fill_mask(torch.from_numpy(m).cuda()).cpu().numpy()
        ref = WO.fill_mask(m)
        assert np.array_equal(got, ref) and ref.sum() > m.sum() + 20
    hsv_filled = preprocessing.find_nuclei(torch.from_numpy(_thumb(2, (90, 120))).cuda(), fill_mask=True).cpu().numpy()
    assert np.array_equal(hsv_filled, WO.fill_mask(WO.find_nuclei_hsv(_thumb(2, (90, 120)))))
