"""Slide-side integer rows pinned by running the REFERENCE itself (r04): tests/golden/{threshold_probs,isforeground,map_points}.npz
were written by oracle/gen_golden.py from /root/reference/utils/preprocessing.py:60-71,156-172 and utils/regiontools.py:15-37,
imported in the build container behind an import-only stub finder for the absent packages (nothing from a stub executes).  Here:
the CPU oracle and the host-side drop-in modules against those fixtures (bit-exact for classes / booleans / points); the GPU
kernels are held to the same fixtures in tests/test_gpu_reference_pins.py.  kmeans_sklearn.npz holds what sklearn's
MiniBatchKMeans returns when called as the reference calls it: the repo's k-means is another algorithm by design, and
test_kmeans_* states the distance between the two."""
import os

import numpy as np
import pytest

from oracle import proposals_oracle as PO
from oracle import wsi_oracle as WO


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_threshold_probs_oracle_equals_reference(golden_dir):
    z = _load(golden_dir, 'threshold_probs.npz')
    for case in z['cases']:
        cls, probs = WO.threshold_probs(z[case + '_pred'], tuple(z[case + '_class_probs']))
        assert cls.dtype == np.uint8 and np.array_equal(cls, z[case + '_classes']), case
        assert np.array_equal(probs, z[case + '_probs']), case              # same torch-CPU float64 softmax: bit-equal
    # an all-zero map (pixels no tile touched) is class 0 with uniform probabilities
    assert not z['zeros_classes'].any() and np.all(z['zeros_probs'] == 0.25)


def test_isforeground_oracle_and_dropin_equal_reference(golden_dir):
    from utils import preprocessing as drop_in
    z = _load(golden_dir, 'isforeground.npz')
    for case in z['cases']:
        want = z[case + '_out']
        for f in (WO.isforeground, drop_in.isforeground):
            got = [bool(f(z[case])), bool(f(z[case], 0.9)), bool(f(z[case], 0.0))]
            assert got == want.tolist(), (case, f.__module__)
    assert z['edge_1_of_20_out'][0] and not z['below_1_of_21_out'][0]        # 0.05 exactly is foreground (>=)


def test_map_points_oracle_and_dropin_equal_reference(golden_dir):
    from utils import regiontools as drop_in

    class P:
        pass
    z = _load(golden_dir, 'map_points.npz')
    for case in z['cases']:
        scan_level, tw, th, iw, ih = (int(v) for v in z[case + '_params'])
        want = z[case + '_out']
        got = WO.map_points(z[case + '_in'], scan_level, tw, th, iw, ih)
        got = got[0] if isinstance(got, tuple) else got
        assert np.array_equal(np.asarray(got).reshape(-1, 2), want.reshape(-1, 2)), case
        p = P()
        p.scan_level, p.tile_w, p.tile_h, p.iw, p.ih = scan_level, tw, th, iw, ih
        pts, n = drop_in.map_points(z[case + '_in'], p)
        assert n == len(want) and np.array_equal(pts.reshape(-1, 2), want.reshape(-1, 2)), case


def matched_centre_distance(a, b):
    """Largest distance between matched centres under the best one-to-one matching (Hungarian), in pixels."""
    from scipy.optimize import linear_sum_assignment
    d = np.sqrt(((a[:, None, :] - b[None, :, :]) ** 2).sum(-1))
    r, c = linear_sum_assignment(d)
    return float(d[r, c].max()), float(d[r, c].mean())


def inertia(coords, centres):
    d = ((coords[:, None, :].astype(np.float64) - centres[None]) ** 2).sum(-1)
    return float(d.min(1).sum())


# Measured in the build container (sklearn 1.7.2; `pytest -s` prints the table).  r04 first quantified the r01-r03 rule (ONE
# raster-stratified seeding): matched-centre distance max 16.2 / 0.2 / 16.3 / 14.0 / 37.5 px, objective 1.626 / 0.998 / 0.916 / 0.990 /
# 1.343 x sklearn's.  With the best of two seedings (stratified and farthest-point, r04):
#   case            points  k   matched-centre distance max / mean (px)   inertia ours / sklearn
#   seed5_us4_k3      120    3         0.2 /  0.1                              0.999
#   seed0_us4_k3      116    3         0.2 /  0.1                              0.998
#   seed1_us2_k5      651    5        16.3 /  7.2                              0.916
#   seed2_us4_k8      118    8         0.8 /  0.2                              0.974
#   seed7_us2_k12     620   12        39.2 /  4.3                              1.096
# i.e. three regions now agree with sklearn to under a pixel; in the other two the answers are DIFFERENT local minima of the same
# objective (ours the better one in one of them): centre points, and with them the 64 x 64 crops of cfg4, are still not guaranteed
# to be the reference's.  A full-batch Lloyd iteration started from sklearn's centres moves them by <= 0.8 px - the objective and
# the update rule agree, the seeds do not.
KMEANS_MAX_PX = 40.0            # bound on the matched-centre distance (the regions are 24 x 32 .. 48 x 64 down-sampled pixels)
KMEANS_MAX_INERTIA_RATIO = 1.10


def test_kmeans_deviation_from_sklearn_is_bounded(golden_dir):
    """/root/reference/utils/regiontools.py:89 is sklearn's MiniBatchKMeans(random_state=0): RNG- and version-dependent (k-means++
    seeding from numpy's global-style RandomState, mini-batch sampling, low-count reassignment), so the repo ships a deterministic
    Lloyd iteration (oracle/proposals_oracle.py kmeans == wsi_kmeans_points on the device).  This test QUANTIFIES the difference
    on the seeded regions of tests/test_gpu_proposals.py: matched-centre distance and the clustering objective of both."""
    z = _load(golden_dir, 'kmeans_sklearn.npz')
    for case in z['cases']:
        coords, ref = z[case + '_coords'], z[case + '_centers']
        k = int(z[case + '_args'][2])
        mine, labels = PO.kmeans(coords, k)
        dmax, dmean = matched_centre_distance(mine, ref)
        ratio = inertia(coords, mine) / inertia(coords, ref)
        print('%-16s n=%5d k=%2d  matched-centre distance max %.2f mean %.2f px   inertia ours / sklearn = %.4f' %
              (case, len(coords), k, dmax, dmean, ratio))
        assert dmax <= KMEANS_MAX_PX and ratio <= KMEANS_MAX_INERTIA_RATIO, (case, dmax, ratio)
        # what does agree: sklearn's solution is (nearly) a fixed point of the repo's update rule
        centres = ref.copy()
        d = ((coords[:, None, :].astype(np.float64) - centres[None]) ** 2).sum(-1)
        lab = d.argmin(1)
        moved = max(float(np.sqrt(((coords[lab == j].mean(0) - centres[j]) ** 2).sum())) for j in range(k) if (lab == j).any())
        assert moved <= 2.5, (case, moved)
