"""The CPU oracle must reproduce the golden vectors produced by the reference itself
(oracle/gen_golden.py).  fp32 torch ops in the same order -> tolerance 1e-5."""
import os

import numpy as np
import torch

from oracle import resnet_oracle as R
from oracle import weights as W
from oracle import wsi_oracle as WO

TOL = 1e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _bag_case(golden_dir, name):
    g = _load(golden_dir, name)
    sd = W.make_resnet18_state_dict(int(g['weight_seed']))
    shape = tuple(int(v) for v in g['input_shape'])
    u8 = W.make_u8_patches(int(g['input_seed']), shape)
    xs = R.normalize_u8(u8.reshape(-1, *shape[2:])).view(*shape)
    return g, sd, xs


def test_bag_forward_64(golden_dir):
    g, sd, xs = _bag_case(golden_dir, 'resnet18_bag64.npz')
    with torch.no_grad():
        singles, ens = R.resnet_forward(sd, xs)
        taps = {}
        R.trunk(sd, xs[:, 0], taps)          # patch p=0 of every bag; image (b=0,p=0) is index 0
    assert np.abs(singles.numpy() - g['singles']).max() <= TOL
    assert np.abs(ens.numpy() - g['ensemble']).max() <= TOL
    cs, ss = int(g['tap_cstride']), int(g['tap_sstride'])
    for name, t in taps.items():
        ref = g['tap_' + name.replace('.', '_')]
        got = t[0, ::cs, ::ss, ::ss].numpy()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), name


def test_bag_forward_cfg1_256(golden_dir):
    torch.set_num_threads(8)
    g, sd, xs = _bag_case(golden_dir, 'resnet18_cfg1_256.npz')
    with torch.no_grad():
        singles, ens = R.resnet_forward(sd, xs)
    assert singles.shape == (64, 4) and ens.shape == (4, 4)
    assert np.abs(singles.numpy() - g['singles']).max() <= TOL
    assert np.abs(ens.numpy() - g['ensemble']).max() <= TOL


def test_heads(golden_dir):
    g = _load(golden_dir, 'heads.npz')
    rng = np.random.Generator(np.random.PCG64(int(g['fmap_seed'])))
    fmap = torch.from_numpy(rng.standard_normal((5, 512, 8, 8), dtype=np.float32)).abs_()
    c = R.classifier(W.make_head_state_dict(int(g['cls_seed']), 'classifier'), fmap)
    r = R.regressor(W.make_head_state_dict(int(g['reg_seed']), 'regressor', num_classes=1), fmap)
    assert np.abs(c.numpy() - g['classifier']).max() <= TOL
    assert np.abs(r.numpy() - g['regressor']).max() <= TOL


def test_tile_logits_256(golden_dir):
    g = _load(golden_dir, 'tile_logits_256.npz')
    sd = W.make_resnet18_state_dict(int(g['weight_seed']), with_fc=False)
    cls_sd = W.make_head_state_dict(int(g['cls_seed']), 'classifier')
    u8 = W.make_u8_patches(int(g['input_seed']), tuple(int(v) for v in g['input_shape']))
    with torch.no_grad():
        logits = R.tile_logits(sd, cls_sd, u8)
    assert np.abs(logits.numpy() - g['logits']).max() <= TOL


def test_esp(golden_dir):
    g = _load(golden_dir, 'esp.npz')
    assert np.abs(WO.evenly_spaced_points_on_a_contour(g['contour'], 16) - g['esp16']).max() <= 1e-12
    assert np.abs(WO.evenly_spaced_points_on_a_contour(g['contour'], 8) - g['esp8']).max() <= 1e-12
    assert np.abs(WO.evenly_spaced_points_on_a_contour(g['square'], 9) - g['esp_sq9']).max() <= 1e-12


def test_state_dict_keys_match_reference_count():
    keys = W.resnet18_key_shapes()
    assert len(keys) == 130
    n_params = sum(int(np.prod(s)) for k, s, kind in keys if kind != 'bn_n' and kind not in ('bn_m', 'bn_v'))
    assert n_params == 44762716
