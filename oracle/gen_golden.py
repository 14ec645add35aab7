#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (in this container only).

Run from the repo root:   PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Imports, read-only and without copying anything into the repo:
  /root/reference/resnets_shift.py   (with a constants-only stand-in for utils.dataset_hr: the only
                                      attributes it reads are HR_NUM_CNT_SAMPLES / HR_NUM_PERIM_SAMPLES = 8)
  /root/reference/models/models.py   (torch only)
  /root/reference/contour_ordering.py
Weights come from oracle/weights.py (seeded; loaded with load_state_dict), inputs are seeded u8
patches pushed through the reference transform arithmetic (ToTensor + Normalize in fp32).
Only seeds and OUTPUTS are written - fixtures are data, never reference source.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import weights as W                      # noqa: E402
from oracle.resnet_oracle import normalize_u8        # noqa: E402  (transform arithmetic only)

REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    saved = {k: sys.modules.get(k) for k in ('utils', 'utils.dataset_hr')}
    pkg = types.ModuleType('utils')
    pkg.__path__ = []
    const = types.ModuleType('utils.dataset_hr')
    const.HR_NUM_CNT_SAMPLES = 8
    const.HR_NUM_PERIM_SAMPLES = 8
    pkg.dataset_hr = const
    sys.modules['utils'] = pkg
    sys.modules['utils.dataset_hr'] = const
    try:
        rs = _load('ref_resnets_shift', os.path.join(REF, 'resnets_shift.py'))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    mm = _load('ref_models', os.path.join(REF, 'models', 'models.py'))
    co = _load('ref_contour_ordering', os.path.join(REF, 'contour_ordering.py'))
    return rs, mm, co


TAP_NAMES = ['stem', 'pool'] + ['layer%d.%d' % (l, b) for l in (1, 2, 3, 4) for b in (0, 1)]


def run_with_taps(net, xs):
    """Forward the reference model, capturing stage outputs of the first patch iteration (p=0)."""
    taps = {}
    mods = {'stem': net.relu, 'pool': net.maxpool}
    for l in (1, 2, 3, 4):
        for b in (0, 1):
            mods['layer%d.%d' % (l, b)] = getattr(net, 'layer%d' % l)[b]
    hooks = []
    for name, mod in mods.items():
        def hook(_m, _i, out, name=name):
            if name not in taps:
                taps[name] = out.detach().clone()
        hooks.append(mod.register_forward_hook(hook))
    with torch.no_grad():
        singles, ens = net(xs)
    for h in hooks:
        h.remove()
    return singles, ens, taps


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    rs, mm, co = load_reference()
    net = rs.resnet18(False)
    sd = W.make_resnet18_state_dict(11)
    net.load_state_dict(sd)
    net.eval()

    # ---- bag forward, 64x64 patches (cfg4 shape) and 256x256 patches (cfg1 shape) -------------
    for tag, in_seed, shape, cs, ss in (('bag64', 12, (2, 16, 3, 64, 64), 4, 1),
                                        ('cfg1_256', 1, (4, 16, 3, 256, 256), 8, 4)):
        u8 = W.make_u8_patches(in_seed, shape)
        xs = normalize_u8(u8.reshape(-1, *shape[2:])).view(*shape)
        singles, ens, taps = run_with_taps(net, xs)
        rec = dict(weight_seed=11, input_seed=in_seed, input_shape=np.array(shape),
                   singles=singles.numpy(), ensemble=ens.numpy(), tap_cstride=cs, tap_sstride=ss)
        for name in TAP_NAMES:               # image (b=0, p=0), subsampled to keep fixtures small
            rec['tap_' + name.replace('.', '_')] = taps[name][0, ::cs, ::ss, ::ss].numpy()
        np.savez_compressed(os.path.join(OUT, 'resnet18_%s.npz' % tag), **rec)
        print(tag, 'singles', singles.shape, float(singles.abs().max()), 'ens', ens.shape)

    # ---- Classifier / Regressor heads --------------------------------------------------------
    rng = np.random.Generator(np.random.PCG64(21))
    fmap = torch.from_numpy(rng.standard_normal((5, 512, 8, 8), dtype=np.float32)).abs_()
    cls = mm.Classifier(512, 4)
    cls.load_state_dict(W.make_head_state_dict(22, 'classifier'))
    reg = mm.Regressor(512, 1)
    reg.load_state_dict(W.make_head_state_dict(23, 'regressor', num_classes=1))
    with torch.no_grad():
        np.savez_compressed(os.path.join(OUT, 'heads.npz'), fmap_seed=21, cls_seed=22, reg_seed=23,
                            classifier=cls.eval()(fmap).numpy(), regressor=reg.eval()(fmap).numpy())

    # ---- sliding-window 'cls' per-tile compute: reference trunk modules + reference Classifier
    #      (utils/eval.py:196-198 with the first-party backbone as encoder) --
    u8 = W.make_u8_patches(31, (8, 3, 256, 256))
    x = normalize_u8(u8)
    with torch.no_grad():
        f = net.maxpool(net.relu(net.bn1(net.conv1(x))))
        f = net.layer4(net.layer3(net.layer2(net.layer1(f))))
        logits = cls(f)
    np.savez_compressed(os.path.join(OUT, 'tile_logits_256.npz'), weight_seed=11, cls_seed=22, input_seed=31,
                        input_shape=np.array(u8.shape), logits=logits.numpy(),
                        fmap_sub=f[:, ::16].numpy())
    print('tile logits', logits.shape, float(logits.abs().max()))

    # ---- precision-margin families (r02): checkpoints / inputs far from the default fixture family, 256x256 tiles through
    #      reference trunk + reference Classifier.  'wide*': per-channel conv scales x0.1..x3.2, BN gamma in [0.25, 3], beta
    #      N(0, 0.5^2), BN running statistics CALIBRATED by the reference model itself (train-mode forward with momentum 1 on
    #      a seeded calibration batch) and stored in the fixture; 'hot': classifier scaled so |logit| >= 10; '*_he': H&E-like
    #      input with saturated 255 background.
    def tile_logits_ref(net_, cls_, u8_):
        with torch.no_grad():
            f_ = net_.maxpool(net_.relu(net_.bn1(net_.conv1(normalize_u8(u8_)))))
            return cls_(net_.layer4(net_.layer3(net_.layer2(net_.layer1(f_)))))

    fam = {}
    for tag, wseed, cal_seed, he_calib in (('wide_a', 101, 102, False), ('wide_b', 111, 112, True)):
        netw = rs.resnet18(False)
        sdw = W.make_wide_resnet18_state_dict(wseed)
        full = dict(netw.state_dict())
        full.update(sdw)
        netw.load_state_dict(full)
        for mmod in netw.modules():
            if isinstance(mmod, torch.nn.BatchNorm2d):
                mmod.momentum = 1.0                                   # running stats := batch stats of the calibration batch
        netw.train()
        cal = W.make_he_patches(cal_seed, 12) if he_calib else W.make_u8_patches(cal_seed, (12, 3, 256, 256))
        with torch.no_grad():
            xcal = normalize_u8(cal)
            fcal = netw.maxpool(netw.relu(netw.bn1(netw.conv1(xcal))))
            netw.layer4(netw.layer3(netw.layer2(netw.layer1(fcal))))
        netw.eval()
        stats = {k: v.numpy().copy() for k, v in netw.state_dict().items() if k.endswith('running_mean') or k.endswith('running_var')}
        vmin = min(float(v.min()) for k, v in stats.items() if k.endswith('var'))
        vmax = max(float(v.max()) for k, v in stats.items() if k.endswith('var'))
        fam[tag] = (netw, wseed, stats)
        print(tag, 'calibrated BN running_var range [%.3g, %.3g]' % (vmin, vmax))
    cls_hot = mm.Classifier(512, 4)
    hot_sd = W.make_head_state_dict(24, 'classifier')
    cases = []
    # (case, net, weight family, head seed, head gain, input kind, input seed)
    for case, tagw, gain, kind, iseed in (('wide_a_uniform', 'wide_a', 1.0, 'uniform', 131), ('wide_a_he', 'wide_a', 1.0, 'he', 132),
                                          ('wide_b_he_hot', 'wide_b', None, 'he', 133), ('default_he', None, 1.0, 'he', 134),
                                          ('default_uniform_hot', None, None, 'uniform', 135)):
        netc = fam[tagw][0] if tagw else net
        u8c = W.make_he_patches(iseed, 8) if kind == 'he' else W.make_u8_patches(iseed, (8, 3, 256, 256))
        head = mm.Classifier(512, 4)
        hsd = {k: v.clone() for k, v in hot_sd.items()}
        if gain is None:                                               # scale the head until the largest |logit| is 16
            head.load_state_dict(hsd)
            with torch.no_grad():
                g0 = float(tile_logits_ref(netc, head.eval(), u8c).abs().max())
            gain = 16.0 / g0
        hsd['fc.0.weight'] *= gain
        hsd['fc.0.bias'] *= gain
        head.load_state_dict(hsd)
        lg = tile_logits_ref(netc, head.eval(), u8c).numpy()
        rec = dict(weight_family=tagw or 'default', weight_seed=fam[tagw][1] if tagw else 11, head_seed=24, head_gain=np.float64(gain),
                   input_kind=kind, input_seed=iseed, logits=lg)
        if tagw:
            rec.update({'bn__' + k.replace('.', '__'): v for k, v in fam[tagw][2].items()})
        np.savez_compressed(os.path.join(OUT, 'margin_%s.npz' % case), **rec)
        print('margin case', case, 'max |logit| %.2f' % float(np.abs(lg).max()))

    # ---- contour_ordering.evenly_spaced_points_on_a_contour ------------------------------------
    t = np.linspace(0, 2 * np.pi, 97)
    contour = np.stack((40 + 30 * np.cos(t) + 3 * np.sin(5 * t), 35 + 20 * np.sin(t)), 1)
    sq = np.array([[0, 0], [10, 0], [10, 10], [0, 10], [0, 0]], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'esp.npz'), contour=contour, esp16=co.evenly_spaced_points_on_a_contour(contour, 16),
                        esp8=co.evenly_spaced_points_on_a_contour(contour, 8), square=sq,
                        esp_sq9=co.evenly_spaced_points_on_a_contour(sq, 9))
    print('golden written to', OUT)


if __name__ == '__main__':
    main()
