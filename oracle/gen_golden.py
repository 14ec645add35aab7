#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (in this container only).

Run from the repo root:   PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Imports, read-only and without copying anything into the repo:
  /root/reference/resnets_shift.py   (with a constants-only stand-in for utils.dataset_hr: the only
                                      attributes it reads are HR_NUM_CNT_SAMPLES / HR_NUM_PERIM_SAMPLES = 8)
  /root/reference/models/models.py   (torch only)
  /root/reference/contour_ordering.py
  /root/reference/utils/preprocessing.py, /root/reference/utils/regiontools.py (r04) for the three slide-side integer
      functions that execute no absent package - threshold_probs (:156-172), isforeground (:60-71), map_points
      (regiontools.py:15-37).  Their modules IMPORT skimage / cv2 / mahotas / openslide / torchvision / ..., none of which is
      installed: an import-only stub finder (`_AbsentPackages`) satisfies those imports with empty modules whose every
      attribute raises when called, so nothing from a stub can execute; the two NumPy aliases the reference still spells
      (`np.int`) exist only while its code runs.
  sklearn 1.7.2's MiniBatchKMeans, called exactly as /root/reference/utils/regiontools.py:89 calls it, on the seeded regions of
      tests/test_gpu_proposals.py (the repo's own k-means is a deterministic Lloyd spec: the fixture quantifies the distance).
Weights come from oracle/weights.py (seeded; loaded with load_state_dict), inputs are seeded u8
patches pushed through the reference transform arithmetic (ToTensor + Normalize in fp32).
Only seeds, inputs and OUTPUTS are written - fixtures are data, never reference source.
"""
import importlib.abc
import importlib.machinery
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


# ---------------------------------------------------------------------------------- r04: the reference's utils package
ABSENT = ('skimage', 'cv2', 'torchvision', 'mahotas', 'openslide', 'segmentation_models_pytorch', 'pretrainedmodels', 'adabound',
          'concave_hull', 'shapely')


class _StubMeta(type):
    """Whatever a reference module pulls out of an absent package at import time is a CLASS that can be named, sub-classed
    (utils/preprocessing.py:35 derives from transforms.Normalize) and never instantiated or called."""
    def __getattr__(cls, item):
        if item.startswith('__'):
            raise AttributeError(item)
        return _StubMeta(cls.__name__ + '.' + item, (_StubBase,), {})


class _StubBase(metaclass=_StubMeta):
    def __new__(cls, *a, **k):
        raise RuntimeError('stub of an absent package executed: %s' % cls.__name__)


class _StubModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith('__'):
            raise AttributeError(item)
        return _StubMeta(self.__name__ + '.' + item, (_StubBase,), {})


class _AbsentPackages(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Import-only stand-ins for packages that are NOT installed (checked: a real package always wins, the finder sits last)."""
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split('.')[0] in ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class reference_utils:
    """Context: the reference's own `utils.preprocessing` / `utils.regiontools` / `myargs`, imported from /root/reference with
    the repo's same-named drop-in modules moved out of the way, absent third-party packages stubbed (import-only) and the removed
    NumPy aliases present.  Everything is restored on exit."""
    SHADOWED = ('utils', 'myargs', 'models', 'contour_ordering', 'resnets_shift')

    def __enter__(self):
        for top in ABSENT:                                   # a stub must never shadow something real
            if importlib.util.find_spec(top) is not None:
                raise RuntimeError('%s is installed: remove it from ABSENT' % top)
        self.saved_modules = {k: v for k, v in sys.modules.items() if k.split('.')[0] in self.SHADOWED}
        for k in self.saved_modules:
            del sys.modules[k]
        self.saved_path, self.saved_argv = list(sys.path), list(sys.argv)
        sys.path[:] = [REF] + [p for p in sys.path if os.path.abspath(p or '.') != ROOT]
        sys.argv[:] = ['reference']                          # myargs parses the command line
        self.finder = _AbsentPackages()
        sys.meta_path.append(self.finder)
        self.had_int = hasattr(np, 'int')
        if not self.had_int:
            np.int = int                                     # removed in NumPy 1.24; regiontools.py:24 still spells it
        import utils.preprocessing as prep                   # noqa: E402  (the reference's)
        import utils.regiontools as rt                       # noqa: E402
        import myargs as rargs                               # noqa: E402
        assert os.path.abspath(prep.__file__).startswith(REF) and os.path.abspath(rt.__file__).startswith(REF)
        return prep, rt, rargs.args

    def __exit__(self, *exc):
        sys.meta_path.remove(self.finder)
        if not self.had_int:
            del np.int
        for k in [k for k in sys.modules if k.split('.')[0] in self.SHADOWED or k.split('.')[0] in ABSENT]:
            del sys.modules[k]
        sys.modules.update(self.saved_modules)
        sys.path[:], sys.argv[:] = self.saved_path, self.saved_argv
        return False


class _Params:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def blob_mask(seed, hw=(96, 128), n=9):
    """The seeded region masks of tests/test_gpu_proposals.py / tests/test_proposals_oracle.py (`blobs`)."""
    rng = np.random.default_rng(seed)
    img = np.zeros(hw, np.uint8)
    yy, xx = np.mgrid[:hw[0], :hw[1]]
    for _ in range(n):
        cy, cx = rng.integers(5, hw[0] - 5), rng.integers(5, hw[1] - 5)
        ry, rx = rng.integers(3, 14), rng.integers(3, 18)
        img[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 1
    return img


def slide_side_fixtures():
    """threshold_probs / isforeground / map_points run by the reference itself; sklearn's k-means as the reference calls it."""
    with reference_utils() as (prep, rt, rargs):
        # ---- threshold_probs (utils/preprocessing.py:156-172): summed-logit maps (float64, as predict_tumorbed accumulates them)
        rec = {}
        cases = [('zeros', (0., 0., 0., 0.)), ('defaults_scaled', (0., 0., 0., 0.)), ('mild', (0.1, 0.2, 0.3, 0.25)),
                 ('strict', (0.5, 0.5, 0.5, 0.5)), ('edge', (0.25, 0.25, 0.25, 0.25)), ('one_class', (0.0, 0.99, 0.0, 0.0))]
        rng = np.random.Generator(np.random.PCG64(41))
        for i, (name, probs) in enumerate(cases):
            pred = rng.standard_normal((4, 23, 31)) * (0.5, 4.0, 1.0, 1.0, 1.0, 2.0)[i]
            if name == 'zeros':
                pred[:] = 0.0                                 # untouched map pixels: uniform softmax, argmax 0
            if name == 'edge':
                pred[:, :5] = 0.0                             # p == class_probs exactly: kept (the test is `<`)
                pred[:, 5:9] = pred[:1, 5:9]                  # ties between all classes
            pred = np.ascontiguousarray(pred, np.float64)
            rargs.num_classes, rargs.class_probs = 4, list(probs)
            cls, pr = prep.threshold_probs(pred.copy())
            rec[name + '_pred'], rec[name + '_class_probs'] = pred, np.array(probs, np.float64)
            rec[name + '_classes'], rec[name + '_probs'] = cls, pr
            assert cls.dtype == np.uint8 and pr.dtype == np.float64
        rec['cases'] = np.array([c[0] for c in cases])
        np.savez_compressed(os.path.join(OUT, 'threshold_probs.npz'), **rec)
        print('threshold_probs', len(cases), 'cases')

        # ---- isforeground (utils/preprocessing.py:60-71), threshold edge 0.05 exactly
        rec, names = {}, []
        rng = np.random.Generator(np.random.PCG64(42))
        arrs = {'edge_1_of_20': np.array([1] + [0] * 19, np.uint8), 'below_1_of_21': np.array([1] + [0] * 20, np.uint8),
                'all_zero': np.zeros((6, 7), np.uint8), 'all_one': np.ones((6, 7), np.uint8),
                'edge_2d_5_of_100': (np.arange(100).reshape(10, 10) % 20 == 0).astype(np.uint8),
                'below_2d_4_of_100': (np.arange(100).reshape(10, 10) % 25 == 0).astype(np.uint8),
                'values_255': (rng.random((16, 16)) < 0.06).astype(np.uint8) * 255,
                'float_mask': (rng.random((9, 11)) < 0.05).astype(np.float64),
                'window_160': (rng.random((160, 160)) < 0.0502).astype(np.uint8)}
        for k, a in arrs.items():
            names.append(k)
            rec[k] = a
            rec[k + '_out'] = np.array([bool(prep.isforeground(a)), bool(prep.isforeground(a, 0.9)), bool(prep.isforeground(a, 0.0))])
        rec['cases'] = np.array(names)
        np.savez_compressed(os.path.join(OUT, 'isforeground.npz'), **rec)
        print('isforeground', {k: rec[k + '_out'].tolist() for k in names})

        # ---- map_points (utils/regiontools.py:15-37): border cases on both axes, scan levels 0-2
        rec, names = {}, []
        rng = np.random.Generator(np.random.PCG64(43))
        for scan_level, tile, (iw, ih) in ((2, 64, (4000, 3000)), (1, 64, (1000, 800)), (0, 32, (300, 200)), (2, 64, (1040, 1040))):
            f = 4 ** scan_level
            pts = rng.integers(0, (iw // f + 2, ih // f + 2), (40, 2))
            h = tile // 2
            border = np.array([[h // f, h // f], [(h + f) // f, (h + f) // f], [0, 0], [iw // f, ih // f],
                               [(iw - tile + h) // f, 10], [(iw - tile + h) // f - 1, 10], [10, (ih - tile + h) // f],
                               [10, (ih - tile + h) // f - 1], [(h // f) + 1, 10], [10, (h // f) + 1]])
            arr = np.concatenate([border, pts]).astype(np.int64)
            params = _Params(scan_level=scan_level, tile_w=tile, tile_h=tile, iw=iw, ih=ih)
            out, n = rt.map_points(arr.copy(), params)
            name = 'L%d_%dx%d' % (scan_level, iw, ih)
            names.append(name)
            rec[name + '_in'], rec[name + '_out'] = arr, np.asarray(out)
            rec[name + '_params'] = np.array([scan_level, tile, tile, iw, ih])
            assert n == len(out)
        # float inputs are truncated by .astype(int) first (centre points come from k-means as floats in other call sites)
        arrf = np.array([[10.9, 20.2], [0.99, 5.0], [250.5, 180.7], [8.0, 8.0]])
        params = _Params(scan_level=1, tile_w=64, tile_h=64, iw=1000, ih=800)
        out, n = rt.map_points(arrf.copy(), params)
        names.append('float_in')
        rec['float_in_in'], rec['float_in_out'], rec['float_in_params'] = arrf, np.asarray(out), np.array([1, 64, 64, 1000, 800])
        rec['cases'] = np.array(names)
        np.savez_compressed(os.path.join(OUT, 'map_points.npz'), **rec)
        print('map_points', {k: (len(rec[k + '_in']), len(rec[k + '_out'])) for k in names})

    # ---- k-means as the reference runs it (utils/regiontools.py:89: `KMeans(n_clusters=num_clusters, random_state=0).fit(coords)`
    #      with KMeans = sklearn.cluster.MiniBatchKMeans), on the foreground coordinates the repo's get_key_points hands its own
    #      k-means for the seeded regions of tests/test_gpu_proposals.py (nearest-resized mask, (x, y) pairs in raster order)
    import sklearn
    from sklearn.cluster import MiniBatchKMeans as KMeans
    from oracle import proposals_oracle as PO
    rec, names = {'sklearn_version': np.array(sklearn.__version__)}, []
    for seed, us, k in ((5, 4, 3), (0, 4, 3), (1, 2, 5), (2, 4, 8), (7, 2, 12)):
        gt = blob_mask(seed)
        y, x = gt.shape
        small = PO.resize_nearest(gt, (y // us, x // us))
        coords = np.transpose(np.nonzero(small))[:, ::-1]
        km = KMeans(n_clusters=k, random_state=0).fit(coords)
        name = 'seed%d_us%d_k%d' % (seed, us, k)
        names.append(name)
        rec[name + '_coords'] = coords.astype(np.int32)
        rec[name + '_centers'] = km.cluster_centers_.astype(np.float64)
        rec[name + '_labels'] = km.labels_.astype(np.int32)
        rec[name + '_inertia'] = np.float64(km.inertia_)
        rec[name + '_args'] = np.array([seed, us, k])
    rec['cases'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'kmeans_sklearn.npz'), **rec)
    print('kmeans_sklearn', names, 'sklearn', sklearn.__version__)


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
    slide_side_fixtures()
    print('golden written to', OUT)


if __name__ == '__main__':
    main()
