#!/usr/bin/env python3
"""Where does a predict_tumorbed(mode='seg') call spend its time?  (r05: the `api` leg of `bench.py --workload seg`.)
Synchronised stopwatch around the stages of one 5888^2 slide (528 tiles of 256x256), three repetitions after a warm-up."""
import os, sys, tempfile, time, collections
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myargs
import utils.dataset as UD
import utils.eval as UE
from PIL import Image
from wsi_segmentation_pipeline_amd import slide as S, engine as E, synthetic as W
from wsi_segmentation_pipeline_amd.unet import UNetSeg, UNetEngine

dev = torch.device('cuda:0')
TILE, side = 256, 23
usd = W.make_unet_state_dict(5, classes=4)
for key in ('decoder.final_conv.weight', 'decoder.final_conv.bias'):
    usd[key] = usd[key] * (8.0 / 216.0)
model = UNetSeg(4, precision=sys.argv[1] if len(sys.argv) > 1 else 'parity')
model.load_state_dict(usd)
model = model.to(dev).eval()
model.hip_engine(dev).max_batch = 512
level0 = torch.randint(0, 256, (side * TILE, side * TILE, 3), dtype=torch.uint8, device=dev)
acc = collections.OrderedDict()
undo = []


def wrap(mod, name, label=None):
    f = getattr(mod, name)
    label = label or name

    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(mod, name, g)
    undo.append((mod, name, f))


wrap(UNetEngine, 'forward_tiles')
wrap(E, 'stitch_add_dense'); wrap(E, 'exponent_span'); wrap(E, 'softmax_threshold_argmax')
wrap(UE, '_to_host'); wrap(S, '_upload'); wrap(torch, 'zeros', 'torch.zeros (the float64 map)')
ma = myargs.args
with tempfile.TemporaryDirectory() as td:
    ma.scan_level, ma.scan_resize, ma.num_classes, ma.class_probs = 0, 1, 4, [0., 0., 0., 0.]
    ma.tile_w = ma.tile_h = ma.tile_stride_w = ma.tile_stride_h = TILE
    ma.wsi_mask_pth, ma.val_save_pth = td, os.path.join(td, 'out')
    Image.fromarray(np.ones((side * TILE, side * TILE), np.uint8)).save(os.path.join(td, 'bench.svs.png'))
    for rep in range(4):
        sl = S.ArraySlide([level0], [1.0]); sl.level_dimensions = ((side * TILE, side * TILE),); sl.name = 'bench.svs'
        t0 = time.perf_counter()
        dsw = UD.Dataset_wsis({'bench.svs': sl}, {'ph': TILE, 'pw': TILE, 'sh': TILE, 'sw': TILE}, bs=ma.batch_size)
        t_ds = time.perf_counter() - t0
        acc.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        UE.predict_tumorbed(model, dsw, 0, mode='seg', save=False)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        print('rep %d: dataset %.1f ms | call %.2f ms (synchronised stages: slower than the free-running call) | ' % (rep, t_ds * 1e3, t * 1e3)
              + ' | '.join('%s %.2f' % (k, v * 1e3) for k, v in acc.items()) + ' | rest %.2f' % ((t - sum(acc.values())) * 1e3))
    # host-side profile of one more (free-running) call: what the stopwatch above files under `rest`
    import cProfile, pstats
    sl = S.ArraySlide([level0], [1.0]); sl.level_dimensions = ((side * TILE, side * TILE),); sl.name = 'bench.svs'
    dsw = UD.Dataset_wsis({'bench.svs': sl}, {'ph': TILE, 'pw': TILE, 'sh': TILE, 'sw': TILE}, bs=ma.batch_size)
    for mod, name, f in undo:
        setattr(mod, name, f)
    # free-running API calls against the hand-written pipeline on the same engine
    eng = model.hip_engine(dev)
    xy_all = torch.tensor(np.asarray(dsw.wsis['bench.svs']['iterator'].dataset.tile_xy), dtype=torch.int32, device=dev)
    mask_dev = torch.ones((side * TILE, side * TILE), dtype=torch.uint8, device=dev)
    for rep in range(4):
        sl = S.ArraySlide([level0], [1.0]); sl.level_dimensions = ((side * TILE, side * TILE),); sl.name = 'bench.svs'
        d2 = UD.Dataset_wsis({'bench.svs': sl}, {'ph': TILE, 'pw': TILE, 'sh': TILE, 'sw': TILE}, bs=ma.batch_size)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        UE.predict_tumorbed(model, d2, 0, mode='seg', save=False)
        torch.cuda.synchronize(); t_api = time.perf_counter() - t0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pred_ = torch.zeros((4, side * TILE, side * TILE), dtype=torch.float64, device=dev)
        lg_ = eng.forward_tiles(level0, xy_all, TILE, TILE)
        E.stitch_add_dense(pred_, lg_, xy_all)
        E.exponent_span(lg_)
        c_, _, h_ = E.softmax_threshold_argmax(pred_, [0., 0., 0., 0.], mask_dev, 'seg', want_probs=False)
        UE._to_host(h_, c_)
        torch.cuda.synchronize(); t_pipe = time.perf_counter() - t0
        del pred_, lg_, c_, h_
        print('free-running rep %d: API call %.2f ms | hand-written pipeline %.2f ms (%d tiles)' % (rep, t_api * 1e3, t_pipe * 1e3, xy_all.shape[0]))
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    UE.predict_tumorbed(model, dsw, 0, mode='seg', save=False)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(12)
    pstats.Stats(pr).sort_stats('cumtime').print_stats(30)
