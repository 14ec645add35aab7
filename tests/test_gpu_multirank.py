"""Two ranks (gloo rendezvous, both on cuda:0 - the test box has one GPU) run the sharded sliding-
window pipeline; every rank must end with exactly the single-rank result.  The N>1 data path is
shard -> per-rank HIP trunk -> ONE all-gather -> stitch; on the 8-GPU node the same
code runs with backend 'nccl' (RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pipeline(rank, world, resident=False):
    from wsi_segmentation_pipeline_amd import slide as S, synthetic as W
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    dev = torch.device('cuda:0')
    sd = W.make_resnet18_state_dict(11, with_fc=False)
    cls = W.make_head_state_dict(22, 'classifier')
    src = S.SyntheticRows(790, 530, 5, dev, block=64)
    tiles = S.tile_grid(790, 530, 64, 64, 48, 48)                       # overlapping tiles: 159 of them
    eng = TrunkEngine(sd, dev, head=(cls['fc.0.weight'], cls['fc.0.bias']), max_batch=64)
    if resident:                                                        # bench.py's cfg3 path: only this rank's slide regions exist
        lo, hi = S.shard_range(len(tiles), rank, world)
        rects, hw, local_xy = S.region_plan(tiles[lo:hi], 64, 64)
        level0 = S.resident_regions(src, rects, hw, dev)
        assert level0.numel() < 0.8 * 790 * 530 * 3 or world == 1
    else:
        level0, local_xy = src.full(), None
    out = S.infer_slide_cls(eng, level0, tiles, 64, 64, 0.25, (132, 197), 4, (0., 0., 0., 0.), None, rank, world,
                            local_xy=local_xy)
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in out.items() if v is not None}


def _worker(rank, world, port, q, resident):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    res = _pipeline(rank, world, resident)
    q.put((rank, {k: v.numpy() for k, v in res.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('resident', [False, True])
def test_two_rank_pipeline_equals_single_rank(resident):
    """resident=True: every rank holds only the slide regions of its own tiles (the band-resident path of bench.py
    cfg3) and must still reproduce the single-rank, whole-slide result bit for bit."""
    ref = _pipeline(0, 1)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, resident)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
    for rank in (0, 1):
        for k in ('logits', 'pred', 'classes', 'heatmap', 'probs'):
            assert np.array_equal(got[rank][k], ref[k].numpy()), (rank, k)


def _few_tiles(rank, world, ntiles):
    """Fewer tiles than ranks (rank 1 owns nothing) or no foreground tile at all, precision='auto' (whose probe is a collective)."""
    from wsi_segmentation_pipeline_amd import slide as S, synthetic as W
    from wsi_segmentation_pipeline_amd.engine import AutoTrunkEngine
    dev = torch.device('cuda:0')
    sd = W.make_resnet18_state_dict(11, with_fc=False)
    cls = W.make_head_state_dict(22, 'classifier')
    src = S.SyntheticRows(790, 530, 5, dev, block=64)
    tiles = S.tile_grid(790, 530, 64, 64, 48, 48)[37:37 + ntiles]
    eng = AutoTrunkEngine(sd, dev, head=(cls['fc.0.weight'], cls['fc.0.bias']), max_batch=64)
    out = S.infer_slide_cls(eng, src.full(), tiles, 64, 64, 0.25, (132, 197), 4, (0., 0., 0., 0.), None, rank, world)
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items() if torch.is_tensor(v)}
    res['mode'] = out['precision']['mode']
    return res


def _few_worker(rank, world, port, q, ntiles):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    q.put((rank, _few_tiles(rank, world, ntiles)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('ntiles', [1, 0])
def test_two_ranks_with_fewer_tiles_than_ranks(ntiles):
    """One tile over two ranks (rank 1's shard is empty: no trunk call, but it still joins the probe reduction and the gather)
    and a slide without a single foreground tile (the reference drops such slides, dataset.py:198-201; here the map is all
    zeros: softmax 1/C everywhere): both ranks return the single-rank result."""
    ref = _few_tiles(0, 1, ntiles)
    assert ref['logits'].shape == (ntiles, 4) and ref['heatmap'].shape == (132, 197)
    if ntiles == 0:
        assert not ref['pred'].any() and int(ref['heatmap'].max()) == int(255 * 0.25)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_few_worker, args=(r, 2, port, q, ntiles)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
    for rank in (0, 1):
        assert got[rank]['mode'] == ref['mode']
        for k in ('logits', 'pred', 'classes', 'heatmap'):
            assert np.array_equal(got[rank][k], ref[k]), (rank, k)


# ------------------------------------------------------------------------------ region bags over two ranks (cfg4)
def _regions(rank, world, precision='parity'):
    import myargs
    import resnets_shift
    import utils.eval as val
    from utils.dataset_hr import GenerateIterator_eval
    from wsi_segmentation_pipeline_amd import synthetic as W
    from wsi_segmentation_pipeline_amd.slide import ArraySlide
    myargs.args.batch_size = 3
    myargs.args.class_probs = [0., 0., 0., 0.]
    rng = np.random.default_rng(4)
    l0 = rng.integers(0, 256, (1024, 1536, 3), dtype=np.uint8)
    slide = ArraySlide([l0, l0[::4, ::4], l0[::16, ::16]], [1.0, 4.0, 16.0])
    label_shape = (64, 96)
    metadata = {}
    for rid in range(11):
        ys, xs = np.nonzero(rng.random(label_shape) < 0.02 * (rid + 1))
        metadata[rid] = {'cnt_xy': rng.integers(8, [88, 56], (10, 2)), 'perim_xy': rng.integers(8, [88, 56], (12, 2)),
                         'wsipath': 'unused', 'scan_level': 2, 'foreground_indices': (ys, xs), 'tile_id': rid}
    model = resnets_shift.resnet18(False, precision=precision)
    model.load_state_dict(W.make_resnet18_state_dict(11))
    model = model.cuda().eval()
    it = GenerateIterator_eval(metadata, scan=slide)
    assert len(it.dataset) == 11
    out = val.predict_regions(model, it, metadata, label_shape, rank=rank, world=world)
    if precision == 'auto' and world > 1:
        rep = model.hip_engine(torch.device('cuda', torch.cuda.current_device())).report
        assert rep['scope'] == 'regions' and rep['mode'] in ('mx', 'parity'), rep      # ONE collective decision from the shard-wide probe
        return out, rep['mode'], rep['probe_error']
    return out


def _regions_worker(rank, world, port, q, precision='parity'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    q.put((rank, _regions(rank, world, precision)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_region_bags_equal_single_rank():
    """predict_regions with the bags sharded over two ranks (greedy balance, one all-gather of the ensemble logits, device paint
    on every rank) == the single-rank label image, bit for bit."""
    ref = _regions(0, 1)
    assert ref.any()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_regions_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
    assert np.array_equal(got[0], ref) and np.array_equal(got[1], ref)


def test_two_rank_region_bags_auto_precision_is_one_decision():
    """precision='auto' over two ranks (r03 advisor finding): the probe is a stratified sample over each rank's WHOLE shard of bags,
    its maximum is all-reduced, and both ranks run - and report - the same mode; the painted label image equals the single-rank
    one of that mode."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_regions_worker, args=(r, 2, port, q, 'auto')) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
    (m0, mode0, err0), (m1, mode1, err1) = got[0], got[1]
    assert mode0 == mode1 and err0 == err1, (mode0, mode1, err0, err1)          # the all-reduced probe error, the same decision
    ref = _regions(0, 1, mode0)
    assert np.array_equal(m0, ref) and np.array_equal(m1, ref)


# ------------------------------------------------------------------------------ dense 'seg' mode over two ranks
def _seg(rank, world, out_dir):
    import myargs
    import utils.dataset as ds
    import utils.eval as val
    from wsi_segmentation_pipeline_amd import synthetic as W
    from wsi_segmentation_pipeline_amd.slide import ArraySlide
    from wsi_segmentation_pipeline_amd.unet import UNetSeg
    a = myargs.args
    a.scan_level, a.scan_resize, a.num_classes, a.class_probs = 2, 1, 4, [0., 0., 0., 0.]
    a.tile_w = a.tile_h = 64
    a.tile_stride_w = a.tile_stride_h = 48
    a.val_save_pth, a.wsi_mask_pth = os.path.join(out_dir, 'out%d' % rank), os.path.join(out_dir, 'nomask')
    rng = np.random.default_rng(13)
    l2 = np.clip(np.kron(rng.integers(60, 250, (7, 9, 3)), np.ones((32, 32, 1))) + rng.integers(-25, 25, (224, 288, 3)), 0, 255).astype(np.uint8)
    slide = ArraySlide([l2[:8, :8], l2[:8, :8], l2], [1.0, 4.0, 16.0])                       # only level 2 is read
    slide.level_dimensions = ((288 * 16, 224 * 16), (288 * 4, 224 * 4), (288, 224))
    slide.name = 'seg.svs'
    model = UNetSeg(4)
    model.load_state_dict(W.make_unet_state_dict(5, classes=4))
    model = model.cuda().eval()
    dataset = ds.Dataset_wsis({'seg.svs': slide}, {'ph': 64, 'pw': 64, 'sh': 48, 'sw': 48}, bs=5)
    r = val.predict_tumorbed(model, dataset, 1, mode='seg', rank=rank, world=world, save=False)['seg.svs']
    return {'heatmap': r['heatmap'], 'classes': r['classes']}


def _seg_worker(rank, world, port, q, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    q.put((rank, _seg(rank, world, out_dir)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_seg_mode_equals_single_rank(tmp_path):
    """predict_tumorbed(mode='seg') with the tile list cut over two ranks, the band gather of the float64 maps to rank 0 and the
    broadcast of the u8 maps (seg mode) == the single-rank classes and heat map, byte for byte, on every rank."""
    ref = _seg(0, 1, str(tmp_path))
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seg_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
    for rank in (0, 1):
        for k in ('heatmap', 'classes'):
            assert np.array_equal(got[rank][k], ref[k]), (rank, k)


@pytest.mark.parametrize('workload', ['cfg3', 'cfg4'])
def test_rccl_path_rehearsal_single_rank(workload):
    """The 'nccl' (= RCCL) code path of bench.py - init_process_group with device_id, device all_gather_into_tensor,
    barrier, all_reduce(MAX), destroy - cannot run with two ranks on a one-GPU box, so it is rehearsed with ONE rank and
    the collective forced (WSI_FORCE_COLLECTIVE=1): same calls, same tensors, a one-rank communicator.  The line must be
    well-formed and the value plausible."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WSI_FORCE_COLLECTIVE='1', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    extra = ['--size', '6000'] if workload == 'cfg3' else ['--regions', '200']
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--workload', workload, '--steps', '1', '--warmup', '1',
                        '--no-cpu-baseline', '--no-bf16-leg'] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['value'] > 1000 and line['scaling'] == 'strong'
