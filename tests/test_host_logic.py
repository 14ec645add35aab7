"""CPU tests of the host logic: vectorised tile grid vs the oracle's restatement of the reference
loops, hand-derived tile counts, sharding, and the N>1 gather path on gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wsi_oracle as WO
from wsi_segmentation_pipeline_amd import slide as S


def test_grid_counts_hand_derived():
    # 40 000^2, tile 256 / stride 256: 156x156 interior + 156 + 156 = 24 648; last interior x = 39 681
    g = S.tile_grid(40000, 40000, 256, 256, 256, 256)
    assert len(g) == 24648
    assert g[155, 0] == 39681 and g[156 * 156, 0] == 39743
    # reference defaults tile 512 / stride 128 (myargs.py:105-112)
    assert len(S.tile_grid(40000, 40000, 512, 512, 128, 128)) == 96099


@pytest.mark.parametrize('case', [
    (1000, 700, 64, 64, 64, 64, 1 / 16), (1000, 700, 128, 64, 32, 48, 0.25), (513, 300, 256, 256, 100, 7, 1.0),
    (300, 300, 256, 256, 256, 256, 1 / 4), (256, 256, 256, 256, 64, 64, 1.0), (100, 50, 256, 256, 64, 64, 1.0),
])
def test_grid_matches_oracle(case):
    iw, ih, ph, pw, sh, sw, m = case
    rng = np.random.default_rng(iw + ih)
    assert S.tile_grid(iw, ih, ph, pw, sh, sw).tolist() == [list(t) for t in WO.tile_grid(iw, ih, ph, pw, sh, sw)]
    mh, mw = max(1, int(ih * m)), max(1, int(iw * m))
    for density in (0.02, 0.06, 0.5):
        mask = (rng.random((mh, mw)) < density).astype(np.uint8)
        mask[: mh // 3] = 0
        with np.errstate(all='ignore'):
            ref = WO.tile_grid(iw, ih, ph, pw, sh, sw, mask, m)
        got = S.tile_grid(iw, ih, ph, pw, sh, sw, mask, m)
        assert got.tolist() == [list(t) for t in ref]


def test_map_coords_truncation():
    xy = np.array([[1, 1], [257, 513], [39743, 39681]])
    m = 1 / 16
    assert S.map_coords(xy, m).tolist() == [[int(m * float(x)), int(m * float(y))] for x, y in xy]


def test_shard_range_partitions():
    for total in (0, 1, 7, 24648, 10000):
        for world in (1, 2, 3, 8):
            spans = [S.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_find_nuclei_and_isforeground():
    rgb = np.array([[[255, 255, 255], [200, 150, 200], [0, 0, 0], [100, 91, 100], [100, 90, 100], [10, 9, 10]]], np.uint8)
    assert WO.find_nuclei_hsv(rgb).tolist() == [[0, 1, 0, 0, 0, 0]]     # S = 0, .25, 0, .09, .1 (not >), .1
    assert WO.isforeground(np.array([1] + [0] * 19)) and not WO.isforeground(np.array([1] + [0] * 20))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gather_worker(rank, world, port, total, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    full = torch.arange(total * 4, dtype=torch.float32).view(total, 4)
    lo, hi = S.shard_range(total, rank, world)
    out = S.gather_tile_logits(full[lo:hi].clone(), total, rank, world)
    q.put((rank, bool(torch.equal(out, full))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('total', [11, 2, 1])
def test_gather_tile_logits_gloo_world2(total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


# ------------------------------------------------------------------------------ rank-resident slide regions
@pytest.mark.parametrize('grid', [(64, 64, 64, 64), (64, 64, 48, 40), (128, 64, 100, 30)])
def test_region_plan_atlas_equals_full_slide(grid):
    """Every tile read from a rank's atlas (only the rectangles its own tiles touch) == the same tile read from the
    whole slide, for every rank of every world size; the atlases together stay close to ONE copy of the slide."""
    ph, pw, sh, sw = grid
    src = S.SyntheticRows(1000, 700, 5, 'cpu', block=64)
    full = src.full()
    tl = S.tile_grid(1000, 700, ph, pw, sh, sw)
    for world in (1, 2, 3, 8):
        total = 0
        for r in range(world):
            lo, hi = S.shard_range(len(tl), r, world)
            rects, hw, loc = S.region_plan(tl[lo:hi], pw, ph)
            at = S.resident_regions(src, rects, hw, 'cpu')
            total += at.numel()
            for (x, y), (lx, ly) in zip(tl[lo:hi], loc):
                assert torch.equal(at[ly:ly + ph, lx:lx + pw], full[y:y + ph, x:x + pw])
        assert total <= 1.6 * full.numel() + world * 3 * ph * 1000 * 2          # bands + a shelf or two per rank
    rects, hw, loc = S.region_plan(np.zeros((0, 2), np.int32), pw, ph)           # a rank without tiles
    assert rects == [] and len(loc) == 0


def test_region_plan_cfg3_footprint():
    """cfg3 (40 000^2, 24 648 tiles): at 8 ranks every rank keeps ~0.6 GB of the 4.8 GB slide, the last one included
    (its 138 right-edge tiles are shelf-packed, not kept as a 35 000-row strip)."""
    tl = S.tile_grid(40000, 40000, 256, 256, 256, 256)
    sizes = []
    for r in range(8):
        lo, hi = S.shard_range(len(tl), r, 8)
        rects, hw, loc = S.region_plan(tl[lo:hi], 256, 256)
        assert loc.min() >= 0 and loc[:, 0].max() + 256 <= hw[1] and loc[:, 1].max() + 256 <= hw[0]
        sizes.append(hw[0] * hw[1] * 3)
    assert max(sizes) < 0.7e9 and sum(sizes) < 1.06 * 4.8e9


def _tile_features(buf, xy, ph, pw):
    # exact per-tile integers (channel sums) + the corner sample: any wrong crop changes them
    return torch.stack([torch.cat([buf[y:y + ph, x:x + pw].to(torch.int64).sum((0, 1)).float() / 64.0,
                                   buf[y, x, :1].float()]) for x, y in xy]) if len(xy) else torch.zeros((0, 4))


def _resident_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    src = S.SyntheticRows(1000, 700, 9, 'cpu', block=64)
    tl = S.tile_grid(1000, 700, 64, 64, 48, 40)
    lo, hi = S.shard_range(len(tl), rank, world)
    rects, hw, loc = S.region_plan(tl[lo:hi], 64, 64)
    atlas = S.resident_regions(src, rects, hw, 'cpu')                        # this rank never builds the whole slide
    out = S.gather_tile_logits(_tile_features(atlas, loc, 64, 64), len(tl), rank, world)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_resident_shards_gathered_equal_full_slide_gloo_world2():
    """The N>1 data path of bench.py's cfg3: shard -> each rank's resident regions -> per-tile values -> ONE
    all-gather; both ranks must end with exactly what one rank computes from the whole slide."""
    src = S.SyntheticRows(1000, 700, 9, 'cpu', block=64)
    tl = S.tile_grid(1000, 700, 64, 64, 48, 40)
    ref = _tile_features(src.full(), tl, 64, 64).numpy()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_resident_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert np.array_equal(got[0], ref) and np.array_equal(got[1], ref)


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher: the parent starts N child processes with the rendezvous
    environment and fails when a child fails (no GPU needed: the children are stubbed by a tiny script)."""
    import bench
    import sys as _sys
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        stub = os.path.join(d, 'stub.py')
        with open(stub, 'w') as f:
            f.write('import os, sys\n'
                    'r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])\n'
                    'assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0\n'
                    'assert int(os.environ["LOCAL_RANK"]) == r and w == 3\n'
                    'open(os.path.join(os.path.dirname(__file__), "rank%d" % r), "w").write("x")\n'
                    'sys.exit(int(os.environ.get("STUB_FAIL", "-1")) == r)\n')
        monkeypatch.setattr(bench, '__file__', stub)
        monkeypatch.setattr(_sys, 'argv', ['bench.py', '--gpus', '3'])
        assert bench.spawn_ranks(3) == 0
        assert sorted(f for f in os.listdir(d) if f.startswith('rank')) == ['rank0', 'rank1', 'rank2']
        monkeypatch.setenv('STUB_FAIL', '1')
        assert bench.spawn_ranks(3) != 0


# ------------------------------------------------------------------------------ region bags across ranks (cfg4)
def test_shard_bags_balances_and_partitions():
    from wsi_segmentation_pipeline_amd import bags as B
    rng = np.random.default_rng(0)
    for R in (0, 1, 7, 1000):
        for world in (1, 2, 3, 8):
            sh = B.shard_bags(np.full(R, 16.0), world)
            assert sorted(np.concatenate(sh).tolist()) == list(range(R))
            assert max(len(s) for s in sh) - min(len(s) for s in sh) <= 1
            assert all(np.all(np.diff(s) > 0) for s in sh)
    costs = rng.lognormal(2, 1, 500)
    sh = B.shard_bags(costs, 8)
    loads = [costs[s].sum() for s in sh]
    assert max(loads) - min(loads) <= costs.max()                      # LPT: within one item of perfect balance


def _bag_gather_worker(rank, world, port, q):
    from wsi_segmentation_pipeline_amd import bags as B
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    R = 11
    full = torch.arange(R * 4, dtype=torch.float32).view(R, 4)
    costs = np.array([5, 1, 9, 2, 2, 7, 3, 3, 8, 1, 4], np.float64)
    shards = B.shard_bags(costs, world)
    out = B.gather_rows(full[torch.as_tensor(shards[rank])].clone(), shards, rank, world)
    q.put((rank, bool(torch.equal(out, full))))
    dist.barrier()
    dist.destroy_process_group()


def test_bag_logits_gather_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bag_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_fastdiv_magic_is_exact_for_31_bit_dividends():
    """csrc/common.h FastDiv: q = mul_hi(i, m) >> s with m = floor(2^(31+l) / d) + 1, l = ceil(log2 d), must equal i // d for
    every 0 <= i < 2^31 (the dense conv kernels decode pixel indices with it).  Checked on the boundary dividends of many
    divisors, including every map size the kernels see."""
    import random
    rng = random.Random(5)
    divisors = list(range(1, 2050)) + [65 * 65, 257 * 257, 4096, 65536, 12345, 999983, 2 ** 20 + 1, 2 ** 30, 2 ** 31 - 1] + \
        [rng.randrange(2, 2 ** 31) for _ in range(300)]
    top = 2 ** 31 - 1
    for d in divisors:
        if d <= 1:
            continue
        l = (d - 1).bit_length()
        m = (1 << (31 + l)) // d + 1
        assert m < 2 ** 32
        s = l - 1
        cands = {0, 1, d - 1, d, d + 1, top, top - 1, top // d * d, top // d * d - 1}
        for k in (2, 3, 7, 1000, rng.randrange(1, max(2, top // d))):
            cands.update({k * d - 1, k * d, k * d + 1})
        cands.update(rng.randrange(0, top + 1) for _ in range(20))
        for i in cands:
            if 0 <= i <= top:
                assert ((i * m) >> 32) >> s == i // d, (d, i)


# ------------------------------------------------------------------------------ dense 'seg' mode: all-reduce of the float64 maps
def _allreduce_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = torch.Generator().manual_seed(3)
    tiles = torch.randn(40, 4, 8, 8, generator=g)                                   # fp32 addends, as the decoder produces them
    xy = torch.randint(0, 24, (40, 2), generator=g)
    lo, hi = S.shard_range(40, rank, world)
    pred = torch.zeros(4, 32, 32, dtype=torch.float64)
    for t in range(lo, hi):
        x, y = int(xy[t, 0]), int(xy[t, 1])
        pred[:, y:y + 8, x:x + 8] += tiles[t].double()
    out = S.allreduce_map(pred)
    ref = torch.zeros(4, 32, 32, dtype=torch.float64)
    for t in torch.randperm(40, generator=g).tolist():                              # any order: the sums are exact
        x, y = int(xy[t, 0]), int(xy[t, 1])
        ref[:, y:y + 8, x:x + 8] += tiles[t].double()
    q.put((rank, bool(torch.equal(out, ref))))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_map_gloo_world2():
    """slide.allreduce_map (the seg-mode exchange): per-rank float64 maps of disjoint tile shares sum to the single-rank map bit
    for bit, whatever the accumulation order."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_allreduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_tile_bands_cover_a_raster_share_disjointly():
    """slide.tile_bands on contiguous raster shares of the reference grid (dataset.py:143-166: interior rows, then the edge column,
    then the bottom row) at stride < tile: the rectangles are disjoint, contain every pixel the share's tiles touch, and the eight
    shares together cost about one map (plus the overlap rows) - not world x the map, as the r02 all-reduce did."""
    H = W = 500
    ph, sh = 32, 24
    ys, xs = list(range(1, H - 1 - ph, sh)), list(range(1, W - 1 - ph, sh))
    tiles = [(x, y) for y in ys for x in xs] + [(W - 1 - ph, y) for y in ys] + [(x, H - 1 - ph) for x in xs]
    tiles = np.asarray(tiles, np.int64)
    total_px = 0
    for world in (8, 3, 1):
        sent = 0
        for rank in range(world):
            lo, hi = S.shard_range(len(tiles), rank, world)
            rects = S.tile_bands(tiles[lo:hi], ph, ph, (H, W))
            cover = np.zeros((H, W), np.int32)
            for y0, y1, x0, x1 in rects.tolist():
                cover[y0:y1, x0:x1] += 1
            assert cover.max() <= 1                                             # disjoint
            touched = np.zeros((H, W), bool)
            for x, y in tiles[lo:hi].tolist():
                touched[y:y + ph, x:x + ph] = True
            assert bool((cover[touched] == 1).all())                            # complete
            assert len(rects) <= 8
            sent += int(cover.sum())
        total_px = max(total_px, sent)
        assert sent <= 1.45 * H * W                                             # overlap rows between shares + edge column / bottom row hulls
    assert len(S.tile_bands(np.zeros((0, 2), np.int64), ph, ph, (H, W))) == 0   # a rank without tiles sends nothing
    # tiles hanging over the map edge are clipped
    r = S.tile_bands(np.asarray([[490, 490]]), ph, ph, (H, W))
    assert r.tolist() == [[490, 500, 490, 500]]


# ------------------------------------------------------------------------------ every collective at world 8 and 3, uneven and empty shards
def _all_collectives_worker(rank, world, port, total, q):
    """One process of `world`: tile-logit gather, bag-row gather, map all-reduce, span / probe-error reductions - with `total`
    tiles / bags not divisible by the world (and total < world: ranks with nothing)."""
    from wsi_segmentation_pipeline_amd import bags as B
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ok = {}
    S.OP_LOG = []                                     # every collective of the path records (op, shape, dtype) here (slide._Wire)
    # (1) slide.gather_tile_logits: contiguous raster shares of `total` tiles
    full = torch.arange(total * 4, dtype=torch.float32).view(total, 4) * 0.5 - 3
    lo, hi = S.shard_range(total, rank, world)
    ok['tiles'] = bool(torch.equal(S.gather_tile_logits(full[lo:hi].clone(), total, rank, world), full))
    # (2) bags.gather_rows: greedy cost-balanced shares (non-contiguous index sets, empty ones when total < world)
    costs = np.random.default_rng(7).lognormal(2, 1, total)
    shards = B.shard_bags(costs, world)
    mine = torch.as_tensor(np.asarray(shards[rank], np.int64))
    ok['bags'] = bool(torch.equal(B.gather_rows(full[mine].clone(), shards, rank, world), full))
    # (3) slide.allreduce_map: exact float64 sums of fp32 addends
    g = torch.Generator().manual_seed(3)
    tiles = torch.randn(total, 4, 8, 8, generator=g)
    xy = torch.randint(0, 24, (total, 2), generator=g)
    pred, ref = torch.zeros(4, 32, 32, dtype=torch.float64), torch.zeros(4, 32, 32, dtype=torch.float64)
    for t in range(total):
        x, y = int(xy[t, 0]), int(xy[t, 1])
        ref[:, y:y + 8, x:x + 8] += tiles[t].double()
        if lo <= t < hi:
            pred[:, y:y + 8, x:x + 8] += tiles[t].double()
    part = pred.clone()                                                              # (allreduce_map sums in place)
    ok['map'] = bool(torch.equal(S.allreduce_map(pred), ref))
    # (3b) slide.gather_map_bands: the same sum as a direct band gather to rank 0 (what predict_tumorbed(mode='seg') uses);
    #      the other ranks keep their partial maps; the u8 maps then travel by broadcast_from
    mine_xy = xy[lo:hi].numpy()
    got = S.gather_map_bands(part.clone(), mine_xy, 8, 8, rank, world, dst=0)
    ok['bands'] = bool(torch.equal(got, ref)) if rank == 0 else bool(torch.equal(got, part))
    u8 = (ref[0] > 0).to(torch.uint8) if rank == 0 else torch.empty(32, 32, dtype=torch.uint8)
    ok['bcast'] = bool(torch.equal(S.broadcast_from(u8, 0), (ref[0] > 0).to(torch.uint8)))
    # (4) the reductions beside it: exponent span (None on a rank without tiles) and the precision probe's maximum
    span = torch.tensor([120 + rank, 130 - rank], dtype=torch.int32) if hi > lo else None
    owners = [r for r in range(world) if S.shard_range(total, r, world)[1] > S.shard_range(total, r, world)[0]]
    sp = S.allreduce_span(span, torch.device('cpu')).tolist()
    ok['span'] = sp == [120 + min(owners), 130 - min(owners)]
    ok['max'] = S.allreduce_max(0.25 * rank if hi > lo else 0.0, torch.device('cpu'), world) == 0.25 * max(owners)
    # The op sequence.  RCCL and gloo run the SAME op list (slide._Wire: only the wire buffer's place differs), so what is asserted
    # here on gloo is the list an 8-GPU RCCL run issues: collectives identical on every rank (shapes included - a mismatch is a
    # hang on hardware), point-to-point ops only between the owners of map bands and rank 0.
    chunk = (total + world - 1) // world
    bchunk = max(len(s_) for s_ in shards)
    log = list(S.OP_LOG)
    coll = [e for e in log if e[0] not in ('send', 'recv')]
    p2p = [e for e in log if e[0] in ('send', 'recv')]
    ok['ops_head'] = coll[:3] == [('all_gather', (chunk, 4), 'float32'), ('all_gather', (bchunk, 4), 'float32'), ('all_reduce_sum', (4, 32, 32), 'float64')]
    ok['ops_bands'] = coll[3][:1] == ('all_gather',) and coll[3][1:] == ((1,), 'int64') and coll[4][0] == 'all_gather' and coll[4][1][1] == 4
    ok['ops_tail'] = coll[5:] == [('broadcast', (32, 32), 'uint8'), ('all_reduce_min', (1,), 'int32'), ('all_reduce_max', (1,), 'int32'),
                                  ('all_reduce_max', (1,), 'float64')]
    if rank == 0:
        ok['ops_p2p'] = [e[0] for e in p2p] == ['recv'] * len([r for r in owners if r != 0])
    else:
        ok['ops_p2p'] = [e[0] for e in p2p] == (['send'] if hi > lo else [])
    q.put((rank, (ok, coll)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,total', [(8, 29), (8, 5), (3, 10), (3, 2)])
def test_collectives_uneven_and_empty_shards(world, total):
    """SURVEY.md 8e at the world sizes the driver scales to: 8 ranks (and 3) with totals that do not divide and totals below the
    world size, so some ranks own nothing - every collective of the path must still return the single-rank result on all ranks."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_all_collectives_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    assert sorted(res) == list(range(world))
    for r in range(world):
        assert all(res[r][0].values()), (r, res[r][0])
        assert res[r][1] == res[0][1], 'rank %d issued other collectives than rank 0' % r     # same ops, same shapes, same order everywhere


def test_bench_builds_eight_rank_environments(monkeypatch):
    """`python bench.py --gpus 8` (the driver's scaling run when no launcher is used): eight children, each with its own
    RANK / LOCAL_RANK, a shared 127.0.0.1 rendezvous and WORLD_SIZE 8; HSA_ENABLE_IPC_MODE_LEGACY stays 0 for RCCL."""
    import bench
    import sys as _sys
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        stub = os.path.join(d, 'stub.py')
        with open(stub, 'w') as f:
            f.write('import os, sys, json\n'
                    'keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"]\n'
                    'json.dump({k: os.environ.get(k) for k in keys} | {"argv": sys.argv[1:]},\n'
                    '          open(os.path.join(os.path.dirname(__file__), "env%s.json" % os.environ["RANK"]), "w"))\n')
        monkeypatch.setattr(bench, '__file__', stub)
        monkeypatch.setattr(_sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '2', '--warmup', '1'])
        assert bench.spawn_ranks(8) == 0
        import json
        envs = [json.load(open(os.path.join(d, 'env%d.json' % r))) for r in range(8)]
    assert [int(e['RANK']) for e in envs] == list(range(8)) and [int(e['LOCAL_RANK']) for e in envs] == list(range(8))
    assert {e['WORLD_SIZE'] for e in envs} == {'8'} and {e['MASTER_ADDR'] for e in envs} == {'127.0.0.1'}
    assert len({e['MASTER_PORT'] for e in envs}) == 1 and {e['HSA_ENABLE_IPC_MODE_LEGACY'] for e in envs} == {'0'}
    assert all(e['argv'] == ['--gpus', '8', '--steps', '2', '--warmup', '1'] for e in envs)


def test_batch_sizes_respect_cap_and_cover():
    """engine.batch_sizes: every batch <= cap, the sizes sum to n, all but the last are whole rounds (multiples of 512 tiles of
    256^2) when that fits the cap, equal batches otherwise."""
    from wsi_segmentation_pipeline_amd.engine import batch_sizes
    assert batch_sizes(24648, 6200) == [6162] * 4
    assert batch_sizes(24648, 6656) == [6144, 6144, 6144, 6216]
    assert batch_sizes(12324, 6656) == [6144, 6180]
    assert batch_sizes(3081, 6656) == [3081] and batch_sizes(0, 5) == [] and batch_sizes(7, 3) == [3, 3, 1]
    for n, cap, h in ((100000, 7000, 256), (64000, 5000, 64), (999, 100, 256), (13000, 6500, 256)):
        sz = batch_sizes(n, cap, h, h)
        assert sum(sz) == n and max(sz) <= cap and min(sz) > 0


# ------------------------------------------------------------------------------ precision='auto': one decision per slide and world
class _FakeTrunk:
    """Stands in for a TrunkEngine on the CPU: logits = per-slide, per-mode constants, calls are recorded."""
    def __init__(self, name, planes, log):
        self.name, self.planes, self.head_k, self.log = name, planes, 4, log

    def forward_tiles(self, slide, xy, ph, pw, feat=False, logits=True, fmap=False, tap=None):
        self.log.append((self.name, int(slide[0, 0, 0]), int(xy.shape[0])))
        # slide "1" is benign, slide "2" is hot ON RANK 1 ONLY (slide[0, 0, 1] carries the rank): mx is 1e-3 off parity there
        hot = int(slide[0, 0, 0]) == 2 and int(slide[0, 0, 1]) == 1
        off = 1e-3 if (self.name == 'mx' and hot) else 0.0
        lg = torch.full((xy.shape[0], 4), 1.0 + off)
        return (torch.zeros(xy.shape[0], 512), lg, None)


def _fake_auto(log):
    from wsi_segmentation_pipeline_amd.engine import AutoTrunkEngine
    eng = object.__new__(AutoTrunkEngine)
    eng._par, eng._mx = _FakeTrunk('parity', 2, log), _FakeTrunk('mx', 3, log)
    eng._dev, eng.tol, eng.probe = 'cpu', 5e-4, 4
    eng._chosen, eng._slide_key, eng._slide_ref, eng._probed = None, None, None, None
    eng.report = {'mode': None}
    return eng


def _auto_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    log = []
    eng = _fake_auto(log)
    modes = []
    xy = torch.zeros((8, 2), dtype=torch.int32)
    for sid in (1, 2, 1):                                   # benign, hot on rank 1 only, benign again (a NEW tensor each time)
        slide = torch.zeros((4, 4, 3), dtype=torch.uint8)
        slide[0, 0, 0], slide[0, 0, 1] = sid, rank
        # the sequence slide.infer_slide_cls runs: local probe, max over the ranks, decide, forward
        eng.decide(S.allreduce_max(eng.probe_tiles(slide, xy, 256, 256), 'cpu', world))
        n0 = len(log)
        eng.forward_tiles(slide, xy, 256, 256)
        eng.forward_tiles(slide, xy[:3], 256, 256)          # a second chunk of the same slide
        ran = [e for e in log[n0:]]
        modes.append((eng.report['mode'], [r[0] for r in ran]))
        del slide
    q.put((rank, modes))
    dist.barrier()
    dist.destroy_process_group()


def test_auto_precision_is_one_decision_per_slide_and_world():
    """r03 advisor finding: decide() left the previous slide's key in place, so from the second slide on forward_tiles probed again
    LOCALLY and could run another mode than the all-reduced decision (and than the `precision` report).  Two ranks, three slides,
    probe errors that straddle tol on one rank only: every rank must run the rank-global mode, with no second probe."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_auto_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    want = [('mx', ['mx', 'mx']), ('parity', ['parity', 'parity']), ('mx', ['mx', 'mx'])]
    assert res[0] == want and res[1] == want, res


def _verified_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    log = []
    eng = _fake_auto(log)
    res = []
    for sid, n in ((1, 8), (2, 8), (1, 8), (2, 0 if rank == 0 else 8)):     # benign, hot on rank 1 only, benign, hot with an EMPTY shard on rank 0
        slide = torch.zeros((4, 4, 3), dtype=torch.uint8)
        slide[0, 0, 0], slide[0, 0, 1] = sid, rank
        xy = torch.zeros((n, 2), dtype=torch.int32)
        n0 = len(log)
        out = eng.forward_tiles_verified(slide, xy, 256, 256, reduce_max=lambda e: S.allreduce_max(e, 'cpu', world))
        res.append((eng.report['mode'], [(r[0], r[2]) for r in log[n0:]], tuple(out[1].shape), float(out[1].max()) if n else None))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_auto_precision_verified_after_the_forward():
    """AutoTrunkEngine.forward_tiles_verified (r05, what slide.infer_slide_cls runs): the shard goes through mx first, the stratified sample
    once more through parity, the sample errors are max-reduced over the ranks and only then is the mode fixed - mx logits stand where every
    rank's sample agrees, otherwise EVERY rank runs its shard again in parity (also the one whose own sample was fine, and a rank without
    tiles still takes part in the reduction)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verified_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    benign = ('mx', [('mx', 8), ('parity', 4)], (8, 4), 1.0)
    for r in (0, 1):
        assert res[r][0] == benign and res[r][2] == benign, res[r]
        mode, calls, shape, mx = res[r][1]
        assert mode == 'parity' and calls == [('mx', 8), ('parity', 4), ('parity', 8)] and shape == (8, 4) and mx == 1.0, res[r][1]   # parity logits returned
    assert res[0][3] == ('parity', [], (0, 4), None), res[0][3]                         # no tiles, same decision, nothing to run
    assert res[1][3] == ('parity', [('mx', 8), ('parity', 4), ('parity', 8)], (8, 4), 1.0), res[1][3]


def test_auto_precision_does_not_trust_a_recycled_address():
    """Direct forward_tiles callers: a new slide tensor is a new slide even when the allocator hands it the old block (same
    data_ptr, same shape) - the engine keys on the tensor object or on a caller-supplied slide_id, never on the address."""
    log = []
    eng = _fake_auto(log)
    xy = torch.zeros((8, 2), dtype=torch.int32)
    store = torch.zeros((4, 4, 3), dtype=torch.uint8)
    a = store[:]                                            # two tensor objects over ONE address
    b = store[:]
    assert a.data_ptr() == b.data_ptr() and a is not b
    eng.forward_tiles(a, xy, 256, 256)
    n_probe_a = sum(1 for e in log if e[2] == 4)
    eng.forward_tiles(a, xy, 256, 256)                      # same object: no new probe
    assert sum(1 for e in log if e[2] == 4) == n_probe_a == 2
    eng.forward_tiles(b, xy, 256, 256)                      # same address, other object: probes again
    assert sum(1 for e in log if e[2] == 4) == 4
    eng.forward_tiles(store[:], xy, 256, 256, slide_id='S7')
    eng.forward_tiles(store[:], xy, 256, 256, slide_id='S7')   # explicit id: one probe for both views
    assert sum(1 for e in log if e[2] == 4) == 6
