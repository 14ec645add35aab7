"""Slide ingestion ring and the scan_resize != 1 input resize (reference utils/dataset.py:171-185) on the GPU."""
import time

import numpy as np
import pytest
import torch

from oracle import resize_oracle as RO

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip('PIL.Image')


@pytest.fixture(scope='module')
def ingest():
    from wsi_segmentation_pipeline_amd import ingest
    return ingest


@pytest.mark.parametrize('seed,in_hw,out_hw', [(0, (512, 512), (256, 256)), (1, (128, 128), (64, 64)), (2, (96, 160), (64, 64)),
                                               (3, (64, 64), (128, 96)), (4, (300, 200), (100, 67)), (5, (64, 64), (64, 64)),
                                               (6, (768, 768), (256, 256))])
def test_resize_tiles_match_pillow(ingest, seed, in_hw, out_hw):
    rng = np.random.default_rng(seed)
    H, W = in_hw[0] * 2 + 37, in_hw[1] * 2 + 11
    level = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    level[: H // 4] = 255
    xy = np.array([[0, 0], [W - in_hw[1], H - in_hw[0]], [13, 29], [in_hw[1] // 2 + 3, 7]], np.int32)
    got = ingest.resize_tiles_bicubic(torch.from_numpy(level).cuda(), xy, in_hw, out_hw).cpu().numpy()
    for i, (x, y) in enumerate(xy):
        crop = level[y:y + in_hw[0], x:x + in_hw[1]]
        want = np.asarray(PIL.fromarray(crop).resize((out_hw[1], out_hw[0])))
        assert np.array_equal(got[i], want), i
        assert np.array_equal(got[i], RO.resize_bicubic_u8(crop, out_hw))


def test_resize_out_of_slide_reads_zero(ingest):
    """A tile hanging over the slide edge reads black there, like OpenSlide's read_region + convert('RGB')."""
    rng = np.random.default_rng(3)
    level = rng.integers(1, 256, (200, 220, 3), dtype=np.uint8)
    xy = np.array([[150, 120], [-20, -30]], np.int32)
    got = ingest.resize_tiles_bicubic(torch.from_numpy(level).cuda(), xy, (128, 128), (64, 64)).cpu().numpy()
    for i, (x, y) in enumerate(xy):
        crop = np.zeros((128, 128, 3), np.uint8)
        y0, y1, x0, x1 = max(y, 0), min(y + 128, 200), max(x, 0), min(x + 128, 220)
        crop[y0 - y:y1 - y, x0 - x:x1 - x] = level[y0:y1, x0:x1]
        assert np.array_equal(got[i], np.asarray(PIL.fromarray(crop).resize((64, 64))))


@pytest.mark.parametrize('h,w,ch,slots,slot_kb', [(1000, 777, 4, 3, 256), (513, 1024, 3, 2, 64), (40, 33, 4, 4, 4), (2048, 2048, 4, 4, 4096)])
def test_ring_upload_exact(ingest, h, w, ch, slots, slot_kb):
    rng = np.random.default_rng(h)
    src = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
    ring = ingest.IngestRing(slots, slot_kb << 10)
    calls = []

    def read_band(y0, rows, dst):
        calls.append((y0, rows))
        time.sleep(0.001)
        dst[...] = src[y0:y0 + rows]
    out = ring.upload_level(read_band, h, w, ch, 'cuda:0')
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), src[..., :3])
    assert sum(r for _, r in calls) == h and len(calls) > 1
    # a second level through the same ring (slots are recycled), consumed on the compute stream right after the fence
    out2 = ring.upload_level(lambda y0, rows, dst: dst.__setitem__(Ellipsis, 255 - src[y0:y0 + rows]), h, w, ch, 'cuda:0')
    s = out2.to(torch.int64).sum().item()
    assert s == int((255 - src[..., :3]).astype(np.int64).sum())
    ring.drain()
    ring.close()


def test_ring_does_not_overtake_the_compute_stream(ingest):
    """`out` may be a block the caching allocator recycled while kernels that still read its old contents are
    queued on the compute stream; wsi_ring_acquire orders the ring's copy stream behind them.  A long kernel chain reads a
    buffer, the buffer is freed, the upload lands in the recycled block: the chain's result must be of the OLD contents."""
    h, w = 512, 1024
    old = torch.full((h, w, 3), 7, dtype=torch.uint8, device='cuda:0')
    ptr = old.data_ptr()
    acc = torch.zeros((), dtype=torch.int64, device='cuda:0')
    big = torch.ones((4096, 4096), device='cuda:0')
    for _ in range(60):                                    # ~tens of ms of queued work in front of the readers
        big = big @ big * 1e-4
    for _ in range(20):
        acc = acc + old.to(torch.int64).sum()              # queued readers of `old`
    del old                                                # the block returns to the allocator (same stream: no sync)
    ring = ingest.IngestRing(2, 1 << 20)
    src = np.full((h, w, 3), 200, np.uint8)
    out = ring.upload_level(lambda y0, rows, dst: dst.__setitem__(Ellipsis, src[y0:y0 + rows]), h, w, 3, 'cuda:0')
    assert ring.device_index == 0
    torch.cuda.synchronize()
    if out.data_ptr() == ptr:                              # (the allocator did recycle the block: the hazard was live)
        print('recycled block: hazard exercised')
    assert int(acc.item()) == 20 * 7 * h * w * 3
    assert np.array_equal(out.cpu().numpy(), src)
    with pytest.raises(ValueError):
        ring.upload_level(lambda *a: None, 4, 4, 3, 'cpu')
    ring.close()
    assert ingest.default_ring('cuda:0') is ingest.default_ring(torch.device('cuda', 0))


def test_ring_rejects_bad_arguments(ingest):
    from wsi_segmentation_pipeline_amd import native
    ring = ingest.IngestRing(2, 4096)
    dst = torch.empty((4, 8, 3), dtype=torch.uint8, device='cuda')
    lib = native.load()
    assert lib.wsi_ring_submit(ring._h, 5, 4, 8, 4, 32, dst.data_ptr(), 24) != 0        # slot out of range
    assert lib.wsi_ring_submit(ring._h, 0, 4, 8, 2, 16, dst.data_ptr(), 24) != 0        # channels
    assert lib.wsi_ring_submit(ring._h, 0, 400, 8, 4, 32, dst.data_ptr(), 24) != 0      # larger than the slot
    assert lib.wsi_ring_submit(ring._h, 0, 4, 8, 4, 32, dst.data_ptr(), 24) == 0
    assert lib.wsi_ring_submit(ring._h, 0, 4, 8, 4, 32, dst.data_ptr(), 24) != 0        # slot still in flight: wait_slot first
    assert lib.wsi_ring_wait_slot(ring._h, 0) == 0
    with pytest.raises(ValueError):
        ring.slot_array(0, (4097,))
    ring.close()


def test_openslide_adapter_level_through_ring(ingest):
    """_OpenSlideAdapter.device_level == the RGB of read_region(...).convert('RGB') of the whole level."""
    from utils import dataset as ds
    from wsi_segmentation_pipeline_amd import slide as S
    rng = np.random.default_rng(11)
    lv0 = rng.integers(0, 256, (1200, 900, 3), dtype=np.uint8)
    fake = S.ArraySlide([lv0, lv0[::4, ::4].copy()])
    ad = ds._OpenSlideAdapter(fake)
    for level in (0, 1):
        got = ad.device_level(level, torch.device('cuda:0'))
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), fake.level_array(level))


def test_tile_iterator_scan_resize(ingest, monkeypatch):
    """DeviceTileIterator with scan_resize = 2 == the host item path (read_region -> PIL resize -> ToTensor + Normalize)."""
    from myargs import args
    from utils import dataset as ds, preprocessing
    from wsi_segmentation_pipeline_amd import slide as S
    monkeypatch.setattr(args, 'scan_resize', 2)
    monkeypatch.setattr(args, 'scan_level', 0)
    monkeypatch.setattr(args, 'tile_w', 64)
    monkeypatch.setattr(args, 'tile_h', 64)
    rng = np.random.default_rng(5)
    lv = rng.integers(0, 256, (700, 650, 3), dtype=np.uint8)
    slide = S.ArraySlide([lv, lv[::4, ::4].copy(), lv[::16, ::16].copy()])
    params = preprocessing.DotDict({'ph': 128, 'pw': 128, 'sh': 96, 'sw': 96})
    d = ds.Dataset_wsi(slide, params)
    assert len(d) > 4
    it = ds.DeviceTileIterator(d, 7)
    k = 0
    for bx, by, img in it:
        assert img.shape[1:] == (3, 64, 64)
        for j in range(img.shape[0]):
            x, y, ref = d[k]
            assert (x, y) == (float(bx[j]), float(by[j]))
            assert torch.equal(img[j].cpu(), ref)
            k += 1
    assert k == len(d)
