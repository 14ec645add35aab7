"""Slide ingestion for the tile producer (reference utils/dataset.py:171-185): decoder threads fill
pinned host slots, the native ring (csrc/ingest.hip, wsi_ring_*) copies each band to the device and unpacks it into the
HBM-resident RGB level on a copy stream of its own; the compute stream only waits on a fence.  The decoder is whatever the
slide object provides (`read_region` of OpenSlide / ArraySlide, or a raw band reader): JPEG/TIFF decode itself is OpenSlide's
and is not rebuilt here.  Also the device side of `scan_resize != 1` (Pillow-exact bicubic tile resize, wsi_resample_*)."""
import collections
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import native
from .engine import _ptr, _require_gpu, _stream


class IngestRing:
    def __init__(self, slots=4, slot_bytes=64 << 20):
        self.lib = native.load()
        if not torch.cuda.is_available():
            raise RuntimeError('IngestRing needs a GPU (pinned slots + copy stream live in libwsi_hip)')
        self.slots, self.slot_bytes = int(slots), int(slot_bytes) & ~3
        h = C.c_void_p()
        native.check(self.lib.wsi_ring_create(C.byref(h), self.slots, self.slot_bytes), 'wsi_ring_create')
        self._h = h
        self.device_index = int(self.lib.wsi_ring_device(h))          # staging buffers + copy stream live on this device

    def close(self):
        if self._h is not None:
            self.lib.wsi_ring_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def slot_array(self, slot, shape):
        """numpy view of a pinned slot."""
        n = int(np.prod(shape))
        if n > self.slot_bytes:
            raise ValueError('band of %d bytes does not fit a %d-byte slot' % (n, self.slot_bytes))
        p = self.lib.wsi_ring_host_slot(self._h, slot)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n,)).reshape(shape)

    def upload_level(self, read_band, h, w, channels, device=None, out=None, workers=None, fence=True):
        """Fill an (h, w, 3) uint8 device level from `read_band(y0, rows, dst)` (dst: (rows, w, channels) pinned uint8 view the
        reader fills; called on worker threads, one band each).  Decode of band i+1.. overlaps the copy of band i.
        Returns the device tensor; with fence=True the current compute stream is ordered after the last copy."""
        dev = torch.device('cuda', self.device_index) if device is None else torch.device(device)
        if dev.type != 'cuda' or (dev.index if dev.index is not None else torch.cuda.current_device()) != self.device_index:
            raise ValueError('this ring stages for cuda:%d, not %s (one ring per device: default_ring(device))' % (self.device_index, dev))
        dev = torch.device('cuda', self.device_index)
        if out is None:
            out = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
        _require_gpu(out, 'level')
        if out.device.index != self.device_index:
            raise ValueError('out lives on %s, the ring on cuda:%d' % (out.device, self.device_index))
        # `out` may be a block the caching allocator just recycled on the compute stream: the ring's copies must not overtake the
        # kernels still queued there that read its previous contents
        with torch.cuda.device(dev):
            native.check(self.lib.wsi_ring_acquire(self._h, _stream()), 'wsi_ring_acquire')
        if tuple(out.shape) != (h, w, 3) or not out.is_contiguous():
            raise ValueError('out must be a contiguous (h, w, 3) uint8 tensor')
        row_bytes = w * channels
        band = max(1, min(h, self.slot_bytes // row_bytes))
        if band * row_bytes > self.slot_bytes or (channels == 4 and row_bytes % 4):
            raise ValueError('slot too small for one row')
        nb = -(-h // band)
        pool = ThreadPoolExecutor(max_workers=min(workers or self.slots, self.slots))
        inflight = collections.deque()

        def decode(i, s):
            y0 = i * band
            rows = min(band, h - y0)
            read_band(y0, rows, self.slot_array(s, (rows, w, channels)))
            return y0, rows

        def submit(fut, s):
            y0, rows = fut.result()
            native.check(self.lib.wsi_ring_submit(self._h, s, rows, w, channels, row_bytes, out.data_ptr() + y0 * w * 3, w * 3), 'wsi_ring_submit')
        try:
            for i in range(nb):
                s = i % self.slots
                if len(inflight) == self.slots:
                    submit(*inflight.popleft())
                native.check(self.lib.wsi_ring_wait_slot(self._h, s), 'wsi_ring_wait_slot')
                inflight.append((pool.submit(decode, i, s), s))
            while inflight:
                submit(*inflight.popleft())
        finally:
            pool.shutdown(wait=True)
        if fence:
            with torch.cuda.device(dev):
                native.check(self.lib.wsi_ring_fence(self._h, _stream()), 'wsi_ring_fence')
        return out

    def drain(self):
        native.check(self.lib.wsi_ring_drain(self._h), 'wsi_ring_drain')


_default_rings = {}


def default_ring(device=None):
    """The process's ring of `device` (default: the current one) - one per device: staging buffers and copy stream are device
    resources."""
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _default_rings:
        with torch.cuda.device(idx):
            _default_rings[idx] = IngestRing()
    return _default_rings[idx]


def level_from_slide(scan, level, device, ring=None):
    """One pyramid level of an OpenSlide-like object (`read_region((x0, y0_level0), level, (w, rows))` -> RGBA PIL image) into
    HBM through the ring."""
    ring = ring or default_ring(device)
    w, h = scan.level_dimensions[level]
    ds = scan.level_downsamples[level]

    def read_band(y0, rows, dst):
        dst[...] = np.asarray(scan.read_region((0, int(y0 * ds)), level, (w, rows)))
    out = ring.upload_level(read_band, h, w, 4, device)
    return out


def level_from_array(arr, device, ring=None):
    """Host ndarray / memmap (h, w, 3|4) uint8 -> HBM through the ring (the band copy into the pinned slot is the "decode")."""
    ring = ring or default_ring(device)
    h, w, ch = arr.shape
    if ch == 4 and (w * 4) % 4 == 0:
        def read_band(y0, rows, dst):
            dst[...] = arr[y0:y0 + rows]
        return ring.upload_level(read_band, h, w, 4, device)

    def read_band3(y0, rows, dst):
        dst[...] = arr[y0:y0 + rows, :, :3]
    return ring.upload_level(read_band3, h, w, 3, device)


# ------------------------------------------------------------------------------------------ scan_resize != 1
_plans = {}


def _plan(lib, in_hw, out_hw, device):
    key = (tuple(in_hw), tuple(out_hw), str(device))
    if key not in _plans:
        h = C.c_void_p()
        with torch.cuda.device(device):
            native.check(lib.wsi_resample_plan_create(C.byref(h), in_hw[0], in_hw[1], out_hw[0], out_hw[1]), 'wsi_resample_plan_create')
        _plans[key] = h
    return _plans[key]


def resize_tiles_bicubic(level, tile_xy, in_hw, out_hw):
    """n tiles of in_hw at tile_xy (level pixels, (x, y)) of the (H,W,3) uint8 GPU level -> (n, out_h, out_w, 3) uint8, exactly
    PIL `Image.resize((out_w, out_h))` of each crop (reference utils/dataset.py:180-181)."""
    lib = native.load()
    _require_gpu(level, 'level')
    if level.dtype != torch.uint8 or level.dim() != 3 or level.shape[2] != 3 or not level.is_contiguous():
        raise ValueError('level must be a contiguous (H,W,3) uint8 tensor')
    xy = torch.as_tensor(tile_xy, dtype=torch.int32).to(level.device).contiguous()
    n = xy.shape[0]
    plan = _plan(lib, in_hw, out_hw, level.device)
    out = torch.empty((n, out_hw[0], out_hw[1], 3), dtype=torch.uint8, device=level.device)
    scratch = torch.empty(max(1, lib.wsi_resample_scratch_bytes(plan, n)), dtype=torch.uint8, device=level.device)
    native.check(lib.wsi_resample_tiles(plan, _ptr(level), level.stride(0), level.shape[0], level.shape[1], _ptr(xy), n, _ptr(out),
                                        _ptr(scratch), _stream()), 'wsi_resample_tiles')
    return out
