"""Host logic of the sliding-window driver: slide sources, the tile grid, tile sharding across
ranks and the per-slide 'cls' inference pipeline (tile list -> HIP trunk -> gather -> stitch).

Mirrors, with vectorised NumPy instead of Python double loops:
  grid        reference utils/dataset.py:143-166
  fg filter   reference utils/preprocessing.py:60-71
  stitch      reference utils/eval.py:182-186,208-215 (device kernel wsi_stitch_add)
"""
import numpy as np
import torch

from . import engine as E


# ------------------------------------------------------------------------------ slide sources
class ArraySlide:
    """In-memory pyramid with the subset of the OpenSlide API the reference touches
    (level_dimensions, level_downsamples, level_count, read_region)."""

    def __init__(self, levels, downsamples=None):
        self.levels = [np.ascontiguousarray(l[..., :3], dtype=np.uint8) if isinstance(l, np.ndarray) else l for l in levels]
        self.level_downsamples = tuple(float(d) for d in (downsamples or [4.0 ** i for i in range(len(levels))]))
        self.level_dimensions = tuple((int(l.shape[1]), int(l.shape[0])) for l in self.levels)   # (w, h) like OpenSlide
        self.level_count = len(self.levels)
        self.dimensions = self.level_dimensions[0]
        self._dev = {}

    def level_array(self, level):
        l = self.levels[level]
        return l.cpu().numpy() if isinstance(l, torch.Tensor) else l

    def read_region(self, location, level, size):
        """(x, y) in level-0 pixels, size (w, h) at `level` -> PIL RGBA image, black/transparent outside."""
        from PIL import Image
        ds = self.level_downsamples[level]
        x, y = int(location[0] // ds), int(location[1] // ds)
        w, h = int(size[0]), int(size[1])
        arr = self.level_array(level)
        H, W = arr.shape[:2]
        out = np.zeros((h, w, 4), np.uint8)
        y0, y1, x0, x1 = max(y, 0), min(y + h, H), max(x, 0), min(x + w, W)
        if y1 > y0 and x1 > x0:
            out[y0 - y:y1 - y, x0 - x:x1 - x, :3] = arr[y0:y1, x0:x1]
            out[y0 - y:y1 - y, x0 - x:x1 - x, 3] = 255
        return Image.fromarray(out, 'RGBA')

    def device_level(self, level, device):
        """(H,W,3) uint8 tensor of one pyramid level resident in HBM."""
        key = (level, str(device))
        t = self._dev.get(key)
        if t is None:
            l = self.levels[level]
            t = l.to(device) if isinstance(l, torch.Tensor) else torch.from_numpy(l).to(device)
            self._dev[key] = t.contiguous()
        return self._dev[key]

    def close(self):
        self._dev.clear()


def synthetic_slide(size, seed, device, levels=3):
    """i.i.d. uniform u8 RGB slide generated on the device; level k is the
    4^k-subsampled level 0 (what a pyramid reader would hand back, minus interpolation)."""
    g = torch.Generator(device=device).manual_seed(seed)
    l0 = torch.randint(0, 256, (size, size, 3), dtype=torch.uint8, device=device, generator=g)
    lv = [l0]
    for k in range(1, levels):
        lv.append(l0[::4 ** k, ::4 ** k].contiguous())
    return ArraySlide(lv)


# ------------------------------------------------------------------------------ tile grid
def _window_nonzero_fraction(mask, xp, yp, dx, dy):
    """count_nonzero(mask[yp:yp+dy, xp:xp+dx]) / window.size with NumPy slice clipping, vectorised
    through a summed-area table.  Empty windows give NaN (-> not foreground), as in the reference."""
    H, W = mask.shape[:2]
    nz = (mask != 0) if mask.ndim == 2 else (mask != 0).any(-1)
    sat = np.zeros((H + 1, W + 1), np.int64)
    sat[1:, 1:] = nz.cumsum(0).cumsum(1)
    y0, y1 = np.clip(yp, 0, H), np.clip(yp + dy, 0, H)
    x0, x1 = np.clip(xp, 0, W), np.clip(xp + dx, 0, W)
    cnt = sat[y1, x1] - sat[y0, x1] - sat[y1, x0] + sat[y0, x0]
    size = (y1 - y0) * (x1 - x0)
    with np.errstate(divide='ignore', invalid='ignore'):
        return cnt / size


def tile_grid(iw, ih, ph, pw, sh, sw, mask=None, m=1.0, thresh=0.05):
    """Tile corners (T,2) int32 (x, y) in the reference's order: interior raster, right-edge column,
    bottom-edge row, no corner tile; keep iff the level-2 mask window is >= thresh nonzero."""
    ys = np.arange(1, ih - 1 - ph, sh, dtype=np.int64)
    xs = np.arange(1, iw - 1 - pw, sw, dtype=np.int64)
    gx, gy = np.meshgrid(xs, ys)                       # row-major: y outer, x inner
    x = np.concatenate([gx.ravel(), np.full(len(ys), iw - 1 - pw, np.int64), xs])
    y = np.concatenate([gy.ravel(), ys, np.full(len(xs), ih - 1 - ph, np.int64)])
    if mask is not None and len(x):
        dx, dy = int(pw * m), int(ph * m)
        yp = (y * m).astype(np.int64)                  # int(ypos * m): truncation of a non-negative float
        xp = (x * m).astype(np.int64)
        frac = _window_nonzero_fraction(np.asarray(mask), xp, yp, dx, dy)
        keep = frac >= thresh                          # NaN compares False
        x, y = x[keep], y[keep]
    return np.stack([x, y], 1).astype(np.int32)


def map_coords(tile_xy, m):
    """int(m * x), int(m * y) in float64, as reference utils/eval.py:214."""
    return np.floor(np.asarray(tile_xy, np.float64) * m).astype(np.int32)


# ------------------------------------------------------------------------------ sharding
def shard_range(total, rank, world):
    """Contiguous raster-order chunk [lo, hi) of the tile list owned by `rank`."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_tile_logits(local_logits, total, rank, world):
    """Concatenate per-rank logits in rank order on every rank.  One RCCL all-gather of equal-size
    padded chunks (payload = total x C fp32: latency-bound, a single collective)."""
    import torch.distributed as dist
    if world == 1:
        return local_logits
    c = local_logits.shape[1]
    chunk = (total + world - 1) // world
    dev = local_logits.device
    # RCCL ("nccl") gathers device buffers directly; gloo (CPU tests, or ranks sharing one GPU) goes via host memory
    via_host = dist.get_backend() != 'nccl' and local_logits.is_cuda
    buf = torch.zeros((chunk, c), dtype=local_logits.dtype, device='cpu' if via_host else dev)
    buf[:local_logits.shape[0]] = local_logits
    out = torch.empty((world * chunk, c), dtype=local_logits.dtype, device=buf.device)
    dist.all_gather_into_tensor(out, buf)
    if via_host:
        out = out.to(dev)
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(out[r * chunk:r * chunk + (hi - lo)])
    return torch.cat(parts, 0)


# ------------------------------------------------------------------------------ per-slide pipeline
def _upload(arr, dtype, dev):
    """Host array -> device without blocking the host on the GPU queue: a pageable-memory copy waits for everything already
    enqueued, so the host could not start enqueuing the next slide while this one computes (1.7 ms bubble per 10 k-tile slide)."""
    t = torch.as_tensor(np.ascontiguousarray(arr), dtype=dtype)
    if dev.type == 'cuda':
        return t.pin_memory().to(dev, non_blocking=True)
    return t.to(dev)


def infer_slide_cls(eng, slide_level_dev, tile_xy, ph, pw, m, map_hw, num_classes, class_probs, mask_dev=None,
                    rank=0, world=1, want_probs=True):
    """predict_tumorbed(mode='cls') for one slide on device (reference utils/eval.py:182-229):
    this rank's tiles -> fused read+transform+trunk+classifier -> (RCCL gather) -> float64 stitch ->
    softmax/threshold/argmax -> u8 heat map.  Returns dict of device tensors."""
    T = int(tile_xy.shape[0])
    lo, hi = shard_range(T, rank, world)
    dev = slide_level_dev.device
    xy_dev = _upload(tile_xy[lo:hi], torch.int32, dev)
    if hi > lo:
        _, logits, _ = eng.forward_tiles(slide_level_dev, xy_dev, ph, pw, logits=True)
    else:
        logits = torch.zeros((0, num_classes), dtype=torch.float32, device=dev)
    logits = gather_tile_logits(logits, T, rank, world)
    pred = torch.zeros((num_classes, map_hw[0], map_hw[1]), dtype=torch.float64, device=dev)
    if T:
        mxy = _upload(map_coords(tile_xy, m), torch.int32, dev)
        E.stitch_add(pred, logits, mxy, int(m * ph), int(m * pw))
    classes, probs, heat = E.softmax_threshold_argmax(pred, class_probs, mask_dev, 'cls', want_probs)
    return {'logits': logits, 'pred': pred, 'classes': classes, 'probs': probs, 'heatmap': heat}
