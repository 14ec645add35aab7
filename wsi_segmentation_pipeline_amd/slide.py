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
            if isinstance(l, torch.Tensor):
                t = l.to(device)[..., :3]
            elif l.nbytes >= (32 << 20):                       # host arrays / memmaps: pinned ring, copies overlap the band reads
                from . import ingest
                t = ingest.level_from_array(l, device)
            else:
                t = torch.from_numpy(np.ascontiguousarray(l[..., :3])).to(device)
            self._dev[key] = t.contiguous()                    # always (H, W, 3): an alpha plane is dropped on every path
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


def tile_grid_device(iw, ih, ph, pw, sh, sw, mask=None, m=1.0, thresh=0.05, device=None):
    """tile_grid on the device (wsi_tile_grid): candidate enumeration, mask-window foreground test and order-preserving
    compaction in HIP kernels; `mask` is a (MH,MW) uint8 GPU tensor (or ndarray, uploaded) or None.  Returns a (T,2) int32 GPU
    tensor equal to tile_grid(...)."""
    import ctypes as C
    from . import native
    from .engine import _ptr, _stream
    lib = native.load()
    dev = torch.device(device) if device is not None else (mask.device if isinstance(mask, torch.Tensor) else torch.device('cuda', torch.cuda.current_device()))
    if dev.type != 'cuda':
        raise RuntimeError('tile_grid_device runs on the GPU')
    n = lib.wsi_tile_grid_candidates(iw, ih, ph, pw, sh, sw)
    if n < 0:
        raise ValueError('bad strides')
    if n == 0:
        return torch.zeros((0, 2), dtype=torch.int32, device=dev)
    mk = None
    if mask is not None:
        mk = (mask if isinstance(mask, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mask))).to(dev)
        mk = mk != 0
        if mk.dim() == 3:                                      # colour mask images: nonzero in any channel (tile_grid's .any(-1))
            mk = mk.any(-1)
        if mk.dim() != 2:
            raise ValueError('mask must be (MH, MW) or (MH, MW, channels), got %s' % (tuple(mk.shape),))
        mk = mk.to(torch.uint8).contiguous()
    out = torch.empty((n, 2), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(lib.wsi_tile_grid_scratch_bytes(n), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        native.check(lib.wsi_tile_grid(iw, ih, ph, pw, sh, sw, _ptr(mk) if mk is not None else None, mk.shape[0] if mk is not None else 0,
                                       mk.shape[1] if mk is not None else 0, float(m), float(thresh), _ptr(out), _ptr(count), _ptr(scratch),
                                       _stream()), 'wsi_tile_grid')
    return out[:int(count.item())]


def map_coords(tile_xy, m):
    """int(m * x), int(m * y) in float64, as reference utils/eval.py:214."""
    return np.floor(np.asarray(tile_xy, np.float64) * m).astype(np.int32)


# ------------------------------------------------------------------------------ sharding
def shard_range(total, rank, world):
    """Contiguous raster-order chunk [lo, hi) of the tile list owned by `rank`."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _collective_forced():
    """WSI_FORCE_COLLECTIVE=1: issue the collective even for world == 1 (exercises the RCCL path on a one-GPU box)."""
    import os
    import torch.distributed as dist
    return os.environ.get('WSI_FORCE_COLLECTIVE') == '1' and dist.is_available() and dist.is_initialized()


# Every collective of the path goes through ONE code path: the op list, its order, the buffer shapes and the packing are the same
# for RCCL ("nccl": device buffers over xGMI) and gloo (CPU tests, or several ranks sharing one GPU); the ONLY difference is where
# the wire buffer lives, and that is `_Wire.put` / `_Wire.get`.  r01-r04 wrote the two cases as separate branches, so the RCCL
# branches - never run on more than one device - shared few lines with the tested ones.  `OP_LOG` (a list, or None) records
# (op, shape, dtype) of every collective issued; tests/test_host_logic.py asserts the sequence.
OP_LOG = None


class _Wire:
    def __init__(self, dev):
        import torch.distributed as dist
        self.dev = torch.device(dev)
        self.nccl = dist.get_backend() == 'nccl'
        self.on = self.dev if (self.nccl or self.dev.type != 'cuda') else torch.device('cpu')

    def put(self, t):                                           # compute device -> wire buffer (identity under RCCL)
        return t.to(self.on)

    def get(self, t):                                           # wire buffer -> compute device (identity under RCCL)
        return t.to(self.dev)

    def new(self, shape, dtype, fill=None):
        return torch.empty(shape, dtype=dtype, device=self.on) if fill is None else torch.full(shape, fill, dtype=dtype, device=self.on)

    @staticmethod
    def log(op, t):
        if OP_LOG is not None:
            OP_LOG.append((op, tuple(t.shape), str(t.dtype).replace('torch.', '')))

    def all_gather(self, out, buf):
        import torch.distributed as dist
        self.log('all_gather', buf)
        dist.all_gather_into_tensor(out, buf)
        return out

    def all_reduce(self, t, op):
        import torch.distributed as dist
        self.log('all_reduce_' + str(op).split('.')[-1].lower(), t)
        dist.all_reduce(t, op=op)
        return t

    def broadcast(self, t, src):
        import torch.distributed as dist
        self.log('broadcast', t)
        dist.broadcast(t, src)
        return t

    def p2p(self, ops):
        """ops: [('send' | 'recv', buffer, peer)] posted together (batch_isend_irecv: concurrent xGMI links under RCCL)."""
        import torch.distributed as dist
        if not ops:
            return
        for kind, t, _ in ops:
            self.log(kind, t)
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend if kind == 'send' else dist.irecv, t, peer) for kind, t, peer in ops]):
            q.wait()


def gather_tile_logits(local_logits, total, rank, world):
    """Concatenate per-rank logits in rank order on every rank.  One RCCL all-gather of equal-size
    padded chunks (payload = total x C fp32: latency-bound, a single collective)."""
    if world == 1 and not _collective_forced():
        return local_logits
    import torch.distributed as dist
    c = local_logits.shape[1]
    chunk = (total + world - 1) // world
    wire = _Wire(local_logits.device)
    buf = wire.new((chunk, c), local_logits.dtype, 0)
    buf[:local_logits.shape[0]] = wire.put(local_logits)
    out = wire.get(wire.all_gather(wire.new((world * chunk, c), local_logits.dtype), buf))
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(out[r * chunk:r * chunk + (hi - lo)])
    return torch.cat(parts, 0)


def allreduce_max(value, dev, world):
    """Maximum of a host scalar over the ranks (world == 1: the value itself)."""
    if world == 1 and not _collective_forced():
        return float(value)
    import torch.distributed as dist
    wire = _Wire(dev)
    t = wire.put(torch.tensor([float(value)], dtype=torch.float64))
    return float(wire.all_reduce(t, dist.ReduceOp.MAX).item())


def allreduce_span(span, dev):
    """Exponent span (2 int32: smallest, largest biased exponent; None = this rank added nothing) over the ranks."""
    import torch.distributed as dist
    wire = _Wire(dev)
    lo = wire.put(span[0:1] if span is not None else torch.full((1,), 255, dtype=torch.int32))
    hi = wire.put(span[1:2] if span is not None else torch.zeros(1, dtype=torch.int32))
    wire.all_reduce(lo, dist.ReduceOp.MIN)
    wire.all_reduce(hi, dist.ReduceOp.MAX)
    return wire.get(torch.cat((lo, hi)))


def allreduce_map(pred):
    """Sum a float64 prediction map over the ranks (the r02 form of the dense 'seg' exchange; gather_map_bands replaced it).
    Float64 sums of the fp32 addends are exact inside the stitch's exponent-span bound, so the result does not depend on the
    reduction order."""
    import torch.distributed as dist
    wire = _Wire(pred.device)
    return wire.get(wire.all_reduce(wire.put(pred), dist.ReduceOp.SUM))


def tile_bands(txy, dy, dx, hw):
    """Disjoint rectangles (y0, y1, x0, x1) of a (H, W) map covering everything a set of tiles touches: `txy` (T, 2) map
    coordinates (x, y) of footprints dy x dx.  Per map row the touched column extent is the hull of the tiles on that row; runs
    of rows with the same extent form one rectangle.  A contiguous raster share of a tile grid gives at most a handful (first
    partial tile row, full rows, last partial row, the edge column, the bottom row)."""
    H, W = int(hw[0]), int(hw[1])
    txy = np.asarray(txy, np.int64).reshape(-1, 2)
    lo, hi = np.full(H, W, np.int64), np.zeros(H, np.int64)
    for y in np.unique(txy[:, 1]) if len(txy) else ():
        xs = txy[txy[:, 1] == y, 0]
        a, b = max(int(y), 0), min(int(y) + dy, H)
        lo[a:b] = np.minimum(lo[a:b], max(int(xs.min()), 0))
        hi[a:b] = np.maximum(hi[a:b], min(int(xs.max()) + dx, W))
    rects, y = [], 0
    while y < H:
        if hi[y] <= lo[y]:
            y += 1
            continue
        e = y + 1
        while e < H and lo[e] == lo[y] and hi[e] == hi[y]:
            e += 1
        rects.append((y, e, int(lo[y]), int(hi[y])))
        y = e
    return np.asarray(rects, np.int64).reshape(-1, 4)


def gather_map_bands(pred, txy, dy, dx, rank, world, dst=0):
    """Dense 'seg' mode exchange ("gather band shards of the map ... no all-reduce on this path"): every rank
    stitched a contiguous raster share of the tiles into its own float64 (C, H, W) map; each sends ONLY the rectangles its tiles
    touch (`tile_bands`), packed into one buffer, straight to `dst`, which posts all its receives at once (7 concurrent xGMI
    links under RCCL) and adds them into its map.  Neighbouring shares overlap by a few rows when stride < tile, hence a sum,
    not a copy - exact in float64 inside the stitch's exponent-span bound, so `dst` ends with the single-rank map bit for bit.
    Bytes: each rank sends its band (~1/world of the map, 200 MB / 8 at 4 x 2500^2) once; the r02 all-reduce moved the whole map
    through every rank twice.  Returns `pred`: complete on `dst`, this rank's partial map elsewhere."""
    if world == 1:
        return pred
    C = pred.shape[0]
    rects = tile_bands(txy, dy, dx, pred.shape[1:])
    wire = _Wire(pred.device)
    # rectangle tables of every rank (a few dozen bytes): count first, then the padded tables
    cnts = wire.all_gather(wire.new((world,), torch.int64), wire.put(torch.tensor([len(rects)], dtype=torch.int64))).cpu().tolist()
    kmax = max(max(cnts), 1)
    tab = torch.zeros((kmax, 4), dtype=torch.int64)
    if len(rects):
        tab[:len(rects)] = torch.from_numpy(rects)
    tabs = wire.all_gather(wire.new((world * kmax, 4), torch.int64), wire.put(tab)).view(world, kmax, 4).cpu().numpy()
    numel = lambda r: int(sum(C * (t[1] - t[0]) * (t[3] - t[2]) for t in tabs[r, :cnts[r]]))
    if rank != dst:
        if len(rects):
            buf = wire.put(torch.cat([pred[:, y0:y1, x0:x1].reshape(-1) for y0, y1, x0, x1 in rects.tolist()]))
            wire.p2p([('send', buf, dst)])
        return pred
    srcs = [r for r in range(world) if r != dst and cnts[r]]
    bufs = {r: wire.new((numel(r),), pred.dtype) for r in srcs}
    wire.p2p([('recv', bufs[r], r) for r in srcs])
    for r in srcs:
        b, off = wire.get(bufs[r]), 0
        for y0, y1, x0, x1 in tabs[r, :cnts[r]].tolist():
            n = C * (y1 - y0) * (x1 - x0)
            pred[:, y0:y1, x0:x1] += b[off:off + n].view(C, y1 - y0, x1 - x0)
            off += n
    return pred


def broadcast_from(t, src=0):
    """Broadcast a tensor (here: the u8 class / heat maps, 6 MB each at 2500^2) from `src`."""
    wire = _Wire(t.device)
    return wire.get(wire.broadcast(wire.put(t), src))


# ------------------------------------------------------------------------------ rank-resident slide regions
class SyntheticRows:
    """Position-deterministic i.i.d. uniform u8 RGB slide level: rows are generated on the
    device in blocks of `block` rows, block b from seed (seed, b), so ANY rank can produce ANY rectangle of the same
    slide without holding the rest of it (40 000^2 x 3 = 4.8 GB when fully resident)."""

    def __init__(self, iw, ih, seed, device, block=64):
        self.iw, self.ih, self.seed, self.device, self.block = int(iw), int(ih), int(seed), torch.device(device), int(block)

    def _block(self, b):
        g = torch.Generator(device=self.device).manual_seed(self.seed * 1000003 + b)
        rows = min(self.block, self.ih - b * self.block)
        return torch.randint(0, 256, (rows, self.iw, 3), dtype=torch.uint8, device=self.device, generator=g)

    def rect(self, x0, y0, x1, y1):
        """(y1-y0, x1-x0, 3) uint8 tensor of slide[y0:y1, x0:x1] (bounds inside the slide)."""
        out = torch.empty((y1 - y0, x1 - x0, 3), dtype=torch.uint8, device=self.device)
        for b in range(y0 // self.block, (y1 - 1) // self.block + 1):
            blk = self._block(b)
            a, e = max(y0, b * self.block), min(y1, b * self.block + blk.shape[0])
            out[a - y0:e - y0] = blk[a - b * self.block:e - b * self.block, x0:x1]
        return out

    def full(self):
        return self.rect(0, 0, self.iw, self.ih)


def region_plan(tile_xy, pw, ph):
    """Cover a tile list by few rectangles and pack them into one atlas, so a rank keeps only the parts of the slide
    its own tiles touch ("each GPU holds only its band").  Tile rows (same y) at least half as wide as
    the widest one are merged vertically when they touch or overlap and have similar x-extents: a rank's interior
    rows become one full-width band.  Narrow rows (a partial first / last row, the rank's share of the right-edge
    column) stay single-row rectangles and are shelf-packed side by side under the bands.  Returns (rects, (atlas_h,
    atlas_w), local_xy): rects = [(x0, y0, x1, y1, atlas_x, atlas_y)] in slide pixels, local_xy (T,2) int32 = each
    tile's corner inside the atlas."""
    xy = np.asarray(tile_xy, np.int64).reshape(-1, 2)
    if len(xy) == 0:
        return [], (1, max(1, pw)), np.zeros((0, 2), np.int32)
    ys = np.unique(xy[:, 1])
    ext = {}
    for y in ys:
        sel = xy[:, 1] == y
        ext[int(y)] = (int(xy[sel, 0].min()), int(xy[sel, 0].max()) + pw)
    wmax = max(x1 - x0 for x0, x1 in ext.values())
    bands, narrow, row_rect = [], [], {}               # bands: [x0, y0, x1, y1]; row_rect: y -> ('b' | 'n', index)
    for y in (int(v) for v in ys):
        x0, x1 = ext[y]
        if (x1 - x0) * 2 < wmax:
            narrow.append([x0, y, x1, y + ph])
            row_rect[y] = ('n', len(narrow) - 1)
            continue
        hit = None
        for i, r in enumerate(bands):
            inter = min(x1, r[2]) - max(x0, r[0])
            union = max(x1, r[2]) - min(x0, r[0])
            if y <= r[3] and inter * 2 >= union:       # touches / overlaps the band above it, similar extent
                hit = i
                break
        if hit is None:
            bands.append([x0, y, x1, y + ph])
            hit = len(bands) - 1
        else:
            r = bands[hit]
            r[0], r[2], r[3] = min(r[0], x0), max(r[2], x1), max(r[3], y + ph)
        row_rect[y] = ('b', hit)
    aw = max(r[2] - r[0] for r in bands + narrow)
    ay, placed_b, placed_n = 0, [], []
    for r in bands:
        placed_b.append((r[0], r[1], r[2], r[3], 0, ay))
        ay += r[3] - r[1]
    ax = 0
    for r in narrow:                                   # shelves of height ph
        w = r[2] - r[0]
        if ax + w > aw:
            ax, ay = 0, ay + ph
        placed_n.append((r[0], r[1], r[2], r[3], ax, ay))
        ax += w
    if narrow:
        ay += ph
    local = np.empty((len(xy), 2), np.int32)
    for i, (x, y) in enumerate(xy):
        kind, j = row_rect[int(y)]
        r = (placed_b if kind == 'b' else placed_n)[j]
        local[i] = (x - r[0] + r[4], y - r[1] + r[5])
    return placed_b + placed_n, (ay, aw), local


def resident_regions(source, rects, atlas_hw, device):
    """Materialise the atlas of `region_plan`: a zero-filled (H, W, 3) uint8 device tensor holding every rectangle
    (clipped to the slide; what hangs outside stays 0, like OpenSlide's transparent black).  `source` has
    .iw, .ih and .rect(x0, y0, x1, y1) -> uint8 (h, w, 3) tensor or array."""
    atlas = torch.zeros((atlas_hw[0], atlas_hw[1], 3), dtype=torch.uint8, device=device)
    for x0, y0, x1, y1, ax, ay in rects:
        cx0, cy0, cx1, cy1 = max(x0, 0), max(y0, 0), min(x1, source.iw), min(y1, source.ih)
        if cx1 > cx0 and cy1 > cy0:
            blk = source.rect(cx0, cy0, cx1, cy1)
            blk = blk if isinstance(blk, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(blk))
            atlas[ay + cy0 - y0:ay + cy1 - y0, ax + cx0 - x0:ax + cx1 - x0] = blk.to(device)
    return atlas


# ------------------------------------------------------------------------------ per-slide pipeline
def _upload(arr, dtype, dev):
    """Host array -> device without blocking the host on the GPU queue: a pageable-memory copy waits for everything already
    enqueued, so the host could not start enqueuing the next slide while this one computes (1.7 ms bubble per 10 k-tile slide)."""
    t = torch.as_tensor(np.ascontiguousarray(arr), dtype=dtype)
    if dev.type == 'cuda':
        return t.pin_memory().to(dev, non_blocking=True)
    return t.to(dev)


def infer_slide_cls(eng, slide_level_dev, tile_xy, ph, pw, m, map_hw, num_classes, class_probs, mask_dev=None,
                    rank=0, world=1, want_probs=True, local_xy=None):
    """predict_tumorbed(mode='cls') for one slide on device (reference utils/eval.py:182-229):
    this rank's tiles -> fused read+transform+trunk+classifier -> (RCCL gather) -> float64 stitch ->
    softmax/threshold/argmax -> u8 heat map.  Returns dict of device tensors.
    `slide_level_dev` is the whole level, or - with `local_xy` - this rank's atlas of resident regions
    (region_plan / resident_regions) and local_xy the corners of ITS tiles [lo, hi) inside that atlas."""
    T = int(tile_xy.shape[0])
    lo, hi = shard_range(T, rank, world)
    dev = slide_level_dev.device
    if local_xy is not None and len(local_xy) != hi - lo:
        raise ValueError('local_xy must list this rank\'s %d tiles' % (hi - lo))
    xy_dev = _upload(tile_xy[lo:hi] if local_xy is None else local_xy, torch.int32, dev)
    precision = None
    fwd_kw = {}
    if hasattr(eng, 'forward_tiles_verified') and getattr(eng, 'head_k', 0):
        # precision='auto', r05: mx first, verified afterwards on the stratified sample (AutoTrunkEngine.forward_tiles_verified) - ONE
        # decision per slide for every rank (the sample errors are max-reduced before the decision; ranks without tiles take part)
        _, logits, _ = eng.forward_tiles_verified(slide_level_dev, xy_dev, ph, pw, reduce_max=lambda e: allreduce_max(e, dev, world))
        precision = dict(eng.report)
    elif hasattr(eng, 'probe_tiles'):
        # precision='auto': ONE mode per slide for every rank - stratified probe of this rank's shard, maximum over the ranks.
        # The probe, the decision and the forward are tied together by an explicit slide id (a token of this call), not by the
        # identity of the tensor object: a caller's fresh view of the same level (`level[...]`, `.contiguous()`) must not make
        # forward_tiles probe again locally and override the all-reduced decision on some ranks only (r04 advisor finding)
        fwd_kw['slide_id'] = object()
        eng.decide(allreduce_max(eng.probe_tiles(slide_level_dev, xy_dev, ph, pw, **fwd_kw), dev, world))
        precision = dict(eng.report)
    if precision is not None and 'order' in precision:
        pass                                                  # (forward_tiles_verified ran the shard)
    elif hi > lo:
        _, logits, _ = eng.forward_tiles(slide_level_dev, xy_dev, ph, pw, logits=True, **fwd_kw)
    else:
        logits = torch.zeros((0, num_classes), dtype=torch.float32, device=dev)
    logits = gather_tile_logits(logits, T, rank, world)
    pred = torch.zeros((num_classes, map_hw[0], map_hw[1]), dtype=torch.float64, device=dev)
    span = None
    if T:
        mxy = _upload(map_coords(tile_xy, m), torch.int32, dev)
        E.stitch_add(pred, logits, mxy, int(m * ph), int(m * pw))
        span = E.exponent_span(logits)                    # guard of the float64 atomics (checked by the caller: no sync here)
    classes, probs, heat = E.softmax_threshold_argmax(pred, class_probs, mask_dev, 'cls', want_probs)
    return {'logits': logits, 'pred': pred, 'classes': classes, 'probs': probs, 'heatmap': heat, 'exponent_span': span,
            'precision': precision}
