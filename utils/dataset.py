"""Drop-in for the sliding-window datasets of the reference's ``utils.dataset``
(/root/reference/utils/dataset.py:83-201): Dataset_wsis, Dataset_wsi, GenerateIterator_wsi.

The tile producer is MI355X-native: the chosen pyramid level is resident in HBM as packed u8 RGB
and batches are produced on the device by wsi_tile_gather (read + ToTensor + Normalize fused), or
- on the fused fast path used by utils.eval - never materialised at all (the stem kernel reads
the slide directly).  Slides come from OpenSlide when it is installed, or from in-memory /
``.npy``/``.npz`` pyramids (`ArraySlide`) otherwise; `open_slide` can be replaced."""
import glob
import os

import numpy as np
import torch

from myargs import args
from utils import preprocessing
from wsi_segmentation_pipeline_amd import engine as E
from wsi_segmentation_pipeline_amd import slide as S


def _open_slide_default(path):
    if isinstance(path, S.ArraySlide):
        return path
    if str(path).endswith('.npz'):
        z = np.load(path)
        return S.ArraySlide([z[k] for k in sorted(z.files) if k.startswith('level')],
                            list(z['downsamples']) if 'downsamples' in z.files else None)
    if str(path).endswith('.npy'):
        return S.ArraySlide([np.load(path, mmap_mode='r')])
    try:
        import openslide
    except ImportError as e:
        raise RuntimeError('cannot open %r: OpenSlide is not installed; pass an ArraySlide or a .npy/.npz pyramid' % (path,)) from e
    return _OpenSlideAdapter(openslide.OpenSlide(path))


class _OpenSlideAdapter:
    """OpenSlide handle + on-demand HBM residency of one level."""

    def __init__(self, scan):
        self.scan = scan
        self.level_dimensions, self.level_downsamples = scan.level_dimensions, scan.level_downsamples
        self.level_count, self.dimensions = scan.level_count, scan.dimensions
        self._dev = {}

    def read_region(self, location, level, size):
        return self.scan.read_region(location, level, size)

    def device_level(self, level, device):
        """One level resident in HBM, filled through the pinned ingestion ring (decode threads || H2D copies || compute)."""
        key = (level, str(device))
        if key not in self._dev:
            from wsi_segmentation_pipeline_amd import ingest
            self._dev[key] = ingest.level_from_slide(self.scan, level, device)
        return self._dev[key]


open_slide = _open_slide_default


class Dataset_wsi:
    """Tile list of one slide at args.scan_level (grid + foreground filter as the reference)."""

    def __init__(self, wsipth, params):
        self.params = params
        self.datalist = []
        self.wsipth = wsipth
        filename = os.path.basename(str(wsipth)) if not isinstance(wsipth, S.ArraySlide) else getattr(wsipth, 'name', 'slide')
        self.scan = open_slide(wsipth)
        if len(self.scan.level_dimensions) - 1 < args.scan_level:
            return                                                        # too few pyramid levels: slide is skipped
        self.params.iw, self.params.ih = self.scan.level_dimensions[args.scan_level]
        from PIL import Image
        msk_pth = '{}/{}.png'.format(args.wsi_mask_pth, filename)
        if os.path.exists(msk_pth):
            mask = np.asarray(Image.open(msk_pth).convert('L'))
        else:
            lvl = min(2, len(self.scan.level_dimensions) - 1)
            thumb = self.scan.read_region((0, 0), lvl, self.scan.level_dimensions[lvl]).convert('RGB')
            mask = preprocessing.find_nuclei(thumb)
            if os.path.isdir(args.wsi_mask_pth):
                Image.fromarray(mask.astype(np.uint8)).save(msk_pth)
        self.mask = mask
        self.image_aug = preprocessing.standard_augmentor(True)
        ref_level = min(2, len(self.scan.level_downsamples) - 1)
        m = self.scan.level_downsamples[args.scan_level] / self.scan.level_downsamples[ref_level]
        self.m = m
        if torch.cuda.is_available():       # enumeration + foreground filter + compaction on the device (wsi_tile_grid)
            grid = S.tile_grid_device(self.params.iw, self.params.ih, self.params.ph, self.params.pw, self.params.sh, self.params.sw,
                                      mask, m).cpu().numpy()
        else:                               # host tooling without a GPU (tile lists only; nothing can be inferred there)
            grid = S.tile_grid(self.params.iw, self.params.ih, self.params.ph, self.params.pw, self.params.sh, self.params.sw,
                               mask, m)
        self.tile_xy = grid
        self.datalist = [(int(x), int(y)) for x, y in grid]

    def __len__(self):
        return len(self.datalist)

    def __getitem__(self, index):
        """(float x, float y, normalised (3,ph,pw) tensor) - host path for single-item access."""
        x, y = self.datalist[index]
        ds = self.scan.level_downsamples[args.scan_level]
        image = self.scan.read_region((int(ds * x), int(ds * y)), args.scan_level, (self.params.pw, self.params.ph)).convert('RGB')
        if args.scan_resize != 1:
            image = image.resize((args.tile_w, args.tile_h))
        return float(x), float(y), self.image_aug(image)


class DeviceTileIterator:
    """Iterable with the DataLoader surface the eval drivers use (``.dataset``, ``len``): yields
    (batch_x float64, batch_y float64, batch_image (B,3,ph,pw) fp32 on the GPU) produced by the
    wsi_tile_gather kernel.  Order is raster order (the reference shuffles; results are
    order-independent, utils/eval.py accumulates sums)."""

    def __init__(self, dataset, batch_size, device=None, span=None):
        self.dataset, self.batch_size = dataset, int(batch_size)
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self._lut = None
        self.span = (0, len(dataset)) if span is None else (int(span[0]), int(span[1]))   # this rank's tiles [lo, hi)

    def __len__(self):
        return (self.span[1] - self.span[0] + self.batch_size - 1) // self.batch_size

    def shard(self, lo, hi):
        """The same producer restricted to tiles [lo, hi) of the raster-order list (one rank's share)."""
        return DeviceTileIterator(self.dataset, self.batch_size, self.device, (lo, hi))

    def __iter__(self):
        ds = self.dataset
        level = ds.scan.device_level(args.scan_level, self.device)
        if self._lut is None:
            self._lut = torch.from_numpy(E.normalize_lut(args.dataset_mean, args.dataset_std)).to(self.device)
        for i in range(self.span[0], self.span[1], self.batch_size):
            xy = ds.tile_xy[i:min(i + self.batch_size, self.span[1])]
            if args.scan_resize != 1:
                # reference :180-181: the (ph, pw) crop is PIL-resized to (tile_h, tile_w) before ToTensor + Normalize
                from wsi_segmentation_pipeline_amd import ingest
                th, tw = int(args.tile_h), int(args.tile_w)
                small = ingest.resize_tiles_bicubic(level, np.ascontiguousarray(xy), (ds.params.ph, ds.params.pw), (th, tw))
                stacked = small.view(-1, tw, 3)
                sxy = np.stack((np.zeros(len(xy), np.int32), np.arange(len(xy), dtype=np.int32) * th), 1)
                img = E.tile_gather(stacked, torch.from_numpy(sxy), th, tw, self._lut)
            else:
                img = E.tile_gather(level, torch.from_numpy(np.ascontiguousarray(xy)), ds.params.ph, ds.params.pw, self._lut)
            yield (torch.from_numpy(xy[:, 0].astype(np.float64)), torch.from_numpy(xy[:, 1].astype(np.float64)), img)


def GenerateIterator_wsi(wsipth, p, bs):
    dataset = Dataset_wsi(wsipth, p)
    return DeviceTileIterator(dataset, bs) if len(dataset) > 0 else None


class Dataset_wsis:
    """All validation slides under ``svs_pth`` (``Case*/*.svs`` like the reference; also
    ``*.npz``/``*.npy`` pyramids), or an explicit ``{name: path-or-ArraySlide}`` mapping."""

    def __init__(self, svs_pth, params, bs=args.batch_size):
        self.params = preprocessing.DotDict(params)
        self.wsis = {}
        if isinstance(svs_pth, dict):
            entries = list(svs_pth.items())
        else:
            paths = sorted(glob.glob('{}/Case*/*.svs'.format(svs_pth)) + glob.glob('{}/*.npz'.format(svs_pth))
                           + glob.glob('{}/*.npy'.format(svs_pth)))
            entries = [(os.path.basename(p), p) for p in paths]
        for filename, src in entries:
            itr = GenerateIterator_wsi(src, preprocessing.DotDict(dict(self.params.__dict__, **self.params)), bs)
            if itr is not None:
                self.wsis[filename] = {'iterator': itr, 'wsipath': src, 'scan': itr.dataset.scan,
                                       'maskpath': '{}/{}.png'.format(args.wsi_mask_pth, filename),
                                       'mask': itr.dataset.mask}
