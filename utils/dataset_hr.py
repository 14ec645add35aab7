"""Drop-in for the region-bag evaluation dataset of the reference's ``utils.dataset_hr``
(/root/reference/utils/dataset_hr.py:14-18 constants, :218-306 Dataset_eval / GenerateIterator_eval).

A region contributes one bag of 16 crops (first 8 perimeter points, then first 8 centre points),
64x64 pixels read at pyramid level 1.  Bags are produced on the device: level 1 is resident in HBM
and the crop + ToTensor + Normalize run in the wsi_tile_gather kernel."""
import numpy as np
import torch

from myargs import args
from utils import preprocessing, regiontools
from wsi_segmentation_pipeline_amd import engine as E

HR_NUM_CNT_SAMPLES = 8
HR_NUM_PERIM_SAMPLES = 8
HR_SCAN_LEVEL = 1
HR_PATCH_W = 64
HR_PATCH_H = 64


class Dataset_eval:
    def __init__(self, metadata, eval=True, remove_white=False, scan=None):
        if remove_white:
            raise NotImplementedError('remove_white=True is unused by the reference eval scripts')
        from utils import dataset as ds
        first = list(metadata.keys())[0]
        self.scan = scan if scan is not None else ds.open_slide(metadata[first]['wsipath'])
        params = preprocessing.DotDict({'iw': self.scan.level_dimensions[0][0], 'ih': self.scan.level_dimensions[0][1],
                                        'tile_w': HR_PATCH_W, 'tile_h': HR_PATCH_H,
                                        'scan_level': metadata[first]['scan_level']})
        self.datalist = []
        for key in metadata:
            region = dict(metadata[key])
            region['cnt_xy'], n_cnt = regiontools.map_points(region['cnt_xy'], params)
            region['perim_xy'], n_per = regiontools.map_points(region['perim_xy'], params)
            if n_cnt >= HR_NUM_CNT_SAMPLES and n_per >= HR_NUM_PERIM_SAMPLES:
                self.datalist.append(region)
        self.eval = eval
        self.image_aug = preprocessing.standard_augmentor(True)

    def __len__(self):
        return len(self.datalist)

    def corners(self, index):
        """(16,2) int64 level-0 top-left corners: perimeter points first, then centre points."""
        r = self.datalist[index]
        return np.vstack((r['perim_xy'][:HR_NUM_PERIM_SAMPLES], r['cnt_xy'][:HR_NUM_CNT_SAMPLES])).astype(np.int64)

    def __getitem__(self, index):
        """Host path for single-item access: ((16,3,64,64) tensor, tile_id)."""
        images = [self.image_aug(self.scan.read_region((int(x), int(y)), HR_SCAN_LEVEL, (HR_PATCH_W, HR_PATCH_H)).convert('RGB'))
                  for x, y in self.corners(index)]
        return torch.stack(images, 0), self.datalist[index]['tile_id']


class DeviceBagIterator:
    """Yields ((B,16,3,64,64) fp32 GPU bags, tile_ids int64 tensor), raster (metadata) order."""

    def __init__(self, dataset, batch_size, device=None, indices=None):
        self.dataset, self.batch_size = dataset, int(batch_size)
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.indices = None if indices is None else [int(i) for i in indices]     # this rank's shard of the bags (utils.eval.predict_regions)

    def __len__(self):
        n = len(self.dataset) if self.indices is None else len(self.indices)
        return (n + self.batch_size - 1) // self.batch_size

    def shard(self, indices):
        """The same producer restricted to a subset of the bags (metadata order kept)."""
        return DeviceBagIterator(self.dataset, self.batch_size, self.device, indices)

    def __iter__(self):
        ds = self.dataset
        level = ds.scan.device_level(HR_SCAN_LEVEL, self.device)
        down = ds.scan.level_downsamples[HR_SCAN_LEVEL]
        lut = torch.from_numpy(E.normalize_lut(args.dataset_mean, args.dataset_std)).to(self.device)
        order = list(range(len(ds))) if self.indices is None else self.indices
        for i in range(0, len(order), self.batch_size):
            idx = order[i:i + self.batch_size]
            xy = np.concatenate([ds.corners(j) for j in idx])                       # level-0 corners
            lxy = np.floor_divide(xy, int(down)).astype(np.int32) if float(down).is_integer() \
                else np.floor(xy / down).astype(np.int32)
            img = E.tile_gather(level, torch.from_numpy(lxy), HR_PATCH_H, HR_PATCH_W, lut)
            yield (img.view(len(idx), HR_NUM_PERIM_SAMPLES + HR_NUM_CNT_SAMPLES, 3, HR_PATCH_H, HR_PATCH_W),
                   torch.tensor([ds.datalist[j]['tile_id'] for j in idx], dtype=torch.int64))


def GenerateIterator_eval(metadata, eval=True, remove_white=False, scan=None):
    return DeviceBagIterator(Dataset_eval(metadata, eval, remove_white, scan), args.batch_size)
