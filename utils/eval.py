"""Drop-in for the whole-slide evaluation drivers of the reference's ``utils.eval``:
predict_tumorbed (/root/reference/utils/eval.py:155-286) and predict_wsis (:22-152), plus the
region-paint stage that the reference keeps as module-level code in scannet.py:145-155.

Everything between "tile list" and "u8 heat map" stays on the GPU: batches come from the device
tile producer, the encoder/classifier run on the HIP trunk, per-tile logits are accumulated into
the float64 map by wsi_stitch_add and the softmax/threshold/argmax/heat-map is one kernel.  Only the
finished u8 images are copied back to be written as PNG.  There is no CPU fallback.

Scope: the 'cls' path composes with the first-party backbone and is implemented
end to end; predict_tumorbed(mode='seg') needs the third-party smp U-Net decoder (absent, parity-
unpinned) and raises; predict_wsis runs any caller-supplied dense GPU module and does the
accumulate / arg-max on the device."""
import os

import numpy as np
import torch

from myargs import args
from wsi_segmentation_pipeline_amd import engine as E
from wsi_segmentation_pipeline_amd import slide as S


class TrunkEncoder(torch.nn.Module):
    """``model.encoder`` surface over a resnets_shift.ResNet: encoder(x) -> [deepest feature map]
    (the reference indexes ``encoding[0]`` for the 512-channel map, utils/eval.py:196-198)."""

    def __init__(self, resnet):
        super().__init__()
        self.net = resnet
        self.out_shapes = (512, 256, 128, 64, 64)

    def forward(self, x):
        return [self.net.features(x)]


class SlideClassifierModel(torch.nn.Module):
    """First-party composition that predict_tumorbed(mode='cls') drives: ResNet-18 trunk as
    ``encoder`` + models.models.Classifier / Regressor heads."""

    def __init__(self, resnet, classifier, regressor=None):
        super().__init__()
        self.encoder = TrunkEncoder(resnet)
        self.classifier = classifier
        self.regressor = regressor if regressor is not None else torch.nn.Identity()
        self.decoder = torch.nn.Identity()

    def fused_engine(self, device):
        """HIP engine with the classifier's Linear fused behind the average pool."""
        eng = self.encoder.net.hip_engine(device)
        lin = self.classifier.fc[0]
        sig = (lin.weight.data_ptr(), lin.weight._version, lin.bias._version)
        if getattr(self, '_head_sig', None) != (id(eng), sig):
            eng.set_head((lin.weight, lin.bias))
            self._head_sig = (id(eng), sig)
        return eng


def _device_of(model):
    p = next(model.parameters())
    if not p.is_cuda:
        raise RuntimeError('move the model to the GPU first (model.cuda()): the eval drivers run on HIP kernels only')
    return p.device


def _save_png(arr, path):
    from PIL import Image
    Image.fromarray(arr).save(path)


def predict_tumorbed(model, dataset, ep, mode='seg', rank=0, world=1, save=True):
    """Tumour-bed heat maps for every slide in ``dataset`` (a utils.dataset.Dataset_wsis).
    Writes <val_save_pth>/<ep>/<key>_<stride>_heatmap.png and _overlay.png (rank 0) and returns
    {key: {'heatmap': u8 (H2,W2) ndarray, 'classes': u8 ndarray, 'logits': (T,C) tensor}}."""
    if mode != 'cls':
        raise NotImplementedError("mode='seg' needs the segmentation_models_pytorch U-Net decoder, which is third-party "
                                  "and absent here (SURVEY.md 8f rank 1); use mode='cls'")
    out_dir = '{}/{}'.format(args.val_save_pth, ep)
    if save and rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    dev = _device_of(model)
    was_training = model.training
    model.eval()
    results = {}
    with torch.no_grad():
        for key in list(dataset.wsis):
            entry = dataset.wsis[key]
            it = entry['iterator']
            ds, scan = it.dataset, entry['scan']
            ref_level = min(2, len(scan.level_dimensions) - 1)
            map_hw = scan.level_dimensions[ref_level][::-1]
            m = scan.level_downsamples[args.scan_level] / scan.level_downsamples[ref_level]
            mask = torch.from_numpy(np.ascontiguousarray(entry['mask'])).to(dev) if entry.get('mask') is not None else None
            if isinstance(model, SlideClassifierModel):
                # fused fast path: the stem kernel reads the HBM-resident slide directly
                level = scan.device_level(args.scan_level, dev)
                r = S.infer_slide_cls(model.fused_engine(dev), level, ds.tile_xy, ds.params.ph, ds.params.pw, m, map_hw,
                                      args.num_classes, args.class_probs, mask, rank, world, want_probs=False)
            else:
                # generic loop over the iterator API (any model exposing .encoder/.classifier on the GPU)
                pred = torch.zeros((args.num_classes,) + tuple(map_hw), dtype=torch.float64, device=dev)
                all_logits = []
                for batch_x, batch_y, batch_image in it:
                    logits = model.classifier(model.encoder(batch_image.to(dev))[0])
                    xy = np.stack((batch_x.numpy(), batch_y.numpy()), 1)
                    E.stitch_add(pred, logits, torch.from_numpy(S.map_coords(xy, m)), int(m * ds.params.ph), int(m * ds.params.pw))
                    all_logits.append(logits)
                classes, _, heat = E.softmax_threshold_argmax(pred, args.class_probs, mask, 'cls', want_probs=False)
                r = {'logits': torch.cat(all_logits), 'pred': pred, 'classes': classes, 'heatmap': heat}
            heat = r['heatmap'].cpu().numpy()
            results[key] = {'heatmap': heat, 'classes': r['classes'].cpu().numpy(), 'logits': r['logits']}
            if save and rank == 0:
                _save_png(heat, '{}/{}_{}_heatmap.png'.format(out_dir, key, args.tile_stride_w))
                thumb = np.asarray(scan.read_region((0, 0), ref_level, scan.level_dimensions[ref_level]).convert('RGB'))
                over = thumb * 0.75 + 255 * np.repeat((heat > 255 * 0.99)[..., None], 3, -1) * 0.25
                _save_png(np.uint8(over), '{}/{}_{}_overlay.png'.format(out_dir, key, args.tile_stride_w))
            dataset.wsis[key] = None                       # the reference frees each slide after use (:282)
    if was_training:
        model.train()
    return results


def predict_wsis(model, dataset, ep, save=True):
    """Dense per-pixel map prediction (reference utils/eval.py:22-152): `model(batch_image)` must return
    (B, C, ph, pw) logits on the GPU (the reference drives a third-party smp model here; any GPU module
    works).  The accumulate `pred[:, y:y+ph, x:x+pw] += pred_src[b]` (:58-60) runs on the device in
    float64 at scan-level resolution (wsi_stitch_add_dense) - 51 GB for a 40k x 40k slide, which the
    host-side reference cannot hold but HBM can - followed by the class arg-max (wsi_softmax_threshold_argmax).
    Returns {key: {'pred': float64 (C,H,W) tensor, 'classes': u8 (H,W) tensor}} and writes
    <val_save_pth>/<ep>/<key>_<stride>.png (class colours on the foreground mask).  The reference's
    score printing and tumour-bed outline (cv2 / skimage / mahotas post-processing)
    are not reproduced."""
    out_dir = '{}/{}'.format(args.val_save_pth, ep)
    if save:
        os.makedirs(out_dir, exist_ok=True)
    dev = _device_of(model)
    was_training = model.training
    model.eval()
    results = {}
    with torch.no_grad():
        for key in list(dataset.wsis):
            entry = dataset.wsis[key]
            it = entry['iterator']
            ds, scan = it.dataset, entry['scan']
            iw, ih = scan.level_dimensions[args.scan_level]
            pred = torch.zeros((args.num_classes, ih, iw), dtype=torch.float64, device=dev)
            for batch_x, batch_y, batch_image in it:
                pred_src = model(batch_image.to(dev))
                if pred_src.dim() != 4 or pred_src.shape[1] != args.num_classes:
                    raise ValueError('predict_wsis needs a dense model returning (B, %d, ph, pw)' % args.num_classes)
                xy = torch.stack((batch_x, batch_y), 1).to(torch.int32)        # int(batch_x[bj]): truncation
                E.stitch_add_dense(pred, pred_src, xy)
            classes, _, _ = E.softmax_threshold_argmax(pred, [0.0] * args.num_classes, want_probs=False)
            results[key] = {'pred': pred, 'classes': classes}
            if save:
                cls = classes.cpu().numpy()
                mask = entry.get('mask')
                rgb = np.zeros(cls.shape + (3,), np.uint8)
                for cj in range(min(args.num_classes - 1, 3)):                  # class k>0 -> channel k-1, like pred_to_mask
                    rgb[cls == cj + 1, cj] = 255
                if mask is not None and tuple(mask.shape) != cls.shape:
                    ys = (np.arange(cls.shape[0]) * mask.shape[0] // cls.shape[0]).clip(0, mask.shape[0] - 1)
                    xs = (np.arange(cls.shape[1]) * mask.shape[1] // cls.shape[1]).clip(0, mask.shape[1] - 1)
                    mask = np.asarray(mask)[ys][:, xs]
                if mask is not None:
                    rgb = rgb * (np.asarray(mask) > 0)[..., None].astype(np.uint8)
                _save_png(rgb, '{}/{}_{}.png'.format(out_dir, key, args.tile_stride_w))
    if was_training:
        model.train()
    return results


def predict_regions(model, iterator, metadata, label_shape, class_probs=None):
    """Region-proposal evaluation stage (reference scannet.py:145-155 / slic.py:93-99, with the
    documented fix: the ensemble logits are soft-maxed over the class axis): bags -> ResNet bag
    forward on the HIP path -> class per region -> painted label image (int64 ndarray)."""
    class_probs = args.class_probs if class_probs is None else class_probs
    dev = _device_of(model)
    was_training = model.training
    model.eval()
    pred_mask = np.zeros(label_shape, dtype=np.int64)
    with torch.no_grad():
        for images, tile_ids in iterator:
            _, ensemble = model(images.to(dev))
            # (B,C) logits as a (C,B,1) "map": softmax over classes / threshold / argmax in one HIP kernel
            as_map = ensemble.t().to(torch.float64).contiguous().view(ensemble.shape[1], -1, 1)
            cls = E.softmax_threshold_argmax(as_map, class_probs, want_probs=False)[0].view(-1).cpu().numpy()
            for tj, tile_id in enumerate(tile_ids.numpy()):
                pred_mask[metadata[int(tile_id)]['foreground_indices']] = cls[tj]
    if was_training:
        model.train()
    return pred_mask
