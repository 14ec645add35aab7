"""Drop-in for the whole-slide evaluation drivers of the reference's ``utils.eval``:
predict_tumorbed (/root/reference/utils/eval.py:155-286) and predict_wsis (:22-152), plus the
region-paint stage that the reference keeps as module-level code in scannet.py:145-155.

Everything between "tile list" and "u8 heat map" stays on the GPU: batches come from the device
tile producer, the encoder/classifier run on the HIP trunk, per-tile logits are accumulated into
the float64 map by wsi_stitch_add and the softmax/threshold/argmax/heat-map is one kernel.  Only the
finished u8 images are copied back to be written as PNG.  There is no CPU fallback.

Scope: the 'cls' path composes with the first-party backbone and is implemented
end to end; predict_tumorbed(mode='seg') (the reference's default) drives ``model.decoder(model.encoder(x))`` of any GPU
model - e.g. wsi_segmentation_pipeline_amd.unet.UNetSeg, the restatement of the third-party smp U-Net the reference
uses, whose accuracy is therefore pinned to this repo's own specification of smp 0.0.x, not to the reference (absent
package, no fixtures: "parity unpinned"); predict_wsis runs any caller-supplied dense GPU module and does the
accumulate / resize / arg-max / tumour-bed post-process on the device (post-process oracles: parity unpinned too)."""
import os

import numpy as np
import torch

from myargs import args
from wsi_segmentation_pipeline_amd import engine as E
from wsi_segmentation_pipeline_amd import slide as S


class TrunkEncoder(torch.nn.Module):
    """``model.encoder`` surface over a resnets_shift.ResNet: encoder(x) -> [deepest feature map]
    (the reference indexes ``encoding[0]`` for the 512-channel map, utils/eval.py:196-198).  The full five-map encoder of
    the dense 'seg' model is wsi_segmentation_pipeline_amd.unet.UNetEncoder."""

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


def _to_host(*tensors):
    """Device tensors -> NumPy arrays: asynchronous copies into pinned host buffers, ONE synchronisation for all of them."""
    outs = []
    for t in tensors:
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=t.is_cuda)
        h.copy_(t, non_blocking=True)
        outs.append(h)
    if any(t.is_cuda for t in tensors):
        torch.cuda.current_stream().synchronize()
    return tuple(h.numpy() for h in outs)


def _save_png(arr, path):
    from PIL import Image
    Image.fromarray(arr).save(path)


def _dense_batches(model, it, scan, dev, forward):
    """(xy (B,2) int ndarray, logits (B,C,ph,pw)) per batch of a dense (per-pixel) model over a slide's tile iterator.
    Fused fast path (r05) when the model is this repo's UNetSeg and the tiles are read at the scan level's own resolution: the
    U-Net engine reads the tiles straight from the HBM-resident level (integer stem on the u8 pixels, no normalised fp32 batch,
    no fp32 round trip of the five encoder maps between `encoder` and `decoder`) in ITS batch size - the reference's
    `batch_size` flag (30 tiles, myargs.py) leaves the chip mostly idle.  Any other model, a resized scan or a host iterator:
    `forward(batch_image)` over the iterator, as the reference writes it (utils/eval.py:52-60, :196-215)."""
    from wsi_segmentation_pipeline_amd.unet import UNetSeg
    ds = getattr(it, 'dataset', None)
    if isinstance(model, UNetSeg) and args.scan_resize == 1 and hasattr(it, 'span') and hasattr(scan, 'device_level'):
        eng = model.hip_engine(dev)
        level = scan.device_level(args.scan_level, dev)
        lo, hi = it.span
        ph, pw = ds.params.ph, ds.params.pw
        nb = max(1, int(np.ceil((hi - lo) / eng._batch(ph, pw) - 0.25)))     # (a quarter over the tuned batch still runs as ONE batch)
        mb = -(-(hi - lo) // nb)                             # equal batches: no short last batch
        xy_dev = S._upload(ds.tile_xy[lo:hi], torch.int32, dev)     # one asynchronous upload: the host keeps enqueuing batches ahead of the GPU
        for i in range(0, hi - lo, mb):
            yield np.ascontiguousarray(ds.tile_xy[lo + i:min(lo + i + mb, hi)]), eng.forward_tiles(level, xy_dev[i:i + mb], ph, pw)
        return
    for batch_x, batch_y, batch_image in it:
        yield np.stack((batch_x.numpy(), batch_y.numpy()), 1), forward(batch_image.to(dev))


def predict_tumorbed(model, dataset, ep, mode='seg', rank=0, world=1, save=True):
    """Tumour-bed heat maps for every slide in ``dataset`` (a utils.dataset.Dataset_wsis).
    Writes <val_save_pth>/<ep>/<key>_<stride>_heatmap.png and _overlay.png (rank 0) and returns
    {key: {'heatmap': u8 (H2,W2) ndarray, 'classes': u8 ndarray, 'logits': (T,C) tensor}}."""
    if mode not in ('cls', 'seg'):
        raise ValueError("mode must be 'cls' or 'seg'")
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
            # (pinned staging + asynchronous copy: a pageable-memory copy would block the host until everything enqueued so far has run)
            def upload_mask():
                return S._upload(entry['mask'], torch.as_tensor(entry['mask']).dtype, dev) if entry.get('mask') is not None else None
            # (mode 'seg' uploads its mask - the size of the scan level there - AFTER the tile loop is enqueued: pinning 35 MB of host
            # memory costs the host 5-13 ms, during which the GPU would sit idle at the head of the call; r05 profiles/r05_api_seg_breakdown.txt)
            mask = upload_mask() if mode == 'cls' else None
            if mode == 'cls' and isinstance(model, SlideClassifierModel):
                # fused fast path: the stem kernel reads the HBM-resident slide directly
                level = scan.device_level(args.scan_level, dev)
                r = S.infer_slide_cls(model.fused_engine(dev), level, ds.tile_xy, ds.params.ph, ds.params.pw, m, map_hw,
                                      args.num_classes, args.class_probs, mask, rank, world, want_probs=False)
            elif mode == 'cls':
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
            else:
                # mode 'seg' (the reference's default, utils/eval.py:196-215): pred_src = model.decoder(model.encoder(x)) is a
                # (B, C, ph, pw) block per tile, optionally nearest-resized to (tile_h * r, tile_w * r), added at int(m * x),
                # int(m * y) over a (dy, dx) = int(m * ph), int(m * pw) footprint - which therefore has to equal the block size
                pred = torch.zeros((args.num_classes,) + tuple(map_hw), dtype=torch.float64, device=dev)
                dy, dx = int(m * ds.params.ph), int(m * ds.params.pw)
                # world > 1 (seg mode): every rank runs a contiguous share of the raster-order tile list into its
                # own map and sends the band its tiles touch to rank 0 (slide.gather_map_bands: a direct gather, no all-reduce);
                # rank 0 sums the bands - exact, hence identical to the single-rank map, inside the exponent-span bound of the
                # stitch - thresholds, and broadcasts the two u8 maps
                my_it = it
                if world > 1:
                    lo, hi = S.shard_range(len(ds), rank, world)
                    my_it = it.shard(lo, hi)
                span_lo, span_hi = None, None                 # exponent range of every addend: the exactness guard of the float64 sums
                my_txy = []

                for xy, pred_src in _dense_batches(model, my_it, scan, dev, lambda x: model.decoder(model.encoder(x))):
                    if args.scan_resize != 1:
                        pred_src = E.resize_nearest(pred_src, (int(args.tile_h * args.scan_resize), int(args.tile_w * args.scan_resize)))
                    if tuple(pred_src.shape[2:]) != (dy, dx):
                        raise ValueError("mode='seg': the decoder output %s does not match the stitch footprint (%d, %d) = int(m * tile); "
                                         "scan at the map's level or set scan_resize (utils/eval.py:202-215)" % (tuple(pred_src.shape[2:]), dy, dx))
                    txy = S.map_coords(xy, m)
                    my_txy.append(np.asarray(txy))
                    E.stitch_add_dense(pred, pred_src, S._upload(txy, torch.int32, dev))      # (pinned + asynchronous: a pageable copy blocks the host)
                    sp = E.exponent_span(pred_src)
                    span_lo = sp[0:1] if span_lo is None else torch.minimum(span_lo, sp[0:1])
                    span_hi = sp[1:2] if span_hi is None else torch.maximum(span_hi, sp[1:2])
                span = torch.cat((span_lo, span_hi)) if span_lo is not None else None
                mask = upload_mask()
                if world > 1:
                    txy_all = np.concatenate(my_txy) if my_txy else np.zeros((0, 2), np.int64)
                    pred = S.gather_map_bands(pred, txy_all, dy, dx, rank, world, dst=0)
                    span = S.allreduce_span(span, dev)
                if world == 1 or rank == 0:
                    classes, _, heat = E.softmax_threshold_argmax(pred, args.class_probs, mask, 'seg', want_probs=False)
                else:
                    classes, heat = (torch.empty(tuple(map_hw), dtype=torch.uint8, device=dev) for _ in range(2))
                if world > 1:
                    classes, heat = S.broadcast_from(classes, 0), S.broadcast_from(heat, 0)
                r = {'logits': None, 'pred': pred, 'classes': classes, 'heatmap': heat, 'exponent_span': span}
            stitch_exact = None
            if r.get('exponent_span') is not None:             # the float64 stitch is exact (order- and rank-independent) inside this
                over = (-(-ds.params.ph // ds.params.sh) + 1) * (-(-ds.params.pw // ds.params.sw) + 1)      # bound; outside it the map
                stitch_exact = E.stitch_is_exact(r['exponent_span'], over)                                   # is right to float64 rounding only
            heat, classes_np = _to_host(r['heatmap'], r['classes'])          # both u8 maps in one round trip through pinned memory
            results[key] = {'heatmap': heat, 'classes': classes_np, 'logits': r['logits'],
                            'precision': r.get('precision'),          # precision='auto': the mode this slide ran in, and why
                            'stitch_exact': stitch_exact}
            if save and rank == 0:
                _save_png(heat, '{}/{}_{}_heatmap.png'.format(out_dir, key, args.tile_stride_w))
                thumb = np.asarray(scan.read_region((0, 0), ref_level, scan.level_dimensions[ref_level]).convert('RGB'))
                over = thumb * 0.75 + 255 * np.repeat((heat > 255 * 0.99)[..., None], 3, -1) * 0.25
                _save_png(np.uint8(over), '{}/{}_{}_overlay.png'.format(out_dir, key, args.tile_stride_w))
            dataset.wsis[key] = None                       # the reference frees each slide after use (:282)
    if was_training:
        model.train()
    return results


def _map_u8(arr, hw, dev):
    """Ground-truth image -> uint8 (H,W) device tensor at map resolution (nearest-neighbour when the size differs; the
    reference resizes with PIL's version-dependent default filter, utils/eval.py:77-78)."""
    a = np.asarray(arr)
    if a.ndim == 3:
        a = a[..., 0]
    if tuple(a.shape) != tuple(hw):
        ys = (np.arange(hw[0]) * a.shape[0] // hw[0]).clip(0, a.shape[0] - 1)
        xs = (np.arange(hw[1]) * a.shape[1] // hw[1]).clip(0, a.shape[1] - 1)
        a = a[ys][:, xs]
    return torch.from_numpy(np.ascontiguousarray(a.astype(np.uint8))).to(dev)


def predict_wsis(model, dataset, ep, save=True):
    """Dense per-pixel map prediction with the tumour-bed post-process (reference utils/eval.py:22-152).
    `model(batch_image)` must return (B, C, ph, pw) logits on the GPU (the reference drives an smp U-Net here; any GPU
    module works, e.g. wsi_segmentation_pipeline_amd.unet.UNetSeg).  Everything between the tile list and the colour
    mask stays on the device:
      :58-60    pred[:, y:y+ph, x:x+pw] += pred_src[b]      float64, scan-level resolution (wsi_stitch_add_dense)
      :66-71    cv2.resize of every class map to level-2     wsi_resize_bilinear_f64
      :82-96    argmax -> (p >= 2) -> open 20x20 -> convex hull -> perimeter -> dilate 20x20     wsi_tumor_bed
      :100-123  tumour-bed IoU, accuracy / score figures against `entry['gt']` / `entry['tb_gt']` (when the dataset
                entry carries them: the reference reads <wsipath>_mask.png / _tumor_bed.png)    wsi_score_counts, wsi_mask_iou_counts
      :138-145  colour mask (threshold_probs classes on the foreground mask, tumour-bed outline in white), saved at half size
    Returns {key: {'pred', 'pred_level2', 'classes' (scan level), 'classes_level2', 'tumor_bed', 'outline', 'scores'}}."""
    from wsi_segmentation_pipeline_amd import postprocess as PP
    out_dir = '{}/{}'.format(args.val_save_pth, ep)
    if save:
        os.makedirs(out_dir, exist_ok=True)
    dev = _device_of(model)
    was_training = model.training
    model.eval()
    results = {}
    ious_tb = 0.0
    with torch.no_grad():
        for key in list(dataset.wsis):
            entry = dataset.wsis[key]
            it = entry['iterator']
            ds, scan = it.dataset, entry['scan']
            iw, ih = scan.level_dimensions[args.scan_level]
            pred = torch.zeros((args.num_classes, ih, iw), dtype=torch.float64, device=dev)
            for xy, pred_src in _dense_batches(model, it, scan, dev, model):
                if pred_src.dim() != 4 or pred_src.shape[1] != args.num_classes:
                    raise ValueError('predict_wsis needs a dense model returning (B, %d, ph, pw)' % args.num_classes)
                E.stitch_add_dense(pred, pred_src, S._upload(xy, torch.int32, dev))        # int(batch_x[bj]): truncation
            classes = PP.argmax_classes(pred)
            ref_level = min(2, len(scan.level_dimensions) - 1)
            map_hw = tuple(scan.level_dimensions[ref_level][::-1])
            pred2 = PP.resize_bilinear(pred, map_hw)                           # :66-71
            p = PP.argmax_classes(pred2)                                       # :82
            tb = PP.tumor_bed(p, 2, 20, 20)                                    # :90-96
            mask = entry.get('mask')
            mask_dev = _map_u8(mask, map_hw, dev) if mask is not None else torch.ones(map_hw, dtype=torch.uint8, device=dev)
            scores = None
            if entry.get('gt') is not None:                                    # :74-135
                scores = PP.wsi_scores(p, _map_u8(entry['gt'], map_hw, dev), (mask_dev > 0).to(torch.uint8), args.epsilon)
                scores['iou_tb'] = -1
                if entry.get('tb_gt') is not None:
                    scores['iou_tb'] = PP.mask_iou((_map_u8(entry['tb_gt'], map_hw, dev) > 0).to(torch.uint8), tb.tb_pred, args.epsilon)
                    ious_tb += scores['iou_tb']
                print('{}, {:.3f}({:.3f}), {:.3f}({:.3f}), {:.3f}, tb iou: {:.3f} '.format(
                    key, scores['s_masked'], scores['s'], scores['acc_masked'], scores['acc'], scores['iou_fg'], scores['iou_tb']))
            results[key] = {'pred': pred, 'pred_level2': pred2, 'classes': classes, 'classes_level2': p,
                            'tumor_bed': tb.tb_pred, 'outline': tb.outline, 'scores': scores}
            if save:                                                           # :138-145
                cls_thr = E.softmax_threshold_argmax(pred2, args.class_probs, want_probs=False)[0]      # pred_to_mask
                rgb = torch.zeros(map_hw + (3,), dtype=torch.uint8, device=dev)
                for cj in range(min(args.num_classes - 1, 3)):                  # class k > 0 -> channel k - 1
                    rgb[..., cj] = (cls_thr == cj + 1).to(torch.uint8) * 255
                rgb = rgb * mask_dev[..., None]
                rgb[tb.outline > 0] = 255
                from PIL import Image
                img = Image.fromarray(rgb.cpu().numpy())
                img.resize((max(1, map_hw[1] // 2), max(1, map_hw[0] // 2))).save('{}/{}_{}.png'.format(out_dir, key, args.tile_stride_w))
        if dataset.wsis:
            print('Average tb iou: {:.3f}'.format(ious_tb / len(dataset.wsis)))
    if was_training:
        model.train()
    return results


def tumor_bed_overlay(heat_u8, thumb_rgb=None):
    """The tumour-bed outline of a stitched u8 heat map (reference paper_tools/overlay_tb_wsi.py:46-64) on the device:
    (heat / 255 >= 0.9) -> open 30x30 -> convex hull -> perimeter -> dilate 20x20.  Returns the
    wsi_segmentation_pipeline_amd.postprocess.TumorBed (opened mask, hull image, outline, .outline_points(n) via esp) and,
    with a (H,W,3) u8 thumbnail on the GPU, the overlay 0.65 * wsi + 0.35 * (heat * opened) with the outline in black."""
    from wsi_segmentation_pipeline_amd import postprocess as PP
    tb = PP.tumor_bed_from_heatmap(heat_u8, 0.9, 30, 20)
    if thumb_rgb is None:
        return tb, None
    hm = (heat_u8 * tb.opened).to(torch.float64)[..., None].expand(-1, -1, 3)
    over = 0.65 * thumb_rgb.to(torch.float64) + 0.35 * hm
    over[tb.outline > 0] = 0
    return tb, over.to(torch.uint8)


def predict_regions(model, iterator, metadata, label_shape, class_probs=None, rank=0, world=1):
    """Region-proposal evaluation stage (reference scannet.py:145-155 / slic.py:93-99, with the documented fix: the
    ensemble logits are soft-maxed over the class axis): bags -> ResNet bag forward on the HIP path -> class per region ->
    painted label image (int64 ndarray).  With world > 1 (torch.distributed initialised) the bags are sharded over the ranks
    by greedy cost balance, each rank runs its share, ONE all-gather of the (R, C) ensemble logits follows and every rank
    paints the identical label image on its device (wsi_paint_regions).  `iterator`: a utils.dataset_hr.DeviceBagIterator
    (`.dataset`, `.shard`: needed for world > 1), or - single rank - any iterable of (images (B,16,3,64,64), tile_ids) batches
    like the reference's DataLoader (scannet.py:147-155)."""
    from wsi_segmentation_pipeline_amd import bags as B
    class_probs = args.class_probs if class_probs is None else class_probs
    dev = _device_of(model)
    was_training = model.training
    model.eval()
    data = getattr(iterator, 'dataset', None)
    if world > 1 and (data is None or not hasattr(iterator, 'shard')):
        raise ValueError('predict_regions over several ranks needs a shardable iterator (utils.dataset_hr.DeviceBagIterator)')
    R = len(data) if data is not None else None
    shards = B.shard_bags(np.full(R, 16.0), world) if world > 1 else None
    mine = iterator.shard(shards[rank]) if world > 1 else iterator
    seen_ids = []                                             # plain iterables: the tile ids come with the batches
    with torch.no_grad():
        parts = []
        eng = model.hip_engine(dev) if world > 1 and hasattr(model, 'hip_engine') else None
        if hasattr(eng, 'probe_f32'):
            # precision='auto' over several ranks: ONE mode for every rank.  The probe is a stratified sample over this rank's
            # WHOLE shard of bags (r03 advisor finding: probing the first batch only is the first-tiles blind spot the per-slide
            # stratified probe was introduced to remove): up to `eng.probe` bags spread evenly over the shard, their crops
            # sub-sampled by probe_f32; a rank without bags still takes part in the collective
            mine_idx = list(shards[rank])
            eng.reset()
            err = 0.0
            if mine_idx:
                nb = min(int(getattr(eng, 'probe', 32)), len(mine_idx))
                pick = sorted({mine_idx[int(round(v))] for v in np.linspace(0, len(mine_idx) - 1, nb)})
                sample = torch.cat([im.to(dev).reshape(-1, *im.shape[2:]) for im, _ in iterator.shard(pick)])
                err = eng.probe_f32(sample)
            eng.decide(S.allreduce_max(err, dev, world), scope='regions')
        for images, tile_ids in mine:
            if data is None:
                seen_ids.extend(int(t) for t in tile_ids)
            parts.append(model(images.to(dev))[1])
        num_classes = len(class_probs)
        local = torch.cat(parts) if parts else torch.zeros((0, num_classes), dtype=torch.float32, device=dev)
        ens = B.gather_rows(local, shards, rank, world) if world > 1 else local
        if R is None:
            R = len(seen_ids)
        if R:
            # (R,C) logits as a (C,R,1) "map": softmax over classes / threshold / argmax in one HIP kernel
            as_map = ens.t().to(torch.float64).contiguous().view(ens.shape[1], -1, 1)
            cls = E.softmax_threshold_argmax(as_map, class_probs, want_probs=False)[0].view(-1)
            ids = [int(r['tile_id']) for r in data.datalist] if data is not None else seen_ids
            index_lists = [metadata[t]['foreground_indices'] for t in ids]
            pred_mask = B.paint_regions(tuple(label_shape), index_lists, cls, dev).cpu().numpy()
        else:
            pred_mask = np.zeros(label_shape, dtype=np.int64)
    if was_training:
        model.train()
    return pred_mask
