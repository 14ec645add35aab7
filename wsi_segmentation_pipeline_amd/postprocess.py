"""Device-side tumour-bed post-process of the stitched prediction map.

Host wrappers over the C ABI (include/wsi_hip.h, csrc/postproc.hip) for what the reference does on the CPU with
OpenCV / scikit-image / mahotas after the sliding-window loop:
  /root/reference/utils/eval.py:66-71,82-96,100-123   resize -> argmax -> (p >= 2) -> open 20x20 -> convex hull ->
                                                      perimeter -> dilate 20x20, tumour-bed IoU, accuracy / score figures
  /root/reference/paper_tools/overlay_tb_wsi.py:46-64 the same outline from a saved u8 heat map
  /root/reference/contour_ordering.py:33-60           evenly_spaced_points_on_a_contour (esp)
Every tensor stays on the GPU; there is no CPU fallback.
"""
import ctypes as C
import math

import torch

from . import native
from .engine import _ptr, _require_gpu, _stream


def _u8(t, what):
    _require_gpu(t, what)
    if t.dtype == torch.bool:
        t = t.to(torch.uint8)
    if t.dtype != torch.uint8 or t.dim() != 2:
        raise ValueError('%s must be a 2-D uint8 tensor' % what)
    return t.contiguous()


def resize_bilinear(pred, out_hw):
    """(C,H,W) float64 GPU -> (C,h,w): reference utils/eval.py:66-71 (cv2.resize, INTER_LINEAR)."""
    lib = native.load()
    _require_gpu(pred, 'prediction map')
    pred = pred.to(torch.float64).contiguous()
    c, hs, ws = pred.shape
    out = torch.empty((c, int(out_hw[0]), int(out_hw[1])), dtype=torch.float64, device=pred.device)
    native.check(lib.wsi_resize_bilinear_f64(_ptr(pred), c, hs, ws, _ptr(out), out.shape[1], out.shape[2], _stream()),
                 'wsi_resize_bilinear_f64')
    return out


def argmax_classes(pred):
    """np.argmax(pred, 0) of a (C,H,W) float64 GPU map -> uint8 (H,W) (utils/eval.py:82)."""
    lib = native.load()
    _require_gpu(pred, 'prediction map')
    pred = pred.to(torch.float64).contiguous()
    c, h, w = pred.shape
    out = torch.empty((h, w), dtype=torch.uint8, device=pred.device)
    native.check(lib.wsi_argmax_classes(_ptr(pred), c, h * w, _ptr(out), _stream()), 'wsi_argmax_classes')
    return out


def morph_rect(img, k, op):
    """cv2.erode ('erode') / cv2.dilate ('dilate') / cv2.morphologyEx(MORPH_OPEN) ('open') with np.ones((k, k))."""
    lib = native.load()
    img = _u8(img, 'mask')
    out, tmp = torch.empty_like(img), torch.empty_like(img)
    native.check(lib.wsi_morph_rect(_ptr(img), _ptr(out), _ptr(tmp), img.shape[0], img.shape[1], int(k),
                                    {'erode': 0, 'dilate': 1, 'open': 2}[op], _stream()), 'wsi_morph_rect')
    return out


def bwperim(img):
    lib = native.load()
    img = _u8(img, 'mask')
    out = torch.empty_like(img)
    native.check(lib.wsi_bwperim(_ptr(img), _ptr(out), img.shape[0], img.shape[1], _stream()), 'wsi_bwperim')
    return out


class TumorBed:
    """Result of `tumor_bed`: .opened, .tb_pred (hull image), .outline (dilated perimeter) uint8 (H,W) GPU tensors;
    .polygon() = the hull as a closed (x, y) float64 contour; .outline_points(n) = esp over that contour."""

    def __init__(self, opened, tb_pred, outline, ws, hw):
        self.opened, self.tb_pred, self.outline, self._ws, self._hw = opened, tb_pred, outline, ws, hw

    def polygon(self):
        lib = native.load()
        h, w = self._hw
        cap = 4 * h + 8
        out = torch.empty((cap, 2), dtype=torch.float64, device=self.tb_pred.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=self.tb_pred.device)
        native.check(lib.wsi_hull_polygon(_ptr(self._ws), h, w, _ptr(out), cap, _ptr(cnt), _stream()), 'wsi_hull_polygon')
        return out[:int(cnt.item())]

    def outline_points(self, num_pts):
        poly = self.polygon()
        if poly.shape[0] == 0:
            return poly
        return esp(poly, num_pts)


def tumor_bed(codes, min_code=2, open_k=20, dilate_k=20):
    """utils/eval.py:90-96 on the device: (codes >= min_code) -> MORPH_OPEN open_k -> convex hull image -> bwperim ->
    dilate dilate_k.  `codes` = u8 class map (min_code 2: classes 2 and 3 are tumour) or u8 heat map."""
    lib = native.load()
    codes = _u8(codes, 'class / heat map')
    h, w = codes.shape
    dev = codes.device
    ws = torch.empty(lib.wsi_tumor_bed_workspace_bytes(h, w), dtype=torch.uint8, device=dev)
    opened = torch.empty((h, w), dtype=torch.uint8, device=dev)
    tb_pred, outline = torch.empty_like(opened), torch.empty_like(opened)
    native.check(lib.wsi_tumor_bed(_ptr(codes), h, w, int(min_code), int(open_k), int(dilate_k), _ptr(opened), _ptr(tb_pred),
                                   _ptr(outline), _ptr(ws), _stream()), 'wsi_tumor_bed')
    return TumorBed(opened, tb_pred, outline, ws, (h, w))


def tumor_bed_from_heatmap(heat_u8, thresh=0.9, open_k=30, dilate_k=20):
    """paper_tools/overlay_tb_wsi.py:46-64: uint8(heat / 255 >= thresh) is `heat >= ceil(255 * thresh)` on u8 codes
    (for thresh = 0.9: 229.5 -> 230; checked against the float64 division in the tests)."""
    lo = int(math.ceil(255.0 * thresh - 1e-9))
    while lo > 0 and (lo - 1) / 255 >= thresh:
        lo -= 1
    while lo / 255 < thresh:
        lo += 1
    return tumor_bed(heat_u8, lo, open_k, dilate_k)


def mask_iou(tb_gt, tb_pred, epsilon=1e-8):
    """utils/eval.py:104: (tb_gt * tb_pred).sum() / (eps + (tb_gt | tb_pred).sum()); counts are exact integers."""
    lib = native.load()
    a, b = _u8(tb_gt, 'ground-truth tumour bed'), _u8(tb_pred, 'predicted tumour bed')
    out = torch.zeros(2, dtype=torch.int64, device=a.device)
    native.check(lib.wsi_mask_iou_counts(_ptr(a), _ptr(b), a.numel(), _ptr(out), _stream()), 'wsi_mask_iou_counts')
    inter, union = (int(v) for v in out.cpu())
    return inter / (epsilon + union)


def wsi_scores(p, gt, mask, epsilon=1e-8):
    """utils/eval.py:107-121: acc, s, acc_masked, s_masked, iou_fg from exact integer sums taken on the device."""
    lib = native.load()
    p, gt, mask = _u8(p, 'class map'), _u8(gt, 'ground truth'), _u8(mask, 'foreground mask')
    out = torch.zeros((2, 6), dtype=torch.int64, device=p.device)
    native.check(lib.wsi_score_counts(_ptr(p), _ptr(gt), None, p.numel(), _ptr(out[0]), _stream()), 'wsi_score_counts')
    native.check(lib.wsi_score_counts(_ptr(p), _ptr(gt), _ptr(mask), p.numel(), _ptr(out[1]), _stream()), 'wsi_score_counts')
    o = out.cpu().numpy()

    def acc_s(r):
        return float(r[1] / r[0]) if r[0] else float('nan'), float(1 - r[2] / r[3]) if r[3] else float('nan')
    acc, s = acc_s(o[0])
    acc_m, s_m = acc_s(o[1])
    return {'acc': acc, 's': s, 'acc_masked': acc_m, 's_masked': s_m, 'iou_fg': float(o[1][4] / (epsilon + o[1][5]))}


def esp(points, num_pts):
    """contour_ordering.evenly_spaced_points_on_a_contour on the device: (N,2) ordered contour -> (num_pts,2) float64."""
    lib = native.load()
    _require_gpu(points, 'contour')
    pts = points.to(torch.float64).contiguous()
    n = pts.shape[0]
    out = torch.empty((int(num_pts), 2), dtype=torch.float64, device=pts.device)
    scratch = torch.empty(n, dtype=torch.float64, device=pts.device)
    native.check(lib.wsi_esp(_ptr(pts), n, int(num_pts), _ptr(out), _ptr(scratch), _stream()), 'wsi_esp')
    return out
