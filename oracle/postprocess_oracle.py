"""CPU spec (NumPy, exact integer / float64 arithmetic) of the tumour-bed post-process that follows the stitched map.

Test infrastructure only (see oracle/__init__.py).  Restates
  /root/reference/utils/eval.py:66-71      cv2.resize of the float64 class maps to level-2 dimensions
  /root/reference/utils/eval.py:82-96      argmax -> (p >= 2) -> MORPH_OPEN 20x20 -> convex hull -> perimeter -> dilate 20x20
  /root/reference/utils/eval.py:100-123    tumour-bed IoU, accuracy / score figures against a ground-truth class map
  /root/reference/paper_tools/overlay_tb_wsi.py:46-64   the same outline from a u8 heat map (>= 0.9, open 30x30)
  /root/reference/contour_ordering.py:33-60             esp (re-exported from wsi_oracle, pinned by tests/golden/esp.npz)

The arithmetic of the morphology / hull / perimeter steps lives in THIRD-PARTY packages that are absent here and
un-pinned in the reference (no requirements file): OpenCV (`cv2.morphologyEx`, `cv2.dilate`, `cv2.resize`),
scikit-image (`skimage.morphology.convex_hull_image`, imported as chull) and mahotas (`mahotas.bwperim`).  The
reference holds no tests or fixtures for them, so this file restates their PUBLISHED algorithms as an exact,
deterministic spec - **parity unpinned** for these steps - and the HIP path is held bit-exact
to this spec:

  erode / dilate (OpenCV, rectangular k x k element, default anchor = (k//2, k//2), default border):
      erode(x, y)  = min over i, j in [0, k) of src(x + i - k//2, y + j - k//2), outside the image = +inf (ignored)
      dilate(x, y) = max over the same offsets, outside the image = -inf (ignored)
      MORPH_OPEN   = dilate(erode(src)); for even k both passes look at offsets -k/2 .. k/2 - 1.
  convex_hull_image (scikit-image, offset_coordinates=True): the hull of the four edge mid-points (r +- 0.5, c),
      (r, c +- 0.5) of every foreground pixel; a pixel belongs to the hull image iff its centre (r, c) lies inside or
      ON the hull polygon.  Here evaluated in exact integer arithmetic on doubled coordinates (skimage itself uses
      Qhull + a float tolerance of 1e-10, which agrees with the exact predicate except for float ties).
  bwperim (mahotas, n = 4): a pixel is perimeter iff it is set and at least one of its four neighbours (outside the
      image = 0) is not set.
  cv2.resize (INTER_LINEAR, float64): half-pixel centres, fx = (dx + 0.5) * (W_src / W_dst) - 0.5, edge-clamped,
      v = (a*(1-wx) + b*wx)*(1-wy) + (c*(1-wx) + d*wx)*wy evaluated in float64 in exactly this order.
"""
import numpy as np

from .wsi_oracle import evenly_spaced_points_on_a_contour  # noqa: F401  (re-export)


# --------------------------------------------------------------------------- cv2.resize INTER_LINEAR (float64)
def _lin_coords(n_dst, n_src):
    scale = n_src / n_dst
    f = (np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5
    i0 = np.floor(f)
    w = f - i0
    i0 = i0.astype(np.int64)
    lo = i0 < 0
    i0[lo], w[lo] = 0, 0.0
    i1 = np.minimum(i0 + 1, n_src - 1)
    hi = i0 >= n_src - 1
    i0[hi], w[hi] = n_src - 1, 0.0
    return i0, i1, w


def resize_bilinear(pred, out_hw):
    """(C, H, W) float64 -> (C, h, w): utils/eval.py:66-71 (`cv2.resize(pred[ij], level_dimensions[2])`)."""
    pred = np.asarray(pred, np.float64)
    y0, y1, wy = _lin_coords(out_hw[0], pred.shape[1])
    x0, x1, wx = _lin_coords(out_hw[1], pred.shape[2])
    a = pred[:, y0][:, :, x0]
    b = pred[:, y0][:, :, x1]
    c = pred[:, y1][:, :, x0]
    d = pred[:, y1][:, :, x1]
    top = a * (1.0 - wx) + b * wx
    bot = c * (1.0 - wx) + d * wx
    return top * (1.0 - wy)[None, :, None] + bot * wy[None, :, None]


# --------------------------------------------------------------------------- morphology
def _rect_pass(img, k, axis, take_min):
    h = k // 2
    n = img.shape[axis]
    fill = 1 if take_min else 0                             # ignored border: neutral element
    out = np.full(img.shape, fill, np.uint8)
    for off in range(-h, k - h):                            # offsets -k//2 .. k - k//2 - 1
        src = [slice(None)] * 2
        dst = [slice(None)] * 2
        a, e = max(0, -off), min(n, n - off)
        if e <= a:
            continue
        dst[axis] = slice(a, e)
        src[axis] = slice(a + off, e + off)
        blk = img[tuple(src)]
        cur = out[tuple(dst)]
        out[tuple(dst)] = np.minimum(cur, blk) if take_min else np.maximum(cur, blk)
    return out


def erode_rect(img, k):
    img = (np.asarray(img) != 0).astype(np.uint8)
    return _rect_pass(_rect_pass(img, k, 1, True), k, 0, True)


def dilate_rect(img, k):
    img = (np.asarray(img) != 0).astype(np.uint8)
    return _rect_pass(_rect_pass(img, k, 1, False), k, 0, False)


def morph_open(img, k):
    """cv2.morphologyEx(img, cv2.MORPH_OPEN, np.ones((k, k)))"""
    return dilate_rect(erode_rect(img, k), k)


# --------------------------------------------------------------------------- convex hull image
def _cross(o, a, b):
    return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])


def hull_vertices_doubled(img):
    """Hull of the diamond offsets of the foreground pixels, doubled integer coordinates (R, C) = (2r, 2c) +- 1, as the
    two monotone chains (left = minimal C per R going down, right = maximal C per R): lists of (R, C), R ascending,
    collinear points dropped.  Empty image -> ([], [])."""
    img = np.asarray(img) != 0
    rows = np.nonzero(img.any(1))[0]
    if len(rows) == 0:
        return [], []
    H = img.shape[0]
    lo = np.full(2 * H + 1, np.iinfo(np.int64).max, np.int64)      # index R + 1 for R in [-1, 2H-1]
    hi = np.full(2 * H + 1, np.iinfo(np.int64).min, np.int64)
    for r in rows:
        cs = np.nonzero(img[r])[0]
        cmin, cmax = int(cs[0]), int(cs[-1])
        for R, cl, ch in ((2 * r, 2 * cmin - 1, 2 * cmax + 1), (2 * r - 1, 2 * cmin, 2 * cmax), (2 * r + 1, 2 * cmin, 2 * cmax)):
            lo[R + 1] = min(lo[R + 1], cl)
            hi[R + 1] = max(hi[R + 1], ch)
    Rs = [R for R in range(-1, 2 * H) if lo[R + 1] <= hi[R + 1]]
    left, right = [], []
    for R in Rs:                                            # Andrew's monotone chain on each side, exact integers
        p = (R, int(lo[R + 1]))
        while len(left) >= 2 and _cross(left[-2], left[-1], p) <= 0:      # lower hull in the (R, C) plane: smallest C
            left.pop()
        left.append(p)
        q = (R, int(hi[R + 1]))
        while len(right) >= 2 and _cross(right[-2], right[-1], q) >= 0:     # upper hull: largest C
            right.pop()
        right.append(q)
    return left, right


def _chain_interval(chain, R, want_min):
    """C-coordinate (exact rational num/den, den > 0) where the chain crosses doubled row R."""
    for (R0, C0), (R1, C1) in zip(chain, chain[1:]):
        if R0 <= R <= R1:
            return C0 * (R1 - R0) + (C1 - C0) * (R - R0), (R1 - R0)
    return (chain[0][1], 1) if R == chain[0][0] else (chain[-1][1], 1)


def convex_hull_image(img):
    """skimage.morphology.convex_hull_image(img) (offset_coordinates=True), exact predicate; uint8 0/1."""
    img = np.asarray(img)
    out = np.zeros(img.shape, np.uint8)
    left, right = hull_vertices_doubled(img)
    if not left:
        return out
    Rmin, Rmax = left[0][0], left[-1][0]
    for r in range(img.shape[0]):
        R = 2 * r
        if R < Rmin or R > Rmax:
            continue
        ln, ld = _chain_interval(left, R, True)
        rn, rd = _chain_interval(right, R, False)
        # pixel c is inside iff  ln/ld <= 2c <= rn/rd   (boundary inclusive)
        c_lo = -((-ln) // (2 * ld))                         # ceil(ln / (2 ld))
        c_hi = rn // (2 * rd)                               # floor(rn / (2 rd))
        c_lo, c_hi = max(c_lo, 0), min(c_hi, img.shape[1] - 1)
        if c_hi >= c_lo:
            out[r, c_lo:c_hi + 1] = 1
    return out


def hull_polygon(img):
    """Hull vertices as an ordered CLOSED contour in pixel coordinates (x, y) float64 (first point repeated at the end):
    down the right chain, back up the left chain - the input of esp for an evenly spaced tumour-bed outline."""
    left, right = hull_vertices_doubled(img)
    if not left:
        return np.zeros((0, 2))
    pts = right + left[::-1]
    dedup = [pts[0]]
    for p in pts[1:]:
        if p != dedup[-1]:
            dedup.append(p)
    if dedup[-1] != dedup[0]:
        dedup.append(dedup[0])
    return np.array([[C / 2.0, R / 2.0] for R, C in dedup], np.float64)


# --------------------------------------------------------------------------- perimeter
def bwperim(img):
    """mahotas.bwperim(img, n=4)"""
    bw = np.asarray(img) != 0
    pad = np.pad(bw, 1)
    inner = pad[:-2, 1:-1] & pad[2:, 1:-1] & pad[1:-1, :-2] & pad[1:-1, 2:]
    return (bw & ~inner).astype(np.uint8)


# --------------------------------------------------------------------------- the pipelines
def tumor_bed(class_map, min_class=2, open_k=20, dilate_k=20):
    """utils/eval.py:90-96: returns (tb_pred = hull image of the opened (p >= min_class) mask, outline = dilated perimeter)."""
    tb = (np.asarray(class_map).astype(np.uint8) >= min_class).astype(np.uint8)
    tb = morph_open(tb, open_k)
    tb_pred = convex_hull_image(tb)
    outline = dilate_rect(bwperim(tb_pred), dilate_k)
    return tb_pred, outline


def tumor_bed_from_heatmap(heat_u8, thresh=0.9, open_k=30, dilate_k=20):
    """paper_tools/overlay_tb_wsi.py:46-64: im = uint8(heat/255 >= 0.9) -> open 30x30 -> chull -> bwperim -> dilate 20x20.
    Returns (opened mask, hull image, outline)."""
    im = (np.asarray(heat_u8).astype(np.float64) / 255 >= thresh).astype(np.uint8)
    im = morph_open(im, open_k)
    hull = convex_hull_image(im)
    return im, hull, dilate_rect(bwperim(hull), dilate_k)


def tumor_bed_iou(tb_gt, tb_pred, epsilon=1e-8):
    """utils/eval.py:104: (tb_gt * tb_pred).sum() / (eps + (tb_gt | tb_pred).sum())"""
    tb_gt = (np.asarray(tb_gt) > 0).astype(np.uint8)
    tb_pred = np.asarray(tb_pred).astype(np.uint8)
    return float((tb_gt * tb_pred).sum() / (epsilon + (tb_gt | tb_pred).sum()))


def wsi_scores(p, gt, mask, epsilon=1e-8):
    """utils/eval.py:107-121: dict of acc, s, acc_masked, s_masked, iou_fg for a class map p (np.argmax: int64), ground truth gt
    and foreground mask.  Operator precedence AND dtypes kept: `1 - gt > 0` is `(1 - gt) > 0` on the uint8 array of a PIL image
    (utils/eval.py:76-78), where 1 - gt wraps (0 -> 1, 1 -> 0, 2 -> 255, 3 -> 254): the factor is gt != 1."""
    p = np.asarray(p).astype(np.int64)
    gt8 = np.asarray(gt).astype(np.uint8)
    gt = gt8.astype(np.int64)
    mask = np.asarray(mask).astype(np.int64)

    def acc_s(p):
        acc = float(np.mean((p == gt)[gt > 0]))
        den = np.sum(np.maximum(np.abs(gt - 0), np.abs(gt - 3.0)) * (1 - (1 - (p > 0)) * (np.uint8(1) - gt8 > 0)))
        return acc, float(1 - np.sum(np.abs(p - gt)) / den)
    acc, s = acc_s(p)
    pm = mask * p
    acc_m, s_m = acc_s(pm)
    iou_fg = float(((pm > 0) * (gt > 0)).sum() / (epsilon + ((pm > 0) | (gt > 0)).sum()))
    return {'acc': acc, 's': s, 'acc_masked': acc_m, 's_masked': s_m, 'iou_fg': iou_fg}
