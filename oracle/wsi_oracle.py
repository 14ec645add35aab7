"""CPU restatement (NumPy / plain Python loops) of the sliding-window and region-bag drivers.

Test infrastructure only (see oracle/__init__.py).  These follow reference code that cannot be
imported here (openslide / cv2 / skimage / mahotas / torchvision are absent) and for which the
reference holds no tests: *parity unpinned by the reference*; pinned by hand-derived counts
(24 648 and 96 099 tiles) and property tests in tests/.

Every function cites the reference lines it restates.  Loops are written the way the reference
writes them (clarity over speed) - the product has its own vectorised / device implementations.
"""
import numpy as np
import torch


# --------------------------------------------------------------------------- foreground / grid
def isforeground(arr, thresh=0.05):
    """/root/reference/utils/preprocessing.py:60-71 - fraction of nonzero >= thresh."""
    return np.count_nonzero(arr) / arr.size >= thresh


def find_nuclei_hsv(rgb_u8, mu_percent=0.1):
    """/root/reference/utils/preprocessing.py:94-98,108 (mode='hsv', fill_mask=False).

    The arithmetic lives in scikit-image (``skimage.color.rgb2hsv``; un-vendored, version unpinned).
    Its published algorithm for the S channel: arr = u8/255 as float64; V = max_c arr;
    delta = max_c - min_c; S = delta / V, with S = 0 where delta == 0.  Mask = S > mu_percent.
    """
    arr = np.asarray(rgb_u8)[..., :3].astype(np.float64) / 255.0
    v = arr.max(-1)
    delta = v - arr.min(-1)
    with np.errstate(divide='ignore', invalid='ignore'):
        s = delta / v
    s[delta == 0.0] = 0.0
    return (s > mu_percent).astype(np.uint8)


def find_nuclei_lab(rgb_u8, mu_percent=0.1):
    """/root/reference/utils/preprocessing.py:88-92 (mode='lab'): a = skimage rgb2lab(image)[..., 1]; mask = a > (1 + mu_percent) *
    mean(a).  skimage is absent (parity unpinned): rgb2lab restated (oracle/proposals_oracle.py) and, to make the mean independent
    of the summation order, `a` is rounded to 2^-20 fixed point: mean = (exact integer sum / 2^20) / count."""
    from .proposals_oracle import rgb2lab
    a = rgb2lab(np.asarray(rgb_u8)[..., :3].astype(np.float64) / 255.0)[..., 1]
    aq = np.rint(a * 1048576.0).astype(np.int64)
    mu = (float(int(aq.sum())) / 1048576.0) / float(aq.size)
    return ((aq.astype(np.float64) / 1048576.0) > (1 + mu_percent) * mu).astype(np.uint8)


def fill_mask(mask, kernel_size=10):
    """/root/reference/utils/preprocessing.py:101-106: scipy.ndimage.binary_fill_holes (SciPy itself: pinned) followed by
    cv2.morphologyEx(MORPH_CLOSE, ones((10, 10))) = erode(dilate(.)) with OpenCV's anchor and border rules
    (oracle/postprocess_oracle.py, parity unpinned)."""
    from scipy.ndimage import binary_fill_holes
    from .postprocess_oracle import dilate_rect, erode_rect
    filled = binary_fill_holes(np.asarray(mask) != 0).astype(np.uint8)
    return erode_rect(dilate_rect(filled, kernel_size), kernel_size).astype(np.uint8)


def tile_grid(iw, ih, ph, pw, sh, sw, mask=None, m=1.0, thresh=0.05):
    """/root/reference/utils/dataset.py:143-166.  Returns [(xpos, ypos)] in the reference's order:
    interior raster, then the right-edge column, then the bottom-edge row (no corner tile).
    ``mask`` is the level-2 foreground mask, ``m`` = downsample[scan_level]/downsample[2]."""
    dx, dy = int(pw * m), int(ph * m)

    def keep(xpos, ypos):
        if mask is None:
            return True
        yp, xp = int(ypos * m), int(xpos * m)
        return isforeground(mask[yp:yp + dy, xp:xp + dx], thresh)

    out = []
    for ypos in range(1, ih - 1 - ph, sh):
        for xpos in range(1, iw - 1 - pw, sw):
            if keep(xpos, ypos):
                out.append((xpos, ypos))
    xpos = iw - 1 - pw
    for ypos in range(1, ih - 1 - ph, sh):
        if keep(xpos, ypos):
            out.append((xpos, ypos))
    ypos = ih - 1 - ph
    for xpos in range(1, iw - 1 - pw, sw):
        if keep(xpos, ypos):
            out.append((xpos, ypos))
    return out


def read_tile(level_rgb_u8, x, y, pw, ph):
    """/root/reference/utils/dataset.py:174-178 with OpenSlide replaced by an in-memory level
    image: `read_region((ds*x, ds*y), level, (pw, ph)).convert('RGB')` is the (ph,pw,3) crop at
    level coordinates (x,y); outside the slide OpenSlide returns transparent black -> RGB 0."""
    H, W = level_rgb_u8.shape[:2]
    out = np.zeros((ph, pw, 3), np.uint8)
    y0, y1, x0, x1 = max(y, 0), min(y + ph, H), max(x, 0), min(x + pw, W)
    if y1 > y0 and x1 > x0:
        out[y0 - y:y1 - y, x0 - x:x1 - x] = level_rgb_u8[y0:y1, x0:x1, :3]
    return out


# --------------------------------------------------------------------------- stitch / threshold
def stitch_tumorbed(tile_xy, tile_pred, num_classes, map_hw, m, pw, ph):
    """/root/reference/utils/eval.py:182-186,208-215: float64 sum of per-tile predictions.
    tile_pred: (T,C) ('cls': broadcast over the tile footprint) or (T,C,dy,dx)."""
    pred = np.zeros((num_classes, map_hw[0], map_hw[1]), dtype=np.float64)
    dx, dy = int(m * pw), int(m * ph)
    for (x, y), p in zip(tile_xy, np.asarray(tile_pred)):
        p = np.asarray(p)
        while pred.ndim > p.ndim:            # eval.py:210-211 (per tile instead of per batch)
            p = np.expand_dims(p, -1)
        tx, ty = int(m * float(x)), int(m * float(y))
        pred[:, ty:ty + dy, tx:tx + dx] += p
    return pred


def stitch_wsis(tile_xy, tile_pred, num_classes, level_hw, pw, ph):
    """/root/reference/utils/eval.py:44-46,58-60: accumulate (T,C,ph,pw) at scan_level resolution."""
    pred = np.zeros((num_classes, level_hw[0], level_hw[1]), dtype=np.float64)
    for (x, y), p in zip(tile_xy, np.asarray(tile_pred)):
        tx, ty = int(x), int(y)
        pred[:, ty:ty + ph, tx:tx + pw] += p
    return pred


def threshold_probs(pred, class_probs=(0., 0., 0., 0.)):
    """/root/reference/utils/preprocessing.py:156-172: softmax over classes (float64, torch CPU as
    in the reference), zero probabilities under the per-class threshold, argmax -> u8."""
    p = torch.softmax(torch.from_numpy(np.asarray(pred, dtype=np.float64)), dim=0)
    for cj in range(p.shape[0]):
        p[cj, p[cj, ...] < class_probs[cj]] = 0
    p = p.numpy()
    return np.argmax(p, axis=0).astype(np.uint8), p


def tumorbed_heatmap(probs, mask, mode='cls'):
    """/root/reference/utils/eval.py:219-228: class-1 prob ('cls') or classes 2+3 ('seg'), times the
    foreground mask, then uint8(255*x) (truncation)."""
    heat = probs[1] if mode == 'cls' else probs[2] + probs[3]
    return np.uint8(255 * (mask * heat))


# --------------------------------------------------------------------------- region bags
HR_NUM_CNT_SAMPLES = 8      # /root/reference/utils/dataset_hr.py:14-18
HR_NUM_PERIM_SAMPLES = 8
HR_SCAN_LEVEL = 1
HR_PATCH_W = 64
HR_PATCH_H = 64


def map_points(arr, scan_level, tile_w, tile_h, iw, ih):
    """/root/reference/utils/regiontools.py:15-37: thumbnail points -> level-0 top-left corners;
    drop points whose tile touches the border."""
    arr = np.asarray(arr).astype(np.int64).reshape(-1, 2).copy()
    arr *= (4 ** scan_level)
    arr -= [tile_w // 2, tile_h // 2]
    valid = (arr[:, 0] > 0) * ((arr[:, 0] + tile_w) < iw) * (arr[:, 1] > 0) * ((arr[:, 1] + tile_h) < ih)
    arr = arr[valid]
    return arr, arr.shape[0]


def build_bags(metadata, iw, ih):
    """/root/reference/utils/dataset_hr.py:239-262,274-276: per region keep iff >=8 centre and
    >=8 perimeter points survive map_points; bag = first 8 perimeter then first 8 centre corners.
    Returns [(tile_id, (16,2) int64 level-0 xy)] in metadata order."""
    first = list(metadata.keys())[0]
    scan_level = metadata[first]['scan_level']
    bags = []
    for key in metadata:
        cnt, ncnt = map_points(metadata[key]['cnt_xy'], scan_level, HR_PATCH_W, HR_PATCH_H, iw, ih)
        per, nper = map_points(metadata[key]['perim_xy'], scan_level, HR_PATCH_W, HR_PATCH_H, iw, ih)
        if ncnt >= HR_NUM_CNT_SAMPLES and nper >= HR_NUM_PERIM_SAMPLES:
            centers = np.vstack((per[:HR_NUM_PERIM_SAMPLES], cnt[:HR_NUM_CNT_SAMPLES])).astype(np.int64)
            bags.append((metadata[key]['tile_id'], centers))
    return bags


def read_bag(level1_rgb_u8, centers, level1_downsample):
    """/root/reference/utils/dataset_hr.py:282-292: 16 crops 64x64 read at HR_SCAN_LEVEL whose
    level-0 top-left corners are ``centers`` -> (16,64,64,3) u8.  Level coordinates =
    floor(level-0 / downsample) (in-memory stand-in for OpenSlide's read_region)."""
    out = np.zeros((len(centers), HR_PATCH_H, HR_PATCH_W, 3), np.uint8)
    for j, (x, y) in enumerate(centers):
        out[j] = read_tile(level1_rgb_u8, int(x // level1_downsample), int(y // level1_downsample),
                           HR_PATCH_W, HR_PATCH_H)
    return out


def paint_regions(label_shape, metadata, tile_ids, ensemble_logits, class_probs=(0., 0., 0., 0.)):
    """/root/reference/scannet.py:145-155 with the documented fix: softmax
    over the class axis of the *ensemble* logits, per-class threshold, argmax, then
    pred_mask[foreground_indices] = class."""
    pred_mask = np.zeros(label_shape, dtype=np.int64)
    p = torch.softmax(torch.as_tensor(np.asarray(ensemble_logits, dtype=np.float32)), 1)
    for cj in range(p.shape[1]):
        p[p[:, cj] < class_probs[cj], cj] = 0
    cls = torch.argmax(p, 1).numpy()
    for tj, tile_id in enumerate(np.asarray(tile_ids)):
        pred_mask[metadata[int(tile_id)]['foreground_indices']] = cls[tj]
    return pred_mask


# --------------------------------------------------------------------------- contour resampling
def evenly_spaced_points_on_a_contour(points, num_pts):
    """/root/reference/contour_ordering.py:33-60: arc-length resampling of an ordered contour
    (pinned by tests/golden/esp.npz, generated from the reference)."""
    points = np.asarray(points)
    x, y = points[:, 0], points[:, 1]
    u = np.concatenate([[0.0], np.cumsum(np.sqrt(np.diff(x) ** 2 + np.diff(y) ** 2))])
    t = np.linspace(0, u.max(), num_pts)
    return np.stack((np.interp(t, u, x), np.interp(t, u, y)), axis=1)
