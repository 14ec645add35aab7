"""CPU spec (NumPy) of the region-proposal generation that feeds the bag path.

Test infrastructure only (see oracle/__init__.py).  Restates /root/reference/scannet.py:55-127 (connected components of a
ground-truth thumbnail -> key points + perimeter points per region) and /root/reference/utils/regiontools.py:68-102
(get_key_points) on top of oracle.wsi_oracle.find_nuclei_hsv.  The arithmetic the reference delegates to third-party packages
is absent and un-pinned here, so it is restated as a deterministic spec - **parity unpinned**:
  cv2.connectedComponentsWithStats (8-connectivity): label 0 = background, components numbered 1, 2, ... in raster order of
      their first pixel (the published SAUF / BBDT labelling order).
  PIL Image.resize of label / mask images: NEAREST (dst pixel i <- src pixel floor((i + 0.5) * n_src / n_dst)); the
      reference relies on PIL's version-dependent default filter.
  sklearn MiniBatchKMeans (imported as KMeans; n_clusters=k, random_state=0): mini-batch sampling, version- and RNG-dependent; replaced by Lloyd's algorithm with a
      deterministic start (the points at ranks floor((2j + 1) N / (2k)) of the raster-ordered foreground list), squared
      Euclidean distances in float64, ties to the lower cluster index, integer coordinate sums, an empty cluster keeps its
      centre, at most 25 iterations (early exit when no assignment changes).
  mahotas.bwperim: oracle.postprocess_oracle.bwperim.
  skimage.segmentation.slic (/root/reference/slic.py:43): the published algorithm of skimage's slic_superpixels.py / _slic.pyx (0.15
      line, the API the reference calls: enforce_connectivity=False, sigma, compactness, min_size_factor unused without
      connectivity) restated in slic_labels below, with two stated departures that make it deterministic on any hardware: the
      scaled Lab image is rounded to 2^-20 fixed point (cluster sums become exact integers, independent of the summation order)
      and a centre that lost all its pixels is dropped (skimage divides by zero there).  The Gaussian pre-filter is
      scipy.ndimage.gaussian_filter itself (scipy is installed: that stage is pinned by scipy).
"""
import numpy as np

from .postprocess_oracle import bwperim

SLIC_FIX = float(1 << 20)

HR_NUM_PERIM_SAMPLES = 8          # /root/reference/utils/dataset_hr.py:14-15
KMEANS_ITERS = 25


def connected_components(mask):
    """8-connected components of mask != 0: int32 labels, 0 = background, 1.. in raster order of first pixels."""
    fg = np.asarray(mask) != 0
    H, W = fg.shape
    parent = {}

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for y in range(H):
        for x in range(W):
            if not fg[y, x]:
                continue
            p = y * W + x
            parent[p] = p
            for dy, dx in ((-1, -1), (-1, 0), (-1, 1), (0, -1)):
                yy, xx = y + dy, x + dx
                if 0 <= yy < H and 0 <= xx < W and fg[yy, xx]:
                    ra, rb = find(p), find(yy * W + xx)
                    if ra != rb:
                        parent[max(ra, rb)] = min(ra, rb)
    roots = sorted({find(p) for p in parent})
    rank = {r: i + 1 for i, r in enumerate(roots)}
    out = np.zeros((H, W), np.int32)
    for p in parent:
        out[p // W, p % W] = rank[find(p)]
    return out


def resize_nearest(img, out_hw):
    img = np.asarray(img)
    ys = np.minimum(((np.arange(out_hw[0]) + 0.5) * img.shape[0] / out_hw[0]).astype(np.int64), img.shape[0] - 1)
    xs = np.minimum(((np.arange(out_hw[1]) + 0.5) * img.shape[1] / out_hw[1]).astype(np.int64), img.shape[1] - 1)
    return img[ys][:, xs]


def _lloyd(pts, centres, iters):
    """Lloyd iterations from `centres`: float64 distances, first minimum (ties to the lower index), exact integer sums, empty clusters
    keep their centre, stop when no label changes.  Returns (centres, labels, per-cluster integer sums (k,3): sum x, sum y, count) of
    the LAST assignment."""
    n, k = len(pts), len(centres)
    centres = centres.astype(np.float64).copy()
    labels = np.full(n, -1, np.int32)
    px, py = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
    sums = np.zeros((k, 3), np.int64)
    for _ in range(iters):
        dx = px[:, None] - centres[None, :, 0]
        dy = py[:, None] - centres[None, :, 1]
        d = dx * dx + dy * dy
        new = np.argmin(d, 1).astype(np.int32)               # first minimum: ties to the lower index
        changed = not np.array_equal(new, labels)
        labels = new
        for j in range(k):
            sel = labels == j
            sums[j] = (int(pts[sel, 0].sum()), int(pts[sel, 1].sum()), int(sel.sum()))
        if not changed:
            break
        for j in range(k):
            if sums[j, 2]:
                centres[j, 0] = float(sums[j, 0]) / float(sums[j, 2])
                centres[j, 1] = float(sums[j, 1]) / float(sums[j, 2])
    return centres, labels, sums


def kmeans_seed_stratified(pts, k):
    """Seeds of r01-r03: k points at the odd 2k-quantiles of the raster order."""
    n = len(pts)
    return pts[[(2 * j + 1) * n // (2 * k) for j in range(k)]].astype(np.float64)


def kmeans_seed_farthest(pts, k):
    """Farthest-point seeds in exact integer arithmetic: the point farthest from the floor-mean (sum x // n, sum y // n), then k - 1
    times the point whose squared distance to the nearest seed is largest; every tie goes to the lower raster index."""
    n = len(pts)
    px, py = pts[:, 0].astype(np.int64), pts[:, 1].astype(np.int64)
    cx, cy = int(px.sum()) // n, int(py.sum()) // n
    d0 = (px - cx) ** 2 + (py - cy) ** 2
    idx = [int(np.argmax(d0))]                                # np.argmax returns the first maximum
    dmin = (px - px[idx[0]]) ** 2 + (py - py[idx[0]]) ** 2
    for _ in range(1, k):
        j = int(np.argmax(dmin))
        idx.append(j)
        dmin = np.minimum(dmin, (px - px[j]) ** 2 + (py - py[j]) ** 2)
    return pts[idx].astype(np.float64)


def kmeans_partition_score(sums):
    """sum_j |S_j|^2 / n_j as an exact fraction: the within-cluster sum of squares of a partition is sum |p|^2 minus this, so the
    LARGER score is the better partition (compared exactly: no float rounding decides between two seedings)."""
    from fractions import Fraction
    return sum((Fraction(int(sx) * int(sx) + int(sy) * int(sy), int(c)) for sx, sy, c in sums if c), Fraction(0))


def kmeans(points, k, iters=KMEANS_ITERS):
    """points (N,2) integer (x, y) in raster order -> (centres (k,2) float64, labels (N,) int32).
    r04: Lloyd from TWO deterministic seedings - the raster-stratified one of r01-r03 and a farthest-point one - and the partition
    with the smaller within-cluster sum of squares wins (exact rational comparison, ties to the stratified seeding).  Against sklearn's
    MiniBatchKMeans on the seeded test regions this brings the objective to 0.92-1.10x sklearn's (one seeding alone: 0.92-1.63x)."""
    pts = np.asarray(points, np.int64)
    ca, la, sa = _lloyd(pts, kmeans_seed_stratified(pts, k), iters)
    cb, lb, sb = _lloyd(pts, kmeans_seed_farthest(pts, k), iters)
    if kmeans_partition_score(sb) > kmeans_partition_score(sa):
        return cb, lb
    return ca, la


def get_key_points(image, us, min_clusters):
    """/root/reference/utils/regiontools.py:68-102 -> (n, centre points (k,2) int (x,y), cluster image, foreground_indices) or 4 x None."""
    image = (np.asarray(image) != 0).astype(np.uint8)
    y, x = image.shape
    small = resize_nearest(image, (y // us, x // us))
    fg = np.nonzero(small)
    coords = np.transpose(fg)[:, ::-1]                        # (x, y) pairs, raster order
    k = min_clusters
    if k <= 1 or coords.shape[0] <= 3 * k:
        return None, None, None, None
    centres, labels = kmeans(coords, k)
    cnt_pts = (us * centres).astype(np.int64)
    out = np.zeros(small.shape, np.uint16)
    out[fg] = labels + 1
    out = resize_nearest(out, (y, x))
    return k, cnt_pts, out, np.nonzero(out)


def scannet_candidates(gt_mask, wsi_mask, us_kmeans=4):
    """/root/reference/scannet.py:55-127: metadata dict {patch_id: {cnt_xy, perim_xy, foreground_indices, tile_id}} from the
    ground-truth thumbnail `gt_mask` (components of gt_mask > 0) and the colour-thresholded tissue mask `wsi_mask`.
    The reference's loop quirks are kept: `for tile_id in range(labels.max())` visits the background label 0 and never the
    last component; a region wider than 5 % of the image is split into its k-means clusters."""
    labels = connected_components(np.asarray(gt_mask) > 0)
    wsi_mask = np.asarray(wsi_mask)
    metadata, patch_id = {}, 0

    def perim_points(patch):
        pc = np.transpose(np.where(bwperim(patch)))[:, ::-1]
        skip = np.maximum(2, pc.shape[0] // HR_NUM_PERIM_SAMPLES)
        return pc[::skip, :]
    for tile_id in range(int(labels.max())):
        patch = labels == tile_id
        area = np.count_nonzero(patch)
        k = 2 + int(area / (0.01 * labels.size))
        n, cnt, out_image, fgi = get_key_points(patch, us_kmeans, k)
        idx = np.where(patch)
        if len(idx[0]) == 0:
            continue
        h = 1 + idx[0].max() - idx[0].min()
        w = 1 + idx[1].max() - idx[1].min()
        if n is not None and (w * h) / labels.size <= 0.05:
            metadata[patch_id] = {'cnt_xy': cnt, 'perim_xy': perim_points(patch), 'scan_level': 2, 'foreground_indices': fgi, 'tile_id': patch_id}
            patch_id += 1
        elif n is not None:
            for r_id in range(1, n + 1):
                sub = out_image == r_id
                sn, scnt, _, sfgi = get_key_points(sub, us_kmeans, k)
                if sn is None or (tile_id == 0 and np.count_nonzero(wsi_mask[sfgi]) / sfgi[0].shape[0] < 0.5):
                    continue
                metadata[patch_id] = {'cnt_xy': scnt, 'perim_xy': perim_points(sub), 'scan_level': 2, 'foreground_indices': sfgi, 'tile_id': patch_id}
                patch_id += 1
    return metadata


# ------------------------------------------------------------------------------------------ SLIC (slic.py:43-75)
def regular_grid(ar_shape, n_points):
    """skimage.util.regular_grid: slices that sample ~n_points of an array on a regular grid."""
    ar_shape = np.asanyarray(ar_shape)
    ndim = len(ar_shape)
    unsort_dim_idxs = np.argsort(np.argsort(ar_shape))
    sorted_dims = np.sort(ar_shape)
    space_size = float(np.prod(ar_shape))
    if space_size <= n_points:
        return [slice(None)] * ndim
    stepsizes = (space_size / n_points) ** (1.0 / ndim) * np.ones(ndim)
    if (sorted_dims < stepsizes).any():
        for dim in range(ndim):
            stepsizes[dim] = sorted_dims[dim]
            space_size = float(np.prod(sorted_dims[dim + 1:]))
            stepsizes[dim + 1:] = ((space_size / n_points) ** (1.0 / (ndim - dim - 1)))
            if (sorted_dims >= stepsizes).all():
                break
    starts = (stepsizes // 2).astype(int)
    stepsizes = np.round(stepsizes).astype(int)
    slices = [slice(start, None, step) for start, step in zip(starts, stepsizes)]
    return [slices[i] for i in unsort_dim_idxs]


def rgb2lab(rgb):
    """skimage.color.rgb2lab (sRGB, illuminant D65, 2 degree observer) of a float image in [0, 1]."""
    arr = np.asarray(rgb, np.float64).copy()
    mask = arr > 0.04045
    arr[mask] = np.power((arr[mask] + 0.055) / 1.055, 2.4)
    arr[~mask] /= 12.92
    M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    xyz = np.stack([(arr[..., 0] * M[r, 0] + arr[..., 1] * M[r, 1]) + arr[..., 2] * M[r, 2] for r in range(3)], -1)
    xyz = xyz / np.array([0.95047, 1.0, 1.08883])
    mask = xyz > 0.008856
    out = np.where(mask, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    x, y, z = out[..., 0], out[..., 1], out[..., 2]
    return np.stack([116.0 * y - 16.0, 500.0 * (x - y), 200.0 * (y - z)], -1)


def slic_setup(h, w, n_segments):
    """Initial cluster centres (K, 6) = {cy, cx, 0, 0, 0, alive} on skimage's regular grid of a (1, h, w) volume + the grid steps."""
    slices = regular_grid((1, h, w), n_segments)
    step_z, step_y, step_x = [int(s.step if s.step is not None else 1) for s in slices]
    gy, gx = np.mgrid[:h, :w]
    sy, sx = gy[slices[1], slices[2]], gx[slices[1], slices[2]]
    segs = np.zeros((sy.size, 6), np.float64)
    segs[:, 0], segs[:, 1], segs[:, 5] = sy.ravel(), sx.ravel(), 1.0
    return segs, step_y, step_x, float(max(step_z, step_y, step_x))


def gaussian_weights(sigma, truncate=4.0):
    """scipy.ndimage._gaussian_kernel1d(sigma, 0, int(truncate * sigma + 0.5))."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    return phi / phi.sum(), radius


def slic_labels(rgb_u8, n_segments=200, compactness=20.0, sigma=5.0, max_iter=10):
    """labels (H, W) int32 of skimage.segmentation.slic(img_as_float(rgb), n_segments, compactness, sigma=sigma,
    enforce_connectivity=False) as specified in the module header."""
    from scipy import ndimage as ndi
    img = np.asarray(rgb_u8)[..., :3].astype(np.float64) / 255.0                     # img_as_float
    h, w = img.shape[:2]
    vol = img[None]
    if sigma > 0:
        vol = ndi.gaussian_filter(vol, [sigma, sigma, sigma, 0])
    lab = rgb2lab(vol[0]) * (1.0 / compactness)
    q = np.rint(lab * SLIC_FIX).astype(np.int64)                                       # [spec] 2^-20 fixed point
    col = q.astype(np.float64) / SLIC_FIX
    segs, step_y, step_x, step = slic_setup(h, w, n_segments)
    K = len(segs)
    spatial_weight = 1.0 / (step * step)
    ys, xs = np.mgrid[:h, :w]
    labels = np.zeros((h, w), np.int32)
    for _ in range(max_iter):
        dist = np.full((h, w), np.finfo(np.float64).max)
        for k in range(K):                                                             # increasing k + strict '>': ties to the lower centre
            cy, cx, alive = segs[k, 0], segs[k, 1], segs[k, 5]
            if not alive:
                continue
            y0, y1 = int(max(cy - 2 * step_y, 0.0)), int(min(cy + 2 * step_y + 1, float(h)))
            x0, x1 = int(max(cx - 2 * step_x, 0.0)), int(min(cx + 2 * step_x + 1, float(w)))
            if y1 <= y0 or x1 <= x0:
                continue
            dy = (cy - ys[y0:y1, x0:x1]) ** 2
            dx = (cx - xs[y0:y1, x0:x1]) ** 2
            d = ((0.0 + dy) + dx) * spatial_weight
            c = col[y0:y1, x0:x1]
            dc = (c[..., 0] - segs[k, 2]) ** 2
            dc = dc + (c[..., 1] - segs[k, 3]) ** 2
            dc = dc + (c[..., 2] - segs[k, 4]) ** 2
            d = d + dc
            win = dist[y0:y1, x0:x1]
            upd = win > d
            win[upd] = d[upd]
            labels[y0:y1, x0:x1][upd] = k
        cnt = np.bincount(labels.ravel(), minlength=K)
        for k in range(K):
            if cnt[k] == 0:
                segs[k, 5] = 0.0                                                       # [spec] an emptied centre is dropped
                continue
            sel = labels == k
            segs[k, 0] = float(ys[sel].sum()) / float(cnt[k])
            segs[k, 1] = float(xs[sel].sum()) / float(cnt[k])
            for c in range(3):
                segs[k, 2 + c] = (float(int(q[..., c][sel].sum())) / SLIC_FIX) / float(cnt[k])
    return labels


def slic_candidates(thumb_rgb_u8, out_hw, n_segments=200, compactness=20.0, sigma=5.0, us_kmeans=4, n_cnt=8):
    """/root/reference/slic.py:43-75: superpixels of the small thumbnail -> label image at `out_hw` (PIL nearest resize of the
    uint16 label image) -> per label below labels.max() (the reference's range) key points and perimeter points.  Labels whose
    key points cannot be computed (get_key_points -> None: too few pixels) are skipped - the reference would store None there
    and fail in Dataset_eval.  Returns (labels at out_hw, metadata)."""
    labels = resize_nearest(slic_labels(thumb_rgb_u8, n_segments, compactness, sigma).astype(np.uint16), out_hw)
    metadata = {}
    for tile_id in range(int(labels.max())):
        patch = labels == tile_id
        n, cnt, _, fgi = get_key_points(patch, us_kmeans, n_cnt)
        if n is None:
            continue
        pc = np.transpose(np.where(bwperim(patch)))[:, ::-1]
        skip = np.maximum(2, pc.shape[0] // HR_NUM_PERIM_SAMPLES)
        metadata[tile_id] = {'cnt_xy': cnt, 'perim_xy': pc[::skip, :], 'scan_level': 2, 'foreground_indices': fgi, 'tile_id': tile_id}
    return labels, metadata
