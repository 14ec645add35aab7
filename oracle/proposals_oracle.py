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
"""
import numpy as np

from .postprocess_oracle import bwperim

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


def kmeans(points, k, iters=KMEANS_ITERS):
    """points (N,2) integer (x, y) in raster order -> (centres (k,2) float64, labels (N,) int32)."""
    pts = np.asarray(points, np.int64)
    n = len(pts)
    centres = pts[[(2 * j + 1) * n // (2 * k) for j in range(k)]].astype(np.float64)
    labels = np.full(n, -1, np.int32)
    px, py = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
    for _ in range(iters):
        dx = px[:, None] - centres[None, :, 0]
        dy = py[:, None] - centres[None, :, 1]
        d = dx * dx + dy * dy
        new = np.argmin(d, 1).astype(np.int32)               # first minimum: ties to the lower index
        if np.array_equal(new, labels):
            break
        labels = new
        for j in range(k):
            sel = labels == j
            c = int(sel.sum())
            if c:
                centres[j, 0] = float(pts[sel, 0].sum()) / float(c)
                centres[j, 1] = float(pts[sel, 1].sum()) / float(c)
    return centres, labels


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
