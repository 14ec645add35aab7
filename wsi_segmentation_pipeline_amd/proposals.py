"""Region-proposal generation on the device: foreground mask, connected components, key points and
perimeter points per candidate region - the `metadata` the bag path consumes (reference scannet.py:55-127,
utils/regiontools.py:68-102, utils/preprocessing.py:74-110).  Spec: oracle/proposals_oracle.py (parity unpinned where the
reference calls cv2 / PIL / sklearn / mahotas; see its header).  The per-pixel work (mask, labelling, k-means assignment,
perimeter) runs in HIP kernels; the host only walks the list of regions."""
import ctypes as C

import numpy as np
import torch

from . import native
from . import postprocess as PP
from .engine import _ptr, _require_gpu, _stream

HR_NUM_PERIM_SAMPLES = 8
KMEANS_ITERS = 25


def find_nuclei(rgb_u8, mu_percent=0.1):
    """(H,W,3|4) uint8 GPU image -> uint8 0/1 mask of HSV saturation > mu_percent (reference find_nuclei, mode 'hsv')."""
    lib = native.load()
    _require_gpu(rgb_u8, 'thumbnail')
    if rgb_u8.dtype != torch.uint8 or rgb_u8.dim() != 3 or rgb_u8.shape[2] < 3:
        raise ValueError('expected an (H,W,3) uint8 image')
    img = rgb_u8.contiguous()
    h, w, c = img.shape
    mask = torch.empty((h, w), dtype=torch.uint8, device=img.device)
    native.check(lib.wsi_find_nuclei_hsv(_ptr(img), h * w, c, float(mu_percent), _ptr(mask), _stream()), 'wsi_find_nuclei_hsv')
    return mask


def find_nuclei_lab(rgb_u8, mu_percent=0.1):
    """reference find_nuclei(mode='lab') on the device: Lab `a` channel above (1 + mu_percent) x its mean (wsi_find_nuclei_lab)."""
    lib = native.load()
    _require_gpu(rgb_u8, 'thumbnail')
    if rgb_u8.dtype != torch.uint8 or rgb_u8.dim() != 3 or rgb_u8.shape[2] < 3:
        raise ValueError('expected an (H,W,3) uint8 image')
    img = rgb_u8.contiguous()
    h, w, c = img.shape
    mask = torch.empty((h, w), dtype=torch.uint8, device=img.device)
    scratch = torch.empty(16 + 4 * h * w, dtype=torch.uint8, device=img.device)
    native.check(lib.wsi_find_nuclei_lab(_ptr(img), h * w, c, float(mu_percent), _ptr(mask), _ptr(scratch), _stream()), 'wsi_find_nuclei_lab')
    return mask


def fill_mask(mask, kernel_size=10):
    """reference find_nuclei(fill_mask=True) tail on the device: binary_fill_holes (wsi_fill_holes) + MORPH_CLOSE kernel_size^2."""
    lib = native.load()
    _require_gpu(mask, 'mask')
    m = (mask != 0).to(torch.uint8).contiguous()
    h, w = m.shape
    out = torch.empty_like(m)
    scratch = torch.empty(lib.wsi_fill_holes_scratch_bytes(h, w), dtype=torch.uint8, device=m.device)
    native.check(lib.wsi_fill_holes(_ptr(m), h, w, _ptr(out), _ptr(scratch), _stream()), 'wsi_fill_holes')
    return PP.morph_rect(PP.morph_rect(out, kernel_size, 'dilate'), kernel_size, 'erode')


def connected_components(mask):
    """uint8 / bool (H,W) GPU mask -> (int32 labels, count): 8-connected, numbered in raster order of first pixels."""
    lib = native.load()
    _require_gpu(mask, 'mask')
    m = (mask != 0).to(torch.uint8).contiguous()
    h, w = m.shape
    labels = torch.empty((h, w), dtype=torch.int32, device=m.device)
    count = torch.zeros(1, dtype=torch.int32, device=m.device)
    scratch = torch.empty(lib.wsi_connected_components_scratch_bytes(h, w), dtype=torch.uint8, device=m.device)
    native.check(lib.wsi_connected_components(_ptr(m), h, w, _ptr(labels), _ptr(count), _ptr(scratch), _stream()), 'wsi_connected_components')
    return labels, int(count.item())


def resize_nearest_idx(n_src, n_dst, device):
    return torch.clamp(((torch.arange(n_dst, device=device, dtype=torch.float64) + 0.5) * n_src / n_dst).to(torch.int64), max=n_src - 1)


def _resize_nearest(img, out_hw):
    ys = resize_nearest_idx(img.shape[0], out_hw[0], img.device)
    xs = resize_nearest_idx(img.shape[1], out_hw[1], img.device)
    return img[ys][:, xs]


def _partition_score(sums_host):
    """sum_j |S_j|^2 / n_j of a partition as an exact fraction (the larger, the smaller its within-cluster sum of squares)."""
    from fractions import Fraction
    return sum((Fraction(int(sx) * int(sx) + int(sy) * int(sy), int(c)) for sx, sy, c in sums_host if c), Fraction(0))


def kmeans(points_xy, k, iters=KMEANS_ITERS):
    """(N,2) int32 GPU points (raster order) -> (centres (k,2) float64, labels (N,) int32) by the deterministic Lloyd spec:
    two runs on the device - from the raster-stratified seeds of r01-r03 and from farthest-point seeds (wsi_kmeans_seed_farthest) -
    and the partition with the smaller within-cluster sum of squares wins (exact comparison of the kernels' integer sums on the
    host: 3 k numbers per run; ties to the stratified run).  oracle/proposals_oracle.py kmeans is the same rule."""
    lib = native.load()
    pts = points_xy.to(torch.int32).contiguous()
    n = pts.shape[0]
    runs = []
    for seeding in ('stratified', 'farthest'):
        if seeding == 'stratified':
            init = torch.tensor([(2 * j + 1) * n // (2 * k) for j in range(k)], device=pts.device)
            centres = pts[init].to(torch.float64).contiguous()
        else:
            centres = torch.empty((k, 2), dtype=torch.float64, device=pts.device)
            dmin = torch.empty(n, dtype=torch.int64, device=pts.device)
            native.check(lib.wsi_kmeans_seed_farthest(_ptr(pts), n, k, _ptr(centres), _ptr(dmin), _stream()), 'wsi_kmeans_seed_farthest')
        labels = torch.empty(n, dtype=torch.int32, device=pts.device)
        scratch = torch.empty(3 * k + 1, dtype=torch.int64, device=pts.device)
        native.check(lib.wsi_kmeans_points(_ptr(pts), n, _ptr(centres), k, iters, _ptr(labels), _ptr(scratch), _stream()), 'wsi_kmeans_points')
        runs.append((centres, labels, _partition_score(scratch[:3 * k].view(k, 3).cpu().tolist())))
    return runs[1][:2] if runs[1][2] > runs[0][2] else runs[0][:2]


def get_key_points(patch, us, min_clusters):
    """reference utils/regiontools.py:68-102 on a bool/uint8 (H,W) GPU patch -> (n, cnt_pts (k,2) int64 ndarray, cluster image
    (H,W) uint16 GPU tensor, foreground_indices (tuple of ndarrays)) or 4 x None."""
    img = (patch != 0).to(torch.uint8)
    y, x = img.shape
    small = _resize_nearest(img, (y // us, x // us))
    fg = torch.nonzero(small)                                   # raster order, (row, col)
    k = int(min_clusters)
    if k <= 1 or fg.shape[0] <= 3 * k:
        return None, None, None, None
    coords = fg.flip(1).to(torch.int32).contiguous()            # (x, y)
    centres, labels = kmeans(coords, k)
    cnt_pts = (us * centres.cpu().numpy()).astype(np.int64)
    out = torch.zeros(small.shape, dtype=torch.int32, device=img.device)
    out[fg[:, 0], fg[:, 1]] = labels + 1
    out = _resize_nearest(out, (y, x))
    nz = torch.nonzero(out)
    return k, cnt_pts, out, (nz[:, 0].cpu().numpy(), nz[:, 1].cpu().numpy())


def _perim_points(patch):
    per = PP.bwperim(patch.to(torch.uint8))
    pc = torch.nonzero(per).flip(1).cpu().numpy()               # (x, y) pairs in np.where order
    skip = max(2, pc.shape[0] // HR_NUM_PERIM_SAMPLES)
    return pc[::skip, :]


def scannet_candidates(gt_mask, wsi_mask, us_kmeans=4):
    """reference scannet.py:55-127 on the device: `metadata` {patch_id: {cnt_xy, perim_xy, scan_level, foreground_indices, tile_id}}
    from the ground-truth thumbnail and the tissue mask (both (H,W) GPU tensors)."""
    _require_gpu(gt_mask, 'ground-truth thumbnail')
    labels, _ = connected_components(gt_mask > 0)
    nlab = int(labels.max().item())
    size = labels.numel()
    wsi_mask = wsi_mask.to(labels.device)
    metadata, patch_id = {}, 0
    for tile_id in range(nlab):                                 # the reference's range: background label 0 first, last label never
        patch = labels == tile_id
        area = int(patch.sum().item())
        if area == 0:
            continue
        k = 2 + int(area / (0.01 * size))
        n, cnt, out_image, fgi = get_key_points(patch, us_kmeans, k)
        rows = torch.nonzero(patch.any(1)).view(-1)
        cols = torch.nonzero(patch.any(0)).view(-1)
        h = 1 + int(rows[-1] - rows[0])
        w = 1 + int(cols[-1] - cols[0])
        if n is not None and (w * h) / size <= 0.05:
            metadata[patch_id] = {'cnt_xy': cnt, 'perim_xy': _perim_points(patch), 'scan_level': 2, 'foreground_indices': fgi, 'tile_id': patch_id}
            patch_id += 1
        elif n is not None:
            for r_id in range(1, n + 1):
                sub = out_image == r_id
                sn, scnt, _, sfgi = get_key_points(sub, us_kmeans, k)
                if sn is None:
                    continue
                if tile_id == 0:
                    inside = int((wsi_mask[torch.from_numpy(sfgi[0]).to(labels.device), torch.from_numpy(sfgi[1]).to(labels.device)] != 0).sum().item())
                    if inside / sfgi[0].shape[0] < 0.5:
                        continue
                metadata[patch_id] = {'cnt_xy': scnt, 'perim_xy': _perim_points(sub), 'scan_level': 2, 'foreground_indices': sfgi, 'tile_id': patch_id}
                patch_id += 1
    return metadata


# ------------------------------------------------------------------------------------------ SLIC candidates (reference slic.py:43-75)
def _regular_grid(ar_shape, n_points):
    """skimage.util.regular_grid (host arithmetic on three numbers)."""
    ar_shape = np.asanyarray(ar_shape)
    ndim = len(ar_shape)
    unsort = np.argsort(np.argsort(ar_shape))
    sorted_dims = np.sort(ar_shape)
    space = float(np.prod(ar_shape))
    if space <= n_points:
        return [slice(None)] * ndim
    steps = (space / n_points) ** (1.0 / ndim) * np.ones(ndim)
    if (sorted_dims < steps).any():
        for dim in range(ndim):
            steps[dim] = sorted_dims[dim]
            space = float(np.prod(sorted_dims[dim + 1:]))
            steps[dim + 1:] = ((space / n_points) ** (1.0 / (ndim - dim - 1)))
            if (sorted_dims >= steps).all():
                break
    starts = (steps // 2).astype(int)
    steps = np.round(steps).astype(int)
    slices = [slice(a, None, b) for a, b in zip(starts, steps)]
    return [slices[i] for i in unsort]


def slic(rgb_u8, n_segments=200, compactness=20.0, sigma=5.0, max_iter=10):
    """skimage.segmentation.slic(img_as_float(rgb), n_segments, compactness, sigma=sigma, enforce_connectivity=False) of an (H,W,3)
    uint8 GPU image on the device (wsi_slic; specification: oracle/proposals_oracle.py slic_labels - own deterministic spec of the
    published algorithm, parity unpinned).  Returns int32 labels (H,W)."""
    lib = native.load()
    _require_gpu(rgb_u8, 'thumbnail')
    if rgb_u8.dtype != torch.uint8 or rgb_u8.dim() != 3 or rgb_u8.shape[2] < 3:
        raise ValueError('expected an (H,W,3) uint8 image')
    img = rgb_u8[..., :3].contiguous()
    h, w = int(img.shape[0]), int(img.shape[1])
    sl = _regular_grid((1, h, w), n_segments)
    step_z, step_y, step_x = [int(s.step if s.step is not None else 1) for s in sl]
    gy, gx = np.mgrid[:h, :w]
    sy, sx = gy[sl[1], sl[2]], gx[sl[1], sl[2]]
    segs = np.zeros((sy.size, 6), np.float64)
    segs[:, 0], segs[:, 1], segs[:, 5] = sy.ravel(), sx.ravel(), 1.0
    k = len(segs)
    radius, fw = 0, None
    if sigma > 0:                                                # scipy.ndimage._gaussian_kernel1d, truncate = 4
        radius = int(4.0 * float(sigma) + 0.5)
        xk = np.arange(-radius, radius + 1)
        phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * xk ** 2)
        fw = torch.from_numpy(phi / phi.sum()).to(img.device)
    segs_d = torch.from_numpy(segs).to(img.device)
    labels = torch.empty((h, w), dtype=torch.int32, device=img.device)
    nbytes = lib.wsi_slic_scratch_bytes(h, w, k)
    if nbytes == 0:
        raise ValueError('slic: %d segments on a %dx%d image is outside the kernel\'s range' % (k, h, w))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=img.device)
    native.check(lib.wsi_slic(_ptr(img), h, w, _ptr(fw) if fw is not None else None, radius, _ptr(segs_d), k, step_y, step_x,
                              float(max(step_z, step_y, step_x)), float(compactness), int(max_iter), _ptr(labels), _ptr(scratch), _stream()),
                 'wsi_slic')
    return labels


def slic_candidates(thumb_rgb_u8, out_hw, n_segments=200, compactness=20.0, sigma=5.0, us_kmeans=4, n_cnt=8):
    """reference slic.py:43-75 on the device: superpixel labels of the small thumbnail, nearest-resized to `out_hw`, then per label
    below labels.max() the key points (get_key_points) and perimeter points.  Returns (labels (H,W) int32 GPU tensor, metadata)."""
    labels = _resize_nearest(slic(thumb_rgb_u8, n_segments, compactness, sigma), out_hw)
    metadata = {}
    for tile_id in range(int(labels.max().item())):
        patch = labels == tile_id
        n, cnt, _, fgi = get_key_points(patch, us_kmeans, n_cnt)
        if n is None:
            continue
        metadata[tile_id] = {'cnt_xy': cnt, 'perim_xy': _perim_points(patch), 'scan_level': 2, 'foreground_indices': fgi, 'tile_id': tile_id}
    return labels, metadata
