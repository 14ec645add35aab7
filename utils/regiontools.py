"""Inference-side piece of the reference's ``utils.regiontools``: map_points
(/root/reference/utils/regiontools.py:15-37) and get_key_points (:68-102) on the device."""
import numpy as np


def get_key_points(image, us, min_clusters, max_clusters=None):
    """Centre points of a region by k-means on its downsampled foreground (reference :68-102; `max_clusters` is accepted and
    unused, as in the reference).  Runs on the HIP kernels of wsi_segmentation_pipeline_amd.proposals; `image` may be a GPU
    tensor or an ndarray (uploaded).  Returns (n, (k,2) int centre points, cluster image ndarray, foreground_indices) or 4 x None.
    The clustering is the deterministic Lloyd spec of oracle/proposals_oracle.py, not sklearn's RNG-dependent mini-batch k-means."""
    import torch
    from wsi_segmentation_pipeline_amd import proposals as P
    if not torch.cuda.is_available():
        raise RuntimeError('get_key_points runs on the HIP kernels (wsi_kmeans_points): no GPU available')
    t = image if torch.is_tensor(image) else torch.from_numpy(np.ascontiguousarray(np.asarray(image) != 0).astype(np.uint8))
    if not t.is_cuda:
        t = t.to(torch.device('cuda', torch.cuda.current_device()))
    n, cnt, out, fgi = P.get_key_points(t, us, min_clusters)
    return n, cnt, (None if out is None else out.cpu().numpy().astype(np.uint16)), fgi


def map_points(arr, params):
    """Thumbnail (x,y) points -> level-0 tile corners; drops tiles that touch the slide border.
    params: .scan_level, .tile_w, .tile_h, .iw, .ih.  Returns (points, count)."""
    pts = np.asarray(arr).astype(np.int64).reshape(-1, 2) * (4 ** params.scan_level)
    pts = pts - np.array([params.tile_w // 2, params.tile_h // 2])
    x, y = pts[:, 0], pts[:, 1]
    ok = (x > 0) & (x + params.tile_w < params.iw) & (y > 0) & (y + params.tile_h < params.ih)
    pts = pts[ok]
    return pts, pts.shape[0]
