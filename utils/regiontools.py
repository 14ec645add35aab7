"""Inference-side piece of the reference's ``utils.regiontools``: map_points
(/root/reference/utils/regiontools.py:15-37).  Candidate generation (k-means key points) is
ranked "next" in SURVEY.md 8f - region candidates are inputs to the hot path."""
import numpy as np


def map_points(arr, params):
    """Thumbnail (x,y) points -> level-0 tile corners; drops tiles that touch the slide border.
    params: .scan_level, .tile_w, .tile_h, .iw, .ih.  Returns (points, count)."""
    pts = np.asarray(arr).astype(np.int64).reshape(-1, 2) * (4 ** params.scan_level)
    pts = pts - np.array([params.tile_w // 2, params.tile_h // 2])
    x, y = pts[:, 0], pts[:, 1]
    ok = (x > 0) & (x + params.tile_w < params.iw) & (y > 0) & (y + params.tile_h < params.ih)
    pts = pts[ok]
    return pts, pts.shape[0]
