"""Drop-in for the piece of the reference's ``contour_ordering`` that the tumour-bed post-process uses
(/root/reference/contour_ordering.py:33-60): ``evenly_spaced_points_on_a_contour``.  Arc-length resampling of an ordered
contour runs on the device (``wsi_esp``: cumulative chord length, then linear interpolation at num_pts evenly spaced
stations, float64 like numpy's diff / cumsum / linspace / interp chain).  The other helpers of that file (angle sort,
`interparc` matrix utilities) are not on the inference path and are not provided."""
import numpy as np
import torch


def evenly_spaced_points_on_a_contour(points, num_pts):
    """(N,2) ordered contour points (ndarray, list or CUDA tensor) -> (num_pts,2) float64 points, evenly spaced along the
    polyline.  Returns the type it was given (ndarray for array-likes, CUDA tensor for tensors)."""
    from wsi_segmentation_pipeline_amd import postprocess as PP
    if not torch.cuda.is_available():
        raise RuntimeError('evenly_spaced_points_on_a_contour runs on the HIP kernel wsi_esp: no GPU available')
    if torch.is_tensor(points):
        t = points if points.is_cuda else points.cuda()
        return PP.esp(t, num_pts)
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(points, np.float64))).cuda()
    return PP.esp(t, num_pts).cpu().numpy()
