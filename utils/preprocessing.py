"""Drop-in for the inference-side helpers of the reference's ``utils.preprocessing``
(/root/reference/utils/preprocessing.py): DotDict :50-57, isforeground :60-71, find_nuclei :74-110
(HSV mode), tile_image :113-153, threshold_probs :156-172, standard_augmentor(eval=True) :206-212,
NormalizeInverse :35-47.  Training-only statistics helpers (cls_weights*, quantize_image) are out
of the inference hot path and are not provided.

threshold_probs runs on the HIP kernel (wsi_softmax_threshold_argmax) - it needs a GPU."""
import numpy as np
import torch

from myargs import args


class DotDict(dict):
    """dict whose items read as attributes; like the reference's, assignments land in the instance
    ``__dict__`` (so `.iw = ...` after construction is visible as an attribute)."""
    __getattr__ = dict.get
    __delattr__ = dict.__delitem__

    def __setattr__(self, key, value):
        self.__dict__[key] = value

    def __setitem__(self, key, value):
        self.__dict__[key] = value


def isforeground(arr, thresh=0.05):
    """True iff at least `thresh` of the window is nonzero."""
    arr = np.asarray(arr)
    return np.count_nonzero(arr) / arr.size >= thresh


def find_nuclei(wsi, mu_percent=0.1, mode='hsv', fill_mask=False):
    """Foreground mask of a thumbnail (u8 mask of 0/1).  mode 'hsv' (the eval scripts' default): HSV saturation > mu_percent, S =
    (max - min) / max on the [0,1]-scaled RGB values, 0 where max == min (scikit-image's rgb2hsv definition, which the reference
    calls).  mode 'lab' (:88-92) and fill_mask (:101-106: fill holes, 10x10 close) run on the device for CUDA thumbnails
    (wsi_find_nuclei_lab, wsi_fill_holes + wsi_morph_rect; own deterministic spec of the absent skimage / cv2: parity unpinned)."""
    if mode not in ('hsv', 'lab'):
        raise ValueError("mode must be 'hsv' or 'lab'")
    if torch.is_tensor(wsi) and wsi.is_cuda:                       # device thumbnails stay on the device
        from wsi_segmentation_pipeline_amd import proposals as P
        mask = P.find_nuclei(wsi, mu_percent) if mode == 'hsv' else P.find_nuclei_lab(wsi, mu_percent)
        return P.fill_mask(mask) if fill_mask else mask
    if mode != 'hsv' or fill_mask:
        raise NotImplementedError("host arrays: only mode='hsv', fill_mask=False; pass a CUDA tensor for mode='lab' / fill_mask")
    rgb = np.asarray(wsi)[..., :3].astype(np.float64) / 255.0
    hi, lo = rgb.max(-1), rgb.min(-1)
    delta = hi - lo
    sat = np.divide(delta, hi, out=np.zeros_like(hi), where=hi > 0)
    sat[delta == 0.0] = 0.0
    return (sat > mu_percent).astype(np.uint8)


def tile_image(image, params):
    """Yield (x, y, crop) over a PIL/ndarray image with the reference's edge handling (:137-153)."""
    from PIL import Image
    if isinstance(image, np.ndarray):
        image = Image.fromarray(image.astype(np.uint8))
    p = DotDict(params)
    if (p.ih - 1 - p.ph) <= 0 or (p.iw - 1 - p.pw) <= 0:
        yield 0, 0, image.crop((0, 0, p.pw, p.ph))
        return
    ys, xs = range(0, p.ih - 1 - p.ph, p.sh), range(0, p.iw - 1 - p.pw, p.sw)
    corners = [(x, y) for y in ys for x in xs] + [(p.iw - 1 - p.pw, y) for y in ys] + [(x, p.ih - 1 - p.ph) for x in xs]
    for x, y in corners:
        yield x, y, image.crop((x, y, x + p.pw, y + p.ph))


def threshold_probs(pred):
    """softmax over classes, zero the probabilities under args.class_probs, argmax.
    pred: (C,H,W) ndarray or tensor -> (classes u8 ndarray, probs float64 ndarray), computed on the GPU."""
    from wsi_segmentation_pipeline_amd import engine as E
    if not torch.cuda.is_available():
        raise RuntimeError('threshold_probs runs on the HIP kernel wsi_softmax_threshold_argmax: no GPU available')
    t = torch.as_tensor(pred)
    dev = t.device if t.is_cuda else torch.device('cuda', torch.cuda.current_device())
    classes, probs, _ = E.softmax_threshold_argmax(t.to(dev, torch.float64), args.class_probs)
    return classes.cpu().numpy(), probs.cpu().numpy()


class _EvalTransform:
    """ToTensor + Normalize for one u8 HWC image (PIL or ndarray) -> fp32 CHW tensor, via the same
    3x256 LUT the device kernels use (exact for u8 input)."""

    def __init__(self, mean, std):
        from wsi_segmentation_pipeline_amd.engine import normalize_lut
        self.lut = normalize_lut(mean, std)

    def __call__(self, image):
        a = np.asarray(image)
        if a.ndim != 3 or a.shape[2] < 3 or a.dtype != np.uint8:
            raise ValueError('expected an 8-bit RGB image')
        out = np.stack([self.lut[c][a[..., c]] for c in range(3)])
        return torch.from_numpy(out)


def standard_augmentor(eval=False):
    if not eval:
        raise NotImplementedError('train-time ColorJitter augmentation is outside the inference path')
    return _EvalTransform(args.dataset_mean, args.dataset_std)


class NormalizeInverse:
    """Undo Normalize(mean, std) on a CHW tensor (returns a new tensor)."""

    def __init__(self, mean, std):
        self.mean = torch.as_tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.as_tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, tensor):
        return tensor * (self.std.to(tensor.device) + 1e-7) + self.mean.to(tensor.device)
