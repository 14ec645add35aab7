"""Drop-in for the reference's ``models.models``: Classifier / Regressor heads (+ the training-only
gradient-reversal function), same constructor arguments and state-dict keys (``fc.0.*``, ``fc.2.*``)
as /root/reference/models/models.py:5-58.  Eval-mode forwards run on the HIP kernels
(wsi_avgpool_fc / wsi_linear); training-mode forwards use torch ops for autograd."""
import torch
from torch import nn


class ReverseLayerF(torch.autograd.Function):
    """Identity forward, gradient scaled by -p backward (training only; models.py:5-17)."""

    @staticmethod
    def forward(ctx, x, p):
        ctx.p = p
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        return -ctx.p * grad_output, None


def _hip_pooled(x):
    """(B,F,h,w) fp32 GPU feature map -> (B,F) mean features through the HIP avgpool kernel."""
    import ctypes as C
    from wsi_segmentation_pipeline_amd import native, engine as E
    if not x.is_cuda:
        raise RuntimeError('eval-mode heads run on HIP kernels only (no CPU fallback): got a CPU tensor')
    if x.shape[1] % 64:
        raise ValueError('feature count must be a multiple of 64')
    lib = native.load()
    b, f, h, w = x.shape
    feat = torch.empty((b, f), dtype=torch.float32, device=x.device)
    buf = E.pf_pack(x, 2)
    native.check(lib.wsi_avgpool_fc(buf.data_ptr(), b, h, w, f, None, None, 0, feat.data_ptr(), None, 2,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'wsi_avgpool_fc')
    return feat


def _hip_linear(x, lin, relu=False):
    import ctypes as C
    from wsi_segmentation_pipeline_amd import native
    lib = native.load()
    w = lin.weight.detach().to(torch.float32).contiguous()
    bias = lin.bias.detach().to(torch.float32).contiguous()
    y = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
    native.check(lib.wsi_linear(x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1],
                                w.shape[0], int(relu), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'wsi_linear')
    return y


class Classifier(nn.Module):
    def __init__(self, num_features, num_classes):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Sequential(nn.Linear(num_features, num_classes))

    def forward(self, x):
        if self.training:
            return self.fc(torch.flatten(self.avgpool(x), 1))
        return _hip_linear(_hip_pooled(x), self.fc[0])


class Regressor(nn.Module):
    def __init__(self, num_features, num_classes):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Sequential(nn.Linear(num_features, num_features // 4), nn.ReLU(True),
                                nn.Linear(num_features // 4, num_classes))

    def forward(self, x):
        if self.training:
            return self.fc(torch.flatten(self.avgpool(x), 1))
        return _hip_linear(_hip_linear(_hip_pooled(x), self.fc[0], relu=True), self.fc[2])
