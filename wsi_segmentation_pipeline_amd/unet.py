"""U-Net segmentation model on the HIP path: the dense per-pixel ('seg') counterpart of the reference's
``smp.Unet('resnet18', classes=C)`` (/root/reference/eval_tumorbed.py:21-28, /root/reference/eval.py:22-27), driven by
``predict_wsis`` (utils/eval.py:51 ``model(batch_image)``) and ``predict_tumorbed(mode='seg')`` (:196-200
``model.decoder(model.encoder(batch_image))``).

segmentation_models_pytorch is third-party, absent here and un-pinned in the reference, so the architecture
and the state-dict key names are restated from its published 0.0.x source - **parity unpinned**; the numerical spec is
oracle/unet_oracle.py (torch fp32 CPU) and the HIP path is held to it:
  encoder  ResNet-18 (torchvision key names under ``encoder.``); forward returns [x4, x3, x2, x1, x0] (deepest first),
           x0 = relu(bn1(conv1(x))) at half resolution
  decoder  ``decoder.layer{1..5}.block.{0,1}.block.0.weight`` (3x3 conv, no bias) + ``.block.1.*`` (BatchNorm) + ReLU, each block
           preceded by nearest x2 upsampling and (blocks 1-4) concatenation of the next skip; channels 256/128/64/32/16;
           ``decoder.final_conv.{weight,bias}`` (1x1 to `classes`)
Eval-mode forwards run on libwsi_hip.so only (conv3x3 MFMA kernels + upsample/concat, head kernels: csrc/unet.hip); training
mode uses torch ops so ``train.py``-style callers still run.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import native
from .engine import BN_EPS, MX, PARITY, SPEED, TrunkEngine, _np_ptr, _ptr, _require_gpu, _stream

DEC_CH = (256, 128, 64, 32, 16)
SKIP_CH = (256, 128, 64, 64, 0)


def _pad64(c):
    return (c + 63) // 64 * 64


def decoder_key_shapes(classes):
    """[(key, shape, kind)] of the decoder in state-dict order."""
    out, cprev = [], 512
    for L in range(5):
        cin, cout = cprev + SKIP_CH[L], DEC_CH[L]
        for j, ci in enumerate((cin, cout)):
            p = 'decoder.layer%d.block.%d.block' % (L + 1, j)
            out.append((p + '.0.weight', (cout, ci, 3, 3), 'conv'))
            for suffix, kind in (('weight', 'bn_w'), ('bias', 'bn_b'), ('running_mean', 'bn_m'), ('running_var', 'bn_v')):
                out.append(('%s.1.%s' % (p, suffix), (cout,), kind))
            out.append((p + '.1.num_batches_tracked', (), 'bn_n'))
        cprev = cout
    out.append(('decoder.final_conv.weight', (classes, DEC_CH[4], 1, 1), 'conv'))
    out.append(('decoder.final_conv.bias', (classes,), 'lin_b'))
    return out


class UNetEngine:
    """ResNet-18 encoder (TrunkEngine) + smp-style decoder on HIP kernels.  state_dict: ``encoder.*`` + ``decoder.*`` keys."""

    def __init__(self, state_dict, device, planes=PARITY, classes=None, max_batch=None):
        self._ws = {}
        self.lib = native.load()
        self.device = torch.device(device)
        enc_sd = {k[len('encoder.'):]: v for k, v in state_dict.items() if k.startswith('encoder.')}
        self.trunk = TrunkEngine(enc_sd, device, planes=planes)
        self.planes = planes
        self.classes = int(classes if classes is not None else state_dict['decoder.final_conv.weight'].shape[0])
        self.max_batch = max_batch
        self._keep, self._ws = [], {}
        self.dw = native.WsiUnetDecoderWeights()

        def f32(key):
            return np.ascontiguousarray(state_dict[key].detach().to('cpu', torch.float32).numpy())

        def dev(a):
            t = torch.from_numpy(a).to(self.device)
            self._keep.append(t)
            return t

        cprev_real, cprev_pad = 512, 512
        for L in range(5):
            cout = DEC_CH[L]
            cout_pad = _pad64(cout) if planes == 1 else -(-cout // 32) * 32        # whole 128-byte lines: 64 channels in speed mode, 32 otherwise
            for j in range(2):
                p = 'decoder.layer%d.block.%d.block' % (L + 1, j)
                w = f32(p + '.0.weight')
                if j == 0:                                     # input = [upsampled x (real cprev of cprev_pad) | skip]
                    cin_pad = cprev_pad + SKIP_CH[L]
                    wp = np.zeros((cout_pad, cin_pad, 3, 3), np.float32)
                    wp[:cout, :cprev_real] = w[:, :cprev_real]
                    wp[:cout, cprev_pad:cprev_pad + SKIP_CH[L]] = w[:, cprev_real:]
                else:
                    cin_pad = cout_pad
                    wp = np.zeros((cout_pad, cin_pad, 3, 3), np.float32)
                    wp[:cout, :cout] = w
                bn = []
                for suffix, fill in (('weight', 1.0), ('bias', 0.0), ('running_mean', 0.0), ('running_var', 1.0)):
                    a = np.full(cout_pad, fill, np.float32)    # padding channels: scale ~1, shift 0 -> relu(0) = 0
                    a[:cout] = f32('%s.1.%s' % (p, suffix))
                    bn.append(a)
                pk = np.empty(self.lib.wsi_prepack_conv_bytes(cout_pad, cin_pad, 3, planes), np.uint8)
                bias = np.empty(cout_pad, np.float32)
                native.check(self.lib.wsi_prepack_conv(_np_ptr(wp), *[_np_ptr(a) for a in bn], BN_EPS, cout_pad, cin_pad, 3, planes,
                                                       _np_ptr(pk), _np_ptr(bias)), 'wsi_prepack_conv')
                i = 2 * L + j
                self.dw.conv_w[i], self.dw.conv_b[i] = dev(pk).data_ptr(), dev(bias).data_ptr()
                self.dw.cin[i], self.dw.cout[i] = cin_pad, cout_pad
            cprev_real, cprev_pad = cout, cout_pad
        hw = f32('decoder.final_conv.weight').reshape(self.classes, DEC_CH[4])
        self.dw.head_w = dev(np.ascontiguousarray(hw)).data_ptr()
        self.dw.head_b = dev(f32('decoder.final_conv.bias')).data_ptr()
        self.dw.head_cin, self.dw.classes = DEC_CH[4], self.classes
        self.dw.tail_w = None
        if planes == PARITY and DEC_CH[3] == 32 and DEC_CH[4] <= 16 and self.classes <= 4:
            # r05: the last block + head as one kernel (csrc/tail.hip) on the block's REAL channels
            p0, p1 = ('decoder.layer5.block.%d.block' % j for j in range(2))
            bn = [[f32('%s.1.%s' % (p, sfx)) for sfx in ('weight', 'bias', 'running_mean', 'running_var')] for p in (p0, p1)]
            blob = np.empty(self.lib.wsi_unet_tail_prepack_bytes(), np.uint8)
            hb = f32('decoder.final_conv.bias')
            native.check(self.lib.wsi_unet_tail_prepack(_np_ptr(f32(p0 + '.0.weight')), *[_np_ptr(a) for a in bn[0]],
                                                        _np_ptr(f32(p1 + '.0.weight')), *[_np_ptr(a) for a in bn[1]], BN_EPS,
                                                        _np_ptr(np.ascontiguousarray(hw)), _np_ptr(hb), DEC_CH[3], DEC_CH[4], self.classes,
                                                        _np_ptr(blob)), 'wsi_unet_tail_prepack')
            self.dw.tail_w = dev(blob).data_ptr()

    def _workspace(self, n, h, w):
        key = (h, w)
        ent = self._ws.get(key)
        if ent is None or ent[1] < n:
            nbytes = self.lib.wsi_unet_workspace_bytes(C.byref(self.dw), n, h, w, self.planes)
            if nbytes == 0:
                raise ValueError('unsupported patch shape %dx%d (need multiples of 32)' % (h, w))
            self.release_workspaces()                          # the library forgets the addresses before the allocator can reuse them
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            native.check(self.lib.wsi_unet_workspace_init(C.byref(self.dw), _ptr(ws), n, h, w, self.planes, _stream()),
                         'wsi_unet_workspace_init')
            ent = self._ws[key] = (ws, n)
        return ent

    def release_workspaces(self):
        """Free every planned workspace and make the library forget its layout tag (capi.hip g_ws_layout)."""
        for ws, _ in self._ws.values():
            self.lib.wsi_trunk_workspace_release(_ptr(ws))
        self._ws.clear()

    def __del__(self):
        try:
            self.release_workspaces()
        except Exception:                                   # interpreter shutdown: the library may be gone
            pass

    TUNED_BATCH_256 = 512                                    # tiles of 256 x 256 per U-Net call (bench.py --seg-batch default)

    def _batch(self, h, w):
        """Tiles per U-Net call: `max_batch`, else the tuned size scaled by patch area (r05 sweep at 256 x 256, parity: 128 / 256 / 512
        tiles -> 28.1 / 31.2 / 33.1 k patches/s: at 128 the 8 x 8 and 16 x 16 levels launch fewer workgroups than the chip has CUs)
        as far as half of the free HBM allows - ~77 MB of workspace per 256 x 256 tile, 39 GB of the 288 GB at 512."""
        if self.max_batch:
            return self.max_batch
        want = max(1, int(self.TUNED_BATCH_256 * 65536 // max(h * w, 1)))
        per = self.lib.wsi_unet_workspace_bytes(C.byref(self.dw), 16, h, w, self.planes) / 16.0
        if per <= 0:
            return want
        try:
            free = torch.cuda.mem_get_info(self.device)[0] + max(0, torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device))
        except Exception:
            return want
        held = sum(int(ws.numel()) for ws, _ in self._ws.values())
        return max(1, min(want, int(0.5 * (free + held) / per)))

    def _run(self, n, h, w, in_f32, slide, tile_xy, want_logits, want_enc):
        ws, cap = self._workspace(n, h, w)
        logits = torch.empty((n, self.classes, h, w), dtype=torch.float32, device=self.device) if want_logits else None
        enc = None
        enc_ptrs = None
        if want_enc:
            shapes = [(512, h // 32, w // 32), (256, h // 16, w // 16), (128, h // 8, w // 8), (64, h // 4, w // 4), (64, h // 2, w // 2)]
            enc = [torch.empty((n,) + s, dtype=torch.float32, device=self.device) for s in shapes]
            enc_ptrs = (C.c_void_p * 5)(*[t.data_ptr() for t in enc])
        if slide is not None:
            sp, pitch, sh, sw = _ptr(slide), slide.stride(0), slide.shape[0], slide.shape[1]
        else:
            sp, pitch, sh, sw = None, 0, 0, 0
        native.check(self.lib.wsi_unet_forward(C.byref(self.trunk.wt), C.byref(self.dw), _ptr(in_f32), sp, pitch, sh, sw, _ptr(tile_xy),
                                               _ptr(self.trunk.lut), n, h, w, _ptr(ws), cap, _ptr(logits),
                                               C.byref(enc_ptrs) if enc_ptrs is not None else None, _stream()), 'wsi_unet_forward')
        return logits, enc

    def forward_f32(self, x, logits=True, enc=False):
        """x (N,3,H,W) normalised fp32 GPU -> logits (N,classes,H,W) [, the five encoder maps deepest first]."""
        _require_gpu(x, 'input batch')
        x = x.to(torch.float32).contiguous()
        n, _, h, w = x.shape
        mb = self._batch(h, w)
        outs = [self._run(min(mb, n - i), h, w, x[i:i + mb], None, None, logits, enc) for i in range(0, n, mb)]
        lg = torch.cat([o[0] for o in outs]) if logits else None
        en = [torch.cat([o[1][k] for o in outs]) for k in range(5)] if enc else None
        return lg, en

    def forward_tiles(self, slide_u8, tile_xy, ph, pw):
        """Tiles read straight from an HBM-resident u8 slide -> logits (N,classes,ph,pw)."""
        _require_gpu(slide_u8, 'slide')
        tile_xy = tile_xy.to(self.device, torch.int32).contiguous()
        n = tile_xy.shape[0]
        mb = -(-n // max(1, int(np.ceil(n / self._batch(ph, pw) - 0.25))))    # equal batches, no short last one; a quarter over the tuned size is still one
        parts = [self._run(min(mb, n - i), ph, pw, None, slide_u8, tile_xy[i:i + mb], True, False)[0] for i in range(0, n, mb)]
        return parts[0] if len(parts) == 1 else torch.cat(parts)

    def decode(self, enc):
        """`model.decoder(encoding)`: five fp32 NCHW GPU maps (deepest first) -> logits."""
        for t in enc:
            _require_gpu(t, 'encoder map')
        enc = [t.to(torch.float32).contiguous() for t in enc]
        n, h, w = enc[0].shape[0], enc[0].shape[2] * 32, enc[0].shape[3] * 32
        ws, cap = self._workspace(n, h, w)
        logits = torch.empty((n, self.classes, h, w), dtype=torch.float32, device=self.device)
        ptrs = (C.c_void_p * 5)(*[t.data_ptr() for t in enc])
        native.check(self.lib.wsi_unet_decoder(C.byref(self.dw), C.byref(ptrs), n, h, w, self.planes, _ptr(ws), cap, _ptr(logits), _stream()),
                     'wsi_unet_decoder')
        return logits


# ---------------------------------------------------------------------------------------------- nn.Module surface
class _Conv2dReLU(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.block = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.block(x)


class _DecoderBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.block = nn.Sequential(_Conv2dReLU(cin, cout), _Conv2dReLU(cout, cout))

    def forward(self, x, skip=None):
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        if skip is not None:
            x = torch.cat([x, skip], 1)
        return self.block(x)


class UNetDecoder(nn.Module):
    def __init__(self, classes, owner=None):
        super().__init__()
        cprev = 512
        for L in range(5):
            setattr(self, 'layer%d' % (L + 1), _DecoderBlock(cprev + SKIP_CH[L], DEC_CH[L]))
            cprev = DEC_CH[L]
        self.final_conv = nn.Conv2d(DEC_CH[4], classes, 1)
        self._owner = [owner]                                  # list: not registered as a sub-module

    def forward(self, enc):
        if not self.training:
            return self._owner[0].hip_engine(enc[0].device).decode(list(enc))
        x = enc[0]
        skips = list(enc[1:]) + [None]
        for L in range(5):
            x = getattr(self, 'layer%d' % (L + 1))(x, skips[L])
        return self.final_conv(x)


class UNetEncoder(nn.Module):
    """ResNet-18 trunk with the smp encoder surface: forward(x) -> [x4, x3, x2, x1, x0]; ``out_shapes``."""

    def __init__(self, owner=None):
        super().__init__()
        import resnets_shift
        net = resnets_shift.resnet18(False)
        for name in ('conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2', 'layer3', 'layer4'):
            setattr(self, name, getattr(net, name))
        self.out_shapes = (512, 256, 128, 64, 64)
        self._owner = [owner]

    def forward(self, x):
        if not self.training:
            return self._owner[0].hip_engine(x.device).forward_f32(x, logits=False, enc=True)[1]
        x0 = self.relu(self.bn1(self.conv1(x)))
        x1 = self.layer1(self.maxpool(x0))
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x4 = self.layer4(x3)
        return [x4, x3, x2, x1, x0]


class UNetSeg(nn.Module):
    """Drop-in for ``smp.Unet('resnet18', classes=C, activation=None)`` as the reference uses it: ``model(x)`` -> (B,C,H,W)
    logits; ``model.encoder`` / ``model.decoder`` callable separately; ``classifier`` / ``regressor`` heads attachable."""

    def __init__(self, classes=4, precision='parity'):
        super().__init__()
        self.encoder = UNetEncoder(self)
        self.decoder = UNetDecoder(classes, self)
        self.classes, self.precision = classes, precision
        self._engine, self._engine_sig = None, None

    def hip_engine(self, device=None):
        device = torch.device(device) if device is not None else self.decoder.final_conv.weight.device
        sig = (str(device), self.precision) + tuple((p.data_ptr(), p._version) for p in self.parameters()) \
            + tuple((b.data_ptr(), b._version) for b in self.buffers())
        if self._engine is None or sig != self._engine_sig:
            self._engine = UNetEngine(self.state_dict(), device, planes={'parity': PARITY, 'mx': MX, 'speed': SPEED}[self.precision],
                                      classes=self.classes)
            self._engine_sig = sig
        return self._engine

    def forward(self, x):
        if self.training:
            return self.decoder(self.encoder(x))
        if not x.is_cuda:
            raise RuntimeError('UNetSeg eval forward runs on HIP kernels only: move the model and the input to the GPU')
        return self.hip_engine(x.device).forward_f32(x)[0]
