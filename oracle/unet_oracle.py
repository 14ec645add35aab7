"""CPU spec (PyTorch fp32, functional) of the dense 'seg' model: ResNet-18 encoder + smp-style U-Net decoder.

Test infrastructure only (see oracle/__init__.py).  The reference builds this model from the THIRD-PARTY package
segmentation_models_pytorch (`smp.Unet('resnet18', ...)`, /root/reference/eval_tumorbed.py:21-28, eval.py:22-27), which is
absent here and un-pinned (no requirements file; the API the reference uses - `encoder.out_shapes`, callable `activation=`,
`model.encoder(x)` returning a list deepest-first - is the 0.0.x line of 2019).  The reference holds no tests or fixtures at that
boundary, so this file restates the package's PUBLISHED architecture as the spec - **parity unpinned**:
  encoder (ResNetEncoder): x0 = relu(bn1(conv1(x))); x1 = layer1(maxpool(x0)); x2..x4 = layer2..4 -> [x4, x3, x2, x1, x0]
  decoder (UnetDecoder, decoder_channels (256, 128, 64, 32, 16)): five DecoderBlocks, each
      x = F.interpolate(x, scale_factor=2, mode='nearest'); x = cat([x, skip], 1) (no skip for the last);
      x = relu(bn(conv3x3(x))); x = relu(bn(conv3x3(x)))        (Conv2dReLU: conv without bias, BatchNorm, ReLU)
    then final_conv (1x1, bias) -> `classes` logits at the input resolution.
The encoder part is the pinned resnet_oracle.trunk (goldens from the reference's own resnets_shift.py).
"""
import torch
import torch.nn.functional as F

from . import resnet_oracle as R


def encoder(sd, x):
    """sd: encoder keys WITHOUT the 'encoder.' prefix.  Returns [x4, x3, x2, x1, x0]."""
    taps = {}
    R.trunk(sd, x, taps)
    return [taps['layer4.1'], taps['layer3.1'], taps['layer2.1'], taps['layer1.1'], taps['stem']]


def decoder(sd, enc):
    """sd: full state dict ('decoder.*' keys).  enc: [x4, x3, x2, x1, x0]."""
    x = enc[0]
    skips = list(enc[1:]) + [None]
    for L in range(5):
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        if skips[L] is not None:
            x = torch.cat([x, skips[L]], 1)
        for j in range(2):
            p = 'decoder.layer%d.block.%d.block' % (L + 1, j)
            x = F.conv2d(x, sd[p + '.0.weight'], None, 1, 1)
            x = F.relu(F.batch_norm(x, sd[p + '.1.running_mean'], sd[p + '.1.running_var'], sd[p + '.1.weight'], sd[p + '.1.bias'],
                                    False, 0.0, R.BN_EPS))
    return F.conv2d(x, sd['decoder.final_conv.weight'], sd['decoder.final_conv.bias'])


def unet_forward(sd, x):
    """(N,3,H,W) normalised fp32 -> (N,classes,H,W) logits."""
    enc_sd = {k[len('encoder.'):]: v for k, v in sd.items() if k.startswith('encoder.')}
    return decoder(sd, encoder(enc_sd, x))
