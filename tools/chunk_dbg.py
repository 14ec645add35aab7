import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import weights as W
from wsi_segmentation_pipeline_amd import native
from wsi_segmentation_pipeline_amd.engine import TrunkEngine
lib = native.load(); dev = torch.device('cuda:0')
sd = W.make_resnet18_state_dict(11)
n = 16
u8 = W.make_u8_patches(5, (n, 3, 256, 256))
strip = np.ascontiguousarray(u8.transpose(0, 2, 3, 1).reshape(-1, 256, 3))
xy = np.stack((np.zeros(n, np.int32), np.arange(n, dtype=np.int32) * 256), 1)
sl, xyd = torch.from_numpy(strip).to(dev), torch.from_numpy(xy).to(dev)
for planes in (1, 2, 3):
    eng = TrunkEngine(sd, dev, planes=planes, head=(sd['fc0.weight'], sd['fc0.bias']))
    base = eng.forward_tiles(sl, xyd, 256, 256, logits=True)[1].clone()
    for cs, c1 in ((8, 8), (0, 8), (8, 0), (4, 16)):
        lib.wsi_trunk_set_chunks(cs, c1)
        try:
            got = eng.forward_tiles(sl, xyd, 256, 256, logits=True)[1]
            print(planes, cs, c1, 'ok', float((got - base).abs().max()))
        except Exception as e:
            print(planes, cs, c1, 'FAIL', e)
        lib.wsi_trunk_set_chunks(0, 0)
    for mode in (128,):
        lib.wsi_conv_set_mode(1 + mode)
        try:
            got = eng.forward_tiles(sl, xyd, 256, 256, logits=True)[1]
            print(planes, 'mode+128 ok', float((got - base).abs().max()))
        except Exception as e:
            print(planes, 'mode+128 FAIL', e)
        lib.wsi_conv_set_mode(1)
