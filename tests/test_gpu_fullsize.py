"""Full-size runs (BASELINE.json configs 3 and 4) checked through size-independent properties,
plus oracle spot checks on a handful of tiles: the CPU oracle cannot cover 24 648 tiles in seconds."""
import numpy as np
import pytest
import torch

from oracle import resnet_oracle as R
from oracle import weights as W

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def test_cfg3_40k_slide_dense_sliding_window(dev):
    from wsi_segmentation_pipeline_amd import slide as S
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    sd = W.make_resnet18_state_dict(11, with_fc=False)
    cls = W.make_head_state_dict(22, 'classifier')
    size, tile = 40000, 256
    g = torch.Generator(device=dev).manual_seed(3)
    level0 = torch.randint(0, 256, (size, size, 3), dtype=torch.uint8, device=dev, generator=g)      # 4.8 GB in HBM
    tiles = S.tile_grid(size, size, tile, tile, tile, tile)
    assert len(tiles) == 24648
    eng = TrunkEngine(sd, dev, head=(cls['fc.0.weight'], cls['fc.0.bias']), max_batch=1000)
    m, map_hw = 1.0 / 16.0, (2500, 2500)
    out1 = S.infer_slide_cls(eng, level0, tiles, tile, tile, m, map_hw, 4, (0., 0., 0., 0.))
    out2 = S.infer_slide_cls(eng, level0, tiles, tile, tile, m, map_hw, 4, (0., 0., 0., 0.))
    # (a) idempotence / determinism: bit-identical
    for k in ('logits', 'pred', 'classes', 'heatmap'):
        assert torch.equal(out1[k], out2[k]), k
    logits = out1['logits']
    assert logits.shape == (24648, 4) and bool(torch.isfinite(logits).all())
    # (b) conservation: sum of the stitched map == sum over tiles of logit x clipped footprint area
    mxy = torch.from_numpy(S.map_coords(tiles, m)).to(dev).long()
    area = (torch.clamp(mxy[:, 0] + 16, max=2500) - mxy[:, 0]) * (torch.clamp(mxy[:, 1] + 16, max=2500) - mxy[:, 1])
    expect = (logits.double() * area[:, None].double()).sum(0)
    got = out1['pred'].sum((1, 2))
    assert float(((got - expect).abs() / expect.abs().clamp_min(1.0)).max()) <= 1e-9
    # (c) batch-composition independence: a random subset recomputed in one small batch is bit-identical
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(len(tiles), 64, replace=False))
    sub = eng.forward_tiles(level0, torch.from_numpy(tiles[pick]).to(dev), tile, tile, logits=True)[1]
    assert torch.equal(sub, logits[torch.from_numpy(pick).to(dev)])
    # (d) oracle spot check
    six = pick[:6]
    u8 = torch.stack([level0[y:y + tile, x:x + tile] for x, y in tiles[six]]).permute(0, 3, 1, 2).contiguous().cpu().numpy()
    with torch.no_grad():
        ref = R.tile_logits(sd, cls, u8)
    assert float((logits[torch.from_numpy(six).to(dev)].cpu() - ref).abs().max()) <= LOGIT_TOL
    # (e) cfg5, full size: the CPU oracle's stitch -> threshold_probs -> heat map (reference utils/eval.py:208-229) fed with
    #     the GPU's logits must reproduce the device's float64 map bit for bit and its classes / u8 heat map with ZERO
    #     differing pixels (a few CPU seconds at 4 x 2500 x 2500)
    from oracle import wsi_oracle as WO
    ref_pred = WO.stitch_tumorbed(tiles, logits.cpu().numpy(), 4, map_hw, m, tile, tile)
    assert np.array_equal(out1['pred'].cpu().numpy(), ref_pred)
    ref_cls, ref_probs = WO.threshold_probs(ref_pred)
    ref_heat = WO.tumorbed_heatmap(ref_probs, np.ones(map_hw, np.uint8), 'cls')
    n_cls = int((out1['classes'].cpu().numpy() != ref_cls).sum())
    n_heat = int((out1['heatmap'].cpu().numpy() != ref_heat).sum())
    ulp = float(np.abs(out1['probs'].cpu().numpy() - ref_probs).max() / np.finfo(np.float64).eps)
    print('cfg5 full-size: %d class pixels, %d heat pixels differ of %d; probs max diff %.1f eps' % (n_cls, n_heat, ref_cls.size, ulp))
    assert n_cls == 0 and n_heat == 0 and ulp <= 4


def test_cfg4_region_bags_at_scale(dev):
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    sd = W.make_resnet18_state_dict(11)
    eng = TrunkEngine(sd, dev, head=(sd['fc0.weight'], sd['fc0.bias']), max_batch=4096)
    R_, P = 1000, 16                                                             # 16 000 crops of 64x64
    g = torch.Generator(device=dev).manual_seed(4)
    u8 = torch.randint(0, 256, (R_ * P * 64, 64, 3), dtype=torch.uint8, device=dev, generator=g)   # crops stacked as a strip
    xy = torch.stack((torch.zeros(R_ * P, dtype=torch.int32), torch.arange(R_ * P, dtype=torch.int32) * 64), 1).to(dev)

    def run(order):
        o = torch.as_tensor(order, device=dev)
        idx = (o[:, None] * P + torch.arange(P, device=dev)[None, :]).reshape(-1)
        feat, logit, _ = eng.forward_tiles(u8, xy[idx], 64, 64, feat=True, logits=True)
        hid = eng.linear(feat.view(len(order), P * 512), sd['fc.0.weight'], sd['fc.0.bias'], relu=True)
        return logit.view(len(order), P, 4), eng.linear(hid, sd['fc.2.weight'], sd['fc.2.bias'])

    base_s, base_e = run(np.arange(R_))
    perm = np.random.default_rng(1).permutation(R_)
    perm_s, perm_e = run(perm)
    pt = torch.from_numpy(perm).to(dev)
    assert torch.equal(perm_s, base_s[pt])                                         # per-bag results independent of order
    assert float((perm_e - base_e[pt]).abs().max()) <= 1e-5
    # oracle spot check on 3 bags
    bags = u8.view(R_, P, 64, 64, 3)[:3].permute(0, 1, 4, 2, 3).contiguous().cpu().numpy()
    xs = R.normalize_u8(bags.reshape(-1, 3, 64, 64)).view(3, P, 3, 64, 64)
    with torch.no_grad():
        rs, re = R.resnet_forward(sd, xs)
    got_s = base_s[:3].transpose(0, 1).reshape(P * 3, 4).cpu()
    assert float((got_s - rs).abs().max()) <= LOGIT_TOL
    assert float((base_e[:3].cpu() - re).abs().max()) <= LOGIT_TOL
