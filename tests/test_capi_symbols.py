"""CPU-only: the C-ABI library builds/loads and exports every symbol include/wsi_hip.h declares;
host-side entry points (no GPU needed) behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from wsi_segmentation_pipeline_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(native.LIB_PATH):
        native.build()
    return native.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, 'include', 'wsi_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(wsi_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 20
    assert declared == set(native.SIGNATURES), declared ^ set(native.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert int(re.search(r'#define WSI_HIP_ABI_VERSION (\d+)', hdr).group(1)) == native.ABI_VERSION == lib.wsi_hip_abi_version()


def test_pf_layout_helpers(lib):
    # pixel (n,y,x) -> (W+2) + n*(H+1)*(W+1) + y*(W+1) + x
    assert lib.wsi_pf_pixel_index(0, 0, 0, 8, 8) == 10
    assert lib.wsi_pf_pixel_index(2, 3, 4, 8, 8) == 10 + 2 * 81 + 3 * 9 + 4
    nbytes = lib.wsi_pf_bytes(3, 8, 8, 512, 2)
    assert nbytes % (512 * 2 * 2) == 0
    assert nbytes // (512 * 4) >= 2 * 10 + 3 * 81
    assert lib.wsi_pf_bytes(0, 8, 8, 512, 2) == 0
    assert lib.wsi_trunk_workspace_bytes(4, 256, 256, 2) > 4 * 4 * 1024 * 1024
    assert lib.wsi_trunk_workspace_bytes(4, 250, 256, 2) == 0           # not a multiple of 32


def test_normalize_lut_matches_transform(lib):
    from oracle.resnet_oracle import normalize_u8, DATASET_MEAN, DATASET_STD
    from wsi_segmentation_pipeline_amd.engine import normalize_lut
    lut = normalize_lut(DATASET_MEAN, DATASET_STD)
    codes = np.arange(256, dtype=np.uint8).reshape(1, 1, 16, 16).repeat(3, 1)
    ref = normalize_u8(codes).numpy().reshape(3, 256)
    assert np.array_equal(lut, ref)                                      # bit-exact: u8 -> fp32 affine


def _bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def test_prepack_conv_roundtrip(lib):
    rng = np.random.default_rng(0)
    cout, cin, k = 64, 128, 3
    w = rng.standard_normal((cout, cin, k, k)).astype(np.float32)
    g = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    m = rng.standard_normal(cout).astype(np.float32)
    v = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for planes in (1, 2):
        nbytes = lib.wsi_prepack_conv_bytes(cout, cin, k, planes)
        nblocks = (cout // 32) * (cin * planes // 64) * k * k * 4096
        assert nbytes == nblocks + (cout * 4 if planes == 2 else 0)      # mode 2: + the inverse per-channel weight scales
        pk = np.zeros(nbytes // 2, np.uint16)
        bias = np.zeros(cout, np.float32)
        assert lib.wsi_prepack_conv(p(w), p(g), p(b), p(m), p(v), 1e-5, cout, cin, k, planes, p(pk), p(bias)) == 0
        scale = g.astype(np.float64) / np.sqrt(v.astype(np.float64) + 1e-5)
        wf = (w.astype(np.float64) * scale[:, None, None, None]).astype(np.float32)
        assert np.allclose(bias, b - m * scale, atol=1e-6)
        # unpack [ntile][line][tap][f][lane][8] back to OIHW and compare hi+lo against the folded weights
        nl = cin * planes // 64
        body = pk[:nblocks // 2]
        if planes == 2:                                                  # fp16 pair of (weight x 2^e), max |.| per channel in [2^13, 2^14)
            frag = body.view(np.float16).astype(np.float32).reshape(cout // 32, nl, k * k, 4, 64, 8)
            inv = pk[nblocks // 2:].view(np.float32)
            assert inv.shape == (cout,) and np.all(np.log2(inv) == np.round(np.log2(inv)))
        else:
            frag = _bf16_to_f32(body).reshape(cout // 32, nl, k * k, 4, 64, 8)
            inv = np.ones(cout, np.float32)
        rec = np.zeros_like(wf)
        for nt in range(cout // 32):
            for l in range(nl):
                for f in range(4):
                    cbase = 32 * l + 16 * (f & 1) if planes == 2 else 64 * l + 16 * f
                    for lane in range(64):
                        co = nt * 32 + (lane & 31)
                        ci = cbase + 8 * (lane >> 5)
                        rec[co, ci:ci + 8] += frag[nt, l, :, f, lane, :].T.reshape(8, k, k)
        if planes == 2:
            amax = np.abs(rec).reshape(cout, -1).max(1)
            assert np.all((amax >= 2.0 ** 13) & (amax < 2.0 ** 14))
        rec = rec * inv[:, None, None, None]
        tol = 2 ** -21 if planes == 2 else 2 ** -8                       # fp16 pair: 22 significand bits where lo is normal
        assert np.abs(rec - wf).max() <= tol * np.abs(wf).max()


def test_bad_arguments_return_einval(lib):
    assert lib.wsi_prepack_conv_bytes(48, 64, 3, 2) == 0
    assert lib.wsi_prepack_conv(None, None, None, None, None, 1e-5, 64, 64, 3, 2, None, None) == -22
    assert lib.wsi_conv3x3_bn_act(None, None, None, None, None, 1, 8, 8, 64, 64, 1, 1, 2, None) == -22
    assert lib.wsi_trunk_forward(None, None, None, 0, 0, 0, None, None, 1, 64, 64, None, 0, None, None, None, None) == -22


def test_prepack_conv_mode3_fp6_planes(lib):
    """Mode-3 weight pack (fp16 hi + MX-fp6 cross-term planes, include/wsi_hip.h): decode the 4 KB blocks on the host and check
    them against the weights - frag 0/1 = fp16(w) in activation line order, frag 2/3 of lanes h=0 = fp6(hi / 2^(s-127)) and of
    lanes h=1 = fp6((w - hi) / 2^(s-127)) in the field order the conv epilogue produces, every element within half an fp6 step
    of its source and the block maximum inside fp6's top binade."""
    rng = np.random.default_rng(3)
    cout, cin, k = 32, 64, 3
    w = (rng.standard_normal((cout, cin, k, k)) * rng.uniform(0.05, 3.0, (1, cin, 1, 1))).astype(np.float32)
    nbytes = lib.wsi_prepack_conv_bytes(cout, cin, k, 3)
    assert nbytes == (cout // 32) * (cin // 32) * 9 * 4096
    out = np.zeros(nbytes, np.uint8)
    bias = np.zeros(cout, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.wsi_prepack_conv(p(w), None, None, None, None, 1e-5, cout, cin, k, 3, p(out), p(bias)) == 0
    assert not bias.any()
    line_chan = lambda pos: 8 * ((pos & 15) >> 2) + 4 * (pos >> 4) + (pos & 3)          # common.h mx_line_chan
    field_chan = lambda f: line_chan(16 * (f & 1) + (f >> 1))                           # common.h mx6_field_chan

    def fp6(code):
        e, m = (code >> 3) & 3, code & 7
        v = (8 + m) * (0.125, 0.25, 0.5)[e - 1] if e else m * 0.125
        return -v if code & 32 else v
    blocks = out.reshape(cin // 32, 9, 4, 64, 16)
    worst = 0.0
    for l in range(cin // 32):
        for t in range(9):
            for r in (0, 5, 31):
                wrow = w[r, 32 * l:32 * l + 32, t // 3, t % 3]
                hi = wrow.astype(np.float16).astype(np.float32)
                lo = wrow - hi
                for h in (0, 1):
                    lane = r + 32 * h
                    f16 = np.concatenate([blocks[l, t, 0, lane].view(np.float16), blocks[l, t, 1, lane].view(np.float16)])
                    pos = [8 * h + j for j in range(8)] + [16 + 8 * h + j for j in range(8)]
                    assert np.array_equal(f16.astype(np.float32), hi[[line_chan(q) for q in pos]])
                    words = np.concatenate([blocks[l, t, 2, lane].view(np.uint32), blocks[l, t, 3, lane].view(np.uint32)[:2]])
                    sbyte = int(blocks[l, t, 3, lane].view(np.uint32)[2])
                    bits = int.from_bytes(words.tobytes(), 'little')
                    src = hi if h == 0 else lo
                    amax = float(np.abs(src).max())
                    assert 1 <= sbyte <= 254
                    scale = 2.0 ** (sbyte - 127)
                    assert 3.75 < amax / scale <= 7.75                                    # the top binade, never more than saturation
                    for f in range(32):
                        q = fp6((bits >> (6 * f)) & 63) * scale
                        x = float(src[field_chan(f)])
                        step = 0.5 if abs(x) / scale >= 4 else 0.25 if abs(x) / scale >= 2 else 0.125
                        worst = max(worst, abs(q - x) / (step * scale))
    assert worst <= 0.5 + 1e-6


def test_unet_tail_prepack_is_the_polyphase_sum(lib):
    """wsi_unet_tail_prepack (host side of csrc/tail.hip): decoding the fp16 hi + lo fragments and undoing the per-channel power-of-two
    scale gives, per output parity (py, px) and low-resolution offset, the SUM of the 3x3 taps whose nearest-upsampled source falls on
    that offset (BN scale folded) - and running that 2 x 3-tap filter bank on a low-resolution image equals the 3x3 conv on its x2
    nearest upsampling.  conv2's fragments: rows 0-15 = output row k (dy = input row), rows 16-31 = output row k + 1 (dy = input row - 1)."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    cin, cmid, classes = 32, 16, 3
    w1 = rng.normal(0, 0.2, (cmid, cin, 3, 3)).astype(np.float32)
    w2 = rng.normal(0, 0.3, (cmid, cmid, 3, 3)).astype(np.float32)
    bn = [[rng.uniform(0.5, 1.5, cmid).astype(np.float32), rng.normal(0, 0.1, cmid).astype(np.float32),
           rng.normal(0, 0.1, cmid).astype(np.float32), rng.uniform(0.5, 1.5, cmid).astype(np.float32)] for _ in range(2)]
    hw = rng.normal(0, 1, (classes, cmid)).astype(np.float32)
    hb = rng.normal(0, 1, classes).astype(np.float32)
    blob = np.zeros(lib.wsi_unet_tail_prepack_bytes(), np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.wsi_unet_tail_prepack(p(w1), *[p(a) for a in bn[0]], p(w2), *[p(a) for a in bn[1]], 1e-5, p(hw), p(hb), cin, cmid, classes, p(blob)) == 0
    assert lib.wsi_unet_tail_prepack(p(w1), *[p(a) for a in bn[0]], p(w2), *[p(a) for a in bn[1]], 1e-5, p(hw), p(hb), 64, cmid, classes, p(blob)) == -22
    fl = blob[(48 + 24) * 1024:].view(np.float32)
    sc = [bn[j][0].astype(np.float64) / np.sqrt(bn[j][3].astype(np.float64) + 1e-5) for j in range(2)]
    sh = [bn[j][1].astype(np.float64) - bn[j][2].astype(np.float64) * sc[j] for j in range(2)]
    assert np.allclose(fl[16:32], sh[0], rtol=1e-6, atol=1e-7) and np.allclose(fl[48:64], sh[1], rtol=1e-6, atol=1e-7)
    assert np.array_equal(fl[64:64 + 64].reshape(4, 16)[:classes], hw) and np.array_equal(fl[128:128 + classes], hb) and not fl[64 + 16 * classes:128].any()
    frags = blob[:(48 + 24) * 1024].view(np.float16).astype(np.float64).reshape(-1, 64, 8)       # [fragment][lane][j]

    def conv1_weight(py, t, row, ci):                            # hi + lo of A row `row`, input channel ci, tap t of parity class py
        base = (py * 6 + t) * 4
        f, lane, j = ci // 16, row + 32 * ((ci % 16) // 8), ci % 8
        return frags[base + f, lane, j] + frags[base + 2 + f, lane, j]
    wsets = {0: ([0], [1, 2]), 1: ([0, 1], [2])}                 # parity -> taps d feeding (first, second) low-resolution offset
    got = np.zeros((2, 2, 2, 3, cmid, cin))                     # [py][px][a][ox index][c][ci]
    for py in range(2):
        for t in range(6):
            for row in range(32):
                px, c = row // 16, row % 16
                for ci in range(cin):
                    got[py, px, t // 3, t % 3, c, ci] = conv1_weight(py, t, row, ci) * fl[c]
    wf = (w1.astype(np.float64) * sc[0][:, None, None, None]).astype(np.float32).astype(np.float64)
    for py in range(2):
        for px in range(2):
            for a_ in range(2):
                for oxi in range(3):
                    b_ = oxi - px
                    exp = np.zeros((cmid, cin))
                    if 0 <= b_ <= 1:
                        for dy in wsets[py][a_]:
                            for dx in wsets[px][b_]:
                                exp += wf[:, :, dy, dx]
                    assert np.abs(got[py, px, a_, oxi] - exp).max() <= 2e-6 * np.abs(wf).max(), (py, px, a_, oxi)
    # the filter bank on a low-resolution image == the 3x3 conv on its nearest x2 upsampling (the identity the kernel rests on)
    x = torch.from_numpy(rng.normal(0, 1, (1, cin, 6, 7)))
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode='nearest'), torch.from_numpy(wf), padding=1)
    xp = F.pad(x, (1, 1, 1, 1))
    out = torch.zeros_like(ref)
    for py in range(2):
        for px in range(2):
            acc = 0
            for a_ in range(2):
                for oxi in range(3):
                    oy, ox = py - 1 + a_, oxi - 1
                    acc = acc + torch.einsum('oc,nchw->nohw', torch.from_numpy(got[py, px, a_, oxi]), xp[:, :, 1 + oy:1 + oy + 6, 1 + ox:1 + ox + 7])
            out[:, :, py::2, px::2] = acc
    assert float((out - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    # conv2: 12 taps = 4 input rows x 3 columns, two output rows per tile
    w2f = (w2.astype(np.float64) * sc[1][:, None, None, None]).astype(np.float32).astype(np.float64)
    for t in range(12):
        for row in range(32):
            rs, c = row // 16, row % 16
            dy, dx = t // 3 - rs, t % 3
            for ci in range(cmid):
                lane, j = row + 32 * (ci // 8), ci % 8
                v = (frags[48 + t * 2, lane, j] + frags[48 + t * 2 + 1, lane, j]) * fl[32 + c]
                exp = w2f[c, ci, dy, dx] if 0 <= dy <= 2 else 0.0
                assert abs(v - exp) <= 2e-6 * np.abs(w2f).max(), (t, row, ci)
