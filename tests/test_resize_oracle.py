"""Pins oracle/resize_oracle.py against the installed Pillow (the reference's own dependency for `image.resize`,
utils/dataset.py:180-181) bit for bit."""
import numpy as np
import pytest

from oracle import resize_oracle as RO

PIL = pytest.importorskip('PIL.Image')


@pytest.mark.parametrize('seed,src,dst', [(0, (512, 512), (256, 256)), (1, (128, 128), (64, 64)), (2, (96, 160), (64, 64)),
                                          (3, (64, 64), (128, 96)), (4, (300, 200), (100, 67)), (5, (33, 47), (33, 20)),
                                          (6, (768, 768), (256, 256))])
def test_matches_pillow(seed, src, dst):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, src + (3,), dtype=np.uint8)
    if seed % 2:
        img[: src[0] // 3] = 255                                            # saturated background: exercises the clip
        img[src[0] // 3: src[0] // 2, ::2] = 0
    want = np.asarray(PIL.fromarray(img).resize((dst[1], dst[0])))
    assert np.array_equal(RO.resize_bicubic_u8(img, dst), want)


def test_default_filter_is_bicubic():
    img = np.random.default_rng(9).integers(0, 256, (40, 40, 3), dtype=np.uint8)
    a = np.asarray(PIL.fromarray(img).resize((20, 20)))
    b = np.asarray(PIL.fromarray(img).resize((20, 20), PIL.BICUBIC))
    assert np.array_equal(a, b)
