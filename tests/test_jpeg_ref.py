"""CPU: the numpy restatement of libjpeg-turbo's lossy JPEG stages (oracle/jpeg_ref.py, the checker of the
device codec) is bit-exact to Pillow, including the committed codec goldens."""
import numpy as np
import pytest

from ddpm_image_restoration_amd import synth
from oracle import ddpm_oracle as O
from oracle import jpeg_ref as J


@pytest.mark.parametrize("quality", [1, 10, 20, 30, 31, 50, 90, 100])
def test_jpeg_ref_equals_pillow(quality):
    rng = np.random.default_rng(quality)
    imgs = [O.to_u8(synth.synth_images(2, 64, seed=quality)), rng.integers(0, 256, (2, 3, 32, 96), dtype=np.uint8),
            rng.integers(0, 2, (1, 3, 64, 64)).astype(np.uint8) * 255, np.zeros((1, 3, 32, 32), np.uint8)]
    for u8 in imgs:
        assert np.array_equal(J.roundtrip_u8(u8, quality), O.codec_u8(u8, quality, "jpeg"))


def test_jpeg_ref_matches_reference_goldens(golden):
    g = golden("codec")
    for q in (10, 20, 50):
        for tag in ("32", "64"):
            got = O.from_u8(J.roundtrip_u8(O.to_u8(g["in_" + tag]), q))
            assert np.array_equal(got, g[f"jpeg_q{q}_{tag}"])
