"""the image files the reference writes after the path — Image::saveImageBMP with toneMap / gammaCorrect
(imageUtil.cu:69-100, 202-257) and Image::saveImageCSV_MONO (:123-142). The product's writers (novum_save_bmp,
novum_save_csv_mono: host code of libptamd.so, no GPU needed) against the oracle's restatement, BYTE for byte."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN


def _frames():
    rng = np.random.default_rng(77)
    out = {}
    for w, h in ((1, 1), (2, 3), (3, 2), (5, 4), (7, 7), (33, 9), (64, 5), (130, 3)):          # row padding 3w % 4 = 3,2,1,1,1,1,0,2
        f = np.zeros((h, w, 4), np.float32)
        f[..., :3] = rng.random((h, w, 3)).astype(np.float32) ** 3 * 4.0                          # HDR-ish radiance
        f[..., 3] = rng.random((h, w)).astype(np.float32)
        out["rand%dx%d" % (w, h)] = f
    nasty = np.zeros((4, 6, 4), np.float32)
    vals = [np.nan, np.inf, -np.inf, -1.0, -0.0, 0.0, 1e-30, 1e-8, 0.0031308, 0.5, 1.0, 1.0 + 1e-7, 2.0, 255.0, 1e30, 3.4e38,
            0.25 / 255.0, 0.5 / 255.0, 1.5 / 255.0, 254.5 / 255.0, 0.49999997 / 255.0, -1e-30, 1e10, 0.9999999]
    nasty[..., 0] = np.array(vals, np.float32).reshape(4, 6)
    nasty[..., 1] = np.array(vals[::-1], np.float32).reshape(4, 6)
    nasty[..., 2] = np.array(vals[5:] + vals[:5], np.float32).reshape(4, 6)
    out["nasty"] = nasty
    g = np.load(os.path.join(GOLDEN, "mixed32_mis.npz"))                                          # a real frame (sum of 8 samples)
    out["mixed32_sum"] = g["colors"].astype(np.float32)
    g = np.load(os.path.join(GOLDEN, "metal32_mis.npz"))
    out["metal32_sum"] = g["colors"].astype(np.float32)
    return out


FRAMES = _frames()


@pytest.mark.parametrize("post", [True, False])
@pytest.mark.parametrize("name", sorted(FRAMES))
def test_bmp_bytes(api, oracle, tmp_path, name, post):
    f = FRAMES[name]
    if name.endswith("_sum"):
        f = oracle.finalise(f, 8).reshape(f.shape)                # what initRender hands to the writer (main.cu:860-886)
    a, b = str(tmp_path / "product.bmp"), str(tmp_path / "oracle.bmp")
    api.save_bmp(a, f, post_process=post)
    oracle.save_bmp(b, f, post_process=post)
    pa, pb = open(a, "rb").read(), open(b, "rb").read()
    h, w = f.shape[:2]
    row = (3 * w + 3) & ~3
    assert len(pb) == 54 + row * h
    assert pa[:54] == pb[:54], "headers differ"
    assert pa == pb, "%s post=%s: %d pixel bytes differ" % (name, post, sum(x != y for x, y in zip(pa, pb)))
    # independent reading of the header fields (imageUtil.cu:234-257)
    assert pb[:2] == b"BM" and struct.unpack_from("<IHHI", pb, 2) == (54 + row * h, 0, 0, 54)
    assert struct.unpack_from("<IiiHHIIiiII", pb, 14) == (40, w, h, 1, 24, 0, row * h, 0, 0, 0, 0)
    # BGR order, bottom-up rows, zero padding — spelled out on one pixel, without post-processing
    if not post and name.startswith("rand"):
        y, x = h - 1, w - 1
        px = pb[54 + y * row + 3 * x: 54 + y * row + 3 * x + 3]
        want = [int(np.float32(min(max(float(f[y, x, c]), 0.0), 1.0)) * np.float32(255.0) + np.float32(0.5)) for c in (2, 1, 0)]
        assert list(px) == want
        assert all(v == 0 for v in pb[54 + y * row + 3 * w: 54 + (y + 1) * row])


def test_tonemap_constants(oracle):
    """ACES fit 2.51 / 0.03 / 2.43 / 0.59 / 0.14, clamp to [0,1], then ^(1/2.2) (imageUtil.cu:202-222), against numpy
    in float64 — a wrong constant or a swapped channel would show at the 1e-6 level."""
    x = np.zeros((6, 4), np.float32)
    x[:, 0] = [0.0, 0.05, 0.18, 0.5, 1.0, 4.0]; x[:, 1] = x[::-1, 0]; x[:, 2] = 0.3
    got = oracle.tonemap_gamma(x)
    c = x[:, :3].astype(np.float64)
    want = np.clip((c * (2.51 * c + 0.03)) / (c * (2.43 * c + 0.59) + 0.14), 0, 1) ** (1 / 2.2)
    assert np.allclose(got[:, :3], want, rtol=2e-6, atol=1e-7) and not got[:, 3].any()


@pytest.mark.parametrize("channel", [0, 1, 2])
def test_csv_mono_bytes(api, oracle, tmp_path, channel):
    for name in ("rand5x4", "rand33x9", "nasty", "mixed32_sum"):
        f = FRAMES[name]
        a, b = str(tmp_path / "product.csv"), str(tmp_path / "oracle.csv")
        api.save_csv_mono(a, f, channel)
        oracle.save_csv_mono(b, f, channel)
        ta, tb = open(a, "rb").read(), open(b, "rb").read()
        assert ta == tb, name
        rows = tb.decode().strip("\n").split("\n")
        assert len(rows) == f.shape[0] and all(len(r.split(",")) == f.shape[1] for r in rows)
    assert rows[0].split(",")[0] == "%.3e" % float(FRAMES["mixed32_sum"][0, 0, channel])
