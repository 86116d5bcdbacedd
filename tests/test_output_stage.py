"""Output stage: util::write_image restated -- .exr = linear RGB f32, .png = 8-bit sRGB."""
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from akari_render_amd import capi


def read_exr_rgb(path):
    b = open(path, "rb").read()
    assert struct.unpack("<I", b[:4])[0] == 20000630 and struct.unpack("<I", b[4:8])[0] == 2
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); ty = b[pos:e].decode(); pos = e + 1
        (n,) = struct.unpack("<I", b[pos:pos + 4]); pos += 4
        attrs[name] = (ty, b[pos:pos + n]); pos += n
    pos += 1
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    assert attrs["compression"][1] == b"\0" and attrs["channels"][1].startswith(b"B\0")
    offs = struct.unpack("<%dQ" % h, b[pos:pos + 8 * h])
    img = np.zeros((h, w, 3), dtype=np.float32)
    for y in range(h):
        yy, nb = struct.unpack("<iI", b[offs[y]:offs[y] + 8])
        row = np.frombuffer(b[offs[y] + 8:offs[y] + 8 + nb], dtype="<f4").reshape(3, w)
        img[yy, :, 2], img[yy, :, 1], img[yy, :, 0] = row[0], row[1], row[2]  # planes B, G, R
    return img


def read_png_rgb8(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(b):
        (n,) = struct.unpack(">I", b[pos:pos + 4]); ty = b[pos + 4:pos + 8]; data = b[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(ty + data) & 0xFFFFFFFF == crc
        if ty == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", data[:10]); assert (depth, ctype) == (8, 2)
        if ty == b"IDAT":
            idat += data
        pos += 12 + n
    raw = zlib.decompress(idat)
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert np.all(rows[:, 0] == 0)
    return rows[:, 1:].reshape(h, w, 3)


def test_exr_and_png_writers(hip_lib, tmp_path):
    rng = np.random.default_rng(0)
    img = (rng.random((37, 53, 3)) * 3).astype(np.float32)
    img[0, 0] = [0.0, 1e-4, 2.5]; img[1, 1] = [-1.0, 0.0031308, 1.0]
    exr = tmp_path / "sub" / "dir" / "a.exr"   # parent directories are created (util/mod.rs:83-84)
    capi.image_write(str(exr), img)
    assert np.array_equal(read_exr_rgb(str(exr)), img)
    png = tmp_path / "a.png"
    capi.image_write(str(png), img)
    got = read_png_rgb8(str(png))
    lin = img.astype(np.float32)
    srgb = np.where(lin <= np.float32(0.0031308), lin * np.float32(12.92), np.power(np.maximum(lin, 0), np.float32(1 / 2.4)) * np.float32(1.055) - np.float32(0.055))
    exp = np.clip(srgb * 255.0, 0, 255).astype(np.uint8)  # (x * 255).clamp(0, 255) as u8, util/mod.rs:88
    assert np.max(np.abs(got.astype(int) - exp.astype(int))) <= 1
    with pytest.raises(capi.AkariError):
        capi.image_write(str(tmp_path / "a.jpg"), img)


@pytest.mark.gpu
def test_cli_end_to_end(ctx, root, tmp_path):
    """akari-cli -s scene.json -m method.json --save-intermediate --save-stats NAME (akari_cli.rs:8-95)."""
    from akari_render_amd import build
    cli = build.build_cli()
    method = {"method": {"type": "pt", "spp": 8, "spp_per_pass": 4, "max_depth": 5}, "sampler": {"type": "pmj02bn", "seed": 3},
              "film": {"out": str(tmp_path / "out" / "img.exr"), "filter": {"type": "gaussian", "radius": 1.5}}}
    mpath = tmp_path / "m.json"
    mpath.write_text(json.dumps(method))
    cmd = [cli, "-s", os.path.join(root, "scenes/cbox/scene.json"), "-m", str(mpath), "--resolution", "64x48", "--save-intermediate",
           "--save-stats", "run1", "--independent-sampler", "-v"]
    res = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert "Rendering finished in" in res.stdout
    img = read_exr_rgb(str(tmp_path / "out" / "img.exr"))
    assert img.shape == (48, 64, 3) and np.all(np.isfinite(img)) and img.mean() > 0.01
    stats = json.load(open(tmp_path / "run1.json"))
    assert [e["spp"] for e in stats["intermediate"]] == [4, 8] and stats["intermediate"][1]["time"] >= stats["intermediate"][0]["time"]
    assert np.array_equal(read_exr_rgb(str(tmp_path / "run1-8.exr")), img)
    # same render through the library API
    from oracle import scene_json
    from akari_render_amd import abi
    cfg = abi.PtConfig.default(); cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.sampler_seed = 8, 4, 5, 3
    scene = capi.Scene(ctx, os.path.join(root, "scenes/cbox/scene.json"), 64, 48)
    film = capi.Film(ctx, 64, 48)
    capi.pt_render(ctx, scene, cfg, film)
    assert np.array_equal(film.resolve(), img)
    # without --independent-sampler the method file's pmj02bn sampler is used: same scene, different sample streams
    res = subprocess.run(cmd[:-2], cwd=tmp_path, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    img_pmj = read_exr_rgb(str(tmp_path / "out" / "img.exr"))
    assert np.all(np.isfinite(img_pmj)) and not np.array_equal(img_pmj, img)
    assert abs(img_pmj.mean() - img.mean()) < 0.05 * img.mean()
    cfg.sampler_type = abi.SAMPLER_PMJ02BN
    film.clear()
    capi.pt_render(ctx, scene, cfg, film)
    assert np.array_equal(film.resolve(), img_pmj)
