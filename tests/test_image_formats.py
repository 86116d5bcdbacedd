"""The TIFF and DDS readers of the scene loader (csrc/host/image_formats.cpp; load.rs:585-603 decodes them with the image crate,
`decode().flipv().to_rgba8()`): files written by libtiff (through Pillow) and by a byte-level writer here, decoded by the library
and compared with the arrays that went in; DXT blocks against a restatement of the crate's integer rules and, loosely, Pillow."""
import io
import json
import struct
import zlib

import numpy as np
import pytest

from akari_render_amd import capi
from oracle import scene_json

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402


def to_rgba8(a):
    """image crate `to_rgba8` of an (H, W, C) array of u8 / u16 / f32 samples, C in 1..4"""
    if a.dtype == np.uint16:
        a = ((a.astype(np.uint32) + 128) // 257).astype(np.uint8)
    elif a.dtype == np.float32:
        a = np.round(np.clip(np.nan_to_num(a, nan=0.0), 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
    h, w, c = a.shape
    out = np.full((h, w, 4), 255, dtype=np.uint8)
    if c <= 2:
        out[:, :, :3] = a[:, :, :1]
        if c == 2:
            out[:, :, 3] = a[:, :, 1]
    else:
        out[:, :, :c] = a
    return out


def pil_tiff(a, compression, predictor=False):
    mode = {1: "L", 2: "LA", 3: "RGB", 4: "RGBA"}[a.shape[2]]
    im = Image.fromarray(a[:, :, 0] if a.shape[2] == 1 else a, mode=mode if a.dtype == np.uint8 else None)
    b = io.BytesIO()
    kw = {"tiffinfo": {317: 2}} if predictor else {}
    im.save(b, format="TIFF", compression=compression, **kw)
    return b.getvalue()


@pytest.mark.parametrize("channels", [1, 2, 3, 4])
@pytest.mark.parametrize("compression", ["raw", "tiff_lzw", "tiff_adobe_deflate", "packbits"])
def test_tiff_written_by_libtiff(channels, compression):
    rng = np.random.default_rng(channels * 7 + len(compression))
    a = rng.integers(0, 256, size=(37, 53, channels), dtype=np.uint8)
    a[5:20, 10:40] = a[5, 10]  # runs, so that PackBits and LZW have something to compress
    got = capi.host_decode_tiff(pil_tiff(a, compression))
    assert np.array_equal(got, to_rgba8(a))


@pytest.mark.parametrize("compression", ["tiff_lzw", "tiff_adobe_deflate"])
def test_tiff_horizontal_predictor_and_long_lzw_streams(compression):
    """A smooth 300 x 200 RGB image with predictor 2, and noise (every LZW code width up to 12 bits, table resets)."""
    y, x = np.mgrid[0:200, 0:300]
    smooth = np.stack([(x + y) % 256, (2 * x) % 256, (x * y // 64) % 256], axis=2).astype(np.uint8)
    data = pil_tiff(smooth, compression, predictor=True)
    assert struct.unpack_from("<H", data, 0)[0] == 0x4949
    assert np.array_equal(capi.host_decode_tiff(data), to_rgba8(smooth))
    noise = np.random.default_rng(5).integers(0, 256, size=(200, 300, 3), dtype=np.uint8)
    assert np.array_equal(capi.host_decode_tiff(pil_tiff(noise, compression)), to_rgba8(noise))


def test_tiff_16_bit_grey_from_libtiff():
    a = np.random.default_rng(3).integers(0, 65536, size=(19, 23), dtype=np.uint16)
    b = io.BytesIO()
    Image.fromarray(a).save(b, format="TIFF", compression="tiff_lzw")
    assert np.array_equal(capi.host_decode_tiff(b.getvalue()), to_rgba8(a[:, :, None]))


def write_tiff(a, big=False, photometric=None, compression=1, tile=None, rows_per_strip=None, predictor=1, extra_tags=()):
    """Byte-level classic-TIFF writer: (H, W, C) u8 / u16 / f32, strips or tiles, none / deflate / PackBits-literal."""
    e = ">" if big else "<"
    h, w, c = a.shape
    bps = a.dtype.itemsize
    native = a.astype(a.dtype.newbyteorder(e))
    if predictor == 2:
        d = native.astype(a.dtype).astype(np.int64)
        d[:, 1:] = d[:, 1:] - d[:, :-1]
        native = (d % (1 << (8 * bps))).astype(a.dtype).astype(a.dtype.newbyteorder(e))
    chunks = []
    if tile:
        tw, th = tile
        for y0 in range(0, h, th):
            for x0 in range(0, w, tw):
                t = np.zeros((th, tw, c), dtype=native.dtype)
                part = native[y0:y0 + th, x0:x0 + tw]
                if predictor == 2:  # the predictor runs over the tile's own rows
                    src = a[y0:y0 + th, x0:x0 + tw].astype(np.int64)
                    src[:, 1:] = src[:, 1:] - src[:, :-1]
                    part = (src % (1 << (8 * bps))).astype(a.dtype).astype(a.dtype.newbyteorder(e))
                t[:part.shape[0], :part.shape[1]] = part
                chunks.append(t.tobytes())
    else:
        rps = rows_per_strip or h
        for y0 in range(0, h, rps):
            chunks.append(native[y0:y0 + rps].tobytes())

    def pack(raw):
        if compression == 8:
            return zlib.compress(raw)
        if compression == 32773:  # literal runs of at most 128 bytes
            return b"".join(bytes([len(raw[i:i + 128]) - 1]) + raw[i:i + 128] for i in range(0, len(raw), 128))
        return raw

    blobs = [pack(x) for x in chunks]
    if photometric is None:
        photometric = 2 if c >= 3 else 1
    tags = [(256, 4, [w]), (257, 4, [h]), (258, 3, [8 * bps] * c), (259, 3, [compression]), (262, 3, [photometric]), (277, 3, [c]),
            (284, 3, [1]), (317, 3, [predictor]), (339, 3, [3 if a.dtype == np.float32 else 1] * c)]
    if c in (2, 4):
        tags.append((338, 3, [2]))
    tags += list(extra_tags)
    n_tags = len(tags) + (4 if tile else 3)
    ifd_at = 8
    data_at = ifd_at + 2 + 12 * n_tags + 4
    extra = b""

    def field(tag, typ, vals):
        nonlocal extra
        size = {3: 2, 4: 4}[typ] * len(vals)
        body = b"".join(struct.pack(e + ("H" if typ == 3 else "I"), v) for v in vals)
        if size <= 4:
            return struct.pack(e + "HHI", tag, typ, len(vals)) + body.ljust(4, b"\0")
        off = data_at + len(extra)
        extra += body + (b"\0" if len(body) % 2 else b"")
        return struct.pack(e + "HHII", tag, typ, len(vals), off)

    # offsets of the chunks are known only after the out-of-line field data: lay those out first with placeholder offsets
    def build(offsets):
        nonlocal extra
        extra = b""
        t = list(tags)
        if tile:
            t += [(322, 4, [tile[0]]), (323, 4, [tile[1]]), (324, 4, offsets), (325, 4, [len(x) for x in blobs])]
        else:
            t += [(273, 4, offsets), (278, 4, [rows_per_strip or h]), (279, 4, [len(x) for x in blobs])]
        t.sort()
        ifd = struct.pack(e + "H", len(t)) + b"".join(field(*x) for x in t) + struct.pack(e + "I", 0)
        return ifd

    build([0] * len(blobs))
    base = data_at + len(extra)
    offsets, o = [], base
    for x in blobs:
        offsets.append(o)
        o += len(x) + (len(x) & 1)
    ifd = build(offsets)
    out = (b"MM" if big else b"II") + struct.pack(e + "HI", 42, ifd_at) + ifd + extra
    assert len(out) == base
    for x in blobs:
        out += x + (b"\0" if len(x) & 1 else b"")
    return out


@pytest.mark.parametrize("big", [False, True], ids=["II", "MM"])
@pytest.mark.parametrize("case", ["rgb16_strips", "rgba16_tiles_deflate", "greya8_tiles_packbits", "rgb_float", "white_is_zero", "rgb16_predictor", "rgb8_tiles_predictor"])
def test_tiff_byte_level_cases(big, case):
    rng = np.random.default_rng(len(case))
    if case == "rgb16_strips":
        a = rng.integers(0, 65536, size=(21, 17, 3), dtype=np.uint16)
        data, want = write_tiff(a, big, rows_per_strip=4), to_rgba8(a)
    elif case == "rgba16_tiles_deflate":
        a = rng.integers(0, 65536, size=(37, 41, 4), dtype=np.uint16)
        data, want = write_tiff(a, big, compression=8, tile=(16, 16)), to_rgba8(a)
    elif case == "greya8_tiles_packbits":
        a = rng.integers(0, 256, size=(33, 18, 2), dtype=np.uint8)
        data, want = write_tiff(a, big, compression=32773, tile=(16, 32)), to_rgba8(a)
    elif case == "rgb_float":
        a = rng.uniform(-0.2, 1.3, size=(9, 11, 3)).astype(np.float32)
        a[0, 0, 0] = np.nan
        data, want = write_tiff(a, big, rows_per_strip=2), to_rgba8(a)
    elif case == "white_is_zero":
        a = rng.integers(0, 256, size=(8, 9, 1), dtype=np.uint8)
        data, want = write_tiff(a, big, photometric=0), to_rgba8(255 - a)
    elif case == "rgb16_predictor":
        a = rng.integers(0, 65536, size=(12, 13, 3), dtype=np.uint16)
        data, want = write_tiff(a, big, compression=8, predictor=2), to_rgba8(a)
    else:
        a = rng.integers(0, 256, size=(20, 45, 3), dtype=np.uint8)
        data, want = write_tiff(a, big, compression=8, tile=(16, 16), predictor=2), to_rgba8(a)
    got = capi.host_decode_tiff(data)
    assert np.array_equal(got, want)
    if a.dtype == np.uint8 and case != "white_is_zero" and a.shape[2] != 2:  # the writer itself, against libtiff
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGBA")), want)


def write_bigtiff(a, big=False, planar=False, compression=1, tile=None, rows_per_strip=None):
    """Byte-level BigTIFF writer (magic 43: 8-byte offsets, 20-byte directory entries with 64-bit counts, up to 8 value bytes inline; strip /
    tile offsets and byte counts as LONG8), optionally with PlanarConfiguration 2: every sample in strips / tiles of its own, all of sample
    0 first. Written from the BigTIFF design note + TIFF 6.0, independently of the reader."""
    e = ">" if big else "<"
    h, w, c = a.shape
    bps = a.dtype.itemsize
    native = a.astype(a.dtype.newbyteorder(e))
    sources = [native[:, :, k:k + 1] for k in range(c)] if planar else [native]
    chunks = []
    for src in sources:
        if tile:
            tw, th = tile
            for y0 in range(0, h, th):
                for x0 in range(0, w, tw):
                    t = np.zeros((th, tw, src.shape[2]), dtype=native.dtype)
                    part = src[y0:y0 + th, x0:x0 + tw]
                    t[:part.shape[0], :part.shape[1]] = part
                    chunks.append(t.tobytes())
        else:
            rps = rows_per_strip or h
            for y0 in range(0, h, rps):
                chunks.append(np.ascontiguousarray(src[y0:y0 + rps]).tobytes())
    blobs = [zlib.compress(x) if compression == 8 else x for x in chunks]
    tags = [(256, 4, [w]), (257, 4, [h]), (258, 3, [8 * bps] * c), (259, 3, [compression]), (262, 3, [2 if c >= 3 else 1]), (277, 3, [c]),
            (284, 3, [2 if planar else 1]), (339, 3, [3 if a.dtype == np.float32 else 1] * c)]
    if c in (2, 4):
        tags.append((338, 3, [2]))
    n_tags = len(tags) + (4 if tile else 3)
    ifd_at = 16
    data_at = ifd_at + 8 + 20 * n_tags + 8
    extra = b""
    fmt = {3: ("H", 2), 4: ("I", 4), 16: ("Q", 8)}

    def field(tag, typ, vals):
        nonlocal extra
        body = b"".join(struct.pack(e + fmt[typ][0], v) for v in vals)
        if len(body) <= 8:
            return struct.pack(e + "HHQ", tag, typ, len(vals)) + body.ljust(8, b"\0")
        off = data_at + len(extra)
        extra += body
        return struct.pack(e + "HHQQ", tag, typ, len(vals), off)

    def build(offsets):
        nonlocal extra
        extra = b""
        t = list(tags)
        if tile:
            t += [(322, 4, [tile[0]]), (323, 4, [tile[1]]), (324, 16, offsets), (325, 16, [len(x) for x in blobs])]
        else:
            t += [(273, 16, offsets), (278, 4, [rows_per_strip or h]), (279, 16, [len(x) for x in blobs])]
        t.sort()
        return struct.pack(e + "Q", len(t)) + b"".join(field(*x) for x in t) + struct.pack(e + "Q", 0)

    build([0] * len(blobs))
    base = data_at + len(extra)
    offsets, o = [], base
    for x in blobs:
        offsets.append(o)
        o += len(x)
    ifd = build(offsets)
    out = (b"MM" if big else b"II") + struct.pack(e + "HHHQ", 43, 8, 0, ifd_at) + ifd + extra
    assert len(out) == base
    return out + b"".join(blobs)


@pytest.mark.parametrize("big", [False, True], ids=["II", "MM"])
@pytest.mark.parametrize("case", ["rgb8_strips", "rgba16_tiles_deflate", "rgb8_planar_strips", "rgba16_planar_tiles", "greya8_planar", "rgb_float_planar"])
def test_bigtiff_and_planar_samples(big, case):
    """Round 6: BigTIFF and PlanarConfiguration 2, both read by the tiff crate 0.9 behind load.rs:586-600."""
    rng = np.random.default_rng(len(case) + big)
    if case == "rgb8_strips":
        a = rng.integers(0, 256, size=(19, 23, 3), dtype=np.uint8); kw = dict(rows_per_strip=5)
    elif case == "rgba16_tiles_deflate":
        a = rng.integers(0, 65536, size=(37, 41, 4), dtype=np.uint16); kw = dict(compression=8, tile=(16, 16))
    elif case == "rgb8_planar_strips":
        a = rng.integers(0, 256, size=(19, 23, 3), dtype=np.uint8); kw = dict(planar=True, rows_per_strip=4)
    elif case == "rgba16_planar_tiles":
        a = rng.integers(0, 65536, size=(33, 20, 4), dtype=np.uint16); kw = dict(planar=True, tile=(16, 16), compression=8)
    elif case == "greya8_planar":
        a = rng.integers(0, 256, size=(9, 31, 2), dtype=np.uint8); kw = dict(planar=True)
    else:
        a = rng.uniform(-0.2, 1.3, size=(9, 11, 3)).astype(np.float32); kw = dict(planar=True, rows_per_strip=3)
    got = capi.host_decode_tiff(write_bigtiff(a, big, **kw))
    assert np.array_equal(got, to_rgba8(a))
    # a truncated BigTIFF is refused, not read past its end
    data = write_bigtiff(a, big, **kw)
    for cut in (20, len(data) - 3):
        with pytest.raises(capi.AkariError):
            capi.host_decode_tiff(data[:cut])


def test_classic_tiff_with_planar_samples():
    """PlanarConfiguration 2 in a classic TIFF: the chunk list holds all strips of R, then of G, then of B."""
    a = np.random.default_rng(3).integers(0, 256, size=(10, 12, 3), dtype=np.uint8)
    chunky = write_tiff(a, rows_per_strip=5)
    planes = b"".join(np.ascontiguousarray(a[y0:y0 + 5, :, k]).tobytes() for k in range(3) for y0 in (0, 5))
    # rebuild by hand: same directory, PlanarConfiguration 2, six strips of 60 bytes
    t = write_tiff(np.zeros((10, 12, 1), dtype=np.uint8), rows_per_strip=5)  # (a one-sample file gives the layout; patched below)
    del t, chunky
    e = "<"
    tags = sorted([(256, 4, [12]), (257, 4, [10]), (258, 3, [8, 8, 8]), (259, 3, [1]), (262, 3, [2]), (277, 3, [3]), (278, 4, [5]), (284, 3, [2]),
                   (273, 4, None), (279, 4, [60] * 6)])
    n = len(tags)
    data_at = 8 + 2 + 12 * n + 4
    extra = b""
    ifd = struct.pack(e + "H", n)
    strip_base = None
    out_fields = []
    for tag, typ, vals in tags:
        if vals is None:
            vals = [0] * 6
        body = b"".join(struct.pack(e + ("H" if typ == 3 else "I"), v) for v in vals)
        if len(body) <= 4:
            out_fields.append((tag, typ, len(vals), body.ljust(4, b"\0"), None))
        else:
            out_fields.append((tag, typ, len(vals), None, len(extra)))
            extra += body + (b"\0" if len(body) % 2 else b"")
    strip_base = data_at + len(extra)
    ex = bytearray(extra)
    for tag, typ, cnt, inline, at in out_fields:
        if tag == 273:
            ex[at:at + 24] = b"".join(struct.pack(e + "I", strip_base + 60 * k) for k in range(6))
    for tag, typ, cnt, inline, at in out_fields:
        ifd += struct.pack(e + "HHI", tag, typ, cnt) + (inline if inline is not None else struct.pack(e + "I", data_at + at))
    ifd += struct.pack(e + "I", 0)
    data = b"II" + struct.pack(e + "HI", 42, 8) + ifd + bytes(ex) + planes
    assert np.array_equal(capi.host_decode_tiff(data), to_rgba8(a))
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGBA")), to_rgba8(a))  # libtiff reads the same file the same way


def test_tiff_refuses_what_it_does_not_read():
    a = np.zeros((4, 4, 3), dtype=np.uint8)
    good = write_tiff(a)
    for bad, what in ((b"II" + struct.pack("<HI", 43, 8) + good[8:], "directory"),  # a classic file that claims to be BigTIFF: its directory offset is garbage
                      (b"II" + struct.pack("<HHH", 43, 4, 0) + good[8:], "BigTIFF"), (good[:20], "truncated"), (b"XX" + good[2:], "not a TIFF"),
                      (write_tiff(a, compression=7), "compression 7"), (write_tiff(a, photometric=6), "photometric"),
                      (write_tiff(a, photometric=3), "photometric")):  # palette images: the image crate refuses them too
        with pytest.raises(capi.AkariError) as ei:
            capi.host_decode_tiff(bad)
        assert what.split()[0].lower() in str(ei.value).lower(), (what, str(ei.value))
    for cut in (len(good) - 1, len(good) - 20):
        with pytest.raises(capi.AkariError):
            capi.host_decode_tiff(good[:cut])


def test_tiff_header_fields_cannot_overflow_or_allocate(tmp_path):
    """Round-2 advisor findings: (a) TileWidth = TileLength = 2^31 made rows x tile_w x bytes wrap to 0, the "enough samples"
    check passed and the pixel loop read past the chunk (segfault with a 140-byte file); (b) a header alone claiming a huge
    picture cost a 1 GiB allocation before the file was refused as truncated."""
    a = np.arange(4 * 4 * 3, dtype=np.uint8).reshape(4, 4, 3)
    good = write_tiff(a, tile=(16, 16))
    assert np.array_equal(capi.host_decode_tiff(good)[..., :3], a)

    def patch(data, tag, value):  # rewrite the inline value of a LONG / SHORT field of the (little-endian) IFD
        n = struct.unpack_from("<H", data, 8)[0]
        out = bytearray(data)
        for i in range(n):
            at = 10 + 12 * i
            t, typ = struct.unpack_from("<HH", data, at)
            if t == tag:
                struct.pack_into("<HI", out, at + 2, 4, 1)  # type LONG, count 1
                struct.pack_into("<I", out, at + 8, value)
                return bytes(out)
        raise KeyError(tag)

    for tw, th in ((1 << 31, 1 << 31), (1 << 30, 4), (4, 1 << 31), (65537, 16)):
        bad = patch(patch(good, 322, tw), 323, th)
        with pytest.raises(capi.AkariError) as ei:
            capi.host_decode_tiff(bad)
        assert "tile size" in str(ei.value)
    # 16384 x 16384 pixels claimed by a file of a few hundred bytes: refused before anything of that size is allocated
    import resource, time
    huge = patch(patch(write_tiff(a), 256, 16384), 257, 16384)
    t0, r0 = time.time(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    with pytest.raises(capi.AkariError) as ei:
        capi.host_decode_tiff(huge)
    assert "too short" in str(ei.value) or "outside" in str(ei.value) or "few" in str(ei.value)
    assert time.time() - t0 < 0.5 and resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - r0 < 200 * 1024  # KiB
    # the same for a PNG header (IHDR says 16384 x 16384, IDAT holds a 4 x 4 picture)
    png = io.BytesIO()
    Image.fromarray(a).save(png, format="PNG")
    raw = bytearray(png.getvalue())
    struct.pack_into(">II", raw, 16, 16384, 16384)
    struct.pack_into(">I", raw, 29, zlib.crc32(bytes(raw[12:29])))
    r0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    with pytest.raises(capi.AkariError):
        capi.host_decode_png(bytes(raw))
    assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - r0 < 200 * 1024


# ------------------------------------------------------------------------------------------------------------------ DDS
def make_dds(w, h, kind, blocks: bytes, dx10=False):
    hdr = bytearray(128)
    hdr[0:4] = b"DDS "
    struct.pack_into("<IIIII", hdr, 4, 124, 0x1 | 0x2 | 0x4 | 0x1000 | 0x80000, h, w, len(blocks))
    struct.pack_into("<II", hdr, 76, 32, 0x4)
    hdr[84:88] = b"DX10" if dx10 else {1: b"DXT1", 3: b"DXT3", 5: b"DXT5"}[kind]
    struct.pack_into("<I", hdr, 108, 0x1000)
    out = bytes(hdr)
    if dx10:
        out += struct.pack("<IIIII", {1: 71, 3: 74, 5: 77}[kind], 3, 0, 1, 0)
    return out + blocks


@pytest.mark.parametrize("dx10", [False, True])
@pytest.mark.parametrize("kind", [1, 3, 5])
def test_dds_blocks_against_the_restated_rules(kind, dx10):
    rng = np.random.default_rng(kind)
    w, h = 22, 13  # not multiples of 4: the last blocks are cropped
    n = ((w + 3) // 4) * ((h + 3) // 4)
    blocks = bytearray(rng.integers(0, 256, size=n * (8 if kind == 1 else 16), dtype=np.uint8).tobytes())
    # force both orderings of the endpoints (colour and alpha) to appear
    step = 8 if kind == 1 else 16
    for b in range(0, len(blocks), 2 * step):
        o = b + (0 if kind == 1 else 8)
        c0, c1 = struct.unpack_from("<HH", blocks, o)
        struct.pack_into("<HH", blocks, o, min(c0, c1), max(c0, c1))
        if kind == 5:
            blocks[b], blocks[b + 1] = min(blocks[b], blocks[b + 1]), max(blocks[b], blocks[b + 1])
    data = make_dds(w, h, kind, bytes(blocks), dx10)
    got = capi.host_decode_dds(data)
    want = scene_json.decode_dds(data)
    assert got.shape == (h, w, 4) and np.array_equal(got, want)
    if kind == 1:
        assert (got[:, :, 3] == 255).all()  # DXT1 decodes to RGB
    if not dx10:  # an independent decoder: same layout, its own rounding of the interpolated colours (a few LSB)
        pil = np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"))
        assert np.abs(pil[:, :, :3].astype(int) - got[:, :, :3].astype(int)).max() <= 8 or kind == 1
        if kind != 1:
            assert np.abs(pil[:, :, 3].astype(int) - got[:, :, 3].astype(int)).max() <= 1


def test_dds_known_block():
    """One DXT5 block by hand: endpoints red / blue, indices 0 1 2 3 repeating; alpha 255 -> 0 ramp."""
    c0, c1 = 0xF800, 0x001F
    idx = sum((i & 3) << (2 * i) for i in range(16))
    abits = sum((i & 7) << (3 * i) for i in range(16))
    block = bytes([255, 0]) + abits.to_bytes(6, "little") + struct.pack("<HHI", c0, c1, idx)
    got = capi.host_decode_dds(make_dds(4, 4, 5, block))
    assert got[0, 0].tolist() == [255, 0, 0, 255] and got[0, 1].tolist() == [0, 0, 255, 0]
    assert got[0, 2].tolist() == [(2 * 255 + 0 + 1) // 3, 0, (0 + 255 + 1) // 3, (6 * 255) // 7]
    assert got[0, 3].tolist() == [(255 + 1) // 3, 0, (2 * 255 + 1) // 3, (5 * 255) // 7]


def test_dds_refuses_what_it_does_not_read():
    good = make_dds(4, 4, 1, bytes(8))
    for bad in (good[:100], b"XXXX" + good[4:], good[:84] + b"ATI2" + good[88:], good[:80] + struct.pack("<I", 0x40) + good[84:], good[:-1]):
        with pytest.raises(capi.AkariError):
            capi.host_decode_dds(bad)


def test_scene_loader_reads_tiff_and_dds_textures(tmp_path):
    from tests.test_textures import _scene_json_with_textures

    rng = np.random.default_rng(12)
    a = rng.integers(0, 256, size=(20, 24, 3), dtype=np.uint8)
    tiff = pil_tiff(a, "tiff_lzw")
    dds = make_dds(24, 20, 5, rng.integers(0, 256, size=6 * 5 * 16, dtype=np.uint8).tobytes())
    for fmt, blob, want in (("tiff", tiff, to_rgba8(a)), ("dds", dds, scene_json.decode_dds(dds))):
        d = tmp_path / fmt
        d.mkdir()
        path = _scene_json_with_textures(d, blob, rng.random((4, 3, 3)).astype(np.float32))
        scene = json.loads(open(path).read())
        for m in ("m_floor", "m_wall"):
            scene["materials"][m]["shader"]["nodes"]["img"]["image"].update(format=fmt, width=24, height=20)
        p2 = d / "scene2.json"
        p2.write_text(json.dumps(scene))
        got = capi.Scene(None, str(p2)).to_scene_data()
        pyl = scene_json.load_scene(str(p2))
        x = [im for im in got.images if im.texels.dtype == np.uint8][0].texels
        y = [im for im in pyl.images if im.texels.dtype == np.uint8][0].texels
        assert np.array_equal(x, want[::-1]) and np.array_equal(x, y)  # flipped vertically like every encoded image


def test_files_written_by_other_libraries_if_the_machine_has_them():
    """CPython ships one small picture in several formats with its test-suite (test/imghdrdata): files written by libpng, libtiff,
    libjpeg and the OpenEXR library -- the only EXR here that this repository's own writer did not produce. Skipped where absent."""
    import glob
    import sysconfig

    roots = [sysconfig.get_paths()["stdlib"]] + glob.glob("/mnt/sandboxing/model_tools_env/*/python/install/lib/python3*") + glob.glob("/usr/lib/python3*")
    base = next((r + "/test/imghdrdata/" for r in roots if glob.glob(r + "/test/imghdrdata/python.exr")), None)
    if base is None:
        pytest.skip("no CPython test data on this machine")
    ref = np.asarray(Image.open(base + "python.png").convert("RGBA"))
    assert np.array_equal(capi.host_decode_png(open(base + "python.png", "rb").read()), ref)
    assert np.array_equal(capi.host_decode_tiff(open(base + "python.tiff", "rb").read()), ref)
    jpg = capi.host_decode_jpeg(open(base + "python.jpg", "rb").read())
    assert np.abs(jpg.astype(int) - np.asarray(Image.open(base + "python.jpg").convert("RGBA")).astype(int)).max() <= 3
    exr = capi.host_decode_exr(open(base + "python.exr", "rb").read())  # uncompressed half RGBA, channels stored A B G R
    assert exr.shape == ref.shape and np.abs(exr - ref.astype(np.float32) / 255.0).max() < 5e-4
