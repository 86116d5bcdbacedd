// image_formats.cpp -- the two remaining encoded texture formats of the reference's scene loader
// (crates/akari_render/src/load.rs:585-592: Png, Jpeg, Tiff, OpenExr, Dds; the first three live in image_io.cpp).
//
// The reference decodes them with the `image` crate (Cargo.lock: image 0.24.7, tiff 0.9.0), `decode().flipv().to_rgba8()`
// (load.rs:596-603). Those crates are not part of the reference mount, so what is restated here is the published formats --
// TIFF 6.0 (Aldus/Adobe, 1992) with the Adobe deflate extension, and S3TC / DXT1-3-5 inside a DDS container -- with the
// integer conventions of the crate where the format leaves a choice:
//   * 16-bit samples -> 8 bit: (v + 128) / 257                         (image `FromPrimitive<u16> for u8`, as the PNG reader)
//   * 32-bit float samples -> 8 bit: round(clamp(v, 0, 1) * 255)        (image `FromPrimitive<f32> for u8`)
//   * grey -> (l, l, l, 255), grey + alpha -> (l, l, l, a), WhiteIsZero inverted
//   * DXT colour endpoints: 5/6-bit channel c -> c * 255 / 31 (63), interpolants (2 a + b + 1) / 3, DXT1's 3-colour mode
//     (a + b + 1) / 2 and black; DXT5 alpha (k a0 + (7 - k) a1) / 7 resp. / 5 without rounding; DXT3 alpha nibble * 17
//     (image/src/codecs/dxt.rs). Other decoders round these differently (e.g. expand 565 by bit replication): a DDS texture is
//     therefore pinned to the crate's arithmetic only as far as this restatement of it is right -- there is no copy of the crate
//     here to check against ("parity unpinned" items).
// Rows come out in file order (top first); the caller flips (load.rs:596).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "scene_build.h"

namespace akr {

namespace {

// ---------------------------------------------------------------------------------------------------------------- TIFF
struct TiffReader {
    const uint8_t* d;
    size_t n;
    bool big;
    uint16_t u16(size_t o) const {
        if (o + 2 > n) throw std::runtime_error("tiff: truncated file");
        return big ? (uint16_t)((d[o] << 8) | d[o + 1]) : (uint16_t)(d[o] | (d[o + 1] << 8));
    }
    uint32_t u32(size_t o) const {
        if (o + 4 > n) throw std::runtime_error("tiff: truncated file");
        return big ? ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]
                   : ((uint32_t)d[o + 3] << 24) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 1] << 8) | d[o];
    }
};
struct TiffField {
    uint16_t type = 0;
    uint64_t count = 0;
    size_t value_at = 0;  // where the values start (inside the entry or at the offset it holds)
};
size_t tiff_type_size(uint16_t t) {
    return t == 1 || t == 2 || t == 6 || t == 7 ? 1 : t == 3 || t == 8 ? 2 : t == 4 || t == 9 || t == 11 || t == 13 ? 4 : t == 5 || t == 10 || t == 12 || (t >= 16 && t <= 18) ? 8 : 0;
}
uint64_t tiff_u64(const TiffReader& r, size_t o) {
    const uint64_t a = r.u32(o), b = r.u32(o + 4);
    return r.big ? (a << 32) | b : (b << 32) | a;
}
// BYTE / SHORT / LONG and BigTIFF's LONG8 / IFD8 (offsets and byte counts of files past 4 GiB; values that do not fit 64-bit size_t do not occur)
uint64_t tiff_value(const TiffReader& r, const TiffField& f, uint64_t i) {
    if (i >= f.count) throw std::runtime_error("tiff: field index out of range");
    if (f.type == 1 || f.type == 7) { if (f.value_at + i >= r.n) throw std::runtime_error("tiff: truncated file"); return r.d[f.value_at + i]; }
    if (f.type == 3) return r.u16(f.value_at + 2 * (size_t)i);
    if (f.type == 4 || f.type == 13) return r.u32(f.value_at + 4 * (size_t)i);
    if (f.type == 16 || f.type == 18) return tiff_u64(r, f.value_at + 8 * (size_t)i);
    throw std::runtime_error("tiff: unexpected field type " + std::to_string(f.type));
}

// TIFF 6.0 section 13: MSB-first codes of 9..12 bits, 256 = clear, 257 = end of information, the width grows one code early
void tiff_lzw(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
    std::vector<uint16_t> prefix(4096);
    std::vector<uint8_t> suffix(4096), first(4096);
    std::vector<uint8_t> stack;
    for (int i = 0; i < 256; i++) { prefix[i] = 0xffff; suffix[i] = first[i] = (uint8_t)i; }
    uint32_t bits = 9, next = 258, acc = 0;
    int have = 0;
    size_t pos = 0;
    int old = -1;
    out.clear();
    out.reserve(std::min<size_t>(expect, 1u << 24));  // `expect` comes from header fields: grow as the data really decodes
    for (;;) {
        while (have < (int)bits) {
            if (pos >= n) return;  // a stream without an end code: what was decoded stands (libtiff writes EOI; others may not)
            acc = (acc << 8) | src[pos++];
            have += 8;
        }
        const uint32_t code = (acc >> (have - (int)bits)) & ((1u << bits) - 1u);
        have -= (int)bits;
        if (code == 257) return;
        if (code == 256) { bits = 9; next = 258; old = -1; continue; }
        if (old < 0) {
            if (code > 255) throw std::runtime_error("tiff: corrupt LZW stream");
            out.push_back((uint8_t)code);
            old = (int)code;
            continue;
        }
        uint32_t cur = code;
        stack.clear();
        if (code >= next) {
            if (code != next) throw std::runtime_error("tiff: corrupt LZW stream");
            stack.push_back(first[old]);
            cur = (uint32_t)old;
        }
        while (cur > 255) { stack.push_back(suffix[cur]); cur = prefix[cur]; }
        stack.push_back((uint8_t)cur);
        for (size_t k = stack.size(); k-- > 0;) out.push_back(stack[k]);
        if (next < 4096) {
            prefix[next] = (uint16_t)old;
            suffix[next] = (uint8_t)cur;
            first[next] = first[old];
            next++;
        }
        if (next + 1 >= (1u << bits) && bits < 12) bits++;
        old = (int)code;
        if (out.size() >= expect) return;
    }
}
void tiff_packbits(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
    out.clear();
    out.reserve(std::min<size_t>(expect, n * 128 + 16));  // PackBits expands at most 128 times
    size_t pos = 0;
    while (pos < n && out.size() < expect) {
        const int8_t c = (int8_t)src[pos++];
        if (c >= 0) {
            const size_t k = (size_t)c + 1;
            if (pos + k > n) throw std::runtime_error("tiff: truncated PackBits run");
            out.insert(out.end(), src + pos, src + pos + k);
            pos += k;
        } else if (c != -128) {
            if (pos >= n) throw std::runtime_error("tiff: truncated PackBits run");
            out.insert(out.end(), (size_t)(1 - (int)c), src[pos++]);
        }
    }
}
uint8_t f32_to_u8(float v) {
    if (!(v > 0.0f)) return 0;  // also NaN
    if (v > 1.0f) v = 1.0f;
    return (uint8_t)std::lround(v * 255.0f);
}
}  // namespace

// TIFF -> RGBA8 in file order. Classic (not Big-) TIFF, first image of the file, either byte order; strips or tiles; chunky
// samples of 8 / 16 bits (unsigned) or 32-bit float; grey (BlackIsZero / WhiteIsZero), grey + alpha, RGB, RGBA; compression
// none, LZW, deflate (8 and the old 32946), PackBits; horizontal predictor. Everything else is refused by name.
void decode_tiff(const uint8_t* data, size_t n, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba) {
    if (n < 8) throw std::runtime_error("tiff: truncated file");
    TiffReader r{data, n, false};
    if (data[0] == 'I' && data[1] == 'I') r.big = false;
    else if (data[0] == 'M' && data[1] == 'M') r.big = true;
    else throw std::runtime_error("tiff: not a TIFF file");
    const uint16_t magic = r.u16(2);
    // BigTIFF (magic 43; the `tiff` crate 0.9 behind load.rs:586-600 reads it): 8-byte offsets, 20-byte directory entries with 64-bit counts, values of
    // up to 8 bytes inline. Everything after the directory is the same file format.
    const bool bigtiff = magic == 43;
    if (!bigtiff && magic != 42) throw std::runtime_error("tiff: not a TIFF file");
    if (bigtiff && (n < 16 || r.u16(4) != 8 || r.u16(6) != 0)) throw std::runtime_error("tiff: bad BigTIFF header");
    const uint64_t ifd64 = bigtiff ? tiff_u64(r, 8) : r.u32(4);
    if (ifd64 > n) throw std::runtime_error("tiff: directory outside the file");
    const size_t ifd = (size_t)ifd64;
    const uint64_t n_entries = bigtiff ? tiff_u64(r, ifd) : r.u16(ifd);
    if (n_entries > 65535) throw std::runtime_error("tiff: directory too large");
    const size_t entry_size = bigtiff ? 20 : 12, entries_at = ifd + (bigtiff ? 8 : 2), inline_bytes = bigtiff ? 8 : 4;
    TiffField f_bits, f_strip_off, f_strip_cnt, f_tile_off, f_tile_cnt, f_extra, f_format;
    uint32_t compression = 1, photometric = 0xffffffffu, spp = 1, rows_per_strip = 0xffffffffu, planar = 1, predictor = 1, tile_w = 0, tile_h = 0;
    width = height = 0;
    for (uint64_t e = 0; e < n_entries; e++) {
        const size_t at = entries_at + entry_size * (size_t)e;
        const uint16_t tag = r.u16(at);
        TiffField f;
        f.type = r.u16(at + 2);
        f.count = bigtiff ? tiff_u64(r, at + 4) : r.u32(at + 4);
        if (f.count > n) throw std::runtime_error("tiff: field count larger than the file");
        const size_t bytes = tiff_type_size(f.type) * (size_t)f.count;
        const size_t val = at + (bigtiff ? 12 : 8);
        const uint64_t where = bytes <= inline_bytes ? val : (bigtiff ? tiff_u64(r, val) : r.u32(val));
        if (where > n || (bytes > inline_bytes && bytes > n - where)) throw std::runtime_error("tiff: field data outside the file");
        f.value_at = (size_t)where;
        auto first = [&] { return (uint32_t)tiff_value(r, f, 0); };
        switch (tag) {
            case 256: width = first(); break;
            case 257: height = first(); break;
            case 258: f_bits = f; break;
            case 259: compression = first(); break;
            case 262: photometric = first(); break;
            case 273: f_strip_off = f; break;
            case 277: spp = first(); break;
            case 278: rows_per_strip = first(); break;
            case 279: f_strip_cnt = f; break;
            case 284: planar = first(); break;
            case 317: predictor = first(); break;
            case 322: tile_w = first(); break;
            case 323: tile_h = first(); break;
            case 324: f_tile_off = f; break;
            case 325: f_tile_cnt = f; break;
            case 338: f_extra = f; break;
            case 339: f_format = f; break;
            default: break;
        }
    }
    if (width == 0 || height == 0) throw std::runtime_error("tiff: missing image size");
    if ((uint64_t)width * height > (1ull << 28)) throw std::runtime_error("tiff: image too large");
    if (spp < 1 || spp > 4) throw std::runtime_error("unsupported: tiff with " + std::to_string(spp) + " samples per pixel");
    uint32_t bits = 1;
    if (f_bits.count) {
        bits = (uint32_t)tiff_value(r, f_bits, 0);
        for (uint32_t i = 1; i < f_bits.count && i < spp; i++)
            if (tiff_value(r, f_bits, i) != bits) throw std::runtime_error("unsupported: tiff with different bit depths per sample");
    }
    uint32_t format = 1;
    if (f_format.count) format = (uint32_t)tiff_value(r, f_format, 0);
    const bool is_float = format == 3;
    if (!((bits == 8 || bits == 16) && format == 1) && !(bits == 32 && is_float))
        throw std::runtime_error("unsupported: tiff sample format (" + std::to_string(bits) + " bits, format " + std::to_string(format) + ")");
    // PlanarConfiguration 2 (TIFF 6.0 section "PlanarConfiguration"): every sample has its own strips / tiles -- all chunks of sample 0, then all of
    // sample 1, ... -- each holding one sample per pixel. Read as `spp` one-sample images written into their component.
    if (planar != 1 && planar != 2) throw std::runtime_error("tiff: bad planar configuration");
    const bool separate = planar == 2 && spp > 1;
    if (photometric == 0xffffffffu) throw std::runtime_error("tiff: missing photometric interpretation");
    if (photometric > 2) throw std::runtime_error("unsupported: tiff photometric interpretation " + std::to_string(photometric));
    if (photometric == 2 ? (spp != 3 && spp != 4) : (spp != 1 && spp != 2)) throw std::runtime_error("unsupported: tiff sample count for its photometric interpretation");
    if (predictor != 1 && predictor != 2) throw std::runtime_error("unsupported: tiff predictor " + std::to_string(predictor));
    if (predictor == 2 && is_float) throw std::runtime_error("unsupported: tiff horizontal predictor on float samples");
    if (compression != 1 && compression != 5 && compression != 8 && compression != 32946 && compression != 32773)
        throw std::runtime_error("unsupported: tiff compression " + std::to_string(compression));
    const bool tiled = f_tile_off.count != 0;
    if (tiled && (tile_w == 0 || tile_h == 0)) throw std::runtime_error("tiff: tile size missing");
    // tile sizes are untrusted header fields: rows * tile_w * bytes_per_pixel below must not wrap around (a 2^31 x 2^31 tile made
    // `expect` 0, the "enough samples" check pass and the pixel loop read past the chunk: round-2 advisor finding)
    if (tiled && (tile_w > 65536u || tile_h > 65536u)) throw std::runtime_error("tiff: tile size out of range");
    if (!tiled && f_strip_off.count == 0) throw std::runtime_error("tiff: no strips and no tiles");
    const TiffField& f_off = tiled ? f_tile_off : f_strip_off;
    const TiffField& f_cnt = tiled ? f_tile_cnt : f_strip_cnt;
    const uint32_t cw = tiled ? tile_w : width;                                             // chunk size in pixels
    const uint32_t ch = tiled ? tile_h : (rows_per_strip > height ? height : rows_per_strip);
    if (ch == 0) throw std::runtime_error("tiff: zero rows per strip");
    const uint32_t across = (width + cw - 1) / cw, down = (height + ch - 1) / ch;
    const uint32_t n_planes = separate ? spp : 1u, chunk_spp = separate ? 1u : spp;
    if ((uint64_t)across * down * n_planes > f_off.count) throw std::runtime_error("tiff: too few strip / tile offsets");
    const size_t bps = bits / 8, px_bytes = bps * chunk_spp;
    // samples of the whole image, native 16 / 32-bit values widened: decoded once, converted at the end
    std::vector<uint8_t> chunk;
    // no allocation on the word of the header alone: the file must be able to hold the picture (stored: byte for byte; LZW /
    // deflate / PackBits cannot expand more than ~1400 times), otherwise a 100-byte file costs a gigabyte before it is refused
    {
        const uint64_t need = (uint64_t)width * height * px_bytes * n_planes;
        if (compression == 1 ? need > n : need / 1400u > n) throw std::runtime_error("tiff: file too short for its image size");
    }
    rgba.assign(4ull * width * height, 255);
    for (uint32_t plane = 0; plane < n_planes; plane++)
    for (uint32_t cy = 0; cy < down; cy++) {
        for (uint32_t cx = 0; cx < across; cx++) {
            const uint64_t idx = ((uint64_t)plane * down + cy) * across + cx;
            const uint64_t off64 = tiff_value(r, f_off, idx);
            if (off64 > n) throw std::runtime_error("tiff: strip / tile data outside the file");
            const size_t off = (size_t)off64;
            const uint32_t rows = tiled ? ch : (cy + 1 == down ? height - cy * ch : ch);  // the last strip may be short; tiles are padded
            const size_t expect = (size_t)rows * cw * px_bytes;
            const uint64_t len64 = f_cnt.count > idx ? tiff_value(r, f_cnt, idx) : (compression == 1 ? expect : n - off);
            if (len64 > n - off) throw std::runtime_error("tiff: strip / tile data outside the file");
            const size_t len = (size_t)len64;
            if (compression == 1) chunk.assign(data + off, data + off + len);
            else if (compression == 5) tiff_lzw(data + off, len, chunk, expect);
            else if (compression == 32773) tiff_packbits(data + off, len, chunk, expect);
            else chunk = inflate_zlib_stream(data + off, len);
            if (chunk.size() < expect) throw std::runtime_error("tiff: strip / tile holds fewer samples than its size says");
            for (uint32_t y = 0; y < rows; y++) {
                const uint32_t iy = cy * ch + y;
                if (iy >= height) break;
                uint8_t* row = chunk.data() + (size_t)y * cw * px_bytes;
                // samples to native order, then the horizontal predictor over the chunk's row (TIFF 6.0 section 14)
                uint32_t prev[4] = {0, 0, 0, 0};
                for (uint32_t x = 0; x < cw; x++) {
                    const uint32_t ix = cx * cw + x;
                    uint32_t v[4];
                    for (uint32_t s = 0; s < chunk_spp; s++) {
                        const uint8_t* p = row + ((size_t)x * chunk_spp + s) * bps;
                        uint32_t q = bps == 1 ? p[0] : bps == 2 ? (r.big ? (uint32_t)((p[0] << 8) | p[1]) : (uint32_t)(p[0] | (p[1] << 8)))
                                                                : (r.big ? ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]
                                                                         : ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0]);
                        if (predictor == 2) { q = (q + prev[s]) & (bps == 1 ? 0xffu : 0xffffu); prev[s] = q; }
                        v[s] = q;
                    }
                    if (ix >= width) continue;
                    uint8_t b[4];
                    for (uint32_t s = 0; s < chunk_spp; s++) {
                        if (is_float) { float fv; std::memcpy(&fv, &v[s], 4); b[s] = f32_to_u8(fv); }
                        else b[s] = bps == 1 ? (uint8_t)v[s] : (uint8_t)((v[s] + 128u) / 257u);
                    }
                    uint8_t* o = rgba.data() + 4 * ((size_t)iy * width + ix);
                    if (separate) {  // this chunk holds sample `plane` of its pixels
                        if (photometric == 2) {
                            o[plane] = b[0];  // R, G, B, A in that order
                        } else {
                            const uint8_t l = photometric == 0 ? (uint8_t)(255 - b[0]) : b[0];
                            if (plane == 0) o[0] = o[1] = o[2] = l; else o[3] = b[0];
                        }
                    } else if (photometric == 2) {
                        o[0] = b[0]; o[1] = b[1]; o[2] = b[2];
                        o[3] = spp == 4 ? b[3] : 255;
                    } else {
                        const uint8_t l = photometric == 0 ? (uint8_t)(255 - b[0]) : b[0];
                        o[0] = o[1] = o[2] = l;
                        o[3] = spp == 2 ? b[1] : 255;
                    }
                }
            }
        }
    }
    (void)f_extra;  // the meaning of a fourth / second sample (associated or not) does not change the bytes `to_rgba8` returns
}

// ----------------------------------------------------------------------------------------------------------------- DDS
namespace {
void dxt_colors(const uint8_t* s, uint8_t out[16][4], bool is_dxt1) {
    const uint32_t c0 = s[0] | (s[1] << 8), c1 = s[2] | (s[3] << 8);
    const uint32_t table = s[4] | (s[5] << 8) | (s[6] << 16) | ((uint32_t)s[7] << 24);
    uint32_t col[4][3];
    auto dec = [](uint32_t v, uint32_t* c) {
        c[0] = ((v >> 11) & 0x1f) * 0xff / 0x1f;
        c[1] = ((v >> 5) & 0x3f) * 0xff / 0x3f;
        c[2] = (v & 0x1f) * 0xff / 0x1f;
    };
    dec(c0, col[0]);
    dec(c1, col[1]);
    if (c0 > c1 || !is_dxt1) {
        for (int i = 0; i < 3; i++) {
            col[2][i] = (col[0][i] * 2 + col[1][i] + 1) / 3;
            col[3][i] = (col[0][i] + col[1][i] * 2 + 1) / 3;
        }
    } else {
        for (int i = 0; i < 3; i++) {
            col[2][i] = (col[0][i] + col[1][i] + 1) / 2;
            col[3][i] = 0;
        }
    }
    for (int i = 0; i < 16; i++) {
        const uint32_t* c = col[(table >> (2 * i)) & 3];
        out[i][0] = (uint8_t)c[0]; out[i][1] = (uint8_t)c[1]; out[i][2] = (uint8_t)c[2];
    }
}
}  // namespace

// DDS holding DXT1 / DXT3 / DXT5 blocks (FourCC, or a DX10 header naming BC1 / BC2 / BC3) -> RGBA8 in file order, top mip level
// only. DXT1 decodes to RGB (alpha 255 everywhere, also in the 3-colour mode), as the crate's decoder does.
void decode_dds(const uint8_t* data, size_t n, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba) {
    auto u32 = [&](size_t o) {
        if (o + 4 > n) throw std::runtime_error("dds: truncated file");
        return (uint32_t)data[o] | ((uint32_t)data[o + 1] << 8) | ((uint32_t)data[o + 2] << 16) | ((uint32_t)data[o + 3] << 24);
    };
    if (n < 128 || std::memcmp(data, "DDS ", 4) != 0) throw std::runtime_error("dds: not a DDS file");
    if (u32(4) != 124) throw std::runtime_error("dds: bad header size");
    height = u32(12);
    width = u32(16);
    if (u32(76) != 32) throw std::runtime_error("dds: bad pixel format size");
    const uint32_t pf_flags = u32(80), fourcc = u32(84);
    if (!(pf_flags & 0x4u)) throw std::runtime_error("unsupported: dds without a FourCC (uncompressed layouts are not read)");
    size_t at = 128;
    int kind = 0;  // 1, 3, 5
    auto cc = [](const char* s) { return (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24); };
    if (fourcc == cc("DXT1")) kind = 1;
    else if (fourcc == cc("DXT3")) kind = 3;
    else if (fourcc == cc("DXT5")) kind = 5;
    else if (fourcc == cc("DX10")) {
        const uint32_t dxgi = u32(128);
        at = 148;
        if (dxgi == 70 || dxgi == 71 || dxgi == 72) kind = 1;       // BC1 typeless / unorm / unorm_srgb
        else if (dxgi == 73 || dxgi == 74 || dxgi == 75) kind = 3;  // BC2
        else if (dxgi == 76 || dxgi == 77 || dxgi == 78) kind = 5;  // BC3
        else throw std::runtime_error("unsupported: dds DXGI format " + std::to_string(dxgi));
    } else {
        throw std::runtime_error("unsupported: dds FourCC");
    }
    if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 28)) throw std::runtime_error("dds: bad image size");
    const uint32_t bw = (width + 3) / 4, bh = (height + 3) / 4;
    const size_t block = kind == 1 ? 8 : 16;
    if (at > n || (size_t)bw * bh * block > n - at) throw std::runtime_error("dds: truncated block data");
    rgba.assign(4ull * width * height, 255);
    for (uint32_t by = 0; by < bh; by++) {
        for (uint32_t bx = 0; bx < bw; bx++) {
            const uint8_t* s = data + at + ((size_t)by * bw + bx) * block;
            uint8_t px[16][4];
            for (int i = 0; i < 16; i++) px[i][3] = 255;
            if (kind == 1) {
                dxt_colors(s, px, true);
            } else if (kind == 3) {
                for (int i = 0; i < 16; i++) px[i][3] = (uint8_t)(((s[i / 2] >> (4 * (i & 1))) & 0xf) * 0x11);
                dxt_colors(s + 8, px, false);
            } else {
                const uint32_t a0 = s[0], a1 = s[1];
                uint32_t tab[8] = {a0, a1, 0, 0, 0, 0, 0, 0xff};
                if (a0 > a1) for (uint32_t i = 2; i < 8; i++) tab[i] = ((8 - i) * a0 + (i - 1) * a1) / 7;
                else for (uint32_t i = 2; i < 6; i++) tab[i] = ((6 - i) * a0 + (i - 1) * a1) / 5;
                uint64_t bits = 0;
                for (int i = 0; i < 6; i++) bits |= (uint64_t)s[2 + i] << (8 * i);
                for (int i = 0; i < 16; i++) px[i][3] = (uint8_t)tab[(bits >> (3 * i)) & 7];
                dxt_colors(s + 8, px, false);
            }
            for (int i = 0; i < 16; i++) {
                const uint32_t x = bx * 4 + (uint32_t)(i & 3), y = by * 4 + (uint32_t)(i >> 2);
                if (x < width && y < height) std::memcpy(rgba.data() + 4 * ((size_t)y * width + x), px[i], 4);
            }
        }
    }
}

}  // namespace akr
