// image_io.cpp -- output stage of the render driver: the reference's util::write_image
// (crates/akari_render/src/util/mod.rs:57-127): ".exr" -> linear RGB f32 OpenEXR, anything else -> 8-bit sRGB.
// Both writers are self-contained: OpenEXR scanline file with no compression (channels B, G, R as 32-bit float, what
// `exr::prelude::write_rgb_file` produces minus its compression), PNG with stored (uncompressed) deflate blocks.
#include <sys/stat.h>

#include <cmath>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace akr {

namespace {
void put_u32(std::vector<uint8_t>& b, uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
void put_u64(std::vector<uint8_t>& b, uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
void put_be32(std::vector<uint8_t>& b, uint32_t v) { for (int i = 3; i >= 0; i--) b.push_back((uint8_t)(v >> (8 * i))); }
void put_str(std::vector<uint8_t>& b, const char* s) { while (*s) b.push_back((uint8_t)*s++); b.push_back(0); }

void mkdir_parents(const std::string& path) {  // std::fs::create_dir_all(parent_dir), util/mod.rs:83-84
    for (size_t i = 1; i < path.size(); i++)
        if (path[i] == '/') {
            std::string d = path.substr(0, i);
            ::mkdir(d.c_str(), 0777);
        }
}
void write_all(const std::string& path, const std::vector<uint8_t>& bytes) {
    mkdir_parents(path);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot open '" + path + "' for writing");
    size_t n = std::fwrite(bytes.data(), 1, bytes.size(), f);
    std::fclose(f);
    if (n != bytes.size()) throw std::runtime_error("cannot open '" + path + "': short write");
}

uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}
void png_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    size_t start = out.size();
    for (int i = 0; i < 4; i++) out.push_back((uint8_t)type[i]);
    out.insert(out.end(), data.begin(), data.end());
    uint32_t crc = crc32_update(0xFFFFFFFFu, out.data() + start, out.size() - start) ^ 0xFFFFFFFFu;
    put_be32(out, crc);
}
}  // namespace

// f32_linear_to_srgb1, color.rs:564-570
float linear_to_srgb1(float l) { return l <= 0.0031308f ? l * 12.92f : powf(l, 1.0f / 2.4f) * 1.055f - 0.055f; }

// write_image_hdr, util/mod.rs:94-127
void write_exr_rgb(const std::string& path, const float* rgb, uint32_t w, uint32_t h) {
    std::vector<uint8_t> b;
    put_u32(b, 20000630u);  // magic
    put_u32(b, 2u);         // version 2, single-part scanline
    auto attr = [&](const char* name, const char* type, const std::vector<uint8_t>& v) {
        put_str(b, name);
        put_str(b, type);
        put_u32(b, (uint32_t)v.size());
        b.insert(b.end(), v.begin(), v.end());
    };
    std::vector<uint8_t> v;
    for (const char* ch : {"B", "G", "R"}) {  // channel list, alphabetical
        put_str(v, ch);
        put_u32(v, 2u);  // FLOAT
        v.push_back(0); v.push_back(0); v.push_back(0); v.push_back(0);  // pLinear + reserved
        put_u32(v, 1u); put_u32(v, 1u);  // sampling
    }
    v.push_back(0);
    attr("channels", "chlist", v);
    v.clear(); v.push_back(0);  // NO_COMPRESSION
    attr("compression", "compression", v);
    v.clear(); put_u32(v, 0); put_u32(v, 0); put_u32(v, w - 1); put_u32(v, h - 1);
    attr("dataWindow", "box2i", v);
    attr("displayWindow", "box2i", v);
    v.clear(); v.push_back(0);  // INCREASING_Y
    attr("lineOrder", "lineOrder", v);
    float one = 1.0f, zero = 0.0f;
    v.clear(); v.resize(4); std::memcpy(v.data(), &one, 4);
    attr("pixelAspectRatio", "float", v);
    v.clear(); v.resize(8); std::memcpy(v.data(), &zero, 4); std::memcpy(v.data() + 4, &zero, 4);
    attr("screenWindowCenter", "v2f", v);
    v.clear(); v.resize(4); std::memcpy(v.data(), &one, 4);
    attr("screenWindowWidth", "float", v);
    b.push_back(0);  // end of header
    const uint64_t row_bytes = 12ull * w, table_pos = b.size();
    uint64_t data_pos = table_pos + 8ull * h;
    for (uint32_t y = 0; y < h; y++) put_u64(b, data_pos + (uint64_t)y * (8 + row_bytes));
    b.reserve(b.size() + (size_t)h * (8 + row_bytes));
    std::vector<float> plane(w);
    for (uint32_t y = 0; y < h; y++) {
        put_u32(b, y);
        put_u32(b, (uint32_t)row_bytes);
        for (int c : {2, 1, 0}) {  // B, G, R planes
            for (uint32_t x = 0; x < w; x++) plane[x] = rgb[3 * ((size_t)y * w + x) + c];
            const uint8_t* p = (const uint8_t*)plane.data();
            b.insert(b.end(), p, p + 4ull * w);
        }
    }
    write_all(path, b);
}

// write_image_ldr, util/mod.rs:64-93: sRGB OETF, (x * 255).clamp(0, 255) as u8
void write_png_srgb8(const std::string& path, const float* rgb, uint32_t w, uint32_t h) {
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (1 + 3ull * w));
    for (uint32_t y = 0; y < h; y++) {
        raw.push_back(0);  // filter type None
        for (uint32_t x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) {
                float s = linear_to_srgb1(rgb[3 * ((size_t)y * w + x) + c]) * 255.0f;
                s = s != s ? 0.0f : (s < 0.0f ? 0.0f : (s > 255.0f ? 255.0f : s));
                raw.push_back((uint8_t)s);
            }
    }
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);  // zlib header, no compression
    uint32_t a = 1, bsum = 0;
    for (uint8_t c : raw) { a = (a + c) % 65521u; bsum = (bsum + a) % 65521u; }
    size_t pos = 0;
    while (pos < raw.size() || raw.empty()) {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00 (stored)
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
        if (raw.empty()) break;
    }
    put_be32(z, (bsum << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w); put_be32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", {});
    write_all(path, out);
}

// util::write_image, util/mod.rs:57-63
void write_image(const std::string& path, const float* rgb, uint32_t w, uint32_t h) {
    auto ends_with = [&](const char* suf) { size_t n = std::strlen(suf); return path.size() >= n && path.compare(path.size() - n, n, suf) == 0; };
    if (ends_with(".exr")) write_exr_rgb(path, rgb, w, h);
    else if (ends_with(".png")) write_png_srgb8(path, rgb, w, h);
    else throw std::runtime_error("unsupported: image format of '" + path + "' (use .exr or .png)");
}

// ------------------------------------------------------------------------------------------------ PNG reader
// What `image::io::Reader::decode()` + `to_rgba8()` give the reference for a PNG texture (load.rs:583-604): expanded
// palette / low bit depths / tRNS, 16-bit samples rounded to 8, grey replicated, alpha 255 when absent; Adam7
// interlaced files included. Rows come out in file order (top first); the caller flips (load.rs:596).
namespace {
struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t buf = 0;
    int cnt = 0;
    uint32_t bits(int k) {
        while (cnt < k) {
            if (pos >= n) throw std::runtime_error("png: truncated deflate stream");
            buf |= (uint32_t)p[pos++] << cnt;
            cnt += 8;
        }
        uint32_t v = buf & ((1u << k) - 1u);
        buf >>= k;
        cnt -= k;
        return v;
    }
    void align() { buf = 0; cnt = 0; }
};
struct Huffman {
    uint16_t count[16] = {0}, symbol[320] = {0};
    void build(const uint8_t* len, int n) {
        for (int i = 0; i < 16; i++) count[i] = 0;
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; i++) offs[i + 1] = (uint16_t)(offs[i] + count[i]);
        for (int i = 0; i < n; i++)
            if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= (int)br.bits(1);
            int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        throw std::runtime_error("png: bad Huffman code");
    }
};
std::vector<uint8_t> inflate_zlib(const uint8_t* data, size_t n) {
    if (n < 6) throw std::runtime_error("png: zlib stream too short");
    if ((data[0] & 0x0f) != 8 || ((data[0] << 8) | data[1]) % 31 != 0 || (data[1] & 0x20)) throw std::runtime_error("png: bad zlib header");
    BitReader br{data + 2, n - 2};
    std::vector<uint8_t> out;
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        uint32_t last = br.bits(1), type = br.bits(2);
        if (type == 0) {
            br.align();
            if (br.pos + 4 > br.n) throw std::runtime_error("png: truncated stored block");
            uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xffffu) != nlen || br.pos + len > br.n) throw std::runtime_error("png: bad stored block");
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman hl, hd;
            uint8_t lens[320];
            if (type == 1) {
                for (int i = 0; i < 144; i++) lens[i] = 8;
                for (int i = 144; i < 256; i++) lens[i] = 9;
                for (int i = 256; i < 280; i++) lens[i] = 7;
                for (int i = 280; i < 288; i++) lens[i] = 8;
                hl.build(lens, 288);
                for (int i = 0; i < 30; i++) lens[i] = 5;
                hd.build(lens, 30);
            } else {
                int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.bits(3);
                Huffman hc;
                hc.build(cl, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = hc.decode(br);
                    if (sym < 16) {
                        lens[idx++] = (uint8_t)sym;
                    } else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) throw std::runtime_error("png: bad code lengths");
                            val = lens[idx - 1];
                            rep = 3 + (int)br.bits(2);
                        } else if (sym == 17) {
                            rep = 3 + (int)br.bits(3);
                        } else {
                            rep = 11 + (int)br.bits(7);
                        }
                        if (idx + rep > nlen + ndist) throw std::runtime_error("png: bad code lengths");
                        while (rep--) lens[idx++] = (uint8_t)val;
                    }
                }
                hl.build(lens, nlen);
                hd.build(lens + nlen, ndist);
            }
            for (;;) {
                int sym = hl.decode(br);
                if (sym < 256) {
                    out.push_back((uint8_t)sym);
                } else if (sym == 256) {
                    break;
                } else {
                    sym -= 257;
                    if (sym >= 29) throw std::runtime_error("png: bad length symbol");
                    uint32_t len = lbase[sym] + br.bits(lext[sym]);
                    int ds = hd.decode(br);
                    if (ds >= 30) throw std::runtime_error("png: bad distance symbol");
                    uint32_t dist = dbase[ds] + br.bits(dext[ds]);
                    if (dist > out.size()) throw std::runtime_error("png: distance too far back");
                    size_t from = out.size() - dist;
                    for (uint32_t i = 0; i < len; i++) out.push_back(out[from + i]);
                }
            }
        } else {
            throw std::runtime_error("png: bad block type");
        }
        if (last) break;
    }
    // adler32 trailer
    br.align();
    if (br.pos + 4 <= br.n) {
        uint32_t want = ((uint32_t)br.p[br.pos] << 24) | ((uint32_t)br.p[br.pos + 1] << 16) | ((uint32_t)br.p[br.pos + 2] << 8) | br.p[br.pos + 3];
        uint32_t a = 1, b = 0;
        for (uint8_t c : out) {
            a = (a + c) % 65521u;
            b = (b + a) % 65521u;
        }
        if (((b << 16) | a) != want) throw std::runtime_error("png: adler32 mismatch");
    }
    return out;
}
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace

std::vector<uint8_t> inflate_zlib_stream(const uint8_t* data, size_t n) { return inflate_zlib(data, n); }  // for image_formats.cpp

void decode_png(const uint8_t* data, size_t n, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (n < 8 || std::memcmp(data, sig, 8) != 0) throw std::runtime_error("png: bad signature");
    size_t pos = 8;
    uint32_t depth = 0, ctype = 0;
    bool have_hdr = false, interlaced = false;
    std::vector<uint8_t> idat, plte, trns;
    while (pos + 12 <= n) {
        uint32_t len = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        if (pos + 12 + (size_t)len > n) throw std::runtime_error("png: truncated chunk");
        const uint8_t* body = data + pos + 8;
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) throw std::runtime_error("png: bad IHDR");
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9];
            if (body[10] != 0 || body[11] != 0) throw std::runtime_error("png: unknown compression / filter method");
            if (body[12] > 1) throw std::runtime_error("png: unknown interlace method");
            interlaced = body[12] == 1;
            have_hdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_hdr || w == 0 || h == 0) throw std::runtime_error("png: missing IHDR");
    if ((uint64_t)w * h > (1ull << 28)) throw std::runtime_error("png: image too large");
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: throw std::runtime_error("png: bad colour type");
    }
    bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                    (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                    ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!depth_ok) throw std::runtime_error("png: bad bit depth");
    if (ctype == 3 && plte.size() < 3) throw std::runtime_error("png: palette image without PLTE");
    const size_t bpp_bits = (size_t)channels * depth, bpp = (bpp_bits + 7) / 8;
    std::vector<uint8_t> raw = inflate_zlib(idat.data(), idat.size());
    // the picture is allocated only once the data can hold it (every pixel's bits are in `raw`, interlaced or not)
    if ((uint64_t)raw.size() * 8 < (uint64_t)w * h * bpp_bits) throw std::runtime_error("png: image data too short");
    rgba.assign(4ull * w * h, 255);
    auto sample = [&](const uint8_t* row, size_t idx) -> uint32_t {  // idx-th sample of the row, raw value
        if (depth == 8) return row[idx];
        if (depth == 16) return ((uint32_t)row[2 * idx] << 8) | row[2 * idx + 1];
        size_t bit = idx * depth;
        return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
    };
    auto to8 = [&](uint32_t v) -> uint8_t {
        if (depth == 8) return (uint8_t)v;
        if (depth == 16) return (uint8_t)((v + 128u) / 257u);
        return (uint8_t)(v * 255u / ((1u << depth) - 1u));  // 1/2/4-bit grey expanded to 8 bits
    };
    // One reduced image: pw x ph pixels whose scanlines start at `src`; pixel (px, py) of it is pixel (x0 + px dx, y0 + py dy) of
    // the picture. The whole picture is one such image with steps 1; an Adam7 file holds seven (PNG specification, section 8.2),
    // each filtered on its own (the "previous row" of a pass's first row is all zeros).
    size_t consumed = 0;
    std::vector<uint8_t> img;
    auto reduced_image = [&](uint32_t pw, uint32_t ph, uint32_t x0, uint32_t y0, uint32_t dx, uint32_t dy) {
        if (pw == 0 || ph == 0) return;
        const size_t stride = ((size_t)pw * bpp_bits + 7) / 8;
        if (raw.size() - consumed < (stride + 1) * (size_t)ph) throw std::runtime_error("png: image data too short");
        img.assign(stride * (size_t)ph, 0);
        for (uint32_t y = 0; y < ph; y++) {  // unfilter
            const uint8_t* src = raw.data() + consumed + (stride + 1) * (size_t)y;
            uint8_t ft = src[0];
            src++;
            uint8_t* cur = img.data() + stride * (size_t)y;
            const uint8_t* up = y ? cur - stride : nullptr;
            for (size_t x = 0; x < stride; x++) {
                int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0, v = src[x];
                switch (ft) {
                    case 0: break;
                    case 1: v += a; break;
                    case 2: v += b; break;
                    case 3: v += (a + b) >> 1; break;
                    case 4: v += paeth(a, b, c); break;
                    default: throw std::runtime_error("png: bad filter type");
                }
                cur[x] = (uint8_t)v;
            }
        }
        consumed += (stride + 1) * (size_t)ph;
        for (uint32_t y = 0; y < ph; y++) {
            const uint8_t* row = img.data() + stride * (size_t)y;
            for (uint32_t x = 0; x < pw; x++) {
                uint8_t* o = rgba.data() + 4ull * ((size_t)w * (y0 + (size_t)y * dy) + (x0 + (size_t)x * dx));
                if (ctype == 3) {
                    uint32_t i = sample(row, x);
                    if (3 * (size_t)i + 2 >= plte.size()) throw std::runtime_error("png: palette index out of range");
                    o[0] = plte[3 * i]; o[1] = plte[3 * i + 1]; o[2] = plte[3 * i + 2];
                    o[3] = i < trns.size() ? trns[i] : 255;
                } else if (ctype == 0) {
                    uint32_t v = sample(row, x);
                    o[0] = o[1] = o[2] = to8(v);
                    if (trns.size() >= 2 && v == (((uint32_t)trns[0] << 8) | trns[1])) o[3] = 0;
                } else if (ctype == 4) {
                    o[0] = o[1] = o[2] = to8(sample(row, 2 * (size_t)x));
                    o[3] = to8(sample(row, 2 * (size_t)x + 1));
                } else if (ctype == 2) {
                    uint32_t r = sample(row, 3 * (size_t)x), g = sample(row, 3 * (size_t)x + 1), b = sample(row, 3 * (size_t)x + 2);
                    o[0] = to8(r); o[1] = to8(g); o[2] = to8(b);
                    if (trns.size() >= 6 && r == (((uint32_t)trns[0] << 8) | trns[1]) && g == (((uint32_t)trns[2] << 8) | trns[3]) &&
                        b == (((uint32_t)trns[4] << 8) | trns[5]))
                        o[3] = 0;
                } else {
                    for (int c = 0; c < 4; c++) o[c] = to8(sample(row, 4 * (size_t)x + c));
                }
            }
        }
    };
    if (!interlaced) {
        reduced_image(w, h, 0, 0, 1, 1);
    } else {
        static const uint32_t X0[7] = {0, 4, 0, 2, 0, 1, 0}, Y0[7] = {0, 0, 4, 0, 2, 0, 1}, DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int k = 0; k < 7; k++) {
            const uint32_t pw = w > X0[k] ? (w - X0[k] + DX[k] - 1) / DX[k] : 0, ph = h > Y0[k] ? (h - Y0[k] + DY[k] - 1) / DY[k] : 0;
            reduced_image(pw, ph, X0[k], Y0[k], DX[k], DY[k]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ JPEG reader
// Baseline / extended-sequential and progressive Huffman JPEG, 8 bit, grey or three components with any sampling
// factors, restart intervals. What `image::io::Reader::decode().to_rgba8()` yields for a JPEG texture in the reference
// comes from the jpeg-decoder crate (0.3.0 per Cargo.lock), whose integer IDCT, triangle-filter chroma upsampling and
// fixed-point YCbCr conversion follow the well-known public-domain formulation (12-bit constants, 10/17-bit descaling;
// (3a + b + 2) >> 2 upsampling; 20-bit BT.601). The same formulation is used here; bit parity with that crate is
// UNPINNED (its source is not in the mount). Rows in file order.
namespace {
struct JpegHuff {
    uint8_t fast[512];
    uint16_t code[256];
    uint8_t values[256], size[257];
    uint32_t maxcode[18];
    int delta[17];
    bool present = false;
    void build(const uint8_t* counts, const uint8_t* vals, int nvals) {
        int k = 0;
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < counts[i]; j++) size[k++] = (uint8_t)(i + 1);
        size[k] = 0;
        if (k != nvals || k > 256) throw std::runtime_error("jpeg: bad Huffman table");
        std::memcpy(values, vals, (size_t)nvals);
        int c = 0;
        k = 0;
        for (int j = 1; j <= 16; j++) {
            delta[j] = k - c;
            if (size[k] == j) {
                while (size[k] == j) code[k++] = (uint16_t)(c++);
                if (c - 1 >= (1 << j)) throw std::runtime_error("jpeg: bad Huffman code lengths");
            }
            maxcode[j] = (uint32_t)c << (16 - j);
            c <<= 1;
        }
        maxcode[17] = 0xffffffffu;
        std::memset(fast, 255, sizeof fast);
        for (int i = 0; i < k; i++) {
            int s = size[i];
            if (s <= 9) {
                int c0 = code[i] << (9 - s), m = 1 << (9 - s);
                for (int j = 0; j < m; j++) fast[c0 + j] = (uint8_t)i;
            }
        }
        present = true;
    }
};
struct JpegBits {
    const uint8_t* p;
    size_t n, pos;
    uint32_t buf = 0;
    int cnt = 0;
    uint8_t marker = 0;  // a marker met while filling (0 = none)
    bool nomore = false;
    void reset() { buf = 0; cnt = 0; marker = 0; nomore = false; }
    void grow() {
        do {
            uint32_t b = 0;
            if (!nomore && pos < n) {
                b = p[pos++];
                if (b == 0xff) {
                    uint8_t c = pos < n ? p[pos++] : 0xd9;
                    while (c == 0xff && pos < n) c = p[pos++];
                    if (c != 0) {
                        marker = c;
                        nomore = true;
                        b = 0;
                    }
                }
            }
            buf |= b << (24 - cnt);
            cnt += 8;
        } while (cnt <= 24);
    }
    int huff(const JpegHuff& h) {
        if (cnt < 16) grow();
        int c = (int)((buf >> 23) & 511u);
        int k = h.fast[c];
        if (k < 255) {
            int s = h.size[k];
            if (s > cnt) throw std::runtime_error("jpeg: truncated entropy data");
            buf <<= s;
            cnt -= s;
            return h.values[k];
        }
        uint32_t temp = buf >> 16;
        int s;
        for (s = 10;; s++)
            if (temp < h.maxcode[s]) break;
        if (s >= 17 || s > cnt) throw std::runtime_error("jpeg: bad Huffman code");
        int idx = (int)((buf >> (32 - s)) & ((1u << s) - 1u)) + h.delta[s];
        if (idx < 0 || idx > 255) throw std::runtime_error("jpeg: bad Huffman code");
        buf <<= s;
        cnt -= s;
        return h.values[idx];
    }
    int extend(int n_) {  // receive n bits and sign-extend (JPEG F.2.2.1)
        if (n_ == 0) return 0;
        if (cnt < n_) grow();
        uint32_t v = buf >> (32 - n_);
        buf <<= n_;
        cnt -= n_;
        int sgn = (int)(v >> (n_ - 1));  // 1 = positive
        return sgn ? (int)v : (int)v - ((1 << n_) - 1);
    }
    int bits(int n_) {
        if (n_ == 0) return 0;
        if (cnt < n_) grow();
        uint32_t v = buf >> (32 - n_);
        buf <<= n_;
        cnt -= n_;
        return (int)v;
    }
    int bit() { return bits(1); }
};
const uint8_t kZigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                  6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                  39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};
struct JpegComp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int bw = 0, bh = 0;  // blocks per row / column (padded to whole MCUs)
    int dc_pred = 0;
    std::vector<int16_t> coef;  // 64 per block
    std::vector<uint8_t> plane; // bw*8 x bh*8 after the IDCT
};
inline uint8_t clamp8(int64_t x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
#define AKR_F2F(x) ((int)(((x) * 4096 + 0.5)))
#define AKR_FSH(x) ((x) * 4096)
#define AKR_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                         \
    int64_t t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3; /* 64 bit: a corrupt file's coefficients must not overflow */ \
    p2 = s2; p3 = s6;                                                        \
    p1 = (p2 + p3) * AKR_F2F(0.5411961f);                                    \
    t2 = p1 + p3 * AKR_F2F(-1.847759065f);                                   \
    t3 = p1 + p2 * AKR_F2F(0.765366865f);                                    \
    p2 = s0; p3 = s4;                                                        \
    t0 = AKR_FSH(p2 + p3); t1 = AKR_FSH(p2 - p3);                            \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                  \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                      \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                  \
    p5 = (p3 + p4) * AKR_F2F(1.175875602f);                                  \
    t0 = t0 * AKR_F2F(0.298631336f); t1 = t1 * AKR_F2F(2.053119869f);        \
    t2 = t2 * AKR_F2F(3.072711026f); t3 = t3 * AKR_F2F(1.501321110f);        \
    p1 = p5 + p1 * AKR_F2F(-0.899976223f); p2 = p5 + p2 * AKR_F2F(-2.562915447f); \
    p3 = p3 * AKR_F2F(-1.961570560f); p4 = p4 * AKR_F2F(-0.390180644f);      \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
void idct_block(uint8_t* out, int stride, const int16_t* coef, const uint16_t* q) {
    int64_t val[64], *v = val, d[64];
    for (int i = 0; i < 64; i++) d[i] = (int64_t)coef[i] * (int64_t)q[i];
    const int64_t* dd = d;
    for (int i = 0; i < 8; i++, dd++, v++) {
        if (dd[8] == 0 && dd[16] == 0 && dd[24] == 0 && dd[32] == 0 && dd[40] == 0 && dd[48] == 0 && dd[56] == 0) {
            int64_t dcterm = dd[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
        } else {
            AKR_IDCT_1D(dd[0], dd[8], dd[16], dd[24], dd[32], dd[40], dd[48], dd[56])
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
    }
    v = val;
    for (int i = 0; i < 8; i++, v += 8, out += stride) {
        AKR_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
        out[0] = clamp8((x0 + t3) >> 17); out[7] = clamp8((x0 - t3) >> 17);
        out[1] = clamp8((x1 + t2) >> 17); out[6] = clamp8((x1 - t2) >> 17);
        out[2] = clamp8((x2 + t1) >> 17); out[5] = clamp8((x2 - t1) >> 17);
        out[3] = clamp8((x3 + t0) >> 17); out[4] = clamp8((x3 - t0) >> 17);
    }
}
#undef AKR_IDCT_1D
// triangle-filter upsampling of one output row
void upsample_h2(uint8_t* out, const uint8_t* in, int w) {  // w input samples -> 2w
    if (w == 1) { out[0] = out[1] = in[0]; return; }
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    int i;
    for (i = 1; i < w - 1; i++) {
        int n = 3 * in[i] + 2;
        out[i * 2 + 0] = (uint8_t)((n + in[i - 1]) >> 2);
        out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
    }
    out[i * 2 + 0] = (uint8_t)((in[w - 1] * 3 + in[w - 2] + 2) >> 2);
    out[i * 2 + 1] = in[w - 1];
}
void upsample_v2(uint8_t* out, const uint8_t* near_, const uint8_t* far_, int w) {
    for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near_[i] + far_[i] + 2) >> 2);
}
void upsample_hv2(uint8_t* out, const uint8_t* near_, const uint8_t* far_, int w) {
    if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near_[0] + far_[0] + 2) >> 2); return; }
    int t0, t1 = 3 * near_[0] + far_[0];
    out[0] = (uint8_t)((t1 + 2) >> 2);
    for (int i = 1; i < w; i++) {
        t0 = t1;
        t1 = 3 * near_[i] + far_[i];
        out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
        out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
    }
    out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
}
}  // namespace

void decode_jpeg(const uint8_t* data, size_t n, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba) {
    if (n < 4 || data[0] != 0xff || data[1] != 0xd8) throw std::runtime_error("jpeg: bad signature");
    size_t pos = 2;
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    JpegHuff hdc[4], hac[4];
    std::vector<JpegComp> comps;
    bool progressive = false, have_frame = false;
    int hmax = 1, vmax = 1, mcux = 0, mcuy = 0, restart_interval = 0;
    int adobe_transform = -1;
    int eobrun = 0;
    auto be16 = [&](size_t p) -> int {
        if (p + 2 > n) throw std::runtime_error("jpeg: truncated");
        return (data[p] << 8) | data[p + 1];
    };
    for (;;) {
        // next marker
        while (pos < n && data[pos] != 0xff) pos++;
        while (pos < n && data[pos] == 0xff) pos++;
        if (pos >= n) break;
        uint8_t m = data[pos++];
        if (m == 0xd9) break;                      // EOI
        if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;  // TEM, stray RSTn
        int len = be16(pos);
        if (len < 2 || pos + (size_t)len > n) throw std::runtime_error("jpeg: bad segment length");
        const uint8_t* seg = data + pos + 2;
        int sl = len - 2;
        if (m == 0xdb) {  // DQT
            int i = 0;
            while (i < sl) {
                int pq = seg[i] >> 4, tq = seg[i] & 15;
                i++;
                if (tq > 3 || pq > 1 || i + 64 * (pq + 1) > sl) throw std::runtime_error("jpeg: bad DQT");
                for (int k = 0; k < 64; k++) {
                    qt[tq][kZigzag[k]] = pq ? (uint16_t)((seg[i] << 8) | seg[i + 1]) : seg[i];
                    i += pq + 1;
                }
                qt_present[tq] = true;
            }
        } else if (m == 0xc4) {  // DHT
            int i = 0;
            while (i < sl) {
                if (i + 17 > sl) throw std::runtime_error("jpeg: bad DHT");
                int tc = seg[i] >> 4, th = seg[i] & 15;
                if (tc > 1 || th > 3) throw std::runtime_error("jpeg: bad DHT");
                int nv = 0;
                for (int k = 0; k < 16; k++) nv += seg[i + 1 + k];
                if (i + 17 + nv > sl || nv > 256) throw std::runtime_error("jpeg: bad DHT");
                (tc ? hac[th] : hdc[th]).build(seg + i + 1, seg + i + 17, nv);
                i += 17 + nv;
            }
        } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {  // SOF0/1/2
            if (have_frame) throw std::runtime_error("jpeg: multiple frames");
            progressive = m == 0xc2;
            if (sl < 6 || seg[0] != 8) throw std::runtime_error("unsupported: JPEG sample precision other than 8 bits");
            height = (uint32_t)((seg[1] << 8) | seg[2]);
            width = (uint32_t)((seg[3] << 8) | seg[4]);
            int nc = seg[5];
            if (width == 0 || height == 0) throw std::runtime_error("jpeg: zero-sized image");
            if ((uint64_t)width * height > (1ull << 28)) throw std::runtime_error("jpeg: image too large");
            if (nc != 1 && nc != 3) throw std::runtime_error("unsupported: JPEG with " + std::to_string(nc) + " components");
            if (sl < 6 + 3 * nc) throw std::runtime_error("jpeg: bad SOF");
            comps.resize((size_t)nc);
            for (int c = 0; c < nc; c++) {
                comps[c].id = seg[6 + 3 * c];
                comps[c].h = seg[7 + 3 * c] >> 4;
                comps[c].v = seg[7 + 3 * c] & 15;
                comps[c].tq = seg[8 + 3 * c];
                if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) throw std::runtime_error("jpeg: bad SOF");
                hmax = std::max(hmax, comps[c].h);
                vmax = std::max(vmax, comps[c].v);
            }
            mcux = ((int)width + 8 * hmax - 1) / (8 * hmax);
            mcuy = ((int)height + 8 * vmax - 1) / (8 * vmax);
            for (auto& c : comps) {
                c.bw = mcux * c.h;
                c.bh = mcuy * c.v;
                c.coef.assign((size_t)c.bw * c.bh * 64, 0);
            }
            have_frame = true;
        } else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) {
            throw std::runtime_error("unsupported: JPEG process (lossless / hierarchical / arithmetic coding)");
        } else if (m == 0xdd) {  // DRI
            if (sl < 2) throw std::runtime_error("jpeg: bad DRI");
            restart_interval = (seg[0] << 8) | seg[1];
        } else if (m == 0xee) {  // APP14 "Adobe"
            if (sl >= 12 && !std::memcmp(seg, "Adobe", 5)) adobe_transform = seg[11];
        } else if (m == 0xda) {  // SOS + entropy-coded data
            if (!have_frame) throw std::runtime_error("jpeg: SOS before SOF");
            int ns = seg[0];
            if (ns < 1 || ns > (int)comps.size() || sl < 4 + 2 * ns) throw std::runtime_error("jpeg: bad SOS");
            int order[3];
            for (int i = 0; i < ns; i++) {
                int which = -1;
                for (size_t c = 0; c < comps.size(); c++)
                    if (comps[c].id == seg[1 + 2 * i]) which = (int)c;
                if (which < 0) throw std::runtime_error("jpeg: bad component in SOS");
                comps[which].td = seg[2 + 2 * i] >> 4;
                comps[which].ta = seg[2 + 2 * i] & 15;
                if (comps[which].td > 3 || comps[which].ta > 3) throw std::runtime_error("jpeg: bad table index in SOS");
                order[i] = which;
            }
            int ss = seg[1 + 2 * ns], se = seg[2 + 2 * ns], ah = seg[3 + 2 * ns] >> 4, al = seg[3 + 2 * ns] & 15;
            if (!progressive) { ss = 0; se = 63; ah = al = 0; }
            else if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13 || (ss == 0 && se != 0)) throw std::runtime_error("jpeg: bad progressive scan");
            JpegBits br{data, n, pos + (size_t)len};
            for (auto& c : comps) c.dc_pred = 0;
            eobrun = 0;
            auto need = [&](const JpegHuff& h) -> const JpegHuff& {
                if (!h.present) throw std::runtime_error("jpeg: missing Huffman table");
                return h;
            };
            auto decode_block = [&](JpegComp& c, int16_t* blk) {
                if (!progressive) {
                    const JpegHuff& dc = need(hdc[c.td]);
                    const JpegHuff& ac = need(hac[c.ta]);
                    int t = br.huff(dc);
                    if (t > 15) throw std::runtime_error("jpeg: bad DC code");
                    int diff = t ? br.extend(t) : 0;
                    c.dc_pred += diff;
                    blk[0] = (int16_t)c.dc_pred;
                    int k = 1;
                    do {
                        int rs = br.huff(ac), s = rs & 15, r = rs >> 4;
                        if (s == 0) {
                            if (rs != 0xf0) break;
                            k += 16;
                        } else {
                            k += r;
                            if (k > 63) throw std::runtime_error("jpeg: bad AC run");
                            blk[kZigzag[k++]] = (int16_t)br.extend(s);
                        }
                    } while (k < 64);
                } else if (ss == 0) {  // DC scan
                    if (ah == 0) {
                        const JpegHuff& dc = need(hdc[c.td]);
                        int t = br.huff(dc);
                        if (t > 15) throw std::runtime_error("jpeg: bad DC code");
                        int diff = t ? br.extend(t) : 0;
                        c.dc_pred += diff;
                        blk[0] = (int16_t)(c.dc_pred * (1 << al));
                    } else if (br.bit()) {
                        blk[0] = (int16_t)(blk[0] + (1 << al));
                    }
                } else if (ah == 0) {  // AC first pass
                    const JpegHuff& ac = need(hac[c.ta]);
                    if (eobrun) { eobrun--; return; }
                    int k = ss;
                    do {
                        int rs = br.huff(ac), s = rs & 15, r = rs >> 4;
                        if (s == 0) {
                            if (r < 15) {
                                eobrun = 1 << r;
                                if (r) eobrun += br.bits(r);
                                eobrun--;
                                break;
                            }
                            k += 16;
                        } else {
                            k += r;
                            if (k > 63) throw std::runtime_error("jpeg: bad AC run");
                            blk[kZigzag[k++]] = (int16_t)(br.extend(s) * (1 << al));
                        }
                    } while (k <= se);
                } else {  // AC refinement
                    const JpegHuff& ac = need(hac[c.ta]);
                    const int16_t bitv = (int16_t)(1 << al);
                    auto refine = [&](int16_t* p) {
                        if (*p != 0 && br.bit() && (*p & bitv) == 0) *p = (int16_t)(*p > 0 ? *p + bitv : *p - bitv);
                    };
                    if (eobrun) {
                        eobrun--;
                        for (int k = ss; k <= se; k++) refine(&blk[kZigzag[k]]);
                        return;
                    }
                    int k = ss;
                    do {
                        int rs = br.huff(ac), s = rs & 15, r = rs >> 4;
                        int newv = 0;
                        if (s == 0) {
                            if (r < 15) {
                                eobrun = (1 << r) - 1;
                                if (r) eobrun += br.bits(r);
                                r = 64;  // force end of block
                            }
                        } else {
                            if (s != 1) throw std::runtime_error("jpeg: bad refinement code");
                            newv = br.bit() ? bitv : -bitv;
                        }
                        while (k <= se) {
                            int16_t* p = &blk[kZigzag[k++]];
                            if (*p != 0) {
                                refine(p);
                            } else {
                                if (r == 0) {
                                    *p = (int16_t)newv;
                                    break;
                                }
                                r--;
                            }
                        }
                    } while (k <= se);
                }
            };
            int todo = restart_interval ? restart_interval : 0x7fffffff;
            auto restart_check = [&]() {
                if (--todo <= 0) {
                    if (br.cnt < 24) br.grow();
                    // a restart marker must follow
                    if (!(br.marker >= 0xd0 && br.marker <= 0xd7)) {
                        // look for it in the byte stream (bits of the current byte are padding)
                        size_t q = br.pos;
                        while (q + 1 < n && !(data[q] == 0xff && data[q + 1] >= 0xd0 && data[q + 1] <= 0xd7)) {
                            if (data[q] == 0xff && data[q + 1] != 0 && data[q + 1] != 0xff) return false;
                            q++;
                        }
                        if (q + 1 >= n) return false;
                        br.pos = q + 2;
                    }
                    br.reset();
                    for (auto& c : comps) c.dc_pred = 0;
                    eobrun = 0;
                    todo = restart_interval;
                }
                return true;
            };
            bool more = true;
            if (ns == 1) {  // non-interleaved: the component's own blocks in raster order, without the MCU padding
                JpegComp& c = comps[order[0]];
                int w = (((int)width * c.h + hmax - 1) / hmax + 7) >> 3, h = (((int)height * c.v + vmax - 1) / vmax + 7) >> 3;
                for (int j = 0; j < h && more; j++)
                    for (int i = 0; i < w && more; i++) {
                        decode_block(c, &c.coef[64 * ((size_t)j * c.bw + i)]);
                        more = restart_check();
                    }
            } else {
                for (int j = 0; j < mcuy && more; j++)
                    for (int i = 0; i < mcux && more; i++) {
                        for (int k = 0; k < ns; k++) {
                            JpegComp& c = comps[order[k]];
                            for (int y = 0; y < c.v; y++)
                                for (int x = 0; x < c.h; x++)
                                    decode_block(c, &c.coef[64 * ((size_t)(j * c.v + y) * c.bw + (i * c.h + x))]);
                        }
                        more = restart_check();
                    }
            }
            // continue after the entropy-coded segment: at the marker the bit reader ran into, or just scan forward
            pos = br.pos;
            if (br.marker) {
                // step back so that the marker loop sees it again
                pos = br.pos >= 2 ? br.pos - 2 : 0;
                while (pos < n && !(data[pos] == 0xff && pos + 1 < n && data[pos + 1] == br.marker)) pos++;
            }
            continue;
        }
        pos += (size_t)len;
    }
    if (!have_frame) throw std::runtime_error("jpeg: no frame");
    // dequantise + IDCT
    for (auto& c : comps) {
        if (!qt_present[c.tq]) throw std::runtime_error("jpeg: missing quantisation table");
        const int stride = c.bw * 8;
        c.plane.assign((size_t)stride * c.bh * 8, 0);
        for (int j = 0; j < c.bh; j++)
            for (int i = 0; i < c.bw; i++) idct_block(&c.plane[(size_t)j * 8 * stride + (size_t)i * 8], stride, &c.coef[64 * ((size_t)j * c.bw + i)], qt[c.tq]);
    }
    // upsample to full resolution, one output row at a time
    const int W = (int)width, H = (int)height;
    rgba.assign(4ull * W * H, 255);
    std::vector<std::vector<uint8_t>> line(comps.size());
    for (auto& l : line) l.resize((size_t)W + 16 * 4 + 8);
    bool rgb_direct = comps.size() == 3 && (adobe_transform == 0 || (comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B'));
    for (int y = 0; y < H; y++) {
        for (size_t ci = 0; ci < comps.size(); ci++) {
            JpegComp& c = comps[ci];
            const int hs = hmax / c.h, vs = vmax / c.v;
            const int stride = c.bw * 8;
            const int cw = (W * c.h + hmax - 1) / hmax, chh = (H * c.v + vmax - 1) / vmax;  // samples of this component
            uint8_t* out = line[ci].data();
            if (hmax % c.h || vmax % c.v) throw std::runtime_error("unsupported: JPEG with fractional sampling ratios");
            if (hs == 1 && vs == 1) {
                std::memcpy(out, &c.plane[(size_t)y * stride], (size_t)W);
            } else if ((hs == 2 && vs == 1) || (hs == 1 && vs == 2) || (hs == 2 && vs == 2)) {
                int sy = vs == 2 ? y >> 1 : y;
                const uint8_t* near_ = &c.plane[(size_t)sy * stride];
                const uint8_t* far_ = near_;
                if (vs == 2) {
                    int fy = (y & 1) ? sy + 1 : sy - 1;
                    fy = fy < 0 ? 0 : (fy > chh - 1 ? chh - 1 : fy);
                    far_ = &c.plane[(size_t)fy * stride];
                }
                if (hs == 2 && vs == 1) upsample_h2(out, near_, cw);
                else if (hs == 1) upsample_v2(out, near_, far_, cw);
                else upsample_hv2(out, near_, far_, cw);
            } else {  // other ratios: replication
                const uint8_t* src = &c.plane[(size_t)(y / vs) * stride];
                for (int x = 0; x < W; x++) out[x] = src[x / hs];
            }
        }
        uint8_t* o = &rgba[4ull * W * y];
        if (comps.size() == 1) {
            for (int x = 0; x < W; x++, o += 4) o[0] = o[1] = o[2] = line[0][x];
        } else if (rgb_direct) {
            for (int x = 0; x < W; x++, o += 4) { o[0] = line[0][x]; o[1] = line[1][x]; o[2] = line[2][x]; }
        } else {
            for (int x = 0; x < W; x++, o += 4) {  // BT.601, 20-bit fixed point
                int yf = (line[0][x] << 20) + (1 << 19);
                int cb = line[1][x] - 128, cr = line[2][x] - 128;
                int r = yf + cr * (int)(1.40200f * 4096.0f + 0.5f) * 256;
                int g = yf + (cr * -((int)(0.71414f * 4096.0f + 0.5f) * 256)) + ((cb * -((int)(0.34414f * 4096.0f + 0.5f) * 256)) & (int)0xffff0000);
                int b = yf + cb * (int)(1.77200f * 4096.0f + 0.5f) * 256;
                o[0] = clamp8(r >> 20); o[1] = clamp8(g >> 20); o[2] = clamp8(b >> 20);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ OpenEXR reader
// Single-part scanline files with NO / RLE / ZIPS / ZIP compression and HALF / FLOAT / UINT channels: what the reference
// gets from `image` (exr crate 1.6.4) as `to_rgba32f()` for an "exr" texture (load.rs:583-611): R, G, B (a lone Y is
// replicated), A = 1 when absent. Round 6: tiled files, PIZ, PXR24, B44 / B44A. Multi-part, deep and DWA files are rejected. Rows in file order
// of the data window (top first).
namespace {
float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u, bits;
    if (e == 0) {
        if (m == 0) {
            bits = sign;
        } else {  // subnormal: normalise
            int sh = 0;
            while (!(m & 1024u)) { m <<= 1; sh++; }
            bits = sign | ((uint32_t)(113 - sh) << 23) | ((m & 1023u) << 13);
        }
    } else if (e == 31) {
        bits = sign | 0x7f800000u | (m << 13);
    } else {
        bits = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}
std::vector<uint8_t> exr_unpredict(const std::vector<uint8_t>& in) {  // delta predictor + even/odd byte interleave of ZIP and RLE
    std::vector<uint8_t> t(in);
    for (size_t i = 1; i < t.size(); i++) t[i] = (uint8_t)(t[i - 1] + t[i] - 128);
    std::vector<uint8_t> out(t.size());
    size_t half = (t.size() + 1) / 2, a = 0, b = half;
    for (size_t i = 0; i < t.size(); i++) out[i] = (i & 1) ? t[b++] : t[a++];
    return out;
}
// ---- PIZ (OpenEXR's wavelet + Huffman compression; round 6). A block's 16-bit words -- per channel a plane of rows x cols x (1 word for
// HALF, 2 for FLOAT / UINT) -- are mapped through a table of the values that occur (a 65536-bit bitmap in the stream), Haar-wavelet
// transformed in place per word plane (14-bit or 16-bit modular variant, by the largest mapped value), and Huffman coded with canonical
// codes of up to 58 bits, a packed table of code lengths with zero runs, and one extra symbol meaning "repeat the last word n times".
struct PizBits {  // MSB-first bit reader
    const uint8_t* p;
    size_t n, pos = 0;
    uint64_t acc = 0;
    int have = 0;
    uint64_t get(int bits) {
        while (have < bits) {
            acc = (acc << 8) | (pos < n ? p[pos] : 0u);  // (reading past the end yields zeros; the caller checks the counts)
            pos++;
            have += 8;
        }
        have -= bits;
        return (acc >> have) & ((bits == 64) ? ~0ull : ((1ull << bits) - 1ull));
    }
};
void piz_huf_decode(const uint8_t* src, size_t n, std::vector<uint16_t>& out, size_t n_out) {
    if (n < 20) throw std::runtime_error("exr: PIZ block too short");
    auto u32 = [&](size_t o) { uint32_t v; std::memcpy(&v, src + o, 4); return v; };
    const uint32_t im = u32(0), iM = u32(4), n_bits = u32(12);
    constexpr uint32_t kEncSize = 65537;
    if (im >= kEncSize || iM >= kEncSize || im > iM) throw std::runtime_error("exr: bad PIZ Huffman header");
    std::vector<uint8_t> len(kEncSize, 0);
    PizBits tb{src + 20, n - 20};
    for (uint32_t i = im; i <= iM;) {  // packed code lengths: 6 bits each; 63 = a zero run of 6 + (8 more bits), 59..62 = a zero run of 2..5
        const uint32_t l = (uint32_t)tb.get(6);
        if (l == 63) {
            const uint32_t run = (uint32_t)tb.get(8) + 6;
            if (i + run > iM + 1) throw std::runtime_error("exr: bad PIZ code table");
            i += run;
        } else if (l >= 59) {
            const uint32_t run = l - 59 + 2;
            if (i + run > iM + 1) throw std::runtime_error("exr: bad PIZ code table");
            i += run;
        } else {
            len[i++] = (uint8_t)l;
        }
    }
    if (tb.pos > n - 20 + 8) throw std::runtime_error("exr: PIZ code table outside the block");
    const size_t table_bytes = (size_t)(tb.pos - (size_t)(tb.have / 8));  // whole bytes consumed (the data starts at the next byte)
    // canonical codes: within a length in symbol order, the LONGEST codes start at 0 (hufCanonicalCodeTable)
    uint64_t count[59] = {0}, base[59] = {0};
    for (uint32_t i = im; i <= iM; i++) count[len[i]]++;
    {
        uint64_t c = 0;
        for (int l = 58; l >= 1; l--) {
            const uint64_t nc = (c + count[l]) >> 1;
            base[l] = c;
            c = nc;
        }
    }
    std::vector<uint32_t> first(60, 0), syms;  // symbols sorted by (length, symbol)
    syms.reserve(iM - im + 1);
    for (int l = 1; l <= 58; l++) {
        first[l] = (uint32_t)syms.size();
        for (uint32_t i = im; i <= iM; i++)
            if (len[i] == l) syms.push_back(i);
    }
    first[59] = (uint32_t)syms.size();
    const uint32_t rlc = iM;
    const size_t data_at = 20 + table_bytes;
    if (data_at > n || ((uint64_t)n_bits + 7) / 8 > n - data_at) throw std::runtime_error("exr: PIZ data outside the block");
    PizBits db{src + data_at, n - data_at};
    out.clear();
    out.reserve(n_out);
    uint64_t used = 0;
    while (used < n_bits) {
        uint64_t code = 0;
        int l = 0;
        uint32_t sym = 0xffffffffu;
        while (l < 58 && used < n_bits) {
            code = (code << 1) | db.get(1);
            l++;
            used++;
            if (count[l] && code >= base[l] && code - base[l] < count[l]) { sym = syms[first[l] + (uint32_t)(code - base[l])]; break; }
        }
        if (sym == 0xffffffffu) {
            if (used >= n_bits) break;  // trailing bits of the last byte
            throw std::runtime_error("exr: bad PIZ Huffman code");
        }
        if (sym == rlc) {
            if (used + 8 > n_bits || out.empty()) throw std::runtime_error("exr: bad PIZ run");
            const uint32_t run = (uint32_t)db.get(8);
            used += 8;
            if (out.size() + run > n_out) throw std::runtime_error("exr: PIZ block decodes to too many words");
            out.insert(out.end(), run, out.back());
        } else {
            if (out.size() >= n_out) throw std::runtime_error("exr: PIZ block decodes to too many words");
            out.push_back((uint16_t)sym);
        }
    }
    if (out.size() != n_out) throw std::runtime_error("exr: PIZ block decodes to the wrong number of words");
}
inline void piz_wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int ls = (int16_t)l, hs = (int16_t)h;
    const int ai = ls + (hs & 1) + (hs >> 1);
    a = (uint16_t)(int16_t)ai;
    b = (uint16_t)(int16_t)(ai - hs);
}
inline void piz_wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xffff;
    const int aa = (d + bb - 0x8000) & 0xffff;
    b = (uint16_t)bb;
    a = (uint16_t)aa;
}
void piz_wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1;
    while (p <= n) p <<= 1;
    p >>= 1;
    int p2 = p;
    p >>= 1;
    auto dec = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { if (w14) piz_wdec14(l, h, a, b); else piz_wdec16(l, h, a, b); };
    while (p >= 1) {
        uint16_t* py = in;
        uint16_t* ey = in + (ptrdiff_t)oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t* px = py;
            uint16_t* ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                dec(*px, *p10, i00, i10);
                dec(*p01, *p11, i01, i11);
                dec(i00, i01, *px, *p01);
                dec(i10, i11, *p10, *p11);
            }
            if (nx & p) {
                uint16_t* p10 = px + oy1;
                dec(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {
            uint16_t* px = py;
            uint16_t* ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t* p01 = px + ox1;
                dec(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p;
        p >>= 1;
    }
}
// one PIZ block -> the block's raw bytes (rows of channel rows, the layout every other method decodes to); types[c] = 0 UINT, 1 HALF, 2 FLOAT
std::vector<uint8_t> piz_decode(const uint8_t* src, size_t n, const std::vector<uint32_t>& types, size_t cols, size_t rows) {
    if (n < 4) throw std::runtime_error("exr: PIZ block too short");
    uint16_t min_nz, max_nz;
    std::memcpy(&min_nz, src, 2);
    std::memcpy(&max_nz, src + 2, 2);
    std::vector<uint8_t> bitmap(8192, 0);
    size_t pos = 4;
    if (min_nz <= max_nz) {
        if (max_nz >= 8192 || pos + (size_t)(max_nz - min_nz + 1) > n) throw std::runtime_error("exr: bad PIZ bitmap");
        std::memcpy(&bitmap[min_nz], src + pos, (size_t)(max_nz - min_nz + 1));
        pos += (size_t)(max_nz - min_nz + 1);
    }
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for (uint32_t i = 0; i < 65536; i++)
        if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t max_value = (uint16_t)(k - 1);
    if (pos + 4 > n) throw std::runtime_error("exr: PIZ block too short");
    int32_t hlen;
    std::memcpy(&hlen, src + pos, 4);
    pos += 4;
    if (hlen < 0 || (size_t)hlen > n - pos) throw std::runtime_error("exr: PIZ Huffman data outside the block");
    size_t n_words = 0;
    for (uint32_t t : types) n_words += cols * rows * (t == 1 ? 1 : 2);
    std::vector<uint16_t> tmp;
    piz_huf_decode(src + pos, (size_t)hlen, tmp, n_words);
    size_t at = 0;
    std::vector<size_t> start(types.size());
    for (size_t c = 0; c < types.size(); c++) {
        const int size = types[c] == 1 ? 1 : 2;
        start[c] = at;
        for (int j = 0; j < size; j++) piz_wav2_decode(&tmp[at + j], (int)cols, size, (int)rows, (int)cols * size, max_value);
        at += cols * rows * size;
    }
    for (uint16_t& w : tmp) w = lut[w];
    std::vector<uint8_t> raw(2 * n_words);
    size_t w = 0;
    std::vector<size_t> cur = start;
    for (size_t r = 0; r < rows; r++)
        for (size_t c = 0; c < types.size(); c++) {
            const size_t words = cols * (types[c] == 1 ? 1 : 2);
            std::memcpy(&raw[w], &tmp[cur[c]], 2 * words);
            w += 2 * words;
            cur[c] += words;
        }
    return raw;
}

// B44 / B44A (lossy, HALF channels only; what the exr crate behind load.rs:586-600 reads as well): a block's channels one after the other;
// a HALF channel in 4 x 4 pixel blocks (the right and bottom edges padded) of 14 bytes each -- the first pixel's 16 bits in an ordered
// representation (sign bit flipped, negative values complemented), a 6-bit shift, fifteen 6-bit running differences, biased by 32,
// down the first column and along the rows, in units of 2^shift -- or, B44A, 3 bytes for a block of one value (shift field >= 13);
// FLOAT and UINT channels are stored as they are. A channel flagged pLinear was packed as 8 log(x): exp(x / 8) of every value.
uint16_t float_to_half_rne(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // rounds to infinity
    if (x < 0x33000001u) return (uint16_t)sign;                // rounds to zero
    const int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift = e < -14 ? 13 + (-14 - e) : 13;  // subnormal halves lose more bits
    uint32_t h = e < -14 ? 0u : (uint32_t)(e + 15) << 10;
    const uint32_t kept = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    uint32_t r = (e < -14 ? kept : (kept & 0x3ffu));
    h |= r;
    if (rem > half || (rem == half && (kept & 1u))) h++;  // (a carry out of the mantissa moves into the exponent: still the right bits)
    return (uint16_t)(sign | h);
}
const uint16_t* b44_exp_table() {  // exp(x / 8) per half bit pattern, as OpenEXR tabulates it (0 for NaN / infinity, HALF_MAX where it overflows)
    static std::vector<uint16_t> t;
    static std::once_flag once;
    std::call_once(once, [] {
        t.resize(65536);
        for (uint32_t i = 0; i < 65536; i++) {
            const float h = half_to_float((uint16_t)i);
            if ((i & 0x7c00u) == 0x7c00u) t[i] = 0;
            else if (h >= 8.0f * (float)std::log(65504.0)) t[i] = 0x7bff;
            else t[i] = float_to_half_rne((float)std::exp((double)h / 8.0));
        }
    });
    return t.data();
}
std::vector<uint8_t> b44_decode(const uint8_t* src, size_t n, const std::vector<uint32_t>& types, const std::vector<uint8_t>& p_linear, size_t cols, size_t rows) {
    std::vector<std::vector<uint8_t>> plane(types.size());  // per channel, rows x cols values, little-endian
    size_t pos = 0;
    for (size_t c = 0; c < types.size(); c++) {
        if (types[c] != 1) {
            const size_t bytes = 4 * cols * rows;
            if (pos + bytes > n) throw std::runtime_error("exr: B44 block too short");
            plane[c].assign(src + pos, src + pos + bytes);
            pos += bytes;
            continue;
        }
        plane[c].resize(2 * cols * rows);
        const uint16_t* table = p_linear[c] ? b44_exp_table() : nullptr;
        for (size_t y = 0; y < rows; y += 4)
            for (size_t x = 0; x < cols; x += 4) {
                if (pos + 3 > n) throw std::runtime_error("exr: B44 block too short");
                const uint8_t* b = src + pos;
                uint16_t s[16];
                if (b[2] >= (13u << 2)) {  // one value for the sixteen pixels
                    for (int i = 0; i < 16; i++) s[i] = (uint16_t)((b[0] << 8) | b[1]);
                    pos += 3;
                } else {
                    if (pos + 14 > n) throw std::runtime_error("exr: B44 block too short");
                    const uint32_t shift = b[2] >> 2, bias = 0x20u << shift;
                    // the fifteen 6-bit fields after the shift, most significant bit first
                    uint32_t r[15];
                    for (int k = 0; k < 15; k++) {
                        const uint32_t bit = 22u + 6u * (uint32_t)k;  // (bits 0-15 the first pixel, 16-21 the shift)
                        const uint32_t w = ((uint32_t)b[bit >> 3] << 8) | (uint32_t)b[(bit >> 3) + 1 < 14 ? (bit >> 3) + 1 : 13];
                        r[k] = (w >> (10u - (bit & 7u))) & 0x3fu;
                    }
                    auto step = [&](uint16_t from, uint32_t rk) { return (uint16_t)(from + (rk << shift) - bias); };
                    s[0] = (uint16_t)((b[0] << 8) | b[1]);
                    s[4] = step(s[0], r[0]); s[8] = step(s[4], r[1]); s[12] = step(s[8], r[2]);          // down the first column
                    for (int col = 1; col < 4; col++)                                                        // then along the four rows
                        for (int row = 0; row < 4; row++) s[4 * row + col] = step(s[4 * row + col - 1], r[3 + 4 * (col - 1) + row]);
                    pos += 14;
                }
                for (int i = 0; i < 16; i++) {
                    s[i] = (s[i] & 0x8000u) ? (uint16_t)(s[i] & 0x7fffu) : (uint16_t)~s[i];  // back from the ordered representation
                    if (table) s[i] = table[s[i]];
                }
                for (size_t yy = 0; yy < 4 && y + yy < rows; yy++)
                    for (size_t xx = 0; xx < 4 && x + xx < cols; xx++) std::memcpy(&plane[c][2 * ((y + yy) * cols + x + xx)], &s[4 * yy + xx], 2);
            }
    }
    if (pos != n) throw std::runtime_error("exr: B44 block has the wrong size");
    // scanline layout: row by row, channel by channel
    std::vector<uint8_t> raw;
    for (size_t y = 0; y < rows; y++)
        for (size_t c = 0; c < types.size(); c++) {
            const size_t sz = types[c] == 1 ? 2 : 4;
            raw.insert(raw.end(), plane[c].begin() + sz * y * cols, plane[c].begin() + sz * (y + 1) * cols);
        }
    return raw;
}

std::vector<uint8_t> exr_rle_decode(const uint8_t* p, size_t n, size_t expect) {
    std::vector<uint8_t> out;
    size_t i = 0;
    while (i < n) {
        int8_t c = (int8_t)p[i++];
        if (c < 0) {
            size_t cnt = (size_t)(-(int)c);
            if (i + cnt > n) throw std::runtime_error("exr: truncated RLE data");
            out.insert(out.end(), p + i, p + i + cnt);
            i += cnt;
        } else {
            if (i >= n) throw std::runtime_error("exr: truncated RLE data");
            out.insert(out.end(), (size_t)c + 1, p[i++]);
        }
        if (out.size() > expect) throw std::runtime_error("exr: RLE data too long");
    }
    return out;
}
}  // namespace

void decode_exr(const uint8_t* data, size_t n, uint32_t& width, uint32_t& height, std::vector<float>& rgba) {
    auto need = [&](size_t pos, size_t len) { if (pos + len > n) throw std::runtime_error("exr: truncated file"); };
    auto rd32 = [&](size_t pos) { need(pos, 4); uint32_t v; std::memcpy(&v, data + pos, 4); return v; };
    auto rd64 = [&](size_t pos) { need(pos, 8); uint64_t v; std::memcpy(&v, data + pos, 8); return v; };
    if (n < 8 || rd32(0) != 20000630u) throw std::runtime_error("exr: bad magic number");
    uint32_t version = rd32(4);
    if ((version & 0xffu) != 2u) throw std::runtime_error("exr: unknown version");
    const bool tiled = (version & 0x200u) != 0;  // single-part tiled file (round 6; the exr crate behind load.rs:586-600 reads them)
    if (version & 0x1800u) throw std::runtime_error("unsupported: deep / multi-part OpenEXR file");
    size_t pos = 8;
    struct Chan { std::string name; uint32_t type; bool p_linear; };
    std::vector<Chan> chans;
    int compression = -1, line_order = 0;
    int32_t dw[4] = {0, 0, -1, -1};
    uint32_t tile_w = 0, tile_h = 0, tile_mode = 0;
    for (;;) {  // attributes
        need(pos, 1);
        if (data[pos] == 0) { pos++; break; }
        auto cstr = [&]() { size_t s0 = pos; while (pos < n && data[pos]) pos++; need(pos, 1); std::string r((const char*)data + s0, pos - s0); pos++; return r; };
        std::string name = cstr(), type = cstr();
        uint32_t size = rd32(pos);
        pos += 4;
        need(pos, size);
        const uint8_t* v = data + pos;
        if (name == "channels") {
            size_t q = 0;
            while (q < size && v[q]) {
                size_t s0 = q;
                while (q < size && v[q]) q++;
                Chan c{std::string((const char*)v + s0, q - s0), 0, false};
                q++;
                if (q + 16 > size) throw std::runtime_error("exr: bad channel list");
                std::memcpy(&c.type, v + q, 4);
                c.p_linear = v[q + 4] != 0;  // (B44: the channel's values were packed on a logarithmic scale)
                uint32_t xs, ys;
                std::memcpy(&xs, v + q + 8, 4);
                std::memcpy(&ys, v + q + 12, 4);
                if (xs != 1 || ys != 1) throw std::runtime_error("unsupported: sub-sampled OpenEXR channels");
                if (c.type > 2) throw std::runtime_error("exr: bad channel type");
                q += 16;
                chans.push_back(c);
            }
        } else if (name == "compression") {
            compression = size ? v[0] : -1;
        } else if (name == "dataWindow") {
            if (size != 16) throw std::runtime_error("exr: bad dataWindow");
            std::memcpy(dw, v, 16);
        } else if (name == "lineOrder") {
            line_order = size ? v[0] : 0;
        } else if (name == "tiles") {  // tiledesc: xSize, ySize, mode (level mode in the low nibble, rounding mode in the high one)
            if (size != 9) throw std::runtime_error("exr: bad tile description");
            std::memcpy(&tile_w, v, 4);
            std::memcpy(&tile_h, v + 4, 4);
            tile_mode = v[8];
        }
        pos += size;
    }
    if (chans.empty() || dw[2] < dw[0] || dw[3] < dw[1]) throw std::runtime_error("exr: missing channels or data window");
    if (compression < 0 || compression > 7)
        throw std::runtime_error("unsupported: OpenEXR compression method " + std::to_string(compression) + " (none, RLE, ZIPS, ZIP, PIZ, PXR24, B44 and B44A are read; DWA is not)");
    if (tiled && (tile_w == 0 || tile_h == 0 || tile_w > 65535 || tile_h > 65535)) throw std::runtime_error("exr: tiled file without a valid tile size");
    if (tiled && (tile_mode & 0xfu) > 2u) throw std::runtime_error("exr: bad tile level mode");
    (void)line_order;  // the offset table is indexed by scanline block in increasing y whatever the order on disk
    const uint64_t W = (uint64_t)((int64_t)dw[2] - (int64_t)dw[0] + 1), H = (uint64_t)((int64_t)dw[3] - (int64_t)dw[1] + 1);
    if (W > 65535 || H > 65535 || W * H > (1ull << 28)) throw std::runtime_error("exr: image too large");
    width = (uint32_t)W;
    height = (uint32_t)H;
    const uint32_t lines_per_block = (compression == 4 || compression >= 6) ? 32u : ((compression == 3 || compression == 5) ? 16u : 1u);
    // chunks: blocks of scanlines, or -- tiled -- the tiles of the full-resolution level (level (0, 0); mip / rip levels follow it in the
    // offset table and are not read), row-major
    const uint64_t tiles_x = tiled ? (W + tile_w - 1) / tile_w : 1, tiles_y = tiled ? (H + tile_h - 1) / tile_h : 0;
    const uint64_t n_blocks = tiled ? tiles_x * tiles_y : (H + lines_per_block - 1) / lines_per_block;
    size_t bytes_per_pixel = 0;
    for (const Chan& c : chans) bytes_per_pixel += (c.type == 1 ? 2 : 4);
    // which file channel feeds which of R, G, B, A
    int src[4] = {-1, -1, -1, -1}, y_chan = -1;
    for (size_t i = 0; i < chans.size(); i++) {
        const std::string& nm = chans[i].name;
        if (nm == "R") src[0] = (int)i; else if (nm == "G") src[1] = (int)i; else if (nm == "B") src[2] = (int)i; else if (nm == "A") src[3] = (int)i;
        else if (nm == "Y") y_chan = (int)i;
    }
    if (src[0] < 0 && src[1] < 0 && src[2] < 0 && y_chan >= 0) src[0] = src[1] = src[2] = y_chan;
    if (src[0] < 0 && src[1] < 0 && src[2] < 0) throw std::runtime_error("unsupported: OpenEXR file without R, G, B or Y channels");
    rgba.assign(4ull * W * H, 0.0f);
    for (uint64_t i = 0; i < W * H; i++) rgba[4 * i + 3] = 1.0f;
    const size_t table = pos;
    need(table, 8 * n_blocks);
    for (uint64_t blk = 0; blk < n_blocks; blk++) {
        uint64_t off = rd64(table + 8 * blk);
        if (off > n) throw std::runtime_error("exr: chunk outside the file");
        int32_t y0;
        uint64_t x0 = 0, cols = W, rows;
        uint32_t csize;
        size_t head;
        if (tiled) {  // tile coordinates, level, size
            need(off, 20);
            int32_t tc[4];
            std::memcpy(tc, data + off, 16);
            csize = rd32(off + 16);
            head = 20;
            if (tc[2] != 0 || tc[3] != 0) throw std::runtime_error("exr: the offset table's first tiles are not those of level 0");
            if (tc[0] < 0 || tc[1] < 0 || (uint64_t)tc[0] >= tiles_x || (uint64_t)tc[1] >= tiles_y) throw std::runtime_error("exr: tile outside the data window");
            x0 = (uint64_t)tc[0] * tile_w;
            y0 = dw[1] + (int32_t)((uint64_t)tc[1] * tile_h);
            cols = std::min<uint64_t>(tile_w, W - x0);
            rows = std::min<uint64_t>(tile_h, H - (uint64_t)tc[1] * tile_h);
        } else {
            need(off, 8);
            std::memcpy(&y0, data + off, 4);
            csize = rd32(off + 4);
            head = 8;
            if (y0 < dw[1] || y0 > dw[3]) throw std::runtime_error("exr: scanline outside the data window");
            rows = std::min<uint64_t>(lines_per_block, (uint64_t)(dw[3] - y0) + 1);
        }
        need(off + head, csize);
        const size_t raw_size = bytes_per_pixel * cols * rows;
        std::vector<uint8_t> raw;
        const uint8_t* src_bytes = data + off + head;
        if (csize == raw_size || compression == 0) {  // stored uncompressed (also when compression did not help)
            if (csize != raw_size) throw std::runtime_error("exr: bad block size");
            raw.assign(src_bytes, src_bytes + csize);
        } else if (compression == 1) {
            raw = exr_unpredict(exr_rle_decode(src_bytes, csize, raw_size));
        } else if (compression == 4) {
            std::vector<uint32_t> types;
            for (const Chan& c : chans) types.push_back(c.type);
            raw = piz_decode(src_bytes, csize, types, (size_t)cols, (size_t)rows);
        } else if (compression >= 6) {
            std::vector<uint32_t> types;
            std::vector<uint8_t> plin;
            for (const Chan& c : chans) { types.push_back(c.type); plin.push_back(c.p_linear ? 1 : 0); }
            raw = b44_decode(src_bytes, csize, types, plin, (size_t)cols, (size_t)rows);
        } else if (compression == 5) {
            // PXR24 (lossy for FLOAT channels: 24 bits kept): zlib over, per scanline and channel, the byte PLANES (most significant first) of the
            // running differences of the pixel values -- 4 planes for UINT, 2 for HALF, 3 for FLOAT (the low byte is dropped)
            const std::vector<uint8_t> planes = inflate_zlib(src_bytes, csize);
            raw.resize(raw_size);
            size_t q = 0, w = 0;
            for (uint64_t r = 0; r < rows; r++)
                for (const Chan& c : chans) {
                    const size_t np = c.type == 0 ? 4 : (c.type == 1 ? 2 : 3);
                    if (q + np * cols > planes.size()) throw std::runtime_error("exr: PXR24 block too short");
                    uint32_t pixel = 0;
                    for (uint64_t x = 0; x < cols; x++) {
                        uint32_t diff = 0;
                        for (size_t k = 0; k < np; k++) diff = (diff << 8) | planes[q + k * cols + x];
                        if (c.type == 2) diff <<= 8;
                        pixel += diff;
                        if (c.type == 1) { const uint16_t hv = (uint16_t)pixel; std::memcpy(&raw[w], &hv, 2); w += 2; }
                        else { std::memcpy(&raw[w], &pixel, 4); w += 4; }
                    }
                    q += np * cols;
                }
        } else {
            raw = exr_unpredict(inflate_zlib(src_bytes, csize));
        }
        if (raw.size() != raw_size) throw std::runtime_error("exr: decompressed block has the wrong size");
        size_t p = 0;
        for (uint64_t r = 0; r < rows; r++) {
            const uint64_t y = (uint64_t)(y0 - dw[1]) + r;
            for (size_t ci = 0; ci < chans.size(); ci++) {  // channels are stored one after the other within a scanline
                const uint32_t ty = chans[ci].type;
                for (uint64_t xx = 0; xx < cols; xx++) {
                    const uint64_t x = x0 + xx;
                    float f;
                    if (ty == 1) { uint16_t hbits; std::memcpy(&hbits, &raw[p], 2); p += 2; f = half_to_float(hbits); }
                    else if (ty == 2) { std::memcpy(&f, &raw[p], 4); p += 4; }
                    else { uint32_t u; std::memcpy(&u, &raw[p], 4); p += 4; f = (float)u; }
                    for (int k = 0; k < 4; k++)
                        if (src[k] == (int)ci) rgba[4 * (y * W + x) + k] = f;
                }
            }
        }
    }
}

}  // namespace akr
