// image_io.cpp -- output stage of the render driver: the reference's util::write_image
// (crates/akari_render/src/util/mod.rs:57-127): ".exr" -> linear RGB f32 OpenEXR, anything else -> 8-bit sRGB.
// Both writers are self-contained: OpenEXR scanline file with no compression (channels B, G, R as 32-bit float, what
// `exr::prelude::write_rgb_file` produces minus its compression), PNG with stored (uncompressed) deflate blocks.
#include <sys/stat.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
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
// palette / low bit depths / tRNS, 16-bit samples rounded to 8, grey replicated, alpha 255 when absent. Interlaced
// files are rejected. Rows come out in file order (top first); the caller flips (load.rs:596).
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

void decode_png(const uint8_t* data, size_t n, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (n < 8 || std::memcmp(data, sig, 8) != 0) throw std::runtime_error("png: bad signature");
    size_t pos = 8;
    uint32_t depth = 0, ctype = 0;
    bool have_hdr = false;
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
            if (body[12] != 0) throw std::runtime_error("unsupported: interlaced PNG");
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
    const size_t bpp_bits = (size_t)channels * depth, stride = ((size_t)w * bpp_bits + 7) / 8, bpp = (bpp_bits + 7) / 8;
    std::vector<uint8_t> raw = inflate_zlib(idat.data(), idat.size());
    if (raw.size() < (stride + 1) * (size_t)h) throw std::runtime_error("png: image data too short");
    std::vector<uint8_t> img(stride * (size_t)h);
    for (uint32_t y = 0; y < h; y++) {  // unfilter
        const uint8_t* src = raw.data() + (stride + 1) * (size_t)y;
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
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* row = img.data() + stride * (size_t)y;
        uint8_t* o = rgba.data() + 4ull * w * y;
        for (uint32_t x = 0; x < w; x++, o += 4) {
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
}

}  // namespace akr
