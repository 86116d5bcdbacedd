// image_io.cpp -- output stage of the render driver: the reference's util::write_image
// (crates/akari_render/src/util/mod.rs:57-127): ".exr" -> linear RGB f32 OpenEXR, anything else -> 8-bit sRGB.
// Both writers are self-contained: OpenEXR scanline file with no compression (channels B, G, R as 32-bit float, what
// `exr::prelude::write_rgb_file` produces minus its compression), PNG with stored (uncompressed) deflate blocks.
#include <sys/stat.h>

#include <cmath>
#include <cstdint>
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

}  // namespace akr
