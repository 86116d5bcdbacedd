// scene_json.cpp -- readers for the reference's on-disk formats.
//
//   scene.json   crates/akari_scenegraph/src/scene.rs:86-117 (Scene, Buffer, BufferView), :333-347 (Mesh/Geometry),
//                :22-57 (TRS / Transform / PerspectiveCamera), shader.rs:117-219 (ShaderNode)
//   loading      crates/akari_render/src/load.rs:129-194 (load_transform, load_camera), :195-237 (load_instance),
//                akari_scenegraph/src/scene.rs:603-647 (MmapScene::open: buffers resolved relative to the JSON)
//   method.json  crates/akari_integrator/src/lib.rs:75-109 (RenderConfig / RenderTask), pt.rs:916-944 (Config)
// Shader graphs are folded to constants here (svm/compiler.rs:116-337 + svm/eval.rs:97-269 for constant
// inputs); graphs with texture nodes are reported as AKR_ERR_UNSUPPORTED by the caller.
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "json.h"
#include "scene_build.h"

namespace akr {

namespace {

std::string read_file(const std::string& path, bool binary) {
    std::ifstream f(path, binary ? std::ios::binary : std::ios::in);
    if (!f) throw std::runtime_error("cannot open '" + path + "'");
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
std::string dirname_of(const std::string& p) {
    size_t k = p.find_last_of('/');
    return k == std::string::npos ? std::string(".") : p.substr(0, k);
}
std::string basename_any(const std::string& p) {  // handles '/' and '\\' (scenes/cbox stores a Windows path)
    size_t k = p.find_last_of("/\\");
    return k == std::string::npos ? p : p.substr(k + 1);
}
bool file_exists(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return (bool)f;
}
std::string base64_decode(const std::string& in) {
    static const std::string tbl = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string out;
    int val = 0, bits = -8;
    for (unsigned char c : in) {
        if (c == '=' || c == '\n' || c == '\r') continue;
        size_t p = tbl.find((char)c);
        if (p == std::string::npos) throw std::runtime_error("bad base64 buffer");
        val = (val << 6) + (int)p;
        bits += 6;
        if (bits >= 0) {
            out.push_back((char)((val >> bits) & 0xFF));
            bits -= 8;
        }
    }
    return out;
}

// 4x4 f32, column-major, glam semantics
struct M4 {
    float m[16];
};
M4 m4_identity() {
    M4 r;
    std::memset(r.m, 0, sizeof r.m);
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
}
M4 m4_mul(const M4& a, const M4& b) {
    M4 r;
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 4; i++)
            r.m[c * 4 + i] = ((a.m[0 * 4 + i] * b.m[c * 4 + 0] + a.m[1 * 4 + i] * b.m[c * 4 + 1]) + a.m[2 * 4 + i] * b.m[c * 4 + 2]) +
                             a.m[3 * 4 + i] * b.m[c * 4 + 3];
    return r;
}
M4 m4_scale(float x, float y, float z) {
    M4 r = m4_identity();
    r.m[0] = x; r.m[5] = y; r.m[10] = z;
    return r;
}
M4 m4_translation(float x, float y, float z) {
    M4 r = m4_identity();
    r.m[12] = x; r.m[13] = y; r.m[14] = z;
    return r;
}
M4 m4_axis_angle(float ax, float ay, float az, float angle) {  // glam Mat4::from_axis_angle
    // sin / cos of the f32 angle, correctly rounded to f32 (double-precision libm, rounded once): what glam's f32::sin_cos
    // returns on every libm that rounds correctly, and bit for bit what the checker's reader (oracle/scene_json.py) computes --
    // sinf / cosf of the host libm may be one ulp off that, which made this reader's camera matrix differ from the other's
    const float s = (float)std::sin((double)angle), c = (float)std::cos((double)angle);
    float sx = ax * s, sy = ay * s, sz = az * s;
    float qx = ax * ax, qy = ay * ay, qz = az * az;
    float omc = 1.0f - c;
    float xyomc = ax * ay * omc, xzomc = ax * az * omc, yzomc = ay * az * omc;
    M4 r = m4_identity();
    r.m[0] = qx * omc + c; r.m[1] = xyomc + sz; r.m[2] = xzomc - sy;
    r.m[4] = xyomc - sz; r.m[5] = qy * omc + c; r.m[6] = yzomc + sx;
    r.m[8] = xzomc + sy; r.m[9] = yzomc - sx; r.m[10] = qz * omc + c;
    return r;
}

// load.rs:129-171
M4 load_transform(const JsonValue& t, bool is_camera) {
    const std::string& ty = t.at("type").as_string();
    const JsonValue& data = t.at("data");
    if (ty == "matrix") {
        // glam::Mat4::from_cols_array_2d(m).transpose(): the JSON rows are the matrix rows
        M4 r;
        for (int row = 0; row < 4; row++)
            for (int col = 0; col < 4; col++) r.m[col * 4 + row] = data.at(row).at(col).as_f32();
        return r;
    }
    if (ty != "trs") throw std::runtime_error("unknown transform type '" + ty + "'");
    float tr[3], ro[3], sc[3];
    for (int i = 0; i < 3; i++) {
        tr[i] = data.at("translation").at(i).as_f32();
        ro[i] = data.at("rotation").at(i).as_f32();
        sc[i] = data.at("scale").at(i).as_f32();
    }
    const std::string& cs = data.at("coordinate_system").as_string();
    M4 m = m4_identity();
    if (!is_camera) m = m4_mul(m4_scale(sc[0], sc[1], sc[2]), m);
    const float kPiF = 3.14159265358979323846f;
    if (cs == "Akari") {
        m = m4_mul(m4_axis_angle(0, 0, 1, ro[2]), m);
        m = m4_mul(m4_axis_angle(1, 0, 0, ro[0]), m);
        m = m4_mul(m4_axis_angle(0, 1, 0, ro[1]), m);
        m = m4_mul(m4_translation(tr[0], tr[1], tr[2]), m);
    } else if (cs == "Blender") {
        if (is_camera) m = m4_mul(m4_axis_angle(1, 0, 0, -kPiF / 2.0f), m);  // Blender cameras look down -Z
        m = m4_mul(m4_axis_angle(1, 0, 0, ro[0]), m);
        m = m4_mul(m4_axis_angle(0, 0, 1, -ro[1]), m);
        m = m4_mul(m4_axis_angle(0, 1, 0, ro[2]), m);
        m = m4_mul(m4_translation(tr[0], tr[2], -tr[1]), m);
    } else {
        throw std::runtime_error("unknown coordinate system '" + cs + "'");
    }
    return m;
}

struct BufferStore {
    const JsonValue& scene;
    std::string base_dir;
    std::map<std::string, std::string> cache;
    const std::string& buffer(const std::string& id) {
        auto it = cache.find(id);
        if (it != cache.end()) return it->second;
        const JsonValue& b = scene.at("buffers").at(id);
        const std::string& ty = b.at("type").as_string();
        std::string data;
        if (ty == "path") {
            const std::string& p = b.at("path").as_string();
            std::string cand = (!p.empty() && p[0] == '/') ? p : base_dir + "/" + p;
            if (!file_exists(cand)) cand = base_dir + "/" + basename_any(p);
            data = read_file(cand, true);
        } else if (ty == "base64") {
            data = base64_decode(b.at("data").as_string());
        } else if (ty == "binary") {
            for (const auto& v : b.at("data").arr) data.push_back((char)(unsigned char)v->as_number());
        } else {
            throw std::runtime_error("unsupported buffer type '" + ty + "'");
        }
        return cache[id] = std::move(data);
    }
    template <typename T>
    std::vector<T> view(const JsonValue& ref) {
        const JsonValue& v = scene.at("buffer_views").at(ref.at("id").as_string());
        const std::string& data = buffer(v.at("buffer").at("id").as_string());
        const double off_d = v.at("offset").as_number(), len_d = v.at("length").as_number();
        // validated as numbers first: a negative or huge value must not wrap around in size_t arithmetic
        if (!(off_d >= 0.0 && len_d >= 0.0 && off_d <= (double)data.size() && len_d <= (double)data.size() - off_d))
            throw std::runtime_error("buffer view out of range");
        const size_t off = (size_t)off_d, len = (size_t)len_d;
        std::vector<T> out(len / sizeof(T));
        if (!out.empty()) std::memcpy(out.data(), data.data() + off, out.size() * sizeof(T));
        return out;
    }
};

struct ConstVal {
    float v[4];
    int n;
    bool aces = false;  // an Rgb constant declared in ACEScg (akari_scenegraph ColorSpace "aces")
};
bool parse_rgb_colorspace(const JsonValue& n) {  // -> is ACEScg
    const std::string cs = n.has("colorspace") ? n.at("colorspace").as_string() : std::string("srgb");
    if (cs == "srgb") return false;
    if (cs == "aces") return true;
    throw std::runtime_error("unsupported: constant colour in colour space '" + cs + "' (RgbColorSpace is srgb | aces, color.rs:6-11)");
}
// constant evaluation of a shader node (svm/eval.rs:97-135): Float, Float3, Rgb (+alpha 1), SpectralUplift
ConstVal fold_const(const JsonValue& nodes, const JsonValue& ref) {
    const JsonValue& n = nodes.at(ref.at("id").as_string());
    const std::string& ty = n.at("type").as_string();
    ConstVal c{{0, 0, 0, 1}, 1, false};
    if (ty == "float") {
        c.v[0] = n.at("value").as_f32();
        c.n = 1;
    } else if (ty == "float3") {
        for (int i = 0; i < 3; i++) c.v[i] = n.at("value").at(i).as_f32();
        c.n = 3;
    } else if (ty == "rgb") {
        c.aces = parse_rgb_colorspace(n);
        for (int i = 0; i < 3; i++) c.v[i] = n.at("value").at(i).as_f32();
        c.v[3] = 1.0f;
        c.n = 4;
    } else if (ty == "spectral_uplift") {
        return fold_const(nodes, n.at("rgb"));
    } else {
        throw std::runtime_error("unsupported: shader node '" + ty + "' (only constant inputs are folded)");
    }
    return c;
}
float fold_float(const JsonValue& nodes, const JsonValue& ref) { return fold_const(nodes, ref).v[0]; }  // eval_float_auto_convert
bool fold_color(const JsonValue& nodes, const JsonValue& ref, float* rgb, float* alpha) {  // -> the constant is in ACEScg
    ConstVal c = fold_const(nodes, ref);
    if (c.n >= 3) { rgb[0] = c.v[0]; rgb[1] = c.v[1]; rgb[2] = c.v[2]; }
    else { rgb[0] = c.v[0]; rgb[1] = 0.0f; rgb[2] = 0.0f; }
    if (alpha) *alpha = c.n == 4 ? c.v[3] : 1.0f;
    return c.aces;
}

// true when the node evaluates to the same value everywhere and fold_const can do it
bool is_const_tree(const JsonValue& nodes, const JsonValue& ref) {
    const JsonValue& n = nodes.at(ref.at("id").as_string());
    const std::string& ty = n.at("type").as_string();
    if (ty == "float" || ty == "float3" || ty == "rgb") return true;
    if (ty == "spectral_uplift") return is_const_tree(nodes, n.at("rgb"));
    return false;
}

// Translates the non-constant part of a shader graph into akr_shader_node lists and registers its images.
struct GraphBuilder {
    const JsonValue& nodes;
    BufferStore& bufs;
    FlatScene& flat;
    std::map<std::string, uint32_t>& image_index;  // (buffer view + sampler) key -> image
    HostGraph graph;
    std::map<std::string, uint32_t> memo;

    uint32_t push(uint32_t op, uint32_t a0 = AKR_NODE_NONE, uint32_t a1 = AKR_NODE_NONE, uint32_t a2 = AKR_NODE_NONE, uint32_t a3 = AKR_NODE_NONE,
                  float k0 = 0.0f, float k1 = 0.0f, float k2 = 0.0f) {
        akr_shader_node nd;
        nd.op = op;
        nd.arg[0] = a0; nd.arg[1] = a1; nd.arg[2] = a2; nd.arg[3] = a3;
        nd.k[0] = k0; nd.k[1] = k1; nd.k[2] = k2;
        graph.nodes.push_back(nd);
        return (uint32_t)graph.nodes.size() - 1;
    }
    uint32_t image(const JsonValue& im) {  // load.rs:477-489, 543-611
        const std::string view = im.at("data").at("id").as_string();
        const std::string fmt = im.at("format").as_string(), ext = im.at("extension").as_string(), interp = im.at("interpolation").as_string();
        const uint32_t w = (uint32_t)im.at("width").as_number(), h = (uint32_t)im.at("height").as_number(), ch = (uint32_t)im.at("channels").as_number();
        const std::string key = view + "|" + fmt + "|" + ext + "|" + interp + "|" + std::to_string(w) + "x" + std::to_string(h) + "x" + std::to_string(ch);
        auto it = image_index.find(key);
        if (it != image_index.end()) return it->second;
        HostImage hi;
        hi.address = ext == "repeat" ? AKR_TEX_REPEAT : ext == "clip" ? AKR_TEX_CLIP : ext == "mirror" ? AKR_TEX_MIRROR : ext == "extend" ? AKR_TEX_EXTEND : 99u;
        if (hi.address == 99u) throw std::runtime_error("unknown image extension '" + ext + "'");
        if (interp == "linear" || interp == "cubic") hi.filter = AKR_TEX_FILTER_LINEAR;  // load.rs:690-699
        else if (interp == "nearest") hi.filter = AKR_TEX_FILTER_NEAREST;
        else throw std::runtime_error("unknown image interpolation '" + interp + "'");
        std::vector<uint8_t> bytes = bufs.view<uint8_t>(im.at("data"));
        if (fmt == "float") {
            if (ch == 0 || ch > 4) throw std::runtime_error("image: invalid number of channels");
            if (bytes.size() != (size_t)w * h * ch * 4) throw std::runtime_error("float image: buffer size does not match width * height * channels");
            hi.width = w; hi.height = h; hi.format = AKR_IMAGE_RGBA32F;
            hi.words.resize(4ull * w * h);
            const uint32_t one = 0x3f800000u;
            for (size_t t = 0; t < (size_t)w * h; t++)
                for (uint32_t c = 0; c < 4; c++) {
                    uint32_t v = c == 3 ? one : 0u;
                    if (c < ch) std::memcpy(&v, bytes.data() + 4 * (t * ch + c), 4);
                    hi.words[4 * t + c] = v;
                }
        } else if (fmt == "png") {
            uint32_t pw = 0, ph = 0;
            std::vector<uint8_t> px;
            decode_png(bytes.data(), bytes.size(), pw, ph, px);
            hi.width = pw; hi.height = ph; hi.format = AKR_IMAGE_RGBA8;
            hi.words.resize((size_t)pw * ph);
            for (uint32_t y = 0; y < ph; y++)  // flipv (load.rs:596): row 0 of the texture is the bottom row of the file
                std::memcpy(hi.words.data() + (size_t)y * pw, px.data() + 4ull * pw * (ph - 1 - y), 4ull * pw);
        } else if (fmt == "jpeg" || fmt == "tiff" || fmt == "dds") {
            uint32_t pw = 0, ph = 0;
            std::vector<uint8_t> px;
            if (fmt == "jpeg") decode_jpeg(bytes.data(), bytes.size(), pw, ph, px);
            else if (fmt == "tiff") decode_tiff(bytes.data(), bytes.size(), pw, ph, px);
            else decode_dds(bytes.data(), bytes.size(), pw, ph, px);
            hi.width = pw; hi.height = ph; hi.format = AKR_IMAGE_RGBA8;
            hi.words.resize((size_t)pw * ph);
            for (uint32_t y = 0; y < ph; y++) std::memcpy(hi.words.data() + (size_t)y * pw, px.data() + 4ull * pw * (ph - 1 - y), 4ull * pw);
        } else if (fmt == "exr") {  // to_rgba32f, flipped like every encoded image (load.rs:596-611)
            uint32_t pw = 0, ph = 0;
            std::vector<float> px;
            decode_exr(bytes.data(), bytes.size(), pw, ph, px);
            hi.width = pw; hi.height = ph; hi.format = AKR_IMAGE_RGBA32F;
            hi.words.resize(4ull * pw * ph);
            for (uint32_t y = 0; y < ph; y++) std::memcpy(hi.words.data() + 4ull * pw * y, px.data() + 4ull * pw * (ph - 1 - y), 16ull * pw);
        } else {
            throw std::runtime_error("unsupported: image format '" + fmt + "' (float, png, jpeg, tiff, exr and dds are read here: load.rs:585-592)");
        }
        uint32_t idx = (uint32_t)flat.images.size();
        flat.images.push_back(std::move(hi));
        image_index[key] = idx;
        return idx;
    }
    uint32_t emit(const JsonValue& ref) {
        const std::string& id = ref.at("id").as_string();
        auto it = memo.find(id);
        if (it != memo.end()) return it->second;
        const JsonValue& n = nodes.at(id);
        const std::string& ty = n.at("type").as_string();
        uint32_t r;
        if (ty == "float") {
            r = push(AKR_NODE_CONST, AKR_NODE_NONE, AKR_NODE_NONE, AKR_NODE_NONE, AKR_NODE_NONE, n.at("value").as_f32());
        } else if (ty == "float3") {
            r = push(AKR_NODE_CONST, AKR_NODE_NONE, AKR_NODE_NONE, AKR_NODE_NONE, AKR_NODE_NONE, n.at("value").at(0).as_f32(), n.at("value").at(1).as_f32(),
                     n.at("value").at(2).as_f32());
        } else if (ty == "rgb") {
            r = push(AKR_NODE_RGB, parse_rgb_colorspace(n) ? 1u : 0u, AKR_NODE_NONE, AKR_NODE_NONE, AKR_NODE_NONE, n.at("value").at(0).as_f32(),
                     n.at("value").at(1).as_f32(), n.at("value").at(2).as_f32());
        } else if (ty == "spectral_uplift") {
            r = push(AKR_NODE_SPECTRAL_UPLIFT, emit(n.at("rgb")));
        } else if (ty == "texcoords") {
            r = push(AKR_NODE_TEXCOORDS);
        } else if (ty == "image") {
            const JsonValue& im = n.at("image");
            uint32_t uv = (n.has("uv") && !n.at("uv").is_null()) ? emit(n.at("uv")) : AKR_NODE_NONE;
            const std::string cs = im.at("colorspace").as_string();
            if (cs != "srgb" && cs != "none") throw std::runtime_error("unsupported: image colour space '" + cs + "'");
            r = push(AKR_NODE_IMAGE, image(im), uv, cs == "srgb" ? 1u : 0u);
        } else if (ty == "mapping") {
            const std::string& mt = n.at("mapping").as_string();
            if (mt != "point" && mt != "texture") throw std::runtime_error("unknown mapping type '" + mt + "'");
            uint32_t v = emit(n.at("vector")), loc = emit(n.at("location")), sc = emit(n.at("scale"));
            r = push(AKR_NODE_MAPPING, v, loc, sc, mt == "point" ? AKR_MAPPING_POINT : AKR_MAPPING_TEXTURE);
        } else if (ty == "checkerboard") {
            uint32_t v = (n.has("vector") && !n.at("vector").is_null()) ? emit(n.at("vector")) : AKR_NODE_NONE;
            uint32_t sc = emit(n.at("scale")), c1 = emit(n.at("color1")), c2 = emit(n.at("color2"));
            r = push(AKR_NODE_CHECKERBOARD, v, sc, c1, c2);
        } else if (ty == "normal_map") {
            if (n.at("space").as_string() != "tangent") throw std::runtime_error("unsupported: only tangent space normal maps (svm/eval.rs:185-189)");
            uint32_t nn = emit(n.at("normal")), st = emit(n.at("strength"));
            r = push(AKR_NODE_NORMAL_MAP, nn, st);
        } else if (ty == "separate_color") {
            if (n.at("mode").as_string() != "rgb") throw std::runtime_error("unknown separate_color mode");
            r = push(AKR_NODE_SEPARATE_COLOR, emit(n.at("color")));
        } else if (ty == "extract") {
            const std::string& f = n.at("field").as_string();
            uint32_t field = f == "Red" ? AKR_FIELD_RED : f == "Green" ? AKR_FIELD_GREEN : f == "Blue" ? AKR_FIELD_BLUE : (f == "uv" || f == "UV") ? AKR_FIELD_UV : 99u;
            if (field == 99u) throw std::runtime_error("unsupported: extract field '" + f + "'");
            r = push(AKR_NODE_EXTRACT, emit(n.at("node")), field);
        } else {
            throw std::runtime_error("unsupported: shader node '" + ty + "'");
        }
        memo[id] = r;
        return r;
    }
};

void load_shader(const JsonValue& shader, BufferStore& bufs, FlatScene& flat, std::map<std::string, uint32_t>& image_index, akr_material_desc& m,
                 HostGraph& graph_out) {
    const JsonValue& nodes = shader.at("nodes");
    const JsonValue& out = nodes.at(shader.at("output").at("id").as_string());
    if (out.at("type").as_string() != "output") throw std::runtime_error("shader graph output is not an output node");
    const JsonValue& n = nodes.at(out.at("node").at("id").as_string());
    const std::string& ty = n.at("type").as_string();
    std::memset(&m, 0, sizeof m);
    m.base_alpha = 1.0f;
    m.ior = 1.0f;
    for (int i = 0; i < 3; i++) m.specular_tint[i] = m.coat_tint[i] = 1.0f;
    m.specular_ior_level = 0.5f;
    GraphBuilder gb{nodes, bufs, flat, image_index, HostGraph(), {}};
    // constant inputs are folded here (svm/eval.rs:97-135), everything else becomes graph nodes
    uint32_t cs_flags = 0;  // AKR_MAT_CS_*: which folded colour constants are ACEScg
    auto color = [&](const char* key, uint32_t input, float* rgb, float* alpha) {
        if (is_const_tree(nodes, n.at(key))) {
            if (fold_color(nodes, n.at(key), rgb, alpha))
                cs_flags |= input == AKR_IN_BASE_COLOR ? AKR_MAT_CS_BASE_COLOR : input == AKR_IN_SPECULAR_TINT ? AKR_MAT_CS_SPECULAR_TINT
                          : input == AKR_IN_COAT_TINT ? AKR_MAT_CS_COAT_TINT : input == AKR_IN_EMISSION_COLOR ? AKR_MAT_CS_EMISSION_COLOR : 0u;
        } else {
            gb.graph.input[input] = gb.emit(n.at(key));
        }
    };
    auto scalar = [&](const char* key, uint32_t input, float* v) {
        if (is_const_tree(nodes, n.at(key))) *v = fold_float(nodes, n.at(key));
        else gb.graph.input[input] = gb.emit(n.at(key));
    };
    if (ty == "principled") {
        m.kind = AKR_MAT_PRINCIPLED;
        color("base_color", AKR_IN_BASE_COLOR, m.base_color, &m.base_alpha);
        scalar("metallic", AKR_IN_METALLIC, &m.metallic);
        scalar("roughness", AKR_IN_ROUGHNESS, &m.roughness);
        scalar("ior", AKR_IN_IOR, &m.ior);
        scalar("specular_ior_level", AKR_IN_SPECULAR_IOR_LEVEL, &m.specular_ior_level);
        color("specular_tint", AKR_IN_SPECULAR_TINT, m.specular_tint, nullptr);
        scalar("transmission_weight", AKR_IN_TRANSMISSION_WEIGHT, &m.transmission_weight);
        scalar("coat_weight", AKR_IN_COAT_WEIGHT, &m.coat_weight);
        scalar("coat_roughness", AKR_IN_COAT_ROUGHNESS, &m.coat_roughness);
        scalar("coat_ior", AKR_IN_COAT_IOR, &m.coat_ior);
        color("coat_tint", AKR_IN_COAT_TINT, m.coat_tint, nullptr);
        color("emission_color", AKR_IN_EMISSION_COLOR, m.emission_color, nullptr);
        scalar("emission_strength", AKR_IN_EMISSION_STRENGTH, &m.emission_strength);
        color("normal", AKR_IN_NORMAL, m.normal, nullptr);
    } else if (ty == "diffuse") {
        m.kind = AKR_MAT_DIFFUSE;
        color("color", AKR_IN_BASE_COLOR, m.base_color, &m.base_alpha);
    } else if (ty == "glass") {
        m.kind = AKR_MAT_GLASS;
        color("color", AKR_IN_BASE_COLOR, m.base_color, nullptr);
        scalar("ior", AKR_IN_IOR, &m.ior);
        scalar("roughness", AKR_IN_ROUGHNESS, &m.roughness);
    } else if (ty == "emission") {
        m.kind = AKR_MAT_EMISSION;
        color("color", AKR_IN_EMISSION_COLOR, m.emission_color, nullptr);
        scalar("strength", AKR_IN_EMISSION_STRENGTH, &m.emission_strength);
    } else {
        throw std::runtime_error("unsupported: surface shader '" + ty + "'");
    }
    m.kind |= cs_flags;
    graph_out = std::move(gb.graph);
}

}  // namespace

FlatScene load_scene_json(const std::string& path) {
    JsonPtr root = JsonParser::parse(read_file(path, false));
    const JsonValue& scene = *root;
    BufferStore bufs{scene, dirname_of(path), {}};
    FlatScene flat;
    // geometries (BTreeMap order)
    std::map<std::string, uint32_t> geom_index, mat_index;
    for (const auto& kv : scene.at("geometries").obj) {
        const JsonValue& g = *kv.second;
        if (g.at("type").as_string() != "mesh") throw std::runtime_error("unsupported geometry type");
        HostMesh m;
        m.vertices = bufs.view<float>(g.at("vertices"));
        m.indices = bufs.view<uint32_t>(g.at("indices"));
        if (g.has("uvs")) m.uvs = bufs.view<float>(g.at("uvs"));
        if (g.has("normals")) m.normals = bufs.view<float>(g.at("normals"));
        if (g.has("tangents")) m.tangents = bufs.view<float>(g.at("tangents"));
        if (g.has("materials")) m.slots = bufs.view<uint32_t>(g.at("materials"));
        const size_t nt = m.indices.size() / 3;
        if (m.indices.empty() || m.indices.size() % 3) throw std::runtime_error("mesh '" + kv.first + "': bad index buffer");
        if (!m.uvs.empty() && m.uvs.size() != 6 * nt) throw std::runtime_error("mesh '" + kv.first + "': uvs must be per corner");
        if (!m.normals.empty() && m.normals.size() != 9 * nt) throw std::runtime_error("mesh '" + kv.first + "': normals must be per corner");
        if (!m.tangents.empty() && m.tangents.size() != 9 * nt) throw std::runtime_error("mesh '" + kv.first + "': tangents must be per corner");
        if (m.slots.size() <= 1) m.slots.clear();  // one entry = slot 0 for all triangles (mesh.rs:139)
        else if (m.slots.size() != nt) throw std::runtime_error("mesh '" + kv.first + "': material slots must be 1 or per triangle");
        for (uint32_t idx : m.indices)
            if (idx >= m.vertices.size() / 3) throw std::runtime_error("mesh '" + kv.first + "': vertex index out of range");
        geom_index[kv.first] = (uint32_t)flat.meshes.size();
        flat.meshes.push_back(std::move(m));
    }
    std::map<std::string, uint32_t> image_index;
    bool any_graph = false;
    std::vector<HostGraph> graphs;
    for (const auto& kv : scene.at("materials").obj) {
        mat_index[kv.first] = (uint32_t)flat.materials.size();
        try {
            akr_material_desc m;
            HostGraph g;
            load_shader(kv.second->at("shader"), bufs, flat, image_index, m, g);
            flat.materials.push_back(m);
            any_graph = any_graph || !g.nodes.empty();
            graphs.push_back(std::move(g));
        } catch (const std::exception& e) {
            throw std::runtime_error("material '" + kv.first + "': " + e.what());
        }
    }
    if (any_graph) flat.graphs = std::move(graphs);
    for (const auto& kv : scene.at("instances").obj) {
        const JsonValue& in = *kv.second;
        HostInstance h;
        auto gi = geom_index.find(in.at("geometry").at("id").as_string());
        if (gi == geom_index.end()) throw std::runtime_error("instance '" + kv.first + "': unknown geometry");
        h.mesh = gi->second;
        for (const auto& mref : in.at("materials").arr) {
            auto mi = mat_index.find(mref->at("id").as_string());
            if (mi == mat_index.end()) throw std::runtime_error("instance '" + kv.first + "': unknown material");
            h.materials.push_back(mi->second);
        }
        if (h.materials.empty()) throw std::runtime_error("instance '" + kv.first + "': no materials");
        M4 t = load_transform(in.at("transform"), false);
        std::memcpy(h.transform, t.m, sizeof h.transform);
        flat.instances.push_back(std::move(h));
    }
    if (!scene.has("camera")) throw std::runtime_error("scene has no camera");
    const JsonValue& cam = scene.at("camera");
    if (cam.at("type").as_string() != "perspective") throw std::runtime_error("unsupported camera type");
    const JsonValue& cd = cam.at("data");
    M4 c2w = load_transform(cd.at("transform"), true);
    std::memcpy(flat.camera.c2w, c2w.m, sizeof flat.camera.c2w);
    const float kPiF = 3.14159265358979323846f;
    flat.camera.fov = cd.at("fov").as_f32() * (kPiF / 180.0f);  // f32::to_radians
    flat.camera.width = (uint32_t)cd.at("sensor_width").as_number();
    flat.camera.height = (uint32_t)cd.at("sensor_height").as_number();
    return flat;
}

static void parse_one_task(const JsonValue* j, akr_pt_config* cfg, std::string* film_out, bool allow_sampler_override, ParsedTask* task = nullptr) {
    akr_pt_config_default(cfg);
    if (task) {
        akr_aov_config_default(&task->aov);
        akr_gpt_config_default(&task->gpt);
        akr_mcmc_config_default(&task->mcmc);
    }
    if (film_out) *film_out = "out.exr";  // FilmConfig::default, lib.rs:82-90
    if (j->has("method")) {
        const JsonValue& m = j->at("method");
        const std::string ty = m.has("type") ? m.at("type").as_string() : std::string("pt");
        if (ty == "aov" && task) {  // aov::Config (aov.rs:23-39)
            task->is_aov = true;
            if (m.has("spp")) task->aov.spp = (uint32_t)m.at("spp").as_number();
            if (m.has("remap")) task->aov.remap = m.at("remap").as_bool() ? 1u : 0u;
            if (m.has("aov")) {
                const std::string& a = m.at("aov").as_string();
                static const char* names[6] = {"ns", "ng", "tangent", "bitangent", "albedo", "roughness"};
                uint32_t k = 0;
                while (k < 6 && a != names[k]) k++;
                if (k == 6) throw std::runtime_error("unknown aov '" + a + "'");
                task->aov.aov = k;
            }
        } else if (ty == "gpt" && task) {  // gpt::Config (gpt.rs:32-65)
            task->is_gpt = true;
            akr_gpt_config& g = task->gpt;
            auto gu = [&](const char* k, uint32_t& dst) { if (m.has(k)) dst = (uint32_t)m.at(k).as_number(); };
            auto gb = [&](const char* k, uint32_t& dst) { if (m.has(k)) dst = m.at(k).as_bool() ? 1u : 0u; };
            gu("spp", g.spp); gu("max_depth", g.max_depth); gu("spp_per_pass", g.spp_per_pass); gu("rr_depth", g.rr_depth);
            gu("stride", g.stride); gu("reconstruction_iter", g.reconstruction_iter);
            gb("use_nee", g.use_nee); gb("indirect_only", g.indirect_only); gb("reconnect", g.reconnect); gb("separate_weights", g.separate_weights);
            if (m.has("seed")) g.seed = (uint64_t)m.at("seed").as_number();
            if (m.has("reconstruction")) {
                const std::string& r = m.at("reconstruction").as_string();
                if (r == "none") g.reconstruction = AKR_GPT_RECON_NONE;
                else if (r == "uniform") g.reconstruction = AKR_GPT_RECON_UNIFORM;
                else if (r == "weighted") g.reconstruction = AKR_GPT_RECON_WEIGHTED;
                else throw std::runtime_error("unknown reconstruction '" + r + "'");
            }
        } else if (ty == "mcmc_opt" && task) {  // mcmc::Config (mcmc.rs:44-80), Method::Kelemen (mcmc.rs:8-32)
            task->is_mcmc = true;
            akr_mcmc_config& g = task->mcmc;
            auto gu = [&](const char* k, uint32_t& dst) { if (m.has(k)) dst = (uint32_t)m.at(k).as_number(); };
            auto gb = [&](const JsonValue& o, const char* k, uint32_t& dst) { if (o.has(k)) dst = o.at(k).as_bool() ? 1u : 0u; };
            gu("spp", g.spp); gu("max_depth", g.max_depth); gu("spp_per_pass", g.spp_per_pass); gu("rr_depth", g.rr_depth);
            gu("n_chains", g.n_chains); gu("n_bootstrap", g.n_bootstrap);
            gb(m, "use_nee", g.use_nee); gb(m, "wis", g.wis);
            if (m.has("mcmc_depth") && m.at("mcmc_depth").type != JsonValue::Null) g.mcmc_depth = (uint32_t)m.at("mcmc_depth").as_number();
            if (m.has("direct_spp")) g.direct_spp = (int32_t)m.at("direct_spp").as_number();
            if (m.has("seed")) g.seed = (uint64_t)m.at("seed").as_number();
            if (m.has("method")) {
                const JsonValue& k = m.at("method");
                if (k.has("type") && k.at("type").as_string() != "kelemen") throw std::runtime_error("unknown mcmc method '" + k.at("type").as_string() + "'");
                gb(k, "exponential_mutation", g.exponential_mutation); gb(k, "adaptive", g.adaptive);
                if (k.has("small_sigma")) g.small_sigma = k.at("small_sigma").as_f32();
                if (k.has("large_step_prob")) g.large_step_prob = k.at("large_step_prob").as_f32();
                if (k.has("image_mutation_prob")) g.image_mutation_prob = k.at("image_mutation_prob").as_f32();
                if (k.has("image_mutation_size") && k.at("image_mutation_size").type != JsonValue::Null) g.image_mutation_size = k.at("image_mutation_size").as_f32();
            }
        } else if (ty != "pt") {
            throw std::runtime_error("unsupported: method type '" + ty + "' (\"pt\", \"aov\", \"gpt\" and \"mcmc_opt\" are implemented" + (task ? ")" : "; this entry point takes \"pt\" only)"));
        }
        auto u32 = [&](const char* k, uint32_t& dst) { if (m.has(k)) dst = (uint32_t)m.at(k).as_number(); };
        auto b32 = [&](const char* k, uint32_t& dst) { if (m.has(k)) dst = m.at(k).as_bool() ? 1u : 0u; };
        u32("spp", cfg->spp); u32("max_depth", cfg->max_depth); u32("spp_per_pass", cfg->spp_per_pass); u32("rr_depth", cfg->rr_depth);
        b32("use_nee", cfg->use_nee); b32("indirect_only", cfg->indirect_only); b32("force_diffuse", cfg->force_diffuse);
        if (m.has("pixel_offset")) {
            cfg->pixel_offset[0] = (int32_t)m.at("pixel_offset").at(0).as_number();
            cfg->pixel_offset[1] = (int32_t)m.at("pixel_offset").at(1).as_number();
        }
        if (m.has("debug_depth")) cfg->debug_depth = (int32_t)m.at("debug_depth").as_number();
    }
    if (j->has("sampler")) {
        const JsonValue& s = j->at("sampler");
        const std::string ty = s.has("type") ? s.at("type").as_string() : std::string("independent");
        if (ty == "independent" || (ty == "pmj02bn" && allow_sampler_override)) cfg->sampler_type = AKR_SAMPLER_INDEPENDENT;
        else if (ty == "pmj02bn") cfg->sampler_type = AKR_SAMPLER_PMJ02BN;  // on regenerated tables, see pmj_tables.cpp
        else if (ty == "sobol") cfg->sampler_type = AKR_SAMPLER_SOBOL;      // Owen-scrambled Sobol' (0,2), device/drng.h
        else throw std::runtime_error("unknown sampler '" + ty + "'");
        if (s.has("seed")) cfg->sampler_seed = (uint64_t)s.at("seed").as_number();
    }
    if (j->has("color")) {  // ColorPipeline (color.rs:663-676)
        const JsonValue& c = j->at("color");
        auto space = [](const std::string& v) -> bool {  // -> is ACEScg
            if (v == "srgb") return false;
            if (v == "aces") return true;
            throw std::runtime_error("unknown rgb colour space '" + v + "' (srgb | aces, color.rs:6-11)");
        };
        cfg->color = 0;
        if (c.has("rgb_colorspace") && space(c.at("rgb_colorspace").as_string())) cfg->color |= AKR_COLOR_RGB_ACESCG;
        if (c.has("color_repr")) {
            // ColorRepr is an internally tagged enum around a bare RgbColorSpace (color.rs:78-86), a shape serde cannot read
            // back; accepted here: {"type": "rgb", "colorspace": "aces"} and the ToString form "rgb_aces" (color.rs:87-93)
            const JsonValue& r = c.at("color_repr");
            std::string ty = (r.type == JsonValue::String) ? r.as_string() : r.at("type").as_string();
            if (ty == "spectral") throw std::runtime_error("unsupported: color_repr 'spectral' (todo!() in the reference as well)");
            bool aces = false;
            if ((r.type == JsonValue::String)) {
                if (ty.rfind("rgb_", 0) != 0) throw std::runtime_error("unknown color_repr '" + ty + "'");
                aces = space(ty.substr(4));
            } else {
                if (ty != "rgb") throw std::runtime_error("unknown color_repr type '" + ty + "'");
                if (r.has("colorspace")) aces = space(r.at("colorspace").as_string());
            }
            if (aces) cfg->color |= AKR_COLOR_REPR_ACESCG;
        }
    }
    if (j->has("film")) {
        const JsonValue& f = j->at("film");
        if (f.has("color")) {  // FilmColorRepr (film.rs:12-19)
            const JsonValue& fc = f.at("color");
            const std::string repr = fc.type == JsonValue::String ? fc.as_string() : std::string("spectral");
            if (repr != "srgb") throw std::runtime_error("unsupported: film colour representation '" + repr + "' (only \"srgb\")");
        }
        if (f.has("filter")) {
            const JsonValue& fl = f.at("filter");
            const std::string& ty = fl.at("type").as_string();
            if (ty == "box") cfg->filter_type = AKR_FILTER_BOX;
            else if (ty == "gaussian") cfg->filter_type = AKR_FILTER_GAUSSIAN;
            else throw std::runtime_error("unknown pixel filter '" + ty + "'");
            cfg->filter_radius = fl.at("radius").as_f32();
        }
        if (film_out && f.has("out")) *film_out = f.at("out").as_string();
    }
    if (task) {  // sampler and film filter are per RenderConfig, whatever the method
        task->aov.color = task->gpt.color = task->mcmc.color = cfg->color;  // RenderConfig.color goes to whichever integrator runs (lib.rs:43,98)
        task->aov.filter_type = cfg->filter_type;
        task->aov.filter_radius = cfg->filter_radius;
        task->aov.sampler_type = cfg->sampler_type;
        task->aov.sampler_seed = cfg->sampler_seed;
        task->gpt.filter_type = cfg->filter_type;
        task->gpt.filter_radius = cfg->filter_radius;
        task->gpt.sampler_type = cfg->sampler_type;
        task->gpt.sampler_seed = cfg->sampler_seed;
        task->mcmc.filter_type = cfg->filter_type;
        task->mcmc.filter_radius = cfg->filter_radius;
        task->mcmc.sampler_type = cfg->sampler_type;
        task->mcmc.sampler_seed = cfg->sampler_seed;
    }
}

std::vector<ParsedTask> parse_render_tasks(const std::string& text, bool allow_sampler_override) {
    JsonPtr root = JsonParser::parse(text);
    std::vector<ParsedTask> out;
    if (root->type == JsonValue::Array) {  // RenderTask::Multi
        for (const auto& t : root->arr) {
            ParsedTask p;
            parse_one_task(t.get(), &p.cfg, &p.film_out, allow_sampler_override, &p);
            out.push_back(p);
        }
    } else {
        ParsedTask p;
        parse_one_task(root.get(), &p.cfg, &p.film_out, allow_sampler_override, &p);
        out.push_back(p);
    }
    if (out.empty()) throw std::runtime_error("empty render task list");
    return out;
}

void parse_method_json(const std::string& text, akr_pt_config* cfg, std::string* film_out) {
    std::vector<ParsedTask> tasks = parse_render_tasks(text, false);
    if (tasks[0].is_mcmc) throw std::runtime_error("unsupported: method type 'mcmc_opt' here (akr_pt_config_from_json fills a pt::Config; use akr_render_task)");
    if (tasks[0].is_gpt) throw std::runtime_error("unsupported: method type 'gpt' here (akr_pt_config_from_json fills a pt::Config; use akr_render_task)");
    if (tasks[0].is_aov) throw std::runtime_error("unsupported: method type 'aov' here (akr_pt_config_from_json fills a pt::Config; use akr_render_task)");
    *cfg = tasks[0].cfg;
    if (film_out) *film_out = tasks[0].film_out;
}

}  // namespace akr
