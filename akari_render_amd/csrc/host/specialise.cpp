// specialise.cpp -- see specialise.h. Code generation for a scene's shader kinds, hiprtc compile, disk + process cache.
#include "specialise.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

extern char** environ;

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "embedded_src.inc"  // build/embedded_src.inc (build.py): kEmbeddedSources, kEmbeddedHash, kEmbeddedFlags

namespace akr {

const EmbeddedSource* embedded_sources(size_t* count) {
    *count = sizeof(kEmbeddedSources) / sizeof(kEmbeddedSources[0]);
    return kEmbeddedSources;
}
const char* embedded_sources_hash() { return kEmbeddedHash; }
const char* const* embedded_compile_flags(size_t* count) {
    *count = sizeof(kEmbeddedFlags) / sizeof(kEmbeddedFlags[0]);
    return kEmbeddedFlags;
}

// ------------------------------------------------------------------------------------------------ code generation
namespace {

std::string fmt(const char* f, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, f);
    std::vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}
std::string lit_f(float v) {  // the exact bits, whatever they are (NaN payloads, -0)
    uint32_t u;
    std::memcpy(&u, &v, 4);
    return fmt("u2f(0x%08xu)", u);
}
uint32_t n_args(uint32_t op) {
    switch (op) {
        case NODE_IMAGE: return 2;
        case NODE_MAPPING: return 3;
        case NODE_CHECKERBOARD: return 4;
        case NODE_SPECTRAL_UPLIFT: case NODE_SEPARATE_COLOR: case NODE_EXTRACT: return 1;
        case NODE_NORMAL_MAP: return 2;
        default: return 0;
    }
}
void refs_of(const akr_shader_node& n, uint32_t out[4], uint32_t& count) {
    count = 0;
    for (uint32_t a = 0; a < n_args(n.op); a++) {
        if (n.op == NODE_IMAGE && a == 0) continue;
        if (n.arg[a] != kNodeNone) out[count++] = n.arg[a];
    }
}

struct KindView {
    const CompiledScene& cs;
    const CompiledScene::ShaderKind& k;
    uint32_t index;
    // node i of material j of the kind
    const akr_shader_node& node(size_t j, uint32_t i) const { return cs.tex_nodes_ssa[cs.materials[k.materials[j]].tex_first_node + i]; }
    uint32_t feeds(uint32_t i) const { return cs.tex_nodes[cs.materials[k.materials[0]].tex_first_node + i].op >> 16; }
    bool uniform_k(uint32_t i, int c) const {
        for (size_t j = 1; j < k.materials.size(); j++)
            if (std::memcmp(&node(j, i).k[c], &node(0, i).k[c], 4) != 0) return false;
        return true;
    }
    bool uniform_arg(uint32_t i, int a) const {
        for (size_t j = 1; j < k.materials.size(); j++)
            if (node(j, i).arg[a] != node(0, i).arg[a]) return false;
        return true;
    }
    std::string kc(uint32_t i, int c) const { return uniform_k(i, c) ? lit_f(node(0, i).k[c]) : fmt("nd[%u].k[%d]", i, c); }
};

// the statements of the nodes `need` marks, in list order, and after each the inputs it feeds (when `with_feeds`)
std::string emit_nodes(const KindView& kv, const std::vector<uint8_t>& need, bool with_feeds) {
    std::string o;
    for (uint32_t i = 0; i < kv.k.n_nodes; i++) {
        if (!need[i]) continue;
        const akr_shader_node& nd = kv.node(0, i);
        auto v = [&](uint32_t a) { return fmt("v%u", nd.arg[a]); };
        std::string e;
        switch (nd.op) {
            case NODE_CONST: e = "node_const(" + kv.kc(i, 0) + ", " + kv.kc(i, 1) + ", " + kv.kc(i, 2) + ")"; break;
            case NODE_RGB: e = "node_rgb(ts.color, " + kv.kc(i, 0) + ", " + kv.kc(i, 1) + ", " + kv.kc(i, 2) + (nd.arg[0] == 1u ? ", true)" : ", false)"); break;
            case NODE_TEXCOORDS: e = "node_texcoords(uv)"; break;
            case NODE_IMAGE: {
                const DImage& im = kv.cs.images[nd.arg[0]];
                std::string img;
                if (kv.uniform_arg(i, 0)) {  // one image for the whole kind: the header is a literal
                    img = fmt("DImage{%uu, %uu, %uu, %uu, %uu, %uu, %uu, 0u}", im.offset_lo, im.offset_hi, im.width, im.height, im.format, im.filter, im.address);
                } else {  // format, filter and address mode are part of the kind's shape; the rest comes from the header
                    o += fmt("        DImage im%u = ts.images[nd[%u].arg[0]]; im%u.format = %uu; im%u.filter = %uu; im%u.address = %uu;\n", i, i, i, im.format, i, im.filter, i,
                             im.address);
                    img = fmt("im%u", i);
                }
                const std::string st = nd.arg[1] == kNodeNone ? "uv" : "mk2(" + v(1) + ".x, " + v(1) + ".y)";
                e = "node_image(ts.texels, " + img + ", " + st + (nd.arg[2] != 0 ? ", true)" : ", false)");
                break;
            }
            case NODE_MAPPING: e = "node_mapping(" + v(0) + ", " + v(1) + ", " + v(2) + fmt(", %uu)", nd.arg[3]); break;
            case NODE_CHECKERBOARD: {
                const std::string st = nd.arg[0] == kNodeNone ? "uv" : "mk2(" + v(0) + ".x, " + v(0) + ".y)";
                e = "node_checker_first(" + st + ", " + v(1) + ".x) ? " + v(2) + " : " + v(3);
                break;
            }
            case NODE_SPECTRAL_UPLIFT: e = "node_uplift(ts.color, " + v(0) + ")"; break;
            case NODE_SEPARATE_COLOR: e = v(0); break;
            case NODE_EXTRACT: e = "node_extract(" + v(0) + fmt(", %uu)", nd.arg[1]); break;
            case NODE_NORMAL_MAP: e = "node_normal_map(" + v(0) + ", " + v(1) + ".x)"; break;
            default: e = "tv(0, 0, 0, 0)"; break;
        }
        o += fmt("        const TexVal v%u = ", i) + e + ";\n";
        const uint32_t f = with_feeds ? kv.feeds(i) : 0u;
        if (!f) continue;
        auto v3 = [&](const char* name) { return fmt("        in.%s[0] = v%u.x; in.%s[1] = v%u.y; in.%s[2] = v%u.z;\n", name, i, name, i, name, i); };
        auto v1 = [&](const char* name) { return fmt("        in.%s = v%u.x;\n", name, i); };
        // principled.rs:13-131 read rules, as apply_fed (dtex.h)
        if (f & (1u << IN_BASE_COLOR)) o += v3("base_color") + fmt("        in.base_alpha = v%u.w;\n", i);
        if (f & (1u << IN_METALLIC)) o += v1("metallic");
        if (f & (1u << IN_ROUGHNESS)) o += v1("roughness");
        if (f & (1u << IN_IOR)) o += v1("ior");
        if (f & (1u << IN_SPECULAR_IOR_LEVEL)) o += v1("specular_ior_level");
        if (f & (1u << IN_SPECULAR_TINT)) o += v3("specular_tint");
        if (f & (1u << IN_TRANSMISSION_WEIGHT)) o += v1("transmission_weight");
        if (f & (1u << IN_COAT_WEIGHT)) o += v1("coat_weight");
        if (f & (1u << IN_COAT_ROUGHNESS)) o += v1("coat_roughness");
        if (f & (1u << IN_COAT_IOR)) o += v1("coat_ior");
        if (f & (1u << IN_COAT_TINT)) o += v3("coat_tint");
        if (f & (1u << IN_EMISSION_COLOR)) o += v3("emission_color");
        if (f & (1u << IN_EMISSION_STRENGTH)) o += v1("emission_strength");
        if (f & (1u << IN_NORMAL)) o += v3("normal");
    }
    return o;
}

// the nodes the inputs of `mask` depend on
std::vector<uint8_t> needed_for(const KindView& kv, uint32_t mask) {
    std::vector<uint8_t> need(kv.k.n_nodes, 0);
    for (uint32_t i = 0; i < kv.k.n_nodes; i++)
        if (kv.feeds(i) & mask) need[i] = 1;
    for (uint32_t i = kv.k.n_nodes; i-- > 0;) {
        if (!need[i]) continue;
        uint32_t refs[4], nr;
        refs_of(kv.node(0, i), refs, nr);
        for (uint32_t r = 0; r < nr; r++) need[refs[r]] = 1;
    }
    return need;
}
uint32_t node_feeding(const KindView& kv, uint32_t input) {
    for (uint32_t i = 0; i < kv.k.n_nodes; i++)
        if (kv.feeds(i) & (1u << input)) return i;
    return kNodeNone;
}

}  // namespace

std::string generate_scene_spec(const CompiledScene& cs) {
    if (!cs.has_textures || cs.shader_kinds.empty() || cs.shader_kinds.size() > kSpecMaxKinds) return std::string();
    std::string o;
    o += "// akr_scene_spec.h -- generated by host/specialise.cpp from the scene's shader graphs: one case per shader kind\n";
    o += "// (svm/compiler.rs:16-76), straight-line node code per case (svm/eval.rs:97-269, 428-467). Included by device/dtex.h.\n";
    o += "// (inside namespace akr)\n";
    o += fmt("constexpr uint32_t kSpecAbsent = 0x%xu;  // lobes no material of the scene can have (dbsdf.h AB_*): the kernel is compiled without them\n", cs.absent);
    std::string body_mat, body_alpha, body_emit;
    for (uint32_t ki = 0; ki < cs.shader_kinds.size(); ki++) {
        const KindView kv{cs, cs.shader_kinds[ki], ki};
        uint32_t fed = 0;
        for (uint32_t i = 0; i < kv.k.n_nodes; i++) fed |= kv.feeds(i);
        std::string mats;
        for (uint32_t m : kv.k.materials) mats += fmt(" %u", m);
        const std::string title = "    case " + std::to_string(ki) + "u: {  // " + kv.k.signature + " feeds " + fmt("0x%x", fed) + "; materials" + mats + "\n";
        // ---- spec_material_at
        body_mat += title;
        body_mat += "        MatInputs in = ts.mat_inputs[material];\n";
        body_mat += emit_nodes(kv, std::vector<uint8_t>(kv.k.n_nodes, 1), true);
        body_mat += fmt("        fold_inputs_fed(%uu, 0x%xu, in, m);\n", kv.k.mat_kind, fed);
        body_mat += "    } break;\n";
        // ---- spec_alpha: w of the node feeding base_color
        const uint32_t nb = node_feeding(kv, IN_BASE_COLOR);
        if (nb != kNodeNone) {
            body_alpha += title;
            body_alpha += emit_nodes(kv, needed_for(kv, 1u << IN_BASE_COLOR), false);
            body_alpha += fmt("        return v%u.w;\n    }\n", nb);
        }
        // ---- spec_emission: emission_color * emission_strength
        const uint32_t ne = node_feeding(kv, IN_EMISSION_COLOR), ns = node_feeding(kv, IN_EMISSION_STRENGTH);
        if (ne != kNodeNone || ns != kNodeNone) {
            body_emit += title;
            body_emit += emit_nodes(kv, needed_for(kv, (1u << IN_EMISSION_COLOR) | (1u << IN_EMISSION_STRENGTH)), false);
            const std::string col = ne != kNodeNone ? fmt("mk3(v%u.x, v%u.y, v%u.z)", ne, ne, ne) : std::string("mk3(in.emission_color[0], in.emission_color[1], in.emission_color[2])");
            const std::string str = ns != kNodeNone ? fmt("v%u.x", ns) : std::string("in.emission_strength");
            body_emit += "        return " + col + " * " + str + ";\n    }\n";
        }
    }
    o += "AKR_HD void spec_material_at(const TexScene& ts, uint32_t material, vec2 uv, DMaterial& m) {\n";
    o += "    const DNode* __restrict__ nd = ts.nodes + m.tex_first_node;\n    (void)nd;\n";
    o += "    switch (m.tex_n_nodes >> kTexKindShift) {\n" + body_mat + "    default: break;\n    }\n}\n";
    o += "AKR_HD float spec_alpha(const TexScene& ts, const DMaterial& m, uint32_t material, vec2 uv) {\n";
    o += "    const DNode* __restrict__ nd = ts.nodes + m.tex_first_node;\n    (void)nd; (void)material;\n";
    o += "    switch (m.tex_n_nodes >> kTexKindShift) {\n" + body_alpha + "    default: break;\n    }\n    return m.base_alpha;\n}\n";
    o += "AKR_HD vec3 spec_emission(const TexScene& ts, const DMaterial& m, uint32_t material, vec2 uv) {\n";
    o += "    const DNode* __restrict__ nd = ts.nodes + m.tex_first_node;\n    (void)nd;\n";
    o += "    const MatInputs& in = ts.mat_inputs[material];\n";
    o += "    switch (m.tex_n_nodes >> kTexKindShift) {\n" + body_emit + "    default: break;\n    }\n";
    o += "    return mk3(in.emission_color[0], in.emission_color[1], in.emission_color[2]) * in.emission_strength;\n}\n";
    return o;
}

// ------------------------------------------------------------------------------------------------ hiprtc (dlopen)
namespace {

struct Rtc {
    void* lib = nullptr;
    int (*CreateProgram)(void**, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    int (*CompileProgram)(void*, int, const char* const*) = nullptr;
    int (*GetProgramLogSize)(void*, size_t*) = nullptr;
    int (*GetProgramLog)(void*, char*) = nullptr;
    int (*GetCodeSize)(void*, size_t*) = nullptr;
    int (*GetCode)(void*, char*) = nullptr;
    int (*DestroyProgram)(void**) = nullptr;
    int (*Version)(int*, int*) = nullptr;
    bool ok = false;
    std::string why;
    Rtc() {
        const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so", "libhiprtc.so.6"};
        for (const char* n : names) {
            lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            why = "hiprtc not available (libhiprtc.so could not be loaded)";
            return;
        }
        auto sym = [&](const char* n) { return dlsym(lib, n); };
        CreateProgram = (decltype(CreateProgram))sym("hiprtcCreateProgram");
        CompileProgram = (decltype(CompileProgram))sym("hiprtcCompileProgram");
        GetProgramLogSize = (decltype(GetProgramLogSize))sym("hiprtcGetProgramLogSize");
        GetProgramLog = (decltype(GetProgramLog))sym("hiprtcGetProgramLog");
        GetCodeSize = (decltype(GetCodeSize))sym("hiprtcGetCodeSize");
        GetCode = (decltype(GetCode))sym("hiprtcGetCode");
        DestroyProgram = (decltype(DestroyProgram))sym("hiprtcDestroyProgram");
        Version = (decltype(Version))sym("hiprtcVersion");
        ok = CreateProgram && CompileProgram && GetProgramLogSize && GetProgramLog && GetCodeSize && GetCode && DestroyProgram;
        if (!ok) why = "hiprtc not available (libhiprtc.so lacks an entry point)";
    }
};
Rtc& rtc() {
    static Rtc r;
    return r;
}

std::string wrapper_source(const SpecRequest& rq) {
    return fmt("// per-scene instantiation of k_pt_pass (host/specialise.cpp)\n#define AKR_SPEC_GRAPHS 1\n#include \"device/pt_pass.h\"\n"
               "extern \"C\" __global__ __launch_bounds__(256, %d) void akr_pt_pass_spec(const akr::PtParams p) {\n"
               "    akr::pt_pass_body<%s, false, true, %s, %s, %s, akr::kSpecAbsent, %s>(p);\n}\n",
               rq.min_waves, (rq.bvh || rq.inst) ? "true" : "false", rq.pmj ? "true" : "false", (rq.stage && !rq.inst) ? "true" : "false",
               (rq.defer && !rq.inst) ? "true" : "false", rq.inst ? "true" : "false");
}
std::vector<std::string> compile_options(const std::string& arch) {
    std::vector<std::string> o;
    o.push_back("--offload-arch=" + arch);
    size_t n;
    const char* const* f = embedded_compile_flags(&n);
    for (size_t i = 0; i < n; i++) o.push_back(f[i]);
    // AKR_SPEC_EXTRA_FLAGS: A/B switches of the device code for measurements (-DAKR_...=...), part of the cache key. Only switches
    // that leave the launch's LDS layout alone are safe here: the host plans the layout from the library's own build (kernels.h).
    if (const char* e = std::getenv("AKR_SPEC_EXTRA_FLAGS")) {
        std::istringstream ss(e);
        std::string w;
        while (ss >> w) o.push_back(w);
    }
    return o;
}
uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) {
        h ^= b[i];
        h *= 0x100000001b3ull;
    }
    return h;
}
uint64_t fnv1a(uint64_t h, const std::string& s) {
    h = fnv1a(h, s.data(), s.size());
    const unsigned char sep = 0xff;
    return fnv1a(h, &sep, 1);
}
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

std::string spec_cache_key(const std::string& spec_header, const SpecRequest& rq, const std::string& arch) {
    uint64_t h = 0xcbf29ce484222325ull;
    h = fnv1a(h, std::string(embedded_sources_hash()));
    h = fnv1a(h, spec_header);
    h = fnv1a(h, wrapper_source(rq));
    for (const std::string& o : compile_options(arch)) h = fnv1a(h, o);
    int major = 0, minor = 0;
    if (rtc().ok && rtc().Version) (void)rtc().Version(&major, &minor);
    h = fnv1a(h, fmt("hiprtc %d.%d", major, minor));
    return fmt("%016llx", (unsigned long long)h);
}

std::string spec_cache_dir() {
    if (const char* e = std::getenv("AKR_KERNEL_CACHE")) return e;
    if (const char* e = std::getenv("XDG_CACHE_HOME"))
        if (*e) return std::string(e) + "/akari_hip";
    if (const char* e = std::getenv("HOME"))
        if (*e) return std::string(e) + "/.cache/akari_hip";
    return fmt("/tmp/akari_hip-%u", (unsigned)getuid());
}

namespace {
// The disk cache is trusted code: a code object found there is loaded and run. So the directory must be ours alone -- created (with its
// parents) mode 0700, and used only if it is a real directory (not a link) owned by this user that nobody else can write to. Otherwise
// the disk cache is skipped (kernels are compiled per process). Matters where the fall-back /tmp/akari_hip-<uid> is in play: another local
// user could have made that directory first and planted a file under a key that is computable from public inputs.
bool cache_dir_usable(const std::string& dir, bool create) {
    if (dir.empty()) return false;
    if (create) {
        for (size_t i = 1; i <= dir.size(); i++)
            if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0700);  // (existing components: EEXIST)
    }
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0) return false;
    return S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
// <directory of this library>/akari-cli, or "" (dladdr on a symbol of the library)
std::string helper_path() {
    Dl_info info;
    if (!dladdr((const void*)&helper_path, &info) || !info.dli_fname) return std::string();
    std::string p = info.dli_fname;
    const size_t slash = p.rfind('/');
    p = (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/akari-cli";
    return access(p.c_str(), X_OK) == 0 ? p : std::string();
}
bool compile_in_helper(const std::string& helper, const std::string& spec_header, const SpecRequest& rq, const std::string& arch, std::vector<char>& code, std::string& log) {
    char dir[] = "/tmp/akr_spec_XXXXXX";
    if (!mkdtemp(dir)) return false;
    const std::string hdr = std::string(dir) + "/akr_scene_spec.h", out = std::string(dir) + "/k.co", err = std::string(dir) + "/err.txt";
    bool ok = false;
    {
        std::ofstream f(hdr, std::ios::binary);
        f.write(spec_header.data(), (std::streamsize)spec_header.size());
    }
    const std::string flags = std::to_string((rq.bvh ? 1 : 0) | (rq.pmj ? 2 : 0) | (rq.stage ? 4 : 0) | (rq.defer ? 8 : 0) | (rq.inst ? 16 : 0)), waves = std::to_string(rq.min_waves);
    // posix_spawn, not fork + setup code: the host process has threads (HIP runtime, the application's own)
    std::vector<std::string> env_store;
    for (char** e = environ; e && *e; e++)
        if (std::strncmp(*e, "AKR_SPEC_INPROCESS=", 19) != 0) env_store.emplace_back(*e);
    env_store.emplace_back("AKR_SPEC_INPROCESS=1");  // the helper compiles in its own process, it does not spawn another
    std::vector<char*> envp;
    for (std::string& e : env_store) envp.push_back(&e[0]);
    envp.push_back(nullptr);
    const char* argv[] = {helper.c_str(), "--spec-compile", hdr.c_str(), out.c_str(), arch.c_str(), flags.c_str(), waves.c_str(), nullptr};
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 2, err.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    pid_t pid = -1;
    if (posix_spawn(&pid, helper.c_str(), &fa, nullptr, const_cast<char* const*>(argv), envp.data()) != 0) pid = -1;
    posix_spawn_file_actions_destroy(&fa);
    if (pid > 0) {
        int status = 0;
        if (waitpid(pid, &status, 0) == pid && WIFEXITED(status) && WEXITSTATUS(status) == 0) {
            std::ifstream f(out, std::ios::binary);
            code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
            ok = !code.empty();
        } else {
            std::ifstream f(err);
            std::stringstream ss;
            ss << f.rdbuf();
            log = "akari-cli --spec-compile failed: " + ss.str().substr(0, 1500);
        }
    }
    (void)std::remove(hdr.c_str()); (void)std::remove(out.c_str()); (void)std::remove(err.c_str()); (void)rmdir(dir);
    return ok;
}
}  // namespace

bool spec_compile(const std::string& spec_header, const SpecRequest& rq, const std::string& arch, std::vector<char>& code, std::string& log, bool in_process) {
    if (!in_process && !std::getenv("AKR_SPEC_INPROCESS")) {
        const std::string helper = helper_path();
        if (!helper.empty() && compile_in_helper(helper, spec_header, rq, arch, code, log)) return true;
        code.clear();  // no helper, or it failed: this process's hiprtc
    }
    Rtc& r = rtc();
    if (!r.ok) {
        log = r.why;
        return false;
    }
    size_t n_src;
    const EmbeddedSource* src = embedded_sources(&n_src);
    std::vector<const char*> texts, names;
    for (size_t i = 0; i < n_src; i++) {
        texts.push_back(src[i].text);
        names.push_back(src[i].name);
    }
    texts.push_back(spec_header.c_str());
    names.push_back("akr_scene_spec.h");
    const std::string wrapper = wrapper_source(rq);
    void* prog = nullptr;
    int rc = r.CreateProgram(&prog, wrapper.c_str(), "akr_pt_pass_spec.hip", (int)texts.size(), texts.data(), names.data());
    if (rc != 0) {
        log = fmt("hiprtcCreateProgram failed (%d)", rc);
        return false;
    }
    const std::vector<std::string> opts = compile_options(arch);
    std::vector<const char*> optv;
    for (const std::string& s : opts) optv.push_back(s.c_str());
    rc = r.CompileProgram(prog, (int)optv.size(), optv.data());
    size_t ls = 0;
    if (r.GetProgramLogSize(prog, &ls) == 0 && ls > 1) {
        log.resize(ls);
        (void)r.GetProgramLog(prog, &log[0]);
    }
    bool ok = rc == 0;
    if (ok) {
        size_t cs = 0;
        ok = r.GetCodeSize(prog, &cs) == 0 && cs > 0;
        if (ok) {
            code.resize(cs);
            ok = r.GetCode(prog, code.data()) == 0;
        }
    } else {
        log = fmt("hiprtcCompileProgram failed (%d): ", rc) + log;
    }
    (void)r.DestroyProgram(&prog);
    return ok;
}

SpecKernel::~SpecKernel() {
    if (module) (void)hipModuleUnload(module);
}

std::shared_ptr<SpecKernel> SpecCache::get(const std::string& spec_header, const SpecRequest& rq, const std::string& arch, bool may_compile) {
    std::lock_guard<std::mutex> lock(mutex_);
    const std::string key = spec_cache_key(spec_header, rq, arch);
    auto it = loaded_.find(key);
    if (it != loaded_.end()) {
        // a second session of this process on the same kernel: nothing to compile or load
        auto again = std::make_shared<SpecKernel>();
        again->fn = it->second->fn;
        again->cache_hit = true;
        again->vgprs = it->second->vgprs;
        again->scratch_bytes = it->second->scratch_bytes;
        again->status = "ok";
        again->module = nullptr;  // the map owns the module (it lives as long as the context)
        return again;
    }
    auto k = std::make_shared<SpecKernel>();
    if (spec_header.empty()) {
        k->status = "no per-scene code: the scene has no texture-fed material or too many shader kinds";
        return k;
    }
    const double t0 = now_ms();
    std::vector<char> code;
    const std::string dir = spec_cache_dir(), path = dir + "/akr_" + key + ".co";
    if (cache_dir_usable(dir, false)) {
        std::ifstream f(path, std::ios::binary);
        if (f) {
            code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
            k->cache_hit = !code.empty();
        }
    }
    if (code.empty() && !may_compile) {
        k->status = "no cached kernel, and the render is below the automatic threshold (option specialise = -1)";
        return k;
    }
    if (code.empty()) {
        std::string log;
        const double c0 = now_ms();
        if (!spec_compile(spec_header, rq, arch, code, log)) {
            k->status = log.substr(0, 2000);
            return k;
        }
        k->compile_ms = now_ms() - c0;
        // keep it for the next process: temporary file + rename, so that a reader never sees half a code object
        const std::string tmp = path + fmt(".%d.tmp", (int)getpid());
        std::ofstream f;
        if (cache_dir_usable(dir, true)) f.open(tmp, std::ios::binary);
        if (f.is_open()) {
            f.write(code.data(), (std::streamsize)code.size());
            f.close();
            if (!f || std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str());
        }
    }
    hipError_t e = hipModuleLoadData(&k->module, code.data());
    if (e != hipSuccess) (void)hipGetLastError();  // (the runtime remembers the failure: the next launch's hipGetLastError would report it)
    if (e != hipSuccess && k->cache_hit) {  // a stale or damaged file: compile again
        (void)std::remove(path.c_str());
        k->cache_hit = false;
        std::string log;
        const double c0 = now_ms();
        code.clear();
        if (!spec_compile(spec_header, rq, arch, code, log)) {
            k->status = log.substr(0, 2000);
            return k;
        }
        k->compile_ms = now_ms() - c0;
        e = hipModuleLoadData(&k->module, code.data());
        if (e != hipSuccess) (void)hipGetLastError();
    }
    if (e != hipSuccess) {
        k->module = nullptr;
        k->status = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
        return k;
    }
    e = hipModuleGetFunction(&k->fn, k->module, "akr_pt_pass_spec");
    if (e != hipSuccess) {
        (void)hipGetLastError();
        k->fn = nullptr;
        k->status = std::string("hipModuleGetFunction: ") + hipGetErrorString(e);
        return k;
    }
    (void)hipFuncGetAttribute(&k->vgprs, HIP_FUNC_ATTRIBUTE_NUM_REGS, k->fn);
    (void)hipFuncGetAttribute(&k->scratch_bytes, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, k->fn);
    k->load_ms = now_ms() - t0 - k->compile_ms;
    k->status = "ok";
    loaded_[key] = k;
    return k;
}

}  // namespace akr
