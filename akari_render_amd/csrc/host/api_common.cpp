// api_common.cpp -- errors, options, contexts, films (C ABI of libakari_hip.so, include/akari_hip.h; shared internals: api_internal.h)
#include "api_internal.h"

thread_local std::string akr_api::g_last_error;

namespace akr {
// for host/comm.cpp (the RCCL film reduce): what it needs to know about a film, and the shared error slot
int32_t film_device_view(akr_film* film, int* device, hipStream_t* stream, float** data, size_t* n_floats) {
    if (!film || !film->ctx) return fail(AKR_ERR_INVALID_ARGUMENT, "film is NULL");
    *device = film->ctx->device;
    *stream = film->ctx->stream;
    *data = film->data;
    *n_floats = film->n_floats();
    return AKR_OK;
}
int32_t api_fail(int32_t code, const std::string& msg) { return fail(code, msg); }
}  // namespace akr

extern "C" {

AKR_API const char* akr_last_error(void) { return g_last_error.c_str(); }
AKR_API uint32_t akr_struct_size(int32_t which) {
    switch (which) {
        case AKR_STRUCT_MESH_DESC: return sizeof(akr_mesh_desc);
        case AKR_STRUCT_INSTANCE_DESC: return sizeof(akr_instance_desc);
        case AKR_STRUCT_MATERIAL_DESC: return sizeof(akr_material_desc);
        case AKR_STRUCT_CAMERA_DESC: return sizeof(akr_camera_desc);
        case AKR_STRUCT_SCENE_DESC: return sizeof(akr_scene_desc);
        case AKR_STRUCT_PT_CONFIG: return sizeof(akr_pt_config);
        case AKR_STRUCT_PT_STATS: return sizeof(akr_pt_stats);
        case AKR_STRUCT_SCENE_INFO: return sizeof(akr_scene_info);
        case AKR_STRUCT_KERNEL_INFO: return sizeof(akr_kernel_info);
        case AKR_STRUCT_AOV_CONFIG: return sizeof(akr_aov_config);
        case AKR_STRUCT_GPT_CONFIG: return sizeof(akr_gpt_config);
        case AKR_STRUCT_MCMC_CONFIG: return sizeof(akr_mcmc_config);
        case AKR_STRUCT_MCMC_RESULT: return sizeof(akr_mcmc_result);
        case AKR_STRUCT_MCMC_PARTIAL: return sizeof(akr_mcmc_partial);
        default: return 0;
    }
}
AKR_API const char* akr_version(void) { return "akari_hip 0.3.0 gfx950"; }  // 0.3.0: akr_struct_size, scenes kept as meshes + instances; 0.2.0: akr_pt_config gained sample_begin / sample_count (88 bytes); akr_kernel_info carries its own size
AKR_API int32_t akr_option_set(const char* name, int32_t value) {
    if (!tuning_set(name, value)) return fail(AKR_ERR_INVALID_ARGUMENT, std::string("akr_option_set: unknown option '") + (name ? name : "(null)") + "' or value out of range");
    return AKR_OK;
}
AKR_API int32_t akr_option_get(const char* name, int32_t* value) {
    int v = 0;
    if (!value || !tuning_get(name, &v)) return fail(AKR_ERR_INVALID_ARGUMENT, std::string("akr_option_get: unknown option '") + (name ? name : "(null)") + "'");
    *value = v;
    return AKR_OK;
}

AKR_API int32_t akr_context_create(int32_t device, akr_context** out) {
    if (!out) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_context_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(AKR_ERR_NO_DEVICE, std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "0 devices") +
                                           "); libakari_hip has no CPU path");
    if (device < 0 || device >= count) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_context_create: device ordinal out of range");
    return guarded([&] {
        auto ctx = std::make_unique<akr_context>();
        ctx->device = device;
        ctx->bind();
        HIP_CHECK(hipGetDeviceProperties(&ctx->props, device));
        HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        *out = ctx.release();
    });
}
AKR_API int32_t akr_context_destroy(akr_context* ctx) {
    if (!ctx) return AKR_OK;
    return guarded([&] {
        (void)hipSetDevice(ctx->device);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
    });
}
AKR_API int32_t akr_context_synchronize(akr_context* ctx) {
    if (!ctx) return fail(AKR_ERR_INVALID_ARGUMENT, "context is NULL");
    return guarded([&] {
        ctx->bind();
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}
AKR_API int32_t akr_context_device_ordinal(akr_context* ctx, int32_t* device) {
    if (!ctx || !device) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_context_device_ordinal: NULL argument");
    *device = ctx->device;
    return AKR_OK;
}
AKR_API int32_t akr_context_device_info(akr_context* ctx, char* name, uint32_t name_len, uint32_t* compute_units, uint64_t* hbm_bytes) {
    if (!ctx) return fail(AKR_ERR_INVALID_ARGUMENT, "context is NULL");
    if (name && name_len) {
        std::snprintf(name, name_len, "%s (%s)", ctx->props.name, ctx->props.gcnArchName);
    }
    if (compute_units) *compute_units = (uint32_t)ctx->props.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)ctx->props.totalGlobalMem;
    return AKR_OK;
}
AKR_API int32_t akr_film_create(akr_context* ctx, uint32_t width, uint32_t height, akr_film** out) {
    if (!ctx || !out || !width || !height) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_create: bad argument");
    *out = nullptr;
    return guarded([&] {
        ctx->bind();
        auto f = std::make_unique<akr_film>();
        f->ctx = ctx;
        f->width = width;
        f->height = height;
        f->own.alloc(f->n_floats() * sizeof(float));
        f->data = f->own.as<float>();
        HIP_CHECK(hipMemsetAsync(f->data, 0, f->n_floats() * sizeof(float), ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *out = f.release();
    });
}
AKR_API int32_t akr_film_wrap(akr_context* ctx, uint32_t width, uint32_t height, void* device_ptr, akr_film** out) {
    if (!ctx || !out || !width || !height || !device_ptr) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_wrap: bad argument");
    *out = nullptr;
    {   // The gpt / mcmc_opt kernels splat with hardware float atomics (global_atomic_add_f32), which CDNA silently drops on
        // host-mapped, managed or fine-grained memory: only plain device allocations (hipMalloc) of this context's GPU pass.
        hipPointerAttribute_t attr;
        hipError_t e = hipPointerGetAttributes(&attr, device_ptr);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_wrap: device_ptr is not a HIP allocation");
        }
        if (attr.type != hipMemoryTypeDevice || attr.isManaged)
            return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_wrap: device_ptr must be plain device memory (hipMalloc), not host-mapped or managed memory");
        if (attr.device != ctx->device) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_wrap: device_ptr belongs to another GPU than the context");
    }
    auto* f = new (std::nothrow) akr_film();
    if (!f) return fail(AKR_ERR_OUT_OF_MEMORY, "out of host memory");
    f->ctx = ctx;
    f->width = width;
    f->height = height;
    f->data = (float*)device_ptr;
    *out = f;
    return AKR_OK;
}
AKR_API int32_t akr_film_destroy(akr_film* film) {
    if (!film) return AKR_OK;
    return guarded([&] {
        (void)hipSetDevice(film->ctx->device);
        delete film;
    });
}
AKR_API int32_t akr_film_clear(akr_film* f) {
    if (!f) return fail(AKR_ERR_INVALID_ARGUMENT, "film is NULL");
    return guarded([&] {
        f->ctx->bind();
        HIP_CHECK(hipMemsetAsync(f->data, 0, f->n_floats() * sizeof(float), f->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    });
}
AKR_API int32_t akr_film_read(akr_film* f, float* dst) {
    if (!f || !dst) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_read: NULL argument");
    return guarded([&] {
        f->ctx->bind();
        HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
        HIP_CHECK(hipMemcpy(dst, f->data, f->n_floats() * sizeof(float), hipMemcpyDeviceToHost));
    });
}
AKR_API int32_t akr_film_write(akr_film* f, const float* src) {
    if (!f || !src) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_write: NULL argument");
    return guarded([&] {
        f->ctx->bind();
        HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
        HIP_CHECK(hipMemcpy(f->data, src, f->n_floats() * sizeof(float), hipMemcpyHostToDevice));
    });
}
AKR_API int32_t akr_film_resolve(akr_film* f, float* dst_rgb) {
    if (!f || !dst_rgb) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_resolve: NULL argument");
    return guarded([&] {
        f->ctx->bind();
        uint64_t n = (uint64_t)f->width * f->height;
        DevBuf tmp;
        tmp.alloc(3 * n * sizeof(float));
        HIP_CHECK(launch_film_resolve(f->data, n, f->splat_scale, tmp.as<float>(), f->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
        HIP_CHECK(hipMemcpy(dst_rgb, tmp.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    });
}
AKR_API int32_t akr_film_set_splat_scale(akr_film* f, float scale) {  // film.rs:152-154
    if (!f) return fail(AKR_ERR_INVALID_ARGUMENT, "film is NULL");
    f->splat_scale = scale;
    return AKR_OK;
}
AKR_API int32_t akr_film_get_splat_scale(const akr_film* f, float* scale) {
    if (!f || !scale) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_get_splat_scale: NULL argument");
    *scale = f->splat_scale;
    return AKR_OK;
}
AKR_API int32_t akr_film_device_ptr(akr_film* f, void** ptr, uint64_t* bytes) {
    if (!f) return fail(AKR_ERR_INVALID_ARGUMENT, "film is NULL");
    if (ptr) *ptr = f->data;
    if (bytes) *bytes = f->n_floats() * sizeof(float);
    return AKR_OK;
}

// ------------------------------------------------------------------------------------------------ render driver
AKR_API int32_t akr_image_write(const char* path, const float* rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb || !width || !height) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_image_write: bad argument");
    return guarded([&] { write_image(path, rgb, width, height); });
}

}  // extern "C"
