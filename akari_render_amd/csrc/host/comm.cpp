// comm.cpp -- the path's one exchange step as part of the C ABI: sum-reduce of the per-GPU films over RCCL (xGMI).
//
// The reference has no multi-device code. This build shards a frame by pixel tiles over one process per GPU
// (akr_pt_config.shard_*); every rank's film is zero outside its tiles, so ONE ncclReduce(sum) of 7 * W * H floats onto the
// root (or ncclAllReduce) assembles the frame exactly: each element receives one non-zero addend. A host in any language binds
// these entry points like the rest of include/akari_hip.h; librccl is loaded on first use (dlopen), so single-GPU users of
// the library do not depend on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>

#include "../../../include/akari_hip.h"

namespace akr {
// defined in api.cpp
int32_t film_device_view(akr_film* film, int* device, hipStream_t* stream, float** data, size_t* n_floats);
int32_t api_fail(int32_t code, const std::string& msg);
int32_t gpt_reduce_view(akr_gpt_session* se, akr_film** film, int* device, hipStream_t* stream, float** sums, size_t* n_sums);
}  // namespace akr
using namespace akr;

namespace {
// the handful of RCCL symbols this file needs (rccl.h: ncclUniqueId is 128 opaque bytes, ncclFloat32 = 7, ncclSum = 0)
struct UniqueId { char internal[128]; };
typedef void* Comm;
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, Comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // An RCCL the process already holds wins (PyTorch ships its own librccl.so: two copies of RCCL in one process corrupt
        // each other's state at exit); only if there is none does the ROCm installation's library get loaded.
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (r.handle) break;
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.handle) break;
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.handle) {
            r.error = std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "?");
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(r.handle, n);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.Reduce = (decltype(r.Reduce))sym("ncclReduce");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}
int32_t rccl_fail(const char* what, int rc) {
    Rccl& r = rccl();
    return api_fail(AKR_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}
}  // namespace

struct akr_comm {
    Comm comm = nullptr;
    int rank = 0, world = 1, device = 0;
    bool owned = true;
};

extern "C" {

AKR_API int32_t akr_device_count(int32_t* count) {
    if (!count) return api_fail(AKR_ERR_INVALID_ARGUMENT, "akr_device_count: NULL argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return AKR_OK;
}

AKR_API int32_t akr_comm_unique_id(uint8_t id[AKR_COMM_ID_BYTES]) {
    if (!id) return api_fail(AKR_ERR_INVALID_ARGUMENT, "akr_comm_unique_id: NULL argument");
    Rccl& r = rccl();
    if (!r.error.empty()) return api_fail(AKR_ERR_UNSUPPORTED, r.error);
    UniqueId u;
    int rc = r.GetUniqueId(&u);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    static_assert(sizeof(UniqueId) == AKR_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(id, u.internal, sizeof u.internal);
    return AKR_OK;
}

AKR_API int32_t akr_comm_create(akr_context* ctx, const uint8_t id[AKR_COMM_ID_BYTES], int32_t rank, int32_t world, akr_comm** out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return api_fail(AKR_ERR_INVALID_ARGUMENT, "akr_comm_create: bad argument");
    *out = nullptr;
    Rccl& r = rccl();
    if (!r.error.empty()) return api_fail(AKR_ERR_UNSUPPORTED, r.error);
    int device = 0;
    if (akr_context_device_ordinal(ctx, &device) != AKR_OK) return AKR_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return api_fail(AKR_ERR_HIP, "hipSetDevice failed");
    UniqueId u;
    std::memcpy(u.internal, id, sizeof u.internal);
    auto* c = new akr_comm();
    c->rank = rank; c->world = world; c->device = device;
    int rc = r.CommInitRank(&c->comm, world, u, rank);
    if (rc != 0) {
        delete c;
        return rccl_fail("ncclCommInitRank", rc);
    }
    *out = c;
    return AKR_OK;
}

AKR_API int32_t akr_comm_wrap(akr_context* ctx, void* nccl_comm, int32_t rank, int32_t world, akr_comm** out) {
    if (!ctx || !nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return api_fail(AKR_ERR_INVALID_ARGUMENT, "akr_comm_wrap: bad argument");
    Rccl& r = rccl();
    if (!r.error.empty()) return api_fail(AKR_ERR_UNSUPPORTED, r.error);
    auto* c = new akr_comm();
    c->comm = nccl_comm; c->rank = rank; c->world = world; c->owned = false;
    (void)akr_context_device_ordinal(ctx, &c->device);
    *out = c;
    return AKR_OK;
}

AKR_API int32_t akr_comm_destroy(akr_comm* comm) {
    if (!comm) return AKR_OK;
    if (comm->owned && comm->comm) {
        (void)hipSetDevice(comm->device);
        (void)rccl().CommDestroy(comm->comm);
    }
    delete comm;
    return AKR_OK;
}

// planes: which parts of the accumulator [rgb 3N | splat 3N | weight N] carry anything to sum. A `pt` / `aov` film never touches
// its splat plane (film.rs:196-229 add_sample: rgb + weight), so its exchange is the 4 N floats SURVEY.md 8(e) names -- rgb and
// weight as two collectives of one RCCL group -- instead of all 7 N; gpt / mcmc_opt films (splats) need all of it. Every rank of
// the communicator must pass the same mask.
AKR_API int32_t akr_film_reduce_planes(akr_film* film, akr_comm* comm, int32_t root, int32_t blocking, uint32_t planes) {
    if (!film || !comm || root >= comm->world) return api_fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_reduce: bad argument");
    if (planes == 0 || (planes & ~(uint32_t)AKR_FILM_PLANES_ALL)) return api_fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_reduce_planes: planes must be a non-empty subset of AKR_FILM_PLANES_ALL");
    int device = 0;
    hipStream_t stream = nullptr;
    float* data = nullptr;
    size_t n = 0;
    int32_t rc0 = film_device_view(film, &device, &stream, &data, &n);
    if (rc0 != AKR_OK) return rc0;
    if (hipSetDevice(device) != hipSuccess) return api_fail(AKR_ERR_HIP, "hipSetDevice failed");
    Rccl& r = rccl();
    const size_t N = n / 7;
    // contiguous runs of the selected planes: [0, 3N) rgb, [3N, 6N) splat, [6N, 7N) weight
    size_t first[3], count[3];
    int runs = 0;
    const size_t lo[3] = {0, 3 * N, 6 * N}, len[3] = {3 * N, 3 * N, N};
    for (int k = 0; k < 3; k++) {
        if (!(planes & (1u << k))) continue;
        if (runs > 0 && first[runs - 1] + count[runs - 1] == lo[k]) count[runs - 1] += len[k];
        else { first[runs] = lo[k]; count[runs] = len[k]; runs++; }
    }
    // in place, on the context's own stream: ordered after the render that filled the film, no extra synchronisation
    auto one = [&](size_t off, size_t cnt) {
        return root < 0 ? r.AllReduce(data + off, data + off, cnt, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm->comm, stream)
                        : r.Reduce(data + off, data + off, cnt, 7, 0, root, comm->comm, stream);
    };
    int rc = 0;
    if (runs == 1) {
        rc = one(first[0], count[0]);
    } else {
        rc = r.GroupStart();
        for (int k = 0; k < runs && rc == 0; k++) rc = one(first[k], count[k]);
        const int rc_end = r.GroupEnd();
        if (rc == 0) rc = rc_end;
    }
    if (rc != 0) return rccl_fail(root < 0 ? "ncclAllReduce" : "ncclReduce", rc);
    if (blocking) {
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return api_fail(AKR_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
    return AKR_OK;
}
AKR_API int32_t akr_film_reduce(akr_film* film, akr_comm* comm, int32_t root, int32_t blocking) {
    return akr_film_reduce_planes(film, comm, root, blocking, AKR_FILM_PLANES_ALL);
}

// The exchange step of a sharded mcmc_opt render: the films (direct lighting of disjoint tiles + every rank's splats) are summed, the
// normalisation sums of all ranks meet in one small all-reduce, and akr_mcmc_combine_host's arithmetic sets the film's splat scale.
AKR_API int32_t akr_mcmc_combine(akr_film* film, akr_comm* comm, int32_t root, const akr_mcmc_partial* mine, akr_mcmc_result* result) {
    if (!film || !comm || !mine || root >= comm->world) return api_fail(AKR_ERR_INVALID_ARGUMENT, "akr_mcmc_combine: bad argument");
    int32_t rc0 = akr_film_reduce_planes(film, comm, root, 0, AKR_FILM_PLANES_ALL);
    if (rc0 != AKR_OK) return rc0;
    int device = 0;
    hipStream_t stream = nullptr;
    float* data = nullptr;
    size_t n = 0;
    rc0 = film_device_view(film, &device, &stream, &data, &n);
    if (rc0 != AKR_OK) return rc0;
    double h[5] = {mine->b_sum, (double)mine->b_cnt, (double)mine->n_accepted, (double)mine->n_mutations, (double)mine->n_executed};  // counts < 2^53: exact
    double* d = nullptr;
    if (hipMalloc((void**)&d, sizeof h) != hipSuccess) return api_fail(AKR_ERR_HIP, "akr_mcmc_combine: hipMalloc failed");
    hipError_t e = hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, stream);
    int rc = e == hipSuccess ? rccl().AllReduce(d, d, 5, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm->comm, stream) : 0;
    if (e == hipSuccess && rc == 0) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && rc == 0) e = hipStreamSynchronize(stream);
    (void)hipFree(d);
    if (rc != 0) return rccl_fail("ncclAllReduce", rc);
    if (e != hipSuccess) return api_fail(AKR_ERR_HIP, std::string("akr_mcmc_combine: ") + hipGetErrorString(e));
    akr_mcmc_partial all = *mine;  // one pseudo-rank that carries everybody's sums
    all.b_sum = h[0]; all.b_cnt = (uint64_t)h[1]; all.n_accepted = (uint64_t)h[2]; all.n_mutations = (uint64_t)h[3]; all.n_executed = (uint64_t)h[4];
    return akr_mcmc_combine_host(film, &all, 1, result);
}

// The exchange step of a sharded gpt render: with reconstruction none the ranks' films (splat channels of disjoint tiles) are
// summed, otherwise the primal / gradient sums, onto `root` (or every rank, root = -1); akr_gpt_finish then reconstructs there.
AKR_API int32_t akr_gpt_reduce(akr_gpt_session* se, akr_comm* comm, int32_t root, int32_t blocking) {
    if (!se || !comm || root >= comm->world) return api_fail(AKR_ERR_INVALID_ARGUMENT, "akr_gpt_reduce: bad argument");
    akr_film* film = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    float* sums = nullptr;
    size_t n = 0;
    int32_t rc0 = gpt_reduce_view(se, &film, &device, &stream, &sums, &n);
    if (rc0 != AKR_OK) return rc0;
    if (n == 0) return akr_film_reduce(film, comm, root, blocking);
    if (hipSetDevice(device) != hipSuccess) return api_fail(AKR_ERR_HIP, "hipSetDevice failed");
    Rccl& r = rccl();
    int rc = root < 0 ? r.AllReduce(sums, sums, n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm->comm, stream) : r.Reduce(sums, sums, n, 7, 0, root, comm->comm, stream);
    if (rc != 0) return rccl_fail(root < 0 ? "ncclAllReduce" : "ncclReduce", rc);
    if (blocking) {
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return api_fail(AKR_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
    return AKR_OK;
}

}  // extern "C"
