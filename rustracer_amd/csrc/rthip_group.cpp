// rthip_group.cpp — single-process multi-GPU form of the render path: one context and one HIP stream per
// device, the image split into INTERLEAVED row bands (band b = rows [b*R, (b+1)*R), owned by device b mod ndev: sky rows cost
// one segment per path, rows through the glass sphere up to max_depth), the scene replicated, and the only exchange one
// gather of every device's RGBA8 (or running-sum) rows to device 0 over RCCL — grouped ncclSend/ncclRecv, one transfer per
// xGMI link, no ring — followed by a de-interleave kernel and one device->host copy. The reference's caller is a single
// render thread (standalone/src/main.rs:65,99-102): it gets ONE call that returns the whole frame.
// RCCL is loaded at run time (dlopen) by the first group that needs it, so single-GPU users of librthip.so do not depend on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rthip.h"

extern "C" int rt_ctx_render_deferred(rthip_ctx *ctx, const rthip_render_params *p, void *d_rgba8, void *d_accum, rthip_stats *stats_wanted);
extern "C" int rt_ctx_collect_stats(rthip_ctx *ctx, rthip_stats *stats);
extern "C" void *rt_ctx_stream(rthip_ctx *ctx);
extern "C" uint64_t rt_ctx_nb_samples(rthip_ctx *ctx);
extern "C" hipError_t rt_launch_deinterleave(void *full, const void *recv, const void *own, uint32_t units, uint32_t height, uint32_t band_rows,
                                             uint32_t ndev, uint32_t pad_rows, hipStream_t stream);

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::mutex mutex;
    std::string load() {  // "" = ok; safe to call from several threads (the first group of each may race here)
        std::lock_guard<std::mutex> lock(mutex);
        if (handle) return "";
        void *h = nullptr;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return std::string("cannot load RCCL: ") + dlerror();
        auto sym = [&](const char *n) { return dlsym(h, n); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !Send || !Recv || !GroupStart || !GroupEnd || !GetErrorString) {
            dlclose(h);
            return "RCCL library lacks an expected symbol";
        }
        handle = h;  // published last: a non-null handle means every pointer above is set
        return "";
    }
};
Rccl g_rccl;

struct Buf {
    void *p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct rthip_group {
    std::vector<int> devices;
    std::vector<rthip_ctx *> ctx;
    std::vector<ncclComm_t> comm;  // empty: peer copies
    std::vector<Buf> band_rgba, band_accum;
    std::vector<hipEvent_t> rendered;  // per device: its pass has been enqueued up to here
    Buf recv_rgba, recv_accum, full_rgba, full_accum;  // on devices[0]
    hipEvent_t g0 = nullptr, g1 = nullptr;
    uint32_t flags = 0;
    uint32_t last_w = 0, last_h = 0, last_band_rows = 0;  // layout of the last frame (rthip_group_pixel_status)
    int debug_fail = 0;  // tests: make the next frame fail inside the gather (1: an RCCL call of the group, 2: right after the render enqueue)
    std::string err;
};

namespace {

thread_local std::string g_group_err;

int gfail(rthip_group *g, int status, const std::string &msg) {
    if (g) g->err = msg;
    g_group_err = msg;
    return status;
}

#define GHIP(g, call)                                                                                                                   \
    do {                                                                                                                                \
        hipError_t e__ = (call);                                                                                                        \
        if (e__ != hipSuccess)                                                                                                          \
            return gfail(g, e__ == hipErrorOutOfMemory ? RTHIP_E_OOM : RTHIP_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define GNCCL(g, call)                                                                                             \
    do {                                                                                                           \
        ncclResult_t r__ = (call);                                                                                 \
        if (r__ != ncclSuccess) return gfail(g, RTHIP_E_RCCL, std::string(#call) + ": " + g_rccl.GetErrorString(r__)); \
    } while (0)
#define GCTX(g, i, call)                                                                                          \
    do {                                                                                                          \
        int s__ = (call);                                                                                         \
        if (s__) return gfail(g, s__, "device " + std::to_string((g)->devices[i]) + ": " + rthip_last_error((g)->ctx[i])); \
    } while (0)

// the calling thread's current device, put back when the entry point returns (the group moves it from device to device)
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

int ensure(rthip_group *g, int device, Buf &b, size_t bytes) {
    if (bytes <= b.bytes) return RTHIP_OK;
    GHIP(g, hipSetDevice(device));
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr, b.bytes = 0;
    GHIP(g, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return RTHIP_OK;
}

}  // namespace

extern "C" {

int rthip_band_row_owner(uint32_t band_rows, uint32_t ndev, uint32_t y, uint32_t *out_rank, uint32_t *out_local_row) {
    if (band_rows == 0 || ndev == 0) return RTHIP_E_BADARG;
    const uint32_t b = y / band_rows;
    if (out_rank) *out_rank = b % ndev;
    if (out_local_row) *out_local_row = (b / ndev) * band_rows + (y - b * band_rows);
    return RTHIP_OK;
}

const char *rthip_group_last_error(const rthip_group *g) { return g ? g->err.c_str() : g_group_err.c_str(); }
int rthip_group_size(const rthip_group *g) { return g ? int(g->devices.size()) : 0; }
rthip_ctx *rthip_group_context(rthip_group *g, int index) { return (g && index >= 0 && size_t(index) < g->ctx.size()) ? g->ctx[index] : nullptr; }

void rthip_group_destroy(rthip_group *g) {
    if (!g) return;
    DeviceGuard restore_device;
    for (size_t i = 0; i < g->ctx.size(); i++) {
        if (!g->ctx[i]) continue;  // creation failed at this device (it may not even exist)
        (void)hipSetDevice(g->devices[i]);
        (void)hipStreamSynchronize((hipStream_t)rt_ctx_stream(g->ctx[i]));
    }
    for (ncclComm_t c : g->comm)
        if (c) (void)g_rccl.CommDestroy(c);
    for (size_t i = 0; i < g->devices.size(); i++) {
        if (!g->ctx[i]) continue;
        (void)hipSetDevice(g->devices[i]);
        if (i < g->band_rgba.size() && g->band_rgba[i].p) (void)hipFree(g->band_rgba[i].p);
        if (i < g->band_accum.size() && g->band_accum[i].p) (void)hipFree(g->band_accum[i].p);
        if (i < g->rendered.size() && g->rendered[i]) (void)hipEventDestroy(g->rendered[i]);
    }
    if (!g->ctx.empty() && g->ctx[0]) (void)hipSetDevice(g->devices[0]);
    for (Buf *b : {&g->recv_rgba, &g->recv_accum, &g->full_rgba, &g->full_accum})
        if (b->p) (void)hipFree(b->p);
    if (g->g0) (void)hipEventDestroy(g->g0);
    if (g->g1) (void)hipEventDestroy(g->g1);
    for (rthip_ctx *c : g->ctx) rthip_destroy(c);
    delete g;
}

int rthip_group_create(const int *devices, int ndev, uint32_t flags, rthip_group **out_group) {
    if (!out_group) return RTHIP_E_BADARG;
    *out_group = nullptr;
    if (!devices || ndev < 1 || ndev > 64) return gfail(nullptr, RTHIP_E_BADARG, "rthip_group_create: 1..64 devices");
    if (flags & ~uint32_t(RTHIP_GROUP_F_PEER_COPY | RTHIP_GROUP_F_FORCE_RCCL)) return gfail(nullptr, RTHIP_E_BADARG, "rthip_group_create: unknown flag bits");
    if ((flags & RTHIP_GROUP_F_PEER_COPY) && (flags & RTHIP_GROUP_F_FORCE_RCCL)) return gfail(nullptr, RTHIP_E_BADARG, "rthip_group_create: contradictory flags");
    DeviceGuard restore_device;
    rthip_group *g = nullptr;
    try {
        g = new rthip_group();
        g->flags = flags;
        g->devices.assign(devices, devices + ndev);
        g->ctx.assign(ndev, nullptr);
        g->band_rgba.resize(ndev), g->band_accum.resize(ndev);
        g->rendered.assign(ndev, nullptr);
    } catch (...) {
        delete g;
        return gfail(nullptr, RTHIP_E_OOM, "rthip_group_create: host allocation failed");
    }
    int st = [&]() -> int {
        const bool rccl = (ndev > 1 && !(flags & RTHIP_GROUP_F_PEER_COPY)) || (flags & RTHIP_GROUP_F_FORCE_RCCL);
        if (rccl)
            for (int i = 0; i < ndev; i++)
                for (int j = 0; j < i; j++)
                    if (devices[i] == devices[j]) return gfail(g, RTHIP_E_BADARG, "rthip_group_create: a device is listed twice (RCCL needs one rank per GPU; RTHIP_GROUP_F_PEER_COPY allows it)");
        for (int i = 0; i < ndev; i++) {
            int s = rthip_create(devices[i], &g->ctx[i]);
            if (s) return gfail(g, s, std::string("rthip_create: ") + rthip_last_error(nullptr));
            GHIP(g, hipSetDevice(devices[i]));
            GHIP(g, hipEventCreateWithFlags(&g->rendered[i], hipEventDisableTiming));
        }
        GHIP(g, hipSetDevice(devices[0]));
        GHIP(g, hipEventCreate(&g->g0));
        GHIP(g, hipEventCreate(&g->g1));
        if (rccl) {
            std::string e = g_rccl.load();
            if (!e.empty()) return gfail(g, RTHIP_E_RCCL, e);
            g->comm.assign(ndev, nullptr);
            GNCCL(g, g_rccl.CommInitAll(g->comm.data(), ndev, devices));  // rccl.h:236
        }
        return RTHIP_OK;
    }();
    if (st) {
        g_group_err = g->err;
        rthip_group_destroy(g);
        return st;
    }
    *out_group = g;
    return RTHIP_OK;
}

int rthip_group_upload_scene(rthip_group *g, const rthip_scene_desc *scene) {
    if (!g) return RTHIP_E_BADARG;
    for (size_t i = 0; i < g->ctx.size(); i++) GCTX(g, i, rthip_upload_scene(g->ctx[i], scene));  // replicated
    return RTHIP_OK;
}

int rthip_group_set_camera(rthip_group *g, const rthip_camera *camera) {
    if (!g) return RTHIP_E_BADARG;
    for (size_t i = 0; i < g->ctx.size(); i++) GCTX(g, i, rthip_set_camera(g->ctx[i], camera));
    return RTHIP_OK;
}

int rthip_group_reset_accum(rthip_group *g) {
    if (!g) return RTHIP_E_BADARG;
    for (size_t i = 0; i < g->ctx.size(); i++) GCTX(g, i, rthip_reset_accum(g->ctx[i]));
    return RTHIP_OK;
}

// internal (tests/test_group_gpu.py): make the next rthip_group_render fail inside its gather
void rt_group_debug_fail_next(rthip_group *g, int what) {
    if (g) g->debug_fail = what;
}

}  // extern "C"

namespace {

struct Plane {
    bool on;
    std::vector<Buf> *band;
    Buf *recv, *full;
    size_t row_bytes;
    void *host;
};

// Grouped point-to-point gather of one plane to device 0 (rccl.h:700,722,923,933): every peer's rows ride their own xGMI link.
// Device 0's rows stay where they are (the de-interleave reads them) unless the group is a single device
// (RTHIP_GROUP_F_FORCE_RCCL), where the self-transfer is what exercises the library. Whatever fails between ncclGroupStart and
// ncclGroupEnd, the group is CLOSED before this returns: the first error is kept, the loop stops, ncclGroupEnd runs.
int rccl_gather(rthip_group *g, const Plane &pl, const std::vector<uint32_t> &rows, size_t slot, hipStream_t s0) {
    const uint32_t ndev = uint32_t(g->devices.size());
    const int d0 = g->devices[0];
    const uint32_t first = ndev == 1 ? 0u : 1u;
    GNCCL(g, g_rccl.GroupStart());
    int status = RTHIP_OK;
    std::string msg;
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e == hipSuccess) return true;
        status = RTHIP_E_HIP, msg = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    };
    auto nccl_ok = [&](ncclResult_t r, const char *what) {
        if (r == ncclSuccess) return true;
        status = RTHIP_E_RCCL, msg = std::string(what) + ": " + g_rccl.GetErrorString(r);
        return false;
    };
    for (uint32_t r = first; r < ndev && status == RTHIP_OK; r++) {
        if (rows[r] == 0) continue;
        const size_t bytes = size_t(rows[r]) * pl.row_bytes;
        if (!hip_ok(hipSetDevice(g->devices[r]), "hipSetDevice")) break;  // (each call with its communicator's device current)
        ncclComm_t send_comm = g->comm[r];
        if (g->debug_fail == 1) send_comm = nullptr, g->debug_fail = 0;  // tests: ncclSend rejects a null communicator
        if (!nccl_ok(g_rccl.Send((*pl.band)[r].p, bytes, ncclUint8, 0, send_comm, (hipStream_t)rt_ctx_stream(g->ctx[r])), "ncclSend")) break;
        if (!hip_ok(hipSetDevice(d0), "hipSetDevice")) break;
        if (!nccl_ok(g_rccl.Recv((char *)pl.recv->p + r * slot, bytes, ncclUint8, int(r), g->comm[0], s0), "ncclRecv")) break;
    }
    const ncclResult_t end = g_rccl.GroupEnd();  // always: an open group would swallow every later RCCL call of this thread
    (void)hipSetDevice(d0);
    if (status != RTHIP_OK) return gfail(g, status, msg);
    if (end != ncclSuccess) return gfail(g, RTHIP_E_RCCL, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(end));
    return RTHIP_OK;
}

// After a failure with passes already enqueued: wait for whatever runs, and forget every device's running sum — some devices
// may have advanced theirs and others not, and a later RTHIP_F_ACCUMULATE frame must not mix sample counts across bands.
void quiesce_after_failure(rthip_group *g) {
    for (size_t i = 0; i < g->ctx.size(); i++) {
        if (!g->ctx[i]) continue;
        if (hipSetDevice(g->devices[i]) == hipSuccess) (void)hipStreamSynchronize((hipStream_t)rt_ctx_stream(g->ctx[i]));
        (void)rthip_reset_accum(g->ctx[i]);
        rthip_stats dummy;
        (void)rt_ctx_collect_stats(g->ctx[i], &dummy);  // drop a pending counter read-out, if any
    }
    (void)hipGetLastError();
}

int group_render(rthip_group *g, const rthip_render_params *params, uint8_t *rgba8_out, float *accum_out, rthip_stats *stats, bool &enqueued) {
    const uint32_t ndev = uint32_t(g->devices.size());
    const uint32_t W = params->width, H = params->height;
    const uint32_t R = params->band_rows ? params->band_rows : 8u;
    if (H > 65535u) return gfail(g, RTHIP_E_BADARG, "rthip_group_render: height above 65535 is not supported");
    std::vector<rthip_render_params> p(ndev, *params);
    std::vector<uint32_t> rows(ndev, 0);
    uint32_t pad_rows = 0;
    for (uint32_t i = 0; i < ndev; i++) {
        p[i].band_rows = R, p[i].band_first = i, p[i].band_stride = ndev;
        if (rthip_owned_rows(&p[i], &rows[i])) return gfail(g, RTHIP_E_BADARG, "bad band parameters");
        if (rows[i] > pad_rows) pad_rows = rows[i];
    }
    const size_t row_rgba = size_t(W) * 4, row_acc = size_t(W) * 12;
    const int d0 = g->devices[0];
    // ---- buffers ----
    for (uint32_t i = 0; i < ndev; i++) {
        int s;
        if (rgba8_out && (s = ensure(g, g->devices[i], g->band_rgba[i], pad_rows * row_rgba + 16))) return s;
        if (accum_out && (s = ensure(g, g->devices[i], g->band_accum[i], pad_rows * row_acc + 16))) return s;
    }
    {
        int s;
        if (rgba8_out && ((s = ensure(g, d0, g->recv_rgba, size_t(ndev) * pad_rows * row_rgba + 16)) || (s = ensure(g, d0, g->full_rgba, size_t(H) * row_rgba + 16)))) return s;
        if (accum_out && ((s = ensure(g, d0, g->recv_accum, size_t(ndev) * pad_rows * row_acc + 16)) || (s = ensure(g, d0, g->full_accum, size_t(H) * row_acc + 16)))) return s;
    }
    // ---- every device renders its bands on its own stream ----
    rthip_stats dummy;
    for (uint32_t i = 0; i < ndev; i++) {
        enqueued = true;  // (from here on a failure leaves passes in flight and running sums out of step: the caller quiesces)
        GCTX(g, i, rt_ctx_render_deferred(g->ctx[i], &p[i], rgba8_out ? g->band_rgba[i].p : nullptr, accum_out ? g->band_accum[i].p : nullptr,
                                          stats ? &dummy : nullptr));
        GHIP(g, hipSetDevice(g->devices[i]));
        GHIP(g, hipEventRecord(g->rendered[i], (hipStream_t)rt_ctx_stream(g->ctx[i])));
    }
    if (g->debug_fail == 2) {
        g->debug_fail = 0;
        return gfail(g, RTHIP_E_INTERNAL, "rthip_group_render: failure requested by rt_group_debug_fail_next");
    }
    // ---- gather to device 0, de-interleave, copy out ----
    hipStream_t s0 = (hipStream_t)rt_ctx_stream(g->ctx[0]);
    GHIP(g, hipSetDevice(d0));
    GHIP(g, hipEventRecord(g->g0, s0));
    const Plane planes[2] = {{rgba8_out != nullptr, &g->band_rgba, &g->recv_rgba, &g->full_rgba, row_rgba, rgba8_out},
                             {accum_out != nullptr, &g->band_accum, &g->recv_accum, &g->full_accum, row_acc, accum_out}};
    for (const Plane &pl : planes) {
        if (!pl.on) continue;
        const size_t slot = size_t(pad_rows) * pl.row_bytes;
        if (!g->comm.empty()) {
            int s = rccl_gather(g, pl, rows, slot, s0);
            if (s) return s;
        } else {
            for (uint32_t r = 1; r < ndev; r++) {
                if (rows[r] == 0) continue;
                GHIP(g, hipStreamWaitEvent(s0, g->rendered[r], 0));
                GHIP(g, hipMemcpyPeerAsync((char *)pl.recv->p + r * slot, d0, (*pl.band)[r].p, g->devices[r], size_t(rows[r]) * pl.row_bytes, s0));
            }
        }
        const bool self_via_recv = !g->comm.empty() && ndev == 1;
        GHIP(g, rt_launch_deinterleave(pl.full->p, pl.recv->p, self_via_recv ? pl.recv->p : (*pl.band)[0].p, uint32_t(pl.row_bytes / 4), H, R, ndev, pad_rows, s0));
    }
    GHIP(g, hipEventRecord(g->g1, s0));
    for (const Plane &pl : planes)
        if (pl.on) GHIP(g, hipMemcpyAsync(pl.host, pl.full->p, size_t(H) * pl.row_bytes, hipMemcpyDeviceToHost, s0));
    GHIP(g, hipStreamSynchronize(s0));
    for (uint32_t i = 1; i < ndev; i++) {  // the peers' streams have nothing left but their sends
        GHIP(g, hipSetDevice(g->devices[i]));
        GHIP(g, hipStreamSynchronize((hipStream_t)rt_ctx_stream(g->ctx[i])));
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (uint32_t i = 0; i < ndev; i++) {
            if (rows[i] == 0) continue;
            rthip_stats s;
            GCTX(g, i, rt_ctx_collect_stats(g->ctx[i], &s));
            stats->paths += s.paths, stats->rays += s.rays, stats->sphere_tests += s.sphere_tests, stats->plane_tests += s.plane_tests;
            stats->tri_tests += s.tri_tests, stats->node_tests += s.node_tests, stats->sky_evals += s.sky_evals, stats->shadow_rays += s.shadow_rays;
            for (int k = 0; k < 4; k++) stats->hits[k] += s.hits[k];
            stats->rows += s.rows, stats->pixels_traced += s.pixels_traced, stats->pixels_final += s.pixels_final;
            if (s.kernel_ms > stats->kernel_ms) stats->kernel_ms = s.kernel_ms;  // the devices run side by side
            stats->nb_samples = s.nb_samples;
        }
        float ms = 0.0f;
        GHIP(g, hipSetDevice(d0));
        GHIP(g, hipEventElapsedTime(&ms, g->g0, g->g1));
        stats->gather_ms = ms;
    }
    g->last_w = W, g->last_h = H, g->last_band_rows = R;
    return RTHIP_OK;
}

}  // namespace

extern "C" {

int rthip_group_render(rthip_group *g, const rthip_render_params *params, uint8_t *rgba8_out, float *accum_out, rthip_stats *stats) {
    if (!g) return RTHIP_E_BADARG;
    if (!params) return gfail(g, RTHIP_E_BADARG, "params is NULL");
    if (params->band_first != 0 || params->band_stride > 1)
        return gfail(g, RTHIP_E_BADARG, "rthip_group_render: band_first/band_stride belong to the group (leave them 0); band_rows = rows per band, 0 = 8");
    DeviceGuard restore_device;
    bool enqueued = false;
    int st;
    try {
        st = group_render(g, params, rgba8_out, accum_out, stats, enqueued);
    } catch (const std::bad_alloc &) {
        st = gfail(g, RTHIP_E_OOM, "host allocation failed in rthip_group_render");
    } catch (...) {
        st = gfail(g, RTHIP_E_INTERNAL, "unexpected exception in rthip_group_render");
    }
    if (st != RTHIP_OK && enqueued) {
        const std::string keep = g->err;
        quiesce_after_failure(g);
        g->err = keep + " (the group's running sums were reset)";
        g_group_err = g->err;
    }
    return st;
}

// RTHIP_F_ADAPTIVE frames: PixelColor.status / same_color_count of every pixel of the last frame, height*width bytes each, rows in
// image order (every device reports its rows, rthip_pixel_status; the band map puts them back).
int rthip_group_pixel_status(rthip_group *g, uint8_t *status_out, uint8_t *same_color_count_out, uint64_t capacity_pixels) {
    if (!g) return RTHIP_E_BADARG;
    if (!status_out) return gfail(g, RTHIP_E_BADARG, "status_out is NULL");
    if (g->last_w == 0) return gfail(g, RTHIP_E_STATE, "no frame has been rendered");
    const uint32_t W = g->last_w, H = g->last_h, R = g->last_band_rows, ndev = uint32_t(g->devices.size());
    if (capacity_pixels < uint64_t(W) * H) return gfail(g, RTHIP_E_BADARG, "rthip_group_pixel_status: the buffers are smaller than the frame");
    DeviceGuard restore_device;
    try {
        std::vector<uint8_t> st, cnt;
        for (uint32_t i = 0; i < ndev; i++) {
            uint32_t rows = 0, width = 0;
            uint64_t nb = 0;
            GCTX(g, i, rthip_accum_layout(g->ctx[i], &rows, &width, &nb));
            if (rows == 0) continue;
            st.resize(size_t(rows) * W), cnt.resize(size_t(rows) * W);
            GCTX(g, i, rthip_pixel_status(g->ctx[i], st.data(), cnt.data(), st.size()));
            for (uint32_t y = 0; y < H; y++) {
                uint32_t rank = 0, lrow = 0;
                rthip_band_row_owner(R, ndev, y, &rank, &lrow);
                if (rank != i) continue;
                std::memcpy(status_out + size_t(y) * W, st.data() + size_t(lrow) * W, W);
                if (same_color_count_out) std::memcpy(same_color_count_out + size_t(y) * W, cnt.data() + size_t(lrow) * W, W);
            }
        }
        return RTHIP_OK;
    } catch (const std::bad_alloc &) {
        return gfail(g, RTHIP_E_OOM, "host allocation failed in rthip_group_pixel_status");
    } catch (...) {
        return gfail(g, RTHIP_E_INTERNAL, "unexpected exception in rthip_group_pixel_status");
    }
}

}  // extern "C"
