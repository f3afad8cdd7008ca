// ndt_multi.hip -- one frame over several contexts (one per GPU, or several on one GPU) from one host thread, and the
// 8-bit output paths.  The reference's row mode deals rows to MPI ranks (ndt.c:812-820) and sum-reduces full-size images
// up a binary tree (mpi_collect_image, ndt.c:1277-1309); here every device renders its rows into its own HBM and then
// PUSHES them -- quantised on the way when the caller wants the 8-bit image -- into the assembled frame on the first
// context's device with plain stores over xGMI.  Nothing is reduced, nothing is zero-padded.
#include "ndt_ctx.hpp"
#include <condition_variable>
#include <mutex>
#include <thread>

// rows of a compact shard -> rows row0, row0+step, ... of the assembled image; F64: 4 doubles per pixel, as 2 x 16 bytes;
// RGBA8: pixel_d2c (image.h:36-39) on the way, 4 bytes per pixel.  `dst` may be peer memory.
__global__ void __launch_bounds__(256) k_push_rows_f64(const double *__restrict__ shard, double *dst, int width, int rows, int row0,
                                                       int step)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one pixel half (2 doubles) per thread
    if (i >= (long long)rows * width * 2) return;
    const long long pix = i >> 1;
    const int r = (int)(pix / width), x = (int)(pix % width);
    const ndt_v2d v = reinterpret_cast<const ndt_v2d *>(shard)[i];
    reinterpret_cast<ndt_v2d *>(dst)[(((long long)row0 + (long long)r * step) * width + x) * 2 + (i & 1)] = v;
}
__device__ __forceinline__ unsigned int d2c(double d)
{
    double m = (1.0 < d) ? 1.0 : d;
    m = (0.0 > m) ? 0.0 : m;
    return (unsigned int)(unsigned char)(sqrt(m) * 255);
}
__global__ void __launch_bounds__(256) k_push_rows_rgba8(const double *__restrict__ shard, unsigned int *dst, int width, int rows,
                                                         int row0, int step)
{
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (long long)rows * width) return;
    const int r = (int)(pix / width), x = (int)(pix % width);
    const ndt_v2d a = reinterpret_cast<const ndt_v2d *>(shard)[2 * pix], b = reinterpret_cast<const ndt_v2d *>(shard)[2 * pix + 1];
    dst[((long long)row0 + (long long)r * step) * width + x] = d2c(a.x) | (d2c(a.y) << 8) | (d2c(b.x) << 16) | (d2c(b.y) << 24);
}

static void launch_push(hipStream_t s, int format, const void *shard, void *dst, int width, int rows, int row0, int step)
{
    const long long pixels = (long long)rows * width;
    if (pixels <= 0) return;
    if (format == NDT_IMAGE_RGBA8)
        hipLaunchKernelGGL(k_push_rows_rgba8, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, s, (const double *)shard,
                           (unsigned int *)dst, width, rows, row0, step);
    else
        hipLaunchKernelGGL(k_push_rows_f64, dim3((unsigned)((2 * pixels + 255) / 256)), dim3(256), 0, s, (const double *)shard,
                           (double *)dst, width, rows, row0, step);
}

static size_t pixel_bytes(int format) { return format == NDT_IMAGE_RGBA8 ? 4 : 4 * sizeof(double); }

static int ensure_bytes(ndt_hip_ctx *ctx, void **buf, size_t *have, size_t want)
{
    if (*have >= want) return NDT_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (*buf) HIP_TRY(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(buf, want));
    *have = want;
    return NDT_OK;
}

extern "C" int ndt_hip_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
extern "C" int ndt_hip_device(ndt_hip_ctx *ctx) { return ctx ? ctx->device : -1; }

extern "C" int ndt_hip_render_rgba8(ndt_hip_ctx *ctx, const ndt_render_params *p, uint8_t *rgba8, ndt_render_stats *stats)
{
    ndt_hip_ctx *one[1] = { ctx };
    return ndt_hip_render_multi(one, 1, p, NDT_IMAGE_RGBA8, rgba8, stats);
}

// render_image + the save-time quantisation, WITHOUT waiting for the bytes to reach the host: the frame is rendered (the call
// returns when it is complete in HBM), quantised into one of two device buffers, and its copy to `rgba8` runs on a copy stream
// of its own behind the next call's rendering.  What a caller that writes frame after frame pays per frame is then the render
// alone (the reference's "rendering took" ends with the pixels in host memory, ndt.c:978-984: ndt_hip_render_rgba8 -- render,
// then 8 MB over PCIe, in sequence -- is 1.80 ms where the render is 1.46).
//   ndt_hip_render_rgba8_async(ctx, p, rgba8, stats)   rgba8: host memory, pinned (hipHostMalloc / torch pin_memory) for the copy
//                                                      to be asynchronous.  When the call returns, THIS frame is rendered and on
//                                                      its way, and every EARLIER frame of the context has arrived (its copy had
//                                                      the whole of this frame's rendering to finish: the wait is free);
//   ndt_hip_render_rgba8_wait(ctx)                     ... the last one too.
extern "C" int ndt_hip_render_rgba8_async(ndt_hip_ctx *ctx, const ndt_render_params *p, uint8_t *rgba8, ndt_render_stats *stats)
{
    if (!ctx || !p || !rgba8) return fail(NDT_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int rows = ndt_hip_shard_rows(p->height, p->row_begin, p->row_step);
    const size_t pixels = (size_t)(rows > 0 ? rows : 0) * (size_t)(p->width > 0 ? p->width : 0);
    if (pixels == 0) {
        if (stats) *stats = ndt_render_stats{};
        return NDT_OK;
    }
    if (!ctx->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(hipEventCreateWithFlags(&ctx->ev_quantised[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ctx->ev_copied[k], hipEventDisableTiming));
        }
    }
    const int k = ctx->rgba8_turn;
    ctx->rgba8_turn ^= 1;
    if (ctx->copy_pending[k]) {             // this buffer's previous frame must have left
        HIP_TRY(hipEventSynchronize(ctx->ev_copied[k]));
        ctx->copy_pending[k] = false;
    }
    int rc = ensure_bytes(ctx, &ctx->d_shard, &ctx->d_shard_bytes, pixels * 4 * sizeof(double));
    if (rc) return rc;
    if ((rc = ensure_bytes(ctx, &ctx->d_rgba8[k], &ctx->d_rgba8_bytes[k], pixels * 4))) return rc;
    if ((rc = ndt_hip_render_device(ctx, p, ctx->d_shard, stats))) return rc;
    launch_push(ctx->stream, NDT_IMAGE_RGBA8, ctx->d_shard, ctx->d_rgba8[k], p->width, rows, 0, 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_quantised[k], ctx->stream));
    HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_quantised[k], 0));
    HIP_TRY(hipMemcpyAsync(rgba8, ctx->d_rgba8[k], pixels * 4, hipMemcpyDeviceToHost, ctx->copy_stream));
    HIP_TRY(hipEventRecord(ctx->ev_copied[k], ctx->copy_stream));
    ctx->copy_pending[k] = true;
    if (ctx->copy_pending[k ^ 1]) {         // the frame before: it travelled while this one was rendered
        HIP_TRY(hipEventSynchronize(ctx->ev_copied[k ^ 1]));
        ctx->copy_pending[k ^ 1] = false;
    }
    return NDT_OK;
}

extern "C" int ndt_hip_render_rgba8_wait(ndt_hip_ctx *ctx)
{
    if (!ctx) return fail(NDT_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    for (int k = 0; k < 2; ++k)
        if (ctx->copy_pending[k]) {
            HIP_TRY(hipEventSynchronize(ctx->ev_copied[k]));
            ctx->copy_pending[k] = false;
        }
    return NDT_OK;
}

void ndt_impl::free_async(ndt_hip_ctx *ctx)
{
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        for (int k = 0; k < 2; ++k) {
            if (ctx->ev_quantised[k]) (void)hipEventDestroy(ctx->ev_quantised[k]);
            if (ctx->ev_copied[k]) (void)hipEventDestroy(ctx->ev_copied[k]);
        }
        (void)hipStreamDestroy(ctx->copy_stream);
        ctx->copy_stream = nullptr;
    }
    for (int k = 0; k < 2; ++k) {
        if (ctx->d_rgba8[k]) (void)hipFree(ctx->d_rgba8[k]);
        ctx->d_rgba8[k] = nullptr;
        ctx->d_rgba8_bytes[k] = 0;
    }
}

// ---- per-context worker threads: a context's frames are enqueued by one thread of its own, so that the contexts of a
// multi-device render run their (host-polled) frames side by side while the caller stays a single thread
struct ndt_impl::CtxWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false, has_job = false, done = false;
    std::function<int()> job;
    int rc = 0;
    char err[512] = "";
    void loop()
    {
        for (;;) {
            std::function<int()> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || has_job; });
                if (stop) return;
                j = job;
                has_job = false;
            }
            const int r = j();
            {
                std::lock_guard<std::mutex> lk(mu);
                rc = r;
                if (r) snprintf(err, sizeof(err), "%s", ndt_hip_last_error());     // the message lives in THIS thread
                done = true;
            }
            cv.notify_all();
        }
    }
    void submit(std::function<int()> j)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(j);
            has_job = true;
            done = false;
        }
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
};

void ndt_impl::worker_stop(ndt_hip_ctx *ctx)
{
    CtxWorker *w = ctx->worker;
    if (!w) return;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->stop = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    delete w;
    ctx->worker = nullptr;
}

static CtxWorker *worker_of(ndt_hip_ctx *ctx)
{
    if (!ctx->worker) {
        ctx->worker = new CtxWorker();
        ctx->worker->th = std::thread([w = ctx->worker] { w->loop(); });
    }
    return ctx->worker;
}

// what context k does for its share of the frame: render rows (begin, step) into its own shard buffer, then push them
// into `d_dst` (the assembled image on the first context's device).  Three ways, reported by ndt_hip_multi_path_taken():
//   NDT_MULTI_LOCAL   the context lives on the first context's device: plain stores
//   NDT_MULTI_PEER    another device with peer access: the push kernel stores over xGMI into the first device's HBM
//   NDT_MULTI_STAGED  no peer access (or option "multi_path" 2): the shard travels as ONE runtime copy (hipMemcpyPeerAsync)
//                     into this context's staging buffer on the first device and is pushed from there by a kernel of that
//                     device, on a stream of that device this context owns -- buffer and stream are kept between frames
// Option "multi_path": 0 auto (local / peer / staged in that order of preference), 1 never staged (fails without peer
// access), 2 always staged -- also for a context on the first device, which is how the staged path is tested on one GPU.
static int render_and_push(ndt_hip_ctx *ctx, ndt_hip_ctx *first, ndt_render_params sp, int format, void *d_dst, int row0_out,
                           int step_out, ndt_render_stats *st)
{
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->multi_path_taken = NDT_MULTI_NONE;
    const int rows = ndt_hip_shard_rows(sp.height, sp.row_begin, sp.row_step);
    if (rows <= 0) {
        *st = ndt_render_stats{};
        return NDT_OK;
    }
    const size_t shard_bytes = (size_t)rows * sp.width * 4 * sizeof(double);
    int rc = ensure_bytes(ctx, &ctx->d_shard, &ctx->d_shard_bytes, shard_bytes);
    if (rc) return rc;
    if ((rc = ndt_hip_render_device(ctx, &sp, ctx->d_shard, st))) return rc;
    int path = NDT_MULTI_STAGED;
    if (ctx->multi_path != 2) {
        if (ctx->device == first->device) {
            path = NDT_MULTI_LOCAL;
        } else {
            // peer stores: enabled once per (device, peer) pair; "already enabled" is success
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, ctx->device, first->device) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(first->device, 0);
                if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) path = NDT_MULTI_PEER;
                (void)hipGetLastError();
            }
            if (path == NDT_MULTI_STAGED && ctx->multi_path == 1)
                return fail(NDT_E_DEVICE, "multi_path 1: device %d has no peer access to device %d", ctx->device, first->device);
        }
    }
    if (path != NDT_MULTI_STAGED) {
        launch_push(ctx->stream, format, ctx->d_shard, d_dst, sp.width, rows, row0_out, step_out);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->multi_path_taken = path;
        return NDT_OK;
    }
    // staged: buffer and stream live on the FIRST device and belong to this context (one per source context, so that the
    // copies of different contexts do not wait for each other)
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    hipError_t e = hipSetDevice(first->device);
    if (e == hipSuccess && ctx->d_stage && ctx->stage_device != first->device) {
        // the first context changed devices since the last frame: the old staging area goes
        (void)hipSetDevice(ctx->stage_device);
        (void)hipStreamDestroy(ctx->stage_stream);
        (void)hipFree(ctx->d_stage);
        ctx->d_stage = nullptr;
        ctx->stage_stream = nullptr;
        ctx->d_stage_bytes = 0;
        e = hipSetDevice(first->device);
    }
    if (e == hipSuccess && !ctx->stage_stream) {
        e = hipStreamCreateWithFlags(&ctx->stage_stream, hipStreamNonBlocking);
        ctx->stage_device = first->device;
    }
    if (e == hipSuccess && ctx->d_stage_bytes < shard_bytes) {
        if (ctx->d_stage) (void)hipFree(ctx->d_stage);
        ctx->d_stage = nullptr;
        ctx->d_stage_bytes = 0;
        e = hipMalloc(&ctx->d_stage, shard_bytes);
        if (e == hipSuccess) ctx->d_stage_bytes = shard_bytes;
    }
    if (e == hipSuccess) e = hipMemcpyPeerAsync(ctx->d_stage, first->device, ctx->d_shard, ctx->device, shard_bytes, ctx->stage_stream);
    if (e == hipSuccess) {
        launch_push(ctx->stage_stream, format, ctx->d_stage, d_dst, sp.width, rows, row0_out, step_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stage_stream);
    (void)hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(NDT_E_DEVICE, "staged gather from device %d: %s", ctx->device, hipGetErrorString(e));
    ctx->multi_path_taken = NDT_MULTI_STAGED;
    return NDT_OK;
}

extern "C" int ndt_hip_multi_path_taken(ndt_hip_ctx *ctx) { return ctx ? ctx->multi_path_taken : NDT_MULTI_NONE; }

// (ndt_hip_destroy) the staging area lives on another device than the context
void ndt_impl::free_stage(ndt_hip_ctx *ctx)
{
    if (!ctx->d_stage && !ctx->stage_stream) return;
    (void)hipSetDevice(ctx->stage_device);
    if (ctx->stage_stream) {
        (void)hipStreamSynchronize(ctx->stage_stream);
        (void)hipStreamDestroy(ctx->stage_stream);
    }
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    ctx->d_stage = nullptr;
    ctx->stage_stream = nullptr;
    ctx->d_stage_bytes = 0;
    (void)hipSetDevice(ctx->device);
}

extern "C" int ndt_hip_render_multi_device(ndt_hip_ctx *const *ctxs, int32_t n_ctx, const ndt_render_params *p, int32_t format,
                                           void *d_out, ndt_render_stats *stats)
{
    if (!ctxs || n_ctx < 1 || n_ctx > 64 || !p || !d_out) return fail(NDT_E_INVALID, "bad argument");
    if (format != NDT_IMAGE_F64 && format != NDT_IMAGE_RGBA8) return fail(NDT_E_INVALID, "image format %d", format);
    for (int k = 0; k < n_ctx; ++k) {
        if (!ctxs[k]) return fail(NDT_E_INVALID, "context %d is NULL", k);
        if (!ctxs[k]->have_scene) return fail(NDT_E_STATE, "context %d has no scene uploaded", k);
        for (int j = 0; j < k; ++j)
            if (ctxs[j] == ctxs[k]) return fail(NDT_E_INVALID, "context %d is listed twice", k);
    }
    if (p->width < 1 || p->height < 1 || p->row_step < 1 || p->row_begin < 0) return fail(NDT_E_INVALID, "bad geometry");
    ndt_hip_ctx *first = ctxs[0];
    std::vector<ndt_render_stats> st((size_t)n_ctx);
    std::vector<int> rcs((size_t)n_ctx, NDT_OK);
    // context k: image rows p->row_begin + (k + i*n_ctx)*p->row_step = rows k, k+n_ctx, ... of the output
    for (int k = n_ctx - 1; k >= 0; --k) {
        ndt_render_params sp = *p;
        sp.row_begin = p->row_begin + k * p->row_step;
        sp.row_step = p->row_step * n_ctx;
        auto job = [=, &st]() { return render_and_push(ctxs[k], first, sp, format, d_out, k, n_ctx, &st[(size_t)k]); };
        if (k > 0) worker_of(ctxs[k])->submit(job);
        else rcs[0] = job();                    // the caller's thread renders the first context's rows itself
    }
    char err[512] = "";
    if (rcs[0]) snprintf(err, sizeof(err), "%s", ndt_hip_last_error());
    for (int k = 1; k < n_ctx; ++k) {
        rcs[(size_t)k] = ctxs[k]->worker->wait();
        if (rcs[(size_t)k] && !err[0]) snprintf(err, sizeof(err), "context %d: %s", k, ctxs[k]->worker->err);
    }
    (void)hipSetDevice(first->device);
    for (int k = 0; k < n_ctx; ++k)
        if (rcs[(size_t)k]) return fail(rcs[(size_t)k], "%s", err);
    if (stats) {
        ndt_render_stats total{};
        for (int k = 0; k < n_ctx; ++k) {
            const ndt_render_stats &s = st[(size_t)k];
            total.rays_primary += s.rays_primary;
            total.rays_secondary += s.rays_secondary;
            total.rays_shadow += s.rays_shadow;
            total.rays_ref_equiv += s.rays_ref_equiv;
            total.pixels_resampled += s.pixels_resampled;
            total.aa_samples += s.aa_samples;
            total.trace_launches += s.trace_launches;
            total.node_capacity += s.node_capacity;
            if (s.levels > total.levels) total.levels = s.levels;
            if (s.trace_ms > total.trace_ms) total.trace_ms = s.trace_ms;
            if (s.frame_ms > total.frame_ms) total.frame_ms = s.frame_ms;
        }
        *stats = total;
    }
    return NDT_OK;
}

extern "C" int ndt_hip_render_multi(ndt_hip_ctx *const *ctxs, int32_t n_ctx, const ndt_render_params *p, int32_t format, void *out,
                                    ndt_render_stats *stats)
{
    if (!ctxs || n_ctx < 1 || !ctxs[0] || !p || !out) return fail(NDT_E_INVALID, "bad argument");
    if (format != NDT_IMAGE_F64 && format != NDT_IMAGE_RGBA8) return fail(NDT_E_INVALID, "image format %d", format);
    ndt_hip_ctx *first = ctxs[0];
    HIP_TRY(hipSetDevice(first->device));
    const int rows = ndt_hip_shard_rows(p->height, p->row_begin, p->row_step);
    const size_t bytes = (size_t)(rows > 0 ? rows : 0) * (size_t)(p->width > 0 ? p->width : 0) * pixel_bytes(format);
    if (bytes == 0) {
        if (stats) *stats = ndt_render_stats{};
        return NDT_OK;
    }
    int rc = ensure_bytes(first, &first->d_image, &first->d_image_bytes, bytes);
    if (rc) return rc;
    if ((rc = ndt_hip_render_multi_device(ctxs, n_ctx, p, format, first->d_image, stats))) return rc;
    HIP_TRY(hipMemcpyAsync(out, first->d_image, bytes, hipMemcpyDeviceToHost, first->stream));
    HIP_TRY(hipStreamSynchronize(first->stream));
    return NDT_OK;
}
