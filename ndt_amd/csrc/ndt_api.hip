// ndt_api.hip -- the plain C ABI of include/ndt_hip.h: contexts, scene upload, trace_kd batches, pixel_d2c.
//
// There is no CPU fallback here: without a usable HIP device every entry point fails with NDT_E_DEVICE.
#include "ndt_ctx.hpp"
#include <ctype.h>

// ------------------------------------------------------------------ errors

static thread_local char g_err[512] = "";
int ndt_impl::fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *ndt_hip_last_error(void) { return g_err; }
extern "C" int ndt_hip_abi_version(void) { return NDT_HIP_ABI_VERSION; }

extern "C" int32_t ndt_hip_shard_rows(int32_t height, int32_t row_begin, int32_t row_step)
{
    if (row_step < 1 || row_begin < 0 || row_begin >= height) return 0;
    return (height - row_begin + row_step - 1) / row_step;
}

static const NdtKernelTable *table_for(int dims)
{
    switch (dims) {
    case 3: return ndt_kernel_table_3();
    case 4: return ndt_kernel_table_4();
    case 5: return ndt_kernel_table_5();
    case 6: return ndt_kernel_table_6();
    case 7: return ndt_kernel_table_7();
    case 8: return ndt_kernel_table_8();
    case 9: return ndt_kernel_table_9();
    case 10: return ndt_kernel_table_10();
    case 11: return ndt_kernel_table_11();
    case 12: return ndt_kernel_table_12();
    default: return nullptr;
    }
}

extern "C" int ndt_hip_create(int device, ndt_hip_ctx **out)
{
    if (!out) return fail(NDT_E_INVALID, "ndt_hip_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(NDT_E_DEVICE, "no HIP device available (%s); libndt_hip has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(NDT_E_INVALID, "device %d out of range (have %d)", device, count);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(NDT_E_DEVICE, "device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);
    ndt_hip_ctx *ctx = new ndt_hip_ctx();
    ctx->device = device;
    {
        // the environment is read here, once: nothing on the render or upload path looks at it
        static const char *const names[] = { "hybrid_level", "stream_below", "stream_below_list", "hull_box", "face_box", "face_tree", "face_groups", "item_sets", "gate_prepass", "gate_prepass_below", "item_boxes", "leaf_history", "leaf_scan", "leaf_scan_group", "coop", "coop_budget_us", "coop_max_live", "coop_tail_only", "coop_waves", "multi_path", "sample_seed", "stream_fused", "fuse_primaries", "shade_pair", "debug_levels",
                                             "exit_probe", "shade_probe", "stream_probe", "test_small_pool" };
        const char *pl = getenv("NDT_HIP_PIPELINE");
        if (pl) (void)ndt_hip_set_option(ctx, "pipeline", !strcmp(pl, "levels") ? 1 : !strcmp(pl, "stream") ? 2 : !strcmp(pl, "hybrid") ? 3 : 0);
        for (const char *nm : names) {
            char env[64] = "NDT_HIP_";
            size_t k = strlen(env);
            for (const char *q = nm; *q && k + 1 < sizeof(env); ++q) env[k++] = (char)toupper((unsigned char)*q);
            env[k] = 0;
            const char *v = getenv(env);
            // (historical spellings: NDT_HIP_NO_HULL_BOX=1, NDT_HIP_NO_FACE_BOX=1, NDT_HIP_NO_SHADE_PAIR=1)
            if (v && *v) (void)ndt_hip_set_option(ctx, nm, !strcmp(nm, "shade_probe") ? atoll(v) + 1 : atoll(v));
        }
        if (getenv("NDT_HIP_NO_HULL_BOX")) ctx->hull_box = false;
        if (getenv("NDT_HIP_NO_FACE_BOX")) ctx->face_box = false;
        if (getenv("NDT_HIP_NO_SHADE_PAIR") && atoi(getenv("NDT_HIP_NO_SHADE_PAIR"))) ctx->shade_pair = false;
    }
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return fail(NDT_E_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    e = hipHostMalloc((void **)&ctx->h_counters, 128 * sizeof(int), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_levels, (NDT_MAX_LEVELS + 1) * sizeof(LevelRange), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_mail, (NDT_MAX_LEVELS + 2) * sizeof(LevelRange), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_mail_tag, (NDT_MAX_LEVELS + 2) * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&ctx->d_mail, ctx->h_mail, 0);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&ctx->d_mail_tag, ctx->h_mail_tag, 0);
    if (e == hipSuccess) memset(ctx->h_mail_tag, 0, (NDT_MAX_LEVELS + 2) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_done, 16 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&ctx->d_done, ctx->h_done, 0);
    if (e == hipSuccess) memset(ctx->h_done, 0, 16 * sizeof(unsigned long long));
    if (e != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return fail(NDT_E_DEVICE, "hipHostMalloc: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return NDT_OK;
}

extern "C" int ndt_hip_destroy(ndt_hip_ctx *ctx)
{
    if (!ctx) return NDT_OK;
    worker_stop(ctx);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_workspace(ctx);
    free_stage(ctx);
    free_async(ctx);
    if (ctx->d_shard) (void)hipFree(ctx->d_shard);
    if (ctx->d_image) (void)hipFree(ctx->d_image);
    for (auto &slot : ctx->pool)
        if (slot.first) (void)hipFree(slot.first);
    for (auto &slot : ctx->pool2)
        if (slot.first) (void)hipFree(slot.first);
    if (ctx->d_blob) (void)hipFree(ctx->d_blob);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    if (ctx->d_eyes) (void)hipFree(ctx->d_eyes);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    if (ctx->h_levels) (void)hipHostFree(ctx->h_levels);
    if (ctx->h_mail) (void)hipHostFree(ctx->h_mail);
    if (ctx->h_mail_tag) (void)hipHostFree(ctx->h_mail_tag);
    if (ctx->h_done) (void)hipHostFree(ctx->h_done);
    for (hipEvent_t ev : ctx->ev_pool) (void)hipEventDestroy(ev);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return NDT_OK;
}

extern "C" int ndt_hip_set_option(ndt_hip_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name) return fail(NDT_E_INVALID, "NULL argument");
    const bool on = value != 0;
    if (!strcmp(name, "pipeline")) {
        if (value < 0 || value > 3) return fail(NDT_E_INVALID, "pipeline %lld", (long long)value);
        ctx->pipeline = (int)value;
    } else if (!strcmp(name, "hybrid_level")) ctx->hybrid_level = (int)value;
    else if (!strcmp(name, "stream_below")) ctx->stream_below = value;
    else if (!strcmp(name, "stream_below_list")) ctx->stream_below_list = value;
    else if (!strcmp(name, "hull_box")) ctx->hull_box = on;
    else if (!strcmp(name, "face_box")) ctx->face_box = on;
    else if (!strcmp(name, "face_tree")) ctx->face_tree = on;
    else if (!strcmp(name, "face_groups")) ctx->face_groups = on;
    else if (!strcmp(name, "item_sets")) ctx->item_sets = on;
    else if (!strcmp(name, "item_boxes")) ctx->item_boxes = on;
    else if (!strcmp(name, "gate_prepass_below")) ctx->gate_prepass_below = value;
    else if (!strcmp(name, "gate_prepass")) ctx->gate_prepass = value < 0 ? 0 : value > 2 ? 2 : (int)value;
    else if (!strcmp(name, "sample_seed")) ctx->sample_seed = value;
    else if (!strcmp(name, "leaf_scan")) ctx->leaf_scan = on;
    else if (!strcmp(name, "coop")) ctx->coop = on;
    else if (!strcmp(name, "coop_budget_us")) ctx->coop_budget_us = value < 0 ? 0 : value > 1000000 ? 1000000 : (int)value;
    else if (!strcmp(name, "coop_max_live")) ctx->coop_max_live = value < 0 ? 0 : value > 64 ? 64 : (int)value;
    else if (!strcmp(name, "coop_tail_only")) ctx->coop_tail_only = on;
    else if (!strcmp(name, "coop_waves")) ctx->coop_waves = value < 1 ? 1 : value > 16 ? 16 : (int)value;
    else if (!strcmp(name, "leaf_scan_group")) ctx->leaf_scan_group = value < 1 ? 1 : value > 64 ? 64 : (int)value;
    else if (!strcmp(name, "leaf_history")) ctx->leaf_history = value < 0 ? 0 : value > 4 ? 4 : (int)value;
    else if (!strcmp(name, "multi_path")) {
        if (value < 0 || value > 2) return fail(NDT_E_INVALID, "multi_path %lld", (long long)value);
        ctx->multi_path = (int)value;
    } else if (!strcmp(name, "stream_fused")) ctx->stream_fused = on;
    else if (!strcmp(name, "fuse_primaries")) ctx->fuse_primaries = value < 0 ? -1 : on ? 1 : 0;
    else if (!strcmp(name, "shade_pair")) ctx->shade_pair = on;
    else if (!strcmp(name, "debug_levels")) ctx->debug_levels = on;
    else if (!strcmp(name, "exit_probe")) ctx->exit_probe = on;
    else if (!strcmp(name, "shade_probe")) ctx->shade_probe = (int)value - 1;
    else if (!strcmp(name, "stream_probe")) ctx->stream_probe = on;
    else if (!strcmp(name, "test_small_pool")) ctx->test_small_pool = on;
    else return fail(NDT_E_INVALID, "unknown option '%s'", name);
    return NDT_OK;
}

extern "C" void *ndt_hip_stream(ndt_hip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" int ndt_hip_synchronize(ndt_hip_ctx *ctx)
{
    if (!ctx) return fail(NDT_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return NDT_OK;
}

extern "C" int ndt_hip_upload_scene(ndt_hip_ctx *ctx, const ndt_flat_scene *fs)
{
    if (!ctx || !fs) return fail(NDT_E_INVALID, "NULL argument");
    if (fs->abi_version != NDT_HIP_ABI_VERSION) return fail(NDT_E_INVALID, "scene ABI %d, library ABI %d", fs->abi_version, NDT_HIP_ABI_VERSION);
    if (fs->dims < NDT_MIN_DIMS || fs->dims > NDT_MAX_DIMS)
        return fail(NDT_E_UNSUPPORTED, "%d dimensions: kernels are built for %d..%d", fs->dims, NDT_MIN_DIMS, NDT_MAX_DIMS);
    if (fs->cam_type < 0 || fs->cam_type > 2) return fail(NDT_E_UNSUPPORTED, "camera type %d", fs->cam_type);
    if (fs->cam_type != 0 && (fs->cam_local_x_off < 0 || fs->cam_local_y_off < 0 || fs->cam_local_z_off < 0))
        return fail(NDT_E_INVALID, "VR / panorama cameras need the local axes (camera.h:69-71) in the flat scene");
    if (fs->n_lights < 0 || fs->n_lights > NDT_MAX_LIGHTS) return fail(NDT_E_UNSUPPORTED, "%d lights (max %d)", fs->n_lights, NDT_MAX_LIGHTS);
    if (fs->n_objects < 0 || fs->n_items < 0 || fs->n_items > fs->n_objects) return fail(NDT_E_INVALID, "object counts");
    if (fs->n_kd_nodes < 0 || fs->n_inf < 0 || fs->n_leaf_refs < 0) return fail(NDT_E_INVALID, "kd-tree counts");
    if (fs->n_objects > 0 && !fs->objects) return fail(NDT_E_INVALID, "objects is NULL");
    if (fs->n_lights > 0 && !fs->lights) return fail(NDT_E_INVALID, "lights is NULL");
    if (!fs->vecs) return fail(NDT_E_INVALID, "vecs is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->have_scene = false;
    int rc = build_blob(ctx, fs);
    if (rc) return rc;
    ctx->dims = fs->dims;
    ctx->kt = table_for(fs->dims);
    if (!ctx->kt) return fail(NDT_E_UNSUPPORTED, "no kernels for %d dimensions", fs->dims);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->d_blob_words < ctx->blob.size()) {
        if (ctx->d_blob) HIP_TRY(hipFree(ctx->d_blob));
        ctx->d_blob = nullptr;
        HIP_TRY(hipMalloc((void **)&ctx->d_blob, ctx->blob.size() * sizeof(double)));
        ctx->d_blob_words = ctx->blob.size();
    }
    HIP_TRY(hipMemcpyAsync(ctx->d_blob, ctx->blob.data(), ctx->blob.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->aperture_radius = fs->cam_aperture_radius;
    ctx->have_scene = true;
    if (ctx->debug_levels)
        fprintf(stderr, "ndt_hip: scene blob: %d-D, %d items / %d objects, %d kd nodes of depth %d; %d words staged by the trace kernels (%d KB), %d in all; tier %d, %d mask word(s)\n",
                ctx->dims, ctx->sd.n_items, ctx->sd.n_objects, ctx->sd.n_kd_nodes, ctx->sd.kd_depth, ctx->sd.trace_words, ctx->sd.trace_words / 128,
                ctx->sd.total_words, ctx->tier, ctx->sd.mask_words);
    return NDT_OK;
}

// pixel_d2c, image.h:36-39
__global__ void k_quantize(const double *rgba, unsigned char *out, long long n_values)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_values) return;
    const double d = rgba[i];
    double m = (1.0 < d) ? 1.0 : d;
    m = (0.0 > m) ? 0.0 : m;
    out[i] = (unsigned char)(sqrt(m) * 255);
}

extern "C" int ndt_hip_quantize_device(ndt_hip_ctx *ctx, const void *d_rgba, void *d_rgba8, int64_t n_pixels)
{
    if (!ctx || !d_rgba || !d_rgba8 || n_pixels < 0) return fail(NDT_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const long long n = n_pixels * 4;
    if (n == 0) return NDT_OK;
    hipLaunchKernelGGL(k_quantize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)d_rgba,
                       (unsigned char *)d_rgba8, n);
    HIP_TRY(hipGetLastError());
    return NDT_OK;
}

// ------------------------------------------------------------------ trace_kd batches

extern "C" int ndt_hip_trace_rays(ndt_hip_ctx *ctx, int64_t n_rays, const double *o, const double *v, const double *dist_limit,
                                  int32_t *obj, double *hit, double *normal)
{
    if (!ctx || n_rays < 0 || (n_rays > 0 && (!o || !v || !dist_limit || !obj || !hit || !normal)))
        return fail(NDT_E_INVALID, "bad argument");
    if (!ctx->have_scene) return fail(NDT_E_STATE, "no scene uploaded");
    if (n_rays == 0) return NDT_OK;
    if (n_rays > 0x3fffffffLL) return fail(NDT_E_UNSUPPORTED, "too many rays for one call");
    HIP_TRY(hipSetDevice(ctx->device));
    const int n = ctx->dims;
    const long long cnt = n_rays;
    int rc = ensure_workspace(ctx, cnt > ctx->ws.cap ? cnt : ctx->ws.cap, ctx->ws.sh_cap > 0 ? ctx->ws.sh_cap : 4096);
    if (rc) return rc;
    Workspace ws = ctx->ws;
    hipStream_t s = ctx->stream;
    // ray-major host arrays -> the pool's tiles of 64 slots (component-major inside a tile)
    const long long padded = (cnt + 63) & ~63LL;
    std::vector<double> so((size_t)n * padded, 0.0), sv((size_t)n * padded, 0.0);
    auto tile_at = [n](long long r, int c) { return (size_t)((r >> 6) * (long long)(n * 64) + c * 64 + (r & 63)); };
    for (long long r = 0; r < cnt; ++r)
        for (int c = 0; c < n; ++c) {
            so[tile_at(r, c)] = o[r * n + c];
            sv[tile_at(r, c)] = v[r * n + c];
        }
    HIP_TRY(hipMemcpyAsync(ws.ray_o, so.data(), so.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ws.ray_v, sv.data(), sv.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ws.frac, dist_limit, cnt * sizeof(double), hipMemcpyHostToDevice, s));
    TraceJob tj{};
    tj.n_seg = 0;
    tj.dense.o = ws.ray_o; tj.dense.v = ws.ray_v; tj.dense.stride = ws.cap; tj.dense.lim = ws.frac; tj.dense.valid = nullptr;
    tj.dense.out_obj = ws.hit_obj; tj.dense.out_prim = ws.hit_prim; tj.begin = 0; tj.count = cnt; tj.levels = nullptr;
    tj.queue = ws.counters + NDT_CNT_QUEUE;
    HIP_TRY(hipMemsetAsync(ws.counters + NDT_CNT_QUEUE, 0, NDT_QUEUE_INTS * sizeof(int), s));
    tj.publish_level = -1;
    coop_setup(ctx, tj, nullptr);
    ctx->kt->trace(s, ctx->d_blob, ctx->sd, ws, tj, ctx->tier, ctx->sd.mask_words, nullptr, nullptr);
    ctx->kt->hitpoints(s, ctx->d_blob, ctx->sd, ws.ray_o, ws.ray_v, ws.cap, ws.hit_prim, ws.hit_p, ws.hit_n, cnt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(obj, ws.hit_obj, cnt * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(so.data(), ws.hit_p, so.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(sv.data(), ws.hit_n, sv.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (long long r = 0; r < cnt; ++r)
        for (int c = 0; c < n; ++c) {
            hit[r * n + c] = so[tile_at(r, c)];
            normal[r * n + c] = sv[tile_at(r, c)];
        }
    return NDT_OK;
}
