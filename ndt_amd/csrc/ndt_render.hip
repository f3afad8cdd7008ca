// ndt_render.hip -- ndt_hip_render*: render_image (ndt.c:900) behind the C ABI: argument checks, and the choice between
// the deterministic pass, recursive anti-aliasing and the sampled paths.
#include "ndt_ctx.hpp"

// true anaglyph (ndt.c:643-647): red = luminance of the left eye's colour, blue = of the right eye's
__global__ void k_anaglyph(const double *left, const double *right, double *out, long long n_pixels)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const double *l = left + 4 * i, *r = right + 4 * i;
    out[4 * i + 0] = 0.299 * l[0] + 0.587 * l[1] + 0.114 * l[2];
    out[4 * i + 1] = 0;
    out[4 * i + 2] = 0.299 * r[0] + 0.587 * r[1] + 0.114 * r[2];
    out[4 * i + 3] = 1.0;
}

void ndt_impl::launch_anaglyph(hipStream_t s, const double *left, const double *right, double *out, long long n_pixels)
{
    if (n_pixels <= 0) return;
    hipLaunchKernelGGL(k_anaglyph, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, s, left, right, out, n_pixels);
}

extern "C" int ndt_hip_render_depth_device(ndt_hip_ctx *ctx, const ndt_render_params *p, void *d_rgba, void *d_depth,
                                           ndt_render_stats *stats)
{
    if (!ctx || !p || !d_rgba) return fail(NDT_E_INVALID, "NULL argument");
    if (!ctx->have_scene) return fail(NDT_E_STATE, "no scene uploaded");
    if (p->samples < 1) return fail(NDT_E_INVALID, "samples=%d", p->samples);
    const bool stochastic = p->samples > 1 || ctx->has_area_lights;
    // (a stochastic anti-aliased render -- a lens, area lights or -n > 1 under -a -- has no depth map here: the map would be
    // the last lens sample's of every first-pass corner)
    if (p->recursive_aa && d_depth && (stochastic || ctx->aperture_radius != 0.0))
        return fail(NDT_E_UNSUPPORTED, "no depth map beside a stochastic anti-aliased render (lens, area lights or samples > 1 with recursive_aa)");
    if (p->samples > 1 && ctx->aperture_radius != 0.0 && !ctx->have_local_axes)
        return fail(NDT_E_INVALID, "depth of field needs the camera's local axes (camera.h:69-71) in the flat scene");
    if (p->width < 1 || p->height < 1 || p->row_step < 1 || p->row_begin < 0) return fail(NDT_E_INVALID, "bad geometry");
    if (p->stereo < NDT_STEREO_MONO || p->stereo > NDT_STEREO_HIDEF) return fail(NDT_E_UNSUPPORTED, "stereo mode %d", p->stereo);
    if (p->stereo != NDT_STEREO_MONO && !ctx->have_eyes) return fail(NDT_E_INVALID, "stereo needs leftEye / rightEye (camera.h:60-61) in the flat scene");
    for (int k = 0; k < 4; ++k)
        if (p->reserved[k] != 0) return fail(NDT_E_INVALID, "reserved render parameter set");
    if (p->recursive_aa && ctx->aperture_radius != 0.0 && !ctx->have_local_axes)
        return fail(NDT_E_INVALID, "depth of field needs the camera's local axes (camera.h:69-71) in the flat scene");
    if (p->recursive_aa && (p->aa_diff < 0 || p->aa_depth > 24)) return fail(NDT_E_INVALID, "bad anti-aliasing parameters");
    // (the reference's recursive_resample averages the alpha of its samples, and the samples that fall on the 45 blank lines
    // of a frame-packed image come back with an alpha nobody set, ndt.c:662, 625-627: there is nothing to be equal to)
    if (p->recursive_aa && p->stereo == NDT_STEREO_HIDEF)
        return fail(NDT_E_UNSUPPORTED, "recursive anti-aliasing of a frame-packed image reads uninitialised alpha in the reference (ndt.c:662)");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int rows = ndt_hip_shard_rows(p->height, p->row_begin, p->row_step);
    ndt_render_stats st{};
    if (rows == 0) {
        if (stats) *stats = st;
        return NDT_OK;
    }
    const long long n_pixels = (long long)rows * p->width;
    if (p->max_optic_depth <= 0) {
        // get_ray_color returns black without tracing (ndt.c:340); averages of black are black
        launch_fill_black(s, (double *)d_rgba, n_pixels);
        if (d_depth) HIP_TRY(hipMemsetAsync(d_depth, 0, (size_t)n_pixels * sizeof(double), s));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
        if (stats) *stats = st;
        return NDT_OK;
    }
    int rc;
    // the reference's switch: -a with depth >= 0 and diff < 256 resamples, otherwise the first pass is copied (ndt.c:1040)
    if (p->recursive_aa) {
        rc = render_antialiased(ctx, p, d_rgba, st, d_depth);
    } else if (stochastic) {
        rc = render_sampled(ctx, p, d_rgba, st, d_depth);
    } else {
        RenderGeom rg{};
        rg.width = p->width;
        rg.height = p->height;
        rg.row_begin = p->row_begin;
        rg.row_step = p->row_step;
        rg.rows = rows;
        rg.tiles_x = (rg.width + 7) / 8;
        rg.tiles_y = (rg.rows + 7) / 8;
        const long long n_primary = (long long)rg.tiles_x * rg.tiles_y * 64;
        if (n_primary > 0x3fffffffLL) return fail(NDT_E_UNSUPPORTED, "image too large for one call");
        rg.n_primary = (int)n_primary;
        rg.max_depth = p->max_optic_depth;
        rg.specular = p->specular ? 1 : 0;
        rg.img_w = p->width;
        rg.img_h = p->height;
        rg.aspect_w = p->width;
        rg.aspect_h = p->height;
        rg.eye = 1;
        if (p->stereo == NDT_STEREO_ANAGLYPH) {
            // two full renders, one per eye (ndt.c:636-647); the depth map is the left eye's
            AaBuffers buf(ctx);
            double *left = nullptr, *right = nullptr;
            if ((rc = buf.get(&left, (size_t)n_pixels * 4))) return rc;
            if ((rc = buf.get(&right, (size_t)n_pixels * 4))) return rc;
            ndt_render_stats one{};
            rg.eye = 0;
            if ((rc = render_pass(ctx, rg, p->profile != 0, left, one, d_depth))) return rc;
            add_stats(st, one);
            rg.eye = 2;
            if ((rc = render_pass(ctx, rg, p->profile != 0, right, one))) return rc;
            add_stats(st, one);
            launch_anaglyph(s, left, right, (double *)d_rgba, n_pixels);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s));
            rc = NDT_OK;
        } else {
            rg.stereo = p->stereo;
            if (p->stereo == NDT_STEREO_HIDEF) {
                // frame packing (ndt.c:614-631, 927-928): the aspect is width/1080 and the 45 blank lines between the
                // eyes stay black (the reference leaves their alpha unset; 1 here)
                rg.aspect_h = 1080;
                launch_fill_black(s, (double *)d_rgba, n_pixels);
                if (d_depth) HIP_TRY(hipMemsetAsync(d_depth, 0, (size_t)n_pixels * sizeof(double), s));
            }
            rc = render_pass(ctx, rg, p->profile != 0, d_rgba, st, d_depth);
        }
    }
    if (rc) return rc;
    if (stats) *stats = st;
    return NDT_OK;
}

extern "C" int ndt_hip_render_device(ndt_hip_ctx *ctx, const ndt_render_params *p, void *d_rgba, ndt_render_stats *stats)
{
    return ndt_hip_render_depth_device(ctx, p, d_rgba, nullptr, stats);
}

extern "C" int ndt_hip_render(ndt_hip_ctx *ctx, const ndt_render_params *p, double *rgba, ndt_render_stats *stats)
{
    return ndt_hip_render_depth(ctx, p, rgba, nullptr, stats);
}

extern "C" int ndt_hip_render_depth(ndt_hip_ctx *ctx, const ndt_render_params *p, double *rgba, double *depth, ndt_render_stats *stats)
{
    if (!ctx || !p || !rgba) return fail(NDT_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int rows = ndt_hip_shard_rows(p->height, p->row_begin, p->row_step);
    const size_t img_bytes = (size_t)rows * (size_t)(p->width > 0 ? p->width : 0) * 4 * sizeof(double);
    const size_t bytes = img_bytes + (depth ? img_bytes / 4 : 0);      // the depth map sits behind the image
    if (img_bytes == 0) return ndt_hip_render_depth_device(ctx, p, (void *)rgba, nullptr, stats);
    if (ctx->d_out_bytes < bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->d_out) HIP_TRY(hipFree(ctx->d_out));
        ctx->d_out = nullptr;
        HIP_TRY(hipMalloc(&ctx->d_out, bytes));
        ctx->d_out_bytes = bytes;
    }
    void *d_depth = depth ? (void *)((char *)ctx->d_out + img_bytes) : nullptr;
    int rc = ndt_hip_render_depth_device(ctx, p, ctx->d_out, d_depth, stats);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(rgba, ctx->d_out, img_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (depth) HIP_TRY(hipMemcpyAsync(depth, d_depth, img_bytes / 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return NDT_OK;
}

