// ndt_sampled.hip -- `-n samples` > 1, depth of field and area lights (ndt.c:116-147, 470-568) on top of render_pass.
#include "ndt_ctx.hpp"

// ------------------------------------------------------------------ -n samples > 1 (ndt.c:470-568)
//
// Jittered samples inside the pixel + a lens sample per ray, the adaptive loop on top: at least
// `samples` samples per pixel, then more while the running mean still moves by more than 1/256
// (at most 10000).  The reference draws from one global drand48 stream in pixel order, which no
// parallel renderer can follow; here every (pixel, sample) has its own counter-based stream, so the
// image is reproducible and independent of sharding, and parity with the reference is statistical
// (tests compare against the oracle, which follows the reference's stream exactly).
// Round r renders sample r of every pixel that is still sampling, through the pipeline in list mode.

// the random stream of sample `round` of image pixel `pixel`
__device__ __forceinline__ unsigned long long ns_sample_key(unsigned long long pixel, unsigned int round)
{
    return ndt_rng_mix(pixel * 0x100000001b3ull + round);
}

// sample r of the active pixels: (i + dx, j - dy) and the lens offsets (ndt.c:505-514, 527-541)
// (`per` consecutive samples per pixel in one pass: sample a*per + r is the pixel's sample number round + r)
// pos (or nullptr): the positions of a list render -- "pixel" k is image position (pos[2k], pos[2k+1]); its streams are named by
// the position itself, so that they do not depend on which shard, level or task asked for it
__global__ void k_ns_samples(const int *active, int n_active, int per, int width, int row_begin, int row_step, unsigned int round0,
                             double aperture, int jitter, int lens, double xs, double ys, double *samples, unsigned long long *keys,
                             unsigned long long seed, const double *pos)
{
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= (long long)n_active * per) return;
    const int pix = active[a / per];
    const unsigned int round = round0 + (unsigned int)(a % per);
    double i, j;
    unsigned long long id;
    if (pos) {
        i = pos[2ll * pix];
        j = pos[2ll * pix + 1];
        id = ndt_rng_mix((unsigned long long)__double_as_longlong(i)) ^ ndt_rng_mix((unsigned long long)__double_as_longlong(j) + 0x7f4a7c159e3779b9ull);
    } else {
        const int l = pix / width;
        i = pix % width;
        j = row_begin + l * row_step;
        id = (unsigned long long)(row_begin + l * row_step) * (unsigned long long)width + (unsigned long long)(pix % width);
    }
    // (seed: option "sample_seed", 0 by default -- another value is another, independent set of streams: what the tests use to
    // draw an ENSEMBLE of device images and compare it with the oracle's, two samples of the same distribution)
    const unsigned long long key = ns_sample_key(id, round) ^ ndt_rng_mix(seed * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull) * (seed != 0ull);
    keys[a] = key;
    double dx = 0.0, dy = 0.0, ax = 0.0, ay = 0.0;
    if (jitter) {       // -n > 1 without recursive anti-aliasing (ndt.c:505)
        dx = ndt_rng_uniform(key, 1000);
        dy = ndt_rng_uniform(key, 1001);
    }
    if (lens) {         // -n > 1 or recursive anti-aliasing (ndt.c:528); with neither the area lights are all that is random
        unsigned int k = 1002;
        do {        // reject samples outside the unit disk
            ax = 2 * ndt_rng_uniform(key, k) - 1.0;
            ay = 2 * ndt_rng_uniform(key, k + 1) - 1.0;
            k += 2;
        } while (ax * ax + ay * ay > 1.0 && k < 1064);
    }
    double *q = samples + 4ll * a;
    // x = orig_x + dx/width, y = orig_y + dy/height (y grows upwards).  The jitter is added AFTER render_pixel has split a
    // side-by-side / over-under image (ndt.c:590-612 before :505-514): in the squeezed direction it spans 1/width of the
    // eye's image, half a pixel of the packed one -- xs / ys = 0.5 there, so that k_primary's "/ 0.5" gives it back whole
    q[0] = i + dx * xs;
    q[1] = j - dy * ys;
    q[2] = ax * aperture;
    q[3] = ay * aperture;
}

// get_pixel_color's loop body after the sample has been traced (ndt.c:553-567), and its continuation test
// A pass may have rendered `per` samples ahead for every pixel; they are consumed one by one exactly as the
// loop would, and the ones after the loop's exit are dropped (they were speculation: fewer, fuller passes).
// sdepth: a depth-map render -- every sample overwrites the pixel's depth (get_ray_color, ndt.c:362-373), so the LAST sample the
// loop takes leaves its value.  blank_from / blank_to: image rows of a frame-packed image that stay black (ndt.c:625-627: no
// sample is taken there)
__global__ void k_ns_accumulate(const int *active, int n_active, int per, const double *colours, const double *sdepth, unsigned int round0,
                                int min_samples, int width, int row_begin, int row_step, int blank_from, int blank_to, double *acc,
                                int *taken, int *next, int *next_count)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = a < n_active;
    const int pix = in_range ? active[a] : 0;
    double *t = acc + 6ll * pix;            // t_clr rgba + clr_diff + depth
    double clr_diff = in_range ? t[4] : 0.0;
    bool go_on = in_range;
    int used = 0;
    if (in_range && blank_to >= blank_from) {
        const int j = row_begin + (pix / width) * row_step;
        if (j >= blank_from && j <= blank_to) {
            go_on = false;
            t[3] = 1.0;         // (the reference leaves the alpha of these lines unset; 1 here, as in the deterministic path)
        }
    }
    for (int r = 0; r < per && go_on; ++r) {
        const double *l = colours + 4ll * ((long long)a * per + r);
        const int i = (int)round0 + r;
        if (i > 1) {
            const double dr = fabs(t[0] / (i - 1) - (t[0] + l[0]) / i);
            const double dg = fabs(t[1] / (i - 1) - (t[1] + l[1]) / i);
            const double db = fabs(t[2] / (i - 1) - (t[2] + l[2]) / i);
            const double gb = (dg > db) ? dg : db;
            clr_diff = (dr > gb) ? dr : gb;
        }
        t[0] += l[0]; t[1] += l[1]; t[2] += l[2]; t[3] += l[3];
        if (sdepth) t[5] = sdepth[(long long)a * per + r];
        ++used;
        const int done = i + 1;
        go_on = done < min_samples || (done < 10000 && clr_diff > 1.0 / 256.0);
    }
    if (in_range) {
        t[4] = clr_diff;
        taken[pix] += used;
    }
    const int slot = block_append(next_count, go_on);
    if (go_on) next[slot] = pix;
}

__global__ void k_ns_init(double *acc, int *active, int *taken, long long n_pixels)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    double *t = acc + 6 * i;
    t[0] = t[1] = t[2] = t[3] = 0.0;
    t[4] = 256.0;
    t[5] = 0.0;
    active[i] = (int)i;
    taken[i] = 0;
}

__global__ void k_ns_finish(const double *acc, const int *taken, double *rgba, double *depth, long long n_pixels, unsigned long long *used_total)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long used = 0;
    if (i < n_pixels) {
        const int n = taken[i] > 0 ? taken[i] : 1;
        for (int c = 0; c < 4; ++c) rgba[4 * i + c] = acc[6 * i + c] / n;
        if (depth) depth[i] = acc[6 * i + 5];
        used = (unsigned long long)taken[i];
    }
    for (int d = 32; d > 0; d >>= 1) used += __shfl_down(used, d, 64);
    // one atomic a workgroup (see block_append)
    __shared__ unsigned long long wave_used[16];
    if ((threadIdx.x & 63) == 0) wave_used[threadIdx.x >> 6] = used;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long sum = 0;
        for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) sum += wave_used[w];
        if (sum) atomicAdd(used_total, sum);
    }
}

// One eye's image (or the whole image, for the modes that split it by position).  eye: 0 left, 1 centre, 2 right; stereo: what
// the ray pipeline is told (0 for the two passes of an anaglyph); salt: the two eyes of an anaglyph draw from different streams,
// as the reference's two get_pixel_color calls draw different numbers (ndt.c:639-640).
// (sl: the positions of a list render instead of the image's pixels -- the samples of an anti-aliased render, which calls this
// from inside its own pass and so hands out the scratch of the nested pool)
static int render_sampled_eye(ndt_hip_ctx *ctx, const ndt_render_params *p, int eye, int stereo, unsigned long long salt, void *d_rgba,
                              ndt_render_stats &total, void *d_depth, const SampledList *sl = nullptr)
{
    hipStream_t s = ctx->stream;
    const int W = sl ? sl->img_w : p->width, H = sl ? sl->img_h : p->height;
    const int rows = ndt_hip_shard_rows(H, p->row_begin, p->row_step);
    const long long n_pixels = sl ? sl->n_pos : (long long)rows * W;
    if (n_pixels > 0x3fffffffLL) return fail(NDT_E_UNSUPPORTED, "image too large for one call");
    if (n_pixels <= 0) return NDT_OK;
    AaBuffers buf(ctx, sl != nullptr);
    int rc;
    double *acc = nullptr, *samples = nullptr, *colours = nullptr;
    unsigned long long *keys = nullptr;
    int *list[2] = { nullptr, nullptr }, *taken = nullptr, *counter = nullptr;
    double *sdepth = nullptr;
    // frame packing: 1080 lines left eye, 45 blank, 1080 right eye (ndt.c:614-631)
    const int blank_from = (stereo == NDT_STEREO_HIDEF && !sl) ? 1080 : 1, blank_to = (stereo == NDT_STEREO_HIDEF && !sl) ? 1080 + 45 : 0;
    if ((rc = buf.get(&acc, (size_t)n_pixels * 6))) return rc;
    if ((rc = buf.get(&list[0], (size_t)n_pixels))) return rc;
    if ((rc = buf.get(&list[1], (size_t)n_pixels))) return rc;
    if ((rc = buf.get(&taken, (size_t)n_pixels))) return rc;
    if ((rc = buf.get(&counter, 1))) return rc;
    const unsigned g_all = (unsigned)((n_pixels + 255) / 256);
    hipLaunchKernelGGL(k_ns_init, dim3(g_all), dim3(256), 0, s, acc, list[0], taken, n_pixels);
    int n_active = (int)n_pixels;
    // samples per pixel and pass: the first `samples` are certain to be needed; after that the loop may stop at
    // any sample, so passes speculate further ahead the fewer pixels are left (about 4 M primaries per pass)
    const long long per_pass = 4ll << 20;
    size_t cap_samples = 0;
    int flip = 0;
    for (unsigned int round = 0; n_active > 0 && round < 10000;) {
        long long per = round < (unsigned int)p->samples ? (long long)p->samples - round : per_pass / n_active;
        if (per > per_pass / n_active) per = per_pass / n_active;
        // ... but never more than have been taken already: the waste stays below a factor of two
        if (round >= (unsigned int)p->samples && per > (long long)(round < 2 ? 1 : round)) per = round < 2 ? 1 : round;
        if (per < 1) per = 1;
        if (per > 64) per = 64;
        if (round + per > 10000) per = 10000 - round;
        const long long n_s = (long long)n_active * per;
        if ((size_t)n_s > cap_samples) {
            cap_samples = (size_t)n_s;
            if ((rc = buf.get(&samples, cap_samples * 4))) return rc;
            if ((rc = buf.get(&colours, cap_samples * 4))) return rc;
            if ((rc = buf.get(&keys, cap_samples + 64))) return rc;
            if (d_depth && (rc = buf.get(&sdepth, cap_samples + 64))) return rc;
        }
        hipLaunchKernelGGL(k_ns_samples, dim3((unsigned)((n_s + 255) / 256)), dim3(256), 0, s, list[flip], n_active, (int)per, W,
                           p->row_begin, p->row_step, round, ctx->aperture_radius, (p->samples > 1 && !sl) ? 1 : 0, (p->samples > 1 || sl) ? 1 : 0,
                           // (the jitter is 1/width x 1/height of the IMAGE; a frame-packed eye image is 1080 lines of its 2205, ndt.c:482-483, 629)
                           stereo == NDT_STEREO_SIDE_SIDE ? 0.5 : 1.0, stereo == NDT_STEREO_OVER_UNDER ? 0.5 : stereo == NDT_STEREO_HIDEF ? 1080.0 / H : 1.0,
                           samples, keys, (unsigned long long)ctx->sample_seed + salt, sl ? sl->d_pos : nullptr);
        RenderGeom gs{};
        gs.samples = samples;
        gs.n_samples = (int)n_s;
        gs.n_primary = (int)((n_s + 63) & ~63LL);
        gs.width = (int)n_s;
        gs.rows = 1;
        gs.max_depth = p->max_optic_depth;
        gs.specular = p->specular ? 1 : 0;
        gs.img_w = W;
        gs.img_h = H;
        gs.aspect_w = sl ? sl->aspect_w : W;
        gs.aspect_h = sl ? sl->aspect_h : stereo == NDT_STEREO_HIDEF ? 1080 : H;
        gs.eye = eye;
        gs.stereo = stereo;
        gs.lens = 1;
        gs.raw_samples = 1;
        gs.pixel_halves = sl ? 0 : 1;
        gs.sample_keys = keys;
        ndt_render_stats st{};
        if ((rc = render_pass(ctx, gs, p->profile != 0, colours, st, d_depth ? sdepth : nullptr))) return rc;
        add_stats(total, st);
        HIP_TRY(hipMemsetAsync(counter, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_ns_accumulate, dim3((unsigned)((n_active + 1023) / 1024)), dim3(1024), 0, s, list[flip], n_active, (int)per, colours, d_depth ? sdepth : nullptr,
                           round, p->samples, W, p->row_begin, p->row_step, blank_from, blank_to, acc, taken, list[flip ^ 1], counter);
        HIP_TRY(hipMemcpyAsync(&n_active, counter, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        flip ^= 1;
        round += (unsigned int)per;
    }
    unsigned long long *used_total = nullptr, used_host = 0;
    if ((rc = buf.get(&used_total, 1))) return rc;
    HIP_TRY(hipMemsetAsync(used_total, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_ns_finish, dim3((unsigned)((n_pixels + 1023) / 1024)), dim3(1024), 0, s, acc, taken, (double *)d_rgba, (double *)d_depth, n_pixels, used_total);
    HIP_TRY(hipMemcpyAsync(&used_host, used_total, sizeof(used_host), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    total.aa_samples += (long long)used_host;      // samples the adaptive loop consumed (the rays_* counts include the speculation)
    return NDT_OK;
}

int ndt_impl::render_sampled_list(ndt_hip_ctx *ctx, const ndt_render_params *p, const SampledList &sl, int eye, int stereo,
                                  unsigned long long salt, void *d_rgba, ndt_render_stats &total)
{
    return render_sampled_eye(ctx, p, eye, stereo, salt, d_rgba, total, nullptr, &sl);
}

int ndt_impl::render_sampled(ndt_hip_ctx *ctx, const ndt_render_params *p, void *d_rgba, ndt_render_stats &total, void *d_depth)
{
    if (p->stereo != NDT_STEREO_ANAGLYPH) return render_sampled_eye(ctx, p, 1, p->stereo, 0ull, d_rgba, total, d_depth);
    // anaglyph (ndt.c:636-647): every pixel runs get_pixel_color for the left eye, then for the right; the depth map is the left's
    const int rows = ndt_hip_shard_rows(p->height, p->row_begin, p->row_step);
    const long long n_pixels = (long long)rows * p->width;
    // The eyes' scratch images: one allocation the context keeps (grown when a frame needs more), not a hipMalloc / hipFree pair a
    // frame -- each of those synchronises the device and stalls every other context on it.  (Not from the AaBuffers pools:
    // render_sampled_eye walks those from their start.)
    const size_t eye_bytes = (size_t)n_pixels * 4 * sizeof(double);
    if (ctx->d_eyes_bytes < 2 * eye_bytes) {
        if (ctx->d_eyes) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->d_eyes);
            ctx->d_eyes = nullptr;
            ctx->d_eyes_bytes = 0;
        }
        hipError_t e = hipMalloc(&ctx->d_eyes, 2 * eye_bytes);
        if (e != hipSuccess) return fail(NDT_E_NOMEM, "hipMalloc: %s", hipGetErrorString(e));
        ctx->d_eyes_bytes = 2 * eye_bytes;
    }
    double *left = (double *)ctx->d_eyes, *right = left + (size_t)n_pixels * 4;
    int rc = render_sampled_eye(ctx, p, 0, 0, 0ull, left, total, d_depth);
    if (!rc) rc = render_sampled_eye(ctx, p, 2, 0, 0x5eed0000000000ffull, right, total, nullptr);
    if (!rc) {
        launch_anaglyph(ctx->stream, left, right, (double *)d_rgba, n_pixels);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(NDT_E_DEVICE, "anaglyph mix");
    }
    return rc;
}

