// ndt_aa.hip -- Whitted's recursive anti-aliasing (`-a diff,depth`; ndt.c:655-733, 1039-1087) on top of render_pass.
#include "ndt_ctx.hpp"

// ------------------------------------------------------------------ Whitted's recursive anti-aliasing (ndt.c:655-733, 1039-1087)
//
// The reference recurses per pixel; here the recursion tree is walked level by level over all
// pixels at once.  A task = one call of recursive_resample: a square of side `step` at (x, y)
// with its four corner colours.  Level L holds the tasks with step 2^-L; each renders 5 new
// samples (through render_pass in list mode), forms the four quarter averages, and spawns a
// child task for every quarter whose corners differ by more than the threshold.  Results flow
// back up: a task's colour is the average of its quarters, a quarter being replaced by its
// child's colour where one was spawned (ndt.c:685-705).  Averages use image_avg_dbl_pixels4's
// operation order (image.c:1175-1197), so every colour equals the recursion's.

struct AaTask {
    double x, y;                // top-left corner, in pixels of the first-pass image
    double p[4][4];             // corner colours p1..p4 (top-left, top-right, bottom-left, bottom-right), rgba
    double sp[4][4];            // quarter colours sp1..sp4
    long long parent;           // level 0: output pixel index; deeper: task index in the level above
    int quad;                   // which quarter of the parent this task refines
    int _pad;
};


__device__ __forceinline__ void aa_avg4(const double *p1, const double *p2, const double *p3, const double *p4, double *avg, double *var)
{
    for (int c = 0; c < 4; ++c) avg[c] = (p1[c] + p2[c] + p3[c] + p4[c]) / 4;
    if (var) {
        double v = 0;
        for (int c = 0; c < 4; ++c)
            v += fabs(avg[c] - p1[c]) + fabs(avg[c] - p2[c]) + fabs(avg[c] - p3[c]) + fabs(avg[c] - p4[c]);
        *var = v;
    }
}

// resample_pixel (ndt.c:709-733): average of the four corners; pixels over the threshold become level-0 tasks
__global__ void k_aa_seed(const double *pass1, int width, int rows, int row_begin, int row_step, int row_pair, double threshold,
                          double *out, AaTask *tasks, int *counter)
{
    // a wavefront looks at an 8 x 8 tile of pixels, the tiles in row-major order: the tasks it appends -- and, roughly, those
    // of the wavefronts around it -- then lie together in the image and their samples' rays in one batch run through the same
    // part of the scene (pixel after pixel along an image row they came from every edge the row crosses)
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int tiles_x = (width + 7) / 8;
    const long long tile = tid >> 6;
    const int lane = (int)(tid & 63);
    const int i_raw = (int)(tile % tiles_x) * 8 + (lane & 7), l_raw = (int)(tile / tiles_x) * 8 + (lane >> 3);
    const bool in_range = i_raw < width && l_raw < rows;
    const int i = in_range ? i_raw : 0, l = in_range ? l_raw : 0;
    const long long idx = (long long)l * width + i;
    const int r0 = row_pair ? 2 * l : l, r1 = r0 + 1;
    const long long w1 = width + 1;
    const double *p1 = pass1 + (r0 * w1 + i) * 4, *p2 = p1 + 4;
    const double *p3 = pass1 + (r1 * w1 + i) * 4, *p4 = p3 + 4;
    double clr[4], var = 0.0;
    aa_avg4(p1, p2, p3, p4, clr, &var);
    const bool refine = in_range && var > threshold;
    const int t = wave_append(counter, refine);
    if (!in_range) return;
    if (refine) {
        AaTask &T = tasks[t];
        T.x = i;
        T.y = row_begin + l * row_step;
        for (int c = 0; c < 4; ++c) { T.p[0][c] = p1[c]; T.p[1][c] = p2[c]; T.p[2][c] = p3[c]; T.p[3][c] = p4[c]; }
        T.parent = idx;
        T.quad = -1;
    } else {
        for (int c = 0; c < 4; ++c) out[idx * 4 + c] = clr[c];
    }
}

// the five new samples of a task: centre, top middle, left edge, right edge, bottom middle (ndt.c:669-678)
__global__ void k_aa_samples(const AaTask *tasks, int n_tasks, double step, double *samples)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tasks) return;
    const double x = tasks[t].x, y = tasks[t].y, hs = step / 2;
    double *q = samples + (long long)t * 10;
    q[0] = x + hs;   q[1] = y + hs;
    q[2] = x + hs;   q[3] = y;
    q[4] = x;        q[5] = y + hs;
    q[6] = x + step; q[7] = y + hs;
    q[8] = x + hs;   q[9] = y + step;
}

// quarter averages and children (ndt.c:680-703)
__global__ void k_aa_split(AaTask *tasks, int n_tasks, double step, double threshold, const double *colours,
                           AaTask *next, int *next_counter)
{
    const int t_raw = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = t_raw < n_tasks;
    const int t = in_range ? t_raw : 0;          // (lanes past the end go through the motions on task 0 and write nothing)
    AaTask &T = tasks[t];
    const double *p5 = colours + (long long)t * 20, *p6 = p5 + 4, *p7 = p5 + 8, *p8 = p5 + 12, *p9 = p5 + 16;
    const double hs = step / 2;
    // corners of the four quarters, in image_avg_dbl_pixels4's argument order
    const double *qa[4][4] = { { T.p[0], p6, p7, p5 }, { T.p[1], p6, p8, p5 }, { T.p[2], p9, p7, p5 }, { T.p[3], p9, p8, p5 } };
    // ... and in the order the recursive call receives them as p1..p4 (ndt.c:687, 692, 697, 702)
    const double *qc[4][4] = { { T.p[0], p6, p7, p5 }, { p6, T.p[1], p5, p8 }, { p7, p5, T.p[2], p9 }, { p5, p8, p9, T.p[3] } };
    const double qx[4] = { T.x, T.x + hs, T.x, T.x + hs }, qy[4] = { T.y, T.y, T.y + hs, T.y + hs };
    // which quarters are refined; then ONE reservation for the workgroup (block_append_n), a task's children side by side
    unsigned refine_mask = 0;
    for (int k = 0; k < 4; ++k) {
        double var = 0.0, avg[4];
        aa_avg4(qa[k][0], qa[k][1], qa[k][2], qa[k][3], avg, &var);
        if (in_range)
            for (int ch = 0; ch < 4; ++ch) T.sp[k][ch] = avg[ch];
        if (in_range && var > threshold) refine_mask |= 1u << k;
    }
    int c = block_append_n(next_counter, __popc(refine_mask));
    for (int k = 0; k < 4; ++k) {
        if (refine_mask & (1u << k)) {
            AaTask &C = next[c++];
            C.x = qx[k];
            C.y = qy[k];
            for (int m = 0; m < 4; ++m)
                for (int ch = 0; ch < 4; ++ch) C.p[m][ch] = qc[k][m][ch];
            C.parent = t;
            C.quad = k;
        }
    }
}

// a task's colour goes to the quarter of its parent it refines, or (level 0) to its pixel.
// leaf != 0: the recursion's cut-off (ndt.c:663-666): the colour is the average of the corners
__global__ void k_aa_resolve(const AaTask *tasks, int n_tasks, int leaf, AaTask *parents, double *out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tasks) return;
    const AaTask &T = tasks[t];
    double res[4];
    if (leaf) aa_avg4(T.p[0], T.p[1], T.p[2], T.p[3], res, nullptr);
    else aa_avg4(T.sp[0], T.sp[1], T.sp[2], T.sp[3], res, nullptr);
    double *dst = parents ? parents[T.parent].sp[T.quad] : out + T.parent * 4;
    for (int c = 0; c < 4; ++c) dst[c] = res[c];
}

// the depth map beside an anti-aliased image (ndt.c:930-935, 753-756): the first pass's depths; the corner samples of the
// last column and row fall outside the width x height map (image.c:126 checks the bounds)
__global__ void k_aa_depth_crop(const double *pass1_depth, int width, int rows, int row_pair, double *out)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)rows * width) return;
    const int l = (int)(idx / width), i = (int)(idx % width);
    out[idx] = pass1_depth[(long long)(row_pair ? 2 * l : l) * (width + 1) + i];
}

// the first pass's corner samples as a position list: sample (r, i) of the (width) x (rows) grid is image position
// (i, row_begin + r * row_step), or with row pairs (row_begin + (r / 2) * row_step + r % 2)
__global__ void k_aa_corner_positions(double *pos, int width, int rows, int row_begin, int row_step, int row_pair)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)rows * width) return;
    const int r = (int)(idx / width), i = (int)(idx % width);
    pos[2 * idx] = i;
    pos[2 * idx + 1] = row_pair ? row_begin + (r >> 1) * row_step + (r & 1) : row_begin + r * row_step;
}

int ndt_impl::render_antialiased(ndt_hip_ctx *ctx, const ndt_render_params *p, void *d_rgba, ndt_render_stats &total, void *d_depth)
{
    hipStream_t s = ctx->stream;
    const bool prof = p->profile != 0;
    const int W = p->width, H = p->height;
    const int rows = ndt_hip_shard_rows(H, p->row_begin, p->row_step);
    AaBuffers buf(ctx);
    int rc;
    // One pass of the ray pipeline for this image's samples: render_pixel (ndt.c:578-653) for every one of them -- the image
    // split of side-by-side / over-under by the sample's position, or, for an anaglyph, BOTH eyes and their mix (ndt.c:636-647):
    // the anti-aliasing works on the mixed colours (its subdivision tests see red = left luminance, blue = right luminance).
    // With a lens (aperture_radius != 0), area lights or -n > 1 a sample is not one ray tree but get_pixel_color's whole
    // adaptive loop over random ones -- the reference samples the lens in this mode too (ndt.c:528: `recursive_aa != 0 ||
    // samples > 1`; the jitter stays off, ndt.c:505) -- and the anti-aliasing, its subdivision tests included, runs on those
    // noisy colours: a stochastic render.  The samples then go through the sampled renderer as a list of positions.
    const bool stochastic = ctx->aperture_radius != 0.0 || ctx->has_area_lights || p->samples > 1;
    auto pass = [&](RenderGeom g, double *out, long long n_out, double *depth_out, ndt_render_stats &st) -> int {
        st = ndt_render_stats{};
        if (stochastic) {
            int r;
            const double *pos = g.samples;
            if (!pos) {
                // the first pass's corner grid as positions
                double *grid = nullptr;
                if ((r = buf.get(&grid, (size_t)n_out * 2))) return r;
                hipLaunchKernelGGL(k_aa_corner_positions, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, grid, g.width, g.rows,
                                   g.row_begin, g.row_step, g.row_pair);
                pos = grid;
            }
            SampledList sl{ pos, n_out, g.img_w, g.img_h, g.aspect_w, g.aspect_h };
            if (p->stereo != NDT_STEREO_ANAGLYPH) return render_sampled_list(ctx, p, sl, 1, p->stereo, 0ull, out, st);
            double *left = nullptr, *right = nullptr;
            if ((r = buf.get(&left, (size_t)n_out * 4))) return r;
            if ((r = buf.get(&right, (size_t)n_out * 4))) return r;
            if ((r = render_sampled_list(ctx, p, sl, 0, 0, 0ull, left, st))) return r;
            if ((r = render_sampled_list(ctx, p, sl, 2, 0, 0x5eed0000000000ffull, right, st))) return r;
            launch_anaglyph(s, left, right, out, n_out);
            return NDT_OK;
        }
        if (p->stereo != NDT_STEREO_ANAGLYPH) {
            g.stereo = p->stereo;
            g.eye = 1;
            return render_pass(ctx, g, prof, out, st, depth_out);
        }
        double *left = nullptr, *right = nullptr;
        int r;
        if ((r = buf.get(&left, (size_t)n_out * 4))) return r;
        if ((r = buf.get(&right, (size_t)n_out * 4))) return r;
        ndt_render_stats one{};
        g.stereo = 0;
        g.eye = 0;
        if ((r = render_pass(ctx, g, prof, left, one, depth_out))) return r;        // (the depth map is the left eye's, ndt.c:640)
        add_stats(st, one);
        g.eye = 2;
        if ((r = render_pass(ctx, g, prof, right, one))) return r;
        add_stats(st, one);
        launch_anaglyph(s, left, right, out, n_out);
        return NDT_OK;
    };
    // ---- first pass: the corner rows this shard touches, (W+1) wide (ndt.c:919-976)
    RenderGeom g1{};
    g1.width = W + 1;
    g1.row_begin = p->row_begin;
    g1.row_step = p->row_step;
    g1.row_pair = p->row_step > 1 ? 1 : 0;
    g1.rows = g1.row_pair ? 2 * rows : rows + 1;
    g1.tiles_x = (g1.width + 7) / 8;
    g1.tiles_y = (g1.rows + 7) / 8;
    const long long n1 = (long long)g1.tiles_x * g1.tiles_y * 64;
    if (n1 > 0x3fffffffLL) return fail(NDT_E_UNSUPPORTED, "image too large for one call");
    g1.n_primary = (int)n1;
    g1.max_depth = p->max_optic_depth;
    g1.specular = p->specular ? 1 : 0;
    g1.img_w = W + 1;
    g1.img_h = H + 1;
    g1.aspect_w = W;
    g1.aspect_h = H;
    double *pass1 = nullptr, *pass1_depth = nullptr;
    if ((rc = buf.get(&pass1, (size_t)g1.rows * g1.width * 4))) return rc;
    if (d_depth && (rc = buf.get(&pass1_depth, (size_t)g1.rows * g1.width))) return rc;
    ndt_render_stats st{};
    if ((rc = pass(g1, pass1, (long long)g1.rows * g1.width, pass1_depth, st))) return rc;
    add_stats(total, st);
    if (d_depth)
        hipLaunchKernelGGL(k_aa_depth_crop, dim3((unsigned)(((long long)rows * W + 255) / 256)), dim3(256), 0, s, pass1_depth, W, rows,
                           g1.row_pair, (double *)d_depth);

    // ---- second pass
    const double threshold = p->aa_diff / 255.0;
    const long long n_out = (long long)rows * W;
    int *counters = nullptr;            // one per level
    const int max_levels = (p->aa_depth > 0 ? p->aa_depth : 0) + 2;
    if ((rc = buf.get(&counters, (size_t)max_levels + 1))) return rc;
    HIP_TRY(hipMemsetAsync(counters, 0, ((size_t)max_levels + 1) * sizeof(int), s));
    std::vector<AaTask *> level_tasks;
    std::vector<int> level_count;
    AaTask *t0 = nullptr;
    if ((rc = buf.get(&t0, (size_t)n_out))) return rc;
    hipLaunchKernelGGL(k_aa_seed, dim3((unsigned)((((long long)((W + 7) / 8) * ((rows + 7) / 8)) * 64 + 255) / 256)), dim3(256), 0, s, pass1, W, rows, p->row_begin, p->row_step,
                       g1.row_pair, threshold, (double *)d_rgba, t0, counters);
    int n_tasks = 0;
    HIP_TRY(hipMemcpyAsync(&n_tasks, counters, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    total.pixels_resampled = n_tasks;
    level_tasks.push_back(t0);
    level_count.push_back(n_tasks);
    double step = 1.0;
    int leaf_level = -1;
    for (int L = 0; n_tasks > 0; ++L) {
        if (L + 1 > max_levels) return fail(NDT_E_STATE, "anti-aliasing recursion deeper than expected");
        // recursive_resample's cut-off (ndt.c:663)
        if (p->aa_depth <= 0 || step < 1.0 / (double)(2 << (p->aa_depth - 1))) {
            leaf_level = L;
            break;
        }
        double *samples = nullptr, *colours = nullptr;
        AaTask *next = nullptr;
        if ((rc = buf.get(&samples, (size_t)n_tasks * 10))) return rc;
        if ((rc = buf.get(&colours, (size_t)n_tasks * 20))) return rc;
        if ((rc = buf.get(&next, (size_t)n_tasks * 4))) return rc;
        hipLaunchKernelGGL(k_aa_samples, dim3((unsigned)((n_tasks + 255) / 256)), dim3(256), 0, s, level_tasks[L], n_tasks, step, samples);
        RenderGeom gs{};
        gs.samples = samples;
        gs.n_samples = 5 * n_tasks;
        gs.n_primary = (gs.n_samples + 63) & ~63;
        gs.width = gs.n_samples;
        gs.rows = 1;
        gs.max_depth = p->max_optic_depth;
        gs.specular = p->specular ? 1 : 0;
        gs.img_w = W + 1;
        gs.img_h = H + 1;
        gs.aspect_w = W;
        gs.aspect_h = H;
        if ((rc = pass(gs, colours, gs.n_samples, nullptr, st))) return rc;
        add_stats(total, st);
        total.aa_samples += gs.n_samples;
        hipLaunchKernelGGL(k_aa_split, dim3((unsigned)((n_tasks + 255) / 256)), dim3(256), 0, s, level_tasks[L], n_tasks, step, threshold,
                           colours, next, counters + L + 1);
        int n_next = 0;
        HIP_TRY(hipMemcpyAsync(&n_next, counters + L + 1, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        level_tasks.push_back(next);
        level_count.push_back(n_next);
        n_tasks = n_next;
        step /= 2;
    }
    // ---- colours back up the tree, deepest level first
    for (int L = (int)level_tasks.size() - 1; L >= 0; --L) {
        const int n = level_count[L];
        if (n <= 0) continue;
        hipLaunchKernelGGL(k_aa_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, level_tasks[L], n, L == leaf_level ? 1 : 0,
                           L > 0 ? level_tasks[L - 1] : (AaTask *)nullptr, (double *)d_rgba);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return NDT_OK;
}

