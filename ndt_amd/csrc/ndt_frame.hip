// ndt_frame.hip -- the device workspace and ONE pass of the ray pipeline over a set of primaries: primary rays, the
// bounce loop, bottom-up resolve, per-primary colour.  render_image (ndt.c:900) for the deterministic path is one
// such pass; recursive anti-aliasing and the sampled paths call it once per level / round.
#include "ndt_ctx.hpp"
#include "ndt_finish.hpp"
#include <stddef.h>

static void free_stream_args(ndt_hip_ctx *ctx)
{
    for (void *p : ctx->sa_allocs) (void)hipFree(p);
    ctx->sa_allocs.clear();
    memset(&ctx->sa, 0, sizeof(ctx->sa));
    ctx->sa_cap = ctx->sa_sh_cap = 0;
    ctx->sa_nseg = 0;
}

void ndt_impl::free_workspace(ndt_hip_ctx *ctx)
{
    for (void *p : ctx->ws_allocs) (void)hipFree(p);
    ctx->ws_allocs.clear();
    free_stream_args(ctx);
    memset(&ctx->ws, 0, sizeof(ctx->ws));
    ctx->ws_slab_words = 0;
    ctx->ws_dims = 0;
    ctx->ws_nseg = 0;
}

// ------------------------------------------------------------------ workspace

template <typename T> static int ws_alloc(ndt_hip_ctx *ctx, T **p, size_t count, bool stream_args = false)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, (count > 0 ? count : 1) * sizeof(T));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(NDT_E_NOMEM, "hipMalloc of %zu bytes: %s", count * sizeof(T), hipGetErrorString(e));
    }
    (stream_args ? ctx->sa_allocs : ctx->ws_allocs).push_back(q);
    *p = (T *)q;
    return NDT_OK;
}

// The cooperative stragglers of one trace launch (TraceJob::coop_ring; ndt_device.hpp:coop_trace): scenes with item sets only.
void ndt_impl::coop_setup(ndt_hip_ctx *ctx, TraceJob &tj, unsigned int *log)
{
    tj.coop_ring = nullptr;
    if (!ctx->coop || ctx->tier != 0 || ctx->sd.mask_words != 1 || !ctx->ws.coop_ring) return;
    tj.coop_ring = ctx->ws.coop_ring;
    if (++ctx->coop_tag == 0u) ++ctx->coop_tag;         // (0 is what a cleared ring holds)
    tj.coop_tag = ctx->coop_tag;
    tj.coop_limit = NDT_COOP_RING_LIMIT;
    tj.coop_budget = ctx->coop_budget_us * 100;         // 100 MHz ticks
    tj.coop_max_live = ctx->coop_max_live;
    tj.coop_tail_only = ctx->coop_tail_only ? 1 : 0;
    tj.coop_waves = ctx->coop_waves;
    tj.coop_log = log;
}

int ndt_impl::ensure_workspace(ndt_hip_ctx *ctx, long long cap, long long sh_cap)
{
    Workspace &ws = ctx->ws;
    const bool need_slab = ctx->tier == 1;
    const long long slab_lanes = 2048LL * NDT_TRACE_BLOCK;
    const long long slab_words = need_slab ? slab_lanes * ctx->sd.mask_words : 0;
    const bool need_ring = ctx->tier == 0 && ctx->sd.mask_words == 1;
    if (ws.cap >= cap && ws.sh_cap >= sh_cap && ctx->ws_dims == ctx->dims && ctx->ws_slab_words >= slab_words &&
        ctx->ws_nseg >= ctx->n_shadow_lights && (ws.coop_ring != nullptr || !need_ring))
        return NDT_OK;
    if (cap < ws.cap) cap = ws.cap;
    if (sh_cap < ws.sh_cap) sh_cap = ws.sh_cap;
    cap = (cap + 63) & ~63LL;           // vectors are stored in tiles of 64 slots (load_soa / store_soa)
    sh_cap = (sh_cap + 63) & ~63LL;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_workspace(ctx);
    const int n = ctx->dims;
    int rc;
    // (an allocation that fails leaves no half-built workspace behind: the next call starts from nothing)
    auto give_up = [&](int code) {
        free_workspace(ctx);
        return code;
    };
    ws.cap = cap;
    ws.sh_cap = sh_cap;
    if ((rc = ws_alloc(ctx, &ws.ray_o, (size_t)n * cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.ray_v, (size_t)n * cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.frac, (size_t)cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.depth, (size_t)cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.rng_key, (size_t)cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.depth_left, (size_t)cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.hit_obj, (size_t)cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.hit_prim, (size_t)cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.hit_p, (size_t)n * cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.hit_n, (size_t)n * cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.clr, (size_t)3 * cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.child_refl, (size_t)cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.child_refr, (size_t)cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.sh_idx, (size_t)cap * (ctx->n_shadow_lights > 0 ? ctx->n_shadow_lights : 1)))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.sh_mask, (size_t)cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.count, (size_t)cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.so, (size_t)n * sh_cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.sv, (size_t)n * sh_cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.slim, (size_t)sh_cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.sobj, (size_t)sh_cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.sprim, (size_t)sh_cap))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.counters, NDT_CNT_TOTAL))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.ref_rays, 64 * 8))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.dbg, 160))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.exit_log, (size_t)NDT_EXIT_LOG_LAUNCHES * NDT_EXIT_LOG_WORDS))) return give_up(rc);
    if (ctx->shade_probe >= 0 && (rc = ws_alloc(ctx, &ws.shade_log, (size_t)2 * NDT_SHADE_LOG_WAVES))) return give_up(rc);
    if ((rc = ws_alloc(ctx, &ws.levels, NDT_MAX_LEVELS + 1))) return give_up(rc);
    if (ctx->tier == 0 && ctx->sd.mask_words == 1) {
        // the straggler ring of the trace launches (TraceJob::coop_ring): cleared once -- an entry counts when it carries its launch's tag
        if ((rc = ws_alloc(ctx, &ws.coop_ring, (size_t)NDT_COOP_RING_ENTRIES))) return give_up(rc);
        if (hipMemsetAsync(ws.coop_ring, 0, (size_t)NDT_COOP_RING_ENTRIES * sizeof(unsigned long long), ctx->stream) != hipSuccess)
            return give_up(fail(NDT_E_DEVICE, "clearing the straggler ring"));
    }
    ws.mask_slab_lanes = slab_lanes;
    if (need_slab) {
        if ((rc = ws_alloc(ctx, &ws.mask_slab, (size_t)slab_words))) return give_up(rc);
    }
    ctx->ws_slab_words = slab_words;
    ctx->ws_dims = ctx->dims;
    ctx->ws_nseg = ctx->n_shadow_lights > 0 ? ctx->n_shadow_lights : 1;
    return NDT_OK;
}

// The queues and counters of the streaming frame kernel for the current workspace: one fill counter, one lighting
// counter and one ring entry per node batch, the same per shadow batch of every light's segment, a parent and a
// wait count per node, an owner per shadow slot.  (They belong to a workspace and a light count: free_workspace releases
// them, and so does a change of either -- the previous set is freed, not kept until the workspace goes.)
static int ensure_stream_args(ndt_hip_ctx *ctx)
{
    const Workspace &ws = ctx->ws;
    const int n_seg = ctx->n_shadow_lights > 0 ? ctx->n_shadow_lights : 1;
    if (ctx->sa.ctl && ctx->sa_cap == ws.cap && ctx->sa_sh_cap == ws.sh_cap && ctx->sa_nseg == n_seg) return NDT_OK;
    if (!ctx->sa_allocs.empty()) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        free_stream_args(ctx);
    }
    StreamArgs &sa = ctx->sa;
    // rings have room for a ticket per wavefront beyond the last entry (a wavefront's ticket may name a slot that is never written)
    const long long margin = NDT_STREAM_LOG_WAVES;
    const long long node_batches = ws.cap / 64 + margin;
    const long long seg_cap = (ws.sh_cap / n_seg) & ~63LL;
    const long long sh_batches = (long long)n_seg * (seg_cap / 64) + margin;
    int rc;
    if ((rc = ws_alloc(ctx, &sa.ctl, 1, true))) return rc;
    if ((rc = ws_alloc(ctx, &sa.node_fill, (size_t)node_batches, true))) return rc;
    if ((rc = ws_alloc(ctx, &sa.sh_pending, (size_t)node_batches, true))) return rc;
    if ((rc = ws_alloc(ctx, &sa.sh_fill, (size_t)sh_batches, true))) return rc;
    if ((rc = ws_alloc(ctx, &sa.sec_ring, (size_t)NDT_PRIM_SHARDS * node_batches, true))) return rc;
    if ((rc = ws_alloc(ctx, &sa.sh_ring, (size_t)sh_batches, true))) return rc;
    if ((rc = ws_alloc(ctx, &sa.fin_ring, (size_t)node_batches, true))) return rc;
    if ((rc = ws_alloc(ctx, &sa.parent, (size_t)ws.cap, true))) return rc;
    if ((rc = ws_alloc(ctx, &sa.pend, (size_t)ws.cap, true))) return rc;
    if ((rc = ws_alloc(ctx, &sa.sowner, (size_t)n_seg * seg_cap, true))) return rc;
    if (ctx->stream_probe && (rc = ws_alloc(ctx, &sa.wave_log, (size_t)24 * NDT_STREAM_LOG_WAVES, true))) return rc;
    sa.n_seg = n_seg;
    sa.seg_cap = (int)seg_cap;
    sa.node_batches = (int)node_batches;
    ctx->sa_cap = ws.cap;
    ctx->sa_sh_cap = ws.sh_cap;
    ctx->sa_nseg = n_seg;
    return NDT_OK;
}

// ------------------------------------------------------------------ dimension-independent kernels

// Everything a frame needs reset, in one launch (five small copies / fills of 10 us each before): node tail and
// overflow flags, the work-queue heads of the launches the frame can have, both parities of the shadow-segment
// counters, the reference-ray partial sums, the diagnostic words, and the primaries' range.
__global__ void k_frame_init(Workspace ws, int n_primary, LevelRange level0, int queue_ints)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (int k = i; k < queue_ints; k += stride) ws.counters[NDT_CNT_QUEUE + k] = 0;
    for (int k = i; k < NDT_CNT_TOTAL - NDT_CNT_SEG; k += stride) ws.counters[NDT_CNT_SEG + k] = 0;
    for (int k = i; k < 64 * 8; k += stride) ws.ref_rays[k] = 0ull;
    for (int k = i; k < 160; k += stride) ws.dbg[k] = 0ull;
    if (i < 4) ws.counters[i] = (i == 0) ? n_primary : 0;
    if (i == 0) ws.levels[0] = level0;
}

// The frame's closing record, written to host-visible memory by the last kernel of the frame: the host polls its
// tag instead of queueing three small read-backs and synchronising the stream.  One wavefront.
//   [0] node tail  [1] overflow flags  [2] shadow slots wanted  [3] shadow rays of the frame  [4] bounces with nodes
//   [5] rays the reference would have traced  [7] tag
// Hybrid pipeline (`c` set): the frame kernel rendered the deeper bounces; its counts are added and its flags reported:
//   [8] its overflow flags  [9] its abort word  [10] nodes it spawned
__global__ void k_frame_done(Workspace ws, int n_run, unsigned long long *done, unsigned long long tag, const StreamCtl *c)
{
    const int lane = threadIdx.x;
    unsigned long long ref = ws.ref_rays[8 * lane];         // 64 partial sums, one 64-byte line each
    for (int d = 32; d > 0; d >>= 1) ref += __shfl_xor(ref, d, 64);
    if (lane != 0) return;
    long long shadow = 0;
    int used = 0;
    for (int b = 0; b < n_run; ++b) {
        if (ws.levels[b].count <= 0) break;
        shadow += ws.levels[b].n_shadow;
        ++used;
    }
    done[0] = (unsigned long long)(long long)ws.counters[0];
    done[1] = (unsigned long long)(long long)ws.counters[2];
    done[2] = (unsigned long long)(long long)ws.counters[3];
    done[8] = done[9] = done[10] = 0ull;
    if (c) {
        shadow += c->n_shadow.v;
        if (c->max_level.v + 1 > used) used = c->max_level.v + 1;
        done[8] = (unsigned long long)(long long)c->overflow.v;
        done[9] = (unsigned long long)(long long)(c->abort.v | (c->timeout_where.v << 8));
        done[10] = (unsigned long long)(long long)c->n_children.v;
    }
    done[3] = (unsigned long long)shadow;
    done[4] = (unsigned long long)used;
    done[5] = ref;
    __threadfence_system();
    __hip_atomic_store(&done[7], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The streaming pipeline's frame reset: control block, fill / lighting counters and rings of the batches the frame can
// have, the reference-ray partial sums.  (Sized by the pool, not by the frame: a few MB of zeros.)
// (frame_too: also the frame's own accumulators -- not in the hybrid pipeline, where the per-bounce kernels own them)
__global__ void __launch_bounds__(256) k_stream_init(Workspace ws, StreamArgs sa, long long node_batches, long long sh_batches, int frame_too)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    // the control block: zeros, except the node tail (the pool behind the primaries is free) and the primaries' share of
    // every shard's outstanding count -- written by the thread that owns the word, so that no zero can land on top of it
    int *ctl = reinterpret_cast<int *>(sa.ctl);
    const int nb0 = sa.root_begin >> 6, nb1 = nb0 + (sa.n_primary >> 6);       // the root batches
    for (long long k = i; k < (long long)(sizeof(StreamCtl) / sizeof(int)); k += stride) {
        int v = 0;
        if (k == (long long)(offsetof(StreamCtl, node_tail) / sizeof(int))) v = sa.root_begin + sa.n_primary;
        for (int sh = 0; sh < NDT_PRIM_SHARDS; ++sh)
            if (k == (long long)((offsetof(StreamCtl, outstanding) + sh * sizeof(StreamWord)) / sizeof(int))) {
                // root batches nb in [nb0, nb1) with nb % 8 == sh
                const int first = nb0 + ((sh - nb0 % NDT_PRIM_SHARDS) + NDT_PRIM_SHARDS) % NDT_PRIM_SHARDS;
                v = first < nb1 ? 128 * ((nb1 - 1 - first) / NDT_PRIM_SHARDS + 1) : 0;
            }
        ctl[k] = v;
    }
    for (long long k = i; k < node_batches; k += stride) {
        sa.node_fill[k] = 0;
        sa.sh_pending[k] = 0;
        for (int sh = 0; sh < NDT_PRIM_SHARDS; ++sh) sa.sec_ring[(long long)sh * node_batches + k] = 0;
        sa.fin_ring[k] = 0;
    }
    for (long long k = i; k < sh_batches; k += stride) {
        sa.sh_fill[k] = 0;
        sa.sh_ring[k] = 0;
    }
    if (frame_too) {
        for (long long k = i; k < 64 * 8; k += stride) ws.ref_rays[k] = 0ull;
        for (long long k = i; k < 4; k += stride) ws.counters[k] = 0;
    }
}

// The streaming pipeline's closing record, in host-visible memory (the host polls the tag):
//   [0] node tail  [1] overflow flags  [2] abort  [3] shadow rays  [4] deepest bounce + 1  [5] reference-equivalent rays
//   [6] children  [7] tag
__global__ void k_stream_done(Workspace ws, StreamArgs sa, unsigned long long *done, unsigned long long tag)
{
    const int lane = threadIdx.x;
    unsigned long long ref = ws.ref_rays[8 * lane];
    for (int d = 32; d > 0; d >>= 1) ref += __shfl_xor(ref, d, 64);
    if (lane != 0) return;
    const StreamCtl *c = sa.ctl;
    done[0] = (unsigned long long)(long long)c->node_tail.v;
    done[1] = (unsigned long long)(long long)c->overflow.v;
    done[2] = (unsigned long long)(long long)(c->abort.v | (c->timeout_where.v << 8));
    done[3] = (unsigned long long)(long long)c->n_shadow.v;
    done[4] = (unsigned long long)(long long)(c->max_level.v + 1);
    done[5] = ref;
    done[6] = (unsigned long long)(long long)c->n_children.v;
    __threadfence_system();
    __hip_atomic_store(&done[7], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Bottom-up combine of one bounce: get_ray_color's blend of its own colour with the colours
// its reflection / refraction children returned (ndt.c:402-429), in the reference's order.
__device__ __forceinline__ void resolve_node(const double *blob, const SceneDesc &sd, const Workspace &ws, int specular, long long g)
{
    if (ws.depth_left[g] <= 0) return;
    const int obj = ws.hit_obj[g];
    if (obj < 0) return;                        // background node: colour and count already final
    const int mw = sd.off_mat + 8 * obj;
    const double hitr[3] = { blob[mw + 3], blob[mw + 4], blob[mw + 5] };
    double c[3] = { ws.clr[0 * ws.cap + g], ws.clr[1 * ws.cap + g], ws.clr[2 * ws.cap + g] };
    int cnt = ws.count[g];
    const int refl = ws.child_refl[g];
    if (refl != -1) {
        double ref[3] = { 0.0, 0.0, 0.0 };
        if (refl >= 0) {
            ref[0] = ws.clr[0 * ws.cap + refl]; ref[1] = ws.clr[1 * ws.cap + refl]; ref[2] = ws.clr[2 * ws.cap + refl];
            cnt += ws.count[refl];
        }
        for (int k = 0; k < 3; ++k) {
            if (specular) c[k] = (1 - hitr[k]) * (c[k]) + (hitr[k]) * ref[k];     // ndt.c:405-407
            else c[k] += hitr[k] * ref[k];                                         // ndt.c:411-413
        }
    }
    const int refr = ws.child_refr[g];
    if (refr != -1) {
        double ref[3] = { 0.0, 0.0, 0.0 };
        if (refr >= 0) {
            ref[0] = ws.clr[0 * ws.cap + refr]; ref[1] = ws.clr[1 * ws.cap + refr]; ref[2] = ws.clr[2 * ws.cap + refr];
            cnt += ws.count[refr];
        }
        for (int k = 0; k < 3; ++k) c[k] += (1.0 - hitr[k]) * ref[k];              // ndt.c:426-428
    }
    ws.clr[0 * ws.cap + g] = c[0];
    ws.clr[1 * ws.cap + g] = c[1];
    ws.clr[2 * ws.cap + g] = c[2];
    ws.count[g] = cnt;
}

__global__ void __launch_bounds__(256) k_resolve(const double *blob, SceneDesc sd, Workspace ws, int specular, int level)
{
    const LevelRange lr = ws.levels[level];
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < lr.count; r += (long long)gridDim.x * blockDim.x)
        resolve_node(blob, sd, ws, specular, lr.begin + r);
}

// (the pixel itself: ndt_finish.hpp)
// resolve0: the bottom-up combine of the primaries' own bounce (the last k_resolve) happens here, in the thread that then
// finishes the pixel: one launch less at the end of a frame
#ifndef NDT_FINISH_BLOCK
#define NDT_FINISH_BLOCK 256
#endif
__global__ void __launch_bounds__(NDT_FINISH_BLOCK) k_finish_pixels(const double *blob, SceneDesc sd, Workspace ws, RenderGeom rg, int N_,
                                                       double *rgba, double *depth_out, int resolve0)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long weighted = 0ull;
    if (resolve0 && g < rg.n_primary) resolve_node(blob, sd, ws, rg.specular, g);
    if (g < rg.n_primary && ws.depth_left[g] > 0) weighted = finish_pixel<false>(blob, sd, ws, rg, N_, g, rgba, depth_out);
    // wavefront sum, then one atomic per wavefront spread over 64 cache lines (a single word
    // saturates near 90 atomics/us, and there are 32k wavefronts at 1080p)
    for (int d = 32; d > 0; d >>= 1) weighted += __shfl_down(weighted, d, 64);
    if ((threadIdx.x & 63) == 0 && weighted) atomicAdd(ws.ref_rays + 8 * ((blockIdx.x * (NDT_FINISH_BLOCK / 64) + (threadIdx.x >> 6)) & 63), weighted);
}

// max_optic_depth <= 0: get_ray_color returns black without tracing (ndt.c:340)
__global__ void k_fill_black(double *rgba, long long n_pixels)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    rgba[4 * i + 0] = 0.0; rgba[4 * i + 1] = 0.0; rgba[4 * i + 2] = 0.0; rgba[4 * i + 3] = 1.0;
}

// pixel_d2c, image.h:36-39
void ndt_impl::launch_fill_black(hipStream_t s, double *rgba, long long n_pixels)
{
    hipLaunchKernelGGL(k_fill_black, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, s, rgba, n_pixels);
}

void ndt_impl::add_stats(ndt_render_stats &acc, const ndt_render_stats &st)
{
    acc.rays_primary += st.rays_primary;
    acc.rays_secondary += st.rays_secondary;
    acc.rays_shadow += st.rays_shadow;
    acc.rays_ref_equiv += st.rays_ref_equiv;
    if (st.levels > acc.levels) acc.levels = st.levels;
    acc.trace_launches += st.trace_launches;
    acc.trace_ms += st.trace_ms;
    acc.frame_ms += st.frame_ms;
    if (st.node_capacity > acc.node_capacity) acc.node_capacity = st.node_capacity;
}

// ------------------------------------------------------------------ render

// NDT_HIP_STREAM_PROBE: what every wavefront of the frame kernel did and when (100 MHz ticks)
static void print_stream_probe(const unsigned int *wave_log, float km)
{
                    std::vector<unsigned int> log((size_t)24 * NDT_STREAM_LOG_WAVES);
                    if (hipMemcpy(log.data(), wave_log, log.size() * sizeof(unsigned int), hipMemcpyDeviceToHost) == hipSuccess) {
                        unsigned long long n[4] = { 0, 0, 0, 0 }, t[3] = { 0, 0, 0 }, parts[5] = { 0, 0, 0, 0, 0 };
                        int waves = 0, busy_waves = 0;
                        unsigned int t0 = 0, max_items = 0;
                        bool any = false;
                        for (int w = 0; w < NDT_STREAM_LOG_WAVES; ++w) {
                            const unsigned int *q = &log[(size_t)24 * w];
                            if (!q[9]) continue;
                            if (!any || (int)(q[10] - t0) < 0) t0 = q[10];
                            any = true;
                        }
                        double first_item = 1e30, last_item = 0, last_exit = 0, start_spread = 0;
                        int hist[32] = { 0 };
                        for (int w = 0; w < NDT_STREAM_LOG_WAVES; ++w) {
                            const unsigned int *q = &log[(size_t)24 * w];
                            if (!q[9]) continue;
                            ++waves;
                            const unsigned int items = q[0] + q[1] + q[2];
                            if (items) ++busy_waves;
                            if (items > max_items) max_items = items;
                            for (int k = 0; k < 4; ++k) n[k] += q[k];
                            for (int k = 0; k < 3; ++k) t[k] += q[4 + k];
                            for (int k = 0; k < 5; ++k) parts[k] += q[12 + k];
                            const double off = (q[10] - t0) / 100.0;
                            if (off > start_spread) start_spread = off;
                            if (q[7] && off + q[7] / 100.0 < first_item) first_item = off + q[7] / 100.0;
                            if (off + q[8] / 100.0 > last_item) last_item = off + q[8] / 100.0;
                            if (off + q[9] / 100.0 > last_exit) last_exit = off + q[9] / 100.0;
                            int bin = (int)((off + q[8] / 100.0) / (km * 1000.0 / 32.0 + 1e-9));
                            ++hist[bin < 0 ? 0 : bin > 31 ? 31 : bin];
                        }
                        std::string line;
                        for (int b = 0; b < 32; ++b) {
                            char buf[16];
                            snprintf(buf, sizeof buf, " %d", hist[b]);
                            line += buf;
                        }
                        fprintf(stderr, "ndt_hip: frame kernel %.3f ms: %d wavefronts (%d with work, at most %u items each) started within %.1f us; "
                                        "node batches %llu (%.1f us each), shadow batches %llu (%.1f us each), lighting batches %llu (%.1f us each), "
                                        "idle rounds %llu; first item at %.1f us, last item done at %.1f us, last exit at %.1f us; "
                                        "wavefronts by the 32nd of the kernel in which they finished their last item:%s\n",
                                km, waves, busy_waves, max_items, start_spread, n[0], n[0] ? t[0] / 100.0 / n[0] : 0.0, n[1],
                                n[1] ? t[1] / 100.0 / n[1] : 0.0, n[2], n[2] ? t[2] / 100.0 / n[2] : 0.0, n[3], first_item, last_item, last_exit,
                                line.c_str());
                        fprintf(stderr, "ndt_hip:    per item: looking for work %.1f us (all kinds); node + shadow items: loading the rays %.1f us; trace_kd: node "
                                        "batches %.1f us, shadow batches %.1f us; colours up the tree + counters %.1f us (node and lighting batches)\n",
                                (n[0] + n[1] + n[2]) ? parts[0] / 100.0 / (n[0] + n[1] + n[2]) : 0.0,
                                (n[0] + n[1]) ? parts[1] / 100.0 / (n[0] + n[1]) : 0.0, n[0] ? (parts[2] - parts[3]) / 100.0 / n[0] : 0.0,
                                n[1] ? parts[3] / 100.0 / n[1] : 0.0, (n[0] + n[2]) ? parts[4] / 100.0 / (n[0] + n[2]) : 0.0);
                    }
}


static hipEvent_t get_event(ndt_hip_ctx *ctx, size_t idx)
{
    while (ctx->ev_pool.size() <= idx) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return nullptr;
        ctx->ev_pool.push_back(ev);
    }
    return ctx->ev_pool[idx];
}

static double wall_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// One pass of the ray pipeline over the primaries `rg` describes: primary rays, the bounce loop,
// bottom-up resolve, per-primary colour (k_finish_pixels) into d_rgba.  Grid mode writes a
// rows x width image, list mode one RGBA per sample.  max_depth > 0 (the callers handle -l 0).
int ndt_impl::render_pass(ndt_hip_ctx *ctx, RenderGeom rg, bool prof, void *d_rgba, ndt_render_stats &st, void *d_depth)
{
    rg.want_depth = d_depth ? 1 : 0;
    hipStream_t s = ctx->stream;
    const long long n_primary = rg.n_primary;
    const long long n_pixels = rg.samples ? (long long)rg.n_samples : (long long)rg.rows * rg.width;
    long long cap = ctx->ws.cap, sh_cap = ctx->ws.sh_cap;
    if (cap < 2 * n_primary + 4096) cap = 2 * n_primary + 4096;
    const long long want_sh = n_primary * (ctx->n_shadow_lights > 0 ? ctx->n_shadow_lights : 1) + 4096;
    if (sh_cap < want_sh) sh_cap = want_sh;
    if (ctx->test_small_pool && ctx->ws.cap == 0) {
        // tests only: a fresh context starts with a node pool that a reflective scene overflows, so that the
        // overflow -> grow -> render-again path below is exercised (tests/test_gpu_parity.py)
        cap = ((n_primary + 63) & ~63LL) + 64;
    }

    const NdtKernelTable *kt = ctx->kt;
    // (the gate prepass of trace_kd: the scene description a pass's kernels get carries the gated items only when it is on)
    SceneDesc sd_pass = ctx->sd;
    if (ctx->gate_prepass == 0 || (ctx->gate_prepass == 2 && n_primary > ctx->gate_prepass_below)) sd_pass.gate_bits = 0ull;
    // (a pass over a LIST of image positions -- recursive anti-aliasing's samples, all of them on the edges the first pass found:
    // 15 rays a sample where a frame has 2.4 a pixel -- crosses over much earlier: profiles/r03_modes_1080p.txt)
    const long long stream_upto = rg.samples ? ctx->stream_below_list : ctx->stream_below;
    ctx->use_stream = ctx->pipeline == 2 || (ctx->pipeline == 0 && n_primary <= stream_upto);
    const bool hybrid = !ctx->use_stream && ctx->pipeline == 3 && ctx->hybrid_level >= 1 && rg.max_depth > ctx->hybrid_level;
    // (auto only) the frame kernel keeps one shadow slot per node AND light for the whole frame: with many lights that can
    // exceed what the per-bounce pipeline, which sizes its segments bounce by bounce, needs by far.  When it does not fit --
    // 2^31 slots, or the allocation fails -- auto renders the pass per bounce instead of failing.
    const long long cap_levels = cap, sh_cap_levels = sh_cap;
    bool stream_gave_up = false;
    if (ctx->use_stream) {
        // ---- the streaming pipeline: one persistent launch for the whole ray tree (ndt_stream.hpp)
        const int n_seg = ctx->n_shadow_lights > 0 ? ctx->n_shadow_lights : 1;
        for (int attempt = 0; attempt < 8; ++attempt) {
            // every light's shadow segment can hold one ray per node
            if (sh_cap < cap * n_seg) sh_cap = cap * n_seg;
            int rc = NDT_OK;
            if (cap > 0x7fffff00LL || sh_cap > 0x7fffff00LL) rc = fail(NDT_E_NOMEM, "ray tree exceeds 2^31 nodes");
            if (!rc) rc = ensure_workspace(ctx, cap, sh_cap);
            if (!rc) rc = ensure_stream_args(ctx);
            if (rc == NDT_E_NOMEM && ctx->pipeline == 0) {
                stream_gave_up = true;
                break;
            }
            if (rc) return rc;
            Workspace ws = ctx->ws;
            StreamArgs sa = ctx->sa;
            sa.root_begin = 0;
            sa.n_primary = rg.n_primary;
            sa.roots_are_primaries = 1;
            sa.valid_begin = 0;
            sa.valid_end = rg.n_primary;
            // the kernel makes its own primaries and writes its own pixels (option stream_fused, on by default)
            sa.fused = ctx->stream_fused ? 1 : 0;
            sa.rgba = (double *)d_rgba;
            sa.depth_out = (double *)d_depth;
            unsigned int *wave_log = sa.wave_log;
            if (!prof) sa.wave_log = nullptr;
            else if (wave_log) HIP_TRY(hipMemsetAsync(wave_log, 0, (size_t)24 * NDT_STREAM_LOG_WAVES * sizeof(unsigned int), s));
            hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_k0 = nullptr, ev_k1 = nullptr;
            if (prof) {
                ev_begin = get_event(ctx, 0);
                ev_end = get_event(ctx, 1);
                ev_k0 = get_event(ctx, 2);
                ev_k1 = get_event(ctx, 3);
            }
            const unsigned long long tag = ++ctx->frame_tag;
            const long long node_batches = sa.node_batches, sh_batches = (long long)sa.n_seg * (sa.seg_cap / 64) + NDT_STREAM_LOG_WAVES;
            if (prof)
                hipExtLaunchKernelGGL(k_stream_init, dim3(512), dim3(256), 0, s, ev_begin, nullptr, 0u, ws, sa, node_batches, sh_batches, 1);
            else
                hipLaunchKernelGGL(k_stream_init, dim3(512), dim3(256), 0, s, ws, sa, node_batches, sh_batches, 1);
            if (!sa.fused) kt->primary(s, ctx->d_blob, sd_pass, ws, rg);
            kt->frame_stream(s, ctx->d_blob, sd_pass, ws, rg, sa, ctx->tier, ctx->sd.mask_words, ev_k0, ev_k1);
            if (!sa.fused)
                hipLaunchKernelGGL(k_finish_pixels, dim3((unsigned)((rg.n_primary + NDT_FINISH_BLOCK - 1) / NDT_FINISH_BLOCK)), dim3(NDT_FINISH_BLOCK), 0, s, ctx->d_blob, sd_pass, ws,
                                   rg, ctx->dims, (double *)d_rgba, (double *)d_depth, 0);
            if (prof)
                hipExtLaunchKernelGGL(k_stream_done, dim3(1), dim3(64), 0, s, nullptr, ev_end, 0u, ws, sa, ctx->d_done, tag);
            else
                hipLaunchKernelGGL(k_stream_done, dim3(1), dim3(64), 0, s, ws, sa, ctx->d_done, tag);
            HIP_TRY(hipGetLastError());
            {
                const double t_wait = wall_s();
                while (__atomic_load_n(&ctx->h_done[7], __ATOMIC_ACQUIRE) != tag) {
                    if (wall_s() - t_wait > 30.0) {
                        HIP_TRY(hipStreamSynchronize(s));
                        if (__atomic_load_n(&ctx->h_done[7], __ATOMIC_ACQUIRE) != tag) return fail(NDT_E_STATE, "the frame never completed");
                    }
                }
            }
            if (prof) HIP_TRY(hipEventSynchronize(ev_end));
            const int overflow = (int)(long long)ctx->h_done[1], aborted = (int)(long long)ctx->h_done[2];
            if (overflow != 0) {
                if (overflow & 1) cap *= 2;
                if (overflow & 2) sh_cap *= 2;
                continue;
            }
            if (aborted != 0)
                return fail(NDT_E_STATE, "the frame kernel gave up (abort %d, where %d): a work item never arrived", aborted & 0xff, aborted >> 8);
            st = ndt_render_stats{};
            st.rays_primary = n_pixels;
            st.rays_secondary = (long long)ctx->h_done[6];
            st.rays_shadow = (long long)ctx->h_done[3];
            st.rays_ref_equiv = (long long)ctx->h_done[5];
            st.levels = (int)ctx->h_done[4];
            st.trace_launches = 1;
            st.node_capacity = ws.cap;
            if (prof) {
                float km = 0, fm = 0;
                HIP_TRY(hipEventElapsedTime(&km, ev_k0, ev_k1));
                HIP_TRY(hipEventElapsedTime(&fm, ev_begin, ev_end));
                st.trace_ms = km;
                st.frame_ms = fm;
                if (wave_log) print_stream_probe(wave_log, km);
            }
            return NDT_OK;
        }
        if (!stream_gave_up) return fail(NDT_E_NOMEM, "ray-tree workspace kept overflowing");
        ctx->use_stream = false;
        cap = cap_levels;
        sh_cap = sh_cap_levels;
        if (cap < ctx->ws.cap) cap = ctx->ws.cap;
        if (sh_cap < ctx->ws.sh_cap) sh_cap = ctx->ws.sh_cap;
    }
    for (int attempt = 0; attempt < 8; ++attempt) {
        if (cap > 0x7fffff00LL || sh_cap > 0x7fffff00LL) return fail(NDT_E_NOMEM, "ray tree exceeds 2^31 nodes");
        int rc = ensure_workspace(ctx, cap, sh_cap);
        if (rc) return rc;
        if (hybrid && (rc = ensure_stream_args(ctx))) return rc;
        Workspace ws = ctx->ws;
        size_t ev_n = 0;
        hipEvent_t ev_begin = nullptr, ev_end = nullptr;
        std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_ev;
        std::vector<std::string> trace_dbg;
        if (prof) {
            // frame time = start of the frame's first kernel .. end of its last (their own dispatch timestamps)
            ev_begin = get_event(ctx, ev_n++);
            ev_end = get_event(ctx, ev_n++);
        }
        int *hc = ctx->h_counters;
        if (prof && ctx->exit_probe)
            HIP_TRY(hipMemsetAsync(ws.exit_log, 0, (size_t)NDT_EXIT_LOG_LAUNCHES * NDT_EXIT_LOG_WORDS * sizeof(unsigned int), s));
        // The stream is never synchronised inside a frame: the range of every bounce is published
        // on the device (k_level_step) and read there; the host only learns, from the mailbox,
        // whether there is a next bounce to enqueue.  Bounce 0 = the primaries.
        ws.mail = ctx->d_mail;
        ws.mail_tag = ctx->d_mail_tag;
        const unsigned long long tag = ++ctx->frame_tag;
        const int n_seg = ctx->n_shadow_lights;
        const int n_levels = rg.max_depth > 1 ? rg.max_depth : 1;      // a node spawns children only while depth_left > 1
        int n_run = n_levels;                                           // bounces actually enqueued
        LevelRange *hl = ctx->h_levels;
        hl[0].begin = 0;
        hl[0].count = rg.n_primary;
        hl[0].seg_stride = (rg.n_primary + 63) & ~63LL;
        hl[0].n_shadow = 0;
        if ((long long)n_seg * hl[0].seg_stride > ws.sh_cap) {
            sh_cap = (long long)n_seg * hl[0].seg_stride;
            continue;
        }
        {
            int slots = n_levels + 2;               // one trace launch per bounce + the primaries' own
            if (slots > NDT_QUEUE_SLOTS) slots = NDT_QUEUE_SLOTS;
            if (prof)
                hipExtLaunchKernelGGL(k_frame_init, dim3(8), dim3(256), 0, s, ev_begin, nullptr, 0u, ws, rg.n_primary, hl[0], slots * NDT_QUEUE_INTS);
            else
                hipLaunchKernelGGL(k_frame_init, dim3(8), dim3(256), 0, s, ws, rg.n_primary, hl[0], slots * NDT_QUEUE_INTS);
        }
        int queue_slot = 0;
        int launches = 0;
        // (the variant is built for the planar camera: VR and panorama frames take k_primary)
        const bool fuse_primaries = (ctx->fuse_primaries < 0 ? ctx->dims >= 4 : ctx->fuse_primaries != 0) && !ctx->coop && ctx->cam_type == 0;
        // (no k_primary: the first trace launch makes the primaries it traces, TraceJob::make_primaries)
        if (!fuse_primaries) kt->primary(s, ctx->d_blob, sd_pass, ws, rg);
        auto traced = [&](TraceJob &tj, const std::string &what) -> int {
            tj.queue = ws.counters + NDT_CNT_QUEUE + (queue_slot++) * NDT_QUEUE_INTS;
            const bool exit_probe = ctx->exit_probe;
            tj.exit_log = (exit_probe && prof && launches < NDT_EXIT_LOG_LAUNCHES) ? ws.exit_log + (size_t)launches * NDT_EXIT_LOG_WORDS : nullptr;
            coop_setup(ctx, tj, tj.exit_log ? reinterpret_cast<unsigned int *>(ws.dbg + 100 + 2 * launches) : nullptr);
            if (prof) {
                hipEvent_t a = get_event(ctx, ev_n++), b2 = get_event(ctx, ev_n++);
                kt->trace(s, ctx->d_blob, sd_pass, ws, tj, ctx->tier, ctx->sd.mask_words, a, b2);
                trace_ev.push_back({ a, b2 });
                trace_dbg.push_back(what);
            } else {
                kt->trace(s, ctx->d_blob, sd_pass, ws, tj, ctx->tier, ctx->sd.mask_words, nullptr, nullptr);
            }
            ++launches;
            return NDT_OK;
        };
        // closest-hit queries of the primaries: the only launch that is not shared
        {
            TraceJob tj{};
            tj.n_seg = 0;
            tj.dense.o = ws.ray_o; tj.dense.v = ws.ray_v; tj.dense.stride = ws.cap; tj.dense.lim = nullptr;
            tj.dense.valid = ws.depth_left; tj.dense.out_obj = ws.hit_obj; tj.dense.out_prim = ws.hit_prim;
            tj.begin = 0; tj.count = rg.n_primary; tj.levels = nullptr;
            tj.publish_level = -1;
            if (fuse_primaries) {
                tj.make_primaries = 1;
                tj.rg = rg;
            }
            if ((rc = traced(tj, "primaries + closest 0"))) return rc;
        }
        long long upper = rg.n_primary;         // node count of the bounce
        std::vector<long long> level_nodes;
        // NDT_HIP_SHADE_PROBE=<k>: the k-th shade launch of the frame logs the life of each of its wavefronts
        const int shade_probe = ctx->shade_probe;
        int shade_launch = 0;
        long long shade_probe_finish_waves = 0;         // wavefronts of the lighting part of the probed launch
        long long shade_probe_emit_waves = 0;           // ... and of the shading part behind it (pair launches)
        auto shade_ws = [&](long long finish_nodes, long long emit_nodes_behind = 0) {
            Workspace w = ws;
            if (shade_launch++ != shade_probe || !prof) {
                w.shade_log = nullptr;
            } else {
                shade_probe_finish_waves = (finish_nodes + 255) / 256 * 4;
                shade_probe_emit_waves = (emit_nodes_behind + 255) / 256 * 4;
                (void)hipMemsetAsync(w.shade_log, 0, (size_t)2 * NDT_SHADE_LOG_WAVES * sizeof(unsigned int), s);
            }
            return w;
        };
        const bool fuse_shade = ctx->shade_pair;
        int pending_finish = -1;                // bounce whose lighting has not been launched yet
        long long pending_upper = 0;
        // Hybrid pipeline: the first `hand` bounces -- where the rays are -- go through the per-bounce kernels (three
        // wavefronts per SIMD in the trace kernel, shade kernels with the whole chip's wavefront slots); the deeper
        // bounces, a few per cent of the rays but a launch latency each (one slow batch: 0.15-0.2 ms per trace launch
        // and 75 us per shade launch, three times over on the benchmark frame), are ONE launch of the frame kernel
        // rooted at the nodes of bounce `hand`.
        const int hand = hybrid ? ctx->hybrid_level : n_levels + 1;
        bool streamed = false;
        for (int b = 0; b < n_levels && b < hand; ++b) {
            if (queue_slot + 1 > NDT_QUEUE_SLOTS || b + 1 > NDT_MAX_LEVELS)
                return fail(NDT_E_UNSUPPORTED, "more than %d bounces", NDT_QUEUE_SLOTS - 1);
            if (b > 0) {
                // posted by k_level_step(b-1), which ran right after shade_emit(b-1)
                const double t_wait = wall_s();
                while (__atomic_load_n(&ctx->h_mail_tag[b], __ATOMIC_ACQUIRE) != tag) {
                    if (wall_s() - t_wait > 30.0) {
                        HIP_TRY(hipStreamSynchronize(s));
                        if (__atomic_load_n(&ctx->h_mail_tag[b], __ATOMIC_ACQUIRE) != tag) return fail(NDT_E_STATE, "bounce %d was never published", b);
                    }
                }
                upper = ctx->h_mail[b].count;
                if (upper <= 0) {
                    n_run = b;
                    break;
                }
            }
            level_nodes.push_back(upper);
            // hit points, shadow rays of this bounce, and the rays of the next bounce -- in the same launch as the
            // lighting of the previous bounce, which is waiting for the shadow answers the last trace launch produced
            if (pending_finish >= 0 && fuse_shade) {
                kt->shade_pair(s, ctx->d_blob, sd_pass, shade_ws(pending_upper, upper), rg, pending_finish, pending_upper, upper);
                pending_finish = -1;
            } else {
                if (pending_finish >= 0) {
                    kt->shade_finish(s, ctx->d_blob, sd_pass, shade_ws(pending_upper), rg, pending_finish, pending_upper, 0);
                    pending_finish = -1;
                }
                kt->shade_emit(s, ctx->d_blob, sd_pass, shade_ws(0), rg, b, upper);
            }
            // (no k_level_step: the trace launch below publishes bounce b + 1, TraceJob::publish_level)
            long long next_upper = 2 * upper;           // each node spawns at most two
            if (next_upper > ws.cap) next_upper = ws.cap;
            {
                // ONE launch: shadow rays of bounce b + closest-hit rays of bounce b+1
                TraceJob tj{};
                tj.n_seg = n_seg;
                tj.seg.o = ws.so; tj.seg.v = ws.sv; tj.seg.stride = ws.sh_cap; tj.seg.lim = ws.slim; tj.seg.valid = nullptr;
                tj.seg_light_origins = sd_pass.light_origins;      // (what shade_emit_node left out: ndt_kernels.hip)
                tj.seg.out_obj = ws.sobj; tj.seg.out_prim = ws.sprim;
                tj.seg_count = NDT_SEG_COUNTERS(ws, b);
                tj.seg_stride = (upper + 63) & ~63LL;           // sizes the grid only
                tj.dense.o = ws.ray_o; tj.dense.v = ws.ray_v; tj.dense.stride = ws.cap; tj.dense.lim = nullptr;
                tj.dense.valid = ws.depth_left; tj.dense.out_obj = ws.hit_obj; tj.dense.out_prim = ws.hit_prim;
                tj.begin = 0;
                tj.count = next_upper;                          // sizes the grid only
                tj.levels = ws.levels; tj.seg_level = b; tj.dense_level = (b + 1 == hand) ? -1 : b + 1;     // the frame kernel traces bounce `hand`
                tj.publish_level = b;
                tj.publish_tag = tag;
                if ((rc = traced(tj, "shadow " + std::to_string(b) + (b + 1 == hand ? "" : " + closest " + std::to_string(b + 1))))) return rc;
            }
            pending_finish = b;
            pending_upper = upper;
        }
        // the lighting of the deepest bounce that has nodes: blended on the spot (its nodes have no child nodes), unless the frame
        // kernel renders deeper bounces behind it (hybrid)
        const bool resolve_with_finish = pending_finish >= 0 && !hybrid && pending_finish >= 1 && pending_finish == n_run - 1;
        if (pending_finish >= 0)
            kt->shade_finish(s, ctx->d_blob, sd_pass, shade_ws(pending_upper), rg, pending_finish, pending_upper, resolve_with_finish ? 1 : 0);
        StreamArgs sa = ctx->sa;
        hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;
        if (hybrid && n_run >= hand && hand < n_levels) {
            // the nodes of bounce `hand`: posted by k_level_step(hand - 1)
            const double t_wait = wall_s();
            while (__atomic_load_n(&ctx->h_mail_tag[hand], __ATOMIC_ACQUIRE) != tag) {
                if (wall_s() - t_wait > 30.0) {
                    HIP_TRY(hipStreamSynchronize(s));
                    if (__atomic_load_n(&ctx->h_mail_tag[hand], __ATOMIC_ACQUIRE) != tag) return fail(NDT_E_STATE, "bounce %d was never published", hand);
                }
            }
            const LevelRange roots = ctx->h_mail[hand];
            if (roots.count > 0) {
                // the root range = the 64-aligned cover of the bounce's node range (the frame kernel works in batches of
                // 64 slots); its queues are reset, and it runs
                sa.root_begin = (int)(roots.begin & ~63LL);
                sa.n_primary = (int)(((roots.begin + roots.count + 63) & ~63LL) - sa.root_begin);
                sa.valid_begin = (int)roots.begin;
                sa.valid_end = (int)(roots.begin + roots.count);
                sa.roots_are_primaries = 0;
                sa.fused = 0;
                if (!prof) sa.wave_log = nullptr;
                else if (sa.wave_log) HIP_TRY(hipMemsetAsync(sa.wave_log, 0, (size_t)24 * NDT_STREAM_LOG_WAVES * sizeof(unsigned int), s));
                if ((long long)sa.root_begin + sa.n_primary > ws.cap) {
                    cap *= 2;
                    continue;
                }
                const long long sh_batches = (long long)sa.n_seg * (sa.seg_cap / 64) + NDT_STREAM_LOG_WAVES;
                hipLaunchKernelGGL(k_stream_init, dim3(512), dim3(256), 0, s, ws, sa, (long long)sa.node_batches, sh_batches, 0);
                if (prof) {
                    ev_k0 = get_event(ctx, ev_n++);
                    ev_k1 = get_event(ctx, ev_n++);
                }
                kt->frame_stream(s, ctx->d_blob, sd_pass, ws, rg, sa, ctx->tier, ctx->sd.mask_words, ev_k0, ev_k1);
                if (prof) {
                    trace_ev.push_back({ ev_k0, ev_k1 });
                    trace_dbg.push_back("frame kernel, bounces " + std::to_string(hand) + " ..");
                }
                ++launches;
                streamed = true;
            }
            n_run = hand;           // the bounces the per-bounce resolve below walks
        }
        // bottom-up colour resolve, deepest bounce first (the primaries last)
        {
            for (int b = n_run; b-- > 1;) {        // (bounce 0, the primaries: inside k_finish_pixels)
                if (resolve_with_finish && b == n_run - 1) continue;
                long long blocks = (level_nodes[b] + 255) / 256;
                if (blocks > NDT_SHADE_MAX_BLOCKS) blocks = NDT_SHADE_MAX_BLOCKS;
                hipLaunchKernelGGL(k_resolve, dim3((unsigned)blocks), dim3(256), 0, s, ctx->d_blob, sd_pass, ws, rg.specular, b);
            }
        }
        hipLaunchKernelGGL(k_finish_pixels, dim3((unsigned)((rg.n_primary + NDT_FINISH_BLOCK - 1) / NDT_FINISH_BLOCK)), dim3(NDT_FINISH_BLOCK), 0, s, ctx->d_blob, sd_pass, ws,
                           rg, ctx->dims, (double *)d_rgba, (double *)d_depth, n_run >= 1 ? 1 : 0);
        const StreamCtl *sctl = streamed ? sa.ctl : nullptr;
        if (prof)
            hipExtLaunchKernelGGL(k_frame_done, dim3(1), dim3(64), 0, s, nullptr, ev_end, 0u, ws, n_run, ctx->d_done, tag, sctl);
        else
            hipLaunchKernelGGL(k_frame_done, dim3(1), dim3(64), 0, s, ws, n_run, ctx->d_done, tag, sctl);
        HIP_TRY(hipGetLastError());
        if (prof && ctx->debug_levels) {
            // the bounce table only feeds the debug output
            HIP_TRY(hipMemcpyAsync(hl, ws.levels, (size_t)(n_run + 1) * sizeof(LevelRange), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        {
            // k_frame_done is the last kernel of the frame: once its tag is here, the image and the record are complete
            const double t_wait = wall_s();
            while (__atomic_load_n(&ctx->h_done[7], __ATOMIC_ACQUIRE) != tag) {
                if (wall_s() - t_wait > 30.0) {
                    HIP_TRY(hipStreamSynchronize(s));
                    if (__atomic_load_n(&ctx->h_done[7], __ATOMIC_ACQUIRE) != tag) return fail(NDT_E_STATE, "the frame never completed");
                }
            }
        }
        if (prof) HIP_TRY(hipEventSynchronize(ev_end));     // the closing kernel has run: its completion is at most microseconds away
        hc[0] = (int)(long long)ctx->h_done[0];
        hc[2] = (int)(long long)ctx->h_done[1];
        hc[3] = (int)(long long)ctx->h_done[2];
        const unsigned long long ref_rays = ctx->h_done[5];
        const int s_overflow = (int)(long long)ctx->h_done[8], s_abort = (int)(long long)ctx->h_done[9];
        if (hc[2] != 0 || s_overflow != 0) {
            if (ctx->debug_levels)
                fprintf(stderr, "ndt_hip: overflow: per-bounce kernels %d (needs %d), frame kernel %d; pool %lld nodes, %lld shadow slots\n", hc[2],
                        hc[3], s_overflow, cap, sh_cap);
            // a pool overflowed somewhere in the frame: grow it and render again
            if ((hc[2] & 1) || (s_overflow & 1)) cap *= 2;
            if ((hc[2] & 2) || (s_overflow & 2)) {
                sh_cap *= 2;
                if (sh_cap < hc[3]) sh_cap = hc[3];
            }
            // (the frame kernel keeps the shadow rays of ALL its bounces: its segments grow with the node pool, together)
            if (s_overflow != 0 && sh_cap < cap * (n_seg > 0 ? n_seg : 1)) sh_cap = cap * (n_seg > 0 ? n_seg : 1);
            continue;
        }
        if (s_abort != 0)
            return fail(NDT_E_STATE, "the frame kernel gave up (abort %d, where %d): a work item never arrived", s_abort & 0xff, s_abort >> 8);
        const long long shadow_total = (long long)ctx->h_done[3];
        const int levels_used = (int)ctx->h_done[4];
        st = ndt_render_stats{};
        st.rays_primary = n_pixels;
        st.rays_secondary = (long long)hc[0] - rg.n_primary + (long long)ctx->h_done[10];
        st.rays_shadow = shadow_total;
        st.rays_ref_equiv = (long long)ref_rays;
        st.levels = levels_used;
        st.trace_launches = launches;
        st.node_capacity = ws.cap;
        if (prof) {
            float ms = 0;
            for (auto &pr : trace_ev) {
                float m = 0;
                HIP_TRY(hipEventElapsedTime(&m, pr.first, pr.second));
                ms += m;
            }
            st.trace_ms = ms;
            if (streamed && sa.wave_log && ev_k0) {
                float km = 0;
                HIP_TRY(hipEventElapsedTime(&km, ev_k0, ev_k1));
                print_stream_probe(sa.wave_log, km);
            }
            if (ctx->debug_levels) {
                for (int b = 0; b < levels_used; ++b)
                    fprintf(stderr, "ndt_hip: bounce %d: %lld nodes, %lld shadow rays\n", b, hl[b].count, hl[b].n_shadow);
                unsigned long long d[160];
                if (hipMemcpy(d, ws.dbg, sizeof(d), hipMemcpyDeviceToHost) != hipSuccess) d[4] = 0;
                if (shade_probe >= 0 && ws.shade_log) {
                    std::vector<unsigned int> log((size_t)2 * NDT_SHADE_LOG_WAVES);
                    if (hipMemcpy(log.data(), ws.shade_log, log.size() * sizeof(unsigned int), hipMemcpyDeviceToHost) == hipSuccess) {
                        unsigned int t0 = 0;
                        bool any = false;
                        for (int w = 0; w < NDT_SHADE_LOG_WAVES; ++w)
                            if (log[2 * w + 1] && (!any || (int)(log[2 * w] - t0) < 0)) {
                                t0 = log[2 * w];
                                any = true;
                            }
                        for (int part = 0; part < 2; ++part) {
                            // part 0: lighting (shade_finish) wavefronts, part 1: shading (shade_emit) wavefronts
                            int hist[48] = { 0 }, n_w = 0;
                            double sum = 0, longest = 0, last_start = 0, last_end = 0;
                            for (long long w = 0; w < NDT_SHADE_LOG_WAVES; ++w) {
                                const bool lighting = w < shade_probe_finish_waves;
                                if (!log[2 * w + 1] || lighting != (part == 0)) continue;
                                const double st_us = (log[2 * w] - t0) / 100.0, dur = (log[2 * w + 1] - log[2 * w]) / 100.0;
                                ++n_w;
                                sum += dur;
                                if (dur > longest) longest = dur;
                                if (st_us > last_start) last_start = st_us;
                                if (st_us + dur > last_end) last_end = st_us + dur;
                                const int bin = (int)(dur / 4.0);
                                ++hist[bin > 47 ? 47 : bin];
                            }
                            if (!n_w) continue;
                            std::string line;
                            for (int bin = 0; bin < 48; ++bin)
                                if (hist[bin]) {
                                    char buf[48];
                                    snprintf(buf, sizeof buf, " %d-%d:%d", bin * 4, bin * 4 + 4, hist[bin]);
                                    line += buf;
                                }
                            fprintf(stderr, "ndt_hip: shade launch %d, %s: %d wavefronts, mean life %.1f us, longest %.1f us, last start at %.1f us, last end at %.1f us; lives per 4 us:%s\n",
                                    shade_probe, part == 0 ? "lighting" : "shading", n_w, sum / n_w, longest, last_start, last_end, line.c_str());
                        }
                    }
                }
                if (ctx->exit_probe) {
                    // the life of every wavefront of every trace launch: when the queue runs dry (first exit), how long the
                    // rest keeps going, and how much of that is the last wavefront's last batch
                    std::vector<unsigned int> log((size_t)NDT_EXIT_LOG_LAUNCHES * NDT_EXIT_LOG_WORDS);
                    if (hipMemcpy(log.data(), ws.exit_log, log.size() * sizeof(unsigned int), hipMemcpyDeviceToHost) == hipSuccess)
                        for (int l = 0; l < NDT_EXIT_LOG_LAUNCHES && l < launches; ++l) {
                            const unsigned int *q = log.data() + (size_t)l * NDT_EXIT_LOG_WORDS;
                            unsigned int t0 = 0;
                            int n_w = 0;
                            for (int w = 0; w < NDT_EXIT_LOG_WORDS / 8; ++w)
                                if (q[8 * w + 2]) {
                                    if (!n_w || (int)(q[8 * w] - t0) < 0) t0 = q[8 * w];
                                    ++n_w;
                                }
                            int hist[64] = { 0 };
                            double first = 1e30, last = 0, last_batch = 0, start_spread = 0, last_exit = 0;
                            int simd_of_wave[16][4] = { { 0 } };        // workgroup wavefront w -> SIMD it ran on
                            int wpw = 12;                               // wavefronts per workgroup of this launch (logged by the kernel)
                            for (int w = 0; w < NDT_EXIT_LOG_WORDS / 8; ++w)
                                if (q[8 * w + 2]) {
                                    // "out of work" = out of batches (a consumer of the straggler ring exits when the launch closes)
                                    const double st_us = (q[8 * w] - t0) / 100.0, ex_us = (q[8 * w + 4] - t0) / 100.0;
                                    wpw = (int)(q[8 * w + 3] >> 24) > 0 && (q[8 * w + 3] >> 24) <= 16 ? (int)(q[8 * w + 3] >> 24) : wpw;
                                    ++simd_of_wave[w % wpw][(q[8 * w + 3] >> 4) & 3];
                                    if (st_us > start_spread) start_spread = st_us;
                                    if (ex_us < first) first = ex_us;
                                    if (ex_us > last) {
                                        last = ex_us;
                                        last_batch = (q[8 * w + 4] - q[8 * w + 1]) / 100.0;
                                    }
                                    if ((q[8 * w + 2] - t0) / 100.0 > last_exit) last_exit = (q[8 * w + 2] - t0) / 100.0;
                                    const int bin = (int)(ex_us / 16.0);
                                    ++hist[bin > 63 ? 63 : bin];
                                }
                            std::string line;
                            for (int bin = 0; bin < 64; ++bin)
                                if (hist[bin]) {
                                    char buf[48];
                                    snprintf(buf, sizeof buf, " %d-%d:%d", bin * 16, bin * 16 + 16, hist[bin]);
                                    line += buf;
                                }
                            if (l == 0) {
                                std::string m;
                                for (int w = 0; w < wpw; ++w) {
                                    char buf[64];
                                    snprintf(buf, sizeof buf, " w%d:%d/%d/%d/%d", w, simd_of_wave[w][0], simd_of_wave[w][1], simd_of_wave[w][2], simd_of_wave[w][3]);
                                    m += buf;
                                }
                                fprintf(stderr, "ndt_hip: SIMD 0/1/2/3 of the workgroup's wavefronts (%d per workgroup):%s\n", wpw, m.c_str());
                            }
                            fprintf(stderr, "ndt_hip: trace launch %d: %d wavefronts start within %.1f us; first out of work at %.1f us, last at %.1f us (its last batch: %.1f us); exits per 16 us:%s\n",
                                    l, n_w, start_spread, first, last, last_batch, line.c_str());
                            {
                                const unsigned int *cl = reinterpret_cast<const unsigned int *>(d + 100 + 2 * l);
                                if (cl[0] || cl[1])
                                    fprintf(stderr, "ndt_hip:    stragglers: %u rays given up, %u traced cooperatively (%.1f us each); the last wavefront left at %.1f us\n",
                                            cl[0], cl[1], cl[1] ? cl[2] / 100.0 / cl[1] : 0.0, last_exit);
                            }
                        }
                }
                if (d[4]) {
                    // NDT_PHASE_TIMING builds only (make -C ndt_amd/csrc timing)
                    fprintf(stderr, "ndt_hip: wave cycles T %llu G %llu I %llu list-end %llu prologue %llu outside %llu over %llu waves\n", d[0], d[1], d[2], d[3], d[5], d[6], d[4]);
                    if (d[7] || d[32])
                        fprintf(stderr, "ndt_hip:    coherent leaf scan: fetching windows %llu, boxes / gates of the windows %llu (its intersections are in I)\n", d[32], d[7]);
                    fprintf(stderr, "ndt_hip: per-ray counts over %llu rays: node visits %llu, face gates %llu (pass %llu), item gates %llu (pass %llu), isect hits %llu\n",
                            d[14], d[8], d[9], d[10], d[11], d[12], d[13]);
                    fprintf(stderr, "ndt_hip: batch time inside trace_kd (100 MHz wall clock): closest max %.1f us mean %.1f us, shadow max %.1f us mean %.1f us\n",
                            d[40] / 100.0, d[44] ? d[42] / 100.0 / d[44] : 0.0, d[41] / 100.0, d[45] ? d[43] / 100.0 / d[45] : 0.0);
                    fprintf(stderr, "ndt_hip: per-ray maxima: %llu node visits, %llu gates, %llu intersections; per-batch maxima: %llu T, %llu G, %llu I iterations\n",
                            d[46], d[47], d[48], d[49], d[50], d[51]);
                    if (d[58])
                        fprintf(stderr, "ndt_hip: shade_emit per wavefront (wall-clock ticks, mean over %llu): load+isect %.0f, light tests %.0f, segment reserve %.0f, shadow stores %.0f, spawn %.0f; slowest wavefront %llu\n",
                                d[58], (double)d[52] / d[58], (double)d[53] / d[58], (double)d[54] / d[58], (double)d[55] / d[58], (double)d[56] / d[58], d[59]);
                    for (int kind = 0; kind < 2; ++kind) {
                        const unsigned long long *q = d + 16 + 8 * kind;
                        fprintf(stderr, "ndt_hip: loop occupancy (%s rays): T %.1f%% of %llu iters, G %.1f%% of %llu, I %.1f%% of %llu\n",
                                kind ? "shadow" : "closest", q[0] ? 100.0 * q[1] / (64.0 * q[0]) : 0.0, q[0],
                                q[2] ? 100.0 * q[3] / (64.0 * q[2]) : 0.0, q[2], q[4] ? 100.0 * q[5] / (64.0 * q[4]) : 0.0, q[4]);
                        if (q[4])
                            fprintf(stderr, "ndt_hip:    I iterations execute %.2f primitive types on average; the commonest type holds %.1f of %.1f active lanes\n",
                                    (double)q[6] / q[4], (double)q[7] / q[4], (double)q[5] / q[4]);
                    }
                }
                for (size_t i = 0; i < trace_ev.size(); ++i) {
                    float m = 0;
                    (void)hipEventElapsedTime(&m, trace_ev[i].first, trace_ev[i].second);
                    fprintf(stderr, "ndt_hip: trace launch %zu: %.3f ms (%s)\n", i, m, trace_dbg[i].c_str());
                }
            }
            float fm = 0;
            HIP_TRY(hipEventElapsedTime(&fm, ev_begin, ev_end));
            st.frame_ms = fm;
        }
        return NDT_OK;
    }
    return fail(NDT_E_NOMEM, "ray-tree workspace kept overflowing");
}

