// ndt_ctx.hpp -- what the host-side translation units of libndt_hip.so share: the context, the error
// convention, and the internal entry points between them.
//
//   ndt_api.hip      create / destroy / upload / trace_rays / quantize: the plain C ABI of include/ndt_hip.h
//   ndt_blob.hip     scene validation, the plugins' prepare() data, hull boxes: the scene blob (host code only)
//   ndt_frame.hip    workspace + one pass of the ray pipeline over a set of primaries (render_pass)
//   ndt_aa.hip       Whitted's recursive anti-aliasing on top of render_pass
//   ndt_sampled.hip  -n samples > 1, lens, area lights on top of render_pass
//   ndt_render.hip   ndt_hip_render*: argument checks and the choice between the three
//   ndt_multi.hip    one frame over several contexts / devices
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <functional>
#include <string>
#include <vector>

#include "../../include/ndt_hip.h"
#include "ndt_kernels.hpp"

namespace ndt_impl {

struct CtxWorker;       // ndt_multi.hip: the thread that drives a context inside a multi-context render

// sets the calling thread's ndt_hip_last_error() text and returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

} // namespace ndt_impl
using namespace ndt_impl;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(NDT_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

struct ndt_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    const NdtKernelTable *kt = nullptr;
    int dims = 0;
    bool have_scene = false;
    double aperture_radius = 0.0;   // camera.h:46 of the uploaded scene
    int cam_type = 0;
    bool have_eyes = false, have_local_axes = false;
    bool has_area_lights = false;   // LIGHT_DISK / LIGHT_RECT: every render is stochastic (ndt.c:116-147)
    SceneDesc sd{};
    std::vector<double> blob;
    double *d_blob = nullptr;
    size_t d_blob_words = 0;
    int tier = 0;
    int n_shadow_lights = 0;
    // workspace
    Workspace ws{};
    std::vector<void *> ws_allocs;
    std::vector<void *> sa_allocs;                  // the frame kernel's queues and counters (ensure_stream_args)
    std::vector<std::pair<void *, size_t>> pool;    // scratch of the multi-pass renderers (AaBuffers)
    std::vector<std::pair<void *, size_t>> pool2;   // ... of a sampled render nested in an anti-aliased one
    long long ws_dims = 0;
    long long ws_slab_words = 0;
    int ws_nseg = 0;
    // the streaming frame kernel (ndt_stream.hpp): its queues and counters live beside the workspace
    // Which pipeline renders a pass (fixed at context creation: NDT_HIP_PIPELINE=auto | levels | stream | hybrid).
    //   levels  one trace launch + shade launches per bounce: three wavefronts per SIMD in the trace kernel, shade kernels
    //           with the whole chip's wavefront slots -- the better one where the rays are many (1080p: 1.46 ms against 1.77);
    //   stream  the streaming frame kernel (ndt_stream.hpp): no per-bounce latency floor -- the better one for passes of up
    //           to about a million primaries (64x36: 0.64 against 0.77 ms, 480x270: 0.61 against 0.72, 960x540: 0.68 against
    //           0.85), which is also what one GPU of eight renders of a 3840x2160 frame;
    //   hybrid  the first hybrid_level bounces per bounce, the deeper ones by the frame kernel: measured, not chosen by
    //           auto (the benchmark frame has a quarter of its rays in bounces 2 and 3: 2.28 ms; hypercube 3-D 0.65 = levels);
    //   auto    stream up to stream_below primaries, levels above.
    int pipeline = 0;               // 0 auto, 1 levels, 2 stream, 3 hybrid
    int hybrid_level = 2;           // hybrid: the bounce from which on the frame kernel renders (NDT_HIP_HYBRID_LEVEL)
    long long stream_below = 1000000;       // where the two cross on the benchmark scene (profiles/r03_frame_time_vs_size_*.txt: 1280x720 stream 0.918 / levels 0.965 ms, 1408x792 1.054 / 1.045; the r::8 shard of a 3840x2160 frame, 1.04 M primaries: 0.977 per bounce, 1.00 streamed)
    long long stream_below_list = 30000;    // ... for passes over a list of samples (-a): 1080p -a 20,4 of the benchmark scene 27.6 -> 24.0 ms, balls 14.0 -> 12.7
    bool use_stream = false;        // the choice for the pass being rendered
    StreamArgs sa{};
    // ndt_hip_set_option / NDT_HIP_* at context creation (include/ndt_hip.h)
    bool stream_probe = false, exit_probe = false, debug_levels = false, test_small_pool = false;
    bool hull_box = true, face_box = true, shade_pair = true;
    bool face_tree = true;          // hcubes of more than 63 faces: a hierarchy over the face boxes (ndt_device.hpp:hull_faces)
    bool face_groups = true;        // ... and an index of the faces by the set of hull axes their boxes are thin on (hull_faces)
    // per-bounce kernels: the first trace launch makes the primaries it traces (no k_primary; k_trace's PRIM variant, planar camera).
    // -1: from 4-D on (measured, 1080p, on / off: benchmark frame 1.297 / 1.304 ms, balls 0.814 / 0.821, 6-D 1.298 / 1.326, 8-D 3.25 / 3.28;
    // 3-D 0.592 / 0.582 -- the variant spills a little more than the plain kernel, which the 2N doubles a ray it does not write and
    // read back pay for from N = 4 on), 0 never, 1 always
    int fuse_primaries = -1;
    bool stream_fused = true;       // frame kernel: makes its primaries and writes its pixels itself (no k_primary / k_finish_pixels)
    bool item_sets = true;          // scenes of up to 64 items: leaf records carry item sets (ndt_blob.hip:build_blob)
    // item sets: the min_dist-free part of every gate before the walk (ndt_device.hpp:trace_kd).  0 never, 1 always, 2 (default)
    // for passes of up to gate_prepass_below primaries -- measured (profiles/r03_gate_prepass.txt): 64x36 -6 %, 480x270 -4 %,
    // 960x540 +2 %, 1080p -1 %, the 3-D scene at 1080p +2 %: it shortens the walk of a lone slow ray, and costs the busy chip
    // about what it saves
    int gate_prepass = 2;
    long long gate_prepass_below = 400000;
    bool item_boxes = true;         // global-memory tier: orthotopes carry a box in one scene-wide frame (ndt_blob.hip:scene_item_boxes)
    int leaf_scan_group = 64;       // ... when at least this many lanes share the leaf
    bool leaf_scan = true;          // global-memory tier: lanes on the same leaf stage its items through LDS (ndt_device.hpp:cls_scan)
    // item-set tier: a batch of the trace kernel that is over coop_budget_us with at most coop_max_live rays left gives them up;
    // they are traced one per wavefront by the wavefronts that have run out of batches (ndt_device.hpp:coop_trace).
    // OFF by default: bit-identical answers (tests/test_gpu_parity.py), but measured slower on every setting tried -- a ray costs a
    // wavefront 9-13 us that way, the slow batches of a launch turn out to be WIDE (dozens of rays alive until late, in step),
    // and a launch has thousands of them, not a handful (profiles/experiments/r04_coop_stragglers.md)
    bool coop = false;
    int coop_budget_us = 25, coop_max_live = 8;
    bool coop_tail_only = true;     // ... and only once the batch's queue shard has run dry (the launch is in its tail)
    int coop_waves = 4;             // wavefronts of a workgroup that stay on as consumers
    unsigned int coop_tag = 0;      // serial number of the last trace launch that had a straggler ring
    int leaf_history = 4;           // global-memory tier: visited = {leaf, cut} pairs per ray (VisitMask<0>); 0: the slab only; 1 .. 3: fewer pairs (tests)
    int shade_probe = -1;           // the k-th shade launch of a frame logs its wavefronts (-1: none)
    long long sa_cap = 0, sa_sh_cap = 0;
    int sa_nseg = 0;
    void *d_eyes = nullptr;         // the two eye images of a stochastic anaglyph render (ndt_sampled.hip)
    size_t d_eyes_bytes = 0;
    void *d_out = nullptr;          // staging for ndt_hip_render (host output)
    size_t d_out_bytes = 0;
    void *d_shard = nullptr;        // ndt_hip_render_multi: this context's rows before they are pushed into the frame
    size_t d_shard_bytes = 0;
    void *d_image = nullptr;        // ndt_hip_render_multi (host output): the assembled frame on the first context's device
    size_t d_image_bytes = 0;
    // ndt_hip_render_rgba8_async: two quantised frames in HBM, a copy stream, and per buffer the events "quantised" / "copied"
    void *d_rgba8[2] = { nullptr, nullptr };
    size_t d_rgba8_bytes[2] = { 0, 0 };
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_quantised[2] = { nullptr, nullptr }, ev_copied[2] = { nullptr, nullptr };
    bool copy_pending[2] = { false, false };
    int rgba8_turn = 0;
    // ndt_hip_render_multi without peer stores: this context's staging buffer and stream ON THE FIRST CONTEXT'S DEVICE
    void *d_stage = nullptr;
    size_t d_stage_bytes = 0;
    hipStream_t stage_stream = nullptr;
    int stage_device = 0;
    long long sample_seed = 0;      // option "sample_seed": selects the set of random streams of the stochastic paths (0: the default set)
    int multi_path = 0;             // option "multi_path": 0 auto, 1 never staged, 2 always staged
    int multi_path_taken = 0;       // ndt_multi_path of the last multi-context frame (ndt_hip_multi_path_taken)
    ndt_impl::CtxWorker *worker = nullptr;
    int *h_counters = nullptr;      // pinned
    LevelRange *h_levels = nullptr; // pinned, NDT_MAX_LEVELS + 1
    LevelRange *h_mail = nullptr;   // mapped + coherent: bounce ranges posted by k_level_step while the frame runs
    unsigned long long *h_mail_tag = nullptr;
    LevelRange *d_mail = nullptr;   // the device's view of the two
    unsigned long long *d_mail_tag = nullptr;
    unsigned long long frame_tag = 0;
    unsigned long long *h_done = nullptr;   // mapped + coherent: the frame's closing record (k_frame_done), [7] = its tag
    unsigned long long *d_done = nullptr;
    std::vector<hipEvent_t> ev_pool;
};

namespace ndt_impl {

// ndt_blob.hip
struct HullFaces {
    // per face of the hcube, in the hull box's frame: N x { centre coordinate, half extent } -- the face's own
    // box, same derivation and margin as the hull box; possible bit f clear = face f can never be hit
    // (possible: one word per 63 faces, bit j of word c = face 63 c + j; the top bit of every word stays clear)
    std::vector<double> rows;
    std::vector<unsigned long long> possible;
    int n_faces = 0;
};
bool hcube_hull_box(const ndt_flat_scene *fs, const ndt_flat_object &o, int n, std::vector<double> &rows, HullFaces *faces = nullptr);
void hcube_face_tree(const HullFaces &hf, int n, std::vector<double> &rows, std::vector<int> &level_off, int &top);
// clusters: n x { centre-, half-, centre+, half+ }; table: 2^n x { start, count } (ints) into members; face_set: the thin axes of every face
void hcube_face_groups(const HullFaces &hf, const std::vector<double> &hull_rows, int n, std::vector<double> &clusters, std::vector<int> &table,
                       std::vector<int> &face_set, std::vector<int> &members);
bool scene_item_boxes(const ndt_flat_scene *fs, int n, std::vector<double> &frame, std::vector<double> &rows, std::vector<char> &has);
int build_blob(ndt_hip_ctx *ctx, const ndt_flat_scene *fs);

// ndt_frame.hip
void free_workspace(ndt_hip_ctx *ctx);
int ensure_workspace(ndt_hip_ctx *ctx, long long cap, long long sh_cap);
void coop_setup(ndt_hip_ctx *ctx, TraceJob &tj, unsigned int *log);
int render_pass(ndt_hip_ctx *ctx, RenderGeom rg, bool prof, void *d_rgba, ndt_render_stats &st, void *d_depth = nullptr);
void launch_fill_black(hipStream_t s, double *rgba, long long n_pixels);
void add_stats(ndt_render_stats &acc, const ndt_render_stats &st);

// ndt_multi.hip
void worker_stop(ndt_hip_ctx *ctx);
void free_stage(ndt_hip_ctx *ctx);
void free_async(ndt_hip_ctx *ctx);

// ndt_aa.hip / ndt_sampled.hip
int render_antialiased(ndt_hip_ctx *ctx, const ndt_render_params *p, void *d_rgba, ndt_render_stats &total, void *d_depth = nullptr);
// ndt_render.hip: true anaglyph (ndt.c:643-647) of two eye images, n_pixels x rgba each
void launch_anaglyph(hipStream_t s, const double *left, const double *right, double *out, long long n_pixels);
int render_sampled(ndt_hip_ctx *ctx, const ndt_render_params *p, void *d_rgba, ndt_render_stats &total, void *d_depth = nullptr);
// get_pixel_color's adaptive loop (ndt.c:488-568) for a LIST of image positions (2 doubles each, pixels of an img_w x img_h
// image): the samples of a stochastic anti-aliased render (a lens, area lights).  No jitter (ndt.c:505: not in this mode).
struct SampledList {
    const double *d_pos;
    long long n_pos;
    int img_w, img_h, aspect_w, aspect_h;
};
int render_sampled_list(ndt_hip_ctx *ctx, const ndt_render_params *p, const SampledList &sl, int eye, int stereo, unsigned long long salt,
                        void *d_rgba, ndt_render_stats &total);

// One slot of a device-side list for every lane that wants one, with ONE atomic per wavefront (a counter serves ~150
// returning atomics per us: a list appended to by every thread of a 2-million-thread launch queues for milliseconds).
// Every lane of the wavefront has to call it.
__device__ __forceinline__ int wave_append(int *counter, bool want)
{
    const unsigned long long vote = __ballot(want);
    if (vote == 0ull) return 0;
    const int lane = __lane_id(), leader = __ffsll((long long)vote) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(vote));
    base = __shfl(base, leader, 64);
    return base + __popcll(vote & ((1ull << lane) - 1ull));
}

// The same for a whole workgroup (up to 1024 lanes): ONE atomic on the counter a workgroup, its wavefronts' places handed out
// through LDS.  Every lane of the workgroup must call it.  (Atomics on one address complete at about a hundred million a
// second: a kernel of 32 000 wavefronts that appends once per wavefront spends 0.3 ms on its counter, whatever else it does.)
__device__ __forceinline__ int block_append(int *counter, bool want)
{
    __shared__ int wave_count[16], wave_base[16];
    const unsigned long long vote = __ballot(want);
    const int lane = __lane_id(), wave = (int)(threadIdx.x >> 6), n_waves = (int)((blockDim.x + 63) >> 6);
    if (lane == 0) wave_count[wave] = __popcll(vote);
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < n_waves; ++w) { wave_base[w] = total; total += wave_count[w]; }
        const int base = total > 0 ? atomicAdd(counter, total) : 0;
        for (int w = 0; w < n_waves; ++w) wave_base[w] += base;
    }
    __syncthreads();
    return wave_base[wave] + __popcll(vote & ((1ull << lane) - 1ull));
}

// ... and for lanes that append `n` items each (0 <= n): returns the lane's first slot; its items are adjacent
__device__ __forceinline__ int block_append_n(int *counter, int n)
{
    __shared__ int wave_count_n[16], wave_base_n[16];
    const int lane = __lane_id(), wave = (int)(threadIdx.x >> 6), n_waves = (int)((blockDim.x + 63) >> 6);
    int incl = n;                                   // inclusive prefix sum over the wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wave_count_n[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < n_waves; ++w) { wave_base_n[w] = total; total += wave_count_n[w]; }
        const int base = total > 0 ? atomicAdd(counter, total) : 0;
        for (int w = 0; w < n_waves; ++w) wave_base_n[w] += base;
    }
    __syncthreads();
    return wave_base_n[wave] + incl - n;
}

// Scratch buffers of the multi-pass renderers (anti-aliasing levels, sample rounds, anaglyph eyes).  The
// requests of a frame come in the same order every frame, so the k-th request reuses the k-th
// allocation of the context's pool (grown when too small) instead of a hipMalloc / hipFree pair, each of
// which synchronises the device.
struct AaBuffers {
    ndt_hip_ctx *ctx;
    std::vector<std::pair<void *, size_t>> *from;
    size_t next = 0;
    explicit AaBuffers(ndt_hip_ctx *c, bool nested = false) : ctx(c), from(nested ? &c->pool2 : &c->pool) {}
    template <typename T> int get(T **ptr, size_t count)
    {
        const size_t bytes = (count > 0 ? count : 1) * sizeof(T);
        if (next == from->size()) from->push_back({ nullptr, 0 });
        auto &slot = (*from)[next++];
        if (slot.second < bytes) {
            if (slot.first) {
                (void)hipStreamSynchronize(ctx->stream);
                (void)hipFree(slot.first);
                slot = { nullptr, 0 };
            }
            const size_t want = bytes + bytes / 4;          // some head room: frame-to-frame counts vary
            void *q = nullptr;
            hipError_t e = hipMalloc(&q, want);
            if (e != hipSuccess) return fail(NDT_E_NOMEM, "hipMalloc of %zu bytes: %s", want, hipGetErrorString(e));
            slot = { q, want };
        }
        *ptr = (T *)slot.first;
        return NDT_OK;
    }
};

} // namespace ndt_impl
