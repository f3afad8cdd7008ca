// ndt_kernels.hip -- the gfx950 kernels of the wavefront tracer, compiled once per dimension
// (-DNDT_DIMS=3..8).  One thread per ray; rays of one bounce form a structure-of-arrays batch;
// live rays of the next bounce and the shadow rays of this one are compacted with a
// wavefront prefix sum and one atomic per wavefront.
//
// Kernels per bounce (host loop: ndt_frame.hip:render_pass):
//   k_trace        trace_kd (object.c:683) for every node of the bounce       -> (object, primitive)
//   k_shade_emit   first half of apply_lights (ndt.c:71-259): hit point, same-side test,
//                  spot cone, one shadow ray per light that passes            -> shadow queue
//   k_trace        trace_kd for the shadow queue (dist_limit per ray)
//   k_shade_finish second half of apply_lights (ndt.c:217-310) + the reflect / refract spawn
//                  of get_ray_color (ndt.c:381-430)                           -> next bounce
#if NDT_DIMS <= 5 && !defined(NDT_OUTLINE_SMALL)
// inlined libm for the small vectors, out of line from 6-D on (ndt_device.hpp, NDT_LIBM)
#define NDT_INLINE_LIBM 1
#endif
#if NDT_DIMS <= 7 && !defined(NDT_OUTLINE_SMALL)
// The hit-point intersection of the lighting kernels inlined up to 7-D, a real function from 8-D on.  (Round 3: the call hands
// 4N doubles in and 2N out, and past 32 argument registers they travel through the stack -- 352 bytes of scratch a lane at
// 6-D, real memory traffic; inlined, 1080p: 6-D 1.41 -> 1.35 ms, 7-D 2.02 -> 1.97, 8-D 3.26 -> 3.28.  shade_emit, which needs
// few registers of its own, always has it inline: 6-D 1.59 -> 1.41 ms, 8-D 3.47 -> 3.26, 127 registers instead of 180 + scratch.)
#define NDT_SHADE_INLINE_ISECT 1
#endif
#include "ndt_kernels.hpp"
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <tuple>

#ifndef NDT_DIMS
#error "compile with -DNDT_DIMS=<3..10>"
#endif
#define NDT_CAT2(a, b) a##b
#define NDT_CAT(a, b) NDT_CAT2(a, b)

// every dimension gets its own namespace: the six translation units define the same kernels
namespace NDT_CAT(ndt_d, NDT_DIMS) {

static constexpr int N = NDT_DIMS;

// ------------------------------------------------------------------ wavefront helpers

// Exclusive prefix sum of x over the 64 lanes of the wavefront; `total` = sum over all lanes.
// Every lane of the wavefront must call it.
NDT_DEV int wave_excl_scan(int x, int &total)
{
    const int lane = __lane_id();
    int s = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(s, d, 64);
        if (lane >= d) s += y;
    }
    total = __shfl(s, 63, 64);
    return s - x;
}

// Reserve `total` consecutive slots for the wavefront with one atomic; returns the base.
NDT_DEV int wave_reserve(int *counter, int total)
{
    int base = 0;
    if (__lane_id() == 0 && total > 0) base = atomicAdd(counter, total);
    return __shfl(base, 0, 64);
}

// K-vectors of the node pool / shadow queue live in tiles of 64 slots, component-major inside the
// tile: x[(g/64)*K*64 + c*64 + g%64].  A wavefront still reads component c of 64 consecutive slots
// as one 512-byte request, but the K requests of a batch now fall into one K*512-byte block
// instead of K regions `cap` doubles apart (one TLB entry / DRAM page per batch instead of K).
// (`stride`, the pool capacity, is kept in the signature for the callers' sake; capacities are
// multiples of 64.)
template <int K> NDT_DEV void load_soa(const double *base, long long stride, long long g, double (&r)[K])
{
    const double *t = base + (g >> 6) * (long long)(K * 64) + (g & 63);
#pragma unroll
    for (int c = 0; c < K; ++c) r[c] = t[c * 64];
}
template <int K> NDT_DEV void store_soa(double *base, long long stride, long long g, const double (&r)[K])
{
    double *t = base + (g >> 6) * (long long)(K * 64) + (g & 63);
#pragma unroll
    for (int c = 0; c < K; ++c) t[c * 64] = r[c];
}

// ------------------------------------------------------------------ primary rays

// render_pixel (ndt.c:578-653, MONO) + the ray set-up of get_pixel_color (ndt.c:516-549) +
// camera_target_point's CAMERA_NORMAL branch (camera.c:557-575).
// Slot g of the pool <-> lane (g % 64) of 8x8 pixel tile (g / 64): one wavefront = one tile.
// One primary: render_pixel + the head of get_pixel_color (ndt.c:578-653, 488-550) for grid slot / sample g: the node's
// record (ray, weight, bounces left) is written, and the ray is returned.  False: a padding slot or a blank line (marked so).
// COH: agent-scope (write-through) stores -- inside the frame kernel other wavefronts of the same launch read the record.
template <bool COH> NDT_DEV void prim_sti(int *p, int v)
{
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool COH> NDT_DEV void prim_st(double *p, double v)
{
    if (COH) __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool COH> NDT_DEV void prim_stu(unsigned long long *p, unsigned long long v)
{
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
// PLANAR: the caller knows the camera is CAMERA_NORMAL (the trace kernel's PRIM variant: without the VR / panorama code the
// variant spills no more than the plain kernel)
template <bool COH, bool PLANAR = false>
NDT_DEV bool primary_node(const double *blob, const SceneDesc &sd, const Workspace &ws, const RenderGeom &rg, long long g, double (&pos)[N],
                          double (&look)[N])
{
    prim_sti<COH>(ws.child_refl + g, -1);
    prim_sti<COH>(ws.child_refr + g, -1);
    prim_sti<COH>(ws.count + g, 0);
    prim_stu<COH>(ws.sh_mask + g, 0ull);
    double ip, jp;              // image position in pixels (ndt.c:590-592)
    bool live;
    int eye = rg.eye;
    if (rg.samples) {
        live = g < rg.n_samples;
        const int rec = rg.lens ? 4 : 2;
        ip = live ? rg.samples[rec * g] : 0.0;
        jp = live ? rg.samples[rec * g + 1] : 0.0;
    } else {
        const int tile = (int)(g >> 6), lane = (int)(g & 63);
        const int px = (tile % rg.tiles_x) * 8 + (lane & 7);
        const int py = (tile / rg.tiles_x) * 8 + (lane >> 3);
        live = px < rg.width && py < rg.rows;
        ip = px;
        jp = rg.row_pair ? rg.row_begin + (py >> 1) * rg.row_step + (py & 1) : rg.row_begin + py * rg.row_step;
    }
    if (!live) {
        prim_sti<COH>(ws.depth_left + g, 0);        // padding slot: never traced, never shaded
        prim_sti<COH>(ws.hit_obj + g, -1);
        return false;
    }
    // (jittered samples: the PIXEL decides the half, the jitter comes after -- i + dx/2 and j - dy/2 lie in [i, i+1) and (j-1, j])
    const double half_i = rg.pixel_halves ? floor(ip) : ip, half_j = rg.pixel_halves ? ceil(jp) : jp;
    if (rg.stereo == 1) {           // SIDE_SIDE_3D, x_scale = 0.5 (ndt.c:591-601)
        if (half_i < rg.img_w / 2) { ip = ip / 0.5; eye = 0; }
        else { ip = (ip - rg.img_w / 2) / 0.5; eye = 2; }
    } else if (rg.stereo == 2) {    // OVER_UNDER_3D, y_scale = 0.5 (ndt.c:602-612)
        if (half_j < rg.img_h / 2) { jp = jp / 0.5; eye = 0; }
        else { jp = (jp - rg.img_h / 2) / 0.5; eye = 2; }
    }
    double y_div = (double)rg.img_h;
    if (rg.stereo == 4) {           // HIDEF_3D frame packing (ndt.c:614-631): 1080 lines left eye, 45 blank, 1080 right eye
        y_div = 1080.0;
        // (jittered samples: the PIXEL's row decides, as for the halves above)
        if (half_j < 1080) {
            eye = 0;
        } else if (half_j > 1080 + 45) {
            jp = jp - (1080 + 45);
            eye = 2;
        } else {
            prim_sti<COH>(ws.depth_left + g, 0);    // blank line: black, written by the host's fill
            prim_sti<COH>(ws.hit_obj + g, -1);
            return false;
        }
    }
    const double x = ip / (double)rg.img_w - 0.5;               // ndt.c:632
    const double y = -(jp / y_div - 0.5);                       // ndt.c:633 (629 for HIDEF_3D)
    double pixel[N], temp[N], cam[N];
    blob_vec<N>(blob, sd.off_cam, pos);
    const double focal = blob[sd.off_cam + 4 * N];
    const int ext = sd.off_cam + 4 * N + 8;                     // type, hFov, vFov, leftEye, rightEye, localX, localY, localZ
    const int cam_type = PLANAR ? 0 : (int)blob[ext];
    if (!PLANAR && cam_type != 0) {
        // camera_target_point, camera.c:506-555: spherical (VR) / cylindrical (panorama) screen
        const double azi = x * blob[ext + 1];
        double view_x, view_y, view_z;
        if (cam_type == 1) {
            const double alt = y * blob[ext + 2];
            view_x = focal * sin(azi) * cos(alt);
            view_y = focal * sin(alt);
            view_z = focal * cos(azi) * cos(alt);
        } else {
            const double y_size = 2.0 * tan(blob[ext + 2] / 2.0) * focal;
            view_x = focal * sin(azi);
            view_y = y * y_size;
            view_z = focal * cos(azi);
        }
        double ax[N];
        v_copy<N>(pixel, pos);
        blob_vec<N>(blob, ext + 3 + 2 * N, ax);
        v_scale<N>(ax, view_x, temp);
        v_add<N>(pixel, temp, pixel);
        blob_vec<N>(blob, ext + 3 + 3 * N, ax);
        v_scale<N>(ax, view_y, temp);
        v_add<N>(pixel, temp, pixel);
        blob_vec<N>(blob, ext + 3 + 4 * N, ax);
        v_scale<N>(ax, view_z, temp);
        v_add<N>(pixel, temp, pixel);
    } else {
        double orig[N], dx[N], dy[N];
        blob_vec<N>(blob, sd.off_cam + N, orig);
        blob_vec<N>(blob, sd.off_cam + 2 * N, dx);
        blob_vec<N>(blob, sd.off_cam + 3 * N, dy);
        v_scale<N>(dx, rg.aspect_w / (double)rg.aspect_h, dx);  // ndt.c:926
        v_copy<N>(pixel, orig);
        v_scale<N>(dx, x, temp);
        v_add<N>(pixel, temp, pixel);
        v_scale<N>(dy, y, temp);
        v_add<N>(pixel, temp, pixel);
        const double screen_dist = v_dist<N>(orig, pos);
        if (screen_dist > NDT_EPS) {
            v_sub<N>(pixel, pos, temp);
            v_scale<N>(temp, focal / screen_dist, temp);
            v_add<N>(pos, temp, pixel);
        }
    }
    // the eye the ray starts from (ndt.c:491-502)
    v_copy<N>(cam, pos);
    if (eye == 0) blob_vec<N>(blob, ext + 3, cam);
    else if (eye == 2) blob_vec<N>(blob, ext + 3 + N, cam);
    if (!PLANAR && cam_type != 0 && eye != 1) {
        // VR: the eye goes round the centre with the view direction (ndt.c:519-525):
        // vectNd_rotate2(virtCam, pos, localX, localZ, azi) = vectNd.c:271-325 with vectNd_orthogonalize (vectNd.c:35-57)
        const double azi = x * blob[ext + 1];
        double lx[N], lz[N], bx[N], bz[N], local[N], px[N], pz[N];
        blob_vec<N>(blob, ext + 3 + 2 * N, lx);
        blob_vec<N>(blob, ext + 3 + 4 * N, lz);
        v_proj<N>(lx, lz, temp);
        v_sub<N>(lx, temp, bx);
        v_copy<N>(bz, lz);
        v_unitize<N>(bx);
        v_unitize<N>(bz);
        v_sub<N>(cam, pos, local);
        v_proj<N>(local, bx, px);
        v_proj<N>(local, bz, pz);
        const double vx = v_dot<N>(px, bx), vz = v_dot<N>(pz, bz);
        double rx[N], rz[N];
        v_scale<N>(bx, vx * cos(azi) - vz * sin(azi), rx);
        v_scale<N>(bz, vz * cos(azi) + vx * sin(azi), rz);
        v_sub<N>(cam, px, cam);
        v_sub<N>(cam, pz, cam);
        v_add<N>(cam, rx, cam);
        v_add<N>(cam, rz, cam);
    }
    if (rg.samples && rg.lens) {
        // lens sample for depth of field (ndt.c:527-542): offsets already scaled by the aperture radius
        double ax[N];
        blob_vec<N>(blob, ext + 3 + 2 * N, ax);
        v_scale<N>(ax, rg.samples[4 * g + 2], temp);
        v_add<N>(cam, temp, cam);
        blob_vec<N>(blob, ext + 3 + 3 * N, ax);
        v_scale<N>(ax, rg.samples[4 * g + 3], temp);
        v_add<N>(cam, temp, cam);
    }
    v_sub<N>(pixel, cam, look);
    v_unitize<N>(look);
    v_copy<N>(pos, cam);
    {
        // (store_soa's layout: 64 slots x N components per batch)
        const long long at = (g >> 6) * (long long)(N * 64) + (g & 63);
#pragma unroll
        for (int c = 0; c < N; ++c) {
            prim_st<COH>(ws.ray_o + at + c * 64, pos[c]);
            prim_st<COH>(ws.ray_v + at + c * 64, look[c]);
        }
    }
    prim_st<COH>(ws.frac + g, 1.0);
    prim_sti<COH>(ws.depth_left + g, rg.max_depth);
    if (rg.samples && rg.sample_keys) prim_stu<COH>(ws.rng_key + g, rg.sample_keys[g]);
    return true;
}

__global__ void __launch_bounds__(256) k_primary(const double *blob, SceneDesc sd, Workspace ws, RenderGeom rg)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= rg.n_primary) return;
    double pos[N], look[N];
    (void)primary_node<false>(blob, sd, ws, rg, g, pos, look);
}


// ------------------------------------------------------------------ trace

// The visit mask of this lane's rays.  MW == 0: its words in the global slab (one column per resident lane) and, when the
// scene carries them, the leaf sets of the history representation (ndt_device.hpp:VisitMask).
// (the bounce table is the launch's to write where it publishes: TraceJob::publish_level)
NDT_DEV LevelRange *job_levels_mut(const TraceJob &job) { return const_cast<LevelRange *>(job.levels); }

template <int MW> NDT_DEV void init_visit_mask(VisitMask<MW> &mask, const double *gblob, const SceneDesc &sd, const Workspace &ws)
{
    mask.ext = nullptr;
    mask.ext_stride = 0;
    mask.h0 = mask.h1 = mask.h2 = mask.h3 = 0u;
    mask.hist_n = -1;
    mask.slab_words = sd.mask_words;
    mask.ls.sets = nullptr;
    mask.ls.blob = gblob;
    mask.ls.words = sd.mask_words;
    mask.ls.off_lrange = sd.off_lrange;
    mask.ls.off_leaf = sd.off_leaf;
    mask.ls.cap = sd.hist_cap;
    if (MW == 0) {
        const long long lane_slot = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        mask.ext = ws.mask_slab + lane_slot;
        mask.ext_stride = (int)ws.mask_slab_lanes;
        if (sd.off_lset > 0) mask.ls.sets = reinterpret_cast<const unsigned long long *>(gblob + sd.off_lset);
    }
}

// One trace_kd query per lane.  LDS tier: the trace sections of the scene blob (kd nodes, leaf
// lists, object headers, bounding spheres, parameters) are staged once per workgroup; the
// workgroups are persistent and every wavefront pulls batches of 64 rays from a device-side
// queue (one atomic per batch), so a wavefront that drew cheap rays (sky) immediately takes
// more work instead of idling until the expensive tiles finish.  Every wavefront exits when
// the queue head passes the ray count.
// COOPK: the variant with the straggler ring (TraceJob::coop_ring; option `coop`); PRIM: the variant whose dense part is the
// pass's primaries, made here (TraceJob::make_primaries).  Both are variants, not run-time branches: the code of either in the
// kernel cost the launches that do not use it 2-3 % (registers and scratch around the traversal loop).
// Where the rays of shadow segment s start, when they all start at one point: the position of the s-th light that is not
// ambient, if it is a point or a spot light (ndt.c:211; light record: ndt_blob.hip) -- else -1 (directional: the nudged hit
// points; area lights: a point of the light per ray).  s: the same in every lane.
NDT_DEV int seg_light_origin(const double *gblob, const SceneDesc &sd, int s)
{
    unsigned long long rest = ~sd.ambient_bits;
    for (int k = 0; k < s; ++k) rest &= rest - 1ull;
    const int li = __ffsll((long long)rest) - 1;
    const int w = sd.off_lights + li * (6 + 4 * N);
    const int type = blob_int(gblob, w, 0);
    return (type == NDT_LIGHT_POINT_ || type == NDT_LIGHT_SPOT_) ? w + 5 : -1;
}

#ifndef NDT_TICKET_AHEAD
#define NDT_TICKET_AHEAD 256
#endif
template <int MW, bool LDS, bool LSTACK = false, bool COOPK = false, bool PRIM = false>
__global__ void __launch_bounds__(MW == 0 ? NDT_TRACE_T1_MAX_BLOCK : NDT_TRACE_MAX_BLOCK) k_trace(const double *__restrict__ gblob, SceneDesc sd, Workspace ws, TraceJob job)
{
    extern __shared__ __attribute__((aligned(16))) double lds_blob[];
    const double *blob = gblob;
    if (LDS) {
        // 16 bytes per lane and four loads in flight per lane: the copy is a few round trips to L2, not one per word
        const int pairs = sd.trace_words >> 1;
        const ndt_v2d *src2 = reinterpret_cast<const ndt_v2d *>(gblob);
        ndt_v2d *dst2 = reinterpret_cast<ndt_v2d *>(lds_blob);
        int i = threadIdx.x;
        for (; i + 3 * (int)blockDim.x < pairs; i += 4 * blockDim.x) {
            const ndt_v2d a = src2[i], b = src2[i + blockDim.x], c = src2[i + 2 * blockDim.x], d = src2[i + 3 * blockDim.x];
            dst2[i] = a; dst2[i + blockDim.x] = b; dst2[i + 2 * blockDim.x] = c; dst2[i + 3 * blockDim.x] = d;
        }
        for (; i < pairs; i += blockDim.x) dst2[i] = src2[i];
        if ((sd.trace_words & 1) && threadIdx.x == 0) lds_blob[sd.trace_words - 1] = gblob[sd.trace_words - 1];
        __syncthreads();
        blob = lds_blob;
    }
    KdStackLds kstack{};
    if (LSTACK) {
        // behind the scene: [kd_depth][lanes] of tu (doubles) and parent node (ints)
        double *base = lds_blob + ((sd.trace_words + 1) & ~1);
        const int depth = sd.kd_depth + 1;
        kstack.stride = blockDim.x;
        kstack.tu = base + threadIdx.x;
        kstack.node = (int *)(base + (size_t)depth * blockDim.x) + threadIdx.x;
    }
    VisitMask<MW> mask;
    init_visit_mask<MW>(mask, gblob, sd, ws);
    // global-memory tier, LDS per wavefront: the rays' projections on the item boxes' frame (N x 64 pairs), then the window
    // of the coherent leaf scan (ndt_device.hpp:item_box_meets, cls_scan)
    ClsLds cls{};
    double *box_slot = nullptr;
    if (MW == 0 && !LDS) {
        const int wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
        const int box_words = sd.off_obox > 0 ? N * 128 : 0;
        if (sd.off_obox > 0) box_slot = lds_blob + (size_t)wave * box_words + 2 * (threadIdx.x & 63);
        if (sd.cls_par_words > 0) {
            cls.base = lds_blob + (size_t)waves * box_words + (size_t)wave * cls_window_words<N>(sd.cls_par_words);
            cls.min_group = sd.cls_min_group;
        }
    }
    const int lane = __lane_id();
    // cooperative stragglers (TraceJob::coop_ring): the item-set tier only
    constexpr bool COOP = COOPK && (MW == 1) && LDS;
    const bool coop_on = COOP && job.coop_ring != nullptr;
    // NDT_HIP_EXIT_PROBE: when does every wavefront start, start its last batch, and run out of work
    const unsigned int probe_start = job.exit_log ? (unsigned int)wall_clock64() : 0u;
    unsigned int probe_batch = probe_start;
#ifdef NDT_PHASE_TIMING
    // diagnostic build: everything is accumulated in registers and flushed once per wavefront at
    // the end of the kernel, so that the counters' atomics do not sit inside the phases they measure
    unsigned long long ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };      // T, G, I, list end, -, prologue, outside trace_kd, -
    unsigned long long acc[24];                                  // [0..6] per-ray counts + rays, [8..13] / [16..21] loop occupancy
    for (int i = 0; i < 24; ++i) acc[i] = 0ull;
    unsigned long long out_last = __builtin_readcyclecounter();
    unsigned long long batch_mark = 0;
#endif
    // batches of `bs` rays (64, or fewer when the launch has too few rays to fill the chip: a
    // wavefront's time is set by its slowest lane, so half-empty wavefronts finish sooner and
    // there are idle CUs to run them); for a segmented queue every segment is padded to whole batches
    const int bs = job.batch;
    const int sh = (bs == 64) ? 6 : (bs == 32) ? 5 : (bs == 16) ? 4 : 3;
    int seg_batches = 0, seg_batches_incl = 0, seg_cnt = 0;     // lane s: inclusive prefix of batches / count of segment s
    if (job.n_seg > 0) {
        seg_cnt = (lane < job.n_seg) ? job.seg_count[lane] : 0;
        const int excl = wave_excl_scan((seg_cnt + bs - 1) >> sh, seg_batches);
        seg_batches_incl = excl + ((seg_cnt + bs - 1) >> sh);
    }
    long long dense_count = job.count, dense_begin = job.begin, seg_stride = job.seg_stride;
    if (job.levels) {
        // The bounce whose shadow rays this launch traces was published earlier on this stream; the NEXT bounce -- its nodes are
        // this launch's closest-hit part -- is published here (TraceJob::publish_level): its range is what the node tail says,
        // the same for every wavefront that looks.
        const LevelRange cur = job.levels[job.seg_level];
        seg_stride = cur.seg_stride;
        LevelRange next;
        if (job.publish_level >= 0) {
            next.begin = cur.begin + cur.count;
            next.count = (long long)ws.counters[0] - next.begin;
            next.n_shadow = 0;
            if (next.count < 0 || (ws.counters[2] & 1) != 0) next.count = 0;        // node pool overflow: the host retries
            next.seg_stride = (next.count + 63) & ~63LL;
            const bool sh_overflow = (long long)job.n_seg * next.seg_stride > ws.sh_cap;
            if (blockIdx.x == 0 && threadIdx.x < 64) {
                // the launch's first wavefront: the table, the other parity's segment counters, the host's mailbox
                long long emitted = seg_cnt;
                for (int d = 32; d > 0; d >>= 1) emitted += __shfl_xor(emitted, d, 64);
                NDT_SEG_COUNTERS(ws, job.publish_level + 1)[lane] = 0;
                if (lane == 0) {
                    job_levels_mut(job)[job.publish_level].n_shadow = emitted;
                    LevelRange out = next;
                    if (sh_overflow) {
                        // the shadow queue cannot hold the next bounce: flag it, tell the host how much it needs, stop here
                        atomicOr(&ws.counters[2], 2);
                        const long long need = (long long)job.n_seg * next.seg_stride;
                        ws.counters[3] = need > 0x7fffffffLL ? 0x7fffffff : (int)need;
                        out.count = 0;
                        out.seg_stride = 0;
                    }
                    job_levels_mut(job)[job.publish_level + 1] = out;
                    ws.mail[job.publish_level + 1] = out;
                    __threadfence_system();
                    __hip_atomic_store(&ws.mail_tag[job.publish_level + 1], job.publish_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            if (sh_overflow) next.count = 0;
        } else {
            next = job.levels[job.dense_level >= 0 ? job.dense_level : 0];
        }
        if (job.dense_level >= 0) {
            dense_begin = next.begin;
            dense_count = next.count;
        } else {
            dense_count = 0;        // shadow rays only (the next bounce's nodes go to the frame kernel: hybrid pipeline)
        }
    }
    const long long dense_batches = (dense_count + bs - 1) >> sh;
    const long long n_batches = dense_batches + seg_batches;
    // Logical batch order: the dense part (closest-hit rays) first, the shadow segments after it.  Logical batch L
    // belongs to queue shard L % SHARDS; each shard has its own head on its own cache line (one head word serves
    // ~150 returning atomics per us).  A wavefront draws from its home shard (workgroup id mod shards) first.
    // (Handing out several batches per pop was measured: the heads get cheaper, the tail gets longer, and the
    // launch slower.  8 shards -- one per XCD -- against 64: the 3-D scene, whose batches are short, 0.83 -> 0.63 ms
    // a frame, C3 1.67 -> 1.62.)
    // When the shard a wavefront draws from is drained it reads every head once (lane i reads head i: one round
    // trip) and moves straight to the next shard that still holds batches; it exits when none does.  (With one
    // probing atomic per drained shard the end of a launch cost up to shards - 1 round trips per wavefront.)
    // The end of a launch: a wavefront cannot move to another SIMD, so when the queue runs dry the SIMDs whose three
    // wavefronts each took one of the last batches run them three to a SIMD -- three batch times -- while most of the
    // chip idles (that, not the content of the batches, is the tail of a launch: exact longest-first order did not
    // shorten it).  So the last `tail_solo` batches of every shard are left to ONE wavefront per SIMD (workgroup
    // wavefronts w, w+4, w+8 share a SIMD -- NDT_HIP_EXIT_PROBE prints the HW_ID census that shows it: the ones with
    // w >= 4 stay away from them and leave early).
    const int reserve = ((threadIdx.x >> 6) >= 4) ? job.tail_solo : 0;
    int cur = blockIdx.x % NDT_QUEUE_SHARDS;                    // home shard
    long long last_k = -1;                                      // this wavefront's last pop from `cur`
    bool any_left = true;
    // A ticket drawn ahead (round 4; global-memory tier only -- measured: 6-D 1.263 -> 1.231 ms, 7-D / 8-D equal, but the LDS tiers
    // slower: benchmark frame 1.253 -> 1.275, balls 0.792 -> 0.854): while the shard is far from its end (NDT_TICKET_AHEAD tickets: every wavefront that draws
    // from it could take several before it runs dry), the next pop's atomic is issued ahead of this batch's ray loads and travels
    // with them -- a pop was a round trip of its own, ~2 us of a 10-40 us batch.  Near the end nothing is held back: a ticket in the
    // pocket of a wavefront in a long batch would be a batch nobody else can start.
    int k_ahead = -1;
    while (true) {
        long long b = -1;           // logical batch
        if (k_ahead >= 0) {
            const int k = __shfl(k_ahead, 0, 64);
            const long long n_here = (n_batches - cur + NDT_QUEUE_SHARDS - 1) / NDT_QUEUE_SHARDS;
            k_ahead = -1;
            if (k < n_here) {
                b = (long long)k * NDT_QUEUE_SHARDS + cur;
                last_k = k;
            }
        }
        while (any_left && b < 0) {
            const long long n_here = (n_batches - cur + NDT_QUEUE_SHARDS - 1) / NDT_QUEUE_SHARDS;     // batches of this shard
            bool may = true;
            if (reserve > 0 && last_k + 1 + 4 * reserve >= n_here) {
                // near the shard's end: look before taking (what is taken cannot be given back)
                const int head_cur = __hip_atomic_load(job.queue + cur * NDT_QUEUE_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                may = (long long)head_cur < n_here - reserve;
            }
            if (may) {
                int k = 0;
                if (lane == 0) k = atomicAdd(job.queue + cur * NDT_QUEUE_STRIDE, 1);
                k = __shfl(k, 0, 64);
                if (k < n_here) {
                    b = (long long)k * NDT_QUEUE_SHARDS + cur;
                    last_k = k;
                    break;
                }
            }
            int head = 0x7fffffff;
            if (lane < NDT_QUEUE_SHARDS)
                head = __hip_atomic_load(job.queue + lane * NDT_QUEUE_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long mine = (n_batches - lane + NDT_QUEUE_SHARDS - 1) / NDT_QUEUE_SHARDS - reserve;
            const unsigned long long live = __ballot(lane < NDT_QUEUE_SHARDS && (long long)head < mine);

            if (live == 0ull) {
                any_left = false;
            } else {
                // the next live shard after `cur`, cyclically
                const unsigned long long above = (cur + 1 < 64) ? (live >> (cur + 1)) << (cur + 1) : 0ull;
                cur = __ffsll((long long)(above ? above : live)) - 1;
                last_k = (long long)__shfl(head, cur, 64) - 1;      // where that shard's head stood a moment ago
            }
        }
        if (b < 0) break;
#ifndef NDT_NO_TICKET_AHEAD
        if (MW == 0 && last_k + NDT_TICKET_AHEAD < (n_batches - cur + NDT_QUEUE_SHARDS - 1) / NDT_QUEUE_SHARDS) {
            // (every lane is here; the atomic travels with this batch's ray loads and is back when they are)
            k_ahead = 0;
            if (lane == 0) k_ahead = atomicAdd(job.queue + cur * NDT_QUEUE_STRIDE, 1);
        }
#endif
        if (job.exit_log) probe_batch = (unsigned int)wall_clock64();
        TraceAbandon ab{};
        if (COOP && coop_on) {
            ab.tail = job.queue + NDT_COOP_TAIL;
            ab.deadline = wall_clock64() + (unsigned long long)job.coop_budget;
            ab.max_live = job.coop_max_live;
            ab.tail_limit = job.coop_limit;
            if (job.coop_tail_only) {
                // the shard this batch came from: dry when its head has passed its last batch
                ab.dry_word = job.queue + cur * NDT_QUEUE_STRIDE;
                ab.dry_at = (int)((n_batches - cur + NDT_QUEUE_SHARDS - 1) / NDT_QUEUE_SHARDS);
            }
        }
        long long g;
        bool live;
        const bool in_seg = b >= dense_batches;         // wave-uniform
        int origin_word = -1;                           // >= 0: every ray of this batch starts at the light whose position stands there
        if (in_seg) {
            // segment of shadow batch sb = number of segments whose inclusive prefix is <= sb
            const int sb = (int)(b - dense_batches);
            const int s = __popcll(__ballot(lane < job.n_seg && seg_batches_incl <= sb));
            if (MW != 0 && job.seg_light_origins) origin_word = seg_light_origin(gblob, sd, __builtin_amdgcn_readfirstlane(s));
            const int first = (s > 0) ? __shfl(seg_batches_incl, s - 1, 64) : 0;
            const int cnt = __shfl(seg_cnt, s, 64);
            const int idx = (sb - first) * bs + lane;
            live = lane < bs && idx < cnt;
            g = (long long)s * seg_stride + (live ? idx : 0);
        } else {
            const long long r = b * bs + lane;
            live = lane < bs && r < dense_count;
            g = dense_begin + (live ? r : 0);
            if (!PRIM && live && job.dense.valid && job.dense.valid[g] <= 0) live = false;
        }
        // The primaries of a pass are made HERE, by the wavefront that traces them (render_pixel + the head of get_pixel_color,
        // ndt.c:578-653, 488-550: primary_node -- what the k_primary launch used to do: 2N doubles a ray written and read back
        // before anything was traced); the node's record still goes to the pool for the kernels behind this one.
        double o[N], v[N];
        const bool make = PRIM && !in_seg;                      // (wave-uniform)
        if (PRIM && make && live) live = primary_node<false, true>(gblob, sd, ws, job.rg, g, o, v);
        // A lane without a ray.  The LDS tiers leave it out of the batch; in the global-memory tier it stays with the
        // wavefront as a helper of the coherent leaf scan (ndt_device.hpp:cls_scan: all 64 lanes fetch), with a ray that is
        // finished before it starts.
        bool gave_up = false;
        if (live || MW == 0) {
        const TracePart &part = in_seg ? job.seg : job.dense;
#ifdef NDT_TRACE_SKIP_KNOB
        if (job.skip_trace == 3) {
            for (int c = 0; c < N; ++c) { o[c] = (double)g; v[c] = 1.0; }
        } else
#endif
        if (live) {
            if (!make) {
                if (origin_word >= 0) blob_vec<N>(gblob, origin_word, o);
                else load_soa<N>(part.o, part.stride, g, o);
                load_soa<N>(part.v, part.stride, g, v);
            }
        } else {
#pragma unroll
            for (int c = 0; c < N; ++c) { o[c] = 0.0; v[c] = 0.0; }
        }
        const double lim = (live && part.lim) ? part.lim[g] : -1.0;
        int obj, prim;
#ifdef NDT_PHASE_TIMING
        unsigned int cnt[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        unsigned int occ[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        ph[6] += __builtin_readcyclecounter() - out_last;
        batch_mark = wall_clock64();
        if (job.skip_trace) {
            obj = (o[0] + v[0] + lim > 1e300) ? 0 : -1;     // keeps the loads alive
            prim = -1;
        } else
        trace_kd<N, MW, LSTACK, MW == 0>(blob, sd, mask, o, v, lim, obj, prim, ph, cnt, occ, kstack, cls, live, box_slot);
        out_last = __builtin_readcyclecounter();
        if (ws.dbg && lane == __ffsll((long long)__ballot(1)) - 1) {
            // duration of this batch inside trace_kd: slowest batch of the launch (dbg[40 + is_shadow])
            const unsigned long long now = wall_clock64();      // constant 100 MHz
            atomicMax(&ws.dbg[40 + (int)in_seg], now - batch_mark);
            atomicAdd(&ws.dbg[42 + (int)in_seg], now - batch_mark);
            atomicAdd(&ws.dbg[44 + (int)in_seg], 1ull);
        }
        {
            // every active lane holds the same per-wave occupancy numbers: the lowest active lane keeps them
            const unsigned long long act = __ballot(1);
            if (lane == __ffsll((long long)act) - 1)
                for (int i = 0; i < 8; ++i) acc[(in_seg ? 16 : 8) + i] += occ[i];
            for (int i = 0; i < 6; ++i) acc[i] += cnt[i];
            acc[6] += 1ull;
            if (ws.dbg) {
                // per-ray maxima: node visits, gates, intersections; and the same summed over the slowest lane's wave
                unsigned int nv = cnt[0], ng = cnt[1] + cnt[3], ni = cnt[2] + cnt[4];
                for (int d = 32; d > 0; d >>= 1) {
                    nv = max(nv, (unsigned int)__shfl_xor((int)nv, d, 64));
                    ng = max(ng, (unsigned int)__shfl_xor((int)ng, d, 64));
                    ni = max(ni, (unsigned int)__shfl_xor((int)ni, d, 64));
                }
                if (lane == __ffsll((long long)__ballot(1)) - 1) {
                    atomicMax(&ws.dbg[46], (unsigned long long)nv);
                    atomicMax(&ws.dbg[47], (unsigned long long)ng);
                    atomicMax(&ws.dbg[48], (unsigned long long)ni);
                    atomicMax(&ws.dbg[49], (unsigned long long)occ[0]);     // T / G / I loop iterations of one batch
                    atomicMax(&ws.dbg[50], (unsigned long long)occ[2]);
                    atomicMax(&ws.dbg[51], (unsigned long long)occ[4]);
                }
            }
        }
#else
#ifdef NDT_TRACE_SKIP_KNOB
        if (job.skip_trace) {       // diagnostic build: what a launch costs without the traversal
            obj = (o[0] + v[0] + lim > 1e300) ? 0 : -1;
            prim = -1;
        } else
#endif
        trace_kd<N, MW, LSTACK, MW == 0>(blob, sd, mask, o, v, lim, obj, prim, kstack, cls, live, box_slot, ab, &gave_up);
#endif
#ifdef NDT_TRACE_SKIP_KNOB
        if (job.skip_trace == 2 && obj == -1) live = false;
#endif
        if (live && !gave_up) {
            part.out_obj[g] = obj;
            part.out_prim[g] = prim;
        }
        }
        if (COOP && coop_on) {
            // the rays this batch gave up: into the straggler ring, one reservation per wavefront
            const unsigned long long gv = __ballot(live && gave_up);
            if (gv != 0ull) {
                int base = 0;
                if (lane == 0) base = atomicAdd(job.queue + NDT_COOP_TAIL, __popcll(gv));
                base = __shfl(base, 0, 64);
                if (live && gave_up) {
                    const unsigned int payload = (unsigned int)g | (in_seg ? 0x80000000u : 0u);
                    __hip_atomic_store(job.coop_ring + base + __popcll(gv & ((1ull << lane) - 1ull)),
                                       ((unsigned long long)job.coop_tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (job.coop_log && lane == 0) atomicAdd(job.coop_log, (unsigned int)__popcll(gv));
            }
        }
    }
    const unsigned int probe_left = job.exit_log ? (unsigned int)wall_clock64() : 0u;
    if (COOP && coop_on) {
        // ---- this wavefront has no batch left.  It says so; the LAST one of the launch to say so closes the straggler ring:
        // every ray that will ever be given up has been by then (a wavefront's pushes come before its own leaving), and each
        // consumer holds exactly one ticket beyond the ring's final tail -- a closing entry goes into each of those slots.
        const int waves_per_block = blockDim.x >> 6;
        const int consumers_per_block = waves_per_block < job.coop_waves ? waves_per_block : job.coop_waves;
        const int n_consumers = gridDim.x * consumers_per_block;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the ring entries above are out
        int closer = 0;
        if (lane == 0) {
            const int grp = blockIdx.x % NDT_COOP_GROUPS;
            const int grp_blocks = ((int)gridDim.x - grp + NDT_COOP_GROUPS - 1) / NDT_COOP_GROUPS;
            const int old = atomicAdd(job.queue + NDT_COOP_LEFT + grp * NDT_QUEUE_STRIDE, 1);
            if (old + 1 == grp_blocks * waves_per_block) {
                const int n_groups = (int)gridDim.x < NDT_COOP_GROUPS ? (int)gridDim.x : NDT_COOP_GROUPS;
                closer = atomicAdd(job.queue + NDT_COOP_GROUPS_DONE, 1) + 1 == n_groups;
            }
        }
        closer = __shfl(closer, 0, 64);
        if (closer) {
            int tail = 0;
            if (lane == 0) tail = __hip_atomic_load(job.queue + NDT_COOP_TAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tail = __shfl(tail, 0, 64);
            for (int j = lane; j < n_consumers; j += 64)
                __hip_atomic_store(job.coop_ring + tail + j, ((unsigned long long)job.coop_tag << 32) | NDT_COOP_CLOSE, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- consumers: one wavefront per SIMD (the first four of a workgroup) take the stragglers, one ray at a time
        if ((int)(threadIdx.x >> 6) < consumers_per_block) {
            const unsigned long long t_enter = wall_clock64();
            unsigned int n_done = 0, t_coop = 0;
            int ticket = -1, polls = 0;
            while (true) {
                unsigned long long e = 0ull;
                if (lane == 0) {
                    if (ticket < 0) ticket = atomicAdd(job.queue + NDT_COOP_HEAD, 1);
                    e = __hip_atomic_load(job.coop_ring + ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                ticket = __shfl(ticket, 0, 64);
                const unsigned int e_tag = (unsigned int)__shfl((int)(e >> 32), 0, 64), payload = (unsigned int)__shfl((int)e, 0, 64);
                if (e_tag != job.coop_tag) {
                    // not written yet (a slot beyond the tail is written when the launch closes)
                    __builtin_amdgcn_s_sleep(4);
                    if ((++polls & 1023) == 0 && wall_clock64() - t_enter > 200000000ull) break;   // 2 s: never on a healthy launch
                    continue;
                }
                if (payload == NDT_COOP_CLOSE) break;
                const unsigned long long t0 = job.coop_log ? wall_clock64() : 0ull;
                const bool in_seg = (payload >> 31) != 0u;
                const long long g = (long long)(payload & 0x7fffffffu);
                const TracePart &part = in_seg ? job.seg : job.dense;
                double o[N], v[N];
                // (a launch with the ring never makes its own primaries: what it is given up was stored by an earlier kernel)
                const int origin_word = (in_seg && job.seg_light_origins) ? seg_light_origin(gblob, sd, (int)(g / seg_stride)) : -1;
                if (origin_word >= 0) blob_vec<N>(gblob, origin_word, o);
                else load_soa<N>(part.o, part.stride, g, o);
                load_soa<N>(part.v, part.stride, g, v);
                const double lim = part.lim ? part.lim[g] : -1.0;
                // the next ticket travels while this ray is traced (issued behind the ray's loads: memory operations return in order)
                ticket = 0;
                if (lane == 0) ticket = atomicAdd(job.queue + NDT_COOP_HEAD, 1);
                int obj, prim;
                coop_trace<N>(blob, sd, o, v, lim, obj, prim);
                if (lane == 0) {
                    part.out_obj[g] = obj;
                    part.out_prim[g] = prim;
                }
                if (job.coop_log) {
                    ++n_done;
                    t_coop += (unsigned int)(wall_clock64() - t0);
                }
            }
            if (job.coop_log && lane == 0 && n_done) {
                atomicAdd(job.coop_log + 1, n_done);
                atomicAdd(job.coop_log + 2, t_coop);
            }
        }
    }
    if (job.exit_log && lane == 0) {
        // one private slot per wavefront: shared counters would serialise the very exits they measure
        const unsigned int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        if (8 * w + 7 < NDT_EXIT_LOG_WORDS) {
            job.exit_log[8 * w] = probe_start;
            job.exit_log[8 * w + 1] = probe_batch;
            job.exit_log[8 * w + 2] = (unsigned int)wall_clock64() | 1u;
            job.exit_log[8 * w + 3] = (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffffu) | ((blockDim.x / 64) << 24);   // HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8]; [31:24] wavefronts per workgroup
            job.exit_log[8 * w + 4] = probe_left | 1u;      // out of batches (with a straggler ring the wavefront stays on as a consumer)
        }
    }
#ifdef NDT_PHASE_TIMING
    ph[6] += __builtin_readcyclecounter() - out_last;
    if (ws.dbg) {
        for (int i = 0; i < 24; ++i) {
            unsigned long long x = acc[i];
            for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d, 64);
            if (lane == 0 && x) atomicAdd(&ws.dbg[8 + i], x);
        }
        if (lane == 0) {
            for (int i = 0; i < 4; ++i) atomicAdd(&ws.dbg[i], ph[i]);
            atomicAdd(&ws.dbg[7], ph[7]);           // coherent leaf scan: gates / boxes of a window
            atomicAdd(&ws.dbg[32], ph[4]);          // ... fetching a window
            atomicAdd(&ws.dbg[4], 1ull);
            atomicAdd(&ws.dbg[5], ph[5]);
            atomicAdd(&ws.dbg[6], ph[6]);
        }
    }
#endif
}

// launch knobs of the trace kernel: environment variables in the diagnostic builds (make timing | skipknob), the defaults
// -- compile-time constants -- in the library
#if defined(NDT_PHASE_TIMING) || defined(NDT_TRACE_SKIP_KNOB)
static int env_int(const char *name, int def)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : def;
}
#else
static constexpr int env_int(const char *, int def) { return def; }
#endif

// Workgroups of `kernel` that are resident on the device at once (occupancy x CUs).  Asked once per (device, kernel
// variant, workgroup size, LDS bytes) and remembered: a frame has five trace launches, the query takes longer than a
// launch, and contexts on several threads (ndt_hip -j K) share the table.
template <typename K> static int resident_blocks(K kernel, int block, size_t lds)
{
    static std::mutex mu;
    static std::map<std::tuple<int, const void *, int, size_t>, int> known;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto key = std::make_tuple(dev, reinterpret_cast<const void *>(kernel), block, lds);
    std::lock_guard<std::mutex> lock(mu);
    auto it = known.find(key);
    if (it != known.end()) return it->second;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    int cus = 256, v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    known[key] = per_cu * cus;
    return per_cu * cus;
}

// ev_start / ev_stop (both or neither): HIP events that take the kernel's own dispatch timestamps
// (hipExtLaunchKernelGGL) -- two hipEventRecord packets around the launch cost ~6 us of stream time each.
#define NDT_LAUNCH_TRACE(kernel, grid, block, lds)                                                             \
    do {                                                                                                       \
        if (ev_start)                                                                                          \
            hipExtLaunchKernelGGL((kernel), dim3((unsigned)(grid)), dim3(block), (std::uint32_t)(lds), s, ev_start, ev_stop, 0u, blob, sd, ws, job); \
        else                                                                                                   \
            hipLaunchKernelGGL((kernel), dim3((unsigned)(grid)), dim3(block), lds, s, blob, sd, ws, job);      \
    } while (0)
typedef void (*TraceKernel)(const double *, SceneDesc, Workspace, TraceJob);
static void launch_trace(hipStream_t s, const double *blob, SceneDesc sd, Workspace ws, TraceJob job, int tier,
                         int mask_words, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (job.count <= 0 && job.n_seg <= 0) {
        if (ev_start) {         // nothing to launch: the events still have to be valid for the caller's elapsed-time query
            (void)hipEventRecord(ev_start, s);
            (void)hipEventRecord(ev_stop, s);
        }
        return;
    }
    static const int block = env_int("NDT_TRACE_BLOCK", NDT_TRACE_BLOCK);
    static const int force_batch = env_int("NDT_TRACE_BATCH", 0);
    job.batch = 64;
    job.skip_trace = env_int("NDT_TRACE_SKIP", 0);
    {
        static const int tail_solo = env_int("NDT_TRACE_TAIL_SOLO", 16);
        job.tail_solo = tail_solo > 0 ? tail_solo : 0;
    }
    if (force_batch == 64 || force_batch == 32 || force_batch == 16 || force_batch == 8) job.batch = force_batch;
    const long long upper = job.count + (job.n_seg > 0 ? job.seg_stride * job.n_seg : 0);      // sizes the grid only
    long long blocks = (upper + job.batch * (block / 64) - 1) / (job.batch * (block / 64));
    if (tier == 0) {
        static const int extra_lds = env_int("NDT_TRACE_EXTRA_LDS", 0);   // experiment knob: lowers occupancy
        const size_t lds = (size_t)sd.trace_words * sizeof(double) + (size_t)extra_lds;
        // Traversal stack in LDS when it fits beside the scene: one workgroup of 768 lanes per CU (the scene is
        // staged once instead of three times), 12 bytes per level and lane.  NDT_TRACE_LSTACK=0 switches back
        // to the scratch stack with NDT_TRACE_BLOCK-sized workgroups (also what deeper trees / bigger scenes get).
        static const int lstack_full = env_int("NDT_TRACE_LSTACK", NDT_TRACE_MAX_BLOCK);
        static const int small_launch = env_int("NDT_TRACE_SMALL", 4096);     // batches below which a launch counts as small
        // A launch with fewer batches than the chip has wavefront slots is pure latency: give every wavefront a
        // SIMD of its own (256-lane workgroups spread over the CUs) instead of packing twelve into one CU.
        int lstack_block = lstack_full;
        if (lstack_full >= 256 && upper / job.batch < small_launch) lstack_block = 256;
        const size_t lds_stack = ((size_t)((sd.trace_words + 1) & ~1) * 8) + (size_t)lstack_block * (sd.kd_depth + 1) * 12;
        // the kernel's variant: with the straggler ring (item sets only), making its own primaries, or plain
        const bool ring = job.coop_ring != nullptr && mask_words <= 1;
        if (!ring) job.coop_ring = nullptr;
        const bool prim = job.make_primaries != 0 && !ring;
        if (lstack_block >= 64 && mask_words <= 1 && lds_stack <= 160 * 1024) {
            const TraceKernel kern = ring ? k_trace<1, true, true, true, false> : prim ? k_trace<1, true, true, false, true> : k_trace<1, true, true>;
            const int res = resident_blocks(kern, lstack_block, lds_stack);
            long long nb = (upper + job.batch * (lstack_block / 64) - 1) / (job.batch * (lstack_block / 64));
            if (nb > res) nb = res;
            NDT_LAUNCH_TRACE(kern, nb, lstack_block, lds_stack);
        } else if (lstack_block >= 64 && mask_words > 1 && lds_stack <= 160 * 1024) {
            // scenes of 65 .. 256 objects (a register mask of four words): the same, with the wider mask
            const TraceKernel kern = prim ? k_trace<NDT_MASK_REG_WORDS, true, true, false, true> : k_trace<NDT_MASK_REG_WORDS, true, true>;
            const int res = resident_blocks(kern, lstack_block, lds_stack);
            long long nb = (upper + job.batch * (lstack_block / 64) - 1) / (job.batch * (lstack_block / 64));
            if (nb > res) nb = res;
            NDT_LAUNCH_TRACE(kern, nb, lstack_block, lds_stack);
        } else if (mask_words <= 1) {
            const TraceKernel kern = ring ? k_trace<1, true, false, true, false> : prim ? k_trace<1, true, false, false, true> : k_trace<1, true>;
            const int res = resident_blocks(kern, block, lds);
            if (blocks > res) blocks = res;
            NDT_LAUNCH_TRACE(kern, blocks, block, lds);
        } else {
            const TraceKernel kern = prim ? k_trace<NDT_MASK_REG_WORDS, true, false, false, true> : k_trace<NDT_MASK_REG_WORDS, true>;
            const int res = resident_blocks(kern, block, lds);
            if (blocks > res) blocks = res;
            NDT_LAUNCH_TRACE(kern, blocks, block, lds);
        }
    } else {
        job.coop_ring = nullptr;
        const TraceKernel kern_t1 = job.make_primaries ? k_trace<0, false, false, false, true> : k_trace<0, false>;
        // (coherent leaf scan: one LDS window per wavefront)
        const size_t lds = (size_t)(block / 64) * ((sd.off_obox > 0 ? N * 128 : 0) + (sd.cls_par_words > 0 ? cls_window_words<N>(sd.cls_par_words) : 0)) * sizeof(double);
        const int res = resident_blocks(kern_t1, block, lds);
        if (blocks > res) blocks = res;
        const long long max_blocks = ws.mask_slab_lanes / block;
        if (blocks > max_blocks) blocks = max_blocks;
        NDT_LAUNCH_TRACE(kern_t1, blocks, block, lds);
    }
}

// ------------------------------------------------------------------ shading, first half
#ifndef NDT_SHADE_WAVES
// wavefronts per SIMD the lighting kernels are compiled for (256 registers; three -- 168 registers with the intersection out of
// line -- measured 2-6 % slower on the 3-D and 4-D scenes)
#define NDT_SHADE_WAVES 2
#endif

// isect<N, true> as ONE real function for the places of the shading code that re-run a primitive for its hit point
// and normal (shading: the primitive the traversal returned; lighting: the primitive the shadow ray met).  Inlined, each
// copy was ~10 KB of code in the middle of a function that already holds four N-vectors: the arguments now travel through
// the wavefront's private memory (4N doubles per call, twice per node) and the register allocator sees two small
// functions instead of one that cannot fit.
struct VecPair {
    double a[N], b[N];
};
// (arguments and results by value: they travel in registers as far as 32 of them go, then through the stack; by pointer the
// caller's vectors would live in private memory altogether)
#if NDT_DIMS >= 6 && !defined(NDT_STREAM_INLINE_ISECT)
#define NDT_HAVE_ISECT_CALL 1
__device__ __attribute__((noinline)) VecPair isect_full_call(const double *blob, const SceneDesc *sd, int prim, VecPair ray)
{
    VecPair out;
#pragma unroll
    for (int c = 0; c < N; ++c) { out.a[c] = 0.0; out.b[c] = 0.0; }
    isect<N, true>(blob, *sd, prim, ray.a, ray.b, out.a, out.b);
    return out;
}
NDT_DEV void isect_full_outlined(const double *blob, const SceneDesc *sd, int prim, const double (&o)[N], const double (&v)[N],
                                 double (&hit)[N], double (&nrm)[N])
{
    VecPair ray;
#pragma unroll
    for (int c = 0; c < N; ++c) { ray.a[c] = o[c]; ray.b[c] = v[c]; }
    const VecPair out = isect_full_call(blob, sd, prim, ray);
#pragma unroll
    for (int c = 0; c < N; ++c) { hit[c] = out.a[c]; nrm[c] = out.b[c]; }
}
#endif
// the lighting kernels of the per-bounce pipeline: inline up to 7-D (NDT_SHADE_INLINE_ISECT, top of this file)
NDT_DEV void isect_full(const double *blob, const SceneDesc *sd, int prim, const double (&o)[N], const double (&v)[N], double (&hit)[N],
                        double (&nrm)[N])
{
#if defined(NDT_SHADE_INLINE_ISECT) || !defined(NDT_HAVE_ISECT_CALL)
    isect<N, true>(blob, *sd, prim, o, v, hit, nrm);
#else
    isect_full_outlined(blob, sd, prim, o, v, hit, nrm);
#endif
}
// the frame kernel (ndt_stream.hpp): a real function from 6-D on (156 -> 65 KB of code, fewer spills: round 2's measurement)
NDT_DEV void isect_full_stream(const double *blob, const SceneDesc *sd, int prim, const double (&o)[N], const double (&v)[N],
                               double (&hit)[N], double (&nrm)[N])
{
#ifdef NDT_HAVE_ISECT_CALL
    isect_full_outlined(blob, sd, prim, o, v, hit, nrm);
#else
    isect<N, true>(blob, *sd, prim, o, v, hit, nrm);
#endif
}



#ifdef NDT_PHASE_TIMING
// diagnostic build: wall-clock (100 MHz) stamps of lane 0 of every wavefront, summed per section
#define NDT_SEC_BEGIN() unsigned long long sec_t[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; unsigned long long sec_last = wall_clock64(); const unsigned long long sec_first = sec_last
#define NDT_SEC(k) do { const unsigned long long now_ = wall_clock64(); sec_t[k] += now_ - sec_last; sec_last = now_; } while (0)
#define NDT_SEC_END(base) do { if (ws.dbg && __lane_id() == 0) { for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&ws.dbg[(base) + i_], sec_t[i_]); \
        atomicAdd(&ws.dbg[(base) + 6], 1ull); atomicMax(&ws.dbg[(base) + 7], wall_clock64() - sec_first); } } while (0)
#else
#define NDT_SEC_BEGIN() do { } while (0)
#define NDT_SEC(k) do { } while (0)
#define NDT_SEC_END(base) do { } while (0)
#endif

struct LightRec {
    int type;
    double red, green, blue, angle;
};
NDT_DEV int light_word(const SceneDesc &sd, int i) { return sd.off_lights + i * (6 + 4 * N); }

// Per light: everything apply_lights does before its trace_kd call (ndt.c:113-208, 230-236).
// Returns false when the light is skipped for this hit (ambient, wrong side, outside the cone).
struct ShadowSetup {
    double ldist2, dist_limit;
};
NDT_DEV bool light_setup(const double *blob, const SceneDesc &sd, int li, const double (&src)[N], const double (&hit)[N],
                         const double (&hit_normal)[N], int &type, double (&lgt_pos)[N], double (&rev_light)[N],
                         double (&light_vec)[N], double (&shadow_o)[N], ShadowSetup &ss, unsigned long long key = 0ull)
{
    const int w = light_word(sd, li);
    type = blob_int(blob, w, 0);
    double ldir[N], rev_view[N];
    blob_vec<N>(blob, w + 5, lgt_pos);
    if (type == 4 || type == 5) {
        // LIGHT_DISK / LIGHT_RECT: a random point of the light, then a point light (ndt.c:116-147).
        // shade_emit and shade_finish call this with the same node key and so see the same point.
        const unsigned long long lk = ndt_rng_mix(key ^ (0x51ed270b27b4f3cfull * (unsigned long long)(li + 1)));
        double x, y;
        unsigned int k = 0;
        do {
            x = 2 * ndt_rng_uniform(lk, k) - 1.0;
            y = 2 * ndt_rng_uniform(lk, k + 1) - 1.0;
            k += 2;
        } while (type == 4 && x * x + y * y > 1.0 && k < 64);
        const double radius = blob[w + 5 + 2 * N];
        double ax[N], temp[N];
        blob_vec<N>(blob, w + 6 + 2 * N, ax);
        v_scale<N>(ax, x * radius, temp);
        v_add<N>(lgt_pos, temp, lgt_pos);
        blob_vec<N>(blob, w + 6 + 3 * N, ax);
        v_scale<N>(ax, y * radius, temp);
        v_add<N>(lgt_pos, temp, lgt_pos);
        type = NDT_LIGHT_POINT_;
    }
    if (type != NDT_LIGHT_POINT_ && type != NDT_LIGHT_DIRECTIONAL_ && type != NDT_LIGHT_SPOT_) return false;
    blob_vec<N>(blob, w + 5 + N, ldir);
    if (type == NDT_LIGHT_DIRECTIONAL_)
        v_scale<N>(ldir, -1, rev_light);                    // ndt.c:157
    else
        v_sub<N>(lgt_pos, hit, rev_light);                  // ndt.c:155
    v_unitize<N>(rev_light);
    v_sub<N>(src, hit, rev_view);
    const double dotRev1 = v_dot<N>(rev_light, hit_normal);
    const double dotRev2 = v_dot<N>(rev_view, hit_normal);
    if ((dotRev1 * dotRev2) <= 0) return false;             // ndt.c:164
    // (dist_limit / ldist2 go through two plain locals and are stored into `ss` once, below: with a store per branch the
    // compiler merged stores to DIFFERENT fields into one store at a run-time offset, which put `ss` in scratch)
    double dist_limit, ldist2;
    double near_pos[N];
    v_zero<N>(near_pos);
    if (type == NDT_LIGHT_DIRECTIONAL_) {
        dist_limit = 0.0;
        ldist2 = 1.0;
        v_copy<N>(near_pos, ldir);                          // ndt.c:232-235
        v_unitize<N>(near_pos);
        v_scale<N>(near_pos, -NDT_EPS, near_pos);
        v_add<N>(near_pos, hit, near_pos);
        v_copy<N>(light_vec, ldir);                         // what light_vec holds after a miss, ndt.c:252
    } else {
        dist_limit = v_dist<N>(hit, lgt_pos);               // ndt.c:187-188
        dist_limit += NDT_EPS;
        v_sub<N>(hit, lgt_pos, light_vec);                  // ndt.c:194-197
        ldist2 = v_dot<N>(light_vec, light_vec);
        v_unitize<N>(light_vec);
        if (type == NDT_LIGHT_SPOT_) {
            const double angle = v_angle<N>(ldir, light_vec);
            if ((angle * 180.0 / NDT_PI) > blob[w + 4]) return false;     // ndt.c:204
        }
    }
    // the origin of the shadow ray, selected by VALUE (copies from two different arrays in the two branches end in scratch)
#pragma unroll
    for (int c = 0; c < N; ++c) shadow_o[c] = (type == NDT_LIGHT_DIRECTIONAL_) ? near_pos[c] : lgt_pos[c];
    ss.dist_limit = dist_limit;
    ss.ldist2 = ldist2;
    return true;
}

// What the wavefronts of one shade_emit workgroup reserve together: every wavefront adds its wants here (LDS atomics,
// which also hand it its offset inside the workgroup's share), ONE wavefront turns the sums into global reservations, and
// the bases come back through LDS.  A global counter takes ~150 returning atomics per us; with a reservation per wavefront
// the 32 k wavefronts of a frame whose primaries all hit something queued on two counters for 320 us (the whole
// shade_emit(0) of the 3-D scene), each wavefront alive for 40 us waiting its turn.
struct EmitShared {
    int spawn_total, spawn_base;
    int seg_total[64], seg_base[64];
};

// (every thread of the workgroup calls this: it holds two workgroup barriers)
NDT_DEV void shade_emit_node(const double *blob, const SceneDesc &sd, const Workspace &ws, const RenderGeom &rg,
                             const LevelRange &lr, int level, long long r, EmitShared *sh)
{
    const bool in_range = r < lr.count;
    const long long g = lr.begin + (in_range ? r : 0);
    bool shaded = false;
    double src[N], look[N], hit[N], nrm[N];
    int obj = -1;
    NDT_SEC_BEGIN();
    // The node's three words in flight TOGETHER (round 4): read one after the other -- is the node alive? -> what did it hit? ->
    // which primitive? -- they were three round trips to HBM in a row at the head of a wavefront that lives for 6 us.
    int dl = 0, obj_in = -1, prim_in = -1;
    if (in_range) {
        dl = ws.depth_left[g];
        obj_in = ws.hit_obj[g];
        prim_in = ws.hit_prim[g];
        asm volatile("" : "+v"(dl), "+v"(obj_in), "+v"(prim_in));
    }
    if (in_range && dl > 0) {
        obj = obj_in;
        if (obj >= 0) {
            load_soa<N>(ws.ray_o, ws.cap, g, src);
            load_soa<N>(ws.ray_v, ws.cap, g, look);
            // the hit point and normal trace_kd would have returned: re-run the one primitive
            // that won the traversal (same arithmetic, same result)
            isect<N, true>(blob, sd, prim_in, src, look, hit, nrm);      // (always inline here: see NDT_SHADE_INLINE_ISECT)
            const double trace_dist = v_dist<N>(hit, src);                  // ndt.c:365
            shaded = trace_dist > NDT_EPS;                                  // ndt.c:376
            if (rg.want_depth && level == 0) ws.depth[g] = shaded ? 1.0 / trace_dist : 0.0;    // ndt.c:366-370
            // (hit point and normal are not stored: shade_finish makes them again from the winning primitive, like every other
            // consumer of a trace result -- 4N doubles less written and read per node; balls 0.967 -> 0.943 ms, the others +-0)
            if (!shaded) {
                ws.hit_obj[g] = -1;
            }
        }
        if (obj < 0 && rg.want_depth && level == 0) ws.depth[g] = 0.0;     // ndt.c:372-373
        if (!shaded) {
            // background (ndt.c:436-442); alpha is applied per pixel at the end
            ws.clr[0 * ws.cap + g] = blob[sd.off_cam + 4 * N + 4];
            ws.clr[1 * ws.cap + g] = blob[sd.off_cam + 4 * N + 5];
            ws.clr[2 * ws.cap + g] = blob[sd.off_cam + 4 * N + 6];
            ws.count[g] = 1;
        }
    }
    // a wavefront of background only (two thirds of the primaries' wavefronts on the benchmark frame) skips the work
    // below -- no light fires, nothing spawns, every collective would come out empty -- but not the workgroup's barriers
    const bool live = __ballot(shaded) != 0ull;
    // One shadow ray per light that passes the same-side / cone tests.  The queue is segmented
    // by light: the rays a wavefront later traces then share their origin (the light) and aim
    // at neighbouring hit points, instead of interleaving five unrelated origins.  Within a
    // segment live rays are compacted with a wavefront ballot; the reservations of all
    // segments go out as ONE atomic instruction (lane s reserves for segment s), so a wavefront
    // waits for a single atomic round trip however many lights there are.
    const int lane = __lane_id();
    unsigned long long fire = 0ull;
    const unsigned long long node_key = (rg.sample_keys && in_range) ? ws.rng_key[g] : 0ull;    // stochastic renders only
    NDT_SEC(0);
    if (shaded) {
        for (int li = 0; li < sd.n_lights; ++li) {
            int type;
            double lgt_pos[N], rev_light[N], light_vec[N], so[N];
            ShadowSetup ss;
            if (light_setup(blob, sd, li, src, hit, nrm, type, lgt_pos, rev_light, light_vec, so, ss, node_key)) fire |= 1ull << li;
        }
    }
    NDT_SEC(1);
    // lane s learns how many lanes fire segment s's light, and which light that is
    int my_total = 0, seg = 0;
    if (live) {
        for (int li = 0; li < sd.n_lights; ++li) {
            if (blob_int(blob, light_word(sd, li), 0) == NDT_LIGHT_AMBIENT_) continue;     // wave-uniform
            const unsigned long long vote = __ballot((fire >> li) & 1ull);
            if (lane == seg) my_total = __popcll(vote);
            ++seg;
        }
    }
    if (shaded) ws.sh_mask[g] = fire;
    NDT_SEC(2);
    // get_ray_color, ndt.c:381-430: spawn reflection / refraction.  The children depend on the
    // hit only, not on the lighting, so they are created here -- before the shadow rays of this
    // bounce are traced -- and the host traces them in the SAME launch as those shadow rays.
    bool want_refl = false, want_refr = false;
    double refl_ray[N], refr_ray[N];
    double refl_frac = 0, refr_frac = 0;
    int depth_next = 0;
    if (shaded) {
        const int mw = sd.off_mat + 8 * obj;
        const double refl_r = blob[mw + 3], refl_g = blob[mw + 4], refl_b = blob[mw + 5];
        const bool transparent = blob[mw + 7] != 0.0;
        const double frac = ws.frac[g];
        depth_next = dl - 1;
        const double gb2 = (refl_g > refl_b) ? refl_g : refl_b;
        const double contrib = (refl_r > gb2) ? refl_r : gb2;
        int c_refl = -1, c_refr = -1;
        if (contrib > 0 && (refl_r != 0.0 || refl_g != 0.0 || refl_b != 0.0)) {
            refl_frac = contrib * frac;
            // child cut-offs (ndt.c:336-341) return black without tracing
            if (refl_frac < (1.0 / 512.0) || depth_next <= 0) {
                c_refl = -2;
            } else {
                v_reflect<N>(look, nrm, refl_ray, 1.0);
                v_unitize<N>(refl_ray);
                want_refl = true;
            }
        }
        if (transparent) {
            refr_frac = (1 - contrib) * frac;
            if (refr_frac < (1.0 / 512.0) || depth_next <= 0) {
                c_refr = -2;
            } else {
                double nrm_u[N];                                    // vectNd_refract unitizes the normal it is given (vectNd.c:155);
                v_copy<N>(nrm_u, nrm);                              // the shadow rays below still need the original
                v_refract<N>(look, nrm_u, refr_ray, blob[mw + 6]);
                v_unitize<N>(refr_ray);
                want_refr = true;
            }
        }
        ws.child_refl[g] = c_refl;
        ws.child_refr[g] = c_refr;
    }
    // Compact the children of this wavefront into the next bounce: one reservation per
    // wavefront, reflection rays first and refraction rays after them.
    const unsigned long long v_refl = __ballot(want_refl), v_refr = __ballot(want_refr);
    const int n_refl = __popcll(v_refl), total = n_refl + __popcll(v_refr);
    // the workgroup's reservations: wants in, one wavefront asks the global counters, bases out
    int my_off = 0, wave_off = 0;
    if (my_total > 0) my_off = atomicAdd(&sh->seg_total[lane], my_total);
    if (lane == 0 && total > 0) wave_off = atomicAdd(&sh->spawn_total, total);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int want = sh->seg_total[threadIdx.x];
        if (want > 0) sh->seg_base[threadIdx.x] = atomicAdd(&NDT_SEG_COUNTERS(ws, level)[threadIdx.x], want);
        if (threadIdx.x == 0 && sh->spawn_total > 0) sh->spawn_base = atomicAdd(&ws.counters[0], sh->spawn_total);
    }
    __syncthreads();
    const int my_base = (my_total > 0) ? sh->seg_base[lane] + my_off : 0;
    const int base = (total > 0) ? sh->spawn_base + __shfl(wave_off, 0, 64) : 0;
    if (total > 0 && (long long)base + total > ws.cap) {
        // node pool overflow: flag it (the host renders the frame again with a larger pool), spawn nothing
        if (lane == 0) atomicOr(&ws.counters[2], 1);
        want_refl = false;
        want_refr = false;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (want_refl) {
        const long long c = (long long)base + __popcll(v_refl & below);
        store_soa<N>(ws.ray_o, ws.cap, c, hit);
        store_soa<N>(ws.ray_v, ws.cap, c, refl_ray);
        ws.frac[c] = refl_frac;
        ws.depth_left[c] = depth_next;
        ws.child_refl[c] = -1;
        ws.child_refr[c] = -1;
        ws.count[c] = 0;
        ws.sh_mask[c] = 0ull;
        if (rg.sample_keys) ws.rng_key[c] = ndt_rng_mix(node_key ^ 0x1ull);
        ws.child_refl[g] = (int)c;
    }
    if (want_refr) {
        const long long c = (long long)base + n_refl + __popcll(v_refr & below);
        store_soa<N>(ws.ray_o, ws.cap, c, hit);
        store_soa<N>(ws.ray_v, ws.cap, c, refr_ray);
        ws.frac[c] = refr_frac;
        ws.depth_left[c] = depth_next;
        ws.child_refl[c] = -1;
        ws.child_refr[c] = -1;
        ws.count[c] = 0;
        ws.sh_mask[c] = 0ull;
        if (rg.sample_keys) ws.rng_key[c] = ndt_rng_mix(node_key ^ 0x2ull);
        ws.child_refr[g] = (int)c;
    }
    NDT_SEC(3);
    // shadow rays into their segments
    // (every lane walking only the lights it fires -- as the frame kernel does, ndt_stream.hpp -- was measured here: the
    // 1080p frames 2-4 % slower; these launches are throughput-bound and the shuffles cost more than the idle iterations)
    seg = 0;
    for (int li = 0; live && li < sd.n_lights; ++li) {
        const int list_type = blob_int(blob, light_word(sd, li), 0);        // (wave-uniform)
        if (list_type == NDT_LIGHT_AMBIENT_) continue;
        // a point or spot light's shadow rays start at the light: the trace launch takes the origin from the light's record
        // (TraceJob::seg_light_origins, k_trace) -- 8 N bytes a ray less to write here and to read there
        const bool origin_known = sd.light_origins && (list_type == NDT_LIGHT_POINT_ || list_type == NDT_LIGHT_SPOT_);
        const bool fires = (fire >> li) & 1ull;
        const unsigned long long vote = __ballot(fires);
        const int base = __shfl(my_base, seg, 64);
        if (fires) {
            int type;
            double lgt_pos[N], rev_light[N], light_vec[N], so[N];
            ShadowSetup ss;
            light_setup(blob, sd, li, src, hit, nrm, type, lgt_pos, rev_light, light_vec, so, ss, node_key);
            const int idx = base + __popcll(vote & ((1ull << lane) - 1ull));
            const long long slot = (long long)seg * lr.seg_stride + idx;
            ws.sh_idx[(long long)seg * ws.cap + g] = idx;
            if (!origin_known) store_soa<N>(ws.so, ws.sh_cap, slot, so);
            // point/spot: from the light along light_vec (ndt.c:211); directional: from the
            // nudged hit point along rev_light (ndt.c:238)
            // (one store of a selected VALUE: two stores from different local arrays make the compiler pick the array
            // through a pointer, which puts both in scratch)
            double dir[N];
#pragma unroll
            for (int c = 0; c < N; ++c) dir[c] = (type == NDT_LIGHT_DIRECTIONAL_) ? rev_light[c] : light_vec[c];
            store_soa<N>(ws.sv, ws.sh_cap, slot, dir);
            ws.slim[slot] = ss.dist_limit;
        }
        ++seg;
    }

    NDT_SEC(4);
    NDT_SEC_END(52);
}

// NDT_HIP_SHADE_PROBE: life of every wavefront of one shade launch, in private slots
#define NDT_SHADE_LOG_BEGIN() const unsigned int shade_t0 = ws.shade_log ? (unsigned int)wall_clock64() : 0u
#define NDT_SHADE_LOG_END()                                                                   \
    do {                                                                                      \
        if (ws.shade_log && (threadIdx.x & 63) == 0) {                                        \
            const unsigned int w_ = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;        \
            if (w_ < NDT_SHADE_LOG_WAVES) {                                                   \
                ws.shade_log[2 * w_] = shade_t0;                                              \
                ws.shade_log[2 * w_ + 1] = (unsigned int)wall_clock64() | 1u;                 \
            }                                                                                 \
        }                                                                                     \
    } while (0)

// The bounce's range comes from the device-side table; the grid is sized for the host's upper
// bound, workgroups past the end leave at once.  (A grid-stride loop here cost 50 VGPRs and
// half the occupancy.)
// (3-D .. 5-D: 512-lane workgroups.  shade_emit(0) of a frame whose primaries all hit is started at the dispatcher's ~120
// workgroups per microsecond -- 8 100 of 256 lanes take 68 us to start: hypercube 3-D 0.636 -> 0.617 ms, balls 0.992 -> 0.982.
// 6-D .. 8-D, 244 registers a lane: slower that way, 2.03 against 1.99 ms and 3.45 against 3.39)
#ifndef NDT_EMIT_BLOCK
#if NDT_DIMS <= 5
#define NDT_EMIT_BLOCK 512
#else
#define NDT_EMIT_BLOCK 256
#endif
#endif
__global__ void __launch_bounds__(NDT_EMIT_BLOCK) k_shade_emit(const double *blob, SceneDesc sd, Workspace ws, RenderGeom rg, int level)
{
    __shared__ EmitShared sh;
    const LevelRange lr = ws.levels[level];
    const long long base = (long long)blockIdx.x * blockDim.x;
    if (base < lr.count) {
        NDT_SHADE_LOG_BEGIN();
        if (threadIdx.x < 64) sh.seg_total[threadIdx.x] = 0;
        if (threadIdx.x == 0) sh.spawn_total = 0;
        __syncthreads();
        shade_emit_node(blob, sd, ws, rg, lr, level, base + threadIdx.x, &sh);
        NDT_SHADE_LOG_END();
    }
}

// ------------------------------------------------------------------ shading, second half

NDT_DEV void shade_finish_node(const double *blob, const SceneDesc &sd, const Workspace &ws, const RenderGeom &rg,
                               const LevelRange &lr, int level, long long r, const bool resolve_here = false)
{
    const bool in_range = r < lr.count;
    const long long g = lr.begin + (in_range ? r : 0);
    int obj = -1;
    // (the node's words in flight together, as in shade_emit_node; and the answer of the FIRST shadow ray the lighting below will
    // look at -- sh_mask -> sh_idx -> sobj / sprim, three more round trips in a row -- fetched while the hit point is recomputed)
    int dl = 0, obj_in = -1, prim_in = -1;
    unsigned long long fire = 0ull;
    if (in_range) {
        dl = ws.depth_left[g];
        obj_in = ws.hit_obj[g];
        prim_in = ws.hit_prim[g];
        fire = ws.sh_mask[g];
        asm volatile("" : "+v"(dl), "+v"(obj_in), "+v"(prim_in), "+v"(fire));
    }
    if (in_range && dl > 0) obj = obj_in;
    const bool shaded = obj >= 0;

    if (shaded) {
        double src[N], look[N], nrm[N], hit[N];
        load_soa<N>(ws.ray_o, ws.cap, g, src);
        load_soa<N>(ws.ray_v, ws.cap, g, look);
        const unsigned long long ambient = sd.ambient_bits;
        const unsigned long long fired = fire & ~ambient;
        const int li_first = fired ? __ffsll((long long)fired) - 1 : -1;
        int sobj_first = -1, sprim_first = -1;
        if (li_first >= 0) {
            const int seg = __popcll(~ambient & ((1ull << li_first) - 1ull));
            const long long slot = (long long)seg * lr.seg_stride + ws.sh_idx[(long long)seg * ws.cap + g];
            sobj_first = ws.sobj[slot];
            sprim_first = ws.sprim[slot];
        }
        // the hit point and normal trace_kd would have returned, as in shade_emit: same function, same operands, same bits
        isect_full(blob, &sd, prim_in, src, look, hit, nrm);
        const int mw = sd.off_mat + 8 * obj;
        const double hit_r = blob[mw], hit_g = blob[mw + 1], hit_b = blob[mw + 2];
        const double refl_r = blob[mw + 3], refl_g = blob[mw + 4], refl_b = blob[mw + 5];
        const bool transparent = blob[mw + 7] != 0.0;
        double hitr_r = 0.0, hitr_g = 0.0, hitr_b = 0.0;
        if (rg.specular) {
            hitr_r = refl_r; hitr_g = refl_g; hitr_b = refl_b;
        }
        // apply_lights, ndt.c:88-92: scn->ambient first
        double cr = hit_r * blob[sd.off_cam + 4 * N + 1];
        double cg = hit_g * blob[sd.off_cam + 4 * N + 2];
        double cb = hit_b * blob[sd.off_cam + 4 * N + 3];
        int n_shadow = 0;
        // Every lane walks ITS lights -- the ones that fired a shadow ray for this hit, and the ambient ones, in the list's
        // order (the sums below are taken in that order, ndt.c:98) -- not the whole list: on the benchmark scene a hit fires
        // 1.2 of its five point lights, and a loop over the list ran all five for every wavefront with a fifth of its lanes.
        for (unsigned long long todo = fire | ambient; todo != 0ull; todo &= todo - 1ull) {
            const int li = __ffsll((long long)todo) - 1;
            const int w = light_word(sd, li);
            const int ltype = blob_int(blob, w, 0);
            const double lr_ = blob[w + 1], lg_ = blob[w + 2], lb_ = blob[w + 3];
            if (ltype == NDT_LIGHT_AMBIENT_) {              // ndt.c:106-111
                cr += hit_r * lr_;
                cg += hit_g * lg_;
                cb += hit_b * lb_;
            } else {
            // (if / else and one `lit`, no `continue`: every early continue of a divergent loop is a flag the compiler carries round it)
            const int seg = __popcll(~ambient & ((1ull << li) - 1ull));         // its segment of the shadow queue
            const long long slot = (long long)seg * lr.seg_stride + ws.sh_idx[(long long)seg * ws.cap + g];
            int type;
            double lgt_pos[N], rev_light[N], light_vec[N], so[N], light_hit_normal[N];
            ShadowSetup ss;
            light_setup(blob, sd, li, src, hit, nrm, type, lgt_pos, rev_light, light_vec, so, ss,
                        rg.sample_keys ? ws.rng_key[g] : 0ull);
            const int sobj = (li == li_first) ? sobj_first : ws.sobj[slot];
            const int sprim = (li == li_first) ? sprim_first : ws.sprim[slot];
            ++n_shadow;
            bool lit;
            if (type == NDT_LIGHT_DIRECTIONAL_) {
                lit = sobj < 0;                             // anything at all shadows it, ndt.c:246
                v_copy<N>(light_hit_normal, nrm);           // ndt.c:252-254
            } else {
                lit = sobj == obj;                          // ndt.c:217
                if (lit) {
                    double light_hit[N];
                    isect_full(blob, &sd, sprim, so, light_vec, light_hit, light_hit_normal);
                    const double dist = v_dist<N>(hit, light_hit);
                    lit = !(dist > NDT_EPS);                // ndt.c:225
                }
            }
            if (lit) {
            double angle = v_angle<N>(nrm, light_vec);      // ndt.c:263
            if (angle > NDT_PI / 2.0) angle = NDT_PI - angle;
            const double light_scale = nd_cos(angle) / ss.ldist2;
            if (!transparent) {
                cr += hit_r * lr_ * light_scale;
                cg += hit_g * lg_ * light_scale;
                cb += hit_b * lb_ * light_scale;
            }
            if (rg.specular) {                              // ndt.c:276-310
                double light_ref[N], rev_look[N];
                v_reflect<N>(light_vec, light_hit_normal, light_ref, 0.5);
                v_unitize<N>(light_ref);
                v_scale<N>(look, -1, rev_look);
                v_unitize<N>(rev_look);
                double rv = v_dot<N>(light_ref, rev_look);
                rv = (0 > rv) ? 0 : rv;                     // MAX(0,rv), image.h:31
                const double rvn = nd_pow(rv, 50.0);
                const double gb = (lg_ > lb_) ? lg_ : lb_;
                const double max_light = (lr_ > gb) ? lr_ : gb;
                cr += hitr_r * lr_ / max_light * rvn;
                cg += hitr_g * lg_ / max_light * rvn;
                cb += hitr_b * lb_ / max_light * rvn;
            }
            }
            }
        }
        if (resolve_here) {
            // The deepest bounce of the frame: its nodes have no child nodes (a child there was cut off, -2: black, ndt.c:336-341),
            // so get_ray_color's blend (ndt.c:402-429; ndt_frame.hip:resolve_node, same operations) needs nothing but this node
            const double hitr[3] = { refl_r, refl_g, refl_b };
            double c[3] = { cr, cg, cb };
            if (ws.child_refl[g] != -1) {
                const double ref = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (rg.specular) c[k] = (1 - hitr[k]) * (c[k]) + (hitr[k]) * ref;       // ndt.c:405-407
                    else c[k] += hitr[k] * ref;                                             // ndt.c:411-413
                }
            }
            if (ws.child_refr[g] != -1) {
                const double ref = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) c[k] += (1.0 - hitr[k]) * ref;                  // ndt.c:426-428
            }
            cr = c[0]; cg = c[1]; cb = c[2];
        }
        ws.clr[0 * ws.cap + g] = cr;
        ws.clr[1 * ws.cap + g] = cg;
        ws.clr[2 * ws.cap + g] = cb;
        ws.count[g] = 1 + n_shadow;

    }
}

__global__ void __launch_bounds__(256, NDT_SHADE_WAVES) k_shade_finish(const double *blob, SceneDesc sd, Workspace ws, RenderGeom rg, int level,
                                                                       int resolve_here)
{
    const LevelRange lr = ws.levels[level];
    const long long base = (long long)blockIdx.x * blockDim.x;
    if (base < lr.count) {
        NDT_SHADE_LOG_BEGIN();
        shade_finish_node(blob, sd, ws, rg, lr, level, base + threadIdx.x, resolve_here != 0);
        NDT_SHADE_LOG_END();
    }
}

// lighting of bounce `level` in the first n_finish workgroups, shading of bounce level+1 in the rest.  (The other way
// round -- the long-lived shading wavefronts first -- was measured: they then all contend for the segment counters at
// once and live 22-32 us instead of 16-19, and the launch takes 20-30 % longer.)
__global__ void __launch_bounds__(256, NDT_SHADE_WAVES) k_shade_pair(const double *blob, SceneDesc sd, Workspace ws, RenderGeom rg, int level,
                                                       unsigned n_finish)
{
    __shared__ EmitShared sh;
    if (blockIdx.x < n_finish) {
        const LevelRange lr = ws.levels[level];
        const long long base = (long long)blockIdx.x * blockDim.x;
        if (base < lr.count) {
            NDT_SHADE_LOG_BEGIN();
            shade_finish_node(blob, sd, ws, rg, lr, level, base + threadIdx.x);
            NDT_SHADE_LOG_END();
        }
    } else {
        const LevelRange lr = ws.levels[level + 1];
        const long long base = (long long)(blockIdx.x - n_finish) * blockDim.x;
        if (base < lr.count) {
            NDT_SHADE_LOG_BEGIN();
            if (threadIdx.x < 64) sh.seg_total[threadIdx.x] = 0;
            if (threadIdx.x == 0) sh.spawn_total = 0;
            __syncthreads();
            shade_emit_node(blob, sd, ws, rg, lr, level + 1, base + threadIdx.x, &sh);
            NDT_SHADE_LOG_END();
        }
    }
}

// ------------------------------------------------------------------ the streaming frame kernel

#include "ndt_stream.hpp"

// ------------------------------------------------------------------ hit points for the trace_kd batch API

__global__ void __launch_bounds__(256) k_hitpoints(const double *blob, SceneDesc sd, const double *o, const double *v,
                                                   long long stride, const int *prim, double *hit, double *nrm,
                                                   long long count)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= count) return;
    double ro[N], rv[N], h[N], n[N];
    v_zero<N>(h);
    v_zero<N>(n);
    const int p = prim[g];
    if (p >= 0) {
        load_soa<N>(o, stride, g, ro);
        load_soa<N>(v, stride, g, rv);
        isect_full(blob, &sd, p, ro, rv, h, n);
    }
    store_soa<N>(hit, stride, g, h);
    store_soa<N>(nrm, stride, g, n);
}

// ------------------------------------------------------------------ launchers

static unsigned grid_for(long long n, int block) { return (unsigned)((n + block - 1) / block); }

static void launch_primary(hipStream_t s, const double *blob, SceneDesc sd, Workspace ws, RenderGeom rg)
{
    hipLaunchKernelGGL(k_primary, dim3(grid_for(rg.n_primary, 256)), dim3(256), 0, s, blob, sd, ws, rg);
}
static unsigned shade_grid(long long upper) { return grid_for(upper, 256); }
static void launch_shade_emit(hipStream_t s, const double *blob, SceneDesc sd, Workspace ws, RenderGeom rg, int level, long long upper)
{
    if (upper <= 0) return;
    hipLaunchKernelGGL(k_shade_emit, dim3(grid_for(upper, NDT_EMIT_BLOCK)), dim3(NDT_EMIT_BLOCK), 0, s, blob, sd, ws, rg, level);
}
static void launch_shade_finish(hipStream_t s, const double *blob, SceneDesc sd, Workspace ws, RenderGeom rg, int level, long long upper,
                                int resolve_here)
{
    if (upper <= 0) return;
    hipLaunchKernelGGL(k_shade_finish, dim3(shade_grid(upper)), dim3(256), 0, s, blob, sd, ws, rg, level, resolve_here);
}
static void launch_shade_pair(hipStream_t s, const double *blob, SceneDesc sd, Workspace ws, RenderGeom rg, int level,
                              long long upper_finish, long long upper_emit)
{
    const unsigned nf = upper_finish > 0 ? shade_grid(upper_finish) : 0, ne = upper_emit > 0 ? shade_grid(upper_emit) : 0;
    if (nf + ne == 0) return;
    hipLaunchKernelGGL(k_shade_pair, dim3(nf + ne), dim3(256), 0, s, blob, sd, ws, rg, level, nf);
}
static void launch_hitpoints(hipStream_t s, const double *blob, SceneDesc sd, const double *o, const double *v,
                             long long stride, const int *prim, double *hit, double *nrm, long long count)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(k_hitpoints, dim3(grid_for(count, 256)), dim3(256), 0, s, blob, sd, o, v, stride, prim, hit, nrm, count);
}

} // namespace

extern "C" const NdtKernelTable *NDT_CAT(ndt_kernel_table_, NDT_DIMS)()
{
    using namespace NDT_CAT(ndt_d, NDT_DIMS);
    static const NdtKernelTable table = { NDT_DIMS, launch_primary, launch_trace, launch_shade_emit, launch_shade_finish,
                                          launch_shade_pair, launch_hitpoints, launch_frame_stream };
    return &table;
}
