// ndt_kernels.hpp -- launch interface between the host side (ndt_frame.hip, ndt_api.hip) and the
// per-dimension kernel translation units (ndt_kernels.hip compiled once per N).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ndt_device.hpp"

// One bounce of the ray tree.  The table lives in device memory (Workspace::levels): bounce b+1
// is written by k_level_step after shade_emit(b) has spawned its nodes, and every kernel of a
// bounce reads its range from there, so the host enqueues a whole frame without reading anything
// back in between.
struct LevelRange {
    long long begin, count;     // nodes of the bounce
    long long seg_stride;       // capacity of one light's shadow segment for this bounce (>= count, multiple of 64)
    long long n_shadow;         // shadow rays the bounce emitted (statistics)
};
#define NDT_MAX_LEVELS 1024

// Device workspace of one render call.  Ray-tree nodes of all bounces live in one pool
// (structure of arrays, component-major: x[c*cap + g]) so that lane g and lane g+1 touch
// adjacent doubles -- one coalesced 512-byte request per component per wavefront.
struct Workspace {
    // node pool
    long long cap;              // nodes the pool holds
    double *ray_o, *ray_v;      // [N][cap]   origin / unit direction
    double *frac;               // [cap]      pixel_frac (ndt.c:330)
    int *depth_left;            // [cap]      max_depth of this node; 0 = padding slot
    int *hit_obj, *hit_prim;    // [cap]      trace_kd result: material owner / primitive; <0 = miss
    double *hit_p, *hit_n;      // [N][cap]   hit point / normal: the trace_rays API's answers only (the shade kernels recompute them)
    double *clr;                // [3][cap]   local colour, then resolved colour
    int *child_refl, *child_refr; // [cap]    -1 none, -2 cut-off (black), >=0 node
    int *sh_idx;                // [n_seg][cap] index of this node's shadow ray inside its light's segment
    unsigned long long *sh_mask;// [cap]      lights that fired a shadow ray
    int *count;                 // [cap]      trace_kd calls in this node's subtree
    double *depth;              // [cap]      primaries of a depth-map render: 1/distance of the hit, 0 on a miss
    unsigned long long *rng_key;// [cap]      stochastic renders: the node's random stream (ndt_device.hpp)
    // shadow queue of the current bounce
    long long sh_cap;
    double *so, *sv;            // [N][sh_cap]
    double *slim;               // [sh_cap]   dist_limit
    int *sobj, *sprim;          // [sh_cap]
    // counters: [0] node tail, [2] overflow flags (1 nodes, 2 shadows),
    // [NDT_CNT_QUEUE ..) work-queue heads, [NDT_CNT_SEG ..) shadow rays per light segment
    int *counters;
    unsigned long long *ref_rays;   // [64 x 8] partial sums (one 64-byte line each) of the rays the reference would have traced
    unsigned long long *mask_slab;  // visit masks for scenes too big for registers
    long long mask_slab_lanes;
    unsigned long long *dbg;        // [64] diagnostic accumulators (NDT_PHASE_TIMING builds only)
    unsigned int *exit_log;         // [NDT_EXIT_LOG_LAUNCHES][NDT_EXIT_LOG_WORDS], NDT_HIP_EXIT_PROBE
    unsigned int *shade_log;        // NDT_HIP_SHADE_PROBE: {start, end} per wavefront of ONE shade launch (set for that launch only)
    unsigned long long *coop_ring;  // [NDT_COOP_RING_ENTRIES] straggler queue of the trace launches (TraceJob::coop_ring)
    LevelRange *levels;             // [NDT_MAX_LEVELS + 1] bounce table
    // the same table in host-visible (mapped, coherent) memory + one tag per entry: k_level_step
    // posts bounce b+1 here, the host polls the tag instead of synchronising the stream
    LevelRange *mail;
    unsigned long long *mail_tag;
};

// What the primary rays of one pass are.  Grid mode: the pixels of `rows` image rows of a
// width-wide image, in 8x8 tiles (one wavefront = one tile); list mode (`samples` set): arbitrary
// image positions, one per slot (the extra samples of the recursive anti-aliasing pass).
struct RenderGeom {
    int width, height;          // grid: pixels per row / (unused); also the output row stride
    int row_begin, row_step;    // shard: grid row r is image row row_begin + r*row_step ...
    int row_pair;               // ... or, when set, row_begin + (r/2)*row_step + r%2 (corner rows of the AA pass)
    int rows;                   // rows of this shard
    int tiles_x, tiles_y;       // 8x8 tiles over (width, rows)
    int n_primary;              // grid: tiles_x*tiles_y*64; list: n_samples rounded up to 64 (padding slots included)
    int max_depth;
    int specular;
    int img_w, img_h;           // x = i/img_w - 0.5, y = -(j/img_h - 0.5)  (ndt.c:632-633)
    int aspect_w, aspect_h;     // cam.dirX *= aspect_w/aspect_h            (ndt.c:926)
    const double *samples;      // list mode: (i, j) per sample, in pixels of the img_w x img_h image
    int n_samples;
    int lens;                   // list mode: records are (i, j, lx, ly): the eye is moved by lx*localX + ly*localY (ndt.c:538-541)
    int raw_samples;            // list mode: one colour per sample as traced (-n > 1), no replay of the samples=1 loop
    int pixel_halves;           // list mode: the samples are jittered pixels -- the PIXEL decides the half / the eye of a split image
    const unsigned long long *sample_keys;  // list mode, stochastic renders: the random stream of every sample
    int stereo;                 // ndt_stereo_mode: 1 side by side, 2 over/under split the image between the eyes (ndt.c:590-612)
    int eye;                    // 0 left, 1 centre, 2 right: the eye when stereo does not split the image (anaglyph renders twice)
    int want_depth;             // record 1/distance of the primary hits (depth maps, ndt.c:362-373)
};

// One k_trace launch = up to two parts, served from one work queue:
//   seg   : a segmented queue (shadow rays of bounce b, one segment per light so that a
//           wavefront's rays share their origin): segment s holds seg_count[s] rays at
//           [s*seg_stride, ...), each with its own dist_limit;
//   dense : one range [begin, begin+count) of closest-hit queries (the nodes of bounce b+1, or
//           the batch of the trace_rays API).  count may live on the device (tail_ptr).
// Shadow rays of one bounce and the closest-hit rays of the next are independent, so the host
// puts them in the same launch: every launch has a latency floor of one incoherent batch
// (~0.3 ms on the benchmark scene), and this halves the number of floors per frame.
struct TracePart {
    const double *o, *v;        // [N][stride]
    long long stride;
    const double *lim;          // per-ray dist_limit, or nullptr => -1 (closest hit)
    const int *valid;           // depth_left, or nullptr => all valid
    int *out_obj, *out_prim;
};
struct TraceJob {
    TracePart seg;
    const int *seg_count;
    long long seg_stride;       // (taken from levels[seg_level] when `levels` is set)
    int n_seg;                  // 0 => no segmented part
    TracePart dense;
    long long begin, count;     // dense range (taken from levels[dense_level] when `levels` is set; count then only sizes the grid)
    const LevelRange *levels;   // device-side bounce table, or nullptr
    int seg_level, dense_level;
    int *queue;                 // device-side work-queue heads for this launch (zeroed by the host)
    int batch;                  // rays per wavefront batch: 64, 32, 16 or 8 (set by the launcher)
    int skip_trace;             // diagnostic build only: pop, load and store but do not traverse
    int tail_solo;              // the last this-many batches of every queue shard go to one wavefront per SIMD only (k_trace)
    unsigned int *exit_log;     // NDT_HIP_EXIT_PROBE: {start, start of the last batch, exit} per wavefront (100 MHz clock, low words)
    // Cooperative stragglers (item-set tier; ndt_device.hpp:coop_trace): the rays a wavefront gives up on -- a batch over its
    // budget with few rays left -- are queued in `coop_ring` and traced, one ray per wavefront, by the wavefronts that have run
    // out of batches.  Entries: tag << 32 | payload (payload: ray slot, bit 31 = shadow part); a slot is valid for THIS launch
    // when its tag is coop_tag, so the ring is never cleared.  The counters live behind the launch's queue heads (NDT_COOP_*).
    unsigned long long *coop_ring;      // nullptr: off
    unsigned int coop_tag;
    int coop_limit;             // nothing is given up once the ring's tail has passed this (the ring has room for what is in flight beyond)
    int coop_budget;            // 100 MHz ticks a batch may take before its last rays are given up
    int coop_max_live;          // ... when at most this many are left
    int coop_tail_only;         // ... and (when set) the batch's queue shard has run dry: only in the tail of the launch
    int coop_waves;             // wavefronts of a workgroup that stay as consumers (4: one per SIMD)
    unsigned int *coop_log;     // NDT_HIP_EXIT_PROBE: [0] rays given up, [1] rays traced cooperatively, [2] ticks spent in coop_trace
    // the dense part = the primaries of the pass `rg` describes, made by the wavefront that traces them (primary_node)
    int make_primaries;
    RenderGeom rg;
    // The launch "shadow rays of bounce b + closest hits of bounce b + 1" also PUBLISHES bounce b + 1 (publish_level = b; -1: no):
    // shade_emit(b) has just spawned it, so its range is what the node tail says -- every wavefront works it out for itself
    // from the counters (final since the kernel boundary), and the first wavefront of the launch writes it to the bounce
    // table, clears the other parity's segment counters and posts it to the host's mailbox (the host enqueues bounce b + 1
    // only once it knows there is one).  Round 3 had a one-wavefront launch for this after every shading launch (12 us of
    // stream time each); counting workgroups out of the shading launches instead (a "last one does it") cost every workgroup
    // a returning atomic before it could leave, and more than it saved (profiles/experiments/README.md, round 4).
    int publish_level;
    unsigned long long publish_tag;
    // The segmented part holds the shadow rays of the scene's lights, segment s = the s-th light that is not ambient, and the
    // rays of a point or spot light all start AT the light (ndt.c:211): their origin is the light's position in the scene
    // blob, not 8 N bytes a ray written by shade_emit and read back here (round 4; area lights jitter theirs: stored as before).
    int seg_light_origins;
};
#define NDT_EXIT_LOG_WORDS 65536    /* per launch: 8 words {start, last batch, exit, HW_ID, out of batches, -, -, -} x 8192 wavefronts */
#define NDT_EXIT_LOG_LAUNCHES 6
#define NDT_SHADE_LOG_WAVES 131072  /* wavefronts the shade probe has slots for */


// ------------------------------------------------------------------ the streaming frame kernel (ndt_stream.hpp)
//
// One persistent launch renders the whole ray tree of a frame: wavefronts pull typed work items -- a batch of 64
// nodes to trace and shade, a batch of 64 shadow rays to trace, a batch of 64 nodes whose shadow answers are all in
// to light -- from device-side queues, and produce the items that depend on theirs.  Nothing waits for a bounce to
// finish: a wavefront that runs out of one kind of work takes another.
// One hot word of the control block.  The words are 4352 bytes apart: memory is interleaved over the L2 / HBM channels
// in blocks of a few KB, so a control block of consecutive cache lines lives in ONE channel and every counter, head and
// tail of the frame is served by it, one after the other (~100 operations per microsecond: with 600 k of them a frame
// the whole kernel ran at the speed of that channel -- 6 ms whatever the number of wavefronts).
struct StreamWord {
    int v;
    int _pad[1087];
};
struct StreamCtl {
    StreamWord node_tail;           // next free node slot; starts at n_primary (a multiple of 64)
    // What is still to do, in node slots, per shard (batch nb belongs to shard nb % 8): +2 for every slot reserved (primaries:
    // set at the frame's start), -1 per slot when its batch has been traced and shaded, -1 when its lighting is done (or it
    // needs none).  All eight zero, with the node tail unchanged around the reading: the frame is complete.
    StreamWord outstanding[8];
    // Ready batches of secondary nodes: one ring per shard (workgroup w pushes to and pops from shard w % 8, and looks at
    // the other shards only when it has nothing to do).  v = tail (low half) and head (high half) of the ring in ONE word:
    // "is there something" is one load, a push one 64-bit add of 1, a pop one add of 1 << 32.
    struct { unsigned long long v; int _pad[1086]; } sec[8];
    // Ready batches of shadow rays / of nodes to light: one ring each.  A wavefront ALWAYS holds a ticket (a slot number)
    // of each and polls that slot -- an address of its own -- so nobody ever reads head or tail.
    StreamWord sh_head, sh_tail;
    StreamWord fin_head, fin_tail;
    StreamWord prim_head[8];        // sharded heads of the primaries' batches (batch b belongs to shard b % 8)
    StreamWord seg_tail[64];        // next free slot of every light's shadow segment
    StreamWord abort;               // != 0: every wavefront leaves (1 pool overflow, 2 timeout)
    StreamWord overflow;            // 1 node pool, 2 shadow segment
    StreamWord n_children, n_shadow, max_level, timeout_where;     // statistics (every wavefront adds its own once, when it leaves)
};
#define NDT_PRIM_SHARDS 8
struct StreamArgs {
    StreamCtl *ctl;
    int *node_fill;         // [cap / 64]      slots of a node batch that have been written
    int *sh_pending;        // [cap / 64]      shadow rays of a node batch that are not answered yet
    int *sh_fill;           // [n_seg][seg_cap / 64]
    int *sec_ring, *sh_ring, *fin_ring;     // entries: id + 1, 0 = not written yet; sec_ring: [8 shards][cap / 64]
    int *parent, *pend;     // [cap]  the node a node reports to; what a node still waits for (own lighting + children)
    int *sowner;            // [n_seg * seg_cap]  the node a shadow ray belongs to, -1 = padding slot
    int n_seg;
    int seg_cap;            // slots per light segment (multiple of 64)
    // The roots of the forest the launch renders: node slots [root_begin, root_begin + n_primary), both multiples of 64.  A
    // whole frame: the primaries (root_begin = 0).  The deep bounces of a frame whose first bounces went through the
    // per-bounce kernels: the nodes of the first bounce the frame kernel takes over (render_pass, hybrid pipeline).
    int root_begin;
    int n_primary;
    int roots_are_primaries;    // 1: the roots are the frame's primary rays (depth maps record their hits)
    // of the root batches' slots only [valid_begin, valid_end) are roots (the first and the last root batch may hold slots
    // of the bounce before / slots nobody has written: the root range is the 64-aligned cover of a bounce's node range)
    int valid_begin, valid_end;
    int node_batches;       // cap / 64: entries per shard of sec_ring
    unsigned int *wave_log; // NDT_HIP_STREAM_PROBE: 16 words per wavefront (what it did and when), else nullptr
    // fused (a whole frame, roots = primaries): the kernel makes its primaries itself (what k_primary does: a root batch's rays
    // are computed by the wavefront that traces them) and writes a pixel the moment its ray tree is resolved (what
    // k_finish_pixels does): two launches and their fill / drain less around the frame kernel
    int fused;
    double *rgba, *depth_out;
};
#define NDT_STREAM_LOG_WAVES 16384

// One table per compiled dimension.
struct NdtKernelTable {
    int dims;
    void (*primary)(hipStream_t, const double *blob, SceneDesc, Workspace, RenderGeom);
    // tier: 0 = scene staged in LDS, visit mask in registers; 1 = scene in global memory, mask in the slab
    // ev_start / ev_stop: optional HIP events that receive the kernel's own start / stop timestamps
    void (*trace)(hipStream_t, const double *blob, SceneDesc, Workspace, TraceJob, int tier, int mask_words, hipEvent_t ev_start,
                  hipEvent_t ev_stop);
    // `upper` bounds the bounce's node count (sizes the grid); the range itself is ws.levels[level]
    void (*shade_emit)(hipStream_t, const double *blob, SceneDesc, Workspace, RenderGeom, int level, long long upper);
    // resolve_here: `level` is the deepest bounce of the frame, its nodes are blended on the spot (no k_resolve for it)
    void (*shade_finish)(hipStream_t, const double *blob, SceneDesc, Workspace, RenderGeom, int level, long long upper, int resolve_here);
    // shade_finish(level) and shade_emit(level + 1) in one launch: they touch different nodes, and both are
    // short latency-bound kernels for the small deep bounces
    void (*shade_pair)(hipStream_t, const double *blob, SceneDesc, Workspace, RenderGeom, int level, long long upper_finish,
                       long long upper_emit);
    void (*hitpoints)(hipStream_t, const double *blob, SceneDesc, const double *o, const double *v, long long stride,
                      const int *prim, double *hit, double *nrm, long long count);
    // the whole ray tree of a frame in one persistent launch (ndt_stream.hpp); primaries already in the pool
    void (*frame_stream)(hipStream_t, const double *blob, SceneDesc, Workspace, RenderGeom, StreamArgs, int tier, int mask_words,
                         hipEvent_t ev_start, hipEvent_t ev_stop);
};

extern "C" const NdtKernelTable *ndt_kernel_table_3();
extern "C" const NdtKernelTable *ndt_kernel_table_4();
extern "C" const NdtKernelTable *ndt_kernel_table_5();
extern "C" const NdtKernelTable *ndt_kernel_table_6();
extern "C" const NdtKernelTable *ndt_kernel_table_7();
extern "C" const NdtKernelTable *ndt_kernel_table_8();
extern "C" const NdtKernelTable *ndt_kernel_table_9();
extern "C" const NdtKernelTable *ndt_kernel_table_10();
extern "C" const NdtKernelTable *ndt_kernel_table_11();
extern "C" const NdtKernelTable *ndt_kernel_table_12();

#define NDT_TRACE_BLOCK 256
// The global-memory tier (visit masks in the slab, scene in global memory: the 6-D .. 8-D hypercubes) launches
// NDT_TRACE_BLOCK-lane workgroups; what its trace kernel is compiled for decides its registers.  For 768 lanes (168
// registers, three wavefronts per SIMD) the 6-D / 7-D / 8-D kernels spilled 77 / 179 / 267 registers -- a vector is 2N of
// them -- into the very memory system the tier waits for; for 512 lanes (256 registers, two per SIMD) they spill none:
// 1080p frames 2.19 -> 2.10, 3.92 -> 3.69, 6.78 -> 6.66 ms.
#ifndef NDT_TRACE_T1_MAX_BLOCK
#define NDT_TRACE_T1_MAX_BLOCK 512
#endif
#ifndef NDT_TRACE_MAX_BLOCK
#define NDT_TRACE_MAX_BLOCK 768
#endif
#define NDT_QUEUE_SLOTS 512                 /* trace launches per render call that get a work queue */
#ifndef NDT_QUEUE_SHARDS
#define NDT_QUEUE_SHARDS 64                 /* queue heads per launch (one lane reads one head); with 8 -- one per XCD -- 384 wavefronts shared a head */
#endif
#define NDT_QUEUE_STRIDE 16                 /* ints between heads: one 64-byte line each */
/* behind the heads, one 64-byte line each: the straggler ring's tail and head, the wavefronts that have left the batch loop
   per group of workgroups (blockIdx % 8), and the groups that are complete */
#define NDT_COOP_TAIL (NDT_QUEUE_SHARDS * NDT_QUEUE_STRIDE)
#define NDT_COOP_HEAD (NDT_COOP_TAIL + NDT_QUEUE_STRIDE)
#define NDT_COOP_LEFT (NDT_COOP_HEAD + NDT_QUEUE_STRIDE)
#define NDT_COOP_GROUPS 8
#define NDT_COOP_GROUPS_DONE (NDT_COOP_LEFT + NDT_COOP_GROUPS * NDT_QUEUE_STRIDE)
#define NDT_QUEUE_INTS (NDT_COOP_GROUPS_DONE + NDT_QUEUE_STRIDE)
#define NDT_COOP_RING_LIMIT (1 << 20)       /* entries that may be given up per launch ... */
#define NDT_COOP_RING_ENTRIES (NDT_COOP_RING_LIMIT + 8192 * 64 + 8192)   /* ... + what the wavefronts in flight can add beyond + one closing entry per consumer */
#define NDT_COOP_CLOSE 0xfffffffeu          /* payload of the closing entries */
#define NDT_CNT_QUEUE 16
#define NDT_CNT_SEG (NDT_CNT_QUEUE + NDT_QUEUE_SLOTS * NDT_QUEUE_INTS)
#define NDT_CNT_TOTAL (NDT_CNT_SEG + 128)         /* shadow-segment counters, double-buffered by bounce parity */
#define NDT_SEG_COUNTERS(ws, level) ((ws).counters + NDT_CNT_SEG + 64 * ((level) & 1))
#define NDT_SHADE_MAX_BLOCKS 8192            /* shade kernels walk longer bounces with a grid-stride loop */
#define NDT_TRACE_LDS_LIMIT (64 * 1024)     /* bytes of scene staged per workgroup: two workgroups per CU */
#define NDT_MASK_REG_WORDS 4                /* 64-bit words of visit mask kept in registers (256 items) */

__device__ __forceinline__ int ndt_ld_cnt(const int *p) { return __hip_atomic_load(const_cast<int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
