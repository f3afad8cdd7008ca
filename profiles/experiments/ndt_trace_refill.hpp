// ndt_trace_refill.hpp -- trace_kd for a queue of rays with LANE REFILL: an experiment that LOST (kept for the record, not
// compiled into the library; it was included by ndt_kernels.hip inside the per-dimension namespace and selected by a
// `refill` field of TraceJob).  Measured on MI355X, round 2 (gpurun_out/r02p): correct (all 98 GPU tests), but the trace
// launches of the benchmark frame took 0.33-0.35 ms against 0.22 ms (refill thresholds 16 / 24 / 32 / 40: the larger the
// better, i.e. the closer to whole batches), hypercube 3-D 0.116-0.125 against 0.077 ms, 8-D 2.0-2.2 against 1.97 ms.  The
// lock-step T / G / I phases want the lanes of a wavefront in step: rays that start together are in the same phase most of the
// time, a fresh ray among half-finished ones is not, and every refill costs a load round trip in the middle of the traversal.
//
// k_trace gives a wavefront 64 rays and the wavefront is busy until the slowest of them is done: on the benchmark scene
// a ray visits 11.6 tree nodes, 6.8 bounding spheres and 2.8 primitives on average but up to 55 / 52 / 40, so a batch lasts
// three to five times its mean ray and the lanes of the finished rays idle (lane utilisation 0.54, VALU issue 39 %).
// Here a lane whose ray is finished gets the next ray of the queue: the wavefront leaves its traversal loop as soon as
// `refill` lanes are finished, stores their answers, hands those lanes new rays (in pieces of the queue's 64-ray batches)
// and goes on -- the unfinished rays keep their state in their registers and in the LDS stack and simply continue.  Per ray
// nothing changes: same list order, same gates, same mask, same dist_limit break, same answer (the known-answer tests
// and every framebuffer test run through this kernel).
//
// The traversal below is trace_kd of ndt_device.hpp with its locals living across rays; see there for the T / G / I
// phases and the reference citations.

template <int MW, bool LDS, bool LSTACK = false>
__global__ void __launch_bounds__(NDT_TRACE_MAX_BLOCK) k_trace_refill(const double *gblob, SceneDesc sd, Workspace ws, TraceJob job)
{
    extern __shared__ __attribute__((aligned(16))) double lds_blob[];
    const double *blob = gblob;
    if (LDS) {
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
    KdStackLds ls{};
    if (LSTACK) {
        double *base = lds_blob + ((sd.trace_words + 1) & ~1);
        const int depth = sd.kd_depth + 1;
        ls.stride = blockDim.x;
        ls.tu = base + threadIdx.x;
        ls.node = (int *)(base + (size_t)depth * blockDim.x) + threadIdx.x;
    }
    VisitMask<MW> mask;
    mask.ext = nullptr;
    mask.ext_stride = 0;
    mask.live0 = mask.live1 = 0ull;
    mask.lazy = false;
    if (MW == 0) {
        const long long lane_slot = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        mask.ext = ws.mask_slab + lane_slot;
        mask.ext_stride = (int)ws.mask_slab_lanes;
    }
    const int lane = __lane_id();
    const unsigned long long below = (1ull << lane) - 1ull;

    // ---- the queue: logical batches of 64 rays, the dense part (closest-hit rays) first, the shadow segments after it;
    // batch L belongs to shard L % SHARDS (k_trace's scheme, without its end-of-launch reserve: there is no last batch
    // that a whole wavefront waits for any more)
    int seg_batches = 0, seg_batches_incl = 0, seg_cnt = 0;     // lane s: inclusive prefix of batches / count of segment s
    if (job.n_seg > 0) {
        seg_cnt = (lane < job.n_seg) ? job.seg_count[lane] : 0;
        const int excl = wave_excl_scan((seg_cnt + 63) >> 6, seg_batches);
        seg_batches_incl = excl + ((seg_cnt + 63) >> 6);
    }
    long long dense_count = job.count, dense_begin = job.begin, seg_stride = job.seg_stride;
    if (job.levels) {
        dense_begin = job.levels[job.dense_level].begin;
        dense_count = job.levels[job.dense_level].count;
        seg_stride = job.levels[job.seg_level].seg_stride;
    }
    const long long dense_batches = (dense_count + 63) >> 6;
    const long long n_batches = dense_batches + seg_batches;
    int cur = blockIdx.x % NDT_QUEUE_SHARDS;
    bool supply_left = n_batches > 0;
    // the piece of the queue this wavefront is handing out to its lanes
    long long cur_g0 = 0;           // slot of the piece's first ray
    int cur_cnt = 0, cur_used = 0;  // rays in the piece / handed out
    bool cur_seg = false;

    // ---- per-lane ray and traversal state (the locals of trace_kd, alive across rays)
    double o[N], v[N], v_inv[N];
#pragma unroll
    for (int c = 0; c < N; ++c) { o[c] = 0.0; v[c] = 0.0; v_inv[c] = 0.0; }
    double dist_limit = -1.0;
    long long slot = -1;            // where this lane's answer goes; < 0: the lane has no ray
    bool slot_seg = false;
    bool fin = true;                // the lane's ray is finished (or it has none)
    double t_inf = NDT_DBL_MAX, lt = NDT_DBL_MAX;
    bool ret_inf = false, lret = false;
    int inf_obj = -1, inf_prim = -1, l_obj = -1, l_prim = -1;
    int st_node[LSTACK ? 1 : NDT_KD_STACK];
    double st_a[LSTACK ? 1 : NDT_KD_STACK], st_tu[LSTACK ? 1 : NDT_KD_STACK];
    int sp = 0, node = 0;
    double ntl = 0, ntu = 0;
    bool have_node = false, trav_done = false, root_pending = false;
    bool have_list = false, list_is_inf = false;
    int sec = 0, pos = 0, end = 0;
    double min_dist = -1;
    int best_obj = -1, best_prim = -1;
    bool in_sub = false;
    int sub_i = 0, sub_end = 0, sub_owner = -1, sub_prim = -1;
    long long sub_live = -1;
    double sub_min = -1;
    const int refill = job.refill > 0 ? job.refill : 24;

    while (true) {
        // ---- (1) the answers of the rays that are finished: kd_tree_intersect's return, kd-tree.c:612
        if (fin && slot >= 0) {
            int out_obj = inf_obj, out_prim = inf_prim;
            if (lret) {
                if (!ret_inf || (lt > NDT_EPS && lt + NDT_EPS < t_inf)) {
                    out_obj = l_obj;
                    out_prim = l_prim;
                }
            }
            const TracePart &part = slot_seg ? job.seg : job.dense;
            part.out_obj[slot] = out_obj;
            part.out_prim[slot] = out_prim;
            slot = -1;
        }
        // ---- (2) lanes without a ray take the next rays of the queue
        unsigned long long idle = __ballot(slot < 0);
        while (idle != 0ull && (cur_used < cur_cnt || supply_left)) {
            if (cur_used == cur_cnt) {
                // the next batch: home shard first, then the shards that still hold something
                long long b = -1;
                while (supply_left && b < 0) {
                    const long long n_here = (n_batches - cur + NDT_QUEUE_SHARDS - 1) / NDT_QUEUE_SHARDS;
                    int k = 0;
                    if (lane == 0) k = atomicAdd(job.queue + cur * NDT_QUEUE_STRIDE, 1);
                    k = __shfl(k, 0, 64);
                    if (k < n_here) {
                        b = (long long)k * NDT_QUEUE_SHARDS + cur;
                    } else {
                        int head = 0x7fffffff;
                        if (lane < NDT_QUEUE_SHARDS)
                            head = __hip_atomic_load(job.queue + lane * NDT_QUEUE_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const long long mine = (n_batches - lane + NDT_QUEUE_SHARDS - 1) / NDT_QUEUE_SHARDS;
                        const unsigned long long live = __ballot(lane < NDT_QUEUE_SHARDS && (long long)head < mine);
                        if (live == 0ull) {
                            supply_left = false;
                        } else {
                            const unsigned long long above = (cur + 1 < 64) ? (live >> (cur + 1)) << (cur + 1) : 0ull;
                            cur = __ffsll((long long)(above ? above : live)) - 1;
                        }
                    }
                }
                if (b < 0) break;
                cur_seg = b >= dense_batches;
                cur_used = 0;
                if (cur_seg) {
                    const int sb = (int)(b - dense_batches);
                    const int s = __popcll(__ballot(lane < job.n_seg && seg_batches_incl <= sb));
                    const int first = (s > 0) ? __shfl(seg_batches_incl, s - 1, 64) : 0;
                    const int cnt = __shfl(seg_cnt, s, 64);
                    const int off = (sb - first) * 64;
                    cur_g0 = (long long)s * seg_stride + off;
                    cur_cnt = cnt - off < 64 ? cnt - off : 64;
                } else {
                    cur_g0 = dense_begin + b * 64;
                    const long long left = dense_count - b * 64;
                    cur_cnt = left < 64 ? (int)left : 64;
                }
            }
            const int n_idle = __popcll(idle);
            const int take = n_idle < cur_cnt - cur_used ? n_idle : cur_cnt - cur_used;
            const int rank = __popcll(idle & below);
            if (slot < 0 && ((idle >> lane) & 1ull) && rank < take) {
                const long long g = cur_g0 + cur_used + rank;
                const TracePart &part = cur_seg ? job.seg : job.dense;
                const bool valid = cur_seg || !job.dense.valid || job.dense.valid[g] > 0;
                if (valid) {
                    load_soa<N>(part.o, part.stride, g, o);
                    load_soa<N>(part.v, part.stride, g, v);
                    dist_limit = part.lim ? part.lim[g] : -1.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const double v_i = v[i];
                        double r;
                        if (v_i < NDT_EPS2 && v_i >= 0.0) r = NDT_INV_EPS2;
                        else if (v_i > -NDT_EPS2 && v_i <= 0.0) r = -NDT_INV_EPS2;
                        else r = 1.0 / v_i;
                        v_inv[i] = r;
                    }
                    slot = g;
                    slot_seg = cur_seg;
                    fin = false;
                    t_inf = NDT_DBL_MAX; ret_inf = false; inf_obj = -1; inf_prim = -1;
                    lt = NDT_DBL_MAX; lret = false; l_obj = -1; l_prim = -1;
                    sp = 0; node = 0; ntl = 0; ntu = 0;
                    have_node = false; trav_done = false; root_pending = true;
                    have_list = false; list_is_inf = false; sec = 0; pos = 0; end = 0;
                    in_sub = false; sub_i = 0; sub_end = 0; sub_owner = -1; sub_prim = -1; sub_live = -1; sub_min = -1;
                    if (sd.n_inf > 0) {
                        have_list = true;
                        list_is_inf = true;
                        sec = sd.off_inf;
                        pos = 0;
                        end = sd.n_inf;
                    }
                }
            }
            cur_used += take;
            // (a lane whose ray was a padding slot is idle again; the lanes this round did not reach too)
            idle = __ballot(slot < 0);
        }
        const unsigned long long busy = __ballot(slot >= 0);
        if (busy == 0ull) break;            // nothing in flight and nothing left to fetch
        // ---- (3) traverse until `refill` rays are finished (all of them, when there is nothing to refill with)
        const int n_busy = __popcll(busy);
        const bool can_refill = cur_used < cur_cnt || supply_left;
        const int want = (can_refill && refill < n_busy) ? refill : n_busy;
        while (__popcll(__ballot(fin && slot >= 0)) < want) {
            if (!fin) {
                // ------------------------------------------------------------ phase T
                if (!have_list && !trav_done && root_pending) {
                    root_pending = false;
                    double tl = -NDT_DBL_MAX, tu = NDT_DBL_MAX;
                    bool box = true;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        if (box) {
                            double v_i = v[i], o_i = o[i];
                            if (!(fabs(v_i) < NDT_EPS2)) {
                                double tl_i = (blob[sd.off_bb + i] - o_i) / v_i;
                                double tu_i = (blob[sd.off_bb + N + i] - o_i) / v_i;
                                if (tl_i > tu_i) {
                                    double tmp = tl_i;
                                    tl_i = tu_i;
                                    tu_i = tmp;
                                }
                                if (tl_i > tl) tl = tl_i;
                                if (tu_i < tu) tu = tu_i;
                                if (tu < -NDT_EPS) box = false;
                            }
                        }
                    }
                    if (box) {
                        tl -= NDT_EPS;
                        tu += NDT_EPS;
                        box = (tu >= -NDT_EPS) && (tl <= tu);
                    }
                    if (!box || sd.n_kd_nodes <= 0) {
                        trav_done = true;
                    } else {
                        mask.clear(sd.mask_words);
                        node = 0;
                        ntl = tl;
                        ntu = tu;
                        have_node = true;
                    }
                }
                while (!have_list && !trav_done) {
                    bool visit = have_node;
                    if (!have_node) {
                        if (sp == 0) {
                            trav_done = true;
                        } else {
                            --sp;
                            int nf;
                            double a;
                            if (LSTACK) {
                                const int parent = ls.node[sp * ls.stride];
                                ntu = ls.tu[sp * ls.stride];
                                const ndt_v2d prec = blob_pair(blob, sd.off_kd + 2 * parent);
                                const long long pw0 = __double_as_longlong(prec.x);
                                const int pdim = (int)(pw0 & 0xffffffffll);
                                const double pv_inv = v_pick<N>(v_inv, pdim);
                                a = (prec.y - v_pick<N>(o, pdim)) * pv_inv;
                                nf = (pv_inv < NDT_EPS2) ? parent + 1 : (int)(pw0 >> 32);
                            } else {
                                nf = st_node[sp];
                                a = st_a[sp];
                                ntu = st_tu[sp];
                            }
                            node = nf & ~NDT_STACK_FLAG;
                            ntl = (nf & NDT_STACK_FLAG) ? a : a - NDT_EPS;
                            visit = lt > a;
                        }
                    }
                    have_node = false;
                    if (visit && !(ntu < 0.0)) {
                        const ndt_v2d rec = blob_pair(blob, sd.off_kd + 2 * node);
                        const long long w0 = __double_as_longlong(rec.x);
                        const int dim = (int)(w0 & 0xffffffffll);
                        if (dim < 0) {
                            const long long w1 = __double_as_longlong(rec.y);
                            const int num = (int)(w1 >> 32);
                            if (num > 0) {
                                have_list = true;
                                list_is_inf = false;
                                sec = sd.off_leaf;
                                pos = (int)(w1 & 0xffffffffll);
                                end = pos + num;
                            }
                        } else {
                            const double boundary = rec.y;
                            const double v_inv_i = v_pick<N>(v_inv, dim);
                            const double o_i = v_pick<N>(o, dim);
                            const bool swap = v_inv_i < NDT_EPS2;
                            const int left = node + 1, right = (int)(w0 >> 32);
                            const int near = swap ? right : left, far = swap ? left : right;
                            if (-NDT_INV_EPS2 <= v_inv_i && v_inv_i <= NDT_INV_EPS2) {
                                const double tp = (boundary - o_i) * v_inv_i;
                                const bool alive = lt > ntl;
                                if (ntu < tp - NDT_EPS && alive) {
                                    node = near; have_node = true;
                                } else if (ntl > tp + NDT_EPS && alive) {
                                    node = far; have_node = true;
                                } else {
                                    if (lt > tp) {
                                        if (LSTACK) { ls.node[sp * ls.stride] = node; ls.tu[sp * ls.stride] = ntu; }
                                        else { st_node[sp] = far; st_a[sp] = tp; st_tu[sp] = ntu; }
                                        ++sp;
                                    }
                                    if (alive) { node = near; ntu = tp + NDT_EPS; have_node = true; }
                                }
                            } else {
                                if (o_i > boundary - NDT_EPS) {
                                    if (!LSTACK) { st_node[sp] = far | NDT_STACK_FLAG; st_a[sp] = ntl; st_tu[sp] = ntu; ++sp; }
                                }
                                if (o_i < boundary + NDT_EPS && lt > ntl) { node = near; have_node = true; }
                            }
                        }
                    }
                }
                if (!have_list) {
                    fin = true;             // no list left: the ray is finished
                } else {
                    // -------------------------------------------------------- phases G + I over the list
                    min_dist = -1;
                    best_obj = -1;
                    best_prim = -1;
                    bool list_open = true;
                    while (list_open) {
                        int prim = -1;
                        bool scanning = true;
                        while (scanning) {
                            if (in_sub && sub_i == sub_end) {
                                in_sub = false;
                                if (sub_min >= 0) {
                                    const double dist = sub_min;
                                    if (dist > NDT_EPS && (dist + NDT_EPS < min_dist || min_dist < 0)) {
                                        min_dist = dist;
                                        best_obj = sub_owner;
                                        best_prim = sub_prim;
                                    }
                                    if (dist_limit == 0.0 || dist < dist_limit) pos = end;
                                }
                            } else if (!in_sub && pos == end) {
                                scanning = false;
                            } else {
                                int id, flags;
                                blob_ref(blob, in_sub ? sd.off_child + sub_i : sec + pos, id, flags);
                                if (in_sub) {
                                    const long long rest = sub_live >> 1;
                                    if (rest == 0) {
                                        sub_i = sub_end;
                                    } else {
                                        const int skip = __ffsll(rest) - 1;
                                        sub_i += 1 + skip;
                                        sub_live = rest >> skip;
                                    }
                                } else {
                                    pos += 1;
                                }
                                bool fresh = true;
                                if (!in_sub && !list_is_inf) fresh = !mask.test_and_set(id);
                                if (fresh) {
                                    const double gate_min = in_sub ? sub_min : min_dist;
                                    if (!(flags & NDT_F_GATE) || bsphere_gate<N>(blob, sd, id, o, v, gate_min)) {
                                        if ((flags & NDT_F_TYPE_MASK) == T_HCUBE) {
                                            const int first = blob_int(blob, sd.off_hdr + 2 * id + 1, 0);
                                            const int nf = blob_int(blob, sd.off_hdr + 2 * id + 1, 1);
                                            long long live = -1;
                                            if (flags & NDT_F_BOX)
                                                live = hull_faces<N>(blob, sd.off_params + blob_int(blob, sd.off_hdr + 2 * id, 1),
                                                                     (flags & NDT_F_FACEBOX) != 0, nf, o, v);
                                            if (live != 0) {
                                                const int skip = __ffsll(live) - 1;
                                                in_sub = true;
                                                sub_owner = id;
                                                sub_i = first + skip;
                                                sub_end = first + nf;
                                                sub_live = live >> skip;
                                                sub_min = -1;
                                                sub_prim = -1;
                                            }
                                        } else {
                                            prim = id;
                                            scanning = false;
                                        }
                                    }
                                }
                            }
                        }
                        if (prim < 0) {
                            list_open = false;
                        } else {
                            double res[N], nrm[N];
                            const bool ok = isect<N, false>(blob, sd, prim, o, v, res, nrm);
                            if (ok) {
                                const double dist = v_dist<N>(o, res);
                                if (in_sub) {
                                    if (dist > NDT_EPS && (dist + NDT_EPS < sub_min || sub_min < 0)) {
                                        sub_min = dist;
                                        sub_prim = prim;
                                    }
                                } else {
                                    if (dist > NDT_EPS && (dist + NDT_EPS < min_dist || min_dist < 0)) {
                                        min_dist = dist;
                                        best_obj = prim;
                                        best_prim = prim;
                                    }
                                    if (dist_limit == 0.0 || dist < dist_limit) pos = end;
                                }
                            }
                        }
                    }
                    // ---- list finished: what trace() returns to its caller
                    have_list = false;
                    if (list_is_inf) {
                        ret_inf = min_dist >= 0;
                        if (min_dist > NDT_EPS) t_inf = min_dist;
                        inf_obj = best_obj;
                        inf_prim = best_prim;
                    } else if (min_dist >= 0) {
                        lret = true;
                        if (min_dist < lt) {
                            lt = min_dist;
                            l_obj = best_obj;
                            l_prim = best_prim;
                        }
                    }
                }
            }
        }
    }
}
