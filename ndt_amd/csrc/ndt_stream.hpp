// ndt_stream.hpp -- the streaming frame kernel: the whole ray tree of a frame in ONE persistent launch.
// Included by ndt_kernels.hip inside the per-dimension namespace (it uses N, the wavefront helpers, light_setup,
// isect and trace_kd of that translation unit).
//
// The bounce-synchronous pipeline (k_trace / k_shade_* per bounce) pays, per bounce, the tail of a trace launch (the
// chip idles while the last batches finish), the fixed latency of the shade launches and the launch boundaries; five
// bounces of that were two thirds of a millisecond of a 1.6 ms frame.  Here nothing waits for a bounce: wavefronts are
// persistent and pull typed work items from device-side queues,
//
//   NODE batch    64 consecutive nodes of the pool: trace_kd (closest hit), then in the SAME wavefront everything
//                 get_ray_color / apply_lights do before their shadow queries (ndt.c:329-430, 71-259): hit point,
//                 background, reflection / refraction children, one shadow ray per light that passes the same-side
//                 and cone tests.  Children are appended to the node pool, shadow rays to their light's segment.
//   SHADOW batch  64 consecutive shadow rays of one light's segment: trace_kd with the ray's dist_limit.
//   LIGHT batch   a node batch whose shadow rays have all been answered: the second half of apply_lights
//                 (ndt.c:217-310), then the node's colour travels up the ray tree (below).
//
// and produce the items that depend on theirs:
//   * a pool batch (nodes or shadow rays) becomes a work item when all 64 of its slots have been written: every
//     writer adds the number of slots it wrote to the batch's fill counter, the one that completes it pushes the
//     batch onto the ready ring.  A wavefront that finds nothing to do CLOSES the partial batch at the end of a pool by
//     reserving the rest of it as padding slots -- so nothing ever waits for a batch to fill up;
//   * a node batch counts its unanswered shadow rays; the wavefront whose answers bring the count to zero pushes it
//     onto the lighting ring;
//   * the ray tree is resolved bottom-up as it completes (get_ray_color's blend, ndt.c:402-429): every node counts
//     what it still waits for (its own lighting + its children); whoever brings that to zero blends the node and
//     reports to its parent.  Same operands, same order of operations as the recursion: the image does not depend on
//     which wavefront did what when.
//
// Memory model (MI355X_MICROARCH.md, inter-workgroup visibility): the XCDs' L2s are not coherent and a CU's L1 is
// never refreshed, so every word that is handed from one wavefront to another INSIDE this launch is written with an
// agent-scope (sc1, write-through) store and read with an agent-scope (sc1) load; a producer drains its stores
// (s_waitcnt vmcnt(0)) before the atomic that publishes them; consumers reach the data only through an index that
// atomic (or a value derived from it) handed them.  Plain stores are used only for what later KERNELS read (the
// resolved colours of the primaries, depth maps).

#include "ndt_finish.hpp"

// ------------------------------------------------------------------ coherent loads / stores, drained publishes

#ifdef NDT_STREAM_PLAIN   /* experiment only: what the coherent accesses cost (results may be stale) */
NDT_DEV double cld(const double *p) { return *p; }
NDT_DEV void cst(double *p, double v) { *p = v; }
NDT_DEV int cldi(const int *p) { return *(const volatile int *)p; }
NDT_DEV void csti(int *p, int v) { *p = v; }
NDT_DEV unsigned long long cldu(const unsigned long long *p) { return *p; }
NDT_DEV void cstu(unsigned long long *p, unsigned long long v) { *p = v; }
#else
NDT_DEV double cld(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(const_cast<double *>(p)),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
NDT_DEV void cst(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
NDT_DEV int cldi(const int *p) { return __hip_atomic_load(const_cast<int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
NDT_DEV void csti(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
NDT_DEV unsigned long long cldu(const unsigned long long *p)
{
    return __hip_atomic_load(const_cast<unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
NDT_DEV void cstu(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
template <int K> NDT_DEV void cload_soa(const double *base, long long g, double (&r)[K])
{
    const double *t = base + (g >> 6) * (long long)(K * 64) + (g & 63);
#pragma unroll
    for (int c = 0; c < K; ++c) r[c] = cld(t + c * 64);
}
template <int K> NDT_DEV void cstore_soa(double *base, long long g, const double (&r)[K])
{
    double *t = base + (g >> 6) * (long long)(K * 64) + (g & 63);
#pragma unroll
    for (int c = 0; c < K; ++c) cst(t + c * 64, r[c]);
}
// every store (and returning atomic) this wavefront has issued has been performed
NDT_DEV void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

NDT_DEV int wave_sum(int x)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

// ------------------------------------------------------------------ queues
//
// Everything here is about NOT having hot words: one address takes ~88 atomics (or agent-scope loads) per microsecond,
// a frame has 90 000 work items, and a memory channel busy with one word makes every other access it serves wait
// (the first version of this kernel read four heads and tails before every item and ran at the speed of those words:
// 4 ms a frame whatever the number of wavefronts).

NDT_DEV unsigned long long cld64(const unsigned long long *p) { return cldu(p); }

// ---- secondary node batches: one ring per shard
NDT_DEV void sec_push(const StreamArgs &sa, int shard, int nb)
{
    const unsigned long long old = atomicAdd(&sa.ctl->sec[shard].v, 1ull);
    csti(sa.sec_ring + (long long)shard * (sa.node_batches) + (int)(old & 0xffffffffull), nb + 1);
}
// a ticket of shard `shard` if its ring is not empty (lane 0 only)
NDT_DEV int sec_ticket(const StreamArgs &sa, int shard)
{
    const unsigned long long v = cld64(&sa.ctl->sec[shard].v);
    if ((unsigned int)(v >> 32) >= (unsigned int)(v & 0xffffffffull)) return -1;
    return (int)(atomicAdd(&sa.ctl->sec[shard].v, 1ull << 32) >> 32);
}
// One secondary node batch for the wavefront (every lane calls; lane 0 works).  A ticket whose slot is still empty (two
// wavefronts saw the same last entry) stays in tk / tk_shard and is looked at again next time.  steal: also look at
// the other shards (a wavefront that has nothing else to do).
NDT_DEV bool sec_pop(const StreamArgs &sa, int home, bool steal, int &tk, int &tk_shard, int &id)
{
    int got = -1, t = tk, ts = tk_shard;
    if (__lane_id() == 0) {
        if (t < 0) {
            t = sec_ticket(sa, home);
            ts = home;
            for (int k = 1; steal && t < 0 && k < NDT_PRIM_SHARDS; ++k) {
                ts = (home + k) % NDT_PRIM_SHARDS;
                t = sec_ticket(sa, ts);
            }
        }
        if (t >= 0) {
            const int e = cldi(sa.sec_ring + (long long)ts * sa.node_batches + t);
            if (e != 0) {
                got = e - 1;
                t = -1;
            }
        }
    }
    tk = __shfl(t, 0, 64);
    tk_shard = __shfl(ts, 0, 64);
    id = __shfl(got, 0, 64);
    return id >= 0;
}

// ---- shadow batches, lighting batches: one ring each; the wavefront always holds a ticket
NDT_DEV void ring_push(int *tail, int *ring, int id)
{
    const int slot = atomicAdd(tail, 1);
    csti(ring + slot, id + 1);
}
// (every lane calls; lane 0 works) the item in the wavefront's slot, if it has been written; the next ticket is taken at
// once.  The FIRST ticket is taken when the wavefront first looks at the ring (ticket < 0) -- not at the start of the
// kernel, where 2 000 wavefronts taking two tickets each kept two words busy for 23 us, and not a fixed one per wavefront
// number either: a ticket must belong to a wavefront that is running (several contexts may share the GPU, and a workgroup
// that has not been scheduled yet would sit on the items pushed into its slots).
NDT_DEV bool ring_pop(int *head, const int *ring, int &ticket, int &id)
{
    int got = -1, tk = ticket;
    if (__lane_id() == 0) {
        if (tk < 0) tk = atomicAdd(head, 1);
        const int e = cldi(ring + tk);
        if (e != 0) {
            got = e - 1;
            tk = atomicAdd(head, 1);
        }
    }
    ticket = __shfl(tk, 0, 64);
    id = __shfl(got, 0, 64);
    return id >= 0;
}

struct WaveStats {          // what a wavefront adds to the frame's statistics when it leaves (not per item: hot words)
    int children, shadow, max_level;
    unsigned long long ref;     // fused: the reference-equivalent rays of the pixels this wavefront finished (all lanes: summed at the end)
};

// ------------------------------------------------------------------ the ray tree, bottom-up

// get_ray_color's blend of a node's own colour with what its children returned (ndt.c:402-429): resolve_node of
// ndt_frame.hip with coherent accesses.  `mat` is the global blob (materials are not staged in LDS).
NDT_DEV void stream_resolve(const double *mat, const SceneDesc &sd, const Workspace &ws, int specular, long long g)
{
    const int obj = cldi(ws.hit_obj + g);
    const int mw = sd.off_mat + 8 * obj;
    const double hitr[3] = { mat[mw + 3], mat[mw + 4], mat[mw + 5] };
    double c[3] = { cld(ws.clr + 0 * ws.cap + g), cld(ws.clr + 1 * ws.cap + g), cld(ws.clr + 2 * ws.cap + g) };
    int cnt = cldi(ws.count + g);
    const int refl = cldi(ws.child_refl + g);
    if (refl != -1) {
        double ref[3] = { 0.0, 0.0, 0.0 };
        if (refl >= 0) {
            ref[0] = cld(ws.clr + 0 * ws.cap + refl); ref[1] = cld(ws.clr + 1 * ws.cap + refl); ref[2] = cld(ws.clr + 2 * ws.cap + refl);
            cnt += cldi(ws.count + refl);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (specular) c[k] = (1 - hitr[k]) * (c[k]) + (hitr[k]) * ref[k];     // ndt.c:405-407
            else c[k] += hitr[k] * ref[k];                                         // ndt.c:411-413
        }
    }
    const int refr = cldi(ws.child_refr + g);
    if (refr != -1) {
        double ref[3] = { 0.0, 0.0, 0.0 };
        if (refr >= 0) {
            ref[0] = cld(ws.clr + 0 * ws.cap + refr); ref[1] = cld(ws.clr + 1 * ws.cap + refr); ref[2] = cld(ws.clr + 2 * ws.cap + refr);
            cnt += cldi(ws.count + refr);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] += (1.0 - hitr[k]) * ref[k];              // ndt.c:426-428
    }
    cst(ws.clr + 0 * ws.cap + g, c[0]);
    cst(ws.clr + 1 * ws.cap + g, c[1]);
    cst(ws.clr + 2 * ws.cap + g, c[2]);
    csti(ws.count + g, cnt);
}

// Lanes with `active` report to node `cur` that one of the things it waits for is done (its own lighting, or a child
// whose colour is final).  The lane that brings the count to zero blends the node and reports to ITS parent, and so on
// up to the primary.  The caller has drained the stores that made its contribution final.  (Every lane calls.)
NDT_DEV void complete_up(const double *mat, const SceneDesc &sd, const Workspace &ws, const StreamArgs &sa, const RenderGeom &rg, int cur,
                         bool active, WaveStats &stats)
{
    const int specular = rg.specular;
    while (__ballot(active) != 0ull) {
        if (active) {
            const int old = atomicSub(sa.pend + cur, 1);
            if (old != 1) {
                active = false;
            } else {
                stream_resolve(mat, sd, ws, specular, cur);
                if (cur < sa.root_begin + sa.n_primary) {
                    active = false;                 // a root of the forest: a primary.  Its pixel: here (fused), or k_finish_pixels
                    if (sa.fused) {
                        drain();                    // (the colour just blended is read back through L2)
                        stats.ref += finish_pixel<true>(mat, sd, ws, rg, N, cur, sa.rgba, sa.depth_out);
                    }
                } else {
                    cur = cldi(sa.parent + cur);
                }
            }
        }
        drain();        // the blended colours are out before the next level hears of them
    }
}

// ------------------------------------------------------------------ lighting of one node batch

// Second half of apply_lights (ndt.c:217-310) for the 64 nodes of batch nb: shade_finish_node with coherent accesses,
// returns whether this lane's node was lit (its colour then starts its way up the tree).  Every lane of the wavefront calls.
NDT_DEV bool stream_light_batch(const double *blob, const double *mat, const SceneDesc &sd, const Workspace &ws, const RenderGeom &rg,
                                const StreamArgs &sa, int nb)
{
    const long long g = (long long)nb * 64 + __lane_id();
    int obj = -1;
    const bool ours = g >= sa.root_begin + sa.n_primary || (g >= sa.valid_begin && g < sa.valid_end);
    if (ours && cldi(ws.depth_left + g) > 0) obj = cldi(ws.hit_obj + g);
    const bool shaded = obj >= 0;
    if (shaded) {
        double src[N], look[N], nrm[N], hit[N];
        cload_soa<N>(ws.ray_o, g, src);
        cload_soa<N>(ws.ray_v, g, look);
        cload_soa<N>(ws.hit_p, g, hit);
        cload_soa<N>(ws.hit_n, g, nrm);
        const int mw = sd.off_mat + 8 * obj;
        const double hit_r = mat[mw], hit_g = mat[mw + 1], hit_b = mat[mw + 2];
        const double refl_r = mat[mw + 3], refl_g = mat[mw + 4], refl_b = mat[mw + 5];
        const bool transparent = mat[mw + 7] != 0.0;
        double hitr_r = 0.0, hitr_g = 0.0, hitr_b = 0.0;
        if (rg.specular) {
            hitr_r = refl_r; hitr_g = refl_g; hitr_b = refl_b;
        }
        // apply_lights, ndt.c:88-92: scn->ambient first
        double cr = hit_r * mat[sd.off_cam + 4 * N + 1];
        double cg = hit_g * mat[sd.off_cam + 4 * N + 2];
        double cb = hit_b * mat[sd.off_cam + 4 * N + 3];
        const unsigned long long fire = cldu(ws.sh_mask + g);
        const unsigned long long key = rg.sample_keys ? cldu(ws.rng_key + g) : 0ull;
        int n_shadow = 0;
        // every lane walks its own lights -- those that fired, and the ambient ones, in the list's order (shade_finish_node)
        unsigned long long ambient = 0ull;
        for (int li = 0; li < sd.n_lights; ++li)
            if (blob_int(mat, light_word(sd, li), 0) == NDT_LIGHT_AMBIENT_) ambient |= 1ull << li;      // wave-uniform
        for (unsigned long long todo = fire | ambient; todo != 0ull; todo &= todo - 1ull) {
            const int li = __ffsll((long long)todo) - 1;
            const int w = light_word(sd, li);
            const int ltype = blob_int(mat, w, 0);
            const double lr_ = mat[w + 1], lg_ = mat[w + 2], lb_ = mat[w + 3];
            if (ltype == NDT_LIGHT_AMBIENT_) {              // ndt.c:106-111
                cr += hit_r * lr_;
                cg += hit_g * lg_;
                cb += hit_b * lb_;
                continue;
            }
            const int seg = __popcll(~ambient & ((1ull << li) - 1ull));
            const long long slot = (long long)seg * sa.seg_cap + cldi(ws.sh_idx + (long long)seg * ws.cap + g);
            int type;
            double lgt_pos[N], rev_light[N], light_vec[N], so[N], light_hit_normal[N];
            ShadowSetup ss;
            light_setup(mat, sd, li, src, hit, nrm, type, lgt_pos, rev_light, light_vec, so, ss, key);
            const int sobj = cldi(ws.sobj + slot);
            const int sprim = cldi(ws.sprim + slot);
            ++n_shadow;
            if (type == NDT_LIGHT_DIRECTIONAL_) {
                if (sobj >= 0) continue;                    // anything at all shadows it, ndt.c:246
                v_copy<N>(light_hit_normal, nrm);           // ndt.c:252-254
            } else {
                if (sobj != obj) continue;                  // ndt.c:217
                double light_hit[N];
                isect_full_stream(blob, &sd, sprim, so, light_vec, light_hit, light_hit_normal);
                const double dist = v_dist<N>(hit, light_hit);
                if (dist > NDT_EPS) continue;               // ndt.c:225
            }
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
        cst(ws.clr + 0 * ws.cap + g, cr);
        cst(ws.clr + 1 * ws.cap + g, cg);
        cst(ws.clr + 2 * ws.cap + g, cb);
        csti(ws.count + g, 1 + n_shadow);
    }
    drain();
    return shaded;          // the caller sends the lit nodes' colours up the tree (complete_up) and counts the batch's lighting done
}

// ------------------------------------------------------------------ shading of one node batch (after its trace_kd)

// What get_ray_color and apply_lights do between the closest-hit query and the shadow queries, for the 64 nodes of
// batch nb, by the wavefront that has just traced them (src / look / obj / prim are still in its registers): hit point
// and normal, background colour, children, shadow rays -- shade_emit_node with per-wavefront reservations (the
// wavefronts of a persistent launch are not in step, so there is no workgroup to share a reservation with; they also
// do not all arrive at the counters at once).  Every lane of the wavefront calls.

NDT_DEV void stream_shade_batch(const double *blob, const double *mat, const SceneDesc &sd, const Workspace &ws, const RenderGeom &rg,
                                const StreamArgs &sa, int nb, bool valid, int depth_left, const double (&src)[N],
                                const double (&look)[N], int obj, int prim, WaveStats &stats, int home, int &up_node, bool &up_active,
                                bool &light_now, bool &any_shaded, bool &final_root)
{
    up_node = 0;
    up_active = false;
    light_now = false;
    any_shaded = false;
    final_root = false;
    StreamCtl *ctl = sa.ctl;
    const int lane = __lane_id();
    const long long g = (long long)nb * 64 + lane;
    bool shaded = false;
    double hit[N], nrm[N];
    if (valid) {
        if (obj >= 0) {
            // the hit point and normal trace_kd would have returned: re-run the one primitive that won the traversal
            isect_full_stream(blob, &sd, prim, src, look, hit, nrm);
            const double trace_dist = v_dist<N>(hit, src);                  // ndt.c:365
            shaded = trace_dist > NDT_EPS;                                  // ndt.c:376
            if (rg.want_depth && sa.roots_are_primaries && g < sa.n_primary) cst(ws.depth + g, shaded ? 1.0 / trace_dist : 0.0);     // ndt.c:366-370
        } else if (rg.want_depth && sa.roots_are_primaries && g < sa.n_primary) {
            cst(ws.depth + g, 0.0);                                         // ndt.c:372-373
        }
        if (shaded) {
            cstore_soa<N>(ws.hit_p, g, hit);
            cstore_soa<N>(ws.hit_n, g, nrm);
            csti(ws.hit_obj + g, obj);
            csti(ws.hit_prim + g, prim);
        } else {
            // background (ndt.c:436-442); alpha is applied per pixel at the end.  Final at once.
            csti(ws.hit_obj + g, -1);
            cst(ws.clr + 0 * ws.cap + g, mat[sd.off_cam + 4 * N + 4]);
            cst(ws.clr + 1 * ws.cap + g, mat[sd.off_cam + 4 * N + 5]);
            cst(ws.clr + 2 * ws.cap + g, mat[sd.off_cam + 4 * N + 6]);
            csti(ws.count + g, 1);
        }
    }
    const bool live = __ballot(shaded) != 0ull;
    int n_sh_batch = 0;             // shadow rays of the whole batch
    if (live) {
        // ---- which lights fire (ndt.c:113-208): one shadow ray per light that passes the same-side / cone tests
        unsigned long long fire = 0ull;
        const unsigned long long node_key = (rg.sample_keys && valid) ? cldu(ws.rng_key + g) : 0ull;       // stochastic renders only
        if (shaded) {
            for (int li = 0; li < sd.n_lights; ++li) {
                int type;
                double lgt_pos[N], rev_light[N], light_vec[N], so[N];
                ShadowSetup ss;
                if (light_setup(mat, sd, li, src, hit, nrm, type, lgt_pos, rev_light, light_vec, so, ss, node_key)) fire |= 1ull << li;
            }
        }
        // lane s learns which lanes fire segment s's light (and so how many)
        int my_total = 0, seg = 0;
        unsigned long long my_vote = 0ull, ambient = 0ull;
        for (int li = 0; li < sd.n_lights; ++li) {
            if (blob_int(mat, light_word(sd, li), 0) == NDT_LIGHT_AMBIENT_) { ambient |= 1ull << li; continue; }      // wave-uniform
            const unsigned long long vote = __ballot((fire >> li) & 1ull);
            if (lane == seg) my_vote = vote;
            ++seg;
        }
        my_total = __popcll(my_vote);
        if (shaded) cstu(ws.sh_mask + g, fire);
        // ---- get_ray_color, ndt.c:381-430: reflection / refraction children
        bool want_refl = false, want_refr = false;
        double refl_ray[N], refr_ray[N];
        double refl_frac = 0, refr_frac = 0;
        int depth_next = 0;
        int c_refl = -1, c_refr = -1;
        if (shaded) {
            const int mw = sd.off_mat + 8 * obj;
            const double refl_r = mat[mw + 3], refl_g = mat[mw + 4], refl_b = mat[mw + 5];
            const bool transparent = mat[mw + 7] != 0.0;
            const double frac = cld(ws.frac + g);
            depth_next = depth_left - 1;
            const double gb2 = (refl_g > refl_b) ? refl_g : refl_b;
            const double contrib = (refl_r > gb2) ? refl_r : gb2;
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
                    v_refract<N>(look, nrm_u, refr_ray, mat[mw + 6]);
                    v_unitize<N>(refr_ray);
                    want_refr = true;
                }
            }
        }
        // ---- reservations: the children at the node tail, the shadow rays in their lights' segments
        const unsigned long long v_refl = __ballot(want_refl), v_refr = __ballot(want_refr);
        const int n_refl = __popcll(v_refl), total = n_refl + __popcll(v_refr);
        int base = 0;
        if (total > 0) {
            if (lane == 0) base = atomicAdd(&ctl->node_tail.v, total);
            stats.children += total;
            base = __shfl(base, 0, 64);
            if ((long long)base + total > ws.cap) {
                // node pool overflow: flag it (the host renders the frame again with a larger pool); everyone leaves
                if (lane == 0) {
                    atomicOr(&ctl->overflow.v, 1);
                    atomicMax(&ctl->abort.v, 1);
                }
                return;
            }
        }
        int my_base = 0;
        n_sh_batch = wave_sum(my_total);
        if (my_total > 0) {
            my_base = atomicAdd(&ctl->seg_tail[lane].v, my_total);
            if ((long long)my_base + my_total > sa.seg_cap) {
                atomicOr(&ctl->overflow.v, 2);
                atomicMax(&ctl->abort.v, 1);
            }
        }
        stats.shadow += n_sh_batch;
        if (__ballot(my_total > 0 && (long long)my_base + my_total > sa.seg_cap) != 0ull) return;
        // ---- the children
        const unsigned long long below = (1ull << lane) - 1ull;
        if (want_refl) {
            const long long c = (long long)base + __popcll(v_refl & below);
            cstore_soa<N>(ws.ray_o, c, hit);
            cstore_soa<N>(ws.ray_v, c, refl_ray);
            cst(ws.frac + c, refl_frac);
            csti(ws.depth_left + c, depth_next);
            csti(sa.parent + c, (int)g);
            if (rg.sample_keys) cstu(ws.rng_key + c, ndt_rng_mix(node_key ^ 0x1ull));
            c_refl = (int)c;
        }
        if (want_refr) {
            const long long c = (long long)base + n_refl + __popcll(v_refr & below);
            cstore_soa<N>(ws.ray_o, c, hit);
            cstore_soa<N>(ws.ray_v, c, refr_ray);
            cst(ws.frac + c, refr_frac);
            csti(ws.depth_left + c, depth_next);
            csti(sa.parent + c, (int)g);
            if (rg.sample_keys) cstu(ws.rng_key + c, ndt_rng_mix(node_key ^ 0x2ull));
            c_refr = (int)c;
        }
        if (shaded) {
            csti(ws.child_refl + g, c_refl);
            csti(ws.child_refr + g, c_refr);
            // what the node waits for before its colour is final: its own lighting and its children
            csti(sa.pend + g, 1 + (c_refl >= 0 ? 1 : 0) + (c_refr >= 0 ? 1 : 0));
        }
        if (total > 0) {
            // statistics: the deepest bounce that has nodes
            int lvl = (want_refl || want_refr) ? rg.max_depth - depth_next : 0;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) lvl = max(lvl, __shfl_xor(lvl, d, 64));
            stats.max_level = max(stats.max_level, lvl);
        }
        // ---- the shadow rays, into their segments: every lane walks the lights it fires (shade_emit_node)
        int rounds = __popcll(fire);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) rounds = max(rounds, __shfl_xor(rounds, d, 64));
        unsigned long long todo = fire;
        for (int r = 0; r < rounds; ++r) {
            const bool fires = todo != 0ull;
            const int li = fires ? __ffsll((long long)todo) - 1 : 0;
            todo &= todo - 1ull;
            const int seg = __popcll(~ambient & ((1ull << li) - 1ull));
            const int sbase = __shfl(my_base, seg, 64);
            const unsigned long long vote = ((unsigned long long)(unsigned int)__shfl((int)(my_vote >> 32), seg, 64) << 32)
                                            | (unsigned long long)(unsigned int)__shfl((int)my_vote, seg, 64);
            if (fires) {
                int type;
                double lgt_pos[N], rev_light[N], light_vec[N], so[N];
                ShadowSetup ss;
                light_setup(mat, sd, li, src, hit, nrm, type, lgt_pos, rev_light, light_vec, so, ss, node_key);
                const int idx = sbase + __popcll(vote & below);
                const long long slot = (long long)seg * sa.seg_cap + idx;
                csti(ws.sh_idx + (long long)seg * ws.cap + g, idx);
                cstore_soa<N>(ws.so, slot, so);
                // point/spot: from the light along light_vec (ndt.c:211); directional: from the nudged hit point along
                // rev_light (ndt.c:238) -- one store of a selected VALUE (two stores from different arrays end in scratch)
                double dir[N];
#pragma unroll
                for (int c = 0; c < N; ++c) dir[c] = (type == NDT_LIGHT_DIRECTIONAL_) ? rev_light[c] : light_vec[c];
                cstore_soa<N>(ws.sv, slot, dir);
                cst(ws.slim + slot, ss.dist_limit);
                csti(sa.sowner + slot, (int)g);
            }
        }
        // the batch's unanswered shadow rays: counted BEFORE any of them can be answered
        if (n_sh_batch > 0 && lane == 0) atomicAdd(sa.sh_pending + nb, n_sh_batch);
        drain();
        // ---- publish: every pool batch we wrote into learns how many of its slots are now written
        if (total > 0 && lane < 3) {
            const int kb = (base >> 6) + lane;
            if (kb <= ((base + total - 1) >> 6)) {
                const int lo = base > kb * 64 ? base : kb * 64, hi = base + total < (kb + 1) * 64 ? base + total : (kb + 1) * 64;
                atomicAdd(&ctl->outstanding[kb % NDT_PRIM_SHARDS].v, 2 * (hi - lo));      // performed before our own "done" below (drain)
                const int old = atomicAdd(sa.node_fill + kb, hi - lo);
                if (old + (hi - lo) == 64) sec_push(sa, home, kb);
            }
        }
        if (my_total > 0) {
            const int bps = sa.seg_cap >> 6;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = (my_base >> 6) + j;
                if (k <= ((my_base + my_total - 1) >> 6)) {
                    const int lo = my_base > k * 64 ? my_base : k * 64, hi = my_base + my_total < (k + 1) * 64 ? my_base + my_total : (k + 1) * 64;
                    const int old = atomicAdd(sa.sh_fill + lane * bps + k, hi - lo);
                    if (old + (hi - lo) == 64) ring_push(&ctl->sh_tail.v, sa.sh_ring, lane * bps + k);
                }
            }
        }
        drain();        // the new slots are counted as outstanding before this batch counts itself done
    } else {
        drain();
    }
    // ---- nodes that are final already (background): their parents hear of it (the caller's complete_up); a root's pixel
    // can be written (fused)
    up_active = valid && !shaded && g >= sa.root_begin + sa.n_primary;
    final_root = valid && !shaded && g < sa.root_begin + sa.n_primary;
    if (up_active) up_node = cldi(sa.parent + g);
    // ---- the batch's lighting: later (when its shadow rays are answered), now (it has none), or never (nothing was hit)
    any_shaded = live;
    light_now = live && n_sh_batch == 0;
}

// ------------------------------------------------------------------ idle wavefronts close the partial batches

// Reserve the rest of the batch at the end of the node pool / of every shadow segment as padding slots, so that
// the batch becomes a work item.  Only called by a wavefront that found no work: while there is other work, batches
// fill up by themselves.  Every lane calls.
NDT_DEV void stream_close_partials(const Workspace &ws, const StreamArgs &sa, int home)
{
    StreamCtl *ctl = sa.ctl;
    const int lane = __lane_id();
    // node pool
    {
        int t = 0, won = 0;
        if (lane == 0) {
            t = cldi(&ctl->node_tail.v);
            if ((t & 63) != 0 && t < ws.cap) won = atomicCAS(&ctl->node_tail.v, t, (t + 63) & ~63) == t;
        }
        t = __shfl(t, 0, 64);
        won = __shfl(won, 0, 64);
        if (won) {
            const int pad = 64 - (t & 63);
            if (lane < pad) {
                csti(ws.depth_left + t + lane, 0);      // never traced, never shaded
                csti(ws.hit_obj + t + lane, -1);
            }
            drain();
            if (lane == 0) {
                atomicAdd(&ctl->outstanding[(t >> 6) % NDT_PRIM_SHARDS].v, 2 * pad);
                const int old = atomicAdd(sa.node_fill + (t >> 6), pad);
                if (old + pad == 64) sec_push(sa, home, t >> 6);
            }
        }
    }
    // shadow segments: lane s closes segment s
    {
        int t = 0, won = 0;
        if (lane < sa.n_seg) {
            t = cldi(&ctl->seg_tail[lane].v);
            if ((t & 63) != 0 && t < sa.seg_cap) won = atomicCAS(&ctl->seg_tail[lane].v, t, (t + 63) & ~63) == t;
        }
        unsigned long long winners = __ballot(won != 0);
        while (winners) {
            const int s = __ffsll((long long)winners) - 1;
            winners &= winners - 1;
            const int ts = __shfl(t, s, 64);
            const int pad = 64 - (ts & 63);
            if (lane < pad) csti(sa.sowner + (long long)s * sa.seg_cap + ts + lane, -1);
            drain();
            if (lane == 0) {
                const int bps = sa.seg_cap >> 6;
                const int old = atomicAdd(sa.sh_fill + s * bps + (ts >> 6), pad);
                if (old + pad == 64) ring_push(&ctl->sh_tail.v, sa.sh_ring, s * bps + (ts >> 6));
            }
        }
    }
}

// ------------------------------------------------------------------ the kernel

// 512 lanes per workgroup at most = two wavefronts per SIMD = 256 registers each: with three (768 lanes, 168 registers) the
// lighting code spilled 800 bytes per lane, 3 072 wavefronts x 51 KB of scratch went through L2 to HBM (0.95 GB written per
// 1080p frame) and every other memory operation of the kernel queued behind that traffic.
#ifndef NDT_STREAM_MAX_BLOCK
#define NDT_STREAM_MAX_BLOCK 512
#endif
template <int MW, bool LDS, bool LSTACK = false>
__global__ void __launch_bounds__(NDT_STREAM_MAX_BLOCK) k_frame_stream(const double *gblob, SceneDesc sd, Workspace ws, RenderGeom rg,
                                                                      StreamArgs sa)
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
    KdStackLds kstack{};
    if (LSTACK) {
        double *base = lds_blob + ((sd.trace_words + 1) & ~1);
        const int depth = sd.kd_depth + 1;
        kstack.stride = blockDim.x;
        kstack.tu = base + threadIdx.x;
        kstack.node = (int *)(base + (size_t)depth * blockDim.x) + threadIdx.x;
    }
    VisitMask<MW> mask;
    init_visit_mask<MW>(mask, gblob, sd, ws);
    // global-memory tier: the rays' projections on the item boxes' frame, one LDS slot per lane (ndt_device.hpp:item_box_meets)
    double *box_slot = nullptr;
    if (MW == 0 && !LDS && sd.off_obox > 0) box_slot = lds_blob + (size_t)(threadIdx.x >> 6) * (N * 128) + 2 * (threadIdx.x & 63);
    // Idle wavefronts: ONE per workgroup (the first: the watcher) looks at the frame's global words -- is everything done, are
    // there partial batches to close -- and tells the others through LDS.  With every idle wavefront doing that (a dozen
    // agent-scope loads of the same few words per round) a launch with more wavefronts than work spent its time in the queue
    // of those words' memory channels: the deep bounces of the benchmark frame, 0.6 M rays, took 1.3 ms.
    __shared__ int wg_over;
    if (threadIdx.x == 0) wg_over = 0;
    __syncthreads();
    const bool watcher = threadIdx.x < 64;
    StreamCtl *ctl = sa.ctl;
    const int lane = __lane_id();
    const int n_prim_batches = sa.n_primary >> 6;
    const int home = blockIdx.x % NDT_PRIM_SHARDS;              // this workgroup's shard of the sharded queues
    int shard = home;
    unsigned shards_alive = (1u << NDT_PRIM_SHARDS) - 1u;
    const unsigned long long t_begin = wall_clock64();          // 100 MHz
    int idle_rounds = 0;
    int tk_sec = -1, tk_sec_shard = home;                       // a ticket of a secondary ring whose slot was still empty (sec_pop)
    // the wavefront's slots in the ring of shadow batches and in the ring of lighting batches (ring_pop): always held
    int tk_sh = -1, tk_fin = -1;
    // NDT_HIP_STREAM_PROBE: items and 100 MHz ticks per kind of work, idle rounds, first / last item
    unsigned int pr_n[4] = { 0, 0, 0, 0 }, pr_t[4] = { 0, 0, 0, 0 }, pr_first = 0, pr_last = 0;
    unsigned int pr_pop = 0, pr_load = 0, pr_trace = 0;         // of the node and shadow items: looking for work, loading rays, trace_kd
    unsigned int pr_trace_sh = 0, pr_up = 0;                    // trace_kd of the shadow items alone; complete_up + counters
    unsigned long long pr_mark = 0, pr_top = 0;
    WaveStats stats = { 0, 0, 0, 0ull };
    int items = 0;
    int light_next = -1;            // a node batch this wavefront shaded that has no shadow rays to wait for: lit next
    while (true) {
        // (the abort word is looked at by idle wavefronts and now and then by busy ones)
        if ((items++ & 15) == 15 && cldi(&ctl->abort.v) != 0) break;
        if (sa.wave_log) pr_top = wall_clock64();
        // ---- take a work item: deeper rays first (they start the chains everything else waits for), lighting last
        int kind = 0, id = -1;          // 1 node batch, 2 shadow batch, 3 lighting batch
        if (light_next >= 0) {
            kind = 3;
            id = light_next;
            light_next = -1;
        } else if (sec_pop(sa, home, false, tk_sec, tk_sec_shard, id)) {
            kind = 1;
        } else {
            while (shards_alive != 0u && kind == 0) {
                if ((shards_alive >> shard) & 1u) {
                    int k = 0;
                    if (lane == 0) k = atomicAdd(&ctl->prim_head[shard].v, 1);
                    k = __shfl(k, 0, 64);
                    const int b = k * NDT_PRIM_SHARDS + shard;
                    if (b < n_prim_batches) {
                        kind = 1;
                        id = (sa.root_begin >> 6) + b;
                    } else {
                        shards_alive &= ~(1u << shard);
                    }
                }
                if (kind == 0) shard = (shard + 1) % NDT_PRIM_SHARDS;
            }
            if (kind == 0) {
                if (ring_pop(&ctl->sh_head.v, sa.sh_ring, tk_sh, id)) kind = 2;
                else if (ring_pop(&ctl->fin_head.v, sa.fin_ring, tk_fin, id)) kind = 3;
                else if ((watcher || (idle_rounds & 3) == 3) && sec_pop(sa, home, true, tk_sec, tk_sec_shard, id))
                    kind = 1;       // nothing at home: the other shards (eight words: not every wavefront every round)
            }
        }
        if (sa.wave_log && kind != 0) {
            pr_mark = wall_clock64();
            if (pr_first == 0) pr_first = (unsigned int)(pr_mark - t_begin) | 1u;
            pr_pop += (unsigned int)(pr_mark - pr_top);
        }
        int up_node = 0;                // what the item leaves to do: nodes whose colour is final report to their parents ...
        bool up_active = false;
        int parts_done = 0;             // ... and how many halves (shading, lighting) of batch `id` are complete
        if (kind == 1 || kind == 2) {
            idle_rounds = 0;
            // ---- trace_kd for 64 rays: the nodes of batch `id`, or the shadow rays of batch `id`
            const bool is_shadow = kind == 2;
            long long slot;
            if (is_shadow) {
                const int bps = sa.seg_cap >> 6;
                const int s = id / bps, k = id - s * bps;
                slot = (long long)s * sa.seg_cap + (long long)k * 64 + lane;
            } else {
                slot = (long long)id * 64 + lane;
            }
            // fused: a root batch's primaries are made here (k_primary's work), not read
            const bool make = sa.fused != 0 && !is_shadow && slot < sa.root_begin + sa.n_primary;        // (wave-uniform)
            double o[N], v[N];
            int tag;
            if (make) tag = primary_node<true>(gblob, sd, ws, rg, slot, o, v) ? rg.max_depth : 0;
            else tag = is_shadow ? cldi(sa.sowner + slot) : cldi(ws.depth_left + slot);                  // owner node / bounces left
            // (a slot of a root batch outside the roots' range is not this launch's node)
            const bool valid = is_shadow ? tag >= 0
                                         : tag > 0 && (slot >= sa.root_begin + sa.n_primary || (slot >= sa.valid_begin && slot < sa.valid_end));
            double lim = -1.0;
            int obj = -1, prim = -1;
            if (valid) {
                if (!make) {
                    cload_soa<N>(is_shadow ? ws.so : ws.ray_o, slot, o);
                    cload_soa<N>(is_shadow ? ws.sv : ws.ray_v, slot, v);
                }
                if (is_shadow) lim = cld(ws.slim + slot);
                unsigned long long pr_a = 0;
                if (sa.wave_log) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    pr_a = wall_clock64();
                    pr_load += (unsigned int)(pr_a - pr_mark);
                }
#ifdef NDT_PHASE_TIMING
                {   // (diagnostic build: the phase stamps of the per-bounce trace kernel are not collected here)
                    unsigned long long ph[8] = {};
                    unsigned int cnt[8] = {}, occ[8] = {};
                    trace_kd<N, MW, LSTACK>(blob, sd, mask, o, v, lim, obj, prim, ph, cnt, occ, kstack, ClsLds{}, true, box_slot);
                }
#else
                trace_kd<N, MW, LSTACK>(blob, sd, mask, o, v, lim, obj, prim, kstack, ClsLds{}, true, box_slot);
#endif
                if (sa.wave_log) {
                    const unsigned int dt = (unsigned int)(wall_clock64() - pr_a);
                    pr_trace += dt;
                    if (is_shadow) pr_trace_sh += dt;
                }
            } else {
#pragma unroll
                for (int c = 0; c < N; ++c) { o[c] = 0.0; v[c] = 0.0; }
            }
            if (is_shadow) {
                if (valid) {
                    csti(ws.sobj + slot, obj);
                    csti(ws.sprim + slot, prim);
                }
                drain();
                // the rays of one node batch are consecutive in a segment: the first lane of every run reports the whole run
                const int nb = tag >> 6;
                const int nb_prev = __shfl_up(nb, 1, 64);
                const bool valid_prev = __shfl_up((int)valid, 1, 64) != 0;
                const bool head = valid && (lane == 0 || !valid_prev || nb_prev != nb);
                const unsigned long long heads = __ballot(head), valids = __ballot(valid);
                if (head) {
                    const unsigned long long above = (lane < 63) ? (heads >> (lane + 1)) << (lane + 1) : 0ull;
                    const int end = above ? __ffsll((long long)above) - 1 : 64;         // the next run's first lane
                    const unsigned long long span = (end < 64 ? (1ull << end) - 1ull : ~0ull) & ~((1ull << lane) - 1ull);
                    const int run = __popcll(valids & span);
                    const int old = atomicSub(sa.sh_pending + nb, run);
                    if (old == run) ring_push(&ctl->fin_tail.v, sa.fin_ring, nb);
                }
            } else {
                bool light_now, any_shaded, final_root;
                stream_shade_batch(blob, gblob, sd, ws, rg, sa, id, valid, tag, o, v, obj, prim, stats, home, up_node, up_active, light_now,
                                   any_shaded, final_root);
                if (sa.fused && final_root) stats.ref += finish_pixel<true>(gblob, sd, ws, rg, N, slot, sa.rgba, sa.depth_out);
                parts_done = any_shaded ? 1 : 2;        // nothing hit: no lighting to wait for
                if (light_now) light_next = id;
            }
        } else if (kind == 3) {
            idle_rounds = 0;
            up_active = stream_light_batch(blob, gblob, sd, ws, rg, sa, id);
            up_node = id * 64 + lane;
            parts_done = 1;
        }
        if (kind == 1 || kind == 3) {
            unsigned long long pr_b = 0;
            if (sa.wave_log) pr_b = wall_clock64();
            complete_up(gblob, sd, ws, sa, rg, up_node, up_active, stats);
            if (sa.wave_log) pr_up += (unsigned int)(wall_clock64() - pr_b);
            if (lane == 0) atomicSub(&ctl->outstanding[id % NDT_PRIM_SHARDS].v, 64 * parts_done);
        }
        if (sa.wave_log && kind != 0) {
            const unsigned long long now = wall_clock64();
            pr_n[kind] += 1;
            pr_t[kind] += (unsigned int)(now - pr_mark);
            pr_last = (unsigned int)(now - t_begin);
        }
        if (kind == 0) {
            pr_n[0] += 1;
            // ---- nothing to do right now: is the frame done?  Every shard's outstanding count zero, and the node tail the
            // same before and after reading them (anything that creates work moves the tail first): nothing is in flight and
            // nothing can appear any more.
            if (watcher) {
                bool over = cldi(&ctl->abort.v) != 0;
                if (!over) {
                    const int tail0 = cldi(&ctl->node_tail.v);
                    int left = (lane < NDT_PRIM_SHARDS) ? cldi(&ctl->outstanding[lane].v) : 0;
                    over = __ballot(left != 0) == 0ull && cldi(&ctl->node_tail.v) == tail0;
                }
                if (over) {
                    if (lane == 0) __hip_atomic_store(&wg_over, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
                // every second idle round: make work items of the partial batches at the ends of the pools
                if ((idle_rounds & 1) == 0) stream_close_partials(ws, sa, home);
            } else if (__hip_atomic_load(&wg_over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {
                break;
            }
            ++idle_rounds;
            if (idle_rounds < 4) __builtin_amdgcn_s_sleep(8);
            else if (idle_rounds < 16) __builtin_amdgcn_s_sleep(32);
            else __builtin_amdgcn_s_sleep(127);
            if ((idle_rounds & 63) == 0 && wall_clock64() - t_begin > 1000000000ull) {    // 10 s: never on a healthy frame
                if (lane == 0) {
                    atomicMax(&ctl->abort.v, 2);
                    ctl->timeout_where.v = 2;
                }
                break;
            }
        }
    }
    if (sa.fused) {
        unsigned long long ref = stats.ref;
        for (int d = 32; d > 0; d >>= 1) ref += __shfl_xor(ref, d, 64);
        if (lane == 0 && ref) atomicAdd(ws.ref_rays + 8 * ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 63), ref);
    }
    if (lane == 0) {
        if (stats.children) atomicAdd(&ctl->n_children.v, stats.children);
        if (stats.shadow) atomicAdd(&ctl->n_shadow.v, stats.shadow);
        if (stats.max_level) atomicMax(&ctl->max_level.v, stats.max_level);
    }
    if (sa.wave_log && lane == 0) {
        const unsigned int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        if (w < NDT_STREAM_LOG_WAVES) {
            unsigned int *q = sa.wave_log + 24 * w;
            q[0] = pr_n[1]; q[1] = pr_n[2]; q[2] = pr_n[3]; q[3] = pr_n[0];
            q[4] = pr_t[1]; q[5] = pr_t[2]; q[6] = pr_t[3];
            q[7] = pr_first; q[8] = pr_last; q[9] = (unsigned int)(wall_clock64() - t_begin) | 1u;
            q[10] = (unsigned int)t_begin;
            q[11] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            q[12] = pr_pop; q[13] = pr_load; q[14] = pr_trace; q[15] = pr_trace_sh; q[16] = pr_up;
        }
    }
}

#define NDT_LAUNCH_STREAM(kernel, grid, block, lds)                                                            \
    do {                                                                                                       \
        if (ev_start)                                                                                          \
            hipExtLaunchKernelGGL((kernel), dim3((unsigned)(grid)), dim3(block), (std::uint32_t)(lds), s, ev_start, ev_stop, 0u, blob, sd, ws, rg, sa); \
        else                                                                                                   \
            hipLaunchKernelGGL((kernel), dim3((unsigned)(grid)), dim3(block), lds, s, blob, sd, ws, rg, sa);   \
    } while (0)

// Persistent grid: as many workgroups as are resident (fewer for a frame with little work: a wavefront without work
// only polls).  Same tiers as launch_trace: scene and traversal stack in LDS with 768-lane workgroups where they fit,
// the scene in LDS with a scratch stack, or everything in global memory with visit masks in the slab.
static void launch_frame_stream(hipStream_t s, const double *blob, SceneDesc sd, Workspace ws, RenderGeom rg, StreamArgs sa, int tier,
                                int mask_words, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const long long batches = sa.n_primary >> 6;
    // work items of a frame: a few per primary batch; a grid beyond that cannot help
    auto grid_for_work = [&](int res, int block) {
        long long want = (batches * 3 + (block / 64) - 1) / (block / 64) + 1;
        // the rings carry a margin of NDT_STREAM_LOG_WAVES entries beyond the last batch: one per wavefront that may hold a
        // ticket for a slot nobody writes (ensure_stream_args), so the grid never has more wavefronts than that
        const long long by_margin = NDT_STREAM_LOG_WAVES / (block / 64);
        if (res > by_margin) res = (int)by_margin;
        // (more workgroups than are resident would be safe -- a ticket is only ever held by a running wavefront -- but
        // they would only start when others leave: measured with 256-lane workgroups, of which the occupancy query
        // admits one per CU and so does the hardware, half of a 2x grid started when the frame was over)
        return (int)(want < res ? want : res);
    };
    if (tier == 0) {
        const size_t lds = (size_t)sd.trace_words * sizeof(double);
        int lstack_block = NDT_STREAM_MAX_BLOCK;
        if (batches < 1024) lstack_block = 256;         // a small frame is pure latency: one wavefront per SIMD
#ifdef NDT_STREAM_KNOBS
        if (getenv("NDT_STREAM_BLOCK")) lstack_block = atoi(getenv("NDT_STREAM_BLOCK"));
#endif
        const size_t lds_stack = ((size_t)((sd.trace_words + 1) & ~1) * 8) + (size_t)lstack_block * (sd.kd_depth + 1) * 12;
        if (lds_stack <= 160 * 1024 && mask_words <= 1) {
            const int res = resident_blocks(k_frame_stream<1, true, true>, lstack_block, lds_stack);
            NDT_LAUNCH_STREAM((k_frame_stream<1, true, true>), grid_for_work(res, lstack_block), lstack_block, lds_stack);
        } else if (lds_stack <= 160 * 1024) {
            const int res = resident_blocks(k_frame_stream<NDT_MASK_REG_WORDS, true, true>, lstack_block, lds_stack);
            NDT_LAUNCH_STREAM((k_frame_stream<NDT_MASK_REG_WORDS, true, true>), grid_for_work(res, lstack_block), lstack_block, lds_stack);
        } else if (mask_words <= 1) {
            const int res = resident_blocks(k_frame_stream<1, true>, NDT_TRACE_BLOCK, lds);
            NDT_LAUNCH_STREAM((k_frame_stream<1, true>), grid_for_work(res, NDT_TRACE_BLOCK), NDT_TRACE_BLOCK, lds);
        } else {
            const int res = resident_blocks(k_frame_stream<NDT_MASK_REG_WORDS, true>, NDT_TRACE_BLOCK, lds);
            NDT_LAUNCH_STREAM((k_frame_stream<NDT_MASK_REG_WORDS, true>), grid_for_work(res, NDT_TRACE_BLOCK), NDT_TRACE_BLOCK, lds);
        }
    } else {
        const size_t lds = sd.off_obox > 0 ? (size_t)(NDT_TRACE_BLOCK / 64) * N * 128 * sizeof(double) : 0;      // the item boxes' ray slots
        int res = resident_blocks(k_frame_stream<0, false>, NDT_TRACE_BLOCK, lds);
        const long long max_blocks = ws.mask_slab_lanes / NDT_TRACE_BLOCK;
        if (res > max_blocks) res = (int)max_blocks;
        NDT_LAUNCH_STREAM((k_frame_stream<0, false>), grid_for_work(res, NDT_TRACE_BLOCK), NDT_TRACE_BLOCK, lds);
    }
}
