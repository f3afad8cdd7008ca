// ndt_device.hpp -- device-side ray math for the gfx950 tracer, templated on the dimension N.
//
// Everything here is the per-ray arithmetic of the reference's hot path, written for one
// ray per lane of a 64-wide wavefront: vectors are register arrays double[N] with N a
// compile-time constant (every loop unrolls, every index is static), the scene is one
// read-only blob of 8-byte words that a kernel addresses either in LDS or in global memory,
// and nothing allocates (the reference mallocs ~6 temporaries per ray).
//
// Arithmetic contract (same as the oracle): IEEE double, compiled with -ffp-contract=off,
// dot products summed in the SSE2 lane-pair order of vectNd.h:215-227.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NDT_EPS   (1e-4)                    /* object.h:15 */
#define NDT_EPS2  ((NDT_EPS) * (NDT_EPS))   /* object.h:17 */
#define NDT_INV_EPS2 (1.0 / (NDT_EPS2))     /* kd-tree.c:480 */
#define NDT_DBL_MAX 1.7976931348623157e308
#define NDT_PI 3.14159265358979323846       /* M_PI */

#define NDT_DEV __device__ __forceinline__

// object header flag bits (word 0, low int)
#define NDT_F_TYPE_MASK   0xff
#define NDT_F_GATE        0x100     /* bounds.radius > 0: bounding-sphere gate (object.c:618) */
#define NDT_F_INF_ENDS    0x200     /* cylinder.c:87 / hcylinder.c:107 */
#define NDT_F_USE_NORMALS 0x400     /* hfacet.c:283 */
#define NDT_F_TRANSPARENT 0x800
#define NDT_F_BOX         0x1000    /* hcube: hull box rows at its parameter offset (ndt_blob.hip:hcube_hull_box) */
#define NDT_F_FACEBOX     0x2000    /* ... followed by the mask of possible faces and every face's own box in that frame */
#define NDT_F_OBOX        0x4000    /* the item carries a box in the scene's frame (ndt_blob.hip:scene_item_boxes) */
#define NDT_F_FACETREE    0x8000    /* hcube of more than 63 faces: behind the face boxes, the boxes of every aligned run of 2^j faces (hull_faces) */
#define NDT_F_FACEGROUPS  0x10000   /* ... and, between the face boxes and that hierarchy, the faces indexed by the hull axes their boxes are thin on (hull_faces) */

enum { T_SPHERE = 0, T_HPLANE, T_HDISK, T_CYLINDER, T_HCYLINDER, T_ORTHOTOPE, T_HCUBE, T_HFACET, T_FACET };
// light_type numbering of the reference, scene.h:23-31
enum { NDT_LIGHT_AMBIENT_ = 0, NDT_LIGHT_POINT_ = 1, NDT_LIGHT_DIRECTIONAL_ = 2, NDT_LIGHT_SPOT_ = 3 };

// Offsets (in 8-byte words) of the sections of the scene blob.  Passed by value as a kernel
// argument, so every field is wave-uniform and lives in SGPRs.
struct SceneDesc {
    int n_items, n_objects, n_kd_nodes, n_inf, n_lights;
    int off_kd;        // 2 words per node: {int dim, int right} {double boundary | int first, int num}
    int off_leaf;      // 1 word per reference {int object, int header flags}: items of all leaves
    int off_inf;       // same, infinite objects
    int off_hdr;       // 2 words per object: {int flags, int param_off} {int aux0, int aux1}
    int off_bs;        // (N+2) words per object: center[N], radius, radius^2
    int off_bb;        // lower[N], upper[N]
    int off_child;     // same, nested primitives of composites
    int off_params;    // per-type parameter records
    int trace_words;   // everything above: what the trace kernel stages in LDS
    int off_mat;       // 8 words per object: rgb, reflect rgb, refract index, transparent
    int off_lights;    // per light: {int type,0}, r,g,b, angle, pos[N], dir[N], radius, u1[N], v1[N]
    int off_cam;       // pos[N], img_orig[N], dir_x[N], dir_y[N], focal, ambient[3], background[4]
    int total_words;
    int mask_words;    // 64-bit words of visit mask per ray (kd-tree.c:600)
    int kd_depth;      // deepest leaf of the kd-tree (root = 1): the traversal stack never holds more entries
    // mask_words == 1 (at most 64 items, every leaf list and the infinite list ascending in item number -- what the
    // reference's kd builder produces, checked at upload): a leaf's second word is the SET of its items as a 64-bit mask
    // instead of {first, num}, and the infinite list is this mask (trace_kd, "item sets")
    unsigned long long inf_bits;
    int off_nset;      // ... and one word per kd node: the set of the items of all leaves below it (a leaf: its own items)
    unsigned long long gate_bits;   // item sets, option "gate_prepass": the items behind a bounding-sphere gate (0: no prepass)
    // global-memory tier with ascending leaf lists (VisitMask<0>, "leaf history"): a kd leaf's record names its ordinal
    // (high half of word 0); per leaf, mask_words words = its items as a bit set, and one word {first, num} = its list
    int off_lset, off_lrange;      // off_lset == 0: no history, visit masks live in the slab
    int hist_cap;                  // history entries per ray (4; tests: 1 .. 3 force the replay into the slab)
    int cls_par_words;             // > 0: the coherent leaf scan is on; the longest parameter record of a leaf item (even)
    int cls_min_group;             // lanes of a wavefront that must stand on the same leaf to scan it together
    // item boxes (global-memory tier): one orthonormal frame, N x axis[N], and per item N x { centre, half extent }
    int off_oframe, off_obox;      // off_obox == 0: none.  An item's slabs are stored thinnest first ...
    int off_oord;                  // ... one word per item: the frame axis of its j-th slab in bits 4j .. 4j+3
    unsigned long long ambient_bits;    // bit l: light l is an ambient one (fires no shadow ray, has no segment of the shadow queue)
    int light_origins;                  // 1: the shadow rays of point / spot lights are stored without their origin (k_trace takes the light's position)
};

// ------------------------------------------------------------------ random streams
// The stochastic paths (-n > 1: jitter + lens; area lights) draw from counter-based streams: every
// number is a hash of (stream key, draw index), so nothing depends on scheduling, sharding or the
// order in which a bounce was compacted.  A primary sample's key comes from its image pixel and
// sample number; a child ray's key from its parent's key and its kind.
NDT_DEV unsigned long long ndt_rng_mix(unsigned long long z)
{
    z += 0x9e3779b97f4a7c15ull;                 // splitmix64
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
NDT_DEV double ndt_rng_uniform(unsigned long long key, unsigned int k)
{
    return (double)(ndt_rng_mix(key + k) >> 11) * (1.0 / 9007199254740992.0);      // [0, 1)
}

// ------------------------------------------------------------------ libm
// acos / asin / sin / cos / pow as real functions (one copy each per translation unit) instead of ocml's inline
// expansions at every call site: the frame kernel had ~25 copies of acos alone (three per facet intersection, and
// isect is itself instantiated three times), its code was 150 KB and most of its register spills sat around them.
// Same ocml routines, same results.
#ifndef NDT_INLINE_LIBM
#define NDT_LIBM __device__ __attribute__((noinline))
#else
#define NDT_LIBM __device__ __forceinline__
#endif
// (measured per dimension, bounce-synchronous pipeline, 1080p: out of line 3-D 0.67 / 4-D 1.65 ms against 0.63 / 1.62 inlined --
// the small-vector kernels have the registers, a call costs them more than it saves -- but 6-D 2.16 / 8-D 6.80 ms against
// 2.24 / 6.95: ndt_kernels.hip compiles the 3-D .. 5-D kernels with NDT_INLINE_LIBM)
NDT_LIBM double nd_acos(double x) { return acos(x); }
NDT_LIBM double nd_asin(double x) { return asin(x); }
NDT_LIBM double nd_sin(double x) { return sin(x); }
NDT_LIBM double nd_cos(double x) { return cos(x); }
NDT_LIBM double nd_pow(double x, double y) { return pow(x, y); }

// ------------------------------------------------------------------ vectNd.h

template <int N> NDT_DEV double v_dot(const double (&a)[N], const double (&b)[N])
{
    // vectNd_dot, vectNd.h:215-227: lane 0 sums even components, lane 1 odd components
    double s0 = a[0] * b[0];
    double s1 = a[1] * b[1];
#pragma unroll
    for (int i = 2; i < N; i += 2) {
        s0 = s0 + a[i] * b[i];
        if (i + 1 < N) s1 = s1 + a[i + 1] * b[i + 1];
    }
    return s0 + s1;
}
template <int N> NDT_DEV void v_add(const double (&a)[N], const double (&b)[N], double (&r)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = a[i] + b[i];
}
template <int N> NDT_DEV void v_sub(const double (&a)[N], const double (&b)[N], double (&r)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = a[i] - b[i];
}
template <int N> NDT_DEV void v_scale(const double (&a)[N], double s, double (&r)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = a[i] * s;
}
template <int N> NDT_DEV void v_copy(double (&d)[N], const double (&s)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = s[i];
}
template <int N> NDT_DEV void v_zero(double (&d)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = 0.0;
}
template <int N> NDT_DEV double v_len(const double (&a)[N]) { return sqrt(v_dot<N>(a, a)); }
template <int N> NDT_DEV void v_unitize(double (&a)[N])
{
    // vectNd_unitize, vectNd.h:323
    double len = v_len<N>(a);
    if (len > NDT_EPS || len < -NDT_EPS) v_scale<N>(a, 1.0 / len, a);
}
template <int N> NDT_DEV double v_dist(const double (&a)[N], const double (&b)[N])
{
    double d[N];
    v_sub<N>(a, b, d);
    return v_len<N>(d);
}
template <int N> NDT_DEV void v_proj_unit(const double (&v)[N], const double (&onto)[N], double (&r)[N])
{
    double ab = v_dot<N>(v, onto);
    v_scale<N>(onto, ab, r);
}
template <int N> NDT_DEV void v_proj(const double (&v)[N], const double (&onto)[N], double (&r)[N])
{
    double bb = v_dot<N>(onto, onto);
    double ab = v_dot<N>(v, onto);
    v_scale<N>(onto, ab / bb, r);
}
template <int N> NDT_DEV double v_angle(const double (&a)[N], const double (&b)[N])
{
    // vectNd_angle, vectNd.c:64
    double dp = v_dot<N>(a, b);
    double l1 = v_len<N>(a);
    double l2 = v_len<N>(b);
    double div = l1 * l2;
    if (fabs(div) > NDT_EPS) return nd_acos(dp / div);
    return -1;
}
template <int N> NDT_DEV double v_sum(const double (&a)[N])
{
    // vectNd_dot(a, ones), hfacet.c:241: same lane order, multiplications by 1.0 are exact
    double s0 = a[0];
    double s1 = a[1];
#pragma unroll
    for (int i = 2; i < N; i += 2) {
        s0 = s0 + a[i];
        if (i + 1 < N) s1 = s1 + a[i + 1];
    }
    return s0 + s1;
}
template <int N> NDT_DEV void v_reflect(const double (&u)[N], const double (&nrm)[N], double (&res)[N], double mag)
{
    // vectNd_reflect, vectNd.c:101
    double nnu[N];
    double nu = v_dot<N>(nrm, u);
    double nn = v_dot<N>(nrm, nrm);
    v_scale<N>(nrm, (1 + mag) * nu / nn, nnu);
    v_sub<N>(u, nnu, res);
}
template <int N> NDT_DEV void v_refract(const double (&u)[N], double (&nrm)[N], double (&res)[N], double index)
{
    // vectNd_refract, vectNd.c:119 (unitizes nrm in place, vectNd.c:155)
    double rev_u[N], rev_n[N], un[N], np[N], ref_n[N], ref_p[N];
    v_scale<N>(u, -1, rev_u);
    v_scale<N>(nrm, -1, rev_n);
    double un_dot = v_dot<N>(rev_u, nrm);
    double theta_in;
    if (un_dot < 0) {
        index = 1 / index;
        theta_in = v_angle<N>(rev_u, rev_n);
    } else {
        theta_in = v_angle<N>(rev_u, nrm);
    }
    double theta_out;
    double sin_out = nd_sin(theta_in) / index;
    if (sin_out <= 1.0)
        theta_out = nd_asin(sin_out);
    else
        theta_out = NDT_PI - theta_in;
    v_unitize<N>(rev_n);
    v_unitize<N>(nrm);
    v_proj_unit<N>(u, rev_n, un);
    v_sub<N>(u, un, np);
    v_unitize<N>(np);
    double rn = nd_cos(theta_out);
    double rp = nd_sin(theta_out);
    if (un_dot < 0)
        v_scale<N>(nrm, rn, ref_n);
    else
        v_scale<N>(rev_n, rn, ref_n);
    v_scale<N>(np, rp, ref_p);
    v_add<N>(ref_n, ref_p, res);
}
// a[i] for a runtime i, as a chain of v_cndmask on register values.  The empty asm makes each
// element an opaque register value first: without it LLVM folds the select chain back into a
// dynamically indexed load and the whole array (and everything else that was an alloca with
// it) moves to scratch -- two L2 round trips per kd-tree step.
#ifdef NDT_PICK_TREE
// (experiment: the same selection as a tree of depth log2 N -- as many selects, a shorter dependent chain)
template <int N> NDT_DEV double v_pick(const double (&a)[N], int i)
{
    double e[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        e[k] = a[k];
        asm volatile("" : "+v"(e[k]));
    }
    int n = N;
#pragma unroll
    for (int bit = 0; bit < 4; ++bit) {
        if (n > 1) {
            const bool odd = ((i >> bit) & 1) != 0;
#pragma unroll
            for (int k = 0; k < N / 2 + 1; ++k) {
                if (2 * k + 1 < n) e[k] = odd ? e[2 * k + 1] : e[2 * k];
                else if (2 * k < n) e[k] = e[2 * k];
            }
            n = (n + 1) / 2;
        }
    }
    return e[0];
}
#else
template <int N> NDT_DEV double v_pick(const double (&a)[N], int i)
{
#ifdef NDT_PICK_COPY
    double e0 = a[0];
    asm volatile("" : "+v"(e0));
    double r = e0;
#pragma unroll
    for (int k = 1; k < N; ++k) {
        double e = a[k];
        asm volatile("" : "+v"(e));
        r = (i == k) ? e : r;
    }
    return r;
#else
    // (the RESULT of every select is made opaque, not its operands: an opaque operand is a copy of a[k] -- a[k] stays live -- and
    // a pick cost N v_mov_b64 on top of its selects)
    double r = a[0];
    asm volatile("" : "+v"(r));
#pragma unroll
    for (int k = 1; k < N; ++k) {
        r = (i == k) ? a[k] : r;
        asm volatile("" : "+v"(r));
    }
    return r;
#endif
}
#endif

NDT_DEV double lane_get(double x, int l)        // x of lane l (l: the same in every lane)
{
    const int ls = __builtin_amdgcn_readfirstlane(l);
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), ls), hi = __builtin_amdgcn_readlane(__double2hiint(x), ls);
    return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------ scene blob access

// The blob pointer is either LDS or global; all reads go through these so that the
// compiler sees one base pointer + index and can pick ds_read / global_load.
NDT_DEV int blob_int(const double *blob, int word, int half)
{
    return reinterpret_cast<const int *>(blob)[2 * word + half];
}
// one object reference {int object, int header flags} in a single 8-byte read
NDT_DEV void blob_ref(const double *blob, int word, int &id, int &flags)
{
    const long long bits = __double_as_longlong(blob[word]);
    id = (int)(bits & 0xffffffffll);
    flags = (int)(bits >> 32);
}
// a kd node's two words in one 16-byte read (off_kd is 16-byte aligned)
typedef double ndt_v2d __attribute__((ext_vector_type(2)));
NDT_DEV ndt_v2d blob_pair(const double *blob, int word) { return *reinterpret_cast<const ndt_v2d *>(blob + word); }
template <int N> NDT_DEV void blob_vec(const double *blob, int word, double (&r)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = blob[word + i];
}

// ------------------------------------------------------------------ object plugins
//
// isect<N, FULL>: one primitive (everything except hcube).  Returns hit/no-hit; `res` is
// always the hit point when it returns true; `normal` is only produced when FULL.

template <int N> NDT_DEV bool within_axes(const double *blob, int p0, int rec, int m, const double (&point)[N])
{
    // hcylinder.c:101-130 / orthotope.c:122-148.  Axis record i at rec + i*(N+3): axis[N], length, AdA, BdA
    double pos[N], Bc[N];
    blob_vec<N>(blob, p0, pos);
    v_sub<N>(point, pos, Bc);
    for (int i = 0; i < m; ++i) {
        int a = rec + i * (N + 3);
        double ax[N];
        blob_vec<N>(blob, a, ax);
        double scale = v_dot<N>(Bc, ax);
        scale = scale / blob[a + N + 1];
        if (scale < -NDT_EPS || scale > blob[a + N] + NDT_EPS) return false;
    }
    return true;
}

template <int N>
NDT_DEV void axes_quadratic(const double *blob, int p0, int rec, int m, const double (&o)[N], const double (&v)[N],
                            double &qa, double &qb, double &qc)
{
    // hcylinder.c:159-185 / orthotope.c:175-199
    // (one walk over the axes feeding both sums reads every axis record once instead of twice, but keeps two N-vector sums
    // alive: measured in round 3, it cost the 8-D trace kernel 16 registers it does not have; the records now come from LDS
    // in the coherent leaf scan, where a second read is cheap)
    double sA[N], sum_A[N], P[N], Q[N], pos[N];
    v_zero<N>(sum_A);
    for (int i = 0; i < m; ++i) {
        int a = rec + i * (N + 3);
        double ax[N];
        blob_vec<N>(blob, a, ax);
        double AdA = blob[a + N + 1];
        double VdA = v_dot<N>(v, ax);
        v_scale<N>(ax, VdA / AdA, sA);
        v_add<N>(sum_A, sA, sum_A);
    }
    v_sub<N>(sum_A, v, P);
    v_zero<N>(sum_A);
    for (int i = 0; i < m; ++i) {
        int a = rec + i * (N + 3);
        double ax[N];
        blob_vec<N>(blob, a, ax);
        double AdA = blob[a + N + 1];
        double BdA = blob[a + N + 2];
        double OdA = v_dot<N>(o, ax);
        v_scale<N>(ax, (OdA - BdA) / AdA, sA);
        v_add<N>(sum_A, sA, sum_A);
    }
    blob_vec<N>(blob, p0, pos);
    v_sub<N>(pos, o, Q);
    v_add<N>(Q, sum_A, Q);
    qa = v_dot<N>(P, P);
    qb = v_dot<N>(P, Q);
    qb *= 2;
    qc = v_dot<N>(Q, Q);
}

template <int N>
NDT_DEV void axes_normal(const double *blob, int p0, int rec, int m, const double (&res)[N], double (&normal)[N])
{
    // hcylinder.c:222-237 / orthotope.c:280-295
    double P[N], Q[N], sA[N], pos[N];
    blob_vec<N>(blob, p0, pos);
    v_sub<N>(res, pos, P);
    v_zero<N>(Q);
    for (int i = 0; i < m; ++i) {
        double ax[N];
        blob_vec<N>(blob, rec + i * (N + 3), ax);
        v_proj<N>(P, ax, sA);
        v_add<N>(Q, sA, Q);
    }
    v_sub<N>(P, Q, normal);
}

template <int N, bool FULL>
NDT_DEV bool isect(const double *blob, const SceneDesc &sd, int prim, const double (&o)[N], const double (&v)[N],
                   double (&res)[N], double (&normal)[N])
{
    const int h = sd.off_hdr + 2 * prim;
    const int flags = blob_int(blob, h, 0);
    const int type = flags & NDT_F_TYPE_MASK;
    const int p = sd.off_params + blob_int(blob, h, 1);
    switch (type) {
    case T_SPHERE: {
        // sphere.c:57-112.  params: center[N], r^2
        double c[N], oc[N];
        blob_vec<N>(blob, p, c);
        v_sub<N>(o, c, oc);
        double oc_len2 = v_dot<N>(oc, oc);
        double voc = v_dot<N>(v, oc);
        double desc = (voc * voc) - oc_len2 + blob[p + N];
        if (desc < 0.0) return false;
        double desc_root = sqrt(desc);
        double d = -(voc + desc_root);
        if (d < NDT_EPS) {
            d = desc_root - voc;
            if (d < NDT_EPS) return false;
        }
        v_scale<N>(v, d, res);
        v_add<N>(o, res, res);
        if (FULL) v_sub<N>(res, c, normal);
        return true;
    }
    case T_HPLANE:
    case T_HDISK: {
        // hplane.c:39-75, hdisk.c:61-85.  params: pos[N], dir[N], radius
        double pos[N], nrm[N], pl[N];
        blob_vec<N>(blob, p, pos);
        blob_vec<N>(blob, p + N, nrm);
        double d = -1;
        v_sub<N>(pos, o, pl);
        double pln = v_dot<N>(pl, nrm);
        double ln = v_dot<N>(v, nrm);
        if (ln > NDT_EPS || ln < -NDT_EPS) d = pln / ln;
        if (d < NDT_EPS) return false;
        v_scale<N>(v, d, pl);
        v_add<N>(o, pl, res);
        if (type == T_HDISK) {
            double dist = v_dist<N>(res, pos);
            if (dist > blob[p + 2 * N] || dist < 0) return false;
        }
        if (FULL) v_copy<N>(normal, nrm);
        return true;
    }
    case T_CYLINDER: {
        // cylinder.c:104-210.  params: pos0[N], axis[N], length, AdA, BdA, radius
        double Be[N], A[N], sA[N], X[N], Y[N], tmp[N];
        blob_vec<N>(blob, p, Be);
        blob_vec<N>(blob, p + N, A);
        const double length = blob[p + 2 * N], AdA = blob[p + 2 * N + 1], BdA = blob[p + 2 * N + 2];
        const double size0 = blob[p + 2 * N + 3];
        double VdA = v_dot<N>(v, A);
        double OdA = v_dot<N>(o, A);
        double Vaaa = VdA / AdA;
        double BOaa = (BdA - OdA) / AdA;
        v_scale<N>(A, Vaaa, sA);
        v_sub<N>(v, sA, Y);
        v_sub<N>(o, Be, tmp);
        v_scale<N>(A, BOaa, sA);
        v_add<N>(tmp, sA, X);
        double qa = v_dot<N>(Y, Y);
        double qb = v_dot<N>(Y, X);
        qb *= 2;
        double qc = v_dot<N>(X, X);
        qc -= size0 * size0;
        double det = qb * qb - 4 * qa * qc;
        if (det <= 0) return false;
        double detRoot = sqrt(det);
        double t1 = (-qb + detRoot) / (2 * qa);
        double t2 = (-qb - detRoot) / (2 * qa);
        bool ret = false;
        const bool inf_ends = (flags & NDT_F_INF_ENDS) != 0;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            double t = pass == 0 ? t2 : t1;
            if (!ret && t > NDT_EPS) {
                v_scale<N>(v, t, sA);
                v_add<N>(o, sA, res);
                if (inf_ends) {
                    ret = true;
                } else {
                    // between_ends, cylinder.c:88-102
                    double Bc[N];
                    v_sub<N>(res, Be, Bc);
                    double scale = v_dot<N>(Bc, A);
                    if (scale > 0 && scale < length) ret = true;
                }
            }
        }
        if (ret && FULL) {
            v_sub<N>(res, Be, X);
            double nCdA = v_dot<N>(A, X);
            v_scale<N>(A, nCdA / AdA, Y);
            v_sub<N>(X, Y, normal);
        }
        return ret;
    }
    case T_HCYLINDER: {
        // hcylinder.c:132-244.  params: pos0[N], radius, then m axis records
        const int m = blob_int(blob, h + 1, 1);
        const int rec = p + N + 1;
        double qa, qb, qc;
        axes_quadratic<N>(blob, p, rec, m, o, v, qa, qb, qc);
        const double radius = blob[p + N];
        qc -= radius * radius;
        double det = qb * qb - 4 * qa * qc;
        if (det < 0.0) return false;
        double detRoot = sqrt(det);
        double t1 = (-qb + detRoot) / (2 * qa);
        double t2 = (-qb - detRoot) / (2 * qa);
        bool ret = false;
        const bool inf_ends = (flags & NDT_F_INF_ENDS) != 0;
        double sA[N];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            double t = pass == 0 ? t2 : t1;
            if (!ret && t > NDT_EPS) {
                v_scale<N>(v, t, sA);
                v_add<N>(o, sA, res);
                if (inf_ends || within_axes<N>(blob, p, rec, m, res)) ret = true;
            }
        }
        if (ret && FULL) axes_normal<N>(blob, p, rec, m, res, normal);
        return ret;
    }
    case T_ORTHOTOPE: {
        // orthotope.c:150-302.  params: pos0[N], pad, then m basis records
        const int m = blob_int(blob, h + 1, 1);
        const int rec = p + N + 1;
        double qa, qb, qc;
        axes_quadratic<N>(blob, p, rec, m, o, v, qa, qb, qc);
        qc -= NDT_EPS;
        double det = qb * qb - 4 * qa * qc;
        bool ret = false;
        double sA[N];
        if (det >= 0.0 && fabs(qa) > NDT_EPS) {
            double detRoot = sqrt(det);
            double half_inv_qa = 0.5 / qa;
            double t1 = (-qb + detRoot) * half_inv_qa;
            double t2 = (-qb - detRoot) * half_inv_qa;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                double t = pass == 0 ? t2 : t1;
                if (!ret && t > NDT_EPS) {
                    v_scale<N>(v, t, sA);
                    v_add<N>(o, sA, res);
                    if (within_axes<N>(blob, p, rec, m, res)) ret = true;
                }
            }
        }
        if (!ret) {
            double t = -1.0;
            if (fabs(qa) < NDT_EPS) {
                if (fabs(qb) < NDT_EPS)   // sic: orthotope.c:238-241
                    t = -qc / qb;
                else
                    t = -1.0;
            } else {
                t = -qb / (2 * qa);
            }
            if (t < NDT_EPS) return false;
            double dist = qa * t * t + qb * t + qc;
            if (fabs(dist) > NDT_EPS) return false;
            v_scale<N>(v, t, sA);
            v_add<N>(o, sA, res);
            if (within_axes<N>(blob, p, rec, m, res)) ret = true;
        }
        if (ret && FULL) axes_normal<N>(blob, p, rec, m, res, normal);
        return ret;
    }
    case T_HFACET: {
        // hfacet.c:211-310.  params: v0[N], unit_edge0[N], edge_perp[N], x2,y2,x3,y3, dir[3][N]
        double v0[N], ue0[N], perp[N], R[N], vE0[N], vE2[N], Q[N], oP0[N];
        blob_vec<N>(blob, p, v0);
        blob_vec<N>(blob, p + N, ue0);
        blob_vec<N>(blob, p + 2 * N, perp);
        v_proj_unit<N>(v, ue0, vE0);
        v_proj_unit<N>(v, perp, vE2);
        v_add<N>(vE0, vE2, R);
        v_sub<N>(R, v, R);
        double Rv = v_sum<N>(R);
        if (fabs(Rv) < NDT_EPS) return false;
        v_sub<N>(o, v0, oP0);
        v_proj_unit<N>(oP0, ue0, vE0);
        v_proj_unit<N>(oP0, perp, vE2);
        v_add<N>(vE0, vE2, Q);
        v_sub<N>(Q, oP0, Q);
        double Qv = v_sum<N>(Q);
        double t = -Qv / Rv;
        if (!(t > NDT_EPS)) return false;
        v_scale<N>(v, t, res);
        v_add<N>(o, res, res);
        // get_barycentric, hfacet.c:156-199
        double C[N];
        v_sub<N>(res, v0, C);
        const double x1 = 0, y1 = 0;
        double xp = v_dot<N>(ue0, C);
        double yp = v_dot<N>(perp, C);
        const double x2 = blob[p + 3 * N], y2 = blob[p + 3 * N + 1], x3 = blob[p + 3 * N + 2], y3 = blob[p + 3 * N + 3];
        double l1 = ((y2 - y3) * (xp - x3) + (x3 - x2) * (yp - y3)) / ((y2 - y3) * (x1 - x3) + (x3 - x2) * (y1 - y3));
        double l2 = ((y3 - y1) * (xp - x3) + (x1 - x3) * (yp - y3)) / ((y2 - y3) * (x1 - x3) + (x3 - x2) * (y1 - y3));
        double l3 = 1 - l1 - l2;
        if (l1 < -NDT_EPS || l1 > 1 + NDT_EPS) return false;
        if (l2 < -NDT_EPS || l2 > 1 + NDT_EPS) return false;
        if (l3 < -NDT_EPS || l3 > 1 + NDT_EPS) return false;
        if (FULL) {
            if (flags & NDT_F_USE_NORMALS) {
                double nd[N];
                v_zero<N>(normal);
                blob_vec<N>(blob, p + 3 * N + 4, nd);
                v_scale<N>(nd, l1, R);
                v_add<N>(normal, R, normal);
                blob_vec<N>(blob, p + 4 * N + 4, nd);
                v_scale<N>(nd, l2, R);
                v_add<N>(normal, R, normal);
                blob_vec<N>(blob, p + 5 * N + 4, nd);
                v_scale<N>(nd, l3, R);
                v_add<N>(normal, R, normal);
            } else {
                // hfacet_point_in_plane, hfacet.c:119-144
                double D[N], U[N], V[N];
                v_sub<N>(o, v0, D);
                v_proj_unit<N>(D, ue0, U);
                v_proj_unit<N>(D, perp, V);
                v_add<N>(U, V, R);
                v_add<N>(R, v0, R);
                v_sub<N>(o, R, normal);
                v_unitize<N>(normal);
            }
        }
        return true;
    }
    case T_FACET: {
        // facet.c:166-269.  params: pos[3][N], basis[2][N], AdA[2], BdA[2], angle[3], dir0[N]
        double pos1[N], P[N], sA[N], sum_A[N], Q[N];
        blob_vec<N>(blob, p + N, pos1);
        v_zero<N>(sum_A);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            double b[N];
            blob_vec<N>(blob, p + 3 * N + i * N, b);
            double VdA = v_dot<N>(v, b);
            double AdA = blob[p + 5 * N + i];
            v_scale<N>(b, VdA / AdA, sA);
            v_add<N>(sum_A, sA, sum_A);
        }
        v_sub<N>(sum_A, v, P);
        v_zero<N>(sum_A);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            double b[N];
            blob_vec<N>(blob, p + 3 * N + i * N, b);
            double OdA = v_dot<N>(o, b);
            double BdA = blob[p + 5 * N + 2 + i];
            double AdA = blob[p + 5 * N + i];
            v_scale<N>(b, (OdA - BdA) / AdA, sA);
            v_add<N>(sum_A, sA, sum_A);
        }
        v_sub<N>(pos1, o, Q);
        v_add<N>(Q, sum_A, Q);
        double qa = v_dot<N>(P, P);
        double qb = v_dot<N>(P, Q);
        qb *= 2;
        double qc = v_dot<N>(Q, Q);
        double t = -1.0;
        if (fabs(qa) < NDT_EPS) {
            if (fabs(qb) < NDT_EPS)
                t = -qc / qb;
            else
                t = -1.0;
        } else {
            t = -qb / (2 * qa);
        }
        if (t < NDT_EPS) return false;
        double dist = qa * t * t + qb * t + qc;
        if (fabs(dist) > NDT_EPS) return false;
        v_scale<N>(v, t, sA);
        v_add<N>(o, sA, res);
        // inside_edges, facet.c:148-164
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = (i + 1) % 3;
            double pi[N], pj[N], a[N], b[N];
            blob_vec<N>(blob, p + i * N, pi);
            blob_vec<N>(blob, p + j * N, pj);
            v_sub<N>(res, pi, a);
            v_sub<N>(pj, pi, b);
            double angle = v_angle<N>(a, b);
            if (angle > blob[p + 5 * N + 4 + i]) return false;
        }
        if (FULL) blob_vec<N>(blob, p + 5 * N + 7, normal);
        return true;
    }
    default:
        return false;
    }
}

// vect_bounding_sphere_intersect, bounding.c:34-85, on a sphere record already read: centre, radius, radius^2
template <int N>
NDT_DEV bool bsphere_gate_rec(const double (&c)[N], const double rad, const double rad2, const double (&o)[N], const double (&v)[N],
                              double min_dist)
{
    double oc[N];
    v_sub<N>(o, c, oc);
    double oc_len2 = v_dot<N>(oc, oc);
    // (one verdict from all the comparisons, & and | instead of early returns: a return in the middle is a divergent branch,
    // and what it skips -- one dot product -- costs less than the branch)
    const double min_dist_r = min_dist + rad;
    const bool too_far = (min_dist > 0) & (oc_len2 > min_dist_r * min_dist_r);
    double voc = v_dot<N>(v, oc);
    double voc2 = voc * voc;
    double desc = voc2 - oc_len2 + rad2;
    return !(too_far | (desc < 0.0) | ((voc > 0.0) & (voc2 > desc)));
}
template <int N>
NDT_DEV bool bsphere_gate(const double *blob, const SceneDesc &sd, int obj, const double (&o)[N], const double (&v)[N],
                          double min_dist)
{
    const int b = sd.off_bs + obj * (N + 2);
    double c[N];
    blob_vec<N>(blob, b, c);
    return bsphere_gate_rec<N>(c, blob[b + N], blob[b + N + 1], o, v, min_dist);
}

// Ray (t >= 0) against the hull box of an hcube: N slabs { axis[N], centre, half extent }.
// A miss proves that no face of the hcube can be hit (see ndt_blob.hip:hcube_hull_box for the
// margin argument), so the face scan is skipped; a pass decides nothing.
// The hull box test plus, for hcubes that carry them (NDT_F_FACEBOX), the same test against every face's own
// box: returns 0 when the ray misses the hull box or every face box (skip the hcube), -1 when all faces are to be
// scanned (no face boxes), otherwise the mask of the faces whose box the ray meets (bit f = nested primitive f,
// 63 of them to a call: `chunk`, below).  The projections u_k.o and 1/(u_k.v) of the hull test are shared by all the faces, so a
// face costs N slab updates -- against ~20 orthotope intersections per hcube visit on the benchmark scene, of which
// almost all missed.
// Cost, as measured on the benchmark frame (profiles/r03_hull_probe.txt): 41 % of its rays pass the bounding-sphere gate of an
// hcube somewhere, 1 % meet a hull box, 0.002 % a face box -- and because ONE lane in this code keeps its wavefront waiting, almost
// every scan step of every wavefront paid for the hull test of two or three lanes.  Hence:
//   * the slabs come thinnest first (ndt_blob.hip sorts them) and after every slab from the second on the wavefront asks whether
//     any of its rays is still inside: two thin slabs usually settle it;
//   * 1/(u_k.v) is the hardware's reciprocal with one Newton step, not an IEEE division (a third of the old test's instructions
//     were its N divisions): the test is a filter, not part of the reference's arithmetic, and it stays conservative because every
//     comparison of interval ends gives way by 2^-30 of their size (NDT_SLAB_GIVE) -- a million times the reciprocal's error;
//   * the face boxes give up the same way, two slabs at a time.
#define NDT_SLAB_GIVE 9.313225746154785e-10         /* 2^-30 */
NDT_DEV double slab_rcp(double d)
{
    const double r = __builtin_amdgcn_rcp(d);
    return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}
// false only when the interval [lo, hi] is empty by more than the reciprocal's error can account for (NaN: not empty)
NDT_DEV bool slab_interval_holds(double lo, double hi)
{
    return !(__builtin_fma(-fabs(lo), NDT_SLAB_GIVE, lo) > __builtin_fma(fabs(hi), NDT_SLAB_GIVE, hi));
}
// (more than 63 faces: the masks come 63 faces at a time -- `chunk` is where to start looking and, on return, the chunk of the
// mask that came back: the first one from there on in which the ray meets a face's box)
#define NDT_HULL_CHUNK 63
// (from 9-D on: in the 6-D .. 8-D trace kernels of the global-memory tier the lookup's mere presence cost 25 spilled registers and
// 1-3 % of the hypercube frames, which have no hcube at all; a 6-D hcube's 472 faces gain 4 % from it, a 9-D one's 16 866 a third.
// As a real function it was no better: the registers live across the call are saved around it -- 110 spilled either way.)
#ifndef NDT_GROUPS_MIN_DIMS
#define NDT_GROUPS_MIN_DIMS 9
#endif
#define NDT_GROUPS_MAX_AXES 6      /* hull_faces: a ray that passes the slivers of more axes than this takes the hierarchy */
// Hcubes of more than 63 faces carry a hierarchy over their face boxes (NDT_F_FACETREE; round 4): level j holds, for every
// aligned run [k 2^j, (k + 1) 2^j) of faces, the box of the union of their boxes.  The faces of an hcube come in groups of
// 2^(N - m) consecutive ones that share their m directions and differ in the corner they hang on, low dimensions first
// (hcube.c:33-153) -- so a run of 2^j of them is thin in the N - m - j dimensions its faces agree on, and a ray that is not within
// the margin of that many planes at once misses the run as a whole.  The walk below visits the runs in face order (an in-order
// descent of an implicit binary tree over the face numbers, no stack: at face i the largest aligned run that starts there is
// tried, then -- the ray meeting it -- its halves), so the faces it reports come out ascending, as the scan needs them.  A ray
// that meets the hull of the 10-D zoo's hcube used to test all 52 904 face boxes; now a few thousand runs.
template <int N>
NDT_DEV bool hull_box_meets(const double *blob, int q, const double (&po)[N], const double (&inv)[N])
{
    if (blob[q + 1] < 0.0) return false;            // (a run none of whose faces can be hit: half extent -1)
    double f0 = 0.0, f1 = NDT_DBL_MAX;
    bool fok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (fok) {
            const double a = po[k] - blob[q + 2 * k];
            const double h = blob[q + 2 * k + 1];
            if (inv[k] == 0.0) {
                if (fabs(a) > h) fok = false;
            } else {
                const double ta = (-h - a) * inv[k], tb = (h - a) * inv[k];
                const double lo = ta < tb ? ta : tb, hi = ta < tb ? tb : ta;
                if (lo > f0) f0 = lo;
                if (hi < f1) f1 = hi;
            }
            if (k & 1) fok = fok && slab_interval_holds(f0, f1);
        }
    }
    return fok && slab_interval_holds(f0, f1);
}

template <int N>
NDT_DEV long long hull_faces(const double *blob, int p, bool face_boxes, int nf, const double (&o)[N], const double (&v)[N], int &chunk,
                             const bool face_tree = false, const bool face_groups = false)
{
    double po[N], inv[N];       // u_k.o and 1/(u_k.v); 0 marks a ray parallel to slab k
    double t0 = 0.0, t1 = NDT_DBL_MAX;
    bool ok = true;
    bool any = true;            // wave-uniform: some ray of the wavefront is still inside the slabs so far
    // (no early exit from the loops: they would not unroll, and po / inv would be indexed in scratch)
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (any) {
            double u[N];
            blob_vec<N>(blob, p + k * (N + 2), u);
            po[k] = v_dot<N>(u, o);
            const double a = po[k] - blob[p + k * (N + 2) + N];
            const double d = v_dot<N>(u, v);
            const double h = blob[p + k * (N + 2) + N + 1];
            if (fabs(d) < 1e-200) {
                inv[k] = 0.0;
                if (fabs(a) > h) ok = false;        // parallel to the slab and outside it
            } else {
                inv[k] = slab_rcp(d);
                const double ta = (-h - a) * inv[k], tb = (h - a) * inv[k];
                const double lo = ta < tb ? ta : tb, hi = ta < tb ? tb : ta;
                if (lo > t0) t0 = lo;
                if (hi < t1) t1 = hi;
            }
            if (k >= 1) {
                ok = ok && slab_interval_holds(t0, t1);
                any = __ballot(ok) != 0ull;
            }
        }
    }
    if (!any || !ok) return 0;
    if (!face_boxes) return -1;
    const int fr = p + N * (N + 2);
    const int n_chunks = (nf + NDT_HULL_CHUNK - 1) / NDT_HULL_CHUNK;
    // (an hcube has more than 63 faces from 5-D on: the 3-D and 4-D kernels -- the benchmark scenes' -- do not carry the walk; in
    // their trace kernel its mere presence cost 2 %)
    const int grp = fr + n_chunks + nf * 2 * N;                 // clusters and table of the face groups, when there are any
    if (N >= NDT_GROUPS_MIN_DIMS && face_groups) {
        // The faces by the hull axes their boxes are thin on (ndt_blob.hip:hcube_face_groups).  On which axes does the ray, while
        // it is inside the hull, pass one of the two clusters of slivers at all?  Usually one or two -- where it enters and
        // leaves the cube.  Only faces pinned on a subset of those axes can be met: a handful of index ranges, each face of
        // them against its own box.  (A ray along a diagonal passes many clusters: beyond NDT_GROUPS_MAX_AXES the walks below.)
        unsigned int th = 0u;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            bool meets = false;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const double a = po[k] - blob[grp + 4 * k + 2 * side];
                const double h = blob[grp + 4 * k + 2 * side + 1];
                bool m;
                if (inv[k] == 0.0) {
                    m = !(fabs(a) > h);
                } else {
                    const double ta = (-h - a) * inv[k], tb = (h - a) * inv[k];
                    double lo = ta < tb ? ta : tb, hi = ta < tb ? tb : ta;
                    if (t0 > lo) lo = t0;
                    if (t1 < hi) hi = t1;
                    m = slab_interval_holds(lo, hi);
                }
                meets |= m & (h >= 0.0);
            }
            th |= meets ? (1u << k) : 0u;
        }
        if (__popc(th) <= NDT_GROUPS_MAX_AXES) {
            const int f_min = chunk * NDT_HULL_CHUNK;
            int found = -1;
            unsigned long long live = 0ull;
            unsigned int sub = th;
            const int mem = grp + 4 * N + (1 << N);             // the faces of every axis set, ascending, two to a word
            while (true) {
                const int start = blob_int(blob, grp + 4 * N + (int)sub, 0), count = blob_int(blob, grp + 4 * N + (int)sub, 1);
                for (int i = start; i < start + count; ++i) {
                    const int f = blob_int(blob, mem + (i >> 1), i & 1);
                    if (f < f_min) continue;
                    if (found >= 0 && f >= (found + 1) * NDT_HULL_CHUNK) break;        // (ascending: nothing of this set matters any more)
                    if (hull_box_meets<N>(blob, fr + n_chunks + f * 2 * N, po, inv) &&
                        (((unsigned long long)__double_as_longlong(blob[fr + f / NDT_HULL_CHUNK]) >> (f % NDT_HULL_CHUNK)) & 1ull) != 0ull) {
                        const int ch = f / NDT_HULL_CHUNK;
                        if (found < 0 || ch < found) {
                            found = ch;
                            live = 0ull;
                        }
                        live |= 1ull << (f - found * NDT_HULL_CHUNK);
                    }
                }
                if (sub == 0u) break;
                sub = (sub - 1u) & th;
            }
            if (found < 0) return 0;
            chunk = found;
            return (long long)live;
        }
    }
    if (N >= 5 && face_tree) {
        // the faces whose box the ray meets, from face 63 `chunk` on, in face order: the first one names the chunk that is
        // reported, the walk goes on to that chunk's end
        const int tree = grp + (face_groups ? 4 * N + (1 << N) + ((nf + 1) >> 1) : 0);    // { levels, offset of level 1 .. } then the levels' rows
        const int top = blob_int(blob, tree, 0);                // the highest level (2^top >= nf)
        int i = chunk * NDT_HULL_CHUNK;
        int found = -1;
        unsigned long long live = 0ull;
        while (i < nf && (found < 0 || i < (found + 1) * NDT_HULL_CHUNK)) {
            int j = (i == 0) ? top : (__ffs(i) - 1);
            if (j > top) j = top;
            bool moved = false;
            while (!moved) {
                const int q = (j == 0) ? fr + n_chunks + i * 2 * N : tree + blob_int(blob, tree + 1 + (j >> 1), j & 1) + (i >> j) * 2 * N;
                if (!hull_box_meets<N>(blob, q, po, inv)) {
                    i += 1 << j;                                // the whole run is missed
                    moved = true;
                } else if (j == 0) {
                    const bool possible = (((unsigned long long)__double_as_longlong(blob[fr + i / NDT_HULL_CHUNK]) >> (i % NDT_HULL_CHUNK)) & 1ull) != 0ull;
                    if (possible) {
                        if (found < 0) found = i / NDT_HULL_CHUNK;
                        if (i / NDT_HULL_CHUNK == found) live |= 1ull << (i - found * NDT_HULL_CHUNK);
                    }
                    i += 1;
                    moved = true;
                } else {
                    --j;                                        // its first half starts at the same face
                }
            }
        }
        if (found < 0) return 0;
        chunk = found;
        return (long long)live;
    }
    for (; chunk < n_chunks; ++chunk) {
        const int f_begin = chunk * NDT_HULL_CHUNK, f_end = (f_begin + NDT_HULL_CHUNK < nf) ? f_begin + NDT_HULL_CHUNK : nf;
        unsigned long long live = 0ull;
        for (int f = f_begin; f < f_end; ++f) {
            const int q = fr + n_chunks + f * 2 * N;
            double f0 = 0.0, f1 = NDT_DBL_MAX;
            bool fok = true;
            bool fany = true;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if (fany) {
                    const double a = po[k] - blob[q + 2 * k];
                    const double h = blob[q + 2 * k + 1];
                    if (inv[k] == 0.0) {
                        if (fabs(a) > h) fok = false;
                    } else {
                        const double ta = (-h - a) * inv[k], tb = (h - a) * inv[k];
                        const double lo = ta < tb ? ta : tb, hi = ta < tb ? tb : ta;
                        if (lo > f0) f0 = lo;
                        if (hi < f1) f1 = hi;
                    }
                    if ((k & 1) && k + 1 < N) {
                        fok = fok && slab_interval_holds(f0, f1);
                        fany = __ballot(fok) != 0ull;
                    }
                }
            }
            if (fany && fok && slab_interval_holds(f0, f1)) live |= 1ull << (f - f_begin);
        }
        live &= (unsigned long long)__double_as_longlong(blob[fr + chunk]);
        // (the lanes of a wavefront may be on different hcubes or chunks: every lane leaves with ITS first non-empty chunk)
        if (live != 0ull) return (long long)live;
    }
    return 0;
}

// ------------------------------------------------------------------ item boxes
//
// The ray in the scene's box frame (ndt_blob.hip:scene_item_boxes): u_k.o and 1/(u_k.v) for the N axes, computed once per
// ray -- the first time it meets a boxed item -- and kept in the lane's slot of the wavefront's LDS area, pair k at
// [k * 128 + 2 * lane]: sixteen bytes a lane, one ds_read_b128 per slab.  (In registers they would be 2N more of them; the
// 8-D trace kernel has none to spare.)
template <int N> NDT_DEV void ray_in_box_frame(const double *blob, const SceneDesc &sd, const double (&o)[N], const double (&v)[N], double *slot)
{
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double u[N];
        blob_vec<N>(blob, sd.off_oframe + k * N, u);
        const double d = v_dot<N>(u, v);
        ndt_v2d pr;
        pr.x = v_dot<N>(u, o);
        pr.y = (fabs(d) < 1e-200) ? 0.0 : 1.0 / d;          // 0 marks a ray parallel to slab k
        *reinterpret_cast<ndt_v2d *>(slot + k * 128) = pr;
    }
}
// Ray (t >= 0) against the box of item `id`: false = the ray misses it, so intersect() cannot accept a point of this item
// (same slab arithmetic as hull_faces, same margin argument); true decides nothing.
// The slabs come thinnest first, and every second one the wavefront asks whether any of its rays is still inside: a miss is
// usually known after two or three.
template <int N> NDT_DEV bool item_box_meets(const double *blob, const SceneDesc &sd, int id, const double *slot)
{
    const int q = sd.off_obox + id * 2 * N;
    const unsigned long long ord = (unsigned long long)__double_as_longlong(blob[sd.off_oord + id]);
    double t0 = 0.0, t1 = NDT_DBL_MAX;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k >= 2 && (k & 1) == 0 && __ballot(ok && t0 <= t1) == 0ull) break;
        const int axis = (int)((ord >> (4 * k)) & 15ull);
        const ndt_v2d pr = *reinterpret_cast<const ndt_v2d *>(slot + axis * 128);
        const ndt_v2d box = blob_pair(blob, q + 2 * k);
        const double a = pr.x - box.x, h = box.y;
        if (pr.y == 0.0) {
            if (fabs(a) > h) ok = false;
        } else {
            const double ta = (-h - a) * pr.y, tb = (h - a) * pr.y;
            const double lo = ta < tb ? ta : tb, hi = ta < tb ? tb : ta;
            if (lo > t0) t0 = lo;
            if (hi < t1) t1 = hi;
        }
    }
    return ok && t0 <= t1;
}

// ------------------------------------------------------------------ trace / kd-tree

// Per-ray visit mask (the reference callocs obj_num bytes per ray, kd-tree.c:600).
// MW > 0: MW 64-bit words in registers, indexed by unrolled selects.
// MW == 0 (large scenes: the 6-D .. 8-D hypercubes, 728 .. 6560 items), two representations of the same set:
//   leaf history  What a ray has visited is "the items of the leaves it has scanned, up to where each scan ended".  Leaf
//                 lists ascend in item number (the reference's kd builder keeps the scene's object order in every leaf;
//                 checked at upload), so a scan that ended after item `last` visited exactly the leaf's items with a
//                 number <= last.  The ray keeps up to four {leaf, last + 1} pairs in registers; the blob carries every
//                 leaf's items as a bit set (read-only, a few hundred KB: cache resident); "was item x visited" = for
//                 some pair, bit x of the leaf's set and x < cut.  No stores, no per-ray memory: the scan of a ray's first
//                 leaf -- most of all scanning -- touches nothing but the scene.
//   slab          `ext` points at this lane's words in a global slab (one bit per item, as the reference's byte per
//                 item).  432 MB for the resident wavefronts of the 8-D scene: every test-and-set is a read-modify-write
//                 that misses every cache, two dependent memory round trips per scanned item -- what the trace kernel of
//                 this tier used to wait for.  Now the fallback: a ray that finishes a fifth leaf replays its history into
//                 the slab (`spill`) and goes on there; scenes whose lists do not ascend never leave it.
struct LeafSets {
    const unsigned long long *sets;     // [leaf][words]: the items of every kd leaf as a bit set (nullptr: no history)
    const double *blob;                 // for the replay: leaf ranges and lists
    int words, off_lrange, off_leaf;
    int cap;                            // history entries a ray may hold (1 .. 4; tests shrink it to force the replay)
};
template <int MW> struct VisitMask {
    unsigned long long w[MW > 0 ? MW : 1];
    unsigned long long *ext;
    int ext_stride;
    // slab, masks of up to 128 words (8192 items): which words of the slab this ray has written.  A word that has
    // not been written yet counts as zero, so a new ray costs two register moves instead of `words` stores to the slab
    // (103 words = 0.8 KB per ray on the 8-D hypercube: more than the whole algorithmic ray record).
    // leaf history: entry = leaf << 16 | cut (items of `leaf` with a number < cut are visited); hist_n < 0: the ray is on the slab.
    // (h0 .. h3 and the slab's live0 / live1 are never needed together: one set of registers)
    LeafSets ls;
    unsigned int h0, h1, h2, h3;
    int hist_n;
    int slab_words;
    NDT_DEV bool lazy() const { return slab_words <= 128; }
    NDT_DEV void clear(int words)
    {
        if (MW > 0) {
#pragma unroll
            for (int i = 0; i < MW; ++i) w[i] = 0ull;
        } else {
            slab_words = words;
            hist_n = 0;
            h0 = h1 = h2 = h3 = 0u;
            if (!ls.sets) begin_slab();
        }
    }
    NDT_DEV void begin_slab()
    {
        hist_n = -1;
        h0 = h1 = h2 = h3 = 0u;         // = live0, live1
        if (!lazy())
            for (int i = 0; i < slab_words; ++i) ext[(size_t)i * ext_stride] = 0ull;
    }
    NDT_DEV bool slab_test_and_set(int id)
    {
        const unsigned long long bit = 1ull << (id & 63);
        const int word = id >> 6;
        bool written = true;
        if (lazy()) {
            // live0 = {h1, h0}, live1 = {h3, h2}: one bit per slab word this ray has written
            const unsigned int wb = 1u << (word & 31);
            const int q = word >> 5;
            written = (((q == 0) ? h0 : (q == 1) ? h1 : (q == 2) ? h2 : h3) & wb) != 0u;
            if (q == 0) h0 |= wb;
            else if (q == 1) h1 |= wb;
            else if (q == 2) h2 |= wb;
            else h3 |= wb;
        }
        unsigned long long cur = 0ull;
        if (written) cur = ext[(size_t)word * ext_stride];
        if (cur & bit) return true;
        ext[(size_t)word * ext_stride] = cur | bit;
        return false;
    }
    NDT_DEV bool in_history(unsigned int h, int id) const
    {
        const unsigned long long set = ls.sets[(size_t)(h >> 16) * ls.words + (id >> 6)];
        return ((set >> (id & 63)) & 1ull) != 0ull && (unsigned int)id < (h & 0xffffu);
    }
    // returns true when `id` was already visited; marks it otherwise (history: the mark is the leaf's entry, made by
    // end_leaf when the scan is over -- a list names an item once, so nothing in between asks for it)
    NDT_DEV bool test_and_set(int id)
    {
        if (MW > 0) {
            const unsigned long long bit = 1ull << (id & 63);
            const int word = id >> 6;
            bool seen = false;
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                if (i == word) {
                    seen = (w[i] & bit) != 0ull;
                    w[i] |= bit;
                }
            }
            return seen;
        } else {
            if (hist_n < 0) return slab_test_and_set(id);
            bool seen = false;
            if (hist_n > 0) seen = in_history(h0, id);
            if (hist_n > 1) seen = seen || in_history(h1, id);
            if (hist_n > 2) seen = seen || in_history(h2, id);
            if (hist_n > 3) seen = seen || in_history(h3, id);
            return seen;
        }
    }
    // history only: bit j set = item ids[j] was visited.  Eight items at once: per history entry the eight set words are
    // fetched before any is looked at (one round trip, not eight).
    NDT_DEV unsigned int seen_of_8(const int (&ids)[8]) const
    {
        unsigned int seen = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e < hist_n) {
                const unsigned int h = (e == 0) ? h0 : (e == 1) ? h1 : (e == 2) ? h2 : h3;
                const unsigned long long *row = ls.sets + (size_t)(h >> 16) * ls.words;
                unsigned long long wd[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wd[j] = row[ids[j] >> 6];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (((wd[j] >> (ids[j] & 63)) & 1ull) != 0ull && (unsigned int)ids[j] < (h & 0xffffu)) seen |= 1u << j;
            }
        }
        return seen;
    }
    // the scan of kd leaf `leaf` is over; `last` = the last item it looked at (-1: none)
    NDT_DEV void end_leaf(int leaf, int last)
    {
        if (MW > 0 || hist_n < 0 || last < 0) return;
        if (hist_n >= ls.cap) spill();
        if (hist_n < 0) {
            spill_leaf((unsigned int)leaf << 16 | (unsigned int)(last + 1));
            return;
        }
        const unsigned int e = (unsigned int)leaf << 16 | (unsigned int)(last + 1);
        if (hist_n == 0) h0 = e;
        else if (hist_n == 1) h1 = e;
        else if (hist_n == 2) h2 = e;
        else h3 = e;
        ++hist_n;
    }
    NDT_DEV void spill_leaf(unsigned int h)
    {
        const int cut = (int)(h & 0xffffu);
        const int first = blob_int(ls.blob, ls.off_lrange + (int)(h >> 16), 0), num = blob_int(ls.blob, ls.off_lrange + (int)(h >> 16), 1);
        for (int j = 0; j < num; ++j) {
            const int id = blob_int(ls.blob, ls.off_leaf + first + j, 0);
            if (id >= cut) break;
            (void)slab_test_and_set(id);
        }
    }
    // the history is full: everything it stands for goes into the slab, and the ray stays there
    NDT_DEV void spill()
    {
        const int n = hist_n;
        const unsigned int e0 = h0, e1 = h1, e2 = h2, e3 = h3;
        begin_slab();
        if (n > 0) spill_leaf(e0);
        if (n > 1) spill_leaf(e1);
        if (n > 2) spill_leaf(e2);
        if (n > 3) spill_leaf(e3);
    }
};


#ifdef NDT_PHASE_TIMING
// diagnostic build only: wave-level cycle stamps per phase (never enabled in the shipped library)
#define NDT_STAMP(slot)                                                   \
    do {                                                                  \
        const unsigned long long now_ = __builtin_readcyclecounter();     \
        ph[slot] += now_ - ph_last;                                       \
        ph_last = now_;                                                   \
    } while (0)
#define NDT_COUNT(slot) (cnt[slot] += 1)
/* wave-level occupancy of a loop body: iterations and active lanes (same value in every active lane) */
#define NDT_OCC(slot) do { occ[2 * (slot)] += 1; occ[2 * (slot) + 1] += __popcll(__ballot(1)); } while (0)
#else
#define NDT_STAMP(slot) do { } while (0)
#define NDT_COUNT(slot) do { } while (0)
#define NDT_OCC(slot) do { } while (0)
#endif

// ------------------------------------------------------------------ coherent leaf scan (global-memory tier)
//
// In the 6-D .. 8-D scenes a leaf holds 60 .. 227 items and the 64 rays of a wavefront -- an 8x8 pixel tile, or 64 shadow
// rays towards one light -- nearly always scan the same leaf.  Item by item, trace()'s loop is a chain of dependent reads:
// list entry -> bounding sphere -> (gate passes) header -> parameter record; under load each is ~0.8 us, a launch lasts as
// long as its slowest batch (686 such steps), and the wavefronts spent half their life in s_waitcnt.  Here the lanes that
// stand on the same leaf scan it TOGETHER, 64 list entries at a time, with the WHOLE wavefront fetching:
//   window    lane i fetches entry i, its header words and its bounding sphere into the wavefront's window in LDS: two
//             round trips for 64 items instead of two per item;
//   gates     every scanning lane walks the window and gates each item against ITS ray with its CURRENT min_dist: a
//             superset of the items it will intersect (min_dist only shrinks; the gate only gets stricter).  No memory;
//   items     the items some lane wants are intersected in list order.  Their parameter records come through two LDS
//             buffers: the record of the next wanted item is on its way while this one is intersected.  Right before an item
//             a lane repeats the gate with the min_dist it has by then -- the reference's gate at the reference's moment.
// Per ray nothing changes: the same items in the same order, the same gate with the same min_dist, the same accept and
// break rules (object.c:692-747, bounding.c:34-85); what changes is who fetches what, and when.  Visited items are known
// from the leaf history (VisitMask<0>), so the scan stores nothing.  Leaves that hold composites (hcube) take the old path.
struct ClsLds {
    double *base;       // this wavefront's window; nullptr = off
    int min_group;      // lanes that must share a leaf for the scan to be done together
};
// 64 entries, 64 x {param_off, words | m << 16}, 64 axis orders, 64 x 2N words (an item's box rows, or its bounding sphere when it has no
// box), two records (2 header words + the sphere + par_words of parameters each)
// ... and the group's beam: per frame axis { u.o of the rays' common origin, the smallest and the largest 1 / (u.v), usable }
template <int N> constexpr int cls_window_words(int par_words) { return 64 + 64 + 64 + 64 * 2 * N + 2 * (2 + N + 2 + par_words) + 4 * N; }
/* sphere (N + 2 words) + parameters are fetched as two words per lane: 128 words a record (ndt_blob.hip:build_blob checks it per N) */

NDT_DEV void cls_lds_sync()
{
    // LDS traffic between lanes of ONE wavefront: the hardware keeps a wavefront's LDS operations in order; this only
    // keeps the compiler from moving them across
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Called by ALL lanes of the wavefront (the fetching is everybody's); `mine` = this lane's ray stands on list [w0, e0) of
// the global blob and knows what it has visited from its history.  For those lanes: the list's result like trace() --
// min_dist (-1: nothing accepted), the accepted item -- and the last item the scan looked at.
// box_slot / boxed: the ray's projections on the item boxes' frame (ray_in_box_frame), made here when the ray has none yet.
template <int N, int MW>
NDT_DEV void cls_scan(const double *blob, const SceneDesc &sd, const VisitMask<MW> &mask, double *win, const bool mine,
                      const double (&o)[N], const double (&v)[N], const double dist_limit, const int w0, const int e0,
                      double *box_slot, bool &boxed, double &min_dist, int &best, int &last
#ifdef NDT_PHASE_TIMING
                      , unsigned long long (&ph)[8], unsigned long long &ph_last
#endif
                      )
{
    const int lane = __lane_id();
    double *l_ref = win, *l_inf = win + 64, *l_ord = win + 128, *l_win = win + 192, *l_par = win + 192 + 64 * 2 * N;
    const int par_stride = 2 + N + 2 + sd.cls_par_words;
    SceneDesc sd_win = sd;      // item_box_meets / bsphere_gate on the window: item k's 2N words at k * 2N
    sd_win.off_obox = 0;
    sd_win.off_oord = -64;      // the items' axis orders sit right in front of the window's rows
    SceneDesc sd_par = sd;      // a staged record: header at words 0 .. 1, the sphere at 2, parameters behind it
    sd_par.off_hdr = 0;
    sd_par.off_bs = 2;
    sd_par.off_params = 2 + N + 2;
    // still scanning (object.c:730 ends a scan early).  An integer, not a boolean: it lives across the loops below, and a boolean that
    // does is an exec mask in a scalar register pair, merged at every join (ndt_device.hpp:trace_kd, the lane's place as `st`)
    int open = mine ? 1 : 0;
    min_dist = -1;
    best = -1;
    last = -1;
    if (mine && box_slot && !boxed) {
        ray_in_box_frame<N>(blob, sd, o, v, box_slot);
        boxed = true;
    }
    // The group's BEAM (round 4).  The rays that scan a leaf together are a tile's primaries or a light's shadow rays: one
    // origin, neighbouring directions.  With one origin the slab arithmetic of item_box_meets is, per axis, a product of a
    // number that is the same for all of them ((+-h - a), a = u.o - centre) and the ray's own 1 / (u.v) -- and a correctly
    // rounded product is monotone in a factor, so over the group the ends of a slab's interval lie between their values at the
    // smallest and at the largest 1 / (u.v) in the group.  An item whose box is missed by that envelope is missed by every ray
    // of the group, exactly as each would compute it: lane i tests item i of the window against the envelope (one pass for 64
    // items), and only the survivors -- a handful of a leaf's 60 .. 227 -- go through the loop in which every lane tests an
    // item against its own ray.  Axes along which some ray runs parallel to the slab, or the reciprocals change sign, give no
    // bound (left out: conservative); rays of several origins: no beam.
    double *l_beam = l_par + 2 * par_stride;
    bool beam = false;
    if (box_slot) {
        cls_lds_sync();
        const unsigned long long grp = __ballot(mine);
        const int first = __ffsll((long long)grp) - 1;
        bool one_origin = true;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            ndt_v2d pr;
            pr.x = 0.0;
            pr.y = 0.0;
            if (mine) pr = *reinterpret_cast<const ndt_v2d *>(box_slot + k * 128);
            const double x0 = lane_get(pr.x, first);
            if (__ballot(mine && pr.x != x0) != 0ull) one_origin = false;
            const bool parallel = __ballot(mine && pr.y == 0.0) != 0ull;
            double lo = mine ? pr.y : NDT_DBL_MAX, hi = mine ? pr.y : -NDT_DBL_MAX;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                const double l2 = __shfl_xor(lo, d, 64), h2 = __shfl_xor(hi, d, 64);
                lo = l2 < lo ? l2 : lo;
                hi = h2 > hi ? h2 : hi;
            }
            if (lane == 0) {
                l_beam[4 * k] = x0;
                l_beam[4 * k + 1] = lo;
                l_beam[4 * k + 2] = hi;
                l_beam[4 * k + 3] = (!parallel && (lo > 0.0 || hi < 0.0)) ? 1.0 : 0.0;
            }
        }
        beam = one_origin;
        cls_lds_sync();
    }
    for (int base = w0; base < e0; base += 64) {
        if (__ballot(open != 0) == 0ull) break;
        const int cnt = (e0 - base < 64) ? e0 - base : 64;
        // ---- the window: entry, header and box (or sphere) of the next `cnt` items, one per lane
        if (lane < cnt) {
            const double ref = blob[base + lane];
            const long long rbits = __double_as_longlong(ref);
            const int id = (int)(rbits & 0xffffffffll);
            const bool has_box = ((int)(rbits >> 32) & NDT_F_OBOX) != 0;
            const ndt_v2d hdr = blob_pair(blob, sd.off_hdr + 2 * id);
            // (2N words from either place: behind a sphere's N + 2 words the next sphere follows, and the blob goes on behind the last)
            const int src = has_box ? sd.off_obox + id * 2 * N : sd.off_bs + id * (N + 2);
            ndt_v2d w2[N];
            if (has_box) {
#pragma unroll
                for (int c = 0; c < N; ++c) w2[c] = blob_pair(blob, src + 2 * c);
            } else {
#pragma unroll
                for (int c = 0; c < N; ++c) {
                    w2[c].x = blob[src + 2 * c];
                    w2[c].y = blob[src + 2 * c + 1];
                }
            }
            l_ref[lane] = ref;
            if (has_box) l_ord[lane] = blob[sd.off_oord + id];
            const long long h0 = __double_as_longlong(hdr.x), h1 = __double_as_longlong(hdr.y);
            l_inf[lane] = __longlong_as_double(((h0 >> 32) & 0xffffffffll) | (((h1 & 0xffffll) | ((h1 >> 32) << 16)) << 32));
#pragma unroll
            for (int c = 0; c < N; ++c) *reinterpret_cast<ndt_v2d *>(l_win + lane * 2 * N + 2 * c) = w2[c];
        }
        cls_lds_sync();
        NDT_STAMP(4);
        // ---- which of them this ray has visited already (rays past their first leaf)
        unsigned long long seen = 0ull;
        if (open && mask.hist_n > 0) {
            for (int k0 = 0; k0 < cnt; k0 += 8) {
                int ids[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) ids[j] = (int)(__double_as_longlong(l_ref[(k0 + j < cnt) ? k0 + j : cnt - 1]) & 0xffffffffll);
                seen |= (unsigned long long)mask.seen_of_8(ids) << k0;
            }
        }
        // ---- the items this ray may have to intersect: its box, or else its sphere gate with the min_dist the ray has
        // now (object.c:618-624; a superset, the gate is repeated at its moment below); and those any ray may
        unsigned long long want = 0ull, any = 0ull;
        if (__ballot(open != 0) != 0ull) {
            // the window's items the beam leaves: item `lane` against the group's envelope (boxed items; the others stay)
            unsigned long long todo = (cnt < 64) ? ((1ull << cnt) - 1ull) : ~0ull;
            if (beam) {
                bool keep = true;
                if (lane < cnt && (((int)(__double_as_longlong(l_ref[lane]) >> 32)) & NDT_F_OBOX)) {
                    const unsigned long long ord = (unsigned long long)__double_as_longlong(l_ord[lane]);
                    double t0 = 0.0, t1 = NDT_DBL_MAX;
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        const int axis = (int)((ord >> (4 * k)) & 15ull);
                        const ndt_v2d box = *reinterpret_cast<const ndt_v2d *>(l_win + lane * 2 * N + 2 * k);
                        const double x0 = l_beam[4 * axis], i0 = l_beam[4 * axis + 1], i1 = l_beam[4 * axis + 2];
                        if (l_beam[4 * axis + 3] != 0.0) {
                            const double a = x0 - box.x, h = box.y;
                            const double p = -h - a, q = h - a;
                            const double c1 = p * i0, c2 = p * i1, c3 = q * i0, c4 = q * i1;
                            const double lo12 = c1 < c2 ? c1 : c2, lo34 = c3 < c4 ? c3 : c4;
                            const double hi12 = c1 < c2 ? c2 : c1, hi34 = c3 < c4 ? c4 : c3;
                            const double lo = lo12 < lo34 ? lo12 : lo34, hi = hi12 > hi34 ? hi12 : hi34;
                            if (lo > t0) t0 = lo;
                            if (hi < t1) t1 = hi;
                        }
                    }
                    keep = t0 <= t1;
                }
                todo &= __ballot(keep);
            }
            while (todo != 0ull) {
                const int k = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const int flags = __builtin_amdgcn_readfirstlane((int)(__double_as_longlong(l_ref[k]) >> 32));
                bool pass = open != 0;
                if (flags & NDT_F_OBOX) {
                    if (open) pass = item_box_meets<N>(l_win, sd_win, k, box_slot);
                } else if (flags & NDT_F_GATE) {
                    // (the sphere sits where a box would: N + 2 of the slot's 2N words)
                    SceneDesc sd_sph = sd;
                    sd_sph.off_bs = k * (2 * N) - k * (N + 2);
                    if (open) pass = bsphere_gate<N>(l_win, sd_sph, k, o, v, min_dist);
                }
                if (pass) want |= 1ull << k;
                if (__ballot(pass && !((seen >> k) & 1ull)) != 0ull) any |= 1ull << k;
            }
            want &= ~seen;
        }
        NDT_STAMP(7);
        // ---- items, in list order; the record of the next one travels while this one is intersected
        int cur = 0;
        auto record_of = [&](int k, int &flags, int &words, int &m, int &at, int &sph) {
            const long long ibits = __double_as_longlong(l_inf[k]);
            const int wm = __builtin_amdgcn_readfirstlane((int)(ibits >> 32));
            at = sd.off_params + __builtin_amdgcn_readfirstlane((int)(ibits & 0xffffffffll));
            words = (wm & 0xffff) + N + 2;          // the sphere in front
            m = wm >> 16;
            const long long rbits = __double_as_longlong(l_ref[k]);
            flags = __builtin_amdgcn_readfirstlane((int)(rbits >> 32));
            sph = sd.off_bs + __builtin_amdgcn_readfirstlane((int)(rbits & 0xffffffffll)) * (N + 2);
        };
        auto word_of = [&](int j, int words, int at, int sph) {
            // word j of a record: the sphere's N + 2 words, then the parameters
            return (j < words) ? blob[(j < N + 2) ? sph + j : at + j - (N + 2)] : 0.0;
        };
        auto put = [&](double *buf, int flags, int words, int m, double t0, double t1) {
            if (lane < words) buf[2 + lane] = t0;
            if (lane + 64 < words) buf[2 + 64 + lane] = t1;
            if (lane == 0) {
                buf[0] = __longlong_as_double((long long)(unsigned int)flags);         // {flags, param_off 0}
                buf[1] = __longlong_as_double((long long)m << 32);                     // {-, m}
            }
        };
        if (any != 0ull) {
            int flags, words, m, at, sph;
            record_of(__ffsll((long long)any) - 1, flags, words, m, at, sph);
            const double t0 = word_of(lane, words, at, sph), t1 = word_of(lane + 64, words, at, sph);
            put(l_par, flags, words, m, t0, t1);
            cls_lds_sync();
        }
        while (any != 0ull) {
            const int k = __ffsll((long long)any) - 1;
            any &= any - 1ull;
            // the next record sets off
            int n_flags = 0, n_words = 0, n_m = 0, n_at = 0, n_sph = 0;
            double t0 = 0.0, t1 = 0.0;
            if (any != 0ull) {
                record_of(__ffsll((long long)any) - 1, n_flags, n_words, n_m, n_at, n_sph);
                t0 = word_of(lane, n_words, n_at, n_sph);
                t1 = word_of(lane + 64, n_words, n_at, n_sph);
            }
            const long long rbits = __double_as_longlong(l_ref[k]);
            const int id = __builtin_amdgcn_readfirstlane((int)(rbits & 0xffffffffll));
            const int flags = __builtin_amdgcn_readfirstlane((int)(rbits >> 32));
            const double *rec = l_par + cur * par_stride;
            bool pass = open != 0 && ((want >> k) & 1ull);
            if ((flags & NDT_F_GATE) && pass) pass = bsphere_gate<N>(rec, sd_par, 0, o, v, min_dist);      // the gate, at its moment
            if (pass) {
                double res[N], nrm[N];
                if (isect<N, false>(rec, sd_par, 0, o, v, res, nrm)) {
                    const double dist = v_dist<N>(o, res);                                       // object.c:721
                    if (dist > NDT_EPS && (dist + NDT_EPS < min_dist || min_dist < 0)) {         // object.c:722
                        min_dist = dist;
                        best = id;
                    }
                    if (dist_limit == 0.0 || dist < dist_limit) {                                // object.c:730
                        open = 0;
                        last = id;          // the scan ends here: the last item it looked at
                    }
                }
            }
            cur ^= 1;
            if (any != 0ull) put(l_par + cur * par_stride, n_flags, n_words, n_m, t0, t1);
            cls_lds_sync();
            if (__ballot(open != 0) == 0ull) any = 0ull;
        }
        cls_lds_sync();
        NDT_STAMP(2);
        // a lane that is still scanning has looked at every item of the window
        if (open) last = (int)(__double_as_longlong(l_ref[cnt - 1]) & 0xffffffffll);
        cls_lds_sync();
    }
}

// trace_kd (object.c:683) = kd_tree_intersect (kd-tree.c:570-625), with
//   kd_node_intersect (kd-tree.c:482-568)  unrolled onto an explicit stack,
//   trace             (object.c:692-747)   as the list scan,
//   hcube.intersect   (hcube.c:236-250)    as a nested list scan,
// restructured for a 64-wide wavefront as three phases that every lane runs in lock-step:
//
//   T  walk the tree until THIS lane stands on a list to scan (infinite list first, then
//      leaves).  The wavefront leaves phase T when every lane has a list or is finished.
//   G  scan forward through the list until THIS lane finds a primitive whose bounding-sphere
//      gate passes (cheap, ~40 instructions per step).
//   I  intersect that primitive (expensive).  All lanes that found one do it together, each
//      on its own object.
//
// The results are exactly the reference's: per lane, objects are visited in list order, the
// gate sees the running min_dist, the visit mask is set as the scan advances, and the
// dist_limit `break` (object.c:730) ends the scan where the reference's loop would.  What
// changes is only which lanes wait for which: a lane never sits through another lane's
// intersection unless it has one of its own to do.
#define NDT_KD_STACK 40
#define NDT_STACK_FLAG 0x40000000   /* stack entry of the (unreachable) ray-parallel branch */

// Traversal stack: one entry per pending far child, in per-lane scratch arrays; the stack
// pointer is a plain local so that it stays in a register (as a member of a struct holding the
// arrays it lived in scratch too, and every push / pop paid two extra dependent round trips).


// Where the traversal stack lives.  Scratch (per-lane arrays in private memory) always works; LDS
// (one slot per lane and level, [level][lane] so that lanes never share a bank) takes the push /
// pop round trips out of the T phase when the workgroup's LDS has room for depth x lanes entries.
struct KdStackLds {
    double *tu;         // already offset by the lane: element d at [d * stride]
    int *node;          // the PARENT of the pending far child: its split plane gives `a` back exactly (12 bytes per entry, not 20)
    int stride;
};

// UNI (the per-bounce trace kernel of the global-memory tier): where every lane of the wavefront looks at the same item --
// coherent rays scan the same leaf in the same order: 80-90 % of the intersections of the 6-D .. 8-D scenes -- its number
// is made a scalar, so that its records are read with scalar loads into scalar registers: no per-lane addresses, no
// vector registers for data that is the same in all lanes (in 8-D a vector is 16 of them).  Same arithmetic.
// When a wavefront may give up on the last rays of a batch (item-set tier, per-bounce trace kernel): its few slowest rays
// keep 64 lanes for tens of microseconds each; given up, they are queued and traced again from the start by coop_trace
// (below), one ray per wavefront, on the wavefronts that have run out of batches.  tail == nullptr: never.
struct TraceAbandon {
    const int *tail;                // the straggler queue's tail: nothing is given up once it has passed tail_limit
    unsigned long long deadline;    // wall_clock64() after which the batch is over its budget
    int max_live;                   // ... and at most this many of its rays are still being traced
    int tail_limit;
    const int *dry_word;            // nullptr, or: only once this word has reached dry_at (the wavefront's queue shard has run
    int dry_at;                     // dry: the launch is in its tail, and wavefronts without a batch are waiting for stragglers)
};

template <int N, int MW, bool LSTACK = false, bool UNI = false>
NDT_DEV void trace_kd(const double *blob, const SceneDesc &sd, VisitMask<MW> &mask, const double (&o)[N],
                      const double (&v)[N], double dist_limit, int &out_obj, int &out_prim
#ifdef NDT_PHASE_TIMING
                      , unsigned long long (&ph)[8], unsigned int (&cnt)[8], unsigned int (&occ)[8]
#endif
                      , KdStackLds ls = KdStackLds{}, ClsLds cl = ClsLds{}, const bool has_ray = true, double *box_slot = nullptr,
                      const TraceAbandon ab = TraceAbandon{}, bool *gave_up = nullptr, const unsigned long long pre_dead = 0ull)
{
#ifdef NDT_PHASE_TIMING
    unsigned long long ph_last = __builtin_readcyclecounter();
#endif
    // v_inv of kd_tree_intersect (kd-tree.c:576-590), clamped to +-1/EPS^2.  The LDS tiers keep all N in registers (a tree
    // step reads one per node).  The global-memory tier -- a ray walks one root-to-leaf path and then scans hundreds of
    // items -- computes the one it needs at each node instead: N registers (16 in 8-D) the list scan can use.
#ifdef NDT_T1_KEEP_INV
    constexpr bool KEEP_INV = true;         // (experiment: all N in registers in the global-memory tier too)
#else
    constexpr bool KEEP_INV = (MW != 0);
#endif
    auto inv_of = [](double v_i) {
        double r;
        if (v_i < NDT_EPS2 && v_i >= 0.0)
            r = NDT_INV_EPS2;
        else if (v_i > -NDT_EPS2 && v_i <= 0.0)
            r = -NDT_INV_EPS2;
        else
            r = 1.0 / v_i;
        return r;
    };
    double v_inv[N];
    if (KEEP_INV) {
#pragma unroll
        for (int i = 0; i < N; ++i) v_inv[i] = inv_of(v[i]);
    }
    auto inv_at = [&](int dim) {
        if constexpr (KEEP_INV) return v_pick<N>(v_inv, dim);
        else return inv_of(v_pick<N>(v, dim));
    };
    NDT_STAMP(5);

    // results
    // (`ret` of the infinite list is inf_obj >= 0, `lret` of the tree l_obj >= 0: a list returns a hit exactly when it accepted one,
    // and the first hit a leaf returns is always taken -- lt starts at DBL_MAX)
    double t_inf = NDT_DBL_MAX;             // `t` of kd_tree_intersect, kd-tree.c:593
    int inf_obj = -1, inf_prim = -1;
    double lt = NDT_DBL_MAX;                // `lt`, kd-tree.c:599
    int l_obj = -1, l_prim = -1;

    // traversal state
    int st_node[LSTACK ? 1 : NDT_KD_STACK];
    double st_a[LSTACK ? 1 : NDT_KD_STACK], st_tu[LSTACK ? 1 : NDT_KD_STACK];
    int sp = 0;
    int node = 0;
    double ntl = 0, ntu = 0;
    // Where the lane stands, as ONE integer (round 4: every boolean that lives across these divergent loops is an exec mask in a
    // scalar register pair, merged with three scalar instructions at every join -- the loops were more scalar than vector code):
    //   0 a node to pop   1 a node to visit   2 a leaf's list to scan   3 finished   4 the root box to test   5 the infinite list to scan
    int st = 4;
    bool boxed = false;             // this ray's projections on the scene's box frame are in its LDS slot (item boxes)

    // Item sets (MW == 1).  The reference scans a leaf's list in order and skips the items this ray has visited
    // (object.c:707-713).  Lists are ascending in item number, so "the unvisited items of the list, in list order" is
    // the leaf's set minus the visit mask, taken from the lowest bit up: no list reads, no iterations spent on items
    // that are skipped anyway (on the benchmark scene an item sits in 20 leaves: half of all scan steps were skips), and
    // a leaf whose items have all been visited is passed without leaving the tree walk.
    constexpr bool BITS = (MW == 1);
    unsigned long long cand = 0ull;         // BITS: items of the current outer list still to be scanned
    // current list
    int sec = 0, pos = 0, end = 0;
    double min_dist = -1;                   // trace()'s min_dist for the current list
    int best_obj = -1, best_prim = -1;
    // nested (hcube) list
    int sub_i = 0, sub_end = 0, sub_owner = -1, sub_prim = -1;
    long long sub_live = -1;                // faces still to scan, bit 0 = the one at sub_i (all ones: every face; 0: sub_i
                                            // stands at the start of a chunk of 63 faces whose boxes are yet to be looked at)
    double sub_min = -1;

    if (!has_ray) {
        // (UNI wavefronts only: a lane without a ray that stays as a helper of the coherent leaf scan)
        st = 3;
    } else if (sd.n_inf > 0) {
        // infinite objects first, linear, unmasked (kd-tree.c:594)
        st = 5;
        if (BITS) {
            cand = sd.inf_bits;
        } else {
            sec = sd.off_inf;
            pos = 0;
            end = sd.n_inf;
        }
    }

    // The inner loops below are written with ONE back edge each and if/else bodies (no
    // `continue` / `break`): hipcc then emits one structured loop per phase instead of the nest
    // of exec-mask bookkeeping loops it builds for early-continue code, which cost about as many
    // scalar instructions as the arithmetic itself.
    while (true) {
        if (MW == 1 && !UNI && ab.tail) {
            // the rays still here (every active lane has one) are the batch's slowest: over the budget, and few enough
            // that a wavefront each serves them better?  (wave-uniform: active lanes, a scalar clock, one address)
            if (__popcll(__ballot(true)) <= ab.max_live && wall_clock64() > ab.deadline &&
                (!ab.dry_word || __builtin_amdgcn_readfirstlane(__hip_atomic_load(ab.dry_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= ab.dry_at) &&
                __builtin_amdgcn_readfirstlane(__hip_atomic_load(ab.tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < ab.tail_limit) {
                *gave_up = true;
                break;
            }
        }
        // ------------------------------------------------------------ phase T
        if (st == 4) {
            // aabb_intersect on the root box, kd-tree.c:84-127
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
                st = 3;
            } else {
                mask.clear(sd.mask_words);
                // item sets: items the caller knows no ray of this batch can pass the gate of count as visited from the start
                if (BITS) mask.w[0] |= pre_dead;
                if (BITS && sd.gate_bits != 0ull) {
                    // Gate prepass (experiment).  The gate of bounding.c:34-85 has a part that does not depend on min_dist: the
                    // ray's line misses the sphere, or the sphere lies behind the ray.  An item that fails it can never pass its
                    // gate, whenever the scan reaches it: the only thing it would do is get its visit mark.  Marking those items
                    // up front -- same arithmetic, item by item, before the walk -- takes them out of every leaf set, and subtrees
                    // that hold nothing else are not entered.
                    unsigned long long todo = sd.gate_bits & ~pre_dead, dead = 0ull;
                    while (todo != 0ull) {
                        const int id = __ffsll((long long)todo) - 1;
                        todo &= todo - 1ull;
                        const int b = sd.off_bs + id * (N + 2);
                        double c[N], oc[N];
                        blob_vec<N>(blob, b, c);
                        v_sub<N>(o, c, oc);
                        const double oc_len2 = v_dot<N>(oc, oc);
                        const double voc = v_dot<N>(v, oc);
                        const double voc2 = voc * voc;
                        const double desc = voc2 - oc_len2 + blob[b + N + 1];
                        if (desc < 0.0 || (voc > 0.0 && voc2 > desc)) dead |= 1ull << id;
                    }
                    mask.w[0] |= dead;
                }
                node = 0;
                ntl = tl;
                ntu = tu;
                st = 1;
            }
        }
        // The lane's place in the walk is ONE integer -- 0: a node to pop, 1: a node to visit, 2: a list to scan, 3: finished --
        // not three booleans: the compiler carries a boolean that lives across a divergent loop as an exec mask in a scalar
        // register pair and merges it at every join (the loop was 99 vector + 118 scalar instructions; with the integer 105 + 60,
        // and the benchmark frame went 1.43 -> 1.35 ms: round 4).  The parallel-plane case of kd-tree.c:555-565 is gone: v_inv is
        // clamped to +-1/EPS^2 when it is made (kd-tree.c:576-590; inv_of above), so `-INV_EPS2 <= v_inv_i <= INV_EPS2` always holds.
        {
            while (st < 2) {
                NDT_OCC(0);
                if (st == 0) {
                    if (sp == 0) {
                        st = 3;
                    } else {
                        // (LDS stack: sp counts in entries' distance, ls.stride -- no multiplication per push and pop)
                        sp -= LSTACK ? ls.stride : 1;
                        double a;
                        if (LSTACK) {
                            // the entry names the parent; tp and the far child are recomputed from its record
                            // with the operations of the push (same operands, same result)
                            const int parent = ls.node[sp];
                            ntu = ls.tu[sp];
                            const ndt_v2d prec = blob_pair(blob, sd.off_kd + 2 * parent);
                            const long long pw0 = __double_as_longlong(prec.x);
                            const int pdim = (int)(pw0 & 0xffffffffll);
                            const double pv_inv = inv_at(pdim);
                            a = (prec.y - v_pick<N>(o, pdim)) * pv_inv;
                            node = (pv_inv < NDT_EPS2) ? parent + 1 : (int)(pw0 >> 32);
                        } else {
                            node = st_node[sp];
                            a = st_a[sp];
                            ntu = st_tu[sp];
                        }
                        ntl = a - NDT_EPS;
                        // `*t_ptr > tp` (kd-tree.c:552), evaluated now that the near subtree is done; `tu < 0` (kd-tree.c:490)
                        st = ((lt > a) & !(ntu < 0.0)) ? 1 : 0;
                    }
                } else if (ntu < 0.0) {
                    st = 0;                                             // kd-tree.c:490
                }
                if (st == 1) {
                    NDT_COUNT(0);
                    ndt_v2d rec = blob_pair(blob, sd.off_kd + 2 * node);
                    // BITS: the items below this node that the ray has not visited yet.  None left: nothing in this subtree can
                    // change the ray's state any more (its leaves would scan nothing), so the walk does not enter it.
                    unsigned long long below = ~0ull;
                    if (BITS) {
                        double ns = blob[sd.off_nset + node];
#ifndef NDT_NO_T_HOIST
                        // both reads in flight before the test of `below` (the compiler sinks the record's read behind it otherwise:
                        // two LDS round trips in a row on the step's critical path)
                        asm volatile("" : "+v"(rec.x), "+v"(rec.y), "+v"(ns));
#endif
                        below = (unsigned long long)__double_as_longlong(ns) & ~mask.w[0];
                    }
                    const long long w0 = __double_as_longlong(rec.x);
                    const int dim = (int)(w0 & 0xffffffffll);
                    if (BITS && below == 0ull) {
                        st = 0;                                         // (subtree exhausted)
                    } else if (dim < 0) {
                        // leaf: trace() over its items (kd-tree.c:497-519)
                        if (BITS) {
                            st = 2;
                            cand = below;
                        } else {
                            const long long w1 = __double_as_longlong(rec.y);
                            const int num = (int)(w1 >> 32);
                            st = num > 0 ? 2 : 0;
                            sec = sd.off_leaf;
                            pos = (int)(w1 & 0xffffffffll);
                            end = pos + num;
                        }
                    } else {
                        const double boundary = rec.y;
                        const double v_inv_i = inv_at(dim);
                        const double o_i = v_pick<N>(o, dim);
                        // preorder: the left child follows its parent; swap for negative directions
                        const bool swap = v_inv_i < NDT_EPS2;
                        const int left = node + 1, right = (int)(w0 >> 32);
                        const int near = swap ? right : left, far = swap ? left : right;
                        // kd-tree.c:541-554, its three cases as selects.  `lt` only ever decreases, so testing `lt > tp` before
                        // pushing the far child is safe; the test that counts is repeated at pop time.
                        // (& and |, not && and ||: a short-circuit is a branch, and every comparison here is cheap)
                        const double tp = (boundary - o_i) * v_inv_i;
                        const bool alive = lt > ntl;
                        const bool near_only = (ntu < tp - NDT_EPS) & alive;                // near, same interval
                        const bool far_only = !near_only & (ntl > tp + NDT_EPS) & alive;    // far, same interval
                        const bool both = !near_only & !far_only;
                        if (both & (lt > tp)) {
                            if (LSTACK) { ls.node[sp] = node; ls.tu[sp] = ntu; }
                            else { st_node[sp] = far; st_a[sp] = tp; st_tu[sp] = ntu; }
                            sp += LSTACK ? ls.stride : 1;               // far: (tp-EPS, tu), gate tp
                        }
                        const bool go_near = near_only | (both & alive);
                        ntu = (both & alive) ? tp + NDT_EPS : ntu;      // near: (tl, tp+EPS)
                        node = go_near ? near : far;
                        st = (go_near | far_only) ? 1 : 0;
                    }
                }
            }
        }
        NDT_STAMP(0);
        // done.  (UNI: a lane whose ray is finished stays with the wavefront until every ray is -- it helps the others fetch
        // in the coherent leaf scan -- and sits out everything else)
        const bool have_list = st == 2 || st == 5, list_is_inf = st == 5;
        if (UNI ? (__ballot(have_list) == 0ull) : !have_list) break;

        // ------------------------------------------------------------ phases G + I over the list
        min_dist = -1;
        best_obj = -1;
        best_prim = -1;
        bool list_open = have_list;
        bool scanned_together = false;
        if (UNI && MW == 0 && cl.base) {
            // coherent leaf scan: the lanes that stand on the same leaf (and know what they visited from their history)
            // scan it together through the wavefront's LDS window, one group of lanes after the other; every lane of the
            // wavefront -- finished ones too -- helps fetching
            const bool elig = have_list && !list_is_inf && mask.hist_n >= 0;
            unsigned long long todo = __ballot(elig);
            const int n_listed = __popcll(__ballot(have_list));
            while (todo != 0ull) {
                const int leader = __ffsll((long long)todo) - 1;
                const int w0 = __builtin_amdgcn_readlane(sec + pos, leader), e0 = __builtin_amdgcn_readlane(sec + end, leader);
                bool mine = elig && sec + pos == w0;
                const unsigned long long group = __ballot(mine);
                todo &= ~group;
                // Together only when the leaf is shared by many: a wavefront of incoherent rays (shadow rays from scattered
                // hit points) would scan its leaves one after the other here, where the per-lane scan below does them side by
                // side -- measured: the mean batch 25-30 % faster with every group scanned together, the slowest twice as slow.
                // (a batch that is not full -- a frame's last, lanes whose rays are finished -- counts as whole when all the
                // rays it still has stand on the leaf)
                if (__popcll(group) < cl.min_group && !(__popcll(group) == n_listed && n_listed >= 16)) mine = false;
                if (__ballot(mine) == 0ull) continue;
                double md;
                int best, last;
                NDT_STAMP(3);
#ifdef NDT_PHASE_TIMING
                cls_scan<N, MW>(blob, sd, mask, cl.base, mine, o, v, dist_limit, w0, e0, box_slot, boxed, md, best, last, ph, ph_last);
#else
                cls_scan<N, MW>(blob, sd, mask, cl.base, mine, o, v, dist_limit, w0, e0, box_slot, boxed, md, best, last);
#endif
                if (mine) {
                    min_dist = md;
                    best_obj = best;
                    best_prim = best;
                    mask.end_leaf(blob_int(blob, sd.off_kd + 2 * node, 1), last);
                    list_open = false;
                    scanned_together = true;
                }
            }
        }
        // Where the lane stands in its list is ONE integer, as in the walk above (booleans that live across these loops cost a
        // scalar register pair and three scalar instructions at every join): 0 scanning the list, 1 scanning an hcube's faces
        // (the nested trace() of hcube.c:241), 2 / 3 a primitive of the list / a face to intersect, 4 the list is done.
        int gs = list_open ? 0 : 4;
        while (gs != 4) {
            // ---- phase G: advance to the next primitive that passes its gate
            int prim = -1;
            while (gs < 2) {
                NDT_OCC(1);
                if (gs == 1 && sub_live == 0) {
                    // The hcube's faces from the chunk sub_i stands on: hull box, then the boxes of 63 faces at a time until
                    // the ray meets one (one call site for the first faces of an hcube and for its later chunks).  Nothing
                    // met: the nested trace() is over -- for an hcube just entered, before it began (its result: no hit).
                    const int oflags = blob_int(blob, sd.off_hdr + 2 * sub_owner, 0);
                    const int first = blob_int(blob, sd.off_hdr + 2 * sub_owner + 1, 0);
                    const bool fresh_hcube = sub_i == first;
                    long long live = -1;
                    int chunk = (sub_i - first) / NDT_HULL_CHUNK;
                    if (oflags & NDT_F_BOX)
                        live = hull_faces<N>(blob, sd.off_params + blob_int(blob, sd.off_hdr + 2 * sub_owner, 1),
                                             (oflags & NDT_F_FACEBOX) != 0, sub_end - first, o, v, chunk, (oflags & NDT_F_FACETREE) != 0,
                                             (oflags & NDT_F_FACEGROUPS) != 0);
                    if (live == 0) {
                        // (an hcube just entered: it never began -- no result to apply, the scan goes on in this very step)
                        if (fresh_hcube) gs = 0;
                        else sub_i = sub_end;
                        sub_live = -1;
                    } else {
                        const int skip = __ffsll(live) - 1;
                        sub_i = first + ((live == -1) ? 0 : chunk * NDT_HULL_CHUNK) + skip;
                        sub_live = live >> skip;
                    }
                }
                if (gs == 1 && sub_i == sub_end) {
                    // nested trace() finished: hcube.intersect returns (hcube.c:241-248),
                    // then the outer trace() applies its accept / break rules
                    gs = 0;
                    if (sub_min >= 0) {
                        const double dist = sub_min;    // == |o - res| of the accepted face
                        if (dist > NDT_EPS && (dist + NDT_EPS < min_dist || min_dist < 0)) {
                            min_dist = dist;
                            best_obj = sub_owner;
                            best_prim = sub_prim;
                        }
                        if (dist_limit == 0.0 || dist < dist_limit) {               // break
                            if (BITS) cand = 0ull;
                            else end = pos;         // (not pos = end: pos - 1 stays the last item the scan looked at)
                        }
                    }
                } else if (gs == 0 && (BITS ? cand == 0ull : pos == end)) {
                    gs = 4;                 // list exhausted
                } else {
                    // one item of the list: visit mark, gate, and what passing the gate leads to
                    // (LDS tiers: the item's sphere record is read together with its header word, ahead of the test of the gate flag
                    // -- one LDS round trip on the step's critical path instead of two; every item has a record slot)
                    #ifdef NDT_NO_SPHERE_AHEAD
                    constexpr bool SPHERE_AHEAD = false;
#else
                    constexpr bool SPHERE_AHEAD = (MW != 0);
#endif
                    double sph_c[N], sph_r = 0.0, sph_r2 = 0.0;
                    auto sphere_ahead = [&](const int id) {
                        if constexpr (SPHERE_AHEAD) {
                            const int b = sd.off_bs + id * (N + 2);
                            blob_vec<N>(blob, b, sph_c);
                            sph_r = blob[b + N];
                            sph_r2 = blob[b + N + 1];
                        }
                    };
                    auto scan_item = [&](const int id, int flags, const bool masked) {
                        if constexpr (SPHERE_AHEAD) {
                            asm volatile("" : "+v"(flags), "+v"(sph_r), "+v"(sph_r2));
#pragma unroll
                            for (int i = 0; i < N; ++i) asm volatile("" : "+v"(sph_c[i]));
                        }
                        bool fresh = true;
                        if (masked) fresh = !mask.test_and_set(id);                             // object.c:707-713
                        // item boxes: a ray that misses the item's box cannot be given a hit by its intersect(); the gate
                        // and the intersection are skipped, their answer -- nothing -- stands
                        if (MW == 0 && box_slot && fresh && (flags & NDT_F_OBOX)) {
                            if (!boxed) {
                                ray_in_box_frame<N>(blob, sd, o, v, box_slot);
                                boxed = true;
                            }
                            fresh = item_box_meets<N>(blob, sd, id, box_slot);
                        }
                        if (fresh) {
                            // vect_object_intersect's gate (object.c:618-624), for composites too
                            const double gate_min = (gs == 1) ? sub_min : min_dist;
                            if (gs == 1) NDT_COUNT(1); else NDT_COUNT(3);
                            bool gate = true;
                            if (flags & NDT_F_GATE) {
                                if constexpr (SPHERE_AHEAD) gate = bsphere_gate_rec<N>(sph_c, sph_r, sph_r2, o, v, gate_min);
                                else gate = bsphere_gate<N>(blob, sd, id, o, v, gate_min);
                            }
                            if (gate) {
                                if ((flags & NDT_F_TYPE_MASK) == T_HCUBE) {
                                    // composites only occur in outer lists (validated at upload)
                                    // (which of its faces are scanned is found at the top of the scanning loop: sub_live == 0)
                                    const int first = blob_int(blob, sd.off_hdr + 2 * id + 1, 0);
                                    const int nf = blob_int(blob, sd.off_hdr + 2 * id + 1, 1);
                                    gs = 1;
                                    sub_owner = id;
                                    sub_i = first;
                                    sub_end = first + nf;
                                    sub_live = 0;
                                    sub_min = -1;
                                    sub_prim = -1;
                                } else {
                                    prim = id;
                                    if (gs == 1) NDT_COUNT(2); else NDT_COUNT(4);
                                    gs += 2;
                                }
                            }
                        }
                    };
                    // UNI: every active lane stands on the same entry of the same outer list (coherent rays): the entry, the
                    // item's header flags and its bounding sphere are then scalars, read with scalar loads; the branches on
                    // them are scalar branches.  Same steps, same order.
                    bool together = false;
                    int w_u = 0;
                    if (UNI && !BITS) {
                        const int w = sec + pos;
                        w_u = __builtin_amdgcn_readfirstlane(w);
                        together = __ballot(gs == 1 || w != w_u) == 0ull;
                    }
                    if (together) {
                        int id, flags;
                        blob_ref(blob, w_u, id, flags);
                        pos += 1;
                        sphere_ahead(id);
                        scan_item(id, flags, !list_is_inf);
                    } else {
                        int id, flags;
                        const bool in_sub = gs == 1;
                        if (BITS && !in_sub) {
                            const unsigned long long bit = cand & (0ull - cand);        // the lowest item of the set
                            id = __ffsll((long long)cand) - 1;
                            cand ^= bit;
                            if (!list_is_inf) mask.w[0] |= bit;                         // object.c:713
                            flags = blob_int(blob, sd.off_hdr + 2 * id, 0);
                            sphere_ahead(id);
                        } else {
                            blob_ref(blob, in_sub ? sd.off_child + sub_i : sec + pos, id, flags);
                            sphere_ahead(id);
                        }
                        if (in_sub) {
                            // on to the next face whose box the ray meets (arithmetic shifts: all ones stays all ones)
                            const long long rest = sub_live >> 1;
                            if (rest == 0) {
                                // this chunk of 63 faces is done: the next ones, if the hcube has more (sub_live is all ones
                                // -- never 0 here -- when there are no face boxes)
                                const int first = blob_int(blob, sd.off_hdr + 2 * sub_owner + 1, 0);
                                const int next = first + ((sub_i - first) / NDT_HULL_CHUNK + 1) * NDT_HULL_CHUNK;
                                if (next < sub_end) { sub_i = next; sub_live = 0; }
                                else sub_i = sub_end;
                            } else {
                                const int skip = __ffsll(rest) - 1;
                                sub_i += 1 + skip;
                                sub_live = rest >> skip;
                            }
                        } else if (!BITS) {
                            pos += 1;
                        }
                        scan_item(id, flags, !BITS && !in_sub && !list_is_inf);
                    }
                }
            }
            NDT_STAMP(1);
            if (gs != 4) {
                // ---- phase I: intersect
                NDT_OCC(2);
#ifdef NDT_PHASE_TIMING
                {
                    // how many different primitive types this I iteration executes, and how many lanes the commonest has
                    const int ty = blob_int(blob, sd.off_hdr + 2 * prim, 0) & NDT_F_TYPE_MASK;
                    int kinds = 0, top = 0;
                    for (int t = 0; t < 9; ++t) {
                        const int c = __popcll(__ballot(ty == t));
                        kinds += c > 0;
                        top = c > top ? c : top;
                    }
                    occ[6] += kinds;
                    occ[7] += top;
                }
#endif
                double res[N], nrm[N];
                bool ok;
                const int prim_u = __builtin_amdgcn_readfirstlane(prim);
                if (UNI && __ballot(prim != prim_u) == 0ull) ok = isect<N, false>(blob, sd, prim_u, o, v, res, nrm);
                else ok = isect<N, false>(blob, sd, prim, o, v, res, nrm);
                if (ok) {
                    NDT_COUNT(5);
                    const double dist = v_dist<N>(o, res);          // object.c:721
                    if (gs == 3) {
                        // inner trace(): dist_limit = -1, no mask (hcube.c:241)
                        if (dist > NDT_EPS && (dist + NDT_EPS < sub_min || sub_min < 0)) {
                            sub_min = dist;
                            sub_prim = prim;
                        }
                    } else {
                        if (dist > NDT_EPS && (dist + NDT_EPS < min_dist || min_dist < 0)) {       // object.c:722
                            min_dist = dist;
                            best_obj = prim;
                            best_prim = prim;
                        }
                        if (dist_limit == 0.0 || dist < dist_limit) {                               // object.c:730
                            if (BITS) cand = 0ull;
                            else end = pos;         // (not pos = end: pos - 1 stays the last item the scan looked at)
                        }
                    }
                }
                gs -= 2;            // back to the list (2 -> 0) or to the hcube's faces (3 -> 1)
                NDT_STAMP(2);
            }
        }

        // ---- list finished: what trace() returns to its caller
        NDT_STAMP(3);
        if (MW == 0 && st == 2 && !scanned_together) {
            // the leaf's ordinal (its record, which `node` still names) and the last item the scan looked at
            int last, flags_;
            blob_ref(blob, sec + pos - 1, last, flags_);
            mask.end_leaf(blob_int(blob, sd.off_kd + 2 * node, 1), last);
        }
        if (st == 5) {
            if (min_dist > NDT_EPS) t_inf = min_dist;           // object.c:736
            inf_obj = best_obj;
            inf_prim = best_prim;
            st = 4;                                             // next: the tree
        } else if (st == 2) {
            if (min_dist >= 0 && min_dist < lt) {               // `ret && t < *t_ptr`, kd-tree.c:506
                lt = min_dist;
                l_obj = best_obj;
                l_prim = best_prim;
            }
            st = 0;                                             // next: what the stack holds
        }
        // (st == 3 here: a finished lane of a UNI wavefront)
    }

    out_obj = inf_obj;
    out_prim = inf_prim;
    if (l_obj >= 0) {
        if (inf_obj < 0 || (lt > NDT_EPS && lt + NDT_EPS < t_inf)) {   // kd-tree.c:612
            out_obj = l_obj;
            out_prim = l_prim;
        }
    }
}

// ------------------------------------------------------------------ cooperative trace: ONE ray, a whole wavefront
//
// The slowest rays of a frame (a ray along the edge of an hcube: ~55 tree steps, ~50 gates, ~40 intersections, every one a
// dependent chain in one lane) set the tail of every trace launch and the latency floor of every pass.  What such a ray
// computes is almost all independent of the order it is computed in:
//   * intersect(item, ray) -- for an hcube the nested trace() over its faces, which starts from nothing (hcube.c:241) --
//     and the part of the bounding-sphere gate that does not involve min_dist (bounding.c:60-85) are functions of the ray
//     and the item alone: lane i computes them for item i, all <= 64 items of the scene in one round;
//   * what depends on the order -- trace()'s running min_dist (the gate's first test, bounding.c:44-53; the accept rule
//     object.c:722; the dist_limit break :730), the visit marks, `*t_ptr` across leaves (kd-tree.c:506) and kd_node_intersect's
//     walk itself (kd-tree.c:482-568) -- is then replayed on those results: scalars only, the same comparisons on the same
//     operands in the same order, so the same decisions and the same answer (object, primitive), bit for bit.
// The replay needs only the items whose intersect() returned a hit: an item that is gated out or missed changes nothing
// whenever the scan reaches it, except its own visit mark -- so every such item counts as visited from the start, leaves
// and whole subtrees without an unvisited hit are not entered (the rule trace_kd's item sets already use: `below`), and a
// ray that met forty bounding spheres and three objects walks towards three objects.
//
// Called with all 64 lanes active and the same ray in every lane.  Item-set scenes only (sd.mask_words == 1, sets in the
// blob: sd.off_nset, sd.inf_bits).

// x with lane l's value replaced by val (val, l: the same in every lane)
NDT_DEV int lane_put(int x, int val, int l) { return ((int)__lane_id() == l) ? val : x; }
NDT_DEV double lane_put(double x, double val, int l) { return ((int)__lane_id() == l) ? val : x; }
// a condition that is the same in every lane, as a scalar branch
#define NDT_UNI(c) (__ballot(c) != 0ull)

template <int N>
NDT_DEV void coop_trace(const double *blob, const SceneDesc &sd, const double (&o)[N], const double (&v)[N], const double dist_limit,
                        int &out_obj, int &out_prim)
{
    const int lane = __lane_id();
    // ---- 1. item `lane` against the ray: gate without min_dist, intersect() (an hcube: its nested trace())
    bool ok = false;                // intersect() was reached and returned a hit
    double dist = 0.0;              // |o - hit|, object.c:721
    double oc2 = -1.0, rad = 0.0;   // the gate's |o - c|^2 and radius (an item without a gate: never rejected)
    int prim = -1;                  // the primitive that was hit (an hcube: the face its trace() accepted)
    {
        const bool has = lane < sd.n_items;
        const int id = has ? lane : 0;
        const int flags = blob_int(blob, sd.off_hdr + 2 * id, 0);
        bool pass = has;
        if (has && (flags & NDT_F_GATE)) {
            // vect_bounding_sphere_intersect without its min_dist test (bsphere_gate with min_dist < 0: same operations)
            const int b = sd.off_bs + id * (N + 2);
            double c[N], oc[N];
            blob_vec<N>(blob, b, c);
            v_sub<N>(o, c, oc);
            const double oc_len2 = v_dot<N>(oc, oc);
            const double voc = v_dot<N>(v, oc);
            const double voc2 = voc * voc;
            const double desc = voc2 - oc_len2 + blob[b + N + 1];
            if (desc < 0.0 || (voc > 0.0 && voc2 > desc)) pass = false;
            oc2 = oc_len2;
            rad = blob[b + N];
        }
        const bool cube = pass && (flags & NDT_F_TYPE_MASK) == T_HCUBE;
        bool single = pass && !cube;
        // an hcube's faces: hull box, then face boxes 63 at a time, as trace_kd's nested scan takes them
        int f_first = 0, nf = 0, pbox = 0;
        if (cube) {
            f_first = blob_int(blob, sd.off_hdr + 2 * id + 1, 0);
            nf = blob_int(blob, sd.off_hdr + 2 * id + 1, 1);
            pbox = sd.off_params + blob_int(blob, sd.off_hdr + 2 * id, 1);
        }
        bool cube_open = cube && nf > 0, all_faces = false;
        long long live = 0;         // faces of chunk `chunk` still to look at (bit j = face 63 chunk + j)
        int chunk = 0, f_next = 0;
        double sub_min = -1;
        int sub_prim = -1;
        while (true) {
            int p = -1;
            bool nested = false;
            if (single) {
                p = id;
                single = false;
            } else {
                while (cube_open && p < 0) {
                    if (live == 0 && !all_faces) {
                        long long lv = -1;
                        int ch = chunk;
                        if (flags & NDT_F_BOX) lv = hull_faces<N>(blob, pbox, (flags & NDT_F_FACEBOX) != 0, nf, o, v, ch, (flags & NDT_F_FACETREE) != 0, (flags & NDT_F_FACEGROUPS) != 0);
                        if (lv == 0) {
                            cube_open = false;          // the ray misses the hull box, or every face box from `chunk` on
                        } else if (lv == -1) {
                            all_faces = true;           // no face boxes: every face, in order
                            f_next = f_first;
                        } else {
                            live = lv;
                            chunk = ch;
                        }
                    }
                    if (cube_open) {
                        int f;
                        if (all_faces) {
                            f = f_next++;
                            if (f_next == f_first + nf) cube_open = false;
                        } else {
                            f = f_first + chunk * NDT_HULL_CHUNK + (__ffsll(live) - 1);
                            live &= live - 1;
                            if (live == 0) {
                                ++chunk;
                                if (chunk * NDT_HULL_CHUNK >= nf) cube_open = false;
                            }
                        }
                        int fid, fflags;
                        blob_ref(blob, sd.off_child + f, fid, fflags);
                        bool gate = true;
                        if (fflags & NDT_F_GATE) gate = bsphere_gate<N>(blob, sd, fid, o, v, sub_min);     // object.c:618-624
                        if (gate) {
                            p = fid;
                            nested = true;
                        }
                    }
                }
            }
            if (__ballot(p >= 0) == 0ull) break;
            if (p >= 0) {
                double res[N], nrm[N];
                if (isect<N, false>(blob, sd, p, o, v, res, nrm)) {
                    const double d = v_dist<N>(o, res);                                     // object.c:721
                    if (nested) {
                        // the hcube's own trace(): dist_limit -1, no mask (hcube.c:241)
                        if (d > NDT_EPS && (d + NDT_EPS < sub_min || sub_min < 0)) {
                            sub_min = d;
                            sub_prim = p;
                        }
                    } else {
                        ok = true;
                        dist = d;
                        prim = p;
                    }
                }
            }
        }
        if (cube) {
            ok = sub_min >= 0;      // hcube.intersect's return (hcube.c:241-248); the distance is the accepted face's
            dist = sub_min;
            prim = sub_prim;
        }
    }
    const unsigned long long okmask = __ballot(ok);

    // ---- 2. the replay: trace() and kd_node_intersect over those results
    unsigned long long visited = ~okmask;
    // trace() (object.c:692-747) over the items of `cand`, ascending
    auto scan = [&](unsigned long long cand, const bool masked, double &min_dist, int &best) {
        min_dist = -1;
        best = -1;
        while (NDT_UNI(cand != 0ull)) {
            const int i = __ffsll((long long)cand) - 1;
            const unsigned long long bit = 1ull << i;
            cand ^= bit;
            if (masked) visited |= bit;                                                     // object.c:713
            const double d = lane_get(dist, i);
            bool pass = true;
            if (NDT_UNI(min_dist > 0)) {                                                    // bounding.c:44-53
                const double min_dist_r = min_dist + lane_get(rad, i);
                if (lane_get(oc2, i) > min_dist_r * min_dist_r) pass = false;
            }
            if (NDT_UNI(pass)) {
                if (d > NDT_EPS && (d + NDT_EPS < min_dist || min_dist < 0)) {              // object.c:722
                    min_dist = d;
                    best = i;
                }
                if (dist_limit == 0.0 || d < dist_limit) cand = 0ull;                       // object.c:730
            }
        }
    };
    // kd_tree_intersect, kd-tree.c:570-625
    double t_inf = NDT_DBL_MAX;
    bool ret_inf = false;
    int inf_best = -1;
    if (sd.n_inf > 0) {
        double md;
        scan(sd.inf_bits & okmask, false, md, inf_best);
        ret_inf = md >= 0;
        if (md > NDT_EPS) t_inf = md;                                                       // object.c:736
    }
    double lt = NDT_DBL_MAX;
    bool lret = false;
    int l_best = -1;
    // aabb_intersect on the root box, kd-tree.c:84-127 (as in trace_kd)
    double tl = -NDT_DBL_MAX, tu = NDT_DBL_MAX;
    bool box = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (box) {
            const double v_i = v[i], o_i = o[i];
            if (!(fabs(v_i) < NDT_EPS2)) {
                double tl_i = (blob[sd.off_bb + i] - o_i) / v_i;
                double tu_i = (blob[sd.off_bb + N + i] - o_i) / v_i;
                if (tl_i > tu_i) {
                    const double tmp = tl_i;
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
    if (NDT_UNI(box && sd.n_kd_nodes > 0)) {
        double v_inv[N];            // kd-tree.c:576-590
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double v_i = v[i];
            if (v_i < NDT_EPS2 && v_i >= 0.0) v_inv[i] = NDT_INV_EPS2;
            else if (v_i > -NDT_EPS2 && v_i <= 0.0) v_inv[i] = -NDT_INV_EPS2;
            else v_inv[i] = 1.0 / v_i;
        }
        // kd_node_intersect on an explicit stack (trace_kd's phase T); entry d of the stack lives in lane d
        int st_node = 0;
        double st_a = 0.0, st_tu = 0.0;
        int sp = 0, node = 0;
        double ntl = tl, ntu = tu;
        bool have_node = true;
        while (true) {
            bool visit = have_node;
            if (NDT_UNI(!have_node)) {
                if (NDT_UNI(sp == 0)) break;
                --sp;
                const int nf = __builtin_amdgcn_readlane(st_node, __builtin_amdgcn_readfirstlane(sp));
                const double a = lane_get(st_a, sp);
                ntu = lane_get(st_tu, sp);
                node = nf & ~NDT_STACK_FLAG;
                ntl = (nf & NDT_STACK_FLAG) ? a : a - NDT_EPS;
                visit = lt > a;                                                             // `*t_ptr > tp`, kd-tree.c:552
            }
            have_node = false;
            if (NDT_UNI(visit && !(ntu < 0.0))) {                                           // kd-tree.c:490
                const int nd = __builtin_amdgcn_readfirstlane(node);
                const ndt_v2d rec = blob_pair(blob, sd.off_kd + 2 * nd);
                const unsigned long long below = (unsigned long long)__double_as_longlong(blob[sd.off_nset + nd]) & ~visited;
                const long long w0 = __double_as_longlong(rec.x);
                const int dim = __builtin_amdgcn_readfirstlane((int)(w0 & 0xffffffffll));
                if (NDT_UNI(below == 0ull)) {
                    // nothing below this node can change the ray's state any more
                } else if (dim < 0) {
                    // a leaf: trace() over its unvisited items (kd-tree.c:497-519)
                    double md;
                    int bi;
                    scan(below, true, md, bi);
                    if (NDT_UNI(md >= 0)) {
                        lret = true;
                        if (md < lt) {                                                      // `ret && t < *t_ptr`, kd-tree.c:506
                            lt = md;
                            l_best = bi;
                        }
                    }
                } else {
                    const double boundary = rec.y;
                    const double v_inv_i = v_pick<N>(v_inv, dim);
                    const double o_i = v_pick<N>(o, dim);
                    const bool swap = v_inv_i < NDT_EPS2;
                    const int left = nd + 1, right = (int)(w0 >> 32);
                    const int near = swap ? right : left, far = swap ? left : right;
                    if (NDT_UNI(-NDT_INV_EPS2 <= v_inv_i && v_inv_i <= NDT_INV_EPS2)) {
                        const double tp = (boundary - o_i) * v_inv_i;
                        const bool alive = lt > ntl;
                        // kd-tree.c:541-554
                        if (NDT_UNI(ntu < tp - NDT_EPS && alive)) {
                            node = near; have_node = true;
                        } else if (NDT_UNI(ntl > tp + NDT_EPS && alive)) {
                            node = far; have_node = true;
                        } else {
                            if (NDT_UNI(lt > tp)) {
                                st_node = lane_put(st_node, far, sp);
                                st_a = lane_put(st_a, tp, sp);
                                st_tu = lane_put(st_tu, ntu, sp);
                                ++sp;
                            }
                            if (alive) { node = near; ntu = tp + NDT_EPS; have_node = true; }
                        }
                    } else {
                        // plane parallel to the ray (kd-tree.c:555-565): unreachable, as in trace_kd
                        if (NDT_UNI(o_i > boundary - NDT_EPS)) {
                            st_node = lane_put(st_node, far | NDT_STACK_FLAG, sp);
                            st_a = lane_put(st_a, ntl, sp);
                            st_tu = lane_put(st_tu, ntu, sp);
                            ++sp;
                        }
                        if (o_i < boundary + NDT_EPS && lt > ntl) { node = near; have_node = true; }
                    }
                }
            }
        }
    }
    int best = inf_best;
    if (lret) {
        if (!ret_inf || (lt > NDT_EPS && lt + NDT_EPS < t_inf)) best = l_best;              // kd-tree.c:612
    }
    best = __builtin_amdgcn_readfirstlane(best);
    out_obj = best;
    out_prim = best >= 0 ? __builtin_amdgcn_readlane(prim, best) : -1;
}
