/*
 * ndt_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of the ndt ray-trace hot path.
 *
 * Plain C99 restatement, operation for operation, of the reference's
 *   render_image -> get_pixel_color -> get_ray_color / apply_lights -> trace_kd ->
 *   kd_tree_intersect -> trace -> vect_object_intersect -> <plugin>.intersect
 * over an `ndt_flat_scene` (include/ndt_hip.h).  Each function cites the reference
 * file:line it follows.  It exists to CHECK the HIP path (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg); nothing in the product may link, import or call it.
 *
 * Pinning: tests/test_oracle_golden.py compares this file, bit for bit, with double
 * framebuffers, trace_kd known answers and trace_kd call counts produced by the compiled
 * reference (oracle/_ref, built by oracle/Makefile from /root/reference; generator script
 * tests/golden/make_golden.py).  The reference ships no test vectors of its own
 * (SURVEY.md section 4).
 *
 * Arithmetic contract: IEEE double, no FMA contraction
 * (-ffp-contract=off), dot products summed in the SSE2 lane-pair order of vectNd.h:215-227
 * (even components in lane 0, odd components in lane 1, lanes added last).
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "ndt_hip.h"

#define EPS  (1e-4)                 /* object.h:15 */
#define EPS2 ((EPS) * (EPS))        /* object.h:17 */
#define INV_EPS2 (1.0 / (EPS2))     /* kd-tree.c:480 */
#define ND 16                       /* stack vectors; dims <= ND */
#define MAXV(a, b) (((a) > (b)) ? (a) : (b))   /* image.h:30-32 (included by ndt.c:24 before ndt.c:32's own #ifndef MAX) */

/* ---------------------------------------------------------------- vectNd.h / vectNd.c */

/* vectNd_dot, vectNd.h:215-227 (SSE2 path): lane sums over even / odd components. */
static double v_dot(const double *a, const double *b, int n)
{
    double s0 = a[0] * b[0];
    double s1 = a[1] * b[1];
    int k = (n + 1) >> 1;
    for (int i = 1; i < k; ++i) {
        s0 = s0 + a[2 * i] * b[2 * i];
        if (2 * i + 1 < n)
            s1 = s1 + a[2 * i + 1] * b[2 * i + 1];
        /* odd n: the pad lane multiplies 0*0 (vectNd.h:147) and adds +0 -- a no-op */
    }
    return s0 + s1;
}
static void v_add(const double *a, const double *b, double *r, int n) { for (int i = 0; i < n; ++i) r[i] = a[i] + b[i]; }
static void v_sub(const double *a, const double *b, double *r, int n) { for (int i = 0; i < n; ++i) r[i] = a[i] - b[i]; }
static void v_scale(const double *a, double s, double *r, int n) { for (int i = 0; i < n; ++i) r[i] = a[i] * s; }
static void v_copy(double *d, const double *s, int n) { memcpy(d, s, (size_t)n * sizeof(double)); }
static void v_zero(double *d, int n) { memset(d, 0, (size_t)n * sizeof(double)); }
/* vectNd_l2norm, vectNd.h:315 */
static double v_len(const double *a, int n) { return sqrt(v_dot(a, a, n)); }
/* vectNd_unitize, vectNd.h:323: untouched when |len| <= EPSILON */
static void v_unitize(double *a, int n)
{
    double len = v_len(a, n);
    if (len > EPS || len < -EPS)
        v_scale(a, 1.0 / len, a, n);
}
/* vectNd_dist, vectNd.h:331 */
static double v_dist(const double *a, const double *b, int n)
{
    double d[ND];
    v_sub(a, b, d, n);
    return v_len(d, n);
}
/* vectNd_proj_unit, vectNd.h:346 */
static void v_proj_unit(const double *v, const double *onto, double *r, int n)
{
    double ab = v_dot(v, onto, n);
    v_scale(onto, ab, r, n);
}
/* vectNd_proj, vectNd.h:355 */
static void v_proj(const double *v, const double *onto, double *r, int n)
{
    double bb = v_dot(onto, onto, n);
    double ab = v_dot(v, onto, n);
    v_scale(onto, ab / bb, r, n);
}
/* vectNd_angle, vectNd.c:64 */
static double v_angle(const double *a, const double *b, int n)
{
    double dp = v_dot(a, b, n);
    double l1 = v_len(a, n);
    double l2 = v_len(b, n);
    double div = l1 * l2;
    if (fabs(div) > EPS)
        return acos(dp / div);
    return -1;
}
/* vectNd_angle3, vectNd.c:83 */
static double v_angle3(const double *p1, const double *p2, const double *p3, int n)
{
    double a[ND], b[ND];
    v_sub(p1, p2, a, n);
    v_sub(p3, p2, b, n);
    return v_angle(a, b, n);
}
/* vectNd_reflect, vectNd.c:101 */
static void v_reflect(const double *u, const double *nrm, double *res, double mag, int n)
{
    double nnu[ND];
    double nu = v_dot(nrm, u, n);
    double nn = v_dot(nrm, nrm, n);
    v_scale(nrm, (1 + mag) * nu / nn, nnu, n);
    v_sub(u, nnu, res, n);
}
/* vectNd_refract, vectNd.c:119.  Unitizes the caller's normal in place (vectNd.c:155). */
static void v_refract(const double *u, double *nrm, double *res, double index, int n)
{
    double rev_u[ND], rev_n[ND], un[ND], np[ND], ref_n[ND], ref_p[ND];
    v_scale(u, -1, rev_u, n);
    v_scale(nrm, -1, rev_n, n);
    double un_dot = v_dot(rev_u, nrm, n);
    double theta_in;
    if (un_dot < 0) {
        index = 1 / index;
        theta_in = v_angle(rev_u, rev_n, n);
    } else {
        theta_in = v_angle(rev_u, nrm, n);
    }
    double theta_out;
    double sin_out = sin(theta_in) / index;
    if (sin_out <= 1.0)
        theta_out = asin(sin_out);
    else
        theta_out = M_PI - theta_in;
    v_unitize(rev_n, n);
    v_unitize(nrm, n);
    v_proj_unit(u, rev_n, un, n);
    v_sub(u, un, np, n);
    v_unitize(np, n);
    double rn = cos(theta_out);
    double rp = sin(theta_out);
    if (un_dot < 0)
        v_scale(nrm, rn, ref_n, n);
    else
        v_scale(rev_n, rn, ref_n, n);
    v_scale(np, rp, ref_p, n);
    v_add(ref_n, ref_p, res, n);
}

/* ---------------------------------------------------------------- prepared scene */

typedef struct {
    int type, transparent, parent;
    const double *pos, *dir, *size;     /* into the flat pools */
    const int *flag, *child;
    int n_pos, n_dir, n_size, n_flag, n_child;
    const double *bcenter;
    double bradius, bradius2;           /* bounding.c:22 */
    double red, green, blue, red_r, green_r, blue_r, refract_index;
    /* plugin prepare() data */
    int m;                              /* number of axes / bases */
    double r2;                          /* sphere.c:25 */
    double *axes;                       /* m x n: cylinder axis, hcylinder axes, orthotope basis, facet basis */
    double *lengths, *AdA, *BdA;        /* per axis */
    double *edge, *unit_edge;           /* 3 x n (hfacet, facet) */
    double *edge_perp;                  /* n (hfacet) */
    double angle[3];                    /* facet.c:62 */
    int inf_ends;                       /* cylinder/hcylinder: skip the ends test */
} pobj;

typedef struct {
    int n;
    const ndt_flat_scene *fs;
    pobj *objs;
    int n_objs, n_items;
    const double *cam_pos, *cam_img_orig, *cam_dir_y;
    double cam_dir_x[ND];               /* scaled per render (ndt.c:926) */
    /* the rest of the camera (ABI 2), NULL when the scene does not carry it */
    const double *cam_left_eye, *cam_right_eye, *cam_local_x, *cam_local_y, *cam_local_z;
    const double *bb_lower, *bb_upper;
} pscene;

static double *dalloc(int k) { return (double *)calloc((size_t)(k > 0 ? k : 1), sizeof(double)); }

static int prepare_object(const ndt_flat_scene *fs, int idx, pobj *p)
{
    const ndt_flat_object *fo = &fs->objects[idx];
    int n = fs->dims;
    memset(p, 0, sizeof(*p));
    p->type = fo->type;
    p->transparent = fo->transparent;
    p->parent = fo->parent;
    p->pos = fs->vecs + fo->pos_off;
    p->dir = fs->vecs + fo->dir_off;
    p->size = fs->sizes + fo->size_off;
    p->flag = fs->flags + fo->flag_off;
    p->child = fs->obj_refs + fo->obj_off;
    p->n_pos = fo->n_pos; p->n_dir = fo->n_dir; p->n_size = fo->n_size; p->n_flag = fo->n_flag; p->n_child = fo->n_obj;
    p->bcenter = fs->vecs + fo->bounds_center_off;
    p->bradius = fo->bounds_radius;
    p->bradius2 = fo->bounds_radius * fo->bounds_radius;
    p->red = fo->red; p->green = fo->green; p->blue = fo->blue;
    p->red_r = fo->red_r; p->green_r = fo->green_r; p->blue_r = fo->blue_r;
    p->refract_index = fo->refract_index;

    switch (p->type) {
    case NDT_OBJ_SPHERE:        /* sphere.c:18-32; pow(r,2.0) is r*r */
        if (p->n_pos < 1 || p->n_size < 1) return -1;
        p->r2 = p->size[0] * p->size[0];
        break;
    case NDT_OBJ_HPLANE:
    case NDT_OBJ_HDISK:
        if (p->n_pos < 1 || p->n_dir < 1 || (p->type == NDT_OBJ_HDISK && p->n_size < 1)) return -1;
        break;
    case NDT_OBJ_CYLINDER: {    /* cylinder.c:22-40 */
        if (p->n_pos < 2 || p->n_size < 1) return -1;
        p->m = 1;
        p->axes = dalloc(n); p->lengths = dalloc(1); p->AdA = dalloc(1); p->BdA = dalloc(1);
        v_sub(p->pos + n, p->pos, p->axes, n);
        v_unitize(p->axes, n);
        p->lengths[0] = v_dist(p->pos + n, p->pos, n);
        p->AdA[0] = v_dot(p->axes, p->axes, n);
        p->BdA[0] = v_dot(p->pos, p->axes, n);
        p->inf_ends = (p->n_flag > 1 && p->flag[1] != 0);   /* cylinder.c:87 */
        break;
    }
    case NDT_OBJ_HCYLINDER: {   /* hcylinder.c:23-54 */
        int m = n - 2;
        if (p->n_pos < n - 1 || p->n_size < 1) return -1;
        p->m = m;
        p->axes = dalloc(m * n); p->lengths = dalloc(m); p->AdA = dalloc(m); p->BdA = dalloc(m);
        for (int i = 0; i < m; ++i) {
            double *ax = p->axes + i * n;
            v_sub(p->pos + (i + 1) * n, p->pos, ax, n);
            v_unitize(ax, n);
            p->lengths[i] = v_dist(p->pos + (i + 1) * n, p->pos, n);
            p->AdA[i] = v_dot(ax, ax, n);
            p->BdA[i] = v_dot(p->pos, ax, n);
        }
        p->inf_ends = (p->n_flag != 0 && p->flag[0] != 0);  /* hcylinder.c:107 */
        break;
    }
    case NDT_OBJ_ORTHOTOPE: {   /* orthotope.c:23-54 */
        if (p->n_flag < 1 || p->n_pos < 1) return -1;
        int m = p->flag[0];
        if (m < 0 || m > p->n_dir) return -1;
        p->m = m;
        p->axes = dalloc(m * n); p->lengths = dalloc(m); p->AdA = dalloc(m); p->BdA = dalloc(m);
        for (int i = 0; i < m; ++i) {
            double *b = p->axes + i * n;
            v_copy(b, p->dir + i * n, n);
            v_unitize(b, n);
            p->lengths[i] = v_len(p->dir + i * n, n);
            p->AdA[i] = v_dot(b, b, n);             /* BdB */
            p->BdA[i] = v_dot(p->pos, b, n);        /* BdP */
        }
        break;
    }
    case NDT_OBJ_HCUBE:
        if (p->n_child < 1) return -1;
        break;
    case NDT_OBJ_HFACET: {      /* hfacet.c:43-87 */
        if (p->n_pos < 3 || p->n_flag < 1) return -1;
        if (p->flag[0] ? p->n_dir < 3 : 0) return -1;
        p->edge = dalloc(3 * n); p->unit_edge = dalloc(3 * n); p->edge_perp = dalloc(n);
        for (int i = 0; i < 3; ++i) {
            int j = (i + 1) % 3;
            v_sub(p->pos + j * n, p->pos + i * n, p->edge + i * n, n);
            v_copy(p->unit_edge + i * n, p->edge + i * n, n);
            v_unitize(p->unit_edge + i * n, n);
        }
        v_scale(p->edge + 2 * n, -1.0, p->edge + 2 * n, n);
        v_scale(p->unit_edge + 2 * n, -1.0, p->unit_edge + 2 * n, n);
        double e2e0[ND];
        v_proj(p->edge + 2 * n, p->edge, e2e0, n);
        v_sub(p->edge + 2 * n, e2e0, p->edge_perp, n);
        v_unitize(p->edge_perp, n);
        break;
    }
    case NDT_OBJ_FACET: {       /* facet.c:42-83 */
        if (p->n_pos < 3 || p->n_dir < 1) return -1;
        p->edge = dalloc(3 * n); p->unit_edge = dalloc(3 * n); p->axes = dalloc(2 * n);
        for (int i = 0; i < 3; ++i) {
            int j = (i + 1) % 3, k = (i + 2) % 3;
            v_sub(p->pos + j * n, p->pos + i * n, p->edge + i * n, n);
            v_copy(p->unit_edge + i * n, p->edge + i * n, n);
            v_unitize(p->unit_edge + i * n, n);
            p->angle[i] = v_angle3(p->pos + k * n, p->pos + i * n, p->pos + j * n, n);
        }
        /* vectNd_orthogonalize(edge0, edge1, basis0, basis1), vectNd.c:35 */
        double tmp[ND];
        v_proj(p->edge, p->edge + n, tmp, n);
        v_sub(p->edge, tmp, p->axes, n);
        v_copy(p->axes + n, p->edge + n, n);
        v_unitize(p->axes, n);
        v_unitize(p->axes + n, n);
        break;
    }
    default:
        return -1;
    }
    return 0;
}

static void free_scene(pscene *S)
{
    if (!S->objs) return;
    for (int i = 0; i < S->n_objs; ++i) {
        pobj *p = &S->objs[i];
        free(p->axes); free(p->lengths); free(p->AdA); free(p->BdA);
        free(p->edge); free(p->unit_edge); free(p->edge_perp);
    }
    free(S->objs);
    S->objs = NULL;
}

static int prepare_scene(const ndt_flat_scene *fs, pscene *S)
{
    memset(S, 0, sizeof(*S));
    if (!fs || fs->abi_version != NDT_HIP_ABI_VERSION) return NDT_E_INVALID;
    if (fs->dims < 3 || fs->dims > ND) return NDT_E_INVALID;
    S->n = fs->dims;
    S->fs = fs;
    S->n_objs = fs->n_objects;
    S->n_items = fs->n_items;
    S->objs = (pobj *)calloc((size_t)(fs->n_objects > 0 ? fs->n_objects : 1), sizeof(pobj));
    for (int i = 0; i < fs->n_objects; ++i) {
        if (prepare_object(fs, i, &S->objs[i]) != 0) {
            free_scene(S);
            return NDT_E_INVALID;
        }
    }
    S->cam_pos = fs->vecs + fs->cam_pos_off;
    S->cam_img_orig = fs->vecs + fs->cam_img_orig_off;
    S->cam_dir_y = fs->vecs + fs->cam_dir_y_off;
    v_copy(S->cam_dir_x, fs->vecs + fs->cam_dir_x_off, S->n);
    S->cam_left_eye = fs->cam_left_eye_off >= 0 ? fs->vecs + fs->cam_left_eye_off : NULL;
    S->cam_right_eye = fs->cam_right_eye_off >= 0 ? fs->vecs + fs->cam_right_eye_off : NULL;
    S->cam_local_x = fs->cam_local_x_off >= 0 ? fs->vecs + fs->cam_local_x_off : NULL;
    S->cam_local_y = fs->cam_local_y_off >= 0 ? fs->vecs + fs->cam_local_y_off : NULL;
    S->cam_local_z = fs->cam_local_z_off >= 0 ? fs->vecs + fs->cam_local_z_off : NULL;
    S->bb_lower = fs->vecs + fs->bb_lower_off;
    S->bb_upper = fs->vecs + fs->bb_upper_off;
    return NDT_OK;
}

/* ---------------------------------------------------------------- object plugins */

static int trace_list(const pscene *S, const double *o, const double *v, const int *objs, int cnt,
                      unsigned char *mask, double *hit, double *hit_normal, int *ptr, double *t_ptr,
                      double dist_limit);

/* sphere.c:57-112 */
static int isect_sphere(const pobj *p, const double *o, const double *v, double *res, double *normal, int n)
{
    const double *center = p->pos;
    double *oc = res;
    v_sub(o, center, oc, n);
    double oc_len2 = v_dot(oc, oc, n);
    double voc = v_dot(v, oc, n);
    double desc = (voc * voc) - oc_len2 + p->r2;
    if (desc < 0.0)
        return 0;
    double desc_root = sqrt(desc);
    double d = -(voc + desc_root);
    if (d < EPS) {
        d = desc_root - voc;
        if (d < EPS) {
            v_zero(res, n);
            v_zero(normal, n);
            return 0;
        }
    }
    v_scale(v, d, res, n);
    v_add(o, res, res, n);
    v_sub(res, center, normal, n);
    return 1;
}

/* hplane.c:39-75 */
static int isect_hplane(const pobj *p, const double *o, const double *v, double *res, double *normal, int n)
{
    double pl[ND];
    double d = -1;
    v_copy(normal, p->dir, n);
    v_sub(p->pos, o, pl, n);
    double pln = v_dot(pl, normal, n);
    double ln = v_dot(v, normal, n);
    if (ln > EPS || ln < -EPS)
        d = pln / ln;
    if (d >= EPS) {
        v_copy(res, o, n);
        v_scale(v, d, pl, n);
        v_add(res, pl, res, n);
    }
    if (d < EPS)
        return 0;
    return 1;
}

/* hdisk.c:61-85 */
static int isect_hdisk(const pobj *p, const double *o, const double *v, double *res, double *normal, int n)
{
    int ret = isect_hplane(p, o, v, res, normal, n);
    if (ret == 0)
        return ret;
    double dist = v_dist(res, p->pos, n);
    if (dist > p->size[0] || dist < 0)
        ret = 0;
    return ret;
}

/* cylinder.c:88-102 */
static int cyl_between_ends(const pobj *p, const double *point, int n)
{
    if (p->inf_ends)
        return 1;
    double Bc[ND];
    v_sub(point, p->pos, Bc, n);
    double scale = v_dot(Bc, p->axes, n);
    if (scale > 0 && scale < p->lengths[0])
        return 1;
    return 0;
}

/* cylinder.c:104-210 */
static int isect_cylinder(const pobj *p, const double *o, const double *v, double *res, double *normal, int n)
{
    int ret = 0;
    const double *Be = p->pos;
    const double *A = p->axes;
    double size0 = p->size[0];
    double sA[ND], X[ND], Y[ND], tmp[ND];
    double AdA = p->AdA[0], BdA = p->BdA[0];
    double VdA = v_dot(v, A, n);
    double OdA = v_dot(o, A, n);
    double Vaaa = VdA / AdA;
    double BOaa = (BdA - OdA) / AdA;
    v_scale(A, Vaaa, sA, n);
    v_sub(v, sA, Y, n);
    v_sub(o, Be, tmp, n);
    v_scale(A, BOaa, sA, n);
    v_add(tmp, sA, X, n);
    double qa = v_dot(Y, Y, n);
    double qb = v_dot(Y, X, n);
    qb *= 2;
    double qc = v_dot(X, X, n);
    qc -= size0 * size0;
    double det = qb * qb - 4 * qa * qc;
    if (det <= 0)
        return 0;
    double detRoot = sqrt(det);
    double t1 = (-qb + detRoot) / (2 * qa);
    double t2 = (-qb - detRoot) / (2 * qa);
    if (t2 > EPS) {
        v_scale(v, t2, sA, n);
        v_add(o, sA, res, n);
        if (cyl_between_ends(p, res, n))
            ret = 1;
    }
    if (ret == 0 && t1 > EPS) {
        v_scale(v, t1, sA, n);
        v_add(o, sA, res, n);
        if (cyl_between_ends(p, res, n))
            ret = 1;
    }
    if (ret != 0) {
        v_sub(res, Be, X, n);
        double nCdA = v_dot(A, X, n);
        v_scale(A, nCdA / AdA, Y, n);
        v_sub(X, Y, normal, n);
    }
    return ret;
}

/* hcylinder.c:101-130 and orthotope.c:122-148: per-axis extent test, identical in form */
static int within_axes(const pobj *p, const double *point, int n)
{
    double Bc[ND];
    v_sub(point, p->pos, Bc, n);
    for (int i = 0; i < p->m; ++i) {
        double scale = v_dot(Bc, p->axes + i * n, n);
        scale = scale / p->AdA[i];
        if (scale < -EPS || scale > p->lengths[i] + EPS)
            return 0;
    }
    return 1;
}

/* common head of hcylinder.c:159-185 / orthotope.c:175-199: builds P, Q and the quadratic */
static void axes_quadratic(const pobj *p, const double *o, const double *v, double *P, double *Q,
                           double *qa, double *qb, double *qc, int n)
{
    double sA[ND], sum_A[ND];
    v_zero(sum_A, n);
    for (int i = 0; i < p->m; ++i) {
        const double *ax = p->axes + i * n;
        double AdA = p->AdA[i];
        double VdA = v_dot(v, ax, n);
        v_scale(ax, VdA / AdA, sA, n);
        v_add(sum_A, sA, sum_A, n);
    }
    v_sub(sum_A, v, P, n);
    v_zero(sum_A, n);
    for (int i = 0; i < p->m; ++i) {
        const double *ax = p->axes + i * n;
        double BdA = p->BdA[i];
        double AdA = p->AdA[i];
        double OdA = v_dot(o, ax, n);
        v_scale(ax, (OdA - BdA) / AdA, sA, n);
        v_add(sum_A, sA, sum_A, n);
    }
    v_sub(p->pos, o, Q, n);
    v_add(Q, sum_A, Q, n);
    *qa = v_dot(P, P, n);
    *qb = v_dot(P, Q, n);
    *qb *= 2;
    *qc = v_dot(Q, Q, n);
}

/* common tail of hcylinder.c:222-237 / orthotope.c:280-295: residual normal */
static void axes_normal(const pobj *p, const double *res, double *normal, int n)
{
    double P[ND], Q[ND], sA[ND];
    v_sub(res, p->pos, P, n);
    v_zero(Q, n);
    for (int i = 0; i < p->m; ++i) {
        v_proj(P, p->axes + i * n, sA, n);
        v_add(Q, sA, Q, n);
    }
    v_sub(P, Q, normal, n);
}

/* hcylinder.c:132-244 */
static int isect_hcylinder(const pobj *p, const double *o, const double *v, double *res, double *normal, int n)
{
    int ret = 0;
    double P[ND], Q[ND], sA[ND];
    double qa, qb, qc;
    double radius = p->size[0];
    axes_quadratic(p, o, v, P, Q, &qa, &qb, &qc, n);
    qc -= radius * radius;
    double det = qb * qb - 4 * qa * qc;
    if (det < 0.0)
        return 0;
    double detRoot = sqrt(det);
    double t1 = (-qb + detRoot) / (2 * qa);
    double t2 = (-qb - detRoot) / (2 * qa);
    if (t2 > EPS) {
        v_scale(v, t2, sA, n);
        v_add(o, sA, res, n);
        if (p->inf_ends || within_axes(p, res, n))
            ret = 1;
    }
    if (ret == 0 && t1 > EPS) {
        v_scale(v, t1, sA, n);
        v_add(o, sA, res, n);
        if (p->inf_ends || within_axes(p, res, n))
            ret = 1;
    }
    if (ret != 0)
        axes_normal(p, res, normal, n);
    return ret;
}

/* orthotope.c:150-302 */
static int isect_orthotope(const pobj *p, const double *o, const double *v, double *res, double *normal, int n)
{
    int ret = 0;
    double P[ND], Q[ND], sA[ND];
    double qa, qb, qc;
    axes_quadratic(p, o, v, P, Q, &qa, &qb, &qc, n);
    qc -= EPS;
    double det = qb * qb - 4 * qa * qc;
    if (det >= 0.0 && fabs(qa) > EPS) {
        double detRoot = sqrt(det);
        double half_inv_qa = 0.5 / qa;
        double t1 = (-qb + detRoot) * half_inv_qa;
        double t2 = (-qb - detRoot) * half_inv_qa;
        if (t2 > EPS) {
            v_scale(v, t2, sA, n);
            v_add(o, sA, res, n);
            if (within_axes(p, res, n))
                ret = 1;
        }
        if (ret == 0 && t1 > EPS) {
            v_scale(v, t1, sA, n);
            v_add(o, sA, res, n);
            if (within_axes(p, res, n))
                ret = 1;
        }
    }
    if (ret == 0) {
        double t = -1.0;
        if (fabs(qa) < EPS) {
            if (fabs(qb) < EPS)     /* sic: orthotope.c:238-241 */
                t = -qc / qb;
            else
                t = -1.0;
        } else {
            t = -qb / (2 * qa);
        }
        if (t < EPS)
            return 0;
        double dist = qa * t * t + qb * t + qc;
        if (fabs(dist) > EPS)
            return 0;
        v_scale(v, t, sA, n);
        v_add(o, sA, res, n);
        if (within_axes(p, res, n))
            ret = 1;
    }
    if (ret != 0)
        axes_normal(p, res, normal, n);
    return ret;
}

/* hcube.c:236-250: nested linear trace over the faces, no mask, no limit */
static int isect_hcube(const pscene *S, const pobj *p, const double *o, const double *v, double *res, double *normal)
{
    int sub = -1;
    return trace_list(S, o, v, p->child, p->n_child, NULL, res, normal, &sub, NULL, -1.0);
}

/* hfacet.c:156-199 */
static void hfacet_barycentric(const pobj *p, const double *point, double *coords, int n)
{
    double C[ND];
    const double *A = p->unit_edge;
    const double *B = p->edge_perp;
    double x1 = 0, y1 = 0;
    v_sub(point, p->pos, C, n);
    double xp = v_dot(A, C, n);
    double yp = v_dot(B, C, n);
    double x2 = v_dot(A, p->edge, n);
    double y2 = v_dot(B, p->edge, n);
    double x3 = v_dot(A, p->edge + 2 * n, n);
    double y3 = v_dot(B, p->edge + 2 * n, n);
    double l1 = ((y2 - y3) * (xp - x3) + (x3 - x2) * (yp - y3)) / ((y2 - y3) * (x1 - x3) + (x3 - x2) * (y1 - y3));
    double l2 = ((y3 - y1) * (xp - x3) + (x1 - x3) * (yp - y3)) / ((y2 - y3) * (x1 - x3) + (x3 - x2) * (y1 - y3));
    double l3 = 1 - l1 - l2;
    coords[0] = l1;
    coords[1] = l2;
    coords[2] = l3;
}

/* sum of components in the SSE lane order: vectNd_dot(R, ones) with ones = fill(1.0), hfacet.c:50-54,241 */
static double v_sum(const double *a, int n)
{
    double s0 = a[0] * 1.0;
    double s1 = a[1] * 1.0;
    int k = (n + 1) >> 1;
    for (int i = 1; i < k; ++i) {
        s0 = s0 + a[2 * i] * 1.0;
        if (2 * i + 1 < n)
            s1 = s1 + a[2 * i + 1] * 1.0;
    }
    return s0 + s1;
}

/* hfacet.c:211-310 */
static int isect_hfacet(const pobj *p, const double *o, const double *v, double *res, double *normal, int n)
{
    const double *unit_edge0 = p->unit_edge;
    const double *edge_perp = p->edge_perp;
    int ret = 0;
    double R[ND], vE0[ND], vE2[ND], Q[ND], oP0[ND];
    v_proj_unit(v, unit_edge0, vE0, n);
    v_proj_unit(v, edge_perp, vE2, n);
    v_add(vE0, vE2, R, n);
    v_sub(R, v, R, n);
    double Rv = v_sum(R, n);
    if (fabs(Rv) < EPS)
        return 0;
    v_sub(o, p->pos, oP0, n);
    v_proj_unit(oP0, unit_edge0, vE0, n);
    v_proj_unit(oP0, edge_perp, vE2, n);
    v_add(vE0, vE2, Q, n);
    v_sub(Q, oP0, Q, n);
    double Qv = v_sum(Q, n);
    double t = -Qv / Rv;
    double lambda[3];
    if (t > EPS) {
        v_scale(v, t, res, n);
        v_add(o, res, res, n);
        hfacet_barycentric(p, res, lambda, n);
        ret = 1;
        for (int i = 0; i < 3; ++i)
            if (lambda[i] < -EPS || lambda[i] > 1 + EPS) { ret = 0; break; }
    }
    if (ret != 0) {
        if (p->flag[0]) {
            v_zero(normal, n);
            for (int i = 0; i < 3; ++i) {
                v_scale(p->dir + i * n, lambda[i], R, n);
                v_add(normal, R, normal, n);
            }
        } else {
            /* hfacet_point_in_plane, hfacet.c:119-144 */
            double D[ND], U[ND], V[ND];
            v_sub(o, p->pos, D, n);
            v_proj_unit(D, unit_edge0, U, n);
            v_proj_unit(D, edge_perp, V, n);
            v_add(U, V, R, n);
            v_add(R, p->pos, R, n);
            v_sub(o, R, normal, n);
            v_unitize(normal, n);
        }
    }
    return ret;
}

/* facet.c:166-269 */
static int isect_facet(const pobj *p, const double *o, const double *v, double *res, double *normal, int n)
{
    int ret = 0;
    const double *pos1 = p->pos + n;
    double P[ND], sA[ND], sum_A[ND], Q[ND];
    v_zero(sum_A, n);
    for (int i = 0; i < 2; ++i) {
        const double *b = p->axes + i * n;
        double VdA = v_dot(v, b, n);
        double AdA = v_dot(b, b, n);
        v_scale(b, VdA / AdA, sA, n);
        v_add(sum_A, sA, sum_A, n);
    }
    v_sub(sum_A, v, P, n);
    v_zero(sum_A, n);
    for (int i = 0; i < 2; ++i) {
        const double *b = p->axes + i * n;
        double OdA = v_dot(o, b, n);
        double BdA = v_dot(pos1, b, n);
        double AdA = v_dot(b, b, n);
        v_scale(b, (OdA - BdA) / AdA, sA, n);
        v_add(sum_A, sA, sum_A, n);
    }
    v_sub(pos1, o, Q, n);
    v_add(Q, sum_A, Q, n);
    double qa = v_dot(P, P, n);
    double qb = v_dot(P, Q, n);
    qb *= 2;
    double qc = v_dot(Q, Q, n);
    double t = -1.0;
    if (fabs(qa) < EPS) {
        if (fabs(qb) < EPS)
            t = -qc / qb;
        else
            t = -1.0;
    } else {
        t = -qb / (2 * qa);
    }
    if (t < EPS)
        return 0;
    double dist = qa * t * t + qb * t + qc;
    if (fabs(dist) > EPS)
        return 0;
    v_scale(v, t, sA, n);
    v_add(o, sA, res, n);
    /* inside_edges, facet.c:148-164 */
    ret = 1;
    for (int i = 0; i < 3; ++i) {
        int j = (i + 1) % 3;
        double angle = v_angle3(res, p->pos + i * n, p->pos + j * n, n);
        if (angle > p->angle[i]) { ret = 0; break; }
    }
    v_copy(normal, p->dir, n);
    return ret;
}

/* bounding.c:34-85 */
static int bsphere_gate(const pobj *p, const double *o, const double *v, double min_dist, int n)
{
    double oc[ND];
    v_sub(o, p->bcenter, oc, n);
    double oc_len2 = v_dot(oc, oc, n);
    if (min_dist > 0) {
        double min_dist_r = min_dist + p->bradius;
        if (oc_len2 > min_dist_r * min_dist_r)
            return 0;
    }
    double voc = v_dot(v, oc, n);
    double voc2 = voc * voc;
    double desc = voc2 - oc_len2 + p->bradius2;
    if (desc < 0.0 || (voc > 0.0 && voc2 > desc))
        return 0;
    return 1;
}

/* vect_object_intersect, object.c:605-630.  *ptr receives the object that owns the material. */
static int object_intersect(const pscene *S, int idx, const double *o, const double *v, double *res, double *normal,
                            int *ptr, double min_dist)
{
    const pobj *p = &S->objs[idx];
    int n = S->n;
    int ret = 0;
    if (p->bradius > 0) {
        if (bsphere_gate(p, o, v, min_dist, n) <= 0) {
            *ptr = -1;
            return 0;
        }
    }
    switch (p->type) {
    case NDT_OBJ_SPHERE:    ret = isect_sphere(p, o, v, res, normal, n); break;
    case NDT_OBJ_HPLANE:    ret = isect_hplane(p, o, v, res, normal, n); break;
    case NDT_OBJ_HDISK:     ret = isect_hdisk(p, o, v, res, normal, n); break;
    case NDT_OBJ_CYLINDER:  ret = isect_cylinder(p, o, v, res, normal, n); break;
    case NDT_OBJ_HCYLINDER: ret = isect_hcylinder(p, o, v, res, normal, n); break;
    case NDT_OBJ_ORTHOTOPE: ret = isect_orthotope(p, o, v, res, normal, n); break;
    case NDT_OBJ_HCUBE:     ret = isect_hcube(S, p, o, v, res, normal); break;
    case NDT_OBJ_HFACET:    ret = isect_hfacet(p, o, v, res, normal, n); break;
    case NDT_OBJ_FACET:     ret = isect_facet(p, o, v, res, normal, n); break;
    default: ret = 0;
    }
    if (ret)
        *ptr = idx;
    return ret;
}

/* ---------------------------------------------------------------- trace / kd-tree */

/* trace, object.c:692-747 */
static int trace_list(const pscene *S, const double *o, const double *v, const int *objs, int cnt,
                      unsigned char *mask, double *hit, double *hit_normal, int *ptr, double *t_ptr,
                      double dist_limit)
{
    int n = S->n;
    double min_dist = -1;
    double res[ND], normal[ND];
    if (ptr != NULL)
        *ptr = -1;
    for (int i = 0; i < cnt; ++i) {
        int id = objs[i];
        if (mask) {
            if (mask[id] != 0)
                continue;
            mask[id] = 1;
        }
        double dist = -1;
        int tmp_ptr = -1;
        int ret = object_intersect(S, id, o, v, res, normal, &tmp_ptr, min_dist);
        if (ret > 0) {
            dist = v_dist(o, res, n);
            if (dist > EPS && (dist + EPS < min_dist || min_dist < 0)) {
                min_dist = dist;
                v_copy(hit, res, n);
                v_copy(hit_normal, normal, n);
                if (ptr != NULL)
                    *ptr = tmp_ptr;
            }
            if (dist_limit == 0.0 || dist < dist_limit)
                break;
        }
    }
    if (t_ptr != NULL && min_dist > EPS)
        *t_ptr = min_dist;
    if (min_dist < 0)
        return 0;
    return 1;
}

/* aabb_intersect, kd-tree.c:84-127 */
static int aabb_intersect(const pscene *S, const double *o, const double *v, double *tl_ptr, double *tu_ptr)
{
    double tl = -DBL_MAX, tu = DBL_MAX;
    for (int i = 0; i < S->n; ++i) {
        double v_i = v[i], o_i = o[i];
        if (fabs(v_i) < EPS2)
            continue;
        double tl_i = (S->bb_lower[i] - o_i) / v_i;
        double tu_i = (S->bb_upper[i] - o_i) / v_i;
        if (tl_i > tu_i) {
            double tmp = tl_i;
            tl_i = tu_i;
            tu_i = tmp;
        }
        if (tl_i > tl) tl = tl_i;
        if (tu_i < tu) tu = tu_i;
        if (tu < -EPS)
            return 0;
    }
    tl -= EPS;
    tu += EPS;
    *tl_ptr = tl;
    *tu_ptr = tu;
    return (tu >= -EPS) && (tl <= tu);
}

/* kd_node_intersect, kd-tree.c:482-568 */
static int kd_node_intersect(const pscene *S, int node_idx, const double *o, const double *v, const double *v_inv,
                             double *hit, double *hit_normal, unsigned char *mask, int *ptr, double *t_ptr,
                             double dist_limit, double tl, double tu)
{
    if (node_idx < 0)
        return 0;
    if (tu < 0.0)
        return 0;
    const ndt_flat_kdnode *node = &S->fs->kd_nodes[node_idx];
    int n = S->n;
    int num = node->num;
    int node_dim = node->dim;
    int ret = 0;
    if (num > 0) {
        double t = 0;
        int obj_ptr = -1;
        double lhit[ND], lhit_normal[ND];
        ret = trace_list(S, o, v, S->fs->leaf_refs + node->first, num, mask, lhit, lhit_normal, &obj_ptr, &t, dist_limit);
        if (ret && t < *t_ptr) {
            *t_ptr = t;
            *ptr = obj_ptr;
            v_copy(hit, lhit, n);
            v_copy(hit_normal, lhit_normal, n);
        }
        if (node_dim < 0)
            return ret;
    }
    if (node_dim < 0)
        return ret;     /* an empty leaf cannot be built (kd-tree.c:407), but never index v_inv[-1] */
    double node_boundary = node->boundary;
    int near = node->left, far = node->right;
    double v_inv_i = v_inv[node_dim];
    double o_i = o[node_dim];
    if (v_inv_i < EPS2) {
        int tmp = near;
        near = far;
        far = tmp;
    }
    if (-INV_EPS2 <= v_inv_i && v_inv_i <= INV_EPS2) {
        double tp = (node_boundary - o_i) * v_inv_i;
        if (tu < tp - EPS && *t_ptr > tl) {
            ret |= kd_node_intersect(S, near, o, v, v_inv, hit, hit_normal, mask, ptr, t_ptr, dist_limit, tl, tu);
        } else if (tl > tp + EPS && *t_ptr > tl) {
            ret |= kd_node_intersect(S, far, o, v, v_inv, hit, hit_normal, mask, ptr, t_ptr, dist_limit, tl, tu);
        } else {
            if (*t_ptr > tl)
                ret |= kd_node_intersect(S, near, o, v, v_inv, hit, hit_normal, mask, ptr, t_ptr, dist_limit, tl, tp + EPS);
            if (*t_ptr > tp)
                ret |= kd_node_intersect(S, far, o, v, v_inv, hit, hit_normal, mask, ptr, t_ptr, dist_limit, tp - EPS, tu);
        }
    } else {
        if (o_i < node_boundary + EPS && *t_ptr > tl)
            ret |= kd_node_intersect(S, near, o, v, v_inv, hit, hit_normal, mask, ptr, t_ptr, dist_limit, tl, tu);
        if (o_i > node_boundary - EPS && *t_ptr > tl)
            ret |= kd_node_intersect(S, far, o, v, v_inv, hit, hit_normal, mask, ptr, t_ptr, dist_limit, tl, tu);
    }
    return ret;
}

/* trace_kd (object.c:683) = kd_tree_intersect, kd-tree.c:570-625.
 * `mask` is caller-provided scratch of n_items bytes (the reference callocs it per ray). */
static int trace_kd(const pscene *S, const double *o, const double *v, double *hit, double *hit_normal, int *ptr,
                    double dist_limit, unsigned char *mask)
{
    int n = S->n;
    int ret = 0;
    double v_inv[ND];
    for (int i = 0; i < n; ++i) {
        double v_i = v[i], v_inv_i;
        if (v_i < EPS2 && v_i >= 0.0)
            v_inv_i = INV_EPS2;
        else if (v_i > -EPS2 && v_i <= 0.0)
            v_inv_i = -INV_EPS2;
        else
            v_inv_i = 1.0 / v_i;
        v_inv[i] = v_inv_i;
    }
    double t = DBL_MAX;
    ret = trace_list(S, o, v, S->fs->inf_refs, S->fs->n_inf, NULL, hit, hit_normal, ptr, &t, dist_limit);
    double tl, tu;
    if (aabb_intersect(S, o, v, &tl, &tu)) {
        double lt = DBL_MAX;
        int obj_ptr = -1;
        double lhit[ND], lhit_normal[ND];
        memset(mask, 0, (size_t)S->n_items);
        int lret = kd_node_intersect(S, S->fs->n_kd_nodes > 0 ? 0 : -1, o, v, v_inv, lhit, lhit_normal, mask, &obj_ptr,
                                     &lt, dist_limit, tl, tu);
        if (lret) {
            if (!ret || (lt > EPS && lt + EPS < t)) {
                v_copy(hit, lhit, n);
                v_copy(hit_normal, lhit_normal, n);
                *ptr = obj_ptr;
                ret |= lret;
            }
        }
    }
    return ret;
}

/* ---------------------------------------------------------------- shading */

typedef struct {
    long long primary, secondary, shadow;
} ray_counts;

typedef struct {
    const pscene *S;
    unsigned char *mask;
    ray_counts cnt;
    int specular;
    int samples;            /* `-n`: > 1 = jittered samples + lens sampling with drand48 (ndt.c:505-542) */
    int stochastic;         /* samples > 1 or area lights: every pass of the adaptive loop is a different ray tree */
    int aa_lens;            /* recursive anti-aliasing with an aperture: every sample draws a lens point (ndt.c:528) */
    double pix_w, pix_h;    /* 1/width, 1/height of the image being rendered (ndt.c:482-483) */
} tctx;

typedef struct { double r, g, b, a; } pix;

/* apply_lights, ndt.c:71-326 */
static void apply_lights(tctx *T, int obj_idx, const double *src, const double *look, const double *hit,
                         const double *hit_normal, pix *color)
{
    const pscene *S = T->S;
    const ndt_flat_scene *fs = S->fs;
    const pobj *obj = &S->objs[obj_idx];
    int n = S->n;
    double hit_r = obj->red, hit_g = obj->green, hit_b = obj->blue;
    double hitr_r = 0.0, hitr_g = 0.0, hitr_b = 0.0;
    if (T->specular) {
        hitr_r = obj->red_r; hitr_g = obj->green_r; hitr_b = obj->blue_r;
    }
    pix clr;
    clr.r = hit_r * fs->ambient[0];
    clr.g = hit_g * fs->ambient[1];
    clr.b = hit_b * fs->ambient[2];
    clr.a = 1.0;
    double rev_view[ND], rev_light[ND], light_vec[ND], light_hit[ND], light_hit_normal[ND], lgt_pos[ND], near_pos[ND];
    v_zero(rev_light, n); v_zero(light_vec, n); v_zero(light_hit, n); v_zero(light_hit_normal, n);
    for (int i = 0; i < fs->n_lights; ++i) {
        const ndt_flat_light *L = &fs->lights[i];
        int lgt_type = L->type;
        const double *Lpos = (L->pos_off >= 0) ? fs->vecs + L->pos_off : NULL;
        const double *Ldir = (L->dir_off >= 0) ? fs->vecs + L->dir_off : NULL;
        if (lgt_type == NDT_LIGHT_AMBIENT) {
            clr.r += hit_r * L->red;
            clr.g += hit_g * L->green;
            clr.b += hit_b * L->blue;
            continue;
        }
        if (Lpos) v_copy(lgt_pos, Lpos, n);
        if (lgt_type == NDT_LIGHT_DISK || lgt_type == NDT_LIGHT_RECT) {
            /* a random point of the areal light (ndt.c:116-147), from the global drand48 stream; from
             * here on the sample is a point light */
            const double *u1 = fs->vecs + L->area_off, *v1 = u1 + n;
            double x, y, temp[ND];
            do {
                x = 2 * drand48() - 1.0;
                y = 2 * drand48() - 1.0;
            } while (lgt_type == NDT_LIGHT_DISK && x * x + y * y > 1.0);
            v_scale(u1, x * L->radius, temp, n);
            v_add(lgt_pos, temp, lgt_pos, n);
            v_scale(v1, y * L->radius, temp, n);
            v_add(lgt_pos, temp, lgt_pos, n);
            lgt_type = NDT_LIGHT_POINT;
        }
        if (lgt_type != NDT_LIGHT_POINT && lgt_type != NDT_LIGHT_DIRECTIONAL && lgt_type != NDT_LIGHT_SPOT)
            continue;

        if (lgt_type == NDT_LIGHT_POINT || lgt_type == NDT_LIGHT_SPOT)
            v_sub(lgt_pos, hit, rev_light, n);
        else
            v_scale(Ldir, -1, rev_light, n);
        v_unitize(rev_light, n);
        v_sub(src, hit, rev_view, n);
        double dotRev1 = v_dot(rev_light, hit_normal, n);
        double dotRev2 = v_dot(rev_view, hit_normal, n);
        if ((dotRev1 * dotRev2) <= 0)
            continue;

        int light_obj_ptr = -1;
        int got_hit = 0;
        double dist_limit = -1.0;
        if (lgt_type == NDT_LIGHT_DIRECTIONAL) {
            dist_limit = 0.0;
        } else {
            dist_limit = v_dist(hit, lgt_pos, n);
            dist_limit += EPS;
        }
        double ldist2 = 1.0;
        if (lgt_type == NDT_LIGHT_POINT || lgt_type == NDT_LIGHT_SPOT) {
            v_sub(hit, lgt_pos, light_vec, n);
            ldist2 = v_dot(light_vec, light_vec, n);
            v_unitize(light_vec, n);
            if (lgt_type == NDT_LIGHT_SPOT) {
                double angle = v_angle(Ldir, light_vec, n);
                if ((angle * 180.0 / M_PI) > L->angle)
                    continue;
            }
            T->cnt.shadow++;
            got_hit = trace_kd(S, lgt_pos, light_vec, light_hit, light_hit_normal, &light_obj_ptr, dist_limit, T->mask);
            if (!got_hit || light_obj_ptr != obj_idx)
                continue;
            double dist = v_dist(hit, light_hit, n);
            if (dist > EPS)
                continue;
        } else {
            v_copy(near_pos, Ldir, n);
            v_unitize(near_pos, n);
            v_scale(near_pos, -EPS, near_pos, n);
            v_add(near_pos, hit, near_pos, n);
            v_scale(Ldir, -1.0, light_vec, n);
            T->cnt.shadow++;
            got_hit = trace_kd(S, near_pos, rev_light, light_hit, light_hit_normal, &light_obj_ptr, 0.0, T->mask);
            if (got_hit)
                continue;
            v_copy(light_vec, Ldir, n);
            v_copy(light_hit, hit, n);
            v_copy(light_hit_normal, hit_normal, n);
            light_obj_ptr = obj_idx;
            ldist2 = 1;
        }

        double angle = v_angle(hit_normal, light_vec, n);
        if (angle > M_PI / 2.0)
            angle = M_PI - angle;
        double light_scale = cos(angle) / ldist2;
        if (!obj->transparent) {
            clr.r += hit_r * L->red * light_scale;
            clr.g += hit_g * L->green * light_scale;
            clr.b += hit_b * L->blue * light_scale;
        }

        if (T->specular) {
            double light_ref[ND], rev_look[ND];
            v_reflect(light_vec, light_hit_normal, light_ref, 0.5, n);
            v_unitize(light_ref, n);
            v_scale(look, -1, rev_look, n);
            v_unitize(rev_look, n);
            double rv = v_dot(light_ref, rev_look, n);
            rv = MAXV(0, rv);
            double rvn = pow(rv, 50);
            double max_light = MAXV(L->red, MAXV(L->green, L->blue));
            clr.r += hitr_r * L->red / max_light * rvn;
            clr.g += hitr_g * L->green / max_light * rvn;
            clr.b += hitr_b * L->blue / max_light * rvn;
            clr.a = 1.0;
        }
    }
    *color = clr;
}

/* get_ray_color, ndt.c:329-450.  `primary` only selects the counter. */
static int get_ray_color_d(tctx *T, const double *src, const double *unit_look, pix *pixel, double pixel_frac,
                           int max_depth, int primary, double *depth);
static int get_ray_color(tctx *T, const double *src, const double *unit_look, pix *pixel, double pixel_frac,
                         int max_depth, int primary)
{
    return get_ray_color_d(T, src, unit_look, pixel, pixel_frac, max_depth, primary, NULL);
}

/* `depth` (primaries of a depth-map render only): 1/distance of the hit, 0 on a miss; left as it is
 * when the hit lies within EPSILON of the ray origin (ndt.c:362-373) */
static int get_ray_color_d(tctx *T, const double *src, const double *unit_look, pix *pixel, double pixel_frac,
                           int max_depth, int primary, double *depth)
{
    const pscene *S = T->S;
    const ndt_flat_scene *fs = S->fs;
    int n = S->n;
    int ret = 0;
    pixel->r = pixel->g = pixel->b = 0.0;
    pixel->a = 1.0;
    if (pixel_frac < (1.0 / 512.0))
        return 1;
    if (max_depth <= 0)
        return 1;
    double hit[ND], hit_normal[ND];
    int obj_ptr = -1;
    pix clr;
    memset(&clr, 0, sizeof(clr));
    v_zero(hit, n);
    v_zero(hit_normal, n);
    if (primary) T->cnt.primary++; else T->cnt.secondary++;
    trace_kd(S, src, unit_look, hit, hit_normal, &obj_ptr, -1.0, T->mask);
    double trace_dist = v_dist(hit, src, n);   /* depth != NULL only for primaries; same value either way */
    if (depth) {
        if (trace_dist > EPS) *depth = 1.0 / trace_dist;
        if (obj_ptr < 0) *depth = 0.0;
    }
    if (obj_ptr >= 0 && trace_dist > EPS) {
        const pobj *obj = &S->objs[obj_ptr];
        apply_lights(T, obj_ptr, src, unit_look, hit, hit_normal, &clr);
        double hitr_r = obj->red_r, hitr_g = obj->green_r, hitr_b = obj->blue_r;
        pix ref;
        double new_ray[ND];
        double contrib = MAXV(hitr_r, MAXV(hitr_g, hitr_b));
        if (contrib > 0) {
            if (hitr_r != 0.0 || hitr_g != 0.0 || hitr_b != 0.0) {
                v_reflect(unit_look, hit_normal, new_ray, 1.0, n);
                v_unitize(new_ray, n);
                get_ray_color(T, hit, new_ray, &ref, contrib * pixel_frac, max_depth - 1, 0);
                if (T->specular) {
                    clr.r = (1 - hitr_r) * (clr.r) + (hitr_r)*ref.r;
                    clr.g = (1 - hitr_g) * (clr.g) + (hitr_g)*ref.g;
                    clr.b = (1 - hitr_b) * (clr.b) + (hitr_b)*ref.b;
                    clr.a = 1.0;
                } else {
                    clr.r += hitr_r * ref.r;
                    clr.g += hitr_g * ref.g;
                    clr.b += hitr_b * ref.b;
                    clr.a = 1.0;
                }
            }
        }
        if (obj->transparent) {
            v_refract(unit_look, hit_normal, new_ray, obj->refract_index, n);
            v_unitize(new_ray, n);
            get_ray_color(T, hit, new_ray, &ref, (1 - contrib) * pixel_frac, max_depth - 1, 0);
            clr.r += (1.0 - hitr_r) * ref.r;
            clr.g += (1.0 - hitr_g) * ref.g;
            clr.b += (1.0 - hitr_b) * ref.b;
            clr.a = 1.0;
        }
        ret = 1;
    } else {
        clr.r = fs->background[0];
        clr.g = fs->background[1];
        clr.b = fs->background[2];
        clr.a = fs->background[3];
        ret = 0;
    }
    *pixel = clr;
    return ret;
}

/* camera_target_point, CAMERA_NORMAL branch, camera.c:557-575 */
/* vectNd_orthogonalize + vectNd_rotate2, vectNd.c:35-57, 271-325 */
static void v_rotate2(const double *v, const double *center, const double *v1, const double *v2, double angle, double *res, int n)
{
    double basisX[ND], basisY[ND], temp[ND], localPos[ND], projX[ND], projY[ND], rotX[ND], rotY[ND];
    v_proj(v1, v2, temp, n);
    v_sub(v1, temp, basisX, n);
    v_copy(basisY, v2, n);
    v_unitize(basisX, n);
    v_unitize(basisY, n);
    v_sub(v, center, localPos, n);
    v_proj(localPos, basisX, projX, n);
    v_proj(localPos, basisY, projY, n);
    double virtX = v_dot(projX, basisX, n);
    double virtY = v_dot(projY, basisY, n);
    v_scale(basisX, virtX * cos(angle) - virtY * sin(angle), rotX, n);
    v_scale(basisY, virtY * cos(angle) + virtX * sin(angle), rotY, n);
    double out[ND];
    v_sub(v, projX, out, n);
    v_sub(out, projY, out, n);
    v_add(out, rotX, out, n);
    v_add(out, rotY, out, n);
    v_copy(res, out, n);
}

/* camera_target_point, camera.c:503-579: planar, spherical (VR) and cylindrical (panorama) screens */
static void camera_target_point(const pscene *S, double x, double y, double dist, double *pixel)
{
    int n = S->n;
    double temp[ND];
    const int type = S->fs->cam_type;
    if (type == 1 || type == 2) {
        double view_x, view_y, view_z;
        if (type == 1) {
            double azi = x * S->fs->cam_h_fov;
            double alt = y * S->fs->cam_v_fov;
            view_x = dist * sin(azi) * cos(alt);
            view_y = dist * sin(alt);
            view_z = dist * cos(azi) * cos(alt);
        } else {
            double azi = x * S->fs->cam_h_fov;
            double y_size = 2.0 * tan(S->fs->cam_v_fov / 2.0) * dist;
            view_x = dist * sin(azi);
            view_y = y * y_size;
            view_z = dist * cos(azi);
        }
        v_copy(pixel, S->cam_pos, n);
        v_scale(S->cam_local_x, view_x, temp, n);
        v_add(pixel, temp, pixel, n);
        v_scale(S->cam_local_y, view_y, temp, n);
        v_add(pixel, temp, pixel, n);
        v_scale(S->cam_local_z, view_z, temp, n);
        v_add(pixel, temp, pixel, n);
        return;
    }
    v_copy(pixel, S->cam_img_orig, n);
    v_scale(S->cam_dir_x, x, temp, n);
    v_add(pixel, temp, pixel, n);
    v_scale(S->cam_dir_y, y, temp, n);
    v_add(pixel, temp, pixel, n);
    double screen_dist = v_dist(S->cam_img_orig, S->cam_pos, n);
    if (screen_dist > EPS) {
        v_sub(pixel, S->cam_pos, temp, n);
        v_scale(temp, dist / screen_dist, temp, n);
        v_add(S->cam_pos, temp, pixel, n);
    }
}

/* get_pixel_color, ndt.c:456-576, samples=1, CAM_CENTER, recursive_aa=0.
 * With samples == 1 every pass of the adaptive loop re-traces the same deterministic ray
 * (SURVEY 8a row A3).  `literal` != 0 re-traces it like the reference does; otherwise the
 * sample is traced once and the loop's arithmetic is replayed on its colour -- identical
 * results, and `*k_out` (the repeat count) scales the ray counters. */
enum { CAM_LEFT = 0, CAM_CENTER = 1, CAM_RIGHT = 2 };      /* camera_mode, ndt.c:452-454 */

static void get_pixel_color_m(tctx *T, double x, double y, pix *clr, int max_optic_depth, int literal, int *k_out,
                              int mode, double *depth);
static void get_pixel_color(tctx *T, double x, double y, pix *clr, int max_optic_depth, int literal, int *k_out)
{
    get_pixel_color_m(T, x, y, clr, max_optic_depth, literal, k_out, CAM_CENTER, NULL);
}

static void get_pixel_color_m(tctx *T, double x, double y, pix *clr, int max_optic_depth, int literal, int *k_out,
                              int mode, double *depth)
{
    const pscene *S = T->S;
    int n = S->n;
    double look[ND], pixel[ND], virtCam[ND];
    const int samples = T->samples > 1 ? T->samples : 1;
    if (T->stochastic) literal = 1;            /* every sample is a different ray tree */
    const double orig_x = x, orig_y = y;
    int min_samples = samples;
    int max_samples = 10000;
    double max_diff = 1.0 / 256.0;
    double clr_diff = 256;
    pix t_clr = { 0.0, 0.0, 0.0, 0.0 };
    int t_samples = 0;
    pix l_clr = { 0, 0, 0, 1 };
    ray_counts before = T->cnt, one = { 0, 0, 0 };
    for (int i = 0; i < min_samples || (i < max_samples && clr_diff > max_diff); ++i) {
        if (i == 0 || literal) {
            v_copy(virtCam, mode == CAM_LEFT ? S->cam_left_eye : mode == CAM_RIGHT ? S->cam_right_eye : S->cam_pos, n);   /* ndt.c:491-502 */
            if (samples > 1) {
                /* jitter inside the pixel (ndt.c:505-514); the reference's global drand48 stream */
                double dx = drand48();
                double dy = drand48();
                x = orig_x + dx * T->pix_w;
                y = orig_y + dy * T->pix_h;
            }
            camera_target_point(S, x, y, S->fs->cam_focal_distance, pixel);
            if ((S->fs->cam_type == 1 || S->fs->cam_type == 2) && mode != CAM_CENTER) {
                /* VR: the eye goes round the centre with the view direction (ndt.c:519-525) */
                double azi = x * S->fs->cam_h_fov;
                v_rotate2(virtCam, S->cam_pos, S->cam_local_x, S->cam_local_z, azi, virtCam, n);
            }
            if (samples > 1 || T->aa_lens) {
                /* lens sample for depth of field (ndt.c:527-542: `recursive_aa != 0 || samples > 1`): rejection in the unit
                 * disk, also drawn when the aperture is 0 (the offsets are then zero vectors; in the anti-aliasing mode the
                 * draws are then skipped here -- nothing else reads the stream) */
                double ax, ay;
                do {
                    ax = 2 * drand48() - 1.0;
                    ay = 2 * drand48() - 1.0;
                } while (ax * ax + ay * ay > 1.0);
                if (S->cam_local_x && S->cam_local_y) {
                    double temp[ND];
                    v_scale(S->cam_local_x, ax * S->fs->cam_aperture_radius, temp, n);
                    v_add(virtCam, temp, virtCam, n);
                    v_scale(S->cam_local_y, ay * S->fs->cam_aperture_radius, temp, n);
                    v_add(virtCam, temp, virtCam, n);
                }
            }
            v_sub(pixel, virtCam, look, n);
            l_clr.r = l_clr.g = l_clr.b = 0.0;
            l_clr.a = 1.0;
            v_unitize(look, n);
            get_ray_color_d(T, virtCam, look, &l_clr, 1.0, max_optic_depth, 1, depth);
            if (i == 0) {
                one.primary = T->cnt.primary - before.primary;
                one.secondary = T->cnt.secondary - before.secondary;
                one.shadow = T->cnt.shadow - before.shadow;
            }
        }
        if (i > 1) {
            clr_diff = MAXV(fabs(t_clr.r / (i - 1) - (t_clr.r + l_clr.r) / i),
                            MAXV(fabs(t_clr.g / (i - 1) - (t_clr.g + l_clr.g) / i),
                                 fabs(t_clr.b / (i - 1) - (t_clr.b + l_clr.b) / i)));
        }
        t_clr.r += l_clr.r;
        t_clr.g += l_clr.g;
        t_clr.b += l_clr.b;
        t_clr.a += l_clr.a;
        t_samples += 1;
    }
    clr->r = t_clr.r / t_samples;
    clr->g = t_clr.g / t_samples;
    clr->b = t_clr.b / t_samples;
    clr->a = t_clr.a / t_samples;
    if (!literal) {
        T->cnt.primary = before.primary + one.primary * t_samples;
        T->cnt.secondary = before.secondary + one.secondary * t_samples;
        T->cnt.shadow = before.shadow + one.shadow * t_samples;
    }
    if (k_out) *k_out = t_samples;
}

/* ---------------------------------------------------------------- render driver */

typedef struct {
    pscene *S;
    const ndt_render_params *p;
    double *rgba;
    int thr, threads, literal;
    ray_counts cnt;         /* reference-equivalent counts (times k) */
    ray_counts unique;      /* one sample per pixel */
    /* recursive anti-aliasing */
    const double *pass1;    /* (width+1) x n1 rows of corner samples */
    const int *row_of;      /* image row of every pass-1 row */
    int n1;
    long long resampled, aa_samples;
    double *depth;          /* depth map (1/distance of the primary hit), or NULL */
    int stochastic;
} job;

/* render_pixel (ndt.c:578-653, MONO) + get_pixel_color: the sample at image position (i, j), both in
 * pixels of a width x height image (fractional for the anti-aliasing samples) */
static void one_eye(job *J, tctx *T, double x, double y, pix *clr, int mode, double *depth)
{
    int k = 0;
    ray_counts b = T->cnt;
    get_pixel_color_m(T, x, y, clr, J->p->max_optic_depth, J->literal, &k, mode, depth);
    if (T->stochastic) k = 1;       /* every sample was a ray tree of its own */
    J->unique.primary += (T->cnt.primary - b.primary) / k;
    J->unique.secondary += (T->cnt.secondary - b.secondary) / k;
    J->unique.shadow += (T->cnt.shadow - b.shadow) / k;
}

static void render_pixel_d(job *J, tctx *T, int width, int height, double i, double j, pix *clr, double *depth)
{
    double ip = i, jp = j;
    int mode = CAM_CENTER;
    const int stereo = J->p->stereo;
    if (stereo == NDT_STEREO_SIDE_SIDE) {           /* x_scale = 0.5, ndt.c:591-601, 913-914 */
        if (i < width / 2) { ip = ip / 0.5; mode = CAM_LEFT; }
        else { ip = (ip - width / 2) / 0.5; mode = CAM_RIGHT; }
    }
    if (stereo == NDT_STEREO_OVER_UNDER) {          /* y_scale = 0.5, ndt.c:602-612 */
        if (j < height / 2) { jp = jp / 0.5; mode = CAM_LEFT; }
        else { jp = (jp - height / 2) / 0.5; mode = CAM_RIGHT; }
    }
    double x = ip / (double)width - 0.5;
    double y = -(jp / (double)height - 0.5);
    if (stereo == NDT_STEREO_HIDEF) {               /* frame packing, ndt.c:614-631: left eye, 45 blank lines, right eye */
        if (j < 1080) {
            mode = CAM_LEFT;
        } else if (j > (1080 + 45)) {
            jp = j - (1080 + 45);
            mode = CAM_RIGHT;
        } else {
            clr->r = clr->g = clr->b = 0;
            clr->a = 1.0;           /* the reference leaves alpha unset here */
            return;
        }
        y = -(jp / 1080.0 - 0.5);
    }
    if (stereo == NDT_STEREO_ANAGLYPH) {            /* ndt.c:636-647 */
        pix left, right;
        one_eye(J, T, x, y, &left, CAM_LEFT, depth);
        one_eye(J, T, x, y, &right, CAM_RIGHT, NULL);
        clr->r = 0.299 * left.r + 0.587 * left.g + 0.114 * left.b;
        clr->g = 0;
        clr->b = 0.299 * right.r + 0.587 * right.g + 0.114 * right.b;
        clr->a = 1.0;
    } else {
        one_eye(J, T, x, y, clr, mode, depth);
    }
}

static void render_pixel(job *J, tctx *T, int width, int height, double i, double j, pix *clr)
{
    render_pixel_d(J, T, width, height, i, j, clr, NULL);
}

static void tctx_open(tctx *T, job *J)
{
    T->S = J->S;
    T->mask = (unsigned char *)malloc((size_t)(J->S->n_items > 0 ? J->S->n_items : 1));
    memset(&T->cnt, 0, sizeof(T->cnt));
    T->specular = J->p->specular;
    T->samples = J->p->samples;
    T->stochastic = J->stochastic;
    T->aa_lens = J->p->recursive_aa && J->S->fs->cam_aperture_radius != 0.0;
    T->pix_w = 1.0 / (J->p->width + (J->p->recursive_aa ? 1 : 0));
    T->pix_h = 1.0 / (J->p->height + (J->p->recursive_aa ? 1 : 0));
}

/* render_lines_thread / render_line, ndt.c:803 / 735 */
static void *render_rows(void *arg)
{
    job *J = (job *)arg;
    const ndt_render_params *p = J->p;
    tctx T;
    tctx_open(&T, J);
    int width = p->width, height = p->height;
    int local = 0;
    for (int j = p->row_begin; j < height; j += p->row_step, ++local) {
        if (local % J->threads != J->thr)
            continue;
        double depth = 0.0;     /* render_line's variable (ndt.c:739) is not initialised; 0 here */
        for (int i = 0; i < width; ++i) {
            pix clr;
            render_pixel_d(J, &T, width, height, i, j, &clr, J->depth ? &depth : NULL);
            double *out = J->rgba + ((size_t)local * width + i) * 4;    /* dbl_image_set_pixel, image.c:126 */
            out[0] = clr.r; out[1] = clr.g; out[2] = clr.b; out[3] = clr.a;
            if (J->depth) J->depth[(size_t)local * width + i] = depth;   /* ndt.c:753-756 */
        }
    }
    J->cnt = T.cnt;
    free(T.mask);
    return NULL;
}

/* ---- Whitted's recursive anti-aliasing, ndt.c:655-733 */

/* image_avg_dbl_pixels4, image.c:1175-1197 */
static void avg4(const pix *p1, const pix *p2, const pix *p3, const pix *p4, pix *avg, double *var)
{
    avg->r = (p1->r + p2->r + p3->r + p4->r) / 4;
    avg->g = (p1->g + p2->g + p3->g + p4->g) / 4;
    avg->b = (p1->b + p2->b + p3->b + p4->b) / 4;
    avg->a = (p1->a + p2->a + p3->a + p4->a) / 4;
    if (var) {
        double v = 0;
        v += fabs(avg->r - p1->r) + fabs(avg->r - p2->r) + fabs(avg->r - p3->r) + fabs(avg->r - p4->r);
        v += fabs(avg->g - p1->g) + fabs(avg->g - p2->g) + fabs(avg->g - p3->g) + fabs(avg->g - p4->g);
        v += fabs(avg->b - p1->b) + fabs(avg->b - p2->b) + fabs(avg->b - p3->b) + fabs(avg->b - p4->b);
        v += fabs(avg->a - p1->a) + fabs(avg->a - p2->a) + fabs(avg->a - p3->a) + fabs(avg->a - p4->a);
        *var = v;
    }
}

/* recursive_resample, ndt.c:655-707.  width/height are the first pass's (image + 1). */
static void recursive_resample(job *J, tctx *T, int width, int height, double x, double y, double step,
                               const pix *p1, const pix *p2, const pix *p3, const pix *p4, pix *res)
{
    const int aa_depth = J->p->aa_depth;
    pix p5, p6, p7, p8, p9;
    if (aa_depth <= 0 || step < 1.0 / (2 << (aa_depth - 1))) {
        avg4(p1, p2, p3, p4, res, NULL);
        return;
    }
    double hs = step / 2;
    render_pixel(J, T, width, height, x + hs, y + hs, &p5);       /* center */
    render_pixel(J, T, width, height, x + hs, y, &p6);            /* top middle */
    render_pixel(J, T, width, height, x, y + hs, &p7);            /* left edge */
    render_pixel(J, T, width, height, x + step, y + hs, &p8);     /* right edge */
    render_pixel(J, T, width, height, x + hs, y + step, &p9);     /* bottom middle */
    J->aa_samples += 5;
    pix sp1, sp2, sp3, sp4;
    double var1 = 0, var2 = 0, var3 = 0, var4 = 0;
    double threshold = J->p->aa_diff / 255.0;
    avg4(p1, &p6, &p7, &p5, &sp1, &var1);
    if (var1 > threshold) recursive_resample(J, T, width, height, x, y, hs, p1, &p6, &p7, &p5, &sp1);
    avg4(p2, &p6, &p8, &p5, &sp2, &var2);
    if (var2 > threshold) recursive_resample(J, T, width, height, x + hs, y, hs, &p6, p2, &p5, &p8, &sp2);
    avg4(p3, &p9, &p7, &p5, &sp3, &var3);
    if (var3 > threshold) recursive_resample(J, T, width, height, x, y + hs, hs, &p7, &p5, p3, &p9, &sp3);
    avg4(p4, &p9, &p8, &p5, &sp4, &var4);
    if (var4 > threshold) recursive_resample(J, T, width, height, x + hs, y + hs, hs, &p5, &p8, &p9, p4, &sp4);
    avg4(&sp1, &sp2, &sp3, &sp4, res, NULL);
}

/* first pass of the anti-aliased render: rows row_of[0..n1) of the (width+1) x (height+1) corner image */
static void *render_corner_rows(void *arg)
{
    job *J = (job *)arg;
    const ndt_render_params *p = J->p;
    tctx T;
    tctx_open(&T, J);
    const int w1 = p->width + 1, h1 = p->height + 1;
    double *img = (double *)J->pass1;
    for (int r = 0; r < J->n1; ++r) {
        if (r % J->threads != J->thr)
            continue;
        /* the depth map of an anti-aliased render (ndt.c:930-935, 753-756): the first pass's depths; the map is width x height,
         * the corner samples of column `width` and row `height` fall outside and are dropped (image.c:126: bounds check) */
        const int j = J->row_of[r];
        const int local = (j >= p->row_begin && (j - p->row_begin) % p->row_step == 0 && j < p->height) ? (j - p->row_begin) / p->row_step : -1;
        double depth = 0.0;
        for (int i = 0; i < w1; ++i) {
            pix clr;
            render_pixel_d(J, &T, w1, h1, i, j, &clr, J->depth ? &depth : NULL);
            double *out = img + ((size_t)r * w1 + i) * 4;
            out[0] = clr.r; out[1] = clr.g; out[2] = clr.b; out[3] = clr.a;
            if (J->depth && local >= 0 && i < p->width) J->depth[(size_t)local * p->width + i] = depth;
        }
    }
    J->cnt = T.cnt;
    free(T.mask);
    return NULL;
}

/* resample_lines_thread / resample_line / resample_pixel, ndt.c:851 / 762 / 709 */
static void *resample_rows(void *arg)
{
    job *J = (job *)arg;
    const ndt_render_params *p = J->p;
    tctx T;
    tctx_open(&T, J);
    const int width = p->width, height = p->height, w1 = width + 1;
    int local = 0;
    for (int j = p->row_begin; j < height; j += p->row_step, ++local) {
        if (local % J->threads != J->thr)
            continue;
        /* pass-1 rows of image rows j and j+1 */
        int r0 = -1, r1 = -1;
        for (int r = 0; r < J->n1; ++r) {
            if (J->row_of[r] == j) r0 = r;
            if (J->row_of[r] == j + 1) r1 = r;
        }
        for (int i = 0; i < width; ++i) {
            pix c[4], clr;
            const double *q[4] = { J->pass1 + ((size_t)r0 * w1 + i) * 4, J->pass1 + ((size_t)r0 * w1 + i + 1) * 4,
                                   J->pass1 + ((size_t)r1 * w1 + i) * 4, J->pass1 + ((size_t)r1 * w1 + i + 1) * 4 };
            for (int k = 0; k < 4; ++k) { c[k].r = q[k][0]; c[k].g = q[k][1]; c[k].b = q[k][2]; c[k].a = q[k][3]; }
            double var = 0.0;
            avg4(&c[0], &c[1], &c[2], &c[3], &clr, &var);
            if (var > p->aa_diff / 255.0) {
                J->resampled += 1;
                recursive_resample(J, &T, width + 1, height + 1, i, j, 1.0, &c[0], &c[1], &c[2], &c[3], &clr);
            }
            double *out = J->rgba + ((size_t)local * width + i) * 4;
            out[0] = clr.r; out[1] = clr.g; out[2] = clr.b; out[3] = clr.a;
        }
    }
    J->cnt = T.cnt;
    free(T.mask);
    return NULL;
}

static int check_supported(const ndt_flat_scene *fs, const ndt_render_params *p)
{
    if (fs->cam_type < 0 || fs->cam_type > 2) return NDT_E_UNSUPPORTED;
    if (fs->cam_type != 0 && (fs->cam_local_x_off < 0 || fs->cam_local_y_off < 0 || fs->cam_local_z_off < 0)) return NDT_E_INVALID;
    if (p && p->stereo != NDT_STEREO_MONO && (fs->cam_left_eye_off < 0 || fs->cam_right_eye_off < 0)) return NDT_E_INVALID;
    for (int i = 0; i < fs->n_lights; ++i) {
        const ndt_flat_light *L = &fs->lights[i];
        if ((L->type == NDT_LIGHT_DISK || L->type == NDT_LIGHT_RECT) && (L->area_off < 0 || L->pos_off < 0))
            return NDT_E_INVALID;
        if ((L->type == NDT_LIGHT_POINT || L->type == NDT_LIGHT_SPOT) && L->pos_off < 0) return NDT_E_INVALID;
        if ((L->type == NDT_LIGHT_DIRECTIONAL || L->type == NDT_LIGHT_SPOT) && L->dir_off < 0) return NDT_E_INVALID;
    }
    if (p && (p->samples < 1 || p->width < 1 || p->height < 1 || p->row_step < 1 || p->row_begin < 0))
        return NDT_E_INVALID;
    /* samples > 1: jitter + lens sampling from drand48 (ndt.c:505-542), in every stereo mode and with a depth map; with
     * recursive anti-aliasing on top (no jitter, `samples` lens samples per anti-aliasing sample) it is not pinned by fixtures */
    if (p && p->samples > 1 && p->recursive_aa) return NDT_E_UNSUPPORTED;
    if (p && p->samples > 1 && fs->cam_aperture_radius != 0.0 && (fs->cam_local_x_off < 0 || fs->cam_local_y_off < 0)) return NDT_E_INVALID;
    if (p && (p->stereo < NDT_STEREO_MONO || p->stereo > NDT_STEREO_HIDEF)) return NDT_E_UNSUPPORTED;
    /* recursive AA: render_pixel does the image split / the two eyes of an anaglyph for every sample (ndt.c:590-650).  Not
     * frame packing: the samples that fall on the 45 blank lines come back with an alpha recursive_resample never set
     * (ndt.c:662: p5 .. p9 are uninitialised) and it averages them in */
    if (p && p->recursive_aa && p->stereo == NDT_STEREO_HIDEF) return NDT_E_UNSUPPORTED;
    /* (recursive AA with a lens: every sample draws its lens point from drand48, ndt.c:528-542 -- a stochastic render) */
    if (p && p->recursive_aa && fs->cam_aperture_radius != 0.0 && (fs->cam_local_x_off < 0 || fs->cam_local_y_off < 0)) return NDT_E_INVALID;
    return NDT_OK;
}

static void run_jobs(job *jobs, int threads, void *(*fn)(void *))
{
    pthread_t *thr = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int i = 0; i < threads; ++i) {
        if (threads > 1) pthread_create(&thr[i], NULL, fn, &jobs[i]);
        else fn(&jobs[i]);
    }
    if (threads > 1)
        for (int i = 0; i < threads; ++i) pthread_join(thr[i], NULL);
    free(thr);
}

/* render_image, ndt.c:900.  flags bit0: literal re-sampling (trace all k samples like the
 * reference; only changes run time and nothing else).  stats->rays_* = unique rays,
 * stats->rays_ref_equiv = what the reference's trace_kd counter reads. */
/* samples > 1: the *rand48 state the next render starts from (the reference's scene programs draw from the
 * same stream before the render does; the fixtures record where it stood) */
static unsigned short g_seed48[3] = { 0x330E, 0xABCD, 0x1234 };     /* glibc's state in a fresh process */
void ndt_oracle_set_seed48(unsigned short s0, unsigned short s1, unsigned short s2)
{
    g_seed48[0] = s0; g_seed48[1] = s1; g_seed48[2] = s2;
}

int ndt_oracle_render_depth(const ndt_flat_scene *fs, const ndt_render_params *p, double *rgba, double *depth,
                            ndt_render_stats *stats, int threads, int flags);
int ndt_oracle_render(const ndt_flat_scene *fs, const ndt_render_params *p, double *rgba, ndt_render_stats *stats,
                      int threads, int flags)
{
    return ndt_oracle_render_depth(fs, p, rgba, NULL, stats, threads, flags);
}

/* the same with the depth map render_image fills when it is given a depth file name (ndt.c:930-935, 753-756):
 * rows x width doubles, 1/distance of the primary hit, 0 where the primary ray misses */
int ndt_oracle_render_depth(const ndt_flat_scene *fs, const ndt_render_params *p, double *rgba, double *depth,
                            ndt_render_stats *stats, int threads, int flags)
{
    pscene S;
    int rc = check_supported(fs, p);
    if (rc != NDT_OK) return rc;
    rc = prepare_scene(fs, &S);
    if (rc != NDT_OK) return rc;
    if (threads < 1) threads = 1;
    int stochastic = p->samples > 1;
    for (int i = 0; i < fs->n_lights; ++i)
        if (fs->lights[i].type == NDT_LIGHT_DISK || fs->lights[i].type == NDT_LIGHT_RECT) stochastic = 1;
    if (p->recursive_aa && fs->cam_aperture_radius != 0.0) stochastic = 1;     /* the lens is sampled in this mode too (ndt.c:528) */
    if (stochastic) {
        /* the random numbers come from one global stream in pixel order: one thread, starting where the
         * reference run that made the fixture stood (ndt_oracle_set_seed48) */
        unsigned short x0[3] = { g_seed48[0], g_seed48[1], g_seed48[2] };
        seed48(x0);
        threads = 1;
    }
    if (p->stereo == NDT_STEREO_HIDEF) v_scale(S.cam_dir_x, p->width / (double)1080, S.cam_dir_x, S.n);      /* ndt.c:928 */
    else v_scale(S.cam_dir_x, p->width / (double)p->height, S.cam_dir_x, S.n);                                /* ndt.c:926 */
    job *jobs = (job *)calloc((size_t)threads, sizeof(job));
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    for (int i = 0; i < threads; ++i) {
        jobs[i].S = &S; jobs[i].p = p; jobs[i].rgba = rgba; jobs[i].thr = i; jobs[i].threads = threads;
        jobs[i].literal = flags & 1;
        jobs[i].depth = depth;
        jobs[i].stochastic = stochastic;
    }
    ray_counts cnt1 = { 0, 0, 0 };
    double *pass1 = NULL;
    int *row_of = NULL;
    if (!p->recursive_aa) {
        run_jobs(jobs, threads, render_rows);
    } else {
        /* first pass (ndt.c:919-976): the corner rows this shard's pixels touch */
        int n1 = 0;
        row_of = (int *)malloc(sizeof(int) * (size_t)(2 * p->height + 2));
        for (int j = p->row_begin; j < p->height; j += p->row_step) {
            if (n1 == 0 || row_of[n1 - 1] != j) row_of[n1++] = j;
            row_of[n1++] = j + 1;
        }
        pass1 = (double *)calloc((size_t)(n1 > 0 ? n1 : 1) * (size_t)(p->width + 1) * 4, sizeof(double));
        for (int i = 0; i < threads; ++i) { jobs[i].pass1 = pass1; jobs[i].row_of = row_of; jobs[i].n1 = n1; }
        run_jobs(jobs, threads, render_corner_rows);
        for (int i = 0; i < threads; ++i) {
            cnt1.primary += jobs[i].cnt.primary; cnt1.secondary += jobs[i].cnt.secondary; cnt1.shadow += jobs[i].cnt.shadow;
        }
        /* second pass (ndt.c:1039-1087) */
        run_jobs(jobs, threads, resample_rows);
    }
    gettimeofday(&t1, NULL);
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->rays_ref_equiv = cnt1.primary + cnt1.secondary + cnt1.shadow;
        for (int i = 0; i < threads; ++i) {
            stats->rays_primary += jobs[i].unique.primary;
            stats->rays_secondary += jobs[i].unique.secondary;
            stats->rays_shadow += jobs[i].unique.shadow;
            stats->rays_ref_equiv += jobs[i].cnt.primary + jobs[i].cnt.secondary + jobs[i].cnt.shadow;
            stats->pixels_resampled += jobs[i].resampled;
            stats->aa_samples += jobs[i].aa_samples;
        }
        stats->frame_ms = 1e3 * ((t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec));
    }
    free(jobs);
    free(pass1);
    free(row_of);
    free_scene(&S);
    return NDT_OK;
}

/* Batch of trace_kd queries; same contract as ndt_hip_trace_rays. */
int ndt_oracle_trace_rays(const ndt_flat_scene *fs, int64_t nrays, const double *o, const double *v,
                          const double *dist_limit, int32_t *obj, double *hit, double *normal)
{
    pscene S;
    int rc = prepare_scene(fs, &S);
    if (rc != NDT_OK) return rc;
    int n = S.n;
    unsigned char *mask = (unsigned char *)malloc((size_t)(S.n_items > 0 ? S.n_items : 1));
    for (int64_t r = 0; r < nrays; ++r) {
        double *h = hit + r * n, *nm = normal + r * n;
        int ptr = -1;
        v_zero(h, n);
        v_zero(nm, n);
        trace_kd(&S, o + r * n, v + r * n, h, nm, &ptr, dist_limit[r], mask);
        obj[r] = ptr;
    }
    free(mask);
    free_scene(&S);
    return NDT_OK;
}

/* pixel_d2c, image.h:36-39: (unsigned char)(sqrt(clamp01(x))*255) */
void ndt_oracle_quantize(const double *rgba, unsigned char *out, int64_t n_values)
{
    for (int64_t i = 0; i < n_values; ++i) {
        double d = rgba[i];
        double m = (1.0 < d) ? 1.0 : d;         /* MIN(1.0,d), image.h:28 */
        m = (0.0 > m) ? 0.0 : m;                /* MAX(0.0,.), image.h:31 */
        out[i] = (unsigned char)(sqrt(m) * 255);
    }
}
