/* ndt_nelder_mead.c -- the Nelder-Mead state machine ndt uses to fit bounding spheres
 * (reference nelder-mead.c), restated.  The caller drives it:
 *     while (!nm_done(..)) { nm_add_result(point, f(point)); nm_next_point(point); }
 * Bounding spheres gate every intersection test and are not perfectly conservative
 * (SURVEY.md 8a row O-bs), so the iteration must reproduce the reference's points exactly:
 * same coefficients (alpha 1, beta .5, gamma 2, delta .5), same stable ascending sort, same
 * transition quirks (e.g. what a rejected contraction falls through to). */
#include "ndt_host_api.h"

enum nm_state { NM_INITIAL, NM_REFLECT, NM_EXPAND, NM_CONTRACT_OUT, NM_CONTRACT_IN, NM_SHRINK, NM_SHRINK2 };

typedef struct {
    double *x;          /* dims parameters */
    double f;
} nm_point;

typedef struct {
    int dims, iterations, count;        /* count = simplex points so far (<= dims+1) */
    enum nm_state state;
    nm_point *simplex;                  /* dims+1 */
    double *seed;
    nm_point xr, xe, xc;
    double *s_shrink;
    double alpha, beta, gamma, delta;
} nm_t;

static double *vec_new(int n) { return (double *)calloc((size_t)n, sizeof(double)); }
static void pt_set(nm_t *nm, nm_point *dst, const double *x, double f)
{
    memcpy(dst->x, x, (size_t)nm->dims * sizeof(double));
    dst->f = f;
}

/* ascending by value, stable: the reference bubbles adjacent out-of-order pairs (strict >)
 * until none is left (nelder-mead.c:56-81), which yields the stable order */
static void nm_sort(nm_t *nm)
{
    for (int i = 1; i < nm->count; ++i) {
        nm_point key = nm->simplex[i];
        int j = i - 1;
        while (j >= 0 && nm->simplex[j].f > key.f) {
            nm->simplex[j + 1] = nm->simplex[j];
            --j;
        }
        nm->simplex[j + 1] = key;
    }
}

void nm_init(void **out, int dimensions)
{
    nm_t *nm = (nm_t *)calloc(1, sizeof(nm_t));
    nm->dims = dimensions;
    nm->state = NM_INITIAL;
    nm->alpha = 1; nm->beta = 0.5; nm->gamma = 2; nm->delta = 0.5;
    nm->simplex = (nm_point *)calloc((size_t)dimensions + 1, sizeof(nm_point));
    for (int i = 0; i <= dimensions; ++i) nm->simplex[i].x = vec_new(dimensions);
    nm->seed = vec_new(dimensions);
    nm->xr.x = vec_new(dimensions);
    nm->xe.x = vec_new(dimensions);
    nm->xc.x = vec_new(dimensions);
    nm->s_shrink = vec_new(dimensions);
    *out = nm;
}

void nm_free(void *p)
{
    nm_t *nm = (nm_t *)p;
    for (int i = 0; i <= nm->dims; ++i) free(nm->simplex[i].x);
    free(nm->simplex); free(nm->seed); free(nm->xr.x); free(nm->xe.x); free(nm->xc.x); free(nm->s_shrink);
    free(nm);
}

void nm_set_seed(void *p, vectNd *seed)
{
    nm_t *nm = (nm_t *)p;
    if (nm->state != NM_INITIAL) return;
    memcpy(nm->seed, seed->v, (size_t)nm->dims * sizeof(double));
}

void nm_best_point(void *p, vectNd *result)
{
    nm_t *nm = (nm_t *)p;
    int best = 0;
    for (int i = 0; i < nm->count; ++i)
        if (nm->simplex[i].f < nm->simplex[best].f) best = i;
    if (best < nm->count) memcpy(result->v, nm->simplex[best].x, (size_t)nm->dims * sizeof(double));
}

void nm_add_result(void *p, vectNd *parameters, double value)
{
    nm_t *nm = (nm_t *)p;
    const int last = nm->dims;          /* index of the worst point once the simplex is full */
    nm->iterations += 1;

    /* the two shrink replacements just take the value (nelder-mead.c:177-187) */
    if (nm->state == NM_SHRINK2) {
        pt_set(nm, &nm->simplex[nm->count - 2], parameters->v, value);
        nm->state = NM_REFLECT;
        return;
    }
    if (nm->state == NM_SHRINK) {
        pt_set(nm, &nm->simplex[nm->count - 1], parameters->v, value);
        nm->state = NM_SHRINK2;
        return;
    }
    /* still collecting the initial simplex */
    if (nm->count <= nm->dims) {
        pt_set(nm, &nm->simplex[nm->count], parameters->v, value);
        nm->count += 1;
        if (nm->count >= nm->dims + 1) nm->state = NM_REFLECT;
        return;
    }
    nm_sort(nm);
    const double fh = nm->simplex[last].f, fs = nm->simplex[last - 1].f, fl = nm->simplex[0].f;
    const double fr = value;

    if (nm->state == NM_REFLECT) {
        pt_set(nm, &nm->xr, parameters->v, value);
        if (fl <= nm->xr.f && nm->xr.f < fs) {
            pt_set(nm, &nm->simplex[last], parameters->v, value);
            return;
        }
    }
    if (nm->state == NM_EXPAND) {
        pt_set(nm, &nm->xe, parameters->v, value);
        if (nm->xe.f < nm->xr.f) pt_set(nm, &nm->simplex[last], nm->xe.x, nm->xe.f);
        else pt_set(nm, &nm->simplex[last], nm->xr.x, nm->xr.f);
        nm->state = NM_REFLECT;
        return;
    }
    if (nm->state == NM_CONTRACT_OUT) {
        pt_set(nm, &nm->xc, parameters->v, value);
        if (nm->xc.f < nm->xr.f) {
            pt_set(nm, &nm->simplex[last], nm->xc.x, nm->xc.f);
            nm->state = NM_REFLECT;
            return;
        }
    }
    if (nm->state == NM_CONTRACT_IN) {
        pt_set(nm, &nm->xc, parameters->v, value);
        if (nm->xc.f < fh) {
            pt_set(nm, &nm->simplex[last], nm->xc.x, nm->xc.f);
            nm->state = NM_REFLECT;
            return;
        }
    }
    /* not accepted: what to evaluate next (nelder-mead.c:279-294) */
    if (fr < fl) {
        nm->state = NM_EXPAND;
    } else if (fr >= fs) {
        nm->state = (fs <= fr && fr < fh) ? NM_CONTRACT_OUT : NM_CONTRACT_IN;
    } else {
        nm->state = NM_SHRINK;
    }
}

void nm_next_point(void *p, vectNd *vector)
{
    nm_t *nm = (nm_t *)p;
    const int d = nm->dims;
    if (nm->state == NM_INITIAL && nm->count < d + 1) {
        /* initial simplex: the seed, then seed + k * e_(k-1) (nelder-mead.c:304-319) */
        if (nm->count > 0) {
            memcpy(vector->v, nm->seed, (size_t)d * sizeof(double));
            vector->v[nm->count - 1] += nm->count;
        } else {
            memcpy(vector->v, nm->seed, (size_t)d * sizeof(double));   /* seed.n == vector.n always here */
        }
        return;
    }
    if (nm->count != d + 1) {
        memcpy(vector->v, nm->seed, (size_t)d * sizeof(double));
        return;
    }
    if (nm->state != NM_SHRINK && nm->state != NM_SHRINK2) nm_sort(nm);
    const double *h = nm->simplex[d].x, *s = nm->simplex[d - 1].x;

    /* centroid of all points but the worst: running sum from zero, then * 1/d */
    double *c = vec_new(d);
    for (int i = 0; i < nm->count - 1; ++i)
        for (int k = 0; k < d; ++k) c[k] = c[k] + nm->simplex[i].x[k];
    const double inv = 1.0 / (nm->count - 1);
    for (int k = 0; k < d; ++k) c[k] = c[k] * inv;

    switch (nm->state) {
    case NM_INITIAL:
        break;
    case NM_REFLECT:
        for (int k = 0; k < d; ++k) vector->v[k] = c[k] + (c[k] - h[k]) * nm->alpha;
        break;
    case NM_EXPAND:
        for (int k = 0; k < d; ++k) vector->v[k] = c[k] + (nm->xr.x[k] - c[k]) * nm->gamma;
        break;
    case NM_CONTRACT_OUT:
        for (int k = 0; k < d; ++k) vector->v[k] = c[k] + (nm->xr.x[k] - c[k]) * nm->beta;
        break;
    case NM_CONTRACT_IN:
        for (int k = 0; k < d; ++k) vector->v[k] = c[k] + (h[k] - c[k]) * nm->beta;
        break;
    case NM_SHRINK:
        /* upstream shrinks towards x_r, not towards the best point (nelder-mead.c:389-396) */
        for (int k = 0; k < d; ++k) nm->s_shrink[k] = (nm->xr.x[k] + s[k]) * 0.5;
        for (int k = 0; k < d; ++k) vector->v[k] = (nm->xr.x[k] + h[k]) * 0.5;
        break;
    case NM_SHRINK2:
        memcpy(vector->v, nm->s_shrink, (size_t)d * sizeof(double));
        memset(nm->s_shrink, 0, (size_t)d * sizeof(double));
        break;
    }
    free(c);
}

int nm_simplex_point(void *p, int which, vectNd *point, double *value)
{
    nm_t *nm = (nm_t *)p;
    if (which >= nm->count) return 0;
    if (point) memcpy(point->v, nm->simplex[which].x, (size_t)nm->dims * sizeof(double));
    if (value) *value = nm->simplex[which].f;
    return 1;
}

int nm_done(void *p, double threshold, int iterations)
{
    nm_t *nm = (nm_t *)p;
    if (nm->state == NM_INITIAL) return 0;
    if (nm->iterations > iterations) return 1;
    if (nm->state != NM_SHRINK && nm->state != NM_SHRINK2) nm_sort(nm);
    /* distance between best and worst parameters, in the lane-pair dot order */
    vectNd a, b;
    double dist;
    vectNd_alloc(&a, nm->dims);
    vectNd_alloc(&b, nm->dims);
    memcpy(a.v, nm->simplex[0].x, (size_t)nm->dims * sizeof(double));
    memcpy(b.v, nm->simplex[nm->count - 1].x, (size_t)nm->dims * sizeof(double));
    vectNd_dist(&a, &b, &dist);
    vectNd_free(&a);
    vectNd_free(&b);
    return dist < threshold;
}
