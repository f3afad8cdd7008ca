/* ndt_vect.c -- out-of-line vectNd operations (reference vectNd.c), restated.
 * Operation order is kept: the results feed bounding-sphere fits, kd builds and camera aiming
 * whose outputs decide pixels. */
#include "ndt_host_api.h"

int vectNd_cross(vectNd *vects, vectNd *res)
{
    /* never implemented upstream either (vectNd.c:16-33): validates sizes only */
    if (!res) return VECTND_FAIL;
    int dim = vects[0].n;
    for (int i = 1; i < dim - 1; ++i)
        if (vects[i].n != dim) return VECTND_FAIL;
    return VECTND_SUCCESS;
}

/* vectNd.c:35-58: out1 = unit(in1 - proj_in2(in1)), out2 = unit(in2) */
int vectNd_orthogonalize(vectNd *in1, vectNd *in2, vectNd *out1, vectNd *out2)
{
    vectNd along;
    vectNd_calloc(&along, in1->n);
    vectNd_proj(in1, in2, &along);
    if (out1) vectNd_sub(in1, &along, out1);
    if (out2) vectNd_copy(out2, in2);
    vectNd_free(&along);
    if (out1) vectNd_unitize(out1);
    if (out2) vectNd_unitize(out2);
    return VECTND_SUCCESS;
}

/* vectNd.c:64-81 */
int vectNd_angle(vectNd *v1, vectNd *v2, double *angle)
{
    double d, l1, l2;
    vectNd_dot(v1, v2, &d);
    vectNd_l2norm(v1, &l1);
    vectNd_l2norm(v2, &l2);
    double div = l1 * l2;
    *angle = (fabs(div) > EPSILON) ? acos(d / div) : -1;
    return VECTND_SUCCESS;
}

/* vectNd.c:83-99: angle at p2 between p1 and p3 */
int vectNd_angle3(vectNd *p1, vectNd *p2, vectNd *p3, double *angle)
{
    vectNd a, b;
    vectNd_alloc(&a, p1->n);
    vectNd_alloc(&b, p1->n);
    vectNd_sub(p1, p2, &a);
    vectNd_sub(p3, p2, &b);
    vectNd_angle(&a, &b, angle);
    vectNd_free(&a);
    vectNd_free(&b);
    return VECTND_SUCCESS;
}

/* vectNd.c:101-117 */
int vectNd_reflect(vectNd *u, vectNd *n, vectNd *res, double mag)
{
    double nu, nn;
    vectNd step;
    vectNd_dot(n, u, &nu);
    vectNd_dot(n, n, &nn);
    vectNd_alloc(&step, u->n);
    vectNd_scale(n, (1 + mag) * nu / nn, &step);
    vectNd_sub(u, &step, res);
    vectNd_free(&step);
    return VECTND_SUCCESS;
}

/* vectNd.c:119-188 (unitizes the caller's normal, as upstream does) */
int vectNd_refract(vectNd *u, vectNd *n, vectNd *res, double index)
{
    int dim = u->n;
    vectNd rev_u, rev_n, un, np, ref_n, ref_p;
    vectNd_alloc(&rev_u, dim);
    vectNd_alloc(&rev_n, dim);
    vectNd_scale(u, -1, &rev_u);
    vectNd_scale(n, -1, &rev_n);
    double un_dot, theta_in, theta_out;
    vectNd_dot(&rev_u, n, &un_dot);
    if (un_dot < 0) {
        index = 1 / index;
        vectNd_angle(&rev_u, &rev_n, &theta_in);
    } else {
        vectNd_angle(&rev_u, n, &theta_in);
    }
    double sin_out = sin(theta_in) / index;
    theta_out = (sin_out <= 1.0) ? asin(sin_out) : M_PI - theta_in;
    vectNd_unitize(&rev_n);
    vectNd_unitize(n);
    vectNd_alloc(&un, dim);
    vectNd_alloc(&np, dim);
    vectNd_proj_unit(u, &rev_n, &un);
    vectNd_sub(u, &un, &np);
    vectNd_unitize(&np);
    double rn = cos(theta_out), rp = sin(theta_out);
    vectNd_alloc(&ref_n, dim);
    vectNd_alloc(&ref_p, dim);
    if (un_dot < 0) vectNd_scale(n, rn, &ref_n);
    else vectNd_scale(&rev_n, rn, &ref_n);
    vectNd_scale(&np, rp, &ref_p);
    vectNd_add(&ref_n, &ref_p, res);
    vectNd_free(&un); vectNd_free(&np); vectNd_free(&ref_n); vectNd_free(&ref_p);
    vectNd_free(&rev_n); vectNd_free(&rev_u);
    return VECTND_SUCCESS;
}

/* vectNd.c:190-200 */
int vectNd_interpolate(vectNd *s, vectNd *e, double t, vectNd *r)
{
    vectNd off;
    vectNd_alloc(&off, s->n);
    vectNd_sub(e, s, &off);
    vectNd_scale(&off, t, &off);
    vectNd_add(s, &off, r);
    vectNd_free(&off);
    return VECTND_SUCCESS;
}

/* vectNd.c:202-269.  Upstream multiplies by an identity matrix with four entries replaced
 * (matrix.c:98-118: each output component is a left-to-right sum over all columns, starting
 * from 0.0); the sums below add the same terms in the same order, zeros included. */
int vectNd_rotate(vectNd *v, vectNd *center, int i, int j, double angle, vectNd *res)
{
    if (i == j) return VECTND_FAIL;
    if (angle == 0.0) return VECTND_SUCCESS;
    int dim = v->n;
    if (i >= dim || j >= dim) {
        fprintf(stderr, "%s: attempt to rotate %i dimensional vector in %i,%i plane.\n", __FUNCTION__, dim, i, j);
        return VECTND_FAIL;
    }
    if (!res) res = v;
    vectNd tmp, out;
    vectNd_alloc(&tmp, dim);
    vectNd_alloc(&out, dim);
    if (center) vectNd_sub(v, center, &tmp);
    else vectNd_copy(&tmp, v);
    const double c = cos(angle), s = sin(angle);
    for (int r = 0; r < dim; ++r) {
        double sum = 0;
        for (int k = 0; k < dim; ++k) {
            double m = (r == k) ? 1.0 : 0.0;
            if (r == i && k == i) m = c;
            if (r == i && k == j) m = -s;
            if (r == j && k == i) m = s;
            if (r == j && k == j) m = c;
            sum += m * tmp.v[k];
        }
        out.v[r] = sum;
    }
    for (int k = 0; k < dim; ++k) {
        tmp.v[k] = out.v[k];
        if (fabs(tmp.v[k]) < EPSILON) tmp.v[k] = 0;        /* snap tiny components (vectNd.c:253) */
    }
    if (center) vectNd_add(&tmp, center, res);
    else vectNd_copy(res, &tmp);
    vectNd_free(&tmp);
    vectNd_free(&out);
    return VECTND_SUCCESS;
}

/* vectNd.c:271-324: rotate within the plane spanned by v1, v2 */
int vectNd_rotate2(vectNd *v, vectNd *center, vectNd *v1, vectNd *v2, double angle, vectNd *res)
{
    int dim = v->n;
    vectNd bx, by, local, px, py, rx, ry;
    if (!res) res = v;
    vectNd_calloc(&bx, dim);
    vectNd_calloc(&by, dim);
    vectNd_orthogonalize(v1, v2, &bx, &by);
    vectNd_calloc(&local, dim);
    if (center) vectNd_sub(v, center, &local);
    else vectNd_copy(&local, v);
    vectNd_calloc(&px, dim);
    vectNd_calloc(&py, dim);
    vectNd_proj(&local, &bx, &px);
    vectNd_proj(&local, &by, &py);
    double x, y;
    vectNd_dot(&px, &bx, &x);
    vectNd_dot(&py, &by, &y);
    vectNd_calloc(&rx, dim);
    vectNd_calloc(&ry, dim);
    vectNd_scale(&bx, x * cos(angle) - y * sin(angle), &rx);
    vectNd_scale(&by, y * cos(angle) + x * sin(angle), &ry);
    vectNd_sub(v, &px, res);
    vectNd_sub(res, &py, res);
    vectNd_add(res, &rx, res);
    vectNd_add(res, &ry, res);
    vectNd_free(&bx); vectNd_free(&by); vectNd_free(&local); vectNd_free(&px); vectNd_free(&py);
    vectNd_free(&rx); vectNd_free(&ry);
    return VECTND_SUCCESS;
}

int vectNd_print(vectNd *v, char *name)
{
    if (name) printf("%s: ", name);
    printf("<");
    for (int i = 0; i < v->n; ++i) printf("%g%s", v->v[i], (i < v->n - 1) ? ", " : "");
    printf(">\n");
    return VECTND_SUCCESS;
}
