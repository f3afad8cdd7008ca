/* ndt_kdtree.c -- kd-tree construction exactly as the reference does it (kd-tree.c:16-81,
 * 294-477 and object.c:633-681), restated.  The tree is an *input contract* of the renderer:
 * tree shape and leaf order decide which object wins EPSILON ties (SURVEY.md 8a rows T1-T3,
 * O-kd), so the exhaustive split search, its candidate order and its first-strictly-best
 * rule are kept, not improved. */
#include <float.h>

#include "ndt_host_internal.h"

void ndt_kd_init(ndt_kd_tree *t, int dims)
{
    memset(t, 0, sizeof(*t));
    t->dims = dims;
    t->bb_lower = (double *)malloc((size_t)dims * sizeof(double));
    t->bb_upper = (double *)malloc((size_t)dims * sizeof(double));
    for (int i = 0; i < dims; ++i) {
        t->bb_lower[i] = DBL_MAX;           /* aabb_init, kd-tree.c:16-22 */
        t->bb_upper[i] = -DBL_MAX;
    }
}

/* aabb_add_point, kd-tree.c:63-81: grows by EPSILON beyond the point, compared against the
 * already padded bound */
static void box_add_point(double *lower, double *upper, const double *p, int dims)
{
    for (int i = 0; i < dims; ++i) {
        if (p[i] < lower[i]) lower[i] = p[i] - EPSILON;
        if (p[i] > upper[i]) upper[i] = p[i] + EPSILON;
    }
}

/* object_kdlist_add, object.c:633-681: clusters are flattened recursively; every other object
 * becomes one item whose box covers its bounding points +/- |radius| */
void ndt_kd_add_object(ndt_kd_tree *t, object *obj)
{
    if (ndt_object_type_id(obj) == NDT_TYPE_CLUSTER) {
        for (int i = 0; i < obj->n_obj; ++i) ndt_kd_add_object(t, obj->obj[i]);
        return;
    }
    if (t->n_items >= t->cap_items) {
        t->cap_items = t->cap_items * 2 + 16;
        t->items = (ndt_kd_item *)realloc(t->items, (size_t)t->cap_items * sizeof(ndt_kd_item));
    }
    const int dims = obj->dimensions;
    ndt_kd_item *it = &t->items[t->n_items];
    it->lower = (double *)malloc((size_t)dims * sizeof(double));
    it->upper = (double *)malloc((size_t)dims * sizeof(double));
    for (int i = 0; i < dims; ++i) {
        it->lower[i] = DBL_MAX;
        it->upper[i] = -DBL_MAX;
    }
    bounds_list pts;
    bounds_list_init(&pts);
    obj->bounding_points(obj, &pts);
    double *p = (double *)malloc((size_t)dims * sizeof(double));
    for (bounds_node *n = pts.head; n; n = n->next) {
        const double r = fabs(n->bounds.radius);
        for (int i = 0; i < dims; ++i) p[i] = n->bounds.center.v[i] + r;
        box_add_point(it->lower, it->upper, p, dims);
        for (int i = 0; i < dims; ++i) p[i] = n->bounds.center.v[i] - r;
        box_add_point(it->lower, it->upper, p, dims);
    }
    free(p);
    bounds_list_free(&pts);
    it->obj = obj;
    it->id = t->n_items;        /* kd_tree_build renumbers to the list position anyway (kd-tree.c:448) */
    t->n_items += 1;
}

/* kdtree_split_score, kd-tree.c:294-313 */
static int split_score(const ndt_kd_tree *t, const int *ids, int n, int dim, double pos, double *score)
{
    int left = 0, right = 0, straddle = 0;
    for (int i = 0; i < n; ++i) {
        const ndt_kd_item *it = &t->items[ids[i]];
        if (it->upper[dim] < pos - EPSILON) ++left;
        else if (it->lower[dim] > pos + EPSILON) ++right;
        else ++straddle;
    }
    *score = n - (abs(left - right) + 2 * straddle);
    return (left > 0 && right > 0) ? 1 : 0;
}

/* kd_tree_split_node, kd-tree.c:315-419 */
static ndt_kd_node *split_node(const ndt_kd_tree *t, const int *ids, int n)
{
    ndt_kd_node *node = (ndt_kd_node *)calloc(1, sizeof(ndt_kd_node));
    int found = 0, split_dim = -1;
    double split_pos = 0.0, score = -DBL_MAX, best = -DBL_MAX;
    for (int dim = 0; dim < t->dims; ++dim) {
        for (int i = 0; i < n; ++i) {
            const ndt_kd_item *it = &t->items[ids[i]];
            double cand = it->lower[dim] - 2 * EPSILON;
            if (split_score(t, ids, n, dim, cand, &score) && score > best) {
                split_dim = dim; split_pos = cand; best = score; found = 1;
            }
            cand = it->upper[dim] + 2 * EPSILON;
            if (split_score(t, ids, n, dim, cand, &score) && score > best) {
                split_dim = dim; split_pos = cand; best = score; found = 1;
            }
        }
    }
    if (!found) {
        node->dim = -1;
        node->num = n;
        node->ids = (int *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
        memcpy(node->ids, ids, (size_t)n * sizeof(int));
        return node;
    }
    node->dim = split_dim;
    node->boundary = split_pos;
    int *l = (int *)malloc((size_t)n * sizeof(int)), *r = (int *)malloc((size_t)n * sizeof(int));
    int nl = 0, nr = 0;
    for (int i = 0; i < n; ++i) {
        const ndt_kd_item *it = &t->items[ids[i]];
        if (it->obj->bounds.radius < 0.0) continue;             /* kd-tree.c:385-389 */
        if (it->upper[split_dim] < split_pos - EPSILON) l[nl++] = ids[i];
        else if (it->lower[split_dim] > split_pos + EPSILON) r[nr++] = ids[i];
        else { l[nl++] = ids[i]; r[nr++] = ids[i]; }            /* straddlers go to both */
    }
    if (nl > 0 && nr > 0) {
        node->left = split_node(t, l, nl);
        node->right = split_node(t, r, nr);
    } else {
        /* cannot happen for a valid split; the reference would leave two empty children */
        node->dim = -1;
        node->num = n;
        node->ids = (int *)malloc((size_t)n * sizeof(int));
        memcpy(node->ids, ids, (size_t)n * sizeof(int));
    }
    free(l);
    free(r);
    return node;
}

/* kd_tree_build, kd-tree.c:421-477 */
void ndt_kd_build(ndt_kd_tree *t)
{
    int *finite = (int *)malloc((size_t)(t->n_items > 0 ? t->n_items : 1) * sizeof(int));
    int nf = 0;
    t->inf_ids = (int *)malloc((size_t)(t->n_items > 0 ? t->n_items : 1) * sizeof(int));
    t->n_inf = 0;
    for (int i = 0; i < t->n_items; ++i) {
        ndt_kd_item *it = &t->items[i];
        it->id = i;
        if (it->obj->bounds.radius >= 0.0) {
            finite[nf++] = i;
            for (int k = 0; k < t->dims; ++k) {                 /* aabb_add, kd-tree.c:42-61 */
                if (it->lower[k] < t->bb_lower[k]) t->bb_lower[k] = it->lower[k];
                if (it->upper[k] > t->bb_upper[k]) t->bb_upper[k] = it->upper[k];
            }
        } else {
            t->inf_ids[t->n_inf++] = i;
        }
    }
    t->root = split_node(t, finite, nf);
    free(finite);
}

static void free_node(ndt_kd_node *n)
{
    if (!n) return;
    free_node(n->left);
    free_node(n->right);
    free(n->ids);
    free(n);
}

void ndt_kd_free(ndt_kd_tree *t)
{
    for (int i = 0; i < t->n_items; ++i) {
        free(t->items[i].lower);
        free(t->items[i].upper);
    }
    free(t->items);
    free(t->inf_ids);
    free(t->bb_lower);
    free(t->bb_upper);
    free_node(t->root);
    memset(t, 0, sizeof(*t));
}
