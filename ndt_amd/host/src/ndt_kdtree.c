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

/* The same score for every candidate of one dimension without looking at every item every time: the reference scores each of
 * its 2n candidates per dimension with a pass over the n items (kd-tree.c:294-313, 330-360) -- 689 M item tests for the root of
 * the 8-D hypercube's 6 560 objects, 7 s of an 8 s scene preparation.  `left` is the number of upper bounds below pos - EPSILON
 * and, for an item with lower <= upper, `right` the number of lower bounds above pos + EPSILON: two binary searches in the
 * node's sorted bounds give the counts the pass gives (the comparisons are the pass's own, on the same doubles).  Items whose
 * box is inverted in a dimension (an infinite hcylinder among the flattened children of a cluster has no bounding points: its
 * box stays at lower = DBL_MAX, upper = -DBL_MAX, object.c:633-681) could count on both sides: they are kept apart, by
 * descending upper bound, and go through the pass's own if / else-if.  A NaN bound anywhere: the pass itself runs.  The items
 * are sorted once, at the root; a child's order is its parent's with the other side's items left out.  Candidates are visited
 * in the reference's order and the first best one wins, as there: the tree is the reference's, node for node
 * (tests/test_host_api.py compares the flattened scenes byte for byte). */
typedef struct {
    int *lo, *up;               /* the items with lower <= upper in this dimension, ascending by lower / by upper bound */
    int *inv;                   /* the others, descending by upper bound */
    int n_reg, n_inv;
} kd_dim_order;
typedef struct { const ndt_kd_tree *t; int dim, upper, descending; } kd_sort_ctx;
static int cmp_item_bound(const void *a, const void *b, void *c)
{
    const kd_sort_ctx *k = (const kd_sort_ctx *)c;
    const ndt_kd_item *ia = &k->t->items[*(const int *)a], *ib = &k->t->items[*(const int *)b];
    const double x = k->upper ? ia->upper[k->dim] : ia->lower[k->dim], y = k->upper ? ib->upper[k->dim] : ib->lower[k->dim];
    const int r = (x > y) - (x < y);
    return k->descending ? -r : r;
}
/* number of elements of the ascending array v[0..n) that are < x */
static int count_below(const double *v, int n, double x)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (v[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
/* number of elements of the ascending array v[0..n) that are > x */
static int count_above(const double *v, int n, double x)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (v[mid] > x) hi = mid; else lo = mid + 1;
    }
    return n - lo;
}
#define KD_SORTED_MIN 32        /* below this many items the pass is as fast */

static void free_order(kd_dim_order *o, int dims)
{
    if (!o) return;
    for (int k = 0; k < dims; ++k) { free(o[k].lo); free(o[k].up); free(o[k].inv); }
    free(o);
}
/* the order of the items with `bit` set in side[] among a parent's */
static kd_dim_order *child_order(const kd_dim_order *parent, int dims, const unsigned char *side, unsigned char bit, int n_child)
{
    kd_dim_order *o = (kd_dim_order *)calloc((size_t)dims, sizeof(kd_dim_order));
    for (int k = 0; k < dims; ++k) {
        const kd_dim_order *p = &parent[k];
        o[k].lo = (int *)malloc((size_t)(n_child > 0 ? n_child : 1) * sizeof(int));
        o[k].up = (int *)malloc((size_t)(n_child > 0 ? n_child : 1) * sizeof(int));
        o[k].inv = (int *)malloc((size_t)(p->n_inv > 0 ? p->n_inv : 1) * sizeof(int));
        int a = 0, b = 0, c = 0;
        for (int i = 0; i < p->n_reg; ++i) {
            if (side[p->lo[i]] & bit) o[k].lo[a++] = p->lo[i];
            if (side[p->up[i]] & bit) o[k].up[b++] = p->up[i];
        }
        for (int i = 0; i < p->n_inv; ++i)
            if (side[p->inv[i]] & bit) o[k].inv[c++] = p->inv[i];
        o[k].n_reg = a;
        o[k].n_inv = c;
    }
    return o;
}

/* kd_tree_split_node, kd-tree.c:315-419.  order: NULL, or the node's items per dimension (above); side: scratch, one byte per
 * item of the tree */
static ndt_kd_node *split_node(const ndt_kd_tree *t, const int *ids, int n, const kd_dim_order *order, unsigned char *side, int *slot_of)
{
    ndt_kd_node *node = (ndt_kd_node *)calloc(1, sizeof(ndt_kd_node));
    int found = 0, split_dim = -1;
    double split_pos = 0.0, score = -DBL_MAX, best = -DBL_MAX;
    const int sorted = order != NULL && n >= KD_SORTED_MIN;
    double *los = sorted ? (double *)malloc((size_t)n * sizeof(double)) : NULL, *ups = sorted ? (double *)malloc((size_t)n * sizeof(double)) : NULL;
    /* scores of the candidates of the regular items, by item (filled per dimension by two sweeps) */
    int *left_of = sorted ? (int *)malloc((size_t)4 * n * sizeof(int)) : NULL;     /* [4 * i + 0 .. 3] for ids[i]: left, right of its lower / upper candidate */
    if (sorted)
        for (int i = 0; i < n; ++i) slot_of[ids[i]] = i;       /* where an item stands in ids[] (tree-sized scratch, like `side`) */
    for (int dim = 0; dim < t->dims; ++dim) {
        const kd_dim_order *od = sorted ? &order[dim] : NULL;
        if (sorted) {
            for (int i = 0; i < od->n_reg; ++i) {
                los[i] = t->items[od->lo[i]].lower[dim];
                ups[i] = t->items[od->up[i]].upper[dim];
            }
            /* The candidates of the regular items in ascending order: both counts only grow, two pointers follow them (the
             * comparisons are count_below's / count_above's).  Lower-bound candidates first, then upper-bound ones. */
            for (int end = 0; end < 2; ++end) {
                int a = 0, b = 0;       /* a = #{ups < x}, b = #{los <= y} */
                for (int k = 0; k < od->n_reg; ++k) {
                    const int id = end == 0 ? od->lo[k] : od->up[k];
                    const double cand = end == 0 ? los[k] - 2 * EPSILON : ups[k] + 2 * EPSILON;
                    const double x = cand - EPSILON, y = cand + EPSILON;
                    while (a < od->n_reg && ups[a] < x) ++a;
                    while (b < od->n_reg && !(los[b] > y)) ++b;
                    left_of[4 * slot_of[id] + 2 * end] = a;
                    left_of[4 * slot_of[id] + 2 * end + 1] = od->n_reg - b;
                }
            }
        }
        for (int i = 0; i < n; ++i) {
            const ndt_kd_item *it = &t->items[ids[i]];
            const int regular = it->lower[dim] <= it->upper[dim];
            for (int end = 0; end < 2; ++end) {
                const double cand = end == 0 ? it->lower[dim] - 2 * EPSILON : it->upper[dim] + 2 * EPSILON;
                int ok;
                if (sorted) {
                    const double x = cand - EPSILON, y = cand + EPSILON;
                    int left, right;
                    if (regular) { left = left_of[4 * i + 2 * end]; right = left_of[4 * i + 2 * end + 1]; }
                    else { left = count_below(ups, od->n_reg, x); right = count_above(los, od->n_reg, y); }
                    /* the inverted ones: descending by upper bound, so those not on the left come first */
                    int j = 0;
                    for (; j < od->n_inv && !(t->items[od->inv[j]].upper[dim] < x); ++j)
                        if (t->items[od->inv[j]].lower[dim] > y) ++right;
                    left += od->n_inv - j;
                    score = n - (abs(left - right) + 2 * (n - left - right));
                    ok = left > 0 && right > 0;
                } else {
                    ok = split_score(t, ids, n, dim, cand, &score);
                }
                if (ok && score > best) {
                    split_dim = dim; split_pos = cand; best = score; found = 1;
                }
            }
        }
    }
    free(left_of);
    free(los);
    free(ups);
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
        side[ids[i]] = 0;
        if (it->obj->bounds.radius < 0.0) continue;             /* kd-tree.c:385-389 */
        if (it->upper[split_dim] < split_pos - EPSILON) { l[nl++] = ids[i]; side[ids[i]] = 1; }
        else if (it->lower[split_dim] > split_pos + EPSILON) { r[nr++] = ids[i]; side[ids[i]] = 2; }
        else { l[nl++] = ids[i]; r[nr++] = ids[i]; side[ids[i]] = 3; }     /* straddlers go to both */
    }
    if (nl > 0 && nr > 0) {
        /* the children's orders (before the recursion reuses `side`) */
        kd_dim_order *lo_order = (sorted && nl >= KD_SORTED_MIN) ? child_order(order, t->dims, side, 1, nl) : NULL;
        kd_dim_order *ro_order = (sorted && nr >= KD_SORTED_MIN) ? child_order(order, t->dims, side, 2, nr) : NULL;
        node->left = split_node(t, l, nl, lo_order, side, slot_of);
        node->right = split_node(t, r, nr, ro_order, side, slot_of);
        free_order(lo_order, t->dims);
        free_order(ro_order, t->dims);
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
    /* the finite items by either bound in every dimension, for split_node's counting; not when a bound is NaN */
    kd_dim_order *order = NULL;
    unsigned char *side = (unsigned char *)calloc((size_t)(t->n_items > 0 ? t->n_items : 1), 1);
    int regular = nf >= KD_SORTED_MIN;
    for (int i = 0; i < nf && regular; ++i)
        for (int k = 0; k < t->dims; ++k)
            if (t->items[finite[i]].lower[k] != t->items[finite[i]].lower[k] || t->items[finite[i]].upper[k] != t->items[finite[i]].upper[k]) regular = 0;
    if (regular) {
        order = (kd_dim_order *)calloc((size_t)t->dims, sizeof(kd_dim_order));
        for (int k = 0; k < t->dims; ++k) {
            kd_dim_order *o = &order[k];
            o->lo = (int *)malloc((size_t)nf * sizeof(int));
            o->up = (int *)malloc((size_t)nf * sizeof(int));
            o->inv = (int *)malloc((size_t)nf * sizeof(int));
            for (int i = 0; i < nf; ++i) {
                const ndt_kd_item *it = &t->items[finite[i]];
                if (it->lower[k] <= it->upper[k]) { o->lo[o->n_reg] = finite[i]; o->up[o->n_reg] = finite[i]; o->n_reg += 1; }
                else o->inv[o->n_inv++] = finite[i];
            }
            kd_sort_ctx c = { t, k, 0, 0 };
            qsort_r(o->lo, (size_t)o->n_reg, sizeof(int), cmp_item_bound, &c);
            c.upper = 1;
            qsort_r(o->up, (size_t)o->n_reg, sizeof(int), cmp_item_bound, &c);
            c.descending = 1;
            qsort_r(o->inv, (size_t)o->n_inv, sizeof(int), cmp_item_bound, &c);
        }
    }
    int *slot_of = (int *)malloc((size_t)(t->n_items > 0 ? t->n_items : 1) * sizeof(int));
    t->root = split_node(t, finite, nf, order, side, slot_of);
    free_order(order, t->dims);
    free(slot_of);
    free(side);
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
