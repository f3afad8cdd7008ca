/* ndt_bounding.c -- bounding-sphere lists and the minimum enclosing sphere fit
 * (reference bounding.c:89-240), restated. */
#include "ndt_host_api.h"

int bounds_list_init(bounds_list *list)
{
    list->head = list->tail = NULL;
    return 0;
}

/* new nodes go to the FRONT (bounding.c:94-112): list order affects the centroid's rounding
 * and through it the Nelder-Mead seed */
int bounds_list_add(bounds_list *list, vectNd *vect, double radius)
{
    bounds_node *node = (bounds_node *)calloc(1, sizeof(bounds_node));
    if (!node) { perror("calloc"); exit(1); }
    node->next = list->head;
    list->head = node;
    if (!list->tail) list->tail = node;
    vectNd_calloc(&node->bounds.center, vect->n);
    vectNd_copy(&node->bounds.center, vect);
    node->bounds.radius = radius;
    return 0;
}

int bounds_list_join(bounds_list *list, bounds_list *other)
{
    if (list->tail) list->tail->next = other->head;
    else list->head = other->head;
    list->tail = other->tail;
    other->head = other->tail = NULL;
    return 0;
}

int bounds_list_free(bounds_list *list)
{
    for (bounds_node *n = list->head; n;) {
        bounds_node *next = n->next;
        vectNd_free(&n->bounds.center);
        free(n);
        n = next;
    }
    list->head = list->tail = NULL;
    return 0;
}

int bounds_list_centroid(bounds_list *list, vectNd *centroid)
{
    vectNd sum;
    int count = 0;
    vectNd_calloc(&sum, centroid->n);
    for (bounds_node *n = list->head; n; n = n->next) {
        vectNd_add(&sum, &n->bounds.center, &sum);
        ++count;
    }
    vectNd_scale(&sum, 1.0 / count, centroid);
    vectNd_free(&sum);
    return 0;
}

int bounds_list_radius(bounds_list *list, vectNd *centroid, double *radius)
{
    double max = -1.0;
    for (bounds_node *n = list->head; n; n = n->next) {
        double dist = -1.0;
        vectNd_dist(centroid, &n->bounds.center, &dist);
        if (n->bounds.radius > 0.0) dist += n->bounds.radius;
        max = (dist > max) ? dist : max;
    }
    *radius = max;
    return 0;
}

/* bounding.c:177-240: Nelder-Mead over the centre, seeded with the centroid, at most 1000
 * iterations or until the simplex is smaller than EPSILON; the centroid wins if the search
 * ends worse than it started */
int bounds_list_optimal(bounds_list *list, vectNd *centroid, double *radius)
{
    int dim = centroid->n;
    void *nm = NULL;
    double curr_radius = -1.0;
    vectNd curr, initial;
    nm_init(&nm, dim);
    vectNd_calloc(&curr, dim);
    bounds_list_centroid(list, &curr);
    bounds_list_radius(list, &curr, &curr_radius);
    nm_set_seed(nm, &curr);
    vectNd_calloc(&initial, dim);
    vectNd_copy(&initial, &curr);
    const double initial_radius = curr_radius;
    while (!nm_done(nm, EPSILON, 1000)) {
        nm_add_result(nm, &curr, curr_radius);
        nm_next_point(nm, &curr);
        bounds_list_radius(list, &curr, &curr_radius);
    }
    nm_best_point(nm, &curr);
    bounds_list_radius(list, &curr, &curr_radius);
    if (curr_radius - initial_radius > EPSILON) {
        vectNd_copy(&curr, &initial);
        bounds_list_radius(list, &curr, &curr_radius);
    }
    vectNd_copy(centroid, &curr);
    *radius = curr_radius;
    vectNd_free(&initial);
    vectNd_free(&curr);
    nm_free(nm);
    return 0;
}

/* bounding.c:34-85.  Host copy of the gate the device evaluates per ray. */
int vect_bounding_sphere_intersect(bounding_sphere *sph, vectNd *o, vectNd *v, double min_dist)
{
    if (!sph->prepared) {
        sph->radius_sqr = sph->radius * sph->radius;
        sph->prepared = 1;
    }
    vectNd oc;
    double oc_len2, voc;
    vectNd_alloc(&oc, o->n);
    vectNd_sub(o, &sph->center, &oc);
    vectNd_dot(&oc, &oc, &oc_len2);
    if (min_dist > 0) {
        double reach = min_dist + sph->radius;
        if (oc_len2 > reach * reach) {
            vectNd_free(&oc);
            return 0;
        }
    }
    vectNd_dot(v, &oc, &voc);
    vectNd_free(&oc);
    double voc2 = voc * voc;
    double desc = voc2 - oc_len2 + sph->radius_sqr;
    if (desc < 0.0 || (voc > 0.0 && voc2 > desc)) return 0;
    return 1;
}
