/* ndt_host_internal.h -- shared between the host sources; not part of the scene API. */
#ifndef NDT_HOST_INTERNAL_H
#define NDT_HOST_INTERNAL_H
#include "ndt_host_api.h"
#include "../../../include/ndt_hip.h"

#define NDT_TYPE_CLUSTER 100
#define NDT_TYPE_OTHER 101

const char *ndt_object_plugin_file(const char *type);    /* file of a host-side-only plugin type, or NULL */
int ndt_object_type_id(object *o);                  /* NDT_OBJ_* for device types, else NDT_TYPE_* */
int ndt_object_has_default_material(object *o);
void ndt_hcube_prepare(object *cube);

/* ---- kd-tree as the reference builds it (kd-tree.c:294-477) ---- */
typedef struct {
    double *lower, *upper;          /* dims each */
    int id;
    object *obj;
} ndt_kd_item;

typedef struct ndt_kd_node {
    int dim;                        /* split dimension, -1 for a leaf */
    double boundary;
    int num;                        /* leaf: number of items */
    int *ids;                       /* leaf: item ids, in list order */
    struct ndt_kd_node *left, *right;
} ndt_kd_node;

typedef struct {
    int dims;
    ndt_kd_item *items; int n_items, cap_items;
    int *inf_ids; int n_inf;
    double *bb_lower, *bb_upper;
    ndt_kd_node *root;
} ndt_kd_tree;

void ndt_kd_init(ndt_kd_tree *t, int dims);
void ndt_kd_add_object(ndt_kd_tree *t, object *obj);        /* object_kdlist_add, object.c:633-681 */
void ndt_kd_build(ndt_kd_tree *t);                          /* kd_tree_build, kd-tree.c:421-477 */
void ndt_kd_free(ndt_kd_tree *t);

/* ---- flattening (the reference-side stub of INTEGRATION.md, against this host model) ---- */
typedef struct {
    ndt_flat_scene fs;
    double *vecs;   long n_vecs, cap_vecs;
    double *sizes;  long n_sizes, cap_sizes;
    int *flags;     long n_flags, cap_flags;
    int *refs;      long n_refs, cap_refs;
    ndt_flat_object *objects; int n_objects, cap_objects;
    ndt_flat_light *lights;   int n_lights;
    ndt_flat_kdnode *nodes;   int n_nodes, cap_nodes;
    int *leaf_refs; int n_leaf_refs, cap_leaf_refs;
    int *inf_refs;  int n_inf;
} ndt_flat_builder;

/* Builds bounds + kd-tree exactly like ndt.c:1899-1908, aims the camera (ndt.c:1925) and
 * flattens.  Returns 0, or -1 with a message in `err` when the scene cannot go to the device. */
int ndt_flatten_scene(scene *scn, ndt_flat_builder *fb, char *err, int err_len);
int ndt_flatten_scene_mt(scene *scn, ndt_flat_builder *fb, char *err, int err_len, int threads);
void ndt_flat_builder_free(ndt_flat_builder *fb);
int ndt_write_ndtscene(const ndt_flat_scene *fs, const char *name, const char *path);

#endif
