/* ndt_flatten.c -- scene graph -> ndt_flat_scene.
 *
 * Does for this host model what ndt's main() does between scene_setup and render_image
 * (ndt.c:1899-1925): fit the top-level bounding spheres, flatten clusters into kd items,
 * build the kd-tree, validate, aim the camera -- and then what the first ray through every
 * object would do lazily in the reference: fit the bounds of flattened children
 * (object.c:609-615) and build hcube faces (hcube.c:164) with their bounds. */
#include "ndt_host_internal.h"
#include <pthread.h>

#define GROW(ptr, n, cap, need, type)                                                  \
    do {                                                                               \
        if ((n) + (need) > (cap)) {                                                    \
            (cap) = ((cap) * 2 > (n) + (need)) ? (cap) * 2 : (n) + (need) + 64;        \
            (ptr) = (type *)realloc((ptr), (size_t)(cap) * sizeof(type));              \
        }                                                                              \
    } while (0)

static int push_vec(ndt_flat_builder *fb, const vectNd *v, int dims)
{
    GROW(fb->vecs, fb->n_vecs, fb->cap_vecs, dims, double);
    int at = (int)fb->n_vecs;
    for (int i = 0; i < dims; ++i) fb->vecs[fb->n_vecs++] = (v && v->v && v->n == dims) ? v->v[i] : 0.0;
    return at;
}
static int push_raw(ndt_flat_builder *fb, const double *p, int dims)
{
    GROW(fb->vecs, fb->n_vecs, fb->cap_vecs, dims, double);
    int at = (int)fb->n_vecs;
    for (int i = 0; i < dims; ++i) fb->vecs[fb->n_vecs++] = p[i];
    return at;
}

static int add_object(ndt_flat_builder *fb, object *o, int parent, int dims, char *err, int err_len)
{
    const int type = ndt_object_type_id(o);
    if (type >= NDT_OBJ_TYPE_COUNT) {
        char tn[OBJ_TYPE_MAX_LEN] = "";
        o->type_name(tn, sizeof(tn));
        const char *file = ndt_object_plugin_file(tn);
        if (file)
            snprintf(err, (size_t)err_len, "object '%s' has type '%s' from the plugin '%s': its intersect() is host code and cannot run on the device", o->name, tn, file);
        else
            snprintf(err, (size_t)err_len, "object '%s' has type '%s', which has no device implementation", o->name, tn);
        return -1;
    }
    if (!ndt_object_has_default_material(o)) {
        snprintf(err, (size_t)err_len, "object '%s' overrides get_color/get_reflect/get_trans", o->name);
        return -1;
    }
    GROW(fb->objects, fb->n_objects, fb->cap_objects, 1, ndt_flat_object);
    ndt_flat_object f;
    memset(&f, 0, sizeof(f));
    f.type = type;
    f.transparent = o->transparent;
    f.parent = parent;
    f.bounds_radius = o->bounds.radius;
    f.bounds_center_off = push_vec(fb, &o->bounds.center, dims);
    f.n_pos = o->n_pos;
    f.pos_off = (int)fb->n_vecs;
    for (int i = 0; i < o->n_pos; ++i) push_vec(fb, &o->pos[i], dims);
    f.n_dir = o->n_dir;
    f.dir_off = (int)fb->n_vecs;
    for (int i = 0; i < o->n_dir; ++i) push_vec(fb, &o->dir[i], dims);
    f.n_size = o->n_size;
    f.size_off = (int)fb->n_sizes;
    GROW(fb->sizes, fb->n_sizes, fb->cap_sizes, o->n_size, double);
    for (int i = 0; i < o->n_size; ++i) fb->sizes[fb->n_sizes++] = o->size[i];
    f.n_flag = o->n_flag;
    f.flag_off = (int)fb->n_flags;
    GROW(fb->flags, fb->n_flags, fb->cap_flags, o->n_flag, int);
    for (int i = 0; i < o->n_flag; ++i) fb->flags[fb->n_flags++] = o->flag[i];
    f.red = o->red; f.green = o->green; f.blue = o->blue;
    f.red_r = o->red_r; f.green_r = o->green_r; f.blue_r = o->blue_r;
    f.refract_index = o->refract_index;
    fb->objects[fb->n_objects] = f;
    return fb->n_objects++;
}

static int count_nodes(const ndt_kd_node *n) { return n ? 1 + count_nodes(n->left) + count_nodes(n->right) : 0; }

/* preorder: a node, its left subtree, its right subtree */
static void flatten_node(ndt_flat_builder *fb, const ndt_kd_node *n, int me)
{
    ndt_flat_kdnode *k = &fb->nodes[me];
    memset(k, 0, sizeof(*k));
    k->dim = n->dim;
    k->boundary = n->boundary;
    if (n->dim >= 0) {
        k->left = me + 1;
        k->right = me + 1 + count_nodes(n->left);
        flatten_node(fb, n->left, k->left);
        flatten_node(fb, n->right, k->right);
    } else {
        k->left = k->right = -1;
        k->first = fb->n_leaf_refs;
        k->num = n->num;
        GROW(fb->leaf_refs, fb->n_leaf_refs, fb->cap_leaf_refs, n->num, int);
        for (int i = 0; i < n->num; ++i) fb->leaf_refs[fb->n_leaf_refs++] = n->ids[i];
    }
}

/* The bounding spheres that are fitted lazily (object.c:609-615: a radius of 0 means "not fitted yet") are fitted here for all
 * kd items and nested faces BEFORE the flattening loops ask for them, on `threads` threads: an object's fit (bounds_list_optimal,
 * a Nelder-Mead search over its bounding points) reads and writes that object alone, and the loops below find every radius
 * already non-zero -- the same numbers, the same order of everything else.  6 560 fits for the 8-D hypercube: 1.1 s on one
 * thread. */
typedef struct { object **objs; int n, begin, step; } fit_job;
static int cmp_object_ptr(const void *a, const void *b)
{
    const object *x = *(object *const *)a, *y = *(object *const *)b;
    return (x > y) - (x < y);
}

static void *fit_worker(void *arg)
{
    fit_job *j = (fit_job *)arg;
    for (int i = j->begin; i < j->n; i += j->step)
        if (j->objs[i]->bounds.radius == 0) object_get_bounds(j->objs[i]);
    return NULL;
}
static void fit_bounds_parallel(object **objs, int n, int threads)
{
    if (threads > 64) threads = 64;
    if (threads < 2 || n < 64) return;          /* (the loops fit what is left) */
    pthread_t th[64];
    fit_job jobs[64];
    int started = 0;
    for (int k = 0; k < threads; ++k) {
        jobs[k].objs = objs; jobs[k].n = n; jobs[k].begin = k; jobs[k].step = threads;
        if (pthread_create(&th[k], NULL, fit_worker, &jobs[k]) != 0) break;
        ++started;
    }
    for (int k = 0; k < started; ++k) pthread_join(th[k], NULL);
    /* (a thread that could not be started leaves its share to the loops) */
}

int ndt_flatten_scene(scene *scn, ndt_flat_builder *fb, char *err, int err_len)
{
    return ndt_flatten_scene_mt(scn, fb, err, err_len, 1);
}

int ndt_flatten_scene_mt(scene *scn, ndt_flat_builder *fb, char *err, int err_len, int threads)
{
    const int dims = scn->dimensions;
    memset(fb, 0, sizeof(*fb));
    if (err_len > 0) err[0] = '\0';

    /* ndt.c:1899-1908 */
    ndt_kd_tree kd;
    ndt_kd_init(&kd, dims);
    for (int i = 0; i < scn->num_objects; ++i) {
        /* (ndt.c:1905 fits every top-level object.  A cluster's own sphere -- one search over the bounding points of all its
         * members, 172 000 of them for the 8-D hypercube -- is never looked at again: the kd-tree takes its members one by
         * one, object.c:633-681, and nothing else of this path reads it.  Not fitted.) */
        if (ndt_object_type_id(scn->object_ptrs[i]) != NDT_TYPE_CLUSTER) object_get_bounds(scn->object_ptrs[i]);
        ndt_kd_add_object(&kd, scn->object_ptrs[i]);
    }
    ndt_kd_build(&kd);
    scene_validate_objects(scn);        /* ndt.c:1913 */
    camera_aim(&scn->cam);              /* ndt.c:1925 */

    int rc = 0;
    if (threads > 1) {
        /* the lazy fits of the two loops below, ahead of them and in parallel */
        int n_fit = 0, cap_fit = kd.n_items;
        object **fit = (object **)malloc((size_t)(cap_fit > 0 ? cap_fit : 1) * sizeof(object *));
        for (int i = 0; i < kd.n_items; ++i) {
            object *o = kd.items[i].obj;
            if (ndt_object_type_id(o) == NDT_OBJ_HCUBE) {
                ndt_hcube_prepare(o);       /* (idempotent: the loop below calls it again) */
                for (int k = 0; k < o->n_obj; ++k) {
                    if (n_fit == cap_fit) { cap_fit = cap_fit * 2 + 64; fit = (object **)realloc(fit, (size_t)cap_fit * sizeof(object *)); }
                    fit[n_fit++] = o->obj[k];
                }
            }
            else {
                if (n_fit == cap_fit) { cap_fit = cap_fit * 2 + 64; fit = (object **)realloc(fit, (size_t)cap_fit * sizeof(object *)); }
                fit[n_fit++] = o;
            }
            /* (an hcube's own sphere is left to the loop below: it is fitted after its faces, on this thread) */
        }
        /* an object reachable twice (the same pointer in two clusters, or as a scene object and a cluster member) must be fitted
         * by ONE thread: the reference serialises the lazy fit under a lock (object.c:610).  Sort the pointers, keep one of each. */
        qsort(fit, (size_t)n_fit, sizeof(object *), cmp_object_ptr);
        int n_uniq = 0;
        for (int i = 0; i < n_fit; ++i)
            if (n_uniq == 0 || fit[n_uniq - 1] != fit[i]) fit[n_uniq++] = fit[i];
        fit_bounds_parallel(fit, n_uniq, threads);
        free(fit);
    }
    /* kd items first, in id order: the visit mask is indexed by this position */
    for (int i = 0; i < kd.n_items && rc == 0; ++i) {
        object *o = kd.items[i].obj;
        if (ndt_object_type_id(o) == NDT_OBJ_HCUBE) ndt_hcube_prepare(o);      /* resets its radius to 0 */
        if (o->bounds.radius == 0) object_get_bounds(o);                        /* object.c:609-615 */
        if (add_object(fb, o, -1, dims, err, err_len) < 0) rc = -1;
    }
    /* then the nested primitives */
    for (int i = 0; i < kd.n_items && rc == 0; ++i) {
        object *o = kd.items[i].obj;
        if (ndt_object_type_id(o) != NDT_OBJ_HCUBE) continue;
        fb->objects[i].obj_off = (int)fb->n_refs;
        fb->objects[i].n_obj = o->n_obj;
        for (int k = 0; k < o->n_obj && rc == 0; ++k) {
            object *face = o->obj[k];
            if (face->bounds.radius == 0) object_get_bounds(face);
            int idx = add_object(fb, face, i, dims, err, err_len);
            if (idx < 0) { rc = -1; break; }
            GROW(fb->refs, fb->n_refs, fb->cap_refs, 1, int);
            fb->refs[fb->n_refs++] = idx;
        }
    }
    if (rc == 0) {
        fb->n_nodes = count_nodes(kd.root);
        fb->nodes = (ndt_flat_kdnode *)calloc((size_t)(fb->n_nodes > 0 ? fb->n_nodes : 1), sizeof(ndt_flat_kdnode));
        if (kd.root) flatten_node(fb, kd.root, 0);
        fb->n_inf = kd.n_inf;
        fb->inf_refs = (int *)malloc((size_t)(kd.n_inf > 0 ? kd.n_inf : 1) * sizeof(int));
        memcpy(fb->inf_refs, kd.inf_ids, (size_t)kd.n_inf * sizeof(int));

        ndt_flat_scene *fs = &fb->fs;
        fs->abi_version = NDT_HIP_ABI_VERSION;
        fs->dims = dims;
        fs->bb_lower_off = push_raw(fb, kd.bb_lower, dims);
        fs->bb_upper_off = push_raw(fb, kd.bb_upper, dims);
        fs->cam_type = (int)scn->cam.type;
        fs->cam_focal_distance = scn->cam.focal_distance;
        fs->cam_pos_off = push_vec(fb, &scn->cam.pos, dims);
        fs->cam_img_orig_off = push_vec(fb, &scn->cam.imgOrig, dims);
        fs->cam_dir_x_off = push_vec(fb, &scn->cam.dirX, dims);
        fs->cam_dir_y_off = push_vec(fb, &scn->cam.dirY, dims);
        /* the rest of the camera (ABI 2): depth of field, stereo eyes, VR / panorama axes */
        fs->cam_aperture_radius = scn->cam.aperture_radius;
        fs->cam_h_fov = scn->cam.hFov;
        fs->cam_v_fov = scn->cam.vFov;
        fs->cam_left_eye_off = push_vec(fb, &scn->cam.leftEye, dims);
        fs->cam_right_eye_off = push_vec(fb, &scn->cam.rightEye, dims);
        fs->cam_local_x_off = push_vec(fb, &scn->cam.localX, dims);
        fs->cam_local_y_off = push_vec(fb, &scn->cam.localY, dims);
        fs->cam_local_z_off = push_vec(fb, &scn->cam.localZ, dims);
        fs->ambient[0] = scn->ambient.red; fs->ambient[1] = scn->ambient.green; fs->ambient[2] = scn->ambient.blue;
        fs->background[0] = scn->bg_red; fs->background[1] = scn->bg_green; fs->background[2] = scn->bg_blue;
        fs->background[3] = scn->bg_alpha;
        fb->n_lights = scn->num_lights;
        fb->lights = (ndt_flat_light *)calloc((size_t)(scn->num_lights > 0 ? scn->num_lights : 1), sizeof(ndt_flat_light));
        for (int i = 0; i < scn->num_lights; ++i) {
            light *l = scn->lights[i];
            ndt_flat_light *fl = &fb->lights[i];
            fl->type = (int)l->type;
            fl->red = l->red; fl->green = l->green; fl->blue = l->blue;
            fl->angle = l->angle;
            fl->pos_off = (l->pos.v && l->pos.n == dims) ? push_vec(fb, &l->pos, dims) : -1;
            fl->dir_off = (l->dir.v && l->dir.n == dims) ? push_vec(fb, &l->dir, dims) : -1;
            fl->area_off = -1;
            fl->radius = 0.0;
            if (l->type == LIGHT_DISK || l->type == LIGHT_RECT) {
                /* the basis the first shading evaluation would derive (ndt.c:123-125, scene.c:182-195) */
                if (!l->prepared) scene_prepare_light(l);
                fl->area_off = push_vec(fb, &l->u1, dims);
                push_vec(fb, &l->v1, dims);
                fl->radius = l->radius;
            }
        }
        fs->vecs = fb->vecs;         fs->n_vecs = fb->n_vecs;
        fs->sizes = fb->sizes;       fs->n_sizes = fb->n_sizes;
        fs->flags = fb->flags;       fs->n_flags = fb->n_flags;
        fs->obj_refs = fb->refs;     fs->n_obj_refs = fb->n_refs;
        fs->lights = fb->lights;     fs->n_lights = fb->n_lights;
        fs->objects = fb->objects;   fs->n_objects = fb->n_objects;   fs->n_items = kd.n_items;
        fs->kd_nodes = fb->nodes;    fs->n_kd_nodes = fb->n_nodes;
        fs->leaf_refs = fb->leaf_refs; fs->n_leaf_refs = fb->n_leaf_refs;
        fs->inf_refs = fb->inf_refs; fs->n_inf = fb->n_inf;
    }
    ndt_kd_free(&kd);
    return rc;
}

void ndt_flat_builder_free(ndt_flat_builder *fb)
{
    free(fb->vecs); free(fb->sizes); free(fb->flags); free(fb->refs); free(fb->objects); free(fb->lights);
    free(fb->nodes); free(fb->leaf_refs); free(fb->inf_refs);
    memset(fb, 0, sizeof(*fb));
}

/* ---- the `ndtscene 1` text form (same grammar oracle/ref_shim.c writes) ---- */

static const char *type_names[] = { "sphere", "hplane", "hdisk", "cylinder", "hcylinder", "orthotope", "hcube", "hfacet",
                                    "facet" };

static void put(FILE *f, const char *key, const double *p, int dims)
{
    fprintf(f, "%s", key);
    for (int i = 0; i < dims; ++i) fprintf(f, " %a", p ? p[i] : 0.0);
    fprintf(f, "\n");
}

int ndt_write_ndtscene(const ndt_flat_scene *fs, const char *name, const char *path)
{
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    const int d = fs->dims;
    /* version 2 adds the camera2 block; plain pinhole cameras are written as version 1, the form
     * the fixtures flattened from the compiled reference have (oracle/ref_shim.c:dump_scene) */
    int area_lights = 0;
    for (int i = 0; i < fs->n_lights; ++i) area_lights |= fs->lights[i].area_off >= 0;
    const int v2 = (fs->cam_type != 0 || fs->cam_aperture_radius != 0.0 || area_lights || getenv("NDT_NDTSCENE_V2")) &&
                   fs->cam_left_eye_off >= 0 && fs->cam_right_eye_off >= 0 && fs->cam_local_x_off >= 0 &&
                   fs->cam_local_y_off >= 0 && fs->cam_local_z_off >= 0;
    fprintf(f, "ndtscene %d\nname %s\ndims %d\n", v2 ? 2 : 1, name, d);
    fprintf(f, "camera type %d focal_distance %a\n", fs->cam_type, fs->cam_focal_distance);
    put(f, "cam_pos", fs->vecs + fs->cam_pos_off, d);
    put(f, "cam_img_orig", fs->vecs + fs->cam_img_orig_off, d);
    put(f, "cam_dir_x", fs->vecs + fs->cam_dir_x_off, d);
    put(f, "cam_dir_y", fs->vecs + fs->cam_dir_y_off, d);
    if (v2) {
        fprintf(f, "camera2 aperture %a hfov %a vfov %a\n", fs->cam_aperture_radius, fs->cam_h_fov, fs->cam_v_fov);
        put(f, "cam_left_eye", fs->vecs + fs->cam_left_eye_off, d);
        put(f, "cam_right_eye", fs->vecs + fs->cam_right_eye_off, d);
        put(f, "cam_local_x", fs->vecs + fs->cam_local_x_off, d);
        put(f, "cam_local_y", fs->vecs + fs->cam_local_y_off, d);
        put(f, "cam_local_z", fs->vecs + fs->cam_local_z_off, d);
    }
    fprintf(f, "ambient %a %a %a\n", fs->ambient[0], fs->ambient[1], fs->ambient[2]);
    fprintf(f, "background %a %a %a %a\n", fs->background[0], fs->background[1], fs->background[2], fs->background[3]);
    fprintf(f, "lights %d\n", fs->n_lights);
    for (int i = 0; i < fs->n_lights; ++i) {
        const ndt_flat_light *l = &fs->lights[i];
        fprintf(f, "light %d type %d color %a %a %a angle %a has_pos %d has_dir %d", i, l->type, l->red, l->green, l->blue,
                l->angle, l->pos_off >= 0, l->dir_off >= 0);
        if (v2) fprintf(f, " radius %a has_area %d", l->radius, l->area_off >= 0);
        fprintf(f, "\n");
        put(f, "lpos", l->pos_off >= 0 ? fs->vecs + l->pos_off : NULL, d);
        put(f, "ldir", l->dir_off >= 0 ? fs->vecs + l->dir_off : NULL, d);
        if (v2 && l->area_off >= 0) {
            put(f, "lu1", fs->vecs + l->area_off, d);
            put(f, "lv1", fs->vecs + l->area_off + d, d);
        }
    }
    fprintf(f, "objects %d items %d\n", fs->n_objects, fs->n_items);
    for (int i = 0; i < fs->n_objects; ++i) {
        const ndt_flat_object *o = &fs->objects[i];
        fprintf(f, "object %d type %s parent %d transparent %d npos %d ndir %d nsize %d nflag %d nobj %d\n", i,
                type_names[o->type], o->parent, o->transparent, o->n_pos, o->n_dir, o->n_size, o->n_flag, o->n_obj);
        fprintf(f, "material %a %a %a %a %a %a %a\n", o->red, o->green, o->blue, o->red_r, o->green_r, o->blue_r,
                o->refract_index);
        fprintf(f, "bounds %a", o->bounds_radius);
        for (int k = 0; k < d; ++k) fprintf(f, " %a", fs->vecs[o->bounds_center_off + k]);
        fprintf(f, "\n");
        for (int k = 0; k < o->n_pos; ++k) put(f, "pos", fs->vecs + o->pos_off + k * d, d);
        for (int k = 0; k < o->n_dir; ++k) put(f, "dir", fs->vecs + o->dir_off + k * d, d);
        fprintf(f, "sizes");
        for (int k = 0; k < o->n_size; ++k) fprintf(f, " %a", fs->sizes[o->size_off + k]);
        fprintf(f, "\nflags");
        for (int k = 0; k < o->n_flag; ++k) fprintf(f, " %d", fs->flags[o->flag_off + k]);
        fprintf(f, "\nchildren");
        for (int k = 0; k < o->n_obj; ++k) fprintf(f, " %d", fs->obj_refs[o->obj_off + k]);
        fprintf(f, "\n");
    }
    fprintf(f, "kdtree nodes %d obj_num %d\n", fs->n_kd_nodes, fs->n_items);
    for (int i = 0; i < fs->n_kd_nodes; ++i) {
        const ndt_flat_kdnode *k = &fs->kd_nodes[i];
        fprintf(f, "kdnode %d dim %d boundary %a left %d right %d num %d ids", i, k->dim, k->boundary, k->left, k->right,
                k->dim < 0 ? k->num : 0);
        if (k->dim < 0)
            for (int j = 0; j < k->num; ++j) fprintf(f, " %d", fs->leaf_refs[k->first + j]);
        fprintf(f, "\n");
    }
    fprintf(f, "inf %d ids", fs->n_inf);
    for (int i = 0; i < fs->n_inf; ++i) fprintf(f, " %d", fs->inf_refs[i]);
    fprintf(f, "\n");
    put(f, "bb_lower", fs->vecs + fs->bb_lower_off, d);
    put(f, "bb_upper", fs->vecs + fs->bb_upper_off, d);
    fprintf(f, "end\n");
    fclose(f);
    return 0;
}
