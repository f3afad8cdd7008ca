/* ndt_objects.c -- object registry, the built-in object types' host-side behaviour (parameter
 * counts, bounding points, hcube face construction) and the object_* API (reference object.c
 * and the host half of objects/ *.c), restated.
 *
 * Intersection itself is not here: rays are traced on the GPU by libndt_hip.so.  What the
 * host has to reproduce exactly is everything that decides *which* tests the device makes:
 * bounding points -> bounding spheres and kd item boxes -> tree shape and leaf order. */
#include <strings.h>
#include <dirent.h>
#include <dlfcn.h>
#include <limits.h>

#include "ndt_host_api.h"
#include "ndt_host_internal.h"

/* ------------------------------------------------------------------ defaults (object.c:23-49) */

static int default_color(object *o, vectNd *at, double *r, double *g, double *b)
{
    (void)at;
    *r = o->red; *g = o->green; *b = o->blue;
    return 0;
}
static int default_reflect(object *o, vectNd *at, double *r, double *g, double *b)
{
    (void)at;
    *r = o->red_r; *g = o->green_r; *b = o->blue_r;
    return 0;
}
static int default_trans(object *o, vectNd *at, int *t)
{
    (void)at;
    *t = o->transparent;
    return 0;
}
int ndt_object_has_default_material(object *o)
{
    return o->get_color == default_color && o->get_reflect == default_reflect && o->get_trans == default_trans;
}

/* obj->intersect of every built-in type: the host never traces */
static int host_intersect_stub(object *o, vectNd *a, vectNd *b, vectNd *c, vectNd *d, object **p)
{
    (void)o; (void)a; (void)b; (void)c; (void)d;
    if (p) *p = NULL;
    fprintf(stderr, "ndt host: obj->intersect called on the host; rays are traced by libndt_hip.so\n");
    return 0;
}

/* ------------------------------------------------------------------ per-type: names */

#define DEF_NAME(fn, str) static int fn(char *name, int size) { strncpy(name, str, (size_t)size); return 0; }
DEF_NAME(name_sphere, "sphere")
DEF_NAME(name_hplane, "hplane")
DEF_NAME(name_hdisk, "hdisk")
DEF_NAME(name_cylinder, "cylinder")
DEF_NAME(name_hcylinder, "hcylinder")
DEF_NAME(name_orthotope, "orthotope")
DEF_NAME(name_hcube, "hcube")
DEF_NAME(name_hfacet, "hfacet")
DEF_NAME(name_facet, "facet")
DEF_NAME(name_cluster, "cluster")
DEF_NAME(name_stubs, "stubs")

/* ------------------------------------------------------------------ per-type: parameter counts */

static int set_params(int *n_pos, int *n_dir, int *n_size, int *n_flags, int *n_obj, int p, int d, int s, int f)
{
    *n_pos = p; *n_dir = d; *n_size = s; *n_flags = f; *n_obj = 0;
    return 0;
}
#define PARAMS_ARGS object *o, int *n_pos, int *n_dir, int *n_size, int *n_flags, int *n_obj
#define PARAMS_FWD n_pos, n_dir, n_size, n_flags, n_obj
static int params_sphere(PARAMS_ARGS) { if (!o) return -1; return set_params(PARAMS_FWD, 1, 0, 1, 0); }          /* sphere.c:40 */
static int params_hplane(PARAMS_ARGS) { if (!o) return -1; return set_params(PARAMS_FWD, 1, 1, 0, 0); }          /* hplane.c:16 */
static int params_hdisk(PARAMS_ARGS) { if (!o) return -1; return set_params(PARAMS_FWD, 1, 1, 1, 0); }           /* hdisk.c:40 */
static int params_cylinder(PARAMS_ARGS) { if (!o) return -1; return set_params(PARAMS_FWD, 2, 0, 1, 1); }        /* cylinder.c:58 */
static int params_hcylinder(PARAMS_ARGS) { if (!o) return -1; return set_params(PARAMS_FWD, o->dimensions - 1, 0, 1, 0); }   /* hcylinder.c:76 */
static int params_orthotope(PARAMS_ARGS)                                                                          /* orthotope.c:76 */
{
    if (!o) return -1;
    return set_params(PARAMS_FWD, 1, (o->n_flag > 0) ? o->flag[0] : 1, 0, 1);
}
static int params_hcube(PARAMS_ARGS) { if (!o) return -1; return set_params(PARAMS_FWD, 1, o->dimensions, o->dimensions, 0); }  /* hcube.c:190 */
static int params_hfacet(PARAMS_ARGS) { if (!o) return -1; return set_params(PARAMS_FWD, 3, 3, 0, 1); }          /* hfacet.c:98 */
static int params_facet(PARAMS_ARGS) { if (!o) return -1; return set_params(PARAMS_FWD, 3, 3, 0, 1); }           /* facet.c:90 */
static int params_cluster(PARAMS_ARGS) { if (!o) return -1; return set_params(PARAMS_FWD, 0, 0, 0, 1); }         /* cluster.c:21 */
static int params_stubs(PARAMS_ARGS) { if (!o) return -1; return set_params(PARAMS_FWD, 0, 0, 0, 0); }           /* stubs.c:57 */

/* ------------------------------------------------------------------ per-type: bounding points
 * (inputs of the bounding-sphere fit and of the kd item boxes; SURVEY.md 8a row O-bp) */

static int bp_sphere(object *o, bounds_list *l) { bounds_list_add(l, &o->pos[0], o->size[0]); return 1; }        /* sphere.c:52 */
static int bp_hplane(object *o, bounds_list *l) { return (o && l) ? 1 : -1; }                                      /* infinite */
static int bp_hdisk(object *o, bounds_list *l) { bounds_list_add(l, &o->pos[0], o->size[0]); return 1; }         /* hdisk.c:55 */
static int bp_cylinder(object *o, bounds_list *l)                                                                 /* cylinder.c:73 */
{
    if (o->n_flag < 2 || o->flag[1] == 0) {
        bounds_list_add(l, &o->pos[0], o->size[0]);
        bounds_list_add(l, &o->pos[1], o->size[0]);
    }
    return 1;
}
static int bp_hcylinder(object *o, bounds_list *l)                                                                /* hcylinder.c:91 */
{
    if (o->n_flag > 0 && o->flag[0] == 0)
        for (int i = 0; i < o->n_pos; ++i) bounds_list_add(l, &o->pos[i], o->size[0]);
    return 1;
}
/* corners pos + sum_j bit_j * dir[j] over the m spanning directions (orthotope.c:94-120) */
static int bp_orthotope(object *o, bounds_list *l)
{
    vectNd corner, step;
    vectNd_calloc(&corner, o->dimensions);
    vectNd_calloc(&step, o->dimensions);
    const int m = o->flag[0];
    for (int c = 0; c < (1 << m); ++c) {
        vectNd_copy(&corner, &o->pos[0]);
        int bits = c;
        for (int j = 0; j < m; ++j) {
            int bit = bits % 2;
            bits >>= 1;
            vectNd_scale(&o->dir[j], bit, &step);
            vectNd_add(&corner, &step, &corner);
        }
        bounds_list_add(l, &corner, 0.0);
    }
    vectNd_free(&corner);
    vectNd_free(&step);
    return 1;
}
/* corners pos + sum_j (0.5 - bit_j) * size[j] * dir[j] (hcube.c:206-234) */
static int bp_hcube(object *o, bounds_list *l)
{
    vectNd corner, step;
    const int n = o->dimensions;
    vectNd_calloc(&corner, n);
    vectNd_calloc(&step, n);
    for (int c = 0; c < (1 << n); ++c) {
        vectNd_copy(&corner, &o->pos[0]);
        int bits = c;
        for (int j = 0; j < n; ++j) {
            int bit = bits % 2;
            bits >>= 1;
            vectNd_scale(&o->dir[j], (0.5 - bit) * o->size[j], &step);
            vectNd_add(&corner, &step, &corner);
        }
        bounds_list_add(l, &corner, 0.0);
    }
    vectNd_free(&corner);
    vectNd_free(&step);
    return 1;
}
static int bp_vertices(object *o, bounds_list *l)                                                                 /* hfacet.c:112, facet.c:104 */
{
    for (int i = 0; i < o->n_pos; ++i) bounds_list_add(l, &o->pos[i], 0.0);
    return 1;
}
/* concatenation of the children's points; emptied if any child is infinite; children named
 * "outline" are skipped (cluster.c:35-62) */
static int bp_cluster(object *o, bounds_list *l)
{
    for (int i = 0; i < o->n_obj; ++i) {
        bounds_list pts;
        bounds_list_init(&pts);
        object *sub = o->obj[i];
        if (strcmp("outline", sub->name) == 0) continue;
        sub->bounding_points(sub, &pts);
        if (!pts.head) {
            bounds_list_free(&pts);
            bounds_list_free(l);
            return 0;
        }
        bounds_list_join(l, &pts);
        bounds_list_free(&pts);
    }
    return 1;
}
static int bp_stubs(object *o, bounds_list *l) { (void)o; (void)l; return 1; }

/* ------------------------------------------------------------------ hcube faces (hcube.c:34-160)
 *
 * An N-cube gets one orthotope per m-face for every m = 2 .. N-1 (lower m first).  The m
 * spanning axes are enumerated as a decreasing index tuple counted upwards like an odometer;
 * for each tuple the 2^(N-m) positions of the remaining axes are counted in binary. */

static int n_choose(int n, int m)
{
    long num = 1, a = 1, b = 1;
    for (int i = 1; i <= n; ++i) num *= i;
    for (int i = 1; i <= m; ++i) a *= i;
    for (int i = 1; i <= n - m; ++i) b *= i;
    return (int)(num / (a * b));
}

static void hcube_add_faces(object *cube, int m)
{
    const int n = cube->dimensions;
    if (m > 2) hcube_add_faces(cube, m - 1);
    const int num_faces = (1 << (n - m)) * n_choose(n, m);
    vectNd pos, step;
    vectNd_calloc(&pos, n);
    vectNd_calloc(&step, n);
    int *axes = (int *)calloc((size_t)m, sizeof(int));          /* spanning axes, axes[0] > axes[1] > ... */
    int *side = (int *)calloc((size_t)(n - m > 0 ? n - m : 1), sizeof(int));
    for (int i = 0; i < m; ++i) axes[i] = m - i - 1;
    int corner_id = 0;
    for (int f = 0; f < num_faces; ++f) {
        /* face origin: centre, minus half an edge along spanning axes, +/- half along the others */
        int bits = corner_id;
        vectNd_reset(&pos);
        vectNd_copy(&pos, &cube->pos[0]);
        for (int i = 0; i < n; ++i) {
            int spans = 0;
            for (int j = 0; !spans && j < m; ++j) spans = (i == axes[j]);
            if (spans) {
                vectNd_scale(&cube->dir[i], -0.5 * cube->size[i], &step);
            } else {
                int bit = bits % 2;
                bits >>= 1;
                vectNd_scale(&cube->dir[i], cube->size[i] * (bit - 0.5), &step);
            }
            vectNd_add(&pos, &step, &pos);
        }
        object *face = object_alloc(n, "orthotope", "");
        object_add_flag(face, m);
        for (int i = 0; i < m; ++i) {
            vectNd_scale(&cube->dir[axes[i]], cube->size[axes[i]], &step);
            object_add_dir(face, &step);
        }
        object_add_pos(face, &pos);
        snprintf(face->name, sizeof(face->name), "%id face %i", m, f);
        object_add_obj(cube, face);

        /* next face: binary count over the sides, then the next axis tuple */
        ++corner_id;
        int i = 0;
        while (i < n - m && side[i] == 1) side[i++] = 0;
        if (i < n - m) {
            side[i] += 1;
        } else {
            int j = 0;
            while (j < m && axes[j] == n - j - 1) {
                axes[j] = (j < m - 1) ? axes[j + 1] + 1 : 0;
                ++j;
            }
            if (j < m) {
                axes[j] += 1;
                for (--j; j >= 0; --j) axes[j] = axes[j + 1] + 1;
            }
        }
    }
    free(axes);
    free(side);
    vectNd_free(&pos);
    vectNd_free(&step);
}

/* hcube.c:164-177: what the first ray through an hcube triggers */
void ndt_hcube_prepare(object *cube)
{
    if (cube->prepared) return;
    hcube_add_faces(cube, cube->dimensions - 1);
    cube->prepared = 1;
}

static int cleanup_hcube(object *cube)      /* hcube.c:179-190 */
{
    for (int i = 0; i < cube->n_obj; ++i) object_free(cube->obj[i]);
    free(cube->obj);
    cube->obj = NULL;
    cube->n_obj = cube->cap_obj = 0;
    cube->bounds.radius = 0.0;
    cube->prepared = 0;
    return 0;
}

/* ------------------------------------------------------------------ registry */

typedef struct {
    const char *name;
    int (*type_name)(char *, int);
    int (*params)(PARAMS_ARGS);
    int (*bounding_points)(object *, bounds_list *);
    int (*cleanup)(object *);
} type_entry;

/* Fixed order.  In the reference the order is whatever readdir() yields (object.c:141-153,
 * prepended: object.c:112-114) and scenes/random.c indexes it; this is the order the golden
 * fixtures were generated with (oracle/ref_shim.c pins the same one). */
static const type_entry type_table[] = {
    { "hcylinder", name_hcylinder, params_hcylinder, bp_hcylinder, NULL },
    { "orthotope", name_orthotope, params_orthotope, bp_orthotope, NULL },
    { "sphere", name_sphere, params_sphere, bp_sphere, NULL },
    { "hcube", name_hcube, params_hcube, bp_hcube, cleanup_hcube },
    { "hdisk", name_hdisk, params_hdisk, bp_hdisk, NULL },
    { "cluster", name_cluster, params_cluster, bp_cluster, NULL },
    { "hplane", name_hplane, params_hplane, bp_hplane, NULL },
    { "cylinder", name_cylinder, params_cylinder, bp_cylinder, NULL },
    { "stubs", name_stubs, params_stubs, bp_stubs, NULL },
    { "hfacet", name_hfacet, params_hfacet, bp_vertices, NULL },
    { "facet", name_facet, params_facet, bp_vertices, NULL },
};
static const int n_types = (int)(sizeof(type_table) / sizeof(type_table[0]));

/* Object plugins (object.c:51-153).  The nine types the device intersects, `cluster` and `stubs` are built in, in the
 * table above.  register_objects(dir) additionally dlopens every .so of a directory the way the reference does and asks it
 * for its type_name (object.c:60-111 names the same five entry points):
 *   - a plugin whose type is built in (the reference's own objects/sphere.so ...) is acknowledged and closed again: the
 *     built-in host functions and the device intersector stand for it;
 *   - a plugin with a type of its own is kept: object_alloc() of that type works, with the plugin's params and
 *     bounding_points, so a scene program that uses it builds its scene; rendering refuses it by name (ndt_flatten.c:
 *     a foreign intersect() cannot run on the device -- DESIGN.md, section 7). */
typedef struct plugin_type {
    struct plugin_type *next;
    char name[OBJ_TYPE_MAX_LEN];
    char file[256];
    void *dl;
    int (*type_name)(char *, int);
    int (*params)(PARAMS_ARGS);
    int (*bounding_points)(object *, bounds_list *);
    int (*cleanup)(object *);
} plugin_type;
static plugin_type *plugins = NULL;
static int n_plugins = 0;

int register_object(char *filename)
{
    void *dl = dlopen(filename, RTLD_NOW);
    if (!dl) {
        fprintf(stderr, "%s\n", dlerror());
        return -1;
    }
    plugin_type *e = (plugin_type *)calloc(1, sizeof(*e));
    *(void **)(&e->type_name) = dlsym(dl, "type_name");
    *(void **)(&e->params) = dlsym(dl, "params");
    *(void **)(&e->bounding_points) = dlsym(dl, "bounding_points");
    *(void **)(&e->cleanup) = dlsym(dl, "cleanup");
    void *isect = dlsym(dl, "intersect");
    const char *missing = !e->type_name ? "type_name" : !e->params ? "params" : !e->bounding_points ? "bounding_points" : !isect ? "intersect" : NULL;
    if (missing) {                          /* object.c:86-101 */
        fprintf(stderr, "%s missing %s function.\n", filename, missing);
        dlclose(dl);
        free(e);
        return -1;
    }
    e->type_name(e->name, sizeof(e->name));
    const char *base = strrchr(filename, '/');
    base = base ? base + 1 : filename;
    for (int i = 0; i < n_types; ++i)
        if (!strcasecmp(type_table[i].name, e->name)) {
            printf("\tobject type '%s' from '%s': built in (device intersector).\n", e->name, base);
            dlclose(dl);
            free(e);
            return 0;
        }
    for (plugin_type *q = plugins; q; q = q->next)
        if (!strcasecmp(q->name, e->name)) {        /* loaded before */
            dlclose(dl);
            free(e);
            return 0;
        }
    snprintf(e->file, sizeof(e->file), "%s", base);
    e->dl = dl;
    e->next = plugins;
    plugins = e;
    ++n_plugins;
    printf("\tloaded object type '%s' from '%s': host side only (no device intersector: scenes that use it cannot be rendered).\n", e->name, base);
    return 0;
}

int register_objects(char *dirname)
{
    if (!dirname) {
        fprintf(stderr, "%s: dirname is NULL\n", __func__);
        return -1;
    }
    DIR *d = opendir(dirname);
    if (!d) return 0;           /* no plugin directory: the built-in types are all there is (the reference would stop here) */
    struct dirent *dp;
    while ((dp = readdir(d)) != NULL) {
        const size_t len = strlen(dp->d_name);
        if (len > 3 && !strncasecmp(dp->d_name + len - 3, ".so", 3)) {
            char path[PATH_MAX];
            snprintf(path, sizeof(path), "%s/%s", dirname, dp->d_name);     /* avoid the library search path (object.c:146) */
            register_object(path);
        }
    }
    closedir(d);
    return 0;
}

int unregister_objects(void)
{
    while (plugins) {
        plugin_type *e = plugins;
        plugins = e->next;
        dlclose(e->dl);
        free(e);
    }
    n_plugins = 0;
    return 0;
}

/* the file a host-side-only type came from (for messages), or NULL for a built-in type */
const char *ndt_object_plugin_file(const char *type)
{
    for (plugin_type *q = plugins; q; q = q->next)
        if (!strcasecmp(q->name, type)) return q->file;
    return NULL;
}

int registered_types(char ***list, int *num)
{
    *num = n_types + n_plugins;
    *list = (char **)calloc((size_t)*num + 1, sizeof(char *));
    if (!*list) { *num = -1; return -1; }
    for (int i = 0; i < n_types; ++i) (*list)[i] = strdup(type_table[i].name);
    int k = n_types;            /* the built-in order first (scenes/random.c indexes the list), plugins behind it */
    for (plugin_type *q = plugins; q; q = q->next) (*list)[k++] = strdup(q->name);
    return *num;
}

int registered_types_free(char **list)
{
    if (!list) return 0;
    for (int i = 0; list[i]; ++i) free(list[i]);
    free(list);
    return 0;
}

int ndt_object_type_id(object *o)
{
    char tn[OBJ_TYPE_MAX_LEN] = "";
    o->type_name(tn, sizeof(tn));
    static const char *device_types[] = { "sphere", "hplane", "hdisk", "cylinder", "hcylinder", "orthotope", "hcube",
                                          "hfacet", "facet" };
    for (int i = 0; i < 9; ++i)
        if (!strcmp(tn, device_types[i])) return i;
    if (!strcmp(tn, "cluster")) return NDT_TYPE_CLUSTER;
    return NDT_TYPE_OTHER;
}

/* ------------------------------------------------------------------ object_* (object.c:226-603) */

object *object_alloc(int dimensions, char *type, char *name)
{
    const type_entry *t = NULL;
    type_entry from_plugin;
    for (int i = 0; i < n_types && !t; ++i)
        if (!strcasecmp(type_table[i].name, type)) t = &type_table[i];
    for (plugin_type *q = plugins; q && !t; q = q->next)
        if (!strcasecmp(q->name, type)) {
            from_plugin.name = q->name;
            from_plugin.type_name = q->type_name;
            from_plugin.params = q->params;
            from_plugin.bounding_points = q->bounding_points;
            from_plugin.cleanup = q->cleanup;
            t = &from_plugin;
        }
    if (!t) {
        fprintf(stderr, "Unknown object type '%s'.\n", type);
        exit(1);                                /* the reference's behaviour (object.c:233-236); scenes rely on it */
    }
    object *o = (object *)calloc(1, sizeof(object));
    o->dimensions = dimensions;
    o->type_name = t->type_name;
    o->params = t->params;
    o->cleanup = t->cleanup;
    o->bounding_points = t->bounding_points;
    o->intersect = host_intersect_stub;
    o->get_color = default_color;
    o->get_reflect = default_reflect;
    o->get_trans = default_trans;
    o->refract_ray = NULL;
    vectNd_calloc(&o->bounds.center, dimensions);
    o->bounds.radius = 0;
    if (!name) name = "unnamed";
    strncpy(o->name, name, sizeof(o->name));
    o->name[sizeof(o->name) - 1] = '\0';
    return o;
}

int object_free(object *o)
{
    if (o->cleanup) o->cleanup(o);
    vectNd_free(&o->bounds.center);
    o->bounds.radius = 0;
    if (o->pos) {
        for (int i = 0; i < o->n_pos; ++i) vectNd_free(&o->pos[i]);
        free(o->pos);
    }
    if (o->dir) {
        for (int i = 0; i < o->n_dir; ++i) vectNd_free(&o->dir[i]);
        free(o->dir);
    }
    free(o->size);
    free(o->flag);
    if (o->obj) {
        for (int i = 0; i < o->n_obj; ++i) object_free(o->obj[i]);
        free(o->obj);
    }
    free(o->prepped);
    free(o);
    return 0;
}

int object_cleanup_all(object *o)
{
    for (int i = 0; i < o->n_obj; ++i) object_cleanup_all(o->obj[i]);
    if (o->cleanup && o->prepared) o->cleanup(o);
    o->prepared = 0;
    vectNd_reset(&o->bounds.center);
    o->bounds.radius = 0;
    return 0;
}

int object_validate(object *o)
{
    char type[256];
    if (!o->type_name || !o->params || !o->bounding_points || !o->intersect || !o->get_color || !o->get_reflect ||
        !o->get_trans) {
        fprintf(stderr, "object %p is missing a type function.\n", (void *)o);
        return -1;
    }
    o->type_name(type, sizeof(type));
    int n_pos, n_dir, n_size, n_flag, n_obj;
    o->params(o, &n_pos, &n_dir, &n_size, &n_flag, &n_obj);
    const struct { const char *what; int have, need; } checks[] = {
        { "positions", o->n_pos, n_pos }, { "directions", o->n_dir, n_dir }, { "sizes", o->n_size, n_size },
        { "flags", o->n_flag, n_flag }, { "objects", o->n_obj, n_obj },
    };
    for (int i = 0; i < 5; ++i) {
        if (checks[i].need > checks[i].have) {
            fprintf(stderr, "insufficient %s set for %s object '%s' %p (%i set, %i required).\n", checks[i].what, type,
                    o->name, (void *)o, checks[i].have, checks[i].need);
            exit(1);                            /* object.c:381-399 */
        }
    }
    for (int i = 0; i < o->n_obj; ++i) object_validate(o->obj[i]);
    return 0;
}

/* deep-copies the vector (object.c:427-454); capacity grows 2c+1 */
static int append_vector(vectNd **list, int *n, int *cap, vectNd *vec)
{
    if (*n >= *cap) {
        int new_cap = *cap * 2 + 1;
        vectNd *grown = NULL;
        if (posix_memalign((void **)&grown, 16, (size_t)new_cap * sizeof(vectNd))) { perror("posix_memalign"); return -1; }
        for (int i = 0; i < *n; ++i) {
            vectNd_alloc(&grown[i], (*list)[i].n);
            vectNd_copy(&grown[i], &(*list)[i]);
            vectNd_free(&(*list)[i]);
        }
        free(*list);
        *list = grown;
        *cap = new_cap;
    }
    vectNd_alloc(&(*list)[*n], vec->n);
    vectNd_copy(&(*list)[*n], vec);
    *n += 1;
    return 0;
}

static int grow(void **list, int *n, int *cap, size_t size)
{
    if (*n >= *cap) {
        int new_cap = *cap * 2 + 1;
        void *p = realloc(*list, (size_t)new_cap * size);
        if (!p) { perror("realloc"); return -1; }
        *list = p;
        *cap = new_cap;
    }
    return 0;
}

int object_add_pos(object *o, vectNd *v) { return append_vector(&o->pos, &o->n_pos, &o->cap_pos, v); }
int object_add_dir(object *o, vectNd *v) { return append_vector(&o->dir, &o->n_dir, &o->cap_dir, v); }
int object_add_posStr(object *o, char *str)
{
    vectNd v;
    vectNd_calloc(&v, o->dimensions);
    vectNd_setStr(&v, str);
    int r = object_add_pos(o, &v);
    vectNd_free(&v);
    return r;
}
int object_add_dirStr(object *o, char *str)
{
    vectNd v;
    vectNd_calloc(&v, o->dimensions);
    vectNd_setStr(&v, str);
    int r = object_add_dir(o, &v);
    vectNd_free(&v);
    return r;
}
int object_add_size(object *o, double s)
{
    if (grow((void **)&o->size, &o->n_size, &o->cap_size, sizeof(double)) < 0) return -1;
    o->size[o->n_size++] = s;
    return 0;
}
int object_add_flag(object *o, int f)
{
    if (grow((void **)&o->flag, &o->n_flag, &o->cap_flag, sizeof(int)) < 0) return -1;
    o->flag[o->n_flag++] = f;
    return 0;
}
/* takes ownership of the child and invalidates the parent's bounds (object.c:495-504) */
int object_add_obj(object *o, object *child)
{
    if (grow((void **)&o->obj, &o->n_obj, &o->cap_obj, sizeof(object *)) < 0) return -1;
    o->obj[o->n_obj++] = child;
    o->bounds.radius = 0.0;
    return 0;
}

int object_move(object *o, vectNd *offset)
{
    object_validate(o);
    for (int i = 0; i < o->n_pos; ++i) vectNd_add(&o->pos[i], offset, &o->pos[i]);
    vectNd_add(&o->bounds.center, offset, &o->bounds.center);
    for (int i = 0; i < o->n_obj; ++i) object_move(o->obj[i], offset);
    return 0;
}

int object_rotate(object *o, vectNd *center, int v1, int v2, double angle)
{
    object_validate(o);
    for (int i = 0; i < o->n_pos; ++i) vectNd_rotate(&o->pos[i], center, v1, v2, angle, &o->pos[i]);
    vectNd_rotate(&o->bounds.center, center, v1, v2, angle, &o->bounds.center);
    for (int i = 0; i < o->n_dir; ++i) vectNd_rotate(&o->dir[i], NULL, v1, v2, angle, &o->dir[i]);
    for (int i = 0; i < o->n_obj; ++i) object_rotate(o->obj[i], center, v1, v2, angle);
    return 0;
}

int object_rotate2(object *o, vectNd *center, vectNd *v1, vectNd *v2, double angle)
{
    object_validate(o);
    for (int i = 0; i < o->n_pos; ++i) vectNd_rotate2(&o->pos[i], center, v1, v2, angle, &o->pos[i]);
    vectNd_rotate2(&o->bounds.center, center, v1, v2, angle, &o->bounds.center);
    for (int i = 0; i < o->n_dir; ++i) vectNd_rotate2(&o->dir[i], NULL, v1, v2, angle, &o->dir[i]);
    for (int i = 0; i < o->n_obj; ++i) object_rotate2(o->obj[i], center, v1, v2, angle);
    return 0;
}

/* object.c:582-603: fit the bounding sphere; an empty point list marks an infinite object */
int object_get_bounds(object *o)
{
    bounds_list pts;
    bounds_list_init(&pts);
    o->bounding_points(o, &pts);
    if (!pts.head) {
        o->bounds.radius = -1.0;
        return 0;
    }
    bounds_list_optimal(&pts, &o->bounds.center, &o->bounds.radius);
    if (o->bounds.radius > 0.0) o->bounds.radius += EPSILON;
    bounds_list_free(&pts);
    return 0;
}
