/*
 * ndt_host_api.h -- the host-side scene API of ndt, as this repository provides it.
 *
 * Scene programs (reference scenes/ *.c) are written against ndt's public C API: they
 * #include "../scene.h", write struct fields directly (obj->red, lgt->type, scn->bg_red,
 * scn->cam ...) and call scene_*, object_*, camera_*, vectNd_*, bounds_list_*, nm_*.  The
 * struct layouts are therefore ABI.  This header declares that API -- same type
 * names, same field order, same function names and argument meaning -- implemented from
 * scratch in ndt_amd/host/src.  The files vectNd.h / object.h / scene.h / camera.h /
 * bounding.h / nelder-mead.h next to it only forward here, so a scene source compiles
 * unchanged, and a scene .so built against the reference's own headers loads unchanged.
 *
 * What differs from the reference by design:
 *   - object types are built in (no dlopen of objects/ *.so): the nine device-capable types
 *     plus `cluster` (flattened before rendering) and `stubs`; registry order is fixed;
 *   - obj->intersect is a host stub: rays are traced by libndt_hip.so on the GPU;
 *   - render_image() flattens the scene and calls the C ABI of include/ndt_hip.h.
 */
#ifndef NDT_HOST_API_H
#define NDT_HOST_API_H

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ constants */

#ifndef EPSILON
#define EPSILON (1e-4)                      /* reference object.h:15 */
#endif
#ifndef EPSILON2
#define EPSILON2 ((EPSILON) * (EPSILON))
#endif
#define VECTND_SUCCESS 1
#define VECTND_FAIL 0
#define VECTND_DEF_SIZE 4                   /* inline storage, reference vectNd.h:19 */
#define OBJ_TYPE_MAX_LEN 64
#define OBJ_NAME_MAX_LEN 32
#define LIGHT_NAME_MAX_LEN 32
#define SCENE_NAME_MAX_LEN 64
#define EYE_OFFSET 0.125
#define rad2deg(x) ((x) * 180.0 / M_PI)
#define deg2rad(x) ((x) * M_PI / 180.0)

/* ------------------------------------------------------------------ vectNd (ABI: vectNd.h:42-51) */

typedef struct vectNd_t {
    double space[VECTND_DEF_SIZE];          /* storage for n <= 4 */
    double *v;                              /* = space, or a 16-byte aligned heap block, even-padded */
    int n;
} __attribute__((__aligned__(16))) vectNd;

/* Storage rules match the reference so that code compiled against either header can share
 * vectors: heap blocks are 16-byte aligned and padded to an even length, and the pad lane of
 * an odd-length vector is zero (the reference's SSE2 loops read and write it). */
static inline int vectNd_alloc(vectNd *v, int dim)
{
    v->n = dim;
    if (dim > VECTND_DEF_SIZE) {
        void *p = NULL;
        int padded = dim + (dim & 1);
        if (posix_memalign(&p, 16, (size_t)padded * sizeof(double))) {
            v->v = NULL;
            return VECTND_FAIL;
        }
        v->v = (double *)p;
    } else {
        v->v = v->space;
    }
    if (dim & 1) v->v[dim] = 0.0;
    return VECTND_SUCCESS;
}
static inline int vectNd_fill(vectNd *v, double val)
{
    int padded = v->n + (v->n & 1);         /* the reference fills the pad lane too (vectNd.h:85-87) */
    for (int i = 0; i < padded; ++i) v->v[i] = val;
    return VECTND_SUCCESS;
}
static inline int vectNd_calloc(vectNd *v, int dim)
{
    vectNd_alloc(v, dim);
    vectNd_fill(v, 0.0);
    return VECTND_SUCCESS;
}
static inline int vectNd_free(vectNd *v)
{
    if (v->n > VECTND_DEF_SIZE) free(v->v);
    v->v = NULL;
    v->n = -1;
    return VECTND_SUCCESS;
}
static inline int vectNd_reset(vectNd *v)
{
    memset(v->v, 0, (size_t)v->n * sizeof(double));
    return VECTND_SUCCESS;
}
static inline int vectNd_get(vectNd *v, int pos, double *val)
{
    if (pos < 0 || pos >= v->n) return VECTND_FAIL;
    *val = v->v[pos];
    return VECTND_SUCCESS;
}
static inline int vectNd_set(vectNd *v, int pos, double val)
{
    if (pos < 0 || pos >= v->n) return VECTND_FAIL;
    v->v[pos] = val;
    return VECTND_SUCCESS;
}
static inline int vectNd_setStr(vectNd *v, char *str)
{
    /* comma separated components, extra ones ignored (vectNd.h:110-122) */
    char *copy = strdup(str), *save = NULL;
    int pos = 0;
    for (char *tok = strtok_r(copy, ",", &save); tok; tok = strtok_r(NULL, ",", &save))
        vectNd_set(v, pos++, atof(tok));
    free(copy);
    return VECTND_SUCCESS;
}
static inline void vectNd_copy(vectNd *dst, vectNd *src) { memcpy(dst->v, src->v, (size_t)src->n * sizeof(double)); }
static inline void vectNd_min(vectNd *v, double *res)
{
    double m = v->v[0];
    for (int i = 1; i < v->n; ++i) m = (v->v[i] < m) ? v->v[i] : m;
    *res = m;
}
static inline void vectNd_max(vectNd *v, double *res)
{
    double m = v->v[0];
    for (int i = 1; i < v->n; ++i) m = (v->v[i] > m) ? v->v[i] : m;
    *res = m;
}
static inline void vectNd_mul(vectNd *a, vectNd *b, vectNd *res)
{
    for (int i = 0; i < a->n; ++i) res->v[i] = a->v[i] * b->v[i];
}
/* Dot product in the reference's SSE2 order (vectNd.h:215-227): even components accumulate in
 * one lane, odd components in the other, lanes added last.  Part of the numerical contract. */
static inline void vectNd_dot(vectNd *a, vectNd *b, double *res)
{
    double lane0 = a->v[0] * b->v[0];
    double lane1 = (a->n > 1) ? a->v[1] * b->v[1] : 0.0;
    for (int i = 2; i < a->n; i += 2) {
        lane0 = lane0 + a->v[i] * b->v[i];
        if (i + 1 < a->n) lane1 = lane1 + a->v[i + 1] * b->v[i + 1];
    }
    *res = lane0 + lane1;
}
static inline void vectNd_add(vectNd *a, vectNd *b, vectNd *res)
{
    for (int i = 0; i < a->n; ++i) res->v[i] = a->v[i] + b->v[i];
}
static inline void vectNd_sub(vectNd *a, vectNd *b, vectNd *res)
{
    for (int i = 0; i < a->n; ++i) res->v[i] = a->v[i] - b->v[i];
}
static inline void vectNd_scale(vectNd *a, double s, vectNd *res)
{
    for (int i = 0; i < a->n; ++i) res->v[i] = a->v[i] * s;
}
static inline void vectNd_l2norm(vectNd *v, double *res)
{
    double d;
    vectNd_dot(v, v, &d);
    *res = sqrt(d);
}
#define vectNd_length vectNd_l2norm
static inline void vectNd_unitize(vectNd *v)
{
    double len;
    vectNd_l2norm(v, &len);
    if (len > EPSILON || len < -EPSILON) vectNd_scale(v, 1.0 / len, v);     /* untouched when tiny */
}
static inline void vectNd_dist(vectNd *a, vectNd *b, double *res)
{
    vectNd d;
    vectNd_alloc(&d, a->n);
    vectNd_sub(a, b, &d);
    vectNd_l2norm(&d, res);
    vectNd_free(&d);
}
static inline void vectNd_proj_unit(vectNd *v, vectNd *onto, vectNd *res)
{
    double ab;
    vectNd_dot(v, onto, &ab);
    vectNd_scale(onto, ab, res);
}
static inline void vectNd_proj(vectNd *v, vectNd *onto, vectNd *res)
{
    double ab, bb;
    vectNd_dot(onto, onto, &bb);
    vectNd_dot(v, onto, &ab);
    vectNd_scale(onto, ab / bb, res);
}

int vectNd_cross(vectNd *vects, vectNd *res);
int vectNd_orthogonalize(vectNd *in1, vectNd *in2, vectNd *out1, vectNd *out2);
int vectNd_angle(vectNd *v1, vectNd *v2, double *angle);
int vectNd_angle3(vectNd *p1, vectNd *p2, vectNd *p3, double *angle);
int vectNd_reflect(vectNd *v, vectNd *normal, vectNd *res, double mag);
int vectNd_refract(vectNd *v, vectNd *normal, vectNd *res, double index);
int vectNd_interpolate(vectNd *s, vectNd *e, double x, vectNd *r);
int vectNd_rotate(vectNd *v, vectNd *center, int i, int j, double angle, vectNd *res);
int vectNd_rotate2(vectNd *v, vectNd *center, vectNd *v1, vectNd *v2, double angle, vectNd *res);
int vectNd_print(vectNd *v, char *name);

/* ------------------------------------------------------------------ bounding spheres (ABI: bounding.h:12-32) */

typedef struct bounding_sphere_t {
    vectNd center;
    double radius;
    unsigned int prepared : 1;
    double radius_sqr;
} bounding_sphere;

typedef struct bounds_node_t {
    bounding_sphere bounds;
    struct bounds_node_t *next;
} bounds_node;

typedef struct bounds_list_t {
    struct bounds_node_t *head;
    struct bounds_node_t *tail;
} bounds_list;

int bounds_list_init(bounds_list *list);
int bounds_list_add(bounds_list *list, vectNd *vect, double radius);
int bounds_list_join(bounds_list *list, bounds_list *other);
int bounds_list_free(bounds_list *list);
int bounds_list_centroid(bounds_list *list, vectNd *centroid);
int bounds_list_radius(bounds_list *list, vectNd *centroid, double *radius);
int bounds_list_optimal(bounds_list *list, vectNd *centroid, double *radius);
int vect_bounding_sphere_intersect(bounding_sphere *sph, vectNd *o, vectNd *v, double min_dist);

/* ------------------------------------------------------------------ Nelder-Mead (nelder-mead.h) */

void nm_init(void **nm, int dimensions);
void nm_free(void *nm);
void nm_set_seed(void *nm, vectNd *seed);
void nm_best_point(void *nm, vectNd *result);
void nm_add_result(void *nm, vectNd *parameters, double value);
void nm_next_point(void *nm, vectNd *vector);
int nm_simplex_point(void *nm_ptr, int which, vectNd *point, double *value);
int nm_done(void *nm, double threshold, int iterations);

/* ------------------------------------------------------------------ objects (ABI: object.h:23-74) */

typedef struct gen_object {
    unsigned int transparent : 1;
    unsigned int prepared : 1;
    int dimensions;
    double red, green, blue;                /* surface colour */
    double red_r, green_r, blue_r;          /* reflectivity */
    double refract_index;
    char name[OBJ_NAME_MAX_LEN];
    vectNd *pos;    int n_pos, cap_pos;     /* positions */
    vectNd *dir;    int n_dir, cap_dir;     /* directions */
    double *size;   int n_size, cap_size;   /* scalars */
    int *flag;      int n_flag, cap_flag;   /* integers */
    struct gen_object **obj; int n_obj, cap_obj;    /* owned sub-objects */
    bounding_sphere bounds;
    void *prepped;                          /* type-private derived data */
    void *dl_handle;                        /* unused here: types are built in */
    int (*type_name)(char *name, int size);
    int (*params)(struct gen_object *obj, int *n_pos, int *n_dir, int *n_size, int *n_flags, int *n_obj);
    int (*cleanup)(struct gen_object *obj);
    int (*bounding_points)(struct gen_object *obj, bounds_list *list);
    int (*intersect)(struct gen_object *obj, vectNd *o, vectNd *v, vectNd *res, vectNd *normal, struct gen_object **obj_ptr);
    int (*get_color)(struct gen_object *obj, vectNd *at, double *red, double *green, double *blue);
    int (*get_reflect)(struct gen_object *obj, vectNd *at, double *red_r, double *green_r, double *blue_r);
    int (*get_trans)(struct gen_object *obj, vectNd *at, int *transparent);
    int (*refract_ray)(struct gen_object *obj, vectNd *at, double *index);
} object;

int register_objects(char *dirname);        /* registers the built-in types; dirname is ignored */
int registered_types(char ***list, int *num);
int registered_types_free(char **list);
int unregister_objects(void);

object *object_alloc(int dimensions, char *type, char *name);
int object_free(object *obj);
int object_cleanup_all(object *obj);
int object_validate(object *obj);
int object_add_pos(object *obj, vectNd *new_pos);
int object_add_posStr(object *obj, char *str);
int object_add_dir(object *obj, vectNd *new_dir);
int object_add_dirStr(object *obj, char *str);
int object_add_size(object *obj, double new_size);
int object_add_flag(object *obj, int new_flag);
int object_add_obj(object *obj, object *new_obj);
int object_move(object *obj, vectNd *offset);
int object_rotate(object *obj, vectNd *center, int v1, int v2, double angle);
int object_rotate2(object *obj, vectNd *center, vectNd *v1, vectNd *v2, double angle);
int object_get_bounds(object *obj);

/* ------------------------------------------------------------------ camera (ABI: camera.h:32-75) */

typedef enum CAMERA_TYPE_ENUM { CAMERA_NORMAL, CAMERA_VR, CAMERA_PANO } camera_type_t;
extern const char *CAMERA_TYPE_STRING[];

typedef struct camera_t {
    camera_type_t type;
    vectNd viewPoint, viewTarget, up;       /* aiming inputs */
    double rotation;
    double eye_offset;
    double aperture_radius, focal_distance; /* depth of field inputs */
    double zoom;
    unsigned int flip_x : 1;
    unsigned int flip_y : 1;
    unsigned int flatten : 1;
    double hFov, vFov;
    unsigned int prepared : 1;              /* everything below is set by camera_aim */
    double leveling;
    vectNd pos, leftEye, rightEye;
    vectNd dirX, dirY, imgOrig;
    vectNd localX, localY, localZ;
} camera;

int camera_alloc(camera *cam, int dim);
int camera_free(camera *cam);
int camera_init(camera *cam);
int camera_reset(camera *cam);
int camera_set_aim(camera *cam, vectNd *pos, vectNd *target, vectNd *up, double rot);
int camera_set_zoom(camera *cam, double zoom);
int camera_set_flip(camera *cam, int x, int y);
int camera_aim_naive(camera *cam);
int camera_aim(camera *cam);
int camera_focus(camera *cam, vectNd *point);
void camera_target_point(camera *cam, double x, double y, double dist, vectNd *p);
void camera_print(camera *cam);
void camera_flip_x(camera *cam);
void camera_flip_y(camera *cam);
void camera_zoom(camera *cam);

/* ------------------------------------------------------------------ scene (ABI: scene.h:23-62) */

typedef enum LIGHT_TYPE_ENUM {
    LIGHT_AMBIENT, LIGHT_POINT, LIGHT_DIRECTIONAL, LIGHT_SPOT, LIGHT_DISK, LIGHT_RECT
} light_type;
extern const char *LIGHT_TYPE_STRING[];

typedef struct light_t {
    vectNd pos, target, dir;
    vectNd u, v;                            /* area-light basis */
    double radius;
    light_type type;
    double red, green, blue;
    double angle;                           /* spot cone, degrees */
    vectNd u1, v1;
    unsigned int prepared : 1;
    char name[LIGHT_NAME_MAX_LEN];
} light;

typedef struct scene_t {
    int dimensions;
    camera cam;
    int num_objects, num_lights;
    object **object_ptrs;
    light **lights;
    light ambient;                          /* only its colour is used (ndt.c:89-91) */
    double bg_red, bg_green, bg_blue, bg_alpha;
    char name[SCENE_NAME_MAX_LEN];
} scene;

int scene_init(scene *scn, char *name, int dim);
int scene_free(scene *scn);
int scene_add_object(scene *scn, object *obj);
int scene_alloc_object(scene *scn, int dimensions, object **obj, char *type);
int scene_remove_object(scene *scn, object *obj);
int scene_alloc_light(scene *scn, light **lgt);
int scene_free_light(light *lgt);
int scene_aim_light(light *lgt, vectNd *target);
int scene_prepare_light(light *lgt);
int scene_validate_objects(scene *scn);
int scene_print(scene *scn);
/* YAML scene files (scene.h:80-86; scene.c:573-2177): one `---` document per animation frame */
int scene_write_yaml(scene *scn, char *fname);
int scene_read_yaml(scene *scn, char *fname, int frame);
int scene_yaml_count_frames(char *fname);

/* ------------------------------------------------------------------ the accelerated entry */

/* render_image (reference ndt.c:900), mono / samples=1 form: flattens `scn` (objects, the
 * bounding spheres and kd-tree this library builds exactly like ndt.c:1899-1908 does, the aimed
 * camera), uploads it and renders on the GPU through include/ndt_hip.h.  `rgba` receives
 * width*height*4 doubles laid out like the reference's dbl image (image.c:126).
 * Returns 1 like the reference on success, 0 with a message on stderr otherwise. */
int ndt_render_image(scene *scn, int width, int height, int threads, int max_optic_depth, double *rgba);
/* the same with Whitted's recursive anti-aliasing, the reference's `-a diff,depth` (ndt.c:1453-1465); aa_depth < 0 = off */
int ndt_render_image_aa(scene *scn, int width, int height, int threads, int aa_diff, int aa_depth, int max_optic_depth,
                        double *rgba);
/* ... and with everything render_image takes (ndt.c:900): samples (-n), stereo mode (-m), specular_enabled (-p clears it),
 * depth map (-z) */
int ndt_render_image_full(scene *scn, int width, int height, int samples, int threads, int aa_diff, int aa_depth, int stereo,
                          int specular, int max_optic_depth, double *rgba, double *depth);

/* The same frame as the bytes the reference stores (pixel_d2c on every channel, image.h:36-39; quantised on the GPU):
 * `rgba8` receives width*height*4 bytes.  What the driver writes PPM / PNG files from. */
int ndt_render_image_rgba8(scene *scn, int width, int height, int samples, int threads, int aa_diff, int aa_depth, int stereo,
                           int specular, int max_optic_depth, unsigned char *rgba8);

/* Which GPUs the calling thread's frames are rendered on (the settings are per thread, like the GPU contexts):
 *   ndt_render_use_device(d)     one context on device d -- `ndt_hip -j K` gives worker w device w mod device count,
 *                                the reference's MPI_MODE_FRAME (one frame per rank, ndt.c:1770-1830);
 *   ndt_render_use_devices(n)    ONE frame over n contexts, context k on device k mod device count, rows dealt
 *                                cyclically -- the reference's MPI_MODE_ROW (ndt.c:812-820), `ndt_hip -g n`.
 * Both drop the thread's contexts if the choice changes.  Default: one context on device 0. */
void ndt_render_use_device(int device);
void ndt_render_use_devices(int n_contexts);

#ifdef __cplusplus
}
#endif
#endif /* NDT_HOST_API_H */
