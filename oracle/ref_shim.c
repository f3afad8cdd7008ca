/*
 * ref_shim.c -- TEST INFRASTRUCTURE.  Driver around the *compiled reference* (oracle/_ref).
 *
 * This file is original code of this repository.  It is compiled against the reference's
 * public headers where they lie (-I/root/reference) and linked with the reference's own
 * object files (built by oracle/Makefile into oracle/_ref/, never committed).  It does what
 * the reference's main() does for one frame (ndt.c:1758-1936: register object plugins,
 * scene_setup, kd-tree build, camera_aim, render_image) and, around that,
 *   1. flattens the prepared scene graph into the `ndtscene` text format (hex floats) that
 *      tests/golden/ holds -- this is the same walk INTEGRATION.md proposes as the
 *      reference-side binding for ndt_hip_upload_scene;
 *   2. dumps the double framebuffer render_image produced (via its img_copy argument);
 *   3. answers batches of trace_kd queries (object.c:683) for known-answer tests;
 *   4. counts trace_kd calls (linked with -Wl,--wrap=trace_kd) and times render_image;
 *   5. --aa diff,depth: runs Whitted's recursive anti-aliasing with the reference's own
 *      render_line / resample_pixel (ndt.c:735 / 709) and dumps the resampled colours as doubles
 *      (render_image itself only keeps the 8-bit copy of them, ndt.c:1124).
 * Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may run it.
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <sys/stat.h>
#include <unistd.h>

#include "vectNd.h"
#include "image.h"
#include "object.h"
#include "scene.h"
#include "kd-tree.h"

/* symbols the reference's ndt.c defines (ndt.c:41,68,900) */
extern kd_tree_t kdtree;
extern int specular_enabled;
int render_image(scene *scn, char *name, char *depth_name, int width, int height, int samples,
                 int mode, int threads, int aa_diff, int aa_depth, int max_optic_depth,
                 image_t *img_copy, image_t *depth_copy);

/* the anti-aliasing pass's own pieces (ndt.c:44, 709, 735); stereo_mode is an enum, MONO == 0 */
extern int recursive_aa;
int render_line(scene *scn, int width, double x_scale, int height, double y_scale, int j, int mode, int samples,
                image_t *img, image_t *depth_map, int max_optic_depth);
int resample_pixel(scene *scn, int width, double x_scale, int height, double y_scale, int i, int j, int mode,
                   int samples, int aa_diff, int aa_depth, image_t *img, dbl_pixel_t *clr, int max_optic_depth);

/* ---- trace_kd call counter (ld --wrap) ---- */
static long long n_trace_closest = 0;   /* dist_limit < 0 : primary + secondary */
static long long n_trace_shadow = 0;    /* dist_limit >= 0 */
static int counting = 0;
int __real_trace_kd(vectNd *pos, vectNd *look, kd_tree_t *kd, vectNd *hit, vectNd *hit_normal,
                    object **ptr, double dist_limit);
int __wrap_trace_kd(vectNd *pos, vectNd *look, kd_tree_t *kd, vectNd *hit, vectNd *hit_normal,
                    object **ptr, double dist_limit)
{
    if (counting) {
        if (dist_limit < 0)
            __atomic_fetch_add(&n_trace_closest, 1, __ATOMIC_RELAXED);
        else
            __atomic_fetch_add(&n_trace_shadow, 1, __ATOMIC_RELAXED);
    }
    return __real_trace_kd(pos, look, kd, hit, hit_normal, ptr, dist_limit);
}

static double now_s(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + 1e-6 * tv.tv_usec;
}

/* ---- flat object table ---- */
typedef struct {
    object *obj;
    int parent;
} flat_obj;

static flat_obj *fobjs = NULL;
static int n_fobjs = 0, cap_fobjs = 0, n_items = 0;

static int fobj_add(object *o, int parent)
{
    if (n_fobjs >= cap_fobjs) {
        cap_fobjs = cap_fobjs * 2 + 64;
        fobjs = realloc(fobjs, cap_fobjs * sizeof(*fobjs));
    }
    fobjs[n_fobjs].obj = o;
    fobjs[n_fobjs].parent = parent;
    return n_fobjs++;
}

static int fobj_find(object *o)
{
    for (int i = 0; i < n_fobjs; ++i)
        if (fobjs[i].obj == o)
            return i;
    return -1;
}

static void type_of(object *o, char *buf, int len)
{
    memset(buf, 0, len);
    o->type_name(buf, len);
}

static void put_vec(FILE *f, const char *key, vectNd *v, int dims)
{
    fprintf(f, "%s", key);
    for (int i = 0; i < dims; ++i)
        fprintf(f, " %a", (v && v->v && v->n == dims) ? v->v[i] : 0.0);
    fprintf(f, "\n");
}

/* Make every lazily derived piece of state exist, the way the first ray through each object
 * would (prepare() in each plugin; lazy bounds at object.c:609-615). */
static void force_prepare(object *o, int dims)
{
    vectNd ro, rv, res, nrm;
    object *ptr = NULL;
    vectNd_calloc(&ro, dims);
    vectNd_calloc(&rv, dims);
    vectNd_calloc(&res, dims);
    vectNd_calloc(&nrm, dims);
    /* a ray far away from everything, pointing along +x0 */
    for (int i = 0; i < dims; ++i)
        ro.v[i] = -1.0e3 - 17.0 * i;
    rv.v[0] = 1.0;
    o->intersect(o, &ro, &rv, &res, &nrm, &ptr);
    if (o->bounds.radius == 0)
        object_get_bounds(o);
    vectNd_free(&ro);
    vectNd_free(&rv);
    vectNd_free(&res);
    vectNd_free(&nrm);
}

static int subtree_size(kd_node_t *n)
{
    if (!n) return 0;
    return 1 + subtree_size(n->left) + subtree_size(n->right);
}

/* preorder numbering: node `me`, then its left subtree, then its right subtree */
static void dump_node(FILE *f, kd_node_t *n, int me)
{
    int left = -1, right = -1;
    if (n->left && n->right) {
        left = me + 1;
        right = left + subtree_size(n->left);
    } else if (n->left || n->right) {
        fprintf(stderr, "ref_shim: kd node with one child\n");
        exit(2);
    }
    fprintf(f, "kdnode %d dim %d boundary %a left %d right %d num %d ids", me, n->dim, n->boundary,
            left, right, n->num);
    for (int i = 0; i < n->num; ++i) {
        int id = n->obj_ids[i];
        if (id < 0 || id >= n_items || fobjs[id].obj != (object *)n->objs[i]) {
            fprintf(stderr, "ref_shim: kd leaf id/pointer mismatch\n");
            exit(2);
        }
        fprintf(f, " %d", id);
    }
    fprintf(f, "\n");
    if (left >= 0) {
        dump_node(f, n->left, left);
        dump_node(f, n->right, right);
    }
}

static int scene_v2 = 0;     /* --scene-v2: also write the camera2 block (aperture, fields of view, eyes, local axes) */

static void dump_scene(const char *path, scene *scn)
{
    int dims = scn->dimensions;
    FILE *f = fopen(path, "w");
    if (!f) { perror(path); exit(2); }
    fprintf(f, "ndtscene %d\n", scene_v2 ? 2 : 1);
    fprintf(f, "name %s\n", scn->name);
    fprintf(f, "dims %d\n", dims);
    fprintf(f, "camera type %d focal_distance %a\n", (int)scn->cam.type, scn->cam.focal_distance);
    put_vec(f, "cam_pos", &scn->cam.pos, dims);
    put_vec(f, "cam_img_orig", &scn->cam.imgOrig, dims);
    put_vec(f, "cam_dir_x", &scn->cam.dirX, dims);
    put_vec(f, "cam_dir_y", &scn->cam.dirY, dims);
    if (scene_v2) {
        fprintf(f, "camera2 aperture %a hfov %a vfov %a\n", scn->cam.aperture_radius, scn->cam.hFov, scn->cam.vFov);
        put_vec(f, "cam_left_eye", &scn->cam.leftEye, dims);
        put_vec(f, "cam_right_eye", &scn->cam.rightEye, dims);
        put_vec(f, "cam_local_x", &scn->cam.localX, dims);
        put_vec(f, "cam_local_y", &scn->cam.localY, dims);
        put_vec(f, "cam_local_z", &scn->cam.localZ, dims);
    }
    fprintf(f, "ambient %a %a %a\n", scn->ambient.red, scn->ambient.green, scn->ambient.blue);
    fprintf(f, "background %a %a %a %a\n", scn->bg_red, scn->bg_green, scn->bg_blue, scn->bg_alpha);
    fprintf(f, "lights %d\n", scn->num_lights);
    for (int i = 0; i < scn->num_lights; ++i) {
        light *l = scn->lights[i];
        int has_pos = (l->pos.v != NULL && l->pos.n == dims);
        int has_dir = (l->dir.v != NULL && l->dir.n == dims);
        int has_area = (l->type == LIGHT_DISK || l->type == LIGHT_RECT);
        fprintf(f, "light %d type %d color %a %a %a angle %a has_pos %d has_dir %d", i, (int)l->type,
                l->red, l->green, l->blue, l->angle, has_pos, has_dir);
        if (scene_v2) fprintf(f, " radius %a has_area %d", l->radius, has_area);
        fprintf(f, "\n");
        put_vec(f, "lpos", has_pos ? &l->pos : NULL, dims);
        put_vec(f, "ldir", has_dir ? &l->dir : NULL, dims);
        if (scene_v2 && has_area) {
            /* the basis apply_lights derives at the first shading evaluation (ndt.c:123-125, scene.c:182-195) */
            if (!l->prepared) scene_prepare_light(l);
            put_vec(f, "lu1", &l->u1, dims);
            put_vec(f, "lv1", &l->v1, dims);
        }
    }
    fprintf(f, "objects %d items %d\n", n_fobjs, n_items);
    for (int i = 0; i < n_fobjs; ++i) {
        object *o = fobjs[i].obj;
        char tn[OBJ_TYPE_MAX_LEN];
        type_of(o, tn, sizeof(tn));
        int nchild = 0;
        if (!strcmp(tn, "hcube"))
            nchild = o->n_obj;      /* faces from add_faces (hcube.c:34); hdisk's helper hplane is derived */
        fprintf(f, "object %d type %s parent %d transparent %d npos %d ndir %d nsize %d nflag %d nobj %d\n", i, tn,
                fobjs[i].parent, (int)o->transparent, o->n_pos, o->n_dir, o->n_size, o->n_flag, nchild);
        fprintf(f, "material %a %a %a %a %a %a %a\n", o->red, o->green, o->blue, o->red_r, o->green_r, o->blue_r,
                o->refract_index);
        fprintf(f, "bounds %a", o->bounds.radius);
        for (int k = 0; k < dims; ++k)
            fprintf(f, " %a", o->bounds.center.v[k]);
        fprintf(f, "\n");
        for (int k = 0; k < o->n_pos; ++k) put_vec(f, "pos", &o->pos[k], dims);
        for (int k = 0; k < o->n_dir; ++k) put_vec(f, "dir", &o->dir[k], dims);
        fprintf(f, "sizes");
        for (int k = 0; k < o->n_size; ++k) fprintf(f, " %a", o->size[k]);
        fprintf(f, "\nflags");
        for (int k = 0; k < o->n_flag; ++k) fprintf(f, " %d", o->flag[k]);
        fprintf(f, "\nchildren");
        for (int k = 0; k < nchild; ++k) fprintf(f, " %d", fobj_find(o->obj[k]));
        fprintf(f, "\n");
    }
    fprintf(f, "kdtree nodes %d obj_num %d\n", subtree_size(kdtree.root), kdtree.obj_num);
    dump_node(f, kdtree.root, 0);
    fprintf(f, "inf %d ids", kdtree.inf_obj_num);
    for (int i = 0; i < kdtree.inf_obj_num; ++i)
        fprintf(f, " %d", fobj_find((object *)kdtree.inf_obj_ptrs[i]));
    fprintf(f, "\n");
    put_vec(f, "bb_lower", &kdtree.bb.lower, dims);
    put_vec(f, "bb_upper", &kdtree.bb.upper, dims);
    fprintf(f, "end\n");
    fclose(f);
}

/* Fixed registration order, so that registered_types() (which scenes/random.c indexes,
 * random.c:51) does not depend on readdir order (object.c:141-153).  register_object
 * prepends (object.c:112-114); this list is the resulting registry order, registered in
 * reverse.  It is the order SURVEY.md Appendix B recorded. */
static const char *registry_order[] = { "hcylinder", "orthotope", "sphere", "hcube", "hdisk", "cluster",
                                        "hplane", "cylinder", "stubs", "hfacet", "facet" };

static const char *arg_str(int argc, char **argv, const char *key, const char *def)
{
    for (int i = 1; i + 1 < argc; ++i)
        if (!strcmp(argv[i], key))
            return argv[i + 1];
    return def;
}
static int arg_flag(int argc, char **argv, const char *key)
{
    for (int i = 1; i < argc; ++i)
        if (!strcmp(argv[i], key))
            return 1;
    return 0;
}

int main(int argc, char **argv)
{
    const char *objdir = arg_str(argc, argv, "--objects", NULL);
    const char *scene_so = arg_str(argc, argv, "--scene", NULL);
    int dims = atoi(arg_str(argc, argv, "--dims", "3"));
    int frame = atoi(arg_str(argc, argv, "--frame", "0"));
    const char *config = arg_str(argc, argv, "--config", NULL);
    int width = 0, height = 0;
    sscanf(arg_str(argc, argv, "--res", "64x64"), "%dx%d", &width, &height);
    int threads = atoi(arg_str(argc, argv, "--threads", "1"));
    int max_depth = atoi(arg_str(argc, argv, "--depth", "128"));
    const char *scene_out = arg_str(argc, argv, "--scene-out", NULL);
    const char *fb_out = arg_str(argc, argv, "--fb-out", NULL);
    const char *rays_in = arg_str(argc, argv, "--rays-in", NULL);
    const char *rays_out = arg_str(argc, argv, "--rays-out", NULL);
    const char *tmpdir = arg_str(argc, argv, "--tmp", "/tmp");
    int no_render = arg_flag(argc, argv, "--no-render");
    const char *aa = arg_str(argc, argv, "--aa", NULL);
    const char *yaml_out = arg_str(argc, argv, "--yaml-out", NULL);
    int stereo = atoi(arg_str(argc, argv, "--stereo", "0"));     /* stereo_mode, ndt.c:46-48 */
    int samples = atoi(arg_str(argc, argv, "--samples", "1"));   /* `-n`: > 1 draws from drand48, run with --threads 1 */
    const char *depth_out = arg_str(argc, argv, "--depth-out", NULL);
    scene_v2 = arg_flag(argc, argv, "--scene-v2");
    if (!objdir || !scene_so || width < 1 || height < 1) {
        fprintf(stderr, "usage: ndt_ref_shim --objects DIR --scene X.so --dims N [--frame F] [--config S] --res WxH\n"
                        "       [--threads T] [--depth L] [--scene-out F] [--fb-out F] [--rays-in F --rays-out F]\n"
                        "       [--tmp DIR] [--no-render] [--aa DIFF,DEPTH] [--yaml-out F]\n"
                        "       [--stereo MODE] [--depth-out F] [--scene-v2] [--samples N]\n");
        return 2;
    }

    /* plugins, in pinned order */
    int nreg = (int)(sizeof(registry_order) / sizeof(registry_order[0]));
    for (int i = nreg - 1; i >= 0; --i) {
        char path[4096];
        snprintf(path, sizeof(path), "%s/%s.so", objdir, registry_order[i]);
        if (register_object(path) != 0) {
            fprintf(stderr, "ref_shim: cannot register %s\n", path);
            return 2;
        }
    }

    void *dl = dlopen(scene_so, RTLD_NOW);
    if (!dl) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    int (*setup)(scene *, int, int, int, char *) = NULL;
    int (*frame_count)(int, char *) = NULL;
    *(void **)(&setup) = dlsym(dl, "scene_setup");
    *(void **)(&frame_count) = dlsym(dl, "scene_frames");
    if (!setup) { fprintf(stderr, "ref_shim: no scene_setup in %s\n", scene_so); return 2; }
    int frames = 300;                                   /* ndt.c:1395 */
    if (frame_count) frames = frame_count(dims, (char *)config);   /* ndt.c:1749-1752 */

    /* frames before the requested one still run scene_setup (ndt.c:1818-1825) */
    scene scn;
    for (int i = 0; i <= frame; ++i) {
        setup(&scn, dims, i, frames, (char *)config);
        if (i < frame) scene_free(&scn);
    }
    printf("ref_shim: scene '%s' %d objects %d lights\n", scn.name, scn.num_objects, scn.num_lights);
    if (yaml_out) scene_write_yaml(&scn, (char *)yaml_out);       /* what `ndt -y` writes per frame (ndt.c:1798-1809) */

    /* kd-tree build, as ndt.c:1899-1908 */
    kd_tree_init(&kdtree, scn.dimensions);
    kd_item_list_t kditems;
    kd_item_list_init(&kditems);
    for (int i = 0; i < scn.num_objects; ++i) {
        object *o = scn.object_ptrs[i];
        object_get_bounds(o);
        object_kdlist_add(&kditems, o, i);
    }
    kd_tree_build(&kdtree, &kditems);
    scene_validate_objects(&scn);       /* ndt.c:1913 */
    camera_aim(&scn.cam);               /* ndt.c:1925 */

    /* flat object table: kd items in id order, then nested primitives */
    n_items = kditems.n;
    for (int i = 0; i < kditems.n; ++i) {
        if (kditems.items[i]->id != i) { fprintf(stderr, "ref_shim: kd id order\n"); return 2; }
        fobj_add((object *)kditems.items[i]->obj_ptr, -1);
    }
    if (scene_out) {
        for (int i = 0; i < n_items; ++i)
            force_prepare(fobjs[i].obj, dims);
        for (int i = 0; i < n_items; ++i) {
            object *o = fobjs[i].obj;
            char tn[OBJ_TYPE_MAX_LEN];
            type_of(o, tn, sizeof(tn));
            if (!strcmp(tn, "hcube")) {
                for (int k = 0; k < o->n_obj; ++k) {
                    force_prepare(o->obj[k], dims);
                    fobj_add(o->obj[k], i);
                }
            }
        }
        dump_scene(scene_out, &scn);
    }

    /* recursive anti-aliasing: first pass of (width+1) x (height+1) corner samples (ndt.c:919-976),
     * then resample_pixel for every pixel (ndt.c:762-781), single-threaded, in doubles */
    if (aa) {
        int aa_diff = 20, aa_depth = 4;
        sscanf(aa, "%d,%d", &aa_diff, &aa_depth);
        recursive_aa = 1;
        image_t img;
        dbl_image_init(&img);
        image_set_size(&img, width + 1, height + 1);
        vectNd_scale(&scn.cam.dirX, width / (double)height, &scn.cam.dirX);      /* ndt.c:926 */
        n_trace_closest = n_trace_shadow = 0;
        counting = 1;
        double t0 = now_s();
        /* the stereo modes that split the image (ndt.c:913-916): render_image hands mode and scales to both passes */
        const double aa_xs = stereo == 1 ? 0.5 : 1.0, aa_ys = stereo == 2 ? 0.5 : 1.0;
        /* (frame packing, stereo 4, is left out: recursive_resample averages the never-set alpha of its blank-line samples) */
        if (stereo < 0 || stereo > 3) { fprintf(stderr, "ref_shim: --aa with --stereo %d is not wired\n", stereo); return 2; }
        {
            /* with an aperture every sample draws its lens point from the *rand48 stream (ndt.c:528): where it stands */
            unsigned short probe[3] = { 0, 0, 0 }, keep[3];
            unsigned short *old = seed48(probe);
            keep[0] = old[0]; keep[1] = old[1]; keep[2] = old[2];
            seed48(keep);
            printf("ref_shim: seed48 %u %u %u\n", keep[0], keep[1], keep[2]);
        }
        /* the depth map render_image makes beside an anti-aliased image: width x height, filled by the first pass
         * (ndt.c:930-935, 753-756; the corner samples of the last column and row fall outside it) */
        image_t aa_depth_img;
        if (depth_out) {
            dbl_image_init(&aa_depth_img);
            image_set_size(&aa_depth_img, width, height);
        }
        for (int j = 0; j < height + 1; ++j)
            render_line(&scn, width + 1, aa_xs, height + 1, aa_ys, j, stereo, 1, &img, depth_out ? &aa_depth_img : NULL, max_depth);
        if (depth_out) {
            FILE *f = fopen(depth_out, "wb");
            if (!f) { perror(depth_out); return 2; }
            for (long i = 0; i < (long)width * height; ++i)
                fwrite((double *)aa_depth_img.pixels + 4 * i, sizeof(double), 1, f);
            fclose(f);
            image_free(&aa_depth_img);
        }
        long long rays_pass1 = n_trace_closest + n_trace_shadow;
        double *out = malloc(sizeof(double) * (size_t)width * height * 4);
        long long resampled = 0;
        for (int j = 0; j < height; ++j)
            for (int i = 0; i < width; ++i) {
                dbl_pixel_t clr;
                resampled += resample_pixel(&scn, width, aa_xs, height, aa_ys, i, j, stereo, 1, aa_diff, aa_depth, &img, &clr, max_depth);
                double *q = out + ((size_t)j * width + i) * 4;
                q[0] = clr.r; q[1] = clr.g; q[2] = clr.b; q[3] = clr.a;
            }
        double t1 = now_s();
        counting = 0;
        printf("ref_shim: render_s %.6f threads 1 width %d height %d\n", t1 - t0, width, height);
        printf("ref_shim: aa_diff %d aa_depth %d pixels_resampled %lld rays_pass1 %lld\n", aa_diff, aa_depth, resampled, rays_pass1);
        printf("ref_shim: rays_closest %lld rays_shadow %lld rays_total %lld\n", n_trace_closest, n_trace_shadow,
               n_trace_closest + n_trace_shadow);
        if (fb_out) {
            FILE *f = fopen(fb_out, "wb");
            if (!f) { perror(fb_out); return 2; }
            fwrite(out, sizeof(double), (size_t)width * height * 4, f);
            fclose(f);
        }
        free(out);
        image_free(&img);
        no_render = 1;
    }

    /* render (ndt.c:1933).  img_copy is only filled when a file name is given (ndt.c:1024). */
    if (!no_render) {
        char name[4096];
        snprintf(name, sizeof(name), "%s/ndt_ref_shim_%d.jpg", tmpdir, (int)getpid());
        image_t img;
        dbl_image_init(&img);
        {
            /* the state of the *rand48 stream the render starts from (scene programs draw from it too,
             * e.g. scenes/random.c:51): seed48 hands back the old state, which is put straight back */
            unsigned short probe[3] = { 0, 0, 0 }, keep[3];
            unsigned short *old = seed48(probe);
            keep[0] = old[0]; keep[1] = old[1]; keep[2] = old[2];
            seed48(keep);
            printf("ref_shim: seed48 %u %u %u\n", keep[0], keep[1], keep[2]);
        }
        n_trace_closest = n_trace_shadow = 0;
        counting = 1;
        double t0 = now_s();
        char dname[4096];
        snprintf(dname, sizeof(dname), "%s/ndt_ref_shim_depth_%d.jpg", tmpdir, (int)getpid());
        image_t dimg;
        dbl_image_init(&dimg);
        render_image(&scn, name, depth_out ? dname : NULL, width, height, samples, stereo, threads, 20, 4, max_depth, &img,
                     depth_out ? &dimg : NULL);
        double t1 = now_s();
        counting = 0;
        while (image_active_saves() > 0) usleep(1000);   /* background save thread (image.c:750) */
        unlink(name);
        if (depth_out) {
            /* the depth map render_line filled (ndt.c:753-756): r = g = b = 1/distance of the primary hit */
            unlink(dname);
            FILE *f = fopen(depth_out, "wb");
            if (!f) { perror(depth_out); return 2; }
            for (long i = 0; i < (long)width * height; ++i)
                fwrite((double *)dimg.pixels + 4 * i, sizeof(double), 1, f);
            fclose(f);
            image_free(&dimg);
        }
        printf("ref_shim: render_s %.6f threads %d width %d height %d\n", t1 - t0, threads, width, height);
        printf("ref_shim: rays_closest %lld rays_shadow %lld rays_total %lld\n", n_trace_closest, n_trace_shadow,
               n_trace_closest + n_trace_shadow);
        if (fb_out) {
            if (img.width != width || img.height != height || img.pixel_width != (int)sizeof(dbl_pixel_t)) {
                fprintf(stderr, "ref_shim: unexpected image copy %dx%d pw %d\n", img.width, img.height, img.pixel_width);
                return 2;
            }
            FILE *f = fopen(fb_out, "wb");
            if (!f) { perror(fb_out); return 2; }
            fwrite(img.pixels, sizeof(double), (size_t)width * height * 4, f);
            fclose(f);
        }
        image_free(&img);
    }

    /* trace_kd known answers.  in: n records of (o[dims], v[dims], dist_limit);
     * out: n records of (ret, obj index, hit[dims], normal[dims]) as doubles. */
    if (rays_in && rays_out) {
        FILE *fi = fopen(rays_in, "rb");
        FILE *fo = fopen(rays_out, "wb");
        if (!fi || !fo) { perror("rays"); return 2; }
        int rec = 2 * dims + 1;
        double *in = malloc(rec * sizeof(double));
        double *out = malloc((2 + 2 * dims) * sizeof(double));
        vectNd o, v, hit, nrm;
        vectNd_calloc(&o, dims);
        vectNd_calloc(&v, dims);
        vectNd_calloc(&hit, dims);
        vectNd_calloc(&nrm, dims);
        long n = 0;
        while (fread(in, sizeof(double), rec, fi) == (size_t)rec) {
            for (int i = 0; i < dims; ++i) { o.v[i] = in[i]; v.v[i] = in[dims + i]; }
            vectNd_fill(&hit, 0.0);
            vectNd_fill(&nrm, 0.0);
            object *ptr = NULL;
            int ret = trace_kd(&o, &v, &kdtree, &hit, &nrm, &ptr, in[2 * dims]);
            out[0] = ret;
            out[1] = ptr ? (double)fobj_find(ptr) : -1.0;
            for (int i = 0; i < dims; ++i) { out[2 + i] = hit.v[i]; out[2 + dims + i] = nrm.v[i]; }
            fwrite(out, sizeof(double), 2 + 2 * dims, fo);
            ++n;
        }
        fclose(fi);
        fclose(fo);
        printf("ref_shim: traced %ld known-answer rays\n", n);
    }
    return 0;
}
