/* parity_zoo.c -- a scene program written for this repository's parity tests (original code,
 * ndt scene API: scene_setup(scene*, dims, frame, frames, config)).
 *
 * The reference's own scenes leave several branches of the hot path untouched: spot lights, a
 * LIGHT_AMBIENT entry next to scn->ambient, finite hcylinders (flag 0), infinite cylinders
 * (flag[1] = 1), hfacets with a computed normal (flag 0), refraction with total internal
 * reflection, mirror chains that run into the 1/512 weight cut-off, and a non-unit hplane
 * normal.  This scene has all of them, for any dims >= 3.  It is compiled against the
 * REFERENCE's headers by oracle/Makefile (-> oracle/_ref/scenes/parity_zoo.so) to generate
 * golden fixtures from the compiled reference, and against this repository's host API in
 * tests/test_host_api.py.  config: "mirror" makes two facing planes near-perfect mirrors.
 */
#include <stdio.h>
#include <string.h>
#include "../scene.h"

static void set_axis(vectNd *v, int dims, double a0, double a1, double a2, double rest)
{
    vectNd_reset(v);
    vectNd_set(v, 0, a0);
    vectNd_set(v, 1, a1);
    vectNd_set(v, 2, a2);
    for (int i = 3; i < dims; ++i) vectNd_set(v, i, rest * (1.0 + 0.25 * (i - 3)));
}

static object *add(scene *scn, int dims, char *type, double r, double g, double b, double refl)
{
    object *o = NULL;
    scene_alloc_object(scn, dims, &o, type);
    o->red = r; o->green = g; o->blue = b;
    o->red_r = refl; o->green_r = refl * 0.8; o->blue_r = refl * 0.6;
    return o;
}

int scene_frames(int dimensions, char *config)
{
    (void)dimensions; (void)config;
    return 4;
}

int scene_setup(scene *scn, int dims, int frame, int frames, char *config)
{
    (void)frames;
    int mirror = (config && strstr(config, "mirror")) ? 1 : 0;
    scene_init(scn, "parity_zoo", dims);
    scn->bg_red = 0.05; scn->bg_green = 0.1; scn->bg_blue = 0.2; scn->bg_alpha = 0.5;
    scn->ambient.red = 0.05; scn->ambient.green = 0.04; scn->ambient.blue = 0.03;

    vectNd p, q, d;
    vectNd_calloc(&p, dims);
    vectNd_calloc(&q, dims);
    vectNd_calloc(&d, dims);

    /* camera: off-axis in every dimension, with an up vector and a roll that changes per frame */
    camera_reset(&scn->cam);
    set_axis(&p, dims, 34.0, 14.0, 27.0, 3.0);
    set_axis(&q, dims, 0.5, 0.0, -0.5, 0.25);
    vectNd_reset(&d);
    vectNd_set(&d, 1, 1.0);
    camera_set_aim(&scn->cam, &p, &q, &d, 0.1 * frame);
    /* config "vr" / "pano": spherical / cylindrical screen (camera.c:506-555) with fields of view that
     * keep the whole zoo in the picture */
    /* config "dof": a lens (depth of field is sampled when -n > 1, ndt.c:527-542) focused short of the zoo */
    if (config && strstr(config, "dof")) {
        scn->cam.aperture_radius = 0.35;
        scn->cam.focal_distance = 38.0;
    }
    if (config && strstr(config, "vr")) {
        scn->cam.type = CAMERA_VR;
        scn->cam.hFov = 1.9;
        scn->cam.vFov = 1.1;
    } else if (config && strstr(config, "pano")) {
        scn->cam.type = CAMERA_PANO;
        scn->cam.hFov = 2.3;
        scn->cam.vFov = 0.9;
    }

    /* lights: ambient entry, point, spot (cone cuts through the scene), directional */
    light *l = NULL;
    scene_alloc_light(scn, &l);
    l->type = LIGHT_AMBIENT;
    l->red = 0.08; l->green = 0.08; l->blue = 0.1;

    scene_alloc_light(scn, &l);
    l->type = LIGHT_POINT;
    vectNd_calloc(&l->pos, dims);
    set_axis(&l->pos, dims, 12.0, 25.0, 18.0, 2.0);
    l->red = 260; l->green = 240; l->blue = 200;

    scene_alloc_light(scn, &l);
    l->type = LIGHT_SPOT;
    vectNd_calloc(&l->pos, dims);
    vectNd_calloc(&l->dir, dims);
    set_axis(&l->pos, dims, -6.0, 30.0, 4.0, -1.0);
    set_axis(&l->dir, dims, 0.25, -1.0, -0.1, 0.05);
    l->angle = 17.5;
    l->red = 500; l->green = 650; l->blue = 500;

    scene_alloc_light(scn, &l);
    l->type = LIGHT_DIRECTIONAL;
    vectNd_calloc(&l->dir, dims);
    set_axis(&l->dir, dims, -0.4, -1.0, -0.7, -0.2);
    l->red = 0.35; l->green = 0.35; l->blue = 0.45;

    /* config "area": a disk and a rectangle light (ndt.c:116-147), bases deliberately not orthogonal
     * (scene_prepare_light orthogonalises them, scene.c:182-195) */
    if (config && strstr(config, "area")) {
        scene_alloc_light(scn, &l);
        l->type = LIGHT_DISK;
        vectNd_calloc(&l->pos, dims);
        vectNd_calloc(&l->u, dims);
        vectNd_calloc(&l->v, dims);
        set_axis(&l->pos, dims, 4.0, 22.0, -9.0, 1.0);
        set_axis(&l->u, dims, 1.0, 0.1, 0.3, 0.0);
        set_axis(&l->v, dims, 0.2, 0.0, 1.0, 0.4);
        l->radius = 2.5;
        l->red = 180; l->green = 150; l->blue = 120;

        scene_alloc_light(scn, &l);
        l->type = LIGHT_RECT;
        vectNd_calloc(&l->pos, dims);
        vectNd_calloc(&l->u, dims);
        vectNd_calloc(&l->v, dims);
        set_axis(&l->pos, dims, -14.0, 18.0, 10.0, -2.0);
        set_axis(&l->u, dims, 0.0, 0.2, 1.0, 0.1);
        set_axis(&l->v, dims, 1.0, 0.5, 0.0, 0.0);
        l->radius = 3.0;
        l->red = 90; l->green = 140; l->blue = 200;
    }

    object *o;
    /* floor: hplane with a non-unit normal */
    o = add(scn, dims, "hplane", 0.7, 0.7, 0.65, mirror ? 0.97 : 0.35);
    set_axis(&p, dims, 0.0, -6.0, 0.0, 0.0);
    object_add_pos(o, &p);
    set_axis(&d, dims, 0.0, 2.5, 0.0, 0.0);
    object_add_dir(o, &d);
    if (mirror) {
        /* a facing mirror: rays bounce between the two until 1/512 or -l stops them */
        o = add(scn, dims, "hplane", 0.1, 0.1, 0.1, 0.97);
        set_axis(&p, dims, 0.0, 16.0, 0.0, 0.0);
        object_add_pos(o, &p);
        set_axis(&d, dims, 0.0, -1.0, 0.0, 0.0);
        object_add_dir(o, &d);
    }

    /* opaque and glass spheres */
    o = add(scn, dims, "sphere", 0.9, 0.2, 0.2, 0.2);
    set_axis(&p, dims, -3.0, -2.0, 2.0, 0.5);
    object_add_pos(o, &p);
    object_add_size(o, 4.0);

    o = add(scn, dims, "sphere", 0.9, 0.95, 1.0, 0.1);
    o->transparent = 1;
    o->refract_index = 1.45;
    set_axis(&p, dims, 7.0, -1.5, 6.0, 0.25);
    object_add_pos(o, &p);
    object_add_size(o, 4.5);

    /* a glass disk seen edge-on-ish: grazing refraction -> total internal reflection inside the sphere chain */
    o = add(scn, dims, "hdisk", 0.3, 0.8, 0.4, 0.0);
    o->transparent = 1;
    o->refract_index = 1.9;
    set_axis(&p, dims, 3.0, 3.0, -4.0, 0.0);
    object_add_pos(o, &p);
    set_axis(&d, dims, 0.3, 1.0, 0.2, 0.1);
    vectNd_unitize(&d);
    object_add_dir(o, &d);
    object_add_size(o, 5.0);

    /* finite cylinder and an infinite one (flag[1] = 1) */
    o = add(scn, dims, "cylinder", 0.2, 0.3, 0.9, 0.15);
    set_axis(&p, dims, -9.0, -6.0, -6.0, 0.0);
    set_axis(&q, dims, -7.0, 6.0, -9.0, 1.0);
    object_add_pos(o, &p);
    object_add_pos(o, &q);
    object_add_size(o, 1.25);
    object_add_flag(o, 0);

    o = add(scn, dims, "cylinder", 0.8, 0.6, 0.1, 0.0);
    set_axis(&p, dims, 14.0, -6.0, -12.0, 0.0);
    set_axis(&q, dims, 14.5, 6.0, -11.0, 0.5);
    object_add_pos(o, &p);
    object_add_pos(o, &q);
    object_add_size(o, 0.8);
    object_add_flag(o, 0);
    object_add_flag(o, 1);

    /* finite hcylinder (flag[0] = 0): needs dims-1 positions */
    o = add(scn, dims, "hcylinder", 0.6, 0.2, 0.7, 0.1);
    object_add_flag(o, 0);
    set_axis(&p, dims, 10.0, -4.0, -3.0, -0.5);
    object_add_pos(o, &p);
    for (int i = 0; i < dims - 2; ++i) {
        vectNd_copy(&q, &p);
        vectNd_set(&q, (i == 0) ? 1 : i + 1, q.v[(i == 0) ? 1 : i + 1] + 7.0 + i);
        vectNd_set(&q, 0, q.v[0] + 0.5 * (i + 1));
        object_add_pos(o, &q);
    }
    object_add_size(o, 1.5);

    /* a thin 2-D orthotope (a parallelogram plate) with non-orthogonal edges */
    o = add(scn, dims, "orthotope", 0.9, 0.8, 0.3, 0.25);
    object_add_flag(o, 2);
    set_axis(&p, dims, -12.0, -5.0, 6.0, 0.0);
    object_add_pos(o, &p);
    set_axis(&d, dims, 6.0, 0.5, 1.0, 0.25);
    object_add_dir(o, &d);
    set_axis(&d, dims, 1.0, 7.0, -1.5, 0.5);
    object_add_dir(o, &d);

    /* an hcube with skewed, non-unit directions (config "nohcube": left out -- it has every m-face for m = 2 .. N-1 as a
     * nested object, half a million of them in 12-D) */
    if (!(config && strstr(config, "nohcube"))) {
        o = add(scn, dims, "hcube", 0.2, 0.7, 0.7, 0.3);
        set_axis(&p, dims, 1.0, -2.5, -9.0, 0.5);
        object_add_pos(o, &p);
        for (int i = 0; i < dims; ++i) {
            vectNd_reset(&d);
            vectNd_set(&d, i, 1.0);
            vectNd_set(&d, (i + 1) % dims, 0.15);
            vectNd_unitize(&d);
            object_add_dir(o, &d);
            object_add_size(o, 3.0 + 0.5 * i);
        }
    }

    /* hfacets: one with interpolated vertex normals, one with the computed normal */
    for (int k = 0; k < 2; ++k) {
        o = add(scn, dims, "hfacet", 0.9, 0.5 + 0.3 * k, 0.2, 0.05);
        set_axis(&p, dims, -2.0 + 9.0 * k, 7.0, -2.0, 0.2);
        object_add_pos(o, &p);
        set_axis(&p, dims, 4.0 + 9.0 * k, 8.0, 1.0, 0.4);
        object_add_pos(o, &p);
        set_axis(&p, dims, 0.0 + 9.0 * k, 11.0, 4.0, 0.0);
        object_add_pos(o, &p);
        set_axis(&d, dims, 0.1, 1.0, 0.1, 0.0); vectNd_unitize(&d); object_add_dir(o, &d);
        set_axis(&d, dims, -0.2, 1.0, 0.0, 0.1); vectNd_unitize(&d); object_add_dir(o, &d);
        set_axis(&d, dims, 0.0, 1.0, -0.3, 0.0); vectNd_unitize(&d); object_add_dir(o, &d);
        object_add_flag(o, k == 0 ? 1 : 0);
    }

    /* facet (angle test) */
    o = add(scn, dims, "facet", 0.4, 0.9, 0.9, 0.4);
    set_axis(&p, dims, -14.0, 2.0, -4.0, 0.0);
    object_add_pos(o, &p);
    set_axis(&p, dims, -10.0, 9.0, -2.0, 0.0);
    object_add_pos(o, &p);
    set_axis(&p, dims, -13.0, 4.0, 5.0, 0.0);
    object_add_pos(o, &p);
    set_axis(&d, dims, 1.0, -0.2, 0.1, 0.0);
    vectNd_unitize(&d);
    for (int i = 0; i < 3; ++i) object_add_dir(o, &d);
    object_add_flag(o, 0);

    /* a cluster (flattened by the kd build) holding two small spheres, rotated per frame */
    object *cl = NULL;
    scene_alloc_object(scn, dims, &cl, "cluster");
    object_add_flag(cl, 2);
    for (int k = 0; k < 2; ++k) {
        object *s = object_alloc(dims, "sphere", k ? "cluster ball b" : "cluster ball a");
        s->red = 0.95; s->green = 0.95; s->blue = 0.2 + 0.6 * k;
        s->red_r = s->green_r = s->blue_r = 0.5;
        set_axis(&p, dims, 16.0 + 4.0 * k, -3.0 + 2.0 * k, 9.0, 0.3);
        object_add_pos(s, &p);
        object_add_size(s, 1.75);
        object_add_obj(cl, s);
    }
    set_axis(&p, dims, 18.0, -2.0, 9.0, 0.3);
    vectNd_reset(&d);
    vectNd_set(&d, 1, 1.0);
    set_axis(&q, dims, 1.0, 0.0, 1.0, 0.0);
    object_rotate2(cl, &p, &d, &q, 0.35 * frame);

    vectNd_free(&p);
    vectNd_free(&q);
    vectNd_free(&d);
    return 1;
}

int scene_cleanup(void) { return 0; }
