/* plug_scene.c -- a scene with one object of a plugin type (config = the type's name) beside a built-in sphere. */
#include <stdio.h>
#include <string.h>
#include "../scene.h"

int scene_frames(int dimensions, char *config) { (void)dimensions; (void)config; return 1; }

int scene_setup(scene *scn, int dims, int frame, int frames, char *config)
{
    (void)frame; (void)frames;
    scene_init(scn, "plug_scene", dims);
    vectNd p, q, up;
    vectNd_calloc(&p, dims);
    vectNd_calloc(&q, dims);
    vectNd_calloc(&up, dims);
    camera_reset(&scn->cam);
    vectNd_set(&p, 0, 20.0);
    vectNd_set(&up, 1, 1.0);
    camera_set_aim(&scn->cam, &p, &q, &up, 0.0);
    object *o = NULL;
    scene_alloc_object(scn, dims, &o, "sphere");
    object_add_pos(o, &q);
    object_add_size(o, 2.0);
    o->red = 0.8; o->green = 0.3; o->blue = 0.2;
    scene_alloc_object(scn, dims, &o, config && *config ? config : "sphere");
    vectNd_set(&q, 1, 5.0);
    object_add_pos(o, &q);
    object_add_size(o, 1.0);
    o->red = 0.2; o->green = 0.3; o->blue = 0.8;
    light *l = NULL;
    scene_alloc_light(scn, &l);
    l->type = LIGHT_POINT;
    vectNd_calloc(&l->pos, dims);
    vectNd_set(&l->pos, 1, 30.0);
    l->red = l->green = l->blue = 300.0;
    vectNd_free(&p); vectNd_free(&q); vectNd_free(&up);
    return 0;
}
