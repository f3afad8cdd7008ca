/* blob.c -- an object plugin written for this repository's tests (original code, ndt object-plugin API: object.c:51-111
 * names the entry points).  A type of its own, "blob": the host side (parameter counts, bounding points) is all a scene
 * program needs to build a scene with it; its intersect() is host code, which the device cannot run. */
#include <stdio.h>
#include "../object.h"

int type_name(char *name, int size)
{
    snprintf(name, (size_t)size, "%s", NDT_TEST_PLUGIN_TYPE);
    return 0;
}

int params(object *obj, int *n_pos, int *n_dir, int *n_size, int *n_flags, int *n_obj)
{
    if (!obj) return -1;
    *n_pos = 1; *n_dir = 0; *n_size = 1; *n_flags = 0; *n_obj = 0;
    return 0;
}

int bounding_points(object *obj, bounds_list *list)
{
    bounds_list_add(list, &obj->pos[0], obj->size[0]);
    return 1;
}

int intersect(object *obj, vectNd *o, vectNd *v, vectNd *res, vectNd *normal, object **ptr)
{
    (void)obj; (void)o; (void)v; (void)res; (void)normal; (void)ptr;
    return 0;
}
