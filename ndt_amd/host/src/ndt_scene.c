/* ndt_scene.c -- scene container (reference scene.c:23-260), restated. */
#include "ndt_host_api.h"

const char *LIGHT_TYPE_STRING[] = { "LIGHT_AMBIENT", "LIGHT_POINT", "LIGHT_DIRECTIONAL", "LIGHT_SPOT", "LIGHT_DISK",
                                    "LIGHT_RECT" };

int scene_init(scene *scn, char *name, int dim)
{
    memset(scn, 0, sizeof(scene));
    strncpy(scn->name, name, sizeof(scn->name));
    scn->name[sizeof(scn->name) - 1] = '\0';
    camera_alloc(&scn->cam, dim);
    camera_init(&scn->cam);
    scn->dimensions = dim;
    scn->bg_alpha = 1.0;
    return 1;
}

int scene_free_light(light *lgt)
{
    if (lgt->type == LIGHT_POINT || lgt->type == LIGHT_SPOT || lgt->type == LIGHT_DISK || lgt->type == LIGHT_RECT)
        vectNd_free(&lgt->pos);
    if (lgt->type == LIGHT_DIRECTIONAL || lgt->type == LIGHT_SPOT) vectNd_free(&lgt->dir);
    if (lgt->type == LIGHT_DISK || lgt->type == LIGHT_RECT) {
        vectNd_free(&lgt->u); vectNd_free(&lgt->v); vectNd_free(&lgt->u1); vectNd_free(&lgt->v1);
    }
    if (lgt->target.n > 0) vectNd_free(&lgt->target);
    return 0;
}

int scene_free(scene *scn)
{
    for (int i = 0; i < scn->num_objects; ++i) object_free(scn->object_ptrs[i]);
    free(scn->object_ptrs);
    scn->object_ptrs = NULL;
    for (int i = 0; i < scn->num_lights; ++i) {
        scene_free_light(scn->lights[i]);
        free(scn->lights[i]);
    }
    free(scn->lights);
    scn->lights = NULL;
    camera_free(&scn->cam);
    return 1;
}

/* scene.c:62-75 */
int scene_add_object(scene *scn, object *obj)
{
    object **grown = (object **)realloc(scn->object_ptrs, ((size_t)scn->num_objects + 1) * sizeof(object *));
    if (!grown) return 0;
    scn->object_ptrs = grown;
    scn->object_ptrs[scn->num_objects++] = obj;
    return 1;
}

int scene_alloc_object(scene *scn, int dimensions, object **obj, char *type)
{
    *obj = object_alloc(dimensions, type, "unnamed");
    if (!*obj) return 0;
    if (!scene_add_object(scn, *obj)) { free(*obj); *obj = NULL; return 0; }
    return 1;
}

int scene_remove_object(scene *scn, object *obj)
{
    for (int i = 0; i < scn->num_objects; ++i) {
        if (scn->object_ptrs[i] != obj) continue;
        for (int j = i; j < scn->num_objects - 1; ++j) scn->object_ptrs[j] = scn->object_ptrs[j + 1];
        scn->object_ptrs[--scn->num_objects] = NULL;
        --i;
    }
    return 0;
}

int scene_alloc_light(scene *scn, light **lgt)
{
    *lgt = (light *)calloc(1, sizeof(light));
    light **grown = (light **)realloc(scn->lights, ((size_t)scn->num_lights + 1) * sizeof(light *));
    if (!grown) return 0;
    scn->lights = grown;
    scn->lights[scn->num_lights++] = *lgt;
    (*lgt)->type = LIGHT_POINT;
    return 1;
}

/* scene.c:148-180 */
int scene_aim_light(light *lgt, vectNd *target)
{
    vectNd aim, tmp;
    vectNd_calloc(&aim, target->n);
    vectNd_sub(target, &lgt->pos, &aim);
    vectNd_unitize(&aim);
    vectNd_alloc(&tmp, target->n);
    vectNd_copy(&tmp, &aim);
    vectNd_set(&tmp, 0, (fabs(aim.v[0]) < EPSILON) ? 1.0 : -aim.v[0]);
    vectNd_orthogonalize(&tmp, &aim, &lgt->u, NULL);
    vectNd_copy(&tmp, &aim);
    vectNd_set(&tmp, 1, (fabs(aim.v[1]) < EPSILON) ? 1.0 : -aim.v[1]);
    vectNd_orthogonalize(&tmp, &aim, &lgt->v, NULL);
    vectNd_free(&tmp);
    vectNd_free(&aim);
    return 0;
}

int scene_prepare_light(light *lgt)
{
    if (lgt->type == LIGHT_DISK || lgt->type == LIGHT_RECT) {
        vectNd_alloc(&lgt->u1, lgt->pos.n);
        vectNd_alloc(&lgt->v1, lgt->pos.n);
        vectNd_orthogonalize(&lgt->u, &lgt->v, &lgt->u1, &lgt->v1);
        vectNd_unitize(&lgt->u1);
        vectNd_unitize(&lgt->v1);
    }
    lgt->prepared = 1;
    return 0;
}

int scene_validate_objects(scene *scn)
{
    for (int i = 0; i < scn->num_objects; ++i) {
        if (object_validate(scn->object_ptrs[i]) != 0) {
            fprintf(stderr, "Unable to validate object %i.\n", i);
            return -1;
        }
    }
    return 0;
}

int scene_print(scene *scn)
{
    printf("scene '%s': %d dimensions, %d objects, %d lights\n", scn->name, scn->dimensions, scn->num_objects, scn->num_lights);
    camera_print(&scn->cam);
    return 0;
}
