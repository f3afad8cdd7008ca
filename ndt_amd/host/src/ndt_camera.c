/* ndt_camera.c -- camera model and aiming (reference camera.c), restated.  Aiming runs once per
 * frame on the host; only its results (pos, imgOrig, dirX, dirY, focal distance) reach the
 * device, so every rotation is applied in the reference's order to get the same vectors. */
#include "ndt_host_api.h"

const char *CAMERA_TYPE_STRING[] = { "CAMERA_NORMAL", "CAMERA_VR", "CAMERA_PANO" };

int camera_alloc(camera *cam, int dim)
{
    vectNd *all[] = { &cam->viewPoint, &cam->viewTarget, &cam->up, &cam->pos, &cam->leftEye, &cam->rightEye,
                      &cam->dirX, &cam->dirY, &cam->imgOrig, &cam->localX, &cam->localY, &cam->localZ };
    for (int i = 0; i < 12; ++i) vectNd_calloc(all[i], dim);
    camera_init(cam);
    cam->prepared = 0;
    return 1;
}

int camera_free(camera *cam)
{
    vectNd *all[] = { &cam->viewPoint, &cam->viewTarget, &cam->up, &cam->pos, &cam->leftEye, &cam->rightEye,
                      &cam->dirX, &cam->dirY, &cam->imgOrig, &cam->localX, &cam->localY, &cam->localZ };
    for (int i = 0; i < 12; ++i) vectNd_free(all[i]);
    return 1;
}

/* unit camera at the origin looking along axis 2, screen at distance 2 (camera.c:62-106) */
int camera_init(camera *cam)
{
    cam->type = CAMERA_NORMAL;
    vectNd_reset(&cam->viewPoint);
    vectNd_reset(&cam->viewTarget);
    vectNd_reset(&cam->up);
    cam->rotation = 0.0;
    cam->eye_offset = EYE_OFFSET;
    cam->zoom = 1.0;
    cam->flip_x = cam->flip_y = cam->flatten = 0;
    cam->leveling = 0.0;
    vectNd_reset(&cam->pos);
    vectNd_reset(&cam->leftEye);  vectNd_set(&cam->leftEye, 0, -EYE_OFFSET);
    vectNd_reset(&cam->rightEye); vectNd_set(&cam->rightEye, 0, EYE_OFFSET);
    vectNd_reset(&cam->dirX);     vectNd_set(&cam->dirX, 0, 1.0);
    vectNd_reset(&cam->dirY);     vectNd_set(&cam->dirY, 1, 1.0);
    vectNd_reset(&cam->imgOrig);  vectNd_set(&cam->imgOrig, 2, 2.0);
    vectNd_reset(&cam->localX);   vectNd_set(&cam->localX, 0, 1.0);
    vectNd_reset(&cam->localY);   vectNd_set(&cam->localY, 1, 1.0);
    vectNd_reset(&cam->localZ);   vectNd_set(&cam->localZ, 2, 1.0);
    cam->hFov = 2.0 * M_PI;
    cam->vFov = M_PI / 2.0;
    cam->focal_distance = 100.0;
    cam->aperture_radius = 0.0;
    cam->prepared = 0;
    return 1;
}

/* back to the origin, keeping the screen's size and distance (camera.c:108-130) */
int camera_reset(camera *cam)
{
    double focal = 0, xlen = 0, ylen = 0;
    cam->prepared = 0;
    vectNd_dist(&cam->pos, &cam->imgOrig, &focal);
    vectNd_l2norm(&cam->dirX, &xlen);
    vectNd_l2norm(&cam->dirY, &ylen);
    camera_init(cam);
    vectNd_reset(&cam->dirX);    vectNd_set(&cam->dirX, 0, xlen);
    vectNd_reset(&cam->dirY);    vectNd_set(&cam->dirY, 1, ylen);
    vectNd_reset(&cam->imgOrig); vectNd_set(&cam->imgOrig, 2, focal);
    cam->hFov = 2.0 * M_PI;
    cam->vFov = M_PI / 2.0;
    return 1;
}

int camera_set_aim(camera *cam, vectNd *pos, vectNd *target, vectNd *up, double rot)
{
    camera_reset(cam);
    vectNd_copy(&cam->viewPoint, pos);
    vectNd_copy(&cam->viewTarget, target);
    if (up) vectNd_copy(&cam->up, up);
    cam->rotation = rot;
    cam->eye_offset = EYE_OFFSET;
    return 0;
}
int camera_set_zoom(camera *cam, double zoom) { cam->zoom = zoom; return 0; }
int camera_set_flip(camera *cam, int x, int y) { cam->flip_x = x; cam->flip_y = y; return 0; }

void camera_flip_x(camera *cam)
{
    vectNd_scale(&cam->dirX, -1, &cam->dirX);
    vectNd tmp;
    vectNd_calloc(&tmp, cam->leftEye.n);
    vectNd_copy(&tmp, &cam->leftEye);
    vectNd_copy(&cam->leftEye, &cam->rightEye);
    vectNd_copy(&cam->rightEye, &tmp);
    vectNd_free(&tmp);
}
void camera_flip_y(camera *cam) { vectNd_scale(&cam->dirY, -1, &cam->dirY); }
void camera_zoom(camera *cam)
{
    if (fabs(cam->zoom) < EPSILON) return;
    vectNd_scale(&cam->dirX, 1 / cam->zoom, &cam->dirX);
    vectNd_scale(&cam->dirY, 1 / cam->zoom, &cam->dirY);
}

/* ---- camera_aim_naive (camera.c:180-327) as a rig of points that turn together.
 *
 * The default camera looks down one axis from the origin.  Aiming it = moving it to viewPoint and turning it, one
 * coordinate plane at a time, until the screen centre lies in the direction of the target.  What turns is a RIG: the
 * screen centre, one point a screen-width to its right and one a screen-height above it (the screen axes are read back
 * from them afterwards), and the two eyes.  The order of the turns and the snapping of small offsets to zero decide the
 * last bits of the aimed camera -- and through it every pixel -- so both are the reference's. */

/* direction of `to` seen from `from`, inside the (i, j) coordinate plane; offsets below EPSILON count as none */
static double bearing_in_plane(const vectNd *from, const vectNd *to, int i, int j)
{
    double along = to->v[i] - from->v[i];
    double across = to->v[j] - from->v[j];
    if (fabs(across) < EPSILON) across = 0;
    if (fabs(along) < EPSILON) along = 0;
    return atan2(across, along);
}

#define RIG_POINTS 5
static void rig_turn(vectNd *const rig[RIG_POINTS], const vectNd *about, int i, int j, double angle)
{
    for (int k = 0; k < RIG_POINTS; ++k) vectNd_rotate(rig[k], (vectNd *)about, i, j, angle, rig[k]);
}

int camera_aim_naive(camera *cam)
{
    const int dim = cam->pos.n;
    /* what the caller configured survives the reset below */
    const camera keep = *cam;           /* scalars only are read from this copy: its vectors alias cam's */
    vectNd from, target, right, above;
    vectNd_calloc(&from, dim);
    vectNd_calloc(&target, dim);
    vectNd_copy(&from, &cam->viewPoint);
    vectNd_copy(&target, &cam->viewTarget);
    const double roll = keep.rotation + keep.leveling;

    camera_reset(cam);
    cam->type = keep.type;
    vectNd_copy(&cam->viewPoint, &from);
    vectNd_copy(&cam->viewTarget, &target);
    cam->rotation = roll;               /* sic: the leveling is folded into the rotation (camera.c:215) */
    cam->eye_offset = EYE_OFFSET;
    cam->zoom = keep.zoom;
    cam->flip_x = keep.flip_x; cam->flip_y = keep.flip_y; cam->flatten = keep.flatten;
    cam->hFov = keep.hFov; cam->vFov = keep.vFov;
    cam->aperture_radius = keep.aperture_radius;
    cam->focal_distance = keep.focal_distance;

    /* the screen goes out to the target's distance, keeping its opening angle */
    double reach = 0.0, default_reach = 0.0;
    vectNd_dist(&from, &target, &reach);
    vectNd_l2norm(&cam->imgOrig, &default_reach);
    vectNd_unitize(&cam->imgOrig);
    vectNd_scale(&cam->imgOrig, reach, &cam->imgOrig);
    vectNd_scale(&cam->dirX, reach / default_reach, &cam->dirX);
    vectNd_scale(&cam->dirY, reach / default_reach, &cam->dirY);

    /* the rig, moved to the view point */
    vectNd_alloc(&right, dim);
    vectNd_add(&cam->imgOrig, &cam->dirX, &right);
    vectNd_alloc(&above, dim);
    vectNd_add(&cam->imgOrig, &cam->dirY, &above);
    vectNd_add(&cam->pos, &from, &cam->pos);
    vectNd_add(&cam->leftEye, &from, &cam->leftEye);
    vectNd_add(&cam->rightEye, &from, &cam->rightEye);
    vectNd_add(&right, &from, &right);
    vectNd_add(&above, &from, &above);
    vectNd_add(&cam->imgOrig, &from, &cam->imgOrig);
    vectNd *const rig[RIG_POINTS] = { &right, &above, &cam->imgOrig, &cam->leftEye, &cam->rightEye };

    /* the roll first, in the screen's own plane; then every ordered pair of axes */
    rig_turn(rig, &cam->pos, 0, 1, roll);
    for (int i = 0; i < dim; ++i)
        for (int j = 0; j < dim; ++j) {
            if (i == j) continue;
            const double have = bearing_in_plane(&cam->pos, &cam->imgOrig, i, j);
            double want = bearing_in_plane(&cam->pos, &target, i, j);
            if (want < have) want += 2 * M_PI;
            rig_turn(rig, &cam->pos, i, j, want - have);
        }

    /* the screen axes and the camera's local frame, read back from the rig */
    vectNd_sub(&right, &cam->imgOrig, &cam->dirX);
    vectNd_sub(&above, &cam->imgOrig, &cam->dirY);
    vectNd_copy(&cam->localX, &cam->dirX);
    vectNd_copy(&cam->localY, &cam->dirY);
    vectNd_sub(&cam->imgOrig, &cam->pos, &cam->localZ);
    vectNd_unitize(&cam->localX);
    vectNd_unitize(&cam->localY);
    vectNd_unitize(&cam->localZ);
    cam->prepared = 1;
    vectNd_free(&right); vectNd_free(&above); vectNd_free(&from); vectNd_free(&target);
    if (keep.flip_x) camera_flip_x(cam);
    if (keep.flip_y) camera_flip_y(cam);
    if (keep.zoom != 1.0) camera_zoom(cam);
    return 1;
}

/* camera.c:132-178: search the roll that brings the screen's Y axis closest to `up` (step
 * pi/10, halved and reversed whenever the angle stops improving), then aim with it */
int camera_aim(camera *cam)
{
    double up_len = 0.0;
    vectNd_l2norm(&cam->up, &up_len);
    if (up_len > 0) {
        vectNd up;
        vectNd_calloc(&up, cam->up.n);
        vectNd_copy(&up, &cam->up);
        double curr = 0, delta = M_PI / 10, angle = 0, last_angle = 0;
        camera probe;
        camera_alloc(&probe, cam->viewPoint.n);
        camera_set_aim(&probe, &cam->viewPoint, &cam->viewTarget, &cam->up, 0.0);
        camera_aim_naive(&probe);
        vectNd_angle(&up, &probe.dirY, &angle);
        while (fabs(delta) > (EPSILON / 1000)) {
            last_angle = angle;
            camera_set_aim(&probe, &cam->viewPoint, &cam->viewTarget, &cam->up, curr);
            camera_aim_naive(&probe);
            vectNd_angle(&up, &probe.dirY, &angle);
            if (angle >= last_angle) delta = -delta / 2.0;
            curr += delta;
        }
        cam->leveling = curr;
        camera_free(&probe);
        vectNd_free(&up);
    }
    return camera_aim_naive(cam);
}

int camera_focus(camera *cam, vectNd *point)
{
    vectNd t;
    vectNd_alloc(&t, point->n);
    vectNd_sub(point, &cam->pos, &t);
    vectNd_proj(&t, &cam->localZ, &t);
    vectNd_l2norm(&t, &cam->focal_distance);
    vectNd_free(&t);
    return 0;
}

/* planar screen only; VR / panorama cameras are outside the device path (camera.c:557-575) */
void camera_target_point(camera *cam, double x, double y, double dist, vectNd *pixel)
{
    vectNd t;
    vectNd_alloc(&t, pixel->n);
    vectNd_copy(pixel, &cam->imgOrig);
    vectNd_scale(&cam->dirX, x, &t);
    vectNd_add(pixel, &t, pixel);
    vectNd_scale(&cam->dirY, y, &t);
    vectNd_add(pixel, &t, pixel);
    double screen_dist = -1;
    vectNd_dist(&cam->imgOrig, &cam->pos, &screen_dist);
    if (screen_dist > EPSILON) {
        vectNd_sub(pixel, &cam->pos, &t);
        vectNd_scale(&t, dist / screen_dist, &t);
        vectNd_add(&cam->pos, &t, pixel);
    }
    vectNd_free(&t);
}

void camera_print(camera *cam)
{
    printf("Camera points:\n");
    vectNd_print(&cam->viewPoint, "\tviewPoint");
    vectNd_print(&cam->viewTarget, "\tviewTarget");
    vectNd_print(&cam->up, "\tup");
    vectNd_print(&cam->pos, "\tposition");
    vectNd_print(&cam->imgOrig, "\timage origin");
    vectNd_print(&cam->dirX, "\timg X");
    vectNd_print(&cam->dirY, "\timg Y");
}
