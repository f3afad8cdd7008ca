/* ndt_render.c -- render_image (reference ndt.c:900) for this host model: flatten, upload,
 * render on the GPU through the C ABI of include/ndt_hip.h.  No CPU rendering exists here. */
#include "ndt_host_internal.h"

/* one GPU context (stream + workspace) per host thread: frames rendered from different threads overlap on the GPU */
static __thread ndt_hip_ctx *g_ctx = NULL;

int ndt_render_image(scene *scn, int width, int height, int threads, int max_optic_depth, double *rgba)
{
    return ndt_render_image_aa(scn, width, height, threads, -1, -1, max_optic_depth, rgba);
}

/* render_image with the reference's `-a diff,depth` (recursive_aa, ndt.c:44, 1039-1087); aa_depth < 0 = plain */
int ndt_render_image_aa(scene *scn, int width, int height, int threads, int aa_diff, int aa_depth, int max_optic_depth,
                        double *rgba)
{
    return ndt_render_image_full(scn, width, height, 1, threads, aa_diff, aa_depth, 0, 1, max_optic_depth, rgba, NULL);
}

/* everything render_image takes (ndt.c:900): samples = `-n`, stereo = the reference's stereo_mode (MONO ..
 * ANAGLYPH_3D), specular = specular_enabled (`-p` clears it), depth = the depth map of `-z` (width*height doubles) or NULL */
int ndt_render_image_full(scene *scn, int width, int height, int samples, int threads, int aa_diff, int aa_depth, int stereo,
                          int specular, int max_optic_depth, double *rgba, double *depth)
{
    (void)threads;      /* the pthread fan-out of ndt.c:949-975 is the GPU's job now */
    char err[256];
    ndt_flat_builder fb;
    if (ndt_flatten_scene(scn, &fb, err, sizeof(err)) != 0) {
        fprintf(stderr, "ndt_render_image: %s\n", err);
        ndt_flat_builder_free(&fb);
        return 0;
    }
    int ok = 0;
    if (!g_ctx && ndt_hip_create(0, &g_ctx) != NDT_OK) {
        fprintf(stderr, "ndt_render_image: %s\n", ndt_hip_last_error());
    } else if (ndt_hip_upload_scene(g_ctx, &fb.fs) != NDT_OK) {
        fprintf(stderr, "ndt_render_image: %s\n", ndt_hip_last_error());
    } else {
        ndt_render_params p;
        memset(&p, 0, sizeof(p));
        p.width = width; p.height = height; p.max_optic_depth = max_optic_depth; p.samples = samples > 1 ? samples : 1;
        p.row_begin = 0; p.row_step = 1; p.specular = specular;
        p.stereo = stereo;
        if (aa_depth >= 0 && aa_diff < 256) {       /* ndt.c:1040: otherwise the first pass is the image */
            p.recursive_aa = 1;
            p.aa_diff = aa_diff;
            p.aa_depth = aa_depth;
        }
        if (ndt_hip_render_depth(g_ctx, &p, rgba, depth, NULL) == NDT_OK) ok = 1;
        else fprintf(stderr, "ndt_render_image: %s\n", ndt_hip_last_error());
    }
    ndt_flat_builder_free(&fb);
    return ok;
}
