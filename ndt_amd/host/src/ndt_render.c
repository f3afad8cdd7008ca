/* ndt_render.c -- render_image (reference ndt.c:900) for this host model: flatten, upload,
 * render on the GPU through the C ABI of include/ndt_hip.h.  No CPU rendering exists here. */
#include "ndt_host_internal.h"
#include <time.h>

/* GPU contexts (stream + workspace) of the calling host thread: frames rendered from different threads overlap on the
 * GPU, or run on different GPUs (ndt_render_use_device); one frame may be spread over several (ndt_render_use_devices) */
#define NDT_MAX_CTX 64
static __thread ndt_hip_ctx *g_ctx[NDT_MAX_CTX];
static __thread int g_n_ctx = 0;            /* contexts that exist */
static __thread int g_want_ctx = 1;         /* contexts a frame is spread over */
static __thread int g_first_device = 0;
static __thread int g_paths_said = 0;

static void drop_contexts(void)
{
    for (int k = 0; k < g_n_ctx; ++k) ndt_hip_destroy(g_ctx[k]);
    g_n_ctx = 0;
    g_paths_said = 0;
}

void ndt_render_use_device(int device)
{
    if (device < 0) device = 0;
    if (g_want_ctx != 1 || g_first_device != device) drop_contexts();
    g_want_ctx = 1;
    g_first_device = device;
}

void ndt_render_use_devices(int n_contexts)
{
    if (n_contexts < 1) n_contexts = 1;
    if (n_contexts > NDT_MAX_CTX) n_contexts = NDT_MAX_CTX;
    if (g_want_ctx != n_contexts || g_first_device != 0) drop_contexts();
    g_want_ctx = n_contexts;
    g_first_device = 0;
}

/* the thread's contexts, created on first use: context k on device (first + k) mod device count */
static int have_contexts(void)
{
    int n_dev = ndt_hip_device_count();
    if (n_dev < 1) n_dev = 1;               /* ndt_hip_create then says why there is no device */
    while (g_n_ctx < g_want_ctx) {
        if (ndt_hip_create((g_first_device + g_n_ctx) % n_dev, &g_ctx[g_n_ctx]) != NDT_OK) return 0;
        ++g_n_ctx;
    }
    return 1;
}

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

static int render_any(scene *scn, int width, int height, int samples, int aa_diff, int aa_depth, int stereo, int specular,
                      int max_optic_depth, int format, void *out, double *depth, int threads)
{
    char err[256];
    ndt_flat_builder fb;
    /* NDT_HOST_TIMING=1: where a frame's host time goes (stderr) */
    /* (read per call: a static written by concurrent host threads -- ndt_hip -j K -- would be a data race) */
    const int timing = getenv("NDT_HOST_TIMING") != NULL;
    struct timespec ts0, ts1, ts2, ts3;
    if (timing) clock_gettime(CLOCK_MONOTONIC, &ts0);
    if (ndt_flatten_scene_mt(scn, &fb, err, sizeof(err), threads) != 0) {
        fprintf(stderr, "ndt_render_image: %s\n", err);
        ndt_flat_builder_free(&fb);
        return 0;
    }
    if (timing) clock_gettime(CLOCK_MONOTONIC, &ts1);
    int ok = have_contexts();
    for (int k = 0; ok && k < g_n_ctx; ++k)
        if (ndt_hip_upload_scene(g_ctx[k], &fb.fs) != NDT_OK) ok = 0;
    if (timing) clock_gettime(CLOCK_MONOTONIC, &ts2);
    if (ok) {
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
        if (depth)      /* the depth map comes from the one-context call (a map is not split over devices) */
            ok = format == NDT_IMAGE_F64 && ndt_hip_render_depth(g_ctx[0], &p, (double *)out, depth, NULL) == NDT_OK;
        else
            ok = ndt_hip_render_multi(g_ctx, g_n_ctx, &p, format, out, NULL) == NDT_OK;
        if (ok && g_n_ctx > 1 && !g_paths_said) {
            /* once per thread: how every context's rows reached the frame (an N-GPU run is diagnosable from its log) */
            static const char *const names[] = { "no rows", "same device", "peer stores", "staged copy" };
            g_paths_said = 1;
            fprintf(stderr, "ndt_render_image: one frame over %d contexts:", g_n_ctx);
            for (int k = 0; k < g_n_ctx; ++k) {
                const int path = ndt_hip_multi_path_taken(g_ctx[k]);
                fprintf(stderr, " [%d] GPU %d %s%s", k, ndt_hip_device(g_ctx[k]), names[path >= 0 && path <= 3 ? path : 0], k + 1 < g_n_ctx ? "," : "\n");
            }
        }
    }
    if (!ok) fprintf(stderr, "ndt_render_image: %s\n", ndt_hip_last_error());
    if (timing) {
        clock_gettime(CLOCK_MONOTONIC, &ts3);
#define NDT_MS(a, b) (((b).tv_sec - (a).tv_sec) * 1e3 + ((b).tv_nsec - (a).tv_nsec) * 1e-6)
        fprintf(stderr, "ndt_render_image: bounds + kd-tree + flatten %.2f ms, upload %.2f ms, render + image to host %.2f ms\n",
                NDT_MS(ts0, ts1), NDT_MS(ts1, ts2), NDT_MS(ts2, ts3));
    }
    ndt_flat_builder_free(&fb);
    return ok;
}

/* everything render_image takes (ndt.c:900): samples = `-n`, stereo = the reference's stereo_mode (MONO ..
 * ANAGLYPH_3D), specular = specular_enabled (`-p` clears it), depth = the depth map of `-z` (width*height doubles) or NULL */
int ndt_render_image_full(scene *scn, int width, int height, int samples, int threads, int aa_diff, int aa_depth, int stereo,
                          int specular, int max_optic_depth, double *rgba, double *depth)
{
    /* (the pthread fan-out of ndt.c:949-975 is the GPU's job now; `threads` fits the scene's bounding spheres in parallel) */
    return render_any(scn, width, height, samples, aa_diff, aa_depth, stereo, specular, max_optic_depth, NDT_IMAGE_F64, rgba, depth, threads);
}

int ndt_render_image_rgba8(scene *scn, int width, int height, int samples, int threads, int aa_diff, int aa_depth, int stereo,
                           int specular, int max_optic_depth, unsigned char *rgba8)
{
    return render_any(scn, width, height, samples, aa_diff, aa_depth, stereo, specular, max_optic_depth, NDT_IMAGE_RGBA8, rgba8, NULL, threads);
}
