/* ndt_main.c -- minimal frame-loop driver around the GPU path, accepting the flags the
 * BASELINE configs use (reference ndt.c:1450: -d -r -f -l -s -t -u -o).  It loads a scene
 * program exactly like the reference does (dlopen + dlsym scene_setup / scene_frames /
 * scene_cleanup, ndt.c:1654-1664), renders each frame with ndt_render_image and writes
 * images/<scene>/<N>d/<WxH>/<scene>_<WxH>_<frame>.ppm (binary PPM of the pixel_d2c bytes; PNG
 * / JPEG encoding is outside this repository's scope).  `--dump-scene F` writes the flattened
 * scene of the last frame as an ndtscene file instead of rendering. */
#include <dlfcn.h>
#include <getopt.h>
#include <pthread.h>
#include <sys/stat.h>
#include <sys/time.h>

#include "ndt_host_internal.h"

static double now_s(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + 1e-6 * tv.tv_usec;
}

static int write_png(const char *path, const unsigned char *rgba8, int w, int h);

/* both writers take the bytes the reference stores: pixel_d2c of every channel (image.h:36-39, image.c:648-651) */
static int write_ppm(const char *path, const unsigned char *rgba8, int w, int h)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    unsigned char *row = (unsigned char *)malloc((size_t)w * 3);
    for (int j = 0; j < h; ++j) {
        for (int i = 0; i < w; ++i)
            for (int c = 0; c < 3; ++c) row[3 * i + c] = rgba8[((size_t)j * w + i) * 4 + c];
        fwrite(row, 1, (size_t)w * 3, f);
    }
    free(row);
    fclose(f);
    return 0;
}

/* pixel_d2c on the host: only for images that are in doubles here anyway (--raw, the normalised depth map) */
static unsigned char *quantise(const double *rgba, long n_values)
{
    unsigned char *out = (unsigned char *)malloc((size_t)n_values);
    for (long i = 0; i < n_values; ++i) {
        double d = rgba[i];
        double m = (1.0 < d) ? 1.0 : d;
        m = (0.0 > m) ? 0.0 : m;
        out[i] = (unsigned char)(sqrt(m) * 255);
    }
    return out;
}


/* ---- one frame: flatten, upload, render, save (what follows scene_setup in the reference's frame loop) */
static struct {
    int dims, width, height, depth, threads, aa_diff, aa_depth, stereo, specular, want_depth, samples, png, gpus;
    const char *raw_path;
} job_opts;

static int render_frame(scene *scn, int i)
{
    const int width = job_opts.width, height = job_opts.height;
    /* the image in doubles only when something asks for it (--raw, the depth map of -z); otherwise the GPU quantises
     * and 4 bytes per pixel come back instead of 32 */
    const int want_f64 = job_opts.raw_path != NULL || job_opts.want_depth;
    double *rgba = want_f64 ? (double *)malloc((size_t)width * height * 4 * sizeof(double)) : NULL;
    double *depth_map = job_opts.want_depth ? (double *)malloc((size_t)width * height * sizeof(double)) : NULL;
    unsigned char *rgba8 = NULL;
    double t0 = now_s();
    int ok;
    if (want_f64) {
        ok = ndt_render_image_full(scn, width, height, job_opts.samples, job_opts.threads, job_opts.aa_diff, job_opts.aa_depth,
                                   job_opts.stereo, job_opts.specular, job_opts.depth, rgba, depth_map);
        if (ok) rgba8 = quantise(rgba, (long)width * height * 4);
    } else {
        rgba8 = (unsigned char *)malloc((size_t)width * height * 4);
        ok = ndt_render_image_rgba8(scn, width, height, job_opts.samples, job_opts.threads, job_opts.aa_diff, job_opts.aa_depth,
                                    job_opts.stereo, job_opts.specular, job_opts.depth, rgba8);
    }
    if (!ok) {
        free(rgba);
        free(rgba8);
        free(depth_map);
        return 0;
    }
    printf("rendering took %.3fs\n", now_s() - t0);
    char dir[512], path[1024];
    mkdir("images", 0700);
    snprintf(dir, sizeof(dir), "images/%s", scn->name); mkdir(dir, 0700);
    snprintf(dir, sizeof(dir), "images/%s/%id", scn->name, job_opts.dims); mkdir(dir, 0700);
    snprintf(dir, sizeof(dir), "images/%s/%id/%ix%i", scn->name, job_opts.dims, width, height); mkdir(dir, 0700);
    if (job_opts.png) {
        snprintf(path, sizeof(path), "%s/%s_%ix%i_%04i.png", dir, scn->name, width, height, i);
        write_png(path, rgba8, width, height);
    } else {
        snprintf(path, sizeof(path), "%s/%s_%ix%i_%04i.ppm", dir, scn->name, width, height, i);
        write_ppm(path, rgba8, width, height);
    }
    printf("\tsaved %s\n", path);
    if (depth_map) {
        /* dbl_image_normalize (image.c:1025-1065) stretches the map to 0..1 before it is saved (ndt.c:1010-1016) */
        double lo = depth_map[0], hi = depth_map[0];
        for (long k = 0; k < (long)width * height; ++k) {
            if (depth_map[k] < lo) lo = depth_map[k];
            if (depth_map[k] > hi) hi = depth_map[k];
        }
        double *norm = (double *)malloc((size_t)width * height * 4 * sizeof(double));
        for (long k = 0; k < (long)width * height; ++k) {
            const double v = hi > lo ? (depth_map[k] - lo) / (hi - lo) : 0.0;
            norm[4 * k] = norm[4 * k + 1] = norm[4 * k + 2] = v;
            norm[4 * k + 3] = 1.0;
        }
        mkdir("depth", 0700);
        snprintf(path, sizeof(path), "depth/%s_%ix%i_%04i.ppm", scn->name, width, height, i);
        unsigned char *norm8 = quantise(norm, (long)width * height * 4);
        write_ppm(path, norm8, width, height);
        printf("\tsaved %s\n", path);
        free(norm8);
        free(norm);
        if (job_opts.raw_path) {
            char dpath[1100];
            snprintf(dpath, sizeof(dpath), "%s.depth", job_opts.raw_path);
            FILE *f = fopen(dpath, "wb");
            if (f) { fwrite(depth_map, sizeof(double), (size_t)width * height, f); fclose(f); }
        }
    }
    if (job_opts.raw_path) {
        FILE *f = fopen(job_opts.raw_path, "wb");
        if (f) { fwrite(rgba, sizeof(double), (size_t)width * height * 4, f); fclose(f); }
    }
    free(rgba);
    free(rgba8);
    free(depth_map);
    return 1;
}

/* ---- bounded queue of frames between the scene program (producer) and the render workers */
typedef struct { scene *scn; int frame; } frame_job;
static frame_job *queue = NULL;
static int queue_cap = 0, queue_n = 0, queue_head = 0, queue_done = 0, worker_failed = 0;
static pthread_mutex_t queue_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t queue_not_full = PTHREAD_COND_INITIALIZER, queue_not_empty = PTHREAD_COND_INITIALIZER;

static void queue_push(scene *scn, int frame)
{
    pthread_mutex_lock(&queue_mu);
    while (queue_n == queue_cap) pthread_cond_wait(&queue_not_full, &queue_mu);
    queue[(queue_head + queue_n) % queue_cap].scn = scn;
    queue[(queue_head + queue_n) % queue_cap].frame = frame;
    ++queue_n;
    pthread_cond_signal(&queue_not_empty);
    pthread_mutex_unlock(&queue_mu);
}

static void queue_close(void)
{
    pthread_mutex_lock(&queue_mu);
    queue_done = 1;
    pthread_cond_broadcast(&queue_not_empty);
    pthread_mutex_unlock(&queue_mu);
}

static void *worker_main(void *arg)
{
    /* worker w renders its frames on device w mod device count: one frame per GPU, the reference's MPI_MODE_FRAME
     * (ndt.c:1770-1830); with -g n every frame is spread over n contexts instead */
    const int w = (int)(long)arg;
    const int n_dev = ndt_hip_device_count();
    if (job_opts.gpus > 1) ndt_render_use_devices(job_opts.gpus);
    else ndt_render_use_device(n_dev > 0 ? w % n_dev : 0);
    for (;;) {
        pthread_mutex_lock(&queue_mu);
        while (queue_n == 0 && !queue_done) pthread_cond_wait(&queue_not_empty, &queue_mu);
        if (queue_n == 0) {
            pthread_mutex_unlock(&queue_mu);
            return NULL;
        }
        frame_job j = queue[queue_head];
        queue_head = (queue_head + 1) % queue_cap;
        --queue_n;
        pthread_cond_signal(&queue_not_full);
        pthread_mutex_unlock(&queue_mu);
        if (!render_frame(j.scn, j.frame)) worker_failed = 1;
        scene_free(j.scn);
        free(j.scn);
    }
}

/* scenes/yaml.c:14-50 */
static int yaml_scene_frames(int dimensions, char *config)
{
    if (config == NULL || dimensions < 3) return 0;
    return scene_yaml_count_frames(config);
}

static int yaml_scene_setup(scene *scn, int dimensions, int frame, int frames, char *config)
{
    (void)frames;
    scene_init(scn, "nameless", dimensions);
    if (config == NULL) {
        fprintf(stderr, "YAML scene requires a filename, use `-u filename`.\n");
        exit(1);
    }
    scene_read_yaml(scn, config, frame);
    return 0;
}

/* ---- PNG, 8-bit RGBA like the reference's writer (image.c:563-660: PNG_COLOR_TYPE_RGB_ALPHA, pixel_d2c on every
 * channel), without libpng: the zlib stream inside IDAT uses stored (uncompressed) deflate blocks */
static unsigned int crc_table[256];
static void crc_init(void)
{
    for (unsigned int n = 0; n < 256; ++n) {
        unsigned int c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
}
static unsigned int crc_update(unsigned int c, const unsigned char *buf, size_t len)
{
    for (size_t i = 0; i < len; ++i) c = crc_table[(c ^ buf[i]) & 0xff] ^ (c >> 8);
    return c;
}
static void put_be32(unsigned char *p, unsigned int v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
static void png_chunk(FILE *f, const char *type, const unsigned char *data, size_t len)
{
    unsigned char hdr[8];
    put_be32(hdr, (unsigned int)len);
    memcpy(hdr + 4, type, 4);
    fwrite(hdr, 1, 8, f);
    if (len) fwrite(data, 1, len, f);
    unsigned int c = crc_update(0xffffffffu, hdr + 4, 4);
    c = crc_update(c, data, len) ^ 0xffffffffu;
    unsigned char tail[4];
    put_be32(tail, c);
    fwrite(tail, 1, 4, f);
}

static int write_png(const char *path, const unsigned char *rgba8, int w, int h)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    crc_init();
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    fwrite(sig, 1, 8, f);
    unsigned char ihdr[13];
    put_be32(ihdr, (unsigned int)w);
    put_be32(ihdr + 4, (unsigned int)h);
    ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;     /* 8 bit, RGBA, no interlace */
    png_chunk(f, "IHDR", ihdr, 13);
    /* raw image: one filter byte (0 = none) + 4 bytes per pixel, per row */
    const size_t row = 1 + (size_t)w * 4, raw_len = row * (size_t)h;
    unsigned char *raw = (unsigned char *)malloc(raw_len);
    for (int j = 0; j < h; ++j) {
        unsigned char *q = raw + (size_t)j * row;
        *q++ = 0;
        memcpy(q, rgba8 + (size_t)j * w * 4, (size_t)w * 4);
    }
    /* zlib: header, stored blocks of at most 65535 bytes, adler32 */
    const size_t n_blocks = (raw_len + 65534) / 65535;
    const size_t z_len = 2 + raw_len + 5 * (n_blocks ? n_blocks : 1) + 4;
    unsigned char *z = (unsigned char *)malloc(z_len), *zp = z;
    *zp++ = 0x78; *zp++ = 0x01;
    unsigned int a = 1, b = 0;
    size_t off = 0;
    do {
        size_t n = raw_len - off > 65535 ? 65535 : raw_len - off;
        *zp++ = (off + n == raw_len) ? 1 : 0;
        *zp++ = n & 0xff; *zp++ = (n >> 8) & 0xff; *zp++ = ~n & 0xff; *zp++ = (~n >> 8) & 0xff;
        memcpy(zp, raw + off, n);
        for (size_t i = 0; i < n; ++i) {        /* adler32; the modulo can wait 5552 bytes, but this is not the hot path */
            a = (a + raw[off + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        zp += n;
        off += n;
    } while (off < raw_len);
    put_be32(zp, (b << 16) | a);
    zp += 4;
    png_chunk(f, "IDAT", z, (size_t)(zp - z));
    png_chunk(f, "IEND", NULL, 0);
    free(z);
    free(raw);
    fclose(f);
    return 0;
}

int main(int argc, char **argv)
{
    int dims = 3, width = 1920, height = 1080, first = 0, last = -1, frames = 300, frames_given = 0;
    int depth = 128, threads = 1;
    int aa_diff = 20, aa_depth = -1;        /* -a: recursive anti-aliasing off unless given (ndt.c:1411-1412, 1453) */
    int jobs = 1;           /* -j: frames in flight, worker w on GPU w mod device count (MPI_MODE_FRAME, ndt.c:1770-1830) */
    int gpus = 1;           /* -g: contexts ONE frame is spread over, context k on GPU k mod device count (MPI_MODE_ROW, ndt.c:812-820) */
    int png = 0;            /* --png: 8-bit RGBA PNG like the reference's default IMAGE_FORMAT (image.h:56-64) instead of PPM */
    int samples = 1;        /* -n (ndt.c:1574-1577) */
    int stereo = 0, specular = 1, want_depth = 0;      /* -m, -p, -z (ndt.c:1533-1573, 1581-1589, 1726-1729) */
    char *scene_path = NULL, *config = NULL, *dump_path = NULL, *raw_path = NULL;
    char *objects_dir = "objects";      /* -o: where object plugins are looked for (object.c:119; ndt.c passes "objects") */
    static struct option longopts[] = { { "dump-scene", required_argument, NULL, 1000 },
                                        { "raw", required_argument, NULL, 1001 }, { "png", no_argument, NULL, 1002 },
                                        { NULL, 0, NULL, 0 } };
    int ch;
    while ((ch = getopt_long(argc, argv, "a:d:g:r:f:j:l:m:3:n:ps:t:u:o:zh", longopts, NULL)) != -1) {
        int a1, a2, a3, n;
        switch (ch) {
        case 'a':       /* -a diff,depth (ndt.c:1453-1465); defaults 20,4 */
            aa_depth = 4;
            n = sscanf(optarg, "%d,%d", &a1, &a2);
            if (n >= 1) aa_diff = a1;
            if (n >= 2) aa_depth = a2;
            printf("anti-aliasing = diff=%i,depth=%i\n", aa_diff, aa_depth);
            break;
        case 'd': dims = atoi(optarg); break;
        case 'r':
            if (!strcmp(optarg, "4k")) { width = 3840; height = 2160; }
            else if (!strcmp(optarg, "1080p")) { width = 1920; height = 1080; }
            else if (!strcmp(optarg, "720p")) { width = 1280; height = 720; }
            else if (!strcmp(optarg, "480p")) { width = 720; height = 480; }
            else sscanf(optarg, "%dx%d", &width, &height);
            break;
        case 'f':       /* first:last:total, first:last, or last (ndt.c:1510-1524) */
            n = sscanf(optarg, "%d:%d:%d", &a1, &a2, &a3);
            if (n >= 3) { first = a1; last = a2; frames = a3; frames_given = 1; }
            else if (n >= 2) { first = a1; last = a2; }
            else if (n >= 1) { last = a1; }
            break;
        case 'g': gpus = atoi(optarg); break;
        case 'j': jobs = atoi(optarg); break;
        case 'l': depth = atoi(optarg); break;
        case 'm':
        case '3':       /* s(ide by side) / o(ver-under) / a(naglyph) / m(ono), ndt.c:1538-1572 */
            switch (optarg[0]) {
            case 'S': case 's': stereo = 1; printf("stereo = SIDE_SIDE_3D\n"); break;
            case 'O': case 'o': stereo = 2; printf("stereo = OVER_UNDER_3D\n"); break;
            case 'A': case 'a': stereo = 3; printf("stereo = ANAGLYPH_3D\n"); break;
            case 'H': case 'h': stereo = 4; width = 1920; height = 2205; printf("stereo = HIDEF_3D\n"); break;     /* ndt.c:1557-1565 */
            default: stereo = 0; printf("stereo = MONO\n"); break;
            }
            break;
        case 'n': samples = atoi(optarg); printf("samples = %i\n", samples); break;
        case 'p': specular = 0; printf("disabling specular highlights.\n"); break;
        case 'z': want_depth = 1; printf("record_depth_map = yes\n"); break;
        case 's': scene_path = optarg; break;
        case 't': threads = atoi(optarg); break;
        case 'u': config = optarg; break;
        case 'o': objects_dir = optarg; break;    /* object.c:119: a directory of object plugins (the built-in types need none) */
        case 1000: dump_path = optarg; break;
        case 1001: raw_path = optarg; break;
        case 1002: png = 1; break;
        default:
            fprintf(stderr, "usage: %s -s scene.so|builtin:yaml [-d dims] [-r WxH|1080p|4k] [-f last|first:last[:total]] [-l depth]\n"
                            "          [-a diff,depth] [-n samples] [-m s|o|a|m] [-p] [-z] [-j frames_in_flight] [-g gpus_per_frame] [-u config] [--dump-scene file.ndtscene] [--raw file.f64] [--png]\n", argv[0]);
            return ch == 'h' ? 0 : 1;
        }
    }
    if (!scene_path || dims < 3 || width < 1 || height < 1) {
        fprintf(stderr, "%s: need -s scene.so, dims >= 3 and a resolution\n", argv[0]);
        return 1;
    }
    int (*setup)(scene *, int, int, int, char *) = NULL;
    int (*frame_count)(int, char *) = NULL;
    int (*cleanup)(void) = NULL;
    if (!strcmp(scene_path, "builtin:yaml")) {
        /* the reference's scenes/yaml.c, built in: -u names the YAML file, one document per frame */
        setup = yaml_scene_setup;
        frame_count = yaml_scene_frames;
    } else {
        void *dl = dlopen(scene_path, RTLD_NOW);
        if (!dl) { fprintf(stderr, "%s\n", dlerror()); return 1; }
        *(void **)(&setup) = dlsym(dl, "scene_setup");
        *(void **)(&frame_count) = dlsym(dl, "scene_frames");
        *(void **)(&cleanup) = dlsym(dl, "scene_cleanup");
    }
    if (!setup) { fprintf(stderr, "%s has no scene_setup\n", scene_path); return 1; }
    if (frame_count && !frames_given) frames = frame_count(dims, config);
    if (last < 0) last = frames - 1;
    register_objects(objects_dir);

    job_opts.dims = dims; job_opts.width = width; job_opts.height = height; job_opts.depth = depth; job_opts.threads = threads;
    job_opts.aa_diff = aa_diff; job_opts.aa_depth = aa_depth; job_opts.stereo = stereo; job_opts.specular = specular;
    job_opts.want_depth = want_depth; job_opts.raw_path = raw_path; job_opts.samples = samples; job_opts.png = png;
    job_opts.gpus = gpus > 1 ? gpus : 1;
    if (job_opts.gpus > 1) {
        printf("one frame over %d GPU contexts (%d device(s) visible)\n", job_opts.gpus, ndt_hip_device_count());
        ndt_render_use_devices(job_opts.gpus);
    }
    /* -j K: K frames in flight.  The scene program runs on this thread, frame after frame (it may
     * keep state between frames, ndt.c:1818-1825); everything after it -- bounding spheres, kd-tree,
     * upload, render, image files -- happens on K worker threads, each with its own GPU context.
     * This is the reference's MPI_MODE_FRAME (rank 0 builds the scenes, the others render, ndt.c:1770-1830)
     * inside one process. */
    pthread_t *workers = NULL;
    if (jobs > 1 && !dump_path) {
        workers = (pthread_t *)calloc((size_t)jobs, sizeof(pthread_t));
        queue_cap = jobs;
        queue = (frame_job *)calloc((size_t)queue_cap, sizeof(frame_job));
        for (int k = 0; k < jobs; ++k) pthread_create(&workers[k], NULL, worker_main, (void *)(long)k);
    }
    double t_all = now_s();
    int rendered = 0;
    for (int i = 0; i < frames && i <= last; ++i) {
        scene *scn = (scene *)calloc(1, sizeof(scene));
        setup(scn, dims, i, frames, config);
        if (i < first) {            /* earlier frames still run scene_setup (ndt.c:1818-1825) */
            scene_free(scn);
            free(scn);
            continue;
        }
        printf("Scene has %i objects and %i lights\n", scn->num_objects, scn->num_lights);
        if (dump_path) {
            char err[256];
            ndt_flat_builder fb;
            if (ndt_flatten_scene_mt(scn, &fb, err, sizeof(err), threads) != 0) { fprintf(stderr, "%s\n", err); return 1; }
            if (i == last || i == frames - 1) ndt_write_ndtscene(&fb.fs, scn->name, dump_path);
            ndt_flat_builder_free(&fb);
            scene_free(scn);
            free(scn);
            continue;
        }
        ++rendered;
        if (workers) {
            queue_push(scn, i);
        } else {
            if (!render_frame(scn, i)) return 1;
            scene_free(scn);
            free(scn);
        }
    }
    if (workers) {
        queue_close();
        for (int k = 0; k < jobs; ++k) pthread_join(workers[k], NULL);
        free(workers);
        if (worker_failed) return 1;
    }
    if (rendered > 1) printf("%d frames in %.3fs (%.2f frames/s)\n", rendered, now_s() - t_all, rendered / (now_s() - t_all));
    if (cleanup) cleanup();
    return 0;
}
