/*
 * ndt_hip.h -- C ABI of libndt_hip.so, the MI355X (gfx950) ray-trace core for ndt.
 *
 * This library replaces exactly one thing in the reference: the body of
 *     int render_image(scene *scn, char *name, ..., int width, int height, int samples,
 *                      stereo_mode mode, int threads, ..., int max_optic_depth, ...)
 * (reference ndt.c:900), i.e. the loop nest render_lines_thread (ndt.c:803) ->
 * render_line (ndt.c:735) -> render_pixel (ndt.c:578) -> get_pixel_color (ndt.c:456) ->
 * get_ray_color (ndt.c:329) / apply_lights (ndt.c:71) -> trace_kd (object.c:683) ->
 * kd_tree_intersect (kd-tree.c:570) -> trace (object.c:692) -> obj->intersect (objects/ *.c).
 *
 * The boundary is plain C: pointers, sizes, ints and doubles.  A scene crosses it as an
 * `ndt_flat_scene`: the reference's pointer-rich `scene` / `object` / `kd_tree_t` graph
 * flattened into index-linked arrays (INTEGRATION.md shows the ~150-line flattening stub a
 * maintainer adds next to render_image).  All reals are IEEE double, as in the reference.
 *
 * Error convention: every entry point returns 0 on success and a negative NDT_E_* code on
 * failure; ndt_hip_last_error() gives the message.  The reference's exit(1)-on-error habit
 * (object.c:233-236) is deliberately not carried into the library.
 */
#ifndef NDT_HIP_H
#define NDT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDT_HIP_ABI_VERSION 3

/* Dimensions the gfx950 kernels are instantiated for.  The reference accepts any N >= 3
 * (ndt.c:1450 `-d`); BASELINE.json's configs span 3..8. */
#define NDT_MIN_DIMS 3
#define NDT_MAX_DIMS 12
#define NDT_MAX_LIGHTS 64

#define NDT_OK              0
#define NDT_E_INVALID      -1   /* malformed scene / argument */
#define NDT_E_UNSUPPORTED  -2   /* valid for the reference, but not for the device path */
#define NDT_E_DEVICE       -3   /* HIP runtime failure */
#define NDT_E_NOMEM        -4
#define NDT_E_STATE        -5   /* call out of order (no scene uploaded, ...) */

/* Object kinds = the reference's built-in plugins (objects/ *.c `type_name`). */
enum ndt_object_type {
    NDT_OBJ_SPHERE    = 0,  /* objects/sphere.c    */
    NDT_OBJ_HPLANE    = 1,  /* objects/hplane.c    */
    NDT_OBJ_HDISK     = 2,  /* objects/hdisk.c     */
    NDT_OBJ_CYLINDER  = 3,  /* objects/cylinder.c  */
    NDT_OBJ_HCYLINDER = 4,  /* objects/hcylinder.c */
    NDT_OBJ_ORTHOTOPE = 5,  /* objects/orthotope.c */
    NDT_OBJ_HCUBE     = 6,  /* objects/hcube.c     */
    NDT_OBJ_HFACET    = 7,  /* objects/hfacet.c    */
    NDT_OBJ_FACET     = 8,  /* objects/facet.c     */
    NDT_OBJ_TYPE_COUNT = 9
};

/* Same numbering as the reference's light_type enum (scene.h:23-31). */
enum ndt_light_type {
    NDT_LIGHT_AMBIENT     = 0,
    NDT_LIGHT_POINT       = 1,
    NDT_LIGHT_DIRECTIONAL = 2,
    NDT_LIGHT_SPOT        = 3,
    NDT_LIGHT_DISK        = 4,  /* area lights: a random point of the disk / rectangle per shading evaluation */
    NDT_LIGHT_RECT        = 5   /* (ndt.c:116-147); rendered by the sampled path, parity is statistical */
};

/* One light (reference `light`, scene.h:36-49).  pos/dir are offsets, in doubles, into
 * ndt_flat_scene.vecs; -1 when the light has none. */
typedef struct ndt_flat_light {
    int32_t type;
    int32_t pos_off;
    int32_t dir_off;
    int32_t area_off;           /* ABI 3, DISK / RECT: u1[dims] then v1[dims] (scene_prepare_light, scene.c:182-195), else -1 */
    double  red, green, blue;
    double  angle;              /* spot cone half-angle, degrees (ndt.c:204) */
    double  radius;             /* ABI 3, DISK / RECT: the sample is pos + x*radius*u1 + y*radius*v1 (ndt.c:126-141) */
} ndt_flat_light;

/* One object (reference `object`, object.h:23-74) with its API-level parameters as the scene
 * wrote them (object_add_pos/dir/size/flag).  Ray-invariant data the plugins derive lazily in
 * their prepare() is recomputed inside the library.
 *
 * Objects [0, n_items) are the kd items in kd id order (kd-tree.c:448); the per-ray visit
 * mask (kd-tree.c:600) is indexed by that position.  Objects >= n_items are nested
 * primitives owned by a composite (`parent` >= 0): the orthotope faces an hcube builds in
 * add_faces (objects/hcube.c:34).  Clusters never appear: object_kdlist_add flattens them
 * (object.c:636-643). */
typedef struct ndt_flat_object {
    int32_t type;               /* enum ndt_object_type */
    int32_t transparent;        /* object.h:24 */
    int32_t parent;             /* -1 for kd items */
    int32_t n_pos,  pos_off;    /* pos[i] = vecs + pos_off + i*dims */
    int32_t n_dir,  dir_off;
    int32_t n_size, size_off;   /* into sizes[] */
    int32_t n_flag, flag_off;   /* into flags[] */
    int32_t n_obj,  obj_off;    /* children: obj_refs[obj_off .. obj_off+n_obj) are object indices */
    int32_t bounds_center_off;  /* into vecs[] */
    double  bounds_radius;      /* >0: gate with bounding sphere; <=0: no gate (object.c:618-624) */
    double  red, green, blue;
    double  red_r, green_r, blue_r;
    double  refract_index;
} ndt_flat_object;

/* One kd-tree node (reference kd_node_t, kd-tree.h:52-59).  Interior: dim >= 0, left/right
 * are node indices.  Leaf: dim < 0, items are leaf_refs[first .. first+num). */
typedef struct ndt_flat_kdnode {
    int32_t dim;
    int32_t num;
    int32_t left, right;
    int32_t first;
    int32_t _pad;
    double  boundary;
} ndt_flat_kdnode;

typedef struct ndt_flat_scene {
    int32_t abi_version;        /* NDT_HIP_ABI_VERSION */
    int32_t dims;

    /* pools */
    const double  *vecs;     int64_t n_vecs;      /* doubles */
    const double  *sizes;    int64_t n_sizes;
    const int32_t *flags;    int64_t n_flags;
    const int32_t *obj_refs; int64_t n_obj_refs;

    /* camera after camera_aim (camera.c:132), before render_image's dirX *= W/H (ndt.c:926);
     * offsets into vecs.  cam_type: 0 CAMERA_NORMAL (camera.c:557-575), 1 CAMERA_VR, 2 CAMERA_PANO (camera.c:506-555; they
     * need the ABI 2 fields below: fields of view, local axes). */
    int32_t cam_type;
    int32_t cam_pos_off, cam_img_orig_off, cam_dir_x_off, cam_dir_y_off;
    double  cam_focal_distance;

    /* scene colours: scn->ambient (scene.h:59) and bg_* (scene.h:60) */
    double ambient[3];
    double background[4];

    const ndt_flat_light  *lights;  int32_t n_lights;
    const ndt_flat_object *objects; int32_t n_objects; int32_t n_items;

    /* kd-tree (kd-tree.h:63-71): node 0 is the root */
    const ndt_flat_kdnode *kd_nodes;  int32_t n_kd_nodes;
    const int32_t         *leaf_refs; int32_t n_leaf_refs;   /* object indices (< n_items) */
    const int32_t         *inf_refs;  int32_t n_inf;         /* infinite objects, kd-tree.c:459 */
    int32_t bb_lower_off, bb_upper_off;                      /* root AABB, into vecs */

    /* ABI 2 -- the rest of the camera (camera.h:34-76), used by recursive anti-aliasing, stereo
     * modes and the VR / panorama cameras; offsets into vecs, -1 when the producer has none.
     * cam_aperture_radius != 0 makes recursive AA and samples>1 sample the lens (ndt.c:528-542: the reference's global
     * drand48 stream; here counter-based streams): such frames are stochastic renders, parity is statistical. */
    double  cam_aperture_radius;
    double  cam_h_fov, cam_v_fov;                            /* camera.h:52-53 */
    int32_t cam_left_eye_off, cam_right_eye_off;             /* camera.h:60-61 */
    int32_t cam_local_x_off, cam_local_y_off, cam_local_z_off;   /* camera.h:69-71 */
    int32_t _pad2;
} ndt_flat_scene;

/* Arguments of one render_image call (ndt.c:900).  Rows are dealt cyclically exactly like the
 * reference's threads/MPI_ROW split (ndt.c:812-820): this call renders image rows
 * j = row_begin, row_begin+row_step, ... < height, and output row k holds image row
 * row_begin + k*row_step.  row_begin=0,row_step=1 renders the whole frame. */
typedef struct ndt_render_params {
    int32_t width, height;
    int32_t max_optic_depth;    /* `-l`, default 128 (ndt.c:1413) */
    int32_t samples;            /* `-n`.  1 = the deterministic path.  > 1 = jittered samples + lens
                                 * sampling + the adaptive loop (ndt.c:470-568) with a per-(pixel, sample) counter-based
                                 * random stream: reproducible, independent of sharding, statistically equivalent to
                                 * the reference's drand48 stream; in every stereo mode, through every camera, with or
                                 * without a depth map (a pixel's depth is then its LAST sample's, ndt.c:362-373) */
    int32_t row_begin, row_step;
    int32_t specular;           /* 1 = specular_enabled (ndt.c:41) */
    int32_t profile;            /* 1 = bracket trace kernels with hipEvents (fills *_ms below) */
    /* ABI 2.  Whitted's recursive anti-aliasing (`-a diff,depth`; ndt.c:655-733, 1039-1087): the
     * first pass renders (width+1) x (height+1) corner samples, every output pixel is the average
     * of its four corners and is subdivided (5 new samples per level, up to aa_depth+1 levels)
     * while the corners differ by more than aa_diff/255.  The output is the resampled image
     * (width x height doubles; the reference quantises it to 8 bits when it stores it). */
    int32_t recursive_aa;       /* 0 = off */
    int32_t aa_diff;            /* reference default 20 (ndt.c:1412) */
    int32_t aa_depth;           /* reference default 4 (ndt.c:1411) */
    int32_t stereo;             /* ndt_stereo_mode (needs the eyes in the flat scene); with recursive_aa: every mode but HIDEF */
    int32_t reserved[4];        /* must be 0 */
} ndt_render_params;

typedef struct ndt_render_stats {
    /* rays actually traced on the device: one per unique trace_kd query */
    int64_t rays_primary, rays_secondary, rays_shadow;
    /* rays the reference executes for the same pixels: every pixel's ray tree times the
     * adaptive loop's repeat count k (ndt.c:488) */
    int64_t rays_ref_equiv;
    int32_t levels;             /* ray-tree depth reached */
    int32_t trace_launches;     /* launches of the trace kernel */
    double  trace_ms;           /* summed device time of those launches (profile=1) */
    double  frame_ms;           /* device time of the whole call (profile=1) */
    int64_t node_capacity;      /* ray-tree nodes the workspace holds */
    /* ABI 2 */
    int64_t pixels_resampled;   /* recursive AA: pixels that were subdivided (the reference's "pixels resampled") */
    int64_t aa_samples;         /* recursive AA: extra get_pixel_color samples rendered by the second pass;
                                 * stochastic renders (-n > 1, area lights): samples the adaptive loop consumed
                                 * (rays_* then also count the samples rendered ahead and dropped) */
} ndt_render_stats;

/* stereo_mode (ndt.c:46-48).  SIDE_SIDE / OVER_UNDER put the left-eye image in the left / top half and
 * the right-eye image in the other (each squeezed to half size), ANAGLYPH mixes the luminance of the
 * two eyes into red and blue (ndt.c:590-650).  HIDEF packs 1080 lines of the left eye, 45 black lines and 1080
 * lines of the right eye (ndt.c:614-631; `-m h` sets 1920x2205); the reference leaves the alpha of the black
 * lines unset, this library writes 1. */
enum ndt_stereo_mode {
    NDT_STEREO_MONO = 0, NDT_STEREO_SIDE_SIDE = 1, NDT_STEREO_OVER_UNDER = 2, NDT_STEREO_ANAGLYPH = 3, NDT_STEREO_HIDEF = 4
};

typedef struct ndt_hip_ctx ndt_hip_ctx;

/* Create a context on HIP device `device` with its own stream.  Fails with NDT_E_DEVICE when
 * no gfx950 device is usable -- there is no CPU fallback in this library. */
int ndt_hip_create(int device, ndt_hip_ctx **out);
int ndt_hip_destroy(ndt_hip_ctx *ctx);

/* Validate the scene, derive the plugins' prepare() data, lay it out for the device and copy
 * it to HBM.  Replaces the per-frame state render_image reads: `scn` and the global `kdtree`
 * (ndt.c:68). */
int ndt_hip_upload_scene(ndt_hip_ctx *ctx, const ndt_flat_scene *scene);

/* render_image (ndt.c:900) for the rows selected by `p`.  `rgba` receives
 * rows*width*4 doubles laid out like the reference's dbl image (image.c:126: r,g,b,a per
 * pixel, row-major).  _device: `rgba` is a device pointer on ctx's device (stays in HBM);
 * plain: `rgba` is host memory.  Both return when the frame is complete: every kernel of it has
 * finished on the context's stream (the last one posts a tag to host-mapped memory the call waits for),
 * so the image may be read from any stream or copied out at once. */
int ndt_hip_render_device(ndt_hip_ctx *ctx, const ndt_render_params *p, void *d_rgba, ndt_render_stats *stats);
int ndt_hip_render(ndt_hip_ctx *ctx, const ndt_render_params *p, double *rgba, ndt_render_stats *stats);

/* The same with the depth map render_image fills when it is given a depth file name (ndt.c:930-935,
 * 753-756): `depth` receives rows*width doubles, 1/distance of the primary hit (the left eye's for
 * ANAGLYPH), 0 where the primary ray misses.  With recursive_aa: the first pass's depths (ndt.c:930-935) -- but not beside
 * a stochastic anti-aliased render (a lens, area lights or samples > 1 under recursive_aa).  (The reference leaves the
 * previous pixel's value when the hit lies within EPSILON of the eye; this library writes 0.) */
int ndt_hip_render_depth_device(ndt_hip_ctx *ctx, const ndt_render_params *p, void *d_rgba, void *d_depth, ndt_render_stats *stats);
int ndt_hip_render_depth(ndt_hip_ctx *ctx, const ndt_render_params *p, double *rgba, double *depth, ndt_render_stats *stats);

/* Batch of trace_kd queries (object.c:683) against the uploaded scene -- the unit the
 * known-answer tests pin.  o, v: n*dims doubles (ray-major); dist_limit: n doubles with the
 * reference's meaning (<0 closest hit, 0 any hit, >0 first hit within limit; ndt.c:177-189).
 * Outputs (host): obj[n] = object index or -1; hit, normal: n*dims doubles, zero when obj<0. */
int ndt_hip_trace_rays(ndt_hip_ctx *ctx, int64_t n, const double *o, const double *v,
                       const double *dist_limit, int32_t *obj, double *hit, double *normal);

/* Quantise a double image like the reference does at save time: pixel_d2c (image.h:36-39),
 * (unsigned char)(sqrt(clamp01(x))*255) per channel.  d_rgba: device, n_pixels*4 doubles;
 * d_rgba8: device, n_pixels*4 bytes. */
int ndt_hip_quantize_device(ndt_hip_ctx *ctx, const void *d_rgba, void *d_rgba8, int64_t n_pixels);

/* What the calls that deliver a finished image write: the double framebuffer (4 doubles per pixel, image.c:126) or the
 * bytes the reference stores at save time (pixel_d2c on every channel, image.h:36-39, image.c:648-651: 4 bytes per pixel). */
enum ndt_image_format { NDT_IMAGE_F64 = 0, NDT_IMAGE_RGBA8 = 1 };

/* render_image followed by the reference's save-time quantisation, on the device: `rgba8` (host) receives
 * rows*width*4 bytes.  An eighth of ndt_hip_render's bytes cross PCIe. */
int ndt_hip_render_rgba8(ndt_hip_ctx *ctx, const ndt_render_params *p, uint8_t *rgba8, ndt_render_stats *stats);

/* The same without waiting for the bytes to arrive: the call returns when the frame is complete in HBM and quantised; its copy to
 * `rgba8` -- pinned host memory -- runs on a copy stream behind the NEXT call's rendering.  When a call returns, every EARLIER
 * frame of the context has arrived in its buffer; the frame of the call itself has when the next call, or
 * ndt_hip_render_rgba8_wait, returns -- until then `rgba8` must stay valid and is not to be read.  What "rendering took"
 * (ndt.c:978-984: pixels in host memory) costs per frame of a sequence is then the render. */
int ndt_hip_render_rgba8_async(ndt_hip_ctx *ctx, const ndt_render_params *p, uint8_t *rgba8, ndt_render_stats *stats);
int ndt_hip_render_rgba8_wait(ndt_hip_ctx *ctx);

/* ONE frame over several contexts -- one per GPU of the node, or several on one GPU -- called from one host thread.
 * The rows `p` selects are dealt cyclically to the contexts exactly as the reference deals rows to MPI ranks in
 * MPI_MODE_ROW (ndt.c:812-820: row_start = rank, row_step = size): context k renders rows
 * p->row_begin + (k + i*n_ctx)*p->row_step.  Every context must hold the same uploaded scene.  Each device pushes
 * its finished rows straight into the assembled image on ctxs[0]'s device (peer stores over xGMI; through a staging
 * copy where peer access is unavailable or option "multi_path" asks for it) -- the reference instead sum-reduces full-size zero-padded images up a
 * binary tree (mpi_collect_image, ndt.c:1277-1309).  `format` = ndt_image_format.  _device: `d_out` is memory of
 * ctxs[0]'s device; plain: `out` is host memory.  stats: ray counts summed over the contexts, times = the slowest's.
 * Not with a depth map.  Returns when the image is complete. */
int ndt_hip_render_multi_device(ndt_hip_ctx *const *ctxs, int32_t n_ctx, const ndt_render_params *p, int32_t format,
                                void *d_out, ndt_render_stats *stats);
int ndt_hip_render_multi(ndt_hip_ctx *const *ctxs, int32_t n_ctx, const ndt_render_params *p, int32_t format,
                         void *out, ndt_render_stats *stats);

/* How a context's rows reached the assembled image in the last ndt_hip_render_multi[_device] call it took part in, so
 * that a multi-GPU run can be diagnosed from its log (option "multi_path" chooses; `ndt_hip -g N` prints it):
 * LOCAL = the context lives on ctxs[0]'s device; PEER = its push kernel stored over xGMI into ctxs[0]'s HBM;
 * STAGED = one hipMemcpyPeerAsync into a staging buffer on ctxs[0]'s device, pushed from there; NONE = no rows. */
enum ndt_multi_path { NDT_MULTI_NONE = 0, NDT_MULTI_LOCAL = 1, NDT_MULTI_PEER = 2, NDT_MULTI_STAGED = 3 };
int ndt_hip_multi_path_taken(ndt_hip_ctx *ctx);

/* Switches of a context.  None changes an image; they choose between equivalent ways of producing it, or turn diagnostics
 * on.  The same names, upper-cased behind NDT_HIP_ (NDT_HIP_PIPELINE, NDT_HIP_DEBUG_LEVELS ...), are read from the
 * environment ONCE, when the context is created; nothing on the render or upload path looks at the environment.
 *   "pipeline"        0 auto, 1 levels (one trace launch + shade launches per bounce), 2 stream (the whole ray tree in
 *                     one persistent launch), 3 hybrid -- DESIGN.md section 3; the environment takes the words
 *   "hybrid_level"    hybrid: the bounce from which on the frame kernel renders (default 2)
 *   "stream_below"    auto: passes of up to this many primaries go to the streaming frame kernel
 *   "stream_below_list"  ... and passes over a list of samples (recursive anti-aliasing) of up to this many (default 30 000)
 *   "hull_box" / "face_box"   0: upload hcubes without the hull box / without the per-face boxes (tests prove them neutral)
 *   "face_tree"               0: hcubes of more than 63 faces without the hierarchy over their face boxes (ndt_hip_hcube_face_tree;
 *                                tests prove it neutral)
 *   "face_groups"             0: ... without the index of their faces by thin hull axes (ndt_hip_hcube_face_groups; neutral)
 *   "fuse_primaries"          per-bounce kernels: -1 auto (the first trace launch makes the primaries itself for a planar
 *                                camera from 4-D on), 0 always a k_primary launch, 1 never one (planar camera)
 *   "gate_prepass_below"      item-set scenes: passes of up to this many primaries mark the items whose bounding sphere the
 *                                ray's line misses as visited before the walk (neutral; default 400 000)
 *   "coop" (+ "coop_budget_us", "coop_max_live", "coop_tail_only", "coop_waves")   item-set scenes, per-bounce kernels: a batch
 *                                over its budget with few rays left gives them up to wavefronts that trace one ray each
 *                                (ndt_device.hpp:coop_trace) -- bit-identical, measured slower, default 0
 *   "stream_fused"            0: the frame kernel between a k_primary and a k_finish_pixels launch instead of making its
 *                                primaries and writing its pixels itself (tests prove it neutral)
 *   "item_sets"               0: upload a scene of up to 64 items with plain leaf lists (the kernels that read the lists,
 *                                as for larger scenes) instead of 64-bit item sets per leaf (tests prove it neutral)
 *   "item_boxes"              0: upload scenes of more than 256 items without the item boxes (ndt_hip_item_boxes)
 *   "leaf_scan" / "leaf_scan_group"   global-memory tier: the lanes of a wavefront that stand on the same kd leaf scan it
 *                                together through LDS (0: never) when at least that many of them do (default 64: all)
 *   "leaf_history"            scenes in the global-memory tier (more than 256 items): a ray remembers what it visited as up to
 *                                this many {leaf, cut} pairs (default and maximum 4) before it falls back to its bit mask in
 *                                the slab; 0: the slab only (tests prove every value neutral)
 *   "sample_seed"     stochastic renders (-n > 1, area lights): which set of counter-based random streams the samples draw
 *                     from (0, the default, and any other value give images that are independent draws of one distribution)
 *   "multi_path"      ndt_hip_render_multi: 0 auto (stores on the same device, peer stores over xGMI, a staged copy where
 *                     there is no peer access), 1 never staged, 2 always staged -- also between contexts of one device
 *   "shade_pair"      0: lighting of a bounce and shading of the next as two launches
 *   "debug_levels"    profiled renders print the bounces and the duration of every trace launch
 *   "exit_probe" / "shade_probe" / "stream_probe"   profiled renders log the life of every wavefront of the trace
 *                     launches / of the k-th shade launch (value k + 1) / of the frame kernel
 *   "test_small_pool" a fresh workspace starts with a node pool a reflective scene overflows (tests of the regrow path)
 * Returns NDT_E_INVALID for a name it does not know. */
int ndt_hip_set_option(ndt_hip_ctx *ctx, const char *name, int64_t value);

/* HIP devices this process sees (0 without a GPU), and the device a context lives on. */
int ndt_hip_device_count(void);
int ndt_hip_device(ndt_hip_ctx *ctx);

/* Number of rows a (row_begin,row_step) shard of a `height`-row image holds. */
int32_t ndt_hip_shard_rows(int32_t height, int32_t row_begin, int32_t row_step);

/* Diagnostic, host only (no GPU needed): the hull box the library derives for hcube `object`
 * of `scene` at upload time -- an oriented box that contains every point orthotope.intersect
 * (orthotope.c:150-300) can return for the hcube's faces; rays that miss it skip the nested
 * trace() over the faces (hcube.c:241), which cannot change the answer.  rows receives
 * dims x { unit axis[dims], centre coordinate, half extent }: the region |axis_k . x - centre_k| <= half_k for every k.  The
 * axes are unit covectors -- an orthonormal frame, or, when the hcube is a parallelotope, the normalised dual basis of its
 * edge directions (axis_k . d_j = 0 for j != k), in which every face is thin exactly on the directions it does not span;
 * they need not be orthogonal to each other.  Returns 1 when the hcube has a
 * box, 0 when it has none (its faces are always scanned), <0 on NDT_E_*. */
int ndt_hip_hcube_hull_box(const ndt_flat_scene *scene, int32_t object, double *rows);

/* Diagnostic, host only: the boxes of the single faces of hcube `object` in the frame of its hull box
 * (same derivation, one face at a time): the nested trace() visits only the faces whose box the ray
 * meets.  face_rows receives n_faces x dims x { centre coordinate, half extent } (room for 63 faces);
 * bit f of *possible is clear when face f can never be hit.  Returns the number of faces, 0 when the
 * hcube has no hull box, or more than 63 faces (then: ndt_hip_hcube_face_boxes_all), <0 on NDT_E_*. */
int ndt_hip_hcube_face_boxes(const ndt_flat_scene *scene, int32_t object, double *face_rows, uint64_t *possible);
/* ... for an hcube of any number of faces (a 6-D one nests 472, a 10-D one 52 904: the device takes their boxes 63 at a time):
 * face_rows: n_faces x dims x { centre, half extent }, possible: one byte per face.  Returns the number of faces; with
 * cap_faces too small (or null pointers) only that -- call again with room.  0: the hcube gets no boxes, <0 on NDT_E_*. */
int64_t ndt_hip_hcube_face_boxes_all(const ndt_flat_scene *scene, int32_t object, int64_t cap_faces, double *face_rows, uint8_t *possible);
/* Diagnostic, host only: the hierarchy the library lays over the face boxes of an hcube of more than 63 faces (option
 * "face_tree"): level j = 1 .. *top holds, for every aligned run [k 2^j, (k + 1) 2^j) of faces, the box of the union of the
 * boxes of its faces that can be hit (half extents of -1: none can); a ray that misses a run's box skips its faces.
 * rows: n_nodes x dims x { centre, half extent }, level j starting at node level_off[j] (level_off: room for 32 ints).
 * Returns the number of nodes; with cap_nodes too small (or rows null) only that and *top.  0: no boxes, <0 on NDT_E_*. */
int64_t ndt_hip_hcube_face_tree(const ndt_flat_scene *scene, int32_t object, int64_t cap_nodes, double *rows, int32_t *level_off, int32_t *top);

/* Diagnostic, host only: the index the library lays over the faces of an hcube of more than 63 faces in 5-D and up (option
 * "face_groups"): the faces by the set of hull axes their boxes are thin on.  clusters: dims x { centre-, half-, centre+, half+ }
 * -- per hull axis the two intervals (one per side of the hull's centre; half -1: none) that hold the thin intervals of all
 * faces thin on that axis; table: 2^dims x { start, count } -- for every subset S of the axes where in `members` (n_faces
 * ints, may be null) the faces thin exactly on S stand, ascending; face_set: per face its S as a bit mask (-1: the face can
 * never be hit).
 * A ray can meet a face's box only if, inside the hull box, it passes a cluster interval on every axis of the face's S; the
 * device looks up the subsets of the axes whose clusters the ray passes and tests their faces' own boxes.
 * Returns the number of faces (0: the hcube gets no boxes), <0 on NDT_E_*. */
int ndt_hip_hcube_face_groups(const ndt_flat_scene *scene, int32_t object, double *clusters, int32_t *table, int32_t *face_set, int32_t *members);

/* Diagnostic, host only: the item boxes the library derives at upload for scenes of more than 256 items -- one orthonormal
 * frame for the scene (frame: dims x unit axis[dims]) and, for every top-level orthotope, the box in that frame of every
 * point its intersect() can return (rows: n_items x dims x { centre coordinate, half extent }; has[i] != 0: item i carries
 * one).  A ray that misses an item's box skips its bounding-sphere gate and its intersect(); tests prove that neutral.
 * Returns the number of boxed items (0: none), <0 on NDT_E_*. */
int ndt_hip_item_boxes(const ndt_flat_scene *scene, double *frame, double *rows, uint8_t *has);

/* The stream the context launches on (a hipStream_t, created hipStreamNonBlocking: nothing the caller queues on the NULL
 * stream or any other is ordered against it implicitly), for callers that time with their own events or order other work
 * against it. */
void *ndt_hip_stream(ndt_hip_ctx *ctx);
int ndt_hip_synchronize(ndt_hip_ctx *ctx);

const char *ndt_hip_last_error(void);
int ndt_hip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NDT_HIP_H */
