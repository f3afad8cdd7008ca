// ndt_host.hip -- host side of libndt_hip.so: the C ABI of include/ndt_hip.h.
//
//   ndt_hip_upload_scene : validate an ndt_flat_scene, derive what the reference's plugins
//                          derive lazily in prepare() (objects/ *.c), lay everything out as
//                          one blob of 8-byte words and copy it to HBM.
//   ndt_hip_render*      : render_image (ndt.c:900) as a bounce-synchronous wavefront tracer.
//
// There is no CPU fallback here: without a usable HIP device every entry point fails with
// NDT_E_DEVICE.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <string>
#include <vector>

#include "../../include/ndt_hip.h"
#include "ndt_kernels.hpp"

// ------------------------------------------------------------------ errors

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(NDT_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

extern "C" const char *ndt_hip_last_error(void) { return g_err; }
extern "C" int ndt_hip_abi_version(void) { return NDT_HIP_ABI_VERSION; }
struct HullFaces {
    // per face of the hcube, in the hull box's frame: N x { centre coordinate, half extent } -- the face's own
    // box, same derivation and margin as the hull box; possible bit f clear = face f can never be hit
    std::vector<double> rows;
    unsigned long long possible = 0;
    int n_faces = 0;
};
static bool hcube_hull_box(const ndt_flat_scene *fs, const ndt_flat_object &o, int n, std::vector<double> &rows, HullFaces *faces = nullptr);

extern "C" int ndt_hip_hcube_hull_box(const ndt_flat_scene *fs, int32_t object, double *rows_out)
{
    if (!fs || !rows_out) return fail(NDT_E_INVALID, "null argument");
    if (fs->abi_version != NDT_HIP_ABI_VERSION) return fail(NDT_E_INVALID, "flat scene ABI %d, library %d", fs->abi_version, NDT_HIP_ABI_VERSION);
    if (fs->dims < 3 || fs->dims > NDT_MAX_DIMS) return fail(NDT_E_UNSUPPORTED, "dims %d", fs->dims);
    if (object < 0 || object >= fs->n_objects || fs->objects[object].type != NDT_OBJ_HCUBE)
        return fail(NDT_E_INVALID, "object %d is not an hcube", object);
    const ndt_flat_object &o = fs->objects[object];
    if (o.n_obj < 1 || o.obj_off < 0 || (int64_t)o.obj_off + o.n_obj > fs->n_obj_refs) return fail(NDT_E_INVALID, "object %d: child range", object);
    for (int k = 0; k < o.n_obj; ++k) {
        const int c = fs->obj_refs[o.obj_off + k];
        if (c < 0 || c >= fs->n_objects) return fail(NDT_E_INVALID, "object %d: bad nested primitive %d", object, c);
    }
    std::vector<double> rows;
    if (!hcube_hull_box(fs, o, fs->dims, rows)) return 0;
    memcpy(rows_out, rows.data(), rows.size() * sizeof(double));
    return 1;
}

extern "C" int ndt_hip_hcube_face_boxes(const ndt_flat_scene *fs, int32_t object, double *face_rows, uint64_t *possible)
{
    if (!face_rows || !possible) return fail(NDT_E_INVALID, "null argument");
    std::vector<double> hull((size_t)(fs && fs->dims > 0 && fs->dims <= NDT_MAX_DIMS ? fs->dims * (fs->dims + 2) : 1));
    const int rc = ndt_hip_hcube_hull_box(fs, object, hull.data());      // validates the arguments
    if (rc <= 0) return rc;
    std::vector<double> rows;
    HullFaces hf;
    if (!hcube_hull_box(fs, fs->objects[object], fs->dims, rows, &hf) || hf.n_faces == 0) return 0;
    memcpy(face_rows, hf.rows.data(), hf.rows.size() * sizeof(double));
    *possible = hf.possible;
    return hf.n_faces;
}

extern "C" int32_t ndt_hip_shard_rows(int32_t height, int32_t row_begin, int32_t row_step)
{
    if (row_step < 1 || row_begin < 0 || row_begin >= height) return 0;
    return (height - row_begin + row_step - 1) / row_step;
}

// ------------------------------------------------------------------ context

struct ndt_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    const NdtKernelTable *kt = nullptr;
    int dims = 0;
    bool have_scene = false;
    double aperture_radius = 0.0;   // camera.h:46 of the uploaded scene
    int cam_type = 0;
    bool have_eyes = false, have_local_axes = false;
    bool has_area_lights = false;   // LIGHT_DISK / LIGHT_RECT: every render is stochastic (ndt.c:116-147)
    SceneDesc sd{};
    std::vector<double> blob;
    double *d_blob = nullptr;
    size_t d_blob_words = 0;
    int tier = 0;
    int n_shadow_lights = 0;
    // workspace
    Workspace ws{};
    std::vector<void *> ws_allocs;
    std::vector<std::pair<void *, size_t>> pool;    // scratch of the multi-pass renderers (AaBuffers)
    long long ws_dims = 0;
    long long ws_slab_words = 0;
    int ws_nseg = 0;
    void *d_out = nullptr;          // staging for ndt_hip_render (host output)
    size_t d_out_bytes = 0;
    int *h_counters = nullptr;      // pinned
    LevelRange *h_levels = nullptr; // pinned, NDT_MAX_LEVELS + 1
    LevelRange *h_mail = nullptr;   // mapped + coherent: bounce ranges posted by k_level_step while the frame runs
    unsigned long long *h_mail_tag = nullptr;
    LevelRange *d_mail = nullptr;   // the device's view of the two
    unsigned long long *d_mail_tag = nullptr;
    unsigned long long frame_tag = 0;
    unsigned long long *h_done = nullptr;   // mapped + coherent: the frame's closing record (k_frame_done), [7] = its tag
    unsigned long long *d_done = nullptr;
    std::vector<hipEvent_t> ev_pool;
};

static const NdtKernelTable *table_for(int dims)
{
    switch (dims) {
    case 3: return ndt_kernel_table_3();
    case 4: return ndt_kernel_table_4();
    case 5: return ndt_kernel_table_5();
    case 6: return ndt_kernel_table_6();
    case 7: return ndt_kernel_table_7();
    case 8: return ndt_kernel_table_8();
    default: return nullptr;
    }
}

extern "C" int ndt_hip_create(int device, ndt_hip_ctx **out)
{
    if (!out) return fail(NDT_E_INVALID, "ndt_hip_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(NDT_E_DEVICE, "no HIP device available (%s); libndt_hip has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(NDT_E_INVALID, "device %d out of range (have %d)", device, count);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(NDT_E_DEVICE, "device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);
    ndt_hip_ctx *ctx = new ndt_hip_ctx();
    ctx->device = device;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return fail(NDT_E_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    e = hipHostMalloc((void **)&ctx->h_counters, 128 * sizeof(int), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_levels, (NDT_MAX_LEVELS + 1) * sizeof(LevelRange), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_mail, (NDT_MAX_LEVELS + 2) * sizeof(LevelRange), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_mail_tag, (NDT_MAX_LEVELS + 2) * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&ctx->d_mail, ctx->h_mail, 0);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&ctx->d_mail_tag, ctx->h_mail_tag, 0);
    if (e == hipSuccess) memset(ctx->h_mail_tag, 0, (NDT_MAX_LEVELS + 2) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_done, 8 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&ctx->d_done, ctx->h_done, 0);
    if (e == hipSuccess) memset(ctx->h_done, 0, 8 * sizeof(unsigned long long));
    if (e != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return fail(NDT_E_DEVICE, "hipHostMalloc: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return NDT_OK;
}

static void free_workspace(ndt_hip_ctx *ctx)
{
    for (void *p : ctx->ws_allocs) (void)hipFree(p);
    ctx->ws_allocs.clear();
    memset(&ctx->ws, 0, sizeof(ctx->ws));
    ctx->ws_slab_words = 0;
    ctx->ws_dims = 0;
    ctx->ws_nseg = 0;
}

extern "C" int ndt_hip_destroy(ndt_hip_ctx *ctx)
{
    if (!ctx) return NDT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_workspace(ctx);
    for (auto &slot : ctx->pool)
        if (slot.first) (void)hipFree(slot.first);
    if (ctx->d_blob) (void)hipFree(ctx->d_blob);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    if (ctx->h_levels) (void)hipHostFree(ctx->h_levels);
    if (ctx->h_mail) (void)hipHostFree(ctx->h_mail);
    if (ctx->h_mail_tag) (void)hipHostFree(ctx->h_mail_tag);
    if (ctx->h_done) (void)hipHostFree(ctx->h_done);
    for (hipEvent_t ev : ctx->ev_pool) (void)hipEventDestroy(ev);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return NDT_OK;
}

extern "C" void *ndt_hip_stream(ndt_hip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" int ndt_hip_synchronize(ndt_hip_ctx *ctx)
{
    if (!ctx) return fail(NDT_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return NDT_OK;
}

// ------------------------------------------------------------------ host vector math (prepare)
//
// Same operation order as the reference's vectNd.h (SSE2 lane-pair dot).  This file is
// compiled with -ffp-contract=off for host and device alike.

static double h_dot(const double *a, const double *b, int n)
{
    double s0 = a[0] * b[0];
    double s1 = a[1] * b[1];
    for (int i = 2; i < n; i += 2) {
        s0 = s0 + a[i] * b[i];
        if (i + 1 < n) s1 = s1 + a[i + 1] * b[i + 1];
    }
    return s0 + s1;
}
static void h_sub(const double *a, const double *b, double *r, int n) { for (int i = 0; i < n; ++i) r[i] = a[i] - b[i]; }
static void h_scale(const double *a, double s, double *r, int n) { for (int i = 0; i < n; ++i) r[i] = a[i] * s; }
static double h_len(const double *a, int n) { return sqrt(h_dot(a, a, n)); }
static void h_unitize(double *a, int n)
{
    double len = h_len(a, n);
    if (len > NDT_EPS || len < -NDT_EPS) h_scale(a, 1.0 / len, a, n);
}
static double h_dist(const double *a, const double *b, int n)
{
    double d[NDT_MAX_DIMS];
    h_sub(a, b, d, n);
    return h_len(d, n);
}
static double h_angle3(const double *p1, const double *p2, const double *p3, int n)
{
    // vectNd_angle3 / vectNd_angle, vectNd.c:83 / :64
    double a[NDT_MAX_DIMS], b[NDT_MAX_DIMS];
    h_sub(p1, p2, a, n);
    h_sub(p3, p2, b, n);
    double dp = h_dot(a, b, n);
    double div = h_len(a, n) * h_len(b, n);
    if (fabs(div) > NDT_EPS) return acos(dp / div);
    return -1;
}

// ------------------------------------------------------------------ scene validation + blob

namespace {

struct BlobBuilder {
    std::vector<double> w;
    int words() const { return (int)w.size(); }
    int push(double x) { w.push_back(x); return words() - 1; }
    int push_vec(const double *v, int n) { int at = words(); for (int i = 0; i < n; ++i) w.push_back(v[i]); return at; }
    int push_ints(int a, int b)
    {
        double d;
        int pair[2] = { a, b };
        memcpy(&d, pair, sizeof(d));
        w.push_back(d);
        return words() - 1;
    }
    // object reference list: one word per entry {int object, int header flags}; the flags are
    // patched in once the headers exist, so a list scan needs one LDS read per entry, not two
    int push_ref_list(const std::vector<int> &v, std::vector<int> &patch)
    {
        int at = words();
        for (size_t i = 0; i < v.size(); ++i) {
            patch.push_back(words());
            push_ints(v[i], 0);
        }
        if (v.empty()) push_ints(0, 0);
        return at;
    }
    void set_ints(int word, int a, int b)
    {
        int pair[2] = { a, b };
        memcpy(&w[word], pair, sizeof(double));
    }
};

bool vec_ok(const ndt_flat_scene *fs, int64_t off, int64_t count)
{
    return off >= 0 && count >= 0 && off + count * fs->dims <= fs->n_vecs;
}

} // namespace

// Renumber the kd-tree in preorder (left child = parent + 1) and check it is a tree.
static int kd_preorder(const ndt_flat_scene *fs, int node, int depth, std::vector<int> &order, std::vector<char> &seen,
                       int &max_depth)
{
    if (node < 0 || node >= fs->n_kd_nodes) return fail(NDT_E_INVALID, "kd node index %d out of range", node);
    if (seen[node]) return fail(NDT_E_INVALID, "kd node %d reached twice", node);
    seen[node] = 1;
    order.push_back(node);
    if (depth > max_depth) max_depth = depth;
    const ndt_flat_kdnode &k = fs->kd_nodes[node];
    if (k.dim >= 0) {
        if (k.dim >= fs->dims) return fail(NDT_E_INVALID, "kd node %d splits dimension %d of a %d-D scene", node, k.dim, fs->dims);
        int rc = kd_preorder(fs, k.left, depth + 1, order, seen, max_depth);
        if (rc) return rc;
        rc = kd_preorder(fs, k.right, depth + 1, order, seen, max_depth);
        if (rc) return rc;
    } else {
        if (k.num < 0 || k.first < 0 || (int64_t)k.first + k.num > fs->n_leaf_refs)
            return fail(NDT_E_INVALID, "kd leaf %d item range out of bounds", node);
        for (int i = 0; i < k.num; ++i) {
            int id = fs->leaf_refs[k.first + i];
            if (id < 0 || id >= fs->n_items) return fail(NDT_E_INVALID, "kd leaf %d lists object %d (n_items %d)", node, id, fs->n_items);
        }
    }
    return NDT_OK;
}

// Hull box of an hcube: an oriented box that contains every point the faces' intersect() can
// return.  NOT part of the reference's algorithm -- an exactness-preserving cull, like the
// kd-tree itself: a ray that misses the box cannot hit any face in the reference's own
// arithmetic, so trace() over the faces (hcube.c:241) returns "no hit", which is what the device
// gets by not scanning them.
//
// What orthotope.intersect (orthotope.c:150-300) accepts, with y = X - pos, unit basis columns
// B = [b_1..b_m] and A = B B^T:  |(A - I) y|^2 <= 2*EPSILON  (the `qc -= EPSILON` roots give
// exactly EPSILON, the closest-approach branch |dist| <= EPSILON) and, within_orthotope
// (orthotope.c:126-148),  -EPSILON <= y.b_i <= |dir_i| + EPSILON.  Split y = y_par + y_perp
// (span of B and its complement): |(A-I)y|^2 = |(A-I)y_par|^2 + |y_perp|^2.  In the orthonormal
// eigenvectors e_j = B w_j / sqrt(l_j) of A on span(B) (G = B^T B = W diag(l) W^T), with
// alpha_j = y.e_j:   |alpha_j| <= d/|l_j - 1|   and   alpha_j = (w_j . c)/sqrt(l_j) for the slab
// coordinates c_i = y.b_i in [-EPSILON, |dir_i|+EPSILON];  |y_perp| <= d;  d = sqrt(2*EPSILON).
// For an orthogonal face (the usual hypercube) l_j = 1 and this is the face grown by EPSILON;
// for the skewed bases scenes/random.c hands to hcube it is a small blob around pos.
// The hcube's box is the bounding box, in one orthonormal frame, of the alpha-box corners of
// all faces, grown by NDT_HULL_MARGIN = 0.02 > d = 0.01415 (y_perp, rounding).
// rows: N x { unit axis[N], centre coordinate, half extent }.
#define NDT_HULL_MARGIN 0.02
#define NDT_HULL_DELTA 0.01485      /* sqrt(2e-4) * 1.05 */
#define NDT_HULL_EPS 1.1e-4
#define NDT_HULL_MAX_FACES 63       /* face boxes: one bit per face in a 64-bit word whose top bit stays clear (trace_kd) */

// cyclic Jacobi: a (m x m, symmetric, row-major) -> eigenvalues on its diagonal, eigenvectors in the columns of w
static void jacobi_eig(std::vector<double> &a, std::vector<double> &w, int m)
{
    w.assign((size_t)m * m, 0.0);
    for (int i = 0; i < m; ++i) w[(size_t)i * m + i] = 1.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0;
        for (int i = 0; i < m; ++i)
            for (int j = i + 1; j < m; ++j) off += a[(size_t)i * m + j] * a[(size_t)i * m + j];
        if (off < 1e-26) break;
        for (int pi = 0; pi < m; ++pi)
            for (int q = pi + 1; q < m; ++q) {
                const double apq = a[(size_t)pi * m + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (a[(size_t)q * m + q] - a[(size_t)pi * m + pi]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), sn = t * c;
                for (int k = 0; k < m; ++k) {       // columns
                    const double akp = a[(size_t)k * m + pi], akq = a[(size_t)k * m + q];
                    a[(size_t)k * m + pi] = c * akp - sn * akq;
                    a[(size_t)k * m + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < m; ++k) {       // rows
                    const double apk = a[(size_t)pi * m + k], aqk = a[(size_t)q * m + k];
                    a[(size_t)pi * m + k] = c * apk - sn * aqk;
                    a[(size_t)q * m + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < m; ++k) {
                    const double wkp = w[(size_t)k * m + pi], wkq = w[(size_t)k * m + q];
                    w[(size_t)k * m + pi] = c * wkp - sn * wkq;
                    w[(size_t)k * m + q] = sn * wkp + c * wkq;
                }
            }
    }
}

// corner points (world coordinates) of a box that contains the in-span part of one face's acceptance region
static bool face_region_corners(const double *pos, const double *dir, int m, int n, std::vector<double> &pts)
{
    if (m == 0) {
        pts.insert(pts.end(), pos, pos + n);
        return true;
    }
    std::vector<double> bu((size_t)m * n), len((size_t)m);
    for (int i = 0; i < m; ++i) {
        len[i] = h_len(dir + i * n, n);
        if (!(len[i] > 0) || !std::isfinite(len[i])) return false;
        memcpy(&bu[(size_t)i * n], dir + i * n, n * sizeof(double));
        h_unitize(&bu[(size_t)i * n], n);       // vectNd_unitize, as orthotope.c:37
    }
    std::vector<double> g((size_t)m * m), w;
    double off = 0;
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
            double d = 0;
            for (int q = 0; q < n; ++q) d += bu[(size_t)i * n + q] * bu[(size_t)j * n + q];
            g[(size_t)i * m + j] = d;
            if (i != j && fabs(d) > off) off = fabs(d);
        }
    if (off < 1e-12) {
        // orthogonal face: keep its own axes (an eigen-solver may return any rotation of a repeated eigenvalue)
        w.assign((size_t)m * m, 0.0);
        for (int i = 0; i < m; ++i) w[(size_t)i * m + i] = 1.0;
    } else {
        jacobi_eig(g, w, m);
    }
    std::vector<double> e((size_t)m * n), lo((size_t)m), hi((size_t)m);
    for (int j = 0; j < m; ++j) {
        const double l = g[(size_t)j * m + j];
        if (!(l > 1e-10) || !std::isfinite(l)) return false;       // dependent directions
        const double sl = sqrt(l);
        double slo = 0, shi = 0;
        for (int q = 0; q < n; ++q) e[(size_t)j * n + q] = 0;
        for (int i = 0; i < m; ++i) {
            const double wij = w[(size_t)i * m + j];
            for (int q = 0; q < n; ++q) e[(size_t)j * n + q] += bu[(size_t)i * n + q] * wij / sl;
            const double c0 = -NDT_HULL_EPS * wij, c1 = (len[i] + NDT_HULL_EPS) * wij;
            slo += (c0 < c1 ? c0 : c1) / sl;
            shi += (c0 < c1 ? c1 : c0) / sl;
        }
        lo[j] = slo;
        hi[j] = shi;
        if (fabs(l - 1) > 1e-9) {
            const double r = NDT_HULL_DELTA / fabs(l - 1);
            if (-r > lo[j]) lo[j] = -r;
            if (r < hi[j]) hi[j] = r;
        }
        if (lo[j] > hi[j]) return true;     // empty region: this face can never be hit; contributes nothing
    }
    for (unsigned bits = 0; bits < (1u << m); ++bits) {
        std::vector<double> pt(pos, pos + n);
        for (int j = 0; j < m; ++j) {
            const double aj = (bits & (1u << j)) ? hi[j] : lo[j];
            for (int q = 0; q < n; ++q) pt[q] += aj * e[(size_t)j * n + q];
        }
        pts.insert(pts.end(), pt.begin(), pt.end());
    }
    return true;
}

static bool hcube_hull_box(const ndt_flat_scene *fs, const ndt_flat_object &o, int n, std::vector<double> &rows, HullFaces *faces)
{
    std::vector<double> pts;
    std::vector<size_t> face_begin;             // first corner point of every face (its region's corners are consecutive)
    std::vector<std::vector<double>> axes;      // unit face axes, for the aligned candidate frame
    for (int k = 0; k < o.n_obj; ++k) {
        face_begin.push_back(pts.size() / n);
        const ndt_flat_object &f = fs->objects[fs->obj_refs[o.obj_off + k]];
        if (f.type != NDT_OBJ_ORTHOTOPE || f.n_flag < 1 || f.n_pos < 1) return false;
        if (f.flag_off < 0 || (int64_t)f.flag_off + f.n_flag > fs->n_flags) return false;
        const int m = fs->flags[f.flag_off];
        if (m < 0 || m > f.n_dir || m > n || m > 16) return false;
        if (!vec_ok(fs, f.pos_off, 1) || !vec_ok(fs, f.dir_off, m)) return false;
        if (!face_region_corners(fs->vecs + f.pos_off, fs->vecs + f.dir_off, m, n, pts)) return false;
        for (int a = 0; a < m; ++a) {
            std::vector<double> u(fs->vecs + f.dir_off + a * n, fs->vecs + f.dir_off + (a + 1) * n);
            h_unitize(u.data(), n);
            axes.push_back(u);
        }
    }
    const size_t n_pts = pts.size() / n;
    if (n_pts == 0) return false;
    for (double x : pts)
        if (!std::isfinite(x)) return false;

    // candidate orthonormal frames: Gram-Schmidt of the face axes, principal axes of the points, world axes
    auto complete = [&](std::vector<std::vector<double>> &frame, const std::vector<std::vector<double>> &cands, double keep) {
        for (const auto &a : cands) {
            if ((int)frame.size() >= n) break;
            std::vector<double> r(a);
            for (const auto &u : frame) {
                double d = 0;
                for (int c = 0; c < n; ++c) d += a[c] * u[c];
                for (int c = 0; c < n; ++c) r[c] -= d * u[c];
            }
            const double l = h_len(r.data(), n);
            if (l > keep) {
                for (int c = 0; c < n; ++c) r[c] /= l;
                frame.push_back(r);
            }
        }
    };
    std::vector<std::vector<double>> world;
    for (int j = 0; j < n; ++j) {
        std::vector<double> e((size_t)n, 0.0);
        e[j] = 1.0;
        world.push_back(e);
    }
    std::vector<std::vector<std::vector<double>>> frames(3);
    complete(frames[0], axes, 0.5);
    {
        std::vector<double> mean((size_t)n, 0.0), cov((size_t)n * n, 0.0), w;
        for (size_t i = 0; i < n_pts; ++i)
            for (int c = 0; c < n; ++c) mean[c] += pts[i * n + c] / (double)n_pts;
        for (size_t i = 0; i < n_pts; ++i)
            for (int a = 0; a < n; ++a)
                for (int c = 0; c < n; ++c) cov[(size_t)a * n + c] += (pts[i * n + a] - mean[a]) * (pts[i * n + c] - mean[c]);
        jacobi_eig(cov, w, n);
        std::vector<std::vector<double>> pc;
        for (int j = 0; j < n; ++j) {
            std::vector<double> u((size_t)n);
            for (int c = 0; c < n; ++c) u[c] = w[(size_t)c * n + j];
            pc.push_back(u);
        }
        complete(frames[1], pc, 0.5);
    }
    double best_cost = 0;
    int best = -1;
    std::vector<double> best_rows;
    for (int fi = 0; fi < 3; ++fi) {
        auto &frame = frames[fi];
        for (double keep = 0.5; (int)frame.size() < n && keep > 1e-4; keep *= 0.5) complete(frame, world, keep);
        if ((int)frame.size() < n) continue;
        std::vector<double> cand, half((size_t)n);
        for (int a = 0; a < n; ++a) {
            double lo = 1e300, hi = -1e300;
            for (size_t i = 0; i < n_pts; ++i) {
                double d = 0;
                for (int c = 0; c < n; ++c) d += pts[i * n + c] * frame[a][c];
                if (d < lo) lo = d;
                if (d > hi) hi = d;
            }
            cand.insert(cand.end(), frame[a].begin(), frame[a].end());
            cand.push_back(0.5 * (lo + hi));
            half[a] = 0.5 * (hi - lo) + NDT_HULL_MARGIN;
            cand.push_back(half[a]);
        }
        // what a random ray sees of a box grows with its surface: sum over axes of the product of the other extents
        double cost = 0;
        for (int a = 0; a < n; ++a) {
            double prod = 1;
            for (int c = 0; c < n; ++c)
                if (c != a) prod *= half[c];
            cost += prod;
        }
        if (best < 0 || cost < best_cost) {
            best = fi;
            best_cost = cost;
            best_rows = cand;
        }
    }
    if (best < 0) return false;
    rows = best_rows;
    if (faces) {
        // Every face's own box in the chosen frame.  The hull box is the union of these: a ray that misses box f
        // cannot produce a point face f's intersect() accepts (same argument, one face at a time), so the device
        // scans only the faces whose box the ray meets -- of the 2-D faces of a 4-D hcube, usually none or two.
        faces->rows.clear();
        faces->possible = 0;
        faces->n_faces = 0;
        if (o.n_obj <= NDT_HULL_MAX_FACES) {
            faces->n_faces = o.n_obj;
            face_begin.push_back(n_pts);
            for (int k = 0; k < o.n_obj; ++k) {
                const size_t p0 = face_begin[k], p1 = face_begin[k + 1];
                if (p1 > p0) faces->possible |= 1ull << k;
                for (int a = 0; a < n; ++a) {
                    double lo = 1e300, hi = -1e300;
                    for (size_t i = p0; i < p1; ++i) {
                        double d = 0;
                        for (int c = 0; c < n; ++c) d += pts[i * n + c] * rows[(size_t)a * (n + 2) + c];
                        if (d < lo) lo = d;
                        if (d > hi) hi = d;
                    }
                    if (p1 == p0) lo = hi = 0;
                    faces->rows.push_back(0.5 * (lo + hi));
                    faces->rows.push_back(0.5 * (hi - lo) + NDT_HULL_MARGIN);
                }
            }
        }
    }
    return true;
}

static int build_blob(ndt_hip_ctx *ctx, const ndt_flat_scene *fs)
{
    const int n = fs->dims;
    BlobBuilder b;
    SceneDesc sd{};
    sd.n_items = fs->n_items;
    sd.n_objects = fs->n_objects;
    sd.n_kd_nodes = fs->n_kd_nodes;
    sd.n_inf = fs->n_inf;
    sd.n_lights = fs->n_lights;
    sd.mask_words = (fs->n_items + 63) / 64;
    if (sd.mask_words < 1) sd.mask_words = 1;

    // ---- kd nodes, preorder
    std::vector<int> order;
    std::vector<char> seen((size_t)(fs->n_kd_nodes > 0 ? fs->n_kd_nodes : 1), 0);
    int max_depth = 0;
    if (fs->n_kd_nodes > 0) {
        int rc = kd_preorder(fs, 0, 1, order, seen, max_depth);
        if (rc) return rc;
        if (max_depth > NDT_KD_STACK)
            return fail(NDT_E_UNSUPPORTED, "kd-tree depth %d exceeds the traversal stack (%d)", max_depth, NDT_KD_STACK);
    }
    sd.kd_depth = max_depth;
    std::vector<int> new_index((size_t)(fs->n_kd_nodes > 0 ? fs->n_kd_nodes : 1), -1);
    for (size_t i = 0; i < order.size(); ++i) new_index[order[i]] = (int)i;
    sd.n_kd_nodes = (int)order.size();
    std::vector<int> leaf_list;
    sd.off_kd = b.words();
    for (size_t i = 0; i < order.size(); ++i) {
        const ndt_flat_kdnode &k = fs->kd_nodes[order[i]];
        if (k.dim >= 0) {
            if (new_index[k.left] != (int)i + 1) return fail(NDT_E_INVALID, "internal: preorder numbering");
            b.push_ints(k.dim, new_index[k.right]);
            b.push(k.boundary);
        } else {
            b.push_ints(-1, 0);
            b.push_ints((int)leaf_list.size(), k.num);
            for (int j = 0; j < k.num; ++j) leaf_list.push_back(fs->leaf_refs[k.first + j]);
        }
    }
    std::vector<int> ref_patch;
    sd.off_leaf = b.push_ref_list(leaf_list, ref_patch);
    std::vector<int> inf_list;
    for (int i = 0; i < fs->n_inf; ++i) {
        int id = fs->inf_refs[i];
        if (id < 0 || id >= fs->n_items) return fail(NDT_E_INVALID, "infinite list names object %d (n_items %d)", id, fs->n_items);
        inf_list.push_back(id);
    }
    sd.off_inf = b.push_ref_list(inf_list, ref_patch);

    // ---- object headers (filled after params are placed), bounding spheres, root box
    sd.off_hdr = b.words();
    for (int i = 0; i < fs->n_objects; ++i) {
        b.push_ints(0, 0);
        b.push_ints(0, 0);
    }
    sd.off_bs = b.words();
    for (int i = 0; i < fs->n_objects; ++i) {
        const ndt_flat_object &o = fs->objects[i];
        if (!vec_ok(fs, o.bounds_center_off, 1)) return fail(NDT_E_INVALID, "object %d: bounds centre out of range", i);
        b.push_vec(fs->vecs + o.bounds_center_off, n);
        b.push(o.bounds_radius);
        b.push(o.bounds_radius * o.bounds_radius);      // bounding.c:22
    }
    if (!vec_ok(fs, fs->bb_lower_off, 1) || !vec_ok(fs, fs->bb_upper_off, 1)) return fail(NDT_E_INVALID, "root box out of range");
    sd.off_bb = b.push_vec(fs->vecs + fs->bb_lower_off, n);
    b.push_vec(fs->vecs + fs->bb_upper_off, n);

    // ---- nested primitive lists
    std::vector<int> child_list;
    std::vector<int> child_first((size_t)(fs->n_objects > 0 ? fs->n_objects : 1), 0);
    for (int i = 0; i < fs->n_objects; ++i) {
        const ndt_flat_object &o = fs->objects[i];
        child_first[i] = (int)child_list.size();
        if (o.type == NDT_OBJ_HCUBE) {
            if (o.n_obj < 1) return fail(NDT_E_UNSUPPORTED, "object %d: hcube without faces (supply add_faces output)", i);
            if (o.obj_off < 0 || (int64_t)o.obj_off + o.n_obj > fs->n_obj_refs) return fail(NDT_E_INVALID, "object %d: child range", i);
            for (int k = 0; k < o.n_obj; ++k) {
                int c = fs->obj_refs[o.obj_off + k];
                if (c < 0 || c >= fs->n_objects || fs->objects[c].type == NDT_OBJ_HCUBE)
                    return fail(NDT_E_INVALID, "object %d: bad nested primitive %d", i, c);
                child_list.push_back(c);
            }
        }
    }
    sd.off_child = b.push_ref_list(child_list, ref_patch);

    // ---- per-type parameters = the plugins' prepare() output
    sd.off_params = b.words();
    for (int i = 0; i < fs->n_objects; ++i) {
        const ndt_flat_object &o = fs->objects[i];
        if (o.type < 0 || o.type >= NDT_OBJ_TYPE_COUNT) return fail(NDT_E_UNSUPPORTED, "object %d: unknown type %d", i, o.type);
        if (!vec_ok(fs, o.pos_off, o.n_pos) || !vec_ok(fs, o.dir_off, o.n_dir)) return fail(NDT_E_INVALID, "object %d: vector range", i);
        if (o.n_size < 0 || o.size_off < 0 || (int64_t)o.size_off + o.n_size > fs->n_sizes) return fail(NDT_E_INVALID, "object %d: size range", i);
        if (o.n_flag < 0 || o.flag_off < 0 || (int64_t)o.flag_off + o.n_flag > fs->n_flags) return fail(NDT_E_INVALID, "object %d: flag range", i);
        const double *pos = fs->vecs + o.pos_off;
        const double *dir = fs->vecs + o.dir_off;
        const double *size = fs->sizes + o.size_off;
        const int *flag = fs->flags + o.flag_off;
        int flags = o.type;
        if (o.bounds_radius > 0) flags |= NDT_F_GATE;
        if (o.transparent) flags |= NDT_F_TRANSPARENT;
        int aux0 = 0, aux1 = 0;
        const int p = b.words() - sd.off_params;
        double tmp[NDT_MAX_DIMS], ax[NDT_MAX_DIMS];
        auto need = [&](bool ok, const char *what) -> int {
            return ok ? NDT_OK : fail(NDT_E_INVALID, "object %d: %s", i, what);
        };
        int rc = NDT_OK;
        switch (o.type) {
        case NDT_OBJ_SPHERE:        // sphere.c:18-32 (pow(r,2.0) == r*r)
            if ((rc = need(o.n_pos >= 1 && o.n_size >= 1, "sphere needs 1 pos, 1 size"))) return rc;
            b.push_vec(pos, n);
            b.push(size[0] * size[0]);
            break;
        case NDT_OBJ_HPLANE:
        case NDT_OBJ_HDISK:
            if ((rc = need(o.n_pos >= 1 && o.n_dir >= 1 && (o.type == NDT_OBJ_HPLANE || o.n_size >= 1), "hplane/hdisk parameters"))) return rc;
            b.push_vec(pos, n);
            b.push_vec(dir, n);
            b.push(o.type == NDT_OBJ_HDISK ? size[0] : 0.0);
            break;
        case NDT_OBJ_CYLINDER: {    // cylinder.c:22-40
            if ((rc = need(o.n_pos >= 2 && o.n_size >= 1, "cylinder needs 2 pos, 1 size"))) return rc;
            h_sub(pos + n, pos, ax, n);
            h_unitize(ax, n);
            b.push_vec(pos, n);
            b.push_vec(ax, n);
            b.push(h_dist(pos + n, pos, n));
            b.push(h_dot(ax, ax, n));
            b.push(h_dot(pos, ax, n));
            b.push(size[0]);
            if (o.n_flag > 1 && flag[1] != 0) flags |= NDT_F_INF_ENDS;     // cylinder.c:87
            break;
        }
        case NDT_OBJ_HCYLINDER: {   // hcylinder.c:23-54
            const int m = n - 2;
            if ((rc = need(o.n_pos >= n - 1 && o.n_size >= 1, "hcylinder needs dims-1 pos, 1 size"))) return rc;
            b.push_vec(pos, n);
            b.push(size[0]);
            for (int k = 0; k < m; ++k) {
                h_sub(pos + (k + 1) * n, pos, ax, n);
                h_unitize(ax, n);
                b.push_vec(ax, n);
                b.push(h_dist(pos + (k + 1) * n, pos, n));
                b.push(h_dot(ax, ax, n));
                b.push(h_dot(pos, ax, n));
            }
            aux1 = m;
            if (o.n_flag != 0 && flag[0] != 0) flags |= NDT_F_INF_ENDS;    // hcylinder.c:107
            break;
        }
        case NDT_OBJ_ORTHOTOPE: {   // orthotope.c:23-54
            if ((rc = need(o.n_flag >= 1 && o.n_pos >= 1, "orthotope needs 1 pos, 1 flag"))) return rc;
            const int m = flag[0];
            if ((rc = need(m >= 0 && m <= o.n_dir && m <= n, "orthotope flag[0] vs directions"))) return rc;
            b.push_vec(pos, n);
            b.push(0.0);
            for (int k = 0; k < m; ++k) {
                memcpy(ax, dir + k * n, n * sizeof(double));
                h_unitize(ax, n);
                b.push_vec(ax, n);
                b.push(h_len(dir + k * n, n));
                b.push(h_dot(ax, ax, n));       // BdB
                b.push(h_dot(pos, ax, n));      // BdP
            }
            aux1 = m;
            break;
        }
        case NDT_OBJ_HCUBE: {
            aux0 = child_first[i];
            aux1 = o.n_obj;
            std::vector<double> rows;
            HullFaces hf;
            if (!getenv("NDT_HIP_NO_HULL_BOX") && hcube_hull_box(fs, o, n, rows, &hf)) {
                flags |= NDT_F_BOX;
                for (double x : rows) b.push(x);
                if (hf.n_faces > 0 && !getenv("NDT_HIP_NO_FACE_BOX")) {
                    // { possible-faces mask } + per face N x { centre, half extent }
                    flags |= NDT_F_FACEBOX;
                    b.push_ints((int)(hf.possible & 0xffffffffull), (int)(hf.possible >> 32));
                    for (double x : hf.rows) b.push(x);
                }
            } else {
                b.push(0.0);
            }
            break;
        }
        case NDT_OBJ_HFACET: {      // hfacet.c:43-87 + the ray-invariant dots of get_barycentric (hfacet.c:176-181)
            if ((rc = need(o.n_pos >= 3 && o.n_flag >= 1, "hfacet needs 3 pos, 1 flag"))) return rc;
            if ((rc = need(!flag[0] || o.n_dir >= 3, "hfacet with vertex normals needs 3 dir"))) return rc;
            double edge[3][NDT_MAX_DIMS], uedge0[NDT_MAX_DIMS], perp[NDT_MAX_DIMS];
            for (int k = 0; k < 3; ++k) h_sub(pos + ((k + 1) % 3) * n, pos + k * n, edge[k], n);
            memcpy(uedge0, edge[0], n * sizeof(double));
            h_unitize(uedge0, n);
            h_scale(edge[2], -1.0, edge[2], n);
            // vectNd_proj(edge2, edge0), vectNd.h:355
            double bb = h_dot(edge[0], edge[0], n);
            double ab = h_dot(edge[2], edge[0], n);
            h_scale(edge[0], ab / bb, tmp, n);
            h_sub(edge[2], tmp, perp, n);
            h_unitize(perp, n);
            b.push_vec(pos, n);
            b.push_vec(uedge0, n);
            b.push_vec(perp, n);
            b.push(h_dot(uedge0, edge[0], n));  // x2
            b.push(h_dot(perp, edge[0], n));    // y2
            b.push(h_dot(uedge0, edge[2], n));  // x3
            b.push(h_dot(perp, edge[2], n));    // y3
            for (int k = 0; k < 3; ++k) {
                if (flag[0]) b.push_vec(dir + k * n, n);
                else { double z[NDT_MAX_DIMS] = { 0 }; b.push_vec(z, n); }
            }
            if (flag[0]) flags |= NDT_F_USE_NORMALS;
            break;
        }
        case NDT_OBJ_FACET: {       // facet.c:42-83
            if ((rc = need(o.n_pos >= 3 && o.n_dir >= 1, "facet needs 3 pos, 1 dir"))) return rc;
            double edge0[NDT_MAX_DIMS], edge1[NDT_MAX_DIMS], b0[NDT_MAX_DIMS], b1[NDT_MAX_DIMS], angle[3];
            for (int k = 0; k < 3; ++k)
                angle[k] = h_angle3(pos + ((k + 2) % 3) * n, pos + k * n, pos + ((k + 1) % 3) * n, n);
            h_sub(pos + n, pos, edge0, n);
            h_sub(pos + 2 * n, pos + n, edge1, n);
            // vectNd_orthogonalize(edge0, edge1, basis0, basis1), vectNd.c:35
            double bb = h_dot(edge1, edge1, n);
            double ab = h_dot(edge0, edge1, n);
            h_scale(edge1, ab / bb, tmp, n);
            h_sub(edge0, tmp, b0, n);
            memcpy(b1, edge1, n * sizeof(double));
            h_unitize(b0, n);
            h_unitize(b1, n);
            b.push_vec(pos, n);
            b.push_vec(pos + n, n);
            b.push_vec(pos + 2 * n, n);
            b.push_vec(b0, n);
            b.push_vec(b1, n);
            b.push(h_dot(b0, b0, n));               // AdA, facet.c:191
            b.push(h_dot(b1, b1, n));
            b.push(h_dot(pos + n, b0, n));          // BdA, facet.c:200
            b.push(h_dot(pos + n, b1, n));
            b.push(angle[0]);
            b.push(angle[1]);
            b.push(angle[2]);
            b.push_vec(dir, n);
            break;
        }
        }
        b.set_ints(sd.off_hdr + 2 * i, flags, p);
        b.set_ints(sd.off_hdr + 2 * i + 1, aux0, aux1);
    }
    for (int w : ref_patch) {
        int pair[2];
        memcpy(pair, &b.w[w], sizeof(pair));
        int hdr[2];
        memcpy(hdr, &b.w[sd.off_hdr + 2 * pair[0]], sizeof(hdr));
        b.set_ints(w, pair[0], hdr[0]);
    }
    sd.trace_words = b.words();

    // ---- shading data: materials, lights, camera
    sd.off_mat = b.words();
    for (int i = 0; i < fs->n_objects; ++i) {
        const ndt_flat_object &o = fs->objects[i];
        b.push(o.red); b.push(o.green); b.push(o.blue);
        b.push(o.red_r); b.push(o.green_r); b.push(o.blue_r);
        b.push(o.refract_index);
        b.push(o.transparent ? 1.0 : 0.0);
    }
    sd.off_lights = b.words();
    int n_shadow_lights = 0;
    bool has_area_lights = false;
    for (int i = 0; i < fs->n_lights; ++i) {
        const ndt_flat_light &l = fs->lights[i];
        double zero[NDT_MAX_DIMS] = { 0 };
        if (l.type < 0 || l.type > NDT_LIGHT_RECT) return fail(NDT_E_INVALID, "light %d: type %d", i, l.type);
        const bool area = l.type == NDT_LIGHT_DISK || l.type == NDT_LIGHT_RECT;
        if (area && !vec_ok(fs, l.area_off, 2)) return fail(NDT_E_INVALID, "light %d: area lights need u1 / v1 (scene.c:182-195) in the flat scene", i);
        if (area) has_area_lights = true;
        const bool want_pos = l.type == NDT_LIGHT_POINT || l.type == NDT_LIGHT_SPOT || area;
        const bool want_dir = l.type == NDT_LIGHT_DIRECTIONAL || l.type == NDT_LIGHT_SPOT;
        if (want_pos && !vec_ok(fs, l.pos_off, 1)) return fail(NDT_E_INVALID, "light %d: position missing", i);
        if (want_dir && !vec_ok(fs, l.dir_off, 1)) return fail(NDT_E_INVALID, "light %d: direction missing", i);
        if (l.type != NDT_LIGHT_AMBIENT) ++n_shadow_lights;
        b.push_ints(l.type, 0);
        b.push(l.red); b.push(l.green); b.push(l.blue);
        b.push(l.angle);
        b.push_vec(want_pos ? fs->vecs + l.pos_off : zero, n);
        b.push_vec(want_dir ? fs->vecs + l.dir_off : zero, n);
        b.push(area ? l.radius : 0.0);
        b.push_vec(area ? fs->vecs + l.area_off : zero, n);         // u1
        b.push_vec(area ? fs->vecs + l.area_off + n : zero, n);     // v1
    }
    ctx->has_area_lights = has_area_lights;
    if (!vec_ok(fs, fs->cam_pos_off, 1) || !vec_ok(fs, fs->cam_img_orig_off, 1) || !vec_ok(fs, fs->cam_dir_x_off, 1) ||
        !vec_ok(fs, fs->cam_dir_y_off, 1))
        return fail(NDT_E_INVALID, "camera vectors out of range");
    sd.off_cam = b.push_vec(fs->vecs + fs->cam_pos_off, n);
    b.push_vec(fs->vecs + fs->cam_img_orig_off, n);
    b.push_vec(fs->vecs + fs->cam_dir_x_off, n);
    b.push_vec(fs->vecs + fs->cam_dir_y_off, n);
    b.push(fs->cam_focal_distance);
    for (int i = 0; i < 3; ++i) b.push(fs->ambient[i]);
    for (int i = 0; i < 4; ++i) b.push(fs->background[i]);
    // the rest of the camera, at off_cam + 4N + 8: type, hFov, vFov, leftEye, rightEye, localX, localY, localZ
    // (camera.h:34-75; zeros where the scene does not carry them -- ndt_hip_render checks before use)
    {
        const int32_t offs[5] = { fs->cam_left_eye_off, fs->cam_right_eye_off, fs->cam_local_x_off, fs->cam_local_y_off,
                                  fs->cam_local_z_off };
        b.push((double)fs->cam_type);
        b.push(fs->cam_h_fov);
        b.push(fs->cam_v_fov);
        for (int k = 0; k < 5; ++k) {
            if (offs[k] >= 0 && !vec_ok(fs, offs[k], 1)) return fail(NDT_E_INVALID, "camera vectors out of range");
            if (offs[k] >= 0) b.push_vec(fs->vecs + offs[k], n);
            else for (int c = 0; c < n; ++c) b.push(0.0);
        }
        ctx->cam_type = fs->cam_type;
        ctx->have_eyes = offs[0] >= 0 && offs[1] >= 0;
        ctx->have_local_axes = offs[2] >= 0 && offs[3] >= 0 && offs[4] >= 0;
    }
    sd.total_words = b.words();

    ctx->sd = sd;
    ctx->blob.swap(b.w);
    ctx->n_shadow_lights = n_shadow_lights;
    // tier 0: trace sections fit the LDS budget and the visit mask fits registers
    const bool fits_lds = (size_t)sd.trace_words * sizeof(double) <= NDT_TRACE_LDS_LIMIT;
    ctx->tier = (fits_lds && sd.mask_words <= NDT_MASK_REG_WORDS) ? 0 : 1;
    return NDT_OK;
}

extern "C" int ndt_hip_upload_scene(ndt_hip_ctx *ctx, const ndt_flat_scene *fs)
{
    if (!ctx || !fs) return fail(NDT_E_INVALID, "NULL argument");
    if (fs->abi_version != NDT_HIP_ABI_VERSION) return fail(NDT_E_INVALID, "scene ABI %d, library ABI %d", fs->abi_version, NDT_HIP_ABI_VERSION);
    if (fs->dims < NDT_MIN_DIMS || fs->dims > NDT_MAX_DIMS)
        return fail(NDT_E_UNSUPPORTED, "%d dimensions: kernels are built for %d..%d", fs->dims, NDT_MIN_DIMS, NDT_MAX_DIMS);
    if (fs->cam_type < 0 || fs->cam_type > 2) return fail(NDT_E_UNSUPPORTED, "camera type %d", fs->cam_type);
    if (fs->cam_type != 0 && (fs->cam_local_x_off < 0 || fs->cam_local_y_off < 0 || fs->cam_local_z_off < 0))
        return fail(NDT_E_INVALID, "VR / panorama cameras need the local axes (camera.h:69-71) in the flat scene");
    if (fs->n_lights < 0 || fs->n_lights > NDT_MAX_LIGHTS) return fail(NDT_E_UNSUPPORTED, "%d lights (max %d)", fs->n_lights, NDT_MAX_LIGHTS);
    if (fs->n_objects < 0 || fs->n_items < 0 || fs->n_items > fs->n_objects) return fail(NDT_E_INVALID, "object counts");
    if (fs->n_kd_nodes < 0 || fs->n_inf < 0 || fs->n_leaf_refs < 0) return fail(NDT_E_INVALID, "kd-tree counts");
    if (fs->n_objects > 0 && !fs->objects) return fail(NDT_E_INVALID, "objects is NULL");
    if (fs->n_lights > 0 && !fs->lights) return fail(NDT_E_INVALID, "lights is NULL");
    if (!fs->vecs) return fail(NDT_E_INVALID, "vecs is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->have_scene = false;
    int rc = build_blob(ctx, fs);
    if (rc) return rc;
    ctx->dims = fs->dims;
    ctx->kt = table_for(fs->dims);
    if (!ctx->kt) return fail(NDT_E_UNSUPPORTED, "no kernels for %d dimensions", fs->dims);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->d_blob_words < ctx->blob.size()) {
        if (ctx->d_blob) HIP_TRY(hipFree(ctx->d_blob));
        ctx->d_blob = nullptr;
        HIP_TRY(hipMalloc((void **)&ctx->d_blob, ctx->blob.size() * sizeof(double)));
        ctx->d_blob_words = ctx->blob.size();
    }
    HIP_TRY(hipMemcpyAsync(ctx->d_blob, ctx->blob.data(), ctx->blob.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->aperture_radius = fs->cam_aperture_radius;
    ctx->have_scene = true;
    return NDT_OK;
}

// ------------------------------------------------------------------ workspace

template <typename T> static int ws_alloc(ndt_hip_ctx *ctx, T **p, size_t count)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, (count > 0 ? count : 1) * sizeof(T));
    if (e != hipSuccess) return fail(NDT_E_NOMEM, "hipMalloc of %zu bytes: %s", count * sizeof(T), hipGetErrorString(e));
    ctx->ws_allocs.push_back(q);
    *p = (T *)q;
    return NDT_OK;
}

static int ensure_workspace(ndt_hip_ctx *ctx, long long cap, long long sh_cap)
{
    Workspace &ws = ctx->ws;
    const bool need_slab = ctx->tier == 1;
    const long long slab_lanes = 2048LL * NDT_TRACE_BLOCK;
    const long long slab_words = need_slab ? slab_lanes * ctx->sd.mask_words : 0;
    if (ws.cap >= cap && ws.sh_cap >= sh_cap && ctx->ws_dims == ctx->dims && ctx->ws_slab_words >= slab_words &&
        ctx->ws_nseg >= ctx->n_shadow_lights)
        return NDT_OK;
    if (cap < ws.cap) cap = ws.cap;
    if (sh_cap < ws.sh_cap) sh_cap = ws.sh_cap;
    cap = (cap + 63) & ~63LL;           // vectors are stored in tiles of 64 slots (load_soa / store_soa)
    sh_cap = (sh_cap + 63) & ~63LL;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_workspace(ctx);
    const int n = ctx->dims;
    int rc;
    ws.cap = cap;
    ws.sh_cap = sh_cap;
    if ((rc = ws_alloc(ctx, &ws.ray_o, (size_t)n * cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.ray_v, (size_t)n * cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.frac, (size_t)cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.depth, (size_t)cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.rng_key, (size_t)cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.depth_left, (size_t)cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.hit_obj, (size_t)cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.hit_prim, (size_t)cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.hit_p, (size_t)n * cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.hit_n, (size_t)n * cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.clr, (size_t)3 * cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.child_refl, (size_t)cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.child_refr, (size_t)cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.sh_idx, (size_t)cap * (ctx->n_shadow_lights > 0 ? ctx->n_shadow_lights : 1)))) return rc;
    if ((rc = ws_alloc(ctx, &ws.sh_mask, (size_t)cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.count, (size_t)cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.so, (size_t)n * sh_cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.sv, (size_t)n * sh_cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.slim, (size_t)sh_cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.sobj, (size_t)sh_cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.sprim, (size_t)sh_cap))) return rc;
    if ((rc = ws_alloc(ctx, &ws.counters, NDT_CNT_TOTAL))) return rc;
    if ((rc = ws_alloc(ctx, &ws.ref_rays, 64 * 8))) return rc;
    if ((rc = ws_alloc(ctx, &ws.dbg, 160))) return rc;
    if ((rc = ws_alloc(ctx, &ws.exit_log, (size_t)NDT_EXIT_LOG_LAUNCHES * NDT_EXIT_LOG_WORDS))) return rc;
    if (getenv("NDT_HIP_SHADE_PROBE") && (rc = ws_alloc(ctx, &ws.shade_log, (size_t)2 * NDT_SHADE_LOG_WAVES))) return rc;
    if ((rc = ws_alloc(ctx, &ws.levels, NDT_MAX_LEVELS + 1))) return rc;
    ws.mask_slab_lanes = slab_lanes;
    if (need_slab) {
        if ((rc = ws_alloc(ctx, &ws.mask_slab, (size_t)slab_words))) return rc;
    }
    ctx->ws_slab_words = slab_words;
    ctx->ws_dims = ctx->dims;
    ctx->ws_nseg = ctx->n_shadow_lights > 0 ? ctx->n_shadow_lights : 1;
    return NDT_OK;
}

// ------------------------------------------------------------------ dimension-independent kernels

// Everything a frame needs reset, in one launch (five small copies / fills of 10 us each before): node tail and
// overflow flags, the work-queue heads of the launches the frame can have, both parities of the shadow-segment
// counters, the reference-ray partial sums, the diagnostic words, and the primaries' range.
__global__ void k_frame_init(Workspace ws, int n_primary, LevelRange level0, int queue_ints)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (int k = i; k < queue_ints; k += stride) ws.counters[NDT_CNT_QUEUE + k] = 0;
    for (int k = i; k < NDT_CNT_TOTAL - NDT_CNT_SEG; k += stride) ws.counters[NDT_CNT_SEG + k] = 0;
    for (int k = i; k < 64 * 8; k += stride) ws.ref_rays[k] = 0ull;
    for (int k = i; k < 160; k += stride) ws.dbg[k] = 0ull;
    if (i < 4) ws.counters[i] = (i == 0) ? n_primary : 0;
    if (i == 0) ws.levels[0] = level0;
}

// The frame's closing record, written to host-visible memory by the last kernel of the frame: the host polls its
// tag instead of queueing three small read-backs and synchronising the stream.  One wavefront.
//   [0] node tail  [1] overflow flags  [2] shadow slots wanted  [3] shadow rays of the frame  [4] bounces with nodes
//   [5] rays the reference would have traced  [7] tag
__global__ void k_frame_done(Workspace ws, int n_run, unsigned long long *done, unsigned long long tag)
{
    const int lane = threadIdx.x;
    unsigned long long ref = ws.ref_rays[8 * lane];         // 64 partial sums, one 64-byte line each
    for (int d = 32; d > 0; d >>= 1) ref += __shfl_xor(ref, d, 64);
    if (lane != 0) return;
    long long shadow = 0;
    int used = 0;
    for (int b = 0; b < n_run; ++b) {
        if (ws.levels[b].count <= 0) break;
        shadow += ws.levels[b].n_shadow;
        ++used;
    }
    done[0] = (unsigned long long)(long long)ws.counters[0];
    done[1] = (unsigned long long)(long long)ws.counters[2];
    done[2] = (unsigned long long)(long long)ws.counters[3];
    done[3] = (unsigned long long)shadow;
    done[4] = (unsigned long long)used;
    done[5] = ref;
    __threadfence_system();
    __hip_atomic_store(&done[7], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// After shade_emit(level) has spawned the next bounce: publish its range, note the shadow rays
// this bounce emitted, and clear the other parity's segment counters for the next shade_emit.
// The range is also posted to host-visible memory: k_level_step runs early in a bounce (before
// its long trace launch), so by the time the host wants to enqueue the next bounce the answer
// is there and the GPU never waits for the host.  One wavefront.
__global__ void k_level_step(Workspace ws, int level, int n_seg, unsigned long long tag)
{
    const int lane = threadIdx.x;
    int *seg = NDT_SEG_COUNTERS(ws, level);
    long long mine = (lane < n_seg) ? seg[lane] : 0;
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
    NDT_SEG_COUNTERS(ws, level + 1)[lane] = 0;
    if (lane != 0) return;
    const LevelRange cur = ws.levels[level];
    ws.levels[level].n_shadow = mine;
    LevelRange next;
    next.begin = cur.begin + cur.count;
    next.count = (long long)ws.counters[0] - next.begin;
    next.n_shadow = 0;
    if (next.count < 0 || ws.counters[2] != 0) next.count = 0;         // node pool overflow: the host retries
    next.seg_stride = (next.count + 63) & ~63LL;
    if ((long long)n_seg * next.seg_stride > ws.sh_cap) {
        // the shadow queue cannot hold this bounce: flag it, tell the host how much it needs, stop here
        atomicOr(&ws.counters[2], 2);
        const long long need = (long long)n_seg * next.seg_stride;
        ws.counters[3] = need > 0x7fffffffLL ? 0x7fffffff : (int)need;
        next.count = 0;
        next.seg_stride = 0;
    }
    ws.levels[level + 1] = next;
    // ... and for the host, which enqueues bounce level+1 only once it knows there is one
    ws.mail[level + 1] = next;
    __threadfence_system();
    __hip_atomic_store(&ws.mail_tag[level + 1], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Bottom-up combine of one bounce: get_ray_color's blend of its own colour with the colours
// its reflection / refraction children returned (ndt.c:402-429), in the reference's order.
__device__ __forceinline__ void resolve_node(const double *blob, const SceneDesc &sd, const Workspace &ws, int specular, long long g)
{
    if (ws.depth_left[g] <= 0) return;
    const int obj = ws.hit_obj[g];
    if (obj < 0) return;                        // background node: colour and count already final
    const int mw = sd.off_mat + 8 * obj;
    const double hitr[3] = { blob[mw + 3], blob[mw + 4], blob[mw + 5] };
    double c[3] = { ws.clr[0 * ws.cap + g], ws.clr[1 * ws.cap + g], ws.clr[2 * ws.cap + g] };
    int cnt = ws.count[g];
    const int refl = ws.child_refl[g];
    if (refl != -1) {
        double ref[3] = { 0.0, 0.0, 0.0 };
        if (refl >= 0) {
            ref[0] = ws.clr[0 * ws.cap + refl]; ref[1] = ws.clr[1 * ws.cap + refl]; ref[2] = ws.clr[2 * ws.cap + refl];
            cnt += ws.count[refl];
        }
        for (int k = 0; k < 3; ++k) {
            if (specular) c[k] = (1 - hitr[k]) * (c[k]) + (hitr[k]) * ref[k];     // ndt.c:405-407
            else c[k] += hitr[k] * ref[k];                                         // ndt.c:411-413
        }
    }
    const int refr = ws.child_refr[g];
    if (refr != -1) {
        double ref[3] = { 0.0, 0.0, 0.0 };
        if (refr >= 0) {
            ref[0] = ws.clr[0 * ws.cap + refr]; ref[1] = ws.clr[1 * ws.cap + refr]; ref[2] = ws.clr[2 * ws.cap + refr];
            cnt += ws.count[refr];
        }
        for (int k = 0; k < 3; ++k) c[k] += (1.0 - hitr[k]) * ref[k];              // ndt.c:426-428
    }
    ws.clr[0 * ws.cap + g] = c[0];
    ws.clr[1 * ws.cap + g] = c[1];
    ws.clr[2 * ws.cap + g] = c[2];
    ws.count[g] = cnt;
}

__global__ void __launch_bounds__(256) k_resolve(const double *blob, SceneDesc sd, Workspace ws, int specular, int level)
{
    const LevelRange lr = ws.levels[level];
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < lr.count; r += (long long)gridDim.x * blockDim.x)
        resolve_node(blob, sd, ws, specular, lr.begin + r);
}

// get_pixel_color's adaptive loop (ndt.c:488-568) replayed on the one deterministic sample:
// with samples == 1 the reference re-traces the identical ray k times, k decided by the
// running-mean test below; the result is (c+...+c)/k and the k-fold ray count.
__global__ void __launch_bounds__(256) k_finish_pixels(const double *blob, SceneDesc sd, Workspace ws, RenderGeom rg, int N_,
                                                       double *rgba, double *depth_out)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long weighted = 0ull;
    if (g < rg.n_primary && ws.depth_left[g] > 0) {
        long long out_idx = g;                          // list mode: one colour per sample
        if (!rg.samples) {
            const int tile = (int)(g >> 6), lane = (int)(g & 63);
            const int px = (tile % rg.tiles_x) * 8 + (lane & 7);
            const int py = (tile / rg.tiles_x) * 8 + (lane >> 3);
            out_idx = (long long)py * rg.width + px;    // dbl_image_set_pixel, image.c:126
        }
        const double l[4] = { ws.clr[0 * ws.cap + g], ws.clr[1 * ws.cap + g], ws.clr[2 * ws.cap + g],
                              ws.hit_obj[g] >= 0 ? 1.0 : blob[sd.off_cam + 4 * N_ + 7] };
        double t[4] = { 0.0, 0.0, 0.0, 0.0 };
        const double max_diff = 1.0 / 256.0;
        double clr_diff = 256;
        int samples = 0;
        // Every sample is the same colour l, so the reference's
        //     clr_diff = max_c |t_c/(i-1) - (t_c+l_c)/i|        (t = l+l+...+l, i terms)
        // is max_c(l_c)/(i(i-1)) up to rounding (relative error < 4 i^2 ulp: a difference of two
        // quotients of an i-term running sum).  The six divisions are only spent when that
        // estimate lies inside the error band around 1/256; otherwise the loop-exit decision
        // is already certain and identical to the exact one.
        const double gb0 = (fabs(l[1]) > fabs(l[2])) ? fabs(l[1]) : fabs(l[2]);
        const double lmax = (fabs(l[0]) > gb0) ? fabs(l[0]) : gb0;
        const bool finite = lmax <= 1.0e300;          // false for inf / nan: always take the exact path
        for (int i = 0; i < 1 || (!rg.raw_samples && i < 10000 && clr_diff > max_diff); ++i) {
            if (i > 1) {
                const double ii = (double)i * (double)(i - 1);
                const double est = lmax / ii;
                const double band = 1.0e-15 * (8.0 * (double)i * (double)i) + 1.0e-12;
                if (finite && est > max_diff * (1.0 + band)) {
                    clr_diff = est;             // certainly still above the threshold: keep sampling
                } else if (finite && est < max_diff * (1.0 - band)) {
                    clr_diff = est;             // certainly converged: the loop ends here
                } else {
                    const double dr = fabs(t[0] / (i - 1) - (t[0] + l[0]) / i);
                    const double dg = fabs(t[1] / (i - 1) - (t[1] + l[1]) / i);
                    const double db = fabs(t[2] / (i - 1) - (t[2] + l[2]) / i);
                    const double gb = (dg > db) ? dg : db;      // MAX, image.h:31
                    clr_diff = (dr > gb) ? dr : gb;
                }
            }
            t[0] += l[0]; t[1] += l[1]; t[2] += l[2]; t[3] += l[3];
            samples += 1;
        }
        double *out = rgba + out_idx * 4;
        out[0] = t[0] / samples;
        out[1] = t[1] / samples;
        out[2] = t[2] / samples;
        out[3] = t[3] / samples;
        if (depth_out) depth_out[out_idx] = ws.depth[g];       // ndt.c:753-756
        weighted = (unsigned long long)samples * (unsigned long long)ws.count[g];
    }
    // wavefront sum, then one atomic per wavefront spread over 64 cache lines (a single word
    // saturates near 90 atomics/us, and there are 32k wavefronts at 1080p)
    for (int d = 32; d > 0; d >>= 1) weighted += __shfl_down(weighted, d, 64);
    if ((threadIdx.x & 63) == 0 && weighted) atomicAdd(ws.ref_rays + 8 * ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63), weighted);
}

// max_optic_depth <= 0: get_ray_color returns black without tracing (ndt.c:340)
__global__ void k_fill_black(double *rgba, long long n_pixels)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    rgba[4 * i + 0] = 0.0; rgba[4 * i + 1] = 0.0; rgba[4 * i + 2] = 0.0; rgba[4 * i + 3] = 1.0;
}

// pixel_d2c, image.h:36-39
__global__ void k_quantize(const double *rgba, unsigned char *out, long long n_values)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_values) return;
    const double d = rgba[i];
    double m = (1.0 < d) ? 1.0 : d;
    m = (0.0 > m) ? 0.0 : m;
    out[i] = (unsigned char)(sqrt(m) * 255);
}

// ------------------------------------------------------------------ render

static hipEvent_t get_event(ndt_hip_ctx *ctx, size_t idx)
{
    while (ctx->ev_pool.size() <= idx) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return nullptr;
        ctx->ev_pool.push_back(ev);
    }
    return ctx->ev_pool[idx];
}

static double wall_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// One pass of the ray pipeline over the primaries `rg` describes: primary rays, the bounce loop,
// bottom-up resolve, per-primary colour (k_finish_pixels) into d_rgba.  Grid mode writes a
// rows x width image, list mode one RGBA per sample.  max_depth > 0 (the callers handle -l 0).
static int render_pass(ndt_hip_ctx *ctx, RenderGeom rg, bool prof, void *d_rgba, ndt_render_stats &st, void *d_depth = nullptr)
{
    rg.want_depth = d_depth ? 1 : 0;
    hipStream_t s = ctx->stream;
    const long long n_primary = rg.n_primary;
    const long long n_pixels = rg.samples ? (long long)rg.n_samples : (long long)rg.rows * rg.width;
    long long cap = ctx->ws.cap, sh_cap = ctx->ws.sh_cap;
    if (cap < 2 * n_primary + 4096) cap = 2 * n_primary + 4096;
    const long long want_sh = n_primary * (ctx->n_shadow_lights > 0 ? ctx->n_shadow_lights : 1) + 4096;
    if (sh_cap < want_sh) sh_cap = want_sh;
    if (getenv("NDT_HIP_TEST_SMALL_POOL") && ctx->ws.cap == 0) {
        // tests only: a fresh context starts with a node pool that a reflective scene overflows, so that the
        // overflow -> grow -> render-again path below is exercised (tests/test_gpu_parity.py)
        cap = ((n_primary + 63) & ~63LL) + 64;
    }

    const NdtKernelTable *kt = ctx->kt;
    for (int attempt = 0; attempt < 8; ++attempt) {
        if (cap > 0x7fffff00LL || sh_cap > 0x7fffff00LL) return fail(NDT_E_NOMEM, "ray tree exceeds 2^31 nodes");
        int rc = ensure_workspace(ctx, cap, sh_cap);
        if (rc) return rc;
        Workspace ws = ctx->ws;
        size_t ev_n = 0;
        hipEvent_t ev_begin = nullptr, ev_end = nullptr;
        std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_ev;
        std::vector<std::string> trace_dbg;
        if (prof) {
            // frame time = start of the frame's first kernel .. end of its last (their own dispatch timestamps)
            ev_begin = get_event(ctx, ev_n++);
            ev_end = get_event(ctx, ev_n++);
        }
        int *hc = ctx->h_counters;
        if (prof && getenv("NDT_HIP_EXIT_PROBE"))
            HIP_TRY(hipMemsetAsync(ws.exit_log, 0, (size_t)NDT_EXIT_LOG_LAUNCHES * NDT_EXIT_LOG_WORDS * sizeof(unsigned int), s));
        // The stream is never synchronised inside a frame: the range of every bounce is published
        // on the device (k_level_step) and read there; the host only learns, from the mailbox,
        // whether there is a next bounce to enqueue.  Bounce 0 = the primaries.
        ws.mail = ctx->d_mail;
        ws.mail_tag = ctx->d_mail_tag;
        const unsigned long long tag = ++ctx->frame_tag;
        const int n_seg = ctx->n_shadow_lights;
        const int n_levels = rg.max_depth > 1 ? rg.max_depth : 1;      // a node spawns children only while depth_left > 1
        int n_run = n_levels;                                           // bounces actually enqueued
        LevelRange *hl = ctx->h_levels;
        hl[0].begin = 0;
        hl[0].count = rg.n_primary;
        hl[0].seg_stride = (rg.n_primary + 63) & ~63LL;
        hl[0].n_shadow = 0;
        if ((long long)n_seg * hl[0].seg_stride > ws.sh_cap) {
            sh_cap = (long long)n_seg * hl[0].seg_stride;
            continue;
        }
        {
            int slots = n_levels + 2;               // one trace launch per bounce + the primaries' own
            if (slots > NDT_QUEUE_SLOTS) slots = NDT_QUEUE_SLOTS;
            if (prof)
                hipExtLaunchKernelGGL(k_frame_init, dim3(8), dim3(256), 0, s, ev_begin, nullptr, 0u, ws, rg.n_primary, hl[0], slots * NDT_QUEUE_INTS);
            else
                hipLaunchKernelGGL(k_frame_init, dim3(8), dim3(256), 0, s, ws, rg.n_primary, hl[0], slots * NDT_QUEUE_INTS);
        }
        int queue_slot = 0;
        int launches = 0;
        kt->primary(s, ctx->d_blob, ctx->sd, ws, rg);
        auto traced = [&](TraceJob &tj, const std::string &what) -> int {
            tj.queue = ws.counters + NDT_CNT_QUEUE + (queue_slot++) * NDT_QUEUE_INTS;
            static const bool exit_probe = getenv("NDT_HIP_EXIT_PROBE") != nullptr;
            tj.exit_log = (exit_probe && prof && launches < NDT_EXIT_LOG_LAUNCHES) ? ws.exit_log + (size_t)launches * NDT_EXIT_LOG_WORDS : nullptr;
            if (prof) {
                hipEvent_t a = get_event(ctx, ev_n++), b2 = get_event(ctx, ev_n++);
                kt->trace(s, ctx->d_blob, ctx->sd, ws, tj, ctx->tier, ctx->sd.mask_words, a, b2);
                trace_ev.push_back({ a, b2 });
                trace_dbg.push_back(what);
            } else {
                kt->trace(s, ctx->d_blob, ctx->sd, ws, tj, ctx->tier, ctx->sd.mask_words, nullptr, nullptr);
            }
            ++launches;
            return NDT_OK;
        };
        // closest-hit queries of the primaries: the only launch that is not shared
        {
            TraceJob tj{};
            tj.n_seg = 0;
            tj.dense.o = ws.ray_o; tj.dense.v = ws.ray_v; tj.dense.stride = ws.cap; tj.dense.lim = nullptr;
            tj.dense.valid = ws.depth_left; tj.dense.out_obj = ws.hit_obj; tj.dense.out_prim = ws.hit_prim;
            tj.begin = 0; tj.count = rg.n_primary; tj.levels = nullptr;
            if ((rc = traced(tj, "closest 0"))) return rc;
        }
        long long upper = rg.n_primary;         // node count of the bounce
        std::vector<long long> level_nodes;
        // NDT_HIP_SHADE_PROBE=<k>: the k-th shade launch of the frame logs the life of each of its wavefronts
        static const int shade_probe = getenv("NDT_HIP_SHADE_PROBE") ? atoi(getenv("NDT_HIP_SHADE_PROBE")) : -1;
        int shade_launch = 0;
        long long shade_probe_finish_waves = 0;         // wavefronts of the lighting part of the probed launch
        long long shade_probe_emit_waves = 0;           // ... and of the shading part behind it (pair launches)
        auto shade_ws = [&](long long finish_nodes, long long emit_nodes_behind = 0) {
            Workspace w = ws;
            if (shade_launch++ != shade_probe || !prof) {
                w.shade_log = nullptr;
            } else {
                shade_probe_finish_waves = (finish_nodes + 255) / 256 * 4;
                shade_probe_emit_waves = (emit_nodes_behind + 255) / 256 * 4;
                (void)hipMemsetAsync(w.shade_log, 0, (size_t)2 * NDT_SHADE_LOG_WAVES * sizeof(unsigned int), s);
            }
            return w;
        };
        static const bool fuse_shade = !(getenv("NDT_HIP_NO_SHADE_PAIR") && atoi(getenv("NDT_HIP_NO_SHADE_PAIR")));
        int pending_finish = -1;                // bounce whose lighting has not been launched yet
        long long pending_upper = 0;
        for (int b = 0; b < n_levels; ++b) {
            if (queue_slot + 1 > NDT_QUEUE_SLOTS || b + 1 > NDT_MAX_LEVELS)
                return fail(NDT_E_UNSUPPORTED, "more than %d bounces", NDT_QUEUE_SLOTS - 1);
            if (b > 0) {
                // posted by k_level_step(b-1), which ran right after shade_emit(b-1)
                const double t_wait = wall_s();
                while (__atomic_load_n(&ctx->h_mail_tag[b], __ATOMIC_ACQUIRE) != tag) {
                    if (wall_s() - t_wait > 30.0) {
                        HIP_TRY(hipStreamSynchronize(s));
                        if (__atomic_load_n(&ctx->h_mail_tag[b], __ATOMIC_ACQUIRE) != tag) return fail(NDT_E_STATE, "bounce %d was never published", b);
                    }
                }
                upper = ctx->h_mail[b].count;
                if (upper <= 0) {
                    n_run = b;
                    break;
                }
            }
            level_nodes.push_back(upper);
            // hit points, shadow rays of this bounce, and the rays of the next bounce -- in the same launch as the
            // lighting of the previous bounce, which is waiting for the shadow answers the last trace launch produced
            if (pending_finish >= 0 && fuse_shade) {
                kt->shade_pair(s, ctx->d_blob, ctx->sd, shade_ws(pending_upper, upper), rg, pending_finish, pending_upper, upper);
                pending_finish = -1;
            } else {
                if (pending_finish >= 0) {
                    kt->shade_finish(s, ctx->d_blob, ctx->sd, shade_ws(pending_upper), rg, pending_finish, pending_upper);
                    pending_finish = -1;
                }
                kt->shade_emit(s, ctx->d_blob, ctx->sd, shade_ws(0), rg, b, upper);
            }
            hipLaunchKernelGGL(k_level_step, dim3(1), dim3(64), 0, s, ws, b, n_seg, tag);
            long long next_upper = 2 * upper;           // each node spawns at most two
            if (next_upper > ws.cap) next_upper = ws.cap;
            {
                // ONE launch: shadow rays of bounce b + closest-hit rays of bounce b+1
                TraceJob tj{};
                tj.n_seg = n_seg;
                tj.seg.o = ws.so; tj.seg.v = ws.sv; tj.seg.stride = ws.sh_cap; tj.seg.lim = ws.slim; tj.seg.valid = nullptr;
                tj.seg.out_obj = ws.sobj; tj.seg.out_prim = ws.sprim;
                tj.seg_count = NDT_SEG_COUNTERS(ws, b);
                tj.seg_stride = (upper + 63) & ~63LL;           // sizes the grid only
                tj.dense.o = ws.ray_o; tj.dense.v = ws.ray_v; tj.dense.stride = ws.cap; tj.dense.lim = nullptr;
                tj.dense.valid = ws.depth_left; tj.dense.out_obj = ws.hit_obj; tj.dense.out_prim = ws.hit_prim;
                tj.begin = 0;
                tj.count = next_upper;                          // sizes the grid only
                tj.levels = ws.levels; tj.seg_level = b; tj.dense_level = b + 1;
                if ((rc = traced(tj, "shadow " + std::to_string(b) + " + closest " + std::to_string(b + 1)))) return rc;
            }
            pending_finish = b;
            pending_upper = upper;
        }
        if (pending_finish >= 0) kt->shade_finish(s, ctx->d_blob, ctx->sd, shade_ws(pending_upper), rg, pending_finish, pending_upper);
        // bottom-up colour resolve, deepest bounce first (the primaries last)
        {
            for (int b = n_run; b-- > 0;) {
                long long blocks = (level_nodes[b] + 255) / 256;
                if (blocks > NDT_SHADE_MAX_BLOCKS) blocks = NDT_SHADE_MAX_BLOCKS;
                hipLaunchKernelGGL(k_resolve, dim3((unsigned)blocks), dim3(256), 0, s, ctx->d_blob, ctx->sd, ws, rg.specular, b);
            }
        }
        hipLaunchKernelGGL(k_finish_pixels, dim3((unsigned)((rg.n_primary + 255) / 256)), dim3(256), 0, s, ctx->d_blob, ctx->sd, ws,
                           rg, ctx->dims, (double *)d_rgba, (double *)d_depth);
        if (prof)
            hipExtLaunchKernelGGL(k_frame_done, dim3(1), dim3(64), 0, s, nullptr, ev_end, 0u, ws, n_run, ctx->d_done, tag);
        else
            hipLaunchKernelGGL(k_frame_done, dim3(1), dim3(64), 0, s, ws, n_run, ctx->d_done, tag);
        HIP_TRY(hipGetLastError());
        if (prof && getenv("NDT_HIP_DEBUG_LEVELS")) {
            // the bounce table only feeds the debug output
            HIP_TRY(hipMemcpyAsync(hl, ws.levels, (size_t)(n_run + 1) * sizeof(LevelRange), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        {
            // k_frame_done is the last kernel of the frame: once its tag is here, the image and the record are complete
            const double t_wait = wall_s();
            while (__atomic_load_n(&ctx->h_done[7], __ATOMIC_ACQUIRE) != tag) {
                if (wall_s() - t_wait > 30.0) {
                    HIP_TRY(hipStreamSynchronize(s));
                    if (__atomic_load_n(&ctx->h_done[7], __ATOMIC_ACQUIRE) != tag) return fail(NDT_E_STATE, "the frame never completed");
                }
            }
        }
        if (prof) HIP_TRY(hipEventSynchronize(ev_end));     // the closing kernel has run: its completion is at most microseconds away
        hc[0] = (int)(long long)ctx->h_done[0];
        hc[2] = (int)(long long)ctx->h_done[1];
        hc[3] = (int)(long long)ctx->h_done[2];
        const unsigned long long ref_rays = ctx->h_done[5];
        if (hc[2] != 0) {
            // a pool overflowed somewhere in the frame: grow it and render again
            if (hc[2] & 1) cap *= 2;
            if (hc[2] & 2) {
                sh_cap *= 2;
                if (sh_cap < hc[3]) sh_cap = hc[3];
            }
            continue;
        }
        const long long shadow_total = (long long)ctx->h_done[3];
        const int levels_used = (int)ctx->h_done[4];
        st = ndt_render_stats{};
        st.rays_primary = n_pixels;
        st.rays_secondary = (long long)hc[0] - rg.n_primary;
        st.rays_shadow = shadow_total;
        st.rays_ref_equiv = (long long)ref_rays;
        st.levels = levels_used;
        st.trace_launches = launches;
        st.node_capacity = ws.cap;
        if (prof) {
            float ms = 0;
            for (auto &pr : trace_ev) {
                float m = 0;
                HIP_TRY(hipEventElapsedTime(&m, pr.first, pr.second));
                ms += m;
            }
            st.trace_ms = ms;
            if (getenv("NDT_HIP_DEBUG_LEVELS")) {
                for (int b = 0; b < levels_used; ++b)
                    fprintf(stderr, "ndt_hip: bounce %d: %lld nodes, %lld shadow rays\n", b, hl[b].count, hl[b].n_shadow);
                unsigned long long d[160];
                if (hipMemcpy(d, ws.dbg, sizeof(d), hipMemcpyDeviceToHost) != hipSuccess) d[4] = 0;
                if (shade_probe >= 0 && ws.shade_log) {
                    std::vector<unsigned int> log((size_t)2 * NDT_SHADE_LOG_WAVES);
                    if (hipMemcpy(log.data(), ws.shade_log, log.size() * sizeof(unsigned int), hipMemcpyDeviceToHost) == hipSuccess) {
                        unsigned int t0 = 0;
                        bool any = false;
                        for (int w = 0; w < NDT_SHADE_LOG_WAVES; ++w)
                            if (log[2 * w + 1] && (!any || (int)(log[2 * w] - t0) < 0)) {
                                t0 = log[2 * w];
                                any = true;
                            }
                        for (int part = 0; part < 2; ++part) {
                            // part 0: lighting (shade_finish) wavefronts, part 1: shading (shade_emit) wavefronts
                            int hist[48] = { 0 }, n_w = 0;
                            double sum = 0, longest = 0, last_start = 0, last_end = 0;
                            for (long long w = 0; w < NDT_SHADE_LOG_WAVES; ++w) {
                                const bool lighting = w < shade_probe_finish_waves;
                                if (!log[2 * w + 1] || lighting != (part == 0)) continue;
                                const double st_us = (log[2 * w] - t0) / 100.0, dur = (log[2 * w + 1] - log[2 * w]) / 100.0;
                                ++n_w;
                                sum += dur;
                                if (dur > longest) longest = dur;
                                if (st_us > last_start) last_start = st_us;
                                if (st_us + dur > last_end) last_end = st_us + dur;
                                const int bin = (int)(dur / 4.0);
                                ++hist[bin > 47 ? 47 : bin];
                            }
                            if (!n_w) continue;
                            std::string line;
                            for (int bin = 0; bin < 48; ++bin)
                                if (hist[bin]) {
                                    char buf[48];
                                    snprintf(buf, sizeof buf, " %d-%d:%d", bin * 4, bin * 4 + 4, hist[bin]);
                                    line += buf;
                                }
                            fprintf(stderr, "ndt_hip: shade launch %d, %s: %d wavefronts, mean life %.1f us, longest %.1f us, last start at %.1f us, last end at %.1f us; lives per 4 us:%s\n",
                                    shade_probe, part == 0 ? "lighting" : "shading", n_w, sum / n_w, longest, last_start, last_end, line.c_str());
                        }
                    }
                }
                if (getenv("NDT_HIP_EXIT_PROBE")) {
                    // the life of every wavefront of every trace launch: when the queue runs dry (first exit), how long the
                    // rest keeps going, and how much of that is the last wavefront's last batch
                    std::vector<unsigned int> log((size_t)NDT_EXIT_LOG_LAUNCHES * NDT_EXIT_LOG_WORDS);
                    if (hipMemcpy(log.data(), ws.exit_log, log.size() * sizeof(unsigned int), hipMemcpyDeviceToHost) == hipSuccess)
                        for (int l = 0; l < NDT_EXIT_LOG_LAUNCHES && l < launches; ++l) {
                            const unsigned int *q = log.data() + (size_t)l * NDT_EXIT_LOG_WORDS;
                            unsigned int t0 = 0;
                            int n_w = 0;
                            for (int w = 0; w < NDT_EXIT_LOG_WORDS / 4; ++w)
                                if (q[4 * w + 2]) {
                                    if (!n_w || (int)(q[4 * w] - t0) < 0) t0 = q[4 * w];
                                    ++n_w;
                                }
                            int hist[64] = { 0 };
                            double first = 1e30, last = 0, last_batch = 0, start_spread = 0;
                            int simd_of_wave[12][4] = { { 0 } };        // workgroup wavefront w (768-lane workgroups) -> SIMD it ran on
                            for (int w = 0; w < NDT_EXIT_LOG_WORDS / 4; ++w)
                                if (q[4 * w + 2]) {
                                    const double st_us = (q[4 * w] - t0) / 100.0, ex_us = (q[4 * w + 2] - t0) / 100.0;
                                    ++simd_of_wave[w % 12][(q[4 * w + 3] >> 4) & 3];
                                    if (st_us > start_spread) start_spread = st_us;
                                    if (ex_us < first) first = ex_us;
                                    if (ex_us > last) {
                                        last = ex_us;
                                        last_batch = (q[4 * w + 2] - q[4 * w + 1]) / 100.0;
                                    }
                                    const int bin = (int)(ex_us / 16.0);
                                    ++hist[bin > 63 ? 63 : bin];
                                }
                            std::string line;
                            for (int bin = 0; bin < 64; ++bin)
                                if (hist[bin]) {
                                    char buf[48];
                                    snprintf(buf, sizeof buf, " %d-%d:%d", bin * 16, bin * 16 + 16, hist[bin]);
                                    line += buf;
                                }
                            if (l == 0) {
                                std::string m;
                                for (int w = 0; w < 12; ++w) {
                                    char buf[64];
                                    snprintf(buf, sizeof buf, " w%d:%d/%d/%d/%d", w, simd_of_wave[w][0], simd_of_wave[w][1], simd_of_wave[w][2], simd_of_wave[w][3]);
                                    m += buf;
                                }
                                fprintf(stderr, "ndt_hip: SIMD 0/1/2/3 of the workgroup's wavefronts (768-lane workgroups):%s\n", m.c_str());
                            }
                            fprintf(stderr, "ndt_hip: trace launch %d: %d wavefronts start within %.1f us; first out of work at %.1f us, last at %.1f us (its last batch: %.1f us); exits per 16 us:%s\n",
                                    l, n_w, start_spread, first, last, last_batch, line.c_str());
                        }
                }
                if (d[4]) {
                    // NDT_PHASE_TIMING builds only (make -C ndt_amd/csrc timing)
                    fprintf(stderr, "ndt_hip: wave cycles T %llu G %llu I %llu list-end %llu prologue %llu outside %llu over %llu waves\n", d[0], d[1], d[2], d[3], d[5], d[6], d[4]);
                    fprintf(stderr, "ndt_hip: per-ray counts over %llu rays: node visits %llu, face gates %llu (pass %llu), item gates %llu (pass %llu), isect hits %llu\n",
                            d[14], d[8], d[9], d[10], d[11], d[12], d[13]);
                    fprintf(stderr, "ndt_hip: batch time inside trace_kd (100 MHz wall clock): closest max %.1f us mean %.1f us, shadow max %.1f us mean %.1f us\n",
                            d[40] / 100.0, d[44] ? d[42] / 100.0 / d[44] : 0.0, d[41] / 100.0, d[45] ? d[43] / 100.0 / d[45] : 0.0);
                    fprintf(stderr, "ndt_hip: per-ray maxima: %llu node visits, %llu gates, %llu intersections; per-batch maxima: %llu T, %llu G, %llu I iterations\n",
                            d[46], d[47], d[48], d[49], d[50], d[51]);
                    if (d[58])
                        fprintf(stderr, "ndt_hip: shade_emit per wavefront (wall-clock ticks, mean over %llu): load+isect %.0f, light tests %.0f, segment reserve %.0f, shadow stores %.0f, spawn %.0f; slowest wavefront %llu\n",
                                d[58], (double)d[52] / d[58], (double)d[53] / d[58], (double)d[54] / d[58], (double)d[55] / d[58], (double)d[56] / d[58], d[59]);
                    for (int kind = 0; kind < 2; ++kind) {
                        const unsigned long long *q = d + 16 + 8 * kind;
                        fprintf(stderr, "ndt_hip: loop occupancy (%s rays): T %.1f%% of %llu iters, G %.1f%% of %llu, I %.1f%% of %llu\n",
                                kind ? "shadow" : "closest", q[0] ? 100.0 * q[1] / (64.0 * q[0]) : 0.0, q[0],
                                q[2] ? 100.0 * q[3] / (64.0 * q[2]) : 0.0, q[2], q[4] ? 100.0 * q[5] / (64.0 * q[4]) : 0.0, q[4]);
                        if (q[4])
                            fprintf(stderr, "ndt_hip:    I iterations execute %.2f primitive types on average; the commonest type holds %.1f of %.1f active lanes\n",
                                    (double)q[6] / q[4], (double)q[7] / q[4], (double)q[5] / q[4]);
                    }
                }
                for (size_t i = 0; i < trace_ev.size(); ++i) {
                    float m = 0;
                    (void)hipEventElapsedTime(&m, trace_ev[i].first, trace_ev[i].second);
                    fprintf(stderr, "ndt_hip: trace launch %zu: %.3f ms (%s)\n", i, m, trace_dbg[i].c_str());
                }
            }
            float fm = 0;
            HIP_TRY(hipEventElapsedTime(&fm, ev_begin, ev_end));
            st.frame_ms = fm;
        }
        return NDT_OK;
    }
    return fail(NDT_E_NOMEM, "ray-tree workspace kept overflowing");
}

// ------------------------------------------------------------------ Whitted's recursive anti-aliasing (ndt.c:655-733, 1039-1087)
//
// The reference recurses per pixel; here the recursion tree is walked level by level over all
// pixels at once.  A task = one call of recursive_resample: a square of side `step` at (x, y)
// with its four corner colours.  Level L holds the tasks with step 2^-L; each renders 5 new
// samples (through render_pass in list mode), forms the four quarter averages, and spawns a
// child task for every quarter whose corners differ by more than the threshold.  Results flow
// back up: a task's colour is the average of its quarters, a quarter being replaced by its
// child's colour where one was spawned (ndt.c:685-705).  Averages use image_avg_dbl_pixels4's
// operation order (image.c:1175-1197), so every colour equals the recursion's.

struct AaTask {
    double x, y;                // top-left corner, in pixels of the first-pass image
    double p[4][4];             // corner colours p1..p4 (top-left, top-right, bottom-left, bottom-right), rgba
    double sp[4][4];            // quarter colours sp1..sp4
    long long parent;           // level 0: output pixel index; deeper: task index in the level above
    int quad;                   // which quarter of the parent this task refines
    int _pad;
};

// One slot of a device-side list for every lane that wants one, with ONE atomic per wavefront (a counter serves ~150
// returning atomics per us: a list appended to by every thread of a 2-million-thread launch queues for milliseconds).
// Every lane of the wavefront has to call it.
__device__ __forceinline__ int wave_append(int *counter, bool want)
{
    const unsigned long long vote = __ballot(want);
    if (vote == 0ull) return 0;
    const int lane = __lane_id(), leader = __ffsll((long long)vote) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(vote));
    base = __shfl(base, leader, 64);
    return base + __popcll(vote & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ void aa_avg4(const double *p1, const double *p2, const double *p3, const double *p4, double *avg, double *var)
{
    for (int c = 0; c < 4; ++c) avg[c] = (p1[c] + p2[c] + p3[c] + p4[c]) / 4;
    if (var) {
        double v = 0;
        for (int c = 0; c < 4; ++c)
            v += fabs(avg[c] - p1[c]) + fabs(avg[c] - p2[c]) + fabs(avg[c] - p3[c]) + fabs(avg[c] - p4[c]);
        *var = v;
    }
}

// resample_pixel (ndt.c:709-733): average of the four corners; pixels over the threshold become level-0 tasks
__global__ void k_aa_seed(const double *pass1, int width, int rows, int row_begin, int row_step, int row_pair, double threshold,
                          double *out, AaTask *tasks, int *counter)
{
    const long long idx_raw = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = idx_raw < (long long)rows * width;
    const long long idx = in_range ? idx_raw : 0;
    const int l = (int)(idx / width), i = (int)(idx % width);
    const int r0 = row_pair ? 2 * l : l, r1 = r0 + 1;
    const long long w1 = width + 1;
    const double *p1 = pass1 + (r0 * w1 + i) * 4, *p2 = p1 + 4;
    const double *p3 = pass1 + (r1 * w1 + i) * 4, *p4 = p3 + 4;
    double clr[4], var = 0.0;
    aa_avg4(p1, p2, p3, p4, clr, &var);
    const bool refine = in_range && var > threshold;
    const int t = wave_append(counter, refine);
    if (!in_range) return;
    if (refine) {
        AaTask &T = tasks[t];
        T.x = i;
        T.y = row_begin + l * row_step;
        for (int c = 0; c < 4; ++c) { T.p[0][c] = p1[c]; T.p[1][c] = p2[c]; T.p[2][c] = p3[c]; T.p[3][c] = p4[c]; }
        T.parent = idx;
        T.quad = -1;
    } else {
        for (int c = 0; c < 4; ++c) out[idx * 4 + c] = clr[c];
    }
}

// the five new samples of a task: centre, top middle, left edge, right edge, bottom middle (ndt.c:669-678)
__global__ void k_aa_samples(const AaTask *tasks, int n_tasks, double step, double *samples)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tasks) return;
    const double x = tasks[t].x, y = tasks[t].y, hs = step / 2;
    double *q = samples + (long long)t * 10;
    q[0] = x + hs;   q[1] = y + hs;
    q[2] = x + hs;   q[3] = y;
    q[4] = x;        q[5] = y + hs;
    q[6] = x + step; q[7] = y + hs;
    q[8] = x + hs;   q[9] = y + step;
}

// quarter averages and children (ndt.c:680-703)
__global__ void k_aa_split(AaTask *tasks, int n_tasks, double step, double threshold, const double *colours,
                           AaTask *next, int *next_counter)
{
    const int t_raw = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = t_raw < n_tasks;
    const int t = in_range ? t_raw : 0;          // (lanes past the end go through the motions on task 0 and write nothing)
    AaTask &T = tasks[t];
    const double *p5 = colours + (long long)t * 20, *p6 = p5 + 4, *p7 = p5 + 8, *p8 = p5 + 12, *p9 = p5 + 16;
    const double hs = step / 2;
    // corners of the four quarters, in image_avg_dbl_pixels4's argument order
    const double *qa[4][4] = { { T.p[0], p6, p7, p5 }, { T.p[1], p6, p8, p5 }, { T.p[2], p9, p7, p5 }, { T.p[3], p9, p8, p5 } };
    // ... and in the order the recursive call receives them as p1..p4 (ndt.c:687, 692, 697, 702)
    const double *qc[4][4] = { { T.p[0], p6, p7, p5 }, { p6, T.p[1], p5, p8 }, { p7, p5, T.p[2], p9 }, { p5, p8, p9, T.p[3] } };
    const double qx[4] = { T.x, T.x + hs, T.x, T.x + hs }, qy[4] = { T.y, T.y, T.y + hs, T.y + hs };
    for (int k = 0; k < 4; ++k) {
        double var = 0.0, avg[4];
        aa_avg4(qa[k][0], qa[k][1], qa[k][2], qa[k][3], avg, &var);
        if (in_range)
            for (int ch = 0; ch < 4; ++ch) T.sp[k][ch] = avg[ch];
        const bool refine = in_range && var > threshold;
        const int c = wave_append(next_counter, refine);
        if (refine) {
            AaTask &C = next[c];
            C.x = qx[k];
            C.y = qy[k];
            for (int m = 0; m < 4; ++m)
                for (int ch = 0; ch < 4; ++ch) C.p[m][ch] = qc[k][m][ch];
            C.parent = t;
            C.quad = k;
        }
    }
}

// a task's colour goes to the quarter of its parent it refines, or (level 0) to its pixel.
// leaf != 0: the recursion's cut-off (ndt.c:663-666): the colour is the average of the corners
__global__ void k_aa_resolve(const AaTask *tasks, int n_tasks, int leaf, AaTask *parents, double *out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tasks) return;
    const AaTask &T = tasks[t];
    double res[4];
    if (leaf) aa_avg4(T.p[0], T.p[1], T.p[2], T.p[3], res, nullptr);
    else aa_avg4(T.sp[0], T.sp[1], T.sp[2], T.sp[3], res, nullptr);
    double *dst = parents ? parents[T.parent].sp[T.quad] : out + T.parent * 4;
    for (int c = 0; c < 4; ++c) dst[c] = res[c];
}

// Scratch buffers of the multi-pass renderers (anti-aliasing levels, sample rounds, anaglyph eyes).  The
// requests of a frame come in the same order every frame, so the k-th request reuses the k-th
// allocation of the context's pool (grown when too small) instead of a hipMalloc / hipFree pair, each of
// which synchronises the device.
struct AaBuffers {
    ndt_hip_ctx *ctx;
    size_t next = 0;
    explicit AaBuffers(ndt_hip_ctx *c) : ctx(c) {}
    template <typename T> int get(T **ptr, size_t count)
    {
        const size_t bytes = (count > 0 ? count : 1) * sizeof(T);
        if (next == ctx->pool.size()) ctx->pool.push_back({ nullptr, 0 });
        auto &slot = ctx->pool[next++];
        if (slot.second < bytes) {
            if (slot.first) {
                (void)hipStreamSynchronize(ctx->stream);
                (void)hipFree(slot.first);
                slot = { nullptr, 0 };
            }
            const size_t want = bytes + bytes / 4;          // some head room: frame-to-frame counts vary
            void *q = nullptr;
            hipError_t e = hipMalloc(&q, want);
            if (e != hipSuccess) return fail(NDT_E_NOMEM, "hipMalloc of %zu bytes: %s", want, hipGetErrorString(e));
            slot = { q, want };
        }
        *ptr = (T *)slot.first;
        return NDT_OK;
    }
};

static void add_stats(ndt_render_stats &acc, const ndt_render_stats &st)
{
    acc.rays_primary += st.rays_primary;
    acc.rays_secondary += st.rays_secondary;
    acc.rays_shadow += st.rays_shadow;
    acc.rays_ref_equiv += st.rays_ref_equiv;
    if (st.levels > acc.levels) acc.levels = st.levels;
    acc.trace_launches += st.trace_launches;
    acc.trace_ms += st.trace_ms;
    acc.frame_ms += st.frame_ms;
    if (st.node_capacity > acc.node_capacity) acc.node_capacity = st.node_capacity;
}

static int render_antialiased(ndt_hip_ctx *ctx, const ndt_render_params *p, void *d_rgba, ndt_render_stats &total)
{
    hipStream_t s = ctx->stream;
    const bool prof = p->profile != 0;
    const int W = p->width, H = p->height;
    const int rows = ndt_hip_shard_rows(H, p->row_begin, p->row_step);
    AaBuffers buf(ctx);
    int rc;
    // ---- first pass: the corner rows this shard touches, (W+1) wide (ndt.c:919-976)
    RenderGeom g1{};
    g1.width = W + 1;
    g1.row_begin = p->row_begin;
    g1.row_step = p->row_step;
    g1.row_pair = p->row_step > 1 ? 1 : 0;
    g1.rows = g1.row_pair ? 2 * rows : rows + 1;
    g1.tiles_x = (g1.width + 7) / 8;
    g1.tiles_y = (g1.rows + 7) / 8;
    const long long n1 = (long long)g1.tiles_x * g1.tiles_y * 64;
    if (n1 > 0x3fffffffLL) return fail(NDT_E_UNSUPPORTED, "image too large for one call");
    g1.n_primary = (int)n1;
    g1.max_depth = p->max_optic_depth;
    g1.specular = p->specular ? 1 : 0;
    g1.img_w = W + 1;
    g1.img_h = H + 1;
    g1.aspect_w = W;
    g1.aspect_h = H;
    g1.eye = 1;
    double *pass1 = nullptr;
    if ((rc = buf.get(&pass1, (size_t)g1.rows * g1.width * 4))) return rc;
    ndt_render_stats st{};
    if ((rc = render_pass(ctx, g1, prof, pass1, st))) return rc;
    add_stats(total, st);

    // ---- second pass
    const double threshold = p->aa_diff / 255.0;
    const long long n_out = (long long)rows * W;
    int *counters = nullptr;            // one per level
    const int max_levels = (p->aa_depth > 0 ? p->aa_depth : 0) + 2;
    if ((rc = buf.get(&counters, (size_t)max_levels + 1))) return rc;
    HIP_TRY(hipMemsetAsync(counters, 0, ((size_t)max_levels + 1) * sizeof(int), s));
    std::vector<AaTask *> level_tasks;
    std::vector<int> level_count;
    AaTask *t0 = nullptr;
    if ((rc = buf.get(&t0, (size_t)n_out))) return rc;
    hipLaunchKernelGGL(k_aa_seed, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, pass1, W, rows, p->row_begin, p->row_step,
                       g1.row_pair, threshold, (double *)d_rgba, t0, counters);
    int n_tasks = 0;
    HIP_TRY(hipMemcpyAsync(&n_tasks, counters, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    total.pixels_resampled = n_tasks;
    level_tasks.push_back(t0);
    level_count.push_back(n_tasks);
    double step = 1.0;
    int leaf_level = -1;
    for (int L = 0; n_tasks > 0; ++L) {
        if (L + 1 > max_levels) return fail(NDT_E_STATE, "anti-aliasing recursion deeper than expected");
        // recursive_resample's cut-off (ndt.c:663)
        if (p->aa_depth <= 0 || step < 1.0 / (double)(2 << (p->aa_depth - 1))) {
            leaf_level = L;
            break;
        }
        double *samples = nullptr, *colours = nullptr;
        AaTask *next = nullptr;
        if ((rc = buf.get(&samples, (size_t)n_tasks * 10))) return rc;
        if ((rc = buf.get(&colours, (size_t)n_tasks * 20))) return rc;
        if ((rc = buf.get(&next, (size_t)n_tasks * 4))) return rc;
        hipLaunchKernelGGL(k_aa_samples, dim3((unsigned)((n_tasks + 255) / 256)), dim3(256), 0, s, level_tasks[L], n_tasks, step, samples);
        RenderGeom gs{};
        gs.samples = samples;
        gs.n_samples = 5 * n_tasks;
        gs.n_primary = (gs.n_samples + 63) & ~63;
        gs.width = gs.n_samples;
        gs.rows = 1;
        gs.max_depth = p->max_optic_depth;
        gs.specular = p->specular ? 1 : 0;
        gs.img_w = W + 1;
        gs.img_h = H + 1;
        gs.aspect_w = W;
        gs.aspect_h = H;
        gs.eye = 1;
        if ((rc = render_pass(ctx, gs, prof, colours, st))) return rc;
        add_stats(total, st);
        total.aa_samples += gs.n_samples;
        hipLaunchKernelGGL(k_aa_split, dim3((unsigned)((n_tasks + 255) / 256)), dim3(256), 0, s, level_tasks[L], n_tasks, step, threshold,
                           colours, next, counters + L + 1);
        int n_next = 0;
        HIP_TRY(hipMemcpyAsync(&n_next, counters + L + 1, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        level_tasks.push_back(next);
        level_count.push_back(n_next);
        n_tasks = n_next;
        step /= 2;
    }
    // ---- colours back up the tree, deepest level first
    for (int L = (int)level_tasks.size() - 1; L >= 0; --L) {
        const int n = level_count[L];
        if (n <= 0) continue;
        hipLaunchKernelGGL(k_aa_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, level_tasks[L], n, L == leaf_level ? 1 : 0,
                           L > 0 ? level_tasks[L - 1] : (AaTask *)nullptr, (double *)d_rgba);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return NDT_OK;
}

// ------------------------------------------------------------------ -n samples > 1 (ndt.c:470-568)
//
// Jittered samples inside the pixel + a lens sample per ray, the adaptive loop on top: at least
// `samples` samples per pixel, then more while the running mean still moves by more than 1/256
// (at most 10000).  The reference draws from one global drand48 stream in pixel order, which no
// parallel renderer can follow; here every (pixel, sample) has its own counter-based stream, so the
// image is reproducible and independent of sharding, and parity with the reference is statistical
// (tests compare against the oracle, which follows the reference's stream exactly).
// Round r renders sample r of every pixel that is still sampling, through the pipeline in list mode.

// the random stream of sample `round` of image pixel `pixel`
__device__ __forceinline__ unsigned long long ns_sample_key(unsigned long long pixel, unsigned int round)
{
    return ndt_rng_mix(pixel * 0x100000001b3ull + round);
}

// sample r of the active pixels: (i + dx, j - dy) and the lens offsets (ndt.c:505-514, 527-541)
// (`per` consecutive samples per pixel in one pass: sample a*per + r is the pixel's sample number round + r)
__global__ void k_ns_samples(const int *active, int n_active, int per, int width, int row_begin, int row_step, unsigned int round0,
                             double aperture, int jitter, double *samples, unsigned long long *keys)
{
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= (long long)n_active * per) return;
    const int pix = active[a / per];
    const unsigned int round = round0 + (unsigned int)(a % per);
    const int l = pix / width, i = pix % width;
    const int j = row_begin + l * row_step;
    const unsigned long long id = (unsigned long long)j * (unsigned long long)width + (unsigned long long)i;
    const unsigned long long key = ns_sample_key(id, round);
    keys[a] = key;
    double dx = 0.0, dy = 0.0, ax = 0.0, ay = 0.0;
    if (jitter) {       // -n > 1 only (ndt.c:505, 528); with -n 1 the area lights are all that is random
        dx = ndt_rng_uniform(key, 1000);
        dy = ndt_rng_uniform(key, 1001);
        unsigned int k = 1002;
        do {        // reject samples outside the unit disk
            ax = 2 * ndt_rng_uniform(key, k) - 1.0;
            ay = 2 * ndt_rng_uniform(key, k + 1) - 1.0;
            k += 2;
        } while (ax * ax + ay * ay > 1.0 && k < 1064);
    }
    double *q = samples + 4ll * a;
    q[0] = i + dx;          // x = orig_x + dx/width
    q[1] = j - dy;          // y = orig_y + dy/height, and y grows upwards
    q[2] = ax * aperture;
    q[3] = ay * aperture;
}

// get_pixel_color's loop body after the sample has been traced (ndt.c:553-567), and its continuation test
// A pass may have rendered `per` samples ahead for every pixel; they are consumed one by one exactly as the
// loop would, and the ones after the loop's exit are dropped (they were speculation: fewer, fuller passes).
__global__ void k_ns_accumulate(const int *active, int n_active, int per, const double *colours, unsigned int round0, int min_samples,
                                double *acc, int *taken, int *next, int *next_count)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = a < n_active;
    const int pix = in_range ? active[a] : 0;
    double *t = acc + 5ll * pix;            // t_clr rgba + clr_diff
    double clr_diff = in_range ? t[4] : 0.0;
    bool go_on = in_range;
    int used = 0;
    for (int r = 0; r < per && go_on; ++r) {
        const double *l = colours + 4ll * ((long long)a * per + r);
        const int i = (int)round0 + r;
        if (i > 1) {
            const double dr = fabs(t[0] / (i - 1) - (t[0] + l[0]) / i);
            const double dg = fabs(t[1] / (i - 1) - (t[1] + l[1]) / i);
            const double db = fabs(t[2] / (i - 1) - (t[2] + l[2]) / i);
            const double gb = (dg > db) ? dg : db;
            clr_diff = (dr > gb) ? dr : gb;
        }
        t[0] += l[0]; t[1] += l[1]; t[2] += l[2]; t[3] += l[3];
        ++used;
        const int done = i + 1;
        go_on = done < min_samples || (done < 10000 && clr_diff > 1.0 / 256.0);
    }
    if (in_range) {
        t[4] = clr_diff;
        taken[pix] += used;
    }
    const int slot = wave_append(next_count, go_on);
    if (go_on) next[slot] = pix;
}

__global__ void k_ns_init(double *acc, int *active, int *taken, long long n_pixels)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    double *t = acc + 5 * i;
    t[0] = t[1] = t[2] = t[3] = 0.0;
    t[4] = 256.0;
    active[i] = (int)i;
    taken[i] = 0;
}

__global__ void k_ns_finish(const double *acc, const int *taken, double *rgba, long long n_pixels, unsigned long long *used_total)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long used = 0;
    if (i < n_pixels) {
        const int n = taken[i] > 0 ? taken[i] : 1;
        for (int c = 0; c < 4; ++c) rgba[4 * i + c] = acc[5 * i + c] / n;
        used = (unsigned long long)taken[i];
    }
    for (int d = 32; d > 0; d >>= 1) used += __shfl_down(used, d, 64);
    if ((threadIdx.x & 63) == 0 && used) atomicAdd(used_total, used);
}

static int render_sampled(ndt_hip_ctx *ctx, const ndt_render_params *p, void *d_rgba, ndt_render_stats &total)
{
    hipStream_t s = ctx->stream;
    const int W = p->width, H = p->height;
    const int rows = ndt_hip_shard_rows(H, p->row_begin, p->row_step);
    const long long n_pixels = (long long)rows * W;
    if (n_pixels > 0x3fffffffLL) return fail(NDT_E_UNSUPPORTED, "image too large for one call");
    AaBuffers buf(ctx);
    int rc;
    double *acc = nullptr, *samples = nullptr, *colours = nullptr;
    unsigned long long *keys = nullptr;
    int *list[2] = { nullptr, nullptr }, *taken = nullptr, *counter = nullptr;
    if ((rc = buf.get(&acc, (size_t)n_pixels * 5))) return rc;
    if ((rc = buf.get(&list[0], (size_t)n_pixels))) return rc;
    if ((rc = buf.get(&list[1], (size_t)n_pixels))) return rc;
    if ((rc = buf.get(&taken, (size_t)n_pixels))) return rc;
    if ((rc = buf.get(&counter, 1))) return rc;
    const unsigned g_all = (unsigned)((n_pixels + 255) / 256);
    hipLaunchKernelGGL(k_ns_init, dim3(g_all), dim3(256), 0, s, acc, list[0], taken, n_pixels);
    int n_active = (int)n_pixels;
    // samples per pixel and pass: the first `samples` are certain to be needed; after that the loop may stop at
    // any sample, so passes speculate further ahead the fewer pixels are left (about 4 M primaries per pass)
    const long long per_pass = 4ll << 20;
    size_t cap_samples = 0;
    int flip = 0;
    for (unsigned int round = 0; n_active > 0 && round < 10000;) {
        long long per = round < (unsigned int)p->samples ? (long long)p->samples - round : per_pass / n_active;
        if (per > per_pass / n_active) per = per_pass / n_active;
        // ... but never more than have been taken already: the waste stays below a factor of two
        if (round >= (unsigned int)p->samples && per > (long long)(round < 2 ? 1 : round)) per = round < 2 ? 1 : round;
        if (per < 1) per = 1;
        if (per > 64) per = 64;
        if (round + per > 10000) per = 10000 - round;
        const long long n_s = (long long)n_active * per;
        if ((size_t)n_s > cap_samples) {
            cap_samples = (size_t)n_s;
            if ((rc = buf.get(&samples, cap_samples * 4))) return rc;
            if ((rc = buf.get(&colours, cap_samples * 4))) return rc;
            if ((rc = buf.get(&keys, cap_samples + 64))) return rc;
        }
        const unsigned g_act = (unsigned)((n_active + 255) / 256);
        hipLaunchKernelGGL(k_ns_samples, dim3((unsigned)((n_s + 255) / 256)), dim3(256), 0, s, list[flip], n_active, (int)per, W,
                           p->row_begin, p->row_step, round, ctx->aperture_radius, p->samples > 1 ? 1 : 0, samples, keys);
        RenderGeom gs{};
        gs.samples = samples;
        gs.n_samples = (int)n_s;
        gs.n_primary = (int)((n_s + 63) & ~63LL);
        gs.width = (int)n_s;
        gs.rows = 1;
        gs.max_depth = p->max_optic_depth;
        gs.specular = p->specular ? 1 : 0;
        gs.img_w = W;
        gs.img_h = H;
        gs.aspect_w = W;
        gs.aspect_h = H;
        gs.eye = 1;
        gs.lens = 1;
        gs.raw_samples = 1;
        gs.sample_keys = keys;
        ndt_render_stats st{};
        if ((rc = render_pass(ctx, gs, p->profile != 0, colours, st))) return rc;
        add_stats(total, st);
        HIP_TRY(hipMemsetAsync(counter, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_ns_accumulate, dim3(g_act), dim3(256), 0, s, list[flip], n_active, (int)per, colours, round, p->samples,
                           acc, taken, list[flip ^ 1], counter);
        HIP_TRY(hipMemcpyAsync(&n_active, counter, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        flip ^= 1;
        round += (unsigned int)per;
    }
    unsigned long long *used_total = nullptr, used_host = 0;
    if ((rc = buf.get(&used_total, 1))) return rc;
    HIP_TRY(hipMemsetAsync(used_total, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_ns_finish, dim3(g_all), dim3(256), 0, s, acc, taken, (double *)d_rgba, n_pixels, used_total);
    HIP_TRY(hipMemcpyAsync(&used_host, used_total, sizeof(used_host), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    total.aa_samples = (long long)used_host;       // samples the adaptive loop consumed (the rays_* counts include the speculation)
    return NDT_OK;
}

// true anaglyph (ndt.c:643-647): red = luminance of the left eye's colour, blue = of the right eye's
__global__ void k_anaglyph(const double *left, const double *right, double *out, long long n_pixels)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const double *l = left + 4 * i, *r = right + 4 * i;
    out[4 * i + 0] = 0.299 * l[0] + 0.587 * l[1] + 0.114 * l[2];
    out[4 * i + 1] = 0;
    out[4 * i + 2] = 0.299 * r[0] + 0.587 * r[1] + 0.114 * r[2];
    out[4 * i + 3] = 1.0;
}

extern "C" int ndt_hip_render_depth_device(ndt_hip_ctx *ctx, const ndt_render_params *p, void *d_rgba, void *d_depth,
                                           ndt_render_stats *stats)
{
    if (!ctx || !p || !d_rgba) return fail(NDT_E_INVALID, "NULL argument");
    if (!ctx->have_scene) return fail(NDT_E_STATE, "no scene uploaded");
    if (p->samples < 1) return fail(NDT_E_INVALID, "samples=%d", p->samples);
    const bool stochastic = p->samples > 1 || ctx->has_area_lights;
    if (stochastic && (p->recursive_aa || p->stereo != NDT_STEREO_MONO || d_depth || ctx->cam_type != 0))
        return fail(NDT_E_UNSUPPORTED, "samples > 1 and area lights are implemented for the mono planar camera without recursive anti-aliasing or a depth map");
    if (p->samples > 1 && ctx->aperture_radius != 0.0 && !ctx->have_local_axes)
        return fail(NDT_E_INVALID, "depth of field needs the camera's local axes (camera.h:69-71) in the flat scene");
    if (p->width < 1 || p->height < 1 || p->row_step < 1 || p->row_begin < 0) return fail(NDT_E_INVALID, "bad geometry");
    if (p->stereo < NDT_STEREO_MONO || p->stereo > NDT_STEREO_HIDEF) return fail(NDT_E_UNSUPPORTED, "stereo mode %d", p->stereo);
    if (p->stereo != NDT_STEREO_MONO && !ctx->have_eyes) return fail(NDT_E_INVALID, "stereo needs leftEye / rightEye (camera.h:60-61) in the flat scene");
    for (int k = 0; k < 4; ++k)
        if (p->reserved[k] != 0) return fail(NDT_E_INVALID, "reserved render parameter set");
    if (p->recursive_aa && ctx->aperture_radius != 0.0)
        return fail(NDT_E_UNSUPPORTED, "recursive anti-aliasing with aperture radius %g samples the lens with drand48 (ndt.c:528): not reproducible", ctx->aperture_radius);
    if (p->recursive_aa && (p->aa_diff < 0 || p->aa_depth > 24)) return fail(NDT_E_INVALID, "bad anti-aliasing parameters");
    if (p->recursive_aa && (p->stereo != NDT_STEREO_MONO || d_depth || ctx->cam_type != 0))
        return fail(NDT_E_UNSUPPORTED, "recursive anti-aliasing is implemented for the mono planar camera without a depth map");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int rows = ndt_hip_shard_rows(p->height, p->row_begin, p->row_step);
    ndt_render_stats st{};
    if (rows == 0) {
        if (stats) *stats = st;
        return NDT_OK;
    }
    const long long n_pixels = (long long)rows * p->width;
    if (p->max_optic_depth <= 0) {
        // get_ray_color returns black without tracing (ndt.c:340); averages of black are black
        hipLaunchKernelGGL(k_fill_black, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, s, (double *)d_rgba, n_pixels);
        if (d_depth) HIP_TRY(hipMemsetAsync(d_depth, 0, (size_t)n_pixels * sizeof(double), s));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
        if (stats) *stats = st;
        return NDT_OK;
    }
    int rc;
    // the reference's switch: -a with depth >= 0 and diff < 256 resamples, otherwise the first pass is copied (ndt.c:1040)
    if (p->recursive_aa) {
        rc = render_antialiased(ctx, p, d_rgba, st);
    } else if (stochastic) {
        rc = render_sampled(ctx, p, d_rgba, st);
    } else {
        RenderGeom rg{};
        rg.width = p->width;
        rg.height = p->height;
        rg.row_begin = p->row_begin;
        rg.row_step = p->row_step;
        rg.rows = rows;
        rg.tiles_x = (rg.width + 7) / 8;
        rg.tiles_y = (rg.rows + 7) / 8;
        const long long n_primary = (long long)rg.tiles_x * rg.tiles_y * 64;
        if (n_primary > 0x3fffffffLL) return fail(NDT_E_UNSUPPORTED, "image too large for one call");
        rg.n_primary = (int)n_primary;
        rg.max_depth = p->max_optic_depth;
        rg.specular = p->specular ? 1 : 0;
        rg.img_w = p->width;
        rg.img_h = p->height;
        rg.aspect_w = p->width;
        rg.aspect_h = p->height;
        rg.eye = 1;
        if (p->stereo == NDT_STEREO_ANAGLYPH) {
            // two full renders, one per eye (ndt.c:636-647); the depth map is the left eye's
            AaBuffers buf(ctx);
            double *left = nullptr, *right = nullptr;
            if ((rc = buf.get(&left, (size_t)n_pixels * 4))) return rc;
            if ((rc = buf.get(&right, (size_t)n_pixels * 4))) return rc;
            ndt_render_stats one{};
            rg.eye = 0;
            if ((rc = render_pass(ctx, rg, p->profile != 0, left, one, d_depth))) return rc;
            add_stats(st, one);
            rg.eye = 2;
            if ((rc = render_pass(ctx, rg, p->profile != 0, right, one))) return rc;
            add_stats(st, one);
            hipLaunchKernelGGL(k_anaglyph, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, s, left, right, (double *)d_rgba, n_pixels);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s));
            rc = NDT_OK;
        } else {
            rg.stereo = p->stereo;
            if (p->stereo == NDT_STEREO_HIDEF) {
                // frame packing (ndt.c:614-631, 927-928): the aspect is width/1080 and the 45 blank lines between the
                // eyes stay black (the reference leaves their alpha unset; 1 here)
                rg.aspect_h = 1080;
                hipLaunchKernelGGL(k_fill_black, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, s, (double *)d_rgba, n_pixels);
                if (d_depth) HIP_TRY(hipMemsetAsync(d_depth, 0, (size_t)n_pixels * sizeof(double), s));
            }
            rc = render_pass(ctx, rg, p->profile != 0, d_rgba, st, d_depth);
        }
    }
    if (rc) return rc;
    if (stats) *stats = st;
    return NDT_OK;
}

extern "C" int ndt_hip_render_device(ndt_hip_ctx *ctx, const ndt_render_params *p, void *d_rgba, ndt_render_stats *stats)
{
    return ndt_hip_render_depth_device(ctx, p, d_rgba, nullptr, stats);
}

extern "C" int ndt_hip_render(ndt_hip_ctx *ctx, const ndt_render_params *p, double *rgba, ndt_render_stats *stats)
{
    return ndt_hip_render_depth(ctx, p, rgba, nullptr, stats);
}

extern "C" int ndt_hip_render_depth(ndt_hip_ctx *ctx, const ndt_render_params *p, double *rgba, double *depth, ndt_render_stats *stats)
{
    if (!ctx || !p || !rgba) return fail(NDT_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int rows = ndt_hip_shard_rows(p->height, p->row_begin, p->row_step);
    const size_t img_bytes = (size_t)rows * (size_t)(p->width > 0 ? p->width : 0) * 4 * sizeof(double);
    const size_t bytes = img_bytes + (depth ? img_bytes / 4 : 0);      // the depth map sits behind the image
    if (img_bytes == 0) return ndt_hip_render_depth_device(ctx, p, (void *)rgba, nullptr, stats);
    if (ctx->d_out_bytes < bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->d_out) HIP_TRY(hipFree(ctx->d_out));
        ctx->d_out = nullptr;
        HIP_TRY(hipMalloc(&ctx->d_out, bytes));
        ctx->d_out_bytes = bytes;
    }
    void *d_depth = depth ? (void *)((char *)ctx->d_out + img_bytes) : nullptr;
    int rc = ndt_hip_render_depth_device(ctx, p, ctx->d_out, d_depth, stats);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(rgba, ctx->d_out, img_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (depth) HIP_TRY(hipMemcpyAsync(depth, d_depth, img_bytes / 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return NDT_OK;
}

extern "C" int ndt_hip_quantize_device(ndt_hip_ctx *ctx, const void *d_rgba, void *d_rgba8, int64_t n_pixels)
{
    if (!ctx || !d_rgba || !d_rgba8 || n_pixels < 0) return fail(NDT_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const long long n = n_pixels * 4;
    if (n == 0) return NDT_OK;
    hipLaunchKernelGGL(k_quantize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)d_rgba,
                       (unsigned char *)d_rgba8, n);
    HIP_TRY(hipGetLastError());
    return NDT_OK;
}

// ------------------------------------------------------------------ trace_kd batches

extern "C" int ndt_hip_trace_rays(ndt_hip_ctx *ctx, int64_t n_rays, const double *o, const double *v, const double *dist_limit,
                                  int32_t *obj, double *hit, double *normal)
{
    if (!ctx || n_rays < 0 || (n_rays > 0 && (!o || !v || !dist_limit || !obj || !hit || !normal)))
        return fail(NDT_E_INVALID, "bad argument");
    if (!ctx->have_scene) return fail(NDT_E_STATE, "no scene uploaded");
    if (n_rays == 0) return NDT_OK;
    if (n_rays > 0x3fffffffLL) return fail(NDT_E_UNSUPPORTED, "too many rays for one call");
    HIP_TRY(hipSetDevice(ctx->device));
    const int n = ctx->dims;
    const long long cnt = n_rays;
    int rc = ensure_workspace(ctx, cnt > ctx->ws.cap ? cnt : ctx->ws.cap, ctx->ws.sh_cap > 0 ? ctx->ws.sh_cap : 4096);
    if (rc) return rc;
    Workspace ws = ctx->ws;
    hipStream_t s = ctx->stream;
    // ray-major host arrays -> the pool's tiles of 64 slots (component-major inside a tile)
    const long long padded = (cnt + 63) & ~63LL;
    std::vector<double> so((size_t)n * padded, 0.0), sv((size_t)n * padded, 0.0);
    auto tile_at = [n](long long r, int c) { return (size_t)((r >> 6) * (long long)(n * 64) + c * 64 + (r & 63)); };
    for (long long r = 0; r < cnt; ++r)
        for (int c = 0; c < n; ++c) {
            so[tile_at(r, c)] = o[r * n + c];
            sv[tile_at(r, c)] = v[r * n + c];
        }
    HIP_TRY(hipMemcpyAsync(ws.ray_o, so.data(), so.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ws.ray_v, sv.data(), sv.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ws.frac, dist_limit, cnt * sizeof(double), hipMemcpyHostToDevice, s));
    TraceJob tj{};
    tj.n_seg = 0;
    tj.dense.o = ws.ray_o; tj.dense.v = ws.ray_v; tj.dense.stride = ws.cap; tj.dense.lim = ws.frac; tj.dense.valid = nullptr;
    tj.dense.out_obj = ws.hit_obj; tj.dense.out_prim = ws.hit_prim; tj.begin = 0; tj.count = cnt; tj.levels = nullptr;
    tj.queue = ws.counters + NDT_CNT_QUEUE;
    HIP_TRY(hipMemsetAsync(ws.counters + NDT_CNT_QUEUE, 0, NDT_QUEUE_INTS * sizeof(int), s));
    ctx->kt->trace(s, ctx->d_blob, ctx->sd, ws, tj, ctx->tier, ctx->sd.mask_words, nullptr, nullptr);
    ctx->kt->hitpoints(s, ctx->d_blob, ctx->sd, ws.ray_o, ws.ray_v, ws.cap, ws.hit_prim, ws.hit_p, ws.hit_n, cnt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(obj, ws.hit_obj, cnt * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(so.data(), ws.hit_p, so.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(sv.data(), ws.hit_n, sv.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (long long r = 0; r < cnt; ++r)
        for (int c = 0; c < n; ++c) {
            hit[r * n + c] = so[tile_at(r, c)];
            normal[r * n + c] = sv[tile_at(r, c)];
        }
    return NDT_OK;
}
