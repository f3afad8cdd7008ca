// ndt_blob.hip -- the scene blob: validate an ndt_flat_scene, derive what the reference's plugins derive lazily in
// prepare() (objects/ *.c), the hull boxes of the hcubes, and lay everything out as one array of 8-byte words.
// Host code only; compiled with -ffp-contract=off like everything else (the prepare() arithmetic is part of the
// numerical contract).
#include "ndt_ctx.hpp"
#include <algorithm>

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
    if (!hcube_hull_box(fs, fs->objects[object], fs->dims, rows, &hf) || hf.n_faces == 0 || hf.n_faces > 63) return 0;
    memcpy(face_rows, hf.rows.data(), hf.rows.size() * sizeof(double));
    *possible = hf.possible[0];
    return hf.n_faces;
}

extern "C" int64_t ndt_hip_hcube_face_tree(const ndt_flat_scene *fs, int32_t object, int64_t cap_nodes, double *rows_out, int32_t *level_off, int32_t *top_out)
{
    std::vector<double> hull((size_t)(fs ? fs->dims : 1) * (fs ? fs->dims + 2 : 1));
    const int rc = ndt_hip_hcube_hull_box(fs, object, hull.data());      // validates the arguments
    if (rc <= 0) return rc;
    if (!level_off || !top_out) return fail(NDT_E_INVALID, "null argument");
    std::vector<double> rows;
    HullFaces hf;
    if (!hcube_hull_box(fs, fs->objects[object], fs->dims, rows, &hf) || hf.n_faces == 0) return 0;
    std::vector<double> trows;
    std::vector<int> loff;
    int top = 0;
    hcube_face_tree(hf, fs->dims, trows, loff, top);
    const int64_t n_nodes = (int64_t)(trows.size() / (2 * (size_t)fs->dims));
    *top_out = top;
    if (!rows_out || cap_nodes < n_nodes) return n_nodes;                // (how many: call again with room for them)
    for (int j = 0; j <= top && j < 32; ++j) level_off[j] = loff[(size_t)j];
    memcpy(rows_out, trows.data(), trows.size() * sizeof(double));
    return n_nodes;
}

extern "C" int ndt_hip_hcube_face_groups(const ndt_flat_scene *fs, int32_t object, double *clusters_out, int32_t *table_out, int32_t *face_set_out,
                                         int32_t *members_out)
{
    std::vector<double> hull((size_t)(fs && fs->dims > 0 && fs->dims <= NDT_MAX_DIMS ? fs->dims * (fs->dims + 2) : 1));
    const int rc = ndt_hip_hcube_hull_box(fs, object, hull.data());      // validates the arguments
    if (rc <= 0) return rc;
    if (!clusters_out || !table_out || !face_set_out) return fail(NDT_E_INVALID, "null argument");
    std::vector<double> rows;
    HullFaces hf;
    if (!hcube_hull_box(fs, fs->objects[object], fs->dims, rows, &hf) || hf.n_faces == 0) return 0;
    std::vector<double> clusters;
    std::vector<int> table, face_set, members;
    hcube_face_groups(hf, rows, fs->dims, clusters, table, face_set, members);
    memcpy(clusters_out, clusters.data(), clusters.size() * sizeof(double));
    memcpy(table_out, table.data(), table.size() * sizeof(int));
    memcpy(face_set_out, face_set.data(), face_set.size() * sizeof(int));
    if (members_out) memcpy(members_out, members.data(), members.size() * sizeof(int));
    return hf.n_faces;
}

extern "C" int64_t ndt_hip_hcube_face_boxes_all(const ndt_flat_scene *fs, int32_t object, int64_t cap_faces, double *face_rows, uint8_t *possible)
{
    std::vector<double> hull((size_t)(fs && fs->dims > 0 && fs->dims <= NDT_MAX_DIMS ? fs->dims * (fs->dims + 2) : 1));
    const int rc = ndt_hip_hcube_hull_box(fs, object, hull.data());      // validates the arguments
    if (rc <= 0) return rc;
    std::vector<double> rows;
    HullFaces hf;
    if (!hcube_hull_box(fs, fs->objects[object], fs->dims, rows, &hf) || hf.n_faces == 0) return 0;
    if (!face_rows || !possible || cap_faces < hf.n_faces) return hf.n_faces;        // (how many: call again with room for them)
    memcpy(face_rows, hf.rows.data(), hf.rows.size() * sizeof(double));
    for (int f = 0; f < hf.n_faces; ++f) possible[f] = (uint8_t)((hf.possible[(size_t)f / NDT_HULL_CHUNK] >> (f % NDT_HULL_CHUNK)) & 1ull);
    return hf.n_faces;
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
#define NDT_HULL_CHUNK 63           /* face boxes: 63 faces a mask word -- one bit per face, the top bit stays clear (trace_kd) */
#define NDT_HULL_MAX_FACES (1 << 22) /* ... and as many words as the hcube needs (a 10-D one nests 52 904 faces) */

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

// The hierarchy over an hcube's face boxes (ndt_device.hpp:hull_faces, NDT_F_FACETREE): level j, node k = the box of the union
// of the boxes of faces [k 2^j, (k + 1) 2^j) that can be hit at all; a run without such a face gets half extents of -1.
// level_off[j] (j = 1 .. top): first node of level j in `rows` (2n doubles a node); level 0 is the faces' own rows.
void ndt_impl::hcube_face_tree(const HullFaces &hf, int n, std::vector<double> &rows, std::vector<int> &level_off, int &top)
{
    const int nf = hf.n_faces;
    top = 0;
    while ((1 << top) < nf) ++top;
    level_off.assign((size_t)top + 1, 0);
    rows.clear();
    // lo / hi of the level below (level 0: from the faces' { centre, half })
    std::vector<double> lo((size_t)nf * n), hi((size_t)nf * n);
    std::vector<char> any((size_t)nf, 0);
    for (int f = 0; f < nf; ++f) {
        any[(size_t)f] = (char)((hf.possible[(size_t)f / NDT_HULL_CHUNK] >> (f % NDT_HULL_CHUNK)) & 1ull);
        for (int a = 0; a < n; ++a) {
            const double c = hf.rows[((size_t)f * n + a) * 2], h = hf.rows[((size_t)f * n + a) * 2 + 1];
            lo[(size_t)f * n + a] = c - h;
            hi[(size_t)f * n + a] = c + h;
        }
    }
    int cnt = nf;
    for (int j = 1; j <= top; ++j) {
        const int up = (cnt + 1) / 2;
        std::vector<double> nlo((size_t)up * n), nhi((size_t)up * n);
        std::vector<char> nany((size_t)up, 0);
        level_off[(size_t)j] = (int)(rows.size() / (2 * (size_t)n));
        for (int k = 0; k < up; ++k) {
            for (int a = 0; a < n; ++a) {
                double l = 1e300, h = -1e300;
                for (int c = 2 * k; c < 2 * k + 2 && c < cnt; ++c)
                    if (any[(size_t)c]) {
                        if (lo[(size_t)c * n + a] < l) l = lo[(size_t)c * n + a];
                        if (hi[(size_t)c * n + a] > h) h = hi[(size_t)c * n + a];
                        nany[(size_t)k] = 1;
                    }
                nlo[(size_t)k * n + a] = l;
                nhi[(size_t)k * n + a] = h;
                if (nany[(size_t)k]) {
                    // (centre +- half must hold [l, h] whatever the rounding of the two: a nanometre more)
                    rows.push_back(0.5 * (l + h));
                    rows.push_back(0.5 * (h - l) + 1e-9 * (1.0 + fabs(l) + fabs(h)));
                } else {
                    rows.push_back(0.0);
                    rows.push_back(-1.0);
                }
            }
        }
        lo.swap(nlo);
        hi.swap(nhi);
        any.swap(nany);
        cnt = up;
    }
}

// The faces of an hcube indexed by the hull axes their boxes are THIN on (ndt_device.hpp:hull_faces, NDT_F_FACEGROUPS).  An m-face
// of an N-cube is pinned on N - m of the cube's axes and spans the other m: its box in the hull's frame is a sliver (two margins
// wide) on the pinned axes.  A ray can meet the box only if, inside the hull, it passes through the sliver on EVERY pinned axis;
// per axis all slivers lie in two clusters (the cube's two sides).  So the device finds the axes on which the ray passes a
// cluster at all -- usually the one or two where it crosses the cube's surface -- and looks only at the faces whose pinned set is
// a subset of those: table[S] = the list (ascending) of the faces pinned exactly on S.  Which axes count as thin is a heuristic
// (a quarter of the hull's extent) and does not matter for correctness: a face listed under S has, on every axis of S, its
// interval inside one of that axis's cluster intervals (they are made from exactly those faces), and every listed face is
// still tested against its own box.
void ndt_impl::hcube_face_groups(const HullFaces &hf, const std::vector<double> &hull_rows, int n, std::vector<double> &clusters,
                                 std::vector<int> &table, std::vector<int> &face_set, std::vector<int> &members)
{
    const int nf = hf.n_faces;
    std::vector<double> lo((size_t)n * 2, 1e300), hi((size_t)n * 2, -1e300);
    face_set.assign((size_t)nf, 0);
    table.assign((size_t)2 << n, 0);
    for (int f = 0; f < nf; ++f) {
        if (!((hf.possible[(size_t)f / NDT_HULL_CHUNK] >> (f % NDT_HULL_CHUNK)) & 1ull)) { face_set[(size_t)f] = -1; continue; }
        int set = 0;
        for (int a = 0; a < n; ++a) {
            const double c = hf.rows[((size_t)f * n + a) * 2], h = hf.rows[((size_t)f * n + a) * 2 + 1];
            const double hc = hull_rows[(size_t)a * (n + 2) + n], hh = hull_rows[(size_t)a * (n + 2) + n + 1];
            if (h < 0.25 * hh) {
                set |= 1 << a;
                const int side = c < hc ? 0 : 1;
                if (c - h < lo[(size_t)a * 2 + side]) lo[(size_t)a * 2 + side] = c - h;
                if (c + h > hi[(size_t)a * 2 + side]) hi[(size_t)a * 2 + side] = c + h;
            }
        }
        face_set[(size_t)f] = set;
        table[(size_t)set * 2 + 1] += 1;
    }
    // table[S] = { start, count } of S's faces in `members`, ascending (the same axis set comes from several direction sets of a
    // sheared cube -- its skewed faces are blobs, thin on every axis -- so the faces of a set need not be neighbours)
    int at = 0;
    for (size_t sset = 0; sset < ((size_t)1 << n); ++sset) {
        table[sset * 2] = at;
        at += table[sset * 2 + 1];
        table[sset * 2 + 1] = 0;
    }
    members.assign((size_t)at, 0);
    for (int f = 0; f < nf; ++f) {
        const int set = face_set[(size_t)f];
        if (set < 0) continue;
        members[(size_t)table[(size_t)set * 2] + (size_t)table[(size_t)set * 2 + 1]++] = f;
    }
    clusters.assign((size_t)n * 4, 0.0);
    for (int a = 0; a < n; ++a)
        for (int side = 0; side < 2; ++side) {
            const double l = lo[(size_t)a * 2 + side], h = hi[(size_t)a * 2 + side];
            clusters[(size_t)a * 4 + side * 2] = (l <= h) ? 0.5 * (l + h) : 0.0;
            // (centre +- half must hold [l, h] whatever the rounding of the two: a nanometre more; -1: no face is thin here)
            clusters[(size_t)a * 4 + side * 2 + 1] = (l <= h) ? 0.5 * (h - l) + 1e-9 * (1.0 + fabs(l) + fabs(h)) : -1.0;
        }
}

bool ndt_impl::hcube_hull_box(const ndt_flat_scene *fs, const ndt_flat_object &o, int n, std::vector<double> &rows, HullFaces *faces)
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
    // A fourth candidate, taken whenever it exists (round 4): the DUAL basis of the cube's edge directions.  A slab test needs
    // covectors, not an orthonormal frame -- w_k.(o + t v) in [lo, hi] bounds a convex region for any w_k -- and in the
    // coordinates c_k = w_k.x with x = sum c_k d_k every face of a parallelotope is pinned EXACTLY on the coordinates of the
    // directions it does not span, sheared cube or not: its box is two margins thin there (|w_k| = 1: a displacement moves a
    // coordinate by at most its length, so the margins mean what they mean in an orthonormal frame).  In an orthonormal frame the
    // faces of a sheared cube are slivers only on the first axes; the boxes of runs of them, and the clusters of
    // hcube_face_groups, are wide.
    bool have_dual = false;
    {
        std::vector<std::vector<double>> dirs;
        for (const auto &a : axes) {
            bool known = false;
            for (const auto &d : dirs) {
                double dot = 0;
                for (int c = 0; c < n; ++c) dot += a[c] * d[c];
                if (fabs(fabs(dot) - 1.0) < 1e-9) known = true;
            }
            if (!known) {
                if ((int)dirs.size() == n) { dirs.clear(); break; }     // more directions than dimensions: not a parallelotope
                dirs.push_back(a);
            }
        }
        if ((int)dirs.size() == n) {
            // x = sum c_k d_k  <=>  c = (D^T)^-1 x: Gauss-Jordan with partial pivoting on [D^T | I]
            std::vector<double> m((size_t)n * 2 * n, 0.0);
            for (int r = 0; r < n; ++r) {
                for (int c = 0; c < n; ++c) m[(size_t)r * 2 * n + c] = dirs[(size_t)c][(size_t)r];
                m[(size_t)r * 2 * n + n + r] = 1.0;
            }
            bool ok = true;
            for (int col = 0; col < n && ok; ++col) {
                int piv = col;
                for (int r = col + 1; r < n; ++r)
                    if (fabs(m[(size_t)r * 2 * n + col]) > fabs(m[(size_t)piv * 2 * n + col])) piv = r;
                if (fabs(m[(size_t)piv * 2 * n + col]) < 1e-3) { ok = false; break; }       // nearly dependent directions
                if (piv != col)
                    for (int c = 0; c < 2 * n; ++c) std::swap(m[(size_t)piv * 2 * n + c], m[(size_t)col * 2 * n + c]);
                const double d = m[(size_t)col * 2 * n + col];
                for (int c = 0; c < 2 * n; ++c) m[(size_t)col * 2 * n + c] /= d;
                for (int r = 0; r < n; ++r)
                    if (r != col) {
                        const double f = m[(size_t)r * 2 * n + col];
                        if (f != 0.0)
                            for (int c = 0; c < 2 * n; ++c) m[(size_t)r * 2 * n + c] -= f * m[(size_t)col * 2 * n + c];
                    }
            }
            if (ok) {
                std::vector<std::vector<double>> dual;
                for (int k = 0; k < n && ok; ++k) {
                    std::vector<double> w(m.begin() + (size_t)k * 2 * n + n, m.begin() + (size_t)k * 2 * n + 2 * n);
                    const double l = h_len(w.data(), n);
                    if (!(l > 0) || !std::isfinite(l) || l > 1e3) ok = false;     // (|w_k| = 1 / sin of d_k's angle to the others' span)
                    else {
                        for (int c = 0; c < n; ++c) w[(size_t)c] /= l;
                        dual.push_back(w);
                    }
                }
                if (ok) {
                    frames.push_back(dual);
                    have_dual = true;
                }
            }
        }
    }
    double best_cost = 0;
    int best = -1;
    std::vector<double> best_rows;
    for (int fi = 0; fi < (int)frames.size(); ++fi) {
        auto &frame = frames[fi];
        if (fi < 3)
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
        if (best < 0 || cost < best_cost || (have_dual && fi == 3)) {
            best = fi;
            best_cost = cost;
            best_rows = cand;
        }
    }
    if (best < 0) return false;
    // thinnest slab first: the device asks after every slab whether any ray of the wavefront is still inside
    // (ndt_device.hpp:hull_faces), and two thin slabs reject what four in any order would
    {
        std::vector<int> order((size_t)n);
        for (int a = 0; a < n; ++a) order[(size_t)a] = a;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
            return best_rows[(size_t)x * (n + 2) + n + 1] < best_rows[(size_t)y * (n + 2) + n + 1];
        });
        rows.clear();
        for (int a = 0; a < n; ++a)
            rows.insert(rows.end(), best_rows.begin() + (size_t)order[(size_t)a] * (n + 2),
                        best_rows.begin() + (size_t)(order[(size_t)a] + 1) * (n + 2));
    }
    if (faces) {
        // Every face's own box in the chosen frame.  The hull box is the union of these: a ray that misses box f
        // cannot produce a point face f's intersect() accepts (same argument, one face at a time), so the device
        // scans only the faces whose box the ray meets -- of the 2-D faces of a 4-D hcube, usually none or two.
        faces->rows.clear();
        faces->possible.clear();
        faces->n_faces = 0;
        if (o.n_obj >= 1 && o.n_obj <= NDT_HULL_MAX_FACES) {
            faces->n_faces = o.n_obj;
            faces->possible.assign((size_t)(o.n_obj + NDT_HULL_CHUNK - 1) / NDT_HULL_CHUNK, 0ull);
            face_begin.push_back(n_pts);
            for (int k = 0; k < o.n_obj; ++k) {
                const size_t p0 = face_begin[k], p1 = face_begin[k + 1];
                if (p1 > p0) faces->possible[(size_t)k / NDT_HULL_CHUNK] |= 1ull << (k % NDT_HULL_CHUNK);
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

// Item boxes: the face boxes of an hcube, for the orthotopes that stand in the scene on their own (the 6-D .. 8-D
// hypercubes of scenes/hypercube.c are thousands of them: every m-face of the cube is an orthotope of its own, behind a
// bounding sphere as wide as its diagonal -- 44 % of the sphere gates pass, and a ray that crosses the cube intersects a
// hundred faces to hit one).  ONE orthonormal frame for the whole scene (Gram-Schmidt of the orthotopes' own axes: for a
// cube, rotated or not, its edges), and per orthotope the box, in that frame, of everything its intersect() can return --
// hcube_hull_box's derivation, one object at a time, same margin.  A ray projects itself on the frame once (2N dot
// products, the first time it meets a boxed item) and then tests an item with N slab updates; an item whose box it misses
// cannot be hit in the reference's arithmetic, so neither its gate nor its intersect() is run -- the answer they would
// give is "no hit" (ndt_device.hpp:item_box_meets).
//   frame: N x unit axis[N];  rows: n_items x N x { centre, half extent };  has[i] = item i carries a box
bool ndt_impl::scene_item_boxes(const ndt_flat_scene *fs, int n, std::vector<double> &frame, std::vector<double> &rows,
                                std::vector<char> &has)
{
    has.assign((size_t)(fs->n_items > 0 ? fs->n_items : 1), 0);
    rows.assign((size_t)(fs->n_items > 0 ? fs->n_items : 0) * 2 * n, 0.0);
    frame.clear();
    std::vector<std::vector<double>> pts_of((size_t)(fs->n_items > 0 ? fs->n_items : 0));
    std::vector<std::vector<double>> axes;
    int boxed = 0;
    for (int i = 0; i < fs->n_items; ++i) {
        const ndt_flat_object &f = fs->objects[i];
        if (f.type != NDT_OBJ_ORTHOTOPE || f.n_flag < 1 || f.n_pos < 1) continue;
        if (f.flag_off < 0 || (int64_t)f.flag_off + f.n_flag > fs->n_flags) continue;
        const int m = fs->flags[f.flag_off];
        if (m < 1 || m > f.n_dir || m > n || m > 16) continue;
        if (!vec_ok(fs, f.pos_off, 1) || !vec_ok(fs, f.dir_off, m)) continue;
        std::vector<double> pts;
        if (!face_region_corners(fs->vecs + f.pos_off, fs->vecs + f.dir_off, m, n, pts) || pts.empty()) continue;
        bool finite = true;
        for (double x : pts)
            if (!std::isfinite(x)) finite = false;
        if (!finite) continue;
        pts_of[(size_t)i].swap(pts);
        ++boxed;
        if ((int)axes.size() < 4 * n)
            for (int a = 0; a < m; ++a) {
                std::vector<double> u(fs->vecs + f.dir_off + a * n, fs->vecs + f.dir_off + (a + 1) * n);
                h_unitize(u.data(), n);
                axes.push_back(u);
            }
    }
    if (boxed == 0) return false;
    // the frame: the orthotopes' axes as far as they are independent, then the world's
    std::vector<std::vector<double>> fr;
    auto complete = [&](const std::vector<std::vector<double>> &cands, double keep) {
        for (const auto &a : cands) {
            if ((int)fr.size() >= n) break;
            std::vector<double> r(a);
            for (const auto &u : fr) {
                double d = 0;
                for (int c = 0; c < n; ++c) d += a[c] * u[c];
                for (int c = 0; c < n; ++c) r[c] -= d * u[c];
            }
            const double l = h_len(r.data(), n);
            if (l > keep) {
                for (int c = 0; c < n; ++c) r[c] /= l;
                fr.push_back(r);
            }
        }
    };
    std::vector<std::vector<double>> world;
    for (int j = 0; j < n; ++j) {
        std::vector<double> e((size_t)n, 0.0);
        e[j] = 1.0;
        world.push_back(e);
    }
    complete(axes, 0.5);
    for (double keep = 0.5; (int)fr.size() < n && keep > 1e-4; keep *= 0.5) complete(world, keep);
    if ((int)fr.size() < n) return false;
    for (const auto &u : fr) frame.insert(frame.end(), u.begin(), u.end());
    for (int i = 0; i < fs->n_items; ++i) {
        const std::vector<double> &pts = pts_of[(size_t)i];
        if (pts.empty()) continue;
        const size_t np = pts.size() / n;
        for (int a = 0; a < n; ++a) {
            double lo = 1e300, hi = -1e300;
            for (size_t q = 0; q < np; ++q) {
                double d = 0;
                for (int c = 0; c < n; ++c) d += pts[q * n + c] * fr[(size_t)a][c];
                if (d < lo) lo = d;
                if (d > hi) hi = d;
            }
            rows[((size_t)i * n + a) * 2] = 0.5 * (lo + hi);
            rows[((size_t)i * n + a) * 2 + 1] = 0.5 * (hi - lo) + NDT_HULL_MARGIN;
        }
        has[(size_t)i] = 1;
    }
    return true;
}

extern "C" int ndt_hip_item_boxes(const ndt_flat_scene *fs, double *frame_out, double *rows_out, uint8_t *has_out)
{
    if (!fs || !frame_out || !rows_out || !has_out) return fail(NDT_E_INVALID, "null argument");
    if (fs->abi_version != NDT_HIP_ABI_VERSION) return fail(NDT_E_INVALID, "flat scene ABI %d, library %d", fs->abi_version, NDT_HIP_ABI_VERSION);
    if (fs->dims < 3 || fs->dims > NDT_MAX_DIMS) return fail(NDT_E_UNSUPPORTED, "dims %d", fs->dims);
    if (fs->n_items < 0 || fs->n_items > fs->n_objects) return fail(NDT_E_INVALID, "n_items %d", fs->n_items);
    std::vector<double> frame, rows;
    std::vector<char> has;
    if (!scene_item_boxes(fs, fs->dims, frame, rows, has)) return 0;
    memcpy(frame_out, frame.data(), frame.size() * sizeof(double));
    memcpy(rows_out, rows.data(), rows.size() * sizeof(double));
    int n = 0;
    for (int i = 0; i < fs->n_items; ++i) {
        has_out[i] = (uint8_t)has[(size_t)i];
        n += has[(size_t)i] != 0;
    }
    return n;
}

int ndt_impl::build_blob(ndt_hip_ctx *ctx, const ndt_flat_scene *fs)
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
    // Item sets (ndt_device.hpp:trace_kd): with at most 64 items and every list ascending in item number -- the reference's
    // kd builder keeps the scene's object order in every leaf -- a leaf record carries its items as a 64-bit set.  Any other
    // scene of up to 256 items takes the kernels with a four-word mask, which read the lists.
    bool item_sets = sd.mask_words == 1 && ctx->item_sets;
    if (item_sets) {
        for (int i = 0; i < fs->n_kd_nodes && item_sets; ++i) {
            const ndt_flat_kdnode &k = fs->kd_nodes[i];
            if (k.dim >= 0) continue;
            if (k.num < 0 || k.first < 0 || (int64_t)k.first + k.num > fs->n_leaf_refs) return fail(NDT_E_INVALID, "kd leaf %d: item range", i);
            for (int j = 0; j < k.num; ++j) {
                const int id = fs->leaf_refs[k.first + j];
                if (id < 0 || id >= 64 || (j > 0 && id <= fs->leaf_refs[k.first + j - 1])) item_sets = false;
            }
        }
        for (int i = 0; i < fs->n_inf; ++i)
            if (fs->inf_refs[i] < 0 || fs->inf_refs[i] >= 64 || (i > 0 && fs->inf_refs[i] <= fs->inf_refs[i - 1])) item_sets = false;
    }
    std::vector<int> new_index((size_t)(fs->n_kd_nodes > 0 ? fs->n_kd_nodes : 1), -1);
    for (size_t i = 0; i < order.size(); ++i) new_index[order[i]] = (int)i;
    sd.n_kd_nodes = (int)order.size();
    std::vector<int> leaf_list, leaf_first, leaf_num;
    sd.off_kd = b.words();
    for (size_t i = 0; i < order.size(); ++i) {
        const ndt_flat_kdnode &k = fs->kd_nodes[order[i]];
        if (k.dim >= 0) {
            if (new_index[k.left] != (int)i + 1) return fail(NDT_E_INVALID, "internal: preorder numbering");
            b.push_ints(k.dim, new_index[k.right]);
            b.push(k.boundary);
        } else {
            b.push_ints(-1, (int)leaf_first.size());       // a leaf's ordinal (VisitMask<0>, leaf history)
            leaf_first.push_back((int)leaf_list.size());
            leaf_num.push_back(k.num);
            b.push_ints((int)leaf_list.size(), k.num);
            for (int j = 0; j < k.num; ++j) leaf_list.push_back(fs->leaf_refs[k.first + j]);
        }
    }
    sd.off_nset = b.words();
    if (item_sets) {
        // children follow their parent in preorder: one pass from the back
        std::vector<unsigned long long> below(order.size(), 0ull);
        for (size_t i = order.size(); i-- > 0;) {
            const ndt_flat_kdnode &k = fs->kd_nodes[order[i]];
            if (k.dim >= 0) below[i] = below[i + 1] | below[(size_t)new_index[k.right]];
            else for (int j = 0; j < k.num; ++j) below[i] |= 1ull << fs->leaf_refs[k.first + j];
        }
        for (unsigned long long set : below) b.push_ints((int)(set & 0xffffffffull), (int)(set >> 32));
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

    // ---- item boxes (scene_item_boxes): for scenes that will live in global memory -- more than 256 items
    std::vector<double> ib_frame, ib_rows;
    std::vector<char> ib_has;
    const bool item_boxes = ctx->item_boxes && sd.mask_words > NDT_MASK_REG_WORDS && scene_item_boxes(fs, n, ib_frame, ib_rows, ib_has);

    // ---- per-type parameters = the plugins' prepare() output
    int max_param_words = 0;
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
        if (item_boxes && i < fs->n_items && ib_has[(size_t)i]) flags |= NDT_F_OBOX;
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
            if (ctx->hull_box && hcube_hull_box(fs, o, n, rows, &hf)) {
                flags |= NDT_F_BOX;
                for (double x : rows) b.push(x);
                if (hf.n_faces > 0 && ctx->face_box) {
                    // { possible-faces masks, one word per 63 faces } + per face N x { centre, half extent }
                    flags |= NDT_F_FACEBOX;
                    for (unsigned long long w : hf.possible) b.push_ints((int)(w & 0xffffffffull), (int)(w >> 32));
                    for (double x : hf.rows) b.push(x);
                    if (hf.n_faces > NDT_HULL_CHUNK && ctx->face_groups && n >= NDT_GROUPS_MIN_DIMS) {
                        // ... + the faces by the hull axes their boxes are thin on: n x { centre-, half-, centre+, half+ } of the
                        // two clusters of slivers per axis, then one word { start, count } per subset of the axes, then the faces by subset
                        flags |= NDT_F_FACEGROUPS;
                        std::vector<double> clusters;
                        std::vector<int> table, face_set, members;
                        hcube_face_groups(hf, rows, n, clusters, table, face_set, members);
                        for (double x : clusters) b.push(x);
                        for (size_t k = 0; k < table.size(); k += 2) b.push_ints(table[k], table[k + 1]);
                        // (one word per two members, as many words as two faces a word need whatever the number of members:
                        // the device finds the hierarchy behind them without reading a length)
                        members.resize((size_t)((hf.n_faces + 1) & ~1), 0);
                        for (size_t k = 0; k < members.size(); k += 2) b.push_ints(members[k], members[k + 1]);
                    }
                    if (hf.n_faces > NDT_HULL_CHUNK && ctx->face_tree) {
                        // ... + the hierarchy over them: { top level, 0 }, the levels' offsets (ints, two a word, in words
                        // from here), the rows of levels 1 .. top
                        flags |= NDT_F_FACETREE;
                        std::vector<double> trows;
                        std::vector<int> loff;
                        int top = 0;
                        hcube_face_tree(hf, n, trows, loff, top);
                        const int table_words = 1 + (top + 2) / 2;
                        b.push_ints(top, 0);
                        for (int j = 0; j <= top; j += 2)
                            b.push_ints(j >= 1 ? table_words + loff[(size_t)j] * 2 * n : 0,
                                        j + 1 <= top ? table_words + loff[(size_t)j + 1] * 2 * n : 0);
                        for (double x : trows) b.push(x);
                    }
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
        // (every type but hcube: aux0 = the words of its parameter record, for the coherent leaf scan's staging copy)
        if (o.type != NDT_OBJ_HCUBE) {
            aux0 = b.words() - sd.off_params - p;
            if (aux0 > max_param_words) max_param_words = aux0;
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
        else if (i < 64) sd.ambient_bits |= 1ull << i;
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
    // tier 0: trace sections fit the LDS budget and the visit mask fits registers
    const bool fits_lds = (size_t)sd.trace_words * sizeof(double) <= NDT_TRACE_LDS_LIMIT;
    ctx->tier = (fits_lds && sd.mask_words <= NDT_MASK_REG_WORDS) ? 0 : 1;
    // (LDS tiers only: in the global-memory tier's trace kernel the few lines that find a segment's light cost more -- spilled
    // registers at 6-D .. 8-D -- than the origins' traffic: hypercube frames +1 .. +2.5 %, measured)
    sd.light_origins = ctx->tier == 0 ? 1 : 0;
    // Leaf history (ndt_device.hpp:VisitMask<0>): the items of every leaf as a bit set + every leaf's list range, when
    // every leaf list ascends in item number (a scan that stopped after item `last` then visited the leaf's items <= last)
    sd.off_lset = sd.off_lrange = 0;
    sd.hist_cap = ctx->leaf_history > 4 ? 4 : ctx->leaf_history;
    if (ctx->tier == 1 && ctx->leaf_history > 0 && fs->n_items < 65535 && leaf_first.size() < 65536 && !leaf_first.empty()) {
        bool ascending = true;
        for (size_t l = 0; l < leaf_first.size() && ascending; ++l)
            for (int j = 0; j < leaf_num[l]; ++j) {
                const int id = leaf_list[(size_t)leaf_first[l] + j];
                if (id < 0 || id >= fs->n_items || (j > 0 && id <= leaf_list[(size_t)leaf_first[l] + j - 1])) ascending = false;
            }
        if (ascending) {
            sd.off_lset = b.words();
            std::vector<unsigned long long> set((size_t)sd.mask_words);
            for (size_t l = 0; l < leaf_first.size(); ++l) {
                std::fill(set.begin(), set.end(), 0ull);
                for (int j = 0; j < leaf_num[l]; ++j) {
                    const int id = leaf_list[(size_t)leaf_first[l] + j];
                    set[(size_t)id >> 6] |= 1ull << (id & 63);
                }
                for (unsigned long long x : set) b.push_ints((int)(x & 0xffffffffull), (int)(x >> 32));
            }
            sd.off_lrange = b.words();
            for (size_t l = 0; l < leaf_first.size(); ++l) b.push_ints(leaf_first[l], leaf_num[l]);
            // Coherent leaf scan (ndt_device.hpp:cls_scan): the wavefronts whose lanes stand on the same leaf stage its
            // items through LDS.  Needs the history (a scan that stores nothing) and leaves without composites.
            bool plain = ctx->leaf_scan;
            for (int id : leaf_list)
                if (fs->objects[id].type == NDT_OBJ_HCUBE) plain = false;
            // (a record travels as two words per lane: the sphere's n + 2 words and the parameters must fit 128)
            if (plain && max_param_words + n + 2 <= 128) sd.cls_par_words = (max_param_words + 1) & ~1;
            sd.cls_min_group = ctx->leaf_scan_group;
        }
    }
    sd.off_oframe = sd.off_obox = sd.off_oord = 0;
    if (item_boxes && ctx->tier == 1) {
        sd.off_oframe = b.words();
        for (double x : ib_frame) b.push(x);
        // per item: its N slabs thinnest first -- a ray that misses a box usually knows after two or three of them -- and
        // the axis of every slab (4 bits each), so that the ray finds its projection
        sd.off_obox = b.words();
        std::vector<unsigned long long> order((size_t)fs->n_items, 0ull);
        for (int i = 0; i < fs->n_items; ++i) {
            int idx[NDT_MAX_DIMS];
            for (int a = 0; a < n; ++a) idx[a] = a;
            const double *r = &ib_rows[(size_t)i * 2 * n];
            std::stable_sort(idx, idx + n, [&](int x, int y) { return r[2 * x + 1] < r[2 * y + 1]; });
            for (int a = 0; a < n; ++a) {
                b.push(r[2 * idx[a]]);
                b.push(r[2 * idx[a] + 1]);
                order[(size_t)i] |= (unsigned long long)idx[a] << (4 * a);
            }
        }
        sd.off_oord = b.words();
        for (unsigned long long x : order) b.push_ints((int)(x & 0xffffffffull), (int)(x >> 32));
    }
    sd.total_words = b.words();

    if (ctx->tier == 0 && sd.mask_words == 1) {
        if (item_sets) {
            for (int i = 0; i < fs->n_inf; ++i) sd.inf_bits |= 1ull << fs->inf_refs[i];
            for (int i = 0; i < fs->n_items && i < 64; ++i)
                    if (fs->objects[i].bounds_radius > 0) sd.gate_bits |= 1ull << i;
        } else {
            sd.mask_words = 2;          // the one-word kernels read sets: this scene takes the NDT_MASK_REG_WORDS-word ones
        }
    }
    ctx->sd = sd;
    ctx->blob.swap(b.w);
    ctx->n_shadow_lights = n_shadow_lights;
    return NDT_OK;
}
