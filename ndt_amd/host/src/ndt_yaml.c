/* ndt_yaml.c -- YAML scene files for this host model: scene_read_yaml / scene_write_yaml /
 * scene_yaml_count_frames (reference scene.c:573-2177, scene.h:80-86; scenes/yaml.c is the scene
 * program that calls them for `-s scenes/yaml.so -u file.yaml`).
 *
 * The reference goes through libyaml; this file carries its own reader for the YAML subset those
 * files use (block mappings and sequences, flow sequences/mappings that may wrap over lines,
 * plain and quoted scalars, `---` documents = animation frames), so the library has no external
 * dependency.  What is mirrored from the reference is what decides the scene:
 *   - keys and their case-insensitive matching (scene.c:1271-1390, 1658-1680, 1806-1860, 1950-2016,
 *     2048-2060); unknown keys are skipped with a warning;
 *   - numbers through atof / atoi (scene.c:1108, 1134): the files hold %.16g, not exact doubles;
 *   - an object is collected first and then built with object_alloc + object_add_pos/dir/size/flag/
 *     obj in that order, colours and flags copied even when the file has none (zeros),
 *     `prepared` objects keep their stored bounding sphere, objects that fail object_validate are
 *     dropped (scene.c:1691-1738);
 *   - lights start as scene_alloc_light leaves them (POINT, all zero) and the reader knows no
 *     `angle` key (scene.c:1806-1860): a spot light's cone does not survive a round trip there either;
 *   - the camera keys overwrite the fields directly; aiming happens later in the frame loop
 *     (camera_aim, ndt.c:1925).
 */
#include "ndt_host_internal.h"

#include <ctype.h>
#include <strings.h>

/* ------------------------------------------------------------------ document model */

typedef enum { Y_SCALAR, Y_MAP, Y_SEQ } ykind;

typedef struct ynode {
    ykind kind;
    char *text;                 /* scalar */
    char **keys;                /* map */
    struct ynode **items;       /* map values / sequence items */
    int n, cap;
} ynode;

static ynode *y_new(ykind k)
{
    ynode *n = (ynode *)calloc(1, sizeof(ynode));
    n->kind = k;
    return n;
}

static void y_free(ynode *n)
{
    if (!n) return;
    for (int i = 0; i < n->n; ++i) {
        if (n->keys) free(n->keys[i]);
        y_free(n->items[i]);
    }
    free(n->keys);
    free(n->items);
    free(n->text);
    free(n);
}

static void y_push(ynode *parent, char *key, ynode *child)
{
    if (parent->n == parent->cap) {
        parent->cap = parent->cap ? 2 * parent->cap : 8;
        parent->items = (ynode **)realloc(parent->items, (size_t)parent->cap * sizeof(ynode *));
        if (parent->kind == Y_MAP) parent->keys = (char **)realloc(parent->keys, (size_t)parent->cap * sizeof(char *));
    }
    if (parent->kind == Y_MAP) parent->keys[parent->n] = key;
    parent->items[parent->n++] = child;
}

/* ------------------------------------------------------------------ reader */

typedef struct {
    char **line;                /* NUL-terminated lines of one document, comments and blanks removed */
    int *indent;
    int n, pos;
    int error;
} ydoc;

static char *dup_range(const char *a, const char *b)
{
    while (a < b && isspace((unsigned char)*a)) ++a;
    while (b > a && isspace((unsigned char)b[-1])) --b;
    char *s = (char *)malloc((size_t)(b - a) + 1);
    memcpy(s, a, (size_t)(b - a));
    s[b - a] = '\0';
    return s;
}

/* a scalar token: plain, 'single quoted' ('' escapes a quote) or "double quoted" (\\ \" \n \t) */
static char *unquote(char *s)
{
    size_t len = strlen(s);
    if (len >= 2 && s[0] == '\'' && s[len - 1] == '\'') {
        char *o = (char *)malloc(len);
        size_t k = 0;
        for (size_t i = 1; i + 1 < len; ++i) {
            if (s[i] == '\'' && s[i + 1] == '\'' && i + 2 < len) ++i;
            o[k++] = s[i];
        }
        o[k] = '\0';
        free(s);
        return o;
    }
    if (len >= 2 && s[0] == '"' && s[len - 1] == '"') {
        char *o = (char *)malloc(len);
        size_t k = 0;
        for (size_t i = 1; i + 1 < len; ++i) {
            if (s[i] == '\\' && i + 2 < len) {
                ++i;
                o[k++] = s[i] == 'n' ? '\n' : s[i] == 't' ? '\t' : s[i];
            } else {
                o[k++] = s[i];
            }
        }
        o[k] = '\0';
        free(s);
        return o;
    }
    if (!strcmp(s, "~") || !strcasecmp(s, "null")) s[0] = '\0';
    return s;
}

/* position of the `:` that ends a mapping key in `s` (top level: outside quotes and brackets,
 * followed by a blank or the end of the line), or NULL */
static const char *key_colon(const char *s)
{
    int depth = 0;
    char quote = 0;
    for (const char *p = s; *p; ++p) {
        if (quote) {
            if (*p == quote) quote = 0;
            continue;
        }
        if (*p == '\'' || *p == '"') { quote = *p; continue; }
        if (*p == '[' || *p == '{') { ++depth; continue; }
        if (*p == ']' || *p == '}') { --depth; continue; }
        if (*p == ':' && depth == 0 && (p[1] == '\0' || p[1] == ' ')) return p;
    }
    return NULL;
}

static int flow_balance(const char *s)
{
    int depth = 0;
    char quote = 0;
    for (const char *p = s; *p; ++p) {
        if (quote) { if (*p == quote) quote = 0; continue; }
        if (*p == '\'' || *p == '"') quote = *p;
        else if (*p == '[' || *p == '{') ++depth;
        else if (*p == ']' || *p == '}') --depth;
    }
    return depth;
}

/* flow collections: [a, b, [c]]  {k: v, k2: v2} */
static ynode *parse_flow(const char **pp)
{
    const char *p = *pp;
    while (*p == ' ') ++p;
    if (*p == '[' || *p == '{') {
        const char open = *p, close = (open == '[') ? ']' : '}';
        ynode *n = y_new(open == '[' ? Y_SEQ : Y_MAP);
        ++p;
        for (;;) {
            while (*p == ' ' || *p == ',') ++p;
            if (*p == close || *p == '\0') break;
            char *key = NULL;
            if (open == '{') {
                const char *k0 = p;
                char quote = 0;
                while (*p && (quote || (*p != ':' && *p != ',' && *p != close))) {
                    if (quote) { if (*p == quote) quote = 0; }
                    else if (*p == '\'' || *p == '"') quote = *p;
                    ++p;
                }
                key = unquote(dup_range(k0, p));
                if (*p == ':') ++p;
            }
            ynode *v = parse_flow(&p);
            y_push(n, key, v);
        }
        if (*p == close) ++p;
        *pp = p;
        return n;
    }
    /* scalar up to the next top-level , ] } */
    const char *a = p;
    char quote = 0;
    while (*p && (quote || (*p != ',' && *p != ']' && *p != '}'))) {
        if (quote) { if (*p == quote) quote = 0; }
        else if (*p == '\'' || *p == '"') quote = *p;
        ++p;
    }
    ynode *n = y_new(Y_SCALAR);
    n->text = unquote(dup_range(a, p));
    *pp = p;
    return n;
}

static ynode *parse_block(ydoc *d, int indent);

/* the value that follows `key:` or `- ` on line d->pos (text = the rest of that line) */
static ynode *parse_value(ydoc *d, const char *text, int own_indent)
{
    while (*text == ' ') ++text;
    if (*text == '\0') {
        /* nested block on the following lines: deeper, or a sequence at the key's own indentation */
        ++d->pos;
        if (d->pos < d->n) {
            const int ni = d->indent[d->pos];
            const char *nl = d->line[d->pos] + ni;
            if (ni > own_indent || (ni == own_indent && nl[0] == '-' && (nl[1] == ' ' || nl[1] == '\0')))
                return parse_block(d, ni);
        }
        ynode *n = y_new(Y_SCALAR);
        n->text = (char *)calloc(1, 1);
        return n;
    }
    if (*text == '[' || *text == '{') {
        /* flow collection, possibly wrapped over the following lines */
        size_t len = strlen(text);
        char *buf = (char *)malloc(len + 1);
        memcpy(buf, text, len + 1);
        while (flow_balance(buf) > 0 && d->pos + 1 < d->n) {
            ++d->pos;
            const char *more = d->line[d->pos] + d->indent[d->pos];
            const size_t ml = strlen(more);
            buf = (char *)realloc(buf, len + ml + 2);
            buf[len++] = ' ';
            memcpy(buf + len, more, ml + 1);
            len += ml;
        }
        const char *p = buf;
        ynode *n = parse_flow(&p);
        free(buf);
        ++d->pos;
        return n;
    }
    ynode *n = y_new(Y_SCALAR);
    n->text = unquote(dup_range(text, text + strlen(text)));
    ++d->pos;
    return n;
}

static ynode *parse_block(ydoc *d, int indent)
{
    if (d->pos >= d->n) return y_new(Y_MAP);
    const char *first = d->line[d->pos] + d->indent[d->pos];
    const int is_seq = first[0] == '-' && (first[1] == ' ' || first[1] == '\0');
    ynode *n = y_new(is_seq ? Y_SEQ : Y_MAP);
    while (d->pos < d->n && d->indent[d->pos] == indent && !d->error) {
        char *ln = d->line[d->pos] + indent;
        const int dash = ln[0] == '-' && (ln[1] == ' ' || ln[1] == '\0');
        if (dash != is_seq) break;
        if (is_seq) {
            char *rest = ln + 1;
            int extra = 1;
            while (*rest == ' ') { ++rest; ++extra; }
            if (*rest != '[' && *rest != '{' && *rest != '\'' && *rest != '"' && key_colon(rest)) {
                /* "- key: value": the item is a mapping whose first entry sits on this line */
                ln[0] = ' ';
                d->indent[d->pos] = indent + extra;
                y_push(n, NULL, parse_block(d, indent + extra));
            } else {
                y_push(n, NULL, parse_value(d, rest, indent));
            }
        } else {
            const char *colon = key_colon(ln);
            if (!colon) {
                fprintf(stderr, "scene_read_yaml: cannot parse line '%s'\n", ln);
                d->error = 1;
                break;
            }
            char *key = unquote(dup_range(ln, colon));
            y_push(n, key, parse_value(d, colon + 1, indent));
        }
    }
    return n;
}

/* splits the file into documents (frames); returns the number of documents; when `want` >= 0 the
 * lines of that document are returned in *out (caller frees with doc_free) */
static int split_documents(const char *path, int want, ydoc *out)
{
    FILE *f = fopen(path, "rb");
    if (!f) {
        perror(path);
        return -1;
    }
    int docs = 0, cur = -1, in_doc = 0, cap = 0;
    char *buf = NULL;
    size_t bufcap = 0;
    ssize_t got;
    memset(out, 0, sizeof(*out));
    while ((got = getline(&buf, &bufcap, f)) >= 0) {
        while (got > 0 && (buf[got - 1] == '\n' || buf[got - 1] == '\r')) buf[--got] = '\0';
        if (!strncmp(buf, "---", 3) && (buf[3] == '\0' || buf[3] == ' ')) {
            ++docs;
            cur = docs - 1;
            in_doc = 1;
            continue;
        }
        if (!strncmp(buf, "...", 3) && (buf[3] == '\0' || buf[3] == ' ')) {
            in_doc = 0;
            continue;
        }
        /* strip comments outside quotes, skip blank lines and directives */
        char quote = 0;
        for (char *p = buf; *p; ++p) {
            if (quote) { if (*p == quote) quote = 0; continue; }
            if (*p == '\'' || *p == '"') quote = *p;
            else if (*p == '#' && (p == buf || p[-1] == ' ')) { *p = '\0'; break; }
        }
        size_t len = strlen(buf);
        while (len > 0 && buf[len - 1] == ' ') buf[--len] = '\0';
        if (len == 0 || buf[0] == '%') continue;
        if (!in_doc) {          /* a stream that starts without a `---` marker */
            ++docs;
            cur = docs - 1;
            in_doc = 1;
        }
        if (cur != want) continue;
        if (out->n == cap) {
            cap = cap ? 2 * cap : 256;
            out->line = (char **)realloc(out->line, (size_t)cap * sizeof(char *));
            out->indent = (int *)realloc(out->indent, (size_t)cap * sizeof(int));
        }
        int ind = 0;
        while (buf[ind] == ' ') ++ind;
        out->line[out->n] = strdup(buf);
        out->indent[out->n] = ind;
        ++out->n;
    }
    free(buf);
    fclose(f);
    return docs;
}

static void doc_free(ydoc *d)
{
    for (int i = 0; i < d->n; ++i) free(d->line[i]);
    free(d->line);
    free(d->indent);
}

/* ------------------------------------------------------------------ tree -> scene */

static const char *scalar_of(const ynode *n) { return (n && n->kind == Y_SCALAR && n->text) ? n->text : ""; }

static void read_vect(const ynode *n, vectNd *vec)
{
    /* scene_yaml_parse_vect (scene.c:1179-1248): the target is re-allocated to the list's length */
    const int len = (n && n->kind == Y_SEQ) ? n->n : 0;
    vectNd_calloc(vec, len);
    for (int i = 0; i < len; ++i) vectNd_set(vec, i, atof(scalar_of(n->items[i])));
}

static void read_color(const ynode *n, double *r, double *g, double *b)
{
    if (!n || n->kind != Y_MAP) return;
    for (int i = 0; i < n->n; ++i) {
        const char *k = n->keys[i];
        const double v = atof(scalar_of(n->items[i]));
        if (!strcasecmp("red", k)) *r = v;
        else if (!strcasecmp("green", k)) *g = v;
        else if (!strcasecmp("blue", k)) *b = v;
    }
}

static object *read_object(const ynode *n)
{
    char type[OBJ_TYPE_MAX_LEN] = "unspecified", name[OBJ_NAME_MAX_LEN] = "";
    int dimensions = 0, prepared = 0, transparent = 0;
    double red = 0, green = 0, blue = 0, red_r = 0, green_r = 0, blue_r = 0, refract_index = 0, bounds_radius = 0;
    const ynode *positions = NULL, *directions = NULL, *sizes = NULL, *flags = NULL, *subs = NULL, *bounds_center = NULL;
    if (!n || n->kind != Y_MAP) return NULL;
    for (int i = 0; i < n->n; ++i) {
        const char *k = n->keys[i];
        const ynode *v = n->items[i];
        if (!strcasecmp("type", k)) snprintf(type, sizeof(type), "%s", scalar_of(v));
        else if (!strcasecmp("dimensions", k)) dimensions = atoi(scalar_of(v));
        else if (!strcasecmp("name", k)) snprintf(name, sizeof(name), "%s", scalar_of(v));
        else if (!strcasecmp("prepared", k)) prepared = atoi(scalar_of(v));
        else if (!strcasecmp("material", k)) {
            for (int m = 0; v->kind == Y_MAP && m < v->n; ++m) {
                const char *mk = v->keys[m];
                const ynode *mv = v->items[m];
                if (!strcasecmp("color", mk)) read_color(mv, &red, &green, &blue);
                else if (!strcasecmp("reflectivity", mk)) read_color(mv, &red_r, &green_r, &blue_r);
                else if (!strcasecmp("transparent", mk)) transparent = atoi(scalar_of(mv));
                else if (!strcasecmp("prepared", mk)) prepared = atoi(scalar_of(mv));
                else if (!strcasecmp("refract_index", mk)) refract_index = atof(scalar_of(mv));
                else fprintf(stderr, "scene_read_yaml: unhandled material key '%s'.\n", mk);
            }
        } else if (!strcasecmp("positions", k)) positions = v;
        else if (!strcasecmp("directions", k)) directions = v;
        else if (!strcasecmp("sizes", k)) sizes = v;
        else if (!strcasecmp("flags", k)) flags = v;
        else if (!strcasecmp("objects", k)) subs = v;
        else if (!strcasecmp("bounds", k)) {
            for (int m = 0; v->kind == Y_MAP && m < v->n; ++m) {
                if (!strcasecmp("radius", v->keys[m])) bounds_radius = atof(scalar_of(v->items[m]));
                else if (!strcasecmp("center", v->keys[m])) bounds_center = v->items[m];
            }
        } else fprintf(stderr, "scene_read_yaml: unhandled object key '%s'.\n", k);
    }
    /* sub-objects are read before the parent exists, as in the reference (scene.c:1678, 1607-1635) */
    object **children = NULL;
    int n_children = 0;
    if (subs && subs->kind == Y_SEQ) {
        children = (object **)calloc((size_t)(subs->n > 0 ? subs->n : 1), sizeof(object *));
        for (int i = 0; i < subs->n; ++i) {
            object *c = read_object(subs->items[i]);
            if (c) children[n_children++] = c;      /* object_add_obj(obj, NULL) is a no-op there */
        }
    }
    object *obj = object_alloc(dimensions, type, name);
    if (!obj) {
        fprintf(stderr, "scene_read_yaml: unknown object type '%s'.\n", type);
        for (int i = 0; i < n_children; ++i) object_free(children[i]);
        free(children);
        return NULL;
    }
    obj->red = red; obj->green = green; obj->blue = blue;
    obj->red_r = red_r; obj->green_r = green_r; obj->blue_r = blue_r;
    obj->transparent = transparent ? 1 : 0;
    obj->refract_index = refract_index;
    if (prepared) {
        /* "loading of prepared objects not fully supported" (scene.c:1706-1713): the stored bounds are kept */
        obj->prepared = 0;
        obj->bounds.radius = bounds_radius;
        if (bounds_center) {
            vectNd c;
            read_vect(bounds_center, &c);
            vectNd_copy(&obj->bounds.center, &c);
            vectNd_free(&c);
        }
    }
    snprintf(obj->name, sizeof(obj->name), "%s", name);
    for (int i = 0; positions && positions->kind == Y_SEQ && i < positions->n; ++i) {
        if (positions->items[i]->kind != Y_SEQ) continue;
        vectNd v;
        read_vect(positions->items[i], &v);
        object_add_pos(obj, &v);
        vectNd_free(&v);
    }
    for (int i = 0; directions && directions->kind == Y_SEQ && i < directions->n; ++i) {
        if (directions->items[i]->kind != Y_SEQ) continue;
        vectNd v;
        read_vect(directions->items[i], &v);
        object_add_dir(obj, &v);
        vectNd_free(&v);
    }
    for (int i = 0; sizes && sizes->kind == Y_SEQ && i < sizes->n; ++i)
        if (sizes->items[i]->kind == Y_SCALAR) object_add_size(obj, atof(scalar_of(sizes->items[i])));
    for (int i = 0; flags && flags->kind == Y_SEQ && i < flags->n; ++i)
        if (flags->items[i]->kind == Y_SCALAR) object_add_flag(obj, atoi(scalar_of(flags->items[i])));
    for (int i = 0; i < n_children; ++i) object_add_obj(obj, children[i]);
    free(children);
    if (object_validate(obj) < 0) {
        fprintf(stderr, "scene_read_yaml: loaded %s failed to validate.\n", type);
        object_free(obj);
        return NULL;
    }
    return obj;
}

static void read_light(const ynode *n, light *lgt)
{
    if (!n || n->kind != Y_MAP) return;
    for (int i = 0; i < n->n; ++i) {
        const char *k = n->keys[i];
        const ynode *v = n->items[i];
        if (!strcasecmp("type", k)) {
            const char *t = scalar_of(v);
            if (!strncasecmp(t, "LIGHT_", 6)) t += 6;
            if (!strcasecmp(t, "AMBIENT")) lgt->type = LIGHT_AMBIENT;
            else if (!strcasecmp(t, "POINT")) lgt->type = LIGHT_POINT;
            else if (!strcasecmp(t, "DIRECTIONAL")) lgt->type = LIGHT_DIRECTIONAL;
            else if (!strcasecmp(t, "SPOT")) lgt->type = LIGHT_SPOT;
            else if (!strcasecmp(t, "DISK")) lgt->type = LIGHT_DISK;
            else if (!strcasecmp(t, "RECT")) lgt->type = LIGHT_RECT;
            else fprintf(stderr, "scene_read_yaml: unknown light type '%s'\n", scalar_of(v));
        } else if (!strcasecmp("color", k)) {
            double r = 0.0, g = 0.0, b = 0.0;
            read_color(v, &r, &g, &b);
            lgt->red = r; lgt->green = g; lgt->blue = b;
        } else if (!strcasecmp("name", k)) snprintf(lgt->name, sizeof(lgt->name), "%s", scalar_of(v));
        else if (!strcasecmp("pos", k)) read_vect(v, &lgt->pos);
        else if (!strcasecmp("target", k)) read_vect(v, &lgt->target);
        else if (!strcasecmp("dir", k)) read_vect(v, &lgt->dir);
        else if (!strcasecmp("radius", k)) lgt->radius = atof(scalar_of(v));
        else if (!strcasecmp("u", k)) read_vect(v, &lgt->u);
        else if (!strcasecmp("v", k)) read_vect(v, &lgt->v);
        else if (!strcasecmp("prepared", k)) lgt->prepared = atoi(scalar_of(v)) ? 1 : 0;
        else if (!strcasecmp("u1", k)) read_vect(v, &lgt->u1);
        else if (!strcasecmp("v1", k)) read_vect(v, &lgt->v1);
        else fprintf(stderr, "scene_read_yaml: unhandled light key '%s'.\n", k);    /* includes `angle`, as in the reference */
    }
}

static void read_camera(const ynode *n, camera *cam)
{
    if (!n || n->kind != Y_MAP) return;
    for (int i = 0; i < n->n; ++i) {
        const char *k = n->keys[i];
        const ynode *v = n->items[i];
        if (!strcasecmp("viewPoint", k)) read_vect(v, &cam->viewPoint);
        else if (!strcasecmp("viewTarget", k)) read_vect(v, &cam->viewTarget);
        else if (!strcasecmp("up", k)) read_vect(v, &cam->up);
        else if (!strcasecmp("rotation", k)) cam->rotation = atof(scalar_of(v));
        else if (!strcasecmp("eye_offset", k)) cam->eye_offset = atof(scalar_of(v));
        else if (!strcasecmp("aperture_radius", k)) cam->aperture_radius = atof(scalar_of(v));
        else if (!strcasecmp("focal_distance", k)) cam->focal_distance = atof(scalar_of(v));
        else if (!strcasecmp("zoom", k)) cam->zoom = atof(scalar_of(v));
        else if (!strcasecmp("type", k)) {
            if (!strcasecmp(scalar_of(v), "vr")) cam->type = CAMERA_VR;
            else if (!strcasecmp(scalar_of(v), "pano")) cam->type = CAMERA_PANO;
            else cam->type = CAMERA_NORMAL;
        } else if (!strcasecmp("hFov", k)) cam->hFov = atof(scalar_of(v));
        else if (!strcasecmp("vFov", k)) cam->vFov = atof(scalar_of(v));
        else if (!strcasecmp("flip_x", k)) cam->flip_x = atoi(scalar_of(v)) ? 1 : 0;
        else if (!strcasecmp("flip_y", k)) cam->flip_y = atoi(scalar_of(v)) ? 1 : 0;
        else if (!strcasecmp("flatten", k)) cam->flatten = atoi(scalar_of(v)) ? 1 : 0;
        else if (!strcasecmp("prepared", k)) cam->prepared = atoi(scalar_of(v)) ? 1 : 0;
        else if (!strcasecmp("leveling", k)) cam->leveling = atof(scalar_of(v));
        else if (!strcasecmp("pos", k)) read_vect(v, &cam->pos);
        else if (!strcasecmp("leftEye", k)) read_vect(v, &cam->leftEye);
        else if (!strcasecmp("rightEye", k)) read_vect(v, &cam->rightEye);
        else if (!strcasecmp("dirX", k)) read_vect(v, &cam->dirX);
        else if (!strcasecmp("dirY", k)) read_vect(v, &cam->dirY);
        else if (!strcasecmp("imgOrig", k)) read_vect(v, &cam->imgOrig);
        else if (!strcasecmp("localX", k)) read_vect(v, &cam->localX);
        else if (!strcasecmp("localY", k)) read_vect(v, &cam->localY);
        else if (!strcasecmp("localZ", k)) read_vect(v, &cam->localZ);
        else fprintf(stderr, "scene_read_yaml: unhandled camera key '%s'.\n", k);
    }
}

int scene_yaml_count_frames(char *fname)
{
    ydoc d;
    const int docs = split_documents(fname, -1, &d);
    doc_free(&d);
    return docs < 0 ? 0 : docs;
}

int scene_read_yaml(scene *scn, char *fname, int frame)
{
    printf("%s reading from '%s'.\n", __FUNCTION__, fname);
    ydoc d;
    const int docs = split_documents(fname, frame, &d);
    if (docs < 0) exit(1);                      /* the reference exits when the file cannot be opened (scene.c:2097) */
    if (frame >= docs || d.n == 0) {
        doc_free(&d);
        return 0;                               /* past the last document: nothing is read */
    }
    ynode *root = parse_block(&d, d.indent[0]);
    if (!d.error && root->kind == Y_MAP) {
        for (int i = 0; i < root->n; ++i) {
            const char *k = root->keys[i];
            const ynode *v = root->items[i];
            if (!strcasecmp("lights", k)) {
                for (int j = 0; v->kind == Y_SEQ && j < v->n; ++j) {
                    if (v->items[j]->kind != Y_MAP) continue;
                    light *lgt = NULL;
                    scene_alloc_light(scn, &lgt);
                    read_light(v->items[j], lgt);
                }
            } else if (!strcasecmp("objects", k)) {
                for (int j = 0; v->kind == Y_SEQ && j < v->n; ++j) {
                    if (v->items[j]->kind != Y_MAP) continue;
                    object *obj = read_object(v->items[j]);
                    if (obj) scene_add_object(scn, obj);
                }
            } else if (!strcasecmp("camera", k)) read_camera(v, &scn->cam);
            else if (!strcasecmp("dimensions", k)) scn->dimensions = atoi(scalar_of(v));
            else if (!strcasecmp("scene", k)) snprintf(scn->name, sizeof(scn->name), "%s", scalar_of(v));
            else if (!strcasecmp("background", k)) read_color(v, &scn->bg_red, &scn->bg_green, &scn->bg_blue);
            else fprintf(stderr, "scene_read_yaml: unhandled key '%s'.\n", k);
        }
    }
    const int failed = d.error;
    y_free(root);
    doc_free(&d);
    return failed ? -1 : 0;
}

/* ------------------------------------------------------------------ writer
 * Same keys, same conditions and the same %.16g as scene_yaml_emit_scene (scene.c:737-1020); the
 * line layout is this writer's own (libyaml wraps long flow sequences at 80 columns, this does not). */

static void put_vect(FILE *f, const char *indent, const char *key, const vectNd *v)
{
    fprintf(f, "%s%s: [", indent, key);
    for (int i = 0; i < v->n; ++i) fprintf(f, "%s%.16g", i ? ", " : "", v->v[i]);
    fprintf(f, "]\n");
}

static void put_string(FILE *f, const char *indent, const char *key, const char *value)
{
    /* quote what would not read back as the same plain scalar */
    int plain = value[0] != '\0' && !isspace((unsigned char)value[0]);
    for (const char *p = value; *p; ++p)
        if (*p == ':' || *p == '#' || *p == '\'' || *p == '"' || *p == '[' || *p == ']' || *p == '{' || *p == '}' || *p == ',')
            plain = 0;
    if (value[0] == '\0') fprintf(f, "%s%s:\n", indent, key);
    else if (plain) fprintf(f, "%s%s: %s\n", indent, key, value);
    else {
        fprintf(f, "%s%s: '", indent, key);
        for (const char *p = value; *p; ++p) {
            if (*p == '\'') fputc('\'', f);
            fputc(*p, f);
        }
        fprintf(f, "'\n");
    }
}

static void put_object(FILE *f, object *obj, int depth)
{
    char pad[64], first[64];
    snprintf(pad, sizeof(pad), "%*s", 2 * depth + 2, "");
    snprintf(first, sizeof(first), "%*s- ", 2 * depth, "");
    char type[OBJ_TYPE_MAX_LEN];
    obj->type_name(type, sizeof(type));
    int lead = 1;
#define LEAD() (lead ? (lead = 0, first) : pad)
    if (obj->name[0]) put_string(f, LEAD(), "name", obj->name);
    put_string(f, LEAD(), "type", type);
    fprintf(f, "%sdimensions: %d\n", pad, obj->dimensions);
    fprintf(f, "%smaterial:\n", pad);
    if (obj->transparent) {
        fprintf(f, "%s  transparent: %d\n", pad, obj->transparent);
        fprintf(f, "%s  refract_index: %.16g\n", pad, obj->refract_index);
    }
    fprintf(f, "%s  color: {red: %.16g, green: %.16g, blue: %.16g}\n", pad, obj->red, obj->green, obj->blue);
    if (obj->red_r != 0 && obj->green_r != 0 && obj->blue_r != 0)       /* all three, scene.c:878 */
        fprintf(f, "%s  reflectivity: {red: %.16g, green: %.16g, blue: %.16g}\n", pad, obj->red_r, obj->green_r, obj->blue_r);
    char pad2[64];
    snprintf(pad2, sizeof(pad2), "%s- ", pad);
    if (obj->n_pos > 0) {
        fprintf(f, "%spositions:\n", pad);
        for (int i = 0; i < obj->n_pos; ++i) {
            fprintf(f, "%s- [", pad);
            for (int c = 0; c < obj->pos[i].n; ++c) fprintf(f, "%s%.16g", c ? ", " : "", obj->pos[i].v[c]);
            fprintf(f, "]\n");
        }
    }
    if (obj->n_dir > 0) {
        fprintf(f, "%sdirections:\n", pad);
        for (int i = 0; i < obj->n_dir; ++i) {
            fprintf(f, "%s- [", pad);
            for (int c = 0; c < obj->dir[i].n; ++c) fprintf(f, "%s%.16g", c ? ", " : "", obj->dir[i].v[c]);
            fprintf(f, "]\n");
        }
    }
    if (obj->n_size > 0) {
        fprintf(f, "%ssizes: [", pad);
        for (int i = 0; i < obj->n_size; ++i) fprintf(f, "%s%.16g", i ? ", " : "", obj->size[i]);
        fprintf(f, "]\n");
    }
    if (obj->n_flag > 0) {
        fprintf(f, "%sflags: [", pad);
        for (int i = 0; i < obj->n_flag; ++i) fprintf(f, "%s%d", i ? ", " : "", obj->flag[i]);
        fprintf(f, "]\n");
    }
    if (obj->n_obj > 0) {
        fprintf(f, "%sobjects:\n", pad);
        for (int i = 0; i < obj->n_obj; ++i) put_object(f, obj->obj[i], depth + 1);
    }
#undef LEAD
}

int scene_write_yaml(scene *scn, char *fname)
{
    FILE *f = fopen(fname, "wb");
    if (!f) {
        perror("fopen");
        return -1;
    }
    fprintf(f, "---\n");
    put_string(f, "", "scene", scn->name);
    fprintf(f, "dimensions: %d\n", scn->dimensions);
    if (scn->bg_red != 0 || scn->bg_green != 0 || scn->bg_blue != 0)
        fprintf(f, "background: {red: %.16g, green: %.16g, blue: %.16g}\n", scn->bg_red, scn->bg_green, scn->bg_blue);
    camera *cam = &scn->cam;
    fprintf(f, "camera:\n");
    put_vect(f, "  ", "viewPoint", &cam->viewPoint);
    put_vect(f, "  ", "viewTarget", &cam->viewTarget);
    if (cam->up.n > 0) {
        int any = 0;
        for (int i = 0; i < cam->up.n; ++i) any |= cam->up.v[i] != 0;
        if (any) put_vect(f, "  ", "up", &cam->up);
    }
    if (cam->rotation != 0) fprintf(f, "  rotation: %.16g\n", cam->rotation);
    if (cam->eye_offset != 0 && cam->eye_offset != EYE_OFFSET) fprintf(f, "  eye_offset: %.16g\n", cam->eye_offset);
    if (cam->flip_x) fprintf(f, "  flip_x: %d\n", cam->flip_x);
    if (cam->flip_y) fprintf(f, "  flip_y: %d\n", cam->flip_y);
    if (cam->zoom != 1.0 && cam->zoom != 0.0) fprintf(f, "  zoom: %.16g\n", cam->zoom);
    if (cam->type != CAMERA_NORMAL) {
        fprintf(f, "  type: %s\n", cam->type == CAMERA_VR ? "vr" : "pano");
        fprintf(f, "  vFov: %.16g\n", cam->vFov);
        fprintf(f, "  hFov: %.16g\n", cam->hFov);
    }
    if (cam->aperture_radius != 0) {
        fprintf(f, "  aperture_radius: %.16g\n", cam->aperture_radius);
        fprintf(f, "  focal_distance: %.16g\n", cam->focal_distance);
    }
    fprintf(f, "lights:\n");
    for (int i = 0; i < scn->num_lights; ++i) {
        light *l = scn->lights[i];
        fprintf(f, "- type: %s\n", LIGHT_TYPE_STRING[l->type]);
        put_string(f, "  ", "name", l->name);
        fprintf(f, "  color: {red: %.16g, green: %.16g, blue: %.16g}\n", l->red, l->green, l->blue);
        if (l->type == LIGHT_POINT || l->type == LIGHT_SPOT || l->type == LIGHT_DISK || l->type == LIGHT_RECT)
            put_vect(f, "  ", "pos", &l->pos);
        if (l->type == LIGHT_DIRECTIONAL || l->type == LIGHT_SPOT) put_vect(f, "  ", "dir", &l->dir);
        if (l->type == LIGHT_DISK || l->type == LIGHT_RECT) {
            put_vect(f, "  ", "u", &l->u);
            put_vect(f, "  ", "v", &l->v);
        }
        if (l->type == LIGHT_DISK) fprintf(f, "  radius: %.16g\n", l->radius);
        if (l->type == LIGHT_SPOT) fprintf(f, "  angle: %.16g\n", l->angle);
    }
    fprintf(f, "objects:\n");
    for (int i = 0; i < scn->num_objects; ++i) put_object(f, scn->object_ptrs[i], 0);
    fclose(f);
    return 0;
}
