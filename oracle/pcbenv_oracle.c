/*
 * pcbenv_oracle.c -- CPU restatement of the reference environments' hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may load this library; the product
 * (rl-environment-for-component-placement_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 * (a) every known-answer test the reference's own suite holds for the path
 *     (/root/reference/tests/{square,rectangular,pin}_environment/*.py, restated
 *     in tests/test_reference_kats.py) and
 * (b) golden episodes recorded from the unmodified reference Python
 *     environments in the build container (tests/golden/make_golden.py).
 *
 * The code is deliberately plain, scalar, one environment at a time, with the
 * observation arrays kept in the reference's own shapes and float64 values.
 * Every function cites the reference lines it follows; `S:` means
 * environment/dummy_env_rectangular_pin_spatial.py, `P:`
 * environment/dummy_env_rectangular_pin.py, `R:`
 * environment/dummy_env_rectangular.py, `Q:` environment/dummy_env_square.py.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off: one IEEE operation
 * per written operator; the single fused multiply-add of the path is the
 * explicit fma() in orc_norm2, SURVEY.md trap T1).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_KIND_SQUARE 0
#define ORC_KIND_RECT 1
#define ORC_KIND_PIN 2
#define ORC_KIND_SPATIAL 3

#define ORC_REWARD_BEAM 0
#define ORC_REWARD_CENTROID 1
#define ORC_REWARD_BOTH 2

#define ORC_MAX_PPN 64 /* pins per net the route code accepts */

typedef struct orc_config {
    int32_t kind, height, width;
    int32_t min_component_w, max_component_w, min_component_h, max_component_h;
    int32_t max_num_components, min_num_components;
    int32_t net_distribution, pin_spread;
    int32_t min_num_nets, max_num_nets, max_num_pins_per_net, min_num_pins_per_net;
    int32_t reward_type, reward_beam_width, component_n;
    double weight_wirelength, weight_num_intersections;
} orc_config;

typedef struct {
    int rel_x, rel_y, abs_x, abs_y, pin_id, comp_id, net_id;
} orc_pin;

typedef struct {
    int h, w, id, placed, pos_x, pos_y;
} orc_comp;

typedef struct orc_env {
    orc_config c;
    int H, W, O;          /* O = leading dim of action_mask (1, 2 or 4) */
    int C, mp, N, F;      /* max comps, max pins/comp, max nets, component feature width */
    int pin_rows, pin_cat_w;
    /* instance */
    int ncomp, nnets, npins, cur; /* cur = index of current component, -1 = sentinel Component(-1,-1,-1) */
    orc_comp *comps;
    orc_pin *pins;         /* order of the reference's self.pins (net-major) */
    int *net_start;        /* nnets+1 offsets into pins */
    /* observation arrays, reference shapes, float64 */
    double *grid;          /* H*W */
    double *action_mask;   /* O*H*W */
    double *comp_feat;     /* C*F */
    double *placement_mask;/* C */
    double *component_mask;/* C (rect) */
    double *pins_num;      /* pin: C*mp*4 ; spatial: (C*mp+1)*4 */
    double *pins_cat;      /* pin: C*mp*1 ; spatial: (C*mp+1)*2 */
    double *pin_grid;      /* spatial: H*W*(N+1) */
    double *component_grid;/* spatial: C*mh*mw*(N+1); rows >= ncomp are zero (reference has only ncomp rows) */
    double reward_wirelength, reward_intersection;
    double max_wirelength, max_num_intersections;
} orc_env;

/* ------------------------------------------------------------------------- */
/* geometry + routing primitives                                             */
/* ------------------------------------------------------------------------- */

/* S:1288-1301 euclidean_distance = np.linalg.norm(p1 - p2).  norm() is
 * sqrt(ddot(d, d)); OpenBLAS' ddot tail loop is fma-contracted, so for the
 * length-2 vector the value is sqrt(fma(dy, dy, dx*dx)) (verified
 * bit-for-bit in the build container; the golden fixtures pin it). */
double orc_norm2(double dx, double dy) { return sqrt(fma(dy, dy, dx * dx)); }

double orc_euclidean_distance(double x1, double y1, double x2, double y2) {
    return orc_norm2(x1 - x2, y1 - y2);
}

/* S:653-702 is_intersect.  Segments are (x1,y1)-(x2,y2) and (x3,y3)-(x4,y4).
 * Integer operands of the reference are exact in float64 at these magnitudes,
 * so one float64 operation per written operator reproduces both its int and its
 * mixed int/float arithmetic. */
int orc_is_intersect(const double *a, const double *b) {
    double x1 = a[0], y1 = a[1], x2 = a[2], y2 = a[3];
    double x3 = b[0], y3 = b[1], x4 = b[2], y4 = b[3];
    if ((x1 == x3 && y1 == y3) || (x1 == x4 && y1 == y4) || (x2 == x3 && y2 == y3) ||
        (x2 == x4 && y2 == y4))
        return 1; /* :674-680 shared end point */
    double det = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4); /* :683 */
    if (det == 0) return 0;
    double x = ((x1 * y2 - y1 * x2) * (x3 - x4) - (x1 - x2) * (x3 * y4 - y3 * x4)) / det; /* :690 */
    double y = ((x1 * y2 - y1 * x2) * (y3 - y4) - (y1 - y2) * (x3 * y4 - y3 * x4)) / det; /* :691 */
    double lo, hi;
    lo = x1 < x2 ? x1 : x2; hi = x1 > x2 ? x1 : x2; if (!(lo <= x && x <= hi)) return 0;
    lo = x3 < x4 ? x3 : x4; hi = x3 > x4 ? x3 : x4; if (!(lo <= x && x <= hi)) return 0;
    lo = y1 < y2 ? y1 : y2; hi = y1 > y2 ? y1 : y2; if (!(lo <= y && y <= hi)) return 0;
    lo = y3 < y4 ? y3 : y4; hi = y3 > y4 ? y3 : y4; if (!(lo <= y && y <= hi)) return 0;
    return 1;
}

/* A route: segments seg[4*i..] grouped by net via net_off[0..nnets]. */
/* S:629-651 find_num_intersection */
int orc_find_num_intersection(const double *seg, const int *net_off, int nnets) {
    int count = 0;
    for (int n = 0; n < nnets; n++)
        for (int m = n + 1; m < nnets; m++)
            for (int i = net_off[n]; i < net_off[n + 1]; i++)
                for (int j = net_off[m]; j < net_off[m + 1]; j++)
                    if (orc_is_intersect(seg + 4 * i, seg + 4 * j)) count++;
    return count;
}

/* S:704-722 find_wirelength: sequential float sum, nets then segments. */
double orc_find_wirelength(const double *seg, const int *net_off, int nnets) {
    double wl = 0.0;
    for (int i = net_off[0]; i < net_off[nnets]; i++)
        wl += orc_euclidean_distance(seg[4 * i], seg[4 * i + 1], seg[4 * i + 2], seg[4 * i + 3]);
    return wl;
}

/* S:1229-1241 get_centroid = np.mean(int array, axis=0): exact integer sums, one division. */
void orc_get_centroid(const int *pts, int n, double *cx, double *cy) {
    double sx = 0, sy = 0;
    for (int i = 0; i < n; i++) { sx += pts[2 * i]; sy += pts[2 * i + 1]; }
    *cx = sx / n; *cy = sy / n;
}

/* S:1243-1271 route_pins_centroid for one net -> segments appended at seg; returns count. */
int orc_route_centroid_net(const int *pts, int n, double *seg) {
    if (n == 2) {
        seg[0] = pts[0]; seg[1] = pts[1]; seg[2] = pts[2]; seg[3] = pts[3];
        return 1;
    }
    double cx, cy;
    orc_get_centroid(pts, n, &cx, &cy);
    for (int i = 0; i < n; i++) {
        seg[4 * i] = pts[2 * i]; seg[4 * i + 1] = pts[2 * i + 1];
        seg[4 * i + 2] = cx; seg[4 * i + 3] = cy;
    }
    return n;
}

/* S:1273-1286 pin_outlier: first arg-max of distance to centroid. */
int orc_pin_outlier(const int *pts, int n) {
    double cx, cy;
    orc_get_centroid(pts, n, &cx, &cy);
    int best = 0; double bd = -1.0;
    for (int i = 0; i < n; i++) {
        double d = orc_norm2(pts[2 * i] - cx, pts[2 * i + 1] - cy);
        if (i == 0 || d > bd) { bd = d; best = i; }
    }
    return best;
}

/* ---- CPython 3.10 set-iteration-order model (SURVEY.md trap T2) ----------
 * beam_search (S:1354-1357) sorts `points_to_visit - visited` with a stable
 * sort, so equal-distance neighbours keep the iteration order of that
 * temporary set.  That order is a pure function of the tuple hashes and of
 * Objects/setobject.c's table mechanics, modelled here for sets of (x, y)
 * tuples of small non-negative ints. */
#define CS_LINEAR_PROBES 9
#define CS_PERTURB_SHIFT 5
#define CS_MAXTAB 256
typedef struct { uint64_t hash; int key; int state; /* 0 unused, 1 active, 2 dummy */ } cs_entry;
typedef struct { int mask, fill, used; cs_entry t[CS_MAXTAB]; } cs_set;

/* Objects/tupleobject.c tuplehash (xxHash-style, CPython >= 3.8); hash(int) = int. */
uint64_t orc_tuple_hash2(int64_t x, int64_t y) {
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
    uint64_t acc = P5;
    uint64_t lanes[2] = {(uint64_t)x, (uint64_t)y};
    for (int i = 0; i < 2; i++) {
        acc += lanes[i] * P2;
        acc = (acc << 31) | (acc >> 33);
        acc *= P1;
    }
    acc += 2ULL ^ (P5 ^ 3527539ULL);
    if (acc == (uint64_t)-1) return 1546275796ULL;
    return acc;
}

static void cs_init(cs_set *s, int size) {
    s->mask = size - 1; s->fill = 0; s->used = 0;
    memset(s->t, 0, sizeof(cs_entry) * (size_t)size);
}
/* set_insert_clean: table known to have no dummies and no equal key. */
static void cs_insert_clean(cs_set *s, int key, uint64_t hash) {
    uint64_t perturb = hash; size_t mask = (size_t)s->mask; size_t i = (size_t)hash & mask;
    for (;;) {
        if (s->t[i].state == 0) break;
        if (i + CS_LINEAR_PROBES <= mask) {
            size_t j, f = 0;
            for (j = 1; j <= CS_LINEAR_PROBES; j++)
                if (s->t[i + j].state == 0) { i = i + j; f = 1; break; }
            if (f) break;
        }
        perturb >>= CS_PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & mask;
    }
    s->t[i].state = 1; s->t[i].key = key; s->t[i].hash = hash;
}
static void cs_resize(cs_set *s, int minused) {
    int newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    cs_set old = *s;
    cs_init(s, newsize);
    for (int i = 0; i <= old.mask; i++)
        if (old.t[i].state == 1) cs_insert_clean(s, old.t[i].key, old.t[i].hash);
    s->fill = s->used = old.used;
}
/* set_add_entry for a key known to be absent (all our keys are distinct points). */
static void cs_add(cs_set *s, int key, uint64_t hash) {
    size_t mask = (size_t)s->mask; size_t i = (size_t)hash & mask; uint64_t perturb = hash;
    int freeslot = -1;
    for (;;) {
        size_t probes = (i + CS_LINEAR_PROBES <= mask) ? CS_LINEAR_PROBES : 0;
        size_t e = i; int found = 0;
        for (size_t p = 0; p <= probes; p++, e++) {
            if (s->t[e].state == 0) { i = e; found = 1; break; }
            if (s->t[e].state == 2) freeslot = (int)e;
        }
        if (found) break;
        perturb >>= CS_PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & mask;
    }
    if (freeslot >= 0) {
        s->used++; s->t[freeslot].state = 1; s->t[freeslot].key = key; s->t[freeslot].hash = hash;
        return;
    }
    s->fill++; s->used++;
    s->t[i].state = 1; s->t[i].key = key; s->t[i].hash = hash;
    if ((size_t)s->fill * 5 < mask * 3) return;
    cs_resize(s, s->used * 4);
}
static void cs_discard(cs_set *s, int key, uint64_t hash) {
    size_t mask = (size_t)s->mask; size_t i = (size_t)hash & mask; uint64_t perturb = hash;
    for (;;) {
        size_t probes = (i + CS_LINEAR_PROBES <= mask) ? CS_LINEAR_PROBES : 0;
        size_t e = i;
        for (size_t p = 0; p <= probes; p++, e++) {
            if (s->t[e].state == 0) return;
            if (s->t[e].state == 1 && s->t[e].key == key) { s->t[e].state = 2; s->used--; return; }
        }
        perturb >>= CS_PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & mask;
    }
}
/* set_copy -> set_merge into an empty set. */
static void cs_copy(cs_set *dst, const cs_set *src) {
    cs_init(dst, 8);
    if ((dst->fill + src->used) * 5 >= dst->mask * 3) {
        int minused = (dst->used + src->used) * 2, newsize = 8;
        while (newsize <= minused) newsize <<= 1;
        cs_init(dst, newsize);
    }
    if (dst->mask == src->mask && src->fill == src->used) {
        memcpy(dst->t, src->t, sizeof(cs_entry) * (size_t)(src->mask + 1));
        dst->fill = src->fill; dst->used = src->used;
        return;
    }
    for (int i = 0; i <= src->mask; i++)
        if (src->t[i].state == 1) cs_insert_clean(dst, src->t[i].key, src->t[i].hash);
    dst->fill = dst->used = src->used;
}

/* Iteration order of `set(points) - visited` where points[0..n) are distinct
 * (x, y) tuples inserted in list order and `visited` is the subset given by
 * bit mask.  out[] receives point indices; returns their number. */
int orc_set_difference_order(const int *pts, int n, uint32_t visited_mask, int *out) {
    static __thread cs_set A, R;
    uint64_t h[ORC_MAX_PPN];
    cs_init(&A, 8);
    for (int i = 0; i < n; i++) { h[i] = orc_tuple_hash2(pts[2 * i], pts[2 * i + 1]); cs_add(&A, i, h[i]); }
    int nb = __builtin_popcount(visited_mask);
    if ((n >> 2) > nb) { /* set_copy_and_difference */
        cs_copy(&R, &A);
        for (int i = 0; i < n; i++) if (visited_mask >> i & 1) cs_discard(&R, i, h[i]);
        /* set_difference_update_internal: "if more than 1/4th are dummies, resize them away" */
        if ((size_t)(R.fill - R.used) > (size_t)R.mask / 4) cs_resize(&R, R.used * 4);
    } else {
        cs_init(&R, 8);
        for (int i = 0; i <= A.mask; i++)
            if (A.t[i].state == 1 && !(visited_mask >> A.t[i].key & 1)) cs_add(&R, A.t[i].key, A.t[i].hash);
    }
    int m = 0;
    for (int i = 0; i <= R.mask; i++) if (R.t[i].state == 1) out[m++] = R.t[i].key;
    return m;
}

/* S:1303-1369 beam_search.  pts[0..n) = points to visit (start excluded),
 * path_out receives 1+n entries: -1 for the start point, else an index into pts.
 * heapq on (priority, path, visited) tuples pops in ascending (priority, path)
 * order; paths within a level are pairwise different, so a selection of the
 * minimum reproduces it. */
typedef struct { double prio; int len; int path[ORC_MAX_PPN + 1]; uint32_t visited; } bs_entry;

static int bs_path_less(const bs_entry *a, const bs_entry *b, const int *pts, int sx, int sy) {
    if (a->prio != b->prio) return a->prio < b->prio;
    int n = a->len < b->len ? a->len : b->len;
    for (int i = 0; i < n; i++) {
        int ax = a->path[i] < 0 ? sx : pts[2 * a->path[i]], ay = a->path[i] < 0 ? sy : pts[2 * a->path[i] + 1];
        int bx = b->path[i] < 0 ? sx : pts[2 * b->path[i]], by = b->path[i] < 0 ? sy : pts[2 * b->path[i] + 1];
        if (ax != bx) return ax < bx;
        if (ay != by) return ay < by;
    }
    return a->len < b->len;
}

int orc_beam_search(int sx, int sy, const int *pts, int n, int beam_width, int *path_out) {
    if (n > 31 || beam_width < 1) return -1;
    int cap = beam_width * beam_width + 1;
    bs_entry *queue = (bs_entry *)malloc(sizeof(bs_entry) * (size_t)cap);
    bs_entry *next = (bs_entry *)malloc(sizeof(bs_entry) * (size_t)cap);
    char *taken = (char *)malloc((size_t)cap);
    int qn = 1, result = -1;
    uint32_t all = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
    queue[0].prio = 0; queue[0].len = 1; queue[0].path[0] = -1; queue[0].visited = 0;
    while (qn > 0 && result < 0) {
        int nn = 0;
        memset(taken, 0, (size_t)cap);
        int pops = beam_width < qn ? beam_width : qn;
        for (int t = 0; t < pops && result < 0; t++) {
            int best = -1; /* heappop = minimum of what is left */
            for (int i = 0; i < qn; i++)
                if (!taken[i] && (best < 0 || bs_path_less(&queue[i], &queue[best], pts, sx, sy))) best = i;
            taken[best] = 1;
            bs_entry *e = &queue[best];
            if (e->visited == all) { /* :1350 */
                for (int i = 0; i < e->len; i++) path_out[i] = e->path[i];
                result = e->len;
                break;
            }
            int cur = e->path[e->len - 1];
            int cx = cur < 0 ? sx : pts[2 * cur], cy = cur < 0 ? sy : pts[2 * cur + 1];
            int order[ORC_MAX_PPN]; double dist[ORC_MAX_PPN];
            int m = orc_set_difference_order(pts, n, e->visited, order);
            for (int i = 0; i < m; i++)
                dist[i] = orc_euclidean_distance(cx, cy, pts[2 * order[i]], pts[2 * order[i] + 1]);
            /* sorted(..., key=dist): stable insertion sort */
            for (int i = 1; i < m; i++) {
                int o = order[i]; double d = dist[i]; int j = i - 1;
                while (j >= 0 && dist[j] > d) { order[j + 1] = order[j]; dist[j + 1] = dist[j]; j--; }
                order[j + 1] = o; dist[j + 1] = d;
            }
            int take = m < beam_width ? m : beam_width;
            for (int i = 0; i < take; i++) { /* :1361-1367 */
                bs_entry *q = &next[nn++];
                *q = *e;
                q->path[q->len++] = order[i];
                q->visited |= 1u << order[i];
                q->prio = e->prio + orc_euclidean_distance(pts[2 * order[i]], pts[2 * order[i] + 1], cx, cy);
            }
        }
        if (result < 0) { bs_entry *tmp = queue; queue = next; next = tmp; qn = nn; }
    }
    free(queue); free(next); free(taken);
    return result;
}

/* S:1371-1406 route_pins_beam_search for one net -> segments; returns count. */
int orc_route_beam_net(const int *pts_in, int n, int beam_width, double *seg) {
    int pts[2 * ORC_MAX_PPN], path[ORC_MAX_PPN + 1];
    int s = orc_pin_outlier(pts_in, n);
    int sx = pts_in[2 * s], sy = pts_in[2 * s + 1];
    int m = 0;
    for (int i = 0; i < n; i++) { /* list.remove(start): first equal element (== index s for distinct points) */
        if (i == s) continue;
        pts[2 * m] = pts_in[2 * i]; pts[2 * m + 1] = pts_in[2 * i + 1]; m++;
    }
    int len = orc_beam_search(sx, sy, pts, m, beam_width, path);
    if (len < 0) return -1;
    for (int i = 0; i + 1 < len; i++) {
        int a = path[i], b = path[i + 1];
        seg[4 * i] = a < 0 ? sx : pts[2 * a]; seg[4 * i + 1] = a < 0 ? sy : pts[2 * a + 1];
        seg[4 * i + 2] = b < 0 ? sx : pts[2 * b]; seg[4 * i + 3] = b < 0 ? sy : pts[2 * b + 1];
    }
    return len - 1;
}

/* Route all nets.  pts: concatenated (x,y) per net, pt_off[0..nnets]. */
int orc_route(const int *pts, const int *pt_off, int nnets, int method, int beam_width,
              double *seg, int *seg_off) {
    int ns = 0;
    seg_off[0] = 0;
    for (int n = 0; n < nnets; n++) {
        int cnt = pt_off[n + 1] - pt_off[n];
        int k = method == ORC_REWARD_BEAM ? orc_route_beam_net(pts + 2 * pt_off[n], cnt, beam_width, seg + 4 * ns)
                                          : orc_route_centroid_net(pts + 2 * pt_off[n], cnt, seg + 4 * ns);
        if (k < 0) return -1;
        ns += k;
        seg_off[n + 1] = ns;
    }
    return ns;
}

/* ------------------------------------------------------------------------- */
/* constants of the reward                                                   */
/* ------------------------------------------------------------------------- */
/* S:724-746 / P:757-783 */
double orc_upper_bound_wirelength(const orc_config *c) {
    double distance = orc_euclidean_distance(0, 0, c->height, c->width);
    double total = 0.5 * distance * (double)(c->max_num_nets * c->max_num_pins_per_net);
    return c->kind == ORC_KIND_SPATIAL ? total / (double)(c->height + c->width) : total;
}
/* S:748-791 (float) / P:785-830 (int(...)) */
double orc_upper_bound_intersections(const orc_config *c) {
    double v = 0.5 * (double)(c->max_num_pins_per_net * c->max_num_pins_per_net) * (double)c->max_num_nets *
               (double)(c->max_num_nets - 1);
    return c->kind == ORC_KIND_PIN ? (double)(long long)v : v;
}
static double mean2(int a, int b) { return (double)(a + b) / 2.0; }
/* S:840-850 */
double orc_intersections_norm(const orc_config *c) {
    double a = mean2(c->min_component_h, c->max_component_h) * mean2(c->min_component_w, c->max_component_w) *
               mean2(c->min_num_components, c->max_num_components);
    double b = mean2(c->min_num_pins_per_net, c->max_num_pins_per_net) * mean2(c->min_num_nets, c->max_num_nets);
    return a < b ? a : b;
}

/* S:793-929 find_reward on explicit net point lists (also serves the reference's
 * hand-built KATs, tests/pin_environment/test_env.py:199-391).
 * out[0] = reward, out[1] = reward_wirelength, out[2] = reward_intersection. */
int orc_find_reward_pts(const orc_config *c, int placed_all, const int *pts, const int *pt_off, int nnets,
                        double *out) {
    double wl_norm = (double)(c->height + c->width);
    double int_norm = orc_intersections_norm(c);
    if (!placed_all) { /* :853-863 */
        double max_wl = orc_upper_bound_wirelength(c), max_int = orc_upper_bound_intersections(c);
        out[0] = -c->weight_wirelength * (max_wl / wl_norm) - c->weight_num_intersections * (max_int / int_norm);
        out[1] = max_wl; out[2] = max_int;
        return 0;
    }
    int total = pt_off[nnets];
    double *seg = (double *)malloc(sizeof(double) * 4 * (size_t)(total + 1) * 2);
    int *off = (int *)malloc(sizeof(int) * (size_t)(nnets + 1) * 2);
    double *seg2 = seg + 4 * (total + 1); int *off2 = off + nnets + 1;
    int rc = 0; double wirelength, nint;
    if (c->reward_type == ORC_REWARD_BEAM || c->reward_type == ORC_REWARD_CENTROID) { /* :866-903 */
        if (orc_route(pts, pt_off, nnets, c->reward_type, c->reward_beam_width, seg, off) < 0) rc = -1;
        else {
            int k = orc_find_num_intersection(seg, off, nnets);
            wirelength = orc_find_wirelength(seg, off, nnets) / wl_norm;
            nint = (double)k / int_norm;
        }
    } else { /* both :905-928; tie -> beam (index 0) */
        if (orc_route(pts, pt_off, nnets, ORC_REWARD_BEAM, c->reward_beam_width, seg, off) < 0 ||
            orc_route(pts, pt_off, nnets, ORC_REWARD_CENTROID, c->reward_beam_width, seg2, off2) < 0) rc = -1;
        else {
            int kb = orc_find_num_intersection(seg, off, nnets), kc = orc_find_num_intersection(seg2, off2, nnets);
            int use_centroid = kc < kb;
            wirelength = (use_centroid ? orc_find_wirelength(seg2, off2, nnets) : orc_find_wirelength(seg, off, nnets)) / wl_norm;
            nint = (double)(use_centroid ? kc : kb) / int_norm;
        }
    }
    if (rc == 0) {
        out[0] = -1 * (c->weight_wirelength * wirelength + c->weight_num_intersections * nint);
        out[1] = wirelength; out[2] = nint;
    }
    free(seg); free(off);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* environment                                                               */
/* ------------------------------------------------------------------------- */
static double *zalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

orc_env *orc_create(const orc_config *cfg) {
    orc_env *e = (orc_env *)calloc(1, sizeof(orc_env));
    e->c = *cfg;
    if (cfg->kind == ORC_KIND_PIN || cfg->kind == ORC_KIND_SPATIAL) { /* P:467-468 / S:450-451 clip */
        e->c.net_distribution = cfg->net_distribution < 0 ? 0 : cfg->net_distribution > 9 ? 9 : cfg->net_distribution;
        e->c.pin_spread = cfg->pin_spread < 0 ? 0 : cfg->pin_spread > 9 ? 9 : cfg->pin_spread;
    }
    e->H = cfg->height; e->W = cfg->width;
    e->O = cfg->kind == ORC_KIND_SQUARE ? 1 : cfg->kind == ORC_KIND_RECT ? 2 : 4;
    e->C = cfg->kind == ORC_KIND_SQUARE ? 0 : cfg->max_num_components;
    e->mp = cfg->max_component_h * cfg->max_component_w; /* S:464 */
    e->N = cfg->max_num_nets;
    e->F = cfg->kind == ORC_KIND_SPATIAL ? 5 + e->mp : 5;
    int HW = e->H * e->W;
    e->grid = zalloc((size_t)HW);
    e->action_mask = zalloc((size_t)e->O * HW);
    e->comp_feat = zalloc((size_t)e->C * e->F);
    e->placement_mask = zalloc((size_t)e->C);
    e->component_mask = zalloc((size_t)e->C);
    e->comps = (orc_comp *)calloc((size_t)e->C + 1, sizeof(orc_comp));
    if (cfg->kind == ORC_KIND_PIN) { e->pin_rows = e->C * e->mp; e->pin_cat_w = 1; }
    if (cfg->kind == ORC_KIND_SPATIAL) { e->pin_rows = e->C * e->mp + 1; e->pin_cat_w = 2; }
    e->pins_num = zalloc((size_t)e->pin_rows * 4);
    e->pins_cat = zalloc((size_t)e->pin_rows * (e->pin_cat_w ? e->pin_cat_w : 1));
    e->pins = (orc_pin *)calloc((size_t)e->C * e->mp + 1, sizeof(orc_pin));
    e->net_start = (int *)calloc((size_t)e->N + 2, sizeof(int));
    if (cfg->kind == ORC_KIND_SPATIAL) {
        e->pin_grid = zalloc((size_t)HW * (e->N + 1));
        e->component_grid = zalloc((size_t)e->C * cfg->max_component_h * cfg->max_component_w * (e->N + 1));
    }
    e->cur = -1;
    e->max_wirelength = orc_upper_bound_wirelength(&e->c);
    e->max_num_intersections = orc_upper_bound_intersections(&e->c);
    e->reward_wirelength = -1.0; e->reward_intersection = -1.0; /* P:497-501 */
    return e;
}

void orc_destroy(orc_env *e) {
    if (!e) return;
    free(e->grid); free(e->action_mask); free(e->comp_feat); free(e->placement_mask); free(e->component_mask);
    free(e->comps); free(e->pins_num); free(e->pins_cat); free(e->pins); free(e->net_start);
    free(e->pin_grid); free(e->component_grid); free(e);
}

/* R:526-567 / P,S compute_action_mask_orientation: ones; zero the last ph-1
 * rows / pw-1 columns; the top-left (H-ph+1) x (W-pw+1) block := (valid-mode
 * window sum of the grid == 0). */
static int mask_orientation(const orc_env *e, int ph, int pw, double *m) {
    int H = e->H, W = e->W;
    if (ph > H || pw > W) return -1; /* scipy.signal.convolve2d(mode="valid") raises in the reference */
    for (int i = 0; i < H * W; i++) m[i] = 1.0;
    for (int i = H - 1; i > H - ph; i--) if (i >= 0) for (int j = 0; j < W; j++) m[i * W + j] = 0.0;
    for (int j = W - 1; j > W - pw; j--) if (j >= 0) for (int i = 0; i < H; i++) m[i * W + j] = 0.0;
    for (int i = 0; i + ph <= H; i++)
        for (int j = 0; j + pw <= W; j++) {
            double s = 0.0;
            for (int a = 0; a < ph; a++) for (int b = 0; b < pw; b++) s += e->grid[(i + a) * W + j + b];
            m[i * W + j] = (s == 0.0) ? 1.0 : 0.0;
        }
    return 0;
}
/* R:569-584 / S:1837-1854 compute_action_mask */
static int compute_action_mask(orc_env *e, const orc_comp *c) {
    int HW = e->H * e->W;
    if (mask_orientation(e, c->h, c->w, e->action_mask) < 0) return -1;
    if (mask_orientation(e, c->w, c->h, e->action_mask + HW) < 0) return -1;
    if (e->O == 4) {
        memcpy(e->action_mask + 2 * HW, e->action_mask, sizeof(double) * (size_t)HW);
        memcpy(e->action_mask + 3 * HW, e->action_mask + HW, sizeof(double) * (size_t)HW);
    }
    return 0;
}

/* R:60-79, P:217-239, S:203-239 Component.calculate_feature */
static void write_comp_feature(orc_env *e, const orc_comp *c) {
    double *f = e->comp_feat + (size_t)c->id * e->F;
    f[0] = c->h; f[1] = c->w; f[2] = c->pos_x; f[3] = c->pos_y;
    f[4] = (double)(c->h * c->w) / (double)(e->H * e->W);
    if (e->c.kind == ORC_KIND_SPATIAL) {
        int k = 0;
        for (int i = 0; i < e->mp; i++) f[5 + i] = -1.0;
        for (int p = 0; p < e->npins; p++) if (e->pins[p].comp_id == c->id) f[5 + k++] = e->pins[p].pin_id;
    }
}

/* P:1521-1542 / S:1469-1485 update_all_pins_feature: components in id order,
 * component.pins in self.pins order; later writers overwrite (quirk Q1). */
static void update_all_pins_feature(orc_env *e) {
    for (int c = 0; c < e->ncomp; c++)
        for (int p = 0; p < e->npins; p++) {
            orc_pin *pin = &e->pins[p];
            if (pin->comp_id != c) continue;
            int row;
            if (e->c.kind == ORC_KIND_SPATIAL) {
                const orc_comp *cm = &e->comps[c]; /* S:85-86 Pin.calculate_feature refreshes abs if placed */
                if (!(cm->pos_x == -1 || cm->pos_y == -1)) { pin->abs_x = cm->pos_x + pin->rel_x; pin->abs_y = cm->pos_y + pin->rel_y; }
                row = pin->pin_id;
                e->pins_cat[row * 2] = pin->net_id; e->pins_cat[row * 2 + 1] = pin->comp_id;
            } else {
                row = pin->comp_id * e->mp + pin->pin_id;
                e->pins_cat[row] = pin->net_id;
            }
            double *f = e->pins_num + (size_t)row * 4;
            f[0] = pin->rel_x; f[1] = pin->rel_y; f[2] = pin->abs_x; f[3] = pin->abs_y;
        }
}

/* S:1663-1675 draw_pins */
static void draw_pins(orc_env *e) {
    int HW = e->H * e->W, K = e->N + 1;
    int *cls = (int *)malloc(sizeof(int) * (size_t)HW);
    for (int i = 0; i < HW; i++) cls[i] = (int)e->grid[i];
    for (int n = 0; n < e->nnets; n++)
        for (int p = e->net_start[n]; p < e->net_start[n + 1]; p++) {
            if (e->pins[p].abs_x == -1 || e->pins[p].abs_y == -1) continue;
            cls[e->pins[p].abs_x * e->W + e->pins[p].abs_y] = n + 2;
        }
    for (int i = 0; i < HW; i++)
        for (int k = 0; k < K; k++) e->pin_grid[(size_t)i * K + k] = (cls[i] == k + 1) ? 1.0 : 0.0;
    free(cls);
}

/* S:1677-1697 draw_components */
static void draw_components(orc_env *e) {
    int mh = e->c.max_component_h, mw = e->c.max_component_w, K = e->N + 1;
    memset(e->component_grid, 0, sizeof(double) * (size_t)e->C * mh * mw * K);
    for (int p = 0; p < e->npins; p++) {
        const orc_pin *pin = &e->pins[p];
        if (pin->rel_x == -1 || pin->rel_y == -1) continue;
        e->component_grid[(((size_t)pin->comp_id * mh + pin->rel_x) * mw + pin->rel_y) * K + pin->net_id + 1] = 1.0;
    }
    for (int c = 0; c < e->ncomp; c++)
        for (int i = 0; i < mh * mw; i++) e->component_grid[((size_t)c * mh * mw + i) * K] = 1.0;
}

/* Q:74-113 reset (square) */
static int reset_square(orc_env *e) {
    int H = e->H, W = e->W, n = e->c.component_n;
    for (int i = 0; i < H * W; i++) { e->grid[i] = 0.0; e->action_mask[i] = 1.0; }
    if (n > 1) {
        for (int i = H - 1; i > H - n; i--) if (i >= 0) for (int j = 0; j < W; j++) e->action_mask[i * W + j] = 0.0;
        for (int j = W - 1; j > W - n; j--) if (j >= 0) for (int i = 0; i < H; i++) e->action_mask[i * W + j] = 0.0;
    }
    return 0;
}

/* reset from an instance (the tables reset() would have drawn).
 * R:310-351, P:1544-1597, S:1487-1549.  Pins in self.pins order (net-major). */
int orc_reset(orc_env *e, int ncomp, const int *comp_h, const int *comp_w, int nnets, int npins,
              const int *rel_x, const int *rel_y, const int *net, const int *comp, const int *pin_id) {
    int kind = e->c.kind;
    if (kind == ORC_KIND_SQUARE) return reset_square(e);
    if (ncomp < 1 || ncomp > e->C || npins > e->C * e->mp || nnets > e->N) return -2;
    int HW = e->H * e->W;
    memset(e->grid, 0, sizeof(double) * (size_t)HW);
    e->ncomp = ncomp; e->nnets = nnets; e->npins = npins;
    for (int i = 0; i < ncomp; i++) {
        orc_comp *c = &e->comps[i];
        c->h = comp_h[i]; c->w = comp_w[i]; c->id = i; c->placed = 0; c->pos_x = -1; c->pos_y = -1;
    }
    int prev = 0;
    for (int n = 0; n <= nnets; n++) e->net_start[n] = 0;
    for (int p = 0; p < npins; p++) {
        orc_pin *q = &e->pins[p];
        q->rel_x = rel_x[p]; q->rel_y = rel_y[p]; q->abs_x = -1; q->abs_y = -1;
        q->pin_id = pin_id[p]; q->comp_id = comp[p]; q->net_id = net[p];
        if (net[p] < prev || net[p] >= nnets) return -3;
        prev = net[p];
        e->net_start[net[p] + 1] = p + 1;
    }
    for (int n = 1; n <= nnets; n++) if (e->net_start[n] < e->net_start[n - 1]) e->net_start[n] = e->net_start[n - 1];
    if (kind == ORC_KIND_SPATIAL) {
        memset(e->pin_grid, 0, sizeof(double) * (size_t)HW * (e->N + 1));
        draw_components(e);
    }
    memset(e->comp_feat, 0, sizeof(double) * (size_t)e->C * e->F);
    if (kind != ORC_KIND_RECT) {
        memset(e->pins_num, 0, sizeof(double) * (size_t)e->pin_rows * 4);
        memset(e->pins_cat, 0, sizeof(double) * (size_t)e->pin_rows * e->pin_cat_w);
        if (kind == ORC_KIND_SPATIAL) { e->pins_cat[(e->pin_rows - 1) * 2] = -1; e->pins_cat[(e->pin_rows - 1) * 2 + 1] = -1; }
    }
    e->cur = 0;
    if (compute_action_mask(e, &e->comps[0]) < 0) return -4;
    if (kind == ORC_KIND_RECT) {
        for (int i = 0; i < e->C; i++) { e->placement_mask[i] = 0.0; e->component_mask[i] = i < ncomp ? 1.0 : 0.0; }
    } else {
        for (int i = 0; i < e->C; i++) e->placement_mask[i] = i < ncomp ? 1.0 : 0.0;
        e->placement_mask[0] = 3.0;
    }
    for (int i = 0; i < ncomp; i++) write_comp_feature(e, &e->comps[i]);
    if (kind != ORC_KIND_RECT) update_all_pins_feature(e);
    return 0;
}

/* S:793-929 find_reward on the env's own nets */
static int find_reward(orc_env *e, double *reward) {
    int placed_all = (e->cur == -1);
    int *pts = (int *)malloc(sizeof(int) * 2 * (size_t)(e->npins + 1));
    for (int p = 0; p < e->npins; p++) { pts[2 * p] = e->pins[p].abs_x; pts[2 * p + 1] = e->pins[p].abs_y; }
    double out[3];
    int rc = orc_find_reward_pts(&e->c, placed_all, pts, e->net_start, e->nnets, out);
    free(pts);
    if (rc < 0) return rc;
    *reward = out[0]; e->reward_wirelength = out[1]; e->reward_intersection = out[2];
    return 0;
}

static int all_mask_zero(const orc_env *e) {
    for (int i = 0; i < e->O * e->H * e->W; i++) if (e->action_mask[i] != 0.0) return 0;
    return 1;
}

/* Q:115-153 step (square), with the incremental mask edit of Q:187-244 */
static int step_square(orc_env *e, int x, int y, double *reward, int *done) {
    int H = e->H, W = e->W, n = e->c.component_n;
    int valid = x >= 0 && x < H && y >= 0 && y < W && e->action_mask[x * W + y] == 1.0;
    if (!valid) { *reward = 0.0; *done = 1; return 0; }
#define CLEAR(ARR, r0, r1, c0, c1, V) \
    for (int i_ = (r0); i_ < (r1) && i_ < H; i_++) for (int j_ = (c0); j_ < (c1) && j_ < W; j_++) ARR[i_ * W + j_] = V
    CLEAR(e->grid, x, x + n, y, y + n, 1.0);
    CLEAR(e->action_mask, x, x + n, y, y + n, 0.0);
    if (n > 1) {
        int c0 = y - n + 1 > 0 ? y - n + 1 : 0, r0 = x - n + 1 > 0 ? x - n + 1 : 0;
        if (y != 0) CLEAR(e->action_mask, x, x + n, c0, y, 0.0);          /* horizontal (left) */
        if (x != 0) CLEAR(e->action_mask, r0, x, y, y + n, 0.0);          /* vertical (up) */
        if (x != 0 && y != 0) CLEAR(e->action_mask, r0, x, c0, y, 0.0);   /* diagonal */
    }
#undef CLEAR
    *done = all_mask_zero(e);
    *reward = 1.0;
    return 0;
}

/* R:353-432, P:1599-1710, S:1551-1661 step.
 * info[0] = wirelength, info[1] = num_intersections, *has_info = 1 iff the reference returns them. */
int orc_step(orc_env *e, int o, int x, int y, double *reward, int *done, double *info, int *has_info) {
    int kind = e->c.kind, H = e->H, W = e->W, HW = H * W;
    *has_info = 0;
    if (kind == ORC_KIND_SQUARE) return step_square(e, x, y, reward, done);
    /* validate_action: action_mask[o, x, y] == 1, IndexError -> False; negative indices are treated as
     * out of range (gym.Discrete never produces them; NumPy would wrap them) */
    int valid = o >= 0 && o < e->O && x >= 0 && x < H && y >= 0 && y < W &&
                e->action_mask[(size_t)o * HW + x * W + y] == 1.0;
    if (!valid) {
        if (kind == ORC_KIND_RECT) { *reward = 0.0; *done = 1; return 0; } /* R:424-432 */
        if (kind == ORC_KIND_SPATIAL) draw_pins(e);                         /* S:1644 */
        if (find_reward(e, reward) < 0) return -1;                          /* S:1656 */
        info[0] = e->reward_wirelength; info[1] = e->reward_intersection; *has_info = 1;
        *done = 1;
        return 0;
    }
    orc_comp *c = &e->comps[e->cur];
    int ph = (o == 0 || o == 2) ? c->h : c->w, pw = (o == 0 || o == 2) ? c->w : c->h; /* S:1742-1747 */
    for (int i = x; i < x + ph && i < H; i++) for (int j = y; j < y + pw && j < W; j++) e->grid[i * W + j] = 1.0;
    c->placed = 1; c->pos_x = x; c->pos_y = y;
    if (kind != ORC_KIND_RECT) { /* S:149-190 place_component: rotate relative coords in place */
        for (int p = 0; p < e->npins; p++) {
            orc_pin *q = &e->pins[p];
            if (q->comp_id != c->id) continue;
            int rx = q->rel_x, ry = q->rel_y;
            if (o == 1) { q->rel_x = ry; q->rel_y = c->h - rx - 1; }
            else if (o == 2) { q->rel_x = c->h - rx - 1; q->rel_y = c->w - ry - 1; }
            else if (o == 3) { q->rel_x = c->w - ry - 1; q->rel_y = rx; }
            q->abs_x = x + q->rel_x; q->abs_y = y + q->rel_y;
        }
    }
    write_comp_feature(e, c);
    if (kind != ORC_KIND_RECT) update_all_pins_feature(e);
    if (kind == ORC_KIND_SPATIAL) draw_pins(e);
    if (kind == ORC_KIND_RECT) e->placement_mask[c->id] = 1.0; else e->placement_mask[c->id] = 2.0;
    if (e->cur + 1 < e->ncomp) {
        e->cur += 1;
        if (kind != ORC_KIND_RECT) e->placement_mask[e->cur] = 3.0;
        if (compute_action_mask(e, &e->comps[e->cur]) < 0) return -4;
    } else {
        e->cur = -1;
        memset(e->action_mask, 0, sizeof(double) * (size_t)e->O * HW);
    }
    *done = e->cur == -1 ? 1 : all_mask_zero(e); /* S:1856-1869 */
    if (kind == ORC_KIND_RECT) { *reward = 1.0; return 0; }
    if (!*done) { *reward = 0.0; return 0; }
    if (find_reward(e, reward) < 0) return -1;
    info[0] = e->reward_wirelength; info[1] = e->reward_intersection; *has_info = 1;
    return 0;
}

/* observation access for the ctypes wrapper: key -> pointer + element count */
enum { ORC_OBS_GRID, ORC_OBS_ACTION_MASK, ORC_OBS_COMP_FEAT, ORC_OBS_PLACEMENT_MASK, ORC_OBS_COMPONENT_MASK,
       ORC_OBS_PINS_NUM, ORC_OBS_PINS_CAT, ORC_OBS_PIN_GRID, ORC_OBS_COMPONENT_GRID };
const double *orc_obs(const orc_env *e, int key, int64_t *count) {
    int HW = e->H * e->W;
    switch (key) {
    case ORC_OBS_GRID: *count = HW; return e->grid;
    case ORC_OBS_ACTION_MASK: *count = (int64_t)e->O * HW; return e->action_mask;
    case ORC_OBS_COMP_FEAT: *count = (int64_t)e->C * e->F; return e->comp_feat;
    case ORC_OBS_PLACEMENT_MASK: *count = e->C; return e->placement_mask;
    case ORC_OBS_COMPONENT_MASK: *count = e->C; return e->component_mask;
    case ORC_OBS_PINS_NUM: *count = (int64_t)e->pin_rows * 4; return e->pins_num;
    case ORC_OBS_PINS_CAT: *count = (int64_t)e->pin_rows * e->pin_cat_w; return e->pins_cat;
    case ORC_OBS_PIN_GRID: *count = e->pin_grid ? (int64_t)HW * (e->N + 1) : 0; return e->pin_grid;
    case ORC_OBS_COMPONENT_GRID:
        *count = e->component_grid ? (int64_t)e->C * e->c.max_component_h * e->c.max_component_w * (e->N + 1) : 0;
        return e->component_grid;
    }
    *count = 0;
    return 0;
}
int orc_current_component(const orc_env *e) { return e->cur; }
double orc_max_wirelength(const orc_env *e) { return e->max_wirelength; }
double orc_max_num_intersections(const orc_env *e) { return e->max_num_intersections; }
