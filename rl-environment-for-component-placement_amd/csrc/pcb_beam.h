// pcb_beam.h -- beam-search routes with the model of CPython's set iteration order
// Part of libpcbenv.so's single translation unit (included by pcbenv_kernels.hip); CDNA4 / gfx950 only.
#pragma once
#include "pcb_reward.h"

// ---- beam-search routing (S:1273-1286 pin_outlier, S:1303-1369 beam_search, S:1371-1406) -----------------
// beam_search keeps, per popped path, the beam_width nearest unvisited points of
// `sorted(points_to_visit - visited, key=distance)`.  Python's sort is stable, so neighbours at equal distance
// keep the iteration order of that temporary CPython set -- a pure function of the tuple hashes and of
// Objects/setobject.c's open-addressing table (SURVEY.md trap T2).  That order can only change WHICH points are
// kept when the beam_width-th and the next distance tie (the order among kept neighbours is irrelevant: heapq
// pops by (priority, path), not by insertion).  So the set model below runs only on such boundary ties.
// One lane per net; all per-net scratch lives in LDS (no private-memory arrays -> no scratch segment).
#define CS_EMPTY 0xFF
#define CS_DUMMY 0xFE
#define BS_MAXPTS (PCBENV_MAX_PINS_PER_NET - 1)
struct CSet { int mask, fill, used; unsigned char t[32]; int pad; };  // 48 bytes
// one partial path of the beam: four 64-bit words so that queue traffic is wide LDS accesses and the popped
// entry lives in registers.  meta = visited (bits 0-15) | length (bits 16-23); p0/p1 = the path, one byte per
// point index (0xFF = the start point).
struct BsEntry {
    double prio; u64 meta, p0, p1;
    __device__ unsigned visited() const { return (unsigned)(meta & 0xFFFFull); }
    __device__ int len() const { return (int)((meta >> 16) & 0xFFull); }
    __device__ int at(int j) const { return (int)(((j < 8 ? p0 : p1) >> ((j & 7) * 8)) & 0xFFull); }
    __device__ void push(int idx) {
        const int l = len();
        const u64 b = (u64)(unsigned)idx << ((l & 7) * 8);
        if (l < 8) p0 |= b; else p1 |= b;
        meta = (meta & ~(0xFFull << 16)) | ((u64)(l + 1) << 16) | (1ull << idx);
    }
};
static_assert(sizeof(BsEntry) == 32 && sizeof(CSet) == 48, "beam LDS records");
#define BEAM_LDS_PER_NET(k) (64 * (k) * (k) + 16 * 8 + 16 + 2 * 48)
#define BEAM_LDS_BYTES(nets, k) ((nets) * BEAM_LDS_PER_NET(k))

// points to visit of one net: the net's pins without the start pin `st`
struct NetPts {  // coordinates packed one byte each into registers (<= 15 points): no LDS round trip per access
    u64 xs0, xs1, ys0, ys1;
    __device__ int x(int i) const { return (int)(((i < 8 ? xs0 : xs1) >> ((i & 7) * 8)) & 0xFFull); }
    __device__ int y(int i) const { return (int)(((i < 8 ? ys0 : ys1) >> ((i & 7) * 8)) & 0xFFull); }
    __device__ static NetPts load(const PinRec *p, int cnt, int st) {
        NetPts n{0ull, 0ull, 0ull, 0ull};
        int m = 0;
        for (int i = 0; i < cnt; i++) {
            if (i == st) continue;
            const u64 x = (u64)(unsigned char)p[i].abs_x << ((m & 7) * 8), y = (u64)(unsigned char)p[i].abs_y << ((m & 7) * 8);
            if (m < 8) { n.xs0 |= x; n.ys0 |= y; } else { n.xs1 |= x; n.ys1 |= y; }
            m++;
        }
        return n;
    }
};

__device__ inline u64 tuple_hash2(int x, int y) {  // Objects/tupleobject.c (xxHash-style), hash(int) == int
    const u64 P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P5 = 2870177450012600261ull;
    u64 acc = P5;
    acc += (u64)(long long)x * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
    acc += (u64)(long long)y * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
    acc += 2ull ^ (P5 ^ 3527539ull);
    return acc == ~0ull ? 1546275796ull : acc;
}
__device__ inline void cs_init(CSet *s, int size) {
    s->mask = size - 1; s->fill = 0; s->used = 0;
    for (int i = 0; i < 32; i++) s->t[i] = CS_EMPTY;
}
// first unused slot on the probe sequence of `hash` (set_insert_clean / the miss path of set_add_entry)
__device__ inline int cs_probe_unused(const CSet *s, u64 hash, int *freeslot) {
    const unsigned mask = (unsigned)s->mask;
    u64 perturb = hash;
    unsigned i = (unsigned)hash & mask;
    for (;;) {
        const unsigned probes = (i + 9u <= mask) ? 9u : 0u;
        for (unsigned k = 0; k <= probes; k++) {
            const unsigned char c = s->t[i + k];
            if (c == CS_EMPTY) return (int)(i + k);
            if (c == CS_DUMMY && freeslot) *freeslot = (int)(i + k);
        }
        perturb >>= 5;
        i = (unsigned)(((u64)i * 5u + 1u + perturb) & mask);
    }
}
// set_table_resize: re-insert the active keys in old slot order (the old table is copied to `tmp` first)
__device__ inline void cs_resize(CSet *s, CSet *tmp, int minused, const NetPts &pt) {
    int newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    *tmp = *s;
    cs_init(s, newsize);
    for (int i = 0; i <= tmp->mask; i++)
        if (tmp->t[i] < CS_DUMMY) s->t[cs_probe_unused(s, tuple_hash2(pt.x(tmp->t[i]), pt.y(tmp->t[i])), 0)] = tmp->t[i];
    s->fill = s->used = tmp->used;
}
__device__ inline void cs_add(CSet *s, CSet *tmp, int key, const NetPts &pt) {
    int freeslot = -1;
    const int slot = cs_probe_unused(s, tuple_hash2(pt.x(key), pt.y(key)), &freeslot);
    if (freeslot >= 0) { s->t[freeslot] = (unsigned char)key; s->used++; return; }
    s->t[slot] = (unsigned char)key; s->fill++; s->used++;
    if (s->fill * 5 >= s->mask * 3) cs_resize(s, tmp, s->used * 4, pt);
}
__device__ inline void cs_discard(CSet *s, int key, const NetPts &pt) {
    const unsigned mask = (unsigned)s->mask;
    const u64 hash = tuple_hash2(pt.x(key), pt.y(key));
    u64 perturb = hash;
    unsigned i = (unsigned)hash & mask;
    for (;;) {
        const unsigned probes = (i + 9u <= mask) ? 9u : 0u;
        for (unsigned k = 0; k <= probes; k++) {
            const unsigned char c = s->t[i + k];
            if (c == CS_EMPTY) return;
            if (c == (unsigned char)key) { s->t[i + k] = CS_DUMMY; s->used--; return; }
        }
        perturb >>= 5;
        i = (unsigned)(((u64)i * 5u + 1u + perturb) & mask);
    }
}
// Iteration order of `set(points) - visited` (set_difference: copy-and-discard when len(A) >> 2 > len(visited),
// else a fresh set filled in A's slot order).  A and R are LDS tables; `order` receives point indices.
__device__ inline int cs_difference_order(CSet *A, CSet *R, int m, unsigned visited, const NetPts &pt, unsigned char *order) {
    // points_to_visit = set(points): inserted in list order.  R doubles as the resize temporary while A is built.
    cs_init(A, 8);
    for (int i = 0; i < m; i++) cs_add(A, R, i, pt);
    if ((m >> 2) > (int)__popc(visited)) {
        cs_init(R, 8);
        if (m * 5 >= R->mask * 3) { int ns = 8; while (ns <= 2 * m) ns <<= 1; cs_init(R, ns); }
        if (R->mask == A->mask) { *R = *A; }  // set_merge: same size, no dummies -> the table is copied as is
        else {
            for (int i = 0; i <= A->mask; i++)
                if (A->t[i] < CS_DUMMY) R->t[cs_probe_unused(R, tuple_hash2(pt.x(A->t[i]), pt.y(A->t[i])), 0)] = A->t[i];
            R->fill = R->used = A->used;
        }
        for (int k = 0; k < m; k++) if (visited >> k & 1u) cs_discard(R, k, pt);
        // "if more than 1/4th are dummies, resize them away" cannot trigger for m <= 15 (<= 2 dummies, mask >= 15)
    } else {
        // fresh result set filled in A's slot order: collect the survivors first, after which A is free to
        // serve as the temporary of R's set_table_resize (5th insert: 8 -> 32 slots)
        int ns = 0;
        for (int i = 0; i <= A->mask; i++)
            if (A->t[i] < CS_DUMMY && !(visited >> A->t[i] & 1u)) order[ns++] = A->t[i];
        cs_init(R, 8);
        for (int i = 0; i < ns; i++) cs_add(R, A, order[i], pt);
    }
    int n = 0;
    for (int i = 0; i <= R->mask; i++) if (R->t[i] < CS_DUMMY) order[n++] = R->t[i];
    return n;
}

// One net, one lane: fills the net's slots [s, s+cnt) of the segment view with the beam route.
// `scratch` = this net's BEAM_LDS_PER_NET(k) bytes of LDS.
__device__ inline void beam_route_net(const SegView &v, const PinRec *pins, int s, int cnt, int k, unsigned char *scratch) {
    BsEntry *queue = (BsEntry *)scratch, *next = queue + k * k;
    double *dist = (double *)(scratch + 64 * k * k);
    unsigned char *order = (unsigned char *)(dist + 16);
    CSet *A = (CSet *)(order + 16), *R = A + 1;
    const double cx = v.cen[pins[s].net], cy = v.cen[PCBENV_MAX_NETS + pins[s].net];
    int st = 0; double bd = 0.0;  // pin_outlier: first arg-max of the distance to the centroid
    for (int i = 0; i < cnt; i++) {
        const double d = norm2((double)pins[s + i].abs_x - cx, (double)pins[s + i].abs_y - cy);
        if (i == 0 || d > bd) { bd = d; st = i; }
    }
    const int sx = pins[s + st].abs_x, sy = pins[s + st].abs_y;
    const int m = cnt - 1;
    const NetPts pt = NetPts::load(pins + s, cnt, st);
    const unsigned all = (1u << m) - 1u;
    int qn = 1;
    { BsEntry e0; e0.prio = 0.0; e0.meta = 1ull << 16; e0.p0 = 0xFFull; e0.p1 = 0ull; queue[0] = e0; }
    bool found = false;
    BsEntry res;
    while (!found) {
        int nn = 0;
        unsigned taken = 0;
        const int pops = k < qn ? k : qn;
        for (int t = 0; t < pops && !found; t++) {
            int best = -1;  // heappop: minimum (priority, path) of what is left
            BsEntry e;
            for (int i = 0; i < qn; i++) {
                if (taken >> i & 1u) continue;
                const BsEntry a = queue[i];
                bool less;
                if (best < 0) less = true;
                else if (a.prio != e.prio) less = a.prio < e.prio;
                else {  // equal priorities: python compares the path lists of (x, y) tuples
                    less = a.len() < e.len();
                    const int n = a.len() < e.len() ? a.len() : e.len();
                    for (int j = 0; j < n; j++) {
                        const int pa = a.at(j), pb = e.at(j);
                        const int ax = pa == 0xFF ? sx : pt.x(pa), ay = pa == 0xFF ? sy : pt.y(pa);
                        const int bx = pb == 0xFF ? sx : pt.x(pb), by = pb == 0xFF ? sy : pt.y(pb);
                        if (ax != bx) { less = ax < bx; break; }
                        if (ay != by) { less = ay < by; break; }
                    }
                }
                if (less) { best = i; e = a; }
            }
            taken |= 1u << best;
            if (e.visited() == all) { found = true; res = e; break; }
            const int cur = e.at(e.len() - 1);
            const int ux = cur == 0xFF ? sx : pt.x(cur), uy = cur == 0xFF ? sy : pt.y(cur);
            // the k+1 nearest unvisited points in registers (ascending distance, index order among equals)
            double td[PCBENV_MAX_BEAM_WIDTH + 1]; int ti[PCBENV_MAX_BEAM_WIDTH + 1];
            #pragma unroll
            for (int q = 0; q <= PCBENV_MAX_BEAM_WIDTH; q++) { td[q] = 0.0; ti[q] = 0; }
            int nfill = 0, cntn = 0;
            for (int i = 0; i < m; i++) {
                if (e.visited() >> i & 1u) continue;
                cntn++;
                double cd = norm2((double)(ux - pt.x(i)), (double)(uy - pt.y(i)));
                int ci = i;
                bool shifting = false, placed = false;
                #pragma unroll
                for (int q = 0; q <= PCBENV_MAX_BEAM_WIDTH; q++) {
                    if (q > k || placed) continue;
                    if (q == nfill) { td[q] = cd; ti[q] = ci; placed = true; }
                    else if (shifting || td[q] > cd) {
                        const double xd = td[q]; const int xi = ti[q];
                        td[q] = cd; ti[q] = ci; cd = xd; ci = xi; shifting = true;
                    }
                }
                if (nfill <= k) nfill++;
            }
            const int take = cntn < k ? cntn : k;
            bool tie = false;
            #pragma unroll
            for (int q = 1; q <= PCBENV_MAX_BEAM_WIDTH; q++) if (q == k && cntn > k) tie = td[q - 1] == td[q];
            if (tie) {  // boundary tie: the CPython set order decides who is kept
                const int nset = cs_difference_order(A, R, m, e.visited(), pt, order);
                for (int i = 0; i < nset; i++) dist[i] = norm2((double)(ux - pt.x(order[i])), (double)(uy - pt.y(order[i])));
                for (int i = 1; i < nset; i++) {  // sorted(key=distance): stable
                    const unsigned char o = order[i]; const double d = dist[i];
                    int j = i - 1;
                    while (j >= 0 && dist[j] > d) { order[j + 1] = order[j]; dist[j + 1] = dist[j]; j--; }
                    order[j + 1] = o; dist[j + 1] = d;
                }
                for (int i = 0; i < take; i++) { BsEntry q = e; q.push(order[i]); q.prio = e.prio + dist[i]; next[nn++] = q; }
            } else {
                #pragma unroll
                for (int q = 0; q < PCBENV_MAX_BEAM_WIDTH; q++)
                    if (q < take) { BsEntry w = e; w.push(ti[q]); w.prio = e.prio + td[q]; next[nn++] = w; }
            }
        }
        if (!found) { BsEntry *tmp = queue; queue = next; next = tmp; qn = nn; if (qn == 0) break; }
    }
    for (int i = 0; i < cnt; i++) v.act[s + i] = 0;
    if (!found) return;
    for (int i = 0; i + 1 < res.len(); i++) {
        const int a = res.at(i), b = res.at(i + 1);
        const double x1 = a == 0xFF ? sx : pt.x(a), y1 = a == 0xFF ? sy : pt.y(a);
        const double x2 = b == 0xFF ? sx : pt.x(b), y2 = b == 0xFF ? sy : pt.y(b);
        v.X1[s + i] = x1; v.Y1[s + i] = y1; v.X2[s + i] = x2; v.Y2[s + i] = y2;
        v.D[s + i] = norm2(x1 - x2, y1 - y2);
        v.act[s + i] = 1;
    }
}

// beam (and, for "both", centroid) routes of the terminal state -> wirelength, #intersections of the chosen route
__device__ inline void route_beam_or_both(const DevParams &p, const EnvHdr *hdr, const PinRec *pins, double *seg,
                                          int lane, double *wirelength, int *nintersections) {
    const SegView v = seg_view(seg, p.P);
    unsigned char *beam = v.beam;
    net_offsets_and_centroids(v, hdr, pins, lane);
    STAMP(5);
    for (int n = lane; n < hdr->nnets; n += NT)
        beam_route_net(v, pins, v.nstart[n], v.nstart[n + 1] - v.nstart[n], p.beam_width, beam + (size_t)n * BEAM_LDS_PER_NET(p.beam_width));
    lds_sync();
    STAMP(24);
    count_and_length(p, v, hdr, pins, lane, wirelength, nintersections);
    STAMP(25);
    if (p.reward_type == PCBENV_REWARD_BOTH) {  // S:609-627 lowest_num_intersections: ties keep the beam route
        double wc; int kc;
        build_centroid_segments(v, hdr, pins, lane);
        count_and_length(p, v, hdr, pins, lane, &wc, &kc);
        if (kc < *nintersections) { *nintersections = kc; *wirelength = wc; }
    }
    STAMP(8);
}

