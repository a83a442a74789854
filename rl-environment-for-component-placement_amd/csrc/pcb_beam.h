// pcb_beam.h -- beam-search routes with the model of CPython's set iteration order
// Part of libpcbenv.so's single translation unit (included by pcbenv_kernels.hip); CDNA4 / gfx950 only.

// ---- beam-search routing (S:1273-1286 pin_outlier, S:1303-1369 beam_search, S:1371-1406) -----------------
// beam_search keeps, per popped path, the beam_width nearest unvisited points of
// `sorted(points_to_visit - visited, key=distance)`.  Python's sort is stable, so neighbours at equal distance
// keep the iteration order of that temporary CPython set -- a pure function of the tuple hashes and of
// Objects/setobject.c's open-addressing table (SURVEY.md trap T2).  That order can only change WHICH points are
// kept when the beam_width-th and the next distance tie (the order among kept neighbours is irrelevant: heapq
// pops by (priority, path), not by insertion).  So the set model below runs only on such boundary ties.
// Four lanes per net (one per heappop of a level, see "laid out for latency" below); all per-net scratch lives in LDS
// (no private-memory arrays -> no scratch segment).
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
#define BEAM_LDS_PER_NET(k) (64 * (k) * (k) + 16 * 8 + 16 + 2 * 48 + 16 * 4)
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

static __device__ inline u64 tuple_hash2(int x, int y) {  // Objects/tupleobject.c (xxHash-style), hash(int) == int
    const u64 P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P5 = 2870177450012600261ull;
    u64 acc = P5;
    acc += (u64)(long long)x * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
    acc += (u64)(long long)y * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
    acc += 2ull ^ (P5 ^ 3527539ull);
    return acc == ~0ull ? 1546275796ull : acc;
}
static __device__ inline void cs_init(CSet *s, int size) {
    s->mask = size - 1; s->fill = 0; s->used = 0;
    for (int i = 0; i < 32; i++) s->t[i] = CS_EMPTY;
}
// first unused slot on the probe sequence of `hash` (set_insert_clean / the miss path of set_add_entry)
static __device__ inline int cs_probe_unused(const CSet *s, u64 hash, int *freeslot) {
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
static __device__ inline void cs_resize(CSet *s, CSet *tmp, int minused, const NetPts &pt) {
    int newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    *tmp = *s;
    cs_init(s, newsize);
    for (int i = 0; i <= tmp->mask; i++)
        if (tmp->t[i] < CS_DUMMY) s->t[cs_probe_unused(s, tuple_hash2(pt.x(tmp->t[i]), pt.y(tmp->t[i])), 0)] = tmp->t[i];
    s->fill = s->used = tmp->used;
}
static __device__ inline void cs_add(CSet *s, CSet *tmp, int key, const NetPts &pt) {
    int freeslot = -1;
    const int slot = cs_probe_unused(s, tuple_hash2(pt.x(key), pt.y(key)), &freeslot);
    if (freeslot >= 0) { s->t[freeslot] = (unsigned char)key; s->used++; return; }
    s->t[slot] = (unsigned char)key; s->fill++; s->used++;
    if (s->fill * 5 >= s->mask * 3) cs_resize(s, tmp, s->used * 4, pt);
}
static __device__ inline void cs_discard(CSet *s, int key, const NetPts &pt) {
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
static __device__ inline int cs_difference_order(CSet *A, CSet *R, int m, unsigned visited, const NetPts &pt, unsigned char *order) {
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

// ---- boundary ties, fast path ------------------------------------------------------------------------------
// `A = set(points)` and the tuple hashes depend on the net only: built once per net (first tie) and kept in LDS.
// (Low 32 bits of each hash: they carry the first five perturb steps of an 8-slot walk; a longer walk -- occupied slots
// can be revisited -- recomputes the full hash.)
static __device__ inline void cs_build_points(CSet *A, CSet *tmp, unsigned *hs, int m, const NetPts &pt) {
    cs_init(A, 8);
    for (int i = 0; i < m; i++) { hs[i] = (unsigned)tuple_hash2(pt.x(i), pt.y(i)); cs_add(A, tmp, i, pt); }
}
// Iteration order of `A - visited` when the result has at most 4 elements and comes from the "fresh set filled in
// A's slot order" branch of set_difference: the result table keeps its 8 slots (no resize before the 5th insert), so
// it lives in one 64-bit register, one byte per slot (mask 7: LINEAR_PROBES never applies, only the perturb walk).
// Returns the number of elements, their point indices in iteration order packed one per byte.
static __device__ inline int cs_small_difference_order(const CSet *A, const unsigned *hs, unsigned visited, const NetPts &pt, unsigned *packed) {
    const unsigned *tw = (const unsigned *)A->t;  // 4-byte aligned (offset 12 of a 16-byte aligned record)
    const int nw = (A->mask + 1) >> 2;            // 2 or 8 words
    unsigned w[8];
    #pragma unroll
    for (int i = 0; i < 8; i++) w[i] = i < nw ? tw[i < nw ? i : 0] : 0xFFFFFFFFu;
    u64 rt = ~0ull;
    int ns = 0;
    #pragma unroll
    for (int i = 0; i < 8; i++) {
        if (w[i] == 0xFFFFFFFFu) continue;  // four empty slots
        #pragma unroll
        for (int b = 0; b < 4; b++) {
            const unsigned c = (w[i] >> (8 * b)) & 0xFFu;
            if (c >= CS_DUMMY || (visited >> c & 1u)) continue;
            u64 perturb = hs[c];
            unsigned slot = (unsigned)perturb & 7u;
            for (int step = 1; ((rt >> (8 * slot)) & 0xFFull) != 0xFFull; step++) {
                if (step == 6) perturb = tuple_hash2(pt.x(c), pt.y(c)) >> 25;  // the cached low word has run out: the full hash, five steps in
                perturb >>= 5;
                slot = (unsigned)(((u64)slot * 5u + 1u + perturb) & 7u);
            }
            rt = (rt & ~(0xFFull << (8 * slot))) | ((u64)c << (8 * slot));
            ns++;
        }
    }
    unsigned out = 0; int n = 0;
    #pragma unroll
    for (int sl = 0; sl < 8; sl++) {
        const unsigned c = (unsigned)(rt >> (8 * sl)) & 0xFFu;
        if (c != 0xFFu) { out |= c << (8 * n); n++; }
    }
    *packed = out;
    return ns;
}

// ---- beam search laid out for latency -----------------------------------------------------------------------
// A terminal wavefront is alone with a short dependent chain (one level per pin of the net): what counts is the
// number of dependent instructions and LDS round trips per level, not lanes.  So:
//  * a net gets PCBENV_MAX_BEAM_WIDTH lanes, one per heappop of a level: all entries of a level have the same length,
//    hence the same number of unvisited points and the same number of children -- lane t selects the t-th smallest
//    queue entry by itself, expands it and writes its children to next[t * take ...): the only thing lanes of a net
//    share is the queue in LDS (wave-level ordering, a net's lanes never span two wavefronts);
//  * "the k nearest unvisited points, index order among equals" is taken on integer keys (dx*dx + dy*dy) << 4 | index
//    held in registers: coordinates are small integers, the squared distance is exact and np.linalg.norm is strictly
//    monotone on it, so the order (and the boundary tie) is the reference's; only the priorities need float64 norms;
//  * the farthest-from-centroid start pin and the route segments are computed one pin per lane before / after;
//  * the rare boundary tie still runs the serial CPython-set model, the lanes of a net taking turns (shared scratch).
// Same results as round 1's one-lane-per-net search (git history): same pop order (first index among fully equal
// entries), same children in the same queue order.
#define BEAM_LANES_PER_NET PCBENV_MAX_BEAM_WIDTH
#if defined(PCBENV_STAMPS) && defined(PCBENV_STAMPS_BEAM)  // phase cycles of the search, accumulated by lane 0 into stamp slots 26..29
#define BEAM_T0() unsigned long long bt0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bt0_) :: "memory")
#define BEAM_ACC(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (threadIdx.x == 0 && beam_dbg) beam_dbg[(size_t)blockIdx.x * 32 + (k)] += t_ - bt0_; bt0_ = t_; } while (0)
#else
#define BEAM_T0() do { } while (0)
#define BEAM_ACC(k) do { } while (0)
#endif
static __device__ inline void wave_lds_order() {  // LDS traffic of one wavefront executes in order: only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
// all pins of a net, original order (index = offset from the net's first slot), one byte per coordinate
template <int MAXC> struct NetAll {
    u64 xs0, xs1, ys0, ys1;
    __device__ int x(int i) const { return (int)(((MAXC <= 8 || i < 8 ? xs0 : xs1) >> ((i & 7) * 8)) & 0xFFull); }
    __device__ int y(int i) const { return (int)(((MAXC <= 8 || i < 8 ? ys0 : ys1) >> ((i & 7) * 8)) & 0xFFull); }
    __device__ static NetAll load(const PinRec *p, int cnt) {
        NetAll n{0ull, 0ull, 0ull, 0ull};
        #pragma unroll
        for (int i = 0; i < MAXC; i++) {  // all loads go out together; slots past the net repeat pin 0 and are masked
            const PinRec pr = p[i < cnt ? i : 0];
            const u64 x = i < cnt ? (u64)(unsigned char)pr.abs_x << ((i & 7) * 8) : 0ull, y = i < cnt ? (u64)(unsigned char)pr.abs_y << ((i & 7) * 8) : 0ull;
            if (i < 8) { n.xs0 |= x; n.ys0 |= y; } else { n.xs1 |= x; n.ys1 |= y; }
        }
        return n;
    }
};
// python's list comparison of two queued paths of equal priority (entries hold pin indices; all paths of a queue have one length)
template <int MAXC> static __device__ inline bool path_less(const BsEntry &a, const BsEntry &e, const NetAll<MAXC> &pt) {
    bool less = a.len() < e.len();
    const int n = a.len() < e.len() ? a.len() : e.len();
    for (int j = 0; j < n; j++) {
        const int pa = a.at(j), pb = e.at(j);
        const int ax = pt.x(pa), ay = pt.y(pa), bx = pt.x(pb), by = pt.y(pb);
        if (ax != bx) { less = ax < bx; break; }
        if (ay != by) { less = ay < by; break; }
    }
    return less;
}
// SMALL: nets of <= 8 pins and beam widths <= 2 (queue <= 4 entries): everything unrolled in registers
template <bool SMALL>
static __device__ __forceinline__ void beam_route_lanes(const SegView &v, const PinRec *pins, int s, int cnt, int st, int k, unsigned char *scratch, int t, unsigned long long *beam_dbg) {
    constexpr int MAXC = SMALL ? 8 : PCBENV_MAX_PINS_PER_NET, MAXQ = SMALL ? 4 : PCBENV_MAX_BEAM_WIDTH * PCBENV_MAX_BEAM_WIDTH;
    constexpr unsigned KINF = 0x7FFFFFFFu;
    BEAM_T0();
    BsEntry *queue = (BsEntry *)scratch, *next = queue + k * k;
    double *dist = (double *)(scratch + 64 * k * k);
    unsigned char *order = (unsigned char *)(dist + 16);
    CSet *A = (CSet *)(order + 16), *R = A + 1;
    unsigned *hs = (unsigned *)(R + 1);  // tuple hashes (low words) of the points to visit, valid once A is built
    const NetAll<MAXC> pt = NetAll<MAXC>::load(pins + s, cnt);
    const unsigned all = (1u << cnt) - 1u;
    int qn = 1;
    if (t == 0) {
        BsEntry e0; e0.prio = 0.0; e0.meta = (1ull << 16) | (1ull << st); e0.p0 = (u64)(unsigned)st; e0.p1 = 0ull; queue[0] = e0;
        A->mask = 0;  // "the points' set is not built yet" (a built table has mask >= 7)
    }
    wave_lds_order();
    BEAM_ACC(26);
    for (;;) {
        const int pops = k < qn ? k : qn;
        const bool worker = t < pops;
        // heappop number t: the (t+1)-th smallest (priority, path) of the queue, first index among fully equal entries
        int sel = 0;
        if (SMALL) {
            double pr[MAXQ];
            #pragma unroll
            for (int i = 0; i < MAXQ; i++) { const double q = queue[i].prio; pr[i] = i < qn ? q : __builtin_inf(); }  // stale slots: read, ranked last
            // rank of every entry by priority alone (six compares); two live entries of equal priority are rare and
            // take the general selection below, where python's path comparison breaks the tie
            int rk[MAXQ]; bool eq = false;
            #pragma unroll
            for (int i = 0; i < MAXQ; i++) rk[i] = 0;
            #pragma unroll
            for (int i = 0; i < MAXQ; i++) {
                #pragma unroll
                for (int j = i + 1; j < MAXQ; j++) {
                    const bool lt = pr[j] < pr[i];
                    rk[i] += lt ? 1 : 0; rk[j] += lt ? 0 : 1;
                    eq |= j < qn && pr[j] == pr[i];
                }
            }
            if (!eq) {
                const int want = t < pops ? t : pops - 1;
                #pragma unroll
                for (int i = 0; i < MAXQ; i++) if (rk[i] == want) sel = i;
            } else {
                unsigned taken = 0;
                for (int it = 0; it <= t && it < pops; it++) {
                    int best = -1; double bp = 0.0;
                    #pragma unroll
                    for (int i = 0; i < MAXQ; i++) {
                        if (i >= qn || (taken >> i & 1u)) continue;
                        bool less = best < 0 || pr[i] < bp;
                        if (best >= 0 && pr[i] == bp) less = path_less<MAXC>(queue[i], queue[best], pt);
                        if (less) { best = i; bp = pr[i]; }
                    }
                    taken |= 1u << best; sel = best;
                }
            }
        } else {
            unsigned taken = 0;
            for (int it = 0; it <= t && it < pops; it++) {
                int best = -1; double bp = 0.0;
                for (int i = 0; i < qn; i++) {
                    if (taken >> i & 1u) continue;
                    const double pi = queue[i].prio;
                    bool less = best < 0 || pi < bp;
                    if (best >= 0 && pi == bp) less = path_less<MAXC>(queue[i], queue[best], pt);
                    if (less) { best = i; bp = pi; }
                }
                taken |= 1u << best; sel = best;
            }
        }
        BEAM_ACC(27);
        const BsEntry e = queue[sel];
        const unsigned vis = e.visited();
        if (vis == all) {  // every entry of this level is a complete path: the first pop is the answer
            wave_lds_order();
            if (t == 0) *(BsEntry *)scratch = e;
            break;
        }
        const int cur = e.at(e.len() - 1);
        const int ux = pt.x(cur), uy = pt.y(cur);
        const int cntn = __popc(all & ~vis), take = cntn < k ? cntn : k;
        unsigned chosen[PCBENV_MAX_BEAM_WIDTH + 1];  // the k + 1 smallest keys, ascending
        if (SMALL) {
            unsigned key[MAXC];
            #pragma unroll
            for (int j = 0; j < MAXC; j++) {
                const int dx = ux - pt.x(j), dy = uy - pt.y(j);
                key[j] = (j < cnt && !(vis >> j & 1u)) ? (((unsigned)(dx * dx + dy * dy) << 4) | (unsigned)j) : KINF;
            }
            #pragma unroll
            for (int r = 0; r <= PCBENV_MAX_BEAM_WIDTH; r++) {
                if (r > 2) { chosen[r] = KINF; continue; }
                unsigned mn = KINF;
                #pragma unroll
                for (int j = 0; j < MAXC; j++) mn = min(mn, key[j]);
                chosen[r] = mn;
                #pragma unroll
                for (int j = 0; j < MAXC; j++) key[j] = key[j] == mn ? KINF : key[j];
            }
        } else {  // wide nets / beams: the keys are recomputed per round instead of held (register pressure, not speed)
            unsigned prev = 0u;
            #pragma unroll
            for (int r = 0; r <= PCBENV_MAX_BEAM_WIDTH; r++) {
                unsigned mn = KINF;
                if (r <= k) {
                    for (int j = 0; j < cnt; j++) {
                        const int dx = ux - pt.x(j), dy = uy - pt.y(j);
                        const unsigned kj = (((unsigned)(dx * dx + dy * dy) << 4) | (unsigned)j) + 1u;  // + 1: above `prev` = 0 in round 0
                        if (!(vis >> j & 1u) && kj > prev) mn = min(mn, kj);
                    }
                }
                prev = mn;
                chosen[r] = mn == KINF ? KINF : mn - 1u;
            }
        }
        bool tie = false;
        #pragma unroll
        for (int q = 1; q <= PCBENV_MAX_BEAM_WIDTH; q++) if (q == k && cntn > k) tie = (chosen[q - 1] >> 4) == (chosen[q] >> 4);
        BsEntry *dst = next + t * take;
        if (worker && !tie) {
            #pragma unroll
            for (int r = 0; r < PCBENV_MAX_BEAM_WIDTH; r++) {
                if (r >= take || (SMALL && r >= 2)) continue;
                const int j = (int)(chosen[r] & 15u);
                BsEntry w = e; w.push(j); w.prio = e.prio + norm2((double)(ux - pt.x(j)), (double)(uy - pt.y(j)));
                dst[r] = w;
            }
        }
        BEAM_ACC(28);
        // boundary tie: the CPython set order decides who is kept.  The model works on the points to visit (the
        // pins without the start pin, list order); the lanes of a net share its scratch and take turns.
        const bool my_tie = worker && tie;
        if (__ballot(my_tie) != 0ull)  // (of the lanes still searching)
        for (int turn = 0; turn < BEAM_LANES_PER_NET; turn++) {
            if (!(my_tie && t == turn)) continue;
            const int m = cnt - 1;
            const NetPts pv = NetPts::load(pins + s, cnt, st);
            const unsigned vpt = (vis & ((1u << st) - 1u)) | ((vis >> (st + 1)) << st);  // visited without the start pin's bit
            if (A->mask == 0) cs_build_points(A, R, hs, m, pv);  // first tie of this net
            int nset; unsigned packed = 0;
            const int nleft = m - (int)__popc(vpt);
            const bool small = !((m >> 2) > (int)__popc(vpt)) && nleft <= 4;
            if (small) nset = cs_small_difference_order(A, hs, vpt, pv, &packed);
            else {  // the general model works on a copy of the points' table (it recycles its first argument)
                CSet *A2 = (CSet *)dist;
                nset = cs_difference_order(A2, R, m, vpt, pv, order);
            }
            // sorted(key=distance) is stable: ties keep the iteration order; exact integer keys as above
            unsigned skey[4];
            if (small) {
                #pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int o = (int)((packed >> (8 * i)) & 0xFFu);
                    const int dx = ux - pv.x(o), dy = uy - pv.y(o);
                    skey[i] = i < nset ? (((unsigned)(dx * dx + dy * dy) << 8) | ((unsigned)i << 4) | (unsigned)o) : KINF;
                }
                #pragma unroll
                for (int r = 0; r < 4; r++) {  // `take` smallest (distance, position in the iteration order)
                    unsigned mn = KINF;
                    #pragma unroll
                    for (int i = 0; i < 4; i++) mn = min(mn, skey[i]);
                    #pragma unroll
                    for (int i = 0; i < 4; i++) skey[i] = skey[i] == mn ? KINF : skey[i];
                    if (r < take) {
                        const int o = (int)(mn & 15u);
                        BsEntry q = e; q.push(o + (o >= st ? 1 : 0));
                        q.prio = e.prio + norm2((double)(ux - pv.x(o)), (double)(uy - pv.y(o)));
                        dst[r] = q;
                    }
                }
            } else {
                for (int i = 0; i < nset; i++) dist[i] = norm2((double)(ux - pv.x(order[i])), (double)(uy - pv.y(order[i])));
                for (int i = 1; i < nset; i++) {
                    const unsigned char o = order[i]; const double dd = dist[i];
                    int j = i - 1;
                    while (j >= 0 && dist[j] > dd) { order[j + 1] = order[j]; dist[j + 1] = dist[j]; j--; }
                    order[j + 1] = o; dist[j + 1] = dd;
                }
                for (int i = 0; i < take; i++) { BsEntry q = e; q.push(order[i] + (order[i] >= st ? 1 : 0)); q.prio = e.prio + dist[i]; dst[i] = q; }
            }
        }
        wave_lds_order();
        BEAM_ACC(29);
        { BsEntry *tmp = queue; queue = next; next = tmp; }
        qn = pops * take;
        if (qn == 0) {  // cannot happen (an incomplete path always has an unvisited point); leave an empty route
            wave_lds_order();
            if (t == 0) { BsEntry z; z.prio = 0.0; z.meta = 0ull; z.p0 = 0ull; z.p1 = 0ull; *(BsEntry *)scratch = z; }
            break;
        }
    }
    wave_lds_order();
}
template <bool SMALL>
static __device__ __forceinline__ void beam_routes_lanes(const SegView &v, const PinRec *pins, int nn, int k, unsigned char *beam, int lane, unsigned long long *beam_dbg) {
    const int groups = NT / BEAM_LANES_PER_NET, gi = lane / BEAM_LANES_PER_NET, t = lane & (BEAM_LANES_PER_NET - 1);
    for (int n0 = 0; n0 < nn; n0 += groups) {
        const int n = n0 + gi;
        if (n >= nn) continue;
        const int s = v.nstart[n], cnt = v.nstart[n + 1] - s;
        // pin_outlier: first arg-max of the distance to the centroid (distances one pin per lane in v.D, below)
        int st = 0; double bd = v.D[s];
        if (SMALL) {
            double dd[8];
            #pragma unroll
            for (int i = 1; i < 8; i++) dd[i] = v.D[s + (i < cnt ? i : 0)];
            #pragma unroll
            for (int i = 1; i < 8; i++) if (i < cnt && dd[i] > bd) { bd = dd[i]; st = i; }
        } else {
            for (int i = 1; i < cnt; i++) { const double d = v.D[s + i]; if (d > bd) { bd = d; st = i; } }
        }
        beam_route_lanes<SMALL>(v, pins, s, cnt, st, k, beam + (size_t)n * BEAM_LDS_PER_NET(k), t, beam_dbg);
    }
}
// all nets of the environment: v.D / v.act / segment slots are filled with the beam routes
static __device__ __forceinline__ void beam_routes(const SegView &v, const EnvHdr *hdr, const PinRec *pins, int k, unsigned char *beam, int lane, unsigned long long *beam_dbg) {
    const int np = hdr->npins, nn = hdr->nnets;
    for (int q = lane; q < np; q += NT) {  // distance of every pin to its net's centroid (v.D is free until the segments are written)
        const PinRec pr = pins[q];
        v.D[q] = norm2((double)pr.abs_x - v.cen[pr.net], (double)pr.abs_y - v.cen[v.N + pr.net]);
    }
    lds_sync();
    // workgroup-uniform: the widest net of this instance decides which build runs (every wavefront looks at all nets)
    const int nl = lane & 63;
    const bool wide = __ballot(nl < nn && v.nstart[nl < nn ? nl + 1 : 0] - v.nstart[nl < nn ? nl : 0] > 8) != 0ull;
    if (!wide && k <= 2) beam_routes_lanes<true>(v, pins, nn, k, beam, lane, beam_dbg);
    else beam_routes_lanes<false>(v, pins, nn, k, beam, lane, beam_dbg);
    lds_sync();
    for (int q = lane; q < np; q += NT) {  // segment i of a net's route = (path[i], path[i + 1]), one slot per lane
        const int n = pins[q].net, s = v.nstart[n], i = q - s;
        const BsEntry res = *(const BsEntry *)(beam + (size_t)n * BEAM_LDS_PER_NET(k));
        const bool act = i + 1 < res.len();
        if (act) {
            const PinRec pa = pins[s + res.at(i)], pb = pins[s + res.at(i + 1)];
            const double x1 = pa.abs_x, y1 = pa.abs_y, x2 = pb.abs_x, y2 = pb.abs_y;
            v.X1[q] = x1; v.Y1[q] = y1; v.X2[q] = x2; v.Y2[q] = y2;
            v.D[q] = norm2(x1 - x2, y1 - y2);
        }
        v.act[q] = act ? 1 : 0;
    }
}

// beam (and, for "both", centroid) routes of the terminal state -> wirelength and this team's share of the #intersections
// of the beam route (wl[0], ni[0]) and, reward_type "both", of the centroid route (wl[1], ni[1]); the caller picks
// (S:609-627 lowest_num_intersections) once the shares of all teams are in.
static __device__ __forceinline__ void route_beam_or_both(const DevParams &p, const EnvHdr *hdr, const PinRec *pins, double *seg,
                                          int lane, int part, int nparts, double *wl, int *ni) {
    const SegView v = seg_view(seg, p.P, p.N);
    unsigned char *beam = v.beam;
    net_offsets_and_centroids(v, hdr, pins, lane);
    STAMP(5);
    STAMP_ZERO(26); STAMP_ZERO(27); STAMP_ZERO(28); STAMP_ZERO(29);
    beam_routes(v, hdr, pins, p.beam_width, beam, lane, p.dbg);
    lds_sync();
    STAMP(24);
    count_and_length(p, v, hdr, pins, lane, part, nparts, &wl[0], &ni[0]);
    STAMP(25);
    if (p.reward_type == PCBENV_REWARD_BOTH) {
        build_centroid_segments(v, hdr, pins, lane);
        count_and_length(p, v, hdr, pins, lane, part, nparts, &wl[1], &ni[1]);
    }
    STAMP(8);
}
