// pcb_reward.h -- float64 geometry of the routing reward: centroid routes, exact extent pre-filter, intersection count, wirelength
// Part of libpcbenv.so's single translation unit (included by pcbenv_kernels.hip); CDNA4 / gfx950 only.

// ----------------------------------------------------------------------------------------------
// float64 geometry of the reward (one IEEE operation per operator, see file header)
// ----------------------------------------------------------------------------------------------
// S:1288-1301 euclidean_distance == np.linalg.norm == sqrt(ddot): sqrt(fma(dy, dy, dx*dx))
static __device__ inline double norm2(double dx, double dy) { return __dsqrt_rn(__fma_rn(dy, dy, __dmul_rn(dx, dx))); }

// S:653-702 is_intersect
static __device__ inline bool is_intersect(double x1, double y1, double x2, double y2, double x3, double y3, double x4, double y4) {
    if ((x1 == x3 && y1 == y3) || (x1 == x4 && y1 == y4) || (x2 == x3 && y2 == y3) || (x2 == x4 && y2 == y4)) return true;
    double det = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
    if (det == 0) return false;
    double a = x1 * y2 - y1 * x2, b = x3 * y4 - y3 * x4;
    double x = (a * (x3 - x4) - (x1 - x2) * b) / det;
    double y = (a * (y3 - y4) - (y1 - y2) * b) / det;
    return fmin(x1, x2) <= x && x <= fmax(x1, x2) && fmin(x3, x4) <= x && x <= fmax(x3, x4) &&
           fmin(y1, y2) <= y && y <= fmax(y1, y2) && fmin(y3, y4) <= y && y <= fmax(y3, y4);
}

// ---- routes -------------------------------------------------------------------------------------
// A route is kept as one segment slot per pin q (slots of net n are nstart[n]..nstart[n+1]-1, so slots are
// net-major like the reference's route lists); act[q] = 1 if the slot carries a segment.
struct SegView { double *X1, *Y1, *X2, *Y2, *D, *A, *DX, *DY, *cen; int *act, *nstart, *nsum; unsigned *bbox; unsigned short *pairs; unsigned char *beam; int N; };  // N = max_num_nets: the per-net tables' size
// compaction buffer of candidate (i, j) pairs, per wavefront: a dense batch is two candidates per lane (128), a
// sweep step appends at most 4 * 64 to a partial batch (< 128), and what does not fill a batch is moved to the front
#define PAIR_ENTRIES_PER_WAVE 384
// [segments X1 Y1 X2 Y2 D | centroids | act nstart total nsum] then a zone used only by the pair count
// (A DX DY bbox pairs), which the beam search -- finished before the count starts -- overlays with its per-net scratch.
// (the per-net tables are sized by the configuration's max_num_nets N, not by PCBENV_MAX_NETS: at c3 / c4 that and a fold
// scratch sized by need bring a workgroup's LDS from 7.8 to 6.6 KB -- 24 instead of 20 one-wavefront workgroups per CU,
// the headroom the reward helpers start in)
#define SEG_INTS(P, N) ((P) + ((N) + 1) + 3 + 2 * (N))
#define SEG_FIXED_BYTES(P, N) ((5 * (P) + 2 * (N)) * 8 + SEG_INTS(P, N) * 4)
#define SEG_COUNT_BYTES(P, NW) (3 * (P) * 8 + (P) * 4 + PAIR_ENTRIES_PER_WAVE * 2 * (NW))
#define SEG_LDS_BYTES(P, N, NW, beam) (((SEG_FIXED_BYTES(P, N) + 7) & ~7) + ((beam) > SEG_COUNT_BYTES(P, NW) ? (beam) : SEG_COUNT_BYTES(P, NW)))
static __device__ inline SegView seg_view(double *seg, int P, int N) {
    SegView v;
    v.N = N;
    v.X1 = seg; v.Y1 = seg + P; v.X2 = seg + 2 * P; v.Y2 = seg + 3 * P; v.D = seg + 4 * P;
    v.cen = seg + 5 * P;                              // cx[N], cy[N]
    v.act = (int *)(v.cen + 2 * N);                   // [P]
    v.nstart = v.act + P;                             // [N + 1], then 3 spare words (nstart[N + 1] = pair counter)
    v.nsum = v.nstart + N + 1 + 3;                    // [2 * N] integer coordinate sums per net
    v.beam = (unsigned char *)seg + ((SEG_FIXED_BYTES(P, N) + 7) & ~7);
    v.A = (double *)v.beam; v.DX = v.A + P; v.DY = v.A + 2 * P;  // per segment: x1*y2 - y1*x2, x1 - x2, y1 - y2
    v.bbox = (unsigned *)(v.A + 3 * P);               // [P] integer extents (x_lo, x_hi, y_lo, y_hi), one byte each
    v.pairs = (unsigned short *)(v.bbox + P);         // [PAIR_ENTRIES_PER_WAVE] per wavefront
    return v;
}

// net_pins offsets (self.pins is net-major) and S:1229-1241 get_centroid per net: np.mean of an integer array =
// (exact integer sum, as float64) / n -- the sums are gathered with LDS integer atomics (order-free because exact).
static __device__ inline void net_offsets_and_centroids(const SegView &v, const EnvHdr *hdr, const PinRec *pins, int lane) {
    const int np = hdr->npins, nn = hdr->nnets;
    lds_sync();  // the segment area aliases the class map of emit_pin_grid
    for (int n = lane; n < 2 * v.N; n += NT) v.nsum[n] = 0;
    lds_sync();
    for (int q = lane; q < np; q += NT) {
        const PinRec pr = pins[q];
        if (q == 0 || pr.net != pins[q - 1].net) v.nstart[pr.net] = q;
        atomicAdd(&v.nsum[pr.net], (int)pr.abs_x);
        atomicAdd(&v.nsum[v.N + pr.net], (int)pr.abs_y);
    }
    if (lane == 0) v.nstart[nn] = np;
    lds_sync();
    for (int n = lane; n < nn; n += NT) {
        const double cnt = (double)(v.nstart[n + 1] - v.nstart[n]);
        v.cen[n] = (double)v.nsum[n] / cnt;
        v.cen[v.N + n] = (double)v.nsum[v.N + n] / cnt;
    }
    lds_sync();
}

// S:1243-1271 route_pins_centroid: (pin, centroid) per pin; a 2-pin net is the single segment (p0, p1)
static __device__ inline void build_centroid_segments(const SegView &v, const EnvHdr *hdr, const PinRec *pins, int lane) {
    const int np = hdr->npins;
    for (int q = lane; q < np; q += NT) {
        const int n = pins[q].net, s = v.nstart[n], cnt = v.nstart[n + 1] - s;
        double x1 = pins[q].abs_x, y1 = pins[q].abs_y, x2, y2;
        int a = 1;
        if (cnt == 2) { a = (q == s); x2 = pins[s + 1].abs_x; y2 = pins[s + 1].abs_y; }
        else { x2 = v.cen[n]; y2 = v.cen[v.N + n]; }
        v.X1[q] = x1; v.Y1[q] = y1; v.X2[q] = x2; v.Y2[q] = y2; v.act[q] = a;
        v.D[q] = norm2(x1 - x2, y1 - y2);
    }
    lds_sync();
}

// is_intersect (S:653-702) on two slots, with the per-segment terms hoisted: the operations and their order are
// exactly the reference's -- (x1*y2 - y1*x2), (x1 - x2), (y1 - y2) are sub-expressions of its formulas.
// Written without branches so that several candidates per lane can be in flight at once (the count is bound by
// the LDS and float64 division latency of one wavefront, not by issue slots): det == 0 gives inf / NaN
// coordinates, which is harmless and masked by the explicit test.
static __device__ inline bool slots_intersect(const SegView &v, int i, int j) {
    const double x1 = v.X1[i], y1 = v.Y1[i], x2 = v.X2[i], y2 = v.Y2[i];
    const double x3 = v.X1[j], y3 = v.Y1[j], x4 = v.X2[j], y4 = v.Y2[j];
    const double dxi = v.DX[i], dyi = v.DY[i], dxj = v.DX[j], dyj = v.DY[j];
    const double a = v.A[i], b = v.A[j];
    const bool shared = ((x1 == x3) & (y1 == y3)) | ((x1 == x4) & (y1 == y4)) | ((x2 == x3) & (y2 == y3)) | ((x2 == x4) & (y2 == y4));
    const double det = dxi * dyj - dyi * dxj;
    const double x = (a * dxj - dxi * b) / det;
    const double y = (a * dyj - dyi * b) / det;
    const bool inside = (fmin(x1, x2) <= x) & (x <= fmax(x1, x2)) & (fmin(x3, x4) <= x) & (x <= fmax(x3, x4)) &
                        (fmin(y1, y2) <= y) & (y <= fmax(y1, y2)) & (fmin(y3, y4) <= y) & (y <= fmax(y3, y4));
    return shared | ((det != 0) & inside);
}
// Exact pre-filter: if the closed x- (or y-) extents of the two segments are disjoint, no x (y) can lie in both,
// so the reference's final range test fails whatever the computed intersection point is (a shared end point,
// its only early "True", puts a common point in both extents).  The extents are kept as conservatively rounded
// integers (floor of the minimum, ceil of the maximum; coordinates are in [0, 127]), four bytes per segment, so
// the filter is one LDS word per segment and a few integer compares; a pair it lets through is decided by the
// full float64 test, a pair it rejects has disjoint real extents.  Saves the two float64 divisions.
static __device__ inline unsigned pack_extents(double x1, double y1, double x2, double y2) {
    const unsigned xl = (unsigned)floor(fmin(x1, x2)), xh = (unsigned)ceil(fmax(x1, x2));
    const unsigned yl = (unsigned)floor(fmin(y1, y2)), yh = (unsigned)ceil(fmax(y1, y2));
    return xl | (xh << 8) | (yl << 16) | (yh << 24) | 0x80000000u;  // bit 31 = slot carries a segment
}
static __device__ inline bool extents_overlap(unsigned a, unsigned b) {  // branch-free
    const unsigned xl = max(a & 0xFFu, b & 0xFFu), xh = min((a >> 8) & 0xFFu, (b >> 8) & 0xFFu);
    const unsigned yl = max((a >> 16) & 0xFFu, (b >> 16) & 0xFFu), yh = min((a >> 24) & 0x7Fu, (b >> 24) & 0x7Fu);
    return ((a & b & 0x80000000u) != 0) & (xl <= xh) & (yl <= yh);
}

// Full test on candidates [0, n) of a wavefront's buffer, two per lane and step so that their LDS reads and
// divisions overlap.
typedef __attribute__((address_space(3))) unsigned short lds_u16;  // keeps the buffer accesses ds_* instead of flat_*
static __device__ inline int count_candidates(const SegView &v, const volatile lds_u16 *buf, int n, int wl_lane) {
    int cnt = 0;
    for (int base = 0; base < n; base += 2 * WAVE) {
        const int i0 = base + wl_lane, i1 = i0 + WAVE;
        const unsigned short p0 = i0 < n ? buf[i0] : (unsigned short)0, p1 = i1 < n ? buf[i1] : (unsigned short)0;
        const bool r0 = slots_intersect(v, p0 & 0xFF, p0 >> 8), r1 = slots_intersect(v, p1 & 0xFF, p1 >> 8);
        cnt += ((i0 < n) & r0) + ((i1 < n) & r1);
    }
    return cnt;
}

// S:629-651 find_num_intersection + S:704-722 find_wirelength over the slots.
// Slots are net-major, so the partners "segment of an earlier net" of the slots of net n are the slots
// 0..nstart[n]-1: net n contributes the block nstart[n] x (its own slots) of pairs.  The blocks are swept one
// after the other, 64 pairs at a time, the sweep steps dealt out to the wavefronts: every lane
// of every sweep step holds one pair, whatever the shape of the block (the triangle of the old per-slot sweep left
// half the lanes idle and cost one step per partner).  A pair is first tested on its packed integer extents; the
// survivors of a step are appended to the wavefront's compaction buffer with a ballot, and the buffer goes through
// the full float64 test in dense batches of 128.  The wirelength is summed sequentially in route order (bit-exact
// with the reference's python float loop).
// count_prepare reads the pins, count_finish only the segment zone.
static __device__ inline void count_prepare(const SegView &v, int np, int lane) {
    int *total_cnt = v.nstart + v.N + 1;  // spare slot behind nstart[0..MAX_NETS]
    for (int q = lane; q < np; q += NT) {
        const double x1 = v.X1[q], y1 = v.Y1[q], x2 = v.X2[q], y2 = v.Y2[q];
        v.A[q] = x1 * y2 - y1 * x2; v.DX[q] = x1 - x2; v.DY[q] = y1 - y2;
        v.bbox[q] = v.act[q] ? pack_extents(x1, y1, x2, y2) : 0u;
    }
    if (lane == 0) *total_cnt = 0;
    lds_sync();
}
// (part, nparts): the sweep steps are dealt to `nparts` teams of a launch (the environment's own wavefront and its reward
// helpers, see run_env), this team being number `part`; *nintersections is then this team's share of the count.
static __device__ inline void count_finish(const DevParams &p, const SegView &v, int np_, int nn_, int lane, int part, int nparts, double *wirelength, int *nintersections) {
    int *total_cnt = v.nstart + v.N + 1;
    const int np = __builtin_amdgcn_readfirstlane(np_), nn = __builtin_amdgcn_readfirstlane(nn_);
    STAMP(12);
#ifndef PCBENV_STAMPS_BEAM
    STAMP_ZERO(26); STAMP_ZERO(27); STAMP_ZERO(28); STAMP_ZERO(29);
#endif
    const int wl_lane = lane & 63, wave = lane >> 6, nwaves = NT / WAVE;
    volatile lds_u16 *buf = (volatile lds_u16 *)(v.pairs + wave * PAIR_ENTRIES_PER_WAVE);  // wave-synchronous: written and read by different lanes
    int cnt = 0, nbuf = 0, step = 0;
    int pm = 0, pq = 0;  // step % nparts, step / nparts (kept incrementally: no division by a run-time value)
    for (int n = 1; n < nn; n++) {  // wave-uniform; net 0 has no earlier net
        const int s = __builtin_amdgcn_readfirstlane(v.nstart[n]), c = __builtin_amdgcn_readfirstlane(v.nstart[n + 1]) - s;
        const int R = s * c;                         // pairs (i, j): i in [0, s) earlier-net slot, j in [s, s + c)
        const unsigned magic = (65536u + (unsigned)c - 1u) / (unsigned)max(c, 1);  // r / c == (r * magic) >> 16 for r < 4160, c <= 16
        // four 64-pair groups per sweep step: their LDS reads go out together and the loop overhead is paid once
        for (int base = 0; base < R; base += 4 * WAVE, step++) {
            const bool mine = pm == part && (pq & (nwaves - 1)) == wave;  // nwaves is 1 or 4
            if (++pm == nparts) { pm = 0; pq++; }
            if (!mine) continue;
            unsigned bi[4], bj[4]; int pi[4], pj[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = base + u * WAVE + wl_lane;
                const bool in = r < R;
                pi[u] = in ? (int)(((unsigned)r * magic) >> 16) : 0;
                pj[u] = in ? s + (r - pi[u] * c) : 0;
                bi[u] = v.bbox[pi[u]]; bj[u] = v.bbox[pj[u]];
                if (!in) bi[u] = 0u;  // fails the "both slots carry a segment" bit
            }
            #pragma unroll
            for (int u = 0; u < 4; u++) {  // at most 4 * 64 new entries: the buffer holds them next to a partial batch
                const bool pass = extents_overlap(bi[u], bj[u]);
                const u64 ball = __ballot(pass);
                if (pass) buf[nbuf + __popcll(ball & ((1ull << wl_lane) - 1ull))] = (unsigned short)(pi[u] | (pj[u] << 8));
                nbuf += __popcll(ball);
            }
            while (nbuf >= 2 * WAVE) {  // dense batches; the rest (< 128 entries) moves to the front
                STAMP_T0();
                cnt += count_candidates(v, buf, 2 * WAVE, wl_lane);
#ifndef PCBENV_STAMPS_BEAM
                STAMP_ACC_SINCE(26, cnt); STAMP_ADD(27, 1); STAMP_ADD(28, 2 * WAVE);
#endif
                nbuf -= 2 * WAVE;
                for (int k = wl_lane; k < nbuf; k += WAVE) { const unsigned short rest = buf[2 * WAVE + k]; buf[k] = rest; }
            }
        }
    }
    STAMP(13);
#ifndef PCBENV_STAMPS_BEAM
    STAMP_ADD(28, nbuf); STAMP_ADD(29, step);
#endif
    cnt += count_candidates(v, buf, nbuf, wl_lane);
    STAMP(14);
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (wl_lane == 0 && cnt) atomicAdd(total_cnt, cnt);
    lds_sync();
    STAMP(15);
    // find_wirelength: the adds happen in route order; empty slots add +0.0, which leaves a non-negative sum
    // unchanged bit for bit.  Every lane fetches the lengths of its own slots once (one LDS round trip), the sum
    // then runs over v_readlane broadcasts.
    double wl = 0.0;
    for (int base = 0; base < np; base += WAVE) {
        const int sidx = base + wl_lane;
        const double d = (sidx < np && v.act[sidx]) ? v.D[sidx] : 0.0;
        const int dlo = __double2loint(d), dhi = __double2hiint(d);
        #pragma unroll
        for (int blk = 0; blk < WAVE; blk += 16) {  // constant lane selects: the broadcasts run ahead of the add chain
            if (base + blk >= np) break;
            #pragma unroll
            for (int il = blk; il < blk + 16; il++)
                wl += __hiloint2double(__builtin_amdgcn_readlane(dhi, il), __builtin_amdgcn_readlane(dlo, il));
        }
    }
    *wirelength = wl;
    *nintersections = *total_cnt;
    lds_sync();
}
static __device__ inline void count_and_length(const DevParams &p, const SegView &v, const EnvHdr *hdr, const PinRec *pins, int lane, int part, int nparts,
                                        double *wirelength, int *nintersections) {
    const int np = hdr->npins, nn = hdr->nnets;
    count_prepare(v, np, lane);
    count_finish(p, v, np, nn, lane, part, nparts, wirelength, nintersections);
}

static __device__ inline void route_centroid(const DevParams &p, const EnvHdr *hdr, const PinRec *pins, double *seg,
                                      int lane, int part, int nparts, double *wirelength, int *nintersections) {
    const SegView v = seg_view(seg, p.P, p.N);
    net_offsets_and_centroids(v, hdr, pins, lane);
    STAMP(5);
    build_centroid_segments(v, hdr, pins, lane);
    STAMP(6);
    count_prepare(v, hdr->npins, lane);
    STAMP(22);
    count_finish(p, v, hdr->npins, hdr->nnets, lane, part, nparts, wirelength, nintersections);
    STAMP(8);
}

