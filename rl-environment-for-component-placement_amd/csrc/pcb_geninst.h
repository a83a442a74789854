// pcb_geninst.h -- on-device instance generator: the reference's generate_instances() as a kernel, one lane per stream
// Part of libpcbenv.so's single translation unit (included by pcbenv_kernels.hip); CDNA4 / gfx950 only.
//
// The reference draws a fresh placement problem at EVERY reset (S:1487-1549 -> generate_instances S:960-989 and
// helpers :931-1212, :1408-1443, sample_truncated_multinomial :250-287; P:1006-1265; R:253-273) from the global
// NumPy legacy stream and the global CPython `random` stream.  At 10^8 env-steps/s that is 10^7 instances/s, far
// beyond what host cores generate, so the generator runs on the device: every environment owns its two MT19937
// states in HBM (seeded like `np.random.seed(s); random.seed(s)`), and k_gen_fill -- one lane per environment, on a
// side stream, off the step kernel's critical path -- tops up the environment's instance queue with the next
// records of its stream, in the wire format of include/pcbenv.h.  The draw order is SURVEY.md Appendix A; the code
// below restates csrc/instance_gen.cpp (the host twin, which tests compare it with record by record):
//
//   NumPy   MT19937 seeded by init_genrand(seed); randint = masked rejection on 32-bit outputs; normal = legacy
//           polar Box-Muller with its cached second value; multinomial = chain of legacy binomials (inversion
//           algorithm -- every call on this path has n*p <= 30, checked)
//   CPython MT19937 seeded by init_by_array([seed]); choice(seq) = seq[_randbelow(len)] with
//           getrandbits(k) = genrand_uint32() >> (32 - k)
//
// Execution model: the generator is an inherently serial program of a few thousand dependent steps per record.  A
// GROUP of G lanes (16, 32 or 64: as many as the configuration has components / nets, a quarter of its pins) serves
// one environment and executes that program UNIFORMLY (every lane of the group computes the same scalars), 64 / G
// environments per wavefront side by side.  That keeps every table in registers (one element per group lane) or LDS
// instead of per-lane scratch memory, whose ~1 us accesses made a lane-per-environment version no faster than the
// host generator, and it lets the data-parallel pieces use the group's lanes: the block regeneration of MT19937,
// the stable sort of the components by free space (rank sort + permute), scans, list.remove(), the copy-out.
// exp / log come from the device math library: like the host twin's libm they may differ from NumPy's SIMD
// kernels in the last bit of a probability, which can change a table only if a uniform variate lands within
// ~1 ulp of a threshold (~1e-15 per draw; see instance_gen.cpp).
#pragma once
#include "pcb_device.h"

struct GenState {            // per environment, in HBM
    unsigned np_mt[624];
    unsigned py_mt[624];
    int np_pos, py_pos;      // words of the current block already handed out (624 = regenerate first)
    int has_gauss, status;   // status: 0 ok, else the PCBENV_* code of the first record that could not be generated
    double gauss;
#ifdef GEN_STAMPS
    unsigned long long stamps[10];  // diagnostic builds of tools/gen_harness.hip only: s_memtime at the phases of the last record
#endif
};
#ifdef GEN_STAMPS
#define GSTAMP(k) do { if ((lane & (G - 1)) == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g->stamps[k] = t_; } } while (0)
#else
#define GSTAMP(k) do { } while (0)
#endif
#define GEN_REC_MAX (16 + 8 * (PCBENV_MAX_COMPONENTS + PCBENV_MAX_PINS))
// A group's LDS holds ONE generator state at a time and the record under construction: NumPy's state serves steps
// 1-9 of a record, CPython's step 10; they are swapped through HBM in between (2.5 KB each way, coalesced) so that the
// generator's wavefronts fit into the LDS the step kernel's workgroups leave free.
#define GEN_MAX_GRID 2048

struct GenParams {           // by value kernel argument
    int kind, C, P, Q, B;
    int min_comp, max_comp, min_h, max_h, min_w, max_w;
    int min_nets, max_nets, min_ppn, max_ppn, net_distribution, pin_spread;
    long long instStride;
    unsigned char *queue;           // instance queue [Q][B][instStride]
    const unsigned *cursor_pub;     // [B] queue cursors as the reset kernels publish them (DevParams::cursor_pub)
    GenState *gen;                  // [B]
    unsigned *produced;             // [B] records generated so far (slot = produced % Q)
};

// LDS-qualified pointers: the accesses compile to ds_* instructions with 32-bit addresses (generic pointers would be
// 64-bit flat accesses, and the kernel would live in spilled address registers)
#define LDS3 __attribute__((address_space(3)))
typedef volatile LDS3 unsigned *LdsU32;
typedef volatile LDS3 int *LdsI32;
typedef volatile LDS3 unsigned long long *LdsU64;

// ---- G lanes per environment ------------------------------------------------------------------------------------
// A wavefront generates 64 / G records at once: environment `lane / G` of its batch lives in the G lanes of its
// group, and everything below that says "uniform" means uniform WITHIN A GROUP.  G = 16 when a configuration has at
// most 16 components, 16 nets and 64 pins (c3, c4: four records per wavefront), 32 up to 32 / 32 / 128 (c5), else 64
// (gen_group_lanes below).  Groups diverge freely (different loop counts); every cross-lane operation
// stays inside a group: v_readlane becomes ds_bpermute (per-lane source), the scans stop at the group width, ballots
// are shifted and masked to the group.
template <int G> __device__ inline int grl(int v, int idx, int lane) {
    if (G == WAVE) return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(idx));
    return __builtin_amdgcn_ds_bpermute(((lane & ~(G - 1)) | idx) << 2, v);
}
template <int G> __device__ inline double grl(double v, int idx, int lane) {
    return __hiloint2double(grl<G>(__double2hiint(v), idx, lane), grl<G>(__double2loint(v), idx, lane));
}
template <int G> __device__ inline int gscan(int x, int lane) {  // inclusive prefix sum inside the group (DPP)
    const int row = lane & 15;
    int t;
    t = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false); if (row >= 1) x += t;   // row_shr:1
    t = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false); if (row >= 2) x += t;   // row_shr:2
    t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false); if (row >= 4) x += t;   // row_shr:4
    t = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false); if (row >= 8) x += t;   // row_shr:8
    if (G >= 32) { t = __builtin_amdgcn_update_dpp(0, x, 0x142, 0xF, 0xF, false); if ((lane & 31) >= 16) x += t; }  // row_bcast:15
    if (G == 64) { t = __builtin_amdgcn_update_dpp(0, x, 0x143, 0xF, 0xF, false); if (lane >= 32) x += t; }         // row_bcast:31
    return x;
}
template <int G> __device__ inline u64 gballot(bool pred, int lane) {  // the group's bits of the ballot, at bit 0
    const u64 b = __ballot(pred);
    if (G == WAVE) return b;
    return (b >> (lane & ~(G - 1))) & ((1ull << G) - 1ull);
}

// ---- MT19937 in LDS: block regeneration by the G lanes of the group, words handed out one by one -----------------
template <int G> __device__ inline void mt_regenerate(LdsU32 mt, int gl) {  // mt19937ar.c genrand_int32's refill, G words per step
    for (int base = 0; base < 624; base += G) {
        const int kk = base + gl;
        unsigned v = 0;
        if (kk < 624) {  // mt[kk + 1] is still the old word (its owner writes after this read), mt[kk + 397 - 624] already the new one
            const int k1 = kk + 1 == 624 ? 0 : kk + 1, km = kk + 397 >= 624 ? kk + 397 - 624 : kk + 397;
            const unsigned y = (mt[kk] & 0x80000000u) | (mt[k1] & 0x7fffffffu);
            v = mt[km] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        __builtin_amdgcn_wave_barrier();
        if (kk < 624) mt[kk] = v;
        __builtin_amdgcn_wave_barrier();
    }
}
// One generator state between HBM and LDS, ten words of a lane in flight at a time (through the volatile LDS pointer
// every element would otherwise wait for the one before).
template <int G> __device__ inline void mt_to_lds(LdsU32 dst, const unsigned *src, int gl) {
    for (int b0 = 0; b0 < 624; b0 += 10 * G) {
        unsigned t[10];
        #pragma unroll
        for (int r = 0; r < 10; r++) t[r] = b0 + r * G + gl < 624 ? src[b0 + r * G + gl] : 0u;
        #pragma unroll
        for (int r = 0; r < 10; r++) if (b0 + r * G + gl < 624) dst[b0 + r * G + gl] = t[r];
    }
}
template <int G> __device__ inline void mt_from_lds(unsigned *dst, LdsU32 src, int gl) {
    for (int b0 = 0; b0 < 624; b0 += 10 * G) {
        unsigned t[10];
        #pragma unroll
        for (int r = 0; r < 10; r++) t[r] = b0 + r * G + gl < 624 ? src[b0 + r * G + gl] : 0u;
        #pragma unroll
        for (int r = 0; r < 10; r++) if (b0 + r * G + gl < 624) dst[b0 + r * G + gl] = t[r];
    }
}
// The stream of tempered words, G at a time in a register (group lane i holds word base + i): a draw is one
// cross-lane read instead of an LDS round trip.
template <int G> struct MtReader {
    LdsU32 mt;
    int pos;        // words of the current block already handed out (624 = regenerate first)
    int base;       // first word index held in `cache`; -1 = none
    unsigned cache;
    __device__ unsigned next(int lane) {
        const int gl = lane & (G - 1);
        if (pos >= 624) { mt_regenerate<G>(mt, gl); pos = 0; base = -1; }
        if (base < 0 || pos >= base + G) {
            base = pos;
            unsigned v = base + gl < 624 ? mt[base + gl] : 0u;
            v ^= (v >> 11);
            v ^= (v << 7) & 0x9d2c5680u;
            v ^= (v << 15) & 0xefc60000u;
            v ^= (v >> 18);
            cache = v;
        }
        const unsigned out = (unsigned)grl<G>((int)cache, pos - base, lane);
        pos++;
        return out;
    }
};
__device__ inline void mt_init_genrand(unsigned *mt, unsigned s) {  // mt19937ar.c init_genrand (one lane, global memory)
    mt[0] = s;
    for (int i = 1; i < 624; i++) { s = 1812433253u * (s ^ (s >> 30)) + (unsigned)i; mt[i] = s; }
}
__device__ inline void mt_init_by_array(unsigned *mt, const unsigned *key, int len) {  // CPython random_seed
    mt_init_genrand(mt, 19650218u);
    int i = 1, j = 0;
    for (int k = (624 > len ? 624 : len); k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (unsigned)j;
        i++; j++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
        if (j >= len) j = 0;
    }
    for (int k = 623; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (unsigned)i;
        i++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
}

// All of the below runs uniformly within a group: every lane of the group holds the same scalars.  Small tables live
// one element per group lane in registers (free space and id of the component at each sorted position, net
// probabilities, counts, the pin records) and are read with cross-lane reads / permutes; LDS holds one generator
// state and the record under construction, per group.
#ifdef GEN_MARGIN
// Diagnostic build of tools/gen_harness.hip only: how close any draw-dependent decision of a multinomial came to its
// threshold.  exp / log of the device library may differ from NumPy's in the last bit, which changes a probability by
// ~1e-16 relative; a record can only differ if a uniform variate (or a probability against 0.5) lands that close to a
// threshold.  [0] = smallest relative distance |U - px| / px seen (double bits; positive doubles order like integers),
// [1] = the same for |P - 0.5| / 0.5, [2] = comparisons recorded, [3 + k] = how many of them were below 10^-(6 + 2 k)
// (k < 4), [7] = comparisons exactly on the threshold.
__device__ unsigned long long gen_margin[8];
__device__ inline void margin_note(double a, double thr, int which) {
    const double rel = fabs(a - thr) / (thr > 0.0 ? thr : 1.0);
    atomicAdd(&gen_margin[2], 1ull);
    // exactly on the threshold: only seen for P == 0.5 from two equal integer free spaces (a / (a + a), an exact division on
    // both sides -- nothing exp / log could move); counted apart
    if (rel == 0.0) { atomicAdd(&gen_margin[7], 1ull); return; }
    atomicMin(&gen_margin[which], (unsigned long long)__double_as_longlong(rel));
    for (int k = 0; k < 4; k++) if (rel < pow(10.0, -(6.0 + 2.0 * k))) atomicAdd(&gen_margin[3 + k], 1ull);
}
#define MARGIN(a, thr, which) margin_note(a, thr, which)
#else
#define MARGIN(a, thr, which) do { } while (0)
#endif
#ifdef GEN_COUNT_FALLBACK
__device__ unsigned gen_fallbacks[2];
#endif
template <int G> struct NpStream {  // NumPy legacy RandomState pieces
    MtReader<G> rd;
    int lane, has_gauss;
    double gauss;
    __device__ unsigned u32() { return rd.next(lane); }
    __device__ double dbl() {  // 53 bits from two outputs
        const unsigned a = u32() >> 5, b = u32() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    __device__ int randint(int low, int high) {  // high exclusive; every range on this path fits 31 bits
        const unsigned rng = (unsigned)(high - 1 - low);
        if (rng == 0) return low;
        unsigned mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        unsigned val;
        do { val = u32() & mask; } while (val > rng);
        return low + (int)val;
    }
    __device__ double legacy_gauss() {
        if (has_gauss) { const double t = gauss; has_gauss = 0; gauss = 0.0; return t; }
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * dbl() - 1.0;
            x2 = 2.0 * dbl() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        f = sqrt(-2.0 * log(r2) / r2);
        gauss = f * x1;
        has_gauss = 1;
        return f * x2;
    }
    // legacy_random_binomial_inversion(n, p) with q = 1 - p and lg = log(q) supplied (computed for all bins at once)
    __device__ int binomial_inversion(int n, double p, double q, double lg) {
        const double qn = exp((double)n * lg), np = (double)n * p;
        const double b = np + 10.0 * sqrt(np * q + 1);
        const int bound = (int)((double)n < b ? (double)n : b);
        int X = 0;
        double px = qn, U = dbl();
        MARGIN(U, px, 0);
        while (U > px) {
            X++;
            if (X > bound) { X = 0; px = qn; U = dbl(); }
            else { U -= px; px = ((double)(n - X + 1) * p * px) / ((double)X * q); }
            MARGIN(U, px, 0);
        }
        return X;
    }
    // RandomState.multinomial(n, p[0..d)), p one element per group lane -> the count of bin `gl`.  NumPy runs a chain
    // of legacy binomials, bin j on whatever is left of n.  Everything that does not depend on the draws -- the
    // running remainder Sum, the conditional probabilities p[j] / Sum, which tail the inversion works on and its
    // log(q) -- is evaluated for all bins at once first (the same operations on the same operands as numpy's loop).
    // Then the fast path: a bin's binomial consumes exactly one uniform variate (two words of the stream) unless it is
    // redrawn (X > bound, never seen), so bin j's variate is known in advance, and lane j tabulates its inversion for
    // EVERY possible remainder 1..n (n <= 15: four bits each); the chain itself is then a table walk.  Whenever an
    // assumption of the fast path does not hold (block refill inside the chain, a redraw, n > 15, a bin beyond the
    // inversion algorithm's range) the chain runs one binomial after the other as numpy does.
    __device__ int multinomial(int n, double p_l, int d, bool *ok, bool p_from_exp = false) {  // p_from_exp: diagnostics only (GEN_MARGIN)
        const int gl = lane & (G - 1);
        double Sum = 1.0, sum_l = 1.0;
        for (int j = 0; j < d - 1; j++) { if (gl == j) sum_l = Sum; Sum -= grl<G>(p_l, j, lane); }
        const double P_l = p_l / sum_l;                      // random_binomial(p = P_l, n = what is left)
        const bool upper_l = !(P_l <= 0.5);                  // p > 0.5: draw the complement with q = 1 - p
        // (only probabilities that come out of exp() can be moved by its last bit: step 9's are ratios of integers)
        if (p_from_exp && gl < d - 1 && P_l != 0.0) MARGIN(P_l, 0.5, 1);
        const double pp_l = upper_l ? 1.0 - P_l : P_l, qq_l = 1.0 - pp_l, lg_l = log(qq_l);
        const bool bin_l = gl < d - 1, draws_l = bin_l && P_l != 0.0;   // random_binomial returns 0 without a draw for p == 0
        if (rd.pos >= 624) { mt_regenerate<G>(rd.mt, gl); rd.pos = 0; rd.base = -1; }
        const int incl = gscan<G>(draws_l ? 1 : 0, lane);
        const int ndraw = grl<G>(incl, G - 1, lane);
        bool fast = n <= 15 && rd.pos + 2 * ndraw <= 624;
#ifdef GEN_COUNT_FALLBACK
        if (lane == 0) atomicAdd(gen_fallbacks + 1, 1u);
#endif
        u64 tab_l = 0ull;
        if (fast) {
            bool bad_l = false;
            if (draws_l) {
                const int w0 = rd.pos + 2 * (incl - 1);
                unsigned a = rd.mt[w0], b = rd.mt[w0 + 1];
                a ^= (a >> 11); a ^= (a << 7) & 0x9d2c5680u; a ^= (a << 15) & 0xefc60000u; a ^= (a >> 18);
                b ^= (b >> 11); b ^= (b << 7) & 0x9d2c5680u; b ^= (b << 15) & 0xefc60000u; b ^= (b >> 18);
                const double U0 = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
                for (int m = 1; m <= n; m++) {  // legacy_random_binomial_inversion(m, pp) on this bin's variate
                    if (!(pp_l * (double)m <= 30.0)) { bad_l = true; break; }
                    const double qn = exp((double)m * lg_l), np = (double)m * pp_l;
                    const double bb = np + 10.0 * sqrt(np * qq_l + 1);
                    const int bound = (int)((double)m < bb ? (double)m : bb);
                    int X = 0;
                    double px = qn, U = U0;
                    MARGIN(U, px, 0);
                    while (U > px) {
                        X++;
                        if (X > bound) { bad_l = true; break; }
                        U -= px; px = ((double)(m - X + 1) * pp_l * px) / ((double)X * qq_l);
                        MARGIN(U, px, 0);
                    }
                    if (bad_l) break;
                    tab_l |= (u64)(unsigned)(upper_l ? m - X : X) << (4 * m);
                }
            }
            if (gballot<G>(bad_l, lane)) fast = false;
        }
        int cnt_l = 0, dn = n;
        if (fast) {
            const u64 draws = gballot<G>(draws_l, lane);
            int used = 0;
            for (int j = 0; j < d - 1; j++) {
                int x = 0;
                if (dn != 0 && ((draws >> j) & 1ull)) {
                    const unsigned lo32 = (unsigned)grl<G>((int)(unsigned)tab_l, j, lane), hi32 = (unsigned)grl<G>((int)(unsigned)(tab_l >> 32), j, lane);
                    x = (int)(((((u64)hi32 << 32) | lo32) >> (4 * dn)) & 15ull);
                    used += 2;
                }
                if (gl == j) cnt_l = x;
                dn -= x;
                if (dn <= 0) break;
            }
            rd.pos += used;
        } else {
#ifdef GEN_COUNT_FALLBACK
            atomicAdd(gen_fallbacks, 1u);
#endif
            for (int j = 0; j < d - 1; j++) {
                const double P = grl<G>(P_l, j, lane);
                int x = 0;
                if (dn != 0 && P != 0.0) {
                    const double pp = grl<G>(pp_l, j, lane);
                    if (!(pp * (double)dn <= 30.0)) { *ok = false; return cnt_l; }  // BTPE would be needed: outside the sizes this library supports
                    x = binomial_inversion(dn, pp, grl<G>(qq_l, j, lane), grl<G>(lg_l, j, lane));
                    if (grl<G>((int)upper_l, j, lane)) x = dn - x;
                }
                if (gl == j) cnt_l = x;
                dn -= x;
                if (dn <= 0) break;
            }
        }
        if (dn > 0 && gl == d - 1) cnt_l = dn;
        return cnt_l;
    }
};

// np.sum of a float64 array held one element per group lane (pairwise summation with 8 accumulators, block 128 -- n <= 64 here)
template <int G> __device__ inline double np_sum_lanes(double a_l, int n, int lane) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += grl<G>(a_l, i, lane);
        return r;
    }
    double r[8];
    #pragma unroll
    for (int j = 0; j < 8; j++) r[j] = grl<G>(a_l, j, lane);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        #pragma unroll
        for (int j = 0; j < 8; j++) r[j] += grl<G>(a_l, i + j, lane);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += grl<G>(a_l, i, lane);
    return res;
}

// The group's LDS: one generator state and the record.
struct GrpLds { LdsU32 mt; LdsU64 rec; };
#define GEN_GROUP_LDS_BYTES(instStride) (624 * 4 + (int)(instStride))

// One record of the group's stream into L.rec (wire format of include/pcbenv.h).  Uniform within the group.
template <int G> __device__ inline int gen_record(const GenParams &c, const GrpLds &L, GenState *g, NpStream<G> &rs, int &py_pos_, int lane) {
    const int gl = lane & (G - 1), gbase = lane & ~(G - 1);
    const int words = (int)(c.instStride / 8);
    GSTAMP(0);
    for (int i = gl; i < words; i += G) L.rec[i] = 0ull;
    LdsI32 hdr = (LdsI32)L.rec;
    LdsU64 crec = L.rec + 2, prec = crec + c.C;
    // steps 1-2: group lane i keeps component i (h | w << 8); a_s / ord = free space and id of the component at sorted position `gl`
    const int ncomp = rs.randint(c.min_comp, c.max_comp + 1);
    int total_area = 0, hw_l = 0, a_s = -1, ord = gl;
    for (int i = 0; i < ncomp; i++) {
        const int h = rs.randint(c.min_h, c.max_h + 1);
        const int w = rs.randint(c.min_w, c.max_w + 1);
        if (gl == i) { hw_l = h | (w << 8); a_s = h * w; }
        total_area += h * w;
    }
    GSTAMP(1);
    if (gl < ncomp) crec[gl] = (unsigned long long)hw_l;
    if (gl == 0) hdr[0] = ncomp;
    if (c.kind == PCBENV_RECT) return PCBENV_OK;
    // steps 3-4
    int nn = rs.randint(c.min_nets, c.max_nets + 1);
    if (nn > total_area / 2) nn = total_area / 2;
    int total = rs.randint(c.min_ppn * nn, c.max_ppn * nn + 1);
    if (total > total_area) total = total_area;
    if (nn < 1 || total > c.P || c.min_ppn * nn > total) return PCBENV_EINVAL;  // the reference raises here
    // step 5: softmax of normal samples (drawn even when unused); group lane i keeps net i
    double pr_l = 0.0;
    for (int i = 0; i < nn; i++) {
        const double z = (1.0 / (double)nn) + (1.0 / (double)(c.net_distribution + 1)) * rs.legacy_gauss();
        if (gl == i) pr_l = z;
    }
    pr_l = exp(pr_l);
    const double sez = np_sum_lanes<G>(pr_l, nn, lane);
    pr_l = pr_l / sez;
    GSTAMP(2);
    // steps 6-7: creation ids -> nets
    const int lo = c.min_ppn;
    int extra_l = 0;
    const int rem = total - lo * nn;
    bool ok = true;
    if (c.max_ppn > lo && rem > 0) {
        const int k = min(c.max_ppn - lo, rem);
        for (int t = 0; t < rem; t++) {
            double q_l = pr_l * (extra_l < k ? 1.0 : 0.0);
            const double sq = np_sum_lanes<G>(q_l, nn, lane);
            q_l = q_l / sq;
            extra_l += rs.multinomial(1, q_l, nn, &ok, true);
            if (!ok) return PCBENV_ELIMIT;
        }
    }
    GSTAMP(3);
    // step 8
    int kcomp;
    if (c.kind == PCBENV_SPATIAL) kcomp = min((int)(((double)c.pin_spread / 10.0) * (double)ncomp) + 1, ncomp);
    else kcomp = min(max((int)(((double)(c.pin_spread + 1) / 10.0) * (double)ncomp), 1), ncomp);
    // step 9: net by net in net order; the components stay sorted by free space (descending, stable) from net to net
    int q_idx = 0, id_cursor = lo * nn;  // first output pin of the net; first creation id of the net's extra pins
    for (int n = 0; n < nn; n++) {
        const int extra_n = grl<G>(extra_l, n, lane);
        int unassigned = lo + extra_n;
        {   // stable sort of the positions by free space: rank = how many must precede this one, then one permute
            int rank = 0;
            for (int j = 0; j < ncomp; j++) {
                const int aj = grl<G>(a_s, j, lane);
                rank += (aj > a_s) | ((aj == a_s) & (j < gl));
            }
            if (gl >= ncomp) rank = gl;
            a_s = __builtin_amdgcn_ds_permute((gbase + rank) << 2, a_s);
            ord = __builtin_amdgcn_ds_permute((gbase + rank) << 2, ord);
        }
        int k;
        {   // the first k >= kcomp positions with enough room for the net (the reference grows k one by one)
            const int cum = gscan<G>(gl < ncomp ? a_s : 0, lane);
            const u64 enough = gballot<G>(gl >= kcomp - 1 && gl < ncomp && cum >= unassigned, lane);
            if (!enough) return PCBENV_EINVAL;
            k = __ffsll((long long)enough);
        }
        int pin_in_net = 0;
        while (unassigned > 0) {
            const int cum = gscan<G>(gl < k ? a_s : 0, lane);
            const int tot = grl<G>(cum, k - 1, lane);
            const double prob_l = (double)a_s / (double)tot;
            const int cnt_l = rs.multinomial(unassigned, prob_l, k, &ok);
            if (!ok) return PCBENV_ELIMIT;
            const int m_l = gl < k ? min(cnt_l, a_s) : 0;
            a_s -= m_l;
            const int incl = gscan<G>(m_l, lane);
            const int assigned = grl<G>(incl, G - 1, lane);
            for (int t = 0; t < m_l; t++) {
                // creation id of the pi-th pin of net n (spatial); index in this batch (pin env, quirk Q1)
                const int pi = pin_in_net + incl - m_l + t;
                const int id = c.kind == PCBENV_SPATIAL ? (pi < lo ? n * lo + pi : id_cursor + pi - lo) : t;
                prec[q_idx + pi] = ((unsigned long long)n << 16) | ((unsigned long long)ord << 24) | ((unsigned long long)(id & 0xFFFF) << 32);
            }
            pin_in_net += assigned;
            unassigned -= assigned;
        }
        q_idx += pin_in_net;
        id_cursor += extra_n;
    }
    GSTAMP(4);
    // NumPy's generator leaves LDS, CPython's comes in
    mt_from_lds<G>(g->np_mt, L.mt, gl);
    mt_to_lds<G>(L.mt, g->py_mt, gl);
    GSTAMP(5);
    rs.rd.base = -1;  // (the words in LDS are CPython's now; NumPy's come back with the next record)
    MtReader<G> py{L.mt, py_pos_, -1, 0u};
    // step 10: per component, random.choice over the remaining cells (row-major), pins in self.pins order.  Group lane j of
    // chunk ch keeps pin G ch + j (total <= 4 G); a component's pins are a ballot; the cell list is in LDS.
    unsigned long long rec_r[4];
    #pragma unroll
    for (int ch = 0; ch < 4; ch++) rec_r[ch] = ch * G + gl < total ? prec[ch * G + gl] : ~0ull;
    for (int cid = 0; cid < ncomp; cid++) {
        const int hw = grl<G>(hw_l, cid, lane), w = hw >> 8;
        int ncell = (hw & 0xFF) * w;
        const unsigned wmagic = (65536u + (unsigned)w - 1u) / (unsigned)w;  // cell / w == (cell * wmagic) >> 16 for cell < 4160, w <= 16
        // `cells` = the component's cells in row-major order with the used ones removed: a bit set.  cells[r] is the
        // r-th set bit, list.remove() clears it (the list stays sorted, so order is preserved as in the reference).
        u64 cells = ncell >= 64 ? ~0ull : ((1ull << ncell) - 1ull);
        #pragma unroll
        for (int ch = 0; ch < 4; ch++) {
            u64 mine = gballot<G>((int)((rec_r[ch] >> 24) & 0xFF) == cid && rec_r[ch] != ~0ull, lane);
            while (mine) {
                const int jl = __ffsll((long long)mine) - 1;
                mine &= mine - 1;
                const int kbits = 32 - __clz(ncell);
                unsigned r;
                do { r = py.next(lane) >> (32 - kbits); } while ((int)r >= ncell);  // _randbelow_with_getrandbits
                int cell = 0, k = (int)r;
                {   // position of the k-th set bit: binary search on popcounts
                    u64 wbits = cells;
                    #pragma unroll
                    for (int width = 32; width >= 1; width >>= 1) {
                        const int cpop = __popcll(wbits & ((1ull << width) - 1ull));
                        if (k >= cpop) { k -= cpop; wbits >>= width; cell += width; }
                    }
                }
                cells &= ~(1ull << cell);
                ncell--;
                const unsigned cx = ((unsigned)cell * wmagic) >> 16, cy = (unsigned)cell - cx * (unsigned)w;
                if (gl == jl) rec_r[ch] |= (unsigned long long)cx | ((unsigned long long)cy << 8);
            }
        }
    }
    GSTAMP(6);
    #pragma unroll
    for (int ch = 0; ch < 4; ch++) if (ch * G + gl < total) prec[ch * G + gl] = rec_r[ch];
    if (gl == 0) { hdr[1] = nn; hdr[2] = total; }
    mt_from_lds<G>(g->py_mt, L.mt, gl);
    GSTAMP(7);
    py_pos_ = py.pos;
    return PCBENV_OK;
}

// Seeds the two generators of every environment like `np.random.seed(s); random.seed(s)` (s < 2^32).
__global__ __launch_bounds__(WAVE) void k_gen_seed(GenParams c, const unsigned *__restrict__ seeds) {
    const int e = blockIdx.x * WAVE + threadIdx.x;
    if (e >= c.B) return;
    GenState *g = c.gen + e;
    const unsigned s = seeds[e];
    mt_init_genrand(g->np_mt, s);
    mt_init_by_array(g->py_mt, &s, 1);
    g->np_pos = 624; g->py_pos = 624; g->has_gauss = 0; g->status = 0; g->gauss = 0.0;
    // the stream's first record is the environment's next reset: it goes where the queue cursor points
    c.produced[e] = load_agent(c.cursor_pub + e);
}

// Tops up the queues: group `lane / G` of a wavefront serves one environment at a time -- records produced[e] ..
// cursor + Q - 1 (slot = index % Q), never overwriting a record the environment has not consumed (the published cursor
// can only be behind the truth).  The records leave this XCD's L2 with the release fence at the end; the step kernels
// read them with agent-scope loads.  The grid is capped (GEN_MAX_GRID): a wavefront walks over several batches of
// environments, so that the generator never holds more than a few wavefront slots per CU next to the step kernels.
template <int G>
__global__ __attribute__((amdgpu_waves_per_eu(4, 8))) __launch_bounds__(WAVE) void k_gen_fill(GenParams c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gen_smem[];
    constexpr int EPW = WAVE / G;
    const int lane = threadIdx.x, grp = lane / G, gl = lane & (G - 1);
    const int words = (int)(c.instStride / 8);
    const int per_group = (GEN_GROUP_LDS_BYTES(c.instStride) + 15) & ~15;
    GrpLds L;
    L.mt = (LdsU32)(gen_smem + grp * per_group);
    L.rec = (LdsU64)(gen_smem + grp * per_group + 624 * 4);
    for (int e0 = blockIdx.x * EPW; e0 < c.B; e0 += gridDim.x * EPW) {
        const int e = e0 + grp;
        if (e >= c.B) continue;
        GenState *g = c.gen + e;
        const unsigned cursor = load_agent(c.cursor_pub + e);
        unsigned produced = c.produced[e];
        if (produced - cursor >= (unsigned)c.Q || g->status != 0) continue;  // nothing to do: the common case
        NpStream<G> rs{MtReader<G>{L.mt, g->np_pos, -1, 0u}, lane, g->has_gauss, g->gauss};
        int py_pos = g->py_pos, status = 0;
        while (produced - cursor < (unsigned)c.Q) {
            mt_to_lds<G>(L.mt, g->np_mt, gl);
            rs.rd.base = -1;
            status = gen_record<G>(c, L, g, rs, py_pos, lane);
            if (status != PCBENV_OK) break;  // (the stream stops here; its state is not needed any more)
            unsigned long long *dst = (unsigned long long *)(c.queue + ((size_t)(produced % (unsigned)c.Q) * c.B + e) * c.instStride);
            for (int i = gl; i < words; i += G) dst[i] = L.rec[i];
            produced++;
            if (c.kind == PCBENV_RECT) mt_from_lds<G>(g->np_mt, L.mt, gl);  // (the others swapped it out before step 10)
        }
        if (gl == 0) { g->np_pos = rs.rd.pos; g->py_pos = py_pos; g->has_gauss = rs.has_gauss; g->gauss = rs.gauss; g->status = status; }
        __threadfence();  // the records before the count
        if (gl == 0) c.produced[e] = produced;
    }
}
// lanes per environment for a configuration (see above); LDS bytes of a refill launch
// Measured on c3 (bench.py fresh-instance leg, 4 096 environments, same box): G = 64: 126 M env-steps/s, G = 32: 145 M,
// G = 16: 156 M (replayed queue: 200 M).  Packing pays although a cross-lane read is a ds_bpermute round trip at G < 64
// instead of a v_readlane: with the multinomial chains tabulated, the dependent chain of a record is short enough.
// `forced` = 32 / 64 (PCBENV_OPT_GEN_LANES) selects a wider group than the narrowest possible (experiments, tests).
static inline int gen_group_lanes(int C, int N, int P, int forced = 0) {
    int g = (C <= 16 && N <= 16 && P <= 64) ? 16 : (C <= 32 && N <= 32 && P <= 128) ? 32 : 64;
    if ((forced == 32 || forced == 64) && forced > g) g = forced;
    return g;
}
#define GEN_LDS_BYTES(instStride, G) ((WAVE / (G)) * ((GEN_GROUP_LDS_BYTES(instStride) + 15) & ~15))
