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
// Execution model: one wavefront per environment, and the generator -- an inherently serial program of a few
// thousand dependent steps -- is executed UNIFORMLY by all 64 lanes (every lane computes the same values; LDS
// writes come from lane 0).  That keeps every table in LDS (both MT19937 states, the component / net / cell
// lists, the record under construction) instead of per-lane scratch memory, whose ~1 us accesses made a
// lane-per-environment version no faster than the host generator, and it lets the few data-parallel pieces use the
// lanes: the block regeneration of MT19937 (624 words in ten 64-lane steps), the stable sort of the components by
// free space (rank sort), list.remove(), the zero fill and the copy-out of the record.
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
#define GSTAMP(k) do { if (lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g->stamps[k] = t_; } } while (0)
#else
#define GSTAMP(k) do { } while (0)
#endif
#define GEN_REC_MAX (16 + 8 * (PCBENV_MAX_COMPONENTS + PCBENV_MAX_PINS))
// One wavefront's LDS: ONE generator at a time and the record under construction.  NumPy's state serves steps 1-9
// of a record, CPython's step 10; they are swapped through HBM in between (2.5 KB each way, coalesced) so that the
// generator's wavefronts fit into the LDS the step kernel's workgroups leave free (of `rec` only instStride bytes
// are allocated: the block is dynamic shared memory).
struct GenLds {
    unsigned mt[624];
    unsigned long long rec[GEN_REC_MAX / 8];
};
#define GEN_LDS_BYTES(instStride) (624 * 4 + (int)(instStride))
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
typedef volatile LDS3 GenLds *GenLdsPtr;
typedef volatile LDS3 unsigned *LdsU32;
typedef volatile LDS3 int *LdsI32;
typedef volatile LDS3 unsigned long long *LdsU64;

// ---- MT19937 in LDS: block regeneration by the whole wavefront, words handed out one by one --------------------
__device__ inline void mt_regenerate(LdsU32 mt, int lane) {  // mt19937ar.c genrand_int32's refill, 64 words per step
    for (int base = 0; base < 624; base += WAVE) {
        const int kk = base + lane;
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
// One generator state between HBM and LDS: all ten loads of a lane go out together, then the stores (through the
// volatile LDS pointer every element would otherwise wait for the one before: 10 x a memory round trip per copy).
__device__ inline void mt_to_lds(LdsU32 dst, const unsigned *src, int lane) {
    unsigned t[10];
    #pragma unroll
    for (int r = 0; r < 10; r++) t[r] = r * WAVE + lane < 624 ? src[r * WAVE + lane] : 0u;
    #pragma unroll
    for (int r = 0; r < 10; r++) if (r * WAVE + lane < 624) dst[r * WAVE + lane] = t[r];
}
__device__ inline void mt_from_lds(unsigned *dst, LdsU32 src, int lane) {
    unsigned t[10];
    #pragma unroll
    for (int r = 0; r < 10; r++) t[r] = r * WAVE + lane < 624 ? src[r * WAVE + lane] : 0u;
    #pragma unroll
    for (int r = 0; r < 10; r++) if (r * WAVE + lane < 624) dst[r * WAVE + lane] = t[r];
}

// The stream of tempered words, 64 at a time in a register (lane i holds word base + i): a draw is one v_readlane
// instead of an LDS round trip.
struct MtReader {
    LdsU32 mt;
    int pos;        // words of the current block already handed out (624 = regenerate first)
    int base;       // first word index held in `cache`; -1 = none
    unsigned cache;
    __device__ unsigned next(int lane) {
        if (pos >= 624) { mt_regenerate(mt, lane); pos = 0; base = -1; }  // wave-uniform
        if (base < 0 || pos >= base + WAVE) {
            base = pos;
            unsigned v = base + lane < 624 ? mt[base + lane] : 0u;
            v ^= (v >> 11);
            v ^= (v << 7) & 0x9d2c5680u;
            v ^= (v << 15) & 0xefc60000u;
            v ^= (v >> 18);
            cache = v;
        }
        const unsigned out = (unsigned)__builtin_amdgcn_readlane((int)cache, __builtin_amdgcn_readfirstlane(pos - base));
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

// All of the below runs wave-uniformly: every lane holds the same scalars.  Small tables live one element per lane in
// registers (free space and id of the component at each sorted position, net probabilities, counts, the remaining
// cells of a component, the pin records) and are read with v_readlane / permutes; LDS holds the two generators and
// the record under construction.
__device__ inline int rl(int v, int idx) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(idx)); }
__device__ inline double rl(double v, int idx) {
    const int i = __builtin_amdgcn_readfirstlane(idx);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), i), __builtin_amdgcn_readlane(__double2loint(v), i));
}
struct NpStream {  // NumPy legacy RandomState pieces
    MtReader rd;
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
        while (U > px) {
            X++;
            if (X > bound) { X = 0; px = qn; U = dbl(); }
            else { U -= px; px = ((double)(n - X + 1) * p * px) / ((double)X * q); }
        }
        return X;
    }
    // RandomState.multinomial(n, p[0..d)), p one element per lane -> the count of bin `lane`.  The chain of legacy
    // binomials is sequential, but everything that does not depend on the draws -- the running remainder Sum, the
    // conditional probabilities p[j] / Sum, which tail the inversion works on and its log(q) -- is evaluated for all
    // bins at once first (the same operations on the same operands as numpy's loop).
    __device__ int multinomial(int n, double p_l, int d, bool *ok) {
        double Sum = 1.0, sum_l = 1.0;
        for (int j = 0; j < d - 1; j++) { if (lane == j) sum_l = Sum; Sum -= rl(p_l, j); }
        const double P_l = p_l / sum_l;                      // random_binomial(p = P_l, n = what is left)
        const bool upper_l = !(P_l <= 0.5);                  // p > 0.5: draw the complement with q = 1 - p
        const double pp_l = upper_l ? 1.0 - P_l : P_l, qq_l = 1.0 - pp_l, lg_l = log(qq_l);
        int cnt_l = 0;
        int dn = n;
        for (int j = 0; j < d - 1; j++) {
            const double P = rl(P_l, j);
            int x = 0;
            if (dn != 0 && P != 0.0) {
                const double pp = rl(pp_l, j);
                if (!(pp * (double)dn <= 30.0)) { *ok = false; return cnt_l; }  // BTPE would be needed: outside the sizes this library supports
                x = binomial_inversion(dn, pp, rl(qq_l, j), rl(lg_l, j));
                if (rl((int)upper_l, j)) x = dn - x;
            }
            if (lane == j) cnt_l = x;
            dn -= x;
            if (dn <= 0) break;
        }
        if (dn > 0 && lane == d - 1) cnt_l = dn;
        return cnt_l;
    }
};

// np.sum of a float64 array held one element per lane (pairwise summation with 8 accumulators, block 128 -- n <= 64 here)
__device__ inline double np_sum_lanes(double a_l, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += rl(a_l, i);
        return r;
    }
    double r[8];
    #pragma unroll
    for (int j = 0; j < 8; j++) r[j] = rl(a_l, j);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        #pragma unroll
        for (int j = 0; j < 8; j++) r[j] += rl(a_l, i + j);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += rl(a_l, i);
    return res;
}

// One record of the environment's stream into L->rec (wire format of include/pcbenv.h).  Wave-uniform.
__device__ inline int gen_record(const GenParams &c, GenLdsPtr L, GenState *g, NpStream &rs, int &py_pos_, int lane) {
    const int words = (int)(c.instStride / 8);
    GSTAMP(0);
    for (int i = lane; i < words; i += WAVE) L->rec[i] = 0ull;
    LdsI32 hdr = (LdsI32)L->rec;
    LdsU64 crec = L->rec + 2, prec = crec + c.C;
    // steps 1-2: lane i keeps component i (h | w << 8); a_s / ord = free space and id of the component at sorted position `lane`
    const int ncomp = rs.randint(c.min_comp, c.max_comp + 1);
    int total_area = 0, hw_l = 0, a_s = -1, ord = lane;
    for (int i = 0; i < ncomp; i++) {
        const int h = rs.randint(c.min_h, c.max_h + 1);
        const int w = rs.randint(c.min_w, c.max_w + 1);
        if (lane == i) { hw_l = h | (w << 8); a_s = h * w; }
        total_area += h * w;
    }
    GSTAMP(1);
    if (lane < ncomp) crec[lane] = (unsigned long long)hw_l;
    if (lane == 0) hdr[0] = ncomp;
    if (c.kind == PCBENV_RECT) return PCBENV_OK;
    // steps 3-4
    int nn = rs.randint(c.min_nets, c.max_nets + 1);
    if (nn > total_area / 2) nn = total_area / 2;
    int total = rs.randint(c.min_ppn * nn, c.max_ppn * nn + 1);
    if (total > total_area) total = total_area;
    if (nn < 1 || total > c.P || c.min_ppn * nn > total) return PCBENV_EINVAL;  // the reference raises here
    // step 5: softmax of normal samples (drawn even when unused); lane i keeps net i
    double pr_l = 0.0;
    for (int i = 0; i < nn; i++) {
        const double z = (1.0 / (double)nn) + (1.0 / (double)(c.net_distribution + 1)) * rs.legacy_gauss();
        if (lane == i) pr_l = z;
    }
    pr_l = exp(pr_l);
    const double sez = np_sum_lanes(pr_l, nn);
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
            const double sq = np_sum_lanes(q_l, nn);
            q_l = q_l / sq;
            extra_l += rs.multinomial(1, q_l, nn, &ok);
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
        const int extra_n = rl(extra_l, n);
        int unassigned = lo + extra_n;
        {   // stable sort of the positions by free space: rank = how many must precede this one, then one permute
            int rank = 0;
            for (int j = 0; j < ncomp; j++) {
                const int aj = rl(a_s, j);
                rank += (aj > a_s) | ((aj == a_s) & (j < lane));
            }
            if (lane >= ncomp) rank = lane;
            a_s = __builtin_amdgcn_ds_permute(rank << 2, a_s);
            ord = __builtin_amdgcn_ds_permute(rank << 2, ord);
        }
        int k;
        {   // the first k >= kcomp positions with enough room for the net (the reference grows k one by one)
            const int cum = wave_inclusive_scan(lane < ncomp ? a_s : 0, lane);
            const u64 enough = __ballot(lane >= kcomp - 1 && lane < ncomp && cum >= unassigned);
            if (!enough) return PCBENV_EINVAL;
            k = __ffsll((long long)enough);
        }
        int pin_in_net = 0;
        while (unassigned > 0) {
            const int cum = wave_inclusive_scan(lane < k ? a_s : 0, lane);
            const int tot = rl(cum, k - 1);
            const double prob_l = (double)a_s / (double)tot;
            const int cnt_l = rs.multinomial(unassigned, prob_l, k, &ok);
            if (!ok) return PCBENV_ELIMIT;
            const int m_l = lane < k ? min(cnt_l, a_s) : 0;
            a_s -= m_l;
            const int incl = wave_inclusive_scan(m_l, lane);
            const int assigned = rl(incl, WAVE - 1);
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
    mt_from_lds(g->np_mt, L->mt, lane);
    mt_to_lds(L->mt, g->py_mt, lane);
    GSTAMP(5);
    rs.rd.base = -1;  // (the words in LDS are CPython's now; NumPy's come back with the next record)
    MtReader py{L->mt, py_pos_, -1, 0u};
    // step 10: per component, random.choice over the remaining cells (row-major), pins in self.pins order.  Lane j of
    // chunk c keeps pin 64 c + j; a component's pins are a ballot; the cell list lives one cell per lane.
    unsigned long long rec_r[PCBENV_MAX_PINS / WAVE];
    #pragma unroll
    for (int ch = 0; ch < PCBENV_MAX_PINS / WAVE; ch++) rec_r[ch] = ch * WAVE + lane < total ? prec[ch * WAVE + lane] : ~0ull;
    for (int cid = 0; cid < ncomp; cid++) {
        const int hw = rl(hw_l, cid), w = hw >> 8;
        int ncell = (hw & 0xFF) * w;
        const unsigned wmagic = (65536u + (unsigned)w - 1u) / (unsigned)w;  // cell / w == (cell * wmagic) >> 16 for cell < 4160, w <= 16
        int cells_l = lane;
        #pragma unroll
        for (int ch = 0; ch < PCBENV_MAX_PINS / WAVE; ch++) {
            if (ch * WAVE >= total) break;
            u64 mine = __ballot((int)((rec_r[ch] >> 24) & 0xFF) == cid && rec_r[ch] != ~0ull);
            while (mine) {
                const int jl = __ffsll((long long)mine) - 1;
                mine &= mine - 1;
                int kbits = 0;
                for (int v = ncell; v; v >>= 1) kbits++;
                unsigned r;
                do { r = py.next(lane) >> (32 - kbits); } while ((int)r >= ncell);  // _randbelow_with_getrandbits
                const int cell = rl(cells_l, (int)r);
                const int nxt = __shfl_down(cells_l, 1);
                if (lane >= (int)r) cells_l = nxt;  // list.remove(value): cells are unique
                ncell--;
                const unsigned cx = ((unsigned)cell * wmagic) >> 16, cy = (unsigned)cell - cx * (unsigned)w;
                if (lane == jl) rec_r[ch] |= (unsigned long long)cx | ((unsigned long long)cy << 8);
            }
        }
    }
    GSTAMP(6);
    #pragma unroll
    for (int ch = 0; ch < PCBENV_MAX_PINS / WAVE; ch++) if (ch * WAVE + lane < total) prec[ch * WAVE + lane] = rec_r[ch];
    if (lane == 0) { hdr[1] = nn; hdr[2] = total; }
    mt_from_lds(g->py_mt, L->mt, lane);
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

// Tops up the queue of environment blockIdx.x (one wavefront): records produced[e] .. cursor + Q - 1 (slot = index
// % Q), never overwriting a record the environment has not consumed (the published cursor can only be behind the
// truth).  The records leave this XCD's L2 with the release fence at the end; the step kernels read them with
// agent-scope loads.
__global__ __attribute__((amdgpu_waves_per_eu(4, 8))) __launch_bounds__(WAVE) void k_gen_fill(GenParams c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gen_smem[];
    GenLdsPtr L = (GenLdsPtr)gen_smem;
    const int lane = threadIdx.x;
    const int words = (int)(c.instStride / 8);
    // The grid is capped (GEN_MAX_GRID): a wavefront walks over several environments, so that the generator never
    // holds more than a few wavefront slots per CU while the step kernels run next to it.
    for (int e = blockIdx.x; e < c.B; e += gridDim.x) {
        GenState *g = c.gen + e;
        const unsigned cursor = load_agent(c.cursor_pub + e);
        unsigned produced = c.produced[e];
        if (produced - cursor >= (unsigned)c.Q || g->status != 0) continue;  // nothing to do: the common case
        NpStream rs{MtReader{L->mt, g->np_pos, -1, 0u}, lane, g->has_gauss, g->gauss};
        int py_pos = g->py_pos, status = 0;
        while (produced - cursor < (unsigned)c.Q) {
            mt_to_lds(L->mt, g->np_mt, lane);
            rs.rd.base = -1;
            status = gen_record(c, L, g, rs, py_pos, lane);
            if (status != PCBENV_OK) break;  // (the stream stops here; its state is not needed any more)
            unsigned long long *dst = (unsigned long long *)(c.queue + ((size_t)(produced % (unsigned)c.Q) * c.B + e) * c.instStride);
            for (int i = lane; i < words; i += WAVE) dst[i] = L->rec[i];
            produced++;
            if (c.kind == PCBENV_RECT) mt_from_lds(g->np_mt, L->mt, lane);  // (the others swapped it out before step 10)
        }
        if (lane == 0) { g->np_pos = rs.rd.pos; g->py_pos = py_pos; g->has_gauss = rs.has_gauss; g->gauss = rs.gauss; g->status = status; }
        __threadfence();  // the records before the count
        if (lane == 0) c.produced[e] = produced;
    }
}
