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
};
#define GEN_REC_MAX (16 + 8 * (PCBENV_MAX_COMPONENTS + PCBENV_MAX_PINS))
struct GenLds {              // one wavefront's working set
    unsigned np_mt[624], py_mt[624];
    double pr[PCBENV_MAX_NETS], q[PCBENV_MAX_NETS], probs[PCBENV_MAX_COMPONENTS];
    int cnt[PCBENV_MAX_COMPONENTS], sample[PCBENV_MAX_NETS];
    short avail[PCBENV_MAX_COMPONENTS];
    unsigned char hs[PCBENV_MAX_COMPONENTS], ws[PCBENV_MAX_COMPONENTS], order[PCBENV_MAX_COMPONENTS], order2[PCBENV_MAX_COMPONENTS];
    unsigned char extra[PCBENV_MAX_NETS], cells[PCBENV_MAX_PINS_PER_COMPONENT];
    unsigned long long rec[GEN_REC_MAX / 8];
};

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
typedef volatile LDS3 const double *LdsCF64;
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
__device__ inline unsigned mt_word(LdsU32 mt, int &pos, int lane) {
    if (pos >= 624) { mt_regenerate(mt, lane); pos = 0; }  // wave-uniform
    unsigned v = mt[pos++];
    v ^= (v >> 11);
    v ^= (v << 7) & 0x9d2c5680u;
    v ^= (v << 15) & 0xefc60000u;
    v ^= (v >> 18);
    return v;
}
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

// All of the below runs wave-uniformly: every lane holds the same scalars; `L` is the wavefront's LDS block, written
// by lane 0 (W) and read by everyone.
#define W(lhs, v) do { if (lane == 0) (lhs) = (v); } while (0)
struct NpStream {  // NumPy legacy RandomState pieces
    GenLdsPtr L;
    int pos, lane, has_gauss;
    double gauss;
    __device__ unsigned u32() { return mt_word(L->np_mt, pos, lane); }
    __device__ double dbl() {  // 53 bits from two outputs
        const unsigned a = u32() >> 5, b = u32() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    __device__ long long randint(long long low, long long high) {  // high exclusive
        const unsigned long long rng = (unsigned long long)(high - 1 - low);
        if (rng == 0) return low;
        unsigned long long mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        unsigned val;
        do { val = u32() & (unsigned)mask; } while (val > rng);
        return low + (long long)val;
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
    __device__ long long binomial_inversion(long long n, double p) {  // legacy_random_binomial_inversion
        const double q = 1.0 - p, qn = exp((double)n * log(q)), np = (double)n * p;
        const double b = np + 10.0 * sqrt(np * q + 1);
        const long long bound = (long long)((double)n < b ? (double)n : b);
        long long X = 0;
        double px = qn, U = dbl();
        while (U > px) {
            X++;
            if (X > bound) { X = 0; px = qn; U = dbl(); }
            else { U -= px; px = ((double)(n - X + 1) * p * px) / ((double)X * q); }
        }
        return X;
    }
    __device__ long long binomial(double p, long long n, bool *ok) {  // as RandomState.multinomial reaches it
        if (n == 0 || p == 0.0) return 0;
        if (p <= 0.5) {
            if (p * (double)n <= 30.0) return binomial_inversion(n, p);
        } else {
            const double q = 1.0 - p;
            if (q * (double)n <= 30.0) return n - binomial_inversion(n, q);
        }
        *ok = false;  // BTPE would be needed: outside the sizes this library supports
        return 0;
    }
    // RandomState.multinomial(n, p[0..d)) -> out[0..d) (both in LDS)
    __device__ void multinomial(long long n, LdsCF64 p, int d, LdsI32 out, bool *ok) {
        double Sum = 1.0;
        long long dn = n;
        for (int j = lane; j < d; j += WAVE) out[j] = 0;
        for (int j = 0; j < d - 1; j++) {
            const double pj = p[j];
            const int x = (int)binomial(pj / Sum, dn, ok);
            W(out[j], x);
            dn -= x;
            if (dn <= 0) break;
            Sum -= pj;
        }
        if (dn > 0) W(out[d - 1], (int)dn);
    }
};

// np.sum of a contiguous float64 array (pairwise summation with 8 accumulators, block 128 -- n <= 128 here)
__device__ inline double np_sum_dev(LdsCF64 a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3]; r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}

// One record of the environment's stream into L->rec (wire format of include/pcbenv.h).  Wave-uniform.
__device__ inline int gen_record(const GenParams &c, GenLdsPtr L, NpStream &rs, int &py_pos, int lane) {
    const int words = (int)(c.instStride / 8);
    for (int i = lane; i < words; i += WAVE) L->rec[i] = 0ull;
    LdsI32 hdr = (LdsI32)L->rec;
    LdsU64 crec = L->rec + 2, prec = crec + c.C;
    // steps 1-2
    const int ncomp = (int)rs.randint(c.min_comp, (long long)c.max_comp + 1);
    int total_area = 0;
    for (int i = 0; i < ncomp; i++) {
        const int h = (int)rs.randint(c.min_h, (long long)c.max_h + 1);
        const int w = (int)rs.randint(c.min_w, (long long)c.max_w + 1);
        if (lane == 0) {
            L->hs[i] = (unsigned char)h; L->ws[i] = (unsigned char)w; L->avail[i] = (short)(h * w); L->order[i] = (unsigned char)i;
            crec[i] = (unsigned long long)h | ((unsigned long long)w << 8);
        }
        total_area += h * w;
    }
    W(hdr[0], ncomp);
    if (c.kind == PCBENV_RECT) return PCBENV_OK;
    // steps 3-4
    int nn = (int)rs.randint(c.min_nets, (long long)c.max_nets + 1);
    if (nn > total_area / 2) nn = total_area / 2;
    int total = (int)rs.randint((long long)c.min_ppn * nn, (long long)c.max_ppn * nn + 1);
    if (total > total_area) total = total_area;
    if (nn < 1 || total > c.P || c.min_ppn * nn > total) return PCBENV_EINVAL;  // the reference raises here
    // step 5: softmax of normal samples (drawn even when unused)
    for (int i = 0; i < nn; i++) {
        const double z = (1.0 / (double)nn) + (1.0 / (double)(c.net_distribution + 1)) * rs.legacy_gauss();
        W(L->pr[i], exp(z));
    }
    const double sez = np_sum_dev(L->pr, nn);
    if (lane < nn) L->pr[lane] = L->pr[lane] / sez;
    // steps 6-7: creation ids -> nets
    const int lo = c.min_ppn;
    if (lane < nn) L->extra[lane] = 0;
    const int rem = total - lo * nn;
    bool ok = true;
    if (c.max_ppn > lo && rem > 0) {
        const int k = min(c.max_ppn - lo, rem);
        for (int t = 0; t < rem; t++) {
            if (lane < nn) L->q[lane] = L->pr[lane] * (L->extra[lane] < k ? 1.0 : 0.0);
            const double sq = np_sum_dev(L->q, nn);
            if (lane < nn) L->q[lane] = L->q[lane] / sq;
            rs.multinomial(1, L->q, nn, L->sample, &ok);
            if (lane < nn) L->extra[lane] = (unsigned char)(L->extra[lane] + L->sample[lane]);
        }
    }
    // step 8
    int kcomp;
    if (c.kind == PCBENV_SPATIAL) kcomp = min((int)(((double)c.pin_spread / 10.0) * (double)ncomp) + 1, ncomp);
    else kcomp = min(max((int)(((double)(c.pin_spread + 1) / 10.0) * (double)ncomp), 1), ncomp);
    // step 9: net by net in net order; `order` = component ids by free space, descending, stable, carried over
    int q_idx = 0, id_cursor = lo * nn;  // first output pin of the net; first creation id of the net's extra pins
    for (int n = 0; n < nn; n++) {
        const int npins_net = lo + L->extra[n];
        int unassigned = npins_net;
        {   // stable sort, descending free space: rank = elements that must precede this one (ncomp <= 64: one lane each)
            const int mine = lane < ncomp ? L->order[lane] : 0, a = lane < ncomp ? L->avail[mine] : -1;
            int rank = 0;
            for (int j = 0; j < ncomp; j++) {
                const int aj = L->avail[L->order[j]];
                rank += (aj > a) | ((aj == a) & (j < lane));
            }
            if (lane < ncomp) L->order2[rank] = (unsigned char)mine;
            if (lane < ncomp) L->order[lane] = L->order2[lane];
        }
        int k = kcomp - 1, space = 0;
        while (space < unassigned) {
            k += 1;
            space = 0;
            for (int i = 0; i < k && i < ncomp; i++) space += L->avail[L->order[i]];
            if (k > ncomp + 1) return PCBENV_EINVAL;
        }
        if (k > ncomp) k = ncomp;
        int pin_in_net = 0;
        while (unassigned > 0) {
            int tot = 0;
            for (int i = 0; i < k; i++) tot += L->avail[L->order[i]];
            if (lane < k) L->probs[lane] = (double)L->avail[L->order[lane]] / (double)tot;
            rs.multinomial(unassigned, L->probs, k, L->cnt, &ok);
            if (!ok) return PCBENV_ELIMIT;
            for (int i = 0; i < k; i++) {
                const int cid = L->order[i], av = L->avail[cid];
                int m = L->cnt[i];
                if (av < m) m = av;
                W(L->avail[cid], (short)(av - m));
                // creation id of the pin_in_net-th pin of net n (spatial); index in this batch (pin env, quirk Q1)
                if (lane < m) {
                    const int pi = pin_in_net + lane;
                    const int id = c.kind == PCBENV_SPATIAL ? (pi < lo ? n * lo + pi : id_cursor + pi - lo) : lane;
                    prec[q_idx + pi] = ((unsigned long long)n << 16) | ((unsigned long long)cid << 24) | ((unsigned long long)(id & 0xFFFF) << 32);
                }
                pin_in_net += m;
                unassigned -= m;
            }
        }
        q_idx += pin_in_net;
        id_cursor += L->extra[n];
    }
    if (!ok) return PCBENV_ELIMIT;
    // step 10: per component, random.choice over the remaining cells (row-major), pins in self.pins order
    for (int cid = 0; cid < ncomp; cid++) {
        const int w = L->ws[cid];
        int ncell = L->hs[cid] * w;
        if (lane < ncell) L->cells[lane] = (unsigned char)lane;
        for (int j = 0; j < total; j++) {
            const unsigned long long rec = prec[j];
            if ((int)((rec >> 24) & 0xFF) != cid) continue;
            int kbits = 0;
            for (int v = ncell; v; v >>= 1) kbits++;
            unsigned r;
            do { r = mt_word(L->py_mt, py_pos, lane) >> (32 - kbits); } while ((int)r >= ncell);  // _randbelow_with_getrandbits
            const int cell = L->cells[r];
            const int nxt = (lane >= (int)r && lane + 1 < ncell) ? L->cells[lane + 1] : 0;  // list.remove(value): cells are unique
            __builtin_amdgcn_wave_barrier();
            if (lane >= (int)r && lane + 1 < ncell) L->cells[lane] = (unsigned char)nxt;
            ncell--;
            W(prec[j], rec | (unsigned long long)(cell / w) | ((unsigned long long)(cell % w) << 8));
        }
    }
    W(hdr[1], nn);
    W(hdr[2], total);
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
    __shared__ GenLds lds;
    GenLdsPtr L = (GenLdsPtr)&lds;
    const int e = blockIdx.x, lane = threadIdx.x;
    GenState *g = c.gen + e;
    const unsigned cursor = load_agent(c.cursor_pub + e);
    unsigned produced = c.produced[e];
    if (produced - cursor >= (unsigned)c.Q || g->status != 0) return;  // nothing to do: the common case
    for (int i = lane; i < 624; i += WAVE) { L->np_mt[i] = g->np_mt[i]; L->py_mt[i] = g->py_mt[i]; }
    NpStream rs{L, g->np_pos, lane, g->has_gauss, g->gauss};
    int py_pos = g->py_pos, status = 0;
    const int words = (int)(c.instStride / 8);
    while (produced - cursor < (unsigned)c.Q) {
        status = gen_record(c, L, rs, py_pos, lane);
        if (status != PCBENV_OK) break;
        unsigned long long *dst = (unsigned long long *)(c.queue + ((size_t)(produced % (unsigned)c.Q) * c.B + e) * c.instStride);
        for (int i = lane; i < words; i += WAVE) dst[i] = L->rec[i];
        produced++;
    }
    for (int i = lane; i < 624; i += WAVE) { g->np_mt[i] = L->np_mt[i]; g->py_mt[i] = L->py_mt[i]; }
    if (lane == 0) { g->np_pos = rs.pos; g->py_pos = py_pos; g->has_gauss = rs.has_gauss; g->gauss = rs.gauss; g->status = status; }
    __threadfence();  // the records before the count
    if (lane == 0) c.produced[e] = produced;
}
