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
// MT19937 is advanced lazily, one word per draw (the block regeneration of the textbook code, evaluated in order,
// reads exactly the same old / new neighbours), so a draw costs three loads and a store of the lane's own state.
// exp / log come from the device math library: like the host twin's libm they may differ from NumPy's SIMD
// kernels in the last bit of a probability, which can change a table only if a uniform variate lands within
// ~1 ulp of a threshold (~1e-15 per draw; see instance_gen.cpp).
#pragma once
#include "pcb_device.h"

struct GenState {            // per environment, in HBM
    unsigned np_mt[624];
    unsigned py_mt[624];
    int np_pos, py_pos;      // next word of the block to regenerate / hand out (0..623)
    int has_gauss, status;   // status: 0 ok, else the PCBENV_* code of the first record that could not be generated
    double gauss;
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

__device__ inline unsigned mt_next(unsigned *mt, int &pos) {
    const int i = pos, i1 = i + 1 == 624 ? 0 : i + 1, im = i + 397 >= 624 ? i + 397 - 624 : i + 397;
    const unsigned y = (mt[i] & 0x80000000u) | (mt[i1] & 0x7fffffffu);
    unsigned v = mt[im] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    mt[i] = v;
    pos = i1;
    v ^= (v >> 11);
    v ^= (v << 7) & 0x9d2c5680u;
    v ^= (v << 15) & 0xefc60000u;
    v ^= (v >> 18);
    return v;
}
__device__ inline void mt_init_genrand(unsigned *mt, unsigned s) {  // mt19937ar.c init_genrand
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

struct NpStream {  // NumPy legacy RandomState pieces on one lane's state
    GenState *g;
    int pos;
    __device__ unsigned u32() { return mt_next(g->np_mt, pos); }
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
        if (g->has_gauss) { const double t = g->gauss; g->has_gauss = 0; g->gauss = 0.0; return t; }
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * dbl() - 1.0;
            x2 = 2.0 * dbl() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        f = sqrt(-2.0 * log(r2) / r2);
        g->gauss = f * x1;
        g->has_gauss = 1;
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
    __device__ void multinomial(long long n, const double *p, int d, int *out, bool *ok) {
        double Sum = 1.0;
        long long dn = n;
        for (int j = 0; j < d; j++) out[j] = 0;
        for (int j = 0; j < d - 1; j++) {
            out[j] = (int)binomial(p[j] / Sum, dn, ok);
            dn -= out[j];
            if (dn <= 0) break;
            Sum -= p[j];
        }
        if (dn > 0) out[d - 1] = (int)dn;
    }
};

// np.sum of a contiguous float64 array (pairwise summation with 8 accumulators, block 128 -- n <= 128 here)
__device__ inline double np_sum_dev(const double *a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

// One record of stream g into rec (wire format of include/pcbenv.h).  Returns PCBENV_OK or the error code.
__device__ inline int gen_record(const GenParams &c, GenState *g, unsigned char *rec) {
    NpStream rs{g, g->np_pos};
    int py_pos = g->py_pos;
    unsigned long long *rec8 = (unsigned long long *)rec;
    for (int i = 0; i < (int)(c.instStride / 8); i++) rec8[i] = 0ull;
    int *hdr = (int *)rec;
    unsigned long long *crec = rec8 + 2, *prec = crec + c.C;
    int rc = PCBENV_OK;
    // steps 1-2
    const int ncomp = (int)rs.randint(c.min_comp, (long long)c.max_comp + 1);
    unsigned char hs[PCBENV_MAX_COMPONENTS], ws[PCBENV_MAX_COMPONENTS];
    short avail[PCBENV_MAX_COMPONENTS];
    unsigned char order[PCBENV_MAX_COMPONENTS];
    int total_area = 0;
    for (int i = 0; i < ncomp; i++) {
        hs[i] = (unsigned char)rs.randint(c.min_h, (long long)c.max_h + 1);
        ws[i] = (unsigned char)rs.randint(c.min_w, (long long)c.max_w + 1);
        avail[i] = (short)(hs[i] * ws[i]);
        order[i] = (unsigned char)i;
        total_area += avail[i];
        crec[i] = (unsigned long long)hs[i] | ((unsigned long long)ws[i] << 8);
    }
    hdr[0] = ncomp;
    if (c.kind != PCBENV_RECT) {
        // steps 3-4
        int nn = (int)rs.randint(c.min_nets, (long long)c.max_nets + 1);
        if (nn > total_area / 2) nn = total_area / 2;
        int total = (int)rs.randint((long long)c.min_ppn * nn, (long long)c.max_ppn * nn + 1);
        if (total > total_area) total = total_area;
        if (nn < 1 || total > c.P || c.min_ppn * nn > total) rc = PCBENV_EINVAL;  // the reference raises here
        if (rc == PCBENV_OK) {
            // step 5: softmax of normal samples (drawn even when unused)
            double pr[PCBENV_MAX_NETS], q[PCBENV_MAX_NETS];
            for (int i = 0; i < nn; i++) pr[i] = (1.0 / (double)nn) + (1.0 / (double)(c.net_distribution + 1)) * rs.legacy_gauss();
            for (int i = 0; i < nn; i++) pr[i] = exp(pr[i]);
            const double sez = np_sum_dev(pr, nn);
            for (int i = 0; i < nn; i++) pr[i] = pr[i] / sez;
            // steps 6-7: creation ids -> nets
            const int lo = c.min_ppn;
            unsigned char extra[PCBENV_MAX_NETS];
            for (int i = 0; i < nn; i++) extra[i] = 0;
            const int rem = total - lo * nn;
            bool ok = true;
            if (c.max_ppn > lo && rem > 0) {
                const int k = min(c.max_ppn - lo, rem);
                int sample[PCBENV_MAX_NETS];
                for (int t = 0; t < rem; t++) {
                    for (int i = 0; i < nn; i++) q[i] = pr[i] * (extra[i] < k ? 1.0 : 0.0);
                    const double sq = np_sum_dev(q, nn);
                    for (int i = 0; i < nn; i++) q[i] /= sq;
                    rs.multinomial(1, q, nn, sample, &ok);
                    for (int i = 0; i < nn; i++) extra[i] = (unsigned char)(extra[i] + sample[i]);
                }
            }
            // step 8
            int kcomp;
            if (c.kind == PCBENV_SPATIAL) kcomp = min((int)(((double)c.pin_spread / 10.0) * (double)ncomp) + 1, ncomp);
            else kcomp = min(max((int)(((double)(c.pin_spread + 1) / 10.0) * (double)ncomp), 1), ncomp);
            // step 9: net by net in net order; `order` = component ids by free space, descending, stable, carried over
            int q_idx = 0, id_cursor = lo * nn;  // first output pin of the net; first creation id of the net's extra pins
            for (int n = 0; n < nn && rc == PCBENV_OK; n++) {
                const int npins_net = lo + extra[n];
                int unassigned = npins_net;
                for (int i = 1; i < ncomp; i++) {  // stable insertion sort, descending free space
                    const unsigned char o = order[i];
                    const int a = avail[o];
                    int j = i - 1;
                    while (j >= 0 && avail[order[j]] < a) { order[j + 1] = order[j]; j--; }
                    order[j + 1] = o;
                }
                int k = kcomp - 1, space = 0;
                while (space < unassigned) {
                    k += 1;
                    space = 0;
                    for (int i = 0; i < k && i < ncomp; i++) space += avail[order[i]];
                    if (k > ncomp + 1) { rc = PCBENV_EINVAL; break; }
                }
                if (rc != PCBENV_OK) break;
                if (k > ncomp) k = ncomp;
                int pin_in_net = 0;
                while (unassigned > 0) {
                    int tot = 0;
                    for (int i = 0; i < k; i++) tot += avail[order[i]];
                    double probs[PCBENV_MAX_COMPONENTS];
                    int cnt[PCBENV_MAX_COMPONENTS];
                    for (int i = 0; i < k; i++) probs[i] = (double)avail[order[i]] / (double)tot;
                    rs.multinomial(unassigned, probs, k, cnt, &ok);
                    for (int i = 0; i < k; i++) {
                        const int cid = order[i];
                        int m = cnt[i];
                        if (avail[cid] < m) m = avail[cid];
                        avail[cid] = (short)(avail[cid] - m);
                        for (int j = 0; j < m; j++) {
                            // creation id of the pin_in_net-th pin of net n (spatial); index in this batch (pin env, quirk Q1)
                            const int id = c.kind == PCBENV_SPATIAL ? (pin_in_net < lo ? n * lo + pin_in_net : id_cursor + pin_in_net - lo) : j;
                            prec[q_idx + pin_in_net] = ((unsigned long long)n << 16) | ((unsigned long long)cid << 24) | ((unsigned long long)(id & 0xFFFF) << 32);
                            pin_in_net++;
                        }
                        unassigned -= m;
                    }
                    if (!ok) break;
                }
                q_idx += pin_in_net;
                id_cursor += extra[n];
                if (!ok) break;
            }
            if (!ok && rc == PCBENV_OK) rc = PCBENV_ELIMIT;
            // step 10: per component, random.choice over the remaining cells (row-major), pins in self.pins order
            if (rc == PCBENV_OK) {
                for (int cid = 0; cid < ncomp; cid++) {
                    unsigned char cells[PCBENV_MAX_PINS_PER_COMPONENT];
                    int ncell = hs[cid] * ws[cid];
                    for (int i = 0; i < ncell; i++) cells[i] = (unsigned char)i;
                    for (int j = 0; j < total; j++) {
                        unsigned long long w = prec[j];
                        if ((int)((w >> 24) & 0xFF) != cid) continue;
                        int kbits = 0;
                        for (int v = ncell; v; v >>= 1) kbits++;
                        unsigned r;
                        do { r = mt_next(g->py_mt, py_pos) >> (32 - kbits); } while ((int)r >= ncell);  // _randbelow_with_getrandbits
                        const int cell = cells[r];
                        for (int t = (int)r; t + 1 < ncell; t++) cells[t] = cells[t + 1];  // list.remove(value): cells are unique
                        ncell--;
                        w |= (unsigned long long)(cell / ws[cid]) | ((unsigned long long)(cell % ws[cid]) << 8);
                        prec[j] = w;
                    }
                }
                hdr[1] = nn;
                hdr[2] = total;
            }
        }
    }
    g->np_pos = rs.pos;
    g->py_pos = py_pos;
    return rc;
}

// Seeds the two generators of every environment like `np.random.seed(s); random.seed(s)` (s < 2^32).
__global__ __launch_bounds__(WAVE) void k_gen_seed(GenParams c, const unsigned *__restrict__ seeds) {
    const int e = blockIdx.x * WAVE + threadIdx.x;
    if (e >= c.B) return;
    GenState *g = c.gen + e;
    const unsigned s = seeds[e];
    mt_init_genrand(g->np_mt, s);
    mt_init_by_array(g->py_mt, &s, 1);
    g->np_pos = 0; g->py_pos = 0; g->has_gauss = 0; g->status = 0; g->gauss = 0.0;
    // the stream's first record is the environment's next reset: it goes where the queue cursor points
    c.produced[e] = load_agent(c.cursor_pub + e);
}

// Tops up every environment's queue: records produced[e] .. cursor + Q - 1 (slot = index % Q), never overwriting a
// record the environment has not consumed (the published cursor can only be behind the truth).  The records leave
// this XCD's L2 with the release fence at the end; the step kernels read them with agent-scope loads.
__global__ __launch_bounds__(WAVE) void k_gen_fill(GenParams c) {
    const int e = blockIdx.x * WAVE + threadIdx.x;
    if (e >= c.B) return;
    GenState *g = c.gen + e;
    const unsigned cursor = load_agent(c.cursor_pub + e);
    unsigned produced = c.produced[e];
    while (produced - cursor < (unsigned)c.Q && g->status == 0) {
        unsigned char *rec = c.queue + ((size_t)(produced % (unsigned)c.Q) * c.B + e) * c.instStride;
        const int rc = gen_record(c, g, rec);
        if (rc != PCBENV_OK) { g->status = rc; break; }
        produced++;
    }
    __threadfence();  // the records before the count
    c.produced[e] = produced;
}
