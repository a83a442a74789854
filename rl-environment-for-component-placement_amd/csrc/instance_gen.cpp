// instance_gen.cpp -- native host-side instance generator (part of libpcbenv.so).
//
// Restates the reference's `generate_instances()` (environment/dummy_env_rectangular_pin_spatial.py:931-1212,
// :1408-1443, sample_truncated_multinomial :250-287; ..._pin.py:1006-1265; ..._rectangular.py:253-273) together
// with the pieces of NumPy's legacy `RandomState` and CPython's `random.Random` it draws from, so that one
// stream `seed` yields the instances the reference yields after `np.random.seed(seed); random.seed(seed)`:
//
//   NumPy   MT19937 seeded by init_genrand(seed); randint = masked rejection on 32-bit outputs; normal =
//           legacy polar Box-Muller with its cached second value; multinomial = chain of legacy binomials
//           (inversion algorithm -- every call on this path has n*p <= 30)
//   CPython MT19937 seeded by init_by_array(words of |seed|); choice(seq) = seq[_randbelow(len)] with
//           getrandbits(k) = genrand_uint32() >> (32 - k)
//
// Draw order: SURVEY.md Appendix A.  The readable reference implementation of the same thing is
// pcbenv/instances.py:InstanceStream (which calls NumPy / random themselves); tests compare the two.
//
// One documented difference: the softmax uses libm's exp(), NumPy's array exp() is a SIMD kernel that differs
// from it in the last bit for a few percent of arguments.  The probabilities only steer binomial draws, so a
// table can differ only when a uniform variate falls within ~1 ulp of a threshold (probability ~1e-15 per draw);
// tests/test_instance_gen_native.py checks equality of the tables on thousands of streams.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "pcbenv.h"

namespace {

struct MT {
    uint32_t mt[624];
    int pos;
    void init_genrand(uint32_t s) {
        for (int i = 0; i < 624; i++) {
            mt[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
    }
    void init_by_array(const uint32_t *key, int len) {  // mt19937ar.c (CPython random_seed)
        init_genrand_ref(19650218u);
        int i = 1, j = 0;
        for (int k = (624 > len ? 624 : len); k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            i++; j++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
            if (j >= len) j = 0;
        }
        for (int k = 623; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            i++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
        pos = 624;
    }
    void init_genrand_ref(uint32_t s) {  // the reference init (mt[i] = f(mt[i-1]) + i)
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        pos = 624;
    }
    void gen() {
        const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, A = 0x9908b0dfu;
        int kk;
        for (kk = 0; kk < 624 - 397; kk++) {
            uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        for (; kk < 623; kk++) {
            uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        uint32_t y = (mt[623] & UPPER) | (mt[0] & LOWER);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        pos = 0;
    }
    uint32_t u32() {
        if (pos >= 624) gen();
        uint32_t y = mt[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double dbl() {  // NumPy next_double / CPython random(): 53 bits from two outputs
        uint32_t a = u32() >> 5, b = u32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

// NumPy legacy RandomState pieces
struct NpRandom {
    MT g;
    bool has_gauss = false;
    double gauss = 0.0;
    void seed(uint32_t s) { g.init_genrand(s); has_gauss = false; gauss = 0.0; }
    // randint(low, high): high exclusive
    int64_t randint(int64_t low, int64_t high) {
        uint64_t rng = (uint64_t)(high - 1 - low);
        if (rng == 0) return low;
        uint64_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        uint32_t val;
        do { val = g.u32() & (uint32_t)mask; } while (val > rng);
        return low + (int64_t)val;
    }
    double legacy_gauss() {
        if (has_gauss) { const double t = gauss; has_gauss = false; gauss = 0.0; return t; }
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * g.dbl() - 1.0;
            x2 = 2.0 * g.dbl() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        f = sqrt(-2.0 * log(r2) / r2);
        gauss = f * x1;
        has_gauss = true;
        return f * x2;
    }
    double normal(double loc, double scale) { return loc + scale * legacy_gauss(); }
    // legacy_random_binomial_inversion (every call on this path has n*min(p, 1-p) <= 30)
    int64_t binomial_inversion(int64_t n, double p) {
        const double q = 1.0 - p, qn = exp(n * log(q)), np = n * p;
        const double b = np + 10.0 * sqrt(np * q + 1);
        const int64_t bound = (int64_t)((double)n < b ? (double)n : b);
        int64_t X = 0;
        double px = qn, U = g.dbl();
        while (U > px) {
            X++;
            if (X > bound) { X = 0; px = qn; U = g.dbl(); }
            else { U -= px; px = ((n - X + 1) * p * px) / (X * q); }
        }
        return X;
    }
    // random_binomial as RandomState.multinomial reaches it (n == 0 or p == 0 return without a draw)
    int64_t binomial(double p, int64_t n, bool *ok) {
        if (n == 0 || p == 0.0) return 0;
        if (p <= 0.5) {
            if (p * n <= 30.0) return binomial_inversion(n, p);
        } else {
            const double q = 1.0 - p;
            if (q * n <= 30.0) return n - binomial_inversion(n, q);
        }
        *ok = false;  // BTPE would be needed: outside the sizes this library supports
        return 0;
    }
    void multinomial(int64_t n, const double *p, int d, int64_t *out, bool *ok) {
        double Sum = 1.0;
        int64_t dn = n;
        for (int j = 0; j < d; j++) out[j] = 0;
        for (int j = 0; j < d - 1; j++) {
            out[j] = binomial(p[j] / Sum, dn, ok);
            dn -= out[j];
            if (dn <= 0) break;
            Sum -= p[j];
        }
        if (dn > 0) out[d - 1] = dn;
    }
};

// np.sum of a contiguous float64 array (pairwise summation with 8 accumulators, block 128 -- n <= 128 here)
double np_sum(const double *a, int n) {
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

struct PyRandom {
    MT g;
    void seed(uint64_t s) {
        uint32_t key[2] = {(uint32_t)(s & 0xffffffffu), (uint32_t)(s >> 32)};
        g.init_by_array(key, key[1] ? 2 : 1);
    }
    int randbelow(int n) {  // _randbelow_with_getrandbits
        int k = 0;
        for (int v = n; v; v >>= 1) k++;
        uint32_t r;
        do { r = g.u32() >> (32 - k); } while ((int)r >= n);
        return (int)r;
    }
};

}  // namespace

struct pcbenv_instgen {
    pcbenv_config cfg;
    int C, P;
    int64_t stride;
    NpRandom np;
    PyRandom py;
};

static int gen_next(pcbenv_instgen *s, unsigned char *rec) {
    const pcbenv_config &c = s->cfg;
    NpRandom &rs = s->np;
    memset(rec, 0, (size_t)s->stride);
    int32_t *hdr = (int32_t *)rec;
    unsigned char *crec = rec + 16, *prec = rec + 16 + 8 * (size_t)s->C;
    // steps 1-2
    const int ncomp = (int)rs.randint(c.min_num_components, (int64_t)c.max_num_components + 1);
    int hs[PCBENV_MAX_COMPONENTS], ws[PCBENV_MAX_COMPONENTS], area[PCBENV_MAX_COMPONENTS];
    int total_area = 0;
    for (int i = 0; i < ncomp; i++) {
        hs[i] = (int)rs.randint(c.min_component_h, (int64_t)c.max_component_h + 1);
        ws[i] = (int)rs.randint(c.min_component_w, (int64_t)c.max_component_w + 1);
        area[i] = hs[i] * ws[i];
        total_area += area[i];
        crec[8 * i] = (unsigned char)hs[i];
        crec[8 * i + 1] = (unsigned char)ws[i];
    }
    hdr[0] = ncomp;
    if (c.kind == PCBENV_RECT) return PCBENV_OK;
    // steps 3-4
    int nn = (int)rs.randint(c.min_num_nets, (int64_t)c.max_num_nets + 1);
    if (nn > total_area / 2) nn = total_area / 2;
    int total = (int)rs.randint((int64_t)c.min_num_pins_per_net * nn, (int64_t)c.max_num_pins_per_net * nn + 1);
    if (total > total_area) total = total_area;
    if (nn < 1 || total > s->P || c.min_num_pins_per_net * nn > total) return PCBENV_EINVAL;  // the reference raises here
    // step 5: softmax of normal samples (drawn even when unused)
    double z[PCBENV_MAX_NETS], ez[PCBENV_MAX_NETS], p[PCBENV_MAX_NETS], q[PCBENV_MAX_NETS];
    for (int i = 0; i < nn; i++) z[i] = rs.normal(1.0 / nn, 1.0 / (c.net_distribution + 1));
    for (int i = 0; i < nn; i++) ez[i] = exp(z[i]);
    const double sez = np_sum(ez, nn);
    for (int i = 0; i < nn; i++) p[i] = ez[i] / sez;
    // steps 6-7: creation ids -> nets
    const int lo = c.min_num_pins_per_net;
    int extra[PCBENV_MAX_NETS] = {0};
    const int rem = total - lo * nn;
    bool ok = true;
    if (c.max_num_pins_per_net > lo && rem > 0) {
        const int k = std::min(c.max_num_pins_per_net - lo, rem);
        int64_t sample[PCBENV_MAX_NETS];
        for (int t = 0; t < rem; t++) {
            for (int i = 0; i < nn; i++) q[i] = p[i] * (extra[i] < k ? 1.0 : 0.0);
            const double sq = np_sum(q, nn);
            for (int i = 0; i < nn; i++) q[i] /= sq;
            rs.multinomial(1, q, nn, sample, &ok);
            for (int i = 0; i < nn; i++) extra[i] += (int)sample[i];
        }
    }
    // net -> list of creation ids (in net-list order)
    std::vector<int> net_ids[PCBENV_MAX_NETS];
    int cursor = lo * nn;
    for (int n = 0; n < nn; n++) {
        for (int j = 0; j < lo; j++) net_ids[n].push_back(n * lo + j);
        for (int j = 0; j < extra[n]; j++) net_ids[n].push_back(cursor + j);
        cursor += extra[n];
    }
    // step 8
    int kcomp;
    if (c.kind == PCBENV_SPATIAL) kcomp = std::min((int)((c.pin_spread / 10.0) * ncomp) + 1, ncomp);
    else kcomp = std::min(std::max((int)(((c.pin_spread + 1) / 10.0) * ncomp), 1), ncomp);
    int order[PCBENV_MAX_COMPONENTS], avail[PCBENV_MAX_COMPONENTS];
    for (int i = 0; i < ncomp; i++) { order[i] = i; avail[i] = area[i]; }
    // step 9
    int q_idx = 0;  // index into the output pin list (net-major)
    for (int n = 0; n < nn; n++) {
        int unassigned = (int)net_ids[n].size();
        std::stable_sort(order, order + ncomp, [&](int a, int b) { return avail[a] > avail[b]; });
        int k = kcomp - 1, space = 0;
        while (space < unassigned) {
            k += 1;
            space = 0;
            for (int i = 0; i < k && i < ncomp; i++) space += avail[order[i]];
            if (k > ncomp + 1) return PCBENV_EINVAL;
        }
        if (k > ncomp) k = ncomp;
        int pin_in_net = 0;
        while (unassigned > 0) {
            int tot = 0;
            for (int i = 0; i < k; i++) tot += avail[order[i]];
            double probs[PCBENV_MAX_COMPONENTS];
            int64_t cnt[PCBENV_MAX_COMPONENTS];
            for (int i = 0; i < k; i++) probs[i] = (double)avail[order[i]] / (double)tot;
            rs.multinomial(unassigned, probs, k, cnt, &ok);
            for (int i = 0; i < k; i++) {
                const int cid = order[i];
                int m = (int)cnt[i];
                if (avail[cid] < m) m = avail[cid];
                avail[cid] -= m;
                for (int j = 0; j < m; j++) {
                    unsigned char *pr = prec + 8 * (size_t)(q_idx + pin_in_net);
                    const int id = c.kind == PCBENV_SPATIAL ? net_ids[n][pin_in_net] : j;  // quirk Q1 (pin env)
                    pr[2] = (unsigned char)n; pr[3] = (unsigned char)cid;
                    pr[4] = (unsigned char)(id & 0xFF); pr[5] = (unsigned char)(id >> 8);
                    pin_in_net++;
                }
                unassigned -= m;
            }
        }
        q_idx += pin_in_net;
    }
    if (!ok) return PCBENV_ELIMIT;
    // step 10: per component, random.choice over the remaining cells (row-major), pins in self.pins order
    for (int cid = 0; cid < ncomp; cid++) {
        int cells[PCBENV_MAX_PINS_PER_COMPONENT * 4];
        int ncell = area[cid];
        if (ncell > (int)(sizeof(cells) / sizeof(int))) return PCBENV_ELIMIT;
        for (int i = 0; i < ncell; i++) cells[i] = i;
        for (int j = 0; j < total; j++) {
            unsigned char *pr = prec + 8 * (size_t)j;
            if (pr[3] != cid) continue;
            const int pick = s->py.randbelow(ncell);
            const int cell = cells[pick];
            for (int t = pick; t + 1 < ncell; t++) cells[t] = cells[t + 1];  // list.remove(value): cells are unique
            ncell--;
            pr[0] = (unsigned char)(cell / ws[cid]);
            pr[1] = (unsigned char)(cell % ws[cid]);
        }
    }
    hdr[1] = nn;
    hdr[2] = total;
    return PCBENV_OK;
}

extern "C" int pcbenv_instgen_create(const pcbenv_config *cfg, uint64_t seed, pcbenv_instgen **out) {
    if (out) *out = 0;
    if (!cfg || !out || cfg->kind == PCBENV_SQUARE) return PCBENV_EINVAL;
    if (seed > 0xffffffffull) return PCBENV_EINVAL;  // RandomState(seed) accepts 32-bit seeds only
    if (cfg->max_num_components > PCBENV_MAX_COMPONENTS || cfg->max_num_nets > PCBENV_MAX_NETS) return PCBENV_ELIMIT;
    pcbenv_instgen *s = new pcbenv_instgen();
    s->cfg = *cfg;
    if (cfg->kind == PCBENV_PIN || cfg->kind == PCBENV_SPATIAL) {
        s->cfg.net_distribution = std::max(0, std::min(9, cfg->net_distribution));
        s->cfg.pin_spread = std::max(0, std::min(9, cfg->pin_spread));
    }
    s->C = cfg->max_num_components;
    s->P = pcbenv_max_total_pins(cfg);
    s->stride = pcbenv_instance_stride(cfg);
    s->np.seed((uint32_t)seed);
    s->py.seed(seed);
    *out = s;
    return PCBENV_OK;
}

extern "C" void pcbenv_instgen_destroy(pcbenv_instgen *s) { delete s; }

extern "C" int pcbenv_instgen_next(pcbenv_instgen *s, void *record_out) {
    if (!s || !record_out) return PCBENV_EINVAL;
    return gen_next(s, (unsigned char *)record_out);
}

extern "C" int pcbenv_instgen_next_batch(pcbenv_instgen *const *streams, int32_t n, void *records_out, int32_t threads) {
    if (!streams || !records_out || n < 0) return PCBENV_EINVAL;
    if (n == 0) return PCBENV_OK;
    const int64_t stride = streams[0]->stride;
    int nt = std::max(1, std::min(threads, n));
    std::vector<int> rc((size_t)nt, PCBENV_OK);
    auto work = [&](int t) {
        for (int i = t; i < n; i += nt) {
            int r = gen_next(streams[i], (unsigned char *)records_out + (size_t)i * (size_t)stride);
            if (r != PCBENV_OK) rc[(size_t)t] = r;
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++) pool.emplace_back(work, t);
        for (auto &th : pool) th.join();
    }
    for (int r : rc) if (r != PCBENV_OK) return r;
    return PCBENV_OK;
}
