/*
 * oracle_batch.c -- batch driver over pcbenv_oracle.c for bench.py's
 * `cpu_baseline` leg and for whole-batch parity checks (TEST INFRASTRUCTURE).
 * One orc_env per environment, OpenMP over environments.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef struct orc_config orc_config;
typedef struct orc_env orc_env;
orc_env *orc_create(const orc_config *cfg);
void orc_destroy(orc_env *e);
int orc_reset(orc_env *e, int ncomp, const int *comp_h, const int *comp_w, int nnets, int npins,
              const int *rel_x, const int *rel_y, const int *net, const int *comp, const int *pin_id);
int orc_step(orc_env *e, int o, int x, int y, double *reward, int *done, double *info, int *has_info);

typedef struct orc_batch { int n; orc_env **envs; } orc_batch;

orc_batch *orc_batch_create(const orc_config *cfg, int n) {
    orc_batch *b = (orc_batch *)calloc(1, sizeof(orc_batch));
    b->n = n;
    b->envs = (orc_env **)calloc((size_t)n, sizeof(orc_env *));
    for (int i = 0; i < n; i++) b->envs[i] = orc_create(cfg);
    return b;
}
void orc_batch_destroy(orc_batch *b) {
    if (!b) return;
    for (int i = 0; i < b->n; i++) orc_destroy(b->envs[i]);
    free(b->envs); free(b);
}
orc_env *orc_batch_env(orc_batch *b, int i) { return b->envs[i]; }
int orc_max_threads(void) { return omp_get_max_threads(); }

/* Instances in the packed wire format of include/pcbenv.h (16-byte header,
 * 8-byte component records, 8-byte pin records); env i takes instance i. */
int orc_batch_reset_packed(orc_batch *b, const uint8_t *packed, int64_t stride, int max_comps,
                           const uint8_t *reset_mask, int threads) {
    int err = 0;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int i = 0; i < b->n; i++) {
        if (reset_mask && !reset_mask[i]) continue;
        const uint8_t *rec = packed + (size_t)i * (size_t)stride;
        const int32_t *hdr = (const int32_t *)rec;
        int nc = hdr[0], nn = hdr[1], np = hdr[2];
        int ch[256], cw[256], rx[1024], ry[1024], nt[1024], cp[1024], id[1024];
        const uint8_t *cr = rec + 16, *pr = rec + 16 + 8 * (size_t)max_comps;
        for (int c = 0; c < nc; c++) { ch[c] = cr[8 * c]; cw[c] = cr[8 * c + 1]; }
        for (int p = 0; p < np; p++) {
            rx[p] = pr[8 * p]; ry[p] = pr[8 * p + 1]; nt[p] = pr[8 * p + 2]; cp[p] = pr[8 * p + 3];
            id[p] = pr[8 * p + 4] | (pr[8 * p + 5] << 8);
        }
        if (orc_reset(b->envs[i], nc, ch, cw, nn, np, rx, ry, nt, cp, id) != 0) {
#pragma omp atomic write
            err = 1;
        }
    }
    return err ? -1 : 0;
}

/* actions: int32 [n,3] (o,x,y); square uses (x,y) = actions[:,1:3]. */
int orc_batch_step(orc_batch *b, const int32_t *actions, double *reward, uint8_t *done, double *info,
                   int threads) {
    int err = 0;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int i = 0; i < b->n; i++) {
        int d = 0, has = 0; double r = 0, inf[2] = {0, 0};
        if (orc_step(b->envs[i], actions[3 * i], actions[3 * i + 1], actions[3 * i + 2], &r, &d, inf, &has) != 0) {
#pragma omp atomic write
            err = 1;
        }
        reward[i] = r; done[i] = (uint8_t)d;
        if (info) { info[2 * i] = has ? inf[0] : 0.0; info[2 * i + 1] = has ? inf[1] : 0.0; }
    }
    return err ? -1 : 0;
}

/* Whole-batch observation check (tests): compares observation `key` of every environment with a host copy of the
 * device tensor -- uint8 cells (elem_bytes 1: value == the oracle's float64 0.0/1.0) or float64 features
 * (elem_bytes 8: identical bit patterns).  Returns the first environment that differs, -1 if none, -2 if the
 * element count per environment is not the oracle's. */
const double *orc_obs(const orc_env *e, int key, int64_t *count);
int64_t orc_batch_first_mismatch(orc_batch *b, int key, const void *dev_copy, int elem_bytes, int64_t per_env,
                                 int threads) {
    int64_t first = -1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int i = 0; i < b->n; i++) {
        int64_t cnt = 0;
        const double *ref = orc_obs(b->envs[i], key, &cnt);
        int64_t bad = -1;
        if (cnt != per_env) bad = -2;
        else if (elem_bytes == 1) {
            const uint8_t *g = (const uint8_t *)dev_copy + (size_t)i * (size_t)per_env;
            for (int64_t k = 0; k < cnt; k++) if ((double)g[k] != ref[k]) { bad = i; break; }
        } else {
            const double *g = (const double *)dev_copy + (size_t)i * (size_t)per_env;
            if (memcmp(g, ref, (size_t)cnt * 8) != 0) bad = i;
        }
        if (bad != -1) {
#pragma omp critical
            { if (first == -1 || bad == -2 || (first >= 0 && bad >= 0 && bad < first)) first = bad; }
        }
    }
    return first;
}
