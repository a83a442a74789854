// pcbenv_kernels.hip -- CDNA4 (gfx950) kernels + C ABI of libpcbenv.so.
//
// One environment per workgroup: one wavefront (64 lanes) up to 64x64 cells, four wavefronts for the 128x128
// spatial configuration (template parameter NW).  Kernels: k_reset, k_step (transition + legal mask +
// observations + terminal routing reward + optional in-launch reset and action sampling), k_sample,
// k_cursor_range.  The occupancy grid lives bit-packed (one row = WW 64-bit words) in a compact
// per-environment state block in HBM that is staged through LDS; the legal
// placement mask is OR-folds of row words (horizontal: shifts; vertical: LDS
// neighbours); the observation tensors the policy consumes (uint8 cells) are a
// pure coalesced 16-byte-per-lane write stream, which is what bounds the kernel.
//
// Reference behaviour restated (file:line in the reference repo; S = environment/
// dummy_env_rectangular_pin_spatial.py, P = ..._pin.py, R = ..._rectangular.py,
// Q = dummy_env_square.py): see the comment on each device function.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (one IEEE operation per
// written operator; the only fused multiply-add is the explicit __fma_rn in norm2).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "pcbenv.h"
#define PCB_HOST_TU
#include "pcb_kernels.h"  // k_sample, k_cursor_range (the per-kind kernels are instantiated in pcb_kind_*.hip)
#include "pcb_launch.h"
#include "pcb_geninst.h"

// ==============================================================================================
// host side: the C ABI (include/pcbenv.h)
// ==============================================================================================
struct pcbenv {
    pcbenv_config cfg;
    int device;
    DevParams dp;
    bool bound;
    int threads;  // workgroup size (threads per environment)
    long long cell_bytes_per_env, stream_threshold;  // store policy (see STORE16): cell-tensor bytes one transition writes per environment
    unsigned *scratch;  // 16 bytes of device memory for small read-backs
    unsigned long long loaded_slots[4];  // bit s = slot s loaded for all environments at least once
    // on-device instance generator (pcbenv_instgen_device_enable): side stream + the bookkeeping that guarantees a
    // record is complete before any launch can consume it (see gen_before_launch)
    bool gen_on, gen_outstanding;
    int gen_grid;  // workgroups of a refill launch (GEN_MAX_GRID; PCBENV_GEN_GRID overrides, for experiments)
    GenParams gp;
    unsigned *cursor_snap;  // the cursors as of a fill's snapshot: what k_gen_fill reads (see gen_start_fill)
    hipStream_t gen_stream;
    hipEvent_t ev_snap, ev_fill;
    long long since_waited, since_outstanding;
    int gen_lanes;      // PCBENV_OPT_GEN_LANES: 0 = the narrowest group the configuration allows
    // terminal list (Team<>::run_env): launch counter, list entries that get helper teams per launch (0 = none)
    unsigned seq;
    int term_wgs;
    unsigned *term_seen_host;     // mapped host memory the step kernel reports its list length to (DevParams::term_seen)
    unsigned char *state_buf[2];  // double-buffered state blocks: dp.state is the current one, a step launch writes the other
    int state_cur;
    char err[256];
};

static char g_err[256] = "";
static int fail(pcbenv *env, int code, const char *fmt, const char *detail = "") {
    char *dst = env ? env->err : g_err;
    snprintf(dst, 256, fmt, detail);
    if (env) snprintf(g_err, 256, "%s", dst);
    return code;
}
#define HIP_TRY(env, call)                                                          \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) return fail(env, PCBENV_EHIP, #call ": %s", hipGetErrorString(e_)); \
    } while (0)

// Every entry point works on the handle's device and leaves the caller's current device as it found it.
struct DeviceGuard {
    int prev = -1; bool ok = true, changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) { ok = hipSetDevice(dev) == hipSuccess; changed = ok && prev >= 0; }
    }
    ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
};
#define DEVICE_GUARD(env) DeviceGuard guard_((env)->device); if (!guard_.ok) return fail(env, PCBENV_EHIP, "hipSetDevice failed")

#define PCBENV_TERM_CAP_MAX 4096  // entries per ring of the terminal list = the most terminal workgroups of a launch
static int align16(long long v) { return (int)((v + 15) & ~15ll); }

extern "C" int pcbenv_abi_version(void) { return PCBENV_ABI_VERSION; }

extern "C" const char *pcbenv_last_error(const pcbenv *env) { return env ? env->err : g_err; }

static bool is_pin_kind(int k) { return k == PCBENV_PIN || k == PCBENV_SPATIAL; }

extern "C" int32_t pcbenv_max_total_pins(const pcbenv_config *c) {
    if (!c || !is_pin_kind(c->kind)) return 0;
    long long a = (long long)c->max_num_pins_per_net * c->max_num_nets;
    long long b = (long long)c->max_num_components * c->max_component_h * c->max_component_w;
    return (int32_t)(a < b ? a : b);
}
extern "C" int64_t pcbenv_instance_stride(const pcbenv_config *c) {
    if (!c || c->kind == PCBENV_SQUARE) return 0;
    return 16 + 8ll * (c->max_num_components + pcbenv_max_total_pins(c));
}

// The reference constructors' checks (quirk Q6), then the HIP path's limits.
static int validate(const pcbenv_config *c) {
    if (c->kind < PCBENV_SQUARE || c->kind > PCBENV_SPATIAL) return fail(0, PCBENV_EINVAL, "unknown environment kind");
    if (c->height < 0 || c->width < 0) return fail(0, PCBENV_EINVAL, "Grid size must not be negative.");
    if (c->num_envs < 1) return fail(0, PCBENV_EINVAL, "num_envs must be at least 1");
    if (c->kind == PCBENV_SQUARE) {
        if (c->component_n > c->height || c->component_n > c->width)
            return fail(0, PCBENV_EINVAL, "Component size must not exceed the grid size.");
        if (c->component_n < 1) return fail(0, PCBENV_ELIMIT, "component_n must be at least 1");
    } else {
        bool too_big = c->kind == PCBENV_PIN ? (c->max_component_w > c->width || c->max_component_h > c->height)
                                             : (c->max_component_w > c->height || c->max_component_h > c->width);
        if (too_big) return fail(0, PCBENV_EINVAL, "Component size must not exceed the grid size.");
        if (c->min_component_w < 1 || c->min_component_h < 1) return fail(0, PCBENV_EINVAL, "Component size must be at least 1.");
        if (c->max_num_components < 1 || c->max_num_components > c->height * c->width)
            return fail(0, PCBENV_EINVAL, "Number of components must be in [1, grid area].");
    }
    if (c->kind == PCBENV_PIN) {
        if (c->min_num_pins_per_net > c->max_num_pins_per_net) return fail(0, PCBENV_EINVAL, "min_num_pins_per_net must not exceed max_num_pins_per_net.");
        if (c->min_num_pins_per_net < 2) return fail(0, PCBENV_EINVAL, "min_num_pins_per_net must be at least 2.");
        if (c->min_num_pins_per_net * c->min_num_nets > c->min_component_w * c->min_component_h * c->min_num_components)
            return fail(0, PCBENV_EINVAL, "min_num_pins_per_net * min_num_nets exceeds the minimum total component area.");
        if (c->reward_beam_width < 1) return fail(0, PCBENV_EINVAL, "Beam width must be a positive integer.");
        if (c->reward_type < 0 || c->reward_type > 2) return fail(0, PCBENV_EINVAL, "Reward type must be 'beam', 'centroid' or 'both'.");
    }
    if (c->kind == PCBENV_SPATIAL) {
        if (c->reward_type < 0 || c->reward_type > 2) return fail(0, PCBENV_EINVAL, "Reward type must be 'beam', 'centroid' or 'both'.");
        if (c->reward_beam_width < 2 || c->reward_beam_width > c->max_num_pins_per_net)
            return fail(0, PCBENV_EINVAL, "Beam width must be an integer in [2, max_num_pins_per_net].");
        if (c->weight_wirelength < 0) return fail(0, PCBENV_EINVAL, "weight_wirelength must not be negative.");
    }
    // limits of this implementation
    if (c->height < 1 || c->width < 1 || c->height > PCBENV_MAX_SIDE || c->width > PCBENV_MAX_SIDE)
        return fail(0, PCBENV_ELIMIT, "grid side must be in [1, 128]");
    if (c->queue_depth < 1 || c->queue_depth > 256) return fail(0, PCBENV_ELIMIT, "queue_depth must be in [1, 256]");
    if (c->kind != PCBENV_SQUARE) {
        int side = c->max_component_h > c->max_component_w ? c->max_component_h : c->max_component_w;
        int shorter = c->height < c->width ? c->height : c->width;
        if (side > shorter) return fail(0, PCBENV_ELIMIT, "a component side exceeds the shorter grid side (the reference raises inside convolve2d)");
        if (c->max_num_components > PCBENV_MAX_COMPONENTS) return fail(0, PCBENV_ELIMIT, "too many components");
        if (c->min_num_components < 1 || c->min_num_components > c->max_num_components) return fail(0, PCBENV_ELIMIT, "min_num_components must be in [1, max_num_components]");
        if (c->min_component_h > c->max_component_h || c->min_component_w > c->max_component_w) return fail(0, PCBENV_ELIMIT, "min component size exceeds max");
    }
    if (is_pin_kind(c->kind)) {
        if (pcbenv_max_total_pins(c) > PCBENV_MAX_PINS || c->max_num_nets > PCBENV_MAX_NETS) return fail(0, PCBENV_ELIMIT, "too many pins or nets");
        if (c->max_num_pins_per_net > PCBENV_MAX_PINS_PER_NET) return fail(0, PCBENV_ELIMIT, "too many pins per net");
        if (c->max_component_h * c->max_component_w > PCBENV_MAX_PINS_PER_COMPONENT) return fail(0, PCBENV_ELIMIT, "too many pins per component");
        if (c->min_num_pins_per_net < 1 || c->min_num_nets < 1 || c->min_num_nets > c->max_num_nets) return fail(0, PCBENV_ELIMIT, "nets / pins per net must be at least 1");
        if (c->reward_type != PCBENV_REWARD_CENTROID && c->reward_beam_width > PCBENV_MAX_BEAM_WIDTH) return fail(0, PCBENV_ELIMIT, "beam width above 4");
    }
    return PCBENV_OK;
}

static double mean2(int a, int b) { return (double)(a + b) / 2.0; }

extern "C" int pcbenv_create(const pcbenv_config *cfg, int device, pcbenv **out) {
    if (out) *out = 0;
    if (!cfg || !out) return fail(0, PCBENV_EINVAL, "null argument");
    int rc = validate(cfg);
    if (rc != PCBENV_OK) return rc;
    pcbenv *env = new pcbenv();
    memset(env, 0, sizeof(*env));
    env->cfg = *cfg;
    env->device = device;
    if (is_pin_kind(cfg->kind)) {  // P:467-468 / S:450-451: clipped after validation
        env->cfg.net_distribution = cfg->net_distribution < 0 ? 0 : cfg->net_distribution > 9 ? 9 : cfg->net_distribution;
        env->cfg.pin_spread = cfg->pin_spread < 0 ? 0 : cfg->pin_spread > 9 ? 9 : cfg->pin_spread;
    }
    DevParams &d = env->dp;
    const pcbenv_config &c = env->cfg;
    d.kind = c.kind; d.H = c.height; d.W = c.width; d.WW = (c.width + 63) / 64;
    d.O = c.kind == PCBENV_SQUARE ? 1 : c.kind == PCBENV_RECT ? 2 : 4;
    d.C = c.kind == PCBENV_SQUARE ? 0 : c.max_num_components;
    d.P = pcbenv_max_total_pins(&c);
    d.N = is_pin_kind(c.kind) ? c.max_num_nets : 0; d.K = d.N + 1;
    d.mh = c.max_component_h; d.mw = c.max_component_w; d.mp = d.mh * d.mw;
    d.F = c.kind == PCBENV_SPATIAL ? 5 + d.mp : 5;
    d.pinRows = c.kind == PCBENV_PIN ? d.C * d.mp : c.kind == PCBENV_SPATIAL ? d.C * d.mp + 1 : 0;
    d.catW = c.kind == PCBENV_SPATIAL ? 2 : 1;
    d.B = c.num_envs; d.Q = c.queue_depth;
    d.reward_type = c.reward_type; d.beam_width = c.reward_beam_width; d.component_n = c.component_n;
    d.flags = c.flags;
    {   // streaming stores when one launch writes well beyond the 256 MiB Infinity Cache (see STORE16)
        const long long cells = (long long)c.height * c.width;
        const long long per_env = (c.flags & PCBENV_FLAG_INCREMENTAL_OBS) ? cells * d.O : cells * (1 + d.O + (c.kind == PCBENV_SPATIAL ? d.K : 0));
        env->cell_bytes_per_env = per_env; env->stream_threshold = 256ll << 20;  // PCBENV_OPT_STREAM_THRESHOLD_BYTES
        d.stream_stores = per_env * c.num_envs > env->stream_threshold;
    }
    d.w_wl = c.weight_wirelength; d.w_int = c.weight_num_intersections;
    d.area = (double)(c.height * c.width);
    if (is_pin_kind(c.kind)) {  // a15 (S:724-791, P:757-830) and the normalisers of find_reward (S:839-850)
        double dist = sqrt(fma((double)c.width, (double)c.width, (double)c.height * (double)c.height));
        double total = 0.5 * dist * (double)(c.max_num_nets * c.max_num_pins_per_net);
        d.max_wl = c.kind == PCBENV_SPATIAL ? total / (double)(c.height + c.width) : total;
        double mi = 0.5 * (double)(c.max_num_pins_per_net * c.max_num_pins_per_net) * (double)c.max_num_nets * (double)(c.max_num_nets - 1);
        d.max_int = c.kind == PCBENV_PIN ? (double)(long long)mi : mi;
        d.wl_norm = (double)(c.height + c.width);
        double a = mean2(c.min_component_h, c.max_component_h) * mean2(c.min_component_w, c.max_component_w) * mean2(c.min_num_components, c.max_num_components);
        double b = mean2(c.min_num_pins_per_net, c.max_num_pins_per_net) * mean2(c.min_num_nets, c.max_num_nets);
        d.int_norm = a < b ? a : b;
    }
    // threads per environment: one wave up to 64x64 cells of output per plane, four waves above
    env->threads = c.threads_per_env == 64 || c.threads_per_env == 256 ? c.threads_per_env
                   : ((long long)c.height * c.width * (c.kind == PCBENV_SPATIAL ? d.K + 5 : 5) > 64 * 1024 ? 256 : 64);
    // state block: header | occ | vm | comps | pins
    d.offOcc = HDR_BYTES;
    d.offVm = d.offOcc + d.H * d.WW * 8;
    d.offComps = d.offVm + 2 * d.H * d.WW * 8;
    d.offPins = d.offComps + 8 * d.C;
    d.offRank = d.offPins + 8 * d.P;  // rank of each pin inside its component (spatial env)
    d.stateStride = align16((long long)d.offRank + (c.kind == PCBENV_SPATIAL ? d.P : 0));
    d.num_slots = 1; d.slot = 0;
    d.instStride = align16(pcbenv_instance_stride(&c));
    // LDS scratch behind the state mirror: the folded rows of window_mask where it stages them in LDS (not the one-row-
    // per-lane cross-lane fold of one-wavefront teams up to 64 rows), doubling as the pin env's row-membership bit map
    d.ldsHf = (int)d.stateStride;
    {
#ifdef PCBENV_FOLD_LDS
        const bool fold_in_lds = true;
#else
        const bool fold_in_lds = !(d.WW == 1 && env->threads == 64 && d.H <= 64);
#endif
        const int fold_words = fold_in_lds ? d.H * d.WW : 0, member_words = c.kind == PCBENV_PIN ? (d.C * d.mp + 63) / 64 : 0;
        d.ldsHfWords = fold_words > member_words ? fold_words : member_words;
    }
    // the class map (pin_grid emission) and the route segments (terminal reward) are never live together
    d.ldsCls = align16(d.ldsHf + d.ldsHfWords * 8);
    d.ldsSeg = d.ldsCls;
    {
        const int beam = (is_pin_kind(c.kind) && c.reward_type != PCBENV_REWARD_CENTROID) ? BEAM_LDS_BYTES(c.max_num_nets, c.reward_beam_width) : 0;
        // class map of emit_pin_grid; at a reset the same zone holds the pin-id and net-mask tables (2 + 4 bytes per component cell)
        int cls = c.kind == PCBENV_SPATIAL ? (d.H * d.W > d.C * d.mp * 6 + 4 ? d.H * d.W : d.C * d.mp * 6 + 4) : 0, seg = is_pin_kind(c.kind) ? SEG_LDS_BYTES(d.P, d.N, env->threads / 64, beam) : 0;
        d.ldsBytes = align16(d.ldsCls + (cls > seg ? cls : seg));
    }
#ifdef PCBENV_EXPERIMENTS
    { const char *ev = getenv("PCBENV_LDS_MIN"); if (ev && atoi(ev) > d.ldsBytes) d.ldsBytes = align16(atoi(ev)); }  // occupancy experiments
#endif
    // Terminal list: on for one-wavefront teams with instances (Team<>::run_env); B / 8 entries cover twice the
    // 1 / max_num_components of the batch that ends an episode per launch when the phases are spread evenly over a
    // 16-component episode (PCBENV_OPT_TERMINAL_TEAMS changes or disables it).
    env->seq = 0;
    env->term_wgs = 0;
    if (is_pin_kind(c.kind)) {
        int wgs = (c.num_envs / 8 + (int)TERM_SHARDS - 1) & ~((int)TERM_SHARDS - 1);
        env->term_wgs = wgs < (int)TERM_SHARDS ? (int)TERM_SHARDS : wgs > PCBENV_TERM_CAP_MAX ? PCBENV_TERM_CAP_MAX : wgs;
    }
    d.term_cap = env->term_wgs;  // one list entry per set of helper teams
    d.term_hpe = REWARD_PARTS + ((c.flags & PCBENV_FLAG_AUTO_RESET) ? 1 : 0);
    DeviceGuard guard_(device);
    if (!guard_.ok) { int r = fail(0, PCBENV_EHIP, "hipSetDevice failed (no such device?)"); delete env; return r; }
    size_t sbytes = (size_t)d.stateStride * d.B, qbytes = (size_t)d.instStride * d.B * d.Q;
    if (hipMalloc((void **)&env->state_buf[0], sbytes) != hipSuccess || hipMalloc((void **)&env->state_buf[1], sbytes) != hipSuccess ||
        hipMalloc((void **)&d.queue, qbytes ? qbytes : 16) != hipSuccess ||
        hipMalloc((void **)&d.cursor_pub, 4 * (size_t)d.B) != hipSuccess ||
        hipMalloc((void **)&d.term_list, 4 * (size_t)4 * PCBENV_TERM_CAP_MAX) != hipSuccess || hipMalloc((void **)&d.term_cnt, 4 * TERM_SHARDS * TERM_CNT_STRIDE * 4 + 32) != hipSuccess || hipMalloc((void **)&d.term_arrive, 8 * (size_t)PCBENV_TERM_CAP_MAX) != hipSuccess ||
        hipHostMalloc((void **)&env->term_seen_host, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&d.term_seen, env->term_seen_host, 0) != hipSuccess) {
        int r = fail(0, PCBENV_EHIP, "hipMalloc failed");
        pcbenv_destroy(env);
        return r;
    }
#ifdef PCBENV_STAMPS
    { const char *ev = getenv("PCBENV_STAMPS"); if (ev && ev[0] == '1') { hipMalloc((void **)&d.dbg, (size_t)(d.B + PCBENV_TERM_CAP_MAX * (REWARD_PARTS + 1)) * 32 * 8); hipMemset(d.dbg, 0, (size_t)(d.B + PCBENV_TERM_CAP_MAX * (REWARD_PARTS + 1)) * 32 * 8); } }
#endif
    hipMemset(env->state_buf[0], 0, sbytes);
    hipMemset(env->state_buf[1], 0, sbytes);
    env->state_cur = 0;
    d.state = d.state_out = env->state_buf[0];
    hipMemset(d.queue, 0, qbytes ? qbytes : 16);
    hipMemset(d.cursor_pub, 0, 4 * (size_t)d.B);
    hipMemset(d.term_cnt, 0, 4 * TERM_SHARDS * TERM_CNT_STRIDE * 4 + 32);
    hipMemset(d.term_list, 0, 4 * (size_t)4 * PCBENV_TERM_CAP_MAX);
    hipMemset(d.term_arrive, 0, 8 * (size_t)PCBENV_TERM_CAP_MAX);
    *env->term_seen_host = 0u;
    hipDeviceSynchronize();
    *out = env;
    return PCBENV_OK;
}

extern "C" void pcbenv_destroy(pcbenv *env) {
    if (!env) return;
    DeviceGuard guard_(env->device);
    if (env->gen_on) {
        hipStreamSynchronize(env->gen_stream);
        hipEventDestroy(env->ev_snap); hipEventDestroy(env->ev_fill);
        hipStreamDestroy(env->gen_stream);
        if (env->gp.gen) hipFree(env->gp.gen);
        if (env->gp.produced) hipFree(env->gp.produced);
        if (env->cursor_snap) hipFree(env->cursor_snap);
    }
    if (env->state_buf[0]) hipFree(env->state_buf[0]);
    if (env->state_buf[1]) hipFree(env->state_buf[1]);
    if (env->dp.queue) hipFree(env->dp.queue);
    if (env->dp.cursor_pub) hipFree(env->dp.cursor_pub);
    if (env->dp.term_list) hipFree(env->dp.term_list);
    if (env->dp.term_cnt) hipFree(env->dp.term_cnt);
    if (env->dp.term_arrive) hipFree(env->dp.term_arrive);
    if (env->dp.feat_cache) hipFree(env->dp.feat_cache);
    if (env->dp.feat_cache_tag) hipFree(env->dp.feat_cache_tag);
    if (env->term_seen_host) hipHostFree(env->term_seen_host);
    if (env->scratch) hipFree(env->scratch);
    delete env;
}

extern "C" int pcbenv_set_option(pcbenv *env, int32_t option, int64_t value) {
    if (!env) return fail(0, PCBENV_EINVAL, "null handle");
    switch (option) {
    case PCBENV_OPT_STREAM_THRESHOLD_BYTES:
        if (value < 0) return fail(env, PCBENV_EINVAL, "threshold must not be negative");
        env->stream_threshold = value;
        env->dp.stream_stores = env->cell_bytes_per_env * env->dp.B > env->stream_threshold;
        return PCBENV_OK;
    case PCBENV_OPT_TERMINAL_TEAMS:
        if (value < 0 || value > PCBENV_TERM_CAP_MAX) return fail(env, PCBENV_EINVAL, "terminal-list entries must be in [0, 4096]");
        if (value > 0 && !is_pin_kind(env->cfg.kind))
            return fail(env, PCBENV_EINVAL, "reward helpers need an environment kind with a routing reward");
        value = (value + TERM_SHARDS - 1) & ~(long long)(TERM_SHARDS - 1);
        {   // The lists built so far were laid out for the old capacity: drop them (counters to zero once everything enqueued
            // has run; no mark matches the next launch's number).  A rare call: it may synchronise.
            DEVICE_GUARD(env);
            HIP_TRY(env, hipDeviceSynchronize());
            HIP_TRY(env, hipMemset(env->dp.term_cnt, 0, 4 * TERM_SHARDS * TERM_CNT_STRIDE * 4 + 32));
            *env->term_seen_host = 0u;
        }
        env->term_wgs = (int)value;
        env->dp.term_cap = (int)value;
        env->seq += 2;
        return PCBENV_OK;
    case PCBENV_OPT_GEN_GRID:
        if (value < 1) return fail(env, PCBENV_EINVAL, "generator grid must be at least 1");
        env->gen_grid = (int)value;
        return PCBENV_OK;
    case PCBENV_OPT_GEN_LANES:
        if (env->gen_on) return fail(env, PCBENV_ESTATE, "set the generator's group width before enabling it");
        if (value != 0 && value != 16 && value != 32 && value != 64) return fail(env, PCBENV_EINVAL, "generator lanes per environment: 0, 16, 32 or 64");
        env->gen_lanes = (int)value;
        return PCBENV_OK;
    }
    return fail(env, PCBENV_EINVAL, "unknown option");
}

extern "C" int pcbenv_bind_buffers_slots(pcbenv *env, const pcbenv_buffers *b, int32_t num_slots);
extern "C" int pcbenv_bind_buffers(pcbenv *env, const pcbenv_buffers *b) { return pcbenv_bind_buffers_slots(env, b, 1); }

extern "C" int pcbenv_select_slot(pcbenv *env, int32_t slot) {
    if (!env) return fail(0, PCBENV_EINVAL, "null handle");
    if (slot < 0 || slot >= env->dp.num_slots) return fail(env, PCBENV_EINVAL, "slot out of range");
    env->dp.slot = slot;
    return PCBENV_OK;
}

extern "C" int pcbenv_bind_buffers_slots(pcbenv *env, const pcbenv_buffers *b, int32_t num_slots) {
    if (!env || !b) return fail(env, PCBENV_EINVAL, "null argument");
    if (num_slots < 1 || num_slots > 4096) return fail(env, PCBENV_EINVAL, "num_slots must be in [1, 4096]");
    if (num_slots > 1 && (env->cfg.flags & PCBENV_FLAG_INCREMENTAL_OBS))
        return fail(env, PCBENV_EINVAL, "PCBENV_FLAG_INCREMENTAL_OBS needs the in-place layout (num_slots = 1)");
    if ((long long)num_slots * env->dp.B > 0x7fffffffll / 8) return fail(env, PCBENV_ELIMIT, "num_slots * num_envs too large");
    env->dp.num_slots = num_slots; env->dp.slot = 0;
    if (!b->reward || !b->done) return fail(env, PCBENV_EINVAL, "reward and done buffers are required");
    env->dp.buf = *b;
    int k = env->cfg.kind;
    if (k != PCBENV_SPATIAL) { env->dp.buf.pin_grid = 0; env->dp.buf.component_grid = 0; }
    if (!is_pin_kind(k)) { env->dp.buf.all_pins_num_feature = 0; env->dp.buf.all_pins_cat_feature = 0; env->dp.buf.info = 0; }
    if (k != PCBENV_RECT) env->dp.buf.component_mask = 0;
    if (k == PCBENV_SQUARE) { env->dp.buf.all_components_feature = 0; env->dp.buf.placement_mask = 0; }
    if (k == PCBENV_SPATIAL && num_slots > 1 && !env->dp.feat_cache) {  // the episode-constant bytes a trajectory step copies
        DevParams &d = env->dp;
        d.featCacheCg = align16(2ll * d.C * d.F);
        d.featCacheStride = align16((long long)d.featCacheCg + (long long)d.C * d.mh * d.mw * d.K);
        DEVICE_GUARD(env);
        if (hipMalloc((void **)&d.feat_cache, (size_t)d.featCacheStride * d.B) != hipSuccess || hipMalloc((void **)&d.feat_cache_tag, 4 * (size_t)d.B) != hipSuccess)
            return fail(env, PCBENV_EHIP, "hipMalloc failed");
    }
    if (env->dp.feat_cache_tag) hipMemset(env->dp.feat_cache_tag, 0xFF, 4 * (size_t)env->dp.B);  // no episode has that number: nothing cached yet
    env->dp.bind_gen += 1;  // feature tensors of these buffers are uninitialised: the next reset of each env fills them
    memset(&env->dp.cbuf, 0, sizeof(env->dp.cbuf));  // compact tensors belong to a binding: bind them again
    env->bound = true;
    return PCBENV_OK;
}

extern "C" int pcbenv_bind_compact_features(pcbenv *env, const pcbenv_compact_features *f) {
    if (!env) return fail(0, PCBENV_EINVAL, "null handle");
    if (!env->bound) return fail(env, PCBENV_ESTATE, "pcbenv_bind_buffers has not been called");
    memset(&env->dp.cbuf, 0, sizeof(env->dp.cbuf));
    if (!f) return PCBENV_OK;
    if (env->dp.num_slots < 2) return fail(env, PCBENV_EINVAL, "compact feature tensors need the trajectory layout (pcbenv_bind_buffers_slots with num_slots > 1)");
    if (env->dp.H > 128 || env->dp.W > 128) return fail(env, PCBENV_ELIMIT, "coordinates do not fit the compact pin tensors");
    env->dp.cbuf = *f;
    const int k = env->cfg.kind;
    if (!is_pin_kind(k)) { env->dp.cbuf.all_pins_num_feature = 0; env->dp.cbuf.all_pins_cat_feature = 0; }
    if (k != PCBENV_RECT) env->dp.cbuf.component_mask = 0;
    if (k == PCBENV_SQUARE) memset(&env->dp.cbuf, 0, sizeof(env->dp.cbuf));
    return PCBENV_OK;
}

extern "C" int pcbenv_load_instances(pcbenv *env, const int32_t *env_ids, int32_t n, int32_t slot,
                                     const void *host_tables, void *stream) {
    if (!env || !host_tables) return fail(env, PCBENV_EINVAL, "null argument");
    const DevParams &d = env->dp;
    if (env->cfg.kind == PCBENV_SQUARE) return PCBENV_OK;  // the square env has no instance
    // refused before anything is copied: the generator owns the records, and k_gen_fill may be writing this very slot
    if (env->gen_on) return fail(env, PCBENV_ESTATE, "the on-device generator owns the queue (pcbenv_instgen_device_enable)");
    if (slot < 0 || slot >= d.Q || n < 0 || n > d.B) return fail(env, PCBENV_EINVAL, "slot or count out of range");
    DEVICE_GUARD(env);
    const long long src_stride = pcbenv_instance_stride(&env->cfg);
    hipStream_t s = (hipStream_t)stream;
    const unsigned char *src = (const unsigned char *)host_tables;
    // sanity-check the records: every index the kernels derive from a record (component, net, feature row, cell
    // inside the component) must stay inside the tables and LDS zones sized from the configuration
    const bool pin_kind = is_pin_kind(env->cfg.kind), spatial = env->cfg.kind == PCBENV_SPATIAL;
    for (int i = 0; i < n; i++) {
        const int32_t *h = (const int32_t *)(src + (size_t)i * src_stride);
        if (h[0] < 1 || h[0] > d.C || h[2] < 0 || h[2] > d.P || h[1] < 0 || h[1] > d.N)
            return fail(env, PCBENV_EINVAL, "instance record out of range (components, nets or pins beyond the configuration)");
        if (!pin_kind && (h[1] != 0 || h[2] != 0)) return fail(env, PCBENV_EINVAL, "instance record carries pins for an environment kind without pins");
        const unsigned char *cr = (const unsigned char *)h + 16, *pr = cr + 8 * (size_t)d.C;
        for (int c = 0; c < h[0]; c++)
            if (cr[8 * c] < 1 || cr[8 * c] > d.mh || cr[8 * c + 1] < 1 || cr[8 * c + 1] > d.mw) return fail(env, PCBENV_EINVAL, "component size out of range");
        int prev = 0;
        unsigned char seen[(PCBENV_MAX_PINS + 7) / 8] = {0};
        int per_net[PCBENV_MAX_NETS] = {0};
        for (int q = 0; q < h[2]; q++) {
            const int net = pr[8 * q + 2], comp = pr[8 * q + 3], id = pr[8 * q + 4] | (pr[8 * q + 5] << 8);
            if (comp >= h[0] || net >= h[1] || (net != prev && net != prev + 1) || (q == 0 && net != 0))
                return fail(env, PCBENV_EINVAL, "pin record out of range, or the pins are not net-major with nets 0, 1, 2, ... in order");
            if (pr[8 * q] >= cr[8 * comp] || pr[8 * q + 1] >= cr[8 * comp + 1]) return fail(env, PCBENV_EINVAL, "pin outside its component");
            if (++per_net[net] > PCBENV_MAX_PINS_PER_NET) return fail(env, PCBENV_EINVAL, "too many pins in one net");
            if (spatial) {  // feature row = the global pin id: a permutation of 0..num_pins-1
                if (id >= h[2] || (seen[id >> 3] >> (id & 7) & 1)) return fail(env, PCBENV_EINVAL, "pin ids must be a permutation of 0..num_pins-1");
                seen[id >> 3] |= (unsigned char)(1u << (id & 7));
            } else if (id >= d.mp) {  // feature row = [component, pin_id]
                return fail(env, PCBENV_EINVAL, "pin id beyond max_num_pins_per_component");
            }
            prev = net;
        }
        if (h[2] > 0 && prev != h[1] - 1) return fail(env, PCBENV_EINVAL, "every net 0..num_nets-1 must have at least one pin");
        if (h[2] == 0 && h[1] != 0 && pin_kind) return fail(env, PCBENV_EINVAL, "nets without pins");
    }
    unsigned char *base = d.queue + (size_t)slot * d.B * d.instStride;
    if (!env_ids && src_stride == d.instStride) {
        HIP_TRY(env, hipMemcpyAsync(base, src, (size_t)n * src_stride, hipMemcpyHostToDevice, s));
    } else {
        for (int i = 0; i < n; i++) {
            int id = env_ids ? env_ids[i] : i;
            if (id < 0 || id >= d.B) return fail(env, PCBENV_EINVAL, "environment id out of range");
            HIP_TRY(env, hipMemcpyAsync(base + (size_t)id * d.instStride, src + (size_t)i * src_stride, (size_t)src_stride, hipMemcpyHostToDevice, s));
        }
    }
    HIP_TRY(env, hipStreamSynchronize(s));
    if (!env_ids && n == d.B) env->loaded_slots[slot >> 6] |= 1ull << (slot & 63);  // partial loads: caller's responsibility
    return PCBENV_OK;
}

static int launch_reset(pcbenv *env, const uint8_t *mask, hipStream_t s) {
    ResetLaunch a{env->dp, mask, env->threads, s};
    a.d.seq = env->seq;  // the next step launch is seq + 1: a reset takes its environments off that launch's terminal list
    switch (env->cfg.kind) {
    case PCBENV_SQUARE: return pcb_launch_reset_square(a);
    case PCBENV_RECT: return pcb_launch_reset_rect(a);
    case PCBENV_PIN: return pcb_launch_reset_pin(a);
    default: return pcb_launch_reset_spatial(a);
    }
}
// Every step launch has a number (DevParams::seq); see k_step_mixed for what the terminal list is.
static int dispatch_step(pcbenv *env, int *actions, int fmt, int sampled, u64 seed, u64 first_env, u64 step_index, int num_steps, hipStream_t s) {
    StepLaunch a;
    a.d = env->dp;
    DevParams &d = a.d;
    a.actions = actions; a.fmt = fmt; a.sampled = sampled; a.seed = seed; a.first_env = first_env; a.step_index = step_index;
    a.num_steps = num_steps; a.threads = env->threads; a.stream = s;
    // in-place build (one transition, store policy compiled in) or the trajectory layout's: one transition into a slot / the rollout loop
    a.traj = d.num_slots > 1 || num_steps > 1;
    // The trajectory layout cycles through num_slots slots: the store policy is chosen on the bytes of all of them (a slot is
    // next written num_slots steps later; one launch per step into a [17, B, ...] trajectory measured + 14 % at c3, + 3 % at
    // c4 with streaming stores).  In place, the steps overwrite the same lines and the per-transition choice of
    // pcbenv_create stands.
    if (a.traj && d.num_slots > 1) d.stream_stores = env->cell_bytes_per_env * d.B * d.num_slots > env->stream_threshold;
    a.routes = is_pin_kind(env->cfg.kind) && env->cfg.reward_type != PCBENV_REWARD_CENTROID;
    if (++env->seq == 0u) env->seq = 1u;  // 0 is "not listed" in the marks
    d.seq = env->seq;
    // A launch that is being captured into a hipGraph will be replayed with these very arguments: no launch number,
    // no buffer swap -- it runs without helpers, keeps no list and works on the state blocks in place.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    d.term_wgs = 0;
    if (num_steps == 1 && !capturing && env->term_wgs > 0) {  // reward helpers: one transition per launch only
        // as many entries' helpers as the lists have lately been long (k_step reports it: + 25 %, + 2 per shard; never none:
        // a shard's first entry)
        const unsigned seen = *(volatile unsigned *)env->term_seen_host;
        const long long want = (long long)TERM_SHARDS * ((long long)seen + seen / 4 + 2);
        d.term_wgs = (int)(want < env->term_wgs ? want : env->term_wgs);
    }
    if (capturing) d.term_cap = 0;
    // double-buffered state blocks: read the current ones, write the others
    d.state = env->state_buf[env->state_cur];
    d.state_out = env->state_buf[env->state_cur ^ (capturing ? 0 : 1)];
    int rc;
    switch (env->cfg.kind) {
    case PCBENV_SQUARE: rc = pcb_launch_step_square(a); break;
    case PCBENV_RECT: rc = pcb_launch_step_rect(a); break;
    case PCBENV_PIN: rc = pcb_launch_step_pin(a); break;
    default: rc = pcb_launch_step_spatial(a); break;
    }
    if (!capturing) env->state_cur ^= 1;
    env->dp.state = env->dp.state_out = env->state_buf[env->state_cur];  // what k_reset / k_sample / get_state work on, in place
    return rc;
}

static int pre_launch(pcbenv *env) {
    if (!env) return fail(0, PCBENV_EINVAL, "null handle");
    if (!env->bound) return fail(env, PCBENV_ESTATE, "pcbenv_bind_buffers has not been called");
    return PCBENV_OK;
}

static int check_queue(pcbenv *env);
static int gen_before_launch(pcbenv *env, int n, hipStream_t main);
static void gen_after_launch(pcbenv *env, int n, hipStream_t main);
extern "C" int pcbenv_reset(pcbenv *env, const uint8_t *mask_dev, void *stream) {
    int rc = pre_launch(env);
    if (rc) return rc;
    DEVICE_GUARD(env);
    rc = check_queue(env);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    rc = gen_before_launch(env, 1, s);
    if (rc) return rc;
    launch_reset(env, mask_dev, s);
    HIP_TRY(env, hipGetLastError());
    gen_after_launch(env, 1, s);
    return PCBENV_OK;
}

static int check_queue(pcbenv *env) {
    if (env->cfg.kind == PCBENV_SQUARE || env->gen_on) return PCBENV_OK;
    for (int s = 0; s < env->dp.Q; s++)
        if (!(env->loaded_slots[s >> 6] >> (s & 63) & 1ull))
            return fail(env, PCBENV_ESTATE, "every queue slot must be loaded (pcbenv_load_instances for all environments) first");
    return PCBENV_OK;
}

// ---- on-device instance generator: host protocol ------------------------------------------------------------
// k_gen_fill runs on env->gen_stream, ordered after everything enqueued on the caller's stream at its snapshot
// (ev_snap): when it has completed, every environment holds queue_depth records ahead of the cursor it had at the
// snapshot.  A launch may consume at most n records per environment (one per reset, one per step with
// PCBENV_FLAG_AUTO_RESET, num_steps per rollout), so a launch is safe as long as the launches since the snapshot of
// the last fill the caller's stream has waited for add up to at most queue_depth; `since_waited` keeps that sum.
// Fills are started early (half of the queue consumed at worst; fewer, larger refills cost the step kernels less than
// many small ones) and waited for late, so they overlap the step
// kernels; the wait is a stream-side event wait, never a host synchronisation.
// one refill launch: 64 / G environments per wavefront (gen_group_lanes), capped grid
static void gen_launch_fill(pcbenv *env, hipStream_t s, bool whole_batch) {
    const int G = gen_group_lanes(env->gp.C, env->cfg.max_num_nets, env->gp.P, env->gen_lanes), epw = WAVE / G;
    int grid = (env->dp.B + epw - 1) / epw;
    if (!whole_batch && grid > env->gen_grid) grid = env->gen_grid;
    const size_t lds = GEN_LDS_BYTES(env->gp.instStride, G);
    if (G == 16) hipLaunchKernelGGL(k_gen_fill<16>, dim3(grid), dim3(WAVE), lds, s, env->gp);
    else if (G == 32) hipLaunchKernelGGL(k_gen_fill<32>, dim3(grid), dim3(WAVE), lds, s, env->gp);
    else hipLaunchKernelGGL(k_gen_fill<64>, dim3(grid), dim3(WAVE), lds, s, env->gp);
}
static void gen_start_fill(pcbenv *env, hipStream_t main) {
    // The fill works from a copy of the published cursors taken in stream order: it learns of a reset only once the launch
    // that made it has COMPLETED, so a record is never overwritten while a team of a running launch may still be reading it
    // (an environment's own team publishes its cursor as soon as it has its copy; its feature helper reads the same record).
    hipMemcpyAsync(env->cursor_snap, env->dp.cursor_pub, 4 * (size_t)env->dp.B, hipMemcpyDeviceToDevice, main);
    hipEventRecord(env->ev_snap, main);
    hipStreamWaitEvent(env->gen_stream, env->ev_snap, 0);
    gen_launch_fill(env, env->gen_stream, false);
    hipEventRecord(env->ev_fill, env->gen_stream);
    env->gen_outstanding = true;
    env->since_outstanding = 0;
}
static int gen_before_launch(pcbenv *env, int n, hipStream_t main) {
    if (!env->gen_on) return PCBENV_OK;
    const long long Q = env->dp.Q;
    if (n > Q) return fail(env, PCBENV_ELIMIT, "this launch may consume more instances per environment than queue_depth holds");
    if (env->since_waited + n > Q) {
        if (!env->gen_outstanding) gen_start_fill(env, main);
        hipStreamWaitEvent(main, env->ev_fill, 0);
        env->since_waited = env->since_outstanding;
        env->gen_outstanding = false;
        if (env->since_waited + n > Q) {  // that fill was snapshotted too long ago for this launch: one more, in stream order
            gen_start_fill(env, main);
            hipStreamWaitEvent(main, env->ev_fill, 0);
            env->since_waited = 0;
            env->gen_outstanding = false;
        }
    }
    return PCBENV_OK;
}
static void gen_after_launch(pcbenv *env, int n, hipStream_t main) {
    if (!env->gen_on) return;
    env->since_waited += n;
    if (env->gen_outstanding) env->since_outstanding += n;
    else if (env->since_waited * 2 >= env->dp.Q) gen_start_fill(env, main);
}

extern "C" int pcbenv_instgen_device_enable(pcbenv *env, const uint32_t *seeds_host, void *stream) {
    if (!env || !seeds_host) return fail(env, PCBENV_EINVAL, "null argument");
    if (env->cfg.kind == PCBENV_SQUARE) return fail(env, PCBENV_EINVAL, "the square environment has no instances");
    if (env->gen_on) return fail(env, PCBENV_ESTATE, "the on-device generator is already enabled");
    DEVICE_GUARD(env);
    const DevParams &d = env->dp;
    const pcbenv_config &c = env->cfg;
    GenParams &g = env->gp;
    g.kind = c.kind; g.C = d.C; g.P = d.P; g.Q = d.Q; g.B = d.B;
    g.min_comp = c.min_num_components; g.max_comp = c.max_num_components;
    g.min_h = c.min_component_h; g.max_h = c.max_component_h; g.min_w = c.min_component_w; g.max_w = c.max_component_w;
    g.min_nets = c.min_num_nets; g.max_nets = c.max_num_nets; g.min_ppn = c.min_num_pins_per_net; g.max_ppn = c.max_num_pins_per_net;
    g.net_distribution = c.net_distribution; g.pin_spread = c.pin_spread;  // clipped at create like the reference does
    g.instStride = d.instStride; g.queue = d.queue;
    HIP_TRY(env, hipMalloc((void **)&env->cursor_snap, 4 * (size_t)d.B));
    HIP_TRY(env, hipMemcpyAsync(env->cursor_snap, d.cursor_pub, 4 * (size_t)d.B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    g.cursor_pub = env->cursor_snap;
    hipStream_t s = (hipStream_t)stream;
    unsigned *seeds_dev = 0;
    HIP_TRY(env, hipMalloc((void **)&g.gen, sizeof(GenState) * (size_t)d.B));
    HIP_TRY(env, hipMalloc((void **)&g.produced, 4 * (size_t)d.B + 4));
    HIP_TRY(env, hipMalloc((void **)&seeds_dev, 4 * (size_t)d.B));
    HIP_TRY(env, hipMemcpyAsync(seeds_dev, seeds_host, 4 * (size_t)d.B, hipMemcpyHostToDevice, s));
    {   // lowest priority: when both queues have workgroups to place, the step kernel's go first
        int lo_prio = 0, hi_prio = 0;
        HIP_TRY(env, hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
        HIP_TRY(env, hipStreamCreateWithPriority(&env->gen_stream, hipStreamNonBlocking, lo_prio));
    }
    HIP_TRY(env, hipEventCreateWithFlags(&env->ev_snap, hipEventDisableTiming));
    HIP_TRY(env, hipEventCreateWithFlags(&env->ev_fill, hipEventDisableTiming));
    const dim3 grid((d.B + WAVE - 1) / WAVE);
    hipLaunchKernelGGL(k_gen_seed, grid, dim3(WAVE), 0, s, g, seeds_dev);
    gen_launch_fill(env, s, true);  // the whole queue, before anything can consume it
    HIP_TRY(env, hipGetLastError());
    HIP_TRY(env, hipStreamSynchronize(s));
    hipFree(seeds_dev);
    env->dp.gen_produced = g.produced;
    env->dp.gen_errors = g.produced + d.B;  // one word behind the counters
    HIP_TRY(env, hipMemsetAsync(env->dp.gen_errors, 0, 4, s));
    if (env->gen_grid < 1) env->gen_grid = GEN_MAX_GRID;  // unless PCBENV_OPT_GEN_GRID set it
    env->gen_on = true; env->gen_outstanding = false;
    env->since_waited = 0; env->since_outstanding = 0;
    return PCBENV_OK;
}

extern "C" int pcbenv_instgen_device_status(pcbenv *env, uint32_t *errors_out, void *stream) {
    if (!env || !errors_out) return fail(env, PCBENV_EINVAL, "null argument");
    if (!env->gen_on) return fail(env, PCBENV_ESTATE, "the on-device generator is not enabled");
    DEVICE_GUARD(env);
    // bring the queue fully up to date (queue_depth records ahead of every cursor as of now) and wait for it
    gen_start_fill(env, (hipStream_t)stream);
    HIP_TRY(env, hipStreamSynchronize(env->gen_stream));
    env->gen_outstanding = false; env->since_waited = 0; env->since_outstanding = 0;
    unsigned err = 0;
    HIP_TRY(env, hipMemcpyAsync(&err, env->dp.gen_errors, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(env, hipStreamSynchronize((hipStream_t)stream));
    std::vector<int> status((size_t)env->dp.B);  // first failing record of any stream (the reference raises there)
    HIP_TRY(env, hipMemcpy2D(status.data(), 4, &env->gp.gen->status, sizeof(GenState), 4, (size_t)env->dp.B, hipMemcpyDeviceToHost));
    for (int i = 0; i < env->dp.B; i++) if (status[(size_t)i] != 0) { err |= 2u; break; }
    *errors_out = err;
    return PCBENV_OK;
}

extern "C" int pcbenv_get_instances(pcbenv *env, int32_t slot, void *host_dst, void *stream) {
    if (!env || !host_dst) return fail(env, PCBENV_EINVAL, "null argument");
    if (slot < 0 || slot >= env->dp.Q) return fail(env, PCBENV_EINVAL, "slot out of range");
    DEVICE_GUARD(env);
    const DevParams &d = env->dp;
    const long long dst_stride = pcbenv_instance_stride(&env->cfg);
    // the copy runs on the stream that last wrote the queue (the generator's, if it is on): in order behind its kernels
    hipStream_t s = env->gen_on ? env->gen_stream : (hipStream_t)stream;
    HIP_TRY(env, hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(env, hipMemcpy2DAsync(host_dst, (size_t)dst_stride, d.queue + (size_t)slot * d.B * d.instStride, (size_t)d.instStride,
                                  (size_t)dst_stride, (size_t)d.B, hipMemcpyDeviceToHost, s));
    HIP_TRY(env, hipStreamSynchronize(s));
    return PCBENV_OK;
}

extern "C" int pcbenv_step(pcbenv *env, const int32_t *actions_dev, int32_t fmt, void *stream) {
    int rc = pre_launch(env);
    if (rc) return rc;
    DEVICE_GUARD(env);
    if (!actions_dev) return fail(env, PCBENV_EINVAL, "null actions");
    if (fmt != PCBENV_ACTION_TUPLE && fmt != PCBENV_ACTION_FLAT) return fail(env, PCBENV_EINVAL, "unknown action format");
    const int consumes = (env->cfg.flags & PCBENV_FLAG_AUTO_RESET) ? 1 : 0;
    rc = gen_before_launch(env, consumes, (hipStream_t)stream);
    if (rc) return rc;
    dispatch_step(env, (int *)actions_dev, fmt, 0, 0, 0, 0, 1, (hipStream_t)stream);
    HIP_TRY(env, hipGetLastError());
    gen_after_launch(env, consumes, (hipStream_t)stream);
    return PCBENV_OK;
}

extern "C" int pcbenv_step_sampled(pcbenv *env, int32_t *actions_out_dev, int32_t fmt, uint64_t seed,
                                   uint64_t first_env_index, uint64_t step_index, void *stream) {
    int rc = pre_launch(env);
    if (rc) return rc;
    DEVICE_GUARD(env);
    if (!actions_out_dev) return fail(env, PCBENV_EINVAL, "null actions");
    if (fmt != PCBENV_ACTION_TUPLE && fmt != PCBENV_ACTION_FLAT) return fail(env, PCBENV_EINVAL, "unknown action format");
    const int consumes = (env->cfg.flags & PCBENV_FLAG_AUTO_RESET) ? 1 : 0;
    rc = gen_before_launch(env, consumes, (hipStream_t)stream);
    if (rc) return rc;
    dispatch_step(env, actions_out_dev, fmt, 1, seed, first_env_index, step_index, 1, (hipStream_t)stream);
    HIP_TRY(env, hipGetLastError());
    gen_after_launch(env, consumes, (hipStream_t)stream);
    return PCBENV_OK;
}

extern "C" int pcbenv_sample_actions(pcbenv *env, int32_t *actions_dev, int32_t fmt, uint64_t seed,
                                     uint64_t first_env_index, uint64_t step_index, void *stream) {
    if (!env || !actions_dev) return fail(env, PCBENV_EINVAL, "null argument");
    if (fmt != PCBENV_ACTION_TUPLE && fmt != PCBENV_ACTION_FLAT) return fail(env, PCBENV_EINVAL, "unknown action format");
    DEVICE_GUARD(env);
    hipLaunchKernelGGL(k_sample, dim3(env->dp.B), dim3(WAVE), 0, (hipStream_t)stream, env->dp, actions_dev, fmt,
                       (u64)seed, (u64)first_env_index, (u64)step_index);
    HIP_TRY(env, hipGetLastError());
    return PCBENV_OK;
}

extern "C" const uint64_t *pcbenv_mask_bits(const pcbenv *env, int64_t *env_stride_bytes) {
    if (!env) return 0;
    if (env_stride_bytes) *env_stride_bytes = env->dp.stateStride;
    return (const uint64_t *)(env->dp.state + env->dp.offVm);
}

// Checkpoint layout: [state blocks, B x stateStride] and, once the on-device generator is enabled,
// [GenState x B | produced, uint32 x B | the instance queue, Q x B x instStride] behind them.
static size_t state_section_bytes(const pcbenv *env) { return (size_t)env->dp.stateStride * env->dp.B; }
static size_t gen_section_bytes(const pcbenv *env) {
    if (!env->gen_on) return 0;
    const size_t B = (size_t)env->dp.B;
    return sizeof(GenState) * B + 4 * B + (size_t)env->dp.instStride * B * (size_t)env->dp.Q;
}
extern "C" int64_t pcbenv_state_bytes(const pcbenv *env) {
    return env ? (int64_t)(state_section_bytes(env) + gen_section_bytes(env)) : 0;
}
extern "C" int pcbenv_get_state(pcbenv *env, void *host_dst, void *stream) {
    if (!env || !host_dst) return fail(env, PCBENV_EINVAL, "null argument");
    DEVICE_GUARD(env);
    hipStream_t s = (hipStream_t)stream;
    const size_t sb = state_section_bytes(env), B = (size_t)env->dp.B;
    unsigned char *dst = (unsigned char *)host_dst;
    HIP_TRY(env, hipMemcpyAsync(dst, env->dp.state, sb, hipMemcpyDeviceToHost, s));
    HIP_TRY(env, hipStreamSynchronize(s));
    if (env->gen_on) {
        // The generator's streams, counters and records are part of what a resumed run continues from.  Bring the queue to
        // its quiescent point first (queue_depth records ahead of every cursor, nothing in flight: the state a restore
        // re-creates), then copy on the generator's stream, in order behind its kernels.
        gen_start_fill(env, s);
        HIP_TRY(env, hipStreamSynchronize(env->gen_stream));
        env->gen_outstanding = false; env->since_waited = 0; env->since_outstanding = 0;
        unsigned char *g = dst + sb;
        HIP_TRY(env, hipMemcpyAsync(g, env->gp.gen, sizeof(GenState) * B, hipMemcpyDeviceToHost, env->gen_stream));
        HIP_TRY(env, hipMemcpyAsync(g + sizeof(GenState) * B, env->gp.produced, 4 * B, hipMemcpyDeviceToHost, env->gen_stream));
        HIP_TRY(env, hipMemcpyAsync(g + sizeof(GenState) * B + 4 * B, env->dp.queue, (size_t)env->dp.instStride * B * (size_t)env->dp.Q,
                                    hipMemcpyDeviceToHost, env->gen_stream));
        HIP_TRY(env, hipStreamSynchronize(env->gen_stream));
    }
    return PCBENV_OK;
}
extern "C" int pcbenv_set_state(pcbenv *env, const void *host_src, void *stream) {
    if (!env || !host_src) return fail(env, PCBENV_EINVAL, "null argument");
    DEVICE_GUARD(env);
    hipStream_t s = (hipStream_t)stream;
    const size_t sb = state_section_bytes(env), B = (size_t)env->dp.B;
    // The terminal-list marks in a checkpoint refer to lists of the run that wrote it: restored environments are not listed.
    std::vector<unsigned char> blob((const unsigned char *)host_src, (const unsigned char *)host_src + sb);
    std::vector<unsigned> cur(B);  // the published copy of the queue cursors follows the restored headers
    for (size_t i = 0; i < B; i++) {
        EnvHdr *hd = (EnvHdr *)(blob.data() + i * env->dp.stateStride);
        hd->term_seq = 0u;
        cur[i] = hd->qcursor;
    }
    if (env->gen_on) {  // nothing of the generator may be in flight while its state is replaced
        HIP_TRY(env, hipStreamSynchronize(s));
        HIP_TRY(env, hipStreamSynchronize(env->gen_stream));
    }
    HIP_TRY(env, hipMemcpyAsync(env->dp.state, blob.data(), sb, hipMemcpyHostToDevice, s));
    HIP_TRY(env, hipMemcpyAsync(env->dp.cursor_pub, cur.data(), 4 * B, hipMemcpyHostToDevice, s));
    if (env->dp.feat_cache_tag) HIP_TRY(env, hipMemsetAsync(env->dp.feat_cache_tag, 0xFF, 4 * B, s));  // the cached bytes are another episode's
    if (env->gen_on) {
        const unsigned char *g = (const unsigned char *)host_src + sb;
        HIP_TRY(env, hipMemcpyAsync(env->gp.gen, g, sizeof(GenState) * B, hipMemcpyHostToDevice, s));
        HIP_TRY(env, hipMemcpyAsync(env->gp.produced, g + sizeof(GenState) * B, 4 * B, hipMemcpyHostToDevice, s));
        HIP_TRY(env, hipMemcpyAsync(env->dp.queue, g + sizeof(GenState) * B + 4 * B, (size_t)env->dp.instStride * B * (size_t)env->dp.Q, hipMemcpyHostToDevice, s));
        env->gen_outstanding = false; env->since_waited = 0; env->since_outstanding = 0;  // the checkpoint was taken at the quiescent point
    }
    HIP_TRY(env, hipStreamSynchronize(s));
    return PCBENV_OK;
}

extern "C" int pcbenv_rollout_sampled(pcbenv *env, int32_t *actions_out_dev, int32_t fmt, int32_t num_steps,
                                      uint64_t seed, uint64_t first_env_index, uint64_t step_index0, void *stream) {
    int rc = pre_launch(env);
    if (rc) return rc;
    DEVICE_GUARD(env);
    if (!actions_out_dev || num_steps < 0) return fail(env, PCBENV_EINVAL, "bad rollout arguments");
    if (fmt != PCBENV_ACTION_TUPLE && fmt != PCBENV_ACTION_FLAT) return fail(env, PCBENV_EINVAL, "unknown action format");
    if (num_steps == 0) return PCBENV_OK;
    const bool auto_reset = (env->cfg.flags & PCBENV_FLAG_AUTO_RESET) != 0;
    if (env->cfg.flags & PCBENV_FLAG_INCREMENTAL_OBS) {  // row-incremental tensors: one launch per step, as before
        const size_t per_step = (size_t)env->dp.B * (fmt == PCBENV_ACTION_TUPLE ? 3 : 1);
        for (int t = 0; t < num_steps; t++) {
            rc = gen_before_launch(env, auto_reset ? 1 : 0, (hipStream_t)stream);
            if (rc) return rc;
            dispatch_step(env, actions_out_dev + per_step * (size_t)t, fmt, 1, seed, first_env_index, step_index0 + (uint64_t)t, 1, (hipStream_t)stream);
            gen_after_launch(env, auto_reset ? 1 : 0, (hipStream_t)stream);
        }
    } else {
        rc = gen_before_launch(env, auto_reset ? num_steps : 0, (hipStream_t)stream);  // every step of it may end an episode
        if (rc) return rc;
        dispatch_step(env, actions_out_dev, fmt, 1, seed, first_env_index, step_index0, num_steps, (hipStream_t)stream);
        gen_after_launch(env, auto_reset ? num_steps : 0, (hipStream_t)stream);
    }
    HIP_TRY(env, hipGetLastError());
    return PCBENV_OK;
}

extern "C" int pcbenv_queue_cursors(pcbenv *env, uint32_t *min_out, uint32_t *max_out, void *stream) {
    if (!env || !min_out || !max_out) return fail(env, PCBENV_EINVAL, "null argument");
    DEVICE_GUARD(env);
    if (!env->scratch) HIP_TRY(env, hipMalloc((void **)&env->scratch, 16));
    hipLaunchKernelGGL(k_cursor_range, dim3(1), dim3(256), 0, (hipStream_t)stream, env->dp, env->scratch);
    unsigned host[2] = {0, 0};
    HIP_TRY(env, hipMemcpyAsync(host, env->scratch, 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(env, hipStreamSynchronize((hipStream_t)stream));
    *min_out = host[0]; *max_out = host[1];
    return PCBENV_OK;
}

#ifdef PCBENV_STAMPS
extern "C" int pcbenv_debug_stamps(pcbenv *env, unsigned long long *host) {  // diagnostic build only
    if (!env || !env->dp.dbg) return -1;
    hipDeviceSynchronize();
    const size_t rows = (size_t)env->dp.B + (size_t)env->dp.term_cap * (REWARD_PARTS + 1);  // environments, then the helpers
    const int rc = hipMemcpy(host, env->dp.dbg, rows * 32 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
    hipMemset(env->dp.dbg, 0, rows * 32 * 8);
    return rc;
}
#endif

