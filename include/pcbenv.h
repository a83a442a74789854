/*
 * pcbenv.h -- C ABI of libpcbenv.so: batched PCB component-placement
 * environments on MI355X (gfx950).  Plain C, plain pointers and sizes, no torch
 * or C++ types.  Everything is asynchronous on the caller's HIP stream.
 *
 * What each entry point replaces in the reference (PBozmarov/RL-Environment-for-
 * Component-Placement, paths relative to its root):
 *
 *   pcbenv_create          DummyPlacementEnv.__init__ + parameter validation
 *                            environment/dummy_env_square.py:37-72
 *                            environment/dummy_env_rectangular.py:152-251
 *                            environment/dummy_env_rectangular_pin.py:396-641
 *                            environment/dummy_env_rectangular_pin_spatial.py:396-607
 *                          and the factory utils/agent/utils.py:317-418 (init_env/create_env)
 *   pcbenv_instgen_*       generate_instances() itself (NumPy legacy RandomState + CPython random restated)
 *   pcbenv_load_instances  the tables generate_instances() draws at reset
 *                            ..._spatial.py:960-989 (and :931-1212, :1408-1443)
 *   pcbenv_reset           DummyPlacementEnv.reset   ..._spatial.py:1487-1549,
 *                            ..._pin.py:1544-1597, ..._rectangular.py:310-351, ..._square.py:74-113
 *   pcbenv_step            DummyPlacementEnv.step    ..._spatial.py:1551-1661,
 *                            ..._pin.py:1599-1710, ..._rectangular.py:353-432, ..._square.py:115-153
 *                          incl. validate_action, update_grid, place_component,
 *                          compute_action_mask, compute_if_done, find_reward
 *   action formats         utils/environment/env_wrappers.py:80-98, :184-199 (flat Discrete action)
 *   pcbenv_sample_actions, pcbenv_step_sampled
 *                          the uniform-random valid-action policy and its simulate() loop body
 *                            agent/random/random_policy_square.py:11-23 (and siblings)
 *
 * Observations are written into caller-owned device buffers (pcbenv_buffers)
 * and updated in place by the next call; the library owns only the handle, the
 * compact per-environment state and the instance queue.  Invalid actions are
 * data (a terminal transition), never an error.
 */
#ifndef PCBENV_H
#define PCBENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCBENV_ABI_VERSION 3

/* status codes (every function returning int) */
#define PCBENV_OK 0
#define PCBENV_EINVAL (-1)  /* bad parameter: the reference raises ValueError here */
#define PCBENV_ELIMIT (-2)  /* valid for the reference but beyond the HIP path's limits below */
#define PCBENV_EHIP (-3)    /* a HIP runtime call failed (no device, out of memory, launch error) */
#define PCBENV_ESTATE (-4)  /* call order: buffers not bound, no instance loaded, ... */

/* limits of the HIP path */
#define PCBENV_MAX_SIDE 128
#define PCBENV_MAX_COMPONENTS 64
#define PCBENV_MAX_PINS 256
#define PCBENV_MAX_NETS 32
#define PCBENV_MAX_PINS_PER_NET 16
#define PCBENV_MAX_PINS_PER_COMPONENT 64
#define PCBENV_MAX_BEAM_WIDTH 4

enum pcbenv_kind {
    PCBENV_SQUARE = 0,  /* environment/dummy_env_square.py */
    PCBENV_RECT = 1,    /* environment/dummy_env_rectangular.py */
    PCBENV_PIN = 2,     /* environment/dummy_env_rectangular_pin.py */
    PCBENV_SPATIAL = 3  /* environment/dummy_env_rectangular_pin_spatial.py */
};
enum pcbenv_reward_type { PCBENV_REWARD_BEAM = 0, PCBENV_REWARD_CENTROID = 1, PCBENV_REWARD_BOTH = 2 };

/* action encodings accepted by pcbenv_step / produced by pcbenv_sample_actions */
enum pcbenv_action_format {
    PCBENV_ACTION_TUPLE = 0, /* int32 [num_envs, 3] = (orientation, x, y); square reads (x, y) from columns 1, 2 */
    PCBENV_ACTION_FLAT = 1   /* int32 [num_envs]: a = o*H*W + x*W + y (square: x*W + y) */
};

/* flags */
#define PCBENV_FLAG_INCREMENTAL_OBS 1u /* grid / pin_grid: a step writes only the rows it changed (the caller must
                                          not modify those buffers between calls); action_mask is always whole */
#define PCBENV_FLAG_AUTO_RESET 2u      /* a terminal transition is followed, inside the same pcbenv_step, by the
                                          reset of that environment: reward / done / info describe the terminal
                                          transition, the observation tensors already show the next episode
                                          (vector-env convention).  Without it pcbenv_step never resets. */

/* Constructor parameters: the reference constructors' arguments, same names. */
typedef struct pcbenv_config {
    int32_t kind;
    int32_t height, width;
    int32_t min_component_w, max_component_w, min_component_h, max_component_h;
    int32_t max_num_components, min_num_components;
    int32_t net_distribution, pin_spread;
    int32_t min_num_nets, max_num_nets, max_num_pins_per_net, min_num_pins_per_net;
    int32_t reward_type;       /* enum pcbenv_reward_type */
    int32_t reward_beam_width;
    int32_t component_n;       /* square env only */
    double weight_wirelength, weight_num_intersections;
    /* batch */
    int32_t num_envs;          /* environments on this device (one handle per process / GPU) */
    int32_t queue_depth;       /* instances queued per environment (>= 1) */
    uint32_t flags;
    int32_t threads_per_env;   /* 0 = choose; 64 / 256 = threads (1 / 4 wavefronts) per environment */
} pcbenv_config;

/* Observation tensors (device pointers, C-contiguous, leading dim num_envs).
 * A null pointer means "do not produce this tensor".  Cell tensors are uint8
 * (values 0/1, identical to the reference's float64 0.0/1.0); the small feature
 * tensors are float64 with exactly the reference's values.
 *   O  = 1 (square), 2 (rect), 4 (pin, spatial)      C = max_num_components
 *   mp = max_component_h*max_component_w             N = max_num_nets
 *   F  = 5 (rect, pin) or 5+mp (spatial)
 *   R  = C*mp (pin) or C*mp+1 (spatial)              Wc = 1 (pin) or 2 (spatial)            */
typedef struct pcbenv_buffers {
    uint8_t *grid;                  /* [B, H, W]                                all kinds  */
    uint8_t *action_mask;           /* [B, O, H, W] (square: [B, H, W])         all kinds  */
    uint8_t *pin_grid;              /* [B, H, W, N+1]                           spatial    */
    uint8_t *component_grid;        /* [B, C, mh, mw, N+1]; rows >= #components are 0   spatial */
    double *all_components_feature; /* [B, C, F]                                rect/pin/spatial */
    double *placement_mask;         /* [B, C]                                   rect/pin/spatial */
    double *component_mask;         /* [B, C]                                   rect       */
    double *all_pins_num_feature;   /* [B, R, 4] (pin: viewed as [B, C, mp, 4]) pin/spatial */
    double *all_pins_cat_feature;   /* [B, R, Wc] (pin: [B, C, mp, 1])          pin/spatial */
    double *reward;                 /* [B]   required                                       */
    uint8_t *done;                  /* [B]   required                                       */
    double *info;                   /* [B, 2] = (wirelength, num_intersections); NaN where the reference's
                                       info dict is empty                       pin/spatial */
    /* marginals of action_mask for the factorised policies p(o) p(x|o) p(y|o,x), straight from the bit rows
     * (utils/agent/factorized_action_distributions.py:358 reduce_max over (H, W); :401 reduce_max over W) */
    uint8_t *mask_orientation;      /* [B, O]    = max over (x, y) of action_mask     optional, all kinds */
    uint8_t *mask_rows;             /* [B, O, H] = max over y of action_mask          optional, all kinds */
} pcbenv_buffers;

/* Compact feature tensors for the trajectory layout (optional, pcbenv_bind_compact_features): the same values as the
 * float64 feature tensors above in the narrowest integer type that holds them -- every element is a small integer
 * except all_components_feature[..., 4] = area / (H * W), which is carried as its numerator h * w (divide by H * W in
 * float64 to get the reference's value bit for bit).  A rollout that keeps every step's observation writes 8x fewer
 * feature bytes this way (c3: 23.8 KB -> 3.0 KB per env-step).  Layout [num_slots, B, ...] like the other tensors;
 * a null pointer means "not wanted".  */
typedef struct pcbenv_compact_features {
    int16_t *all_components_feature; /* [B, C, F]: h, w, x, y (-1 unplaced), h * w, (spatial) pin ids padded with -1    */
    uint8_t *placement_mask;         /* [B, C]  the reference's codes 0..3 (rect: 0 / 1)                                 */
    uint8_t *component_mask;         /* [B, C]  rect                                                                     */
    int8_t *all_pins_num_feature;    /* [B, R, 4]  rel_x, rel_y, abs_x, abs_y (-1 unplaced; coordinates < 128)            */
    int8_t *all_pins_cat_feature;    /* [B, R, Wc] net (, component); spatial: last row -1                                */
} pcbenv_compact_features;

/* Instance wire format (host memory), one record of pcbenv_instance_stride() bytes:
 *   int32 num_components, num_nets, num_pins, reserved                      (16 bytes)
 *   max_num_components x { uint8 h, w; uint8 pad[6] }                       (8 bytes each)
 *   max_total_pins     x { uint8 rel_x, rel_y, net, component; uint16 pin_id; uint16 pad }
 * Pins are in the order of the reference's `self.pins` list (net-major);
 * max_total_pins = min(max_num_pins_per_net*max_num_nets,
 *                      max_num_components*max_component_h*max_component_w). */
typedef struct pcbenv pcbenv;

int pcbenv_abi_version(void);

/* Validates like the reference constructors (PCBENV_EINVAL) and against the
 * limits above (PCBENV_ELIMIT), selects `device`, allocates state + queue.
 * On failure *out is NULL and pcbenv_last_error(NULL) holds the message. */
int pcbenv_create(const pcbenv_config *cfg, int device, pcbenv **out);
void pcbenv_destroy(pcbenv *env);
const char *pcbenv_last_error(const pcbenv *env);

/* Sizes a host needs to allocate tensors and instance tables. */
int64_t pcbenv_instance_stride(const pcbenv_config *cfg);
int32_t pcbenv_max_total_pins(const pcbenv_config *cfg);

int pcbenv_bind_buffers(pcbenv *env, const pcbenv_buffers *buffers);

/* Tuning options of a handle (none changes a result; defaults are the measured choices of DESIGN.md).  They replace
 * the environment variables earlier builds read: nothing in the library calls getenv. */
enum pcbenv_option {
    PCBENV_OPT_STREAM_THRESHOLD_BYTES = 1, /* cell-tensor bytes per launch above which observation stores bypass the
                                              caches (`nt`); default 256 MiB = the Infinity Cache; 0 = always */
    PCBENV_OPT_TERMINAL_TEAMS = 2,         /* capacity of the terminal list: environments whose next transition is certain to
                                              end their episode get helper wavefronts in that launch (two that share the routing
                                              reward, one that writes the feature half of the reset), so that a batch whose
                                              episodes end at different times steps as fast as one in lock-step; default
                                              num_envs / 8 for the pin kinds, 0 = off */
    PCBENV_OPT_GEN_GRID = 3,               /* workgroups of a refill launch of the on-device generator (default 2 048) */
    PCBENV_OPT_GEN_LANES = 4               /* lanes per environment of the generator kernel: 0 = narrowest the
                                              configuration allows, 32 / 64 force a wider group; before enabling it */
};
int pcbenv_set_option(pcbenv *env, int32_t option, int64_t value);

/* Trajectory layout: every tensor of `buffers` is [num_slots, num_envs, ...] (C-contiguous) instead of
 * [num_envs, ...].  pcbenv_reset / pcbenv_step* write their outputs (observations, reward, done, info, marginals)
 * into the slot chosen with pcbenv_select_slot (0 after binding); step t of pcbenv_rollout_sampled writes slot
 * (selected + t) % num_slots.  This is how a rollout loop (RLlib's sampler, the simulate() loops under agent/random) keeps
 * the observation of every step -- obs[t] -- without copying tensors after each call.  With num_slots > 1 a step
 * cannot rely on what an earlier step left in its destination, so every bound tensor is written whole (the float64
 * feature tensors and component_grid included); PCBENV_FLAG_INCREMENTAL_OBS requires num_slots == 1.  A masked
 * pcbenv_reset writes only the masked environments' rows of the selected slot (the explicit loop "step, then
 * reset the finished ones" therefore keeps one slot per step).
 * pcbenv_bind_buffers(env, b) == pcbenv_bind_buffers_slots(env, b, 1). */
int pcbenv_bind_buffers_slots(pcbenv *env, const pcbenv_buffers *buffers, int32_t num_slots);
int pcbenv_select_slot(pcbenv *env, int32_t slot);

/* Binds (or, with NULL, unbinds) compact feature tensors next to the buffers of pcbenv_bind_buffers_slots; needs the
 * trajectory layout (num_slots > 1, where every step writes every bound tensor whole).  Float64 feature pointers left
 * NULL in pcbenv_buffers are then simply not produced. */
int pcbenv_bind_compact_features(pcbenv *env, const pcbenv_compact_features *features);

/* Copies n packed instance records (host memory) into queue slot `slot`
 * (0 <= slot < queue_depth) of environments env_ids[0..n) (env_ids == NULL:
 * environments 0..n-1).  Synchronous with respect to `stream`. */
int pcbenv_load_instances(pcbenv *env, const int32_t *env_ids, int32_t n, int32_t slot,
                          const void *host_tables, void *stream);

/* reset(): every environment whose mask byte is non-zero (mask_dev == NULL: all)
 * takes the next instance of its queue (round robin over the slots) and
 * rewrites its observations.  Never called implicitly by pcbenv_step. */
int pcbenv_reset(pcbenv *env, const uint8_t *mask_dev, void *stream);

/* step(): one transition of every environment with the given actions.
 * (A launch issued while `stream` is being captured into a hipGraph will be replayed with the very same arguments: it
 * then runs without the helper wavefronts of the terminal list and updates the state blocks in place -- same results.) */
int pcbenv_step(pcbenv *env, const int32_t *actions_dev, int32_t action_format, void *stream);

/* Uniform draw over the currently legal actions of every environment
 * (counter-based generator keyed by (seed, first_env_index + env, step_index));
 * environments without a legal action get action 0. */
int pcbenv_sample_actions(pcbenv *env, int32_t *actions_dev, int32_t action_format, uint64_t seed,
                          uint64_t first_env_index, uint64_t step_index, void *stream);

/* pcbenv_sample_actions + pcbenv_step in one launch: draws the action exactly as pcbenv_sample_actions would,
 * stores it in actions_out_dev (the trajectory record) and applies it.  (The launch also draws, for the same seed and
 * step_index + 1, the action of the next call and keeps it in the state block; it is used only if the next call asks
 * for exactly that (seed, step, environment) and the mask has not changed since -- the result is the same function
 * of (mask, seed, environment, step) either way.) */
int pcbenv_step_sampled(pcbenv *env, int32_t *actions_out_dev, int32_t action_format, uint64_t seed,
                        uint64_t first_env_index, uint64_t step_index, void *stream);

/* num_steps consecutive pcbenv_step_sampled transitions in ONE persistent kernel launch: the per-environment
 * state stays in LDS between the steps (no reload, no launch latency per step), step t draws with step_index0 + t
 * -- the same action pcbenv_step_sampled would draw -- records it in actions_out_dev[t] (int32
 * [num_steps, num_envs, 3], or [num_steps, num_envs] for the flat format) and writes its outputs into slot
 * (selected + t) % num_slots.  The counterpart of the reference's simulate() loop
 * (agent/random/random_policy_square.py:25-58).  Intended with PCBENV_FLAG_AUTO_RESET; every environment may consume
 * up to one queued instance per terminal transition, so queue_depth bounds the resets per environment between
 * refills.  (With PCBENV_FLAG_INCREMENTAL_OBS the steps are separate launches.) */
int pcbenv_rollout_sampled(pcbenv *env, int32_t *actions_out_dev, int32_t action_format, int32_t num_steps,
                           uint64_t seed, uint64_t first_env_index, uint64_t step_index0, void *stream);

/* Native host-side instance generator: stream `seed` (< 2^32) yields, record by record (wire format above), the
 * instances the reference's generate_instances() draws after `np.random.seed(seed); random.seed(seed)`
 * (dummy_env_rectangular_pin_spatial.py:931-1212, :1408-1443; rect/pin siblings).  Not for PCBENV_SQUARE.
 * pcbenv_instgen_next_batch advances n independent streams with `threads` host threads. */
typedef struct pcbenv_instgen pcbenv_instgen;
int pcbenv_instgen_create(const pcbenv_config *cfg, uint64_t seed, pcbenv_instgen **out);
void pcbenv_instgen_destroy(pcbenv_instgen *gen);
int pcbenv_instgen_next(pcbenv_instgen *gen, void *record_out);
int pcbenv_instgen_next_batch(pcbenv_instgen *const *streams, int32_t n, void *records_out, int32_t threads);

/* On-device instance generator: the same streams as pcbenv_instgen_* (stream seeds_host[i] < 2^32 for environment i:
 * what the reference draws after `np.random.seed(s); random.seed(s)`), generated by a kernel on a stream of the
 * library's own, one lane per environment, straight into the instance queue -- so that EVERY reset takes a fresh
 * instance, as the reference's reset() does (..._spatial.py:1487-1549), at the rate the step kernels consume them.
 * Enable once, before or after the first reset; from then on the library owns the queue (pcbenv_load_instances is
 * refused): every pcbenv_reset / pcbenv_step* / pcbenv_rollout_sampled first makes the caller's stream wait (an
 * event wait on the device, never a host synchronisation) for a fill whose records cover whatever that launch can
 * consume -- one record per environment for a reset or an auto-reset step, num_steps for a rollout, which must not
 * exceed queue_depth -- and starts the next fill early enough to overlap the following launches.  A queue of
 * 2-4x the records one launch can consume keeps the generator off the critical path (queue_depth <= 256).
 * pcbenv_instgen_device_status brings the queue fully up to date (queue_depth records ahead of every cursor),
 * synchronises, and reports 0 unless a reset ever found its record missing (bit 0; it never should) or a stream hit
 * a draw the reference itself fails on / this library does not support (bit 1).  pcbenv_get_instances copies one queue slot (num_envs packed
 * records) to the host, e.g. to replay an episode on the CPU. */
int pcbenv_instgen_device_enable(pcbenv *env, const uint32_t *seeds_host, void *stream);
int pcbenv_instgen_device_status(pcbenv *env, uint32_t *errors_out, void *stream);
int pcbenv_get_instances(pcbenv *env, int32_t slot, void *host_dst, void *stream);

/* Smallest and largest number of resets any environment has performed so far (= queue cursors; the next reset of
 * an environment with cursor c reads slot c % queue_depth).  A slot whose episode index is below *min_out has been
 * consumed by every environment and may be refilled.  Synchronises with `stream`. */
int pcbenv_queue_cursors(pcbenv *env, uint32_t *min_out, uint32_t *max_out, void *stream);

/* Checkpoint / resume of the library-owned environment state: pcbenv_state_bytes() bytes of host memory holding all state
 * blocks and -- once pcbenv_instgen_device_enable has been called, when the size grows accordingly -- the on-device
 * generator's streams, counters and queued records, so that a resumed run draws the very instances the original would
 * have (restore into a handle of the same configuration with the generator enabled as well).  A host-fed queue is an
 * input and is reloaded by the caller.  Synchronous w.r.t. `stream`. */
int64_t pcbenv_state_bytes(const pcbenv *env);
int pcbenv_get_state(pcbenv *env, void *host_dst, void *stream);
int pcbenv_set_state(pcbenv *env, const void *host_src, void *stream);

/* Bit-packed legal-action mask of the current component, library-owned device
 * memory: uint64 [B, 2, H, ceil(W/64)] (orientation 0/1; pin kinds: 2 = 0, 3 = 1;
 * square: plane 0 only), bit y of word [b, o, x, y/64] = action_mask[b, o, x, y].
 * Also the row stride between environments in bytes.  The state blocks are double-buffered (a step launch reads one
 * set and writes the other): ask again after every pcbenv_step* / pcbenv_rollout_sampled, the pointer alternates. */
const uint64_t *pcbenv_mask_bits(const pcbenv *env, int64_t *env_stride_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PCBENV_H */
