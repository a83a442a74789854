// pcb_step.h -- the step kernel (transition, mask, observations, terminal reward, optional reset and next-action draw) and the queue-cursor reduction
// Part of libpcbenv.so's single translation unit (included by pcbenv_kernels.hip); CDNA4 / gfx950 only.

// ----------------------------------------------------------------------------------------------
// step kernel (R:353-432, P:1599-1710, S:1551-1661, Q:115-153)
//   sampled != 0: the action is drawn here (same generator as k_sample) and written to `actions`
//   PCBENV_FLAG_AUTO_RESET: a terminal transition is followed, in the same launch, by the reset
// ----------------------------------------------------------------------------------------------
// Draw the next fused-sampler action from the mask now in l.vm (see EnvHdr::pre_action), or clear a stale one.
static __device__ inline void presample_next(const DevParams &p, Lds &l, int sampled, int genv, u64 seed, u64 next_step, int lane) {
    if (lane >= WAVE) return;
#ifdef PCBENV_NO_PRESAMPLE
    sampled = 0;
#endif
    if (!sampled) { if (lane == 0) l.hdr->pre_action = 0u; return; }
    int o, x, y;
    sample_action(l.vm, p, genv, lane, seed, next_step, &o, &x, &y);
    if (lane == 0) {
        l.hdr->pre_seed = seed; l.hdr->pre_step = next_step; l.hdr->pre_genv = (unsigned)genv;
        l.hdr->pre_action = (unsigned)o | ((unsigned)x << 8) | ((unsigned)y << 16) | 0x80000000u;
    }
}

// Who runs what of a transition (run_env has the story of the terminal list and its helper teams):
//   MODE_ALL        the environment's own team does everything (no helpers in this launch, or the environment is not listed);
//   MODE_DELEGATED  the environment's own team; its episode is certain to end with this transition and REWARD_PARTS helper
//                   teams of the same launch compute the routing reward meanwhile: it does everything else -- the
//                   transition, the reset that follows (PCBENV_FLAG_AUTO_RESET), the state block -- and the worst-case
//                   reward if the action turns out invalid (the helpers then leave);
//   MODE_REWARD     a reward helper: replays the placement of the last component's pins in its own LDS copy of the state
//                   and counts its part of the segment pairs; writes nothing but (if its share arrives last) reward and info;
//   MODE_FEATURES   the feature helper (PCBENV_FLAG_AUTO_RESET only: the reset is then as certain as the end of the
//                   episode): writes the half of the reset's observations that the environment's own team leaves out --
//                   feature tensors, pin_grid, component_grid (reset_env's RESET_FEATURES) -- from its own copy of the old
//                   state and the same queued instance.
// Every team derives what happens (valid action or not, last component or not) from the same state block and the same
// action, so they agree without talking to each other; only the environment's own team ever writes its state block.
#define MODE_ALL 0
#define MODE_DELEGATED 1
#define MODE_REWARD 2
#define MODE_FEATURES 3
#define REWARD_PARTS 2  // reward helpers per listed environment (+ one feature helper with PCBENV_FLAG_AUTO_RESET)

// One transition of environment e with action (o, x, y): validate_action, update_grid, place_component, features,
// legal mask + observation stream, done, terminal reward, and -- PCBENV_FLAG_AUTO_RESET -- the reset that follows a
// terminal transition.  State is in LDS (l); outputs go to row `row` of the bound tensors.
template <int KIND, int WW, bool ROUTES, bool TRAJ>
static __device__ __forceinline__ void transition(const DevParams &p, Lds &l, int e, int row, int lane, int o, int x, int y, int mode, int part, unsigned pos) {
    const int H = p.H, W = p.W, plane = H * WW;
    const bool auto_reset = p.flags & PCBENV_FLAG_AUTO_RESET;
    if (mode == MODE_FEATURES) { reset_env<KIND, WW, TRAJ>(p, l, e, row, lane, RESET_FEATURES); return; }
    const bool full = TRAJ && p.num_slots > 1;  // trajectory layout: every tensor of the destination slot is written whole
    const int cur = l.hdr->cur, ncomp = l.hdr->ncomp, npins = l.hdr->npins;
    // validate_action (S:1699-1723): action_mask[o, x, y] == 1; anything out of range is invalid
    bool valid = o >= 0 && o < p.O && x >= 0 && x < H && y >= 0 && y < W && (KIND == PCBENV_SQUARE || cur >= 0);
    if (valid) valid = (l.vm[(o & 1) * plane + x * WW + (y >> 6)] >> (y & 63)) & 1ull;
    if (mode == MODE_REWARD && !valid) return;  // the worst-case reward of an invalid action is the environment's own team's
    const bool obs = mode != MODE_REWARD;  // a reward helper writes no observation byte
    // Delegated and valid: the last component is being placed, the helpers route the reward from their own replay of it.
    // With the reset to follow in this launch nothing of the placement is left to do here -- every observation byte and
    // the whole state are about to be rewritten.
    const bool helpers_route = mode == MODE_DELEGATED && valid;

    // (a delegated transition is terminal for certain: its info is written with the reward, by the last helper or below)
    if (lane == 0 && p.buf.info && mode == MODE_ALL) { p.buf.info[2 * (size_t)row] = nan(""); p.buf.info[2 * (size_t)row + 1] = nan(""); }
    if (NT > WAVE) lds_sync();  // every wavefront has read the cursor before wavefront 0 advances it (one wavefront: program order)

    bool done = true;  // an invalid action is a terminal transition with state and observations unchanged (quirk Q8 iii)
    if (valid && !(helpers_route && auto_reset)) {
        int ph, pw;
        CompRec cr = CompRec();
        if (KIND == PCBENV_SQUARE) ph = pw = p.component_n;
        else {
            cr = l.comps[cur];
            ph = (o & 1) ? cr.w : cr.h;  // S:1742-1747 update_grid
            pw = (o & 1) ? cr.h : cr.w;
        }
        // update_grid: rows x..x+ph-1, columns y..y+pw-1
        for (int r = x + lane; r < x + ph && r < H; r += NT) {
            for (int w = 0; w < WW; w++) {
                const int lo = max(y, 64 * w) - 64 * w, hi = min(y + pw, 64 * w + 64) - 64 * w;  // bit range in word w
                if (hi > lo) l.occ[r * WW + w] |= ((hi - lo) >= 64 ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << lo;
            }
        }
        if (KIND != PCBENV_SQUARE) {
            if (lane == 0) { l.comps[cur].px = (signed char)x; l.comps[cur].py = (signed char)y; l.comps[cur].o = (unsigned char)o; }
            if (KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL) {
                const int ch = cr.h, cw = cr.w;
                for (int q = lane; q < npins; q += NT) {  // S:149-190 place_component
                    PinRec pr = l.pins[q];
                    if (pr.comp != cur) continue;
                    const int rx = pr.rel_x, ry = pr.rel_y;
                    if (o == 1) { pr.rel_x = ry; pr.rel_y = ch - rx - 1; }
                    else if (o == 2) { pr.rel_x = ch - rx - 1; pr.rel_y = cw - ry - 1; }
                    else if (o == 3) { pr.rel_x = cw - ry - 1; pr.rel_y = rx; }
                    pr.abs_x = (signed char)(x + pr.rel_x); pr.abs_y = (signed char)(y + pr.rel_y);
                    l.pins[q] = pr;
                    if (!full && obs) write_pin_num<KIND>(p, row, pr);
                }
            }
            if (lane == 0) {
                const int next = cur + 1 < ncomp ? cur + 1 : -1;
                if (!full && obs) {
                    if (p.buf.all_components_feature) {
                        double *cf = p.buf.all_components_feature + ((size_t)row * p.C + cur) * p.F;
                        cf[2] = x; cf[3] = y;
                    }
                    if (p.buf.placement_mask) {
                        double *pm = p.buf.placement_mask + (size_t)row * p.C;
                        pm[cur] = KIND == PCBENV_RECT ? 1.0 : 2.0;
                        if (next >= 0 && KIND != PCBENV_RECT) pm[next] = 3.0;
                    }
                }
                l.hdr->cur = (short)next;
            }
        }
        lds_sync();
        STAMP(3);
        if (obs) {
            // When the last component has just been placed and the reset follows in this launch, the terminal cell
            // tensors would be overwritten at once: skip them (terminal by "no legal cell left" is rare and only
            // costs a double write).
            const bool inc = (p.flags & PCBENV_FLAG_INCREMENTAL_OBS) != 0;
            const int r0 = inc ? x : 0, r1 = inc ? min(x + ph, H) : H;
            const bool skip_emit = auto_reset && KIND != PCBENV_SQUARE && l.hdr->cur < 0;
            // Trajectory layout: every feature tensor of the fresh slot, and the episode-constant component_grid -- copied
            // from the episode's cache where there is one.  They go LAST where nothing they touch in LDS is needed by the
            // cell tensors' emission: their reads (the cache: HBM reads on a chip saturated with writes, ~2 us a round
            // trip, five of them) and small stores then overlap the drain of the 56 KB the wavefront has just issued,
            // instead of holding those back at a time when every wavefront of the launch is doing the same (c4:
            // 71.5 -> 64-67 us per step, profiles/r3/ab_trajectory_slot_build.txt).
            const bool slot_features = TRAJ && full && !skip_emit;
            const bool from_cache = slot_features && KIND == PCBENV_SPATIAL && feat_cache_valid(p, l, e);
            const bool late = slot_features && (KIND == PCBENV_SPATIAL ? from_cache : KIND == PCBENV_PIN);
            auto emit_slot_features = [&]() {
                if (from_cache) {
                    STAMP(5);  // (stamps 5-7 are the reward's in a terminal transition: a non-terminal one has them free)
                    emit_features_full<KIND>(p, l, row, lane, true);
                    STAMP(6);
                    feat_cache_emit(p, l, e, row, lane);
                    STAMP(7);
                } else {
                    if (KIND == PCBENV_SPATIAL) build_pin_tables(p, l, lane);
                    emit_features_full<KIND>(p, l, row, lane);
                    if (KIND == PCBENV_SPATIAL) { emit_component_grid(p, l, row, lane); lds_sync(); }
                }
            };
            if (slot_features && !late) emit_slot_features();
            const bool any = mask_and_emit<KIND, WW>(p, l, row, lane, !skip_emit, r0, r1);
            STAMP(23);
            if (KIND == PCBENV_SPATIAL && !skip_emit) emit_pin_grid<WW>(p, l, row, lane, r0, r1);
            STAMP(4);
            if (late) emit_slot_features();
            done = KIND == PCBENV_SQUARE ? !any : (l.hdr->cur < 0 || !any);  // S:1856-1869
        }  // (a reward helper: the last component has just been placed -- that is what being listed is conditional on)
    } else if (!valid && TRAJ && full && !auto_reset && obs) {  // a fresh slot: the unchanged observation has to be written out all the same
        if (KIND == PCBENV_SPATIAL) build_pin_tables(p, l, lane);
        emit_features_full<KIND>(p, l, row, lane);
        if (KIND == PCBENV_SPATIAL) { emit_component_grid(p, l, row, lane); lds_sync(); }
        mask_and_emit<KIND, WW>(p, l, row, lane, true, 0, H);
        if (KIND == PCBENV_SPATIAL) emit_pin_grid<WW>(p, l, row, lane, 0, H);
    }
    if (mode != MODE_REWARD) {
        if (lane == 0) p.buf.done[row] = done ? 1 : 0;
        if (KIND == PCBENV_SQUARE || KIND == PCBENV_RECT) { if (lane == 0) p.buf.reward[row] = valid ? 1.0 : 0.0; }  // R:424-432
        else if (!done) { if (lane == 0) p.buf.reward[row] = 0.0; }
    }
    // routed if everything is placed (the reward helpers' work when delegated), else the worst case (S:853-863)
    if ((KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL) && done && !helpers_route)
        terminal_reward<KIND, ROUTES>(p, l, row, lane, part, mode == MODE_REWARD ? REWARD_PARTS : 1, pos);
    STAMP(9);
    if (done && auto_reset && mode != MODE_REWARD) {
        // The reset rewrites every observation, some of them bytes this launch has just stored from other lanes
        // and (four-wavefront environments) other wavefronts: feature rows of the placed component, and -- when the
        // episode ended with no legal cell left, so that nothing was skipped above -- grid / pin_grid / mask chunks.
        // lds_sync() orders LDS only, so drain the stores (s_waitcnt vmcnt(0)) and meet before overwriting them.
        // Terminal wavefronts are latency-bound on the reward; the drain is free by the time they get here.
        store_drain_sync();
        reset_env<KIND, WW, TRAJ>(p, l, e, row, lane, mode == MODE_DELEGATED ? RESET_STATE_MASKS : RESET_ALL);
    }
    STAMP(10);
}

// One team's share of a step launch: state block -> LDS, num_steps transitions of environment e, state block back.
// TRAJ = false is the lean build for the in-place layout and one transition per launch (num_slots == 1 and num_steps == 1
// are then compile-time facts: no step loop, no whole-tensor feature emission -- 84 instead of 112-125 VGPRs); TRAJ =
// true serves the trajectory layout -- with num_steps a compile-time 1 in k_step's STEP_BUILD_SLOT (no loop either: 80
// VGPRs) -- and the persistent rollout.  num_steps == 1: one transition with the given (or,
// `sampled`, a uniformly drawn legal) action.  num_steps > 1 (sampled only) is the persistent rollout: num_steps
// transitions of environment e in ONE launch (step t draws with step_index + t, exactly what k_sample or a single-step
// launch would draw).  The state block stays in LDS for the whole rollout -- no reload / write-back, no launch latency
// per step, and the wavefronts drift apart freely, so a terminal transition (routing reward + reset) delays only its
// own environment instead of the whole batch.  Step t writes its outputs into slot (slot + t) % num_slots of the
// bound [num_slots, B, ...] tensors and its action into actions[t].
//
// The terminal list.  With one launch per step and episodes that end at different times (a policy between the steps,
// staggered phases) a launch lasts as long as its slowest wavefront, and the wavefront of an environment that ends an
// episode walks ~67 k cycles (routing reward 33 k, reset 20 k) against 27 k for a plain transition.  Those
// environments are known a launch ahead: once the last component is the current one, the next transition ends the
// episode whatever the action is.  Such an environment puts itself on the list of the next launch (below), and that
// launch starts extra one-wavefront HELPER teams per list entry (k_step): REWARD_PARTS that each count a share of the
// routing reward's segment pairs from their own copy of the state (ROLE_REWARD) and, with PCBENV_FLAG_AUTO_RESET, one
// that writes the feature half of the reset's observations (ROLE_FEATURES), while the environment's own team (ROLE_ENV)
// goes straight on to its half of the reset, so that no wavefront of the launch has much more to do than a plain
// transition.  The environment's own team sees that it is listed from the same
// word the helpers check -- term_mark[seq & 1][e] = (launch number, list position), written together with the list
// entry by the previous launch and by nothing during this one -- and that the last component is still the current one
// from the same state block: state blocks are double-buffered -- a launch reads p.state and writes p.state_out, the host
// swaps them -- so what a helper loads is the state the launch started from however late it starts and however early the
// environment's own team has written the next one.  So all of them agree on who does what whatever has happened to the
// environment in between (a k_reset takes it off the list).  The list is a scheduling hint, never a fact a result
// depends on.
#define ROLE_ENV 0
#define ROLE_REWARD 1
#define ROLE_FEATURES 2
template <int KIND, int WW, bool ROUTES, bool TRAJ>
static __device__ __forceinline__ void run_env(const DevParams &p_launch, unsigned char *smem, int e, int lane0, int *actions, int fmt, int sampled,
                                               u64 seed, u64 first_env, u64 step_index, int num_steps, int role, int part, unsigned pos) {
    // `p`: the launch's parameters (diagnostic build: with the stamp rows indexed by environment, the helpers' rows behind)
    STAMP_ROWS_BY_ENV(p_launch, role == ROLE_ENV ? e : p_launch.B + (int)pos * (REWARD_PARTS + 1) + part);
    const int H = p.H, W = p.W, HW = H * W;
    STAMP_RT(30);
    STAMP(0);
    load_state(smem, p, e, lane0);
    Lds l = carve(smem, p);
    STAMP(1);
    int mode = MODE_ALL;
    if (KIND != PCBENV_SQUARE && p.term_wgs > 0) {  // team-uniform: every lane of every team of e reads the same words
        const u64 mk = ((u64)l.hdr->term_seq << 32) | l.hdr->term_pos;
        // (listed AND this launch has started the entry's helpers: the helper grid follows the list lengths the host has seen)
        const bool listed = (unsigned)(mk >> 32) == p.seq && (unsigned)mk < (unsigned)p.term_wgs && (role == ROLE_ENV || (unsigned)mk == pos);
        const int cur = l.hdr->cur;
        const bool last = cur >= 0 && cur == l.hdr->ncomp - 1;
        if (listed && last) mode = role == ROLE_ENV ? MODE_DELEGATED : role == ROLE_REWARD ? MODE_REWARD : MODE_FEATURES;
        else if (role != ROLE_ENV) return;  // a stale list entry: the environment's own team does everything
    }
    const int genv = (int)first_env + e;
    const size_t per_step = (size_t)p.B * (fmt == PCBENV_ACTION_TUPLE ? 3 : 1);
    int slot = p.slot;
    for (int t = 0; t < num_steps; t++) {
        // Every iteration sees the lane index as a fresh value: otherwise the per-lane addressing of all the emission
        // loops is loop-invariant, gets hoisted out of this loop and stays live across the whole body (3x the VGPRs,
        // a third of the wavefronts resident).
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        int o = 0, x = 0, y = 0;
        int *act = actions + per_step * (size_t)t;
        if (mode == MODE_FEATURES) {
            // (the action plays no part in a reset)
        } else if (sampled) {
            // the tag is read whole before it is tested: `&&` would chain four dependent LDS round trips at the head of every launch
            const unsigned pa = l.hdr->pre_action, pg = l.hdr->pre_genv;
            const u64 ps = l.hdr->pre_seed, pst = l.hdr->pre_step;
            if ((t == 0) & (pa >> 31) & (ps == seed) & (pst == step_index) & (pg == (unsigned)genv)) {
                o = (int)(pa & 0xFFu); x = (int)((pa >> 8) & 0xFFu); y = (int)((pa >> 16) & 0xFFu);  // drawn by the previous launch
                STAMP(21);
            } else {
                if (lane < WAVE) {  // wavefront 0 draws (the result is wave-uniform), the others take it from LDS
                    sample_action(l.vm, p, genv, lane, seed, step_index + (u64)t, &o, &x, &y);
                    if (NT > WAVE && lane == 0) { l.hdr->pad[0] = (unsigned)o; l.hdr->pad[1] = (unsigned)x; l.hdr->flag = (unsigned)y; }
                }
                if (NT > WAVE) {
                    lds_sync();
                    o = (int)l.hdr->pad[0]; x = (int)l.hdr->pad[1]; y = (int)l.hdr->flag;
                }
            }
            if (lane == 0 && role == ROLE_ENV) {
                if (fmt == PCBENV_ACTION_FLAT) act[e] = o * HW + x * W + y;
                else { act[3 * e] = o; act[3 * e + 1] = x; act[3 * e + 2] = y; }
            }
        } else if (fmt == PCBENV_ACTION_FLAT) {  // utils/environment/env_wrappers.py:80-98, :184-199
            const int a = act[e];
            if (a < 0 || a >= p.O * HW) { o = -1; x = y = 0; }
            else { o = a / HW; const int r = a - o * HW; x = r / W; y = r - x * W; }
        } else {
            o = act[3 * e]; x = act[3 * e + 1]; y = act[3 * e + 2];
            if (KIND == PCBENV_SQUARE) o = 0;
        }
        STAMP(2);
        const unsigned episode = l.hdr->episode;
        transition<KIND, WW, ROUTES, TRAJ>(p, l, e, TRAJ ? out_row(p, slot, e) : e, lane, o, x, y, mode, part, pos);
        if (t + 1 < num_steps) {
            // A later step of this launch revisits these addresses (in place, or when the slots wrap around); a
            // reset maps bytes to lanes differently from a step, so order its stores before going on.
            if (l.hdr->episode != episode) store_drain_sync(); else lds_sync();
            if (++slot == p.num_slots) slot = 0;
        }
    }
    if (mode == MODE_REWARD || mode == MODE_FEATURES) { STAMP(11); STAMP_RT(31); return; }  // what is left belongs to the environment's own team
    presample_next(p, l, sampled && num_steps == 1, genv, seed, step_index + 1, lane0);
    // Terminal list of the NEXT launch (see above).
    if (KIND != PCBENV_SQUARE && p.term_cap > 0 && lane0 == 0) {
        const int cur = l.hdr->cur;
        u64 mark = 0ull;  // launch number 0: not listed
        if (cur >= 0 && cur == l.hdr->ncomp - 1) {
            // Sixteen counters (a hashed shard each, on lines of their own): when the whole batch is about to finish
            // together (episodes in lock-step) 4 096 returning atomic adds on one word took 37 us.
            const unsigned ring = (p.seq + 1u) & 3u, cps = (unsigned)p.term_cap / TERM_SHARDS;
            const unsigned shard = ((unsigned)e * 0x9E3779B1u) >> (32 - TERM_SHARD_BITS);
            const unsigned idx = atomicAdd(p.term_cnt + (ring * TERM_SHARDS + shard) * TERM_CNT_STRIDE, 1u);
            if (idx < cps) {
                const unsigned mpos = idx * TERM_SHARDS + shard;  // entries in the order their helpers are launched: the first of every shard first
                p.term_list[ring * (unsigned)p.term_cap + mpos] = e;
                mark = ((u64)(p.seq + 1u) << 32) | mpos;
            }
        }
        l.hdr->term_seq = (unsigned)(mark >> 32); l.hdr->term_pos = (unsigned)mark;
    }
    STAMP(20);
    store_state(smem, p, e, lane0);
    STAMP(11);
    STAMP_RT(31);
}
