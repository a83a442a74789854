// pcb_step.h -- the step kernel (transition, mask, observations, terminal reward, optional reset and next-action draw) and the queue-cursor reduction
// Part of libpcbenv.so's single translation unit (included by pcbenv_kernels.hip); CDNA4 / gfx950 only.
#pragma once
#include "pcb_reset.h"

// ----------------------------------------------------------------------------------------------
// step kernel (R:353-432, P:1599-1710, S:1551-1661, Q:115-153)
//   sampled != 0: the action is drawn here (same generator as k_sample) and written to `actions`
//   PCBENV_FLAG_AUTO_RESET: a terminal transition is followed, in the same launch, by the reset
// ----------------------------------------------------------------------------------------------
// Draw the next fused-sampler action from the mask now in l.vm (see EnvHdr::pre_action), or clear a stale one.
__device__ inline void presample_next(const DevParams &p, Lds &l, int sampled, int genv, u64 seed, u64 next_step, int lane) {
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

// One transition of environment e with action (o, x, y): validate_action, update_grid, place_component, features,
// legal mask + observation stream, done, terminal reward, and -- PCBENV_FLAG_AUTO_RESET -- the reset that follows a
// terminal transition.  State is in LDS (l); outputs go to row `row` of the bound tensors.
template <int KIND, int WW, bool ROUTES, bool TRAJ>
__device__ __forceinline__ void transition(const DevParams &p, Lds &l, int e, int row, int lane, int o, int x, int y) {
    const int H = p.H, W = p.W, plane = H * WW;
    const bool auto_reset = p.flags & PCBENV_FLAG_AUTO_RESET;
    const bool full = TRAJ && p.num_slots > 1;  // trajectory layout: every tensor of the destination slot is written whole
    const int cur = l.hdr->cur, ncomp = l.hdr->ncomp, npins = l.hdr->npins;
    // validate_action (S:1699-1723): action_mask[o, x, y] == 1; anything out of range is invalid
    bool valid = o >= 0 && o < p.O && x >= 0 && x < H && y >= 0 && y < W && (KIND == PCBENV_SQUARE || cur >= 0);
    if (valid) valid = (l.vm[(o & 1) * plane + x * WW + (y >> 6)] >> (y & 63)) & 1ull;

    if (lane == 0 && p.buf.info) { p.buf.info[2 * (size_t)row] = nan(""); p.buf.info[2 * (size_t)row + 1] = nan(""); }
    if (NT > WAVE) lds_sync();  // every wavefront has read the cursor before wavefront 0 advances it (one wavefront: program order)

    bool done = true;  // an invalid action is a terminal transition with state and observations unchanged (quirk Q8 iii)
    if (valid) {
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
                    if (!full) write_pin_num<KIND>(p, row, pr);
                }
            }
            if (lane == 0) {
                const int next = cur + 1 < ncomp ? cur + 1 : -1;
                if (!full) {
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
        // When the last component has just been placed and the reset follows in this launch, the terminal cell
        // tensors would be overwritten at once: skip them (terminal by "no legal cell left" is rare and only
        // costs a double write).
        const bool inc = (p.flags & PCBENV_FLAG_INCREMENTAL_OBS) != 0;
        const int r0 = inc ? x : 0, r1 = inc ? min(x + ph, H) : H;
        const bool skip_emit = auto_reset && KIND != PCBENV_SQUARE && l.hdr->cur < 0;
        if (TRAJ && full && !skip_emit) {  // every float64 tensor of the fresh slot, and the episode-constant component_grid
            if (KIND == PCBENV_SPATIAL) build_pin_tables(p, l, lane);
            emit_features_full<KIND>(p, l, row, lane);
            if (KIND == PCBENV_SPATIAL) { emit_component_grid(p, l, row, lane); lds_sync(); }
        }
        const bool any = mask_and_emit<KIND, WW>(p, l, row, lane, !skip_emit, r0, r1);
        STAMP(23);
        if (KIND == PCBENV_SPATIAL && !skip_emit) emit_pin_grid<WW>(p, l, row, lane, r0, r1);
        STAMP(4);
        done = KIND == PCBENV_SQUARE ? !any : (l.hdr->cur < 0 || !any);  // S:1856-1869
    } else if (TRAJ && full && !auto_reset) {  // a fresh slot: the unchanged observation has to be written out all the same
        if (KIND == PCBENV_SPATIAL) build_pin_tables(p, l, lane);
        emit_features_full<KIND>(p, l, row, lane);
        if (KIND == PCBENV_SPATIAL) { emit_component_grid(p, l, row, lane); lds_sync(); }
        mask_and_emit<KIND, WW>(p, l, row, lane, true, 0, H);
        if (KIND == PCBENV_SPATIAL) emit_pin_grid<WW>(p, l, row, lane, 0, H);
    }
    if (lane == 0) p.buf.done[row] = done ? 1 : 0;
    if (KIND == PCBENV_SQUARE || KIND == PCBENV_RECT) { if (lane == 0) p.buf.reward[row] = valid ? 1.0 : 0.0; }  // R:424-432
    else if (!done) { if (lane == 0) p.buf.reward[row] = 0.0; }
    else terminal_reward<KIND, ROUTES>(p, l, row, lane);  // routed if everything is placed, else the worst case (S:853-863)
    STAMP(9);
    if (done && auto_reset) {
        // The reset rewrites every observation, some of them bytes this launch has just stored from other lanes
        // and (four-wavefront environments) other wavefronts: feature rows of the placed component, and -- when the
        // episode ended with no legal cell left, so that nothing was skipped above -- grid / pin_grid / mask chunks.
        // lds_sync() orders LDS only, so drain the stores (s_waitcnt vmcnt(0)) and meet before overwriting them.
        // Terminal wavefronts are latency-bound on the reward; the drain is free by the time they get here.
        __syncthreads();
        reset_env<KIND, WW, TRAJ>(p, l, e, row, lane);
    }
    STAMP(10);
}

// The step kernel.  TRAJ = false is the lean build for the in-place layout and one transition per launch (num_slots
// == 1 and num_steps == 1 are then compile-time facts: no step loop, no whole-tensor feature emission -- 84 instead
// of 112-121 VGPRs); TRAJ = true serves the trajectory layout and the persistent rollout.  num_steps == 1: one transition with the given (or, `sampled`, a uniformly drawn legal) action.
// num_steps > 1 (sampled only) is the persistent rollout: num_steps transitions of environment e in ONE launch
// (step t draws with step_index + t, exactly what k_sample or a single-step launch would draw).  The state block
// stays in LDS for the whole rollout -- no reload / write-back, no launch latency per step, and the wavefronts
// drift apart freely, so a terminal transition (routing reward + reset) delays only its own environment instead
// of the whole batch.  Step t writes its outputs into slot (slot + t) % num_slots of the bound [num_slots, B, ...]
// tensors and its action into actions[t].
template <int KIND, int WW, int NW, bool ROUTES, bool STREAM, bool TRAJ>
// Four wavefronts per SIMD (16 one-wavefront workgroups per CU) is all a launch of up to ~4 workgroups per SIMD needs and
// what LDS allows anyway; holding the lean build to 72 VGPRs for 7 wavefronts (spills inside the routing reward and one
// at entry) measured 2-5 % slower at every batch size, so both builds may use up to 128.
__global__ __attribute__((amdgpu_waves_per_eu(4, 8))) __launch_bounds__(64 * NW) void k_step(DevParams p, int *__restrict__ actions, int fmt, int sampled,
                                               u64 seed, u64 first_env, u64 step_index, int num_steps_) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (!TRAJ) p.stream_stores = STREAM;  // the launch's choice as a compile-time constant: only one store policy is compiled in
    const int num_steps = TRAJ ? num_steps_ : 1;
    // above the generator's wavefronts (priority 0) when both share a SIMD: the step kernel is the latency-critical one
    __builtin_amdgcn_s_setprio(3);
    const int e = blockIdx.x, lane = threadIdx.x;
    const int H = p.H, W = p.W, HW = H * W;
    STAMP_RT(30);
    STAMP(0);
    load_state(smem, p, e, lane);
    Lds l = carve(smem, p);
    STAMP(1);
    const int genv = (int)first_env + e;
    const size_t per_step = (size_t)p.B * (fmt == PCBENV_ACTION_TUPLE ? 3 : 1);
    int slot = p.slot;
    const int lane0 = lane;
    for (int t = 0; t < num_steps; t++) {
        // Every iteration sees the lane index as a fresh value: otherwise the per-lane addressing of all the emission
        // loops is loop-invariant, gets hoisted out of this loop and stays live across the whole body (3x the VGPRs,
        // a third of the wavefronts resident).
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        int o = 0, x = 0, y = 0;
        int *act = actions + per_step * (size_t)t;
        if (sampled) {
            // the tag is read whole before it is tested: `&&` would chain four dependent LDS round trips at the head of every launch
            const unsigned pa = l.hdr->pre_action, pg = l.hdr->pre_genv;
            const u64 ps = l.hdr->pre_seed, pst = l.hdr->pre_step;
            if ((t == 0) & (pa >> 31) & (ps == seed) & (pst == step_index) & (pg == (unsigned)genv)) {
                o = (int)(pa & 0xFFu); x = (int)((pa >> 8) & 0xFFu); y = (int)((pa >> 16) & 0xFFu);  // drawn by the previous launch
                STAMP(21);
            } else {
                if (lane < WAVE) {  // wavefront 0 draws (the result is wave-uniform), the others take it from LDS
                    sample_action(l.vm, p, genv, lane, seed, step_index + (u64)t, &o, &x, &y);
                    if (NW > 1 && lane == 0) { l.hdr->pad[0] = (unsigned)o; l.hdr->pad[1] = (unsigned)x; l.hdr->flag = (unsigned)y; }
                }
                if (NW > 1) {
                    lds_sync();
                    o = (int)l.hdr->pad[0]; x = (int)l.hdr->pad[1]; y = (int)l.hdr->flag;
                }
            }
            if (lane == 0) {
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
        transition<KIND, WW, ROUTES, TRAJ>(p, l, e, TRAJ ? out_row(p, slot, e) : e, lane, o, x, y);
        if (t + 1 < num_steps) {
            // A later step of this launch revisits these addresses (in place, or when the slots wrap around); a
            // reset maps bytes to lanes differently from a step, so order its stores before going on.
            if (l.hdr->episode != episode) __syncthreads(); else lds_sync();
            if (++slot == p.num_slots) slot = 0;
        }
    }
    presample_next(p, l, sampled && num_steps == 1, genv, seed, step_index + 1, lane0);
    STAMP(20);
    store_state(smem, p, e, lane0);
    STAMP(11);
    STAMP_RT(31);
}

// min / max of the per-environment queue cursors (one small workgroup; B <= a few thousand headers)
__global__ __launch_bounds__(256) void k_cursor_range(DevParams p, unsigned *out) {
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    for (int e = threadIdx.x; e < p.B; e += 256) {
        const unsigned c = load_agent(p.cursor_pub + e);  // not the state block: that copy is only coherent on its own XCD
        lo = min(lo, c); hi = max(hi, c);
    }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (unsigned)__shfl_xor((int)lo, o)); hi = max(hi, (unsigned)__shfl_xor((int)hi, o)); }
    __shared__ unsigned slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { lo = min(lo, slo[w]); hi = max(hi, shi[w]); }
        out[0] = lo; out[1] = hi;
    }
}

