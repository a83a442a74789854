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

template <int KIND, int WW, int NW, bool ROUTES, bool STREAM>
__global__ __launch_bounds__(64 * NW) void k_step(DevParams p, int *__restrict__ actions, int fmt, int sampled,
                                               u64 seed, u64 first_env, u64 step_index) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    p.stream_stores = STREAM;  // == the launch's choice; a compile-time constant here, so only one store policy is compiled in
    const int e = blockIdx.x, lane = threadIdx.x;
    const int H = p.H, W = p.W, HW = H * W, plane = H * WW;
    STAMP_RT(30);
    STAMP(0);
    load_state(smem, p, e, lane);
    Lds l = carve(smem, p);
    const bool auto_reset = p.flags & PCBENV_FLAG_AUTO_RESET;
    STAMP(1);

    int o = 0, x = 0, y = 0;
    const int genv = (int)first_env + e;
    if (sampled) {
        const unsigned pa = l.hdr->pre_action;
        if ((pa >> 31) && l.hdr->pre_seed == seed && l.hdr->pre_step == step_index && l.hdr->pre_genv == (unsigned)genv) {
            o = (int)(pa & 0xFFu); x = (int)((pa >> 8) & 0xFFu); y = (int)((pa >> 16) & 0xFFu);  // drawn by the previous launch
            STAMP(21);
        } else {
            if (lane < WAVE) {  // wavefront 0 draws (the result is wave-uniform), the others take it from LDS
                sample_action(l.vm, p, genv, lane, seed, step_index, &o, &x, &y);
                if (NW > 1 && lane == 0) { l.hdr->pad[0] = (unsigned)o; l.hdr->pad[1] = (unsigned)x; l.hdr->flag = (unsigned)y; }
            }
            if (NW > 1) {
                lds_sync();
                o = (int)l.hdr->pad[0]; x = (int)l.hdr->pad[1]; y = (int)l.hdr->flag;
            }
        }
        if (lane == 0) {
            if (fmt == PCBENV_ACTION_FLAT) actions[e] = o * HW + x * W + y;
            else { actions[3 * e] = o; actions[3 * e + 1] = x; actions[3 * e + 2] = y; }
        }
    } else if (fmt == PCBENV_ACTION_FLAT) {  // utils/environment/env_wrappers.py:80-98, :184-199
        const int a = actions[e];
        if (a < 0 || a >= p.O * HW) { o = -1; x = y = 0; }
        else { o = a / HW; const int r = a - o * HW; x = r / W; y = r - x * W; }
    } else {
        o = actions[3 * e]; x = actions[3 * e + 1]; y = actions[3 * e + 2];
        if (KIND == PCBENV_SQUARE) o = 0;
    }
    STAMP(2);
    const int cur = l.hdr->cur;
    // validate_action (S:1699-1723): action_mask[o, x, y] == 1; anything out of range is invalid
    bool valid = o >= 0 && o < p.O && x >= 0 && x < H && y >= 0 && y < W && (KIND == PCBENV_SQUARE || cur >= 0);
    if (valid) valid = (l.vm[(o & 1) * plane + x * WW + (y >> 6)] >> (y & 63)) & 1ull;

    if (lane == 0 && p.buf.info) { p.buf.info[2 * e] = nan(""); p.buf.info[2 * e + 1] = nan(""); }
    lds_sync();

    if (!valid) {  // terminal transition, state and observations unchanged (quirk Q8 iii)
        if (lane == 0) p.buf.done[e] = 1;
        if (KIND == PCBENV_SQUARE || KIND == PCBENV_RECT) { if (lane == 0) p.buf.reward[e] = 0.0; }
        else terminal_reward<KIND, ROUTES>(p, l, e, lane);
        if (auto_reset) {
            __syncthreads();  // as below: the reset rewrites addresses this launch may still be storing to
            reset_env<KIND, WW>(p, l, e, lane);
            presample_next(p, l, sampled, genv, seed, step_index + 1, lane);
            store_state(smem, p, e, lane);
        }
        return;
    }

    int ph, pw;
    if (KIND == PCBENV_SQUARE) ph = pw = p.component_n;
    else {
        const CompRec cr = l.comps[cur];
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
        if (lane == 0) { l.comps[cur].px = (signed char)x; l.comps[cur].py = (signed char)y; }
        if (KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL) {
            const int ch = l.comps[cur].h, cw = l.comps[cur].w;
            for (int q = lane; q < l.hdr->npins; q += NT) {  // S:149-190 place_component
                PinRec pr = l.pins[q];
                if (pr.comp != cur) continue;
                const int rx = pr.rel_x, ry = pr.rel_y;
                if (o == 1) { pr.rel_x = ry; pr.rel_y = ch - rx - 1; }
                else if (o == 2) { pr.rel_x = ch - rx - 1; pr.rel_y = cw - ry - 1; }
                else if (o == 3) { pr.rel_x = cw - ry - 1; pr.rel_y = rx; }
                pr.abs_x = (signed char)(x + pr.rel_x); pr.abs_y = (signed char)(y + pr.rel_y);
                l.pins[q] = pr;
                write_pin_num<KIND>(p, e, pr);
            }
        }
        if (lane == 0) {
            if (p.buf.all_components_feature) {
                double *cf = p.buf.all_components_feature + ((size_t)e * p.C + cur) * p.F;
                cf[2] = x; cf[3] = y;
            }
            const int next = cur + 1 < l.hdr->ncomp ? cur + 1 : -1;
            if (p.buf.placement_mask) {
                double *pm = p.buf.placement_mask + (size_t)e * p.C;
                pm[cur] = KIND == PCBENV_RECT ? 1.0 : 2.0;
                if (next >= 0 && KIND != PCBENV_RECT) pm[next] = 3.0;
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
    const bool any = mask_and_emit<KIND, WW>(p, l, e, lane, !skip_emit, r0, r1);
    STAMP(23);
    if (KIND == PCBENV_SPATIAL && !skip_emit) emit_pin_grid<WW>(p, l, e, lane, r0, r1);
    STAMP(4);
    const bool done = KIND == PCBENV_SQUARE ? !any : (l.hdr->cur < 0 || !any);  // S:1856-1869
    if (lane == 0) p.buf.done[e] = done ? 1 : 0;
    if (KIND == PCBENV_SQUARE || KIND == PCBENV_RECT) { if (lane == 0) p.buf.reward[e] = 1.0; }
    else if (!done) { if (lane == 0) p.buf.reward[e] = 0.0; }
    else terminal_reward<KIND, ROUTES>(p, l, e, lane);
    STAMP(9);
    if (done && auto_reset) {
        // The reset rewrites every observation, some of them bytes this launch has just stored from other lanes
        // and (four-wavefront environments) other wavefronts: feature rows of the placed component, and -- when the
        // episode ended with no legal cell left, so that nothing was skipped above -- grid / pin_grid / mask chunks.
        // lds_sync() orders LDS only, so drain the stores (s_waitcnt vmcnt(0)) and meet before overwriting them.
        // Terminal wavefronts are latency-bound on the reward; the drain is free by the time they get here.
        __syncthreads();
        reset_env<KIND, WW>(p, l, e, lane);
    }
    STAMP(10);
    presample_next(p, l, sampled, genv, seed, step_index + 1, lane);
    STAMP(20);
    store_state(smem, p, e, lane);
    STAMP(11);
    STAMP_RT(31);
}

// min / max of the per-environment queue cursors (one small workgroup; B <= a few thousand headers)
__global__ __launch_bounds__(256) void k_cursor_range(DevParams p, unsigned *out) {
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    for (int e = threadIdx.x; e < p.B; e += 256) {
        const unsigned c = ((const EnvHdr *)(p.state + (size_t)e * p.stateStride))->qcursor;
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

