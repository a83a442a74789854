// pcb_kernels.h -- the __global__ kernels: k_reset, k_step, k_sample, k_cursor_range
// Part of libpcbenv.so (CDNA4 / gfx950 only).
#pragma once
#include "pcb_team.h"

template <int KIND, int WW, int NW>
__global__ __launch_bounds__(64 * NW) void k_reset(DevParams p, const unsigned char *__restrict__ mask) {
    typedef Team<64 * NW> T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int e = blockIdx.x, lane = threadIdx.x;
    if (mask && !mask[e]) return;
    T::load_state(smem, p, e, lane);  // cursor / episode survive; the old pins tell which feature rows to clear
    typename T::Lds l = T::carve(smem, p);
    const int row = T::out_row(p, p.slot, e);
    T::template reset_env<KIND, WW, true>(p, l, e, row, lane);
    if (lane == 0) {
        l.hdr->pre_action = 0u;  // the mask changed under any presampled action
        if (p.term_cap > 0) p.term_mark[(size_t)((p.seq + 1u) & 1u) * p.B + e] = 0ull;  // and the environment leaves the terminal list of the next step launch (its helpers find a stale entry)
        p.buf.reward[row] = 0.0;
        p.buf.done[row] = 0;
        if (p.buf.info) { p.buf.info[2 * (size_t)row] = nan(""); p.buf.info[2 * (size_t)row + 1] = nan(""); }
    }
    T::store_state(smem, p, e, lane);
}

// The step kernel, one team of NW wavefronts per workgroup (Team<>::run_env has the story).  Workgroups [0, B) are the
// environments' own teams; a launch with helpers (p.term_wgs > 0: one transition per launch, one-wavefront teams) has
// term_wgs * term_hpe more behind them, the reward helpers of the terminal list: workgroup B + (k * term_hpe + part)
// serves the k-th entry in idx-major order (k = idx * TERM_SHARDS + shard, so that the occupied entries -- the low idx
// of every shard -- come first and start first); unused ones look at their shard's counter and leave.
// Four wavefronts per SIMD (16 one-wavefront workgroups per CU) is all a launch of up to ~4 workgroups per SIMD needs and
// what LDS allows anyway; holding the lean build to 72 VGPRs for 7 wavefronts (spills inside the routing reward and one
// at entry) measured 2-5 % slower at every batch size, so both builds may use up to 128.
template <int KIND, int WW, int NW, bool ROUTES, bool STREAM, bool TRAJ>
__global__ __attribute__((amdgpu_waves_per_eu(4, 8))) __launch_bounds__(64 * NW) void k_step(DevParams p, int *__restrict__ actions, int fmt, int sampled,
                                               u64 seed, u64 first_env, u64 step_index, int num_steps_) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (!TRAJ) p.stream_stores = STREAM;  // the launch's choice as a compile-time constant: only one store policy is compiled in
    const int num_steps = TRAJ ? num_steps_ : 1;
    // above the generator's wavefronts (priority 0) when both share a SIMD: the step kernel is the latency-critical one
    __builtin_amdgcn_s_setprio(3);
    if (p.term_cap > 0 && blockIdx.x == 0 && threadIdx.x < TERM_SHARDS)  // the ring after next starts empty
        store_agent(p.term_cnt + ((((p.seq + 2u) & 3u) * TERM_SHARDS + threadIdx.x) * TERM_CNT_STRIDE), 0u);
    int e = blockIdx.x, role = ROLE_ENV, part = 0;
    unsigned pos = 0u;
    if (NW == 1 && (KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL) && (int)blockIdx.x >= p.B) {  // a reward helper (only launched with p.term_wgs > 0)
        const unsigned hb = blockIdx.x - (unsigned)p.B, hpe = (unsigned)p.term_hpe, k = hb / hpe;
        part = (int)(hb - k * hpe);
        const unsigned ring = p.seq & 3u, cps = (unsigned)p.term_cap / TERM_SHARDS, shard = k & (TERM_SHARDS - 1u), idx = k >> TERM_SHARD_BITS;
        const unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane((int)load_agent(p.term_cnt + (ring * TERM_SHARDS + shard) * TERM_CNT_STRIDE));
        if (idx >= cps || idx >= cnt) return;
        pos = shard * cps + idx;
        // (wave-uniform values the compiler cannot see as such: kept in scalar registers, like blockIdx.x)
        e = __builtin_amdgcn_readfirstlane(p.term_list[ring * (unsigned)p.term_cap + pos]);
        role = ROLE_REWARD;
    }
    Team<64 * NW>::template run_env<KIND, WW, ROUTES, TRAJ>(p, smem, e, threadIdx.x, actions, fmt, sampled, seed, first_env, step_index,
                                                           num_steps, role, part, pos);
}

#ifdef PCB_HOST_TU  // plain (non-template) kernels: defined once, in pcbenv_kernels.hip
__global__ __launch_bounds__(WAVE) void k_sample(DevParams p, int *__restrict__ actions, int fmt, u64 seed,
                                                 u64 first_env, u64 step_index) {
    const int e = blockIdx.x, lane = threadIdx.x;
    const u64 *vm = (const u64 *)(p.state + (size_t)e * p.stateStride + p.offVm);
    int o, x, y;
    Team<64>::sample_action(vm, p, (int)first_env + e, lane, seed, step_index, &o, &x, &y);
    if (lane == 0) {
        if (fmt == PCBENV_ACTION_FLAT) actions[e] = o * p.H * p.W + x * p.W + y;
        else { actions[3 * e] = o; actions[3 * e + 1] = x; actions[3 * e + 2] = y; }
    }
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
#endif
