// pcb_kernels.h -- the __global__ kernels: k_reset, k_step, k_sample, k_cursor_range
// Part of libpcbenv.so (CDNA4 / gfx950 only).
#pragma once
#include "pcb_team.h"

// Workgroups go to the eight XCDs round-robin (blockIdx.x % 8).  Environment of workgroup `block` (the environments'
// workgroups follow `head` others): XCD * B/8 + turn, so that each XCD -- each L2 -- owns a contiguous eighth of every
// tensor.  Rows smaller than a cache line (reward, done, info, actions, the compact features) and the ends of the
// others then share their lines with neighbours under the SAME L2, which merges them into whole-line writes; with
// environment = blockIdx.x every such line went to memory in up to eight pieces (c3: 19.65 -> 19.1 us per launch, c4
// 44.8 -> 44.3, same-box A/B in profiles/r3/ab_xcd_contiguous_environments.txt).  Any bijection serves: the
// environments are independent, and nothing else depends on which workgroup runs which.
__device__ inline int xcd_contiguous_env(int block, int head, int B) {
    if (B & 7) return block - head;
    const int x = block & 7, first = head + ((x - head) & 7);  // first: the XCD's first environment workgroup
    return x * (B >> 3) + (block - first) / 8;
}

template <int KIND, int WW, int NW>
__global__ __launch_bounds__(64 * NW) void k_reset(DevParams p, const unsigned char *__restrict__ mask) {
    typedef Team<64 * NW> T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int e = xcd_contiguous_env((int)blockIdx.x, 0, p.B), lane = threadIdx.x;
    if (mask && !mask[e]) return;
    T::load_state(smem, p, e, lane);  // cursor / episode survive; the old pins tell which feature rows to clear
    typename T::Lds l = T::carve(smem, p);
    const int row = T::out_row(p, p.slot, e);
    T::template reset_env<KIND, WW, true>(p, l, e, row, lane);
    if (lane == 0) {
        l.hdr->pre_action = 0u;  // the mask changed under any presampled action
        l.hdr->term_seq = 0u;    // and the environment leaves the terminal list of the next step launch (its helpers find a stale entry)
        p.buf.reward[row] = 0.0;
        p.buf.done[row] = 0;
        if (p.buf.info) { p.buf.info[2 * (size_t)row] = nan(""); p.buf.info[2 * (size_t)row + 1] = nan(""); }
    }
    T::store_state(smem, p, e, lane);
}

// The step kernel, one team of NW wavefronts per workgroup (Team<>::run_env has the story).  A launch with helpers
// (p.term_wgs > 0: one transition per launch, pin kinds) starts with term_wgs * term_hpe
// helpers -- workgroup k * term_hpe + part serves entry k of the terminal list, entries being numbered idx *
// TERM_SHARDS + shard so that the occupied ones (the low idx of every shard) come first; unused ones look at their
// shard's counter and leave -- followed by the B environments' own teams.  The helpers come FIRST so that they hold a
// slot from the first cycle of the launch (behind 4 096 environments' workgroups the last of them found none until the
// first environments had finished, 14 us into a 20 us launch), and there are only as many as the lists have lately been
// long: the first environment workgroup reports this launch's longest shard to host memory, pcbenv_step* sizes the next
// helper grids from it (an environment is only delegated if its entry's helpers were started).
// Four wavefronts per SIMD (16 one-wavefront workgroups per CU) is all a launch of up to ~4 workgroups per SIMD needs and
// what LDS allows anyway; holding the lean build to 72 VGPRs for 7 wavefronts (spills inside the routing reward and one
// at entry) measured 2-5 % slower at every batch size, so both builds may use up to 128.
// BUILD: what is a compile-time fact of the launch.  STEP_BUILD_INPLACE / _INPLACE_STREAM: the in-place layout, one
// transition, write-through / streaming stores (the lean build: no step loop, no whole-tensor feature emission);
// STEP_BUILD_SLOT: the trajectory layout, one transition per launch (what a PPO collect runs: whole-tensor emission, no
// loop -- 80 VGPRs and 140 spilled scalars against the rollout build's 125 and 800, c4 68 instead of 80 us per step);
// STEP_BUILD_ROLLOUT: the trajectory layout and num_steps transitions per launch (the persistent rollout).
#define STEP_BUILD_INPLACE 0
#define STEP_BUILD_INPLACE_STREAM 1
#define STEP_BUILD_SLOT 2
#define STEP_BUILD_ROLLOUT 3
template <int KIND, int WW, int NW, bool ROUTES, int BUILD>
__global__ __attribute__((amdgpu_waves_per_eu(4, 8))) __launch_bounds__(64 * NW) void k_step(DevParams p, int *__restrict__ actions, int fmt, int sampled,
                                               u64 seed, u64 first_env, u64 step_index, int num_steps_) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool TRAJ = BUILD == STEP_BUILD_SLOT || BUILD == STEP_BUILD_ROLLOUT;
    if (!TRAJ) p.stream_stores = BUILD == STEP_BUILD_INPLACE_STREAM;  // the launch's choice as a compile-time constant: only one store policy is compiled in
    const int num_steps = BUILD == STEP_BUILD_ROLLOUT ? num_steps_ : 1;
    // above the generator's wavefronts (priority 0) when both share a SIMD: the step kernel is the latency-critical one
    __builtin_amdgcn_s_setprio(3);
    constexpr bool HELPERS = KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL;
    const int nh = HELPERS ? p.term_wgs * p.term_hpe : 0;  // helper workgroups at the head of the grid
    int e = (int)blockIdx.x - nh, role = ROLE_ENV, part = 0;
    if (e >= 0) e = xcd_contiguous_env((int)blockIdx.x, nh, p.B);
    unsigned pos = 0u;
    if (HELPERS && e < 0) {  // a reward helper
        const unsigned hb = blockIdx.x, hpe = (unsigned)p.term_hpe, k = hb / hpe;  // hpe = REWARD_PARTS (+ 1: the feature helper)
        part = (int)(hb - k * hpe);
        const unsigned ring = p.seq & 3u, cps = (unsigned)p.term_cap / TERM_SHARDS, shard = k & (TERM_SHARDS - 1u), idx = k >> TERM_SHARD_BITS;
        const unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane((int)load_agent(p.term_cnt + (ring * TERM_SHARDS + shard) * TERM_CNT_STRIDE));
        if (idx >= cps || idx >= cnt) return;
        pos = k;
        // (wave-uniform values the compiler cannot see as such: kept in scalar registers, like blockIdx.x)
        e = __builtin_amdgcn_readfirstlane(p.term_list[ring * (unsigned)p.term_cap + pos]);
        if ((unsigned)e >= (unsigned)p.B) return;  // (never in a list this library wrote; an index is checked before it addresses memory all the same)
        role = part < REWARD_PARTS ? ROLE_REWARD : ROLE_FEATURES;
    } else if (p.term_cap > 0 && e == 0 && threadIdx.x < TERM_SHARDS) {
        // List bookkeeping, by the first environment workgroup: the ring after next starts empty, and the host learns how
        // long the lists are (any later launch may read it, whenever: it only sizes helper grids).  Loads first, stores
        // last: on gfx9 a wavefront's loads and stores retire through one in-order counter, so a load issued behind a
        // store waits for that store's acknowledgement -- microseconds while the chip is saturated with stores, and this
        // wavefront has a whole transition to do afterwards (0.3 us per launch on the lock-step loop).
        unsigned *hist = p.term_cnt + 4u * TERM_SHARDS * TERM_CNT_STRIDE;  // behind the counters: four launches' figures, then the one the host was last told
        unsigned longest = load_agent(p.term_cnt + (((p.seq & 3u) * TERM_SHARDS + threadIdx.x) * TERM_CNT_STRIDE));
        const unsigned h1 = hist[(p.seq + 1u) & 3u], h2 = hist[(p.seq + 2u) & 3u], h3 = hist[(p.seq + 3u) & 3u], told = hist[4];
        for (int o = TERM_SHARDS / 2; o > 0; o >>= 1) longest = max(longest, (unsigned)__shfl_xor((int)longest, o, TERM_SHARDS));
        // ... the SHORTEST of the last four launches' longest shards: with episodes in lock-step the list is full once per
        // episode and empty otherwise -- sized on that one launch, the launches that follow would each start ~2 000 idle
        // helper workgroups (+ 4 % on the lock-step loop) -- while staggered phases give steady lengths, which the minimum
        // tracks as well.  (Kept per launch on the device: the host reads whenever it enqueues, many times per launch or
        // once in many.)
        const unsigned least = min(min(longest, h1), min(h2, h3));
        store_agent(p.term_cnt + ((((p.seq + 2u) & 3u) * TERM_SHARDS + threadIdx.x) * TERM_CNT_STRIDE), 0u);
        if (threadIdx.x == 0) {
            hist[p.seq & 3u] = longest;
            // (a store to host memory: only when the figure has grown, or shrunk by more than an eighth + 2)
            if (least > told || least + (told >> 3) + 2u < told) {
                hist[4] = least;
                __hip_atomic_store(p.term_seen, least, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
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
