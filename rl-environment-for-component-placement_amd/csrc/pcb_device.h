// pcb_device.h -- device-side parameter block, state-block records, LDS barrier, bit rows, 16-byte plane emission, wave scan
// Part of libpcbenv.so's single translation unit (included by pcbenv_kernels.hip); CDNA4 / gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "pcbenv.h"

typedef unsigned long long u64;

#define WAVE 64
#define MAX_NT 256
#define HDR_BYTES 64
#define TERM_CNT_STRIDE 32u  // unsigned words between the shard counters of the terminal list: one 128-byte line each
#define TERM_SHARD_BITS 4
#define TERM_SHARDS (1u << TERM_SHARD_BITS)

// ----------------------------------------------------------------------------------------------
// device-side parameter block (kernel argument, by value)
// ----------------------------------------------------------------------------------------------
struct DevParams {
    int kind, H, W, WW, O, C, P, N, K, mp, mh, mw, F, pinRows, catW, B, Q;
    int reward_type, beam_width, component_n;
    unsigned flags, bind_gen;
    double w_wl, w_int, max_wl, max_int, wl_norm, int_norm, area;
    long long stateStride, instStride;
    int offOcc, offVm, offComps, offPins, offRank;   // byte offsets inside a state block
    int ldsHf, ldsCls, ldsSeg, ldsBytes;    // byte offsets of LDS scratch behind the state mirror
    int ldsHfWords;                         // 64-bit words of the fold scratch / row-membership bit map at ldsHf (sized by need)
    unsigned char *state, *queue;
    pcbenv_buffers buf;
    pcbenv_compact_features cbuf;           // compact feature tensors of the trajectory layout (all null unless bound)
    // episode-constant observation bytes per environment (spatial, trajectory layout; pcb_observe.h feat_cache_*)
    unsigned char *feat_cache; unsigned *feat_cache_tag; int featCacheStride, featCacheCg;
    unsigned long long *dbg;                // diagnostic build only (-DPCBENV_STAMPS): [B][32] s_memtime stamps
    int stream_stores;                      // observation stores bypass the caches (`nt`): see STORE16.  (Behind the
                                            // fields every wave loads first, so that their kernarg offsets stay put.)
    // Trajectory layout (pcbenv_bind_buffers_slots): every bound tensor is [num_slots, B, ...]; a launch writes its
    // outputs into `slot` (the persistent rollout kernel: one slot per step).  With num_slots > 1 nothing may rely on
    // what an earlier step left in the destination, so the float64 feature tensors are written whole every step.
    int num_slots, slot;
    // on-device instance generator (pcb_geninst.h), null when the host feeds the queue: records generated so far per
    // environment, and a sticky error word a reset raises if it ever finds its record missing (it never should:
    // the host-side bookkeeping of pcbenv_kernels.hip orders every fill before the launches that can consume it)
    unsigned *gen_produced, *gen_errors;
    // Queue cursor of every environment, published with agent-scope (write-through) stores at each reset.  The copy
    // in the state block is written back lazily and only ever re-read on the environment's own XCD; a kernel on
    // another stream (k_gen_fill) may run on any XCD, whose L2 is not coherent with the writer's.
    unsigned *cursor_pub;
    // Terminal list (Team<>::run_env has the story): `seq` numbers the step launches of this handle; launch seq starts
    // REWARD_PARTS helper teams for each of the first term_wgs entries of ring seq & 3, appends to ring (seq + 1) & 3 and
    // clears the counters of ring (seq + 2) & 3.  A ring is TERM_SHARDS shards of term_cap / TERM_SHARDS entries, each with
    // a counter on a line of its own (term_cnt[(ring * TERM_SHARDS + shard) * TERM_CNT_STRIDE]); entry (shard, idx) has
    // the number idx * TERM_SHARDS + shard.  term_cap == 0: no lists are kept; term_wgs == 0: this launch has no helpers.
    unsigned seq;
    int term_wgs, term_cap, term_hpe;  // term_hpe = helper teams per entry: REWARD_PARTS, + 1 feature helper with PCBENV_FLAG_AUTO_RESET
    int *term_list;
    unsigned *term_cnt;
    u64 *term_arrive;  // [term_cap]: where the shares of a routing reward meet (terminal_reward)
    unsigned *term_seen;  // host memory: the longest shard of the latest launch's list (sizes later helper grids)
    unsigned char *state_out;  // the state blocks this launch writes (p.state: the ones it reads); equal for in-place kernels
};
// Loads of data another stream's kernel (or a DMA) has written since this XCD last read the same addresses: instance
// records and the generator's counters.  Agent-scope loads (`sc1`) are served coherently; a plain load may hit a
// stale clean line in this XCD's L2.
__device__ inline u64 load_agent(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned load_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int load_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void store_agent(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// In-kernel stamps (cdna_hip_programming.md §7): only in a separate diagnostic build, written to a buffer nothing
// else reads; `PCBENV_STAMPS=1` in the environment allocates it, tools/kernel_stamps.py prints the phase profile.
#ifdef PCBENV_STAMPS
#define STAMP(k) do { if ((threadIdx.x & (NT - 1)) == 0 && p.dbg) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.dbg[(size_t)blockIdx.x * 32 + (k)] = t_; } } while (0)
#define STAMP_RT(k) do { if ((threadIdx.x & (NT - 1)) == 0 && p.dbg) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.dbg[(size_t)blockIdx.x * 32 + (k)] = t_; } } while (0)
// accumulate elapsed shader cycles of a region / an arbitrary value into slot k (the kernel's first STAMP must zero it)
#define STAMP_T0() unsigned long long st0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st0_) :: "memory")
#define STAMP_ACC_SINCE(k, dep) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory"); if ((threadIdx.x & (NT - 1)) == 0 && p.dbg) p.dbg[(size_t)blockIdx.x * 32 + (k)] += t_ - st0_; } while (0)
#define STAMP_ADD(k, v) do { if ((threadIdx.x & (NT - 1)) == 0 && p.dbg) p.dbg[(size_t)blockIdx.x * 32 + (k)] += (unsigned long long)(v); } while (0)
#define STAMP_ZERO(k) do { if ((threadIdx.x & (NT - 1)) == 0 && p.dbg) p.dbg[(size_t)blockIdx.x * 32 + (k)] = 0ull; } while (0)
// the stamps above index their rows by blockIdx.x: shift the table so that this team's rows are environment e's
#define STAMP_ROWS_BY_ENV(launch, e) DevParams p = (launch); if (p.dbg) p.dbg += ((long long)(e) - (long long)blockIdx.x) * 32
#else
#define STAMP(k) do { } while (0)
#define STAMP_RT(k) do { } while (0)
#define STAMP_T0() do { } while (0)
#define STAMP_ACC_SINCE(k, dep) do { } while (0)
#define STAMP_ADD(k, v) do { } while (0)
#define STAMP_ZERO(k) do { } while (0)
#define STAMP_ROWS_BY_ENV(launch, e) const DevParams &p = (launch)
#endif

// per-environment header at the start of a state block
struct __attribute__((aligned(16))) EnvHdr {
    short ncomp, nnets, npins, cur;  // cur = index of the current component, -1 = sentinel (all placed)
    unsigned episode;                // completed resets
    unsigned qcursor;                // next queue slot
    unsigned flag;                   // LDS scratch word: workgroup-wide any(), and y of the sampled action
    unsigned pad[2];                 // LDS scratch: (o, x) of the action drawn by wavefront 0 (fused sampler)
    unsigned feat_gen;               // bind generation for which the pin-feature tensors hold only this env's rows
    // Action of the NEXT fused-sampler step, drawn at the end of the launch that produced the mask (while its
    // stores drain) instead of at the head of the next launch, where the whole grid would wait for it.  Valid
    // (bit 31 of pre_action) only for exactly this (seed, step index, global env index) and only while vm is the
    // mask it was drawn from: every launch that rewrites vm redraws or clears it.
    u64 pre_seed, pre_step;
    unsigned pre_action;             // o | x << 8 | y << 16 | 1 << 31
    unsigned pre_genv;
    // Terminal list (run_env): the launch number for which this environment sits on the list of environments that are
    // certain to end their episode, and its entry there; anything else = not listed.
    unsigned term_seq, term_pos;
};
static_assert(sizeof(EnvHdr) == HDR_BYTES, "header size");

// 8-byte records (state block and instance wire format share the pin layout up to abs_x/abs_y)
struct CompRec { unsigned char h, w; signed char px, py; unsigned char o, pad[3]; };  // o = orientation it was placed with
struct PinRec { unsigned char rel_x, rel_y; signed char abs_x, abs_y; unsigned char net, comp; unsigned short id; };
#define PIN_ID_MASK 0x7FFF
#define PIN_LOSER 0x8000  // pin env quirk Q1: a later pin of the same component shares this feature row

// ----------------------------------------------------------------------------------------------
// bit rows
// ----------------------------------------------------------------------------------------------
template <int WW> struct Row;
template <> struct Row<1> {
    u64 a;
    __device__ static Row load(const u64 *p) { return Row{p[0]}; }
    __device__ void store(u64 *p) const { p[0] = a; }
    __device__ Row operator|(Row o) const { return Row{a | o.a}; }
    __device__ Row shr(int k) const { return Row{k >= 64 ? 0ull : a >> k}; }
    __device__ static Row zero() { return Row{0ull}; }
    __device__ bool any() const { return a != 0; }
    // valid = ~occ restricted to columns [0, n)
    __device__ Row free_below(int n) const { return Row{n <= 0 ? 0ull : (~a & (n >= 64 ? ~0ull : ((1ull << n) - 1ull)))}; }
};
template <> struct Row<2> {
    u64 a, b;
    __device__ static Row load(const u64 *p) { return Row{p[0], p[1]}; }
    __device__ void store(u64 *p) const { p[0] = a; p[1] = b; }
    __device__ Row operator|(Row o) const { return Row{a | o.a, b | o.b}; }
    __device__ Row shr(int k) const {
        if (k == 0) return *this;
        if (k >= 128) return Row{0ull, 0ull};
        if (k >= 64) return Row{b >> (k - 64), 0ull};
        return Row{(a >> k) | (b << (64 - k)), b >> k};
    }
    __device__ static Row zero() { return Row{0ull, 0ull}; }
    __device__ bool any() const { return (a | b) != 0; }
    __device__ Row free_below(int n) const {
        u64 ma = n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1ull));
        u64 mb = n <= 64 ? 0ull : (n >= 128 ? ~0ull : ((1ull << (n - 64)) - 1ull));
        return Row{~a & ma, ~b & mb};
    }
};

// OR_{k < pw} (row >> k): bit j set iff some cell j..j+pw-1 of the row is occupied (log-step doubling).
template <int WW> __device__ inline Row<WW> hfold(Row<WW> r, int pw) {
    Row<WW> f = r;
    int s = 1;
    while (2 * s <= pw) { f = f | f.shr(s); s *= 2; }
    if (s < pw) f = f | f.shr(pw - s);
    return f;
}

// 16-byte observation store, agent-scope write-through (`sc1`): every line written here is next read by another
// launch (usually on another XCD) or by the policy, never by this workgroup, so leaving it dirty in the XCD's L2
// only defers the write to the end-of-kernel release, where the whole grid waits for it (+5 % at c3).
// STREAM adds `nt`: when one launch writes well beyond the 256 MiB Infinity Cache, lines allocated there are
// evicted before anything reads them and only cost fabric traffic (c5, 3.1 GB per launch: +10..20 %; c3 at 65 536
// environments: 192 M -> 271 M env-steps/s), while below that size the cache absorbs the burst and streaming is the
// slower choice (c3 / c4 at 4 096 environments: -6 %).  pcbenv_create picks by bytes per launch (DevParams).
// The stores are raw buffer stores (buffer_store_dwordx4 ... sc1 [nt]) through a per-environment resource
// descriptor: the cache policy travels in the builtin's aux operand, so the compiler schedules them (and their
// gfx950 store-data wait states) itself, and the descriptor's byte count makes an out-of-range chunk a dropped
// store instead of a fault.  -DPCBENV_STORE_PLAIN keeps plain global stores for A/B runs.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#if defined(PCBENV_STORE_PLAIN)
struct ObsDst { unsigned char *base; };
__device__ inline ObsDst obs_dst(unsigned char *base, long long) { return ObsDst{base}; }
template <bool STREAM> __device__ inline void STORE16(const ObsDst &d, unsigned off, uint4 v) { *(uint4 *)(d.base + off) = v; }
#else
struct ObsDst { __amdgpu_buffer_rsrc_t rsrc; };
// base must be wave-uniform (it is: tensor pointer + blockIdx.x * per-environment bytes)
__device__ inline ObsDst obs_dst(unsigned char *base, long long bytes) {
    // The descriptor has to sit in scalar registers.  Where the compiler cannot prove base / bytes uniform it wraps every
    // store in a "waterfall" loop (readfirstlane x 4, compare, saveexec, branch -- per store); they ARE uniform, say so.
    const uintptr_t b = (uintptr_t)base;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    unsigned char *ub = (unsigned char *)(((uintptr_t)hi << 32) | lo);
    return ObsDst{__builtin_amdgcn_make_buffer_rsrc(ub, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000)};
}
#define PCBENV_AUX_SC1 16
#define PCBENV_AUX_NT 2
template <bool STREAM> __device__ inline void STORE16(const ObsDst &d, unsigned off, uint4 v) {
    const v4u w{v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(w, d.rsrc, (int)off, 0, STREAM ? (PCBENV_AUX_SC1 | PCBENV_AUX_NT) : PCBENV_AUX_SC1);
}
#endif

// 4 mask bits -> 4 bytes of 0/1
__device__ inline unsigned expand4(unsigned b) { return (b * 0x00204081u) & 0x01010101u; }
__device__ inline uint4 expand16(unsigned bits) {
    return make_uint4(expand4(bits & 15u), expand4((bits >> 4) & 15u), expand4((bits >> 8) & 15u), expand4((bits >> 12) & 15u));
}

__device__ inline void STORE16_dyn(const ObsDst &d, unsigned off, uint4 v, bool stream) { if (stream) STORE16<true>(d, off, v); else STORE16<false>(d, off, v); }
// value of `a` held by lane + s (0 beyond the wavefront): one cross-lane read per 32-bit half
__device__ inline u64 lane_down(u64 a, int s, int lane) {
    const int src = (lane + s) << 2;
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)(a >> 32));
    return lane + s < WAVE ? (((u64)hi << 32) | lo) : 0ull;
}
// Inclusive prefix sum over the 64 lanes with DPP row shifts / row broadcasts (no LDS round trips).
__device__ inline int wave_inclusive_scan(int x, int lane) {
    const int row = lane & 15;
    int t;
    t = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false); if (row >= 1) x += t;   // row_shr:1
    t = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false); if (row >= 2) x += t;   // row_shr:2
    t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false); if (row >= 4) x += t;   // row_shr:4
    t = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false); if (row >= 8) x += t;   // row_shr:8
    t = __builtin_amdgcn_update_dpp(0, x, 0x142, 0xF, 0xF, false); if ((lane & 31) >= 16) x += t;  // row_bcast:15
    t = __builtin_amdgcn_update_dpp(0, x, 0x143, 0xF, 0xF, false); if (lane >= 32) x += t;         // row_bcast:31
    return x;
}

