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

#include "pcbenv.h"

typedef unsigned long long u64;

#define WAVE 64
#define NT ((int)blockDim.x)   // threads per environment: 64 (one wave) or 256 (four waves, large grids)
#define MAX_NT 256
#define HDR_BYTES 64

// ----------------------------------------------------------------------------------------------
// device-side parameter block (kernel argument, by value)
// ----------------------------------------------------------------------------------------------
struct DevParams {
    int kind, H, W, WW, O, C, P, N, K, mp, mh, mw, F, pinRows, catW, B, Q;
    int reward_type, beam_width, component_n;
    unsigned flags, bind_gen;
    double w_wl, w_int, max_wl, max_int, wl_norm, int_norm, area;
    long long stateStride, instStride;
    int offOcc, offVm, offComps, offPins;   // byte offsets inside a state block
    int ldsHf, ldsCls, ldsSeg, ldsBytes;    // byte offsets of LDS scratch behind the state mirror
    unsigned char *state, *queue;
    pcbenv_buffers buf;
    unsigned long long *dbg;                // diagnostic build only (-DPCBENV_STAMPS): [B][32] s_memtime stamps
};
// In-kernel stamps (cdna_hip_programming.md §7): only in a separate diagnostic build, written to a buffer nothing
// else reads; `PCBENV_STAMPS=1` in the environment allocates it, tools/kernel_stamps.py prints the phase profile.
#ifdef PCBENV_STAMPS
#define STAMP(k) do { if (threadIdx.x == 0 && p.dbg) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.dbg[(size_t)blockIdx.x * 32 + (k)] = t_; } } while (0)
#define STAMP_RT(k) do { if (threadIdx.x == 0 && p.dbg) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.dbg[(size_t)blockIdx.x * 32 + (k)] = t_; } } while (0)
#else
#define STAMP(k) do { } while (0)
#define STAMP_RT(k) do { } while (0)
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
    unsigned rsv[2];
};
static_assert(sizeof(EnvHdr) == HDR_BYTES, "header size");

// 8-byte records (state block and instance wire format share the pin layout up to abs_x/abs_y)
struct CompRec { unsigned char h, w; signed char px, py; unsigned char pad[4]; };
struct PinRec { unsigned char rel_x, rel_y; signed char abs_x, abs_y; unsigned char net, comp; unsigned short id; };
#define PIN_ID_MASK 0x7FFF
#define PIN_LOSER 0x8000  // pin env quirk Q1: a later pin of the same component shares this feature row

// Workgroup barrier that waits for LDS traffic only.  __syncthreads() also drains the global stores in flight
// (s_waitcnt vmcnt(0)), which would serialise the observation write stream between kernel phases.
__device__ inline void lds_sync() {
#ifdef PCBENV_FULL_SYNC
    __syncthreads();
    return;
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// any() over the workgroup; `flag` is an LDS word
__device__ inline bool block_any(bool v, unsigned *flag) {
    if (NT == WAVE) return __any(v);
    if (threadIdx.x == 0) *flag = 0;
    lds_sync();
    if (__any(v) && (threadIdx.x & 63) == 0) *flag = 1;
    lds_sync();
    return *flag != 0;
}

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

// 16-byte observation / state store, agent-scope write-through (`sc1`): every line written here is next read by
// another launch (usually on another XCD) or by the policy, never by this workgroup, so leaving it dirty in the
// XCD's L2 only defers the write to the end-of-kernel release, where the whole grid waits for it (+5 % at c3).
// -DPCBENV_STORE_PLAIN / -DPCBENV_NT_STORES / -DPCBENV_STORE_ASM="..." keep the alternatives for A/B runs.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define PCB_STR_(x) #x
#define PCB_STR(x) PCB_STR_(x)
#if defined(PCBENV_STORE_PLAIN)
__device__ inline void STORE16(uint4 *p, uint4 v) { *p = v; }
#elif defined(PCBENV_NT_STORES)
__device__ inline void STORE16(uint4 *p, uint4 v) { __builtin_nontemporal_store(v4u{v.x, v.y, v.z, v.w}, (v4u *)p); }
#else
#ifndef PCBENV_STORE_ASM
#define PCBENV_STORE_ASM sc1
#endif
__device__ inline void STORE16(uint4 *p, uint4 v) {
    v4u w{v.x, v.y, v.z, v.w};
    // s_nop: a store wider than 64 bits may read its data VGPRs up to two wait states after issue (gfx940+ VMEM store-data
    // hazard); the compiler pads that for its own stores but cannot see into this statement.
    asm volatile("global_store_dwordx4 %0, %1, off " PCB_STR(PCBENV_STORE_ASM) "\n\ts_nop 1" :: "v"(p), "v"(w) : "memory");
}
#endif

// 4 mask bits -> 4 bytes of 0/1
__device__ inline unsigned expand4(unsigned b) { return (b * 0x00204081u) & 0x01010101u; }
__device__ inline uint4 expand16(unsigned bits) {
    return make_uint4(expand4(bits & 15u), expand4((bits >> 4) & 15u), expand4((bits >> 8) & 15u), expand4((bits >> 12) & 15u));
}

// Write one H x W uint8 plane (0/1) from bit rows in LDS: 16 bytes per lane, 1 KiB per wave instruction.
// Rows [r0, r1) only (full plane: 0, H).
template <int WW> __device__ inline void emit_plane(unsigned char *dst, const u64 *bits, int r0, int r1, int W, int lane) {
    if ((W & 15) == 0 && (((uintptr_t)dst) & 15) == 0) {
        uint4 *d4 = (uint4 *)dst;
        const int sh = (W & (W - 1)) == 0 ? __ffs(W) - 1 : -1;
        for (int c = r0 * W / 16 + lane; c < r1 * W / 16; c += NT) {
            int cell = c * 16, r = sh >= 0 ? cell >> sh : cell / W, col = cell - r * W;
            unsigned b = (unsigned)(bits[r * WW + (col >> 6)] >> (col & 63)) & 0xFFFFu;
            STORE16(d4 + c, expand16(b));
        }
    } else {  // odd widths (the reference's small test grids): byte path
        for (int i = r0 * W + lane; i < r1 * W; i += NT) {
            int r = i / W, col = i - r * W;
            dst[i] = (unsigned char)((bits[r * WW + (col >> 6)] >> (col & 63)) & 1ull);
        }
    }
}
__device__ inline void emit_zero(unsigned char *dst, long long bytes, int lane) {
    if ((bytes & 15) == 0 && (((uintptr_t)dst) & 15) == 0) {
        uint4 *d4 = (uint4 *)dst;
        for (long long c = lane; c < bytes / 16; c += NT) STORE16(d4 + c, make_uint4(0, 0, 0, 0));
    } else {
        for (long long i = lane; i < bytes; i += NT) dst[i] = 0;
    }
}

// Legal-placement bit mask for a ph x pw window (R:526-567, S:1792-1835):
// vm[r] bit j = 1 iff r <= H-ph and j <= W-pw and occ[r..r+ph-1][j..j+pw-1] is empty.
// Returns (wave-uniform) whether any bit is set.
template <int WW>
__device__ inline bool window_mask(const u64 *occ, u64 *hf, u64 *vm, int H, int W, int ph, int pw, int lane, unsigned *flag) {
    for (int r = lane; r < H; r += NT) hfold<WW>(Row<WW>::load(occ + r * WW), pw).store(hf + r * WW);
    lds_sync();
    bool any = false;
    for (int r = lane; r < H; r += NT) {
        Row<WW> v = Row<WW>::zero();
        if (r + ph <= H) {
            Row<WW> acc = Row<WW>::load(hf + r * WW);
            for (int k = 1; k < ph; k++) acc = acc | Row<WW>::load(hf + (r + k) * WW);
            v = acc.free_below(W - pw + 1);
        }
        v.store(vm + r * WW);
        any |= v.any();
    }
    return block_any(any, flag);
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

// ----------------------------------------------------------------------------------------------
// float64 geometry of the reward (one IEEE operation per operator, see file header)
// ----------------------------------------------------------------------------------------------
// S:1288-1301 euclidean_distance == np.linalg.norm == sqrt(ddot): sqrt(fma(dy, dy, dx*dx))
__device__ inline double norm2(double dx, double dy) { return __dsqrt_rn(__fma_rn(dy, dy, __dmul_rn(dx, dx))); }

// S:653-702 is_intersect
__device__ inline bool is_intersect(double x1, double y1, double x2, double y2, double x3, double y3, double x4, double y4) {
    if ((x1 == x3 && y1 == y3) || (x1 == x4 && y1 == y4) || (x2 == x3 && y2 == y3) || (x2 == x4 && y2 == y4)) return true;
    double det = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
    if (det == 0) return false;
    double a = x1 * y2 - y1 * x2, b = x3 * y4 - y3 * x4;
    double x = (a * (x3 - x4) - (x1 - x2) * b) / det;
    double y = (a * (y3 - y4) - (y1 - y2) * b) / det;
    return fmin(x1, x2) <= x && x <= fmax(x1, x2) && fmin(x3, x4) <= x && x <= fmax(x3, x4) &&
           fmin(y1, y2) <= y && y <= fmax(y1, y2) && fmin(y3, y4) <= y && y <= fmax(y3, y4);
}

// ---- routes -------------------------------------------------------------------------------------
// A route is kept as one segment slot per pin q (slots of net n are nstart[n]..nstart[n+1]-1, so slots are
// net-major like the reference's route lists); act[q] = 1 if the slot carries a segment.
struct SegView { double *X1, *Y1, *X2, *Y2, *D, *A, *DX, *DY, *cen; int *act, *nstart; unsigned *bbox; unsigned short *ns, *pairs; unsigned char *beam; };
// compaction buffer of candidate (i, j) pairs: 1024 entries for a one-wavefront workgroup, 512 per wavefront for four
#define PAIR_ENTRIES(NW) ((NW) == 1 ? 1024 : 2048)
// [segments X1 Y1 X2 Y2 D | centroids | act nstart] then a zone used only by the pair count (A DX DY bbox ns pairs),
// which the beam search -- finished before the count starts -- overlays with its per-net scratch.
#define SEG_FIXED_BYTES(P) ((5 * (P) + 2 * PCBENV_MAX_NETS) * 8 + ((P) + PCBENV_MAX_NETS + 4) * 4)
#define SEG_COUNT_BYTES(P, NW) (3 * (P) * 8 + (P) * 4 + (((P) + 1) & ~1) * 2 + PAIR_ENTRIES(NW) * 2)
#define SEG_LDS_BYTES(P, NW, beam) (((SEG_FIXED_BYTES(P) + 7) & ~7) + ((beam) > SEG_COUNT_BYTES(P, NW) ? (beam) : SEG_COUNT_BYTES(P, NW)))
__device__ inline SegView seg_view(double *seg, int P) {
    SegView v;
    v.X1 = seg; v.Y1 = seg + P; v.X2 = seg + 2 * P; v.Y2 = seg + 3 * P; v.D = seg + 4 * P;
    v.cen = seg + 5 * P;                              // cx[MAX_NETS], cy[MAX_NETS]
    v.act = (int *)(v.cen + 2 * PCBENV_MAX_NETS);     // [P]
    v.nstart = v.act + P;                             // [nnets + 1] (+ spare counter slot)
    v.beam = (unsigned char *)seg + ((SEG_FIXED_BYTES(P) + 7) & ~7);
    v.A = (double *)v.beam; v.DX = v.A + P; v.DY = v.A + 2 * P;  // per segment: x1*y2 - y1*x2, x1 - x2, y1 - y2
    v.bbox = (unsigned *)(v.A + 3 * P);               // [P] integer extents (x_lo, x_hi, y_lo, y_hi), one byte each
    v.ns = (unsigned short *)(v.bbox + P);            // [P] first slot of the slot's own net (= number of earlier-net slots)
    v.pairs = v.ns + ((P + 1) & ~1);                  // [PAIR_ENTRIES] shared out among the wavefronts
    return v;
}

// net_pins offsets (self.pins is net-major) and S:1229-1241 get_centroid per net (exact integer sums, one division)
__device__ inline void net_offsets_and_centroids(const SegView &v, const EnvHdr *hdr, const PinRec *pins, int lane) {
    const int np = hdr->npins, nn = hdr->nnets;
    lds_sync();  // the segment area aliases the class map of emit_pin_grid
    for (int q = lane; q < np; q += NT)
        if (q == 0 || pins[q].net != pins[q - 1].net) v.nstart[pins[q].net] = q;
    if (lane == 0) v.nstart[nn] = np;
    lds_sync();
    for (int n = lane; n < nn; n += NT) {
        const int s = v.nstart[n], e = v.nstart[n + 1];
        double sx = 0, sy = 0;
        for (int q = s; q < e; q++) { sx += (double)pins[q].abs_x; sy += (double)pins[q].abs_y; }
        v.cen[n] = sx / (double)(e - s);
        v.cen[PCBENV_MAX_NETS + n] = sy / (double)(e - s);
    }
    lds_sync();
}

// S:1243-1271 route_pins_centroid: (pin, centroid) per pin; a 2-pin net is the single segment (p0, p1)
__device__ inline void build_centroid_segments(const SegView &v, const EnvHdr *hdr, const PinRec *pins, int lane) {
    const int np = hdr->npins;
    for (int q = lane; q < np; q += NT) {
        const int n = pins[q].net, s = v.nstart[n], cnt = v.nstart[n + 1] - s;
        double x1 = pins[q].abs_x, y1 = pins[q].abs_y, x2, y2;
        int a = 1;
        if (cnt == 2) { a = (q == s); x2 = pins[s + 1].abs_x; y2 = pins[s + 1].abs_y; }
        else { x2 = v.cen[n]; y2 = v.cen[PCBENV_MAX_NETS + n]; }
        v.X1[q] = x1; v.Y1[q] = y1; v.X2[q] = x2; v.Y2[q] = y2; v.act[q] = a;
        v.D[q] = norm2(x1 - x2, y1 - y2);
    }
    lds_sync();
}

// is_intersect (S:653-702) on two slots, with the per-segment terms hoisted: the operations and their order are
// exactly the reference's -- (x1*y2 - y1*x2), (x1 - x2), (y1 - y2) are sub-expressions of its formulas.
// Written without branches so that several candidates per lane can be in flight at once (the count is bound by
// the LDS and float64 division latency of one wavefront, not by issue slots): det == 0 gives inf / NaN
// coordinates, which is harmless and masked by the explicit test.
__device__ inline bool slots_intersect(const SegView &v, int i, int j) {
    const double x1 = v.X1[i], y1 = v.Y1[i], x2 = v.X2[i], y2 = v.Y2[i];
    const double x3 = v.X1[j], y3 = v.Y1[j], x4 = v.X2[j], y4 = v.Y2[j];
    const double dxi = v.DX[i], dyi = v.DY[i], dxj = v.DX[j], dyj = v.DY[j];
    const double a = v.A[i], b = v.A[j];
    const bool shared = ((x1 == x3) & (y1 == y3)) | ((x1 == x4) & (y1 == y4)) | ((x2 == x3) & (y2 == y3)) | ((x2 == x4) & (y2 == y4));
    const double det = dxi * dyj - dyi * dxj;
    const double x = (a * dxj - dxi * b) / det;
    const double y = (a * dyj - dyi * b) / det;
    const bool inside = (fmin(x1, x2) <= x) & (x <= fmax(x1, x2)) & (fmin(x3, x4) <= x) & (x <= fmax(x3, x4)) &
                        (fmin(y1, y2) <= y) & (y <= fmax(y1, y2)) & (fmin(y3, y4) <= y) & (y <= fmax(y3, y4));
    return shared | ((det != 0) & inside);
}
// Exact pre-filter: if the closed x- (or y-) extents of the two segments are disjoint, no x (y) can lie in both,
// so the reference's final range test fails whatever the computed intersection point is (a shared end point,
// its only early "True", puts a common point in both extents).  The extents are kept as conservatively rounded
// integers (floor of the minimum, ceil of the maximum; coordinates are in [0, 127]), four bytes per segment, so
// the filter is one LDS word per segment and a few integer compares; a pair it lets through is decided by the
// full float64 test, a pair it rejects has disjoint real extents.  Saves the two float64 divisions.
__device__ inline unsigned pack_extents(double x1, double y1, double x2, double y2) {
    const unsigned xl = (unsigned)floor(fmin(x1, x2)), xh = (unsigned)ceil(fmax(x1, x2));
    const unsigned yl = (unsigned)floor(fmin(y1, y2)), yh = (unsigned)ceil(fmax(y1, y2));
    return xl | (xh << 8) | (yl << 16) | (yh << 24) | 0x80000000u;  // bit 31 = slot carries a segment
}
__device__ inline bool extents_overlap(unsigned a, unsigned b) {  // branch-free
    const unsigned xl = max(a & 0xFFu, b & 0xFFu), xh = min((a >> 8) & 0xFFu, (b >> 8) & 0xFFu);
    const unsigned yl = max((a >> 16) & 0xFFu, (b >> 16) & 0xFFu), yh = min((a >> 24) & 0x7Fu, (b >> 24) & 0x7Fu);
    return ((a & b & 0x80000000u) != 0) & (xl <= xh) & (yl <= yh);
}

// Full test on the n candidates a wavefront has collected, two per lane and step so that their LDS reads and
// divisions overlap.
typedef __attribute__((address_space(3))) unsigned short lds_u16;  // keeps the buffer accesses ds_* instead of flat_*
__device__ inline int count_candidates(const SegView &v, const volatile lds_u16 *buf, int n, int wl_lane) {
    int cnt = 0;
    for (int base = 0; base < n; base += 2 * WAVE) {
        const int i0 = base + wl_lane, i1 = i0 + WAVE;
        const unsigned short p0 = i0 < n ? buf[i0] : (unsigned short)0, p1 = i1 < n ? buf[i1] : (unsigned short)0;
        const bool r0 = slots_intersect(v, p0 & 0xFF, p0 >> 8), r1 = slots_intersect(v, p1 & 0xFF, p1 >> 8);
        cnt += ((i0 < n) & r0) + ((i1 < n) & r1);
    }
    return cnt;
}

// S:629-651 find_num_intersection + S:704-722 find_wirelength over the slots.  The (segment, later-net segment)
// pairs are first filtered by extent overlap, the survivors compacted into an LDS buffer and run through the full
// test in dense batches (see count_finish).  The wirelength
// is summed sequentially in route order (bit-exact with the reference's python float loop).
// count_prepare reads the pins, count_finish only the segment zone.
__device__ inline void count_prepare(const SegView &v, int np, const PinRec *pins, int lane) {
    int *total_cnt = v.nstart + PCBENV_MAX_NETS + 1;  // spare slot behind nstart[0..MAX_NETS]
    for (int q = lane; q < np; q += NT) {
        const double x1 = v.X1[q], y1 = v.Y1[q], x2 = v.X2[q], y2 = v.Y2[q];
        v.A[q] = x1 * y2 - y1 * x2; v.DX[q] = x1 - x2; v.DY[q] = y1 - y2;
        v.bbox[q] = v.act[q] ? pack_extents(x1, y1, x2, y2) : 0u;
        v.ns[q] = (unsigned short)v.nstart[pins[q].net];
    }
    if (lane == 0) *total_cnt = 0;
    lds_sync();
}
__device__ inline void count_finish(const DevParams &p, const SegView &v, int np_, int lane, double *wirelength, int *nintersections) {
    int *total_cnt = v.nstart + PCBENV_MAX_NETS + 1;
    const int np = __builtin_amdgcn_readfirstlane(np_);
    STAMP(12);
    const int wl_lane = lane & 63, wave = lane >> 6, nwaves = NT / WAVE;
    const int cap = PAIR_ENTRIES(nwaves) / nwaves;
    volatile lds_u16 *buf = (volatile lds_u16 *)(v.pairs + wave * cap);  // wave-synchronous: written and read by different lanes
    // Slots are net-major, so the partners "segment of an earlier net" of slot j are the slots i < ns[j].  The work
    // is cut into 64 x 64 tiles (j chunk, i chunk <= j chunk) dealt out to the wavefronts.  In a tile lane j keeps
    // its packed extents in a register and the wavefront sweeps the i chunk: one broadcast LDS word per step, no
    // dependent reads; the survivors of a step are appended to the compaction buffer with a ballot.  That leaves the
    // buffer i-major with ascending j, so a dense batch reads the i side as broadcasts and the j side from
    // consecutive addresses.  The buffer is run through the full test whenever another step might not fit.
    const int nchunk = (np + WAVE - 1) / WAVE, ntiles = nchunk * (nchunk + 1) / 2;
    int cnt = 0, nbuf = 0;
    for (int tile = wave; tile < ntiles; tile += nwaves) {  // wave-uniform
        int jc = 0, ic = tile;
        while (ic > jc) { ic -= jc + 1; jc++; }
        const int j = WAVE * jc + wl_lane;
        const unsigned bj = j < np ? v.bbox[j] : 0u;
        const int lim = j < np ? (int)v.ns[j] : 0;
        // ns grows with j, so the last slot of the chunk bounds the sweep; readfirstlane keeps the trip count in an SGPR
        const int i0 = WAVE * ic;
        const int i1 = __builtin_amdgcn_readfirstlane(min(i0 + WAVE, (int)v.ns[min(np - 1, WAVE * jc + WAVE - 1)]));
        for (int ib = i0; ib < i1; ib += 4) {
            unsigned bi[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) bi[u] = v.bbox[min(ib + u, i1 - 1)];  // the four reads go out together
            #pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = ib + u;
                const bool pass = (i < i1) & (i < lim) & extents_overlap(bi[u], bj);
                const u64 ball = __ballot(pass);
                if (pass) buf[nbuf + __popcll(ball & ((1ull << wl_lane) - 1ull))] = (unsigned short)(i | (j << 8));
                nbuf += __popcll(ball);
            }
            if (nbuf > cap - 4 * WAVE) { cnt += count_candidates(v, buf, nbuf, wl_lane); nbuf = 0; }  // no room for another group
        }
    }
    STAMP(13);
    cnt += count_candidates(v, buf, nbuf, wl_lane);
    STAMP(14);
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (wl_lane == 0 && cnt) atomicAdd(total_cnt, cnt);
    lds_sync();
    STAMP(15);
    // find_wirelength: the adds happen in route order; empty slots add +0.0, which leaves a non-negative sum
    // unchanged bit for bit.  Every lane fetches the lengths of its own slots once (one LDS round trip), the sum
    // then runs over v_readlane broadcasts.
    double wl = 0.0;
    for (int base = 0; base < np; base += WAVE) {
        const int sidx = base + wl_lane;
        const double d = (sidx < np && v.act[sidx]) ? v.D[sidx] : 0.0;
        const int dlo = __double2loint(d), dhi = __double2hiint(d);
        #pragma unroll
        for (int blk = 0; blk < WAVE; blk += 16) {  // constant lane selects: the broadcasts run ahead of the add chain
            if (base + blk >= np) break;
            #pragma unroll
            for (int il = blk; il < blk + 16; il++)
                wl += __hiloint2double(__builtin_amdgcn_readlane(dhi, il), __builtin_amdgcn_readlane(dlo, il));
        }
    }
    *wirelength = wl;
    *nintersections = *total_cnt;
    lds_sync();
}
__device__ inline void count_and_length(const DevParams &p, const SegView &v, const EnvHdr *hdr, const PinRec *pins, int lane,
                                        double *wirelength, int *nintersections) {
    const int np = hdr->npins;
    count_prepare(v, np, pins, lane);
    count_finish(p, v, np, lane, wirelength, nintersections);
}

__device__ inline void route_centroid(const DevParams &p, const EnvHdr *hdr, const PinRec *pins, double *seg,
                                      int lane, double *wirelength, int *nintersections) {
    const SegView v = seg_view(seg, p.P);
    net_offsets_and_centroids(v, hdr, pins, lane);
    STAMP(5);
    build_centroid_segments(v, hdr, pins, lane);
    STAMP(6);
    count_prepare(v, hdr->npins, pins, lane);
    STAMP(22);
    count_finish(p, v, hdr->npins, lane, wirelength, nintersections);
    STAMP(8);
}

// ---- beam-search routing (S:1273-1286 pin_outlier, S:1303-1369 beam_search, S:1371-1406) -----------------
// beam_search keeps, per popped path, the beam_width nearest unvisited points of
// `sorted(points_to_visit - visited, key=distance)`.  Python's sort is stable, so neighbours at equal distance
// keep the iteration order of that temporary CPython set -- a pure function of the tuple hashes and of
// Objects/setobject.c's open-addressing table (SURVEY.md trap T2).  That order can only change WHICH points are
// kept when the beam_width-th and the next distance tie (the order among kept neighbours is irrelevant: heapq
// pops by (priority, path), not by insertion).  So the set model below runs only on such boundary ties.
// One lane per net; all per-net scratch lives in LDS (no private-memory arrays -> no scratch segment).
#define CS_EMPTY 0xFF
#define CS_DUMMY 0xFE
#define BS_MAXPTS (PCBENV_MAX_PINS_PER_NET - 1)
struct CSet { int mask, fill, used; unsigned char t[32]; int pad; };  // 48 bytes
// one partial path of the beam: four 64-bit words so that queue traffic is wide LDS accesses and the popped
// entry lives in registers.  meta = visited (bits 0-15) | length (bits 16-23); p0/p1 = the path, one byte per
// point index (0xFF = the start point).
struct BsEntry {
    double prio; u64 meta, p0, p1;
    __device__ unsigned visited() const { return (unsigned)(meta & 0xFFFFull); }
    __device__ int len() const { return (int)((meta >> 16) & 0xFFull); }
    __device__ int at(int j) const { return (int)(((j < 8 ? p0 : p1) >> ((j & 7) * 8)) & 0xFFull); }
    __device__ void push(int idx) {
        const int l = len();
        const u64 b = (u64)(unsigned)idx << ((l & 7) * 8);
        if (l < 8) p0 |= b; else p1 |= b;
        meta = (meta & ~(0xFFull << 16)) | ((u64)(l + 1) << 16) | (1ull << idx);
    }
};
static_assert(sizeof(BsEntry) == 32 && sizeof(CSet) == 48, "beam LDS records");
#define BEAM_LDS_PER_NET(k) (64 * (k) * (k) + 16 * 8 + 16 + 2 * 48)
#define BEAM_LDS_BYTES(nets, k) ((nets) * BEAM_LDS_PER_NET(k))

// points to visit of one net: the net's pins without the start pin `st`
struct NetPts {  // coordinates packed one byte each into registers (<= 15 points): no LDS round trip per access
    u64 xs0, xs1, ys0, ys1;
    __device__ int x(int i) const { return (int)(((i < 8 ? xs0 : xs1) >> ((i & 7) * 8)) & 0xFFull); }
    __device__ int y(int i) const { return (int)(((i < 8 ? ys0 : ys1) >> ((i & 7) * 8)) & 0xFFull); }
    __device__ static NetPts load(const PinRec *p, int cnt, int st) {
        NetPts n{0ull, 0ull, 0ull, 0ull};
        int m = 0;
        for (int i = 0; i < cnt; i++) {
            if (i == st) continue;
            const u64 x = (u64)(unsigned char)p[i].abs_x << ((m & 7) * 8), y = (u64)(unsigned char)p[i].abs_y << ((m & 7) * 8);
            if (m < 8) { n.xs0 |= x; n.ys0 |= y; } else { n.xs1 |= x; n.ys1 |= y; }
            m++;
        }
        return n;
    }
};

__device__ inline u64 tuple_hash2(int x, int y) {  // Objects/tupleobject.c (xxHash-style), hash(int) == int
    const u64 P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P5 = 2870177450012600261ull;
    u64 acc = P5;
    acc += (u64)(long long)x * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
    acc += (u64)(long long)y * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
    acc += 2ull ^ (P5 ^ 3527539ull);
    return acc == ~0ull ? 1546275796ull : acc;
}
__device__ inline void cs_init(CSet *s, int size) {
    s->mask = size - 1; s->fill = 0; s->used = 0;
    for (int i = 0; i < 32; i++) s->t[i] = CS_EMPTY;
}
// first unused slot on the probe sequence of `hash` (set_insert_clean / the miss path of set_add_entry)
__device__ inline int cs_probe_unused(const CSet *s, u64 hash, int *freeslot) {
    const unsigned mask = (unsigned)s->mask;
    u64 perturb = hash;
    unsigned i = (unsigned)hash & mask;
    for (;;) {
        const unsigned probes = (i + 9u <= mask) ? 9u : 0u;
        for (unsigned k = 0; k <= probes; k++) {
            const unsigned char c = s->t[i + k];
            if (c == CS_EMPTY) return (int)(i + k);
            if (c == CS_DUMMY && freeslot) *freeslot = (int)(i + k);
        }
        perturb >>= 5;
        i = (unsigned)(((u64)i * 5u + 1u + perturb) & mask);
    }
}
// set_table_resize: re-insert the active keys in old slot order (the old table is copied to `tmp` first)
__device__ inline void cs_resize(CSet *s, CSet *tmp, int minused, const NetPts &pt) {
    int newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    *tmp = *s;
    cs_init(s, newsize);
    for (int i = 0; i <= tmp->mask; i++)
        if (tmp->t[i] < CS_DUMMY) s->t[cs_probe_unused(s, tuple_hash2(pt.x(tmp->t[i]), pt.y(tmp->t[i])), 0)] = tmp->t[i];
    s->fill = s->used = tmp->used;
}
__device__ inline void cs_add(CSet *s, CSet *tmp, int key, const NetPts &pt) {
    int freeslot = -1;
    const int slot = cs_probe_unused(s, tuple_hash2(pt.x(key), pt.y(key)), &freeslot);
    if (freeslot >= 0) { s->t[freeslot] = (unsigned char)key; s->used++; return; }
    s->t[slot] = (unsigned char)key; s->fill++; s->used++;
    if (s->fill * 5 >= s->mask * 3) cs_resize(s, tmp, s->used * 4, pt);
}
__device__ inline void cs_discard(CSet *s, int key, const NetPts &pt) {
    const unsigned mask = (unsigned)s->mask;
    const u64 hash = tuple_hash2(pt.x(key), pt.y(key));
    u64 perturb = hash;
    unsigned i = (unsigned)hash & mask;
    for (;;) {
        const unsigned probes = (i + 9u <= mask) ? 9u : 0u;
        for (unsigned k = 0; k <= probes; k++) {
            const unsigned char c = s->t[i + k];
            if (c == CS_EMPTY) return;
            if (c == (unsigned char)key) { s->t[i + k] = CS_DUMMY; s->used--; return; }
        }
        perturb >>= 5;
        i = (unsigned)(((u64)i * 5u + 1u + perturb) & mask);
    }
}
// Iteration order of `set(points) - visited` (set_difference: copy-and-discard when len(A) >> 2 > len(visited),
// else a fresh set filled in A's slot order).  A and R are LDS tables; `order` receives point indices.
__device__ inline int cs_difference_order(CSet *A, CSet *R, int m, unsigned visited, const NetPts &pt, unsigned char *order) {
    // points_to_visit = set(points): inserted in list order.  R doubles as the resize temporary while A is built.
    cs_init(A, 8);
    for (int i = 0; i < m; i++) cs_add(A, R, i, pt);
    if ((m >> 2) > __popc(visited)) {
        cs_init(R, 8);
        if (m * 5 >= R->mask * 3) { int ns = 8; while (ns <= 2 * m) ns <<= 1; cs_init(R, ns); }
        if (R->mask == A->mask) { *R = *A; }  // set_merge: same size, no dummies -> the table is copied as is
        else {
            for (int i = 0; i <= A->mask; i++)
                if (A->t[i] < CS_DUMMY) R->t[cs_probe_unused(R, tuple_hash2(pt.x(A->t[i]), pt.y(A->t[i])), 0)] = A->t[i];
            R->fill = R->used = A->used;
        }
        for (int k = 0; k < m; k++) if (visited >> k & 1u) cs_discard(R, k, pt);
        // "if more than 1/4th are dummies, resize them away" cannot trigger for m <= 15 (<= 2 dummies, mask >= 15)
    } else {
        // fresh result set filled in A's slot order: collect the survivors first, after which A is free to
        // serve as the temporary of R's set_table_resize (5th insert: 8 -> 32 slots)
        int ns = 0;
        for (int i = 0; i <= A->mask; i++)
            if (A->t[i] < CS_DUMMY && !(visited >> A->t[i] & 1u)) order[ns++] = A->t[i];
        cs_init(R, 8);
        for (int i = 0; i < ns; i++) cs_add(R, A, order[i], pt);
    }
    int n = 0;
    for (int i = 0; i <= R->mask; i++) if (R->t[i] < CS_DUMMY) order[n++] = R->t[i];
    return n;
}

// One net, one lane: fills the net's slots [s, s+cnt) of the segment view with the beam route.
// `scratch` = this net's BEAM_LDS_PER_NET(k) bytes of LDS.
__device__ inline void beam_route_net(const SegView &v, const PinRec *pins, int s, int cnt, int k, unsigned char *scratch) {
    BsEntry *queue = (BsEntry *)scratch, *next = queue + k * k;
    double *dist = (double *)(scratch + 64 * k * k);
    unsigned char *order = (unsigned char *)(dist + 16);
    CSet *A = (CSet *)(order + 16), *R = A + 1;
    const double cx = v.cen[pins[s].net], cy = v.cen[PCBENV_MAX_NETS + pins[s].net];
    int st = 0; double bd = 0.0;  // pin_outlier: first arg-max of the distance to the centroid
    for (int i = 0; i < cnt; i++) {
        const double d = norm2((double)pins[s + i].abs_x - cx, (double)pins[s + i].abs_y - cy);
        if (i == 0 || d > bd) { bd = d; st = i; }
    }
    const int sx = pins[s + st].abs_x, sy = pins[s + st].abs_y;
    const int m = cnt - 1;
    const NetPts pt = NetPts::load(pins + s, cnt, st);
    const unsigned all = (1u << m) - 1u;
    int qn = 1;
    { BsEntry e0; e0.prio = 0.0; e0.meta = 1ull << 16; e0.p0 = 0xFFull; e0.p1 = 0ull; queue[0] = e0; }
    bool found = false;
    BsEntry res;
    while (!found) {
        int nn = 0;
        unsigned taken = 0;
        const int pops = k < qn ? k : qn;
        for (int t = 0; t < pops && !found; t++) {
            int best = -1;  // heappop: minimum (priority, path) of what is left
            BsEntry e;
            for (int i = 0; i < qn; i++) {
                if (taken >> i & 1u) continue;
                const BsEntry a = queue[i];
                bool less;
                if (best < 0) less = true;
                else if (a.prio != e.prio) less = a.prio < e.prio;
                else {  // equal priorities: python compares the path lists of (x, y) tuples
                    less = a.len() < e.len();
                    const int n = a.len() < e.len() ? a.len() : e.len();
                    for (int j = 0; j < n; j++) {
                        const int pa = a.at(j), pb = e.at(j);
                        const int ax = pa == 0xFF ? sx : pt.x(pa), ay = pa == 0xFF ? sy : pt.y(pa);
                        const int bx = pb == 0xFF ? sx : pt.x(pb), by = pb == 0xFF ? sy : pt.y(pb);
                        if (ax != bx) { less = ax < bx; break; }
                        if (ay != by) { less = ay < by; break; }
                    }
                }
                if (less) { best = i; e = a; }
            }
            taken |= 1u << best;
            if (e.visited() == all) { found = true; res = e; break; }
            const int cur = e.at(e.len() - 1);
            const int ux = cur == 0xFF ? sx : pt.x(cur), uy = cur == 0xFF ? sy : pt.y(cur);
            // the k+1 nearest unvisited points in registers (ascending distance, index order among equals)
            double td[PCBENV_MAX_BEAM_WIDTH + 1]; int ti[PCBENV_MAX_BEAM_WIDTH + 1];
            #pragma unroll
            for (int q = 0; q <= PCBENV_MAX_BEAM_WIDTH; q++) { td[q] = 0.0; ti[q] = 0; }
            int nfill = 0, cntn = 0;
            for (int i = 0; i < m; i++) {
                if (e.visited() >> i & 1u) continue;
                cntn++;
                double cd = norm2((double)(ux - pt.x(i)), (double)(uy - pt.y(i)));
                int ci = i;
                bool shifting = false, placed = false;
                #pragma unroll
                for (int q = 0; q <= PCBENV_MAX_BEAM_WIDTH; q++) {
                    if (q > k || placed) continue;
                    if (q == nfill) { td[q] = cd; ti[q] = ci; placed = true; }
                    else if (shifting || td[q] > cd) {
                        const double xd = td[q]; const int xi = ti[q];
                        td[q] = cd; ti[q] = ci; cd = xd; ci = xi; shifting = true;
                    }
                }
                if (nfill <= k) nfill++;
            }
            const int take = cntn < k ? cntn : k;
            bool tie = false;
            #pragma unroll
            for (int q = 1; q <= PCBENV_MAX_BEAM_WIDTH; q++) if (q == k && cntn > k) tie = td[q - 1] == td[q];
            if (tie) {  // boundary tie: the CPython set order decides who is kept
                const int nset = cs_difference_order(A, R, m, e.visited(), pt, order);
                for (int i = 0; i < nset; i++) dist[i] = norm2((double)(ux - pt.x(order[i])), (double)(uy - pt.y(order[i])));
                for (int i = 1; i < nset; i++) {  // sorted(key=distance): stable
                    const unsigned char o = order[i]; const double d = dist[i];
                    int j = i - 1;
                    while (j >= 0 && dist[j] > d) { order[j + 1] = order[j]; dist[j + 1] = dist[j]; j--; }
                    order[j + 1] = o; dist[j + 1] = d;
                }
                for (int i = 0; i < take; i++) { BsEntry q = e; q.push(order[i]); q.prio = e.prio + dist[i]; next[nn++] = q; }
            } else {
                #pragma unroll
                for (int q = 0; q < PCBENV_MAX_BEAM_WIDTH; q++)
                    if (q < take) { BsEntry w = e; w.push(ti[q]); w.prio = e.prio + td[q]; next[nn++] = w; }
            }
        }
        if (!found) { BsEntry *tmp = queue; queue = next; next = tmp; qn = nn; if (qn == 0) break; }
    }
    for (int i = 0; i < cnt; i++) v.act[s + i] = 0;
    if (!found) return;
    for (int i = 0; i + 1 < res.len(); i++) {
        const int a = res.at(i), b = res.at(i + 1);
        const double x1 = a == 0xFF ? sx : pt.x(a), y1 = a == 0xFF ? sy : pt.y(a);
        const double x2 = b == 0xFF ? sx : pt.x(b), y2 = b == 0xFF ? sy : pt.y(b);
        v.X1[s + i] = x1; v.Y1[s + i] = y1; v.X2[s + i] = x2; v.Y2[s + i] = y2;
        v.D[s + i] = norm2(x1 - x2, y1 - y2);
        v.act[s + i] = 1;
    }
}

// beam (and, for "both", centroid) routes of the terminal state -> wirelength, #intersections of the chosen route
__device__ inline void route_beam_or_both(const DevParams &p, const EnvHdr *hdr, const PinRec *pins, double *seg,
                                          int lane, double *wirelength, int *nintersections) {
    const SegView v = seg_view(seg, p.P);
    unsigned char *beam = v.beam;
    net_offsets_and_centroids(v, hdr, pins, lane);
    for (int n = lane; n < hdr->nnets; n += NT)
        beam_route_net(v, pins, v.nstart[n], v.nstart[n + 1] - v.nstart[n], p.beam_width, beam + (size_t)n * BEAM_LDS_PER_NET(p.beam_width));
    lds_sync();
    count_and_length(p, v, hdr, pins, lane, wirelength, nintersections);
    if (p.reward_type == PCBENV_REWARD_BOTH) {  // S:609-627 lowest_num_intersections: ties keep the beam route
        double wc; int kc;
        build_centroid_segments(v, hdr, pins, lane);
        count_and_length(p, v, hdr, pins, lane, &wc, &kc);
        if (kc < *nintersections) { *nintersections = kc; *wirelength = wc; }
    }
}

// ----------------------------------------------------------------------------------------------
// shared pieces of reset / step
// ----------------------------------------------------------------------------------------------
struct Lds {
    EnvHdr *hdr; u64 *occ, *vm; CompRec *comps; PinRec *pins;
    u64 *hf; unsigned char *cls; double *seg;
};
__device__ inline Lds carve(unsigned char *smem, const DevParams &p) {
    Lds l;
    l.hdr = (EnvHdr *)smem;
    l.occ = (u64 *)(smem + p.offOcc);
    l.vm = (u64 *)(smem + p.offVm);
    l.comps = (CompRec *)(smem + p.offComps);
    l.pins = (PinRec *)(smem + p.offPins);
    l.hf = (u64 *)(smem + p.ldsHf);
    l.cls = smem + p.ldsCls;
    l.seg = (double *)(smem + p.ldsSeg);
    return l;
}
__device__ inline void load_state(unsigned char *smem, const DevParams &p, int e, int lane) {
    const uint4 *src = (const uint4 *)(p.state + (size_t)e * p.stateStride);
    uint4 *dst = (uint4 *)smem;
    for (int i = lane; i < (int)(p.stateStride / 16); i += NT) dst[i] = src[i];
    lds_sync();
}
__device__ inline void store_state(const unsigned char *smem, const DevParams &p, int e, int lane) {
    lds_sync();
    uint4 *dst = (uint4 *)(p.state + (size_t)e * p.stateStride);
    const uint4 *src = (const uint4 *)smem;
    // plain write-back stores: environment e runs on XCD e % 8 in every launch, so its state block is an L2 hit next step
    for (int i = lane; i < (int)(p.stateStride / 16); i += NT) dst[i] = src[i];
}

// Marginals of the legal mask for factorised policies (factorized_action_distributions.py:358, :401): per
// orientation "any legal cell" and per (orientation, row) "any legal column", read off the bit rows in LDS.
template <int KIND, int WW> __device__ inline void emit_marginals(const DevParams &p, Lds &l, int e, int lane) {
    if (!p.buf.mask_rows && !p.buf.mask_orientation) return;
    const int H = p.H, plane = H * WW, O = p.O;
    for (int i = lane; i < O * H; i += NT) {
        const int o = i / H, r = i - o * H;
        const u64 *row = l.vm + (o & 1) * plane + r * WW;
        bool a = false;
        for (int w = 0; w < WW; w++) a |= row[w] != 0;
        if (p.buf.mask_rows) p.buf.mask_rows[(size_t)e * O * H + i] = a ? 1 : 0;
    }
    if (p.buf.mask_orientation) {
        for (int o = (int)(lane / WAVE); o < O; o += NT / WAVE) {  // one wavefront per orientation
            bool a = false;
            for (int i = (lane & 63); i < plane; i += WAVE) a |= l.vm[(o & 1) * plane + i] != 0;
            a = __any(a);
            if ((lane & 63) == 0) p.buf.mask_orientation[(size_t)e * O + o] = a ? 1 : 0;
        }
    }
}

// Mask of the current component (or zeros) into l.vm, both orientations, and -- when `emit` -- the grid rows
// [gr0, gr1) and the action_mask planes, each written as soon as its bits exist so that the HBM write stream
// starts before the second orientation is folded.  Returns "some action is legal".
template <int KIND, int WW>
__device__ inline bool mask_and_emit(const DevParams &p, Lds &l, int e, int lane, bool emit, int gr0, int gr1) {
    const int H = p.H, W = p.W, HW = H * W, plane = H * WW;
    const int cur = l.hdr->cur;
    unsigned char *m = (emit && p.buf.action_mask) ? p.buf.action_mask + (size_t)e * p.O * HW : 0;
    if (emit && p.buf.grid) emit_plane<WW>(p.buf.grid + (size_t)e * HW, l.occ, gr0, gr1, W, lane);
    bool any = false;
    if (KIND == PCBENV_SQUARE) {
        any = window_mask<WW>(l.occ, l.hf, l.vm, H, W, p.component_n, p.component_n, lane, &l.hdr->flag);
        if (m) emit_plane<WW>(m, l.vm, 0, H, W, lane);
        if (emit) emit_marginals<KIND, WW>(p, l, e, lane);
        return any;
    }
    const bool four = (KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL);  // S:1852-1853 mask[2] = mask[0], mask[3] = mask[1]
    if (cur >= 0) {
        const int h = l.comps[cur].h, w = l.comps[cur].w;
        any = window_mask<WW>(l.occ, l.hf, l.vm, H, W, h, w, lane, &l.hdr->flag);
        if (m) { emit_plane<WW>(m, l.vm, 0, H, W, lane); if (four) emit_plane<WW>(m + 2 * HW, l.vm, 0, H, W, lane); }
        if (h == w) {
            for (int i = lane; i < plane; i += NT) l.vm[plane + i] = l.vm[i];
            lds_sync();
        } else {
            any |= window_mask<WW>(l.occ, l.hf, l.vm + plane, H, W, w, h, lane, &l.hdr->flag);
        }
    } else {
        for (int i = lane; i < 2 * plane; i += NT) l.vm[i] = 0ull;
        lds_sync();
        if (m) { emit_plane<WW>(m, l.vm, 0, H, W, lane); if (four) emit_plane<WW>(m + 2 * HW, l.vm, 0, H, W, lane); }
    }
    if (m) { emit_plane<WW>(m + HW, l.vm + plane, 0, H, W, lane); if (four) emit_plane<WW>(m + 3 * HW, l.vm + plane, 0, H, W, lane); }
    if (emit) emit_marginals<KIND, WW>(p, l, e, lane);
    return any;
}

// S:1663-1675 draw_pins: class map (0 empty, 1 occupied without pin, n+2 pin of net n) -> one-hot[:, :, 1:];
// rows [r0, r1) of the (H, W, K) tensor (a step only changes the rows of the placed rectangle).
template <int WW> __device__ inline void emit_pin_grid(const DevParams &p, Lds &l, int e, int lane, int r0, int r1) {
    if (!p.buf.pin_grid) return;
    const int W = p.W, HW = p.H * W, K = p.K;
    const int c0 = r0 * W, c1 = r1 * W;
    unsigned char *dst = p.buf.pin_grid + (size_t)e * HW * K;
    const long long b0 = (long long)c0 * K, b1 = (long long)c1 * K;
    for (int i = c0 + lane; i < c1; i += NT) {
        int r = i / W, c = i - r * W;
        l.cls[i] = (unsigned char)((l.occ[r * WW + (c >> 6)] >> (c & 63)) & 1ull);
    }
    lds_sync();
    for (int q = lane; q < l.hdr->npins; q += NT) {
        const PinRec pr = l.pins[q];
        if (pr.abs_x >= r0 && pr.abs_x < r1 && pr.abs_y >= 0) l.cls[pr.abs_x * W + pr.abs_y] = (unsigned char)(pr.net + 2);
    }
    lds_sync();
    if ((b0 & 15) == 0 && (b1 & 15) == 0 && (((uintptr_t)dst) & 15) == 0) {
        uint4 *d4 = (uint4 *)dst;
        // every cell owns K consecutive bytes with at most one 1 (at class-1): visit the <= 16/K + 2 cells a
        // 16-byte chunk overlaps and drop their 1-bytes into two 64-bit halves
        const unsigned kinv = 0xFFFFFFFFu / (unsigned)K + 1u;  // floor(b / K) == umulhi(b, kinv) for b < 2^32 / K
        for (int c = (int)(b0 / 16) + lane; c < (int)(b1 / 16); c += NT) {
            const int bb = c * 16;
            int cell = (int)__umulhi((unsigned)bb, kinv);
            if (cell * K > bb) cell--;  // (never taken at these sizes; keeps the division exact regardless)
            u64 lo = 0, hi = 0;
            for (int base = cell * K; base < bb + 16 && cell < c1; base += K, cell++) {
                const unsigned cl = l.cls[cell];
                const int off = base + (int)cl - 1 - bb;  // byte of this cell's 1 inside the chunk
                if (cl != 0 && off >= 0 && off < 16) {
                    if (off < 8) lo |= 1ull << (8 * off); else hi |= 1ull << (8 * (off - 8));
                }
            }
            STORE16(d4 + c, make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)));
        }
    } else {
        for (long long i = b0 + lane; i < b1; i += NT) {
            int cell = (int)(i / K), ch = (int)(i - (long long)cell * K);
            dst[i] = (unsigned char)(l.cls[cell] == ch + 1);
        }
    }
}

// Feature rows of one pin (P:72-103 / S:70-104 Pin.calculate_feature): [rel_x, rel_y, abs_x, abs_y]
template <int KIND> __device__ inline void write_pin_num(const DevParams &p, int e, const PinRec &pr) {
    if (!p.buf.all_pins_num_feature) return;
    int row;
    if (KIND == PCBENV_SPATIAL) row = pr.id & PIN_ID_MASK;
    else { if (pr.id & PIN_LOSER) return; row = pr.comp * p.mp + (pr.id & PIN_ID_MASK); }
    double *f = p.buf.all_pins_num_feature + ((size_t)e * p.pinRows + row) * 4;
    f[0] = pr.rel_x; f[1] = pr.rel_y; f[2] = pr.abs_x; f[3] = pr.abs_y;
}

// Terminal reward (S:793-929 find_reward), all three reward types, inside the step kernel.
// ROUTES = false compiles the beam-search code out (reward_type centroid: what every shipped reference config uses).
template <int KIND, bool ROUTES>
__device__ inline void terminal_reward(const DevParams &p, Lds &l, int e, int lane) {
    const bool placed_all = l.hdr->cur < 0;
    double reward, wl, ni;
    if (!placed_all) {  // S:853-863 worst case: the upper bounds, normalised (spatial: twice, quirk Q3)
        reward = -p.w_wl * (p.max_wl / p.wl_norm) - p.w_int * (p.max_int / p.int_norm);
        wl = p.max_wl; ni = p.max_int;
    } else {
        double wsum; int cnt;
        if (!ROUTES) route_centroid(p, l.hdr, l.pins, l.seg, lane, &wsum, &cnt);
        else route_beam_or_both(p, l.hdr, l.pins, l.seg, lane, &wsum, &cnt);
        wl = wsum / p.wl_norm;
        ni = (double)cnt / p.int_norm;
        reward = -1 * (p.w_wl * wl + p.w_int * ni);
    }
    if (lane == 0) {
        p.buf.reward[e] = reward;
        if (p.buf.info) { p.buf.info[2 * e] = wl; p.buf.info[2 * e + 1] = ni; }
    }
}

// ----------------------------------------------------------------------------------------------
// uniform legal-action sampler (rollout driver; agent/random/random_policy_*.py counterpart)
// ----------------------------------------------------------------------------------------------
__device__ inline u64 mix64(u64 z) {  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ inline int select_bit(u64 w, int k) {  // position of the k-th (0-based) set bit: binary search on popcounts
    int pos = 0;
    #pragma unroll
    for (int width = 32; width >= 1; width >>= 1) {
        const int c = __popcll(w & ((1ull << width) - 1ull));
        if (k >= c) { k -= c; w >>= width; pos += width; }
    }
    return pos;
}
// Uniform draw over the set bits of the legal-action bit mask vm (planes 0/1; pin kinds also mirror them as
// orientations 2/3): per-lane popcounts of a contiguous run of words, wave prefix sum, the owner lane selects
// the k-th set bit.  rnd = mix64(mix64(seed ^ GOLDEN*(env+1)) + step); pick = hi32(rnd) * n >> 32.
__device__ inline void sample_action(const u64 *vm, const DevParams &p, int genv, int lane, u64 seed, u64 step_index,
                                     int *o, int *x, int *y) {
    const int WW = p.WW, plane = p.H * WW;
    const int words = (p.kind == PCBENV_SQUARE ? 1 : 2) * plane;
    const int per = (words + WAVE - 1) / WAVE;
    int mine = 0;
    for (int i = lane * per; i < (lane + 1) * per && i < words; i++) mine += __popcll(vm[i]);
    const int incl = wave_inclusive_scan(mine, lane);
    const int total = __builtin_amdgcn_readlane(incl, WAVE - 1);
    *o = 0; *x = 0; *y = 0;
    if (total <= 0) return;
    const u64 rnd = mix64(mix64(seed ^ 0x9E3779B97F4A7C15ull * ((u64)genv + 1)) + step_index);
    const bool mirrored = (p.kind == PCBENV_PIN || p.kind == PCBENV_SPATIAL);  // two orientations per mask plane
    const unsigned pick = (unsigned)(((rnd >> 32) * (u64)(mirrored ? 2 * total : total)) >> 32);
    const int rep = pick >= (unsigned)total ? 1 : 0, k = (int)pick - rep * total;
    const int excl = incl - mine;
    const bool owner = k >= excl && k < incl;
    int found = 0;
    if (owner) {
        int rem = k - excl;
        for (int i = lane * per; i < (lane + 1) * per && i < words; i++) {
            const u64 w = vm[i];
            const int c = __popcll(w);
            if (rem < c) { found = i * 64 + select_bit(w, rem); break; }
            rem -= c;
        }
    }
    const u64 ball = __ballot(owner);
    found = __builtin_amdgcn_readlane(found, __builtin_amdgcn_readfirstlane(__ffsll((long long)ball) - 1));
    const int word = found >> 6, bit = found & 63;
    const int pl = word >= plane ? 1 : 0, rw = word - pl * plane;
    *o = pl + 2 * rep;
    *x = WW == 1 ? rw : rw >> 1;
    *y = (rw - *x * WW) * 64 + bit;
}
__global__ __launch_bounds__(WAVE) void k_sample(DevParams p, int *__restrict__ actions, int fmt, u64 seed,
                                                 u64 first_env, u64 step_index) {
    const int e = blockIdx.x, lane = threadIdx.x;
    const u64 *vm = (const u64 *)(p.state + (size_t)e * p.stateStride + p.offVm);
    int o, x, y;
    sample_action(vm, p, (int)first_env + e, lane, seed, step_index, &o, &x, &y);
    if (lane == 0) {
        if (fmt == PCBENV_ACTION_FLAT) actions[e] = o * p.H * p.W + x * p.W + y;
        else { actions[3 * e] = o; actions[3 * e + 1] = x; actions[3 * e + 2] = y; }
    }
}

// ----------------------------------------------------------------------------------------------
// reset (R:310-351, P:1544-1597, S:1487-1549, Q:74-113): header is in LDS; builds the new episode's state in
// LDS from the next queued instance and rewrites every observation tensor of environment e.
// ----------------------------------------------------------------------------------------------
// The next queued instance of environment e: header and 8-byte records, all loads issued together.
struct InstRegs { int nc, nn, np; u64 comp; u64 pin[4]; };
__device__ inline void fetch_instance(const DevParams &p, unsigned qcursor, int e, int lane, InstRegs &ir) {
    const unsigned slot = qcursor % (unsigned)p.Q;
    const unsigned char *rec = p.queue + ((size_t)slot * p.B + e) * p.instStride;
    const int *ih = (const int *)rec;
    const u64 *crec = (const u64 *)(rec + 16), *prec = crec + p.C;  // 8-byte records, one load each
    ir.nc = ih[0]; ir.nn = ih[1]; ir.np = ih[2];
    ir.comp = lane < p.C ? crec[lane] : 0ull;
    #pragma unroll
    for (int r = 0; r < 4; r++) { const int q = lane + r * NT; ir.pin[r] = q < p.P ? prec[q] : 0ull; }
}

template <int KIND, int WW> __device__ inline void reset_env(const DevParams &p, Lds &l, int e, int lane) {
    const int H = p.H, W = p.W, HW = H * W;
    lds_sync();
    // The float64 pin-feature tensors are maintained row-wise (a step rewrites only the placed component's
    // rows), so a reset clears just the rows the finished episode used -- unless these buffers have not been
    // initialised for this environment yet (first reset after pcbenv_bind_buffers): then a full zero fill.
    // Rows of the finished episode that the new episode rewrites are left alone (no write-after-write on a row, so
    // no ordering wait between the clear and the later row writes): spatial rows are the pin ids 0..np-1; the pin
    // env's rows [component, pin_id] go through a membership bit map in the fold scratch.
    InstRegs ir;
    if (KIND != PCBENV_SQUARE) fetch_instance(p, l.hdr->qcursor, e, lane, ir);
    bool rows_cleared = false;
    if ((KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL) && l.hdr->feat_gen == p.bind_gen &&
        (KIND == PCBENV_SPATIAL || p.C * p.mp <= H * WW * 64)) {
        u64 *rowbits = l.hf;
        if (KIND == PCBENV_PIN) {
            for (int i = lane; i < H * WW; i += NT) rowbits[i] = 0ull;
            lds_sync();
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                const int q = lane + r * NT;
                if (q < ir.np && q < p.P) {
                    const u64 w = ir.pin[r];
                    const int row = (int)((w >> 24) & 0xFF) * p.mp + (int)((w >> 32) & PIN_ID_MASK);
                    atomicOr((unsigned long long *)&rowbits[row >> 6], 1ull << (row & 63));
                }
            }
            lds_sync();
        }
        for (int q = lane; q < l.hdr->npins; q += NT) {
            const PinRec pr = l.pins[q];
            const int row = KIND == PCBENV_SPATIAL ? (pr.id & PIN_ID_MASK) : pr.comp * p.mp + (pr.id & PIN_ID_MASK);
            if (KIND == PCBENV_SPATIAL ? row < ir.np : (int)((rowbits[row >> 6] >> (row & 63)) & 1ull)) continue;
            if (p.buf.all_pins_num_feature) {
                double *f = p.buf.all_pins_num_feature + ((size_t)e * p.pinRows + row) * 4;
                f[0] = 0.0; f[1] = 0.0; f[2] = 0.0; f[3] = 0.0;
            }
            if (p.buf.all_pins_cat_feature) {
                double *f = p.buf.all_pins_cat_feature + ((size_t)e * p.pinRows + row) * p.catW;
                f[0] = 0.0; if (KIND == PCBENV_SPATIAL) f[1] = 0.0;
            }
        }
        rows_cleared = true;
    }
    lds_sync();
    for (int i = lane; i < H * WW; i += NT) l.occ[i] = 0ull;
    if (KIND != PCBENV_SQUARE) {
        const int nc = ir.nc, nn = ir.nn, np = ir.np;
        if (lane < p.C) {
            const u64 w = ir.comp;
            CompRec cr; cr.h = (unsigned char)w; cr.w = (unsigned char)(w >> 8); cr.px = -1; cr.py = -1;
            cr.pad[0] = cr.pad[1] = cr.pad[2] = cr.pad[3] = 0;
            if (lane >= nc) { cr.h = 0; cr.w = 0; }
            l.comps[lane] = cr;
        }
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            const int q = lane + r * NT;
            if (q >= p.P) break;
            const u64 w = ir.pin[r];
            PinRec pr; pr.rel_x = (unsigned char)w; pr.rel_y = (unsigned char)(w >> 8); pr.abs_x = -1; pr.abs_y = -1;
            pr.net = (unsigned char)(w >> 16); pr.comp = (unsigned char)(w >> 24);
            pr.id = (unsigned short)(w >> 32);
            if (q >= np) { pr.rel_x = pr.rel_y = 0; pr.net = 0xFF; pr.comp = 0xFF; pr.id = 0; }
            l.pins[q] = pr;
        }
        if (lane == 0) {
            l.hdr->ncomp = (short)nc; l.hdr->nnets = (short)nn; l.hdr->npins = (short)np; l.hdr->cur = 0;
            l.hdr->qcursor += 1; l.hdr->episode += 1;
        }
        lds_sync();
        if (KIND == PCBENV_PIN && lane < WAVE) {
            // quirk Q1: rows [component, pin_id] collide; the last writer in self.pins order wins.  Wavefront 0 keeps
            // the (component, pin_id) keys of its lanes' slots in registers and walks the pins with v_readlane:
            // a slot loses when a later slot carries the same key.
            if (np <= WAVE) {  // one slot per lane: one ballot per distinct key, its highest lane is the last writer
                const unsigned key = lane < np ? ((unsigned)l.pins[lane].comp << 16) | (l.pins[lane].id & PIN_ID_MASK) : 0xFFFFFFFFu;
                u64 remaining = __ballot(lane < np);
                bool lose = false;
                while (remaining) {
                    const unsigned k = (unsigned)__builtin_amdgcn_readlane((int)key, __ffsll((long long)remaining) - 1);
                    const u64 m = __ballot(key == k);
                    if (key == k && lane != 63 - __clzll((long long)m)) lose = true;
                    remaining &= ~m;
                }
                if (lose) l.pins[lane].id |= PIN_LOSER;
            } else {
                unsigned key[4]; bool lose[4];
                #pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int sidx = 64 * r + lane;
                    key[r] = sidx < np ? ((unsigned)l.pins[sidx].comp << 16) | (l.pins[sidx].id & PIN_ID_MASK) : 0xFFFFFFFFu;
                    lose[r] = false;
                }
                #pragma unroll
                for (int ri = 0; ri < 4; ri++) {
                    if (64 * ri >= np) break;
                    const int lim = min(64, np - 64 * ri);
                    for (int il = 0; il < lim; il++) {
                        const unsigned ki = (unsigned)__builtin_amdgcn_readlane((int)key[ri], il);
                        const int i = 64 * ri + il;
                        #pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (64 * r < np && 64 * r + lane < i && key[r] == ki) lose[r] = true;
                    }
                }
                #pragma unroll
                for (int r = 0; r < 4; r++) if (lose[r]) l.pins[64 * r + lane].id |= PIN_LOSER;
            }
        }
    } else if (lane == 0) {
        l.hdr->ncomp = 0; l.hdr->nnets = 0; l.hdr->npins = 0; l.hdr->cur = 0; l.hdr->episode += 1;
    }
    lds_sync();
    STAMP(16);
    mask_and_emit<KIND, WW>(p, l, e, lane, true, 0, H);
    STAMP(17);

    if (KIND != PCBENV_SQUARE) {
        const int nc = l.hdr->ncomp, np = l.hdr->npins;
        // all_components_feature (R:60-79, S:203-239): [h, w, -1, -1, area/(H*W), (spatial: pin ids, -1 pad)]; absent rows 0
        // spatial scratch in the class-map zone (free until the next emit_pin_grid): pid[c][k] = id of the k-th pin
        // of component c in self.pins order (0xFFFF = none), netmask[c][rel_x][rel_y] = nets with a pin on that cell
        unsigned short *pid = (unsigned short *)l.cls;
        unsigned *netmask = (unsigned *)(l.cls + ((p.C * p.mp * 2 + 3) & ~3));
        if (KIND == PCBENV_SPATIAL) {
            for (int i = lane; i < p.C * p.mp; i += NT) { pid[i] = 0xFFFFu; netmask[i] = 0u; }
            lds_sync();
            for (int q = lane; q < np; q += NT) {
                const PinRec pr = l.pins[q];
                int rank = 0;
                #pragma unroll 4
                for (int q2 = 0; q2 < np; q2++) rank += (q2 < q) & (l.pins[q2].comp == pr.comp);  // broadcast reads
                pid[pr.comp * p.mp + rank] = (unsigned short)(pr.id & PIN_ID_MASK);
                atomicOr(&netmask[(int)pr.comp * p.mp + pr.rel_x * p.mw + pr.rel_y], 1u << pr.net);
            }
            lds_sync();
        }
        // all_components_feature (R:60-79, S:203-239): [h, w, -1, -1, area/(H*W), (spatial: pin ids, -1 pad)]; absent rows 0
        if (p.buf.all_components_feature) {
            double *cf = p.buf.all_components_feature + (size_t)e * p.C * p.F;
            for (int i = lane; i < p.C * p.F; i += NT) {
                const int c = i / p.F, k = i - c * p.F;
                double v = 0.0;
                if (c < nc) {
                    const CompRec cr = l.comps[c];
                    if (k == 0) v = cr.h; else if (k == 1) v = cr.w; else if (k == 2 || k == 3) v = -1.0;
                    else if (k == 4) v = (double)(cr.h * cr.w) / p.area;
                    else {
                        const unsigned id = KIND == PCBENV_SPATIAL ? pid[c * p.mp + k - 5] : 0xFFFFu;
                        v = id == 0xFFFFu ? -1.0 : (double)id;
                    }
                }
                cf[i] = v;
            }
        }
        STAMP(18);
        if (p.buf.placement_mask) {
            double *pm = p.buf.placement_mask + (size_t)e * p.C;
            for (int c = lane; c < p.C; c += NT)
                pm[c] = KIND == PCBENV_RECT ? 0.0 : (c == 0 ? 3.0 : (c < nc ? 1.0 : 0.0));
        }
        if (KIND == PCBENV_RECT && p.buf.component_mask) {
            double *cm = p.buf.component_mask + (size_t)e * p.C;
            for (int c = lane; c < p.C; c += NT) cm[c] = c < nc ? 1.0 : 0.0;
        }
        if (KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL) {
            if (!rows_cleared && p.buf.all_pins_num_feature) {
                double *f = p.buf.all_pins_num_feature + (size_t)e * p.pinRows * 4;
                for (int i = lane; i < p.pinRows * 4; i += NT) f[i] = 0.0;
            }
            if (!rows_cleared && p.buf.all_pins_cat_feature) {
                double *f = p.buf.all_pins_cat_feature + (size_t)e * p.pinRows * p.catW;
                for (int i = lane; i < p.pinRows * p.catW; i += NT)
                    f[i] = (KIND == PCBENV_SPATIAL && i >= (p.pinRows - 1) * p.catW) ? -1.0 : 0.0;  // S:1520
            }
            if (lane == 0) l.hdr->feat_gen = p.bind_gen;
            if (!rows_cleared) {  // first reset after a bind: the full zero fill above must land before the row writes
                __syncthreads();
                __threadfence_block();
            }
            for (int q = lane; q < np; q += NT) {
                const PinRec pr = l.pins[q];
                write_pin_num<KIND>(p, e, pr);
                if (p.buf.all_pins_cat_feature) {
                    if (KIND == PCBENV_SPATIAL) {
                        double *f = p.buf.all_pins_cat_feature + ((size_t)e * p.pinRows + (pr.id & PIN_ID_MASK)) * 2;
                        f[0] = pr.net; f[1] = pr.comp;
                    } else if (!(pr.id & PIN_LOSER)) {
                        p.buf.all_pins_cat_feature[(size_t)e * p.pinRows + pr.comp * p.mp + (pr.id & PIN_ID_MASK)] = pr.net;
                    }
                }
            }
        }
        STAMP(19);
        if (KIND == PCBENV_SPATIAL) {
            if (p.buf.pin_grid) emit_zero(p.buf.pin_grid + (size_t)e * HW * p.K, (long long)HW * p.K, lane);  // S:1504
            if (p.buf.component_grid) {  // S:1677-1697 draw_components (unrotated rel coords; channel 0 == 1)
                const int cells = p.mh * p.mw, cgsz = cells * p.K, total = p.C * cgsz;
                unsigned char *cg = p.buf.component_grid + (size_t)e * total;
                // byte (cell, ch) = ch == 0 ? component exists : net ch-1 has a pin on the cell; each byte written once
                if ((total & 15) == 0 && (((uintptr_t)cg) & 15) == 0) {
                    for (int c16 = lane; c16 < total / 16; c16 += NT) {
                        const int bb = c16 * 16;
                        int cell = bb / p.K, ch = bb - cell * p.K;
                        u64 field = ((u64)netmask[cell] << 1) | (u64)(cell / cells < nc);  // bit ch = byte value of channel ch
                        u64 lo = 0ull, hi = 0ull;
                        #pragma unroll
                        for (int k = 0; k < 16; k++) {
                            const u64 bit = (field >> ch) & 1ull;
                            if (k < 8) lo |= bit << (8 * k); else hi |= bit << (8 * (k - 8));
                            if (++ch == p.K) { ch = 0; cell++; field = cell < p.C * cells ? (((u64)netmask[cell] << 1) | (u64)(cell / cells < nc)) : 0ull; }
                        }
                        STORE16((uint4 *)cg + c16, make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)));
                    }
                } else {
                    for (int i = lane; i < total; i += NT) {
                        const int cell = i / p.K, ch = i - cell * p.K;
                        cg[i] = (unsigned char)(ch == 0 ? (cell / cells < nc) : ((netmask[cell] >> (ch - 1)) & 1u));
                    }
                }
            }
        }
    }
    lds_sync();
}

template <int KIND, int WW, int NW>
__global__ __launch_bounds__(64 * NW) void k_reset(DevParams p, const unsigned char *__restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int e = blockIdx.x, lane = threadIdx.x;
    if (mask && !mask[e]) return;
    load_state(smem, p, e, lane);  // cursor / episode survive; the old pins tell which feature rows to clear
    Lds l = carve(smem, p);
    reset_env<KIND, WW>(p, l, e, lane);
    if (lane == 0) {
        l.hdr->pre_action = 0u;  // the mask changed under any presampled action
        p.buf.reward[e] = 0.0;
        p.buf.done[e] = 0;
        if (p.buf.info) { p.buf.info[2 * e] = nan(""); p.buf.info[2 * e + 1] = nan(""); }
    }
    store_state(smem, p, e, lane);
}

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

template <int KIND, int WW, int NW, bool ROUTES>
__global__ __launch_bounds__(64 * NW) void k_step(DevParams p, int *__restrict__ actions, int fmt, int sampled,
                                               u64 seed, u64 first_env, u64 step_index) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int e = blockIdx.x, lane = threadIdx.x;
    const int H = p.H, W = p.W, HW = H * W, plane = H * WW;
    STAMP_RT(30);
    STAMP(0);
    load_state(smem, p, e, lane);
    Lds l = carve(smem, p);
    const bool auto_reset = p.flags & PCBENV_FLAG_AUTO_RESET;
    STAMP(1);

    int o, x, y;
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
    if (done && auto_reset) reset_env<KIND, WW>(p, l, e, lane);  // rewrites every observation
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

// ==============================================================================================
// host side: the C ABI (include/pcbenv.h)
// ==============================================================================================
struct pcbenv {
    pcbenv_config cfg;
    int device;
    DevParams dp;
    bool bound;
    int threads;  // workgroup size (threads per environment)
    unsigned *scratch;  // 16 bytes of device memory for small read-backs
    unsigned loaded_slots;  // bit s = slot s loaded for all environments at least once
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
    if (c->queue_depth < 1 || c->queue_depth > 32) return fail(0, PCBENV_ELIMIT, "queue_depth must be in [1, 32]");
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
    d.stateStride = align16((long long)d.offPins + 8ll * d.P);
    d.instStride = align16(pcbenv_instance_stride(&c));
    // LDS scratch behind the state mirror
    d.ldsHf = (int)d.stateStride;
    // the class map (pin_grid emission) and the route segments (terminal reward) are never live together
    d.ldsCls = align16(d.ldsHf + d.H * d.WW * 8);
    d.ldsSeg = d.ldsCls;
    {
        const int beam = (is_pin_kind(c.kind) && c.reward_type != PCBENV_REWARD_CENTROID) ? BEAM_LDS_BYTES(c.max_num_nets, c.reward_beam_width) : 0;
        // class map of emit_pin_grid; at a reset the same zone holds the pin-id and net-mask tables (2 + 4 bytes per component cell)
        int cls = c.kind == PCBENV_SPATIAL ? (d.H * d.W > d.C * d.mp * 6 + 4 ? d.H * d.W : d.C * d.mp * 6 + 4) : 0, seg = is_pin_kind(c.kind) ? SEG_LDS_BYTES(d.P, env->threads / 64, beam) : 0;
        d.ldsBytes = align16(d.ldsCls + (cls > seg ? cls : seg));
    }
    { const char *ev = getenv("PCBENV_LDS_MIN"); if (ev && atoi(ev) > d.ldsBytes) d.ldsBytes = align16(atoi(ev)); }  // occupancy experiments
    DeviceGuard guard_(device);
    if (!guard_.ok) { int r = fail(0, PCBENV_EHIP, "hipSetDevice failed (no such device?)"); delete env; return r; }
    size_t sbytes = (size_t)d.stateStride * d.B, qbytes = (size_t)d.instStride * d.B * d.Q;
    if (hipMalloc((void **)&d.state, sbytes) != hipSuccess || hipMalloc((void **)&d.queue, qbytes ? qbytes : 16) != hipSuccess ) {
        int r = fail(0, PCBENV_EHIP, "hipMalloc failed");
        pcbenv_destroy(env);
        return r;
    }
#ifdef PCBENV_STAMPS
    { const char *ev = getenv("PCBENV_STAMPS"); if (ev && ev[0] == '1') { hipMalloc((void **)&d.dbg, (size_t)d.B * 32 * 8); hipMemset(d.dbg, 0, (size_t)d.B * 32 * 8); } }
#endif
    hipMemset(d.state, 0, sbytes);
    hipMemset(d.queue, 0, qbytes ? qbytes : 16);
    hipDeviceSynchronize();
    *out = env;
    return PCBENV_OK;
}

extern "C" void pcbenv_destroy(pcbenv *env) {
    if (!env) return;
    DeviceGuard guard_(env->device);
    if (env->dp.state) hipFree(env->dp.state);
    if (env->dp.queue) hipFree(env->dp.queue);
    if (env->scratch) hipFree(env->scratch);
    delete env;
}

extern "C" int pcbenv_bind_buffers(pcbenv *env, const pcbenv_buffers *b) {
    if (!env || !b) return fail(env, PCBENV_EINVAL, "null argument");
    if (!b->reward || !b->done) return fail(env, PCBENV_EINVAL, "reward and done buffers are required");
    env->dp.buf = *b;
    int k = env->cfg.kind;
    if (k != PCBENV_SPATIAL) { env->dp.buf.pin_grid = 0; env->dp.buf.component_grid = 0; }
    if (!is_pin_kind(k)) { env->dp.buf.all_pins_num_feature = 0; env->dp.buf.all_pins_cat_feature = 0; env->dp.buf.info = 0; }
    if (k != PCBENV_RECT) env->dp.buf.component_mask = 0;
    if (k == PCBENV_SQUARE) { env->dp.buf.all_components_feature = 0; env->dp.buf.placement_mask = 0; }
    env->dp.bind_gen += 1;  // feature tensors of these buffers are uninitialised: the next reset of each env fills them
    env->bound = true;
    return PCBENV_OK;
}

extern "C" int pcbenv_load_instances(pcbenv *env, const int32_t *env_ids, int32_t n, int32_t slot,
                                     const void *host_tables, void *stream) {
    if (!env || !host_tables) return fail(env, PCBENV_EINVAL, "null argument");
    const DevParams &d = env->dp;
    if (env->cfg.kind == PCBENV_SQUARE) return PCBENV_OK;  // the square env has no instance
    if (slot < 0 || slot >= d.Q || n < 0 || n > d.B) return fail(env, PCBENV_EINVAL, "slot or count out of range");
    DEVICE_GUARD(env);
    const long long src_stride = pcbenv_instance_stride(&env->cfg);
    hipStream_t s = (hipStream_t)stream;
    const unsigned char *src = (const unsigned char *)host_tables;
    // sanity-check the records (bad tables would index out of bounds on the device)
    for (int i = 0; i < n; i++) {
        const int32_t *h = (const int32_t *)(src + (size_t)i * src_stride);
        if (h[0] < 1 || h[0] > d.C || h[2] < 0 || h[2] > d.P || h[1] < 0 || h[1] > PCBENV_MAX_NETS)
            return fail(env, PCBENV_EINVAL, "instance record out of range");
        const unsigned char *cr = (const unsigned char *)h + 16, *pr = cr + 8 * (size_t)d.C;
        for (int c = 0; c < h[0]; c++)
            if (cr[8 * c] < 1 || cr[8 * c] > d.mh || cr[8 * c + 1] < 1 || cr[8 * c + 1] > d.mw) return fail(env, PCBENV_EINVAL, "component size out of range");
        int prev = 0;
        for (int q = 0; q < h[2]; q++) {
            int net = pr[8 * q + 2], comp = pr[8 * q + 3];
            if (comp >= h[0] || net >= h[1] || net < prev) return fail(env, PCBENV_EINVAL, "pin record out of range or not net-major");
            if (pr[8 * q] >= cr[8 * comp] || pr[8 * q + 1] >= cr[8 * comp + 1]) return fail(env, PCBENV_EINVAL, "pin outside its component");
            prev = net;
        }
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
    if (!env_ids && n == d.B) env->loaded_slots |= 1u << slot;
    else if (slot == 0 && env->loaded_slots == 0) env->loaded_slots |= 0;  // partial loads: caller's responsibility
    return PCBENV_OK;
}

template <int KIND> static int launch_reset(pcbenv *env, const uint8_t *mask, hipStream_t s) {
    const DevParams &d = env->dp;
#define LAUNCH_RESET(WW_, NW_) hipLaunchKernelGGL((k_reset<KIND, WW_, NW_>), dim3(d.B), dim3(64 * NW_), d.ldsBytes, s, d, mask)
    if (d.WW == 1) { if (env->threads == 64) LAUNCH_RESET(1, 1); else LAUNCH_RESET(1, 4); }
    else { if (env->threads == 64) LAUNCH_RESET(2, 1); else LAUNCH_RESET(2, 4); }
    return 0;
}
template <int KIND> static int launch_step(pcbenv *env, int *actions, int fmt, int sampled, u64 seed, u64 first_env,
                                           u64 step_index, hipStream_t s) {
    const DevParams &d = env->dp;
#define LAUNCH_STEP(WW_, NW_, RT_) hipLaunchKernelGGL((k_step<KIND, WW_, NW_, RT_>), dim3(d.B), dim3(64 * NW_), d.ldsBytes, s, d, actions, fmt, sampled, seed, first_env, step_index)
    constexpr bool PINK = (KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL);
    const bool routes = PINK && env->cfg.reward_type != PCBENV_REWARD_CENTROID;
    if (routes) {
        if (d.WW == 1) { if (env->threads == 64) LAUNCH_STEP(1, 1, PINK); else LAUNCH_STEP(1, 4, PINK); }
        else { if (env->threads == 64) LAUNCH_STEP(2, 1, PINK); else LAUNCH_STEP(2, 4, PINK); }
    } else {
        if (d.WW == 1) { if (env->threads == 64) LAUNCH_STEP(1, 1, false); else LAUNCH_STEP(1, 4, false); }
        else { if (env->threads == 64) LAUNCH_STEP(2, 1, false); else LAUNCH_STEP(2, 4, false); }
    }
    return 0;
}
static int dispatch_step(pcbenv *env, int *actions, int fmt, int sampled, u64 seed, u64 first_env, u64 step_index, hipStream_t s) {
    switch (env->cfg.kind) {
    case PCBENV_SQUARE: return launch_step<PCBENV_SQUARE>(env, actions, fmt, sampled, seed, first_env, step_index, s);
    case PCBENV_RECT: return launch_step<PCBENV_RECT>(env, actions, fmt, sampled, seed, first_env, step_index, s);
    case PCBENV_PIN: return launch_step<PCBENV_PIN>(env, actions, fmt, sampled, seed, first_env, step_index, s);
    default: return launch_step<PCBENV_SPATIAL>(env, actions, fmt, sampled, seed, first_env, step_index, s);
    }
}

static int pre_launch(pcbenv *env) {
    if (!env) return fail(0, PCBENV_EINVAL, "null handle");
    if (!env->bound) return fail(env, PCBENV_ESTATE, "pcbenv_bind_buffers has not been called");
    return PCBENV_OK;
}

static int check_queue(pcbenv *env);
extern "C" int pcbenv_reset(pcbenv *env, const uint8_t *mask_dev, void *stream) {
    int rc = pre_launch(env);
    if (rc) return rc;
    DEVICE_GUARD(env);
    rc = check_queue(env);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (env->cfg.kind) {
    case PCBENV_SQUARE: launch_reset<PCBENV_SQUARE>(env, mask_dev, s); break;
    case PCBENV_RECT: launch_reset<PCBENV_RECT>(env, mask_dev, s); break;
    case PCBENV_PIN: launch_reset<PCBENV_PIN>(env, mask_dev, s); break;
    default: launch_reset<PCBENV_SPATIAL>(env, mask_dev, s); break;
    }
    HIP_TRY(env, hipGetLastError());
    return PCBENV_OK;
}

static int check_queue(pcbenv *env) {
    if (env->cfg.kind != PCBENV_SQUARE && env->loaded_slots != (env->dp.Q >= 32 ? ~0u : ((1u << env->dp.Q) - 1u)))
        return fail(env, PCBENV_ESTATE, "every queue slot must be loaded (pcbenv_load_instances for all environments) first");
    return PCBENV_OK;
}

extern "C" int pcbenv_step(pcbenv *env, const int32_t *actions_dev, int32_t fmt, void *stream) {
    int rc = pre_launch(env);
    if (rc) return rc;
    DEVICE_GUARD(env);
    if (!actions_dev) return fail(env, PCBENV_EINVAL, "null actions");
    if (fmt != PCBENV_ACTION_TUPLE && fmt != PCBENV_ACTION_FLAT) return fail(env, PCBENV_EINVAL, "unknown action format");
    dispatch_step(env, (int *)actions_dev, fmt, 0, 0, 0, 0, (hipStream_t)stream);
    HIP_TRY(env, hipGetLastError());
    return PCBENV_OK;
}

extern "C" int pcbenv_step_sampled(pcbenv *env, int32_t *actions_out_dev, int32_t fmt, uint64_t seed,
                                   uint64_t first_env_index, uint64_t step_index, void *stream) {
    int rc = pre_launch(env);
    if (rc) return rc;
    DEVICE_GUARD(env);
    if (!actions_out_dev) return fail(env, PCBENV_EINVAL, "null actions");
    if (fmt != PCBENV_ACTION_TUPLE && fmt != PCBENV_ACTION_FLAT) return fail(env, PCBENV_EINVAL, "unknown action format");
    dispatch_step(env, actions_out_dev, fmt, 1, seed, first_env_index, step_index, (hipStream_t)stream);
    HIP_TRY(env, hipGetLastError());
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

extern "C" int64_t pcbenv_state_bytes(const pcbenv *env) {
    return env ? (int64_t)env->dp.stateStride * env->dp.B : 0;
}
extern "C" int pcbenv_get_state(pcbenv *env, void *host_dst, void *stream) {
    if (!env || !host_dst) return fail(env, PCBENV_EINVAL, "null argument");
    DEVICE_GUARD(env);
    const size_t sb = (size_t)env->dp.stateStride * env->dp.B;
    HIP_TRY(env, hipMemcpyAsync(host_dst, env->dp.state, sb, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(env, hipStreamSynchronize((hipStream_t)stream));
    return PCBENV_OK;
}
extern "C" int pcbenv_set_state(pcbenv *env, const void *host_src, void *stream) {
    if (!env || !host_src) return fail(env, PCBENV_EINVAL, "null argument");
    DEVICE_GUARD(env);
    const size_t sb = (size_t)env->dp.stateStride * env->dp.B;
    HIP_TRY(env, hipMemcpyAsync(env->dp.state, host_src, sb, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(env, hipStreamSynchronize((hipStream_t)stream));
    return PCBENV_OK;
}

extern "C" int pcbenv_rollout_sampled(pcbenv *env, int32_t *actions_out_dev, int32_t fmt, int32_t num_steps,
                                      uint64_t seed, uint64_t first_env_index, uint64_t step_index0, void *stream) {
    int rc = pre_launch(env);
    if (rc) return rc;
    DEVICE_GUARD(env);
    if (!actions_out_dev || num_steps < 0) return fail(env, PCBENV_EINVAL, "bad rollout arguments");
    if (fmt != PCBENV_ACTION_TUPLE && fmt != PCBENV_ACTION_FLAT) return fail(env, PCBENV_EINVAL, "unknown action format");
    const size_t per_step = (size_t)env->dp.B * (fmt == PCBENV_ACTION_TUPLE ? 3 : 1);
    for (int t = 0; t < num_steps; t++)
        dispatch_step(env, actions_out_dev + per_step * (size_t)t, fmt, 1, seed, first_env_index, step_index0 + (uint64_t)t, (hipStream_t)stream);
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
    return hipMemcpy(host, env->dp.dbg, (size_t)env->dp.B * 32 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif
