// pcb_team_io.h -- team-size dependent basics: LDS barrier, workgroup any(), 16-byte plane emission, the legal-mask window fold
// Textually included INSIDE `template <int TN> struct Team` (pcb_team.h): NT == TN threads work on one environment.
// Team barrier that waits for LDS traffic only.  __syncthreads() also drains the global stores in flight
// (s_waitcnt vmcnt(0)), which would serialise the observation write stream between kernel phases.
// A team of ONE wavefront needs no s_barrier: its LDS traffic executes in program order, only the compiler (and the
// lgkmcnt wait the fence emits) stand between a write and a read of another lane -- and it MUST not execute one when
// four independent one-wavefront teams share a workgroup (k_step_mixed), where their barrier counts differ.
static __device__ inline void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (NT > WAVE) __builtin_amdgcn_s_barrier(); else __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// The same, after this team's global stores have left the wavefronts (write-after-write on bytes another lane or
// wavefront of the team stored earlier in the launch).
static __device__ inline void store_drain_sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_sync();
}
// any() over the team; `flag` is an LDS word
static __device__ inline bool block_any(bool v, unsigned *flag) {
    if (NT == WAVE) return __any(v);
    if (threadIdx.x == 0) *flag = 0;
    lds_sync();
    if (__any(v) && (threadIdx.x & 63) == 0) *flag = 1;
    lds_sync();
    return *flag != 0;
}

// Write one H x W uint8 plane (0/1) from bit rows in LDS: 16 bytes per lane, 1 KiB per wave instruction.
// Rows [r0, r1) only (full plane: 0, H).
template <int WW, bool STREAM> static __device__ inline void emit_plane_(unsigned char *dst, const u64 *bits, int r0, int r1, int W, int lane) {
    if ((W & 15) == 0 && (((uintptr_t)dst) & 15) == 0) {
        const ObsDst d = obs_dst(dst, (long long)r1 * W);
        const int sh = (W & (W - 1)) == 0 ? __ffs(W) - 1 : -1;
        for (int c = r0 * W / 16 + lane; c < r1 * W / 16; c += NT) {
            int cell = c * 16, r = sh >= 0 ? cell >> sh : cell / W, col = cell - r * W;
            unsigned b = (unsigned)(bits[r * WW + (col >> 6)] >> (col & 63)) & 0xFFFFu;
            STORE16<STREAM>(d, (unsigned)cell, expand16(b));
        }
    } else {  // odd widths (the reference's small test grids): byte path
        for (int i = r0 * W + lane; i < r1 * W; i += NT) {
            int r = i / W, col = i - r * W;
            dst[i] = (unsigned char)((bits[r * WW + (col >> 6)] >> (col & 63)) & 1ull);
        }
    }
}
template <bool STREAM> static __device__ inline void emit_zero_(unsigned char *dst, long long bytes, int lane) {
    if ((bytes & 15) == 0 && (((uintptr_t)dst) & 15) == 0) {
        const ObsDst d = obs_dst(dst, bytes);
        for (int c = lane; c < (int)(bytes / 16); c += NT) STORE16<STREAM>(d, (unsigned)c * 16u, make_uint4(0, 0, 0, 0));
    } else {
        for (long long i = lane; i < bytes; i += NT) dst[i] = 0;
    }
}
// The same plane to two destinations (action_mask[o] and action_mask[o + 2] of the pin environments are equal,
// S:1852-1853): the bits are read and expanded once, stored twice.
template <int WW, bool STREAM> static __device__ inline void emit_plane2_(unsigned char *dst, unsigned char *dst2, const u64 *bits, int H, int W, int lane) {
    if ((W & 15) == 0 && (((uintptr_t)dst) & 15) == 0 && (((uintptr_t)dst2) & 15) == 0) {
        const ObsDst d = obs_dst(dst, (long long)H * W), d2 = obs_dst(dst2, (long long)H * W);
        const int sh = (W & (W - 1)) == 0 ? __ffs(W) - 1 : -1;
        for (int c = lane; c < H * W / 16; c += NT) {
            int cell = c * 16, r = sh >= 0 ? cell >> sh : cell / W, col = cell - r * W;
            unsigned b = (unsigned)(bits[r * WW + (col >> 6)] >> (col & 63)) & 0xFFFFu;
            const uint4 v = expand16(b);
            STORE16<STREAM>(d, (unsigned)cell, v);
            STORE16<STREAM>(d2, (unsigned)cell, v);
        }
    } else {
        emit_plane_<WW, STREAM>(dst, bits, 0, H, W, lane);
        emit_plane_<WW, STREAM>(dst2, bits, 0, H, W, lane);
    }
}
// the policy is chosen once per plane (wave-uniform branch), not per store
template <int WW> static __device__ inline void emit_plane(unsigned char *dst, const u64 *bits, int r0, int r1, int W, int lane, bool stream) {
    if (stream) emit_plane_<WW, true>(dst, bits, r0, r1, W, lane); else emit_plane_<WW, false>(dst, bits, r0, r1, W, lane);
}
template <int WW> static __device__ inline void emit_plane2(unsigned char *dst, unsigned char *dst2, const u64 *bits, int H, int W, int lane, bool stream) {
    if (stream) emit_plane2_<WW, true>(dst, dst2, bits, H, W, lane); else emit_plane2_<WW, false>(dst, dst2, bits, H, W, lane);
}
static __device__ inline void emit_zero(unsigned char *dst, long long bytes, int lane, bool stream) {
    if (stream) emit_zero_<true>(dst, bytes, lane); else emit_zero_<false>(dst, bytes, lane);
}

// Legal-placement bit mask for a ph x pw window (R:526-567, S:1792-1835):
// vm[r] bit j = 1 iff r <= H-ph and j <= W-pw and occ[r..r+ph-1][j..j+pw-1] is empty.
// Returns (wave-uniform) whether any bit is set.
template <int WW>
static __device__ inline bool window_mask(const u64 *occ, u64 *hf, u64 *vm, int H, int W, int ph, int pw, int lane, unsigned *flag) {
#ifndef PCBENV_FOLD_LDS
    if (WW == 1 && NT == WAVE && H <= WAVE) {
        // One row per lane: the vertical OR over ph rows by log-step doubling across lanes, like the horizontal one across
        // bits -- no staging of the folded rows in LDS, no barrier, at most three cross-lane steps for ph <= 8.
        u64 a = lane < H ? hfold<1>(Row<1>::load(occ + lane), pw).a : 0ull;
        int s = 1;
        while (2 * s <= ph) { a |= lane_down(a, s, lane); s *= 2; }
        if (s < ph) a |= lane_down(a, ph - s, lane);
        const u64 v = (lane + ph <= H) ? Row<1>{a}.free_below(W - pw + 1).a : 0ull;
        if (lane < H) vm[lane] = v;
        return __any(v != 0ull);
    }
#endif
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

