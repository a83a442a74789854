// pcb_sampler.h -- uniform legal-action sampler (k_sample and the fused sampler of k_step)
// Part of libpcbenv.so's single translation unit (included by pcbenv_kernels.hip); CDNA4 / gfx950 only.

// ----------------------------------------------------------------------------------------------
// uniform legal-action sampler (rollout driver; agent/random/random_policy_*.py counterpart)
// ----------------------------------------------------------------------------------------------
static __device__ inline u64 mix64(u64 z) {  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static __device__ inline int select_bit(u64 w, int k) {  // position of the k-th (0-based) set bit: binary search on popcounts
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
static __device__ inline void sample_action(const u64 *vm, const DevParams &p, int genv, int lane, u64 seed, u64 step_index,
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

