// pcb_observe.h -- state block staging through LDS, legal-mask fold + observation emission, pin_grid, feature rows, terminal reward
// Part of libpcbenv.so's single translation unit (included by pcbenv_kernels.hip); CDNA4 / gfx950 only.

// ----------------------------------------------------------------------------------------------
// shared pieces of reset / step
// ----------------------------------------------------------------------------------------------
struct Lds {
    EnvHdr *hdr; u64 *occ, *vm; CompRec *comps; PinRec *pins; unsigned char *rank;
    u64 *hf; unsigned char *cls; double *seg;
};
// Row of environment e in the [num_slots, B, ...] output tensors for the slot a step writes (DevParams::slot).
static __device__ inline int out_row(const DevParams &p, int slot, int e) { return slot * p.B + e; }
static __device__ inline Lds carve(unsigned char *smem, const DevParams &p) {
    Lds l;
    l.hdr = (EnvHdr *)smem;
    l.occ = (u64 *)(smem + p.offOcc);
    l.vm = (u64 *)(smem + p.offVm);
    l.comps = (CompRec *)(smem + p.offComps);
    l.pins = (PinRec *)(smem + p.offPins);
    l.rank = smem + p.offRank;  // rank[q] = position of pin q among the pins of its component (self.pins order)
    l.hf = (u64 *)(smem + p.ldsHf);
    l.cls = smem + p.ldsCls;
    l.seg = (double *)(smem + p.ldsSeg);
    return l;
}
static __device__ inline void load_state(unsigned char *smem, const DevParams &p, int e, int lane) {
    const uint4 *src = (const uint4 *)(p.state + (size_t)e * p.stateStride);
    uint4 *dst = (uint4 *)smem;
    const int n = (int)(p.stateStride / 16);
    if (n <= NT) {  // small blocks (c2): one chunk per lane
        if (lane < n) dst[lane] = src[lane];
    } else {
        // Every wavefront of the launch starts here: all of a lane's chunks are requested before the first one is awaited
        // (one memory round trip for the block instead of one per 1 KiB), up to four per lane (c3 / c4: 3, c5: 2), a plain
        // loop beyond.
        uint4 t[4];
        #pragma unroll
        for (int k = 0; k < 4; k++) { const int i = lane + k * NT; t[k] = i < n ? src[i] : make_uint4(0, 0, 0, 0); }
        #pragma unroll
        for (int k = 0; k < 4; k++) { const int i = lane + k * NT; if (i < n) dst[i] = t[k]; }
        for (int i = lane + 4 * NT; i < n; i += NT) dst[i] = src[i];
    }
    lds_sync();
}
static __device__ inline void store_state(const unsigned char *smem, const DevParams &p, int e, int lane) {
    lds_sync();
    uint4 *dst = (uint4 *)(p.state_out + (size_t)e * p.stateStride);
    const uint4 *src = (const uint4 *)smem;
    // plain write-back stores: environment e runs on the same XCD in every launch (xcd_contiguous_env), so its state block is an L2 hit next step
    for (int i = lane; i < (int)(p.stateStride / 16); i += NT) dst[i] = src[i];
}

// Marginals of the legal mask for factorised policies (factorized_action_distributions.py:358, :401): per
// orientation "any legal cell" and per (orientation, row) "any legal column", read off the bit rows in LDS.
template <int KIND, int WW> static __device__ inline void emit_marginals(const DevParams &p, Lds &l, int row, int lane) {
    if (!p.buf.mask_rows && !p.buf.mask_orientation) return;
    const int H = p.H, plane = H * WW, O = p.O;
    for (int i = lane; i < O * H; i += NT) {
        const int o = i / H, r = i - o * H;
        const u64 *bits = l.vm + (o & 1) * plane + r * WW;
        bool a = false;
        for (int w = 0; w < WW; w++) a |= bits[w] != 0;
        if (p.buf.mask_rows) p.buf.mask_rows[(size_t)row * O * H + i] = a ? 1 : 0;
    }
    if (p.buf.mask_orientation) {
        for (int o = (int)(lane / WAVE); o < O; o += NT / WAVE) {  // one wavefront per orientation
            bool a = false;
            for (int i = (lane & 63); i < plane; i += WAVE) a |= l.vm[(o & 1) * plane + i] != 0;
            a = __any(a);
            if ((lane & 63) == 0) p.buf.mask_orientation[(size_t)row * O + o] = a ? 1 : 0;
        }
    }
}

// Mask of the current component (or zeros) into l.vm, both orientations, and -- when `emit` -- the grid rows
// [gr0, gr1) and the action_mask planes, each written as soon as its bits exist so that the HBM write stream
// starts before the second orientation is folded.  Returns "some action is legal".
template <int KIND, int WW>
static __device__ inline bool mask_and_emit(const DevParams &p, Lds &l, int row, int lane, bool emit, int gr0, int gr1) {
    const int H = p.H, W = p.W, HW = H * W, plane = H * WW;
    const int cur = l.hdr->cur;
    unsigned char *m = (emit && p.buf.action_mask) ? p.buf.action_mask + (size_t)row * p.O * HW : 0;
    if (emit && p.buf.grid) emit_plane<WW>(p.buf.grid + (size_t)row * HW, l.occ, gr0, gr1, W, lane, p.stream_stores);
    bool any = false;
    if (KIND == PCBENV_SQUARE) {
        any = window_mask<WW>(l.occ, l.hf, l.vm, H, W, p.component_n, p.component_n, lane, &l.hdr->flag);
        if (m) emit_plane<WW>(m, l.vm, 0, H, W, lane, p.stream_stores);
        if (emit) emit_marginals<KIND, WW>(p, l, row, lane);
        return any;
    }
    const bool four = (KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL);  // S:1852-1853 mask[2] = mask[0], mask[3] = mask[1]
    if (cur >= 0) {
        const int h = l.comps[cur].h, w = l.comps[cur].w;
        any = window_mask<WW>(l.occ, l.hf, l.vm, H, W, h, w, lane, &l.hdr->flag);
        if (m) { if (four) emit_plane2<WW>(m, m + 2 * HW, l.vm, H, W, lane, p.stream_stores); else emit_plane<WW>(m, l.vm, 0, H, W, lane, p.stream_stores); }
        if (h == w) {
            for (int i = lane; i < plane; i += NT) l.vm[plane + i] = l.vm[i];
            lds_sync();
        } else {
            any |= window_mask<WW>(l.occ, l.hf, l.vm + plane, H, W, w, h, lane, &l.hdr->flag);
        }
    } else {
        for (int i = lane; i < 2 * plane; i += NT) l.vm[i] = 0ull;
        lds_sync();
        if (m) { if (four) emit_plane2<WW>(m, m + 2 * HW, l.vm, H, W, lane, p.stream_stores); else emit_plane<WW>(m, l.vm, 0, H, W, lane, p.stream_stores); }
    }
    if (m) { if (four) emit_plane2<WW>(m + HW, m + 3 * HW, l.vm + plane, H, W, lane, p.stream_stores); else emit_plane<WW>(m + HW, l.vm + plane, 0, H, W, lane, p.stream_stores); }
    if (emit) emit_marginals<KIND, WW>(p, l, row, lane);
    return any;
}

// S:1663-1675 draw_pins: class map (CLS_EMPTY, 1 occupied without pin, n+2 pin of net n) -> one-hot[:, :, 1:];
// rows [r0, r1) of the (H, W, K) tensor (a step only changes the rows of the placed rectangle).  An empty cell's class
// is 0x80, not 0: "byte of the cell's 1 = first byte of the cell + class - 1" then lands far outside any chunk without a
// test of its own.
#define CLS_EMPTY 0x80u
template <int WW> static __device__ inline void emit_pin_grid(const DevParams &p, Lds &l, int row, int lane, int r0, int r1) {
    if (!p.buf.pin_grid) return;
    const int W = p.W, HW = p.H * W, K = p.K;
    const int c0 = r0 * W, c1 = r1 * W;
    unsigned char *dst = p.buf.pin_grid + (size_t)row * HW * K;
    const long long b0 = (long long)c0 * K, b1 = (long long)c1 * K;
    if ((W & 15) == 0) {  // sixteen cells per lane and trip: one occupancy word read, one 16-byte write (was: a read, a division and a byte per cell)
        const int sh = (W & (W - 1)) == 0 ? __ffs(W) - 1 : -1;
        for (int ch = c0 / 16 + lane; ch < c1 / 16; ch += NT) {
            const int cell = ch * 16, r = sh >= 0 ? cell >> sh : cell / W, c = cell - r * W;
            uint4 v = expand16((unsigned)(l.occ[r * WW + (c >> 6)] >> (c & 63)) & 0xFFFFu);  // bytes 0 / 1 -> CLS_EMPTY / 1
            v.x |= (v.x ^ 0x01010101u) << 7; v.y |= (v.y ^ 0x01010101u) << 7; v.z |= (v.z ^ 0x01010101u) << 7; v.w |= (v.w ^ 0x01010101u) << 7;
            *(uint4 *)(l.cls + cell) = v;
        }
    } else {
        for (int i = c0 + lane; i < c1; i += NT) {
            int r = i / W, c = i - r * W;
            l.cls[i] = (unsigned char)(((l.occ[r * WW + (c >> 6)] >> (c & 63)) & 1ull) ? 1u : CLS_EMPTY);
        }
    }
    lds_sync();
    for (int q = lane; q < l.hdr->npins; q += NT) {
        const PinRec pr = l.pins[q];
        if (pr.abs_x >= r0 && pr.abs_x < r1 && pr.abs_y >= 0) l.cls[pr.abs_x * W + pr.abs_y] = (unsigned char)(pr.net + 2);
    }
    lds_sync();
    if ((b0 & 15) == 0 && (b1 & 15) == 0 && (((uintptr_t)dst) & 15) == 0) {
        const ObsDst d = obs_dst(dst, b1);
        // every cell owns K consecutive bytes with at most one 1 (at class-1): visit the <= 16/K + 2 cells a
        // 16-byte chunk overlaps and drop their 1-bytes into two 64-bit halves
        // first cell of a lane's first chunk and the chunk's offset inside it by one division; from chunk to chunk (16 * NT
        // bytes on) by an add and a conditional carry (two quarter-rate multiplies per chunk before)
        const int first = ((int)(b0 / 16) + lane) * 16, dcell = 16 * NT / K, drem = 16 * NT - dcell * K;
        auto chunks = [&](auto stream_tag, auto cells_tag) {  // store policy and cells per chunk fixed per call, not per store
            constexpr int NC = decltype(cells_tag)::value;
            int cell0 = first / K, rem = first - cell0 * K;  // chunk byte bb = cell0 * K + rem, 0 <= rem < K
            for (int c = (int)(b0 / 16) + lane; c < (int)(b1 / 16); c += NT) {
                const int bb = c * 16;
                int cell = cell0;
                const int rem_now = rem;
                cell0 += dcell; rem += drem;
                if (rem >= K) { rem -= K; cell0++; }
                unsigned ones = 0u;  // bit b = byte b of the chunk is 1
                if (K >= 5) {  // a chunk overlaps at most ceil((15 + K) / K) cells -- 3 from K = 8 on (c4 / c5: K = 9), 4 below: their classes are read together (one LDS round trip per store, not one per cell)
                    unsigned cl[NC];
                    #pragma unroll
                    for (int j = 0; j < NC; j++) cl[j] = cell + j < c1 ? (unsigned)l.cls[cell + j] : CLS_EMPTY;
                    const int rel = -rem_now - 1;  // = cell * K - bb - 1; byte of cell j's 1 inside the chunk: rel + j * K + class
                    #pragma unroll
                    for (int j = 0; j < NC; j++) {  // branch-free: an add, a compare, a shift, a select and an or per cell
                        const unsigned off = (unsigned)(rel + j * K + (int)cl[j]);
                        ones |= off < 16u ? 1u << off : 0u;
                    }
                } else {
                    for (int base = bb - rem_now; base < bb + 16 && cell < c1; base += K, cell++) {
                        const unsigned off = (unsigned)(base + (int)l.cls[cell] - 1 - bb);  // byte of this cell's 1 inside the chunk
                        if (off < 16u) ones |= 1u << off;
                    }
                }
                STORE16<decltype(stream_tag)::value>(d, (unsigned)bb, expand16(ones));
            }
        };
        typedef std::integral_constant<int, 3> three; typedef std::integral_constant<int, 4> four;
        if (K >= 8) { if (p.stream_stores) chunks(std::true_type{}, three{}); else chunks(std::false_type{}, three{}); }
        else { if (p.stream_stores) chunks(std::true_type{}, four{}); else chunks(std::false_type{}, four{}); }
    } else {
        for (long long i = b0 + lane; i < b1; i += NT) {
            int cell = (int)(i / K), ch = (int)(i - (long long)cell * K);
            dst[i] = (unsigned char)(l.cls[cell] == ch + 1);
        }
    }
}

// Feature rows of one pin (P:72-103 / S:70-104 Pin.calculate_feature): [rel_x, rel_y, abs_x, abs_y]
template <int KIND> static __device__ inline void write_pin_num(const DevParams &p, int row_, const PinRec &pr) {
    if (!p.buf.all_pins_num_feature) return;
    int row;
    if (KIND == PCBENV_SPATIAL) row = pr.id & PIN_ID_MASK;
    else { if (pr.id & PIN_LOSER) return; row = pr.comp * p.mp + (pr.id & PIN_ID_MASK); }
    double *f = p.buf.all_pins_num_feature + ((size_t)row_ * p.pinRows + row) * 4;
    f[0] = pr.rel_x; f[1] = pr.rel_y; f[2] = pr.abs_x; f[3] = pr.abs_y;
}

// Scratch tables in the class-map zone (free between two emit_pin_grid calls), spatial env only:
// pid[c][k] = global id of the k-th pin of component c in self.pins order (0xFFFF = none) -- the tail of
// all_components_feature (S:203-239); netmask[c][rel_x][rel_y] = nets with a pin on that cell of the component at
// its UNROTATED relative coordinates -- draw_components (S:1677-1697) runs at reset only, so component_grid never
// shows the in-place rotation of place_component (quirk Q4): for a placed component the rotation is undone here
// with the orientation kept in its record.
struct PinTables { unsigned short *pid; unsigned *netmask; };
static __device__ inline PinTables pin_tables(const DevParams &p, Lds &l) {
    PinTables t;
    t.pid = (unsigned short *)l.cls;
    t.netmask = (unsigned *)(l.cls + ((p.C * p.mp * 2 + 3) & ~3));
    return t;
}
static __device__ inline void build_pin_tables(const DevParams &p, Lds &l, int lane) {
    const PinTables t = pin_tables(p, l);
    const int np = l.hdr->npins;
    lds_sync();
    for (int i = lane; i < p.C * p.mp; i += NT) { t.pid[i] = 0xFFFFu; t.netmask[i] = 0u; }
    lds_sync();
    for (int q = lane; q < np; q += NT) {
        const PinRec pr = l.pins[q];
        const CompRec cr = l.comps[pr.comp];
        int rx = pr.rel_x, ry = pr.rel_y;
        if (cr.px >= 0) {  // inverse of S:149-190 place_component
            const int ax = rx, ay = ry;
            if (cr.o == 1) { rx = cr.h - 1 - ay; ry = ax; }
            else if (cr.o == 2) { rx = cr.h - 1 - ax; ry = cr.w - 1 - ay; }
            else if (cr.o == 3) { rx = ay; ry = cr.w - 1 - ax; }
        }
        t.pid[pr.comp * p.mp + l.rank[q]] = (unsigned short)(pr.id & PIN_ID_MASK);
        atomicOr(&t.netmask[(int)pr.comp * p.mp + rx * p.mw + ry], 1u << pr.net);
    }
    lds_sync();
}
// S:1677-1697 draw_components from the tables above: byte (cell, ch) = ch == 0 ? component exists : net ch-1 has a
// pin on the cell; each byte written once.
// chunk [bb, bb + 16) of a `total`-byte tensor row (total a multiple of 4): whole, or -- the last one -- the dwords inside the row
static __device__ inline void store16_or_tail(const ObsDst &d, unsigned char *dst, int bb, int total, uint4 v, bool stream) {
    if (bb + 16 <= total) { STORE16_dyn(d, (unsigned)bb, v, stream); return; }
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    #pragma unroll
    for (int j = 0; j < 4; j++) if (bb + 4 * j < total) *(unsigned *)(dst + bb + 4 * j) = w[j];
}
static __device__ inline void emit_component_grid_to(const DevParams &p, Lds &l, unsigned char *cg, int lane) {
    const PinTables t = pin_tables(p, l);
    const int nc = l.hdr->ncomp;
    const int cells = p.mh * p.mw, cgsz = cells * p.K, total = p.C * cgsz;
    // 16-byte chunks at 4-byte alignment (c4: 2 916 bytes per environment, every row a multiple of 4 only -- gfx9 under HSA
    // runs in unaligned-access mode, where a 16-byte access needs dword alignment), the last chunk's dwords one by one
    if ((total & 3) == 0 && (((uintptr_t)cg) & 3) == 0) {
        const ObsDst d = obs_dst(cg, total);
        for (int c16 = lane; c16 < (total + 15) / 16; c16 += NT) {
            const int bb = c16 * 16;
            int cell = bb / p.K, ch = bb - cell * p.K;
            u64 field = ((u64)t.netmask[cell] << 1) | (u64)(cell / cells < nc);  // bit ch = byte value of channel ch
            u64 lo = 0ull, hi = 0ull;
            #pragma unroll
            for (int k = 0; k < 16; k++) {
                const u64 bit = (field >> ch) & 1ull;
                if (k < 8) lo |= bit << (8 * k); else hi |= bit << (8 * (k - 8));
                if (++ch == p.K) { ch = 0; cell++; field = cell < p.C * cells ? (((u64)t.netmask[cell] << 1) | (u64)(cell / cells < nc)) : 0ull; }
            }
            store16_or_tail(d, cg, bb, total, make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)), p.stream_stores);
        }
    } else {
        for (int i = lane; i < total; i += NT) {
            const int cell = i / p.K, ch = i - cell * p.K;
            cg[i] = (unsigned char)(ch == 0 ? (cell / cells < nc) : ((t.netmask[cell] >> (ch - 1)) & 1u));
        }
    }
}

static __device__ inline void emit_component_grid(const DevParams &p, Lds &l, int row, int lane) {
    if (!p.buf.component_grid) return;
    emit_component_grid_to(p, l, p.buf.component_grid + (size_t)row * p.C * p.mh * p.mw * p.K, lane);
}

// The episode-constant part of a spatial environment's trajectory slot, kept per environment in library memory so that a
// step of the trajectory layout copies it instead of rebuilding the pin tables: [compact all_components_feature, C x F
// int16 with x = y = -1 | component_grid].  Written by the reset that starts the episode (tagged with the episode number;
// a restored checkpoint invalidates the tags), used by steps whose slot takes no float64 all_components_feature.
static __device__ inline bool feat_cache_valid(const DevParams &p, const Lds &l, int e) {
    return p.feat_cache && !p.buf.all_components_feature && p.feat_cache_tag[e] == l.hdr->episode;
}
static __device__ inline void feat_cache_fill(const DevParams &p, Lds &l, int e, int lane) {  // spatial, build_pin_tables() has run
    if (!p.feat_cache) return;
    unsigned char *base = p.feat_cache + (size_t)e * p.featCacheStride;
    const PinTables t = pin_tables(p, l);
    const int nc = l.hdr->ncomp;
    short *cf = (short *)base;
    int c = lane / p.F, k = lane - c * p.F;
    const int dc = NT / p.F, dk = NT - dc * p.F;
    for (int i = lane; i < p.C * p.F; i += NT) {
        int v = 0;
        if (c < nc) {
            const CompRec cr = l.comps[c];
            if (k == 0) v = cr.h; else if (k == 1) v = cr.w; else if (k == 2 || k == 3) v = -1;
            else if (k == 4) v = cr.h * cr.w;
            else { const unsigned id = t.pid[c * p.mp + k - 5]; v = id == 0xFFFFu ? -1 : (int)id; }
        }
        cf[i] = (short)v;
        c += dc; k += dk;
        if (k >= p.F) { k -= p.F; c++; }
    }
    emit_component_grid_to(p, l, base + p.featCacheCg, lane);
    if (lane == 0) p.feat_cache_tag[e] = l.hdr->episode;
}
// A step's copy: compact all_components_feature with the components' current positions patched in, component_grid as is.
static __device__ inline void feat_cache_emit(const DevParams &p, Lds &l, int e, int row, int lane) {
    const unsigned char *base = p.feat_cache + (size_t)e * p.featCacheStride;
    if (p.cbuf.all_components_feature) {
        const short *src = (const short *)base;
        short *cf = p.cbuf.all_components_feature + (size_t)row * p.C * p.F;
        int c = lane / p.F, k = lane - c * p.F;
        const int dc = NT / p.F, dk = NT - dc * p.F;
        for (int i = lane; i < p.C * p.F; i += NT) {
            short v = src[i];
            if (k == 2 && c < l.hdr->ncomp) v = l.comps[c].px;
            if (k == 3 && c < l.hdr->ncomp) v = l.comps[c].py;
            cf[i] = v;
            c += dc; k += dk;
            if (k >= p.F) { k -= p.F; c++; }
        }
    }
    if (p.buf.component_grid) {
        const int total = p.C * p.mh * p.mw * p.K;
        unsigned char *cg = p.buf.component_grid + (size_t)row * total;
        const uint4 *src = (const uint4 *)(base + p.featCacheCg);  // (16-byte aligned, and padded to whole chunks)
        if ((total & 3) == 0 && (((uintptr_t)cg) & 3) == 0) {
            const ObsDst d = obs_dst(cg, total);
            for (int c16 = lane; c16 < (total + 15) / 16; c16 += NT) store16_or_tail(d, cg, c16 * 16, total, src[c16], p.stream_stores);
        } else {
            for (int i = lane; i < total; i += NT) cg[i] = base[p.featCacheCg + i];
        }
    }
}

// Every float64 feature tensor of environment `row`, whole, from the state in LDS -- each element written exactly
// once (no write-after-write inside the launch).  Used when the destination holds nothing of this environment:
// the trajectory layout (num_slots > 1), where every step lands in a fresh slot.  all_components_feature
// (R:60-79, S:203-239), placement / component masks (S:1445-1451, :1592-1602; R:275-298), pin features
// (P:72-103 / S:70-104, quirk Q1 for the pin env, S:1520 last row).  Spatial: build_pin_tables() must have run.
// The compact twins of the float64 feature tensors (pcbenv_compact_features): same values, same "each element written
// exactly once" discipline, 2 / 1 bytes per element.  Spatial: build_pin_tables() must have run.
template <int KIND> static __device__ inline void emit_features_compact(const DevParams &p, Lds &l, int row, int lane, bool comp_from_cache = false) {
    const int nc = l.hdr->ncomp, np = l.hdr->npins, cur = l.hdr->cur;
    if (p.cbuf.all_components_feature && !comp_from_cache) {
        const PinTables t = pin_tables(p, l);
        short *cf = p.cbuf.all_components_feature + (size_t)row * p.C * p.F;
        int c = lane / p.F, k = lane - c * p.F;
        const int dc = NT / p.F, dk = NT - dc * p.F;
        for (int i = lane; i < p.C * p.F; i += NT) {
            int v = 0;
            if (c < nc) {
                const CompRec cr = l.comps[c];
                if (k == 0) v = cr.h; else if (k == 1) v = cr.w; else if (k == 2) v = cr.px; else if (k == 3) v = cr.py;
                else if (k == 4) v = cr.h * cr.w;  // the numerator of area / (H * W)
                else {
                    const unsigned id = KIND == PCBENV_SPATIAL ? t.pid[c * p.mp + k - 5] : 0xFFFFu;
                    v = id == 0xFFFFu ? -1 : (int)id;
                }
            }
            cf[i] = (short)v;
            c += dc; k += dk;
            if (k >= p.F) { k -= p.F; c++; }
        }
    }
    if (p.cbuf.placement_mask) {
        unsigned char *pm = p.cbuf.placement_mask + (size_t)row * p.C;
        for (int c = lane; c < p.C; c += NT) {
            const bool placed = c < nc && l.comps[c].px >= 0;
            pm[c] = (unsigned char)(KIND == PCBENV_RECT ? (placed ? 1 : 0) : (c >= nc ? 0 : placed ? 2 : c == cur ? 3 : 1));
        }
    }
    if (KIND == PCBENV_RECT && p.cbuf.component_mask) {
        unsigned char *cm = p.cbuf.component_mask + (size_t)row * p.C;
        for (int c = lane; c < p.C; c += NT) cm[c] = (unsigned char)(c < nc ? 1 : 0);
    }
    if (KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL) {
        // one 32-bit word per pin row of all_pins_num_feature, one byte / 16-bit word per row of all_pins_cat_feature
        unsigned *fn = (unsigned *)p.cbuf.all_pins_num_feature; if (fn) fn += (size_t)row * p.pinRows;
        signed char *fc = p.cbuf.all_pins_cat_feature ? p.cbuf.all_pins_cat_feature + (size_t)row * p.pinRows * p.catW : 0;
        if (!fn && !fc) return;
        auto quad = [](const PinRec &pr) { return (unsigned)pr.rel_x | ((unsigned)pr.rel_y << 8) | ((unsigned)(unsigned char)pr.abs_x << 16) | ((unsigned)(unsigned char)pr.abs_y << 24); };
        if (KIND == PCBENV_SPATIAL) {  // row = global pin id: rows 0..np-1 belong to pins, the others are constant
            for (int r = np + lane; r < p.pinRows; r += NT) {
                if (fn) fn[r] = 0u;
                if (fc) ((unsigned short *)fc)[r] = (unsigned short)(r == p.pinRows - 1 ? 0xFFFFu : 0u);  // (net, component) of a row as one 16-bit word
            }
            for (int q = lane; q < np; q += NT) {
                const PinRec pr = l.pins[q];
                const int r = pr.id & PIN_ID_MASK;
                if (fn) fn[r] = quad(pr);
                if (fc) ((unsigned short *)fc)[r] = (unsigned short)(pr.net | (pr.comp << 8));
            }
        } else {  // pin env: rows [component, pin_id] through the membership bit map (always large enough for this kind)
            u64 *rowbits = l.hf;
            lds_sync();
            for (int i = lane; i < p.ldsHfWords; i += NT) rowbits[i] = 0ull;
            lds_sync();
            for (int q = lane; q < np; q += NT) {
                const PinRec pr = l.pins[q];
                const int r = pr.comp * p.mp + (pr.id & PIN_ID_MASK);
                atomicOr((unsigned long long *)&rowbits[r >> 6], 1ull << (r & 63));
            }
            lds_sync();
            for (int r = lane; r < p.pinRows; r += NT) {
                if ((rowbits[r >> 6] >> (r & 63)) & 1ull) continue;
                if (fn) fn[r] = 0u;
                if (fc) fc[r] = (signed char)0;
            }
            for (int q = lane; q < np; q += NT) {
                const PinRec pr = l.pins[q];
                if (pr.id & PIN_LOSER) continue;  // quirk Q1: the last pin with this [component, pin_id] owns the row
                const int r = pr.comp * p.mp + (pr.id & PIN_ID_MASK);
                if (fn) fn[r] = quad(pr);
                if (fc) fc[r] = (signed char)pr.net;
            }
            lds_sync();
        }
    }
}

template <int KIND> static __device__ inline void emit_features_full(const DevParams &p, Lds &l, int row, int lane, bool comp_from_cache = false) {
    if (KIND == PCBENV_SQUARE) return;
    emit_features_compact<KIND>(p, l, row, lane, comp_from_cache);
    const int nc = l.hdr->ncomp, np = l.hdr->npins, cur = l.hdr->cur;
    if (p.buf.all_components_feature) {
        const PinTables t = pin_tables(p, l);
        double *cf = p.buf.all_components_feature + (size_t)row * p.C * p.F;
        int c = lane / p.F, k = lane - c * p.F;  // (component, field) advanced without a division per element
        const int dc = NT / p.F, dk = NT - dc * p.F;
        for (int i = lane; i < p.C * p.F; i += NT) {
            double v = 0.0;
            if (c < nc) {
                const CompRec cr = l.comps[c];
                if (k == 0) v = cr.h; else if (k == 1) v = cr.w; else if (k == 2) v = cr.px; else if (k == 3) v = cr.py;
                else if (k == 4) v = (double)(cr.h * cr.w) / p.area;
                else {
                    const unsigned id = KIND == PCBENV_SPATIAL ? t.pid[c * p.mp + k - 5] : 0xFFFFu;
                    v = id == 0xFFFFu ? -1.0 : (double)id;
                }
            }
            cf[i] = v;
            c += dc; k += dk;
            if (k >= p.F) { k -= p.F; c++; }
        }
    }
    if (p.buf.placement_mask) {
        double *pm = p.buf.placement_mask + (size_t)row * p.C;
        for (int c = lane; c < p.C; c += NT) {
            const bool placed = c < nc && l.comps[c].px >= 0;
            pm[c] = KIND == PCBENV_RECT ? (placed ? 1.0 : 0.0) : (c >= nc ? 0.0 : placed ? 2.0 : c == cur ? 3.0 : 1.0);
        }
    }
    if (KIND == PCBENV_RECT && p.buf.component_mask) {
        double *cm = p.buf.component_mask + (size_t)row * p.C;
        for (int c = lane; c < p.C; c += NT) cm[c] = c < nc ? 1.0 : 0.0;
    }
    if (KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL) {
        double *fn = p.buf.all_pins_num_feature ? p.buf.all_pins_num_feature + (size_t)row * p.pinRows * 4 : 0;
        double *fc = p.buf.all_pins_cat_feature ? p.buf.all_pins_cat_feature + (size_t)row * p.pinRows * p.catW : 0;
        if (!fn && !fc) return;
        if (KIND == PCBENV_SPATIAL) {  // row = global pin id: rows 0..np-1 belong to pins, the others are constant
            for (int r = np + lane; r < p.pinRows; r += NT) {
                if (fn) { fn[4 * r] = 0.0; fn[4 * r + 1] = 0.0; fn[4 * r + 2] = 0.0; fn[4 * r + 3] = 0.0; }
                if (fc) { const double v = r == p.pinRows - 1 ? -1.0 : 0.0; fc[2 * r] = v; fc[2 * r + 1] = v; }
            }
            for (int q = lane; q < np; q += NT) {
                const PinRec pr = l.pins[q];
                const int r = pr.id & PIN_ID_MASK;
                if (fn) { fn[4 * r] = pr.rel_x; fn[4 * r + 1] = pr.rel_y; fn[4 * r + 2] = pr.abs_x; fn[4 * r + 3] = pr.abs_y; }
                if (fc) { fc[2 * r] = pr.net; fc[2 * r + 1] = pr.comp; }
            }
        } else if (p.pinRows <= p.ldsHfWords * 64) {  // pin env: rows [component, pin_id] through a membership bit map
            u64 *rowbits = l.hf;
            lds_sync();
            for (int i = lane; i < p.ldsHfWords; i += NT) rowbits[i] = 0ull;
            lds_sync();
            for (int q = lane; q < np; q += NT) {
                const PinRec pr = l.pins[q];
                const int r = pr.comp * p.mp + (pr.id & PIN_ID_MASK);
                atomicOr((unsigned long long *)&rowbits[r >> 6], 1ull << (r & 63));
            }
            lds_sync();
            for (int r = lane; r < p.pinRows; r += NT) {
                if ((rowbits[r >> 6] >> (r & 63)) & 1ull) continue;
                if (fn) { fn[4 * r] = 0.0; fn[4 * r + 1] = 0.0; fn[4 * r + 2] = 0.0; fn[4 * r + 3] = 0.0; }
                if (fc) fc[r] = 0.0;
            }
            for (int q = lane; q < np; q += NT) {
                const PinRec pr = l.pins[q];
                if (pr.id & PIN_LOSER) continue;  // quirk Q1: the last pin with this [component, pin_id] owns the row
                const int r = pr.comp * p.mp + (pr.id & PIN_ID_MASK);
                if (fn) { fn[4 * r] = pr.rel_x; fn[4 * r + 1] = pr.rel_y; fn[4 * r + 2] = pr.abs_x; fn[4 * r + 3] = pr.abs_y; }
                if (fc) fc[r] = pr.net;
            }
            lds_sync();
        } else {  // more rows than the bit map holds (tiny grids with large components): zero everything, then the rows
            for (int i = lane; i < p.pinRows * 4; i += NT) if (fn) fn[i] = 0.0;
            for (int i = lane; i < p.pinRows; i += NT) if (fc) fc[i] = 0.0;
            store_drain_sync();
            for (int q = lane; q < np; q += NT) {
                const PinRec pr = l.pins[q];
                if (pr.id & PIN_LOSER) continue;
                const int r = pr.comp * p.mp + (pr.id & PIN_ID_MASK);
                if (fn) { fn[4 * r] = pr.rel_x; fn[4 * r + 1] = pr.rel_y; fn[4 * r + 2] = pr.abs_x; fn[4 * r + 3] = pr.abs_y; }
                if (fc) fc[r] = pr.net;
            }
        }
    }
}

// Terminal reward (S:793-929 find_reward), all three reward types, inside the step kernel.
// ROUTES = false compiles the beam-search code out (reward_type centroid: what every shipped reference config uses).
// nparts > 1: this team is one of `nparts` (the environment's own wavefront = part 0 and its reward helpers, run_env)
// that each count a share of the segment pairs: the shares meet in term_arrive[pos] -- one returning 64-bit atomic add
// of (1 arrival | share of the first route's count << 8 | share of the second's << 36) -- and the team whose add comes
// last has the totals, writes reward and info, and clears the word for the next launch.  Everything else (routes,
// wirelength: a sequential float64 sum) every team computes for itself, so the result does not depend on who is last.
template <int KIND, bool ROUTES>
static __device__ __forceinline__ void terminal_reward(const DevParams &p, Lds &l, int row, int lane, int part, int nparts, unsigned pos) {
    const bool placed_all = l.hdr->cur < 0;
    double reward, wl, ni;
    if (!placed_all) {  // S:853-863 worst case: the upper bounds, normalised (spatial: twice, quirk Q3)
        reward = -p.w_wl * (p.max_wl / p.wl_norm) - p.w_int * (p.max_int / p.int_norm);
        wl = p.max_wl; ni = p.max_int;
    } else {
        double wsum[2] = {0.0, 0.0}; int cnt[2] = {0, 0};
        if (!ROUTES) route_centroid(p, l.hdr, l.pins, l.seg, lane, part, nparts, &wsum[0], &cnt[0]);
        else route_beam_or_both(p, l.hdr, l.pins, l.seg, lane, part, nparts, wsum, cnt);
        if (nparts > 1) {
            if (lane != 0) return;  // one lane carries the shares; nobody else writes anything below
            const u64 mine = 1ull | ((u64)(unsigned)cnt[0] << 8) | ((u64)(unsigned)cnt[1] << 36);
            const u64 before = atomicAdd((unsigned long long *)(p.term_arrive + pos), (unsigned long long)mine);
            if ((int)(before & 0xFFull) != nparts - 1) return;  // not the last: the totals are somebody else's to write
            const u64 total = before + mine;
            cnt[0] = (int)((total >> 8) & 0xFFFFFFFull); cnt[1] = (int)(total >> 36);
            __hip_atomic_store(p.term_arrive + pos, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // S:609-627 lowest_num_intersections: ties keep the beam route (quirk Q9)
        const int pick = (ROUTES && p.reward_type == PCBENV_REWARD_BOTH && cnt[1] < cnt[0]) ? 1 : 0;
        wl = wsum[pick] / p.wl_norm;
        ni = (double)cnt[pick] / p.int_norm;
        reward = -1 * (p.w_wl * wl + p.w_int * ni);
    }
    if (lane == 0) {
        p.buf.reward[row] = reward;
        if (p.buf.info) { p.buf.info[2 * (size_t)row] = wl; p.buf.info[2 * (size_t)row + 1] = ni; }
    }
}
