// pcb_reset.h -- reset of one environment from the instance queue (k_reset and the in-launch reset of k_step)
// Part of libpcbenv.so's single translation unit (included by pcbenv_kernels.hip); CDNA4 / gfx950 only.

// ----------------------------------------------------------------------------------------------
// reset (R:310-351, P:1544-1597, S:1487-1549, Q:74-113): header is in LDS; builds the new episode's state in
// LDS from the next queued instance and rewrites every observation tensor of environment e.
// ----------------------------------------------------------------------------------------------
// The next queued instance of environment e: header and 8-byte records, all loads issued together.
struct InstRegs { int nc, nn, np; u64 comp; u64 pin[4]; };
static __device__ inline void fetch_instance(const DevParams &p, unsigned qcursor, int e, int lane, InstRegs &ir) {
    const unsigned slot = qcursor % (unsigned)p.Q;
    const unsigned char *rec = p.queue + ((size_t)slot * p.B + e) * p.instStride;
    const int *ih = (const int *)rec;
    const u64 *crec = (const u64 *)(rec + 16), *prec = crec + p.C;  // 8-byte records, one load each
    ir.nc = load_agent(ih); ir.nn = load_agent(ih + 1); ir.np = load_agent(ih + 2);
    ir.comp = lane < p.C ? load_agent(crec + lane) : 0ull;
    #pragma unroll
    for (int r = 0; r < 4; r++) { const int q = lane + r * NT; ir.pin[r] = q < p.P ? load_agent(prec + q) : 0ull; }
}

// `what`: RESET_ALL, or the two halves a delegated reset is split into (run_env): RESET_STATE_MASKS = the new episode's
// state in LDS (for the state block) + grid and action_mask, by the environment's own team; RESET_FEATURES = everything
// else the reset writes (feature tensors, pin_grid, component_grid), by a helper team working from its own copy of the
// old state and the same queued instance.  The two write disjoint tensors.
#define RESET_STATE_MASKS 1
#define RESET_FEATURES 2
#define RESET_ALL 3
template <int KIND, int WW, bool TRAJ> static __device__ inline void reset_env(const DevParams &p, Lds &l, int e, int row, int lane, int what = RESET_ALL) {
    const int H = p.H, W = p.W, HW = H * W;
    const bool owner = what & RESET_STATE_MASKS, feats = what & RESET_FEATURES;
    const bool full = TRAJ && p.num_slots > 1;  // trajectory layout: the destination slot holds nothing of this environment yet
    lds_sync();
    // The float64 pin-feature tensors are maintained row-wise (a step rewrites only the placed component's
    // rows), so a reset clears just the rows the finished episode used -- unless these buffers have not been
    // initialised for this environment yet (first reset after pcbenv_bind_buffers): then a full zero fill.
    // Rows of the finished episode that the new episode rewrites are left alone (no write-after-write on a row, so
    // no ordering wait between the clear and the later row writes): spatial rows are the pin ids 0..np-1; the pin
    // env's rows [component, pin_id] go through a membership bit map in the fold scratch.
    InstRegs ir;
    if (KIND != PCBENV_SQUARE) {
        fetch_instance(p, l.hdr->qcursor, e, lane, ir);
        if (owner && p.gen_produced && lane == 0 && load_agent(p.gen_produced + e) <= l.hdr->qcursor) atomicOr(p.gen_errors, 1u);
    }
    bool rows_cleared = false;
    if (feats && (KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL) && !full && l.hdr->feat_gen == p.bind_gen &&
        (KIND == PCBENV_SPATIAL || p.C * p.mp <= p.ldsHfWords * 64)) {
        u64 *rowbits = l.hf;
        if (KIND == PCBENV_PIN) {
            for (int i = lane; i < p.ldsHfWords; i += NT) rowbits[i] = 0ull;
            lds_sync();
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                const int q = lane + r * NT;
                if (q < ir.np && q < p.P) {
                    const u64 w = ir.pin[r];
                    const int prow = (int)((w >> 24) & 0xFF) * p.mp + (int)((w >> 32) & PIN_ID_MASK);
                    atomicOr((unsigned long long *)&rowbits[prow >> 6], 1ull << (prow & 63));
                }
            }
            lds_sync();
        }
        for (int q = lane; q < l.hdr->npins; q += NT) {
            const PinRec pr = l.pins[q];
            const int prow = KIND == PCBENV_SPATIAL ? (pr.id & PIN_ID_MASK) : pr.comp * p.mp + (pr.id & PIN_ID_MASK);
            if (KIND == PCBENV_SPATIAL ? prow < ir.np : (int)((rowbits[prow >> 6] >> (prow & 63)) & 1ull)) continue;
            if (p.buf.all_pins_num_feature) {
                double *f = p.buf.all_pins_num_feature + ((size_t)row * p.pinRows + prow) * 4;
                f[0] = 0.0; f[1] = 0.0; f[2] = 0.0; f[3] = 0.0;
            }
            if (p.buf.all_pins_cat_feature) {
                double *f = p.buf.all_pins_cat_feature + ((size_t)row * p.pinRows + prow) * p.catW;
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
            cr.o = 0; cr.pad[0] = cr.pad[1] = cr.pad[2] = 0;
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
        // The advanced cursor tells the generator that the record's slot may be overwritten: published only now that every
        // wavefront of the team has its part of the record (the loads were waited for before the LDS writes above).
        if (owner && lane == 0) store_agent(p.cursor_pub + e, l.hdr->qcursor);
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
    if (owner) mask_and_emit<KIND, WW>(p, l, row, lane, true, 0, H);
    STAMP(17);

    if (KIND != PCBENV_SQUARE) {
        const int nc = l.hdr->ncomp, np = l.hdr->npins;
        if (KIND == PCBENV_SPATIAL) {
            // rank of every pin among the pins of its component (self.pins order), kept in the state block
            if (np <= WAVE) {  // one pin per lane of wavefront 0: one ballot per component instead of np broadcast reads per pin
                if (lane < WAVE) {
                    const int comp = lane < np ? (int)l.pins[lane].comp : -1;
                    int rank = 0;
                    for (int c = 0; c < nc; c++) {
                        const u64 m = __ballot(comp == c);
                        if (comp == c) rank = __popcll(m & ((1ull << lane) - 1ull));
                    }
                    if (lane < np) l.rank[lane] = (unsigned char)rank;
                }
            } else {
                for (int q = lane; q < np; q += NT) {
                    const int comp = l.pins[q].comp;
                    int rank = 0;
                    #pragma unroll 4
                    for (int q2 = 0; q2 < np; q2++) rank += (q2 < q) & (l.pins[q2].comp == comp);  // broadcast reads
                    l.rank[q] = (unsigned char)rank;
                }
            }
            if (feats) build_pin_tables(p, l, lane);  // pid / netmask scratch in the class-map zone (free until the next emit_pin_grid)
        }
        if ((KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL) && lane == 0) l.hdr->feat_gen = p.bind_gen;  // (state: whoever fills the rows, they are filled)
        if (!feats) {
            // the feature tensors, pin_grid and component_grid are the helper's
        } else if (full) {
            emit_features_full<KIND>(p, l, row, lane);
        } else {
            const PinTables t = pin_tables(p, l);
            // all_components_feature (R:60-79, S:203-239): [h, w, -1, -1, area/(H*W), (spatial: pin ids, -1 pad)]; absent rows 0
            if (p.buf.all_components_feature) {
                double *cf = p.buf.all_components_feature + (size_t)row * p.C * p.F;
                // one (component, field) element per lane and trip, the pair advanced without a division per element
                int c = lane / p.F, k = lane - c * p.F;
                const int dc = NT / p.F, dk = NT - dc * p.F;
                for (int i = lane; i < p.C * p.F; i += NT) {
                    double v = 0.0;
                    if (c < nc) {
                        const CompRec cr = l.comps[c];
                        if (k == 0) v = cr.h; else if (k == 1) v = cr.w; else if (k == 2 || k == 3) v = -1.0;
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
            STAMP(18);
            if (p.buf.placement_mask) {
                double *pm = p.buf.placement_mask + (size_t)row * p.C;
                for (int c = lane; c < p.C; c += NT)
                    pm[c] = KIND == PCBENV_RECT ? 0.0 : (c == 0 ? 3.0 : (c < nc ? 1.0 : 0.0));
            }
            if (KIND == PCBENV_RECT && p.buf.component_mask) {
                double *cm = p.buf.component_mask + (size_t)row * p.C;
                for (int c = lane; c < p.C; c += NT) cm[c] = c < nc ? 1.0 : 0.0;
            }
            if (KIND == PCBENV_PIN || KIND == PCBENV_SPATIAL) {
                if (!rows_cleared && p.buf.all_pins_num_feature) {
                    double *f = p.buf.all_pins_num_feature + (size_t)row * p.pinRows * 4;
                    for (int i = lane; i < p.pinRows * 4; i += NT) f[i] = 0.0;
                }
                if (!rows_cleared && p.buf.all_pins_cat_feature) {
                    double *f = p.buf.all_pins_cat_feature + (size_t)row * p.pinRows * p.catW;
                    for (int i = lane; i < p.pinRows * p.catW; i += NT)
                        f[i] = (KIND == PCBENV_SPATIAL && i >= (p.pinRows - 1) * p.catW) ? -1.0 : 0.0;  // S:1520
                }
                if (!rows_cleared) {  // first reset after a bind: the full zero fill above must land before the row writes
                    store_drain_sync();
                    __threadfence_block();
                }
                for (int q = lane; q < np; q += NT) {
                    const PinRec pr = l.pins[q];
                    write_pin_num<KIND>(p, row, pr);
                    if (p.buf.all_pins_cat_feature) {
                        if (KIND == PCBENV_SPATIAL) {
                            double *f = p.buf.all_pins_cat_feature + ((size_t)row * p.pinRows + (pr.id & PIN_ID_MASK)) * 2;
                            f[0] = pr.net; f[1] = pr.comp;
                        } else if (!(pr.id & PIN_LOSER)) {
                            p.buf.all_pins_cat_feature[(size_t)row * p.pinRows + pr.comp * p.mp + (pr.id & PIN_ID_MASK)] = pr.net;
                        }
                    }
                }
            }
        }
        STAMP(19);
        if (KIND == PCBENV_SPATIAL && feats) {
            if (p.buf.pin_grid) emit_zero(p.buf.pin_grid + (size_t)row * HW * p.K, (long long)HW * p.K, lane, p.stream_stores);  // S:1504
            emit_component_grid(p, l, row, lane);  // S:1677-1697 draw_components (unrotated rel coords; channel 0 == 1)
            if (full) feat_cache_fill(p, l, e, lane);  // what the steps of this episode copy into their slots
        }
    }
    lds_sync();
}


