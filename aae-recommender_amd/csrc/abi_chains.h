// The hidden stacks as layer-chain programs (chain.h / chain4.h), the grouped weight-gradient launches, tile buckets and the
// first layer's per-item update, next-batch prefetch, VAE / discriminator / generator programs.
// (one of the parts of aae_abi.hip's translation unit: included there in order, not on its own)
#pragma once

namespace {

// ==========================================================================================
// chain path (chain.h): the hidden stacks as row-blocked programs
// ==========================================================================================
ChainOp cop(int kind, int src, int dst, int N) {
    ChainOp o; memset(&o, 0, sizeof(o));
    o.kind = kind; o.src = src; o.dst = dst; o.N = N; o.one_col = -1; o.scale = 1.f; o.yslot = 0; o.fake_slot = -1;
    return o;
}
ChainOp cop_load(const float* g, int ld, int dst, int N, int row0 = 0) {
    ChainOp o = cop(COP_LOAD, 0, dst, N); o.W = g; o.ldw = ld; o.out_row0 = row0; return o;
}
ChainOp cop_linear(int kind, int src, int dst, const Ten& W, int K, int N, int epi) {
    ChainOp o = cop(kind, src, dst, N); o.W = W.p; o.ldw = (int)W.ld; o.K = K; o.epi = epi;
    if (kind == COP_LINEAR_DX) { o.Wkn = W.p; o.ldkn = (int)W.ld; }      // already k-major
    return o;
}
void cop_out(ChainOp& o, float* out, int ld, int row0 = 0) { o.out = out; o.ldo = ld; o.out_row0 = row0; }

W4Copies w4_of(const aae_model* m, int pid) {
    return W4Copies{m->PT[pid].p, m->D4[pid].p, (int)m->P[pid].rows, (int)m->P[pid].cols, m->FXi[pid], m->DXi[pid], m->FXBi[pid], kCWide};
}
// where dL/d(a1) of the encoder backward goes: gb3, or (external first layer, export mode) the tail of ga1x
float* ga1_ptr(const aae_model* m) {
    return (m->ext_first && m->ga1x.p) ? m->ga1x.p + (size_t)(m->R - m->rows) * m->ldh : m->gb3.p;
}
// (re-)derive the k4-interleaved copies of a hidden layer after something other than the optimiser kernels wrote the weights
void ensure_pt(aae_model* m, int pid, hipStream_t s) {
    const Ten& T = m->PT[pid];
    if (!T.p || m->pt_ok[pid]) return;
    const Ten& W = m->P[pid];
    hipLaunchKernelGGL(interleave4_kernel, dim3(grid1d((size_t)W.rows * ((W.cols + 3) / 4))), dim3(256), 0, s, W.p, (int)W.ld,
                       w4_of(m, pid));
    m->pt_ok[pid] = true;
}

// forward layer: dst[rows][N] = epi(src[rows][K] * W[N][K]^T), K = in + 1 (the bias input is the last column)
ChainOp cop_fwd(aae_model* m, int pid, int src, int dst, int K, int N, int epi, hipStream_t s) {
    ChainOp o = cop_linear(COP_LINEAR, src, dst, m->P[pid], K, N, epi);
    if (m->PT[pid].p) { ensure_pt(m, pid, s); o.W4 = m->PT[pid].p; o.ns4 = (int)m->P[pid].rows; }
    if (m->FXi[pid]) { o.WX = m->FXi[pid]; o.xpl = ((int)m->P[pid].rows + 15) & ~15; }
    return o;
}

// dX of a hidden layer: dst[rows][N] = epi(src[rows][K] * W[K][0:N]).  chain4.h reads the D4 copy (k = output row) with
// 16-byte loads, or the matrix itself (k-major for this product); chain.h walks the matrix.
ChainOp cop_dx(aae_model* m, int pid, int src, int dst, int K, int N, int epi, hipStream_t s) {
    ChainOp o = cop_linear(COP_LINEAR_DX, src, dst, m->P[pid], K, N, epi);       // (Wkn = the matrix itself: k-major for this product)
    if (m->D4[pid].p) { ensure_pt(m, pid, s); o.W4 = m->D4[pid].p; o.ns4 = (int)m->P[pid].cols; }
    if (m->DXi[pid]) { o.WX = m->DXi[pid]; o.xpl = ((int)m->P[pid].cols + 15) & ~15; }
    return o;
}

struct ChainBuilder;
// The decoder's first layer reads [z | condition | 1] - wider than a slot row when the condition is (C4: 50 + 300 + 1).
// Such a layer runs as two ops (4-row kernel): input columns [0, 208) from one slot, the rest from a second one, the second op
// adding to the first one's products before the epilogue (ChainOp::acc_in); its dX as two ops over the input columns.
inline bool wide_dec_in(const aae_model* m) { return m->cp + 1 > kCWide; }

struct ChainBuilder {
    ChainProgram P;
    ChainBuilder(const aae_model* m, int rows) {
        memset(&P, 0, sizeof(P));
        P.rows = rows; P.act = m->cfg.activation; P.seed = m->cfg.seed; P.step_ctr = m->step_ctr;
        P.loss_out = m->losses; P.loss_slot = 3;
        P.dbg = m->opt.chain_skip; P.kslices = (m->opt.chain_kslices || m->rows > 16 * kMB || rows > 32 * kMB) ? 1 : 0;     // (chain4.h: the column-owner form for batches of one fused launch)
    }
    ChainOp& add(const ChainOp& o) { P.ops[P.nops] = o; return P.ops[P.nops++]; }
    int x16_rows = 0;      // > 0: this program's own row threshold for the wide-batch kernel (beside_deferred)
    // The discriminator and generator programs of a row-blocked step over a LARGE vocabulary run beside its deferred launch,
    // which holds 3/4 of the chip for most of the step's tail: a 4-row launch of 128 workgroups (512 rows) then takes two
    // rounds on the CUs that are left - there the 16-row kernel pays from 512 rows on (C3 at batch 512: generator program 64 ->
    // 46 us).  The ae programs run in front of / right behind the critical launch, on a free chip: they keep the model's rule.
    void beside_deferred(const aae_model* m) {
        const int ntiles = (m->N + kTI - 1) / kTI;
        if (m->rows > 16 * kMB && ntiles >= 1024 && m->x16_rows > 512) x16_rows = 512;
    }
};

// y of an ACTBWD epilogue that a kernel before this one left in global memory: the 4-row kernel reads it from there (the cells
// of a thread requested before the op's products: no op, no barrier, no slot); the 16-row kernel gets it loaded into `slot`
void set_y(ChainBuilder& cb, aae_model* m, ChainOp& op, const float* y, int ld, int slot) {
    op.yslot = slot;
    if (m->use_chain4) { op.y_glb = y; op.y_ld = ld; }
}
bool y_needs_load(const aae_model* m) { return !m->use_chain4; }

// forward of dec.lin1: srcA holds input columns [0, 208) (all of them when the input fits), srcB the rest with the constant 1
// last; returns the op that carries the epilogue (dropout, activation, stores)
ChainOp& add_dec_in_fwd(ChainBuilder& cb, aae_model* m, int srcA, int srcB, int dst, hipStream_t s) {
    const int h = m->h, cp = m->cp;
    if (!wide_dec_in(m)) return cb.add(cop_fwd(m, P_V1, srcA, dst, cp + 1, h, CEPI_DROPACT, s));
    cb.add(cop_fwd(m, P_V1, srcA, dst, kCWide, h, CEPI_NONE, s));
    ChainOp b = cop_fwd(m, P_V1, srcB, dst, cp + 1 - kCWide, h, CEPI_DROPACT, s);
    b.W += kCWide;
    if (b.W4) b.W4 += (size_t)(kCWide / 4) * b.ns4 * 4;          // (F4: chunk stride = outputs x 4 floats)
    b.WX = m->FXBi[P_V1];                                         // (the split copy counted from input column kCWide, or NULL)
    b.acc_in = 1;
    return cb.add(b);
}
// dX of dec.lin1 -> gzc (+ dzc_out): the columns beyond 208 into slot dstB first, then columns [0, 208) (dL/dz among them)
// into dstA; returns that last op
ChainOp& add_dec_in_dx(ChainBuilder& cb, aae_model* m, int src, int dstA, int dstB, float* dzc_out, hipStream_t s) {
    const int h = m->h, cp = m->cp;
    if (wide_dec_in(m)) {
        ChainOp b = cop_dx(m, P_V1, src, dstB, h, cp - kCWide, CEPI_NONE, s);
        b.W += kCWide; b.Wkn += kCWide;
        if (b.W4) b.W4 += (size_t)kCWide * 4;                     // (D4: [k chunk][input column][4])
        if (b.WX) b.WX += (size_t)kCWide * 32;                    // (DX: [k-step][term][input column][32])
        cop_out(b, m->gzc.p + kCWide, m->ldc);
        if (dzc_out) { b.out2 = dzc_out + kCWide; b.ldo2 = cp; }
        cb.add(b);
    }
    ChainOp& a = cb.add(cop_dx(m, P_V1, src, dstA, h, std::min(cp, kCWide), CEPI_NONE, s));
    cop_out(a, m->gzc.p, m->ldc);
    if (dzc_out) { a.out2 = dzc_out; a.ldo2 = cp; }
    return a;
}

// chain16x3.h keeps 7 slots: the program's slot numbers (up to 10, one per value for readability) are renamed by live range.
// A value = one full write of a slot and the reads up to the next full write; it holds a physical slot from its write to its
// last read.  false: more than 7 values live at once (the caller stays on the 4-row kernel).
static bool x16_remap_slots(ChainProgram& P) {
    constexpr int kL = 16, kMaxVal = 2 * kCMaxOps + kL;
    struct Use { int src, y, fake, dst; };          // the values an op's slot fields name (-1: field not used)
    Use use[kCMaxOps];
    int cur[kL], last[kMaxVal], phys[kMaxVal], nval = 0;
    for (int i = 0; i < kL; ++i) cur[i] = -1;
    bool bad = false;
    auto read_of = [&](int logical, int at) {      // the value a slot holds when op `at` reads it
        if (logical < 0 || logical >= kL) { bad = true; return 0; }
        if (cur[logical] < 0) { cur[logical] = nval; phys[nval] = -1; ++nval; }      // (read before any write: a value of its own)
        last[cur[logical]] = at;
        return cur[logical];
    };
    for (int i = 0; i < P.nops && !bad; ++i) {
        const ChainOp& o = P.ops[i];
        Use u = {-1, -1, -1, -1};
        bool modify = false;                       // the op reads (part of) what dst holds: no new value
        switch (o.kind) {
            case COP_LINEAR: case COP_LINEAR_DX:
                u.src = read_of(o.src, i);
                if (o.epi == CEPI_ACTBWD && !o.y_glb) u.y = read_of(o.yslot, i);
                modify = o.acc_in != 0;
                break;
            case COP_LOAD: modify = o.dst_col0 > 0; break;
            case COP_DROPACT: u.src = read_of(o.src, i); break;
            case COP_STORE: case COP_FINAL_FWD: modify = true; break;
            case COP_FINAL_BWD: case COP_DISC_HEAD: u.src = read_of(o.src, i); u.y = read_of(o.yslot, i); break;
            case COP_PRIOR: if (o.fake_slot >= 0) u.fake = read_of(o.fake_slot, i); break;
            default: break;
        }
        if (modify) u.dst = read_of(o.dst, i);
        else {
            if (o.dst < 0 || o.dst >= kL || nval >= kMaxVal - 1) return false;
            cur[o.dst] = nval; last[nval] = i; phys[nval] = -1; u.dst = nval++;
        }
        if (nval >= kMaxVal - 1) return false;
        use[i] = u;
    }
    if (bad) return false;
    int owner[kX16Slots];                            // physical slot -> the value that holds it
    for (int p = 0; p < kX16Slots; ++p) owner[p] = -1;
    auto place = [&](int v, int at) {
        if (v < 0 || phys[v] >= 0) return true;
        for (int p = 0; p < kX16Slots; ++p)
            if (owner[p] < 0 || last[owner[p]] < at) { owner[p] = v; phys[v] = p; return true; }
        return false;
    };
    for (int i = 0; i < P.nops; ++i) {
        const Use& u = use[i];
        if (!place(u.src, i) || !place(u.y, i) || !place(u.fake, i) || !place(u.dst, i)) return false;
    }
    for (int i = 0; i < P.nops; ++i) {
        ChainOp& o = P.ops[i];
        const Use& u = use[i];
        if (u.src >= 0) o.src = phys[u.src];
        if (u.y >= 0) o.yslot = phys[u.y];
        if (u.fake >= 0) o.fake_slot = phys[u.fake];
        o.dst = phys[u.dst];
        if (o.kind == COP_FINAL_FWD) o.src = o.dst;
    }
    return true;
}

static bool x16_program_ok(const ChainProgram& P) {
    for (int i = 0; i < P.nops; ++i) {
        const ChainOp& o = P.ops[i];
        switch (o.kind) {
            case COP_LINEAR: case COP_LINEAR_DX:
                if (!o.WX || o.N > kCWide || o.K > kX16Kp) return false;
                break;
            case COP_LOAD: if (o.dst_col0 + o.N > kX16Kp) return false; break;
            case COP_DROPACT: case COP_SLABSUM: case COP_STORE: case COP_FINAL_FWD: case COP_FINAL_BWD: case COP_PRIOR: case COP_DISC_HEAD:
                if (o.N > kCWide) return false;
                break;
            default: return false;                 // (COP_ADV, COP_ACTBWD, COP_REPARAM*: the VAE's programs and the 16-row fp32 kernel's)
        }
        if (o.one_col >= kX16Kp) return false;
    }
    return true;
}

int launch_chain(aae_model* m, ChainBuilder& cb, hipStream_t s) {
    if (cb.P.nops > kCMaxOps) return fail(AAE_ESTATE, "chain program too long");
    const int grid = (cb.P.rows + kCR - 1) / kCR + (cb.P.bk.enabled ? 1 : 0);
    const bool want_ts = m->opt.chain_ts;      // debug: per-op timeline of workgroup 0
    static unsigned long long* ts_dev = nullptr;
    if (want_ts) {
        if (!ts_dev && hipMalloc(&ts_dev, 128 * sizeof(unsigned long long)) != hipSuccess) return fail(AAE_EHIP, "ts alloc");
        cb.P.ts = ts_dev;
    }
    ProfScope ps(m, AAE_K_CHAIN, s);
    // 4-row workgroups (chain4.h) whenever every linear op of the program has its k-major matrix (all but the VAE's)
    bool four = m->use_chain4;
    for (int i = 0; i < cb.P.nops && four; ++i)
        if ((cb.P.ops[i].kind == COP_LINEAR || cb.P.ops[i].kind == COP_LINEAR_DX) && !cb.P.ops[i].Wkn && !cb.P.ops[i].W4) four = false;
    for (int i = 0; i < cb.P.nops && four; ++i)
        if (cb.P.ops[i].kind == COP_ADV || cb.P.ops[i].kind == COP_REPARAM || cb.P.ops[i].kind == COP_REPARAM_BWD) four = false;
    for (int i = 0; i < cb.P.nops; ++i)
        if ((cb.P.ops[i].row_lo > 0 || cb.P.ops[i].acc_in || cb.P.ops[i].y_glb) && !four)
            return fail(AAE_ESTATE, "a program prefix for the upper rows / a layer in two k-parts needs the 4-row chain kernel");
    // wide batches: 16 rows per workgroup on the bf16 matrix cores (chain16x3.h)
    if (four && m->x16_ok && cb.P.rows >= (cb.x16_rows > 0 ? cb.x16_rows : m->x16_rows) && !cb.P.bk.enabled && x16_program_ok(cb.P)) {
        ChainProgram X = cb.P;
        if (x16_remap_slots(X)) {
            // the look-ahead chains of the weight prefetch: per workgroup class, every op's next linear op
            X.x16_row_lo = 1 << 30;
            for (int i = 0; i < X.nops; ++i) if (X.ops[i].row_lo > 0) X.x16_row_lo = X.ops[i].row_lo;
            for (int cls = 0; cls < 2; ++cls) {
                int nxt = -1;
                for (int i = X.nops - 1; i >= 0; --i) {
                    X.ops[i].x16_next_lin[cls] = nxt;
                    const bool runs = cls == 1 || X.ops[i].row_lo == 0;
                    if (runs && (X.ops[i].kind == COP_LINEAR || X.ops[i].kind == COP_LINEAR_DX)) nxt = i;
                }
                X.x16_first_lin[cls] = nxt;
            }
            const int grid16 = (X.rows + kX16R - 1) / kX16R;
            if (want_ts && !m->bf16) hipLaunchKernelGGL((chain16x3_kernel<false, true>), dim3(grid16), dim3(kX16T), kX16Lds, s, X);
            else if (m->bf16) hipLaunchKernelGGL(chain16x3_kernel<true>, dim3(grid16), dim3(kX16T), kX16Lds, s, X);
            else hipLaunchKernelGGL(chain16x3_kernel<false>, dim3(grid16), dim3(kX16T), kX16Lds, s, X);
            LAUNCHCHK("chain16x3_kernel");
            if (want_ts && !m->bf16) {
                unsigned long long h[128];
                HIPCHK(hipStreamSynchronize(s));
                HIPCHK(hipMemcpy(h, ts_dev, sizeof(h), hipMemcpyDeviceToHost));
                static const char* names[] = {"LOAD", "LINEAR", "LINEAR_DX", "FINAL_FWD", "FINAL_BWD", "ADV", "DROPACT", "SLABSUM", "ACTBWD", "STORE", "REPARAM", "REPARAM_BWD", "DISC_HEAD", "PRIOR"};
                fprintf(stderr, "[chain16x3 rows=%d nops=%d total=%.2fus; per op: whole (products | epilogue | barrier)]", X.rows, X.nops, (h[3 * X.nops] - h[0]) * 0.01);
                for (int i = 0; i < X.nops; ++i) {
                    const bool lin = X.ops[i].kind == COP_LINEAR || X.ops[i].kind == COP_LINEAR_DX;
                    fprintf(stderr, " %s(K%d,N%d)=%.2f", names[X.ops[i].kind], X.ops[i].K, X.ops[i].N, (h[3 * i + 3] - h[3 * i]) * 0.01);
                    if (lin) fprintf(stderr, "(%.2f|%.2f|%.2f)", (h[3 * i + 1] - h[3 * i]) * 0.01, (h[3 * i + 2] - h[3 * i + 1]) * 0.01, (h[3 * i + 3] - h[3 * i + 2]) * 0.01);
                }
                fprintf(stderr, "\n");
            }
            return AAE_OK;
        }
    }
    if (four) {
        const int grid4 = (cb.P.rows + kR4 - 1) / kR4 + (cb.P.bk.enabled ? 1 : 0);
        if (want_ts && !m->bf16) hipLaunchKernelGGL((chain4_kernel<false, true>), dim3(grid4), dim3(kC4T), kCSlots * kCR * kCL * sizeof(float), s, cb.P);
        else if (m->bf16 && !cb.P.kslices) hipLaunchKernelGGL((chain4_kernel<true, false, true>), dim3(grid4), dim3(kC4T), kCSlots * kCR * kCL * sizeof(float), s, cb.P);
        else if (m->bf16) hipLaunchKernelGGL(chain4_kernel<true>, dim3(grid4), dim3(kC4T), kCSlots * kCR * kCL * sizeof(float), s, cb.P);
        else hipLaunchKernelGGL(chain4_kernel<false>, dim3(grid4), dim3(kC4T), kCSlots * kCR * kCL * sizeof(float), s, cb.P);
    } else if (m->act_nm) {
        if (m->bf16) hipLaunchKernelGGL((chain_kernel<true, true>), dim3(grid), dim3(kCT), kCSlots * kCR * kCL * sizeof(float), s, cb.P);
        else hipLaunchKernelGGL((chain_kernel<false, true>), dim3(grid), dim3(kCT), kCSlots * kCR * kCL * sizeof(float), s, cb.P);
    } else if (m->bf16) hipLaunchKernelGGL(chain_kernel<true>, dim3(grid), dim3(kCT), kCSlots * kCR * kCL * sizeof(float), s, cb.P);
    else hipLaunchKernelGGL(chain_kernel<false>, dim3(grid), dim3(kCT), kCSlots * kCR * kCL * sizeof(float), s, cb.P);
    LAUNCHCHK("chain_kernel");
    if (want_ts) {
        unsigned long long h[128];
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipMemcpy(h, ts_dev, sizeof(h), hipMemcpyDeviceToHost));
        static const char* names[] = {"LOAD", "LINEAR", "LINEAR_DX", "FINAL_FWD", "FINAL_BWD", "ADV", "DROPACT", "SLABSUM", "ACTBWD", "STORE", "REPARAM", "REPARAM_BWD", "DISC_HEAD", "PRIOR"};
        fprintf(stderr, "[chain rows=%d nops=%d total=%.2fus]", cb.P.rows, cb.P.nops, (h[cb.P.nops] - h[0]) * 0.01);
        for (int i = 0; i < cb.P.nops; ++i)
            fprintf(stderr, " %s(K%d,N%d%s%s)=%.2f", names[cb.P.ops[i].kind], cb.P.ops[i].K, cb.P.ops[i].N,
                    cb.P.ops[i].out ? ",st" : "", cb.P.ops[i].out2 ? ",st2" : "", (h[i + 1] - h[i]) * 0.01);
        fprintf(stderr, "\n");
        if (cb.P.nops > 2)
            fprintf(stderr, "   [op 2, wave 0 of workgroup 0] loads+mfma+partials=%.2f wait-barrier=%.2f epi-ctx=%.2f epilogue=%.2f barrier=%.2f (us)\n",
                    (h[21] - h[20]) * 0.01, (h[22] - h[21]) * 0.01, (h[23] - h[22]) * 0.01, (h[24] - h[23]) * 0.01, (h[25] - h[24]) * 0.01);
        if (cb.P.nops > 2 && four)
            fprintf(stderr, "   [op 2: matrix phase %.2f us = %.0f shader clocks -> %.2f GHz]\n", (h[64 + 50] - h[64 + 48]) * 0.01,
                    (double)(h[64 + 51] - h[64 + 49]), (double)(h[64 + 51] - h[64 + 49]) / ((h[64 + 50] - h[64 + 48]) * 10.0));
        if (cb.P.nops > 2 && four)
            for (int k = 0; k < 3; ++k) {
                fprintf(stderr, "   [op 2, every wave, us after the op's start: %s]", k == 0 ? "loads issued" : k == 1 ? "first chunk multiplied" : "partial sums stored");
                for (int w = 0; w < 16; ++w) fprintf(stderr, " %.2f", ((double)h[64 + 16 * k + w] - (double)h[64 + 48]) * 0.01);
                fprintf(stderr, "\n");
            }
    }
    return AAE_OK;
}

// up to 4 weight-gradient jobs in one launch
int launch_w1_hot(aae_model* m, hipStream_t s);
struct DwBuilder {
    aae_model* mh = nullptr;       // set by add_first_layer: the wave form's hot list is worked off behind the launch
    DwGroup g; int tiles;
    int ksplit_rows = 0;           // set by add(): the model's threshold for the k-split form of a tile
    int dw_ts = -1;                // ... its DW_TS diagnostic (aae_options)
    DwBuilder() { memset(&g, 0, sizeof(g)); tiles = 0; }
    void add(aae_model* m, const float* G, int ldg, const float* X, int ldx, int rows, int pid, int which) {
        DwJob& J = g.jobs[g.njobs++];
        ksplit_rows = m->dw_ksplit_rows; dw_ts = m->opt.dw_ts;
        const Ten& W = m->P[pid];
        const int set = (which == O_GEN) ? 1 : 0;
        J.G = G; J.ldg = ldg; J.X = X; J.ldx = ldx; J.rows = rows; J.M = (int)W.rows; J.N = (int)W.cols;
        J.p = W.p; J.m = m->M[set][pid].p; J.v = m->V[set][pid].p; J.ld = (int)W.ld; J.sc = m->sc + which;
        J.grad = m->cfg.grad_mode == AAE_GRAD_EXPORT ? m->Gr[pid].p : nullptr;
        J.w4 = w4_of(m, pid);                                  // (fused optimiser: the k4-interleaved copies follow p)
        J.tile0 = tiles; J.tiles_n = (J.N + 31) / 32;
        tiles += ((J.M + 31) / 32) * J.tiles_n;
    }
    // the first encoder layer's bias gradient + update of optimiser `which` ride along (the row-sparse weight gradient +
    // optimiser follow as a launch of their own: encoder_first_layer_update(..., merged = true))
    int add_first_layer(aae_model* m, const float* ga1, int which, hipStream_t s) {
        const int set = (which == O_GEN) ? 1 : 0;
        W1Job& w = g.w1;
        w.enabled = 1; w.ga1 = ga1; w.ld = m->ldh; w.h = m->h; w.rows = m->rows;
        w.bp = m->P[P_B1].p; w.bm = m->M[set][P_B1].p; w.bv1 = m->V[set][P_B1].p;
        w.bgrad = m->cfg.grad_mode == AAE_GRAD_EXPORT ? m->Gr[P_B1].p : nullptr; w.sc = m->sc + which;
        w.ncol = (m->h + 63) / 64;
        // the row-sparse weight gradient + optimiser of the layer rides along while its row lists fit the kernel's static LDS
        // (batches up to ~2 500 rows); not for the dense noisy input (a dense product follows) or an external
        // first layer (the rows live with their item slices) - unless this handle IS the owner of its slice's rows
        // (own_first: aae_shard_step)
        constexpr bool no_merge = false;
        w.nitem = 0;
        m->w1_items_merged = false;
        w.wave_form = 0;
        if (!no_merge && !m->dense_step && (!m->ext_first || m->own_first) && sizeof(int) * w1_items_lds_words(m->rows) <= kDwSmemBytes) {
            TRY(ensure_buckets(m, s));
            w.items = w1_items_args(m, ga1, 0, 0, which);
            w.nitem = std::max(1, std::min(std::min(m->cfg.max_nnz, std::max(256, m->rows * 32)), m->N));      // (<= distinct items possible)
            m->w1_items_merged = true;
            // wide batches (beyond one fused launch) over a large vocabulary: one wave per item, the head items through the hot
            // list behind the launch.  With few items per row of the batch (N < 40 rows) most items sit in many rows and the
            // workgroup form stays ahead (tools/debug/c4_w1_ab.sh, ms/step wave | workgroup form: 1000 rows x 4.6 k items
            // 0.459 | 0.413, 10 k 0.501 | 0.465, 20 k 0.575 | 0.553, 40 k 0.723 | 0.728; 512 rows x 4.6 k 0.304 | 0.282,
            // 10 k 0.329 | 0.314, 20 k 0.362 | 0.363).
            constexpr bool no_wave = false;
            // batches of one fused launch: four items per workgroup (w1_item_hybrid_body: same sums, a quarter of the workgroups)
            constexpr bool no_hybrid = false, hyb_any = false;
            // (ms/step, four items per workgroup | one: C3 0.2528 | 0.2585, C2's shape 0.1685 | 0.1728, C1's - 1 k items, most of them
            //  in many rows - 0.1438 | 0.1419: as for the wide batches' wave form, only with >= 40 items per row of the batch)
            if (!no_hybrid && m->rows <= 16 * kMB && sizeof(int) * w1_hybrid_lds_words(m->rows) <= kDwSmemBytes &&
                (int64_t)m->cfg.max_nnz <= 512ll * w.nitem && ((int64_t)m->N >= 40ll * m->rows || hyb_any)) {
                w.wave_form = 2;
                w.nitem = (w.nitem + 3) / 4;
            } else
            if (!no_wave && m->rows > 16 * kMB && m->hot_list && (int64_t)m->N >= 40ll * m->rows) {
                w.wave_form = 1;
                w.nitem = (w.nitem + 3) / 4;
                const int set = m->hot_flip; m->hot_flip ^= 1;
                w.hot = m->hot_list + (size_t)set * m->hot_cap; w.hot_count = m->hot_count + set; w.hot_zero = m->hot_count + (set ^ 1);
                m->w1_hot_pending = true; m->w1_hot_set = set; m->w1_hot_which = which; m->w1_hot_ga1 = ga1;
                mh = m;
            }
            // r5: the wide batches' forms walk the item list with a stride, so the launch needs no workgroup per POSSIBLE item - an
            // item slice of 12 500 items had 12 500 of them for the ~800 items of a batch, and the ~11 700 that find nothing to do
            // still have to be dealt out: 32 us per launch beside the deferred launch where the busy ones need 13 (per-workgroup
            // clocks, AAE_DW_TS).  How many a batch needs only the device knows: a workgroup of every such launch leaves the
            // step's count in host memory, and the launch is sized by the last count seen there + a quarter (a batch with more
            // items walks the rest with the stride; fewer workgroups than items cost C4, whose items are nearly all busy,
            // 0.302 -> 0.313 ms/step at 1 024).  One rank's step at world 8: 0.275 -> 0.261 ms.
            w.cnt_out = nullptr;
            // (batches of one fused launch too: C3 0.2363-0.2374 -> 0.2336-0.2361 ms/step with ~560 workgroups of four busy waves instead of 800)
            constexpr bool hyb_count = true;
            if ((w.wave_form != 2 || hyb_count) && m->cnt_host) {
                w.cnt_out = m->cnt_host_dev;
                const int seen = *reinterpret_cast<volatile int*>(m->cnt_host);
                if (seen > 0) {
                    const int per = w.wave_form ? 4 : 1;                        // items per workgroup and pass
                    int want = ((seen + seen / 4 + per - 1) / per + 63) & ~63;
                    // (hybrid form: its bitmap of deferred items holds 512 rounds of the launch's workgroups - a count left by a much
                    //  smaller batch must not shrink the launch below what the largest possible batch walks in 512 rounds)
                    if (w.wave_form == 2) want = std::max(want, (int)((std::min<int64_t>(m->cfg.max_nnz, m->N) + 2047) / 2048));
                    w.nitem = std::min(w.nitem, std::max(256, want));
                }
            }
        }
        return AAE_OK;
    }
    // (done: an event riding on the launch's completion - only taken when nothing of this builder follows the launch)
    // the chain program in front of this launch left its adversarial loss as n per-row terms: one more workgroup sums them
    void add_loss(aae_model* m, int n, int slot) { g.loss.enabled = 1; g.loss.terms = m->adv_terms; g.loss.n = n; g.loss.out = m->losses + slot; }
    int launch(hipStream_t s, hipEvent_t done = nullptr, bool* marked = nullptr) {
        int blocks = tiles;
        if (g.w1.enabled) { g.w1.blk0 = tiles; blocks += g.w1.ncol + g.w1.nitem; }
        if (g.loss.enabled) blocks += 1;
        if (marked) *marked = done && !mh;
        // wide batches (r5): jobs of 256 rows and more take the k-split form of a tile (chain.h: every wave the whole tile over a
        // quarter of the rows, operands straight into the matrix instructions' registers, no LDS in the loop).  Per launch of three
        // 200 x 201 layers + the bias column sums, slab loop | k-split form (tools/debug/ubench/dw_real.hip): 256 rows 9.7 | 9.2 us,
        // 512 12.8 | 10.4, 800 17.0 | 13.3, 1 000 19.4 | 13.6, 2 000 32.0 | 19.4 (a 16-wave kernel with four k-groups of the slab
        // loop, this round's first answer for 1 536+ rows: 23.3 - removed).  AAE_DW_KSPLIT_ROWS (read by aae_create) = 0: never, 1: every batch (tests)
        g.ksplit = ksplit_rows;
        // AAE_DW_TS=<first launch to report>: per-workgroup clocks of six launches (three steps' worth would be nine), by block kind
        const int ts_from = dw_ts;
        static int ts_seen = 0;
        static unsigned long long* ts_dev = nullptr;
        constexpr int kTsCap = 1 << 16;
        const bool ts_now = ts_from >= 0 && ts_seen >= ts_from && ts_seen < ts_from + 6;
        if (ts_from >= 0) {
            ++ts_seen;
            if (!ts_dev) HIPCHK(hipMalloc(&ts_dev, sizeof(unsigned long long) * 2 * kTsCap));
        }
        if (ts_now) { HIPCHK(hipMemsetAsync(ts_dev, 0, sizeof(unsigned long long) * 2 * kTsCap, s)); g.ts = ts_dev; g.ts_cap = kTsCap; }
        if (done && !mh) hipExtLaunchKernelGGL(grouped_dw_kernel, dim3(blocks), dim3(256), 0, s, nullptr, done, 0, g);
        else
        hipLaunchKernelGGL(grouped_dw_kernel, dim3(blocks), dim3(256), 0, s, g);
        LAUNCHCHK("grouped_dw_kernel");
        if (ts_now) {
            HIPCHK(hipStreamSynchronize(s));
            const int nb = std::min(blocks, kTsCap);
            std::vector<unsigned long long> h(2 * (size_t)nb);
            HIPCHK(hipMemcpy(h.data(), ts_dev, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost));
            unsigned long long t_lo = ~0ull, t_hi = 0;
            for (int b = 0; b < nb; ++b) if (h[2 * b]) { t_lo = std::min(t_lo, h[2 * b]); t_hi = std::max(t_hi, h[2 * b + 1]); }
            const int ncol = g.w1.enabled ? g.w1.ncol : 0, nitem = g.w1.enabled ? g.w1.nitem : 0;
            int max_rows = 0;
            for (int q = 0; q < g.njobs; ++q) max_rows = std::max(max_rows, g.jobs[q].rows);
            fprintf(stderr, "[grouped_dw launch %d: %d workgroups = %d tiles (rows %d) + %d column-sum + %d item (form %d) + %d loss; first start -> last end %.2f us]\n",
                    ts_seen - 1, blocks, tiles, max_rows, ncol, nitem, g.w1.wave_form, g.loss.enabled, (t_hi - t_lo) * 0.01);
            const struct { const char* name; int lo, hi; } kinds[] = {{"tiles", 0, tiles}, {"column sums", tiles, tiles + ncol}, {"items", tiles + ncol, tiles + ncol + nitem}};
            for (const auto& k : kinds) {
                if (k.hi <= k.lo) continue;
                unsigned long long first = ~0ull, last = 0, worst = 0; int worst_b = -1, busy = 0; double sum = 0;
                for (int b = k.lo; b < std::min(k.hi, nb); ++b) {
                    if (!h[2 * b]) continue;
                    const unsigned long long d = h[2 * b + 1] - h[2 * b];
                    first = std::min(first, h[2 * b]); last = std::max(last, h[2 * b + 1]);
                    if (d > worst) { worst = d; worst_b = b - k.lo; }
                    if (d > 150) ++busy;      // (a workgroup with nothing to do leaves within 1.5 us)
                    sum += d * 0.01;
                }
                fprintf(stderr, "   %-12s start %.2f us after the launch's first, end %.2f; longest workgroup %.2f us (#%d); %d of %d ran longer than 1.5 us, mean %.2f us\n",
                        k.name, (first - t_lo) * 0.01, (last - t_lo) * 0.01, worst * 0.01, worst_b, busy, k.hi - k.lo, sum / std::max(1, k.hi - k.lo));
            }
        }
        if (mh) TRY(launch_w1_hot(mh, s));
        return AAE_OK;
    }
};

// the head items a wave-form first-layer update left on its hot list: the workgroup form over that list (w1_update.h)
int launch_w1_hot(aae_model* m, hipStream_t s) {
    if (!m->w1_hot_pending) return AAE_OK;
    m->w1_hot_pending = false;
    W1Items a = w1_items_args(m, m->w1_hot_ga1, 0, 0, m->w1_hot_which);
    a.ulist = m->hot_list + (size_t)m->w1_hot_set * m->hot_cap; a.ucount = m->hot_count + m->w1_hot_set;
    const size_t lds = sizeof(int) * w1_items_lds_words(m->rows);
    hipLaunchKernelGGL(w1_item_update_kernel, dim3(std::min(m->hot_cap, 256)), dim3(256), lds, s, a);
    LAUNCHCHK("w1_item_update (hot list)");
    return AAE_OK;
}

// Encoder hidden stack from the gathered first layer (eh1 in global): lin2, lin3, output activation.
// ops appended to `cb`; z ends in slot 2.
void chain_encoder_tail(aae_model* m, ChainBuilder& cb, bool train, const uint8_t* mk2, uint32_t sid2, int rows,
                        float* eh2_out, hipStream_t s, const uint8_t* mk1 = nullptr, uint32_t sid1 = 0) {
    const int h = m->h;
    if (m->ext_first) {
        // the first layer lives with the caller (aae_set_first_layer_external): a1 -> dropout -> activation here
        cb.add(cop_load(m->a1.p, m->ldh, 3, h));
        ChainOp& e1 = cb.add(cop(COP_DROPACT, 3, 0, h));
        e1.d = make_drop(m, 0, train, mk1, nullptr, rows, h, sid1); e1.one_col = h; cop_out(e1, m->eh1.p, m->ldh);
    } else {
        ChainOp& l = cb.add(cop_load(m->eh1.p, m->ldh, 0, h)); l.one_col = h;
    }
    ChainOp& a = cb.add(cop_fwd(m, P_W2, 0, 1, h + 1, h, CEPI_DROPACT, s));
    a.d = make_drop(m, 1, train, mk2, nullptr, rows, h, sid2); a.one_col = h;
    if (eh2_out) cop_out(a, eh2_out, m->ldh);
    cb.add(cop_fwd(m, P_W3, 1, 2, h + 1, m->c, CEPI_NONE, s));
}

// The fused decoder's tile buckets depend on the batch only: the step's first chain launch carries their builder
// as one extra workgroup (chain.h), off the critical path.
static int row_blocks(const aae_model* m) { return m->rows <= 16 * kMB ? 1 : (m->rows + kRowBlock - 1) / kRowBlock; }
// The fused output-layer kernels address dec.lin3 (and its moments) as  descriptor base + tile's byte offset (a 32-bit scalar
// register) + the lane's offset inside the tile (vector register; 2^31 for a lane without a cell: dropped by the descriptor's
// range check).  The range check sees the SUM of both offsets (measured, r5: with a fixed descriptor the tiles beyond byte 2^31
// of a 2.9 M-item layer read zeros and dropped their stores), so r1-r4 kept these layers below 2^31 bytes - 2.63 M items at
// hidden 200, while PubMed's and ACM's vocabularies (nmi.txt:68,85 of the reference) are 2.9 M and 2.6 M.  Since r5 the
// descriptors cover a moving WINDOW of the tensors (X3Window, dec_fused.h) and the scalar offset counts from the window's first
// tile: the limit is the Gt buffer's and the 32-bit tile arithmetic's, 4 GiB.
static size_t fused_span_limit(const aae_model* m) {
    const int bits = out_bf16(m) ? 31 : 32;     // (dec_fused_bf16.h's kernels keep fixed descriptors)
    return ((size_t)1 << bits) - ((size_t)1 << 24);       // (head room: the padding tiles behind the tensor, the last span of a tile)
}
static bool fused_decoder_applies(const aae_model* m) {
    const bool one = m->rows <= 16 * kMB;
    // The row-blocked form pays while its deferred half (2 * rows * N * (h + 1) flop of GEMM2 at the optimiser kernel's
    // ~30 TFLOP/s) fits beside the rest of the step: 800 rows x 12.5 k items (an item slice at world 8) 0.40 against 0.46 ms
    // per step, 208 x 100 k 0.64 against 0.70; beyond ~32 M cells the next step waits for it and the three GEMMs win
    // (512 x 100 k: 1.48 against 1.12 ms; 512 x 275 k, a C5 slice: 3.7 against 2.7 ms).  AAE_BLOCKED_ANY lifts the cap (tests).
    // r3: with both launches on the emulated product (dec_crit_x3.h: the deferred half of all blocks in ONE launch for any
    // vocabulary, dec_opt_blocks_x3_kernel) the cap is gone: 512 x 100 k 0.77 ms/step against 0.93 on the three GEMMs.
    // (bf16 mode on the rounded-operand kernels: only with the one-launch deferred half - the per-block launches are fp32 kernels)
    const bool bf_blocks_ok = !m->bf16 || (m->x3_ok && m->dh2f.p && true);
    const bool blocked = !one && m->blocked_ok && !out_bf16(m) && bf_blocks_ok && m->split_ok && m->split_wgs > 0 && m->Gacc.p && row_blocks(m) <= kMaxRowBlocks &&
                         m->cfg.grad_mode == AAE_GRAD_FUSED &&
                         (m->blocked_any || (size_t)m->rows * m->N <= ((size_t)32 << 20) || (m->x3_ok && m->dh2f.p));
    return m->fused_ok && !m->force_unfused && (one || blocked) &&
           ((size_t)m->N + 2 * kTI) * m->ldh * sizeof(float) < fused_span_limit(m) &&
           /* (the stored dL/dlogits tiles of a row block, [tiles][rows][32] floats, stay behind ONE fixed descriptor) */
           ((size_t)m->N + 2 * kTI) * (size_t)((m->rows + row_blocks(m) - 1) / row_blocks(m)) * sizeof(float) < ((size_t)1 << 31) &&
           (m->bf16 ? dec_fused_bf16_lds_bytes(m->fused_nb) : dec_fused_lds_bytes((m->rows + row_blocks(m) - 1) / row_blocks(m), m->h)) <= 160 * 1024;
}
// counting sort of the running batch's entries into the fused output layer's 32-item tiles (buckets.h / dec_fused.h)
static void flip_bucket_set(aae_model* m) {
    std::swap(m->tstart, m->tstart2); std::swap(m->teb, m->teb2); std::swap(m->ten, m->ten2); std::swap(m->tev, m->tev2);
}
int build_tile_buckets(aae_model* m, hipStream_t s) {
    const int ntiles = (m->N + kTI - 1) / kTI, B = m->rows;
    flip_bucket_set(m);
    const size_t lds1 = sizeof(int) * ((size_t)ntiles + 1 + kBucketMaxDocs + 1 + 1024);
    if (ntiles <= kBucketMaxTiles && B <= kBucketMaxDocs && (m->fused_ok || lds1 <= 48 * 1024)) {      // (fused_ok: the LDS limit of the kernel was raised)
        hipLaunchKernelGGL(tile_bucket_kernel, dim3(1), dim3(1024), lds1, s, m->bv, ntiles, m->tstart, m->teb, m->ten, m->tev);
    } else if (ntiles <= kBucketMaxTiles && B <= kBucketWideDocs && m->bucket_wide_ok) {
        // (the global batch of an item slice: one launch instead of four, 25 -> 9 us)
        size_t lds = sizeof(int) * ((size_t)ntiles + 1 + kBucketWideDocs + 1 + 1024);
        // r5: this ONE workgroup runs ~90 us of LDS atomics beside the step (side stream).  With its natural ~9 KB of LDS the
        // dispatcher dealt workgroups of the step's own launches onto its CU, and whichever tile of a weight-gradient launch
        // landed there took 2.4x its time - the launch waits for its slowest workgroup: 31 or 75 us per launch at C4, the
        // generator program 32 or 57, by where the workgroups fell (profiles/r5_step_timeline_c4.txt).  It claims the CU's LDS
        // now, as the deferred launch does: nothing of the step fits beside it.
        constexpr bool bk_claim = true;
        if (bk_claim) lds = std::max(lds, (size_t)(160 * 1024 - 16384));
        hipLaunchKernelGGL(tile_bucket_wide_kernel, dim3(1), dim3(1024), lds, s, m->bv, ntiles, m->tstart, m->teb, m->ten, m->tev);
    } else {
        const int gy = std::max(1, std::min(16, m->chunks / 16 + 1));
        hipLaunchKernelGGL(zero_int_kernel, dim3(std::min(64, ntiles / 256 + 1)), dim3(256), 0, s, m->tcount, ntiles + 1);
        hipLaunchKernelGGL(tile_hist_kernel, dim3(B, gy), dim3(256), 0, s, m->bv, m->tcount);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, m->tcount, m->tstart, ntiles);
        hipLaunchKernelGGL(tile_fill_kernel, dim3(B, gy), dim3(256), 0, s, m->bv, m->tstart, m->tcount, m->teb, m->ten, m->tev);
    }
    LAUNCHCHK("tile buckets");
    m->buckets_valid = true;
    return AAE_OK;
}
// the tile buckets of the running batch exist and are visible to stream s (the first layer's update reads them, w1_update.h)
int ensure_buckets(aae_model* m, hipStream_t s) {
    if (m->bk_pending) {                // built on the side stream (aae_first_layer_forward)
        HIPCHK(hipStreamWaitEvent(s, m->ev_bk, 0));
        m->bk_pending = false;
    }
    if (!m->buckets_valid) TRY(build_tile_buckets(m, s));
    return AAE_OK;
}

// The sparse first layer's weight gradient over the running batch and optimiser `which` on the touched rows (or the
// gradient rows -> AAE_T_GRAD + ENC_W1T in export mode), in a fixed summation order (w1_update.h)
W1Items w1_items_args(aae_model* m, const float* ga1, int rpb, size_t bstride, int which) {
    const int set = (which == O_GEN) ? 1 : 0;
    W1Items a;
    a.ulist = m->ulist; a.ucount = m->ucount;
    a.tstart = m->tstart; a.eb = m->teb; a.en = m->ten; a.ev = m->tev;
    a.ga1 = ga1; a.ld = m->ldh; a.rpb = rpb; a.bstride = bstride;
    a.rscale = m->rscale; a.rows = m->rows; a.h = m->h;
    a.W = m->P[P_W1T].p; a.M = m->M[set][P_W1T].p; a.V = m->V[set][P_W1T].p; a.ldw = m->ldw1;
    a.gout = m->cfg.grad_mode == AAE_GRAD_EXPORT ? m->Gr[P_W1T].p : nullptr;
    a.sc = m->sc + which; a.tsync = m->tsync; a.step_ctr = m->step_ctr;
    a.mark_synced = (which == O_GEN || m->ae_only) ? 1 : 0;
    return a;
}
int launch_w1_items(aae_model* m, const float* ga1, int rpb, size_t bstride, int which, hipStream_t s) {
    TRY(ensure_buckets(m, s));
    const W1Items a = w1_items_args(m, ga1, rpb, bstride, which);
    // one 256-thread workgroup per item
    const size_t lds = sizeof(int) * w1_items_lds_words(m->rows);
    if (lds > 64 * 1024 && !m->w1_big_lds) return fail(AAE_ESTATE, "first-layer update: batch too large for the LDS row lists");
    ProfScope ps(m, AAE_K_ENC_W1_ADAM, s);
    // (grid-stride over the distinct items: never more workgroups than the vocabulary - an item slice - has items)
    const int items = std::max(1, std::min(std::min(m->cfg.max_nnz, std::max(256, m->rows * 32)), m->N));
    hipLaunchKernelGGL(w1_item_update_kernel, dim3(items), dim3(256), lds, s, a);
    LAUNCHCHK("w1_item_update");
    return AAE_OK;
}

static void piggyback_buckets(aae_model* m, ChainBuilder& cb) {
    const int ntiles = (m->N + kTI - 1) / kTI;
    const size_t need = sizeof(int) * ((size_t)ntiles + 1 + kBucketMaxDocs + 1 + 1024);
    if (m->x16_ok && m->rows >= m->x16_rows) return;      // (the wide-batch chain kernel carries no builder; only a forced AAE_X16_ROWS meets a batch this small)
    if (m->buckets_valid || !fused_decoder_applies(m) || ntiles > kBucketMaxTiles || m->rows > kBucketMaxDocs ||
        need > (size_t)kCSlots * kCR * kCL * sizeof(float))
        return;
    flip_bucket_set(m);
    BucketJob& b = cb.P.bk;
    b.bv = m->bv; b.ntiles = ntiles; b.tstart = m->tstart; b.eb = m->teb; b.en = m->ten; b.ev = m->tev; b.enabled = 1;
    m->buckets_valid = true;
}

// ae forward after the gather: encoder tail (+ optionally the decoder's two hidden layers)
int chain_ae_forward(aae_model* m, bool with_dec, const float* cond_dev, float* z_out, hipStream_t s) {
    const int B = m->rows, h = m->h, c = m->c, cp = m->cp;
    const aae_rng_inject& I = m->inj;
    ChainBuilder cb(m, B);
    piggyback_buckets(m, cb);
    chain_encoder_tail(m, cb, true, I.masks_dev[1], 1, B, m->eh2.p, s, I.masks_dev[0], 0);
    // the encoder's output activation; the identity (gauss prior, aae.py:97-101) is no op of its own: its stores and
    // bias-input column ride on the last linear layer
    ChainOp& f = m->cfg.enc_final == AAE_FINAL_LINEAR ? cb.P.ops[cb.P.nops - 1] : cb.add(cop(COP_FINAL_FWD, 2, 2, c));
    f.aux = m->cfg.enc_final;
    cop_out(f, m->zc.p, m->ldc); f.out2 = m->zsave.p; f.ldo2 = m->ldz;
    if (z_out) { ChainOp& st = cb.add(cop(COP_STORE, 2, 2, c)); cop_out(st, z_out, c); }
    if (with_dec) {
        if (wide_dec_in(m)) {
            // the condition's first columns behind z in slot 2 (208 columns in all), the others + the constant 1 in slot 5
            const int nA = kCWide - c, ci = m->cfg.cond_inc;
            ChainOp& ca = cb.add(cop_load(cond_dev, ci, 2, nA)); ca.dst_col0 = c;
            ChainOp& cl = cb.add(cop_load(cond_dev + nA, ci, 5, ci - nA)); cl.one_col = cp - kCWide;
        } else if (m->cfg.cond_inc > 0) {
            ChainOp& cl = cb.add(cop_load(cond_dev, m->cfg.cond_inc, 2, m->cfg.cond_inc)); cl.dst_col0 = c; cl.one_col = cp;
        } else {
            f.one_col = cp;
        }
        ChainOp& v1 = add_dec_in_fwd(cb, m, 2, 5, 3, s);
        v1.d = make_drop(m, 0, true, I.masks_dev[2], nullptr, B, h, 2); v1.one_col = h; cop_out(v1, m->dh1.p, m->ldh);
        ChainOp& v2 = cb.add(cop_fwd(m, P_V2, 3, 4, h + 1, h, CEPI_DROPACT, s));
        v2.d = make_drop(m, 1, true, I.masks_dev[3], nullptr, B, h, 3); v2.one_col = h; cop_out(v2, m->dh2.p, m->ldh);
    }
    m->dec_hidden_done = with_dec;
    return launch_chain(m, cb, s);
}

// decoder hidden layers from zc (global): split API and predict
int chain_dec_hidden(aae_model* m, bool train, int rows, hipStream_t s) {
    const int h = m->h, cp = m->cp;
    const aae_rng_inject& I = m->inj;
    ChainBuilder cb(m, rows);
    if (wide_dec_in(m)) {
        cb.add(cop_load(m->zc.p, m->ldc, 0, kCWide));
        ChainOp& lb = cb.add(cop_load(m->zc.p + kCWide, m->ldc, 3, cp - kCWide)); lb.one_col = cp - kCWide;
    } else {
        ChainOp& l = cb.add(cop_load(m->zc.p, m->ldc, 0, cp)); l.one_col = cp;
    }
    ChainOp& v1 = add_dec_in_fwd(cb, m, 0, 3, 1, s);
    v1.d = make_drop(m, 0, train, I.masks_dev[2], nullptr, rows, h, 2); v1.one_col = h; cop_out(v1, m->dh1.p, m->ldh);
    ChainOp& v2 = cb.add(cop_fwd(m, P_V2, 1, 2, h + 1, h, CEPI_DROPACT, s));
    v2.d = make_drop(m, 1, train, I.masks_dev[3], nullptr, rows, h, 3); v2.one_col = h; cop_out(v2, m->dh2.p, m->ldh);
    return launch_chain(m, cb, s);
}

// decoder backward below the output layer (+ optionally the encoder backward) as one program.
//   dec: sum of the 16 dA2 partial slabs -> act'/dropout -> V2 -> V1 -> gzc
//   enc: dz (slot or external) -> output activation' -> W3 -> W2 -> ga1
int chain_ae_backward(aae_model* m, bool dec_part, bool enc_part, const float* part_slabs, size_t slab_stride,
                      const float* gz_ext, int ld_gz, float* dzc_out, int which, hipStream_t s, int nslab = 16) {
    const int B = m->rows, h = m->h, c = m->c, cp = m->cp;
    const aae_rng_inject& I = m->inj;
    ChainBuilder cb(m, B);
    if (dec_part) {
        if (part_slabs) {
            // sum of the dA2 partial slabs times act'(dh2) and the dropout scale in one op (dh2 read from global)
            ChainOp& ss = cb.add(cop(COP_SLABSUM, 0, 2, h)); ss.W = part_slabs; ss.ldw = m->ldh; ss.aux = nslab; ss.stride = slab_stride;
            ss.epi = CEPI_ACTBWD; ss.aux_ptr = m->dh2.p; ss.aux_ld = m->ldh;
            ss.d = make_drop(m, 1, true, I.masks_dev[3], nullptr, B, h, 3); cop_out(ss, m->gb0.p, m->ldh);
        } else {
            cb.add(cop_load(m->gb0.p, m->ldh, 2, h));      // unfused decoder path: gb0 already holds dL/da2
        }
        if (y_needs_load(m)) cb.add(cop_load(m->dh1.p, m->ldh, 3, h));
        ChainOp& x2 = cb.add(cop_dx(m, P_V2, 2, 4, h, h, CEPI_ACTBWD, s)); set_y(cb, m, x2, m->dh1.p, m->ldh, 3);
        x2.d = make_drop(m, 0, true, I.masks_dev[2], nullptr, B, h, 2); cop_out(x2, m->gb1.p, m->ldh);
        add_dec_in_dx(cb, m, 4, 5, 3, dzc_out, s);        // (slot 3 - dh1, the y of the op above - is free again)
    }
    if (enc_part) {
        if (!dec_part || gz_ext) cb.add(cop_load(gz_ext ? gz_ext : m->gzc.p, gz_ext ? ld_gz : m->ldc, 5, c));
        // the encoder's output activation backward; for the identity (gauss prior) with nothing concatenated the dX op
        // that produced dL/dz stores it as the W3 weight-gradient operand itself and no op is needed
        int sg = 7;
        ChainOp* prod = (dec_part && !gz_ext && cb.P.nops) ? &cb.P.ops[cb.P.nops - 1] : nullptr;
        if (m->cfg.enc_final == AAE_FINAL_LINEAR && prod && cp == c && !prod->out2) {
            prod->out2 = m->ga3.p; prod->ldo2 = m->ldz; sg = 5;
        } else {
            if (m->cfg.enc_final != AAE_FINAL_LINEAR) cb.add(cop_load(m->zsave.p, m->ldz, 6, c));   // z: only the derivative of a softmax / sigmoid output needs it
            ChainOp& fb = cb.add(cop(COP_FINAL_BWD, 5, 7, c)); fb.yslot = 6; fb.aux = m->cfg.enc_final;
            cop_out(fb, m->ga3.p, m->ldz);
        }
        if (y_needs_load(m)) cb.add(cop_load(m->eh2.p, m->ldh, 8, h));
        ChainOp& x3 = cb.add(cop_dx(m, P_W3, sg, 9, c, h, CEPI_ACTBWD, s)); set_y(cb, m, x3, m->eh2.p, m->ldh, 8);
        x3.d = make_drop(m, 1, true, I.masks_dev[which == O_GEN ? 9 : 1], nullptr, B, h, which == O_GEN ? 9 : 1);
        cop_out(x3, m->gb2.p, m->ldh);
        if (y_needs_load(m)) cb.add(cop_load(m->eh1.p, m->ldh, 0, h));
        ChainOp& x2 = cb.add(cop_dx(m, P_W2, 9, 1, h, h, CEPI_ACTBWD, s)); set_y(cb, m, x2, m->eh1.p, m->ldh, 0);
        x2.d = make_drop(m, 0, true, I.masks_dev[which == O_GEN ? 8 : 0], nullptr, B, h, which == O_GEN ? 8 : 0);
        cop_out(x2, ga1_ptr(m), m->ldh);
    }
    return launch_chain(m, cb, s);
}

// first encoder layer's weight gradient (sparse scatter), bias gradient and their optimiser
int encoder_first_layer_update(aae_model* m, const float* ga1, int which, hipStream_t s, bool merged = false) {
    const int B = m->rows, h = m->h;
    const int set = (which == O_GEN) ? 1 : 0;
    const bool exportg = m->cfg.grad_mode == AAE_GRAD_EXPORT;
    if (m->dense_step) {
        // dW1T [N][h] = x^T [N][B] * dL/da1 [B][h] with the fused optimiser on EVERY row (torch.optim.Adam is dense), then
        // all rows carry this step
        if (!merged) {
            hipLaunchKernelGGL(colsum_adam_kernel, dim3((h + 63) / 64), dim3(1024), 0, s, ga1, B, h, m->ldh, m->P[P_B1].p,
                               m->M[set][P_B1].p, m->V[set][P_B1].p, (float*)nullptr, m->sc + which);
            LAUNCHCHK("colsum_adam");
        }
        TRY(linear_dw(m, m->Xn.p, m->ldn, B, ga1, m->ldh, P_W1T, which, s));
        hipLaunchKernelGGL(fill_tsync_kernel, dim3(grid1d((size_t)m->N)), dim3(256), 0, s, m->tsync, m->N, m->step_ctr);
        LAUNCHCHK("fill_tsync");
        return AAE_OK;
    }
    if (!merged) {
        hipLaunchKernelGGL(colsum_adam_kernel, dim3((h + 63) / 64), dim3(1024), 0, s, ga1, B, h, m->ldh, m->P[P_B1].p,
                           m->M[set][P_B1].p, m->V[set][P_B1].p, exportg ? m->Gr[P_B1].p : (float*)nullptr, m->sc + which);
        LAUNCHCHK("colsum_adam");
    }
    if (merged && m->w1_items_merged) { m->w1_items_merged = false; return AAE_OK; }      // (rode in the grouped dW launch)
    // (export mode: the gradient rows -> AAE_T_GRAD + ENC_W1T; aae_w1_export / exchange / aae_w1_import follow)
    return launch_w1_items(m, ga1, 0, 0, which, s);
}

// done_ev: an event that rides on the launch's completion signal (the side stream's "the step has begun" mark)
// head: tell the side stream that the main stream has passed this launch (ev_head rides on its completion signal)
int gather_first_layer(aae_model* m, bool train, const uint8_t* mk1, uint32_t sid1, hipStream_t s, bool head = false,
                       bool open_step = false) {
    const int B = m->rows, h = m->h;
    DropSpec d1 = make_drop(m, 0, train, mk1, nullptr, B, h, sid1);
    ProfScope ps(m, AAE_K_ENC_GATHER, s);
    size_t shm = (size_t)16 * r4(h) * sizeof(float);
    // wide batches: four waves per document (eight documents per CU instead of two; a document is ~20 entries: five per wave).
    // ms/step, 16 | 4 waves: C3 at batch 512 0.651 | 0.646, C4 (1 000 rows) 0.389 | 0.381; C2's shape at batch 500 0.327 | 0.331,
    // C3 at batch 100 0.239 | 0.251 (tools/debug/gather4_ab.sh)
    constexpr int g4_rows = 512;
    if (B >= g4_rows) {
        hipExtLaunchKernelGGL(enc_gather_kernel_t<4>, dim3(B), dim3(256), (uint32_t)(shm / 4), s, nullptr, head ? m->ev_head : nullptr, 0, m->bv,
                              (const float*)m->P[P_W1T].p, m->ldw1, (const float*)m->P[P_B1].p, h, (int)m->cfg.normalize_inputs,
                              m->a1.p, m->eh1.p, m->ldh, (int)m->cfg.activation, d1, (uint64_t)m->cfg.seed,
                              (const long long*)m->step_ctr, m->rscale, m->doc_l1,
                              AdvanceJob{m->sc, m->step_ctr, m->lazy ? m->tab : nullptr, m->losses, open_step ? 1 : 0,
                                         head ? m->stamp2 : nullptr, head ? m->ucount2 : nullptr},
                              (long long)(open_step ? m->hstep : -1));
    } else
    hipExtLaunchKernelGGL(enc_gather_kernel, dim3(B), dim3(1024), (uint32_t)shm, s, nullptr, head ? m->ev_head : nullptr, 0, m->bv,
                          (const float*)m->P[P_W1T].p, m->ldw1, (const float*)m->P[P_B1].p, h, (int)m->cfg.normalize_inputs,
                          m->a1.p, m->eh1.p, m->ldh, (int)m->cfg.activation, d1, (uint64_t)m->cfg.seed,
                          (const long long*)m->step_ctr, m->rscale, m->doc_l1,
                          // open_step: the step-opening bookkeeping rides in this launch (advance_step_body) and the
                          // workgroups take the step number from the host (m->hstep == *step_ctr once the step is open)
                          // head: the prefetch's first launch (new stamp, empty list) rides here as well - the catch-up behind it then
                          // starts a launch earlier, ahead of the output layer's critical launch instead of beside its first workgroups
                          AdvanceJob{m->sc, m->step_ctr, m->lazy ? m->tab : nullptr, m->losses, open_step ? 1 : 0,
                                     head ? m->stamp2 : nullptr, head ? m->ucount2 : nullptr},
                          (long long)(open_step ? m->hstep : -1));
    LAUNCHCHK("enc_gather");
    m->pf_bumped = head;
    return AAE_OK;
}

static bool same_batch(const aae_batch& a, const aae_batch& b) {
    // (ABI 3: pointer identity is not content identity - the caller's generation id has to agree as well, and 0 means "none given")
    return a.indptr_dev == b.indptr_dev && a.indices_dev == b.indices_dev && a.values_dev == b.values_dev &&
           a.rows_dev == b.rows_dev && a.row_start == b.row_start && a.n_rows == b.n_rows &&
           a.generation != 0 && a.generation == b.generation;
}

// aae_prefetch_batch, second half: the hinted batch's unique-item list + deferred-Adam catch-up (through the RUNNING
// step, whose scalars advance_step has published by the time ev_head fires) on the side stream, into the second list
// set.  Rows of the running batch are skipped: the step's own updates bring them to the same step.
// (early: behind the mark on the step BEFORE this one - ev_end - instead of on this step's opening gather)
int launch_prefetch(aae_model* m, bool wait_head = true, bool early = false) {
    const aae_batch& b = m->pf_batch;
    m->pf_armed = false;
    m->pf_this_step = true;
    m->pf_after_opt = !wait_head;
    if (!m->side || !m->mark2 || !m->lazy) return AAE_OK;
    BatchView bv; bv.indptr = b.indptr_dev; bv.indices = b.indices_dev; bv.values = b.values_dev;
    bv.rows = b.rows_dev; bv.row_start = b.row_start; bv.n_rows = b.n_rows;
    const int mr = b.max_row_nnz > 0 ? b.max_row_nnz : 1024;
    const int chunks = std::max(1, std::min(64, (mr + 15) / 16));
    const int gy = std::max(1, std::min(16, chunks / 16 + 1));
    hipStream_t q = m->side;
    if (early) HIPCHK(hipStreamWaitEvent(q, m->ev_end, 0));
    else if (wait_head) HIPCHK(hipStreamWaitEvent(q, m->ev_head, 0));      // (else: the caller enqueues behind work that is ordered behind the step's head)
    if (early || !(wait_head && m->pf_bumped)) hipLaunchKernelGGL(bump_stamp_kernel, dim3(1), dim3(1), 0, q, m->stamp2, m->ucount2);
    m->pf_bumped = false;
    hipLaunchKernelGGL(uniq_items_kernel, dim3(b.n_rows, gy), dim3(256), 0, q, bv, m->mark2, m->stamp2, m->ulist2, m->ucount2);
    if (m->cfg.optimizer == AAE_OPT_ADAM) {
        const int grid = std::min(m->cfg.max_nnz, std::max(256, b.n_rows * 32));
        hipLaunchKernelGGL(w1_catchup_kernel, dim3(grid), dim3(256), 0, q, m->ulist2, m->ucount2, m->N, m->tsync,
                           m->P[P_W1T].p, m->M[0][P_W1T].p, m->V[0][P_W1T].p, m->M[1][P_W1T].p, m->V[1][P_W1T].p,
                           m->ldw1, m->h, m->tab, m->step_ctr, 0, m->mark, m->stamp, early ? m->hstep : -1ll);
    }
    LAUNCHCHK("prefetch (unique items + catch-up of the next batch)");
    TRY(side_done(m, m->ev_pf));
    m->pf_pending = true;
    m->pf_built = true; m->pf_step = m->hstep + 1; m->pf_built_batch = b;
    return AAE_OK;
}

// ---- VAE (reference vae.py:47-266) -----------------------------------------------------------
// forward: eh1 (the gather's act(fc1 x)) -> [mu | logvar] = [fc21; fc22] eh1 -> z = mu + eps * exp(logvar/2)
// -> (constant condition block) -> dh2 = act(fc3 z): the input of the vocabulary-wide output layer fc4
int chain_vae_forward(aae_model* m, const float* cond_dev, const float* eps_dev, int rows, hipStream_t s) {
    const int h = m->h, c = m->c, cp = m->cp;
    ChainBuilder cb(m, rows);
    ChainOp& l = cb.add(cop_load(m->eh1.p, m->ldh, 0, h)); l.one_col = h;
    ChainOp& ml = cb.add(cop_fwd(m, P_W3, 0, 1, h + 1, 2 * c, CEPI_NONE, s));
    cop_out(ml, m->mulv.p, (int)m->mulv.ld);
    ChainOp& rp = cb.add(cop(COP_REPARAM, 1, 2, c));
    rp.W = eps_dev; rp.ldw = c; rp.aux = 12; rp.aux_ptr = m->veps.p; rp.aux_ld = (int)m->veps.ld;
    if (m->cfg.cond_inc > 0) {
        ChainOp& cl = cb.add(cop_load(cond_dev, m->cfg.cond_inc, 2, m->cfg.cond_inc)); cl.dst_col0 = c; cl.one_col = cp;
    } else {
        rp.one_col = cp;
    }
    ChainOp& st = cb.add(cop(COP_STORE, 2, 2, cp)); cop_out(st, m->zc.p, m->ldc);
    ChainOp& v1 = cb.add(cop_fwd(m, P_V1, 2, 3, cp + 1, h, CEPI_DROPACT, s));
    v1.one_col = h; cop_out(v1, m->dh2.p, m->ldh);        // no dropout in the VAE: DropSpec stays disabled
    return launch_chain(m, cb, s);
}

// the two halves of chain_vae_forward for a caller that imposes its conditions between them (aae_vae_encode / the
// decoder half inside aae_vae_decode_backward and aae_decode)
int chain_vae_encode(aae_model* m, const float* eps_dev, float* z_out, int rows, hipStream_t s) {
    const int h = m->h, c = m->c;
    ChainBuilder cb(m, rows);
    ChainOp& l = cb.add(cop_load(m->eh1.p, m->ldh, 0, h)); l.one_col = h;
    ChainOp& ml = cb.add(cop_fwd(m, P_W3, 0, 1, h + 1, 2 * c, CEPI_NONE, s));
    cop_out(ml, m->mulv.p, (int)m->mulv.ld);
    ChainOp& rp = cb.add(cop(COP_REPARAM, 1, 2, c));
    rp.W = eps_dev; rp.ldw = c; rp.aux = 12; rp.aux_ptr = m->veps.p; rp.aux_ld = (int)m->veps.ld;
    ChainOp& st = cb.add(cop(COP_STORE, 2, 2, c)); cop_out(st, m->zc.p, m->ldc);
    if (z_out) { st.out2 = z_out; st.ldo2 = c; }
    return launch_chain(m, cb, s);
}
int chain_vae_dec_hidden(aae_model* m, int rows, hipStream_t s) {
    const int h = m->h, cp = m->cp;
    ChainBuilder cb(m, rows);
    ChainOp& l = cb.add(cop_load(m->zc.p, m->ldc, 2, cp)); l.one_col = cp;
    ChainOp& v1 = cb.add(cop_fwd(m, P_V1, 2, 3, cp + 1, h, CEPI_DROPACT, s));
    v1.one_col = h; cop_out(v1, m->dh2.p, m->ldh);
    return launch_chain(m, cb, s);
}
// ... and of chain_vae_backward: down to dL/d(decoder input) (-> gzc and the caller), then from dL/dz on
int chain_vae_backward_dec(aae_model* m, const float* part_slabs, size_t slab_stride, float* dzc_out, hipStream_t s) {
    const int B = m->rows, h = m->h, cp = m->cp;
    ChainBuilder cb(m, B);
    if (part_slabs) {
        ChainOp& ss = cb.add(cop(COP_SLABSUM, 0, 0, h)); ss.W = part_slabs; ss.ldw = m->ldh; ss.aux = 16; ss.stride = slab_stride;
        cb.add(cop_load(m->dh2.p, m->ldh, 1, h));
        ChainOp& ab = cb.add(cop(COP_ACTBWD, 0, 2, h)); ab.yslot = 1; cop_out(ab, m->gb0.p, m->ldh);
    } else {
        cb.add(cop_load(m->gb0.p, m->ldh, 2, h));
    }
    ChainOp& dzc = cb.add(cop_linear(COP_LINEAR_DX, 2, 3, m->P[P_V1], h, cp, CEPI_NONE));
    cop_out(dzc, m->gzc.p, m->ldc);
    if (dzc_out) { dzc.out2 = dzc_out; dzc.ldo2 = cp; }
    return launch_chain(m, cb, s);
}
int chain_vae_backward_enc(aae_model* m, const float* dz_dev, int ld_dz, hipStream_t s) {
    const int B = m->rows, h = m->h, c = m->c;
    ChainBuilder cb(m, B);
    cb.P.loss_slot = 1;                                   // KL sum -> losses[1]
    cb.add(cop_load(dz_dev, ld_dz, 3, c));
    cb.add(cop_load(m->mulv.p, (int)m->mulv.ld, 4, 2 * c));
    ChainOp& rb = cb.add(cop(COP_REPARAM_BWD, 3, 5, 2 * c)); rb.yslot = 4; rb.scale = m->grad_scale;
    rb.aux_ptr = m->veps.p; rb.aux_ld = (int)m->veps.ld; cop_out(rb, m->gmulv.p, (int)m->gmulv.ld);
    cb.add(cop_load(m->eh1.p, m->ldh, 6, h));
    ChainOp& x1 = cb.add(cop_linear(COP_LINEAR_DX, 5, 7, m->P[P_W3], 2 * c, h, CEPI_ACTBWD)); x1.yslot = 6;
    cop_out(x1, m->gb3.p, m->ldh);
    return launch_chain(m, cb, s);
}

// backward below the output layer: dL/d(dh2) -> fc3 -> dz -> (dmu, dlogvar) incl. the KL term -> [fc21; fc22] -> ga1
int chain_vae_backward(aae_model* m, const float* part_slabs, size_t slab_stride, hipStream_t s) {
    const int B = m->rows, h = m->h, c = m->c, cp = m->cp;
    ChainBuilder cb(m, B);
    cb.P.loss_slot = 1;                                   // KL sum -> losses[1]
    if (part_slabs) {
        ChainOp& ss = cb.add(cop(COP_SLABSUM, 0, 0, h)); ss.W = part_slabs; ss.ldw = m->ldh; ss.aux = 16; ss.stride = slab_stride;
        cb.add(cop_load(m->dh2.p, m->ldh, 1, h));
        ChainOp& ab = cb.add(cop(COP_ACTBWD, 0, 2, h)); ab.yslot = 1; cop_out(ab, m->gb0.p, m->ldh);
    } else {
        cb.add(cop_load(m->gb0.p, m->ldh, 2, h));         // unfused decoder path: gb0 already holds dL/d(pre-activation)
    }
    ChainOp& dzc = cb.add(cop_linear(COP_LINEAR_DX, 2, 3, m->P[P_V1], h, cp, CEPI_NONE));
    if (cp > c) cop_out(dzc, m->gzc.p, m->ldc);          // dL/d(decoder input): its condition columns train a device-native CategoricalCondition
    cb.add(cop_load(m->mulv.p, (int)m->mulv.ld, 4, 2 * c));
    ChainOp& rb = cb.add(cop(COP_REPARAM_BWD, 3, 5, 2 * c)); rb.yslot = 4; rb.scale = m->grad_scale;
    rb.aux_ptr = m->veps.p; rb.aux_ld = (int)m->veps.ld; cop_out(rb, m->gmulv.p, (int)m->gmulv.ld);
    cb.add(cop_load(m->eh1.p, m->ldh, 6, h));
    ChainOp& x1 = cb.add(cop_linear(COP_LINEAR_DX, 5, 7, m->P[P_W3], 2 * c, h, CEPI_ACTBWD)); x1.yslot = 6;
    cop_out(x1, m->gb3.p, m->ldh);
    return launch_chain(m, cb, s);
}

// disc_step on the chain path
int chain_disc_step(aae_model* m, hipStream_t s) {
    const int B = m->rows, h = m->h, c = m->c;
    const aae_rng_inject& I = m->inj;
    if (!m->ext_first) TRY(gather_first_layer(m, false, nullptr, 0, s));     // (external: the caller refreshed AAE_T_ACT_A1)
    // Enc_eval is row-local like the discriminator program behind it: with 4-row workgroups and a batch that is a
    // multiple of 4 it runs as a PREFIX of that program in the workgroups of the z_fake rows (ChainOp::row_lo), z_fake
    // handed over in a slot - one launch (and its ~4.5 us floor) less per step
    constexpr bool merge_ok = true;
    bool merged = merge_ok && m->use_chain4 && B % kR4 == 0 && !m->vae;
    for (int pid : {P_W2, P_W3, P_D1, P_D2}) merged = merged && m->PT[pid].p != nullptr;
    if (!merged) {   // z_fake = Enc_eval(X) -> zin rows [B, 2B)
        ChainBuilder cb(m, B);
        chain_encoder_tail(m, cb, false, nullptr, 0, B, nullptr, s);
        ChainOp& f = m->cfg.enc_final == AAE_FINAL_LINEAR ? cb.P.ops[cb.P.nops - 1] : cb.add(cop(COP_FINAL_FWD, 2, 2, c));
        f.aux = m->cfg.enc_final; cop_out(f, m->zin.p, m->ldz, B);
        TRY(launch_chain(m, cb, s));
    }
    {   // D on [z_real; z_fake], loss, and the activation-gradient half of its backward
        ChainBuilder cb(m, 2 * B);
        cb.beside_deferred(m);
        cb.P.loss_slot = 1; cb.P.loss_terms = m->adv_terms;
        if (merged) {
            chain_encoder_tail(m, cb, false, nullptr, 0, B, nullptr, s);
            if (m->cfg.enc_final != AAE_FINAL_LINEAR) { ChainOp& f = cb.add(cop(COP_FINAL_FWD, 2, 2, c)); f.aux = m->cfg.enc_final; }
            for (int i = 0; i < cb.P.nops; ++i) {
                ChainOp& o = cb.P.ops[i];
                o.row_lo = B;                               // program row r >= B = document r - B
                if (o.kind == COP_LOAD || o.out) o.out_row0 = -B;
            }
        }
        // rows [0, B): z_real drawn (or injected) right here; rows [B, 2B): z_fake of the program above
        ChainOp& l = cb.add(cop(COP_PRIOR, 0, 0, c)); l.one_col = c;
        l.W = m->zin.p; l.ldw = m->ldz; l.row_split = B; l.aux = m->cfg.prior;
        l.scale = m->cfg.has_prior_scale ? m->cfg.prior_scale : 1.0f;
        l.grow0 = m->rng_row0;
        l.aux_ptr = m->cfg.rng_mode == AAE_RNG_DEVICE ? nullptr : const_cast<float*>(I.z_real_dev); l.aux_ld = c;
        cop_out(l, m->zin.p, m->ldz);                      // the weight-gradient GEMM of D1 reads all 2B rows
        if (merged) l.fake_slot = 2;                       // (z_fake sits in slot 2 of the workgroup that just computed it)
        ChainOp& d1 = cb.add(cop_fwd(m, P_D1, 0, 1, c + 1, h, CEPI_DROPACT, s));
        d1.d = make_drop(m, 0, true, I.masks_dev[4], I.masks_dev[6], B, h, 4); d1.one_col = h; cop_out(d1, m->xh1.p, m->ldh);
        ChainOp& d2 = cb.add(cop_fwd(m, P_D2, 1, 2, h + 1, h, CEPI_DROPACT, s));
        d2.d = make_drop(m, 1, true, I.masks_dev[5], I.masks_dev[7], B, h, 5); d2.one_col = h; cop_out(d2, m->xh2.p, m->ldh);
        // D3 (h -> 1) + sigmoid + adversarial loss + its dX in one op
        ChainOp& x3 = cb.add(cop_linear(COP_DISC_HEAD, 2, 5, m->P[P_D3], h + 1, h, CEPI_ACTBWD)); x3.yslot = 2; x3.d = d2.d;
        x3.aux = 0; x3.row_split = B; x3.scale = m->grad_scale; x3.aux_ptr = m->ga3.p; x3.aux_ld = 4;
        cop_out(x3, m->gb0.p, m->ldh);
        ChainOp& x2 = cb.add(cop_dx(m, P_D2, 5, 6, h, h, CEPI_ACTBWD, s)); x2.yslot = 1; x2.d = d1.d;
        cop_out(x2, m->gb1.p, m->ldh);
        TRY(launch_chain(m, cb, s));
    }
    DwBuilder dw;
    dw.add(m, m->ga3.p, 4, m->xh2.p, m->ldh, 2 * B, P_D3, O_DISC);
    dw.add(m, m->gb0.p, m->ldh, m->xh1.p, m->ldh, 2 * B, P_D2, O_DISC);
    dw.add(m, m->gb1.p, m->ldh, m->zin.p, m->ldz, 2 * B, P_D1, O_DISC);
    dw.add_loss(m, 2 * B, 1);
    return dw.launch(s);
}

// gen_step on the chain path: everything between the shared gather and the weight gradients is
// row-local and runs as ONE program
int chain_gen_step(aae_model* m, hipStream_t s) {
    const int B = m->rows, h = m->h, c = m->c;
    const aae_rng_inject& I = m->inj;
    ChainBuilder cb(m, B);
    cb.beside_deferred(m);
    cb.P.loss_slot = 2; cb.P.loss_terms = m->adv_terms;
    cb.add(cop_load(m->a1.p, m->ldh, 0, h));
    ChainOp& e1 = cb.add(cop(COP_DROPACT, 0, 1, h));
    e1.d = make_drop(m, 0, true, I.masks_dev[8], nullptr, B, h, 8); e1.one_col = h; cop_out(e1, m->eh1.p, m->ldh);
    ChainOp& e2 = cb.add(cop_fwd(m, P_W2, 1, 2, h + 1, h, CEPI_DROPACT, s));
    e2.d = make_drop(m, 1, true, I.masks_dev[9], nullptr, B, h, 9); e2.one_col = h; cop_out(e2, m->eh2.p, m->ldh);
    ChainOp& l3 = cb.add(cop_fwd(m, P_W3, 2, 3, h + 1, c, CEPI_NONE, s));
    ChainOp& f = m->cfg.enc_final == AAE_FINAL_LINEAR ? l3 : cb.add(cop(COP_FINAL_FWD, 3, 3, c));
    f.aux = m->cfg.enc_final; f.one_col = c;
    ChainOp& d1 = cb.add(cop_fwd(m, P_D1, 3, 4, c + 1, h, CEPI_DROPACT, s));
    d1.d = make_drop(m, 0, true, I.masks_dev[10], nullptr, B, h, 10); d1.one_col = h;
    ChainOp& d2 = cb.add(cop_fwd(m, P_D2, 4, 5, h + 1, h, CEPI_DROPACT, s));
    d2.d = make_drop(m, 1, true, I.masks_dev[11], nullptr, B, h, 11); d2.one_col = h;
    ChainOp& x3 = cb.add(cop_linear(COP_DISC_HEAD, 5, 8, m->P[P_D3], h + 1, h, CEPI_ACTBWD)); x3.yslot = 5; x3.d = d2.d;
    x3.aux = 1; x3.row_split = B; x3.scale = m->grad_scale;
    ChainOp& x2 = cb.add(cop_dx(m, P_D2, 8, 9, h, h, CEPI_ACTBWD, s)); x2.yslot = 4; x2.d = d1.d;
    ChainOp& dz = cb.add(cop_dx(m, P_D1, 9, 0, h, c, CEPI_NONE, s));    // dL/dz
    int sg = 6;
    if (m->cfg.enc_final == AAE_FINAL_LINEAR) {        // identity output activation: dL/dz is dL/da3 already
        cop_out(dz, m->ga3.p, m->ldz); sg = 0;
    } else {
        ChainOp& fb = cb.add(cop(COP_FINAL_BWD, 0, 6, c)); fb.yslot = 3; fb.aux = m->cfg.enc_final;
        cop_out(fb, m->ga3.p, m->ldz);
    }
    ChainOp& w3 = cb.add(cop_dx(m, P_W3, sg, 7, c, h, CEPI_ACTBWD, s)); w3.yslot = 2; w3.d = e2.d;
    cop_out(w3, m->gb2.p, m->ldh);
    ChainOp& w2 = cb.add(cop_dx(m, P_W2, 7, 8, h, h, CEPI_ACTBWD, s)); w2.yslot = 1; w2.d = e1.d;
    cop_out(w2, ga1_ptr(m), m->ldh);
    TRY(launch_chain(m, cb, s));
    DwBuilder dw;
    dw.add(m, m->ga3.p, m->ldz, m->eh2.p, m->ldh, B, P_W3, O_GEN);
    dw.add(m, m->gb2.p, m->ldh, m->eh1.p, m->ldh, B, P_W2, O_GEN);
    TRY(dw.add_first_layer(m, ga1_ptr(m), O_GEN, s));
    dw.add_loss(m, B, 2);
    // the step's last launch when the first layer's update rode in it: the side stream's mark for the NEXT step's early
    // prefetch rides on its completion (only while batches are being named ahead: a mark nobody waits for costs ~2 us)
    const bool mark_end = m->early_enabled && m->pf_this_step && m->w1_items_merged && !m->ext_first && m->side && m->ev_end &&
                          (m->rows > 16 * kMB || m->early_any);
    bool marked = false;
    TRY(dw.launch(s, mark_end ? m->ev_end : nullptr, &marked));
    m->end_marked = marked;
    if (m->ext_first && !m->own_first) return AAE_OK;           // dL/d(a1) waits in AAE_T_ACT_GA1 for the owner(s) of the first layer
    return encoder_first_layer_update(m, m->gb3.p, O_GEN, s, true);
}

}  // namespace
