// The handle: error reporting, tensors and the arena layout of an aae_model (reference aaerec/aae.py:782-804), config
// validation, stream / profiling / join helpers, dropout specs.
// (one of the parts of aae_abi.hip's translation unit: included there in order, not on its own)
#pragma once

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(AAE_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)
#define LAUNCHCHK(what)                                                                     \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess) return fail(AAE_EHIP, std::string(what) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define TRY(expr) do { int rc_ = (expr); if (rc_ != AAE_OK) return rc_; } while (0)

namespace {

inline int r4(int x) { return (x + 3) & ~3; }

struct Ten {
    float* p = nullptr; size_t off = 0; int64_t rows = 0, cols = 0, ld = 0;
    size_t floats() const { return (size_t)rows * ld; }
};

enum { P_W1T = 0, P_B1, P_W2, P_W3, P_V1, P_V2, P_V3, P_D1, P_D2, P_D3, NP };
enum { O_ENC = 0, O_DEC = 1, O_GEN = 2, O_DISC = 3 };

}  // namespace

// Every switch of the library, in ONE place (r6; VERDICT r5: 47 getenv calls, many latched in function-local statics, A/B
// switches of experiments long decided).  A handle reads them ONCE, in aae_create / aae_arena_bytes: a value set with
// aae_set_option(name, value) wins over the environment variable AAE_<NAME>; nothing is read after that and nothing is latched
// per process, so a test may create two handles with different switches.  What is left forces a PATH (the parity suites replay
// every fixture on each of them) or is a diagnostic of a debug run; the tuning knobs and the on/off switches of r1-r5's
// experiments are gone with their decisions taken (HISTORY.md has the numbers).
struct aae_options {
    // paths (tests)
    bool no_chain;          // NO_CHAIN: one GEMM launch per hidden layer instead of the layer-chain programs
    bool chain16;           // CHAIN16: the 16-row chain kernel on the fp32 pipe for every program (no 4-row / wide-batch kernel)
    bool split_any;         // SPLIT_ANY: the output layer as critical + deferred launch at every size
    bool blocked_any;       // BLOCKED_ANY: the row-blocked output layer at every size
    bool early_any;         // EARLY_ANY: the next batch's item list started from the step before, at every batch size
    bool no_late_join;      // NO_LATE_JOIN: a step waits for the deferred launch of the step before at its opening
    bool no_item_count;     // NO_ITEM_COUNT: first-layer item workgroups sized by the bound, not by the device's last count
    bool w1_serial;         // W1_SERIAL: data-parallel import of the peers' first-layer rows one launch per peer
    bool no_rank_fused;     // NO_RANK_FUSED: predict -> rank as two kernels over the [rows, N] score matrix
    bool chain_kslices;     // CHAIN_KSLICES: the 4-row chain kernel's linear ops in the k-slice form at every batch size (bf16 mode takes the
                            // column-owner form for batches of one fused launch: chain4.h)
    int x16_rows;           // X16_ROWS: programs of at least this many rows run on the wide-batch chain kernel (default 1024)
    int dw_ksplit_rows;     // DW_KSPLIT_ROWS: weight-gradient tiles take the k-split form from this many rows on (default 256; -1: unset)
    // diagnostics (debug runs: tools/debug/*)
    char dec_ts[8];         // DEC_TS: in-kernel timeline of the output layer ("1", "x3", "obk")
    bool chain_ts;          // CHAIN_TS: per-op timeline of a chain program's workgroup 0
    int dw_ts;              // DW_TS: per-workgroup clocks of the weight-gradient launches from this launch on (-1: off)
    int dec_skip, chain_skip, rank_skip;     // *_SKIP: timing-only ablation masks (results are wrong by design)
};
const char* option_value(const char* name);      // (aae_abi.hip: aae_set_option's table, then the environment)
inline void read_options(aae_options& o) {
    auto on = [](const char* n) { return option_value(n) != nullptr; };
    auto num = [](const char* n, int dflt) { const char* e = option_value(n); return e ? atoi(e) : dflt; };
    o.no_chain = on("NO_CHAIN"); o.chain16 = on("CHAIN16"); o.split_any = on("SPLIT_ANY"); o.blocked_any = on("BLOCKED_ANY");
    o.early_any = on("EARLY_ANY"); o.no_late_join = on("NO_LATE_JOIN"); o.no_item_count = on("NO_ITEM_COUNT");
    o.w1_serial = on("W1_SERIAL"); o.no_rank_fused = on("NO_RANK_FUSED"); o.chain_kslices = on("CHAIN_KSLICES");
    o.x16_rows = num("X16_ROWS", 1024); o.dw_ksplit_rows = num("DW_KSPLIT_ROWS", -1);
    const char* ts = option_value("DEC_TS");
    snprintf(o.dec_ts, sizeof(o.dec_ts), "%s", ts ? ts : "");
    o.chain_ts = on("CHAIN_TS"); o.dw_ts = num("DW_TS", -1);
    o.dec_skip = num("DEC_SKIP", 0); o.chain_skip = num("CHAIN_SKIP", 0); o.rank_skip = num("RANK_SKIP", 0);
}

struct aae_model {
    aae_config cfg;
    aae_options opt;         // read once, in aae_create
    char* base; size_t bytes;
    int N, h, c, cp, R, R2;
    int ldh, ldw1, ldc, ldz, ldn;
    bool alpha_mode;
    float grad_scale;
    int rng_row0, rng_global;   // device rng: this rank's rows are [rng_row0, rng_row0 + rows) of a global batch of rng_global (0: local)
    // parameters, two Adam-state sets (index 0: the owning optimiser, 1: gen_optim for enc),
    // gradients (all in export mode, gW1T always)
    Ten P[NP], M[2][NP], V[2][NP], Gr[NP];
    // transposed copies [in + 1][out] of the hidden layers' augmented weights: chain.h's dX ops read rows [0, in) with
    // the forward layers' access pattern, chain4.h's forward layers read all of it (n contiguous).  Kept in step by the fused / grouped optimiser kernels;
    // pt_ok[pid] = false after any other writer (ensure_pt() re-derives the copy before its next use)
    unsigned short* FXi[NP]; unsigned short* DXi[NP]; unsigned short* FXBi[NP];   // the split (three bf16 planes) copies for chain16x3.h, or NULL
    bool x16_ok; int x16_rows;                   // wide-batch chain kernel usable / from how many rows of a program on
    Ten PT[NP]; Ten D4[NP]; bool pt_ok[NP];      // PT = the F4 copy, D4 = the dX copy (device_common.h W4Copies)
    // activations
    Ten a1, eh1, eh2, zc, dh1, dh2, G, slabs, gb0, gb1, gb2, gb3, gzc, ga3, zin, xh1, xh2, dout, zsave, da2;
    Ten ga1x;                // export mode: see aae_create
    bool only_output_layer;  // aae_output_layer_step: stop after the output layer, dL/d(dh2) summed into da2
    const float* doc_l1;     // aae_set_doc_l1: L1 norms of the complete documents (a handle that holds an item slice of them)
    bool ext_first;          // aae_set_first_layer_external: AAE_T_ACT_A1 comes from the caller, dL/d(a1) goes back to it
    bool own_first;          // ... but the layer's rows of THIS handle's items live here and are trained here (aae_shard_step: the
                             // pre-activations come all-reduced over the item slices, the per-item update rides in the weight-gradient launch)
    bool ae_only;            // plain AutoEncoder (reference aae.py:221-458): no disc_step / gen_step
    bool vae;                // VAE (reference vae.py:47-266): P_W3 = [fc21; fc22] (2c rows), no V2/W2, KL term
    bool bf16;               // cfg.dtype = 1: bf16 matrix-core inputs for the GEMM-shaped products (fp32 accumulate / master / Adam)
    bool vae_bwd;            // aae_vae_step is running: aae_ae_decode_backward continues with the VAE's backward
    bool vae_cut;            // ... cut at the condition boundary (aae_vae_encode / _decode_backward / _encoder_backward)
    Ten mulv, gmulv, veps;   // VAE: [mu | logvar], its gradient, eps of the step
    bool use_chain;          // row-blocked layer chains (chain.h) instead of one GEMM launch per layer
    bool use_chain4;         // ... with 4 rows per workgroup (chain4.h) where a program allows it
    bool act_nm;             // cfg.activation is one of the r6 classes (AAE_ACT_SOFTPLUS ..): its layer programs run on chain_kernel<.., true> only
    bool dec_hidden_done;    // the ae forward already ran the decoder's hidden layers (fused aae_step)
    bool fuse_enc_bwd;       // aae_step: run the encoder backward in the decoder-backward program
    bool enc_bwd_done;
    int max_slabs;
    float* bce_partials; int bce_partials_cap;
    float* fix_partials;
    float* rscale;           // [R] 1/L1 of the rows of the running batch
    bool w1_merged;          // the first layer's bias update of this phase rode in the grouped dW launch
    bool w1_items_merged;    // ... and so did its row-sparse weight gradient + optimiser (w1_update.h)
    bool buckets_valid;      // the per-tile entry buckets (tstart/teb/ten/tev) describe the running batch
    // lazy Adam on W1T (kernels.h): per-row sync step, unique-row scratch, per-step scalar table
    bool lazy;
    // fused decoder output layer (dec_fused.h): tile-bucketed batch entries, eligibility
    int n_cu; bool fused_ok; int fused_nb; bool force_unfused;
    int* tcount; int* tstart; int* teb; int* ten; float* tev;
    // second set of the bucket arrays: every build goes to the set the previous batch is NOT in, so a build on the side
    // stream (aae_first_layer_forward) never races the previous step's last reader on the caller's stream (the first
    // layer's gen_optim update, w1_update.h)
    int* tstart2; int* teb2; int* ten2; float* tev2;
    int* tsync; int* mark; int* ulist; int* ucount; int* stamp; LazyTab* tab;
    int* hot_list = nullptr; int* hot_count = nullptr; int hot_cap = 0, hot_flip = 0;      // (w1_update.h, the wave form's head items)
    bool w1_hot_pending = false; int w1_hot_set = 0, w1_hot_which = 0; const float* w1_hot_ga1 = nullptr;
    int* pslot; int* ptag;   // data parallel, peers > 1: [N][peers] slot of an item's row in each peer's packet / its stamp
    int chunks;              // grid.y of the per-entry kernels for the running batch
    float* losses;
    float* adv_terms;
    OptScalars* sc;          // [4]
    long long* step_ctr;
    // state of the running step
    BatchView bv; bool have_batch; int rows; int phase;
    aae_rng_inject inj;      // randomness of the running step (inject mode)
    // optional per-kernel timing (hipEvent pairs on the launch stream)
    bool prof_on; unsigned prof_mask;   // bit k: time kernel id k
    std::vector<std::pair<hipEvent_t, hipEvent_t>>* prof_ev;   // [AAE_K_N]
    size_t prof_used[AAE_K_N];
    // split form of the fused decoder output layer (dec_fused.h, kDecCrit / kDecOpt): the optimiser launch of a step
    // runs on `side` behind the rest of the step; ev_crit = the critical launch is done (the side stream waits for it),
    // ev_opt = the optimiser launch is done (join_deferred() makes a caller's stream wait for it)
    bool split_ok; int split_wgs; bool opt_pending;
    hipStream_t side; hipEvent_t ev_crit, ev_opt;
    bool bf16_one = false;   // ... on their one-term instantiations (AAE_BF16_ONE=1; default: the run-time form, five zero products - see aae_create)
    bool bf16_x3 = false;    // bf16 mode with the output layer on the dec_crit_x3.h kernels (operands rounded to bf16: DecFusedArgs::one_term)
    float* Gt;               // [ntiles][rows][32] dL/dlogits of the running step (aliases the [R][N] scratch G)
    Ten Xn; const float* noise_next; int64_t noise_ld; bool dense_step;   // cfg.dense_noise: dense noisy encoder input (DenoisingAutoEncoder corrupt='gauss')
    bool bucket_wide_ok = false;   // tile_bucket_wide_kernel may take its LDS
    bool split_any = false;        // AAE_SPLIT_ANY at creation: the split form of the output layer at any size (tests: small fixtures through the critical / deferred kernels)
    float* dp_scratch = nullptr; size_t dp_scratch_floats = 0;   // aae_dp_step: the ranks' gathered packets (hipMalloc, owned by the handle)
    bool x3_gemm = false;          // gemm_f32.h gemm_x3_kernel: the streaming GEMMs (batches beyond the fused output layer, predict) likewise
    bool x3_ok = false;            // dec_crit_x3.h: the critical launch's fp32 products on the bf16 matrix cores (3-term split)
    bool w1_big_lds = false;       // w1_item_update_kernel may take more than 64 KB of LDS (batches beyond ~7 k rows)
    bool blocked_any = false;      // AAE_BLOCKED_ANY at creation: the row-blocked output layer at any size (tests)
    Ten dh2f;                      // dec_opt_blocks_x3_kernel: the step's dh2 as split matrix-core fragments (dh2_frag_kernel)
    bool blocked_ok; Ten Gacc;   // cfg.blocked_output: batches beyond 112 rows as row-blocked launches of the split form; dV3 partial
    // aae_prefetch_batch: the NEXT step's unique-item list and deferred-Adam catch-up, built on `side` while this step
    // runs, in the second list set (mark2 / ulist2 / ucount2 / stamp2; a step that consumes it swaps the sets)
    int* mark2; int* ulist2; int* ucount2; int* stamp2;
    // the distinct-item count of a recent step, written by a workgroup of that step's weight-gradient launch into host memory the
    // device can reach: the NEXT launches size their first-layer item workgroups by it (abi_chains.h).  Never waited for.
    int* cnt_host; int* cnt_host_dev;
    int dw_ksplit_rows;            // (set by aae_create: 256) weight-gradient jobs of at least this many rows: the k-split form of a tile (0: never; AAE_DW_KSPLIT_ROWS)
    aae_batch pf_batch, pf_built_batch; bool pf_armed; bool pf_built; long long pf_step; long long hstep;
    bool pf_pending; hipEvent_t ev_head, ev_pf;
    bool pf_bumped = false;  // the running step's gather bumped the next batch's stamp (launch_prefetch skips its own launch)
    bool pf_after_opt = false;                             // the pending prefetch was enqueued behind the pending deferred launch
    hipEvent_t ev_bk = nullptr; bool bk_pending = false;   // the tile buckets of the running batch, built on the side stream (aae_first_layer_forward)
    bool last_out_split = false;                           // the last output-layer pass ran as critical + deferred launch(es)
    // Late join: the deferred launch of step t may run into step t+1's forward pass when it reads private copies of what
    // that pass rewrites (dh2s / sc_snap, written by step t's critical launch): the wait for it then sits in front of step
    // t+1's output layer (aae_ae_decode_backward) instead of at the step's opening.  late_ok = the pending launch is one.
    Ten dh2s; OptScalars* sc_snap = nullptr;
    bool late_enabled = false, late_ok = false;
    // Early prefetch (abi_chains.h: launch_prefetch): the side stream's mark rides on the completion of the step's LAST launch
    // (a signal in front of the next step's wait costs ~2 us, on the opening gather in front of a kernel 5-6) and the next
    // batch's list + catch-up start beside the opening gather.  end_marked: the step before carried the mark; spec_tab_ok: the
    // optimiser table's entry of the coming step was written a step early and no host call has touched the scalars since
    hipEvent_t ev_end = nullptr; bool end_marked = false, spec_tab_ok = false, pf_this_step = false, early_enabled = false, early_any = false;
    long long flushed_hstep = -1;                          // hstep at the last whole-matrix deferred-Adam flush of a rank call (abi_rank.h)
    bool rank_ok = false;                                  // rank_x3.h: predict -> rank fused (aae_predict_topk / aae_decode_topk), abi_rank.h
    bool side_ordered = false;                             // ... or put its dV3 GEMM there: the side stream is in order behind that step's output layer
};

namespace {

struct Arena {
    char* base; size_t off = 0; bool dry;
    float* take(size_t nfloats, size_t* off_out) {
        off = (off + 255) & ~(size_t)255;
        size_t o = off; off += nfloats * sizeof(float);
        if (off_out) *off_out = o;
        return dry ? nullptr : reinterpret_cast<float*>(base + o);
    }
    // pad_rows: extra rows behind the tensor that stay zero for the life of the arena (the layer-chain
    // kernel reads whole 4-row k-steps of a weight matrix without clamping the row index)
    Ten mat(int64_t rows, int64_t cols, int64_t ld, int64_t pad_rows = 0) {
        Ten t; t.rows = rows; t.cols = cols; t.ld = ld;
        t.p = take((size_t)(rows + pad_rows) * ld, &t.off);
        return t;
    }
};

// row-blocked fused output layer: at most kMaxRowBlocks launches of at most kRowBlock rows each
constexpr int kRowBlock = 104, kMaxRowBlocks = 16;

// layer widths the fused decoder output-layer kernel (dec_fused.h) is instantiated for
inline bool fused_width_ok(int h, int ldh) {
    return (h + 1 + 15) / 16 <= 13 && ldh <= 256 && (ldh % 4) == 0 && ldh <= kSD - 2;
}

int validate(const aae_config* c) {
    if (!c) return fail(AAE_EINVAL, "cfg is NULL");
    if (c->abi_version != AAE_ABI_VERSION) return fail(AAE_EINVAL, "abi_version mismatch");
    if (c->n_items < 1 || c->n_hidden < 1 || c->n_code < 1 || c->cond_inc < 0)
        return fail(AAE_EINVAL, "n_items/n_hidden/n_code must be positive");
    if (c->n_hidden > 4096) return fail(AAE_EINVAL, "n_hidden > 4096 not supported");
    if (c->max_batch < 1 || c->max_nnz < 1) return fail(AAE_EINVAL, "max_batch/max_nnz must be positive");
    if (c->max_batch > 16384) return fail(AAE_EINVAL, "max_batch > 16384 not supported");
    if (c->activation < 0 || c->activation >= AAE_ACT_COUNT) return fail(AAE_EINVAL, "unknown activation");
    if (c->enc_final < 0 || c->enc_final > AAE_FINAL_SIGMOID) return fail(AAE_EINVAL, "unknown enc_final");
    if (c->optimizer != AAE_OPT_ADAM && c->optimizer != AAE_OPT_SGD) return fail(AAE_EINVAL, "unknown optimizer");
    if (c->rng_mode != AAE_RNG_INJECT && c->rng_mode != AAE_RNG_DEVICE) return fail(AAE_EINVAL, "unknown rng_mode");
    if (c->grad_mode != AAE_GRAD_FUSED && c->grad_mode != AAE_GRAD_EXPORT) return fail(AAE_EINVAL, "unknown grad_mode");
    if (!(c->dropout1 >= 0.f && c->dropout1 < 1.f && c->dropout2 >= 0.f && c->dropout2 < 1.f))
        return fail(AAE_EINVAL, "dropout must be in [0,1)");
    if (c->reserved[0] || c->reserved[1]) return fail(AAE_EINVAL, "reserved fields must be zero");
    if (c->dense_noise != 0 && c->dense_noise != 1) return fail(AAE_EINVAL, "dense_noise must be 0 or 1 (dense noisy encoder input)");
    if (c->dense_noise == 1 && (c->model_kind != 1 || c->grad_mode != AAE_GRAD_FUSED || c->dtype != 0))
        return fail(AAE_EINVAL, "dense_noise = 1 (dense noisy encoder input) needs the plain autoencoder (model_kind = 1), fp32, fused optimiser");
    if (c->blocked_output != 0 && c->blocked_output != 1) return fail(AAE_EINVAL, "blocked_output must be 0 or 1 (row-blocked fused output layer)");
    if (c->dtype != 0 && c->dtype != 1) return fail(AAE_EINVAL, "dtype must be 0 (fp32) or 1 (bf16 matrix-core inputs)");
    if (c->dtype == 1 && c->model_kind == 3) return fail(AAE_EINVAL, "bf16 arithmetic is not available in VAE mode");
    if (c->model_kind < 0 || c->model_kind > 3 || c->model_kind == 2)
        return fail(AAE_EINVAL, "model_kind must be 0 (AAE), 1 (plain autoencoder) or 3 (VAE)");
    if (c->model_kind == 3 && (c->n_hidden + 1 > 208 || c->n_code + c->cond_inc + 1 > 208 || 2 * c->n_code > 208))
        return fail(AAE_EINVAL, "VAE mode needs n_hidden <= 207, n_code + cond_inc <= 207, 2 * n_code <= 208");
    if (c->dp_world < 0 || c->dp_world > 64) return fail(AAE_EINVAL, "dp_world (data-parallel world size) out of range");
    if (c->unfused_decoder != 0 && c->unfused_decoder != 1) return fail(AAE_EINVAL, "unfused_decoder must be 0 or 1");
    return AAE_OK;
}

// Programs of at least aae_options::x16_rows rows (default 1024) run on the 16-row chain kernel (chain16x3.h).
// A layer op is bound by getting its weights through ONE CU's vector-memory path (64 B/clk): 160 KB of fp32 per 201 x 200 layer on
// the 4-row kernel, 280 KB of split terms on the 16-row one - its workgroup takes ~2x the time for 4x the rows, so it pays where
// the 4-row launch needs two rounds and more of the CUs it gets (r5, same-box A/B, from 256 | 1024 rows | never: C4 0.397 | 0.351 |
// 0.365 ms/step, C3 at batch 512 0.665 | 0.657 | 0.673, one rank's step at world 8 0.340 | 0.289 | 0.298).

// lays the model out; with dry=true only measures
size_t layout(aae_model* m, char* base, bool dry) {
    const aae_config& c = m->cfg;
    m->N = c.n_items; m->h = c.n_hidden; m->c = c.n_code; m->cp = c.n_code + c.cond_inc;
    m->R = c.max_batch; m->R2 = 2 * c.max_batch;
    m->ldh = r4(m->h + 1); m->ldw1 = r4(m->h); m->ldc = r4(m->cp + 1); m->ldz = r4(m->c + 1); m->ldn = r4(m->N);
    Arena a{base, 0, dry};
    const int N = m->N, h = m->h, cc = m->c, cp = m->cp;
    m->P[P_W1T] = a.mat(N, h, m->ldw1);
    m->P[P_B1] = a.mat(1, h, m->ldw1);
    m->P[P_W2] = a.mat(h, h + 1, m->ldh, 16);
    m->P[P_W3] = a.mat(c.model_kind == 3 ? 2 * cc : cc, h + 1, m->ldh, 16);   // VAE: [fc21; fc22]
    m->P[P_V1] = a.mat(h, cp + 1, m->ldc, 16);
    m->P[P_V2] = a.mat(h, h + 1, m->ldh, 16);
    m->P[P_V3] = a.mat(N, h + 1, m->ldh, 2 * kTI);  // (+ two tiles of padding rows: dec_fused_bf16.h reads whole tiles unclamped and parks the stores of lanes without a cell there)
    m->P[P_D1] = a.mat(h, cc + 1, m->ldz, 16);
    m->P[P_D2] = a.mat(h, h + 1, m->ldh, 16);
    m->P[P_D3] = a.mat(1, h + 1, m->ldh, 16);
    for (int i = 0; i < NP; ++i) {
        m->M[0][i] = a.mat(m->P[i].rows, m->P[i].cols, m->P[i].ld, i == P_V3 ? 2 * kTI : 0);
        m->V[0][i] = a.mat(m->P[i].rows, m->P[i].cols, m->P[i].ld, i == P_V3 ? 2 * kTI : 0);
    }
    for (int i = P_W1T; i <= P_W3; ++i) {
        m->M[1][i] = a.mat(m->P[i].rows, m->P[i].cols, m->P[i].ld);
        m->V[1][i] = a.mat(m->P[i].rows, m->P[i].cols, m->P[i].ld);
    }
    m->Gr[P_W1T] = a.mat(N, h, m->ldw1);
    // export mode: dL/d(a1) of a replica with an external first layer, RIGHT-aligned in a buffer that ends where the small
    // layers' gradient span begins - [its rows | b1, W2, W3, V1, V2 gradients] is then one contiguous packet for the
    // all-gather of the both-sharded scheme (no packing launch)
    if (c.grad_mode == AAE_GRAD_EXPORT) m->ga1x = a.mat(m->R, h + 1, m->ldh);
    if (c.grad_mode == AAE_GRAD_EXPORT)
        for (int i = P_B1; i < NP; ++i) m->Gr[i] = a.mat(m->P[i].rows, m->P[i].cols, m->P[i].ld, i == P_V3 ? 2 * kTI : 0);
    for (int i = 0; i < NP; ++i) { m->PT[i] = Ten(); m->D4[i] = Ten(); m->pt_ok[i] = false; }
    const bool chain_copies = h + 1 <= 208 && m->c + 1 <= 208 && cp + 1 <= 2 * 208 && c.model_kind != 3;
    if (chain_copies)      // layer-chain models (not the VAE's programs); a decoder input beyond 208 columns runs in two k-parts (abi_chains.h)
        for (int pid : {P_W2, P_W3, P_V1, P_V2, P_D1, P_D2})
        {
            const int64_t M = m->P[pid].rows, Nc = m->P[pid].cols;
            // (+ kW4Pad zero rows behind each: a wave of the 4-row chain kernel reads whole runs of k-chunks unclamped,
            //  chain4.h; the A operand is zero there)
            m->PT[pid] = a.mat((Nc + 3) / 4, 4 * M, 4 * M, kW4Pad);      // F4 [(in + 1 + 3) / 4][out][4]: k = input column (the bias is k = in)
            m->D4[pid] = a.mat((M + 3) / 4, 4 * Nc, 4 * Nc, kW4Pad);     // D4 [(out + 3) / 4][in + 1][4]: k = output row
        }
    // the split weight copies of the wide-batch chain kernel (chain16x3.h): only for models that can see such a batch
    for (int i = 0; i < NP; ++i) m->FXi[i] = m->DXi[i] = m->FXBi[i] = nullptr;
    if (chain_copies && m->R2 >= m->opt.x16_rows)
        for (int pid : {P_W2, P_W3, P_V1, P_V2, P_D1, P_D2}) {
            const int64_t M = m->P[pid].rows, Nc = m->P[pid].cols, Mp = (M + 15) & ~15, Np = (Nc + 15) & ~15;
            m->FXi[pid] = reinterpret_cast<unsigned short*>(a.take((size_t)(((Nc + 31) / 32) * 3 * Mp * 32 + 1) / 2, nullptr));
            m->DXi[pid] = reinterpret_cast<unsigned short*>(a.take((size_t)(((M + 31) / 32) * 3 * Np * 32 + 1) / 2, nullptr));
            if (Nc > kCWide) m->FXBi[pid] = reinterpret_cast<unsigned short*>(a.take((size_t)(((Nc - kCWide + 31) / 32) * 3 * Mp * 32 + 1) / 2, nullptr));
        }
    const int R = m->R, R2 = m->R2;
    m->a1 = a.mat(R, h, m->ldh);   m->eh1 = a.mat(R, h + 1, m->ldh);  m->eh2 = a.mat(R, h + 1, m->ldh);
    m->zc = a.mat(R, cp + 1, m->ldc);
    m->dh1 = a.mat(R, h + 1, m->ldh); m->dh2 = a.mat(R, h + 1, m->ldh);
    m->G = a.mat(R, N, m->ldn, (32 * (int64_t)R + m->ldn - 1) / m->ldn + 1 + (c.blocked_output ? kMaxRowBlocks : 0));   // (+ room for the tile-major form [ceil(N/32)][R][32] of dec_fused.h's split launches)
    // split-K slabs for dA2 = G * V3: enough slices to put >= ~512 workgroups on the chip
    {
        int tiles = ((R + 63) / 64) * ((h + 63) / 64);
        m->max_slabs = std::max(1, std::min(128, 2048 / tiles));
        // the fused decoder kernel writes one dA2 slab per workgroup (<= 304 CUs assumed for sizing) of at most
        // 16 * kMB rows: a model with a larger max_batch still takes it for its short (tail) batches
        int64_t slab_rows = (int64_t)m->max_slabs * R;
        if (fused_width_ok(h, m->ldh)) slab_rows = std::max(slab_rows, (int64_t)(304 + 16) * std::min(R, 16 * kMB));
        // row-blocked form: every workgroup's slab spans the whole batch (each launch fills its rows)
        if (fused_width_ok(h, m->ldh) && c.blocked_output && R <= kMaxRowBlocks * kRowBlock) slab_rows = std::max(slab_rows, (int64_t)(304 + 16) * R);
        m->slabs = a.mat(slab_rows, h, m->ldh);
    }
    m->gb0 = a.mat(R2, h + 1, m->ldh); m->gb1 = a.mat(R2, h + 1, m->ldh);
    m->gb2 = a.mat(R2, h + 1, m->ldh); m->gb3 = a.mat(R2, h + 1, m->ldh);
    m->gzc = a.mat(R, cp + 1, m->ldc);
    m->ga3 = a.mat(R2, cc + 1, m->ldz);
    m->zin = a.mat(R2, cc + 1, m->ldz);
    m->xh1 = a.mat(R2, h + 1, m->ldh); m->xh2 = a.mat(R2, h + 1, m->ldh);
    m->dout = a.mat(R2, 1, 4);
    m->zsave = a.mat(R, cc, m->ldz);
    m->da2 = a.mat(R, h + 1, m->ldh);
    m->Xn = Ten();
    if (c.dense_noise == 1) m->Xn = a.mat(R, N, m->ldn);
    m->Gacc = Ten();
    if (c.blocked_output && c.grad_mode == AAE_GRAD_FUSED && R > 16 * kMB) m->Gacc = a.mat(N, h + 1, m->ldh, 2 * kTI);
    m->dh2f = Ten();
    if (c.blocked_output && c.grad_mode == AAE_GRAD_FUSED && R > 16 * kMB && R <= kMaxRowBlocks * kRowBlock)
        m->dh2f = a.mat((int64_t)((R + kXCH - 1) / kXCH) * 13 * (kXCH / 32) * 3, 256, 256);      // [chunk][column block][k-step][term] x 1 KB
    if (c.model_kind == 3) {
        m->mulv = a.mat(R, 2 * cc, r4(2 * cc)); m->gmulv = a.mat(R, 2 * cc, r4(2 * cc)); m->veps = a.mat(R, cc, r4(cc));
    }
    m->bce_partials_cap = std::max(512 * (c.blocked_output ? kMaxRowBlocks : 1), ((N + 31) / 32) * ((R + 31) / 32));
    m->bce_partials = a.take(m->bce_partials_cap, nullptr);
    m->fix_partials = a.take((size_t)R * 64, nullptr);
    m->rscale = a.take(R, nullptr);
    m->tsync = reinterpret_cast<int*>(a.take(N, nullptr));
    m->mark = reinterpret_cast<int*>(a.take(N, nullptr));
    m->ulist = reinterpret_cast<int*>(a.take((size_t)c.max_nnz * (size_t)std::max(1, c.dp_world), nullptr));
    m->pslot = m->ptag = nullptr;
    if (c.grad_mode == AAE_GRAD_EXPORT && c.dp_world > 1) {
        m->pslot = reinterpret_cast<int*>(a.take((size_t)N * c.dp_world, nullptr));
        m->ptag = reinterpret_cast<int*>(a.take((size_t)N * c.dp_world, nullptr));
    }
    m->ucount = reinterpret_cast<int*>(a.take(4, nullptr));
    m->stamp = m->ucount ? m->ucount + 1 : nullptr;
    // wide batches: the head items of the first layer's one-wave-per-item update (w1_update.h): two alternating lists + counters
    m->hot_cap = (int)std::max<int64_t>(256, c.max_nnz / kW1WaveRows + 64);
    m->hot_list = reinterpret_cast<int*>(a.take((size_t)2 * m->hot_cap, nullptr));
    m->hot_count = reinterpret_cast<int*>(a.take(4, nullptr));
    m->hot_flip = 0;
    m->mark2 = m->ulist2 = m->ucount2 = m->stamp2 = nullptr;
    if (c.grad_mode == AAE_GRAD_FUSED) {      // second list set for aae_prefetch_batch (single-process training only)
        m->mark2 = reinterpret_cast<int*>(a.take(N, nullptr));
        m->ulist2 = reinterpret_cast<int*>(a.take((size_t)c.max_nnz, nullptr));
        m->ucount2 = reinterpret_cast<int*>(a.take(4, nullptr));
        m->stamp2 = m->ucount2 ? m->ucount2 + 1 : nullptr;
    }
    m->tab = reinterpret_cast<LazyTab*>(a.take((size_t)kLazyTabCap * 4, nullptr));
    {
        const size_t nt = (size_t)(N + kTI - 1) / kTI + 1;
        m->tcount = reinterpret_cast<int*>(a.take(nt, nullptr));
        m->tstart = reinterpret_cast<int*>(a.take(nt, nullptr));
        m->teb = reinterpret_cast<int*>(a.take((size_t)c.max_nnz, nullptr));
        m->ten = reinterpret_cast<int*>(a.take((size_t)c.max_nnz, nullptr));
        m->tev = a.take((size_t)c.max_nnz, nullptr);
        m->tstart2 = reinterpret_cast<int*>(a.take(nt, nullptr));
        m->teb2 = reinterpret_cast<int*>(a.take((size_t)c.max_nnz, nullptr));
        m->ten2 = reinterpret_cast<int*>(a.take((size_t)c.max_nnz, nullptr));
        m->tev2 = a.take((size_t)c.max_nnz, nullptr);
    }
    m->losses = a.take(4, nullptr);
    m->adv_terms = a.take((size_t)R2, nullptr);     // per-row adversarial loss terms of the running chain program (chain.h: loss_terms)
    m->sc = reinterpret_cast<OptScalars*>(a.take(4 * sizeof(OptScalars) / sizeof(float), nullptr));
    m->step_ctr = reinterpret_cast<long long*>(a.take(2, nullptr));
    m->dh2s = Ten(); m->sc_snap = nullptr;
    if (c.grad_mode == AAE_GRAD_FUSED) {        // the late join's operand copies (one row block: <= 16 kMB rows)
        m->dh2s = a.mat(std::min(R, 16 * kMB), h + 1, m->ldh);
        m->sc_snap = reinterpret_cast<OptScalars*>(a.take((sizeof(OptScalars) + 3) / sizeof(float), nullptr));
    }
    return (a.off + 255) & ~(size_t)255;
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
// one attribute on a kernel and on its moving-window instantiation (layers beyond 2^31 bytes, dec_fused.h X3WindowT)
inline hipError_t aae_attr2(const void* k, const void* k_win, hipFuncAttribute attr, int value) {
    const hipError_t e = hipFuncSetAttribute(k, attr, value);
    return e != hipSuccess ? e : hipFuncSetAttribute(k_win, attr, value);
}

// scoped hipEvent pair around one kernel launch when profiling is enabled
struct ProfScope {
    aae_model* m; int k; hipStream_t s; bool on;
    ProfScope(aae_model* m_, int k_, hipStream_t s_) : m(m_), k(k_), s(s_), on(m_->prof_on && ((m_->prof_mask >> k_) & 1)) {
        if (!on) return;
        auto& v = m->prof_ev[k];
        if (m->prof_used[k] == v.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            v.emplace_back(a, b);
        }
        (void)hipEventRecord(v[m->prof_used[k]].first, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(m->prof_ev[k][m->prof_used[k]].second, s);
        m->prof_used[k]++;
    }
};

// A timing pair for a launch through hipExtLaunchKernelGGL (the events ride on the kernel's own start / completion
// signals: no marker packets on the stream); false when kernel id k is not being timed
bool prof_pair(aae_model* m, int k, hipEvent_t* a, hipEvent_t* b) {
    if (!(m->prof_on && ((m->prof_mask >> k) & 1))) return false;
    auto& v = m->prof_ev[k];
    if (m->prof_used[k] == v.size()) {
        hipEvent_t x, y;
        if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) return false;
        v.emplace_back(x, y);
    }
    *a = v[m->prof_used[k]].first; *b = v[m->prof_used[k]].second;
    m->prof_used[k]++;
    return true;
}

// bf16 mode whose output layer runs dec_fused_bf16.h's own kernels (not the rounded-operand form of the dec_crit_x3.h ones)
static inline bool out_bf16(const aae_model* m) { return m->bf16 && !m->bf16_x3; }

// The previous step's deferred optimiser launch (dec_fused.h kDecOpt on m->side) writes DEC_V3 and its moments and
// reads dh2 / the G scratch / the decoder's step scalars: everything that touches those waits for it here.
int join_deferred(aae_model* m, hipStream_t s) {
    // (one side stream, in order: the later record covers the earlier - the prefetch is enqueued in front of the deferred
    //  launch by aae_step's path and behind it by an item slice's, so both marks are waited for when both are pending)
    if (m->opt_pending) HIPCHK(hipStreamWaitEvent(s, m->ev_opt, 0));
    if (m->pf_pending && (!m->opt_pending || m->pf_after_opt)) HIPCHK(hipStreamWaitEvent(s, m->ev_pf, 0));
    m->opt_pending = m->pf_pending = m->late_ok = false;
    return AAE_OK;
}

// ... at the opening of a training step (ae_encode_impl): a pending deferred launch that reads its private operand copies
// (late_ok) keeps running - only the prefetch enqueued in front of it is waited for; join_output_layer() below is its join
int join_step_open(aae_model* m, hipStream_t s) {
    if (!(m->opt_pending && m->late_ok && !m->pf_after_opt)) return join_deferred(m, s);
    if (m->pf_pending) { HIPCHK(hipStreamWaitEvent(s, m->ev_pf, 0)); m->pf_pending = false; }
    return AAE_OK;
}

// ... in front of the output layer (it reads and writes what the deferred launch of the step before does: dec.lin3 and
// its Adam moments, the stored dL/dlogits tiles).  A prefetch started by THIS step stays pending: the side stream runs
// it behind the launch waited for here.
int join_output_layer(aae_model* m, hipStream_t s) {
    if (!m->opt_pending) return AAE_OK;
    HIPCHK(hipStreamWaitEvent(s, m->ev_opt, 0));
    m->opt_pending = m->late_ok = false;
    return AAE_OK;
}
// behind work enqueued on the side stream: what join_deferred waits for
int side_done(aae_model* m, hipEvent_t ev) {
    HIPCHK(hipEventRecord(ev, m->side));
    return AAE_OK;
}
// ... for the entry points without a stream (host-synchronous state import / export)
int join_host(aae_model* m) {
    if (!m->opt_pending && !m->pf_pending) return AAE_OK;
    HIPCHK(hipStreamSynchronize(m->side));
    m->opt_pending = m->pf_pending = false;
    return AAE_OK;
}

DropSpec make_drop(const aae_model* m, int layer, bool train, const uint8_t* ma, const uint8_t* mb, int split,
                   int width, uint32_t stream_id) {
    DropSpec d; memset(&d, 0, sizeof(d));
    float p = layer == 0 ? m->cfg.dropout1 : m->cfg.dropout2;
    d.enabled = (train && p > 0.f) ? 1 : 0;
    if (!d.enabled) return d;
    d.mask_a = ma; d.mask_b = mb; d.split_row = split; d.width = width;
    d.device_rng = m->cfg.rng_mode == AAE_RNG_DEVICE;
    d.goff_a = m->rng_row0;
    d.goff_b = m->rng_row0 + (m->rng_global > 0 ? m->rng_global - split : 0);
    d.keep_threshold = (uint32_t)std::min(4294967295.0, (double)p * 4294967296.0);
    d.stream_id = stream_id;
    if (m->alpha_mode) {
        const double alpha = 1.7580993408473766;
        double a = 1.0 / sqrt((alpha * alpha * p + 1.0) * (1.0 - p));
        d.mul_keep = (float)a;
        d.add_keep = (float)(alpha * a * p);
        d.add_drop = (float)(-alpha * a) + (float)(alpha * a * p);
    } else {
        d.mul_keep = 1.0f / (1.0f - p);
        d.add_keep = 0.f; d.add_drop = 0.f;
    }
    if (!d.device_rng && !ma && !mb) d.enabled = 0;   // inject mode without masks: identity
    return d;
}

inline int grid1d(size_t n, int block = 256) { return (int)std::min<size_t>((n + block - 1) / block, 2048); }


}  // namespace
