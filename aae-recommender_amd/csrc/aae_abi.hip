// libaaerec_hip.so - C ABI (include/aaerec_hip.h) and step orchestration.
//
// One `aae_model` = the three nets + four optimisers that AdversarialAutoEncoder.fit builds
// (reference aaerec/aae.py:782-804), laid out in one caller-owned HBM arena:
//
//   parameters      item-major for the two vocabulary-sized layers (ENC_W1T [N][h], DEC_V3
//                   [N][h+1]) so that a gathered / streamed row is one contiguous 800-byte line;
//                   every other Linear "augmented" [out][in+1] (bias = last column) so that the
//                   forward GEMM adds the bias and the weight-gradient GEMM yields the bias
//                   gradient for free (activations carry a constant-1 column).
//   optimiser state exp_avg / exp_avg_sq with the layout of their parameter, two independent
//                   states for the encoder (enc_optim, gen_optim; aae.py:800,803).
//   activations     fp32 [rows][ld], ld = round4(width+1).
//
// The step is a fixed sequence of kernel launches on one stream (no host sync, no allocation),
// so a caller may capture it into a hipGraph.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/aaerec_hip.h"
#include "gemm_f32.h"
#include "kernels.h"
#include "dec_fused.h"
#include "dec_fused_bf16.h"
#include "dec_crit_x3.h"
#include "chain.h"
#include "chain4.h"
#include "cond_embed.h"
#include "w1_update.h"

using namespace aae;

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

struct aae_model {
    aae_config cfg;
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
    Ten PT[NP]; Ten D4[NP]; bool pt_ok[NP];      // PT = the F4 copy, D4 = the dX copy (device_common.h W4Copies)
    // activations
    Ten a1, eh1, eh2, zc, dh1, dh2, G, slabs, gb0, gb1, gb2, gb3, gzc, ga3, zin, xh1, xh2, dout, zsave, da2;
    Ten ga1x;                // export mode: see aae_create
    bool only_output_layer;  // aae_output_layer_step: stop after the output layer, dL/d(dh2) summed into da2
    const float* doc_l1;     // aae_set_doc_l1: L1 norms of the complete documents (a handle that holds an item slice of them)
    bool ext_first;          // aae_set_first_layer_external: AAE_T_ACT_A1 comes from the caller, dL/d(a1) goes back to it
    bool ae_only;            // plain AutoEncoder (reference aae.py:221-458): no disc_step / gen_step
    bool vae;                // VAE (reference vae.py:47-266): P_W3 = [fc21; fc22] (2c rows), no V2/W2, KL term
    bool bf16;               // cfg.dtype = 1: bf16 matrix-core inputs for the GEMM-shaped products (fp32 accumulate / master / Adam)
    bool vae_bwd;            // aae_vae_step is running: aae_ae_decode_backward continues with the VAE's backward
    bool vae_cut;            // ... cut at the condition boundary (aae_vae_encode / _decode_backward / _encoder_backward)
    Ten mulv, gmulv, veps;   // VAE: [mu | logvar], its gradient, eps of the step
    bool use_chain;          // row-blocked layer chains (chain.h) instead of one GEMM launch per layer
    bool use_chain4;         // ... with 4 rows per workgroup (chain4.h) where a program allows it
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
    int* pslot; int* ptag;   // data parallel, peers > 1: [N][peers] slot of an item's row in each peer's packet / its stamp
    int chunks;              // grid.y of the per-entry kernels for the running batch
    float* losses;
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
    aae_batch pf_batch, pf_built_batch; bool pf_armed; bool pf_built; long long pf_step; long long hstep;
    bool pf_pending; hipEvent_t ev_head, ev_pf;
    bool pf_after_opt = false;                             // the pending prefetch was enqueued behind the pending deferred launch
    hipEvent_t ev_bk = nullptr; bool bk_pending = false;   // the tile buckets of the running batch, built on the side stream (aae_first_layer_forward)
    bool last_out_split = false;                           // the last output-layer pass ran as critical + deferred launch(es)
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
    if (c->activation < 0 || c->activation > AAE_ACT_LEAKYRELU) return fail(AAE_EINVAL, "unknown activation");
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
    if (h + 1 <= 208 && cp + 1 <= 208 && c.model_kind != 3)          // layer-chain models (not the VAE's programs)
        for (int pid : {P_W2, P_W3, P_V1, P_V2, P_D1, P_D2})
        {
            const int64_t M = m->P[pid].rows, Nc = m->P[pid].cols;
            m->PT[pid] = a.mat((Nc + 3) / 4, 4 * M, 4 * M, 1);      // F4 [(in + 1 + 3) / 4][out][4]: k = input column (the bias is k = in)
            m->D4[pid] = a.mat((M + 3) / 4, 4 * Nc, 4 * Nc, 1);     // D4 [(out + 3) / 4][in + 1][4]: k = output row
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
    m->sc = reinterpret_cast<OptScalars*>(a.take(4 * sizeof(OptScalars) / sizeof(float), nullptr));
    m->step_ctr = reinterpret_cast<long long*>(a.take(2, nullptr));
    return (a.off + 255) & ~(size_t)255;
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

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

// The previous step's deferred optimiser launch (dec_fused.h kDecOpt on m->side) writes DEC_V3 and its moments and
// reads dh2 / the G scratch / the decoder's step scalars: everything that touches those waits for it here.
int join_deferred(aae_model* m, hipStream_t s) {
    // (one side stream, in order: the later record covers the earlier - the prefetch is enqueued in front of the deferred
    //  launch by aae_step's path and behind it by an item slice's, so both marks are waited for when both are pending)
    if (m->opt_pending) HIPCHK(hipStreamWaitEvent(s, m->ev_opt, 0));
    if (m->pf_pending && (!m->opt_pending || m->pf_after_opt)) HIPCHK(hipStreamWaitEvent(s, m->ev_pf, 0));
    m->opt_pending = m->pf_pending = false;
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

// ------------------------------------------------------------------------------------------
// GEMM wrappers (see gemm_f32.h for operand forms)
// ------------------------------------------------------------------------------------------
// Y[rows][out] = epi( X[rows][in+1] * Wa[out][in+1]^T )
// bf: arithmetic of the product (gemm_f32.h: kGemmF32 / kGemmBf16 - cfg.dtype, every forward and dX product of a Linear layer
// takes it - / kGemmX3: fp32 emulated on the bf16 matrix cores, the vocabulary-wide streaming GEMMs only)
static inline int gmode(const aae_model* m) { return m->bf16 ? kGemmBf16 : m->x3_gemm ? kGemmX3 : kGemmF32; }
template <class Epi>
int linear_fwd(const float* X, int ldx, int rows, const Ten& Wa, const Epi& epi, hipStream_t s, int bf = 0) {
    GemmShape g{X, Wa.p, rows, (int)Wa.rows, (int)Wa.cols, ldx, (int)Wa.ld, r4((int)Wa.cols) + 16};
    g.k_per_split = ((int)Wa.cols + 63) / 64 * 64;
    if (Wa.rows > 4096) (void)launch_gemm_mode<0, 1, true>(bf, g, epi, 1, s);   // vocabulary-wide: streaming regime
    else (void)launch_gemm_mode<0, 1, false>(bf, g, epi, 1, s);
    LAUNCHCHK("linear_fwd");
    return AAE_OK;
}
// dX[rows][n_in] = epi( Gd[rows][out] * Wa[out][0:n_in] )
template <class Epi>
int linear_dx(const float* Gd, int ldg, int rows, const Ten& Wa, int n_in, const Epi& epi, hipStream_t s, int bf = 0) {
    GemmShape g{Gd, Wa.p, rows, n_in, (int)Wa.rows, ldg, (int)Wa.ld, 0};
    g.k_per_split = ((int)Wa.rows + 63) / 64 * 64;
    (void)launch_gemm_mode<0, 0, false>(bf, g, epi, 1, s);
    LAUNCHCHK("linear_dx");
    return AAE_OK;
}
// dWa[out][in+1] = Gd[rows][out]^T * X[rows][in+1]  -> optimiser update (or gradient export)
int linear_dw(aae_model* m, const float* Gd, int ldg, int rows, const float* X, int ldx, int pid, int which,
              hipStream_t s) {
    const Ten& W = m->P[pid];
    GemmShape g{Gd, X, (int)W.rows, (int)W.cols, rows, ldg, ldx, 0};
    g.k_per_split = (rows + 63) / 64 * 64;
    const bool big = W.rows > 4096;
    // bf16 mode: of the weight gradients only the decoder output layer's is a bf16 product (the hidden layers' and the
    // sparse first layer's stay fp32: they are launch-latency, not matrix-pipe, bound)
    const int bf = pid == P_V3 ? gmode(m) : kGemmF32;
    if (m->cfg.grad_mode == AAE_GRAD_EXPORT) {
        EpiStore e; e.out = m->Gr[pid].p; e.ld = (int)W.ld;
        if (big) (void)launch_gemm_mode<1, 0, true>(bf, g, e, 1, s); else (void)launch_gemm_mode<1, 0, false>(bf, g, e, 1, s);
    } else {
        const int set = (which == O_GEN) ? 1 : 0;
        EpiAdam e; e.p = W.p; e.m = m->M[set][pid].p; e.v = m->V[set][pid].p; e.ld = (int)W.ld; e.sc = m->sc + which;
        m->pt_ok[pid] = false;
        if (big) (void)launch_gemm_mode<1, 0, true>(bf, g, e, 1, s); else (void)launch_gemm_mode<1, 0, false>(bf, g, e, 1, s);
    }
    LAUNCHCHK("linear_dw");
    return AAE_OK;
}

int set_batch(aae_model* m, const aae_batch* b) {
    if (!b || !b->indptr_dev || !b->indices_dev || !b->values_dev) return fail(AAE_EINVAL, "batch pointers are NULL");
    if (b->n_rows < 1 || b->n_rows > m->R) return fail(AAE_EINVAL, "batch n_rows outside [1, max_batch]");
    if (b->nnz_bound > m->cfg.max_nnz) return fail(AAE_EINVAL, "batch nnz_bound > max_nnz");
    m->bv.indptr = b->indptr_dev; m->bv.indices = b->indices_dev; m->bv.values = b->values_dev;
    m->bv.rows = b->rows_dev; m->bv.row_start = b->row_start; m->bv.n_rows = b->n_rows;
    m->rows = b->n_rows; m->have_batch = true; m->buckets_valid = false; m->w1_merged = false;
    {   // 16 entries per workgroup pass; unknown row bound -> 64 strided chunks
        int mr = b->max_row_nnz > 0 ? b->max_row_nnz : 1024;
        m->chunks = std::max(1, std::min(64, (mr + 15) / 16));
    }
    return AAE_OK;
}

// lazy Adam: list the distinct items of the running batch and bring their W1T rows up to date
// (through step t-1 before a training gather, through step t for predict / export)
int lazy_prepare(aae_model* m, int upto_off, bool bump, hipStream_t s) {
    if (bump) hipLaunchKernelGGL(bump_stamp_kernel, dim3(1), dim3(1), 0, s, m->stamp, m->ucount);
    int gy = std::max(1, std::min(16, m->chunks / 16 + 1));
    hipLaunchKernelGGL(uniq_items_kernel, dim3(m->rows, gy), dim3(256), 0, s, m->bv, m->mark, m->stamp, m->ulist,
                       m->ucount);
    LAUNCHCHK("uniq_items");
    if (m->cfg.optimizer == AAE_OPT_ADAM) {
        int grid = std::min(m->cfg.max_nnz, std::max(256, m->rows * 32));
        hipLaunchKernelGGL(w1_catchup_kernel, dim3(grid), dim3(256), 0, s, m->ulist, m->ucount, m->N, m->tsync,
                           m->P[P_W1T].p, m->M[0][P_W1T].p, m->V[0][P_W1T].p, m->M[1][P_W1T].p, m->V[1][P_W1T].p,
                           m->ldw1, m->h, m->tab, m->step_ctr, upto_off);
        LAUNCHCHK("w1_catchup");
    }
    return AAE_OK;
}

// lazy Adam: every row of W1T (and its four moment tensors) through the current step
int lazy_flush(aae_model* m, hipStream_t s) {
    if (!m->lazy || m->cfg.optimizer != AAE_OPT_ADAM) return AAE_OK;
    hipLaunchKernelGGL(w1_catchup_kernel, dim3(std::min(m->N, 8192)), dim3(256), 0, s, (const int*)nullptr,
                       (const int*)nullptr, m->N, m->tsync, m->P[P_W1T].p, m->M[0][P_W1T].p, m->V[0][P_W1T].p,
                       m->M[1][P_W1T].p, m->V[1][P_W1T].p, m->ldw1, m->h, m->tab, m->step_ctr, 0);
    LAUNCHCHK("w1_catchup all");
    return AAE_OK;
}

// Encoder forward (aae.py:129-146) into `z_dst` [rows][ldz_dst] (first n_code columns).
// train=false: eval mode (no dropout).  reuse_a1: skip the gather, start from m->a1.
int encoder_forward(aae_model* m, bool train, const uint8_t* mk1, const uint8_t* mk2, uint32_t sid1, uint32_t sid2,
                    bool reuse_a1, float* z_dst, int ldz_dst, hipStream_t s) {
    const int B = m->rows, h = m->h;
    DropSpec d1 = make_drop(m, 0, train, mk1, nullptr, B, h, sid1);
    DropSpec d2 = make_drop(m, 1, train, mk2, nullptr, B, h, sid2);
    if (!reuse_a1) {
        ProfScope ps(m, AAE_K_ENC_GATHER, s);
        size_t shm = (size_t)16 * r4(h) * sizeof(float);
        hipLaunchKernelGGL(enc_gather_kernel, dim3(B), dim3(1024), shm, s, m->bv, m->P[P_W1T].p, m->ldw1,
                           m->P[P_B1].p, h, m->cfg.normalize_inputs, m->a1.p, m->eh1.p, m->ldh, m->cfg.activation,
                           d1, m->cfg.seed, m->step_ctr, m->rscale, m->doc_l1, AdvanceJob{nullptr, nullptr, nullptr, nullptr, 0},
                           (long long)-1);
        LAUNCHCHK("enc_gather");
    } else {
        hipLaunchKernelGGL(drop_act_kernel, dim3(grid1d((size_t)B * h)), dim3(256), 0, s, m->a1.p, m->eh1.p, B, h,
                           m->ldh, m->cfg.activation, d1, m->cfg.seed, m->step_ctr);
        LAUNCHCHK("drop_act");
    }
    EpiDropAct e2; e2.out = m->eh2.p; e2.ld = m->ldh; e2.act = m->cfg.activation; e2.d = d2; e2.seed = m->cfg.seed;
    e2.step_ctr = m->step_ctr;
    TRY(linear_fwd(m->eh1.p, m->ldh, B, m->P[P_W2], e2, s, gmode(m)));
    EpiStore e3; e3.out = z_dst; e3.ld = ldz_dst;
    TRY(linear_fwd(m->eh2.p, m->ldh, B, m->P[P_W3], e3, s, gmode(m)));
    if (m->cfg.enc_final != AAE_FINAL_LINEAR) {
        hipLaunchKernelGGL(final_act_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, s, z_dst, B, m->c, ldz_dst,
                           m->cfg.enc_final, (float*)nullptr, 0);
        LAUNCHCHK("final_act_fwd");
    }
    return AAE_OK;
}

int launch_w1_items(aae_model* m, const float* ga1, int rpb, size_t bstride, int which, hipStream_t s);
W1Items w1_items_args(aae_model* m, const float* ga1, int rpb, size_t bstride, int which);
int ensure_buckets(aae_model* m, hipStream_t s);

// Encoder backward from dL/dz (gz [rows][ldgz]) + optimiser `which` (O_ENC or O_GEN) on all
// encoder parameters.  z [rows][ldzz] is the encoder output of the matching forward.
int encoder_backward(aae_model* m, const float* gz, int ldgz, const float* z, int ldzz, const uint8_t* mk1,
                     const uint8_t* mk2, uint32_t sid1, uint32_t sid2, int which, hipStream_t s) {
    const int B = m->rows, h = m->h, cc = m->c;
    DropSpec d1 = make_drop(m, 0, true, mk1, nullptr, B, h, sid1);
    DropSpec d2 = make_drop(m, 1, true, mk2, nullptr, B, h, sid2);
    const float* ga3 = gz; int ldga3 = ldgz;
    if (m->cfg.enc_final != AAE_FINAL_LINEAR) {
        hipLaunchKernelGGL(final_act_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, s, z, ldzz, gz, ldgz, m->ga3.p,
                           m->ldz, B, cc, m->cfg.enc_final);
        LAUNCHCHK("final_act_bwd");
        ga3 = m->ga3.p; ldga3 = m->ldz;
    }
    // lin3: dX first (needs the old weights), then dW + update
    EpiActBwd b2; b2.out = m->gb0.p; b2.ld = m->ldh; b2.y = m->eh2.p; b2.ldy = m->ldh; b2.act = m->cfg.activation;
    b2.d = d2; b2.seed = m->cfg.seed; b2.step_ctr = m->step_ctr;
    TRY(linear_dx(ga3, ldga3, B, m->P[P_W3], h, b2, s, gmode(m)));
    TRY(linear_dw(m, ga3, ldga3, B, m->eh2.p, m->ldh, P_W3, which, s));
    // lin2
    EpiActBwd b1; b1.out = m->gb1.p; b1.ld = m->ldh; b1.y = m->eh1.p; b1.ldy = m->ldh; b1.act = m->cfg.activation;
    b1.d = d1; b1.seed = m->cfg.seed; b1.step_ctr = m->step_ctr;
    TRY(linear_dx(m->gb0.p, m->ldh, B, m->P[P_W2], h, b1, s, gmode(m)));
    TRY(linear_dw(m, m->gb0.p, m->ldh, B, m->eh1.p, m->ldh, P_W2, which, s));
    // lin1: bias column sum + its optimiser, then the row-sparse weight gradient + optimiser (w1_update.h)
    const int set = (which == O_GEN) ? 1 : 0;
    const bool exportg = m->cfg.grad_mode == AAE_GRAD_EXPORT;
    hipLaunchKernelGGL(colsum_adam_kernel, dim3((h + 63) / 64), dim3(1024), 0, s, m->gb1.p, B, h, m->ldh,
                       m->P[P_B1].p, m->M[set][P_B1].p, m->V[set][P_B1].p, exportg ? m->Gr[P_B1].p : (float*)nullptr,
                       m->sc + which);
    LAUNCHCHK("colsum_adam");
    return launch_w1_items(m, m->gb1.p, 0, 0, which, s);
}

// Discriminator forward on `rows` rows of m->zin (aae.py:195-213) -> m->dout (sigmoid)
int disc_forward(aae_model* m, int rows, const uint8_t* m1a, const uint8_t* m1b, const uint8_t* m2a,
                 const uint8_t* m2b, int split, uint32_t sid1, uint32_t sid2, hipStream_t s) {
    const int h = m->h;
    DropSpec d1 = make_drop(m, 0, true, m1a, m1b, split, h, sid1);
    DropSpec d2 = make_drop(m, 1, true, m2a, m2b, split, h, sid2);
    EpiDropAct e1; e1.out = m->xh1.p; e1.ld = m->ldh; e1.act = m->cfg.activation; e1.d = d1; e1.seed = m->cfg.seed;
    e1.step_ctr = m->step_ctr;
    TRY(linear_fwd(m->zin.p, m->ldz, rows, m->P[P_D1], e1, s, gmode(m)));
    EpiDropAct e2 = e1; e2.out = m->xh2.p; e2.d = d2;
    TRY(linear_fwd(m->xh1.p, m->ldh, rows, m->P[P_D2], e2, s, gmode(m)));
    EpiSigmoid e3; e3.out = m->dout.p; e3.ld = 4;
    TRY(linear_fwd(m->xh2.p, m->ldh, rows, m->P[P_D3], e3, s));
    return AAE_OK;
}

int finalize_bce_loss(aae_model* m, int nblocks, hipStream_t s) {
    const int B = m->rows;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, m->bce_partials, nblocks, m->fix_partials,
                       B * m->chunks,
                       1.0f / ((float)B * (float)m->N), m->losses, 0);
    LAUNCHCHK("loss_finalize");
    return AAE_OK;
}

int decoder_hidden_forward(aae_model* m, bool train, const uint8_t* mk1, const uint8_t* mk2, int rows,
                           hipStream_t s) {
    DropSpec d1 = make_drop(m, 0, train, mk1, nullptr, rows, m->h, 2);
    DropSpec d2 = make_drop(m, 1, train, mk2, nullptr, rows, m->h, 3);
    EpiDropAct e1; e1.out = m->dh1.p; e1.ld = m->ldh; e1.act = m->cfg.activation; e1.d = d1; e1.seed = m->cfg.seed;
    e1.step_ctr = m->step_ctr;
    TRY(linear_fwd(m->zc.p, m->ldc, rows, m->P[P_V1], e1, s, gmode(m)));
    EpiDropAct e2 = e1; e2.out = m->dh2.p; e2.d = d2;
    TRY(linear_fwd(m->dh1.p, m->ldh, rows, m->P[P_V2], e2, s, gmode(m)));
    return AAE_OK;
}

int stage_zc(aae_model* m, const float* src, int64_t ld, int rows, hipStream_t s) {
    if (src != m->zc.p) {
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)rows * m->cp)), dim3(256), 0, s, src, (int)ld, m->zc.p,
                           m->ldc, rows, m->cp, 1.0f);
        LAUNCHCHK("copy zc");
    }
    return AAE_OK;
}


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
    return W4Copies{m->PT[pid].p, m->D4[pid].p, (int)m->P[pid].rows, (int)m->P[pid].cols};
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
    return o;
}

// dX of a hidden layer: dst[rows][N] = epi(src[rows][K] * W[K][0:N]).  chain4.h reads the D4 copy (k = output row) with
// 16-byte loads, or the matrix itself (k-major for this product); chain.h walks the matrix.
ChainOp cop_dx(aae_model* m, int pid, int src, int dst, int K, int N, int epi, hipStream_t s) {
    ChainOp o = cop_linear(COP_LINEAR_DX, src, dst, m->P[pid], K, N, epi);       // (Wkn = the matrix itself: k-major for this product)
    if (m->D4[pid].p) { ensure_pt(m, pid, s); o.W4 = m->D4[pid].p; o.ns4 = (int)m->P[pid].cols; }
    return o;
}

struct ChainBuilder {
    ChainProgram P;
    ChainBuilder(const aae_model* m, int rows) {
        memset(&P, 0, sizeof(P));
        P.rows = rows; P.act = m->cfg.activation; P.seed = m->cfg.seed; P.step_ctr = m->step_ctr;
        P.loss_out = m->losses; P.loss_slot = 3;
        { const char* e = getenv("AAE_CHAIN_SKIP"); P.dbg = e ? atoi(e) : 0; }
    }
    ChainOp& add(const ChainOp& o) { P.ops[P.nops] = o; return P.ops[P.nops++]; }
};

int launch_chain(aae_model* m, ChainBuilder& cb, hipStream_t s) {
    if (cb.P.nops > kCMaxOps) return fail(AAE_ESTATE, "chain program too long");
    const int grid = (cb.P.rows + kCR - 1) / kCR + (cb.P.bk.enabled ? 1 : 0);
    static const bool want_ts = getenv("AAE_CHAIN_TS") != nullptr;      // debug: per-op timeline of workgroup 0
    static unsigned long long* ts_dev = nullptr;
    if (want_ts) {
        if (!ts_dev && hipMalloc(&ts_dev, 32 * sizeof(unsigned long long)) != hipSuccess) return fail(AAE_EHIP, "ts alloc");
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
        if (cb.P.ops[i].row_lo > 0 && !four) return fail(AAE_ESTATE, "a program prefix for the upper rows needs the 4-row chain kernel");
    if (four) {
        const int grid4 = (cb.P.rows + kR4 - 1) / kR4 + (cb.P.bk.enabled ? 1 : 0);
        if (m->bf16) hipLaunchKernelGGL(chain4_kernel<true>, dim3(grid4), dim3(kC4T), kCSlots * kCR * kCL * sizeof(float), s, cb.P);
        else hipLaunchKernelGGL(chain4_kernel<false>, dim3(grid4), dim3(kC4T), kCSlots * kCR * kCL * sizeof(float), s, cb.P);
    } else if (m->bf16) hipLaunchKernelGGL(chain_kernel<true>, dim3(grid), dim3(kCT), kCSlots * kCR * kCL * sizeof(float), s, cb.P);
    else hipLaunchKernelGGL(chain_kernel<false>, dim3(grid), dim3(kCT), kCSlots * kCR * kCL * sizeof(float), s, cb.P);
    LAUNCHCHK("chain_kernel");
    if (want_ts) {
        unsigned long long h[32];
        hipStreamSynchronize(s);
        hipMemcpy(h, ts_dev, sizeof(h), hipMemcpyDeviceToHost);
        static const char* names[] = {"LOAD", "LINEAR", "LINEAR_DX", "FINAL_FWD", "FINAL_BWD", "ADV", "DROPACT", "SLABSUM", "ACTBWD", "STORE", "REPARAM", "REPARAM_BWD", "DISC_HEAD", "PRIOR"};
        fprintf(stderr, "[chain rows=%d nops=%d total=%.2fus]", cb.P.rows, cb.P.nops, (h[cb.P.nops] - h[0]) * 0.01);
        for (int i = 0; i < cb.P.nops; ++i)
            fprintf(stderr, " %s(K%d,N%d%s%s)=%.2f", names[cb.P.ops[i].kind], cb.P.ops[i].K, cb.P.ops[i].N,
                    cb.P.ops[i].out ? ",st" : "", cb.P.ops[i].out2 ? ",st2" : "", (h[i + 1] - h[i]) * 0.01);
        fprintf(stderr, "\n");
        if (cb.P.nops > 2)
            fprintf(stderr, "   [op 2, wave 0 of workgroup 0] loads+mfma+partials=%.2f wait-barrier=%.2f epi-ctx=%.2f epilogue=%.2f barrier=%.2f (us)\n",
                    (h[21] - h[20]) * 0.01, (h[22] - h[21]) * 0.01, (h[23] - h[22]) * 0.01, (h[24] - h[23]) * 0.01, (h[25] - h[24]) * 0.01);
    }
    return AAE_OK;
}

// up to 4 weight-gradient jobs in one launch
struct DwBuilder {
    DwGroup g; int tiles;
    DwBuilder() { memset(&g, 0, sizeof(g)); tiles = 0; }
    void add(aae_model* m, const float* G, int ldg, const float* X, int ldx, int rows, int pid, int which) {
        DwJob& J = g.jobs[g.njobs++];
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
        // first layer (the rows live with their item slices)
        static const bool no_merge = getenv("AAE_NO_W1_MERGE") != nullptr;
        w.nitem = 0;
        m->w1_items_merged = false;
        if (!no_merge && !m->dense_step && !m->ext_first && sizeof(int) * w1_items_lds_words(m->rows) <= kDwSmemBytes) {
            TRY(ensure_buckets(m, s));
            w.items = w1_items_args(m, ga1, 0, 0, which);
            w.nitem = std::min(m->cfg.max_nnz, std::max(256, m->rows * 32));
            m->w1_items_merged = true;
        }
        return AAE_OK;
    }
    int launch(hipStream_t s) {
        int blocks = tiles;
        if (g.w1.enabled) { g.w1.blk0 = tiles; blocks += g.w1.ncol + g.w1.nitem; }
        hipLaunchKernelGGL(grouped_dw_kernel, dim3(blocks), dim3(256), 0, s, g);
        LAUNCHCHK("grouped_dw_kernel");
        return AAE_OK;
    }
};

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
static bool fused_decoder_applies(const aae_model* m) {
    const bool one = m->rows <= 16 * kMB;
    // The row-blocked form pays while its deferred half (2 * rows * N * (h + 1) flop of GEMM2 at the optimiser kernel's
    // ~30 TFLOP/s) fits beside the rest of the step: 800 rows x 12.5 k items (an item slice at world 8) 0.40 against 0.46 ms
    // per step, 208 x 100 k 0.64 against 0.70; beyond ~32 M cells the next step waits for it and the three GEMMs win
    // (512 x 100 k: 1.48 against 1.12 ms; 512 x 275 k, a C5 slice: 3.7 against 2.7 ms).  AAE_BLOCKED_ANY lifts the cap (tests).
    // r3: with both launches on the emulated product (dec_crit_x3.h: the deferred half of all blocks in ONE launch for any
    // vocabulary, dec_opt_blocks_x3_kernel) the cap is gone: 512 x 100 k 0.77 ms/step against 0.93 on the three GEMMs.
    const bool blocked = !one && m->blocked_ok && !m->bf16 && m->split_ok && m->split_wgs > 0 && m->Gacc.p && row_blocks(m) <= kMaxRowBlocks &&
                         m->cfg.grad_mode == AAE_GRAD_FUSED &&
                         (m->blocked_any || (size_t)m->rows * m->N <= ((size_t)32 << 20) || (m->x3_ok && m->dh2f.p && getenv("AAE_NO_OPT_BLOCKS_X3") == nullptr));
    return m->fused_ok && !m->force_unfused && (one || blocked) &&
           ((size_t)m->N + 2 * kTI) * m->ldh * sizeof(float) < ((size_t)1 << 31) &&      /* (stores without a cell are dropped by a buffer bounds check at offset 2^31) */
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
        const size_t lds = sizeof(int) * ((size_t)ntiles + 1 + kBucketWideDocs + 1 + 1024);
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
    const int items = std::min(m->cfg.max_nnz, std::max(256, m->rows * 32));
    hipLaunchKernelGGL(w1_item_update_kernel, dim3(items), dim3(256), lds, s, a);
    LAUNCHCHK("w1_item_update");
    return AAE_OK;
}

static void piggyback_buckets(aae_model* m, ChainBuilder& cb) {
    const int ntiles = (m->N + kTI - 1) / kTI;
    const size_t need = sizeof(int) * ((size_t)ntiles + 1 + kBucketMaxDocs + 1 + 1024);
    if (m->buckets_valid || !fused_decoder_applies(m) || ntiles > kBucketMaxTiles || m->rows > kBucketMaxDocs ||
        need > (size_t)kCSlots * kCR * kCL * sizeof(float) || getenv("AAE_NO_PIGGYBACK"))
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
        if (m->cfg.cond_inc > 0) {
            ChainOp& cl = cb.add(cop_load(cond_dev, m->cfg.cond_inc, 2, m->cfg.cond_inc)); cl.dst_col0 = c; cl.one_col = cp;
        } else {
            f.one_col = cp;
        }
        ChainOp& v1 = cb.add(cop_fwd(m, P_V1, 2, 3, cp + 1, h, CEPI_DROPACT, s));
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
    ChainOp& l = cb.add(cop_load(m->zc.p, m->ldc, 0, cp)); l.one_col = cp;
    ChainOp& v1 = cb.add(cop_fwd(m, P_V1, 0, 1, cp + 1, h, CEPI_DROPACT, s));
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
        cb.add(cop_load(m->dh1.p, m->ldh, 3, h));
        ChainOp& x2 = cb.add(cop_dx(m, P_V2, 2, 4, h, h, CEPI_ACTBWD, s)); x2.yslot = 3;
        x2.d = make_drop(m, 0, true, I.masks_dev[2], nullptr, B, h, 2); cop_out(x2, m->gb1.p, m->ldh);
        ChainOp& x1 = cb.add(cop_dx(m, P_V1, 4, 5, h, cp, CEPI_NONE, s));
        cop_out(x1, m->gzc.p, m->ldc);
        if (dzc_out) { x1.out2 = dzc_out; x1.ldo2 = cp; }
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
        cb.add(cop_load(m->eh2.p, m->ldh, 8, h));
        ChainOp& x3 = cb.add(cop_dx(m, P_W3, sg, 9, c, h, CEPI_ACTBWD, s)); x3.yslot = 8;
        x3.d = make_drop(m, 1, true, I.masks_dev[which == O_GEN ? 9 : 1], nullptr, B, h, which == O_GEN ? 9 : 1);
        cop_out(x3, m->gb2.p, m->ldh);
        cb.add(cop_load(m->eh1.p, m->ldh, 0, h));
        ChainOp& x2 = cb.add(cop_dx(m, P_W2, 9, 1, h, h, CEPI_ACTBWD, s)); x2.yslot = 0;
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
    hipExtLaunchKernelGGL(enc_gather_kernel, dim3(B), dim3(1024), (uint32_t)shm, s, nullptr, head ? m->ev_head : nullptr, 0, m->bv,
                          (const float*)m->P[P_W1T].p, m->ldw1, (const float*)m->P[P_B1].p, h, (int)m->cfg.normalize_inputs,
                          m->a1.p, m->eh1.p, m->ldh, (int)m->cfg.activation, d1, (uint64_t)m->cfg.seed,
                          (const long long*)m->step_ctr, m->rscale, m->doc_l1,
                          // open_step: the step-opening bookkeeping rides in this launch (advance_step_body) and the
                          // workgroups take the step number from the host (m->hstep == *step_ctr once the step is open)
                          AdvanceJob{m->sc, m->step_ctr, m->lazy ? m->tab : nullptr, m->losses, open_step ? 1 : 0},
                          (long long)(open_step ? m->hstep : -1));
    LAUNCHCHK("enc_gather");
    return AAE_OK;
}

static bool same_batch(const aae_batch& a, const aae_batch& b) {
    return a.indptr_dev == b.indptr_dev && a.indices_dev == b.indices_dev && a.values_dev == b.values_dev &&
           a.rows_dev == b.rows_dev && a.row_start == b.row_start && a.n_rows == b.n_rows;
}

// aae_prefetch_batch, second half: the hinted batch's unique-item list + deferred-Adam catch-up (through the RUNNING
// step, whose scalars advance_step has published by the time ev_head fires) on the side stream, into the second list
// set.  Rows of the running batch are skipped: the step's own updates bring them to the same step.
int launch_prefetch(aae_model* m, bool wait_head = true) {
    const aae_batch& b = m->pf_batch;
    m->pf_armed = false;
    m->pf_after_opt = !wait_head;
    if (!m->side || !m->mark2 || !m->lazy) return AAE_OK;
    BatchView bv; bv.indptr = b.indptr_dev; bv.indices = b.indices_dev; bv.values = b.values_dev;
    bv.rows = b.rows_dev; bv.row_start = b.row_start; bv.n_rows = b.n_rows;
    const int mr = b.max_row_nnz > 0 ? b.max_row_nnz : 1024;
    const int chunks = std::max(1, std::min(64, (mr + 15) / 16));
    const int gy = std::max(1, std::min(16, chunks / 16 + 1));
    hipStream_t q = m->side;
    if (wait_head) HIPCHK(hipStreamWaitEvent(q, m->ev_head, 0));      // (else: the caller enqueues behind work that is ordered behind the step's head)
    hipLaunchKernelGGL(bump_stamp_kernel, dim3(1), dim3(1), 0, q, m->stamp2, m->ucount2);
    hipLaunchKernelGGL(uniq_items_kernel, dim3(b.n_rows, gy), dim3(256), 0, q, bv, m->mark2, m->stamp2, m->ulist2, m->ucount2);
    if (m->cfg.optimizer == AAE_OPT_ADAM) {
        const int grid = std::min(m->cfg.max_nnz, std::max(256, b.n_rows * 32));
        hipLaunchKernelGGL(w1_catchup_kernel, dim3(grid), dim3(256), 0, q, m->ulist2, m->ucount2, m->N, m->tsync,
                           m->P[P_W1T].p, m->M[0][P_W1T].p, m->V[0][P_W1T].p, m->M[1][P_W1T].p, m->V[1][P_W1T].p,
                           m->ldw1, m->h, m->tab, m->step_ctr, 0, m->mark, m->stamp);
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
    static const bool merge_ok = getenv("AAE_NO_DISC_MERGE") == nullptr;
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
        cb.P.loss_slot = 1;
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
    return dw.launch(s);
}

// gen_step on the chain path: everything between the shared gather and the weight gradients is
// row-local and runs as ONE program
int chain_gen_step(aae_model* m, hipStream_t s) {
    const int B = m->rows, h = m->h, c = m->c;
    const aae_rng_inject& I = m->inj;
    ChainBuilder cb(m, B);
    cb.P.loss_slot = 2;
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
    TRY(dw.launch(s));
    if (m->ext_first) return AAE_OK;           // dL/d(a1) waits in AAE_T_ACT_GA1 for the owner(s) of the first layer
    return encoder_first_layer_update(m, m->gb3.p, O_GEN, s, true);
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" {

int aae_abi_version(void) { return AAE_ABI_VERSION; }
const char* aae_last_error(void) { return g_err.c_str(); }

int aae_arena_bytes(const aae_config* cfg, size_t* bytes_out) {
    TRY(validate(cfg));
    if (!bytes_out) return fail(AAE_EINVAL, "bytes_out is NULL");
    aae_model tmp; memset((void*)&tmp, 0, sizeof(tmp)); tmp.cfg = *cfg;
    *bytes_out = layout(&tmp, nullptr, true);
    return AAE_OK;
}

int aae_create(const aae_config* cfg, void* arena_dev, size_t arena_bytes, void* stream, aae_handle* out) {
    TRY(validate(cfg));
    if (!arena_dev || !out) return fail(AAE_EINVAL, "arena/out is NULL");
    if (reinterpret_cast<uintptr_t>(arena_dev) & 255) return fail(AAE_EINVAL, "arena must be 256-byte aligned");
    aae_model* m = new aae_model();
    memset((void*)m, 0, sizeof(*m));
    m->cfg = *cfg;
    size_t need = layout(m, static_cast<char*>(arena_dev), false);
    if (need > arena_bytes) { delete m; return fail(AAE_ENOMEM, "arena smaller than aae_arena_bytes()"); }
    m->base = static_cast<char*>(arena_dev); m->bytes = need;
    m->alpha_mode = cfg->activation == AAE_ACT_SELU;
    m->vae = cfg->model_kind == 3; m->vae_bwd = false; m->vae_cut = false;
    m->bf16 = cfg->dtype == 1;
    m->blocked_ok = cfg->blocked_output == 1;
    m->blocked_any = getenv("AAE_BLOCKED_ANY") != nullptr;
    m->split_any = getenv("AAE_SPLIT_ANY") != nullptr;
    m->x3_gemm = !m->bf16 && getenv("AAE_NO_GEMM_X3") == nullptr;
    m->noise_next = nullptr; m->noise_ld = 0; m->dense_step = false;
    m->ae_only = cfg->model_kind == 1 || m->vae;
    m->lazy = true;    // deferred Adam on W1T in both gradient modes (export mode exchanges packed rows)
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
            delete m; return fail(AAE_EHIP, "no HIP device");
        }
        m->n_cu = std::min(cus, 304);
        const int nb = (m->h + 1 + 15) / 16;
        m->fused_nb = nb <= 4 ? 4 : nb <= 7 ? 7 : nb <= 13 ? 13 : 0;
        m->fused_ok = m->fused_nb != 0 && fused_width_ok(m->h, m->ldh);       // (the arena's slab area is sized by the same test)
        m->use_chain = (m->h + 1 <= 208) && (m->cp + 1 <= 208) && getenv("AAE_NO_CHAIN") == nullptr;
        if (m->use_chain && (hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 kCSlots * kCR * kCL * (int)sizeof(float)) != hipSuccess ||
                             hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 kCSlots * kCR * kCL * (int)sizeof(float)) != hipSuccess))
            m->use_chain = false;
        m->use_chain4 = m->use_chain && getenv("AAE_CHAIN16") == nullptr &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(chain4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            kCSlots * kCR * kCL * (int)sizeof(float)) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(chain4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            kCSlots * kCR * kCL * (int)sizeof(float)) == hipSuccess;
        m->force_unfused = cfg->unfused_decoder == 1;   // debugging / A-B switch: unfused_decoder = 1 keeps the 3-kernel path
        if (m->fused_ok) {
            const int maxlds = 160 * 1024;
            hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
            hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_kernel<13>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
            hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_bucket_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds - 2048);
            m->bucket_wide_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_bucket_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds - 16384) == hipSuccess;
            if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) m->fused_ok = false;
            if (m->bf16 &&
                (hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_bf16_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds) != hipSuccess ||
                 hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_bf16_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds) != hipSuccess ||
                 hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_bf16_kernel<13>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds) != hipSuccess))
                m->fused_ok = false;
        }
    }
    m->w1_big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(w1_item_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(sizeof(int) * w1_items_lds_words(16384))) == hipSuccess;
    (void)hipGetLastError();
    m->grad_scale = 1.f;
    m->rng_row0 = 0; m->rng_global = 0;
    m->Gt = m->G.p;
    m->split_ok = false; m->opt_pending = false; m->side = nullptr; m->ev_crit = m->ev_opt = nullptr;
    // workgroups of the deferred optimiser launch.  r2: half the CUs (it had 50 us of slack then).  Since the critical launch
    // runs on the bf16 matrix cores (r3, 116 -> 84 us) the NEXT step waits for this launch; measured at C3, batch 100
    // (AAE_SPLIT_WGS sweep, ms/step), with the fp32 deferred kernel: 112 0.306 | 128 0.291 | 144 0.274 | 160 0.274 | 176 0.284
    // | 192 0.285 | 224 0.312; with the bf16-emulated one (dec_opt_x3_kernel: 160 -> 138 us on 128 workgroups): 112 0.279 |
    // 128 0.273 | 144 0.276 | 160 0.278 | 176 0.286.  Half the CUs again (set below once x3_ok is known; 5/8 without it):
    // beyond that the chain / weight-gradient kernels it runs beside lose more than the launch gains.
    m->split_wgs = std::max(1, (m->n_cu * 5) / 8);
    { const char* e = getenv("AAE_SPLIT_WGS"); if (e) m->split_wgs = atoi(e); }
    m->pf_armed = m->pf_built = m->pf_pending = false; m->pf_step = -1; m->hstep = 0; m->ev_head = m->ev_pf = nullptr;
    bool side_ok = false;
    if (cfg->grad_mode == AAE_GRAD_FUSED) {
        // the handle's side stream (lowest priority) for work off the step's critical path: the deferred optimiser
        // launch of the output layer and the next batch's deferred-Adam catch-up.  The events order work of this device
        // only: no system-scope release (an L2 write-back) at the record
        int lo = 0, hi = 0;
        const unsigned evflags = hipEventDisableTiming | hipEventDisableSystemFence;
        side_ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess &&
                  hipStreamCreateWithPriority(&m->side, hipStreamNonBlocking, lo) == hipSuccess &&
                  hipEventCreateWithFlags(&m->ev_crit, evflags) == hipSuccess &&
                  hipEventCreateWithFlags(&m->ev_opt, evflags) == hipSuccess &&
                  hipEventCreateWithFlags(&m->ev_head, evflags) == hipSuccess &&
                  hipEventCreateWithFlags(&m->ev_pf, evflags) == hipSuccess &&
                  hipEventCreateWithFlags(&m->ev_bk, evflags) == hipSuccess;
        if (!side_ok) {
            if (m->side) (void)hipStreamDestroy(m->side);
            m->side = nullptr;
        }
        (void)hipGetLastError();
    }
    if (side_ok && m->fused_ok && m->bf16 && m->split_wgs > 0 &&
        (size_t)((m->N + kBU - 1) / kBU) * kBU * bf_stride(4) <= (size_t)m->R * m->ldn) {      // (room for the stored gT images in the G scratch)
        m->split_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_bf16_kernel<4, kDecCrit>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_bf16_kernel<7, kDecCrit>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_bf16_kernel<13, kDecCrit>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_bf16_kernel<4, kDecOpt>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_bf16_kernel<7, kDecOpt>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_bf16_kernel<13, kDecOpt>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        (void)hipGetLastError();
    }
    if (side_ok && m->fused_ok && !m->bf16 && m->split_wgs > 0 &&
        (size_t)((m->N + kTI - 1) / kTI) * kTI * (size_t)std::min(m->R, 16 * kMB) * sizeof(float) < (size_t)0x7FFFFFF0u) {
        bool ok = true;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_kernel<4, kDecCrit>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_kernel<7, kDecCrit>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_kernel<13, kDecCrit>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_kernel<4, kDecOpt>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_kernel<7, kDecOpt>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_kernel<13, kDecOpt>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_kernel<4, kDecOptAcc>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_kernel<7, kDecOptAcc>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_kernel<13, kDecOptAcc>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_opt_blocks_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_opt_blocks_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_opt_blocks_kernel<13>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        m->split_ok = ok;
        static const bool no_x3 = getenv("AAE_NO_X3") != nullptr;
        m->x3_ok = ok && !no_x3
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_crit_x3_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_crit_x3_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_crit_x3_kernel<13>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_opt_blocks_x3_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_opt_blocks_x3_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_opt_blocks_x3_kernel<13>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_opt_blocks_x3_kernel<13, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        (void)hipGetLastError();
        if (m->x3_ok && getenv("AAE_NO_OPT_X3") == nullptr && getenv("AAE_SPLIT_WGS") == nullptr) m->split_wgs = std::max(1, m->n_cu / 2);
    }
    hipStream_t s = S(stream);
    hipError_t e = hipMemsetAsync(arena_dev, 0, need, s);
    if (e != hipSuccess) { delete m; return fail(AAE_EHIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e)); }
    // constant-1 columns of the activation buffers that feed augmented weights
    struct { Ten* t; int col; } ones[] = {{&m->eh1, m->h}, {&m->eh2, m->h}, {&m->zc, m->cp}, {&m->dh1, m->h},
                                           {&m->dh2, m->h}, {&m->zin, m->c}, {&m->xh1, m->h}, {&m->xh2, m->h}};
    for (auto& o : ones)
        hipLaunchKernelGGL(fill_col_kernel, dim3(((int)o.t->rows + 255) / 256), dim3(256), 0, s, o.t->p, (int)o.t->ld,
                           (int)o.t->rows, o.col, 1.0f);
    OptScalars hs[4]; memset(hs, 0, sizeof(hs));
    for (int i = 0; i < 4; ++i) {
        hs[i].t = 0; hs[i].is_sgd = cfg->optimizer == AAE_OPT_SGD;
        hs[i].lr = (i == O_ENC || i == O_DEC) ? (double)cfg->gen_lr : (double)cfg->reg_lr;
        hs[i].neg_step_size = 0.f; hs[i].bc2_sqrt = 1.f; hs[i].inv_bc2_sqrt = 1.f;
    }
    e = hipMemcpyAsync(m->sc, hs, sizeof(hs), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { delete m; return fail(AAE_EHIP, std::string("init: ") + hipGetErrorString(e)); }
    *out = m;
    return AAE_OK;
}

int aae_destroy(aae_handle h) {
    if (!h) return AAE_OK;
    if (h->side) {
        (void)hipStreamSynchronize(h->side);     // the arena is the caller's: nothing of ours may still write it
        (void)hipStreamDestroy(h->side);
    }
    if (h->dp_scratch) (void)hipFree(h->dp_scratch);
    if (h->ev_crit) (void)hipEventDestroy(h->ev_crit);
    if (h->ev_opt) (void)hipEventDestroy(h->ev_opt);
    if (h->ev_head) (void)hipEventDestroy(h->ev_head);
    if (h->ev_pf) (void)hipEventDestroy(h->ev_pf);
    if (h->ev_bk) (void)hipEventDestroy(h->ev_bk);
    if (h->prof_ev) {
        for (int k = 0; k < AAE_K_N; ++k)
            for (auto& pr : h->prof_ev[k]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        delete[] h->prof_ev;
    }
    delete h;
    return AAE_OK;
}

// ---- CategoricalCondition (cond_embed.h): stateless entry points over caller-owned tables --------------------
static int cat_check(int32_t vocab, int32_t dim, const int32_t* idx_dev, int32_t rows, int32_t width, int32_t reduce) {
    if (vocab < 1 || dim < 1 || dim > kCatMaxDim) return fail(AAE_EINVAL, "categorical condition: need vocab >= 1 and 1 <= dim <= 256");
    if (!idx_dev || rows < 1 || width < 1) return fail(AAE_EINVAL, "categorical condition: empty index block");
    if ((int64_t)rows * width > (1 << 22)) return fail(AAE_EINVAL, "categorical condition: rows * width > 2^22");
    if (reduce != AAE_CAT_SUM && reduce != AAE_CAT_MEAN) return fail(AAE_EINVAL, "categorical condition: reduce must be sum or mean");
    return AAE_OK;
}

int aae_cat_encode(const float* table_dev, int32_t vocab, int32_t dim, const int32_t* idx_dev, int32_t rows,
                   int32_t width, int32_t reduce, float* out_dev, int64_t out_ld, void* stream) {
    TRY(cat_check(vocab, dim, idx_dev, rows, width, reduce));
    if (!table_dev || !out_dev || out_ld < dim) return fail(AAE_EINVAL, "aae_cat_encode: table/out is NULL or out_ld < dim");
    hipLaunchKernelGGL(cat_encode_kernel, dim3(rows, (dim + 63) / 64), dim3(64), 0, S(stream), table_dev, vocab, dim,
                       idx_dev, width, reduce == AAE_CAT_MEAN, out_dev, (long long)out_ld);
    LAUNCHCHK("cat_encode");
    return AAE_OK;
}

int aae_cat_update(float* table_dev, float* exp_avg_dev, float* exp_avg_sq_dev, float* grad_scratch_dev, int32_t vocab,
                   int32_t dim, const int32_t* idx_dev, int32_t rows, int32_t width, int32_t reduce,
                   const float* dout_dev, int64_t dout_ld, int32_t optimizer, double lr, int64_t step, void* stream) {
    TRY(cat_check(vocab, dim, idx_dev, rows, width, reduce));
    if (!table_dev || !exp_avg_dev || !exp_avg_sq_dev || !dout_dev || dout_ld < dim)
        return fail(AAE_EINVAL, "aae_cat_update: table/state/dout is NULL or dout_ld < dim");
    if (optimizer != AAE_CAT_SPARSE_ADAM && optimizer != AAE_CAT_ADAM) return fail(AAE_EINVAL, "aae_cat_update: unknown optimizer");
    if (optimizer == AAE_CAT_ADAM && !grad_scratch_dev) return fail(AAE_EINVAL, "aae_cat_update: dense Adam needs grad_scratch_dev");
    if (step < 1) return fail(AAE_EINVAL, "aae_cat_update: step counts from 1");
    const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
    CatUpdate a;
    a.table = table_dev; a.m = exp_avg_dev; a.v = exp_avg_sq_dev;
    a.gdense = optimizer == AAE_CAT_ADAM ? grad_scratch_dev : nullptr;
    a.idx = idx_dev; a.d = dout_dev; a.ldd = dout_ld; a.vocab = vocab; a.dim = dim; a.rows = rows; a.width = width;
    a.mean = reduce == AAE_CAT_MEAN;
    a.neg_step_size = (float)(-(lr * sqrt(bc2) / bc1));
    const int n = rows * width;
    hipLaunchKernelGGL(cat_update_kernel, dim3((n + kCatWaves - 1) / kCatWaves), dim3(64 * kCatWaves), 0, S(stream), a);
    LAUNCHCHK("cat_update");
    if (optimizer == AAE_CAT_ADAM) {
        OptScalars sc; memset(&sc, 0, sizeof(sc));
        sc.t = step; sc.neg_step_size = (float)(-(lr / bc1)); sc.bc2_sqrt = (float)sqrt(bc2); sc.lr = lr;
        sc.inv_bc2_sqrt = 1.0f / sc.bc2_sqrt;
        const size_t total = (size_t)vocab * dim;
        hipLaunchKernelGGL(cat_dense_adam_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream), table_dev,
                           exp_avg_dev, exp_avg_sq_dev, grad_scratch_dev, total, sc);
        LAUNCHCHK("cat_dense_adam");
    }
    return AAE_OK;
}

int aae_csr_embed(const int64_t* indptr_dev, const int32_t* indices_dev, const float* values_dev, int32_t n_rows,
                  const float* table_dev, int32_t n_table_rows, int32_t dim, int64_t table_ld, float* out_dev,
                  int64_t out_ld, void* stream) {
    if (!indptr_dev || !table_dev || !out_dev) return fail(AAE_EINVAL, "aae_csr_embed: indptr/table/out is NULL");
    if (n_rows < 0 || n_table_rows < 1 || dim < 1 || table_ld < dim || out_ld < dim)
        return fail(AAE_EINVAL, "aae_csr_embed: bad shape (need n_table_rows >= 1, dim >= 1, leading dimensions >= dim)");
    if (n_rows == 0) return AAE_OK;
    if (!indices_dev || !values_dev) return fail(AAE_EINVAL, "aae_csr_embed: indices/values is NULL");
    hipLaunchKernelGGL(csr_embed_kernel, dim3(n_rows), dim3(256), 0, S(stream), (const long long*)indptr_dev, indices_dev,
                       values_dev, table_dev, n_table_rows, dim, (long long)table_ld, out_dev, (long long)out_ld);
    LAUNCHCHK("csr_embed");
    return AAE_OK;
}

int aae_dense_to_csr(const void* dense_dev, int32_t elem_bytes, int64_t ld, int32_t rows, int32_t n_cols,
                     int64_t* indptr_dev, int32_t* indices_dev, float* values_dev, int64_t capacity,
                     int32_t* scratch_dev, int32_t* stats_out_host, void* stream) {
    if (!dense_dev || !indptr_dev || !indices_dev || !values_dev || !scratch_dev || !stats_out_host)
        return fail(AAE_EINVAL, "aae_dense_to_csr: NULL argument");
    if (rows < 1 || n_cols < 1 || ld < n_cols || capacity < 1) return fail(AAE_EINVAL, "aae_dense_to_csr: bad shape");
    if (elem_bytes != 4 && elem_bytes != 8) return fail(AAE_EINVAL, "aae_dense_to_csr: elem_bytes must be 4 (float32) or 8 (float64)");
    hipStream_t s = S(stream);
    int* stats = scratch_dev;                 // [0..3] statistics, [8..] row counts
    int* rowcnt = scratch_dev + 8;
    HIPCHK(hipMemsetAsync(stats, 0, 8 * sizeof(int), s));
    if (elem_bytes == 4) hipLaunchKernelGGL(dense_count_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)dense_dev, (long long)ld, n_cols, rowcnt, stats);
    else hipLaunchKernelGGL(dense_count_kernel<double>, dim3(rows), dim3(256), 0, s, (const double*)dense_dev, (long long)ld, n_cols, rowcnt, stats);
    hipLaunchKernelGGL(dense_scan_kernel, dim3(1), dim3(1024), 0, s, rowcnt, rows, (long long)capacity, (long long*)indptr_dev, stats);
    if (elem_bytes == 4) hipLaunchKernelGGL(dense_fill_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)dense_dev, (long long)ld, n_cols, (const long long*)indptr_dev, stats, indices_dev, values_dev);
    else hipLaunchKernelGGL(dense_fill_kernel<double>, dim3(rows), dim3(256), 0, s, (const double*)dense_dev, (long long)ld, n_cols, (const long long*)indptr_dev, stats, indices_dev, values_dev);
    LAUNCHCHK("dense_to_csr");
    HIPCHK(hipMemcpyAsync(stats_out_host, stats, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return AAE_OK;
}

int aae_profile_enable(aae_handle h, int on) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    if (on && !h->prof_ev) h->prof_ev = new std::vector<std::pair<hipEvent_t, hipEvent_t>>[AAE_K_N];
    h->prof_on = on != 0;
    h->prof_mask = on == 1 ? ~0u : ((unsigned)on >> 1);     // 1 = every kernel id; else bit (k + 1) selects id k
    return AAE_OK;
}

int aae_profile_read(aae_handle h, int kernel_id, double* total_ms, int64_t* launches) {
    if (!h || !total_ms || !launches) return fail(AAE_EINVAL, "NULL argument");
    if (kernel_id < 0 || kernel_id >= AAE_K_N) return fail(AAE_EINVAL, "bad kernel id");
    *total_ms = 0.0; *launches = 0;
    if (!h->prof_ev) return AAE_OK;
    auto& v = h->prof_ev[kernel_id];
    for (size_t i = 0; i < h->prof_used[kernel_id]; ++i) {
        HIPCHK(hipEventSynchronize(v[i].second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, v[i].first, v[i].second));
        *total_ms += ms; *launches += 1;
    }
    h->prof_used[kernel_id] = 0;
    return AAE_OK;
}

int aae_prefetch_batch(aae_handle h, const aae_batch* next) {
    if (!h || !next) return fail(AAE_EINVAL, "NULL argument");
    if (!next->indptr_dev || !next->indices_dev || !next->values_dev) return fail(AAE_EINVAL, "batch pointers are NULL");
    if (next->n_rows < 1 || next->n_rows > h->R) return fail(AAE_EINVAL, "batch n_rows outside [1, max_batch]");
    if (next->nnz_bound > h->cfg.max_nnz) return fail(AAE_EINVAL, "batch nnz_bound > max_nnz");
    h->pf_batch = *next;
    h->pf_armed = true;
    return AAE_OK;
}

int aae_set_input_noise(aae_handle h, const float* noise_dev, int64_t noise_ld) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    if (!h->Xn.p) return fail(AAE_ESTATE, "aae_set_input_noise: the model was not created with cfg.dense_noise = 1");
    if (noise_dev && noise_ld < h->N) return fail(AAE_EINVAL, "noise_ld < n_items");
    h->noise_next = noise_dev; h->noise_ld = noise_ld;
    return AAE_OK;
}

// ... the deferred optimiser launch alone (activation views: the next batch's prefetch - aae_prefetch_batch - touches
// enc.lin1 and its bookkeeping only, and keeps running beside the step)
int aae_join_output_layer(aae_handle h, void* stream) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    // (only the optimiser launch is settled here.  A prefetch enqueued BEHIND it - an item slice's, pf_after_opt - is not
    //  covered by ev_opt and stays pending for the next step's join_deferred; one enqueued in front of it - aae_step's -
    //  is covered by the later record of the in-order side stream)
    if (h->opt_pending) {
        HIPCHK(hipStreamWaitEvent(S(stream), h->ev_opt, 0));
        h->opt_pending = false;
        if (!h->pf_after_opt) h->pf_pending = false;
    }
    return AAE_OK;
}

int aae_join(aae_handle h, void* stream) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    return join_deferred(h, S(stream));
}

int aae_set_split(aae_handle h, int32_t workgroups) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    if (workgroups < 0) return fail(AAE_EINVAL, "workgroups must be >= 0");
    if (h->opt_pending) return fail(AAE_ESTATE, "aae_set_split with a deferred launch pending: call aae_join / aae_sync first");
    h->split_wgs = workgroups;
    return AAE_OK;
}

int aae_set_grad_scale(aae_handle h, float scale) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    h->grad_scale = scale; return AAE_OK;
}

// learning rates are float32 in aae_config; callers that need the exact Python double can set it here
int aae_set_rng_rows(aae_handle h, int64_t row_offset, int64_t global_rows) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    if (row_offset < 0 || global_rows < 0 || row_offset > (1 << 30) || global_rows > (1 << 30))
        return fail(AAE_EINVAL, "aae_set_rng_rows: bad offset / global_rows");
    h->rng_row0 = (int)row_offset; h->rng_global = (int)global_rows;
    return AAE_OK;
}

int aae_params_changed(aae_handle h) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    for (int i = 0; i < NP; ++i) h->pt_ok[i] = false;
    return AAE_OK;
}

int aae_set_lr(aae_handle h, double gen_lr, double reg_lr) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    TRY(join_host(h));
    OptScalars hs[4];
    HIPCHK(hipMemcpy(hs, h->sc, sizeof(hs), hipMemcpyDeviceToHost));
    hs[O_ENC].lr = gen_lr; hs[O_DEC].lr = gen_lr; hs[O_GEN].lr = reg_lr; hs[O_DISC].lr = reg_lr;
    HIPCHK(hipMemcpy(h->sc, hs, sizeof(hs), hipMemcpyHostToDevice));
    return AAE_OK;
}

int aae_tensor_info(aae_handle h, int id, aae_tensor* out) {
    if (!h || !out) return fail(AAE_EINVAL, "handle/out is NULL");
    const Ten* t = nullptr;
    Ten tmp;
    auto adam = [&](int base, int lo, int hi, int set) -> const Ten* {
        int k = id - base, pid = lo + k / 2;
        if (k < 0 || pid > hi) return nullptr;
        return (k & 1) ? &h->V[set][pid] : &h->M[set][pid];
    };
    if (id >= 0 && id < AAE_T_N_PARAMS) t = &h->P[id];
    else if (id >= AAE_T_ADAM_ENC && id < AAE_T_ADAM_ENC + 8) t = adam(AAE_T_ADAM_ENC, P_W1T, P_W3, 0);
    else if (id >= AAE_T_ADAM_GEN && id < AAE_T_ADAM_GEN + 8) t = adam(AAE_T_ADAM_GEN, P_W1T, P_W3, 1);
    else if (id >= AAE_T_ADAM_DEC && id < AAE_T_ADAM_DEC + 6) t = adam(AAE_T_ADAM_DEC, P_V1, P_V3, 0);
    else if (id >= AAE_T_ADAM_DISC && id < AAE_T_ADAM_DISC + 6) t = adam(AAE_T_ADAM_DISC, P_D1, P_D3, 0);
    else if (id >= AAE_T_GRAD && id < AAE_T_GRAD + NP) { t = &h->Gr[id - AAE_T_GRAD]; if (!t->rows) t = nullptr; }
    else if (id == AAE_T_ACT_Z) t = &h->zsave;
    else if (id == AAE_T_ACT_A1) t = &h->a1;
    else if (id == AAE_T_ACT_DH2) t = &h->dh2;
    else if (id == AAE_T_ACT_DA2) t = &h->da2;
    else if (id == AAE_T_ACT_GA1) t = (h->ext_first && h->ga1x.p) ? &h->ga1x : &h->gb3;   // (ga1x: the LAST rows hold the batch)
    else if (id == AAE_T_ACT_DZC) { tmp = h->gzc; tmp.cols = h->cp; t = &tmp; }
    else if (id == AAE_T_ACT_LOSSES) {
        tmp.rows = 1; tmp.cols = 4; tmp.ld = 4; tmp.off = (size_t)((char*)h->losses - h->base); t = &tmp;
    }
    if (!t) return fail(AAE_EINVAL, "unknown tensor id");
    out->byte_offset = t->off; out->rows = t->rows; out->cols = t->cols; out->ld = t->ld;
    return AAE_OK;
}

// brings every deferred update up to date so that the arena views of ENC_W1T and its optimiser
// state hold the values an eager implementation would
int aae_sync(aae_handle h, void* stream) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    TRY(join_deferred(h, S(stream)));
    return lazy_flush(h, S(stream));
}

// ---- state_dict import / export ----------------------------------------------------------
static int param_id(int net, int layer) {
    if (layer < 1 || layer > 3 || net < 0 || net > 2) return -1;
    static const int tab[3][3] = {{P_W1T, P_W2, P_W3}, {P_V1, P_V2, P_V3}, {P_D1, P_D2, P_D3}};
    return tab[net][layer - 1];
}

// host [out][in] (+ bias[out]) <-> device tensor `t` (+ bias tensor tb for enc.lin1)
static int put_linear(aae_handle h, int pid, const Ten& t, const Ten* tb, const float* w, const float* b) {
    std::vector<float> buf(t.floats(), 0.f);
    if (pid == P_W1T) {   // torch [h][N] -> item-major [N][h]
        const int N = h->N, hh = h->h;
        if (w) for (int o = 0; o < hh; ++o) for (int i = 0; i < N; ++i) buf[(size_t)i * t.ld + o] = w[(size_t)o * N + i];
        if (w) HIPCHK(hipMemcpy(t.p, buf.data(), buf.size() * 4, hipMemcpyHostToDevice));
        if (b && tb) {
            std::vector<float> bb(tb->floats(), 0.f);
            memcpy(bb.data(), b, sizeof(float) * hh);
            HIPCHK(hipMemcpy(tb->p, bb.data(), bb.size() * 4, hipMemcpyHostToDevice));
        }
        return AAE_OK;
    }
    const int out = (int)t.rows, in = (int)t.cols - 1;
    h->pt_ok[pid] = false;
    HIPCHK(hipMemcpy(buf.data(), t.p, buf.size() * 4, hipMemcpyDeviceToHost));
    for (int o = 0; o < out; ++o) {
        if (w) memcpy(&buf[(size_t)o * t.ld], &w[(size_t)o * in], sizeof(float) * in);
        if (b) buf[(size_t)o * t.ld + in] = b[o];
    }
    HIPCHK(hipMemcpy(t.p, buf.data(), buf.size() * 4, hipMemcpyHostToDevice));
    return AAE_OK;
}

static int get_linear(aae_handle h, int pid, const Ten& t, const Ten* tb, float* w, float* b) {
    std::vector<float> buf(t.floats());
    HIPCHK(hipMemcpy(buf.data(), t.p, buf.size() * 4, hipMemcpyDeviceToHost));
    if (pid == P_W1T) {
        const int N = h->N, hh = h->h;
        if (w) for (int o = 0; o < hh; ++o) for (int i = 0; i < N; ++i) w[(size_t)o * N + i] = buf[(size_t)i * t.ld + o];
        if (b && tb) {
            std::vector<float> bb(tb->floats());
            HIPCHK(hipMemcpy(bb.data(), tb->p, bb.size() * 4, hipMemcpyDeviceToHost));
            memcpy(b, bb.data(), sizeof(float) * hh);
        }
        return AAE_OK;
    }
    const int out = (int)t.rows, in = (int)t.cols - 1;
    for (int o = 0; o < out; ++o) {
        if (w) memcpy(&w[(size_t)o * in], &buf[(size_t)o * t.ld], sizeof(float) * in);
        if (b) b[o] = buf[(size_t)o * t.ld + in];
    }
    return AAE_OK;
}

int aae_load_linear(aae_handle h, int net, int layer, const float* w, const float* b) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    int pid = param_id(net, layer);
    if (pid < 0) return fail(AAE_EINVAL, "bad net/layer");
    TRY(join_host(h));
    TRY(lazy_flush(h, nullptr));
    HIPCHK(hipDeviceSynchronize());
    return put_linear(h, pid, h->P[pid], pid == P_W1T ? &h->P[P_B1] : nullptr, w, b);
}
int aae_store_linear(aae_handle h, int net, int layer, float* w, float* b) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    int pid = param_id(net, layer);
    if (pid < 0) return fail(AAE_EINVAL, "bad net/layer");
    TRY(join_host(h));
    TRY(lazy_flush(h, nullptr));
    HIPCHK(hipDeviceSynchronize());
    return get_linear(h, pid, h->P[pid], pid == P_W1T ? &h->P[P_B1] : nullptr, w, b);
}

static int adam_sel(int which, int layer, int* pid, int* set) {
    if (layer < 1 || layer > 3) return -1;
    switch (which) {
        case O_ENC: *pid = param_id(0, layer); *set = 0; return 0;
        case O_DEC: *pid = param_id(1, layer); *set = 0; return 0;
        case O_GEN: *pid = param_id(0, layer); *set = 1; return 0;
        case O_DISC: *pid = param_id(2, layer); *set = 0; return 0;
    }
    return -1;
}
int aae_load_adam(aae_handle h, int which, int layer, const float* m_w, const float* v_w, const float* m_b,
                  const float* v_b, int64_t step) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    int pid, set;
    if (adam_sel(which, layer, &pid, &set)) return fail(AAE_EINVAL, "bad optimiser/layer");
    TRY(join_host(h));
    TRY(lazy_flush(h, nullptr));
    HIPCHK(hipDeviceSynchronize());
    const Ten* mb = pid == P_W1T ? &h->M[set][P_B1] : nullptr;
    const Ten* vb = pid == P_W1T ? &h->V[set][P_B1] : nullptr;
    TRY(put_linear(h, pid, h->M[set][pid], mb, m_w, m_b));
    TRY(put_linear(h, pid, h->V[set][pid], vb, v_w, v_b));
    if (step >= 0) {
        OptScalars hs;
        HIPCHK(hipMemcpy(&hs, h->sc + which, sizeof(hs), hipMemcpyDeviceToHost));
        hs.t = step;
        HIPCHK(hipMemcpy(h->sc + which, &hs, sizeof(hs), hipMemcpyHostToDevice));
        if (which == O_ENC || which == O_GEN) {
            // enc_optim and gen_optim step together; the rng/lazy step counter follows them
            long long t = step;
            HIPCHK(hipMemcpy(h->step_ctr, &t, sizeof(t), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(fill_int_kernel, dim3(256), dim3(256), 0, 0, h->tsync, (size_t)h->N, (int)step);
            HIPCHK(hipDeviceSynchronize());
        }
    }
    return AAE_OK;
}
int aae_store_adam(aae_handle h, int which, int layer, float* m_w, float* v_w, float* m_b, float* v_b,
                   int64_t* step) {
    if (!h) return fail(AAE_EINVAL, "handle is NULL");
    int pid, set;
    if (adam_sel(which, layer, &pid, &set)) return fail(AAE_EINVAL, "bad optimiser/layer");
    TRY(join_host(h));
    TRY(lazy_flush(h, nullptr));
    HIPCHK(hipDeviceSynchronize());
    const Ten* mb = pid == P_W1T ? &h->M[set][P_B1] : nullptr;
    const Ten* vb = pid == P_W1T ? &h->V[set][P_B1] : nullptr;
    TRY(get_linear(h, pid, h->M[set][pid], mb, m_w, m_b));
    TRY(get_linear(h, pid, h->V[set][pid], vb, v_w, v_b));
    if (step) {
        OptScalars hs;
        HIPCHK(hipMemcpy(&hs, h->sc + which, sizeof(hs), hipMemcpyDeviceToHost));
        *step = hs.t;
    }
    return AAE_OK;
}

// ---- the step ----------------------------------------------------------------------------
// The aae_rng_inject handed to a phase stays in force for the later phases of the same step;
// its device buffers must stay valid until the step's kernels have run.
static void remember_inject(aae_model* m, const aae_rng_inject* inj, bool reset) {
    if (inj) m->inj = *inj;
    else if (reset) memset(&m->inj, 0, sizeof(m->inj));
}

static int ae_encode_impl(aae_handle m, const aae_batch* batch, const aae_rng_inject* inj, float* z_out, bool with_dec,
                          const float* cond_dev, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    TRY(set_batch(m, batch));
    remember_inject(m, inj, true);
    hipStream_t s = S(stream);
    TRY(join_deferred(m, s));       // the previous step's optimiser pass over DEC_V3 reads the step scalars and dh2
    m->hstep++;
    // the list of this batch's distinct items and their catch-up were built while the previous step ran
    const bool ahead = m->pf_built && m->pf_step == m->hstep && same_batch(m->pf_built_batch, *batch) && m->lazy;
    m->pf_built = false;
    if (ahead) { std::swap(m->mark, m->mark2); std::swap(m->ulist, m->ulist2); std::swap(m->ucount, m->ucount2); std::swap(m->stamp, m->stamp2); }
    // With the batch's list built ahead nothing sits between the step-opening bookkeeping and the first gather: it rides
    // in that launch (one launch floor, ~4.5 us, less per step)
    static const bool fold_ok = getenv("AAE_NO_FOLD_ADVANCE") == nullptr;
    const bool fold_advance = fold_ok && ahead && m->use_chain && !m->ext_first && m->noise_next == nullptr;
    if (!fold_advance)
    hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, m->sc, m->step_ctr, m->lazy ? m->tab : nullptr,
                       ahead ? (int*)nullptr : m->stamp, ahead ? (int*)nullptr : m->ucount, m->losses);
    m->dense_step = m->noise_next != nullptr;
    const float* noise = m->noise_next; m->noise_next = nullptr;
    if (m->dense_step) {
        if (!m->use_chain) return fail(AAE_ESTATE, "the dense noisy input needs the layer-chain kernels");
        // every row of W1T is read: all of them through the previous step (a scan of tsync after the first such step)
        if (m->lazy && m->cfg.optimizer == AAE_OPT_ADAM) {
            hipLaunchKernelGGL(w1_catchup_kernel, dim3(std::min(m->N, 8192)), dim3(256), 0, s, (const int*)nullptr,
                               (const int*)nullptr, m->N, m->tsync, m->P[P_W1T].p, m->M[0][P_W1T].p, m->V[0][P_W1T].p,
                               m->M[1][P_W1T].p, m->V[1][P_W1T].p, m->ldw1, m->h, m->tab, m->step_ctr, -1);
            LAUNCHCHK("w1_catchup all");
        }
        hipLaunchKernelGGL(uniq_items_kernel, dim3(m->rows, std::max(1, std::min(16, m->chunks / 16 + 1))), dim3(256), 0, s, m->bv,
                           m->mark, m->stamp, m->ulist, m->ucount);       // (the list later phases of the step expect)
        hipLaunchKernelGGL(dense_input_kernel, dim3(m->rows), dim3(1024), 0, s, m->bv, noise, m->noise_ld, m->N,
                           (int)m->cfg.normalize_inputs, m->Xn.p, m->ldn);
        LAUNCHCHK("dense_input");
        const int B = m->rows, h = m->h, N = m->N;
        int tiles = ((B + 63) / 64) * ((h + 63) / 64);
        int splits = std::max(1, std::min(m->max_slabs, 2048 / tiles));
        int kps = ((N + splits - 1) / splits + 63) / 64 * 64;
        splits = (N + kps - 1) / kps;
        GemmShape g{m->Xn.p, m->P[P_W1T].p, B, h, N, m->ldn, m->ldw1, kps};
        EpiSlab e; e.out = m->slabs.p; e.ld = m->ldh; e.slab_stride = (size_t)m->R * m->ldh;
        (void)launch_gemm_mode<0, 0, true>(gmode(m), g, e, splits, s);
        LAUNCHCHK("dense first layer");
        DropSpec d1 = make_drop(m, 0, true, m->inj.masks_dev[0], nullptr, B, h, 0);
        hipLaunchKernelGGL(slab_reduce_fwd_kernel, dim3(grid1d((size_t)B * h)), dim3(256), 0, s, m->slabs.p, splits,
                           e.slab_stride, B, h, m->ldh, m->P[P_B1].p, m->a1.p, m->eh1.p, (int)m->cfg.activation, d1,
                           (uint64_t)m->cfg.seed, m->step_ctr);
        LAUNCHCHK("slab_reduce_fwd");
        m->dec_hidden_done = false; m->enc_bwd_done = false;
        TRY(chain_ae_forward(m, with_dec, cond_dev, z_out, s));
        m->phase = 1;
        if (m->pf_armed) m->pf_armed = false;
        return AAE_OK;
    }
    if (m->ext_first) {
        // the first layer is the caller's (aae_set_first_layer_external): AAE_T_ACT_A1 holds this batch's pre-activations
        if (!m->use_chain || m->vae) return fail(AAE_ESTATE, "an external first layer needs the layer-chain kernels (and no VAE mode)");
        m->dec_hidden_done = false; m->enc_bwd_done = false; m->pf_armed = false;
        TRY(chain_ae_forward(m, with_dec, cond_dev, z_out, s));
        m->phase = 1;
        return AAE_OK;
    }
    if (m->lazy && !ahead) TRY(lazy_prepare(m, -1, false, s));
    m->dec_hidden_done = false; m->enc_bwd_done = false;
    // Batches beyond one fused launch (112 rows): their tile buckets (the row-blocked output layer's entry lists and the first
    // layer's per-item update read them; one wide launch, 47 us at 512 rows x 100 k items alone, 100 us beside a streaming
    // GEMM) depend on the batch only - built on the side stream beside the list building, the gather and the forward
    // chain, as aae_first_layer_forward does for the item slices, instead of in front of their first reader.  (Up to 112
    // rows the builder rides in the step's first chain launch.)  The side stream is in order behind the previous step's
    // deferred launch, which waited for that step's output layer - the alternate bucket set's last readers are older.
    // (Not on the three-GEMM path: there the side stream holds the previous step's dV3 GEMM for most of this step.)
    {
        static const bool bk_ahead = getenv("AAE_NO_BUCKETS_AHEAD") == nullptr;
        if (bk_ahead && m->side && m->ev_bk && m->last_out_split && m->use_chain && m->rows > 16 * kMB && !m->buckets_valid && fused_decoder_applies(m)) {
            TRY(build_tile_buckets(m, m->side));
            HIPCHK(hipEventRecord(m->ev_bk, m->side));
            m->bk_pending = true;
        }
    }
    const bool pf = m->pf_armed && m->side && m->mark2 && m->lazy && m->use_chain;
    if (m->pf_armed && !pf) m->pf_armed = false;
    if (m->use_chain) {
        TRY(gather_first_layer(m, true, m->inj.masks_dev[0], 0, s, pf, fold_advance));
        if (pf) TRY(launch_prefetch(m));
        TRY(chain_ae_forward(m, with_dec, cond_dev, z_out, s));
        m->phase = 1;
        return AAE_OK;
    }
    TRY(encoder_forward(m, true, m->inj.masks_dev[0], m->inj.masks_dev[1], 0, 1, false, m->zc.p, m->ldc, s));
    // keep a copy of z for the encoder backward (condition plugins replace zc)
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)m->rows * m->c)), dim3(256), 0, s, m->zc.p, m->ldc,
                       m->zsave.p, m->ldz, m->rows, m->c, 1.0f);
    if (z_out)
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)m->rows * m->c)), dim3(256), 0, s, m->zc.p, m->ldc,
                           z_out, m->c, m->rows, m->c, 1.0f);
    LAUNCHCHK("ae_encode copies");
    m->phase = 1;
    return AAE_OK;
}

int aae_ae_encode(aae_handle m, const aae_batch* batch, const aae_rng_inject* inj, float* z_out, void* stream) {
    return ae_encode_impl(m, batch, inj, z_out, false, nullptr, stream);
}

int aae_ae_decode_backward(aae_handle m, const float* zc_dev, int64_t zc_ld, const aae_rng_inject* inj,
                           float* dzc_out, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->phase != 1) return fail(AAE_ESTATE, "aae_ae_decode_backward without aae_ae_encode");
    remember_inject(m, inj, false);
    hipStream_t s = S(stream);
    const int B = m->rows, N = m->N, h = m->h, cp = m->cp;
    if (zc_dev) TRY(stage_zc(m, zc_dev, zc_ld, B, s));
    const uint8_t* mk2 = m->inj.masks_dev[2];
    const uint8_t* mk3 = m->inj.masks_dev[3];
    if (m->only_output_layer) { /* ACT_DH2 is the input */ }
    else if (m->use_chain) { if (!m->dec_hidden_done) TRY(chain_dec_hidden(m, true, B, s)); }
    else TRY(decoder_hidden_forward(m, true, mk2, mk3, B, s));
    const float gscale = m->grad_scale / ((float)B * (float)N);
    DropSpec d1 = make_drop(m, 0, true, mk2, nullptr, B, h, 2);
    DropSpec d2 = make_drop(m, 1, true, mk3, nullptr, B, h, 3);
    // row blocks of the fused output layer: one launch covers at most 112 rows; larger batches (cfg.blocked_output) run as
    // nblk launches of the split form over equal row blocks
    const int nblk = m->have_batch ? row_blocks(m) : 1;
    const int Bb = (B + nblk - 1) / nblk;                        // rows per block (the last one may be shorter)
    const size_t fused_lds = m->bf16 ? dec_fused_bf16_lds_bytes(m->fused_nb ? m->fused_nb : 13) : dec_fused_lds_bytes(Bb, h);
    const float* chain_part = nullptr; size_t chain_stride = 0;
    if (fused_decoder_applies(m)) {
        // ---- fused path (dec_fused.h): logits, BCE, dV3 + dec_optim and dA2 in one persistent kernel
        const int ntiles = (N + kTI - 1) / kTI;
        if (m->bk_pending) {            // built on the side stream while this step's forward ran (aae_first_layer_forward)
            HIPCHK(hipStreamWaitEvent(s, m->ev_bk, 0));
            m->bk_pending = false;
        }
        if (!m->buckets_valid) TRY(build_tile_buckets(m, s));   // (else: the extra workgroup of this step's first chain launch did)
        DecFusedArgs fa;
        fa.dh2 = m->dh2.p; fa.ldh = m->ldh;
        fa.V3a = m->P[P_V3].p; fa.M = m->M[0][P_V3].p; fa.V = m->V[0][P_V3].p; fa.ldv = m->ldh;
        fa.gradV3 = m->cfg.grad_mode == AAE_GRAD_EXPORT ? m->Gr[P_V3].p : nullptr;
        fa.N = N; fa.B = B; fa.h = h; fa.gscale = gscale;
        fa.te.start = m->tstart; fa.te.eb = m->teb; fa.te.en = m->ten; fa.te.ev = m->tev;
        fa.slabs = m->slabs.p; fa.slab_stride = (size_t)(nblk > 1 ? B : std::min(m->R, 16 * kMB)) * m->ldh; fa.ld_slab = m->ldh;
        fa.partials = m->bce_partials; fa.sc = m->sc + O_DEC;
        fa.erow0 = 0; fa.acc = nullptr; fa.nblk = 1; fa.Bb = B;
        { const char* e = getenv("AAE_DEC_SKIP"); fa.dbg_skip = e ? atoi(e) : 0; }
        static const bool want_ts = getenv("AAE_DEC_TS") != nullptr;        // debug: phase timeline of one tile
        static unsigned long long* ts_dev = nullptr;
        fa.ts = nullptr;
        if (want_ts) {
            if (!ts_dev && hipMalloc(&ts_dev, 128 * sizeof(unsigned long long)) != hipSuccess) return fail(AAE_EHIP, "ts alloc");
            fa.ts = ts_dev;
        }
        const int grid = std::min(ntiles, m->n_cu);
        fa.Gt = m->Gt;
        int n_loss_partials = grid, crit_slabs = grid;
        // The split pays when the deferred half FITS beside the rest of the step and the layer is big enough to matter:
        // below ~2 tiles per CU the two event hops cost more than the optimiser pass they hide (C1, N = 1 k: 0.173 -> 0.184
        // ms/step), and beyond ~32 M parameters the deferred launch on half the CUs outlasts the rest of the step and the
        // next step waits for it (one rank's C5 share, 442 M parameters: 3.5 -> 4.8 ms/step) - both take the single launch.
        const bool split_fits = nblk > 1 || m->split_any || (ntiles >= 2 * m->n_cu && (size_t)N * m->ldh <= ((size_t)32 << 20));
        // (AAE_DEC_TS: the timeline of the single launch - or, AAE_DEC_TS=x3, of the split form's critical launch dec_crit_x3.h)
        static const bool ts_x3 = want_ts && strcmp(getenv("AAE_DEC_TS"), "x3") == 0;
        static const bool ts_obk = want_ts && strcmp(getenv("AAE_DEC_TS"), "obk") == 0;
        if (m->split_ok && m->split_wgs > 0 && split_fits && fa.gradV3 == nullptr && (!want_ts || ((ts_x3 || ts_obk) && m->x3_ok && !m->bf16)) && (fa.dbg_skip & ~(256 | 0xF000)) == 0) {
            // ---- split form: the critical launch(es) here, the optimiser launch(es) on the side stream behind the rest of
            // the step.  nblk > 1: one critical launch per row block (each with its block of dh2 in LDS; dA2 rows, loss
            // partials and stored dL/dlogits tiles of its own), then per row block one deferred launch that adds its dV3
            // to the partial of the blocks before it - the last one runs the optimiser.
            if (m->opt_pending) TRY(join_deferred(m, s));   // (never: every step-opening entry point joins)
            auto block_args = [&](int r) {
                DecFusedArgs b = fa;
                const int r0 = r * Bb;
                b.B = std::min(Bb, B - r0); b.erow0 = r0;
                b.dh2 = fa.dh2 + (size_t)r0 * m->ldh;
                b.slabs = fa.slabs + (size_t)r0 * m->ldh;
                b.partials = fa.partials + (size_t)r * grid;
                b.Gt = fa.Gt + (size_t)r * ntiles * Bb * kTI;
                return b;
            };
            // nblk > 1: ONE critical launch for all row blocks - workgroup w works on block w % nblk with its block of dh2
            // in LDS and takes every (grid / nblk)-th tile (dec_fused.h); 8 launches of 1.5 tile rounds each (-> 2, plus an
            // 84 KB prologue per workgroup and launch) cost 8 x 26.5 us on a 12.5 k-item slice, one launch of 12.2 rounds
            // what the 100-row step's critical launch costs
            const int wgs = nblk > 1 ? std::max(1, m->n_cu / nblk) : grid;
            const int crit_grid = nblk > 1 ? wgs * nblk : grid;
            n_loss_partials = crit_grid;
            crit_slabs = wgs;
            {
                // "this launch is done" rides on the kernel's own completion signal (a hipEventRecord behind the launch is a
                // marker packet the next kernel of the stream waits for: +30 us per step); when the launch is being timed,
                // the timing pair's stop event doubles as that event.
                DecFusedArgs b = fa;
                b.nblk = nblk; b.Bb = Bb;
                const int grid = crit_grid;
                const int r = nblk - 1;
                hipEvent_t start = nullptr, stop = r == nblk - 1 ? m->ev_crit : nullptr;
                (void)prof_pair(m, AAE_K_DEC_CRIT, &start, &stop);
                if (m->bf16) switch (m->fused_nb) {
                    case 4: hipExtLaunchKernelGGL((dec_fused_bf16_kernel<4, kDecCrit>), dim3(grid), dim3(kBT), (uint32_t)fused_lds, s, start, stop, 0, b); break;
                    case 7: hipExtLaunchKernelGGL((dec_fused_bf16_kernel<7, kDecCrit>), dim3(grid), dim3(kBT), (uint32_t)fused_lds, s, start, stop, 0, b); break;
                    default: hipExtLaunchKernelGGL((dec_fused_bf16_kernel<13, kDecCrit>), dim3(grid), dim3(kBT), (uint32_t)fused_lds, s, start, stop, 0, b); break;
                } else if (m->x3_ok) {
                    const uint32_t lds3 = (uint32_t)dec_crit_x3_lds_bytes(m->fused_nb);
                    switch (m->fused_nb) {
                    case 4: hipExtLaunchKernelGGL((dec_crit_x3_kernel<4>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); break;
                    case 7: hipExtLaunchKernelGGL((dec_crit_x3_kernel<7>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); break;
                    default: hipExtLaunchKernelGGL((dec_crit_x3_kernel<13>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); break;
                    }
                } else switch (m->fused_nb) {
                    case 4: hipExtLaunchKernelGGL((dec_fused_kernel<4, kDecCrit>), dim3(grid), dim3(kNT), (uint32_t)fused_lds, s, start, stop, 0, b); break;
                    case 7: hipExtLaunchKernelGGL((dec_fused_kernel<7, kDecCrit>), dim3(grid), dim3(kNT), (uint32_t)fused_lds, s, start, stop, 0, b); break;
                    default: hipExtLaunchKernelGGL((dec_fused_kernel<13, kDecCrit>), dim3(grid), dim3(kNT), (uint32_t)fused_lds, s, start, stop, 0, b); break;
                }
                LAUNCHCHK("dec_fused (critical launch)");
                if (r == nblk - 1) HIPCHK(hipStreamWaitEvent(m->side, stop, 0));
            }
            const int g2 = std::min(ntiles, std::min(m->split_wgs, m->n_cu));
            // nblk > 1 and at most kOBT tiles per workgroup on the chip: the deferred half of every block in ONE launch
            // (dec_opt_blocks_kernel), else one launch per block with the dV3 partial going through Gacc
            static const bool no_obk = getenv("AAE_NO_OPT_BLOCKS") != nullptr;
            static const bool no_opt_x3 = getenv("AAE_NO_OPT_X3") != nullptr;
            const bool one_opt = nblk > 1 && !m->bf16 && !no_obk && ntiles <= kOBT * m->n_cu &&
                                 dec_opt_blocks_lds_bytes(Bb) <= 160 * 1024;
            // (r3) the same on the emulated product, any vocabulary size: dec_opt_blocks_x3_kernel (dec_crit_x3.h)
            static const bool no_obk_x3 = getenv("AAE_NO_OPT_BLOCKS_X3") != nullptr;
            if (nblk > 1 && !m->bf16 && m->x3_ok && !no_opt_x3 && !no_obk_x3 && !no_obk && m->dh2f.p) {
                DecFusedArgs b = fa;
                b.nblk = nblk; b.Bb = Bb;
                hipLaunchKernelGGL(dh2_frag_kernel, dim3((B + kXCH - 1) / kXCH, m->fused_nb), dim3(128), 0, m->side, m->dh2.p, m->ldh, B,
                                   reinterpret_cast<u32x4_t*>(m->dh2f.p));
                b.acc = m->dh2f.p;                      // (this kernel's reading of the field: the fragment image)
                // tile groups of at most kXBT tiles, the same number (+-1 tile) for every workgroup and round
                // (workgroups: 5/8 of the CUs - 512 rows x 100 k items 0.827 / 0.809 / 0.823 / 0.839 ms per step on 128 / 160 / 192 / 224;
                //  half of them on an item slice of 12.5 k items x 800 rows: 0.382 / 0.379 / 0.400 / 0.387 ms of per-rank compute on 96 / 128 / 160 / 192)
                static const int obk_env = getenv("AAE_OBK_WGS") ? atoi(getenv("AAE_OBK_WGS")) : 0;
                const int g3 = std::max(1, std::min(obk_env > 0 ? obk_env : (getenv("AAE_SPLIT_WGS") ? g2 : ntiles < 1024 ? m->n_cu / 2 : m->n_cu * 5 / 8), std::min(ntiles, m->n_cu)));
                const int rounds = (ntiles + g3 * kXBT - 1) / (g3 * kXBT);
                b.tpp = g3 * rounds;
                const uint32_t lds3 = (uint32_t)dec_opt_blocks_x3_lds_bytes();
                hipEvent_t start = nullptr, stop = nullptr;
                (void)prof_pair(m, AAE_K_DEC_OPT, &start, &stop);
                if (ts_obk && m->fused_nb == 13) {
                    hipExtLaunchKernelGGL((dec_opt_blocks_x3_kernel<13, true>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b);
                    hipStreamSynchronize(m->side);
                    unsigned long long t[128];
                    hipMemcpy(t, ts_dev, sizeof(t), hipMemcpyDeviceToHost);
                    for (int w = 0; w < 2; ++w) {
                        fprintf(stderr, "[dec_opt_blocks_x3 wave %d, steps 8..15, us: products | split | to the next barrier;  spare wave: - | split | requests | wait for the slot | to the next barrier]", w ? 12 : 0);
                        for (int q = 0; q < 8; ++q) {
                            const unsigned long long* u = t + 64 * w + 4 * q;
                            if (w == 0) fprintf(stderr, "  %.2f %.2f %.2f", (u[1] - u[0]) * 0.01, (u[2] - u[1]) * 0.01, q < 7 ? ((double)u[4] - (double)u[2]) * 0.01 : 0.0);
                            else fprintf(stderr, "  %.2f %.2f %.2f %.2f", (u[1] - u[0]) * 0.01, (u[2] - u[1]) * 0.01, (u[3] - u[2]) * 0.01, q < 7 ? ((double)u[4] - (double)u[3]) * 0.01 : 0.0);
                        }
                        fprintf(stderr, "\n");
                    }
                    for (int k = 0; k < 2; ++k) {
                        const unsigned long long* u = t + (k ? 96 : 32);
                        fprintf(stderr, "[dec_opt_blocks_x3 step %d: every wave's arrival at the step's closing barrier, us after wave 0 finished its products]", k ? 12 : 9);
                        for (int w = 0; w < 16; ++w) fprintf(stderr, " %.2f", ((double)u[w] - (double)u[16]) * 0.01);
                        fprintf(stderr, "\n");
                    }
                } else
                switch (m->fused_nb) {
                    case 4: hipExtLaunchKernelGGL((dec_opt_blocks_x3_kernel<4>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b); break;
                    case 7: hipExtLaunchKernelGGL((dec_opt_blocks_x3_kernel<7>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b); break;
                    default: hipExtLaunchKernelGGL((dec_opt_blocks_x3_kernel<13>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b); break;
                }
                LAUNCHCHK("dec_opt_blocks_x3");
            } else if (one_opt) {
                DecFusedArgs b = fa;
                b.nblk = nblk; b.Bb = Bb;
                const int g3 = std::max(std::min(g2, ntiles), (ntiles + kOBT - 1) / kOBT);
                const uint32_t lds3 = (uint32_t)dec_opt_blocks_lds_bytes(Bb);
                hipEvent_t start = nullptr, stop = nullptr;
                (void)prof_pair(m, AAE_K_DEC_OPT, &start, &stop);
                switch (m->fused_nb) {
                    case 4: hipExtLaunchKernelGGL((dec_opt_blocks_kernel<4>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b); break;
                    case 7: hipExtLaunchKernelGGL((dec_opt_blocks_kernel<7>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b); break;
                    default: hipExtLaunchKernelGGL((dec_opt_blocks_kernel<13>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b); break;
                }
                LAUNCHCHK("dec_opt_blocks");
            } else
            for (int r = 0; r < nblk; ++r) {
                DecFusedArgs b = block_args(r);
                if (nblk > 1) { b.acc = m->Gacc.p; b.gradV3 = r == nblk - 1 ? nullptr : m->Gacc.p; }
                hipEvent_t start = nullptr, stop = nullptr;
                (void)prof_pair(m, AAE_K_DEC_OPT, &start, &stop);
                if (m->bf16) switch (m->fused_nb) {
                    case 4: hipExtLaunchKernelGGL((dec_fused_bf16_kernel<4, kDecOpt>), dim3(g2), dim3(kBT), (uint32_t)fused_lds, m->side, start, stop, 0, b); break;
                    case 7: hipExtLaunchKernelGGL((dec_fused_bf16_kernel<7, kDecOpt>), dim3(g2), dim3(kBT), (uint32_t)fused_lds, m->side, start, stop, 0, b); break;
                    default: hipExtLaunchKernelGGL((dec_fused_bf16_kernel<13, kDecOpt>), dim3(g2), dim3(kBT), (uint32_t)fused_lds, m->side, start, stop, 0, b); break;
                } else if (r == 0 && nblk == 1 && m->x3_ok && !no_opt_x3) {
                    // (the 3-term bf16 emulation of dV3 = G^T dh2, dec_crit_x3.h; AAE_NO_OPT_X3: the fp32 matrix pipe)
                    const uint32_t lds3 = (uint32_t)dec_opt_x3_lds_bytes();
                    switch (m->fused_nb) {
                    case 4: hipExtLaunchKernelGGL((dec_opt_x3_kernel<4>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); break;
                    case 7: hipExtLaunchKernelGGL((dec_opt_x3_kernel<7>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); break;
                    default: hipExtLaunchKernelGGL((dec_opt_x3_kernel<13>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); break;
                    }
                } else if (r == 0) switch (m->fused_nb) {
                    case 4: hipExtLaunchKernelGGL((dec_fused_kernel<4, kDecOpt>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); break;
                    case 7: hipExtLaunchKernelGGL((dec_fused_kernel<7, kDecOpt>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); break;
                    default: hipExtLaunchKernelGGL((dec_fused_kernel<13, kDecOpt>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); break;
                } else switch (m->fused_nb) {
                    case 4: hipExtLaunchKernelGGL((dec_fused_kernel<4, kDecOptAcc>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); break;
                    case 7: hipExtLaunchKernelGGL((dec_fused_kernel<7, kDecOptAcc>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); break;
                    default: hipExtLaunchKernelGGL((dec_fused_kernel<13, kDecOptAcc>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); break;
                }
                LAUNCHCHK("dec_fused (optimiser launch)");
            }
            TRY(side_done(m, m->ev_opt));
            m->opt_pending = true;
            m->last_out_split = true; m->side_ordered = true;
            // an item slice's next batch (named ahead): its distinct items and their deferred-Adam catch-up behind the
            // deferred launch on the same stream (ordered behind this step's head by ev_crit; rows of the running batch
            // are skipped there, the step's own updates bring them to the same step)
            if (m->only_output_layer && m->pf_armed && m->mark2 && m->lazy) TRY(launch_prefetch(m, false));
        } else
        {
            m->last_out_split = false; m->side_ordered = false;
            ProfScope ps(m, AAE_K_DEC_FUSED, s);
            if (m->bf16) switch (m->fused_nb) {
                case 4: hipLaunchKernelGGL(dec_fused_bf16_kernel<4>, dim3(grid), dim3(kBT), fused_lds, s, fa); break;
                case 7: hipLaunchKernelGGL(dec_fused_bf16_kernel<7>, dim3(grid), dim3(kBT), fused_lds, s, fa); break;
                default: hipLaunchKernelGGL(dec_fused_bf16_kernel<13>, dim3(grid), dim3(kBT), fused_lds, s, fa); break;
            } else switch (m->fused_nb) {
                case 4: hipLaunchKernelGGL(dec_fused_kernel<4>, dim3(grid), dim3(kNT), fused_lds, s, fa); break;
                case 7: hipLaunchKernelGGL(dec_fused_kernel<7>, dim3(grid), dim3(kNT), fused_lds, s, fa); break;
                default: hipLaunchKernelGGL(dec_fused_kernel<13>, dim3(grid), dim3(kNT), fused_lds, s, fa); break;
            }
        }
        LAUNCHCHK("dec_fused");
        if (want_ts) {
            unsigned long long t[128];
            hipStreamSynchronize(s);
            hipMemcpy(t, ts_dev, sizeof(t), hipMemcpyDeviceToHost);
            if (m->last_out_split && ts_obk) { /* printed at the launch */ }
            else if (m->last_out_split)
                fprintf(stderr, "[dec_crit_x3 tile 5] barrier=%.2f S0=%.2f GEMM1=%.2f BCE=%.2f GEMM3=%.2f | wg 0: prologue=%.2f loop=%.2f (%llu tiles, %.2f each) epilogue=%.2f us\n",
                        (t[14] - t[0]) * 0.01, (t[1] - t[14]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01,
                        (t[11] - t[10]) * 0.01, (t[7] - t[11]) * 0.01, t[13], (t[7] - t[11]) * 0.01 / (double)(t[13] ? t[13] : 1),
                        (t[12] - t[7]) * 0.01);
            else {
            if (m->bf16)
                for (int k = 0; k < 5; ++k) {
                    fprintf(stderr, "[dec_fused_bf16 arrivals at barrier %d, us after the unit's start]", k);
                    for (int w = 0; w < 16; ++w) fprintf(stderr, " %.2f", ((double)t[16 + 16 * k + w] - (double)t[0]) * 0.01);
                    fprintf(stderr, "\n");
                }
            fprintf(stderr, "[dec_fused tile 5] S0=%.2f GEMM1+BCE0=%.2f entries=%.2f GEMM2+GEMM3=%.2f S5=%.2f | tile=%.2f us, %.0f shader clocks -> %.2f GHz\n",
                    (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01,
                    (t[6] - t[4]) * 0.01, (t[6] - t[0]) * 0.01, (double)(t[9] - t[8]),
                    (double)(t[9] - t[8]) / ((t[6] - t[0]) * 10.0));
            if (m->bf16) fprintf(stderr, "[dec_fused_bf16 S0] barrier A=%.2f work=%.2f barrier B=%.2f us\n", (t[14] - t[0]) * 0.01, (t[15] - t[14]) * 0.01, (t[1] - t[15]) * 0.01);
            fprintf(stderr, "[dec_fused wg 0] prologue=%.2f loop=%.2f (%llu tiles, %.2f each) epilogue=%.2f us\n",
                    (t[11] - t[10]) * 0.01, (t[7] - t[11]) * 0.01, t[13], (t[7] - t[11]) * 0.01 / (double)(t[13] ? t[13] : 1),
                    (t[12] - t[7]) * 0.01);
            }
        }
        // 256+ slabs -> 16 partial slabs (stored behind the per-workgroup ones) -> sum + act'/dropout; the same
        // launch reduces the per-workgroup loss partials
        float* part = m->slabs.p + (size_t)304 * fa.slab_stride;
        const size_t n4 = (size_t)B * m->ldh / 4;
        if (m->only_output_layer && crit_slabs <= 64) {
            // (row blocks in one launch: 256 / nblk slabs - one pass sums them straight into dL/d(dh2), with the loss)
            hipLaunchKernelGGL(slab_partial_kernel, dim3((unsigned)((n4 + 255) / 256), 1), dim3(256), 0, s, m->slabs.p, crit_slabs,
                               fa.slab_stride, n4, m->da2.p, (size_t)0, m->bce_partials, n_loss_partials,
                               1.0f / ((float)B * (float)N), m->losses, 0);
            LAUNCHCHK("slabs -> da2");
            m->phase = 2;
            return AAE_OK;
        }
        hipLaunchKernelGGL(slab_partial_kernel, dim3((unsigned)((n4 + 255) / 256), 16), dim3(256), 0, s, m->slabs.p, crit_slabs,
                           fa.slab_stride, n4, part, fa.slab_stride, m->bce_partials, n_loss_partials,
                           1.0f / ((float)B * (float)N), m->losses, 0);
        if (m->only_output_layer) {
            hipLaunchKernelGGL(slab_partial_kernel, dim3((unsigned)((n4 + 255) / 256), 1), dim3(256), 0, s, part, 16,
                               fa.slab_stride, n4, m->da2.p, (size_t)0, (const float*)nullptr, 0, 0.f, m->losses, 0);
            LAUNCHCHK("slab_partial -> da2");
            m->phase = 2;
            return AAE_OK;
        }
        if (m->use_chain) {
            chain_part = part; chain_stride = fa.slab_stride;
        } else {
            hipLaunchKernelGGL(slab_reduce_actbwd_kernel, dim3(grid1d((size_t)B * h, 64)), dim3(64), 0, s, part, 16,
                               fa.slab_stride, B, h, m->ldh, m->dh2.p, m->ldh, m->gb0.p, m->cfg.activation, d2,
                               m->cfg.seed, m->step_ctr);
            LAUNCHCHK("slab_reduce");
        }
    } else {
    // ---- unfused path: output layer + BCE: G = dL/dlogits [B][N]
    {
        EpiBce e; e.G = m->G.p; e.ldg = m->ldn; e.gscale = gscale; e.partials = m->bce_partials;
        {
            ProfScope ps(m, AAE_K_DEC_BCE_FWD, s);
            TRY(linear_fwd(m->dh2.p, m->ldh, B, m->P[P_V3], e, s, gmode(m)));
        }
        hipLaunchKernelGGL(bce_fixup_kernel, dim3(B, m->chunks), dim3(256), 0, s, m->bv, m->dh2.p, m->ldh, m->P[P_V3].p, m->ldh,
                           h + 1, m->G.p, m->ldn, gscale, m->fix_partials);
        LAUNCHCHK("bce_fixup");
        const int ts = m->P[P_V3].rows > 4096 ? 64 : 32;   // tile edge linear_fwd picks for this layer
        TRY(finalize_bce_loss(m, ((N + ts - 1) / ts) * ((B + ts - 1) / ts), s));
    }
    // dA2 = G * V3 (K = N items, split-K slabs), then back through act2/drop2
    {
        int tiles = ((B + 63) / 64) * ((h + 63) / 64);
        int splits = std::max(1, std::min(m->max_slabs, 2048 / tiles));
        int kps = ((N + splits - 1) / splits + 63) / 64 * 64;
        splits = (N + kps - 1) / kps;
        GemmShape g{m->G.p, m->P[P_V3].p, B, h, N, m->ldn, m->ldh, kps};
        EpiSlab e; e.out = m->slabs.p; e.ld = m->ldh; e.slab_stride = (size_t)m->R * m->ldh;
        {
            ProfScope ps(m, AAE_K_DEC_DA2, s);
            (void)launch_gemm_mode<0, 0, true>(gmode(m), g, e, splits, s);
        }
        LAUNCHCHK("dA2 gemm");
        if (m->only_output_layer) {
            const size_t n4 = (size_t)B * m->ldh / 4;
            hipLaunchKernelGGL(slab_partial_kernel, dim3((unsigned)((n4 + 255) / 256), 1), dim3(256), 0, s, m->slabs.p, splits,
                               e.slab_stride, n4, m->da2.p, (size_t)0, (const float*)nullptr, 0, 0.f, m->losses, 0);
            LAUNCHCHK("slabs -> da2");
        } else
        hipLaunchKernelGGL(slab_reduce_actbwd_kernel, dim3(grid1d((size_t)B * h)), dim3(256), 0, s, m->slabs.p, splits,
                           e.slab_stride, B, h, m->ldh, m->dh2.p, m->ldh, m->gb0.p, m->cfg.activation, d2, m->cfg.seed,
                           m->step_ctr);
        LAUNCHCHK("slab_reduce");
    }
    // dV3 = G^T * dh2 -> dec_optim on V3 (the 24 B/param streaming kernel).  Like the fused path's optimiser half
    // (section 3.2c) only the NEXT step reads its result: with the fused optimiser it goes to the handle's low-priority
    // side stream, behind the rest of the step (G and dh2 stay untouched until the next step's join).
    static const bool defer_dv3 = getenv("AAE_NO_DEFER_DV3") == nullptr;
    // (not for the item slices of the vocabulary-sharded scheme: there the background GEMM slowed the replica handle's
    // kernels by more than it saved - 0.496 -> 0.560 ms of per-rank compute at world 8, tools/vocab_rank_time.py)
    if (defer_dv3 && m->side && m->cfg.grad_mode == AAE_GRAD_FUSED && !m->bf16 && !m->only_output_layer) {
        HIPCHK(hipEventRecord(m->ev_crit, s));
        HIPCHK(hipStreamWaitEvent(m->side, m->ev_crit, 0));
        {
            ProfScope ps(m, AAE_K_DEC_DV3_ADAM, m->side);
            TRY(linear_dw(m, m->G.p, m->ldn, B, m->dh2.p, m->ldh, P_V3, O_DEC, m->side));
        }
        TRY(side_done(m, m->ev_opt));
        m->opt_pending = true;
        m->side_ordered = true;
    } else {
        m->side_ordered = false;
        ProfScope ps(m, AAE_K_DEC_DV3_ADAM, s);
        TRY(linear_dw(m, m->G.p, m->ldn, B, m->dh2.p, m->ldh, P_V3, O_DEC, s));
    }
    if (m->only_output_layer) { m->phase = 2; return AAE_OK; }
    }
    if (m->use_chain && m->vae_bwd && m->vae_cut) {
        // cut at the condition boundary: stop at dL/d(decoder input); fc3's weight gradient + optimiser here, the rest
        // of the backward pass comes with the caller's dL/dz (aae_vae_encoder_backward)
        TRY(chain_vae_backward_dec(m, chain_part, chain_stride, dzc_out, s));
        DwBuilder dw;
        dw.add(m, m->gb0.p, m->ldh, m->zc.p, m->ldc, B, P_V1, O_DEC);
        TRY(dw.launch(s));
        m->phase = 2;
        return AAE_OK;
    }
    if (m->use_chain && m->vae_bwd) {
        TRY(chain_vae_backward(m, chain_part, chain_stride, s));
        DwBuilder dw;
        dw.add(m, m->gb0.p, m->ldh, m->zc.p, m->ldc, B, P_V1, O_DEC);
        dw.add(m, m->gmulv.p, (int)m->gmulv.ld, m->eh1.p, m->ldh, B, P_W3, O_ENC);
        TRY(dw.add_first_layer(m, m->gb3.p, O_ENC, s)); m->w1_merged = true;
        TRY(dw.launch(s));
        m->phase = 2;
        return AAE_OK;
    }
    if (m->use_chain) {
        // decoder hidden backward (+ the encoder backward when called from aae_step) in one program,
        // then every small weight gradient + optimiser update in one grouped launch
        const bool enc_too = m->fuse_enc_bwd;
        TRY(chain_ae_backward(m, true, enc_too, chain_part, chain_stride, nullptr, 0, dzc_out, O_ENC, s));
        DwBuilder dw;
        dw.add(m, m->gb0.p, m->ldh, m->dh1.p, m->ldh, B, P_V2, O_DEC);
        dw.add(m, m->gb1.p, m->ldh, m->zc.p, m->ldc, B, P_V1, O_DEC);
        if (enc_too) {
            dw.add(m, m->ga3.p, m->ldz, m->eh2.p, m->ldh, B, P_W3, O_ENC);
            dw.add(m, m->gb2.p, m->ldh, m->eh1.p, m->ldh, B, P_W2, O_ENC);
            TRY(dw.add_first_layer(m, m->gb3.p, O_ENC, s)); m->w1_merged = true;
            m->enc_bwd_done = true;
        }
        TRY(dw.launch(s));
        m->phase = 2;
        return AAE_OK;
    }
    // lin2
    EpiActBwd b1; b1.out = m->gb1.p; b1.ld = m->ldh; b1.y = m->dh1.p; b1.ldy = m->ldh; b1.act = m->cfg.activation;
    b1.d = d1; b1.seed = m->cfg.seed; b1.step_ctr = m->step_ctr;
    TRY(linear_dx(m->gb0.p, m->ldh, B, m->P[P_V2], h, b1, s, gmode(m)));
    TRY(linear_dw(m, m->gb0.p, m->ldh, B, m->dh1.p, m->ldh, P_V2, O_DEC, s));
    // lin1
    EpiStore ez; ez.out = m->gzc.p; ez.ld = m->ldc;
    TRY(linear_dx(m->gb1.p, m->ldh, B, m->P[P_V1], cp, ez, s, gmode(m)));
    TRY(linear_dw(m, m->gb1.p, m->ldh, B, m->zc.p, m->ldc, P_V1, O_DEC, s));
    if (dzc_out) {
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)B * cp)), dim3(256), 0, s, m->gzc.p, m->ldc, dzc_out, cp,
                           B, cp, 1.0f);
        LAUNCHCHK("copy dzc");
    }
    m->phase = 2;
    return AAE_OK;
}

int aae_decoder_step(aae_handle m, const aae_batch* batch, const float* zin_dev, int64_t zin_ld,
                     const aae_rng_inject* inj, float* dzin_out, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (!zin_dev) return fail(AAE_EINVAL, "zin_dev is NULL");
    if (zin_ld < m->cp) return fail(AAE_EINVAL, "zin_ld < n_code + cond_inc");
    TRY(set_batch(m, batch));
    remember_inject(m, inj, true);
    hipStream_t s = S(stream);
    TRY(join_deferred(m, s));       // the previous step's optimiser pass over DEC_V3 reads the step scalars and dh2
    m->hstep++; m->pf_armed = false;
    hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, m->sc, m->step_ctr, m->lazy ? m->tab : nullptr,
                       m->stamp, m->ucount, m->losses);
    LAUNCHCHK("advance_step");
    m->dec_hidden_done = false; m->enc_bwd_done = false; m->fuse_enc_bwd = false;
    m->phase = 1;
    TRY(aae_ae_decode_backward(m, zin_dev, zin_ld, nullptr, dzin_out, stream));
    m->phase = 0;
    return AAE_OK;
}

// VAE.partial_fit (vae.py:147-186): loss = mean BCE + KL sum (vae.py:132-145), one Adam over all five Linears
int aae_vae_step(aae_handle m, const aae_batch* batch, const float* cond_dev, const float* eps_dev, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (!m->vae) return fail(AAE_ESTATE, "model was not created in VAE mode (cfg.model_kind = 3)");
    if (!m->use_chain) return fail(AAE_ESTATE, "VAE mode needs the layer-chain kernels");
    if (m->cfg.cond_inc > 0 && !cond_dev) return fail(AAE_EINVAL, "cond_inc > 0 needs cond_dev");
    if (m->cfg.rng_mode == AAE_RNG_INJECT && !eps_dev) return fail(AAE_EINVAL, "rng_mode inject needs eps_dev");
    TRY(set_batch(m, batch));
    remember_inject(m, nullptr, true);
    hipStream_t s = S(stream);
    TRY(join_deferred(m, s));
    m->hstep++; m->pf_armed = false;
    hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, m->sc, m->step_ctr, m->lazy ? m->tab : nullptr,
                       m->stamp, m->ucount, m->losses);
    LAUNCHCHK("advance_step");
    if (m->lazy) TRY(lazy_prepare(m, -1, false, s));
    TRY(gather_first_layer(m, false, nullptr, 0, s));                 // eh1 = act(fc1(normalize(x))), vae.py:111-113
    TRY(chain_vae_forward(m, cond_dev, eps_dev, m->rows, s));
    m->dec_hidden_done = true; m->enc_bwd_done = false; m->fuse_enc_bwd = false;
    m->vae_bwd = true; m->vae_cut = false; m->phase = 1;
    const int rc = aae_ae_decode_backward(m, nullptr, 0, nullptr, nullptr, stream);
    m->vae_bwd = false;
    if (rc != AAE_OK) return rc;
    TRY(encoder_first_layer_update(m, m->gb3.p, O_ENC, s, m->w1_merged));
    m->phase = 0;
    return AAE_OK;
}

// VAE.predict (vae.py:229-266): the same stochastic forward (the reference samples eps in eval mode too)
int aae_vae_predict(aae_handle m, const aae_batch* batch, const float* cond_dev, const float* eps_dev, float* out_dev,
                    int64_t out_ld, void* stream) {
    if (!m || !out_dev) return fail(AAE_EINVAL, "handle/out is NULL");
    if (!m->vae || !m->use_chain) return fail(AAE_ESTATE, "model was not created in VAE mode (cfg.model_kind = 3)");
    if (m->cfg.cond_inc > 0 && !cond_dev) return fail(AAE_EINVAL, "cond_inc > 0 needs cond_dev");
    if (m->cfg.rng_mode == AAE_RNG_INJECT && !eps_dev) return fail(AAE_EINVAL, "rng_mode inject needs eps_dev");
    if (out_ld < m->N || (out_ld & 3) || (reinterpret_cast<uintptr_t>(out_dev) & 15))
        return fail(AAE_EINVAL, "out_dev must be 16-byte aligned with out_ld >= n_items and out_ld % 4 == 0");
    TRY(set_batch(m, batch));
    hipStream_t s = S(stream);
    TRY(join_deferred(m, s));
    if (m->lazy) TRY(lazy_prepare(m, 0, true, s));
    TRY(gather_first_layer(m, false, nullptr, 0, s));
    TRY(chain_vae_forward(m, cond_dev, eps_dev, m->rows, s));
    EpiSigmoid e; e.out = out_dev; e.ld = (int)out_ld;
    TRY(linear_fwd(m->dh2.p, m->ldh, m->rows, m->P[P_V3], e, s, gmode(m)));
    m->phase = 0;
    return AAE_OK;
}

// The VAE step cut at the condition boundary (vae.py:120-130: `z = self.conditions.encode_impose(z, condition_data)`
// between reparametrize and decode), for condition plugins that run in the host framework:
//   aae_vae_encode            x -> fc1 -> (mu, logvar) -> z = mu + eps * exp(logvar / 2)     (train != 0 opens a step)
//   [host: zc = conditions.encode_impose(z, c)]
//   aae_vae_decode_backward   fc3 -> fc4 -> BCE, backward to dL/dzc, fc3 / fc4 updates
//   [host: backprop dzc through the conditions -> dz; conditions.step()]
//   aae_vae_encoder_backward  reparametrize' + KL gradient -> [fc21; fc22] -> fc1, their updates
int aae_vae_encode(aae_handle m, const aae_batch* batch, const float* eps_dev, float* z_out_dev, int32_t train, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (!m->vae || !m->use_chain) return fail(AAE_ESTATE, "model was not created in VAE mode (cfg.model_kind = 3)");
    if (m->cfg.rng_mode == AAE_RNG_INJECT && !eps_dev) return fail(AAE_EINVAL, "rng_mode inject needs eps_dev");
    TRY(set_batch(m, batch));
    remember_inject(m, nullptr, true);
    hipStream_t s = S(stream);
    TRY(join_deferred(m, s));
    if (train) {
        m->hstep++; m->pf_armed = false;
        hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, m->sc, m->step_ctr, m->lazy ? m->tab : nullptr,
                           m->stamp, m->ucount, m->losses);
        LAUNCHCHK("advance_step");
        if (m->lazy) TRY(lazy_prepare(m, -1, false, s));
    } else if (m->lazy) TRY(lazy_prepare(m, 0, true, s));
    TRY(gather_first_layer(m, false, nullptr, 0, s));
    TRY(chain_vae_encode(m, eps_dev, z_out_dev, m->rows, s));
    m->dec_hidden_done = false; m->enc_bwd_done = false; m->fuse_enc_bwd = false;
    m->phase = train ? 1 : 0; m->vae_cut = train != 0;
    return AAE_OK;
}

int aae_vae_decode_backward(aae_handle m, const float* zc_dev, int64_t zc_ld, float* dzc_out_dev, void* stream) {
    if (!m || !zc_dev) return fail(AAE_EINVAL, "NULL argument");
    if (!m->vae || m->phase != 1 || !m->vae_cut) return fail(AAE_ESTATE, "aae_vae_decode_backward without aae_vae_encode(train)");
    if (zc_ld < m->cp) return fail(AAE_EINVAL, "zc_ld < n_code + cond_inc");
    hipStream_t s = S(stream);
    TRY(stage_zc(m, zc_dev, zc_ld, m->rows, s));
    TRY(chain_vae_dec_hidden(m, m->rows, s));
    m->dec_hidden_done = true; m->vae_bwd = true;
    const int rc = aae_ae_decode_backward(m, nullptr, 0, nullptr, dzc_out_dev, stream);
    m->vae_bwd = false;
    return rc;
}

int aae_vae_encoder_backward(aae_handle m, const float* dz_dev, int64_t dz_ld, void* stream) {
    if (!m || !dz_dev) return fail(AAE_EINVAL, "NULL argument");
    if (!m->vae || m->phase != 2 || !m->vae_cut) return fail(AAE_ESTATE, "aae_vae_encoder_backward without aae_vae_decode_backward");
    if (dz_ld < m->c) return fail(AAE_EINVAL, "dz_ld < n_code");
    hipStream_t s = S(stream);
    const int B = m->rows;
    TRY(chain_vae_backward_enc(m, dz_dev, (int)dz_ld, s));
    DwBuilder dw;
    dw.add(m, m->gmulv.p, (int)m->gmulv.ld, m->eh1.p, m->ldh, B, P_W3, O_ENC);
    TRY(dw.add_first_layer(m, m->gb3.p, O_ENC, s)); m->w1_merged = true;
    TRY(dw.launch(s));
    TRY(encoder_first_layer_update(m, m->gb3.p, O_ENC, s, m->w1_merged));
    m->phase = 0; m->vae_cut = false;
    return AAE_OK;
}

int aae_ae_encoder_backward(aae_handle m, const float* dz_dev, int64_t dz_ld, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->phase != 2) return fail(AAE_ESTATE, "aae_ae_encoder_backward without aae_ae_decode_backward");
    hipStream_t s = S(stream);
    if (m->use_chain) {
        if (!m->enc_bwd_done) {
            TRY(chain_ae_backward(m, false, true, nullptr, 0, dz_dev, (int)dz_ld, nullptr, O_ENC, s));
            DwBuilder dw;
            dw.add(m, m->ga3.p, m->ldz, m->eh2.p, m->ldh, m->rows, P_W3, O_ENC);
            dw.add(m, m->gb2.p, m->ldh, m->eh1.p, m->ldh, m->rows, P_W2, O_ENC);
            TRY(dw.add_first_layer(m, ga1_ptr(m), O_ENC, s)); m->w1_merged = true;
            TRY(dw.launch(s));
        }
        if (!m->ext_first) TRY(encoder_first_layer_update(m, m->gb3.p, O_ENC, s, m->w1_merged));
        m->phase = 3;
        return AAE_OK;
    }
    const float* gz = dz_dev ? dz_dev : m->gzc.p;
    int ld = dz_dev ? (int)dz_ld : m->ldc;
    TRY(encoder_backward(m, gz, ld, m->zsave.p, m->ldz, m->inj.masks_dev[0], m->inj.masks_dev[1], 0, 1, O_ENC, s));
    m->phase = 3;
    return AAE_OK;
}

// ---- the ae phase cut at the decoder's output layer (vocabulary-sharded data parallelism) -----------------------
// aae_ae_forward: encoder + the decoder's hidden layers on this rank's documents; dh2 stays in AAE_T_ACT_DH2.
int aae_ae_forward(aae_handle m, const aae_batch* batch, const float* cond_dev, const aae_rng_inject* inj, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (!m->use_chain || m->vae) return fail(AAE_ESTATE, "aae_ae_forward needs the layer-chain kernels (and no VAE mode)");
    if (m->cfg.cond_inc > 0 && !cond_dev) return fail(AAE_EINVAL, "cond_inc > 0 needs cond_dev");
    hipStream_t s = S(stream);
    TRY(ae_encode_impl(m, batch, inj, nullptr, true, cond_dev, stream));
    if (m->cfg.cond_inc > 0) {
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)m->rows * m->cfg.cond_inc)), dim3(256), 0, s, cond_dev,
                           m->cfg.cond_inc, m->zc.p + m->c, m->ldc, m->rows, m->cfg.cond_inc, 1.0f);
        LAUNCHCHK("copy cond");
    }
    return AAE_OK;
}

// aae_output_layer_step: the decoder's output layer alone over the handle's items - logits from AAE_T_ACT_DH2, BCE
// against the batch, dV3 + dec_optim on V3 (or its gradient in export mode), dL/d(dh2) summed into AAE_T_ACT_DA2.
//   batch != NULL: a step of its own (a handle that owns a shard of the vocabulary: the caller filled ACT_DH2 with
//                  the hidden activations of batch->n_rows documents, e.g. by an all-gather);
//   batch == NULL: continues the step aae_ae_forward started on this handle.
int aae_output_layer_step(aae_handle m, const aae_batch* batch, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->vae) return fail(AAE_ESTATE, "aae_output_layer_step: not in VAE mode");
    hipStream_t s = S(stream);
    if (batch || m->opt_pending) TRY(join_deferred(m, s));      // (batch = NULL: a prefetch started by this step keeps running)
    if (batch) {
        TRY(set_batch(m, batch));
        remember_inject(m, nullptr, true);
        m->hstep++; m->pf_armed = false;
        hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, m->sc, m->step_ctr, m->lazy ? m->tab : nullptr,
                           m->stamp, m->ucount, m->losses);
        LAUNCHCHK("advance_step");
        m->enc_bwd_done = false; m->fuse_enc_bwd = false;
    } else if (m->phase != 1 || !m->dec_hidden_done) {
        return fail(AAE_ESTATE, "aae_output_layer_step(batch = NULL) without aae_ae_forward");
    }
    m->dec_hidden_done = true;
    m->phase = 1;
    m->only_output_layer = true;
    int rc = aae_ae_decode_backward(m, nullptr, 0, nullptr, nullptr, stream);
    m->only_output_layer = false;
    TRY(rc);
    m->phase = batch ? 0 : 2;
    return AAE_OK;
}

// aae_ae_backward: the rest of the ae phase on this rank's documents from dL/d(dh2) (dA2_dev [rows][ld = ACT_DH2's],
// NULL = AAE_T_ACT_DA2 of this handle): decoder hidden layers backward, encoder backward, their optimiser updates
// (or exported gradients).  Follows aae_ae_forward.
int aae_ae_backward(aae_handle m, const float* dA2_dev, int64_t dA2_ld, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (!m->use_chain || m->vae) return fail(AAE_ESTATE, "aae_ae_backward needs the layer-chain kernels (and no VAE mode)");
    if ((m->phase != 1 && m->phase != 2) || !m->dec_hidden_done) return fail(AAE_ESTATE, "aae_ae_backward without aae_ae_forward");
    if (dA2_dev && dA2_ld != m->ldh) return fail(AAE_EINVAL, "aae_ae_backward: dA2_ld must equal the leading dimension of AAE_T_ACT_DH2");
    hipStream_t s = S(stream);
    const int B = m->rows;
    TRY(chain_ae_backward(m, true, true, dA2_dev ? dA2_dev : m->da2.p, 0, nullptr, 0, nullptr, O_ENC, s, 1));
    DwBuilder dw;
    dw.add(m, m->gb0.p, m->ldh, m->dh1.p, m->ldh, B, P_V2, O_DEC);
    dw.add(m, m->gb1.p, m->ldh, m->zc.p, m->ldc, B, P_V1, O_DEC);
    dw.add(m, m->ga3.p, m->ldz, m->eh2.p, m->ldh, B, P_W3, O_ENC);
    dw.add(m, m->gb2.p, m->ldh, m->eh1.p, m->ldh, B, P_W2, O_ENC);
    TRY(dw.add_first_layer(m, ga1_ptr(m), O_ENC, s)); m->w1_merged = true;      // (external first layer: its bias blocks only)
    TRY(dw.launch(s));
    m->enc_bwd_done = true;
    m->phase = 2;
    return aae_ae_encoder_backward(m, nullptr, 0, stream);
}

// ---- the first encoder layer sharded over the vocabulary (with the decoder's output layer: both [n_items, n_hidden]
// matrices live with the owner of their item slice, the ranks exchange [global rows, n_hidden] activations) ----------
int aae_set_doc_l1(aae_handle m, const float* doc_l1_dev) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    m->doc_l1 = doc_l1_dev;
    return AAE_OK;
}

int aae_set_first_layer_external(aae_handle m, int on) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (on && (!m->use_chain || m->vae)) return fail(AAE_ESTATE, "an external first layer needs the layer-chain kernels (and no VAE mode)");
    m->ext_first = on != 0;
    return AAE_OK;
}

// This handle's share of the first layer's pre-activations for the batch: sum over ITS items of x[b][i] * enc.lin1[:, i]
// (+ bias_dev[n_hidden] when given: the bias stays with the replicas, exactly one share adds it) -> AAE_T_ACT_A1
// [rows][n_hidden].
//   batch != NULL: a new step of this handle (step scalars advance, the batch's rows of the deferred Adam are caught up);
//   batch == NULL: the running batch again with the weights as they are now (disc_step's Enc_eval after enc_optim).
int aae_first_layer_forward(aae_handle m, const aae_batch* batch, const float* bias_dev, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->vae || m->cfg.grad_mode != AAE_GRAD_FUSED) return fail(AAE_ESTATE, "aae_first_layer_forward: fused optimiser, no VAE mode");
    hipStream_t s = S(stream);
    bool pf = false, fold_advance = false;
    if (batch) {
        TRY(join_deferred(m, s));       // (batch = NULL: a deferred optimiser launch of the output layer keeps running - it
                                        //  touches dec.lin3, its moments, the stored dL/dlogits and dh2, nothing of this layer)
        TRY(set_batch(m, batch));
        remember_inject(m, nullptr, true);
        m->hstep++;
        // (as in aae_step: the list of this batch's distinct items and their catch-up were built while the previous step
        //  ran, if the caller named the batch with aae_prefetch_batch)
        const bool ahead = m->pf_built && m->pf_step == m->hstep && same_batch(m->pf_built_batch, *batch) && m->lazy;
        m->pf_built = false;
        if (ahead) { std::swap(m->mark, m->mark2); std::swap(m->ulist, m->ulist2); std::swap(m->ucount, m->ucount2); std::swap(m->stamp, m->stamp2); }
        static const bool fold_ok = getenv("AAE_NO_FOLD_ADVANCE") == nullptr;
        fold_advance = fold_ok && ahead;          // (as in aae_step: nothing between the bookkeeping and the gather)
        if (!fold_advance)
        hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, m->sc, m->step_ctr, m->lazy ? m->tab : nullptr,
                           ahead ? (int*)nullptr : m->stamp, ahead ? (int*)nullptr : m->ucount, m->losses);
        LAUNCHCHK("advance_step");
        if (m->lazy && !ahead) TRY(lazy_prepare(m, -1, false, s));
        m->enc_bwd_done = false; m->fuse_enc_bwd = false; m->dense_step = false;
        // The output layer's tile buckets depend on the batch only: they are built on the side stream beside this
        // handle's list building and gather (and the caller's forward pass) instead of in front of the critical launch.
        // The side stream is in order behind the last deferred launch, which waited for the last critical launch - the
        // last reader of the bucket arrays; without such a launch to order it the build stays where it was.
        static const bool bk_ahead = getenv("AAE_NO_BUCKETS_AHEAD") == nullptr;
        if (bk_ahead && m->side && m->ev_bk && m->last_out_split && fused_decoder_applies(m)) {
            TRY(build_tile_buckets(m, m->side));
            HIPCHK(hipEventRecord(m->ev_bk, m->side));
            m->bk_pending = true;
        }
        // (a batch named with aae_prefetch_batch stays armed: its list and catch-up are enqueued behind this step's
        //  deferred optimiser launch - aae_output_layer_step - where they need no mark on this stream; a mark riding on
        //  the gather below cost the stream more than the 16 us it moved away)
        if (m->pf_armed && !(m->side && m->mark2 && m->lazy)) m->pf_armed = false;
    } else if (!m->have_batch) {
        return fail(AAE_ESTATE, "aae_first_layer_forward(batch = NULL) without a running batch");
    }
    {
        ProfScope ps(m, AAE_K_ENC_GATHER, s);
        const size_t shm = (size_t)16 * r4(m->h) * sizeof(float);
        DropSpec none; memset(&none, 0, sizeof(none));
        hipExtLaunchKernelGGL(enc_gather_kernel, dim3(m->rows), dim3(1024), (uint32_t)shm, s, nullptr, pf ? m->ev_head : nullptr, 0,
                              m->bv, (const float*)m->P[P_W1T].p, m->ldw1, bias_dev, m->h, (int)m->cfg.normalize_inputs,
                              m->a1.p, (float*)nullptr, m->ldh, (int)m->cfg.activation, none, (uint64_t)m->cfg.seed,
                              (const long long*)m->step_ctr, m->rscale, m->doc_l1,
                              AdvanceJob{m->sc, m->step_ctr, m->lazy ? m->tab : nullptr, m->losses, fold_advance ? 1 : 0},
                              (long long)(fold_advance ? m->hstep : -1));
        LAUNCHCHK("enc_gather (partial)");
    }
    if (pf) TRY(launch_prefetch(m));
    if (batch) { m->phase = 1; m->dec_hidden_done = true; }       // aae_output_layer_step(batch = NULL) may follow on this handle
    return AAE_OK;
}

// The first layer's weight gradient from dL/d(a1) of the running batch (ga1_dev [rows][ld], NULL = AAE_T_ACT_GA1 of this
// handle; rows_per_block > 0: blocks of that many rows, block_stride floats apart - the ranks' packets of an all-gather
// read where they landed) restricted to this handle's items, and optimiser `which` (enc_optim 0 / gen_optim 2) on its rows.  (The bias is
// a small replicated parameter: its gradient is a column sum of the replicas' own dL/d(a1), aae_ae_backward / aae_gen_step
// export it with the other small layers'.)
int aae_first_layer_update(aae_handle m, const float* ga1_dev, int64_t ld, int32_t rows_per_block, int64_t block_stride,
                           int which, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->vae || m->cfg.grad_mode != AAE_GRAD_FUSED) return fail(AAE_ESTATE, "aae_first_layer_update: fused optimiser, no VAE mode");
    if (which != O_ENC && which != O_GEN) return fail(AAE_EINVAL, "which must be enc_optim (0) or gen_optim (2)");
    if (!m->have_batch) return fail(AAE_ESTATE, "aae_first_layer_update without a running batch");
    if (ga1_dev && ld != m->ldh) return fail(AAE_EINVAL, "aae_first_layer_update: ld must equal the leading dimension of AAE_T_ACT_GA1");
    if (rows_per_block < 0 || (rows_per_block > 0 && (!ga1_dev || block_stride < (int64_t)rows_per_block * ld)))
        return fail(AAE_EINVAL, "aae_first_layer_update: blocks need ga1_dev and block_stride >= rows_per_block * ld");
    hipStream_t s = S(stream);
    const float* ga1 = ga1_dev ? ga1_dev : m->gb3.p;
    return launch_w1_items(m, ga1, (int)rows_per_block, (size_t)block_stride, which, s);
}

// disc_step (aae.py:713-732)
int aae_disc_step(aae_handle m, const aae_rng_inject* inj, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->ae_only) return fail(AAE_ESTATE, "model was created as a plain autoencoder (no discriminator steps)");
    if (m->phase != 3) return fail(AAE_ESTATE, "aae_disc_step before the ae phases of the step");
    remember_inject(m, inj, false);
    hipStream_t s = S(stream);
    const int B = m->rows, h = m->h, cc = m->c;
    const aae_rng_inject& I = m->inj;
    const float pscale = m->cfg.has_prior_scale ? m->cfg.prior_scale : 1.0f;
    // ---- disc_step: z_real rows [0,B), z_fake = Enc_eval(X) rows [B,2B)
    if (m->cfg.rng_mode != AAE_RNG_DEVICE && !I.z_real_dev) return fail(AAE_EINVAL, "rng_mode=inject needs z_real_dev");
    if (!m->use_chain) {                 // (the layer-chain program draws / copies z_real itself: COP_PRIOR)
        if (m->cfg.rng_mode == AAE_RNG_DEVICE) {
            hipLaunchKernelGGL(prior_kernel, dim3(grid1d((size_t)B * cc)), dim3(256), 0, s, m->zin.p, m->ldz, B, cc,
                               m->cfg.prior, pscale, m->cfg.seed, m->step_ctr, m->rng_row0);
        } else {
            hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)B * cc)), dim3(256), 0, s, I.z_real_dev, cc, m->zin.p,
                               m->ldz, B, cc, pscale);
        }
        LAUNCHCHK("prior");
    }
    if (m->use_chain) {
        TRY(chain_disc_step(m, s));
        m->phase = 4;
        return AAE_OK;
    }
    TRY(encoder_forward(m, false, nullptr, nullptr, 0, 0, false, m->zin.p + (size_t)B * m->ldz, m->ldz, s));
    TRY(disc_forward(m, 2 * B, I.masks_dev[4], I.masks_dev[6], I.masks_dev[5], I.masks_dev[7], B, 4, 5, s));
    hipLaunchKernelGGL(adv_loss_kernel, dim3(1), dim3(256), 0, s, m->dout.p, 4, B, 0, m->grad_scale, m->ga3.p, 4,
                       m->losses, 1);
    LAUNCHCHK("adv_loss disc");
    {
        DropSpec d1 = make_drop(m, 0, true, I.masks_dev[4], I.masks_dev[6], B, h, 4);
        DropSpec d2 = make_drop(m, 1, true, I.masks_dev[5], I.masks_dev[7], B, h, 5);
        EpiActBwd b2; b2.out = m->gb0.p; b2.ld = m->ldh; b2.y = m->xh2.p; b2.ldy = m->ldh; b2.act = m->cfg.activation;
        b2.d = d2; b2.seed = m->cfg.seed; b2.step_ctr = m->step_ctr;
        TRY(linear_dx(m->ga3.p, 4, 2 * B, m->P[P_D3], h, b2, s));
        TRY(linear_dw(m, m->ga3.p, 4, 2 * B, m->xh2.p, m->ldh, P_D3, O_DISC, s));
        EpiActBwd b1 = b2; b1.out = m->gb1.p; b1.y = m->xh1.p; b1.d = d1;
        TRY(linear_dx(m->gb0.p, m->ldh, 2 * B, m->P[P_D2], h, b1, s, gmode(m)));
        TRY(linear_dw(m, m->gb0.p, m->ldh, 2 * B, m->xh1.p, m->ldh, P_D2, O_DISC, s));
        TRY(linear_dw(m, m->gb1.p, m->ldh, 2 * B, m->zin.p, m->ldz, P_D1, O_DISC, s));
    }
    m->phase = 4;
    return AAE_OK;
}

// gen_step (aae.py:734-743): Enc_train(X) - layer-1 pre-activations are unchanged since the
// disc_step forward, so m->a1 is re-used - then D on rows [0,B)
int aae_gen_step(aae_handle m, const aae_rng_inject* inj, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->phase != 4) return fail(AAE_ESTATE, "aae_gen_step before aae_disc_step");
    remember_inject(m, inj, false);
    hipStream_t s = S(stream);
    const int B = m->rows, h = m->h, cc = m->c;
    const aae_rng_inject& I = m->inj;
    if (m->use_chain) {
        TRY(chain_gen_step(m, s));
        m->phase = 0;
        return AAE_OK;
    }
    TRY(encoder_forward(m, true, I.masks_dev[8], I.masks_dev[9], 8, 9, true, m->zin.p, m->ldz, s));
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)B * cc)), dim3(256), 0, s, m->zin.p, m->ldz, m->zsave.p,
                       m->ldz, B, cc, 1.0f);
    TRY(disc_forward(m, B, I.masks_dev[10], nullptr, I.masks_dev[11], nullptr, B, 10, 11, s));
    hipLaunchKernelGGL(adv_loss_kernel, dim3(1), dim3(256), 0, s, m->dout.p, 4, B, 1, m->grad_scale, m->ga3.p, 4,
                       m->losses, 2);
    LAUNCHCHK("adv_loss gen");
    {
        DropSpec d1 = make_drop(m, 0, true, I.masks_dev[10], nullptr, B, h, 10);
        DropSpec d2 = make_drop(m, 1, true, I.masks_dev[11], nullptr, B, h, 11);
        EpiActBwd b2; b2.out = m->gb0.p; b2.ld = m->ldh; b2.y = m->xh2.p; b2.ldy = m->ldh; b2.act = m->cfg.activation;
        b2.d = d2; b2.seed = m->cfg.seed; b2.step_ctr = m->step_ctr;
        TRY(linear_dx(m->ga3.p, 4, B, m->P[P_D3], h, b2, s));
        EpiActBwd b1 = b2; b1.out = m->gb1.p; b1.y = m->xh1.p; b1.d = d1;
        TRY(linear_dx(m->gb0.p, m->ldh, B, m->P[P_D2], h, b1, s, gmode(m)));
        EpiStore ez; ez.out = m->gzc.p; ez.ld = m->ldc;
        TRY(linear_dx(m->gb1.p, m->ldh, B, m->P[P_D1], cc, ez, s, gmode(m)));
    }
    TRY(encoder_backward(m, m->gzc.p, m->ldc, m->zsave.p, m->ldz, I.masks_dev[8], I.masks_dev[9], 8, 9, O_GEN, s));
    m->phase = 0;
    return AAE_OK;
}

int aae_disc_gen(aae_handle m, const aae_rng_inject* inj, void* stream) {
    TRY(aae_disc_step(m, inj, stream));
    return aae_gen_step(m, nullptr, stream);
}

int aae_step(aae_handle m, const aae_batch* batch, const float* cond_dev, const aae_rng_inject* inj, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->ext_first) return fail(AAE_ESTATE, "aae_step: the first layer is external (aae_set_first_layer_external): drive the phases");
    if (m->cfg.cond_inc > 0 && !cond_dev) return fail(AAE_EINVAL, "cond_inc > 0 needs cond_dev");
    hipStream_t s = S(stream);
    TRY(ae_encode_impl(m, batch, inj, nullptr, true, cond_dev, stream));
    if (m->cfg.cond_inc > 0) {
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)m->rows * m->cfg.cond_inc)), dim3(256), 0, s, cond_dev,
                           m->cfg.cond_inc, m->zc.p + m->c, m->ldc, m->rows, m->cfg.cond_inc, 1.0f);
        LAUNCHCHK("copy cond");
    }
    m->fuse_enc_bwd = true;
    int rc = aae_ae_decode_backward(m, nullptr, 0, nullptr, nullptr, stream);
    m->fuse_enc_bwd = false;
    TRY(rc);
    TRY(aae_ae_encoder_backward(m, nullptr, 0, stream));
    if (!m->ae_only) TRY(aae_disc_gen(m, nullptr, stream));
    return AAE_OK;
}

int aae_read_losses(aae_handle m, float out[3], void* stream) {
    if (!m || !out) return fail(AAE_EINVAL, "handle/out is NULL");
    float tmp[4];
    HIPCHK(hipMemcpyAsync(tmp, m->losses, sizeof(tmp), hipMemcpyDeviceToHost, S(stream)));
    HIPCHK(hipStreamSynchronize(S(stream)));
    out[0] = tmp[0]; out[1] = tmp[1]; out[2] = tmp[2];
    return AAE_OK;
}

// ---- predict (aae.py:840-870) -----------------------------------------------------------
int aae_encode(aae_handle m, const aae_batch* batch, float* z_out, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    TRY(set_batch(m, batch));
    hipStream_t s = S(stream);
    TRY(join_deferred(m, s));
    if (m->lazy) TRY(lazy_prepare(m, 0, true, s));
    if (m->use_chain) {
        TRY(gather_first_layer(m, false, nullptr, 0, s));
        ChainBuilder cb(m, m->rows);
        chain_encoder_tail(m, cb, false, nullptr, 0, m->rows, nullptr, s);
        ChainOp& f = cb.add(cop(COP_FINAL_FWD, 2, 2, m->c)); f.aux = m->cfg.enc_final; cop_out(f, m->zc.p, m->ldc);
        if (z_out) { f.out2 = z_out; f.ldo2 = m->c; }
        TRY(launch_chain(m, cb, s));
        m->phase = 0;
        return AAE_OK;
    }
    TRY(encoder_forward(m, false, nullptr, nullptr, 0, 0, false, m->zc.p, m->ldc, s));
    if (z_out) {
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)m->rows * m->c)), dim3(256), 0, s, m->zc.p, m->ldc,
                           z_out, m->c, m->rows, m->c, 1.0f);
        LAUNCHCHK("copy z");
    }
    m->phase = 0;
    return AAE_OK;
}

int aae_decode(aae_handle m, const float* zc_dev, int64_t zc_ld, int32_t n_rows, float* out_dev, int64_t out_ld,
               void* stream) {
    if (!m || !out_dev) return fail(AAE_EINVAL, "handle/out is NULL");
    if (n_rows < 1 || n_rows > m->R) return fail(AAE_EINVAL, "n_rows outside [1, max_batch]");
    if (out_ld < m->N || (out_ld & 3) || (reinterpret_cast<uintptr_t>(out_dev) & 15))
        return fail(AAE_EINVAL, "out_dev must be 16-byte aligned with out_ld >= n_items and out_ld % 4 == 0");
    hipStream_t s = S(stream);
    TRY(join_deferred(m, s));
    if (zc_dev) TRY(stage_zc(m, zc_dev, zc_ld, n_rows, s));
    if (m->vae) TRY(chain_vae_dec_hidden(m, n_rows, s));       // (VAE: one hidden layer, fc3)
    else if (m->use_chain) TRY(chain_dec_hidden(m, false, n_rows, s));
    else TRY(decoder_hidden_forward(m, false, nullptr, nullptr, n_rows, s));
    EpiSigmoid e; e.out = out_dev; e.ld = (int)out_ld;
    TRY(linear_fwd(m->dh2.p, m->ldh, n_rows, m->P[P_V3], e, s, gmode(m)));
    return AAE_OK;
}

int aae_predict(aae_handle m, const aae_batch* batch, const float* cond_dev, float* out_dev, int64_t out_ld,
                void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->cfg.cond_inc > 0 && !cond_dev) return fail(AAE_EINVAL, "cond_inc > 0 needs cond_dev");
    TRY(aae_encode(m, batch, nullptr, stream));
    if (m->cfg.cond_inc > 0) {
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)m->rows * m->cfg.cond_inc)), dim3(256), 0, S(stream),
                           cond_dev, m->cfg.cond_inc, m->zc.p + m->c, m->ldc, m->rows, m->cfg.cond_inc, 1.0f);
        LAUNCHCHK("copy cond");
    }
    return aae_decode(m, nullptr, 0, m->rows, out_dev, out_ld, stream);
}

// predict + on-device remove_non_missing / argtopk (evaluation.py:183-199, 20-58): only the k best
// items per row (ids and min-max-scaled scores) leave the GPU
int aae_predict_topk(aae_handle m, const aae_batch* batch, const float* cond_dev, int32_t k, int32_t exclude_known,
                     int32_t* idx_out_dev, float* val_out_dev, void* stream) {
    if (!m || !idx_out_dev || !val_out_dev) return fail(AAE_EINVAL, "NULL argument");
    if (k < 1 || k > 32 || k > m->N) return fail(AAE_EINVAL, "k must be in [1, min(32, n_items)]");
    TRY(aae_predict(m, batch, cond_dev, m->G.p, m->ldn, stream));      // scores into the [rows][N] scratch
    hipStream_t s = S(stream);
    if (k <= 10)
        hipLaunchKernelGGL(topk_rows_kernel<10>, dim3(m->rows), dim3(256), 0, s, m->G.p, m->ldn, m->N, m->bv,
                           exclude_known, k, reinterpret_cast<int*>(idx_out_dev), val_out_dev);
    else if (k <= 20)
        hipLaunchKernelGGL(topk_rows_kernel<20>, dim3(m->rows), dim3(256), 0, s, m->G.p, m->ldn, m->N, m->bv,
                           exclude_known, k, reinterpret_cast<int*>(idx_out_dev), val_out_dev);
    else
        hipLaunchKernelGGL(topk_rows_kernel<32>, dim3(m->rows), dim3(256), 0, s, m->G.p, m->ldn, m->N, m->bv,
                           exclude_known, k, reinterpret_cast<int*>(idx_out_dev), val_out_dev);
    LAUNCHCHK("topk_rows");
    return AAE_OK;
}

// the same for a caller-built decoder input (code | imposed conditions of any plugin kind): the second half of predict
// (aae.py:855-866) + remove_non_missing / argtopk; `batch` names the input rows whose items are excluded
int aae_decode_topk(aae_handle m, const float* zc_dev, int64_t zc_ld, const aae_batch* batch, int32_t k,
                    int32_t exclude_known, int32_t* idx_out_dev, float* val_out_dev, void* stream) {
    if (!m || !zc_dev || !idx_out_dev || !val_out_dev) return fail(AAE_EINVAL, "NULL argument");
    if (k < 1 || k > 32 || k > m->N) return fail(AAE_EINVAL, "k must be in [1, min(32, n_items)]");
    if (zc_ld < m->cp) return fail(AAE_EINVAL, "zc_ld < n_code + cond_inc");
    TRY(set_batch(m, batch));
    TRY(aae_decode(m, zc_dev, zc_ld, m->rows, m->G.p, m->ldn, stream));   // scores into the [rows][N] scratch
    hipStream_t s = S(stream);
    if (k <= 10)
        hipLaunchKernelGGL(topk_rows_kernel<10>, dim3(m->rows), dim3(256), 0, s, m->G.p, m->ldn, m->N, m->bv,
                           exclude_known, k, reinterpret_cast<int*>(idx_out_dev), val_out_dev);
    else if (k <= 20)
        hipLaunchKernelGGL(topk_rows_kernel<20>, dim3(m->rows), dim3(256), 0, s, m->G.p, m->ldn, m->N, m->bv,
                           exclude_known, k, reinterpret_cast<int*>(idx_out_dev), val_out_dev);
    else
        hipLaunchKernelGGL(topk_rows_kernel<32>, dim3(m->rows), dim3(256), 0, s, m->G.p, m->ldn, m->N, m->bv,
                           exclude_known, k, reinterpret_cast<int*>(idx_out_dev), val_out_dev);
    LAUNCHCHK("topk_rows");
    m->phase = 0;
    return AAE_OK;
}

// ---- data parallel: row-sparse exchange of the first encoder layer's gradient -------------
// floats from the start of Gr[B1] to the end of Gr[W3] (adjacent in the arena, 256-byte gaps included)
static size_t enc_small_floats(const aae_model* m) {
    const Ten& a = m->Gr[P_B1]; const Ten& b = m->Gr[P_W3];
    return (b.off + b.floats() * sizeof(float) - a.off) / sizeof(float);
}

int aae_w1_export(aae_handle m, int32_t* hdr_dev, float* vals_dev, int32_t cap, void* stream) {
    if (!m || !hdr_dev || !vals_dev) return fail(AAE_EINVAL, "NULL argument");
    if (m->cfg.grad_mode != AAE_GRAD_EXPORT) return fail(AAE_ESTATE, "aae_w1_export needs grad_mode=export");
    if (cap < 1) return fail(AAE_EINVAL, "cap must be positive");
    hipStream_t s = S(stream);
    hipLaunchKernelGGL(w1_pack_kernel, dim3(std::min(cap, 4096)), dim3(256), 0, s, m->ulist, m->ucount, m->Gr[P_W1T].p,
                       m->ldw1, m->h, cap, reinterpret_cast<int*>(hdr_dev), vals_dev);
    LAUNCHCHK("w1_pack");
    // the encoder's small-layer gradients (b1, W2, W3: one contiguous arena span) ride behind the rows
    const size_t nsmall = enc_small_floats(m);
    HIPCHK(hipMemcpyAsync(vals_dev + (size_t)cap * m->h, m->Gr[P_B1].p, nsmall * sizeof(float), hipMemcpyDeviceToDevice, s));
    return AAE_OK;
}

int aae_w1_packet_floats(aae_handle m, int32_t cap, int64_t* hdr_words, int64_t* total_floats) {
    if (!m || !hdr_words || !total_floats) return fail(AAE_EINVAL, "NULL argument");
    if (m->cfg.grad_mode != AAE_GRAD_EXPORT) return fail(AAE_ESTATE, "needs grad_mode=export");
    *hdr_words = (1 + (int64_t)cap + 3) & ~(int64_t)3;
    *total_floats = *hdr_words + (int64_t)cap * m->h + (int64_t)enc_small_floats(m);
    return AAE_OK;
}

int aae_w1_import(aae_handle m, const int32_t* hdr_dev, const float* vals_dev, int32_t cap, int32_t n_peers,
                  int64_t peer_stride_bytes, int which, void* stream) {
    if (!m || !hdr_dev || !vals_dev) return fail(AAE_EINVAL, "NULL argument");
    if (m->cfg.grad_mode != AAE_GRAD_EXPORT) return fail(AAE_ESTATE, "aae_w1_import needs grad_mode=export");
    if (which != O_ENC && which != O_GEN) return fail(AAE_EINVAL, "which must be enc_optim (0) or gen_optim (2)");
    if (n_peers < 1 || n_peers > std::max(1, m->cfg.dp_world)) return fail(AAE_EINVAL, "n_peers exceeds cfg.dp_world");
    hipStream_t s = S(stream);
    hipLaunchKernelGGL(bump_stamp_kernel, dim3(1), dim3(1), 0, s, m->stamp, m->ucount);
    const size_t nsmall = enc_small_floats(m);
    const char* small0 = reinterpret_cast<const char*>(vals_dev + (size_t)cap * m->h);
    if (n_peers > 1 && m->pslot && !getenv("AAE_W1_SERIAL")) {
        // every peer in one launch each: slot map + union list, rank-ordered row sums, rank-ordered small-layer sums
        const int W = m->cfg.dp_world;
        hipLaunchKernelGGL(w1_map_kernel, dim3(std::max(1, std::min((cap + 255) / 256, 64)), n_peers), dim3(256), 0, s,
                           reinterpret_cast<const char*>(hdr_dev), (long long)peer_stride_bytes, W, m->pslot, m->ptag,
                           m->mark, m->stamp, m->ulist, m->ucount);
        hipLaunchKernelGGL(w1_sum_kernel, dim3(std::min(cap * n_peers, 8192)), dim3(256), 0, s,
                           reinterpret_cast<const char*>(vals_dev), (long long)peer_stride_bytes, m->h, n_peers, W, m->pslot,
                           m->ptag, m->stamp, m->ulist, m->ucount, m->Gr[P_W1T].p, m->ldw1);
        LAUNCHCHK("w1_map/sum");
        hipLaunchKernelGGL(accumulate_peers_kernel, dim3(grid1d(nsmall)), dim3(256), 0, s, m->Gr[P_B1].p, small0,
                           (long long)peer_stride_bytes, n_peers, nsmall);
        LAUNCHCHK("accumulate peers");
    } else {
        for (int p = 0; p < n_peers; ++p) {
            const char* hb = reinterpret_cast<const char*>(hdr_dev) + (size_t)p * peer_stride_bytes;
            const char* vb = reinterpret_cast<const char*>(vals_dev) + (size_t)p * peer_stride_bytes;
            hipLaunchKernelGGL(w1_unpack_kernel, dim3(std::min(cap, 4096)), dim3(256), 0, s, reinterpret_cast<const int*>(hb),
                               reinterpret_cast<const float*>(vb), m->h, m->Gr[P_W1T].p, m->ldw1, m->mark, m->stamp,
                               m->ulist, m->ucount);
        }
        LAUNCHCHK("w1_unpack");
        // small encoder layers: sum the peers' spans in rank order
        for (int p = 0; p < n_peers; ++p) {
            const float* src = reinterpret_cast<const float*>(small0 + (size_t)p * peer_stride_bytes);
            hipLaunchKernelGGL(accumulate_kernel, dim3(grid1d(nsmall)), dim3(256), 0, s, m->Gr[P_B1].p, src, nsmall,
                               p == 0 ? 1 : 0);
        }
        LAUNCHCHK("accumulate small");
    }
    {
        TRY(aae_apply_updates(m, which, stream));      // b1, W2, W3 (W1T is skipped there: sparse path below)
    }
    const int set = which == O_GEN ? 1 : 0;
    const int grid = std::min(m->cfg.max_nnz * std::max(1, m->cfg.dp_world), 8192);
    if (m->cfg.optimizer == AAE_OPT_ADAM) {
        hipLaunchKernelGGL(w1_catchup_kernel, dim3(grid), dim3(256), 0, s, m->ulist, m->ucount, m->N, m->tsync,
                           m->P[P_W1T].p, m->M[0][P_W1T].p, m->V[0][P_W1T].p, m->M[1][P_W1T].p, m->V[1][P_W1T].p,
                           m->ldw1, m->h, m->tab, m->step_ctr, -1);
        LAUNCHCHK("w1_catchup union");
    }
    hipLaunchKernelGGL(w1_sparse_adam_kernel, dim3(grid), dim3(256), 0, s, m->ulist, m->ucount, m->P[P_W1T].p,
                       m->M[set][P_W1T].p, m->V[set][P_W1T].p, m->Gr[P_W1T].p, m->ldw1, m->h, m->sc + which, m->tsync,
                       m->step_ctr, (which == O_GEN || m->ae_only) ? 1 : 0);
    LAUNCHCHK("w1_sparse_adam union");
    return AAE_OK;
}

// Adam/SGD of optimiser `which` on rows [row_begin, row_end) of parameter tensor `tensor_id` with
// a gradient shard supplied by the caller (reduce-scatter output): the sharded-optimiser half of
// reduce-scatter -> update 1/world of DEC_V3 -> all-gather.
int aae_apply_shard(aae_handle m, int tensor_id, int64_t row_begin, int64_t row_end, const float* grad_shard_dev,
                    int which, void* stream) {
    if (!m || !grad_shard_dev) return fail(AAE_EINVAL, "NULL argument");
    if (tensor_id < 0 || tensor_id >= NP || tensor_id == P_W1T) return fail(AAE_EINVAL, "bad tensor id");
    if (which < 0 || which > 3) return fail(AAE_EINVAL, "bad optimiser id");
    const Ten& P = m->P[tensor_id];
    if (row_begin < 0 || row_end > P.rows || row_begin >= row_end) return fail(AAE_EINVAL, "bad row range");
    const int set = which == O_GEN ? 1 : 0;
    const size_t off = (size_t)row_begin * P.ld, n4 = (size_t)(row_end - row_begin) * P.ld / 4;
    hipLaunchKernelGGL(adam_dense_kernel, dim3(grid1d(n4)), dim3(256), 0, S(stream), P.p + off,
                       m->M[set][tensor_id].p + off, m->V[set][tensor_id].p + off, const_cast<float*>(grad_shard_dev),
                       n4, m->sc + which, 0);
    LAUNCHCHK("adam shard");
    return AAE_OK;
}

// ---- data parallel: optimiser step on all-reduced gradients ------------------------------
int aae_apply_updates_except(aae_handle m, int which, int skip_tensor_id, void* stream);
int aae_apply_updates(aae_handle m, int which, void* stream) { return aae_apply_updates_except(m, which, -1, stream); }

int aae_apply_updates_except(aae_handle m, int which, int skip_tensor_id, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->cfg.grad_mode != AAE_GRAD_EXPORT) return fail(AAE_ESTATE, "aae_apply_updates needs grad_mode=export");
    if (which < 0 || which > 3) return fail(AAE_EINVAL, "bad optimiser id");
    hipStream_t s = S(stream);
    int lo = which == O_DEC ? P_V1 : which == O_DISC ? P_D1 : P_W1T;
    int hi = which == O_DEC ? P_V3 : which == O_DISC ? P_D3 : P_W3;
    const int set = which == O_GEN ? 1 : 0;
    // small tensors (every hidden layer) share one launch; a vocabulary-sized one (DEC_V3 when the caller does
    // not shard it) streams through the plain grid-stride kernel
    AdamGroup grp; grp.njobs = 0;
    unsigned blocks = 0;
    for (int pid = lo; pid <= hi; ++pid) {
        if (pid == P_W1T) continue;        // row-sparse: aae_w1_import applies it
        if (pid == skip_tensor_id) continue; // sharded by the caller: aae_apply_shard
        const size_t n4 = m->P[pid].floats() / 4;
        if (n4 <= (size_t)1 << 18 && grp.njobs < 8) {
            AdamJob& j = grp.jobs[grp.njobs++];
            j.p = m->P[pid].p; j.m = m->M[set][pid].p; j.v = m->V[set][pid].p; j.g = m->Gr[pid].p;
            j.n4 = (unsigned)n4; j.blk0 = blocks;
            j.w4 = w4_of(m, pid); j.ld = (int)m->P[pid].ld;
            j.sc = nullptr; j.npeers = 0; j.pstride = 0;
            blocks += (unsigned)((n4 + 255) / 256);
            continue;
        }
        hipLaunchKernelGGL(adam_dense_kernel, dim3(grid1d(n4)), dim3(256), 0, s, m->P[pid].p, m->M[set][pid].p,
                           m->V[set][pid].p, m->Gr[pid].p, n4, m->sc + which, 0);
        LAUNCHCHK("adam_dense");
        m->pt_ok[pid] = false;
    }
    if (grp.njobs) {
        hipLaunchKernelGGL(adam_group_kernel, dim3(blocks), dim3(256), 0, s, grp, m->sc + which);
        LAUNCHCHK("adam_group");
    }
    return AAE_OK;
}

// The small layers of optimiser which_a (enc_optim 0 / gen_optim 2: enc.lin1's bias, enc.lin2, enc.lin3) and, which_b = 1,
// of dec_optim (dec.lin1, dec.lin2) in ONE launch, their gradients read as the sum over n_peers gathered packets: packet q
// holds at packets_dev + q * peer_stride + span_offset (floats) a copy of the arena span that starts at AAE_T_GRAD +
// AAE_T_ENC_B1 (the spans the ranks of the both-sharded scheme all-gather behind their dL/d(a1) rows).
// Summed in peer order: bitwise the same on every rank.  Handles with an external first layer only.
int aae_apply_gathered(aae_handle m, int which_a, int which_b, const float* packets_dev, int64_t peer_stride,
                       int32_t n_peers, int64_t span_offset, void* stream) {
    if (!m || !packets_dev) return fail(AAE_EINVAL, "NULL argument");
    if (m->cfg.grad_mode != AAE_GRAD_EXPORT || !m->ext_first) return fail(AAE_ESTATE, "aae_apply_gathered: grad_mode=export with an external first layer");
    if ((which_a != O_ENC && which_a != O_GEN) || (which_b != -1 && which_b != O_DEC)) return fail(AAE_EINVAL, "aae_apply_gathered: which_a enc/gen, which_b -1/dec");
    if (n_peers < 1 || (peer_stride & 3) || (span_offset & 3)) return fail(AAE_EINVAL, "aae_apply_gathered: n_peers >= 1, strides in whole float4");
    AdamGroup grp; grp.njobs = 0;
    unsigned blocks = 0;
    const size_t base = m->Gr[P_B1].off;
    auto add = [&](int pid, int which) {
        const int set = which == O_GEN ? 1 : 0;
        AdamJob& j = grp.jobs[grp.njobs++];
        j.p = m->P[pid].p; j.m = m->M[set][pid].p; j.v = m->V[set][pid].p;
        j.g = const_cast<float*>(packets_dev) + span_offset + (m->Gr[pid].off - base) / sizeof(float);
        j.n4 = (unsigned)(m->P[pid].floats() / 4); j.blk0 = blocks;
        j.w4 = w4_of(m, pid); j.ld = (int)m->P[pid].ld;
        j.sc = m->sc + which; j.npeers = n_peers; j.pstride = peer_stride;
        blocks += (j.n4 + 255) / 256;
    };
    for (int pid = P_B1; pid <= P_W3; ++pid) add(pid, which_a);
    if (which_b == O_DEC) for (int pid = P_V1; pid < P_V3; ++pid) add(pid, O_DEC);
    hipLaunchKernelGGL(adam_group_kernel, dim3(blocks), dim3(256), 0, S(stream), grp, m->sc + which_a);
    LAUNCHCHK("adam_group (gathered)");
    return AAE_OK;
}

}  // extern "C"

#include "dp_step.h"
