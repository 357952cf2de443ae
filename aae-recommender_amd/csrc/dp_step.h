// The data-parallel training step as ONE library call (include/aaerec_hip.h: aae_dp_step, aae_rccl_*).
//
// Scheme: DESIGN.md 5.0 - both vocabulary-wide matrices (enc.lin1, dec.lin3) live with item slices, one per rank; the
// ranks exchange [global batch, n_hidden]-sized blocks only: 7 small collectives per step.  r1/r2 drove the phases from
// Python (aaerec/parallel.py: ~12 ABI calls + 7 torch.distributed calls per step): 0.35 ms of host time per step at
// world 8 against 0.33 ms of GPU work - the loop was host-bound before a single byte crossed xGMI.  Here the same
// choreography is enqueued by the library: kernels and collectives on the caller's stream, no host work in between
// (aae_comm_init + one entry point that enqueues kernels and RCCL calls).
//
// The collectives are a small table of function pointers (aae_collectives): aae_rccl_init fills it with RCCL calls on a
// communicator the library creates (librccl.so is opened at run time - the library has no link-time dependency on it);
// tests fill it with host-staged gloo collectives or single-process stand-ins.  Included at the end of aae_abi.hip
// (one translation unit); written against the public entry points.
#pragma once
#include <dlfcn.h>

namespace {

struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    void* CommInitRank = nullptr;                    // ncclCommInitRank(comm*, nranks, ncclUniqueId BY VALUE, rank): nccl_comm_init_rank_t
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, void*) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
struct NcclId { char internal[128]; };
typedef int (*nccl_comm_init_rank_t)(void**, int, NcclId, int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;       // ncclDataType_t / ncclRedOp_t of rccl.h

RcclApi* rccl_open() {           // runs once (function-local static initialiser below: thread-safe since C++11)
    static RcclApi api;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
    }
    if (!api.lib) return nullptr;
    api.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(api.lib, "ncclGetUniqueId"));
    api.CommInitRank = dlsym(api.lib, "ncclCommInitRank");
    api.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(api.lib, "ncclCommDestroy"));
    *reinterpret_cast<void**>(&api.AllGather) = dlsym(api.lib, "ncclAllGather");
    *reinterpret_cast<void**>(&api.ReduceScatter) = dlsym(api.lib, "ncclReduceScatter");
    *reinterpret_cast<void**>(&api.AllReduce) = dlsym(api.lib, "ncclAllReduce");
    *reinterpret_cast<void**>(&api.GetErrorString) = dlsym(api.lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.ReduceScatter || !api.AllReduce) {
        dlclose(api.lib); api.lib = nullptr; return nullptr;
    }
    return &api;
}
RcclApi* rccl_api() {
    static RcclApi* const api = rccl_open();
    return api;
}

struct RcclCtx { void* comm; };

int rccl_fail(RcclApi* a, const char* what, int rc) {
    return fail(AAE_EHIP, std::string(what) + ": " + (a->GetErrorString ? a->GetErrorString(rc) : "rccl error " + std::to_string(rc)));
}
int rccl_all_gather(void* ctx, const float* send, float* recv, int64_t count, void* stream) {
    RcclApi* a = rccl_api();
    const int rc = a->AllGather(send, recv, (size_t)count, kNcclFloat32, static_cast<RcclCtx*>(ctx)->comm, stream);
    return rc == 0 ? AAE_OK : rccl_fail(a, "ncclAllGather", rc);
}
int rccl_reduce_scatter(void* ctx, const float* send, float* recv, int64_t count, void* stream) {
    RcclApi* a = rccl_api();
    const int rc = a->ReduceScatter(send, recv, (size_t)count, kNcclFloat32, kNcclSum, static_cast<RcclCtx*>(ctx)->comm, stream);
    return rc == 0 ? AAE_OK : rccl_fail(a, "ncclReduceScatter", rc);
}
int rccl_all_reduce(void* ctx, float* buf, int64_t count, void* stream) {
    RcclApi* a = rccl_api();
    const int rc = a->AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, static_cast<RcclCtx*>(ctx)->comm, stream);
    return rc == 0 ? AAE_OK : rccl_fail(a, "ncclAllReduce", rc);
}

// arena view of a tensor id
int dp_view(aae_handle h, int id, float** p, aae_tensor* t) {
    TRY(aae_tensor_info(h, id, t));
    *p = reinterpret_cast<float*>(h->base + t->byte_offset);
    return AAE_OK;
}

}  // namespace

extern "C" {

int aae_rccl_unique_id(char id_out[128]) {
    if (!id_out) return fail(AAE_EINVAL, "id_out is NULL");
    RcclApi* a = rccl_api();
    if (!a) return fail(AAE_EHIP, "librccl.so could not be opened");
    const int rc = a->GetUniqueId(id_out);
    return rc == 0 ? AAE_OK : rccl_fail(a, "ncclGetUniqueId", rc);
}

int aae_rccl_init(const char id[128], int32_t world, int32_t rank, aae_collectives* out) {
    if (!id || !out) return fail(AAE_EINVAL, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(AAE_EINVAL, "aae_rccl_init: need 0 <= rank < world");
    RcclApi* a = rccl_api();
    if (!a) return fail(AAE_EHIP, "librccl.so could not be opened");
    NcclId nid; memcpy(nid.internal, id, sizeof(nid.internal));
    RcclCtx* ctx = new RcclCtx{nullptr};
    const int rc = reinterpret_cast<nccl_comm_init_rank_t>(a->CommInitRank)(&ctx->comm, world, nid, rank);
    if (rc != 0) { delete ctx; return rccl_fail(a, "ncclCommInitRank", rc); }
    out->ctx = ctx; out->world = world; out->rank = rank;
    out->all_gather = rccl_all_gather; out->reduce_scatter = rccl_reduce_scatter; out->all_reduce = rccl_all_reduce;
    return AAE_OK;
}

int aae_rccl_destroy(aae_collectives* c) {
    if (!c || !c->ctx) return AAE_OK;
    RcclApi* a = rccl_api();
    RcclCtx* ctx = static_cast<RcclCtx*>(c->ctx);
    if (a && ctx->comm) (void)a->CommDestroy(ctx->comm);
    delete ctx;
    c->ctx = nullptr;
    return AAE_OK;
}

// single-process stand-in (tools/vocab_rank_time.py: a rank's compute with the collectives replaced by device copies of
// the same shapes): all_gather = the operand repeated `world` times, reduce_scatter = the first chunk, all_reduce = identity
static int echo_all_gather(void* ctx, const float* send, float* recv, int64_t count, void* stream) {
    const int world = (int)(intptr_t)ctx;
    for (int r = 0; r < world; ++r)
        HIPCHK(hipMemcpyAsync(recv + (size_t)r * count, send, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, S(stream)));
    return AAE_OK;
}
static int echo_reduce_scatter(void*, const float* send, float* recv, int64_t count, void* stream) {
    HIPCHK(hipMemcpyAsync(recv, send, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, S(stream)));
    return AAE_OK;
}
static int echo_all_reduce(void*, float*, int64_t, void*) { return AAE_OK; }

int aae_echo_collectives(int32_t world, aae_collectives* out) {
    if (!out || world < 1) return fail(AAE_EINVAL, "aae_echo_collectives: bad argument");
    out->ctx = (void*)(intptr_t)world; out->world = world; out->rank = 0;
    out->all_gather = echo_all_gather; out->reduce_scatter = echo_reduce_scatter; out->all_reduce = echo_all_reduce;
    return AAE_OK;
}

int aae_memcpy_sync(void* dst, const void* src, size_t bytes, void* stream) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, S(stream)));
    HIPCHK(hipStreamSynchronize(S(stream)));
    return AAE_OK;
}

// floats of one rank's gathered packet [dL/d(a1) rows of n documents | small-layer gradients incl. the decoder's]
static int64_t dp_packet_floats(const aae_model* m, int n) {
    const Ten& e = m->Gr[P_V2];
    const size_t start = m->ga1x.off + (size_t)(m->ga1x.rows - n) * m->ga1x.ld * sizeof(float);
    const size_t end = e.off + (size_t)e.rows * e.ld * sizeof(float);
    return (int64_t)((end - start) / sizeof(float));
}

// The scratch aae_dp_step gathers the ranks' packets into (world x the larger packet of n_rows local documents), allocated
// HERE - at set-up time - instead of inside the first step (a hipMalloc in the step); aae_dp_step still grows it
// when a caller skipped this.
int aae_dp_reserve(aae_handle m, int32_t n_rows, int32_t world) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->cfg.grad_mode != AAE_GRAD_EXPORT || !m->ga1x.p) return fail(AAE_ESTATE, "aae_dp_reserve: the replica needs grad_mode = export");
    if (n_rows < 1 || n_rows > m->R || world < 1) return fail(AAE_EINVAL, "aae_dp_reserve: n_rows in [1, max_batch], world >= 1");
    const size_t need = (size_t)dp_packet_floats(m, n_rows) * world;
    if (m->dp_scratch_floats >= need) return AAE_OK;
    if (m->dp_scratch) (void)hipFree(m->dp_scratch);
    m->dp_scratch = nullptr; m->dp_scratch_floats = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->dp_scratch), need * sizeof(float)));
    m->dp_scratch_floats = need;
    return AAE_OK;
}

// One partial_fit of the data-parallel model (aae.py:745-766 over the global batch): `replica` = this rank's replica
// handle (grad_mode export, first layer external), `slice` = its item-slice handle (fused optimiser; enc.lin1 and dec.lin3
// rows of its items), local = this rank's documents in the replica's corpus, global_slice = the GLOBAL batch (rank-major:
// rank r's documents are rows [r * n, (r + 1) * n)) in the slice's corpus (its items' columns, ids rebased).
int aae_dp_step(aae_handle m, aae_handle sl, const aae_collectives* c, const aae_batch* local, const aae_batch* global_slice,
                const aae_batch* next_global_slice, const float* cond_dev, const aae_rng_inject* inject, void* stream) {
    if (!m || !sl || !c || !local || !global_slice) return fail(AAE_EINVAL, "aae_dp_step: NULL argument");
    if (!c->all_gather || !c->reduce_scatter || !c->all_reduce || c->world < 1 || c->rank < 0 || c->rank >= c->world)
        return fail(AAE_EINVAL, "aae_dp_step: incomplete collectives table");
    if (m->cfg.grad_mode != AAE_GRAD_EXPORT || !m->ext_first || !m->ga1x.p)
        return fail(AAE_ESTATE, "aae_dp_step: the replica needs grad_mode = export and an external first layer (aae_set_first_layer_external)");
    if (sl->cfg.grad_mode != AAE_GRAD_FUSED) return fail(AAE_ESTATE, "aae_dp_step: the slice handle needs the fused optimiser");
    if (m->h != sl->h || m->ldh != sl->ldh) return fail(AAE_EINVAL, "aae_dp_step: replica and slice disagree on n_hidden");
    const int world = c->world, rank = c->rank;
    const int n = local->n_rows, G = global_slice->n_rows;
    if (G != n * world) return fail(AAE_EINVAL, "aae_dp_step: the global batch must be world x the local batch");
    if (G > sl->R || n > m->R) return fail(AAE_EINVAL, "aae_dp_step: batch larger than max_batch");
    const int ldh = m->ldh;
    const int64_t blk = (int64_t)n * ldh;               // floats of one rank's [n, ldh] activation block

    TRY(aae_set_grad_scale(m, (float)n / (float)G));
    TRY(aae_set_rng_rows(m, (int64_t)rank * n, G));
    if (next_global_slice) TRY(aae_prefetch_batch(sl, next_global_slice));
    TRY(aae_join(sl, stream));                          // (a deferred optimiser launch of the slice's last step, if any)

    float *a1, *a1_all, *dh2, *dh2_all, *da2, *da2_all, *b1;
    aae_tensor t;
    TRY(dp_view(m, AAE_T_ACT_A1, &a1, &t));   TRY(dp_view(sl, AAE_T_ACT_A1, &a1_all, &t));
    TRY(dp_view(m, AAE_T_ACT_DH2, &dh2, &t)); TRY(dp_view(sl, AAE_T_ACT_DH2, &dh2_all, &t));
    TRY(dp_view(m, AAE_T_ACT_DA2, &da2, &t)); TRY(dp_view(sl, AAE_T_ACT_DA2, &da2_all, &t));
    TRY(dp_view(m, AAE_T_ENC_B1, &b1, &t));
    const float* bias = rank == 0 ? b1 : nullptr;       // exactly one share adds the (replicated) bias

    // the gathered packets [dL/d(a1) rows | small layers' gradients] of all ranks: scratch of the replica handle
    auto packet = [&](bool with_decoder, float** pk, int64_t* pk_floats, int64_t* span_off) {
        const Ten& e = m->Gr[with_decoder ? P_V2 : P_W3];
        const size_t start = m->ga1x.off + (size_t)(m->ga1x.rows - n) * m->ga1x.ld * sizeof(float);
        const size_t end = e.off + (size_t)e.rows * e.ld * sizeof(float);
        *pk = reinterpret_cast<float*>(m->base + start);
        *pk_floats = (int64_t)((end - start) / sizeof(float));
        *span_off = (int64_t)((m->Gr[P_B1].off - start) / sizeof(float));
    };
    float* pk; int64_t pkf, soff;
    packet(true, &pk, &pkf, &soff);                     // (the larger of the two packets sizes the scratch)
    if (m->dp_scratch_floats < (size_t)pkf * world) TRY(aae_dp_reserve(m, n, world));      // (a caller that did not reserve at set-up)
    float* allp = m->dp_scratch;

    // ---- ae phase
    TRY(aae_first_layer_forward(sl, global_slice, bias, stream));
    { ProfScope ps(m, AAE_K_COLLECTIVE, S(stream)); TRY(c->reduce_scatter(c->ctx, a1_all, a1, blk, stream)); }
    TRY(aae_ae_forward(m, local, cond_dev, inject, stream));
    { ProfScope ps(m, AAE_K_COLLECTIVE, S(stream)); TRY(c->all_gather(c->ctx, dh2, dh2_all, blk, stream)); }
    TRY(aae_output_layer_step(sl, nullptr, stream));
    { ProfScope ps(m, AAE_K_COLLECTIVE, S(stream)); TRY(c->reduce_scatter(c->ctx, da2_all, da2, blk, stream)); }
    TRY(aae_ae_backward(m, nullptr, 0, stream));
    { ProfScope ps(m, AAE_K_COLLECTIVE, S(stream)); TRY(c->all_gather(c->ctx, pk, allp, pkf, stream)); }
    TRY(aae_apply_gathered(m, O_ENC, O_DEC, allp, pkf, world, soff, stream));
    TRY(aae_first_layer_update(sl, allp, ldh, n, pkf, O_ENC, stream));
    if (m->ae_only) return AAE_OK;
    // ---- disc phase (Enc_eval with the updated first layer), gen phase
    TRY(aae_first_layer_forward(sl, nullptr, bias, stream));
    { ProfScope ps(m, AAE_K_COLLECTIVE, S(stream)); TRY(c->reduce_scatter(c->ctx, a1_all, a1, blk, stream)); }
    TRY(aae_disc_step(m, nullptr, stream));
    {
        const Ten& d1 = m->Gr[P_D1]; const Ten& d3 = m->Gr[P_D3];
        const int64_t cnt = (int64_t)((d3.off + (size_t)d3.rows * d3.ld * sizeof(float) - d1.off) / sizeof(float));
        { ProfScope ps(m, AAE_K_COLLECTIVE, S(stream)); TRY(c->all_reduce(c->ctx, reinterpret_cast<float*>(m->base + d1.off), cnt, stream)); }
    }
    TRY(aae_apply_updates(m, O_DISC, stream));
    TRY(aae_gen_step(m, nullptr, stream));
    packet(false, &pk, &pkf, &soff);
    { ProfScope ps(m, AAE_K_COLLECTIVE, S(stream)); TRY(c->all_gather(c->ctx, pk, allp, pkf, stream)); }
    TRY(aae_apply_gathered(m, O_GEN, -1, allp, pkf, world, soff, stream));
    TRY(aae_first_layer_update(sl, allp, ldh, n, pkf, O_GEN, stream));
    return AAE_OK;
}

// One partial_fit of the ITEM-SHARDED model with REPLICATED hidden stacks (dp_mode = 'shard'; r4): ONE handle
// per rank holds its item slice of the two vocabulary-wide layers (rows of dec.lin3, columns of enc.lin1, their optimiser
// states) and a full copy of every hidden layer, and runs the WHOLE global batch through the hidden stacks.  Every rank then
// computes the same activations, the same small-layer gradients and the same optimiser updates from the same inputs - the
// replicas stay identical with NO gradient exchange; what crosses the ranks are only the three partial sums over the item
// slices, each an all-reduce of [global rows, n_hidden] floats (0.65 MB at 8 x 100 documents):
//     x * enc.lin1^T (ae phase)  |  dL/d(dh2) = sum over the slices' items of G * V3  |  x * enc.lin1^T again (Enc_eval)
// 3 collectives per step instead of the both-sharded scheme's 7 (aae_dp_step), no packets, no second handle's step to open
// (-14 launches per rank and step); the price is the hidden stacks at world x the rows (their launches are latency-bound
// on a mostly idle chip: +~20 us each at 800 rows).
//   handle      fused optimiser, aae_set_first_layer_external(1), aae_set_doc_l1 (whole-document L1 norms), created with
//               max_batch = the global batch (cfg.blocked_output = 1 beyond 112 rows)
//   batch       the GLOBAL batch in the handle's corpus (its items' columns, ids rebased); next_batch: named ahead or NULL
//   item_share  items of this handle / items of the model: the BCE is a mean over all items
//   cond_dev    the condition block of ALL rows of the batch; inject as aae_step
int aae_shard_step(aae_handle m, const aae_collectives* c, const aae_batch* batch, const aae_batch* next_batch,
                   const float* cond_dev, const aae_rng_inject* inject, float item_share, void* stream) {
    if (!m || !c || !batch) return fail(AAE_EINVAL, "aae_shard_step: NULL argument");
    if (!c->all_reduce || c->world < 1 || c->rank < 0 || c->rank >= c->world) return fail(AAE_EINVAL, "aae_shard_step: incomplete collectives table");
    if (m->cfg.grad_mode != AAE_GRAD_FUSED || !m->ext_first || !m->use_chain || m->vae)
        return fail(AAE_ESTATE, "aae_shard_step: the handle needs the fused optimiser, the layer-chain kernels and aae_set_first_layer_external(1)");
    if (m->cfg.cond_inc > 0 && !cond_dev) return fail(AAE_EINVAL, "cond_inc > 0 needs cond_dev");
    if (!(item_share > 0.f && item_share <= 1.f)) return fail(AAE_EINVAL, "aae_shard_step: item_share must be in (0, 1]");
    hipStream_t s = S(stream);
    m->own_first = true;            // its rows of enc.lin1 are trained here: their per-item update rides in the weight-gradient launches
    const float* bias = c->rank == 0 ? m->P[P_B1].p : nullptr;      // exactly one share adds the (replicated) bias
    if (next_batch) TRY(aae_prefetch_batch(m, next_batch));
    // ---- ae phase
    TRY(aae_first_layer_forward(m, batch, bias, stream));           // opens the step: this slice's share of x * enc.lin1^T
    const int64_t cnt = (int64_t)m->rows * m->ldh;
    { ProfScope ps(m, AAE_K_COLLECTIVE, s); TRY(c->all_reduce(c->ctx, m->a1.p, cnt, stream)); }
    remember_inject(m, inject, false);
    m->dec_hidden_done = false; m->enc_bwd_done = false;
    TRY(chain_ae_forward(m, true, cond_dev, nullptr, s));           // every row of the batch: dropout + activation on a1, hidden layers -> dh2
    m->phase = 1;
    if (m->cfg.cond_inc > 0) {
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)m->rows * m->cfg.cond_inc)), dim3(256), 0, s, cond_dev,
                           m->cfg.cond_inc, m->zc.p + m->c, m->ldc, m->rows, m->cfg.cond_inc, 1.0f);
        LAUNCHCHK("copy cond");
    }
    m->grad_scale = item_share;
    const int rc = aae_output_layer_step(m, nullptr, stream);       // its items' logits, BCE, dV3 + dec_optim, dL/d(dh2) partial
    m->grad_scale = 1.f;
    TRY(rc);
    { ProfScope ps(m, AAE_K_COLLECTIVE, s); TRY(c->all_reduce(c->ctx, m->da2.p, cnt, stream)); }
    TRY(aae_ae_backward(m, nullptr, 0, stream));                     // hidden layers backward + their optimisers (the same on every rank), enc_optim on its rows of enc.lin1
    if (m->ae_only) { m->phase = 0; return AAE_OK; }
    // ---- disc phase (Enc_eval with the updated first layer), gen phase
    TRY(aae_first_layer_forward(m, nullptr, bias, stream));
    { ProfScope ps(m, AAE_K_COLLECTIVE, s); TRY(c->all_reduce(c->ctx, m->a1.p, cnt, stream)); }
    TRY(aae_disc_step(m, nullptr, stream));
    TRY(aae_gen_step(m, nullptr, stream));                           // (gen_optim on its rows of enc.lin1 included)
    return AAE_OK;
}

}  // extern "C"
