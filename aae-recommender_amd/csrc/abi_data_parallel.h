// Data parallel entry points: row-sparse exchange of the first layer's gradient, optimiser steps on reduced / gathered gradients.
// (one of the parts of aae_abi.hip's translation unit: included there in order, not on its own)
#pragma once

extern "C" {

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
    if (n_peers > 1 && m->pslot && !m->opt.w1_serial) {
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
