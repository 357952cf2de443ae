// state_dict import / export: parameters and optimiser state in the reference's layouts (aae.py:782-804).
// (one of the parts of aae_abi.hip's translation unit: included there in order, not on its own)
#pragma once

extern "C" {

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
        h->spec_tab_ok = false;
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


}  // extern "C"
