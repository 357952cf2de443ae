// The step, part 2: the decoder's output layer + BCE + its backward (aae.py:176-177, 693-706) - fused / split / row-blocked / three-GEMM forms.
// (one of the parts of aae_abi.hip's translation unit: included there in order, not on its own)
#pragma once

extern "C" {

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
    TRY(join_output_layer(m, s));       // a deferred launch of the step before that was left running at this step's opening (late join)
    const float gscale = m->grad_scale / ((float)B * (float)N);
    DropSpec d1 = make_drop(m, 0, true, mk2, nullptr, B, h, 2);
    DropSpec d2 = make_drop(m, 1, true, mk3, nullptr, B, h, 3);
    // row blocks of the fused output layer: one launch covers at most 112 rows; larger batches (cfg.blocked_output) run as
    // nblk launches of the split form over equal row blocks
    const int nblk = m->have_batch ? row_blocks(m) : 1;
    const int Bb = (B + nblk - 1) / nblk;                        // rows per block (the last one may be shorter)
    const bool win = x3_big_span(m->N, m->ldh);                  // dec.lin3 beyond 2^31 bytes: the kernels' moving-window instantiations (dec_fused.h)
    const size_t fused_lds = out_bf16(m) ? dec_fused_bf16_lds_bytes(m->fused_nb ? m->fused_nb : 13) : dec_fused_lds_bytes(Bb, h);
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
        fa.one_term = m->bf16_x3 ? 1 : 0;
        fa.dbg_skip = m->opt.dec_skip;
        const bool want_ts = m->opt.dec_ts[0] != 0;        // debug: phase timeline of one tile
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
        const bool ts_x3 = want_ts && strcmp(m->opt.dec_ts, "x3") == 0;
        const bool ts_obk = want_ts && strcmp(m->opt.dec_ts, "obk") == 0;
        if (m->split_ok && m->split_wgs > 0 && split_fits && fa.gradV3 == nullptr && (!want_ts || ((ts_x3 || ts_obk) && m->x3_ok && !out_bf16(m))) && (fa.dbg_skip & ~(256 | 0xF000 | 0x30000)) == 0) {
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
            // late join (abi_model.h): single row block on the emulated product - dec_crit_x3_kernel sets the deferred launch's
            // dh2 and step scalars aside, dec_opt_x3_kernel reads the copies
            const bool late = m->late_enabled && nblk == 1 && m->x3_ok && !out_bf16(m) && !want_ts &&
                              m->dh2s.p && m->sc_snap && B <= m->dh2s.rows && grid >= B;
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
                if (late) { b.dh2_snap = m->dh2s.p; b.sc_snap = m->sc_snap; }
                const int grid = crit_grid;
                const int r = nblk - 1;
                hipEvent_t start = nullptr, stop = r == nblk - 1 ? m->ev_crit : nullptr;
                (void)prof_pair(m, AAE_K_DEC_CRIT, &start, &stop);
                if (out_bf16(m)) switch (m->fused_nb) {
                    case 4: hipExtLaunchKernelGGL((dec_fused_bf16_kernel<4, kDecCrit>), dim3(grid), dim3(kBT), (uint32_t)fused_lds, s, start, stop, 0, b); break;
                    case 7: hipExtLaunchKernelGGL((dec_fused_bf16_kernel<7, kDecCrit>), dim3(grid), dim3(kBT), (uint32_t)fused_lds, s, start, stop, 0, b); break;
                    default: hipExtLaunchKernelGGL((dec_fused_bf16_kernel<13, kDecCrit>), dim3(grid), dim3(kBT), (uint32_t)fused_lds, s, start, stop, 0, b); break;
                } else if (m->x3_ok) {
                    const uint32_t lds3 = (uint32_t)dec_crit_x3_lds_bytes(m->fused_nb);
                    if (m->bf16_one) switch (m->fused_nb) {     // bf16 mode: the one-term instantiation (one matrix instruction per product)
                    case 4: { if (win) hipExtLaunchKernelGGL((dec_crit_x3_kernel<4, false, true, true>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_crit_x3_kernel<4, false, true>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); } break;
                    case 7: { if (win) hipExtLaunchKernelGGL((dec_crit_x3_kernel<7, false, true, true>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_crit_x3_kernel<7, false, true>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); } break;
                    default: { if (win) hipExtLaunchKernelGGL((dec_crit_x3_kernel<13, false, true, true>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_crit_x3_kernel<13, false, true>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); } break;
                    } else
                    switch (m->fused_nb) {
                    case 4: { if (win) hipExtLaunchKernelGGL((dec_crit_x3_kernel<4, false, false, true>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_crit_x3_kernel<4>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); } break;
                    case 7: { if (win) hipExtLaunchKernelGGL((dec_crit_x3_kernel<7, false, false, true>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_crit_x3_kernel<7>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); } break;
                    default:
                        if (b.ts) hipExtLaunchKernelGGL((dec_crit_x3_kernel<13, true>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b);
                        else { if (win) hipExtLaunchKernelGGL((dec_crit_x3_kernel<13, false, false, true>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_crit_x3_kernel<13>), dim3(grid), dim3(kNT), lds3, s, start, stop, 0, b); }
                        break;
                    }
                } else switch (m->fused_nb) {
                    case 4: { if (win) hipExtLaunchKernelGGL((dec_fused_kernel<4, kDecCrit, true>), dim3(grid), dim3(kNT), (uint32_t)fused_lds, s, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_fused_kernel<4, kDecCrit>), dim3(grid), dim3(kNT), (uint32_t)fused_lds, s, start, stop, 0, b); } break;
                    case 7: { if (win) hipExtLaunchKernelGGL((dec_fused_kernel<7, kDecCrit, true>), dim3(grid), dim3(kNT), (uint32_t)fused_lds, s, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_fused_kernel<7, kDecCrit>), dim3(grid), dim3(kNT), (uint32_t)fused_lds, s, start, stop, 0, b); } break;
                    default: { if (win) hipExtLaunchKernelGGL((dec_fused_kernel<13, kDecCrit, true>), dim3(grid), dim3(kNT), (uint32_t)fused_lds, s, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_fused_kernel<13, kDecCrit>), dim3(grid), dim3(kNT), (uint32_t)fused_lds, s, start, stop, 0, b); } break;
                }
                LAUNCHCHK("dec_fused (critical launch)");
                if (r == nblk - 1) HIPCHK(hipStreamWaitEvent(m->side, stop, 0));
            }
            const int g2 = std::min(ntiles, std::min(m->split_wgs, m->n_cu));
            bool late_launched = false;
            // nblk > 1 and at most kOBT tiles per workgroup on the chip: the deferred half of every block in ONE launch
            // (dec_opt_blocks_kernel), else one launch per block with the dV3 partial going through Gacc
            constexpr bool no_obk = false, no_opt_x3 = false;
            const bool one_opt = nblk > 1 && !out_bf16(m) && !no_obk && ntiles <= kOBT * m->n_cu &&
                                 dec_opt_blocks_lds_bytes(Bb) <= 160 * 1024;
            // (r3) the same on the emulated product, any vocabulary size: dec_opt_blocks_x3_kernel (dec_crit_x3.h)
            constexpr bool no_obk_x3 = false;
            if (nblk > 1 && !out_bf16(m) && m->x3_ok && !no_opt_x3 && !no_obk_x3 && !no_obk && m->dh2f.p) {
                DecFusedArgs b = fa;
                b.nblk = nblk; b.Bb = Bb;
                hipLaunchKernelGGL(dh2_frag_kernel, dim3((B + kXCH - 1) / kXCH, m->fused_nb), dim3(128), 0, m->side, m->dh2.p, m->ldh, B,
                                   reinterpret_cast<u32x4_t*>(m->dh2f.p), b.one_term);
                b.acc = m->dh2f.p;                      // (this kernel's reading of the field: the fragment image)
                // tile groups of at most kXBT tiles, the same number (+-1 tile) for every workgroup and round
                // Workgroups: one per ~16 tiles, between half and three quarters of the CUs (tools/debug/sweep_obk_wgs*.sh, late r3,
                // ms per step): 100 k items x 512 rows 0.774 / 0.725 / 0.710 / 0.703 / 0.690 / 0.755 / 0.749 on 128 / 144 / 160 /
                // 176 / 192 / 208 / 224; x 256 rows 0.506 / 0.468 / 0.443 / 0.506 on 128 / 160 / 192 / 208; x 1024 rows 1.260 /
                // 1.245 / 1.356 on 160 / 192 / 208; 47 k items x 500 rows 0.388 / 0.372 / 0.369 / 0.381 / 0.378 on 96 / 112 / 128 /
                // 144 / 160; an item slice of 12.5 k items x 800 rows 0.382 / 0.379 / 0.400 / 0.387 ms of per-rank compute on
                // 96 / 128 / 160 / 192.  (Beyond 3/4 of the chip the step's own launches lose more than this one gains.)
                // (a slice of thousands of tiles - C5: 275 k items x 512 rows, 8 594 tiles - outlasts the step's tail by far: 7/8 of the
                //  chip there, r4: one rank's step 1.58 | 1.52 | 1.61 | 1.59 ms on 192 | 224 | 240 | 256 workgroups)
                const int wg_cap = ntiles >= 4096 ? m->n_cu * 7 / 8 : m->n_cu * 3 / 4;
                // (r5: a layer of a few hundred tiles - C4: 144 - is two or three tiles per workgroup on a sixth of the chip: 0.3549 |
                //  0.3493 | 0.3489 | 0.3486 | 0.3524 ms/step on 128 | 32 | 48 | 64 | 96 workgroups, tools/debug/c4_obk_sweep.sh)
                const int by_tiles = ntiles < 384 ? std::max(32, std::min(m->n_cu / 2, ntiles / 3))
                                                  : std::max(m->n_cu / 2, std::min(wg_cap, (int)(ntiles / 16.3 / 8.0 + 0.5) * 8));
                const int g3 = std::max(1, std::min(by_tiles, std::min(ntiles, m->n_cu)));
                const int rounds = (ntiles + g3 * kXBT - 1) / (g3 * kXBT);
                b.tpp = g3 * rounds;
                const uint32_t lds3 = (uint32_t)dec_opt_blocks_x3_lds_bytes();
                hipEvent_t start = nullptr, stop = nullptr;
                (void)prof_pair(m, AAE_K_DEC_OPT, &start, &stop);
                if (ts_obk && m->fused_nb == 13) {
                    hipExtLaunchKernelGGL((dec_opt_blocks_x3_kernel<13, true>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b);
                    HIPCHK(hipStreamSynchronize(m->side));
                    unsigned long long t[128];
                    HIPCHK(hipMemcpy(t, ts_dev, sizeof(t), hipMemcpyDeviceToHost));
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
                    case 4: { if (win) hipExtLaunchKernelGGL((dec_opt_blocks_x3_kernel<4, false, true>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_opt_blocks_x3_kernel<4>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b); } break;
                    case 7: { if (win) hipExtLaunchKernelGGL((dec_opt_blocks_x3_kernel<7, false, true>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_opt_blocks_x3_kernel<7>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b); } break;
                    default: { if (win) hipExtLaunchKernelGGL((dec_opt_blocks_x3_kernel<13, false, true>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_opt_blocks_x3_kernel<13>), dim3(g3), dim3(kNT), lds3, m->side, start, stop, 0, b); } break;
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
                if (out_bf16(m)) switch (m->fused_nb) {
                    case 4: hipExtLaunchKernelGGL((dec_fused_bf16_kernel<4, kDecOpt>), dim3(g2), dim3(kBT), (uint32_t)fused_lds, m->side, start, stop, 0, b); break;
                    case 7: hipExtLaunchKernelGGL((dec_fused_bf16_kernel<7, kDecOpt>), dim3(g2), dim3(kBT), (uint32_t)fused_lds, m->side, start, stop, 0, b); break;
                    default: hipExtLaunchKernelGGL((dec_fused_bf16_kernel<13, kDecOpt>), dim3(g2), dim3(kBT), (uint32_t)fused_lds, m->side, start, stop, 0, b); break;
                } else if (r == 0 && nblk == 1 && m->x3_ok && !no_opt_x3) {
                    // (the 3-term bf16 emulation of dV3 = G^T dh2, dec_crit_x3.h)
                    // (one-term instantiation: 78 VGPRs - six of its waves fit a SIMD, so the step's own launches would be dealt onto
                    //  its CUs and run beside its streams; its LDS claim is raised until no other workgroup of the step fits there)
                    const uint32_t lds_nat = (uint32_t)dec_opt_x3_lds_bytes();
                    const uint32_t lds3 = m->bf16_one ? std::max(lds_nat, 150u * 1024u) : lds_nat;
                    if (late) { b.dh2 = m->dh2s.p; b.sc = m->sc_snap; late_launched = true; }
                    if (m->bf16_one) switch (m->fused_nb) {
                    case 4: { if (win) hipExtLaunchKernelGGL((dec_opt_x3_kernel<4, true, true>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_opt_x3_kernel<4, true>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); } break;
                    case 7: { if (win) hipExtLaunchKernelGGL((dec_opt_x3_kernel<7, true, true>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_opt_x3_kernel<7, true>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); } break;
                    default: { if (win) hipExtLaunchKernelGGL((dec_opt_x3_kernel<13, true, true>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_opt_x3_kernel<13, true>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); } break;
                    } else
                    switch (m->fused_nb) {
                    case 4: { if (win) hipExtLaunchKernelGGL((dec_opt_x3_kernel<4, false, true>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_opt_x3_kernel<4>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); } break;
                    case 7: { if (win) hipExtLaunchKernelGGL((dec_opt_x3_kernel<7, false, true>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_opt_x3_kernel<7>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); } break;
                    default: { if (win) hipExtLaunchKernelGGL((dec_opt_x3_kernel<13, false, true>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_opt_x3_kernel<13>), dim3(g2), dim3(kNT), lds3, m->side, start, stop, 0, b); } break;
                    }
                } else if (r == 0) switch (m->fused_nb) {
                    case 4: { if (win) hipExtLaunchKernelGGL((dec_fused_kernel<4, kDecOpt, true>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_fused_kernel<4, kDecOpt>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); } break;
                    case 7: { if (win) hipExtLaunchKernelGGL((dec_fused_kernel<7, kDecOpt, true>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_fused_kernel<7, kDecOpt>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); } break;
                    default: { if (win) hipExtLaunchKernelGGL((dec_fused_kernel<13, kDecOpt, true>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_fused_kernel<13, kDecOpt>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); } break;
                } else switch (m->fused_nb) {
                    case 4: { if (win) hipExtLaunchKernelGGL((dec_fused_kernel<4, kDecOptAcc, true>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_fused_kernel<4, kDecOptAcc>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); } break;
                    case 7: { if (win) hipExtLaunchKernelGGL((dec_fused_kernel<7, kDecOptAcc, true>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_fused_kernel<7, kDecOptAcc>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); } break;
                    default: { if (win) hipExtLaunchKernelGGL((dec_fused_kernel<13, kDecOptAcc, true>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); else hipExtLaunchKernelGGL((dec_fused_kernel<13, kDecOptAcc>), dim3(g2), dim3(kNT), (uint32_t)fused_lds, m->side, start, stop, 0, b); } break;
                }
                LAUNCHCHK("dec_fused (optimiser launch)");
            }
            TRY(side_done(m, m->ev_opt));
            m->opt_pending = true;
            m->late_ok = late_launched;
            m->last_out_split = true; m->side_ordered = true;
            // an item slice's next batch (named ahead): its distinct items and their deferred-Adam catch-up behind the
            // deferred launch on the same stream (ordered behind this step's head by ev_crit; rows of the running batch
            // are skipped there, the step's own updates bring them to the same step)
            if (m->only_output_layer && m->pf_armed && m->mark2 && m->lazy) TRY(launch_prefetch(m, false));
        } else
        {
            m->last_out_split = false; m->side_ordered = false;
            ProfScope ps(m, AAE_K_DEC_FUSED, s);
            // (the single launch of bf16 mode is always dec_fused_bf16.h's own kernel: there is no rounded-operand form of it)
            const size_t bf_lds = dec_fused_bf16_lds_bytes(m->fused_nb ? m->fused_nb : 13);
            if (m->bf16) switch (m->fused_nb) {
                case 4: hipLaunchKernelGGL(dec_fused_bf16_kernel<4>, dim3(grid), dim3(kBT), bf_lds, s, fa); break;
                case 7: hipLaunchKernelGGL(dec_fused_bf16_kernel<7>, dim3(grid), dim3(kBT), bf_lds, s, fa); break;
                default: hipLaunchKernelGGL(dec_fused_bf16_kernel<13>, dim3(grid), dim3(kBT), bf_lds, s, fa); break;
            } else switch (m->fused_nb) {
                case 4: { if (win) hipLaunchKernelGGL((dec_fused_kernel<4, kDecFused, true>), dim3(grid), dim3(kNT), fused_lds, s, fa); else hipLaunchKernelGGL((dec_fused_kernel<4>), dim3(grid), dim3(kNT), fused_lds, s, fa); } break;
                case 7: { if (win) hipLaunchKernelGGL((dec_fused_kernel<7, kDecFused, true>), dim3(grid), dim3(kNT), fused_lds, s, fa); else hipLaunchKernelGGL((dec_fused_kernel<7>), dim3(grid), dim3(kNT), fused_lds, s, fa); } break;
                default: { if (win) hipLaunchKernelGGL((dec_fused_kernel<13, kDecFused, true>), dim3(grid), dim3(kNT), fused_lds, s, fa); else hipLaunchKernelGGL((dec_fused_kernel<13>), dim3(grid), dim3(kNT), fused_lds, s, fa); } break;
            }
        }
        LAUNCHCHK("dec_fused");
        if (want_ts) {
            unsigned long long t[128];
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipMemcpy(t, ts_dev, sizeof(t), hipMemcpyDeviceToHost));
            if (m->last_out_split && ts_obk) { /* printed at the launch */ }
            else if (m->last_out_split)
                fprintf(stderr, "[dec_crit_x3 tile 5] barrier=%.2f S0=%.2f GEMM1=%.2f BCE=%.2f GEMM3=%.2f | wg 0: prologue=%.2f loop=%.2f (%llu tiles, %.2f each) epilogue=%.2f us\n",
                        (t[14] - t[0]) * 0.01, (t[1] - t[14]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01,
                        (t[11] - t[10]) * 0.01, (t[7] - t[11]) * 0.01, t[13], (t[7] - t[11]) * 0.01 / (double)(t[13] ? t[13] : 1),
                        (t[12] - t[7]) * 0.01);
            else {
            if (out_bf16(m))
                for (int k = 0; k < 5; ++k) {
                    fprintf(stderr, "[dec_fused_bf16 arrivals at barrier %d, us after the unit's start]", k);
                    for (int w = 0; w < 16; ++w) fprintf(stderr, " %.2f", ((double)t[16 + 16 * k + w] - (double)t[0]) * 0.01);
                    fprintf(stderr, "\n");
                }
            fprintf(stderr, "[dec_fused tile 5] S0=%.2f GEMM1+BCE0=%.2f entries=%.2f GEMM2+GEMM3=%.2f S5=%.2f | tile=%.2f us, %.0f shader clocks -> %.2f GHz\n",
                    (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01,
                    (t[6] - t[4]) * 0.01, (t[6] - t[0]) * 0.01, (double)(t[9] - t[8]),
                    (double)(t[9] - t[8]) / ((t[6] - t[0]) * 10.0));
            if (out_bf16(m)) fprintf(stderr, "[dec_fused_bf16 S0] barrier A=%.2f work=%.2f barrier B=%.2f us\n", (t[14] - t[0]) * 0.01, (t[15] - t[14]) * 0.01, (t[1] - t[15]) * 0.01);
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
    constexpr bool defer_dv3 = true;
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


}  // extern "C"
