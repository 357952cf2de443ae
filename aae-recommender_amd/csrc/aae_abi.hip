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
#include "rank_x3.h"
#include "chain.h"
#include "chain4.h"
#include "chain16x3.h"
#include "cond_embed.h"
#include "w1_update.h"

using namespace aae;

#include <map>
static std::map<std::string, std::string> g_options;      // aae_set_option: name -> value (wins over the environment)
const char* option_value(const char* name) {
    auto it = g_options.find(name);
    if (it != g_options.end()) return it->second.c_str();
    return getenv((std::string("AAE_") + name).c_str());
}

#include "abi_model.h"
#include "abi_layers.h"
#include "abi_chains.h"
#include "abi_rank.h"

// ------------------------------------------------------------------------------------------
extern "C" {

int aae_abi_version(void) { return AAE_ABI_VERSION; }

int aae_set_option(const char* name, const char* value) {
    if (!name || !*name) return fail(AAE_EINVAL, "option name is empty");
    static const char* known[] = {"NO_CHAIN", "CHAIN16", "SPLIT_ANY", "BLOCKED_ANY", "EARLY_ANY", "NO_LATE_JOIN", "NO_ITEM_COUNT", "W1_SERIAL",
                                  "NO_RANK_FUSED", "CHAIN_KSLICES", "X16_ROWS", "DW_KSPLIT_ROWS", "DEC_TS", "CHAIN_TS", "DW_TS", "DEC_SKIP", "CHAIN_SKIP", "RANK_SKIP"};
    bool ok = false;
    for (const char* k : known) ok = ok || strcmp(k, name) == 0;
    if (!ok) return fail(AAE_EINVAL, "unknown option (aae_options, csrc/abi_model.h, lists them)");
    if (value) g_options[name] = value; else g_options.erase(name);
    return AAE_OK;
}
const char* aae_last_error(void) { return g_err.c_str(); }

int aae_arena_bytes(const aae_config* cfg, size_t* bytes_out) {
    TRY(validate(cfg));
    if (!bytes_out) return fail(AAE_EINVAL, "bytes_out is NULL");
    aae_model tmp; memset((void*)&tmp, 0, sizeof(tmp)); tmp.cfg = *cfg;
    read_options(tmp.opt);
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
    read_options(m->opt);                // every switch of the library, once per handle (abi_model.h aae_options)
    size_t need = layout(m, static_cast<char*>(arena_dev), false);
    if (need > arena_bytes) { delete m; return fail(AAE_ENOMEM, "arena smaller than aae_arena_bytes()"); }
    m->base = static_cast<char*>(arena_dev); m->bytes = need;
    m->alpha_mode = cfg->activation == AAE_ACT_SELU;
    m->vae = cfg->model_kind == 3; m->vae_bwd = false; m->vae_cut = false;
    m->bf16 = cfg->dtype == 1;
    m->blocked_ok = cfg->blocked_output == 1;
    m->blocked_any = m->opt.blocked_any;
    m->split_any = m->opt.split_any;
    m->x3_gemm = !m->bf16;
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
        // (a decoder input [z | condition | 1] of up to 2 x 208 columns: two k-parts on the 4-row kernel; the VAE's programs keep 208)
        m->use_chain = (m->h + 1 <= 208) && (m->c + 1 <= 208) && (m->cp + 1 <= (cfg->model_kind == 3 ? 208 : 2 * 208)) &&
                       !m->opt.no_chain;      // (the code itself has to fit a slot: only the condition block may exceed it)
        if (m->use_chain && (hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 kCSlots * kCR * kCL * (int)sizeof(float)) != hipSuccess ||
                             hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 kCSlots * kCR * kCL * (int)sizeof(float)) != hipSuccess))
            m->use_chain = false;
        // (the r6 activation classes live in chain_kernel<.., true> alone: device_common.h act_fwd)
        m->act_nm = cfg->activation > AAE_ACT_LEAKYRELU;
        if (m->use_chain && m->act_nm &&
            (hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 kCSlots * kCR * kCL * (int)sizeof(float)) != hipSuccess ||
             hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 kCSlots * kCR * kCL * (int)sizeof(float)) != hipSuccess))
            m->use_chain = false;
        m->use_chain4 = m->use_chain && !m->act_nm && !m->opt.chain16 &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(chain4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            kCSlots * kCR * kCL * (int)sizeof(float)) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(chain4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            kCSlots * kCR * kCL * (int)sizeof(float)) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(chain4_kernel<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            kCSlots * kCR * kCL * (int)sizeof(float)) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(chain4_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            kCSlots * kCR * kCL * (int)sizeof(float)) == hipSuccess;
        if (m->use_chain && m->cp + 1 > 208 && !m->use_chain4) m->use_chain = false;
        m->x16_rows = m->opt.x16_rows;
        m->dw_ksplit_rows = 256;                       // (the handle is zero-filled after construction: no default member values)
        if (m->opt.dw_ksplit_rows >= 0) m->dw_ksplit_rows = m->opt.dw_ksplit_rows;
        if (!m->opt.no_item_count) {
            void* hp = nullptr; void* dp = nullptr;
            if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
                m->cnt_host = static_cast<int*>(hp); m->cnt_host_dev = static_cast<int*>(dp); *m->cnt_host = 0;
            } else { if (hp) (void)hipHostFree(hp); (void)hipGetLastError(); }
        }
        m->x16_ok = m->use_chain4 && m->FXi[P_W2] != nullptr &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(chain16x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kX16Lds) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(chain16x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kX16Lds) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(chain16x3_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kX16Lds) == hipSuccess;
        m->force_unfused = cfg->unfused_decoder == 1;   // debugging / A-B switch: unfused_decoder = 1 keeps the 3-kernel path
        if (m->fused_ok) {
            const int maxlds = 160 * 1024;
            hipError_t e1 = aae_attr2(reinterpret_cast<const void*>(dec_fused_kernel<4>), reinterpret_cast<const void*>(dec_fused_kernel<4, kDecFused, true>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
            hipError_t e2 = aae_attr2(reinterpret_cast<const void*>(dec_fused_kernel<7>), reinterpret_cast<const void*>(dec_fused_kernel<7, kDecFused, true>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
            hipError_t e3 = aae_attr2(reinterpret_cast<const void*>(dec_fused_kernel<13>), reinterpret_cast<const void*>(dec_fused_kernel<13, kDecFused, true>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
            hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_bucket_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds - 2048);
            m->bucket_wide_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_bucket_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds - 16384) == hipSuccess;
            {   //  the wide bucket builder keeps other workgroups off its CU by its register count (buckets.h: a clobber of
                // v127) on top of its LDS claim: say so once if a compiler stops honouring the clobber - results do not depend on it
                hipFuncAttributes fa;
                static bool told = false;
                if (m->bucket_wide_ok && !told && hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(tile_bucket_wide_kernel)) == hipSuccess && fa.numRegs < 121) {
                    told = true;
                    fprintf(stderr, "aaerec: tile_bucket_wide_kernel allocates %d vector registers, not the 128 it claims to keep its CU to itself (buckets.h)\n", fa.numRegs);
                }
                (void)hipGetLastError();
            }
            if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) m->fused_ok = false;
            if (m->bf16 &&
                (hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_bf16_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds) != hipSuccess ||
                 hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_bf16_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds) != hipSuccess ||
                 hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fused_bf16_kernel<13>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds) != hipSuccess))
                m->fused_ok = false;
        }
    }
    m->late_enabled = !m->opt.no_late_join;
    m->early_enabled = true;
    m->early_any = m->opt.early_any;        // (tests: the early prefetch at every batch size)
    m->rank_ok = m->fused_ok && m->use_chain4 && !m->vae && !m->opt.no_rank_fused && rank_set_attributes();
    m->w1_big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(w1_item_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(sizeof(int) * w1_items_lds_words(16384))) == hipSuccess;
    (void)hipGetLastError();
    m->grad_scale = 1.f;
    m->rng_row0 = 0; m->rng_global = 0;
    m->Gt = m->G.p;
    m->split_ok = false; m->opt_pending = false; m->side = nullptr; m->ev_crit = m->ev_opt = nullptr;
    // workgroups of the deferred optimiser launch.  r2: half the CUs (it had 50 us of slack then).  Since the critical launch
    // runs on the bf16 matrix cores (r3, 116 -> 84 us) the NEXT step waits for this launch; measured at C3, batch 100
    // (r2-r3 sweep of the launch's width, ms/step), with the fp32 deferred kernel: 112 0.306 | 128 0.291 | 144 0.274 | 160 0.274 | 176 0.284
    // | 192 0.285 | 224 0.312; with the bf16-emulated one (dec_opt_x3_kernel: 160 -> 138 us on 128 workgroups): 112 0.279 |
    // 128 0.273 | 144 0.276 | 160 0.278 | 176 0.286.  Half the CUs again (set below once x3_ok is known; 5/8 without it):
    // beyond that the chain / weight-gradient kernels it runs beside lose more than the launch gains.
    m->split_wgs = std::max(1, (m->n_cu * 5) / 8);
    m->pf_armed = m->pf_built = m->pf_pending = false; m->pf_step = -1; m->hstep = 0; m->flushed_hstep = -1; m->ev_head = m->ev_pf = nullptr;
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
                  hipEventCreateWithFlags(&m->ev_bk, evflags) == hipSuccess &&
                  hipEventCreateWithFlags(&m->ev_end, evflags) == hipSuccess;
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
    // (bf16 mode too: its output layer runs the same kernels with every operand rounded to bf16 - DecFusedArgs::one_term - when
    //  they are available; dec_fused_bf16.h's own pair otherwise, or with AAE_NO_BF16_X3)
    if (side_ok && m->fused_ok && m->split_wgs > 0 && 
        (size_t)((m->N + kTI - 1) / kTI) * kTI * (size_t)std::min(m->R, 16 * kMB) * sizeof(float) < (size_t)0x7FFFFFF0u) {
        bool ok = true;
        ok = ok && aae_attr2(reinterpret_cast<const void*>(dec_fused_kernel<4, kDecCrit>), reinterpret_cast<const void*>(dec_fused_kernel<4, kDecCrit, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_fused_kernel<7, kDecCrit>), reinterpret_cast<const void*>(dec_fused_kernel<7, kDecCrit, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_fused_kernel<13, kDecCrit>), reinterpret_cast<const void*>(dec_fused_kernel<13, kDecCrit, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_fused_kernel<4, kDecOpt>), reinterpret_cast<const void*>(dec_fused_kernel<4, kDecOpt, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_fused_kernel<7, kDecOpt>), reinterpret_cast<const void*>(dec_fused_kernel<7, kDecOpt, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_fused_kernel<13, kDecOpt>), reinterpret_cast<const void*>(dec_fused_kernel<13, kDecOpt, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_fused_kernel<4, kDecOptAcc>), reinterpret_cast<const void*>(dec_fused_kernel<4, kDecOptAcc, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_fused_kernel<7, kDecOptAcc>), reinterpret_cast<const void*>(dec_fused_kernel<7, kDecOptAcc, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_fused_kernel<13, kDecOptAcc>), reinterpret_cast<const void*>(dec_fused_kernel<13, kDecOptAcc, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_opt_blocks_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_opt_blocks_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_opt_blocks_kernel<13>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        if (!m->bf16) m->split_ok = ok;
        m->x3_ok = ok
                && aae_attr2(reinterpret_cast<const void*>(dec_crit_x3_kernel<4>), reinterpret_cast<const void*>(dec_crit_x3_kernel<4, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_crit_x3_kernel<7>), reinterpret_cast<const void*>(dec_crit_x3_kernel<7, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_crit_x3_kernel<13>), reinterpret_cast<const void*>(dec_crit_x3_kernel<13, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_crit_x3_kernel<13, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_opt_blocks_x3_kernel<4>), reinterpret_cast<const void*>(dec_opt_blocks_x3_kernel<4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_opt_blocks_x3_kernel<7>), reinterpret_cast<const void*>(dec_opt_blocks_x3_kernel<7, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_opt_blocks_x3_kernel<13>), reinterpret_cast<const void*>(dec_opt_blocks_x3_kernel<13, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_opt_blocks_x3_kernel<13, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        (void)hipGetLastError();
        if (m->bf16) {
            m->bf16_x3 = m->x3_ok && ok;
            if (!m->bf16_x3) m->x3_ok = false;
            else m->split_ok = true;
            // The one-term instantiations (first terms only, ONE matrix instruction per product; AAE_NO_BF16_ONE: the three-term
            // kernels multiplying five zero terms).  Both launches get faster - critical 35 -> 27 us, deferred 85 -> 74 us at C2 -
            // and for most of r4 the STEP got slower, 0.173 -> 0.186 ms (profiles/r4_c2_bf16_one_term.txt): the one-term deferred
            // kernel needs 78 VGPRs instead of 126, so six of its waves fit a SIMD, its 1024-thread workgroup no longer fills
            // its CU, and the dispatcher deals the step's own weight-gradient / gather workgroups onto the CUs where it streams
            // 24 B per parameter (the weight-gradient launch beside it: 32 us instead of 11).  Its launch now claims 150 KB of
            // LDS (abi_output_layer.h), which no other workgroup of the step fits beside: 0.1648 ms at the old width, 0.162-0.164
            // on 48-64 workgroups (tools/debug/bf16_one_sweep.sh; a claim of 120 KB leaves the 24 KB weight-gradient workgroups
            // in: 0.1865).
            m->bf16_one = m->bf16_x3
                && aae_attr2(reinterpret_cast<const void*>(dec_crit_x3_kernel<4, false, true>), reinterpret_cast<const void*>(dec_crit_x3_kernel<4, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_crit_x3_kernel<7, false, true>), reinterpret_cast<const void*>(dec_crit_x3_kernel<7, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_crit_x3_kernel<13, false, true>), reinterpret_cast<const void*>(dec_crit_x3_kernel<13, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_opt_x3_kernel<4, true>), reinterpret_cast<const void*>(dec_opt_x3_kernel<4, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_opt_x3_kernel<7, true>), reinterpret_cast<const void*>(dec_opt_x3_kernel<7, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_opt_x3_kernel<13, true>), reinterpret_cast<const void*>(dec_opt_x3_kernel<13, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_opt_x3_kernel<4>), reinterpret_cast<const void*>(dec_opt_x3_kernel<4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_opt_x3_kernel<7>), reinterpret_cast<const void*>(dec_opt_x3_kernel<7, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                && aae_attr2(reinterpret_cast<const void*>(dec_opt_x3_kernel<13>), reinterpret_cast<const void*>(dec_opt_x3_kernel<13, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
            (void)hipGetLastError();
        }
        if (m->x3_ok) {
            // Workgroups of the deferred launch (batches of one fused launch).  It has to end before the step does (the next
            // step opens behind it), and every CU it holds is one the step's own launches share with it: wide enough to take
            // ~80 us - every shape's step has that much work left behind the critical launch - and never more than 9/16 of
            // the chip.  A tile of 32 items costs a workgroup 3.3 us + 13 ns per hidden unit (measured: 5.9 us at 200, 4.6 us
            // at 100).  tools/debug/sweep_split_wgs*.sh, r3: C3 (3125 tiles, 200) 128 -> 0.2652-0.2733 ms/step, 144 ->
            // 0.2651-0.2655, 160 -> 0.2683, 176 -> 0.282; C2 (1469 tiles, 100) 64 -> 0.1806, 80 -> 0.1710, 96 -> 0.1721,
            // 128 -> 0.177, 160 -> 0.185.
            const int ntiles = (m->N + 31) / 32;
            const double t_tile = (3.3 + 0.0131 * m->h) * (m->bf16_one ? 0.87 : 1.0);     // (one-term products: 4.0 us at 100)
            int w = (int)(ntiles * t_tile / (m->bf16_one ? 165.0 : 82.0) / 8.0 + 0.5) * 8;      // (C3's shape in bf16: 0.2230 / 0.2264 / 0.2317 / 0.2374 ms on 96 / 112 / 128 / 144)
            w = std::max(w, std::min(ntiles, m->bf16_one ? 48 : 64));
            // r4, late join (abi_model.h: the launch may run on into the next step's forward pass, so it need not end with the
            // step): half the chip at most - C3 0.2493 (early join, 144) -> 0.2430 ms/step on 128; 0.2515 / 0.2483 / 0.2497 /
            // 0.2492 on 112 / 120 / 136 / 144 (sharp: 124 -> 0.2459, 132 -> 0.2503); the early join on 128: 0.2608.  C2 (formula:
            // 80) is flat from 64 to 96 either way (tools/debug/late_join_sweep*.sh).
            // r5: the deferred launch no longer requests the 416 float4 slots beyond a tile's span (it read 1.25x the layer): 139.7 ->
            // 130.9 us on 128 workgroups, and the optimum moved down with it - same box, C3: 0.2469 / 0.2359 / 0.2368 / 0.2386 / 0.2378 /
            // 0.2412 ms/step on 104 / 112 / 120 / 124 / 128 / 136 (tools/debug/sweep_split_wgs.sh): 15/32 of the chip
            const int cap = m->late_enabled ? m->n_cu * 15 / 32 : m->n_cu * 9 / 16;
            m->split_wgs = std::max(1, std::min(w, cap));
        }
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

#ifdef W1_TS
int aae_debug_w1_ts(unsigned long long* out, int n) {      // (debug builds only)
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(aae::w1_ts), sizeof(unsigned long long) * (size_t)n);
}
#endif
int aae_destroy(aae_handle h) {
    if (!h) return AAE_OK;
    if (h->side) {
        (void)hipStreamSynchronize(h->side);     // the arena is the caller's: nothing of ours may still write it
        (void)hipStreamDestroy(h->side);
    }
    if (h->dp_scratch) (void)hipFree(h->dp_scratch);
    if (h->cnt_host) (void)hipHostFree(h->cnt_host);
    if (h->ev_crit) (void)hipEventDestroy(h->ev_crit);
    if (h->ev_opt) (void)hipEventDestroy(h->ev_opt);
    if (h->ev_head) (void)hipEventDestroy(h->ev_head);
    if (h->ev_pf) (void)hipEventDestroy(h->ev_pf);
    if (h->ev_bk) (void)hipEventDestroy(h->ev_bk);
    if (h->ev_end) (void)hipEventDestroy(h->ev_end);
    if (h->prof_ev) {
        for (int k = 0; k < AAE_K_N; ++k)
            for (auto& pr : h->prof_ev[k]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        delete[] h->prof_ev;
    }
    delete h;
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
    if (next->generation == 0) { h->pf_armed = false; return AAE_OK; }      // (no content id: no step could ever match it)
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
    h->spec_tab_ok = false;             // (the table entry written a step early was computed with the old rates)
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


}  // extern "C"

#include "abi_condition.h"
#include "abi_state_dict.h"
#include "abi_step_open.h"
#include "abi_output_layer.h"
#include "abi_step_phases.h"
#include "abi_predict.h"
#include "abi_data_parallel.h"

#include "dp_step.h"
#include "ipc_collectives.h"
