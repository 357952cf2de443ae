/*
 * aaerec_hip.h - C ABI of libaaerec_hip.so: the MI355X (gfx950) implementation of the
 * adversarial-autoencoder training step of lgalke/aae-recommender.
 *
 * The reference has no FFI boundary of its own: the hot path sits behind duck-typed Python
 * protocols.  Each entry point below names the reference interface it
 * replaces (file:line, relative to the reference checkout).  INTEGRATION.md shows the ctypes
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative AAE_E* code; aae_last_error() gives
 *     text.  Nothing throws, nothing calls exit().
 *   - all `*_dev` pointers are device pointers on the current HIP device; work is enqueued on
 *     the `stream` argument (a hipStream_t passed as void*; NULL = default stream).  No entry
 *     point synchronises the device except aae_read_losses / aae_load_* / aae_store_*.
 *   - one handle per model replica and per rank; a handle is not thread-safe.
 *   - the caller owns the arena (one device allocation of aae_arena_bytes() bytes, 256-byte
 *     aligned); the library lays out parameters, optimiser state and activations inside it
 *     (aae_tensor_info gives every view), so a host framework can alias them (e.g. as
 *     torch tensors for RCCL collectives) without copies.
 */
#ifndef AAEREC_HIP_H
#define AAEREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4 (r6): aae_set_option, aae_ipc_create / _init / _destroy, AAE_ACT_SOFTPLUS .. AAE_ACT_HARDSWISH; no structure changed since 3 */
#define AAE_ABI_VERSION 4

/* error codes */
#define AAE_OK 0
#define AAE_EINVAL (-1)   /* bad argument / unsupported configuration */
#define AAE_ENOMEM (-2)   /* arena too small */
#define AAE_EHIP (-3)     /* a HIP runtime call failed */
#define AAE_ESTATE (-4)   /* call sequence violated (e.g. decode before encode) */

/* activation ids: getattr(nn, activation) in aaerec/aae.py:110 */
enum { AAE_ACT_RELU = 0, AAE_ACT_SELU = 1, AAE_ACT_TANH = 2, AAE_ACT_SIGMOID = 3,
       AAE_ACT_ELU = 4, AAE_ACT_LEAKYRELU = 5,
       /* r6: the other parameter-free element-wise torch.nn classes, at their default arguments.  The layer programs of a model
          with one of them run on the 16-row chain kernel (no 4-row / wide-batch kernel, no fused ranking launch -
          aae_predict_topk takes its two-kernel form): supported, not tuned */
       AAE_ACT_SOFTPLUS = 6, AAE_ACT_HARDTANH = 7, AAE_ACT_RELU6 = 8, AAE_ACT_CELU = 9, AAE_ACT_SOFTSIGN = 10,
       AAE_ACT_HARDSIGMOID = 11, AAE_ACT_LOGSIGMOID = 12, AAE_ACT_SOFTSHRINK = 13, AAE_ACT_HARDSHRINK = 14,
       AAE_ACT_IDENTITY = 15,
       /* ... not monotone: the derivative is evaluated through a branch bit kept in the stored output's last place */
       AAE_ACT_GELU = 16, AAE_ACT_SILU = 17, AAE_ACT_MISH = 18, AAE_ACT_HARDSWISH = 19, AAE_ACT_COUNT = 20 };
/* encoder output activation: PRIOR_ACTIVATIONS, aaerec/aae.py:97-101 */
enum { AAE_FINAL_LINEAR = 0, AAE_FINAL_SOFTMAX = 1, AAE_FINAL_SIGMOID = 2 };
/* TORCH_OPTIMIZERS, aaerec/aae.py:216-219 */
enum { AAE_OPT_ADAM = 0, AAE_OPT_SGD = 1 };
/* where dropout masks and z_real come from */
enum { AAE_RNG_INJECT = 0,   /* caller supplies them (parity tests, reference-RNG mode) */
       AAE_RNG_DEVICE = 1 }; /* counter-based generator inside the kernels */
/* prior for AAE_RNG_DEVICE: PRIOR_SAMPLERS, aaerec/aae.py:90-94 */
enum { AAE_PRIOR_GAUSS = 0, AAE_PRIOR_CATEGORICAL = 1, AAE_PRIOR_BERNOULLI = 2 };
/* which model of the reference the handle trains (cfg.model_kind) */
enum { AAE_MODEL_AAE = 0,    /* AdversarialAutoEncoder, aae.py:587-870 */
       AAE_MODEL_AE = 1,     /* plain AutoEncoder (aae.py:221-458): the step ends after the encoder backward,
                                aae_disc_step / aae_gen_step are errors */
       AAE_MODEL_VAE = 3 };  /* VAE (vae.py:47-266): see aae_vae_step */
/* arithmetic of the GEMM-shaped products (cfg.dtype) */
enum { AAE_DTYPE_F32 = 0,    /* fp32 throughout (the reference's precision) */
       AAE_DTYPE_BF16 = 1 }; /* bf16 matrix-core inputs, fp32 accumulation, fp32 master weights and Adam (config C2) */
/* what happens to gradients */
enum { AAE_GRAD_FUSED = 0,   /* optimiser update fused into the weight-gradient kernels */
       AAE_GRAD_EXPORT = 1 };/* gradients are materialised (data-parallel: all-reduce, then
                                aae_apply_updates) */

/* AdversarialAutoEncoder.__init__ keyword arguments, aaerec/aae.py:588-606, plus sizes that
 * the reference infers in fit() (aae.py:773-793) and build-only knobs. */
typedef struct aae_config {
    int32_t abi_version;      /* AAE_ABI_VERSION */
    int32_t n_items;          /* X.shape[1] */
    int32_t n_hidden;
    int32_t n_code;
    int32_t cond_inc;         /* conditions.size_increment(), 0 without conditions */
    int32_t max_batch;        /* largest number of rows per step / per predict call */
    int32_t max_nnz;          /* largest number of CSR entries in one batch */
    int32_t activation;       /* AAE_ACT_* */
    int32_t enc_final;        /* AAE_FINAL_* */
    int32_t optimizer;        /* AAE_OPT_* */
    int32_t normalize_inputs; /* F.normalize(inp, 1) at aae.py:132-133 */
    int32_t rng_mode;         /* AAE_RNG_* */
    int32_t prior;            /* AAE_PRIOR_* (device rng only) */
    int32_t grad_mode;        /* AAE_GRAD_* */
    float dropout1, dropout2; /* dropout=(.2,.2) */
    float gen_lr, reg_lr;
    float prior_scale;        /* used when has_prior_scale != 0 (aae.py:717-718) */
    int32_t has_prior_scale;
    uint64_t seed;            /* device rng */
    /* build-time knobs of the model (named fields since ABI version 2; they travelled in reserved[0..6] before) */
    int32_t unfused_decoder;  /* 1: keep the decoder output layer on the three-kernel path (A/B measurements, tests) */
    int32_t dp_world;         /* AAE_GRAD_EXPORT: number of data-parallel peers whose packed rows aae_w1_import may
                               * receive (0 = 1) */
    int32_t model_kind;       /* AAE_MODEL_*: which of the reference's models the handle trains */
    int32_t dtype;            /* AAE_DTYPE_*: arithmetic of the matrix-core products (master weights / Adam stay fp32) */
    int32_t blocked_output;   /* 1: batches beyond 112 rows run the fused output layer over row blocks */
    int32_t dense_noise;      /* 1: room for aae_set_input_noise (DenoisingAutoEncoder corrupt='gauss'; needs
                               * model_kind = AAE_MODEL_AE, fp32, fused optimiser) */
    int32_t reserved[2];      /* must be zero */
} aae_config;

typedef struct aae_model* aae_handle;

/* A batch = `n_rows` rows picked out of a CSR matrix that is already resident in HBM
 * (replaces X_shuf[start:end].toarray() + torch.FloatTensor(X).cuda(), aae.py:823,751-754).
 * rows_dev == NULL means rows row_start .. row_start+n_rows-1.  Column indices must be unique
 * within a row (canonical CSR, as scipy's tocsr() / sum_duplicates() produce: a repeated (row, item) pair counts
 * once in the BCE target and in the first layer's weight gradient) and values in [0,1] (the reference's BCE raises
 * otherwise). */
typedef struct aae_batch {
    const int64_t* indptr_dev;
    const int32_t* indices_dev;
    const float* values_dev;
    const int32_t* rows_dev;
    int32_t row_start;
    int32_t n_rows;
    int32_t nnz_bound;        /* upper bound on the entries of these rows (<= cfg.max_nnz) */
    int32_t max_row_nnz;      /* upper bound on the entries of any one of these rows; 0 = unknown */
    int64_t generation;       /* ABI 3: caller-supplied id of the CONTENT these pointers name (the CSR arrays and rows_dev): a new
                               * value whenever any of them is rewritten in place or freed and allocated again.  0 = none given.
                               * Only aae_prefetch_batch's matching reads it: a step takes the work built ahead for a named
                               * batch only when pointers, row window AND a non-zero generation are equal. */
} aae_batch;

/* Injected randomness for one partial_fit, in the reference's draw order:
 * keep masks (uint8 0/1, dense row-major [rows][width]) for
 *   0 enc.drop1 1 enc.drop2 2 dec.drop1 3 dec.drop2            (ae_step,  aae.py:687-691)
 *   4 disc.drop1 5 disc.drop2 on z_real, 6,7 the same on z_fake (disc_step, aae.py:724)
 *   8 enc.drop1 9 enc.drop2 10 disc.drop1 11 disc.drop2         (gen_step, aae.py:736-737)
 * and z_real [rows][n_code] BEFORE prior_scale.  A NULL mask = keep everything. */
typedef struct aae_rng_inject {
    const uint8_t* masks_dev[12];
    const float* z_real_dev;
} aae_rng_inject;

/* tensor ids for aae_tensor_info.  Linear layers are stored "augmented": row o holds
 * weight[o, 0:in] followed by bias[o] at column `in`, rows padded to `ld` floats (zeros).
 * The two vocabulary-sized layers are item-major: ENC_W1T row i = enc.lin1.weight[:, i]
 * (bias separate, ENC_B1); DEC_V3 row i = dec.lin3.weight[i, :] ++ dec.lin3.bias[i]. */
enum {
    AAE_T_ENC_W1T = 0, AAE_T_ENC_B1, AAE_T_ENC_W2, AAE_T_ENC_W3,
    AAE_T_DEC_V1, AAE_T_DEC_V2, AAE_T_DEC_V3,
    AAE_T_DISC_D1, AAE_T_DISC_D2, AAE_T_DISC_D3,
    AAE_T_N_PARAMS,
    /* optimiser state: base + 2*param_id (+1 for exp_avg_sq) */
    AAE_T_ADAM_ENC = 16,   /* enc_optim  (aae.py:800), params 0..3 */
    AAE_T_ADAM_GEN = 32,   /* gen_optim  (aae.py:803), params 0..3 */
    AAE_T_ADAM_DEC = 48,   /* dec_optim  (aae.py:801), params 4..6 -> slots 0..2 */
    AAE_T_ADAM_DISC = 64,  /* disc_optim (aae.py:804), params 7..9 -> slots 0..2 */
    /* gradients (AAE_GRAD_EXPORT only), same shapes as the parameters */
    AAE_T_GRAD = 80,       /* + param_id */
    /* activations useful to callers */
    AAE_T_ACT_Z = 96,      /* encoder output of the last encode call [rows][n_code] */
    AAE_T_ACT_LOSSES = 97, /* float[4]: R, D, G of the last step, spare */
    AAE_T_ACT_A1 = 98,     /* first-layer pre-activations of the last encode [rows][n_hidden] */
    AAE_T_ACT_DH2 = 100,   /* the decoder's last hidden activation [rows][n_hidden + 1] (last column = 1: the bias input) */
    AAE_T_ACT_DA2 = 101,   /* dL/d(ACT_DH2) written by aae_output_layer_step, same layout */
    AAE_T_ACT_GA1 = 102,   /* dL/d(ACT_A1) of the last encoder backward [rows][n_hidden] */
    AAE_T_ACT_DZC = 99     /* dL/d(decoder input) of the last ae phase [rows][n_code + cond_inc]: columns n_code.. are
                            * the gradient of the condition block handed to aae_step (trainable conditions) */
};

typedef struct aae_tensor {
    size_t byte_offset;   /* from the arena base */
    int64_t rows, cols;   /* logical shape (cols includes the bias column where augmented) */
    int64_t ld;           /* row stride in floats */
} aae_tensor;

int aae_abi_version(void);
const char* aae_last_error(void);
/* A switch of the library for the handles created FROM NOW ON (value NULL: back to the environment variable AAE_<name>, which
   is what a handle reads otherwise).  Names: struct aae_options, csrc/abi_model.h - paths the parity suites force (NO_CHAIN,
   SPLIT_ANY, BLOCKED_ANY, X16_ROWS, DW_KSPLIT_ROWS ...) and the diagnostics of debug runs (DEC_TS ...).  No counterpart in
   the reference (it has no switches: one eager path); a handle reads its switches once, in aae_create. */
int aae_set_option(const char* name, const char* value);

/* construction: the nets + 4 optimisers of fit(), aae.py:782-804 */
int aae_arena_bytes(const aae_config* cfg, size_t* bytes_out);
int aae_create(const aae_config* cfg, void* arena_dev, size_t arena_bytes, void* stream,
               aae_handle* out);
int aae_destroy(aae_handle h);
int aae_tensor_info(aae_handle h, int tensor_id, aae_tensor* out);
/* The Adam updates of ENC_W1T rows whose items are absent from a batch are deferred (they do not
 * depend on the batch) and replayed when a row is next read.  aae_sync replays everything that
 * is pending, after which the arena views of ENC_W1T / ADAM_ENC / ADAM_GEN hold the values an
 * eager implementation would.  aae_load_* / aae_store_* call it themselves. */
int aae_sync(aae_handle h, void* stream);
/* Device RNG (AAE_RNG_DEVICE) under data parallelism: dropout masks and the prior sample are drawn from a counter
 * generator keyed by (seed, step, stream, ROW, column).  With the ranks' handles created with the SAME seed and each
 * told which rows of the global batch it holds - rows [row_offset, row_offset + n_rows) of global_rows - the ranks
 * together draw exactly what one process draws for the whole batch, so 1-GPU and N-GPU runs of a seed are comparable
 * (SURVEY 8e).  (0, 0) = a batch of its own (the default). */
int aae_set_rng_rows(aae_handle h, int64_t row_offset, int64_t global_rows);
/* Call after writing PARAMETER tensors through the arena views of aae_tensor_info (instead of aae_load_linear): the
 * library keeps derived copies of the hidden layers' weights (transposed, for the backward layer chains) and
 * re-derives them before their next use. */
int aae_params_changed(aae_handle h);
/* gen_lr / reg_lr as the exact Python doubles (aae_config carries them as float32) */
int aae_set_lr(aae_handle h, double gen_lr, double reg_lr);

/* nn.Linear state in the reference's state_dict layout (weight [out,in] row-major, bias
 * [out]), host pointers.  Synchronous. net: 0 enc 1 dec 2 disc; layer: 1..3. */
int aae_load_linear(aae_handle h, int net, int layer, const float* weight_host,
                    const float* bias_host);
int aae_store_linear(aae_handle h, int net, int layer, float* weight_host, float* bias_host);
/* optimiser state_dict (exp_avg / exp_avg_sq in [out,in] + [out] layout, step count).
 * which: 0 enc_optim 1 dec_optim 2 gen_optim 3 disc_optim. */
int aae_load_adam(aae_handle h, int which, int layer, const float* m_w, const float* v_w,
                  const float* m_b, const float* v_b, int64_t step);
int aae_store_adam(aae_handle h, int which, int layer, float* m_w, float* v_w, float* m_b,
                   float* v_b, int64_t* step);

/* AdversarialAutoEncoder.partial_fit, aae.py:745-766, without conditions or with a constant
 * concatenated condition block cond_dev [rows][cond_inc] (PretrainedWordEmbeddingCondition,
 * condition.py:345-369).  inject may be NULL when cfg.rng_mode == AAE_RNG_DEVICE. */
int aae_step(aae_handle h, const aae_batch* batch, const float* cond_dev,
             const aae_rng_inject* inject, void* stream);

/* The same step cut at the condition boundary, for arbitrary ConditionList plugins whose
 * encode_impose runs in the host framework (condition.py:90-99):
 *   aae_ae_encode          Encoder forward in train mode            (aae.py:687)
 *   [host: zc = conditions.encode_impose(z, cond)]                   (aae.py:688-690)
 *   aae_ae_decode_backward Decoder forward, BCE, decoder backward + dec_optim.step;
 *                          writes dL/dzc [rows][n_code+cond_inc]     (aae.py:692-707)
 *   [host: backprop dzc through the conditions -> dz; conditions.step()]
 *   aae_ae_encoder_backward encoder backward + enc_optim.step        (aae.py:703-706)
 *   aae_disc_gen           disc_step + gen_step                      (aae.py:713-743)
 * The aae_rng_inject handed to a phase stays in force for the later phases of the same step;
 * its device buffers must stay valid until the step's kernels have run. */
int aae_ae_encode(aae_handle h, const aae_batch* batch, const aae_rng_inject* inject,
                  float* z_out_dev, void* stream);
int aae_ae_decode_backward(aae_handle h, const float* zc_dev, int64_t zc_ld,
                           const aae_rng_inject* inject, float* dzc_out_dev, void* stream);
int aae_ae_encoder_backward(aae_handle h, const float* dz_dev, int64_t dz_ld, void* stream);
int aae_disc_gen(aae_handle h, const aae_rng_inject* inject, void* stream);
/* DecodingRecommender.partial_fit, aae.py:489-517 ("only the decoder part of the AAE, basically
 * 2-MLP"): the decoder maps an input block computed by the host's condition plugins to the items.
 *   zin_dev [batch->n_rows][n_code + cond_inc]  the encoded / imposed conditions (aae.py:494-502)
 *   batch                                        the targets y as CSR rows (aae.py:506-507)
 * Decoder forward in train mode, BCE against the batch, decoder backward + its optimiser step
 * (learning rate = cfg.gen_lr); dzin_out_dev (may be NULL) receives dL/dzin for trainable
 * conditions.  Uses masks_dev[2], masks_dev[3] of `inject`.  Predict = aae_decode. */
int aae_decoder_step(aae_handle h, const aae_batch* batch, const float* zin_dev, int64_t zin_ld,
                     const aae_rng_inject* inject, float* dzin_out_dev, void* stream);
/* VAE.partial_fit / VAE.predict, vae.py:147-186, 229-266.  The model must have been created with
 * cfg.model_kind = AAE_MODEL_VAE: ENC_W1T/B1 = fc1, ENC_W3 = [fc21; fc22] (2 * n_code rows: mu, then logvar), DEC_V1 = fc3,
 * DEC_V3 = fc4; ENC_W2 / DEC_V2 / the discriminator are unused; cfg.gen_lr is the single learning rate and
 * cfg.dropout must be (0, 0).  loss = mean BCE (losses[0]) + KL sum (losses[1]), vae.py:132-145.
 *   cond_dev  constant concatenated condition block [rows][cond_inc] or NULL
 *   eps_dev   [rows][n_code] standard-normal draws of reparametrize() (vae.py:115-118); NULL = counter generator
 *             (required when cfg.rng_mode == AAE_RNG_INJECT).  The reference samples eps in predict as well. */
int aae_vae_step(aae_handle h, const aae_batch* batch, const float* cond_dev, const float* eps_dev, void* stream);
int aae_vae_predict(aae_handle h, const aae_batch* batch, const float* cond_dev, const float* eps_dev,
                    float* out_dev, int64_t out_ld, void* stream);
/* The VAE step cut at the condition boundary, for ConditionList plugins whose encode_impose runs in the host framework
 * (vae.py:120-130: `z = self.conditions.encode_impose(z, condition_data)` between reparametrize and decode):
 *   aae_vae_encode            x -> fc1 -> (mu, logvar) -> z = mu + eps * exp(logvar / 2) -> z_out_dev [rows][n_code];
 *                             train != 0 opens a training step, 0 = the forward half of VAE.predict (then aae_decode)
 *   [host: zc = conditions.encode_impose(z, c)]
 *   aae_vae_decode_backward   fc3 -> fc4 -> BCE, backward to dzc_out_dev [rows][n_code + cond_inc], fc3 / fc4 updates
 *   [host: backprop dzc through the conditions -> dz; conditions.step()]                              (vae.py:175-181)
 *   aae_vae_encoder_backward  reparametrize' + the KL gradient -> [fc21; fc22] -> fc1, their updates */
int aae_vae_encode(aae_handle h, const aae_batch* batch, const float* eps_dev, float* z_out_dev, int32_t train, void* stream);
int aae_vae_decode_backward(aae_handle h, const float* zc_dev, int64_t zc_ld, float* dzc_out_dev, void* stream);
int aae_vae_encoder_backward(aae_handle h, const float* dz_dev, int64_t dz_ld, void* stream);
/* The ae phase (aae.py:676-711) cut at the decoder's output layer, for vocabulary-sharded data parallelism: the output
 * layer holds ~all of the decoder's parameters and every item receives gradient, so instead of exchanging its
 * [n_items][n_hidden + 1] gradient each rank owns the rows of a slice of the items (a second handle created with
 * n_items = slice size) and the ranks exchange hidden activations, [global rows][n_hidden], instead:
 *   aae_ae_forward         encoder + decoder hidden layers on this rank's documents -> AAE_T_ACT_DH2
 *   [all-gather ACT_DH2 of all ranks into the slice handle's ACT_DH2]
 *   aae_output_layer_step  on the slice handle, batch = the GLOBAL batch restricted to the slice's items (ids rebased):
 *                          logits, BCE (scale: aae_set_grad_scale(slice items / all items)), dV3 + dec_optim on the
 *                          slice's rows, dL/d(dh2) partial -> AAE_T_ACT_DA2; batch = NULL continues the step of
 *                          aae_ae_forward on the same handle (single device: forward + output_layer_step(NULL) +
 *                          backward == the ae phase of aae_step)
 *   [reduce-scatter ACT_DA2 over the ranks -> this rank's rows]
 *   aae_ae_backward        decoder hidden backward + encoder backward + their updates / exported gradients from
 *                          dA2_dev (NULL = this handle's ACT_DA2), leading dimension = ACT_DH2's
 * followed by aae_disc_step / aae_gen_step as in the replicated scheme.  aae_ae_forward / aae_ae_backward: layer-chain models only
 * (n_hidden <= 207, n_code + cond_inc <= 415); the slice handle may be any size. */
int aae_ae_forward(aae_handle h, const aae_batch* batch, const float* cond_dev, const aae_rng_inject* inject, void* stream);
int aae_output_layer_step(aae_handle h, const aae_batch* batch, void* stream);
int aae_ae_backward(aae_handle h, const float* dA2_dev, int64_t dA2_ld, void* stream);
/* ... and the FIRST encoder layer sharded the same way (r2): enc.lin1 is the other [n_items][n_hidden] matrix
 * (reference aae.py:116, 132-135), and under replication its row-sparse gradient is the step's largest exchange (the
 * packed rows of aae_w1_export: ~60 MB gathered per step at 8 x 100 documents).  The slice handle already holds the
 * columns of its items; with these entries it also computes and trains them, and the ranks exchange
 * [global rows][n_hidden] blocks instead (0.6 MB):
 *   aae_set_doc_l1                 slice handle: L1 norms of the COMPLETE documents, indexed by document id (the batch it
 *                                  sees holds only its own columns; F.normalize(x, 1) divides by the whole row,
 *                                  aae.py:133)
 *   aae_set_first_layer_external   replica handle: AAE_T_ACT_A1 is filled by the caller before aae_ae_forward /
 *                                  aae_disc_step, dL/d(a1) of the ae / gen phase is left in AAE_T_ACT_GA1, enc.lin1 is
 *                                  neither read nor updated here; the bias (AAE_T_ENC_B1) stays a small replicated
 *                                  parameter: its gradient is exported with the other small layers'
 *   aae_first_layer_forward        slice handle: its items' share of x * enc.lin1^T for the (global) batch -> its
 *                                  AAE_T_ACT_A1; bias_dev (the replica's AAE_T_ENC_B1) on exactly one share, else
 *                                  NULL.  batch != NULL opens the handle's step
 *                                  (aae_output_layer_step(batch = NULL) continues it), NULL = the running batch again
 *                                  (Enc_eval of disc_step, aae.py:722, after enc_optim's update)
 *   [reduce-scatter the slices' ACT_A1 over the ranks -> the replica's ACT_A1 rows]
 *   [all-gather the replicas' ACT_GA1 -> the slice handle's ACT_GA1]
 *   aae_first_layer_update         slice handle: x^T * dL/d(a1) on its rows of enc.lin1 and optimiser `which` (0
 *                                  enc_optim after the ae phase, 2 gen_optim after gen_step) on them; ga1_dev = NULL:
 *                                  its ACT_GA1, else [rows][ld] - or, rows_per_block > 0, the gathered packets
 *                                  themselves: block r = rows r * rows_per_block .. at ga1_dev + r * block_stride floats
 * aae_prefetch_batch on the slice handle names the NEXT global batch: its distinct-item list and deferred-Adam catch-up
 * then run beside the running step, as for aae_step. */
int aae_set_doc_l1(aae_handle h, const float* doc_l1_dev);
int aae_set_first_layer_external(aae_handle h, int on);
int aae_first_layer_forward(aae_handle h, const aae_batch* batch, const float* bias_dev, void* stream);
int aae_first_layer_update(aae_handle h, const float* ga1_dev, int64_t ld, int32_t rows_per_block, int64_t block_stride,
                           int which, void* stream);
/* The replica's small layers after such an all-gather, in ONE launch: optimiser which_a (0 enc_optim / 2 gen_optim:
 * enc.lin1's bias, enc.lin2, enc.lin3) and, which_b = 1, dec_optim's dec.lin1, dec.lin2 (which_b = -1: none), their
 * gradients read as the sum over the n_peers packets in peer order (bitwise the same on every rank): packet q holds at
 * packets_dev + q * peer_stride + span_offset (floats, multiples of 4) a copy of the arena span that starts at
 * AAE_T_GRAD + AAE_T_ENC_B1.  Replaces a reduction over the peers + aae_apply_updates per optimiser. */
int aae_apply_gathered(aae_handle h, int which_a, int which_b, const float* packets_dev, int64_t peer_stride,
                       int32_t n_peers, int64_t span_offset, void* stream);

/* CategoricalCondition (condition.py:397-508): a trainable embedding of a categorical attribute, reduced over the
 * document's (batch-padded) value list and concatenated to the code.  The table and its optimiser state belong to
 * the caller (plain device arrays [vocab][dim], row-major, dim <= 256); index 0 is the padding / out-of-vocabulary
 * token: it reads as zero and never receives a gradient (nn.Embedding(padding_idx=0)).
 *   idx_dev [rows][width] int32; reduce: sum (also "no reduction" with width = 1) or mean over the padded width
 *   (hid.mean(1), condition.py:485-487, padding included).
 * aae_cat_encode   out_dev[r][0:dim] = reduce_w table[idx[r][w]]   - write it into the cond block of aae_step
 * aae_cat_update   the embedding's backward from dout_dev [rows][>= dim] (AAE_T_ACT_DZC columns of this condition)
 *                  followed by the condition's own optimiser step (condition.py:491-497), default betas / eps:
 *                  AAE_CAT_SPARSE_ADAM  torch.optim.SparseAdam (nn.Embedding(sparse=True), the reference's default):
 *                                       only rows named by the batch move;
 *                  AAE_CAT_ADAM         torch.optim.Adam over the whole table; grad_scratch_dev [vocab][dim] must be
 *                                       zero on the first call and is left zero.
 *                  `step` is the 1-based count of this update (the optimiser's state['step'] after it). */
enum { AAE_CAT_SUM = 0, AAE_CAT_MEAN = 1 };
enum { AAE_CAT_SPARSE_ADAM = 0, AAE_CAT_ADAM = 1 };
int aae_cat_encode(const float* table_dev, int32_t vocab, int32_t dim, const int32_t* idx_dev, int32_t rows,
                   int32_t width, int32_t reduce, float* out_dev, int64_t out_ld, void* stream);
int aae_cat_update(float* table_dev, float* exp_avg_dev, float* exp_avg_sq_dev, float* grad_scratch_dev, int32_t vocab,
                   int32_t dim, const int32_t* idx_dev, int32_t rows, int32_t width, int32_t reduce,
                   const float* dout_dev, int64_t dout_ld, int32_t optimizer, double lr, int64_t step, void* stream);
/* PretrainedWordEmbeddingCondition's preprocessing (condition.py:345-369 -> ub.py:52-57, EmbeddedVectorizer.transform:
 * `sparse_scores @ self.embedding`): out_dev[r][0:dim] = sum over the CSR entries e of row r of
 * values[e] * table_dev[indices[e]][0:dim], accumulated in CSR order.  indptr int64 [n_rows + 1] (absolute offsets
 * into indices / values), indices int32, values float32 (the TF-IDF weights); indices outside the table are skipped. */
int aae_csr_embed(const int64_t* indptr_dev, const int32_t* indices_dev, const float* values_dev, int32_t n_rows,
                  const float* table_dev, int32_t n_table_rows, int32_t dim, int64_t table_ld, float* out_dev,
                  int64_t out_ld, void* stream);
/* The reference's call form: partial_fit(X) / predict(X) take the DENSE [rows][n_cols] batch that
 * X_shuf[start:end].toarray() produced (aae.py:745-754, 823, 848-853) and upload it whole.  aae_dense_to_csr compacts
 * such a matrix, already on the device as float32 (elem_bytes = 4) or float64 (8), into the CSR form aae_batch takes:
 * indptr_dev int64 [rows + 1], indices_dev int32 / values_dev float32 [capacity], columns ascending within a row.
 * scratch_dev: int32 [rows + 8].  stats_out_host (int32[4], after a stream synchronisation): longest row, total
 * entries, != 0 if a value lies outside [0, 1] (the reference's F.binary_cross_entropy raises on such targets),
 * != 0 if the matrix has more than `capacity` entries (nothing was written). */
int aae_dense_to_csr(const void* dense_dev, int32_t elem_bytes, int64_t ld, int32_t rows, int32_t n_cols,
                     int64_t* indptr_dev, int32_t* indices_dev, float* values_dev, int64_t capacity,
                     int32_t* scratch_dev, int32_t* stats_out_host, void* stream);
/* the two halves of aae_disc_gen (data parallel needs the discriminator update applied
 * between them) */
int aae_disc_step(aae_handle h, const aae_rng_inject* inject, void* stream);
int aae_gen_step(aae_handle h, const aae_rng_inject* inject, void* stream);

/* recon / disc / gen loss of the last step (the three .item() calls, aae.py:711,732,743).
 * Synchronises `stream`. */
int aae_read_losses(aae_handle h, float out_host[3], void* stream);

/* AdversarialAutoEncoder.predict, aae.py:840-870: eval-mode encoder -> (constant concat
 * condition) -> decoder -> sigmoid, out_dev [rows][out_ld] float32. */
int aae_predict(aae_handle h, const aae_batch* batch, const float* cond_dev, float* out_dev,
                int64_t out_ld, void* stream);
/* predict followed on the device by what Evaluation does on the host with the dense matrix:
 * remove_non_missing (row-wise min-max scaling; items present in the input row excluded when
 * exclude_known != 0; evaluation.py:183-199) and argtopk (evaluation.py:20-58).  Writes the k
 * (<= 32) best item ids per row, best first, and their scaled scores: [rows][k].
 * Items of EQUAL fp32 score (saturated sigmoids of a trained model, collisions near 1) are in np.argpartition's arbitrary
 * order in the reference; here the fused form (aae_rank_max_rows) orders them by logit, the dense form by the smaller item
 * id - the k scores are identical, the named items may differ exactly at ties. */
int aae_predict_topk(aae_handle h, const aae_batch* batch, const float* cond_dev, int32_t k,
                     int32_t exclude_known, int32_t* idx_out_dev, float* val_out_dev, void* stream);
/* Rows ONE aae_predict_topk / aae_decode_topk call may rank (>= max_batch).  Where the fused form applies (r4: the output
 * layer, its sigmoid, the row minimum / maximum, the known-item mask and the top-k selection in one pass over dec.lin3 -
 * the [rows, n_items] score matrix never exists in HBM; hidden widths of the layer-chain kernels) a call takes far more
 * rows than a training batch - the parameter stream is then read once per call, not once per max_batch rows - and such a
 * batch is exempt from the max_batch / max_nnz bounds of aae_batch.  Otherwise *rows_out = max_batch. */
int aae_rank_max_rows(aae_handle h, int32_t k, int32_t* rows_out);
/* The same behind a decoder input the caller built (AdversarialAutoEncoder.predict's second half, aae.py:855-866:
 * `z = conditions.encode_impose(z, c_batch)` with plugins of any kind, then dec): zc_dev [batch->n_rows][zc_ld],
 * zc_ld >= n_code + cond_inc; `batch` names the input rows (their items are the ones exclude_known removes). */
int aae_decode_topk(aae_handle h, const float* zc_dev, int64_t zc_ld, const aae_batch* batch, int32_t k,
                    int32_t exclude_known, int32_t* idx_out_dev, float* val_out_dev, void* stream);
/* split form for generic conditions */
int aae_encode(aae_handle h, const aae_batch* batch, float* z_out_dev, void* stream);
int aae_decode(aae_handle h, const float* zc_dev, int64_t zc_ld, int32_t n_rows,
               float* out_dev, int64_t out_ld, void* stream);

/* data parallel (AAE_GRAD_EXPORT): after the caller has all-reduced the AAE_T_GRAD tensors of
 * optimiser `which` (0 enc 1 dec 2 gen 3 disc), apply torch.optim.Adam/SGD.step to them. */
int aae_apply_updates(aae_handle h, int which, void* stream);
/* Row-sparse exchange of the first encoder layer's gradient (only the rows of the items in the
 * batch are non-zero).  aae_w1_export packs this rank's rows into hdr_dev (int32[1 + cap]: count,
 * then item ids) and vals_dev (float[cap][n_hidden]) and clears them; after an all-gather the
 * caller hands the n_peers packets (peer p at byte offset p * peer_stride_bytes from both base
 * pointers) to aae_w1_import, which sums them in peer order, brings the union of rows up to date
 * and runs the optimiser `which` (0 enc_optim after the ae phases, 2 gen_optim after gen_step) on
 * them.  cfg.dp_world must hold the number of peers. */
int aae_w1_export(aae_handle h, int32_t* hdr_dev, float* vals_dev, int32_t cap, void* stream);
/* packet layout for `cap` rows: hdr_words int32 words of header, then cap * n_hidden floats of
 * rows, then the encoder's small-layer gradients (b1, W2, W3), which aae_w1_import also sums and
 * applies - so one all-gather per exchange point carries everything the encoder optimisers need */
int aae_w1_packet_floats(aae_handle h, int32_t cap, int64_t* hdr_words, int64_t* total_floats);
int aae_w1_import(aae_handle h, const int32_t* hdr_dev, const float* vals_dev, int32_t cap, int32_t n_peers,
                  int64_t peer_stride_bytes, int which, void* stream);
/* Sharded optimiser: aae_apply_updates_except is aae_apply_updates without tensor
 * `skip_tensor_id`; aae_apply_shard runs optimiser `which` on rows [row_begin, row_end) of that
 * tensor with the caller's gradient shard (e.g. a reduce-scatter result).  Used for DEC_V3:
 * reduce-scatter -> update 1/world of the rows -> all-gather the updated rows. */
int aae_apply_updates_except(aae_handle h, int which, int skip_tensor_id, void* stream);
int aae_apply_shard(aae_handle h, int tensor_id, int64_t row_begin, int64_t row_end,
                    const float* grad_shard_dev, int which, void* stream);
/* ---- the data-parallel step as ONE call (r3; SURVEY 8b "aae_comm_init + entry points that enqueue kernels and RCCL calls") ----
 * Scheme: both vocabulary-wide matrices sharded over the ranks (aae_first_layer_* / aae_output_layer_step above; DESIGN.md
 * 5.0): per partial_fit the ranks exchange seven small blocks - reduce-scatter of the first layer's shares, all-gather of
 * the decoder's last hidden activations, reduce-scatter of dL/d(hidden), all-gather of [dL/d(a1) | small-layer gradients]
 * (twice: enc_optim, gen_optim), reduce-scatter for Enc_eval, all-reduce of the discriminator's gradients.  aae_dp_step
 * enqueues ALL of it - kernels and collectives - on `stream`; no host work in between (driven phase by phase from Python
 * the loop needed 0.35 ms of host time per step at 8 ranks against 0.33 ms of GPU work).
 * The collectives come as a table of function pointers over device float buffers: count = floats PER RANK (all_gather:
 * what each rank sends; reduce_scatter: what each rank receives, sums over the ranks; all_reduce: the whole buffer, in
 * place); each enqueues on `stream` and returns 0 or a negative AAE_E* code.  aae_rccl_init fills the table with RCCL
 * (xGMI) collectives on a communicator the library creates from a 128-byte ncclUniqueId (rank 0: aae_rccl_unique_id, then
 * hand it to the other ranks by any means - torch.distributed.broadcast, MPI, a file); librccl.so is opened at run time.
 * Any other table works as well (the tests use host-staged gloo collectives and single-process stand-ins).
 *   replica            this rank's replica handle: grad_mode = AAE_GRAD_EXPORT, aae_set_first_layer_external(1)
 *   slice              this rank's item-slice handle (fused optimiser, aae_set_doc_l1, aae_set_grad_scale(slice / all items))
 *   local              this rank's documents (its share of the global batch) in the replica's corpus
 *   global_slice       the GLOBAL batch, rank-major, in the slice's corpus (its items' columns); n_rows = world x local's
 *   next_global_slice  the global batch of the NEXT step (aae_prefetch_batch on the slice handle) or NULL
 *   cond_dev / inject  as aae_step */
typedef struct aae_collectives {
    void* ctx;
    int (*all_gather)(void* ctx, const float* send_dev, float* recv_dev, int64_t count, void* stream);
    int (*reduce_scatter)(void* ctx, const float* send_dev, float* recv_dev, int64_t count, void* stream);
    int (*all_reduce)(void* ctx, float* buf_dev, int64_t count, void* stream);
    int32_t world, rank;
} aae_collectives;
int aae_rccl_unique_id(char id_out[128]);
int aae_rccl_init(const char id[128], int32_t world, int32_t rank, aae_collectives* out);
int aae_rccl_destroy(aae_collectives* c);
/* set-up: the replica's scratch for the gathered packets of `world` ranks x n_rows local documents (else aae_dp_step
 * allocates it in its first call) */
int aae_dp_reserve(aae_handle replica, int32_t n_rows, int32_t world);
int aae_dp_step(aae_handle replica, aae_handle slice, const aae_collectives* coll, const aae_batch* local,
                const aae_batch* global_slice, const aae_batch* next_global_slice, const float* cond_dev,
                const aae_rng_inject* inject, void* stream);
/* The third data-parallel scheme (r4): ONE handle per rank = its item slice of the two vocabulary-wide layers
 * + a full copy of the hidden layers, the WHOLE global batch through the hidden stacks on every rank (identical inputs ->
 * identical small-layer gradients and updates: no gradient exchange), and three all-reduces of [global rows, n_hidden]
 * partial sums per partial_fit (the first layer's pre-activations, dL/d(dh2), the first layer again for Enc_eval) - against 7
 * collectives, gradient packets and a second handle in aae_dp_step.  north_star's "RCCL all-reduce ... over xGMI": these are
 * the only all-reduces the step needs.  Reference: aae.py:745-766 over the global batch.
 *   handle      fused optimiser, aae_set_first_layer_external(1), aae_set_doc_l1, max_batch = the global batch
 *   batch       the GLOBAL batch in the handle's corpus (its items' columns); next_batch: named ahead (aae_prefetch_batch) or NULL
 *   item_share  items of this handle / items of the model (the BCE is a mean over all items)
 *   cond_dev    the condition block of ALL rows; inject as aae_step (masks / z_real of all rows) */
int aae_shard_step(aae_handle h, const aae_collectives* coll, const aae_batch* batch, const aae_batch* next_batch,
                   const float* cond_dev, const aae_rng_inject* inject, float item_share, void* stream);
/* r6: the same table as a ONE-SHOT all-reduce over peer-mapped mailboxes, for the ranks of one node and the small exchanges of
 * aae_shard_step (csrc/ipc_collectives.h): every rank copies its operand into its own mailbox, publishes the call, waits for
 * its peers' flags and adds the world mailboxes in rank order - one launch, one hop over xGMI, the same bits on every rank.
 * all_reduce only (all_gather / reduce_scatter fail: aae_dp_step's exchanges are RCCL's).  No counterpart in the reference.
 *   aae_ipc_create   allocates this rank's mailbox for operands of up to max_floats floats; handle_out: 64 bytes
 *                    (hipIpcMemHandle_t) for the caller to carry to every peer (torch.distributed all_gather)
 *   aae_ipc_init     handles = world x 64 bytes in rank order (this rank's own entry is not read); maps the peers' mailboxes
 *   aae_ipc_destroy  after a barrier of the caller's (no peer may still read this rank's mailbox); reports a wait that timed out */
int aae_ipc_create(int64_t max_floats, char handle_out[64], void** mailbox_out);
int aae_ipc_init(void* mailbox, const char* handles, int32_t world, int32_t rank, int64_t max_floats, aae_collectives* out);
int aae_ipc_destroy(aae_collectives* c, void* mailbox);
/* a single-process stand-in table (measurement aid: one rank's compute with every exchange replaced by device copies of the
 * same shapes - all_gather = the operand repeated `world` times, reduce_scatter = its first chunk, all_reduce = identity;
 * rank 0 of `world`) */
int aae_echo_collectives(int32_t world, aae_collectives* out);
/* blocking copy between any two buffers (host or device) behind `stream`: what a host-side collectives table needs to
 * stage operands without a HIP binding of its own */
int aae_memcpy_sync(void* dst, const void* src, size_t bytes, void* stream);

/* scale applied to this rank's loss gradients (local_rows / global_rows) so that the
 * all-reduced sum equals the single-process mean over the global batch. */
int aae_set_grad_scale(aae_handle h, float scale);

/* Live per-kernel timing for bench.py's roofline line: when enabled, a hipEvent pair is
 * recorded on the launch stream around each launch of the kernels below; aae_profile_read
 * waits for them, returns the summed duration and resets the counter. */
enum { AAE_K_ENC_GATHER = 0,   /* sparse row gather of the first encoder layer */
       AAE_K_DEC_BCE_FWD,      /* decoder output GEMM + sigmoid/BCE epilogue */
       AAE_K_DEC_DA2,          /* dL/dlogits * V3 (split-K) */
       AAE_K_DEC_DV3_ADAM,     /* dV3 GEMM + fused Adam on V3 */
       AAE_K_ENC_W1_ADAM,      /* Adam over the touched rows of the encoder's first layer */
       AAE_K_DEC_FUSED,        /* fused decoder output layer: logits + BCE + dV3/Adam + dA2 (B <= ~104) */
       AAE_K_CHAIN,            /* a layer-chain program: the hidden stacks of one phase (5 launches per step) */
       AAE_K_DEC_CRIT,         /* split form of the fused output layer, critical launch: logits + BCE + dA2 (+ dL/dlogits tiles) */
       AAE_K_DEC_OPT,          /* ... deferred launch on the library's side stream: dV3 + dec_optim behind the rest of the step */
       AAE_K_RANK,             /* fused predict -> rank: output layer + sigmoid + min/max + known-item mask + per-workgroup top-k */
       AAE_K_COLLECTIVE,       /* a collective of aae_dp_step / aae_shard_step: event pair on the step's stream around the table's call
                                * (what the rank waits: the transfer AND the slowest rank's arrival) */
       AAE_K_N };
/* on = 0: off; 1: every kernel id above; otherwise a selection: bit (k + 1) of `on` times kernel id k
 * (an event pair costs a few microseconds of stream time, so a timed run selects only what it reports) */
int aae_profile_enable(aae_handle h, int on);
int aae_profile_read(aae_handle h, int kernel_id, double* total_ms, int64_t* launches);

/* The fused decoder output layer (B <= ~104, fused optimiser) runs as TWO launches: what the rest of the step waits
 * for (logits, BCE, dL/d(hidden)) on the caller's stream, and the weight gradient + dec_optim pass over DEC_V3 - work
 * only the NEXT step's forward needs (reference aae.py:709 `self.dec_optim.step()` followed by aae.py:713-743, which
 * never read dec) - on a stream the handle owns, concurrently with the rest of the step.  Every entry point of this
 * library that touches DEC_V3, its moments or the step's scratch first makes its `stream` wait for that launch, so
 * callers that go through the ABI see the reference's ordering.  A caller that reads DEC_V3 / ADAM_DEC + 4,5 through
 * its own arena views (aae_tensor_info) calls aae_join first: it makes `stream` wait for the deferred launch (a
 * no-op when none is pending).  aae_sync includes it.  aae_set_split(h, 0) turns the split form off (one launch,
 * everything on the caller's stream: needed to capture a step into a hipGraph without joining), n > 0 sets the number
 * of workgroups of the deferred launch (default: by shape - enough for the launch to take ~80 us, at most 9/16 of the
 * CUs in fp32, 5/8 with bf16 inputs). */
/* DenoisingAutoEncoder(corrupt='gauss'), reference dae.py:40-45 and 191: `self.enc(self.corrupt(batch, noise_factor))`
 * with gauss_noise = batch + randn(batch.size()) * noise_factor - the encoder of the NEXT step-opening call reads the
 * DENSE batch plus noise_dev [rows][noise_ld >= n_items] (already scaled) on all n_items columns, L1-normalised over all
 * of them (aae.py:132-133); its first layer runs as a dense product, its weight gradient as a dense product with the
 * optimiser on every row of ENC_W1T.  The BCE target stays the clean batch.  Needs cfg.dense_noise = 1 (room for the
 * dense input; plain autoencoder, fp32, fused optimiser); noise_dev must stay valid until the step has run. */
int aae_set_input_noise(aae_handle h, const float* noise_dev, int64_t noise_ld);
int aae_join(aae_handle h, void* stream);
/* ... the deferred optimiser launch alone: enough before reading or writing the AAE_T_ACT_* tensors (a prefetch started
 * with aae_prefetch_batch touches enc.lin1 and its bookkeeping only and keeps running). */
int aae_join_output_layer(aae_handle h, void* stream);
/* The epoch loop (aae.py:808-831) knows the batch AFTER the one it is about to run.  Named here before the step that
 * precedes it (aae_step / aae_ae_encode / aae_ae_forward), that batch's share of the step-opening work - the list of
 * its distinct items and the replay of the deferred zero-gradient Adam steps on their enc.lin1 rows (what
 * torch.optim.Adam did eagerly in enc_optim.step() / gen_optim.step(), aae.py:706,742) - runs on the handle's side
 * stream while the step before it executes, instead of opening the next step.  A hint, not a promise: a next step on
 * any other batch (other pointers / row window) ignores it and does the work itself.
 * The step recognises the named batch by its pointers, its row window and aae_batch.generation (ABI 3; r1-r4 went by the
 * pointers alone, and a buffer refilled in place or re-allocated at the same address was silently taken for the batch
 * whose item list had been built ahead - rows missing from that list got no update).  The caller bumps `generation`
 * whenever it rewrites or re-allocates any array the batch names; a batch with generation 0 is never matched, and naming
 * one here is accepted and ignored.  What stays the caller's duty is what any asynchronous reader asks: the named arrays
 * must stay allocated and unchanged until the work built from them has run (the step after the next, or aae_join).
 * grad_mode = fused only (otherwise accepted and ignored). */
int aae_prefetch_batch(aae_handle h, const aae_batch* next);
int aae_set_split(aae_handle h, int32_t workgroups);

#ifdef __cplusplus
}
#endif
#endif /* AAEREC_HIP_H */
