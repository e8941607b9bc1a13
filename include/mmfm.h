/* mmfm.h — C-ABI of libmmfm_hip.so: hand-written HIP (gfx950 / CDNA4) kernels for the
 * masked-pretraining hot path of yzhang511/multi_modal_foundation_model.
 *
 * The reference has no FFI: the path sits behind Python classes.  Each entry
 * point below names the reference site (file:line under /root/reference/src) whose device work
 * it replaces.  INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every entry returns 0 on success, a hipError_t (>0) or -1 (argument check) otherwise;
 *    the message is available through mmfm_last_error() (thread-local);
 *  - all buffers are caller-allocated DEVICE pointers; the library never allocates, frees,
 *    retains or synchronises (every launch function is hipGraph-capturable);
 *  - `stream` is a hipStream_t passed as void*;
 *  - dtype: 0 = f32 storage (parity mode), 1 = bf16 storage with fp32 accumulate/statistics;
 *  - row-major everywhere; "ld" = leading dimension in elements.
 */
#ifndef MMFM_H
#define MMFM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MMFM_VERSION 401
#define MMFM_F32 0
#define MMFM_BF16 1

typedef void* mmfm_stream;

/* Counter-based dropout: keep(idx) = f(state[0], state[1], site, idx); state = 2 x uint32 in
 * DEVICE memory (so graph replays see new masks after mmfm_rng_advance).  p <= 0 disables.
 * Replaces nn.Dropout / SDPA dropout_p (mm_utils.py:52,111,114; encoder_embeddings.py:61). */
typedef struct {
    const void* state;
    uint32_t site;
    float p;
} mmfm_dropout;

int mmfm_version(void);
const char* mmfm_last_error(void);
/* 0 iff `device` is a gfx950 part. */
int mmfm_device_check(int device);
/* state[0]=lo32(mix(seed)), state[1]=step counter start */
int mmfm_rng_seed(void* state, uint64_t seed, mmfm_stream stream);
int mmfm_rng_advance(void* state, mmfm_stream stream);

/* ---------------------------------------------------------------------------------- GEMM
 * C[m][n] = epi( sum_k A(m,k) * B(k,n) ).  Replaces every nn.Linear forward/backward on the
 * path (mm_utils.py:46-52,88-95,107-114; encoder_embeddings.py:28-30,50-54;
 * decoder_embeddings.py:83,107; mm.py:74,292) and their autograd.
 *   a_kcontig: 1 -> A(m,k) = A[m*lda + k];  0 -> A(m,k) = A[k*lda + m]
 *   b_kcontig: 1 -> B(k,n) = B[n*ldb + k] (an nn.Linear weight [N,K]);  0 -> B(k,n) = B[k*ldb + n]
 *   splits > 1: split-K; split z covers k in [z*kchunk, (z+1)*kchunk) and writes its raw fp32
 *               partial tile to C + z*slab_stride (no epilogue); reduce with mmfm_reduce_slabs.
 * epilogue (splits == 1), in this order:
 *   v = acc + bias[n]; pre_out[m*ldc+n] = v;
 *   act 1: v = gelu_erf(v)           act 2: v = softsign(v) * act_scale          (forward)
 *   act 3: v *= gelu_erf'(gradmul_pre[m*ldc+n])   act 4: v *= softsign'(gradmul_pre[..]) * act_scale
 *          (backward through the activation whose pre-activation the forward stored via pre_out);
 *   act 5: v *= (1 - |y| / act_scale)^2 * act_scale with y = gradmul_pre[..] the forward's act-2 OUTPUT: the same softsign'
 *          without a saved pre-activation (bf16 throughput mode; the fp32 parity path keeps act 4);
 *   v = dropout(v) (counter m*N+n);  v += residual[m*ldr+n];  C[m*ldc+n] = v
 */
typedef struct {
    int dtype;       /* storage of A, B, pre_out, gradmul_pre, residual and (unless c_f32) C */
    int c_f32;       /* 1: C is fp32 regardless of dtype */
    const void* A;
    const void* B;
    void* C;
    int M, N, K;
    int lda, ldb, ldc;
    int a_kcontig, b_kcontig;
    int splits, kchunk;
    int64_t slab_stride;
    const float* bias;
    void* pre_out;
    int act;
    float act_scale;
    const void* gradmul_pre;
    mmfm_dropout drop;
    const void* residual;
    int ldr;
    float* colsum;   /* optional, dtype bf16 with a_kcontig == 0 (the dW = dY^T X launches): colsum[z*slab_stride + m] =
                        sum over split z's k range of A(m,k), i.e. the bias gradient of the same nn.Linear, computed from
                        the A tiles the GEMM stages anyway (replaces a separate mmfm_colsum pass over dY).  With
                        colsum == C + M*ldc the partials sit behind each dW slab and ONE mmfm_reduce_slabs over
                        M*N + M elements finishes weight and bias gradient together. */
} mmfm_gemm_desc;
int mmfm_gemm(const mmfm_gemm_desc* d, mmfm_stream stream);
/* Two independent products, same results as two mmfm_gemm calls.  Two bf16 weight-gradient descriptors (the dY^T X form with split-K
 * slabs) run as ONE launch, each on its share of the CUs: the caller then sizes their split counts so that
 * tiles(a) * splits(a) + tiles(b) * splits(b) = 256 with the first term a multiple of 8 (mmfm_gemm_dw_tiles) - each product makes half
 * the slabs it would make alone.  The weight gradients of two linears whose dY are both at hand (MLP up / down, attention qkv / out_proj). */
int mmfm_gemm_pair(const mmfm_gemm_desc* a, const mmfm_gemm_desc* b, mmfm_stream stream);

/* dst[i] (+)= sum_s src[s*slab_stride + i], fp32, deterministic order.  `src` is scratch: it may be
 * clobbered (a tall-skinny reduction first sums groups of slabs in place). */
/* Work items per K-split that a bf16 weight-gradient launch (a_kcontig == b_kcontig == 0, fp32 output, 16-B aligned rows) over K rows is
 * cut into: the caller sizes `splits` so that items x splits fills the 256 CUs once (each item writes one M-tile x N-tile fp32 slab). */
int mmfm_gemm_dw_tiles(int M, int N, int K);
int mmfm_reduce_slabs(float* dst, const float* src, int64_t n, int nslabs, int64_t slab_stride,
                      int accumulate, mmfm_stream stream);
/* Several slab reductions in ONE launch (small batches: a backward segment's weight-gradient GEMMs each leave a few small
 * slabs, and a launch per reduction costs more than the reduction).  `table` is a DEVICE array of `count` entries; every entry
 * keeps its own slab region until the call.  chunk0 = number of 256-float chunks of the entries before it (exclusive prefix sum of
 * ceil(n / 256)); `total_chunks` = that sum over all entries.  Deterministic slab order, like mmfm_reduce_slabs. */
typedef struct {
    float* dst;
    const float* src;
    int64_t n, slab_stride;
    int32_t nslabs, accumulate, chunk0, pad_;
} mmfm_reduce_entry;
int mmfm_reduce_slabs_multi(const mmfm_reduce_entry* table, int count, int total_chunks, mmfm_stream stream);
/* out[n] (+)= sum_r x[r*ld + n]   (bias gradients).  workspace >= mmfm_colsum_workspace bytes. */
int64_t mmfm_colsum_workspace(int64_t R, int N);
int mmfm_colsum(int dtype, const void* x, int64_t R, int N, int ld, float* out, int accumulate,
                void* workspace, int64_t workspace_bytes, mmfm_stream stream);

/* ---------------------------------------------------------------------------------- LayerNorm
 * nn.LayerNorm(H), eps 1e-5, affine: encoder_embeddings.py:98,100; decoder_embeddings.py:118-126;
 * mm.py:72,77.  One wavefront per row, fp32 statistics.
 * destitch_T > 0: output row for input row r=(b,l) is (l/T)*(B*T) + b*T + l%T, i.e. the final
 * decoder_norm writes per-modality contiguous [M][B*T][H] blocks (the boolean gather of
 * decoder_embeddings.py:105 becomes a plain slice); the backward reads dy the same way. */
int mmfm_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y,
                       float* mean, float* rstd, int64_t R, int H, float eps,
                       int destitch_L, int destitch_T, mmfm_stream stream);
int64_t mmfm_layernorm_bwd_workspace(int64_t R, int H);
/* dx = dres + LN'(dy);  dgamma/dbeta (+)= column sums.  dres may be NULL; dx may alias dres. */
int mmfm_layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* rstd,
                       const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta,
                       int accumulate, int64_t R, int H, int destitch_L, int destitch_T,
                       void* workspace, int64_t workspace_bytes, mmfm_stream stream);

/* ---------------------------------------------------------------------------------- attention
 * F.scaled_dot_product_attention with the reference's masks (mm_utils.py:105-111,143-149;
 * mm.py:152-158,178-194) without materialising [B,h,L,L]:
 *   allowed(b,q,k) = (DIAG && q==k) | (CAUSAL ? k<=q : keypad[b][k]) | (SEP && mod_id[q]!=mod_id[k])
 * q/k/v/o are [B, L, heads*dh] views with row strides ldq/ldk/ldv/ldo (so a fused QKV buffer
 * works); lse is [B, heads, Lq] fp32.  drop_p acts on the probabilities, drop_o on the output
 * (the nn.Dropout in front of out_proj, mm_utils.py:114). */
#define MMFM_ATTN_DIAG 1
#define MMFM_ATTN_CAUSAL 2
#define MMFM_ATTN_SEP 4
typedef struct {
    int dtype;
    int B, heads, Lq, Lk, dh;
    const void* q; const void* k; const void* v;
    int ldq, ldk, ldv;
    void* o; int ldo;            /* fwd: output (after drop_o).  bwd: the forward's output */
    float* lse;
    const uint8_t* keypad;       /* [B][Lk] */
    const uint8_t* mod_id;       /* [max(Lq,Lk)] or NULL */
    int flags;
    float scale;
    mmfm_dropout drop_p, drop_o;
    /* backward only */
    const void* d_o; int lddo;   /* grad wrt the out_proj input (i.e. AFTER drop_o) */
    void* dq; void* dk; void* dv;
    int lddq, lddk, lddv;
    /* optional (round 4): workspace of mmfm_attn_keepbits_bytes(B, heads, Lq, Lk) bytes for the keep decisions of drop_p, one bit
     * per (b, head, query, key).  With it, launches the dh = 32 fast kernels take (bf16, Lq <= 256, Lk <= 224, both % 8 == 0, no
     * CAUSAL / SEP, 16-B aligned operands) draw the decisions ONCE - mmfm_attn_fwd runs a generator kernel in front of the
     * forward - and mmfm_attn_bwd reads the same bits: the caller leaves the buffer alone between the two calls.  The drop
     * probability is then honoured to 2^-10: keep = mmfm_attn_keep_prob(drop_p.p), survivors are scaled by 1 / keep.
     * NULL (or a shape the fast kernels do not take): both directions re-derive the decisions from the counter hash. */
    void* keepbits;
} mmfm_attn_desc;
int mmfm_attn_fwd(const mmfm_attn_desc* d, mmfm_stream stream);
int mmfm_attn_bwd(const mmfm_attn_desc* d, mmfm_stream stream);
int64_t mmfm_attn_keepbits_bytes(int B, int heads, int Lq, int Lk);
/* the keep probability the keep-bit path applies for a drop probability p */
float mmfm_attn_keep_prob(float p);

/* ---------------------------------------------------------------------------------- masks / stitch
 * mm.py:245-275 (mask = eval_mask[:,:,0] & attn_mask), :102 (mod_mask), :145,167 (sample-0 ids),
 * :229-233 (n_examples).  For modality m: mask_src[m] is int64 with element (b,t) at
 * mask_src[m][(b*T+t)*mask_stride[m]]; attn is int64 [B][T].
 * Outputs: tokmask u8 [B][M*T], keypad u8 [B][M*T], keep0 u8 [M*T] (0 where sample 0 is masked),
 * mod_id u8 [M*T], count int64 [M] = channels[m] * sum(tokmask of modality m).  Bit-exact. */
int mmfm_mask_prep(int B, int T, int M, const int64_t* const* mask_src, const int64_t* mask_stride,
                   const int64_t* attn, const int64_t* channels, uint8_t* tokmask, uint8_t* keypad,
                   uint8_t* keep0, uint8_t* mod_id, int64_t* count, mmfm_stream stream);

/* encoder_embeddings.py:56-61 + mm.py:90-110,143-149,289 (and the decoder twins):
 *   emb[b, m*T+t] = mod_emb[mod_row] + pos_emb[ts[b][t]]
 *   x  [b, m*T+t] = keep0[m*T+t] * tok[b*T+t] + emb[...]
 * called once per modality m; tok is [B*T][H]; x/emb are [B][L][H]; emb may be NULL; pos_emb has
 * max_F rows (time stamps are clamped into [0, max_F) for memory safety). */
int mmfm_stitch_fwd(int dtype, const void* tok, const float* mod_emb_row, const float* pos_emb,
                    const int64_t* ts, const uint8_t* keep0, void* x, void* emb,
                    int B, int T, int L, int m, int H, int max_F, mmfm_stream stream);
/* autograd of the above for one modality: d_tok[b*T+t] = keep0 * dropout'(dx[b, m*T+t]);
 * d_mod_row (+)= sum_{b,t} (dx + dextra);  d_pos[ts[b][t]] (+)= dx + dextra  (dextra may be NULL:
 * it is d_context -> encoder_emb, mm.py:292).  acc_mod / acc_pos select += for each output (the
 * modality embedding is shared by the encoder and decoder tokenisers, mm.py:84-87, the position
 * tables are not).  Deterministic (no atomics): fp32 mode scatters through per-column LDS tables and reduces the
 * per-chunk partials; bf16 mode (H % 8 == 0) multiplies by a one-hot matrix on the MFMA GEMM ([d_pos; d_mod] = OH^T E,
 * split-K slabs, fixed-order reduction). */
int64_t mmfm_stitch_bwd_workspace(int dtype, int B, int T, int L, int H, int max_F);
int mmfm_stitch_bwd(int dtype, const void* dx, const void* dextra, const int64_t* ts, const uint8_t* keep0,
                    mmfm_dropout drop, void* d_tok, float* d_mod_row, float* d_pos, int acc_mod, int acc_pos,
                    int B, int T, int L, int m, int H, int max_F,
                    void* workspace, int64_t workspace_bytes, mmfm_stream stream);

/* ---------------------------------------------------------------------------------- loader collate
 * BaseDataset._preprocess_ibl_data + get_binned_spikes_from_sparse (loader/base.py:304-450,
 * utils/dataset_utils.py:38-43), pad_to_right path: B trials given as concatenated CSR pieces
 * (uint8 counts, int32 column indices, int64 row pointers) -> dense spikes [B][max_T][max_N] fp32,
 * rows/columns beyond a trial's (T_b, N_b) = pad_value, longer trials truncated; time_mask [B][max_T]
 * and space_mask [B][max_N] int64 (1 = real data).  Duplicate (row, col) entries add up. */
int mmfm_collate_csr(int B, int max_T, int max_N, float pad_value, const uint8_t* data, const int32_t* indices,
                     const int64_t* indptr, const int64_t* indptr_off, const int64_t* nnz_off, const int32_t* T_b,
                     const int32_t* N_b, float* out, int64_t* time_mask, int64_t* space_mask, mmfm_stream stream);

/* ---------------------------------------------------------------------------------- masked loss
 * mm.py:79-82,217-239.  kind 0: PoissonNLL(log_input) exp(p) - t*p;  kind 1: MSE (p-t)^2.
 * pred [R][N] (dtype), target [R][N] fp32, rowmask u8 [R] (element (b,t) at rowmask[b*mask_ld + t]).
 * fwd writes the modality's masked SUM to loss_sum[0] (fp32, deterministic two-stage). */
int64_t mmfm_masked_loss_workspace(int64_t R, int N);
int mmfm_masked_loss_fwd(int dtype, int kind, const void* pred, const float* target, const uint8_t* rowmask,
                         int mask_ld, int T, int64_t R, int N, float* loss_sum,
                         void* workspace, int64_t workspace_bytes, mmfm_stream stream);
/* loss = sum_m loss_sum[m] / sum_m count[m]   (0/0 -> NaN like the reference);  inv_n = 1/sum count */
int mmfm_loss_finalize(const float* loss_sum, const int64_t* count, int M, float* loss, float* inv_n,
                       mmfm_stream stream);
/* dpred = grad_out[0] * inv_n[0] * rowmask * d/dp loss_elem */
int mmfm_masked_loss_bwd(int dtype, int kind, const void* pred, const float* target, const uint8_t* rowmask,
                         int mask_ld, int T, int64_t R, int N, const float* grad_out, const float* inv_n,
                         void* dpred, mmfm_stream stream);

/* ---------------------------------------------------------------------------------- elementwise
 * dst = dropout(src) with counter row*N + col (the backward of a dropout whose forward was fused
 * into a GEMM epilogue). */
int mmfm_dropout_apply(int dtype, const void* src, void* dst, int64_t R, int N, mmfm_dropout drop,
                       mmfm_stream stream);
int mmfm_cast_f32_to_bf16(const float* src, void* dst, int64_t n, mmfm_stream stream);

/* ---------------------------------------------------------------------------------- optimiser
 * torch.optim.AdamW single step over a flat fp32 parameter range (train_multi_modal.py:197-202,
 * trainer/base.py:196).  hyper is a DEVICE array of 8 floats the host computes in double:
 *   [1-lr*wd, 1-beta1, beta2, 1-beta2, lr/bias_correction1, sqrt(bias_correction2), eps, grad_scale]
 * (OneCycleLR rewrites lr AND beta1 every step, so they are data, not launch constants; g is
 * multiplied by grad_scale first, e.g. 1/world_size after a SUM all-reduce).
 * If p_bf16 != NULL the updated parameters are also written as bf16 (throughput mode weights). */
int mmfm_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n,
                    const float* hyper, mmfm_stream stream);

/* ---------------------------------------------------------------------------------- evaluation metrics
 * utils/utils.py:107-115 (metrics_list "r2" = torcheval R2Score per series; trainer/base.py:252-262 calls it for 50
 * neurons x every trial on the host):  out[g*C + c] = 1 - sum_s (y-p)^2 / sum_s (y - mean_s y)^2 over the S elements
 * y = gt[g*gs[0] + s*gs[1] + c*gs[2]] (element strides, so transposed / sliced views need no copy), fp64 sums.
 * A constant series gives -inf / nan like the reference; the caller masks invalids (np.ma.masked_invalid). */
int mmfm_r2_series(const float* gt, const int64_t* gt_strides, const float* pred, const int64_t* pred_strides,
                   int G, int S, int C, float* out, mmfm_stream stream);
/* utils/eval_utils.py:1051-1119 (neg_log_likelihood, bits_per_spike): rates, spikes fp32 [R][N] (R = trials x bins);
 * out[0] = bits per spike, out[1] = nll(model), out[2] = nll(null = per-neuron mean rate), out[3] = total spikes.
 * rates == 0 -> 1e-9 as upstream; NaN spikes are not supported (upstream masks them). */
int64_t mmfm_bits_per_spike_workspace(int64_t R, int N);
int mmfm_bits_per_spike(const float* rates, const float* spikes, int64_t R, int N, float* out,
                        void* workspace, int64_t workspace_bytes, mmfm_stream stream);
/* utils/eval_utils.py:846-851 (spiking_activity_recon_eval: bits_per_spike on each neuron's own slice, N host calls
 * upstream): out[n] = bits per spike of neuron n against ITS mean rate, all N in one pass.  A silent neuron gives
 * inf / nan like upstream (which then records NaN). */
int64_t mmfm_bits_per_spike_neurons_workspace(int64_t R, int N);
int mmfm_bits_per_spike_neurons(const float* rates, const float* spikes, int64_t R, int N, float* out,
                                void* workspace, int64_t workspace_bytes, mmfm_stream stream);


/* ---------------------------------------------------------------------------------- row-owner fused kernels (bf16, width 256)
 * Throughput-mode kernels in which a wavefront owns 32 token rows and keeps them in registers through a chain of ops while
 * the weights stream through LDS (csrc/rowchain.h).  They replace, for hidden_size 256 / inter_size 512 in bf16 mode, the
 * LayerNorm + nn.Linear pairs, the attention out_proj + residual, the whole MLP block and their autograd
 * (encoder_embeddings.py:106-116, decoder_embeddings.py:133-147, mm_utils.py:42-52,107-114,145-152, mm.py:290-292).
 *
 * mmfm_prep_weights: per training step (weights change) the LayerNorm affine is folded into the linear it feeds,
 *   Wp[n][k] = bf16(W[n][k] * gamma[k]),  WpT = Wp^T,  bp[n] = bias[n] + sum_k W[n][k] * beta[k]
 * so that  linear(layernorm(x)) = Wp . x_hat + bp  with  x_hat = (x - mean) * rstd.  gamma / beta / bias / Wp / WpT / bp / WpP / WpTP
 * may be NULL (plain bf16 copies / transposes of a weight).  `entries` is a DEVICE array; entry e covers blocks
 * [tile0, tile0 + ceil(N/32)); total_tiles = sum of ceil(N/32). */
typedef struct {
    const float* W;          /* [N][K] fp32 master weight */
    const float* gamma;      /* [K] or NULL */
    const float* beta;       /* [K] or NULL */
    const float* bias;       /* [N] or NULL */
    void* Wp;                /* bf16 [N][K] or NULL */
    void* WpT;               /* bf16 [K][N] or NULL */
    float* bp;               /* fp32 [N] or NULL */
    int N, K;
    int tile0;
    int pad_;
    void* WpP;               /* bf16 [N][K] or NULL: Wp with the 8-byte units of every aligned 32-byte group of a row in the order
                                0, 2, 1, 3 ("unit-permuted": the order in which an MFMA accumulator tile, used as the next product's
                                operand, holds its k index - rowchain.h).  LDS-DMA cannot permute on the way in, so the kernels that
                                multiply such operands read these copies: mmfm_mlp_fwd's w_down */
    void* WpTP;              /* bf16 [K][N] or NULL: WpT unit-permuted along N: mmfm_mlp_bwd's w_up_t.
                                The permutation acts on aligned groups of 16 elements: WpP requires K % 16 == 0 and WpTP requires
                                N % 16 == 0 (a ragged last group has no permuted position inside its row: the kernel leaves those
                                elements unwritten rather than spill into the next row) */
} mmfm_prep_entry;
int mmfm_prep_weights(const mmfm_prep_entry* entries, int n_entries, int total_tiles, mmfm_stream stream);

/* y[R][N] = epi( pro(x)[R][K] . w[N][K]^T ), bf16 storage, fp32 accumulate; K in {256, 512, 768}, N % 32 == 0.
 *   ln != 0 (K = 256): pro(x) = x_hat = (x - mean(x)) * rstd(x) per row (statistics in fp32); x_hat (bf16 [R][256]) and rstd
 *                      (fp32 [R]) are written when non-NULL (the backward's saved tensors); w / bias are the PREPARED Wp / bp.
 *   epilogue: + bias[n], + residual[m*ldr + n], store.
 *   ln_bwd != 0 (N = 256): v = x . w^T is d(x_hat) of a LayerNorm whose output fed the forward linear (w = WpT of it) and
 *                      y = residual + bwd_rstd * (v - mean(v) - bwd_xhat * mean(v * bwd_xhat))   (residual = running gradient or NULL). */
typedef struct {
    int64_t R;
    int K, N;
    const void* x; int ldx;
    const void* w; int ldw;
    const float* bias;
    int ln; float eps;
    void* xhat; float* rstd;
    const void* residual; int ldr;
    void* y; int ldy;
    int stream_out;          /* 1: non-temporal stores of y */
    int rotate;              /* 1: workgroups start at different weight tiles (spreads the concurrent L2 reads) */
    int ln_bwd;
    const void* bwd_xhat; const float* bwd_rstd;
} mmfm_rowgemm_desc;
int mmfm_rowgemm(const mmfm_rowgemm_desc* d, mmfm_stream stream);

/* The MLP block in one launch (mm_utils.py:42-52 behind ln2, encoder_embeddings.py:114, decoder_embeddings.py:145):
 *   fwd:  y = x + dropout( down( gelu_erf( up( layernorm(x) ) ) ) )        the 512-wide intermediate never leaves the CU
 *         w_up / b_up are the prepared (gamma / beta folded) [512][256] / [512]; w_down = bf16 [256][512], UNIT-PERMUTED
 *         (mmfm_prep_entry.WpP of down_proj: the weights reach LDS by DMA, which cannot permute); x_hat / rstd are written for the backward.
 *   bwd:  recomputes u = up(x_hat) and g = gelu(u) from the saved x_hat instead of loading them, and produces
 *         t1 = dropout'(dy)                      [R][256]   (operand of dW_down = t1^T g, db_down = colsum t1)
 *         g                                      [R][512]
 *         du = (t1 . W_down) * gelu'(u)          [R][512]   (operand of G_up = du^T x_hat, db_up = colsum du)
 *         dx = dy + LayerNorm'(du . Wp_up)       [R][256]   (LayerNorm backward in registers; needs w_up_t = Wp_up^T UNIT-PERMUTED =
 *                                                            mmfm_prep_entry.WpTP of up_proj, w_down_t = W_down^T)
 *   Weight / LayerNorm-parameter gradients then follow from mmfm_gemm (dW slabs) + mmfm_ln_linear_grad. */
typedef struct {
    int64_t R;
    const void* x; int ldx;           /* fwd: residual stream in (bf16 [R][256]) */
    float eps;
    const void* w_up; const float* b_up;
    const void* w_down; const float* b_down;      /* w_down: unit-permuted (WpP) */
    mmfm_dropout drop;
    void* y; int ldy;
    void* xhat; float* rstd;          /* fwd: out;  bwd: in */
    /* backward only */
    const void* dy; int lddy;
    const void* w_down_t;             /* bf16 [512][256] */
    const void* w_up_t;               /* bf16 [256][512] (prepared, unit-permuted: WpTP) */
    void* t1; void* g; void* du;
    void* dx; int lddx;               /* dx == NULL: front half only - t1, g, du are written and the call returns; rstd / w_up_t are not read.
                                         The caller finishes with mmfm_rowgemm(x = du, w = WpT of up_proj, K = 512, residual = dy, ln_bwd) */
    int rotate;                       /* 1: workgroups start at different intermediate tiles (spreads the concurrent L2 reads) */
} mmfm_mlp_desc;
int mmfm_mlp_fwd(const mmfm_mlp_desc* d, mmfm_stream stream);
int mmfm_mlp_bwd(const mmfm_mlp_desc* d, mmfm_stream stream);

/* Gradients of a LayerNorm-fed linear from the reduced weight-gradient GEMM against x_hat:
 *   Gdb = [ G[N][K] | db[N] ],  G = dY^T x_hat,  db = colsum dY   (one mmfm_gemm + mmfm_reduce_slabs)
 *   dW[n][k] = gamma[k] * G[n][k] + db[n] * beta[k];   dbias = db;
 *   dgamma[k] (+)= sum_n W[n][k] * G[n][k];   dbeta[k] (+)= sum_n W[n][k] * db[n]      (accumulate_ln selects +=)
 * No per-row reduction is needed for the LayerNorm parameters.  Deterministic.
 * workspace: mmfm_ln_linear_grad_workspace(K) bytes, ZEROED once before its first use (partial rows + arrival tickets that
 * the kernel re-arms itself); one workspace per concurrently running launch. */
int64_t mmfm_ln_linear_grad_workspace(int K);
int mmfm_ln_linear_grad(const float* Gdb, const float* W, const float* gamma, const float* beta, int N, int K,
                        float* dW, float* dbias, float* dgamma, float* dbeta, int accumulate_ln,
                        void* workspace, int64_t workspace_bytes, mmfm_stream stream);

#ifdef __cplusplus
}
#endif
#endif
