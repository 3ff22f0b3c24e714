/*
 * libmixgrpo_hip.so -- C ABI of the MI355X-native MixGRPO rollout-and-update hot path.
 *
 * The reference (zqqqqz2000/MixGRPO) is pure Python and has no FFI layer; its boundary for this path is
 * the function surface of fastvideo/utils/sampling_utils.py and fastvideo/train_grpo_flux.py, which calls
 * PyTorch eager ops.  Every entry point below replaces the eager-op sequence of one reference function
 * (cited per function, paths relative to the reference root).  The Python mirror in mixgrpo_amd/ binds
 * them with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions: plain pointers to DEVICE memory unless a parameter says "host"; sizes as int/long; scalars
 * by value; `stream` is a hipStream_t passed as void*.  No allocation, no host sync, no global mutable
 * state except the thread-local last-error string.  Returns 0 (MGX_OK) or a negative error code; the
 * message is available from mgx_last_error().  bf16 tensors are passed as uint16_t*.
 */
#ifndef MIXGRPO_HIP_H
#define MIXGRPO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int mgx_version(void);
const char* mgx_last_error(void);

/* ------------------------------------------------------------------------------------------------ solver
 * Host-computed per-step scalars.  They are produced on the host (mixgrpo_amd/sampling_utils.py) in the
 * reference's own fp32 operation order, including which scalars PyTorch rounds to bf16 before multiplying
 * a bf16 tensor (oracle/solver.py header), so the kernels only apply them.
 */
typedef struct {
  float sigma_x0; /* x0   = x - bf16(v * sigma_x0)                       sampling_utils.py:175 */
  float c_x;      /* mean = x * c_x + bf16(bf16(v * c_v) * dt_mean)      sampling_utils.py:186 */
  float c_v;
  float dt_mean;
  float sd_noise; /* prev = mean + bf16(noise * sd_noise)                sampling_utils.py:195 */
  float dt_det;   /* deterministic: prev = x + bf16(v * dt_det)          sampling_utils.py:199 */
  float den;      /* 2*sd^2                                              sampling_utils.py:202 */
  float log_sd;   /* log(sd)                                             sampling_utils.py:203 */
  float log_c;    /* log(sqrt(2*pi))                                     sampling_utils.py:204 */
} mgx_flow_coeffs;

/* Elements of `ws` (double) the log-prob reduction needs for a [B, n] problem. */
long mgx_logp_workspace_elems(int B, long n);

/* flow_grpo_step forward (sampling_utils.py:157-210).  x fp32 [B,n]; v bf16 [B,n].
 * Rollout: noise bf16 [B,n] given, prev_in NULL -> writes prev_out.  Replay (training, :149-157 of
 * train_grpo_flux.py): prev_in fp32 given, noise NULL, prev_out may be NULL.  deterministic!=0 applies the
 * ODE override (:198-199) after the draw.  x0_out / mean_out are optional (NULL to skip).
 * logp fp32 [B] = mean over n of the Gaussian log-density (:201-208). */
int mgx_flow_step_fwd(const float* x, const uint16_t* v, const uint16_t* noise, const float* prev_in,
                      float* prev_out, float* x0_out, float* mean_out, float* logp, double* ws, int B, long n,
                      const mgx_flow_coeffs* k /*host*/, int deterministic, void* stream);

/* d logp / d v for the replay path (autograd of sampling_utils.py:186,201-208): dv bf16 [B,n],
 * g_logp fp32 [B] is dLoss/dlogp. */
int mgx_flow_step_bwd(const float* x, const uint16_t* v, const float* prev, const float* g_logp, uint16_t* dv,
                      int B, long n, const mgx_flow_coeffs* k /*host*/, void* stream);

typedef struct {
  float ds_r;     /* mean0 = x + bf16(v * ds_r)        sampling_utils.py:224 (scalar rounded to bf16) */
  float s_r;      /* x0    = x - bf16(v * s_r)         sampling_utils.py:226 */
  float ds;       /* unrounded sigma_next - sigma (score-correction term :234) */
  float ds_b;     /* scalar autograd uses for d(ds*v)/dv (unrounded on CPU semantics) */
  float s_b;      /* scalar autograd uses for d(s*v)/dv */
  float one_m_s;  /* 1 - sigma                         sampling_utils.py:232 */
  float s_sq;     /* sigma^2 */
  float half_eta2;/* -0.5*eta^2 as fp32                sampling_utils.py:233 */
  float sd;       /* eta*sqrt(sigma - sigma_next) as fp32 :229 */
  float den;      /* 2*sd^2 as fp32                    :245 */
} mgx_dance_coeffs;

/* dance_grpo_step forward with grpo=True (sampling_utils.py:212-253).  noise fp32 [B,n] (SDE rollout) or
 * NULL; prev_in fp32 (replay) or NULL; sde!=0 adds the score correction (:231-234). */
int mgx_dance_step_fwd(const float* x, const uint16_t* v, const float* noise, const float* prev_in,
                       float* prev_out, float* x0_out, float* logp, double* ws, int B, long n,
                       const mgx_dance_coeffs* k /*host*/, int sde, void* stream);
int mgx_dance_step_bwd(const float* x, const uint16_t* v, const float* prev, const float* g_logp, uint16_t* dv,
                       int B, long n, const mgx_dance_coeffs* k /*host*/, int sde, void* stream);

typedef struct {
  int order;        /* 1..3: which multistep update (sampling_utils.py:327-357) */
  int sde;          /* x = mean + sd_noise*noise instead of the ODE combination */
  float sigma_x0;   /* x0 = sample - bf16(v*sigma_x0)            :387-396 */
  float inv_r0;     /* D1_0 = inv_r0*(m0-m1)                     :490,608 */
  float inv_r1;     /* D1_1 = inv_r1*(m1-m2)                     :608 */
  float c_r;        /* D1 = D1_0 + c_r*(D1_0-D1_1)               :609 */
  float inv_r01;    /* D2 = inv_r01*(D1_0-D1_1)                  :610 */
  float cm[4];      /* mean = cm0*sample + cm1*D0 + cm2*D1 + cm3*D2 (signed, summed left to right) */
  float cx[4];      /* x    = cx0*sample + cx1*D0 + cx2*D1 + cx3*D2 */
  float sd_noise;   /* std*dt_sqrt */
  float den;        /* 2*(std*dt_sqrt)^2 */
  float log_sd;
  float log_c;
} mgx_dpm_coeffs;

/* dpm_step (sampling_utils.py:273-385) for DPM-Solver / DPM-Solver++ orders 1-3.  m1/m2 are the previous
 * x0 predictions (fp32, NULL when order is lower); x0_out receives this step's x0 (the new m0). */
int mgx_dpm_step_fwd(const float* sample, const uint16_t* v, const float* m1, const float* m2, const float* noise,
                     float* x_out, float* x0_out, float* logp, double* ws, int B, long n,
                     const mgx_dpm_coeffs* k /*host*/, void* stream);

/* d log_prob / d model_output of a first-order SDE dpm_step whose sample x_t is held fixed: the training replay under
 * dpm_apply_strategy="all" (train_grpo_flux.py:170-180 calls dpm_step with dpm_state=None; the log-prob's gradient runs
 * through prev_sample_mean, sampling_utils.py:376-383).  sigma_b: the scalar autograd multiplies by in d(sigma*v)/dv. */
int mgx_dpm_step_bwd(const float* sample, const uint16_t* v, const float* x_t, const float* g_logp, uint16_t* dv, int B,
                     long n, const mgx_dpm_coeffs* k /*host, order 1*/, float sigma_b, void* stream);

/* convert_model_output alone (sampling_utils.py:387-396; used at :116 to feed DPMState inside the window) */
int mgx_x0_pred(const float* sample, const uint16_t* v, float* x0_out, long total, float sigma_x0, void* stream);

/* pack_latents / unpack_latents (train_grpo_flux.py:94-115): [B,C,H,W] <-> [B,(H/2)(W/2),4C]; elem_size 2|4 */
int mgx_pack_latents(const void* in, void* out, int B, int C, int H, int W, int elem_size, void* stream);
int mgx_unpack_latents(const void* in, void* out, int B, int C, int H, int W, int elem_size, void* stream);

/* ------------------------------------------------------------------------------------------------ GRPO
 * Group-relative advantage (train_grpo_flux.py:440-491): per group of G consecutive rewards,
 * (r-mean)/(unbiased std+1e-8), statistics over the upper (1-trimmed_ratio) part when trimmed_ratio>0;
 * out[i] (+)= weight * advantage (accumulate!=0 implements the multi-reward sum of :465-467). */
int mgx_group_advantage(const float* rewards, float* out, int n, int G, float trimmed_ratio, float weight,
                        int accumulate, void* stream);
/* Global normalisation for use_group=False (:498): (r - mean(all))/(std(all)+1e-8); all = gathered rewards */
int mgx_global_advantage(const float* rewards, const float* gathered, float* out, int n, int n_all, void* stream);

/* PPO-clip GRPO loss per replayed transition (train_grpo_flux.py:560-583), batch of B independent
 * (sample, step) pairs: loss/policy/kl/clip_frac per element and g_logp = dloss/dnew_logp. */
int mgx_grpo_loss(const float* new_logp, const float* old_logp, const float* adv, int B, float clip_range,
                  float adv_clip_max, float kl_coeff, float denom, float* loss, float* policy, float* kl,
                  float* clip_frac, float* g_logp, void* stream);

/* ------------------------------------------------------------------------------------------------ MMDiT
 * The FLUX MMDiT is third-party code for the reference (diffusers 0.32.2 FluxTransformer2DModel, call sites
 * fastvideo/utils/sampling_utils.py:68-82 and fastvideo/train_grpo_flux.py:134-144); each entry point below
 * replaces the torch/cuBLAS/SDPA kernels that module launches under autocast(bf16).
 *
 * Row-batched matrices: row m lives at base + (m / rpb) * bstride + (m % rpb) * ld (elements).  A plain
 * matrix has rpb >= M.  This lets the text rows and image rows of the joint [B, S, d] stream be operands
 * without a concat/split copy.
 */
enum { MGX_EPI_BIAS = 0, MGX_EPI_BIAS_GELU = 1, MGX_EPI_BIAS_GATE_RES = 2, MGX_EPI_F32_ACC = 3, MGX_EPI_DGELU = 4 };

/* C[M,N] = epi(A[M,K] @ W[N,K]^T + bias[N]); bf16 in, fp32 accumulate (MFMA), one rounding to bf16
 * (nn.Linear under autocast).  Epilogues:
 *   MGX_EPI_BIAS          C = bf16(acc + bias)
 *   MGX_EPI_BIAS_GELU     C = bf16(gelu_tanh(bf16(acc + bias)));  aux (optional) receives the pre-activation
 *   MGX_EPI_BIAS_GATE_RES C = bf16(C + bf16(gate[m/c_rpb, n] * bf16(acc + bias)));  aux (optional) <- pre-gate
 *   MGX_EPI_F32_ACC       C(fp32) = beta * C + acc                  (weight gradients)
 *   MGX_EPI_DGELU         C = bf16(bf16(acc) * gelu_tanh'(aux))     (input gradient through GELU)
 * aux is a plain [M, ldaux] matrix.  K % 64 == 0, N % 4 == 0; M, N tails are masked. */
int mgx_gemm_bf16(const uint16_t* A, const uint16_t* W, const uint16_t* bias, void* C, const uint16_t* gate,
                  uint16_t* aux, long ldaux, int M, int N, int K, long lda, long a_rpb, long a_bstride, long ldw, long ldc,
                  long c_rpb, long c_bstride, long gate_ld, int epilogue, float beta, void* stream);
/* The same GEMM with a caller-owned fp32 workspace of mgx_gemm_sk_workspace_elems() floats (one per stream in use): the
 * persistent kernel may then share the tiles of its last, partial round out along K ("stream-K tail": partial sums through
 * the workspace, added in K order by a second launch -- deterministic, and a given shape is always split the same way).
 * Results can differ from mgx_gemm_bf16's in the last bit of the fp32 sum.  sk_workspace == NULL: identical to mgx_gemm_bf16.
 * Replaces the same torch/cuBLAS nn.Linear launches (fastvideo/utils/sampling_utils.py:68-82, train_grpo_flux.py:134-144,600). */
long mgx_gemm_sk_workspace_elems(void);
int mgx_gemm_bf16_sk(const uint16_t* A, const uint16_t* W, const uint16_t* bias, void* C, const uint16_t* gate,
                     uint16_t* aux, long ldaux, int M, int N, int K, long lda, long a_rpb, long a_bstride, long ldw, long ldc,
                     long c_rpb, long c_bstride, long gate_ld, int epilogue, float beta, float* sk_workspace,
                     long sk_workspace_elems, void* stream);
/* The q | k projection of an attention layer with mgx_qk_norm_rope_fwd_qs applied in the GEMM's epilogue, on the tile while
 * it is in registers: X [B * rows_per_batch, K] (plain), Wqk [2 * H * 128, K] (to_q rows, then to_k rows), bias [2 * H * 128]
 * -> Q, K [B, H, S, 128] at sequence positions s0 .. s0 + rows_per_batch - 1, Q times q_scale.  The [tokens, 2 H 128]
 * projection output is neither written nor re-read.  Same bits as mgx_gemm_bf16 followed by mgx_qk_norm_rope_fwd_qs
 * (diffusers' to_q / to_k Linears + norm_q / norm_k + apply_rotary_emb; call sites fastvideo/utils/sampling_utils.py:68-82).
 * cos_sin_pairs (optional): [S, 64, 2] fp32 = (cos[s][2i], sin[s][2i]) for tables whose two entries of a rotation pair are
 * equal (diffusers' FluxPosEmbed repeat_interleaves them) -- the caller's promise; the epilogue then reads half the table
 * bytes, which are what it costs (profiles/r04_qknorm_epilogue_prices.log).
 * Returns 1 -- nothing launched -- when the persistent kernel cannot take the problem (fewer than 128 output tiles of
 * 256 x 256, H odd, rows_per_batch % 128 != 0, alignments below 16 bytes, MGX_GEMM_QKNORM=0): keep the two-pass form. */
int mgx_linear_qk_norm_rope(const uint16_t* X, const uint16_t* Wqk, const uint16_t* bias, const float* wq, const float* wk,
                            const float* cos, const float* sin, const float* cos_sin_pairs /* optional, see above */,
                            uint16_t* Q, uint16_t* K, int B, int H, int S, int rows_per_batch, int s0, int Kdim, long ldx,
                            long ldw, float q_scale, void* stream);

/* A Linear whose output leaves TRANSPOSED: Ct[b][f][t] = bf16(X[b * tok_rpb + t, :] . W[f, :] + bias[f]), ld_ct elements
 * between feature rows, ct_bstride between token batches -- the V^T [B, H, 128, Sp] operand of mgx_attn_fwd* straight from
 * the value projection (the nn.Linear to_v / add_v_proj of diffusers' FluxAttnProcessor2_0 followed by the transpose SDPA's
 * kernels do internally; call sites fastvideo/utils/sampling_utils.py:68-82), mgx_qk_norm_rope_fwd* then called with Vt = NULL.
 * Same persistent kernel with the operand roles swapped.  Returns 1 -- nothing launched -- for shapes that kernel cannot take
 * (fewer than 128 output tiles of 256 x 256, tok_rpb % 64 != 0 with several batches, alignments below 16 bytes): the caller
 * then keeps the plain projection + mgx_qk_norm_rope_fwd's transposing pass. */
int mgx_linear_bf16_t(const uint16_t* X, const uint16_t* W, const uint16_t* bias, uint16_t* Ct, int tokens, int F, int K,
                      long ldx, long ldw, long ld_ct, long tok_rpb, long ct_bstride, float* sk_workspace,
                      long sk_workspace_elems, void* stream);

/* TWO such problems with equal N, K, epilogue and leading dimensions in ONE launch of the persistent kernel: the text- and the
 * image-stream Linear of a FLUX double block, which diffusers issues as separate nn.Linear calls (to_q/k/v | add_q/k/v_proj,
 * to_out | to_add_out, ff | ff_context; call sites fastvideo/utils/sampling_utils.py:68-82, train_grpo_flux.py:134-144,600).
 * Problem 1's M1 rows come first in the tile walk, so the text stream's few tile rows ride the image stream's rounds.
 * Grouped when M1 % 256 == 0, the epilogue is not MGX_EPI_F32_ACC, the two A operands lie within 4 GiB of each other and so do
 * the two W operands (same activation buffer, same block of the parameter store) and the problem is large enough for the persistent kernel; otherwise
 * the call issues the two problems one after the other (mgx_gemm_bf16_sk) -- same results either way, bit for bit per tile. */
int mgx_gemm_bf16_pair(const uint16_t* A1, const uint16_t* W1, const uint16_t* bias1, void* C1, const uint16_t* gate1,
                       uint16_t* aux1, int M1, long a1_rpb, long a1_bstride, long c1_rpb, long c1_bstride,
                       const uint16_t* A2, const uint16_t* W2, const uint16_t* bias2, void* C2, const uint16_t* gate2,
                       uint16_t* aux2, int M2, long a2_rpb, long a2_bstride, long c2_rpb, long c2_bstride, int N, int K,
                       long lda, long ldw, long ldc, long ldaux, long gate_ld, int epilogue, float beta,
                       float* sk_workspace, long sk_workspace_elems, void* stream);

/* out[N, ld_out] = in[M, N]^T (bf16; columns M..ld_out-1 are zero-filled) and, optionally, fp32 column sums
 * colsum_out[n] = beta*colsum_out[n] + sum_m in[m, n] (bias gradients) via a deterministic two-stage reduction;
 * colsum_partial needs mgx_transpose_partial_elems(M, N) floats. */
long mgx_transpose_partial_elems(int M, int N);
int mgx_transpose_bf16(const uint16_t* in, uint16_t* out, float* colsum_partial, float* colsum_out, float colsum_beta,
                       int M, int N, long ld_in, long in_rpb, long in_bstride, long ld_out, void* stream);

/* y[M, D] = bf16( LayerNorm(x[m,:]; eps 1e-6, no affine) * bf16(1 + scale[b,:]) + shift[b,:] ), b = m / x_rpb
 * (AdaLayerNormZero / ...Single / ...Continuous normalisation).  shift/scale point at their chunk inside the
 * modulation vector [B, mod_ld].  stats (optional) receives (mean, rstd) per row.  D % 512 == 0, D <= 4096. */
int mgx_ln_modulate_fwd(const uint16_t* x, long ldx, long x_rpb, long x_bstride, const uint16_t* shift,
                        const uint16_t* scale, long mod_ld, uint16_t* y, long ldy, float* stats, long M, int D,
                        void* stream);
/* Backward of the above: dx (+)= dLN, dshift/dscale (bf16, [B, mod_ld] chunks) = per-batch column sums. */
long mgx_ln_modulate_bwd_workspace(long M, long rpb, int D);
int mgx_ln_modulate_bwd(const uint16_t* dy, long lddy, const uint16_t* x, long ldx, long x_rpb, long x_bstride,
                        const uint16_t* scale, long mod_ld, uint16_t* dx, long lddx, long dx_rpb, long dx_bstride,
                        int accumulate, uint16_t* dshift, uint16_t* dscale, float* ws, long M, int D, void* stream);

/* Per-head RMSNorm(eps 1e-6, fp32 weight[128]) on q,k + interleaved-pair RoPE (fp32 cos/sin [S,128]) + head
 * split: qkv [B*rows_per_batch, 3*H*128] -> Q,K [B,H,S,128] (rounded once to bf16), Vt [B,H,128,Sp], written
 * at sequence positions s0 .. s0+rows_per_batch-1 of the joint sequence (diffusers FluxAttnProcessor2_0). */
/* V, Qt, Kt (all or none): extra layouts the attention backward consumes -- V row-major [B,H,S,128] and
 * Q^T, K^T [B,H,128,Sp] (padding must be finite: allocate zeroed).  Vt = NULL (without the extras): the v columns are not
 * read and V^T is not written (mgx_linear_bf16_t wrote it). */
int mgx_qk_norm_rope_fwd(const uint16_t* qkv, long ld, const float* wq, const float* wk, const float* cos,
                         const float* sin, uint16_t* Q, uint16_t* K, uint16_t* Vt, uint16_t* V, uint16_t* Qt,
                         uint16_t* Kt, int B, int H, int S, int Sp, int rows_per_batch, int s0, void* stream);
/* ..._qs: Q (and Qt) leave as bf16(q * q_scale) -- one rounding, of the scaled value.  With q_scale = softmax scale * log2(e)
 * the scores Q K^T are exponents of two: what mgx_attn_fwd_log2 consumes; every other consumer of that Q takes `scale` = ln 2
 * (mgx_attn_bwd, mgx_attn_fwd_fp8), and mgx_qk_norm_rope_bwd_qs multiplies the incoming dQ by the same q_scale. */
int mgx_qk_norm_rope_fwd_qs(const uint16_t* qkv, long ld, const float* wq, const float* wk, const float* cos,
                            const float* sin, uint16_t* Q, uint16_t* K, uint16_t* Vt, uint16_t* V, uint16_t* Qt,
                            uint16_t* Kt, int B, int H, int S, int Sp, int rows_per_batch, int s0, float q_scale,
                            void* stream);
long mgx_qk_norm_rope_bwd_workspace(int B, int H, int rows_per_batch);
int mgx_qk_norm_rope_bwd(const uint16_t* qkv, long ld, const float* wq, const float* wk, const float* cos,
                         const float* sin, const uint16_t* dQ, const uint16_t* dK, const uint16_t* dV, uint16_t* dqkv,
                         long ld_dqkv /* elements between rows of dqkv, >= ld */, float* gwq, float* gwk, float* ws, int B,
                         int H, int S, int Sp, int rows_per_batch, int s0, void* stream);

int mgx_qk_norm_rope_bwd_qs(const uint16_t* qkv, long ld, const float* wq, const float* wk, const float* cos,
                            const float* sin, const uint16_t* dQ, const uint16_t* dK, const uint16_t* dV, uint16_t* dqkv,
                            long ld_dqkv, float* gwq, float* gwk, float* ws, int B, int H, int S, int Sp,
                            int rows_per_batch, int s0, float q_scale, void* stream);

/* O = softmax(scale * Q K^T) V, non-causal, head_dim 128 (F.scaled_dot_product_attention under autocast):
 * Q,K [B,H,S,128], Vt [B,H,128,Sp] (Sp = S rounded up to 64, padding finite), O [B,S,ldo] at column h*128,
 * lse [B,H,S] (optional, natural log) for the backward pass. */
int mgx_attn_fwd(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, uint16_t* O, float* lse, int B, int H, int S,
                 int Sp, long ldo, long o_bstride, float scale, void* stream);

/* The same with Q2 = Q * scale * log2(e) (mgx_qk_norm_rope_fwd_qs): P = 2^(Q2 K^T - m), lse still the natural logarithm.
 * On the 64-query kernel (S % 256 == 0) the running maximum is subtracted by the matrix pipe -- a score tile's accumulator
 * starts at -m -- and the per-score multiply-add of the softmax is gone (csrc/gen/attn_fwd64.py, ACC). */
int mgx_attn_fwd_log2(const uint16_t* Q2, const uint16_t* K, const uint16_t* Vt, uint16_t* O, float* lse, int B, int H,
                      int S, int Sp, long ldo, long o_bstride, void* stream);

/* fp8 (OCP e4m3) variant of mgx_attn_fwd: the "fp8 MFMA attention path" of BASELINE.json configs[4].  The reference
 * has no fp8 attention; the entry points serve the same SDPA call sites (fastvideo/utils/sampling_utils.py:68-82,
 * train_grpo_flux.py:134-144).
 * mgx_attn_fp8_quantize: amax [3][B*H] fp32 (|max| of Q, K, V per batch-head, V over the S valid keys), Q8/K8
 * [B,H,S,128] = e4m3(x * 448 / amax), V8t [B,H,128,Sp] likewise with the keys of every 64-key block in the order
 * p = 32h + 16kb + i  <-  key 32kb + 8(i>>2) + 4h + (i&3)  (the order the kernel's P^T fragment holds them).
 * mgx_attn_fwd_fp8: O, lse as mgx_attn_fwd from the quantised operands (both contractions on e4m3 MFMA, P in e4m3,
 * fp32 statistics / accumulators).  The backward stays mgx_attn_bwd on the bf16 operands with this O / lse. */
int mgx_attn_fp8_quantize(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, uint8_t* Q8, uint8_t* K8,
                          uint8_t* V8t, float* amax, int B, int H, int S, int Sp, void* stream);
int mgx_attn_fwd_fp8(const uint8_t* Q8, const uint8_t* K8, const uint8_t* V8t, const float* amax, uint16_t* O,
                     float* lse, int B, int H, int S, int Sp, long ldo, long o_bstride, float scale, void* stream);

/* Backward of mgx_attn_fwd (P recomputed from lse): dQ, dK, dV [B,H,S,128].  Inputs Q,K,V row-major, Qt,Kt
 * [B,H,128,Sp] (mgx_qk_norm_rope_fwd extras), O and dO [B,S,ldo] at column h*128.  delta [B,H,S] fp32 and dOt
 * [B,H,128,Sp] bf16 are caller-provided scratch filled by the internal prep kernel. */
int mgx_attn_bwd(const uint16_t* Q, const uint16_t* K, const uint16_t* V, const uint16_t* Qt, const uint16_t* Kt,
                 const uint16_t* O, const uint16_t* dO, const float* lse, float* delta, uint16_t* dOt, uint16_t* dQ,
                 uint16_t* dK, uint16_t* dV, int B, int H, int S, int Sp, long ldo, long o_bstride, float scale,
                 void* stream);

/* out[b, :] = bf16(x[b, :] @ W[N,K]^T + bias), 1 <= Bn <= 16 rows (temb MLPs, AdaLN modulation linears) */
int mgx_skinny_linear(const uint16_t* x, long ldx, const uint16_t* W, long ldw, const uint16_t* bias, uint16_t* out,
                      long ldo, int Bn, int N, int K, void* stream);
/* dW[N,K] (fp32) += dout[Bn,N]^T x[Bn,K]; dbias[N] (fp32, optional) += column sums of dout */
int mgx_skinny_wgrad(const uint16_t* dout, long ldd, const uint16_t* x, long ldx, float* dW, long ldw, float* dbias,
                     int Bn, int N, int K, void* stream);
/* dx[Bn,K] (bf16; += when accumulate) = bf16(dout[Bn,N] W[N,K]), 1 <= Bn <= 8: the input gradient of the same skinny linears
 * (autograd of diffusers' AdaLayerNormZero(.Single).linear / AdaLayerNormContinuous.linear, called from
 * train_grpo_flux.py:600 `loss.backward()`), read straight from the row-major weight.  ws: fp32 scratch of
 * mgx_skinny_dgrad_workspace(Bn, K) elements. */
long mgx_skinny_dgrad_workspace(int Bn, int K);
int mgx_skinny_dgrad(const uint16_t* dout, long ldd, const uint16_t* W, long ldw, uint16_t* dx, long lddx, float* ws,
                     int Bn, int N, int K, int accumulate, void* stream);
/* bf16 elementwise: op 0 y=silu(a) | 1 y=b*silu'(a) | 2 y=a+b | 3 y+=a */
int mgx_ew_bf16(const uint16_t* a, const uint16_t* b, uint16_t* y, long n, int op, void* stream);
/* diffusers Timesteps(256, flip_sin_to_cos=True, shift 0): out[b] = bf16([cos(t_b f) | sin(t_b f)]) */
int mgx_sincos_embed(const float* t, uint16_t* out, int Bn, void* stream);
int mgx_cast_f32_bf16(const float* x, uint16_t* y, long n, void* stream);
/* y = scale * float(x) (n % 8 == 0).  With mgx_cast_f32_bf16: the bf16 gradient buckets of the data-parallel all-reduce
 * (the reference's FSDP reduce-scatters its gradients per wrapped block, fastvideo/utils/fsdp_util.py:56-66) */
int mgx_cast_bf16_f32(const uint16_t* x, float* y, long n, float scale, void* stream);
/* out[N, ld_out] = (bf16(gelu_tanh(float(in[M, N]))))^T, columns M..ld_out-1 zero-filled: the weight-gradient operand of
 * ff.net.2 / proj_out formed straight from a KEPT pre-activation (the values mgx_gelu_bf16 would write, transposed; autograd of
 * diffusers' FeedForward / FLUX single-block act_mlp under train_grpo_flux.py:600 `loss.backward()`).  N, ld_in, ld_out % 8 == 0. */
int mgx_transpose_gelu_bf16(const uint16_t* in, uint16_t* out, int M, int N, long ld_in, long ld_out, void* stream);
/* y[m][0..N) = bf16(gelu_tanh(float(x[m][0..N)))), rows ldx / ldy elements apart (N, ldx, ldy % 8 == 0, 16-byte aligned):
 * the activation of diffusers' FeedForward(activation_fn="gelu-approximate") / FLUX single block `act_mlp` re-created from a
 * KEPT pre-activation, bit-identical to what mgx_gemm_bf16's bias+GELU epilogue writes (it applies GELU to the bf16-rounded
 * Linear output); used by the activation-recompute pass the reference gets from torch.utils.checkpoint
 * (fastvideo/utils/fsdp_util.py:26-66) to skip a GEMM whose input it kept. */
int mgx_gelu_bf16(const uint16_t* x, long ldx, uint16_t* y, long ldy, long M, int N, void* stream);
/* Backward of x + gate*y: dy = bf16(gate[b]*dout), dgate[b,:] = sum_rows dout*y (bf16) */
long mgx_gate_bwd_workspace(long batches, long rows_per_batch, int D);
int mgx_gate_bwd(const uint16_t* dout, long ldd, long d_bstride, const uint16_t* y, long ldy, const uint16_t* gate,
                 long gate_ld, uint16_t* dy, long lddy, uint16_t* dgate, float* ws, int batches, long rows_per_batch,
                 int D, void* stream);

/* ------------------------------------------------------------------------------------------------ optimizer
 * out[0] = beta*out[0] + sum g^2 (fp64 two-stage reduction; ws >= mgx_sqnorm_workspace() doubles):
 * the global grad norm of transformer.clip_grad_norm_ (train_grpo_flux.py:606). */
long mgx_sqnorm_workspace(void);
int mgx_sqnorm_f32(const float* g, long n, double* ws, float* out, float beta, void* stream);
/* torch.optim.AdamW step (train_grpo_flux.py:715-721,607) on flat fp32 master weights, fused with the
 * clip-by-global-norm scaling (gnorm_sq: device pointer to sum g^2 of the UNSCALED grads, or NULL for no
 * clipping; grad_scale multiplies every gradient first, e.g. 1/world_size) and the bf16 compute-copy refresh. */
/* lr, betas, eps and weight decay are DOUBLES, as the Python floats torch.optim.AdamW holds them: every derived coefficient
 * (1 - beta2, lr / (1 - beta1^step), ...) is formed in double and rounded to fp32 once, like torch's foreach step does. */
int mgx_adamw_step(float* w, uint16_t* w16, const float* g, float* m, float* v, long n, double lr, double beta1,
                   double beta2, double eps, double weight_decay, int step, const float* gnorm_sq, float max_norm,
                   float grad_scale, void* stream);
int mgx_scale_f32(float* x, long n, float s, void* stream);


/* ------------------------------------------------------------------------------------------------ VAE decode
 * `vae.decode(latents)` of fastvideo/train_grpo_flux.py:279-289: diffusers' AutoencoderKL decoder (bf16 weights, bf16 autocast;
 * its block structure is the 2-D original of the reference's vendored fastvideo/models/hunyuan/vae/unet_causal_3d_blocks.py
 * :404-462 ResnetBlock, :667-693 mid block, :148-206 upsample, and of vae.py:251-312 Decoder.forward).  Activations are NHWC
 * bf16 for one image: "plain" = [H W][C], "padded" = [(H + 2)(W + 2)][C] with a zero border that is the convolution's padding.
 *
 * 3x3 convolution, stride 1, padding 1, as an implicit GEMM on the MFMA GEMM kernels (K = 9 taps x C channels, no im2col):
 * x padded NHWC, C in {64, 128, 256, 512}; Wt [Cout][3][3][C] bf16 (tap-major); out plain [H W][ld_out], Cout % 4 == 0.
 * residual != 0: out = bf16(out + bf16(conv + bias)) (`input_tensor + hidden_states` of a ResnetBlock2D); `ones` = Cout bf16 ones. */
int mgx_conv3x3_nhwc(const uint16_t* x, const uint16_t* Wt, const uint16_t* bias, uint16_t* out, long ld_out,
                     const uint16_t* ones, int H, int W, int C, int Cout, int residual, void* stream);
/* y = [silu](GroupNorm(x; G groups, eps, affine gamma / beta [C] bf16)) in fp32 on a plain bf16 image x [H W][ld], one rounding
 * to bf16 at the store.  Output pixel (y, x) goes to out + y * out_row + x * out_px (elements): a plain matrix or the interior of
 * a padded image.  ws: fp32 scratch of mgx_group_norm_workspace(H W, C, G) elements. */
long mgx_group_norm_workspace(long M, int C, int G);
int mgx_group_norm_nhwc(const uint16_t* x, long ld, const uint16_t* gamma, const uint16_t* beta, uint16_t* out, long out_row,
                        long out_px, float* ws, int H, int W, int C, int G, float eps, int silu, void* stream);
/* nearest-neighbour 2x (F.interpolate(scale_factor=2, mode="nearest")): plain [H W][C] -> interior of a padded
 * [(2H + 2)(2W + 2)][C] image; `out` points at its pixel (1, 1) */
int mgx_upsample2x_pad_nhwc(const uint16_t* x, uint16_t* out, int H, int W, int C, void* stream);
/* latents [Cin][H][W] fp32 -> bf16 interior of a padded NHWC image of C >= Cin channels (`out` at pixel (1, 1)) */
int mgx_latents_to_pad_nhwc(const float* z, uint16_t* out, int Cin, int H, int W, int C, void* stream);
/* first Cout channels of a plain NHWC image -> [Cout][H][W] bf16 */
int mgx_nhwc_to_image(const uint16_t* x, long ld, uint16_t* img, int Cout, int H, int W, void* stream);
/* P[r, :n] = bf16(softmax(scale * S[r, :n])), S fp32 (the mid block's single-head attention over H W <= 16384 tokens) */
int mgx_softmax_rows_f32(const float* S, long lds, uint16_t* P, long ldp, int M, int n, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
