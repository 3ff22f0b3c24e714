// Group-relative advantage and PPO-clip GRPO loss on tiny [G]/[B] vectors: device-resident so the train
// step has no host round trip between the rollout, the replay forward and the backward seed.
// Reference: fastvideo/train_grpo_flux.py:440-501 (advantages), :560-583 (loss).
#include "../../include/mixgrpo_hip.h"
#include "common.h"

namespace {

constexpr int kMaxGroup = 1024;

// one block per group; thread 0 does the (<=1024-element) statistics serially in the reference's order
__global__ void group_adv_kernel(const float* __restrict__ r, float* __restrict__ out, int G, float trimmed_ratio,
                                 float weight, int accumulate) {
  __shared__ float srt[kMaxGroup];
  __shared__ float mean_s, std_s;
  const float* g = r + (long)blockIdx.x * G;
  for (int i = threadIdx.x; i < G; i += blockDim.x) {
    // rank sort (stable): position = #elements smaller (+ equal ones before it)
    const float x = g[i];
    int pos = 0;
    for (int j = 0; j < G; ++j) {
      const float y = g[j];
      pos += (y < x) || (y == x && j < i);
    }
    srt[pos] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int lo = 0, cnt = G;
    const float* src = g;  // untrimmed statistics run over the group in its own order
    if (trimmed_ratio > 0.f) {
      int t = (int)((float)G * trimmed_ratio);
      if (t > G - 1) t = G - 1;
      lo = t;
      cnt = G - t;
      src = srt;
    }
    float s = 0.f;
    for (int i = 0; i < cnt; ++i) s += src[lo + i];
    const float mean = s / (float)cnt;
    float q = 0.f;
    for (int i = 0; i < cnt; ++i) {
      const float d = src[lo + i] - mean;
      q += d * d;
    }
    mean_s = mean;
    std_s = sqrtf(q / (float)(cnt - 1)) + 1e-8f;   // unbiased; cnt==1 -> NaN like torch.std
  }
  __syncthreads();
  for (int i = threadIdx.x; i < G; i += blockDim.x) {
    const float a = (g[i] - mean_s) / std_s;
    const long o = (long)blockIdx.x * G + i;
    out[o] = accumulate ? out[o] + a * weight : a * weight;
  }
}

__global__ void global_adv_kernel(const float* __restrict__ r, const float* __restrict__ all, float* __restrict__ out,
                                  int n, int n_all) {
  __shared__ float mean_s, std_s;
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n_all; ++i) s += all[i];
    const float mean = s / (float)n_all;
    float q = 0.f;
    for (int i = 0; i < n_all; ++i) {
      const float d = all[i] - mean;
      q += d * d;
    }
    mean_s = mean;
    std_s = sqrtf(q / (float)(n_all - 1)) + 1e-8f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = (r[i] - mean_s) / std_s;
}

// torch.clamp / torch.maximum propagate NaN (fminf / fmaxf return the other operand): a NaN advantage (use_group=False with
// one gathered reward: unbiased std of one element) or a NaN log-prob must reach the loss as NaN, as in the reference
__device__ __forceinline__ float clamp_nan(float x, float lo, float hi) { return x != x ? x : fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ float max_nan(float a, float b) { return (a != a || b != b) ? (a + b) : fmaxf(a, b); }

__global__ void grpo_loss_kernel(const float* __restrict__ nlp, const float* __restrict__ olp,
                                 const float* __restrict__ adv, int B, float clip_range, float adv_clip_max,
                                 float kl_coeff, float denom, float* __restrict__ loss, float* __restrict__ policy,
                                 float* __restrict__ kl, float* __restrict__ clip_frac, float* __restrict__ g_logp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float a = clamp_nan(adv[b], -adv_clip_max, adv_clip_max);
  const float diff = nlp[b] - olp[b];
  const float ratio = expf(diff);
  const float lo = 1.0f - clip_range, hi = 1.0f + clip_range;
  const float rc = clamp_nan(ratio, lo, hi);
  const float unclipped = -a * ratio;
  const float clipped = -a * rc;
  const float pol = max_nan(unclipped, clipped) / denom;
  const float klv = 0.5f * (diff * diff) / denom;
  policy[b] = pol;
  kl[b] = klv;
  loss[b] = pol + kl_coeff * klv;
  clip_frac[b] = fabsf(ratio - 1.0f) > clip_range ? 1.f : 0.f;
  // d loss / d new_logp.  torch.maximum splits the gradient on ties; clamp passes it inside [lo, hi].
  const float gmax = 1.0f / denom;
  float w_un, w_cl;
  if (unclipped > clipped) { w_un = 1.f; w_cl = 0.f; }
  else if (unclipped < clipped) { w_un = 0.f; w_cl = 1.f; }
  else { w_un = 0.5f; w_cl = 0.5f; }
  float g_ratio = (gmax * w_un) * (-a);
  if (ratio >= lo && ratio <= hi) g_ratio += (gmax * w_cl) * (-a);
  float g = g_ratio * ratio;
  g += ((kl_coeff / denom) * 0.5f) * (2.f * diff);
  g_logp[b] = g;
}

}  // namespace

extern "C" int mgx_group_advantage(const float* rewards, float* out, int n, int G, float trimmed_ratio, float weight,
                                   int accumulate, void* stream) {
  MGX_REQUIRE(rewards && out, "null argument");
  MGX_REQUIRE(G > 0 && G <= kMaxGroup, "group size must be in 1..1024");
  MGX_REQUIRE(n >= 0, "negative length");
  const int groups = n / G;   // a ragged tail is left untouched, as in the reference's range(n // G)
  if (groups == 0) return MGX_OK;
  group_adv_kernel<<<groups, 64, 0, (hipStream_t)stream>>>(rewards, out, G, trimmed_ratio, weight, accumulate);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_global_advantage(const float* rewards, const float* gathered, float* out, int n, int n_all,
                                    void* stream) {
  MGX_REQUIRE(rewards && gathered && out && n > 0 && n_all > 0, "bad argument");
  global_adv_kernel<<<1, 64, 0, (hipStream_t)stream>>>(rewards, gathered, out, n, n_all);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_grpo_loss(const float* new_logp, const float* old_logp, const float* adv, int B, float clip_range,
                             float adv_clip_max, float kl_coeff, float denom, float* loss, float* policy, float* kl,
                             float* clip_frac, float* g_logp, void* stream) {
  MGX_REQUIRE(new_logp && old_logp && adv && loss && policy && kl && clip_frac && g_logp, "null argument");
  MGX_REQUIRE(B > 0 && denom > 0.f, "bad size");
  grpo_loss_kernel<<<cdiv(B, 64), 64, 0, (hipStream_t)stream>>>(new_logp, old_logp, adv, B, clip_range, adv_clip_max,
                                                                 kl_coeff, denom, loss, policy, kl, clip_frac, g_logp);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
