// Fused AdamW over the flat parameter store + global grad-norm clipping (HBM-bound, one pass each).
// Replaces torch.optim.AdamW.step (reference fastvideo/train_grpo_flux.py:715-721, :607) and
// FSDP.clip_grad_norm_ (:606).  fp32 master weights, fp32 grads, fp32 moments; the bf16 compute copy of the
// weights is rewritten in the same pass.
#include "../../include/mixgrpo_hip.h"
#include "common.h"

#include <cmath>

namespace {

__global__ void __launch_bounds__(256) sqnorm_kernel(const float* __restrict__ g, long n, double* __restrict__ part) {
  float acc = 0.f;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 4 <= n) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (long j = i; j < n; ++j) acc += g[j] * g[j];
    }
  }
  __shared__ double red[4];
  double d = wave_sum_d((double)acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void sqnorm_finish_kernel(const double* __restrict__ part, int nb, float* __restrict__ out, float beta) {
  __shared__ double red[4];
  double a = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) a += part[i];
  a = wave_sum_d(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = beta * out[0] + (float)(red[0] + red[1] + red[2] + red[3]);
}

// Scalar coefficients exactly as torch.optim.AdamW's (foreach) step forms them: in Python DOUBLE precision on the host, then
// one rounding to fp32 where the elementwise op takes them (1 - beta2 = 0.0010000000000000009 -> 0.001f, NOT 1.0f - 0.999f =
// 0.00099998713, which is 1.3e-5 off in the second moment).
struct AdamArgs {
  float decay;       // 1 - lr * weight_decay             (_foreach_mul_(params, ...))
  float omb1;        // 1 - beta1                         (_foreach_lerp_(exp_avgs, grads, ...))
  float beta2, omb2; // beta2, 1 - beta2                  (_foreach_mul_, _foreach_addcmul_)
  float bc2_sqrt;    // sqrt(1 - beta2^step)              (_foreach_div_(sqrt(exp_avg_sq), ...))
  float eps;
  float step_size;   // lr / (1 - beta1^step)             (_foreach_addcdiv_(params, exp_avgs, denom, -step_size))
  float max_norm;
};

__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ w, bf16_raw* __restrict__ w16,
                                                    const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, AdamArgs a,
                                                    const float* __restrict__ gnorm_sq, float grad_scale) {
  float clip = grad_scale;
  if (gnorm_sq) {
    const float total = sqrtf(gnorm_sq[0]) * grad_scale;
    const float c = a.max_norm / (total + 1e-6f);
    clip *= c < 1.0f ? c : 1.0f;
  }
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    float4 pw = *reinterpret_cast<float4*>(w + i);
    const float4 pg = *reinterpret_cast<const float4*>(g + i);
    float4 pm = *reinterpret_cast<float4*>(m + i);
    float4 pv = *reinterpret_cast<float4*>(v + i);
    float* fw = reinterpret_cast<float*>(&pw);
    const float* fg = reinterpret_cast<const float*>(&pg);
    float* fm = reinterpret_cast<float*>(&pm);
    float* fv = reinterpret_cast<float*>(&pv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = fg[j] * clip;
      fw[j] = fw[j] * a.decay;
      fm[j] = fm[j] + a.omb1 * (gr - fm[j]);                    // lerp, weight < 0.5
      fv[j] = fv[j] * a.beta2;
      fv[j] = fv[j] + a.omb2 * gr * gr;
      const float denom = sqrtf(fv[j]) / a.bc2_sqrt + a.eps;
      fw[j] = fw[j] - a.step_size * (fm[j] / denom);
    }
    *reinterpret_cast<float4*>(w + i) = pw;
    *reinterpret_cast<float4*>(m + i) = pm;
    *reinterpret_cast<float4*>(v + i) = pv;
    uint2 o;
    o.x = (uint32_t)f2bf(fw[0]) | ((uint32_t)f2bf(fw[1]) << 16);
    o.y = (uint32_t)f2bf(fw[2]) | ((uint32_t)f2bf(fw[3]) << 16);
    *reinterpret_cast<uint2*>(w16 + i) = o;
  }
}

__global__ void scale_f32_kernel(float* __restrict__ x, long n, float s) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    float4 v = *reinterpret_cast<float4*>(x + i);
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    *reinterpret_cast<float4*>(x + i) = v;
  }
}

}  // namespace

extern "C" long mgx_sqnorm_workspace(void) { return 4096; }

extern "C" int mgx_sqnorm_f32(const float* g, long n, double* ws, float* out, float beta, void* stream) {
  MGX_REQUIRE(g && ws && out && n > 0, "bad argument");
  MGX_REQUIRE((uintptr_t)g % 16 == 0, "gradient buffer must be 16-byte aligned");
  int nb = cdiv(n, 1024 * 8);
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  hipStream_t st = (hipStream_t)stream;
  sqnorm_kernel<<<nb, 256, 0, st>>>(g, n, ws);
  sqnorm_finish_kernel<<<1, 256, 0, st>>>(ws, nb, out, beta);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_adamw_step(float* w, uint16_t* w16, const float* g, float* m, float* v, long n, double lr, double beta1,
                              double beta2, double eps, double weight_decay, int step, const float* gnorm_sq,
                              float max_norm, float grad_scale, void* stream) {
  MGX_REQUIRE(w && w16 && g && m && v && n > 0 && step >= 1, "bad argument");
  MGX_REQUIRE(n % 4 == 0, "parameter count must be padded to a multiple of 4");
  AdamArgs a;
  a.decay = (float)(1.0 - lr * weight_decay);
  a.omb1 = (float)(1.0 - beta1);
  a.beta2 = (float)beta2;
  a.omb2 = (float)(1.0 - beta2);
  a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
  a.eps = (float)eps;
  a.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
  a.max_norm = max_norm;
  int nb = cdiv(n, 1024 * 4);
  if (nb > 8192) nb = 8192;
  adamw_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(w, w16, g, m, v, n, a, gnorm_sq, grad_scale);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_scale_f32(float* x, long n, float s, void* stream) {
  MGX_REQUIRE(x && n > 0 && n % 4 == 0, "bad argument");
  int nb = cdiv(n, 1024 * 4);
  if (nb > 8192) nb = 8192;
  scale_f32_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(x, n, s);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
