// Per-step solver kernels: Flow-GRPO SDE / Euler ODE step + Gaussian log-prob (fwd, bwd), DanceGRPO step,
// DPM-Solver(++) multistep update.  HBM-bound streaming kernels: 16-byte loads/stores, 8 elements per lane,
// one fused pass per step.  Compiled with -ffp-contract=off: results must be bit-identical to the
// reference's separately-rounded fp32 eager ops (see oracle/solver.py), so no FMA contraction.
//
// Reference: fastvideo/utils/sampling_utils.py:157-210 (flow), :212-253 (dance), :273-639 (dpm).
#include "../../include/mixgrpo_hip.h"
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kVec = 8;
constexpr int kMaxBlocksPerSample = 512;

struct F8 {
  float v[8];
};

__device__ __forceinline__ F8 ld_f32x8(const float* p) {
  F8 r;
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
__device__ __forceinline__ void st_f32x8(float* p, const F8& r) {
  *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
__device__ __forceinline__ F8 ld_bf16x8(const bf16_raw* p) {
  uint4 u = *reinterpret_cast<const uint4*>(p);
  F8 r;
  r.v[0] = __builtin_bit_cast(float, u.x << 16); r.v[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
  r.v[2] = __builtin_bit_cast(float, u.y << 16); r.v[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
  r.v[4] = __builtin_bit_cast(float, u.z << 16); r.v[5] = __builtin_bit_cast(float, u.z & 0xffff0000u);
  r.v[6] = __builtin_bit_cast(float, u.w << 16); r.v[7] = __builtin_bit_cast(float, u.w & 0xffff0000u);
  return r;
}
__device__ __forceinline__ void st_bf16x8(bf16_raw* p, const F8& r) {
  uint4 u;
  u.x = (uint32_t)f2bf(r.v[0]) | ((uint32_t)f2bf(r.v[1]) << 16);
  u.y = (uint32_t)f2bf(r.v[2]) | ((uint32_t)f2bf(r.v[3]) << 16);
  u.z = (uint32_t)f2bf(r.v[4]) | ((uint32_t)f2bf(r.v[5]) << 16);
  u.w = (uint32_t)f2bf(r.v[6]) | ((uint32_t)f2bf(r.v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = u;
}

// block-level sum of per-thread fp32 partials, accumulated in double; thread 0 holds the result
__device__ __forceinline__ double block_sum(float part) {
  __shared__ double red[kThreads / 64];
  double d = wave_sum_d((double)part);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = d;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < kThreads / 64; ++i) t += red[i];
  }
  return t;
}

__global__ void __launch_bounds__(kThreads) logp_finalize_kernel(const double* ws, float* logp, int nblk, long n) {
  const int b = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblk; i += kThreads) acc += ws[(long)b * nblk + i];
  __shared__ double red[kThreads / 64];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < kThreads / 64; ++i) t += red[i];
    logp[b] = (float)(t / (double)n);
  }
}

// ------------------------------------------------------------------------------------------- flow step
template <bool REPLAY, bool DET>
__global__ void __launch_bounds__(kThreads)
flow_fwd_kernel(const float* __restrict__ x, const bf16_raw* __restrict__ v, const bf16_raw* __restrict__ noise,
                const float* __restrict__ prev_in, float* __restrict__ prev_out, float* __restrict__ x0_out,
                float* __restrict__ mean_out, double* __restrict__ ws, long n, mgx_flow_coeffs k) {
  const long base = (long)blockIdx.y * n;
  float part = 0.f;
  for (long i = ((long)blockIdx.x * kThreads + threadIdx.x) * kVec; i < n; i += (long)gridDim.x * kThreads * kVec) {
    const long o = base + i;
    F8 xv = ld_f32x8(x + o), vv = ld_bf16x8(v + o), pv, mv;
    F8 nz;
    if (REPLAY) pv = ld_f32x8(prev_in + o);
    else nz = ld_bf16x8(noise + o);
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      const float t1 = rbf(vv.v[j] * k.c_v);
      const float t2 = rbf(t1 * k.dt_mean);
      const float m = xv.v[j] * k.c_x + t2;
      mv.v[j] = m;
      if (!REPLAY) {
        float p = m + rbf(nz.v[j] * k.sd_noise);
        if (DET) p = xv.v[j] + rbf(vv.v[j] * k.dt_det);
        pv.v[j] = p;
      }
      const float d = pv.v[j] - m;
      float lp = -(d * d);
      lp = lp / k.den;
      lp = lp - k.log_sd;
      lp = lp - k.log_c;
      part += lp;
    }
    if (!REPLAY) st_f32x8(prev_out + o, pv);
    if (mean_out) st_f32x8(mean_out + o, mv);
    if (x0_out) {
      F8 q;
#pragma unroll
      for (int j = 0; j < kVec; ++j) q.v[j] = xv.v[j] - rbf(vv.v[j] * k.sigma_x0);
      st_f32x8(x0_out + o, q);
    }
  }
  const double t = block_sum(part);
  if (threadIdx.x == 0) ws[(long)blockIdx.y * gridDim.x + blockIdx.x] = t;
}

__global__ void __launch_bounds__(kThreads)
flow_bwd_kernel(const float* __restrict__ x, const bf16_raw* __restrict__ v, const float* __restrict__ prev,
                const float* __restrict__ g_logp, bf16_raw* __restrict__ dv, long n, mgx_flow_coeffs k) {
  const long base = (long)blockIdx.y * n;
  // mean backward: grad / n ; then / den (sampling_utils.py:201-208 under autograd)
  const float g = (g_logp[blockIdx.y] / (float)n) / k.den;
  for (long i = ((long)blockIdx.x * kThreads + threadIdx.x) * kVec; i < n; i += (long)gridDim.x * kThreads * kVec) {
    const long o = base + i;
    F8 xv = ld_f32x8(x + o), vv = ld_bf16x8(v + o), pv = ld_f32x8(prev + o), out;
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      const float t1 = rbf(vv.v[j] * k.c_v);
      const float t2 = rbf(t1 * k.dt_mean);
      const float m = xv.v[j] * k.c_x + t2;
      const float d = pv.v[j] - m;
      const float gm = g * (2.f * d);           // d logp / d mean
      const float g2 = rbf(gm);                 // fp32 + bf16 add: grad reaches the bf16 term rounded
      const float g1 = rbf(g2 * k.dt_mean);
      out.v[j] = g1 * k.c_v;                    // rounded to bf16 by the store
    }
    st_bf16x8(dv + o, out);
  }
}

// ------------------------------------------------------------------------------------------- dance step
template <bool REPLAY, bool SDE>
__global__ void __launch_bounds__(kThreads)
dance_fwd_kernel(const float* __restrict__ x, const bf16_raw* __restrict__ v, const float* __restrict__ noise,
                 const float* __restrict__ prev_in, float* __restrict__ prev_out, float* __restrict__ x0_out,
                 double* __restrict__ ws, long n, mgx_dance_coeffs k) {
  const long base = (long)blockIdx.y * n;
  float part = 0.f;
  for (long i = ((long)blockIdx.x * kThreads + threadIdx.x) * kVec; i < n; i += (long)gridDim.x * kThreads * kVec) {
    const long o = base + i;
    F8 xv = ld_f32x8(x + o), vv = ld_bf16x8(v + o), pv, q, nz;
    if (REPLAY) pv = ld_f32x8(prev_in + o);
    else if (SDE) nz = ld_f32x8(noise + o);
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      float m = xv.v[j] + rbf(vv.v[j] * k.ds_r);
      const float x0 = xv.v[j] - rbf(vv.v[j] * k.s_r);
      q.v[j] = x0;
      if (SDE) {
        float sc = xv.v[j] - x0 * k.one_m_s;
        sc = -sc;
        sc = sc / k.s_sq;
        m = m + (k.half_eta2 * sc) * k.ds;
      }
      if (!REPLAY) pv.v[j] = SDE ? (m + nz.v[j] * k.sd) : m;
      const float d = pv.v[j] - m;
      part += (-(d * d)) / k.den;
    }
    if (!REPLAY) st_f32x8(prev_out + o, pv);
    if (x0_out) st_f32x8(x0_out + o, q);
  }
  const double t = block_sum(part);
  if (threadIdx.x == 0) ws[(long)blockIdx.y * gridDim.x + blockIdx.x] = t;
}

template <bool SDE>
__global__ void __launch_bounds__(kThreads)
dance_bwd_kernel(const float* __restrict__ x, const bf16_raw* __restrict__ v, const float* __restrict__ prev,
                 const float* __restrict__ g_logp, bf16_raw* __restrict__ dv, long n, mgx_dance_coeffs k) {
  const long base = (long)blockIdx.y * n;
  const float g = (g_logp[blockIdx.y] / (float)n) / k.den;
  for (long i = ((long)blockIdx.x * kThreads + threadIdx.x) * kVec; i < n; i += (long)gridDim.x * kThreads * kVec) {
    const long o = base + i;
    F8 xv = ld_f32x8(x + o), vv = ld_bf16x8(v + o), pv = ld_f32x8(prev + o), out;
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      float m = xv.v[j] + rbf(vv.v[j] * k.ds_r);
      const float x0 = xv.v[j] - rbf(vv.v[j] * k.s_r);
      if (SDE) {
        float sc = xv.v[j] - x0 * k.one_m_s;
        sc = -sc;
        sc = sc / k.s_sq;
        m = m + (k.half_eta2 * sc) * k.ds;
      }
      const float d = pv.v[j] - m;
      const float gm = g * (2.f * d);
      // path 1: mean0 = x + ds*v  (bf16 product; autograd multiplies by the unrounded scalar)
      float gv = rbf(rbf(gm) * k.ds_b);
      if (SDE) {
        // path 2: correction = (half_eta2*score)*ds, score = -(x - x0*(1-s))/s^2, x0 = x - s*v
        float gs = gm * k.ds;
        gs = gs * k.half_eta2;
        float gn = -(gs / k.s_sq);
        float gx0 = -(gn * k.one_m_s);        // d/dx0 of (x - x0*(1-s)) is -(1-s)
        float gt = rbf(-gx0);                 // x0 = x - T3, T3 bf16
        float gv2 = rbf(gt * k.s_b);
        gv = gv + gv2;                        // autograd accumulates the two bf16 grads
      }
      out.v[j] = gv;
    }
    st_bf16x8(dv + o, out);
  }
}

// ------------------------------------------------------------------------------------------- DPM update
__global__ void __launch_bounds__(kThreads)
dpm_fwd_kernel(const float* __restrict__ s, const bf16_raw* __restrict__ v, const float* __restrict__ m1,
               const float* __restrict__ m2, const float* __restrict__ noise, float* __restrict__ x_out,
               float* __restrict__ x0_out, double* __restrict__ ws, long n, mgx_dpm_coeffs k) {
  const long base = (long)blockIdx.y * n;
  float part = 0.f;
  for (long i = ((long)blockIdx.x * kThreads + threadIdx.x) * kVec; i < n; i += (long)gridDim.x * kThreads * kVec) {
    const long o = base + i;
    F8 sv = ld_f32x8(s + o), vv = ld_bf16x8(v + o), a1, a2, nz, xo, q;
    if (k.order >= 2) a1 = ld_f32x8(m1 + o);
    if (k.order >= 3) a2 = ld_f32x8(m2 + o);
    if (k.sde) nz = ld_f32x8(noise + o);
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      const float m0 = sv.v[j] - rbf(vv.v[j] * k.sigma_x0);
      q.v[j] = m0;
      float D1 = 0.f, D2 = 0.f;
      if (k.order >= 2) D1 = k.inv_r0 * (m0 - a1.v[j]);
      if (k.order >= 3) {
        const float D10 = D1;
        const float D11 = k.inv_r1 * (a1.v[j] - a2.v[j]);
        D1 = D10 + k.c_r * (D10 - D11);
        D2 = k.inv_r01 * (D10 - D11);
      }
      float mean = k.cm[0] * sv.v[j] + k.cm[1] * m0;
      if (k.order >= 2) mean = mean + k.cm[2] * D1;
      if (k.order >= 3) mean = mean + k.cm[3] * D2;
      float xt;
      if (k.sde) {
        xt = mean + k.sd_noise * nz.v[j];
      } else {
        xt = k.cx[0] * sv.v[j] + k.cx[1] * m0;
        if (k.order >= 2) xt = xt + k.cx[2] * D1;
        if (k.order >= 3) xt = xt + k.cx[3] * D2;
      }
      xo.v[j] = xt;
      const float d = xt - mean;
      float lp = -(d * d);
      lp = lp / k.den;
      lp = lp - k.log_sd;
      lp = lp - k.log_c;
      part += lp;
    }
    st_f32x8(x_out + o, xo);
    if (x0_out) st_f32x8(x0_out + o, q);
  }
  const double t = block_sum(part);
  if (threadIdx.x == 0) ws[(long)blockIdx.y * gridDim.x + blockIdx.x] = t;
}

// d logp / d v of a first-order DPM-Solver(++) SDE step whose sample x_t is held fixed (prev_sample.detach(), reference
// sampling_utils.py:376-383): logp = mean(-(x_t - mean)^2 / den - ...), mean = cm0 * s + cm1 * x0, x0 = s - bf16(sigma * v).
// This is the training replay under dpm_apply_strategy="all" (train_grpo_flux.py:170-180: dpm_state=None -> first order).
__global__ void __launch_bounds__(kThreads)
dpm_bwd_kernel(const float* __restrict__ s, const bf16_raw* __restrict__ v, const float* __restrict__ xt,
               const float* __restrict__ g_logp, bf16_raw* __restrict__ dv, long n, mgx_dpm_coeffs k, float sigma_b) {
  const long base = (long)blockIdx.y * n;
  const float g = (g_logp[blockIdx.y] / (float)n) / k.den;
  for (long i = ((long)blockIdx.x * kThreads + threadIdx.x) * kVec; i < n; i += (long)gridDim.x * kThreads * kVec) {
    const long o = base + i;
    F8 sv = ld_f32x8(s + o), vv = ld_bf16x8(v + o), xv = ld_f32x8(xt + o), out;
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      const float m0 = sv.v[j] - rbf(vv.v[j] * k.sigma_x0);
      const float mean = k.cm[0] * sv.v[j] + k.cm[1] * m0;
      const float d = xv.v[j] - mean;
      const float gm = g * (2.f * d);           // d logp / d mean
      const float gx0 = gm * k.cm[1];           // mean = cm0 * s + cm1 * x0
      const float gt = rbf(-gx0);               // x0 = s - T, T = sigma * v in bf16: the gradient reaches T rounded
      out.v[j] = gt * sigma_b;                  // rounded to bf16 by the store
    }
    st_bf16x8(dv + o, out);
  }
}

__global__ void __launch_bounds__(kThreads)
x0_pred_kernel(const float* __restrict__ s, const bf16_raw* __restrict__ v, float* __restrict__ out, long total,
               float sigma_x0) {
  for (long i = ((long)blockIdx.x * kThreads + threadIdx.x) * kVec; i < total; i += (long)gridDim.x * kThreads * kVec) {
    F8 sv = ld_f32x8(s + i), vv = ld_bf16x8(v + i), q;
#pragma unroll
    for (int j = 0; j < kVec; ++j) q.v[j] = sv.v[j] - rbf(vv.v[j] * sigma_x0);
    st_f32x8(out + i, q);
  }
}

// ------------------------------------------------------------------------------------------- pack / unpack
// out[b, (h2*W2 + w2), c*4 + dh*2 + dw] = in[b, c, 2*h2+dh, 2*w2+dw]   (train_grpo_flux.py:94-99)
template <typename T, bool UNPACK>
__global__ void pack_kernel(const T* __restrict__ in, T* __restrict__ out, int C, int H, int W, long total) {
  const int W2 = W / 2, H2 = H / 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int ch = (int)(r % (4 * C)); r /= 4 * C;
    const int w2 = (int)(r % W2); r /= W2;
    const int h2 = (int)(r % H2); r /= H2;
    const long b = r;
    const int c = ch >> 2, dh = (ch >> 1) & 1, dw = ch & 1;
    const long img = ((b * C + c) * H + (2 * h2 + dh)) * (long)W + (2 * w2 + dw);
    if (UNPACK) out[img] = in[idx];
    else out[idx] = in[img];
  }
}

int blocks_per_sample(long n) {
  long b = (n + (long)kThreads * kVec - 1) / ((long)kThreads * kVec);
  return (int)(b < 1 ? 1 : (b > kMaxBlocksPerSample ? kMaxBlocksPerSample : b));
}

int check_common(int B, long n) {
  MGX_REQUIRE(B > 0 && n > 0, "empty problem");
  MGX_REQUIRE(n % kVec == 0, "elements per sample must be a multiple of 8");
  return MGX_OK;
}

}  // namespace

extern "C" long mgx_logp_workspace_elems(int B, long n) { return (long)B * blocks_per_sample(n); }

extern "C" int mgx_flow_step_fwd(const float* x, const uint16_t* v, const uint16_t* noise, const float* prev_in,
                                 float* prev_out, float* x0_out, float* mean_out, float* logp, double* ws, int B,
                                 long n, const mgx_flow_coeffs* k, int deterministic, void* stream) {
  if (int e = check_common(B, n)) return e;
  MGX_REQUIRE(x && v && logp && ws && k, "null argument");
  MGX_REQUIRE((noise != nullptr) != (prev_in != nullptr), "exactly one of noise / prev_in must be given");
  MGX_REQUIRE(prev_in || prev_out, "rollout mode needs prev_out");
  const int nb = blocks_per_sample(n);
  dim3 grid(nb, B);
  hipStream_t st = (hipStream_t)stream;
  if (prev_in)
    flow_fwd_kernel<true, false><<<grid, kThreads, 0, st>>>(x, v, nullptr, prev_in, nullptr, x0_out, mean_out, ws, n, *k);
  else if (deterministic)
    flow_fwd_kernel<false, true><<<grid, kThreads, 0, st>>>(x, v, noise, nullptr, prev_out, x0_out, mean_out, ws, n, *k);
  else
    flow_fwd_kernel<false, false><<<grid, kThreads, 0, st>>>(x, v, noise, nullptr, prev_out, x0_out, mean_out, ws, n, *k);
  logp_finalize_kernel<<<B, kThreads, 0, st>>>(ws, logp, nb, n);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_flow_step_bwd(const float* x, const uint16_t* v, const float* prev, const float* g_logp,
                                 uint16_t* dv, int B, long n, const mgx_flow_coeffs* k, void* stream) {
  if (int e = check_common(B, n)) return e;
  MGX_REQUIRE(x && v && prev && g_logp && dv && k, "null argument");
  flow_bwd_kernel<<<dim3(blocks_per_sample(n), B), kThreads, 0, (hipStream_t)stream>>>(x, v, prev, g_logp, dv, n, *k);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_dance_step_fwd(const float* x, const uint16_t* v, const float* noise, const float* prev_in,
                                  float* prev_out, float* x0_out, float* logp, double* ws, int B, long n,
                                  const mgx_dance_coeffs* k, int sde, void* stream) {
  if (int e = check_common(B, n)) return e;
  MGX_REQUIRE(x && v && logp && ws && k, "null argument");
  MGX_REQUIRE(prev_in || prev_out, "rollout mode needs prev_out");
  MGX_REQUIRE(prev_in || !sde || noise, "SDE rollout needs noise");
  const int nb = blocks_per_sample(n);
  dim3 grid(nb, B);
  hipStream_t st = (hipStream_t)stream;
  if (prev_in && sde) dance_fwd_kernel<true, true><<<grid, kThreads, 0, st>>>(x, v, nullptr, prev_in, nullptr, x0_out, ws, n, *k);
  else if (prev_in) dance_fwd_kernel<true, false><<<grid, kThreads, 0, st>>>(x, v, nullptr, prev_in, nullptr, x0_out, ws, n, *k);
  else if (sde) dance_fwd_kernel<false, true><<<grid, kThreads, 0, st>>>(x, v, noise, nullptr, prev_out, x0_out, ws, n, *k);
  else dance_fwd_kernel<false, false><<<grid, kThreads, 0, st>>>(x, v, nullptr, nullptr, prev_out, x0_out, ws, n, *k);
  logp_finalize_kernel<<<B, kThreads, 0, st>>>(ws, logp, nb, n);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_dance_step_bwd(const float* x, const uint16_t* v, const float* prev, const float* g_logp,
                                  uint16_t* dv, int B, long n, const mgx_dance_coeffs* k, int sde, void* stream) {
  if (int e = check_common(B, n)) return e;
  MGX_REQUIRE(x && v && prev && g_logp && dv && k, "null argument");
  dim3 grid(blocks_per_sample(n), B);
  if (sde) dance_bwd_kernel<true><<<grid, kThreads, 0, (hipStream_t)stream>>>(x, v, prev, g_logp, dv, n, *k);
  else dance_bwd_kernel<false><<<grid, kThreads, 0, (hipStream_t)stream>>>(x, v, prev, g_logp, dv, n, *k);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_dpm_step_fwd(const float* sample, const uint16_t* v, const float* m1, const float* m2,
                                const float* noise, float* x_out, float* x0_out, float* logp, double* ws, int B,
                                long n, const mgx_dpm_coeffs* k, void* stream) {
  if (int e = check_common(B, n)) return e;
  MGX_REQUIRE(sample && v && x_out && logp && ws && k, "null argument");
  MGX_REQUIRE(k->order >= 1 && k->order <= 3, "order must be 1..3");
  MGX_REQUIRE(k->order < 2 || m1, "order>=2 needs m1");
  MGX_REQUIRE(k->order < 3 || m2, "order 3 needs m2");
  MGX_REQUIRE(!k->sde || noise, "SDE update needs noise");
  const int nb = blocks_per_sample(n);
  hipStream_t st = (hipStream_t)stream;
  dpm_fwd_kernel<<<dim3(nb, B), kThreads, 0, st>>>(sample, v, m1, m2, noise, x_out, x0_out, ws, n, *k);
  logp_finalize_kernel<<<B, kThreads, 0, st>>>(ws, logp, nb, n);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_dpm_step_bwd(const float* sample, const uint16_t* v, const float* x_t, const float* g_logp, uint16_t* dv,
                                int B, long n, const mgx_dpm_coeffs* k, float sigma_b, void* stream) {
  if (int e = check_common(B, n)) return e;
  MGX_REQUIRE(sample && v && x_t && g_logp && dv && k, "null argument");
  MGX_REQUIRE(k->order == 1, "the replayed DPM step has no multistep history: first order only");
  dpm_bwd_kernel<<<dim3(blocks_per_sample(n), B), kThreads, 0, (hipStream_t)stream>>>(sample, v, x_t, g_logp, dv, n, *k, sigma_b);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_x0_pred(const float* sample, const uint16_t* v, float* x0_out, long total, float sigma_x0,
                           void* stream) {
  MGX_REQUIRE(sample && v && x0_out && total > 0 && total % kVec == 0, "bad argument");
  int nb = cdiv(total, (long)kThreads * kVec);
  if (nb > 4096) nb = 4096;
  x0_pred_kernel<<<nb, kThreads, 0, (hipStream_t)stream>>>(sample, v, x0_out, total, sigma_x0);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

template <bool UNPACK>
static int pack_impl(const void* in, void* out, int B, int C, int H, int W, int elem_size, void* stream) {
  MGX_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0, "bad argument");
  MGX_REQUIRE(H % 2 == 0 && W % 2 == 0, "latent height/width must be even");
  MGX_REQUIRE(elem_size == 2 || elem_size == 4, "elem_size must be 2 or 4");
  const long total = (long)B * C * H * W;
  int nb = cdiv(total, 256);
  if (nb > 8192) nb = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (elem_size == 2) pack_kernel<uint16_t, UNPACK><<<nb, 256, 0, st>>>((const uint16_t*)in, (uint16_t*)out, C, H, W, total);
  else pack_kernel<uint32_t, UNPACK><<<nb, 256, 0, st>>>((const uint32_t*)in, (uint32_t*)out, C, H, W, total);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
extern "C" int mgx_pack_latents(const void* in, void* out, int B, int C, int H, int W, int es, void* stream) {
  return pack_impl<false>(in, out, B, C, H, W, es, stream);
}
extern "C" int mgx_unpack_latents(const void* in, void* out, int B, int C, int H, int W, int es, void* stream) {
  return pack_impl<true>(in, out, B, C, H, W, es, stream);
}
