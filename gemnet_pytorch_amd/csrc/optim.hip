// Trainer-step fusion: what Trainer.train_on_batch does after loss.backward()
// (gemnet/training/trainer.py:250-278 shared-gradient rescale, :353-356 global-norm clip, :115-160,:358 AdamW/Adam
// with amsgrad and eps 1e-7, gemnet/training/ema_decay.py:68-93 EMA update) — ~100 small launches over ~60
// parameter tensors in the reference — as TWO launches over one flat fp32 buffer (the same buffer the gradient
// all-reduce uses):
//   sqnorm_kernel   partial sums of (g * gscale)^2 per block (deterministic two-stage reduction, no atomics)
//   adamw_ema_kernel every block folds the partials (fixed order) into the clip coefficient, then one pass:
//                    p, m, v, vmax, ema updated in place.
// Per-element vectors `gscale` (1/num_blocks for the shared basis projections, 1 elsewhere) and `wd` (weight decay
// of the AdamW group, 0 for embeddings / frequencies / biases) encode the parameter groups.
#include "common.h"

namespace {

constexpr int OPT_NT = 256;

__global__ __launch_bounds__(OPT_NT) void sqnorm_kernel(const float* __restrict__ g, const float* __restrict__ gscale,
                                                       int64_t n, double* __restrict__ partial) {
  __shared__ double sh[OPT_NT / 64];
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)OPT_NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * OPT_NT) {
    const double v = (double)g[i] * (double)gscale[i];
    acc += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < OPT_NT / 64; ++w) s += sh[w];
    partial[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(OPT_NT) void adamw_ema_kernel(
    float* __restrict__ p, const float* __restrict__ g, const float* __restrict__ gscale, const float* __restrict__ wd,
    float* __restrict__ m, float* __restrict__ v, float* __restrict__ vmax, float* __restrict__ ema, int64_t n,
    const double* __restrict__ partial, int n_partial, float max_norm, float lr, float beta1, float beta2, float eps,
    float bias1, float bias2_sqrt, float ema_decay, float* __restrict__ norm_out, int32_t* __restrict__ flag, int flag_bit) {
  __shared__ float clip_s;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n_partial; ++i) s += partial[i];   // same fixed order in every block
    const float norm = (float)sqrt(s);
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    const float coef = max_norm / (norm + 1e-6f);
    // a non-finite gradient norm with `flag` given: the step is SKIPPED (clip < 0 marks it) and reported
    const bool bad = flag != nullptr && !(fabsf(norm) <= 3.4028234e38f);
    clip_s = bad ? -1.0f : (coef < 1.0f ? coef : 1.0f);
    if (blockIdx.x == 0 && norm_out) *norm_out = norm;
    // bits 0-7 of the word: the flags; bits 8+: the number of skipped steps since the last reset (the host corrects Adam's
    // bias-correction counter by exactly that many: a poll may come one or two calls late and cover several skipped steps)
    if (blockIdx.x == 0 && bad) { atomicOr(flag, flag_bit); atomicAdd(flag, 256); }
  }
  __syncthreads();
  const float clip = clip_s;
  if (clip < 0.0f) return;      // (every block decides from the same partials: all or none)
  const float step = lr / bias1;
  const float omd = 1.0f - ema_decay;
  for (int64_t i = blockIdx.x * (int64_t)OPT_NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * OPT_NT) {
    const float gi = g[i] * gscale[i] * clip;
    float pi = p[i];
    pi *= 1.0f - lr * wd[i];                               // decoupled weight decay (0 in the Adam group)
    const float mi = m[i] + (1.0f - beta1) * (gi - m[i]);   // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    const float vm = fmaxf(vmax[i], vi);                    // amsgrad
    const float denom = sqrtf(vm) / bias2_sqrt + eps;
    pi -= step * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    vmax[i] = vm;
    if (ema) ema[i] -= omd * (ema[i] - pi);
  }
}

// flag[0] |= bit when x[0 .. n) holds a non-finite value: the range check of a replayed hipGraph (no host read-back inside
// the graph; the word is sticky until the host clears it)
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float* __restrict__ x, int64_t n, int32_t* __restrict__ flag,
                                                            int bit) {
  bool bad = false;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    bad = bad || !(fabsf(x[i]) <= 3.4028234e38f);      // inf and nan both fail the comparison
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, bit);   // (a constant bit: order-independent)
}

}  // namespace

extern "C" int gn_nonfinite_flag_f32(const float* x, int64_t n, int32_t* flag, int bit, void* stream) {
  if (n <= 0) return 0;
  if (!flag || !bit) return (int)hipErrorInvalidValue;
  int64_t nb = (n + 255) / 256;
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(nonfinite_flag_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, flag, bit);
  GN_LAUNCH_CHECK();
  return 0;
}

// ---- the training loss and its cotangents in one launch ---------------------------------------------------------------------
// loss = w_e sum |E - Et| + w_f sum_a m_a |F_a - Ft_a|_2   (trainer.py:330-343: (1 - rho) MAE(E) + rho mean L2(F), the weights
// carry rho and the global counts); gE = w_e sign(E - Et), gF_a = w_f m_a (F_a - Ft_a) / |F_a - Ft_a| (0 where the norm is 0,
// as ATen's norm backward).  One workgroup, fixed order of addition: bit-reproducible.  The composite form was 16 ATen launches
// in the forward of the loss and as many in its backward, each a 4 us node of the captured step.
__global__ __launch_bounds__(1024) void force_loss_kernel(const float* __restrict__ E, const float* __restrict__ Et, int64_t nE,
                                                          const float* __restrict__ F, const float* __restrict__ Ft, int64_t A,
                                                          const float* __restrict__ mask, float w_e, float w_f,
                                                          const float* __restrict__ w_f_dev, float* __restrict__ loss,
                                                          float* __restrict__ gE, float* __restrict__ gF) {
  __shared__ double red[1024];
  const int tid = threadIdx.x;
  if (w_f_dev) w_f *= *w_f_dev;
  double acc = 0.0;
  for (int64_t i = tid; i < nE; i += 1024) {
    const float d = E[i] - Et[i];
    acc += (double)w_e * fabsf(d);
    gE[i] = d > 0.f ? w_e : (d < 0.f ? -w_e : 0.f);
  }
  for (int64_t a = tid; a < A; a += 1024) {
    const float m = mask ? mask[a] : 1.f;
    const float dx = F[3 * a] - Ft[3 * a], dy = F[3 * a + 1] - Ft[3 * a + 1], dz = F[3 * a + 2] - Ft[3 * a + 2];
    const float n = sqrtf(dx * dx + dy * dy + dz * dz);
    const float s = (m != 0.f && n > 0.f) ? w_f * m / n : 0.f;
    if (m != 0.f) acc += (double)w_f * m * n;
    gF[3 * a] = s * dx; gF[3 * a + 1] = s * dy; gF[3 * a + 2] = s * dz;
  }
  red[tid] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) *loss = (float)red[0];
}

extern "C" int gn_force_loss_f32(const float* E, const float* Et, int64_t nE, const float* F, const float* Ft, int64_t A,
                                 const float* mask, float w_e, float w_f, const float* w_f_dev, float* loss, float* gE, float* gF,
                                 void* stream) {
  if (nE < 0 || A < 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(force_loss_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), E, Et, nE, F, Ft, A, mask, w_e,
                     w_f, w_f_dev, loss, gE, gF);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_optim_blocks(int64_t n) {
  int64_t b = (n + OPT_NT * 8 - 1) / (OPT_NT * 8);
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" int gn_adamw_ema_step_f32(float* p, const float* g, const float* gscale, const float* wd, float* m, float* v,
                                     float* vmax, float* ema, int64_t n, double* partial, float max_norm, float lr,
                                     float beta1, float beta2, float eps, int step, float ema_decay, float* norm_out,
                                     int32_t* flag, int flag_bit, void* stream) {
  if (n <= 0) return 0;
  if (step < 1 || !partial) return (int)hipErrorInvalidValue;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nb = gn_optim_blocks(n);
  hipLaunchKernelGGL(sqnorm_kernel, dim3(nb), dim3(OPT_NT), 0, st, g, gscale, n, partial);
  GN_LAUNCH_CHECK();
  const float bias1 = 1.0f - powf(beta1, (float)step);
  const float bias2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(nb), dim3(OPT_NT), 0, st, p, g, gscale, wd, m, v, vmax, ema, n, partial, nb,
                     max_norm, lr, beta1, beta2, eps, bias1, bias2_sqrt, ema_decay, norm_out, flag, flag_bit);
  GN_LAUNCH_CHECK();
  return 0;
}
