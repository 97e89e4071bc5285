// Row gather and deterministic CSR segmented sum (mutual adjoints).
//
// Replaces the ATen `index` gathers of the reference (interaction_block.py:543,548,562,678,693;
// embedding_block.py:70-71) and torch_scatter.scatter(reduce="add")
// (atom_update_block.py:67,172; gemnet.py:580) — the latter without atomics: rows are grouped
// by destination through a CSR (offsets + optional permutation) built once per batch.
#include "common.h"

namespace {

__global__ void gather_rows_v4(const float4* __restrict__ x, const int32_t* __restrict__ idx,
                               float4* __restrict__ y, int64_t T, int C4) {
  const int64_t n = T * C4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / C4;
    const int c = (int)(i - t * C4);
    y[i] = x[(int64_t)idx[t] * C4 + c];
  }
}

// y[t] = scale * x[idx[t]] (.) m[t]  (row gather fused with a Hadamard product: the operand of the weight gradient of
// the radial-weighted aggregation, q[e] = scale * g[id_a[e]] (.) m[e])
__global__ void gather_mul_v4(const float4* __restrict__ x, const int32_t* __restrict__ idx, const float4* __restrict__ m,
                              float4* __restrict__ y, int64_t T, int C4, float scale) {
  const int64_t n = T * C4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / C4;
    const int c = (int)(i - t * C4);
    const float4 a = x[(int64_t)idx[t] * C4 + c], b = m[i];
    y[i] = make_float4(scale * a.x * b.x, scale * a.y * b.y, scale * a.z * b.z, scale * a.w * b.w);
  }
}

__global__ void gather_rows_s(const float* __restrict__ x, const int32_t* __restrict__ idx,
                              float* __restrict__ y, int64_t T, int C) {
  const int64_t n = T * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / C;
    const int c = (int)(i - t * C);
    y[i] = x[(int64_t)idx[t] * C + c];
  }
}

// One wave per output row.  A row's CW lanes-per-entry (C/4 float4 lanes, or C scalar lanes) are
// replicated EPL = 64/CW times so EPL segment entries are accumulated concurrently; the EPL partial
// sums are combined through LDS in a fixed order (deterministic).  Long segments (the (T,3) -> (A,3)
// position gradients: ~1000 entries per atom) no longer serialise on one lane.
template <typename VT>
__device__ __forceinline__ VT vzero();
template <> __device__ __forceinline__ float vzero<float>() { return 0.f; }
template <> __device__ __forceinline__ float4 vzero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void vadd(float& a, const float& b) { a += b; }
__device__ __forceinline__ void vadd(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

template <typename VT>
__global__ __launch_bounds__(256) void segsum_wave(const VT* __restrict__ y, const int32_t* __restrict__ perm,
                                                   const int32_t* __restrict__ seg_off, VT* __restrict__ x,
                                                   int64_t N, int CW) {
  __shared__ VT part[4][64];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  const bool row_ok = r < N;
  const int k0 = row_ok ? seg_off[r] : 0, k1 = row_ok ? seg_off[r + 1] : 0;
  for (int cb = 0; cb < CW; cb += 64) {           // channel chunks (CW > 64 only for C > 256)
    const int cw = min(64, CW - cb);
    const int epl = 64 / cw;
    const int slot = lane / cw;
    const int c = cb + lane - slot * cw;
    VT acc = vzero<VT>();
    if (slot < epl)
      for (int k = k0 + slot; k < k1; k += epl) {
        const int64_t src = perm ? perm[k] : k;
        vadd(acc, y[src * CW + c]);
      }
    part[wave][lane] = acc;
    __syncthreads();
    if (row_ok && lane < cw) {
      VT t = part[wave][lane];
      for (int s2 = 1; s2 < epl; ++s2) vadd(t, part[wave][s2 * cw + lane]);
      x[r * CW + cb + lane] = t;
    }
    __syncthreads();
  }
}

// Few output rows (the per-atom sums: 1024 rows at B = 32) leave most CUs without a wave and each wave walking its
// segment alone (9 dependent perm -> row round trips: 9.6 us for 9 MB).  Here a whole workgroup owns a row: wave w
// takes entries k0 + w*EPL + slot, stride 4*EPL; the 4*EPL partial sums are combined in a fixed order.
template <typename VT>
__global__ __launch_bounds__(256) void segsum_block(const VT* __restrict__ y, const int32_t* __restrict__ perm,
                                                    const int32_t* __restrict__ seg_off, VT* __restrict__ x, int CW) {
  __shared__ VT part[4][64];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t r = blockIdx.x;
  const int k0 = seg_off[r], k1 = seg_off[r + 1];
  for (int cb = 0; cb < CW; cb += 64) {
    const int cw = min(64, CW - cb);
    const int epl = 64 / cw;
    const int slot = lane / cw;
    const int c = cb + lane - slot * cw;
    VT acc = vzero<VT>(), acc2 = vzero<VT>();
    if (slot < epl) {
      const int step = 4 * epl;
      int k = k0 + wave * epl + slot;
      for (; k + step < k1; k += 2 * step) {   // two entries in flight
        const int64_t sa = perm ? perm[k] : k, sb = perm ? perm[k + step] : k + step;
        const VT va = y[sa * CW + c], vb = y[sb * CW + c];
        vadd(acc, va), vadd(acc2, vb);
      }
      if (k < k1) vadd(acc, y[(int64_t)(perm ? perm[k] : k) * CW + c]);
      vadd(acc, acc2);
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && lane < cw) {
      VT t = vzero<VT>();
      for (int w = 0; w < 4; ++w)
        for (int s2 = 0; s2 < epl; ++s2) vadd(t, part[w][s2 * cw + lane]);
      x[r * CW + cb + lane] = t;
    }
    __syncthreads();
  }
}

// x[r, c] = sum_k sign_k * sum_{i in seg_k(r)} y_k[perm_k[i], c]: up to four CSR lists over the same rows in ONE launch.
// The position gradient is such a sum: dE/dR[a] collects per-triplet terms through the triplet's three atoms and
// per-edge terms through the edge's two atoms (gemnet.py:420-451, :334-350 differentiated); as separate segmented sums
// plus their combining adds that was 10 launches of ~5 us on the tail of the force pass.
// One workgroup per row, C <= 4 narrow rows: thread t takes entries t, t + 256, .. of every list (fixed assignment),
// partial sums fold through LDS in a fixed order — deterministic, no atomics.
struct SegTerms {
  const float* y[4];
  const int32_t* perm[4];
  const int32_t* seg[4];
  float sign[4];
  int n;
};

__global__ __launch_bounds__(256) void segsum_multi_kernel(const SegTerms T, float* __restrict__ x, int C) {
  __shared__ float part[256][4];
  const int64_t r = blockIdx.x;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < T.n; ++k) {
    const int k0 = T.seg[k][r], k1 = T.seg[k][r + 1];
    const float* __restrict__ y = T.y[k];
    const int32_t* __restrict__ perm = T.perm[k];
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = k0 + threadIdx.x; i < k1; i += 256) {
      const int64_t s = perm ? perm[i] : i;
      for (int c = 0; c < C; ++c) a[c] += y[s * C + c];
    }
    for (int c = 0; c < C; ++c) acc[c] += T.sign[k] * a[c];
  }
  for (int c = 0; c < 4; ++c) part[threadIdx.x][c] = acc[c];
  __syncthreads();
  for (int h = 128; h >= 1; h >>= 1) {
    if ((int)threadIdx.x < h)
      for (int c = 0; c < 4; ++c) part[threadIdx.x][c] += part[threadIdx.x + h][c];
    __syncthreads();
  }
  if ((int)threadIdx.x < C) x[r * C + threadIdx.x] = part[0][threadIdx.x];
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int gn_gather_rows_f32(const float* x, const int32_t* idx, float* y, int64_t T, int C,
                                  void* stream) {
  if (T <= 0 || C <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (C % 4 == 0 && aligned16(x) && aligned16(y)) {
    hipLaunchKernelGGL(gather_rows_v4, dim3(grid_for(T * (C / 4))), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(x), idx, reinterpret_cast<float4*>(y), T, C / 4);
  } else {
    hipLaunchKernelGGL(gather_rows_s, dim3(grid_for(T * C)), dim3(256), 0, st, x, idx, y, T, C);
  }
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_gather_mul_f32(const float* x, const int32_t* idx, const float* m, float* y, int64_t T, int C, float scale,
                                 void* stream) {
  if (T <= 0 || C <= 0) return 0;
  if (C % 4 != 0 || !aligned16(x) || !aligned16(m) || !aligned16(y)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(gather_mul_v4, dim3(grid_for(T * (C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(x), idx, reinterpret_cast<const float4*>(m),
                     reinterpret_cast<float4*>(y), T, C / 4, scale);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_segsum_rows_f32(const float* y, const int32_t* perm, const int32_t* seg_off,
                                  float* x, int64_t N, int C, void* stream) {
  if (N <= 0 || C <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((N + 3) / 4)), block(256);
  const bool v4 = C % 4 == 0 && aligned16(x) && aligned16(y);
  if (N <= 4096) {   // fewer rows than the chip has wave slots: a workgroup per row
    if (v4)
      hipLaunchKernelGGL(segsum_block<float4>, dim3((unsigned)N), block, 0, st, reinterpret_cast<const float4*>(y), perm,
                         seg_off, reinterpret_cast<float4*>(x), C / 4);
    else
      hipLaunchKernelGGL(segsum_block<float>, dim3((unsigned)N), block, 0, st, y, perm, seg_off, x, C);
  } else if (v4) {
    hipLaunchKernelGGL(segsum_wave<float4>, grid, block, 0, st, reinterpret_cast<const float4*>(y), perm,
                       seg_off, reinterpret_cast<float4*>(x), N, C / 4);
  } else {
    hipLaunchKernelGGL(segsum_wave<float>, grid, block, 0, st, y, perm, seg_off, x, N, C);
  }
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_segsum_multi_f32(int n_terms, const float* const* y, const int32_t* const* perm,
                                   const int32_t* const* seg_off, const float* sign, float* x, int64_t N, int C,
                                   void* stream) {
  if (N <= 0) return 0;
  if (n_terms < 1 || n_terms > 4 || C < 1 || C > 4) return (int)hipErrorInvalidValue;
  SegTerms T;
  T.n = n_terms;
  for (int k = 0; k < 4; ++k) {
    T.y[k] = k < n_terms ? y[k] : nullptr;
    T.perm[k] = k < n_terms ? perm[k] : nullptr;
    T.seg[k] = k < n_terms ? seg_off[k] : nullptr;
    T.sign[k] = k < n_terms ? sign[k] : 0.f;
  }
  hipLaunchKernelGGL(segsum_multi_kernel, dim3((unsigned)N), dim3(256), 0, static_cast<hipStream_t>(stream), T, x, C);
  GN_LAUNCH_CHECK();
  return 0;
}
