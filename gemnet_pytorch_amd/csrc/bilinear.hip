// CSR-segmented bilinear aggregation (kernel K1 and its two adjoints).
//
// The reference (gemnet/model/layers/efficient.py:159-189) scatters the (T,C) gathered edge
// embeddings into a zero-padded (E,Kmax,C) tensor, does the same with the harmonics
// (basis_layers.py:153-159) and multiplies them with bmm.  Triplets/quadruplets arrive already
// sorted by their reduce edge (data_container.py:324-328,369-375), so here a group of C lanes
// owns one reduce edge, walks its contiguous segment and keeps the S partial sums in registers:
// no padding, no zero-fill, no atomics, each output written exactly once.
#include "common.h"

namespace {

// Sm[e,s,c] = sum_{t in seg(e)} Y[t,s] * x[g(t),c]
template <int S>
__global__ __launch_bounds__(256) void bil_reduce_kernel(const float* __restrict__ Y,
                                                         const float* __restrict__ x,
                                                         const int32_t* __restrict__ expand_idx,
                                                         const int32_t* __restrict__ seg_off,
                                                         float* __restrict__ Sm, int64_t E, int C) {
  const int epb = blockDim.x / C;
  const int el = threadIdx.x / C;
  const int c = threadIdx.x - el * C;
  const int64_t e = (int64_t)blockIdx.x * epb + el;
  if (el >= epb || e >= E) return;
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  float acc[S];
#pragma unroll
  for (int s = 0; s < S; ++s) acc[s] = 0.f;
  // 4 triplets per trip: the 4 index loads, then the 4 gathered x rows, are independent and in
  // flight together (the serial idx -> x dependency otherwise exposes one L2 latency per triplet)
  int t = t0;
  for (; S <= 8 && t + 4 <= t1; t += 4) {   // (S = 49: the 4x register footprint costs more than it buys)
    const int g0 = expand_idx[t], g1 = expand_idx[t + 1], g2 = expand_idx[t + 2], g3 = expand_idx[t + 3];
    const float x0 = x[(int64_t)g0 * C + c], x1 = x[(int64_t)g1 * C + c];
    const float x2 = x[(int64_t)g2 * C + c], x3 = x[(int64_t)g3 * C + c];
    const float* __restrict__ y = Y + (int64_t)t * S;
#pragma unroll
    for (int s = 0; s < S; ++s)
      acc[s] = fmaf(y[3 * S + s], x3, fmaf(y[2 * S + s], x2, fmaf(y[S + s], x1, fmaf(y[s], x0, acc[s]))));
  }
  for (; t < t1; ++t) {
    const float xv = x[(int64_t)expand_idx[t] * C + c];
    const float* __restrict__ y = Y + (int64_t)t * S;
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = fmaf(y[s], xv, acc[s]);
  }
  float* __restrict__ o = Sm + e * S * C + c;
#pragma unroll
  for (int s = 0; s < S; ++s) o[(int64_t)s * C] = acc[s];
}

// dx[j,c] = sum_{k in segT(j)} sum_s Y[t,s] * dSm[r(t),s,c],  t = permT[k]
template <int S>
__global__ __launch_bounds__(256) void bil_reduce_t_kernel(const float* __restrict__ Y,
                                                           const float* __restrict__ dSm,
                                                           const int32_t* __restrict__ reduce_idx,
                                                           const int32_t* __restrict__ permT,
                                                           const int32_t* __restrict__ segT_off,
                                                           float* __restrict__ dx, int64_t J, int C) {
  const int rpb = blockDim.x / C;
  const int rl = threadIdx.x / C;
  const int c = threadIdx.x - rl * C;
  const int64_t j = (int64_t)blockIdx.x * rpb + rl;
  if (rl >= rpb || j >= J) return;
  const int k0 = segT_off[j], k1 = segT_off[j + 1];
  float acc = 0.f, acc2 = 0.f;
  int k = k0;
  // two entries in flight (permT -> reduce_idx -> dSm row is a 3-deep dependent chain); for the
  // tensor basis (S = 49) the doubled register footprint costs more than the overlap buys
  for (; S <= 8 && k + 2 <= k1; k += 2) {
    const int ta = permT[k], tb = permT[k + 1];
    const int ra = reduce_idx[ta], rb = reduce_idx[tb];
    const float* __restrict__ ya = Y + (int64_t)ta * S;
    const float* __restrict__ yb = Y + (int64_t)tb * S;
    const float* __restrict__ da = dSm + (int64_t)ra * S * C + c;
    const float* __restrict__ db = dSm + (int64_t)rb * S * C + c;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      acc = fmaf(ya[s], da[(int64_t)s * C], acc);
      acc2 = fmaf(yb[s], db[(int64_t)s * C], acc2);
    }
  }
  for (; k < k1; ++k) {
    const int t = permT[k];
    const float* __restrict__ y = Y + (int64_t)t * S;
    const float* __restrict__ d = dSm + (int64_t)reduce_idx[t] * S * C + c;
#pragma unroll
    for (int s = 0; s < S; ++s) acc = fmaf(y[s], d[(int64_t)s * C], acc);
  }
  dx[j * C + c] = acc + acc2;
}

// Same adjoint for index sets where r(t) and g(t) always fall in the same small GROUP of rows — the triplets
// c->a<-b: reduce edge (c->a) and expand edge (b->a) both end in atom a, so the ~deg(a) blocks dSm[e] of the edges
// into a serve all deg(a)^2 triplets around a.  The ungrouped kernel above pulls dSm[r(t)] (S*C floats) through
// L2 once per triplet (332 k x 1.8 KB = 600 MB, 315 MB of it missing to HBM at B = 32); here one workgroup owns one
// atom, parks its dSm blocks in LDS once (coalesced), and every expand row of the atom reads them from there.
// Traffic: Y + dSm once + dx.
// One wave per expand row (C = 64 lanes = channels), 16 waves per workgroup, two rows of a wave in flight.  The
// entries of a row's transposed segment are fetched 64 at a time, one per lane (grp_kseg -> permT, rposT -> Y:
// three dependent load rounds per ROW, overlapping the tile fill, instead of three per triplet) and handed to
// the whole wave with v_readlane; the inner loop is then LDS reads and FMAs only.
template <int S>
struct grouped_row {
  int j, k0, k1, p;
  float y[S];
};

template <int S>
__global__ __launch_bounds__(1024) void bil_reduce_t_grouped_kernel(
    const float* __restrict__ Y, const float* __restrict__ dSm, const int32_t* __restrict__ grp_rows,
    const int32_t* __restrict__ grp_off, const int2* __restrict__ grp_kseg, const int32_t* __restrict__ permT,
    const int32_t* __restrict__ rposT, float* __restrict__ dx, int max_rows) {
  constexpr int C = 64, SC = S * C, NV = SC / 4;
  extern __shared__ float gtile[];  // [rows of the group][S*C]
  const int g = blockIdx.x;
  const int r0 = grp_off[g], n = grp_off[g + 1] - r0;
  if (n <= 0) return;
  // the tile was sized for `max_rows` rows — a caller-supplied bound when the plan is built inside a hipGraph
  // (padded.py: largest in-degree of an atom).  A group beyond it would write past the LDS allocation: abort loudly.
  if (n > max_rows) __builtin_trap();
  for (int i = threadIdx.x; i < n * NV; i += 1024) {
    const int l = i / NV, v = i - l * NV;
    const float4 d = reinterpret_cast<const float4*>(dSm + (int64_t)grp_rows[r0 + l] * SC)[v];
    reinterpret_cast<float4*>(gtile + l * SC)[v] = d;
  }
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto fetch = [&](grouped_row<S>& r, int kb) {  // lane i takes entry kb + i of the row's transposed segment
    const bool on = kb + lane < r.k1;
    const int t = on ? permT[kb + lane] : 0;
    r.p = on ? rposT[kb + lane] * SC : 0;
#pragma unroll
    for (int s = 0; s < S; ++s) r.y[s] = on ? Y[(int64_t)t * S + s] : 0.f;
  };
  auto open_row = [&](grouped_row<S>& r, int l) {
    r.j = -1, r.k0 = r.k1 = 0;
    if (l >= n) return;
    const int2 ks = grp_kseg[r0 + l];
    r.j = __builtin_amdgcn_readfirstlane(grp_rows[r0 + l]);
    r.k0 = __builtin_amdgcn_readfirstlane(ks.x);
    r.k1 = __builtin_amdgcn_readfirstlane(ks.y);
    fetch(r, r.k0);
  };
  auto finish_row = [&](grouped_row<S>& r) {
    if (r.j < 0) return;
    float acc = 0.f, acc2 = 0.f;
    for (int kb = r.k0; kb < r.k1; kb += 64) {
      if (kb != r.k0) fetch(r, kb);
      const int m = min(64, r.k1 - kb);
      int k = 0;
      for (; k + 2 <= m; k += 2) {
        const float* da = gtile + __builtin_amdgcn_readlane(r.p, k) + lane;
        const float* db = gtile + __builtin_amdgcn_readlane(r.p, k + 1) + lane;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.y[s]), k)), da[s * C], acc);
          acc2 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.y[s]), k + 1)), db[s * C], acc2);
        }
      }
      if (k < m) {
        const float* da = gtile + __builtin_amdgcn_readlane(r.p, k) + lane;
#pragma unroll
        for (int s = 0; s < S; ++s)
          acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.y[s]), k)), da[s * C], acc);
      }
    }
    dx[(int64_t)r.j * C + lane] = acc + acc2;
  };
  grouped_row<S> ra, rb;
  open_row(ra, w), open_row(rb, w + 16);  // index loads of the first rows overlap the tile fill
  __syncthreads();
  for (int l = w; l < n; l += 32) {
    finish_row(ra), finish_row(rb);
    open_row(ra, l + 32), open_row(rb, l + 48);
  }
}

// Same adjoint, grouped by REDUCE edge instead of by expand row:  dxt[t,c] = sum_s Y[t,s] * dSm[r(t),s,c].
// For the tensor basis (S = 49) the expand-row form above re-reads the 6 KB block dSm[r(t)] for every quadruplet
// (9 M x 6.3 KB = 56 GB of L2 traffic per launch at B = 32: 3.8 ms).  Quadruplets are sorted by reduce edge, so
// here a group of C lanes owns one edge, keeps dSm[e,:,c] in S registers, streams its segment of Y through LDS
// (coalesced load, broadcast ds_read_b128) and writes the per-quadruplet rows; the sum over each expand row is
// then one ordinary CSR segmented sum (gn_segsum_rows_f32).  Traffic: Y + dSm once + 2 x (Q x C) floats.
template <int S, int CH>
__global__ __launch_bounds__(256) void bil_expand_kernel(const float* __restrict__ Y, const float* __restrict__ dSm,
                                                         const int32_t* __restrict__ seg_off,
                                                         float* __restrict__ dxt, int64_t E, int C) {
  constexpr int SP = (S + 3) / 4 * 4;            // padded row (16-B aligned rows in LDS)
  extern __shared__ __attribute__((aligned(16))) float ys[];   // [groups][CH][SP]
  __shared__ int nchunk_s;
  const int gpb = blockDim.x / C;                 // edge groups per block
  const int g = threadIdx.x / C;
  const int c = threadIdx.x - g * C;
  const int64_t e = (int64_t)blockIdx.x * gpb + g;
  const bool live = g < gpb && e < E;
  const int t0 = live ? seg_off[e] : 0;
  const int t1 = live ? seg_off[e + 1] : 0;
  if (threadIdx.x == 0) nchunk_s = 0;
  __syncthreads();
  if (live && c == 0) atomicMax(&nchunk_s, (t1 - t0 + CH - 1) / CH);
  __syncthreads();
  const int nchunk = nchunk_s;                    // uniform trip count: barriers below are unconditional
  float d[S];
  if (live) {
#pragma unroll
    for (int s = 0; s < S; ++s) d[s] = dSm[(e * S + s) * C + c];
  }
  float* __restrict__ yg = ys + (size_t)g * CH * SP;
  for (int ch = 0; ch < nchunk; ++ch) {
    const int tb = t0 + ch * CH;
    const int nt = min(CH, t1 - tb);              // <= 0 when this group is done
    __syncthreads();                               // previous chunk fully consumed
    if (live && nt > 0) {
      const float* __restrict__ src = Y + (int64_t)tb * S;
      for (int i = c; i < nt * S; i += C) {
        const int r = i / S;
        yg[r * SP + (i - r * S)] = src[i];
      }
      if (SP > S)
        for (int r = c; r < nt; r += C)
          for (int p = S; p < SP; ++p) yg[r * SP + p] = 0.f;
    }
    __syncthreads();
    if (live && nt > 0) {
      for (int tt = 0; tt < nt; ++tt) {
        const float4* __restrict__ row = reinterpret_cast<const float4*>(yg + tt * SP);
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < SP / 4; ++q) {
          const float4 y = row[q];
          acc = fmaf(y.x, d[4 * q], acc);
          if (4 * q + 1 < S) acc = fmaf(y.y, d[4 * q + 1], acc);
          if (4 * q + 2 < S) acc = fmaf(y.z, d[4 * q + 2], acc);
          if (4 * q + 3 < S) acc = fmaf(y.w, d[4 * q + 3], acc);
        }
        dxt[(int64_t)(tb + tt) * C + c] = acc;
      }
    }
  }
}

// dY[t,s] = sum_c dSm[r(t),s,c] * x[g(t),c]; one block per reduce edge, dSm[e] staged in LDS
// with row stride C+4 (lanes of different s hit different 16-B slots; same s broadcasts).
__global__ __launch_bounds__(256) void bil_dot_kernel(const float* __restrict__ dSm,
                                                      const float* __restrict__ x,
                                                      const int32_t* __restrict__ expand_idx,
                                                      const int32_t* __restrict__ seg_off,
                                                      float* __restrict__ dY, int S, int C, int vec) {
  extern __shared__ __attribute__((aligned(16))) float dS[];
  const int64_t e = blockIdx.x;
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  if (t1 <= t0) return;
  const int ld = C + 4;
  const float* __restrict__ src = dSm + e * S * C;
  for (int i = threadIdx.x; i < S * C; i += blockDim.x) {
    const int s = i / C;
    dS[s * ld + (i - s * C)] = src[i];
  }
  __syncthreads();
  const int n = (t1 - t0) * S;
  for (int p = threadIdx.x; p < n; p += blockDim.x) {
    const int tt = p / S;
    const int s = p - tt * S;
    const int t = t0 + tt;
    const float* __restrict__ xr = x + (int64_t)expand_idx[t] * C;
    const float* dr = dS + s * ld;
    float acc = 0.f;
    if (vec) {
      for (int c = 0; c < C; c += 4) {
        const float4 a = *reinterpret_cast<const float4*>(dr + c);
        const float4 b = *reinterpret_cast<const float4*>(xr + c);
        acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc);
        acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
      }
    } else {
      for (int c = 0; c < C; ++c) acc = fmaf(dr[c], xr[c], acc);
    }
    dY[(int64_t)t * S + s] = acc;
  }
}

// Fused K1 + K2: the S partial sums of an edge stay in registers and are
// immediately projected with the edge's rbf_W1 block:
//   Sm[e,s,c] = sum_{t in seg(e)} Y[t,s] x[g(t),c]        (also written: the adjoint needs it)
//   P[e,i,c]  = sum_s B[e,s,i] Sm[e,s,c]
// replacing the separate per-edge bmm launch (torch.matmul(rbf_W1, sum_k), efficient.py:180).
// B[e] (S x I) is staged in LDS once per edge group; lanes of one edge read it as broadcasts.
template <int S>
__global__ __launch_bounds__(256) void bil_reduce_project_kernel(
    const float* __restrict__ Y, const float* __restrict__ x, const int32_t* __restrict__ expand_idx,
    const int32_t* __restrict__ seg_off, const float* __restrict__ B, float* __restrict__ Sm,
    float* __restrict__ P, int64_t E, int C, int I) {
  extern __shared__ __attribute__((aligned(16))) float Bl[];   // [epb][S*I]
  const int epb = blockDim.x / C;
  const int el = threadIdx.x / C;
  const int c = threadIdx.x - el * C;
  const int64_t e0 = (int64_t)blockIdx.x * epb;
  const int SI = S * I;
  for (int i = threadIdx.x; i < epb * SI; i += blockDim.x) {
    const int64_t ee = e0 + i / SI;
    Bl[i] = ee < E ? B[ee * SI + (i % SI)] : 0.f;
  }
  __syncthreads();
  const int64_t e = e0 + el;
  if (e >= E) return;
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  float acc[S];
#pragma unroll
  for (int s = 0; s < S; ++s) acc[s] = 0.f;
  // 4 triplets per trip: the 4 index loads, then the 4 gathered x rows, are independent and in
  // flight together (the serial idx -> x dependency otherwise exposes one L2 latency per triplet)
  int t = t0;
  for (; S <= 8 && t + 4 <= t1; t += 4) {   // (S = 49: the 4x register footprint costs more than it buys)
    const int g0 = expand_idx[t], g1 = expand_idx[t + 1], g2 = expand_idx[t + 2], g3 = expand_idx[t + 3];
    const float x0 = x[(int64_t)g0 * C + c], x1 = x[(int64_t)g1 * C + c];
    const float x2 = x[(int64_t)g2 * C + c], x3 = x[(int64_t)g3 * C + c];
    const float* __restrict__ y = Y + (int64_t)t * S;
#pragma unroll
    for (int s = 0; s < S; ++s)
      acc[s] = fmaf(y[3 * S + s], x3, fmaf(y[2 * S + s], x2, fmaf(y[S + s], x1, fmaf(y[s], x0, acc[s]))));
  }
  for (; t < t1; ++t) {
    const float xv = x[(int64_t)expand_idx[t] * C + c];
    const float* __restrict__ y = Y + (int64_t)t * S;
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = fmaf(y[s], xv, acc[s]);
  }
  float* __restrict__ so = Sm + e * S * C + c;
#pragma unroll
  for (int s = 0; s < S; ++s) so[(int64_t)s * C] = acc[s];
  const float* bl = Bl + el * SI;
  float* __restrict__ po = P + e * (int64_t)I * C + c;
  for (int i = 0; i < I; ++i) {
    float p = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) p = fmaf(bl[s * I + i], acc[s], p);
    po[(int64_t)i * C] = p;
  }
}

// Tensor-basis (S = 49, C = I = 32) form of the fused K1 + K2 on the matrix cores.  For one reduce edge
//   Sm[e] (S x C) = Yseg^T (S x K4) @ Xseg (K4 x C),   Xseg[t] = x[g(t)],   K4 ~ 500 quadruplets,
// is a GEMM with M = S (padded to 64), N = C, K = K4: one wave per edge, v_mfma_f32_16x16x4_f32 with the operand
// fragments loaded straight from global memory in fragment layout — lane (l15, lg) of a K-step of 4 quadruplets
// reads Y[t + lg][16 mt + l15] (16 lanes = 64 contiguous bytes of one Y row) and x[g(t + lg)][16 nt + l15] — so the
// per-quadruplet cost drops from 49 scalar loads + 49 FMAs per lane to 7 loads + 8 MFMAs per 4 quadruplets.
// K2 (P = B[e]^T Sm, 32 x 49 x 32) runs on the same cores with Sm passed through LDS.
typedef float v4f_b __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void bil_reduce_project_mfma49_kernel(
    const float* __restrict__ Y, const float* __restrict__ x, const int32_t* __restrict__ expand_idx,
    const int32_t* __restrict__ seg_off, const float* __restrict__ B, float* __restrict__ Sm,
    float* __restrict__ P, int64_t E) {
  constexpr int S = 49, C = 32, I = 32, LD = C + 4;
  __shared__ __attribute__((aligned(16))) float sml[4][52][LD];   // Sm of this wave's edge, rows 49..51 zero
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= E) return;
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  v4f_b acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (v4f_b){0.f, 0.f, 0.f, 0.f};
  // Loads are unconditional at clamped (valid) addresses and masked afterwards: `cond ? *p : 0` costs a branch per
  // load.  Tile mt = 3 holds only s = 48: its other rows read the same element (they feed rows >= 49, zeroed below).
  const int tlast = max(t1 - 1, t0);
  auto load = [&](int t, float (&a)[4], float (&b)[2]) {
    const int tq = t + lg;
    const bool ok = tq < t1;
    const int tc = min(tq, tlast);
    const float* __restrict__ yr = Y + (int64_t)tc * S;
    const float y0 = yr[l15], y1 = yr[16 + l15], y2 = yr[32 + l15], y3 = yr[48];
    a[0] = ok ? y0 : 0.f;
    a[1] = ok ? y1 : 0.f;
    a[2] = ok ? y2 : 0.f;
    a[3] = ok ? y3 : 0.f;
    const float* __restrict__ xr = x + (int64_t)expand_idx[tc] * C + l15;
    const float x0 = xr[0], x1 = xr[16];
    b[0] = ok ? x0 : 0.f;
    b[1] = ok ? x1 : 0.f;
  };
  float a0[4] = {0.f, 0.f, 0.f, 0.f}, b0[2] = {0.f, 0.f}, a1[4], b1[2];
  if (t0 < t1) load(t0, a0, b0);   // (an edge without quadruplets touches neither Y nor x)
  for (int t = t0; t < t1; t += 8) {
    load(t + 4, a1, b1);               // next K-step in flight under this one's MFMAs
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[mt], b0[nt], acc[mt][nt], 0, 0, 0);
    load(t + 8, a0, b0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[mt], b1[nt], acc[mt][nt], 0, 0, 0);
  }
  // D layout: col = l15 (c within tile), row = 4 lg + r (s within tile)
  float* __restrict__ so = Sm + e * (int64_t)S * C;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int srow = 16 * mt + 4 * lg + r;
        const float v = acc[mt][nt][r];
        if (srow < S) so[srow * C + 16 * nt + l15] = v;
        if (srow < 52) sml[wave][srow][16 * nt + l15] = srow < S ? v : 0.f;
      }
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes are visible to its own reads
  __builtin_amdgcn_wave_barrier();
  // K2: P[i,c] = sum_s B[e,s,i] Sm[s,c]:  A[m = i][k = s] = B[e][s][i],  Bop[k = s][n = c] = Sm[s][c]
  v4f_b pacc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) pacc[mt][nt] = (v4f_b){0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ be = B + e * (int64_t)S * I;
#pragma unroll
  for (int kk = 0; kk < 13; ++kk) {
    const int sk = 4 * kk + lg;
    const bool ok = sk < S;
    const float a_0 = ok ? be[sk * I + l15] : 0.f;
    const float a_1 = ok ? be[sk * I + 16 + l15] : 0.f;
    const float b_0 = sml[wave][sk][l15];
    const float b_1 = sml[wave][sk][16 + l15];
    pacc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_0, b_0, pacc[0][0], 0, 0, 0);
    pacc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_0, b_1, pacc[0][1], 0, 0, 0);
    pacc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_1, b_0, pacc[1][0], 0, 0, 0);
    pacc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_1, b_1, pacc[1][1], 0, 0, 0);
  }
  float* __restrict__ po = P + e * (int64_t)I * C;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) po[(16 * mt + 4 * lg + r) * C + 16 * nt + l15] = pacc[mt][nt][r];
}

// Spherical-basis (S = 7, C = 64, I = 16: GemNet-T / the triplet branch) form of the fused K1 + K2 on the matrix cores,
// one wave per reduce edge.  The scalar kernel above issues 7 FMAs per (triplet, lane) plus 112 per lane for K2 and
// is instruction-bound (52 us for 18 k edges / 332 k triplets against 21 us of HBM traffic); here
//   K1  Sm[s,c] = sum_t Y[t,s] x[g(t),c]   M = s (7 of 16 rows), N = c (4 tiles), K = t (4 triplets per MFMA):
//       lane (l15, lg) reads Y[t + lg][l15] and x[g(t + lg)][16 nt + l15] straight into fragment layout;
//   K2  P[i,c] = sum_s B[s,i] Sm[s,c]      consumed from the K1 accumulators in place: in K-step r lane group lg
//       supplies s = 4 lg + r for BOTH operands (its own D row r), so Sm never passes through LDS.
// EXT (the tangent sweep of the training step, gn_bil_reduce_project2_f32): the K1 accumulators start from Sm_init[e]
// (dSm = K1(dY, x) + K1(Y, dx) over two calls), K2 takes a second term from given blocks (P = B^T Sm + B2^T Sm2, i.e.
// dP = B^T dSm + dB^T Sm) and is skipped altogether when P is null.
template <bool EXT>
__global__ __launch_bounds__(256) void bil_reduce_project_mfma7_kernel(
    const float* __restrict__ Y, const float* __restrict__ x, const int32_t* __restrict__ expand_idx,
    const int32_t* __restrict__ seg_off, const float* __restrict__ B, float* __restrict__ Sm,
    float* __restrict__ P, int64_t E, const float* __restrict__ Sm_init, const float* __restrict__ B2,
    const float* __restrict__ Sm2) {
  constexpr int S = 7, C = 64, I = 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= E) return;
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  const bool srow = l15 < S;
  const float* __restrict__ be = B + e * (int64_t)S * I;
  // Loads are UNCONDITIONAL at clamped (always valid) addresses and masked with a select afterwards: a load inside
  // `cond ? *p : 0` becomes a branch per dword (and splits float4 loads into four of them).
  float bk[4];   // B[s = 4 lg + r][i = l15]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = be[min(4 * lg + r, S - 1) * I + l15];
    bk[r] = (4 * lg + r) < S ? v : 0.f;
  }
  v4f_b acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = (v4f_b){0.f, 0.f, 0.f, 0.f};
  if (EXT && Sm_init) {   // D layout: row 4 lg + r (s), col 16 nt + l15 (c); rows s >= 7 stay zero
    const float* __restrict__ si = Sm_init + e * (int64_t)S * C;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = si[min(4 * lg + r, S - 1) * C + 16 * nt + l15];
        acc[nt][r] = (4 * lg + r) < S ? v : 0.f;
      }
  }
  const int tlast = max(t1 - 1, t0), scl = min(l15, S - 1);
  auto load = [&](int t, float& a, float (&b)[4]) {
    const int tq = t + lg;
    const bool ok = tq < t1;
    const int tc = min(tq, tlast);
    const float yv = Y[(int64_t)tc * S + scl];
    a = (ok && srow) ? yv : 0.f;
    const float* __restrict__ xr = x + (int64_t)expand_idx[tc] * C + l15;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const float v = xr[16 * nt];
      b[nt] = ok ? v : 0.f;
    }
  };
  float a0 = 0.f, b0[4] = {0.f, 0.f, 0.f, 0.f}, a1, b1[4];
  if (t0 < t1) load(t0, a0, b0);   // (an edge without triplets touches neither Y nor x: they may be empty)
  for (int t = t0; t < t1; t += 8) {
    load(t + 4, a1, b1);               // next K-step in flight under this one's MFMAs
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[nt], acc[nt], 0, 0, 0);
    load(t + 8, a0, b0);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[nt], acc[nt], 0, 0, 0);
  }
  // D layout: col = l15 (c within tile), row = 4 lg + r (s)
  float* __restrict__ so = Sm + e * (int64_t)S * C;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * lg + r < S) so[(4 * lg + r) * C + 16 * nt + l15] = acc[nt][r];
  if (EXT && !P) return;
  v4f_b pacc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) pacc[nt] = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) pacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bk[r], acc[nt][r], pacc[nt], 0, 0, 0);
  if (EXT && B2) {   // + B2[e]^T Sm2[e]: the same K-steps with both operands read from memory in fragment layout
    const float* __restrict__ b2 = B2 + e * (int64_t)S * I;
    const float* __restrict__ s2 = Sm2 + e * (int64_t)S * C;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int srw = min(4 * lg + r, S - 1);
      const bool ok = (4 * lg + r) < S;
      const float bv = b2[srw * I + l15];
      const float bq = ok ? bv : 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float sv = s2[srw * C + 16 * nt + l15];
        pacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq, ok ? sv : 0.f, pacc[nt], 0, 0, 0);
      }
    }
  }
  float* __restrict__ po = P + e * (int64_t)I * C;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) po[(4 * lg + r) * C + 16 * nt + l15] = pacc[nt][r];
}

// K1 + K2 + K3 of the bilinear layer in ONE launch for the spherical basis (S = 7, C = 64, I = 16, O = 64; eval /
// frozen weights, where P is not needed afterwards): 16 reduce edges per workgroup, 16 waves.  Each wave runs K1 / K2
// for one edge exactly like bil_reduce_project_mfma7_kernel but parks P[e] (16 x 64 = one 1024-long row of the K3
// product) in LDS instead of HBM; then the workgroup multiplies its 16 x 1024 tile with the (64 x 1024, k-contiguous)
// bilinear weight on the matrix cores (wave (kq, nt): one 16 x 16 tile over a quarter of K, quarters summed in a fixed
// order through LDS).  Saves the 74 MB write + read of P and one launch: 79 -> 67 us per block stand-alone
// (tools/k3fused_bench.py; 32-edge tiles with 8 waves: 111 us, 16-edge tiles with 8 waves: 75 us).
constexpr int FTE = 16, FLDP = 1024 + 4;

// HP: K3 on the fp16 matrix pipe with split operands.  On the f32-input MFMA the 16 x 1024 x 64 product of a workgroup is
// 1024 v_mfma_f32_16x16x4_f32 = 8.2 k cycles of every SIMD's matrix pipe (the f32 vector rate, chain2.hip); with P written to
// LDS as two fp16 planes (hi, 2^11 lo: the K2 epilogue splits its 16 values per lane) and the weight given PRE-SPLIT in
// fragment order (gn_pack_weight_split_fmt(W2T, 64, 1024, GN_SPLIT_F16X2): no conversion in the kernel) it is 384
// v_mfma_f32_16x16x32_f16 = 1.5 k cycles.  Computed transposed (A = 16 rows of W2T, B = the P rows of the 16 edges), so a lane
// ends with four consecutive output columns of one edge: one float4 store.
typedef _Float16 h8_b __attribute__((ext_vector_type(8)));
constexpr int FPH = 1024 + 8;      // fp16 elements per P-plane row (2 064 B: the 16 rows of a b128 read fall in distinct banks)

template <bool HP>
__global__ __launch_bounds__(1024) void bil_fused_fwd_mfma7_kernel(const float* __restrict__ Y, const float* __restrict__ x,
                                                     const int32_t* __restrict__ expand_idx,
                                                     const int32_t* __restrict__ seg_off, const float* __restrict__ B,
                                                     const float* __restrict__ W2T, const uint4* __restrict__ W2Tp,
                                                     float* __restrict__ Sm,
                                                     float* __restrict__ out, int64_t E, float alpha) {
  constexpr int S = 7, C = 64, I = 16;
  extern __shared__ __attribute__((aligned(16))) float Pl[];   // [FFTE][FFLDP] (HP: two fp16 planes [FTE][FPH]) + partial tiles
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e0 = (int64_t)blockIdx.x * FTE;
  const bool srow = l15 < S;
  const int scl = min(l15, S - 1);
  for (int q = 0; q < 1; ++q) {
    const int row = wave;
    const int64_t e = e0 + row;
    float* __restrict__ prow = Pl + row * FLDP;
    v4f_b pacc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) pacc[nt] = (v4f_b){0.f, 0.f, 0.f, 0.f};
    if (e < E) {   // (wave-uniform)
      const int t0 = seg_off[e], t1 = seg_off[e + 1];
      const float* __restrict__ be = B + e * (int64_t)S * I;
      float bk[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = be[min(4 * lg + r, S - 1) * I + l15];
        bk[r] = (4 * lg + r) < S ? v : 0.f;
      }
      v4f_b acc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = (v4f_b){0.f, 0.f, 0.f, 0.f};
      const int tlast = max(t1 - 1, t0);
      auto load = [&](int t, float& a, float (&b)[4]) {
        const int tq = t + lg;
        const bool ok = tq < t1;
        const int tc = min(tq, tlast);
        const float yv = Y[(int64_t)tc * S + scl];
        a = (ok && srow) ? yv : 0.f;
        const float* __restrict__ xr = x + (int64_t)expand_idx[tc] * C + l15;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const float v = xr[16 * nt];
          b[nt] = ok ? v : 0.f;
        }
      };
      // (requesting the gathers of the whole segment — up to 32 triplets — before the first MFMA was measured in round 6:
      //  56-58 us against 43-47 us per launch, profiles/r6_bil_fused_prefetch_all.txt; the two-step look-ahead stays)
      float a0 = 0.f, b0[4] = {0.f, 0.f, 0.f, 0.f}, a1, b1[4];
      if (t0 < t1) load(t0, a0, b0);
      for (int t = t0; t < t1; t += 8) {
        load(t + 4, a1, b1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[nt], acc[nt], 0, 0, 0);
        load(t + 8, a0, b0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[nt], acc[nt], 0, 0, 0);
      }
      float* __restrict__ so = Sm + e * (int64_t)S * C;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * lg + r < S) so[(4 * lg + r) * C + 16 * nt + l15] = acc[nt][r];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) pacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bk[r], acc[nt][r], pacc[nt], 0, 0, 0);
    }
    // P[e][i = 4 lg + r][c = 16 nt + l15] -> K3 row, k = i * 64 + c  (zeros for the rows past E)
    if constexpr (HP) {
      _Float16* __restrict__ ph = reinterpret_cast<_Float16*>(Pl) + row * FPH;
      _Float16* __restrict__ pl = ph + FTE * FPH;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = pacc[nt][r];
          const _Float16 h = (_Float16)v;
          const int k = (4 * lg + r) * C + 16 * nt + l15;
          ph[k] = h;
          pl[k] = (_Float16)((v - (float)h) * 2048.f);
        }
    } else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) prow[(4 * lg + r) * C + 16 * nt + l15] = pacc[nt][r];
    }
  }
  __syncthreads();
  if constexpr (HP) {
    // wave (kq, nt): output columns 16 nt .. 16 nt + 15 of all 16 edges over the k-chunks 8 kq .. 8 kq + 7
    const int kq = wave >> 2, nt = wave & 3;
    const _Float16* __restrict__ ph = reinterpret_cast<const _Float16*>(Pl) + l15 * FPH + 8 * lg;
    const _Float16* __restrict__ pl = ph + FTE * FPH;
    const uint4* __restrict__ wp = W2Tp + ((size_t)(nt * 32 + 8 * kq) * 2) * 64 + lane;     // [tile][chunk][plane][lane]
    float* red = reinterpret_cast<float*>(reinterpret_cast<_Float16*>(Pl) + 2 * FTE * FPH);  // [3 kq][4 nt][64 lanes][4]
    v4f_b ch = (v4f_b){0.f, 0.f, 0.f, 0.f}, cx = ch;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const h8_b ah = __builtin_bit_cast(h8_b, wp[(2 * c) * 64]);
      const h8_b al = __builtin_bit_cast(h8_b, wp[(2 * c + 1) * 64]);
      const h8_b bh = *reinterpret_cast<const h8_b*>(ph + 32 * (8 * kq + c));
      const h8_b bl = *reinterpret_cast<const h8_b*>(pl + 32 * (8 * kq + c));
      ch = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, ch, 0, 0, 0);
      cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, cx, 0, 0, 0);
      cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, cx, 0, 0, 0);
    }
    v4f_b c0 = ch + cx * (1.f / 2048.f);
    if (kq > 0) *reinterpret_cast<v4f_b*>(red + (((kq - 1) * 4 + nt) * 64 + lane) * 4) = c0;
    __syncthreads();
    if (kq == 0) {
#pragma unroll
      for (int z = 0; z < 3; ++z) c0 += *reinterpret_cast<const v4f_b*>(red + ((z * 4 + nt) * 64 + lane) * 4);
      // D^T layout: row = output column 16 nt + 4 lg + r, col = edge l15
      const int64_t e = e0 + l15;
      if (e < E) *reinterpret_cast<float4*>(out + e * 64 + 16 * nt + 4 * lg) = make_float4(alpha * c0[0], alpha * c0[1], alpha * c0[2], alpha * c0[3]);
    }
    return;
  }
  // K3: out[16 x 64] = Pl[16 x 1024] @ W2T^T; wave (kq, nt) owns one 16 x 16 tile over a quarter of K; the quarters meet
  // in LDS.  K-step (j, comp): lane group lg supplies k = 16 j + 4 lg + comp for both operands (float4 along k).
  const int kq = wave >> 2, nt = wave & 3;
  const float* __restrict__ wrow = W2T + (int64_t)(16 * nt + l15) * 1024 + 256 * kq + 4 * lg;
  const float* arow = Pl + l15 * FLDP + 256 * kq + 4 * lg;
  float* red = Pl + FTE * FLDP;   // [3 kq][4 nt][64 lanes][4]
  v4f_b c0 = (v4f_b){0.f, 0.f, 0.f, 0.f}, c1 = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int j = 0; j < 16; j += 2) {
    const float4 w0 = *reinterpret_cast<const float4*>(wrow + 16 * j);
    const float4 w1 = *reinterpret_cast<const float4*>(wrow + 16 * j + 16);
    const float4 p0 = *reinterpret_cast<const float4*>(arow + 16 * j);
    const float4 p1 = *reinterpret_cast<const float4*>(arow + 16 * j + 16);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.x, w0.x, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.x, w1.x, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.y, w0.y, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.y, w1.y, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.z, w0.z, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.z, w1.z, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.w, w0.w, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.w, w1.w, c1, 0, 0, 0);
  }
  c0 += c1;
  if (kq > 0) *reinterpret_cast<v4f_b*>(red + (((kq - 1) * 4 + nt) * 64 + lane) * 4) = c0;
  __syncthreads();
  if (kq == 0) {
#pragma unroll
    for (int z = 0; z < 3; ++z) c0 += *reinterpret_cast<const v4f_b*>(red + ((z * 4 + nt) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t e = e0 + 4 * lg + r;
      if (e < E) out[e * 64 + 16 * nt + l15] = alpha * c0[r];
    }
  }
}

// Adjoint of K2 fused with the K1 adjoint w.r.t. Y, one workgroup per reduce edge:
//   gB[e,s,i]  = sum_c Sm[e,s,c] dP[e,i,c]
//   dSm[e,s,c] = sum_i B[e,s,i] dP[e,i,c]                  (written: bil_reduce_t consumes it)
//   dY[t,s]    = sum_c dSm[e,s,c] x[g(t),c]   for t in seg(e)
// replacing two bmm launches and bil_dot.  LDS: dP[e] (I x (C+4)), Sm[e]/dSm[e] (S x (C+4)), B[e].
__global__ __launch_bounds__(256) void bil_project_bwd_kernel(
    const float* __restrict__ dP, const float* __restrict__ Sm, const float* __restrict__ B,
    const float* __restrict__ x, const int32_t* __restrict__ expand_idx, const int32_t* __restrict__ seg_off,
    float* __restrict__ gB, float* __restrict__ dSm, float* __restrict__ dY, int S, int C, int I, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ld = C + 4;
  float* dPl = sm;                 // [I][ld]
  float* Sl = dPl + I * ld;        // [S][ld]  Sm, later dSm
  float* Dl = Sl + S * ld;         // [S][ld]  dSm
  float* Bl = Dl + S * ld;         // [S*I]
  const int64_t e = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < I * C; i += nt) dPl[(i / C) * ld + (i % C)] = dP[e * (int64_t)I * C + i];
  for (int i = tid; i < S * C; i += nt) Sl[(i / C) * ld + (i % C)] = Sm[e * (int64_t)S * C + i];
  for (int i = tid; i < S * I; i += nt) Bl[i] = B[e * (int64_t)S * I + i];
  __syncthreads();
  // gB[s,i] = <Sm[s,:], dP[i,:]>
  for (int o = tid; o < S * I; o += nt) {
    const int s = o / I, i = o - s * I;
    const float* a = Sl + s * ld;
    const float* b = dPl + i * ld;
    float acc = 0.f;
    for (int c = 0; c < C; c += 4) {
      const float4 u = *reinterpret_cast<const float4*>(a + c);
      const float4 v = *reinterpret_cast<const float4*>(b + c);
      acc = fmaf(u.x, v.x, acc); acc = fmaf(u.y, v.y, acc); acc = fmaf(u.z, v.z, acc); acc = fmaf(u.w, v.w, acc);
    }
    float* go = gB + e * (int64_t)S * I + o;
    *go = (accumulate & 2) ? *go + acc : acc;     // bit 1: running gradient of the radial basis shared by the blocks
  }
  // dSm[s,c] = sum_i B[s,i] dP[i,c]
  for (int o = tid; o < S * C; o += nt) {
    const int s = o / C, c = o - s * C;
    float acc = 0.f;
    for (int i = 0; i < I; ++i) acc = fmaf(Bl[s * I + i], dPl[i * ld + c], acc);
    Dl[s * ld + c] = acc;
    dSm[e * (int64_t)S * C + o] = acc;
  }
  __syncthreads();
  if (!dY) return;
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  const int n = (t1 - t0) * S;
  for (int p = tid; p < n; p += nt) {
    const int tt = p / S, s = p - tt * S;
    const int t = t0 + tt;
    const float* __restrict__ xr = x + (int64_t)expand_idx[t] * C;
    const float* dr = Dl + s * ld;
    float acc = 0.f;
    for (int c = 0; c < C; c += 4) {
      const float4 a = *reinterpret_cast<const float4*>(dr + c);
      const float4 b = *reinterpret_cast<const float4*>(xr + c);
      acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
    }
    if (accumulate & 1) dY[(int64_t)t * S + s] += acc;
    else dY[(int64_t)t * S + s] = acc;
  }
}

// Spherical-basis (S = 7, C = 64, I = 16) form of the fused bilinear adjoint on the matrix cores, one wave per reduce
// edge (the scalar kernel above spends ~700 instructions per lane and edge: 91 us against 28 us of HBM traffic):
//   gB[s,i]  = sum_c Sm[s,c] dP[i,c]       (7/16 x 16 x 64)
//   dSm[s,c] = sum_i B[s,i]  dP[i,c]       (7/16 x 64 x 16)   -> global + LDS
//   dY[t,s]  = sum_c x[g(t),c] dSm[s,c]    (16 triplets per row tile x 7/16 x 64)
// Contiguous contraction indices are fetched as float4 and consumed component-wise (K-step (j, comp): lane group lg
// supplies k = 16 j + 4 lg + comp for both operands).
template <bool ACC>
__global__ __launch_bounds__(256) void bil_project_bwd_mfma7_kernel(
    const float* __restrict__ dP, const float* __restrict__ Sm, const float* __restrict__ B,
    const float* __restrict__ x, const int32_t* __restrict__ expand_idx, const int32_t* __restrict__ seg_off,
    float* __restrict__ gB, float* __restrict__ dSm, float* __restrict__ dY, int64_t E, int gb_acc) {
  constexpr int S = 7, C = 64, I = 16, LD = C + 4;
  const int dsm_acc = gb_acc & 2;     // bit 1 of the flag word: dSm += (the cross term dB mu_P of the second adjoint)
  gb_acc &= 1;
  __shared__ __attribute__((aligned(16))) float dsl[4][8][LD];   // dSm of this wave's edge (row 7: MFMA padding)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= E) return;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto comp = [](const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); };
  const float* __restrict__ dPe = dP + e * (int64_t)I * C;
  const float* __restrict__ Sme = Sm + e * (int64_t)S * C;
  const float* __restrict__ Be = B + e * (int64_t)S * I;
  const bool srow = l15 < S;
  const int t0 = dY ? seg_off[e] : 0, t1 = dY ? seg_off[e + 1] : 0;
  // Every load below is UNCONDITIONAL at a clamped (valid) address: `cond ? *p : 0` compiles to one branch per dword
  // and splits the float4 loads.  Rows that only exist as MFMA padding (s >= 7, t >= t1) may hold duplicates: each
  // padded row / column only feeds outputs that are never stored.
  const int tlast = t1 - 1, scl = min(l15, S - 1);
  auto loadx = [&](int tb, float4 (&xa)[4]) {   // x rows of the 16 triplets tb.., one per l15, k = c contiguous
    if (tb >= t1) return;                       // (wave-uniform)
    const float* __restrict__ xr = x + (int64_t)expand_idx[min(tb + l15, tlast)] * C + 4 * lg;
#pragma unroll
    for (int j = 0; j < 4; ++j) xa[j] = *reinterpret_cast<const float4*>(xr + 16 * j);
  };
  float4 smf[4], dpf[4], xa[4], xb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) xa[j] = xb[j] = z4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    smf[j] = *reinterpret_cast<const float4*>(Sme + scl * C + 16 * j + 4 * lg);
    dpf[j] = *reinterpret_cast<const float4*>(dPe + l15 * C + 16 * j + 4 * lg);
  }
  const float4 bf = *reinterpret_cast<const float4*>(Be + scl * I + 4 * lg);
  float dpk[4][4];   // dP[i = 4 lg + comp][c = 16 nt + l15]
#pragma unroll
  for (int cp = 0; cp < 4; ++cp)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) dpk[cp][nt] = dPe[(4 * lg + cp) * C + 16 * nt + l15];
  loadx(t0, xa), loadx(t0 + 16, xb);   // first two row tiles of the Y gradient in flight under the two products
  // ---- (1) gB = Sm dP^T
  v4f_b g = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) g = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(smf[j], cp), comp(dpf[j], cp), g, 0, 0, 0);
  float* __restrict__ gbo = gB + e * (int64_t)S * I;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * lg + r < S) gbo[(4 * lg + r) * I + l15] = gb_acc ? gbo[(4 * lg + r) * I + l15] + g[r] : g[r];
  // ---- (2) dSm = B dP
  v4f_b d[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) d[nt] = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cp = 0; cp < 4; ++cp)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(bf, cp), dpk[cp][nt], d[nt], 0, 0, 0);
  float* __restrict__ dso = dSm + e * (int64_t)S * C;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int srw = 4 * lg + r;
      if (dsm_acc && srw < S) d[nt][r] += dso[srw * C + 16 * nt + l15];
      if (srw < S) dso[srw * C + 16 * nt + l15] = d[nt][r];
      if (srw < 8) dsl[wave][srw][16 * nt + l15] = d[nt][r];   // row 7: padding, feeds the unstored column s = 7
    }
  if (!dY) return;
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes are visible to its own reads
  __builtin_amdgcn_wave_barrier();
  // ---- (3) dY = Xseg dSm^T
  float4 bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bs[j] = *reinterpret_cast<const float4*>(&dsl[wave][min(l15, 7)][16 * j + 4 * lg]);
  for (int tb = t0; tb < t1; tb += 16) {
    float4 xn[4] = {z4, z4, z4, z4};
    loadx(tb + 32, xn);
    v4f_b y = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) y = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(xa[j], cp), comp(bs[j], cp), y, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = tb + 4 * lg + r;
      if (srow && t < t1) {
        float* o = dY + (int64_t)t * S + l15;
        *o = ACC ? *o + y[r] : y[r];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) xa[j] = xb[j], xb[j] = xn[j];
  }
}

// Tensor-basis (S = 49, C = I = 32) form of the fused bilinear adjoint on the matrix cores, one wave per reduce edge:
//   gB[s,i]  = sum_c Sm[s,c] dP[i,c]       (64 x 32 x 32)
//   dSm[s,c] = sum_i B[s,i]  dP[i,c]       (64 x 32 x 32)     -> global + LDS
//   dY[t,s]  = sum_c x[g(t),c] dSm[s,c]    (K4 x 64 x 32, 16 quadruplets per MFMA row tile)
// Operands whose contraction index is contiguous are fetched as float4 and consumed component-wise: in K-step
// (j, comp) lane group lg supplies k = 16 j + 4 lg + comp for BOTH operands, so the sum over k is complete.
template <bool ACC>
__global__ __launch_bounds__(256) void bil_project_bwd_mfma49_kernel(
    const float* __restrict__ dP, const float* __restrict__ Sm, const float* __restrict__ B,
    const float* __restrict__ x, const int32_t* __restrict__ expand_idx, const int32_t* __restrict__ seg_off,
    float* __restrict__ gB, float* __restrict__ dSm, float* __restrict__ dY, int64_t E, int gb_acc) {
  constexpr int S = 49, C = 32, I = 32, LD = C + 4;
  __shared__ __attribute__((aligned(16))) float dsl[4][64][LD];   // dSm of this wave's edge (rows >= 49 zero)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= E) return;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto comp = [](const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); };
  const float* __restrict__ dPe = dP + e * (int64_t)I * C;
  const float* __restrict__ Sme = Sm + e * (int64_t)S * C;
  const float* __restrict__ Be = B + e * (int64_t)S * I;
  // ---- (1) gB and (2) dSm share the row-fragments of dP
  float4 dprow[2][2];   // dP[16 nt + l15][16 j + 4 lg ..]   (k = c contiguous)
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      dprow[nt][j] = *reinterpret_cast<const float4*>(dPe + (16 * nt + l15) * C + 16 * j + 4 * lg);
  v4f_b acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int srow = 16 * mt + l15;
    const bool ok = srow < S;
    float4 a[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {   // unconditional at a clamped row, then masked (a load under `?:` is split per dword)
      const float4 v = *reinterpret_cast<const float4*>(Sme + min(srow, S - 1) * C + 16 * j + 4 * lg);
      a[j] = ok ? v : z4;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      v4f_b c4 = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(a[j], q), comp(dprow[nt][j], q), c4, 0, 0, 0);
      acc[mt][nt] = c4;
    }
  }
  float* __restrict__ gBe = gB + e * (int64_t)S * I;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int srow = 16 * mt + 4 * lg + r;
        if (srow < S) gBe[srow * I + 16 * nt + l15] = gb_acc ? gBe[srow * I + 16 * nt + l15] + acc[mt][nt][r] : acc[mt][nt][r];
      }
  // (2) dSm[s,c] = sum_i B[s,i] dP[i,c]: A rows of B (k = i contiguous), Bop[k = i][n = c] = dP[i][c] (scalar loads)
  float dpcol[2][2][4];   // dP[16 j + 4 lg + q][16 nt + l15]
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) dpcol[nt][j][q] = dPe[(16 * j + 4 * lg + q) * C + 16 * nt + l15];
  float* __restrict__ dSe = dSm + e * (int64_t)S * C;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int srow = 16 * mt + l15;
    const bool ok = srow < S;
    float4 a[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(Be + min(srow, S - 1) * I + 16 * j + 4 * lg);
      a[j] = ok ? v : z4;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      v4f_b c4 = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(a[j], q), dpcol[nt][j][q], c4, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int sr = 16 * mt + 4 * lg + r;
        const float v = sr < S ? c4[r] : 0.f;
        if (sr < S) dSe[sr * C + 16 * nt + l15] = v;
        dsl[wave][sr][16 * nt + l15] = v;
      }
    }
  }
  if (!dY) return;   // deferred: the caller sums the Y gradient of several blocks in gn_bil_dy_multi_f32
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  // ---- (3) dY[t,s] = sum_c x[g(t),c] dSm[s,c]: Bop fragments (rows of dSm, k = c contiguous) once per edge
  float4 bd[4][2];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      bd[nt][j] = *reinterpret_cast<const float4*>(&dsl[wave][16 * nt + l15][16 * j + 4 * lg]);
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  auto loadx = [&](int tb, float4 (&ax)[2]) {
    const int tq = tb + l15;
    if (tq < t1) {
      const float* __restrict__ xr = x + (int64_t)expand_idx[tq] * C + 4 * lg;
      ax[0] = *reinterpret_cast<const float4*>(xr);
      ax[1] = *reinterpret_cast<const float4*>(xr + 16);
    } else {
      ax[0] = z4; ax[1] = z4;
    }
  };
  float4 ax[2], an[2];
  loadx(t0, ax);
  for (int tb = t0; tb < t1; tb += 16) {
    loadx(tb + 16, an);
    v4f_b y4[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      v4f_b c4 = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(ax[j], q), comp(bd[nt][j], q), c4, 0, 0, 0);
      y4[nt] = c4;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tq = tb + 4 * lg + r;
      if (tq < t1) {
        float* __restrict__ yo = dY + (int64_t)tq * S + l15;
        if (ACC) {   // dY is the running sum over the interaction blocks that share this basis
          yo[0] += y4[0][r];
          yo[16] += y4[1][r];
          yo[32] += y4[2][r];
          if (l15 == 0) yo[48] += y4[3][r];
        } else {
          yo[0] = y4[0][r];
          yo[16] = y4[1][r];
          yo[32] = y4[2][r];
          if (l15 == 0) yo[48] = y4[3][r];
        }
      }
    }
    ax[0] = an[0]; ax[1] = an[1];
  }
}

// Matrix-core form of bil_expand for the tensor basis (S = 49, C = 32): per reduce edge
//   dxt[seg(e)] (K4 x C) = Yseg (K4 x S) @ dSm[e] (S x C),
// one wave per edge, 16 quadruplets per MFMA row tile, the 26 B-operand fragments of dSm[e] held in registers.
__global__ __launch_bounds__(256) void bil_expand_mfma49_kernel(const float* __restrict__ Y,
                                                                const float* __restrict__ dSm,
                                                                const int32_t* __restrict__ seg_off,
                                                                float* __restrict__ dxt, int64_t E) {
  constexpr int S = 49, C = 32, TT = 16, LDY = 53;   // 16 quadruplets per row tile; LDS row pitch 53 (odd: no conflicts)
  __shared__ float ysm[4][2][TT * LDY];               // per wave, double buffered
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= E) return;
  const float* __restrict__ De = dSm + e * (int64_t)S * C;
  float bd[13][2];   // dSm[4 kk + lg][16 nt + l15]
#pragma unroll
  for (int kk = 0; kk < 13; ++kk) {
    const int sr = 4 * kk + lg;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {   // unconditional at a clamped row, then masked (rows >= 49 pad the contraction)
      const float v = De[min(sr, S - 1) * C + 16 * nt + l15];
      bd[kk][nt] = sr < S ? v : 0.f;
    }
  }
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  // the 16 x 49 block of Y of a row tile is one contiguous run of 784 floats: 13 coalesced loads per lane, parked in
  // this wave's LDS buffer in [t][s] layout, then read back as MFMA A fragments (lane (l15, lg) <- Y[t = l15][4 kk + lg])
  float st[13];
  auto fetch = [&](int tb) {
    const int n = (t1 - tb < TT ? t1 - tb : TT) * S;
    const float* __restrict__ src = Y + (int64_t)tb * S;
#pragma unroll
    for (int j = 0; j < 13; ++j) st[j] = src[min(lane + 64 * j, n - 1)];   // rows past t1: duplicates, never stored
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      const int i = lane + 64 * j;
      if (i < TT * S) {
        const int r = i / S;
        ysm[wave][buf][r * LDY + (i - r * S)] = st[j];
      }
    }
  };
  int buf = 0;
  if (t0 < t1) { fetch(t0); park(0); }
  for (int tb = t0; tb < t1; tb += TT) {
    const bool more = tb + TT < t1;
    if (more) fetch(tb + TT);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // this wave's parked rows are visible to its own reads
    __builtin_amdgcn_wave_barrier();
    const float* __restrict__ yb = ysm[wave][buf] + l15 * LDY + lg;
    v4f_b c0 = (v4f_b){0.f, 0.f, 0.f, 0.f}, c1 = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 13; ++kk) {
      const float yv = yb[4 * kk];
      const float a = (kk < 12 || lg == 0) ? yv : 0.f;
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bd[kk][0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bd[kk][1], c1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tq = tb + 4 * lg + r;
      if (tq < t1) {
        float* __restrict__ o = dxt + (int64_t)tq * C + l15;
        o[0] = c0[r];
        o[16] = c1[r];
      }
    }
    if (more) park(buf ^ 1);
    buf ^= 1;
  }
}

// dY[t,s] = sum_b sum_c x_b[g(t),c] dSm_b[r(t),s,c] over the nb <= 4 interaction blocks that share one tensor basis:
// written ONCE instead of written by the first block and read-modify-written by every further one (the (Q,49) array is
// 1.8 GB at B = 32).  One wave per reduce edge; the B fragments of all nb blocks stay in registers.
struct gn_dy_multi_args {
  const float* dS[4];
  const float* x[4];
  int nb;
};

__global__ __launch_bounds__(256) void bil_dy_multi_mfma49_kernel(const gn_dy_multi_args a,
                                                                  const int32_t* __restrict__ expand_idx,
                                                                  const int32_t* __restrict__ seg_off,
                                                                  float* __restrict__ dY, int64_t E) {
  constexpr int S = 49, C = 32;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= E) return;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto comp = [](const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); };
  const int nb = a.nb;
  float4 bd[4][4][2];   // [block][s tile][j]: dSm_b[e][16 nt + l15][16 j + 4 lg ..]
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int sr = 16 * nt + l15;
        bd[b][nt][j] = z4;
        if (b < nb) {   // (uniform)  unconditional load at a clamped row, then masked
          const float4 v = *reinterpret_cast<const float4*>(a.dS[b] + (e * S + min(sr, S - 1)) * C + 16 * j + 4 * lg);
          bd[b][nt][j] = sr < S ? v : z4;
        }
      }
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  for (int tb = t0; tb < t1; tb += 16) {
    const int tq = tb + l15;
    const bool ok = tq < t1;
    const int64_t g = ok ? expand_idx[tq] : 0;
    float4 ax[4][2];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b < nb && ok) {
        const float* __restrict__ xr = a.x[b] + g * C + 4 * lg;
        ax[b][0] = *reinterpret_cast<const float4*>(xr);
        ax[b][1] = *reinterpret_cast<const float4*>(xr + 16);
      } else {
        ax[b][0] = z4; ax[b][1] = z4;
      }
    }
    v4f_b y4[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      v4f_b c4 = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (b < nb) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(ax[b][j], q), comp(bd[b][nt][j], q), c4, 0, 0, 0);
        }
      }
      y4[nt] = c4;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tr = tb + 4 * lg + r;
      if (tr < t1) {
        float* __restrict__ yo = dY + (int64_t)tr * S + l15;
        yo[0] = y4[0][r];
        yo[16] = y4[1][r];
        yo[32] = y4[2][r];
        if (l15 == 0) yo[48] = y4[3][r];
      }
    }
  }
}

// The same one-pass Y gradient for the spherical basis (S = 7, C = 64: the triplet branch of all nb blocks shares
// `sph`): dY[t,s] = sum_b sum_c x_b[g(t),c] dSm_b[e,s,c], written once instead of written by the first block and
// read-modify-written by each further one.  One wave per reduce edge, 16 triplets per MFMA row tile; the x rows of
// block b + 1 are in flight under the 16 MFMAs of block b.
__global__ __launch_bounds__(256) void bil_dy_multi_mfma7_kernel(const gn_dy_multi_args a,
                                                                 const int32_t* __restrict__ expand_idx,
                                                                 const int32_t* __restrict__ seg_off,
                                                                 float* __restrict__ dY, int64_t E) {
  constexpr int S = 7, C = 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= E) return;
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  if (t0 >= t1) return;
  auto comp = [](const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); };
  const int nb = a.nb;
  const int scl = min(l15, S - 1);   // columns s >= 7 are MFMA padding: duplicates, never stored
  float4 bd[4][4];                   // [block][j]: dSm_b[e][s = l15][16 j + 4 lg ..]
#pragma unroll
  for (int b = 0; b < 4; ++b)
    if (b < nb) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bd[b][j] = *reinterpret_cast<const float4*>(a.dS[b] + (e * S + scl) * C + 16 * j + 4 * lg);
    }
  auto loadx = [&](int b, int64_t g, float4 (&ax)[4]) {
    if (b >= nb) return;
    const float* __restrict__ xr = a.x[b] + g * C + 4 * lg;
#pragma unroll
    for (int j = 0; j < 4; ++j) ax[j] = *reinterpret_cast<const float4*>(xr + 16 * j);
  };
  for (int tb = t0; tb < t1; tb += 16) {
    const int64_t g = expand_idx[min(tb + l15, t1 - 1)];   // rows t >= t1: duplicates, never stored
    float4 ax[4], an[4];
    loadx(0, g, ax);
    v4f_b y = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b < nb) {
        loadx(b + 1, g, an);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            y = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(ax[j], q), comp(bd[b][j], q), y, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) ax[j] = an[j];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = tb + 4 * lg + r;
      if (l15 < S && t < t1) dY[(int64_t)t * S + l15] = y[r];
    }
  }
}

inline bool ok_channels(int C) { return C > 0 && C <= 256 && (256 % C) == 0; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int gn_bil_reduce_f32(const float* Y, const float* x, const int32_t* expand_idx,
                                 const int32_t* seg_off, float* Sm, int64_t E, int S, int C,
                                 void* stream) {
  if (E <= 0) return 0;
  if (!ok_channels(C)) return (int)hipErrorInvalidValue;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int epb = 256 / C;
  dim3 grid(gn_cdiv(E, epb)), block(256);
  if (S == 7) {
    hipLaunchKernelGGL(bil_reduce_kernel<7>, grid, block, 0, st, Y, x, expand_idx, seg_off, Sm, E, C);
  } else if (S == 49) {
    hipLaunchKernelGGL(bil_reduce_kernel<49>, grid, block, 0, st, Y, x, expand_idx, seg_off, Sm, E, C);
  } else {
    return (int)hipErrorInvalidValue;
  }
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_reduce_t_f32(const float* Y, const float* dSm, const int32_t* reduce_idx,
                                   const int32_t* permT, const int32_t* segT_off, float* dx,
                                   int64_t J, int S, int C, void* stream) {
  if (J <= 0) return 0;
  if (!ok_channels(C)) return (int)hipErrorInvalidValue;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rpb = 256 / C;
  dim3 grid(gn_cdiv(J, rpb)), block(256);
  if (S == 7) {
    hipLaunchKernelGGL(bil_reduce_t_kernel<7>, grid, block, 0, st, Y, dSm, reduce_idx, permT, segT_off, dx, J, C);
  } else if (S == 49) {
    hipLaunchKernelGGL(bil_reduce_t_kernel<49>, grid, block, 0, st, Y, dSm, reduce_idx, permT, segT_off, dx, J, C);
  } else {
    return (int)hipErrorInvalidValue;
  }
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_reduce_t_grouped_f32(const float* Y, const float* dSm, const int32_t* grp_rows,
                                           const int32_t* grp_off, const int32_t* grp_kseg, const int32_t* permT,
                                           const int32_t* rposT, float* dx, int64_t G, int max_rows, int S, int C,
                                           void* stream) {
  if (G <= 0) return 0;
  if (C != 64 || S != 7 || max_rows < 1) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)max_rows * S * C * sizeof(float);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  hipStream_t st = static_cast<hipStream_t>(stream);
  static size_t lds_max = 0;
  if (lds > 64 * 1024 && lds > lds_max) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bil_reduce_t_grouped_kernel<7>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    lds_max = 160 * 1024;
  }
  hipLaunchKernelGGL(bil_reduce_t_grouped_kernel<7>, dim3((unsigned)G), dim3(1024), lds, st, Y, dSm, grp_rows, grp_off,
                     reinterpret_cast<const int2*>(grp_kseg), permT, rposT, dx, max_rows);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_expand_f32(const float* Y, const float* dSm, const int32_t* seg_off, float* dxt, int64_t E, int S,
                                 int C, void* stream) {
  if (E <= 0) return 0;
  if (!ok_channels(C)) return (int)hipErrorInvalidValue;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int gpb = 256 / C;
  dim3 grid(gn_cdiv(E, gpb)), block(256);
  if (S == 49 && C == 32) {
    hipLaunchKernelGGL(bil_expand_mfma49_kernel, dim3(gn_cdiv(E, 4)), dim3(256), 0, st, Y, dSm, seg_off, dxt, E);
  } else if (S == 49) {
    constexpr int CH = 32;
    hipLaunchKernelGGL((bil_expand_kernel<49, CH>), grid, block, (size_t)gpb * CH * 52 * sizeof(float), st, Y, dSm,
                       seg_off, dxt, E, C);
  } else if (S == 7) {
    constexpr int CH = 32;
    hipLaunchKernelGGL((bil_expand_kernel<7, CH>), grid, block, (size_t)gpb * CH * 8 * sizeof(float), st, Y, dSm,
                       seg_off, dxt, E, C);
  } else {
    return (int)hipErrorInvalidValue;
  }
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_dot_f32(const float* dSm, const float* x, const int32_t* expand_idx,
                              const int32_t* seg_off, float* dY, int64_t E, int S, int C,
                              void* stream) {
  if (E <= 0) return 0;
  if (S <= 0 || C <= 0) return (int)hipErrorInvalidValue;
  const size_t smem = (size_t)S * (C + 4) * sizeof(float);
  if (smem > 64 * 1024) return (int)hipErrorInvalidValue;
  const int vec = (C % 4 == 0) && aligned16(x);
  hipLaunchKernelGGL(bil_dot_kernel, dim3((unsigned)E), dim3(S <= 7 ? 128 : 256), smem,
                     static_cast<hipStream_t>(stream), dSm, x, expand_idx, seg_off, dY, S, C, vec);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_reduce_project_f32(const float* Y, const float* x, const int32_t* expand_idx,
                                         const int32_t* seg_off, const float* B, float* Sm, float* P,
                                         int64_t E, int S, int C, int I, void* stream) {
  if (E <= 0) return 0;
  if (!ok_channels(C) || I <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int epb = 256 / C;
  const size_t smem = (size_t)epb * S * I * sizeof(float);
  if (smem > 64 * 1024) return (int)hipErrorInvalidValue;
  dim3 grid(gn_cdiv(E, epb)), block(256);
  if (S == 7 && C == 64 && I == 16) {
    hipLaunchKernelGGL(bil_reduce_project_mfma7_kernel<false>, dim3(gn_cdiv(E, 4)), dim3(256), 0, st, Y, x, expand_idx, seg_off,
                       B, Sm, P, E, nullptr, nullptr, nullptr);
  } else if (S == 7) {
    hipLaunchKernelGGL(bil_reduce_project_kernel<7>, grid, block, smem, st, Y, x, expand_idx, seg_off, B, Sm, P, E, C, I);
  } else if (S == 49 && C == 32 && I == 32) {
    hipLaunchKernelGGL(bil_reduce_project_mfma49_kernel, dim3(gn_cdiv(E, 4)), dim3(256), 0, st, Y, x, expand_idx, seg_off, B,
                       Sm, P, E);
  } else if (S == 49) {
    hipLaunchKernelGGL(bil_reduce_project_kernel<49>, grid, block, smem, st, Y, x, expand_idx, seg_off, B, Sm, P, E, C, I);
  } else {
    return (int)hipErrorInvalidValue;
  }
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_reduce_project2_f32(const float* Y, const float* x, const int32_t* expand_idx, const int32_t* seg_off,
                                          const float* B, const float* Sm_init, const float* B2, const float* Sm2, float* Sm,
                                          float* P, int64_t E, int S, int C, int I, void* stream) {
  if (E <= 0) return 0;
  if (!(S == 7 && C == 64 && I == 16) || ((B2 == nullptr) != (Sm2 == nullptr))) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(bil_reduce_project_mfma7_kernel<true>, dim3(gn_cdiv(E, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     Y, x, expand_idx, seg_off, B, Sm, P, E, Sm_init, B2, Sm2);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_fused_fwd_f32(const float* Y, const float* x, const int32_t* expand_idx, const int32_t* seg_off,
                                    const float* B, const float* W2T, const void* W2T_planes, float* Sm, float* out, int64_t E,
                                    int S, int C, int I, int O, float alpha, void* stream) {
  if (E <= 0) return 0;
  if (S != 7 || C != 64 || I != 16 || O != 64 || !aligned16(W2T) || !aligned16(W2T_planes)) return (int)hipErrorInvalidValue;
  const size_t lds_f = ((size_t)FTE * FLDP + 3 * 4 * 64 * 4) * sizeof(float);
  const size_t lds_h = (size_t)2 * FTE * FPH * 2 + (size_t)3 * 4 * 64 * 4 * sizeof(float);
  const size_t lds = W2T_planes ? lds_h : lds_f;
  static std::atomic<bool> configured{false};   // set-once flag of an idempotent attribute (two racing threads both set it)
  if (!configured.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bil_fused_fwd_mfma7_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bil_fused_fwd_mfma7_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);
    if (e != hipSuccess) return (int)e;
    configured.store(true, std::memory_order_release);
  }
  if (W2T_planes)
    hipLaunchKernelGGL(bil_fused_fwd_mfma7_kernel<true>, dim3(gn_cdiv(E, FTE)), dim3(1024), lds, static_cast<hipStream_t>(stream),
                       Y, x, expand_idx, seg_off, B, W2T, static_cast<const uint4*>(W2T_planes), Sm, out, E, alpha);
  else
    hipLaunchKernelGGL(bil_fused_fwd_mfma7_kernel<false>, dim3(gn_cdiv(E, FTE)), dim3(1024), lds, static_cast<hipStream_t>(stream),
                       Y, x, expand_idx, seg_off, B, W2T, nullptr, Sm, out, E, alpha);
  GN_LAUNCH_CHECK();
  return 0;
}

// Adjoint of the bilinear tail in ONE launch for the spherical basis (S = 7, C = 64, I = 16, O = 64), Y gradient
// deferred (gn_bil_dy_multi_f32):
//   dP[e, k] = alpha * sum_o g[e,o] W2[k,o]      16-edge tile x 1024, K = 64: stays in LDS (never the 74 MB in HBM that the
//                                                K = 64 / N = 1024 GEMM wrote and bil_project_bwd_mfma7 re-read)
//   gB[e]    = Sm[e] dP[e]^T,   dSm[e] = B[e] dP[e]      one wave per edge, as bil_project_bwd_mfma7_kernel
// W2 = the bilinear weight as (I*C, O), o contiguous.  Same products in the same order as the two-launch form: the
// results are bit-identical (tests/test_gpu_kernels.py).
namespace {
// HP: phase 1 on the fp16 matrix pipe with split operands, computed transposed (A = 16 rows of the pre-split weight W2,
// gn_pack_weight_split_fmt(W2, 1024, 64, GN_SPLIT_F16X2); B = the g rows of the 16 edges): 24 v_mfma_f32_16x16x32_f16 per wave
// instead of 64 f32 MFMAs of twice the length.  g is a COTANGENT: every edge's row gets one exact power-of-two scale (row
// maximum -> [0.25, 0.5)) before the split, and the lane that ends up with four consecutive k of that edge multiplies it back.
template <bool HP>
__global__ __launch_bounds__(1024) void bil_fused_bwd_mfma7_kernel(const float* __restrict__ g, const float* __restrict__ W2,
                                                      const uint4* __restrict__ W2p,
                                                      const float* __restrict__ Sm, const float* __restrict__ B,
                                                      float* __restrict__ gB, float* __restrict__ dSm, int64_t E,
                                                      float alpha, int gb_acc) {
  constexpr int S = 7, C = 64, I = 16, TE = 16, LDP = 1024 + 4;
  extern __shared__ __attribute__((aligned(16))) float dPl[];   // [TE][LDP]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e0 = (int64_t)blockIdx.x * TE;
  auto comp = [](const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); };
  // ---- phase 1: dP tile = g tile (16 x 64) @ W2^T (64 x 1024).  Wave w owns the 64 columns k = 64 w .. 64 w + 63
  // (= i = w, all c) as four 16 x 16 tiles; K-step (j, comp): lane group lg supplies o = 16 j + 4 lg + comp.
  if constexpr (HP) {
    const int64_t er = min(e0 + l15, E - 1);   // rows past E: duplicates, never used
    // B operand: g[e = l15][o = 32 c + 8 lg + i]
    float gv[2][8];
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float* __restrict__ gr = g + er * 64 + 32 * c + 8 * lg;
      const float4 u0 = *reinterpret_cast<const float4*>(gr), u1 = *reinterpret_cast<const float4*>(gr + 4);
      gv[c][0] = u0.x; gv[c][1] = u0.y; gv[c][2] = u0.z; gv[c][3] = u0.w;
      gv[c][4] = u1.x; gv[c][5] = u1.y; gv[c][6] = u1.z; gv[c][7] = u1.w;
#pragma unroll
      for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf(gv[c][i]));
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));       // the four lanes (lg) that hold the row of edge l15
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sigma = 1.f, inv_sigma = 1.f;
    {
      const uint32_t ex = __float_as_uint(m) >> 23;
      const uint32_t ec = ex < 2u ? 2u : (ex > 250u ? 250u : ex);
      if (m > 0.f) {
        sigma = __uint_as_float((252u - ec) << 23);
        inv_sigma = __uint_as_float((2u + ec) << 23);
      }
    }
    h8_b bh[2], bl[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = gv[c][i] * sigma;
        const _Float16 h = (_Float16)v;
        bh[c][i] = h;
        bl[c][i] = (_Float16)((v - (float)h) * 2048.f);
      }
    const float post = alpha * inv_sigma;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const uint4* __restrict__ wp = W2p + ((size_t)(4 * wave + nt) * 2 * 2) * 64 + lane;   // [k tile][chunk][plane][lane]
      v4f_b ch = (v4f_b){0.f, 0.f, 0.f, 0.f}, cx = ch;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const h8_b ah = __builtin_bit_cast(h8_b, wp[(2 * c) * 64]);
        const h8_b al = __builtin_bit_cast(h8_b, wp[(2 * c + 1) * 64]);
        ch = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[c], ch, 0, 0, 0);
        cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[c], cx, 0, 0, 0);
        cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[c], cx, 0, 0, 0);
      }
      // D^T: row = k = 64 w + 16 nt + 4 lg + r, col = edge l15
      const v4f_b d = (ch + cx * (1.f / 2048.f)) * post;
      *reinterpret_cast<float4*>(dPl + l15 * LDP + 64 * wave + 16 * nt + 4 * lg) = make_float4(d[0], d[1], d[2], d[3]);
    }
  } else {
    const int64_t er = min(e0 + l15, E - 1);   // rows past E: duplicates, never used
    float4 ga[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ga[j] = *reinterpret_cast<const float4*>(g + er * 64 + 16 * j + 4 * lg);
    v4f_b acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const float* __restrict__ wr = W2 + (int64_t)(64 * wave + 16 * nt + l15) * 64 + 4 * lg;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 wb = *reinterpret_cast<const float4*>(wr + 16 * j);
#pragma unroll
        for (int cp = 0; cp < 4; ++cp)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(ga[j], cp), comp(wb, cp), acc[nt], 0, 0, 0);
      }
    }
    // D: row = edge 4 lg + r, col = k = 64 w + 16 nt + l15
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dPl[(4 * lg + r) * LDP + 64 * wave + 16 * nt + l15] = alpha * acc[nt][r];
  }
  __syncthreads();
  // ---- phase 2: one wave per edge, dP[e] (16 x 64, k = i * 64 + c) read from the LDS tile
  const int64_t e = e0 + wave;
  if (e >= E) return;
  const float* dPe = dPl + wave * LDP;
  const float* __restrict__ Sme = Sm + e * (int64_t)S * C;
  const float* __restrict__ Be = B + e * (int64_t)S * I;
  const int scl = min(l15, S - 1);
  float4 smf[4], dpf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    smf[j] = *reinterpret_cast<const float4*>(Sme + scl * C + 16 * j + 4 * lg);
    dpf[j] = *reinterpret_cast<const float4*>(dPe + l15 * C + 16 * j + 4 * lg);
  }
  const float4 bf = *reinterpret_cast<const float4*>(Be + scl * I + 4 * lg);
  float dpk[4][4];   // dP[i = 4 lg + comp][c = 16 nt + l15]
#pragma unroll
  for (int cp = 0; cp < 4; ++cp)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) dpk[cp][nt] = dPe[(4 * lg + cp) * C + 16 * nt + l15];
  v4f_b gb = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) gb = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(smf[j], cp), comp(dpf[j], cp), gb, 0, 0, 0);
  float* __restrict__ gbo = gB + e * (int64_t)S * I;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * lg + r < S) gbo[(4 * lg + r) * I + l15] = gb_acc ? gbo[(4 * lg + r) * I + l15] + gb[r] : gb[r];
  v4f_b d[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) d[nt] = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cp = 0; cp < 4; ++cp)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(bf, cp), dpk[cp][nt], d[nt], 0, 0, 0);
  float* __restrict__ dso = dSm + e * (int64_t)S * C;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * lg + r < S) dso[(4 * lg + r) * C + 16 * nt + l15] = d[nt][r];
}

}  // namespace

extern "C" int gn_bil_fused_bwd_f32(const float* g, const float* W2, const void* W2_planes, const float* Sm, const float* B,
                                    float* gB, float* dSm, int64_t E, int S, int C, int I, int O, float alpha, int accumulate,
                                    void* stream) {
  if (E <= 0) return 0;
  if (S != 7 || C != 64 || I != 16 || O != 64) return (int)hipErrorInvalidValue;
  if (!aligned16(g) || !aligned16(W2) || !aligned16(Sm) || !aligned16(B)) return (int)hipErrorInvalidValue;
  if (!aligned16(W2_planes)) return (int)hipErrorInvalidValue;
  constexpr size_t lds = (size_t)16 * (1024 + 4) * sizeof(float);
  static std::atomic<bool> configured{false};   // set-once flag of an idempotent attribute (two racing threads both set it)
  if (!configured.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bil_fused_bwd_mfma7_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bil_fused_bwd_mfma7_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    configured.store(true, std::memory_order_release);
  }
  if (W2_planes)
    hipLaunchKernelGGL(bil_fused_bwd_mfma7_kernel<true>, dim3((unsigned)gn_cdiv(E, 16)), dim3(1024), lds,
                       static_cast<hipStream_t>(stream), g, W2, static_cast<const uint4*>(W2_planes), Sm, B, gB, dSm, E, alpha,
                       (accumulate >> 1) & 1);
  else
    hipLaunchKernelGGL(bil_fused_bwd_mfma7_kernel<false>, dim3((unsigned)gn_cdiv(E, 16)), dim3(1024), lds,
                       static_cast<hipStream_t>(stream), g, W2, nullptr, Sm, B, gB, dSm, E, alpha, (accumulate >> 1) & 1);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_project_bwd_acc_f32(const float* dP, const float* Sm, const float* B, const float* x,
                                          const int32_t* expand_idx, const int32_t* seg_off, float* gB, float* dSm,
                                          float* dY, int64_t E, int S, int C, int I, int accumulate, void* stream) {
  if (E <= 0) return 0;
  if (S <= 0 || C <= 0 || (C % 4) != 0 || I <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int gb_acc = (accumulate >> 1) & 1;
  // bit 2: dSm += (spherical-basis kernel only; anything else must fail loudly, never drop the accumulation)
  if ((accumulate & 4) && !(S == 7 && C == 64 && I == 16 && aligned16(dP) && aligned16(Sm) && aligned16(B) && aligned16(x)))
    return (int)hipErrorInvalidValue;
  if (S == 49 && C == 32 && I == 32 && aligned16(dP) && aligned16(Sm) && aligned16(B) && aligned16(x)) {
    if (accumulate & 1)
      hipLaunchKernelGGL(bil_project_bwd_mfma49_kernel<true>, dim3(gn_cdiv(E, 4)), dim3(256), 0, st, dP, Sm, B, x,
                         expand_idx, seg_off, gB, dSm, dY, E, gb_acc);
    else
      hipLaunchKernelGGL(bil_project_bwd_mfma49_kernel<false>, dim3(gn_cdiv(E, 4)), dim3(256), 0, st, dP, Sm, B, x,
                         expand_idx, seg_off, gB, dSm, dY, E, gb_acc);
    GN_LAUNCH_CHECK();
    return 0;
  }
  if (S == 7 && C == 64 && I == 16 && aligned16(dP) && aligned16(Sm) && aligned16(B) && aligned16(x)) {
    const dim3 grid(gn_cdiv(E, 4));
    const int acc7 = gb_acc | (((accumulate >> 2) & 1) << 1);
    if (accumulate & 1)
      hipLaunchKernelGGL(bil_project_bwd_mfma7_kernel<true>, grid, dim3(256), 0, st, dP, Sm, B, x, expand_idx, seg_off, gB,
                         dSm, dY, E, acc7);
    else
      hipLaunchKernelGGL(bil_project_bwd_mfma7_kernel<false>, grid, dim3(256), 0, st, dP, Sm, B, x, expand_idx, seg_off, gB,
                         dSm, dY, E, acc7);
    GN_LAUNCH_CHECK();
    return 0;
  }
  const size_t smem = ((size_t)(I + 2 * S) * (C + 4) + (size_t)S * I) * sizeof(float);
  if (smem > 64 * 1024) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(bil_project_bwd_kernel, dim3((unsigned)E), dim3(128), smem, st, dP, Sm, B, x, expand_idx, seg_off,
                     gB, dSm, dY, S, C, I, accumulate);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_dy_multi_f32(const float* const* dSm_list, const float* const* x_list, int nb,
                                   const int32_t* expand_idx, const int32_t* seg_off, float* dY, int64_t E, int S, int C,
                                   void* stream) {
  if (E <= 0 || nb <= 0) return 0;
  if (nb > 4 || !((S == 49 && C == 32) || (S == 7 && C == 64))) return (int)hipErrorInvalidValue;
  gn_dy_multi_args a;
  a.nb = nb;
  for (int b = 0; b < 4; ++b) {
    a.dS[b] = b < nb ? dSm_list[b] : nullptr;
    a.x[b] = b < nb ? x_list[b] : nullptr;
    if (b < nb && (!aligned16(a.dS[b]) || !aligned16(a.x[b]))) return (int)hipErrorInvalidValue;
  }
  if (S == 7)
    hipLaunchKernelGGL(bil_dy_multi_mfma7_kernel, dim3(gn_cdiv(E, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                       expand_idx, seg_off, dY, E);
  else
    hipLaunchKernelGGL(bil_dy_multi_mfma49_kernel, dim3(gn_cdiv(E, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                       expand_idx, seg_off, dY, E);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_project_bwd_f32(const float* dP, const float* Sm, const float* B, const float* x,
                                      const int32_t* expand_idx, const int32_t* seg_off, float* gB, float* dSm,
                                      float* dY, int64_t E, int S, int C, int I, void* stream) {
  return gn_bil_project_bwd_acc_f32(dP, Sm, B, x, expand_idx, seg_off, gB, dSm, dY, E, S, C, I, 0, stream);
}
