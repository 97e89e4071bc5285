// Tensor-basis (GemNet-Q, S = 49, C = I = 32) bilinear kernels in ANGLE form (include/gemnet_hip.h, gn_bil_*_ang_f32).
//
// The reference materialises the real spherical harmonics of every quadruplet, Y (Q, 49) (basis_layers.py:239-295),
// and every interaction block reads them again (efficient.py:173-177 and its autograd): at 9 M quadruplets that is
// 1.76 GB per pass, 11 passes per forward+force step — 7 of 17 ms on MI355X (profiles/r1_final_*).  Here the per-quadruplet
// input is 16 B: (sin, cos) of the polar angle Phi_cab and of the azimuth Theta_cabd (csrc/geometry.hip,
// quad_angles_fwd_kernel).  Each wave rebuilds the Y_lm rows of 64 (resp. 16) quadruplets in its private LDS tile
// — one lane per quadruplet, the shared recurrences of basis_math.h — and feeds the MFMA fragments from there:
//   reduce_project_ang   Sm[e] = Yseg^T Xseg (K1) + P = B[e]^T Sm (K2)      replaces bil_reduce_project_mfma49
//   expand_ang           dxt[seg(e)] = Yseg dSm[e]                          replaces bil_expand_mfma49
//   dy_multi_ang         g_ang[t] = sum_s dY[t,s] dY_s/d(angles),  dY[t,s] = sum_b x_b[g(t)] . dSm_b[r(t),s]
//                        — the (Q, 49) gradient array is never written either (was: written once, read back by the
//                        geometry adjoint)
// Numerics: rows by the fully unrolled f32 recurrences of basis_math.h (ylm7_row_T<float>: ~200 FMAs per row; the f64
// jet visitors of the geometry kernels cost ~10 k cycles per 64 rows and made these kernels compute-bound), within 2e-6
// of the f64 rows (tests/test_host_math.py).
#include "common.h"
#include "basis_math.h"

typedef float v4f_a __attribute__((ext_vector_type(4)));

namespace {

constexpr int S = 49, C = 32, I = 32;
constexpr int LDY = 53;      // LDS row pitch of a Y row (odd: the per-lane row writes spread over the banks)

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS traffic has landed
  __builtin_amdgcn_wave_barrier();
}

// ---- K1 + K2 ----------------------------------------------------------------------------------------------------------
// F16: K1 on the fp16 matrix pipe with split operands.  The f32-input MFMA runs at the f32 VECTOR rate (chain2.hip): a
// tile of 32 quadruplets costs 64 v_mfma_f32_16x16x4_f32 = 2 k cycles of a pipe that the Y_lm rebuild (VALU) cannot overlap
// within one wave.  A tile of 32 quadruplets is exactly one K = 32 chunk of v_mfma_f32_16x16x32_f16: Y and the gathered x rows
// are split in registers into hi = f16(v) and lo = f16(v - hi) (UNSCALED: both operands are O(1) — harmonics and activations
// — so the fp16 subnormals the hardware keeps bound the error at 6e-8 absolute, below fp32 rounding of the sum) and
// hh + hl + lh accumulate into the same fp32 registers: 24 MFMAs of 16 cycles per tile instead of 64 of 32.  The same
// number of LDS reads and gathered loads as the f32 form; the gathers are issued before the Y rebuild and land under it.
typedef _Float16 h8_a __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split8(const float (&v)[8], h8_a& hi, h8_a& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 h = (_Float16)v[i];
    hi[i] = h;
    lo[i] = (_Float16)(v[i] - (float)h);
  }
}

// the scaled form of chain2.hip: lo = f16((v - hi) 2^11) keeps 22 significand bits for every element down to 1.2e-4 of the
// largest one; its products accumulate in registers of their own and are scaled back once
__device__ __forceinline__ void split8s(const float (&v)[8], h8_a& hi, h8_a& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 h = (_Float16)v[i];
    hi[i] = h;
    lo[i] = (_Float16)((v[i] - (float)h) * 2048.f);
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void bil_reduce_project_ang_kernel(
    const float4* __restrict__ ang, const float* __restrict__ x, const int32_t* __restrict__ expand_idx,
    const int32_t* __restrict__ seg_off, const float* __restrict__ B, float* __restrict__ Sm, float* __restrict__ P,
    int64_t E) {
  // 32 quadruplets per tile: 27 KB of LDS per workgroup keeps 5 workgroups (20 waves) per CU — the gathered x rows are
  // two dependent L2 round trips per K-step, and with 64-quadruplet tiles (2 workgroups per CU) nothing hid them
  constexpr int TQ = 32, LD = C + 4;
  constexpr int YSZ = (TQ * LDY > 52 * LD) ? TQ * LDY : 52 * LD;
  __shared__ __attribute__((aligned(16))) float ysm[4][YSZ];   // Y rows of one tile; later Sm[52][LD] of the edge
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= E) return;
  float* __restrict__ ys = ysm[wave];
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  v4f_a acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (v4f_a){0.f, 0.f, 0.f, 0.f};
  for (int tb = t0; tb < t1; tb += TQ) {
    const int nq = min(TQ, t1 - tb);
    if constexpr (F16) {
      int gq = 0;
      float4 a4 = make_float4(0.f, 1.f, 0.f, 1.f);
      if (lane < nq) {
        gq = expand_idx[tb + lane];
        a4 = ang[tb + lane];
      }
      // B operand: x[g(q)][16 nt + l15] for the 8 quadruplets q = 8 lg + i of this lane's K group — requested now, used
      // after the Y rebuild
      float xb[2][8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = 8 * lg + i;
        const bool ok = q < nq;
        const float* __restrict__ xr = x + (int64_t)__shfl(gq, min(q, nq - 1), 64) * C + l15;
        const float x0 = xr[0], x1 = xr[16];
        xb[0][i] = ok ? x0 : 0.f;
        xb[1][i] = ok ? x1 : 0.f;
      }
      if (lane < nq) ylm7_row_T<float>(a4.x, a4.y, a4.z, a4.w, ys + lane * LDY);
      wave_lds_sync();
      h8_a bh[2], bl[2];
      split8(xb[0], bh[0], bl[0]);
      split8(xb[1], bh[1], bl[1]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        // A operand: Y[q = 8 lg + i][s = 16 mt + l15]; tile 3 carries only s = 48
        float ya[8];
        const int sc = 16 * mt + l15;
        const bool s_ok = sc < S;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int q = 8 * lg + i;
          const float v = ys[min(q, nq - 1) * LDY + min(sc, S - 1)];
          ya[i] = (s_ok && q < nq) ? v : 0.f;
        }
        h8_a ah, al;
        split8(ya, ah, al);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    } else {
    int gq = 0;                          // expand row of quadruplet tb + lane: one coalesced load per tile,
    if (lane < nq) {                     // handed to the lanes that need it by a wave shuffle (no dependent index load)
      gq = expand_idx[tb + lane];
      const float4 a4 = ang[tb + lane];
      ylm7_row_T<float>(a4.x, a4.y, a4.z, a4.w, ys + lane * LDY);
    }
    wave_lds_sync();
    // K-steps of 4 quadruplets: lane (l15, lg) supplies Y[q = k + lg][16 mt + l15] and x[g(q)][16 nt + l15]
    auto loadx = [&](int k, float (&b)[2]) {
      const int q = k + lg;
      const bool ok = q < nq;
      const float* __restrict__ xr = x + (int64_t)__shfl(gq, min(q, nq - 1), 64) * C + l15;
      const float x0 = xr[0], x1 = xr[16];
      b[0] = ok ? x0 : 0.f;
      b[1] = ok ? x1 : 0.f;
    };
    float b0[2], b1[2];
    loadx(0, b0);
    for (int k = 0; k < nq; k += 8) {
      loadx(k + 4, b1);                 // next K-step's gathered rows in flight under this one's MFMAs
      {
        const int q = k + lg;
        const bool ok = q < nq;
        const float* __restrict__ yr = ys + min(q, nq - 1) * LDY;
        const float a0 = ok ? yr[l15] : 0.f, a1 = ok ? yr[16 + l15] : 0.f, a2 = ok ? yr[32 + l15] : 0.f, a3 = ok ? yr[48] : 0.f;
        const float a[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b0[nt], acc[mt][nt], 0, 0, 0);
      }
      loadx(k + 8, b0);
      {
        const int q = k + 4 + lg;
        const bool ok = q < nq;
        const float* __restrict__ yr = ys + min(q, nq - 1) * LDY;
        const float a0 = ok ? yr[l15] : 0.f, a1 = ok ? yr[16 + l15] : 0.f, a2 = ok ? yr[32 + l15] : 0.f, a3 = ok ? yr[48] : 0.f;
        const float a[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b1[nt], acc[mt][nt], 0, 0, 0);
      }
    }
    }
    wave_lds_sync();   // every fragment read of this tile has returned before the next tile overwrites it
  }
  // D layout: col = l15 (c within tile), row = 4 lg + r (s within tile); tile mt = 3 carries only s = 48
  float* __restrict__ so = Sm + e * (int64_t)S * C;
  float (*sml)[LD] = reinterpret_cast<float (*)[LD]>(ys);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int srow = 16 * mt + 4 * lg + r;
        const float v = acc[mt][nt][r];
        if (srow < S) so[srow * C + 16 * nt + l15] = v;
        if (srow < 52) sml[srow][16 * nt + l15] = srow < S ? v : 0.f;
      }
  wave_lds_sync();
  // K2: P[i,c] = sum_s B[e,s,i] Sm[s,c]
  v4f_a pacc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) pacc[mt][nt] = (v4f_a){0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ be = B + e * (int64_t)S * I;
#pragma unroll
  for (int kk = 0; kk < 13; ++kk) {
    const int sk = 4 * kk + lg;
    const bool ok = sk < S;
    const float a_0 = ok ? be[sk * I + l15] : 0.f;
    const float a_1 = ok ? be[sk * I + 16 + l15] : 0.f;
    const float b_0 = sml[sk][l15];
    const float b_1 = sml[sk][16 + l15];
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

// ---- x-adjoint, grouped by reduce edge: dxt[seg(e)] (K4 x C) = Yseg (K4 x S) @ dSm[e] (S x C) -------------------------
// F16 (arith = GN_ANG_F16): the product on the fp16 matrix pipe — Y (O(1)) and sigma dSm[e] (one exact power-of-two
// scale per edge: the block is a cotangent of arbitrary magnitude) split into hi + 2^-11 lo planes, the correction products
// in accumulators of their own (chain2.hip "format H"): 24 MFMAs of 16 cycles per tile of 32 quadruplets instead of 52 of 32.
template <bool F16>
__global__ __launch_bounds__(256) void bil_expand_ang_kernel(const float4* __restrict__ ang, const float* __restrict__ dSm,
                                                             const int32_t* __restrict__ seg_off, float* __restrict__ dxt,
                                                             int64_t E, const int32_t* __restrict__ row_pos) {
  // row_pos (optional): row of dxt that receives quadruplet q's result — the position of q in the order of its EXPAND row
  // (round 5: the segmented sum that follows then reads contiguous rows instead of gathering 128-byte rows through a permutation)
  constexpr int TQ = 32;
  __shared__ float ysm[4][TQ * LDY];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= E) return;
  float* __restrict__ ys = ysm[wave];
  const float* __restrict__ De = dSm + e * (int64_t)S * C;
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  if constexpr (F16) {
    // B operand: dSm[e][s = 32 kc + 8 lg + i][c = 16 nt + l15], i < 8 (s >= 49: zero)
    float bv[2][2][8];
    float vmax = 0.f;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int sr = 32 * kc + 8 * lg + i;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const float v = De[min(sr, S - 1) * C + 16 * nt + l15];
          bv[kc][nt][i] = sr < S ? v : 0.f;
          vmax = fmaxf(vmax, fabsf(bv[kc][nt][i]));
        }
      }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    float sigma = 1.f, inv_sigma = 1.f;
    {
      const uint32_t ex = __float_as_uint(vmax) >> 23;
      const uint32_t ec = ex < 2u ? 2u : (ex > 250u ? 250u : ex);
      if (vmax > 0.f) {
        sigma = __uint_as_float((252u - ec) << 23);      // sigma max|dSm[e]| in [0.25, 0.5)
        inv_sigma = __uint_as_float((2u + ec) << 23);
      }
    }
    h8_a bh[2][2], bl[2][2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) bv[kc][nt][i] *= sigma;
        split8s(bv[kc][nt], bh[kc][nt], bl[kc][nt]);
      }
    for (int tb = t0; tb < t1; tb += TQ) {
      const int nq = min(TQ, t1 - tb);
      if (lane < nq) {
        const float4 a4 = ang[tb + lane];
        ylm7_row_T<float>(a4.x, a4.y, a4.z, a4.w, ys + lane * LDY);
      }
      wave_lds_sync();
      for (int sub = 0; sub < nq; sub += 16) {
        // A operand: Y[q = sub + l15][s = 32 kc + 8 lg + i]
        const float* __restrict__ yb = ys + min(sub + l15, nq - 1) * LDY;
        v4f_a c0 = (v4f_a){0.f, 0.f, 0.f, 0.f}, c1 = c0, x0 = c0, x1 = c0;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          float ya[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int sr = 32 * kc + 8 * lg + i;
            const float v = yb[min(sr, S - 1)];
            ya[i] = sr < S ? v : 0.f;
          }
          h8_a ah, al;
          split8s(ya, ah, al);
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[kc][0], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[kc][1], c1, 0, 0, 0);
          x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[kc][0], x0, 0, 0, 0);
          x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[kc][1], x1, 0, 0, 0);
          x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[kc][0], x0, 0, 0, 0);
          x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[kc][1], x1, 0, 0, 0);
        }
        c0 = (c0 + x0 * (1.f / 2048.f)) * inv_sigma;
        c1 = (c1 + x1 * (1.f / 2048.f)) * inv_sigma;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = sub + 4 * lg + r;
          if (q < nq) {
            float* __restrict__ o = dxt + (int64_t)(row_pos ? row_pos[tb + q] : tb + q) * C + l15;
            o[0] = c0[r];
            o[16] = c1[r];
          }
        }
      }
      wave_lds_sync();
    }
    return;
  }
  float bd[13][2];   // dSm[4 kk + lg][16 nt + l15]
#pragma unroll
  for (int kk = 0; kk < 13; ++kk) {
    const int sr = 4 * kk + lg;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float v = De[min(sr, S - 1) * C + 16 * nt + l15];
      bd[kk][nt] = sr < S ? v : 0.f;
    }
  }
  for (int tb = t0; tb < t1; tb += TQ) {
    const int nq = min(TQ, t1 - tb);
    if (lane < nq) {
      const float4 a4 = ang[tb + lane];
      ylm7_row_T<float>(a4.x, a4.y, a4.z, a4.w, ys + lane * LDY);
    }
    wave_lds_sync();
    for (int sub = 0; sub < nq; sub += 16) {   // row tiles of 16 quadruplets: lane (l15, lg) <- Y[sub + l15][4 kk + lg]
      const float* __restrict__ yb = ys + min(sub + l15, nq - 1) * LDY + lg;
      v4f_a c0 = (v4f_a){0.f, 0.f, 0.f, 0.f}, c1 = (v4f_a){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 13; ++kk) {
        const float yv = yb[4 * kk];
        const float a = (kk < 12 || lg == 0) ? yv : 0.f;
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bd[kk][0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bd[kk][1], c1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = sub + 4 * lg + r;
        if (q < nq) {
          float* __restrict__ o = dxt + (int64_t)(row_pos ? row_pos[tb + q] : tb + q) * C + l15;
          o[0] = c0[r];
          o[16] = c1[r];
        }
      }
    }
    wave_lds_sync();
  }
}

// ---- x-adjoint, fused per TARGET ATOM: no per-quadruplet rows in memory ----------------------------------------------
// dx[j, :] = sum_{q: g(q) = j} Y[q, :] dSm[r(q)]   (interaction_block.py:517-566 backward through the gather of x).
// expand_ang above writes the per-quadruplet rows dxt (Q x 32 floats: 1.15 GB at 9 M quadruplets) and a CSR segmented sum
// reads them back (bil_expand_ang 0.42 ms + segsum 0.30 ms per block, 732 MB counted against 225 MB algorithmic).
// A quadruplet c -> a <- b <- d reduces into the edge c -> a and expands from the intermediate triplet (a <- b <- d): both
// end in atom a, and the intermediate triplets are sorted by a.  So ONE workgroup owns one atom: its rows dx[J_a] live in
// LDS (|J_a| <= 848 rows of 32 floats), the workgroup walks the edges into a IN ORDER — all eight waves on the tiles of one
// edge at a time: the quadruplets of one edge expand from distinct rows, so the adds of one step never collide, and a
// barrier between the steps fixes the order of the sum (deterministic, no atomics) — and writes its rows once.
// Inner tile = expand_ang's: Y rows of 32 quadruplets rebuilt in the wave's LDS tile, dSm[e] as B fragments in registers,
// f32 MFMA 16x16x4.
template <int NW, int TQ>
__global__ __launch_bounds__(NW * 64) void bil_expand_atoms_ang_kernel(
    const float4* __restrict__ ang, const float* __restrict__ dSm, const int32_t* __restrict__ seg_off,
    const int32_t* __restrict__ expand_idx, const int32_t* __restrict__ a_perm, const int32_t* __restrict__ a_seg,
    const int32_t* __restrict__ j_off, float* __restrict__ dx, int max_J) {
  extern __shared__ __attribute__((aligned(16))) float alds[];   // dx rows [max_J][C], then the waves' Y tiles [NW][TQ * LDY]
  const int a = blockIdx.x;
  const int j0 = j_off[a], nJ = j_off[a + 1] - j0;
  if (nJ <= 0) return;
  if (nJ > max_J) __builtin_trap();      // the caller sized the LDS for max_J rows
  float* __restrict__ dxl = alds;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  float* __restrict__ ys = alds + (size_t)max_J * C + wave * (TQ * LDY);
  for (int i = threadIdx.x; i < nJ * (C / 4); i += NW * 64) reinterpret_cast<float4*>(dxl)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int e0 = a_seg[a], e1 = a_seg[a + 1];
  for (int ei = e0; ei < e1; ++ei) {
    const int e = a_perm ? a_perm[ei] : ei;
    const int t0 = seg_off[e], t1 = seg_off[e + 1];
    if (t0 + wave * TQ < t1) {             // (wave-uniform) this wave has a tile of the edge
      const float* __restrict__ De = dSm + e * (int64_t)S * C;
      float bd[13][2];   // dSm[4 kk + lg][16 nt + l15]
#pragma unroll
      for (int kk = 0; kk < 13; ++kk) {
        const int sr = 4 * kk + lg;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const float v = De[min(sr, S - 1) * C + 16 * nt + l15];
          bd[kk][nt] = sr < S ? v : 0.f;
        }
      }
      for (int tb = t0 + wave * TQ; tb < t1; tb += NW * TQ) {
        const int nq = min(TQ, t1 - tb);
        int jl = 0;
        if (lane < nq) {
          jl = expand_idx[tb + lane] - j0;
          const float4 a4 = ang[tb + lane];
          ylm7_row_T<float>(a4.x, a4.y, a4.z, a4.w, ys + lane * LDY);
        }
        wave_lds_sync();
        for (int sub = 0; sub < nq; sub += 16) {
          const float* __restrict__ yb = ys + min(sub + l15, nq - 1) * LDY + lg;
          v4f_a c0 = (v4f_a){0.f, 0.f, 0.f, 0.f}, c1 = (v4f_a){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 13; ++kk) {
            const float yv = yb[4 * kk];
            const float av = (kk < 12 || lg == 0) ? yv : 0.f;
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bd[kk][0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bd[kk][1], c1, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = sub + 4 * lg + r;
            const int row = __shfl(jl, min(q, nq - 1), 64);
            if (q < nq) {
              float* __restrict__ o = dxl + row * C + l15;     // rows of one edge are distinct: no two lanes share an element
              o[0] += c0[r];
              o[16] += c1[r];
            }
          }
        }
        wave_lds_sync();
      }
    }
    __syncthreads();     // every add of edge e has landed before an add of the next edge may touch the same row
  }
  float4* __restrict__ out = reinterpret_cast<float4*>(dx + (int64_t)j0 * C);
  for (int i = threadIdx.x; i < nJ * (C / 4); i += NW * 64) out[i] = reinterpret_cast<const float4*>(dxl)[i];
}

// ---- x-adjoint, ROW-STATIONARY per target atom: no per-quadruplet rows in memory, no barriers (round 6) -----------------
// The same sum as bil_expand_atoms_ang_kernel — dx[J_a] (|J_a| x 32) = sum over the edges e into atom a of Y_e (|J_a| x 49) dSm[e]
// (49 x 32), one GEMM per atom with K = 49 deg(a) — with the loops turned inside out: a WAVE owns 32 of the atom's rows
// (intermediate triplets a <- b <- d) and walks the atom's edges with the accumulators in registers.  Nothing is shared between
// waves: no LDS accumulation, no barrier per edge (what made the per-atom workgroup 5-11 % slower than expand + segmented sum),
// rows written once.  The quadruplet of (edge e, row j) comes from a dense per-atom grid qmap[a][e_local][j_local] (-1: the
// pair is excluded, c = b or c = d: data_container.py:470-473) built once per batch with the index plan (graph.py).
// Same order of addition as the two-pass form (edges of an atom in ascending order).
template <bool F16, int TR>
__global__ __launch_bounds__(256) void bil_expand_rows_ang_kernel(
    const float4* __restrict__ ang, const float* __restrict__ dSm, const int32_t* __restrict__ a_perm,
    const int32_t* __restrict__ a_seg, const int32_t* __restrict__ j_off, const int32_t* __restrict__ qmap,
    const int32_t* __restrict__ g_off, const int32_t* __restrict__ task_atom, const int32_t* __restrict__ task_row0,
    int n_tasks, float* __restrict__ dx) {
  constexpr int NST = TR / 16;
  __shared__ float ysm[4][TR * LDY];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int task = blockIdx.x * 4 + wave;
  if (task >= n_tasks) return;
  float* __restrict__ ys = ysm[wave];
  const int a = task_atom[task], r0 = task_row0[task];
  if (a < 0) return;       // (a slot of a capacity-sized task table beyond the batch's tasks)
  const int j0 = j_off[a], nJ = j_off[a + 1] - j0;
  const int nr = min(TR, nJ - r0);
  const int e0 = a_seg[a], e1 = a_seg[a + 1];
  const int32_t* __restrict__ qrow = qmap + g_off[a] + r0 + lane;
  v4f_a acc[NST][2];
#pragma unroll
  for (int st = 0; st < NST; ++st)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[st][nt] = (v4f_a){0.f, 0.f, 0.f, 0.f};
  int q = -1;
  float4 a4 = make_float4(0.f, 1.f, 0.f, 1.f);
  if (e0 < e1 && lane < nr) {
    q = qrow[0];
    if (q >= 0) a4 = ang[q];
  }
  for (int ei = e0; ei < e1; ++ei) {
    const int e = a_perm ? a_perm[ei] : ei;
    const float* __restrict__ De = dSm + e * (int64_t)S * C;
    // the next edge's quadruplets and angles are requested now and land under this edge's work
    int qn = -1;
    float4 an = make_float4(0.f, 1.f, 0.f, 1.f);
    if (ei + 1 < e1 && lane < nr) {
      qn = qrow[(int64_t)(ei + 1 - e0) * nJ];
      if (qn >= 0) an = ang[qn];
    }
    const unsigned long long valid = __ballot(q >= 0);
    if (q >= 0) ylm7_row_T<float>(a4.x, a4.y, a4.z, a4.w, ys + lane * LDY);
    if constexpr (F16) {
      // B operand: dSm[e][s = 32 kc + 8 lg + i][c = 16 nt + l15] under one exact power-of-two scale per edge (bil_expand_ang_kernel)
      float bv[2][2][8];
      float vmax = 0.f;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int sr = 32 * kc + 8 * lg + i;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const float v = De[min(sr, S - 1) * C + 16 * nt + l15];
            bv[kc][nt][i] = sr < S ? v : 0.f;
            vmax = fmaxf(vmax, fabsf(bv[kc][nt][i]));
          }
        }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
      float sigma = 1.f, inv_sigma = 1.f;
      {
        const uint32_t ex = __float_as_uint(vmax) >> 23;
        const uint32_t ec = ex < 2u ? 2u : (ex > 250u ? 250u : ex);
        if (vmax > 0.f) {
          sigma = __uint_as_float((252u - ec) << 23);      // sigma max|dSm[e]| in [0.25, 0.5)
          inv_sigma = __uint_as_float((2u + ec) << 23);
        }
      }
      h8_a bh[2][2], bl[2][2];
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
          for (int i = 0; i < 8; ++i) bv[kc][nt][i] *= sigma;
          split8s(bv[kc][nt], bh[kc][nt], bl[kc][nt]);
        }
      wave_lds_sync();
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const int sub = 16 * st;
        if (sub < nr) {
          const bool rv = (valid >> (sub + l15)) & 1ull;      // a row without a quadruplet for this edge contributes nothing
          const float* __restrict__ yb = ys + (sub + l15) * LDY;
          v4f_a c0 = (v4f_a){0.f, 0.f, 0.f, 0.f}, c1 = c0, x0 = c0, x1 = c0;
#pragma unroll
          for (int kc = 0; kc < 2; ++kc) {
            float ya[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int sr = 32 * kc + 8 * lg + i;
              const float v = yb[min(sr, S - 1)];
              ya[i] = (sr < S && rv) ? v : 0.f;
            }
            h8_a ah, al;
            split8s(ya, ah, al);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[kc][0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[kc][1], c1, 0, 0, 0);
            x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[kc][0], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[kc][1], x1, 0, 0, 0);
            x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[kc][0], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[kc][1], x1, 0, 0, 0);
          }
          acc[st][0] += (c0 + x0 * (1.f / 2048.f)) * inv_sigma;
          acc[st][1] += (c1 + x1 * (1.f / 2048.f)) * inv_sigma;
        }
      }
    } else {
      float bd[13][2];   // dSm[4 kk + lg][16 nt + l15]
#pragma unroll
      for (int kk = 0; kk < 13; ++kk) {
        const int sr = 4 * kk + lg;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const float v = De[min(sr, S - 1) * C + 16 * nt + l15];
          bd[kk][nt] = sr < S ? v : 0.f;
        }
      }
      wave_lds_sync();
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const int sub = 16 * st;
        if (sub < nr) {
          const bool rv = (valid >> (sub + l15)) & 1ull;
          const float* __restrict__ yb = ys + (sub + l15) * LDY + lg;
          v4f_a c0 = (v4f_a){0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
          for (int kk = 0; kk < 13; ++kk) {
            const float yv = yb[4 * kk];
            const float av = (rv && (kk < 12 || lg == 0)) ? yv : 0.f;
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bd[kk][0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bd[kk][1], c1, 0, 0, 0);
          }
          acc[st][0] += c0;
          acc[st][1] += c1;
        }
      }
    }
    wave_lds_sync();
    q = qn;
    a4 = an;
  }
#pragma unroll
  for (int st = 0; st < NST; ++st)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * st + 4 * lg + r;
      if (row < nr) {
        float* __restrict__ o = dx + (int64_t)(j0 + r0 + row) * C + l15;
        o[0] = acc[st][0][r];
        o[16] = acc[st][1][r];
      }
    }
}

// ---- angle gradient of all blocks that share the basis, one pass ------------------------------------------------------
struct gn_dy_ang_args {
  const float* dS[4];
  const float* x[4];
  int nb;
};

// One WORKGROUP per reduce edge: the dSm_b[e] blocks of all nb blocks (nb x 6.3 KB) are staged in LDS once and shared by
// the four waves, which split the edge's quadruplets in tiles of 16.  (One wave per edge with the 4 x 32 B fragments in
// registers — the layout of bil_dy_multi_mfma49 — needs 256 VGPRs once the angle contraction is added: one wave per
// SIMD, and the gathered x rows were fully exposed: 3.1 ms instead of 1.6 ms at 9 M quadruplets.)
// F16: the contraction dY[q, s] = sum_b sum_c x_b[g(q), c] dSm_b[e, s, c] on the fp16 matrix pipe with split operands (see
// bil_reduce_project_ang_kernel): K = c = 32 is one chunk of v_mfma_f32_16x16x32_f16, 12 MFMAs of 16 cycles per block and
// 16-quadruplet tile instead of 32 f32 MFMAs of 32 cycles.  x is an activation (O(1)); dSm is a COTANGENT of arbitrary
// magnitude (1e-6 on the quadruplet path of a force pass): the staged blocks of the edge are scaled by one exact power of two
// (largest |value| of all nb blocks -> [0.25, 0.5)) before the split, the dY rows by its inverse afterwards.
template <bool F16>
__global__ __launch_bounds__(256) void bil_dy_multi_ang_kernel(const gn_dy_ang_args a, const float4* __restrict__ ang,
                                                               const int32_t* __restrict__ expand_idx,
                                                               const int32_t* __restrict__ seg_off, float4* __restrict__ g_ang,
                                                               int64_t E) {
  constexpr int LDD = C + 4;                          // row pitch of a staged f32 dSm block: 16-byte aligned, conflict-free
  // F16: the blocks are staged ALREADY SPLIT — [nb][plane hi | lo][64 rows][32 halves], rows 49..63 zero — so a B fragment
  // of a tile is two ds_read_b128 and no arithmetic: every 16-quadruplet tile of the edge (~30 of them) reuses the same
  // nb x 49 x 32 values, and splitting them per tile (16 splits of 8 values per lane) made the VALU the bound of this form.
  constexpr int PLB = 64 * C * 2;                     // bytes of one plane of one block: 4 KB
  extern __shared__ __attribute__((aligned(16))) float dsm[];   // f32: [nb][S][LDD]; F16: planes; then [4 waves][16][LDY]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e = blockIdx.x;
  const int nb = a.nb;
  unsigned char* const planes = reinterpret_cast<unsigned char*>(dsm);
  float* __restrict__ dy = (F16 ? reinterpret_cast<float*>(planes + (size_t)nb * 2 * PLB) : dsm + nb * S * LDD) + wave * 16 * LDY;
  __shared__ float wmax[4];
  float sigma = 1.f, inv_sigma = 1.f;
  if constexpr (F16) {
    constexpr int PER = (S * C / 4 + 255) / 256;      // float4 of one block per thread: 2
    float4 held[4][PER];
    float vmax = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b < nb) {
        const float4* __restrict__ src = reinterpret_cast<const float4*>(a.dS[b] + e * (int64_t)S * C);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          const int i = threadIdx.x + 256 * k;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (i < S * C / 4) v = src[i];
          held[b][k] = v;
          vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if (lane == 0) wmax[wave] = vmax;
    // rows 49..63 of every plane: zero (15 rows x 64 bytes x 2 nb planes)
    for (int i = threadIdx.x; i < nb * 2 * 15 * 4; i += 256) {
      const int pl = i / 60, q = i - pl * 60;
      *reinterpret_cast<uint4*>(planes + (size_t)pl * PLB + S * 64 + q * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const uint32_t ex = __float_as_uint(m) >> 23;                      // m >= 0: biased exponent; m 2^(125 - ex) in [0.25, 0.5)
    const uint32_t ec = ex < 2u ? 2u : (ex > 250u ? 250u : ex);
    if (m > 0.f) {
      sigma = __uint_as_float((252u - ec) << 23);
      inv_sigma = __uint_as_float((2u + ec) << 23);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b < nb) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          const int i = threadIdx.x + 256 * k;
          if (i < S * C / 4) {
            const int r = i >> 3, c4i = i & 7;
            const float4 v = held[b][k];
            const float vs[4] = {v.x * sigma, v.y * sigma, v.z * sigma, v.w * sigma};
            typedef _Float16 h4_a __attribute__((ext_vector_type(4)));
            h4_a hh, ll;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const _Float16 h = (_Float16)vs[j];
              hh[j] = h;
              ll[j] = (_Float16)((vs[j] - (float)h) * 2048.f);
            }
            unsigned char* dst = planes + (size_t)(2 * b) * PLB + r * 64 + c4i * 8;
            *reinterpret_cast<h4_a*>(dst) = hh;
            *reinterpret_cast<h4_a*>(dst + PLB) = ll;
          }
        }
      }
    }
    __syncthreads();
  } else {
    for (int b = 0; b < nb; ++b) {                      // stage dSm_b[e] (49 x 32): 392 float4 per block
      const float4* __restrict__ src = reinterpret_cast<const float4*>(a.dS[b] + e * (int64_t)S * C);
      for (int i = threadIdx.x; i < S * C / 4; i += 256) {
        const int r = i >> 3, c4i = i & 7;
        *reinterpret_cast<float4*>(dsm + (b * S + r) * LDD + 4 * c4i) = src[i];
      }
    }
    __syncthreads();
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto comp = [](const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); };
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  for (int tb = t0 + 16 * wave; tb < t1; tb += 64) {
    const int tq = tb + l15;
    const bool ok = tq < t1;
    const int64_t g = ok ? expand_idx[tq] : 0;
    v4f_a c4[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) c4[nt] = (v4f_a){0.f, 0.f, 0.f, 0.f};
    if constexpr (F16) {
      v4f_a cx[4];      // the correction products hl + lh (operands' lo planes carry a factor 2^11)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) cx[nt] = (v4f_a){0.f, 0.f, 0.f, 0.f};
      for (int b = 0; b < nb; ++b) {
        // A operand: x_b[g(q = l15)][c = 8 lg .. 8 lg + 7] — 32 contiguous bytes of the gathered row
        float xa[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ok) {
          const float* __restrict__ xr = a.x[b] + g * C + 8 * lg;
          const float4 u0 = *reinterpret_cast<const float4*>(xr), u1 = *reinterpret_cast<const float4*>(xr + 4);
          xa[0] = u0.x; xa[1] = u0.y; xa[2] = u0.z; xa[3] = u0.w; xa[4] = u1.x; xa[5] = u1.y; xa[6] = u1.z; xa[7] = u1.w;
        }
        h8_a ah, al;
        split8s(xa, ah, al);
        const unsigned char* __restrict__ db = planes + (size_t)(2 * b) * PLB + l15 * 64 + lg * 16;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          // B operand: the staged planes of sigma dSm_b[e][s = 16 nt + l15][c = 8 lg .. 8 lg + 7] (rows >= 49 are zero)
          const h8_a bh = *reinterpret_cast<const h8_a*>(db + nt * 1024);
          const h8_a bl = *reinterpret_cast<const h8_a*>(db + nt * 1024 + PLB);
          c4[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c4[nt], 0, 0, 0);
          cx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, cx[nt], 0, 0, 0);
          cx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, cx[nt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) c4[nt] = (c4[nt] + cx[nt] * (1.f / 2048.f)) * inv_sigma;
    } else {
    for (int b = 0; b < nb; ++b) {
      float4 ax0 = z4, ax1 = z4;
      if (ok) {
        const float* __restrict__ xr = a.x[b] + g * C + 4 * lg;
        ax0 = *reinterpret_cast<const float4*>(xr);
        ax1 = *reinterpret_cast<const float4*>(xr + 16);
      }
      const float* __restrict__ db = dsm + b * S * LDD;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        // B fragments: dSm_b[e][s = 16 nt + l15][16 j + 4 lg ..]; rows >= 49 (tile nt = 3, l15 > 0) contribute nothing
        const int sr = 16 * nt + l15;
        const float* __restrict__ row = db + min(sr, S - 1) * LDD + 4 * lg;
        float4 b0 = *reinterpret_cast<const float4*>(row), b1 = *reinterpret_cast<const float4*>(row + 16);
        if (sr >= S) { b0 = z4; b1 = z4; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          c4[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(ax0, q), comp(b0, q), c4[nt], 0, 0, 0);
          c4[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(ax1, q), comp(b1, q), c4[nt], 0, 0, 0);
        }
      }
    }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      // D layout: row = quadruplet 4 lg + r, col = s = 16 nt + l15 (tile nt = 3: only s = 48)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (nt < 3 || l15 == 0) dy[(4 * lg + r) * LDY + 16 * nt + l15] = c4[nt][r];
    }
    wave_lds_sync();
    if (lane < 16 && tb + lane < t1) {   // one lane per quadruplet: contract its dY row with dY/d(polar, azimuth)
      const float4 a4 = ang[tb + lane];
      float g_first, g_second;
      ylm7_dot_grad_T<float>(a4.x, a4.y, a4.z, a4.w, dy + lane * LDY, g_first, g_second);
      g_ang[tb + lane] = make_float4(g_first, g_second, 0.f, 0.f);
    }
    wave_lds_sync();
  }
}

// ---- second-order sweeps of GemNet-Q force training: TANGENT rows ------------------------------------------------------
// `loss.backward()` through the force (trainer.py:338-346) differentiates the first adjoint of the quadruplet bilinear layer
// (interaction_block.py:517-566, efficient.py:159-189) once more.  With the basis in angle form the tangent of Y along the
// position tangent u = dL/dF is  dY[q] = Y_theta[q] dtheta[q] + Y_phi[q] dphi[q]  — rebuilt per quadruplet with dual numbers
// through the same unrolled recurrences (basis_math.h: ylm7_row_tangent), 32 B per quadruplet (angles + their tangents)
// instead of two (Q, 49) arrays.  Both kernels run the f32-input MFMA: tangents carry the scale of the caller's loss (no
// fp16 range), and they are 2 of the ~11 quadruplet passes of a training step.
//   S3   Smd[e] = sum_{q in seg(e)} ( dY[q] (x) x[g(q)] + Y[q] (x) tx[g(q)] );   Pd[e] = B[e]^T Smd[e] + tB[e]^T Sm[e]
//   S4   dxt[q] = Y[q] D1[e] + dY[q] D2[e]          (x-adjoint rows: D1 = the second adjoint of Sm, D2 = mu_Sm of the first)
template <bool HAS_T, bool HAS_TX>
__global__ __launch_bounds__(256) void bil_reduce_project_ang_tan_kernel(
    const float4* __restrict__ ang, const float4* __restrict__ tang, const float* __restrict__ x, const float* __restrict__ tx,
    const int32_t* __restrict__ expand_idx, const int32_t* __restrict__ seg_off, const float* __restrict__ B,
    const float* __restrict__ tB, const float* __restrict__ Sm, float* __restrict__ Smd, float* __restrict__ Pd, int64_t E) {
  constexpr int TQ = 32, LD = C + 4;
  constexpr int YSZ = (TQ * LDY > 52 * LD) ? TQ * LDY : 52 * LD;
  __shared__ __attribute__((aligned(16))) float ysm[4][YSZ];   // Y rows of one tile; later Smd[52][LD] of the edge
  __shared__ __attribute__((aligned(16))) float ytm[4][HAS_T ? TQ * LDY : 1];   // dY rows of the tile
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= E) return;
  float* __restrict__ ys = ysm[wave];
  float* __restrict__ yt = ytm[wave];
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  v4f_a acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (v4f_a){0.f, 0.f, 0.f, 0.f};
  for (int tb = t0; tb < t1; tb += TQ) {
    const int nq = min(TQ, t1 - tb);
    int gq = 0;
    if (lane < nq) {
      gq = expand_idx[tb + lane];
      const float4 a4 = ang[tb + lane];
      if constexpr (HAS_T) {
        const float4 t4 = tang[tb + lane];
        ylm7_row_tangent(a4.x, a4.y, a4.z, a4.w, t4.x, t4.y, HAS_TX ? ys + lane * LDY : nullptr, yt + lane * LDY);
      } else {
        ylm7_row_T<float>(a4.x, a4.y, a4.z, a4.w, ys + lane * LDY);
      }
    }
    wave_lds_sync();
    for (int k = 0; k < nq; k += 4) {       // K-steps of 4 quadruplets: lane (l15, lg) supplies row q = k + lg
      const int q = k + lg;
      const bool ok = q < nq;
      const int64_t row = (int64_t)__shfl(gq, min(q, nq - 1), 64) * C + l15;
      float b[2] = {0.f, 0.f}, bt[2] = {0.f, 0.f};
      if constexpr (HAS_T) { const float v0 = x[row], v1 = x[row + 16]; b[0] = ok ? v0 : 0.f; b[1] = ok ? v1 : 0.f; }
      if constexpr (HAS_TX) { const float v0 = tx[row], v1 = tx[row + 16]; bt[0] = ok ? v0 : 0.f; bt[1] = ok ? v1 : 0.f; }
      const int qr = min(q, nq - 1) * LDY;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int sc = mt < 3 ? 16 * mt + l15 : 48;
        if constexpr (HAS_T) {
          const float a = ok ? yt[qr + sc] : 0.f;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nt], acc[mt][nt], 0, 0, 0);
        }
        if constexpr (HAS_TX) {
          const float a = ok ? ys[qr + sc] : 0.f;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bt[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    }
    wave_lds_sync();
  }
  // D layout: col = l15 (c within tile), row = 4 lg + r (s within tile).  Tile mt = 3 was fed s = 48 in EVERY row: only its
  // row 0 (s = 48) is meaningful
  float* __restrict__ so = Smd + e * (int64_t)S * C;
  float (*sml)[LD] = reinterpret_cast<float (*)[LD]>(ys);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int srow = 16 * mt + 4 * lg + r;
        const float v = acc[mt][nt][r];
        if (srow < S) so[srow * C + 16 * nt + l15] = v;
        if (srow < 52) sml[srow][16 * nt + l15] = srow < S ? v : 0.f;
      }
  wave_lds_sync();
  if (!Pd) return;
  // K2: Pd[i, c] = sum_s B[e, s, i] Smd[s, c] + sum_s tB[e, s, i] Sm[e, s, c]
  v4f_a pacc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) pacc[mt][nt] = (v4f_a){0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ be = B + e * (int64_t)S * I;
#pragma unroll
  for (int kk = 0; kk < 13; ++kk) {
    const int sk = 4 * kk + lg;
    const bool ok = sk < S;
    const float a_0 = ok ? be[min(sk, S - 1) * I + l15] : 0.f;
    const float a_1 = ok ? be[min(sk, S - 1) * I + 16 + l15] : 0.f;
    const float b_0 = sml[sk][l15];
    const float b_1 = sml[sk][16 + l15];
    pacc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_0, b_0, pacc[0][0], 0, 0, 0);
    pacc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_0, b_1, pacc[0][1], 0, 0, 0);
    pacc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_1, b_0, pacc[1][0], 0, 0, 0);
    pacc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_1, b_1, pacc[1][1], 0, 0, 0);
  }
  if (tB) {
    const float* __restrict__ tbe = tB + e * (int64_t)S * I;
    const float* __restrict__ sme = Sm + e * (int64_t)S * C;
#pragma unroll
    for (int kk = 0; kk < 13; ++kk) {
      const int sk = 4 * kk + lg;
      const bool ok = sk < S;
      const int sr = min(sk, S - 1);
      const float a_0 = ok ? tbe[sr * I + l15] : 0.f;
      const float a_1 = ok ? tbe[sr * I + 16 + l15] : 0.f;
      const float b_0 = ok ? sme[sr * C + l15] : 0.f;
      const float b_1 = ok ? sme[sr * C + 16 + l15] : 0.f;
      pacc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_0, b_0, pacc[0][0], 0, 0, 0);
      pacc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_0, b_1, pacc[0][1], 0, 0, 0);
      pacc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_1, b_0, pacc[1][0], 0, 0, 0);
      pacc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_1, b_1, pacc[1][1], 0, 0, 0);
    }
  }
  float* __restrict__ po = Pd + e * (int64_t)I * C;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) po[(16 * mt + 4 * lg + r) * C + 16 * nt + l15] = pacc[mt][nt][r];
}

template <bool HAS_D1>
__global__ __launch_bounds__(256) void bil_expand_ang_tan_kernel(const float4* __restrict__ ang, const float4* __restrict__ tang,
                                                                 const float* __restrict__ D1, const float* __restrict__ D2,
                                                                 const int32_t* __restrict__ seg_off, float* __restrict__ dxt,
                                                                 int64_t E) {
  constexpr int TQ = 32;
  __shared__ float ysm[4][HAS_D1 ? TQ * LDY : 1];
  __shared__ float ytm[4][TQ * LDY];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= E) return;
  float* __restrict__ ys = ysm[wave];
  float* __restrict__ yt = ytm[wave];
  const int t0 = seg_off[e], t1 = seg_off[e + 1];
  float b1[13][2], b2[13][2];   // D1 / D2 [e][4 kk + lg][16 nt + l15]
#pragma unroll
  for (int kk = 0; kk < 13; ++kk) {
    const int sr = 4 * kk + lg;
    const int64_t off = e * (int64_t)S * C + min(sr, S - 1) * C + l15;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float v2 = D2[off + 16 * nt];
      b2[kk][nt] = sr < S ? v2 : 0.f;
      if constexpr (HAS_D1) {
        const float v1 = D1[off + 16 * nt];
        b1[kk][nt] = sr < S ? v1 : 0.f;
      }
    }
  }
  for (int tb = t0; tb < t1; tb += TQ) {
    const int nq = min(TQ, t1 - tb);
    if (lane < nq) {
      const float4 a4 = ang[tb + lane], t4 = tang[tb + lane];
      ylm7_row_tangent(a4.x, a4.y, a4.z, a4.w, t4.x, t4.y, HAS_D1 ? ys + lane * LDY : nullptr, yt + lane * LDY);
    }
    wave_lds_sync();
    for (int sub = 0; sub < nq; sub += 16) {   // row tiles of 16 quadruplets: lane (l15, lg) <- rows [sub + l15][4 kk + lg]
      const int qr = min(sub + l15, nq - 1) * LDY + lg;
      v4f_a c0 = (v4f_a){0.f, 0.f, 0.f, 0.f}, c1 = (v4f_a){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 13; ++kk) {
        const bool s_ok = kk < 12 || lg == 0;
        const float tv = yt[qr + (kk < 12 ? 4 * kk : 48 - lg)];     // (kk = 12: only s = 48 exists; lanes lg > 0 feed zero)
        const float at = s_ok ? tv : 0.f;
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(at, b2[kk][0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(at, b2[kk][1], c1, 0, 0, 0);
        if constexpr (HAS_D1) {
          const float yv = ys[qr + (kk < 12 ? 4 * kk : 48 - lg)];
          const float a = s_ok ? yv : 0.f;
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1[kk][0], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1[kk][1], c1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = sub + 4 * lg + r;
        if (q < nq) {
          float* __restrict__ o = dxt + (int64_t)(tb + q) * C + l15;
          o[0] = c0[r];
          o[16] = c1[r];
        }
      }
    }
    wave_lds_sync();
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// Matrix-pipe arithmetic of the angle-form kernels: the `arith` ARGUMENT of every entry point (GN_ANG_F16 = 1: the products on
// v_mfma_f32_16x16x32_f16 with split fp16 operands; 0: the f32-input MFMA) — per call, no library state (ABI 13; until ABI 12 a
// process-global mask that the host toggled around single launches, racing with the autograd engine's thread).  Measured on
// MI355X (profiles/r4_q_f16.txt): K1 -25 %, the x-adjoint -6 %, the angle gradient -29 % once the edge's dSm blocks are staged
// ALREADY SPLIT (split per tile it was VALU-bound and no faster than the f32 MFMA).
extern "C" int gn_bil_reduce_project_ang_f32(const float* ang, const float* x, const int32_t* expand_idx,
                                             const int32_t* seg_off, const float* B, float* Sm, float* P, int64_t E, int S_,
                                             int C_, int I_, int arith, void* stream) {
  if (E <= 0) return 0;
  if (S_ != S || C_ != C || I_ != I || !aligned16(ang) || (arith & ~1)) return (int)hipErrorInvalidValue;
  if (arith & 1)
    hipLaunchKernelGGL(bil_reduce_project_ang_kernel<true>, dim3(gn_cdiv(E, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(ang), x, expand_idx, seg_off, B, Sm, P, E);
  else
    hipLaunchKernelGGL(bil_reduce_project_ang_kernel<false>, dim3(gn_cdiv(E, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(ang), x, expand_idx, seg_off, B, Sm, P, E);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_expand_ang_f32(const float* ang, const float* dSm, const int32_t* seg_off, float* dxt, int64_t E, int S_,
                                     int C_, int arith, const int32_t* row_pos, void* stream) {
  if (E <= 0) return 0;
  if (S_ != S || C_ != C || !aligned16(ang) || (arith & ~1)) return (int)hipErrorInvalidValue;
  if (arith & 1)
    hipLaunchKernelGGL(bil_expand_ang_kernel<true>, dim3(gn_cdiv(E, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(ang), dSm, seg_off, dxt, E, row_pos);
  else
    hipLaunchKernelGGL(bil_expand_ang_kernel<false>, dim3(gn_cdiv(E, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(ang), dSm, seg_off, dxt, E, row_pos);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_expand_atoms_ang_f32(const float* ang, const float* dSm, const int32_t* seg_off, const int32_t* expand_idx,
                                           const int32_t* a_perm, const int32_t* a_seg, const int32_t* j_off, float* dx,
                                           int64_t n_atoms, int max_J, int S_, int C_, void* stream) {
  if (n_atoms <= 0) return 0;
  if (S_ != S || C_ != C || !aligned16(ang) || max_J < 1) return (int)hipErrorInvalidValue;
  // 16 waves with tiles of 16 quadruplets (the same 13 568 floats of Y tiles as 8 waves x 32): twice the waves per step
  constexpr int NW = 16, TQ = 16;
  const size_t lds = ((size_t)max_J * C + (size_t)NW * TQ * LDY) * sizeof(float);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  static std::atomic<bool> configured{false};   // set-once flag of an idempotent attribute (two racing threads both set it)
  if (!configured.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bil_expand_atoms_ang_kernel<NW, TQ>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    configured.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((bil_expand_atoms_ang_kernel<NW, TQ>), dim3((unsigned)n_atoms), dim3(NW * 64), lds,
                     static_cast<hipStream_t>(stream), reinterpret_cast<const float4*>(ang), dSm, seg_off, expand_idx, a_perm,
                     a_seg, j_off, dx, max_J);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_expand_rows_ang_f32(const float* ang, const float* dSm, const int32_t* a_perm, const int32_t* a_seg,
                                          const int32_t* j_off, const int32_t* qmap, const int32_t* g_off,
                                          const int32_t* task_atom, const int32_t* task_row0, int64_t n_tasks, float* dx, int S_,
                                          int C_, int arith, void* stream) {
  if (n_tasks <= 0) return 0;
  const int tile = (arith >> 8) & 0xff;       // rows per task the caller built its task table for (0: 32)
  if (S_ != S || C_ != C || !aligned16(ang) || n_tasks > (1ll << 30) || (tile != 0 && tile != 32 && tile != 64))
    return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned)gn_cdiv(n_tasks, 4)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define GN_ROWS_LAUNCH(F, T) hipLaunchKernelGGL((bil_expand_rows_ang_kernel<F, T>), grid, block, 0, st, reinterpret_cast<const float4*>(ang), \
                                                dSm, a_perm, a_seg, j_off, qmap, g_off, task_atom, task_row0, (int)n_tasks, dx)
  if (arith & GN_ANG_F16) { if (tile == 64) GN_ROWS_LAUNCH(true, 64); else GN_ROWS_LAUNCH(true, 32); }
  else { if (tile == 64) GN_ROWS_LAUNCH(false, 64); else GN_ROWS_LAUNCH(false, 32); }
#undef GN_ROWS_LAUNCH
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_dy_multi_ang_f32(const float* const* dSm_list, const float* const* x_list, int nb, const float* ang,
                                       const int32_t* expand_idx, const int32_t* seg_off, float* g_ang, int64_t E, int S_,
                                       int C_, int arith, void* stream) {
  if (E <= 0 || nb <= 0) return 0;
  if (nb > 4 || S_ != S || C_ != C || !aligned16(ang) || !aligned16(g_ang) || (arith & ~1)) return (int)hipErrorInvalidValue;
  gn_dy_ang_args a;
  a.nb = nb;
  for (int b = 0; b < 4; ++b) {
    a.dS[b] = b < nb ? dSm_list[b] : nullptr;
    a.x[b] = b < nb ? x_list[b] : nullptr;
    if (b < nb && (!aligned16(a.dS[b]) || !aligned16(a.x[b]))) return (int)hipErrorInvalidValue;
  }
  const size_t smem = ((size_t)nb * S * (C + 4) + 4 * 16 * LDY) * sizeof(float);   // 42 KB at nb = 4
  const size_t smem16 = (size_t)nb * 2 * 64 * C * 2 + (size_t)4 * 16 * LDY * sizeof(float);   // split planes: 46 KB at nb = 4
  if (arith & 1)
    hipLaunchKernelGGL(bil_dy_multi_ang_kernel<true>, dim3((unsigned)E), dim3(256), smem16, static_cast<hipStream_t>(stream), a,
                       reinterpret_cast<const float4*>(ang), expand_idx, seg_off, reinterpret_cast<float4*>(g_ang), E);
  else
    hipLaunchKernelGGL(bil_dy_multi_ang_kernel<false>, dim3((unsigned)E), dim3(256), smem, static_cast<hipStream_t>(stream), a,
                       reinterpret_cast<const float4*>(ang), expand_idx, seg_off, reinterpret_cast<float4*>(g_ang), E);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_reduce_project_ang_tan_f32(const float* ang, const float* tang, const float* x, const float* tx,
                                                 const int32_t* expand_idx, const int32_t* seg_off, const float* B,
                                                 const float* tB, const float* Sm, float* Smd, float* Pd, int64_t E, int S_,
                                                 int C_, int I_, void* stream) {
  if (E <= 0) return 0;
  if (S_ != S || C_ != C || I_ != I || !aligned16(ang) || (tang && !aligned16(tang))) return (int)hipErrorInvalidValue;
  if ((!tang && !tx) || (tang && !x) || !Smd || (tB && !Sm)) return (int)hipErrorInvalidValue;
  const dim3 grid(gn_cdiv(E, 4)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float4* a4 = reinterpret_cast<const float4*>(ang);
  const float4* t4 = reinterpret_cast<const float4*>(tang);
  if (tang && tx)
    hipLaunchKernelGGL((bil_reduce_project_ang_tan_kernel<true, true>), grid, block, 0, st, a4, t4, x, tx, expand_idx, seg_off, B,
                       tB, Sm, Smd, Pd, E);
  else if (tang)
    hipLaunchKernelGGL((bil_reduce_project_ang_tan_kernel<true, false>), grid, block, 0, st, a4, t4, x, tx, expand_idx, seg_off,
                       B, tB, Sm, Smd, Pd, E);
  else
    hipLaunchKernelGGL((bil_reduce_project_ang_tan_kernel<false, true>), grid, block, 0, st, a4, t4, x, tx, expand_idx, seg_off,
                       B, tB, Sm, Smd, Pd, E);
  GN_LAUNCH_CHECK();
  return 0;
}

// ---- x-adjoint of S4 (Y D1 + dY D2), ROW-STATIONARY per target atom (round 6): the tangent counterpart of
// bil_expand_rows_ang_kernel — a wave owns 64 expand rows of one atom (two halves of 32 share the LDS tiles), walks the atom's
// reduce edges with the accumulators in registers; no per-quadruplet rows in memory, no segmented sum behind it.  f32 MFMA (the
// cotangent blocks follow the caller's loss scale), same products and order of addition as gn_bil_expand_ang_tan_f32 + segsum.
namespace {
template <bool HAS_D1>
__global__ __launch_bounds__(256, 2) void bil_expand_rows_ang_tan_kernel(
    const float4* __restrict__ ang, const float4* __restrict__ tang, const float* __restrict__ D1, const float* __restrict__ D2,
    const int32_t* __restrict__ a_perm, const int32_t* __restrict__ a_seg, const int32_t* __restrict__ j_off,
    const int32_t* __restrict__ qmap, const int32_t* __restrict__ g_off, const int32_t* __restrict__ task_atom,
    const int32_t* __restrict__ task_row0, int n_tasks, float* __restrict__ dx) {
  constexpr int TR = 64, TH = 32;
  __shared__ float ysm[4][HAS_D1 ? TH * LDY : 1];
  __shared__ float ytm[4][TH * LDY];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int task = blockIdx.x * 4 + wave;
  if (task >= n_tasks) return;
  float* __restrict__ ys = ysm[wave];
  float* __restrict__ yt = ytm[wave];
  const int a = task_atom[task], r0 = task_row0[task];
  if (a < 0) return;       // (a slot of a capacity-sized task table beyond the batch's tasks)
  const int j0 = j_off[a], nJ = j_off[a + 1] - j0;
  const int nr = min(TR, nJ - r0);
  const int e0 = a_seg[a], e1 = a_seg[a + 1];
  const int32_t* __restrict__ qrow = qmap + g_off[a] + r0 + lane;
  v4f_a acc[4][2];
#pragma unroll
  for (int st = 0; st < 4; ++st)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[st][nt] = (v4f_a){0.f, 0.f, 0.f, 0.f};
  int q = -1;
  float4 a4 = make_float4(0.f, 1.f, 0.f, 1.f), t4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e0 < e1 && lane < nr) {
    q = qrow[0];
    if (q >= 0) { a4 = ang[q]; t4 = tang[q]; }
  }
  for (int ei = e0; ei < e1; ++ei) {
    const int e = a_perm ? a_perm[ei] : ei;
    int qn = -1;
    float4 an = make_float4(0.f, 1.f, 0.f, 1.f), tn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ei + 1 < e1 && lane < nr) {
      qn = qrow[(int64_t)(ei + 1 - e0) * nJ];
      if (qn >= 0) { an = ang[qn]; tn = tang[qn]; }
    }
    const unsigned long long valid = __ballot(q >= 0);
    float b1[13][2], b2[13][2];   // D1 / D2 [e][4 kk + lg][16 nt + l15]
#pragma unroll
    for (int kk = 0; kk < 13; ++kk) {
      const int sr = 4 * kk + lg;
      const int64_t off = e * (int64_t)S * C + min(sr, S - 1) * C + l15;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float v2 = D2[off + 16 * nt];
        b2[kk][nt] = sr < S ? v2 : 0.f;
        if constexpr (HAS_D1) {
          const float v1 = D1[off + 16 * nt];
          b1[kk][nt] = sr < S ? v1 : 0.f;
        }
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (TH * h < nr) {
        // the 32 lanes that hold this half's rows rebuild their harmonics and tangent rows in the tiles
        if ((lane >> 5) == h && q >= 0)
          ylm7_row_tangent(a4.x, a4.y, a4.z, a4.w, t4.x, t4.y, HAS_D1 ? ys + (lane & 31) * LDY : nullptr, yt + (lane & 31) * LDY);
        wave_lds_sync();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int sub = TH * h + 16 * s2;
          if (sub < nr) {
            const bool rv = (valid >> (sub + l15)) & 1ull;
            const int qr = (16 * s2 + l15) * LDY + lg;
            v4f_a c0 = (v4f_a){0.f, 0.f, 0.f, 0.f}, c1 = (v4f_a){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 13; ++kk) {
              const bool s_ok = rv && (kk < 12 || lg == 0);
              const float tv = yt[qr + (kk < 12 ? 4 * kk : 48 - lg)];
              const float at = s_ok ? tv : 0.f;
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(at, b2[kk][0], c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(at, b2[kk][1], c1, 0, 0, 0);
              if constexpr (HAS_D1) {
                const float yv = ys[qr + (kk < 12 ? 4 * kk : 48 - lg)];
                const float av = s_ok ? yv : 0.f;
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1[kk][0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1[kk][1], c1, 0, 0, 0);
              }
            }
            acc[2 * h + s2][0] += c0;
            acc[2 * h + s2][1] += c1;
          }
        }
        wave_lds_sync();
      }
    }
    q = qn;
    a4 = an;
    t4 = tn;
  }
#pragma unroll
  for (int st = 0; st < 4; ++st)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * st + 4 * lg + r;
      if (row < nr) {
        float* __restrict__ o = dx + (int64_t)(j0 + r0 + row) * C + l15;
        o[0] = acc[st][0][r];
        o[16] = acc[st][1][r];
      }
    }
}
}  // namespace

extern "C" int gn_bil_expand_rows_ang_tan_f32(const float* ang, const float* tang, const float* D1, const float* D2,
                                              const int32_t* a_perm, const int32_t* a_seg, const int32_t* j_off,
                                              const int32_t* qmap, const int32_t* g_off, const int32_t* task_atom,
                                              const int32_t* task_row0, int64_t n_tasks, float* dx, int S_, int C_, int tile,
                                              void* stream) {
  if (n_tasks <= 0) return 0;
  if (S_ != S || C_ != C || !aligned16(ang) || !tang || !aligned16(tang) || !D2 || tile != 64 || n_tasks > (1ll << 30))
    return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned)gn_cdiv(n_tasks, 4)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (D1)
    hipLaunchKernelGGL(bil_expand_rows_ang_tan_kernel<true>, grid, block, 0, st, reinterpret_cast<const float4*>(ang),
                       reinterpret_cast<const float4*>(tang), D1, D2, a_perm, a_seg, j_off, qmap, g_off, task_atom, task_row0,
                       (int)n_tasks, dx);
  else
    hipLaunchKernelGGL(bil_expand_rows_ang_tan_kernel<false>, grid, block, 0, st, reinterpret_cast<const float4*>(ang),
                       reinterpret_cast<const float4*>(tang), D1, D2, a_perm, a_seg, j_off, qmap, g_off, task_atom, task_row0,
                       (int)n_tasks, dx);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_bil_expand_ang_tan_f32(const float* ang, const float* tang, const float* D1, const float* D2,
                                         const int32_t* seg_off, float* dxt, int64_t E, int S_, int C_, void* stream) {
  if (E <= 0) return 0;
  if (S_ != S || C_ != C || !aligned16(ang) || !tang || !aligned16(tang) || !D2) return (int)hipErrorInvalidValue;
  const dim3 grid(gn_cdiv(E, 4)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (D1)
    hipLaunchKernelGGL(bil_expand_ang_tan_kernel<true>, grid, block, 0, st, reinterpret_cast<const float4*>(ang),
                       reinterpret_cast<const float4*>(tang), D1, D2, seg_off, dxt, E);
  else
    hipLaunchKernelGGL(bil_expand_ang_tan_kernel<false>, grid, block, 0, st, reinterpret_cast<const float4*>(ang),
                       reinterpret_cast<const float4*>(tang), D1, D2, seg_off, dxt, E);
  GN_LAUNCH_CHECK();
  return 0;
}
