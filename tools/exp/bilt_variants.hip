// EXPERIMENT (not part of the product library): variants of the atom-grouped triplet adjoint, to find what bounds it.
// Built by tools/bilt_bench.py into tools/exp/libbilt.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int S>
struct grow {
  int j, k0, k1, p;
  float y[S];
};

// MODE 0 full, 1 no compute, 2 no tile fill, 3 no Y fetch
template <int S, int MODE, int NT>
__global__ __launch_bounds__(NT) void grouped(const float* __restrict__ Y, const float* __restrict__ dSm,
                                              const int32_t* __restrict__ grp_rows, const int32_t* __restrict__ grp_off,
                                              const int2* __restrict__ grp_kseg, const int32_t* __restrict__ permT,
                                              const int32_t* __restrict__ rposT, float* __restrict__ dx) {
  constexpr int C = 64, SC = S * C, NV = SC / 4, NW = NT / 64;
  extern __shared__ float gtile[];
  const int g = blockIdx.x;
  const int r0 = grp_off[g], n = grp_off[g + 1] - r0;
  if (n <= 0) return;
  if (MODE != 2) {
    for (int i = threadIdx.x; i < n * NV; i += NT) {
      const int l = i / NV, v = i - l * NV;
      const float4 d = reinterpret_cast<const float4*>(dSm + (int64_t)grp_rows[r0 + l] * SC)[v];
      reinterpret_cast<float4*>(gtile + l * SC)[v] = d;
    }
  }
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto fetch = [&](grow<S>& r, int kb) {
    const bool on = kb + lane < r.k1;
    const int t = on ? permT[kb + lane] : 0;
    r.p = on ? rposT[kb + lane] * SC : 0;
#pragma unroll
    for (int s = 0; s < S; ++s) r.y[s] = MODE == 3 ? (float)t : (on ? Y[(int64_t)t * S + s] : 0.f);
  };
  auto open_row = [&](grow<S>& r, int l) {
    r.j = -1, r.k0 = r.k1 = 0;
    if (l >= n) return;
    const int2 ks = grp_kseg[r0 + l];
    r.j = __builtin_amdgcn_readfirstlane(grp_rows[r0 + l]);
    r.k0 = __builtin_amdgcn_readfirstlane(ks.x);
    r.k1 = __builtin_amdgcn_readfirstlane(ks.y);
    fetch(r, r.k0);
  };
  auto finish_row = [&](grow<S>& r) {
    if (r.j < 0) return;
    float acc = 0.f, acc2 = 0.f;
    if (MODE == 1) {
      acc = r.p;
      for (int s = 0; s < S; ++s) acc += r.y[s];
      acc += gtile[lane];
    } else {
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
    }
    dx[(int64_t)r.j * C + lane] = acc + acc2;
  };
  grow<S> ra, rb;
  open_row(ra, w), open_row(rb, w + NW);
  __syncthreads();
  for (int l = w; l < n; l += 2 * NW) {
    finish_row(ra), finish_row(rb);
    open_row(ra, l + 2 * NW), open_row(rb, l + 3 * NW);
  }
}

// Y of the group staged in LDS too (its rows are deg contiguous runs of the reduce-sorted triplet list).
// grp_yseg[i] = {t0, t1, ybase, -}: triplets [t0,t1) of reduce row i go to ybuf[ybase*S ...];
// packT[k] = (yloc << 8) | rpos.
template <int S, int NT>
__global__ __launch_bounds__(NT) void staged(const float* __restrict__ Y, const float* __restrict__ dSm,
                                             const int32_t* __restrict__ grp_rows, const int32_t* __restrict__ grp_off,
                                             const int2* __restrict__ grp_kseg, const int4* __restrict__ grp_yseg,
                                             const int32_t* __restrict__ packT, float* __restrict__ dx, int tile_rows) {
  constexpr int C = 64, SC = S * C, NV = SC / 4, NW = NT / 64;
  extern __shared__ float gtile[];
  float* ybuf = gtile + tile_rows * SC;
  const int g = blockIdx.x;
  const int r0 = grp_off[g], n = grp_off[g + 1] - r0;
  if (n <= 0) return;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < n * NV; i += NT) {
    const int l = i / NV, v = i - l * NV;
    const float4 d = reinterpret_cast<const float4*>(dSm + (int64_t)grp_rows[r0 + l] * SC)[v];
    reinterpret_cast<float4*>(gtile + l * SC)[v] = d;
  }
  for (int l = w; l < n; l += NW) {
    const int4 ys = grp_yseg[r0 + l];
    const int len = (ys.y - ys.x) * S;
    const float* src = Y + (int64_t)ys.x * S;
    float* dst = ybuf + ys.z * S;
    for (int f = lane; f < len; f += 64) dst[f] = src[f];
  }
  int ja, jb, ka0 = 0, ka1 = 0, kb0 = 0, kb1 = 0, pa = 0, pb = 0;
  auto open_row = [&](int l, int& j, int& k0, int& k1, int& pk) {
    j = -1, k0 = k1 = 0;
    if (l >= n) return;
    const int2 ks = grp_kseg[r0 + l];
    j = __builtin_amdgcn_readfirstlane(grp_rows[r0 + l]);
    k0 = __builtin_amdgcn_readfirstlane(ks.x);
    k1 = __builtin_amdgcn_readfirstlane(ks.y);
    pk = k0 + lane < k1 ? packT[k0 + lane] : 0;
  };
  auto finish_row = [&](int j, int k0, int k1, int pk) {
    if (j < 0) return;
    float acc = 0.f, acc2 = 0.f;
    for (int kb = k0; kb < k1; kb += 64) {
      if (kb != k0) pk = kb + lane < k1 ? packT[kb + lane] : 0;
      const int m = min(64, k1 - kb);
      float y[S];
      const float* yl = ybuf + (pk >> 8) * S;
#pragma unroll
      for (int s = 0; s < S; ++s) y[s] = yl[s];
      const int p = (pk & 255) * SC;
      int k = 0;
      for (; k + 2 <= m; k += 2) {
        const float* da = gtile + __builtin_amdgcn_readlane(p, k) + lane;
        const float* db = gtile + __builtin_amdgcn_readlane(p, k + 1) + lane;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(y[s]), k)), da[s * C], acc);
          acc2 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(y[s]), k + 1)), db[s * C], acc2);
        }
      }
      if (k < m) {
        const float* da = gtile + __builtin_amdgcn_readlane(p, k) + lane;
#pragma unroll
        for (int s = 0; s < S; ++s)
          acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(y[s]), k)), da[s * C], acc);
      }
    }
    dx[(int64_t)j * C + lane] = acc + acc2;
  };
  open_row(w, ja, ka0, ka1, pa), open_row(w + NW, jb, kb0, kb1, pb);
  __syncthreads();
  for (int l = w; l < n; l += 2 * NW) {
    finish_row(ja, ka0, ka1, pa), finish_row(jb, kb0, kb1, pb);
    open_row(l + 2 * NW, ja, ka0, ka1, pa), open_row(l + 3 * NW, jb, kb0, kb1, pb);
  }
}

// Wave-uniform indices and Y rows through the scalar cache (s_load): no v_readlane, FMAs take Y from SGPRs.
template <int S, int NT, int UN>
__global__ __launch_bounds__(NT) void scalar_y(const float* __restrict__ Y, const float* __restrict__ dSm,
                                               const int32_t* __restrict__ grp_rows, const int32_t* __restrict__ grp_off,
                                               const int2* __restrict__ grp_kseg, const int32_t* __restrict__ permT,
                                               const int32_t* __restrict__ rposT, float* __restrict__ dx) {
  constexpr int C = 64, SC = S * C, NV = SC / 4, NW = NT / 64;
  extern __shared__ float gtile[];
  const int g = blockIdx.x;
  const int r0 = grp_off[g], n = grp_off[g + 1] - r0;
  if (n <= 0) return;
  for (int i = threadIdx.x; i < n * NV; i += NT) {
    const int l = i / NV, v = i - l * NV;
    const float4 d = reinterpret_cast<const float4*>(dSm + (int64_t)grp_rows[r0 + l] * SC)[v];
    reinterpret_cast<float4*>(gtile + l * SC)[v] = d;
  }
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __syncthreads();
  for (int l = w; l < n; l += NW) {
    const int j = __builtin_amdgcn_readfirstlane(grp_rows[r0 + l]);
    const int k0 = __builtin_amdgcn_readfirstlane(grp_kseg[r0 + l].x);
    const int k1 = __builtin_amdgcn_readfirstlane(grp_kseg[r0 + l].y);
    float acc[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) acc[u] = 0.f;
    int k = k0;
    for (; k + UN <= k1; k += UN) {
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int t = permT[k + u];
        const float* __restrict__ y = Y + (int64_t)t * S;
        const float* d = gtile + rposT[k + u] * SC + lane;
#pragma unroll
        for (int s = 0; s < S; ++s) acc[u] = fmaf(y[s], d[s * C], acc[u]);
      }
    }
    for (; k < k1; ++k) {
      const int t = permT[k];
      const float* __restrict__ y = Y + (int64_t)t * S;
      const float* d = gtile + rposT[k] * SC + lane;
#pragma unroll
      for (int s = 0; s < S; ++s) acc[0] = fmaf(y[s], d[s * C], acc[0]);
    }
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < UN; ++u) a += acc[u];
    dx[(int64_t)j * C + lane] = a;
  }
}

template <typename K>
static int big_lds(K k, size_t lds) {
  if (lds > 64 * 1024) return (int)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return 0;
}

#define LAUNCH_G(MODE, NT)                                                                                      \
  {                                                                                                             \
    if (big_lds(grouped<7, MODE, NT>, lds)) return -2;                                                          \
    hipLaunchKernelGGL((grouped<7, MODE, NT>), dim3(G), dim3(NT), lds, st, Y, dSm, grp_rows, grp_off,           \
                       (const int2*)grp_kseg, permT, rposT, dx);                                                \
  }

extern "C" int bilt_grouped(int mode, int nt, const float* Y, const float* dSm, const int32_t* grp_rows,
                            const int32_t* grp_off, const int32_t* grp_kseg, const int32_t* permT, const int32_t* rposT,
                            float* dx, int G, int lds_rows, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  size_t lds = (size_t)lds_rows * 7 * 64 * 4;
  if (nt == 1024) {
    if (mode == 0) LAUNCH_G(0, 1024) else if (mode == 1) LAUNCH_G(1, 1024) else if (mode == 2) LAUNCH_G(2, 1024) else LAUNCH_G(3, 1024)
  } else if (nt == 512) {
    if (mode == 0) LAUNCH_G(0, 512) else return -1;
  } else if (nt == 256) {
    if (mode == 0) LAUNCH_G(0, 256) else return -1;
  } else return -1;
  return (int)hipGetLastError();
}

extern "C" int bilt_staged(int nt, const float* Y, const float* dSm, const int32_t* grp_rows, const int32_t* grp_off,
                           const int32_t* grp_kseg, const int32_t* grp_yseg, const int32_t* packT, float* dx, int G,
                           int tile_rows, int y_rows, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  size_t lds = ((size_t)tile_rows * 7 * 64 + (size_t)y_rows * 7) * 4;
  if (lds > 160 * 1024) return -3;
  if (nt == 1024) {
    if (big_lds(staged<7, 1024>, lds)) return -2;
    hipLaunchKernelGGL((staged<7, 1024>), dim3(G), dim3(1024), lds, st, Y, dSm, grp_rows, grp_off, (const int2*)grp_kseg,
                       (const int4*)grp_yseg, packT, dx, tile_rows);
  } else if (nt == 512) {
    if (big_lds(staged<7, 512>, lds)) return -2;
    hipLaunchKernelGGL((staged<7, 512>), dim3(G), dim3(512), lds, st, Y, dSm, grp_rows, grp_off, (const int2*)grp_kseg,
                       (const int4*)grp_yseg, packT, dx, tile_rows);
  } else return -1;
  return (int)hipGetLastError();
}

#define LAUNCH_S(NT, UN)                                                                                  \
  {                                                                                                       \
    if (big_lds(scalar_y<7, NT, UN>, lds)) return -2;                                                     \
    hipLaunchKernelGGL((scalar_y<7, NT, UN>), dim3(G), dim3(NT), lds, st, Y, dSm, grp_rows, grp_off,      \
                       (const int2*)grp_kseg, permT, rposT, dx);                                          \
  }

extern "C" int bilt_scalar(int nt, int un, const float* Y, const float* dSm, const int32_t* grp_rows,
                           const int32_t* grp_off, const int32_t* grp_kseg, const int32_t* permT, const int32_t* rposT,
                           float* dx, int G, int lds_rows, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  size_t lds = (size_t)lds_rows * 7 * 64 * 4;
  if (nt == 1024 && un == 2) LAUNCH_S(1024, 2)
  else if (nt == 1024 && un == 4) LAUNCH_S(1024, 4)
  else if (nt == 1024 && un == 1) LAUNCH_S(1024, 1)
  else if (nt == 512 && un == 4) LAUNCH_S(512, 4)
  else if (nt == 512 && un == 2) LAUNCH_S(512, 2)
  else return -1;
  return (int)hipGetLastError();
}

typedef float v4f_b __attribute__((ext_vector_type(4)));
template <bool ACC, int MODE>
__global__ __launch_bounds__(256) void bwd7_variant(
    const float* __restrict__ dP, const float* __restrict__ Sm, const float* __restrict__ B,
    const float* __restrict__ x, const int32_t* __restrict__ expand_idx, const int32_t* __restrict__ seg_off,
    float* __restrict__ gB, float* __restrict__ dSm, float* __restrict__ dY, int64_t E) {
  constexpr int S = 7, C = 64, I = 16, LD = C + 4;
  __shared__ __attribute__((aligned(16))) float dsl[4][8][LD];   // dSm of this wave's edge, row 7 zero
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
  const int gl = t0 + lane < t1 ? expand_idx[t0 + lane] : 0;   // expand rows of the segment, 64 per wave
  auto loadx = [&](int tb, float4 (&xa)[4]) {   // x rows of the 16 triplets tb.., one per l15, k = c contiguous
    const int tq = tb + l15;
    const bool ok = tq < t1;
    const int rel = tq - t0;
    int gx = __shfl(gl, rel & 63);
    if (rel >= 64) gx = ok ? expand_idx[tq] : 0;
    const float* __restrict__ xr = x + (int64_t)gx * C + 4 * lg;
#pragma unroll
    for (int j = 0; j < 4; ++j) xa[j] = (ok && MODE != 1) ? *reinterpret_cast<const float4*>(xr + 16 * j) : make_float4(tb, j, 1.f, 2.f);
  };
  float4 smf[4], dpf[4], xa[4], xb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    smf[j] = srow ? *reinterpret_cast<const float4*>(Sme + l15 * C + 16 * j + 4 * lg) : z4;
    dpf[j] = *reinterpret_cast<const float4*>(dPe + l15 * C + 16 * j + 4 * lg);
  }
  const float4 bf = srow ? *reinterpret_cast<const float4*>(Be + l15 * I + 4 * lg) : z4;
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
    if (MODE == 4 ? g[r] == 1.2345f : (4 * lg + r < S)) gbo[(4 * lg + r) * I + l15] = g[r];
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
      if (MODE == 4 ? d[nt][r] == 1.2345f : (srw < S)) dso[srw * C + 16 * nt + l15] = d[nt][r];
      if (srw < 8) dsl[wave][srw][16 * nt + l15] = d[nt][r];   // rows >= 7 of D are exact zeros (A rows were)
    }
  if (!dY) return;
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes are visible to its own reads
  __builtin_amdgcn_wave_barrier();
  // ---- (3) dY = Xseg dSm^T
  float4 bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bs[j] = l15 < 8 ? *reinterpret_cast<const float4*>(&dsl[wave][l15][16 * j + 4 * lg]) : z4;
  for (int tb = t0; tb < t1; tb += 16) {
    float4 xn[4];
    loadx(tb + 32, xn);
    v4f_b y = (v4f_b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) { if (MODE == 3) y[cp] += comp(xa[j], cp) * comp(bs[j], cp); else y = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(xa[j], cp), comp(bs[j], cp), y, 0, 0, 0); }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = tb + 4 * lg + r;
      if (MODE == 2 ? (y[r] == 1.2345f) : (srow && t < t1)) {
        float* o = dY + (int64_t)t * S + l15;
        *o = ACC ? *o + y[r] : y[r];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) xa[j] = xb[j], xb[j] = xn[j];
  }
}


extern "C" int bwd7(int mode, const float* dP, const float* Sm, const float* B, const float* x, const int32_t* expand_idx,
                    const int32_t* seg_off, float* gB, float* dSm, float* dY, int64_t E, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((E + 3) / 4)), block(256);
#define L7(M) hipLaunchKernelGGL((bwd7_variant<false, M>), grid, block, 0, st, dP, Sm, B, x, expand_idx, seg_off, gB, dSm, dY, E)
  if (mode == 0) L7(0); else if (mode == 1) L7(1); else if (mode == 2) L7(2); else if (mode == 3) L7(3); else L7(4);
  return (int)hipGetLastError();
}
