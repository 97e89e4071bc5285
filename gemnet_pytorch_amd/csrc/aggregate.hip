// Radial-weighted edge -> atom aggregation of AtomUpdateBlock / OutputBlock (include/gemnet_hip.h,
// gn_rbf_aggregate_fwd_f32 / gn_rbf_aggregate_bwd_f32).
//
// Reference (atom_update_block.py:60-68, :157-172):   x_e = m_e (.) (W rbf_e);   out_a = scale * sum_{e -> a} x_e
// i.e. a K = 16 Dense over all edges, a Hadamard product and torch_scatter.scatter(add).  As separate launches that
// is a (E,16)x(16,128) GEMM writing (E,128), a segmented sum re-reading it, and in the adjoint a row gather, an
// elementwise product and a (E,128)x(128,16) GEMM: five passes over (E,128) tensors for ~0.1 GFLOP.  Here each is ONE
// pass: m is read once, W (8 KB) lives in registers, the per-edge products never reach memory.
//   forward : one 256-thread workgroup per atom; wave w walks the atom's incoming edges w, w+4, .. (CSR by target
//             atom), lane l owns columns 2l, 2l+1; four partial rows are summed through LDS in fixed order (no atomics)
//   adjoint : waves stride over the edges; g_m[e] = scale * g_out[a(e)] (.) (W rbf_e),  g_rbf[e] = scale * W^T (g_out[a(e)] (.) m_e)
//             (a 128 x 16 mat-vec through 512 B of wave-private LDS; a reduce-scatter over the lanes with 17 shuffles
//             instead measured 45 us against 27)
// Constraints: C (columns) == 128, R (radial features) == 16 — the shapes of every published GemNet configuration
// (emb_size_edge 128, emb_size_rbf 16); other shapes take the GEMM + segmented-sum path.
#include "common.h"

#ifdef GN_AGG_SELFCHECK
// Diagnosis build only (tools/exp/agg_selfcheck.py; never defined in the product library).  The adjoint keeps 32 values
// of the CONSTANT weight W per lane in registers for the whole launch; in a replayed multi-queue hipGraph some waves
// produced g_m rows that are only explained by wrong w0[] values in one 16-lane row (profiles/r4_hb_forensics.txt).
// The instrumented kernel re-reads W (volatile: a second, independent load) right after the first load and again at
// every edge, compares with the registers and logs (kind, block, wave, lane, k, register bits, memory bits, HW_ID, XCC_ID):
//   kind 1: the two loads at kernel start disagree          -> the LOAD returned wrong data
//   kind 2: register != memory at the time of use           -> the REGISTER changed after a correct load
//   kind 3: as 2, and a third load still agrees with the second (memory is stable, the register is the odd one)
__device__ unsigned int gn_agg_selfcheck_n;
__device__ unsigned int gn_agg_selfcheck_log[4096][10];
__device__ __forceinline__ void agg_log(unsigned kind, unsigned lane, unsigned k, float reg, float mem, unsigned extra) {
  const unsigned i = atomicAdd(&gn_agg_selfcheck_n, 1u);
  if (i < 4096u) {
    unsigned int* r = gn_agg_selfcheck_log[i];
    r[0] = kind; r[1] = blockIdx.x; r[2] = threadIdx.x >> 6; r[3] = lane; r[4] = k;
    r[5] = __float_as_uint(reg); r[6] = __float_as_uint(mem);
    r[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID: wave / simd / cu / sh / se
    r[8] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
    r[9] = extra;
  }
}
// compare the 32 register-resident weights of this lane with a fresh (volatile) load; one log record per call: the mask of
// mismatching k (bits 0-15: w0, 16-31: w1) in `k`, register / memory bits of the first mismatch; kind 2 -> 3 when a third
// load of that element agrees with the second one
__device__ __forceinline__ void agg_check(const float* W, const float (&w0)[16], const float (&w1)[16], int lane, unsigned kind,
                                          unsigned extra) {
  const volatile float* Wv = W;
  unsigned mask = 0, first = 0;
  float freg = 0.f, fmem = 0.f;
  // (a quarter of the values: checking all 32 took the kernel from 114 to 238 VGPRs — and the symptom was gone)
#pragma unroll
  for (int k = 0; k < 16; k += GN_AGG_SELFCHECK) {
    const float a0 = Wv[(size_t)(2 * lane) * 16 + k], a1 = Wv[(size_t)(2 * lane + 1) * 16 + k];
    const bool m0 = __float_as_uint(a0) != __float_as_uint(w0[k]), m1 = __float_as_uint(a1) != __float_as_uint(w1[k]);
    if (m0 && !mask) { first = k; freg = w0[k]; fmem = a0; }
    if (m0) mask |= 1u << k;
    if (m1 && !mask) { first = 16 + k; freg = w1[k]; fmem = a1; }
    if (m1) mask |= 1u << (16 + k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (mask) {
    if (kind == 2) {
      const float again = first < 16 ? Wv[(size_t)(2 * lane) * 16 + first] : Wv[(size_t)(2 * lane + 1) * 16 + first - 16];
      if (__float_as_uint(again) == __float_as_uint(fmem)) kind = 3;
    }
    agg_log(kind | (first << 8), lane, mask, freg, fmem, extra);
  }
}
extern "C" int gn_agg_selfcheck_read(unsigned int* host, int reset) {
  unsigned int n = 0;
  hipError_t e = hipMemcpyFromSymbol(&n, HIP_SYMBOL(gn_agg_selfcheck_n), sizeof(n));
  if (e != hipSuccess) return -1;
  e = hipMemcpyFromSymbol(host, HIP_SYMBOL(gn_agg_selfcheck_log), sizeof(gn_agg_selfcheck_log));
  if (e != hipSuccess) return -1;
  if (reset) { unsigned int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(gn_agg_selfcheck_n), &z, sizeof(z)); }
  return (int)n;
}
#endif

// NO packed-FP32 VALU instructions in these two kernels.  On gfx950 (ROCm 7.2) the adjoint, compiled with v_pk_mul / v_pk_fma_f32,
// returned wrong g_m values — always the low halves of the packed results in lanes 48..63, i.e. the even columns 96..126 of a
// few rows — when its waves shared CUs with the Dense-stack chain kernels of ANOTHER branch of a replayed hipGraph: constant
// inputs, no memory conflict (gemnet_pytorch_amd/hbcheck.py), never in a single-branch graph or alone.  Minimal repro (two
// kernels, no model): tools/exp/graph_corun.py, 25 / 60 (fp16-plane stacks) and 45 / 60 (bf16-plane stacks) replays wrong;
// with this attribute 0 / 60 (profiles/r4_corun_*.txt).  This was the mechanism behind the three "hipGraph replay != eager"
// findings of round 3 (docs/HISTORY.md section 11).  -DGN_AGG_PK restores the packed instructions for the repro.
#if !defined(GN_AGG_PK) && defined(__HIP_DEVICE_COMPILE__)
#define GN_AGG_ATTR __attribute__((target("no-packed-fp32-ops")))
#else
#define GN_AGG_ATTR
#endif

namespace {

constexpr int C = 128, R = 16;

__global__ __launch_bounds__(256) GN_AGG_ATTR void rbf_aggregate_fwd_kernel(const float* __restrict__ m, const float* __restrict__ rbf,
                                                                const float* __restrict__ W, const int32_t* __restrict__ perm,
                                                                const int32_t* __restrict__ seg_off, float* __restrict__ out,
                                                                float scale) {
  __shared__ float2 part[4][64];
  const int a = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float w0[R], w1[R];
#pragma unroll
  for (int q = 0; q < R / 4; ++q) {
    const float4 u = *reinterpret_cast<const float4*>(W + (size_t)(2 * lane) * R + 4 * q);
    const float4 v = *reinterpret_cast<const float4*>(W + (size_t)(2 * lane + 1) * R + 4 * q);
    w0[4 * q] = u.x; w0[4 * q + 1] = u.y; w0[4 * q + 2] = u.z; w0[4 * q + 3] = u.w;
    w1[4 * q] = v.x; w1[4 * q + 1] = v.y; w1[4 * q + 2] = v.z; w1[4 * q + 3] = v.w;
  }
  const int beg = seg_off[a], end = seg_off[a + 1];
  float2 acc = make_float2(0.f, 0.f);
  // Two edges per trip, every load of the pair issued before the first use: the trip is a chain of dependent loads
  // (perm -> row address -> 512 B row from HBM), and an atom has only ~4 edges per wave to hide it behind.
  // The per-edge products are added in the same order as in the one-edge form (edge i, then edge i + 4).
  for (int i = beg + wave; i < end; i += 8) {
    const bool two = i + 4 < end;
    const int e0 = perm ? perm[i] : i;
    const int e1 = two ? (perm ? perm[i + 4] : i + 4) : e0;
    const float2 m0 = *reinterpret_cast<const float2*>(m + (size_t)e0 * C + 2 * lane);
    const float2 m1 = *reinterpret_cast<const float2*>(m + (size_t)e1 * C + 2 * lane);
    float4 b0[R / 4], b1[R / 4];
#pragma unroll
    for (int q = 0; q < R / 4; ++q) {
      b0[q] = *reinterpret_cast<const float4*>(rbf + (size_t)e0 * R + 4 * q);   // same address in every lane
      b1[q] = *reinterpret_cast<const float4*>(rbf + (size_t)e1 * R + 4 * q);
    }
    float r0 = 0.f, r1 = 0.f, s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < R / 4; ++q) {
      r0 += w0[4 * q] * b0[q].x + w0[4 * q + 1] * b0[q].y + w0[4 * q + 2] * b0[q].z + w0[4 * q + 3] * b0[q].w;
      r1 += w1[4 * q] * b0[q].x + w1[4 * q + 1] * b0[q].y + w1[4 * q + 2] * b0[q].z + w1[4 * q + 3] * b0[q].w;
      s0 += w0[4 * q] * b1[q].x + w0[4 * q + 1] * b1[q].y + w0[4 * q + 2] * b1[q].z + w0[4 * q + 3] * b1[q].w;
      s1 += w1[4 * q] * b1[q].x + w1[4 * q + 1] * b1[q].y + w1[4 * q + 2] * b1[q].z + w1[4 * q + 3] * b1[q].w;
    }
    acc.x += m0.x * r0;
    acc.y += m0.y * r1;
    if (two) {
      acc.x += m1.x * s0;
      acc.y += m1.y * s1;
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    const float2 p1 = part[1][lane], p2 = part[2][lane], p3 = part[3][lane];
    float2 o;
    o.x = ((acc.x + p1.x) + (p2.x + p3.x)) * scale;
    o.y = ((acc.y + p1.y) + (p2.y + p3.y)) * scale;
    *reinterpret_cast<float2*>(out + (size_t)a * C + 2 * lane) = o;
  }
}

// Forward, second form (round 5; same arithmetic in the same order: bit-identical): the wave index is made provably uniform,
// so the edge ids (perm) and the 64-byte rbf rows of a wave's edges are SCALAR loads, and W goes through LDS once per workgroup
// instead of 8 KB per wave from L2.
__global__ __launch_bounds__(256) GN_AGG_ATTR void rbf_aggregate_fwd_kernel_v2(const float* __restrict__ m, const float* __restrict__ rbf,
                                                                   const float* __restrict__ W, const int32_t* __restrict__ perm,
                                                                   const int32_t* __restrict__ seg_off, float* __restrict__ out,
                                                                   float scale) {
  constexpr int WP = R + 1;
  __shared__ float Wl[C * WP];
  __shared__ float2 part[4][64];
  const int a = blockIdx.x, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int beg = seg_off[a], end = seg_off[a + 1];
  for (int i = threadIdx.x; i < C * R / 4; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(W + 4 * i);
    float* d = Wl + (i >> 2) * WP + 4 * (i & 3);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  float w0[R], w1[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    w0[k] = Wl[(2 * lane) * WP + k];
    w1[k] = Wl[(2 * lane + 1) * WP + k];
  }
  float2 acc = make_float2(0.f, 0.f);
  for (int i = beg + wave; i < end; i += 8) {
    const bool two = i + 4 < end;
    const int e0 = perm ? perm[i] : i;
    const int e1 = two ? (perm ? perm[i + 4] : i + 4) : e0;
    const float2 m0 = *reinterpret_cast<const float2*>(m + (size_t)e0 * C + 2 * lane);
    const float2 m1 = *reinterpret_cast<const float2*>(m + (size_t)e1 * C + 2 * lane);
    const float* __restrict__ b0 = rbf + (size_t)e0 * R;      // uniform addresses: scalar loads
    const float* __restrict__ b1 = rbf + (size_t)e1 * R;
    float r0 = 0.f, r1 = 0.f, s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < R / 4; ++q) {
      r0 += w0[4 * q] * b0[4 * q] + w0[4 * q + 1] * b0[4 * q + 1] + w0[4 * q + 2] * b0[4 * q + 2] + w0[4 * q + 3] * b0[4 * q + 3];
      r1 += w1[4 * q] * b0[4 * q] + w1[4 * q + 1] * b0[4 * q + 1] + w1[4 * q + 2] * b0[4 * q + 2] + w1[4 * q + 3] * b0[4 * q + 3];
      s0 += w0[4 * q] * b1[4 * q] + w0[4 * q + 1] * b1[4 * q + 1] + w0[4 * q + 2] * b1[4 * q + 2] + w0[4 * q + 3] * b1[4 * q + 3];
      s1 += w1[4 * q] * b1[4 * q] + w1[4 * q + 1] * b1[4 * q + 1] + w1[4 * q + 2] * b1[4 * q + 2] + w1[4 * q + 3] * b1[4 * q + 3];
    }
    acc.x += m0.x * r0;
    acc.y += m0.y * r1;
    if (two) {
      acc.x += m1.x * s0;
      acc.y += m1.y * s1;
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    const float2 p1 = part[1][lane], p2 = part[2][lane], p3 = part[3][lane];
    float2 o;
    o.x = ((acc.x + p1.x) + (p2.x + p3.x)) * scale;
    o.y = ((acc.y + p1.y) + (p2.y + p3.y)) * scale;
    *reinterpret_cast<float2*>(out + (size_t)a * C + 2 * lane) = o;
  }
}

// Adjoint: a wave walks edges e = wave id, + #waves, ...  Per edge
//   g_m[e][c]   = scale * g_out[a(e)][c] * (W rbf[e])[c]            lane l owns columns 2l, 2l+1 (as in the forward)
//   g_rbf[e][k] = scale * sum_c g_out[a(e)][c] m[e][c] W[c][k]      a 128 x 16 mat-vec: the products t_c pass through
//                 this wave's 512 B of LDS; lane (k = l % 16, part = l / 16) sums its 32 columns against W[c][k] held
//                 in registers, two xor-shuffles fold the four parts (a 64-lane butterfly over 16 values took 96
//                 ds_bpermute per edge: 27 us per launch instead of 11)
__global__ __launch_bounds__(256) GN_AGG_ATTR void rbf_aggregate_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ m,
                                                                const float* __restrict__ rbf, const float* __restrict__ W,
                                                                const int32_t* __restrict__ id_a, float* g_m, float* g_rbf,
                                                                int64_t E, float scale, int accum) {
  __shared__ __attribute__((aligned(16))) float tsm[4][C];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int kq = lane & 15, part = lane >> 4;
  float w0[R], w1[R];          // W rows 2l, 2l+1 (for g_m)
#pragma unroll
  for (int q = 0; q < R / 4; ++q) {
    const float4 u = *reinterpret_cast<const float4*>(W + (size_t)(2 * lane) * R + 4 * q);
    const float4 v = *reinterpret_cast<const float4*>(W + (size_t)(2 * lane + 1) * R + 4 * q);
    w0[4 * q] = u.x; w0[4 * q + 1] = u.y; w0[4 * q + 2] = u.z; w0[4 * q + 3] = u.w;
    w1[4 * q] = v.x; w1[4 * q + 1] = v.y; w1[4 * q + 2] = v.z; w1[4 * q + 3] = v.w;
  }
  float wt[32];                // W[32 part + j][kq] (for g_rbf)
  if (g_rbf) {
#pragma unroll
    for (int j = 0; j < 32; ++j) wt[j] = W[(size_t)(32 * part + j) * R + kq];
  }
#ifdef GN_AGG_SELFCHECK
  agg_check(W, w0, w1, lane, 1, 0u);
#endif
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t e = (int64_t)blockIdx.x * 4 + wave; e < E; e += stride) {
    const int a = id_a[e];
#ifdef GN_AGG_SELFCHECK
    agg_check(W, w0, w1, lane, 2, (unsigned)e);
#endif
    const float2 g = *reinterpret_cast<const float2*>(g_out + (size_t)a * C + 2 * lane);
#ifdef GN_AGG_VSCALE   // experiment: the scale factor from a VGPR instead of an SGPR pair operand of the packed multiply
    float sc;
    asm volatile("v_mov_b32 %0, %1" : "=v"(sc) : "s"(scale));
    const float gx = g.x * sc, gy = g.y * sc;
#else
    const float gx = g.x * scale, gy = g.y * scale;
#endif
    if (g_m) {
      float r0 = 0.f, r1 = 0.f;
#pragma unroll
      for (int q = 0; q < R / 4; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(rbf + (size_t)e * R + 4 * q);
        r0 += w0[4 * q] * b.x + w0[4 * q + 1] * b.y + w0[4 * q + 2] * b.z + w0[4 * q + 3] * b.w;
        r1 += w1[4 * q] * b.x + w1[4 * q + 1] * b.y + w1[4 * q + 2] * b.z + w1[4 * q + 3] * b.w;
      }
      float2 o = make_float2(gx * r0, gy * r1);
      if (accum & 1) {   // running gradient of m (ops.accumulate_gradient): the same lane reads and rewrites its element
        const float2 p = *reinterpret_cast<const float2*>(g_m + (size_t)e * C + 2 * lane);
        o.x += p.x; o.y += p.y;
      }
      *reinterpret_cast<float2*>(g_m + (size_t)e * C + 2 * lane) = o;
    }
    if (g_rbf) {
      const float2 me = *reinterpret_cast<const float2*>(m + (size_t)e * C + 2 * lane);
      *reinterpret_cast<float2*>(&tsm[wave][2 * lane]) = make_float2(gx * me.x, gy * me.y);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      float s = 0.f;
#pragma unroll
      for (int j4 = 0; j4 < 8; ++j4) {
        const float4 t = *reinterpret_cast<const float4*>(&tsm[wave][32 * part + 4 * j4]);
        s += t.x * wt[4 * j4] + t.y * wt[4 * j4 + 1] + t.z * wt[4 * j4 + 2] + t.w * wt[4 * j4 + 3];
      }
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lane < R) g_rbf[(size_t)e * R + lane] = (accum & 2) ? g_rbf[(size_t)e * R + lane] + s : s;
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();   // all lanes have read tsm before the next edge overwrites it
    }
  }
}

// Adjoint, second form (round 5): the same arithmetic in the same order (bit-identical results), restructured around what the
// first form spends its time on — a chain of dependent loads per edge (id_a[e] -> g_out row) issued through the VECTOR memory
// pipe by 8 192 short-lived waves that each fetch the 8 KB weight matrix into registers for ~2 edges of work:
//   * the wave index is made provably uniform (readfirstlane), so the edge id, id_a[e] and the 64-byte rbf row of the edge
//     are SCALAR loads (scalar cache, SGPR operands of the FMAs: no VGPRs, no vector-memory slots);
//   * W goes through LDS once per workgroup (two coalesced float4 loads per thread) instead of 40 loads per lane;
//   * one resident round of waves (grid = 4 waves x 4 workgroups per CU), each walking ~4-5 edges with the NEXT edge's
//     operands requested before the current edge is computed.
__global__ __launch_bounds__(256) GN_AGG_ATTR void rbf_aggregate_bwd_kernel_v2(const float* __restrict__ g_out, const float* __restrict__ m,
                                                                   const float* __restrict__ rbf, const float* __restrict__ W,
                                                                   const int32_t* __restrict__ id_a, float* g_m, float* g_rbf,
                                                                   int64_t E, float scale, int accum) {
  constexpr int WP = R + 1;                                   // LDS row pitch of W: conflict-free column reads
  __shared__ float Wl[C * WP];
  __shared__ __attribute__((aligned(16))) float tsm[4][C];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kq = lane & 15, part = lane >> 4;
  for (int i = threadIdx.x; i < C * R / 4; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(W + 4 * i);
    float* d = Wl + (i >> 2) * WP + 4 * (i & 3);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  float w0[R], w1[R], wt[32];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    w0[k] = Wl[(2 * lane) * WP + k];
    w1[k] = Wl[(2 * lane + 1) * WP + k];
  }
#pragma unroll
  for (int j = 0; j < 32; ++j) wt[j] = Wl[(32 * part + j) * WP + kq];
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= E) return;
  // operands of edge e (requested one trip ahead)
  int a = id_a[e];
  float2 g = *reinterpret_cast<const float2*>(g_out + (size_t)a * C + 2 * lane);
  float2 me = g_rbf ? *reinterpret_cast<const float2*>(m + (size_t)e * C + 2 * lane) : make_float2(0.f, 0.f);
  for (; e < E; e += stride) {
    const int64_t en = e + stride;
    const bool more = en < E;
    const int64_t ec = more ? en : e;                         // (clamped: the loads below are unconditional)
    const int an = id_a[ec];
    const float2 gn = *reinterpret_cast<const float2*>(g_out + (size_t)an * C + 2 * lane);
    const float2 mn = g_rbf ? *reinterpret_cast<const float2*>(m + (size_t)ec * C + 2 * lane) : make_float2(0.f, 0.f);
    const float gx = g.x * scale, gy = g.y * scale;
    if (g_m) {
      const float* __restrict__ b = rbf + (size_t)e * R;      // uniform address: scalar loads
      float r0 = 0.f, r1 = 0.f;
#pragma unroll
      for (int q = 0; q < R / 4; ++q) {
        const float bx = b[4 * q], by = b[4 * q + 1], bz = b[4 * q + 2], bw = b[4 * q + 3];
        r0 += w0[4 * q] * bx + w0[4 * q + 1] * by + w0[4 * q + 2] * bz + w0[4 * q + 3] * bw;
        r1 += w1[4 * q] * bx + w1[4 * q + 1] * by + w1[4 * q + 2] * bz + w1[4 * q + 3] * bw;
      }
      float2 o = make_float2(gx * r0, gy * r1);
      if (accum & 1) {
        const float2 p = *reinterpret_cast<const float2*>(g_m + (size_t)e * C + 2 * lane);
        o.x += p.x; o.y += p.y;
      }
      *reinterpret_cast<float2*>(g_m + (size_t)e * C + 2 * lane) = o;
    }
    if (g_rbf) {
      *reinterpret_cast<float2*>(&tsm[wave][2 * lane]) = make_float2(gx * me.x, gy * me.y);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      float s = 0.f;
#pragma unroll
      for (int j4 = 0; j4 < 8; ++j4) {
        const float4 t = *reinterpret_cast<const float4*>(&tsm[wave][32 * part + 4 * j4]);
        s += t.x * wt[4 * j4] + t.y * wt[4 * j4 + 1] + t.z * wt[4 * j4 + 2] + t.w * wt[4 * j4 + 3];
      }
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lane < R) g_rbf[(size_t)e * R + lane] = (accum & 2) ? g_rbf[(size_t)e * R + lane] + s : s;
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
    a = an; g = gn; me = mn;
  }
}

}  // namespace

extern "C" int gn_rbf_aggregate_fwd_f32(const float* m, const float* rbf, const float* W, const int32_t* perm,
                                        const int32_t* seg_off, float* out, int64_t n_atoms, int C_, int R_, float scale,
                                        void* stream) {
  if (C_ != C || R_ != R) return (int)hipErrorInvalidValue;
  if (n_atoms <= 0) return 0;
#ifdef GN_AGG_V1
  hipLaunchKernelGGL(rbf_aggregate_fwd_kernel, dim3((unsigned)n_atoms), dim3(256), 0, static_cast<hipStream_t>(stream), m, rbf,
                     W, perm, seg_off, out, scale);
#else
  hipLaunchKernelGGL(rbf_aggregate_fwd_kernel_v2, dim3((unsigned)n_atoms), dim3(256), 0, static_cast<hipStream_t>(stream), m, rbf,
                     W, perm, seg_off, out, scale);
#endif
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_rbf_aggregate_bwd_f32(const float* g_out, const float* m, const float* rbf, const float* W,
                                        const int32_t* id_a, float* g_m, float* g_rbf, int64_t E, int C_, int R_, float scale,
                                        int accum, void* stream) {
  if (C_ != C || R_ != R) return (int)hipErrorInvalidValue;
  if (E <= 0) return 0;
  const int64_t blocks = gn_cdiv(E, 4);
#ifdef GN_AGG_V1      // the round-2 form (A/B: tools/exp/agg_v2_bench.py)
  hipLaunchKernelGGL(rbf_aggregate_bwd_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g_out, m, rbf, W, id_a, g_m, g_rbf, E, scale, accum);
#else
  // one resident round: 4 workgroups of 4 waves per CU on 256 CUs (measured on MI355X, profiles/r5_agg_v2.txt: grid 512 / 1024 /
  // 2048 -> 10.6 / 8.5 / 10.6 us at E = 18 122 against 16.7 us for the round-2 form)
#ifndef GN_AGG_GRID
#define GN_AGG_GRID 1024
#endif
  hipLaunchKernelGGL(rbf_aggregate_bwd_kernel_v2, dim3((unsigned)(blocks < GN_AGG_GRID ? blocks : GN_AGG_GRID)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g_out, m, rbf, W, id_a, g_m, g_rbf, E, scale, accum);
#endif
  GN_LAUNCH_CHECK();
  return 0;
}
