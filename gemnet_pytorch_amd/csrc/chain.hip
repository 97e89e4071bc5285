// LDS-resident layer chains (include/gemnet_hip.h, gn_chain_f32).
//
// The reference runs a ResidualLayer / AtomUpdate MLP as 2..6 separate `mm` + `silu` + `mul` + `add`
// launches (gemnet/model/layers/base_layers.py:44-89, atom_update_block.py:60-72), each re-reading
// and re-writing the (E,128) activations.  At a 32-molecule batch (E = 18 k rows) a stand-alone
// 128x128 layer is bound by fixed costs, not by the matrix pipe (profiles/r1_gemm_tiles.txt: 12.7 us
// for 3.8 us of MFMA work): ~3.5 us per graph node, a load->LDS->MFMA->store latency chain per
// workgroup, and — with 32-row tiles — 567 workgroups each pulling the same 64 KB weight matrix
// through its CU's L2 port (36 MB per layer at ~6.5 TB/s chip-wide fill rate).  Here a row tile walks
// the whole stack with its activations in LDS:
//   * tile height 16*RT rows, RT = 1..5 chosen so that the launch is ONE round of <= 256 workgroups
//     (E = 18 k -> 80 rows, 227 workgroups, one per CU): each CU pulls each W exactly once;
//   * 8 waves, v_mfma_f32_16x16x4_f32 (exact f32): wave w owns output columns 16w..16w+15 for all RT
//     row blocks -> one B fragment feeds RT independent accumulators (no dependent-issue stalls);
//   * a wave only ever needs ITS 16 rows of W: they are loaded from L2 straight into registers in MFMA
//     fragment layout (8 float4 per lane for K = 128) — W never passes through LDS — and the NEXT GEMM
//     op's fragments are prefetched while the current op's MFMAs run;
//   * one barrier per GEMM op (two when the output slot aliases the input slot);
//   * only pre-activations (needed by the adjoint) and final rows touch memory.
// The adjoint of a stack is another chain program (SCALE ops apply ssilu'(z) in LDS, GEMMs use the
// transposed weights), so forward and backward share this kernel.
// LDS: 2 slots x (16 RT) x 132 floats = 84 KB at RT = 5 (dynamic, opt-in above 64 KB).
#include "common.h"

typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

constexpr int SW = 128;           // max slot width / max N, K of a GEMM op
constexpr int SLD = SW + 4;       // slot leading dimension (floats): rows 528 B apart
constexpr int NSLOT = 2;
constexpr int NT = 512;

template <int RT>
__global__ __launch_bounds__(NT) void chain_kernel(const gn_chain_args P) {
  constexpr int BM = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float (*S)[BM][SLD] = reinterpret_cast<float (*)[BM][SLD]>(smem);
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int l15 = lane & 15;
  const int lg = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  const int M = P.M;

  auto sel = [&](int slot, const float* g, const int32_t* rows, int row, int64_t grow, int col, int N) -> float {
    if (slot >= 0) return S[slot][row][col];
    const int64_t r = rows ? (int64_t)rows[grow] : grow;
    return g[r * N + col];
  };

  // this wave's B fragments of one GEMM op: W rows 16 wave .. +15, lane (l15, lg) holds
  // W[16 wave + l15][kc + 4 lg .. +3] for the 16-k chunks kc = 0, 16, .. (K <= 128 -> 8 float4)
  float4 bcur[8], bnext[8];
  bool prefetched = false;
  auto wload = [&](float4 (&dst)[8], const float* __restrict__ W, int N, int K) {
    const int n = wave * 16 + l15;
    const float* __restrict__ row = W + (size_t)n * K + (lg << 2);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      dst[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N && c * 16 < K) dst[c] = *reinterpret_cast<const float4*>(row + c * 16);
    }
  };

  for (int oi = 0; oi < P.n_ops; ++oi) {
    const gn_chain_op& op = P.ops[oi];
    const int kind = op.kind;
    if (kind == GN_OP_LOAD) {
      const int w4 = op.width >> 2, slot = op.slot, ld = op.ld;
      const float* __restrict__ const src = op.src;
      const int32_t* __restrict__ const rows = op.rows;
      for (int f = tid; f < BM * w4; f += NT) {
        const int r = f / w4, c = (f - r * w4) << 2;
        const int64_t gr = row0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < M) {
          const int64_t sr = rows ? (int64_t)rows[gr] : gr;
          v = *reinterpret_cast<const float4*>(src + sr * ld + c);
        }
        *reinterpret_cast<float4*>(&S[slot][r][c]) = v;
      }
      __syncthreads();
    } else if (kind == GN_OP_SCALE) {
      const int w4 = op.width >> 2, slot = op.slot, a_slot = op.a_slot, ld = op.ld;
      const float alpha = op.alpha;
      const float* __restrict__ const src = op.src;
      float* __restrict__ const out = op.out;
      for (int f = tid; f < BM * w4; f += NT) {
        const int r = f / w4, c = (f - r * w4) << 2;
        const int64_t gr = row0 + r;
        float4 v = *reinterpret_cast<const float4*>(&S[a_slot][r][c]);
        v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
        if (src && gr < M) {
          const float4 z = *reinterpret_cast<const float4*>(src + gr * ld + c);
          v.x *= gn_dssilu(z.x); v.y *= gn_dssilu(z.y); v.z *= gn_dssilu(z.z); v.w *= gn_dssilu(z.w);
        }
        *reinterpret_cast<float4*>(&S[slot][r][c]) = v;
        if (out && gr < M) *reinterpret_cast<float4*>(out + gr * ld + c) = v;
      }
      __syncthreads();
    } else if (kind == GN_OP_STORE) {
      const int w4 = op.width >> 2, slot = op.slot, ld = op.ld;
      float* __restrict__ const out = op.out;
      for (int f = tid; f < BM * w4; f += NT) {
        const int r = f / w4, c = (f - r * w4) << 2;
        const int64_t gr = row0 + r;
        if (gr < M) *reinterpret_cast<float4*>(out + gr * ld + c) = *reinterpret_cast<const float4*>(&S[slot][r][c]);
      }
      __syncthreads();
    } else {  // GN_OP_GEMM
      // Copy the op descriptor into registers once: the kernarg struct is large and indexed
      // dynamically; left as `op.field` the compiler re-issues an s_load + s_waitcnt at every use.
      const float* __restrict__ const W = op.W;
      const int N = op.N, K = op.K, a_slot = op.a_slot, y_slot = op.slot, act = op.act;
      const float alpha = op.alpha, beta = op.beta, beta2 = op.beta2;
      const float* __restrict__ const gadd1 = op.gadd1;
      const float* __restrict__ const gadd2 = op.gadd2;
      const int32_t* __restrict__ const gidx1 = op.gidx1;
      const int32_t* __restrict__ const gidx2 = op.gidx2;
      float* __restrict__ const pre_out = op.pre_out;
      float* __restrict__ const out = op.out;
      const int mul_slot = op.mul_slot, res_slot = op.res_slot, res2_slot = op.res2_slot;
      const float* __restrict__ const mul_g = op.mul_g;
      const float* __restrict__ const res_g = op.res_g;
      const float* __restrict__ const res2_g = op.res2_g;
      const int32_t* __restrict__ const res_rows = op.res_rows;
      const bool has_mul = mul_slot >= 0 || mul_g, has_res = res_slot >= 0 || res_g, has_res2 = res2_slot >= 0 || res2_g;
      const float* nW = nullptr;
      int nN = 0, nK = 0;
      for (int oj = oi + 1; oj < P.n_ops; ++oj)
        if (P.ops[oj].kind == GN_OP_GEMM) { nW = P.ops[oj].W; nN = P.ops[oj].N; nK = P.ops[oj].K; break; }

      if (prefetched) {
#pragma unroll
        for (int c = 0; c < 8; ++c) bcur[c] = bnext[c];
      } else {
        wload(bcur, W, N, K);
      }
      prefetched = false;
      if (nW) {        // the next GEMM's fragments fly in under this op's MFMAs and epilogue
        wload(bnext, nW, nN, nK);
        prefetched = true;
      }
      const bool active = wave * 16 < N;   // this wave's 16 output columns exist
      v4f acc[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
      if (active) {
        const int kq = lg << 2;
        // lanes of group lg supply k = kc + 4 lg + j to MFMA j (same permutation for A and B)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (c * 16 < K) {
            const float4 b = bcur[c];
            float4 a[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) a[t] = *reinterpret_cast<const float4*>(&S[a_slot][16 * t + l15][c * 16 + kq]);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, b.x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, b.y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, b.z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, b.w, acc[t], 0, 0, 0);
          }
        }
      }
      if (y_slot == a_slot) __syncthreads();   // all reads of a_slot must finish before it is overwritten
      if (active) {
        const int col = wave * 16 + l15;   // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
        for (int t = 0; t < RT; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + (lg << 2) + r;
            const int64_t gr = row0 + row;
            float y = 0.f;
            if (gr < M) {
              float z = acc[t][r];
              if (gadd1) z += gadd1[(size_t)gidx1[gr] * N + col];
              if (gadd2) z += gadd2[(size_t)gidx2[gr] * N + col];
              if (pre_out) pre_out[gr * N + col] = z;
              y = act ? gn_ssilu(z) : z;
              if (has_mul) y *= sel(mul_slot, mul_g, nullptr, row, gr, col, N);
              y *= alpha;
              if (has_res) y = (y + sel(res_slot, res_g, res_rows, row, gr, col, N)) * beta;
              if (has_res2) y = (y + sel(res2_slot, res2_g, nullptr, row, gr, col, N)) * beta2;
              if (out) out[gr * N + col] = y;
            }
            if (y_slot >= 0) S[y_slot][row][col] = y;
          }
        }
      }
      __syncthreads();
    }
  }
}

template <int RT>
int launch_chain(const gn_chain_args* args, hipStream_t st) {
  constexpr int BM = 16 * RT;
  constexpr size_t smem = (size_t)NSLOT * BM * SLD * sizeof(float);
  static bool configured = false;   // idempotent attribute; a benign race sets it twice
  if (!configured) {
    if (smem > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel<RT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
    }
    configured = true;
  }
  hipLaunchKernelGGL((chain_kernel<RT>), dim3(gn_cdiv(args->M, BM)), dim3(NT), smem, st, *args);
  GN_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int gn_chain_f32(const gn_chain_args* args, void* stream) {
  if (args->M <= 0 || args->n_ops <= 0) return 0;
  if (args->n_ops > GN_CHAIN_MAX_OPS) return (int)hipErrorInvalidValue;
  for (int i = 0; i < args->n_ops; ++i) {
    const gn_chain_op& o = args->ops[i];
    if (o.kind == GN_OP_GEMM) {
      if (o.N <= 0 || o.N > SW || o.K <= 0 || o.K > SW || (o.K % 16) != 0) return (int)hipErrorInvalidValue;
      if (o.a_slot < 0 || o.a_slot >= NSLOT || o.slot >= NSLOT) return (int)hipErrorInvalidValue;
      if ((reinterpret_cast<uintptr_t>(o.W) & 15u) != 0) return (int)hipErrorInvalidValue;
    } else {
      if (o.width <= 0 || o.width > SW || (o.width % 4) != 0 || (o.ld % 4) != 0) return (int)hipErrorInvalidValue;
      if (o.slot < 0 || o.slot >= NSLOT) return (int)hipErrorInvalidValue;
    }
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  // one round of <= 256 workgroups (one per CU) whenever 80-row tiles allow it
  const int rt = gn_cdiv(args->M, 256 * 16);
  switch (rt <= 1 ? 1 : (rt >= 5 ? 5 : rt)) {
    case 1: return launch_chain<1>(args, st);
    case 2: return launch_chain<2>(args, st);
    case 3: return launch_chain<3>(args, st);
    case 4: return launch_chain<4>(args, st);
    default: return launch_chain<5>(args, st);
  }
}
