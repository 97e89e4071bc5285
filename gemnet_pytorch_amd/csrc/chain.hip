// LDS-resident layer chains (include/gemnet_hip.h, gn_chain_f32).
//
// The reference runs a ResidualLayer / AtomUpdate MLP as 2..6 separate `mm` + `silu` + `mul` + `add`
// launches (gemnet/model/layers/base_layers.py:44-89, atom_update_block.py:60-72), each re-reading
// and re-writing the (E,128) activations.  At a 32-molecule batch every such launch is latency-bound
// (profiles/r1_gemm_tiles.txt: 12.7 us for 3.8 us of matrix-pipe work), so the lever is the number of
// dependent launches, not the inner loop: here a 32-row tile walks the whole stack with its
// activations in LDS, W streamed in 32-deep K-steps through a register-prefetch pipeline
// (v_mfma_f32_32x32x2_f32, exact f32), and only pre-activations (needed by the adjoint) and the
// final rows touch memory.  The adjoint of a stack is another chain program (SCALE ops apply
// ssilu'(z) in LDS, GEMMs use the transposed weights), so forward and backward share this kernel.
#include "common.h"

typedef float v16f __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 32;            // rows per tile
constexpr int SW = 128;           // max slot width
constexpr int SLD = SW + 4;       // slot leading dimension (floats): rows 528 B apart
constexpr int WLD = 36;           // W K-step buffer leading dimension
constexpr int NSLOT = 2;           // 2 x 16.5 KB + 18 KB W buffer = 52 KB -> 3 workgroups per CU
constexpr int NT = 256;

__device__ __forceinline__ float load_sel(int slot, const float* g, const int32_t* rows, float (*S)[BM][SLD],
                                          int row, int64_t grow, int col, int N) {
  if (slot >= 0) return S[slot][row][col];
  const int64_t r = rows ? (int64_t)rows[grow] : grow;
  return g[r * N + col];
}

__global__ __launch_bounds__(NT) void chain_kernel(const gn_chain_args P) {
  __shared__ __attribute__((aligned(16))) float S[NSLOT][BM][SLD];
  __shared__ __attribute__((aligned(16))) float Wb[SW][WLD];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  const int M = P.M;

  // W K-step staging: rows n < N, 32 k's -> 8 float4 per row; 128*8/256 = 4 float4 per thread
  float4 rb[4];
  bool prefetched = false;
  auto wload = [&](const float* __restrict__ W, int N, int K, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + i * NT;
      const int n = f >> 3, kv = (f & 7) << 2;
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N && k0 + kv < K) rb[i] = *reinterpret_cast<const float4*>(W + (size_t)n * K + k0 + kv);
    }
  };
  auto wstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + i * NT;
      *reinterpret_cast<float4*>(&Wb[f >> 3][(f & 7) << 2]) = rb[i];
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
      // dynamically; left as `op.field` the compiler re-issues an s_load + s_waitcnt at every use
      // (329 scalar loads in the first version: ~10 us per GEMM op).
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
      // first W K-step of the NEXT GEMM op of the program (prefetched during this op's tail)
      const float* nW = nullptr;
      int nN = 0, nK = 0;
      for (int oj = oi + 1; oj < P.n_ops; ++oj)
        if (P.ops[oj].kind == GN_OP_GEMM) { nW = P.ops[oj].W; nN = P.ops[oj].N; nK = P.ops[oj].K; break; }

      const bool active = wave * 32 < N;   // this wave's 32 output columns exist
      v16f acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int arow = lane & 31;
      const int brow = wave * 32 + (lane & 31);
      const int kh = (lane >> 5) << 2;
      if (!prefetched) wload(W, N, K, 0);
      prefetched = false;
      for (int k0 = 0; k0 < K; k0 += 32) {
        wstore();
        __syncthreads();
        if (k0 + 32 < K) {
          wload(W, N, K, k0 + 32);
        } else if (nW) {
          // last K-step: start streaming the first W K-step of the next GEMM so its L2 latency
          // hides under this step's MFMAs, the epilogue and any SCALE ops in between
          wload(nW, nN, nK, 0);
          prefetched = true;
        }
        if (active) {
          const int kend = min(32, K - k0);
          for (int kb = 0; kb < kend; kb += 8) {
            const float4 a = *reinterpret_cast<const float4*>(&S[a_slot][arow][k0 + kb + kh]);
            const float4 b = *reinterpret_cast<const float4*>(&Wb[brow][kb + kh]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
          }
        }
        __syncthreads();
      }
      // epilogue (all waves are past their last read of a_slot: y_slot may alias it)
      if (active) {
        const int col = wave * 32 + (lane & 31);
        const int rh = (lane >> 5) << 2;
        if (col < N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + rh;
            const int64_t gr = row0 + row;
            const bool ok = gr < M;
            float y = 0.f;
            if (ok) {
              float z = acc[r];
              if (gadd1) z += gadd1[(size_t)gidx1[gr] * N + col];
              if (gadd2) z += gadd2[(size_t)gidx2[gr] * N + col];
              if (pre_out) pre_out[gr * N + col] = z;
              y = act ? gn_ssilu(z) : z;
              if (has_mul) y *= load_sel(mul_slot, mul_g, nullptr, S, row, gr, col, N);
              y *= alpha;
              if (has_res) y = (y + load_sel(res_slot, res_g, res_rows, S, row, gr, col, N)) * beta;
              if (has_res2) y = (y + load_sel(res2_slot, res2_g, nullptr, S, row, gr, col, N)) * beta2;
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

}  // namespace

extern "C" int gn_chain_f32(const gn_chain_args* args, void* stream) {
  if (args->M <= 0 || args->n_ops <= 0) return 0;
  if (args->n_ops > GN_CHAIN_MAX_OPS) return (int)hipErrorInvalidValue;
  for (int i = 0; i < args->n_ops; ++i) {
    const gn_chain_op& o = args->ops[i];
    if (o.kind == GN_OP_GEMM) {
      if (o.N <= 0 || o.N > SW || o.K <= 0 || o.K > SW || (o.K % 8) != 0) return (int)hipErrorInvalidValue;
      if (o.a_slot < 0 || o.a_slot >= NSLOT || o.slot >= NSLOT) return (int)hipErrorInvalidValue;
    } else {
      if (o.width <= 0 || o.width > SW || (o.width % 4) != 0 || (o.ld % 4) != 0) return (int)hipErrorInvalidValue;
      if (o.slot < 0 || o.slot >= NSLOT) return (int)hipErrorInvalidValue;
    }
  }
  hipLaunchKernelGGL(chain_kernel, dim3(gn_cdiv(args->M, BM)), dim3(NT), 0, static_cast<hipStream_t>(stream), *args);
  GN_LAUNCH_CHECK();
  return 0;
}
