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
// LDS: 3 slots x (16 RT) x 132 floats = 127 KB at RT = 5 (dynamic, opt-in above 64 KB).  Two slots carry the
// alternating activations; the third parks a tensor across several ops (the gradient of a skip connection).
#include "common.h"

typedef float v4f __attribute__((ext_vector_type(4)));

#ifdef GN_CHAIN_TRACE
// diagnosis build only (tools/chain_trace.py): shader-clock stamps of wave 0 of two workgroups
__device__ unsigned long long gn_chain_trace_buf[2][GN_CHAIN_MAX_OPS][8];
#define GN_STAMP(i) do { if (lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) \
    gn_chain_trace_buf[blockIdx.x == 100][oi][i] = clock64(); } while (0)
#else
#define GN_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int SW = 128;           // max slot width / max N, K of a GEMM op
constexpr int SLD = SW + 4;       // slot leading dimension (floats): rows 528 B apart
constexpr int NSLOT = 3;
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
  // op indices of the GEMM ops in program order (their weights are prefetched two GEMMs ahead), resolved once
  __shared__ int gemm_list[GN_CHAIN_MAX_OPS + 2];
  if (tid == 0) {
    int g = 0;
    for (int j = 0; j < P.n_ops; ++j)
      if (P.ops[j].kind == GN_OP_GEMM) gemm_list[g++] = j;
    gemm_list[g] = gemm_list[g + 1] = -1;
  }
  __syncthreads();

  // this wave's B fragments of one GEMM op: W rows 16 wave .. +15, lane (l15, lg) holds
  // W[16 wave + l15][kc + 4 lg .. +3] for the 16-k chunks kc = 0, 16, .. (K <= 128 -> 8 float4)
  float4 bcur[8], bnext[8];
  // Weight distribution is the slow resource of this kernel: every CU pulls the same 64 KB per op and the
  // L2 -> CU path delivers it in ~8-14 k cycles when all CUs ask at once (traced with -DGN_CHAIN_TRACE), about
  // one op's MFMA phase.  The next GEMM's fragments are in flight during the current op (bnext); the first GEMM's
  // are requested before the program's LOAD ops.  (Two-deep prefetch was tried: a register rotation by copies
  // waits on the youngest loads, a static 3-buffer rotation triples the MFMA code and spills at RT >= 4.)
  auto wload = [&](float4 (&dst)[8], const float* __restrict__ W, int N, int K) {
    const int n = wave * 16 + l15;
    const float* __restrict__ row = W + (size_t)n * K + (lg << 2);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      dst[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N && c * 16 < K) dst[c] = *reinterpret_cast<const float4*>(row + c * 16);
    }
  };
  auto wload_op = [&](float4 (&dst)[8], int ord) {
    const int j = __builtin_amdgcn_readfirstlane(gemm_list[ord]);
    if (j >= 0) wload(dst, P.ops[j].W, P.ops[j].N, P.ops[j].K);
  };
  wload_op(bnext, 0);   // in flight under the LOAD op(s) that start every program
  int gord = 0;         // ordinal of the next GEMM op

  for (int oi = 0; oi < P.n_ops; ++oi) {
    const gn_chain_op& op = P.ops[oi];
    const int kind = op.kind;
    if (kind == GN_OP_LOAD) {
      const int w4 = op.width >> 2, slot = op.slot, ld = op.ld, y2_slot = op.y2_slot, mode2 = op.mode2, width = op.width;
      const float alpha = op.alpha, alpha2 = op.alpha2;
      const float* __restrict__ const src = op.src;
      const float* __restrict__ const Z2 = op.Z2;
      const int32_t* __restrict__ const rows = op.rows;
      for (int f = tid; f < BM * w4; f += NT) {
        const int r = f / w4, c = (f - r * w4) << 2;
        const int64_t gr = row0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < M) {
          const int64_t sr = rows ? (int64_t)rows[gr] : gr;
          v = *reinterpret_cast<const float4*>(src + sr * ld + c);
          v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
        }
        *reinterpret_cast<float4*>(&S[slot][r][c]) = v;
        if (y2_slot >= 0) {   // second tensor derived from the loaded rows: v * alpha2 * phi2(Z2)
          float4 u = make_float4(v.x * alpha2, v.y * alpha2, v.z * alpha2, v.w * alpha2);
          if (Z2 && gr < M) {
            const float4 z = *reinterpret_cast<const float4*>(Z2 + gr * width + c);
            if (mode2 == 0) { u.x *= gn_dssilu(z.x); u.y *= gn_dssilu(z.y); u.z *= gn_dssilu(z.z); u.w *= gn_dssilu(z.w); }
            else if (mode2 == 1) { u.x *= z.x; u.y *= z.y; u.z *= z.z; u.w *= z.w; }
            else { u.x *= gn_ssilu(z.x); u.y *= gn_ssilu(z.y); u.z *= gn_ssilu(z.z); u.w *= gn_ssilu(z.w); }
          }
          *reinterpret_cast<float4*>(&S[y2_slot][r][c]) = u;
        }
      }
      __syncthreads();
    } else if (kind == GN_OP_SCALE) {
      const int w4 = op.width >> 2, slot = op.slot, a_slot = op.a_slot, ld = op.ld;
      const float alpha = op.alpha;
      const int mode = op.act;   // factor taken from src: 0 ssilu'(src), 1 src, 2 ssilu(src)
      const float* __restrict__ const src = op.src;
      float* __restrict__ const out = op.out;
      for (int f = tid; f < BM * w4; f += NT) {
        const int r = f / w4, c = (f - r * w4) << 2;
        const int64_t gr = row0 + r;
        float4 v = *reinterpret_cast<const float4*>(&S[a_slot][r][c]);
        v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
        if (src && gr < M) {
          const float4 z = *reinterpret_cast<const float4*>(src + gr * ld + c);
          if (mode == 0) { v.x *= gn_dssilu(z.x); v.y *= gn_dssilu(z.y); v.z *= gn_dssilu(z.z); v.w *= gn_dssilu(z.w); }
          else if (mode == 1) { v.x *= z.x; v.y *= z.y; v.z *= z.z; v.w *= z.w; }
          else { v.x *= gn_ssilu(z.x); v.y *= gn_ssilu(z.y); v.z *= gn_ssilu(z.z); v.w *= gn_ssilu(z.w); }
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
      const int mul_mode = op.mul_mode, y2_slot = op.y2_slot, y2_src = op.y2_src, mode2 = op.mode2;
      const float alpha2 = op.alpha2;
      const float* __restrict__ const Z2 = op.Z2;
      float* __restrict__ const out2 = op.out2;
      GN_STAMP(0);
      const bool active = wave * 16 < N;   // this wave's 16 output columns exist
      // RT = 1 has a single 16x16 tile per wave: its 32 MFMAs would form one dependent chain, so even and odd
      // K-chunks accumulate separately (two independent chains, summed at the end)
      constexpr int KSP = RT == 1 ? 2 : 1;
      v4f accs[KSP][RT];
#pragma unroll
      for (int q = 0; q < KSP; ++q)
#pragma unroll
        for (int t = 0; t < RT; ++t) accs[q][t] = (v4f){0.f, 0.f, 0.f, 0.f};
      // lanes of group lg supply k = kc + 4 lg + j to MFMA j (same permutation for A and B)
#pragma unroll
      for (int c = 0; c < 8; ++c) bcur[c] = bnext[c];   // requested one GEMM ago (or at kernel start)
      wload_op(bnext, gord + 1);                          // the next GEMM's fragments fly in under this op
      ++gord;
#ifdef GN_CHAIN_TRACE
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // bcur landed (the 8 newer loads are bnext)
#endif
      GN_STAMP(1);
      auto mfma_phase = [&]() {
        if (!active) return;
        const int kq = lg << 2;
        auto chunk = [&](int c, const float4 (&a)[RT]) {
          const float4 b = bcur[c];
          v4f (&acc)[RT] = accs[c & (KSP - 1)];
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, b.x, acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, b.y, acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, b.z, acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, b.w, acc[t], 0, 0, 0);
        };
        auto afrag = [&](int c, float4 (&a)[RT]) {
#pragma unroll
          for (int t = 0; t < RT; ++t) a[t] = *reinterpret_cast<const float4*>(&S[a_slot][16 * t + l15][c * 16 + kq]);
        };
        if (K == SW) {
          // straight-line, double-buffered fragments: chunk c+1's ds_reads are issued before chunk c's MFMAs
          float4 a0[RT], a1[RT];
          afrag(0, a0);
#pragma unroll
          for (int c = 0; c < 8; c += 2) {
            afrag(c + 1, a1);
            chunk(c, a0);
            if (c + 2 < 8) afrag(c + 2, a0);
            chunk(c + 1, a1);
          }
        } else {
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            if (c * 16 < K) {
              float4 a[RT];
              afrag(c, a);
              chunk(c, a);
            }
          }
        }
      };
      mfma_phase();
      v4f acc[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = KSP == 2 ? accs[0][t] + accs[KSP - 1][t] : accs[0][t];
#ifdef GN_CHAIN_TRACE
      if (active) { float sink = 0.f; for (int t = 0; t < RT; ++t) sink += acc[t][0]; if (sink == 1.2345e30f) S[0][0][0] = sink; }
#endif
      GN_STAMP(2);
      if (y_slot == a_slot || y2_slot == a_slot) __syncthreads();   // all reads of a_slot must finish before it is overwritten
      if (active) {
        // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg.  The epilogue is written
        // stage-major (one uniform branch per stage, the RT*4 values of a stage unrolled and independent) — the
        // value-major form serialised ~8 branches and up to 3 flat loads per value (10 k cycles per op at RT = 5).
        const int col = wave * 16 + l15;
        const int rbase = lg << 2;
        const int64_t growb = row0 + rbase;                    // global row of (t = 0, r = 0)
        const uint32_t off0 = (uint32_t)growb * (uint32_t)N + (uint32_t)col;
        GN_STAMP(6);
        float v[RT][4];
        bool ok[RT][4];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[t][r] = acc[t][r];
            ok[t][r] = growb + 16 * t + r < M;
          }
#define GN_EACH(body)                                                     \
  _Pragma("unroll") for (int t = 0; t < RT; ++t)                          \
  _Pragma("unroll") for (int r = 0; r < 4; ++r) {                         \
    const int row = 16 * t + rbase + r;                                   \
    const uint32_t off = off0 + (uint32_t)(16 * t + r) * (uint32_t)N;     \
    (void)row; (void)off;                                                 \
    body                                                                  \
  }
        if (gadd1) GN_EACH(if (ok[t][r]) v[t][r] += gadd1[(size_t)gidx1[row0 + row] * N + col];)
        if (gadd2) GN_EACH(if (ok[t][r]) v[t][r] += gadd2[(size_t)gidx2[row0 + row] * N + col];)
        if (pre_out) GN_EACH(if (ok[t][r]) pre_out[off] = v[t][r];)
        if (act) GN_EACH(v[t][r] = gn_ssilu(v[t][r]);)
        const bool want2 = y2_slot >= 0 || out2;
        float v2[RT][4];
        if (want2 && y2_src) GN_EACH(v2[t][r] = v[t][r];)
        if (mul_slot >= 0) GN_EACH(v[t][r] *= S[mul_slot][row][col];)
        else if (mul_g) {
          if (mul_mode == 2) GN_EACH(if (ok[t][r]) v[t][r] *= gn_dssilu(mul_g[off]);)
          else if (mul_mode == 3) GN_EACH(if (ok[t][r]) v[t][r] *= gn_ssilu(mul_g[off]);)
          else GN_EACH(if (ok[t][r]) v[t][r] *= mul_g[off];)
        }
        if (alpha != 1.0f) GN_EACH(v[t][r] *= alpha;)
        if (res_slot >= 0) GN_EACH(v[t][r] = (v[t][r] + S[res_slot][row][col]) * beta;)
        else if (res_g) {
          if (res_rows) GN_EACH(if (ok[t][r]) v[t][r] = (v[t][r] + res_g[(size_t)res_rows[row0 + row] * N + col]) * beta;)
          else GN_EACH(if (ok[t][r]) v[t][r] = (v[t][r] + res_g[off]) * beta;)
        }
        if (res2_slot >= 0) GN_EACH(v[t][r] = (v[t][r] + S[res2_slot][row][col]) * beta2;)
        else if (res2_g) GN_EACH(if (ok[t][r]) v[t][r] = (v[t][r] + res2_g[off]) * beta2;)
        GN_STAMP(5);
        if (out) GN_EACH(if (ok[t][r]) out[off] = v[t][r];)
        if (y_slot >= 0) GN_EACH(S[y_slot][row][col] = ok[t][r] ? v[t][r] : 0.f;)
        if (want2) {
          if (!y2_src) GN_EACH(v2[t][r] = v[t][r];)
          GN_EACH(v2[t][r] *= alpha2;)
          if (Z2) {
            if (mode2 == 0) GN_EACH(if (ok[t][r]) v2[t][r] *= gn_dssilu(Z2[off]);)
            else if (mode2 == 1) GN_EACH(if (ok[t][r]) v2[t][r] *= Z2[off];)
            else GN_EACH(if (ok[t][r]) v2[t][r] *= gn_ssilu(Z2[off]);)
          }
          if (out2) GN_EACH(if (ok[t][r]) out2[off] = v2[t][r];)
          if (y2_slot >= 0) GN_EACH(S[y2_slot][row][col] = ok[t][r] ? v2[t][r] : 0.f;)
        }
#undef GN_EACH
      }
      GN_STAMP(3);
      __syncthreads();
      GN_STAMP(4);
    }
  }
}

template <int RT>
int launch_chain(const gn_chain_args* args, hipStream_t st) {
  constexpr int BM = 16 * RT;
  constexpr size_t smem = (size_t)NSLOT * BM * SLD * sizeof(float);
  static std::atomic<bool> configured{false};   // set-once flag of an idempotent attribute (two racing threads both set it)
  if (!configured.load(std::memory_order_acquire)) {
    if (smem > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel<RT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
    }
    configured.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((chain_kernel<RT>), dim3(gn_cdiv(args->M, BM)), dim3(NT), smem, st, *args);
  GN_LAUNCH_CHECK();
  return 0;
}

}  // namespace

#ifdef GN_CHAIN_TRACE
extern "C" int gn_chain_trace_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gn_chain_trace_buf), sizeof(gn_chain_trace_buf));
}
#endif

extern "C" int gn_chain_f32(const gn_chain_args* args, void* stream) {
  if (args->M <= 0 || args->n_ops <= 0) return 0;
  if (args->n_ops > GN_CHAIN_MAX_OPS) return (int)hipErrorInvalidValue;
  if (args->M > (1 << 24)) return (int)hipErrorInvalidValue;   // 32-bit element offsets (M * 128 < 2^32)
  for (int i = 0; i < args->n_ops; ++i) {
    const gn_chain_op& o = args->ops[i];
    // the second-order source terms (training step) exist on the split-operand kernel only: fail loudly, never drop them
    if (o.src_stage != 0) return (int)hipErrorInvalidValue;
    if (o.kind == GN_OP_GEMM && (o.act & ~1) != 0) return (int)hipErrorInvalidValue;   // pre_out = ssilu'(z): chain2.hip only
    if (o.kind == GN_OP_GEMM) {
      if (o.N <= 0 || o.N > SW || o.K <= 0 || o.K > SW || (o.K % 16) != 0) return (int)hipErrorInvalidValue;
      if (o.a_slot < 0 || o.a_slot >= NSLOT || o.slot >= NSLOT || o.y2_slot >= NSLOT) return (int)hipErrorInvalidValue;
      if (o.y2_slot >= 0 && (o.y2_slot == o.slot || o.y2_slot == o.mul_slot || o.y2_slot == o.res_slot ||
                             o.y2_slot == o.res2_slot)) return (int)hipErrorInvalidValue;
      if ((reinterpret_cast<uintptr_t>(o.W) & 15u) != 0) return (int)hipErrorInvalidValue;
    } else {
      if (o.width <= 0 || o.width > SW || (o.width % 4) != 0 || (o.ld % 4) != 0) return (int)hipErrorInvalidValue;
      if (o.slot < 0 || o.slot >= NSLOT) return (int)hipErrorInvalidValue;
      if (o.kind == GN_OP_LOAD && (o.y2_slot >= NSLOT || o.y2_slot == o.slot)) return (int)hipErrorInvalidValue;
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
