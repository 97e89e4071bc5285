// Row-resident layout of the two-plane fp16 chain arithmetic (include/gemnet_hip.h: gn_chain_split_f32 with
// nprod = GN_CHAIN_F16X2 | GN_CHAIN_ROW).  Replaces: Dense / ResidualLayer stacks (base_layers.py:44-89) and their adjoints,
// exactly like chain2.hip — same programs, same op semantics.
//
// Why another layout.  chain2.hip gives a wave 16 output COLUMNS of an 80-row tile: every layer ends in a workgroup barrier
// (the next layer's K dimension spans all waves' columns), activations make an LDS round trip per layer, and an op is a
// per-wave chain of latencies (descriptor, weight wait, fragment reads -> MFMA, epilogue, barrier: 7.2 k cycles for 1 k
// cycles of matrix-pipe work).  Here a wave owns 16 ROWS and all 128 columns:
//   * the MFMA computes the transposed tile D[n][m] (A = 16 weight rows, B = the wave's 16 activation rows), so lane
//     (m = lane % 16, g = lane / 16) ends up with columns 16 j + 4 g .. + 3 of row m for every column tile j — and the B
//     operand of the NEXT layer wants 8 K-values per lane of that same row m.  The order of K inside a dot product is
//     free as long as both operands agree, so the weights are packed (gn_pack_weight_split_fmt, GN_SPLIT_F16X2_ROW) with
//     the K order in which the accumulators already hold the activations: chunk c, lane group g, element e  <->
//     k = 16 (2 c + e / 4) + 4 g + e % 4.  A layer's output registers ARE the next layer's B fragments after the fp16
//     split: activations never leave registers, no transposes, no cross-lane traffic, no barrier between the waves' data;
//   * the three "slots" of a program are fp32 register arrays (8 float4 each) — exact fp32 residual stream; nothing that
//     stays resident is rounded to the planes, so the "h3 hazards" of chain2.hip (a foreign global tensor added into a
//     row whose fp16 scale was fixed at LOAD time) do not exist;
//   * the fp16 planes are formed per GEMM from the operand as it is, under a fresh power-of-two ROW scale
//     (sigma * max|row| in [2^-4, 2^-3)): always, forward programs included — an activation beyond 65 504 cannot overflow;
//   * only the weights go through LDS: one extra LOADER wave per workgroup streams the packed planes of GEMM g + 1
//     (<= 64 KB, contiguous) with global_load_lds_dwordx4 into the other half of a 128 KB double buffer while the compute
//     waves run GEMM g; one barrier per GEMM op (weights ready / previous buffer free).  Compute waves never wait for a
//     vmcnt of the weight stream, and the stores of their epilogues are never drained by a barrier;
//   * a workgroup has ceil(row blocks / 256) (<= 7) compute waves: 18 122 rows = 1 133 row blocks = 227 workgroups of 5,
//     1 024 rows = 64 workgroups of one.
// Per 128 x 128 layer a wave issues 96 v_mfma_f32_16x16x32_f16 (1.6 k cycles of its SIMD's matrix pipe) behind
// 64 ds_read_b128 of weight fragments.
//
// MEASURED (round 6, tools/chain4_trace.py, profiles/r6_chain4_*.txt): exact, and SLOWER than chain2.hip at
// the batch sizes of BASELINE.json — 1 675 vs 1 102 us for the 50 chain launches of a forward+force step.  (i) A lone wave's
// ds_read_b128 stream runs at ~28 B/clk, so the 64 KB of weight fragments of one layer take 3.5 k cycles (with the reads
// knocked out the phase takes exactly 96 x 17 cycles); (ii) 18 122 rows are 1 133 sixteen-row blocks on 1 024 SIMDs: one SIMD
// of every CU carries two waves (32 rows, where the column split gives every SIMD 20 rows' worth), in lock step behind the
// per-op barrier; (iii) ~800 VALU instructions per op and wave (fp16 split, epilogue) at 6-8 cycles each for a lone wave;
// (iv) the adjoint instance needs 372 VGPRs (96 slot registers + the epilogue's operands): one wave per SIMD, three compute
// waves per workgroup.  The layout is kept selectable (GEMNET_CHAIN_LAYOUT=row) for what it does better: no fp16 range limit
// (the 64-atom GemNet-Q fixture q64s needs no fall-back and is most accurate here) and no restriction on programs.
#include "common.h"

#include <type_traits>

#include "chain_split.h"

#ifdef GN_CHAIN_TRACE
// diagnosis build only (tools/chain4_trace.py): shader-clock stamps of compute wave 0 ([0]) and of the loader wave ([1]) of one workgroup
__device__ unsigned long long gn_chain4_trace_buf[2][GN_CHAIN_MAX_OPS + 1][8];
#define GN4_STAMP(w, o, i) do { if (lane == 0 && blockIdx.x == (gridDim.x > 100 ? 100u : 0u)) gn_chain4_trace_buf[w][o][i] = clock64(); } while (0)
#else
#define GN4_STAMP(w, o, i) do { } while (0)
#endif

namespace {
using namespace gn_split;

constexpr int WBUF = 64 * 1024;      // packed planes of one 128 x 128 weight
// compute waves per workgroup (+ 1 loader): 8 waves = 2 per SIMD at <= 256 VGPRs for the plain forward programs; the adjoint
// programs (parking slot, second outputs, source terms: 96 slot registers + the epilogue's operands) get one wave per SIMD
constexpr int row_max_waves(bool adj) { return adj ? 3 : 7; }

// max over the four lanes m, m + 16, m + 32, m + 48 (the lanes that hold one row)
__device__ __forceinline__ float row4_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4mul(float4 v, float a) { return make_float4(v.x * a, v.y * a, v.z * a, v.w * a); }
__device__ __forceinline__ float amax4(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
__device__ __forceinline__ float4 sel4(bool c, float4 a, float4 b) {
  return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
// phi by mode: 0 ssilu', 1 identity, 2 ssilu
__device__ __forceinline__ float4 phi4(float4 z, int mode) {
  if (mode == 0) return make_float4(gn_dssilu(z.x), gn_dssilu(z.y), gn_dssilu(z.z), gn_dssilu(z.w));
  if (mode == 1) return z;
  return make_float4(gn_ssilu(z.x), gn_ssilu(z.y), gn_ssilu(z.z), gn_ssilu(z.w));
}

template <bool ADJ>
__global__ __launch_bounds__(64 * (row_max_waves(ADJ) + 1)) void chain_row_kernel(const gn_chain_args P, const int nw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int n_ops = P.n_ops;

  if (wave == nw) {
    // ---- loader wave: the packed planes of GEMM g -> buffer g & 1, one barrier per GEMM op
    int g = 0;
    for (int oi = 0; oi < n_ops; ++oi) {
      if (P.ops[oi].kind != GN_OP_GEMM) continue;
      const unsigned char* __restrict__ const Wp = reinterpret_cast<const unsigned char*>(P.ops[oi].W);
      const int pieces = (P.ops[oi].N >> 4) * ((P.ops[oi].K + 31) >> 5) * 2;      // 1 KB each
      const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (uint32_t)(g & 1) * WBUF;
      const unsigned char* src = Wp + lane * 16;
      GN4_STAMP(1, g, 0);
      for (int i = 0; i < pieces; ++i) g2lds16(reinterpret_cast<const float*>(src + (size_t)i * 1024), base + (uint32_t)i * 1024u);
      GN4_STAMP(1, g, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      GN4_STAMP(1, g, 2);
      lds_barrier();
      GN4_STAMP(1, g, 3);
      ++g;
    }
    return;
  }

  // ---- compute waves
  const int m = lane & 15, g4 = lane >> 4;
  const int M = P.M;
  const int64_t grow = ((int64_t)blockIdx.x * nw + wave) * 16 + m;
  const bool ok = grow < M;
  const int64_t srow = ok ? grow : 0;     // a valid row for address arithmetic of masked lanes
  kernarg_warm();

  float4 S0[8], S1[8], PK[ADJ ? 8 : 1];
#pragma unroll
  for (int j = 0; j < 8; ++j) { S0[j] = f4zero(); S1[j] = f4zero(); }
#pragma unroll
  for (int j = 0; j < (ADJ ? 8 : 1); ++j) PK[j] = f4zero();

#define ROW_EACH8(...) _Pragma("unroll") for (int j = 0; j < 8; ++j) { __VA_ARGS__ }
  // (value selects on purpose: branches that copy from one of two arrays become selects of POINTERS, and an array whose
  // address is selected stays in scratch memory)
  auto get = [&](int slot, float4 (&v)[8]) {
    const bool s0 = slot == 0, s1 = slot == 1 || !ADJ;
    ROW_EACH8(v[j] = sel4(s0, S0[j], sel4(s1, S1[j], PK[ADJ ? j : 0]));)
  };
  // tiles j < nt of the slot <- v (the other tiles keep their content)
  auto put = [&](int slot, const float4 (&v)[8], int nt) {
    const bool s0 = slot == 0, s1 = slot == 1, s2 = ADJ && slot == 2;
    ROW_EACH8(const bool w = j < nt;
              S0[j] = sel4(w && s0, v[j], S0[j]); S1[j] = sel4(w && s1, v[j], S1[j]);
              if (ADJ) PK[ADJ ? j : 0] = sel4(w && s2, v[j], PK[ADJ ? j : 0]);)
  };

  int gord = 0;
  // One lane = one row: a lane whose row is beyond M works on row 0 (valid memory, results never stored) — every load is
  // unconditional and every group of stores sits under ONE `if (ok)`: no per-tile exec-mask branches.
  const float4 one4 = make_float4(1.f, 1.f, 1.f, 1.f);
  for (int oi = 0; oi < n_ops; ++oi) {
    const gn_chain_op op = P.ops[oi];
    // The whole descriptor in one round of scalar loads and one wait: left alone the compiler sinks every field's load to its
    // first use — ~25 serial scalar-cache round trips inside the epilogue (2 k cycles per op, tools/chain4_trace.py).
    asm volatile("" :: "s"(op.kind), "s"(op.slot), "s"(op.a_slot), "s"(op.width), "s"(op.ld), "s"(op.src), "s"(op.rows),
                 "s"(op.N), "s"(op.K), "s"(op.act), "s"(op.alpha), "s"(op.gadd1), "s"(op.gidx1), "s"(op.gadd2), "s"(op.gidx2),
                 "s"(op.pre_out), "s"(op.mul_slot), "s"(op.mul_g), "s"(op.res_slot), "s"(op.res_g), "s"(op.res_rows),
                 "s"(op.beta), "s"(op.res2_slot), "s"(op.res2_g), "s"(op.beta2), "s"(op.out), "s"(op.mul_mode));
    if (ADJ)
      asm volatile("" :: "s"(op.y2_slot), "s"(op.y2_src), "s"(op.mode2), "s"(op.alpha2), "s"(op.Z2), "s"(op.out2),
                   "s"(op.src_stage), "s"(op.src_mode), "s"(op.src_alpha), "s"(op.srcP), "s"(op.srcQ));
    const int kind = op.kind;
    if (wave == 0) GN4_STAMP(0, oi, 0);
    if (kind == GN_OP_LOAD) {
      const int width = op.width, slot = op.slot, ld = op.ld, y2_slot = op.y2_slot, mode2 = op.mode2;
      const int nt = width >> 4;
      const float alpha = op.alpha, alpha2 = op.alpha2;
      const float* __restrict__ const Z2 = op.Z2;
      const bool want2 = ADJ && y2_slot >= 0;
      const bool lsrc = want2 && op.src_stage == 2 && op.srcP;
      const int64_t sr = op.rows ? (int64_t)op.rows[srow] : srow;
      const float* __restrict__ const sp = op.src + sr * ld + (g4 << 2);
      float4 v[8];
      ROW_EACH8(v[j] = f4zero(); if (j < nt) v[j] = *reinterpret_cast<const float4*>(sp + 16 * j);)
      ROW_EACH8(v[j] = f4mul(v[j], alpha);)
      put(slot, v, 8);                      // columns beyond `width` are zero
      if (want2) {
        const size_t o2 = (size_t)srow * width + (g4 << 2);
        float4 u[8], z[8];
        ROW_EACH8(u[j] = f4mul(v[j], alpha2); z[j] = f4zero();)
        if (Z2) {
          ROW_EACH8(if (j < nt) z[j] = *reinterpret_cast<const float4*>(Z2 + o2 + 16 * j);)
          ROW_EACH8(if (j < nt) { const float4 f = phi4(z[j], mode2);
                                  u[j] = make_float4(u[j].x * f.x, u[j].y * f.y, u[j].z * f.z, u[j].w * f.w); })
        }
        if (lsrc) {
          const float* __restrict__ const srcP = op.srcP;
          const float* __restrict__ const srcQ = op.srcQ;
          const int src_mode = op.src_mode;
          const float src_alpha = op.src_alpha;
          ROW_EACH8(if (j < nt) {
            const float4 pp = *reinterpret_cast<const float4*>(srcP + o2 + 16 * j);
            const float4 qq = srcQ ? *reinterpret_cast<const float4*>(srcQ + o2 + 16 * j) : one4;
            const float4 sv = src_term(z[j], pp, qq, src_mode, src_alpha);
            u[j].x += sv.x; u[j].y += sv.y; u[j].z += sv.z; u[j].w += sv.w;
          })
        }
        put(y2_slot, u, 8);
      }
    } else if (kind == GN_OP_SCALE) {
      const int slot = op.slot, a_slot = op.a_slot, ld = op.ld, width = op.width;
      const int nt = width >> 4;
      const float alpha = op.alpha;
      const int mode = op.act;   // factor taken from src: 0 ssilu'(src), 1 src, 2 ssilu(src)
      const float* __restrict__ const src = op.src;
      float* __restrict__ const out = op.out;
      float4 v[8];
      get(a_slot, v);
      ROW_EACH8(v[j] = f4mul(v[j], alpha);)
      if (ADJ && slot == 2) {
        put(2, v, 8);             // park: plain scale of the whole slot
      } else {
        const size_t o = (size_t)srow * ld + (g4 << 2);
        float4 z[8];
        ROW_EACH8(z[j] = f4zero();)
        if (src) {
          ROW_EACH8(if (j < nt) z[j] = *reinterpret_cast<const float4*>(src + o + 16 * j);)
          ROW_EACH8(if (j < nt) { const float4 f = phi4(z[j], mode);
                                  v[j] = make_float4(v[j].x * f.x, v[j].y * f.y, v[j].z * f.z, v[j].w * f.w); })
        }
        if (ADJ && op.src_stage == 1 && op.srcP) {
          const float* __restrict__ const srcP = op.srcP;
          const float* __restrict__ const srcQ = op.srcQ;
          const int src_mode = op.src_mode;
          const float src_alpha = op.src_alpha;
          ROW_EACH8(if (j < nt) {
            const float4 pp = *reinterpret_cast<const float4*>(srcP + o + 16 * j);
            const float4 qq = srcQ ? *reinterpret_cast<const float4*>(srcQ + o + 16 * j) : one4;
            const float4 sv = src_term(z[j], pp, qq, src_mode, src_alpha);
            v[j].x += sv.x; v[j].y += sv.y; v[j].z += sv.z; v[j].w += sv.w;
          })
        }
        put(slot, v, nt);
        if (out && ok) { ROW_EACH8(if (j < nt) *reinterpret_cast<float4*>(out + o + 16 * j) = v[j];) }
      }
    } else if (kind == GN_OP_STORE) {
      const int nt = op.width >> 4, slot = op.slot, ld = op.ld;
      float* __restrict__ const out = op.out + (size_t)srow * ld + (g4 << 2);
      float4 v[8];
      get(slot, v);
      if (ok) { ROW_EACH8(if (j < nt) *reinterpret_cast<float4*>(out + 16 * j) = v[j];) }
    } else {  // GN_OP_GEMM
      const int N = op.N, K = op.K, a_slot = op.a_slot, y_slot = op.slot;
      const int nt = N >> 4, kt = K >> 4, kc = (K + 31) >> 5;
      // ---- B operand: the wave's 16 rows of the A-slot, split into two fp16 planes under a fresh row scale — one k-chunk
      // (column tiles 2 c and 2 c + 1 of the slot) at a time, the next chunk's planes are formed under this chunk's MFMAs
      const bool a0 = a_slot == 0;
      float sigma, inv_sigma;
      {
        float mx = 0.f;
        ROW_EACH8(if (j < kt) mx = fmaxf(mx, amax4(sel4(a0, S0[j], S1[j])));)
        sigma = row_sigma(row4_max(mx));
        inv_sigma = inv_pow2(sigma);
      }
      auto bsplit = [&](int c, uint4& bh, uint4& bl) {
        uint2 H0, L0, H1, L1;
        float4 x0 = f4zero(), x1 = f4zero();
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          if (cc == c) {
            if (2 * cc < kt) x0 = f4mul(sel4(a0, S0[2 * cc], S1[2 * cc]), sigma);
            if (2 * cc + 1 < kt) x1 = f4mul(sel4(a0, S0[2 * cc + 1], S1[2 * cc + 1]), sigma);
          }
        split4h(x0, H0, L0);
        split4h(x1, H1, L1);
        bh = make_uint4(H0.x, H0.y, H1.x, H1.y);
        bl = make_uint4(L0.x, L0.y, L1.x, L1.y);
      };
      uint4 bh[2], bl[2];
      bsplit(0, bh[0], bl[0]);
      if (wave == 0) GN4_STAMP(0, oi, 1);
      lds_barrier();            // the loader has staged this op's weights (and every wave has left the previous GEMM)
      if (wave == 0) GN4_STAMP(0, oi, 2);
      const unsigned char* const wb = smem + (size_t)(gord & 1) * WBUF + lane * 16;
      ++gord;
      // accumulators: hh | the correction products hl and lh (GN4_ACC = 3: in registers of their own; 2: in ONE set, issued
      // lh, hh, hl so that the two dependent MFMAs have an independent one between them)
#ifndef GN4_ACC
#define GN4_ACC 2
#endif
#ifndef GN4_PD
#define GN4_PD 2
#endif
      v4f acc[GN4_ACC][8];
#pragma unroll
      for (int k = 0; k < GN4_ACC; ++k) ROW_EACH8(acc[k][j] = (v4f){0.f, 0.f, 0.f, 0.f};)
#define GN4_MFMA3(j, ah, al, yh, yl)                                                                      \
      acc[GN4_ACC - 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, yh, acc[GN4_ACC - 1][j], 0, 0, 0); \
      acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yh, acc[0][j], 0, 0, 0);                      \
      acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yl, acc[1][j], 0, 0, 0);
      // fragment (j, c): hi plane at ((j kc + c) 2) KB, lo plane 1 KB behind; read PD steps (one column tile each) ahead.
      // The shapes the model uses are compile-time instances (static buffer rotation of the look-ahead); others: generic.
      auto mfma_phase = [&](auto nt_tag, auto kc_tag) {
        constexpr int NT = decltype(nt_tag)::value, KC = decltype(kc_tag)::value;
        constexpr int PD = GN4_PD, STEPS = NT * KC;
        uint4 wf[PD + 1][2];
        auto wload = [&](uint4 (&f)[2], int c, int j) {
          const unsigned char* p0 = wb + (size_t)(j * KC + c) * 2048;
          f[0] = *reinterpret_cast<const uint4*>(p0);
          f[1] = *reinterpret_cast<const uint4*>(p0 + 1024);
        };
#pragma unroll
        for (int s0 = 0; s0 < PD && s0 < STEPS; ++s0) wload(wf[s0 % (PD + 1)], s0 / NT, s0 % NT);
#pragma unroll
        for (int c = 0; c < KC; ++c) {
#if !defined(GN4_EXP) || !(GN4_EXP & 2)
          if (c + 1 < KC) bsplit(c + 1, bh[(c + 1) & 1], bl[(c + 1) & 1]);
#else
          bh[(c + 1) & 1] = bh[c & 1]; bl[(c + 1) & 1] = bl[c & 1];
#endif
          const f16x8 yh = __builtin_bit_cast(f16x8, bh[c & 1]), yl = __builtin_bit_cast(f16x8, bl[c & 1]);
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int s = c * NT + j, s2 = s + PD;
#if !defined(GN4_EXP) || !(GN4_EXP & 1)
            if (s2 < STEPS) wload(wf[s2 % (PD + 1)], s2 / NT, s2 % NT);
#else
            if (s2 < STEPS && s2 < PD + 1) wload(wf[s2 % (PD + 1)], s2 / NT, s2 % NT);
#endif
            uint4 (&f)[2] = wf[s % (PD + 1)];
            const f16x8 ah = __builtin_bit_cast(f16x8, f[0]), al = __builtin_bit_cast(f16x8, f[1]);
            GN4_MFMA3(j, ah, al, yh, yl)
          }
        }
      };
      auto mfma_generic = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c < kc) {
            if (c > 0) bsplit(c, bh[c & 1], bl[c & 1]);
            const f16x8 yh = __builtin_bit_cast(f16x8, bh[c & 1]), yl = __builtin_bit_cast(f16x8, bl[c & 1]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (j < nt) {
                const unsigned char* p0 = wb + (size_t)(j * kc + c) * 2048;
                const f16x8 ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(p0));
                const f16x8 al = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(p0 + 1024));
                GN4_MFMA3(j, ah, al, yh, yl)
              }
            }
          }
        }
      };
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      using I4 = std::integral_constant<int, 4>;
      using I8 = std::integral_constant<int, 8>;
      if (nt == 8 && kc == 4) mfma_phase(I8{}, I4{});            // 128 x 128
      else if (nt == 8 && kc == 2) mfma_phase(I8{}, I2{});       // 128 x 64
      else if (nt == 4 && kc == 4) mfma_phase(I4{}, I4{});       // 64 x 128
      else if (nt == 8 && kc == 1) mfma_phase(I8{}, I1{});       // 128 x 16 / 128 x 32
      else if (nt == 1 && kc == 4) mfma_phase(I1{}, I4{});       // 16 x 128
      else mfma_generic();
#undef GN4_MFMA3
#ifdef GN_CHAIN_TRACE
      { float sink = 0.f; ROW_EACH8(sink += acc[0][j][0] + acc[1][j][0] + acc[GN4_ACC - 1][j][0];) if (sink == 1.2345e30f) smem[0] = 1; }
      if (wave == 0) GN4_STAMP(0, oi, 3);
#endif

      // ---- epilogue: lane holds columns 16 j + 4 g4 .. + 3 of row m for j < nt
      const int act = op.act & 1;
      const bool pre_deriv = (op.act & 2) != 0;
      const float alpha = op.alpha, beta = op.beta, beta2 = op.beta2;
      const float* __restrict__ const gadd1 = op.gadd1;
      const float* __restrict__ const gadd2 = op.gadd2;
      float* __restrict__ const pre_out = op.pre_out;
      float* __restrict__ const out = op.out;
      const int mul_slot = op.mul_slot, res_slot = op.res_slot, res2_slot = op.res2_slot;
      const float* __restrict__ const mul_g = op.mul_g;
      const float* __restrict__ const res_g = op.res_g;
      const float* __restrict__ const res2_g = op.res2_g;
      const int mul_mode = op.mul_mode, y2_slot = op.y2_slot, y2_src = op.y2_src, mode2 = op.mode2;
      const float alpha2 = op.alpha2;
      const float* __restrict__ const Z2 = op.Z2;
      float* __restrict__ const out2 = op.out2;
      const int src_stage = ADJ ? op.src_stage : 0, src_mode = op.src_mode;
      const float src_alpha = op.src_alpha;
      const float* __restrict__ const srcP = op.srcP;
      const float* __restrict__ const srcQ = op.srcQ;
      const uint32_t off = (uint32_t)srow * (uint32_t)N + (uint32_t)(g4 << 2);   // (row, 4 g4) of an (M, N) matrix; tile j: + 16 j
      float4 v[8];
      ROW_EACH8(const v4f s = (acc[0][j] + (GN4_ACC == 3 ? acc[1][j] + acc[GN4_ACC - 1][j] : acc[1][j]) * H_DOWN) * inv_sigma;
                v[j] = make_float4(s[0], s[1], s[2], s[3]);)
      auto epilogue = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;       // nt == 8
#define ROW_EACH(...) _Pragma("unroll") for (int j = 0; j < 8; ++j) if (FULL || j < nt) { __VA_ARGS__ }
#define ROW_ADD(q) v[j].x += q.x; v[j].y += q.y; v[j].z += q.z; v[j].w += q.w;
#define ROW_MUL(q) v[j].x *= q.x; v[j].y *= q.y; v[j].z *= q.z; v[j].w *= q.w;
#define ROW_RES(q, b) v[j].x = (v[j].x + q.x) * b; v[j].y = (v[j].y + q.y) * b; v[j].z = (v[j].z + q.z) * b; v[j].w = (v[j].w + q.w) * b;
        if (gadd1) {
          const float* __restrict__ const gp = gadd1 + (size_t)op.gidx1[srow] * N + (g4 << 2);
          float4 q[8];
          ROW_EACH(q[j] = *reinterpret_cast<const float4*>(gp + 16 * j);)
          ROW_EACH(ROW_ADD(q[j]))
        }
        if (gadd2) {
          const float* __restrict__ const gp = gadd2 + (size_t)op.gidx2[srow] * N + (g4 << 2);
          float4 q[8];
          ROW_EACH(q[j] = *reinterpret_cast<const float4*>(gp + 16 * j);)
          ROW_EACH(ROW_ADD(q[j]))
        }
        if (pre_out && !pre_deriv && ok) { ROW_EACH(*reinterpret_cast<float4*>(pre_out + off + 16 * j) = v[j];) }
        if (act && pre_out && pre_deriv) {
          float4 d[8];
          ROW_EACH(gn_ssilu_pair(v[j].x, v[j].x, d[j].x); gn_ssilu_pair(v[j].y, v[j].y, d[j].y);
                   gn_ssilu_pair(v[j].z, v[j].z, d[j].z); gn_ssilu_pair(v[j].w, v[j].w, d[j].w);)
          if (ok) { ROW_EACH(*reinterpret_cast<float4*>(pre_out + off + 16 * j) = d[j];) }
        } else if (act) { ROW_EACH(v[j] = phi4(v[j], 2);) }
        const bool want2 = ADJ && (y2_slot >= 0 || out2);
        // second output = v * alpha2 * phi2(Z2) (+ source term), from the current value of v
        auto emit_y2 = [&]() {
          float4 q[8], z[8];
          ROW_EACH8(q[j] = f4zero(); z[j] = f4zero();)
          if (Z2) {
            ROW_EACH(z[j] = *reinterpret_cast<const float4*>(Z2 + off + 16 * j);)
            ROW_EACH(const float4 f = phi4(z[j], mode2);
                     q[j] = make_float4(f.x * v[j].x * alpha2, f.y * v[j].y * alpha2, f.z * v[j].z * alpha2, f.w * v[j].w * alpha2);)
          } else {
            ROW_EACH(q[j] = f4mul(v[j], alpha2);)
          }
          if (src_stage == 2 && srcP) {
            ROW_EACH(
              const float4 pp = *reinterpret_cast<const float4*>(srcP + off + 16 * j);
              const float4 qq = srcQ ? *reinterpret_cast<const float4*>(srcQ + off + 16 * j) : one4;
              const float4 sv = src_term(z[j], pp, qq, src_mode, src_alpha);
              q[j].x += sv.x; q[j].y += sv.y; q[j].z += sv.z; q[j].w += sv.w;)
          }
          if (out2 && ok) { ROW_EACH(*reinterpret_cast<float4*>(out2 + off + 16 * j) = q[j];) }
          if (y2_slot >= 0) put(y2_slot, q, nt);
        };
        if (want2 && y2_src) emit_y2();
        if (mul_slot >= 0) {
          float4 q[8];
          get(mul_slot, q);
          ROW_EACH(ROW_MUL(q[j]))
        } else if (mul_g) {
          float4 q[8];
          ROW_EACH(q[j] = *reinterpret_cast<const float4*>(mul_g + off + 16 * j);)
          if (mul_mode == 2) { ROW_EACH(q[j] = phi4(q[j], 0);) }
          else if (mul_mode == 3) { ROW_EACH(q[j] = phi4(q[j], 2);) }
          ROW_EACH(ROW_MUL(q[j]))
        }
        if (alpha != 1.0f) { ROW_EACH(v[j] = f4mul(v[j], alpha);) }
        if (src_stage == 1 && srcP) {     // y += src_alpha * phis(mul_g) * P * Q
          ROW_EACH(
            const float4 zs = (mul_g && src_mode == 1) ? *reinterpret_cast<const float4*>(mul_g + off + 16 * j) : f4zero();
            const float4 pp = *reinterpret_cast<const float4*>(srcP + off + 16 * j);
            const float4 qq = srcQ ? *reinterpret_cast<const float4*>(srcQ + off + 16 * j) : one4;
            const float4 sv = src_term(zs, pp, qq, src_mode, src_alpha);
            v[j].x += sv.x; v[j].y += sv.y; v[j].z += sv.z; v[j].w += sv.w;)
        }
        if (res_slot >= 0) {
          float4 q[8];
          get(res_slot, q);
          ROW_EACH(ROW_RES(q[j], beta))
        } else if (res_g) {
          const float* __restrict__ const rp = op.res_rows ? res_g + (size_t)op.res_rows[srow] * N + (g4 << 2) : res_g + off;
          float4 q[8];
          ROW_EACH(q[j] = *reinterpret_cast<const float4*>(rp + 16 * j);)
          ROW_EACH(ROW_RES(q[j], beta))
        }
        if (res2_slot >= 0) {
          float4 q[8];
          get(res2_slot, q);
          ROW_EACH(ROW_RES(q[j], beta2))
        } else if (res2_g) {
          float4 q[8];
          ROW_EACH(q[j] = *reinterpret_cast<const float4*>(res2_g + off + 16 * j);)
          ROW_EACH(ROW_RES(q[j], beta2))
        }
        if (out && ok) { ROW_EACH(*reinterpret_cast<float4*>(out + off + 16 * j) = v[j];) }
        if (y_slot >= 0) {
          put(y_slot, v, nt);
          // N = 16 (mod 32): the next GEMM reads k-chunks of 32 columns, the 16 columns after N are zeroed
          if (!FULL && (N & 16) && y_slot < 2) {
            ROW_EACH8(const bool w = j == nt; S0[j] = sel4(w && y_slot == 0, f4zero(), S0[j]); S1[j] = sel4(w && y_slot == 1, f4zero(), S1[j]);)
          }
        }
        if (want2 && !y2_src) emit_y2();
#undef ROW_EACH
#undef ROW_ADD
#undef ROW_MUL
#undef ROW_RES
      };
      if (nt == 8) epilogue(std::true_type{}); else epilogue(std::false_type{});
      if (wave == 0) GN4_STAMP(0, oi, 4);
    }
  }
  if (wave == 0) GN4_STAMP(0, GN_CHAIN_MAX_OPS, 0);
#undef ROW_EACH8
}

template <bool ADJ>
int launch_chain_row(const gn_chain_args* args, hipStream_t st) {
  const int nblocks = gn_cdiv(args->M, 16);
  int nw = gn_cdiv(nblocks, 256);
  nw = nw < 1 ? 1 : (nw > row_max_waves(ADJ) ? row_max_waves(ADJ) : nw);
  constexpr size_t smem = (size_t)2 * WBUF;
  static std::atomic<bool> configured{false};
  if (!configured.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_row_kernel<ADJ>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    configured.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((chain_row_kernel<ADJ>), dim3(gn_cdiv(nblocks, nw)), dim3(64 * (nw + 1)), smem, st, *args, nw);
  GN_LAUNCH_CHECK();
  return 0;
}

}  // namespace

#ifdef GN_CHAIN_TRACE
extern "C" int gn_chain4_trace_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gn_chain4_trace_buf), sizeof(gn_chain4_trace_buf));
}
#endif

// (argument checks common to all layouts are done by the caller, gn_chain_split_f32)
int gn_chain_row_dispatch(const gn_chain_args* args, bool adj, hipStream_t st) {
  for (int i = 0; i < args->n_ops; ++i) {
    const gn_chain_op& o = args->ops[i];
    if (o.kind == GN_OP_GEMM) {
      if ((o.K % 16) != 0) return (int)hipErrorInvalidValue;
    } else if ((o.width % 16) != 0) return (int)hipErrorInvalidValue;
  }
  return adj ? launch_chain_row<true>(args, st) : launch_chain_row<false>(args, st);
}
