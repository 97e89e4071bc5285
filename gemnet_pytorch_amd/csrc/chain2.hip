// LDS-resident layer chains on the bf16 matrix pipe with split operands (include/gemnet_hip.h, gn_chain_split_f32).
//
// Why: gfx950 runs f32-input MFMA at the f32 VECTOR rate (256 FLOP/clk/CU: 10.2 k cycles for one 80-row x 128 x 128
// layer per CU) while `v_mfma_f32_16x16x32_bf16` runs 16x faster.  An fp32 value is EXACTLY the sum of three bf16
// planes (8 + 8 + 8 significand bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)); keeping the six
// largest of the nine cross products of two split operands,
//       a b ~= ah bh + (ah bm + am bh) + (ah bl + al bh + am bm),             fp32 accumulation,
// drops only terms below 2^-24 |a b|: the same order as fp32 rounding itself (measured on the full 4-block model,
// tools/exp/split_precision_cpu.py: force MAE 2.8e-6 vs 3.6e-6 for plain fp32, 7.9e-5 with 3 products, 1.9e-2 with 1).
// Six bf16 MFMAs cost 6/16 of one f32 MFMA pass.  `nprod` selects 6 (fp32-equivalent, default), 3 or 1 (plain bf16
// operands, fp32 accumulate and fp32 everywhere else: the BASELINE "bf16" configuration).
//
// Layout of one row tile (BM = 16 RT rows, RT = 1..5 so that a launch is one round of <= 256 workgroups):
//   * two LDS slots, each THREE bf16 planes [BM][128]; the 16-byte unit u of row r lives at unit u ^ (r & 15): the
//     ds_read_b128 fragment reads are conflict-free in the b128 lane groups and the ds_write_b64 of the transposed
//     epilogue 2-way (4-way with padded rows; no 16-byte-aligned layout does better, tools/exp/lds_swizzle.py);
//     a slot holds exact fp32 values (residuals / Hadamard operands are rebuilt as hi + mid + lo);
//     "slot 2" is a register-resident parking slot in accumulator layout (skip-connection gradient of the adjoint);
//   * 8 waves; wave w owns output columns 16w..16w+15 for all RT row blocks and computes the TRANSPOSED tile
//     (A operand = its 16 weight rows, B operand = the activations): lane (m = lane % 16, g = lane / 16) ends up with
//     4 CONSECUTIVE columns 16w + 4g .. +3 of row 16t + m, so every epilogue access is a float4 / ds_write_b64;
//   * weights arrive pre-split and pre-packed in fragment order (gn_pack_weight_split): one 1 KB contiguous load per
//     (k-chunk, plane) and wave — the strided 16 x 64 B fragment loads of chain.hip pulled 64 KB in 4.0 k cycles with
//     every CU asking at once, the packed form in 1.1 k (profiles/r2_wfetch.txt); next op's fragments are prefetched;
//   * two accumulator sets per row block (hh | the five cross terms): the small terms are summed among themselves
//     first; a third set (tried) pushed the RT = 5 kernel over 256 VGPRs.
#include "common.h"

#include "chain_split.h"

// Diagnosis switches of the trace build (never defined in the product library; GN_TRACE_DEFS of tools/chain2_trace.py):
//   GN_EXP == 1  the MFMA phase without its LDS fragment reads (pipe time alone)
//   GN_EXP == 2  the fragment reads without the MFMAs (LDS time alone)          -> DESIGN.md section 5
#ifdef GN_CHAIN_TRACE
// diagnosis build only (tools/chain2_trace.py): shader-clock stamps of wave 0 of two workgroups
__device__ unsigned long long gn_chain2_trace_buf[2][GN_CHAIN_MAX_OPS][8];
#define GN2_STAMP(i) do { if (lane == 0 && (wave == 0 || wave == GN_TRACE_WAVE) && blockIdx.x == (gridDim.x > 100 ? 100u : 0u)) \
    gn_chain2_trace_buf[wave != 0][oi][i] = clock64(); } while (0)
#ifndef GN_TRACE_WAVE
#define GN_TRACE_WAVE 7
#endif
#else
#define GN2_STAMP(i) do { } while (0)
#endif

namespace {
using namespace gn_split;

// NPL = planes that enter the MFMAs: 1 -> 1 product (bf16 operands), 2 -> 3 products, 3 -> 6 products
// ADJ: the program uses the register parking slot or second outputs (the adjoint programs); plain forward stacks run the
// leaner variant (park / y2 registers would push the RT = 5 kernel into scratch, and a kernel with a scratch segment pays
// ~1 us more per launch).
// HF: the two-plane fp16 format (NPL = 2 there: both planes enter the MFMAs, three products)
// `meta` (computed by the launcher's pass over the program): bits 0-7 index of the first GEMM op (0xff: none), bit 8 the
// program is linear (no GEMM applies an activation) — what the prologue needs before the op table exists.
// GN2_TWO_WG (experiment build only, tools/exp: DESIGN.md section 9): registers capped at 128 (4 waves per SIMD) and row tiles sized
// for TWO co-resident workgroups per CU
#ifdef GN2_TWO_WG
#define GN2_BOUNDS __launch_bounds__(NT, 4)
#else
#define GN2_BOUNDS __launch_bounds__(NT)
#endif
template <int RT, int NPL, bool ADJ, bool HF>
__global__ GN2_BOUNDS void chain_split_kernel(const gn_chain_args P, const int meta) {
  static_assert(!HF || NPL == 2, "format H has two planes");
  constexpr int BM = 16 * RT;
  constexpr int PLANE = BM * ROWB;          // bytes of one plane
  constexpr int SP = HF ? 2 : 3;            // planes of a slot (bf16: always three, exact fp32 content) and of a packed weight
  constexpr int SLOT = SP * PLANE;          // bytes of one slot
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int l15 = lane & 15;
  const int lg = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  const int M = P.M;
  // weight pointer / shape of the GEMM ops in program order, staged in LDS: the next op's fragments are requested at the
  // top of every GEMM op and must not wait for a scalar load of its descriptor from the kernarg segment
  __shared__ const void* gemm_W[GN_CHAIN_MAX_OPS + 2];
  __shared__ int gemm_NK[GN_CHAIN_MAX_OPS + 2];
  // format H: row scales of the two LDS slots.  `rs` is the copy the row-linear ops (LOAD / SCALE / STORE) use, `sg` the
  // register copy in accumulator layout (rows 16 t + l15) that the GEMM epilogues use and update without LDS traffic.
  // (Every op that writes a slot writes the scales of all its rows: no initial state.)
  __shared__ float rs[2][16 * 5];
#ifdef GN_CHAIN_TRACE
  if (lane == 0 && wave == 0 && blockIdx.x == (gridDim.x > 100 ? 100u : 0u)) gn_chain2_trace_buf[0][GN_CHAIN_MAX_OPS - 1][0] = clock64();
#endif
  // Prologue.  The one-thread loop over P.ops[j] that used to build the op table was a chain of 2 n_ops dependent scalar
  // loads — cache misses — in front of the first instruction of every workgroup (6.8 k cycles at 227 workgroups,
  // tools/chain2_trace.py), and every later op began with the scalar-cache miss of its own descriptor.  Now: one lane per
  // op requests the GEMM fields of its descriptor with vector loads; while they are in flight every wave touches each
  // line of the argument block through the scalar cache (kernarg_warm); the table is written without a barrier of its
  // own (its first reader is the weight prefetch issued inside the first GEMM op, behind the barrier of the LOAD that
  // every program starts with); the first GEMM's fragments are requested straight from its descriptor.
  // (every wave issues the loads — a branch around them makes the compiler wait for the values at its join — wave 0 uses them)
  const gn_chain_op* __restrict__ const t_o =
      &((const gn_chain_args*)__builtin_amdgcn_kernarg_segment_ptr())->ops[lane < GN_CHAIN_MAX_OPS ? lane : 0];
  int t_kind = t_o->kind, t_N = t_o->N, t_K = t_o->K;
  unsigned long long t_W = (unsigned long long)(uintptr_t)t_o->W;
  kernarg_warm();
  // (the loaded values are consumed behind this statement: without it the compiler waits for them in front of the warm-up)
  asm volatile("" : "+v"(t_kind), "+v"(t_W), "+v"(t_N), "+v"(t_K));
  if (wave == 0) {
    const bool is_gemm = lane < P.n_ops && t_kind == GN_OP_GEMM;
    const unsigned long long gm = __ballot(is_gemm);
    const int g = __popcll(gm & ((1ull << lane) - 1ull)), ng = __popcll(gm);
    if (is_gemm) { gemm_W[g] = (const void*)(uintptr_t)t_W; gemm_NK[g] = (t_N << 16) | t_K; }
    if (lane < 2) { gemm_W[ng + lane] = nullptr; gemm_NK[ng + lane] = 0; }
  }
#ifdef GN_CHAIN_TRACE
  if (lane == 0 && wave == 0 && blockIdx.x == (gridDim.x > 100 ? 100u : 0u)) gn_chain2_trace_buf[0][GN_CHAIN_MAX_OPS - 1][1] = clock64();
#endif
#ifdef GN_H3_NOSCALE   // diagnosis build only (tools/chain2_trace.py): the cost of the row scales
  const bool scaled = false;
#else
  const bool scaled = HF && (meta & 0x100) != 0;   // uniform
#endif
  float sg[2][HF ? RT : 1];
#pragma unroll
  for (int t = 0; t < (HF ? RT : 1); ++t) sg[0][t] = sg[1][t] = 1.f;
  auto sg_get = [&](int slot, int t) -> float { return HF ? (slot ? sg[1][HF ? t : 0] : sg[0][HF ? t : 0]) : 1.f; };
  auto sg_set = [&](int slot, int t, float v) { if (HF) { if (slot) sg[1][HF ? t : 0] = v; else sg[0][HF ? t : 0] = v; } };

  // weight fragments of one GEMM op: [k-chunk c][plane p] -> 8 bf16 (A operand rows = this wave's 16 weight rows)
  uint4 bcur[4][NPL];
  auto wload = [&](uint4 (&dst)[4][NPL], const void* Wp, const int nk) {
    // packed layout: [col tile][k-chunk][plane (3)][lane][8 bf16]
    const int N = nk >> 16, kc = ((nk & 0xffff) + 31) >> 5;
    const uint4* __restrict__ base = reinterpret_cast<const uint4*>(Wp) + ((size_t)wave * kc * SP) * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        dst[c][p] = make_uint4(0u, 0u, 0u, 0u);
        if (wave * 16 < N && c < kc) dst[c][p] = base[(c * SP + p) * 64];
      }
  };
  auto wload_op = [&](uint4 (&dst)[4][NPL], int ord) { wload(dst, gemm_W[ord], __builtin_amdgcn_readfirstlane(gemm_NK[ord])); };
  // One fragment set: the NEXT GEMM's weights are requested right after the current op's MFMA phase (its fragments
  // are dead then) and land under the epilogue + barrier (96 KB per CU arrive in ~2.2 k cycles with every CU asking,
  // profiles/r2_wfetch.txt); a second set held across the MFMA phase pushed the RT = 5 kernel into scratch.
  {
    const int fg = meta & 0xff;      // first GEMM op: fragments requested from its descriptor (scalar loads, warm cache)
    if (fg != 0xff) wload(bcur, P.ops[fg].W, (P.ops[fg].N << 16) | P.ops[fg].K);
    else wload(bcur, nullptr, 0);
  }
  int gord = 0;
  // Row tiles of one or two blocks (M <= 8 k rows: the atom-side stacks, 64 workgroups) are latency-bound (4.8 k cycles
  // per op for 0.8 k of MFMA pipe time, tools/chain2_trace.py --small).  Tried for them and dropped, none moved the op
  // time: a second weight-fragment set requested one op ahead, one accumulator set per product (six independent MFMA
  // chains), X fragments read three steps ahead (all reads of the op in flight before the first MFMA).

  float4 park[ADJ ? RT : 1];   // register slot 2 (accumulator layout)
#pragma unroll
  for (int t = 0; t < (ADJ ? RT : 1); ++t) park[t] = make_float4(0.f, 0.f, 0.f, 0.f);

  auto plane_ptr = [&](int slot, int p) -> unsigned char* { return smem + slot * SLOT + p * PLANE; };
  // format H leaves LDS room for a per-wave staging area (RT KB per wave): the global factor of a GEMM epilogue (`mul_g`)
  // is requested before the MFMA phase with global_load_lds (no registers) and read back lane-privately after it
  unsigned char* const stage = smem + 2 * SLOT + wave * (RT * 1024);
  const uint32_t stage_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)stage;   // LDS byte address
  // accumulator-layout element block of this lane in row block t: row 16t + l15, columns 16 wave + 4 lg .. +3
  // byte offset of columns col .. col+3 (col % 4 == 0) of row `row` inside a plane
  auto sw_off = [&](int row, int col) -> int { return row * ROWB + ((((col >> 3) ^ row) & 15) << 4) + ((col & 4) << 1); };
  auto slot_write = [&](int slot, int row, int col, const float4 v) {
    const int off = sw_off(row, col);
    if constexpr (HF) {
      uint2 H, L;
      split4h(v, H, L);
      *reinterpret_cast<uint2*>(plane_ptr(slot, 0) + off) = H;
      *reinterpret_cast<uint2*>(plane_ptr(slot, 1) + off) = L;
    } else {
      uint2 H, Mi, L;
      split4(v, H, Mi, L);
      *reinterpret_cast<uint2*>(plane_ptr(slot, 0) + off) = H;
      *reinterpret_cast<uint2*>(plane_ptr(slot, 1) + off) = Mi;
      *reinterpret_cast<uint2*>(plane_ptr(slot, 2) + off) = L;
    }
  };
  auto slot_read = [&](int slot, int row, int col) -> float4 {
    const int off = sw_off(row, col);
    if constexpr (HF)
      return join4h(*reinterpret_cast<const uint2*>(plane_ptr(slot, 0) + off),
                    *reinterpret_cast<const uint2*>(plane_ptr(slot, 1) + off));
    else
      return join4(*reinterpret_cast<const uint2*>(plane_ptr(slot, 0) + off),
                   *reinterpret_cast<const uint2*>(plane_ptr(slot, 1) + off),
                   *reinterpret_cast<const uint2*>(plane_ptr(slot, 2) + off));
  };
  auto mul4 = [](float4 v, float a) -> float4 { return make_float4(v.x * a, v.y * a, v.z * a, v.w * a); };
  // accumulator-layout read of an LDS slot in TRUE scale
  auto slot_read_acc = [&](int slot, int t) -> float4 {
    const float4 v = slot_read(slot, 16 * t + l15, wave * 16 + (lg << 2));
    return HF ? mul4(v, inv_pow2(sg_get(slot, t))) : v;
  };

  for (int oi = 0; oi < P.n_ops; ++oi) {
    // by value: the whole descriptor in one round of scalar loads, one wait.  (Staging the descriptors in LDS and reading
    // them back through v_readfirstlane was measured: the gap between ops grew from 0.6 - 0.9 k to 1.2 - 1.5 k cycles.)
    const gn_chain_op op = P.ops[oi];
    const int kind = op.kind;
    GN2_STAMP(0);
    if (kind == GN_OP_LOAD) {
      // linear mapping (coalesced 512 B rows); columns up to the next multiple of 32 are zero-filled so that the
      // zero-padded k-chunk of a K = 16 weight never multiplies stale LDS bits
      const int width = op.width, slot = op.slot, ld = op.ld, y2_slot = op.y2_slot, mode2 = op.mode2;
      const int w4 = ((width + 31) & ~31) >> 2;
      const float alpha = op.alpha, alpha2 = op.alpha2;
      const float* __restrict__ const src = op.src;
      const float* __restrict__ const Z2 = op.Z2;
      const int32_t* __restrict__ const rows = op.rows;
      const bool lsrc = ADJ && op.src_stage == 2 && y2_slot >= 0 && op.srcP;
      const float* __restrict__ const srcP = op.srcP;
      const float* __restrict__ const srcQ = op.srcQ;
      const int src_mode = op.src_mode;
      const float src_alpha = op.src_alpha;
      // all global loads of the tile are issued before the first one is consumed (a tile is at most RT passes of the
      // 512 threads): the serial load -> split -> write form took 6.7 k cycles for 40 KB (tools/chain2_trace.py)
      float4 v[RT], zz[RT];
      bool in[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const int f = tid + i * NT;
        const int r = f / w4, c = (f - r * w4) << 2;
        const int64_t gr = row0 + r;
        in[i] = f < BM * w4 && gr < M && c < width;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        zz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in[i]) {
          const int64_t sr = rows ? (int64_t)rows[gr] : gr;
          v[i] = *reinterpret_cast<const float4*>(src + sr * ld + c);
          if (ADJ && y2_slot >= 0 && Z2) zz[i] = *reinterpret_cast<const float4*>(Z2 + gr * width + c);
        }
      }
      float sig[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        v[i].x *= alpha; v[i].y *= alpha; v[i].z *= alpha; v[i].w *= alpha;
        sig[i] = 1.f;
      }
      if (scaled && (w4 == 8 || w4 == 16 || w4 == 32)) {
        // row maximum over the w4 consecutive lanes that hold the row (w4 divides the wave and NT); other widths: unscaled
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const float m = fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
          sig[i] = row_sigma(group_max(m, w4));
        }
      }
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const int f = tid + i * NT;
        if (f < BM * w4) {
          const int r = f / w4, c = (f - r * w4) << 2;
          slot_write(slot, r, c, HF ? mul4(v[i], sig[i]) : v[i]);
          if (HF && c == 0) { rs[slot][r] = sig[i]; if (ADJ && y2_slot >= 0) rs[y2_slot][r] = sig[i]; }
          if (ADJ && y2_slot >= 0) {   // second tensor derived from the loaded rows: v * alpha2 * phi2(Z2)
            float4 u = make_float4(v[i].x * alpha2, v[i].y * alpha2, v[i].z * alpha2, v[i].w * alpha2);
            if (Z2 && in[i]) {
              const float4 z = zz[i];
              if (mode2 == 0) { u.x *= gn_dssilu(z.x); u.y *= gn_dssilu(z.y); u.z *= gn_dssilu(z.z); u.w *= gn_dssilu(z.w); }
              else if (mode2 == 1) { u.x *= z.x; u.y *= z.y; u.z *= z.z; u.w *= z.w; }
              else { u.x *= gn_ssilu(z.x); u.y *= gn_ssilu(z.y); u.z *= gn_ssilu(z.z); u.w *= gn_ssilu(z.w); }
            }
            if (lsrc && in[i]) {
              const int64_t gr = row0 + r;
              const float4 p = *reinterpret_cast<const float4*>(srcP + gr * width + c);
              const float4 q = srcQ ? *reinterpret_cast<const float4*>(srcQ + gr * width + c) : make_float4(1.f, 1.f, 1.f, 1.f);
              const float4 sv = src_term(zz[i], p, q, src_mode, src_alpha);
              u.x += sv.x; u.y += sv.y; u.z += sv.z; u.w += sv.w;
            }
            slot_write(y2_slot, r, c, HF ? mul4(u, sig[i]) : u);
          }
        }
      }
      GN2_STAMP(3);
      lds_barrier();
      if (HF) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          sg_set(slot, t, rs[slot][16 * t + l15]);
          if (ADJ && y2_slot >= 0) sg_set(y2_slot, t, rs[y2_slot][16 * t + l15]);
        }
      }
      GN2_STAMP(4);
    } else if (kind == GN_OP_SCALE) {
      const int slot = op.slot, a_slot = op.a_slot, ld = op.ld, width = op.width;
      const float alpha = op.alpha;
      const int mode = op.act;   // factor taken from src: 0 ssilu'(src), 1 src, 2 ssilu(src)
      const float* __restrict__ const src = op.src;
      float* __restrict__ const out = op.out;
      if (ADJ && slot == 2) {
        // park: register slot in accumulator layout (plain scale only)
        if (wave * 16 < width) {
#pragma unroll
          for (int t = 0; t < RT; ++t) {
            float4 v = slot_read_acc(a_slot, t);
            park[ADJ ? t : 0] = make_float4(v.x * alpha, v.y * alpha, v.z * alpha, v.w * alpha);
          }
        }
        lds_barrier();   // the parked values are read before any later op (other thread mapping) rewrites a_slot
      } else {
        const int w4 = width >> 2;
        for (int f = tid; f < BM * w4; f += NT) {
          const int r = f / w4, c = (f - r * w4) << 2;
          const int64_t gr = row0 + r;
          const float sc = HF ? rs[a_slot][r] : 1.f;       // the output inherits the row scale of its operand
          float4 v = slot_read(a_slot, r, c);
          const float a_true = HF ? alpha * inv_pow2(sc) : alpha;
          v.x *= a_true; v.y *= a_true; v.z *= a_true; v.w *= a_true;
          float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
          if (src && gr < M) {
            z = *reinterpret_cast<const float4*>(src + gr * ld + c);
            if (mode == 0) { v.x *= gn_dssilu(z.x); v.y *= gn_dssilu(z.y); v.z *= gn_dssilu(z.z); v.w *= gn_dssilu(z.w); }
            else if (mode == 1) { v.x *= z.x; v.y *= z.y; v.z *= z.z; v.w *= z.w; }
            else { v.x *= gn_ssilu(z.x); v.y *= gn_ssilu(z.y); v.z *= gn_ssilu(z.z); v.w *= gn_ssilu(z.w); }
          }
          if (ADJ && op.src_stage == 1 && op.srcP && gr < M) {
            const float4 p = *reinterpret_cast<const float4*>(op.srcP + gr * ld + c);
            const float4 q = op.srcQ ? *reinterpret_cast<const float4*>(op.srcQ + gr * ld + c) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 sv = src_term(z, p, q, op.src_mode, op.src_alpha);
            v.x += sv.x; v.y += sv.y; v.z += sv.z; v.w += sv.w;
          }
          slot_write(slot, r, c, HF ? mul4(v, sc) : v);
          if (HF && c == 0 && slot != a_slot) rs[slot][r] = sc;
          if (out && gr < M) *reinterpret_cast<float4*>(out + gr * ld + c) = v;
        }
        lds_barrier();
        if (HF) {
#pragma unroll
          for (int t = 0; t < RT; ++t) sg_set(slot, t, sg_get(a_slot, t));
        }
      }
    } else if (kind == GN_OP_STORE) {
      const int w4 = op.width >> 2, slot = op.slot, ld = op.ld;
      float* __restrict__ const out = op.out;
      for (int f = tid; f < BM * w4; f += NT) {
        const int r = f / w4, c = (f - r * w4) << 2;
        const int64_t gr = row0 + r;
        if (gr < M) {
          const float4 v = slot_read(slot, r, c);
          *reinterpret_cast<float4*>(out + gr * ld + c) = HF ? mul4(v, inv_pow2(rs[slot][r])) : v;
        }
      }
      lds_barrier();
    } else {  // GN_OP_GEMM
      const int N = op.N, K = op.K, a_slot = op.a_slot, y_slot = op.slot, act = op.act & 1;
      const bool pre_deriv = (op.act & 2) != 0;
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
      const int src_stage = ADJ ? op.src_stage : 0, src_mode = op.src_mode;
      const float src_alpha = op.src_alpha;
      const float* __restrict__ const srcP = op.srcP;
      const float* __restrict__ const srcQ = op.srcQ;
      const bool active = wave * 16 < N;
      ++gord;
#ifdef GN_CHAIN_TRACE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this op's fragments landed
#endif
      GN2_STAMP(1);

#ifdef GN_NO_STAGE     // diagnosis build only
      const bool staged = false;
#else
      const bool staged = HF && active && mul_g != nullptr && mul_slot < 0;
#endif
      if (HF && staged) {
        const int n0 = wave * 16 + (lg << 2);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const int64_t grow = row0 + 16 * t + l15;
          if (grow < M) g2lds16(mul_g + (uint32_t)grow * (uint32_t)N + (uint32_t)n0, stage_lds + t * 1024);
        }
      }
      // hh | the cross terms hm + mh + hl + lh + mm (summed among themselves first).  Format H: hl and lh in registers of their
      // own — issued back to back into ONE accumulator the second waits for the first (the matrix pipe alone took 2.65 k
      // cycles per op for 1.92 k of issue time, profiles/r4_chain_mfma_phase.txt)
      constexpr int NACC = HF ? 3 : 2;
      v4f acc[NACC][RT];
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k][t] = (v4f){0.f, 0.f, 0.f, 0.f};
#define GN2_ACC(k) acc[(k) != 0][t]     /* product k: 0 hh, 1 hm, 2 mh, 3 hl, 4 lh, 5 mm */
      if (active) {
        const unsigned char* xb = smem + a_slot * SLOT + l15 * ROWB;   // row 16 t + l15: swizzle key = l15
        const int kc = (K + 31) >> 5;
        // X fragments are double-buffered by hand: the reads of step (c, t) + 1 are issued before the MFMAs of step
        // (c, t) — the compiler's own schedule waited on lgkmcnt(0) in front of every group of six MFMAs.
        // The widest instance (adjoint programs at RT = 5) has no registers left for the third plane's second buffer:
        // there the lo plane — used by one MFMA, issued last — is read at the top of its own step.
        constexpr bool LATE_LO = ADJ && RT == 5 && NPL >= 3 && !HF;
#ifndef GN_CHAIN_PD
#define GN_CHAIN_PD 1
#endif
        constexpr int PD = HF ? GN_CHAIN_PD : 1;   // steps of look-ahead (PD + 1 register buffers); 2 and 3 measured with six bf16 products: no change
        uint4 xf[PD + 1][3];
        auto xload = [&](uint4 (&f)[3], int c, int t) {
          const unsigned char* xp = xb + (16 * t) * ROWB + ((((c << 2) | lg) ^ l15) << 4);
          f[0] = *reinterpret_cast<const uint4*>(xp);
          if (NPL >= 2) f[1] = *reinterpret_cast<const uint4*>(xp + PLANE);
          if (NPL >= 3 && !LATE_LO) f[2] = *reinterpret_cast<const uint4*>(xp + 2 * PLANE);
        };
        xload(xf[0], 0, 0);
#pragma unroll
        for (int s0 = 1; s0 < PD; ++s0) {     // steps 1 .. PD - 1 of the look-ahead window
          const int c0 = s0 / RT, t0 = s0 - c0 * RT;
          if (c0 < kc) xload(xf[s0], c0, t0);
        }
#if defined(GN_EXP) && GN_EXP == 1
        xload(xf[1], 0, 0);
#endif
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c < kc) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, bcur[c][0]);
            bf16x8 wm, wl;
            if (NPL >= 2) wm = __builtin_bit_cast(bf16x8, bcur[c][NPL >= 2 ? 1 : 0]);
            if (NPL >= 3) wl = __builtin_bit_cast(bf16x8, bcur[c][NPL >= 3 ? 2 : 0]);
#pragma unroll
            for (int t = 0; t < RT; ++t) {
              const int cur = (c * RT + t) % (PD + 1);
#if !defined(GN_EXP) || GN_EXP != 1
              {
                const int s2 = c * RT + t + PD, c2 = s2 / RT, t2 = s2 - c2 * RT;   // the step PD ahead
                if (c2 < 4 && c2 < kc) xload(xf[s2 % (PD + 1)], c2, t2);
              }
#endif
#if defined(GN_EXP) && GN_EXP == 2
              { uint4 q = xf[cur][0]; for (int pl = 1; pl < NPL; ++pl) { q.x ^= xf[cur][pl].x; q.y ^= xf[cur][pl].y; q.z ^= xf[cur][pl].z; q.w ^= xf[cur][pl].w; }
                GN2_ACC(0)[0] += __uint_as_float(q.x); GN2_ACC(0)[1] += __uint_as_float(q.y); GN2_ACC(0)[2] += __uint_as_float(q.z); GN2_ACC(0)[3] += __uint_as_float(q.w); }
              continue;
#endif
              if constexpr (HF) {
                const f16x8 ah = __builtin_bit_cast(f16x8, bcur[c][0]), al = __builtin_bit_cast(f16x8, bcur[c][NPL >= 2 ? 1 : 0]);
                const f16x8 yh = __builtin_bit_cast(f16x8, xf[cur][0]), yl = __builtin_bit_cast(f16x8, xf[cur][1]);
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yh, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yl, acc[1][t], 0, 0, 0);
                acc[HF ? 2 : 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, yh, acc[HF ? 2 : 1][t], 0, 0, 0);
                continue;
              }
              const bf16x8 xh = __builtin_bit_cast(bf16x8, xf[cur][0]);
              bf16x8 xm, xl;
              if (NPL >= 2) xm = __builtin_bit_cast(bf16x8, xf[cur][1]);
              if (NPL >= 3 && !LATE_LO) xl = __builtin_bit_cast(bf16x8, xf[cur][2]);
              if (LATE_LO)
                xl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(
                    xb + (16 * t) * ROWB + ((((c << 2) | lg) ^ l15) << 4) + 2 * PLANE));
              if (NPL >= 3 && !LATE_LO) GN2_ACC(3) = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, GN2_ACC(3), 0, 0, 0);
              GN2_ACC(0) = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, GN2_ACC(0), 0, 0, 0);
              if (NPL >= 3) GN2_ACC(4) = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, GN2_ACC(4), 0, 0, 0);
              if (NPL >= 3) GN2_ACC(5) = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, GN2_ACC(5), 0, 0, 0);
              if (NPL >= 2) GN2_ACC(1) = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, GN2_ACC(1), 0, 0, 0);
              if (NPL >= 2) GN2_ACC(2) = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, GN2_ACC(2), 0, 0, 0);
              if (LATE_LO) GN2_ACC(3) = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, GN2_ACC(3), 0, 0, 0);
            }
          }
        }
      }
      float4 v[RT];            // the three partial sums collapse here: their registers are free for the prefetch below
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const v4f s = HF ? acc[0][t] + (acc[1][t] + acc[HF ? 2 : 1][t]) * H_DOWN : (NPL >= 2 ? acc[0][t] + acc[1][t] : acc[0][t]);
        v[t] = make_float4(s[0], s[1], s[2], s[3]);
      }
      // format H: the products are sigma_a times the true values; what this op leaves in LDS takes the smallest scale
      // (= the largest magnitude) among its LDS operands.  Every wave keeps its register copy current, active or not.
      float sy[HF ? RT : 1];
      if (HF) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const float sa = sg_get(a_slot, t);
          v[t] = mul4(v[t], inv_pow2(sa));
          float m = sa;
          if (res_slot == 0 || res_slot == 1) m = fminf(m, sg_get(res_slot, t));
          if (res2_slot == 0 || res2_slot == 1) m = fminf(m, sg_get(res2_slot, t));
          sy[HF ? t : 0] = m;
        }
      }
#undef GN2_ACC
      // next GEMM's fragments (no-op after the last one): in flight under the epilogue.  (Requested behind the epilogue's
      // global stores instead — so that the in-order vmcnt wait of the next op never covers a store — measured round 6:
      // no change, 1 082 vs 1 088 us per step of chain launches.)
      wload_op(bcur, gord);
#ifdef GN_CHAIN_TRACE
      if (active) { float sink = 0.f; for (int t = 0; t < RT; ++t) sink += v[t].x; if (sink == 1.2345e30f) smem[0] = 1; }
#endif
      GN2_STAMP(2);
      if (y_slot == a_slot || y2_slot == a_slot) lds_barrier();   // all reads of a_slot must finish before it is overwritten
      if (active) {
        // transposed tile: this lane holds columns n0 .. n0+3 of row 16 t + l15.  Written stage-major (one uniform
        // branch per stage, the RT float4 of a stage unrolled and independent): the row-major form serialised ~25
        // scalar branches and every dependent load per row block (5 k cycles per op at RT = 5, tools/chain2_trace.py).
        const int n0 = wave * 16 + (lg << 2);
        bool ok[RT];
        uint32_t off[RT];        // element offset of (row, n0) in an (M, N) matrix: M * 128 < 2^32 (checked on the host)
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const int64_t grow = row0 + 16 * t + l15;
          ok[t] = grow < M;
          off[t] = (uint32_t)grow * (uint32_t)N + (uint32_t)n0;
        }
#define GN2_EACH(body) _Pragma("unroll") for (int t = 0; t < RT; ++t) { body }
#define GN2_ADD(q) v[t].x += q.x; v[t].y += q.y; v[t].z += q.z; v[t].w += q.w;
#define GN2_MUL(q) v[t].x *= q.x; v[t].y *= q.y; v[t].z *= q.z; v[t].w *= q.w;
#define GN2_RES(q, b) v[t].x = (v[t].x + q.x) * b; v[t].y = (v[t].y + q.y) * b; v[t].z = (v[t].z + q.z) * b; v[t].w = (v[t].w + q.w) * b;
        if (gadd1) {
          float4 q[RT];
          GN2_EACH(q[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                   if (ok[t]) q[t] = *reinterpret_cast<const float4*>(gadd1 + (size_t)gidx1[row0 + 16 * t + l15] * N + n0);)
          GN2_EACH(GN2_ADD(q[t]))
        }
        if (gadd2) {
          float4 q[RT];
          GN2_EACH(q[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                   if (ok[t]) q[t] = *reinterpret_cast<const float4*>(gadd2 + (size_t)gidx2[row0 + 16 * t + l15] * N + n0);)
          GN2_EACH(GN2_ADD(q[t]))
        }
        if (pre_out && !pre_deriv) GN2_EACH(if (ok[t]) *reinterpret_cast<float4*>(pre_out + off[t]) = v[t];)
        if (act && pre_out && pre_deriv) {
          // the activation and its derivative from ONE sigmoid; `pre_out` receives ssilu'(z): a first-order adjoint then
          // multiplies by a stored factor instead of evaluating exp + rcp per element again (its epilogue was VALU-bound:
          // 4.5 k cycles per op against 2.2 k, tools/chain2_trace.py --adj)
          GN2_EACH(
            float4 d;
            gn_ssilu_pair(v[t].x, v[t].x, d.x); gn_ssilu_pair(v[t].y, v[t].y, d.y);
            gn_ssilu_pair(v[t].z, v[t].z, d.z); gn_ssilu_pair(v[t].w, v[t].w, d.w);
            if (ok[t]) *reinterpret_cast<float4*>(pre_out + off[t]) = d;)
        } else if (act) GN2_EACH(v[t].x = gn_ssilu(v[t].x); v[t].y = gn_ssilu(v[t].y); v[t].z = gn_ssilu(v[t].z); v[t].w = gn_ssilu(v[t].w);)
        GN2_STAMP(5);      // (trace build) epilogue: gathered adds, pre-activation store and activation done
        const bool want2 = ADJ && (y2_slot >= 0 || out2);
        // second output = v * alpha2 * phi2(Z2), written straight from the current value of v (no copy is kept):
        // before the `mul` stage when y2_src = 1, after the last stage otherwise
        auto emit_y2 = [&]() {
          float4 q[RT];
          if (Z2) {
            GN2_EACH(q[t] = make_float4(0.f, 0.f, 0.f, 0.f); if (ok[t]) q[t] = *reinterpret_cast<const float4*>(Z2 + off[t]);)
            if (mode2 == 0) GN2_EACH(q[t] = make_float4(gn_dssilu(q[t].x), gn_dssilu(q[t].y), gn_dssilu(q[t].z), gn_dssilu(q[t].w));)
            else if (mode2 == 2) GN2_EACH(q[t] = make_float4(gn_ssilu(q[t].x), gn_ssilu(q[t].y), gn_ssilu(q[t].z), gn_ssilu(q[t].w));)
            GN2_EACH(q[t].x *= v[t].x * alpha2; q[t].y *= v[t].y * alpha2; q[t].z *= v[t].z * alpha2; q[t].w *= v[t].w * alpha2;)
          } else {
            GN2_EACH(q[t] = make_float4(v[t].x * alpha2, v[t].y * alpha2, v[t].z * alpha2, v[t].w * alpha2);)
          }
          if (src_stage == 2 && srcP) {   // one row block at a time: no register room for RT more float4 triples here
            GN2_EACH(if (ok[t]) {
              const float4 zs = Z2 ? *reinterpret_cast<const float4*>(Z2 + off[t]) : make_float4(0.f, 0.f, 0.f, 0.f);
              const float4 pp = *reinterpret_cast<const float4*>(srcP + off[t]);
              const float4 qq = srcQ ? *reinterpret_cast<const float4*>(srcQ + off[t]) : make_float4(1.f, 1.f, 1.f, 1.f);
              const float4 sv = src_term(zs, pp, qq, src_mode, src_alpha);
              q[t].x += sv.x; q[t].y += sv.y; q[t].z += sv.z; q[t].w += sv.w;
            })
          }
          if (out2) GN2_EACH(if (ok[t]) *reinterpret_cast<float4*>(out2 + off[t]) = q[t];)
          if (y2_slot >= 0) GN2_EACH(slot_write(y2_slot, 16 * t + l15, n0, ok[t] ? (HF ? mul4(q[t], sy[HF ? t : 0]) : q[t]) : make_float4(0.f, 0.f, 0.f, 0.f));)
        };
        if (want2 && y2_src) emit_y2();
        if (ADJ && mul_slot == 2) GN2_EACH(GN2_MUL(park[ADJ ? t : 0]))
        else if (mul_slot >= 0) GN2_EACH(const float4 q = slot_read_acc(mul_slot, t); GN2_MUL(q))
        else if (mul_g) {
          float4 q[RT];
          if (HF && staged) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the LDS writes of the staged loads have landed
            GN2_EACH(q[t] = make_float4(1.f, 1.f, 1.f, 1.f);
                     if (ok[t]) q[t] = *reinterpret_cast<const float4*>(stage + t * 1024 + lane * 16);)
          } else {
            GN2_EACH(q[t] = make_float4(1.f, 1.f, 1.f, 1.f); if (ok[t]) q[t] = *reinterpret_cast<const float4*>(mul_g + off[t]);)
          }
          if (mul_mode == 2) GN2_EACH(q[t] = make_float4(gn_dssilu(q[t].x), gn_dssilu(q[t].y), gn_dssilu(q[t].z), gn_dssilu(q[t].w));)
          else if (mul_mode == 3) GN2_EACH(q[t] = make_float4(gn_ssilu(q[t].x), gn_ssilu(q[t].y), gn_ssilu(q[t].z), gn_ssilu(q[t].w));)
          GN2_EACH(GN2_MUL(q[t]))
        }
        if (alpha != 1.0f) GN2_EACH(v[t].x *= alpha; v[t].y *= alpha; v[t].z *= alpha; v[t].w *= alpha;)
        if (src_stage == 1 && srcP) {     // y += src_alpha * phis(mul_g) * P * Q, one row block at a time (see y2)
          GN2_EACH(if (ok[t]) {
            const float4 zs = (mul_g && src_mode == 1) ? *reinterpret_cast<const float4*>(mul_g + off[t]) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 pp = *reinterpret_cast<const float4*>(srcP + off[t]);
            const float4 qq = srcQ ? *reinterpret_cast<const float4*>(srcQ + off[t]) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 sv = src_term(zs, pp, qq, src_mode, src_alpha);
            v[t].x += sv.x; v[t].y += sv.y; v[t].z += sv.z; v[t].w += sv.w;
          })
        }
        if (ADJ && res_slot == 2) GN2_EACH(GN2_RES(park[ADJ ? t : 0], beta))
        else if (res_slot >= 0) GN2_EACH(const float4 q = slot_read_acc(res_slot, t); GN2_RES(q, beta))
        else if (res_g) {
          float4 q[RT];
          if (res_rows) GN2_EACH(q[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                                 if (ok[t]) q[t] = *reinterpret_cast<const float4*>(res_g + (size_t)res_rows[row0 + 16 * t + l15] * N + n0);)
          else GN2_EACH(q[t] = make_float4(0.f, 0.f, 0.f, 0.f); if (ok[t]) q[t] = *reinterpret_cast<const float4*>(res_g + off[t]);)
          GN2_EACH(if (ok[t]) { GN2_RES(q[t], beta) })
        }
        if (ADJ && res2_slot == 2) GN2_EACH(GN2_RES(park[ADJ ? t : 0], beta2))
        else if (res2_slot >= 0) GN2_EACH(const float4 q = slot_read_acc(res2_slot, t); GN2_RES(q, beta2))
        else if (res2_g) {
          float4 q[RT];
          GN2_EACH(q[t] = make_float4(0.f, 0.f, 0.f, 0.f); if (ok[t]) q[t] = *reinterpret_cast<const float4*>(res2_g + off[t]);)
          GN2_EACH(if (ok[t]) { GN2_RES(q[t], beta2) })
        }
        if (out) GN2_EACH(if (ok[t]) *reinterpret_cast<float4*>(out + off[t]) = v[t];)
        GN2_STAMP(6);      // mul / alpha / residual stages and the global store done
        if (ADJ && y_slot == 2) GN2_EACH(park[ADJ ? t : 0] = v[t];)
        else if (y_slot >= 0) GN2_EACH(slot_write(y_slot, 16 * t + l15, n0, ok[t] ? (HF ? mul4(v[t], sy[HF ? t : 0]) : v[t]) : make_float4(0.f, 0.f, 0.f, 0.f));)
        if (want2 && !y2_src) emit_y2();
        GN2_STAMP(7);      // plane split + LDS write (+ second output) issued
#undef GN2_EACH
#undef GN2_ADD
#undef GN2_MUL
#undef GN2_RES
      } else if ((N & 16) && wave * 16 == N && y_slot >= 0 && y_slot < 2) {
        // N = 16 (mod 32): the next GEMM reads k-chunks of 32 columns, so the 16 columns after N are zeroed
        const int n0 = wave * 16 + (lg << 2);
#pragma unroll
        for (int t = 0; t < RT; ++t) slot_write(y_slot, 16 * t + l15, n0, make_float4(0.f, 0.f, 0.f, 0.f));
      }
      if (HF) {
        // nobody reads `rs` inside a GEMM op (the epilogue uses the registers): wave 0 refreshes the LDS copy
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const float m = sy[HF ? t : 0];
          if (y_slot == 0 || y_slot == 1) { sg_set(y_slot, t, m); if (wave == 0 && lg == 0) rs[y_slot][16 * t + l15] = m; }
          if (ADJ && y2_slot >= 0) { sg_set(y2_slot, t, m); if (wave == 0 && lg == 0) rs[y2_slot][16 * t + l15] = m; }
        }
      }
      GN2_STAMP(3);
      lds_barrier();
      GN2_STAMP(4);
    }
  }
}

template <int RT, int NPL, bool ADJ, bool HF>
int launch_chain_split(const gn_chain_args* args, hipStream_t st) {
  int first = 0xff, linear = 1;
  for (int i = 0; i < args->n_ops; ++i)
    if (args->ops[i].kind == GN_OP_GEMM) {
      if (first == 0xff) first = i;
      if (args->ops[i].act & 1) linear = 0;
    }
  const int meta = first | (linear << 8);
  constexpr int BM = 16 * RT;
  constexpr size_t smem = (size_t)2 * (HF ? 2 : 3) * BM * ROWB + (HF ? (size_t)8 * RT * 1024 : 0);
  static std::atomic<bool> configured{false};   // set-once flag of an idempotent attribute (two racing threads both set it)
  if (!configured.load(std::memory_order_acquire)) {
    if (smem > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_split_kernel<RT, NPL, ADJ, HF>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
    }
    configured.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((chain_split_kernel<RT, NPL, ADJ, HF>), dim3(gn_cdiv(args->M, BM)), dim3(NT), smem, st, *args, meta);
  GN_LAUNCH_CHECK();
  return 0;
}

template <int NPL, bool ADJ, bool HF>
int dispatch_rt(const gn_chain_args* args, hipStream_t st) {
#ifdef GN2_TWO_WG
  const int rt = gn_cdiv(args->M, 512 * 16);
#else
  const int rt = gn_cdiv(args->M, 256 * 16);
#endif
  switch (rt <= 1 ? 1 : (rt >= 5 ? 5 : rt)) {
    case 1: return launch_chain_split<1, NPL, ADJ, HF>(args, st);
    case 2: return launch_chain_split<2, NPL, ADJ, HF>(args, st);
    case 3: return launch_chain_split<3, NPL, ADJ, HF>(args, st);
    case 4: return launch_chain_split<4, NPL, ADJ, HF>(args, st);
    default: return launch_chain_split<5, NPL, ADJ, HF>(args, st);
  }
}

template <int NPL, bool HF = false>
int dispatch_adj(const gn_chain_args* args, bool adj, hipStream_t st) {
  return adj ? dispatch_rt<NPL, true, HF>(args, st) : dispatch_rt<NPL, false, HF>(args, st);
}

// K index of element i (0..7) that lane group g (= lane / 16) supplies to k-chunk c of an MFMA: the natural order, or
// (GN_SPLIT_F16X2_ROW, csrc/chain4.hip) the order in which the accumulators of the PREVIOUS layer hold a row's columns —
// lane group g owns columns 16 j + 4 g .. + 3 of every column tile j, chunk c takes tiles 2 c and 2 c + 1
__device__ __forceinline__ int pack_k(int fmt, int c, int g, int i) {
  return fmt == GN_SPLIT_F16X2_ROW ? 16 * (2 * c + (i >> 2)) + 4 * g + (i & 3) : c * 32 + (g << 3) + i;
}

// eight k of one weight row -> the planes of (tile, chunk) unit tc: fmt 0 three bf16 planes, 1 two fp16 planes
__device__ __forceinline__ void pack_unit(const float (&x)[8], uint4* __restrict__ out, int tc, int lane, int fmt) {
  if (fmt == GN_SPLIT_F16X2 || fmt == GN_SPLIT_F16X2_ROW) {
    uint2 H0, L0, H1, L1;
    split4h(make_float4(x[0], x[1], x[2], x[3]), H0, L0);
    split4h(make_float4(x[4], x[5], x[6], x[7]), H1, L1);
    uint4* dst = out + ((size_t)tc * 2) * 64 + lane;
    dst[0] = make_uint4(H0.x, H0.y, H1.x, H1.y);
    dst[64] = make_uint4(L0.x, L0.y, L1.x, L1.y);
    return;
  }
  uint2 H0, M0, L0, H1, M1, L1;
  split4(make_float4(x[0], x[1], x[2], x[3]), H0, M0, L0);
  split4(make_float4(x[4], x[5], x[6], x[7]), H1, M1, L1);
  uint4* dst = out + ((size_t)tc * 3) * 64 + lane;
  dst[0] = make_uint4(H0.x, H0.y, H1.x, H1.y);
  dst[64] = make_uint4(M0.x, M0.y, M1.x, M1.y);
  dst[128] = make_uint4(L0.x, L0.y, L1.x, L1.y);
}

// W (N,K) row-major with pitch ldw (or, trans != 0, the (K,N) matrix whose transpose is the weight) -> fragment-major
// planes [col tile][k-chunk][plane][lane][8]: lane (n = lane % 16, g = lane / 16) of (tile, chunk) holds
// W[16 tile + n][32 chunk + 8 g + i], i = 0..7; rows >= N and columns >= K are zero.
__global__ void pack_weight_split_kernel(const float* __restrict__ W, int N, int K, int ldw, int trans, int fmt,
                                         uint4* __restrict__ out, int n_tiles, int kc) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // one thread per (tile, chunk, lane)
  if (idx >= n_tiles * kc * 64) return;
  const int lane = idx & 63, tc = idx >> 6, c = tc % kc, tile = tc / kc;
  const int n = tile * 16 + (lane & 15);
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = pack_k(fmt, c, lane >> 4, i);
    x[i] = (n < N && k < K) ? (trans ? W[(size_t)k * ldw + n] : W[(size_t)n * ldw + k]) : 0.f;
  }
  pack_unit(x, out, tc, lane, fmt);
}

// All weights of one training step in ONE launch: thread -> (job, tile, chunk, lane) by binary search over the jobs'
// first-unit table (a "unit" is one thread of pack_weight_split_kernel).
__global__ void pack_weight_split_grouped_kernel(const gn_pack_job* __restrict__ jobs, int n_jobs, int total_units) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total_units) return;
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].unit_begin <= idx) lo = mid; else hi = mid - 1;
  }
  const gn_pack_job j = jobs[lo];
  const int u = idx - j.unit_begin;
  const int kc = (j.K + 31) >> 5;
  const int lane = u & 63, tc = u >> 6, c = tc % kc, tile = tc / kc;
  const int n = tile * 16 + (lane & 15);
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = pack_k(j.fmt, c, lane >> 4, i);
    x[i] = (n < j.N && k < j.K) ? (j.trans ? j.W[(size_t)k * j.ldw + n] : j.W[(size_t)n * j.ldw + k]) : 0.f;
  }
  pack_unit(x, static_cast<uint4*>(j.out), tc, lane, j.fmt);
}

}  // namespace

extern "C" int gn_pack_weight_split_grouped(const gn_pack_job* jobs, int n_jobs, int total_units, void* stream) {
  if (n_jobs <= 0 || total_units <= 0) return 0;
  hipLaunchKernelGGL(pack_weight_split_grouped_kernel, dim3(gn_cdiv(total_units, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), jobs, n_jobs, total_units);
  GN_LAUNCH_CHECK();
  return 0;
}

#ifdef GN_CHAIN_TRACE
extern "C" int gn_chain2_trace_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gn_chain2_trace_buf), sizeof(gn_chain2_trace_buf));
}
#endif

extern "C" int64_t gn_pack_weight_split_bytes(int N, int K) {
  return (int64_t)gn_cdiv(N, 16) * gn_cdiv(K, 32) * 3 * 64 * 16;
}

extern "C" int gn_pack_weight_split_fmt(const float* W, int N, int K, int ldw, int trans, int fmt, void* out, void* stream) {
  if (N <= 0 || K <= 0) return 0;
  if (fmt != GN_SPLIT_BF16X3 && fmt != GN_SPLIT_F16X2 && fmt != GN_SPLIT_F16X2_ROW) return (int)hipErrorInvalidValue;
  const int n_tiles = gn_cdiv(N, 16), kc = gn_cdiv(K, 32);
  const int total = n_tiles * kc * 64;
  hipLaunchKernelGGL(pack_weight_split_kernel, dim3(gn_cdiv(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), W, N, K, ldw, trans, fmt, static_cast<uint4*>(out), n_tiles, kc);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_pack_weight_split(const float* W, int N, int K, int ldw, int trans, void* out, void* stream) {
  return gn_pack_weight_split_fmt(W, N, K, ldw, trans, GN_SPLIT_BF16X3, out, stream);
}

extern "C" int gn_chain_split_f32(const gn_chain_args* args, int nprod, void* stream) {
  if (args->M <= 0 || args->n_ops <= 0) return 0;
  if (args->n_ops > GN_CHAIN_MAX_OPS) return (int)hipErrorInvalidValue;
  if (args->M > (1 << 24)) return (int)hipErrorInvalidValue;
  // wide layout: tile height (GN_CHAIN_WIDE_ROWS) and start-up stagger (GN_CHAIN_WIDE_STAGGER) ride in the upper bits of `nprod`
  // — per call, no library state (ABI 13)
  const bool row = nprod == (GN_CHAIN_F16X2 | GN_CHAIN_ROW);
  if (row) nprod = GN_CHAIN_F16X2;
  const bool wide = (nprod & 0xfff) == (GN_CHAIN_F16X2 | GN_CHAIN_WIDE);
  const int wide_rows = ((nprod >> 12) & 0xf) * 8, wide_stagger = (nprod >> 16) & 0xffff;
  if (!wide && (nprod & ~0xfff)) return (int)hipErrorInvalidValue;
  if (wide && wide_rows > 48) return (int)hipErrorInvalidValue;
  if (nprod != 1 && nprod != 3 && nprod != 6 && nprod != GN_CHAIN_F16X2 && !wide) return (int)hipErrorInvalidValue;
  // the prologue requests the first GEMM's weights and publishes the op table without a barrier of its own: it relies on the
  // barrier every LOAD ends with, so a program must start with one (every generated program does)
  if (args->ops[0].kind != GN_OP_LOAD) return (int)hipErrorInvalidValue;
  for (int i = 0; i < args->n_ops; ++i) {
    const gn_chain_op& o = args->ops[i];
    if (o.kind == GN_OP_GEMM) {
      if (o.N <= 0 || o.N > SW || (o.N % 16) != 0 || o.K <= 0 || o.K > SW || (o.K % 4) != 0) return (int)hipErrorInvalidValue;
      if (o.a_slot < 0 || o.a_slot > 1 || o.slot > 2) return (int)hipErrorInvalidValue;
      if ((reinterpret_cast<uintptr_t>(o.W) & 15u) != 0) return (int)hipErrorInvalidValue;
      if (o.mul_slot > 2 || o.res_slot > 2 || o.res2_slot > 2 || o.y2_slot > 1) return (int)hipErrorInvalidValue;
      if (o.y2_slot >= 0 && (o.y2_slot == o.slot || o.y2_slot == o.mul_slot || o.y2_slot == o.res_slot ||
                             o.y2_slot == o.res2_slot)) return (int)hipErrorInvalidValue;
      if (o.src_stage < 0 || o.src_stage > 2 || (o.src_stage && !o.srcP)) return (int)hipErrorInvalidValue;
      if (o.src_stage == 1 && o.src_mode == 1 && !o.mul_g) return (int)hipErrorInvalidValue;
      if (o.src_stage == 2 && ((o.y2_slot < 0 && !o.out2) || (o.src_mode == 1 && !o.Z2))) return (int)hipErrorInvalidValue;
    } else {
      if (o.width <= 0 || o.width > SW || (o.width % 4) != 0 || (o.ld % 4) != 0) return (int)hipErrorInvalidValue;
      if (o.slot < 0 || o.slot > 2) return (int)hipErrorInvalidValue;
      if (o.slot == 2 && (o.kind != GN_OP_SCALE || o.src || o.out)) return (int)hipErrorInvalidValue;
      if (o.kind == GN_OP_LOAD && (o.y2_slot > 1 || o.y2_slot == o.slot)) return (int)hipErrorInvalidValue;
      if (o.kind == GN_OP_SCALE && (o.a_slot < 0 || o.a_slot > 1)) return (int)hipErrorInvalidValue;
      if (o.src_stage && (!o.srcP || o.kind == GN_OP_STORE || (o.kind == GN_OP_SCALE && (o.src_stage != 1 || o.slot == 2)) ||
                          (o.kind == GN_OP_LOAD && (o.src_stage != 2 || o.y2_slot < 0)))) return (int)hipErrorInvalidValue;
      if (o.src_stage && o.src_mode == 1 && !(o.kind == GN_OP_SCALE ? o.src : o.Z2)) return (int)hipErrorInvalidValue;
    }
  }
  bool adj = false;   // does the program touch the parking slot or a second output?
  for (int i = 0; i < args->n_ops; ++i) {
    const gn_chain_op& o = args->ops[i];
    if (o.kind == GN_OP_GEMM)
      adj = adj || o.slot == 2 || o.mul_slot == 2 || o.res_slot == 2 || o.res2_slot == 2 || o.y2_slot >= 0 || o.out2 ||
            o.src_stage != 0;
    else
      adj = adj || o.slot == 2 || (o.kind == GN_OP_LOAD && o.y2_slot >= 0) || o.src_stage != 0;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (row) return gn_chain_row_dispatch(args, adj, st);     // chain4.hip (its weights: GN_SPLIT_F16X2_ROW)
  if (wide) {      // chain3.hip: 4 waves x 32 columns, two workgroups per CU — it has no parking slot
    bool park = false;
    for (int i = 0; i < args->n_ops; ++i) {
      const gn_chain_op& o = args->ops[i];
      park = park || o.slot == 2 || (o.kind == GN_OP_GEMM && (o.mul_slot == 2 || o.res_slot == 2 || o.res2_slot == 2));
    }
    if (!park) return gn_chain_wide_dispatch(args, adj, wide_rows, wide_stagger, st);
    nprod = GN_CHAIN_F16X2;
  }
  if (nprod == GN_CHAIN_F16X2) return dispatch_adj<2, true>(args, adj, st);
  if (nprod == 6) return dispatch_adj<3>(args, adj, st);
  if (nprod == 3) return dispatch_adj<2>(args, adj, st);
  return dispatch_adj<1>(args, adj, st);
}
