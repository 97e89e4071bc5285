// Shared pieces of the split-operand chain kernels (chain2.hip: 8 waves x 16 columns, one workgroup per CU;
// chain3.hip: 4 waves x 32 columns, several workgroups per CU): plane formats, LDS swizzle constants, small helpers.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

// chain3.hip: the wide layout of the two-plane fp16 arithmetic (argument checks are done by the caller, gn_chain_split_f32)
int gn_chain_wide_dispatch(const gn_chain_args* args, bool adj, int forced_tile_rows, int stagger, hipStream_t st);
// chain4.hip: the row-resident layout (a wave owns 16 rows and all columns; weights packed with GN_SPLIT_F16X2_ROW)
int gn_chain_row_dispatch(const gn_chain_args* args, bool adj, hipStream_t st);

namespace gn_split {


// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: every op would wait for its
// pre-activation / output stores (and the weight prefetch) to complete before the next op may start.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

constexpr int SW = 128;            // max N, K
constexpr int ROWB = 256;          // bytes per plane row: 128 bf16, no padding — 16-byte units are XOR-swizzled by the row
constexpr int NT = 512;

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  f32x2v v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));   // v_cvt_pk_bf16_f32, a in the low half
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// x (4 floats) -> three planes of 4 bf16 each; hi + mid + lo == x exactly (barring underflow)
__device__ __forceinline__ void split4(const float4 x, uint2& H, uint2& M, uint2& L) {
  H.x = pk_bf16(x.x, x.y); H.y = pk_bf16(x.z, x.w);
  const float r0 = x.x - bf_lo(H.x), r1 = x.y - bf_hi(H.x), r2 = x.z - bf_lo(H.y), r3 = x.w - bf_hi(H.y);
  M.x = pk_bf16(r0, r1); M.y = pk_bf16(r2, r3);
  const float s0 = r0 - bf_lo(M.x), s1 = r1 - bf_hi(M.x), s2 = r2 - bf_lo(M.y), s3 = r3 - bf_hi(M.y);
  L.x = pk_bf16(s0, s1); L.y = pk_bf16(s2, s3);
}
__device__ __forceinline__ float4 join4(const uint2 H, const uint2 M, const uint2 L) {
  return make_float4((bf_lo(H.x) + bf_lo(M.x)) + bf_lo(L.x), (bf_hi(H.x) + bf_hi(M.x)) + bf_hi(L.x),
                     (bf_lo(H.y) + bf_lo(M.y)) + bf_lo(L.y), (bf_hi(H.y) + bf_hi(M.y)) + bf_hi(L.y));
}

// The two-plane fp16 form (format H): x ~= hi + 2^-11 lo with hi = f16(x), lo = f16((x - hi) * 2^11) — 22 significand
// bits; three products hh + 2^-11 (hl + lh) (the correction terms accumulate in their own registers and are scaled once),
// the dropped ll term is below 2^-22 of the product.  Error-corrected half-precision GEMM in the manner of Ootomo & Yokota
// (2022); fp16 keeps 11 bits per plane where bf16 keeps 8, so two planes and three MFMAs do the work of three and six.
// Range: |x| < 65504 (larger values become inf and propagate: loud), full accuracy for |x| >= 2^-14 relative to the
// largest operands of a dot product — activations and first-order adjoints of the model; NOT for quantities whose
// magnitude follows an arbitrary loss scale (sweeps S3 / S4 of force training keep the bf16 planes).
constexpr float H_UP = 2048.f, H_DOWN = 1.f / 2048.f;
__device__ __forceinline__ uint32_t pk_f16(float a, float b) {
  f32x2v v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2v));   // round to nearest even, a in the low half
}
// v_fma_mix_f32 reads an fp16 half-register as an fma operand (no separate conversion): 12 instead of 16 VALU ops per split,
// 4 instead of 12 per join.  All of it exact: x - hi is representable in fp32 and so is every partial result.
// fmix_lo / fmix_hi: fma(f16 low / high half of h, b, c)
__device__ __forceinline__ float fmix_lo(uint32_t h, float b, float c) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float fmix_hi(uint32_t h, float b, float c) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(b), "v"(c));
  return d;
}
// fma(f16 half of l, b, f16 half of h)
__device__ __forceinline__ float fmix2_lo(uint32_t l, float b, uint32_t h) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(l), "v"(b), "v"(h));
  return d;
}
__device__ __forceinline__ float fmix2_hi(uint32_t l, float b, uint32_t h) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(l), "v"(b), "v"(h));
  return d;
}
__device__ __forceinline__ void split4h(const float4 x, uint2& H, uint2& L) {
  H.x = pk_f16(x.x, x.y); H.y = pk_f16(x.z, x.w);
  L.x = pk_f16(fmix_lo(H.x, -H_UP, x.x * H_UP), fmix_hi(H.x, -H_UP, x.y * H_UP));
  L.y = pk_f16(fmix_lo(H.y, -H_UP, x.z * H_UP), fmix_hi(H.y, -H_UP, x.w * H_UP));
}
__device__ __forceinline__ float4 join4h(const uint2 H, const uint2 L) {
  return make_float4(fmix2_lo(L.x, H_DOWN, H.x), fmix2_hi(L.x, H_DOWN, H.x), fmix2_lo(L.y, H_DOWN, H.y), fmix2_hi(L.y, H_DOWN, H.y));
}

// Row scale of format H in LINEAR programs (no activation: adjoint sweeps, whose rows may be arbitrarily small):
// sigma = 2^k with sigma * max|row| in [2^-4, 2^-3) — 2^18 of head room for what the program computes from the row before
// fp16 overflows, elements down to 2^-10 of the row maximum at full accuracy.  Exact (power of two); 1 for a zero row.
__device__ __forceinline__ float row_sigma(float m) {
  const uint32_t e = __float_as_uint(m) >> 23;                 // m >= 0: biased exponent
  const uint32_t ec = e < 4u ? 4u : (e > 250u ? 250u : e);
  return m > 0.f ? __uint_as_float((250u - ec) << 23) : 1.f;    // 2^(123 - e)
}
// max over the w4 (8 / 16 / 32) consecutive lanes that hold one row, in every lane of the group: DPP butterflies inside the
// 16-lane rows and one v_permlane16_swap across the row pair — the ds_bpermute form (__shfl_xor) cost 3.8 k cycles per LOAD
__device__ __forceinline__ float group_max(float m, int w4) {
#define GN2_DPP_MAX(ctrl) m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), ctrl, 0xf, 0xf, true)))
  GN2_DPP_MAX(0xB1);    // quad_perm [1,0,3,2]
  GN2_DPP_MAX(0x4E);    // quad_perm [2,3,0,1]
  GN2_DPP_MAX(0x141);   // row_half_mirror
  if (w4 >= 16) GN2_DPP_MAX(0x140);   // row_mirror
#undef GN2_DPP_MAX
  if (w4 >= 32) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  return m;
}
__device__ __forceinline__ float inv_pow2(float s) { return __uint_as_float(0x7f000000u - __float_as_uint(s)); }

// 16 bytes per lane, global -> LDS at lds_base + 16 * lane, no registers (global_load_lds_dwordx4, M0 = LDS base).
// Inline asm on purpose: with the builtin the compiler orders EVERY later ds_read behind the transfer (it cannot tell the
// staging area from the operand planes) and the MFMA phase waited for the loads it was meant to hide (3.2 k -> 7.4 k cycles).
// The consumer waits with an explicit s_waitcnt vmcnt(0); the compiler's own vmcnt bookkeeping stays conservative-correct
// (returns are in order: extra transfers in flight only make its waits cover more).
__device__ __forceinline__ void g2lds16(const float* gptr, uint32_t lds_base) {
  const uint32_t base = __builtin_amdgcn_readfirstlane(lds_base);     // wave-uniform by construction
  // M0 is a reserved register the compiler does not model as clobberable: it is saved and restored inside the statement
  // (the transfer reads M0 when it is issued)
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(base) : "memory");
}

// The op descriptors (gn_chain_op, 248 bytes = 4 cache lines each) are read from the kernel-argument segment one op at a
// time: every op began with a scalar-cache miss (0.6 - 0.9 k cycles between the barrier of one op and the first
// instruction of the next, tools/chain2_trace.py).  One dword of every 64-byte line of the segment (sizeof(gn_chain_args)
// = 4968 bytes: 78 lines) is requested here, once, while the first weight fragments are in flight; the descriptors then
// come from the scalar cache.  One statement with its own wait: the destination register is dead before and after.
__device__ __forceinline__ void kernarg_warm() {
  static_assert(sizeof(gn_chain_args) >= 77 * 64 + 4 && sizeof(gn_chain_args) <= 78 * 64, "one load per line of the argument block");
  uint32_t t;
  asm volatile(
    "s_load_dword %0, %1, 0x0\n\t"
    "s_load_dword %0, %1, 0x40\n\t"
    "s_load_dword %0, %1, 0x80\n\t"
    "s_load_dword %0, %1, 0xc0\n\t"
    "s_load_dword %0, %1, 0x100\n\t"
    "s_load_dword %0, %1, 0x140\n\t"
    "s_load_dword %0, %1, 0x180\n\t"
    "s_load_dword %0, %1, 0x1c0\n\t"
    "s_load_dword %0, %1, 0x200\n\t"
    "s_load_dword %0, %1, 0x240\n\t"
    "s_load_dword %0, %1, 0x280\n\t"
    "s_load_dword %0, %1, 0x2c0\n\t"
    "s_load_dword %0, %1, 0x300\n\t"
    "s_load_dword %0, %1, 0x340\n\t"
    "s_load_dword %0, %1, 0x380\n\t"
    "s_load_dword %0, %1, 0x3c0\n\t"
    "s_load_dword %0, %1, 0x400\n\t"
    "s_load_dword %0, %1, 0x440\n\t"
    "s_load_dword %0, %1, 0x480\n\t"
    "s_load_dword %0, %1, 0x4c0\n\t"
    "s_load_dword %0, %1, 0x500\n\t"
    "s_load_dword %0, %1, 0x540\n\t"
    "s_load_dword %0, %1, 0x580\n\t"
    "s_load_dword %0, %1, 0x5c0\n\t"
    "s_load_dword %0, %1, 0x600\n\t"
    "s_load_dword %0, %1, 0x640\n\t"
    "s_load_dword %0, %1, 0x680\n\t"
    "s_load_dword %0, %1, 0x6c0\n\t"
    "s_load_dword %0, %1, 0x700\n\t"
    "s_load_dword %0, %1, 0x740\n\t"
    "s_load_dword %0, %1, 0x780\n\t"
    "s_load_dword %0, %1, 0x7c0\n\t"
    "s_load_dword %0, %1, 0x800\n\t"
    "s_load_dword %0, %1, 0x840\n\t"
    "s_load_dword %0, %1, 0x880\n\t"
    "s_load_dword %0, %1, 0x8c0\n\t"
    "s_load_dword %0, %1, 0x900\n\t"
    "s_load_dword %0, %1, 0x940\n\t"
    "s_load_dword %0, %1, 0x980\n\t"
    "s_load_dword %0, %1, 0x9c0\n\t"
    "s_load_dword %0, %1, 0xa00\n\t"
    "s_load_dword %0, %1, 0xa40\n\t"
    "s_load_dword %0, %1, 0xa80\n\t"
    "s_load_dword %0, %1, 0xac0\n\t"
    "s_load_dword %0, %1, 0xb00\n\t"
    "s_load_dword %0, %1, 0xb40\n\t"
    "s_load_dword %0, %1, 0xb80\n\t"
    "s_load_dword %0, %1, 0xbc0\n\t"
    "s_load_dword %0, %1, 0xc00\n\t"
    "s_load_dword %0, %1, 0xc40\n\t"
    "s_load_dword %0, %1, 0xc80\n\t"
    "s_load_dword %0, %1, 0xcc0\n\t"
    "s_load_dword %0, %1, 0xd00\n\t"
    "s_load_dword %0, %1, 0xd40\n\t"
    "s_load_dword %0, %1, 0xd80\n\t"
    "s_load_dword %0, %1, 0xdc0\n\t"
    "s_load_dword %0, %1, 0xe00\n\t"
    "s_load_dword %0, %1, 0xe40\n\t"
    "s_load_dword %0, %1, 0xe80\n\t"
    "s_load_dword %0, %1, 0xec0\n\t"
    "s_load_dword %0, %1, 0xf00\n\t"
    "s_load_dword %0, %1, 0xf40\n\t"
    "s_load_dword %0, %1, 0xf80\n\t"
    "s_load_dword %0, %1, 0xfc0\n\t"
    "s_load_dword %0, %1, 0x1000\n\t"
    "s_load_dword %0, %1, 0x1040\n\t"
    "s_load_dword %0, %1, 0x1080\n\t"
    "s_load_dword %0, %1, 0x10c0\n\t"
    "s_load_dword %0, %1, 0x1100\n\t"
    "s_load_dword %0, %1, 0x1140\n\t"
    "s_load_dword %0, %1, 0x1180\n\t"
    "s_load_dword %0, %1, 0x11c0\n\t"
    "s_load_dword %0, %1, 0x1200\n\t"
    "s_load_dword %0, %1, 0x1240\n\t"
    "s_load_dword %0, %1, 0x1280\n\t"
    "s_load_dword %0, %1, 0x12c0\n\t"
    "s_load_dword %0, %1, 0x1300\n\t"
    "s_load_dword %0, %1, 0x1340\n\t"
    "s_waitcnt lgkmcnt(0)"
    : "=&s"(t) : "s"(__builtin_amdgcn_kernarg_segment_ptr()) : "memory");
}

// s = src_alpha * phis(z) * p * q  (second-order source term, include/gemnet_hip.h); z is only read when mode == 1
__device__ __forceinline__ float4 src_term(const float4 z, const float4 p, const float4 q, const int mode, const float a) {
  float4 s = make_float4(a * p.x * q.x, a * p.y * q.y, a * p.z * q.z, a * p.w * q.w);
  if (mode == 1) { s.x *= gn_d2ssilu(z.x); s.y *= gn_d2ssilu(z.y); s.z *= gn_d2ssilu(z.z); s.w *= gn_d2ssilu(z.w); }
  return s;
}

}  // namespace gn_split
