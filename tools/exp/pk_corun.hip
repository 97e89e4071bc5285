// Standalone repro of the round-4 packed-FP32 finding (docs/HISTORY.md section 11): NO torch, NO product library —
// this file + three kernel sources of the repo, one hipcc command:
//
//   cd tools/exp && sh pk_corun.sh            (builds both variants, dumps the ISA of the victim kernel, runs them on gfx950)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGN_AGG_PK -I ../../include pk_corun.hip \
//         ../../gemnet_pytorch_amd/csrc/aggregate.hip ../../gemnet_pytorch_amd/csrc/chain2.hip \
//         ../../gemnet_pytorch_amd/csrc/chain3.hip -o pk_corun_pk         (victim WITH v_pk_*_f32: the failing form)
//   ... the same without -DGN_AGG_PK -> pk_corun_nopk                      (victim compiled with no-packed-fp32-ops: passes)
//
// What it does: constant inputs; branch A of a two-branch hipGraph = six Dense-stack chain programs (MFMA-heavy,
// gn_chain_split_f32 in the two-fp16-plane arithmetic), branch B = six launches of the fused edge -> atom aggregation, forward
// and adjoint (gn_rbf_aggregate_fwd/bwd_f32: bandwidth-bound VALU kernels), every launch with output buffers of its own.  The
// eager (serial) results are the reference; every replay's outputs are compared BIT FOR BIT.  No kernel writes memory another
// one reads, every kernel alone and the one-branch graph are reproducible: any mismatch is produced below the application.
// Observed on MI355X / ROCm 7.2 with the packed build: 25-45 of 60 replays wrong, always the low halves of the float2 of lanes
// 48..63 of rbf_aggregate_bwd_kernel (even columns 96..126 of a few rows of g_m); 0 of 60 with no-packed-fp32-ops.
//
// Usage: pk_corun [replays = 200] [branches = 2 | 1]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/gemnet_hip.h"

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) { std::printf("%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(2); } \
  } while (0)
#define GN(x)                                                                                     \
  do {                                                                                            \
    int e_ = (x);                                                                                 \
    if (e_ != 0) { std::printf("%s:%d: gn error %d (%s)\n", __FILE__, __LINE__, e_, hipGetErrorString((hipError_t)e_)); std::exit(2); } \
  } while (0)

// gn_error_string / gn_abi_version live in basis.hip, which this repro does not link
extern "C" const char* gn_error_string(int code) { return hipGetErrorString((hipError_t)code); }

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float rnd() {   // uniform in (-1, 1), deterministic
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (float)((double)(rng_state >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}
static float* dev_rand(size_t n, float scale) {
  std::vector<float> h(n);
  for (auto& v : h) v = rnd() * scale;
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}
static float* dev_empty(size_t n) {
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemset(d, 0, n * sizeof(float)));
  return d;
}

struct Program {
  gn_chain_args args;
  std::vector<float*> outs;
  int M;
};

// LOAD x; three GEMMs 128 x 128 (adjoint form: global Hadamard factor + LDS residual; forward form: ScaledSiLU + stored
// derivative), each writing its own (M, 128) output — tools/exp/h3_concurrency.py::program
static Program make_program(int M, bool adj) {
  Program p{};
  p.M = M;
  std::memset(&p.args, 0, sizeof(p.args));
  p.args.M = M;
  float* x = dev_rand((size_t)M * 128, 1.0f);
  int n = 0;
  auto blank = [&](int kind) -> gn_chain_op& {
    gn_chain_op& o = p.args.ops[n++];
    o.kind = kind;
    o.slot = o.a_slot = o.mul_slot = o.res_slot = o.res2_slot = o.y2_slot = -1;
    o.alpha = o.beta = o.beta2 = o.alpha2 = 1.0f;
    return o;
  };
  gn_chain_op& ld = blank(GN_OP_LOAD);
  ld.slot = 0; ld.width = 128; ld.ld = 128; ld.src = x;
  int cur = 0, oth = 1;
  for (int i = 0; i < 3; ++i) {
    float* W = dev_rand(128 * 128, 1.0f / 11.0f);
    void* Wp;
    CK(hipMalloc(&Wp, (size_t)gn_pack_weight_split_bytes(128, 128)));
    GN(gn_pack_weight_split_fmt(W, 128, 128, 128, 0, GN_SPLIT_F16X2, Wp, nullptr));
    float* z = dev_rand((size_t)M * 128, 1.0f);
    float* out = dev_empty((size_t)M * 128);
    gn_chain_op& g = blank(GN_OP_GEMM);
    g.W = static_cast<const float*>(Wp); g.N = 128; g.K = 128; g.a_slot = cur; g.slot = oth; g.out = out;
    if (adj) { g.mul_g = z; g.mul_mode = 1; g.res_slot = cur; }
    else { g.act = 3; g.pre_out = z; }
    p.outs.push_back(out);
    std::swap(cur, oth);
  }
  p.args.n_ops = n;
  return p;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? std::atoi(argv[1]) : 200;
  const int branches = argc > 2 ? std::atoi(argv[2]) : 2;
  const int A = 512, E = 11824, C = 128, R = 16, NB = 6;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
#ifdef GN_AGG_PK
  const char* variant = "victim WITH packed FP32 (v_pk_*_f32)";
#else
  const char* variant = "victim compiled with no-packed-fp32-ops";
#endif
  std::printf("device %s (%s), %s, %d replays, %d branch(es)\n", prop.name, prop.gcnArchName, variant, reps, branches);

  // ---- branch B operands: edges sorted by target atom (CSR), constant
  float* m = dev_rand((size_t)E * C, 1.0f);
  float* rbf = dev_rand((size_t)E * R, 1.0f);
  float* W = dev_rand((size_t)C * R, 0.25f);
  float* gout = dev_rand((size_t)A * C, 1.0f);
  std::vector<int32_t> ida(E), seg(A + 1, 0);
  for (int e = 0; e < E; ++e) ida[e] = (int32_t)((int64_t)e * A / E);
  for (int e = 0; e < E; ++e) seg[ida[e] + 1]++;
  for (int a = 0; a < A; ++a) seg[a + 1] += seg[a];
  int32_t *d_ida, *d_seg;
  CK(hipMalloc(&d_ida, E * sizeof(int32_t)));
  CK(hipMalloc(&d_seg, (A + 1) * sizeof(int32_t)));
  CK(hipMemcpy(d_ida, ida.data(), E * sizeof(int32_t), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_seg, seg.data(), (A + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
  struct BOut { float *fwd, *gm, *grbf; };
  std::vector<BOut> bo(NB);
  for (auto& o : bo) o = {dev_empty((size_t)A * C), dev_empty((size_t)E * C), dev_empty((size_t)E * R)};

  // ---- branch A: six chain programs (edge rows and atom rows, adjoint and forward forms)
  const int Ms[6] = {E, E, A, E, E, A};
  const bool adjs[6] = {true, false, true, true, false, false};
  std::vector<Program> progs;
  for (int i = 0; i < 6; ++i) progs.push_back(make_program(Ms[i], adjs[i]));
  CK(hipDeviceSynchronize());

  auto branch_a = [&](hipStream_t st) {
    for (auto& p : progs) GN(gn_chain_split_f32(&p.args, GN_CHAIN_F16X2, st));
  };
  auto branch_b = [&](hipStream_t st) {
    for (auto& o : bo) {
      GN(gn_rbf_aggregate_fwd_f32(m, rbf, W, nullptr, d_seg, o.fwd, A, C, R, 0.3f, st));
      GN(gn_rbf_aggregate_bwd_f32(gout, m, rbf, W, d_ida, o.gm, o.grbf, E, C, R, 0.3f, 0, st));
    }
  };

  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreate(&sb));
  // eager reference (serial: one stream)
  branch_a(sa);
  branch_b(sa);
  CK(hipStreamSynchronize(sa));
  auto fetch = [&](const float* d, size_t n) {
    std::vector<float> h(n);
    CK(hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
    return h;
  };
  std::vector<std::vector<float>> ref_a, ref_gm, ref_grbf, ref_fwd;
  for (auto& p : progs)
    for (float* o : p.outs) ref_a.push_back(fetch(o, (size_t)p.M * 128));
  for (auto& o : bo) {
    ref_fwd.push_back(fetch(o.fwd, (size_t)A * C));
    ref_gm.push_back(fetch(o.gm, (size_t)E * C));
    ref_grbf.push_back(fetch(o.grbf, (size_t)E * R));
  }

  // ---- capture
  hipGraph_t graph;
  hipGraphExec_t exec;
  hipEvent_t fork, join;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  CK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
  if (branches == 2) {
    CK(hipEventRecord(fork, sa));
    CK(hipStreamWaitEvent(sb, fork, 0));
    branch_b(sb);
    branch_a(sa);
    CK(hipEventRecord(join, sb));
    CK(hipStreamWaitEvent(sa, join, 0));
  } else {
    branch_b(sa);
    branch_a(sa);
  }
  CK(hipStreamEndCapture(sa, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));

  int bad_chain = 0, bad_agg = 0, shown = 0;
  long lane_hist[64] = {0};
  long low_half = 0, high_half = 0;
  for (int it = 0; it < reps; ++it) {
    for (auto& o : bo) {     // poison the outputs: a replay must rewrite every element
      CK(hipMemsetAsync(o.gm, 0xff, (size_t)E * C * sizeof(float), sa));
      CK(hipMemsetAsync(o.grbf, 0xff, (size_t)E * R * sizeof(float), sa));
    }
    CK(hipGraphLaunch(exec, sa));
    CK(hipStreamSynchronize(sa));
    bool wa = false, wb = false;
    size_t k = 0;
    for (auto& p : progs)
      for (float* o : p.outs) {
        auto h = fetch(o, (size_t)p.M * 128);
        wa = wa || std::memcmp(h.data(), ref_a[k].data(), h.size() * sizeof(float)) != 0;
        ++k;
      }
    for (int b = 0; b < NB; ++b) {
      auto hf = fetch(bo[b].fwd, (size_t)A * C), hg = fetch(bo[b].gm, (size_t)E * C), hr = fetch(bo[b].grbf, (size_t)E * R);
      const bool w = std::memcmp(hf.data(), ref_fwd[b].data(), hf.size() * 4) != 0 ||
                     std::memcmp(hg.data(), ref_gm[b].data(), hg.size() * 4) != 0 ||
                     std::memcmp(hr.data(), ref_grbf[b].data(), hr.size() * 4) != 0;
      wb = wb || w;
      if (w) {
        long n_gm = 0;
        int first_row = -1;
        for (size_t i = 0; i < hg.size(); ++i)
          if (std::memcmp(&hg[i], &ref_gm[b][i], 4) != 0) {
            ++n_gm;
            const int col = (int)(i % C);
            lane_hist[col / 2]++;                 // the adjoint's lane l owns columns 2 l, 2 l + 1 (a float2)
            (col & 1 ? high_half : low_half)++;
            if (first_row < 0) first_row = (int)(i / C);
          }
        if (shown < 4) {
          std::printf("  replay %d, aggregation launch %d: %ld wrong elements of g_m (first row %d)\n", it, b, n_gm, first_row);
          ++shown;
        }
      }
    }
    bad_chain += wa;
    bad_agg += wb;
  }
  std::printf("chain outputs differ from the serial run in %d / %d replays, aggregation outputs in %d / %d\n", bad_chain, reps,
              bad_agg, reps);
  if (bad_agg) {
    std::printf("wrong g_m elements by lane of the adjoint kernel (lane = column / 2): ");
    for (int l = 0; l < 64; ++l)
      if (lane_hist[l]) std::printf("%d:%ld ", l, lane_hist[l]);
    std::printf("\nlow halves (even columns) %ld, high halves (odd columns) %ld\n", low_half, high_half);
  }
  std::printf("RESULT %s\n", (bad_chain || bad_agg) ? "MISMATCH" : "bit-identical");
  return 0;
}
