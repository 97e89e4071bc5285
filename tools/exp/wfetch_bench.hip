// Micro-benchmark (diagnosis, not product): how fast can every CU pull the SAME weight matrix out of L2 when all
// workgroups of a chain launch ask for it at once?  DESIGN.md §5 measured 64 KB in 8-14 k cycles inside chain_kernel
// (5-8 B/clk/CU against a 64 B/clk/CU L2 port).  Variants: the chain kernel's strided fragment loads, a pre-packed
// contiguous layout, the same with a per-workgroup rotated start (spreads the 32 CUs of an XCD over the L2 channels),
// and cooperative global->LDS copies.   hipcc --offload-arch=gfx950 -O3 wfetch_bench.hip -o bin/wfetch_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NT = 512;
constexpr int NOPS = 24;     // ops per launch, each with its own weight
constexpr int NW = 32;       // distinct weights in the pool (ops cycle through them)

// mode 0: chain_kernel pattern: wave w, lane (l15, lg): row 16w + l15, 8 float4 at col 16c + 4lg (row stride 512 B)
// mode 1: packed: wave w reads 8 KB contiguous, 8 x (64 lanes x 16 B)
// mode 2: packed + chunk order rotated by blockIdx
// mode 3: cooperative copy to LDS, linear
// mode 4: cooperative copy to LDS, start rotated by blockIdx (in 4 KB units)
// mode 5: like 4 but rotation in 256 B units x prime
template <int MODE, int KB>
__global__ __launch_bounds__(NT) void fetch_kernel(const float* __restrict__ pool, float* __restrict__ sink,
                                                   unsigned long long* __restrict__ cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  constexpr int WBYTES = KB * 1024;
  constexpr int WFLOATS = WBYTES / 4;
  constexpr int V4_PER_THREAD = WBYTES / 16 / NT;   // 8 for 64 KB, 12 for 96 KB
  float acc = 0.f;
  unsigned long long t0 = clock64();
  for (int op = 0; op < NOPS; ++op) {
    const float* __restrict__ W = pool + (size_t)((op * 7 + 3) % NW) * WFLOATS;
    if (MODE == 0) {
      const float* row = W + (size_t)(wave * 16 + l15) * (WFLOATS / 128) + (lg << 2);
      float4 v[V4_PER_THREAD];
#pragma unroll
      for (int c = 0; c < V4_PER_THREAD; ++c) v[c] = *reinterpret_cast<const float4*>(row + c * 16);
#pragma unroll
      for (int c = 0; c < V4_PER_THREAD; ++c) acc += v[c].x + v[c].y + v[c].z + v[c].w;
    } else if (MODE == 1 || MODE == 2) {
      const float* base = W + (size_t)wave * (WFLOATS / 8);
      float4 v[V4_PER_THREAD];
      const int rot = MODE == 2 ? (blockIdx.x * 5) % V4_PER_THREAD : 0;
#pragma unroll
      for (int c = 0; c < V4_PER_THREAD; ++c) {
        int cc = c + rot; if (cc >= V4_PER_THREAD) cc -= V4_PER_THREAD;
        v[c] = *reinterpret_cast<const float4*>(base + cc * 256 + lane * 4);
      }
#pragma unroll
      for (int c = 0; c < V4_PER_THREAD; ++c) acc += v[c].x + v[c].y + v[c].z + v[c].w;
    } else {
      float4 v[V4_PER_THREAD];
      int rot = 0;
      if (MODE == 4) rot = ((blockIdx.x * 3) % (WBYTES / 4096)) * 1024;          // floats
      if (MODE == 5) rot = ((blockIdx.x * 37) % (WBYTES / 256)) * 64;
#pragma unroll
      for (int c = 0; c < V4_PER_THREAD; ++c) {
        int off = (c * NT + tid) * 4 + rot; if (off >= WFLOATS) off -= WFLOATS;
        v[c] = *reinterpret_cast<const float4*>(W + off);
      }
      float* dst = lds + (op & 1) * 0;   // single buffer: the point is the fetch rate
#pragma unroll
      for (int c = 0; c < V4_PER_THREAD; ++c) {
        int off = (c * NT + tid) * 4 + rot; if (off >= WFLOATS) off -= WFLOATS;
        *reinterpret_cast<float4*>(dst + off) = v[c];
      }
      __syncthreads();
      acc += lds[(tid * 33 + op) % WFLOATS];
    }
    __syncthreads();
  }
  unsigned long long t1 = clock64();
  if (acc == 1.2345e30f) sink[0] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int KB>
void run(const char* name, const float* pool, float* sink, unsigned long long* cyc, int grid) {
  size_t smem = MODE >= 3 ? (size_t)KB * 1024 : 0;
  if (smem > 64 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fetch_kernel<MODE, KB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fetch_kernel<MODE, KB>), dim3(grid), dim3(NT), smem, 0, pool, sink, cyc);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((fetch_kernel<MODE, KB>), dim3(grid), dim3(NT), smem, 0, pool, sink, cyc);
  CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned long long> h(grid);
  CK(hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double mean = 0, mx = 0; for (auto c : h) { mean += c; if (c > mx) mx = c; } mean /= grid;
  const double us_op = ms * 1e3 / reps / NOPS;
  printf("%-44s grid %3d  %2d KB/op  %7.2f us/op  wave-clock cycles/op mean %7.0f max %7.0f  -> %5.1f B/clk/CU (mean)  aggregate %6.2f TB/s\n",
         name, grid, KB, us_op, mean / NOPS, mx / NOPS, KB * 1024.0 / (mean / NOPS), grid * KB * 1024.0 / (us_op * 1e-6) / 1e12);
}

int main() {
  float* pool; float* sink; unsigned long long* cyc;
  const size_t pool_bytes = (size_t)NW * 96 * 1024;
  CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 0, pool_bytes));
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&cyc, 4096 * sizeof(unsigned long long)));
  for (int grid : {227, 256, 512}) {
    run<0, 64>("0 chain fragment loads (16 rows x 64 B)", pool, sink, cyc, grid);
    run<1, 64>("1 packed contiguous per wave", pool, sink, cyc, grid);
    run<2, 64>("2 packed + rotated chunk order", pool, sink, cyc, grid);
    run<3, 64>("3 cooperative copy -> LDS, linear", pool, sink, cyc, grid);
    run<4, 64>("4 cooperative copy -> LDS, rot 4 KB", pool, sink, cyc, grid);
    run<5, 64>("5 cooperative copy -> LDS, rot 256 B x 37", pool, sink, cyc, grid);
    run<3, 96>("3 cooperative copy -> LDS, linear", pool, sink, cyc, grid);
    run<4, 96>("4 cooperative copy -> LDS, rot 4 KB", pool, sink, cyc, grid);
    run<5, 96>("5 cooperative copy -> LDS, rot 256 B x 37", pool, sink, cyc, grid);
  }
  return 0;
}
