// ubench_issue.hip -- gfx950 issue-rate microbenchmarks behind two DESIGN.md decisions:
//
//  (1) does a wave64 v_fma_f64 cost fewer cycles when only 32 / 16 of its lanes are active?  (If the SIMD skipped the passes of
//      inactive lane groups, B = 4096 could run as 32- or 16-lane groups: twice / four times the waves, each instruction cheaper.)
//  (2) what does fp64 MFMA (v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64) deliver against the fp64 vector rate, alone and
//      interleaved with v_fma_f64 (separate pipes?) -- the evidence for "no MFMA on this path".
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_issue tools/ubench_issue.hip ;  run on the GPU box: ./tools/ubench_issue
// One wave per workgroup; cycles are s_memtime ticks of the measuring wave (wave 0 of block 0 reports; medians over blocks too).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                                   \
   do                                                                                              \
   {                                                                                               \
      hipError_t e_ = (x);                                                                         \
      if (e_ != hipSuccess)                                                                        \
      {                                                                                            \
         fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                   \
         return 1;                                                                                 \
      }                                                                                            \
   } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int ITER = 2048;

// CHAINS independent dependent-FMA chains, `lanes` active lanes
template <int CHAINS>
__global__ void __launch_bounds__(64) fma_chain(double *out, long long *cycles, int lanes, double seed)
{
   const int lane = threadIdx.x;
   double acc[CHAINS];
#pragma unroll
   for (int c = 0; c < CHAINS; c++)
      acc[c] = seed + c + lane;
   const double m = 1.0000001, a = 1e-9;
   long long t0 = 0, t1 = 0;
   if (lane < lanes)
   {
      t0 = __builtin_amdgcn_s_memtime();
      for (int i = 0; i < ITER; i++)
      {
#pragma unroll
         for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++)
               asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(m), "v"(a));
      }
      asm volatile("s_waitcnt lgkmcnt(0)");
      t1 = __builtin_amdgcn_s_memtime();
   }
   double s = 0;
#pragma unroll
   for (int c = 0; c < CHAINS; c++)
      s += acc[c];
   out[blockIdx.x * 64 + lane] = s;
   if (lane == 0)
      cycles[blockIdx.x] = t1 - t0;
}

// back-to-back fp64 MFMA 16x16x4 (one accumulator chain or NACC independent ones)
template <int NACC>
__global__ void __launch_bounds__(64) mfma_f64_16(double *out, long long *cycles, double seed)
{
   const int lane = threadIdx.x;
   v4d acc[NACC];
#pragma unroll
   for (int c = 0; c < NACC; c++)
      acc[c] = v4d{seed, seed + 1, seed + 2, seed + 3};
   const double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * lane;
   long long t0 = __builtin_amdgcn_s_memtime();
   for (int i = 0; i < ITER; i++)
   {
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
         for (int c = 0; c < NACC; c++)
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
   }
   double s = 0;
#pragma unroll
   for (int c = 0; c < NACC; c++)
      s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
   long long t1 = __builtin_amdgcn_s_memtime();
   out[blockIdx.x * 64 + lane] = s;
   if (lane == 0)
      cycles[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ void __launch_bounds__(64) mfma_f64_4(double *out, long long *cycles, double seed)
{
   const int lane = threadIdx.x;
   double acc[NACC];
#pragma unroll
   for (int c = 0; c < NACC; c++)
      acc[c] = seed + c;
   const double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * lane;
   long long t0 = __builtin_amdgcn_s_memtime();
   for (int i = 0; i < ITER; i++)
   {
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
         for (int c = 0; c < NACC; c++)
            acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
   }
   double s = 0;
#pragma unroll
   for (int c = 0; c < NACC; c++)
      s += acc[c];
   long long t1 = __builtin_amdgcn_s_memtime();
   out[blockIdx.x * 64 + lane] = s;
   if (lane == 0)
      cycles[blockIdx.x] = t1 - t0;
}
// interleaved: per step one 16x16x4 MFMA (independent accumulators) and NF independent v_fma_f64
template <int NF>
__global__ void __launch_bounds__(64) mixed(double *out, long long *cycles, double seed)
{
   const int lane = threadIdx.x;
   v4d acc[4];
   double f[8];
#pragma unroll
   for (int c = 0; c < 4; c++)
      acc[c] = v4d{seed, seed + 1, seed + 2, seed + 3};
#pragma unroll
   for (int c = 0; c < 8; c++)
      f[c] = seed + c + lane;
   const double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * lane, m = 1.0000001, ad = 1e-9;
   long long t0 = __builtin_amdgcn_s_memtime();
   for (int i = 0; i < ITER; i++)
   {
#pragma unroll
      for (int c = 0; c < 4; c++)
      {
         acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
#pragma unroll
         for (int k = 0; k < NF; k++)
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[(c * NF + k) & 7]) : "v"(m), "v"(ad));
      }
   }
   double s = 0;
#pragma unroll
   for (int c = 0; c < 4; c++)
      s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
#pragma unroll
   for (int c = 0; c < 8; c++)
      s += f[c];
   long long t1 = __builtin_amdgcn_s_memtime();
   out[blockIdx.x * 64 + lane] = s;
   if (lane == 0)
      cycles[blockIdx.x] = t1 - t0;
}

static double median(std::vector<long long> v)
{
   std::sort(v.begin(), v.end());
   return (double)v[v.size() / 2];
}

int main()
{
   int ndev = 0;
   CHECK(hipGetDeviceCount(&ndev));
   if (!ndev)
   {
      fprintf(stderr, "no HIP device\n");
      return 1;
   }
   const int MAXB = 2048;
   double *out;
   long long *cyc;
   CHECK(hipMalloc(&out, sizeof(double) * 64 * MAXB));
   CHECK(hipMalloc(&cyc, sizeof(long long) * MAXB));
   std::vector<long long> h(MAXB);
   hipEvent_t e0, e1;
   CHECK(hipEventCreate(&e0));
   CHECK(hipEventCreate(&e1));
   auto report = [&](const char *name, int blocks, double insts_per_iter, double flop_per_inst, float ms) {
      hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
      std::vector<long long> v(h.begin(), h.begin() + blocks);
      const double c = median(v) / (ITER * insts_per_iter);
      const double total_flop = (double)blocks * ITER * insts_per_iter * flop_per_inst;
      printf("%-44s blocks %5d  ticks/inst %7.2f  wall %8.3f ms  %8.2f TFLOP/s\n", name, blocks, c, ms, total_flop / (ms * 1e-3) / 1e12);
   };
#define RUN(name, kern, blocks, ipi, fpi, ...)                               \
   do                                                                        \
   {                                                                         \
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, __VA_ARGS__);   \
      hipDeviceSynchronize();                                                \
      hipEventRecord(e0, 0);                                                 \
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, __VA_ARGS__);   \
      hipEventRecord(e1, 0);                                                 \
      hipEventSynchronize(e1);                                               \
      float ms_ = 0;                                                         \
      hipEventElapsedTime(&ms_, e0, e1);                                     \
      report(name, blocks, ipi, fpi, ms_);                                   \
   } while (0)

   printf("# s_memtime ticks: 100 MHz constant clock on gfx950 (1 tick = 10 ns = ~24 shader cycles at 2.4 GHz) unless ticks/inst says otherwise\n");
   for (int blocks : {1, 1024})
   {
      for (int lanes : {64, 32, 16, 1})
      {
         char nm[96];
         snprintf(nm, sizeof nm, "v_fma_f64 1 chain, %2d active lanes", lanes);
         RUN(nm, (fma_chain<1>), blocks, 8.0, 2.0 * lanes, out, cyc, lanes, 1.0);
         snprintf(nm, sizeof nm, "v_fma_f64 4 chains, %2d active lanes", lanes);
         RUN(nm, (fma_chain<4>), blocks, 32.0, 2.0 * lanes, out, cyc, lanes, 1.0);
      }
      RUN("v_mfma_f64_16x16x4 1 accumulator", (mfma_f64_16<1>), blocks, 4.0, 2048.0, out, cyc, 1.0);
      RUN("v_mfma_f64_16x16x4 4 accumulators", (mfma_f64_16<4>), blocks, 16.0, 2048.0, out, cyc, 1.0);
      RUN("v_mfma_f64_4x4x4_4b 1 accumulator", (mfma_f64_4<1>), blocks, 4.0, 512.0, out, cyc, 1.0);
      RUN("v_mfma_f64_4x4x4_4b 4 accumulators", (mfma_f64_4<4>), blocks, 16.0, 512.0, out, cyc, 1.0);
      RUN("mixed: 1 mfma16 + 0 fma per step", (mixed<0>), blocks, 4.0, 2048.0, out, cyc, 1.0);
      RUN("mixed: 1 mfma16 + 2 fma per step", (mixed<2>), blocks, 4.0, 2048.0 + 2 * 128.0, out, cyc, 1.0);
      RUN("mixed: 1 mfma16 + 4 fma per step", (mixed<4>), blocks, 4.0, 2048.0 + 4 * 128.0, out, cyc, 1.0);
      RUN("mixed: 1 mfma16 + 8 fma per step", (mixed<8>), blocks, 4.0, 2048.0 + 8 * 128.0, out, cyc, 1.0);
   }
   return 0;
}
