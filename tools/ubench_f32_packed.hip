// ubench_f32_packed.hip -- does v_pk_fma_f32 (two fp32 FMAs per lane and instruction) raise the fp32 vector rate of gfx950 for the shapes
// the run-time-topology kernels run at (one to four waves per SIMD, no MFMA beside them)?  If it does, two configurations per lane
// (float2 arithmetic) would halve the vector instructions per configuration of the fp32 kernels (config 5).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_f32_packed tools/ubench_f32_packed.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

template <int CHAINS>
__global__ void __launch_bounds__(64) fma32(float *out, float seed)
{
   float acc[CHAINS];
#pragma unroll
   for (int c = 0; c < CHAINS; c++)
      acc[c] = seed + c + threadIdx.x;
   const float m = 1.0000001f, a = 1e-9f;
   for (int i = 0; i < ITER; i++)
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
         for (int c = 0; c < CHAINS; c++)
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(m), "v"(a));
   float s = 0;
#pragma unroll
   for (int c = 0; c < CHAINS; c++)
      s += acc[c];
   out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CHAINS>
__global__ void __launch_bounds__(64) pkfma32(float *out, float seed)
{
   f2 acc[CHAINS];
#pragma unroll
   for (int c = 0; c < CHAINS; c++)
      acc[c] = f2{seed + c + threadIdx.x, seed - c};
   const f2 m{1.0000001f, 0.9999999f}, a{1e-9f, 2e-9f};
   for (int i = 0; i < ITER; i++)
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
         for (int c = 0; c < CHAINS; c++)
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(m), "v"(a));
   float s = 0;
#pragma unroll
   for (int c = 0; c < CHAINS; c++)
      s += acc[c].x + acc[c].y;
   out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CHAINS>
__global__ void __launch_bounds__(64) pkmuladd32(float *out, float seed)
{ // v_pk_mul_f32 + v_pk_add_f32 (the non-fused pair the kernels' written order mostly uses)
   f2 acc[CHAINS];
#pragma unroll
   for (int c = 0; c < CHAINS; c++)
      acc[c] = f2{seed + c + threadIdx.x, seed - c};
   const f2 m{1.0000001f, 0.9999999f}, a{1e-9f, 2e-9f};
   for (int i = 0; i < ITER; i++)
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
         for (int c = 0; c < CHAINS; c++)
         {
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[c]) : "v"(m));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[c]) : "v"(a));
         }
   float s = 0;
#pragma unroll
   for (int c = 0; c < CHAINS; c++)
      s += acc[c].x + acc[c].y;
   out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main()
{
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || !ndev)
   {
      fprintf(stderr, "no HIP device\n");
      return 1;
   }
   float *out;
   if (hipMalloc(&out, sizeof(float) * 64 * 8192) != hipSuccess)
      return 1;
   hipEvent_t e0, e1;
   hipEventCreate(&e0), hipEventCreate(&e1);
#define RUN(name, kern, blocks, flop_per_inst)                                                                                   \
   do                                                                                                                            \
   {                                                                                                                             \
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 1.0f);                                                         \
      hipDeviceSynchronize();                                                                                                    \
      hipEventRecord(e0, 0);                                                                                                     \
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 1.0f);                                                         \
      hipEventRecord(e1, 0);                                                                                                     \
      hipEventSynchronize(e1);                                                                                                   \
      float ms = 0;                                                                                                              \
      hipEventElapsedTime(&ms, e0, e1);                                                                                          \
      const double insts = (double)ITER * 4 * 8; /* (the one-chain runs execute an eighth of that: their ns/inst and TFLOP/s read x8 and /8) */                                                                                 \
      printf("%-34s waves %5d  wall %8.3f ms  ns/inst/wave %6.3f  %7.2f TFLOP/s\n", name, blocks, ms, ms * 1e6 / insts,          \
             (double)blocks * insts * 64 * flop_per_inst / (ms * 1e-3) / 1e12);                                                  \
   } while (0)
   for (int blocks : {256, 1024, 2048, 4096, 8192})
   {
      RUN("v_fma_f32, 8 chains", fma32<8>, blocks, 2);
      RUN("v_pk_fma_f32, 8 chains", pkfma32<8>, blocks, 4);
      RUN("v_pk_mul_f32 + v_pk_add_f32, 8 chains", pkmuladd32<8>, blocks, 2); /* 2 flops per instruction and lane */
   }
   RUN("v_fma_f32, 1 chain (dependent)", fma32<1>, 1024, 2);
   RUN("v_pk_fma_f32, 1 chain (dependent)", pkfma32<1>, 1024, 4);
   return 0;
}
