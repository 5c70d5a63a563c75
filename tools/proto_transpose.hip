// Stand-alone timing of the AoS <-> SoA row-block transposers (mh_kernels.h) and variants, against a float4 copy of the same bytes.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mecano_amd/csrc tools/proto_transpose.hip -o build/proto_transpose && build/proto_transpose [B] [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float VT __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int V = 4, R = 32, LPC = 8, MAXN = 512;

template <bool NT> __device__ __forceinline__ VT ld(const float *p) { if constexpr (NT) return __builtin_nontemporal_load((const VT *)p); else return *(const VT *)p; }
template <bool NT> __device__ __forceinline__ void st(float *p, VT v) { if constexpr (NT) __builtin_nontemporal_store(v, (VT *)p); else *(VT *)p = v; }

// ORDER 0: block b, b + grid, ...; 1: a contiguous run of blocks per workgroup
template <int ORDER, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) r2c(const float *__restrict__ src, float *__restrict__ dst, long B, int n, long blocks)
{
   constexpr int NU = MAXN * R / (256 * V);
   extern __shared__ double lds_raw[];
   float *const blk = (float *)lds_raw;
   const int rb = (threadIdx.x % LPC) * V, jl = threadIdx.x / LPC;
   VT reg[NU];
   long b, end, step;
   if (ORDER == 0) b = blockIdx.x, end = blocks, step = gridDim.x;
   else { const long per = (blocks + gridDim.x - 1) / gridDim.x; b = blockIdx.x * per; end = std::min(blocks, b + per); step = 1; }
   auto request = [&](long bb) {
      const long r0 = bb * R;
      const int len = (int)(B - r0 < R ? B - r0 : R) * n;
      const float *const flat = src + r0 * n;
#pragma unroll
      for (int u = 0; u < NU; u++)
         if ((threadIdx.x + 256 * u) * V + V <= len)
            reg[u] = ld<NTL>(flat + (threadIdx.x + 256 * u) * V);
   };
   if (b < end) request(b);
   for (; b < end; b += step)
   {
      const long r0 = b * R;
      const int rows = (int)(B - r0 < R ? B - r0 : R), len = rows * n;
#pragma unroll
      for (int u = 0; u < NU; u++)
         if ((threadIdx.x + 256 * u) * V + V <= len)
            *(VT *)(blk + (threadIdx.x + 256 * u) * V) = reg[u];
      lds_only_barrier();
      if (b + step < end) request(b + step);
      if (rb < rows)
         for (int j = jl; j < n; j += 256 / LPC)
         {
            VT w;
#pragma unroll
            for (int k = 0; k < V; k++) w[k] = blk[(rb + k) * n + j];
            st<NTS>(dst + (long)j * B + r0 + rb, w);
         }
      lds_only_barrier();
   }
}
// variant: a workgroup of 512 threads takes 64 rows of HALF the columns?  no: the AoS side would be in pieces.  Variant W: wider column
// segments -- R2 = 64 rows in two LDS halves, 256-byte segments per column (two lanes' vectors adjacent)
template <bool NTL, bool NTS>
__global__ void __launch_bounds__(256) c2r(const float *__restrict__ src, float *__restrict__ dst, long B, int n, long blocks)
{
   constexpr int NU = MAXN / (256 / LPC);
   extern __shared__ double lds_raw[];
   float *const blk = (float *)lds_raw;
   const int rb = (threadIdx.x % LPC) * V, jl = threadIdx.x / LPC;
   VT reg[NU];
   auto request = [&](long b) {
      const long r0 = b * R;
      const int rows = (int)(B - r0 < R ? B - r0 : R);
      if (rb < rows)
      {
#pragma unroll
         for (int u = 0; u < NU; u++)
            if (jl + (256 / LPC) * u < n)
               reg[u] = ld<NTL>(src + (long)(jl + (256 / LPC) * u) * B + r0 + rb);
      }
   };
   long b = blockIdx.x;
   if (b < blocks) request(b);
   for (; b < blocks; b += gridDim.x)
   {
      const long r0 = b * R;
      const int rows = (int)(B - r0 < R ? B - r0 : R), len = rows * n;
      if (rb < rows)
      {
#pragma unroll
         for (int u = 0; u < NU; u++)
            if (jl + (256 / LPC) * u < n)
            {
#pragma unroll
               for (int k = 0; k < V; k++) blk[(rb + k) * n + jl + (256 / LPC) * u] = reg[u][k];
            }
      }
      lds_only_barrier();
      if (b + gridDim.x < blocks) request(b + gridDim.x);
      float *const flat = dst + r0 * n;
      for (int i = threadIdx.x * V; i + V <= len; i += 256 * V)
         st<NTS>(flat + i, *(const VT *)(blk + i));
      lds_only_barrier();
   }
}
template <bool NT>
__global__ void __launch_bounds__(256) copy4(const float *__restrict__ src, float *__restrict__ dst, long nvec)
{
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256)
      st<NT>(dst + 4 * i, ld<NT>(src + 4 * i));
}
// the round-4 form: one block per workgroup
__global__ void __launch_bounds__(256) r2c_old(const float *__restrict__ src, float *__restrict__ dst, long B, int n)
{
   extern __shared__ double lds_raw[];
   float *const blk = (float *)lds_raw;
   const long r0 = (long)blockIdx.x * R;
   const int rows = (int)(B - r0 < R ? B - r0 : R), len = rows * n;
   const float *const flat = src + r0 * n;
   for (int i = threadIdx.x * V; i + V <= len; i += 256 * V)
      *(VT *)(blk + i) = *(const VT *)(flat + i);
   __syncthreads();
   const int rb = (threadIdx.x % LPC) * V, jl = threadIdx.x / LPC;
   if (rb < rows)
      for (int j = jl; j < n; j += 256 / LPC)
      {
         VT w;
#pragma unroll
         for (int k = 0; k < V; k++) w[k] = blk[(rb + k) * n + j];
         *(VT *)(dst + (long)j * B + r0 + rb) = w;
      }
}
template <class F>
static double time_us(F f, int iters = 20)
{
   hipEvent_t a, b;
   CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
   for (int i = 0; i < 3; i++) f();
   std::vector<float> ts;
   for (int i = 0; i < iters; i++)
   {
      CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
   }
   std::sort(ts.begin(), ts.end());
   return ts[ts.size() / 2] * 1e3;
}
int main(int argc, char **argv)
{
   const long B = argc > 1 ? atol(argv[1]) : 131072;
   const int n = argc > 2 ? atoi(argv[2]) : 323;
   const long N = B * n, blocks = (B + R - 1) / R;
   const size_t lds = 128 * (size_t)n;
   float *a, *c, *d;
   CHECK(hipMalloc(&a, N * 4)); CHECK(hipMalloc(&c, N * 4)); CHECK(hipMalloc(&d, N * 4));
   std::vector<float> h(N);
   for (long i = 0; i < N; i++) h[i] = (float)(i % 1000003);
   CHECK(hipMemcpy(a, h.data(), N * 4, hipMemcpyHostToDevice));
   const double bytes = 2.0 * N * 4;
   auto report = [&](const char *name, double us) { printf("B %ld n %d  %-44s %8.1f us  %5.2f TB/s\n", B, n, name, us, bytes / us * 1e-6); fflush(stdout); };
   for (int g : {1024, 2048, 4096, 8192})
   {
      char nm[64];
      snprintf(nm, 64, "float4 copy, %d workgroups", g); report(nm, time_us([&] { hipLaunchKernelGGL(copy4<false>, dim3(g), dim3(256), 0, 0, a, c, N / 4); }));
      snprintf(nm, 64, "float4 copy nt, %d workgroups", g); report(nm, time_us([&] { hipLaunchKernelGGL(copy4<true>, dim3(g), dim3(256), 0, 0, a, c, N / 4); }));
   }
   report("rows->columns, one block per workgroup (r4)", time_us([&] { hipLaunchKernelGGL(r2c_old, dim3(blocks), dim3(256), lds, 0, a, c, B, n); }));
   auto check = [&](const char *what) {
      std::vector<float> r(N);
      CHECK(hipMemcpy(r.data(), c, N * 4, hipMemcpyDeviceToHost));
      long bad = 0;
      for (long i = 0; i < B && bad == 0; i += 977) for (int j = 0; j < n; j++) if (r[(long)j * B + i] != h[i * n + j]) { bad++; break; }
      if (bad) printf("  MISMATCH %s\n", what);
   };
   check("r4");
   for (int wpc : {1, 2, 3})
   {
      if ((size_t)wpc * lds > 160 * 1024) continue;
      const unsigned g = (unsigned)std::min<long>(blocks, 256L * wpc);
      char nm[96];
      CHECK(hipMemset(c, 0, N * 4));
      snprintf(nm, 96, "rows->columns strided  %d/CU", wpc); report(nm, time_us([&] { hipLaunchKernelGGL((r2c<0, false, false>), dim3(g), dim3(256), lds, 0, a, c, B, n, blocks); })); check(nm);
      CHECK(hipMemset(c, 0, N * 4));
      snprintf(nm, 96, "rows->columns chunked  %d/CU", wpc); report(nm, time_us([&] { hipLaunchKernelGGL((r2c<1, false, false>), dim3(g), dim3(256), lds, 0, a, c, B, n, blocks); })); check(nm);
      snprintf(nm, 96, "rows->columns strided nt-store %d/CU", wpc); report(nm, time_us([&] { hipLaunchKernelGGL((r2c<0, false, true>), dim3(g), dim3(256), lds, 0, a, c, B, n, blocks); })); check(nm);
      snprintf(nm, 96, "rows->columns strided nt-load+store %d/CU", wpc); report(nm, time_us([&] { hipLaunchKernelGGL((r2c<0, true, true>), dim3(g), dim3(256), lds, 0, a, c, B, n, blocks); })); check(nm);
      snprintf(nm, 96, "columns->rows strided  %d/CU", wpc); report(nm, time_us([&] { hipLaunchKernelGGL((c2r<false, false>), dim3(g), dim3(256), lds, 0, c, d, B, n, blocks); }));
      snprintf(nm, 96, "columns->rows strided nt-store %d/CU", wpc); report(nm, time_us([&] { hipLaunchKernelGGL((c2r<false, true>), dim3(g), dim3(256), lds, 0, c, d, B, n, blocks); }));
      snprintf(nm, 96, "columns->rows strided nt %d/CU", wpc); report(nm, time_us([&] { hipLaunchKernelGGL((c2r<true, true>), dim3(g), dim3(256), lds, 0, c, d, B, n, blocks); }));
      std::vector<float> r(N);
      CHECK(hipMemcpy(r.data(), d, N * 4, hipMemcpyDeviceToHost));
      if (r != h) printf("  MISMATCH columns->rows\n");
   }
   return 0;
}
