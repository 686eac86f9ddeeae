// micro-benchmark: cycles per dependent fp32 add of one wave (the floor of the verification chains)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_chain(const float* in, float* out, int n, long long* cyc)
{
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = in[i];
  __syncthreads();
  if (threadIdx.x >= 64) return;
  float acc = 0.f;
  const long long t0 = clock64();
  for (int r = 0; r < n; ++r) {
    const float* p = s + (threadIdx.x & 3) * 4;
#pragma unroll 1
    for (int j = 0; j < 4096 - 64; j += 16) {
      const float4 a = *reinterpret_cast<const float4*>(p + j), b = *reinterpret_cast<const float4*>(p + j + 4);
      const float4 c = *reinterpret_cast<const float4*>(p + j + 8), d = *reinterpret_cast<const float4*>(p + j + 12);
      acc += a.x; acc += a.y; acc += a.z; acc += a.w; acc += b.x; acc += b.y; acc += b.z; acc += b.w;
      acc += c.x; acc += c.y; acc += c.z; acc += c.w; acc += d.x; acc += d.y; acc += d.z; acc += d.w;
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
  float *in, *out; long long* cyc;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 4); hipMalloc(&cyc, 8);
  hipMemset(in, 0, 4096 * 4);
  const int n = 200;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), 0, 0, in, out, n, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double adds = (double)n * (4096 - 64);
    printf("adds %.0f  ms %.3f  ns/add %.2f  clock64 ticks/add %.2f\n", adds, ms, ms * 1e6 / adds, (double)h / adds);
  }
  return 0;
}
