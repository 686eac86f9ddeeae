// micro-benchmark: what one dependent fp32 add costs a lone wave (the floor of the verification chains of
// k_cluster_verify_b), and whether the LDS feed or the add itself sets it.
//   V0 registers only, 64 lanes      V1 registers only, 6 lanes        V2 registers only, 1 lane
//   V3 LDS "read 16, add 16"         V4 LDS, reads a block of 32 ahead  V5 two independent chains (issue rate)
//   V6 V0 at s_setprio 3
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float chain16(float acc, float a, float b, float c, float d)
{
  acc += a; acc += b; acc += c; acc += d; acc += a; acc += b; acc += c; acc += d;
  acc += a; acc += b; acc += c; acc += d; acc += a; acc += b; acc += c; acc += d;
  return acc;
}

template <int V>
__global__ void k_chain(const float* in, float* out, int n, long long* cyc)
{
  __shared__ __attribute__((aligned(16))) float s[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = in[i];
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int lanes = V == 1 ? 6 : (V == 2 ? 1 : 64);
  if ((int)threadIdx.x >= lanes) return;
  if (V == 6) __builtin_amdgcn_s_setprio(3);
  float acc = 0.f, acc2 = 0.f;
  const float a = in[threadIdx.x], b = in[threadIdx.x + 64], c = in[threadIdx.x + 128], d = in[threadIdx.x + 192];
  const long long t0 = __builtin_amdgcn_s_memtime();
  const float* p = s + (threadIdx.x & 3) * 4;
  for (int r = 0; r < n; ++r) {
    if (V <= 2 || V == 6) {
#pragma unroll 1
      for (int j = 0; j < 4096 - 64; j += 16) {
        acc = chain16(acc, a, b, c, d);
        asm volatile("" : "+v"(acc));
      }
    } else if (V == 5) {
#pragma unroll 1
      for (int j = 0; j < 4096 - 64; j += 32) {
        acc = chain16(acc, a, b, c, d);
        acc2 = chain16(acc2, b, c, d, a);
        asm volatile("" : "+v"(acc), "+v"(acc2));
      }
    } else if (V == 3) {
#pragma unroll 1
      for (int j = 0; j < 4096 - 64; j += 16) {
        const float4 q0 = *reinterpret_cast<const float4*>(p + j), q1 = *reinterpret_cast<const float4*>(p + j + 4);
        const float4 q2 = *reinterpret_cast<const float4*>(p + j + 8), q3 = *reinterpret_cast<const float4*>(p + j + 12);
        acc += q0.x; acc += q0.y; acc += q0.z; acc += q0.w; acc += q1.x; acc += q1.y; acc += q1.z; acc += q1.w;
        acc += q2.x; acc += q2.y; acc += q2.z; acc += q2.w; acc += q3.x; acc += q3.y; acc += q3.z; acc += q3.w;
      }
    } else {  // V4: two register sets of 32 values, the next block in flight while this one is added
      float4 ra[8], rb[8];
      auto load = [&](const float* q, float4 (&r)[8]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = *reinterpret_cast<const float4*>(q + 4 * k);
        asm volatile("" : "+v"(acc) : : "memory");
      };
      auto add = [&](const float4 (&r)[8]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc += r[k].x; acc += r[k].y; acc += r[k].z; acc += r[k].w; }
      };
      load(p, ra);
#pragma unroll 1
      for (int j = 0; j + 64 <= 4096 - 64; j += 64) {
        load(p + j + 32, rb);
        add(ra);
        load(p + j + 64, ra);
        add(rb);
      }
      add(ra);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = acc + acc2;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

template <int V>
void run(const char* name, const float* in, float* out, long long* cyc)
{
  const int n = 200;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_chain<V>, dim3(1), dim3(256), 0, 0, in, out, n, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double adds = (double)n * (4096 - 64);
    if (rep == 1) printf("%-44s ns/add %.2f  s_memtime ticks/add %.2f\n", name, ms * 1e6 / adds, (double)h / adds);
  }
}

int main()
{
  float *in, *out; long long* cyc;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 4); hipMalloc(&cyc, 8);
  hipMemset(in, 0, 4096 * 4);
  run<0>("V0 registers, 64 lanes", in, out, cyc);
  run<1>("V1 registers, 6 lanes", in, out, cyc);
  run<2>("V2 registers, 1 lane", in, out, cyc);
  run<3>("V3 LDS: read 16, add 16", in, out, cyc);
  run<4>("V4 LDS: reads one block of 32 ahead", in, out, cyc);
  run<5>("V5 two independent chains (per add of one)", in, out, cyc);
  run<6>("V6 registers, 64 lanes, s_setprio 3", in, out, cyc);
  return 0;
}
