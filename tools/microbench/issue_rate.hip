// Issue-rate microbenchmark for gfx950 (diagnostic; not part of the product): cycles per instruction of one wave's
// stream of independent v_fma_f32 / v_pk_fma_f32 / ds_read_b32 / s_add, with 1, 2, 3 or 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -o issue_rate issue_rate.hip ; run: ./issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define REP 64
template <int MODE>
__global__ void k(unsigned long long* out, int iters) {
  __shared__ float lds[4096];
  float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  f2 c = {1.0001f, 0.9999f};
  lds[threadIdx.x] = a0;
  __syncthreads();
  int addr = (threadIdx.x & 63) * 4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < REP / 8; r++)
        asm volatile(
            "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
            "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(c.x));
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < REP / 8; r++)
        asm volatile(
            "v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
            "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
            : "v"(c));
    } else if (MODE == 2) {  // independent LDS reads, drained once per 8
#pragma unroll
      for (int r = 0; r < REP / 8; r++)
        asm volatile(
            "ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
            "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n"
            "s_waitcnt lgkmcnt(0)\n"
            : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
            : "v"(addr));
    } else if (MODE == 3) {  // dependent LDS read chain (latency)
#pragma unroll
      for (int r = 0; r < REP; r++) {
        int x;
        asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n" : "=v"(x) : "v"(addr));
        addr = (addr + (x & 0)) & 0x3fff;
      }
    } else if (MODE == 4) {  // scalar ALU
      int s0 = it, s1 = 1;
#pragma unroll
      for (int r = 0; r < REP; r++) asm volatile("s_add_i32 %0, %0, %1\n" : "+s"(s0) : "s"(s1));
      a0 += s0;
    } else if (MODE == 5) {  // v_fma with DPP-like dependent chain: dependent v_fma chain (latency)
#pragma unroll
      for (int r = 0; r < REP; r++) asm volatile("v_fma_f32 %0, %0, %1, %0\n" : "+v"(a0) : "v"(c.x));
    } else if (MODE == 6) {  // v_add_f32 dpp row_shr
#pragma unroll
      for (int r = 0; r < REP / 8; r++)
        asm volatile(
            "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 7) {  // dependent DPP chain (a wave reduction's shape)
#pragma unroll
      for (int r = 0; r < REP; r++) asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n" : "+v"(a0));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x;
  if (sink == 12345.678f) out[1 << 20] = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE>
void run(const char* name, int waves_per_simd) {
  int threads = 256 * waves_per_simd, blocks = 256, iters = 2000;
  unsigned long long* d;
  size_t n = (size_t)blocks * threads / 64;
  hipMalloc(&d, ((1 << 20) + 8) * sizeof(unsigned long long));
  k<MODE><<<blocks, threads>>>(d, 10);
  hipDeviceSynchronize();
  k<MODE><<<blocks, threads>>>(d, iters);
  hipDeviceSynchronize();
  unsigned long long* h = (unsigned long long*)malloc(n * 8);
  hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (size_t i = 0; i < n; i++) s += h[i];
  printf("%-28s waves/SIMD=%d  cycles per instruction per wave = %.2f  (per SIMD: %.2f)\n", name, waves_per_simd,
         s / n / iters / REP, s / n / iters / REP / waves_per_simd);
  free(h);
  hipFree(d);
}
int main() {
  for (int w = 1; w <= 4; w++) {
    run<0>("v_fma_f32 independent", w);
    run<1>("v_pk_fma_f32 independent", w);
    run<2>("ds_read_b32 x8 + wait", w);
    run<3>("ds_read_b32 dependent", w);
    run<4>("s_add_i32", w);
    run<5>("v_fma_f32 dependent", w);
    run<6>("v_add_f32_dpp independent", w);
    run<7>("v_add_f32_dpp dependent+nop", w);
  }
  return 0;
}
