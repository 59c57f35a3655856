// TEST-ONLY stand-in for <hip/hip_runtime.h>: lets g++ compile the product's
// csrc/vnl_lib.hip unchanged into a host library in which every "kernel launch"
// is a serial loop over (block, thread).  It exists so that the kernel LOGIC can be
// parity-tested against the oracle in the CPU-only container (`-m "not gpu"`),
// before a GPU is available.  It is never built into, nor reachable from, the
// product package: the product library is compiled by hipcc against the real HIP
// runtime and has no CPU path.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
static thread_local dim3 blockIdx, threadIdx, blockDim, gridDim;
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
static inline const char* hipGetErrorString(hipError_t) { return "hostsim"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
template <class K>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 1; return hipSuccess; }
// Fork-join primitives of csrc/vnl_body.h, host flavour: one pass per workgroup in which every
// parallel-for region simply runs over all its items, serial regions run once, and cross-lane
// reductions are the identity (the single pass has already accumulated every item).
#define __syncthreads()
#define VNL_FORKJOIN_DEFINED
#define VNL_HD inline
#define VNL_LANES 1
#define VNL_ROWS_PER_LANE 512
#define VNL_ROWS_SMALL 320
#define VNL_CHAIN_WIDTH 8
#define VNL_ROWSETS_1 64
#define VNL_POST_THREADS 1
#define VNL_ADAM_THREADS 1
#define VNL_HEAD_THREADS 1
#define VNL_FINISH_THREADS 1
#define VNL_FINISH_SUM(v) (void)(v)
#define VNL_HEAD_GROUP 1
#define VNL_GROUP_SUM(v) (void)(v)
#define __shared__
#define VNL_ROWSETS_2 128
#define VNL_FOR(i, n) for (int i = 0; i < (n); ++i)
#define VNL_SERIAL if (true)
#define VNL_SYNC()
#define VNL_SYNC_GLOBAL()
#define VNL_LDS_DECL(name) static thread_local vreal name[32768]
#define vnl_wave_sum(x) (x)
#define vnl_wave_sum16(x) (x)
#define vnl_wave_any(x) (x)
#define VNL_SCAN_ADD(x, run) (run += (x), x = run)
#define VNL_WAVE_ITEMS(n) (n)
#define VNL_PAD_ITEMS(n) (n)
#define VNL_SCAN_ADD_C(x, run, carry) (run += (x), x = run)
#define VNL_PERLANE(T, name) T name[64]
#define VNL_AT(name, j) name[j]
#define VNL_GETF(name, a) name[a]
#define VNL_GETI(name, a) name[a]
#define VNL_ROWGETI(name, q, l) name[q]
#define VNL_ROWGETF(expr, q, l) (expr)
#define VNL_WAVE_FENCE()
// segment sums of blk_apply: the device's shift-and-add steps on the 64 'lanes' of the array (ascending l: lane l + k still
// holds the previous step's value when lane l reads it, as in the simultaneous DPP step; shifts stay inside rows of 16)
#define VNL_SEG_SUM(part, dsc, steps)                                                      \
  do {                                                                                     \
    for (int k_ = 0; k_ < (steps); k_++)                                                   \
      for (int l_ = 0; l_ < 64; l_++)                                                      \
        if (((dsc)[l_] >> (28 + k_)) & 1u) (part)[l_] += ((l_ & 15) + (1 << k_) < 16) ? (part)[l_ + (1 << k_)] : 0; \
  } while (0)
#define VNL_COUNT(pred) ((pred) ? 1 : 0)
#define VNL_RANK(pred, run, rank) do { rank = run; if (pred) run++; } while (0)
#define VNL_LINE_HEADERS(out, base, stride) \
  do { for (int k_ = 0; k_ < VNL_FAC_LINES; k_++) out[k_] = (int)(base)[k_ * (stride)]; } while (0)
#define VNL_UNIFORM_I(x) (x)
#define vnl_recip(x) (vreal(1.) / (x))
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)            \
  do {                                                                         \
    dim3 g_ = (grid), b_ = (block);                                            \
    gridDim = g_, blockDim = b_;                                               \
    for (unsigned by_ = 0; by_ < g_.y; by_++)                                  \
      for (unsigned bx_ = 0; bx_ < g_.x; bx_++) {                              \
        blockIdx = dim3(bx_, by_), threadIdx = dim3(0);                        \
        kernel(__VA_ARGS__);                                                   \
      }                                                                        \
  } while (0)
