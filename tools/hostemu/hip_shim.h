// hip_shim.h — DEBUGGING HARNESS, not a product path.
//
// Compiles raytracer_amd/csrc/rtcore.hip as plain host C++ so that the device code can be run under
// AddressSanitizer / UBSan / gdb on the CPU (GPU sanitizers and rocgdb are not available on the GPU pool).
// One "lane" runs at a time: a kernel launch is a loop over blocks and threads, wave votes see a single lane.
// The library built from this (build/librtcore_emu.so) is only ever loaded by a developer who points RTCORE_LIB
// at it; nothing in raytracer_amd/, tests/ or bench.py uses it.
#pragma once
#include <cfenv>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __noinline__
#define __shared__ static
#define __launch_bounds__(...)

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct double2 { double x, y; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static thread_local dim3 blockIdx, threadIdx, blockDim, gridDim;

static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline long long __double_as_longlong(double d) { long long u; std::memcpy(&u, &d, 8); return u; }
static inline float __double2float_rd(double d) {
  float f = (float)d;
  if ((double)f > d) f = std::nextafterf(f, -INFINITY);
  return f;
}
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool b) { return b ? 1ull : 0ull; }
static inline long long clock64() { return 0; }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
template <class T> static inline T __shfl_down(T, int, int) { return T(0); }
template <class T> static inline T __shfl(T v, int, int) { return v; }
#define __HIP_MEMORY_SCOPE_AGENT 0
template <class T> static inline T __hip_atomic_load(T* p, int, int) { return *p; }
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p += v; return o; }
static inline uint32_t atomicMax(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
static inline uint32_t atomicOr(uint32_t* p, uint32_t v) { uint32_t o = *p; *p |= v; return o; }
// (rt_mesh.h: its kernels cooperate across a block — barriers, LDS accumulators — which one lane at a time cannot emulate;
// these only let the file compile, and rt_mesh_create refuses under RT_HOST_EMU)
static inline void __syncthreads() {}
template <class T, class V> static inline T atomicMin(T* p, V v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T expect, T v) { T o = *p; if (o == expect) *p = v; return o; }
template <class T> static inline T __shfl_up(T v, int, int) { return v; }

typedef int hipError_t;
typedef void* hipStream_t;
struct hipEventRec { double t; };
typedef hipEventRec* hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
static inline const char* hipGetErrorString(hipError_t) { return "host emulation"; }
static inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }
static inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t, uint32_t v) { return v; }  // single lane: nothing below it
static inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t, uint32_t v) { return v; }
static inline uint32_t __builtin_amdgcn_readfirstlane(uint32_t v) { return v; }
static inline double __longlong_as_double(long long u) { double d; std::memcpy(&d, &u, 8); return d; }
enum { hipDeviceAttributeMultiprocessorCount = 0 };
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 1; return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)6 << 30; *t = (size_t)8 << 30; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::aligned_alloc(256, (n + 255) & ~(size_t)255); return hipSuccess; }  // (hipMalloc's alignment)
#define __align__(n) alignas(n)
typedef void (*hipHostFn_t)(void*);
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipLaunchHostFunc(hipStream_t, hipHostFn_t fn, void* data) { fn(data); return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
#define HIP_SYMBOL(x) (&(x))
template <class T> static inline hipError_t hipGetSymbolAddress(void** p, T* symbol) { *p = (void*)symbol; return hipSuccess; }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
enum { hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = std::malloc(n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyWithStream(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEventRec{0.}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipEventRec{0.}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

template <class K, class... A>
static inline void rt_emu_launch(K kernel, dim3 grid, dim3 block, A... args) {
  gridDim = grid;
  blockDim = block;
  for (unsigned b = 0; b < grid.x; b++)
    for (unsigned t = 0; t < block.x; t++) {
      blockIdx = dim3(b);
      threadIdx = dim3(t);
      kernel(args...);
    }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) rt_emu_launch(kernel, grid, block, __VA_ARGS__)
