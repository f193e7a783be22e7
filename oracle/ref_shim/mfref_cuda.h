/*
 * mfref_cuda.h -- host stand-in for the CUDA toolkit headers, so that the reference's own CUDA translation units
 * (Core/Cuda/reduce.cu, cudafuncs.cu, segmentation.cu, containers/device_memory.cpp of martinruenz/maskfusion) compile
 * with plain g++ and run on the CPU.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as oracle/mf_oracle.h).  Nothing of the product includes this.  The reference
 * sources are NOT copied into the repository: oracle/build_ref.py reads them where they lie under /root/reference,
 * rewrites only the `kernel<<<grid, block>>>(args)` launch syntax (which is not C++) into MFREF_LAUNCH(...) in memory,
 * and feeds the result to g++ on stdin; the only output is oracle/_ref/libmf_ref.so.
 *
 * Execution model: MFREF_LAUNCH runs the grid block after block; the threads of a block are cooperative fibers
 * (ucontext) scheduled round-robin on the calling OS thread, so __syncthreads() and __shfl_down() -- which the
 * reference's blockReduceSum / warpReduceSum use (reduce.cu:92-167) -- have their CUDA meaning with 32-wide warps.
 * Everything is deterministic.  Arithmetic: the build uses -ffp-contract=off (one rounding per operation) and IEEE
 * division / sqrt, where nvcc would contract a*b+c into FMAs and the reference's CMake passes --prec-div=false
 * --prec-sqrt=false --ftz=true; results therefore pin the reference's ALGORITHM (operation order, border rules, index
 * arithmetic, integer conversions), not the last bit of its device arithmetic.
 */
#ifndef MFREF_CUDA_H_
#define MFREF_CUDA_H_

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>
#include <cstdlib>
#include <functional>

/* the reference tests these: a "device pass of nvcc for sm_61" so that types.cuh skips Eigen and reduce.cu does not
 * define its own pre-sm_30 __shfl_down / pre-sm_35 __ldg fallbacks (reduce.cu:58-90) */
#ifndef __CUDACC__
#define __CUDACC__ 1
#endif
#ifndef __CUDA_ARCH__
#define __CUDA_ARCH__ 610
#endif

#define __host__
#define __device__
#define __global__
#define __shared__
#define __constant__
#define __forceinline__ inline
#define __inline__ inline
#define __restrict__

/* ---------------- vector types ---------------- */
struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float1 { float x; };
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct short2 { short x, y; };
struct uchar3 { unsigned char x, y, z; };
struct uchar4 { unsigned char x, y, z, w; };
static inline float1 make_float1(float x) { return float1{x}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline short2 make_short2(short x, short y) { return short2{x, y}; }
static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return uchar4{x, y, z, w}; }

/* ---------------- thread coordinates (set by the fiber scheduler before a fiber resumes) ---------------- */
extern uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
static const int warpSize = 32;

void mfref_launch(dim3 grid, dim3 block, const std::function<void()>& body);
void mfref_syncthreads();
unsigned mfref_shfl_down_bits(unsigned v, int offset, int width);
#define MFREF_LAUNCH(kernel, grid, block, ...) mfref_launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { mfref_syncthreads(); }
static inline float __shfl_down(float v, int offset, int width = 32) {
    unsigned b; memcpy(&b, &v, 4);
    b = mfref_shfl_down_bits(b, offset, width);
    memcpy(&v, &b, 4);
    return v;
}
static inline int __shfl_down(int v, int offset, int width = 32) { return (int)mfref_shfl_down_bits((unsigned)v, offset, width); }
template <typename T> static inline T __ldg(const T* p) { return *p; }

/* ---------------- device math ---------------- */
static inline int __float2int_rn(float x) { return (int)nearbyintf(x); }   /* round to nearest even */
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float fminf_(float a, float b) { return fminf(a, b); }
using std::abs;
using std::isnan;
using std::isfinite;

/* ---------------- runtime API (device memory is host memory) ---------------- */
typedef int cudaError;
typedef int cudaError_t;
#define cudaSuccess 0
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
typedef unsigned long long cudaSurfaceObject_t;
struct cudaArray { int width, height; const void* data; };   /* 2D array of 4-byte texels (only uchar4 is used) */
enum cudaTextureReadMode { cudaReadModeElementType };
template <typename T, int Dim, cudaTextureReadMode Mode> struct texture { const cudaArray* arr = nullptr; };

static inline const char* cudaGetErrorString(cudaError_t) { return "mfref host shim"; }
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline cudaError_t cudaMemGetInfo(size_t* f, size_t* t) { *f = *t = 0; return cudaSuccess; }
static inline cudaError_t cudaMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? cudaSuccess : 2; }
template <typename T> static inline cudaError_t cudaMalloc(T** p, size_t n) { return cudaMalloc((void**)p, n); }
static inline cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memmove(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaMallocPitch(void** p, size_t* pitch, size_t widthBytes, size_t rows) {
    *pitch = widthBytes;   /* dense rows: RGBReduction indexes corresImg.data[i] linearly (reduce.cu:551) */
    return cudaMalloc(p, widthBytes * rows);
}
static inline cudaError_t cudaMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t wb, size_t rows, cudaMemcpyKind) {
    for (size_t r = 0; r < rows; ++r) memmove((char*)d + r * dp, (const char*)s + r * sp, wb);
    return cudaSuccess;
}
template <typename S> static inline cudaError_t cudaMemcpyToSymbol(S& sym, const void* src, size_t n) { memcpy(&sym, src, n); return cudaSuccess; }
template <typename T, int D, cudaTextureReadMode M>
static inline cudaError_t cudaBindTextureToArray(texture<T, D, M>& t, const cudaArray* a) { t.arr = a; return cudaSuccess; }
template <typename T, int D, cudaTextureReadMode M>
static inline cudaError_t cudaUnbindTexture(texture<T, D, M>& t) { t.arr = nullptr; return cudaSuccess; }
/* surfaces: the error surfaces are debug outputs (RGBDOdometry passes 0 / a never-read texture); the entry points
 * here always pass 0, so a write is a bug in the harness */
template <typename T> static inline void surf2Dwrite(T, cudaSurfaceObject_t, int, int) { fprintf(stderr, "mfref: surf2Dwrite on a null surface\n"); abort(); }
/* unnormalised coordinates, point sampling, clamp addressing (the defaults of a texture reference) */
template <typename T, int D, cudaTextureReadMode M>
static inline T tex2D(const texture<T, D, M>& t, float x, float y) {
    int xi = (int)floorf(x), yi = (int)floorf(y);
    xi = xi < 0 ? 0 : (xi >= t.arr->width ? t.arr->width - 1 : xi);
    yi = yi < 0 ? 0 : (yi >= t.arr->height ? t.arr->height - 1 : yi);
    return ((const T*)t.arr->data)[(size_t)yi * t.arr->width + xi];
}

#endif /* MFREF_CUDA_H_ */
