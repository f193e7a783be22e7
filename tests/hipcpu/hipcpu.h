// hipcpu.h -- TEST TOOLING, never part of the product: a host stand-in for the HIP device language and runtime, so that the product's
// own kernels (maskfusion_amd/csrc/*.hip, unmodified but for three mechanical in-memory edits listed in tests/hipcpu/build.py) can be
// compiled with g++ and EXECUTED ON THE CPU by the logic tests when no GPU is at hand (this container has none; GPU minutes are
// rationed).  It answers "is the kernel logic right" -- indexing, barriers, wave exchanges, atomics, control flow -- not "is it fast",
// and not the device's own float rounding (v_rcp / v_exp / contraction).  The product never loads the library built from this; the
// parity claims rest on the -m gpu runs on MI355X.
//
// Execution model (hipcpu_runtime.cpp): a launch runs its workgroups one after the other; the threads of a workgroup are cooperative
// fibers; wavefronts are 64 consecutive threads.  __syncthreads, __shfl*, __ballot, DPP (quad_perm, row_ror), LDS, atomics and the
// few amdgcn builtins the kernels use have their device meaning.  Streams and events are trivial: every call completes before it returns.
#ifndef HIPCPU_H_
#define HIPCPU_H_
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

#include <hip/hip_runtime.h>   // a host compiler's view: vector types, dim3, the runtime API prototypes (defined in hipcpu_runtime.cpp)

#undef __shared__
// Two builds (tests/hipcpu/build.py): the default one runs the workgroups of a launch one after the other on the calling thread -- LDS
// is one static copy per kernel; the HIPCPU_COOP build can also run ALL workgroups of a launch at once, one OS thread each (what a
// kernel with a device-wide barrier needs), so LDS and the built-in indices are thread_local there (about 2.5x slower overall).
#ifdef HIPCPU_COOP
#define HIPCPU_TLS thread_local
#else
#define HIPCPU_TLS
#endif
#define __shared__ alignas(64) static HIPCPU_TLS   // LDS allocations are at least 16-byte aligned on the device (float4 accesses)
#undef __launch_bounds__
#define __launch_bounds__(...)
#undef __noinline__
#define __noinline__ __attribute__((noinline))

extern HIPCPU_TLS uint3 threadIdx, blockIdx;
extern HIPCPU_TLS dim3 blockDim, gridDim;
static const int warpSize = 64;

namespace hipcpu {
// name: the kernel as written at the launch site.  Kernels listed in the environment variable HIPCPU_COOPERATIVE (comma separated
// prefixes) are launched with ALL workgroups resident at once, one OS thread per workgroup -- what a kernel with a device-wide barrier
// needs (HIPCPU_COOP build only); every other launch runs its workgroups one after the other on the calling thread.
void launch(const char* name, dim3 grid, dim3 block, size_t dyn_shared, const std::function<void()>& body);
void syncthreads();
void yield();
int lane();                                        // linear thread id % 64
bool lane_alive(int l);                            // lane l of the calling thread's wavefront exists and has not returned
const uint64_t* wave_publish(uint64_t v);          // every live lane of the wavefront publishes v; returns the wavefront's 64 slots
void* dyn_shared();
unsigned long long clock();
template <class T> inline uint64_t to_bits(T v) { static_assert(sizeof(T) <= 8, ""); uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
}  // namespace hipcpu

// kernel arguments are copied when the launch is issued (as on the device): a launch recorded by a stream capture is replayed later
namespace hipcpu {
template <class F, class... A>
inline void launch_args(const char* name, dim3 grid, dim3 block, size_t dyn_shared, F f, A... a) {
    launch(name, grid, block, dyn_shared, [=]() { f(a...); });
}
}  // namespace hipcpu
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipcpu::launch_args(#kernel, dim3(grid), dim3(block), (size_t)(shmem), [](auto... hipcpu_a) { kernel(hipcpu_a...); }, ##__VA_ARGS__)

// ---- synchronisation, wave exchanges ----------------------------------------------------------------------------------------
static inline void __syncthreads() { hipcpu::syncthreads(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __builtin_amdgcn_s_sleep(n) hipcpu::yield()
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)   // (a scheduling fence of the device compiler: no meaning here)
static inline unsigned long long __builtin_amdgcn_s_memtime() { return hipcpu::clock(); }
long long hipcpu_wall_clock();   // real time at 1 MHz: the device counter runs at 100 MHz, so a time-out written for the GPU is 100x longer here
static inline long long wall_clock64() { return hipcpu_wall_clock(); }

static inline unsigned long long __ballot(int pred) {
    const uint64_t* s = hipcpu::wave_publish(pred ? 1u : 0u);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (hipcpu::lane_alive(l) && s[l]) m |= 1ull << l;
    return m;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    const int me = hipcpu::lane();
    const uint64_t* s = hipcpu::wave_publish(hipcpu::to_bits(v));
    const int l = (me & ~(width - 1)) | (src & (width - 1));
    return hipcpu::lane_alive(l) ? hipcpu::from_bits<T>(s[l]) : v;
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    const int me = hipcpu::lane();
    const uint64_t* s = hipcpu::wave_publish(hipcpu::to_bits(v));
    const int l = me ^ mask;
    return ((l & ~(width - 1)) == (me & ~(width - 1)) && hipcpu::lane_alive(l)) ? hipcpu::from_bits<T>(s[l]) : v;
}
template <class T> static inline T __shfl_up(T v, unsigned delta, int width = 64) {
    const int me = hipcpu::lane();
    const uint64_t* s = hipcpu::wave_publish(hipcpu::to_bits(v));
    const int l = me - (int)delta;
    return ((me & (width - 1)) >= (int)delta && hipcpu::lane_alive(l)) ? hipcpu::from_bits<T>(s[l]) : v;
}
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    const int me = hipcpu::lane();
    const uint64_t* s = hipcpu::wave_publish(hipcpu::to_bits(v));
    const int l = me + (int)delta;
    return ((me & (width - 1)) + (int)delta < width && hipcpu::lane_alive(l)) ? hipcpu::from_bits<T>(s[l]) : v;
}
// DPP: the controls the kernels use -- quad_perm (0x00..0xFF) and row_ror:n (0x121..0x12F); full row / bank masks
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int me = hipcpu::lane();
    const uint64_t* s = hipcpu::wave_publish((uint32_t)src);
    int l;
    if (ctrl >= 0 && ctrl <= 0xFF) l = (me & ~3) | ((ctrl >> (2 * (me & 3))) & 3);
    else if (ctrl >= 0x121 && ctrl <= 0x12F) l = (me & ~15) | (((me & 15) - (ctrl - 0x120)) & 15);
    else { fprintf(stderr, "hipcpu: DPP control 0x%x not emulated\n", ctrl); abort(); }
    if (row_mask != 0xf || bank_mask != 0xf) { fprintf(stderr, "hipcpu: partial DPP masks not emulated\n"); abort(); }
    return hipcpu::lane_alive(l) ? (int)(uint32_t)s[l] : (bound_ctrl ? 0 : old);
}
// the halves of a double (CUDA / HIP device intrinsics)
static inline int __double2loint(double v) { uint64_t b; memcpy(&b, &v, 8); return (int)(uint32_t)(b & 0xFFFFFFFFu); }
static inline int __double2hiint(double v) { uint64_t b; memcpy(&b, &v, 8); return (int)(uint32_t)(b >> 32); }
static inline double __hiloint2double(int hi, int lo) { const uint64_t b = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double v; memcpy(&v, &b, 8); return v; }
// v_readfirstlane_b32 / v_readlane_b32: the value of the first active lane / of a given lane, uniform across the wavefront
static inline int __builtin_amdgcn_readfirstlane(int v) {
    const uint64_t* s = hipcpu::wave_publish((uint32_t)v);
    for (int l = 0; l < 64; ++l) if (hipcpu::lane_alive(l)) return (int)(uint32_t)s[l];
    return v;
}
static inline int __builtin_amdgcn_readlane(int v, int lane) {
    const uint64_t* s = hipcpu::wave_publish((uint32_t)v);
    return hipcpu::lane_alive(lane & 63) ? (int)(uint32_t)s[lane & 63] : 0;
}
static inline int __builtin_amdgcn_mbcnt_lo(unsigned mask, int base) {
    const int me = hipcpu::lane();
    return base + __builtin_popcount(mask & (me >= 32 ? 0xFFFFFFFFu : ((1u << me) - 1u)));
}
static inline int __builtin_amdgcn_mbcnt_hi(unsigned mask, int base) {
    const int me = hipcpu::lane();
    return base + (me > 32 ? __builtin_popcount(mask & ((1u << (me - 32)) - 1u)) : 0);
}

// ---- scalar builtins ----------------------------------------------------------------------------------------------------------
static inline float __int_as_float(int i) { return hipcpu::from_bits<float>((uint32_t)i); }
static inline float __uint_as_float(unsigned i) { return hipcpu::from_bits<float>(i); }
static inline int __float_as_int(float f) { return (int)(uint32_t)hipcpu::to_bits(f); }
static inline unsigned __float_as_uint(float f) { return (uint32_t)hipcpu::to_bits(f); }
static inline double __longlong_as_double(long long v) { return hipcpu::from_bits<double>((uint64_t)v); }
static inline long long __double_as_longlong(double v) { return (long long)hipcpu::to_bits(v); }
static inline int __float2int_rn(float f) { return (int)nearbyintf(f); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
static inline double __builtin_amdgcn_rsq(double x) { return 1.0 / sqrt(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
using std::max;
using std::min;
using std::isnan;
using std::isinf;
using std::isfinite;
static inline int min(int a, unsigned b) { return a < (int)b ? a : (int)b; }
static inline int min(unsigned a, int b) { return (int)a < b ? (int)a : b; }
static inline float fminf(float a, int b) { return ::fminf(a, (float)b); }

// ---- atomics (threads interleave only at barriers and wave exchanges: a plain read-modify-write is atomic here) -----------------
template <class T, class U> static inline T atomicAdd(T* p, U v) { const T o = *p; *p = o + (T)v; return o; }
template <class T, class U> static inline T atomicSub(T* p, U v) { const T o = *p; *p = o - (T)v; return o; }
template <class T, class U> static inline T atomicMin(T* p, U v) { const T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicMax(T* p, U v) { const T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicOr(T* p, U v) { const T o = *p; *p = o | (T)v; return o; }
template <class T, class U> static inline T atomicAnd(T* p, U v) { const T o = *p; *p = o & (T)v; return o; }
template <class T, class U> static inline T atomicExch(T* p, U v) { const T o = *p; *p = (T)v; return o; }
template <class T, class U> static inline T atomicCAS(T* p, U cmp, U v) { const T o = *p; if (o == (T)cmp) *p = (T)v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)     // real atomics: cooperative launches
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)

#endif  // HIPCPU_H_
