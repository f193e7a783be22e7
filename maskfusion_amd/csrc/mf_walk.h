// mf_walk.h -- walking a surfel buffer on the device: by the runs of its table (Surfels::box, FrameDev::runs) or, a dense buffer without one,
// slot by slot.  (Kernel code: uses threadIdx / blockIdx -- kept out of mf_device.h, whose math the host-compiled tests include.)
#pragma once

#include "mf_device.h"

namespace mf {

// The live surfels of a buffer in wavefront-uniform slices of 256 slots: f(slot, live) is called by every thread of the workgroup (the index
// scatter exchanges keys between the lanes of a wavefront).  vis_list != nullptr: only the runs k_cull listed (Surfels::box) -- the others hold
// no surfel that could pass the caller's tests, so the keys are the same bits either way; else every run of the table (frame->runs > 0), or
// -- a dense buffer without a table -- slots [0, count).  kLanes neighbouring lanes share one surfel (the sprite passes of the object models).
template <int kLanes, class F>
__device__ __forceinline__ void for_each_surfel_slice(const Surfels& src, const FrameDev* __restrict__ frame, const int* __restrict__ vis_list,
                                                      const int* __restrict__ vis_count, F&& f) {
    constexpr int kPer = 256 / kLanes;
    const int sub = (int)threadIdx.x / kLanes;
    const int runs = frame->runs;
    if (vis_list || runs > 0) {
        const int nv = vis_list ? *vis_count : runs;
        for (int v = blockIdx.x; v < nv; v += gridDim.x) {
            const int r = vis_list ? vis_list[v] : v;
            const int beg = run_start(src.box, r), end = beg + run_len(src.box, r);
            for (int i0 = beg; i0 < end; i0 += kPer) f(i0 + sub, i0 + sub < end);     // (wavefront-uniform bounds)
        }
        return;
    }
    const int n = frame->count;
    for (int i0 = blockIdx.x * kPer; i0 < n; i0 += gridDim.x * kPer) f(i0 + sub, i0 + sub < n);
}


}  // namespace mf
