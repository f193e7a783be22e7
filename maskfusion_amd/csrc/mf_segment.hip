// mf_segment.hip -- device half of the multi-model coupling:
//   geometric edge map, threshold, closing, invert   <- Core/Cuda/segmentation.cu:98-177, 217-269, 277-354
//                                                       (MfSegmentation::performSegmentation, MfSegmentation.cpp:149-208)
//   model-ID z-buffer of all models                   <- GlobalProjection::project / downloadDirect
//                                                       (Core/Model/GlobalProjection.cpp:43-114; splat_models.vert,
//                                                        combo_splat_models.frag)
//   object-model pose bookkeeping                     <- Model::updateStaticPose / makeStatic (Core/Model/Model.h:263-264),
//                                                       the 0.2 m jump test of MaskFusion.cpp:268-272
#pragma clang fp contract(off)
#include "mf_device.h"
#include "mf_walk.h"

namespace mf {

// ------------------------------------------------------------------------------------------------
// geometric edge map (concavity + distance terms over the 8 neighbours)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edge_map(const float* __restrict__ vmap, const float* __restrict__ nmap,
                                                  float* __restrict__ out, int W, int H, float wD, float wC) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int P = W * H, i = y * W + x;
    if (x < 1 || x >= W - 1 || y < 1 || y >= H - 1) { out[i] = 1.0f; return; }
    const float3 v = f3(vmap[i], vmap[P + i], vmap[2 * P + i]);
    const float3 n = f3(nmap[i], nmap[P + i], nmap[2 * P + i]);
    // the reference's invalid vertices carry z = 0; ours carry NaN in every plane
    if (v.z <= 0.0f || isnan(v.z)) { out[i] = 1.0f; return; }
    float c = 0.0f, d = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ox = (k == 0 || k == 3 || k == 5) ? -1 : ((k == 1 || k == 6) ? 0 : 1);
        const int oy = k < 3 ? -1 : (k < 5 ? 0 : 1);
        const int j = (y + oy) * W + (x + ox);
        const float3 vn = f3(vmap[j], vmap[P + j], vmap[2 * P + j]);
        const float3 nn = f3(nmap[j], nmap[P + j], nmap[2 * P + j]);
        const float dd = dot3(vn - v, n);
        const float ct = (dd < 0) ? 0.f : 1.f - dot3(nn, n);  // getConcavityTerm, segmentation.cu:106-112
        c = fmaxf(ct, c);                                      // fmax drops NaN operands, as in the reference
        d = fmaxf(fabsf(dd), d);                               // getDistanceTerm, :115-119
    }
    c = fmaxf(c, 0.0f) * wC;
    d *= wD;
    out[i] = fminf(1.0f, (c > d) ? c : d);
}

void launch_edge_map(const float* vmap, const float* nmap, float* out, int W, int H, float wD, float wC, hipStream_t s) {
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    hipLaunchKernelGGL(k_edge_map, grid, dim3(256), 0, s, vmap, nmap, out, W, H, wD, wC);
}

// threshold (segmentation.cu:257-262), optionally fused with the final invert (:264-269) when no closing runs
__global__ void k_threshold(const float* __restrict__ in, uint8_t* __restrict__ out, int n, float threshold, int invert) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint8_t v = in[i] > threshold ? 255 : 0;
        out[i] = invert ? (uint8_t)(255 - v) : v;
    }
}
__global__ void k_invert(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = (uint8_t)(255 - in[i]);
}
// dilate_Kernel / erode_Kernel (segmentation.cu:217-255): square window, centre excluded
template <bool kDilate>
__global__ __launch_bounds__(256) void k_morph(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int W, int H, int radius) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    uint8_t r = kDilate ? 0 : 255;
    for (int cy = max(y - radius, 0); cy <= min(y + radius, H - 1); ++cy)
        for (int cx = max(x - radius, 0); cx <= min(x + radius, W - 1); ++cx) {
            if (cy == y && cx == x) continue;
            const uint8_t v = in[cy * W + cx];
            if (kDilate ? v == 255 : v == 0) r = kDilate ? 255 : 0;
        }
    out[y * W + x] = r;
}

// thresholdMap -> morphGeometricSegmentationMap(radius, iterations) -> invertMap; result in `out`, `tmp` is scratch
void launch_edge_binary(const float* edge, uint8_t* out, uint8_t* tmp, int W, int H, float threshold, int radius, int iterations,
                        hipStream_t s) {
    const int n = W * H;
    const int blocks = min((n + 255) / 256, 2048);
    if (iterations <= 0) {
        hipLaunchKernelGGL(k_threshold, dim3(blocks), dim3(256), 0, s, edge, out, n, threshold, 1);
        return;
    }
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    hipLaunchKernelGGL(k_threshold, dim3(blocks), dim3(256), 0, s, edge, out, n, threshold, 0);
    for (int i = 0; i < iterations; ++i) {
        hipLaunchKernelGGL(k_morph<true>, grid, dim3(256), 0, s, out, tmp, W, H, radius);
        hipLaunchKernelGGL(k_morph<false>, grid, dim3(256), 0, s, tmp, out, W, H, radius);
    }
    hipLaunchKernelGGL(k_invert, dim3(blocks), dim3(256), 0, s, out, tmp, n);
    (void)hipMemcpyAsync(out, tmp, n, hipMemcpyDeviceToDevice, s);
}

// ------------------------------------------------------------------------------------------------
// GlobalProjection: the splat scatter of mf_surfel.hip with (model order, model id) as payload
// key = z bits << 32 | order << 8 | id  -> LESS on z, earlier model in the list wins ties (GL draw order)
// ------------------------------------------------------------------------------------------------
// kLanes neighbouring lanes share one surfel and split its sprite's pixels between them as a kLX x kLY block (1: the thread-per-surfel form).
// The object models' launches use 4: a few thousand sprites of 4-10 px a side kept a handful of threads busy for ~36 us each (round 4 trace),
// the rest of the GPU idle; the keys are the same bits in any split (atomicMin is order independent).
template <int kLanes>
__device__ __forceinline__ void global_scatter_body(Surfels src, const FrameDev* __restrict__ frame,
                                                    const PoseDev* __restrict__ pose, int W, int H, Intr k, float maxDepth,
                                                    float confThreshold, int timeDelta, unsigned payload,
                                                    unsigned long long* __restrict__ keys, bool pretest = false) {
    if (pose->alive == 0) return;  // model dropped by the jump test earlier in this frame
    const float time = (float)frame->tick;
    float Ri[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Ri[q] = pose->Ri[q];
    const float3 ti = f3(pose->ti[0], pose->ti[1], pose->ti[2]);
    constexpr int kLX = kLanes >= 2 ? 2 : 1, kLY = kLanes / kLX;
    const int sub = threadIdx.x % kLanes, sx = sub % kLX, sy = sub / kLX;
    for_each_surfel_slice<kLanes>(src, frame, nullptr, nullptr, [&](int i, bool live) {
        if (!live) return;
        const float4 pc = src.pc[i];
        if (pc.w < confThreshold) return;
        const float lastTime = src.ct[i].w;
        const float3 h = mul33(Ri, f3(pc.x, pc.y, pc.z)) + ti;
        if (h.z > maxDepth || h.z < 0 || time - lastTime > (float)timeDelta || lastTime > time) return;  // splat_models.vert:57
        const float u = ((k.fx * h.x) / h.z) + k.cx, v = ((k.fy * h.y) / h.z) + k.cy;
        if (!(u >= 0.f && u <= (float)W && v >= 0.f && v <= (float)H)) return;
        const float4 n4 = src.nr[i];
        const float3 nrm = normalize_gl(mul33(Ri, f3(n4.x, n4.y, n4.z)));
        const float rad = n4.w;
        const float3 x1 = normalize_gl(f3(nrm.y - nrm.z, -nrm.x, nrm.x)) * (rad * 1.41421356f);
        const float3 y1 = cross3(nrm, x1);
        float xs0 = INFINITY, xs1 = -INFINITY, ys0 = INFINITY, ys1 = -INFINITY;
        const float3 corners[4] = {h + x1, h + y1, h - y1, h - x1};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float pxq = ((k.fx * corners[q].x) / corners[q].z) + k.cx;
            const float pyq = ((k.fy * corners[q].y) / corners[q].z) + k.cy;
            xs0 = fminf(xs0, pxq); xs1 = fmaxf(xs1, pxq);
            ys0 = fminf(ys0, pyq); ys1 = fmaxf(ys1, pyq);
        }
        float size = fmaxf(0.f, fmaxf(fabsf(xs1 - xs0), fabsf(ys1 - ys0)));
        if (!(size > 0.f)) return;
        size = fminf(fmaxf(size, 1.0f), 64.0f);   // GL clamps gl_PointSize to the point size range: at least 1 px (see mf_splat.hip)
        const float half = size * 0.5f;
        const int px0 = max(0, (int)ceilf(u - half - 0.5f)), px1 = min(W - 1, (int)ceilf(u + half - 0.5f) - 1);
        const int py0 = max(0, (int)ceilf(v - half - 0.5f)), py1 = min(H - 1, (int)ceilf(v + half - 0.5f) - 1);
        const float sqrRad = rad * rad;
        const float pn = dot3(h, nrm);
        for (int py = py0 + sy; py <= py1; py += kLY)
            for (int px = px0 + sx; px <= px1; px += kLX) {
                const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
                const float3 l = normalize_gl(f3((fcx - k.cx) / k.fx, (fcy - k.cy) / k.fy, 1.0f));
                const float3 cp = l * (pn / dot3(l, nrm));
                const float3 diff = cp - h;
                if (!(dot3(diff, diff) <= sqrRad)) continue;
                if (!(cp.z > 0.f)) continue;
                if (pretest) zmin_key_pretested(&keys[py * W + px], ((unsigned long long)__float_as_uint(cp.z) << 32) | payload);   // (object models)
                else zmin_key(&keys[py * W + px], ((unsigned long long)__float_as_uint(cp.z) << 32) | payload);
            }
    });
}

__global__ __launch_bounds__(256) void k_global_scatter(Surfels src, const FrameDev* __restrict__ frame,
                                                        const PoseDev* __restrict__ pose, int W, int H, Intr k, float maxDepth,
                                                        float confThreshold, int timeDelta, unsigned payload,
                                                        unsigned long long* __restrict__ keys) {
    global_scatter_body<1>(src, frame, pose, W, H, k, maxDepth, confThreshold, timeDelta, payload, keys);
}
// every object model of the list in one launch (grid.z = model; ObjBatch, mf_internal.h): all of them z-test into the one key image
__global__ __launch_bounds__(256) void k_obj_global_scatter(const ObjBatch b) {
    const ObjPassArgs& m = b.m[blockIdx.z];
    if (b.denseSprites) global_scatter_body<1>(m.a, m.frame, m.pose, b.W, b.H, b.k, b.globalMaxDepth, 12.0f, b.timeDelta, m.global_payload, b.global_keys, true);
    else global_scatter_body<4>(m.a, m.frame, m.pose, b.W, b.H, b.k, b.globalMaxDepth, 12.0f, b.timeDelta, m.global_payload, b.global_keys, true);
}
void launch_obj_global_scatter(const ObjBatch& b, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_obj_global_scatter, dim3(blocks, 1, b.n), dim3(256), 0, s, b);
}

void launch_global_scatter(Surfels src, const FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, float maxDepth,
                           float confThreshold, int timeDelta, int order, int id, unsigned long long* keys, hipStream_t s, int blocks) {
    const unsigned payload = ((unsigned)order << 8) | ((unsigned)id & 255u);
    hipLaunchKernelGGL(k_global_scatter, dim3(blocks), dim3(256), 0, s, src, frame, pose, W, H, k, maxDepth, confThreshold, timeDelta,
                       payload, keys);
}

__global__ void k_global_resolve(unsigned long long* __restrict__ keys, uint8_t* __restrict__ ids, int P) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const unsigned long long key = keys[p];
        keys[p] = kEmptyKey;
        ids[p] = key == kEmptyKey ? 0 : (uint8_t)(key & 0xFFull);
    }
}
void launch_global_resolve(unsigned long long* keys, uint8_t* ids, int P, hipStream_t s) {
    hipLaunchKernelGGL(k_global_resolve, dim3(min((P + 255) / 256, 2048)), dim3(256), 0, s, keys, ids, P);
}

// ------------------------------------------------------------------------------------------------
// object-model pose bookkeeping (single thread each)
// ------------------------------------------------------------------------------------------------
// spawn: Model::Model (pose = I) + makeStatic(globalPose): initialC2Winv = pose * globalPose^-1 (MaskFusion.cpp:671-684)
__global__ void k_spawn_pose(PoseDev* obj, const PoseDev* bg, FrameDev* objFrame, const FrameDev* bgFrame, PoseDev* host_mirror) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    PoseDev q;
    memset(&q, 0, sizeof(q));
    for (int k = 0; k < 9; ++k) q.R[k] = q.Ri[k] = q.lastR[k] = (k % 4 == 0) ? 1.f : 0.f;
    q.fusionWeight = 1.f;
    q.alive = 1;
    q.weightLiteral = bg->weightLiteral;   // a context-wide switch: the new model inherits it from the background
    m33_inverse_f(bg->R, q.initR);
    const float3 v = mul33(q.initR, f3(bg->t[0], bg->t[1], bg->t[2]));
    q.initT[0] = -v.x; q.initT[1] = -v.y; q.initT[2] = -v.z;
    *obj = q;
    objFrame->tick = bgFrame->tick; objFrame->count = 0; objFrame->countNext = 0; objFrame->phys = 0; objFrame->runs = 0; objFrame->first = 0; objFrame->first_run = 0; objFrame->cover = 0; objFrame->useFillIn = 0; objFrame->done_cover = 0ull;
    MF_FRAME_BBOX_RESET(objFrame);
    objFrame->pad[0] = objFrame->pad[1] = objFrame->pad[2] = 0;
    if (host_mirror) *host_mirror = q;
}
void launch_spawn_pose(PoseDev* obj, const PoseDev* bg, FrameDev* objFrame, const FrameDev* bgFrame, PoseDev* host_mirror, hipStream_t s) {
    hipLaunchKernelGGL(k_spawn_pose, dim3(1), dim3(64), 0, s, obj, bg, objFrame, bgFrame, host_mirror);
}

// Model::updateStaticPose (Model.h:263): overridePose(initialC2Winv * globalPose)
__global__ void k_static_pose(PoseDev* obj, const PoseDev* bg, PoseDev* host_mirror) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    PoseDev p = *obj;
    for (int k = 0; k < 9; ++k) p.lastR[k] = p.R[k];
    for (int k = 0; k < 3; ++k) p.lastT[k] = p.t[k];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            p.R[r * 3 + c] = p.initR[r * 3] * bg->R[c] + p.initR[r * 3 + 1] * bg->R[3 + c] + p.initR[r * 3 + 2] * bg->R[6 + c];
    const float3 t = mul33(p.initR, f3(bg->t[0], bg->t[1], bg->t[2]));
    p.t[0] = t.x + p.initT[0]; p.t[1] = t.y + p.initT[1]; p.t[2] = t.z + p.initT[2];
    pose_derive(p);
    *obj = p;
    if (host_mirror) *host_mirror = p;
}
void launch_static_pose(PoseDev* obj, const PoseDev* bg, PoseDev* host_mirror, hipStream_t s) {
    hipLaunchKernelGGL(k_static_pose, dim3(1), dim3(64), 0, s, obj, bg, host_mirror);
}

// Model::overridePose (Core/Model/Model.h:235-238): lastPose = pose; pose = in  (mode 0: inPose replaces tracking,
// MaskFusion.cpp:413-415) or pose = pose * in (mode 1: bootstrap, :280-283).  in: row-major R, t.
struct PoseArg { float R[9]; float t[3]; };
__global__ void k_override_pose(PoseDev* pose, PoseArg in, int mode, PoseDev* host_mirror) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    PoseDev p = *pose;
    for (int k = 0; k < 9; ++k) p.lastR[k] = p.R[k];
    for (int k = 0; k < 3; ++k) p.lastT[k] = p.t[k];
    if (mode == 0) {
        for (int k = 0; k < 9; ++k) p.R[k] = in.R[k];
        for (int k = 0; k < 3; ++k) p.t[k] = in.t[k];
    } else {
        float R[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[r * 3 + c] = p.lastR[r * 3] * in.R[c] + p.lastR[r * 3 + 1] * in.R[3 + c] + p.lastR[r * 3 + 2] * in.R[6 + c];
        const float3 t = mul33(p.lastR, f3(in.t[0], in.t[1], in.t[2]));
        for (int k = 0; k < 9; ++k) p.R[k] = R[k];
        p.t[0] = t.x + p.lastT[0]; p.t[1] = t.y + p.lastT[1]; p.t[2] = t.z + p.lastT[2];
    }
    pose_derive(p);
    *pose = p;
    if (host_mirror) *host_mirror = p;
}
void launch_override_pose(PoseDev* pose, const float* in_pose16_colmajor, int mode, PoseDev* host_mirror, hipStream_t s) {
    PoseArg a;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) a.R[r * 3 + c] = in_pose16_colmajor[c * 4 + r];
        a.t[r] = in_pose16_colmajor[12 + r];
    }
    hipLaunchKernelGGL(k_override_pose, dim3(1), dim3(64), 0, s, pose, a, mode, host_mirror);
}

// 16-float per-model record for the multi-GPU gather: R(9) t(3) lastICPError lastICPCount surfels alive
__global__ void k_model_state(const PoseDev* pose, const FrameDev* frame, float* out16) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int k = 0; k < 9; ++k) out16[k] = pose->R[k];
    for (int k = 0; k < 3; ++k) out16[9 + k] = pose->t[k];
    out16[12] = pose->lastICPError; out16[13] = pose->lastICPCount; out16[14] = (float)frame->count; out16[15] = (float)pose->alive;
}
void launch_model_state(const PoseDev* pose, const FrameDev* frame, float* out16, hipStream_t s) {
    hipLaunchKernelGGL(k_model_state, dim3(1), dim3(64), 0, s, pose, frame, out16);
}

}  // namespace mf
