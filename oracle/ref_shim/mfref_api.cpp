// mfref_api.cpp -- plain-pointer C entry points around the reference's OWN host wrappers (Core/Cuda/cudafuncs.cuh,
// segmentation.cuh), compiled for the CPU through mfref_cuda.h.  TEST INFRASTRUCTURE ONLY.
//
// Every function below uploads its arguments into the reference's DeviceArray containers, calls the reference
// function named in the comment -- the real one, compiled from /root/reference/Core/Cuda/*.cu -- and downloads the
// result.  Argument conventions mirror oracle/mf_oracle.h (planar [3][H][W] maps, row-major 3x3, dense images), so a
// test can hand the same arrays to mfo_* and mfref_*.
#include "cudafuncs.cuh"
#include "segmentation.cuh"

#include <vector>

namespace {

template <typename T>
DeviceArray2D<T> up2(const T* p, int rows, int cols) {
    DeviceArray2D<T> a;
    a.upload(p, (size_t)cols * sizeof(T), rows, cols);
    return a;
}
template <typename T>
void down2(const DeviceArray2D<T>& a, T* p) { a.download(p, (size_t)a.cols() * sizeof(T)); }

mat33 m33(const float* r) { mat33 m; memcpy(m.data, r, sizeof(m.data)); return m; }
float3 f3(const float* v) { return make_float3(v[0], v[1], v[2]); }

}  // namespace

extern "C" {

// pyrDownGaussF (cudafuncs.cu:510-532)
void mfref_pyrdown_gauss_f(const float* src, float* dst, int sw, int sh) {
    DeviceArray2D<float> s = up2(src, sh, sw), d;
    pyrDownGaussF(s, d);
    down2(d, dst);
}
// pyrDownUcharGauss (cudafuncs.cu:566-588)
void mfref_pyrdown_gauss_u8(const uint8_t* src, uint8_t* dst, int sw, int sh) {
    DeviceArray2D<unsigned char> s = up2(src, sh, sw), d;
    pyrDownUcharGauss(s, d);
    down2(d, dst);
}
// createVMap (cudafuncs.cu:136-150)
void mfref_create_vmap(const float* depth, float* vmap, int W, int H, float fx, float fy, float cx, float cy, float depthCutoff) {
    DeviceArray2D<float> d = up2(depth, H, W), v;
    createVMap(CameraModel(fx, fy, cx, cy), d, v, depthCutoff);
    down2(v, vmap);
}
// createNMap (cudafuncs.cu:191-205)
void mfref_create_nmap(const float* vmap, float* nmap, int W, int H) {
    DeviceArray2D<float> v = up2(vmap, 3 * H, W), n;
    createNMap(v, n);
    down2(n, nmap);
}
// copyMaps (cudafuncs.cu:313-332): AoS float4 -> planar
void mfref_copy_maps(const float* v4, const float* n4, float* vmap, float* nmap, int W, int H) {
    DeviceArray<float> vs, ns;
    vs.upload(v4, (size_t)W * H * 4);
    ns.upload(n4, (size_t)W * H * 4);
    DeviceArray2D<float> vd, nd;
    vd.create(3 * H, W);   // copyMaps reads the size of its destination (RGBDOdometry.cpp:98-100 pre-creates them)
    nd.create(3 * H, W);
    copyMaps(vs, ns, vd, nd);
    down2(vd, vmap);
    down2(nd, nmap);
}
// resizeVMap / resizeNMap (cudafuncs.cu:419-445)
void mfref_resize_map(const float* in, float* out, int sw, int sh, int normalize) {
    DeviceArray2D<float> i = up2(in, 3 * sh, sw), o;
    if (normalize) resizeNMap(i, o); else resizeVMap(i, o);
    down2(o, out);
}
// tranformMaps (cudafuncs.cu:251-269)
void mfref_transform_maps(const float* vsrc, const float* nsrc, const float* R, const float* t, float* vdst, float* ndst, int W, int H) {
    DeviceArray2D<float> v = up2(vsrc, 3 * H, W), n = up2(nsrc, 3 * H, W), vo, no;
    tranformMaps(v, n, m33(R), f3(t), vo, no);
    down2(vo, vdst);
    down2(no, ndst);
}
// icpStep (reduce.cu:446-525) with the launch configuration the caller picks (GPUConfig upstream)
void mfref_icp_step(const float* Rcurr, const float* tcurr, const float* vmap_curr, const float* nmap_curr, const float* Rprev_inv,
                    const float* tprev, float fx, float fy, float cx, float cy, const float* vmap_g_prev, const float* nmap_g_prev,
                    float distThres, float angleThres, int W, int H, float* A, float* b, float* residual, int threads, int blocks) {
    DeviceArray2D<float> vc = up2(vmap_curr, 3 * H, W), nc = up2(nmap_curr, 3 * H, W), vp = up2(vmap_g_prev, 3 * H, W),
                         np = up2(nmap_g_prev, 3 * H, W);
    DeviceArray<JtJJtrSE3> sum, out;
    sum.create(blocks);   // RGBDOdometry.cpp:65-66 sizes them MAX_THREADS / 1
    out.create(1);
    DeviceArray2D<unsigned char> noMask;
    icpStep(m33(Rcurr), f3(tcurr), vc, nc, m33(Rprev_inv), f3(tprev), CameraModel(fx, fy, cx, cy), vp, np, distThres, angleThres, sum,
            out, A, b, residual, threads, blocks, 0, noMask, 0);
}
// verticesToDepth (cudafuncs.cu:602-622)
void mfref_vertices_to_depth(const float* v4, float* depth, int W, int H, float cutOff) {
    DeviceArray<float> vs;
    vs.upload(v4, (size_t)W * H * 4);
    DeviceArray2D<float> d;
    d.create(H, W);
    verticesToDepth(vs, d, cutOff);
    down2(d, depth);
}
// imageBGRToIntensity (cudafuncs.cu:626-654): 4-byte texels of a cudaArray
void mfref_image_to_intensity(const uint8_t* rgba, uint8_t* dst, int W, int H) {
    cudaArray arr{W, H, rgba};
    DeviceArray2D<unsigned char> d;
    d.create(H, W);
    imageBGRToIntensity(&arr, d);
    down2(d, dst);
}
// computeDerivativeImages (cudafuncs.cu:658-718)
void mfref_derivative_images(const uint8_t* src, int16_t* dx, int16_t* dy, int W, int H) {
    DeviceArray2D<unsigned char> s = up2(src, H, W);
    DeviceArray2D<short> x, y;
    x.create(H, W);
    y.create(H, W);
    computeDerivativeImages(s, x, y);
    down2(x, (short*)dx);
    down2(y, (short*)dy);
}
// projectToPointCloud (cudafuncs.cu:722-751); the level-0 call with the level's own intrinsics
void mfref_project_to_cloud(const float* depth, float* cloud3, int W, int H, float fx, float fy, float cx, float cy) {
    DeviceArray2D<float> d = up2(depth, H, W);
    DeviceArray2D<float3> c;
    c.create(H, W);
    CameraModel k(fx, fy, cx, cy);
    projectToPointCloud(d, c, k, 0);
    down2(c, (float3*)cloud3);
}
// computeRgbResidual (reduce.cu:918-997).  corres16: W*H records {int16 zero.x, zero.y, one.x, one.y; float diff; int32 valid}
void mfref_rgb_residual(float minScale, const int16_t* dIdx, const int16_t* dIdy, const float* lastDepth, const float* nextDepth,
                        const uint8_t* lastImage, const uint8_t* nextImage, void* corres16, float maxDepthDelta, const float* kt,
                        const float* krkinv, int W, int H, int32_t* sigmaSum, int32_t* count, int threads, int blocks) {
    DeviceArray2D<short> dx = up2((const short*)dIdx, H, W), dy = up2((const short*)dIdy, H, W);
    DeviceArray2D<float> ld = up2(lastDepth, H, W), nd = up2(nextDepth, H, W);
    DeviceArray2D<unsigned char> li = up2(lastImage, H, W), ni = up2(nextImage, H, W), noMask;
    DeviceArray2D<DataTerm> corres;
    corres.create(H, W);
    DeviceArray<int2> sumResidual;
    sumResidual.create(blocks);
    int sig = 0, cnt = 0;
    computeRgbResidual(minScale, dx, dy, ld, nd, li, ni, noMask, noMask, corres, sumResidual, maxDepthDelta, f3(kt), m33(krkinv), sig, cnt,
                       threads, blocks, 0, 0);
    *sigmaSum = sig;
    *count = cnt;
    std::vector<DataTerm> h((size_t)W * H);
    down2(corres, h.data());
    struct Out { int16_t zx, zy, ox, oy; float diff; int32_t valid; };
    Out* o = (Out*)corres16;
    for (size_t i = 0; i < h.size(); ++i)
        o[i] = Out{h[i].zero.x, h[i].zero.y, h[i].one.x, h[i].one.y, h[i].diff, h[i].valid ? 1 : 0};
}
// rgbStep (reduce.cu:662-713)
void mfref_rgb_step(const void* corres16, float sigma, const float* cloud3, float fx, float fy, const int16_t* dIdx, const int16_t* dIdy,
                    float sobelScale, int W, int H, float* A, float* b, int threads, int blocks) {
    struct In { int16_t zx, zy, ox, oy; float diff; int32_t valid; };
    const In* in = (const In*)corres16;
    std::vector<DataTerm> h((size_t)W * H);
    for (size_t i = 0; i < h.size(); ++i) {
        memset(&h[i], 0, sizeof(DataTerm));
        h[i].zero = make_short2(in[i].zx, in[i].zy);
        h[i].one = make_short2(in[i].ox, in[i].oy);
        h[i].diff = in[i].diff;
        h[i].valid = in[i].valid != 0;
    }
    DeviceArray2D<DataTerm> corres = up2(h.data(), H, W);
    DeviceArray2D<float3> cloud = up2((const float3*)cloud3, H, W);
    DeviceArray2D<short> dx = up2((const short*)dIdx, H, W), dy = up2((const short*)dIdy, H, W);
    DeviceArray<JtJJtrSE3> sum, out;
    sum.create(blocks);
    out.create(1);
    rgbStep(corres, sigma, cloud, fx, fy, dx, dy, sobelScale, sum, out, A, b, threads, blocks);
}
// so3Step (reduce.cu:1143-1202)
void mfref_so3_step(const uint8_t* lastImage, const uint8_t* nextImage, const float* imageBasis, const float* kinv, const float* krlr,
                    int W, int H, float* A, float* b, float* residual, int threads, int blocks) {
    DeviceArray2D<unsigned char> li = up2(lastImage, H, W), ni = up2(nextImage, H, W);
    DeviceArray<JtJJtrSO3> sum, out;
    sum.create(blocks);
    out.create(1);
    so3Step(li, ni, m33(imageBasis), m33(kinv), m33(krlr), sum, out, A, b, residual, threads, blocks);
}
// computeGeometricSegmentationMap (segmentation.cu:277-292)
void mfref_geometric_edge_map(const float* vmap, const float* nmap, float* out, int W, int H, float wD, float wC) {
    DeviceArray2D<float> v = up2(vmap, 3 * H, W), n = up2(nmap, 3 * H, W), o;
    o.create(H, W);
    computeGeometricSegmentationMap(v, n, o, wD, wC);
    down2(o, out);
}
// thresholdMap (segmentation.cu:294-302)
void mfref_threshold_map(const float* in, uint8_t* out, int W, int H, float threshold) {
    DeviceArray2D<float> i = up2(in, H, W);
    DeviceArray2D<unsigned char> o;
    o.create(H, W);
    thresholdMap(i, o, threshold);
    down2(o, out);
}
// invertMap (segmentation.cu:304-311)
void mfref_invert_map(const uint8_t* in, uint8_t* out, int W, int H) {
    DeviceArray2D<unsigned char> i = up2(in, H, W), o;
    o.create(H, W);
    invertMap(i, o);
    down2(o, out);
}
// morphGeometricSegmentationMap, the uchar overload MfSegmentation uses (segmentation.cu:334-354); in place on data
void mfref_morph_closing_u8(uint8_t* data, int W, int H, int radius, int iterations) {
    DeviceArray2D<unsigned char> d = up2(data, H, W), buf;
    buf.create(H, W);
    morphGeometricSegmentationMap(d, buf, radius, iterations);
    down2(d, data);
}

}  // extern "C"
