/*
 * mf_oracle.c -- CPU restatement of the MaskFusion::processFrame hot path (see mf_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into or called by the product (maskfusion_amd/).
 * PARITY UNPINNED by the reference (no tests / golden vectors upstream, reference not buildable here);
 * pinned by analytic KATs in tests/test_oracle_kat.py.  Citations are relative to /root/reference/.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC mf_oracle.c -lm  (see Makefile)
 * -ffp-contract=off keeps every float op individually rounded, like a plain reading of the sources.
 */
#include "mf_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MFO_NAN (__builtin_nanf(""))

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* __float2int_rn: round-to-nearest-even, NaN -> 0, saturating (CUDA semantics; reduce.cu:304-305) */
static inline int f2i_rn(float v) {
    if (isnan(v)) return 0;
    if (v >= 2147483648.0f) return INT_MAX;
    if (v <= -2147483648.0f) return INT_MIN;
    return (int)rintf(v);
}

/* ------------------------------------------------------------------------------------------------
 * small float 3-vector helpers (Core/Cuda/operators.cuh:57-91)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } f3;
static inline f3 f3_make(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 f3_add(f3 a, f3 b) { return f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 f3_scale(f3 a, float s) { return f3_make(a.x * s, a.y * s, a.z * s); }
static inline f3 f3_cross(f3 a, f3 b) {
    return f3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float f3_dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float f3_norm(f3 a) { return sqrtf(f3_dot(a, a)); }
static inline f3 f3_normalized(f3 a) {
    const float rn = 1.0f / sqrtf(f3_dot(a, a)); /* rsqrtf */
    return f3_make(a.x * rn, a.y * rn, a.z * rn);
}
/* GLSL normalize(): v / length(v) */
static inline f3 f3_glnormalize(f3 a) {
    const float l = sqrtf(f3_dot(a, a));
    return f3_make(a.x / l, a.y / l, a.z / l);
}
static inline f3 m33_mul(const float* R, f3 a) { /* row-major 3x3 */
    return f3_make(R[0] * a.x + R[1] * a.y + R[2] * a.z, R[3] * a.x + R[4] * a.y + R[5] * a.z,
                   R[6] * a.x + R[7] * a.y + R[8] * a.z);
}

/* float 3x3 inverse by cofactors (Eigen fixed-size inverse stand-in; RGBDOdometry.cpp:332) */
static void m33_inverse(const float* m, float* inv) {
    const float c00 = m[4] * m[8] - m[5] * m[7];
    const float c01 = m[5] * m[6] - m[3] * m[8];
    const float c02 = m[3] * m[7] - m[4] * m[6];
    const float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const float id = 1.0f / det;
    inv[0] = c00 * id;
    inv[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    inv[3] = c01 * id;
    inv[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    inv[6] = c02 * id;
    inv[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

/* column-major 4x4 <-> (R row-major, t) */
static void pose16_to_Rt(const float* p, float* R, float* t) {
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = p[c * 4 + r];
        t[r] = p[12 + r];
    }
}
static void Rt_to_pose16(const float* R, const float* t, float* p) {
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) p[c * 4 + r] = R[r * 3 + c];
        p[12 + r] = t[r];
        p[r * 4 + 3] = 0.f;
    }
    p[15] = 1.f;
}
/* pose.inverse() for an affine [R|t;0 0 0 1] (Eigen general inverse stand-in; ModelProjection.cpp:114) */
static void pose_inverse_Rt(const float* R, const float* t, float* Ri, float* ti) {
    m33_inverse(R, Ri);
    f3 v = m33_mul(Ri, f3_make(t[0], t[1], t[2]));
    ti[0] = -v.x; ti[1] = -v.y; ti[2] = -v.z;
}

/* the t_inv uniform every projection pass is handed (column-major 4x4 in and out) */
void mfo_pose_inverse16(const float* pose16, float* out16) {
    float R[9], t[3], Ri[9], ti[3];
    pose16_to_Rt(pose16, R, t);
    pose_inverse_Rt(R, t, Ri, ti);
    Rt_to_pose16(Ri, ti, out16);
}

/* ------------------------------------------------------------------------------------------------
 * a2: bilateral filter (Core/Shaders/depth_bilateral_metric.frag:30-76)
 * ---------------------------------------------------------------------------------------------- */
void mfo_bilateral(const float* depth, float* out, int W, int H) {
    const float sigma_space2_inv_half = 0.024691358f;
    const float sigma_color2_inv_half = 555.556f;
    const int R = 6, D = R * 2 + 1;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            const float value = depth[y * W + x];
            if (value <= 0.03f) { out[y * W + x] = 0.f; continue; }
            const int tx = imin(x - D / 2 + D, W);
            const int ty = imin(y - D / 2 + D, H);
            float sum1 = 0.f, sum2 = 0.f;
            for (int cy = imax(y - D / 2, 0); cy < ty; ++cy) {
                for (int cx = imax(x - D / 2, 0); cx < tx; ++cx) {
                    const float tmp = depth[cy * W + cx];
                    const float space2 = ((float)x - (float)cx) * ((float)x - (float)cx) +
                                         ((float)y - (float)cy) * ((float)y - (float)cy);
                    const float color2 = (value - tmp) * (value - tmp);
                    const float weight = expf(-(space2 * sigma_space2_inv_half + color2 * sigma_color2_inv_half));
                    sum1 += tmp * weight;
                    sum2 += weight;
                }
            }
            out[y * W + x] = sum1 / sum2;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * a3: pyramids + vertex / normal maps
 * ---------------------------------------------------------------------------------------------- */
static const float kGauss5[25] = {1, 4, 6, 4, 1, 4, 16, 24, 16, 4, 6, 24, 36, 24, 6, 4, 16, 24, 16, 4, 1, 4, 6, 4, 1};

/* Core/Cuda/cudafuncs.cu:333-364; quirk Q9: upper bounds clamp to cols-1 / rows-1 EXCLUSIVE and the kernel
 * is indexed from the far corner. */
void mfo_pyrdown_gauss_f(const float* src, float* dst, int sw, int sh) {
    const int dw = sw / 2, dh = sh / 2, D = 5;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        for (int x = 0; x < dw; ++x) {
            const int tx = imin(2 * x - D / 2 + D, sw - 1);
            const int ty = imin(2 * y - D / 2 + D, sh - 1);
            float sum = 0.f;
            int count = 0;
            for (int cy = imax(0, 2 * y - D / 2); cy < ty; ++cy) {
                for (int cx = imax(0, 2 * x - D / 2); cx < tx; ++cx) {
                    const float v = src[cy * sw + cx];
                    if (!isnan(v)) {
                        const float w = kGauss5[(ty - cy - 1) * 5 + (tx - cx - 1)];
                        sum += v * w;
                        count += (int)w;
                    }
                }
            }
            dst[y * dw + x] = sum / (float)count; /* 0/0 -> NaN, as on the device */
        }
    }
}

/* Core/Cuda/cudafuncs.cu:534-564 */
void mfo_pyrdown_gauss_u8(const uint8_t* src, uint8_t* dst, int sw, int sh) {
    const int dw = sw / 2, dh = sh / 2, D = 5;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        for (int x = 0; x < dw; ++x) {
            const int tx = imin(2 * x - D / 2 + D, sw - 1);
            const int ty = imin(2 * y - D / 2 + D, sh - 1);
            float sum = 0.f;
            int count = 0;
            for (int cy = imax(0, 2 * y - D / 2); cy < ty; ++cy) {
                for (int cx = imax(0, 2 * x - D / 2); cx < tx; ++cx) {
                    const uint8_t v = src[cy * sw + cx];
                    if (v > 0) {
                        const float w = kGauss5[(ty - cy - 1) * 5 + (tx - cx - 1)];
                        sum += (float)v * w;
                        count += (int)w;
                    }
                }
            }
            /* float -> uchar conversion truncates; 0/0 = NaN converts to 0 on the device */
            dst[y * dw + x] = count ? (uint8_t)(sum / (float)count) : 0;
        }
    }
}

/* Core/Cuda/cudafuncs.cu:109-134.  Deviation (documented): invalid pixels get NaN in all three planes; the
 * reference writes x=NaN, z=0 and leaves y stale -- every consumer tests only isnan(x). */
void mfo_create_vmap(const float* depth, float* vmap, int W, int H, float fx, float fy, float cx, float cy,
                     float depthCutoff) {
    const float fx_inv = 1.f / fx, fy_inv = 1.f / fy;
    const int P = W * H;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < H; ++v) {
        for (int u = 0; u < W; ++u) {
            const float z = depth[v * W + u];
            if (z > 0.0f && z < depthCutoff) {
                vmap[v * W + u] = z * ((float)u - cx) * fx_inv;
                vmap[P + v * W + u] = z * ((float)v - cy) * fy_inv;
                vmap[2 * P + v * W + u] = z;
            } else {
                vmap[v * W + u] = MFO_NAN;
                vmap[P + v * W + u] = MFO_NAN;
                vmap[2 * P + v * W + u] = MFO_NAN;
            }
        }
    }
}

/* Core/Cuda/cudafuncs.cu:152-189 */
void mfo_create_nmap(const float* vmap, float* nmap, int W, int H) {
    const int P = W * H;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < H; ++v) {
        for (int u = 0; u < W; ++u) {
            const int i = v * W + u;
            int ok = !(u == W - 1 || v == H - 1);
            if (ok) {
                const float x00 = vmap[i], x01 = vmap[i + 1], x10 = vmap[i + W];
                ok = !isnan(x00) && !isnan(x01) && !isnan(x10);
            }
            if (ok) {
                f3 v00 = f3_make(vmap[i], vmap[P + i], vmap[2 * P + i]);
                f3 v01 = f3_make(vmap[i + 1], vmap[P + i + 1], vmap[2 * P + i + 1]);
                f3 v10 = f3_make(vmap[i + W], vmap[P + i + W], vmap[2 * P + i + W]);
                f3 r = f3_normalized(f3_cross(f3_sub(v01, v00), f3_sub(v10, v00)));
                nmap[i] = r.x; nmap[P + i] = r.y; nmap[2 * P + i] = r.z;
            } else {
                nmap[i] = MFO_NAN; nmap[P + i] = MFO_NAN; nmap[2 * P + i] = MFO_NAN;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * a4: model-side maps
 * ---------------------------------------------------------------------------------------------- */
/* Core/Cuda/cudafuncs.cu:271-311 */
void mfo_copy_maps(const float* v4, const float* n4, float* vmap, float* nmap, int W, int H) {
    const int P = W * H;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        const float vx = v4[i * 4 + 0], vy = v4[i * 4 + 1], vz = v4[i * 4 + 2];
        if (!(vz == 0)) {
            vmap[i] = vx; vmap[P + i] = vy; vmap[2 * P + i] = vz;
            nmap[i] = n4[i * 4 + 0]; nmap[P + i] = n4[i * 4 + 1]; nmap[2 * P + i] = n4[i * 4 + 2];
        } else {
            vmap[i] = vmap[P + i] = vmap[2 * P + i] = MFO_NAN;
            nmap[i] = nmap[P + i] = nmap[2 * P + i] = MFO_NAN;
        }
    }
}

/* Core/Cuda/cudafuncs.cu:366-417 */
void mfo_resize_map(const float* in, float* out, int sw, int sh, int normalize) {
    const int dw = sw / 2, dh = sh / 2, SP = sw * sh, DP = dw * dh;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        for (int x = 0; x < dw; ++x) {
            const int s = (2 * y) * sw + 2 * x, d = y * dw + x;
            const float x00 = in[s], x01 = in[s + 1], x10 = in[s + sw], x11 = in[s + sw + 1];
            if (isnan(x00) || isnan(x01) || isnan(x10) || isnan(x11)) {
                out[d] = out[DP + d] = out[2 * DP + d] = MFO_NAN;
                continue;
            }
            f3 n;
            n.x = (x00 + x01 + x10 + x11) / 4;
            n.y = (in[SP + s] + in[SP + s + 1] + in[SP + s + sw] + in[SP + s + sw + 1]) / 4;
            n.z = (in[2 * SP + s] + in[2 * SP + s + 1] + in[2 * SP + s + sw] + in[2 * SP + s + sw + 1]) / 4;
            if (normalize) n = f3_normalized(n);
            out[d] = n.x; out[DP + d] = n.y; out[2 * DP + d] = n.z;
        }
    }
}

/* Core/Cuda/cudafuncs.cu:207-249 */
void mfo_transform_maps(const float* vsrc, const float* nsrc, const float* R, const float* t, float* vdst,
                        float* ndst, int W, int H) {
    const int P = W * H;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        f3 vd = f3_make(MFO_NAN, MFO_NAN, MFO_NAN), nd = vd;
        if (!isnan(vsrc[i])) {
            f3 v = m33_mul(R, f3_make(vsrc[i], vsrc[P + i], vsrc[2 * P + i]));
            vd = f3_make(v.x + t[0], v.y + t[1], v.z + t[2]);
        }
        if (!isnan(nsrc[i])) nd = m33_mul(R, f3_make(nsrc[i], nsrc[P + i], nsrc[2 * P + i]));
        vdst[i] = vd.x; vdst[P + i] = vd.y; vdst[2 * P + i] = vd.z;
        ndst[i] = nd.x; ndst[P + i] = nd.y; ndst[2 * P + i] = nd.z;
    }
}

/* ------------------------------------------------------------------------------------------------
 * a7: ICP normal equations (Core/Cuda/reduce.cu:259-525).  Sums accumulate in double (the reference's
 * float tree order is unspecified and fast-math; double is the neutral anchor).
 * ---------------------------------------------------------------------------------------------- */
void mfo_icp_step(const float* Rcurr, const float* tcurr, const float* vmap_curr, const float* nmap_curr,
                  const float* Rprev_inv, const float* tprev, float fx, float fy, float cx, float cy,
                  const float* vmap_g_prev, const float* nmap_g_prev, float distThres, float angleThres, int W,
                  int H, float* A, float* b, float* residual) {
    const int P = W * H;
    double acc[29];
    for (int k = 0; k < 29; ++k) acc[k] = 0.0;
    const f3 tc = f3_make(tcurr[0], tcurr[1], tcurr[2]);
    const f3 tp = f3_make(tprev[0], tprev[1], tprev[2]);

#pragma omp parallel
    {
        double loc[29];
        for (int k = 0; k < 29; ++k) loc[k] = 0.0;
#pragma omp for schedule(static) nowait
        for (int y = 0; y < H; ++y) {
            for (int x = 0; x < W; ++x) {
                const int i = y * W + x;
                /* search(): reduce.cu:292-353 */
                const f3 vcurr = f3_make(vmap_curr[i], vmap_curr[P + i], vmap_curr[2 * P + i]);
                const f3 vcurr_g = f3_add(m33_mul(Rcurr, vcurr), tc);
                const f3 vcurr_cp = m33_mul(Rprev_inv, f3_sub(vcurr_g, tp));
                const int ux = f2i_rn(vcurr_cp.x * fx / vcurr_cp.z + cx);
                const int uy = f2i_rn(vcurr_cp.y * fy / vcurr_cp.z + cy);
                if (ux < 0 || uy < 0 || ux >= W || uy >= H || vcurr_cp.z < 0) continue;
                const int j = uy * W + ux;
                const f3 vprev_g = f3_make(vmap_g_prev[j], vmap_g_prev[P + j], vmap_g_prev[2 * P + j]);
                const f3 ncurr = f3_make(nmap_curr[i], nmap_curr[P + i], nmap_curr[2 * P + i]);
                const f3 ncurr_g = m33_mul(Rcurr, ncurr);
                const f3 nprev_g = f3_make(nmap_g_prev[j], nmap_g_prev[P + j], nmap_g_prev[2 * P + j]);
                const float dist = f3_norm(f3_sub(vprev_g, vcurr_g));
                const float sine = f3_norm(f3_cross(ncurr_g, nprev_g));
                if (!(sine < angleThres && dist <= distThres && !isnan(ncurr.x) && !isnan(nprev_g.x))) continue;
                /* getProducts(): reduce.cu:355-415 */
                const f3 s_cp = m33_mul(Rprev_inv, f3_sub(vcurr_g, tp));
                const f3 d_cp = m33_mul(Rprev_inv, f3_sub(vprev_g, tp));
                const f3 n_cp = m33_mul(Rprev_inv, nprev_g);
                const f3 sxn = f3_cross(s_cp, n_cp);
                float row[7];
                row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z;
                row[3] = sxn.x; row[4] = sxn.y; row[5] = sxn.z;
                row[6] = f3_dot(n_cp, f3_sub(s_cp, d_cp));
                int k = 0;
                for (int a = 0; a < 7; ++a)
                    for (int c = a; c < 7; ++c) loc[k++] += (double)(row[a] * row[c]);
                loc[28] += 1.0;
            }
        }
#pragma omp critical
        for (int k = 0; k < 29; ++k) acc[k] += loc[k];
    }
    /* host unpack: reduce.cu:507-524.  acc order: aa..ag, bb..bg, ..., ff, fg, gg(=residual), inliers */
    int shift = 0;
    for (int i = 0; i < 6; ++i) {
        for (int j = i; j < 7; ++j) {
            const float value = (float)acc[shift++];
            if (j == 6) b[i] = value;
            else A[j * 6 + i] = A[i * 6 + j] = value;
        }
    }
    residual[0] = (float)acc[27];
    residual[1] = (float)acc[28];
}

/* ------------------------------------------------------------------------------------------------
 * a11: solve + SE(3) update
 * ---------------------------------------------------------------------------------------------- */
/* Symmetric LDL^T with diagonal pivoting, double (stand-in for Eigen::LDLT, RGBDOdometry.cpp:313,447-459).
 * Like Eigen, a (near-)zero pivot yields a zero component rather than inf. */
int mfo_ldlt_solve(const double* Ain, const double* bin, double* x, int n) {
    double A[36], b[6];
    int perm[6];
    if (n > 6) return -1;
    for (int i = 0; i < n * n; ++i) A[i] = Ain[i];
    for (int i = 0; i < n; ++i) { b[i] = bin[i]; perm[i] = i; }
    double maxdiag = 0;
    for (int i = 0; i < n; ++i) maxdiag = fmax(maxdiag, fabs(A[i * n + i]));
    const double tol = maxdiag * 1e-300 + 1e-300;
    for (int k = 0; k < n; ++k) {
        /* pivot: largest remaining |diagonal| */
        int p = k;
        for (int i = k + 1; i < n; ++i)
            if (fabs(A[i * n + i]) > fabs(A[p * n + p])) p = i;
        if (p != k) {
            for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
            for (int j = 0; j < n; ++j) { double t = A[j * n + k]; A[j * n + k] = A[j * n + p]; A[j * n + p] = t; }
            { double t = b[k]; b[k] = b[p]; b[p] = t; }
            { int t = perm[k]; perm[k] = perm[p]; perm[p] = t; }
        }
        const double d = A[k * n + k];
        if (fabs(d) <= tol) continue;
        for (int i = k + 1; i < n; ++i) {
            const double l = A[i * n + k] / d;
            for (int j = k + 1; j < n; ++j) A[i * n + j] -= l * A[k * n + j];
            A[i * n + k] = l; /* store L */
        }
    }
    /* forward: L y = b */
    double y[6];
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int j = 0; j < i; ++j) s -= A[i * n + j] * y[j];
        y[i] = s;
    }
    /* diagonal */
    for (int i = 0; i < n; ++i) y[i] = (fabs(A[i * n + i]) > tol) ? y[i] / A[i * n + i] : 0.0;
    /* backward: L^T z = y */
    double z[6];
    for (int i = n - 1; i >= 0; --i) {
        double s = y[i];
        for (int j = i + 1; j < n; ++j) s -= A[j * n + i] * z[j];
        z[i] = s;
    }
    for (int i = 0; i < n; ++i) x[perm[i]] = z[i];
    return 0;
}

/* Core/Utils/OdometryProvider.h:32-67 */
void mfo_rodrigues(const double* w, double* R) {
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    double rx = w[0], ry = w[1], rz = w[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta >= 2.2204460492503131e-16) {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double c = cos(theta), s = sin(theta), c1 = 1. - c;
        const double itheta = theta ? 1. / theta : 0.;
        rx *= itheta; ry *= itheta; rz *= itheta;
        const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        const double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        for (int k = 0; k < 9; ++k) R[k] = c * I[k] + c1 * rrt[k] + s * rx_[k];
    }
}

/* Core/Utils/OdometryProvider.h:69-90 (the float Isometry out-parameter is derived by the caller) */
void mfo_update_se3(double* resultRt, const double* x6) {
    double Rt[16], R[9], out[16];
    mfo_rodrigues(x6 + 3, R);
    for (int k = 0; k < 16; ++k) Rt[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Rt[r * 4 + c] = R[r * 3 + c];
        Rt[r * 4 + 3] = x6[r];
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += Rt[r * 4 + k] * resultRt[k * 4 + c];
            out[r * 4 + c] = s;
        }
    memcpy(resultRt, out, sizeof(out));
}

/* Core/Utils/RGBDOdometry.cpp:227-497, icp && !rgb branch (so3 pre-alignment handled by the caller when the
 * photometric data exists; here resultRt starts at identity). */
/* Stated domain of the Gauss-Newton solve (DESIGN.md finding F4): an iteration is "ill" when its system has fewer than 6 inliers or a pivot of
 * the UNPIVOTED LDL^T below 1e-8 of the largest diagonal entry (=> cond(A) > 1e8).  There this restatement (Eigen-style pivoted LDLT on the
 * float-rounded sums, as the reference) and the device (unpivoted LDL^T on fp64 sums) may step differently; both count such iterations. */
static int g_track_ill = 0;
int mfo_last_track_ill(void) { return g_track_ill; }
/* the reduced geometric systems of the last tracking step, one row per iteration in the device log's layout: 27 packed upper-triangle
 * products of the 7-vector row (reduce.cu:378-411 order: A[i][i..5], b[i]), then the residual and the inlier count (test tooling) */
static float g_track_log[20][32]; static int g_track_log_n = 0;
static void track_log_push(const float* A36, const float* b6, const float* residual2) {
    if (g_track_log_n >= 20) return;
    float* row = g_track_log[g_track_log_n++];
    int k = 0;
    for (int i = 0; i < 6; ++i) { for (int j = i; j < 6; ++j) row[k++] = A36[i * 6 + j]; row[k++] = b6[i]; }
    row[27] = residual2[0]; row[28] = residual2[1]; row[29] = row[30] = row[31] = 0.f;
}
int mfo_last_track_log(float* out /* [20][32] */) { memcpy(out, g_track_log, sizeof(g_track_log)); return g_track_log_n; }
static int gn_system_ill(const double* A36, float inliers) {
    double M[6][6], maxdiag = 0.0, minpiv = 1.7976931348623157e308;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) M[i][j] = A36[i * 6 + j];
    for (int i = 0; i < 6; ++i) maxdiag = fmax(maxdiag, fabs(M[i][i]));
    const double tol = maxdiag * 1e-14;
    for (int k = 0; k < 6; ++k) {
        const double d = M[k][k];
        const int ok = d > tol;
        minpiv = fmin(minpiv, ok ? d : 0.0);
        const double dinv = ok ? 1.0 / d : 0.0;
        for (int i = k + 1; i < 6; ++i) {
            const double l = M[i][k] * dinv;
            for (int j = k + 1; j <= i; ++j) M[i][j] -= l * M[j][k];
        }
    }
    const double ratio = maxdiag > 0.0 ? minpiv / maxdiag : 0.0;
    return (inliers < 6.f || !(ratio >= 1e-8)) ? 1 : 0;
}

void mfo_track_icp(const float* const curr_v[3], const float* const curr_n[3], const float* const prev_v[3],
                   const float* const prev_n[3], int W, int H, float fx, float fy, float cx, float cy,
                   const mfo_track_opts* o, float* R, float* t, float* out_inc16, float* lastICPError,
                   float* lastICPCount, mfo_track_log* log) {
    g_track_ill = 0;
    float Rprev[9], tprev[3], Rcurr[9], tcurr[3], Rprev_inv[9];
    memcpy(Rprev, R, sizeof(Rprev)); memcpy(tprev, t, sizeof(tprev));
    memcpy(Rcurr, R, sizeof(Rcurr)); memcpy(tcurr, t, sizeof(tcurr));
    int iterations[3];
    iterations[0] = o->fastOdom ? 3 : 10;
    iterations[1] = o->pyramid ? 5 : 0;
    iterations[2] = o->pyramid ? 4 : 0;
    m33_inverse(Rprev, Rprev_inv);
    double resultRt[16];
    for (int k = 0; k < 16; ++k) resultRt[k] = (k % 5 == 0) ? 1.0 : 0.0;
    float trR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, trt[3] = {0, 0, 0}; /* Isometry3f transform */
    int it = 0;
    if (log) log->n_iters = 0;
    for (int i = 2; i >= 0; --i) {
        const int div = 1 << i;
        const float lfx = fx / div, lfy = fy / div, lcx = cx / div, lcy = cy / div; /* types.cuh:94-98 */
        const int lw = W >> i, lh = H >> i;
        for (int j = 0; j < iterations[i]; ++j) {
            float A[36], b[6], residual[2];
            mfo_icp_step(Rcurr, tcurr, curr_v[i], curr_n[i], Rprev_inv, tprev, lfx, lfy, lcx, lcy, prev_v[i],
                         prev_n[i], o->distThresh, o->angleThresh, lw, lh, A, b, residual);
            *lastICPError = sqrtf(residual[0]) / residual[1];
            *lastICPCount = residual[1];
            double dA[36], db[6], x[6];
            for (int k = 0; k < 36; ++k) dA[k] = A[k];
            for (int k = 0; k < 6; ++k) db[k] = b[k];
            g_track_ill += gn_system_ill(dA, residual[1]);
            mfo_ldlt_solve(dA, db, x, 6);
            mfo_update_se3(resultRt, x);
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) trR[r * 3 + c] = (float)resultRt[r * 4 + c];
                trt[r] = (float)resultRt[r * 4 + 3];
            }
            /* currentT = [Rprev|tprev] * transform.inverse()  (Isometry inverse: R^T, -R^T t) */
            float iR[9], it3[3];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) iR[r * 3 + c] = trR[c * 3 + r];
            f3 itv = m33_mul(iR, f3_make(trt[0], trt[1], trt[2]));
            it3[0] = -itv.x; it3[1] = -itv.y; it3[2] = -itv.z;
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c)
                    Rcurr[r * 3 + c] = Rprev[r * 3 + 0] * iR[0 * 3 + c] + Rprev[r * 3 + 1] * iR[1 * 3 + c] +
                                       Rprev[r * 3 + 2] * iR[2 * 3 + c];
            }
            f3 tv = m33_mul(Rprev, f3_make(it3[0], it3[1], it3[2]));
            tcurr[0] = tv.x + tprev[0]; tcurr[1] = tv.y + tprev[1]; tcurr[2] = tv.z + tprev[2];
            if (log && it < 19) {
                memcpy(log->A[it], A, sizeof(A)); memcpy(log->b[it], b, sizeof(b));
                memcpy(log->residual[it], residual, sizeof(residual)); memcpy(log->x[it], x, sizeof(x));
                log->n_iters = it + 1;
            }
            ++it;
        }
    }
    memcpy(R, Rcurr, sizeof(Rcurr)); memcpy(t, tcurr, sizeof(tcurr));
    if (out_inc16) Rt_to_pose16(trR, trt, out_inc16);
}

/* =====================================================================================================
 * a5, a8-a10, a12: photometric term + SO(3) pre-alignment (Core/Utils/RGBDOdometry.cpp, Core/Cuda/reduce.cu,
 * Core/Cuda/cudafuncs.cu).  Where a float expression feeds an integer truncation / rounding the nvcc-style fused
 * form (fmaf) is written out explicitly; the HIP kernels use the same form.
 * ===================================================================================================== */
/* verticesToDepthKernel, cudafuncs.cu:602-614 (v4: float4 per pixel) */
void mfo_vertices_to_depth(const float* v4, float* depth, int n, float cutOff) {
    for (int i = 0; i < n; ++i) {
        const float z = v4[(size_t)i * 4 + 2];
        depth[i] = (z > cutOff || z <= 0) ? MFO_NAN : z;
    }
}

/* bgr2IntensityKernel, cudafuncs.cu:626-639: int(x * 0.114 + y * 0.299 + z * 0.587) on the first three channels AS STORED
 * (the frame texture holds R,G,B, so red gets the blue weight -- reproduced). */
void mfo_image_to_intensity(const uint8_t* img, int channels, uint8_t* dst, int n) {
    for (int i = 0; i < n; ++i) {
        const uint8_t* p = img + (size_t)i * channels;
        const float v = fmaf((float)p[2], 0.587f, fmaf((float)p[1], 0.299f, (float)p[0] * 0.114f));
        dst[i] = (uint8_t)(int)v;
    }
}

/* applyKernel, cudafuncs.cu:658-683 (kernel walked backwards from index 8 over the CLAMPED window -- border quirk) */
void mfo_derivative_images(const uint8_t* src, int16_t* dx, int16_t* dy, int W, int H) {
    static const float gx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
    static const float gy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float dxVal = 0, dyVal = 0;
            int k = 8;
            for (int j = imax(y - 1, 0); j <= imin(y + 1, H - 1); ++j)
                for (int i = imax(x - 1, 0); i <= imin(x + 1, W - 1); ++i) {
                    const float s = (float)src[j * W + i];
                    dxVal = fmaf(s, gx[k], dxVal);
                    dyVal = fmaf(s, gy[k], dyVal);
                    --k;
                }
            dx[y * W + x] = (int16_t)dxVal;  /* float -> short: truncation (|value| < 470) */
            dy[y * W + x] = (int16_t)dyVal;
        }
}

/* projectPointsKernel, cudafuncs.cu:722-738 */
void mfo_project_to_cloud(const float* depth, float* cloud3, int W, int H, float fx, float fy, float cx, float cy) {
    const float invFx = 1.0f / fx, invFy = 1.0f / fy;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float z = depth[y * W + x];
            float* c = cloud3 + (size_t)(y * W + x) * 3;
            c[0] = ((float)x - cx) * z * invFx;
            c[1] = ((float)y - cy) * z * invFy;
            c[2] = z;
        }
}

/* RGBResidual::getProducts + computeRgbResidual, reduce.cu:812-997 (MASK_RGB_RESIDUAL is not defined upstream: the
 * mask pyramids are allocated but never consulted).  count / sigmaSum are int32 with wrap-around like the int2 sums. */
void mfo_rgb_residual(float minScale, const int16_t* dIdx, const int16_t* dIdy, const float* lastDepth,
                      const float* nextDepth, const uint8_t* lastImage, const uint8_t* nextImage, mfo_dataterm* corres,
                      float maxDepthDelta, const float* kt, const float* krkinv, int W, int H, int32_t* sigmaSum,
                      int32_t* count) {
    uint32_t cnt = 0, sig = 0;
    for (int i = 0; i < H; ++i)
        for (int j0 = 0; j0 < W; ++j0) {
            mfo_dataterm c;
            memset(&c, 0, sizeof(c));
            if (j0 < W - 5 && i < H - 1) {
                int valid = 1;
                for (int u = imax(i - 2, 0); u < imin(i + 2, H); ++u)
                    for (int v = imax(j0 - 2, 0); v < imin(j0 + 2, W); ++v) valid = valid && (nextImage[u * W + v] > 0);
                if (valid) {
                    const int valx = dIdx[i * W + j0], valy = dIdy[i * W + j0];
                    const float mTwo = (float)((valx * valx) + (valy * valy));
                    if (mTwo >= minScale) {
                        const int y = i, x = j0;
                        const float d1 = nextDepth[y * W + x];
                        if (!isnan(d1)) {
                            const float fx_ = (float)x, fy_ = (float)y;
                            const float l2 = fmaf(krkinv[7], fy_, krkinv[6] * fx_) + krkinv[8];
                            const float l0 = fmaf(krkinv[1], fy_, krkinv[0] * fx_) + krkinv[2];
                            const float l1 = fmaf(krkinv[4], fy_, krkinv[3] * fx_) + krkinv[5];
                            const float td1 = fmaf(d1, l2, kt[2]);
                            const int u0 = f2i_rn(fmaf(d1, l0, kt[0]) / td1);
                            const int v0 = f2i_rn(fmaf(d1, l1, kt[1]) / td1);
                            if (u0 >= 0 && v0 >= 0 && u0 < W && v0 < H) {
                                const float d0 = lastDepth[v0 * W + u0];
                                if (d0 > 0 && fabsf(td1 - d0) <= maxDepthDelta && lastImage[v0 * W + u0] != 0) {
                                    c.zx = (int16_t)u0; c.zy = (int16_t)v0; c.ox = (int16_t)x; c.oy = (int16_t)y;
                                    c.diff = (float)nextImage[y * W + x] - (float)lastImage[v0 * W + u0];
                                    c.valid = 1;
                                    cnt += 1u;
                                    sig += (uint32_t)(int32_t)(c.diff * c.diff);
                                }
                            }
                        }
                    }
                }
            }
            corres[i * W + j0] = c;
        }
    *count = (int32_t)cnt;
    *sigmaSum = (int32_t)sig;
}

/* RGBReduction::getProducts + rgbStep, reduce.cu:529-713.  Sums in double, handed back as float like the device. */
void mfo_rgb_step(const mfo_dataterm* corres, float sigma, const float* cloud3, float fx, float fy, const int16_t* dIdx,
                  const int16_t* dIdy, float sobelScale, int W, int H, float* A, float* b) {
    double acc[29];
    for (int k = 0; k < 29; ++k) acc[k] = 0;
    for (int i = 0; i < W * H; ++i) {
        const mfo_dataterm* c = &corres[i];
        if (!c->valid) continue;
        float w = sigma + fabsf(c->diff);
        w = w > 1.1920929e-07f ? 1.0f / w : 1.0f;
        if (sigma == -1) w = 1;
        float row[7];
        row[6] = -w * c->diff;
        const float* cp = cloud3 + (size_t)(c->zy * W + c->zx) * 3;
        const float invz = (float)(1.0 / cp[2]);
        const float dI_dx_val = w * sobelScale * (float)dIdx[c->oy * W + c->ox];
        const float dI_dy_val = w * sobelScale * (float)dIdy[c->oy * W + c->ox];
        const float v0 = dI_dx_val * fx * invz;
        const float v1 = dI_dy_val * fy * invz;
        const float v2 = -(v0 * cp[0] + v1 * cp[1]) * invz;
        row[0] = v0; row[1] = v1; row[2] = v2;
        row[3] = -cp[2] * v1 + cp[1] * v2;
        row[4] = cp[2] * v0 - cp[0] * v2;
        row[5] = -cp[1] * v0 + cp[0] * v1;
        int k = 0;
        for (int r = 0; r < 7; ++r)
            for (int cc = r; cc < 7; ++cc) acc[k++] += (double)(row[r] * row[cc]);
        acc[28] += 1.0;
    }
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const float value = (float)acc[shift++];
            if (j == 6) b[i] = value;
            else A[j * 6 + i] = A[i * 6 + j] = value;
        }
}

static inline float u8at(const uint8_t* img, int W, int x, int y) { return (float)img[y * W + x]; }
static inline void so3_gradient(const uint8_t* img, int W, int x, int y, float* gx, float* gy) { /* reduce.cu:1015-1031 */
    const float actu = u8at(img, W, x, y);
    float back = u8at(img, W, x - 1, y), fore = u8at(img, W, x + 1, y);
    *gx = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
    back = u8at(img, W, x, y - 1); fore = u8at(img, W, x, y + 1);
    *gy = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
}

/* SO3Reduction::getProducts + so3Step, reduce.cu:999-1202.  A: 3x3 row-major, b: 3, residual = {sum r^2, inliers}. */
void mfo_so3_step(const uint8_t* lastImage, const uint8_t* nextImage, const float* imageBasis, const float* kinv,
                  const float* krlr, int W, int H, float* A, float* b, float* residual) {
    double acc[11];
    for (int k = 0; k < 11; ++k) acc[k] = 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const f3 p = f3_make((float)x, (float)y, 1.0f);
            const f3 wp = m33_mul(imageBasis, p);
            const int wx = f2i_rn(wp.x / wp.z), wy = f2i_rn(wp.y / wp.z);
            if (!(wx >= 1 && wx < W - 1 && wy >= 1 && wy < H - 1 && x >= 1 && x < W - 1 && y >= 1 && y < H - 1)) continue;
            float gnx, gny, glx, gly;
            so3_gradient(nextImage, W, wx, wy, &gnx, &gny);
            so3_gradient(lastImage, W, x, y, &glx, &gly);
            const float gx = (gnx + glx) / 2.0f, gy = (gny + gly) / 2.0f;
            const f3 point = m33_mul(kinv, p);
            const float z2 = point.z * point.z;
            const float a = krlr[0], b_ = krlr[1], c = krlr[2], d = krlr[3], e = krlr[4], f = krlr[5], g = krlr[6], h = krlr[7],
                        i_ = krlr[8];
            const float fy_ = (float)y, fx_ = (float)x;
            const f3 left = f3_make(((point.z * (d * gy + a * gx)) - (gy * g * fy_) - (gx * g * fx_)) / z2,
                                    ((point.z * (e * gy + b_ * gx)) - (gy * h * fy_) - (gx * h * fx_)) / z2,
                                    ((point.z * (f * gy + c * gx)) - (gy * i_ * fy_) - (gx * i_ * fx_)) / z2);
            const f3 jac = f3_cross(left, point);
            const float row[4] = {jac.x, jac.y, jac.z, -(u8at(nextImage, W, wx, wy) - u8at(lastImage, W, x, y))};
            int k = 0;
            for (int r = 0; r < 3; ++r)
                for (int cc = r; cc < 4; ++cc) acc[k++] += (double)(row[r] * row[cc]);
            acc[9] += (double)(row[3] * row[3]);
            acc[10] += 1.0;
        }
    int shift = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 4; ++j) {
            const float value = (float)acc[shift++];
            if (j == 3) b[i] = value;
            else A[j * 3 + i] = A[i * 3 + j] = value;
        }
    residual[0] = (float)acc[9];
    residual[1] = (float)acc[10];
}

/* Eigen::Matrix3f::ldlt().solve() stand-in (RGBDOdometry.cpp:313): float, diagonal pivoting */
void mfo_ldlt3f_solve(const float* Ain, const float* bin, float* x) {
    float A[9], b[3];
    int perm[3] = {0, 1, 2};
    memcpy(A, Ain, sizeof(A)); memcpy(b, bin, sizeof(b));
    for (int k = 0; k < 3; ++k) {
        int p = k;
        for (int i = k + 1; i < 3; ++i)
            if (fabsf(A[i * 3 + i]) > fabsf(A[p * 3 + p])) p = i;
        if (p != k) {
            for (int j = 0; j < 3; ++j) { float t = A[k * 3 + j]; A[k * 3 + j] = A[p * 3 + j]; A[p * 3 + j] = t; }
            for (int j = 0; j < 3; ++j) { float t = A[j * 3 + k]; A[j * 3 + k] = A[j * 3 + p]; A[j * 3 + p] = t; }
            { float t = b[k]; b[k] = b[p]; b[p] = t; }
            { int t = perm[k]; perm[k] = perm[p]; perm[p] = t; }
        }
        const float d = A[k * 3 + k];
        if (d == 0.f) continue;
        for (int i = k + 1; i < 3; ++i) {
            const float l = A[i * 3 + k] / d;
            for (int j = k + 1; j < 3; ++j) A[i * 3 + j] -= l * A[k * 3 + j];
            A[i * 3 + k] = l;
        }
    }
    float y[3], z[3];
    for (int i = 0; i < 3; ++i) {
        float s = b[i];
        for (int j = 0; j < i; ++j) s -= A[i * 3 + j] * y[j];
        y[i] = s;
    }
    for (int i = 0; i < 3; ++i) y[i] = (A[i * 3 + i] != 0.f) ? y[i] / A[i * 3 + i] : 0.f;
    for (int i = 2; i >= 0; --i) {
        float s = y[i];
        for (int j = i + 1; j < 3; ++j) s -= A[j * 3 + i] * z[j];
        z[i] = s;
    }
    for (int i = 0; i < 3; ++i) x[perm[i]] = z[i];
}

static void m33d_mul(const double* a, const double* b, double* out) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out[r * 3 + c] = a[r * 3 + 0] * b[0 * 3 + c] + a[r * 3 + 1] * b[1 * 3 + c] + a[r * 3 + 2] * b[2 * 3 + c];
}
static void m33d_inverse(const double* m, double* inv) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02, id = 1.0 / det;
    inv[0] = c00 * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    inv[3] = c01 * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    inv[6] = c02 * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}
static void k_matrix(double* K, double* Kinv, float fx, float fy, float cx, float cy) {
    const double k[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
    memcpy(K, k, sizeof(k));
    const double ki[9] = {1.0 / fx, 0, -(double)cx / fx, 0, 1.0 / fy, -(double)cy / fy, 0, 0, 1};
    memcpy(Kinv, ki, sizeof(ki));
}

/* SO(3) pre-alignment, RGBDOdometry.cpp:264-324 (level-2 images and intrinsics); resultR row-major double out */
void mfo_so3_prealign(const uint8_t* lastNext2, const uint8_t* next2, int W2, int H2, float fx2, float fy2, float cx2,
                      float cy2, double* resultR, float* lastSO3Error, float* lastSO3Count, int* iterations_run) {
    double K[9], Kinv[9];
    k_matrix(K, Kinv, fx2, fy2, cx2, cy2);
    for (int k = 0; k < 9; ++k) resultR[k] = (k % 4 == 0) ? 1.0 : 0.0;
    float R_lr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float lastError = 3.4028234664e38f / 2, lastCount = 3.4028234664e38f / 2;
    double lastResultR[9];
    memcpy(lastResultR, resultR, sizeof(lastResultR));
    *lastSO3Error = 0; *lastSO3Count = 0;
    int it = 0;
    for (int i = 0; i < 10; ++i) {
        double tmp[9], hom[9], krl[9];
        m33d_mul(K, resultR, krl);
        m33d_mul(krl, Kinv, hom);
        (void)tmp;
        float imageBasis[9], kinv[9], krlr[9];
        for (int k = 0; k < 9; ++k) { imageBasis[k] = (float)hom[k]; kinv[k] = (float)Kinv[k]; krlr[k] = (float)krl[k]; }
        float jtj[9], jtr[3], residual[2];
        mfo_so3_step(lastNext2, next2, imageBasis, kinv, krlr, W2, H2, jtj, jtr, residual);
        ++it;
        *lastSO3Error = sqrtf(residual[0]) / residual[1];
        *lastSO3Count = residual[1];
        if (*lastSO3Error < lastError && fabsf(lastError - *lastSO3Count) < 0.001f) break;  /* sic: error vs count */
        else if (*lastSO3Error > lastError + 0.001f) {
            *lastSO3Error = lastError; *lastSO3Count = lastCount;
            memcpy(resultR, lastResultR, sizeof(lastResultR));
            break;
        }
        lastError = *lastSO3Error; lastCount = *lastSO3Count;
        memcpy(lastResultR, resultR, sizeof(lastResultR));
        float delta[3];
        mfo_ldlt3f_solve(jtj, jtr, delta);
        const double dd[3] = {delta[0], delta[1], delta[2]};
        double rotUpdate[9];
        mfo_rodrigues(dd, rotUpdate);
        float ru[9], nl[9];
        for (int k = 0; k < 9; ++k) ru[k] = (float)rotUpdate[k];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) nl[r * 3 + c] = ru[r * 3 + 0] * R_lr[0 * 3 + c] + ru[r * 3 + 1] * R_lr[1 * 3 + c] + ru[r * 3 + 2] * R_lr[2 * 3 + c];
        memcpy(R_lr, nl, sizeof(nl));
        for (int k = 0; k < 9; ++k) resultR[k] = R_lr[k];
    }
    if (iterations_run) *iterations_run = it;
}

/* RGBDOdometry::getIncrementalTransformation (RGBDOdometry.cpp:227-497), all branches.
 * in->lastDepth/nextDepth/lastImage/nextImage: 3-level pyramids (populateRGBDData); lastNextImage2: level-2 intensity
 * of the previous frame (so3).  Fills the derivative images / cloud scratch itself. */
void mfo_track_rgbd(const float* const curr_v[3], const float* const curr_n[3], const float* const prev_v[3],
                    const float* const prev_n[3], const mfo_rgbd_inputs* in, int W, int H, float fx, float fy, float cx,
                    float cy, const mfo_track_opts* o, float* R, float* t, float* out_inc16, mfo_track_stats* st) {
    const int icp = !o->rgbOnly && o->icpWeight > 0;
    const int rgb = o->rgbOnly || o->icpWeight < 100;
    float Rprev[9], tprev[3], Rcurr[9], tcurr[3], Rprev_inv[9];
    memcpy(Rprev, R, sizeof(Rprev)); memcpy(tprev, t, sizeof(tprev));
    memcpy(Rcurr, R, sizeof(Rcurr)); memcpy(tcurr, t, sizeof(tcurr));
    memset(st, 0, sizeof(*st));
    g_track_ill = 0; g_track_log_n = 0;
    const float sobelScale = (float)(1.0 / 8.0);      /* 1 / 2^sobelSize, RGBDOdometry.cpp:31-32 */
    const float maxDepthDeltaRGB = 0.07f;              /* :33 */
    const float minGrad[3] = {5.f, 3.f, 1.f};          /* :102-105 */
    int16_t* dIdx[3] = {0, 0, 0}; int16_t* dIdy[3] = {0, 0, 0};
    float* cloud = NULL; mfo_dataterm* corres = NULL;
    if (rgb) {
        for (int i = 0; i < 3; ++i) {
            const int lp = (W >> i) * (H >> i);
            dIdx[i] = (int16_t*)malloc(sizeof(int16_t) * lp); dIdy[i] = (int16_t*)malloc(sizeof(int16_t) * lp);
            mfo_derivative_images(in->nextImage[i], dIdx[i], dIdy[i], W >> i, H >> i);
        }
        cloud = (float*)malloc(sizeof(float) * 3 * W * H);
        corres = (mfo_dataterm*)malloc(sizeof(mfo_dataterm) * W * H);
    }
    double resultR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (o->so3)
        mfo_so3_prealign(in->lastNextImage2, in->nextImage[2], W >> 2, H >> 2, fx / 4, fy / 4, cx / 4, cy / 4, resultR,
                         &st->lastSO3Error, &st->lastSO3Count, &st->so3Iterations);
    int iterations[3];
    iterations[0] = o->fastOdom ? 3 : 10;
    iterations[1] = o->pyramid ? 5 : 0;
    iterations[2] = o->pyramid ? 4 : 0;
    m33_inverse(Rprev, Rprev_inv);
    double resultRt[16];
    for (int k = 0; k < 16; ++k) resultRt[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (o->so3)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) resultRt[r * 4 + c] = resultR[r * 3 + c];
    float trR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, trt[3] = {0, 0, 0};
    for (int i = 2; i >= 0; --i) {
        const int div = 1 << i;
        const float lfx = fx / div, lfy = fy / div, lcx = cx / div, lcy = cy / div;
        const int lw = W >> i, lh = H >> i;
        if (rgb) mfo_project_to_cloud(in->lastDepth[i], cloud, lw, lh, lfx, lfy, lcx, lcy);
        double K[9], Kinv[9];
        k_matrix(K, Kinv, lfx, lfy, lcx, lcy);
        st->lastRGBError = 3.4028234664e38f;
        for (int j = 0; j < iterations[i]; ++j) {
            /* Rt = resultRt.inverse() (rigid: general inverse of the 3x3 block, -Rinv * t) */
            double Rr[9], Ri[9], ti[3];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) Rr[r * 3 + c] = resultRt[r * 4 + c];
            m33d_inverse(Rr, Ri);
            for (int r = 0; r < 3; ++r)
                ti[r] = -(Ri[r * 3 + 0] * resultRt[3] + Ri[r * 3 + 1] * resultRt[7] + Ri[r * 3 + 2] * resultRt[11]);
            double KR[9], KRK[9];
            m33d_mul(K, Ri, KR);
            m33d_mul(KR, Kinv, KRK);
            float krkInv[9], kt[3];
            for (int k = 0; k < 9; ++k) krkInv[k] = (float)KRK[k];
            for (int r = 0; r < 3; ++r) kt[r] = (float)(K[r * 3 + 0] * ti[0] + K[r * 3 + 1] * ti[1] + K[r * 3 + 2] * ti[2]);
            int32_t sigma = 0, rgbSize = 0;
            if (rgb)
                mfo_rgb_residual((float)(pow(minGrad[i], 2.0) / pow(sobelScale, 2.0)), dIdx[i], dIdy[i], in->lastDepth[i],
                                 in->nextDepth[i], in->lastImage[i], in->nextImage[i], corres, maxDepthDeltaRGB, kt, krkInv,
                                 lw, lh, &sigma, &rgbSize);
            const float tmpError = (float)(sqrt((double)sigma) / (double)rgbSize);
            float sigmaVal = (tmpError == 0) ? 1.f : (float)rgbSize;
            if (o->rgbOnly && tmpError > st->lastRGBError) break;
            st->lastRGBError = tmpError;
            st->lastRGBCount = (float)rgbSize;
            if (o->rgbOnly) sigmaVal = -1;
            float A_icp[36], b_icp[6], residual[2] = {0, 0};
            memset(A_icp, 0, sizeof(A_icp)); memset(b_icp, 0, sizeof(b_icp));
            if (icp) {
                mfo_icp_step(Rcurr, tcurr, curr_v[i], curr_n[i], Rprev_inv, tprev, lfx, lfy, lcx, lcy, prev_v[i], prev_n[i],
                             o->distThresh, o->angleThresh, lw, lh, A_icp, b_icp, residual);
                st->lastICPError = sqrtf(residual[0]) / residual[1];
                st->lastICPCount = residual[1];
                track_log_push(A_icp, b_icp, residual);
            }
            float A_rgbd[36], b_rgbd[6];
            memset(A_rgbd, 0, sizeof(A_rgbd)); memset(b_rgbd, 0, sizeof(b_rgbd));
            if (rgb) mfo_rgb_step(corres, sigmaVal, cloud, lfx, lfy, dIdx[i], dIdy[i], sobelScale, lw, lh, A_rgbd, b_rgbd);
            double dA[36], db[6], x[6];
            if (icp && rgb) {
                const double w = o->icpWeight;
                for (int k = 0; k < 36; ++k) dA[k] = (double)A_rgbd[k] + w * w * (double)A_icp[k];
                for (int k = 0; k < 6; ++k) db[k] = (double)b_rgbd[k] + w * (double)b_icp[k];
            } else if (icp) {
                for (int k = 0; k < 36; ++k) dA[k] = A_icp[k];
                for (int k = 0; k < 6; ++k) db[k] = b_icp[k];
                g_track_ill += gn_system_ill(dA, residual[1]);   /* the geometric loop's stated domain (finding F4) */
            } else {
                for (int k = 0; k < 36; ++k) dA[k] = A_rgbd[k];
                for (int k = 0; k < 6; ++k) db[k] = b_rgbd[k];
            }
            mfo_ldlt_solve(dA, db, x, 6);
            mfo_update_se3(resultRt, x);
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) trR[r * 3 + c] = (float)resultRt[r * 4 + c];
                trt[r] = (float)resultRt[r * 4 + 3];
            }
            float iR[9], it3[3];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) iR[r * 3 + c] = trR[c * 3 + r];
            f3 itv = m33_mul(iR, f3_make(trt[0], trt[1], trt[2]));
            it3[0] = -itv.x; it3[1] = -itv.y; it3[2] = -itv.z;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                    Rcurr[r * 3 + c] = Rprev[r * 3 + 0] * iR[0 * 3 + c] + Rprev[r * 3 + 1] * iR[1 * 3 + c] + Rprev[r * 3 + 2] * iR[2 * 3 + c];
            f3 tv = m33_mul(Rprev, f3_make(it3[0], it3[1], it3[2]));
            tcurr[0] = tv.x + tprev[0]; tcurr[1] = tv.y + tprev[1]; tcurr[2] = tv.z + tprev[2];
            st->iterationsRun++;
        }
    }
    if (rgb) {
        const float dx = tcurr[0] - tprev[0], dy = tcurr[1] - tprev[1], dz = tcurr[2] - tprev[2];
        if ((double)sqrtf(dx * dx + dy * dy + dz * dz) > 0.3) {  /* RGBDOdometry.cpp:477-481 */
            memcpy(Rcurr, Rprev, sizeof(Rprev)); memcpy(tcurr, tprev, sizeof(tprev));
            const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            memcpy(trR, I, sizeof(I)); trt[0] = trt[1] = trt[2] = 0;
            st->rejected = 1;
        }
    }
    memcpy(R, Rcurr, sizeof(Rcurr)); memcpy(t, tcurr, sizeof(tcurr));
    if (out_inc16) Rt_to_pose16(trR, trt, out_inc16);
    for (int i = 0; i < 3; ++i) { free(dIdx[i]); free(dIdy[i]); }
    free(cloud); free(corres);
}

/* RGBDOdometry::populateRGBDData (RGBDOdometry.cpp:187-204) without the (unused) mask pyramid.  v4: the vertex map that
 * initICPModel copied into vmaps_tmp (Q1: BOTH the "last" and the "next" depth come from it); img/channels: the image. */
void mfo_populate_rgbd(const float* v4, const uint8_t* img, int channels, int W, int H, float* depth[3], uint8_t* image[3]) {
    mfo_vertices_to_depth(v4, depth[0], W * H, 6.0f);  /* maxDepthRGB, RGBDOdometry.cpp:34 */
    for (int i = 0; i + 1 < 3; ++i) mfo_pyrdown_gauss_f(depth[i], depth[i + 1], W >> i, H >> i);
    mfo_image_to_intensity(img, channels, image[0], W * H);
    for (int i = 0; i + 1 < 3; ++i) mfo_pyrdown_gauss_u8(image[i], image[i + 1], W >> i, H >> i);
}

/* ---- per-context scratch for populateRGBDData + the per-model "lastNextImage" pyramid (RGBDOdometry.h:120-135) ---- */
typedef struct { float* lastDepth[3]; uint8_t* lastImage[3]; uint8_t* nextImage[3]; } rgbd_scratch;
static void pyr_u8_alloc(uint8_t* p[3], int W, int H) { for (int i = 0; i < 3; ++i) p[i] = (uint8_t*)calloc((size_t)(W >> i) * (H >> i), 1); }
static void pyr_u8_free(uint8_t* p[3]) { for (int i = 0; i < 3; ++i) { free(p[i]); p[i] = NULL; } }
static void rgbd_scratch_alloc(rgbd_scratch* r, int W, int H) {
    for (int i = 0; i < 3; ++i) r->lastDepth[i] = (float*)calloc((size_t)(W >> i) * (H >> i), sizeof(float));
    pyr_u8_alloc(r->lastImage, W, H); pyr_u8_alloc(r->nextImage, W, H);
}
static void rgbd_scratch_free(rgbd_scratch* r) {
    for (int i = 0; i < 3; ++i) free(r->lastDepth[i]);
    pyr_u8_free(r->lastImage); pyr_u8_free(r->nextImage);
}
/* RGBDOdometry::initFirstRGB, RGBDOdometry.cpp:216-225 */
static void first_rgb(const uint8_t* rgb, int W, int H, uint8_t* lastNext[3]) {
    mfo_image_to_intensity(rgb, 3, lastNext[0], W * H);
    for (int i = 0; i + 1 < 3; ++i) mfo_pyrdown_gauss_u8(lastNext[i], lastNext[i + 1], W >> i, H >> i);
}
/* Model::initICP's RGB half (Model.cpp:391-409) + getIncrementalTransformation + the lastNextImage swap (:483-487).
 * v4src / lastImg: what initICPModel / initRGBModel were given (prediction, or fill-in when doFillIn). */
static void track_model(const mfo_config* g, rgbd_scratch* rs, const float* const cv[3], const float* const cn[3],
                        const float* const pv[3], const float* const pn[3], const float* v4src, const uint8_t* lastImgRGBA,
                        const uint8_t* rgb, uint8_t* lastNext[3], float* R, float* t, float* inc16, mfo_track_stats* st) {
    const int W = g->W, H = g->H;
    mfo_track_opts o;
    o.pyramid = g->pyramid; o.fastOdom = g->fastOdom; o.so3 = g->so3; o.rgbOnly = g->rgbOnly; o.icpWeight = g->icpWeight;
    o.distThresh = 0.10f; o.angleThresh = sinf(20.f * 3.14159254f / 180.f); /* RGBDOdometry.h:35-36 */
    /* initRGBModel then initRGB: both read vmaps_tmp (Q1) -> nextDepth == lastDepth */
    mfo_populate_rgbd(v4src, lastImgRGBA, 4, W, H, rs->lastDepth, rs->lastImage);
    mfo_image_to_intensity(rgb, 3, rs->nextImage[0], W * H);
    for (int i = 0; i + 1 < 3; ++i) mfo_pyrdown_gauss_u8(rs->nextImage[i], rs->nextImage[i + 1], W >> i, H >> i);
    mfo_rgbd_inputs in;
    for (int i = 0; i < 3; ++i) {
        in.lastDepth[i] = rs->lastDepth[i]; in.nextDepth[i] = rs->lastDepth[i];
        in.lastImage[i] = rs->lastImage[i]; in.nextImage[i] = rs->nextImage[i];
    }
    in.lastNextImage2 = lastNext[2];
    mfo_track_rgbd(cv, cn, pv, pn, &in, W, H, g->fx, g->fy, g->cx, g->cy, &o, R, t, inc16, st);
    if (o.so3)
        for (int i = 0; i < 3; ++i) memcpy(lastNext[i], rs->nextImage[i], (size_t)(W >> i) * (H >> i));
}

/* ------------------------------------------------------------------------------------------------
 * surfel helpers (Core/Shaders/color_encoding.glsl, surfels.glsl)
 * ---------------------------------------------------------------------------------------------- */
float mfo_encode_color(float r, float g, float b) {
    int rgb = (int)roundf(r * 255.0f);
    rgb = (rgb << 8) + (int)roundf(g * 255.0f);
    rgb = (rgb << 8) + (int)roundf(b * 255.0f);
    return (float)rgb;
}
void mfo_decode_color(float c, float* rgb) {
    const int ci = (int)c;
    rgb[0] = (float)((ci >> 16) & 0xFF) / 255.0f;
    rgb[1] = (float)((ci >> 8) & 0xFF) / 255.0f;
    rgb[2] = (float)(ci & 0xFF) / 255.0f;
}
float mfo_get_radius(float depth, float norm_z, float fx, float fy) {
    /* cam.z = 1/fx, cam.w = 1/fy  => meanFocal = (fx' + fy')/2 with fx' = 1/(1/fx) */
    const float camz = 1.0f / fx, camw = 1.0f / fy;
    const float meanFocal = ((1.0f / fabsf(camz)) + (1.0f / fabsf(camw))) / 2.0f;
    const float sqrt2 = 1.41421356237f;
    const float radius = (depth / meanFocal) * sqrt2;
    float radius_n = radius / fabsf(norm_z);
    radius_n = fminf(2.0f * radius, radius_n);
    return radius_n;
}
/* ---- exp() and acos() of the surfel shaders (surfels.glsl:44, data.vert:167): GLSL leaves their last bits to the GPU vendor.
 * Both sides of the parity tests (oracle/mf_oracle.c and maskfusion_amd/csrc/mf_device.h) evaluate them with the SAME
 * sequence of individually rounded fp32 operations (Cephes-style range reduction + polynomial, <= 2 ulp), so that a confidence
 * or an angle test can never differ between them in the last bit -- which the life cycle of a surfel would amplify into a
 * different keep / merge decision a few frames later. ---- */
static inline float bits_to_float_(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float shader_exp(float x) {
    /* x <= 0 here (the argument is -(r/400)^2 / 0.72); valid for |x| < 87 */
    const float n = rintf(x * 1.44269504088896341f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float p = 1.9875691500E-4f;
    p = p * r + 1.3981999507E-3f;
    p = p * r + 8.3334519073E-3f;
    p = p * r + 4.1665795894E-2f;
    p = p * r + 1.6666665459E-1f;
    p = p * r + 5.0000001201E-1f;
    const float y = p * (r * r) + r + 1.0f;
    return y * bits_to_float_((unsigned)((int)n + 127) << 23);   /* exact scaling by 2^n */
}
static inline float shader_asin_core(float a) {
    const float z = a * a;
    float p = 4.2163199048E-2f;
    p = p * z + 2.4181311049E-2f;
    p = p * z + 4.5470025998E-2f;
    p = p * z + 7.4953002686E-2f;
    p = p * z + 1.6666752422E-1f;
    return p * z * a + a;
}
static inline float shader_acos(float x) {
    if (x > 0.5f) return 2.0f * shader_asin_core(sqrtf(0.5f * (1.0f - x)));
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * shader_asin_core(sqrtf(0.5f * (1.0f + x)));
    return 1.57079632679489661923f - shader_asin_core(x);   /* NaN falls through to here and stays NaN */
}

float mfo_shader_exp(float x) { return shader_exp(x); }
float mfo_shader_acos(float x) { return shader_acos(x); }

float mfo_confidence(float x, float y, float weighting, float cx, float cy) {
    const float maxRadDist = 400.f;        /* quirk Q6 */
    const float twoSigmaSquared = 0.72f;
    const float dx = x - cx, dy = y - cy;
    const float radialDist = sqrtf(dx * dx + dy * dy) / maxRadDist;
    return shader_exp(-(radialDist * radialDist) / twoSigmaSquared) * weighting;
}

/* nearest fetch with GL_CLAMP_TO_EDGE */
static inline float texf(const float* img, int W, int H, int x, int y) {
    return img[iclamp(y, 0, H - 1) * W + iclamp(x, 0, W - 1)];
}
/* geometry.glsl:21-26, float-coordinate overload; cam = (cx, cy, 1/fx, 1/fy) */
static inline f3 get_vertex_f(const float* depth, int W, int H, int px, int py, float x, float y, const mfo_cam* c) {
    const float z = texf(depth, W, H, px, py);
    return f3_make((x - c->cx) * z * (1.0f / c->fx), (y - c->cy) * z * (1.0f / c->fy), z);
}
/* geometry.glsl:28-40 central differences */
static inline f3 get_normal_central(const float* depth, const mfo_cam* c, int px, int py, float x, float y, f3 vPos) {
    const int W = c->W, H = c->H;
    const f3 xf = get_vertex_f(depth, W, H, px + 1, py, x + 1, y, c);
    const f3 xb = get_vertex_f(depth, W, H, px - 1, py, x - 1, y, c);
    const f3 yf = get_vertex_f(depth, W, H, px, py + 1, x, y + 1, c);
    const f3 yb = get_vertex_f(depth, W, H, px, py - 1, x, y - 1, c);
    const f3 del_x = f3_sub(f3_scale(f3_add(xb, vPos), 0.5f), f3_scale(f3_add(xf, vPos), 0.5f));
    const f3 del_y = f3_sub(f3_scale(f3_add(yb, vPos), 0.5f), f3_scale(f3_add(yf, vPos), 0.5f));
    return f3_glnormalize(f3_cross(del_x, del_y));
}
/* geometry.glsl:42-62 int overloads: forward differences */
static inline f3 get_normal_forward(const float* depth, const mfo_cam* c, int px, int py, f3 vPos) {
    const int W = c->W, H = c->H;
    const f3 vx = get_vertex_f(depth, W, H, px + 1, py, (float)(px + 1), (float)py, c);
    const f3 vy = get_vertex_f(depth, W, H, px, py + 1, (float)px, (float)(py + 1), c);
    return f3_glnormalize(f3_cross(f3_sub(vx, vPos), f3_sub(vy, vPos)));
}

/* ------------------------------------------------------------------------------------------------
 * a21: first-frame initialisation
 * ---------------------------------------------------------------------------------------------- */
int mfo_init_surfels(const mfo_cam* c, const uint8_t* rgb, const float* depthRaw, const float* depthF, int tick,
                     float maxDepth, float* surfels, int capacity) {
    const int W = c->W, H = c->H;
    int count = 0;
    for (int i = 0; i < W; ++i) {          /* column-major: FeedbackBuffer.cpp:44-50 */
        for (int j = 0; j < H; ++j) {
            const float x = (float)i + 0.5f, y = (float)j + 0.5f;
            const f3 vraw = get_vertex_f(depthRaw, W, H, i, j, x, y, c);
            if (vraw.z <= 0 || vraw.z > maxDepth) continue;  /* vertex_feedback.vert:53-60, .geom:35 */
            if (count >= capacity) return count;
            const f3 vfil = get_vertex_f(depthF, W, H, i, j, x, y, c);
            const f3 n = get_normal_central(depthF, c, i, j, x, y, vfil);
            float* s = surfels + (size_t)count * 12;
            s[0] = vraw.x; s[1] = vraw.y; s[2] = vraw.z;
            s[3] = mfo_confidence(x, y, 1.0f, c->cx, c->cy);
            const uint8_t* p = rgb + (size_t)(j * W + i) * 3;
            s[4] = (float)((p[0] << 16) + (p[1] << 8) + p[2]); /* encodeColor of c/255 texels */
            s[5] = 0.f;
            s[6] = 1.f;            /* init_unstable.vert:34 */
            s[7] = (float)tick;    /* vertex_feedback.vert:68 */
            s[8] = n.x; s[9] = n.y; s[10] = n.z;
            s[11] = mfo_get_radius(vfil.z, n.z, c->fx, c->fy);
            ++count;
        }
    }
    return count;
}

/* ------------------------------------------------------------------------------------------------
 * a13: index map (index_map.vert/.frag; ModelProjection.cpp:100-152)
 * Raster rule (documented): 1-px point -> texel floor(u), floor(v); z-test LESS on the float z, first
 * (lowest index) surfel wins ties.
 * ---------------------------------------------------------------------------------------------- */
/* z-test of surfels [i0, i1) in index order into zbuf / win (win: -1 = none): the serial loop of the pass */
static void predict_indices_range(const mfo_cam* c, const float* Ri, const float* ti, const float* surfels, int i0, int i1, int time,
                                  float maxDepth, int timeDelta, float* zbuf, int32_t* win) {
    const int W = c->W, H = c->H;
    for (int i = i0; i < i1; ++i) {
        const float* s = surfels + (size_t)i * 12;
        f3 h = m33_mul(Ri, f3_make(s[0], s[1], s[2]));
        h = f3_make(h.x + ti[0], h.y + ti[1], h.z + ti[2]);
        if (h.z > maxDepth || h.z <= 0 || (float)time - s[7] > (float)timeDelta) continue;
        const float u = ((c->fx * h.x) / h.z) + c->cx;
        const float v = ((c->fy * h.y) / h.z) + c->cy;
        if (!(u >= 0.f && u < (float)W && v >= 0.f && v < (float)H)) continue;
        const int px = (int)floorf(u), py = (int)floorf(v);
        const int p = py * W + px;
        if (!(h.z < zbuf[p])) continue;
        zbuf[p] = h.z;
        win[p] = i;
    }
}
/* how many threads the z-buffer passes split a big surfel buffer over (each gets a private z-buffer; see splat_zbuffer) */
static int zbuffer_threads(int count) {
    int T = 1;
#ifdef _OPENMP
    T = omp_get_max_threads();
#endif
    if (T > 16) T = 16;
    if (count < 400000 || T < 2) return 1;
    return T;
}
void mfo_predict_indices(const mfo_cam* c, const float* pose16, const float* surfels, int count, int time,
                         float maxDepth, int timeDelta, int32_t* index, float* vertConf, float* colorTime,
                         float* normRad) {
    const int W = c->W, H = c->H, P = W * H;
    (void)H;
    float R[9], t[3], Ri[9], ti[3];
    pose16_to_Rt(pose16, R, t);
    pose_inverse_Rt(R, t, Ri, ti);
    /* The z-test visits the surfels in index order; LESS keeps the first (lowest-index) surfel among equal depths.  A big buffer is cut into
     * T contiguous index ranges, each tested into a private z-buffer in index order, and the private buffers are merged in range order with
     * the same strict LESS: per pixel the winner is the smallest z and, among equal z, the lowest index -- the serial loop's result. */
    const int T = zbuffer_threads(count);
    float* zbuf = (float*)malloc(sizeof(float) * (size_t)P * T);
    int32_t* win = (int32_t*)malloc(sizeof(int32_t) * (size_t)P * T);
#pragma omp parallel for schedule(static) num_threads(T)
    for (int k = 0; k < T; ++k) {
        float* zb = zbuf + (size_t)k * P; int32_t* wn = win + (size_t)k * P;
        for (int i = 0; i < P; ++i) { zb[i] = INFINITY; wn[i] = -1; }
        const int i0 = (int)((long long)count * k / T), i1 = (int)((long long)count * (k + 1) / T);
        predict_indices_range(c, Ri, ti, surfels, i0, i1, time, maxDepth, timeDelta, zb, wn);
    }
#pragma omp parallel for schedule(static)
    for (int p = 0; p < P; ++p) {
        for (int k = 1; k < T; ++k)
            if (zbuf[(size_t)k * P + p] < zbuf[p]) { zbuf[p] = zbuf[(size_t)k * P + p]; win[p] = win[(size_t)k * P + p]; }
        /* the attachments of the winning fragment (index_map.frag); an untouched texel keeps the clear value 0 */
        const int i = win[p];
        if (i < 0) {
            index[p] = 0;
            memset(vertConf + (size_t)p * 4, 0, 4 * sizeof(float)); memset(colorTime + (size_t)p * 4, 0, 4 * sizeof(float));
            memset(normRad + (size_t)p * 4, 0, 4 * sizeof(float));
            continue;
        }
        const float* s = surfels + (size_t)i * 12;
        f3 h = m33_mul(Ri, f3_make(s[0], s[1], s[2]));
        h = f3_make(h.x + ti[0], h.y + ti[1], h.z + ti[2]);
        index[p] = i;
        vertConf[p * 4 + 0] = h.x; vertConf[p * 4 + 1] = h.y; vertConf[p * 4 + 2] = h.z; vertConf[p * 4 + 3] = s[3];
        memcpy(colorTime + p * 4, s + 4, 4 * sizeof(float));
        const f3 n = f3_glnormalize(m33_mul(Ri, f3_make(s[8], s[9], s[10])));
        normRad[p * 4 + 0] = n.x; normRad[p * 4 + 1] = n.y; normRad[p * 4 + 2] = n.z; normRad[p * 4 + 3] = s[11];
    }
    free(zbuf); free(win);
}

/* Window rule for data.vert:139-141 (documented oracle rule): pixel-centre offsets
 * {-1,-0.5,0,+0.5} map to texels {x-1, x, x, x+1}. */
static const int kWinPix[4] = {-1, 0, 0, 1};

/* ANALYSIS SWITCH (never on in the parity tests): a literal fp32 reading of the window loops of data.vert:139-141 and
 * copy_unstable.vert:85-86, `for (i = c - 2s; i < c + 2s; i += s)` with an fp32 induction variable.  In exact arithmetic the loop
 * makes 4 steps; in fp32 it makes 4 or 5 depending on the rounding of c.  With the switch on, the taps are the
 * values the fp32 loop visits and a tap's texel is floor(snap_1/256(i * size)) -- the texel rule of oracle/glsl_shim/mfgl.h, under which
 * the reference's own shader text (oracle/_ref/libmf_glsl.so) makes exactly these decisions (tests/test_glsl_pin.py), used only to measure how much
 * the ambiguity matters (tools/window_ambiguity.py). */
static int g_window_literal = 0;
void mfo_set_window_literal(int on) { g_window_literal = on; }
/* The clean pass (copy_unstable.vert) DEFAULTS to the literal reading: there every tap counts, the reference's shader text compiled
 * from its source makes exactly these decisions (tests/test_glsl_pin.py), and the device kernel implements the same
 * (window_slots_literal in mf_surfel.hip).  mfo_set_clean_literal(0) selects the exact-arithmetic 4 x 4 window (round-1 behaviour). */
static int g_clean_literal = 1;
void mfo_set_clean_literal(int on) { g_clean_literal = on; }
/* taps of one axis: centre coordinate c (normalised), size = cols or rows; returns the number of taps (<= 8) */
static int window_taps_literal(float c, float size, int* texels) {
    const float step = (1.0f / (size * 1.0f)) * 0.5f;          /* indexXStep, scale = FACTOR = 1 */
    const float half = (1.0f * step) * 2.0f;                   /* scale * indexXStep * windowMultiplier */
    const float end = c + half;
    int n = 0;
    for (float i = c - half; i < end && n < 8; i += step) {
        int t = (int)floorf(rintf(i * size * 256.0f) * (1.0f / 256.0f));   /* texel of a normalised coordinate: 1/256 fixed-point snap, then floor */
        t = t < 0 ? 0 : (t > (int)size - 1 ? (int)size - 1 : t);
        texels[n++] = t;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * a14 part 1: data association (data.vert; Model.cpp:466-581)
 * Candidate enumeration: quarter-rate pixels (x%2 == t%2 && y%2 == t%2) in column-major order:
 *   c = xi * nyc + yi, x = 2*xi + (t&1), y = 2*yi + (t&1), nxc = (W - (t&1) + 1)/2, nyc = (H - (t&1) + 1)/2.
 * ---------------------------------------------------------------------------------------------- */
void mfo_fuse_data(const mfo_cam* c, const float* pose16, const uint8_t* rgb, const float* depthRaw,
                   const float* depthF, const uint8_t* mask, int maskID, int time, float weighting, float maxDepth,
                   const int32_t* index, const float* vertConf, const float* normRad, uint8_t* cand_op,
                   int32_t* cand_best, float* cand_rec, int* n_cand) {
    const int W = c->W, H = c->H;
    const int par = time & 1;
    const int nxc = (W - par + 1) / 2, nyc = (H - par + 1) / 2;
    *n_cand = nxc * nyc;
    float R[9], t[3];
    pose16_to_Rt(pose16, R, t);
#pragma omp parallel for schedule(static)
    for (int xi = 0; xi < nxc; ++xi) {
        for (int yi = 0; yi < nyc; ++yi) {
            const int cidx = xi * nyc + yi;
            const int px = 2 * xi + par, py = 2 * yi + par;
            cand_op[cidx] = 0;
            cand_best[cidx] = 0;
            const float x = (float)px + 0.5f, y = (float)py + 0.5f;
            const f3 vLocal = get_vertex_f(depthRaw, W, H, px, py, x, y, c);
            if (mask[py * W + px] != maskID) continue;
            /* checkNeighbours: data.vert:53-72 */
            if (texf(depthRaw, W, H, px - 1, py) == 0 || texf(depthRaw, W, H, px, py - 1) == 0 ||
                texf(depthRaw, W, H, px + 1, py) == 0 || texf(depthRaw, W, H, px, py + 1) == 0)
                continue;
            if (!(vLocal.z > 0 && vLocal.z <= maxDepth)) continue;

            const f3 vGlobal3 = m33_mul(R, vLocal);
            const f3 vGlobal = f3_make(vGlobal3.x + t[0], vGlobal3.y + t[1], vGlobal3.z + t[2]);
            const f3 vF = get_vertex_f(depthF, W, H, px, py, x, y, c);
            const f3 nLocal = get_normal_central(depthF, c, px, py, x, y, vF);
            const f3 nGlobal = m33_mul(R, nLocal);
            const uint8_t* pc = rgb + (size_t)(py * W + px) * 3;

            float* rec = cand_rec + (size_t)cidx * 12;
            rec[0] = vGlobal.x; rec[1] = vGlobal.y; rec[2] = vGlobal.z;
            rec[3] = mfo_confidence(x, y, weighting, c->cx, c->cy);
            rec[4] = (float)((pc[0] << 16) + (pc[1] << 8) + pc[2]);
            rec[5] = 0.f;
            rec[6] = (float)time;
            rec[8] = nGlobal.x; rec[9] = nGlobal.y; rec[10] = nGlobal.z;
            rec[11] = mfo_get_radius(vF.z, nLocal.z, c->fx, c->fy);

            const float xl = (x - c->cx) * (1.0f / c->fx), yl = (y - c->cy) * (1.0f / c->fy);
            const float lambda = sqrtf(xl * xl + yl * yl + 1);
            const f3 ray = f3_make(xl, yl, 1);
            float bestDist = 1000;
            int best = 0, operation = 0;
            int wx[8], wy[8], nwx = 4, nwy = 4;
            for (int a = 0; a < 4; ++a) { wx[a] = iclamp(px + kWinPix[a], 0, W - 1); wy[a] = iclamp(py + kWinPix[a], 0, H - 1); }
            if (g_window_literal) {   /* texcoord as FeedbackBuffer.cpp:44-50 / Model.cpp build the uv buffer */
                nwx = window_taps_literal((float)((double)((float)px / (float)W) + 1.0 / (2.0 * (double)(float)W)), (float)W, wx);
                nwy = window_taps_literal((float)((double)((float)py / (float)H) + 1.0 / (2.0 * (double)(float)H)), (float)H, wy);
            }
            for (int a = 0; a < nwx; ++a) {
                for (int b = 0; b < nwy; ++b) {
                    const int tx = wx[a], ty = wy[b];
                    const int tp = ty * W + tx;
                    const int current = index[tp];
                    if (current > 0) {
                        const float* vc = vertConf + tp * 4;
                        const float zdiff = vc[2] - vLocal.z;
                        if (fabsf(zdiff * lambda) < 0.05f) {
                            const float dist = f3_norm(f3_cross(ray, f3_make(vc[0], vc[1], vc[2])));
                            const float* nr = normRad + tp * 4;
                            const f3 nn = f3_make(nr[0], nr[1], nr[2]);
                            const float ang = shader_acos(f3_dot(nn, nLocal) / (f3_norm(nn) * f3_norm(nLocal)));
                            if (dist < bestDist && (fabsf(nr[2]) < 0.75f || fabsf(ang) < 0.5f)) {
                                operation = 1; bestDist = dist; best = current;
                            }
                        }
                    }
                }
            }
            if (operation == 1) { cand_op[cidx] = 1; cand_best[cidx] = best; rec[7] = -1.f; }
            else { cand_op[cidx] = 2; rec[7] = -2.f; }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * a14 part 2: update (update.vert; Model.cpp:583-646).  First writer in candidate order wins.
 * ---------------------------------------------------------------------------------------------- */
void mfo_fuse_update(const float* src, float* dst, int count, int time, const uint8_t* cand_op,
                     const int32_t* cand_best, const float* cand_rec, int n_cand) {
    int32_t* first = (int32_t*)malloc(sizeof(int32_t) * (count > 0 ? count : 1));
    for (int i = 0; i < count; ++i) first[i] = -1;
    for (int c = 0; c < n_cand; ++c)
        if (cand_op[c] == 1) {
            const int b = cand_best[c];
            if (b >= 0 && b < count && first[b] < 0) first[b] = c;
        }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < count; ++i) {
        const float* s = src + (size_t)i * 12;
        float* d = dst + (size_t)i * 12;
        memcpy(d, s, 12 * sizeof(float));
        if (first[i] < 0) continue;
        const float* m = cand_rec + (size_t)first[i] * 12;
        const float c_k = s[3], a = m[3];
        if (m[11] < (1.0f + 0.5f) * s[11]) {
            for (int k = 0; k < 3; ++k) d[k] = ((c_k * s[k]) + (a * m[k])) / (c_k + a);
            d[3] = c_k + a;
            float oc[3], nc[3];
            mfo_decode_color(s[4], oc);
            mfo_decode_color(m[4], nc);
            float avg[3];
            for (int k = 0; k < 3; ++k) avg[k] = ((c_k * oc[k]) + (a * nc[k])) / (c_k + a);
            d[4] = mfo_encode_color(avg[0], avg[1], avg[2]);
            d[5] = s[5]; d[6] = s[6]; d[7] = (float)time;
            float nr[4];
            for (int k = 0; k < 4; ++k) nr[k] = ((c_k * s[8 + k]) + (a * m[8 + k])) / (c_k + a);
            const f3 nn = f3_glnormalize(f3_make(nr[0], nr[1], nr[2]));
            d[8] = nn.x; d[9] = nn.y; d[10] = nn.z; d[11] = nr[3];
        } else {
            d[3] = c_k + a;
            d[7] = (float)time;
        }
    }
    free(first);
}

/* ------------------------------------------------------------------------------------------------
 * a16: clean (copy_unstable.vert:53-157, .geom; Model.cpp:649-772); deformation block inert (nodes == 0).
 * ---------------------------------------------------------------------------------------------- */
static int clean_one(const mfo_cam* c, const float* Ri, const float* ti, const float* in, float* out, int time,
                     int timeDelta, float confThreshold, float outlierCoeff, int maskID, const int32_t* index,
                     const float* vertConf, const float* colorTime, const float* depthF, const uint8_t* mask) {
    const int W = c->W, H = c->H;
    memcpy(out, in, 12 * sizeof(float));
    int test = 1;
    f3 lp = m33_mul(Ri, f3_make(in[0], in[1], in[2]));
    lp = f3_make(lp.x + ti[0], lp.y + ti[1], lp.z + ti[2]);
    const float x = ((c->fx * lp.x) / lp.z) + c->cx;
    const float y = ((c->fy * lp.y) / lp.z) + c->cy;
    const f3 ln = f3_glnormalize(m33_mul(Ri, f3_make(in[8], in[9], in[10])));
    int count = 0, zCount = 0;
    if ((float)time - in[7] < (float)timeDelta && lp.z > 0 && x > 0 && y > 0 && x < (float)W && y < (float)H) {
        static const float off[4] = {-1.0f, -0.5f, 0.0f, 0.5f};
        int wx[8], wy[8], nwx = 4, nwy = 4;
        for (int a = 0; a < 4; ++a) { wx[a] = iclamp((int)floorf(x + off[a]), 0, W - 1); wy[a] = iclamp((int)floorf(y + off[a]), 0, H - 1); }
        if (g_window_literal || g_clean_literal) { nwx = window_taps_literal(x / (float)W, (float)W, wx); nwy = window_taps_literal(y / (float)H, (float)H, wy); }
        for (int a = 0; a < nwx; ++a) {
            for (int b = 0; b < nwy; ++b) {
                const int tx = wx[a], ty = wy[b];
                const int tp = ty * W + tx;
                if (index[tp] > 0) {
                    const float* vc = vertConf + tp * 4;
                    const float* ct = colorTime + tp * 4;
                    const float dx = vc[0] - lp.x, dy = vc[1] - lp.y;
                    if (ct[2] < in[6] && vc[3] > confThreshold && vc[2] > lp.z && vc[2] - lp.z < 0.01f &&
                        sqrtf(dx * dx + dy * dy) < in[11] * 1.4f)
                        count++;
                    if (ct[3] == (float)time && vc[3] > confThreshold && vc[2] > lp.z && vc[2] - lp.z > 0.01f &&
                        fabsf(ln.z) > 0.85f)
                        zCount++;
                }
            }
        }
    }
    if (count > 8 || zCount > 4) test = 0;
    if (out[7] == -2.f) out[7] = (float)time;
    if (out[7] == -1.f || (((float)time - out[7]) > 20 && out[3] < confThreshold)) test = 0;
    if (out[7] > 0 && (float)time - out[7] > (float)timeDelta) test = 1;

    /* mask-disagreement decay (copy_unstable.vert:139-156); nearest fetch, clamp-to-edge, NaN -> texel 0 */
    int fx_ = isnan(x) ? 0 : iclamp((int)fminf(fmaxf(floorf(x), -1.f), (float)W), 0, W - 1);
    int fy_ = isnan(y) ? 0 : iclamp((int)fminf(fmaxf(floorf(y), -1.f), (float)H), 0, H - 1);
    const float wDepth = depthF[fy_ * W + fx_];
    const int maskValue = mask[fy_ * W + fx_];
    if (maskValue != maskID && maskValue < 255 && (wDepth > lp.z - 0.05f && wDepth < lp.z + 0.05f)) {
        const float k = 0.5f + 0.5f * (1 - outlierCoeff / 10.0f);
        if (maskValue == 0) out[3] *= k;
        else if (maskID == 0) out[3] *= 0.25f * k;
        else out[3] *= k;
    }
    return test;
}

int mfo_clean(const mfo_cam* c, const float* pose16, const float* src, int count, const uint8_t* cand_op,
              const float* cand_rec, int n_cand, int time, int timeDelta, float confThreshold, float maxDepth,
              float outlierCoeff, int maskID, const int32_t* index, const float* vertConf, const float* colorTime,
              const float* normRad, const float* depthF, const uint8_t* mask, float* dst, int capacity) {
    (void)maxDepth; (void)normRad; /* uniforms set but unused by the live part of the shader */
    float R[9], t[3], Ri[9], ti[3];
    pose16_to_Rt(pose16, R, t);
    pose_inverse_Rt(R, t, Ri, ti);
    const int total = count + n_cand;
    uint8_t* keep = (uint8_t*)malloc((size_t)(total > 0 ? total : 1));
    float* tmp = (float*)malloc(sizeof(float) * 12 * (size_t)(total > 0 ? total : 1));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < total; ++i) {
        const float* in;
        if (i < count) in = src + (size_t)i * 12;
        else {
            const int cidx = i - count;
            if (cand_op[cidx] != 2) { keep[i] = 0; continue; } /* op==1 records carry w=-1 and are dropped */
            in = cand_rec + (size_t)cidx * 12;
        }
        keep[i] = (uint8_t)clean_one(c, Ri, ti, in, tmp + (size_t)i * 12, time, timeDelta, confThreshold,
                                     outlierCoeff, maskID, index, vertConf, colorTime, depthF, mask);
    }
    int n = 0;
    for (int i = 0; i < total && n < capacity; ++i)
        if (keep[i]) { memcpy(dst + (size_t)n * 12, tmp + (size_t)i * 12, 12 * sizeof(float)); ++n; }
    free(keep); free(tmp);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * a17: splat prediction (splat.vert:54-88, combo_splat.frag:37-65; ModelProjection.cpp:187-268)
 * Raster rule (documented): sprite side s centred on the projected centre (u,v) covers pixel (px,py) iff
 * u - s/2 <= px+0.5 < u + s/2 (same in y); z-test LESS on the corrected z; first surfel wins ties.
 * Sprites wider than MFO_MAX_SPRITE px are clamped (GL_POINT_SIZE_RANGE stand-in).
 * ---------------------------------------------------------------------------------------------- */
#define MFO_MAX_SPRITE 64.0f
/* z-buffered splat of one surfel buffer.  zbuf / winner (surfel index, -1 = none) / owner (caller tag per pixel, may be
 * NULL) are in-out so that several models can share one z-buffer (GlobalProjection). */
static void splat_zbuffer_range(const mfo_cam* c, const float* pose16, const float* surfels, int i0, int i1, float maxDepth,
                          float confThreshold, int time, int maxTime, int timeDelta, float* zbuf, int32_t* winner,
                          uint8_t* owner, uint8_t tag) {
    const int W = c->W, H = c->H;
    float R[9], t[3], Ri[9], ti[3];
    pose16_to_Rt(pose16, R, t);
    pose_inverse_Rt(R, t, Ri, ti);
    for (int i = i0; i < i1; ++i) {
        const float* s = surfels + (size_t)i * 12;
        f3 h = m33_mul(Ri, f3_make(s[0], s[1], s[2]));
        h = f3_make(h.x + ti[0], h.y + ti[1], h.z + ti[2]);
        if (h.z > maxDepth || h.z < 0 || s[3] < confThreshold || (float)time - s[7] > (float)timeDelta ||
            s[7] > (float)maxTime)
            continue;
        const float u = ((c->fx * h.x) / h.z) + c->cx, v = ((c->fy * h.y) / h.z) + c->cy;
        if (!(u >= 0.f && u <= (float)W && v >= 0.f && v <= (float)H)) continue; /* point clipped by centre */
        const f3 n = f3_glnormalize(m33_mul(Ri, f3_make(s[8], s[9], s[10])));
        const float rad = s[11];
        const f3 x1 = f3_scale(f3_glnormalize(f3_make(n.y - n.z, -n.x, n.x)), rad * 1.41421356f);
        const f3 y1 = f3_cross(n, x1);
        float xs0 = INFINITY, xs1 = -INFINITY, ys0 = INFINITY, ys1 = -INFINITY;
        const f3 corners[4] = {f3_add(h, x1), f3_add(h, y1), f3_sub(h, y1), f3_sub(h, x1)};
        for (int k = 0; k < 4; ++k) {
            const float pxk = ((c->fx * corners[k].x) / corners[k].z) + c->cx;
            const float pyk = ((c->fy * corners[k].y) / corners[k].z) + c->cy;
            xs0 = fminf(xs0, pxk); xs1 = fmaxf(xs1, pxk);
            ys0 = fminf(ys0, pyk); ys1 = fmaxf(ys1, pyk);
        }
        float size = fmaxf(0.f, fmaxf(fabsf(xs1 - xs0), fabsf(ys1 - ys0)));
        if (!(size > 0.f)) continue; /* also drops NaN sizes */
        /* gl_PointSize is clamped to the point size range (OpenGL 3.3 core, 3.4 "Points"): never below 1 px on the reference's
         * platform (ALIASED_POINT_SIZE_RANGE = [1, 2047] on NVIDIA); the upper end is capped at MFO_MAX_SPRITE here */
        size = fminf(fmaxf(size, 1.0f), MFO_MAX_SPRITE);
        const float half = size * 0.5f;
        const int px0 = imax(0, (int)ceilf(u - half - 0.5f)), px1 = imin(W - 1, (int)ceilf(u + half - 0.5f) - 1);
        const int py0 = imax(0, (int)ceilf(v - half - 0.5f)), py1 = imin(H - 1, (int)ceilf(v + half - 0.5f) - 1);
        const float sqrRad = rad * rad;
        const float pn = f3_dot(h, n);
        for (int py = py0; py <= py1; ++py) {
            for (int px = px0; px <= px1; ++px) {
                const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
                const f3 l = f3_glnormalize(f3_make((fcx - c->cx) / c->fx, (fcy - c->cy) / c->fy, 1.0f));
                const f3 cp = f3_scale(l, pn / f3_dot(l, n));
                const f3 diff = f3_sub(cp, h);
                if (!(f3_dot(diff, diff) <= sqrRad)) continue; /* discard if > (NaN discards too) */
                const float z = cp.z;
                if (!(z > 0.f)) continue;
                const int p = py * W + px;
                if (!(z < zbuf[p])) continue;
                zbuf[p] = z;
                winner[p] = i;
                if (owner) owner[p] = tag;
            }
        }
    }
}

/* The whole buffer.  A big one is cut into contiguous index ranges, each splatted in index order into a private z-buffer, and the private
 * buffers are merged into the caller's in range order with the same strict LESS -- per pixel the smallest z wins and, among equal z, the
 * earlier surfel (and the earlier model: the caller's buffer holds the models drawn before this one): the serial loop's result. */
static void splat_zbuffer(const mfo_cam* c, const float* pose16, const float* surfels, int count, float maxDepth,
                          float confThreshold, int time, int maxTime, int timeDelta, float* zbuf, int32_t* winner,
                          uint8_t* owner, uint8_t tag) {
    const int T = zbuffer_threads(count);
    if (T == 1) {
        splat_zbuffer_range(c, pose16, surfels, 0, count, maxDepth, confThreshold, time, maxTime, timeDelta, zbuf, winner, owner, tag);
        return;
    }
    const int P = c->W * c->H;
    float* pz = (float*)malloc(sizeof(float) * (size_t)P * T);
    int32_t* pw = (int32_t*)malloc(sizeof(int32_t) * (size_t)P * T);
#pragma omp parallel for schedule(static) num_threads(T)
    for (int k = 0; k < T; ++k) {
        float* zb = pz + (size_t)k * P; int32_t* wn = pw + (size_t)k * P;
        for (int i = 0; i < P; ++i) { zb[i] = INFINITY; wn[i] = -1; }
        const int i0 = (int)((long long)count * k / T), i1 = (int)((long long)count * (k + 1) / T);
        splat_zbuffer_range(c, pose16, surfels, i0, i1, maxDepth, confThreshold, time, maxTime, timeDelta, zb, wn, NULL, 0);
    }
#pragma omp parallel for schedule(static)
    for (int p = 0; p < P; ++p)
        for (int k = 0; k < T; ++k)
            if (pz[(size_t)k * P + p] < zbuf[p]) {
                zbuf[p] = pz[(size_t)k * P + p]; winner[p] = pw[(size_t)k * P + p];
                if (owner) owner[p] = tag;
            }
    free(pz); free(pw);
}

void mfo_combined_predict(const mfo_cam* c, const float* pose16, const float* surfels, int count, float maxDepth,
                          float confThreshold, int time, int maxTime, int timeDelta, uint8_t* image,
                          float* vertexConf, float* normalRad, uint16_t* timeMap) {
    const int W = c->W, H = c->H, P = W * H;
    float R[9], t[3], Ri[9], ti[3];
    pose16_to_Rt(pose16, R, t);
    pose_inverse_Rt(R, t, Ri, ti);
    float* zbuf = (float*)malloc(sizeof(float) * P);
    int32_t* winner = (int32_t*)malloc(sizeof(int32_t) * P);
    for (int i = 0; i < P; ++i) { zbuf[i] = INFINITY; winner[i] = -1; }
    splat_zbuffer(c, pose16, surfels, count, maxDepth, confThreshold, time, maxTime, timeDelta, zbuf, winner, NULL, 0);
    memset(image, 0, (size_t)P * 4);
    memset(vertexConf, 0, sizeof(float) * 4 * P);
    memset(normalRad, 0, sizeof(float) * 4 * P);
    memset(timeMap, 0, sizeof(uint16_t) * P);
    for (int p = 0; p < P; ++p) {
        if (winner[p] < 0) continue;
        const float* s = surfels + (size_t)winner[p] * 12;
        const int px = p % W, py = p / W;
        const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
        const float z = zbuf[p];
        const f3 n = f3_glnormalize(m33_mul(Ri, f3_make(s[8], s[9], s[10])));
        const int ci = (int)s[4];
        image[p * 4 + 0] = (uint8_t)((ci >> 16) & 0xFF);
        image[p * 4 + 1] = (uint8_t)((ci >> 8) & 0xFF);
        image[p * 4 + 2] = (uint8_t)(ci & 0xFF);
        image[p * 4 + 3] = 255;
        vertexConf[p * 4 + 0] = (fcx - c->cx) * z * (1.f / c->fx);   /* combo_splat.frag:56 */
        vertexConf[p * 4 + 1] = (fcy - c->cy) * z * (1.f / c->fy);
        vertexConf[p * 4 + 2] = z;
        vertexConf[p * 4 + 3] = s[3];
        normalRad[p * 4 + 0] = n.x; normalRad[p * 4 + 1] = n.y; normalRad[p * 4 + 2] = n.z;
        normalRad[p * 4 + 3] = s[11];
        timeMap[p] = (uint16_t)(unsigned)s[6];
    }
    free(zbuf); free(winner);
}

/* GlobalProjection::project + downloadDirect (Core/Model/GlobalProjection.cpp:43-114): every model splatted into one
 * z-buffer with the fixed surfel-confidence threshold 12 (:61); earlier models in the list win depth ties. */
void mfo_global_projection(const mfo_cam* c, const mfo_model_view* models, int n_models, int time, int maxTime,
                           int timeDelta, float depthCutoff, uint8_t* ids) {
    const int P = c->W * c->H;
    float* zbuf = (float*)malloc(sizeof(float) * P);
    int32_t* winner = (int32_t*)malloc(sizeof(int32_t) * P);
    for (int i = 0; i < P; ++i) { zbuf[i] = INFINITY; winner[i] = -1; }
    memset(ids, 0, P);
    for (int m = 0; m < n_models; ++m)
        splat_zbuffer(c, models[m].pose16, models[m].surfels, models[m].count, depthCutoff, 12.0f, time, maxTime,
                      timeDelta, zbuf, winner, ids, (uint8_t)models[m].id);
    free(zbuf); free(winner);
}

/* ------------------------------------------------------------------------------------------------
 * a18: fill-in (fill_vertex.frag:37-53, fill_normal.frag:34-50, fill_rgb.frag:29-37; FillIn.cpp)
 * ---------------------------------------------------------------------------------------------- */
void mfo_fill_in(const mfo_cam* c, const uint8_t* predImage, const float* predVertex, const float* predNormal,
                 const uint8_t* rawRgb, const float* rawDepth, int passthrough, uint8_t* fillImage,
                 float* fillVertex, float* fillNormal) {
    const int W = c->W, H = c->H;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            const int p = y * W + x;
            if (predVertex[p * 4 + 2] == 0 || (passthrough & 1)) {   /* bit 0: `lost` (all three passes); bit 1: frameToFrameRGB (image pass only, Model.cpp:979-981) */
                const float z = rawDepth[p];
                fillVertex[p * 4 + 0] = ((float)x - c->cx) * z * (1.0f / c->fx);
                fillVertex[p * 4 + 1] = ((float)y - c->cy) * z * (1.0f / c->fy);
                fillVertex[p * 4 + 2] = z;
                fillVertex[p * 4 + 3] = 1.f;
            } else memcpy(fillVertex + p * 4, predVertex + p * 4, 4 * sizeof(float));
            if (predNormal[p * 4 + 2] == 0 || (passthrough & 1)) {
                const f3 vp = get_vertex_f(rawDepth, W, H, x, y, (float)x, (float)y, c);
                const f3 n = get_normal_forward(rawDepth, c, x, y, vp);
                fillNormal[p * 4 + 0] = n.x; fillNormal[p * 4 + 1] = n.y; fillNormal[p * 4 + 2] = n.z;
                fillNormal[p * 4 + 3] = 1.f;
            } else memcpy(fillNormal + p * 4, predNormal + p * 4, 4 * sizeof(float));
            if ((predImage[p * 4] == 0 && predImage[p * 4 + 1] == 0 && predImage[p * 4 + 2] == 0) || passthrough) {
                fillImage[p * 4 + 0] = rawRgb[p * 3 + 0]; fillImage[p * 4 + 1] = rawRgb[p * 3 + 1];
                fillImage[p * 4 + 2] = rawRgb[p * 3 + 2]; fillImage[p * 4 + 3] = 255;
            } else memcpy(fillImage + p * 4, predImage + p * 4, 4);
        }
    }
}

/* MaskFusion.cpp:630-648 + GPUResize::image (nearest; the output texel (i,j) of the W/20 x H/20 target samples
 * source texel (20i+10, 20j+10)). */
int mfo_requires_fill_in(const uint8_t* predImage, int W, int H, float ratio) {
    const int cs = 20, rw = W / cs, rh = H / cs;
    int sum = 0;
    for (int j = 0; j < rh; ++j)
        for (int i = 0; i < rw; ++i) {
            const uint8_t* p = predImage + (size_t)((j * cs + cs / 2) * W + (i * cs + cs / 2)) * 4;
            sum += (p[0] > 0 && p[1] > 0 && p[2] > 0);
        }
    return (float)sum / (float)(rw * rh) < ratio;
}

/* Model.cpp:449-464; rodrigues2 :891-932 (the SVD re-orthonormalisation U V^T is replaced by Newton polar
 * iterations, identical to rounding for the near-rotations that occur). */
/* Literal mode (finding F5): U V^T, its off-diagonal differences and its trace are FLOAT in the reference (Eigen::Matrix3f,
 * Model.cpp:892-901), so cos(theta) is quantised in steps of 1.2e-7 and theta = acos(c) in steps of ~4.9e-4 rad near 0.  ON by default
 * since round 3 (the device's matching mode "literalFusionWeight" is its default too); mfo_set_weight_literal(0) selects the accurate
 * double log map of rounds 1-2.  tests/test_weight_pin.py compares both modes with the reference's compiled text. */
static int g_weight_literal = 1;
void mfo_set_weight_literal(int on) { g_weight_literal = on; }
static void rodrigues2(const float* Rin, double* r) {
    double R[9], Rn[9];
    for (int k = 0; k < 9; ++k) R[k] = Rin[k];
    for (int it = 0; it < 4; ++it) { /* R <- (R + R^-T)/2 */
        const double c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
        const double det = R[0] * c00 + R[1] * c01 + R[2] * c02;
        double cof[9] = {c00, c01, c02,
                         R[2] * R[7] - R[1] * R[8], R[0] * R[8] - R[2] * R[6], R[1] * R[6] - R[0] * R[7],
                         R[1] * R[5] - R[2] * R[4], R[2] * R[3] - R[0] * R[5], R[0] * R[4] - R[1] * R[3]};
        const double idet = 1.0 / det;   /* one division per iteration (the device evaluates the same expression: bit-identical) */
        for (int k = 0; k < 9; ++k) Rn[k] = 0.5 * (R[k] + cof[k] * idet); /* cof/det = R^-T */
        memcpy(R, Rn, sizeof(R));
    }
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double cth = (R[0] + R[4] + R[8] - 1) * 0.5;
    if (g_weight_literal) {
        float Rf[9];
        for (int k = 0; k < 9; ++k) { Rf[k] = (float)R[k]; R[k] = (double)Rf[k]; }     /* the branches below read the float matrix too */
        rx = (double)(Rf[7] - Rf[5]); ry = (double)(Rf[2] - Rf[6]); rz = (double)(Rf[3] - Rf[1]);
        const float tr = (Rf[0] + Rf[4]) + Rf[8];
        cth = (double)(tr - 1.0f) * 0.5;
    }
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    cth = cth > 1. ? 1. : cth < -1. ? -1. : cth;
    double theta = acos(cth);
    if (s < 1e-5) {
        if (cth > 0) rx = ry = rz = 0;
        else {
            double tt = (R[0] + 1) * 0.5;
            rx = sqrt(fmax(tt, 0.0));
            tt = (R[4] + 1) * 0.5;
            ry = sqrt(fmax(tt, 0.0)) * (R[1] < 0 ? -1.0 : 1.0);
            tt = (R[8] + 1) * 0.5;
            rz = sqrt(fmax(tt, 0.0)) * (R[2] < 0 ? -1.0 : 1.0);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        const double vth = 1 / (2 * s) * theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    r[0] = rx; r[1] = ry; r[2] = rz;
}

float mfo_fusion_weight(const float* pose16, const float* lastPose16, float weightMultiplier) {
    float R[9], t[3], Rl[9], tl[3], Ri[9], ti[3], Rd[9];
    pose16_to_Rt(pose16, R, t);
    pose16_to_Rt(lastPose16, Rl, tl);
    pose_inverse_Rt(R, t, Ri, ti);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            Rd[r * 3 + c] = Ri[r * 3] * Rl[c] + Ri[r * 3 + 1] * Rl[3 + c] + Ri[r * 3 + 2] * Rl[6 + c];
    f3 td = m33_mul(Ri, f3_make(tl[0], tl[1], tl[2]));
    td = f3_make(td.x + ti[0], td.y + ti[1], td.z + ti[2]);
    double rv[3];
    rodrigues2(Rd, rv);
    const float rn = (float)sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    float weighting = fmaxf(f3_norm(td), rn);
    const float largest = 0.01f, minWeight = 0.5f;
    if (weighting > largest) weighting = largest;
    weighting = fmaxf(1.0f - (weighting / largest), minWeight) * weightMultiplier;
    return weighting;
}

/* ------------------------------------------------------------------------------------------------
 * a1: MaskFusion::processFrame, -static (single background model).  Core/MaskFusion.cpp:200-607.
 * ---------------------------------------------------------------------------------------------- */
struct mfo_ctx {
    mfo_config cfg;
    mfo_cam cam;
    int tick;
    float pose[16], lastPose[16];
    int count;
    float* surf[2];
    int cur; /* index of the live surfel buffer */
    /* frame images */
    uint8_t* rgb; float* depth; float* depthF; uint8_t* mask;
    /* current-frame pyramid (Model::GPUSetup) */
    float* depthPyr[3]; float* vmap[3]; float* nmap[3];
    /* model-side pyramid (RGBDOdometry) */
    float* vmap_g[3]; float* nmap_g[3];
    /* index map */
    int32_t* index; float* ivc; float* ict; float* inr;
    /* splat prediction + fill-in */
    uint8_t* predImage; float* predVertex; float* predNormal; uint16_t* predTime;
    uint8_t* fillImage; float* fillVertex; float* fillNormal;
    /* candidates */
    uint8_t* cand_op; int32_t* cand_best; float* cand_rec; int n_cand;
    float lastICPError, lastICPCount;
    int lastFillIn;
    double tms[8];
    rgbd_scratch rs; uint8_t* lastNext[3]; mfo_track_stats stats;
    const float* depthF_override;   /* test isolation: see mfo_override_filtered_depth */
    int frameToFrameRGB;            /* MaskFusion::frameToFrameRGB ("-ftf"): Model.cpp:399-400 (initRGBModel source), :981 (fill_rgb passthrough) */
};
void mfo_set_frame_to_frame_rgb(mfo_ctx* x, int on) { x->frameToFrameRGB = on; }

void mfo_default_config(mfo_config* c, int W, int H, float fx, float fy, float cx, float cy) {
    memset(c, 0, sizeof(*c));
    c->W = W; c->H = H; c->fx = fx; c->fy = fy; c->cx = cx; c->cy = cy;
    c->timeDelta = 200; c->confGlobal = 4.f; c->depthCutoff = 3.f; c->icpWeight = 10.f;
    c->maxDepthProcessed = 20.f; c->outlierCoeff = 0.9f;
    c->fastOdom = 0; c->pyramid = 1; c->so3 = 1;
    c->capacity = 3072 * 3072;
}

mfo_ctx* mfo_create(const mfo_config* cfg) {
    mfo_ctx* x = (mfo_ctx*)calloc(1, sizeof(mfo_ctx));
    x->cfg = *cfg;
    x->cam.W = cfg->W; x->cam.H = cfg->H; x->cam.fx = cfg->fx; x->cam.fy = cfg->fy; x->cam.cx = cfg->cx; x->cam.cy = cfg->cy;
    const int W = cfg->W, H = cfg->H, P = W * H;
    x->tick = 1;
    for (int k = 0; k < 16; ++k) x->pose[k] = x->lastPose[k] = (k % 5 == 0) ? 1.f : 0.f;
    x->surf[0] = (float*)calloc((size_t)cfg->capacity * 12, sizeof(float));
    x->surf[1] = (float*)calloc((size_t)cfg->capacity * 12, sizeof(float));
    x->rgb = (uint8_t*)calloc((size_t)P * 3, 1);
    x->depth = (float*)calloc(P, sizeof(float));
    x->depthF = (float*)calloc(P, sizeof(float));
    x->mask = (uint8_t*)calloc(P, 1);
    for (int i = 0; i < 3; ++i) {
        const int lp = (W >> i) * (H >> i);
        x->depthPyr[i] = (float*)calloc(lp, sizeof(float));
        x->vmap[i] = (float*)calloc((size_t)lp * 3, sizeof(float));
        x->nmap[i] = (float*)calloc((size_t)lp * 3, sizeof(float));
        x->vmap_g[i] = (float*)calloc((size_t)lp * 3, sizeof(float));
        x->nmap_g[i] = (float*)calloc((size_t)lp * 3, sizeof(float));
    }
    x->index = (int32_t*)calloc(P, sizeof(int32_t));
    x->ivc = (float*)calloc((size_t)P * 4, sizeof(float));
    x->ict = (float*)calloc((size_t)P * 4, sizeof(float));
    x->inr = (float*)calloc((size_t)P * 4, sizeof(float));
    x->predImage = (uint8_t*)calloc((size_t)P * 4, 1);
    x->predVertex = (float*)calloc((size_t)P * 4, sizeof(float));
    x->predNormal = (float*)calloc((size_t)P * 4, sizeof(float));
    x->predTime = (uint16_t*)calloc(P, sizeof(uint16_t));
    x->fillImage = (uint8_t*)calloc((size_t)P * 4, 1);
    x->fillVertex = (float*)calloc((size_t)P * 4, sizeof(float));
    x->fillNormal = (float*)calloc((size_t)P * 4, sizeof(float));
    const int maxc = ((W + 1) / 2) * ((H + 1) / 2);
    x->cand_op = (uint8_t*)calloc(maxc, 1);
    x->cand_best = (int32_t*)calloc(maxc, sizeof(int32_t));
    x->cand_rec = (float*)calloc((size_t)maxc * 12, sizeof(float));
    rgbd_scratch_alloc(&x->rs, W, H); pyr_u8_alloc(x->lastNext, W, H);
    return x;
}

void mfo_destroy(mfo_ctx* x) {
    if (!x) return;
    rgbd_scratch_free(&x->rs); pyr_u8_free(x->lastNext);
    free(x->surf[0]); free(x->surf[1]); free(x->rgb); free(x->depth); free(x->depthF); free(x->mask);
    for (int i = 0; i < 3; ++i) { free(x->depthPyr[i]); free(x->vmap[i]); free(x->nmap[i]); free(x->vmap_g[i]); free(x->nmap_g[i]); }
    free(x->index); free(x->ivc); free(x->ict); free(x->inr);
    free(x->predImage); free(x->predVertex); free(x->predNormal); free(x->predTime);
    free(x->fillImage); free(x->fillVertex); free(x->fillNormal);
    free(x->cand_op); free(x->cand_best); free(x->cand_rec);
    free(x);
}

/* MaskFusion::predict (MaskFusion.cpp:616-628) for the background model */
static void oracle_predict(mfo_ctx* x) {
    const mfo_config* g = &x->cfg;
    mfo_combined_predict(&x->cam, x->pose, x->surf[x->cur], x->count, g->maxDepthProcessed, g->confGlobal, x->tick,
                         x->tick, g->timeDelta, x->predImage, x->predVertex, x->predNormal, x->predTime);
    /* performFillIn(textureRGB, textureDepthMetricFiltered, frameToFrameRGB, lost=false): FillIn::image gets passthrough = lost ||
     * frameToFrameRGB (Model.cpp:981), FillIn::vertex / normal get `lost` only */
    mfo_fill_in(&x->cam, x->predImage, x->predVertex, x->predNormal, x->rgb, x->depthF, x->frameToFrameRGB ? 2 : 0, x->fillImage,
                x->fillVertex, x->fillNormal);
}

/* fusion, Core/MaskFusion.cpp:539-565 (single model) */
static void oracle_fuse(mfo_ctx* x, float weightMultiplier) {
    const mfo_config* g = &x->cfg;
    double t0;
    /* (the predict() at :423 only feeds the dead loop-closure block and is overwritten at :569 -- skipped) */
    const int src = x->cur, dst = 1 - x->cur;
    t0 = now_ms();
    mfo_predict_indices(&x->cam, x->pose, x->surf[src], x->count, x->tick, g->maxDepthProcessed, g->timeDelta,
                        x->index, x->ivc, x->ict, x->inr);
    x->tms[3] += now_ms() - t0;
    t0 = now_ms();
    const float weighting = mfo_fusion_weight(x->pose, x->lastPose, weightMultiplier);
    /* Model::fuse maxDepth uniform: min(depthCutoff, model.maxDepth = FLT_MAX, bb_max_z = FLT_MAX) */
    mfo_fuse_data(&x->cam, x->pose, x->rgb, x->depth, x->depthF, x->mask, 0, x->tick, weighting, g->depthCutoff,
                  x->index, x->ivc, x->inr, x->cand_op, x->cand_best, x->cand_rec, &x->n_cand);
    x->tms[4] += now_ms() - t0;
    t0 = now_ms();
    mfo_fuse_update(x->surf[src], x->surf[dst], x->count, x->tick, x->cand_op, x->cand_best, x->cand_rec, x->n_cand);
    x->tms[5] += now_ms() - t0;
    t0 = now_ms();
    mfo_predict_indices(&x->cam, x->pose, x->surf[dst], x->count, x->tick, g->maxDepthProcessed, g->timeDelta,
                        x->index, x->ivc, x->ict, x->inr);
    x->tms[3] += now_ms() - t0;
    t0 = now_ms();
    x->count = mfo_clean(&x->cam, x->pose, x->surf[dst], x->count, x->cand_op, x->cand_rec, x->n_cand, x->tick,
                         g->timeDelta, g->confGlobal, g->maxDepthProcessed, g->outlierCoeff, 0, x->index, x->ivc,
                         x->ict, x->inr, x->depthF, x->mask, x->surf[src], g->capacity);
    /* two swaps (fuse, clean) leave the live buffer where it started */
    x->tms[6] += now_ms() - t0;
}

static void mat4_mul_cm_early(const float* a, const float* b, float* out) { /* column-major 4x4 */
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float sum = 0;
            for (int k = 0; k < 4; ++k) sum += a[k * 4 + r] * b[c * 4 + k];
            out[c * 4 + r] = sum;
        }
}

int mfo_process_frame(mfo_ctx* x, const uint8_t* rgb, const float* depth, float weightMultiplier) {
    return mfo_process_frame_ex(x, rgb, depth, weightMultiplier, NULL, 0);
}

/* processFrame(frame, inPose, weightMultiplier, bootstrap), Core/MaskFusion.cpp:200-607; inPose16 column-major or NULL */
int mfo_process_frame_ex(mfo_ctx* x, const uint8_t* rgb, const float* depth, float weightMultiplier, const float* inPose16,
                         int bootstrap) {
    const mfo_config* g = &x->cfg;
    const int W = g->W, H = g->H, P = W * H;
    double t0 = now_ms();
    memcpy(x->rgb, rgb, (size_t)P * 3);
    memcpy(x->depth, depth, sizeof(float) * P);
    if (x->depthF_override) { memcpy(x->depthF, x->depthF_override, sizeof(float) * P); x->depthF_override = NULL; }
    else mfo_bilateral(x->depth, x->depthF, W, H);        /* filterDepth, :217 */
    memset(x->mask, 0, P);                                /* !enableMultipleModels, :223-230 */
    x->tms[0] += now_ms() - t0;

    if (x->tick == 1) {
        /* :235-238 */
        x->cur = 0;
        x->count = mfo_init_surfels(&x->cam, x->rgb, x->depth, x->depthF, x->tick, g->maxDepthProcessed,
                                    x->surf[0], g->capacity);
        first_rgb(x->rgb, W, H, x->lastNext);             /* initFirstRGB, :238 */
    } else if (inPose16 && !bootstrap) {
        /* :413-415 -- Model::overridePose (Model.h:235-238): lastPose = pose; pose = inPose.  No tracking. */
        memcpy(x->lastPose, x->pose, sizeof(x->pose));
        memcpy(x->pose, inPose16, sizeof(x->pose));
        if (!g->rgbOnly) oracle_fuse(x, weightMultiplier);   /* :539: if (!rgbOnly && trackingOk && !lost) */
    } else {
        /* Model::generateCUDATextures(depthFiltered, mask, K, depthCutoff), Model.cpp:350-389 */
        t0 = now_ms();
        memcpy(x->depthPyr[0], x->depthF, sizeof(float) * P);
        for (int i = 1; i < 3; ++i) mfo_pyrdown_gauss_f(x->depthPyr[i - 1], x->depthPyr[i], W >> (i - 1), H >> (i - 1));
        for (int i = 0; i < 3; ++i) {
            const int div = 1 << i;
            mfo_create_vmap(x->depthPyr[i], x->vmap[i], W >> i, H >> i, g->fx / div, g->fy / div, g->cx / div,
                            g->cy / div, g->depthCutoff);
            mfo_create_nmap(x->vmap[i], x->nmap[i], W >> i, H >> i);
        }
        x->tms[0] += now_ms() - t0;

        /* Model::performTracking, Model.cpp:427-447 */
        t0 = now_ms();
        memcpy(x->lastPose, x->pose, sizeof(x->pose));
        const int doFillIn = mfo_requires_fill_in(x->predImage, W, H, 0.75f);
        x->lastFillIn = doFillIn;
        /* initICPModel, RGBDOdometry.cpp:153-185 */
        mfo_copy_maps(doFillIn ? x->fillVertex : x->predVertex, doFillIn ? x->fillNormal : x->predNormal,
                      x->vmap_g[0], x->nmap_g[0], W, H);
        for (int i = 1; i < 3; ++i) {
            mfo_resize_map(x->vmap_g[i - 1], x->vmap_g[i], W >> (i - 1), H >> (i - 1), 0);
            mfo_resize_map(x->nmap_g[i - 1], x->nmap_g[i], W >> (i - 1), H >> (i - 1), 1);
        }
        float R[9], t[3];
        pose16_to_Rt(x->pose, R, t);
        for (int i = 0; i < 3; ++i)
            mfo_transform_maps(x->vmap_g[i], x->nmap_g[i], R, t, x->vmap_g[i], x->nmap_g[i], W >> i, H >> i);
        x->tms[1] += now_ms() - t0;

        t0 = now_ms();
        const float* cv[3] = {x->vmap[0], x->vmap[1], x->vmap[2]};
        const float* cn[3] = {x->nmap[0], x->nmap[1], x->nmap[2]};
        const float* pv[3] = {x->vmap_g[0], x->vmap_g[1], x->vmap_g[2]};
        const float* pn[3] = {x->nmap_g[0], x->nmap_g[1], x->nmap_g[2]};
        /* initRGBModel(doFillIn || (frameToFrameRGB && allowsFillIn()) ? fill-in image : RGB projection), Model.cpp:395-401 */
        track_model(g, &x->rs, cv, cn, pv, pn, doFillIn ? x->fillVertex : x->predVertex,
                    (doFillIn || x->frameToFrameRGB) ? x->fillImage : x->predImage,
                    x->rgb, x->lastNext, R, t, NULL, &x->stats);
        x->lastICPError = x->stats.lastICPError; x->lastICPCount = x->stats.lastICPCount;
        Rt_to_pose16(R, t, x->pose);
        x->tms[2] += now_ms() - t0;

        if (bootstrap && inPose16) { /* :280-283: overridePose(pose * inPose) */
            float np[16];
            memcpy(x->lastPose, x->pose, sizeof(x->pose));
            mat4_mul_cm_early(x->pose, inPose16, np);
            memcpy(x->pose, np, sizeof(np));
        }
        if (!g->rgbOnly) oracle_fuse(x, weightMultiplier);   /* :539: if (!rgbOnly && trackingOk && !lost) */
    }
    t0 = now_ms();
    oracle_predict(x); /* :569 */
    x->tms[7] += now_ms() - t0;
    x->tick++;
    return 0;
}

/* Test isolation: the NEXT mfo_process_frame takes this image as the bilateral filter's output instead of running the filter
 * (so that a pipeline-level comparison of the surfel life cycle is not perturbed by the 1e-6 difference between two exp()
 * implementations; the filter itself is compared on its own).  The pointer must stay valid during that call. */
void mfo_override_filtered_depth(mfo_ctx* x, const float* depthF) { x->depthF_override = depthF; }

void mfo_get_pose(const mfo_ctx* x, float* p) { memcpy(p, x->pose, sizeof(x->pose)); }
int mfo_get_count(const mfo_ctx* x) { return x->count; }
int mfo_get_tick(const mfo_ctx* x) { return x->tick; }
const float* mfo_get_surfels(const mfo_ctx* x) { return x->surf[x->cur]; }
void mfo_get_icp_stats(const mfo_ctx* x, float* e, float* c) { *e = x->lastICPError; *c = x->lastICPCount; }
void mfo_get_track_stats(const mfo_ctx* x, mfo_track_stats* out) { *out = x->stats; }
void mfo_get_timings(const mfo_ctx* x, double* ms8) { memcpy(ms8, x->tms, sizeof(x->tms)); }
const float* mfo_dbg_depthF(const mfo_ctx* x) { return x->depthF; }
const float* mfo_dbg_pred_vertex(const mfo_ctx* x) { return x->predVertex; }
const float* mfo_dbg_pred_normal(const mfo_ctx* x) { return x->predNormal; }
const uint8_t* mfo_dbg_pred_image(const mfo_ctx* x) { return x->predImage; }
int mfo_dbg_last_fillin(const mfo_ctx* x) { return x->lastFillIn; }

/* =====================================================================================================
 * a20: MfSegmentation -- GPU half (Core/Cuda/segmentation.cu) restated on the CPU
 * ===================================================================================================== */
/* segmentation.cu:122-177.  Note: an invalid centre vertex has z = 0 in the reference (x = NaN only); the maps here
 * carry NaN in all planes, so "v.z <= 0" is extended by isnan(v.z) to keep the reference's answer (edge = 1). */
void mfo_geometric_edge_map(const float* vmap, const float* nmap, float* out, int W, int H, float wD, float wC) {
    const int P = W * H;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            const int i = y * W + x;
            if (x < 1 || x >= W - 1 || y < 1 || y >= H - 1) { out[i] = 1.0f; continue; }
            const f3 v = f3_make(vmap[i], vmap[P + i], vmap[2 * P + i]);
            const f3 n = f3_make(nmap[i], nmap[P + i], nmap[2 * P + i]);
            if (v.z <= 0.0f || isnan(v.z)) { out[i] = 1.0f; continue; }
            static const int ox[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
            static const int oy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
            float c = 0.0f, d = 0.0f;
            for (int k = 0; k < 8; ++k) {
                const int j = (y + oy[k]) * W + (x + ox[k]);
                const f3 vn = f3_make(vmap[j], vmap[P + j], vmap[2 * P + j]);
                const f3 nn = f3_make(nmap[j], nmap[P + j], nmap[2 * P + j]);
                /* getConcavityTerm, :106-112 */
                float ct = (f3_dot(f3_sub(vn, v), n) < 0) ? 0.f : 1.f - f3_dot(nn, n);
                c = fmaxf(ct, c); /* fmax ignores a NaN operand, like CUDA's fmax */
                /* getDistanceTerm, :115-119 */
                d = fmaxf(fabsf(f3_dot(f3_sub(vn, v), n)), d);
            }
            c = fmaxf(c, 0.0f);
            c *= wC;
            d *= wD;
            const float edgeness = (c > d) ? c : d; /* max(c,d) */
            out[i] = fminf(1.0f, edgeness);
        }
    }
}

void mfo_threshold_map(const float* in, uint8_t* out, int n, float threshold) {
    for (int i = 0; i < n; ++i) out[i] = in[i] > threshold ? 255 : 0;
}
void mfo_invert_map(const uint8_t* in, uint8_t* out, int n) {
    for (int i = 0; i < n; ++i) out[i] = (uint8_t)(255 - in[i]);
}
static void dilate_u8(const uint8_t* in, uint8_t* out, int W, int H, int radius) { /* segmentation.cu:237-255 */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t r = 0;
            for (int cy = imax(y - radius, 0); cy <= imin(y + radius, H - 1) && !r; ++cy)
                for (int cx = imax(x - radius, 0); cx <= imin(x + radius, W - 1); ++cx) {
                    if (cy == y && cx == x) continue;
                    if (in[cy * W + cx] == 255) { r = 255; break; }
                }
            out[y * W + x] = r;
        }
}
static void erode_u8(const uint8_t* in, uint8_t* out, int W, int H, int radius) { /* segmentation.cu:217-235 */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t r = 255;
            for (int cy = imax(y - radius, 0); cy <= imin(y + radius, H - 1) && r; ++cy)
                for (int cx = imax(x - radius, 0); cx <= imin(x + radius, W - 1); ++cx) {
                    if (cy == y && cx == x) continue;
                    if (in[cy * W + cx] == 0) { r = 0; break; }
                }
            out[y * W + x] = r;
        }
}
void mfo_morph_closing_u8(uint8_t* data, uint8_t* buffer, int W, int H, int radius, int iterations) {
    for (int i = 0; i < iterations; ++i) { dilate_u8(data, buffer, W, H, radius); erode_u8(buffer, data, W, H, radius); }
}

/* ---- CPU half ---------------------------------------------------------------------------------- */
int mfo_connected_components4(const uint8_t* bin, int32_t* labels, int32_t* stats, int max_comp, int W, int H) {
    const int P = W * H;
    for (int i = 0; i < P; ++i) labels[i] = 0;
    int* stack = (int*)malloc(sizeof(int) * (size_t)P);
    int n = 1;
    if (max_comp > 0) { stats[0] = stats[1] = 0; stats[2] = W; stats[3] = H; stats[4] = 0; }
    for (int i = 0; i < P; ++i) {
        if (!bin[i] || labels[i]) continue;
        if (n >= max_comp) break;
        int sp = 0, x0 = W, y0 = H, x1 = -1, y1 = -1, area = 0;
        stack[sp++] = i; labels[i] = n;
        while (sp) {
            const int p = stack[--sp], x = p % W, y = p / W;
            ++area;
            if (x < x0) x0 = x;
            if (x > x1) x1 = x;
            if (y < y0) y0 = y;
            if (y > y1) y1 = y;
            if (x > 0 && bin[p - 1] && !labels[p - 1]) { labels[p - 1] = n; stack[sp++] = p - 1; }
            if (x < W - 1 && bin[p + 1] && !labels[p + 1]) { labels[p + 1] = n; stack[sp++] = p + 1; }
            if (y > 0 && bin[p - W] && !labels[p - W]) { labels[p - W] = n; stack[sp++] = p - W; }
            if (y < H - 1 && bin[p + W] && !labels[p + W]) { labels[p + W] = n; stack[sp++] = p + W; }
        }
        stats[n * 5 + 0] = x0; stats[n * 5 + 1] = y0; stats[n * 5 + 2] = x1 - x0 + 1; stats[n * 5 + 3] = y1 - y0 + 1;
        stats[n * 5 + 4] = area;
        ++n;
    }
    int bg = 0;
    for (int i = 0; i < P; ++i) bg += labels[i] == 0;
    if (max_comp > 0) stats[4] = bg;
    free(stack);
    return n;
}

void mfo_default_seg_params(mfo_seg_params* p) {
    p->threshold = 0.1f; p->weightDistance = 1.f; p->weightConvexity = 1.f;
    p->morphEdgeIterations = 3; p->morphEdgeRadius = 1; p->morphMaskIterations = 3; p->morphMaskRadius = 1;
    p->removeEdges = 1; p->minRelSizeNew = 0.07f; p->maxRelSizeNew = 0.4f; p->personClassID = 255;
}

/* grayscale dilate / erode with OpenCV's MORPH_ELLIPSE element (cv::getStructuringElement) and the default constant
 * border (ignored outside the image) -- stand-in for cv::morphologyEx(MORPH_CLOSE), MfSegmentation.cpp:424-426 */
static void ellipse_rows(int r, int* j1, int* j2) {
    const int ks = 2 * r + 1;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0;
    for (int i = 0; i < ks; ++i) {
        const int dy = i - r;
        int dx = (int)lrint(r * sqrt((r * r - dy * dy) * inv_r2));
        j1[i] = imax(r - dx, 0); j2[i] = imin(r + dx + 1, ks);
    }
}
static void morph_gray(const uint8_t* in, uint8_t* out, int W, int H, int r, int dilate) {
    int j1[64], j2[64];
    ellipse_rows(r, j1, j2);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int v = dilate ? 0 : 255;
            for (int i = 0; i < 2 * r + 1; ++i) {
                const int yy = y + i - r;
                if (yy < 0 || yy >= H) continue;
                for (int j = j1[i]; j < j2[i]; ++j) {
                    const int xx = x + j - r;
                    if (xx < 0 || xx >= W) continue;
                    const int s = in[yy * W + xx];
                    if (dilate ? s > v : s < v) v = s;
                }
            }
            out[y * W + x] = (uint8_t)v;
        }
}

/* cv::morphologyEx(img, img, MORPH_CLOSE, ellipse(2r+1), Point(-1,-1), iterations) stand-in on its own (used by the compiled slice of
 * MfSegmentation.cpp, oracle/build_seg.py): n dilations, then n erosions */
void mfo_morph_close_ellipse(uint8_t* img, int W, int H, int radius, int iterations) {
    uint8_t* tmp = (uint8_t*)malloc((size_t)W * H);
    for (int it = 0; it < iterations; ++it) { morph_gray(img, tmp, W, H, radius, 1); memcpy(img, tmp, (size_t)W * H); }
    for (int it = 0; it < iterations; ++it) { morph_gray(img, tmp, W, H, radius, 0); memcpy(img, tmp, (size_t)W * H); }
    free(tmp);
}

/* mask values without a class id count as "no mask" (0): upstream reads classIDs[mask] unchecked (MfSegmentation.cpp:226,311),
 * which is undefined for such inputs; the product and this restatement define it the same way */
static inline int mask_id(int value, int nMasks) { return value < nMasks ? value : 0; }

void mfo_mf_segmentation_cpu(const mfo_seg_params* prm, int W, int H, const uint8_t* binaryIn, const float* depth,
                             const uint8_t* mask, const int32_t* classIDs, int nMasks, const uint8_t* projectedIDs,
                             const int32_t* modelIDs, const int32_t* modelClassIDs, int nModels, int nextModelID,
                             int allowNew, uint8_t* ignoreMap, uint8_t* full, int* hasNewLabel, int* newClassID) {
    const int total = W * H;
    const size_t minNewMaskPixels = (size_t)(prm->minRelSizeNew * total);
    const size_t maxNewMaskPixels = (size_t)(prm->maxRelSizeNew * total);
    uint8_t* binary = (uint8_t*)malloc(total);
    memcpy(binary, binaryIn, total);
    *hasNewLabel = 0; *newClassID = -1;
    /* ignore map, :221-235 */
    if (nMasks) {
        for (int i = 0; i < total; ++i) {
            if (classIDs[mask_id(mask[i], nMasks)] == prm->personClassID) { ignoreMap[i] = 255; binary[i] = 0; }
            else ignoreMap[i] = 0;
        }
    } else {
        for (int i = 0; i < total; ++i)
            if (ignoreMap[i]) binary[i] = 0;
    }
    /* connected components, :239 */
    const int maxComp = total / 2 + 2;
    int32_t* labels = (int32_t*)malloc(sizeof(int32_t) * total);
    int32_t* stats = (int32_t*)malloc(sizeof(int32_t) * 5 * (size_t)maxComp);
    const int nComponents = mfo_connected_components4(binary, labels, stats, maxComp, W, H);
    /* removeEdges, :243-291: 5 Jacobi sweeps (neighbours are read from the previous sweep's labels) */
    if (prm->removeEdges) {
        int32_t* r = (int32_t*)malloc(sizeof(int32_t) * total);
        static const int ox[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
        static const int oy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
        for (int it = 0; it < 5; ++it) {
            memcpy(r, labels, sizeof(int32_t) * total);
            for (int y = 1; y < H - 1; ++y)
                for (int x = 1; x < W - 1; ++x) {
                    const int c = r[y * W + x];
                    const float d = depth[y * W + x];
                    if (c == 0 || stats[c * 5 + 4] < 50) {
                        for (int k = 0; k < 8; ++k) {
                            const int n = labels[(y + oy[k]) * W + (x + ox[k])];
                            if (n != 0 && fabsf(depth[(y + oy[k]) * W + (x + ox[k])] - d) < 0.008 && stats[n * 5 + 4] > 50) {
                                r[y * W + x] = n;
                                break;
                            }
                        }
                    }
                }
            memcpy(labels, r, sizeof(int32_t) * total);
        }
        free(r);
    }
    /* overlaps, :299-318 */
    int* mapComponentToMask = (int*)calloc(nComponents, sizeof(int));
    int* maskComponentPixels = (int*)calloc(nMasks > 0 ? nMasks : 1, sizeof(int));
    int* compMaskOverlap = (int*)calloc((size_t)nComponents * (nMasks > 0 ? nMasks : 1), sizeof(int));
    int* compModelOverlap = (int*)calloc((size_t)nComponents * nModels, sizeof(int));
    int idToIndex[256];
    for (int k = 0; k < 256; ++k) idToIndex[k] = 0; /* std::map default */
    for (int m = 0; m < nModels; ++m) idToIndex[modelIDs[m] & 255] = m;
    for (int i = 0; i < total; ++i) {
        int mi = idToIndex[projectedIDs[i]];
        if (mi >= nModels) mi = 0;
        compModelOverlap[(size_t)labels[i] * nModels + mi]++;
    }
    if (nMasks) {
        for (int i = 0; i < total; ++i) compMaskOverlap[(size_t)labels[i] * nMasks + mask_id(mask[i], nMasks)]++;
        for (int c = 1; c < nComponents; ++c) {
            const int csize = stats[c * 5 + 4];
            if (csize > 160) { /* minMappedComponentSize */
                const int t = (int)(0.65f * csize);
                for (int m = 1; m < nMasks; ++m)
                    if (compMaskOverlap[(size_t)c * nMasks + m] > t) {
                        mapComponentToMask[c] = m;
                        maskComponentPixels[m] += csize;
                    }
            } else mapComponentToMask[c] = 0;
        }
    }
    for (int i = 0; i < total; ++i) full[i] = (uint8_t)mapComponentToMask[labels[i]];
    for (int i = 0; i < total; ++i)
        if (ignoreMap[i]) full[i] = 255;
    int maskToID[256];
    for (int k = 0; k < 256; ++k) maskToID[k] = 0;
    maskToID[255] = 255; maskToID[0] = 0;
    if (nMasks) {
        /* closing on the label image, :424-426 (iterations == 0 copies) */
        if (prm->morphMaskIterations > 0) {
            uint8_t* tmp = (uint8_t*)malloc(total);
            for (int it = 0; it < prm->morphMaskIterations; ++it) { morph_gray(full, tmp, W, H, prm->morphMaskRadius, 1); memcpy(full, tmp, total); }
            for (int it = 0; it < prm->morphMaskIterations; ++it) { morph_gray(full, tmp, W, H, prm->morphMaskRadius, 0); memcpy(full, tmp, total); }
            free(tmp);
        }
        for (int m = 1; m < nMasks; ++m) { maskToID[m] = 0; if (classIDs[m] == prm->personClassID) maskToID[m] = 255; }
        /* mask x model overlap, :441-447 */
        unsigned* maskOverlap = (unsigned*)calloc((size_t)nModels * 256, sizeof(unsigned));
        for (int i = 0; i < total; ++i)
            for (int b = 0; b < nModels; ++b)
                if (projectedIDs[i] == modelIDs[b]) maskOverlap[(size_t)b * 256 + full[i]]++;
        for (int midx = 1; midx < nMasks; ++midx) {
            if (maskToID[midx] == 255) continue;
            int bestModelIndex = 0;
            unsigned bestOverlap = 0;
            const int maskClassID = classIDs[midx];
            for (int j = 1; j < nModels; ++j)
                if (maskOverlap[(size_t)j * 256 + midx] > bestOverlap) { bestOverlap = maskOverlap[(size_t)j * 256 + midx]; bestModelIndex = j; }
            const int bestModelMatchesClass = modelClassIDs[bestModelIndex] == maskClassID;
            if (bestOverlap < 0.05f * maskComponentPixels[midx]) bestModelIndex = 0;
            if (bestModelIndex != 0 && bestModelMatchesClass) maskToID[midx] = modelIDs[bestModelIndex];
            else if (!*hasNewLabel && allowNew && (size_t)maskComponentPixels[midx] > minNewMaskPixels &&
                     (size_t)maskComponentPixels[midx] < maxNewMaskPixels && bestModelIndex == 0) {
                maskToID[midx] = nextModelID;
                *hasNewLabel = 1;
                *newClassID = maskClassID;
            } else maskToID[midx] = 255;
        }
        free(maskOverlap);
    }
    for (int i = 0; i < total; ++i) full[i] = (uint8_t)maskToID[full[i]];
    /* unused components -> existing models, :500-522 (the reference's inclusive box bounds are clamped to the image) */
    for (int c = 1; c < nComponents; ++c) {
        if (mapComponentToMask[c] != 0) continue;
        int model_index = 0, overlap = compModelOverlap[(size_t)c * nModels];
        for (int m = 1; m < nModels; ++m)
            if (compModelOverlap[(size_t)c * nModels + m] > overlap) { overlap = compModelOverlap[(size_t)c * nModels + m]; model_index = m; }
        const int model_id = modelIDs[model_index];
        if (model_id > 0 && overlap > 0.6f * stats[c * 5 + 4]) {
            const int x1 = stats[c * 5 + 0], x2 = imin(stats[c * 5 + 0] + stats[c * 5 + 2], W - 1);
            const int y1 = stats[c * 5 + 1], y2 = imin(stats[c * 5 + 1] + stats[c * 5 + 3], H - 1);
            for (int y = y1; y <= y2; ++y)
                for (int x = x1; x <= x2; ++x)
                    if (labels[y * W + x] == c) full[y * W + x] = (uint8_t)model_id;
        }
    }
    free(binary); free(labels); free(stats); free(mapComponentToMask); free(maskComponentPixels); free(compMaskOverlap);
    free(compModelOverlap);
}

/* =====================================================================================================
 * a1, multi-model branch: MaskFusion::processFrame with enableMultipleModels (Core/MaskFusion.cpp:200-607)
 * ===================================================================================================== */
typedef struct {
    int id, classID, isStatic, age, count, cur, cap;
    float pose[16], lastPose[16], initialC2Winv[16];
    float confThr, maxDepth;
    float* surf[2];
    uint8_t* predImage; float* predVertex; float* predNormal; uint16_t* predTime;
    float lastICPError, lastICPCount;
    uint8_t* lastNext[3];
    int bbox[6];   /* Model::lastBoundingBox in mm, {min xyz, max xyz}; Model.cpp:315 initial value = empty */
    float trackedPose[16]; int trackedThisFrame;   /* what THIS side's tracking step returned from its own state (teacher forcing keeps it readable) */
    float trackedPoseAlt[16];                      /* ... and what it returns when the start pose is moved by one micrometre (its own sensitivity) */
    float trackLog[20][32]; int trackLogN;         /* reduced systems of the last tracking step, per iteration (mfo_last_track_log) */
    float probeLog[20][32]; int probeLogN;         /* ... and the systems of the same step evaluated at poses handed in from outside (mfo_mm_set_probe_poses) */
} mm_model;
static void mm_bbox_reset(mm_model* m) { for (int k = 0; k < 3; ++k) { m->bbox[k] = 100000; m->bbox[3 + k] = -100000; } }
/* Model::renderPointCloud (Model.cpp:287-346) + draw_global_surface.vert:55-78, as the GUI runs it after every frame with its defaults
 * (drawUnstable = false): the box of the surfels whose confidence exceeds the model's threshold, model coordinates, int(1000 * x) */
static void mm_bbox_update(mm_model* m) {
    mm_bbox_reset(m);
    const float* s = m->surf[m->cur];
    for (int i = 0; i < m->count; ++i) {
        if (!(s[(size_t)i * 12 + 3] > m->confThr)) continue;
        for (int k = 0; k < 3; ++k) {
            const int v = (int)(1000.f * s[(size_t)i * 12 + k]);
            if (v < m->bbox[k]) m->bbox[k] = v;
            if (v > m->bbox[3 + k]) m->bbox[3 + k] = v;
        }
    }
}

struct mfo_mm {
    mfo_mm_config cfg;
    mfo_cam cam;
    int tick, nextID, spawnOffset, nModels;
    mm_model* models;
    uint8_t* rgb; float* depth; float* depthF; uint8_t* mask;
    float* depthPyr[3]; float* vmap[3]; float* nmap[3];
    float* vmap_g[3]; float* nmap_g[3];
    int32_t* index; float* ivc; float* ict; float* inr;
    uint8_t* fillImage; float* fillVertex; float* fillNormal;
    uint8_t* cand_op; int32_t* cand_best; float* cand_rec; int n_cand;
    rgbd_scratch rs;
    float* edge; uint8_t* binEdge; uint8_t* ucharBuf; uint8_t* projIDs; uint8_t* ignoreMap; uint8_t* fullSeg;
    const float* depthF_override;   /* test isolation, as mfo_override_filtered_depth: consumed by the next mfo_mm_process_frame */
    int frameToFrameRGB;            /* as mfo_ctx */
    int bboxLimit;                  /* object models limit their fusion depth by lastBoundingBox (upstream with its GUI; default 1) */
    int32_t trackable[256]; int nTrackable;   /* MaskFusion::trackableClassIds (MaskFusion.cpp:261,940); empty: every class */
    /* test isolation ("teacher forcing"): poses another implementation obtained for this frame, by model id; consumed by the next frame */
    int32_t forceIDs[64]; float forcePoses[64][16]; int nForce; int forceOn;
    /* per-iteration teacher forcing of the Gauss-Newton loop: for model ids probeIDs[k], the pose (Rcurr row-major 9, tcurr 3) another
     * implementation used in each of its iterations; consumed by the next frame */
    int32_t probeIDs[64]; float (*probePoses)[20][12]; int nProbe;
};
void mfo_mm_set_frame_to_frame_rgb(mfo_mm* x, int on) { x->frameToFrameRGB = on; }
void mfo_mm_set_bbox_limit(mfo_mm* x, int on) { x->bboxLimit = on; }
void mfo_mm_override_filtered_depth(mfo_mm* x, const float* depthF) { x->depthF_override = depthF; }
/* MaskFusion::setTrackableClassIds (MaskFusion.cpp:940) */
void mfo_mm_set_trackable_class_ids(mfo_mm* x, const int32_t* ids, int n) {
    x->nTrackable = n < 0 ? 0 : (n > 256 ? 256 : n);
    for (int i = 0; i < x->nTrackable; ++i) x->trackable[i] = ids[i];
}
/* Model::makeNonStatic / makeStatic(globalPose) (Model.h:264-265): initialC2Winv = pose * globalPose^-1 */
void mfo_mm_make_nonstatic(mfo_mm* x, int i) { if (i > 0 && i < x->nModels) x->models[i].isStatic = 0; }
int mfo_mm_is_nonstatic(const mfo_mm* x, int i) { return (i >= 0 && i < x->nModels) ? !x->models[i].isStatic : 0; }
int mfo_mm_model_class(const mfo_mm* x, int i) { return x->models[i].classID; }
/* Teacher forcing (test isolation, like mfo_mm_override_filtered_depth): the NEXT mfo_mm_process_frame tracks every model from its own
 * state as usual -- that result stays readable through mfo_mm_model_tracked_pose -- and then continues with the poses given here (by model
 * id; ids this side does not hold yet are ignored).  A tracked object model whose id is missing from the list is dropped as if by the 0.2 m
 * jump rule (MaskFusion.cpp:268-272), one that is listed is kept whatever the length of its own step.  A chaotic quantity (the pose of an
 * ill-conditioned object) then cannot make the two sides' model lists drift apart, and every pass of every frame is compared on equal input. */
void mfo_mm_force_tracking(mfo_mm* x, const int32_t* ids, const float* poses16, int n) {
    x->nForce = n < 0 ? 0 : (n > 64 ? 64 : n);
    for (int i = 0; i < x->nForce; ++i) { x->forceIDs[i] = ids[i]; memcpy(x->forcePoses[i], poses16 + 16 * i, sizeof(float) * 16); }
    x->forceOn = 1;
}
/* Test isolation for the tracked models' Gauss-Newton loops: the NEXT mfo_mm_process_frame also evaluates, for every model listed here, the
 * reduced normal equations (icpStep, reduce.cu:446-525) of each of its `n_it` iterations at the pose given for that iteration -- on this side's
 * own maps, which teacher forcing keeps identical to the other implementation's.  Row k of mfo_mm_model_probe_log is then directly comparable
 * with the other side's k-th system whatever its ill-conditioned steps did before: chaos cannot compound across iterations.
 * poses: [n][20][12] floats (Rcurr row-major, tcurr). */
void mfo_mm_set_probe_poses(mfo_mm* x, const int32_t* ids, const float* poses, int n) {
    x->nProbe = n < 0 ? 0 : (n > 64 ? 64 : n);
    if (!x->probePoses) x->probePoses = (float (*)[20][12])calloc(64, sizeof(float[20][12]));
    for (int i = 0; i < x->nProbe; ++i) { x->probeIDs[i] = ids[i]; memcpy(x->probePoses[i], poses + (size_t)i * 240, sizeof(float) * 240); }
}
int mfo_mm_model_probe_log(const mfo_mm* x, int i, float* out /* [20][32] */) {
    memcpy(out, x->models[i].probeLog, sizeof(x->models[i].probeLog));
    return x->models[i].probeLogN;
}
static const float* mm_forced_pose(const mfo_mm* x, int id) {
    for (int i = 0; i < x->nForce; ++i) if (x->forceIDs[i] == id) return x->forcePoses[i];
    return NULL;
}
void mfo_mm_model_tracked_pose(const mfo_mm* x, int i, float* p, int* tracked) {
    memcpy(p, x->models[i].trackedPose, sizeof(float) * 16);
    *tracked = x->models[i].trackedThisFrame;
}
/* Conditioning probe of the same step (teacher-forced frames only): the pose the tracker returns when its START pose is shifted by 1e-6 m
 * along x.  A well-posed step forgets a micrometre; an object seen as two or three small planar faces does not (its 6x6 system has a
 * condition number of 1e5..1e7 and the fp32 sums feeding it carry 1e-7 of noise whatever the summation order), and then no two
 * implementations -- the reference's own float reduction tree included -- can agree more closely than this probe moves. */
int mfo_mm_model_track_log(const mfo_mm* x, int i, float* out /* [20][32] */) {
    memcpy(out, x->models[i].trackLog, sizeof(x->models[i].trackLog));
    return x->models[i].trackLogN;
}
void mfo_mm_model_tracked_pose_alt(const mfo_mm* x, int i, float* p) { memcpy(p, x->models[i].trackedPoseAlt, sizeof(float) * 16); }

void mfo_mm_default_config(mfo_mm_config* c, int W, int H, float fx, float fy, float cx, float cy) {
    memset(c, 0, sizeof(*c));
    mfo_default_config(&c->base, W, H, fx, fy, cx, cy);
    c->confObject = 2.f; c->capacityObject = 1024 * 1024; c->trackAllModels = 1; c->modelSpawnOffset = 20; c->maxModels = 16;
    mfo_default_seg_params(&c->seg);
}

static void mat4_identity(float* m) { for (int k = 0; k < 16; ++k) m[k] = (k % 5 == 0) ? 1.f : 0.f; }
static void mat4_mul_cm(const float* a, const float* b, float* out) { /* column-major 4x4 */
    float r[16];
    for (int c = 0; c < 4; ++c)
        for (int rr = 0; rr < 4; ++rr) {
            float s = 0;
            for (int k = 0; k < 4; ++k) s += a[k * 4 + rr] * b[c * 4 + k];
            r[c * 4 + rr] = s;
        }
    memcpy(out, r, sizeof(r));
}
static void mat4_rigid_inverse_cm(const float* p, float* out) {
    float R[9], t[3], Ri[9], ti[3];
    pose16_to_Rt(p, R, t);
    pose_inverse_Rt(R, t, Ri, ti);
    Rt_to_pose16(Ri, ti, out);
}

void mfo_mm_make_static(mfo_mm* x, int i) {
    if (i <= 0 || i >= x->nModels) return;
    float ginv[16];
    mat4_rigid_inverse_cm(x->models[0].pose, ginv);
    mat4_mul_cm(x->models[i].pose, ginv, x->models[i].initialC2Winv);
    x->models[i].isStatic = 1;
}

static void mm_model_init(mfo_mm* x, mm_model* m, int id, float confThr, int cap) {
    const int P = x->cam.W * x->cam.H;
    memset(m, 0, sizeof(*m));
    m->id = id; m->classID = -1; m->isStatic = 1; m->confThr = confThr; m->maxDepth = 3.402823466e38f; m->cap = cap;
    mat4_identity(m->pose); mat4_identity(m->lastPose); mat4_identity(m->initialC2Winv);
    mm_bbox_reset(m);
    m->surf[0] = (float*)calloc((size_t)cap * 12, sizeof(float));
    m->surf[1] = (float*)calloc((size_t)cap * 12, sizeof(float));
    m->predImage = (uint8_t*)calloc((size_t)P * 4, 1);
    m->predVertex = (float*)calloc((size_t)P * 4, sizeof(float));
    m->predNormal = (float*)calloc((size_t)P * 4, sizeof(float));
    m->predTime = (uint16_t*)calloc(P, sizeof(uint16_t));
    pyr_u8_alloc(m->lastNext, x->cam.W, x->cam.H);
}
static void mm_model_free(mm_model* m) {
    pyr_u8_free(m->lastNext);
    free(m->surf[0]); free(m->surf[1]); free(m->predImage); free(m->predVertex); free(m->predNormal); free(m->predTime);
}

mfo_mm* mfo_mm_create(const mfo_mm_config* cfg) {
    mfo_mm* x = (mfo_mm*)calloc(1, sizeof(mfo_mm));
    x->cfg = *cfg;
    const mfo_config* g = &cfg->base;
    x->cam.W = g->W; x->cam.H = g->H; x->cam.fx = g->fx; x->cam.fy = g->fy; x->cam.cx = g->cx; x->cam.cy = g->cy;
    const int W = g->W, H = g->H, P = W * H;
    x->tick = 1; x->nextID = 0; x->spawnOffset = 0;
    x->bboxLimit = 1;
    x->models = (mm_model*)calloc(cfg->maxModels, sizeof(mm_model));
    mm_model_init(x, &x->models[0], x->nextID++, g->confGlobal, g->capacity); /* getNextModelID(true), :80 */
    x->nModels = 1;
    x->rgb = (uint8_t*)calloc((size_t)P * 3, 1);
    x->depth = (float*)calloc(P, sizeof(float)); x->depthF = (float*)calloc(P, sizeof(float));
    x->mask = (uint8_t*)calloc(P, 1);
    for (int i = 0; i < 3; ++i) {
        const int lp = (W >> i) * (H >> i);
        x->depthPyr[i] = (float*)calloc(lp, sizeof(float));
        x->vmap[i] = (float*)calloc((size_t)lp * 3, sizeof(float)); x->nmap[i] = (float*)calloc((size_t)lp * 3, sizeof(float));
        x->vmap_g[i] = (float*)calloc((size_t)lp * 3, sizeof(float)); x->nmap_g[i] = (float*)calloc((size_t)lp * 3, sizeof(float));
    }
    x->index = (int32_t*)calloc(P, sizeof(int32_t));
    x->ivc = (float*)calloc((size_t)P * 4, sizeof(float)); x->ict = (float*)calloc((size_t)P * 4, sizeof(float));
    x->inr = (float*)calloc((size_t)P * 4, sizeof(float));
    x->fillImage = (uint8_t*)calloc((size_t)P * 4, 1);
    x->fillVertex = (float*)calloc((size_t)P * 4, sizeof(float)); x->fillNormal = (float*)calloc((size_t)P * 4, sizeof(float));
    const int maxc = ((W + 1) / 2) * ((H + 1) / 2);
    x->cand_op = (uint8_t*)calloc(maxc, 1); x->cand_best = (int32_t*)calloc(maxc, sizeof(int32_t));
    x->cand_rec = (float*)calloc((size_t)maxc * 12, sizeof(float));
    x->edge = (float*)calloc(P, sizeof(float)); x->binEdge = (uint8_t*)calloc(P, 1); x->ucharBuf = (uint8_t*)calloc(P, 1);
    x->projIDs = (uint8_t*)calloc(P, 1); x->ignoreMap = (uint8_t*)calloc(P, 1); x->fullSeg = (uint8_t*)calloc(P, 1);
    rgbd_scratch_alloc(&x->rs, W, H);
    return x;
}

void mfo_mm_destroy(mfo_mm* x) {
    if (!x) return;
    for (int i = 0; i < x->nModels; ++i) mm_model_free(&x->models[i]);
    free(x->models); free(x->rgb); free(x->depth); free(x->depthF); free(x->mask);
    for (int i = 0; i < 3; ++i) { free(x->depthPyr[i]); free(x->vmap[i]); free(x->nmap[i]); free(x->vmap_g[i]); free(x->nmap_g[i]); }
    free(x->index); free(x->ivc); free(x->ict); free(x->inr); free(x->fillImage); free(x->fillVertex); free(x->fillNormal);
    free(x->cand_op); free(x->cand_best); free(x->cand_rec);
    free(x->edge); free(x->binEdge); free(x->ucharBuf); free(x->projIDs); free(x->ignoreMap); free(x->fullSeg);
    rgbd_scratch_free(&x->rs);
    free(x->probePoses);
    free(x);
}

/* Model::performTracking for one model (Model.cpp:427-447); returns |translation of the increment| */
static float mm_track(mfo_mm* x, mm_model* m, int allowFillIn) {
    const mfo_config* g = &x->cfg.base;
    const int W = g->W, H = g->H;
    memcpy(m->lastPose, m->pose, sizeof(m->pose));
    const int doFillIn = allowFillIn && mfo_requires_fill_in(m->predImage, W, H, 0.75f);
    mfo_copy_maps(doFillIn ? x->fillVertex : m->predVertex, doFillIn ? x->fillNormal : m->predNormal, x->vmap_g[0],
                  x->nmap_g[0], W, H);
    for (int i = 1; i < 3; ++i) {
        mfo_resize_map(x->vmap_g[i - 1], x->vmap_g[i], W >> (i - 1), H >> (i - 1), 0);
        mfo_resize_map(x->nmap_g[i - 1], x->nmap_g[i], W >> (i - 1), H >> (i - 1), 1);
    }
    float R[9], t[3], inc[16];
    pose16_to_Rt(m->pose, R, t);
    for (int i = 0; i < 3; ++i)
        mfo_transform_maps(x->vmap_g[i], x->nmap_g[i], R, t, x->vmap_g[i], x->nmap_g[i], W >> i, H >> i);
    const float* cv[3] = {x->vmap[0], x->vmap[1], x->vmap[2]};
    const float* cn[3] = {x->nmap[0], x->nmap[1], x->nmap[2]};
    const float* pv[3] = {x->vmap_g[0], x->vmap_g[1], x->vmap_g[2]};
    const float* pn[3] = {x->nmap_g[0], x->nmap_g[1], x->nmap_g[2]};
    m->probeLogN = 0;
    for (int q = 0; q < x->nProbe; ++q) {
        if (x->probeIDs[q] != m->id) continue;
        /* the iteration schedule of RGBDOdometry.cpp:327-329 and the per-level intrinsics of :346-352, as mfo_track_icp walks them */
        float Rpi[9];
        m33_inverse(R, Rpi);
        const int iters[3] = {g->fastOdom ? 3 : 10, g->pyramid ? 5 : 0, g->pyramid ? 4 : 0};
        int it = 0;
        for (int lv = 2; lv >= 0; --lv) {
            const int div = 1 << lv;
            for (int j = 0; j < iters[lv] && it < 20; ++j, ++it) {
                float A[36], b[6], residual[2];
                const float* p12 = x->probePoses[q][it];
                mfo_icp_step(p12, p12 + 9, cv[lv], cn[lv], Rpi, t, g->fx / div, g->fy / div, g->cx / div, g->cy / div, pv[lv], pn[lv], 0.10f,
                             sinf(20.f * 3.14159254f / 180.f), W >> lv, H >> lv, A, b, residual);
                float* row = m->probeLog[it];
                int k = 0;
                for (int a2 = 0; a2 < 6; ++a2) { for (int c2 = a2; c2 < 6; ++c2) row[k++] = A[a2 * 6 + c2]; row[k++] = b[a2]; }
                row[27] = residual[0]; row[28] = residual[1]; row[29] = row[30] = row[31] = 0.f;
            }
        }
        m->probeLogN = it;
    }
    mfo_track_stats st;
    /* only the background model allows fill-in (Model.cpp:400: frameToFrameRGB && allowsFillIn()) */
    track_model(g, &x->rs, cv, cn, pv, pn, doFillIn ? x->fillVertex : m->predVertex,
                (doFillIn || (x->frameToFrameRGB && m == &x->models[0])) ? x->fillImage : m->predImage,
                x->rgb, m->lastNext, R, t, inc, &st);
    m->lastICPError = st.lastICPError; m->lastICPCount = st.lastICPCount;
    m->trackLogN = mfo_last_track_log(&m->trackLog[0][0]);
    Rt_to_pose16(R, t, m->pose);
    return sqrtf(inc[12] * inc[12] + inc[13] * inc[13] + inc[14] * inc[14]);
}

/* the probe: runs the step from the shifted start, keeps the result in trackedPoseAlt and restores every piece of state the step touches */
static void mm_track_probe(mfo_mm* x, mm_model* m, int allowFillIn) {
    const int W = x->cfg.base.W, H = x->cfg.base.H;
    float pose0[16], last0[16];
    memcpy(pose0, m->pose, sizeof(pose0)); memcpy(last0, m->lastPose, sizeof(last0));
    const float e0 = m->lastICPError, c0 = m->lastICPCount;
    uint8_t* keep[3];
    for (int i = 0; i < 3; ++i) {
        const size_t n = (size_t)(W >> i) * (H >> i);
        keep[i] = (uint8_t*)malloc(n); memcpy(keep[i], m->lastNext[i], n);
    }
    m->pose[12] += 1e-6f;
    mm_track(x, m, allowFillIn);
    memcpy(m->trackedPoseAlt, m->pose, sizeof(m->pose));
    if (getenv("MFO_PROBE_DEBUG")) {   /* exploratory: spread of the step over several micrometre shifts */
        float ref[16]; memcpy(ref, m->pose, sizeof(ref));
        for (int q = 0; q < 8; ++q) {
            memcpy(m->pose, pose0, sizeof(pose0));
            m->pose[12 + q % 3] += (q & 1 ? -1.f : 1.f) * (q < 4 ? 1e-6f : 1e-7f) * (1 + q);
            mm_track(x, m, allowFillIn);
            float d = 0; for (int k = 0; k < 16; ++k) d = fmaxf(d, fabsf(m->pose[k] - ref[k]));
            fprintf(stderr, "probe model %d q %d: |step - step0| %.3e\n", m->id, q, d);
        }
    }
    memcpy(m->pose, pose0, sizeof(pose0)); memcpy(m->lastPose, last0, sizeof(last0));
    m->lastICPError = e0; m->lastICPCount = c0;
    for (int i = 0; i < 3; ++i) { memcpy(m->lastNext[i], keep[i], (size_t)(W >> i) * (H >> i)); free(keep[i]); }
}

static void mm_predict_indices(mfo_mm* x, mm_model* m, const float* surf) {
    const mfo_config* g = &x->cfg.base;
    mfo_predict_indices(&x->cam, m->pose, surf, m->count, x->tick, g->maxDepthProcessed, g->timeDelta, x->index, x->ivc,
                        x->ict, x->inr);
}
/* predictIndices -> fuse -> predictIndices -> clean for one model */
static void mm_fuse_clean(mfo_mm* x, mm_model* m, float fuseDepthCutoff, float weightMultiplier, int secondIndexPass) {
    const mfo_config* g = &x->cfg.base;
    const int src = m->cur, dst = 1 - m->cur;
    mm_predict_indices(x, m, m->surf[src]);
    const float weighting = mfo_fusion_weight(m->pose, m->lastPose, weightMultiplier);
    float md = fminf(fuseDepthCutoff, m->maxDepth); /* Model.cpp:527 */
    if (m->id != 0 && x->bboxLimit && m->bbox[0] <= m->bbox[3] && m->bbox[1] <= m->bbox[4] && m->bbox[2] <= m->bbox[5]) {
        /* Model.cpp:480-501: the two corners of lastBoundingBox in the camera frame (pose^-1), z only, + 5 % */
        float R[9], t[3], Ri[9], ti[3];
        pose16_to_Rt(m->pose, R, t);
        pose_inverse_Rt(R, t, Ri, ti);
        const float bbscale = 0.001f;
        const float zmin = ((Ri[6] * (bbscale * (float)m->bbox[0]) + Ri[7] * (bbscale * (float)m->bbox[1])) + Ri[8] * (bbscale * (float)m->bbox[2])) + ti[2];
        const float zmax = ((Ri[6] * (bbscale * (float)m->bbox[3]) + Ri[7] * (bbscale * (float)m->bbox[4])) + Ri[8] * (bbscale * (float)m->bbox[5])) + ti[2];
        const float lo = zmin < zmax ? zmin : zmax, hi = zmin < zmax ? zmax : zmin;
        md = fminf(md, hi + 0.05f * fabsf(hi - lo));
    }
    mfo_fuse_data(&x->cam, m->pose, x->rgb, x->depth, x->depthF, x->mask, m->id, x->tick, weighting, md, x->index, x->ivc,
                  x->inr, x->cand_op, x->cand_best, x->cand_rec, &x->n_cand);
    mfo_fuse_update(m->surf[src], m->surf[dst], m->count, x->tick, x->cand_op, x->cand_best, x->cand_rec, x->n_cand);
    if (secondIndexPass) mm_predict_indices(x, m, m->surf[dst]);
    m->count = mfo_clean(&x->cam, m->pose, m->surf[dst], m->count, x->cand_op, x->cand_rec, x->n_cand, x->tick,
                         g->timeDelta, m->confThr, fminf(g->maxDepthProcessed, m->maxDepth), g->outlierCoeff, m->id,
                         x->index, x->ivc, x->ict, x->inr, x->depthF, x->mask, m->surf[src], m->cap);
}

int mfo_mm_process_frame(mfo_mm* x, const uint8_t* rgb, const float* depth, const uint8_t* mask, const int32_t* classIDs,
                         int nMasks, float weightMultiplier) {
    const mfo_mm_config* cfg = &x->cfg;
    const mfo_config* g = &cfg->base;
    const int W = g->W, H = g->H, P = W * H;
    memcpy(x->rgb, rgb, (size_t)P * 3);
    memcpy(x->depth, depth, sizeof(float) * P);
    if (x->depthF_override) { memcpy(x->depthF, x->depthF_override, sizeof(float) * P); x->depthF_override = NULL; }
    else mfo_bilateral(x->depth, x->depthF, W, H);
    mm_model* bg = &x->models[0];
    if (x->tick == 1) {
        bg->cur = 0;
        bg->count = mfo_init_surfels(&x->cam, x->rgb, x->depth, x->depthF, x->tick, g->maxDepthProcessed, bg->surf[0], bg->cap);
        first_rgb(x->rgb, W, H, bg->lastNext);
    } else {
        memcpy(x->depthPyr[0], x->depthF, sizeof(float) * P);
        for (int i = 1; i < 3; ++i) mfo_pyrdown_gauss_f(x->depthPyr[i - 1], x->depthPyr[i], W >> (i - 1), H >> (i - 1));
        for (int i = 0; i < 3; ++i) {
            const int div = 1 << i;
            mfo_create_vmap(x->depthPyr[i], x->vmap[i], W >> i, H >> i, g->fx / div, g->fy / div, g->cx / div, g->cy / div, g->depthCutoff);
            mfo_create_nmap(x->vmap[i], x->nmap[i], W >> i, H >> i);
        }
        /* tracking, :247-276 */
        const int forced = x->forceOn;
        x->forceOn = 0;
        for (int i = 0; i < x->nModels; ++i) x->models[i].trackedThisFrame = 0;
        if (forced) mm_track_probe(x, bg, 1);
        mm_track(x, bg, 1);
        memcpy(bg->trackedPose, bg->pose, sizeof(bg->pose)); bg->trackedThisFrame = 1;
        if (forced && mm_forced_pose(x, bg->id)) memcpy(bg->pose, mm_forced_pose(x, bg->id), sizeof(bg->pose));
        for (int i = 1; i < x->nModels; ++i) {
            mm_model* m = &x->models[i];
            int trackable = x->nTrackable == 0;   /* trackableClassIds.empty() || trackableClassIds.count(classID), :261 */
            for (int q = 0; q < x->nTrackable; ++q) trackable |= (x->trackable[q] == m->classID);
            if ((!m->isStatic || cfg->trackAllModels) && trackable) {
                if (forced) mm_track_probe(x, m, 0);
                const float d = mm_track(x, m, 0);
                memcpy(m->trackedPose, m->pose, sizeof(m->pose)); m->trackedThisFrame = 1;
                const float* fp = forced ? mm_forced_pose(x, m->id) : NULL;
                if (fp) memcpy(m->pose, fp, sizeof(m->pose));
                if (forced ? (fp == NULL) : (d >= 0.2f)) { /* inactivateModel, :268-272: `float d > 0.2` compares in double -- 0.2f is the smallest float above 0.2 */
                    mm_model_free(m);
                    memmove(&x->models[i], &x->models[i + 1], sizeof(mm_model) * (size_t)(x->nModels - i - 1));
                    x->nModels--; i--;
                }
            } else { /* updateStaticPose: pose = initialC2Winv * globalPose (Model.h:263) */
                memcpy(m->lastPose, m->pose, sizeof(m->pose));
                mat4_mul_cm(m->initialC2Winv, bg->pose, m->pose);
            }
        }
        /* segmentation, :287-375 */
        {
            mfo_model_view views[64];
            int32_t ids[64], cls[64];
            for (int i = 0; i < x->nModels; ++i) {
                views[i].surfels = x->models[i].surf[x->models[i].cur]; views[i].count = x->models[i].count;
                views[i].pose16 = x->models[i].pose; views[i].id = x->models[i].id;
                ids[i] = x->models[i].id; cls[i] = x->models[i].classID;
            }
            mfo_global_projection(&x->cam, views, x->nModels, x->tick, x->tick, g->timeDelta, g->depthCutoff, x->projIDs);
            if (x->spawnOffset < cfg->modelSpawnOffset) x->spawnOffset++;
            /* MfSegmentation::performSegmentation, GPU half: :149-208 */
            mfo_geometric_edge_map(x->vmap[0], x->nmap[0], x->edge, W, H, cfg->seg.weightDistance, cfg->seg.weightConvexity);
            mfo_threshold_map(x->edge, x->binEdge, P, cfg->seg.threshold);
            mfo_morph_closing_u8(x->binEdge, x->ucharBuf, W, H, cfg->seg.morphEdgeRadius, cfg->seg.morphEdgeIterations);
            mfo_invert_map(x->binEdge, x->ucharBuf, P);
            int hasNew = 0, newClass = -1;
            static const int32_t zeroClass[1] = {0};
            mfo_mf_segmentation_cpu(&cfg->seg, W, H, x->ucharBuf, x->depth, mask ? mask : x->mask /*unused when nMasks==0*/,
                                    nMasks ? classIDs : zeroClass, mask ? nMasks : 0, x->projIDs, ids, cls, x->nModels,
                                    x->nextID, x->spawnOffset >= cfg->modelSpawnOffset, x->ignoreMap, x->fullSeg, &hasNew,
                                    &newClass);
            memcpy(x->mask, x->fullSeg, P); /* textureMask upload, :297 */
            if (hasNew && x->nModels < cfg->maxModels) {
                mm_model* nm = &x->models[x->nModels];
                mm_model_init(x, nm, x->nextID, cfg->confObject, cfg->capacityObject);
                /* getNextModelID(true), :715-731 */
                for (;;) {
                    x->nextID = (x->nextID + 1) & 255;
                    int occ = 0;
                    for (int i = 0; i < x->nModels; ++i) occ |= x->models[i].id == x->nextID;
                    if (!occ) break;
                }
                /* makeStatic(globalPose): initialC2Winv = pose * globalPose^-1 */
                float ginv[16];
                mat4_rigid_inverse_cm(bg->pose, ginv);
                mat4_mul_cm(nm->pose, ginv, nm->initialC2Winv);
                nm->isStatic = 1;
                nm->classID = newClass;
                first_rgb(x->rgb, W, H, nm->lastNext);  /* spawnObjectModel: initFirstRGB, :680 */
                x->spawnOffset = 0;
                x->nModels++;
            } else hasNew = 0;
            for (int i = 1; i < x->nModels; ++i) x->models[i].maxDepth = 30.f + 30.f * 1.2f; /* :335-339 */
            if (hasNew) { /* :342-359: predictIndices; fuse(maxDepthProcessed, weight 100); clean -- no second index pass */
                mm_fuse_clean(x, &x->models[x->nModels - 1], g->maxDepthProcessed, 100.f, 0);
            }
            for (int i = 1; i < x->nModels; ++i) x->models[i].confThr = fminf(4.5f, x->models[i].age / 25.0f); /* :369-374 */
        }
        /* fusion, :539-565 */
        if (!g->rgbOnly)   /* :539: if (!rgbOnly && trackingOk && !lost); the spawn-frame fuse above is not under this guard */
            for (int i = 0; i < x->nModels; ++i) mm_fuse_clean(x, &x->models[i], g->depthCutoff, weightMultiplier, 1);
    }
    /* predict(), :569 */
    for (int i = 0; i < x->nModels; ++i) {
        mm_model* m = &x->models[i];
        mfo_combined_predict(&x->cam, m->pose, m->surf[m->cur], m->count, g->maxDepthProcessed, m->confThr, x->tick, x->tick,
                             g->timeDelta, m->predImage, m->predVertex, m->predNormal, m->predTime);
        if (i == 0)
            mfo_fill_in(&x->cam, m->predImage, m->predVertex, m->predNormal, x->rgb, x->depthF, x->frameToFrameRGB ? 2 : 0, x->fillImage,
                        x->fillVertex, x->fillNormal);
    }
    /* the GUI's render pass after processFrame (GUI/MainController.cpp:704-717): lastBoundingBox of every object model */
    for (int i = 1; i < x->nModels; ++i) mm_bbox_update(&x->models[i]);
    x->tick++;
    for (int i = 0; i < x->nModels; ++i) x->models[i].age++;
    x->nProbe = 0;
    return 0;
}

/* Test tooling, the twin of mf_model_upload_map (include/maskfusion_amd.h): replace the live surfel buffer of model i with `count` records
 * (12 floats each).  No upstream twin -- a long orbit would fill the map; the configs[4] parity / stress case loads a pre-filled map instead
 * (SURVEY.md 8d S3).  Returns 0, or -1 when the model does not exist or the records do not fit its capacity. */
int mfo_mm_upload_map(mfo_mm* x, int i, const float* surfels12, int count) {
    if (i < 0 || i >= x->nModels || count < 0 || count > x->models[i].cap) return -1;
    mm_model* m = &x->models[i];
    memcpy(m->surf[m->cur], surfels12, sizeof(float) * 12 * (size_t)count);
    m->count = count;
    return 0;
}
int mfo_mm_num_models(const mfo_mm* x) { return x->nModels; }
int mfo_mm_model_id(const mfo_mm* x, int i) { return x->models[i].id; }
int mfo_mm_model_count(const mfo_mm* x, int i) { return x->models[i].count; }
void mfo_mm_model_pose(const mfo_mm* x, int i, float* p) { memcpy(p, x->models[i].pose, sizeof(float) * 16); }
const float* mfo_mm_model_surfels(const mfo_mm* x, int i) { return x->models[i].surf[x->models[i].cur]; }
const uint8_t* mfo_mm_segmentation(const mfo_mm* x) { return x->fullSeg; }
const uint8_t* mfo_mm_projected_ids(const mfo_mm* x) { return x->projIDs; }
const float* mfo_mm_edge_map(const mfo_mm* x) { return x->edge; }
/* debug taps: model-side pyramid of the model tracked LAST in the last frame (planar [3][h][w]); the current frame's maps */
const float* mfo_mm_dbg_map(const mfo_mm* x, int which, int level) {
    switch (which) { case 0: return x->vmap_g[level]; case 1: return x->nmap_g[level]; case 2: return x->vmap[level]; default: return x->nmap[level]; }
}
const float* mfo_mm_dbg_pred(const mfo_mm* x, int model, int which) { return which == 0 ? x->models[model].predVertex : x->models[model].predNormal; }
