/*
 * mfgl.h -- just enough of GLSL 3.30 as C++ types and functions for the reference's own shader sources
 * (Core/Shaders/*.vert / *.frag / *.glsl of martinruenz/maskfusion) to compile with plain g++ and run on the CPU.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as oracle/mf_oracle.h and oracle/ref_shim/): nothing of the product includes this.  The
 * shader sources are NOT copied into the repository: oracle/build_glsl.py reads them where they lie under /root/reference,
 * resolves their #include lines the way pangolin's loader does, strips what is not C++ (#version, layout(...), the in / out /
 * uniform / flat qualifiers), gives every floating literal an `f` suffix (GLSL has no double; C++ would otherwise promote
 * `1.0 / cols` to double), wraps each shader in its own namespace and pipes the text to g++ on stdin together with
 * mfgl_api.cpp.  Output: oracle/_ref/libmf_glsl.so.
 *
 * What this pins and what it does not.  Everything a shader COMPUTES -- conditions, thresholds, loop bounds (the fp32 induction
 * variables of the association windows included), operation order -- is executed from the reference's text, with one rounding
 * per operation (-ffp-contract=off).  What OpenGL does AROUND a shader is the documented rule set of the oracle:
 * nearest texel = floor(u * width) clamped to the edge, a point lands in the pixel that contains it, sprites cover pixel centres
 * in [u - s/2, u + s/2), depth test LESS with the earlier primitive winning ties.  exp() / acos() are vendor-defined in GLSL:
 * they are routed to the oracle's shared fp32 polynomials so that a confidence or an angle never differs in the last bit.
 */
#ifndef MFGL_H_
#define MFGL_H_

#include <math.h>
#include <stdint.h>
#include <string.h>

extern "C" float mfo_shader_exp(float x);   /* oracle/mf_oracle.c */
extern "C" float mfo_shader_acos(float x);
extern "C" void mfo_pose_inverse16(const float* pose16, float* out16);

namespace mfgl {

typedef unsigned int uint;
struct vec2;
struct vec3;
struct vec4;

/* swizzle proxies: views onto the components of the vector they live in (a union member of it) */
template <class V2, int A, int B>
struct swz2 {
    float d[4];
    operator V2() const { return V2(d[A], d[B]); }
    swz2& operator=(const V2& v) { d[A] = v.x; d[B] = v.y; return *this; }
};
template <class V3, int A, int B, int C>
struct swz3 {
    float d[4];
    operator V3() const { return V3(d[A], d[B], d[C]); }
    swz3& operator=(const V3& v) { d[A] = v.x; d[B] = v.y; d[C] = v.z; return *this; }
};

struct vec2 {
    union {
        struct { float x, y; };
        swz2<vec2, 0, 1> xy;
    };
    vec2() : x(0), y(0) {}
    vec2(float a, float b) : x(a), y(b) {}
    vec2(const vec2& o) : x(o.x), y(o.y) {}
    vec2& operator=(const vec2& o) { x = o.x; y = o.y; return *this; }
    explicit vec2(const vec4& v);
};
struct vec3 {
    union {
        struct { float x, y, z; };
        swz2<vec2, 0, 1> xy;
        swz3<vec3, 0, 1, 2> xyz;
    };
    vec3() : x(0), y(0), z(0) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    vec3(const vec2& v, float c) : x(v.x), y(v.y), z(c) {}
    vec3(const vec3& o) : x(o.x), y(o.y), z(o.z) {}
    vec3& operator=(const vec3& o) { x = o.x; y = o.y; z = o.z; return *this; }
    vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
};
struct vec4 {
    union {
        struct { float x, y, z, w; };
        swz2<vec2, 0, 1> xy;
        swz2<vec2, 2, 3> zw;
        swz3<vec3, 0, 1, 2> xyz;
    };
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    vec4(const vec4& o) : x(o.x), y(o.y), z(o.z), w(o.w) {}
    vec4& operator=(const vec4& o) { x = o.x; y = o.y; z = o.z; w = o.w; return *this; }
    explicit operator float() const { return x; }   /* float(textureLod(...)) */
};
inline vec2::vec2(const vec4& v) : x(v.x), y(v.y) {}
struct uvec4 {
    uint x, y, z, w;
    explicit operator uint() const { return x; }     /* uint(textureLod(usampler2D, ...)) */
};

/* component-wise arithmetic, one rounding per operation */
inline vec2 operator+(const vec2& a, const vec2& b) { return vec2(a.x + b.x, a.y + b.y); }
inline vec2 operator-(const vec2& a, const vec2& b) { return vec2(a.x - b.x, a.y - b.y); }
inline vec2 operator/(const vec2& a, const vec2& b) { return vec2(a.x / b.x, a.y / b.y); }
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, const vec3& a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(const vec3& a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec4 operator+(const vec4& a, const vec4& b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator*(float s, const vec4& a) { return vec4(s * a.x, s * a.y, s * a.z, s * a.w); }
inline vec4 operator/(const vec4& a, float s) { return vec4(a.x / s, a.y / s, a.z / s, a.w / s); }

inline float dot(const vec2& a, const vec2& b) { return a.x * b.x + a.y * b.y; }
inline float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(const vec3& a, const vec3& b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float sqrt(float v) { return sqrtf(v); }
inline float length(const vec3& a) { return sqrtf(dot(a, a)); }
inline vec3 normalize(const vec3& a) { const float l = sqrtf(dot(a, a)); return vec3(a.x / l, a.y / l, a.z / l); }   /* v / length(v) */
inline float abs(float v) { return fabsf(v); }
inline int abs(int v) { return v < 0 ? -v : v; }
inline float min(float a, float b) { return b < a ? b : a; }   /* GLSL: y < x ? y : x */
inline float max(float a, float b) { return a < b ? b : a; }   /* GLSL: x < y ? y : x */
inline int min(int a, int b) { return b < a ? b : a; }
inline int max(int a, int b) { return a < b ? b : a; }
inline float min(int a, float b) { return min((float)a, b); }
inline float max(int a, float b) { return max((float)a, b); }
inline float min(float a, int b) { return min(a, (float)b); }
inline float max(float a, int b) { return max(a, (float)b); }
inline float pow(float a, float b) { return powf(a, b); }
inline float round(float v) { return roundf(v); }
int exp_uses_libm();   /* mfgl_api.cpp: the oracle restates the bilateral filter's exp() with the C library's expf */
inline float exp(float v) { return exp_uses_libm() ? expf(v) : mfo_shader_exp(v); }
inline float acos(float v) { return mfo_shader_acos(v); }

/* matrices: column-major like GLSL (c[k] is a column) */
struct mat4 {
    float m[16];   /* column-major: m[col * 4 + row] -- exactly what Eigen::Matrix4f::data() / glUniformMatrix4fv hand over */
};
struct mat3 {
    vec3 c[3];
    mat3() {}
    mat3(const vec3& a, const vec3& b, const vec3& d) { c[0] = a; c[1] = b; c[2] = d; }
    explicit mat3(const mat4& M) {   /* GLSL mat3(mat4): the upper-left 3x3 */
        c[0] = vec3(M.m[0], M.m[1], M.m[2]); c[1] = vec3(M.m[4], M.m[5], M.m[6]); c[2] = vec3(M.m[8], M.m[9], M.m[10]);
    }
};
inline vec3 operator*(const mat3& M, const vec3& v) {
    return vec3(M.c[0].x * v.x + M.c[1].x * v.y + M.c[2].x * v.z, M.c[0].y * v.x + M.c[1].y * v.y + M.c[2].y * v.z,
                M.c[0].z * v.x + M.c[1].z * v.y + M.c[2].z * v.z);
}
inline vec4 operator*(const mat4& M, const vec4& v) {
    return vec4(M.m[0] * v.x + M.m[4] * v.y + M.m[8] * v.z + M.m[12] * v.w, M.m[1] * v.x + M.m[5] * v.y + M.m[9] * v.z + M.m[13] * v.w,
                M.m[2] * v.x + M.m[6] * v.y + M.m[10] * v.z + M.m[14] * v.w, M.m[3] * v.x + M.m[7] * v.y + M.m[11] * v.z + M.m[15] * v.w);
}
inline mat3 transpose(const mat3& M) {
    return mat3(vec3(M.c[0].x, M.c[1].x, M.c[2].x), vec3(M.c[0].y, M.c[1].y, M.c[2].y), vec3(M.c[0].z, M.c[1].z, M.c[2].z));
}
inline mat3 inverse(const mat3& M) {   /* only reached by the deformation block of copy_unstable.vert, which is inert (nodes == 0) */
    const vec3 a = M.c[0], b = M.c[1], c = M.c[2];
    const vec3 r0 = cross(b, c), r1 = cross(c, a), r2 = cross(a, b);
    const float inv = 1.0f / dot(a, r0);
    return transpose(mat3(r0 * inv, r1 * inv, r2 * inv));
}
}  // namespace mfgl

namespace mfgl {
/* samplers: nearest, clamp to edge; u, v normalised.  The scaled coordinate u * width is snapped to 1/256 texel (round to nearest)
 * before the floor -- texture units work in fixed point with 8 fractional bits -- so a coordinate that is meant to sit ON a texel
 * edge (the bilateral filter samples at cx / cols, the association windows at half-pixel steps) selects the texel the exact value
 * would, instead of flipping with the last bit of the fp32 division.  NaN -> texel 0. */
struct sampler2D {
    const float* data; int w, h, c;   /* c floats per texel (1 or 4) */
};
struct usampler2D {
    const uint32_t* data; int w, h;
};
inline int texel_index(float u, int n) {
    if (!(u == u)) return 0;
    float t = floorf(rintf(u * (float)n * 256.0f) * (1.0f / 256.0f));
    if (t < 0.f) t = 0.f;
    if (t > (float)(n - 1)) t = (float)(n - 1);
    return (int)t;
}
inline vec4 textureLod(const sampler2D& s, const vec2& uv, float) {
    const int ix = texel_index(uv.x, s.w), iy = texel_index(uv.y, s.h);
    const float* p = s.data + ((size_t)iy * s.w + ix) * s.c;
    return s.c == 4 ? vec4(p[0], p[1], p[2], p[3]) : vec4(p[0], p[0], p[0], 1.0f);
}
inline uvec4 textureLod(const usampler2D& s, const vec2& uv, float) {
    const int ix = texel_index(uv.x, s.w), iy = texel_index(uv.y, s.h);
    uvec4 r;
    r.x = s.data[(size_t)iy * s.w + ix]; r.y = r.z = 0; r.w = 1;
    return r;
}
inline vec4 texture(const sampler2D& s, const vec2& uv) { return textureLod(s, uv, 0.f); }
inline vec4 texture2D(const sampler2D& s, const vec2& uv) { return textureLod(s, uv, 0.f); }

/* pipeline built-ins, set / read by the harness around every main() */
extern vec4 gl_Position;
extern vec4 gl_FragCoord;
extern float gl_PointSize;
extern float gl_FragDepth;
extern int gl_VertexID;
extern bool g_discarded;
}  // namespace mfgl

#define discard do { mfgl::g_discarded = true; return; } while (0)

#endif /* MFGL_H_ */
