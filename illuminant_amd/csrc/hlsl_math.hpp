// hlsl_math.hpp -- device-side scalar helpers with the HLSL semantics the
// reference shaders rely on (saturate, lerp, sign, float %, normalize) plus the
// shared SDF atlas sampler.  gfx950 only; compiled with -ffp-contract=off so
// that +,-,*,/ and sqrt round exactly like the CPU oracle's and only the
// transcendental functions (sin/cos/acos/atan2/pow) differ by a few ulp.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/illuminant_hip.h"

namespace ilm {

#define ILM_DEV __device__ __forceinline__

constexpr float kPi = 3.14159265358979323846f;      // ParticleCommon.fxh:23
constexpr float kVelocityConstantScale = 1000.0f;   // ParticleCommon.fxh:24
constexpr float kDistanceZero = 192.0f / 255.0f;    // DistanceFieldCommon.fxh:8

struct f3 { float x, y, z; };

ILM_DEV float sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
ILM_DEV float lerp(float a, float b, float t) { return a + (b - a) * t; }
// lerp with the multiply and the add rounded separately even when the translation unit allows FMA
// contraction: used where the result is a life value (liveness must be bit-exact)
ILM_DEV float lerp_exact(float a, float b, float t) {
#pragma clang fp contract(off)
    return a + (b - a) * t;
}
ILM_DEV float sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
ILM_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

ILM_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
ILM_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
ILM_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
ILM_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
ILM_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
ILM_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
ILM_DEV float len3(f3 a) { return sqrtf(dot3(a, a)); }
ILM_DEV f3 norm3(f3 a) { float l = len3(a); return mk3(a.x / l, a.y / l, a.z / l); }
ILM_DEV f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
ILM_DEV f3 abs3(f3 a) { return mk3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
ILM_DEV f3 max03(f3 a) { return mk3(fmaxf(a.x, 0.0f), fmaxf(a.y, 0.0f), fmaxf(a.z, 0.0f)); }

ILM_DEV float4 mk4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
ILM_DEV float4 add4(float4 a, float4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
ILM_DEV float4 sub4(float4 a, float4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
ILM_DEV float4 mul4(float4 a, float4 b) { return mk4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
ILM_DEV float4 lerp4(float4 a, float4 b, float t) { return mk4(lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t), lerp(a.w, b.w, t)); }
ILM_DEV float4 ld4(const IlmFloat4& v) { return mk4(v.x, v.y, v.z, v.w); }
ILM_DEV f3 xyz(float4 a) { return mk3(a.x, a.y, a.z); }

// Approximate (about 1 ulp) reciprocal / rsqrt / sqrt: single VALU instructions.  Used only where
// the result feeds neither a table index, a slot index nor a life value (see DESIGN.md "numerics");
// everything index- or liveness-critical uses the IEEE operators above.
ILM_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
ILM_DEV float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
ILM_DEV float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
ILM_DEV float len3_fast(f3 a) { return fast_sqrt(dot3(a, a)); }
ILM_DEV f3 norm3_fast(f3 a) { const float r = fast_rsq(dot3(a, a)); return mk3(a.x * r, a.y * r, a.z * r); }

// positive modulo of an integer-valued float without integer division.  Domain: |t| < 2^23, which the
// callers guarantee (randomness-table coordinates are bounded by the table offsets a*253, b*127 and the
// per-index offsets < 65531).  q may be off by one after the reciprocal multiply; the remainder t - q*size is
// exact in fp32 and is folded back into [0, size), so the result equals the integer modulo bit for bit.
ILM_DEV int wrap_index_fast(float t, int size) {
#pragma clang fp contract(off)
    const float fs = (float)size;
    const float q = floorf(t * fast_rcp(fs));
    float r = t - q * fs;
    r = (r < 0.0f) ? r + fs : r;
    r = (r >= fs) ? r - fs : r;
    return (int)r;
}

// sign(d) * m for m >= 0 (HLSL sign() is 0 at 0)
ILM_DEV float sign_times(float d, float m) { return (d == 0.0f) ? 0.0f * m : copysignf(m, d); }

// positive modulo of a float tap index (D3D WRAP addressing)
ILM_DEV int wrap_index(float t, int size) {
    int i = (int)t;
    i %= size;
    if (i < 0) i += size;
    return i;
}

// ---------------------------------------------------------------------------------------------
// SDF atlas sampling -- sampleDistanceFieldEx, DistanceFieldCommon.fxh:313-353.
// The atlas keeps the reference layout (RGBA, 16 bit per channel, three virtual
// z-slices + the first of the next triplet per texel); one sample = 4 bilinear
// taps of 8 bytes, two channels used.
// ---------------------------------------------------------------------------------------------
struct SdfView {
    const uint2* texels;   // one RGBA16 texel = 8 bytes
    int width, height;
    int format;            // ILM_SDF_UNORM16 / ILM_SDF_FP16
};

template <int FORMAT>
ILM_DEV void sdf_unpack2(uint2 t, int pair, float& a, float& b) {
    // pair 0: (r,g)  1: (g,b)  2: (b,a)
    uint32_t lo, hi;
    if (pair == 0) { lo = t.x & 0xFFFFu; hi = t.x >> 16; }
    else if (pair == 1) { lo = t.x >> 16; hi = t.y & 0xFFFFu; }
    else { lo = t.y & 0xFFFFu; hi = t.y >> 16; }
    if (FORMAT == ILM_SDF_FP16) {
        a = __half2float(__ushort_as_half((unsigned short)lo));
        b = __half2float(__ushort_as_half((unsigned short)hi));
    } else {
        a = (float)lo / 65535.0f;
        b = (float)hi / 65535.0f;
    }
}

template <int FORMAT>
ILM_DEV float sample_distance_field(f3 position, const IlmDistanceFieldUniforms& df, const SdfView& sdf) {
#pragma clang fp contract(off)
    position.z -= df.ConeAndMisc.y;
    const float ex = df.Extent.x, ey = df.Extent.y, ez = df.Extent.z;
    const float cx = clampf(position.x, 0.0f, ex), cy = clampf(position.y, 0.0f, ey), cz = clampf(position.z, 0.0f, ez);
    const f3 dtv = mk3(-fminf(position.x, 0.0f) + (fmaxf(position.x, ex) - ex),
                       -fminf(position.y, 0.0f) + (fmaxf(position.y, ey) - ey),
                       -fminf(position.z, 0.0f) + (fmaxf(position.z, ez) - ez));
    const float distance_to_volume = len3(dtv);

    const float slice_position = fminf(cz, df.Packed1.z) * df.Packed1.y;
    const float vslice = floorf(slice_position);

    const float column_index = floorf(vslice / 3.0f);
    const float row_index = floorf(vslice * df.Packed1.x);
    const float u = column_index * df.TextureSliceAndTexelSize.x + cx * df.TextureSliceAndTexelSize.z;
    const float v = row_index * df.TextureSliceAndTexelSize.y + cy * df.TextureSliceAndTexelSize.w;

    // LINEAR, U WRAP, V CLAMP, texel centres at +0.5
    const float x = u * (float)sdf.width - 0.5f;
    const float y = v * (float)sdf.height - 0.5f;
    const float x0f = floorf(x), y0f = floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    const int x0 = wrap_index(x0f, sdf.width), x1 = wrap_index(x0f + 1.0f, sdf.width);
    int y0 = (int)y0f, y1 = (int)y0f + 1;
    y0 = min(max(y0, 0), sdf.height - 1);
    y1 = min(max(y1, 0), sdf.height - 1);

    const uint2* row0 = sdf.texels + (size_t)y0 * (size_t)sdf.width;
    const uint2* row1 = sdf.texels + (size_t)y1 * (size_t)sdf.width;
    const uint2 t00 = row0[x0], t10 = row0[x1], t01 = row1[x0], t11 = row1[x1];

    const float m = fmodf(vslice, 3.0f);
    const int pair = (m >= 2.0f) ? 2 : ((m >= 1.0f) ? 1 : 0);
    float a00, b00, a10, b10, a01, b01, a11, b11;
    sdf_unpack2<FORMAT>(t00, pair, a00, b00);
    sdf_unpack2<FORMAT>(t10, pair, a10, b10);
    sdf_unpack2<FORMAT>(t01, pair, a01, b01);
    sdf_unpack2<FORMAT>(t11, pair, a11, b11);
    const float lo = lerp(lerp(a00, a10, fx), lerp(a01, a11, fx), fy);
    const float hi = lerp(lerp(b00, b10, fx), lerp(b01, b11, fx), fy);
    const float blended = lerp(lo, hi, slice_position - vslice);

    return (kDistanceZero - blended) * df.Extent.w + distance_to_volume;
}

}  // namespace ilm
