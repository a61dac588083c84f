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
#include "reference_constants.hpp"

namespace ilm {

#define ILM_DEV __device__ __forceinline__

using ref::kPi;                       // the reference's constants by name: reference_constants.hpp
using ref::kVelocityConstantScale;
using ref::kDistanceZero;

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

// pow(x, y) for x >= 0 as exp2(y * log2(x)): v_log_f32 / v_exp_f32 are 1-ulp hardware transcendentals, so the result is within
// |y log2 x| * 2^-23 relative of powf (a few 1e-7 for the bases in [0, 1] and exponents in use) -- far inside the 1e-4 parity
// tolerance -- at 3 instructions instead of OCML powf's ~160.  pow(0, y > 0) = exp2(-inf) = 0 as powf gives; pow(x, 0) = 1.
ILM_DEV float pow_pos(float x, float y) { return (y == 0.0f) ? 1.0f : __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }

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
    float wf, hf;          // (float)width, (float)height
    float inv_wf;          // 1 / wf (seed of the exact integer wrap; any value within 1 ulp works)
    float wrap_half;       // 0.5 * inv_wf when every tap column this field can ask for stays below 2^20 (make_sdf_view), else 0
    // The in-volume sampler of the cone trace (sample_inside_table below) works from a per-slice table in LDS.  table_slices = number
    // of virtual slices (<= kMaxTableSlices) when the uniforms describe exactly this atlas as columns x rows whole slices
    // (make_sdf_view checks it), else 0: the trace then always uses the general sampler.
    int table_slices;
    int columns;           // slices per atlas row (integer value of TextureSliceCount.x)
    // world-space box inside which a SAMPLE position meets the table sampler's assumptions (InsideBox, make_sdf_view)
    float box_x0, box_x1, box_y0, box_y1, box_z0, box_z1;
    // The particle path's cells (r06; api.hip ensure_slice0_cells, particles.hip build_slice0_cells_kernel): for a UNORM16 field sampled
    // through the slice-0 sampler (DistanceFieldPacked1 = 0: every lookup is the bilinear fetch of channel r of virtual slice 0), one
    // 8-byte cell per (wrapped tap column x0, tap row yi + 1) holding channel r of the FOUR taps -- (x0, y0), (x1, y0), (x0, y1), (x1, y1)
    // with the U WRAP of x1 and the V CLAMP of y0 / y1 folded in -- so that a sample is ONE typed 16_16_16_16 load instead of four
    // 16_16 ones.  Four divergent gathers per lookup are what bounds the collision step -- the texture path serves a wave's 64 scattered
    // addresses one at a time: cfg4's share 369 -> 245 us, cfg2 36 -> 29 us (tools/collision_probe.py).  nullptr: none (the four-tap form).
    const void* cells0;
};
// The cone trace's view: the in-volume sampler reads the field through its CELL array (api.hip make_trace_view, lighting.hip
// build_sdf_cells_kernel): one 16-byte cell per (virtual slice, texel) holding the four bilinear taps of that texel's sample footprint,
// each as the channel pair (slice v, slice v + 1) -- the eight values of one trilinear sample side by side.  Without cells
// table_slices is 0.  (A separate type: the particle step's descriptor carries the plain view and has no room to spare.)
struct TraceSdfView : SdfView {
    const void* cells;
    uint32_t cells_bytes;
    int slice_w, slice_h;
};

// The texture path's own UNORM16 -> f32 conversion.  A typed buffer load (buffer_load_format_xy through a resource of DATA_FORMAT 16_16,
// NUM_FORMAT unorm) returns float(code) / 65535 for both halves of the addressed 4-byte element -- correctly rounded for every one of the
// 65 536 codes, at 2-byte aligned lane offsets too, at the price of an untyped dword load (tools/ubench/unorm.hip, profiles/
// r03_typed_unorm16_loads.txt; on parity: tests/test_sdf_sample_gpu.py walks all codes through both samplers).  The channel pair of a tap
// therefore arrives decoded: no byte permute, no conversions, no x / 65535 (unorm16_to_float below: 4 VALU instructions per channel,
// 32 per sample).  The resource is uniform (4 SGPRs): raw addressing (stride 0), num_records = the atlas in bytes.
typedef float f32x2 __attribute__((ext_vector_type(2)));
extern "C" __device__ f32x2 ilm_llvm_buffer_load_format_xy(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.ptr.buffer.load.format.v2f32");
// word 3 of a gfx9 buffer resource: DST_SEL_X = R (4), DST_SEL_Y = G (5), NUM_FORMAT unorm (0) [14:12], DATA_FORMAT 16_16 (5) [18:15]
constexpr int kRsrcWord3Unorm16x2 = 4 | (5 << 3) | (0 << 12) | (5 << 15);
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern "C" __device__ f32x4 ilm_llvm_buffer_load_format_xyzw(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.ptr.buffer.load.format.v4f32");
// DST_SEL = R, G, B, A; NUM_FORMAT unorm; DATA_FORMAT 16_16_16_16 (12)
constexpr int kRsrcWord3Unorm16x4 = 4 | (5 << 3) | (6 << 6) | (7 << 9) | (0 << 12) | (12 << 15);
// a wave-uniform 64-bit value, said so explicitly (two v_readfirstlane_b32 when it sits in vector registers, nothing when it is scalar already)
ILM_DEV uint64_t uniform_u64(uint64_t p) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p >> 32)) << 32);
}
ILM_DEV __amdgpu_buffer_rsrc_t sdf_unorm_rsrc(const SdfView& sdf) {
    // (the view is a kernel argument, i.e. uniform; said explicitly, because behind divergent control flow the backend otherwise
    // tries to carry the resource through vector registers: "illegal VGPR to SGPR copy")
    return __builtin_amdgcn_make_buffer_rsrc((void*)uniform_u64((uint64_t)sdf.texels), 0, __builtin_amdgcn_readfirstlane((sdf.width * sdf.height) << 3),
                                             kRsrcWord3Unorm16x2);
}

ILM_DEV __amdgpu_buffer_rsrc_t sdf_cells0_rsrc(const SdfView& sdf) {     // (width x (height + 1) cells of 8 bytes)
    return __builtin_amdgcn_make_buffer_rsrc((void*)uniform_u64((uint64_t)sdf.cells0), 0, __builtin_amdgcn_readfirstlane((sdf.width * (sdf.height + 1)) << 3), 4 | (5 << 3) | (6 << 6) | (7 << 9) | (0 << 12) | (12 << 15));
}
ILM_DEV __amdgpu_buffer_rsrc_t sdf_cells_unorm_rsrc(const TraceSdfView& sdf) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)uniform_u64((uint64_t)sdf.cells), 0, __builtin_amdgcn_readfirstlane((int)sdf.cells_bytes), kRsrcWord3Unorm16x4);
}

// x / 65535 for an integer-valued x in [0, 65535], correctly rounded: one multiply by fl(1/65535) and one
// Markstein correction step (two FMAs) instead of the ~10-instruction IEEE division sequence.  Equality with the
// division for all 65536 inputs is checked on the device by tests/test_sdf_sample_gpu.py.
ILM_DEV float unorm16_to_float(float x) {
    const float r = 1.0f / 65535.0f;
    const float q = x * r;
    const float e = __builtin_fmaf(-q, 65535.0f, x);
    return __builtin_fmaf(e, r, q);
}

// One 32-bit word holding the two channels of a texel that virtual slice 3k+m blends: low half = slice m's
// channel, high half = the next one.  m = 0: (r,g)  1: (g,b)  2: (b,a).
// Branch-free: the texel as a 64-bit value shifted right by 16 * m.
ILM_DEV uint32_t sdf_pair_word(uint2 t, uint32_t m) {
    const uint32_t w01 = __builtin_amdgcn_alignbit(t.y, t.x, (m & 1u) << 4);   // m = 0: (r,g)   m = 1: (g,b)
    return (m >= 2u) ? t.y : w01;
}

template <int FORMAT>
ILM_DEV void sdf_unpack_word(uint32_t w, float& a, float& b) {
    if (FORMAT == ILM_SDF_FP16) {
        a = __half2float(__ushort_as_half((unsigned short)(w & 0xFFFFu)));
        b = __half2float(__ushort_as_half((unsigned short)(w >> 16)));
    } else {
        a = unorm16_to_float((float)(w & 0xFFFFu));
        b = unorm16_to_float((float)(w >> 16));
    }
}

// sampleDistanceFieldEx.  Same IEEE operations on every value the result depends on as the CPU oracle (the cone
// trace's loop exits are discontinuous in this value, and the tests require the oracle's exact sample counts).  Multiply-adds are
// FUSED exactly where the oracle calls fmaf (uv assembly, texel-space coordinates, the seven lerps, the decode): the HLSL leaves
// that open and v_fma_f32 halves those chains.  What is rewritten is the integer bookkeeping around them:
//   floor(vslice / 3) and vslice % 3      -> multiply-shift on the integer slice number (exact for vslice < 65536)
//   WRAP / CLAMP tap indices              -> exact float-reciprocal wrap (|index| < 2^23) and integer clamps
//   distance to the volume                -> the sqrt is skipped when every lane of the wave is inside the volume
//   unorm16 decode                        -> unorm16_to_float
// CHECK_NAN = false: the caller guarantees finite coordinates (the cone trace hoists the test out of its loop).
ILM_DEV float lerp_fused(float a, float b, float t) { return __builtin_fmaf(t, b - a, a); }

// n / d, correctly rounded, for 2^-60 <= |d| <= 2^60 and |n| <= 2^60 (or zero / infinite / NaN): the instruction sequence the
// compiler emits for an IEEE division without its two v_div_scale_f32 -- inside that range they do not scale (the scaled and
// unscaled operands are the same bits, v_div_fmas_f32 applies no post-scale), so every intermediate is identical; v_div_fixup_f32
// still patches the zero / infinite / NaN operands.  Checked against `/` on the device by tests/test_sdf_sample_gpu.py.
ILM_DEV float refined_rcp(float d) {
    float y = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
// n / d given y = refined_rcp(d) -- or any y at least as close to 1 / d (the correctly rounded reciprocal): the two correction steps
// land on the correctly rounded quotient, which is unique.  A divisor shared by several numerators (normalize) or fixed per light /
// per kernel pays the three reciprocal instructions once.
ILM_DEV float div_with_rcp(float n, float d, float y) {
    float q = n * y;
    float r = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(r, y, q);
    return __builtin_amdgcn_div_fixupf(q, d, n);
}
ILM_DEV float div_no_scale(float n, float d) { return div_with_rcp(n, d, refined_rcp(d)); }

// The general form: any position (clamped to the volume, distance to the volume added), U WRAP / V CLAMP on the real atlas.
// SLICE0: the uniforms as the reference's PARTICLE path binds them -- DistanceFieldPacked1 left at zero (ParticleSystem.cs
// SetDistanceFieldUniforms; docs/experiments.md 1): zToSliceIndex = 0 puts every lookup at slice position 0 * min(z, maximumValidZ) = 0, i.e. virtual
// slice 0 (column 0, row 0, channel pair (r, g)) with a z weight of 0 -- the result is the bilinear fetch of channel r alone:
// lerp(lo, hi, 0) = fma(0, hi - lo, lo) = lo for every finite lo, hi (up to the sign of a zero that `kDistanceZero - blended` does not
// see).  The caller selects it when Packed1.y == 0 and Packed1.x, .z are finite; one channel per tap, three lerps instead of seven, no
// slice arithmetic: a quarter of the sampler's instructions.
template <int FORMAT, bool CHECK_NAN = true, bool SLICE0 = false, bool CELLS0 = false>
ILM_DEV float sample_distance_field(f3 position, const IlmDistanceFieldUniforms& df, const SdfView& sdf) {
#pragma clang fp contract(off)
    static_assert(!CELLS0 || (SLICE0 && FORMAT == ILM_SDF_UNORM16), "the slice-0 cells serve the slice-0 sampler of a UNORM16 field");
    position.z -= df.ConeAndMisc.y;
    const float ex = df.Extent.x, ey = df.Extent.y, ez = df.Extent.z;
    // clamp3(position, 0, extent) as one median-of-three each.  The distance to the volume per axis,
    // -min(p, 0) + (max(p, e) - e), equals |p - clamp(p, 0, e)| with the same single rounding in every case
    // (p < 0: -p;  p > e: p - e;  inside: 0), and only its square is used.
    // A NaN coordinate (normalize(0) upstream) behaves like 0 in the min/max form -- clamp gives 0 and both terms of
    // the distance vanish -- so it is replaced by 0 up front and the two forms agree on every input.
    if (CHECK_NAN) {
        position.x = (position.x != position.x) ? 0.0f : position.x;
        position.y = (position.y != position.y) ? 0.0f : position.y;
        position.z = (position.z != position.z) ? 0.0f : position.z;
    }
    float distance_to_volume = 0.0f;                 // sqrt(+0) == +0: skipping it inside the volume is exact
    const float cx = __builtin_amdgcn_fmed3f(position.x, 0.0f, ex), cy = __builtin_amdgcn_fmed3f(position.y, 0.0f, ey);
    const float cz = __builtin_amdgcn_fmed3f(position.z, 0.0f, ez);
    {
        const f3 dtv = mk3(position.x - cx, position.y - cy, position.z - cz);
        const float d2 = __builtin_fmaf(dtv.z, dtv.z, __builtin_fmaf(dtv.y, dtv.y, dtv.x * dtv.x));
        if (__builtin_amdgcn_ballot_w64(d2 != 0.0f) != 0ull) {
            asm volatile("" ::: "memory");               // keep this a (wave-uniform) branch: if-conversion would run the sqrt every time
            distance_to_volume = sqrtf(d2);
        }
    }

    float slice_position = 0.0f, vslice = 0.0f;
    uint32_t vi = 0u, third = 0u;
    float u, v;
    if (SLICE0) {
        // column_index = row_index = 0: fma(0, size, t) = t for t >= +0
        u = cx * df.TextureSliceAndTexelSize.z;
        v = cy * df.TextureSliceAndTexelSize.w;
    } else {
        slice_position = (CHECK_NAN ? fminf(cz, df.Packed1.z) : __builtin_elementwise_minimum(cz, df.Packed1.z)) * df.Packed1.y;   // cz is finite
        vslice = floorf(slice_position);
        vi = (uint32_t)vslice;                                   // 0 <= vslice < 65536
        third = __umul24(vi, 0xAAABu) >> 17;                     // vi / 3 (24-bit multiply: full rate)
        const float column_index = (float)third;                 // floor(vslice / 3)
        const float row_index = floorf(vslice * df.Packed1.x);   // floor(vslice * invCols / 3): the reference's float form
        u = __builtin_fmaf(column_index, df.TextureSliceAndTexelSize.x, cx * df.TextureSliceAndTexelSize.z);
        v = __builtin_fmaf(row_index, df.TextureSliceAndTexelSize.y, cy * df.TextureSliceAndTexelSize.w);
    }

    // LINEAR, U WRAP, V CLAMP, texel centres at +0.5
    const float x = __builtin_fmaf(u, sdf.wf, -0.5f);
    const float y = __builtin_fmaf(v, sdf.hf, -0.5f);
    const float x0f = floorf(x), y0f = floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    // U WRAP is what folds physical slice p onto atlas column p % cols (u = p / cols + ... runs past 1 for every atlas row but
    // the first, DistanceFieldCommon.fxh:303-311): a positive modulo of x0f by the atlas width.  q may be off by one after the
    // reciprocal multiply; the remainder x0f - q * width is exact in fp32 and is folded back into [0, width).
    int x0;
    if (sdf.wrap_half > 0.0f) {
        // (x0f + 0.5) / width is at least 0.5 / width away from every integer, and fl(x0f * inv_wf + 0.5 * inv_wf) is within
        // (x0f + 0.5) / width * 2^-23 of it: for x0f < 2^22 the floor is the exact quotient and no fold is needed.  x0f >= -1 (u >= 0).
        const float q = floorf(__builtin_fmaf(x0f, sdf.inv_wf, sdf.wrap_half));
        x0 = (int)__builtin_fmaf(-q, sdf.wf, x0f);
    } else {
        const float q = floorf(x0f * sdf.inv_wf);
        float r = __builtin_fmaf(-q, sdf.wf, x0f);
        r = (r < 0.0f) ? r + sdf.wf : r;
        r = (r >= sdf.wf) ? r - sdf.wf : r;
        x0 = (int)r;
    }
    int x1 = x0 + 1;
    x1 = (x1 == sdf.width) ? 0 : x1;
    // V CLAMP: y0 = clamp(yi, 0, h - 1), y1 = clamp(yi + 1, 0, h - 1); y1 is the next row exactly when 0 <= yi < h - 1
    const int yi = (int)y0f;
    int y0;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(y0) : "v"(yi), "s"(sdf.height - 1));
    const bool next_row = (uint32_t)yi < (uint32_t)(sdf.height - 1);

    // byte offsets from the (uniform) atlas base: 32-bit lane offsets on an SGPR base pointer
    // (atlas <= 8192^2 texels of 8 bytes = 2^29 bytes); one 24-bit multiply, the second row is the first + pitch
    const uint32_t pitch = (uint32_t)sdf.width << 3;
    const uint32_t r0 = __umul24((uint32_t)y0, pitch);
    const uint32_t r1 = next_row ? r0 + pitch : r0;
    // The two channels virtual slice 3k+m blends -- (r,g), (g,b) or (b,a) -- are the 4 bytes at offset 2m inside the 8-byte texel:
    // one dword load per tap at that (2-byte aligned) address (tools/ubench/gather: a 2-byte aligned dword gather costs the same as
    // an aligned one).  2 * (vi % 3) = 2 * vi - 6 * third as one 24-bit multiply-add.
    uint32_t sub = 0u;
    if (!SLICE0) asm("v_mad_i32_i24 %0, %1, -6, %2" : "=v"(sub) : "v"(third), "v"(vi << 1));
    typedef const char __attribute__((address_space(1))) gbyte;
    typedef const uint32_t __attribute__((address_space(1), aligned(2))) gword;
    gbyte* base = (gbyte*)sdf.texels;
    asm("" : "+s"(base));
    const uint32_t c0 = ((uint32_t)x0 << 3) + sub, c1 = ((uint32_t)x1 << 3) + sub;
    float a00, b00, a10, b10, a01, b01, a11, b11;
    if (CELLS0) {
        // one cell = channel r of the four taps, wrap and clamp folded in at build time: row yi + 1 (yi = -1 .. height - 1), column x0
#ifndef ILM_CELLS0_UNTYPED
        // ONE typed load (16_16_16_16 unorm): the four channels arrive decoded by the texture path, as the 16_16 taps do (sdf_unorm_rsrc)
        const f32x4 t = ilm_llvm_buffer_load_format_xyzw(sdf_cells0_rsrc(sdf), (int)(__umul24((uint32_t)(yi + 1), pitch) + ((uint32_t)x0 << 3)), 0, 0);
        a00 = t.x; a10 = t.y; a01 = t.z; a11 = t.w;
#else
        // (A/B: an untyped 8-byte load and the exact decode on the vector ALU -- unorm16_to_float is the texture path's conversion bit for
        // bit; 2-3 % slower on both sizes of tools/collision_probe.py)
        typedef const char __attribute__((address_space(1))) gbyte0;
        typedef uint32_t u32x2c __attribute__((ext_vector_type(2)));
        typedef const u32x2c __attribute__((address_space(1), aligned(8))) gcell0;
        gbyte0* cbase = (gbyte0*)uniform_u64((uint64_t)sdf.cells0);
        const u32x2c t = *(gcell0*)(cbase + (__umul24((uint32_t)(yi + 1), pitch) + ((uint32_t)x0 << 3)));
        a00 = unorm16_to_float((float)(t.x & 0xFFFFu)); a10 = unorm16_to_float((float)(t.x >> 16));
        a01 = unorm16_to_float((float)(t.y & 0xFFFFu)); a11 = unorm16_to_float((float)(t.y >> 16));
#endif
        b00 = b10 = b01 = b11 = 0.0f;
    } else if (SLICE0 && FORMAT == ILM_SDF_UNORM16) {
        const __amdgpu_buffer_rsrc_t rsrc = sdf_unorm_rsrc(sdf);       // channel r of the four taps (the pair's second channel is not needed)
        a00 = ilm_llvm_buffer_load_format_xy(rsrc, (int)(r0 + c0), 0, 0).x; a10 = ilm_llvm_buffer_load_format_xy(rsrc, (int)(r0 + c1), 0, 0).x;
        a01 = ilm_llvm_buffer_load_format_xy(rsrc, (int)(r1 + c0), 0, 0).x; a11 = ilm_llvm_buffer_load_format_xy(rsrc, (int)(r1 + c1), 0, 0).x;
        b00 = b10 = b01 = b11 = 0.0f;
    } else if (FORMAT == ILM_SDF_UNORM16) {
        // typed taps: the texture path decodes the channel pair (see sdf_unorm_rsrc)
        const __amdgpu_buffer_rsrc_t rsrc = sdf_unorm_rsrc(sdf);
        const f32x2 t00 = ilm_llvm_buffer_load_format_xy(rsrc, (int)(r0 + c0), 0, 0), t10 = ilm_llvm_buffer_load_format_xy(rsrc, (int)(r0 + c1), 0, 0);
        const f32x2 t01 = ilm_llvm_buffer_load_format_xy(rsrc, (int)(r1 + c0), 0, 0), t11 = ilm_llvm_buffer_load_format_xy(rsrc, (int)(r1 + c1), 0, 0);
        a00 = t00.x; b00 = t00.y; a10 = t10.x; b10 = t10.y; a01 = t01.x; b01 = t01.y; a11 = t11.x; b11 = t11.y;
    } else {
        const uint32_t w00 = *(gword*)(base + (r0 + c0)), w10 = *(gword*)(base + (r0 + c1));
        const uint32_t w01 = *(gword*)(base + (r1 + c0)), w11 = *(gword*)(base + (r1 + c1));
        sdf_unpack_word<FORMAT>(w00, a00, b00);
        sdf_unpack_word<FORMAT>(w10, a10, b10);
        sdf_unpack_word<FORMAT>(w01, a01, b01);
        sdf_unpack_word<FORMAT>(w11, a11, b11);
    }
    const float lo = lerp_fused(lerp_fused(a00, a10, fx), lerp_fused(a01, a11, fx), fy);
    float blended = lo;
    if (!SLICE0) {
        const float hi = lerp_fused(lerp_fused(b00, b10, fx), lerp_fused(b01, b11, fx), fy);
        blended = lerp_fused(lo, hi, slice_position - vslice);
    }

    return __builtin_fmaf(kDistanceZero - blended, df.Extent.w, distance_to_volume);
}

// ---------------------------------------------------------------------------------------------
// The cone trace's in-volume sampler.  What limits the trace loop is vector-instruction issue (profiles/r02_summary.md), and half
// of the general sampler's instructions only find WHERE the four taps are.  Inside the box of SdfView (every tap of the sample lies in
// the interior of one slice: no clamp, no U WRAP crossing, no V CLAMP, the right-hand tap is the next texel, the lower tap the next
// row) everything that depends on the virtual slice number alone comes from a 16-byte table entry in LDS, built once per workgroup:
//     column index (float), row index (the reference's float form), and the offset of the slice's grid in the CELL array.
// The FLOAT path that decides the taps and the weights is the oracle's, operation for operation (z - zOffset, * sliceCount / extentZ,
// floor, the two fma of u / v, the two fma to texel space, floor, the three fractions), so taps, weights and result are bit-identical
// to sample_distance_field's; the INTEGER path is: two float -> int conversions, one shift-add, one 24-bit multiply-add.
// The taps themselves come from the cell array (SdfView::cells; r03): the atlas keeps the reference's packing, where the eight values of
// a trilinear sample sit in four texels of two rows (two 16-byte loads + a byte permute per tap: the texture path's data return was
// 78 % busy on cfg3, its address path 56 %); the cell array holds them side by side, so an fp16 sample is ONE 16-byte load with the
// f16 pairs already in place, a unorm16 sample two 8-byte typed loads whose channels arrive decoded (see sdf_unorm_rsrc).
// ---------------------------------------------------------------------------------------------
#ifndef ILM_MAX_TABLE_SLICES
#define ILM_MAX_TABLE_SLICES 256
#endif
constexpr int kMaxTableSlices = ILM_MAX_TABLE_SLICES;
struct __attribute__((aligned(16))) SliceEntry {
    float column_index;   // floor(vslice / 3) as float
    float row_index;      // floor(vslice * DistanceFieldPacked1.x): the reference's float form
    uint32_t cell_off;    // byte offset (mod 2^32) that takes (atlas row y0, unfolded atlas column x0) of a tap to the slice's cell:
                          // cell address = y0 * cell_pitch + 16 * x0 + cell_off
    uint32_t pad;
};

// uniform values of the in-volume sampler (SGPRs)
struct InsideConsts {
    float z_offset, slice_scale;                    // DistanceFieldZOffset, sliceCount / extentZ
    float tsx, tsy, tsz, tsw;                       // TextureSliceAndTexelSize
    float wf, hf;                                   // atlas size
    uint32_t pitch;                                 // row pitch of the cell array in bytes (16 * slice width)
    float max_distance;                             // Extent.w
};

ILM_DEV InsideConsts make_inside_consts(const IlmDistanceFieldUniforms& df, const TraceSdfView& sdf) {
    InsideConsts c;
    c.z_offset = df.ConeAndMisc.y; c.slice_scale = df.Packed1.y;
    c.tsx = df.TextureSliceAndTexelSize.x; c.tsy = df.TextureSliceAndTexelSize.y; c.tsz = df.TextureSliceAndTexelSize.z; c.tsw = df.TextureSliceAndTexelSize.w;
    c.wf = sdf.wf; c.hf = sdf.hf;
    c.pitch = (uint32_t)sdf.slice_w << 4;
    c.max_distance = df.Extent.w;
    return c;
}

// entry `vi` of the table (any thread of the workgroup; the caller synchronises)
ILM_DEV SliceEntry make_slice_entry(uint32_t vi, const IlmDistanceFieldUniforms& df, const TraceSdfView& sdf) {
#pragma clang fp contract(off)
    SliceEntry e;
    const uint32_t third = vi / 3u, m = vi - 3u * third;
    e.column_index = (float)third;
    e.row_index = floorf((float)vi * df.Packed1.x);
    // Slice vi's cells form a slice_w x slice_h grid at cell index vi * slice_h * slice_w.  A tap's float path yields the atlas row
    // y0 = row * slice_h + local y (row = physical slice / columns) and the UNFOLDED atlas column x0 = third * slice_w + local x
    // (column_index = third, not third % columns: the U WRAP the table sampler used to fold is simply never applied).
    const uint32_t row = third / (uint32_t)sdf.columns, sw = (uint32_t)sdf.slice_w, sh = (uint32_t)sdf.slice_h;
    e.cell_off = 16u * sw * (sh * (vi - row) - third);
    e.pad = m;
    return e;
}

// the f16 halves of the permuted tap words feed the lerps directly: v_fma_mix_f32 converts its f16 sources exactly and rounds once,
// i.e. it IS cvt + v_sub / v_fma -- without the eight conversions
ILM_DEV float mix_sub_lo(uint32_t b, uint32_t a) { float d; asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(b), "v"(a)); return d; }
ILM_DEV float mix_sub_hi(uint32_t b, uint32_t a) { float d; asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(b), "v"(a)); return d; }
ILM_DEV float mix_fma_lo(float t, float d, uint32_t a) { float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(t), "v"(d), "v"(a)); return r; }
ILM_DEV float mix_fma_hi(float t, float d, uint32_t a) { float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(t), "v"(d), "v"(a)); return r; }

// Precondition (the caller's, per sample): position inside SdfView's box.  `table` = the workgroup's LDS table.
template <int FORMAT>
ILM_DEV float sample_inside_table(f3 position, const InsideConsts& c, const TraceSdfView& sdf, const SliceEntry* table) {
#pragma clang fp contract(off)
    const float pz = position.z - c.z_offset;
    const float slice_position = pz * c.slice_scale;          // min(clamp(z), validZ) is the identity inside the box
    // Inside the box slice_position, x and y are >= 0 (z >= the offset; every tap at least 1/16 texel inside its slice), so
    //   floor(t)     = the truncating float -> int conversion, and
    //   t - floor(t) = v_fract_f32(t): the subtraction is exact for every finite t and below 1 for t >= 0, where the instruction's
    //                  clamp to 1 - 2^-24 never acts
    // -- the oracle's floor / subtract / convert as two instructions per axis instead of three (r03).
    const SliceEntry e = table[(uint32_t)slice_position];
    const float u = __builtin_fmaf(e.column_index, c.tsx, position.x * c.tsz);
    const float v = __builtin_fmaf(e.row_index, c.tsy, position.y * c.tsw);
    const float x = __builtin_fmaf(u, c.wf, -0.5f);
    const float y = __builtin_fmaf(v, c.hf, -0.5f);
    const float fx = __builtin_amdgcn_fractf(x), fy = __builtin_amdgcn_fractf(y), fz = __builtin_amdgcn_fractf(slice_position);
    // the sample's cell: row floor(y), unfolded column floor(x) of the atlas (no clamp, no wrap inside the box), moved to the slice's grid
    const uint32_t column16 = ((uint32_t)(int)x << 4) + e.cell_off;
    uint32_t offset;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(offset) : "v"((uint32_t)(int)y), "s"(c.pitch), "v"(column16));
    float lo0, lo1, hi0, hi1;
    if (FORMAT == ILM_SDF_FP16) {
        // ONE 16-byte load: (w00, w10, w01, w11), each word the f16 pair (slice v, slice v + 1) of a tap
        typedef const char __attribute__((address_space(1))) gbyte;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef const u32x4 __attribute__((address_space(1), aligned(16))) gcell;
        gbyte* base = (gbyte*)uniform_u64((uint64_t)sdf.cells);
        const u32x4 t = *(gcell*)(base + offset);
        lo0 = mix_fma_lo(fx, mix_sub_lo(t.y, t.x), t.x); hi0 = mix_fma_hi(fx, mix_sub_hi(t.y, t.x), t.x);
        lo1 = mix_fma_lo(fx, mix_sub_lo(t.w, t.z), t.z); hi1 = mix_fma_hi(fx, mix_sub_hi(t.w, t.z), t.z);
    } else {
        // two typed loads (16_16_16_16 unorm): the upper tap row (a00, b00, a10, b10) and the lower one, decoded by the texture path
        const __amdgpu_buffer_rsrc_t rsrc = sdf_cells_unorm_rsrc(sdf);
        const f32x4 t0 = ilm_llvm_buffer_load_format_xyzw(rsrc, (int)offset, 0, 0), t1 = ilm_llvm_buffer_load_format_xyzw(rsrc, (int)offset + 8, 0, 0);
        lo0 = lerp_fused(t0.x, t0.z, fx); hi0 = lerp_fused(t0.y, t0.w, fx);
        lo1 = lerp_fused(t1.x, t1.z, fx); hi1 = lerp_fused(t1.y, t1.w, fx);
    }
    const float lo = lerp_fused(lo0, lo1, fy), hi = lerp_fused(hi0, hi1, fy);
    const float blended = lerp_fused(lo, hi, fz);
    return (kDistanceZero - blended) * c.max_distance;      // fma(x, maxDistance, +0): the distance to the volume is +0 inside it
}

}  // namespace ilm
