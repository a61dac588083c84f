// output.hip -- the callers on the output side of the two hot paths, for gfx950:
//   * particle read-back: FillReadbackResult (Illuminant/Particles/ParticleReadback.cs:73-167) as an ordered device-side
//     compaction of the live particles into draw-call records (the reference copies three float4 planes per chunk to the host
//     and filters there: 48 B/slot over PCIe; here 48 B per LIVE particle);
//   * lightmap resolve: the LightingResolve[WithAlbedo] techniques of Illuminant/Shaders/Resolve.fx:25-233 + HDR.fxh, a pure stream
//     (8 B read + 4..16 B written per pixel): HBM-bound, no LDS, no MFMA.
#include "internal.hpp"

namespace ilm {

constexpr int kRbBlock = 1024;

ILM_DEV bool readback_live(const ReadbackLaunch& a, int chunk, int slot, float& life) {
    const int count = (a.element_counts != nullptr) ? a.element_counts[chunk] : a.slots;
    if (slot >= count || slot >= a.slots)
        return false;
    life = a.chunk_bases[chunk][3 * a.stride + slot];
    return life > 0.0f;
}

__global__ __launch_bounds__(kRbBlock) void readback_count_kernel(const ReadbackLaunch a, int blocks_per_chunk) {
    __shared__ int wave_counts[kRbBlock / 64];
    const int chunk = (int)blockIdx.x / blocks_per_chunk, blk = (int)blockIdx.x - chunk * blocks_per_chunk;
    float life;
    const bool live = readback_live(a, chunk, blk * kRbBlock + (int)threadIdx.x, life);
    const unsigned long long m = __ballot(live);
    if ((threadIdx.x & 63u) == 0u) wave_counts[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = 0;
        for (int w = 0; w < kRbBlock / 64; w++) n += wave_counts[w];
        a.block_counts[blockIdx.x] = n;
    }
}

__global__ __launch_bounds__(kRbBlock) void readback_emit_kernel(const ReadbackLaunch a, int blocks_per_chunk) {
    __shared__ int wave_counts[kRbBlock / 64];
    __shared__ int partial[kRbBlock / 64];
    __shared__ int block_base;
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    int sum = 0;
    for (int i = (int)threadIdx.x; i < (int)blockIdx.x; i += kRbBlock) sum += a.block_counts[i];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if (lane == 0) partial[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int b = 0;
        for (int w = 0; w < kRbBlock / 64; w++) b += partial[w];
        block_base = b;
        if (blockIdx.x == gridDim.x - 1)
            *a.out_count = b + a.block_counts[blockIdx.x];      // the TOTAL, even beyond the capacity
    }
    const int chunk = (int)blockIdx.x / blocks_per_chunk, blk = (int)blockIdx.x - chunk * blocks_per_chunk;
    const int slot = blk * kRbBlock + (int)threadIdx.x;
    float life;
    const bool live = readback_live(a, chunk, slot, life);
    const unsigned long long m = __ballot(live);
    if (lane == 0) wave_counts[wave] = __popcll(m);
    __syncthreads();
    if (!live)
        return;
    int index = block_base + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) index += wave_counts[w];
    if (index >= a.capacity)
        return;

    // FillReadbackResult's loop body, ParticleReadback.cs:118-163
    const IlmReadbackParams& p = a.params;
    const float* base = a.chunk_bases[chunk];
    const int64_t S = a.stride;
    const float px = base[slot], py = base[S + slot];
    const float4 rc = mk4(base[12 * S + slot], base[13 * S + slot], base[14 * S + slot], base[15 * S + slot]);
    const float4 rd = mk4(base[16 * S + slot], base[17 * S + slot], base[18 * S + slot], base[19 * S + slot]);
    const float sz = rd.x;
    const float rot = fmodf(rd.y, 6.28318530717958647692f);          // (float)(2 * Math.PI)
    IlmReadbackDrawCall dc;
    dc.TextureRegion[0] = p.TextureRegion[0]; dc.TextureRegion[1] = p.TextureRegion[1];
    dc.TextureRegion[2] = p.TextureRegion[2]; dc.TextureRegion[3] = p.TextureRegion[3];
    if ((a.frame_count_x > 1) || (a.frame_count_y > 1)) {
        float fx = floorf(fabsf(p.AnimationRate[0]) * life), fy = floorf(fabsf(p.AnimationRate[1]) * life);
        fy += (float)floor((double)rd.w);
        if (p.ColumnFromVelocity) fx += (float)rint((double)rot / a.max_angle_x);        // Math.Round: half to even, in double
        if (p.RowFromVelocity)    fy += (float)rint((double)rot / a.max_angle_y);
        fx = fmodf(fmaxf(0.0f, fx), (float)a.frame_count_x);
        fy = clampf(fy, 0.0f, (float)(a.frame_count_y - 1));
        if (p.AnimationRate[0] < 0.0f) fx = (float)a.frame_count_x - fx;
        if (p.AnimationRate[1] < 0.0f) fy = (float)a.frame_count_y - fy;
        const float ox = fx * a.region_w, oy = fy * a.region_h;
        dc.TextureRegion[0] += ox; dc.TextureRegion[1] += oy; dc.TextureRegion[2] += ox; dc.TextureRegion[3] += oy;
    }
    dc.Position[0] = px; dc.Position[1] = py;
    dc.SortOrder = p.SortedReadback ? (py + p.ZToY) : 0.0f;
    dc.Scale[0] = p.Size[0] * sz; dc.Scale[1] = p.Size[1] * sz;
    dc.MultiplyColor[0] = (uint8_t)(int32_t)(rc.x * 255.0f);
    dc.MultiplyColor[1] = (uint8_t)(int32_t)(rc.y * 255.0f);
    dc.MultiplyColor[2] = (uint8_t)(int32_t)(rc.z * 255.0f);
    dc.MultiplyColor[3] = (uint8_t)(int32_t)(rc.w * 255.0f);
    dc.Rotation = (float)((p.RotationFromVelocity ? 1.0 : 0.0) * (double)rot);
    dc._pad = 0;
    a.out[index] = dc;
}

hipError_t launch_readback(const ReadbackLaunch& a, hipStream_t stream) {
    const int blocks_per_chunk = (a.slots + kRbBlock - 1) / kRbBlock;
    const int blocks = a.chunk_count * blocks_per_chunk;
    if (blocks <= 0) return hipMemsetAsync(a.out_count, 0, sizeof(int32_t), stream);
    hipLaunchKernelGGL(readback_count_kernel, dim3(blocks), dim3(kRbBlock), 0, stream, a, blocks_per_chunk);
    hipLaunchKernelGGL(readback_emit_kernel, dim3(blocks), dim3(kRbBlock), 0, stream, a, blocks_per_chunk);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// lightmap resolve
// ---------------------------------------------------------------------------------------------
ILM_DEV float4 load_lightmap_texel(const void* texels, int format, size_t o) {
    if (format == ILM_LIGHTMAP_FLOAT4)
        return reinterpret_cast<const float4*>(texels)[o];
    if (format == ILM_LIGHTMAP_HALF4) {
        const uint2 v = reinterpret_cast<const uint2*>(texels)[o];
        return mk4(__half2float(__ushort_as_half((unsigned short)(v.x & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(v.x >> 16))),
                   __half2float(__ushort_as_half((unsigned short)(v.y & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(v.y >> 16))));
    }
    const uint32_t v = reinterpret_cast<const uint32_t*>(texels)[o];
    return mk4((float)(v & 0xFFu) / 255.0f, (float)((v >> 8) & 0xFFu) / 255.0f, (float)((v >> 16) & 0xFFu) / 255.0f, (float)(v >> 24) / 255.0f);
}

ILM_DEV void store_lightmap_texel(void* texels, int format, size_t o, float4 c) {
    if (format == ILM_LIGHTMAP_FLOAT4) {
        reinterpret_cast<float4*>(texels)[o] = c;
    } else if (format == ILM_LIGHTMAP_HALF4) {
        uint2 v;
        v.x = (uint32_t)__half_as_ushort(__float2half_rn(c.x)) | ((uint32_t)__half_as_ushort(__float2half_rn(c.y)) << 16);
        v.y = (uint32_t)__half_as_ushort(__float2half_rn(c.z)) | ((uint32_t)__half_as_ushort(__float2half_rn(c.w)) << 16);
        reinterpret_cast<uint2*>(texels)[o] = v;
    } else {
        const uint32_t r = (uint32_t)rintf(sat(c.x) * 255.0f), g = (uint32_t)rintf(sat(c.y) * 255.0f);
        const uint32_t b = (uint32_t)rintf(sat(c.z) * 255.0f), al = (uint32_t)rintf(sat(c.w) * 255.0f);
        reinterpret_cast<uint32_t*>(texels)[o] = r | (g << 8) | (b << 16) | (al << 24);
    }
}

// pow_pos: hlsl_math.hpp

ILM_DEV float4 resolve_texel(const ResolveLaunch& a, float4 color, float4 albedo) {
    float r, g, b, alpha;
    if (a.albedo != nullptr) {
        // ResolveWithAlbedoCommon, Resolve.fx:43-60 (AlbedoIsSRGB = 0): light *= InverseScaleFactor * 2, then
        // lerp(albedo.rgb, albedo.rgb * light.rgb, saturate(light.a)); the alpha is the albedo's
        const float k = a.inverse_scale * 2.0f;
        const float t = sat(color.w * k);
        r = lerp(albedo.x, albedo.x * (color.x * k), t); g = lerp(albedo.y, albedo.y * (color.y * k), t); b = lerp(albedo.z, albedo.z * (color.z * k), t);
        alpha = albedo.w;
    } else {
        // ResolveCommon, Resolve.fx:25-40 (scale 1: the pixel's own texel)
        r = color.x * a.inverse_scale; g = color.y * a.inverse_scale; b = color.z * a.inverse_scale;
        alpha = 1.0f;
    }
    if (a.mode == ILM_HDR_GAMMA_COMPRESS) {
        // GammaCompress, HDR.fxh:11-18
        r = fmaxf(r + a.offset, 0.0f); g = fmaxf(g + a.offset, 0.0f); b = fmaxf(b + a.offset, 0.0f);
        const float result_luminance = r * 0.299f + g * 0.587f + b * 0.114f;
        const float scaled = (result_luminance * a.middle_gray) * a.inv_average_luminance;
        const float compressed = (scaled * (1.0f + (scaled * a.inv_maximum_luminance_squared))) * fast_rcp(1.0f + scaled);
        const float rescale = compressed * fast_rcp(result_luminance);
        // 0 / 0 at a black pixel is NaN in the shader as well (compressedLuminance / resultLuminance)
        r *= rescale; g *= rescale; b *= rescale;
    } else if (a.mode == ILM_HDR_TONE_MAP) {
        // ToneMappedLightingResolve[WithAlbedo]PixelShader, Resolve.fx:113-139,209-233; Uncharted2Tonemap, HDR.fxh:38-44
        const float kA = 0.15f, kB = 0.50f, kC = 0.10f, kD = 0.20f, kE = 0.02f, kF = 0.30f;
        const float e = a.exposure_minus_one + 1.0f, gm = a.gamma_minus_one + 1.0f;
        float v[3] = { fmaxf(0.0f, r + a.offset) * e, fmaxf(0.0f, g + a.offset) * e, fmaxf(0.0f, b + a.offset) * e };
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float t = ((v[k] * (kA * v[k] + kC * kB) + kD * kE) * fast_rcp(v[k] * (kA * v[k] + kB) + kD * kF)) - kE / kF;
            v[k] = pow_pos(t * a.inv_white, gm);
        }
        r = v[0]; g = v[1]; b = v[2];
    } else {
        // LightingResolve[WithAlbedo]PixelShader, Resolve.fx:62-83,141-158
        const float e = a.exposure_minus_one + 1.0f, gm = a.gamma_minus_one + 1.0f;
        r = pow_pos(fmaxf(0.0f, r + a.offset) * e, gm);
        g = pow_pos(fmaxf(0.0f, g + a.offset) * e, gm);
        b = pow_pos(fmaxf(0.0f, b + a.offset) * e, gm);
    }
    return mk4(r, g, b, alpha);
}

// Pure stream: 2 pixels per lane (16 B half4 loads, 8 B RGBA8 stores; a wave moves 1 KiB / 512 B per instruction).
__global__ __launch_bounds__(256) void resolve_kernel(const ResolveLaunch a) {
    const size_t pair = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)(a.row_end - a.row_begin) * (size_t)a.width;
    const size_t i = pair * 2;
    if (i >= n) return;
    const size_t o = (size_t)a.row_begin * (size_t)a.width + i;
    const bool two = (i + 1 < n) && ((o & 1) == 0);
    if (two && a.src_format == ILM_LIGHTMAP_HALF4 && a.dst_format == ILM_LIGHTMAP_RGBA8 && (a.albedo == nullptr || a.albedo_format == ILM_LIGHTMAP_RGBA8)) {
        // the back-buffer case: one 16-byte load, one 8-byte store
        const uint4 v = reinterpret_cast<const uint4*>(a.src)[o >> 1];
        const float4 c0 = mk4(__half2float(__ushort_as_half((unsigned short)(v.x & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(v.x >> 16))),
                              __half2float(__ushort_as_half((unsigned short)(v.y & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(v.y >> 16))));
        const float4 c1 = mk4(__half2float(__ushort_as_half((unsigned short)(v.z & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(v.z >> 16))),
                              __half2float(__ushort_as_half((unsigned short)(v.w & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(v.w >> 16))));
        float4 a0 = mk4(0.0f, 0.0f, 0.0f, 0.0f), a1 = a0;
        if (a.albedo != nullptr) {       // a Color texture: one 8-byte load for the two texels
            const uint2 av = reinterpret_cast<const uint2*>(a.albedo)[o >> 1];
            a0 = mk4((float)(av.x & 0xFFu) / 255.0f, (float)((av.x >> 8) & 0xFFu) / 255.0f, (float)((av.x >> 16) & 0xFFu) / 255.0f, (float)(av.x >> 24) / 255.0f);
            a1 = mk4((float)(av.y & 0xFFu) / 255.0f, (float)((av.y >> 8) & 0xFFu) / 255.0f, (float)((av.y >> 16) & 0xFFu) / 255.0f, (float)(av.y >> 24) / 255.0f);
        }
        const float4 r0 = resolve_texel(a, c0, a0), r1 = resolve_texel(a, c1, a1);
        uint2 out;
        out.x = (uint32_t)rintf(sat(r0.x) * 255.0f) | ((uint32_t)rintf(sat(r0.y) * 255.0f) << 8) | ((uint32_t)rintf(sat(r0.z) * 255.0f) << 16) | ((uint32_t)rintf(sat(r0.w) * 255.0f) << 24);
        out.y = (uint32_t)rintf(sat(r1.x) * 255.0f) | ((uint32_t)rintf(sat(r1.y) * 255.0f) << 8) | ((uint32_t)rintf(sat(r1.z) * 255.0f) << 16) | ((uint32_t)rintf(sat(r1.w) * 255.0f) << 24);
        reinterpret_cast<uint2*>(a.dst)[o >> 1] = out;
        return;
    }
    const float4 none = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    store_lightmap_texel(a.dst, a.dst_format, o, resolve_texel(a, load_lightmap_texel(a.src, a.src_format, o),
                                                                a.albedo ? load_lightmap_texel(a.albedo, a.albedo_format, o) : none));
    if (i + 1 < n)
        store_lightmap_texel(a.dst, a.dst_format, o + 1, resolve_texel(a, load_lightmap_texel(a.src, a.src_format, o + 1),
                                                                        a.albedo ? load_lightmap_texel(a.albedo, a.albedo_format, o + 1) : none));
}

hipError_t launch_resolve(const ResolveLaunch& a, hipStream_t stream) {
    const size_t n = (size_t)(a.row_end - a.row_begin) * (size_t)a.width;
    if (n == 0) return hipSuccess;
    const size_t pairs = (n + 1) / 2;
    hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// Sixteen bytes written by the DEVICE at `where` (group.hip, prove_ipc_mappings: the proof goes through the same kind of access the
// store mode's mirror stores use -- a kernel's store through the mapping -- not through a copy engine).
__global__ void stamp16_kernel(uint4* where, uint4 stamp) { *where = stamp; }

hipError_t launch_stamp16(void* where, const uint32_t stamp[4], hipStream_t stream) {
    hipLaunchKernelGGL(stamp16_kernel, dim3(1), dim3(1), 0, stream, static_cast<uint4*>(where), make_uint4(stamp[0], stamp[1], stamp[2], stamp[3]));
    return hipGetLastError();
}

}  // namespace ilm
