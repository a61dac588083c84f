// raster.hip -- particle rasterisation for gfx950: technique RasterizeParticlesNoTexture
// (Illuminant/Shaders/RasterizeParticleSystem.fx:61-163,228-241; ParticleSystem.Render / RenderChunk,
// Illuminant/Particles/ParticleSystem.cs:876-1041).
//
// The reference draws one instanced quad per slot, chunk after chunk, and lets the ROP blend them in that order.  Blending is
// order-dependent, so the order is kept: every live particle becomes a sprite record (the inverse of its affine map unit square ->
// pixels), each (16 x 16 pixel tile, sprite) pair a 64-bit key (tile << 32 | global slot), the keys are radix-sorted by tile (rocPRIM, a
// plain library sort; stable, and the keys are emitted in slot order) and one workgroup per tile walks its run of keys in slot order: 256 sprites at a time through LDS, every
// lane = one pixel, coverage + shading + blending in registers, the target texel read once and written once.
//
// Compiled with -ffp-contract=off: coverage is decided by the same IEEE operations as the CPU oracle.
#include <cstring>

#include "internal.hpp"
#include "bezier.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace ilm {

constexpr int kRasterTile = 16;

struct alignas(16) Sprite {
    float cx, cy;               // centre, pixels
    float ex, ey;               // half extents of the bounding box, pixels (+ 1 pixel of slack): culling only
    float i00, i01, i10, i11;   // unit = I * (pixel centre - centre)
    float r, g, b, a;           // RenderColor (x GlobalColor for NoTexture; the textured pixel shaders apply it after the texel)
    float rounding;
    float frame_u, frame_v;     // frameTexCoord: offset of the animation frame inside the sheet
    float dither_frame;         // floor(index % 4) of premultipliedToDithered; index = the slot (ParticleEngine.cs:476-478)
};
static_assert(sizeof(Sprite) == 64, "Sprite is 16 words");

// VS_PosVelAttr, RasterizeParticleSystem.fx:61-148, for one slot: the sprite record and the number of tiles its clipped bounding
// box touches (0: dead, degenerate or off-screen).
__global__ __launch_bounds__(256) void raster_setup_kernel(const RasterLaunch a) {
    const int g = (int)blockIdx.x * 256 + (int)threadIdx.x;
    bool live = false;
    uint32_t count = 0;
    if (g < a.total_slots) {
        const int chunk = g / a.slots, slot = g - chunk * a.slots;
        const int quads = a.quad_counts ? a.quad_counts[chunk] : a.slots;
        const IlmRasterizeParams& p = a.params;
        if (slot < quads) {
            const float* base = a.chunk_bases[chunk];
            const int64_t S = a.stride;
            const float life = base[3 * S + slot];
            if (!(life <= 0.0f)) {       // `life <= 0` rejects; a NaN life does not, as in the shader
                const float px = base[slot], py = base[S + slot], pz = base[2 * S + slot];
                const float rd_x = base[16 * S + slot], rd_y = base[17 * S + slot];
                const float angle = fmodf(rd_y, 2.0f * kPi);
                float sx = rd_x * p.SystemSize[0] * p.SizeFactorAndPosition.x;
                float sy = rd_x * p.SystemSize[1] * p.SizeFactorAndPosition.y;
                const float zf = fmaxf(0.0f, 1.0f + (pz * p.ZConfiguration.x));
                sx *= zf; sy *= zf;
                const float s = sinf(angle), c = cosf(angle);
                const float display_x = (px * p.Scale.x) + p.SizeFactorAndPosition.z;
                const float display_y = ((py - (pz * p.ZToY)) * p.Scale.y) + p.SizeFactorAndPosition.w;
                Sprite sp;
                sp.cx = (display_x - p.ViewportPosition[0]) * p.ViewportScale[0];
                sp.cy = (display_y - p.ViewportPosition[1]) * p.ViewportScale[1];
                const float kx = p.Scale.x * p.ViewportScale[0], ky = p.Scale.y * p.ViewportScale[1];
                const float a00 = (c * sx) * kx, a01 = -(s * sy) * kx;
                const float a10 = (s * sx) * ky, a11 = (c * sy) * ky;
                const float det = (a00 * a11) - (a01 * a10);
                if ((fabsf(det) > 0.0f) && isfinite(det) && isfinite(sp.cx) && isfinite(sp.cy)) {
                    live = true;
                    sp.i00 = a11 / det;  sp.i01 = -a01 / det;
                    sp.i10 = -a10 / det; sp.i11 = a00 / det;
                    const float ex = fabsf(a00) + fabsf(a01), ey = fabsf(a10) + fabsf(a11);
                    const bool textured = p.BitmapFilter != ILM_BITMAP_NONE;
                    sp.r = base[12 * S + slot]; sp.g = base[13 * S + slot]; sp.b = base[14 * S + slot]; sp.a = base[15 * S + slot];
                    if (!textured) { sp.r *= p.GlobalColor.x; sp.g *= p.GlobalColor.y; sp.b *= p.GlobalColor.z; sp.a *= p.GlobalColor.w; }
                    sp.rounding = clampf(bezier1(p.RoundingPowerFromLife, life), 0.001f, 1.0f);
                    sp.frame_u = sp.frame_v = 0.0f;
                    if (textured) {
                        // frame selection, RasterizeParticleSystem.fx:112-139
                        const float tex_w = p.BitmapTextureRegion.z - p.BitmapTextureRegion.x, tex_h = p.BitmapTextureRegion.w - p.BitmapTextureRegion.y;
                        const float count_x = floorf(1.0f / tex_w), count_y = floorf(1.0f / tex_h);
                        float fx = floorf(fabsf(p.AnimationRate[0]) * life), fy = floorf(fabsf(p.AnimationRate[1]) * life);
                        const float max_angle_x = (2.0f * kPi) / count_x, max_angle_y = (2.0f * kPi) / count_y;
                        fy += floorf(base[19 * S + slot]);
                        if (p.RenderingOptions[2] != 0.0f) fx += rintf(angle / max_angle_x);      // HLSL round(): ties to even
                        if (p.RenderingOptions[3] != 0.0f) fy += rintf(angle / max_angle_y);
                        fx = fmodf(fmaxf(fx, 0.0f), count_x);
                        fy = clampf(fy, 0.0f, count_y - 1.0f);
                        if (p.AnimationRate[0] < 0.0f) fx = (count_x - fx) - 1.0f;
                        if (p.AnimationRate[1] < 0.0f) fy = (count_y - fy) - 1.0f;
                        sp.frame_u = fx * tex_w; sp.frame_v = fy * tex_h;
                    }
                    // pixel centres within the bounding box, one pixel of slack (the unit-square test decides), clipped to the target
                    float fx0 = floorf(sp.cx - ex - 0.5f) - 1.0f, fx1 = ceilf(sp.cx + ex - 0.5f) + 1.0f;
                    float fy0 = floorf(sp.cy - ey - 0.5f) - 1.0f, fy1 = ceilf(sp.cy + ey - 0.5f) + 1.0f;
                    fx0 = fmaxf(fx0, 0.0f); fy0 = fmaxf(fy0, 0.0f);
                    fx1 = fminf(fx1, (float)(a.width - 1)); fy1 = fminf(fy1, (float)(a.height - 1));
                    sp.dither_frame = (float)(slot & 3);
                    sp.ex = ex + 1.0f; sp.ey = ey + 1.0f;
                    uint2 rect = make_uint2(0u, 0u);
                    if ((fx0 <= fx1) && (fy0 <= fy1)) {
                        const uint32_t tx0 = (uint32_t)fx0 / kRasterTile, tx1 = (uint32_t)fx1 / kRasterTile;
                        const uint32_t ty0 = (uint32_t)fy0 / kRasterTile, ty1 = (uint32_t)fy1 / kRasterTile;
                        rect = make_uint2(tx0 | (tx1 << 16), ty0 | (ty1 << 16));
                        count = (tx1 - tx0 + 1u) * (ty1 - ty0 + 1u);
                    }
                    a.sprites[g] = sp;
                    a.rects[g] = rect;
                }
            }
        }
        a.counts[g] = count;
    }
    if (a.count_shaded) {       // statistics only: 20 k same-address atomics serialise (~11 ns each)
        const unsigned long long m = __ballot(live);
        if (((threadIdx.x & 63u) == 0u) && m != 0ull)
            atomicAdd(&a.stats[0], (unsigned long long)__popcll(m));
    }
}

// one key per (tile, sprite): tile << 32 | global slot -- sorting them lists every tile's sprites in chunk / slot order
__global__ __launch_bounds__(256) void raster_emit_kernel(const RasterLaunch a) {
    const int g = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (g >= a.total_slots) return;
    const uint32_t count = a.counts[g];
    if (count == 0u) return;
    const uint2 rect = a.rects[g];      // first | last << 16 tile column / row of the clipped bounding box
    const uint32_t tx0 = rect.x & 0xFFFFu, tx1 = rect.x >> 16, ty0 = rect.y & 0xFFFFu, ty1 = rect.y >> 16;
    unsigned long long* out = a.keys + a.offsets[g];
    for (uint32_t ty = ty0; ty <= ty1; ty++)
        for (uint32_t tx = tx0; tx <= tx1; tx++)
            *out++ = ((unsigned long long)(ty * (uint32_t)a.tiles_x + tx) << 32) | (unsigned long long)(uint32_t)g;
}

template <int FORMAT>
ILM_DEV float4 load_target(const void* texels, size_t o) {
    if (FORMAT == ILM_LIGHTMAP_FLOAT4) return reinterpret_cast<const float4*>(texels)[o];
    if (FORMAT == ILM_LIGHTMAP_HALF4) {
        const uint2 v = reinterpret_cast<const uint2*>(texels)[o];
        return mk4(__half2float(__ushort_as_half((unsigned short)(v.x & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(v.x >> 16))),
                   __half2float(__ushort_as_half((unsigned short)(v.y & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(v.y >> 16))));
    }
    const uint32_t v = reinterpret_cast<const uint32_t*>(texels)[o];
    return mk4((float)(v & 255u) / 255.0f, (float)((v >> 8) & 255u) / 255.0f, (float)((v >> 16) & 255u) / 255.0f, (float)(v >> 24) / 255.0f);
}
template <int FORMAT>
ILM_DEV void store_target(void* texels, size_t o, float4 c) {
    if (FORMAT == ILM_LIGHTMAP_FLOAT4) {
        reinterpret_cast<float4*>(texels)[o] = c;
    } else if (FORMAT == ILM_LIGHTMAP_HALF4) {
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

// tex2D on a bitmap without mips: BitmapPointSampler (POINT, CLAMP) or BitmapSampler (LINEAR, CLAMP), texel centres at + 0.5
ILM_DEV float4 bitmap_fetch(const float4* __restrict__ tex, int w, int h, float u, float v, int filter) {
    if (filter == ILM_BITMAP_POINT) {
        float xf = floorf(u * (float)w), yf = floorf(v * (float)h);
        xf = (xf >= 0.0f) ? xf : 0.0f; xf = fminf(xf, (float)(w - 1));
        yf = (yf >= 0.0f) ? yf : 0.0f; yf = fminf(yf, (float)(h - 1));
        return tex[(int)yf * w + (int)xf];
    }
    const float sx = u * (float)w - 0.5f, sy = v * (float)h - 0.5f;
    float x0f = floorf(sx), y0f = floorf(sy);
    const float fx = sx - x0f, fy = sy - y0f;
    float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    x0f = (x0f >= 0.0f) ? x0f : 0.0f; x0f = fminf(x0f, (float)(w - 1));
    x1f = (x1f >= 0.0f) ? x1f : 0.0f; x1f = fminf(x1f, (float)(w - 1));
    y0f = (y0f >= 0.0f) ? y0f : 0.0f; y0f = fminf(y0f, (float)(h - 1));
    y1f = (y1f >= 0.0f) ? y1f : 0.0f; y1f = fminf(y1f, (float)(h - 1));
    const int x0 = (int)x0f, x1 = (int)x1f, y0 = (int)y0f, y1 = (int)y1f;
    return lerp4(lerp4(tex[y0 * w + x0], tex[y0 * w + x1], fx), lerp4(tex[y1 * w + x0], tex[y1 * w + x1], fx), fy);
}

// first index whose key is >= value
ILM_DEV int64_t key_lower_bound(const unsigned long long* keys, int64_t n, unsigned long long value) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < value) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// A tile's run of keys is cut into segments of at most kRasterSegment sprites, one work item (workgroup) each: particle systems
// cluster (attractors), and one workgroup walking the 50 000 sprites of the densest tile alone made the frame 5x longer than the
// same number of sprites spread out.  `over` is associative -- a run of fragments acts on what lies beneath it as
// dst' = C + T * dst with C the blended colour of the run over black and T the product of its (1 - alpha) -- so the segments of a
// tile are shaded independently into (C, T) per pixel and combined in order afterwards (float rounding moves by an ulp or two
// with the bracketing; blending is otherwise the same chain of operations).  Tiles with a single segment blend straight into the
// target.
constexpr int kRasterSegment = 2048;

__global__ __launch_bounds__(256) void raster_tile_ranges_kernel(const RasterLaunch a) {
    const int tile = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int tiles = a.tiles_x * a.tiles_y;
    if (tile > tiles) return;
    if (tile == tiles) { a.tile_begin[tile] = (uint32_t)a.pair_count; a.tile_segments[tile] = 0u; a.tile_multi[tile] = 0u; return; }
    const int64_t begin = key_lower_bound(a.sorted_keys, a.pair_count, (unsigned long long)(uint32_t)tile << 32);
    const int64_t end = key_lower_bound(a.sorted_keys, a.pair_count, (unsigned long long)(uint32_t)(tile + 1) << 32);
    const uint32_t segments = (uint32_t)((end - begin + kRasterSegment - 1) / kRasterSegment);
    a.tile_begin[tile] = (uint32_t)begin;
    a.tile_segments[tile] = segments;
    a.tile_multi[tile] = (segments > 1u) ? segments : 0u;
}

// Exclusive prefix sums of tile_segments (-> the tile's first work item) and tile_multi (-> its first partial slot) over the tiles + 1
// entries, both in ONE workgroup: a thread sums a run of consecutive tiles, the runs' sums are scanned across the workgroup (shuffles
// within a wave, the sixteen waves' totals through LDS), the thread writes its run's prefixes.  (r03: two library scans = four launches
// of ~5 us each for 8 161 numbers.)
__global__ __launch_bounds__(1024) void raster_tile_scan_kernel(const RasterLaunch a) {
    __shared__ uint32_t wave_sum[2][16];
    const int count = a.tiles_x * a.tiles_y + 1;
    const int per = (count + 1023) / 1024;
    const int first = min((int)threadIdx.x * per, count), last = min(first + per, count);
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    // (eight tiles at a time with all sixteen loads in flight: one thread's run is a chain of dependent cache misses otherwise)
    uint32_t s0 = 0u, s1 = 0u;
    for (int base = first; base < last; base += 8) {
        uint32_t x[8], y[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const int i = min(base + k, count - 1); x[k] = a.tile_segments[i]; y[k] = a.tile_multi[i]; }
#pragma unroll
        for (int k = 0; k < 8; k++) { if (base + k < last) { s0 += x[k]; s1 += y[k]; } }
    }
    uint32_t i0 = s0, i1 = s1;                                   // inclusive scan of the runs' sums within the wave
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t0 = __shfl_up(i0, off), t1 = __shfl_up(i1, off);
        if (lane >= off) { i0 += t0; i1 += t1; }
    }
    if (lane == 63) { wave_sum[0][wave] = i0; wave_sum[1][wave] = i1; }
    __syncthreads();
    uint32_t e0 = i0 - s0, e1 = i1 - s1;
    for (int w = 0; w < wave; w++) { e0 += wave_sum[0][w]; e1 += wave_sum[1][w]; }
    for (int base = first; base < last; base += 8) {
        uint32_t x[8], y[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const int i = min(base + k, count - 1); x[k] = a.tile_segments[i]; y[k] = a.tile_multi[i]; }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (base + k < last) {
                a.tile_first_item[base + k] = e0; e0 += x[k];
                a.tile_first_partial[base + k] = e1; e1 += y[k];
            }
        }
    }
}

// PS_NoTexture + the blend, RasterizeParticleSystem.fx:150-163,228-241: one workgroup per work item (a tile, or a segment of a
// crowded tile's run), one lane per pixel.
// The sprites come 256 at a time: thread t fetches sprite t of the batch into LDS and tests its bounding box against the four
// 8 x 8 quadrants of the tile; a ballot + prefix count per quadrant turns that into four slot-ordered index lists, and wave q then
// walks only the sprites that can touch its quadrant (for 8 x 8-pixel sprites about a third of the tile's), the next record
// already in flight while the current one is shaded.
#ifdef ILM_RASTER_TRACE    // EXPERIMENT (tools/raster_trace_probe.py): per-workgroup start / end of the last launch (100 MHz clock), sprite count, segment
__device__ unsigned long long g_raster_trace[4 * 131072];
extern "C" int ilm_experiment_raster_trace(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_raster_trace), sizeof(unsigned long long) * (size_t)n);
}
#endif
// Eight waves per SIMD (64 VGPRs, no scratch once only the next sprite's geometry is prefetched).  Per-workgroup timestamps
// (-DILM_RASTER_TRACE, tools/raster_trace_probe.py) showed 4.7 workgroups per CU in flight at 78 VGPRs and the pass waiting on LDS /
// barriers rather than issuing: cfg2's frame 1.155 -> 1.062 ms (tools/ab_raster.sh).  Sprite records through scalar loads straight
// from global memory instead of the LDS batch: 1.30 ms.
#ifndef ILM_RASTER_WAVES
#define ILM_RASTER_WAVES 8
#endif
#if ILM_RASTER_WAVES > 0
#define ILM_RASTER_OCCUPANCY __attribute__((amdgpu_waves_per_eu(ILM_RASTER_WAVES, ILM_RASTER_WAVES)))
#else
#define ILM_RASTER_OCCUPANCY
#endif
template <int FORMAT>
__global__ __launch_bounds__(256) ILM_RASTER_OCCUPANCY void raster_tiles_kernel(const RasterLaunch a) {
#ifdef ILM_RASTER_TRACE
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    __shared__ Sprite batch[256];
    __shared__ uint8_t list[4][256];
    __shared__ int wave_count[4][4];          // [loader wave][quadrant]
    const int tiles = a.tiles_x * a.tiles_y;
    const uint32_t item = blockIdx.x;
    if (item >= a.tile_first_item[tiles]) return;           // the grid is an upper bound of the work items
    // the tile this work item belongs to: last tile whose first item is <= item (tiles without sprites own no item)
    int lo = 0, hi = tiles;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.tile_first_item[mid] <= item) lo = mid; else hi = mid;
    }
    const int tile = lo;
    const uint32_t segment = item - a.tile_first_item[tile], segments = a.tile_segments[tile];
    const int64_t run_begin = a.tile_begin[tile], run_end = a.tile_begin[tile + 1];
    const int64_t begin = run_begin + (int64_t)segment * kRasterSegment;
    const int64_t end = (begin + kRasterSegment < run_end) ? begin + kRasterSegment : run_end;
    const bool direct = segments == 1u;
    const int tid = (int)threadIdx.x;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    // each wave owns an 8 x 8 quadrant of the tile (lane = 8 * row + column)
    const int wave = tid >> 6, lane = tid & 63;
    const int qx = tx * kRasterTile + (wave & 1) * 8, qy = ty * kRasterTile + (wave >> 1) * 8;
    const int x = qx + (lane & 7), y = qy + (lane >> 3);
    const bool in_image = (x < a.width) && (y < a.height);
    const size_t o = (size_t)y * (size_t)a.width + (size_t)x;
    float4 dst = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    float transmittance = 1.0f;
    if (in_image && direct) dst = load_target<FORMAT>(a.target, o);
    const float pcx = (float)x + 0.5f, pcy = (float)y + 0.5f;
    const float tcx = (float)(tx * kRasterTile) + 4.0f, tcy = (float)(ty * kRasterTile) + 4.0f;     // centre of quadrant 0
    const bool rounded = a.params.RenderingOptions[0] != 0.0f;
    const bool dithered = a.params.RenderingOptions[1] >= 0.5f;
    const bool additive = a.params.BlendMode == ILM_BLEND_ADDITIVE;
    const int filter = a.params.BitmapFilter;
    const float region_x = a.params.BitmapTextureRegion.x, region_y = a.params.BitmapTextureRegion.y;
    const float region_w = a.params.BitmapTextureRegion.z - region_x, region_h = a.params.BitmapTextureRegion.w - region_y;
    uint32_t shaded = 0;
    for (int64_t base = begin; base < end; base += 256) {
        const int n = (int)((end - base < 256) ? (end - base) : 256);
        __syncthreads();                                    // the previous batch has been consumed
        bool hit[4] = { false, false, false, false };
        if (tid < n) {
            const Sprite sp = a.sprites[(uint32_t)(a.sorted_keys[base + tid] & 0xFFFFFFFFull)];
            batch[tid] = sp;
            // pixel centres of a quadrant lie within +-3.5 of its centre; ex / ey carry a pixel of slack
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float cx = tcx + (float)((q & 1) * 8), cy = tcy + (float)((q >> 1) * 8);
                hit[q] = (fabsf(sp.cx - cx) - sp.ex <= 3.5f) && (fabsf(sp.cy - cy) - sp.ey <= 3.5f);
            }
        }
        unsigned long long mask[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            mask[q] = __ballot(hit[q]);
            if (lane == 0) wave_count[wave][q] = __popcll(mask[q]);
        }
        __syncthreads();
        int total = 0;                                      // sprites on this wave's list
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int c = wave_count[w][q];
                before += (w < wave) ? c : 0;
                all += c;
            }
            if (hit[q])
                list[q][before + __popcll(mask[q] & ((1ull << lane) - 1ull))] = (uint8_t)tid;
            total = (q == wave) ? all : total;
        }
        __syncthreads();
        total = __builtin_amdgcn_readfirstlane(total);
        if (!in_image || total == 0) continue;
        // a record = four 16-byte quads: (cx, cy, ex, ey) (i00, i01, i10, i11) (r, g, b, a) (rounding, frame u, v, dither frame).  The
        // two geometry quads of the next sprite are in flight while this one is shaded; the other two are fetched only when some lane
        // of the wave is covered
        const float4* quads = reinterpret_cast<const float4*>(batch);
        int cur = (int)list[wave][0];
        float4 ng0 = quads[4 * cur], ng1 = quads[4 * cur + 1];
        for (int k = 0; k < total; k++) {
            const float4 g0 = ng0, g1 = ng1;
            const int here = cur;
            cur = (int)list[wave][(k + 1 < total) ? k + 1 : k];
            ng0 = quads[4 * cur]; ng1 = quads[4 * cur + 1];
            const float dx = pcx - g0.x, dy = pcy - g0.y;
            const float u = (g1.x * dx) + (g1.y * dy), v = (g1.z * dx) + (g1.w * dy);
            const bool covered = (u >= -1.0f) && (u < 1.0f) && (v >= -1.0f) && (v < 1.0f);
            if (__ballot(covered) == 0ull)
                continue;
            const float4 c0 = quads[4 * here + 2], c1 = quads[4 * here + 3];
            struct { float r, g, b, a, rounding, frame_u, frame_v, dither_frame; } sp = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
            if (!covered)
                continue;
            float alpha = 1.0f;
            if (rounded) {
                // computeCircularAlpha
                const float distance = sqrtf((u * u) + (v * v));
                const float power = fmaxf(sp.rounding, 0.01f);
                const float divisor = fmaxf(sat(1.0f - power), 0.001f);
                const float distance_from_edge = sat(distance - power) / divisor;
                alpha = sat(1.0f - pow_pos(distance_from_edge, power));
            }
            float cr = sp.r, cg = sp.g, cb = sp.b, ca = sp.a;
            if ((filter != ILM_BITMAP_NONE) && (ca > 0.0f)) {       // PS_Texture: `color.a > (1 / 512)`, an integer division
                // texCoord = lerp(region.xy, region.zw, unit / 2 + 0.5) + frameTexCoord, interpolated over the quad
                const float tu = (region_x + (region_w * ((u / 2.0f) + 0.5f))) + sp.frame_u;
                const float tv = (region_y + (region_h * ((v / 2.0f) + 0.5f))) + sp.frame_v;
                const float4 t = bitmap_fetch(a.bitmap, a.bitmap_w, a.bitmap_h, tu, tv, filter);
                cr = (cr * t.x) * a.params.GlobalColor.x; cg = (cg * t.y) * a.params.GlobalColor.y;
                cb = (cb * t.z) * a.params.GlobalColor.z; ca = (ca * t.w) * a.params.GlobalColor.w;
            }
            float sr = cr * alpha, sg = cg * alpha, sb = cb * alpha, sa = ca * alpha;
            if (dithered) {
                // premultipliedToDithered, RasterizeParticleSystem.fx:158-175, with Dither64 of Fracture's DitherCommon.fxh (outside the tree)
                // = Jimenez' published frac(dot(float3(vpos, frame), uint3(33, 52, 25) / 64.0)); GET_VPOS = floor(vpos); every term is a
                // multiple of 1/64 below 2^18, so the sum is exact in any order
                const float dd = (((float)x * (33.0f / 64.0f)) + ((float)y * (52.0f / 64.0f))) + (sp.dither_frame * (25.0f / 64.0f));
                const float d64 = dd - floorf(dd);
                if ((sa <= d64) || (sa <= (ref::kDitherDiscardNumerator / 255.0f))) {
                    sr = sg = sb = sa = 0.0f;
                } else {
                    const float al = fmaxf(sa, 0.0001f);
                    sr = sr / al; sg = sg / al; sb = sb / al; sa = 1.0f;
                }
            }
            if (sa <= 0.0f)                                 // `result.a <= (1 / 512)`: an integer division in the shader, i.e. <= 0
                continue;
            shaded++;
            const float keep = additive ? 1.0f : (1.0f - sa);
            dst.x = sr + (dst.x * keep); dst.y = sg + (dst.y * keep);
            dst.z = sb + (dst.z * keep); dst.w = sa + (dst.w * keep);
            transmittance *= keep;
        }
    }
    if (direct) {
        if (in_image) store_target<FORMAT>(a.target, o, dst);
    } else {
        // (C, T) of this segment over black, lane-major so the combine pass reads it coalesced
        float* out = a.partials + ((size_t)(a.tile_first_partial[tile] + segment) * 5u) * 256u;
        out[tid] = dst.x; out[256 + tid] = dst.y; out[512 + tid] = dst.z; out[768 + tid] = dst.w; out[1024 + tid] = transmittance;
    }
    if (a.count_shaded) {
        for (int off = 32; off > 0; off >>= 1) shaded += __shfl_down(shaded, off);
        if ((lane == 0) && shaded != 0u) atomicAdd(&a.stats[2], (unsigned long long)shaded);
    }
#ifdef ILM_RASTER_TRACE
    if (tid == 0) {
        const unsigned w = item & 131071u;
        g_raster_trace[4 * w] = trace_t0; g_raster_trace[4 * w + 1] = __builtin_amdgcn_s_memrealtime();
        g_raster_trace[4 * w + 2] = (unsigned long long)(end - begin); g_raster_trace[4 * w + 3] = ((unsigned long long)segments << 32) | segment;
    }
#endif
}

// crowded tiles: dst = C_j + T_j * dst for the segments j in order
template <int FORMAT>
__global__ __launch_bounds__(256) void raster_combine_kernel(const RasterLaunch a) {
    const int tile = (int)blockIdx.x;
    const uint32_t segments = a.tile_segments[tile];
    if (segments <= 1u) return;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int x = tx * kRasterTile + (wave & 1) * 8 + (lane & 7), y = ty * kRasterTile + (wave >> 1) * 8 + (lane >> 3);
    if ((x >= a.width) || (y >= a.height)) return;
    const size_t o = (size_t)y * (size_t)a.width + (size_t)x;
    float4 dst = load_target<FORMAT>(a.target, o);
    const float* part = a.partials + ((size_t)a.tile_first_partial[tile] * 5u) * 256u;
    for (uint32_t j = 0; j < segments; j++, part += 5 * 256) {
        const float t = part[1024 + tid];
        dst.x = part[tid] + (dst.x * t); dst.y = part[256 + tid] + (dst.y * t);
        dst.z = part[512 + tid] + (dst.z * t); dst.w = part[768 + tid] + (dst.w * t);
    }
    store_target<FORMAT>(a.target, o, dst);
}

template <int FORMAT>
__global__ __launch_bounds__(256) void clear_target_kernel(void* texels, size_t n, float4 c) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) store_target<FORMAT>(texels, i, c);
}

hipError_t launch_clear_target(void* texels, int format, size_t n, float4 color, hipStream_t stream) {
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (format == ILM_LIGHTMAP_FLOAT4) hipLaunchKernelGGL(clear_target_kernel<ILM_LIGHTMAP_FLOAT4>, grid, block, 0, stream, texels, n, color);
    else if (format == ILM_LIGHTMAP_HALF4) hipLaunchKernelGGL(clear_target_kernel<ILM_LIGHTMAP_HALF4>, grid, block, 0, stream, texels, n, color);
    else hipLaunchKernelGGL(clear_target_kernel<ILM_LIGHTMAP_RGBA8>, grid, block, 0, stream, texels, n, color);
    return hipGetLastError();
}

#define RASTER_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

static hipError_t grow(void** p, size_t* cap, size_t bytes, hipStream_t stream) {
    if (bytes <= *cap) return hipSuccess;
    RASTER_TRY(hipStreamSynchronize(stream));
    if (*p) RASTER_TRY(hipFree(*p));
    *p = nullptr; *cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    RASTER_TRY(hipMalloc(p, want));
    *cap = want;
    return hipSuccess;
}

void free_raster_scratch(RasterScratch& s) {
    void** ptrs[] = { &s.sprites, &s.counts, &s.offsets, &s.keys, &s.sorted_keys, &s.temp, &s.stats, &s.tiles, &s.partials, &s.rects };
    for (void** p : ptrs) { if (*p) (void)hipFree(*p); *p = nullptr; }
    s.sprites_cap = s.counts_cap = s.offsets_cap = s.keys_cap = s.sorted_cap = s.temp_cap = s.tiles_cap = s.partials_cap = s.rects_cap = 0;
    if (s.host_words) (void)hipHostFree(s.host_words);
    s.host_words = nullptr;
}

// the frame's pair count (and, at the end, its statistics) reach the host through a pinned word the kernel writes: one stream
// synchronisation, no staged device-to-host copies (two of them cost the r03 frame 55 us of idle device around 8 us of copying)
__global__ void raster_total_kernel(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts, size_t n, unsigned long long* host_word) {
    *host_word = (unsigned long long)offsets[n - 1] + (unsigned long long)counts[n - 1];
    __threadfence_system();
}
__global__ void raster_stats_kernel(const unsigned long long* __restrict__ stats, unsigned long long* host_words) {
    for (int k = 0; k < 4; k++) host_words[k] = stats[k];
    __threadfence_system();
}

// setup -> scan -> emit -> sort -> tiles.  One host synchronisation (the pair count sizes the key buffers): a one-thread kernel writes it to a
// pinned word the host reads after the stream drains.
struct SaturatingAdd {
    __host__ __device__ uint32_t operator()(uint32_t x, uint32_t y) const {
        const uint32_t sum = x + y;
        return (sum < x) ? 0xFFFFFFFFu : sum;
    }
};

hipError_t render_particles(RasterLaunch& a, RasterScratch& s, hipStream_t stream, unsigned long long out_stats[3], bool* too_many) {
    *too_many = false;
    const size_t n = (size_t)a.total_slots;
    RASTER_TRY(grow(&s.sprites, &s.sprites_cap, n * sizeof(Sprite), stream));
    RASTER_TRY(grow(&s.counts, &s.counts_cap, n * sizeof(uint32_t), stream));
    RASTER_TRY(grow(&s.offsets, &s.offsets_cap, n * sizeof(uint32_t), stream));
    RASTER_TRY(grow(&s.rects, &s.rects_cap, n * sizeof(uint2), stream));
    a.rects = static_cast<uint2*>(s.rects);
    if (!s.stats) RASTER_TRY(hipMalloc(&s.stats, 4 * sizeof(unsigned long long)));
    RASTER_TRY(hipMemsetAsync(s.stats, 0, 4 * sizeof(unsigned long long), stream));
    a.sprites = static_cast<Sprite*>(s.sprites);
    a.counts = static_cast<uint32_t*>(s.counts);
    a.offsets = static_cast<uint32_t*>(s.offsets);
    a.stats = static_cast<unsigned long long*>(s.stats);
    const dim3 block(256), slot_grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(raster_setup_kernel, slot_grid, block, 0, stream, a);
    RASTER_TRY(hipGetLastError());
    size_t scan_bytes = 0;
    // saturating addition (associative): a total beyond 32 bits reads 0xFFFFFFFF instead of wrapping below the bound checked next
    RASTER_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, a.counts, a.offsets, 0u, n, SaturatingAdd(), stream));
    RASTER_TRY(grow(&s.temp, &s.temp_cap, scan_bytes, stream));
    RASTER_TRY(rocprim::exclusive_scan(s.temp, scan_bytes, a.counts, a.offsets, 0u, n, SaturatingAdd(), stream));
    if (!s.host_words) RASTER_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.host_words), 8 * sizeof(unsigned long long), hipHostMallocMapped));
    unsigned long long* host_words_dev = nullptr;
    RASTER_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&host_words_dev), s.host_words, 0));
    hipLaunchKernelGGL(raster_total_kernel, dim3(1), dim3(1), 0, stream, a.offsets, a.counts, n, host_words_dev);
    RASTER_TRY(hipGetLastError());
    RASTER_TRY(hipStreamSynchronize(stream));
    const unsigned long long pairs = s.host_words[0];
    // (one pair per slot and tile; 2^28 keys = 2 GiB.  The scan saturates, so a wrapped total cannot slip under this bound.)
    if (pairs > (1ull << 28)) { *too_many = true; return hipSuccess; }
    a.pair_count = (int64_t)pairs;
    if (pairs > 0) {
        RASTER_TRY(grow(&s.keys, &s.keys_cap, pairs * sizeof(unsigned long long), stream));
        RASTER_TRY(grow(&s.sorted_keys, &s.sorted_cap, pairs * sizeof(unsigned long long), stream));
        a.keys = static_cast<unsigned long long*>(s.keys);
        a.sorted_keys = static_cast<unsigned long long*>(s.sorted_keys);
        hipLaunchKernelGGL(raster_emit_kernel, slot_grid, block, 0, stream, a);
        RASTER_TRY(hipGetLastError());
        int tile_bits = 1;
        while ((1 << tile_bits) < a.tiles_x * a.tiles_y) tile_bits++;
        size_t sort_bytes = 0;
        // The keys are emitted in slot order (the scan runs over the slots), and a radix sort is stable: sorting on the tile bits
        // alone leaves every tile's run in slot order -- two digit passes instead of six.
        RASTER_TRY(rocprim::radix_sort_keys(nullptr, sort_bytes, a.keys, a.sorted_keys, (size_t)pairs, 32u, (unsigned)(32 + tile_bits), stream));
        RASTER_TRY(grow(&s.temp, &s.temp_cap, sort_bytes, stream));
        RASTER_TRY(rocprim::radix_sort_keys(s.temp, sort_bytes, a.keys, a.sorted_keys, (size_t)pairs, 32u, (unsigned)(32 + tile_bits), stream));
        // tile runs -> segments -> work items (both scans over the tiles in one workgroup; no host round trip: the grid is an upper bound)
        const int tiles = a.tiles_x * a.tiles_y;
        RASTER_TRY(grow(&s.tiles, &s.tiles_cap, (size_t)(tiles + 1) * 5 * sizeof(uint32_t), stream));
        a.tile_begin = static_cast<uint32_t*>(s.tiles);
        a.tile_segments = a.tile_begin + (tiles + 1);
        a.tile_first_item = a.tile_segments + (tiles + 1);
        a.tile_multi = a.tile_first_item + (tiles + 1);
        a.tile_first_partial = a.tile_multi + (tiles + 1);
        hipLaunchKernelGGL(raster_tile_ranges_kernel, dim3((unsigned)((tiles + 1 + 255) / 256)), block, 0, stream, a);
        RASTER_TRY(hipGetLastError());
        hipLaunchKernelGGL(raster_tile_scan_kernel, dim3(1), dim3(1024), 0, stream, a);
        RASTER_TRY(hipGetLastError());
        // sum over tiles of ceil(len / S) <= non-empty tiles + pairs / S; partial slots only for tiles with >= 2 segments: <= 2 pairs / S
        a.work_items = (int32_t)std::min<unsigned long long>((unsigned long long)tiles + pairs / kRasterSegment + 1ull, 0x7FFFFFFFull);
        const size_t partial_slots = (size_t)(2 * (pairs / kRasterSegment) + 2);
        RASTER_TRY(grow(&s.partials, &s.partials_cap, partial_slots * 5 * 256 * sizeof(float), stream));
        a.partials = static_cast<float*>(s.partials);
        const dim3 item_grid((unsigned)a.work_items), tile_grid((unsigned)tiles);
        if (a.format == ILM_LIGHTMAP_FLOAT4) {
            hipLaunchKernelGGL(raster_tiles_kernel<ILM_LIGHTMAP_FLOAT4>, item_grid, block, 0, stream, a);
            hipLaunchKernelGGL(raster_combine_kernel<ILM_LIGHTMAP_FLOAT4>, tile_grid, block, 0, stream, a);
        } else if (a.format == ILM_LIGHTMAP_HALF4) {
            hipLaunchKernelGGL(raster_tiles_kernel<ILM_LIGHTMAP_HALF4>, item_grid, block, 0, stream, a);
            hipLaunchKernelGGL(raster_combine_kernel<ILM_LIGHTMAP_HALF4>, tile_grid, block, 0, stream, a);
        } else {
            hipLaunchKernelGGL(raster_tiles_kernel<ILM_LIGHTMAP_RGBA8>, item_grid, block, 0, stream, a);
            hipLaunchKernelGGL(raster_combine_kernel<ILM_LIGHTMAP_RGBA8>, tile_grid, block, 0, stream, a);
        }
        RASTER_TRY(hipGetLastError());
    }
    if (out_stats) {
        hipLaunchKernelGGL(raster_stats_kernel, dim3(1), dim3(1), 0, stream, static_cast<const unsigned long long*>(s.stats), host_words_dev + 1);
        RASTER_TRY(hipGetLastError());
        RASTER_TRY(hipStreamSynchronize(stream));
        out_stats[0] = s.host_words[1]; out_stats[1] = pairs; out_stats[2] = s.host_words[3];
    }
    return hipSuccess;
}

}  // namespace ilm
