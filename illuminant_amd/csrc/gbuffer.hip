// gbuffer.hip -- RenderGBuffer with the host's meshes: 2.5D height volumes (top + front faces under the depth test) and billboards
// (ilm_gbuffer_render_meshes; Illuminant/Lighting/LightingRenderer.GBuffer.cs:102-478, Illuminant/Shaders/GBuffer.fx, GBufferBitmap.fx,
// GBufferShaderCommon.fxh).  gfx950 only.
//
// The reference hands triangle lists to the Direct3D rasteriser, one draw call after the other into one render target with a 24-bit
// depth buffer.  Here the whole frame is ONE pass over the pixels: a setup kernel runs the three vertex shaders and snaps every
// triangle to the 1/256-pixel grid (one thread per triangle, records in draw order); a binning kernel lists, per 64 x 64 pixel block,
// the triangles whose pixel bounds touch it; the raster kernel gives every wave an 8 x 8 pixel square, reads its block's list 64
// entries at a time (one candidate per lane: bounds and vertices), and walks the candidates that touch the square in list order --
// vertices out of the lanes' registers, attributes through wave-uniform (scalar) loads only for a triangle that covers a pixel -- and
// keeps each pixel's colour and depth in registers until its single store: 16 B (or 8 B) written per texel, nothing read back, no
// atomics, and the reference's draw order is the loop order.  Coverage is exact integer arithmetic (64-bit edge functions, top-left
// rule), so a pixel belongs to exactly one of two triangles sharing an edge, as on the hardware.
#include "internal.hpp"
#include "hlsl_math.hpp"

namespace ilm {

namespace {

constexpr int kGround = 0, kTop = 1, kFace = 2, kMask = 3, kGData = 4;

// round-to-nearest onto the 1/256-pixel grid; positions are confined to +-2^30 so that edge functions fit 64 bits
ILM_DEV int32_t snap(float s) {
    double v = floor((double)s * 256.0 + 0.5);
    if (!(v > -1073741824.0)) v = -1073741824.0;
    if (v > 1073741824.0) v = 1073741824.0;
    return (int32_t)v;
}

ILM_DEV int64_t edge(int32_t ax, int32_t ay, int32_t bx, int32_t by, int64_t px, int64_t py) {
    return ((int64_t)bx - ax) * (py - ay) - ((int64_t)by - ay) * (px - ax);
}

// a * b for numbers within 24 signed bits: the full-rate 24-bit multiplier's two halves instead of a 64 x 64 product
ILM_DEV int64_t mul_i24(int32_t a, int32_t b) {
    const int32_t a24 = (int32_t)((uint32_t)a << 8) >> 8, b24 = (int32_t)((uint32_t)b << 8) >> 8;
    return (int64_t)a24 * (int64_t)b24;
}

// how the raster kernel decides a record's coverage (GBufferPrim::kind keeps the pixel shader; these ride above it in `verts`)
constexpr int kShapeRect = 0x100;      // an axis-aligned rectangle whose pixels ARE its pixel bounds (the ground plane's two triangles as one record)
constexpr int kShapeSmall = 0x200;     // every coordinate and every pixel centre of the frame within +-2^22: the edge functions' factors fit 24 bits

// top-left rule on a clockwise triangle (y down): top edges run left to right, left edges run upwards
ILM_DEV bool edge_owns(int32_t ax, int32_t ay, int32_t bx, int32_t by) { return (by < ay) || ((by == ay) && (bx > ax)); }

// encodeNormalSpherical, EnvironmentCommon.fxh:33-40; encodeGBufferSample, GBufferShaderCommon.fxh:10-35 (fullbright = false)
ILM_DEV void encode_normal(f3 n, float* ex, float* ey) {
    *ex = 0.0f; *ey = 0.0f;
    if ((n.x != 0.0f) || (n.y != 0.0f) || (n.z != 0.0f)) {
        const float nx = (fabsf(n.x) < 0.0001f) ? 0.0001f : n.x;
        *ex = ((atan2f(n.y, nx) / kPi) + 1.0f) * 0.5f;
        *ey = (n.z + 1.0f) * 0.5f;
    }
}
ILM_DEV float4 encode_sample(float ex, float ey, float relative_y, float z, bool dead, bool enable_shadows) {
    if (dead)
        return mk4(0.0f, 0.0f, -ref::kDeadTexel, -ref::kDeadTexel);
    const float w = (((z + ref::kGBufferZOffset) / ref::kGBufferZScale) * (enable_shadows ? 1.0f : -1.0f)) + (enable_shadows ? 0.0f : -1.0f);
    return mk4(ex, ey, relative_y, w);
}

constexpr int kFlatNormalBit = 16;                              // GBufferPrim::flat: enc_x / enc_y hold the triangle's encoded normal

// An attribute with the same finite value v at the three vertices interpolates to (v + 0 f1) + 0 f2 = v + 0 at every covered pixel
// (f1, f2 are finite and not negative there), whatever the weights: the raster kernel skips the weights for a triangle of such
// attributes only -- the ground plane, every frame -- and takes a flat normal's encoding from here.
ILM_DEV void finish_attributes(GBufferPrimSetup& p) {
    p.flat = 0; p.enc_x = 0.0f; p.enc_y = 0.0f;
    for (int k = 0; k < kGBufferAttrs; k++) {
        const uint32_t b0 = __float_as_uint(p.a[0][k]), b1 = __float_as_uint(p.a[1][k]), b2 = __float_as_uint(p.a[2][k]);
        if ((b0 == b1) && (b0 == b2) && ((b0 & 0x7F800000u) != 0x7F800000u))
            p.flat |= 1 << k;
    }
    if ((p.kind == kGround) || (((p.flat >> 3) & 7) == 7 && ((p.kind == kTop) || (p.kind == kFace)))) {
        const f3 n = (p.kind == kGround) ? mk3(0.0f, 0.0f, 1.0f) : mk3(p.a[0][3] + 0.0f, p.a[0][4] + 0.0f, p.a[0][5] + 0.0f);
        encode_normal(n, &p.enc_x, &p.enc_y);
        p.flat |= 1 << kFlatNormalBit;
    }
}

ILM_DEV void finish(GBufferPrimSetup& p, const float sx[3], const float sy[3]) {
    for (int k = 0; k < 3; k++) { p.x[k] = snap(sx[k]); p.y[k] = snap(sy[k]); }
    const int64_t area = edge(p.x[0], p.y[0], p.x[1], p.y[1], p.x[2], p.y[2]);
    p.flat = 0; p.enc_x = 0.0f; p.enc_y = 0.0f;
    if (area == 0) { p.kind = -1; p.i0 = p.j0 = 1; p.i1 = p.j1 = 0; return; }
    if (area < 0) {                                              // CullMode.None: the other winding is drawn too
        int32_t t = p.x[1]; p.x[1] = p.x[2]; p.x[2] = t;
        t = p.y[1]; p.y[1] = p.y[2]; p.y[2] = t;
        for (int k = 0; k < kGBufferAttrs; k++) { const float f = p.a[1][k]; p.a[1][k] = p.a[2][k]; p.a[2][k] = f; }
    }
    const int32_t x0 = min(p.x[0], min(p.x[1], p.x[2])), x1 = max(p.x[0], max(p.x[1], p.x[2]));
    const int32_t y0 = min(p.y[0], min(p.y[1], p.y[2])), y1 = max(p.y[0], max(p.y[1], p.y[2]));
    // pixel centres 256 i + 128 inside [x0, x1]
    p.i0 = (int32_t)(((int64_t)x0 - 128 + 255) >> 8); p.i1 = (int32_t)(((int64_t)x1 - 128) >> 8);
    p.j0 = (int32_t)(((int64_t)y0 - 128 + 255) >> 8); p.j1 = (int32_t)(((int64_t)y1 - 128) >> 8);
    finish_attributes(p);
}

// GroundPlaneVertexShader / HeightVolumeVertexShader / HeightVolumeFaceVertexShader, GBuffer.fx:7-55
// attributes: 0-2 worldPosition, 3-5 normal, 6 enableShadows, 7 result.z, 8 dead
ILM_DEV void volume_prim(GBufferPrimSetup& p, int kind, const IlmHeightVolumeVertex& v0, const IlmHeightVolumeVertex& v1,
                         const IlmHeightVolumeVertex& v2, const IlmGBufferMeshDesc& d) {
    const IlmHeightVolumeVertex* v[3] = { &v0, &v1, &v2 };
    float sx[3], sy[3];
    p.kind = kind; p.texture = -1;
    for (int k = 0; k < 3; k++) {
        const float x = v[k]->Position[0], z = v[k]->Position[2];
        float y = v[k]->Position[1];
        for (int c = 0; c < kGBufferAttrs; c++) p.a[k][c] = 0.0f;
        p.a[k][0] = x; p.a[k][1] = y; p.a[k][2] = z;
        p.a[k][3] = v[k]->Normal[0]; p.a[k][4] = v[k]->Normal[1]; p.a[k][5] = v[k]->Normal[2];
        p.a[k][6] = v[k]->EnableShadows;
        if (kind == kGround) {
            p.a[k][7] = 0.0f;
            p.a[k][8] = (z < -9999.0f) ? 1.0f : 0.0f;
        } else {
            y -= d.ZToYMultiplier * z;
            p.a[k][7] = z / d.DistanceFieldExtentZ;
        }
        sx[k] = (x - d.ViewportPosition[0]) * d.ViewportScale[0];
        sy[k] = (y - d.ViewportPosition[1]) * d.ViewportScale[1];
    }
    finish(p, sx, sy);
}

// BillboardVertexShader, GBufferBitmap.fx:12-27 (POSITION0 carries two floats, Vertices.cs:89: position.z reads 0)
// attributes: 0-2 worldPosition, 3-5 normal, 6-7 texCoord, 8 screenPosition.y, 9-10 dataScaleAndDynamicFlag
ILM_DEV void billboard_prim(GBufferPrimSetup& p, int kind, int texture, const IlmBillboardVertex& v0, const IlmBillboardVertex& v1,
                            const IlmBillboardVertex& v2, const IlmGBufferMeshDesc& d) {
    const IlmBillboardVertex* v[3] = { &v0, &v1, &v2 };
    float sx[3], sy[3];
    p.kind = kind; p.texture = texture;
    for (int k = 0; k < 3; k++) {
        for (int c = 0; c < kGBufferAttrs; c++) p.a[k][c] = 0.0f;
        for (int c = 0; c < 3; c++) {
            p.a[k][c] = v[k]->WorldPosition[c] + (d.SelfOcclusionHack * v[k]->Normal[c]);
            p.a[k][3 + c] = v[k]->Normal[c];
        }
        p.a[k][6] = v[k]->TexCoord[0]; p.a[k][7] = v[k]->TexCoord[1];
        p.a[k][8] = v[k]->ScreenPosition[1];
        p.a[k][9] = v[k]->DataScaleAndDynamicFlag[0]; p.a[k][10] = v[k]->DataScaleAndDynamicFlag[1];
        sx[k] = (v[k]->ScreenPosition[0] - d.ViewportPosition[0]) * d.ViewportScale[0];
        sy[k] = (v[k]->ScreenPosition[1] - d.ViewportPosition[1]) * d.ViewportScale[1];
    }
    finish(p, sx, sy);
}

// tex2D through the POINT / CLAMP sampler (LightingRenderer.GBuffer.cs:301-307); nothing bound reads (0, 0, 0, 1)
ILM_DEV float4 sample_point(const GBufferTex* textures, int index, float u, float v) {
    if (index < 0)
        return mk4(0.0f, 0.0f, 0.0f, 1.0f);
    const GBufferTex t = textures[index];
    if (t.texels == nullptr)
        return mk4(0.0f, 0.0f, 0.0f, 1.0f);
    const float fx = floorf(u * (float)t.width), fy = floorf(v * (float)t.height);
    const int x = !(fx >= 0.0f) ? 0 : ((fx > (float)(t.width - 1)) ? t.width - 1 : (int)fx);
    const int y = !(fy >= 0.0f) ? 0 : ((fy > (float)(t.height - 1)) ? t.height - 1 : (int)fy);
    const size_t o = (size_t)y * (size_t)t.width + (size_t)x;
    if (t.format == ILM_LIGHTMAP_RGBA8) {
        const uint32_t c = reinterpret_cast<const uint32_t*>(t.texels)[o];
        return mk4((float)(c & 0xFFu) / 255.0f, (float)((c >> 8) & 0xFFu) / 255.0f, (float)((c >> 16) & 0xFFu) / 255.0f, (float)(c >> 24) / 255.0f);
    }
    if (t.format == ILM_LIGHTMAP_HALF4) {
        const uint2 hh = reinterpret_cast<const uint2*>(t.texels)[o];
        return mk4(__half2float(__ushort_as_half((unsigned short)(hh.x & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(hh.x >> 16))),
                   __half2float(__ushort_as_half((unsigned short)(hh.y & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(hh.y >> 16))));
    }
    return reinterpret_cast<const float4*>(t.texels)[o];
}

}  // namespace

// one thread per triangle of the frame, in draw order: [ground plane 2][top][front][billboard quads 2 each]
__global__ __launch_bounds__(64) void gbuffer_setup_kernel(const GBufferMeshLaunch a) {
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    // (the vertex arrays and tables are read where the host left them -- a pinned slot -- when the call passes them in place; the
    // texture table, which the raster kernel reads per billboard, moves to device memory here)
    if (a.textures_in != nullptr)
        for (int k = t; k < a.texture_count; k += (int)(gridDim.x * blockDim.x))
            const_cast<GBufferTex*>(a.textures)[k] = a.textures_in[k];
    if (t >= a.prim_count) return;
    GBufferPrimSetup p;
    const IlmGBufferMeshDesc& d = a.desc;
    const int quad_indices[6] = { 0, 1, 3, 1, 2, 3 };           // QuadIndices, LightingRenderer.cs:421-423
    int shape = 0;
    if (t < 2) {
        // RenderGroundPlane, LightingRenderer.GBuffer.cs:271-299: the quad's two triangles.  They carry the same attributes at every
        // corner and share the diagonal, so together they cover -- under the top-left rule: left and top edge in, right and bottom edge
        // out, every pixel of the diagonal in exactly one of them -- the pixels whose centres the snapped rectangle holds.  Record 0 is
        // that rectangle (the raster kernel compares pixel indices instead of evaluating six edge functions with 2^30-sized corners
        // under every pixel of every frame); record 1 is empty.
        const float gz = d.GroundZ + (d.RenderGroundPlane ? 0.0f : ref::kGroundLift);
        const float e = ref::kGroundHalfExtent;
        const float cx[4] = { -e, e, e, -e }, cy[4] = { -e, -e, e, e };
        IlmHeightVolumeVertex g[3];
        auto corners = [&](int first) {
            for (int k = 0; k < 3; k++) {
                const int c = quad_indices[first + k];           // triangle 0: corners 0, 1, 3 = (x0, y0), (x1, y0), (x0, y1) of the rectangle
                g[k].Position[0] = cx[c]; g[k].Position[1] = cy[c]; g[k].Position[2] = gz;
                g[k].Normal[0] = 0.0f; g[k].Normal[1] = 0.0f; g[k].Normal[2] = 1.0f;
                g[k].ZRange[0] = d.GroundZ; g[k].ZRange[1] = d.GroundZ;
                g[k].EnableShadows = d.EnableGroundShadows ? 1.0f : 0.0f;
            }
        };
        corners(0);
        volume_prim(p, kGround, g[0], g[1], g[2], d);
        // The rectangle from the extremes of the three snapped corners, whatever order finish() left them in (a negative ViewportScale on
        // one axis mirrors the quad and swaps two vertices; on both axes it reverses left and right, top and bottom -- ADVICE r04: the
        // equality test on fixed vertex slots then dropped the ground plane): an axis-aligned right triangle has exactly two x and two y.
        const int32_t x0 = min(min(p.x[0], p.x[1]), p.x[2]), x1 = max(max(p.x[0], p.x[1]), p.x[2]);
        const int32_t y0 = min(min(p.y[0], p.y[1]), p.y[2]), y1 = max(max(p.y[0], p.y[1]), p.y[2]);
        bool axis_aligned = (x1 > x0) && (y1 > y0);
        for (int k = 0; k < 3; k++) axis_aligned = axis_aligned && (p.x[k] == x0 || p.x[k] == x1) && (p.y[k] == y0 || p.y[k] == y1);
        if ((p.kind == kGround) && axis_aligned) {
            p.i0 = (int32_t)(((int64_t)x0 - 128 + 255) >> 8); p.i1 = (int32_t)(((int64_t)x1 - 128 + 255) >> 8) - 1;
            p.j0 = (int32_t)(((int64_t)y0 - 128 + 255) >> 8); p.j1 = (int32_t)(((int64_t)y1 - 128 + 255) >> 8) - 1;
            shape = kShapeRect;
            if ((t != 0) || (p.i1 < p.i0) || (p.j1 < p.j0)) { p.kind = -1; p.i0 = p.j0 = 1; p.i1 = p.j1 = 0; shape = 0; }
        } else if (p.kind == kGround) {
            // not a rectangle the shortcut can describe (no transform of the reference produces this): the quad's two real triangles
            corners(3 * t);
            volume_prim(p, kGround, g[0], g[1], g[2], d);
        } else if (t != 0) {
            p.kind = -1; p.i0 = p.j0 = 1; p.i1 = p.j1 = 0;
        }
    } else if (t < 2 + a.top_triangles) {
        const IlmHeightVolumeVertex* v = a.top + 3 * (size_t)(t - 2);
        volume_prim(p, d.TwoPointFiveD ? kTop : kGround, v[0], v[1], v[2], d);
    } else if (t < 2 + a.top_triangles + a.front_triangles) {
        const IlmHeightVolumeVertex* v = a.front + 3 * (size_t)(t - 2 - a.top_triangles);
        volume_prim(p, kFace, v[0], v[1], v[2], d);
    } else {
        const int b = t - 2 - a.top_triangles - a.front_triangles;
        const int4 q = a.quads[b >> 1];                          // (quad, texture, kind, -)
        const IlmBillboardVertex* v = a.billboards + 4 * (size_t)q.x;
        const int* ix = quad_indices + 3 * (b & 1);
        billboard_prim(p, q.z, q.y, v[ix[0]], v[ix[1]], v[ix[2]], d);
    }
    GBufferPrim out;
    for (int k = 0; k < kGBufferAttrs; k++) out.attr[k] = mk4(p.a[0][k], p.a[1][k] - p.a[0][k], p.a[2][k] - p.a[0][k], 0.0f);
    out.flat = p.flat; out.enc_x = p.enc_x; out.enc_y = p.enc_y; out._pad = 0;
    a.prims[t] = out;
    a.bounds[t] = make_int4(p.i0, p.i1, p.j0, p.j1);
    a.verts[2 * (size_t)t] = make_int4(p.x[0], p.y[0], p.x[1], p.y[1]);
    const int32_t reach = 1 << 22;
    bool small = (a.width <= 16384) && (a.height <= 16384);
    for (int k = 0; k < 3; k++)
        small = small && (p.x[k] > -reach) && (p.x[k] < reach) && (p.y[k] > -reach) && (p.y[k] < reach);
    if ((shape == 0) && small) shape = kShapeSmall;
    a.verts[2 * (size_t)t + 1] = make_int4(p.x[2], p.y[2], (p.kind < 0) ? -1 : (p.kind | shape), p.texture);
}

// Binning, in draw order.  gbuffer_bin_kernel: one workgroup per BLOCK of pixels (64 x 64, or larger when the frame's triangle count
// times its block count would outgrow the scratch budget) tests every triangle's pixel bounds against the block, 1024 per round (four
// coalesced 16 B loads in flight per thread), and appends the hits to the block's list in global memory (ballot + popcount prefix
// within a wave, the waves' counts through LDS).  gbuffer_meshes_kernel: one workgroup per 16 x 16 pixel tile, one wave per 8 x 8
// quarter of it; each wave filters ITS BLOCK'S list -- tens to hundreds of entries instead of the frame's thousands -- by ballot, 64
// candidates at a time, and walks the set bits.  No list in LDS, no barrier: a tile crossed by any number of triangles is the same loop.
constexpr int kBinRound = 1024;                                 // candidates per round: 4 per thread

// appends the round's hits (thread: candidates `wave * 256 + k * 64 + lane`, k = 0..3) behind `len` entries in draw order; returns the new length
template <typename Put>
ILM_DEV int append_hits(const bool (&hit)[4], int lane, int wave, int len, int* s_count, Put put) {
    uint64_t mask[4];
    int mine = 0;
    for (int k = 0; k < 4; k++) { mask[k] = __ballot(hit[k]); mine += __popcll(mask[k]); }
    if (lane == 0) s_count[wave] = mine;
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < 4; w++) { const int c = s_count[w]; total += c; if (w < wave) before += c; }
    int pos = len + before;
    for (int k = 0; k < 4; k++) {
        if (hit[k]) put(pos + __popcll(mask[k] & ((1ull << lane) - 1ull)), k);
        pos += __popcll(mask[k]);
    }
    __syncthreads();
    return len + total;
}

__global__ __launch_bounds__(256) void gbuffer_bin_kernel(const GBufferMeshLaunch a) {
    __shared__ int s_count[4];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int block = (int)blockIdx.x;
    const int lo_i = (block % a.block_cols) << a.block_shift, lo_j = (block / a.block_cols) << a.block_shift;
    const int hi_i = lo_i + (1 << a.block_shift) - 1, hi_j = lo_j + (1 << a.block_shift) - 1;
    int* list = a.block_list + (size_t)block * (size_t)a.prim_count;
    int len = 0;
    for (int next = 0; next < a.prim_count; next += kBinRound) {
        bool hit[4];
        int t[4];
        for (int k = 0; k < 4; k++) {
            t[k] = next + wave * 256 + k * 64 + lane;
            const int4 b = a.bounds[min(t[k], a.prim_count - 1)];     // (i0, i1, j0, j1); empty for a degenerate triangle
            hit[k] = (t[k] < a.prim_count) && (b.x <= b.y) && (b.z <= b.w) && !((b.y < lo_i) || (b.x > hi_i) || (b.w < lo_j) || (b.z > hi_j));
        }
        len = append_hits(hit, lane, wave, len, s_count, [&](int at, int k) { list[at] = t[k]; });
    }
    if (threadIdx.x == 0) a.block_count[block] = len;
}

__global__ __launch_bounds__(256) void gbuffer_meshes_kernel(const GBufferMeshLaunch a) {
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int bi0 = (int)blockIdx.x * 16, bj0 = (int)blockIdx.y * 16;
    const int ti0 = bi0 + (wave & 1) * 8, tj0 = bj0 + (wave >> 1) * 8;                                    // wave-uniform
    const int i = ti0 + (lane & 7), j = tj0 + (lane >> 3);
    const int32_t px = 256 * i + 128, py = 256 * j + 128;     // the pixel's centre on the 1/256 grid (a frame side is below 2^22 pixels)
    const IlmGBufferMeshDesc& d = a.desc;
    float4 texel = mk4(0.0f, 0.0f, 0.0f, 0.0f);                  // ClearBatch(Color.Transparent, clearZ: 0), :147-150
    uint32_t depth = 0;
    const int tile_i0 = __builtin_amdgcn_readfirstlane(ti0), tile_j0 = __builtin_amdgcn_readfirstlane(tj0);
    const int block = (bj0 >> a.block_shift) * a.block_cols + (bi0 >> a.block_shift);
    const int* candidates = a.block_list + (size_t)block * (size_t)a.prim_count;
    const int candidate_count = a.block_count[block];
    for (int next = 0; next < candidate_count; next += 64) {
        // 64 candidates at a time, one per lane: its pixel bounds against this wave's 8 x 8 pixels and its vertices for the walk
        const int mine = candidates[min(next + lane, candidate_count - 1)];
        const int4 bounds = a.bounds[mine];
        const int4 v01 = a.verts[2 * (size_t)mine], v2k = a.verts[2 * (size_t)mine + 1];
        const bool hit = (next + lane < candidate_count) &&
                         !((bounds.y < tile_i0) || (bounds.x > tile_i0 + 7) || (bounds.w < tile_j0) || (bounds.z > tile_j0 + 7));
        uint64_t todo = __ballot(hit);
        while (todo != 0) {                                      // draw order: the list's order, lane by lane
            const int from = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const int t = __builtin_amdgcn_readlane(mine, from);
            const int32_t x0 = __builtin_amdgcn_readlane(v01.x, from), y0 = __builtin_amdgcn_readlane(v01.y, from);
            const int32_t x1 = __builtin_amdgcn_readlane(v01.z, from), y1 = __builtin_amdgcn_readlane(v01.w, from);
            const int32_t x2 = __builtin_amdgcn_readlane(v2k.x, from), y2 = __builtin_amdgcn_readlane(v2k.y, from);
            const int kind_and_shape = __builtin_amdgcn_readlane(v2k.z, from), kind = kind_and_shape & 0xFF;
            int64_t w1, w2;
            bool in;
            if (kind_and_shape & kShapeRect) {
                in = (i >= __builtin_amdgcn_readlane(bounds.x, from)) && (i <= __builtin_amdgcn_readlane(bounds.y, from)) &&
                     (j >= __builtin_amdgcn_readlane(bounds.z, from)) && (j <= __builtin_amdgcn_readlane(bounds.w, from));
                w1 = 0; w2 = 0;                                  // its attributes are flat: no weights
            } else {
                int64_t w0;
                if (kind_and_shape & kShapeSmall) {
                    const int32_t qx0 = px - x0, qy0 = py - y0, qx1 = px - x1, qy1 = py - y1, qx2 = px - x2, qy2 = py - y2;
                    w0 = mul_i24(x2 - x1, qy1) - mul_i24(y2 - y1, qx1);
                    w1 = mul_i24(x0 - x2, qy2) - mul_i24(y0 - y2, qx2);
                    w2 = mul_i24(x1 - x0, qy0) - mul_i24(y1 - y0, qx0);
                } else {
                    w0 = edge(x1, y1, x2, y2, px, py);
                    w1 = edge(x2, y2, x0, y0, px, py);
                    w2 = edge(x0, y0, x1, y1, px, py);
                }
                in = (w0 | w1 | w2) >= 0;
                // a centre exactly on an edge belongs to the triangle that owns the edge: rare enough to ask the wave first
                if (__any(in && ((w0 == 0) || (w1 == 0) || (w2 == 0)))) {
                    in = in && ((w0 != 0) || edge_owns(x1, y1, x2, y2));
                    in = in && ((w1 != 0) || edge_owns(x2, y2, x0, y0));
                    in = in && ((w2 != 0) || edge_owns(x0, y0, x1, y1));
                }
            }
            if (!in)
                continue;
            const GBufferPrim& p = a.prims[t];                  // wave-uniform: the attributes arrive through scalar loads
            // the weights, unless every attribute this triangle's shader reads is flat (setup kernel)
            const int flat = p.flat;
            const int reads = (kind == kGround) ? 0x1C4 : (((kind == kTop) || (kind == kFace)) ? 0xFC : 0x7FF);
            // interpolated as (a0 + (a1 - a0) f1) + (a2 - a0) f2; with every attribute flat the weights are left at 0 and the same
            // expression returns a0 + 0, as the setup kernel's note on `flat` says (one branch per triangle, not one per attribute)
            float f1 = 0.0f, f2 = 0.0f;
            if ((~flat & reads) != 0) {
                const double area = (double)edge(x0, y0, x1, y1, x2, y2);       // = w0 + w1 + w2 at every pixel (wave-uniform)
                f1 = (float)((double)w1 / area); f2 = (float)((double)w2 / area);
            }
            auto at = [&](int k) { const float4 A = p.attr[k]; return (A.x + A.y * f1) + A.z * f2; };
            // Clip against the near / far plane (w = 1).  Volumes only: a billboard's POSITION0 is a Vector2 (Vertices.cs:89), so
            // BillboardVertexShader's result.z = position.z / DistanceFieldExtent.z is 0 and never clipped -- and attribute 7 of a
            // billboard is TexCoord.y, which may leave [0, 1] (atlas sub-rectangles with a margin; the sampler clamps)
            const float z = ((kind == kMask) || (kind == kGData)) ? 0.0f : at(7);
            if (!((z >= 0.0f) && (z <= 1.0f)))
                continue;
            float4 out;
            if (kind == kGround) {                                   // GroundPlanePixelShader, GBuffer.fx:57-70
                const float wz = at(2);
                if (wz < d.GroundZ) continue;
                out = encode_sample(p.enc_x, p.enc_y, 0.0f, wz, at(8) != 0.0f, at(6) > 0.5f);
            } else if ((kind == kTop) || (kind == kFace)) {          // HeightVolumePixelShader :72-85 / HeightVolumeFacePixelShader :87-103
                const float wz = at(2);
                const f3 n = mk3(at(3), at(4), at(5));
                f3 bias = mk3(0.0f, 0.0f, d.ZSelfOcclusionHack);
                if (kind == kFace) {
                    if (wz < d.GroundZ) continue;
                    bias = mk3(d.SelfOcclusionHack, d.SelfOcclusionHack, d.ZSelfOcclusionHack) * n;
                }
                const float relative_y = (((wz * d.ZToYMultiplier) * d.ViewportScale[0]) / d.RenderScale[0]) + bias.y;
                float ex = p.enc_x, ey = p.enc_y;
                if (((flat >> kFlatNormalBit) & 1) == 0)
                    encode_normal(n, &ex, &ey);
                out = encode_sample(ex, ey, relative_y, wz + bias.z, false, at(6) > 0.5f);
                // DepthFormat.Depth24, CompareFunction.GreaterEqual with writes (LightingRenderer.cs:539-551)
                const uint32_t d24 = (uint32_t)floor((double)z * 16777215.0 + 0.5);
                if (!(d24 >= depth)) continue;
                depth = d24;
            } else {
                const float4 data = sample_point(a.textures, __builtin_amdgcn_readlane(v2k.w, from), at(6), at(7));
                const float data_scale = at(9);
                const f3 wp = mk3(at(0), at(1), at(2));
                const f3 n = mk3(at(3), at(4), at(5));
                if (kind == kMask) {                                 // MaskBillboardPixelShader, GBufferBitmap.fx:29-59
                    const float discard_threshold = ref::kMaskDiscardNumerator / 255.0f;
                    if ((data.w - discard_threshold) < 0.0f) continue;
                    const float relative_y = (wp.y - at(8)) * data_scale;
                    out = mk4((n.x / 2.0f) + 0.5f, (n.z / 2.0f) + 0.5f, relative_y,
                              ((wp.z + ref::kGBufferZOffset) / ref::kGBufferZScale) * at(10));
                } else if (kind == kGData) {                         // GDataBillboardPixelShader, GBufferBitmap.fx:61-113
                    const float discard_threshold = ref::kGDataDiscardNumerator / 255.0f;
                    if (data.w < discard_threshold) continue;
                    const float tx = (data.x - 0.5f) * 2.0f, ty = (data.y - 0.5f) * 2.0f;
                    const float tz = sqrtf(1.0f - (tx * tx + ty * ty));
                    const f3 world_normal = mk3((1.0f * tx + 0.0f * ty) + 0.0f * tz, (0.0f * tx + -1.0f * ty) + 0.0f * tz, (0.0f * tx + 0.0f * ty) + 1.0f * tz);
                    const f3 result_normal = norm3(world_normal);
                    const float effective_z = wp.z + (data.z * data_scale);
                    float ex, ey;
                    encode_normal(result_normal, &ex, &ey);
                    out = encode_sample(ex, ey, effective_z * d.ZToYMultiplier, effective_z, false, true);
                } else {
                    continue;                                        // a degenerate triangle's record (empty bounds: not reached)
                }
            }
            texel = out;
        }
    }
    if (i >= a.width || j >= a.height) return;
    const size_t o = (size_t)j * (size_t)a.width + (size_t)i;
    if (a.format == ILM_GBUFFER_HALF4) {
        uint2 h;
        h.x = (uint32_t)__half_as_ushort(__float2half_rn(texel.x)) | ((uint32_t)__half_as_ushort(__float2half_rn(texel.y)) << 16);
        h.y = (uint32_t)__half_as_ushort(__float2half_rn(texel.z)) | ((uint32_t)__half_as_ushort(__float2half_rn(texel.w)) << 16);
        reinterpret_cast<uint2*>(a.texels)[o] = h;
    } else {
        reinterpret_cast<float4*>(a.texels)[o] = texel;
    }
}

hipError_t launch_gbuffer_meshes(const GBufferMeshLaunch& a, hipStream_t stream) {
    if (a.width <= 0 || a.height <= 0) return hipSuccess;
    hipLaunchKernelGGL(gbuffer_setup_kernel, dim3((unsigned)((a.prim_count + 63) / 64)), dim3(64), 0, stream, a);
    hipLaunchKernelGGL(gbuffer_bin_kernel, dim3((unsigned)(a.block_cols * a.block_rows)), dim3(256), 0, stream, a);
    const dim3 grid((unsigned)((a.width + 15) / 16), (unsigned)((a.height + 15) / 16)), block(256);
    hipLaunchKernelGGL(gbuffer_meshes_kernel, grid, block, 0, stream, a);
    return hipGetLastError();
}

}  // namespace ilm
