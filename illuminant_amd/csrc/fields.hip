// fields.hip -- distance-field atlas generation for gfx950.
//
// The reference rasterises one instanced quad per obstruction and slice triplet into the atlas
// render target and lets the ROP MAX-blend the encoded distances
// (Illuminant/Lighting/LightingRenderer.DistanceField.cs:80-152,347-400, techniques of
// Illuminant/Shaders/DistanceFunction.fx:33-155 and Illuminant/Shaders/DistanceField.fx:101-115,
// BlendFunction.Max in Illuminant/LoadMaterials.cs:164-176): every obstruction re-reads and re-writes
// the 8-byte texel.  Here a 32x8-texel tile of one physical slice is one workgroup (four 8x8 waves): wave 0 bins the
// obstruction quads / height-volume boxes that touch the tile into an LDS list (wave64 ballot + popcount), the
// workgroup orders the list nearest-first, every lane walks it with the record in SGPRs (readfirstlane index ->
// scalar loads), keeps the four running maxima of its texel in registers and stores the texel once: the atlas is
// written exactly once per pass (8 B / texel).  An obstruction whose provable lower distance bound cannot raise
// any covered texel of the wave is not evaluated at all (only the nearest surface survives a MAX blend).
//
// Compiled with -ffp-contract=off: every operation rounds as in the CPU oracle, so the stored codes are
// bit-identical to it.  No MFMA (per-texel scalar distance functions), HBM-write-bound by definition
// (8 B per texel) and in practice ALU-bound on scenes with many overlapping obstructions.
#include "internal.hpp"
#include "distance_functions.hpp"

namespace ilm {

// A workgroup covers 32 x 8 texels, each of its four waves an 8 x 8 square (lane = 8 * row + column): a compact footprint keeps the
// per-texel quad-coverage tests and the culling decisions nearly wave-uniform (a 64 x 1 strip measured 47-60 % active lanes).
constexpr int kFieldTileW = 32, kFieldTileH = 8;
constexpr int kFieldListCapacity = 2048;
constexpr int kFieldSortCapacity = 512;      // lists up to this length are ordered nearest-first

// evaluate* by LightObstructionType (LightObstruction.cs:10-16): Ellipsoid, Box, Cylinder, Spheroid, Octagon
// are cases 1..5 of evaluateByTypeId (DistanceFunctionCommon.fxh:170-187)
ILM_DEV float evaluate_obstruction(int type, f3 wp, const FieldObstruction& o) {
    const f3 local = wp - mk3(o.cx, o.cy, o.cz);
    // identity orientation (uniform flag set by the host): rotateLocalPosition returns its input up to the sign of zero components
    const f3 p = ((o._pad & 1) != 0) ? local : rotate_local_q(local, mk4(o.qx, o.qy, o.qz, o.qw));
    return evaluate_shape(type + 1, p, mk3(o.sx, o.sy, o.sz));
}

// computeDistanceZ, DistanceField.fx:47-56
ILM_DEV float compute_distance_z(float slice_z, float z0, float z1) {
    if (slice_z >= z0) {
        if (slice_z <= z1)
            return fmaxf(slice_z - z1, z0 - slice_z);
        return slice_z - z1;
    }
    return z0 - slice_z;
}

// finalEval, DistanceField.fx:58-73 (PolygonXyBias 1.5)
ILM_DEV float final_eval(float z, float z0, float z1, float dist_xy_biased) {
    const float distance_z = compute_distance_z(z, z0, z1);
    if (dist_xy_biased <= 0.0f)
        return (distance_z <= 0.0f) ? dist_xy_biased + distance_z : distance_z;
    return fmaxf(dist_xy_biased, 0.0f) + fmaxf(distance_z, 0.0f);
}

// render-target write of one channel: saturate, then D3D float -> unorm16 (c * 65535 + 0.5, truncated) or the
// IEEE half of the saturated value
template <int FORMAT>
ILM_DEV uint32_t store_channel(float enc) {
    const float c = sat(enc);
    if (FORMAT == ILM_SDF_FP16)
        return (uint32_t)__half_as_ushort(__float2half_rn(c));
    return (uint32_t)floorf(c * 65535.0f + 0.5f);
}

#ifdef ILM_FIELD_TRACE     // EXPERIMENT (tools/field_trace_probe.py): per-workgroup start / end of the last launch (100 MHz clock), list length
__device__ unsigned long long g_field_trace[4 * 65536];
extern "C" int ilm_experiment_field_trace(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_field_trace), sizeof(unsigned long long) * (size_t)n);
}
#endif
#ifndef ILM_FIELD_WAVES
#define ILM_FIELD_WAVES 0
#endif
#if ILM_FIELD_WAVES > 0
#define ILM_FIELD_OCCUPANCY __attribute__((amdgpu_waves_per_eu(ILM_FIELD_WAVES, ILM_FIELD_WAVES)))
#else
#define ILM_FIELD_OCCUPANCY
#endif
template <int FORMAT>
__global__ __launch_bounds__(256) ILM_FIELD_OCCUPANCY void render_slices_kernel(const FieldLaunch a) {
#ifdef ILM_FIELD_TRACE
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();
    int trace_n = 0, trace_evaluated = 0;
#endif
    __shared__ uint16_t list[kFieldListCapacity];
    __shared__ float sort_key[kFieldSortCapacity];
    __shared__ uint16_t sort_index[kFieldSortCapacity];
    __shared__ int list_count;

    const int tiles_x = (a.slice_w + kFieldTileW - 1) / kFieldTileW;
    const int tiles_y = (a.slice_h + kFieldTileH - 1) / kFieldTileH;
    const int tiles_per_slice = tiles_x * tiles_y;
    const int b = (int)blockIdx.x;
    const int triplet = b / tiles_per_slice, tile = b - triplet * tiles_per_slice;
    const int first = a.first_slices[triplet];
    const int physical = first / 3;
    const int col = physical % a.columns, row = physical / a.columns;
    const int slice_x = col * a.slice_w, slice_y = row * a.slice_h;
    const float vpx = -(float)(col * a.virtual_w), vpy = -(float)(row * a.virtual_h);

    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int tx0 = (tile % tiles_x) * kFieldTileW, ty0 = (tile / tiles_x) * kFieldTileH;
    const int i = tx0 + wave * 8 + (lane & 7), j = ty0 + (lane >> 3);
    const bool in_slice = (i < a.slice_w) && (j < a.slice_h);
    const int ax = slice_x + i, ay = slice_y + j;

    // SliceIndexToZ, LightingRenderer.DistanceField.cs:32-35
    float slice_z[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        slice_z[k] = (((float)(first + k) / a.slice_count_f) * a.virtual_depth) + a.z_offset;

    // getPositionXy, DistanceFunction.fx:28-31 (vpos = integer atlas pixel)
    const float wx = ((float)ax * a.inv_scale_x) + vpx;
    const float wy = ((float)ay * a.inv_scale_y) + vpy;
    const float cxp = (float)i + 0.5f, cyp = (float)j + 0.5f;
    const float tminx = (float)tx0 + 0.5f, tmaxx = (float)(tx0 + kFieldTileW - 1) + 0.5f;
    const float tminy = (float)ty0 + 0.5f, tmaxy = (float)(ty0 + kFieldTileH - 1) + 0.5f;

    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;   // saturate() at the target floors every write at 0
    // distance / MaximumEncodedDistance (encodeDistance, DistanceFieldCommon.fxh:264-266): a uniform divisor of ordinary size shares one
    // refined reciprocal (the same correctly rounded quotient); a distance beyond 2^60 takes the IEEE division
    // "Ordinary magnitudes" (wave-uniform): every coordinate this wave hands to a distance function within 2^20, MaximumEncodedDistance in
    // [2^-10, 2^20]; per record (api.hip, flag bit 1): sizes in [2^-10, 2^20], centre within 2^20.  Distances, quotients and residuals of
    // the unscaled division then stay far inside the normal range (no operand needs v_div_scale's rescaling), and its result is the
    // correctly rounded quotient -- `/`.  Anything else takes the IEEE division.
    const bool ordinary_coordinates = (a.max_encoded >= 0x1p-10f) && (a.max_encoded <= 0x1p20f) &&
        (__ballot(!((fabsf(wx) <= 0x1p20f) && (fabsf(wy) <= 0x1p20f) && (fabsf(slice_z[0]) <= 0x1p20f) && (fabsf(slice_z[3]) <= 0x1p20f))) == 0ull);
    const float max_encoded_rcp = refined_rcp(a.max_encoded);

    // ---- analytic obstructions ----------------------------------------------------------------------
    // tile centre in world units (for the nearest-first order of the list)
    const float tcx = (((float)slice_x + (float)tx0 + 0.5f * (float)kFieldTileW) * a.inv_scale_x) + vpx;
    const float tcy = (((float)slice_y + (float)ty0 + 0.5f * (float)kFieldTileH) * a.inv_scale_y) + vpy;
    for (int batch = 0; batch < a.obstruction_count; batch += kFieldListCapacity) {
        const int batch_n = min(kFieldListCapacity, a.obstruction_count - batch);
        __syncthreads();
        if (wave == 0) {
            int base = 0;
            for (int o0 = 0; o0 < batch_n; o0 += 64) {
                const int oi = o0 + lane;
                bool hit = false;
                if (oi < batch_n) {
                    const FieldObstruction& R = a.obstructions[batch + oi];
                    hit = (R.x0 <= tmaxx) && (R.x1 > tminx) && (R.y0 <= tmaxy) && (R.y1 > tminy);
                }
                const unsigned long long m = __ballot(hit);
                if (hit)
                    list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)oi;
                base += __popcll(m);
            }
            if (lane == 0) list_count = base;
        }
        __syncthreads();
        const int n = list_count;
#ifdef ILM_FIELD_TRACE
        trace_n += n;
#endif
        // Nearest first (BlendFunction.Max does not care about the order): the sooner a texel's maximum is high, the more of the
        // remaining obstructions the bound below rejects.  Rank sort by the distance of the tile centre to the bounding sphere,
        // ties by list position; lists longer than the key buffer stay in index order.
        const bool sorted = (n > 1) && (n <= kFieldSortCapacity);
        if (sorted) {
            for (int e = (int)threadIdx.x; e < n; e += 256) {
                const FieldObstruction& R = a.obstructions[batch + list[e]];
                const float dx = R.cx - tcx, dy = R.cy - tcy;
                sort_key[e] = sqrtf(dx * dx + dy * dy) - ((R.cull_radius < 3.0e38f) ? R.cull_radius : 0.0f);
            }
            __syncthreads();
            for (int e = (int)threadIdx.x; e < n; e += 256) {
                const float key = sort_key[e];
                int rank = 0;
                for (int f = 0; f < n; f++) {
                    const float other = sort_key[f];
                    rank += ((other < key) || ((other == key) && (f < e))) ? 1 : 0;
                }
                sort_index[rank] = list[e];
            }
            __syncthreads();
        }
        const uint16_t* order = sorted ? sort_index : list;
        const float z_low = fminf(slice_z[0], slice_z[3]), z_high = fmaxf(slice_z[0], slice_z[3]);
        // The record of list entry k, by value: five 16-byte scalar loads issued together (read field by field behind the raster test's
        // short-circuits, the compiler fetched x0, x1, y0, y1, the centre, the bound and the orientation one dependent s_load after the
        // other -- about ten scalar-cache round trips per obstruction and wave).  The NEXT entry's record is requested before this one is
        // evaluated, so its latency hides behind the distance functions.
        auto load_record = [&](int k) {
            const int oi = __builtin_amdgcn_readfirstlane((int)order[k]);
            return a.obstructions[batch + oi];
        };
        FieldObstruction next_record;
        if (n > 0) next_record = load_record(0);
        for (int k = 0; k < n; k++) {
            const FieldObstruction R = next_record;
            if (k + 1 < n) next_record = load_record(k + 1);
            // DistanceFunctionVertexShader's quad (DistanceFunction.fx:16-26): pixel centre inside [x0, x1) x [y0, y1)
            const bool covered = in_slice & (cxp >= R.x0) & (cxp < R.x1) & (cyp >= R.y0) & (cyp < R.y1);
            // Culling: every slice value of this obstruction is at most kDistanceZero - bound / max_encoded with
            // bound = (e - cull_radius) / cull_inv_scale <= f (internal.hpp); if that cannot exceed the smallest of the four running
            // maxima, MAX leaves the texel as it is.  Evaluate only when some covered lane can still change.
            const float dx = wx - R.cx, dy = wy - R.cy;
            const float dz = fmaxf(fmaxf(z_low - R.cz, R.cz - z_high), 0.0f);      // the four slices lie in [z_low, z_high]
            const float e2 = (dx * dx + dy * dy) + dz * dz;
            const float least = fminf(fminf(acc0, acc1), fminf(acc2, acc3));
            const float reach = fmaxf(((kDistanceZero - least) * a.max_encoded) * R.cull_inv_scale + R.cull_radius, 0.0f);
            const bool can_change = covered && !(e2 >= reach * reach);
            if (__ballot(can_change) == 0ull)
                continue;
#ifdef ILM_FIELD_TRACE
            trace_evaluated++;
#endif
            if (!covered)
                continue;
            const int type = R.type;
            float d[4];
            if ((R._pad & 1) != 0) {
                // unrotated (the common case): the four slices of the texel share x and y -- one evaluation with the xy work done once;
                // bit 1 of the flag: the record's sizes are of ordinary magnitude (api.hip), and so are this wave's coordinates
                const float pz[4] = { slice_z[0] - R.cz, slice_z[1] - R.cz, slice_z[2] - R.cz, slice_z[3] - R.cz };
                if (((R._pad & 2) != 0) && ordinary_coordinates)
                    evaluate_shape4<true>(type + 1, wx - R.cx, wy - R.cy, pz, mk3(R.sx, R.sy, R.sz), d);
                else
                    evaluate_shape4<false>(type + 1, wx - R.cx, wy - R.cy, pz, mk3(R.sx, R.sy, R.sz), d);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) d[k] = evaluate_obstruction(type, mk3(wx, wy, slice_z[k]), R);
            }
            // encodeDistance: distance / MaximumEncodedDistance (DistanceFieldCommon.fxh:264-266), a uniform divisor
            if (((R._pad & 2) != 0) && ordinary_coordinates) {
                acc0 = fmaxf(acc0, kDistanceZero - div_with_rcp(d[0], a.max_encoded, max_encoded_rcp));
                acc1 = fmaxf(acc1, kDistanceZero - div_with_rcp(d[1], a.max_encoded, max_encoded_rcp));
                acc2 = fmaxf(acc2, kDistanceZero - div_with_rcp(d[2], a.max_encoded, max_encoded_rcp));
                acc3 = fmaxf(acc3, kDistanceZero - div_with_rcp(d[3], a.max_encoded, max_encoded_rcp));
            } else {
                acc0 = fmaxf(acc0, kDistanceZero - (d[0] / a.max_encoded));
                acc1 = fmaxf(acc1, kDistanceZero - (d[1] / a.max_encoded));
                acc2 = fmaxf(acc2, kDistanceZero - (d[2] / a.max_encoded));
                acc3 = fmaxf(acc3, kDistanceZero - (d[3] / a.max_encoded));
            }
        }
    }

    // ---- height volumes: DistanceToPolygon, DistanceField.fx:75-115 -------------------------------
    // The tile's list first (wave 0, as for the obstructions): the volumes whose expanded bounds touch the tile AND whose circle is
    // within the encoded reach of some texel of it -- the per-texel culling below at its most generous (running maxima of 0), taken at
    // the tile's centre with the tile's half diagonal and one more unit of slack.  A frame of hundreds of volumes leaves a tile a handful.
    const float tile_half_diagonal = 0.5f * sqrtf(((float)kFieldTileW * a.inv_scale_x) * ((float)kFieldTileW * a.inv_scale_x) +
                                                   ((float)kFieldTileH * a.inv_scale_y) * ((float)kFieldTileH * a.inv_scale_y));
    const float widest_need = kDistanceZero * a.max_encoded + 1.0f;
    const bool volumes_cullable = (widest_need <= 999.0f) && (a.max_encoded <= 65536.0f);
    for (int vbatch = 0; vbatch < a.volume_count; vbatch += kFieldListCapacity) {
    const int vbatch_n = min(kFieldListCapacity, a.volume_count - vbatch);
    __syncthreads();
    if (wave == 0) {
        int base = 0;
        for (int o0 = 0; o0 < vbatch_n; o0 += 64) {
            const int oi = o0 + lane;
            bool hit = false;
            if (oi < vbatch_n) {
                const FieldVolume& V = a.volumes[vbatch + oi];
                hit = (V.x0 <= tmaxx) && (V.x1 > tminx) && (V.y0 <= tmaxy) && (V.y1 > tminy);
                const float ddx = V.cx - tcx, ddy = V.cy - tcy;
                if (volumes_cullable && (sqrtf(ddx * ddx + ddy * ddy) - tile_half_diagonal - 1.0f >= widest_need + V.radius))
                    hit = false;
            }
            const unsigned long long m = __ballot(hit);
            if (hit)
                list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)oi;
            base += __popcll(m);
        }
        if (lane == 0) list_count = base;
    }
    __syncthreads();
    const int vn = list_count;
    for (int vk = 0; vk < vn; vk++) {
        const int v = vbatch + __builtin_amdgcn_readfirstlane((int)list[vk]);
        const FieldVolume& V = a.volumes[v];
        const bool covered = in_slice && (cxp >= V.x0) && (cxp < V.x1) && (cyp >= V.y0) && (cyp < V.y1);
        // Culling (r04; the reference rasterises the polygon's bounds expanded by DistanceLimit = 520 units, LightingRenderer.cs:316, but a
        // texel farther than DISTANCE_ZERO x MaximumEncodedDistance from the volume encodes a value <= 0, which MAX over a target
        // floored at 0 never keeps): outside the circle around the polygon the distance the shader computes is
        //     min(distance to the polygon, sqrt(999999)) + 1.5 + max(dz, 0)  >=  min(|p - c| - radius, 999),
        // so if that bound exceeds (DISTANCE_ZERO - the smallest of the four running maxima) x MaximumEncodedDistance by a unit -- far
        // more than the roundings of the bound and of the encoding can move either side -- MAX leaves the texel as it is for all four
        // slices.  The volume is evaluated only when some covered lane of the wave can still change.
        {
            const float ddx = wx - V.cx, ddy = wy - V.cy;
            const float to_centre = sqrtf(ddx * ddx + ddy * ddy);
            const float least = fminf(fminf(acc0, acc1), fminf(acc2, acc3));
            const float need = fmaxf((kDistanceZero - least) * a.max_encoded + 1.0f, 1.0f);     // (the bound only speaks about texels outside the circle)
            const bool cannot_change = (need <= 999.0f) && (a.max_encoded <= 65536.0f) && (to_centre >= need + V.radius);
            if (__ballot(covered && !cannot_change) == 0ull)
                continue;
        }
        // Inigo Quilez' sdPolygon (Fracture SDF2D.fxh sdPolygonInit / sdPolygonVertex): every edge exactly once
        float dist_sq = 999999.0f, sign = 1.0f;
        const float2* P = a.polygon_xy + V.first_vertex;
        for (int e = 0; e < V.vertex_count; e++) {
            const int nx = (e + 1 == V.vertex_count) ? 0 : e + 1;
            const float2 vi = P[nx], vj = P[e];
            const float ex = vj.x - vi.x, ey = vj.y - vi.y;
            const float qx = wx - vi.x, qy = wy - vi.y;
            const float t = clampf((qx * ex + qy * ey) / (ex * ex + ey * ey), 0.0f, 1.0f);
            const float bx = qx - ex * t, by = qy - ey * t;
            dist_sq = fminf(dist_sq, bx * bx + by * by);
            const bool c0 = wy >= vi.y, c1 = wy < vj.y, c2 = (ex * qy) > (ey * qx);
            if ((c0 && c1 && c2) || (!c0 && !c1 && !c2))
                sign = -sign;
        }
        if (!covered)
            continue;
        const float dxy = (sqrtf(dist_sq) * sign) + 1.5f;
        acc0 = fmaxf(acc0, kDistanceZero - (final_eval(slice_z[0], V.z0, V.z1, dxy) / a.max_encoded));
        acc1 = fmaxf(acc1, kDistanceZero - (final_eval(slice_z[1], V.z0, V.z1, dxy) / a.max_encoded));
        acc2 = fmaxf(acc2, kDistanceZero - (final_eval(slice_z[2], V.z0, V.z1, dxy) / a.max_encoded));
        acc3 = fmaxf(acc3, kDistanceZero - (final_eval(slice_z[3], V.z0, V.z1, dxy) / a.max_encoded));
    }
    }

#ifdef ILM_FIELD_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 65536u) {
        g_field_trace[4 * blockIdx.x] = trace_t0; g_field_trace[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        g_field_trace[4 * blockIdx.x + 2] = (unsigned long long)trace_n; g_field_trace[4 * blockIdx.x + 3] = (unsigned long long)trace_evaluated;
    }
#endif
    if (!in_slice)
        return;
    const size_t o = (size_t)ay * (size_t)a.atlas_w + (size_t)ax;
    uint32_t c0 = store_channel<FORMAT>(acc0), c1 = store_channel<FORMAT>(acc1);
    uint32_t c2 = store_channel<FORMAT>(acc2), c3 = store_channel<FORMAT>(acc3);
    if (a.clear_source != nullptr) {
        // ClearDistanceFieldSlice with the static texture (ClearDistanceField.fx:30-44), then BlendFunction.Max on the
        // stored codes (non-negative halves order like their bit patterns)
        const uint2 s = a.clear_source[o];
        c0 = max(c0, s.x & 0xFFFFu); c1 = max(c1, s.x >> 16);
        c2 = max(c2, s.y & 0xFFFFu); c3 = max(c3, s.y >> 16);
    }
    a.atlas[o] = make_uint2(c0 | (c1 << 16), c2 | (c3 << 16));
}

hipError_t launch_render_slices(const FieldLaunch& a, int format, hipStream_t stream) {
    if (a.triplet_count <= 0 || a.slice_w <= 0 || a.slice_h <= 0) return hipSuccess;
    const int tiles_x = (a.slice_w + kFieldTileW - 1) / kFieldTileW, tiles_y = (a.slice_h + kFieldTileH - 1) / kFieldTileH;
    const dim3 grid((unsigned)(a.triplet_count * tiles_x * tiles_y)), block(256);
    if (format == ILM_SDF_FP16) hipLaunchKernelGGL(render_slices_kernel<ILM_SDF_FP16>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(render_slices_kernel<ILM_SDF_UNORM16>, grid, block, 0, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// G-buffer generation, non-2.5D: RenderGBuffer (LightingRenderer.GBuffer.cs:127-219) = clear + ground plane + height-volume top
// faces, lowest to highest, all with technique GroundPlane (GBuffer.fx:7-19,57-70).  One pass, one store per texel
// (16 B float4 or 8 B half4): a pure HBM-write stream plus an edge loop for pixels inside a volume's bounding box.
// ---------------------------------------------------------------------------------------------
// encodeGBufferSample with normal (0, 0, 1), relativeY 0 (GBufferShaderCommon.fxh:10-35; encodeNormalSpherical, EnvironmentCommon.fxh:33-40)
ILM_DEV float4 encode_gbuffer_up(float z, bool enable_shadows) {
    const float nx = 0.0001f;                                          // |n.x| < 0.0001 => 0.0001
    const float ex = ((atan2f(0.0f, nx) / kPi) + 1.0f) * 0.5f;
    const float ey = (1.0f + 1.0f) * 0.5f;
    const float w = (((z + ref::kGBufferZOffset) / ref::kGBufferZScale) * (enable_shadows ? 1.0f : -1.0f)) + (enable_shadows ? 0.0f : -1.0f);
    return mk4(ex, ey, 0.0f, w);
}

// A workgroup is a 64 x 4 pixel tile.  256 volumes at a time, thread t tests volume t's bounds against the TILE and the hits are
// listed in LDS in the volumes' order (ballot + popcount prefix, the waves' counts through LDS); every pixel then walks the list --
// a handful of volumes -- instead of comparing itself with every volume of the frame (r04: 256 volumes at 1080p 0.177 -> 0.027 ms, tools/gbuffer_polygon_probe.py).
__global__ __launch_bounds__(256) void render_gbuffer_kernel(const GBufferLaunch a) {
    __shared__ uint16_t s_list[256];
    __shared__ int s_count[4];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int i = (int)blockIdx.x * 64 + lane;
    const int j = (int)blockIdx.y * 4 + wave;
    const bool in_image = (i < a.width) && (j < a.height);
    const float wx = ((float)i + 0.5f) / a.desc.ViewportScale[0] + a.desc.ViewportPosition[0];
    const float wy = ((float)j + 0.5f) / a.desc.ViewportScale[1] + a.desc.ViewportPosition[1];
    // the tile's extreme pixel centres in world units, by the same expression (either sign of the scale)
    const float ex0 = ((float)((int)blockIdx.x * 64) + 0.5f) / a.desc.ViewportScale[0] + a.desc.ViewportPosition[0];
    const float ex1 = ((float)((int)blockIdx.x * 64 + 63) + 0.5f) / a.desc.ViewportScale[0] + a.desc.ViewportPosition[0];
    const float ey0 = ((float)((int)blockIdx.y * 4) + 0.5f) / a.desc.ViewportScale[1] + a.desc.ViewportPosition[1];
    const float ey1 = ((float)((int)blockIdx.y * 4 + 3) + 0.5f) / a.desc.ViewportScale[1] + a.desc.ViewportPosition[1];
    const float tx0 = fminf(ex0, ex1), tx1 = fmaxf(ex0, ex1), ty0 = fminf(ey0, ey1), ty1 = fmaxf(ey0, ey1);
    // (a NaN anywhere makes the tile test unreliable: every volume is listed then, and the per-pixel test decides as before)
    const bool tile_ordered = (tx0 <= tx1) && (ty0 <= ty1);
    const float ground_z = a.desc.GroundZ + (a.desc.RenderGroundPlane ? 0.0f : ref::kGroundLift);
    float4 texel = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    if (!(ground_z < a.desc.GroundZ))
        texel = encode_gbuffer_up(ground_z, a.desc.EnableGroundShadows != 0);
    for (int v0 = 0; v0 < a.volume_count; v0 += 256) {
        const int mine = v0 + (int)threadIdx.x;
        bool hit = false;
        if (mine < a.volume_count) {
            const GBufferVolume& V = a.volumes[mine];
            hit = !tile_ordered || !((V.x1 < tx0) || (V.x0 > tx1) || (V.y1 < ty0) || (V.y0 > ty1));
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) s_count[wave] = __popcll(m);
        __syncthreads();
        int before = 0, n = 0;
        for (int w = 0; w < 4; w++) { const int c = s_count[w]; n += c; if (w < wave) before += c; }
        if (hit) s_list[before + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)threadIdx.x;
        __syncthreads();
        for (int k = 0; k < n; k++) {
            const int v = v0 + (int)s_list[k];
            const GBufferVolume& V = a.volumes[v];
            if (!((wx >= V.x0) && (wx <= V.x1) && (wy >= V.y0) && (wy <= V.y1)))      // outside the polygon's bounds: no crossing can make it inside
                continue;
            const float2* P = a.polygon_xy + V.first_vertex;
            bool inside = false;
            for (int e = 0; e < V.vertex_count; e++) {
                const int nx = (e + 1 == V.vertex_count) ? 0 : e + 1;
                const float2 pa = P[e], pb = P[nx];
                if ((pa.y > wy) != (pb.y > wy)) {
                    const float xi = ((pb.x - pa.x) * (wy - pa.y)) / (pb.y - pa.y) + pa.x;
                    if (wx < xi) inside = !inside;
                }
            }
            if (!inside || (V.top < a.desc.GroundZ))
                continue;
            texel = encode_gbuffer_up(V.top, V.enable_shadows != 0);
        }
        __syncthreads();                                        // the list is rewritten by the next batch
    }
    if (!in_image) return;
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

hipError_t launch_render_gbuffer(const GBufferLaunch& a, hipStream_t stream) {
    if (a.width <= 0 || a.height <= 0) return hipSuccess;
    const dim3 grid((unsigned)((a.width + 63) / 64), (unsigned)((a.height + 3) / 4)), block(256);
    hipLaunchKernelGGL(render_gbuffer_kernel, grid, block, 0, stream, a);
    return hipGetLastError();
}

}  // namespace ilm
