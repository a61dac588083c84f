// distance_functions.hpp -- the analytic distance functions of
// Illuminant/Shaders/DistanceFunctionCommon.fxh:7-187, shared by the particle area transforms
// (FMA / Noise weights, particles.hip) and the distance-field generation pass (fields.hip).
#pragma once

#include "hlsl_math.hpp"

namespace ilm {

// qmul, DistanceFunctionCommon.fxh:16-21
ILM_DEV float4 qmul(float4 q1, float4 q2) {
    const f3 a = xyz(q2) * q1.w, b = xyz(q1) * q2.w, c = cross3(xyz(q1), xyz(q2));
    const f3 s = (a + b) + c;
    return mk4(s.x, s.y, s.z, q1.w * q2.w - dot3(xyz(q1), xyz(q2)));
}
// rotateLocalPosition, :24-27
ILM_DEV f3 rotate_local_q(f3 p, float4 rot) {
    const float4 r_c = mk4(rot.x * -1.0f, rot.y * -1.0f, rot.z * -1.0f, rot.w * 1.0f);
    return xyz(qmul(rot, qmul(mk4(p.x, p.y, p.z, 0.0f), r_c)));
}
// the particle transforms pass the scalar AreaRotation, which HLSL promotes to float4(r,r,r,r) (FMA.fx:11,16-18)
ILM_DEV f3 rotate_local(f3 p, float r) { return rotate_local_q(p, mk4(r, r, r, r)); }

// opElongate, :43-46
ILM_DEV float4 op_elongate(f3 p, f3 h) {
    const f3 q = abs3(p) - h;
    const f3 m = max03(q);
    return mk4(sgn(p.x) * m.x, sgn(p.y) * m.y, sgn(p.z) * m.z, fminf(fmaxf(q.x, fmaxf(q.y, q.z)), 0.0f));
}
// sdOctogonPrism, :141-156
ILM_DEV float sd_octogon_prism(f3 p, float r, float h) {
    const float kx = -0.9238795325f, ky = 0.3826834323f, kz = 0.4142135623f;
    p = abs3(p);
    const float d1 = fminf(kx * p.x + ky * p.y, 0.0f);
    p.x -= 2.0f * d1 * kx;
    p.y -= 2.0f * d1 * ky;
    const float d2 = fminf(-kx * p.x + ky * p.y, 0.0f);
    p.x -= 2.0f * d2 * -kx;
    p.y -= 2.0f * d2 * ky;
    p.x -= clampf(p.x, -kz * r, kz * r);
    p.y -= r;
    const float dx = sqrtf(p.x * p.x + p.y * p.y) * sgn(p.y);
    const float dy = p.z - h;
    const float mx = fmaxf(dx, 0.0f), my = fmaxf(dy, 0.0f);
    return fminf(fmaxf(dx, dy), 0.0f) + sqrtf(mx * mx + my * my);
}

// evaluateByTypeId's cases (:170-187) on a position already moved into the shape's frame
// (p = rotateLocalPosition(worldPosition - center, rotation)); shape 1..5 as in the switch.
ILM_DEV float evaluate_shape(int shape, f3 p, f3 size) {
    switch (shape) {
        case 1: {  // evaluateEllipsoid -> sdEllipsoid_improvedV2, :92-109
            const float k0 = len3(mk3(p.x / size.x, p.y / size.y, p.z / size.z));
            const float k1 = len3(mk3(p.x / (size.x * size.x), p.y / (size.y * size.y), p.z / (size.z * size.z)));
            return (k0 < 1.0f) ? (k0 - 1.0f) * fminf(fminf(size.x, size.y), size.z) : k0 * (k0 - 1.0f) / k1;
        }
        case 2: {  // evaluateBox, :48-63
            const f3 d = abs3(p) - size;
            return fminf(fmaxf(d.x, fmaxf(d.y, d.z)), 0.0f) + len3(max03(d));
        }
        case 3: {  // evaluateCylinder -> sdCappedCylinder, :111-124
            const float h = size.z, r = sqrtf(size.x * size.x + size.y * size.y);
            const float dx = fabsf(sqrtf(p.x * p.x + p.y * p.y)) - r;
            const float dy = fabsf(p.z) - h;
            const float mx = fmaxf(dx, 0.0f), my = fmaxf(dy, 0.0f);
            return fminf(fmaxf(dx, dy), 0.0f) + sqrtf(mx * mx + my * my);
        }
        case 4: {  // evaluateSpheroid, :65-75
            const float min_size = fminf(size.x, fminf(size.y, size.z));
            const float4 w = op_elongate(p, mk3(size.x - min_size, size.y - min_size, size.z - min_size));
            return w.w + (len3(xyz(w)) - min_size);
        }
        default: {  // evaluateOctagon, :158-168
            const float min_size = fminf(size.x, size.y);
            const float4 w = op_elongate(p, mk3(size.x - min_size, size.y - min_size, 0.0f));
            return w.w + sd_octogon_prism(xyz(w), min_size, size.z);
        }
    }
}

// The same functions for FOUR points that differ in z only -- the four virtual slices of one atlas texel (fields.hip): whatever depends
// on x and y alone is formed once.  Every value is produced by the same operations in the same order as in evaluate_shape (a sum
// (x^2 + y^2) + z^2 keeps its inner sum, a maximum max(x, max(y, z)) is regrouped only where max is exact anyway), so the four results
// are bit-identical to four calls of evaluate_shape; what is saved are the repeats: an ellipsoid's four divisions by size.xy and
// size.xy^2, a cylinder's two square roots, the xy half of every norm.
// ORDINARY: the caller has checked that sizes, their squares and the coordinates lie in the operand range of the unscaled division
// (hlsl_math.hpp div_with_rcp: 2^-60 <= |d| <= 2^60, |n| <= 2^60): a division by a per-obstruction constant then shares one refined
// reciprocal -- the same correctly rounded quotient as `/` for 6 instead of ~10 instructions.
template <bool ORDINARY>
ILM_DEV void evaluate_shape4(int shape, float px, float py, const float pz[4], f3 size, float out[4]) {
    switch (shape) {
        case 1: {  // sdEllipsoid_improvedV2
            const float sxx = size.x * size.x, syy = size.y * size.y, szz = size.z * size.z;
            float ax, ay, bx, by, rz = 0.0f, rzz = 0.0f;
            if (ORDINARY) {
                ax = div_with_rcp(px, size.x, refined_rcp(size.x)); ay = div_with_rcp(py, size.y, refined_rcp(size.y));
                bx = div_with_rcp(px, sxx, refined_rcp(sxx)); by = div_with_rcp(py, syy, refined_rcp(syy));
                rz = refined_rcp(size.z); rzz = refined_rcp(szz);
            } else {
                ax = px / size.x; ay = py / size.y;
                bx = px / sxx; by = py / syy;
            }
            const float a2 = ax * ax + ay * ay, b2 = bx * bx + by * by;
            const float min_size = fminf(fminf(size.x, size.y), size.z);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float az = ORDINARY ? div_with_rcp(pz[k], size.z, rz) : pz[k] / size.z;
                const float bz = ORDINARY ? div_with_rcp(pz[k], szz, rzz) : pz[k] / szz;
                const float k0 = sqrtf(a2 + az * az);
                const float k1 = sqrtf(b2 + bz * bz);
                out[k] = (k0 < 1.0f) ? (k0 - 1.0f) * min_size : k0 * (k0 - 1.0f) / k1;
            }
            return;
        }
        case 2: {  // evaluateBox
            const float dx = fabsf(px) - size.x, dy = fabsf(py) - size.y;
            const float mx = fmaxf(dx, 0.0f), my = fmaxf(dy, 0.0f);
            const float m2 = mx * mx + my * my;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float dz = fabsf(pz[k]) - size.z;
                const float mz = fmaxf(dz, 0.0f);
                out[k] = fminf(fmaxf(dx, fmaxf(dy, dz)), 0.0f) + sqrtf(m2 + mz * mz);
            }
            return;
        }
        case 3: {  // sdCappedCylinder
            const float h = size.z, r = sqrtf(size.x * size.x + size.y * size.y);
            const float dx = fabsf(sqrtf(px * px + py * py)) - r;
            const float mx = fmaxf(dx, 0.0f);
            const float mx2 = mx * mx;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float dy = fabsf(pz[k]) - h;
                const float my = fmaxf(dy, 0.0f);
                out[k] = fminf(fmaxf(dx, dy), 0.0f) + sqrtf(mx2 + my * my);
            }
            return;
        }
        case 4: {  // evaluateSpheroid: opElongate per axis, then a sphere
            const float min_size = fminf(size.x, fminf(size.y, size.z));
            const float hx = size.x - min_size, hy = size.y - min_size, hz = size.z - min_size;
            const float qx = fabsf(px) - hx, qy = fabsf(py) - hy;
            const float wx = sgn(px) * fmaxf(qx, 0.0f), wy = sgn(py) * fmaxf(qy, 0.0f);
            const float w2 = wx * wx + wy * wy;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float qz = fabsf(pz[k]) - hz;
                const float wz = sgn(pz[k]) * fmaxf(qz, 0.0f);
                const float ww = fminf(fmaxf(qx, fmaxf(qy, qz)), 0.0f);
                out[k] = ww + (sqrtf(w2 + wz * wz) - min_size);
            }
            return;
        }
        default: {  // evaluateOctagon: elongated in x and y only, so the prism's xy half is shared too
            const float min_size = fminf(size.x, size.y);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float4 w = op_elongate(mk3(px, py, pz[k]), mk3(size.x - min_size, size.y - min_size, 0.0f));
                out[k] = w.w + sd_octogon_prism(xyz(w), min_size, size.z);
            }
            return;
        }
    }
}

}  // namespace ilm