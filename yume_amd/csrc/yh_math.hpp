// yh_math.hpp — register-resident Mat4 math for the gfx950 kernels (and the scalar host helpers).
//
// Mirrors the scalar semantics of the reference's engine/math3d.zig: column-major 16 x f32,
// values[c*4+r]; every expression is evaluated unfused and left to right (the translation unit is
// built with -ffp-contract=off), so results are the same IEEE values the reference's CPU systems
// produce.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define YH_HD __host__ __device__ __forceinline__

struct M4 {
    float v[16];
};

// engine/math3d.zig:583-610 — full 4x4 product; element (c,r) = A.i[r]*B.c.x + A.j[r]*B.c.y +
// A.k[r]*B.c.z + A.t[r]*B.c.w, left to right.  Affine structure is deliberately not exploited.
YH_HD M4 m4_mul(const M4& a, const M4& b) {
    M4 r;
    // (two rows of a column per packed instruction - v_pk_mul_f32 / v_pk_add_f32, the same IEEE results - saves 18 % of the
    // vector instructions of k_frame_chain<3> and costs 17 VGPRs: spills or an occupancy step, 1.5 % slower either way;
    // profiles/r02/ab_pk_mul.txt.  The kernels wait for memory, not for the vector unit.)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float bx = b.v[c * 4 + 0], by = b.v[c * 4 + 1], bz = b.v[c * 4 + 2], bw = b.v[c * 4 + 3];
#pragma unroll
        for (int row = 0; row < 4; ++row) {
            r.v[c * 4 + row] = a.v[0 + row] * bx + a.v[4 + row] * by + a.v[8 + row] * bz + a.v[12 + row] * bw;
        }
    }
    return r;
}

YH_HD M4 m4_identity() {
    M4 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r.v[i] = 0.0f;
    r.v[0] = r.v[5] = r.v[10] = r.v[15] = 1.0f;
    return r;
}

// engine/math3d.zig:727-749 (fromQuat; quaternion used as given)
YH_HD M4 m4_from_quat(float x, float y, float z, float w) {
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, xz = x * z, yz = y * z;
    const float wx = w * x, wy = w * y, wz = w * z;
    M4 r = m4_identity();
    r.v[0] = 1.0f - 2.0f * (yy + zz);
    r.v[1] = 2.0f * (xy + wz);
    r.v[2] = 2.0f * (xz - wy);
    r.v[4] = 2.0f * (xy - wz);
    r.v[5] = 1.0f - 2.0f * (xx + zz);
    r.v[6] = 2.0f * (yz + wx);
    r.v[8] = 2.0f * (xz + wy);
    r.v[9] = 2.0f * (yz - wx);
    r.v[10] = 1.0f - 2.0f * (xx + yy);
    return r;
}

// engine/math3d.zig:816-822 evaluated literally: translation.mul(rotation).mul(scale).
YH_HD M4 m4_compose_literal(float tx, float ty, float tz, float qx, float qy, float qz, float qw, float sx, float sy,
                            float sz) {
    M4 t = m4_identity();
    t.v[12] = tx;
    t.v[13] = ty;
    t.v[14] = tz;
    M4 s = m4_identity();
    s.v[0] = sx;
    s.v[5] = sy;
    s.v[10] = sz;
    const M4 r = m4_from_quat(qx, qy, qz, qw);
    return m4_mul(m4_mul(t, r), s);
}

// Batched Mat4.compose.  The two products of compose only ever add exact zeros to one surviving term
// per element, so for finite operands the result equals  [R.i*s.x | R.j*s.y | R.k*s.z | (t,1)]  value for
// value (only the sign of a zero can differ).  A non-finite intermediate anywhere shows up as a
// non-finite element of that closed form (0*inf = NaN in the literal products), so those lanes take the
// literal two-product path and produce the reference's NaN pattern exactly.
YH_HD M4 m4_compose(float tx, float ty, float tz, float qx, float qy, float qz, float qw, float sx, float sy,
                    float sz) {
    M4 r = m4_from_quat(qx, qy, qz, qw);
    r.v[0] = r.v[0] * sx;
    r.v[1] = r.v[1] * sx;
    r.v[2] = r.v[2] * sx;
    r.v[4] = r.v[4] * sy;
    r.v[5] = r.v[5] * sy;
    r.v[6] = r.v[6] * sy;
    r.v[8] = r.v[8] * sz;
    r.v[9] = r.v[9] * sz;
    r.v[10] = r.v[10] * sz;
    r.v[12] = tx;
    r.v[13] = ty;
    r.v[14] = tz;
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 15; ++i) {
        if ((i & 3) != 3) finite = finite && (__builtin_fabsf(r.v[i]) < __builtin_inff());
    }
    if (!finite) r = m4_compose_literal(tx, ty, tz, qx, qy, qz, qw, sx, sy, sz);
    return r;
}

// engine/math3d.zig:763-814 — cofactor inverse; returns false iff det == 0 exactly.
YH_HD bool m4_inverse(const M4& m, M4& out) {
    const float n11 = m.v[0], n21 = m.v[1], n31 = m.v[2], n41 = m.v[3];
    const float n12 = m.v[4], n22 = m.v[5], n32 = m.v[6], n42 = m.v[7];
    const float n13 = m.v[8], n23 = m.v[9], n33 = m.v[10], n43 = m.v[11];
    const float n14 = m.v[12], n24 = m.v[13], n34 = m.v[14], n44 = m.v[15];

    const float t11 = n23 * n34 * n42 - n24 * n33 * n42 + n24 * n32 * n43 - n22 * n34 * n43 - n23 * n32 * n44 + n22 * n33 * n44;
    const float t12 = n14 * n33 * n42 - n13 * n34 * n42 - n14 * n32 * n43 + n12 * n34 * n43 + n13 * n32 * n44 - n12 * n33 * n44;
    const float t13 = n13 * n24 * n42 - n14 * n23 * n42 + n14 * n22 * n43 - n12 * n24 * n43 - n13 * n22 * n44 + n12 * n23 * n44;
    const float t14 = n14 * n23 * n32 - n13 * n24 * n32 - n14 * n22 * n33 + n12 * n24 * n33 + n13 * n22 * n34 - n12 * n23 * n34;

    const float det = n11 * t11 + n21 * t12 + n31 * t13 + n41 * t14;
    if (det == 0.0f) return false;
    const float di = 1.0f / det;

    out.v[0] = t11 * di;
    out.v[1] = (n24 * n33 * n41 - n23 * n34 * n41 - n24 * n31 * n43 + n21 * n34 * n43 + n23 * n31 * n44 - n21 * n33 * n44) * di;
    out.v[2] = (n22 * n34 * n41 - n24 * n32 * n41 + n24 * n31 * n42 - n21 * n34 * n42 - n22 * n31 * n44 + n21 * n32 * n44) * di;
    out.v[3] = (n23 * n32 * n41 - n22 * n33 * n41 - n23 * n31 * n42 + n21 * n33 * n42 + n22 * n31 * n43 - n21 * n32 * n43) * di;
    out.v[4] = t12 * di;
    out.v[5] = (n13 * n34 * n41 - n14 * n33 * n41 + n14 * n31 * n43 - n11 * n34 * n43 - n13 * n31 * n44 + n11 * n33 * n44) * di;
    out.v[6] = (n14 * n32 * n41 - n12 * n34 * n41 - n14 * n31 * n42 + n11 * n34 * n42 + n12 * n31 * n44 - n11 * n32 * n44) * di;
    out.v[7] = (n12 * n33 * n41 - n13 * n32 * n41 + n13 * n31 * n42 - n11 * n33 * n42 - n12 * n31 * n43 + n11 * n32 * n43) * di;
    out.v[8] = t13 * di;
    out.v[9] = (n14 * n23 * n41 - n13 * n24 * n41 - n14 * n21 * n43 + n11 * n24 * n43 + n13 * n21 * n44 - n11 * n23 * n44) * di;
    out.v[10] = (n12 * n24 * n41 - n14 * n22 * n41 + n14 * n21 * n42 - n11 * n24 * n42 - n12 * n21 * n44 + n11 * n22 * n44) * di;
    out.v[11] = (n13 * n22 * n41 - n12 * n23 * n41 - n13 * n21 * n42 + n11 * n23 * n42 + n12 * n21 * n43 - n11 * n22 * n43) * di;
    out.v[12] = t14 * di;
    out.v[13] = (n13 * n24 * n31 - n14 * n23 * n31 + n14 * n21 * n33 - n11 * n24 * n33 - n13 * n21 * n34 + n11 * n23 * n34) * di;
    out.v[14] = (n14 * n22 * n31 - n12 * n24 * n31 - n14 * n21 * n32 + n11 * n24 * n32 + n12 * n21 * n34 - n11 * n22 * n34) * di;
    out.v[15] = (n12 * n23 * n31 - n13 * n22 * n31 + n13 * n21 * n32 - n11 * n23 * n32 - n12 * n21 * n33 + n11 * n22 * n33) * di;
    return true;
}

// engine/math3d.zig:828-981 — Mat4.decompose (determinant3x3 :984-988, Vec3 helpers :80-156, epsilonEql :1113-1119).
// unnamed[i][j] = values[i*4+j]: the function's "rows" are the matrix columns.  The perspective vector it solves for
// is never returned; only that branch's side effect (the perspective partition is cleared) is kept.  Every operation
// is a correctly rounded IEEE add / mul / div / sqrt (see the Makefile flags), in the reference's order.  Returns
// false where the reference returns error.InvalidMatrix.  out: t.xyz, q.xyzw, s.xyz.
YH_HD bool m4_decompose(const M4& in, float (&out)[10]) {
    const float eps = 1.1920928955078125e-07f;
    auto eql0 = [&](float a) { return __builtin_fabsf(a - 0.0f) <= eps; };     // epsilonEql(a, 0), math3d.zig:1113-1115
    auto not_eql0 = [&](float a) { return __builtin_fabsf(a - 0.0f) > eps; };  // epsilonNotEql(a, 0), :1117-1119 (false for NaN)
    auto len3 = [](const float* v) { return __builtin_sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    auto dot3 = [](const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    auto normalize3 = [&](float* v) {
        const float l = len3(v);
        if (l > 0.0f) {
            v[0] = v[0] / l;
            v[1] = v[1] / l;
            v[2] = v[2] / l;
        } else {
            v[0] = v[1] = v[2] = 0.0f;
        }
    };
    auto combine3 = [](float* v, const float* o, float sv, float so) {
        v[0] = v[0] * sv + o[0] * so;
        v[1] = v[1] * sv + o[1] * so;
        v[2] = v[2] * sv + o[2] * so;
    };
    float m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = in.v[i];
    if (eql0(m[15])) return false;
    const float w = m[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) m[i] = m[i] / w;
    m[15] = m[15] / m[15];
    const float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
    if (eql0(det)) return false;
    if (not_eql0(m[3]) || not_eql0(m[7]) || not_eql0(m[11])) {  // math3d.zig:854-856
        m[3] = m[7] = m[11] = 0.0f;
        m[15] = 1.0f;
    }
    float r0[3] = {m[0], m[1], m[2]}, r1[3] = {m[4], m[5], m[6]}, r2[3] = {m[8], m[9], m[10]};
    float sx = len3(r0);
    normalize3(r0);
    const float skew_z = dot3(r0, r1);
    combine3(r1, r0, 1.0f, -skew_z);
    float sy = len3(r1);
    if (eql0(sy)) return false;
    normalize3(r1);
    const float skew_y = dot3(r0, r2);
    combine3(r2, r0, 1.0f, -skew_y);
    const float skew_x = dot3(r1, r2);
    combine3(r2, r1, 1.0f, -skew_x);
    float sz = len3(r2);
    if (eql0(sz)) return false;
    normalize3(r2);
    const float p[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    if (dot3(r0, p) < 0.0f) {
        sx *= -1.0f;
        sy *= -1.0f;
        sz *= -1.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            r0[c] = r0[c] * -1.0f;
            r1[c] = r1[c] * -1.0f;
            r2[c] = r2[c] * -1.0f;
        }
    }
    float qx, qy, qz, qw;
    const float trace = r0[0] + r1[1] + r2[2];
    if (trace > 0.0f) {
        float root = __builtin_sqrtf(trace + 1.0f);
        qw = 0.5f * root;
        root = 0.5f / root;
        qx = root * (r1[2] - r2[1]);
        qy = root * (r2[0] - r0[2]);
        qz = root * (r0[1] - r1[0]);
    } else {
        int i = 0;
        if (r1[1] > r0[0]) i = 1;
        if (r2[2] > (i == 0 ? r0[0] : r1[1])) i = 2;
        if (i == 0) {
            float root = __builtin_sqrtf(r0[0] - r1[1] - r2[2] + 1.0f);
            qx = 0.5f * root;
            root = 0.5f / root;
            qy = root * (r0[1] + r1[0]);
            qz = root * (r0[2] + r2[0]);
            qw = root * (r1[2] - r2[1]);
        } else if (i == 1) {
            float root = __builtin_sqrtf(r1[1] - r2[2] - r0[0] + 1.0f);
            qy = 0.5f * root;
            root = 0.5f / root;
            qx = root * (r0[1] + r1[0]);
            qz = root * (r1[2] + r2[1]);
            qw = root * (r2[0] - r0[2]);
        } else {
            float root = __builtin_sqrtf(r2[2] - r0[0] - r1[1] + 1.0f);
            qz = 0.5f * root;
            root = 0.5f / root;
            qx = root * (r0[2] + r2[0]);
            qy = root * (r1[2] + r2[1]);
            qw = root * (r0[1] - r1[0]);
        }
    }
    out[0] = m[12];
    out[1] = m[13];
    out[2] = m[14];
    out[3] = qx;
    out[4] = qy;
    out[5] = qz;
    out[6] = qw;
    out[7] = sx;
    out[8] = sy;
    out[9] = sz;
    return true;
}

// ---- culling (build-defined) -------------------------------------------------------
// Box centre c = (min+max)*0.5, half extents h = (max-min)*0.5, world centre cw = M*(c,1) in mulVec4 order
// (engine/math3d.zig:612-620).  Per plane (n,d): s = n.cw + d, r = |n.M.i|*h.x + |n.M.j|*h.y + |n.M.k|*h.z.
// visible iff s + r >= 0 on all six planes; boundary iff |s + r| < 1e-5 on some plane.
struct CullBox {
    float cwx, cwy, cwz, hx, hy, hz;
};

YH_HD CullBox cull_box(const M4& m, float minx, float miny, float minz, float maxx, float maxy, float maxz) {
    const float cx = (minx + maxx) * 0.5f, cy = (miny + maxy) * 0.5f, cz = (minz + maxz) * 0.5f;
    CullBox b;
    b.hx = (maxx - minx) * 0.5f;
    b.hy = (maxy - miny) * 0.5f;
    b.hz = (maxz - minz) * 0.5f;
    b.cwx = cx * m.v[0] + cy * m.v[4] + cz * m.v[8] + m.v[12];
    b.cwy = cx * m.v[1] + cy * m.v[5] + cz * m.v[9] + m.v[13];
    b.cwz = cx * m.v[2] + cy * m.v[6] + cz * m.v[10] + m.v[14];
    return b;
}

// planes: 24 floats (6 x nx,ny,nz,d).  Returns bit0 = visible, bit1 = boundary.
YH_HD uint32_t cull_frustum(const M4& m, const CullBox& b, const float* __restrict__ planes) {
    bool vis = true, bnd = false;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const float a = planes[p * 4 + 0], bb = planes[p * 4 + 1], c = planes[p * 4 + 2], d = planes[p * 4 + 3];
        const float s = a * b.cwx + bb * b.cwy + c * b.cwz + d;
        const float ri = __builtin_fabsf(a * m.v[0] + bb * m.v[1] + c * m.v[2]);
        const float rj = __builtin_fabsf(a * m.v[4] + bb * m.v[5] + c * m.v[6]);
        const float rk = __builtin_fabsf(a * m.v[8] + bb * m.v[9] + c * m.v[10]);
        const float r = ri * b.hx + rj * b.hy + rk * b.hz;
        const float t = s + r;
        vis = vis && (t >= 0.0f);
        bnd = bnd || (__builtin_fabsf(t) < 1e-5f);
    }
    return (vis ? 1u : 0u) | (bnd ? 2u : 0u);
}

// The same test with the work that planes SHARE done once (cfg3: the four shadow cascades look along one light
// direction, so their 24 planes have 3 distinct normals up to sign; the planes differ only in d).  Everything of a
// plane's test except "+ d" depends on its normal alone:
//     s = ((a*cwx + b*cwy) + c*cwz) + d   = p(n) + d          r = r(n)          t = s + r
// and for the opposite normal -n both p and r come out as exactly -p and r (IEEE negation is exact and rounding is
// symmetric; only the sign of a zero can differ, which no comparison below sees).  The host groups the frame's planes
// by normal (yh_frame.hip build_cull_plan): per distinct normal one evaluation of p and r, per plane
// t = fma(sgn, p, d) + r with sgn = +-1 — the product is exact, so the fused form rounds once, exactly like
// (sgn*p) + d — and the two comparisons.  Bit for bit the visible / boundary flags of cull_frustum on every plane.
struct CullPlanNormal {
    float a, b, c;
    uint32_t mid_end;  // planes [end of the previous normal, mid) use this normal, [mid, end) its negation; mid | end << 16
};
struct CullPlanPlane {
    float d;
    uint32_t frustum;
};
#define YH_PLAN_MAX_PLANES 48
struct CullPlan {
    uint32_t n_normals;  // 0: no plan, every plane is tested on its own (cull_frustum)
    uint32_t pad[3];
    CullPlanNormal normal[YH_PLAN_MAX_PLANES];
    CullPlanPlane plane[YH_PLAN_MAX_PLANES];
};

// WITH_BOUNDARY = false: bit f of the result = NOT visible in frustum f, and min_abs = the smallest |t| met on any
// plane (NaNs skipped, as `fabs(NaN) < 1e-5` is false): a boundary case exists iff min_abs < 1e-5, which is rare (a
// handful per 16M entities), so the per-frustum boundary flags are only worked out — by a second pass with
// WITH_BOUNDARY = true, bits [8, 16) — by waves that hold one.  One plane costs six vector instructions this way
// (add or subtract, add, compare, select, shift-or, min) instead of ten.
// min(lo, |t|) with a NaN t skipped.  On the device one v_min_f32 with the |.| source modifier (written out: the
// compiler would first re-canonicalise the loop-carried `lo`, one more instruction per plane); t comes out of an
// addition, so it is never a signalling NaN.
YH_HD float min_abs_f32(float lo, float t) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_min_f32 %0, %0, |%1|" : "+v"(lo) : "v"(t));
    return lo;
#else
    return __builtin_fminf(lo, __builtin_fabsf(t));
#endif
}

template <bool WITH_BOUNDARY>
YH_HD uint32_t cull_planned(const M4& m, const CullBox& b, const CullPlan& plan, float& min_abs) {
    uint32_t acc = 0;
    uint32_t k = 0;
    float lo = __builtin_inff();
#pragma unroll 1
    for (uint32_t j = 0; j < plan.n_normals; ++j) {
        const float a = plan.normal[j].a, bb = plan.normal[j].b, c = plan.normal[j].c;
        const uint32_t mid = plan.normal[j].mid_end & 0xFFFFu, end = plan.normal[j].mid_end >> 16;
        const float p = a * b.cwx + bb * b.cwy + c * b.cwz;
        const float ri = __builtin_fabsf(a * m.v[0] + bb * m.v[1] + c * m.v[2]);
        const float rj = __builtin_fabsf(a * m.v[4] + bb * m.v[5] + c * m.v[6]);
        const float rk = __builtin_fabsf(a * m.v[8] + bb * m.v[9] + c * m.v[10]);
        const float r = ri * b.hx + rj * b.hy + rk * b.hz;
        // (two loops, so that the sign choice costs no instruction: d - p = (-p) + d exactly)
#define YH_PLAN_TEST(T_EXPR)                                                                              \
    {                                                                                                     \
        const float d = plan.plane[k].d;                                                                  \
        const float t = (T_EXPR) + r;                                                                     \
        const uint32_t f = plan.plane[k].frustum;                                                         \
        acc |= ((t >= 0.0f) ? 0u : 1u) << f;                                                              \
        if (WITH_BOUNDARY) acc |= ((__builtin_fabsf(t) < 1e-5f) ? 256u : 0u) << f;                        \
        else lo = min_abs_f32(lo, t);                                                                     \
    }
#pragma unroll 1
        for (; k < mid; ++k) YH_PLAN_TEST(p + d)
#pragma unroll 1
        for (; k < end; ++k) YH_PLAN_TEST(d - p)
#undef YH_PLAN_TEST
    }
    min_abs = lo;
    return acc;
}

// ---- synthetic scene generator -------------------------------------------------------
YH_HD uint64_t yh_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
YH_HD uint64_t yh_synth_key(uint64_t seed, uint32_t stream) {
    return yh_mix64(seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(stream + 1u)));
}
YH_HD uint64_t yh_synth_raw(uint64_t seed, uint32_t stream, uint64_t index) {
    return yh_mix64(yh_synth_key(seed, stream) + index);
}
YH_HD float yh_synth_uniform(uint64_t seed, uint32_t stream, uint64_t index, float lo, float hi) {
    const float u = (float)(uint32_t)(yh_synth_raw(seed, stream, index) >> 40) * 5.9604644775390625e-08f;
    const float span = hi - lo;
    return lo + u * span;
}
