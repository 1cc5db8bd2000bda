/*
 * yume_oracle.c — CPU ORACLE (test infrastructure, NOT the product).  See yume_oracle.h.
 *
 * Scalar restatement of the reference's transform path, one function per reference function, written to
 * evaluate every f32 expression in the reference's order (no FMA: build with -ffp-contract=off).
 * Parity status: math pinned by restatement + algebraic identities (the reference holds no vectors);
 * ECS semantics pinned against real flecs by oracle/flecs_harness.c; culling is build-defined
 * ("parity unpinned").
 */
#include "yume_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------ math3d */

/* engine/math3d.zig:553-558 */
void yo_identity(yo_mat4* out) {
    memset(out, 0, sizeof *out);
    out->v[0] = out->v[5] = out->v[10] = out->v[15] = 1.0f;
}

/* engine/math3d.zig:583-610.  Element (column c, row r) = A.i[r]*B.c.x + A.j[r]*B.c.y + A.k[r]*B.c.z +
 * A.t[r]*B.c.w, summed left to right; the full 4x4 product, affine structure not exploited. */
void yo_mul(const yo_mat4* a, const yo_mat4* b, yo_mat4* out) {
    yo_mat4 r;
    for (int c = 0; c < 4; ++c) {
        const float bx = b->v[c * 4 + 0], by = b->v[c * 4 + 1], bz = b->v[c * 4 + 2], bw = b->v[c * 4 + 3];
        for (int row = 0; row < 4; ++row) {
            r.v[c * 4 + row] = a->v[0 + row] * bx + a->v[4 + row] * by + a->v[8 + row] * bz + a->v[12 + row] * bw;
        }
    }
    *out = r;
}

/* engine/math3d.zig:612-620 */
void yo_mul_vec4(const yo_mat4* m, const float v[4], float out[4]) {
    float r[4];
    for (int row = 0; row < 4; ++row) {
        r[row] = v[0] * m->v[0 + row] + v[1] * m->v[4 + row] + v[2] * m->v[8 + row] + v[3] * m->v[12 + row];
    }
    memcpy(out, r, sizeof r);
}

/* engine/math3d.zig:628-635 */
void yo_translation(const float t[3], yo_mat4* out) {
    yo_identity(out);
    out->v[12] = t[0];
    out->v[13] = t[1];
    out->v[14] = t[2];
    out->v[15] = 1.0f;
}

/* engine/math3d.zig:718-725 */
void yo_scale(const float s[3], yo_mat4* out) {
    memset(out, 0, sizeof *out);
    out->v[0] = s[0];
    out->v[5] = s[1];
    out->v[10] = s[2];
    out->v[15] = 1.0f;
}

/* engine/math3d.zig:727-749.  The quaternion is used as given (not normalised). */
void yo_from_quat(const float q[4], yo_mat4* out) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, xz = x * z, yz = y * z;
    const float wx = w * x, wy = w * y, wz = w * z;
    memset(out, 0, sizeof *out);
    out->v[0] = 1.0f - 2.0f * (yy + zz);
    out->v[1] = 2.0f * (xy + wz);
    out->v[2] = 2.0f * (xz - wy);
    out->v[4] = 2.0f * (xy - wz);
    out->v[5] = 1.0f - 2.0f * (xx + zz);
    out->v[6] = 2.0f * (yz + wx);
    out->v[8] = 2.0f * (xz + wy);
    out->v[9] = 2.0f * (yz - wx);
    out->v[10] = 1.0f - 2.0f * (xx + yy);
    out->v[15] = 1.0f;
}

/* engine/math3d.zig:816-822: translation_mat.mul(rotation_mat).mul(scale_mat), two full products. */
void yo_compose(const float t[3], const float q[4], const float s[3], yo_mat4* out) {
    yo_mat4 sm, rm, tm, tr;
    yo_scale(s, &sm);
    yo_from_quat(q, &rm);
    yo_translation(t, &tm);
    yo_mul(&tm, &rm, &tr);
    yo_mul(&tr, &sm, out);
}

/* engine/math3d.zig:763-814: cofactor inverse; DeterminantZero iff det == 0 exactly. */
int yo_inverse(const yo_mat4* m, yo_mat4* out) {
    const float n11 = m->v[0], n21 = m->v[1], n31 = m->v[2], n41 = m->v[3];
    const float n12 = m->v[4], n22 = m->v[5], n32 = m->v[6], n42 = m->v[7];
    const float n13 = m->v[8], n23 = m->v[9], n33 = m->v[10], n43 = m->v[11];
    const float n14 = m->v[12], n24 = m->v[13], n34 = m->v[14], n44 = m->v[15];

    const float t11 = n23 * n34 * n42 - n24 * n33 * n42 + n24 * n32 * n43 - n22 * n34 * n43 - n23 * n32 * n44 + n22 * n33 * n44;
    const float t12 = n14 * n33 * n42 - n13 * n34 * n42 - n14 * n32 * n43 + n12 * n34 * n43 + n13 * n32 * n44 - n12 * n33 * n44;
    const float t13 = n13 * n24 * n42 - n14 * n23 * n42 + n14 * n22 * n43 - n12 * n24 * n43 - n13 * n22 * n44 + n12 * n23 * n44;
    const float t14 = n14 * n23 * n32 - n13 * n24 * n32 - n14 * n22 * n33 + n12 * n24 * n33 + n13 * n22 * n34 - n12 * n23 * n34;

    const float det = n11 * t11 + n21 * t12 + n31 * t13 + n41 * t14;
    if (det == 0.0f) return 0;
    const float di = 1.0f / det;

    yo_mat4 r;
    r.v[0] = t11 * di;
    r.v[1] = (n24 * n33 * n41 - n23 * n34 * n41 - n24 * n31 * n43 + n21 * n34 * n43 + n23 * n31 * n44 - n21 * n33 * n44) * di;
    r.v[2] = (n22 * n34 * n41 - n24 * n32 * n41 + n24 * n31 * n42 - n21 * n34 * n42 - n22 * n31 * n44 + n21 * n32 * n44) * di;
    r.v[3] = (n23 * n32 * n41 - n22 * n33 * n41 - n23 * n31 * n42 + n21 * n33 * n42 + n22 * n31 * n43 - n21 * n32 * n43) * di;

    r.v[4] = t12 * di;
    r.v[5] = (n13 * n34 * n41 - n14 * n33 * n41 + n14 * n31 * n43 - n11 * n34 * n43 - n13 * n31 * n44 + n11 * n33 * n44) * di;
    r.v[6] = (n14 * n32 * n41 - n12 * n34 * n41 - n14 * n31 * n42 + n11 * n34 * n42 + n12 * n31 * n44 - n11 * n32 * n44) * di;
    r.v[7] = (n12 * n33 * n41 - n13 * n32 * n41 + n13 * n31 * n42 - n11 * n33 * n42 - n12 * n31 * n43 + n11 * n32 * n43) * di;

    r.v[8] = t13 * di;
    r.v[9] = (n14 * n23 * n41 - n13 * n24 * n41 - n14 * n21 * n43 + n11 * n24 * n43 + n13 * n21 * n44 - n11 * n23 * n44) * di;
    r.v[10] = (n12 * n24 * n41 - n14 * n22 * n41 + n14 * n21 * n42 - n11 * n24 * n42 - n12 * n21 * n44 + n11 * n22 * n44) * di;
    r.v[11] = (n13 * n22 * n41 - n12 * n23 * n41 - n13 * n21 * n42 + n11 * n23 * n42 + n12 * n21 * n43 - n11 * n22 * n43) * di;

    r.v[12] = t14 * di;
    r.v[13] = (n13 * n24 * n31 - n14 * n23 * n31 + n14 * n21 * n33 - n11 * n24 * n33 - n13 * n21 * n34 + n11 * n23 * n34) * di;
    r.v[14] = (n14 * n22 * n31 - n12 * n24 * n31 - n14 * n21 * n32 + n11 * n24 * n32 + n12 * n21 * n34 - n11 * n22 * n34) * di;
    r.v[15] = (n12 * n23 * n31 - n13 * n22 * n31 + n13 * n21 * n32 - n11 * n23 * n32 - n12 * n21 * n33 + n11 * n22 * n33) * di;
    *out = r;
    return 1;
}

/* engine/math3d.zig:649-659: right-handed, depth 0..1; clip w = -z_view. */
void yo_perspective(float fovy_rad, float aspect, float near_z, float far_z, yo_mat4* out) {
    const float f = 1.0f / tanf(fovy_rad / 2.0f);
    memset(out, 0, sizeof *out);
    out->v[0] = f / aspect;
    out->v[5] = f;
    out->v[10] = far_z / (near_z - far_z);
    out->v[11] = -1.0f;
    out->v[14] = -(far_z * near_z) / (far_z - near_z);
}

/* engine/math3d.zig:661-683 (note 2/(bottom-top): Y points down, depth 0..1). */
void yo_orthographic(float l, float r, float b, float t, float n, float f, yo_mat4* out) {
    memset(out, 0, sizeof *out);
    out->v[0] = 2.0f / (r - l);
    out->v[5] = 2.0f / (b - t);
    out->v[10] = 1.0f / (n - f);
    out->v[12] = -(r + l) / (r - l);
    out->v[13] = -(b + t) / (b - t);
    out->v[14] = n / (n - f);
    out->v[15] = 1.0f;
}

/* engine/math3d.zig:84-86, 116-123 */
static void v3_normalized(const float a[3], float out[3]) {
    const float l = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (l > 0.0f) {
        out[0] = a[0] / l;
        out[1] = a[1] / l;
        out[2] = a[2] / l;
    } else {
        out[0] = out[1] = out[2] = 0.0f;
    }
}
/* engine/math3d.zig:125-127 */
static float v3_dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
/* engine/math3d.zig:129-135 */
static void v3_cross(const float a[3], const float b[3], float out[3]) {
    const float x = a[1] * b[2] - a[2] * b[1];
    const float y = a[2] * b[0] - a[0] * b[2];
    const float z = a[0] * b[1] - a[1] * b[0];
    out[0] = x;
    out[1] = y;
    out[2] = z;
}

/* engine/math3d.zig:991-1021 */
void yo_look_at(const float eye[3], const float at[3], const float up[3], yo_mat4* out) {
    float x[3], y[3], z[3], tmp[3];
    tmp[0] = eye[0] - at[0];
    tmp[1] = eye[1] - at[1];
    tmp[2] = eye[2] - at[2];
    v3_normalized(tmp, z);
    v3_normalized(up, y);
    v3_cross(y, z, tmp);
    v3_normalized(tmp, x);
    v3_cross(z, x, tmp);
    v3_normalized(tmp, y);
    out->v[0] = x[0]; out->v[1] = y[0]; out->v[2] = z[0]; out->v[3] = 0.0f;
    out->v[4] = x[1]; out->v[5] = y[1]; out->v[6] = z[1]; out->v[7] = 0.0f;
    out->v[8] = x[2]; out->v[9] = y[2]; out->v[10] = z[2]; out->v[11] = 0.0f;
    out->v[12] = -v3_dot(x, eye);
    out->v[13] = -v3_dot(y, eye);
    out->v[14] = -v3_dot(z, eye);
    out->v[15] = 1.0f;
}

/* engine/math3d.zig:687-710.  @cos / @sin are libm's here; Zig lowers them to its own compiler-rt routines, so the
 * trigonometric values themselves are "parity unpinned" (everything built from them follows the reference). */
void yo_rotation(const float axis_in[3], float angle_rad, yo_mat4* out) {
    float axis[3] = {axis_in[0], axis_in[1], axis_in[2]};
    const float c = cosf(angle_rad), s = sinf(angle_rad);
    const float t = 1.0f - c;
    const float sqr_norm = axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2];
    if (sqr_norm == 0.0f) {
        yo_identity(out);
        return;
    }
    if (fabsf(sqr_norm - 1.0f) > 0.0001f) {
        const float norm = sqrtf(sqr_norm);
        axis[0] = axis[0] / norm;
        axis[1] = axis[1] / norm;
        axis[2] = axis[2] / norm;
    }
    const float x = axis[0], y = axis[1], z = axis[2];
    out->v[0] = x * x * t + c; out->v[1] = y * x * t + z * s; out->v[2] = z * x * t - y * s; out->v[3] = 0.0f;
    out->v[4] = x * y * t - z * s; out->v[5] = y * y * t + c; out->v[6] = z * y * t + x * s; out->v[7] = 0.0f;
    out->v[8] = x * z * t + y * s; out->v[9] = y * z * t - x * s; out->v[10] = z * z * t + c; out->v[11] = 0.0f;
    out->v[12] = 0.0f; out->v[13] = 0.0f; out->v[14] = 0.0f; out->v[15] = 1.0f;
}

/* engine/components/camera.zig:161-166 (updateMatrices) -> :253-302: view = Rx * Ry * Rz * T(-pos); perspective:
 * angles in degrees, projection with unnamed[1][1] negated (Vulkan Y flip); orthographic: angles taken as they come
 * (radians), fixed +-10 box, depth 0..200; view_projection = projection * view. */
void yo_camera_update_matrices(int orthographic, const float pos[3], const float rot[3], float aspect, float fovy_rad,
                               float near_z, float far_z, yo_mat4* view, yo_mat4* projection, yo_mat4* view_projection) {
    const float rad_per_deg = 0.017453292519943295769f;
    static const float ax[3] = {1.0f, 0.0f, 0.0f}, ay[3] = {0.0f, 1.0f, 0.0f}, az[3] = {0.0f, 0.0f, 1.0f};
    yo_mat4 rx, ry, rz, tr, a, b;
    yo_rotation(ax, orthographic ? rot[0] : rot[0] * rad_per_deg, &rx);
    yo_rotation(ay, orthographic ? rot[1] : rot[1] * rad_per_deg, &ry);
    yo_rotation(az, orthographic ? rot[2] : rot[2] * rad_per_deg, &rz);
    const float g[3] = {-pos[0], -pos[1], -pos[2]};
    yo_translation(g, &tr);
    yo_mul(&rx, &ry, &a);
    yo_mul(&a, &rz, &b);
    yo_mul(&b, &tr, view);
    if (orthographic) {
        yo_orthographic(-10.0f, 10.0f, -10.0f, 10.0f, 0.0f, 200.0f, projection);
    } else {
        yo_perspective(fovy_rad, aspect, near_z, far_z, projection);
        projection->v[5] *= -1.0f;
    }
    yo_mul(projection, view, view_projection);
}

void yo_camera_view_projection(int orthographic, const float pos[3], const float rot[3], float aspect, float fovy_rad,
                               float near_z, float far_z, yo_mat4* out) {
    yo_mat4 view, proj;
    yo_camera_update_matrices(orthographic, pos, rot, aspect, fovy_rad, near_z, far_z, &view, &proj, out);
}

/* engine/math3d.zig:418-442 (Quat.toEuler): roll about x, pitch about y (clamped to +-90 degrees when |sinp| >= 1),
 * yaw about z, then radians -> degrees as `mulf(180.0).divf(pi)`.  std.math.atan2 / asin are libm's here ("parity
 * unpinned", like yo_rotation's cos / sin). */
void yo_quat_to_euler(const float q[4], float euler_deg[3]) {
    const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    const float half_pi = (float)(3.14159265358979323846 / 2.0);
    const float pi_f = (float)3.14159265358979323846;
    float e[3] = {0.0f, 0.0f, 0.0f};
    const float sinr_cosp = 2.0f * (qw * qx + qy * qz);
    const float cosr_cosp = 1.0f - 2.0f * (qx * qx + qy * qy);
    e[0] = atan2f(sinr_cosp, cosr_cosp);
    const float sinp = 2.0f * (qw * qy - qz * qx);
    if (fabsf(sinp) >= 1.0f) e[1] = copysignf(half_pi, sinp);
    else e[1] = asinf(sinp);
    const float siny_cosp = 2.0f * (qw * qz + qx * qy);
    const float cosy_cosp = 1.0f - 2.0f * (qy * qy + qz * qz);
    e[2] = atan2f(siny_cosp, cosy_cosp);
    for (int i = 0; i < 3; ++i) euler_deg[i] = (e[i] * 180.0f) / pi_f;
}

/* editor/windows/GameWindow.zig:154-178, per Camera entity: `transform.decompose()` (components/transform.zig:68-70:
 * IDENTITY on InvalidMatrix) -> `rotation.toEuler()` -> `camera.updateMatrices(translation, euler, aspect)`.
 * Returns what yo_decompose returns. */
int yo_camera_update_from_world(int orthographic, const yo_mat4* world, float aspect, float fovy_rad, float near_z,
                                float far_z, yo_mat4* view, yo_mat4* projection, yo_mat4* view_projection) {
    float t[3], q[4], sc[3], euler[3];
    const int ok = yo_decompose(world, t, q, sc);
    if (!ok) { /* math3d.zig:535-539 */
        t[0] = t[1] = t[2] = 0.0f;
        q[0] = q[1] = q[2] = 0.0f; q[3] = 1.0f;
        sc[0] = sc[1] = sc[2] = 1.0f;
    }
    yo_quat_to_euler(q, euler);
    yo_camera_update_matrices(orthographic, t, euler, aspect, fovy_rad, near_z, far_z, view, projection, view_projection);
    return ok;
}

/* ----------------------------------------------------------------------------------------------- systems */

/* engine/systems/TransformSyncSystem.zig:60-73, with the `Up` field read as one shared value (F4). */
void yo_transform_sync_batch(yo_transform68* world, yo_transform68* local, const yo_transform68* parent,
                             size_t count) {
    for (size_t i = 0; i < count; ++i) {
        yo_transform68* wt = &world[i];
        yo_transform68* lt = &local[i];
        if (!parent->dirty && !lt->dirty && !wt->dirty) continue; /* :62-64 */
        wt->dirty = 1;                                            /* :65 */
        if (lt->dirty || parent->dirty) {                         /* :67-68 */
            yo_mul(&parent->matrix, &lt->matrix, &wt->matrix);
        } else {                                                  /* :69-72 */
            yo_mat4 inv;
            if (!yo_inverse(&parent->matrix, &inv)) yo_identity(&inv);
            yo_mul(&inv, &wt->matrix, &lt->matrix);
        }
    }
}

/* engine/systems/PostTransformSystem.zig:31-35 */
void yo_post_transform(yo_transform68* column, size_t count) {
    for (size_t i = 0; i < count; ++i) column[i].dirty = 0;
}

size_t yo_frame_aos68(yo_transform68* world, yo_transform68* local, const uint32_t* parent,
                      const uint32_t* level_offsets, uint32_t n_levels, yo_transform68* root, int n_threads) {
    size_t batches = 0;
    if (n_levels == 0) return 0;
    const size_t n = level_offsets[n_levels];
    /* level 0: every entity is a child of the scene root -> one table, one batch (GameApp.zig:211-215) */
    {
        const size_t lo = level_offsets[0], hi = level_offsets[1];
        if (hi > lo) {
            if (n_threads <= 1) {
                yo_transform_sync_batch(world + lo, local + lo, root, hi - lo);
            } else {
#pragma omp parallel for schedule(static) num_threads(n_threads)
                for (long long c = (long long)lo; c < (long long)hi; c += 4096) {
                    size_t cnt = (size_t)((long long)hi - c < 4096 ? (long long)hi - c : 4096);
                    yo_transform_sync_batch(world + c, local + c, root, cnt);
                }
            }
            batches += 1;
        }
    }
    for (uint32_t k = 1; k < n_levels; ++k) {
        const size_t lo = level_offsets[k], hi = level_offsets[k + 1];
        if (hi <= lo) continue;
        if (n_threads <= 1) {
            /* one batch per run of equal parent: flecs keeps one table per (ChildOf, parent) pair */
            size_t s = lo;
            while (s < hi) {
                size_t e = s + 1;
                while (e < hi && parent[e] == parent[s]) ++e;
                yo_transform_sync_batch(world + s, local + s, &world[parent[s]], e - s);
                ++batches;
                s = e;
            }
        } else {
            /* same rule, rows distributed over threads (each row reads its own parent: identical results) */
#pragma omp parallel for schedule(static) num_threads(n_threads)
            for (long long i = (long long)lo; i < (long long)hi; ++i) {
                yo_transform_sync_batch(world + i, local + i, &world[parent[i]], 1);
            }
            for (size_t s = lo; s < hi; ++s)
                if (s == lo || parent[s] != parent[s - 1]) ++batches;
        }
    }
    /* PostWorldTransform, PostLocalTransform (PostTransformSystem.zig:12-25); the root owns a
     * WorldTransform too, so the world sweep clears it. */
    yo_post_transform(world, n);
    yo_post_transform(local, n);
    yo_post_transform(root, 1);
    return batches;
}

void yo_compose_batch(const float* trs, size_t stride, size_t first, size_t count, yo_transform68* local) {
    for (size_t i = 0; i < count; ++i) {
        const size_t e = first + i;
        const float t[3] = {trs[0 * stride + e], trs[1 * stride + e], trs[2 * stride + e]};
        const float q[4] = {trs[3 * stride + e], trs[4 * stride + e], trs[5 * stride + e], trs[6 * stride + e]};
        const float s[3] = {trs[7 * stride + e], trs[8 * stride + e], trs[9 * stride + e]};
        yo_compose(t, q, s, &local[i].matrix);
        local[i].dirty = 1;
        local[i].pad[0] = local[i].pad[1] = local[i].pad[2] = 0;
    }
}

/* engine/math3d.zig:828-981 (Mat4.decompose; determinant3x3 :984-988; Vec3 helpers :80-156; epsilonEql :1113-1119,
 * epsilon = floatEps(f32)).  unnamed[i][j] = values[i*4+j]: the function's "rows" are the matrix columns.  The
 * perspective vector it solves for is never returned, so only that branch's side effect is kept (the perspective
 * partition is cleared).  Returns 0 where the reference returns error.InvalidMatrix. */
#define YO_EPS 1.1920928955078125e-07f
static int eps_eql(float a, float b) { return fabsf(a - b) <= YO_EPS; }     /* math3d.zig:1113-1115 */
static int eps_not_eql(float a, float b) { return fabsf(a - b) > YO_EPS; }  /* math3d.zig:1117-1119: false for NaN, unlike !eps_eql */
static float v3_len(const float v[3]) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
static void v3_norm_zero(float v[3]) { /* Vec3.normalized: zero vector when the length is not > 0 */
    const float l = v3_len(v);
    if (l > 0.0f) {
        v[0] = v[0] / l; v[1] = v[1] / l; v[2] = v[2] / l;
    } else {
        v[0] = v[1] = v[2] = 0.0f;
    }
}
static void v3_combine(float v[3], const float o[3], float sv, float so) {
    v[0] = v[0] * sv + o[0] * so; v[1] = v[1] * sv + o[1] * so; v[2] = v[2] * sv + o[2] * so;
}
int yo_decompose(const yo_mat4* in, float t[3], float q[4], float sc[3]) {
    float m[16];
    memcpy(m, in->v, sizeof m);
    if (eps_eql(m[15], 0.0f)) return 0;
    const float w = m[15];
    for (int i = 0; i < 15; ++i) m[i] /= w;
    m[15] /= m[15];
    const float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) +
                      m[2] * (m[4] * m[9] - m[5] * m[8]);
    if (eps_eql(det, 0.0f)) return 0;
    if (eps_not_eql(m[3], 0.0f) || eps_not_eql(m[7], 0.0f) || eps_not_eql(m[11], 0.0f)) { /* math3d.zig:854-856 */
        m[3] = m[7] = m[11] = 0.0f;
        m[15] = 1.0f;
    }
    t[0] = m[12]; t[1] = m[13]; t[2] = m[14];
    float row[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) row[i][j] = m[i * 4 + j];
    sc[0] = v3_len(row[0]);
    v3_norm_zero(row[0]);
    float skew_z = v3_dot(row[0], row[1]);
    v3_combine(row[1], row[0], 1.0f, -skew_z);
    sc[1] = v3_len(row[1]);
    if (eps_eql(sc[1], 0.0f)) return 0;
    v3_norm_zero(row[1]);
    float skew_y = v3_dot(row[0], row[2]);
    v3_combine(row[2], row[0], 1.0f, -skew_y);
    float skew_x = v3_dot(row[1], row[2]);
    v3_combine(row[2], row[1], 1.0f, -skew_x);
    sc[2] = v3_len(row[2]);
    if (eps_eql(sc[2], 0.0f)) return 0;
    v3_norm_zero(row[2]);
    float pdum3[3];
    v3_cross(row[1], row[2], pdum3);
    if (v3_dot(row[0], pdum3) < 0.0f) {
        for (int i = 0; i < 3; ++i) {
            sc[i] *= -1.0f;
            row[i][0] = row[i][0] * -1.0f; row[i][1] = row[i][1] * -1.0f; row[i][2] = row[i][2] * -1.0f;
        }
    }
    const float trace = row[0][0] + row[1][1] + row[2][2];
    if (trace > 0.0f) {
        float root = sqrtf(trace + 1.0f);
        q[3] = 0.5f * root;
        root = 0.5f / root;
        q[0] = root * (row[1][2] - row[2][1]);
        q[1] = root * (row[2][0] - row[0][2]);
        q[2] = root * (row[0][1] - row[1][0]);
    } else {
        static const int next[3] = {1, 2, 0};
        int i = 0;
        if (row[1][1] > row[0][0]) i = 1;
        if (row[2][2] > row[i][i]) i = 2;
        const int j = next[i], k = next[j];
        float root = sqrtf(row[i][i] - row[j][j] - row[k][k] + 1.0f);
        q[i] = 0.5f * root;
        root = 0.5f / root;
        if (i == 0) {
            q[1] = root * (row[0][1] + row[1][0]);
            q[2] = root * (row[0][2] + row[2][0]);
            q[3] = root * (row[1][2] - row[2][1]);
        } else if (i == 1) {
            q[0] = root * (row[0][1] + row[1][0]);
            q[2] = root * (row[1][2] + row[2][1]);
            q[3] = root * (row[2][0] - row[0][2]);
        } else {
            q[0] = root * (row[0][2] + row[2][0]);
            q[1] = root * (row[1][2] + row[2][1]);
            q[3] = root * (row[0][1] - row[1][0]);
        }
    }
    (void)skew_x; (void)skew_y; (void)skew_z; /* skew_* /= sc_* in the reference: results are discarded there too */
    return 1;
}

/* The accessors' use of it (components/transform.zig:14-28, 68-82): `decompose() catch Decomposed.IDENTITY`, over
 * a column.  Output: 10 planes t.xyz, q.xyzw, s.xyz of `stride` floats; valid[i] = 0 where IDENTITY was substituted. */
void yo_decompose_batch(const yo_transform68* column, size_t count, float* trs, size_t stride, uint8_t* valid) {
    for (size_t i = 0; i < count; ++i) {
        float t[3], q[4], s[3];
        const int ok = yo_decompose(&column[i].matrix, t, q, s);
        if (!ok) {
            t[0] = t[1] = t[2] = 0.0f;
            q[0] = q[1] = q[2] = 0.0f; q[3] = 1.0f;
            s[0] = s[1] = s[2] = 1.0f;
        }
        for (int c = 0; c < 3; ++c) trs[(size_t)c * stride + i] = t[c];
        for (int c = 0; c < 4; ++c) trs[(size_t)(3 + c) * stride + i] = q[c];
        for (int c = 0; c < 3; ++c) trs[(size_t)(7 + c) * stride + i] = s[c];
        if (valid) valid[i] = (uint8_t)ok;
    }
}

/* ----------------------------------------------------------------------------------------------- culling */

/* Build-defined.  Rows of VP: row_r = (v[r], v[4+r], v[8+r], v[12+r]).
 * L = row3+row0, R = row3-row0, B = row3+row1, T = row3-row1, N = row2 (depth 0..1), F = row3-row2,
 * each divided by |n|. */
void yo_extract_planes(const yo_mat4* vp, float planes[24]) {
    float row[4][4];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) row[r][c] = vp->v[c * 4 + r];
    float p[6][4];
    for (int c = 0; c < 4; ++c) {
        p[0][c] = row[3][c] + row[0][c];
        p[1][c] = row[3][c] - row[0][c];
        p[2][c] = row[3][c] + row[1][c];
        p[3][c] = row[3][c] - row[1][c];
        p[4][c] = row[2][c];
        p[5][c] = row[3][c] - row[2][c];
    }
    for (int i = 0; i < 6; ++i) {
        const float len = sqrtf(p[i][0] * p[i][0] + p[i][1] * p[i][1] + p[i][2] * p[i][2]);
        for (int c = 0; c < 4; ++c) planes[i * 4 + c] = p[i][c] / len;
    }
}

/* One entity against one frustum.  Box centre c = (min+max)*0.5, half extents h = (max-min)*0.5;
 * world centre cw = M*(c,1) (xyz, mulVec4 order math3d.zig:612-620); per plane (n,d):
 *   s = n.cw + d,  r = |n.M.i|*h.x + |n.M.j|*h.y + |n.M.k|*h.z,  visible iff s + r >= 0 for all 6;
 * boundary iff |s + r| < 1e-5 for some plane. */
static void cull_one(const yo_mat4* m, const float* b, const float* planes, uint8_t* vis, uint8_t* bnd) {
    const float cx = (b[0] + b[3]) * 0.5f, cy = (b[1] + b[4]) * 0.5f, cz = (b[2] + b[5]) * 0.5f;
    const float hx = (b[3] - b[0]) * 0.5f, hy = (b[4] - b[1]) * 0.5f, hz = (b[5] - b[2]) * 0.5f;
    const float* v = m->v;
    const float cwx = cx * v[0] + cy * v[4] + cz * v[8] + v[12];
    const float cwy = cx * v[1] + cy * v[5] + cz * v[9] + v[13];
    const float cwz = cx * v[2] + cy * v[6] + cz * v[10] + v[14];
    int visible = 1, boundary = 0;
    for (int p = 0; p < 6; ++p) {
        const float a = planes[p * 4 + 0], bb = planes[p * 4 + 1], c = planes[p * 4 + 2], d = planes[p * 4 + 3];
        const float s = a * cwx + bb * cwy + c * cwz + d;
        const float ri = fabsf(a * v[0] + bb * v[1] + c * v[2]);
        const float rj = fabsf(a * v[4] + bb * v[5] + c * v[6]);
        const float rk = fabsf(a * v[8] + bb * v[9] + c * v[10]);
        const float r = ri * hx + rj * hy + rk * hz;
        const float t = s + r;
        if (!(t >= 0.0f)) visible = 0;
        if (fabsf(t) < 1e-5f) boundary = 1;
    }
    *vis = (uint8_t)visible;
    *bnd = (uint8_t)boundary;
}

void yo_cull(const yo_transform68* world, size_t n, const float* bounds, const uint32_t* mesh_id,
             const float* planes, uint32_t n_frusta, uint8_t* vis, uint8_t* boundary, int n_threads) {
    (void)n_threads;
#pragma omp parallel for schedule(static) num_threads(n_threads > 1 ? n_threads : 1)
    for (long long e = 0; e < (long long)n; ++e) {
        const float* b = mesh_id ? bounds + (size_t)mesh_id[e] * 6 : bounds + (size_t)e * 6;
        for (uint32_t f = 0; f < n_frusta; ++f) {
            cull_one(&world[e].matrix, b, planes + (size_t)f * 24, &vis[(size_t)f * n + e],
                     &boundary[(size_t)f * n + e]);
        }
    }
}

/* ------------------------------------------------------------------------------------------------ layout */

void yo_soa_to_aos68(const float* planes16, size_t stride, size_t first, size_t count, uint8_t dirty,
                     yo_transform68* out) {
    for (size_t i = 0; i < count; ++i) {
        for (int p = 0; p < 16; ++p) out[i].matrix.v[p] = planes16[(size_t)p * stride + first + i];
        out[i].dirty = dirty;
        out[i].pad[0] = out[i].pad[1] = out[i].pad[2] = 0;
    }
}

void yo_aos68_to_soa(const yo_transform68* in, size_t count, float* planes16, size_t stride, size_t first) {
    for (size_t i = 0; i < count; ++i)
        for (int p = 0; p < 16; ++p) planes16[(size_t)p * stride + first + i] = in[i].matrix.v[p];
}

/* ------------------------------------------------------------------------------------- synthetic scenes */

static uint64_t mix64(uint64_t z) { /* splitmix64 finaliser */
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

uint64_t yo_synth_raw(uint64_t seed, uint32_t stream, uint64_t index) {
    const uint64_t key = mix64(seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(stream + 1u)));
    return mix64(key + index);
}

/* 24 random bits -> [0,1) exactly representable in f32; lo + u*(hi-lo), unfused */
static float synth_uniform(uint64_t seed, uint32_t stream, uint64_t index, float lo, float hi) {
    const float u = (float)(uint32_t)(yo_synth_raw(seed, stream, index) >> 40) * 5.9604644775390625e-08f;
    const float span = hi - lo;
    return lo + u * span;
}

void yo_synth_level_offsets(uint64_t n, uint32_t n_levels, uint32_t* level_offsets) {
    const unsigned __int128 denom = ((unsigned __int128)1 << n_levels) - 1;
    uint64_t acc = 0;
    level_offsets[0] = 0;
    for (uint32_t k = 0; k < n_levels; ++k) {
        uint64_t nk = (uint64_t)(((unsigned __int128)n << k) / denom);
        if (k == n_levels - 1) nk = n - acc;
        acc += nk;
        level_offsets[k + 1] = (uint32_t)acc;
    }
}

static void radix_sort_u32(uint32_t* a, size_t n) {
    uint32_t* tmp = (uint32_t*)malloc(n * sizeof(uint32_t));
    if (!tmp) abort();
    uint32_t* src = a;
    uint32_t* dst = tmp;
    for (int pass = 0; pass < 4; ++pass) {
        size_t cnt[257] = {0};
        const int sh = pass * 8;
        for (size_t i = 0; i < n; ++i) cnt[((src[i] >> sh) & 255u) + 1]++;
        for (int i = 0; i < 256; ++i) cnt[i + 1] += cnt[i];
        for (size_t i = 0; i < n; ++i) dst[cnt[(src[i] >> sh) & 255u]++] = src[i];
        uint32_t* t = src;
        src = dst;
        dst = t;
    }
    /* 4 passes: result is back in `a` */
    free(tmp);
}

void yo_synth_parents(uint64_t seed, uint64_t index_base, const uint32_t* level_offsets, uint32_t n_levels,
                      uint32_t* parent) {
    for (uint32_t e = level_offsets[0]; e < level_offsets[1]; ++e) parent[e] = 0xFFFFFFFFu;
    for (uint32_t k = 1; k < n_levels; ++k) {
        const uint32_t lo = level_offsets[k], hi = level_offsets[k + 1];
        const uint32_t plo = level_offsets[k - 1], n_prev = level_offsets[k] - level_offsets[k - 1];
        for (uint32_t e = lo; e < hi; ++e) {
            const uint32_t pick = (uint32_t)(yo_synth_raw(seed, 14, index_base + e) >> 32) % n_prev;
            parent[e] = plo + pick;
        }
        radix_sort_u32(parent + lo, hi - lo);
    }
}

void yo_synth_trs(uint64_t seed, uint64_t index_base, uint64_t n_level0, size_t first, size_t count, float* trs,
                  size_t stride) {
    for (size_t i = 0; i < count; ++i) {
        const size_t e = first + i;
        const uint64_t g = index_base + e;
        const int top = e < n_level0;
        const float tr = top ? 100.0f : 5.0f;
        for (int c = 0; c < 3; ++c) trs[(size_t)c * stride + e] = synth_uniform(seed, (uint32_t)c, g, -tr, tr);
        float q[4];
        for (int c = 0; c < 4; ++c) q[c] = synth_uniform(seed, 3u + (uint32_t)c, g, -1.0f, 1.0f);
        const float len = sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]); /* math3d.zig:404-412 */
        if (len > 0.0f) {
            for (int c = 0; c < 4; ++c) q[c] = q[c] / len;
        } else {
            q[0] = q[1] = q[2] = 0.0f;
            q[3] = 1.0f;
        }
        for (int c = 0; c < 4; ++c) trs[(size_t)(3 + c) * stride + e] = q[c];
        const float slo = top ? 0.5f : 0.8f, shi = top ? 2.0f : 1.25f;
        for (int c = 0; c < 3; ++c) trs[(size_t)(7 + c) * stride + e] = synth_uniform(seed, 7u + (uint32_t)c, g, slo, shi);
    }
}

void yo_synth_aabb(uint64_t seed, uint64_t index_base, size_t first, size_t count, float* bounds) {
    for (size_t i = 0; i < count; ++i) {
        const uint64_t g = index_base + first + i;
        for (int c = 0; c < 3; ++c) {
            const float h = synth_uniform(seed, 10u + (uint32_t)c, g, 0.25f, 2.0f);
            bounds[i * 6 + c] = -h;
            bounds[i * 6 + 3 + c] = h;
        }
    }
}

void yo_synth_mesh_table(uint64_t seed, uint32_t n_meshes, float* table) {
    for (uint32_t m = 0; m < n_meshes; ++m) {
        for (int c = 0; c < 3; ++c) {
            const float h = synth_uniform(seed, 16u + (uint32_t)c, m, 0.25f, 2.0f);
            table[m * 6 + c] = -h;
            table[m * 6 + 3 + c] = h;
        }
    }
}

void yo_synth_mesh_ids(uint64_t seed, uint64_t index_base, uint32_t n_meshes, size_t first, size_t count,
                       uint32_t* mesh_id) {
    for (size_t i = 0; i < count; ++i)
        mesh_id[i] = (uint32_t)(yo_synth_raw(seed, 13, index_base + first + i) >> 32) % n_meshes;
}
