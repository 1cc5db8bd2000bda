// yh_hostmath.hip — host-only helpers of the ABI: Mat4.mul / perspective / orthographic / lookAt, plane extraction, the
// camera pipeline (decompose -> toEuler -> updateMatrices), subtree-closed sharding.  No kernels.
#include "yh_internal.hpp"

// ------------------------------------------------------------------------------------ frustum helpers
extern "C" void yh_mat4_mul(const float a[16], const float b[16], float out[16]) {
    M4 x, y;
    memcpy(x.v, a, 64);
    memcpy(y.v, b, 64);
    const M4 r = m4_mul(x, y);
    memcpy(out, r.v, 64);
}

// engine/math3d.zig:649-659
extern "C" void yh_perspective(float fovy_rad, float aspect, float near_z, float far_z, float out[16]) {
    const float f = 1.0f / tanf(fovy_rad / 2.0f);
    memset(out, 0, 64);
    out[0] = f / aspect;
    out[5] = f;
    out[10] = far_z / (near_z - far_z);
    out[11] = -1.0f;
    out[14] = -(far_z * near_z) / (far_z - near_z);
}

// engine/math3d.zig:661-683
extern "C" void yh_orthographic(float l, float r, float b, float t, float n, float f, float out[16]) {
    memset(out, 0, 64);
    out[0] = 2.0f / (r - l);
    out[5] = 2.0f / (b - t);
    out[10] = 1.0f / (n - f);
    out[12] = -(r + l) / (r - l);
    out[13] = -(b + t) / (b - t);
    out[14] = n / (n - f);
    out[15] = 1.0f;
}

static void h_normalized(const float a[3], float o[3]) {  // math3d.zig:116-123
    const float l = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (l > 0.0f) {
        o[0] = a[0] / l;
        o[1] = a[1] / l;
        o[2] = a[2] / l;
    } else {
        o[0] = o[1] = o[2] = 0.0f;
    }
}
static float h_dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void h_cross(const float a[3], const float b[3], float o[3]) {
    const float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x;
    o[1] = y;
    o[2] = z;
}

// engine/math3d.zig:991-1021
extern "C" void yh_look_at(const float eye[3], const float at[3], const float up[3], float out[16]) {
    float x[3], y[3], z[3], tmp[3] = {eye[0] - at[0], eye[1] - at[1], eye[2] - at[2]};
    h_normalized(tmp, z);
    h_normalized(up, y);
    h_cross(y, z, tmp);
    h_normalized(tmp, x);
    h_cross(z, x, tmp);
    h_normalized(tmp, y);
    const float m[16] = {x[0], y[0], z[0], 0.0f, x[1], y[1], z[1], 0.0f, x[2], y[2], z[2], 0.0f,
                         -h_dot(x, eye), -h_dot(y, eye), -h_dot(z, eye), 1.0f};
    memcpy(out, m, 64);
}

// engine/math3d.zig:687-710 (host libm cos / sin; Zig uses its own routines, so the trig values are not pinned)
static void h_rotation(const float axis_in[3], float angle_rad, float out[16]) {
    float axis[3] = {axis_in[0], axis_in[1], axis_in[2]};
    const float c = cosf(angle_rad), s = sinf(angle_rad);
    const float t = 1.0f - c;
    const float sqr_norm = axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2];
    if (sqr_norm == 0.0f) {
        for (int i = 0; i < 16; ++i) out[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        return;
    }
    if (fabsf(sqr_norm - 1.0f) > 0.0001f) {
        const float norm = sqrtf(sqr_norm);
        for (int i = 0; i < 3; ++i) axis[i] = axis[i] / norm;
    }
    const float x = axis[0], y = axis[1], z = axis[2];
    const float m[16] = {x * x * t + c, y * x * t + z * s, z * x * t - y * s, 0.0f,
                         x * y * t - z * s, y * y * t + c, z * y * t + x * s, 0.0f,
                         x * z * t + y * s, y * z * t - x * s, z * z * t + c, 0.0f,
                         0.0f, 0.0f, 0.0f, 1.0f};
    memcpy(out, m, 64);
}

// engine/components/camera.zig:161-166, 253-302: Camera.updateMatrices — view, projection and view_projection from the
// camera's position and Euler angles (updatePerspective* / updateOrthographic*, updateViewProjection).
extern "C" void yh_camera_update_matrices(int orthographic, const float pos[3], const float rot[3], float aspect,
                                          float fovy_rad, float near_z, float far_z, float view[16], float projection[16],
                                          float view_projection[16]) {
    const float rad_per_deg = 0.017453292519943295769f;  // std.math.degreesToRadians (camera.zig:261-263)
    const float ax[3] = {1.0f, 0.0f, 0.0f}, ay[3] = {0.0f, 1.0f, 0.0f}, az[3] = {0.0f, 0.0f, 1.0f};
    float rx[16], ry[16], rz[16], tr[16], a[16], b[16], v[16], proj[16];
    h_rotation(ax, orthographic ? rot[0] : rot[0] * rad_per_deg, rx);  // camera.zig:283-285 takes the angles as they come
    h_rotation(ay, orthographic ? rot[1] : rot[1] * rad_per_deg, ry);
    h_rotation(az, orthographic ? rot[2] : rot[2] * rad_per_deg, rz);
    for (int i = 0; i < 16; ++i) tr[i] = (i % 5 == 0) ? 1.0f : 0.0f;  // Mat4.translation(-pos), math3d.zig:628-635
    tr[12] = -pos[0];
    tr[13] = -pos[1];
    tr[14] = -pos[2];
    yh_mat4_mul(rx, ry, a);
    yh_mat4_mul(a, rz, b);
    yh_mat4_mul(b, tr, v);
    if (orthographic) {
        yh_orthographic(-10.0f, 10.0f, -10.0f, 10.0f, 0.0f, 200.0f, proj);  // camera.zig:296
    } else {
        yh_perspective(fovy_rad, aspect, near_z, far_z, proj);
        proj[5] *= -1.0f;  // camera.zig:272
    }
    if (view) memcpy(view, v, 64);
    if (projection) memcpy(projection, proj, 64);
    yh_mat4_mul(proj, v, view_projection);  // camera.zig:300-302
}

extern "C" void yh_camera_view_projection(int orthographic, const float pos[3], const float rot[3], float aspect,
                                          float fovy_rad, float near_z, float far_z, float out[16]) {
    yh_camera_update_matrices(orthographic, pos, rot, aspect, fovy_rad, near_z, far_z, nullptr, nullptr, out);
}

// components/transform.zig:68-70: `self.matrix.decompose() catch Mat4.Decomposed.IDENTITY` on one matrix, on the host
// (m4_decompose is the kernels' own restatement of math3d.zig:828-981).  Returns 0 where IDENTITY was substituted.
extern "C" int yh_mat4_decompose(const float m[16], float t[3], float q[4], float s[3]) {
    M4 x;
    memcpy(x.v, m, 64);
    float o[10];
    const bool ok = m4_decompose(x, o);
    if (!ok) {  // math3d.zig:535-539
        o[0] = o[1] = o[2] = 0.0f;
        o[3] = o[4] = o[5] = 0.0f;
        o[6] = 1.0f;
        o[7] = o[8] = o[9] = 1.0f;
    }
    memcpy(t, o, 12);
    memcpy(q, o + 3, 16);
    memcpy(s, o + 7, 12);
    return ok ? 1 : 0;
}

// engine/math3d.zig:418-442 — Quat.toEuler: roll / pitch / yaw in DEGREES.  atan2 / asin are libm's here; Zig 0.13 uses
// its own std.math routines (ports of musl's atan2f / asinf), absent from this image: their last bit is "parity unpinned".
extern "C" void yh_quat_to_euler(const float q[4], float euler_deg[3]) {
    const float half_pi = 1.57079632679489661923f, pi = 3.14159265358979323846f;
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float sinr_cosp = 2.0f * (w * x + y * z);
    const float cosr_cosp = 1.0f - 2.0f * (x * x + y * y);
    const float ex = atan2f(sinr_cosp, cosr_cosp);
    const float sinp = 2.0f * (w * y - z * x);
    const float ey = fabsf(sinp) >= 1.0f ? copysignf(half_pi, sinp) : asinf(sinp);
    const float siny_cosp = 2.0f * (w * z + x * y);
    const float cosy_cosp = 1.0f - 2.0f * (y * y + z * z);
    const float ez = atan2f(siny_cosp, cosy_cosp);
    euler_deg[0] = ex * 180.0f / pi;  // euler.mulf(180.0).divf(std.math.pi)
    euler_deg[1] = ey * 180.0f / pi;
    euler_deg[2] = ez * 180.0f / pi;
}

// editor/windows/GameWindow.zig:160-161 — what the engine does per Camera entity before drawing with it:
//   const decomposed = transform.decompose(); camera.updateMatrices(decomposed.translation, decomposed.rotation.toEuler(), aspect);
extern "C" int yh_camera_update_from_world(int orthographic, const float world[16], float aspect, float fovy_rad,
                                           float near_z, float far_z, float view[16], float projection[16],
                                           float view_projection[16]) {
    float t[3], q[4], s[3], euler[3];
    const int ok = yh_mat4_decompose(world, t, q, s);
    yh_quat_to_euler(q, euler);
    yh_camera_update_matrices(orthographic, t, euler, aspect, fovy_rad, near_z, far_z, view, projection, view_projection);
    return ok;
}

// Gribb-Hartmann, depth 0..1: L,R,B,T,N,F, each divided by |n|.
extern "C" void yh_extract_planes(const float vp[16], float planes_out[24]) {
    float p[6][4];
    for (int c = 0; c < 4; ++c) {
        const float r0 = vp[c * 4 + 0], r1 = vp[c * 4 + 1], r2 = vp[c * 4 + 2], r3 = vp[c * 4 + 3];
        p[0][c] = r3 + r0;
        p[1][c] = r3 - r0;
        p[2][c] = r3 + r1;
        p[3][c] = r3 - r1;
        p[4][c] = r2;
        p[5][c] = r3 - r2;
    }
    for (int i = 0; i < 6; ++i) {
        const float len = sqrtf(p[i][0] * p[i][0] + p[i][1] * p[i][1] + p[i][2] * p[i][2]);
        for (int c = 0; c < 4; ++c) planes_out[i * 4 + c] = p[i][c] / len;
    }
}

// ------------------------------------------------------------------------------------------- sharding
extern "C" int yh_partition_subtrees(const uint32_t* parent, const uint32_t* lo, uint32_t n_levels, uint32_t n_parts,
                                     uint32_t* part_of, uint64_t* part_sizes) {
    if (!parent || !lo || !part_of || !part_sizes) return fail(nullptr, YH_ERR_BAD_ARG, "yh_partition_subtrees: NULL argument");
    if (n_levels < 1 || n_levels > YH_MAX_LEVELS || n_parts < 1) return fail(nullptr, YH_ERR_BAD_ARG, "yh_partition_subtrees: bad n_levels / n_parts");
    const uint32_t n = lo[n_levels], n0 = lo[1];
    // root_of[e] = level-0 ancestor; parents precede children, so one forward sweep is enough
    std::vector<uint32_t> root_of(n);
    std::vector<uint64_t> subtree(n0, 0);
    for (uint32_t e = 0; e < n0; ++e) root_of[e] = e;
    for (uint32_t k = 1; k < n_levels; ++k)
        for (uint32_t e = lo[k]; e < lo[k + 1]; ++e) {
            const uint32_t p = parent[e];
            if (p < lo[k - 1] || p >= lo[k]) return fail(nullptr, YH_ERR_BAD_ARG, "yh_partition_subtrees: entity %u has a parent outside the previous level", e);
            root_of[e] = root_of[p];
        }
    for (uint32_t e = 0; e < n; ++e) subtree[root_of[e]]++;
    // greedy largest-first bin packing (ties: lower slot first, so the result is deterministic)
    std::vector<uint32_t> order(n0);
    for (uint32_t i = 0; i < n0; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return subtree[a] > subtree[b]; });
    std::vector<uint64_t> load(n_parts, 0);
    std::vector<uint32_t> part_of_root(n0);
    for (uint32_t r : order) {
        uint32_t best = 0;
        for (uint32_t p = 1; p < n_parts; ++p)
            if (load[p] < load[best]) best = p;
        part_of_root[r] = best;
        load[best] += subtree[r];
    }
    for (uint32_t e = 0; e < n; ++e) part_of[e] = part_of_root[root_of[e]];
    for (uint32_t p = 0; p < n_parts; ++p) part_sizes[p] = load[p];
    return YH_OK;
}

extern "C" int yh_shard_extract(const uint32_t* parent, const uint32_t* lo, uint32_t n_levels, const uint32_t* part_of,
                                uint32_t part, uint32_t* local_parent, uint32_t* local_lo, uint32_t* global_of_local) {
    if (!parent || !lo || !part_of || !local_parent || !local_lo || !global_of_local)
        return fail(nullptr, YH_ERR_BAD_ARG, "yh_shard_extract: NULL argument");
    if (n_levels < 1 || n_levels > YH_MAX_LEVELS) return fail(nullptr, YH_ERR_BAD_ARG, "yh_shard_extract: bad n_levels");
    const uint32_t n = lo[n_levels];
    std::vector<uint32_t> local_of(n, YH_NO_PARENT);
    uint32_t next = 0;
    local_lo[0] = 0;
    for (uint32_t k = 0; k < n_levels; ++k) {
        for (uint32_t e = lo[k]; e < lo[k + 1]; ++e) {
            if (part_of[e] != part) continue;
            local_of[e] = next;
            global_of_local[next] = e;
            if (k == 0) {
                local_parent[next] = YH_NO_PARENT;
            } else {
                const uint32_t lp = local_of[parent[e]];
                if (lp == YH_NO_PARENT) return fail(nullptr, YH_ERR_BAD_ARG, "yh_shard_extract: entity %u and its parent are in different parts", e);
                local_parent[next] = lp;  // global sibling order is kept, so siblings stay sorted by parent
            }
            ++next;
        }
        local_lo[k + 1] = next;
    }
    return YH_OK;
}

