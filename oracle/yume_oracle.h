/*
 * yume_oracle.h — CPU ORACLE (test infrastructure, NOT the product).
 *
 * A scalar C restatement of the reference's algorithm for the transform-and-visibility path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this; the product
 * (yume_amd/, include/yume_hip.h) never does.
 *
 * Parity pinning: the reference (rzvxa/yume, Zig 0.13) has no tests, golden vectors or fixtures for
 * this path and cannot be compiled here (no zig).  The math below is therefore a
 * line-by-line restatement of engine/math3d.zig, and the *ECS semantics* (table batches, the shared
 * `Up` parent, parent-first order, dirty propagation across frames) are pinned against the real
 * vendored flecs 4.0.4 by oracle/flecs_harness.c (this container only).  The culling test has no
 * counterpart in the reference: its definition is SURVEY.md §8a A6 — "parity unpinned" for culling.
 *
 * Build with:  gcc -O2 -ffp-contract=off -fno-fast-math   (Zig emits strict, unfused f32).
 * Every function cites the reference file:line it follows (paths under the reference tree).
 */
#ifndef YUME_ORACLE_H
#define YUME_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* engine/math3d.zig:529-549 — column-major, values[c*4+r]; columns i,j,k,t. */
typedef struct yo_mat4 {
    float v[16];
} yo_mat4;

/* engine/components/transform.zig:8-12, 62-66 — extern struct { matrix: Mat4, dirty: bool }.
 * @sizeOf = 68, @alignOf = 4. */
typedef struct yo_transform68 {
    yo_mat4 matrix;
    uint8_t dirty;
    uint8_t pad[3];
} yo_transform68;

/* ---- math3d ---- */
void yo_identity(yo_mat4* out);                                              /* math3d.zig:553-558 */
void yo_mul(const yo_mat4* a, const yo_mat4* b, yo_mat4* out);               /* math3d.zig:583-610 */
void yo_mul_vec4(const yo_mat4* m, const float v[4], float out[4]);          /* math3d.zig:612-620 */
void yo_translation(const float t[3], yo_mat4* out);                         /* math3d.zig:628-635 */
void yo_scale(const float s[3], yo_mat4* out);                               /* math3d.zig:718-725 */
void yo_from_quat(const float q[4] /* x,y,z,w */, yo_mat4* out);             /* math3d.zig:727-749 */
void yo_compose(const float t[3], const float q[4], const float s[3], yo_mat4* out); /* :816-822 */
int yo_inverse(const yo_mat4* m, yo_mat4* out); /* 0 = DeterminantZero        math3d.zig:763-814 */
void yo_perspective(float fovy_rad, float aspect, float near_z, float far_z, yo_mat4* out); /* :649-659 */
void yo_orthographic(float l, float r, float b, float t, float n, float f, yo_mat4* out);   /* :661-683 */
void yo_rotation(const float axis[3], float angle_rad, yo_mat4* out);          /* math3d.zig:687-710 */
void yo_camera_view_projection(int orthographic, const float pos[3], const float rot[3], float aspect, float fovy_rad,
                               float near_z, float far_z, yo_mat4* out);       /* components/camera.zig:253-302 */
void yo_look_at(const float eye[3], const float at[3], const float up[3], yo_mat4* out);    /* :991-1021 */
void yo_camera_update_matrices(int orthographic, const float pos[3], const float rot[3], float aspect, float fovy_rad,
                               float near_z, float far_z, yo_mat4* view, yo_mat4* projection,
                               yo_mat4* view_projection);                      /* components/camera.zig:161-166, 253-302 */
void yo_quat_to_euler(const float q[4] /* x,y,z,w */, float euler_deg[3]);     /* math3d.zig:418-442 */
/* editor/windows/GameWindow.zig:160-161: decompose (IDENTITY on error) -> toEuler -> updateMatrices */
int yo_camera_update_from_world(int orthographic, const yo_mat4* world, float aspect, float fovy_rad, float near_z,
                                float far_z, yo_mat4* view, yo_mat4* projection, yo_mat4* view_projection);

/* ---- systems ---- */
/* engine/systems/TransformSyncSystem.zig:53-74 applied to one flecs table batch.  `parent` is the ONE
 * value matched by the `Up` term (intended semantics; the literal code indexes parent[i]). */
void yo_transform_sync_batch(yo_transform68* world, yo_transform68* local, const yo_transform68* parent,
                             size_t count);
/* engine/systems/PostTransformSystem.zig:27-37 */
void yo_post_transform(yo_transform68* column, size_t count);
/* One `world.progress` worth of the two systems over a depth-sorted scene: levels in order (parent
 * first), within a level one batch per run of equal parent (= one flecs table), then the two
 * dirty-reset sweeps.  root = WorldTransform of the scene root (GameApp.zig:95); its dirty flag is read
 * like any parent's and cleared by the sweep.  Returns the number of batches. */
size_t yo_frame_aos68(yo_transform68* world, yo_transform68* local, const uint32_t* parent,
                      const uint32_t* level_offsets, uint32_t n_levels, yo_transform68* root, int n_threads);
/* Batched Mat4.compose into a LocalTransform column (dirty = 1), from 10 SoA planes of stride `stride`. */
int yo_decompose(const yo_mat4* m, float t[3], float q[4], float s[3]);     /* math3d.zig:828-981; 0 = InvalidMatrix */
void yo_decompose_batch(const yo_transform68* column, size_t count, float* trs, size_t stride, uint8_t* valid);
void yo_compose_batch(const float* trs, size_t stride, size_t first, size_t count, yo_transform68* local);

/* ---- culling (build-defined) ---- */
void yo_extract_planes(const yo_mat4* view_projection, float planes[24]); /* L,R,B,T,N,F; normalised */
/* vis/boundary: n_frusta x n bytes (row f = frustum f).  bounds: per entity 6 floats (min.xyz,max.xyz) when
 * mesh_id == NULL, else a table of 6-float rows indexed by mesh_id[e]. */
void yo_cull(const yo_transform68* world, size_t n, const float* bounds, const uint32_t* mesh_id,
             const float* planes, uint32_t n_frusta, uint8_t* vis, uint8_t* boundary, int n_threads);

/* ---- layout helpers ---- */
void yo_soa_to_aos68(const float* planes16, size_t stride, size_t first, size_t count, uint8_t dirty,
                     yo_transform68* out);
void yo_aos68_to_soa(const yo_transform68* in, size_t count, float* planes16, size_t stride, size_t first);

/* ---- synthetic scenes: counter-based, keyed by (seed, stream, global index) ---- */
uint64_t yo_synth_raw(uint64_t seed, uint32_t stream, uint64_t index);
/* level sizes proportional to 2^k, summing to n */
void yo_synth_level_offsets(uint64_t n, uint32_t n_levels, uint32_t* level_offsets);
/* parent picks: uniform over the previous level, siblings sorted by parent */
void yo_synth_parents(uint64_t seed, uint64_t index_base, const uint32_t* level_offsets, uint32_t n_levels,
                      uint32_t* parent);
/* trs: 10 planes of stride `stride`; entities [first, first+count); local index < n_level0 is level 0 */
void yo_synth_trs(uint64_t seed, uint64_t index_base, uint64_t n_level0, size_t first, size_t count, float* trs,
                  size_t stride);
/* bounds: count x 6 floats (min.xyz,max.xyz), AoS */
void yo_synth_aabb(uint64_t seed, uint64_t index_base, size_t first, size_t count, float* bounds);
void yo_synth_mesh_table(uint64_t seed, uint32_t n_meshes, float* table);
void yo_synth_mesh_ids(uint64_t seed, uint64_t index_base, uint32_t n_meshes, size_t first, size_t count,
                       uint32_t* mesh_id);

#ifdef __cplusplus
}
#endif
#endif
