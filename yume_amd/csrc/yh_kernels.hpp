// yh_kernels.hpp — gfx950 kernels of the transform-and-visibility path.
//
// Data layout in HBM (per scene): every per-entity quantity is an f32/u32 SoA plane of `stride`
// elements (stride = n_entities rounded up to 64, so each plane starts 256-B aligned):
//   trs     10 planes  tx,ty,tz,qx,qy,qz,qw,sx,sy,sz        (Mat4.compose arguments, math3d.zig:816)
//   lmat    16 planes  local Mat4 values[p]                 (imported flecs LocalTransform column)
//   world   16 planes  world Mat4 values[p]                 (WorldTransform.matrix)
//   parent  u32        slot of the parent (previous level)  (ChildOf pair, ecs.zig:470-477)
//   bounds   6 planes  min.xyz,max.xyz  or  u32 mesh id + table (meshes.zig:72-95)
//   ldirty/wdirty  u8  LocalTransform.dirty / WorldTransform.dirty (transform.zig:12,66)
// Entities are stored level-major (depth-sorted), siblings contiguous and sorted by parent, so a wave's
// 64 consecutive children read a short contiguous run of parents.
//
// All kernels are HBM-bound streaming kernels: one entity per lane, 4-byte coalesced plane accesses
// (256 B per wave-instruction), no MFMA.
#pragma once
#include "yh_math.hpp"

#define YH_K_MAX_FRUSTA 8
#define YH_BLOCK 256
#define YH_ITERS 8
#define YH_TILE (YH_BLOCK * YH_ITERS) /* entities per workgroup = per global draw-list reservation */
#define YH_MAX_GROUP 8                /* levels one launch may span */

struct FrameArgs {
    const float* trs;
    const float* lmat;
    const uint32_t* parent;
    const float* bounds;      // per-entity planes, or mesh table (n_meshes x 6)
    const uint32_t* mesh_id;  // mesh-table mode
    float* world;
    uint8_t* ldirty;
    uint8_t* wdirty;
    uint32_t* counts;   // [YH_K_MAX_FRUSTA] visible, then [YH_K_MAX_FRUSTA] boundary
    uint32_t* draw[YH_K_MAX_FRUSTA];
    uint64_t stride;
    uint64_t draw_cap;
    uint32_t begin, end;                 // entity range of this launch (levels lo..hi)
    uint32_t lvl_begin[YH_MAX_GROUP];    // first slot of levels lo+1, lo+2, ... (0xFFFFFFFF when absent)
    uint32_t has_base;                   // lo > 0: the group's top level has stored parents
    uint32_t n_frusta;
    uint32_t id_base;
    uint32_t n_meshes;
    float root[16];
    float planes[YH_K_MAX_FRUSTA * 24];
};

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// LDS layout of the frame kernels: 32 control words, then n_frusta x YH_TILE staged draw indices.
//   [0..7] visible count per frustum   [8..15] boundary count   [16..23] global base of this tile
__device__ __forceinline__ void stage_visible(uint32_t* smem, uint32_t n_frusta, uint32_t vmask, uint32_t bmask,
                                              uint32_t id) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t f = 0; f < n_frusta; ++f) {
        const bool v = (vmask >> f) & 1u;
        const uint64_t m = __ballot(v);
        if (m) {  // wave-uniform
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&smem[f], (uint32_t)__popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (v) smem[32 + f * YH_TILE + base + lane_rank(m)] = id;
        }
        const uint64_t bm = __ballot((bmask >> f) & 1u);
        if (bm && lane == 0) atomicAdd(&smem[8 + f], (uint32_t)__popcll(bm));
    }
}

// One global reservation per tile per frustum, then a coalesced copy of the staged indices.
__device__ __forceinline__ void flush_visible(uint32_t* smem, const FrameArgs& a) {
    __syncthreads();
    const uint32_t tid = threadIdx.x;
    if (tid < a.n_frusta) {
        const uint32_t c = smem[tid];
        smem[16 + tid] = c ? atomicAdd(&a.counts[tid], c) : 0u;
        const uint32_t bc = smem[8 + tid];
        if (bc) atomicAdd(&a.counts[YH_K_MAX_FRUSTA + tid], bc);
    }
    __syncthreads();
    for (uint32_t f = 0; f < a.n_frusta; ++f) {
        const uint32_t c = smem[f];
        const uint64_t g = smem[16 + f];
        uint32_t* __restrict__ out = a.draw[f];
        for (uint32_t i = tid; i < c; i += YH_BLOCK) {
            if (g + i < a.draw_cap) out[g + i] = smem[32 + f * YH_TILE + i];
        }
    }
}

template <int BOUNDS>
__device__ __forceinline__ void cull_entity(const FrameArgs& a, const M4& w, uint32_t e, uint32_t& vmask,
                                            uint32_t& bmask) {
    float b[6];
    if (BOUNDS == 1) {
#pragma unroll
        for (int p = 0; p < 6; ++p) b[p] = a.bounds[(uint64_t)p * a.stride + e];
    } else {
        const uint32_t m = a.mesh_id[e];
#pragma unroll
        for (int p = 0; p < 6; ++p) b[p] = a.bounds[(uint64_t)m * 6 + p];
    }
    const CullBox box = cull_box(w, b[0], b[1], b[2], b[3], b[4], b[5]);
    for (uint32_t f = 0; f < a.n_frusta; ++f) {
        const uint32_t r = cull_frustum(w, box, &a.planes[f * 24]);
        vmask |= (r & 1u) << f;
        bmask |= (r >> 1) << f;
    }
}

// ---------------------------------------------------------------------------------------------------
// Fused frame kernel, TRS locals, every LocalTransform dirty (forward branch of
// TransformSyncSystem.zig:67-68 on every row).
//
// One launch covers a GROUP of up to G consecutive levels.  An entity at depth d inside the group walks
// its d ancestors inside the group, starts from the stored world matrix of the group's base parent (or
// the scene root) and multiplies down: W = ((base * L_a0) * L_a1) ... * L_self — the same products in the
// same order as level-by-level propagation, so the values are identical; the recomputation trades a few
// hundred VALU ops (free under the HBM bound) for G-1 fewer dependent launches.
//   stage 1: Mat4.compose per chain element   stage 2: Mat4.mul chain   stage 3: cull + compaction.
// BOUNDS: 0 none, 1 per-entity planes, 2 mesh table.
template <int G, int BOUNDS>
__global__ __launch_bounds__(YH_BLOCK) void k_frame_trs(const FrameArgs a) {
    extern __shared__ uint32_t smem[];
    const uint32_t tid = threadIdx.x;
    if (tid < 32) smem[tid] = 0;
    __syncthreads();
    const uint64_t tile0 = (uint64_t)a.begin + (uint64_t)blockIdx.x * YH_TILE;
    const uint64_t S = a.stride;

#pragma unroll 1
    for (int it = 0; it < YH_ITERS; ++it) {
        const uint64_t e64 = tile0 + (uint64_t)it * YH_BLOCK + tid;
        const bool valid = e64 < a.end;
        const uint32_t e = (uint32_t)e64;
        uint32_t vmask = 0, bmask = 0;
        if (valid) {
            // depth of e inside the group
            int d = 0;
#pragma unroll
            for (int j = 0; j < G - 1; ++j) d += (e >= a.lvl_begin[j]) ? 1 : 0;
            // ancestor chain: chain[0] = e, chain[j] = parent of chain[j-1]
            uint32_t chain[G];
            uint32_t p = e;
#pragma unroll
            for (int j = 0; j < G; ++j) {
                chain[j] = p;
                if (j < d) p = a.parent[p];
            }
            M4 w;
            if (a.has_base) {
                const uint32_t bp = a.parent[p];
#pragma unroll
                for (int q = 0; q < 16; ++q) w.v[q] = a.world[(uint64_t)q * S + bp];
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) w.v[q] = a.root[q];
            }
#pragma unroll
            for (int j = G - 1; j >= 0; --j) {
                if (j <= d) {
                    const uint32_t c = chain[j];
                    float t[10];
#pragma unroll
                    for (int q = 0; q < 10; ++q) t[q] = a.trs[(uint64_t)q * S + c];
                    const M4 l = m4_compose(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9]);
                    w = m4_mul(w, l);
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) a.world[(uint64_t)q * S + e] = w.v[q];
            if (BOUNDS != 0) cull_entity<BOUNDS>(a, w, e, vmask, bmask);
        }
        if (BOUNDS != 0) stage_visible(smem, a.n_frusta, vmask, bmask, e + a.id_base);
    }
    if (BOUNDS != 0) flush_visible(smem, a);
}

// ---------------------------------------------------------------------------------------------------
// Dirty-aware level kernel, Mat4 locals (imported flecs columns): the literal three-branch rule of
// TransformSyncSystem.zig:60-73, one level per launch (the parent's dirty flag must already be this
// frame's), followed by the culling stage.  The dirty reset of PostTransformSystem.zig:31-35 is done by
// k_clear_dirty after the last level.
template <int BOUNDS>
__global__ __launch_bounds__(YH_BLOCK) void k_level_mat4(const FrameArgs a, const uint32_t root_dirty) {
    extern __shared__ uint32_t smem[];
    const uint32_t tid = threadIdx.x;
    if (tid < 32) smem[tid] = 0;
    __syncthreads();
    const uint64_t tile0 = (uint64_t)a.begin + (uint64_t)blockIdx.x * YH_TILE;
    const uint64_t S = a.stride;

#pragma unroll 1
    for (int it = 0; it < YH_ITERS; ++it) {
        const uint64_t e64 = tile0 + (uint64_t)it * YH_BLOCK + tid;
        const bool valid = e64 < a.end;
        const uint32_t e = (uint32_t)e64;
        uint32_t vmask = 0, bmask = 0;
        if (valid) {
            uint32_t bp = 0;
            bool pd;
            if (a.has_base) {
                bp = a.parent[e];
                pd = a.wdirty[bp] != 0;
            } else {
                pd = root_dirty != 0;
            }
            const bool ld = a.ldirty[e] != 0;
            const bool wd = a.wdirty[e] != 0;
            M4 w;
            if (pd || ld || wd) {  // :62-64
                M4 pm;
                if (a.has_base) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) pm.v[q] = a.world[(uint64_t)q * S + bp];
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q) pm.v[q] = a.root[q];
                }
                a.wdirty[e] = 1;  // :65
                if (ld || pd) {   // :67-68
                    M4 l;
#pragma unroll
                    for (int q = 0; q < 16; ++q) l.v[q] = a.lmat[(uint64_t)q * S + e];
                    w = m4_mul(pm, l);
#pragma unroll
                    for (int q = 0; q < 16; ++q) a.world[(uint64_t)q * S + e] = w.v[q];
                } else {  // :69-72 world-edited entity: re-derive the local matrix
#pragma unroll
                    for (int q = 0; q < 16; ++q) w.v[q] = a.world[(uint64_t)q * S + e];
                    M4 inv;
                    if (!m4_inverse(pm, inv)) inv = m4_identity();
                    const M4 l = m4_mul(inv, w);
#pragma unroll
                    for (int q = 0; q < 16; ++q) ((float*)a.lmat)[(uint64_t)q * S + e] = l.v[q];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) w.v[q] = a.world[(uint64_t)q * S + e];
            }
            if (BOUNDS != 0) cull_entity<BOUNDS>(a, w, e, vmask, bmask);
        }
        if (BOUNDS != 0) stage_visible(smem, a.n_frusta, vmask, bmask, e + a.id_base);
    }
    if (BOUNDS != 0) flush_visible(smem, a);
}

// PostWorldTransform + PostLocalTransform (PostTransformSystem.zig:31-35): both dirty planes to 0.
__global__ __launch_bounds__(YH_BLOCK) void k_clear_dirty(uint32_t* __restrict__ ld, uint32_t* __restrict__ wd,
                                                          uint64_t n_words) {
    const uint64_t i = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (i < n_words) {
        ld[i] = 0;
        wd[i] = 0;
    }
}

__global__ __launch_bounds__(YH_BLOCK) void k_fill_u8(uint8_t* __restrict__ p, uint8_t v, uint64_t first, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (i < count) p[first + i] = v;
}

// Stage 1 alone: batched Mat4.compose (math3d.zig:816-822) into 16 planes.
__global__ __launch_bounds__(YH_BLOCK) void k_compose(const float* __restrict__ trs, float* __restrict__ out, uint64_t S,
                                                      uint64_t out_stride, uint64_t first, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint64_t e = first + i;
    float t[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) t[q] = trs[(uint64_t)q * S + e];
    const M4 l = m4_compose(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9]);
#pragma unroll
    for (int q = 0; q < 16; ++q) out[(uint64_t)q * out_stride + i] = l.v[q];
}

// ---------------------------------------------------------------------------------------------------
// flecs column import/export: 68-byte rows {f32 m[16]; u8 dirty; u8 pad[3]} (transform.zig:8-12) <-> SoA.
// A workgroup moves 256 rows = 4352 dwords with coalesced dword accesses on the AoS side and transposes
// through LDS; the row stride of 17 dwords is odd, so both LDS phases are bank-conflict free.
__global__ __launch_bounds__(YH_BLOCK) void k_aos68_to_soa(const uint32_t* __restrict__ rows, uint64_t count,
                                                           float* __restrict__ planes, uint8_t* __restrict__ dirty,
                                                           uint64_t S, uint64_t first) {
    __shared__ uint32_t tile[YH_BLOCK * 17];
    const uint64_t r0 = (uint64_t)blockIdx.x * YH_BLOCK;
    const uint64_t nrows = (count - r0 < YH_BLOCK) ? (count - r0) : YH_BLOCK;
    const uint64_t ndw = nrows * 17;
    for (uint32_t i = threadIdx.x; i < ndw; i += YH_BLOCK) tile[i] = rows[r0 * 17 + i];
    __syncthreads();
    if (threadIdx.x < nrows) {
        const uint64_t e = first + r0 + threadIdx.x;
#pragma unroll
        for (int q = 0; q < 16; ++q) planes[(uint64_t)q * S + e] = __uint_as_float(tile[threadIdx.x * 17 + q]);
        dirty[e] = (uint8_t)((tile[threadIdx.x * 17 + 16] & 0xFFu) != 0);
    }
}

__global__ __launch_bounds__(YH_BLOCK) void k_soa_to_aos68(const float* __restrict__ planes,
                                                           const uint8_t* __restrict__ dirty, uint64_t S, uint64_t first,
                                                           uint64_t count, uint32_t* __restrict__ rows) {
    __shared__ uint32_t tile[YH_BLOCK * 17];
    const uint64_t r0 = (uint64_t)blockIdx.x * YH_BLOCK;
    const uint64_t nrows = (count - r0 < YH_BLOCK) ? (count - r0) : YH_BLOCK;
    if (threadIdx.x < nrows) {
        const uint64_t e = first + r0 + threadIdx.x;
#pragma unroll
        for (int q = 0; q < 16; ++q) tile[threadIdx.x * 17 + q] = __float_as_uint(planes[(uint64_t)q * S + e]);
        tile[threadIdx.x * 17 + 16] = dirty ? (uint32_t)(dirty[e] != 0) : 0u;  // bool + 3 zero pad bytes
    }
    __syncthreads();
    const uint64_t ndw = nrows * 17;
    for (uint32_t i = threadIdx.x; i < ndw; i += YH_BLOCK) rows[r0 * 17 + i] = tile[i];
}

// ---------------------------------------------------------------------------------------------------
// Synthetic scene generator: TRS + bounds from the counter-based hash, bit-identical to
// the oracle's host generator.  Level 0 (local index < n_level0) draws from the wide ranges.
__global__ __launch_bounds__(YH_BLOCK) void k_synth(float* __restrict__ trs, float* __restrict__ bounds,
                                                    uint32_t* __restrict__ mesh_id, uint32_t n_meshes, uint64_t S,
                                                    uint64_t n, uint64_t n_level0, uint64_t seed, uint64_t index_base) {
    const uint64_t e = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (e >= n) return;
    const uint64_t g = index_base + e;
    const bool top = e < n_level0;
    const float tr = top ? 100.0f : 5.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) trs[(uint64_t)c * S + e] = yh_synth_uniform(seed, (uint32_t)c, g, -tr, tr);
    float q[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) q[c] = yh_synth_uniform(seed, 3u + (uint32_t)c, g, -1.0f, 1.0f);
    const float len = __builtin_sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);  // IEEE (see Makefile)
    if (len > 0.0f) {
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = q[c] / len;
    } else {
        q[0] = q[1] = q[2] = 0.0f;
        q[3] = 1.0f;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) trs[(uint64_t)(3 + c) * S + e] = q[c];
    const float slo = top ? 0.5f : 0.8f, shi = top ? 2.0f : 1.25f;
#pragma unroll
    for (int c = 0; c < 3; ++c) trs[(uint64_t)(7 + c) * S + e] = yh_synth_uniform(seed, 7u + (uint32_t)c, g, slo, shi);
    if (bounds) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float h = yh_synth_uniform(seed, 10u + (uint32_t)c, g, 0.25f, 2.0f);
            bounds[(uint64_t)c * S + e] = -h;
            bounds[(uint64_t)(3 + c) * S + e] = h;
        }
    }
    if (mesh_id) mesh_id[e] = (uint32_t)(yh_synth_raw(seed, 13, g) >> 32) % n_meshes;
}
