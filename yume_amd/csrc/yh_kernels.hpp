// yh_kernels.hpp — gfx950 kernels of the transform-and-visibility path.
//
// Data layout in HBM (per scene): "SoA in wave-sized chunks".  Entities are cut into chunks of 64 (one
// wavefront); an array with NP f32 planes stores chunk c as NP consecutive 256-byte rows, row q holding
// plane q of the chunk's 64 entities:
//       byte offset of (entity e, plane q) = (e >> 6) * NP * 256  +  q * 256  +  (e & 63) * 4
// A wave reading plane q of its own chunk issues one fully coalesced 256-B access, all NP accesses of an
// entity share ONE per-lane address (the plane is the instruction's immediate offset, q*256 < 4096), and a
// wave's whole working set is a handful of contiguous 1.5-4 KiB blocks (DRAM-page and TLB friendly) instead
// of 48 streams hundreds of MB apart.
//   trs     NP=10  tx,ty,tz,qx,qy,qz,qw,sx,sy,sz          (Mat4.compose arguments, math3d.zig:816)
//   lmat    NP=16  local Mat4 values[p]                   (imported flecs LocalTransform column)
//   world   NP=16  world Mat4 values[p]                   (WorldTransform.matrix)
//   bounds  NP=6   min.xyz,max.xyz   or  u32 mesh id per entity + table (meshes.zig:72-95)
//   parent  u32    slot of the parent (previous level)    (ChildOf pair, ecs.zig:470-477)
//   ldirty/wdirty  u8  LocalTransform.dirty / WorldTransform.dirty (transform.zig:12,66)
// Entities are stored level-major (depth-sorted), siblings contiguous and sorted by parent, so a wave's
// 64 consecutive children gather from a short contiguous run of parent chunks.
//
// All kernels are HBM-bound streaming kernels: one entity per lane, no MFMA.
//
// The library is built from several translation units (yh_internal.hpp); argument structs, inline device functions and
// kernel TEMPLATES are visible to all of them, a plain __global__ kernel is compiled into the one unit that asks for
// its group before including this header: YH_K_FRAME (yh_frame.hip), YH_K_SCENE (yh_scene.hip), YH_K_CONSUMERS
// (yh_consumers.hip), YH_K_GATHER (yh_comm.hip).
#pragma once
#include <utility>

#include "yh_math.hpp"

#define YH_K_MAX_FRUSTA 8
#define YH_BLOCK 256
#define YH_MAX_GROUP 8 /* levels one launch may span (k_frame_chain) */
#define YH_CHUNK 64

struct FrameArgs {
    const float* trs;
    const float* lmat;
    const uint32_t* parent;
    const float* bounds;      // chunked planes, or mesh table (n_meshes x 6)
    const uint32_t* mesh_id;  // mesh-table mode
    float* world;
    uint8_t* ldirty;
    uint8_t* wdirty;
    const uint64_t* alive;  // [n_chunks] or NULL: slots that hold an entity (Mat4 scenes with spare slots); dead ones are never visible
    uint32_t* bcount;    // [YH_K_MAX_FRUSTA] boundary-case counters of this frame
    uint64_t* vismask;   // [frustum][n_chunks] one visibility bit per entity (wave ballots)
    uint8_t* viscnt;     // [frustum][cnt_stride] popcount of each visibility word (read by k_expand of small scenes)
    uint32_t cnt_stride; // n_chunks rounded up to 16 (rows are read as uint4)
    uint32_t n_chunks;
    uint32_t begin, end;               // entity range of this launch (levels lo..hi)
    uint32_t n_tiles, xcd_per, xcd_k;  // 256-entity tiles of the launch; XCD-aware order (both 0: tile = workgroup index)
    uint32_t lvl_begin[YH_MAX_GROUP];  // chain kernel: first slot of levels lo+1, lo+2, ... (0xFFFFFFFF when absent)
    uint32_t n_frusta;
    uint32_t id_base;
    uint32_t n_meshes;
    float root[16];
    float planes[YH_K_MAX_FRUSTA * 24];
    CullPlan plan;  // the same planes grouped by normal, when some share one (n_normals = 0: not in use)
};

// ---- chunked-plane addressing -------------------------------------------------------------------------
template <int NP>
__host__ __device__ __forceinline__ uint64_t eoff(uint32_t e) {
    return (uint64_t)(e >> 6) * (uint64_t)(NP * 256) + (uint64_t)((e & 63u) << 2);
}
template <bool STREAM = false>
__device__ __forceinline__ float lde(const float* __restrict__ base, uint64_t off, int q) {
    const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off + q * 256);
    if (STREAM) return __builtin_nontemporal_load(p);  // read-once inputs of a scene too large for the Infinity Cache
    return *p;
}
// World matrices are written once per frame and read back at most by the next level: streaming (nontemporal)
// stores keep them from evicting the parents and inputs other waves still need (measured +3..5 % per frame).
// (Other cache policies for these accesses - plain, sc1, sc0 sc1, sc1 nt, sc0 nt, and sc1 / sc1 nt loads - were built and
// measured in round 2: profiles/r02/ab_store_policy.txt, ab_cache_policy.txt.)
__device__ __forceinline__ void ste(float* __restrict__ base, uint64_t off, int q, float v) {
    __builtin_nontemporal_store(v, reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off + q * 256));
}
template <int NP, bool STREAM = false>
__device__ __forceinline__ void load_row(const float* __restrict__ base, uint32_t e, float (&out)[NP]) {
    const uint64_t off = eoff<NP>(e);
#pragma unroll
    for (int q = 0; q < NP; ++q) out[q] = lde<STREAM>(base, off, q);
}
// World matrices of the fused frame kernels: streaming stores that also DROP the line from the XCD's L2 (`sc1 nt`;
// `nt` alone keeps it there).  Written once per frame, read at most by a later launch (through the Infinity Cache) or
// by a consumer: measured against nt / plain / sc1 / sc0 sc1 / sc0 nt (profiles/r02/ab_store_policy.txt) cfg5's
// kernel 4.543 -> 4.490 ms, cfg3's 0.367 -> 0.356, cfg4's 0.2842 -> 0.2824; plain stores 4.623 / 0.381 / 0.309.
// One 64-bit row address, the plane as the instruction's immediate offset (as the compiler addresses store_row).
// PACED (the flat kernels of scenes that stream past the caches, whose waves do nothing but stream; cache-resident
// flat scenes were 0.5-1.4 % slower paced - profiles/r02/ab_pace_small_flat.txt - and are not): sixteen stores issued back to back fill the CU's
// memory pipeline and hold up the other waves' loads behind them; a few idle issue slots after each store let those
// loads in between (profiles/r02/ab_store_pace.txt: s_nop 0 / 7 / 11 / 15, s_sleep 1 / 2 -> cfg5 4.510 / 4.490 /
// 4.484 / 4.485 / 4.479 / 4.494 ms, cfg3 0.3601 / 0.3575 / 0.3569 / 0.3568 / 0.3584 / 0.3596).  Tree levels, whose
// waves wait on their parents' rows anyway, were 0.3 % slower paced (cfg4) and stay unpaced.
template <int Q, bool PACED>
__device__ __forceinline__ void ste_world(float* row, float v) {
    if (PACED) asm volatile("global_store_dword %0, %1, off offset:%2 sc1 nt\n\ts_nop 11" ::"v"(row), "v"(v), "n"(Q * 256) : "memory");
    else asm volatile("global_store_dword %0, %1, off offset:%2 sc1 nt" ::"v"(row), "v"(v), "n"(Q * 256) : "memory");
}
template <bool PACED, int... Q>
__device__ __forceinline__ void store_world_row_impl(float* row, const float (&v)[16], std::integer_sequence<int, Q...>) {
    (ste_world<Q, PACED>(row, v[Q]), ...);
}
template <bool PACED>
__device__ __forceinline__ void store_world_row(float* __restrict__ base, uint32_t e, const float (&v)[16]) {
    store_world_row_impl<PACED>(reinterpret_cast<float*>(reinterpret_cast<char*>(base) + eoff<16>(e)), v, std::make_integer_sequence<int, 16>{});
}

template <int NP>
__device__ __forceinline__ void store_row(float* __restrict__ base, uint32_t e, const float (&v)[NP]) {
    const uint64_t off = eoff<NP>(e);
#pragma unroll
    for (int q = 0; q < NP; ++q) ste(base, off, q, v[q]);
}

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is a workgroup-scope fence over every address space:
// the compiler puts s_waitcnt vmcnt(0) in front of it, i.e. every global load a thread has in flight must have landed
// and every global store must have been ACKNOWLEDGED before the barrier - which serialises a software pipeline whose
// stages talk to each other through LDS alone (k_bucket_scatter: next tile's rows in flight, last tile's rows on their
// way out).  Only for kernels whose threads exchange data through LDS and nothing else between two barriers.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- stage 3 back end, part 1: wavefront ballots ---------------------------------------------------------
// Every wave owns exactly one 64-entity chunk (launches are chunk-aligned), so its ballot IS the chunk's
// visibility word.  Lane 0 stores it: no LDS, no barrier, no atomic on a shared address (tile counters fed by
// one atomic per wave cost 10-70 % of the launch when measured) and nothing to zero between frames.  A chunk
// that straddles two launches is stored by the earlier one (upper lanes zero) and OR-ed into by the first wave
// of the later one; launches of a frame run in stream order, so the plain read-modify-write is race-free.
__device__ __forceinline__ void publish_visible(const FrameArgs& a, uint32_t chunk, bool merge_with_previous_launch,
                                                uint32_t vmask, uint32_t bmask) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t live = a.alive ? a.alive[chunk] : ~0ull;
    // (all frusta's words in ONE store per wave, lane f keeping frustum f's: no gain at 5 frusta, profiles/r02/ab_publish.txt)
    for (uint32_t f = 0; f < a.n_frusta; ++f) {
        const uint64_t m = __ballot((vmask >> f) & 1u) & live;
        const uint64_t bm = __ballot((bmask >> f) & 1u) & live;
        if (lane == 0) {
            uint64_t* w = &a.vismask[(uint64_t)f * a.n_chunks + chunk];
            const uint64_t word = merge_with_previous_launch ? (*w | m) : m;
            *w = word;
            a.viscnt[(uint64_t)f * a.cnt_stride + chunk] = (uint8_t)__popcll(word);
            if (bm) atomicAdd(&a.bcount[f], (uint32_t)__popcll(bm));
        }
    }
}

template <int BOUNDS>
__device__ __forceinline__ void cull_entity(const FrameArgs& a, const M4& w, const float* b6, uint32_t mesh,
                                            uint32_t& vmask, uint32_t& bmask) {
    float b[6];
    if (BOUNDS == 1) {
#pragma unroll
        for (int p = 0; p < 6; ++p) b[p] = b6[p];
    } else {
#pragma unroll
        for (int p = 0; p < 6; ++p) b[p] = a.bounds[mesh * 6u + p];
    }
    const CullBox box = cull_box(w, b[0], b[1], b[2], b[3], b[4], b[5]);
    if (a.plan.n_normals) {  // wave-uniform: planes that share normals are tested per normal (yh_math.hpp cull_planned)
        float min_abs;
        uint32_t acc = cull_planned<false>(w, box, a.plan, min_abs);
        if (__ballot(min_abs < 1e-5f)) acc = cull_planned<true>(w, box, a.plan, min_abs);  // some lane holds a boundary case: rare
        vmask |= ~acc & ((1u << a.n_frusta) - 1u);
        bmask |= (acc >> 8) & 0xFFu;
        return;
    }
#pragma unroll 1
    for (uint32_t f = 0; f < a.n_frusta; ++f) {
        const uint32_t r = cull_frustum(w, box, &a.planes[f * 24]);
        vmask |= (r & 1u) << f;
        bmask |= (r >> 1) << f;
    }
}

// Per-entity streamed inputs of the TRS frame kernel (stage-1 operands, bounds, parent slot).
template <int BOUNDS>
struct TrsInputs {
    float t[10];
    float b[6];
    uint32_t pidx;
    uint32_t mesh;
};

template <int BOUNDS, bool NEED_PARENT, bool STREAM = false, bool STREAM_BOUNDS = false>
__device__ __forceinline__ void load_inputs(const FrameArgs& a, uint32_t e, TrsInputs<BOUNDS>& in) {
    load_row<10, STREAM>(a.trs, e, in.t);
    if (BOUNDS == 1) load_row<6, STREAM || STREAM_BOUNDS>(a.bounds, e, in.b);
    constexpr bool S = STREAM || STREAM_BOUNDS;  // read-once indices of a scene too large for the caches
    in.mesh = (BOUNDS == 2) ? (S ? __builtin_nontemporal_load(a.mesh_id + e) : a.mesh_id[e]) : 0u;
    in.pidx = NEED_PARENT ? (S ? __builtin_nontemporal_load(a.parent + e) : a.parent[e]) : 0u;
}

// ---- stage 3 back end, part 2: prefix sums + expansion into ordered draw lists -----------------------------
// (Defined ahead of the frame kernels: the first launch of a small scene's frame can carry the previous frame's
// expansion, see k_frame_*_x.)  An expand tile is `words` visibility words (32, 64 = 4096 entities, or 256 for scenes
// beyond 8M entities).
// k_count reduces every tile to its visible count; k_expand gives one workgroup to each (tile, frustum): its base
// is the sum of the tile counts before it (read redundantly: a few thousand L2-resident words, cheaper than a
// dependent scan launch), the tile's words are prefix-summed through LDS and the 4 waves expand them
// round-robin, one word per step: lane l emits entity chunk*64+l at rank mbcnt(word).  Ascending entity order,
// coalesced stores, no atomics.  Scenes of at most YH_SELF_COUNT_TILES tiles of at most 64 words (4M entities) skip k_count: the frame
// kernels store each word's popcount as a byte next to it, and every expand workgroup adds up the bytes before its
// tile itself (16 per load, v_dot4: at most 64 KiB of L2-resident bytes) — one dependent launch less per frame, and
// the expansion can ride on the next frame's first launch (4M flat: 86.5 -> 79.1 us; 2048 tiles: worse at 8M).
#define YH_EXPAND_WORDS_MAX 256
#define YH_SEG_SHIFT 12  // k_count segment: 4096 words (262 144 entities), a multiple of every tile size
#define YH_SELF_COUNT_TILES 1024

struct ExpandArgs {
    const uint64_t* vismask;
    const uint8_t* viscnt;  // [frustum][cnt_stride] popcounts of the words, written with them
    uint32_t cnt_stride;
    uint32_t* tilecnt;      // [frustum][tc_stride]: per tile, the entries of its 4096-word segment before it; then per segment, its entries (k_count)
    uint32_t tc_stride;     // n_tiles + segments
    uint32_t* counts;       // device totals [frustum] (input of the RCCL all-gather)
    uint32_t* bcount_cur;   // boundary counters accumulated by this frame's launches
    uint32_t* bcount_next;  // the set of the frame after next: zeroed here (the next frame's launches may be running)
    uint32_t* host_counts;  // host-mapped: [frustum] totals, [8 + frustum] boundary counts
    uint32_t* draw[YH_K_MAX_FRUSTA];
    uint64_t draw_cap;
    uint32_t n_chunks, n_tiles, id_base, self_count;
    uint32_t words;  // visibility words per tile: 32, 64 or 256
    uint32_t fuse_blocks;  // k_frame_*_x: the launch's first fuse_blocks workgroups expand these (the previous frame's) words
};

__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t* s_red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63u) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t r = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    return r;
}

// Expansion of one wave's run of words (lane i of m_lo / m_hi holds word i, zero past the run's end): entry j of the
// run's stretch of the list, out[j], is id0 + 64 * i + (bit index) of the j-th set bit.  Written out in ISA because
// the loop is bound by instruction ISSUE, and by the ONE scalar unit of a CU before the four vector units: per word two
// v_readlane (lane index an immediate: the sixteen words of a group are unrolled, no counter, compare or branch) bring
// the word to the scalar unit, EXEC is narrowed to the word itself (no lane tests its own bit; an empty word simply
// executes nothing), v_mbcnt gives the rank among the set bits, one v_add_lshl the byte offset from the scalar running
// position, one v_add the entity id, and ONE store writes them; s_bcnt1 advances the position.  6 vector + 4 scalar
// instructions per word (round 2's first version, a loop: 6 + 8 and a branch, 13.6 cycles per word and CU at cfg3 =
// the scalar unit's issue rate; round 1: 14 vector instructions per word and two LDS reads).  Every wave that gets
// here has all 64 lanes active (whole 256-thread workgroups, wave-uniform exits only), so EXEC is simply set to the
// word and restored before the block ends; vector stores only.
#define YH_XW(K, STORE)                                 \
    "v_readlane_b32 %[lo], %[mlo], " #K "\n\t"          \
    "v_readlane_b32 %[hi], %[mhi], " #K "\n\t"          \
    "s_nop 0\n\t"                                       \
    "s_mov_b32 exec_lo, %[lo]\n\t"                      \
    "s_mov_b32 exec_hi, %[hi]\n\t"                      \
    "v_mbcnt_lo_u32_b32 %[rank], exec_lo, 0\n\t"        \
    "v_mbcnt_hi_u32_b32 %[rank], exec_hi, %[rank]\n\t"  \
    "v_add_u32 %[id], 64*" #K ", %[id0]\n\t"            \
    "v_add_lshl_u32 %[off], %[rank], %[pos], 2\n\t"     \
    "s_bcnt1_i32_b64 %[cnt], exec\n\t"                  \
    STORE "\n\t"                                        \
    "s_add_u32 %[pos], %[pos], %[cnt]\n\t"
#define YH_XW16(STORE)                                                                                                 \
    YH_XW(0, STORE) YH_XW(1, STORE) YH_XW(2, STORE) YH_XW(3, STORE) YH_XW(4, STORE) YH_XW(5, STORE) YH_XW(6, STORE)     \
    YH_XW(7, STORE) YH_XW(8, STORE) YH_XW(9, STORE) YH_XW(10, STORE) YH_XW(11, STORE) YH_XW(12, STORE) YH_XW(13, STORE) \
    YH_XW(14, STORE) YH_XW(15, STORE)
// `out` as a value the compiler knows to be uniform (the "s" operands of the asm must BE scalar registers; readfirstlane
// returns a signed int: widen each half through uint32_t, or a low half with bit 31 set smears ones over the high half)
__device__ __forceinline__ uint32_t* uniform_ptr(uint32_t* p) {
    const uint64_t o = (uint64_t)(uintptr_t)p;
    const uint32_t o_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)o);
    const uint32_t o_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(o >> 32));
    return (uint32_t*)(uintptr_t)(((uint64_t)o_hi << 32) | (uint64_t)o_lo);
}
// Sixteen words (lanes 0..15 of the masks) straight to the list; returns the entries written.
__device__ __forceinline__ uint32_t expand_run16(uint32_t* out /* uniform */, uint32_t m_lo, uint32_t m_hi, uint32_t id0) {
    uint64_t saved;
    uint32_t lo, hi, pos, cnt, rank, off, id;
    out = uniform_ptr(out);
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "s_mov_b32 %[pos], 0\n\t"
        YH_XW16("global_store_dword %[off], %[id], %[out]")
        "s_mov_b64 exec, %[saved]"
        : [saved] "=&s"(saved), [lo] "=&s"(lo), [hi] "=&s"(hi), [pos] "=&s"(pos), [cnt] "=&s"(cnt),
          [rank] "=&v"(rank), [off] "=&v"(off), [id] "=&v"(id)
        : [mlo] "v"(m_lo), [mhi] "v"(m_hi), [id0] "v"(id0), [out] "s"(out)
        : "memory", "scc");
    return pos;
}
// (The same through LDS - a wave's entries compacted into its own LDS rows and written as whole aligned 256-byte rows -
// was measured slower: profiles/r02/ab_expand.txt, expansion stage cfg3 48.1 vs 39.5 us, cfg4 22.9 vs 18.3.)
#undef YH_XW16
#undef YH_XW
// A run of n_words <= 64 words: groups of sixteen, the masks rotated down by sixteen lanes between groups.
__device__ __forceinline__ void expand_run(uint32_t* out /* uniform: the run's first list entry */, uint32_t m_lo, uint32_t m_hi,
                                           uint32_t n_words /* uniform */, uint32_t id0) {
    const uint32_t src = (((threadIdx.x & 63u) + 16u) & 63u) << 2;
    n_words = __builtin_amdgcn_readfirstlane(n_words);
    for (uint32_t g = 0;;) {
        const uint32_t written = expand_run16(out, m_lo, m_hi, id0);
        g += 16u;
        if (g >= n_words) break;
        out += written;
        id0 += 16u * 64u;
        m_lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)m_lo);
        m_hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)m_hi);
    }
}

// Expansion of tile t of frustum f's visibility words into its stretch of the ordered list.  Every wave owns a CONTIGUOUS
// run of the tile's words: lane l loads word l of the run (one coalesced load), the run's words then come back one at a
// time as wave-uniform values (v_readlane), so the running output position lives on the scalar unit (expand_run).
// Larger tiles are faster (profiles/r02/ab_tile_words.txt: the stage's time is ~1.3 ns x workgroups + the time of the
// words); several consecutive tiles per workgroup - base carried along, the next tile's words prefetched - measured
// equal or slower at every size (profiles/r02/ab_expand_tiles_per_wg.txt) and is gone: one tile per workgroup.
__device__ __forceinline__ void expand_tile(const ExpandArgs& x, const uint32_t t, const uint32_t f) {
    __shared__ uint32_t s_red[4];
    __shared__ uint32_t s_wave[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t* __restrict__ masks = x.vismask + (uint64_t)f * x.n_chunks;
    const uint32_t c0 = t * x.words;
    const uint32_t wpw = x.words >> 2;       // words per wave: 8, 16 or 64
    const uint32_t cw = c0 + wv * wpw;       // first word of this wave's run
    const uint64_t m = (lane < wpw && cw + lane < x.n_chunks) ? masks[cw + lane] : 0ull;
    // base = visible entities in the tiles before this one
    uint32_t part = 0;
    if (x.self_count) {  // 16 byte-counts per load; c0 is a multiple of 32
        const uint4* __restrict__ c16 = reinterpret_cast<const uint4*>(x.viscnt + (uint64_t)f * x.cnt_stride);
        for (uint32_t i = tid; i < c0 / 16; i += YH_BLOCK) {
            const uint4 v = c16[i];
            part = __builtin_amdgcn_udot4(v.x, 0x01010101u, part, false);
            part = __builtin_amdgcn_udot4(v.y, 0x01010101u, part, false);
            part = __builtin_amdgcn_udot4(v.z, 0x01010101u, part, false);
            part = __builtin_amdgcn_udot4(v.w, 0x01010101u, part, false);
        }
    } else {  // k_count left: per tile its segment's entries before it, per segment its entries (a few hundred at most)
        const uint32_t* __restrict__ seg = x.tilecnt + (uint64_t)f * x.tc_stride + x.n_tiles;
        for (uint32_t i = tid; i < (c0 >> YH_SEG_SHIFT); i += YH_BLOCK) part += seg[i];
    }
    uint32_t base = block_sum_256(part, s_red);
    if (!x.self_count) base += x.tilecnt[(uint64_t)f * x.tc_stride + t];
    // (entity counts stay below 2^31: positions and the capacity compare in 32 bits)
    const uint32_t cap = x.draw_cap > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)x.draw_cap;
    uint32_t* __restrict__ out = x.draw[f];
    // visible entities of each wave's run
    uint32_t run = (uint32_t)__popcll(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) run += __shfl_xor(run, o);
    run = __builtin_amdgcn_readfirstlane(run);  // the same in every lane: say so, the branches below are then scalar
    if (lane == 0) s_wave[wv] = run;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wv; ++w) before += s_wave[w];
    const uint32_t own = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (tid == 0) {
        if (t == 0 && f == 0)  // all of them: the frame that uses this set next may cull more frusta than this one
            for (uint32_t q = 0; q < YH_K_MAX_FRUSTA; ++q) x.bcount_next[q] = 0;
        if (t + 1 == x.n_tiles) {  // totals of this frustum, to the device (all-gather) and straight to the host
            x.counts[f] = base + own;
            x.host_counts[f] = base + own;
            x.host_counts[YH_K_MAX_FRUSTA + f] = x.bcount_cur[f];
        }
    }
    const uint32_t n_words = min(wpw, x.n_chunks > cw ? x.n_chunks - cw : 0u);
    if (run == 0 || n_words == 0) return;  // (wave-uniform after the reduction)
    uint32_t pos = __builtin_amdgcn_readfirstlane(base + before);  // scalar: list position of the run's next entry
    const uint32_t m_lo = (uint32_t)m, m_hi = (uint32_t)(m >> 32);
    const uint32_t id0 = (cw << 6) + lane + x.id_base;
    if (pos + run <= cap) {  // the whole run fits the list: the issue-lean loop
        expand_run(out + pos, m_lo, m_hi, n_words, id0);
        return;
    }
    for (uint32_t i = 0; i < n_words; ++i) {  // the list overflows inside this run: every lane checks its own entry
        const uint32_t lo = __builtin_amdgcn_readlane(m_lo, i), hi = __builtin_amdgcn_readlane(m_hi, i);
        const uint64_t word = ((uint64_t)hi << 32) | lo;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
        if (((word >> lane) & 1ull) && pos + rank < cap) out[pos + rank] = id0 + (i << 6);
        pos += (uint32_t)__popcll(word);
    }
}


// ---------------------------------------------------------------------------------------------------
// Fused frame kernel, TRS locals, every LocalTransform dirty (forward branch of
// TransformSyncSystem.zig:67-68 on every row), ONE level per launch:
//   stage 1: Mat4.compose   stage 2: W = W_parent * L   stage 3: cull + visibility ballot.
// One entity per lane, one 64-entity chunk per wave, 256 entities per workgroup (more slices per workgroup and a
// register double buffer of the next slice's inputs were measured and dropped: DESIGN.md §4).  BOUNDS: 0 none,
// 1 per-entity planes, 2 mesh table.  HAS_BASE: the level is > 0 (its entities have stored parents), else the
// parent is the scene root.
// Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one, each XCD has its own L2).  xcd_k: every
// XCD takes runs of xcd_k consecutive tiles, so the parent rows that neighbouring tiles of sorted siblings share are
// fetched into ONE L2 instead of up to eight, while all XCDs keep working in the same region of the arrays (tree
// launches).  xcd_per: one contiguous run of tiles per XCD (flat scenes).  The host picks what was measured faster.
__device__ __forceinline__ uint32_t xcd_tile(const FrameArgs& a, const uint32_t b) {
    if (a.xcd_k) {
        const uint32_t grp = b >> 3, x = b & 7u;
        return ((grp / a.xcd_k) * 8u + x) * a.xcd_k + grp % a.xcd_k;
    }
    return a.xcd_per ? (b & 7u) * a.xcd_per + (b >> 3) : b;
}

template <int BOUNDS, bool HAS_BASE, bool BIG = false>
__device__ __forceinline__ void frame_trs_tile(const FrameArgs& a, const uint32_t tile) {
    const uint32_t tid = threadIdx.x;
    // chunk-aligned mapping: every wave covers exactly one 64-entity chunk of the arrays
    const uint64_t e64 = (uint64_t)(a.begin & ~63u) + (uint64_t)tile * YH_BLOCK + tid;
    if (e64 - (tid & 63u) >= a.end) return;  // this wave's chunk starts past the end (wave-uniform)
    const bool valid = e64 >= a.begin && e64 < a.end;
    // lanes outside [begin, end) redo a valid entity of the launch and store nothing
    const uint32_t e = e64 < a.begin ? a.begin : (e64 < a.end ? (uint32_t)e64 : a.end - 1u);
    TrsInputs<BOUNDS> in;
    load_inputs<BOUNDS, HAS_BASE, BIG>(a, e, in);  // read-once inputs: streamed only when the scene outgrows the cache
    M4 w;
    if (HAS_BASE) {
        load_row<16>(a.world, in.pidx, w.v);
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) w.v[q] = a.root[q];
    }
    w = m4_mul(w, m4_compose(in.t[0], in.t[1], in.t[2], in.t[3], in.t[4], in.t[5], in.t[6], in.t[7], in.t[8], in.t[9]));
    if (valid) store_world_row<!HAS_BASE && BIG>(a.world, e, w.v);
    if (BOUNDS != 0) {
        uint32_t vmask = 0, bmask = 0;
        cull_entity<BOUNDS>(a, w, in.b, in.mesh, vmask, bmask);
        if (!valid) vmask = bmask = 0;
        publish_visible(a, (uint32_t)(e64 >> 6), e64 - (tid & 63u) < a.begin, vmask, bmask);
    }
}

// ---------------------------------------------------------------------------------------------------
// Chain kernel: one launch for a GROUP of up to G consecutive levels.  An entity at depth d inside the group
// walks its d ancestors inside the group, starts from the stored world matrix of the group's base parent (or
// the scene root) and multiplies down: W = ((base * L_a0) * L_a1) ... * L_self — the same products in the
// same order as level-by-level propagation, so the values are identical; the recomputation trades VALU work
// and a dependent walk up the parent chain for G-1 fewer dependent launches.  Best for shallow groups of
// small levels (the top of a tree), where a launch boundary costs more than the level's own traffic.
template <int G, int BOUNDS, bool HAS_BASE, bool BIG = false>
__device__ __forceinline__ void frame_chain_tile(const FrameArgs& a, const uint32_t tile) {
    const uint32_t tid = threadIdx.x;
    const uint64_t e64 = (uint64_t)(a.begin & ~63u) + (uint64_t)tile * YH_BLOCK + tid;
    if (e64 - (tid & 63u) >= a.end) return;  // this wave's chunk starts past the end (wave-uniform)
    const bool valid = e64 >= a.begin && e64 < a.end;
    const uint32_t e = e64 < a.begin ? a.begin : (e64 < a.end ? (uint32_t)e64 : a.end - 1u);
    TrsInputs<BOUNDS> cur;
    load_inputs<BOUNDS, true, false, BIG>(a, e, cur);
    // depth of e inside the group, and its ancestor chain: chain[0] = e, chain[j] = parent of chain[j-1]
    int d = 0;
#pragma unroll
    for (int j = 0; j < G - 1; ++j) d += (e >= a.lvl_begin[j]) ? 1 : 0;
    uint32_t chain[G];
    uint32_t p = e;
    uint32_t pp = cur.pidx;  // parent slot of p (meaningful while p has a stored parent)
#pragma unroll
    for (int j = 0; j < G; ++j) {
        chain[j] = p;
        if (j < d) {
            p = pp;
            if (HAS_BASE || j + 1 < d) pp = a.parent[p];
        }
    }
    M4 w;
    if (HAS_BASE) {
        load_row<16>(a.world, pp, w.v);
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) w.v[q] = a.root[q];
    }
#pragma unroll
    for (int j = G - 1; j >= 1; --j) {
        if (j <= d) {
            float t[10];
            load_row<10>(a.trs, chain[j], t);
            w = m4_mul(w, m4_compose(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9]));
        }
    }
    w = m4_mul(w, m4_compose(cur.t[0], cur.t[1], cur.t[2], cur.t[3], cur.t[4], cur.t[5], cur.t[6], cur.t[7], cur.t[8],
                             cur.t[9]));
    if (valid) store_world_row<false>(a.world, e, w.v);
    if (BOUNDS != 0) {
        uint32_t vmask = 0, bmask = 0;
        cull_entity<BOUNDS>(a, w, cur.b, cur.mesh, vmask, bmask);
        if (!valid) vmask = bmask = 0;
        publish_visible(a, (uint32_t)(e64 >> 6), e64 - (tid & 63u) < a.begin, vmask, bmask);
    }
}

// The one-level kernel with the parents STAGED IN LDS (what north_star names; kept as a selectable variant and an A/B,
// DESIGN.md 4): siblings are sorted by parent, so the 256 children of a workgroup read parents pidx(first) ..
// pidx(last); when that span fits (<= YH_LDS_PARENTS rows) the workgroup loads it once, coalesced, plane by plane, and
// every lane takes its parent's 16 floats from LDS (row q of the staging area holds plane q of the span: lanes with
// neighbouring parents hit neighbouring banks, lanes with the same parent broadcast) instead of through L1/L2.
#define YH_LDS_PARENTS 192
template <int BOUNDS, bool BIG>
__global__ __launch_bounds__(YH_BLOCK) void k_frame_trs_lds(const FrameArgs a) {
    __shared__ float s_par[16 * YH_LDS_PARENTS];
    __shared__ uint32_t s_span[8];
    const uint32_t tile = xcd_tile(a, blockIdx.x);
    if (tile >= a.n_tiles) return;  // whole workgroup
    const uint32_t tid = threadIdx.x;
    const uint64_t e64 = (uint64_t)(a.begin & ~63u) + (uint64_t)tile * YH_BLOCK + tid;
    const bool wave_on = e64 - (tid & 63u) < a.end;  // wave-uniform; no early exit: every wave meets the barriers
    const bool valid = e64 >= a.begin && e64 < a.end;
    const uint32_t e = e64 < a.begin ? a.begin : (e64 < a.end ? (uint32_t)e64 : a.end - 1u);
    TrsInputs<BOUNDS> in;
    load_inputs<BOUNDS, true, BIG>(a, e, in);
    // span of parents the workgroup reads (a true min / max: yh_upload_hierarchy does not insist on sorted siblings)
    uint32_t lo = in.pidx, hi = in.pidx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor(lo, o));
        hi = max(hi, (uint32_t)__shfl_xor(hi, o));
    }
    if ((tid & 63u) == 0) {
        s_span[tid >> 6] = lo;
        s_span[4 + (tid >> 6)] = hi;
    }
    __syncthreads();
    const uint32_t pmin = min(min(s_span[0], s_span[1]), min(s_span[2], s_span[3]));
    const uint32_t span = max(max(s_span[4], s_span[5]), max(s_span[6], s_span[7])) - pmin + 1u;
    M4 w;
    if (span <= YH_LDS_PARENTS) {  // workgroup-uniform
        for (uint32_t i = tid; i < span * 16u; i += YH_BLOCK) {
            const uint32_t q = i / span, p = pmin + (i - q * span);
            s_par[q * YH_LDS_PARENTS + (p - pmin)] = lde(a.world, eoff<16>(p), (int)q);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) w.v[q] = s_par[q * YH_LDS_PARENTS + (in.pidx - pmin)];
    } else {
        load_row<16>(a.world, in.pidx, w.v);
    }
    if (!wave_on) return;
    w = m4_mul(w, m4_compose(in.t[0], in.t[1], in.t[2], in.t[3], in.t[4], in.t[5], in.t[6], in.t[7], in.t[8], in.t[9]));
    if (valid) store_world_row<false>(a.world, e, w.v);
    if (BOUNDS != 0) {
        uint32_t vmask = 0, bmask = 0;
        cull_entity<BOUNDS>(a, w, in.b, in.mesh, vmask, bmask);
        if (!valid) vmask = bmask = 0;
        publish_visible(a, (uint32_t)(e64 >> 6), e64 - (tid & 63u) < a.begin, vmask, bmask);
    }
}

template <int BOUNDS, bool HAS_BASE>
__global__ __launch_bounds__(YH_BLOCK) void k_frame_trs(const FrameArgs a) {
    const uint32_t tile = xcd_tile(a, blockIdx.x);
    if (tile < a.n_tiles) frame_trs_tile<BOUNDS, HAS_BASE>(a, tile);
}
// Scenes that do not fit the 256 MB Infinity Cache stream their read-once inputs past it (nontemporal loads: 16M flat
// entities 382 -> 367 us per frame); smaller scenes find their inputs still cached from the previous frame and are
// 13-25 % FASTER with ordinary loads (4M flat: 99.8 -> 86.4 us, 1M: 24.9 -> 20.9, 500k: 13.5 -> 10.2).  The host picks.
template <int BOUNDS, bool HAS_BASE>
__global__ __launch_bounds__(YH_BLOCK) void k_frame_trs_big(const FrameArgs a) {
    const uint32_t tile = xcd_tile(a, blockIdx.x);
    if (tile < a.n_tiles) frame_trs_tile<BOUNDS, HAS_BASE, true>(a, tile);
}
// (6 waves per SIMD: the chain kernels came out at 81-82 VGPRs, one over the 80 that allow a sixth wave; asking for it
// costs no spill and gives cfg2 +2 %, the dominant launch 0.73 -> 0.76 of the HBM peak.  7 or 8 waves spill: -20..-30 %.)
template <int G, int BOUNDS, bool HAS_BASE>
__global__ __launch_bounds__(YH_BLOCK) __attribute__((amdgpu_waves_per_eu(6))) void k_frame_chain(const FrameArgs a) {
    const uint32_t tile = xcd_tile(a, blockIdx.x);
    if (tile < a.n_tiles) frame_chain_tile<G, BOUNDS, HAS_BASE>(a, tile);
}
// The same for scenes that do not fit the 256 MB Infinity Cache: the bounds, read exactly once, are loaded
// nontemporally (12.5M entities: 0.357 -> 0.342 ms per frame; at 1M entities, where a frame's whole working set
// stays cached from one frame to the next, the same loads cost 6 %: the host picks by scene size).
template <int G, int BOUNDS, bool HAS_BASE>
__global__ __launch_bounds__(YH_BLOCK) __attribute__((amdgpu_waves_per_eu(6))) void k_frame_chain_big(const FrameArgs a) {
    const uint32_t tile = xcd_tile(a, blockIdx.x);
    if (tile < a.n_tiles) frame_chain_tile<G, BOUNDS, HAS_BASE, true>(a, tile);
}
// The first launch of a frame (levels from 0: no stored base parent) with the PREVIOUS frame's expansion on board: the
// launch's first x.fuse_blocks workgroups run expand_tile on the previous frame's words, the others are the frame's
// own tiles.  For scenes of up to ~1M entities a frame is a handful of 5-20 us launches and the expansion's own
// dependent launch is a tenth of it (cfg2: 41.1 -> 37.5 us per frame); the two halves touch different word sets,
// count bytes and boundary counters (the host alternates them).  Larger trees whose first launch is a small one use
// it too (1.5-2 %); flat large scenes keep the separate k_expand launch (the extra LDS and arguments cost their one
// big kernel 1-2 %, more than the boundary).
template <int BOUNDS>
__global__ __launch_bounds__(YH_BLOCK) void k_frame_trs_x(const FrameArgs a, const ExpandArgs x) {
    if (blockIdx.x < x.fuse_blocks) {
        expand_tile(x, blockIdx.x % x.n_tiles, blockIdx.x / x.n_tiles);
        return;
    }
    const uint32_t tile = xcd_tile(a, blockIdx.x - x.fuse_blocks);
    if (tile < a.n_tiles) frame_trs_tile<BOUNDS, false>(a, tile);
}
template <int G, int BOUNDS>
__global__ __launch_bounds__(YH_BLOCK) __attribute__((amdgpu_waves_per_eu(6))) void k_frame_chain_x(const FrameArgs a, const ExpandArgs x) {
    if (blockIdx.x < x.fuse_blocks) {
        expand_tile(x, blockIdx.x % x.n_tiles, blockIdx.x / x.n_tiles);
        return;
    }
    const uint32_t tile = xcd_tile(a, blockIdx.x - x.fuse_blocks);
    if (tile < a.n_tiles) frame_chain_tile<G, BOUNDS, false>(a, tile);
}

// YH_FRAME_CULL_ONLY: the world matrices of the last frame against this frame's frusta, every slot of the scene in one
// launch (no hierarchy: a row is a row).
template <int BOUNDS>
__global__ __launch_bounds__(YH_BLOCK) void k_cull_only(const FrameArgs a, uint32_t* __restrict__ changed_counts) {
    const uint32_t tid = threadIdx.x;
    if (changed_counts && blockIdx.x == 0 && tid < 2) changed_counts[tid] = 0;  // change tracking: this frame changed no row
    const uint64_t e64 = (uint64_t)blockIdx.x * YH_BLOCK + tid;
    if (e64 - (tid & 63u) >= a.end) return;  // wave-uniform
    const bool valid = e64 < a.end;
    const uint32_t e = valid ? (uint32_t)e64 : a.end - 1u;
    float b[6] = {0, 0, 0, 0, 0, 0};
    uint32_t mesh = 0;
    if (BOUNDS == 1) load_row<6>(a.bounds, e, b);
    else mesh = a.mesh_id[e];
    M4 w;
    load_row<16>(a.world, e, w.v);
    uint32_t vmask = 0, bmask = 0;
    cull_entity<BOUNDS>(a, w, b, mesh, vmask, bmask);
    if (!valid) vmask = bmask = 0;
    publish_visible(a, (uint32_t)(e64 >> 6), false, vmask, bmask);
}

// WorldTransform.dirty on the device is "non-zero".  A row that took the reverse branch (its LOCAL matrix was re-derived,
// TransformSyncSystem.zig:69-72) carries this value instead of 1 until the frame's sweep, which files it under the rows
// whose LocalTransform has to go back to the host; every other reader tests != 0 only.  (Uploads normalise to 0 / 1.)
#define YH_WDIRTY_LOCAL_REWRITTEN 3

// ---------------------------------------------------------------------------------------------------
// Dirty-aware level kernel, Mat4 locals (imported flecs columns): the literal three-branch rule of
// TransformSyncSystem.zig:60-73, one level per launch (the parent's dirty flag must already be this
// frame's), followed by the culling stage.  The dirty reset of PostTransformSystem.zig:31-35 is done by
// k_clear_dirty after the last level.
// SPEC (frames in which most rows are expected to be clean - every frame that is not flagged all-dirty and whose root
// did not move): a clean row needs its stored world matrix for the cull, and which branch a row takes is known only
// after parent index -> parent flag, two dependent loads.  SPEC issues the row's own loads - flags, bounds, the stored
// world matrix - before that chain instead of after it: three memory round trips become two, and a dirty row has
// fetched 64 bytes it does not use (the reverse branch uses them).  The bounds, needed by every row, come first in
// both variants.
template <int BOUNDS, bool HAS_BASE, bool BIG, bool SPEC>  // BIG: the scene outgrows the Infinity Cache, read-once rows are streamed
__global__ __launch_bounds__(YH_BLOCK) void k_level_mat4(const FrameArgs a, const uint32_t root_dirty) {
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = xcd_tile(a, blockIdx.x);
    if (tile >= a.n_tiles) return;
    const uint64_t e64 = (uint64_t)(a.begin & ~63u) + (uint64_t)tile * YH_BLOCK + tid;
    if (e64 - (tid & 63u) >= a.end) return;  // this wave's chunk starts past the end (wave-uniform)
    const bool valid = e64 >= a.begin && e64 < a.end;
    const uint32_t e = (uint32_t)e64;
    uint32_t vmask = 0, bmask = 0;
    if (valid) {
        uint32_t bp = 0;
        if (HAS_BASE) bp = a.parent[e];
        const uint8_t ld8 = a.ldirty[e], wd8 = a.wdirty[e];
        float b[6] = {0, 0, 0, 0, 0, 0};
        uint32_t mesh = 0;
        if (BOUNDS == 1) load_row<6, BIG>(a.bounds, e, b);
        else if (BOUNDS == 2) mesh = a.mesh_id[e];
        M4 w;
        if (SPEC) load_row<16>(a.world, e, w.v);
        const bool pd = HAS_BASE ? a.wdirty[bp] != 0 : root_dirty != 0;
        const bool ld = ld8 != 0, wd = wd8 != 0;
        if (pd || ld || wd) {  // :62-64
            M4 pm;
            if (HAS_BASE) {
                load_row<16>(a.world, bp, pm.v);
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) pm.v[q] = a.root[q];
            }
            a.wdirty[e] = (ld || pd) ? 1 : YH_WDIRTY_LOCAL_REWRITTEN;  // :65 (non-zero = dirty; see k_sweep_dirty)
            if (ld || pd) {   // :67-68
                M4 l;
                load_row<16, BIG>(a.lmat, e, l.v);
                w = m4_mul(pm, l);
                store_row<16>(a.world, e, w.v);
            } else {  // :69-72 world-edited entity: re-derive the local matrix
                if (!SPEC) load_row<16>(a.world, e, w.v);
                M4 inv;
                if (!m4_inverse(pm, inv)) inv = m4_identity();
                const M4 l = m4_mul(inv, w);
                store_row<16>(const_cast<float*>(a.lmat), e, l.v);
            }
        } else if (!SPEC) {
            load_row<16>(a.world, e, w.v);
        }
        if (BOUNDS != 0) cull_entity<BOUNDS>(a, w, b, mesh, vmask, bmask);
    }
    if (BOUNDS != 0) publish_visible(a, (uint32_t)(e64 >> 6), e64 - (tid & 63u) < a.begin, vmask, bmask);
}

// The same rule for a GROUP of consecutive levels in one launch (the Mat4 twin of k_frame_chain).  An entity walks up
// to the group's first level, takes the stored world matrix AND this frame's dirty flag of the base parent (written by
// an earlier launch, or the scene root's), and replays the rule down its chain in registers: an ancestor that is
// dirty through its local or its parent contributes parent * local (forward branch), any other ancestor contributes
// its stored world matrix (clean, or edited in world space: the reverse branch leaves the world matrix alone), and
// "dirty" propagates exactly as WorldTransform.dirty does level by level.  The entity's own row then goes through
// the literal three branches.  Flags read from an ancestor may already have been rewritten by that ancestor's own
// lane in this launch: the only rewrite is wdirty -> non-zero on a row the rule found dirty anyway, so the replay's
// outcome is the same; world[a] is only rewritten in the forward branch, where the replay does not read it; lmat[a]
// only in the reverse branch, where the replay does not read it either.
// INVARIANT (a checkable rule for whoever adds a writer to a level launch): inside k_level_mat4 / k_chain_mat4 the only
// store to wdirty is "to a non-zero value" (1, or 3 on a reverse-branch row whose flag was non-zero already) on the lane's
// OWN row, every reader tests != 0 only, no launch clears a flag (that is k_clear_dirty / k_sweep_dirty's
// job, after the last level), a lane stores world[e] only where ld || pd and lmat[e] only where !(ld || pd) && wd of
// its own row e, and a replay reads world[anc] only where !(ld_anc || pd) and lmat[anc] only where ld_anc || pd: the
// sets "rows a lane may store" and "rows a replay may load" are disjoint per array.  Pinned by
// tests/test_gpu_columns.py::test_chain_replay_equals_one_level_per_launch_on_tiny_mixed_levels (8 tiny mixed levels
// in one launch against one launch per level and the oracle).
template <int G, int BOUNDS, bool HAS_BASE>
__global__ __launch_bounds__(YH_BLOCK) void k_chain_mat4(const FrameArgs a, const uint32_t root_dirty) {
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = xcd_tile(a, blockIdx.x);
    if (tile >= a.n_tiles) return;
    const uint64_t e64 = (uint64_t)(a.begin & ~63u) + (uint64_t)tile * YH_BLOCK + tid;
    if (e64 - (tid & 63u) >= a.end) return;  // wave-uniform
    const bool valid = e64 >= a.begin && e64 < a.end;
    const uint32_t e = (uint32_t)e64;
    uint32_t vmask = 0, bmask = 0;
    if (valid) {
        int d = 0;
#pragma unroll
        for (int j = 0; j < G - 1; ++j) d += (e >= a.lvl_begin[j]) ? 1 : 0;
        uint32_t chain[G];
        uint32_t p = e, pp = 0;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            chain[j] = p;
            if (j < d || (HAS_BASE && j == d)) {
                pp = a.parent[p];
                if (j < d) p = pp;
            }
        }
        M4 pm;
        bool pd;
        if (HAS_BASE) {  // pp = parent of chain[d]
            load_row<16>(a.world, pp, pm.v);
            pd = a.wdirty[pp] != 0;
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) pm.v[q] = a.root[q];
            pd = root_dirty != 0;
        }
#pragma unroll
        for (int j = G - 1; j >= 1; --j) {
            if (j <= d) {
                const uint32_t anc = chain[j];
                const bool ld = a.ldirty[anc] != 0;
                const bool wd = a.wdirty[anc] != 0;
                if (ld || pd) {
                    M4 l;
                    load_row<16>(a.lmat, anc, l.v);
                    pm = m4_mul(pm, l);
                } else {
                    load_row<16>(a.world, anc, pm.v);
                }
                pd = pd || ld || wd;
            }
        }
        const bool ld = a.ldirty[e] != 0;
        const bool wd = a.wdirty[e] != 0;
        M4 w;
        if (pd || ld || wd) {  // TransformSyncSystem.zig:62-64
            a.wdirty[e] = (ld || pd) ? 1 : YH_WDIRTY_LOCAL_REWRITTEN;  // :65
            if (ld || pd) {    // :67-68
                M4 l;
                load_row<16>(a.lmat, e, l.v);
                w = m4_mul(pm, l);
                store_row<16>(a.world, e, w.v);
            } else {  // :69-72
                load_row<16>(a.world, e, w.v);
                M4 inv;
                if (!m4_inverse(pm, inv)) inv = m4_identity();
                const M4 l = m4_mul(inv, w);
                store_row<16>(const_cast<float*>(a.lmat), e, l.v);
            }
        } else {
            load_row<16>(a.world, e, w.v);
        }
        if (BOUNDS != 0) {
            float b[6] = {0, 0, 0, 0, 0, 0};
            uint32_t mesh = 0;
            if (BOUNDS == 1) load_row<6>(a.bounds, e, b);
            else mesh = a.mesh_id[e];
            cull_entity<BOUNDS>(a, w, b, mesh, vmask, bmask);
        }
    }
    if (BOUNDS != 0) publish_visible(a, (uint32_t)(e64 >> 6), e64 - (tid & 63u) < a.begin, vmask, bmask);
}

#ifdef YH_K_FRAME
// Tile bases of the expansion, from the popcount BYTES the frame kernels left next to the words: one workgroup per
// segment of 4096 words (16 bytes = one load per thread), grid = (segments, frusta).  Writes, per tile, the entries of
// its segment before it, and per segment its entries: an expand workgroup then adds up the segments before its own (at
// most a few hundred values even at 256M entities) instead of every tile before its own (round 2's first version:
// 15 625 tiles at 256M entities, 31 KB read per workgroup, more time than the expansion itself).
__global__ __launch_bounds__(YH_BLOCK) void k_count(const ExpandArgs x) {
    __shared__ uint32_t s_w[4];
    const uint32_t seg = blockIdx.x, f = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t w0 = (seg << YH_SEG_SHIFT) + tid * 16u;  // first word of this thread's 16 count bytes
    uint32_t c = 0;
    if (w0 < x.n_chunks) {
        const uint4 v = *reinterpret_cast<const uint4*>(x.viscnt + (uint64_t)f * x.cnt_stride + w0);
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
        const uint32_t valid = x.n_chunks - w0;  // bytes past the last word are padding: whatever was there
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t nb = valid > 4u * q ? valid - 4u * q : 0u;
            const uint32_t keep = nb >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nb)) - 1u);
            c = __builtin_amdgcn_udot4(d[q] & keep, 0x01010101u, c, false);
        }
    }
    uint32_t inc = c;  // inclusive scan over the wave, then over the four waves
#pragma unroll
    for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63u) s_w[wv] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wv; ++w) before += s_w[w];
    const uint32_t excl = before + inc - c;  // entries of this segment before this thread's words
    uint32_t* __restrict__ row = x.tilecnt + (uint64_t)f * x.tc_stride;
    if (w0 < x.n_chunks && w0 % x.words == 0) row[w0 / x.words] = excl;  // x.words is 32, 64 or 256: a tile starts at a thread
    if (tid == YH_BLOCK - 1) row[x.n_tiles + seg] = excl + c;
}

// grid = (n_tiles, frusta)
__global__ __launch_bounds__(YH_BLOCK) void k_expand(const ExpandArgs x) { expand_tile(x, blockIdx.x, blockIdx.y); }
#endif  // YH_K_FRAME

// The same sweep when changes are tracked (yh_scene_track_changes): one wave per 64-entity chunk first publishes two
// ballots as the chunk's words of the two change masks, then clears both flags.
//   mask 0: LocalTransform.dirty or WorldTransform.dirty - every row this frame recomputed or found edited (its
//           WorldTransform row, matrix or flag, differs from what the host holds);
//   mask 1: LocalTransform.dirty, or the frame re-derived the local matrix (reverse branch) - the rows whose
//           LocalTransform differs from what the host holds (rewritten_only: the reverse-branch rows alone - a host
//           that cleared the flag of every LocalTransform row it sent holds the others as the frame leaves them).  A root that moves dirties every world matrix below it and
//           not one local matrix: only mask 0's rows need to cross PCIe.
// k_count / k_expand turn the masks into ascending lists of slots, exactly as they do for two frusta.
// Sixteen entities per lane (one 16-byte load per flag plane: a byte per lane and instruction costs the memory pipeline
// as much as sixteen; 12.5M entities: 30.9 -> 9 us); four neighbouring lanes make one chunk's words.
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x) {  // bit i = byte i of x is not zero
    const uint32_t t = ((x | ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) & 0x80808080u) >> 7;
    return (t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu;
}
#ifdef YH_K_FRAME
__global__ __launch_bounds__(YH_BLOCK) void k_sweep_dirty(uint8_t* __restrict__ ld, uint8_t* __restrict__ wd, uint32_t n,
                                                          uint64_t* __restrict__ words, uint8_t* __restrict__ counts,
                                                          uint32_t n_chunks, uint32_t cnt_stride,
                                                          const uint64_t* __restrict__ alive, const uint32_t rewritten_only) {
    const uint64_t e0 = ((uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x) * 16u;  // this lane's first entity
    const uint32_t keep_set = rewritten_only ? 0u : 0xFu;  // YH_TRACK_REWRITTEN_LOCALS: mask 1 = reverse-branch rows only
    const uint64_t padded = ((uint64_t)n + 63u) & ~63ull;                         // the planes' length: whole chunks
    uint32_t m = 0, ml = 0;
    if (e0 < padded) {
        const uint4 l = *reinterpret_cast<const uint4*>(ld + e0), w = *reinterpret_cast<const uint4*>(wd + e0);
        const uint32_t lq[4] = {l.x, l.y, l.z, l.w}, wq[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t lnz = nonzero_bytes(lq[q]);
            m |= (lnz | nonzero_bytes(wq[q])) << (4 * q);
            // bytes equal to YH_WDIRTY_LOCAL_REWRITTEN: the zero bytes of the xor
            ml |= ((lnz & keep_set) | (~nonzero_bytes(wq[q] ^ (0x01010101u * YH_WDIRTY_LOCAL_REWRITTEN)) & 0xFu)) << (4 * q);
        }
        const uint32_t valid = e0 + 16u <= n ? 0xFFFFu : (e0 < n ? (1u << (uint32_t)(n - e0)) - 1u : 0u);  // rows past n are padding
        m &= valid;
        ml &= valid;
        *reinterpret_cast<uint4*>(ld + e0) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(wd + e0) = make_uint4(0, 0, 0, 0);
    }
    // lanes 4k .. 4k+3 hold the four quarters of chunk k's words
    uint32_t lo = m | ((uint32_t)__shfl_down((int)m, 1) << 16), lol = ml | ((uint32_t)__shfl_down((int)ml, 1) << 16);
    const uint32_t hi = (uint32_t)__shfl_down((int)lo, 2), hil = (uint32_t)__shfl_down((int)lol, 2);
    if ((threadIdx.x & 3u) == 0 && e0 < padded) {
        const uint64_t c = e0 >> 6, live = alive ? alive[c] : ~0ull;  // a dead slot's rows are nobody's: never reported
        const uint64_t word = (((uint64_t)hi << 32) | lo) & live, wordl = (((uint64_t)hil << 32) | lol) & live;
        words[c] = word;
        counts[c] = (uint8_t)__popcll(word);
        words[(uint64_t)n_chunks + c] = wordl;
        counts[(uint64_t)cnt_stride + c] = (uint8_t)__popcll(wordl);
    }
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

// Boundary-counter sets first .. first + n_sets - 1 (mod ring) to zero by plain stores: what k_expand does for one set
// per frame, done for many frames at once when frames leave their lists unexpanded (YH_FRAME_LISTS_ON_DEMAND).
__global__ __launch_bounds__(YH_BLOCK) void k_zero_sets(uint32_t* __restrict__ sets, uint32_t first, uint32_t n_sets, uint32_t ring) {
    const uint32_t t = threadIdx.x;  // YH_BLOCK threads = 32 sets of YH_K_MAX_FRUSTA (8) words
    if (t < n_sets * 8u) sets[((first + t / 8u) % ring) * 8u + (t & 7u)] = 0u;
}
#endif  // YH_K_FRAME

#ifdef YH_K_SCENE
__global__ __launch_bounds__(YH_BLOCK) void k_fill_u8(uint8_t* __restrict__ p, uint8_t v, uint64_t first, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (i < count) p[first + i] = v;
}

// Measurement harness (yh_stream_probe): the plainest read + write stream there is — 16-byte loads of nr vectors,
// 16-byte stores of nw vectors that depend on them, one vector of each per thread and step — over buffers that stay
// the same from launch to launch, so a working set that fits the Infinity Cache is served from it.  What a frame
// kernel's rate on a cache-resident scene is to be compared with (bench.py: roofline.cache_ceiling).
__global__ __launch_bounds__(YH_BLOCK) void k_stream_probe(const float4* __restrict__ src, float4* __restrict__ dst,
                                                           uint64_t nr, uint64_t nw) {
    const uint64_t stride = (uint64_t)gridDim.x * YH_BLOCK;
    const uint64_t top = nr > nw ? nr : nw;
    for (uint64_t i = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x; i < top; i += stride) {
        float4 v = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
        if (i < nr) v = src[i];
        if (i < nw) dst[i] = make_float4(v.x + 1.0f, v.y, v.z, v.w);
    }
}

// Fill plane q of a chunked array with a constant (scene creation: world = IDENTITY).
// sparse updates of per-slot words and boxes (yh_upload_parent_rows / _mesh_id_rows / _aabb_rows): value i goes to slot slots[i]
__global__ __launch_bounds__(YH_BLOCK) void k_scatter_u32(uint32_t* __restrict__ dst, const uint32_t* __restrict__ slots,
                                                          const uint32_t* __restrict__ vals, uint64_t m) {
    const uint64_t i = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (i < m) dst[slots[i]] = vals[i];
}
__global__ __launch_bounds__(YH_BLOCK) void k_scatter_aabb(float* __restrict__ bounds, const uint32_t* __restrict__ slots,
                                                           const float* __restrict__ mm /* m x {min.xyz, max.xyz} */, uint64_t m) {
    const uint64_t i = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (i >= m) return;
    const uint64_t off = eoff<6>(slots[i]);
#pragma unroll
    for (int q = 0; q < 6; ++q) *reinterpret_cast<float*>(reinterpret_cast<char*>(bounds) + off + q * 256) = mm[i * 6 + q];
}
__global__ __launch_bounds__(YH_BLOCK) void k_fill_plane(float* __restrict__ arr, uint32_t np, uint32_t q, float v,
                                                         uint64_t n) {
    const uint64_t e = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (e < n) arr[(e >> 6) * (np * 64ull) + q * 64ull + (e & 63)] = v;
}

// Host planes <-> chunked arrays (uploads / downloads; not on the per-frame path).
__global__ __launch_bounds__(YH_BLOCK) void k_scatter_plane(const float* __restrict__ src, float* __restrict__ arr,
                                                            uint32_t np, uint32_t q, uint64_t first, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint64_t e = first + i;
    arr[(e >> 6) * (np * 64ull) + q * 64ull + (e & 63)] = src[i];
}
__global__ __launch_bounds__(YH_BLOCK) void k_gather_plane(const float* __restrict__ arr, float* __restrict__ dst,
                                                           uint32_t np, uint32_t q, uint64_t first, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint64_t e = first + i;
    dst[i] = arr[(e >> 6) * (np * 64ull) + q * 64ull + (e & 63)];
}

// Stage 1 alone: batched Mat4.compose (math3d.zig:816-822) into 16 plain planes of `out_stride` floats.
__global__ __launch_bounds__(YH_BLOCK) void k_compose(const float* __restrict__ trs, float* __restrict__ out,
                                                      uint64_t out_stride, uint64_t first, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (i >= count) return;
    float t[10];
    load_row<10>(trs, (uint32_t)(first + i), t);
    const M4 l = m4_compose(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9]);
#pragma unroll
    for (int q = 0; q < 16; ++q) out[(uint64_t)q * out_stride + i] = l.v[q];
}

// Batched Mat4.decompose over a chunked matrix array: what the position() / rotation() / scale()
// accessors compute per call (components/transform.zig:14-28, 68-82: `decompose() catch Decomposed.IDENTITY`).
// out: 10 plain planes of out_stride floats; valid: 1 where the matrix decomposed, 0 where IDENTITY was substituted.
__global__ __launch_bounds__(YH_BLOCK) void k_decompose(const float* __restrict__ arr, float* __restrict__ out,
                                                        uint8_t* __restrict__ valid, uint64_t out_stride, uint64_t first,
                                                        uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (i >= count) return;
    M4 m;
    load_row<16>(arr, (uint32_t)(first + i), m.v);
    float d[10];
    const bool ok = m4_decompose(m, d);
    if (!ok) {
        d[0] = d[1] = d[2] = d[3] = d[4] = d[5] = 0.0f;
        d[6] = d[7] = d[8] = d[9] = 1.0f;
    }
#pragma unroll
    for (int q = 0; q < 10; ++q) out[(uint64_t)q * out_stride + i] = d[q];
    valid[i] = ok ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------
// flecs column import/export: 68-byte rows {f32 m[16]; u8 dirty; u8 pad[3]} (transform.zig:8-12) <-> chunked
// planes.  A workgroup moves 256 rows = 4352 dwords with coalesced dword accesses on the AoS side and
// transposes through LDS; the row stride of 17 dwords is odd, so both LDS phases are bank-conflict free.
// `slots` (optional): row r goes to / comes from entity slots[r] instead of first + r (sparse dirty rows).
__global__ __launch_bounds__(YH_BLOCK) void k_aos68_to_soa(const uint32_t* __restrict__ rows, uint64_t count,
                                                           float* __restrict__ arr, uint8_t* __restrict__ dirty,
                                                           uint64_t first, const uint32_t* __restrict__ slots) {
    __shared__ uint32_t tile[YH_BLOCK * 17];
    const uint64_t r0 = (uint64_t)blockIdx.x * YH_BLOCK;
    const uint64_t nrows = (count - r0 < YH_BLOCK) ? (count - r0) : YH_BLOCK;
    const uint64_t ndw = nrows * 17;
    for (uint32_t i = threadIdx.x; i < ndw; i += YH_BLOCK) tile[i] = rows[r0 * 17 + i];
    __syncthreads();
    if (threadIdx.x < nrows) {
        const uint32_t e = slots ? slots[r0 + threadIdx.x] : (uint32_t)(first + r0 + threadIdx.x);
        const uint64_t off = eoff<16>(e);
#pragma unroll
        for (int q = 0; q < 16; ++q) ste(arr, off, q, __uint_as_float(tile[threadIdx.x * 17 + q]));
        dirty[e] = (uint8_t)((tile[threadIdx.x * 17 + 16] & 0xFFu) != 0);
    }
}

__global__ __launch_bounds__(YH_BLOCK) void k_copy_u32(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ __launch_bounds__(YH_BLOCK) void k_soa_to_aos68(const float* __restrict__ arr,
                                                           const uint8_t* __restrict__ dirty, uint64_t first,
                                                           uint64_t count, uint32_t* __restrict__ rows,
                                                           const uint32_t* __restrict__ slots) {
    __shared__ uint32_t tile[YH_BLOCK * 17];
    const uint64_t r0 = (uint64_t)blockIdx.x * YH_BLOCK;
    const uint64_t nrows = (count - r0 < YH_BLOCK) ? (count - r0) : YH_BLOCK;
    if (threadIdx.x < nrows) {
        const uint32_t e = slots ? slots[first + r0 + threadIdx.x] : (uint32_t)(first + r0 + threadIdx.x);
        const uint64_t off = eoff<16>(e);
#pragma unroll
        for (int q = 0; q < 16; ++q) tile[threadIdx.x * 17 + q] = __float_as_uint(lde(arr, off, q));
        tile[threadIdx.x * 17 + 16] = dirty ? (uint32_t)(dirty[e] != 0) : 0u;  // bool + 3 zero pad bytes
    }
    __syncthreads();
    const uint64_t ndw = nrows * 17;
    for (uint32_t i = threadIdx.x; i < ndw; i += YH_BLOCK) rows[r0 * 17 + i] = tile[i];
}
#endif  // YH_K_SCENE

// Multi-GPU, no-host-sync gather.  What crosses xGMI is the visibility WORDS, not the lists: one bit per entity
// instead of four bytes per visible entity (32x fewer bytes than lists padded to capacity, and a fixed size, so a
// single ncclAllGather needs no counts first).  Every rank then expands all R masks itself, exactly as k_count /
// k_expand do for its own: gmask = [rank][frustum][words] (rows zero-padded to the longest shard), output = the R
// lists back to back in rank order with each rank's id_base added (ascending global ids when shards own ascending id
// ranges).  grid = (tiles of 64 or 256 words, frusta, ranks).
struct GatherArgs {
    const uint64_t* gmask;    // [R][F][words]
    const uint32_t* peer;     // [R][4]: n_chunks, id_base, frusta, -
    uint32_t* gtile;          // [F][R][n_tiles]
    uint32_t* cum;            // [R][YH_K_MAX_FRUSTA]: entries of ranks 0..r in frustum f (running total)
    uint32_t* out[YH_K_MAX_FRUSTA];
    uint32_t words, n_tiles, n_ranks, n_frusta;
    uint32_t tile_words;  // words per tile: 64, or 256 for long rows (as k_expand)
};

#ifdef YH_K_GATHER
__global__ __launch_bounds__(YH_BLOCK) void k_gather_count(const GatherArgs g) {
    __shared__ uint32_t s_red[4];
    const uint32_t t = blockIdx.x, f = blockIdx.y, r = blockIdx.z;
    const uint32_t c = t * g.tile_words + threadIdx.x;
    const uint64_t m = (threadIdx.x < g.tile_words && c < g.words) ? g.gmask[((uint64_t)r * g.n_frusta + f) * g.words + c] : 0ull;
    const uint32_t total = block_sum_256((uint32_t)__popcll(m), s_red);
    if (threadIdx.x == 0) g.gtile[((uint64_t)f * g.n_ranks + r) * g.n_tiles + t] = total;
}

__global__ __launch_bounds__(YH_BLOCK) void k_gather_expand(const GatherArgs g) {
    __shared__ uint32_t s_red[4];
    __shared__ uint32_t s_wave[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t t = blockIdx.x, f = blockIdx.y, r = blockIdx.z;
    const uint32_t wpw = g.tile_words >> 2;  // every wave owns a contiguous run of the tile's words (as expand_tile)
    const uint32_t cw = t * g.tile_words + wv * wpw;
    const uint64_t m = (lane < wpw && cw + lane < g.words) ? g.gmask[((uint64_t)r * g.n_frusta + f) * g.words + cw + lane] : 0ull;
    // base = entries of the ranks before this one and of this rank's tiles before this one
    const uint32_t* __restrict__ cnt = g.gtile + (uint64_t)f * g.n_ranks * g.n_tiles;
    const uint32_t before_tiles = r * g.n_tiles + t;
    uint32_t part = 0;
    for (uint32_t i = tid; i < before_tiles; i += YH_BLOCK) part += cnt[i];
    const uint32_t base = block_sum_256(part, s_red);
    uint32_t run = (uint32_t)__popcll(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) run += __shfl_xor(run, o);
    run = __builtin_amdgcn_readfirstlane(run);  // the same in every lane: say so, the branches below are then scalar
    if (lane == 0) s_wave[wv] = run;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wv; ++w) before += s_wave[w];
    const uint32_t own = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (tid == 0 && t + 1 == g.n_tiles) g.cum[r * YH_K_MAX_FRUSTA + f] = base + own;
    if (run == 0) return;  // wave-uniform
    const uint32_t n_words = min(wpw, g.words > cw ? g.words - cw : 0u);
    if (n_words == 0) return;
    const uint32_t pos = __builtin_amdgcn_readfirstlane(base + before);
    expand_run(g.out[f] + pos, (uint32_t)m, (uint32_t)(m >> 32), n_words, (cw << 6) + lane + g.peer[r * 4 + 1]);
}
#endif  // YH_K_GATHER

#ifdef YH_K_CONSUMERS
// ---------------------------------------------------------------------------------------------------
// Draw-list consumer: what the reference's draw loop does on the CPU for every object it is handed
// (GAL/vulkan/RenderApi.zig:1397-1404: object_data_arr[batch_offset + index] = GPUObjectData{model_matrix};
// :1441 vkCmdDraw(vertices_count, 1, 0, batch_offset + index)), here for the visible entities of one frustum in list
// order: 64-byte model matrices (transposed through LDS so the stores are coalesced) and one
// VkDrawIndirectCommand-shaped record each.  The list length is read on the device: no host sync before the launch.
__global__ __launch_bounds__(YH_BLOCK) void k_build_objects(const float* __restrict__ world, const uint32_t* __restrict__ list,
                                                            const uint32_t* __restrict__ count_ptr, uint64_t cap,
                                                            uint32_t id_base, const uint32_t* __restrict__ vcount,
                                                            const uint32_t* __restrict__ mesh_id, uint32_t batch_offset,
                                                            uint32_t* __restrict__ obj, uint4* __restrict__ cmds) {
    __shared__ uint32_t tile[YH_BLOCK * 17];
    uint64_t count = *count_ptr;
    if (count > cap) count = cap;
    const uint64_t r0 = (uint64_t)blockIdx.x * YH_BLOCK;
    if (r0 >= count) return;
    const uint64_t nrows = (count - r0 < YH_BLOCK) ? (count - r0) : YH_BLOCK;
    if (threadIdx.x < nrows) {
        const uint32_t e = list[r0 + threadIdx.x] - id_base;
        const uint64_t off = eoff<16>(e);
#pragma unroll
        for (int q = 0; q < 16; ++q) tile[threadIdx.x * 17 + q] = __float_as_uint(lde(world, off, q));
        const uint32_t vc = vcount ? vcount[mesh_id ? mesh_id[e] : e] : 0u;
        cmds[r0 + threadIdx.x] = make_uint4(vc, 1u, 0u, batch_offset + (uint32_t)(r0 + threadIdx.x));
    }
    __syncthreads();
    const uint32_t ndw = (uint32_t)nrows * 16u;
    for (uint32_t i = threadIdx.x; i < ndw; i += YH_BLOCK) obj[r0 * 16 + i] = tile[(i >> 4) * 17 + (i & 15u)];
}
#endif  // YH_K_CONSUMERS

// ---------------------------------------------------------------------------------------------------
// Per-mesh bucketing of a draw list.  The reference's draw loop rebinds the vertex buffer
// whenever "mesh != previous mesh" (RenderApi.zig:1434): handing it the visible entities grouped by mesh makes that
// one bind per mesh.  A stable counting sort by mesh id: n_slices workgroups own one contiguous slice of the
// (ascending) list each; pass 1 histograms the slices, two small launches turn hist[mesh][slice] into output offsets,
// pass 2 walks every slice again, 256 entries at a time: the tile is sorted by mesh inside LDS (ballot ranks per
// wave, wave and mesh offsets by a scan) together with its model matrices, and leaves in bucket order, so entity id,
// matrix and draw record go out in runs of consecutive positions — inside a bucket the entities stay ascending.
#define YH_BUCKET_MAX_MESHES 256u

struct BucketArgs {
    const uint32_t* list;       // the frustum's draw list (global ids, ascending)
    const uint32_t* count_ptr;  // its length, on the device
    uint64_t cap;
    uint32_t id_base;
    const uint32_t* mesh_id;
    uint32_t n_meshes;
    uint32_t n_slices; // slices the list is cut into (a multiple of 256, chosen by the host)
    uint32_t* hist;    // [n_meshes][n_slices] counts, then (k_bucket_rows) exclusive offsets inside the row
    uint32_t* totals;  // [n_meshes]
    uint32_t* segtot;  // [n_meshes][n_seg] totals of the row's segments of YH_BUCKET_SEG slices, then their exclusive prefix
    uint32_t n_seg;
    uint32_t* starts;  // [n_meshes] first output position of each bucket
    uint32_t* buckets; // [n_meshes] x {mesh, first, count}
    const float* world;
    const uint32_t* vcount;
    uint32_t batch_offset;
    uint32_t* out_list;
    uint32_t* obj;
    uint4* cmds;
};

__device__ __forceinline__ void bucket_slice(const BucketArgs& a, uint32_t slice, uint64_t& lo, uint64_t& hi) {
    uint64_t count = *a.count_ptr;
    if (count > a.cap) count = a.cap;
    uint64_t per = (count + a.n_slices - 1) / a.n_slices;
    per = (per + (YH_BLOCK - 1)) & ~(uint64_t)(YH_BLOCK - 1);
    lo = (uint64_t)slice * per;
    hi = lo + per;
    if (lo > count) lo = count;
    if (hi > count) hi = count;
}

#ifdef YH_K_CONSUMERS
__global__ __launch_bounds__(YH_BLOCK) void k_bucket_hist(BucketArgs a) {
    __shared__ uint32_t h[YH_BUCKET_MAX_MESHES];
    for (uint32_t m = threadIdx.x; m < a.n_meshes; m += YH_BLOCK) h[m] = 0;
    __syncthreads();
    uint64_t lo, hi;
    bucket_slice(a, blockIdx.x, lo, hi);
    for (uint64_t i = lo + threadIdx.x; i < hi; i += YH_BLOCK) atomicAdd(&h[a.mesh_id[a.list[i] - a.id_base]], 1u);
    __syncthreads();
    for (uint32_t m = threadIdx.x; m < a.n_meshes; m += YH_BLOCK) a.hist[(uint64_t)m * a.n_slices + blockIdx.x] = h[m];
}
#endif  // YH_K_CONSUMERS

// Inclusive scan of one value per thread over the workgroup (256 threads): inside each wave by lane shuffles, the four
// wave totals through LDS (two barriers instead of the seventeen of a scan that steps through LDS; ends with a barrier).
__device__ __forceinline__ uint32_t block_scan_256(uint32_t v, uint32_t* part) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63u) part[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += part[w];
    __syncthreads();  // (part is the caller's again)
    return before + inc;
}

// (the same between LDS-only barriers: k_bucket_scatter)
__device__ __forceinline__ uint32_t block_scan_256_lds(uint32_t v, uint32_t* part) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63u) part[wave] = inc;
    lds_barrier();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += part[w];
    lds_barrier();
    return before + inc;
}

#ifdef YH_K_CONSUMERS
// hist[key][slice] -> exclusive offsets inside the key's row, in three small launches that use the whole chip whatever
// the number of keys (one workgroup per key left 16 keys x 32768 slices to 16 workgroups: 59 us; 124 us at 4096 keys):
// k_bucket_segsum   one workgroup per (key, segment of YH_BUCKET_SEG slices): the segment's total;
// k_bucket_starts   one workgroup: per key the exclusive prefix over its segments and its total, then the exclusive scan
//                   of the totals = the buckets' first positions, and the bucket table the renderer reads;
// k_bucket_rows     one workgroup per (key, segment) again: exclusive scan inside the segment + the segment's prefix.
// Eight consecutive slices per thread (two 16-byte loads; n_slices is a multiple of 256, so a thread's eight are all
// inside the row or all outside).
#define YH_BUCKET_SEG 2048u
__global__ __launch_bounds__(YH_BLOCK) void k_bucket_segsum(BucketArgs a) {
    __shared__ uint32_t s_red[4];
    const uint32_t key = blockIdx.x / a.n_seg, seg = blockIdx.x - key * a.n_seg;
    const uint32_t s0 = seg * YH_BUCKET_SEG + threadIdx.x * 8u;
    uint32_t sum = 0;
    if (s0 < a.n_slices) {
        const uint4* p = reinterpret_cast<const uint4*>(a.hist + (uint64_t)key * a.n_slices + s0);
        const uint4 u = p[0], v = p[1];
        sum = u.x + u.y + u.z + u.w + v.x + v.y + v.z + v.w;
    }
    const uint32_t total = block_sum_256(sum, s_red);
    if (threadIdx.x == 0) a.segtot[blockIdx.x] = total;
}

__global__ __launch_bounds__(YH_BLOCK) void k_bucket_starts(BucketArgs a) {
    __shared__ uint32_t part[YH_BLOCK];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t m0 = 0; m0 < a.n_meshes; m0 += YH_BLOCK) {
        const uint32_t m = m0 + threadIdx.x;
        uint32_t total = 0;
        if (m < a.n_meshes)
            for (uint32_t g = 0; g < a.n_seg; ++g) {
                const uint32_t v = a.segtot[(uint64_t)m * a.n_seg + g];
                a.segtot[(uint64_t)m * a.n_seg + g] = total;
                total += v;
            }
        const uint32_t inc = block_scan_256(total, part);
        const uint32_t before = carry;
        if (m < a.n_meshes) {
            const uint32_t first = before + inc - total;
            a.totals[m] = total;
            a.starts[m] = first;
            a.buckets[3 * m + 0] = m;
            a.buckets[3 * m + 1] = first;
            a.buckets[3 * m + 2] = total;
        }
        __syncthreads();
        if (threadIdx.x == YH_BLOCK - 1) carry = before + inc;
        __syncthreads();
    }
}

__global__ __launch_bounds__(YH_BLOCK) void k_bucket_rows(BucketArgs a) {
    __shared__ uint32_t part[YH_BLOCK];
    const uint32_t key = blockIdx.x / a.n_seg, seg = blockIdx.x - key * a.n_seg;
    const uint32_t s0 = seg * YH_BUCKET_SEG + threadIdx.x * 8u;
    const bool in = s0 < a.n_slices;
    uint4* p = reinterpret_cast<uint4*>(a.hist + (uint64_t)key * a.n_slices + (in ? s0 : 0u));
    uint4 u = make_uint4(0, 0, 0, 0), v = u;
    if (in) u = p[0], v = p[1];
    const uint32_t sum = u.x + u.y + u.z + u.w + v.x + v.y + v.z + v.w;
    const uint32_t inc = block_scan_256(sum, part);
    uint32_t run = a.segtot[blockIdx.x] + inc - sum;
    if (in) {
        uint4 o, q;
        o.x = run, run += u.x;
        o.y = run, run += u.y;
        o.z = run, run += u.z;
        o.w = run, run += u.w;
        q.x = run, run += v.x;
        q.y = run, run += v.y;
        q.z = run, run += v.z;
        q.w = run;
        p[0] = o, p[1] = q;
    }
}

__global__ __launch_bounds__(YH_BLOCK) void k_bucket_scatter(BucketArgs a) {
    __shared__ uint32_t base[YH_BUCKET_MAX_MESHES];        // next output position of each bucket, for this slice
    __shared__ uint32_t wcnt[4][YH_BUCKET_MAX_MESHES];     // entries of each mesh in each wave's 64, then wave offsets
    __shared__ uint32_t lstart[YH_BUCKET_MAX_MESHES];      // first tile-local sorted index of each mesh
    __shared__ uint32_t s_vcount[YH_BUCKET_MAX_MESHES];    // the vertex-count table (no gather from memory inside a tile)
    __shared__ uint32_t part[YH_BLOCK];
    __shared__ uint32_t s_pos[YH_BLOCK], s_id[YH_BLOCK], s_vc[YH_BLOCK];  // the tile in bucket order
    __shared__ uint32_t rows[YH_BLOCK * 17];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    for (uint32_t m = tid; m < a.n_meshes; m += YH_BLOCK) {
        base[m] = a.starts[m] + a.hist[(uint64_t)m * a.n_slices + blockIdx.x];
        s_vcount[m] = a.vcount ? a.vcount[m] : 0u;
    }
    uint64_t lo, hi;
    bucket_slice(a, blockIdx.x, lo, hi);
    // Software pipeline over the slice's tiles with ONE set of row registers and no register copies: a tile's rows are
    // loaded right after the tile before it has been staged in LDS (the registers are free again) and BEFORE that tile's
    // stores are issued, its list entries one tile earlier still.  Everything in the loop body is straight-line code
    // with a fixed number of loads and stores (the sixteen row stores are unrolled and predicated, not a loop), so the
    // compiler waits for exactly the loads it needs (s_waitcnt vmcnt(n), n = what was issued after them) and a tile's
    // scattered stores stay in flight while the next tile is sorted.  (Before: a register copy at the loop's end and a
    // store loop of unknown length each cost an s_waitcnt vmcnt(0) per tile, i.e. every tile waited for the
    // acknowledgement of its own stores.  It bought 1 %: the pass is bound elsewhere, profiles/r03/bucket_slices.txt.)
    // (No load or store below sits behind a branch: a lane past the slice's end re-reads the slice's last entry and
    // re-writes the tile's last row - the same value to the same address - so the counts stay fixed on every path.)
    if (lo >= hi) return;  // (whole workgroup)
    const uint64_t last = hi - 1;
    uint32_t id_c = a.list[min(lo + tid, last)];
    uint32_t id_n = a.list[min(lo + YH_BLOCK + tid, last)];
    uint32_t m_c = a.mesh_id[id_c - a.id_base];
    float w[16];
    {
        const uint64_t off = eoff<16>(id_c - a.id_base);
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q] = lde(a.world, off, q);
    }
    auto tile = [&](const uint64_t i0) __attribute__((always_inline)) {
        const uint32_t n_tile = (uint32_t)((hi - i0 < YH_BLOCK) ? (hi - i0) : YH_BLOCK);
        for (uint32_t m = tid; m < a.n_meshes; m += YH_BLOCK) wcnt[0][m] = wcnt[1][m] = wcnt[2][m] = wcnt[3][m] = 0;
        lds_barrier();
        const bool active = tid < n_tile;
        const uint32_t id = id_c;
        const uint32_t m = active ? m_c : 0xFFFFFFFFu;
        // rank among the wave's entries of the same mesh (entries are in list order lane by lane, wave by wave)
        uint32_t rank = 0;
        uint64_t todo = __ballot(active);
        while (todo) {  // one turn per distinct mesh among the wave's 64 entries
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const uint32_t lm = __shfl(m, leader);
            const uint64_t same = __ballot(m == lm);
            if (m == lm) rank = lane_rank(same);
            if ((int)lane == leader) wcnt[wave][lm] = (uint32_t)__popcll(same);
            todo &= ~same;
        }
        lds_barrier();
        // per mesh: counts of the four waves -> exclusive wave offsets, tile total -> tile-local start by a scan
        uint32_t tile_cnt = 0;
        if (tid < a.n_meshes) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const uint32_t c = wcnt[v][tid];
                wcnt[v][tid] = tile_cnt;
                tile_cnt += c;
            }
        }
        const uint32_t inc = block_scan_256_lds(tile_cnt, part);  // (ends with a barrier)
        if (tid < a.n_meshes) lstart[tid] = inc - tile_cnt;
        lds_barrier();
        if (active) {
            const uint32_t in_bucket = wcnt[wave][m] + rank;
            const uint32_t ls = lstart[m] + in_bucket;
            s_pos[ls] = base[m] + in_bucket;
            s_id[ls] = id;
            s_vc[ls] = s_vcount[m];
#pragma unroll
            for (int q = 0; q < 16; ++q) rows[ls * 17 + q] = __float_as_uint(w[q]);
        }
        lds_barrier();
        if (tid < a.n_meshes) base[tid] += tile_cnt;
        // the next tile's mesh ids and rows (its list entries arrived a tile ago), the list entries of the one after
        id_c = id_n;
        id_n = a.list[min(i0 + 2 * YH_BLOCK + tid, last)];
        m_c = a.mesh_id[id_c - a.id_base];
        {
            const uint64_t off = eoff<16>(id_c - a.id_base);
#pragma unroll
            for (int q = 0; q < 16; ++q) w[q] = lde(a.world, off, q);
        }
        // this tile leaves in bucket order
        {
            const uint32_t t = min(tid, n_tile - 1u);
            const uint32_t pos = s_pos[t];
            a.out_list[pos] = s_id[t];
            a.cmds[pos] = make_uint4(s_vc[t], 1u, 0u, a.batch_offset + pos);
        }
        // (streaming stores: the rows are written once and read by another engine; 2.12 -> 1.94 ms at 64M / 16 meshes)
        // thread t writes dword t & 15 of rows t / 16 + 16 j: a wave's store is four whole 64-byte rows
        {
            const uint32_t r0 = tid >> 4, c = tid & 15u;
            uint32_t v[16], p[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t r = min(r0 + 16u * j, n_tile - 1u);
                v[j] = rows[r * 17u + c];
                p[j] = s_pos[r];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) __builtin_nontemporal_store(v[j], &a.obj[(uint64_t)p[j] * 16 + c]);
        }
        lds_barrier();
    };
    // The first tile stands outside the loop: nothing has been stored before it, so a wait that leaves "the stores issued
    // after these loads" outstanding would be wrong there, and one loop for both would have to wait for everything.
    tile(lo);
    for (uint64_t i0 = lo + YH_BLOCK; i0 < hi; i0 += YH_BLOCK) tile(i0);  // uniform trip count for the workgroup
}
#endif  // YH_K_CONSUMERS

// ---- the same for MANY keys (257 .. YH_BUCKET_MAX_KEYS: one key per (material, mesh) pair of a scene) ------------------
// With thousands of keys a 256-entry tile holds a key once or not at all, so there is nothing to sort inside a tile and
// no room for per-key arrays in LDS.  Here every WAVE owns one contiguous slice of the list (n_slices of them: a power
// of two chosen by the host so that keys x slices stays within 16M counters) and hist[key][slice] lives in global
// memory only: pass 1 counts into it, k_bucket_rows / k_bucket_starts turn it into exclusive offsets, pass 2 walks
// the slices again and, per 64 entries, lets the first lane of every distinct key fetch-and-add that key's running
// offset of this slice (one atomic per distinct key and 64 entries, all of a wave's in flight together; the returned
// values order the wave's visits of a key, lane ranks order the entries inside a visit: stable).  Matrices go through a
// wave-private LDS transpose and leave as whole 64-byte rows.
#define YH_BUCKET_MAX_KEYS 65536u
__device__ __forceinline__ void bucket_wave_slice(const BucketArgs& a, uint32_t slice, uint64_t& lo, uint64_t& hi) {
    uint64_t count = *a.count_ptr;
    if (count > a.cap) count = a.cap;
    uint64_t per = (count + a.n_slices - 1) / a.n_slices;
    per = (per + 63u) & ~63ull;
    lo = (uint64_t)slice * per;
    hi = lo + per;
    if (lo > count) lo = count;
    if (hi > count) hi = count;
}

// rank of this lane among the wave's active lanes with the same key (lane order), their number, and whether this lane
// is the first of them: one turn per distinct key
__device__ __forceinline__ void match_keys(bool active, uint32_t key, uint32_t& rank, uint32_t& same_count, bool& first, uint32_t& leader_lane) {
    rank = 0;
    same_count = 0;
    first = false;
    leader_lane = threadIdx.x & 63u;
    uint64_t todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint32_t lk = __shfl(key, leader);
        const uint64_t same = __ballot(active && key == lk);
        if (active && key == lk) {
            rank = lane_rank(same);
            same_count = (uint32_t)__popcll(same);
            first = (int)(threadIdx.x & 63u) == leader;
            leader_lane = (uint32_t)leader;
        }
        todo &= ~same;
    }
}

#ifdef YH_K_CONSUMERS
__global__ __launch_bounds__(YH_BLOCK) void k_bucketw_hist(BucketArgs a) {
    const uint32_t slice = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    uint64_t lo, hi;
    bucket_wave_slice(a, slice, lo, hi);
    for (uint64_t i0 = lo; i0 < hi; i0 += 64u) {  // wave-uniform
        const bool active = i0 + lane < hi;
        const uint32_t key = active ? a.mesh_id[a.list[i0 + lane] - a.id_base] : 0u;
        uint32_t rank, cnt, leader;
        bool first;
        match_keys(active, key, rank, cnt, first, leader);
        if (first) atomicAdd(&a.hist[(uint64_t)key * a.n_slices + slice], cnt);  // (this wave's own counter: ordering, not contention)
    }
}

__global__ __launch_bounds__(YH_BLOCK) void k_bucketw_scatter(BucketArgs a) {
    __shared__ uint32_t rows[4][64 * 17];  // one transpose area per wave: nothing is shared between waves, no workgroup barrier
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t slice = blockIdx.x * 4u + wave;
    uint32_t* __restrict__ mine = rows[wave];
    uint64_t lo, hi;
    bucket_wave_slice(a, slice, lo, hi);
    uint32_t id_n = lo + lane < hi ? a.list[lo + lane] : a.id_base;
    for (uint64_t i0 = lo; i0 < hi; i0 += 64u) {  // wave-uniform
        const bool active = i0 + lane < hi;
        const uint32_t id = id_n;
        id_n = i0 + 64u + lane < hi ? a.list[i0 + 64u + lane] : a.id_base;  // the next entries, on their way
        const uint32_t e = id - a.id_base;
        const uint32_t key = active ? a.mesh_id[e] : 0u;
        float w[16];
        {
            const uint64_t off = eoff<16>(e);  // (an inactive lane reads entity 0's row: in bounds)
#pragma unroll
            for (int q = 0; q < 16; ++q) w[q] = lde(a.world, off, q);
        }
        uint32_t rank, cnt, leader;
        bool first;
        match_keys(active, key, rank, cnt, first, leader);
        uint32_t base = 0;
        if (first) base = atomicAdd(&a.hist[(uint64_t)key * a.n_slices + slice], cnt);  // running offset of (key, slice)
        base = (uint32_t)__shfl((int)base, (int)leader);  // every lane takes its key's base from the first lane of that key
        const uint32_t pos = active ? a.starts[key] + base + rank : 0u;
        if (active) {
            a.out_list[pos] = id;
            a.cmds[pos] = make_uint4(a.vcount ? a.vcount[key] : 0u, 1u, 0u, a.batch_offset + pos);
        }
        __builtin_amdgcn_wave_barrier();  // (the wave's earlier reads of `mine` stay ahead of these writes)
#pragma unroll
        for (int q = 0; q < 16; ++q) mine[lane * 17 + q] = __float_as_uint(w[q]);
        __builtin_amdgcn_wave_barrier();
        const uint32_t n_tile = (uint32_t)(hi - i0 < 64u ? hi - i0 : 64u);
        // lanes 16 r' .. 16 r' + 15 write row r = 4 k + r' as one 64-byte piece
#pragma unroll 4
        for (uint32_t k = 0; k < 16u; ++k) {
            const uint32_t r = 4u * k + (lane >> 4);
            const uint32_t rp = (uint32_t)__shfl((int)pos, (int)r);
            if (r < n_tile) a.obj[(uint64_t)rp * 16 + (lane & 15u)] = mine[r * 17 + (lane & 15u)];
        }
    }
}
#endif  // YH_K_CONSUMERS

#ifdef YH_K_SCENE
// ---------------------------------------------------------------------------------------------------
// Synthetic scene generator: TRS + bounds from the counter-based hash, bit-identical to
// the oracle's host generator.  Level 0 (local index < n_level0) draws from the wide ranges.
__global__ __launch_bounds__(YH_BLOCK) void k_synth(float* __restrict__ trs, float* __restrict__ bounds,
                                                    uint32_t* __restrict__ mesh_id, uint32_t n_meshes, uint64_t n,
                                                    uint64_t n_level0, uint64_t seed, uint64_t index_base,
                                                    const uint32_t* __restrict__ global_of_local) {
    const uint64_t e = (uint64_t)blockIdx.x * YH_BLOCK + threadIdx.x;
    if (e >= n) return;
    // a shard draws entity e's values from its GLOBAL index, so that the shards of one scene hold the scene's data
    const uint64_t g = index_base + (global_of_local ? (uint64_t)global_of_local[e] : e);
    const bool top = e < n_level0;
    const float tr = top ? 100.0f : 5.0f;
    float t[10];
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c] = yh_synth_uniform(seed, (uint32_t)c, g, -tr, tr);
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
    for (int c = 0; c < 4; ++c) t[3 + c] = q[c];
    const float slo = top ? 0.5f : 0.8f, shi = top ? 2.0f : 1.25f;
#pragma unroll
    for (int c = 0; c < 3; ++c) t[7 + c] = yh_synth_uniform(seed, 7u + (uint32_t)c, g, slo, shi);
    store_row<10>(trs, (uint32_t)e, t);
    if (bounds) {
        float b[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float h = yh_synth_uniform(seed, 10u + (uint32_t)c, g, 0.25f, 2.0f);
            b[c] = -h;
            b[3 + c] = h;
        }
        store_row<6>(bounds, (uint32_t)e, b);
    }
    if (mesh_id) mesh_id[e] = (uint32_t)(yh_synth_raw(seed, 13, g) >> 32) % n_meshes;
}
#endif  // YH_K_SCENE
