// Host-side sanitizer run of the C ABI (include/yume_hip.h): this file and the library's translation units (yume_amd/csrc/yh_*.hip) are compiled
// into ONE executable with AddressSanitizer + UBSan on the host code only (make -C yume_amd/csrc asan_driver; the
// device code is unchanged), and the calls below walk the library's host paths: scene set-up, uploads, frames of both
// scene kinds, lists, consumers, buckets, change tracking, decompose, the gather self-test, a one-rank RCCL gather, error returns.
// Results are only sanity-checked here (the parity tests own correctness); what matters is that the sanitizers stay
// silent.  Run on a GPU box:  ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 ./tools/bin/abi_asan_driver
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "yume_hip.h"

#define CK(call)                                                                            \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != YH_OK) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, yh_last_error(ctx)); \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)
#define EXPECT(cond)                                                     \
    do {                                                                 \
        if (!(cond)) {                                                   \
            fprintf(stderr, "%s:%d expected %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                    \
        }                                                                \
    } while (0)

static void tree(uint32_t n, uint32_t levels, std::vector<uint32_t>& lo, std::vector<uint32_t>& parent) {
    lo.assign(levels + 1, 0);
    for (uint32_t l = 0; l < levels; ++l) lo[l + 1] = (uint32_t)((uint64_t)n * (l + 1) / levels);
    parent.assign(n, YH_NO_PARENT);
    for (uint32_t l = 1; l < levels; ++l) {
        const uint32_t pb = lo[l - 1], pc = lo[l] - lo[l - 1], cb = lo[l], cc = lo[l + 1] - lo[l];
        for (uint32_t i = 0; i < cc; ++i) parent[cb + i] = pb + (uint32_t)((uint64_t)i * pc / cc);  // sorted by parent
    }
}

int main() {
    yh_ctx* ctx = nullptr;
    if (yh_init(0, &ctx) != YH_OK) {
        fprintf(stderr, "yh_init failed: %s\n", yh_last_error(nullptr));
        return 1;
    }
    float vp[16], proj[16], view[16], planes[3 * 24];
    const float eye[3] = {0, 40, 150}, at[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    yh_perspective(1.3089969f, 2.0f, 0.1f, 200.0f, proj);
    yh_look_at(eye, at, up, view);
    yh_mat4_mul(proj, view, vp);
    for (int f = 0; f < 3; ++f) yh_extract_planes(vp, planes + 24 * f);

    // ---- TRS tree scene, per-entity bounds, 3 frusta, ragged size ----
    {
        const uint32_t n = 70001, levels = 5;
        std::vector<uint32_t> lo, parent;
        tree(n, levels, lo, parent);
        yh_scene_desc d;
        memset(&d, 0, sizeof d);
        d.n_entities = n;
        d.n_levels = levels;
        d.n_frusta = 3;
        d.local_kind = YH_LOCAL_TRS;
        d.bounds_kind = YH_BOUNDS_PER_ENTITY;
        yh_scene* s = nullptr;
        CK(yh_scene_create(ctx, &d, &s));
        CK(yh_upload_hierarchy(s, parent.data(), lo.data()));
        yh_synth_desc sd;
        memset(&sd, 0, sizeof sd);
        sd.seed = 0x59554D45ull;
        CK(yh_scene_fill_synthetic(s, &sd));
        for (uint32_t g : {0u, 1u, 3u, 8u}) {
            CK(yh_scene_set_grouping(s, g));
            CK(yh_frame(s, planes, 3, YH_FRAME_ALL_DIRTY));
        }
        const uint32_t plan[2] = {1, 4};
        CK(yh_scene_set_plan(s, 2, plan));
        CK(yh_scene_set_kernel_timing(s, 1));
        CK(yh_frame(s, planes, 2, YH_FRAME_ALL_DIRTY | YH_FRAME_ASYNC));
        CK(yh_wait(ctx));
        uint32_t quads[4 * 64], ng = 0, samples[64];
        float ms[64];
        CK(yh_scene_get_groups(s, quads, 64, &ng));
        EXPECT(ng == 2);
        CK(yh_scene_get_group_timings(s, ms, samples, 64));
        CK(yh_scene_set_kernel_timing(s, 0));
        uint32_t counts[8], bc[8];
        CK(yh_get_counts(s, counts, bc));
        EXPECT(counts[0] == counts[1] && counts[0] > 0 && counts[0] <= n);
        std::vector<uint32_t> list(counts[0]);
        uint64_t wr = 0;
        CK(yh_get_draw_list(s, 0, list.data(), list.size(), &wr));
        EXPECT(wr == counts[0]);
        for (size_t i = 1; i < list.size(); ++i) EXPECT(list[i] > list[i - 1]);
        EXPECT(yh_get_draw_list(s, 0, list.data(), 3, &wr) == YH_ERR_CAPACITY && wr == counts[0]);
        // cull what the device holds: the same lists from one launch, deferred expansion of the frame before flushed
        CK(yh_frame(s, planes, 2, YH_FRAME_ALL_DIRTY | YH_FRAME_ASYNC));
        CK(yh_frame(s, planes, 1, YH_FRAME_CULL_ONLY | YH_FRAME_ASYNC));
        CK(yh_frame(s, planes, 3, YH_FRAME_CULL_ONLY));
        uint32_t counts2[8];
        CK(yh_get_counts(s, counts2, nullptr));
        EXPECT(counts2[0] == counts[0]);
        EXPECT(yh_frame(s, planes, 1, YH_FRAME_CULL_ONLY | YH_FRAME_ALL_DIRTY) == YH_ERR_BAD_ARG);
        CK(yh_frame_static(s, planes, 3, YH_FRAME_CULL_ONLY));
        CK(yh_get_counts(s, counts2, nullptr));
        EXPECT(counts2[0] == counts[0] && counts2[1] == counts[1]);
        // world rows out, both formats
        std::vector<float> soa(16 * (size_t)n);
        float* pl[16];
        for (int q = 0; q < 16; ++q) pl[q] = soa.data() + (size_t)q * n;
        CK(yh_download_world_soa(s, pl, 0, n));
        std::vector<unsigned char> col((size_t)n * 68);
        CK(yh_download_world_aos68(s, col.data(), 0, n));
        float m0[16];
        memcpy(m0, col.data() + 68 * 777, 64);
        for (int q = 0; q < 16; ++q) EXPECT(m0[q] == soa[(size_t)q * n + 777]);
        CK(yh_download_world_aos68(s, col.data(), 5, 1000));
        {   // boxes of single slots
            const uint32_t sl[2] = {3, n - 1}, bad[1] = {n};
            const float mm[12] = {-1, -1, -1, 1, 1, 1, -2, -2, -2, 2, 2, 2};
            CK(yh_upload_aabb_rows(s, sl, mm, 2));
            EXPECT(yh_upload_aabb_rows(s, bad, mm, 1) == YH_ERR_BAD_ARG);
        }
        // consumers
        std::vector<uint32_t> vc(n, 36);
        CK(yh_upload_vertex_counts(s, vc.data(), n));
        CK(yh_build_object_data(s, 1, 10));
        std::vector<float> mats((size_t)counts[1] * 16);
        std::vector<yh_draw_cmd> cmds(counts[1]);
        CK(yh_get_object_data(s, 1, mats.data(), cmds.data(), counts[1], &wr));
        EXPECT(wr == counts[1] && cmds[0].first_instance == 10 && cmds[0].vertex_count == 36);
        EXPECT(yh_build_mesh_buckets(s, 0, 0) == YH_ERR_STATE);
        // decompose
        std::vector<float> trs(10 * (size_t)n);
        float* tp[10];
        for (int q = 0; q < 10; ++q) tp[q] = trs.data() + (size_t)q * n;
        std::vector<uint8_t> valid(n);
        CK(yh_decompose(s, 0, tp, valid.data(), 0, n));
        CK(yh_decompose(s, 0, tp, nullptr, 100, 33));
        // gather self-test
        CK(yh_frame(s, planes, 3, YH_FRAME_ALL_DIRTY));
        CK(yh_gather_selftest(s, 5));
        uint32_t gc[5 * 3];
        CK(yh_get_gathered_counts(s, gc));
        CK(yh_get_counts(s, counts, bc));
        EXPECT(gc[0] == counts[0] && gc[4 * 3 + 2] == counts[2]);
        std::vector<uint32_t> gl((size_t)counts[0] * 5);
        CK(yh_get_gathered_draw_list(s, 0, gl.data(), gl.size(), &wr));
        EXPECT(wr == (uint64_t)counts[0] * 5 && gl[counts[0]] == gl[0] + n);
        // argument errors
        EXPECT(yh_frame(s, planes, 4, YH_FRAME_ALL_DIRTY) == YH_ERR_BAD_ARG);
        EXPECT(yh_frame(s, planes, 1, 0) == YH_ERR_UNSUPPORTED);
        EXPECT(yh_download_world_aos68(s, col.data(), n - 1, 5) != YH_OK);
        yh_scene_destroy(s);
    }

    // ---- flat instanced scene: mesh table, buckets, small draw capacity ----
    {
        const uint32_t n = 100003, nm = 16;
        yh_scene_desc d;
        memset(&d, 0, sizeof d);
        d.n_entities = n;
        d.n_levels = 1;
        d.n_frusta = 1;
        d.local_kind = YH_LOCAL_TRS;
        d.bounds_kind = YH_BOUNDS_MESH_TABLE;
        d.max_meshes = nm;
        d.id_base = 1000;
        yh_scene* s = nullptr;
        CK(yh_scene_create(ctx, &d, &s));
        std::vector<uint32_t> parent(n, YH_NO_PARENT);
        const uint32_t lo[2] = {0, n};
        CK(yh_upload_hierarchy(s, parent.data(), lo));
        yh_synth_desc sd;
        memset(&sd, 0, sizeof sd);
        sd.seed = 1;
        CK(yh_scene_fill_synthetic(s, &sd));
        std::vector<uint32_t> vcount(nm);
        for (uint32_t m = 0; m < nm; ++m) vcount[m] = 3 * (m + 1);
        CK(yh_upload_vertex_counts(s, vcount.data(), nm));
        {   // mesh ids of single slots
            const uint32_t sl[3] = {0, 17, n - 1}, ids[3] = {nm - 1, 0, 3}, big[3] = {0, nm, 1};
            CK(yh_upload_mesh_id_rows(s, sl, ids, 3));
            EXPECT(yh_upload_mesh_id_rows(s, sl, big, 3) == YH_ERR_BAD_ARG);
        }
        CK(yh_frame(s, planes, 1, YH_FRAME_ALL_DIRTY));
        uint32_t counts[8], bc[8];
        CK(yh_get_counts(s, counts, bc));
        CK(yh_build_mesh_buckets(s, 0, 7));
        std::vector<yh_bucket> b(nm);
        std::vector<uint32_t> ids(counts[0]);
        uint64_t wr = 0;
        CK(yh_get_mesh_buckets(s, 0, b.data(), nm, ids.data(), ids.size(), &wr));
        uint32_t total = 0;
        for (uint32_t m = 0; m < nm; ++m) {
            EXPECT(b[m].mesh == m && b[m].first == total);
            total += b[m].count;
        }
        EXPECT(total == counts[0] && wr == counts[0]);
        EXPECT(yh_get_mesh_buckets(s, 0, b.data(), nm - 1, nullptr, 0, &wr) == YH_ERR_CAPACITY);
        yh_scene_destroy(s);
    }

    // ---- Mat4 (flecs column) scene: AoS-68 in, dirty rule, sparse rows, change tracking ----
    {
        const uint32_t n = 30011, levels = 4;
        std::vector<uint32_t> lo, parent;
        tree(n, levels, lo, parent);
        yh_scene_desc d;
        memset(&d, 0, sizeof d);
        d.n_entities = n;
        d.n_levels = levels;
        d.n_frusta = 0;
        d.local_kind = YH_LOCAL_MAT4;
        d.bounds_kind = YH_BOUNDS_NONE;
        yh_scene* s = nullptr;
        CK(yh_scene_create(ctx, &d, &s));
        CK(yh_upload_hierarchy(s, parent.data(), lo.data()));
        std::vector<unsigned char> col((size_t)n * 68, 0);
        for (uint32_t e = 0; e < n; ++e) {
            float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, (float)(e % 7), 1, 2, 1};
            memcpy(col.data() + (size_t)e * 68, m, 64);
            col[(size_t)e * 68 + 64] = 1;
        }
        CK(yh_upload_local_aos68(s, col.data(), 0, n));
        CK(yh_upload_world_aos68(s, col.data(), 0, n));
        CK(yh_scene_track_changes(s, 1));
        CK(yh_frame(s, nullptr, 0, 0));
        uint64_t changed = 0;
        CK(yh_get_changed_count(s, &changed));
        EXPECT(changed == n);
        CK(yh_frame(s, nullptr, 0, 0));
        CK(yh_get_changed_count(s, &changed));
        EXPECT(changed == 0);
        const uint32_t slots[3] = {5, lo[1] + 2, n - 1};
        CK(yh_upload_local_rows_aos68(s, slots, col.data(), 3));
        CK(yh_frame(s, nullptr, 0, 0));
        CK(yh_get_changed_count(s, &changed));
        EXPECT(changed >= 3 && changed < n);
        std::vector<uint32_t> cs(changed);
        std::vector<unsigned char> wrow(changed * 68), lrow(changed * 68);
        uint64_t wr = 0;
        CK(yh_download_changed_aos68(s, cs.data(), wrow.data(), lrow.data(), changed, &wr));
        EXPECT(wr == changed && cs[0] == 5);
        EXPECT(yh_download_changed_aos68(s, cs.data(), nullptr, nullptr, 1, &wr) == YH_ERR_CAPACITY);
        CK(yh_mark_dirty(s, 0, 1, 10, 5));
        CK(yh_frame(s, nullptr, 0, 0));
        CK(yh_download_local_aos68(s, col.data(), 0, n));
        const float root[16] = {2, 0, 0, 0, 0, 2, 0, 0, 0, 0, 2, 0, 1, 1, 1, 1};
        CK(yh_set_root(s, root));
        CK(yh_frame(s, nullptr, 0, 0));
        CK(yh_get_changed_count(s, &changed));
        EXPECT(changed == n);
        {   // the same through page-locked buffers: a few rows go through the import / export kernels alone, and the two
            // ways must agree (rows and slots)
            uint32_t* pslots = (uint32_t*)yh_host_alloc(ctx, 3 * 4);
            unsigned char* prows = (unsigned char*)yh_host_alloc(ctx, (uint64_t)n * 68);
            uint32_t* pcs = (uint32_t*)yh_host_alloc(ctx, (uint64_t)n * 4);
            EXPECT(pslots && prows && pcs);
            memcpy(pslots, slots, sizeof slots);
            memcpy(prows, col.data(), 3 * 68);
            for (int r = 0; r < 3; ++r) prows[r * 68 + 64] = 1;  // (as the onSet hook leaves a row that was set)
            CK(yh_upload_local_rows_aos68(s, pslots, prows, 3));
            CK(yh_frame(s, nullptr, 0, 0));
            uint64_t nw = 0, nl = 0, got = 0;
            CK(yh_get_changed_counts(s, &nw, &nl));
            EXPECT(nw >= 3 && nw < n && nw <= 32768);
            CK(yh_download_changed_rows_aos68(s, 0, pcs, prows, n, &got, nullptr, nullptr));
            std::vector<uint32_t> cs2(nw);
            std::vector<unsigned char> rows2(nw * 68);
            CK(yh_download_changed_rows_aos68(s, 0, cs2.data(), rows2.data(), nw, &got, nullptr, nullptr));
            EXPECT(got == nw && memcmp(cs2.data(), pcs, nw * 4) == 0 && memcmp(rows2.data(), prows, nw * 68) == 0);
            yh_host_free(ctx, pcs);
            yh_host_free(ctx, prows);
            yh_host_free(ctx, pslots);
        }
        {   // rows that change slots on the device: a rotation of three slots, and refused arguments
            const uint32_t src[3] = {5, 6, 7}, dst[3] = {6, 7, 5}, bad[1] = {n};
            std::vector<unsigned char> before(n * 68), after(n * 68);
            CK(yh_download_world_aos68(s, before.data(), 0, n));
            CK(yh_move_rows(s, src, dst, 3));
            CK(yh_download_world_aos68(s, after.data(), 0, n));
            for (int i = 0; i < 3; ++i) EXPECT(memcmp(&after[dst[i] * 68], &before[src[i] * 68], 64) == 0);
            EXPECT(yh_move_rows(s, bad, dst, 1) == YH_ERR_BAD_ARG);
            CK(yh_move_rows(s, nullptr, nullptr, 0));
        }
        {   // many runs / single slots in one call each (what a structural patch of the drop-in uses), and refused arguments
            const uint32_t rf[3] = {n - 4, n - 2, n - 4}, rn[3] = {3, 2, 1};
            const uint8_t ra[3] = {0, 0, 1};
            CK(yh_scene_set_alive_runs(s, rf, rn, ra, 3));
            const uint32_t bf[1] = {n - 1}, bc[1] = {2};
            EXPECT(yh_scene_set_alive_runs(s, bf, bc, ra, 1) == YH_ERR_BAD_ARG);
            CK(yh_scene_set_alive_runs(s, nullptr, nullptr, nullptr, 0));
            const uint32_t ps[2] = {lo[1] + 1, lo[2] + 3}, pv[2] = {lo[0] + 1, lo[1] + 2}, wrong[2] = {lo[0] + 1, lo[0]};
            CK(yh_upload_parent_rows(s, ps, pv, 2));
            EXPECT(yh_upload_parent_rows(s, ps, wrong, 2) == YH_ERR_BAD_ARG);  // slot of level 2 with a parent of level 0
            EXPECT(yh_upload_mesh_id_rows(s, ps, pv, 2) == YH_ERR_STATE);        // no mesh table in this scene
            EXPECT(yh_upload_aabb_rows(s, ps, root, 1) == YH_ERR_STATE);          // nor per-entity boxes
            CK(yh_frame(s, nullptr, 0, 0));
        }
        yh_scene_destroy(s);
    }
    // ---- one-rank communicator: both forms of the draw-list gather ----
    {
        unsigned char id[YH_UNIQUE_ID_BYTES];
        CK(yh_comm_unique_id(id));
        CK(yh_comm_init(ctx, 1, 0, id));
        const uint32_t n = 200003, levels = 3;
        std::vector<uint32_t> lo, parent;
        tree(n, levels, lo, parent);
        yh_scene_desc d;
        memset(&d, 0, sizeof d);
        d.n_entities = n;
        d.n_levels = levels;
        d.n_frusta = 2;
        d.local_kind = YH_LOCAL_TRS;
        d.bounds_kind = YH_BOUNDS_PER_ENTITY;
        d.id_base = 77;
        yh_scene* s = nullptr;
        CK(yh_scene_create(ctx, &d, &s));
        CK(yh_upload_hierarchy(s, parent.data(), lo.data()));
        yh_synth_desc sd;
        memset(&sd, 0, sizeof sd);
        sd.seed = 3;
        CK(yh_scene_fill_synthetic(s, &sd));
        for (int k = 0; k < 9; ++k) {
            CK(yh_frame(s, planes, 1 + (k & 1), YH_FRAME_ALL_DIRTY | YH_FRAME_ASYNC));
            CK(yh_allgather_draw_lists_async(s));
        }
        uint32_t counts[8], bc[8], gc[2];
        CK(yh_get_gathered_counts(s, gc));
        CK(yh_get_counts(s, counts, bc));
        EXPECT(gc[0] == counts[0] && counts[0] > 0);
        std::vector<uint32_t> own(counts[0]), got(counts[0]);
        uint64_t wr = 0;
        CK(yh_get_draw_list(s, 0, own.data(), own.size(), &wr));
        CK(yh_get_gathered_draw_list(s, 0, got.data(), got.size(), &wr));
        EXPECT(wr == counts[0] && memcmp(own.data(), got.data(), own.size() * 4) == 0);
        CK(yh_allgather_draw_lists(s, gc));
        EXPECT(gc[0] == counts[0]);
        EXPECT(yh_get_gathered_draw_list(s, 0, got.data(), 1, &wr) == YH_ERR_CAPACITY);
        yh_scene_destroy(s);
        yh_comm_destroy(ctx);
    }
    yh_shutdown(ctx);
    printf("abi_asan_driver: ok\n");
    return 0;
}
