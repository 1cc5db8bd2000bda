// yh_consumers.hip — draw-list consumers: object data and draw records in list order, buckets per key.
#define YH_K_CONSUMERS
#include "yh_internal.hpp"

// ---- draw-list consumer -----------------------------------------------------------------------
extern "C" int yh_upload_vertex_counts(yh_scene* s, const uint32_t* counts, uint64_t n) {
    if (!s || !counts) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_upload_vertex_counts: NULL argument");
    yh_ctx* ctx = s->ctx;
    const bool table = s->desc.bounds_kind == YH_BOUNDS_MESH_TABLE;
    const uint64_t want = table ? s->desc.max_meshes : s->n;
    if (table ? (n == 0 || n > want) : (n != want))
        return fail(ctx, YH_ERR_BAD_ARG, "yh_upload_vertex_counts: %llu counts, scene wants %s%llu", (unsigned long long)n,
                    table ? "at most " : "", (unsigned long long)want);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_scene(s)) return rc;
    if (!s->vcount) {
        HIP_TRY(ctx, hipMalloc((void**)&s->vcount, want * 4));
        HIP_TRY(ctx, hipMemset(s->vcount, 0, want * 4));
    }
    HIP_TRY(ctx, hipMemcpy(s->vcount, counts, n * 4, hipMemcpyHostToDevice));
    return YH_OK;
}

extern "C" int yh_build_object_data(yh_scene* s, uint32_t frustum, uint32_t batch_offset) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_build_object_data: scene is NULL");
    yh_ctx* ctx = s->ctx;
    if (frustum >= s->last_n_frusta) return fail(ctx, YH_ERR_STATE, "yh_build_object_data: frustum %u was not culled by the last frame", frustum);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = flush_expand(s)) return rc;
    if (!s->obj[frustum]) {
        HIP_TRY(ctx, hipMalloc((void**)&s->obj[frustum], s->draw_cap * 64));
        HIP_TRY(ctx, hipMalloc((void**)&s->cmd[frustum], s->draw_cap * 16));
    }
    hipLaunchKernelGGL(k_build_objects, dim3(cdiv(std::min<uint64_t>(s->n, s->draw_cap), YH_BLOCK)), dim3(YH_BLOCK), 0, ctx->stream,
                       s->world, s->draw[frustum], s->counts + frustum, s->draw_cap, s->desc.id_base, s->vcount,
                       s->desc.bounds_kind == YH_BOUNDS_MESH_TABLE ? s->mesh_id : nullptr, batch_offset, s->obj[frustum],
                       s->cmd[frustum]);
    HIP_TRY(ctx, hipGetLastError());
    s->frame_pending = true;
    s->bucketed[frustum] = false;
    return YH_OK;
}

// Per-mesh bucketing: the same object data and draw records, grouped by mesh id.
extern "C" int yh_build_mesh_buckets(yh_scene* s, uint32_t frustum, uint32_t batch_offset) {
    if (!s) return fail(nullptr, YH_ERR_BAD_ARG, "yh_build_mesh_buckets: scene is NULL");
    yh_ctx* ctx = s->ctx;
    if (s->desc.bounds_kind != YH_BOUNDS_MESH_TABLE)
        return fail(ctx, YH_ERR_STATE, "yh_build_mesh_buckets: the scene has no mesh table (entities carry no mesh id)");
    if (frustum >= s->last_n_frusta) return fail(ctx, YH_ERR_STATE, "yh_build_mesh_buckets: frustum %u was not culled by the last frame", frustum);
    if (s->n_meshes < 1 || s->n_meshes > YH_BUCKET_MAX_KEYS)
        return fail(ctx, YH_ERR_UNSUPPORTED, "yh_build_mesh_buckets: %u meshes, 1..%u supported", s->n_meshes, YH_BUCKET_MAX_KEYS);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = flush_expand(s)) return rc;
    if (!s->obj[frustum]) {
        HIP_TRY(ctx, hipMalloc((void**)&s->obj[frustum], s->draw_cap * 64));
        HIP_TRY(ctx, hipMalloc((void**)&s->cmd[frustum], s->draw_cap * 16));
    }
    // Up to 256 keys: 4096 workgroup slices, tiles sorted inside LDS.  More (one key per (material, mesh) pair of a large
    // scene): one slice per WAVE, counters in global memory only, keys x slices kept within 16M counters.
    const uint32_t M = std::min<uint32_t>(std::max<uint32_t>(s->desc.max_meshes, 1u), YH_BUCKET_MAX_KEYS);
    const bool wide = s->n_meshes > YH_BUCKET_MAX_MESHES || getenv("YH_BUCKET_WIDE") != nullptr;  // (the knob: A/B runs of the two paths)
    // Slices the list is cut into.  Few keys: up to 32768 workgroup slices of at least four 256-entry tiles each, keys x
    // slices within 4M counters (64M instances / 16 keys: 4096 slices 1.945 ms, 32768 1.898, one slice per tile - 250k -
    // 2.448: the scans over the slices grow; profiles/r03/bucket_slices.txt).  Many keys: one slice per wave, <= 16384.
    uint32_t slices;
    if (wide) {
        slices = 16384u;
        while (slices > 256u && (uint64_t)slices * s->n_meshes > (16ull << 20)) slices >>= 1;
    } else {
        const uint64_t tiles = (std::min<uint64_t>(s->draw_cap, s->n) + YH_BLOCK - 1) / YH_BLOCK;
        const uint64_t room = (4ull << 20) / s->n_meshes;
        slices = (uint32_t)std::max<uint64_t>(std::min<uint64_t>(std::min<uint64_t>(tiles / 4, 32768), room), 256);
        if (const char* v = getenv("YH_BUCKET_SLICES")) slices = (uint32_t)strtoul(v, nullptr, 10);  // (A/B runs)
        slices = (slices + 255u) & ~255u;
    }
    if (!s->blist[frustum]) {
        HIP_TRY(ctx, hipMalloc((void**)&s->blist[frustum], s->draw_cap * 4));
        HIP_TRY(ctx, hipMalloc((void**)&s->bkt[frustum], (uint64_t)M * 12));
    }
    const uint32_t n_seg = (slices + YH_BUCKET_SEG - 1u) / YH_BUCKET_SEG;
    const uint64_t scratch_words = (uint64_t)s->n_meshes * slices + 2ull * M + (uint64_t)s->n_meshes * n_seg;
    if (scratch_words > s->bscratch_words) {  // (the mesh table grew, or went from few keys to many)
        if (s->bscratch) {
            if (int rc = sync_scene(s)) return rc;
            HIP_TRY(ctx, hipFree(s->bscratch));
            s->bscratch = nullptr;
            s->bscratch_words = 0;
        }
        const uint64_t words = scratch_words + scratch_words / 4;
        HIP_TRY(ctx, hipMalloc((void**)&s->bscratch, words * 4));
        s->bscratch_words = words;
    }
    BucketArgs a{};
    a.list = s->draw[frustum];
    a.count_ptr = s->counts + frustum;
    a.cap = s->draw_cap;
    a.id_base = s->desc.id_base;
    a.mesh_id = s->mesh_id;
    a.n_meshes = s->n_meshes;
    a.hist = s->bscratch;
    a.totals = s->bscratch + (uint64_t)s->n_meshes * slices;
    a.starts = a.totals + M;
    a.segtot = a.starts + M;
    a.n_seg = n_seg;
    a.buckets = s->bkt[frustum];
    a.world = s->world;
    a.vcount = s->vcount;
    a.batch_offset = batch_offset;
    a.out_list = s->blist[frustum];
    a.obj = s->obj[frustum];
    a.cmds = s->cmd[frustum];
    a.n_slices = slices;
    if (wide) {
        HIP_TRY(ctx, hipMemsetAsync(a.hist, 0, (uint64_t)s->n_meshes * slices * 4, ctx->stream));
        hipLaunchKernelGGL(k_bucketw_hist, dim3(slices / 4), dim3(YH_BLOCK), 0, ctx->stream, a);
    } else {
        hipLaunchKernelGGL(k_bucket_hist, dim3(slices), dim3(YH_BLOCK), 0, ctx->stream, a);
    }
    hipLaunchKernelGGL(k_bucket_segsum, dim3(s->n_meshes * n_seg), dim3(YH_BLOCK), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_bucket_starts, dim3(1), dim3(YH_BLOCK), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_bucket_rows, dim3(s->n_meshes * n_seg), dim3(YH_BLOCK), 0, ctx->stream, a);
    if (wide) hipLaunchKernelGGL(k_bucketw_scatter, dim3(slices / 4), dim3(YH_BLOCK), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_bucket_scatter, dim3(slices), dim3(YH_BLOCK), 0, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    s->frame_pending = true;
    s->bucketed[frustum] = true;
    return YH_OK;
}

extern "C" int yh_get_mesh_buckets(yh_scene* s, uint32_t frustum, yh_bucket* buckets, uint32_t cap_buckets,
                                   uint32_t* entities, uint64_t cap, uint64_t* written) {
    if (!s || !written) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_get_mesh_buckets: NULL argument");
    yh_ctx* ctx = s->ctx;
    if (frustum >= YH_MAX_FRUSTA || !s->bucketed[frustum]) return fail(ctx, YH_ERR_STATE, "yh_get_mesh_buckets: call yh_build_mesh_buckets first");
    if (int rc = sync_scene(s)) return rc;
    const uint64_t n = std::min<uint64_t>(s->h_counts[frustum], s->draw_cap);
    *written = n;
    if (buckets && cap_buckets < s->n_meshes)
        return fail(ctx, YH_ERR_CAPACITY, "yh_get_mesh_buckets: %u buckets, room for %u", s->n_meshes, cap_buckets);
    if (entities && n > cap) return fail(ctx, YH_ERR_CAPACITY, "yh_get_mesh_buckets: %llu entities, room for %llu", (unsigned long long)n, (unsigned long long)cap);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (buckets) HIP_TRY(ctx, hipMemcpy(buckets, s->bkt[frustum], (uint64_t)s->n_meshes * 12, hipMemcpyDeviceToHost));
    if (entities && n) HIP_TRY(ctx, hipMemcpy(entities, s->blist[frustum], n * 4, hipMemcpyDeviceToHost));
    return YH_OK;
}

extern "C" int yh_get_object_data(yh_scene* s, uint32_t frustum, void* model_matrices, yh_draw_cmd* cmds, uint64_t cap,
                                  uint64_t* written) {
    if (!s || !written) return fail(s ? s->ctx : nullptr, YH_ERR_BAD_ARG, "yh_get_object_data: NULL argument");
    yh_ctx* ctx = s->ctx;
    if (frustum >= YH_MAX_FRUSTA || !s->obj[frustum]) return fail(ctx, YH_ERR_STATE, "yh_get_object_data: call yh_build_object_data first");
    if (int rc = sync_scene(s)) return rc;
    const uint64_t n = std::min<uint64_t>(s->h_counts[frustum], s->draw_cap);
    *written = n;
    if (n > cap) return fail(ctx, YH_ERR_CAPACITY, "yh_get_object_data: %llu objects, room for %llu", (unsigned long long)n, (unsigned long long)cap);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (model_matrices && n) HIP_TRY(ctx, hipMemcpy(model_matrices, s->obj[frustum], n * 64, hipMemcpyDeviceToHost));
    if (cmds && n) HIP_TRY(ctx, hipMemcpy(cmds, s->cmd[frustum], n * 16, hipMemcpyDeviceToHost));
    return YH_OK;
}

extern "C" void* yh_object_data_device(yh_scene* s, uint32_t frustum, void** cmds_device) {
    if (!s || frustum >= YH_MAX_FRUSTA) return nullptr;
    if (cmds_device) *cmds_device = s->cmd[frustum];
    return s->obj[frustum];
}

