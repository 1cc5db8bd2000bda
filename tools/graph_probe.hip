// A/B: a cfg2-like scene's frames (two dependent launches each, ~33 us) enqueued one by one against the same frames
// replayed from a hipGraph (64 frames captured on the library's stream - the ring of boundary-counter sets and the two
// sets of visibility words come round after 64 - and launched 32 times).  Links the library; the context's stream comes
// from yh_stream().  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o tools/bin/graph_probe tools/graph_probe.hip -Lyume_amd -lyume_hip -Wl,-rpath,$PWD/yume_amd
#include <chrono>
#include <cstdio>
#include <vector>

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "yume_hip.h"

#define CK(call)                                                                                       \
    do {                                                                                               \
        int rc_ = (call);                                                                              \
        if (rc_ != 0) {                                                                                \
            fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, yh_last_error(ctx)); \
            return 1;                                                                                  \
        }                                                                                              \
    } while (0)

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000000u, levels = 8;
    yh_ctx* ctx = nullptr;
    if (yh_init(0, &ctx) != YH_OK) return 1;
    std::vector<uint32_t> lo(levels + 1, 0), parent(n, YH_NO_PARENT);
    for (uint32_t l = 0; l < levels; ++l) lo[l + 1] = (uint32_t)((uint64_t)n * ((2u << l) - 1u) / 255u);  // level sizes ~ 2^l
    lo[levels] = n;
    for (uint32_t l = 1; l < levels; ++l) {
        const uint32_t pb = lo[l - 1], pc = lo[l] - lo[l - 1], cb = lo[l], cc = lo[l + 1] - lo[l];
        for (uint32_t i = 0; i < cc; ++i) parent[cb + i] = pb + (uint32_t)((uint64_t)i * pc / cc);
    }
    yh_scene_desc d;
    memset(&d, 0, sizeof d);
    d.n_entities = n;
    d.n_levels = levels;
    d.n_frusta = 1;
    d.local_kind = YH_LOCAL_TRS;
    d.bounds_kind = YH_BOUNDS_PER_ENTITY;
    yh_scene* s = nullptr;
    CK(yh_scene_create(ctx, &d, &s));
    CK(yh_upload_hierarchy(s, parent.data(), lo.data()));
    yh_synth_desc sd;
    memset(&sd, 0, sizeof sd);
    sd.seed = 0x59554D45ull;
    CK(yh_scene_fill_synthetic(s, &sd));
    float vp[16], proj[16], view[16], planes[24];
    const float eye[3] = {0, 40, 150}, at[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    yh_perspective(1.3089969f, 2.0f, 0.1f, 200.0f, proj);
    yh_look_at(eye, at, up, view);
    yh_mat4_mul(proj, view, vp);
    yh_extract_planes(vp, planes);
    const uint32_t flags = YH_FRAME_ALL_DIRTY | YH_FRAME_ASYNC;
    const int FRAMES = 2048, PER_GRAPH = 64;
    uint32_t counts[8], bc[8];
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 200; ++i) CK(yh_frame(s, planes, 1, flags));
        CK(yh_wait(ctx));
        double t0 = now_us();
        for (int i = 0; i < FRAMES; ++i) CK(yh_frame(s, planes, 1, flags));
        CK(yh_wait(ctx));
        const double eager = (now_us() - t0) / FRAMES;
        CK(yh_get_counts(s, counts, bc));
        const uint32_t visible_eager = counts[0];

        // one more eager frame leaves a pending expansion, as every captured frame will find one
        CK(yh_frame(s, planes, 1, flags));
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        if (hipStreamBeginCapture((hipStream_t)yh_stream(ctx), hipStreamCaptureModeThreadLocal) != hipSuccess) return 2;
        for (int i = 0; i < PER_GRAPH; ++i) CK(yh_frame(s, planes, 1, flags));
        if (hipStreamEndCapture((hipStream_t)yh_stream(ctx), &graph) != hipSuccess) return 3;
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) return 4;
        size_t nodes = 0;
        (void)hipGraphGetNodes(graph, nullptr, &nodes);
        for (int i = 0; i < 4; ++i)
            if (hipGraphLaunch(exec, (hipStream_t)yh_stream(ctx)) != hipSuccess) return 5;
        if (hipStreamSynchronize((hipStream_t)yh_stream(ctx)) != hipSuccess) return 6;
        t0 = now_us();
        for (int i = 0; i < FRAMES / PER_GRAPH; ++i)
            if (hipGraphLaunch(exec, (hipStream_t)yh_stream(ctx)) != hipSuccess) return 5;
        if (hipStreamSynchronize((hipStream_t)yh_stream(ctx)) != hipSuccess) return 6;
        const double replay = (now_us() - t0) / FRAMES;
        CK(yh_get_counts(s, counts, bc));  // (flushes the last captured frame's pending expansion)
        printf("n=%u: eager %.2f us per frame, hipGraph replay (%d frames, %zu nodes per graph) %.2f us per frame; visible %u / %u\n", n,
               eager, PER_GRAPH, nodes, replay, visible_eager, counts[0]);
        (void)hipGraphExecDestroy(exec);
        (void)hipGraphDestroy(graph);
    }
    yh_scene_destroy(s);
    yh_shutdown(ctx);
    return 0;
}
