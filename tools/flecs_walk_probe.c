/* Where does the time of walking the system's query over a tree of ~10^5 one-parent tables go?  CPU only (no GPU, no
 * yume_hip): builds n entities in an 8-level random tree inside the reference's flecs (the shape of tools/flecs_bench.py),
 * registers the three-term query of TransformSyncSystem.zig:12-51 and times (a) the bare iteration, (b) + the two column
 * index lookups per table, (c) + copying the entity ids, (d) ecs_each_id over one component instead of the query.
 * gcc -O2 -DFLECS_CONFIG_HEADER=1 -I<reference>/vendor/flecs tools/flecs_walk_probe.c oracle/_ref/flecs.o -lm -lpthread */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "flecs.h"

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}
static ecs_entity_t comp(ecs_world_t* w, const char* name, int size) {
    ecs_entity_desc_t ed;
    memset(&ed, 0, sizeof ed);
    ed.name = name;
    ed.use_low_id = true;
    ecs_component_desc_t cd;
    memset(&cd, 0, sizeof cd);
    cd.entity = ecs_entity_init(w, &ed);
    cd.type.size = size;
    cd.type.alignment = 4;
    return ecs_component_init(w, &cd);
}
static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd(void) {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000000u;
    const uint32_t levels = 8;
    ecs_world_t* w = ecs_init();
    const ecs_entity_t W = comp(w, "WorldTransform", 68), L = comp(w, "LocalTransform", 68);
    char row[68];
    memset(row, 0, sizeof row);
    const ecs_entity_t root = ecs_new(w);
    ecs_set_id(w, root, W, 68, row);
    /* level sizes grow by ~2.7x per level (the synthetic tree of the benches: 2^k-ish), parents random in the level above */
    uint32_t lo[16];
    {
        double total = 0, f = 1;
        for (uint32_t l = 0; l < levels; ++l, f *= 2.0) total += f;
        uint32_t at = 0;
        f = 1;
        for (uint32_t l = 0; l < levels; ++l, f *= 2.0) {
            lo[l] = at;
            at += (uint32_t)(n * f / total);
        }
        lo[levels] = n;
    }
    ecs_entity_t* e = (ecs_entity_t*)malloc(sizeof(ecs_entity_t) * n);
    double t0 = now_ms();
    for (uint32_t l = 0; l < levels; ++l)
        for (uint32_t i = lo[l]; i < lo[l + 1]; ++i) {
            const ecs_entity_t parent = l == 0 ? root : e[lo[l - 1] + rnd() % (lo[l] - lo[l - 1])];
            e[i] = ecs_new_w_id(w, ecs_pair(EcsChildOf, parent));
            ecs_set_id(w, e[i], L, 68, row);
            ecs_set_id(w, e[i], W, 68, row);
        }
    printf("built %u entities in %.0f ms\n", n, now_ms() - t0);
    ecs_query_desc_t qd;
    memset(&qd, 0, sizeof qd);
    qd.terms[0].id = W;
    qd.terms[1].id = L;
    qd.terms[2].id = W;
    qd.terms[2].inout = EcsIn;
    qd.terms[2].src.id = EcsUp;
    qd.cache_kind = EcsQueryCacheAuto;
    t0 = now_ms();
    ecs_query_t* q = ecs_query_init(w, &qd);
    printf("query init %.0f ms\n", now_ms() - t0);
    ecs_entity_t* ents = (ecs_entity_t*)malloc(sizeof(ecs_entity_t) * n);
    for (int rep = 0; rep < 3; ++rep) {
        uint64_t tables = 0, rows = 0, sink = 0;
        t0 = now_ms();
        ecs_iter_t it = ecs_query_iter(w, q);
        while (ecs_query_next(&it)) {
            tables++;
            rows += (uint64_t)it.count;
        }
        const double a = now_ms() - t0;
        t0 = now_ms();
        it = ecs_query_iter(w, q);
        while (ecs_query_next(&it)) {
            sink += (uint64_t)ecs_table_get_column_index(w, it.table, W) + (uint64_t)ecs_table_get_column_index(w, it.table, L);
            sink += ecs_field_src(&it, 2);
        }
        const double b = now_ms() - t0;
        t0 = now_ms();
        it = ecs_query_iter(w, q);
        uint64_t at = 0;
        while (ecs_query_next(&it)) {
            sink += (uint64_t)ecs_table_get_column_index(w, it.table, W) + (uint64_t)ecs_table_get_column_index(w, it.table, L);
            sink += ecs_field_src(&it, 2);
            memcpy(ents + at, it.entities, sizeof(ecs_entity_t) * (size_t)it.count);
            at += (uint64_t)it.count;
        }
        const double c = now_ms() - t0;
        t0 = now_ms();
        uint64_t tables_each = 0;
        it = ecs_each_id(w, L);
        while (ecs_each_next(&it)) {
            tables_each++;
            sink += (uint64_t)it.count;
        }
        const double d = now_ms() - t0;
        t0 = now_ms();
        it = ecs_each_id(w, L);
        at = 0;
        while (ecs_each_next(&it)) {
            ecs_entity_t parent = ecs_get_target(w, it.entities[0], EcsChildOf, 0);
            sink += parent;
            memcpy(ents + at, it.entities, sizeof(ecs_entity_t) * (size_t)it.count);
            at += (uint64_t)it.count;
        }
        const double f = now_ms() - t0;
        printf("%llu tables, %llu rows: bare query iteration %.1f ms, + column indices and source %.1f, + entity ids %.1f; "
               "ecs_each_id(LocalTransform) over %llu tables %.1f ms, + parent of the first row and entity ids %.1f  (%llu)\n",
               (unsigned long long)tables, (unsigned long long)rows, a, b, c, (unsigned long long)tables_each, d, f, (unsigned long long)sink);
    }
    return 0;
}
