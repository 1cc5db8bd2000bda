/* CPU probe of the hand-written data structures inside GpuTransformCullSystem.c (no GPU, nothing is run on a device): the
 * partitioned open-addressing map filled by helper threads against a serial fill and against a sorted copy of the keys,
 * the in-place rewrite of its values, and the radix sort of tables by parent slot against qsort.  The source file is
 * included, so the static functions are tested as they are.  Prints "<name> ok|FAILED"; exit code 0 iff all pass.
 * Built and run by tests/test_flecs_harness.py. */
#include "../yume_amd/host/GpuTransformCullSystem.c"

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) {
    rng ^= rng << 13;
    rng ^= rng >> 7;
    rng ^= rng << 17;
    return rng;
}
static int failed;
static void report(const char* name, int ok) {
    printf("%s %s\n", name, ok ? "ok" : "FAILED");
    if (!ok) failed = 1;
}

static void* pool_probe_job(void* arg) {
    atomic_fetch_add((_Atomic uint32_t*)arg, 1);
    return NULL;
}

int main(void) {
#ifndef PROBE_POOL_ONLY /* (the ThreadSanitizer build is after the worker pool alone) */
    /* keys as flecs makes them: a 32-bit index and a generation in the high half, some slots empty (0) */
    const uint64_t n = 300000;
    ecs_entity_t* keys = (ecs_entity_t*)malloc(sizeof(ecs_entity_t) * n);
    for (uint64_t i = 0; i < n; ++i) keys[i] = (i % 97 == 0) ? 0 : ((uint64_t)(i + 1000) | ((rnd() % 5) << 32));
    for (int threads = 1; threads <= 16; threads *= 4) {
        Map m = {0};
        int ok = map_init(&m, n) == 0 && m.pbits == MAP_PART_BITS; /* large enough to be partitioned */
        map_fill_mt(&m, keys, n, threads);
        ok = ok && !m.overflow;
        uint64_t found = 0;
        for (uint64_t i = 0; i < n && ok; ++i) {
            uint32_t v = 0;
            if (!keys[i]) continue;
            ok = map_get(&m, keys[i], &v) && v == (uint32_t)i;
            found++;
        }
        uint32_t v;
        ok = ok && !map_get(&m, 999, &v) && !map_get(&m, (ecs_entity_t)(n + 5000), &v); /* absent keys */
        uint64_t used = 0;
        for (uint64_t q = 0; q < m.cap; ++q) used += m.e[q].key != 0;
        ok = ok && used == found;
        /* a key re-put keeps one entry; a new key is reported as new */
        ok = ok && map_put(&m, keys[1], 77) == 0 && map_get(&m, keys[1], &v) && v == 77 && map_put(&m, 123456789012ull, 5) == 1;
        /* map_reuse: emptied, same table */
        ok = ok && map_reuse(&m, n) && !map_get(&m, keys[1], &v);
        char name[64];
        snprintf(name, sizeof name, "map_fill_%d_threads", threads);
        report(name, ok);
        map_free(&m);
    }
    {   /* a small map is one partition and behaves the same */
        Map m = {0};
        int ok = map_init(&m, 100) == 0 && m.pbits == 0;
        for (uint32_t i = 1; i <= 100; ++i) map_put(&m, i * 7919u, i);
        uint32_t v;
        for (uint32_t i = 1; i <= 100 && ok; ++i) ok = map_get(&m, i * 7919u, &v) && v == i;
        report("map_small", ok && !map_get(&m, 5, &v));
        map_free(&m);
    }
    {   /* stable sort by key: equal keys keep their batch order, as qsort with by_key (which breaks ties by batch) */
        const uint32_t nb = 200000;
        KeyedBatch* a = (KeyedBatch*)malloc(sizeof(KeyedBatch) * nb * 2);
        KeyedBatch* ref = (KeyedBatch*)malloc(sizeof(KeyedBatch) * nb);
        for (uint32_t i = 0; i < nb; ++i) {
            a[i].key = (uint32_t)(rnd() % 50000) * ((i & 1) ? 1u : 20000u); /* up to 2^30, many ties */
            a[i].batch = i;
            ref[i] = a[i];
        }
        qsort(ref, nb, sizeof(KeyedBatch), by_key);
        sort_keyed(a, a + nb, nb);
        report("sort_keyed_200000", memcmp(a, ref, sizeof(KeyedBatch) * nb) == 0);
        for (uint32_t i = 0; i < 500; ++i) a[i].key = (uint32_t)(rnd() % 40), a[i].batch = i, ref[i] = a[i];
        qsort(ref, 500, sizeof(KeyedBatch), by_key);
        sort_keyed(a, a + nb, 500);
        report("sort_keyed_500", memcmp(a, ref, sizeof(KeyedBatch) * 500) == 0);
        free(a);
        free(ref);
    }
    free(keys);
#endif
    {   /* the workers that outlive a frame: every job of every batch runs exactly once, whoever takes it - workers awake,
         * workers asleep (a pause before the batch), more jobs than workers, no jobs, begin .. the caller's own work .. finish,
         * stop and start again with another number of threads */
        int ok = 1;
        static _Atomic uint32_t ran[MAX_THREADS];
        for (int threads = 1; threads <= 16 && ok; threads = threads == 1 ? 5 : threads * 3 + 1) { /* 1, 5, 16 */
            Pool p;
            memset(&p, 0, sizeof p);
            pool_start(&p, threads);
            ok = ok && p.n_workers == (uint32_t)threads - 1;
            for (uint32_t batch = 0; batch < 3000 && ok; ++batch) {
                const uint32_t jobs = (uint32_t)(rnd() % (MAX_THREADS + 1));
                void* args[MAX_THREADS];
                for (uint32_t j = 0; j < MAX_THREADS; ++j) {
                    atomic_store(&ran[j], 0);
                    args[j] = (void*)&ran[j];
                }
                if (batch % 500 == 499) usleep(3000); /* the workers have gone to sleep */
                pool_begin(&p, pool_probe_job, args, jobs);
                if (batch % 3 == 0) sched_yield(); /* (the caller's own part of the work) */
                pool_finish(&p);
                for (uint32_t j = 0; j < MAX_THREADS && ok; ++j) ok = atomic_load(&ran[j]) == (j < jobs ? 1u : 0u);
            }
            pool_stop(&p);
            pool_stop(&p); /* (a second stop finds nothing to do) */
        }
        report("worker_pool", ok);
    }
    return failed;
}
