/* CPU probe of GpuTransformCullSystem's structural-change tracking inside the real flecs (no GPU: the system is
 * registered, never run).  Every operation that lets rows enter, leave or move inside a table holding one of the two
 * transform components must raise the flag; operations that leave those tables alone must not.  Prints one line per
 * case: "<name> <expected> <got>"; exit code 0 iff all agree.  Built and run by tests/test_flecs_harness.py. */
#include <stdio.h>
#include <string.h>

#include "GpuTransformCullSystem.h"

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

static int bad;
static void expect(GpuTransformCullSystem* s, const char* name, int want) {
    const int got = GpuTransformCullSystem_structureDirty(s);
    printf("%s %d %d\n", name, want, got);
    if (got != want) bad = 1;
    GpuTransformCullSystem_clearStructureDirty(s);
}

int main(void) {
    ecs_world_t* w = ecs_init();
    const ecs_entity_t W = comp(w, "WorldTransform", 68), L = comp(w, "LocalTransform", 68), X = comp(w, "Other", 4);
    char row[68];
    memset(row, 0, sizeof row);
    const ecs_entity_t root = ecs_new(w);
    ecs_set_id(w, root, W, 68, row);
    ecs_entity_t e[4];
    for (int i = 0; i < 4; ++i) {
        e[i] = ecs_new_w_id(w, ecs_pair(EcsChildOf, i < 2 ? root : e[0]));
        ecs_set_id(w, e[i], L, 68, row);
        ecs_set_id(w, e[i], W, 68, row);
    }
    GpuTransformCullSystem* s = NULL;
    yh_ctx* fake_ctx = (yh_ctx*)&bad; /* never dereferenced: the system is not run */
    if (!GpuTransformCullSystem_registerTo(w, L, W, fake_ctx, &s)) return 2;
    expect(s, "after_registration", 1);
    expect(s, "nothing_happened", 0);
    ecs_set_id(w, e[1], L, 68, row); /* a set of an existing component: rows stay where they are */
    ecs_modified_id(w, e[2], W);
    expect(s, "set_and_modified_existing", 0);
    ecs_set_id(w, e[1], X, 4, row); /* an unrelated component moves the row to another archetype */
    expect(s, "unrelated_component_added", 1);
    ecs_remove_id(w, e[1], X);
    expect(s, "unrelated_component_removed", 1);
    const ecs_entity_t tag = ecs_new(w);
    expect(s, "entity_without_transforms_created", 0);
    ecs_add_id(w, e[3], tag);
    expect(s, "tag_added", 1);
    ecs_add_id(w, e[3], ecs_pair(EcsChildOf, e[1])); /* Editor.zig:590-636 */
    expect(s, "reparented", 1);
    const ecs_entity_t fresh = ecs_new_w_id(w, ecs_pair(EcsChildOf, root)); /* GameApp.zig:204-215 */
    ecs_set_id(w, fresh, L, 68, row);
    expect(s, "entity_with_local_created", 1);
    const ecs_entity_t other = ecs_new(w);
    ecs_set_id(w, other, X, 4, row);
    ecs_add_id(w, other, tag);
    ecs_delete(w, other);
    expect(s, "unrelated_entity_lifecycle", 0);
    ecs_delete(w, e[0]); /* children go with it */
    expect(s, "deleted_with_children", 1);
    ecs_remove_id(w, fresh, L);
    expect(s, "local_removed", 1);
    ecs_add_id(w, root, tag); /* the root holds a WorldTransform only: its table is one the Up term reads */
    expect(s, "root_moved_archetype", 1);
    /* world.enable(e, false) (ecs.zig:546) adds the Disabled tag: the row moves to a table the system's query does not
     * match (flecs leaves disabled entities and prefabs out); enabling brings it back.  Both must be seen. */
    ecs_enable(w, e[1], false);
    expect(s, "entity_disabled", 1);
    ecs_enable(w, e[1], true);
    expect(s, "entity_enabled_again", 1);
    ecs_add_id(w, e[1], EcsPrefab);
    expect(s, "entity_made_a_prefab", 1);
    GpuTransformCullSystem_destroy(s);
    ecs_fini(w);
    return bad;
}
