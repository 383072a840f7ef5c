// Prints the byte offsets of the FhRenderState fields used by the assembly kernels
// (gen_interp.py); host and device layouts are identical (plain C, 64-bit pointers).
#include <cstddef>
#include <cstdio>

#include "render_state.h"

int main() {
#define O(name, expr) printf("%s\"%s\": %zu", first ? "{" : ", ", name, (size_t)offsetof(FhRenderState, expr)), first = false
    bool first = true;
    O("P.mat", P.mat); O("P.width", P.width); O("P.height", P.height); O("P.tiles", P.tiles); O("P.slab", P.slab);
    O("P.in_kind", P.in_kind); O("P.in_value", P.in_value);
    O("arena", arena); O("leaves", leaves); O("leaf_table", leaf_table); O("slab_z", slab_z); O("zbuf", zbuf); O("normals", normals);
    O("arena_cap", arena_cap); O("arena_head", arena_head); O("arena_overflow", arena_overflow);
    O("chw", chw); O("frame_stamp", frame_stamp); O("tgroup", tgroup); O("n_tgroups", n_tgroups); O("chwr", chwr); O("slots", slots); O("slot_cap", slot_cap); O("n_slots", n_slots); O("eval_cur", eval_cur);
    O("fp_list", fp_list); O("fp_count", fp_count); O("fp_cursor", fp_cursor); O("hit_list", hit_list); O("stat", stat);
    O("count", count); O("count_big", count_big); O("queue", queue); O("qcap", qcap); O("squeue", squeue); O("n_leaves", n_leaves);
    O("leaf_cap", leaf_cap); O("P.depth", P.depth); O("P.n_levels", P.n_levels); O("P.max_regs", P.max_regs);
    printf(", \"sizeof_state\": %zu, \"sizeof_slot\": %zu, \"sizeof_group\": %zu, \"sizeof_leaf\": %zu", sizeof(FhRenderState), sizeof(FhSlot),
           sizeof(FhGroup), sizeof(FhLeaf));
    printf("}\n");
    return 0;
}
