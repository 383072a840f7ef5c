/* A native caller of the drop-in boundary, in plain C11: the closest stand-in available in this image for the Rust FFI
 * of INTEGRATION.md (no rustc / cargo here).  Compiled with `gcc -std=c11 -Wall -Wextra -Werror -pedantic` against
 * include/fidget_hip.h alone - the header is the only thing it knows of the library - and linked to libfidget_hip.so.
 *
 *   render_hi <model.vm> <golden.txt>
 *
 * builds the shape from its text (Context::from_text -> MathFunction::new), renders it at 32 x 32 through fhip_render2d
 * with a HOST output buffer (the blocking call the Rust trait's render makes) and compares the inside / outside bitmap
 * with the reference's golden ASCII image (fidget/tests/pixel_render.rs:75-106, tests/golden/).  Exit status 0 = equal.
 * `render_hi --symbols` only checks that the program links and loads, `render_hi --layout` prints the config structs'
 * field offsets as this compiler sees them (neither needs a GPU). */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fidget_hip.h"

static char* slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* s = (char*)malloc((size_t)n + 1);
    if (!s || fread(s, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(s); return NULL; }
    s[n] = 0;
    fclose(f);
    return s;
}

/* RawDistancePixel::inside (fidget-raster/src/pixel.rs:187-193): a fill is a NaN carrying 0xF6 in bits 9..16 and the
 * `inside` flag in bit 0 (pixel.rs:225-229); anything else is a distance */
static int pixel_inside(float v) {
    uint32_t bits;
    memcpy(&bits, &v, 4);
    const int is_nan = (bits & 0x7F800000u) == 0x7F800000u && (bits & 0x007FFFFFu) != 0;
    if (is_nan && (bits & (0xFFu << 9)) == (0xF6u << 9)) return (int)(bits & 1u);
    return v < 0.0f;
}

int main(int argc, char** argv) {
    if (argc == 2 && strcmp(argv[1], "--symbols") == 0) {
        /* every entry point this program uses resolved at load time, or we would not be here */
        float m[9];
        const uint32_t size[2] = {32, 32};
        fhip_screen_to_world(size, 2, m);
        printf("linked: screen_to_world[0] = %g\n", (double)m[0]);
        return m[0] == 0.0625f ? 0 : 1;
    }
    if (argc == 2 && strcmp(argv[1], "--layout") == 0) {
        /* the config structs as this C compiler lays them out, for the check against the ctypes mirror (fidget_amd/__init__.py) */
#define F(T, f) printf("\"" #T "." #f "\": [%zu, %zu], ", offsetof(T, f), sizeof(((T*)0)->f))
        printf("{");
        F(fhip_render2d_config, width); F(fhip_render2d_config, height); F(fhip_render2d_config, world_to_model);
        F(fhip_render2d_config, z); F(fhip_render2d_config, pixel_perfect); F(fhip_render2d_config, tile_sizes);
        F(fhip_render2d_config, n_tile_sizes); F(fhip_render2d_config, var_keys); F(fhip_render2d_config, var_values);
        F(fhip_render2d_config, n_vars); F(fhip_render2d_config, axis_slots);
        F(fhip_render3d_config, width); F(fhip_render3d_config, height); F(fhip_render3d_config, depth);
        F(fhip_render3d_config, world_to_model); F(fhip_render3d_config, tile_sizes); F(fhip_render3d_config, n_tile_sizes);
        F(fhip_render3d_config, var_keys); F(fhip_render3d_config, var_values); F(fhip_render3d_config, n_vars);
        F(fhip_render3d_config, axis_slots);
        printf("\"sizeof.fhip_render2d_config\": %zu, \"sizeof.fhip_render3d_config\": %zu}\n", sizeof(fhip_render2d_config),
               sizeof(fhip_render3d_config));
        return 0;
    }
    if (argc != 3) { fprintf(stderr, "usage: %s <model.vm> <golden.txt> | --symbols\n", argv[0]); return 2; }
    char* text = slurp(argv[1]);
    char* gold = slurp(argv[2]);
    if (!text || !gold) { fprintf(stderr, "cannot read the inputs\n"); return 2; }

    fhip_ctx* ctx = NULL;
    fhip_status st = fhip_ctx_create(0, NULL, &ctx);
    if (st != FHIP_OK) { fprintf(stderr, "fhip_ctx_create: status %d (no HIP device, or the library's kernels failed to load)\n", (int)st); return 3; }

    fhip_graph* g = fhip_graph_new();
    const uint32_t root = fhip_graph_from_text(g, text);
    if (root == 0xFFFFFFFFu) { fprintf(stderr, "from_text failed\n"); return 4; }
    fhip_tape* tape = NULL;
    st = fhip_tape_from_graph(ctx, g, &root, 1, &tape);
    if (st != FHIP_OK) { fprintf(stderr, "fhip_tape_from_graph: %s\n", fhip_last_error(ctx)); return 4; }

    enum { W = 32, H = 32 };
    static float image[W * H];
    fhip_render2d_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.width = W; cfg.height = H;      /* everything else: identity transform, z = 0, the HIP shape's tile sizes, no variables */
    st = fhip_render2d(ctx, tape, &cfg, image, 0);
    if (st != FHIP_OK) { fprintf(stderr, "fhip_render2d: %s\n", fhip_last_error(ctx)); return 5; }

    int row = 0, bad = 0;
    for (char* line = strtok(gold, "\n"); line; line = strtok(NULL, "\n")) {
        if (line[0] == '#' || line[0] == 0) continue;
        if (row >= H || (int)strlen(line) != W) { fprintf(stderr, "golden image is not %d x %d\n", W, H); return 6; }
        for (int x = 0; x < W; x++) bad += (line[x] == '#') != pixel_inside(image[row * W + x]);
        row++;
    }
    if (row != H) { fprintf(stderr, "golden image has %d rows\n", row); return 6; }
    for (int y = 0; y < H && bad; y++) {
        for (int x = 0; x < W; x++) fputc(pixel_inside(image[y * W + x]) ? '#' : '.', stderr);
        fputc('\n', stderr);
    }
    printf("%s: %d of %d pixels differ from the golden image (tape: %u ops, %u registers)\n", argv[1], bad, W * H,
           (unsigned)fhip_tape_len(tape), (unsigned)fhip_tape_reg_count(tape));
    fhip_tape_free(tape);
    fhip_graph_free(g);
    fhip_ctx_destroy(ctx);
    free(text);
    free(gold);
    return bad ? 1 : 0;
}
