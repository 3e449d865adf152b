/* include/rustray_hip.h must be a plain C header (the boundary a Rust `extern "C"` block or any C host binds):
 * compiled as C99 by tests/test_abi.py and linked against librustray_hip.so.  Runs without a GPU: only the entry
 * points that validate on the host are exercised. */
#include "../../include/rustray_hip.h"

#include <stdio.h>
#include <string.h>

#define CHECK(c) do { if (!(c)) { printf("FAILED %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #c, rr_last_error()); return 1; } } while (0)

int main(void) {
    uint16_t xy[2 * 16];
    uint32_t cell = 0;
    rr_flat_scene fs;
    rr_scene* scene = NULL;
    rr_camera cam;
    rr_config cfg;
    rr_frame frame;
    rr_region region = {32u, 8u, 8u, 0u};
    rr_tuning tuning;
    uint64_t total = 0;
    uint32_t r;

    CHECK(sizeof(rr_camera) == 136 && sizeof(rr_config) == 40 && sizeof(rr_region) == 16 && sizeof(rr_tuning) == 40);
    CHECK(rr_device_count() >= 0);
    CHECK(rr_sample_table(16, xy, &cell) == RR_OK && cell == 16u);
    CHECK(rr_sample_table(16, NULL, &cell) == RR_ERR_INVALID_ARGUMENT);
    for (r = 0; r < 8u; r++) { region.rank = r; total += rr_region_pixel_count(1280u, 720u, &region); }
    CHECK(total == 1280u * 720u);

    memset(&fs, 0, sizeof fs);
    CHECK(rr_scene_create(&fs, 0, &scene) == RR_ERR_INVALID_ARGUMENT && scene == NULL);   /* abi_version 0 */
    CHECK(strstr(rr_last_error(), "abi_version") != NULL);
    fs.abi_version = RR_ABI_VERSION;                                                        /* an empty scene is valid ... */
    if (rr_device_count() == 0) CHECK(rr_scene_create(&fs, 0, &scene) == RR_ERR_NO_DEVICE); /* ... but needs a device: no fallback */

    memset(&cam, 0, sizeof cam); memset(&cfg, 0, sizeof cfg); memset(&frame, 0, sizeof frame); memset(&tuning, 0, sizeof tuning);
    CHECK(rr_render(NULL, &cam, &cfg, NULL, &frame, NULL) == RR_ERR_INVALID_ARGUMENT);
    CHECK(rr_render_multi(NULL, 0u, &cam, &cfg, NULL, &frame, NULL) == RR_ERR_INVALID_ARGUMENT);
    CHECK(rr_scene_update_materials(NULL, NULL, 0u) == RR_ERR_INVALID_ARGUMENT);
    CHECK(rr_scene_set_tuning(NULL, &tuning) == RR_ERR_INVALID_ARGUMENT);
    CHECK(rr_pick(NULL, &cam, 0, 0, NULL) == RR_ERR_INVALID_ARGUMENT);
    rr_scene_destroy(NULL);
    printf("abi c99 OK\n");
    return 0;
}
