/* "No function aborts or throws across the ABI" (include/rustray_hip.h:21-23), seen from a C host: a scene of eight small meshes goes
 * through the host half of rr_scene_create (rr_test_host_build: validation + the threaded per-mesh tree builds) while the test-only
 * fault hook makes the calling thread, then a worker thread, throw.  Every call must RETURN a status; the process must survive.
 * Built and run by tests/test_abi.py; needs no GPU. */
#include "../../include/rustray_hip.h"

#include <stdio.h>
#include <string.h>

int rr_test_fault(const char* point, int kind, int skip);
int rr_test_host_build(const rr_flat_scene* scene, uint64_t* n_nodes_out);

#define CHECK(c) do { if (!(c)) { printf("FAILED %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #c, rr_last_error()); return 1; } } while (0)
#define N_MESH 8
#define GRID 24 /* GRID x GRID quads per mesh */

static float pos[N_MESH][(GRID + 1) * (GRID + 1) * 3];
static uint32_t idx[N_MESH][GRID * GRID * 6];

int main(void) {
    rr_flat_scene fs;
    rr_mesh meshes[N_MESH];
    uint64_t nodes = 0, nodes2 = 0;
    int m, x, y, kind, skip;
    memset(&fs, 0, sizeof fs); memset(meshes, 0, sizeof meshes);
    for (m = 0; m < N_MESH; m++) {
        uint32_t k = 0;
        for (y = 0; y <= GRID; y++)
            for (x = 0; x <= GRID; x++) {
                float* p = &pos[m][((y * (GRID + 1)) + x) * 3];
                p[0] = (float)x; p[1] = (float)((x * 7 + y * 13 + m) % 5) * 0.25f; p[2] = (float)y;
            }
        for (y = 0; y < GRID; y++)
            for (x = 0; x < GRID; x++) {
                uint32_t a = (uint32_t)(y * (GRID + 1) + x), b = a + 1u, c = a + (uint32_t)(GRID + 1), d = c + 1u;
                idx[m][k++] = a; idx[m][k++] = b; idx[m][k++] = c; idx[m][k++] = b; idx[m][k++] = d; idx[m][k++] = c;
            }
        meshes[m].positions = pos[m]; meshes[m].indices = idx[m];
        meshes[m].n_vertices = (GRID + 1) * (GRID + 1); meshes[m].n_triangles = GRID * GRID * 2;
    }
    fs.abi_version = RR_ABI_VERSION; fs.n_meshes = N_MESH; fs.meshes = meshes;
    CHECK(rr_test_host_build(&fs, &nodes) == RR_OK && nodes > 0);
    for (kind = 1; kind <= 3; kind++) {
        CHECK(rr_test_fault("scene_create.host", kind, 0) == RR_OK);
        CHECK(rr_test_host_build(&fs, NULL) == (kind == 1 ? RR_ERR_OUT_OF_MEMORY : RR_ERR_DEVICE));
        CHECK(strstr(rr_last_error(), "rr_test_host_build") != NULL);
        for (skip = 0; skip < N_MESH; skip += 3) {
            CHECK(rr_test_fault("scene_create.mesh_worker", kind, skip) == RR_OK);
            CHECK(rr_test_host_build(&fs, NULL) == (kind == 1 ? RR_ERR_OUT_OF_MEMORY : RR_ERR_DEVICE));
        }
    }
    CHECK(rr_test_fault("", 0, 0) == RR_OK);
    CHECK(rr_test_host_build(&fs, &nodes2) == RR_OK && nodes2 == nodes);
    meshes[3].indices = NULL; /* validation still comes first */
    CHECK(rr_test_host_build(&fs, NULL) == RR_ERR_INVALID_ARGUMENT);
    printf("guard c99 OK\n");
    return 0;
}
