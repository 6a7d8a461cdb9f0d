"""include/mspa.h is a plain C header and libmspa.so a plain C-ABI library: a C99 program compiled with gcc links against it,
reads the version and gets MSPA_EINVAL + a message for a bad call (no GPU needed -- validation precedes any HIP call)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "multi-spatialmllm_amd")

PROGRAM = r"""
#include <stdio.h>
#include <string.h>
#include "mspa.h"
int main(void) {
    if (mspa_version() < 100) return 2;
    if (mspa_pair_overlap(NULL, 1, 1, NULL, 0, NULL, NULL, NULL, NULL) != MSPA_EINVAL) return 3;
    if (!strstr(mspa_last_error_string(), "null pointer")) return 4;
    if (mspa_track_rigidity_loss(NULL, 3, 0, 0.01, NULL, NULL) != MSPA_OK) return 5;      /* empty input: nothing to do */
    if (mspa_object_extents(NULL, 0, 2, NULL, 100, NULL, NULL, 0, 2, NULL, NULL, NULL, NULL) != MSPA_OK) return 6;
    if (mspa_corr_tiles(480, 640) != 100 || mspa_corr_tiles(968, 1296) != 21 * 21 || mspa_corr_tiles(1, 640) != -1) return 7;
    if (mspa_pair_correspondences_workspace_bytes(10, 480, 640, 480, 640, MSPA_PAIR_FAST) != 0) return 8;   /* fused */
    if (mspa_pair_correspondences_workspace_bytes(10, 480, 640, 480, 640, 0) != 10LL * 480 * 640 * 4) return 9;
    if (mspa_pair_correspondences_workspace_bytes(2, 480, 640, 968, 1296, MSPA_PAIR_FAST) != 0) return 10;   /* fused at ScanNet's shape too (round 4) */
    if (mspa_pair_correspondences_workspace_bytes(2, 480, 640, 968, 1296, 0) != 2LL * 968 * 1296 * 4) return 14;
    if (mspa_pair_correspondences_workspace_bytes(2, 60, 81, 121, 162, MSPA_PAIR_FAST) != 2LL * 121 * 162 * 4) return 15;   /* W % 16 != 0: dense route */
    {   /* the host-side helpers that fill the guard-bound slots of the frame / image records */
        double frec[MSPA_FRAME_MATS * 16] = {0}, crec[MSPA_CAM_MATS * 16] = {0};
        int s, k;
        for (s = 0; s < MSPA_MAT_BOUNDS; ++s) for (k = 0; k < 4; ++k) frec[s * 16 + 5 * k] = 1.0;   /* identities */
        for (s = 0; s < MSPA_CAM_BOUNDS; ++s) for (k = 0; k < 4; ++k) crec[s * 16 + 5 * k] = 1.0;
        if (mspa_frame_bounds_host(frec, 1) != MSPA_OK || mspa_camera_bounds_host(crec, 1) != MSPA_OK) return 16;
        if (!(frec[MSPA_MAT_BOUNDS * 16 + 0] == 1.0 && frec[MSPA_MAT_BOUNDS * 16 + 3] == 0.0 && frec[MSPA_MAT_BOUNDS * 16 + 4] > 0.0)) return 17;
        if (!(crec[MSPA_CAM_BOUNDS * 16 + 0] > 0.0 && crec[MSPA_CAM_BOUNDS * 16 + 4] == 0.0)) return 18;
        if (mspa_frame_bounds_host(NULL, 1) != MSPA_EINVAL || mspa_camera_bounds_host(NULL, 1) != MSPA_EINVAL) return 19;
    }
    if (mspa_pair_correspondences(NULL, NULL, 1, NULL, 1, 480, 640, 480, 640, NULL, NULL, NULL, NULL, NULL, 0, MSPA_PAIR_FAST,
                                  NULL) != MSPA_EINVAL) return 11;
    if (!strstr(mspa_last_error_string(), "required")) return 12;
    if (mspa_compact_correspondences(NULL, NULL, 0, 480, 640, NULL, NULL, NULL) != MSPA_OK) return 13;   /* nothing to do */
    printf("ok %d\n", mspa_version());
    return 0;
}
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_header_is_c99_and_library_links_from_c(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libmspa.so")):
        pytest.skip("libmspa.so not built")
    src = tmp_path / "cabi.c"
    src.write_text(PROGRAM)
    exe = str(tmp_path / "cabi")
    build = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                            str(src), "-o", exe, "-L", LIBDIR, "-lmspa", f"-Wl,-rpath,{LIBDIR}"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("ok "), (run.returncode, run.stdout, run.stderr)
