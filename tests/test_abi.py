"""The drop-in boundary: librustray_hip.so loads and exports every function include/*.h declares;
argument validation works without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rustray_amd import capi
from rustray_amd.flat import FlatScene, Item, Material, RR_ITEM_SPHERE, rr_flat_scene
from tests.helpers import ROOT, load_scene


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rustray_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rr_[a-z_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    lib = C.CDLL(capi.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rustray_hip.h but not exported"
    assert set(capi.EXPORTS) <= set(names)


def test_abi_version_is_reported_and_checked():
    """rr_frame_stats grew without a size field.  The layouts are versioned instead: the header's RR_ABI_VERSION,
    the library's rr_abi_version() and the binding's constant agree, and a flat scene that names another version is refused
    before anything is written through a caller's pointers."""
    from rustray_amd import flat
    src = open(os.path.join(ROOT, "include", "rustray_hip.h")).read()
    header_version = int(re.search(r"#define RR_ABI_VERSION (\d+)u", src).group(1))
    L = capi.lib()
    L.rr_abi_version.restype = C.c_uint32
    assert header_version == flat.RR_ABI_VERSION == L.rr_abi_version() == 3
    # rr_frame_stats as the header lists it: 17 8-byte fields of round 2 + 4 u32 + 1 double of the multi-GPU exchange + 4 8-byte level-1 fields (ABI 3)
    assert C.sizeof(flat.rr_frame_stats) == 17 * 8 + 4 * 4 + 8 + 4 * 8


def test_render_multi_locks_handles_in_address_order():
    """Two rr_render_multi calls that share handles in opposite orders must take the handle mutexes in ONE order
    (ADVICE r2 / VERDICT r2 item 4).  rr_multi_lock_order reports that order from the pointer values alone."""
    L = capi.lib()
    L.rr_multi_lock_order.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    fake = [0x7000, 0x1000, 0x9000, 0x3000]   # never dereferenced
    for perm in ([0, 1, 2, 3], [3, 2, 1, 0], [2, 0, 3, 1]):
        handles = (C.c_void_p * 4)(*[fake[i] for i in perm])
        order = (C.c_uint32 * 4)()
        assert L.rr_multi_lock_order(handles, 4, order) == 0
        locked = [handles[i] for i in order]
        assert locked == sorted(fake), (perm, list(order))
    assert L.rr_multi_lock_order(None, 0, None) == -1


def test_struct_sizes_match_header():
    # sizes implied by the header's field lists (natural alignment)
    from rustray_amd import flat
    assert C.sizeof(flat.rr_material) == 9 * 4 + 7 * 4 + 8 * 4 + 8
    assert C.sizeof(flat.rr_item) == 6 * 4 + 32 * 4 + 6 * 4 + 4
    assert C.sizeof(flat.rr_light) == 11 * 4 + 4 + 4
    assert C.sizeof(flat.rr_camera) == 8 + 128
    assert C.sizeof(flat.rr_config) == 8 + 6 * 4 + 4 + 4
    assert C.sizeof(flat.rr_region) == 16 and C.sizeof(flat.rr_pick_result) == 16
    assert C.sizeof(flat.rr_tuning) == 40


def test_sample_count_limit():
    """u16 `(samples + 2).next_power_of_two()` of the reference overflows beyond 32766 (RR_MAX_SAMPLES_WITH_TABLE: accepted with the
    caller's table, tests/test_gpu_limits.py); the built-in table stops at RR_MAX_SAMPLES."""
    xy = np.zeros((20000, 2), np.uint16)
    cs = C.c_uint32(0)
    assert capi.lib().rr_sample_table(C.c_uint16(16383), xy.ctypes.data_as(C.c_void_p), C.byref(cs)) == -2
    assert "samples" in capi.lib().rr_last_error().decode()
    got, cell = capi.sample_table(6)
    assert cell == 4 and len(got) == 6 and len({(int(a), int(b)) for a, b in got}) == 6


def test_region_pixel_count_partitions_the_frame():
    for (w, h, tw, th, n) in ((1280, 720, 32, 8, 8), (100, 37, 32, 8, 3), (7, 5, 8, 8, 2), (33, 9, 32, 8, 4)):
        counts = [capi.region_pixel_count(w, h, tw, th, n, r) for r in range(n)]
        assert sum(counts) == w * h
    assert capi.region_pixel_count(10, 10, 0, 8, 1, 0) == 0  # invalid region -> 0


def _create(fs_struct):
    h = C.c_void_p(None)
    rc = capi.lib().rr_scene_create(C.byref(fs_struct), 0, C.byref(h))
    return rc, h, capi.lib().rr_last_error().decode()


def test_scene_validation_without_gpu():
    """Invalid scenes are rejected with RR_ERR_INVALID_ARGUMENT before any device is touched; a valid one
    fails with RR_ERR_NO_DEVICE on a machine without a GPU (never a silent fallback)."""
    fs = load_scene("spheres")
    bad = fs.c_struct()
    bad.abi_version = 99
    rc, _, msg = _create(bad)
    assert rc == -1 and "abi_version" in msg
    fs2 = load_scene("spheres")
    fs2.items[0].material = 1000
    rc, _, msg = _create(fs2.c_struct())
    assert rc == -1 and "material" in msg
    fs3 = load_scene("monkey")
    fs3.meshes[0].indices = fs3.meshes[0].indices.copy()
    fs3.meshes[0].indices[5, 1] = 10 ** 6
    rc, _, msg = _create(fs3.c_struct())
    assert rc == -1 and "vertex index" in msg
    fs4 = load_scene("spheres")
    fs4.materials[fs4.items[0].material_cache].texture[0] = 0
    rc, _, msg = _create(fs4.c_struct())
    assert rc == -1 and "material_cache" in msg
    if capi.device_count() == 0:
        rc, _, msg = _create(load_scene("spheres").c_struct())
        assert rc == -3 and "no HIP device" in msg
    assert capi.lib().rr_scene_create(None, 0, C.byref(C.c_void_p())) == -1


def test_render_multi_validates_before_touching_a_device():
    from rustray_amd.flat import rr_camera, rr_config, rr_frame
    L = capi.lib()
    assert L.rr_render_multi(None, 0, C.byref(rr_camera()), C.byref(rr_config()), None, C.byref(rr_frame()), None) == -1
    handles = (C.c_void_p * 2)(None, None)
    assert L.rr_render_multi(handles, 2, C.byref(rr_camera()), C.byref(rr_config()), None, C.byref(rr_frame()), None) == -1
    assert "NULL" in L.rr_last_error().decode()


def test_header_is_plain_c99_and_links(tmp_path):
    """include/rustray_hip.h compiled as C99 (-pedantic -Werror) into a C host that links the library and runs the
    host-side validation paths (tests/native/abi_c99.c)."""
    import subprocess
    exe = str(tmp_path / "abi_c99")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "native", "abi_c99.c"),
                           "-L" + libdir, "-lrustray_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "abi c99 OK" in out.stdout, out.stdout + out.stderr


# ---- nothing unwinds across the ABI: exceptions on the calling thread and in host worker threads -------------
def _host_build(fs, fault=None):
    """rr_test_host_build = the host half of rr_scene_create (validation + threaded mesh tree builds), reachable without a GPU."""
    L = capi.lib()
    L.rr_test_fault.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.rr_test_host_build.argtypes = [C.POINTER(rr_flat_scene), C.POINTER(C.c_uint64)]
    cs = fs.c_struct()
    n = C.c_uint64(0)
    if fault:
        assert L.rr_test_fault(fault[0].encode(), fault[1], fault[2]) == 0
    try:
        rc = L.rr_test_host_build(C.byref(cs), C.byref(n))
    finally:
        L.rr_test_fault(b"", 0, 0)
    return rc, n.value, L.rr_last_error().decode()


def test_exceptions_do_not_cross_the_abi():
    """The host is Rust with panic = "abort" (reference Cargo.toml:9-11): a C++ exception must come back as a status code.
    std::bad_alloc -> RR_ERR_OUT_OF_MEMORY, any other exception -> RR_ERR_DEVICE with what() in rr_last_error(), on the calling
    thread AND inside the worker threads that build the per-mesh trees (a worker that throws used to end in std::terminate)."""
    fs = load_scene("kbert_room")   # several meshes: the tree builds run on several threads
    assert len(fs.meshes) >= 4
    rc, nodes, _ = _host_build(fs)
    assert rc == 0 and nodes > 0
    # calling thread
    rc, _, msg = _host_build(fs, ("scene_create.host", 1, 0))
    assert rc == -5 and "rr_test_host_build" in msg and "memory" in msg
    rc, _, msg = _host_build(fs, ("scene_create.host", 2, 0))
    assert rc == -4 and "injected fault at scene_create.host" in msg
    rc, _, msg = _host_build(fs, ("scene_create.host", 3, 0))
    assert rc == -4 and "unknown exception" in msg
    # worker threads: the first mesh any worker picks up, and a later one (other workers are mid-build then)
    for skip in (0, 2):
        rc, _, msg = _host_build(fs, ("scene_create.mesh_worker", 1, skip))
        assert rc == -5, (skip, rc, msg)
        rc, _, msg = _host_build(fs, ("scene_create.mesh_worker", 2, skip))
        assert rc == -4 and "injected fault" in msg, (skip, rc, msg)
    # and the library still works afterwards (no thread left behind, no lock held)
    rc, nodes2, _ = _host_build(fs)
    assert rc == 0 and nodes2 == nodes


def test_every_entry_point_is_guarded():
    """Every `extern "C" int rr_*` definition in rr_api.hip is a function-try-block closed by RR_GUARD_END (the no-throw promise of
    include/rustray_hip.h:21-23 is structural, not case by case)."""
    src = open(os.path.join(ROOT, "rustray_amd", "csrc", "rr_api.hip")).read()
    names = re.findall(r'extern "C" int (rr_[a-z_]+)\(', src)
    assert len(names) >= 20
    exempt = {"rr_test_fault", "rr_exp_util"}   # noexcept by construction (atomics and a strcpy) / developer build only
    for n in names:
        if n in exempt:
            continue
        assert re.search(r'extern "C" int ' + n + r'\([^{]*\) try \{', src), f"{n} is not a function-try-block"
        assert f'RR_GUARD_END("{n}")' in src, f"{n} has no RR_GUARD_END"
    assert "std::thread> pool" not in src and "threads.emplace_back(work" not in src   # worker threads only through `Workers`


def test_guard_from_a_c_host(tmp_path):
    """The same through a C program (no Python frames in between): a throwing worker must not terminate the process."""
    import subprocess
    exe = str(tmp_path / "guard_c99")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "native", "guard_c99.c"),
                           "-L" + libdir, "-lrustray_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "guard c99 OK" in out.stdout, out.stdout + out.stderr


def test_every_declared_symbol_is_in_the_rust_block():
    """INTEGRATION.md section 1 is the binding a maintainer would paste into src/hip_ffi.rs (no Rust toolchain here, so it cannot be
    compiled): every function the header declares must appear in its `extern "C"` block, with the header's number of parameters."""
    hdr = open(os.path.join(ROOT, "include", "rustray_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index('#[link(name = "rustray_hip")]'):]
    block = block[:block.index("```")]
    block = re.sub(r"//[^\n]*", "", block)
    # the callback type of rr_render_progressive has parameters of its own: fold nested parentheses away before counting commas
    def n_params(arglist):
        depth, flat = 0, []
        for ch in arglist:
            if ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            elif depth == 0:
                flat.append(ch)
        t = "".join(flat).strip()
        return 0 if t in ("", "void") else t.count(",") + 1
    def arglist_after(text, pos):
        depth, i = 1, pos
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        return text[pos:i - 1]
    for name in declared_functions():
        m = re.search(r"\b" + name + r"\s*\(", hdr)
        want = n_params(arglist_after(hdr, m.end()))
        r = re.search(r"pub fn " + name + r"\s*\(", block)
        assert r, f"{name} is declared in the header but missing from INTEGRATION.md's extern block"
        got = n_params(arglist_after(block, r.end()))
        assert got == want, f"{name}: header has {want} parameters, INTEGRATION.md {got}"
