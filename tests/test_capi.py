"""The C-ABI library loads and exports every symbol include/rtcore.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rtcore.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for n in ("rt_render", "rt_scene_create", "rt_scene_destroy", "rt_render_tiles_device", "rt_render_device",
              "rt_assemble_tiles_device", "rt_quantize_device", "rt_trace_rays", "rt_last_error", "rt_device_count", "rt_tile_count",
              "rt_abi_version"):
        assert n in names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "raytracer_amd", "lib", "librtcore.so"))
    for name in declared_functions():
        assert hasattr(lib, name), f"librtcore.so does not export {name}"
    lib.rt_abi_version.restype = ctypes.c_uint32
    assert lib.rt_abi_version() == 1
    lib.rt_tile_count.restype = ctypes.c_uint32
    lib.rt_tile_count.argtypes = [ctypes.c_uint32] * 4
    assert lib.rt_tile_count(800, 800, 0, 1) == 10000 and lib.rt_tile_count(20, 9, 1, 4) == 2  # 3x2 tiles: part 1 of 4 owns tiles {1, 5}


def test_ctypes_structs_match_the_header(tmp_path):
    """raytracer_amd/_native.py mirrors the structs of include/rtcore.h by hand: their sizes must be what a C compiler sees."""
    import subprocess
    from raytracer_amd import _native as N
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "rtcore.h"\nint main(void) { printf("%zu %zu %zu %zu\\n", sizeof(rt_scene_desc), '
                   'sizeof(rt_camera), sizeof(rt_options), sizeof(rt_stats)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert sizes == [ctypes.sizeof(N.rt_scene_desc), ctypes.sizeof(N.rt_camera), ctypes.sizeof(N.rt_options), ctypes.sizeof(N.rt_stats)]
