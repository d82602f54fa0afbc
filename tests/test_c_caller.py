"""A caller of the C-ABI that is not Python: tests/c_caller/render_dump.c fills the structs of include/rtcore.h by field name
and calls rt_render(scene, camera, options, out, NULL) single-threaded, as ffi/hook.rs does (which cannot be compiled here:
no Rust toolchain).  Also: ffi/rtcore_sys.rs must declare the same fields in the same order as the header."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, N

EXE = os.path.join(ROOT, "tests", "c_caller", "render_dump")
LIB = os.path.join(ROOT, "raytracer_amd", "lib", "librtcore.so")
TABLES = ["node_bbox", "node_child_begin", "child_ref", "prim_shape", "prim_material", "prim_xform", "prim_geom", "prim_csg_left",
          "xform_inv", "tri_p1", "tri_e1", "tri_e2", "tri_n1", "tri_n2", "tri_n3", "quadric_y_min", "quadric_y_max", "quadric_closed",
          "csg_op", "csg_left", "csg_right", "csg_depth", "material", "pattern_kind", "pattern_c1", "pattern_c2", "pattern_inv",
          "light_pos", "light_intensity"]
DTYPES = {"node_child_begin": np.uint32, "child_ref": np.uint32, "prim_shape": np.uint8, "prim_material": np.uint32, "prim_xform": np.uint32,
          "prim_geom": np.uint32, "prim_csg_left": np.uint32, "quadric_closed": np.uint8, "csg_op": np.uint8, "csg_left": np.uint32,
          "csg_right": np.uint32, "csg_depth": np.uint8, "pattern_kind": np.uint8}


def dump_scene(path, flat, cam, opt, n_devices=0):
    with open(path, "wb") as f:
        f.write(b"RTSCENE1")
        for name in TABLES:
            a = np.ascontiguousarray(flat.array(name), dtype=DTYPES.get(name, np.float64))
            f.write(struct.pack("<Q", a.nbytes))
            f.write(a.tobytes())
        f.write(struct.pack("<II", cam.width, cam.height))
        f.write(struct.pack("<19d", *cam.inv_view, cam.half_width, cam.half_height, cam.pixel_size))
        f.write(struct.pack("<I5dII", opt.n_offsets, *opt.offsets, opt.max_depth, opt.shadow_intensity))
        f.write(struct.pack("<i", n_devices))


def _load(name, **kw):
    from raytracer_amd.harness import scene_yaml as Y
    rs = Y.load_scene(os.path.join(GOLDEN, "scenes", name + ".yml"), **kw)
    return rs.builder.flatten(), rs.camera, rs.options()


def test_c_caller_without_a_gpu_fails_loudly(tmp_path):
    """No device here: the C caller gets a negative status and a message, not a CPU render (and not a crash)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    flat, cam, opt = _load("images/chapters/making-scene", width=16, height=16, depth=2, supersampling_level=1)
    dump_scene(tmp_path / "s.bin", flat, cam, opt)
    r = subprocess.run([EXE, LIB, str(tmp_path / "s.bin"), str(tmp_path / "o.f64")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "rt_render failed: -3" in r.stderr
    assert not os.path.exists(tmp_path / "o.f64")


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,n_devices", [
    ("images/chapters/refractions", dict(width=120, height=90, depth=5, supersampling_level=2), 0),
    ("animations/general/csg-lens", dict(width=96, height=96, depth=8, supersampling_level=2), -1),
    ("images/general/dragons", dict(width=160, height=120, depth=4, supersampling_level=2), 0)])
def test_c_caller_image_equals_oracle(tmp_path, name, kw, n_devices):
    import oracle as O
    flat, cam, opt = _load(name, **kw)
    dump_scene(tmp_path / "s.bin", flat, cam, opt, n_devices)
    r = subprocess.run([EXE, LIB, str(tmp_path / "s.bin"), str(tmp_path / "o.f64"), "progress", str(tmp_path / "o.u8")], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "o.f64", dtype=np.float64).reshape(cam.height, cam.width, 3)
    got8 = np.fromfile(tmp_path / "o.u8", dtype=np.uint8).reshape(cam.height, cam.width, 3)
    assert np.array_equal(got8, O.quantize(got))  # rt_render_u8 == Color::as_scaled_values of rt_render's pixels, through the C-ABI
    want, _ = O.render(flat, cam, opt)
    assert float(np.abs(got - want).max()) < 1e-9
    assert np.array_equal(O.quantize(got), O.quantize(want))
    m = re.search(r"progress_calls (\d+) progress_pixels (\d+)", r.stdout)
    assert m and int(m.group(1)) >= 1 and int(m.group(2)) == cam.width * cam.height  # rt_options.progress: image.rs:73's unit


def _c_struct_fields(text, name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        base = decl.split(",")
        for part in base:
            ident = re.findall(r"([A-Za-z_][A-Za-z_0-9]*)\s*(?:\[[^\]]*\])?\s*$", part.strip())
            fields.append(ident[0])
    return fields


def _rust_struct_fields(text, name):
    body = re.search(r"pub struct %s \{(.*?)\n\}" % name, text, flags=re.S).group(1)
    return re.findall(r"pub ([a-z_0-9]+)\s*:", body)


def test_rust_ffi_declarations_follow_the_header():
    """ffi/rtcore_sys.rs (never compiled here) against include/rtcore.h: same ABI version, same fields in the same order
    for every struct the hook fills; the ctypes mirror likewise."""
    h = open(os.path.join(ROOT, "include", "rtcore.h")).read()
    rs = open(os.path.join(ROOT, "ffi", "rtcore_sys.rs")).read()
    abi = int(re.search(r"#define RT_ABI_VERSION (\d+)u", h).group(1))
    assert int(re.search(r"RT_ABI_VERSION: u32 = (\d+);", rs).group(1)) == abi == N.RT_ABI_VERSION
    for name, ctype in (("rt_scene_desc", N.rt_scene_desc), ("rt_camera", N.rt_camera), ("rt_options", N.rt_options)):
        c_fields = _c_struct_fields(h, name)
        assert _rust_struct_fields(rs, name) == c_fields, name
        assert [f for f, _ in ctype._fields_] == c_fields, name
    assert [f for f, _ in N.rt_stats._fields_] == _c_struct_fields(h, "rt_stats")
