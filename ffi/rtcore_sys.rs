//! `#[repr(C)]` mirror of include/rtcore.h, ABI version 3.  Field order and types follow the header exactly;
//! tests/test_capi.py checks the same layout for the ctypes mirror against a C compiler.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const RT_ABI_VERSION: u32 = 3;
pub const RT_NONE: u32 = 0xFFFF_FFFF;
pub const RT_TIMING_ALL: u32 = 1;
pub const RT_TIMING_ONLY: u32 = 0x100;
pub const RT_TIMING_CLASS_TRACE_SHADOW: u32 = 3;
pub const RT_REF_NODE: u32 = 0;
pub const RT_REF_PRIM: u32 = 1;
pub const RT_REF_CSG: u32 = 2;
pub const RT_REF_MESH: u32 = 3;
pub const fn rt_make_ref(kind: u32, index: u32) -> u32 {
    (kind << 30) | (index & 0x3FFF_FFFF)
}

#[repr(C)]
pub struct rt_scene_desc {
    pub abi_version: u32,
    pub n_nodes: u32,
    pub node_bbox: *const f64,
    pub node_child_begin: *const u32,
    pub n_child_refs: u32,
    pub child_ref: *const u32,
    pub n_prims: u32,
    pub prim_shape: *const u8,
    pub prim_material: *const u32,
    pub prim_xform: *const u32,
    pub prim_geom: *const u32,
    pub prim_csg_left: *const u32,
    pub n_xforms: u32,
    pub xform_inv: *const f64,
    pub n_tris: u32,
    pub tri_p1: *const f64,
    pub tri_e1: *const f64,
    pub tri_e2: *const f64,
    pub tri_n1: *const f64,
    pub tri_n2: *const f64,
    pub tri_n3: *const f64,
    pub n_quadrics: u32,
    pub quadric_y_min: *const f64,
    pub quadric_y_max: *const f64,
    pub quadric_closed: *const u8,
    pub n_csgs: u32,
    pub csg_op: *const u8,
    pub csg_left: *const u32,
    pub csg_right: *const u32,
    pub csg_depth: *const u8,
    pub n_materials: u32,
    pub material: *const f64,
    pub pattern_kind: *const u8,
    pub pattern_c1: *const f64,
    pub pattern_c2: *const f64,
    pub pattern_inv: *const f64,
    pub n_lights: u32,
    pub light_pos: *const f64,
    pub light_intensity: *const f64,
    pub n_meshes: u32,
    pub mesh: *const *mut rt_mesh,
    pub mesh_bbox: *const f64,
    pub mesh_built: *const u8,
    pub mesh_material_begin: *const u32,
    pub mesh_material: *const u32,
}

#[repr(C)]
pub struct rt_mesh {
    _opaque: [u8; 0],
}

#[repr(C)]
pub struct rt_mesh_desc {
    pub n_tris: u32,
    pub shape: *const u8,
    pub p1: *const f64,
    pub p2: *const f64,
    pub p3: *const f64,
    pub n1: *const f64,
    pub n2: *const f64,
    pub n3: *const f64,
    pub bbox: *const f64,
    pub material: *const u32,
    pub n_palette: u32,
}

#[repr(C)]
pub struct rt_camera {
    pub width: u32,
    pub height: u32,
    pub inv_view: [f64; 16],
    pub half_width: f64,
    pub half_height: f64,
    pub pixel_size: f64,
}

#[repr(C)]
pub struct rt_options {
    pub n_offsets: u32,
    pub offsets: [f64; 5],
    pub max_depth: u32,
    pub shadow_intensity: u32,
    pub reference_order: u32, // 1: visit the BVH exactly as ObjectGroup::intersect_iter does (validation; slow)
    pub pipeline: u32,        // 0: production (streamed kernels; the single kernel for scenes without secondary rays); 1: single kernel; 2: streamed always
    pub timing: u32,          // RT_TIMING_ALL (with stats): per-kernel HIP-event times instead of counters; RT_TIMING_ONLY | class bits: those classes only
    pub n_devices: i32,       // rt_render: 0/1 = the current device, n > 1 = tiles over the first n devices, -1 = all
    pub progress: Option<unsafe extern "C" fn(pixels_done: u64, user: *mut c_void)>, // pixels finished since the last call
    pub user: *mut c_void,
}

extern "C" {
    pub fn rt_abi_version() -> u32;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_device_count() -> i32;
    /// One-shot render: uploads the scene, renders, downloads `width*height*3` f64 (row-major r,g,b).
    pub fn rt_render(scene: *const rt_scene_desc, camera: *const rt_camera, options: *const rt_options,
                     out_rgb: *mut f64, stats: *mut c_void /* rt_stats*, nullable */) -> i32;
    /// The same with `Color::as_scaled_values` (color.rs:55-66) applied on the device: `width*height*3` bytes, what
    /// `Image::as_u8_rgb` (image.rs:84-89) hands the PNG / video encoders.
    pub fn rt_render_u8(scene: *const rt_scene_desc, camera: *const rt_camera, options: *const rt_options,
                        out_rgb8: *mut u8, stats: *mut c_void) -> i32;
    // resident-scene API (animations, multi-GPU): see include/rtcore.h
    pub fn rt_scene_create(scene: *const rt_scene_desc, out: *mut *mut c_void) -> i32;
    pub fn rt_scene_destroy(scene: *mut c_void);
    pub fn rt_render_device(scene: *const c_void, camera: *const rt_camera, options: *const rt_options,
                            d_out_rgb: *mut c_void, hip_stream: *mut c_void, stats: *mut c_void) -> i32;
    pub fn rt_quantize_device(d_rgb: *const f64, n_values: u64, d_out_u8: *mut u8, hip_stream: *mut c_void) -> i32;
    // device-side per-frame scene preparation (animated triangle groups): see include/rtcore.h
    pub fn rt_mesh_create(desc: *const rt_mesh_desc, out: *mut *mut rt_mesh) -> i32;
    pub fn rt_mesh_destroy(mesh: *mut rt_mesh);
    pub fn rt_mesh_pose(mesh: *mut rt_mesh, n_matrices: u32, matrices: *const f64, out_bbox6: *mut f64) -> i32;
    pub fn rt_shutdown();
}
