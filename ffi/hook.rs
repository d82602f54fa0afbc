//! The hook: goes into src/render/renderer.rs.  `fill_image_on_gpu` replaces the closure loop
//! `image.set_each_pixel(|x, y| self.color_at_pixel(x, y), pb)` at renderer.rs:71 (render_animation_frame) and :95 (render).
#[cfg(feature = "mi355x")]
impl Renderer {
    fn fill_image_on_gpu(&self, image: &mut Image, progressbar: Option<indicatif::ProgressBar>) -> Result<(), String> {
        use super::{flatten::{camera_desc, FlatScene}, rtcore_sys::*};
        if unsafe { rt_abi_version() } != RT_ABI_VERSION {
            return Err("librtcore.so: ABI version mismatch".into());
        }
        let flat = FlatScene::new(self.integrator.scene()); // after compute_bvh(): renderer.rs:77 / scene.rs:33
        let desc = flat.desc();
        let cam = camera_desc(&self.camera);
        let n = self.supersampling_offsets.len(); // renderer.rs:24,40-48: at most 5
        let mut opt = rt_options {
            n_offsets: n as u32,
            offsets: [0.; 5],
            max_depth: self.integrator.max_recursive_depth() as u32,         // needs a pub(crate) getter (integrator.rs:25)
            shadow_intensity: self.integrator.use_shadow_intensity() as u32, // likewise (integrator.rs:31)
            reference_order: 0,
            pipeline: 0,
            timing: 0,
            n_devices: if image.width() * image.height() * n * n >= (1 << 24) { -1 } else { 1 }, // big frames: every visible GPU (tiles dealt inside rt_render); small ones: one
            progress: progressbar.as_ref().map(|_| bump_progress as unsafe extern "C" fn(u64, *mut std::ffi::c_void)),
            user: progressbar.as_ref().map_or(std::ptr::null_mut(), |pb| pb as *const indicatif::ProgressBar as *mut std::ffi::c_void),
        };
        opt.offsets[..n].copy_from_slice(&self.supersampling_offsets);
        let mut rgb = vec![0f64; image.width() * image.height() * 3];
        let rc = unsafe { rt_render(&desc, &cam, &opt, rgb.as_mut_ptr(), std::ptr::null_mut()) };
        if rc != 0 {
            let msg = unsafe { std::ffi::CStr::from_ptr(rt_last_error()) }.to_string_lossy().into_owned();
            return Err(format!("rtcore error {rc}: {msg}"));
        }
        // Color is not repr(C) (color.rs:5-12): copy, do not transmute.  Pixel id = y * width + x (image.rs:78-79).
        for (px, c) in image.pixels_mut().iter_mut().zip(rgb.chunks_exact(3)) {
            *px = Color::new(c[0], c[1], c[2]);
        }
        Ok(())
    }
}

/// rt_options.progress: called by the library once per finished batch with the pixels finished since the last call — the
/// unit of the reference's bar (image.rs:73).  ProgressBar is Send + Sync; `user` outlives the rt_render call.
#[cfg(feature = "mi355x")]
unsafe extern "C" fn bump_progress(pixels_done: u64, user: *mut std::ffi::c_void) {
    let pb = &*(user as *const indicatif::ProgressBar);
    pb.inc(pixels_done);
}

// render_animation_frame (renderer.rs:68-73):
//     #[cfg(feature = "mi355x")]
//     self.fill_image_on_gpu(&mut image, progressbar).unwrap();
//     #[cfg(not(feature = "mi355x"))]
//     image.set_each_pixel(|x, y| self.color_at_pixel(x, y), progressbar);
// render (renderer.rs:75-98): the same swap at :95.
//
// ---------------------------------------------------------------------------------------------------------------------
// Animations (animation_renderer.rs:61-66, 98-121): frame i is rendered from scene.animate(time) = clone + animate +
// compute_bvh (scene.rs:30-36).  For a model of thousands of triangles that preparation is 10x the GPU render, so the model's
// group lives on the device (include/rtcore.h "rt_mesh"): a GpuMesh per model is created once per animation, its `pose` stands in for
// the per-triangle part of ObjectGroup::transform + recalculate_bbox, and FlatScene emits the group as ONE RT_REF_MESH
// child.  rt_scene_create then builds the group's subtree on the device — the tree build_bvh would have built, bit for bit.

/// One animated model: an ObjectGroup whose children were all untransformed Triangle / SmoothTriangle primitives.
#[cfg(feature = "mi355x")]
pub struct GpuMesh {
    handle: *mut super::rtcore_sys::rt_mesh,
    palette: Vec<Material>,        // distinct materials of the group's triangles, in first-seen order
    pending: Vec<[f64; 16]>,       // matrices received since the clone, row-major (Matrix::data, matrix.rs)
}

#[cfg(feature = "mi355x")]
impl GpuMesh {
    /// Once, on the scene as loaded: `group.children()` (group.rs:206) are the base pose.
    pub fn new(group: &ObjectGroup) -> Result<Self, String> {
        use super::rtcore_sys::*;
        let n = group.children().len();
        let (mut shape, mut mat) = (Vec::with_capacity(n), Vec::with_capacity(n));
        let (mut p, mut nrm, mut bb) = ([vec![], vec![], vec![]], [vec![], vec![], vec![]], Vec::with_capacity(6 * n));
        let mut palette: Vec<Material> = Vec::new();
        for child in group.children() {
            let prim = child.as_primitive().ok_or("not a triangle group")?;          // object.rs:73-79
            if prim.transformation_inverse().is_some() { return Err("child has a transformation".into()); }
            let (pts, normals, tag) = match prim.shape() {                            // shape.rs:15-32
                Shape::Triangle(t) => ([t.p1(), t.p2(), t.p3()], [t.normal(); 3], RT_SHAPE_TRIANGLE),
                Shape::SmoothTriangle(t) => ([t.p1(), t.p2(), t.p3()], [t.n1(), t.n2(), t.n3()], RT_SHAPE_SMOOTH_TRIANGLE),
                _ => return Err("not a triangle group".into()),
            };
            for k in 0..3 {
                p[k].extend_from_slice(&[pts[k].x(), pts[k].y(), pts[k].z()]);
                nrm[k].extend_from_slice(&[normals[k].x(), normals[k].y(), normals[k].z()]);
            }
            let b = prim.bounding_box();                                              // object.rs:362: transformed by its corners per frame
            bb.extend_from_slice(&[b.min.x(), b.min.y(), b.min.z(), b.max.x(), b.max.y(), b.max.z()]);
            let m = prim.material();
            let idx = palette.iter().position(|q| q == m).unwrap_or_else(|| { palette.push(m.clone()); palette.len() - 1 });
            shape.push(tag as u8);
            mat.push(idx as u32);
        }
        let desc = rt_mesh_desc { n_tris: n as u32, shape: shape.as_ptr(), p1: p[0].as_ptr(), p2: p[1].as_ptr(), p3: p[2].as_ptr(),
                                  n1: nrm[0].as_ptr(), n2: nrm[1].as_ptr(), n3: nrm[2].as_ptr(), bbox: bb.as_ptr(),
                                  material: mat.as_ptr(), n_palette: palette.len() as u32 };
        let mut handle = std::ptr::null_mut();
        if unsafe { rt_mesh_create(&desc, &mut handle) } != 0 { return Err(last_error()); }
        Ok(GpuMesh { handle, palette, pending: Vec::new() })
    }
    /// ObjectGroup::transform (group.rs:242-247) for the device-resident children: remember the matrix.
    pub fn transform(&mut self, m: &Matrix) { self.pending.push(m.to_row_major()); }
    /// ObjectGroup::recalculate_bbox (group.rs:230-235): poses the triangles (all pending matrices, in order, from the base
    /// pose — Scene::animate starts from a clone of the base scene every frame) and returns the union of their boxes.
    pub fn pose(&mut self) -> Result<BoundingBox, String> {
        let mut b = [0f64; 6];
        let flat: Vec<f64> = self.pending.iter().flatten().copied().collect();
        let rc = unsafe { super::rtcore_sys::rt_mesh_pose(self.handle, self.pending.len() as u32, flat.as_ptr(), b.as_mut_ptr()) };
        self.pending.clear();
        if rc != 0 { return Err(last_error()); }
        Ok(BoundingBox { min: Point::new(b[0], b[1], b[2]), max: Point::new(b[3], b[4], b[5]) })
    }
}
#[cfg(feature = "mi355x")]
impl Drop for GpuMesh { fn drop(&mut self) { unsafe { super::rtcore_sys::rt_mesh_destroy(self.handle) } } }

#[cfg(feature = "mi355x")]
fn last_error() -> String { unsafe { std::ffi::CStr::from_ptr(super::rtcore_sys::rt_last_error()) }.to_string_lossy().into_owned() }

// Wiring (three small changes in the reference, all behind the feature):
//  * ObjectGroup gains `mesh: Option<Arc<Mutex<GpuMesh>>>`; AnimationRenderer::new (animation_renderer.rs:33-47) walks the
//    scene once and, for every group of >= 64 untransformed triangles, moves the children into GpuMesh::new(group) and leaves
//    `children` empty.  Clone of the group clones the Arc.
//  * ObjectGroup::transform with `mesh` set calls mesh.transform(matrix) instead of visiting children; recalculate_bbox
//    with `mesh` set takes mesh.pose()?; build_bvh skips a group with `mesh` set (nothing to split on the host).
//  * FlatScene::new emits such a group as RT_MAKE_REF(RT_REF_MESH, k) in its parent's child list and fills
//    desc.mesh[k] = handle, mesh_bbox[k] = group.bounding_box(), mesh_built[k] = every ancestor has <= 2 children
//    (group.rs:146-150: that is when build_bvh recurses into it), mesh_material_begin / mesh_material = the indices this
//    frame's material table gave mesh.palette.
// render_animation_frame then calls fill_image_on_gpu as before — or rt_render_u8 into the frame's byte buffer, which is what
// animation_renderer.rs:98-121 hands the encoder (as_u8_rgb) — holding the mesh's lock from animate() to the return of
// rt_render*: a posed mesh belongs to one scene at a time.  The Python stand-in of exactly this loop
// (raytracer_amd/harness/animation.py over harness/host/scene_capi.cpp) renders rotating-dragon 800x800 at 272-292 frames/s
// with one host thread, against 27 (one thread) / 164 (eight) with the host preparation.
//
// Scenes without such models: prepare frames ahead on a few threads (std::thread::scope over chunks of frame indices, each
// doing `self.renderer.scene().animate(time)`), keep the rt_render calls and encode_fun in frame order.
