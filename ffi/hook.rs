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
// Animations (animation_renderer.rs:98-121) call render_frame once per frame; on the GPU scene.animate(time) costs more
// than the render, so prepare frames ahead on a few threads (std::thread::scope over chunks of frame indices, each doing
// `self.renderer.scene().animate(time)`), keep the rt_render calls and encode_fun in frame order: what
// raytracer_amd/animation.py does (rotating-dragon 800x800: 27 -> 164 frames/s with 8 workers).
