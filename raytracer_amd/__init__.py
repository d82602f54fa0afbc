"""raytracer_amd — MI355X-native render core for karolzwolak/raytracer's per-pixel hot path.

Layout:
  csrc/          hand-written HIP kernels + the C-ABI of include/rtcore.h  -> lib/librtcore.so   (the product)
  renderer.py    mirror of Renderer::render / render_animation_frame over librtcore.so
  distributed.py tile / frame partition over one process per GPU
  harness/       NOT product: stands in for the reference's Rust host (scene model, BVH builder, YAML DSL, animation
                 driver) so that the core can be fed and tested here                -> lib/librtscene.so
"""
__all__ = ["renderer", "distributed", "harness"]
