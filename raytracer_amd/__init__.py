"""raytracer_amd — MI355X-native render core for karolzwolak/raytracer's per-pixel hot path.

Layout:
  csrc/        hand-written HIP kernels + the C-ABI of include/rtcore.h  -> lib/librtcore.so
  host/        host-side scene model + flattener (C++)                  -> lib/librtscene.so
  scene.py     mirror of SceneBuilder / Material / Camera over librtscene.so
  scene_yaml.py  YAML scene DSL front end for the harness
  renderer.py  mirror of Renderer::render / render_animation_frame over librtcore.so
"""
__all__ = ["scene", "scene_yaml", "renderer"]
