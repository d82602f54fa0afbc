"""Harness, not product: what stands in for the reference's Rust host in THIS repository so that the render core can be
fed and tested without a Rust toolchain — the scene model, transforms, OBJ loader and BVH builder restated in C++
(`host/`, a27 of SURVEY.md §8: bit-identical inverse matrices and the same tree are what parity needs), the YAML scene DSL
(`scene_yaml.py`), the Python mirror of the scene interface (`scene.py`) and an animation driver (`animation.py`).  In the
product all of this stays the reference's Rust; the product is `csrc/` (librtcore.so) behind `include/rtcore.h`."""
