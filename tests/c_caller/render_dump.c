/* render_dump.c — a plain C caller of the C-ABI (include/rtcore.h), standing in for the reference-side hook that cannot be
 * compiled here (ffi/hook.rs; no Rust toolchain): it reads a flattened scene from a file, fills rt_scene_desc / rt_camera /
 * rt_options BY FIELD NAME through the header, calls rt_render(scene, camera, options, out, NULL) single-threaded exactly as
 * the hook does, and writes the f64 image.  tests/test_c_caller.py compares that image with the oracle's: an ABI or
 * struct-layout mistake that the hand-written ctypes mirror would hide shows up here.
 *
 * File format (written by tests/test_c_caller.py::dump_scene, native endianness): the magic "RTSCENE1", then for every
 * table in the order of TABLES below: u64 byte count + the bytes; then the camera (2 x u32, 19 x f64), the options
 * (u32 n_offsets, 5 x f64, u32 max_depth, u32 shadow_intensity) and u32 n_devices.
 *
 *   render_dump <librtcore.so> <scene.bin> <out.f64> [progress]
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtcore.h"

typedef int (*rt_render_fn)(const rt_scene_desc*, const rt_camera*, const rt_options*, double*, rt_stats*);
typedef int (*rt_render_u8_fn)(const rt_scene_desc*, const rt_camera*, const rt_options*, uint8_t*, rt_stats*);
typedef const char* (*rt_last_error_fn)(void);
typedef uint32_t (*rt_abi_version_fn)(void);

static void* read_table(FILE* f, uint64_t* n_bytes) {
  if (fread(n_bytes, sizeof(*n_bytes), 1, f) != 1) { fprintf(stderr, "short file\n"); exit(2); }
  void* p = malloc(*n_bytes ? *n_bytes : 1);
  if (*n_bytes && fread(p, 1, *n_bytes, f) != *n_bytes) { fprintf(stderr, "short table\n"); exit(2); }
  return p;
}

static uint64_t g_pixels = 0, g_calls = 0;
static void on_progress(uint64_t pixels_done, void* user) {
  (void)user;
  __atomic_add_fetch(&g_pixels, pixels_done, __ATOMIC_RELAXED);
  __atomic_add_fetch(&g_calls, 1, __ATOMIC_RELAXED);
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: render_dump <librtcore.so> <scene.bin> <out.f64> [progress] [out.u8]\n"); return 2; }
  void* lib = dlopen(argv[1], RTLD_NOW);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  rt_render_fn render = (rt_render_fn)dlsym(lib, "rt_render");
  rt_last_error_fn last_error = (rt_last_error_fn)dlsym(lib, "rt_last_error");
  rt_abi_version_fn abi = (rt_abi_version_fn)dlsym(lib, "rt_abi_version");
  if (!render || !last_error || !abi) { fprintf(stderr, "missing symbols\n"); return 2; }
  if (abi() != RT_ABI_VERSION) { fprintf(stderr, "ABI version mismatch: library %u, header %u\n", abi(), RT_ABI_VERSION); return 2; }
  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 2; }
  char magic[8];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "RTSCENE1", 8) != 0) { fprintf(stderr, "bad magic\n"); return 2; }
  rt_scene_desc d;
  memset(&d, 0, sizeof(d));
  d.abi_version = RT_ABI_VERSION;
  uint64_t n;
#define TABLE(field, type, per) d.field = (const type*)read_table(f, &n); 
#define COUNTED(count, field, type, per) d.field = (const type*)read_table(f, &n); d.count = (uint32_t)(n / (sizeof(type) * (per)));
  COUNTED(n_nodes, node_bbox, double, 6)
  TABLE(node_child_begin, uint32_t, 1)
  COUNTED(n_child_refs, child_ref, uint32_t, 1)
  COUNTED(n_prims, prim_shape, uint8_t, 1)
  TABLE(prim_material, uint32_t, 1)
  TABLE(prim_xform, uint32_t, 1)
  TABLE(prim_geom, uint32_t, 1)
  TABLE(prim_csg_left, uint32_t, 1)
  COUNTED(n_xforms, xform_inv, double, 16)
  COUNTED(n_tris, tri_p1, double, 3)
  TABLE(tri_e1, double, 3)
  TABLE(tri_e2, double, 3)
  TABLE(tri_n1, double, 3)
  TABLE(tri_n2, double, 3)
  TABLE(tri_n3, double, 3)
  COUNTED(n_quadrics, quadric_y_min, double, 1)
  TABLE(quadric_y_max, double, 1)
  TABLE(quadric_closed, uint8_t, 1)
  COUNTED(n_csgs, csg_op, uint8_t, 1)
  TABLE(csg_left, uint32_t, 1)
  TABLE(csg_right, uint32_t, 1)
  TABLE(csg_depth, uint8_t, 1)
  COUNTED(n_materials, material, double, RT_MAT_STRIDE)
  TABLE(pattern_kind, uint8_t, 1)
  TABLE(pattern_c1, double, 3)
  TABLE(pattern_c2, double, 3)
  TABLE(pattern_inv, double, 16)
  COUNTED(n_lights, light_pos, double, 3)
  TABLE(light_intensity, double, 3)
  rt_camera cam;
  memset(&cam, 0, sizeof(cam));
  double cd[19];
  if (fread(&cam.width, 4, 1, f) != 1 || fread(&cam.height, 4, 1, f) != 1 || fread(cd, 8, 19, f) != 19) { fprintf(stderr, "short camera\n"); return 2; }
  memcpy(cam.inv_view, cd, 16 * sizeof(double));
  cam.half_width = cd[16];
  cam.half_height = cd[17];
  cam.pixel_size = cd[18];
  rt_options opt;
  memset(&opt, 0, sizeof(opt));
  int32_t n_devices = 0;
  if (fread(&opt.n_offsets, 4, 1, f) != 1 || fread(opt.offsets, 8, 5, f) != 5 || fread(&opt.max_depth, 4, 1, f) != 1 ||
      fread(&opt.shadow_intensity, 4, 1, f) != 1 || fread(&n_devices, 4, 1, f) != 1) { fprintf(stderr, "short options\n"); return 2; }
  opt.n_devices = n_devices;
  fclose(f);
  if (argc > 4) {
    opt.progress = on_progress;
    opt.user = NULL;
  }
  size_t n_values = (size_t)cam.width * cam.height * 3;
  double* out = (double*)malloc(n_values * sizeof(double));
  int rc = render(&d, &cam, &opt, out, NULL); /* stats = NULL: the hook's call */
  if (rc != 0) { fprintf(stderr, "rt_render failed: %d: %s\n", rc, last_error()); return 1; }
  FILE* g = fopen(argv[3], "wb");
  if (!g || fwrite(out, sizeof(double), n_values, g) != n_values) { perror(argv[3]); return 2; }
  fclose(g);
  if (argc > 5) { /* the same frame as the writers' bytes: rt_render_u8 (Image::as_u8_rgb, image.rs:84-89, on the device) */
    rt_render_u8_fn render_u8 = (rt_render_u8_fn)dlsym(lib, "rt_render_u8");
    if (!render_u8) { fprintf(stderr, "missing rt_render_u8\n"); return 2; }
    uint8_t* out8 = (uint8_t*)malloc(n_values);
    opt.progress = NULL;
    rc = render_u8(&d, &cam, &opt, out8, NULL);
    if (rc != 0) { fprintf(stderr, "rt_render_u8 failed: %d: %s\n", rc, last_error()); return 1; }
    FILE* g8 = fopen(argv[5], "wb");
    if (!g8 || fwrite(out8, 1, n_values, g8) != n_values) { perror(argv[5]); return 2; }
    fclose(g8);
  }
  printf("rendered %ux%u nodes %u prims %u tris %u progress_calls %llu progress_pixels %llu\n", cam.width, cam.height, d.n_nodes, d.n_prims,
         d.n_tris, (unsigned long long)g_calls, (unsigned long long)g_pixels);
  return 0;
}
