/*
 * rtcore.h — C-ABI of the MI355X render core (librtcore.so).
 *
 * Drop-in boundary for the per-pixel hot path of karolzwolak/raytracer:
 * the reference's pixel loop `image.set_each_pixel(|x,y| self.color_at_pixel(x,y), pb)`
 * (src/render/renderer.rs:71 and :95, src/render/image.rs:66-82) is replaced by one call that
 * fills the f64 RGB pixel array.  The reference has no FFI of its own (pure Rust); every entry point
 * below names the reference interface it replaces.  The Rust-side binding is shown in INTEGRATION.md.
 *
 * Conventions: every real is IEEE f64; Point/Vector/Color are 3 x f64;
 * a Matrix is 16 x f64 row-major exactly as src/math/matrix.rs:8-10,266-270 stores it.
 * All arrays are borrowed for the duration of the call that receives them.
 * Every function returns 0 on success and a negative rt_status otherwise; rt_last_error()
 * gives a thread-local message.  Nothing unwinds or aborts across this boundary.
 */
#ifndef RTCORE_H
#define RTCORE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 3u

/* ---- child references --------------------------------------------------------------------- */
/* A child reference packs the kind of an `Object` (src/scene/object.rs:26-30) in bits 31..30   */
/* and the index into the matching table in bits 29..0.                                         */
#define RT_REF_KIND_SHIFT 30u
#define RT_REF_INDEX_MASK 0x3FFFFFFFu
#define RT_REF_NODE 0u /* ObjectKind::Group  -> index into the node tables  */
#define RT_REF_PRIM 1u /* ObjectKind::Primitive -> index into the prim tables */
#define RT_REF_CSG  2u /* ObjectKind::Csg    -> index into the csg tables   */
#define RT_REF_MESH 3u /* ObjectKind::Group whose children are all triangle primitives and live on the device
                          (rt_mesh below) -> index into rt_scene_desc::mesh */
#define RT_MAKE_REF(kind, index) (((uint32_t)(kind) << RT_REF_KIND_SHIFT) | ((uint32_t)(index) & RT_REF_INDEX_MASK))
#define RT_NONE 0xFFFFFFFFu

/* ---- shape tags: src/scene/object/primitive/shape.rs:15-32 --------------------------------- */
enum rt_shape {
  RT_SHAPE_SPHERE = 0,
  RT_SHAPE_PLANE = 1,
  RT_SHAPE_CUBE = 2,
  RT_SHAPE_CYLINDER = 3,
  RT_SHAPE_CONE = 4,
  RT_SHAPE_TRIANGLE = 5,
  RT_SHAPE_SMOOTH_TRIANGLE = 6,
  RT_SHAPE_BBOX = 7 /* behaves as a cube (shape.rs:48,60) */
};

/* ---- pattern tags: src/scene/object/material/pattern.rs:7-37 ------------------------------- */
enum rt_pattern {
  RT_PATTERN_CONST = 0,
  RT_PATTERN_STRIPE = 1,
  RT_PATTERN_GRADIENT = 2, /* c1 = c_start, c2 = c_dist (= end - start), pattern.rs:48-54 */
  RT_PATTERN_RING = 3,
  RT_PATTERN_CHECKERS = 4,
  RT_PATTERN_TEST = 5
};

/* ---- CSG operations: src/scene/object/csg.rs:29-34 ----------------------------------------- */
enum rt_csg_op { RT_CSG_UNION = 0, RT_CSG_INTERSECTION = 1, RT_CSG_DIFFERENCE = 2 };

enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID_ARGUMENT = -1, /* NULL where data is required, inconsistent counts, bad refs   */
  RT_ERR_UNSUPPORTED = -2,      /* scene exceeds a documented device limit      */
  RT_ERR_DEVICE = -3,           /* a HIP call failed, or no gfx950 device is usable             */
  RT_ERR_OVERFLOW = -4,         /* a per-ray bounded list overflowed while rendering: reported by rt_render and rt_trace_rays
                                   always, by the device entry points only with stats != NULL (they are asynchronous
                                   otherwise); rt_scene_create's limits make it unreachable for a scene it accepted */
  RT_ERR_HOST = -5              /* host memory or threads exhausted (no C++ exception ever leaves the library) */
};

/* Material scalars, in the order of `struct Material` (src/scene/object/material.rs:10-20). */
#define RT_MAT_AMBIENT 0
#define RT_MAT_DIFFUSE 1
#define RT_MAT_SPECULAR 2
#define RT_MAT_SHININESS 3
#define RT_MAT_REFLECTIVITY 4
#define RT_MAT_TRANSPARENCY 5
#define RT_MAT_REFRACTIVE_INDEX 6
#define RT_MAT_STRIDE 8 /* [7] is padding and must be 0 */

/*
 * The flattened, immutable scene: what `Scene::objects()` / `Scene::light_sources()`
 * (src/scene.rs:22-28) hold after `Scene::compute_bvh()` (src/scene.rs:38), written out as tables.
 *
 * Nodes mirror `ObjectGroup` (src/scene/object/group.rs:16-20): node i owns the child references
 * child_ref[node_child_begin[i] .. node_child_begin[i+1]) IN THE ORDER of `ObjectGroup::children`
 * (traversal order decides ties, group.rs:171-201) and the group's own bounding box.
 * Node 0 is the scene root (`Scene::objects`).
 */
typedef struct rt_scene_desc {
  uint32_t abi_version; /* RT_ABI_VERSION */

  /* groups */
  uint32_t n_nodes;
  const double* node_bbox;          /* [n_nodes][6]  min x,y,z then max x,y,z (bounding_box.rs:25-28) */
  const uint32_t* node_child_begin; /* [n_nodes + 1] */
  uint32_t n_child_refs;
  const uint32_t* child_ref;        /* [n_child_refs] RT_MAKE_REF(kind, index) */

  /* primitives: `PrimitiveObject` (object.rs:357-363). The index IS the object identity that the
   * refractive-index container walk compares (intersection.rs:145-153,280-287). */
  uint32_t n_prims;
  const uint8_t* prim_shape;     /* [n_prims] enum rt_shape */
  const uint32_t* prim_material; /* [n_prims] index into the material tables */
  const uint32_t* prim_xform;    /* [n_prims] index into xform_inv, RT_NONE when transformation_inverse is None */
  const uint32_t* prim_geom;     /* [n_prims] triangle index (shapes 5,6), quadric index (shapes 3,4), else RT_NONE */
  const uint32_t* prim_csg_left; /* [n_prims] bit d set <=> `csg.left.includes(prim)` (csg.rs:93, object.rs:98-104)
                                    for the enclosing CSG object whose nesting depth is d (outermost = 0); 0 outside CSG */

  uint32_t n_xforms;
  const double* xform_inv; /* [n_xforms][16] `transformation_inverse` exactly as the reference holds it */

  /* triangles: Triangle (triangle.rs:10-17) and SmoothTriangle (smooth_triangle.rs:13-22), world space */
  uint32_t n_tris;
  const double* tri_p1; /* [n_tris][3] */
  const double* tri_e1; /* [n_tris][3] p2 - p1 */
  const double* tri_e2; /* [n_tris][3] p3 - p1 */
  const double* tri_n1; /* [n_tris][3] smooth: n1;  flat: the precomputed unit normal */
  const double* tri_n2; /* [n_tris][3] smooth: n2;  flat: unused (0) */
  const double* tri_n3; /* [n_tris][3] smooth: n3;  flat: unused (0) */

  /* cylinders and cones: cylinder.rs:14-18, cone.rs:13-17 */
  uint32_t n_quadrics;
  const double* quadric_y_min;   /* [n_quadrics] may be -inf */
  const double* quadric_y_max;   /* [n_quadrics] may be +inf */
  const uint8_t* quadric_closed; /* [n_quadrics] 0/1 */

  /* CSG objects: csg.rs:68-74 */
  uint32_t n_csgs;
  const uint8_t* csg_op;      /* [n_csgs] enum rt_csg_op */
  const uint32_t* csg_left;   /* [n_csgs] child reference */
  const uint32_t* csg_right;  /* [n_csgs] child reference */
  const uint8_t* csg_depth;   /* [n_csgs] CSG nesting depth of this object (outermost = 0), < 32 */

  /* materials: material.rs:10-20 + pattern.rs:7-37 */
  uint32_t n_materials;
  const double* material;       /* [n_materials][RT_MAT_STRIDE] */
  const uint8_t* pattern_kind;  /* [n_materials] enum rt_pattern */
  const double* pattern_c1;     /* [n_materials][3] */
  const double* pattern_c2;     /* [n_materials][3] */
  const double* pattern_inv;    /* [n_materials][16] inverse pattern transform (identity for CONST) */

  /* lights: point_light.rs:7-10 */
  uint32_t n_lights;
  const double* light_pos;       /* [n_lights][3] */
  const double* light_intensity; /* [n_lights][3] */

  /* device-resident triangle groups (rt_mesh, below), referenced from child_ref as RT_MAKE_REF(RT_REF_MESH, k): the group's
   * children are NOT in the tables above; rt_scene_create transforms nothing (rt_mesh_pose has) and, on the device, builds the
   * subtree ObjectGroup::build_bvh (group.rs:144-167) makes of that group.  0 meshes: everything as before. */
  uint32_t n_meshes;
  struct rt_mesh* const* mesh;         /* [n_meshes] posed meshes, each referenced exactly once, outside CSG objects */
  const double* mesh_bbox;             /* [n_meshes][6] the group's ObjectGroup::bounding_box as the host's tree holds it: after
                                          Scene::animate that is the union rt_mesh_pose returned (group.rs:230-241); on the
                                          `image` path whatever the group's own transforms left (group.rs:263-270 transforms
                                          the box by its corners).  It positions the candidate planes of the group's first split */
  const uint8_t* mesh_built;           /* [n_meshes] 1: build_bvh reached this group (every ancestor has at most two children
                                          after the build, group.rs:145-156); 0: it keeps its flat child list */
  const uint32_t* mesh_material_begin; /* [n_meshes + 1] into mesh_material */
  const uint32_t* mesh_material;       /* palette entry p of mesh k -> index into this scene's material tables */
} rt_scene_desc;

/* `Camera` (src/scene/camera.rs:9-18); the three derived scalars are computed by the caller exactly
 * as camera.rs:166-175 does (they are private fields there). */
typedef struct rt_camera {
  uint32_t width, height;
  double inv_view[16];
  double half_width, half_height, pixel_size;
} rt_camera;

/* `Renderer::supersampling_offsets` (renderer.rs:24,40-48), `Integrator::max_recursive_depth` and
 * `Integrator::use_shadow_intensity` (integrator.rs:19-32). */
typedef void (*rt_progress_fn)(uint64_t pixels_done, void* user);

/* rt_options.timing */
#define RT_TIMING_ALL 1u
#define RT_TIMING_ONLY 0x100u
#define RT_TIMING_CLASS_TRACE_CLOSEST 0
#define RT_TIMING_CLASS_RETRACE 1
#define RT_TIMING_CLASS_SHADE 2
#define RT_TIMING_CLASS_TRACE_SHADOW 3
#define RT_TIMING_CLASS_COMBINE 4
#define RT_TIMING_CLASS_OTHER 5

typedef struct rt_options {
  uint32_t n_offsets;        /* 1..5 */
  double offsets[5];
  uint32_t max_depth;        /* max_recursive_depth; device limit RT_MAX_DEPTH */
  uint32_t shadow_intensity; /* 1: production (intensity sums); 0: boolean shadows of the unit tests */
  uint32_t reference_order;  /* 0: production traversal (near child first, LDS stack).  1: visit the BVH exactly in the
                                order of ObjectGroup::intersect_iter (group.rs:171-201): validation and counting mode —
                                same pixels, slower; its node/triangle counts are the reference's algorithmic work */
  uint32_t pipeline;         /* RT_PIPELINE_STREAM (0, production): rays streamed through trace / shade / combine kernels —
                                except for a scene without triangles none of whose materials reflects or transmits (every ray
                                tree is a primary ray and its shadow rays; the README scene): that one is rendered by the
                                single kernel, whose one launch is quicker than the pipeline's six (same pixels; with `timing`
                                set the pipeline runs all the same).
                                RT_PIPELINE_SINGLE_KERNEL (1): one kernel holds a pixel's whole recursion (same pixels, bit
                                for bit; kept for comparison and as the overflow fallback of the streamed pipeline).
                                RT_PIPELINE_STREAM_ALWAYS (2): the streamed pipeline whatever the scene.
                                Ignored when reference_order = 1 */
  uint32_t timing;           /* with stats != NULL and the streamed pipeline: launch the production kernels (no counters; the
                                counter fields of rt_stats stay 0) with a HIP event before and after every launch, and report
                                the time per kernel class in rt_stats.ms_*.  RT_TIMING_ALL (1): every launch (~90 events per
                                frame: ~0.3 ms of a 4K frame).  RT_TIMING_ONLY | (1 << RT_TIMING_CLASS_x) | ...: only the
                                launches of the classes named (the other ms_* fields stay 0) */
  int32_t n_devices;         /* rt_render only (the reference's host is ONE process with a sequential frame loop,
                                animation_renderer.rs:109-122, so this is how it reaches more than one GPU): 0 or 1 = the current
                                HIP device; n > 1 = the frame's 8x8 tiles are dealt cyclically to the first n visible devices
                                (fewer if fewer are visible), each renders its part from its own copy of the scene, the parts are
                                copied to the first device, un-interleaved there and downloaded; -1 = all visible devices */
  rt_progress_fn progress;   /* nullable.  Called with the number of pixels finished since the previous call (the unit of the
                                reference's progress bar, image.rs:73: `progress_with(pb)` over the pixel iterator), from a thread
                                of the HIP runtime, once per finished batch of tiles; the calls of one render sum to
                                width * height.  With n_devices > 1 several runtime threads report; the library serialises the
                                calls (one at a time, in any order).  Must not call back into this library */
  void* user;                /* passed to `progress` */
} rt_options;
#define RT_PIPELINE_STREAM 0u
#define RT_PIPELINE_SINGLE_KERNEL 1u
#define RT_PIPELINE_STREAM_ALWAYS 2u

#define RT_MAX_DEPTH 16u

/* Per-render ray statistics (unique rays: each reflect/refract subtree is traced once per hit). */
typedef struct rt_stats {
  uint64_t rays_primary, rays_shadow, rays_reflect, rays_refract;
  uint64_t node_visits, tri_tests, prim_tests;
  uint64_t hits;      /* shaded hits (one IntersecComputations each) */
  uint64_t hit_bytes; /* algorithmic bytes of those hits: normals/inverse + material (+ pattern) + lights, SURVEY.md §8d */
  uint64_t overflow_pixels;
  uint64_t wave_steps;  /* production traversal only: wavefront-level traversal steps ... */
  uint64_t lane_steps;  /* ... and the lanes active in them (lane_steps / (64 * wave_steps) = SIMD utilisation) */
  uint64_t rays_retraced; /* production traversal only: rays whose hit was settled by a reference-order walk because two
                             surfaces coincide within 1e-10 (see DESIGN.md "Ties") */
  uint64_t clocks_traversal, clocks_machine; /* production traversal, statistics launches only: shader clocks summed over
                             lanes spent in the traversal loop / in the shading state machine */
  uint64_t clocks_trace_total; /* streamed pipeline, statistics launches: wavefront clocks of the trace kernels from start to
                              end, summed over wavefronts; there clocks_traversal = wavefront clocks in the node loop and
                              clocks_machine = wavefront clocks in leaves (the rest: fetching rays, setting them up, storing
                              results, and the drain after the queue is dry) */
  uint64_t shadow_node_visits, shadow_tri_tests, shadow_prim_tests; /* streamed pipeline: the part of node_visits / tri_tests /
                              prim_tests done by trace_shadow_kernel */
  uint64_t shadow_wave_steps, shadow_lane_steps, shadow_clocks_node, shadow_clocks_leaf, shadow_clocks_total; /* likewise */
  uint64_t rays_shadow_culled; /* streamed pipeline: (hit, light) pairs whose shadow ray was NOT traced (and is not in rays_shadow)
                              because the light is behind the surface: there point_light.rs:50-55 returns the ambient term
                              without using the shadow value.  rays_shadow + rays_shadow_culled = the reference's shadow rays */
  uint64_t launches_trace_closest, launches_trace_shadow; /* rt_options.timing: launches behind the ms_* sums below */
  uint64_t fallback_flags; /* streamed pipeline: times a ray or hit found its batch's queues full; the samples concerned were
                              re-rendered by the single-kernel state machine (same pixels, slower).  0 in a well-sized batch */
  double ms_upload, ms_kernel, ms_download;
  double ms_trace_closest, ms_retrace, ms_shade, ms_trace_shadow, ms_combine, ms_other; /* rt_options.timing: GPU time of the
                              streamed pipeline's launches by kernel class, HIP events around each launch, summed over the frame.
                              The shadow rays of the first 1-3 levels run on a second stream of the scene's own beside the next
                              levels' kernels (ordered against the caller's stream by events; nothing of a frame is still running
                              when the work this call enqueued on `hip_stream` has completed): an event pair around such a launch
                              also times its wait for wavefront slots, the classes' spans overlap, and their sum can exceed
                              ms_kernel.  RT_SHADOW_SIDE_SAMPLES=0 in the environment (read per batch) keeps every kernel on
                              `hip_stream`, one after the other: a class's time is then its kernels' own */
} rt_stats;

/* ---- library ------------------------------------------------------------------------------ */
uint32_t rt_abi_version(void);
const char* rt_last_error(void); /* thread-local, static storage */
int rt_device_count(void);       /* number of usable gfx950 devices, <0 on error */

/* ---- one-shot render: replaces Image::set_each_pixel(|x,y| color_at_pixel(x,y))
 *      at renderer.rs:71,95 / image.rs:66-82.
 * out_rgb: caller-owned HOST buffer, width*height*3 f64, row-major (id = y*width + x, image.rs:78-79),
 * channels r,g,b.  Uploads the scene to the current HIP device, renders, copies back. */
int rt_render(const rt_scene_desc* scene, const rt_camera* camera, const rt_options* options,
              double* out_rgb, rt_stats* stats /* nullable */);

/* The same render with the image quantised on the device: `Color::as_scaled_values` (src/math/color.rs:55-66) per channel,
 * i.e. what `Image::as_u8_rgb` (image.rs:84-89) hands the PNG / PPM / video encoders.  out_rgb8: caller-owned HOST buffer,
 * width*height*3 bytes, row-major, r,g,b.  3 bytes per pixel instead of 24 come back over PCIe (and, with n_devices > 1, travel
 * between the devices).  The bytes equal as_scaled_values of rt_render's f64 pixels exactly. */
int rt_render_u8(const rt_scene_desc* scene, const rt_camera* camera, const rt_options* options,
                 uint8_t* out_rgb8, rt_stats* stats /* nullable */);

/* ---- resident-scene API (what an animation loop or a multi-GPU driver uses) ----------------- */
typedef struct rt_scene rt_scene; /* opaque: the scene re-laid-out in device memory */

/* Validates the tables, builds the device layout and uploads it to the current HIP device.
 * Replaces nothing in the reference (there the scene is already in host memory); it is the
 * "hand a flattened scene to the kernels" step of the north star. */
int rt_scene_create(const rt_scene_desc* scene, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

/* ---- device-side per-frame scene preparation -------------------------------
 * An animation renders frame i from Scene::animate(time) = clone + animate + compute_bvh (scene.rs:30-36,
 * animation_renderer.rs:61-66): every frame the host transforms every triangle of an animated model
 * (PrimitiveObject::transform, object.rs:454-472) and rebuilds the tree over them (ObjectGroup::build_bvh, group.rs:87-167).
 * An rt_mesh keeps such a group — one whose children are all triangle primitives without a transformation of their own: what
 * obj_model.rs makes of an OBJ file or of one of its `g` groups — on the device, and both steps run there:
 *   rt_mesh_create   once: the triangles as the scene holds them BEFORE any animation (the clone's starting point).
 *   rt_mesh_pose     per frame: Object::transform(m_0), ..., transform(m_{k-1}) applied to every triangle in this order from
 *                    the base pose — the matrices the group received on its way down Object::animate (object.rs:108-121,
 *                    group.rs:242-247, 263-270), in the reference's per-vertex operation order (triangle.rs:24-47,
 *                    smooth_triangle.rs:29-55; boxes by their 8 corners, bounding_box.rs:47-63) — and the group's
 *                    recalculate_bbox (group.rs:230-235) handed back, which the caller's own (small) tree above needs.
 *   rt_scene_create  with the mesh referenced from the scene's tree (RT_REF_MESH): builds the reference's subtree of the
 *                    group on the device, bit for bit the tree build_bvh builds (same candidate planes, same strict
 *                    comparisons, stable partitions), and lays it out for the kernels.
 * A posed mesh belongs to ONE scene at a time: pose, create the scene, and only then pose for the next frame (the scene has
 * its own copy of everything by then).  Not thread-safe per mesh.  A mesh lives on the device that was current when it was
 * created: a scene that refers to it is created on that device (rt_options.n_devices > 1 with meshes is refused with
 * RT_ERR_INVALID_ARGUMENT; a host with several GPUs keeps one mesh per device and model). */
typedef struct rt_mesh rt_mesh;
typedef struct rt_mesh_desc {
  uint32_t n_tris;
  const uint8_t* shape;     /* [n_tris] RT_SHAPE_TRIANGLE or RT_SHAPE_SMOOTH_TRIANGLE */
  const double* p1;         /* [n_tris][3] Triangle::p1/p2/p3 (triangle.rs:10-17; e1, e2 and a flat normal are derived) */
  const double* p2;
  const double* p3;
  const double* n1;         /* [n_tris][3] SmoothTriangle::n1/n2/n3 (ignored for flat triangles; may be NULL without smooth ones) */
  const double* n2;
  const double* n3;
  const double* bbox;       /* [n_tris][6] PrimitiveObject::bbox (object.rs:362): it is transformed by its corners, not recomputed */
  const uint32_t* material; /* [n_tris] palette entry of each triangle's material, < n_palette */
  uint32_t n_palette;
} rt_mesh_desc;
#define RT_MESH_MAX_MATRICES 8u
int rt_mesh_create(const rt_mesh_desc* desc, rt_mesh** out);
void rt_mesh_destroy(rt_mesh* mesh);
/* matrices: [n_matrices][16] row-major, n_matrices <= RT_MESH_MAX_MATRICES (0: the base pose); out_bbox6: min xyz, max xyz.
 * Synchronous (it returns the box). */
int rt_mesh_pose(rt_mesh* mesh, uint32_t n_matrices, const double* matrices, double* out_bbox6);

/* Debugging / tests: copies one of a scene's device tables to the host.  which: 0 tri_isect, 1 tri_norm, 2 prim_rank,
 * 3 prim_info, 4 entries, 5 nodes, 6 prim_flags.  Returns the table's size in bytes (negative rt_status on error); copies
 * min(size, capacity) bytes. */
long long rt_scene_debug_table(const rt_scene* scene, int which, void* out, unsigned long long capacity);

/* Debugging / tests: the traversal's conservative f32 box test beside the exact f64 slab test of the reference
 * (BoundingBox::intersection_time, bounding_box.rs:192-225) that it stands in for, for n (ray, box, closest-hit-so-far) triples:
 * origin / direction [n][3] f64, boxes [n][6] f32 (min xyz, max xyz), limit [n] f64 (all HOST arrays); all_t = 1: the container
 * walk's variant (boxes behind the origin are wanted too).  out[i]: bit 0 the f32 test keeps the box, bit 1 the exact test
 * does, bit 2 the ray is one production tests exactly anyway (an axis-parallel direction).  The property the renderer rests
 * on: bit 1 implies bit 0 wherever bit 2 is clear. */
int rt_debug_box_tests(uint32_t n, const double* origin, const double* direction, const float* boxes, const double* limit, int all_t,
                       uint8_t* out);

/* Library housekeeping, no counterpart in the reference.  Destroyed scenes leave their device blocks (table arena,
 * render workspace) in a process-wide pool for the next rt_scene_create / render on the same device: an animation makes
 * one scene per frame (animation_renderer.rs:61-66) and would otherwise pay hipMalloc / hipFree, which also
 * synchronise the device, every frame.  The pool holds at most 24 blocks / 96 GiB (RT_POOL_MAX_BYTES) and is emptied
 * when an allocation fails; this call empties it on request. */
void rt_release_cached_memory(void);

/* Tears down what the library keeps for itself: waits for the current device, empties the pools above and
 * the pinned staging blocks.  Scenes are the caller's to destroy first.  The library can be used again afterwards. */
void rt_shutdown(void);

/* Tile partition for multi-GPU frames: the image is cut into 8x8-pixel tiles in
 * row-major tile order; part p of n_parts owns the tiles with tile_id % n_parts == p.  The output
 * is PACKED: the k-th owned tile occupies d_out_rgb[k*64*3 .. (k+1)*64*3), pixel (tx,ty) of the tile
 * at offset (ty*8+tx)*3.  rt_tile_count() gives the number of owned tiles.  With n_parts == 1 use
 * rt_render_device() to get the plain row-major image instead. */
#define RT_TILE_DIM 8u
uint32_t rt_tile_count(uint32_t width, uint32_t height, uint32_t part, uint32_t n_parts);

/* Renders part `part` of `n_parts` into DEVICE memory on `hip_stream` (a hipStream_t, may be NULL).
 * Asynchronous with respect to the host unless stats != NULL (then it synchronises the stream to read
 * the counters).  Same per-pixel semantics as rt_render: Renderer::color_at_pixel (renderer.rs:50-66). */
int rt_render_tiles_device(const rt_scene* scene, const rt_camera* camera, const rt_options* options,
                           uint32_t part, uint32_t n_parts, double* d_out_tiles, void* hip_stream,
                           rt_stats* stats /* nullable */);

/* Scatters packed tiles of all parts (in part order, as a gather delivers them) into a row-major
 * width*height*3 image, both in device memory.  The un-interleave step after the RCCL gather.
 * part_stride_tiles: distance between the blocks of consecutive parts, in tiles (a gather of equally sized,
 * padded buffers); 0 means the blocks are tightly concatenated (block p holds rt_tile_count(.., p, n_parts) tiles). */
int rt_assemble_tiles_device(uint32_t width, uint32_t height, uint32_t n_parts, uint32_t part_stride_tiles,
                             const double* d_tiles_all_parts, double* d_out_rgb, void* hip_stream);

/* The same for tiles that were quantised before the gather (rt_quantize_device on each part's packed tiles: the order of the
 * values is the same, 1 byte each): packed u8 tiles of all parts -> row-major width*height*3 bytes. */
int rt_assemble_tiles_u8_device(uint32_t width, uint32_t height, uint32_t n_parts, uint32_t part_stride_tiles,
                                const uint8_t* d_tiles_all_parts, uint8_t* d_out_rgb8, void* hip_stream);

/* Whole image, row-major, into DEVICE memory (n_parts == 1 path; no assembly step needed). */
int rt_render_device(const rt_scene* scene, const rt_camera* camera, const rt_options* options,
                     double* d_out_rgb, void* hip_stream, rt_stats* stats /* nullable */);

/* rt_render_device for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  It has no counterpart in the reference.  Same pixels as rt_render_device (streamed production
 * pipeline; options->reference_order and options->pipeline must be 0), but the ray queues, hit and node records, shadow
 * values and counters of the pipeline live in ONE static device array of RT_NOALLOC_SCRATCH_BYTES per GPU instead of a
 * workspace the library allocates, and the call makes no hipMalloc / hipFree / hipHostMalloc / hipEventCreate, nor
 * anything else that allocates: only kernel launches and hipMemsetAsync, ALL on `hip_stream` (the scene's side stream, which
 * the allocating entry points use for small batches, is not used here), then one hipEventRecord of the scene's own event on
 * `hip_stream`, which is what rt_scene_destroy waits for.  While `hip_stream` is being captured into a HIP graph
 * (hipStreamIsCapturing) that event record is left out, and so is every synchronisation of the error paths: the caller then
 * keeps the scene alive until the graph's launches have finished.  A frame larger
 * than the array is rendered in batches of 8x8 tiles; more than RT_NOALLOC_MAX_BATCHES batches (or a scene with so many
 * lights that one tile's batch does not fit) is refused with RT_ERR_UNSUPPORTED and a message giving the limit in
 * samples (about 90 000 samples per batch for a 2-light scene: 1200x1200 at supersampling 2 fits).  Every call on a GPU
 * uses that GPU's array: calls that overlap on one GPU (other streams, other threads, other scenes) race; the array keeps
 * the previous call's contents, everything that has to start at zero is cleared by the call itself. */
#define RT_NOALLOC_SCRATCH_BYTES (64u * 1024u * 1024u)
#define RT_NOALLOC_MAX_BATCHES 64u
int rt_render_device_noalloc(const rt_scene* scene, const rt_camera* camera, const rt_options* options,
                             double* d_out_rgb, void* hip_stream);

/* `Color::as_scaled_values` (src/math/color.rs:55-66) + `Image::as_u8_rgb` (image.rs:84-89) on device:
 * n_pixels*3 f64 -> n_pixels*3 u8 (round-half-away(clamp(v,0,1)*255), NaN -> 0). */
int rt_quantize_device(const double* d_rgb, uint64_t n_values, uint8_t* d_out_u8, void* hip_stream);

/* Traces arbitrary rays: `Integrator::color_at(Ray)` (src/shading/integrator.rs:68) for n rays given
 * as HOST arrays origin[n][3], direction[n][3]; writes out_rgb[n][3] (host).  Used by the parity tests to
 * replay the reference's unit-test rays through the device path. */
int rt_trace_rays(const rt_scene* scene, const rt_options* options, uint32_t n, const double* origin,
                  const double* direction, double* out_rgb);

#ifdef __cplusplus
}
#endif
#endif /* RTCORE_H */
