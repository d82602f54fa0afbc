// rt_device.h — device-side f64 helpers and scene layout for the gfx950 render core.
//
// Operation order follows the reference exactly (no fma contraction: the file is compiled with
// -ffp-contract=off), so EPSILON / acne-offset semantics match src/math/approx_eq.rs.
// Reference lines are cited per function; nothing here is shared with oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtcore.h"

#define RT_EPS 1.0e-8  // src/math/approx_eq.rs:1

struct d3 {
  double x, y, z;
};

__device__ __forceinline__ d3 mk3(double x, double y, double z) { return d3{x, y, z}; }
__device__ __forceinline__ d3 add3(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 sub3(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 neg3(d3 a) { return d3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ d3 scale3(d3 a, double f) { return d3{f * a.x, f * a.y, f * a.z}; }  // vector.rs:136-146
__device__ __forceinline__ d3 cmul3(d3 a, d3 b) { return d3{a.x * b.x, a.y * b.y, a.z * b.z}; }  // color.rs:114-124
__device__ __forceinline__ double dot3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vector.rs:77-79
__device__ __forceinline__ d3 cross3(d3 a, d3 b) {                                               // vector.rs:69-75
  return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double mag3(d3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // vector.rs:54-56
__device__ __forceinline__ d3 normalize3(d3 a) {                                                  // vector.rs:57-67
  double len = mag3(a);
  if (len == 0.) return d3{0., 0., 0.};
  return d3{a.x / len, a.y / len, a.z / len};
}
// vector.rs:81-83: *self - normal * 2. * self.dot(normal)
__device__ __forceinline__ d3 reflect3(d3 v, d3 n) { return sub3(v, scale3(scale3(n, 2.), dot3(v, n))); }

__device__ __forceinline__ bool approx_eq(double a, double b) { return a == b || fabs(a - b) < RT_EPS; }  // approx_eq.rs:16-20
__device__ __forceinline__ bool sign_positive(double x) { return !__builtin_signbit(x); }

// Matrix * Tuple (matrix.rs:316-337): the w term is kept (it can turn -0 into +0, and 1/±0 differs).
__device__ __forceinline__ d3 mat_point(const double* __restrict__ m, d3 p) {
  return d3{m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 1., m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 1.,
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 1.};
}
__device__ __forceinline__ d3 mat_vector(const double* __restrict__ m, d3 p) {
  return d3{m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 0., m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 0.,
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 0.};
}
// Matrix::mul_transposed (matrix.rs:77-92) on a vector
__device__ __forceinline__ d3 mat_transposed_vector(const double* __restrict__ m, d3 p) {
  return d3{m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12] * 0., m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13] * 0.,
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] * 0.};
}

// ---------------------------------------------------------------------------------------------
// Device scene layout (built by rt_scene_create from rt_scene_desc; see DESIGN.md "Data layout in HBM")
// ---------------------------------------------------------------------------------------------

// --- fast path (rt_fast.h): 4-wide nodes, 128 B (one cache line), one dependent fetch per traversal step --------
// Traversal is bound by the latency of that dependent fetch (about one L2 miss per wavefront step), so a node holds the
// boxes of up to FOUR children: the reference's binary tree with every other level folded away (the boxes of the folded
// level are simply not tested), which halves the number of dependent fetches per ray.  The boxes are the reference's f64
// boxes rounded OUTWARD to f32: a superset of each box, tested with the reference's f64 slab arithmetic, so no box is
// culled that the exact box would keep.  Groups of the reference that are not "two sub-groups" are rewritten at scene
// creation: a group of primitives becomes a LEAF reference in its parent (no node of its own); n-ary or mixed groups
// become short chains of binary nodes (whose synthetic boxes repeat the group's own box) before the fold.
#define RT_FREF_INNER 0u  // index = node
#define RT_FREF_TRIS 1u   // index = first triangle of a run in tri_isect/tri_norm, count = run length
#define RT_FREF_PRIMS 2u  // index = first entry of a run in leaf_ref (RT_REF_PRIM / RT_REF_CSG references)
#define RT_FREF_EMPTY 3u
#define RT_FREF_KIND_SHIFT 30u
#define RT_FREF_COUNT_SHIFT 24u
#define RT_FREF_COUNT_MASK 0x3Fu
#define RT_FREF_INDEX_MASK 0x00FFFFFFu
#define RT_FREF_LEAF_MAX 63u
#define RT_FREF_NONE 0xFFFFFFFFu  // kind EMPTY: nothing left to visit
#define RT_MAKE_FREF(kind, count, index) (((uint32_t)(kind) << RT_FREF_KIND_SHIFT) | ((uint32_t)(count) << RT_FREF_COUNT_SHIFT) | (uint32_t)(index))

#define RT_NODE_WIDTH 4
#ifndef RT_QNODE
#define RT_QNODE 0
#endif
#if RT_QNODE
// EXPERIMENT, off by default (-DRT_QNODE=1 builds it; parity is exact either way).  64-B node: the four child boxes
// QUANTISED to 8 bits per coordinate on a grid of the node's own (per axis: origin = the smallest child minimum rounded
// down to f32, step = a power of two with 255 steps covering the children), minima rounded down and maxima up: again a
// superset of each box.  A node step fetches 64 B in four loads instead of 112 B in seven.  Measured on the 4K bench frame:
// 36.74 ms against 35.81 ms for the f32 node — the looser boxes visit 3 % more nodes and the unpacking (one convert per
// coordinate) costs more issue slots than the three saved loads give back; the kernels wait on the LATENCY of the dependent
// fetch, which is the same for 64 and 128 bytes.
struct __attribute__((aligned(64))) DevNode {
  float org[3];         // grid origin
  uint32_t exps;        // byte k: biased f32 exponent of axis k's grid step (step = 2^(byte - 127))
  uint32_t ref[RT_NODE_WIDTH];  // RT_MAKE_FREF; unused slots hold RT_FREF_NONE and come last
  uint32_t q[6];        // q[k] byte c: child c's box coordinate k in grid steps (k = 0..2 min x,y,z; 3..5 max x,y,z)
  uint32_t pad[2];
};
#else
struct __attribute__((aligned(128))) DevNode {
  float bb[RT_NODE_WIDTH][6];   // child boxes: min x,y,z (rounded down), max x,y,z (rounded up) (bounding_box.rs:25-28)
  uint32_t ref[RT_NODE_WIDTH];  // RT_MAKE_FREF; unused slots hold RT_FREF_NONE and come last
  uint32_t pad[RT_NODE_WIDTH];
};
#endif

// Triangles are stored in leaf order, RT_TRI_STRIDE doubles each (80 B, 16-B aligned): p1, e1, e2, then the
// primitive index in the low 32 bits of the tenth slot (bit RT_TRI_OPAQUE_BIT: its material is opaque).
#define RT_TRI_STRIDE 10

// --- reference-order path (rt_reforder.h): one record per child reference of the input tree -------
// 64 B, 16-B aligned: a node visit reads its children's records contiguously; a group child's bounding box and
// child range come with its record.  Record 0 is the scene root.
struct __attribute__((aligned(16))) DevEntry {
  double bb[6];    // bounding box of the child when it is a group (bounding_box.rs:25-28)
  uint32_t ref;    // RT_MAKE_REF(kind, index)
  uint32_t begin;  // group child: first record of ITS children
  uint32_t count;  // group child: number of children
  uint32_t pad;
};

// One entry of a PRIMS leaf, self-contained: testing an analytic primitive is ONE 128-B fetch instead of the chain
// leaf_ref -> prim_info -> xform_inv / quadric.  Rows 0-2 of the inverse transform (row 3 is 0,0,0,1 and is not read by
// the intersection code); triangles and CSG objects inside such leaves keep the table path (flags say so).
#define RT_LEAFREC_INLINE 1u   // flags: shape / matrix / y range below are valid (an analytic primitive)
#define RT_LEAFREC_XFORM 2u    // flags: the primitive has a transformation_inverse (object.rs:432-439)
#define RT_LEAFREC_CLOSED 4u   // flags: cylinder / cone is closed
#define RT_LEAFREC_OPAQUE 8u   // flags: the material's transparency is 0: a shadow ray's hit counts 1 without reading the material
#define RT_TRI_OPAQUE_BIT 32   // the same, bit of a triangle record's tenth slot (the primitive index is its low word)
#define RT_MAT_OPAQUE 0xFFFFFFFFu  // add_roots' material argument: transparency 0, no table to read
struct __attribute__((aligned(128))) DevLeafRec {
  uint32_t ref;       // RT_MAKE_REF(kind, index), as in leaf_ref
  uint32_t shape;     // enum rt_shape
  uint32_t material;
  uint32_t flags;
  double inv[12];     // rows 0..2 of transformation_inverse
  double y_min, y_max;
};

// A CSG object of two analytic primitives with at most two roots each (csg-lens.yml's lens and carved cube): both operands
// with everything their tests read, side by side, so that the object costs ONE 256-B fetch behind its leaf record instead of
// the csg_left -> prim_info -> transform / quadric chain per operand and the hit list in scratch (fast::add_csg2).
// left.flags bits RT_CSG2_OP_SHIFT..: enum rt_csg_op.  left.ref / right.ref: the operands' primitive indices.
#define RT_LEAFREC_CSG2 64u  // DevLeafRec::flags of a CSG leaf entry: DevLeafRec::material indexes DevScene::csg2_rec
#define RT_CSG2_OP_SHIFT 16u
struct __attribute__((aligned(128))) DevCsg2Rec {
  DevLeafRec left, right;
};

struct DevScene {
  const DevNode* nodes;       // fast path; node 0 is a synthetic root whose child 0 is the scene's root group
  const uint32_t* leaf_ref;   // fast path: primitive / CSG references of PRIMS leaves
  const DevLeafRec* leaf_rec; // fast path: the same entries with everything an analytic primitive's test reads, 128 B each
  const DevCsg2Rec* csg2_rec; // fast path: two-primitive CSG objects (RT_LEAFREC_CSG2)
  const uint32_t* prim_rank;  // visit order of each primitive in ObjectGroup::intersect_iter (decides ties in t)
  const DevEntry* entries;    // reference-order path
  const uint4* prim_info;     // x shape, y material, z xform (RT_NONE = none), w geom (triangles: index in leaf order)
  const uint32_t* prim_csg_left;
  const uint8_t* prim_flags;  // bit 0: the primitive's material has !approx_eq(transparency, 0) (intersection.rs:184-190) and the
                              // scene has a refractive index other than 1.0: the container walk follows a closest hit on it
  const double* xform_inv;  // [n][16]
  const double* tri_isect;  // [n][RT_TRI_STRIDE]  p1, e1, e2, prim
  const double* tri_norm;   // [n][9]  n1, n2, n3
  const double* quadric;    // [n][2]  y_min, y_max
  const uint8_t* quadric_closed;
  const uint8_t* csg_op;
  const uint8_t* csg_depth;
  const uint32_t* csg_left;
  const uint32_t* csg_right;
  const double* material;  // [n][8]
  const uint8_t* pattern_kind;
  const double* pattern_c1;
  const double* pattern_c2;
  const double* pattern_inv;  // [n][16]
  const double* light_pos;
  const double* light_intensity;
  uint32_t n_lights;
  uint32_t n_prims;
  uint32_t has_walks;  // 0: no closest hit of this scene is followed by the container walk (prim_flags all clear), so the
                       // refractive indices of every hit are (1.0, 1.0) and the streamed pipeline neither stores nor reads them
};

struct DevOptions {
  uint32_t n_offsets;
  double offsets[5];
  uint32_t max_depth;
  uint32_t shadow_intensity;
};
