// rt_mesh.h — device-side per-frame scene preparation (include/rtcore.h "rt_mesh").
//
// What the reference does on the host for every frame of an animation — Scene::animate = clone + animate + compute_bvh
// (src/scene.rs:30-36, src/render/animation_renderer.rs:61-66) — for a group of triangle primitives, as gfx950 kernels:
//
//   pose_kernel      PrimitiveObject::transform (object.rs:454-472) per triangle and matrix, in the reference's operation
//                    order: Triangle / SmoothTriangle::transform_new (triangle.rs:24-47, smooth_triangle.rs:29-55), the box
//                    by its 8 corners (bounding_box.rs:47-63); then the group's recalculate_bbox (group.rs:230-235).
//   bin / choose / count / scatter, once per tree level
//                    ObjectGroup::build_bvh (group.rs:144-167) for ALL groups of a level at once.  The children of a group
//                    are a contiguous segment of one order array, because ObjectGroup::split (group.rs:131-143) is a STABLE
//                    partition.  Every quantity of choose_split / calculate_split_cost (group.rs:87-129) is independent of
//                    accumulation order — box unions are exact min / max, counts are integers, half_area x count follows
//                    from them — so the 15 candidates' sides are accumulated with atomics per (axis, class of the centre
//                    among the 5 candidate planes) and combined by ONE thread per group, which walks the candidates in the
//                    reference's fixed order with its strict `<`, applies the "not cheaper than sah_cost -> stop" rule
//                    (group.rs:153-156) and records the split.  The partition is a scan of the "centre below the plane"
//                    flags.
//   fold / finalize  the tables the render kernels read (rt_device.h): the 4-wide f32 nodes, the triangles in leaf order, visit
//                    ranks (ObjectGroup::intersect_iter's order, group.rs:171-201), the reference-order child records.
//
// Naming of the groups needs no allocator: a group that splits is named by its split position m (the first position of its
// right part), which no other group of the tree can have; a group that does not split by its first position.
#pragma once
#include "rt_device.h"

namespace mesh {

#define RT_MESH_BLOCK 256u
#define RT_MESH_MAX_BLOCKS 4096u                       // scatter_kernel keeps one prefix per block in LDS
#define RT_MESH_MAX_TRIS (RT_MESH_BLOCK * RT_MESH_MAX_BLOCKS)
#define RT_MESH_MAX_LEVELS 64u                         // (the reference-order traversal stack holds 64 groups, rt_reforder.h)
#define RT_MESH_CLASSES 6u                             // classes of a centre on one axis: below candidate planes k..4, k = 0..5

enum { SEG_OPEN = 0, SEG_LEAF = 1, SEG_SPLIT = 2 };
#define RT_MESH_CHILD_LEAF 0xFFFFFFFFu
enum { MESH_STATUS_UNORDERED = 1u, MESH_STATUS_NOT_FINITE = 2u, MESH_STATUS_TOO_DEEP = 4u };

// Order-preserving map of f64 (finite and +-inf) onto u64, so that box unions are integer atomic min / max.
__device__ __forceinline__ unsigned long long dkey(double d) {
  unsigned long long u = (unsigned long long)__double_as_longlong(d);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dunkey(unsigned long long k) {
  unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)u);
}
#define RT_MESH_KEY_POS_INF 0xFFF0000000000000ull  // dkey(+inf): an empty box's minimum
#define RT_MESH_KEY_NEG_INF 0x000FFFFFFFFFFFFFull  // dkey(-inf): an empty box's maximum

struct Acc {  // one class of one axis of one group: union of the member boxes + their primitive count (group.rs:110-124)
  unsigned long long mn[3], mx[3];
  uint32_t count, pad;
};
#define RT_MESH_ACC_PER_SEG (3u * RT_MESH_CLASSES)

struct Seg {  // a group being built = positions [start, end) of the order array; the record is indexed by `start`
  uint32_t end;
  uint32_t parent;  // split position of the group that made it | side << 31; RT_NONE: the mesh's root group
  uint32_t state;   // SEG_*
  uint32_t level;
  double bb[6];     // ObjectGroup::bounding_box: the union of the children's boxes (group.rs:76-80)
};
struct Dec {  // this level's decision for the group that starts here (read by count / scatter)
  uint32_t split;  // 0: none
  uint32_t axis, level, pad;
  double pos;
};
struct Bin {  // a group that split, indexed by its split position
  double bb[2][6];    // boxes of the two new groups (group.rs:131-143)
  uint32_t lo, hi;    // the group was [lo, hi): children [lo, m) and [m, hi)
  uint32_t child[2];  // 0: not decided yet; RT_MESH_CHILD_LEAF: keeps its children; else: its own split position
  uint32_t level, parent;
  uint32_t valid, pad;
};
struct Ctrl {
  uint32_t active[RT_MESH_MAX_LEVELS + 2];  // groups of each level that are still open
  uint32_t status, max_level, root_split /* 0: the root group keeps its children */, max_leaf;
};

struct MeshDev {
  uint32_t n;
  // the base pose (rt_mesh_create)
  const uint8_t* shape;
  const double *p1, *p2, *p3, *n1, *n2, *n3, *bbox;
  const uint32_t* mat;
  // the frame's pose (pose_kernel)
  double *q1, *e1, *e2, *m1, *m2, *m3;  // [n][3] p1, e1, e2, n1 (flat: the normal), n2, n3
  double* pbb;                          // [n][6]
  double* ctr;                          // [n][3] BoundingBox::center (bounding_box.rs:307-313)
  unsigned long long* pose_box;         // [6] keys of the group's box
  // the build
  uint32_t* order[2];   // triangle at each position (ping-pong per level)
  uint32_t* seg_of[2];  // start of the group each position belongs to
  Seg* segs;
  Dec* dec;
  Acc* acc;
  Bin* bins;
  uint32_t* blocksum;
  uint32_t* segtail;
  Ctrl* ctrl;
};

// Where the build's results go: the scene's tables (rt_device.h) and this mesh's ranges in them.
struct MeshOut {
  DevNode* nodes;
  double* tri_isect;
  double* tri_norm;
  uint32_t* prim_rank;
  uint4* prim_info;
  uint32_t* prim_csg_left;
  uint8_t* prim_flags;
  DevEntry* entries;
  const uint32_t* mat_map;   // palette entry -> material index of the scene
  const uint8_t* mat_flags;  // palette entry: bit 0 opaque (transparency == 0), bit 1 the container walk follows a hit
  uint32_t node0;            // wide nodes [node0, node0 + 2n): [0, n) by split position (0 = the root's), [n, 2n) chains of big leaves
  uint32_t tri0, prim0, rank0;
  uint32_t entry_groups0;    // child records of the groups that split: 2 per split position
  uint32_t entry_prims0;     // child records of the triangles, by position
  uint32_t root_entry;       // the record of the mesh's root group in the scene's own part of the table
  uint32_t built;
  double group_bb[6];        // the group's ObjectGroup::bounding_box as the host's tree holds it (rt_scene_desc::mesh_bbox)
};

struct PoseArgs {
  uint32_t n_mats, pad;
  double m[RT_MESH_MAX_MATRICES][16];
};

// BoundingBox::add_point with Tuple::min / max (bounding_box.rs:107-110, tuple.rs:26-40)
__device__ __forceinline__ void box_add_point(double* bb, d3 p) {
  bb[0] = fmin(p.x, bb[0]);
  bb[1] = fmin(p.y, bb[1]);
  bb[2] = fmin(p.z, bb[2]);
  bb[3] = fmax(p.x, bb[3]);
  bb[4] = fmax(p.y, bb[4]);
  bb[5] = fmax(p.z, bb[5]);
}

__device__ __forceinline__ d3 ld_d3(const double* p, uint32_t i) { return mk3(p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]); }
__device__ __forceinline__ void st_d3(double* p, uint32_t i, d3 v) {
  p[3 * (size_t)i] = v.x;
  p[3 * (size_t)i + 1] = v.y;
  p[3 * (size_t)i + 2] = v.z;
}

__global__ __launch_bounds__(RT_MESH_BLOCK) void pose_init_kernel(MeshDev M) {
  if (threadIdx.x < 3) M.pose_box[threadIdx.x] = RT_MESH_KEY_POS_INF;
  else if (threadIdx.x < 6) M.pose_box[threadIdx.x] = RT_MESH_KEY_NEG_INF;
}

// Object::transform(m_0) ... transform(m_{k-1}) for every triangle of the group, from the base pose.
__global__ __launch_bounds__(RT_MESH_BLOCK) void pose_kernel(MeshDev M, PoseArgs A) {
  __shared__ unsigned long long sh_box[6];
  if (threadIdx.x < 3) sh_box[threadIdx.x] = RT_MESH_KEY_POS_INF;
  else if (threadIdx.x < 6) sh_box[threadIdx.x] = RT_MESH_KEY_NEG_INF;
  __syncthreads();
  const uint32_t i = blockIdx.x * RT_MESH_BLOCK + threadIdx.x;
  if (i < M.n) {
    const bool smooth = M.shape[i] == RT_SHAPE_SMOOTH_TRIANGLE;
    d3 a = ld_d3(M.p1, i), b = ld_d3(M.p2, i), c = ld_d3(M.p3, i);
    d3 na = mk3(0., 0., 0.), nb = na, nc = na;
    if (smooth) {
      na = ld_d3(M.n1, i);
      nb = ld_d3(M.n2, i);
      nc = ld_d3(M.n3, i);
    }
    double bb[6];
    for (int k = 0; k < 6; k++) bb[k] = M.bbox[6 * (size_t)i + k];
    for (uint32_t k = 0; k < A.n_mats; k++) {
      const double* m = A.m[k];
      // self.bbox.transform(matrix): the 8 corners in the order of bounding_box.rs:49-58
      const double lo[3] = {bb[0], bb[1], bb[2]}, hi[3] = {bb[3], bb[4], bb[5]};
      double nbx[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
      for (int corner = 0; corner < 8; corner++) {
        d3 p = mk3((corner & 4) ? hi[0] : lo[0], (corner & 2) ? hi[1] : lo[1], (corner & 1) ? hi[2] : lo[2]);
        box_add_point(nbx, mat_point(m, p));
      }
      for (int j = 0; j < 6; j++) bb[j] = nbx[j];
      a = mat_point(m, a);
      b = mat_point(m, b);
      c = mat_point(m, c);
      if (smooth) {  // the normals by the FORWARD matrix (smooth_triangle.rs:33-35)
        na = mat_vector(m, na);
        nb = mat_vector(m, nb);
        nc = mat_vector(m, nc);
      }
    }
    const d3 v12 = sub3(b, a), v13 = sub3(c, a);  // Triangle::new / SmoothTriangle::new
    if (!smooth) na = normalize3(cross3(v13, v12));
    st_d3(M.q1, i, a);
    st_d3(M.e1, i, v12);
    st_d3(M.e2, i, v13);
    st_d3(M.m1, i, na);
    st_d3(M.m2, i, nb);
    st_d3(M.m3, i, nc);
    bool finite = true;
    for (int k = 0; k < 6; k++) {
      M.pbb[6 * (size_t)i + k] = bb[k];
      finite = finite && fabs(bb[k]) <= 1.0e18;  // (the device's limit on box coordinates)
    }
    st_d3(M.ctr, i, mk3((bb[0] + bb[3]) / 2., (bb[1] + bb[4]) / 2., (bb[2] + bb[5]) / 2.));
    if (!finite) atomicOr(&M.ctrl->status, (uint32_t)MESH_STATUS_NOT_FINITE);
    for (int k = 0; k < 3; k++) {  // group.rs:230-235: add_bounding_box adds both corners to both extremes
      atomicMin(&sh_box[k], min(dkey(bb[k]), dkey(bb[k + 3])));
      atomicMax(&sh_box[k + 3], max(dkey(bb[k]), dkey(bb[k + 3])));
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&M.pose_box[threadIdx.x], sh_box[threadIdx.x]);
  else if (threadIdx.x < 6) atomicMax(&M.pose_box[threadIdx.x], sh_box[threadIdx.x]);
}

// A frame's build starts: every triangle in the base order, one group holding them all.
__global__ __launch_bounds__(RT_MESH_BLOCK) void begin_kernel(MeshDev M, MeshOut O, uint32_t built) {
  const uint32_t p = blockIdx.x * RT_MESH_BLOCK + threadIdx.x;
  if (p < M.n) {
    M.order[0][p] = p;
    M.seg_of[0][p] = 0u;
  }
  if (p == 0) {
    Seg r;
    r.end = M.n;
    r.parent = RT_NONE;
    r.level = 0;
    for (int k = 0; k < 6; k++) r.bb[k] = O.group_bb[k];
    r.state = (M.n <= 2u || !built) ? SEG_LEAF : SEG_OPEN;  // group.rs:145-152
    M.segs[0] = r;
    Ctrl* c = M.ctrl;
    for (uint32_t l = 0; l < RT_MESH_MAX_LEVELS + 2; l++) c->active[l] = 0u;
    c->active[0] = r.state == SEG_OPEN ? 1u : 0u;
    c->max_level = 0u;
    c->root_split = 0u;
    c->max_leaf = r.state == SEG_LEAF ? M.n : 0u;
  }
}

// Empties every accumulator (once per mesh, and after a build that was abandoned half-way: choose_kernel leaves them empty).
__global__ __launch_bounds__(RT_MESH_BLOCK) void acc_reset_kernel(MeshDev M) {
  const size_t i = (size_t)blockIdx.x * RT_MESH_BLOCK + threadIdx.x;
  if (i >= (size_t)M.n * RT_MESH_ACC_PER_SEG) return;
  Acc a;
  for (int k = 0; k < 3; k++) {
    a.mn[k] = RT_MESH_KEY_POS_INF;
    a.mx[k] = RT_MESH_KEY_NEG_INF;
  }
  a.count = 0;
  a.pad = 0;
  M.acc[i] = a;
}

// Candidate plane i of axis a of a group with box bb: `self.bounding_box.min + bbox_len_vec * coef` (group.rs:91-97)
__device__ __forceinline__ double candidate_pos(const double* bb, int a, int i) {
  // (selects, not indexed loads: a box held in registers stays there when the axis is a run-time value)
  const double lo = a == 0 ? bb[0] : (a == 1 ? bb[1] : bb[2]), hi = a == 0 ? bb[3] : (a == 1 ? bb[4] : bb[5]);
  const double coef = ((double)i + 1.) / (5. + 1.);
  const double len = hi - lo;
  return lo + coef * len;  // Vector * f64 is `rhs * self.x` (vector.rs:136-146)
}

// Class of a centre coordinate among the 5 candidate planes of its axis: k = the first plane it lies below (5: none).  The
// planes are non-decreasing for a finite box, so "below plane i" (group.rs:117) holds exactly for i >= k.  false: they are not
// (a NaN somewhere): the build reports MESH_STATUS_UNORDERED and the caller prepares the frame on the host.
__device__ __forceinline__ bool centre_class(const double* bb, int a, double c, uint32_t& k) {
  uint32_t mask = 0;
  for (int i = 0; i < 5; i++) mask |= (c < candidate_pos(bb, a, i)) ? (1u << i) : 0u;
  k = 5u - (uint32_t)__builtin_popcount(mask);
  return mask == ((0x1Fu << k) & 0x1Fu);
}

// Groups of at most RT_MESH_SERIAL children are summed by the one thread that also chooses their split (choose_kernel): no
// atomics at all, and the deep levels of a tree hold thousands of such groups.  Larger groups: every child adds its box and
// count to its class on each axis — first in LDS, in the accumulators of the (at most RT_MESH_SLOTS) groups a block of 256
// consecutive positions can meet, then one atomic per class and block: device-wide atomics are served at the memory side, a
// few per nanosecond whatever the address, and one per child and class (0.5 M per level of a 23 000-triangle model) was half
// of the build's time.
#ifndef RT_MESH_SERIAL
#define RT_MESH_SERIAL 16u
#endif
#define RT_MESH_SLOTS (RT_MESH_BLOCK / (RT_MESH_SERIAL + 1u) + 2u)
#define RT_MESH_CHOOSE_POS 64u                                            // positions a block of choose_kernel looks after
#define RT_MESH_CHOOSE_WINDOW (RT_MESH_CHOOSE_POS + RT_MESH_SERIAL)   // ... and holds in LDS
__global__ __launch_bounds__(RT_MESH_BLOCK) void bin_kernel(MeshDev M, uint32_t cur, uint32_t level) {
  if (M.ctrl->active[level] == 0u) return;
  __shared__ unsigned long long sh_mn[RT_MESH_SLOTS][RT_MESH_ACC_PER_SEG][3], sh_mx[RT_MESH_SLOTS][RT_MESH_ACC_PER_SEG][3];
  __shared__ uint32_t sh_cnt[RT_MESH_SLOTS][RT_MESH_ACC_PER_SEG];
  __shared__ uint32_t sh_key[RT_MESH_SLOTS];
  const uint32_t p = blockIdx.x * RT_MESH_BLOCK + threadIdx.x;
  for (uint32_t i = threadIdx.x; i < RT_MESH_SLOTS * RT_MESH_ACC_PER_SEG; i += RT_MESH_BLOCK) {
    const uint32_t sl = i / RT_MESH_ACC_PER_SEG, c = i % RT_MESH_ACC_PER_SEG;
    for (int k = 0; k < 3; k++) {
      sh_mn[sl][c][k] = RT_MESH_KEY_POS_INF;
      sh_mx[sl][c][k] = RT_MESH_KEY_NEG_INF;
    }
    sh_cnt[sl][c] = 0u;
  }
  if (threadIdx.x < RT_MESH_SLOTS) sh_key[threadIdx.x] = RT_NONE;
  __syncthreads();
  uint32_t s = RT_NONE;
  bool open = false;
  if (p < M.n) {
    s = M.seg_of[cur][p];
    const Seg& r = M.segs[s];
    open = r.state == SEG_OPEN && r.level == level && r.end - s > RT_MESH_SERIAL;
  }
  if (open) {
    uint32_t slot = RT_MESH_SLOTS;  // (a block meets at most RT_MESH_SLOTS groups of more than RT_MESH_SERIAL children)
    for (uint32_t i = 0; i < RT_MESH_SLOTS; i++) {
      const uint32_t old = atomicCAS(&sh_key[i], RT_NONE, s);
      if (old == RT_NONE || old == s) {
        slot = i;
        break;
      }
    }
    const uint32_t elem = M.order[cur][p];
    double gb[6], bb[6];
    for (int k = 0; k < 6; k++) {
      gb[k] = M.segs[s].bb[k];
      bb[k] = M.pbb[6 * (size_t)elem + k];
    }
    unsigned long long kmn[3], kmx[3];
    for (int k = 0; k < 3; k++) {
      kmn[k] = min(dkey(bb[k]), dkey(bb[k + 3]));
      kmx[k] = max(dkey(bb[k]), dkey(bb[k + 3]));
    }
    for (int a = 0; a < 3; a++) {
      uint32_t k;
      if (!centre_class(gb, a, M.ctr[3 * (size_t)elem + a], k)) atomicOr(&M.ctrl->status, (uint32_t)MESH_STATUS_UNORDERED);
      const uint32_t c = (uint32_t)a * RT_MESH_CLASSES + k;
      if (slot < RT_MESH_SLOTS) {
        for (int j = 0; j < 3; j++) {
          atomicMin(&sh_mn[slot][c][j], kmn[j]);
          atomicMax(&sh_mx[slot][c][j], kmx[j]);
        }
        atomicAdd(&sh_cnt[slot][c], 1u);
      } else {
        Acc* acc = M.acc + (size_t)s * RT_MESH_ACC_PER_SEG + c;
        for (int j = 0; j < 3; j++) {
          atomicMin(&acc->mn[j], kmn[j]);
          atomicMax(&acc->mx[j], kmx[j]);
        }
        atomicAdd(&acc->count, 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < RT_MESH_SLOTS * RT_MESH_ACC_PER_SEG; i += RT_MESH_BLOCK) {
    const uint32_t sl = i / RT_MESH_ACC_PER_SEG, c = i % RT_MESH_ACC_PER_SEG;
    if (sh_key[sl] == RT_NONE || sh_cnt[sl][c] == 0u) continue;
    Acc* acc = M.acc + (size_t)sh_key[sl] * RT_MESH_ACC_PER_SEG + c;
    for (int j = 0; j < 3; j++) {
      atomicMin(&acc->mn[j], sh_mn[sl][c][j]);
      atomicMax(&acc->mx[j], sh_mx[sl][c][j]);
    }
    atomicAdd(&acc->count, sh_cnt[sl][c]);
  }
}

struct Side {  // one side of a candidate plane: calculate_split_cost's left_bbox / left_count (group.rs:110-124)
  double mn[3], mx[3];
  uint32_t count;
};
__device__ __forceinline__ void side_reset(Side& s) {
  for (int k = 0; k < 3; k++) {
    s.mn[k] = INFINITY;
    s.mx[k] = -INFINITY;
  }
  s.count = 0;
}
__device__ __forceinline__ void side_add(Side& s, const Side& o) {
  s.count += o.count;
  if (o.count == 0u) return;
  for (int k = 0; k < 3; k++) {
    s.mn[k] = fmin(o.mn[k], s.mn[k]);
    s.mx[k] = fmax(o.mx[k], s.mx[k]);
  }
}
__device__ __forceinline__ double half_area(const double* mn, const double* mx) {  // bounding_box.rs:328-336
  if (mn[0] == INFINITY && mn[1] == INFINITY && mn[2] == INFINITY && mx[0] == -INFINITY && mx[1] == -INFINITY && mx[2] == -INFINITY) return 0.;
  const double x_len = mx[0] - mn[0], y_len = mx[1] - mn[1], z_len = mx[2] - mn[2];
  return x_len * y_len + y_len * z_len + z_len * x_len;
}

// Both sides of candidate plane i of axis a of the group [p, r.end) (calculate_split_cost, group.rs:109-124).  A small
// group (bin_kernel skipped it): straight from its children, as the reference does — from the block's LDS copy, or from
// memory for the children beyond the block.  Otherwise from the classes bin_kernel accumulated: left = classes 0..i.
__device__ __forceinline__ void candidate_sides(const MeshDev& M, uint32_t cur, const Seg& r, uint32_t p, int a, int i, bool small, uint32_t block0,
                                                const double (*sh_bb)[6], const double (*sh_ctr)[3], Side& left, Side& right) {
  // (sh_bb / sh_ctr hold positions [block0, block0 + RT_MESH_CHOOSE_WINDOW))
  side_reset(left);
  side_reset(right);
  if (!small) {
    // (all six classes are fetched before any is looked at; an empty class is the identity of min / max, so none is skipped)
    unsigned long long kmn[RT_MESH_CLASSES][3], kmx[RT_MESH_CLASSES][3];
    uint32_t cnt[RT_MESH_CLASSES];
    const Acc* acc = M.acc + (size_t)p * RT_MESH_ACC_PER_SEG + (uint32_t)a * RT_MESH_CLASSES;
#pragma unroll
    for (uint32_t k = 0; k < RT_MESH_CLASSES; k++) {
      for (int j = 0; j < 3; j++) {
        kmn[k][j] = acc[k].mn[j];
        kmx[k][j] = acc[k].mx[j];
      }
      cnt[k] = acc[k].count;
    }
#pragma unroll
    for (uint32_t k = 0; k < RT_MESH_CLASSES; k++) {
      const bool to_left = k <= (uint32_t)i;
      for (int j = 0; j < 3; j++) {
        const double mn = dunkey(kmn[k][j]), mx = dunkey(kmx[k][j]);
        left.mn[j] = to_left ? fmin(mn, left.mn[j]) : left.mn[j];
        left.mx[j] = to_left ? fmax(mx, left.mx[j]) : left.mx[j];
        right.mn[j] = to_left ? right.mn[j] : fmin(mn, right.mn[j]);
        right.mx[j] = to_left ? right.mx[j] : fmax(mx, right.mx[j]);
      }
      left.count += to_left ? cnt[k] : 0u;
      right.count += to_left ? 0u : cnt[k];
    }
    return;
  }
  const double pos = candidate_pos(r.bb, a, i);
  for (uint32_t q = p; q < r.end; q++) {
    double c, bb[6];
    if (q - block0 < RT_MESH_CHOOSE_WINDOW) {
      c = sh_ctr[q - block0][a];
      for (int j = 0; j < 6; j++) bb[j] = sh_bb[q - block0][j];
    } else {  // (never for a small group that starts in this block: the window reaches RT_MESH_SERIAL positions past it)
      const uint32_t elem = M.order[cur][q];
      c = M.ctr[3 * (size_t)elem + a];
      for (int j = 0; j < 6; j++) bb[j] = M.pbb[6 * (size_t)elem + j];
    }
    const bool below = c < pos;  // group.rs:117
    for (int j = 0; j < 3; j++) {  // add_bounding_box: both corners to both extremes (bounding_box.rs:111-117)
      const double mn = below ? left.mn[j] : right.mn[j], mx = below ? left.mx[j] : right.mx[j];
      const double nmn = fmin(bb[j + 3], fmin(bb[j], mn)), nmx = fmax(bb[j + 3], fmax(bb[j], mx));
      left.mn[j] = below ? nmn : left.mn[j];
      left.mx[j] = below ? nmx : left.mx[j];
      right.mn[j] = below ? right.mn[j] : nmn;
      right.mx[j] = below ? right.mx[j] : nmx;
    }
    left.count += below ? 1u : 0u;
    right.count += below ? 0u : 1u;
  }
}

// choose_split (group.rs:87-108) over the 15 candidates, the sah_cost test (group.rs:153-156), and the two new groups of
// split (group.rs:131-143, 157-166), for every open group of this level.  The 15 candidates of a group are dealt to up to 15
// threads — those of the group's first positions, which have nothing else to do —, each of which keeps the best of its own in
// the reference's order (positions outer, axes inner, strict <); the group's first thread then takes the best of those
// (smallest cost, ties to the earlier candidate: what the reference's single loop keeps) and makes the split.
#define RT_MESH_HELPERS 15u
__device__ __forceinline__ void choose_tail(const MeshDev& M, uint32_t cur, uint32_t level, const Seg& r, uint32_t p, bool small, uint32_t helpers,
                                            uint32_t lanes, uint32_t block0, const double (*sh_bb)[6], const double (*sh_ctr)[3], const double* sh_cost,
                                            const uint32_t* sh_cand, uint32_t* sh_opened, uint32_t* sh_max_leaf);
// A block of 256 threads looks after RT_MESH_CHOOSE_POS = 64 positions, four threads per position (wavefront g holds thread g
// of every position): a deep level has thousands of small groups and little else, and with one thread per position the
// whole level ran on 368 wavefronts of long serial work, a third of the GPU's SIMDs with one wavefront each.
__global__ __launch_bounds__(RT_MESH_BLOCK) void choose_kernel(MeshDev M, uint32_t cur, uint32_t level) {
  if (M.ctrl->active[level] == 0u) return;
  // the block's children (and the RT_MESH_SERIAL positions behind them) in LDS, box and centre: what the small groups are summed
  // from, instead of a chain of dependent global loads per child
  __shared__ double sh_bb[RT_MESH_CHOOSE_WINDOW][6];
  __shared__ double sh_ctr[RT_MESH_CHOOSE_WINDOW][3];
  __shared__ double sh_cost[RT_MESH_BLOCK];
  __shared__ uint32_t sh_cand[RT_MESH_BLOCK];
  const uint32_t block0 = blockIdx.x * RT_MESH_CHOOSE_POS, g = threadIdx.x / RT_MESH_CHOOSE_POS, t = block0 + threadIdx.x % RT_MESH_CHOOSE_POS;
  for (uint32_t i = threadIdx.x; i < RT_MESH_CHOOSE_WINDOW * 9u; i += RT_MESH_BLOCK) {  // one f64 per thread and turn
    const uint32_t pos = block0 + i / 9u, k = i % 9u;
    if (pos >= M.n) continue;
    const uint32_t elem = M.order[cur][pos];
    if (k < 6u) sh_bb[i / 9u][k] = M.pbb[6 * (size_t)elem + k];
    else sh_ctr[i / 9u][k - 6u] = M.ctr[3 * (size_t)elem + (k - 6u)];
  }
  uint32_t p = RT_NONE;  // the group this thread's position belongs to
  if (t < M.n) p = M.seg_of[cur][t];
  __syncthreads();
  Seg r;
  bool open = false;
  if (p != RT_NONE && p >= block0) {  // (a group that starts in an earlier block is that block's)
    r = M.segs[p];
    open = r.state == SEG_OPEN && r.level == level;
  }
  const bool small = open && r.end - p <= RT_MESH_SERIAL;
  // the group's positions in this block, four threads each: its helpers, RT_MESH_HELPERS at most
  const uint32_t lanes = open ? min(r.end, block0 + RT_MESH_CHOOSE_POS) - p : 0u;
  const uint32_t helpers = min(lanes * (RT_MESH_BLOCK / RT_MESH_CHOOSE_POS), RT_MESH_HELPERS);
  const uint32_t h = open ? g * lanes + (t - p) : RT_NONE;
  double my_cost = INFINITY;
  uint32_t my_cand = 0u;
  if (open && h < helpers)
    for (uint32_t cand = h; cand < 15u; cand += helpers) {  // cand = 3 i + a: the reference's order
      Side l, rt;
      candidate_sides(M, cur, r, p, (int)(cand % 3u), (int)(cand / 3u), small, block0, sh_bb, sh_ctr, l, rt);
      const double cost = (l.count == 0u || rt.count == 0u) ? INFINITY
                                                            : half_area(l.mn, l.mx) * (double)l.count + half_area(rt.mn, rt.mx) * (double)rt.count;
      if (cost < my_cost) {
        my_cost = cost;
        my_cand = cand;
      }
    }
  sh_cost[threadIdx.x] = my_cost;
  sh_cand[threadIdx.x] = my_cand;
  // what the block's groups add to the build's counters goes through LDS: atomics on ONE address are served one after the
  // other (~10 ns each), and a deep level has thousands of groups
  __shared__ uint32_t sh_opened, sh_max_leaf;
  if (threadIdx.x == 0) {
    sh_opened = 0u;
    sh_max_leaf = 0u;
  }
  __syncthreads();
  if (open && h == 0u) choose_tail(M, cur, level, r, p, small, helpers, lanes, block0, sh_bb, sh_ctr, sh_cost, sh_cand, &sh_opened, &sh_max_leaf);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (sh_opened) {
      if (level + 1u >= RT_MESH_MAX_LEVELS) atomicOr(&M.ctrl->status, (uint32_t)MESH_STATUS_TOO_DEEP);
      else atomicAdd(&M.ctrl->active[level + 1u], sh_opened);
    }
    if (sh_max_leaf) atomicMax(&M.ctrl->max_leaf, sh_max_leaf);
  }
}

// The group's first thread: the best of the helpers' candidates, the sah_cost test, the split.
__device__ __forceinline__ void choose_tail(const MeshDev& M, uint32_t cur, uint32_t level, const Seg& r, uint32_t p, bool small, uint32_t helpers,
                                            uint32_t lanes, uint32_t block0, const double (*sh_bb)[6], const double (*sh_ctr)[3], const double* sh_cost,
                                            const uint32_t* sh_cand, uint32_t* sh_opened, uint32_t* sh_max_leaf) {
  double best_cost = INFINITY;
  uint32_t best_cand = 0u;
  for (uint32_t k = 0; k < helpers; k++) {  // helper k = thread k / lanes of position p + k % lanes
    const uint32_t tid = (k / lanes) * RT_MESH_CHOOSE_POS + (p - block0) + k % lanes;
    const double c = sh_cost[tid];
    const uint32_t cand = sh_cand[tid];
    if (c < best_cost || (c == best_cost && c < INFINITY && cand < best_cand)) {
      best_cost = c;
      best_cand = cand;
    }
  }
  const int best_axis = (int)(best_cand % 3u), best_i = (int)(best_cand / 3u);
  const double best_pos = candidate_pos(r.bb, best_axis, best_i);
  const bool serial = small;
  auto reset_acc = [&]() {  // the accumulators are left empty for the group that starts here next
    if (serial) return;
    for (uint32_t k = 0; k < RT_MESH_ACC_PER_SEG; k++) {
      Acc* acc = M.acc + (size_t)p * RT_MESH_ACC_PER_SEG + k;
      for (int j = 0; j < 3; j++) {
        acc->mn[j] = RT_MESH_KEY_POS_INF;
        acc->mx[j] = RT_MESH_KEY_NEG_INF;
      }
      acc->count = 0u;
    }
  };
  const uint32_t size = r.end - p;
  const double sah = half_area(r.bb, r.bb + 3) * (double)size;  // ObjectGroup::sah_cost (group.rs:220-222)
  Ctrl* c = M.ctrl;
  uint32_t* parent_slot = r.parent == RT_NONE ? nullptr : &M.bins[r.parent & 0x7FFFFFFFu].child[r.parent >> 31];
  if (best_cost >= sah) {  // group.rs:154-156: the children stay as they are
    reset_acc();
    M.segs[p].state = SEG_LEAF;
    if (parent_slot) *parent_slot = RT_MESH_CHILD_LEAF;
    atomicMax(sh_max_leaf, size);
    return;
  }
  Side l, rt;
  candidate_sides(M, cur, r, p, best_axis, best_i, small, block0, sh_bb, sh_ctr, l, rt);
  reset_acc();
  const uint32_t m = p + l.count;
  if (parent_slot) *parent_slot = m;
  else c->root_split = m;
  Dec d;
  d.split = m;
  d.axis = (uint32_t)best_axis;
  d.level = level;
  d.pad = 0;
  d.pos = best_pos;
  M.dec[p] = d;
  Bin b;
  for (int k = 0; k < 3; k++) {
    b.bb[0][k] = l.mn[k];
    b.bb[0][k + 3] = l.mx[k];
    b.bb[1][k] = rt.mn[k];
    b.bb[1][k + 3] = rt.mx[k];
  }
  b.lo = p;
  b.hi = r.end;
  b.level = level;
  b.parent = r.parent;
  b.valid = 1u;
  b.pad = 0u;
  uint32_t opened = 0;
  for (uint32_t side = 0; side < 2; side++) {
    Seg ch;
    const uint32_t start = side ? m : p;
    ch.end = side ? r.end : m;
    ch.parent = m | (side << 31);
    ch.level = level + 1u;
    for (int k = 0; k < 6; k++) ch.bb[k] = b.bb[side][k];
    const bool leaf = ch.end - start <= 2u;  // group.rs:145-152 (triangles have no groups below them)
    ch.state = leaf ? SEG_LEAF : SEG_OPEN;
    b.child[side] = leaf ? RT_MESH_CHILD_LEAF : 0u;
    opened += leaf ? 0u : 1u;
    M.segs[start] = ch;
  }
  M.bins[m] = b;
  if (opened) atomicAdd(sh_opened, opened);
}

// "centre below the chosen plane" (group.rs:136) for the child at position p of a group that splits this level.
__device__ __forceinline__ bool moves_left(const MeshDev& M, uint32_t cur, uint32_t level, uint32_t p, uint32_t& s, Dec& d, bool& moving) {
  moving = false;
  if (p >= M.n) return false;
  s = M.seg_of[cur][p];
  d = M.dec[s];
  moving = d.level == level && d.split != 0u;
  if (!moving) return false;
  const uint32_t elem = M.order[cur][p];
  return M.ctr[3 * (size_t)elem + d.axis] < d.pos;
}

// Exclusive scan of a flag over the block (256 threads = 4 wavefronts); total = the block's sum.
__device__ __forceinline__ uint32_t block_scan(bool flag, uint32_t* sh_wave /* [4] */, uint32_t& total) {
  const unsigned long long m = __builtin_amdgcn_ballot_w64(flag);
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  if (lane == 0) sh_wave[wv] = (uint32_t)__builtin_popcountll(m);
  __syncthreads();
  uint32_t base = 0;
  total = 0;
  for (uint32_t q = 0; q < 4; q++) {
    if (q == wv) base = total;
    total += sh_wave[q];
  }
  __syncthreads();
  return base + before;
}

__global__ __launch_bounds__(RT_MESH_BLOCK) void count_kernel(MeshDev M, uint32_t cur, uint32_t level) {
  if (M.ctrl->active[level] == 0u) return;
  __shared__ uint32_t sh_wave[4];
  const uint32_t p = blockIdx.x * RT_MESH_BLOCK + threadIdx.x;
  uint32_t s = RT_NONE;
  Dec d;
  bool moving;
  const bool flag = moves_left(M, cur, level, p, s, d, moving);
  uint32_t total;
  const uint32_t before = block_scan(flag, sh_wave, total);
  if (threadIdx.x == 0) M.blocksum[blockIdx.x] = total;
  if (moving && s == p) M.segtail[p] = total - before;  // children of this group moving left in [p, end of this block)
}

// ObjectGroup::split (group.rs:131-143) for every group that splits this level: a stable partition of its segment.
__global__ __launch_bounds__(RT_MESH_BLOCK) void scatter_kernel(MeshDev M, uint32_t cur, uint32_t level) {
  const uint32_t p = blockIdx.x * RT_MESH_BLOCK + threadIdx.x;
  const uint32_t nxt = cur ^ 1u;
  if (M.ctrl->active[level] == 0u) {  // nothing moves any more: the order is carried over (the host flips buffers per level)
    if (p < M.n) {
      M.order[nxt][p] = M.order[cur][p];
      M.seg_of[nxt][p] = M.seg_of[cur][p];
    }
    return;
  }
  __shared__ uint32_t sh_prefix[RT_MESH_MAX_BLOCKS + 1];  // sh_prefix[b] = children moving left in the blocks before b
  __shared__ uint32_t sh_wave[4], sh_carry;
  __shared__ uint32_t sh_before[RT_MESH_BLOCK];
  const uint32_t nb = gridDim.x;
  if (threadIdx.x == 0) sh_carry = 0u;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += RT_MESH_BLOCK) {
    const uint32_t b = base + threadIdx.x;
    const uint32_t v = b < nb ? M.blocksum[b] : 0u;
    // inclusive scan of v over the block, by wavefront
    uint32_t x = v;
    for (uint32_t off = 1; off < 64u; off <<= 1) {
      const uint32_t y = __shfl_up(x, off, 64);
      if ((threadIdx.x & 63u) >= off) x += y;
    }
    if ((threadIdx.x & 63u) == 63u) sh_wave[threadIdx.x >> 6] = x;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t q = 0; q < (threadIdx.x >> 6); q++) wbase += sh_wave[q];
    const uint32_t incl = sh_carry + wbase + x;
    if (b < nb) sh_prefix[b] = incl - v;
    __syncthreads();
    if (threadIdx.x == RT_MESH_BLOCK - 1u) sh_carry = incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) sh_prefix[nb] = sh_carry;
  __syncthreads();
  uint32_t s = RT_NONE;
  Dec d;
  bool moving;
  const bool flag = moves_left(M, cur, level, p, s, d, moving);
  uint32_t total;
  const uint32_t before = block_scan(flag, sh_wave, total);
  sh_before[threadIdx.x] = before;
  __syncthreads();
  if (p >= M.n) return;
  const uint32_t elem = M.order[cur][p];
  uint32_t q = p, ns = s;
  if (moving) {
    const uint32_t bs = s / RT_MESH_BLOCK;
    uint32_t left_before;  // children of the group moving left in [s, p)
    if (bs == blockIdx.x) left_before = before - sh_before[s - bs * RT_MESH_BLOCK];
    else left_before = M.segtail[s] + (sh_prefix[blockIdx.x] - sh_prefix[bs + 1u]) + before;
    if (flag) {
      q = s + left_before;
      ns = s;
    } else {
      q = d.split + (p - s - left_before);
      ns = d.split;
    }
  }
  M.order[nxt][q] = elem;
  M.seg_of[nxt][q] = ns;
}

// f32 box of a wide node's slot: the f64 box rounded OUTWARD (rtcore.hip set_node_box).
__device__ __forceinline__ void set_slot_box(DevNode& n, int slot, const double* bb) {
  for (int k = 0; k < 6; k++) {
    float f = (float)bb[k];
    if (k < 3) {
      if ((double)f > bb[k]) f = nextafterf(f, -INFINITY);
    } else {
      if ((double)f < bb[k]) f = nextafterf(f, INFINITY);
    }
    n.bb[slot][k] = f;
  }
}

// How a wide node refers to the leaf group [a, b): a triangle run, or (more than RT_FREF_LEAF_MAX) the first node of its chain.
__device__ __forceinline__ uint32_t leaf_fref(const MeshDev& M, const MeshOut& O, uint32_t a, uint32_t b) {
  const uint32_t c = b - a;
  if (c <= RT_FREF_LEAF_MAX) return RT_MAKE_FREF(RT_FREF_TRIS, c, O.tri0 + a);
  return RT_MAKE_FREF(RT_FREF_INNER, 0, O.node0 + M.n + a);
}

// One thread per group that split: its children's records for the reference-order walk (rt_reforder.h), and — groups at even
// levels — the 4-wide node of the fast path: the group's two children, each replaced by ITS two children when it split too
// (the boxes of the odd levels are not tested: rt_device.h).
__global__ __launch_bounds__(RT_MESH_BLOCK) void fold_kernel(MeshDev M, MeshOut O) {
  const uint32_t m = blockIdx.x * RT_MESH_BLOCK + threadIdx.x;
  if (m == 0u || m >= M.n) return;
  const Bin b = M.bins[m];
  if (!b.valid) return;
  for (uint32_t side = 0; side < 2; side++) {
    DevEntry e;
    for (int k = 0; k < 6; k++) e.bb[k] = b.bb[side][k];
    e.ref = RT_MAKE_REF(RT_REF_NODE, 0);
    const uint32_t lo = side ? m : b.lo, hi = side ? b.hi : m;
    if (b.child[side] == RT_MESH_CHILD_LEAF) {
      e.begin = O.entry_prims0 + lo;
      e.count = hi - lo;
    } else {
      e.begin = O.entry_groups0 + 2u * b.child[side];
      e.count = 2u;
    }
    e.pad = 0u;
    O.entries[O.entry_groups0 + 2u * m + side] = e;
  }
  if (b.parent == RT_NONE) {  // the mesh's root group: its record is in the scene's own part of the table
    O.entries[O.root_entry].begin = O.entry_groups0 + 2u * m;
    O.entries[O.root_entry].count = 2u;
  }
  if (b.level & 1u) return;
  DevNode n;
  for (int sl = 0; sl < RT_NODE_WIDTH; sl++) {
    for (int k = 0; k < 6; k++) n.bb[sl][k] = 0.f;
    n.ref[sl] = RT_FREF_NONE;
    n.pad[sl] = 0u;
  }
  int slots = 0;
  for (uint32_t side = 0; side < 2; side++) {
    const uint32_t lo = side ? m : b.lo, hi = side ? b.hi : m;
    if (b.child[side] == RT_MESH_CHILD_LEAF) {
      set_slot_box(n, slots, b.bb[side]);
      n.ref[slots++] = leaf_fref(M, O, lo, hi);
    } else {
      const uint32_t mc = b.child[side];
      const Bin c = M.bins[mc];
      for (uint32_t s2 = 0; s2 < 2; s2++) {
        const uint32_t lo2 = s2 ? mc : c.lo, hi2 = s2 ? c.hi : mc;
        set_slot_box(n, slots, c.bb[s2]);
        n.ref[slots++] = c.child[s2] == RT_MESH_CHILD_LEAF ? leaf_fref(M, O, lo2, hi2) : RT_MAKE_FREF(RT_FREF_INNER, 0, O.node0 + c.child[s2]);
      }
    }
  }
  O.nodes[O.node0 + (b.parent == RT_NONE ? 0u : m)] = n;
}

// One thread per position: the triangle that ended up there, its primitive record, its visit rank, its child record; the
// first position of a leaf group with more triangles than a run holds also writes the group's chain of wide nodes.
__global__ __launch_bounds__(RT_MESH_BLOCK) void finalize_kernel(MeshDev M, MeshOut O, uint32_t cur) {
  const uint32_t p = blockIdx.x * RT_MESH_BLOCK + threadIdx.x;
  if (p >= M.n) return;
  const uint32_t elem = M.order[cur][p], a = M.seg_of[cur][p];
  const Seg g = M.segs[a];
  const uint32_t b = g.end, prim = O.prim0 + p, slot = O.tri0 + p;
  const uint32_t pal = M.mat[elem];
  const uint8_t mflags = O.mat_flags[pal];
  double* ti = O.tri_isect + (size_t)slot * RT_TRI_STRIDE;
  double* tn = O.tri_norm + (size_t)slot * 9;
  for (int k = 0; k < 3; k++) {
    ti[k] = M.q1[3 * (size_t)elem + k];
    ti[3 + k] = M.e1[3 * (size_t)elem + k];
    ti[6 + k] = M.e2[3 * (size_t)elem + k];
    tn[k] = M.m1[3 * (size_t)elem + k];
    tn[3 + k] = M.m2[3 * (size_t)elem + k];
    tn[6 + k] = M.m3[3 * (size_t)elem + k];
  }
  unsigned long long bits = prim;
  if (mflags & 1u) bits |= 1ull << RT_TRI_OPAQUE_BIT;
  ti[9] = __longlong_as_double((long long)bits);
  O.prim_info[prim] = make_uint4(M.shape[elem], O.mat_map[pal], RT_NONE, slot);
  // ObjectGroup::intersect_iter (group.rs:185-199): a popped group's primitives are intersected in REVERSE child order; the
  // leaf groups of the subtree pop in position order
  O.prim_rank[prim] = O.rank0 + a + (b - 1u - p);
  O.prim_csg_left[prim] = 0u;
  O.prim_flags[prim] = (mflags >> 1) & 1u;
  DevEntry e;
  for (int k = 0; k < 6; k++) e.bb[k] = 0.;
  e.ref = RT_MAKE_REF(RT_REF_PRIM, prim);
  e.begin = e.count = e.pad = 0u;
  O.entries[O.entry_prims0 + p] = e;
  if (p != a) return;
  const bool is_root = g.parent == RT_NONE && M.ctrl->root_split == 0u;  // the whole mesh is one leaf group
  if (is_root) {
    O.entries[O.root_entry].begin = O.entry_prims0;
    O.entries[O.root_entry].count = M.n;
    DevNode n;
    for (int sl = 0; sl < RT_NODE_WIDTH; sl++) {
      for (int k = 0; k < 6; k++) n.bb[sl][k] = 0.f;
      n.ref[sl] = RT_FREF_NONE;
      n.pad[sl] = 0u;
    }
    set_slot_box(n, 0, g.bb);
    n.ref[0] = leaf_fref(M, O, a, b);
    O.nodes[O.node0] = n;
  }
  if (b - a <= RT_FREF_LEAF_MAX) return;
  // the chain: each node holds three runs and the next node, the last one up to four runs; every slot behind the group's box
  for (uint32_t c = a; c < b;) {
    DevNode n;
    for (int sl = 0; sl < RT_NODE_WIDTH; sl++) {
      for (int k = 0; k < 6; k++) n.bb[sl][k] = 0.f;
      n.ref[sl] = RT_FREF_NONE;
      n.pad[sl] = 0u;
    }
    const uint32_t runs_left = (b - c + RT_FREF_LEAF_MAX - 1u) / RT_FREF_LEAF_MAX;
    const uint32_t here = runs_left <= 4u ? runs_left : 3u;
    const uint32_t node = O.node0 + M.n + c;
    for (uint32_t sl = 0; sl < here; sl++) {
      const uint32_t cnt = min(RT_FREF_LEAF_MAX, b - c);
      set_slot_box(n, (int)sl, g.bb);
      n.ref[sl] = RT_MAKE_FREF(RT_FREF_TRIS, cnt, O.tri0 + c);
      c += cnt;
    }
    if (c < b) {
      set_slot_box(n, 3, g.bb);
      n.ref[3] = RT_MAKE_FREF(RT_FREF_INNER, 0, O.node0 + M.n + c);
    }
    O.nodes[node] = n;
  }
}

}  // namespace mesh
