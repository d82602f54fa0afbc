// rt_shapes.h — ray/shape intersection in object space and the AABB slab test, f64, reference operation order.
// Shared by the fast traversal (rt_fast.h) and the reference-order traversal (rt_reforder.h).
#pragma once
#include "rt_device.h"

struct Counters {
  uint32_t rays_primary, rays_shadow, rays_reflect, rays_refract, node_visits, tri_tests, prim_tests, hits, hit_bytes, overflow;
  uint32_t wave_steps, lane_steps;  // fast path: traversal steps of the wavefront / lanes active in them
  uint32_t retraced;                // fast path: rays settled by the reference-order walk (coincident surfaces)
  uint32_t cyc_trav, cyc_adv;       // statistics builds: shader clocks / 64 spent traversing / in the state machine (single kernel);
                                    // wavefront clocks / 64 in the node loop / in leaves (streamed trace kernels)
  uint32_t shadow_culled;           // streamed pipeline: shadow rays not traced because the light is behind the surface
  uint32_t cyc_total;               // streamed trace kernels: wavefront clocks / 64 from the kernel's start to its end
};

// ---------------------------------------------------------------------------------------------
// Shapes: local intersect in object space.  Each returns the number of roots written to ts[0..4).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int sphere_roots(d3 o, d3 d, double* ts) {  // sphere.rs:20-35
  d3 sr = sub3(o, mk3(0., 0., 0.));
  double a = dot3(d, d);
  double b = 2. * dot3(d, sr);
  double c = dot3(sr, sr) - 1.;
  double disc = b * b - 4. * a * c;
  if (disc < 0. || a == 0.) return 0;
  double ds = sqrt(disc);
  ts[0] = (-b - ds) / (2. * a);
  ts[1] = (-b + ds) / (2. * a);
  return 2;
}

__device__ __forceinline__ int plane_roots(d3 o, d3 d, double* ts) {  // plane.rs:13-19
  if (approx_eq(d.y, 0.)) return 0;
  ts[0] = -o.y / d.y;
  return 1;
}

__device__ __forceinline__ void cube_axis(double origin, double dir_inv, double& tmin, double& tmax) {  // cube.rs:51-63
  double a = (-1. - origin) * dir_inv;
  double b = (1. - origin) * dir_inv;
  if (a < b) { tmin = a; tmax = b; } else { tmin = b; tmax = a; }
}
__device__ __forceinline__ int cube_roots(d3 o, d3 dinv, double* ts) {  // cube.rs:25-42
  double xmin, xmax, ymin, ymax, zmin, zmax;
  cube_axis(o.x, dinv.x, xmin, xmax);
  cube_axis(o.y, dinv.y, ymin, ymax);
  cube_axis(o.z, dinv.z, zmin, zmax);
  double tmin = fmax(fmax(xmin, ymin), zmin);
  double tmax = fmin(fmin(xmax, ymax), zmax);
  if (tmin > tmax) return 0;
  ts[0] = tmin;
  ts[1] = tmax;
  return 2;
}

__device__ __forceinline__ int cylinder_roots(d3 o, d3 d, double y_min, double y_max, bool closed, double* ts) {  // cylinder.rs:54-120
  if (approx_eq(y_min, y_max)) return 0;
  int n = 0;
  if (closed && !approx_eq(d.y, 0.)) {  // caps first (:107-120), radius test x^2+z^2 <= 1 (:101-106)
    double tmin = (y_min - o.y) / d.y;
    double tmax = (y_max - o.y) / d.y;
    double x = o.x + tmin * d.x, z = o.z + tmin * d.z;
    if (x * x + z * z <= 1.) ts[n++] = tmin;
    x = o.x + tmax * d.x;
    z = o.z + tmax * d.z;
    if (x * x + z * z <= 1.) ts[n++] = tmax;
  }
  double a = d.x * d.x + d.z * d.z;
  if (approx_eq(a, 0.)) return n;
  double b = 2. * o.x * d.x + 2. * o.z * d.z;
  double c = o.x * o.x + o.z * o.z - 1.;
  double disc = b * b - 4. * a * c;
  if (disc < 0.) return n;
  double ds = sqrt(disc);
  double t0 = (-b - ds) / (2. * a);
  double t1 = (-b + ds) / (2. * a);
  double y0 = o.y + t0 * d.y;
  if (y_min < y0 && y0 < y_max) ts[n++] = t0;
  double y1 = o.y + t1 * d.y;
  if (y_min < y1 && y1 < y_max) ts[n++] = t1;
  return n;
}

__device__ __forceinline__ int cone_roots(d3 o, d3 d, double y_min, double y_max, bool closed, double* ts) {  // cone.rs:56-127
  if (approx_eq(y_min, y_max)) return 0;
  int n = 0;
  if (closed && !approx_eq(d.y, 0.)) {  // caps: squared distance against |y| (sic, :106-112,121,124)
    double tmin = (y_min - o.y) / d.y;
    double tmax = (y_max - o.y) / d.y;
    double x = o.x + tmin * d.x, z = o.z + tmin * d.z;
    if (x * x + z * z <= fabs(y_min)) ts[n++] = tmin;
    x = o.x + tmax * d.x;
    z = o.z + tmax * d.z;
    if (x * x + z * z <= fabs(y_max)) ts[n++] = tmax;
  }
  double a = d.x * d.x - d.y * d.y + d.z * d.z;
  double b = 2. * (o.x * d.x - o.y * d.y + o.z * d.z);
  double c = o.x * o.x - o.y * o.y + o.z * o.z;
  if (approx_eq(a, 0.)) {
    if (approx_eq(b, 0.)) return n;
    ts[n++] = -c / (2. * b);  // no y-range check (:71-76)
    return n;
  }
  double disc = b * b - 4. * a * c;
  if (disc < 0.) return n;
  double ds = sqrt(disc);
  double t0 = (-b - ds) / (2. * a);
  double t1 = (-b + ds) / (2. * a);
  double y0 = o.y + t0 * d.y;
  if (y_min < y0 && y0 < y_max) ts[n++] = t0;
  double y1 = o.y + t1 * d.y;
  if (y_min < y1 && y1 < y_max) ts[n++] = t1;
  return n;
}

// Moeller-Trumbore exactly as triangle.rs:48-71 / smooth_triangle.rs:60-83
__device__ __forceinline__ int triangle_roots(const double* __restrict__ tri, d3 o, d3 d, double* ts, double& u, double& v) {
  d3 p1 = mk3(tri[0], tri[1], tri[2]);
  d3 e1 = mk3(tri[3], tri[4], tri[5]);
  d3 e2 = mk3(tri[6], tri[7], tri[8]);
  d3 dir_cross_e2 = cross3(d, e2);
  double det = dot3(e1, dir_cross_e2);
  if (approx_eq(det, 0.)) return 0;
  double f = 1. / det;
  d3 p1_to_origin = sub3(o, p1);
  u = f * dot3(p1_to_origin, dir_cross_e2);
  if (!(u >= 0. && u <= 1.)) return 0;
  d3 origin_cross_e1 = cross3(p1_to_origin, e1);
  v = f * dot3(d, origin_cross_e1);
  if (v < 0. || u + v > 1.) return 0;
  ts[0] = f * dot3(e2, origin_cross_e1);
  return 1;
}

// BoundingBox::intersection_time (bounding_box.rs:192-225); no tmax < 0 rejection, NaN dropped by fmax/fmin
__device__ __forceinline__ void axis_times(double origin, double dir_inv, double mn, double mx, double& tmin, double& tmax) {
  double a = (mn - origin) * dir_inv;
  double b = (mx - origin) * dir_inv;
  if (a < b) { tmin = a; tmax = b; } else { tmin = b; tmax = a; }
}
__device__ __forceinline__ bool bbox_time(const double* bb, d3 o, d3 dinv, double& out) {
  double xmin, xmax, ymin, ymax, zmin, zmax;
  axis_times(o.x, dinv.x, bb[0], bb[3], xmin, xmax);
  axis_times(o.y, dinv.y, bb[1], bb[4], ymin, ymax);
  double tmin = fmax(xmin, ymin);
  double tmax = fmin(xmax, ymax);
  if (tmin > tmax) return false;
  axis_times(o.z, dinv.z, bb[2], bb[5], zmin, zmax);
  tmin = fmax(tmin, zmin);
  tmax = fmin(tmax, zmax);
  if (tmin > tmax) return false;
  out = tmin;
  return true;
}

// PrimitiveObject::intersect (object.rs:432-439) + Shape::local_intersect (shape.rs:51-62)
template <bool STATS>
__device__ __forceinline__ int prim_roots(const DevScene& s, uint4 info, d3 o, d3 d, double* ts, double& u, double& v, Counters& cnt) {
  uint32_t shape = info.x;
  u = -INFINITY;
  v = -INFINITY;
  if (shape >= RT_SHAPE_TRIANGLE && shape != RT_SHAPE_BBOX) {
    if (STATS) cnt.tri_tests++;
    double uu, vv;
    int n = triangle_roots(s.tri_isect + (size_t)info.w * RT_TRI_STRIDE, o, d, ts, uu, vv);
    if (n && shape == RT_SHAPE_SMOOTH_TRIANGLE) {  // add_uv (intersection.rs:104-109); flat: u = v = -inf (:156-163)
      u = uu;
      v = vv;
    }
    return n;
  }
  if (STATS) cnt.prim_tests++;
  if (info.z != RT_NONE) {  // Ray::transform_new (ray.rs:21-23)
    const double* m = s.xform_inv + (size_t)info.z * 16;
    o = mat_point(m, o);
    d = mat_vector(m, d);
  }
  switch (shape) {
    case RT_SHAPE_SPHERE: return sphere_roots(o, d, ts);
    case RT_SHAPE_PLANE: return plane_roots(o, d, ts);
    case RT_SHAPE_CUBE:
    case RT_SHAPE_BBOX: return cube_roots(o, mk3(1. / d.x, 1. / d.y, 1. / d.z), ts);  // dir_inv of Ray::new (ray.rs:27-34), the only shape that reads it
    case RT_SHAPE_CYLINDER: {
      const double* q = s.quadric + (size_t)info.w * 2;
      return cylinder_roots(o, d, q[0], q[1], s.quadric_closed[info.w] != 0, ts);
    }
    default: {
      const double* q = s.quadric + (size_t)info.w * 2;
      return cone_roots(o, d, q[0], q[1], s.quadric_closed[info.w] != 0, ts);
    }
  }
}

// The same for an analytic primitive whose record is already in registers (DevLeafRec: rt_fast.h's PRIMS leaves).
template <bool STATS>
__device__ __forceinline__ int analytic_roots(uint32_t shape, uint32_t flags, const double* __restrict__ inv, double y_min, double y_max,
                                              d3 o, d3 d, double* ts, double& u, double& v, Counters& cnt) {
  u = -INFINITY;
  v = -INFINITY;
  if (STATS) cnt.prim_tests++;
  if (flags & RT_LEAFREC_XFORM) {  // Ray::transform_new (ray.rs:21-23)
    o = mat_point(inv, o);
    d = mat_vector(inv, d);
  }
  switch (shape) {
    case RT_SHAPE_SPHERE: return sphere_roots(o, d, ts);
    case RT_SHAPE_PLANE: return plane_roots(o, d, ts);
    case RT_SHAPE_CUBE:
    case RT_SHAPE_BBOX: return cube_roots(o, mk3(1. / d.x, 1. / d.y, 1. / d.z), ts);  // dir_inv of Ray::new (ray.rs:27-34)
    case RT_SHAPE_CYLINDER: return cylinder_roots(o, d, y_min, y_max, (flags & RT_LEAFREC_CLOSED) != 0, ts);
    default: return cone_roots(o, d, y_min, y_max, (flags & RT_LEAFREC_CLOSED) != 0, ts);
  }
}

#define RT_CSG_LIST_MAX 16   // hits kept per outermost CSG evaluation (rt_scene_create bounds the worst case of every CSG object by it)
#define RT_CSG_NEST_MAX 4    // nested CSG objects
#define RT_CSG_GROUP_STACK 16  // traversal stack of a group that is a CSG operand

// ---------------------------------------------------------------------------------------------
// CSG: CsgObject::intersect (csg.rs:112-122).  Children are primitives, nested CSG objects, or groups of primitives
// (a CSG object below a group below a CSG object is rejected at scene creation).  Iterative over an explicit frame stack;
// the hit list doubles as the nested collectors: a child CSG's survivors stay in place for its parent.
// Every level's temporary collector (IntersectionCollector::new(None, false), csg.rs:113) keeps all intersections and
// tracks its closest positive t (intersection.rs:65-77): that is what prunes the walk through a group operand
// (group.rs:182), so intersections deep inside a grouped operand can be missing from the list exactly as in the reference.
// ---------------------------------------------------------------------------------------------
struct CsgHit {
  double t, u, v;
  uint32_t prim;
};

// One primitive's roots -> the level's collector: list[n..) and its closest positive t.
template <bool STATS>
__device__ __forceinline__ void csg_add_prim(const DevScene& s, uint32_t idx, d3 o, d3 d, CsgHit* list, int& n, double& hit_time,
                                             Counters& cnt) {
  double ts[4], u, v;
  int k = prim_roots<STATS>(s, s.prim_info[idx], o, d, ts, u, v, cnt);
  for (int i = 0; i < k; i++) {
    if (sign_positive(ts[i]) && ts[i] < hit_time) hit_time = ts[i];
    if (n < RT_CSG_LIST_MAX) {
      list[n].t = ts[i];
      list[n].u = u;
      list[n].v = v;
      list[n].prim = idx;
      n++;
    } else {
      cnt.overflow = 1;
    }
  }
}

// ObjectGroup::intersect_iter (group.rs:171-201) for a group that is a CSG operand, into the level's collector.
// `entry`: the group's DevEntry (rt_scene_create appends one per group operand).
template <bool STATS>
__device__ __forceinline__ void csg_add_group(const DevScene& s, uint32_t entry, d3 o, d3 d, CsgHit* list, int& n,
                                              double& hit_time, Counters& cnt) {
  uint32_t stk_begin[RT_CSG_GROUP_STACK], stk_count[RT_CSG_GROUP_STACK];
  double stk_t[RT_CSG_GROUP_STACK];
  const d3 dinv = mk3(1. / d.x, 1. / d.y, 1. / d.z);  // Ray::new (ray.rs:27-34)
  int sp = 0;
  {
    const DevEntry* root = s.entries + entry;
    double bb[6], t;
    for (int k = 0; k < 6; k++) bb[k] = root->bb[k];
    if (STATS) cnt.node_visits++;
    if (!bbox_time(bb, o, dinv, t)) return;
    stk_begin[0] = root->begin;
    stk_count[0] = root->count;
    stk_t[0] = t;
    sp = 1;
  }
  while (sp > 0) {
    sp--;
    if (stk_t[sp] >= hit_time) continue;
    uint32_t begin = stk_begin[sp], count = stk_count[sp];
    for (uint32_t i = count; i-- > 0;) {
      const DevEntry* e = s.entries + (size_t)(begin + i);
      uint32_t ref = e->ref, kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
      if (kind == RT_REF_NODE) {
        double bb[6], t;
        for (int k = 0; k < 6; k++) bb[k] = e->bb[k];
        if (STATS) cnt.node_visits++;
        if (bbox_time(bb, o, dinv, t)) {
          if (sp < RT_CSG_GROUP_STACK) {
            stk_begin[sp] = e->begin;
            stk_count[sp] = e->count;
            stk_t[sp] = t;
            sp++;
          } else {
            cnt.overflow = 1;  // (cannot happen: rt_scene_create checks the nesting of group operands)
          }
        }
      } else if (kind == RT_REF_PRIM) {
        csg_add_prim<STATS>(s, idx, o, d, list, n, hit_time, cnt);
      }  // (a CSG object here is rejected by rt_scene_create)
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The GENERAL evaluation (rt_scene::general_csg, reference-order kernels only): what the bounded one above refuses — more than
// RT_CSG_LIST_MAX intersections per outermost object, nesting beyond RT_CSG_NEST_MAX, a CSG object inside a group that is
// itself a CSG operand (csg.rs:113-115 + group.rs:171-201 + csg.rs again) — with the reference's recursion
//   CsgObject::intersect -> Object::intersect(left / right, temp) -> { primitive | ObjectGroup::intersect_iter | CsgObject::intersect }
// as ONE explicit frame stack: a CSG frame owns a temporary collector (its part of the hit list, from `start`, and its
// closest positive t), a GROUP frame is a suspended intersect_iter (its pending sub-groups in a pool shared by the nested
// group frames, its cursor in the child list being walked) feeding the collector of the nearest CSG frame below it.
// Bounds, all checked by rt_scene_create except the list (a run-time flag -> RT_ERR_OVERFLOW): RT_CSG_BIG_LIST intersections
// alive at once, RT_CSG_GEN_FRAMES frames, RT_CSG_GROUP_NEST nested group operands of RT_CSG_BIG_GROUP_STACK pending sub-groups.
// ---------------------------------------------------------------------------------------------
#define RT_CSG_BIG_LIST 256
#define RT_CSG_GEN_FRAMES 16
#define RT_CSG_GROUP_NEST 4
#define RT_CSG_BIG_GROUP_STACK 24  // pending sub-groups of ONE group operand in the general evaluation (a model of 2^20 triangles
                                   // under the reference's builder is ~21 levels deep, one sibling pending per level)
#define RT_CSG_GEN_NEST_MAX 12  // CSG objects nested at most this deep (frames = CSG frames + group frames)
template <bool STATS>
__device__ __forceinline__ void csg_big_add_prim(const DevScene& s, uint32_t idx, d3 o, d3 d, CsgHit* list, int& n, double& hit_time,
                                                 Counters& cnt) {
  double ts[4], u, v;
  int k = prim_roots<STATS>(s, s.prim_info[idx], o, d, ts, u, v, cnt);
  for (int i = 0; i < k; i++) {
    if (sign_positive(ts[i]) && ts[i] < hit_time) hit_time = ts[i];
    if (n < RT_CSG_BIG_LIST) {
      list[n].t = ts[i];
      list[n].u = u;
      list[n].v = v;
      list[n].prim = idx;
      n++;
    } else {
      cnt.overflow = 1;
    }
  }
}
template <bool STATS>
__device__ int csg_survivors_general(const DevScene& s, uint32_t csg_root, d3 o, d3 d, CsgHit* list, Counters& cnt) {
  struct GFrame {
    uint32_t id;        // CSG frame: the object
    uint32_t begin;     // GROUP frame: the child list being walked (DevEntry index) ...
    uint32_t left;      // ... and how many of its children are still to visit (in reverse: the next is begin + left - 1)
    int start;          // CSG frame: where its collector starts in the list
    int owner;          // the CSG frame whose collector this frame's intersections go to (-1: the caller's, the outermost frame)
    int gs_base;        // GROUP frame: its part of the pool of pending sub-groups starts here
    uint8_t group, stage;
    double hit;         // CSG frame: hit_time() of its temporary collector
  };
  GFrame fr[RT_CSG_GEN_FRAMES];
  uint32_t gs_begin[RT_CSG_GROUP_NEST * RT_CSG_BIG_GROUP_STACK], gs_count[RT_CSG_GROUP_NEST * RT_CSG_BIG_GROUP_STACK];
  double gs_t[RT_CSG_GROUP_NEST * RT_CSG_BIG_GROUP_STACK];
  const d3 dinv = mk3(1. / d.x, 1. / d.y, 1. / d.z);  // Ray::new (ray.rs:27-34)
  int n = 0, fp = 0, gsp = 0;
  fr[0].id = csg_root;
  fr[0].group = 0;
  fr[0].stage = 0;
  fr[0].start = 0;
  fr[0].owner = -1;
  fr[0].hit = INFINITY;
  // Object::intersect(ref, collector of CSG frame `owner`) (object.rs:193-202): a primitive at once, anything else as a new frame
  auto enter = [&](uint32_t ref, int owner) {
    const uint32_t kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
    if (kind == RT_REF_PRIM) {
      csg_big_add_prim<STATS>(s, idx, o, d, list, n, fr[owner].hit, cnt);
      return;
    }
    if (fp + 1 >= RT_CSG_GEN_FRAMES) {  // (cannot happen: rt_scene_create bounds the nesting)
      cnt.overflow = 1;
      return;
    }
    GFrame& g = fr[fp + 1];
    g.owner = owner;
    if (kind == RT_REF_CSG) {
      g.id = idx;
      g.group = 0;
      g.stage = 0;
      g.start = n;
      g.hit = INFINITY;
      fp++;
      return;
    }
    // a group operand (idx: its DevEntry): ObjectGroup::intersect_iter starts with the group's own box (group.rs:172-175)
    const DevEntry* root = s.entries + idx;
    double bb[6], t;
    for (int k = 0; k < 6; k++) bb[k] = root->bb[k];
    if (STATS) cnt.node_visits++;
    if (!bbox_time(bb, o, dinv, t)) return;
    if (gsp + 1 > RT_CSG_GROUP_NEST * RT_CSG_BIG_GROUP_STACK) {
      cnt.overflow = 1;
      return;
    }
    g.group = 1;
    g.gs_base = gsp;
    g.left = 0;
    g.begin = 0;
    gs_begin[gsp] = root->begin;
    gs_count[gsp] = root->count;
    gs_t[gsp] = t;
    gsp++;
    fp++;
  };
  while (fp >= 0) {
    GFrame& F = fr[fp];
    if (F.group) {
      if (F.left == 0) {  // the next pending sub-group (group.rs:176-183)
        if (gsp == F.gs_base) {
          fp--;
          continue;
        }
        gsp--;
        if (gs_t[gsp] >= fr[F.owner].hit) continue;
        F.begin = gs_begin[gsp];
        F.left = gs_count[gsp];
        continue;
      }
      F.left--;  // children.iter().rev() (group.rs:185)
      const DevEntry* e = s.entries + (size_t)(F.begin + F.left);
      const uint32_t ref = e->ref, kind = ref >> RT_REF_KIND_SHIFT;
      if (kind == RT_REF_NODE) {
        double bb[6], t;
        for (int k = 0; k < 6; k++) bb[k] = e->bb[k];
        if (STATS) cnt.node_visits++;
        if (bbox_time(bb, o, dinv, t)) {
          if (gsp < RT_CSG_GROUP_NEST * RT_CSG_BIG_GROUP_STACK) {
            gs_begin[gsp] = e->begin;
            gs_count[gsp] = e->count;
            gs_t[gsp] = t;
            gsp++;
          } else {
            cnt.overflow = 1;
          }
        }
      } else {
        enter(ref, F.owner);  // (a CSG child suspends this walk: F.left is its cursor)
      }
      continue;
    }
    if (F.stage < 2) {
      const uint32_t ref = F.stage == 0 ? s.csg_left[F.id] : s.csg_right[F.id];
      F.stage++;
      enter(ref, fp);
      continue;
    }
    // both operands collected: collect_sorted (intersection.rs:110-114; a stable insertion sort) and filter_intersections
    // (csg.rs:87-111)
    const int start = F.start;
    for (int i = start + 1; i < n; i++) {
      CsgHit h = list[i];
      int j = i - 1;
      while (j >= start && list[j].t > h.t) {
        list[j + 1] = list[j];
        j--;
      }
      list[j + 1] = h;
    }
    bool inside_left = false, inside_right = false;
    const uint32_t op = s.csg_op[F.id], depth = s.csg_depth[F.id];
    int w = start;
    for (int i = start; i < n; i++) {
      const bool hit_left = (s.prim_csg_left[list[i].prim] >> depth) & 1u;  // self.left.includes(i.object())
      if (hit_left) inside_left = !inside_left;
      else inside_right = !inside_right;
      bool allowed;
      if (op == RT_CSG_UNION) allowed = hit_left ? !inside_right : !inside_left;
      else if (op == RT_CSG_INTERSECTION) allowed = hit_left ? inside_right : inside_left;
      else allowed = hit_left ? !inside_right : inside_left;
      if (allowed) list[w++] = list[i];
    }
    n = w;
    const int owner = F.owner;
    fp--;
    if (owner >= 0)  // the survivors are push()ed to the collector this object was intersected into (csg.rs:119-121)
      for (int i = start; i < n; i++)
        if (sign_positive(list[i].t) && list[i].t < fr[owner].hit) fr[owner].hit = list[i].t;
  }
  return n;
}

// Fills list[0..n) with the surviving intersections of the outermost CSG object `csg_root`, ascending t.
template <bool STATS>
__device__ __forceinline__ int csg_survivors(const DevScene& s, uint32_t csg_root, d3 o, d3 d, CsgHit* list, Counters& cnt) {
  uint32_t f_csg[RT_CSG_NEST_MAX];
  uint8_t f_stage[RT_CSG_NEST_MAX];
  uint8_t f_start[RT_CSG_NEST_MAX];
  double f_hit[RT_CSG_NEST_MAX];  // hit_time() of the level's temporary collector
  int n = 0, fp = 0;
  f_csg[0] = csg_root;
  f_stage[0] = 0;
  f_start[0] = 0;
  f_hit[0] = INFINITY;
  while (fp >= 0) {
    uint32_t csg = f_csg[fp];
    if (f_stage[fp] < 2) {
      uint32_t ref = f_stage[fp] == 0 ? s.csg_left[csg] : s.csg_right[csg];
      f_stage[fp]++;
      uint32_t kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
      if (kind == RT_REF_PRIM) {
        csg_add_prim<STATS>(s, idx, o, d, list, n, f_hit[fp], cnt);
      } else if (kind == RT_REF_NODE) {  // idx: the operand's DevEntry
        csg_add_group<STATS>(s, idx, o, d, list, n, f_hit[fp], cnt);
      } else if (kind == RT_REF_CSG && fp + 1 < RT_CSG_NEST_MAX) {
        fp++;
        f_csg[fp] = idx;
        f_stage[fp] = 0;
        f_start[fp] = (uint8_t)n;
        f_hit[fp] = INFINITY;
      }
      continue;
    }
    // both children collected: sort (collect_sorted, intersection.rs:110-114; stable insertion sort == the
    // reference's sort for short slices) and filter (csg.rs:87-111)
    int start = f_start[fp];
    for (int i = start + 1; i < n; i++) {
      CsgHit h = list[i];
      int j = i - 1;
      while (j >= start && list[j].t > h.t) {
        list[j + 1] = list[j];
        j--;
      }
      list[j + 1] = h;
    }
    bool inside_left = false, inside_right = false;
    uint32_t op = s.csg_op[csg];
    uint32_t depth = s.csg_depth[csg];
    int w = start;
    for (int i = start; i < n; i++) {
      bool hit_left = (s.prim_csg_left[list[i].prim] >> depth) & 1u;  // self.left.includes(i.object())
      if (hit_left) inside_left = !inside_left;
      else inside_right = !inside_right;
      bool allowed;
      if (op == RT_CSG_UNION) allowed = hit_left ? !inside_right : !inside_left;
      else if (op == RT_CSG_INTERSECTION) allowed = hit_left ? inside_right : inside_left;
      else allowed = hit_left ? !inside_right : inside_left;
      if (allowed) list[w++] = list[i];
    }
    n = w;
    fp--;
    if (fp >= 0)  // the survivors are push()ed to the parent's collector (csg.rs:119-121)
      for (int i = start; i < n; i++)
        if (sign_positive(list[i].t) && list[i].t < f_hit[fp]) f_hit[fp] = list[i].t;
  }
  return n;
}
