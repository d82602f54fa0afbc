// rt_kernels.h — the per-pixel render core as gfx950 device code.
//
// One wavefront (64 lanes) renders one 8x8 pixel tile; each lane owns one pixel and runs the whole
// Renderer::color_at_pixel (src/render/renderer.rs:50-66) for it: supersample loop, Whitted integrator with
// the reflect/refract recursion unrolled into an explicit per-lane frame stack (post-order, so the arithmetic
// order equals the recursion of src/shading/integrator.rs:199-228), BVH traversal with an explicit stack in
// the reference's DFS order (src/scene/object/group.rs:171-201), all seven shapes + CSG.
#pragma once
#include "rt_device.h"

#define RT_STACK_MAX 64      // BVH traversal stack entries per lane (validated against the scene at create time)
#define RT_CSG_LIST_MAX 16   // hits kept per outermost CSG evaluation
#define RT_CSG_NEST_MAX 4    // nested CSG objects
#define RT_FRAMES (RT_MAX_DEPTH + 1)

enum { MODE_CLOSEST = 0, MODE_SHADOW = 1 };

struct Counters {
  uint32_t rays_primary, rays_shadow, rays_reflect, rays_refract, node_visits, tri_tests, prim_tests, hits, hit_bytes, overflow;
};

// Closest-hit / shadow collector: IntersectionCollector (src/render/ray/intersection.rs:17-135) without the Vec.
//  * closest mode keeps the hit (strict `<`, first found wins, +0 accepted, -0 rejected: :65-77) and, instead of
//    the list of negative-t intersections, what get_refractive (:258-302) derives from it: the two objects with
//    an odd number of negative-t hits whose largest negative t is greatest (the top two "containers"), and
//    whether the hit object itself has an odd count.
//  * shadow mode accumulates 1 - transparency for every intersection with +0 <= t < dist, !approx_eq(t, dist),
//    while the sum is < 1 (:78-87); `limit` is the constant light distance.
struct Coll {
  double limit;  // hit_time(): closest hit so far (INF) or light distance
  uint32_t hit_prim;
  double hu, hv;
  bool hit_neg_odd;
  double ta, tb;  // containers: largest negative t of the best / second best odd-count object
  uint32_t pa, pb;
  double intensity;
};

__device__ __forceinline__ void coll_init_closest(Coll& c) {
  c.limit = INFINITY;
  c.hit_prim = RT_NONE;
  c.hu = c.hv = -INFINITY;
  c.hit_neg_odd = false;
  c.ta = c.tb = -INFINITY;
  c.pa = c.pb = RT_NONE;
  c.intensity = 0.;
}
__device__ __forceinline__ void coll_init_shadow(Coll& c, double dist) {
  coll_init_closest(c);
  c.limit = dist;
}

__device__ __forceinline__ void coll_container(Coll& c, double t, uint32_t prim) {
  if (t >= c.ta) {  // later insertions win ties (stable sort keeps insertion order)
    c.tb = c.ta; c.pb = c.pa;
    c.ta = t; c.pa = prim;
  } else if (t >= c.tb) {
    c.tb = t; c.pb = prim;
  }
}

// All roots of ONE primitive, in the order the reference add()s them.
template <int MODE>
__device__ __forceinline__ void coll_add_roots(const DevScene& s, Coll& c, uint32_t prim, uint32_t material, int n,
                                               const double* ts, double u, double v) {
  if (MODE == MODE_CLOSEST) {
    int neg = 0;
    double maxneg = -INFINITY;
    bool took = false;
    for (int i = 0; i < n; i++) {
      double t = ts[i];
      if (sign_positive(t)) {
        if (t < c.limit) {
          c.limit = t;
          c.hit_prim = prim;
          c.hu = u;
          c.hv = v;
          took = true;
        }
      } else {
        neg++;
        maxneg = fmax(maxneg, t);
      }
    }
    if (neg & 1) coll_container(c, maxneg, prim);
    if (took) c.hit_neg_odd = (neg & 1) != 0;
  } else {
    double transparency = 0.;
    bool loaded = false;
    for (int i = 0; i < n; i++) {
      double t = ts[i];
      if (sign_positive(t) && !approx_eq(t, c.limit) && t < c.limit && c.intensity < 1.) {
        if (!loaded) {
          transparency = s.material[(size_t)material * RT_MAT_STRIDE + RT_MAT_TRANSPARENCY];
          loaded = true;
        }
        c.intensity += 1. - transparency;
      }
    }
  }
}

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

// PrimitiveObject::intersect (object.rs:432-439) + Shape::local_intersect (shape.rs:51-62)
template <bool STATS>
__device__ __forceinline__ int prim_roots(const DevScene& s, uint4 info, d3 o, d3 d, d3 dinv, double* ts, double& u, double& v,
                                          Counters& cnt) {
  uint32_t shape = info.x;
  u = -INFINITY;
  v = -INFINITY;
  if (shape >= RT_SHAPE_TRIANGLE && shape != RT_SHAPE_BBOX) {
    if (STATS) cnt.tri_tests++;
    double uu, vv;
    int n = triangle_roots(s.tri_isect + (size_t)info.w * 9, o, d, ts, uu, vv);
    if (n && shape == RT_SHAPE_SMOOTH_TRIANGLE) {  // add_uv (intersection.rs:104-109); flat: u = v = -inf (:156-163)
      u = uu;
      v = vv;
    }
    return n;
  }
  if (STATS) cnt.prim_tests++;
  if (info.z != RT_NONE) {  // Ray::transform_new (ray.rs:21-23) -> Ray::new recomputes dir_inv (ray.rs:27-34)
    const double* m = s.xform_inv + (size_t)info.z * 16;
    o = mat_point(m, o);
    d = mat_vector(m, d);
    dinv = mk3(1. / d.x, 1. / d.y, 1. / d.z);
  }
  switch (shape) {
    case RT_SHAPE_SPHERE: return sphere_roots(o, d, ts);
    case RT_SHAPE_PLANE: return plane_roots(o, d, ts);
    case RT_SHAPE_CUBE:
    case RT_SHAPE_BBOX: return cube_roots(o, dinv, ts);
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

// ---------------------------------------------------------------------------------------------
// CSG: CsgObject::intersect (csg.rs:112-122).  Children are primitives or nested CSG objects
// (groups inside CSG are rejected at scene creation).  Iterative over an explicit frame stack; the
// hit list doubles as the nested collectors: a child CSG's survivors stay in place for its parent.
// ---------------------------------------------------------------------------------------------
struct CsgHit {
  double t, u, v;
  uint32_t prim;
};

template <int MODE, bool STATS>
__device__ __noinline__ void csg_intersect(const DevScene& s, uint32_t csg_root, d3 o, d3 d, d3 dinv, Coll& c, Counters& cnt) {
  CsgHit list[RT_CSG_LIST_MAX];
  uint32_t f_csg[RT_CSG_NEST_MAX];
  uint8_t f_stage[RT_CSG_NEST_MAX];
  uint8_t f_start[RT_CSG_NEST_MAX];
  int n = 0, fp = 0;
  f_csg[0] = csg_root;
  f_stage[0] = 0;
  f_start[0] = 0;
  while (fp >= 0) {
    uint32_t csg = f_csg[fp];
    if (f_stage[fp] < 2) {
      uint32_t ref = f_stage[fp] == 0 ? s.csg_left[csg] : s.csg_right[csg];
      f_stage[fp]++;
      uint32_t kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
      if (kind == RT_REF_PRIM) {
        double ts[4], u, v;
        int k = prim_roots<STATS>(s, s.prim_info[idx], o, d, dinv, ts, u, v, cnt);
        for (int i = 0; i < k; i++) {
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
      } else if (kind == RT_REF_CSG && fp + 1 < RT_CSG_NEST_MAX) {
        fp++;
        f_csg[fp] = idx;
        f_stage[fp] = 0;
        f_start[fp] = (uint8_t)n;
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
  }
  // survivors -> outer collector, ascending t: collector.push(i) (csg.rs:119-121)
  if (MODE == MODE_CLOSEST) {
    int took = -1;
    for (int i = 0; i < n; i++) {
      double t = list[i].t;
      if (sign_positive(t) && t < c.limit) {
        c.limit = t;
        c.hit_prim = list[i].prim;
        c.hu = list[i].u;
        c.hv = list[i].v;
        took = i;
      }
    }
    // negative-t survivors, aggregated per leaf object (they all come from this one call)
    for (int i = 0; i < n; i++) {
      if (sign_positive(list[i].t)) continue;
      bool first = true;
      for (int j = 0; j < i; j++)
        if (!sign_positive(list[j].t) && list[j].prim == list[i].prim) first = false;
      if (!first) continue;
      int neg = 0;
      double maxneg = -INFINITY;
      for (int j = i; j < n; j++)
        if (!sign_positive(list[j].t) && list[j].prim == list[i].prim) {
          neg++;
          maxneg = fmax(maxneg, list[j].t);
        }
      if (neg & 1) coll_container(c, maxneg, list[i].prim);
    }
    if (took >= 0) {
      int neg = 0;
      for (int j = 0; j < n; j++)
        if (!sign_positive(list[j].t) && list[j].prim == c.hit_prim) neg++;
      c.hit_neg_odd = (neg & 1) != 0;
    }
  } else {
    for (int i = 0; i < n; i++) {
      double t = list[i].t;
      if (sign_positive(t) && !approx_eq(t, c.limit) && t < c.limit && c.intensity < 1.) {
        uint32_t material = s.prim_info[list[i].prim].y;
        c.intensity += 1. - s.material[(size_t)material * RT_MAT_STRIDE + RT_MAT_TRANSPARENCY];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ObjectGroup::intersect_iter (group.rs:171-201): explicit stack of (group, t_enter); a popped group is skipped
// when t_enter >= hit_time(); its children are visited in REVERSE order, primitives/CSG intersected at once,
// hit child groups pushed (so they pop in forward order).
// ---------------------------------------------------------------------------------------------
template <int MODE, bool STATS>
__device__ __forceinline__ void traverse(const DevScene& s, d3 o, d3 d, Coll& c, Counters& cnt) {
  uint32_t stk_begin[RT_STACK_MAX];
  uint32_t stk_count[RT_STACK_MAX];
  double stk_t[RT_STACK_MAX];
  d3 dinv = mk3(1. / d.x, 1. / d.y, 1. / d.z);  // Ray::new (ray.rs:27-34)
  int sp = 0;
  {
    const DevEntry* root = s.entries;
    double bb[6];
    for (int k = 0; k < 6; k++) bb[k] = root->bb[k];
    double t;
    if (STATS) cnt.node_visits++;
    if (!bbox_time(bb, o, dinv, t)) return;
    stk_begin[0] = root->begin;
    stk_count[0] = root->count;
    stk_t[0] = t;
    sp = 1;
  }
  while (sp > 0) {
    sp--;
    double time = stk_t[sp];
    if (time >= c.limit) continue;
    uint32_t begin = stk_begin[sp], count = stk_count[sp];
    for (uint32_t i = count; i-- > 0;) {
      const DevEntry* e = s.entries + (size_t)(begin + i);
      uint32_t ref = e->ref;
      uint32_t kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
      if (kind == RT_REF_NODE) {
        double bb[6];
        for (int k = 0; k < 6; k++) bb[k] = e->bb[k];
        double t;
        if (STATS) cnt.node_visits++;
        if (bbox_time(bb, o, dinv, t)) {
          if (sp < RT_STACK_MAX) {
            stk_begin[sp] = e->begin;
            stk_count[sp] = e->count;
            stk_t[sp] = t;
            sp++;
          } else {
            cnt.overflow = 1;
          }
        }
      } else if (kind == RT_REF_PRIM) {
        uint4 info = s.prim_info[idx];
        double ts[4], u, v;
        int n = prim_roots<STATS>(s, info, o, d, dinv, ts, u, v, cnt);
        if (n) coll_add_roots<MODE>(s, c, idx, info.y, n, ts, u, v);
      } else {
        csg_intersect<MODE, STATS>(s, idx, o, d, dinv, c, cnt);
      }
    }
    if (MODE == MODE_SHADOW && c.intensity >= 1.) return;  // min(intensity, 1) is already 1 (intersection.rs:124-126)
  }
}

// ---------------------------------------------------------------------------------------------
// Normals: PrimitiveObject::normal_vector_at_with_intersection (object.rs:415-430), Shape::local_normal_at (shape.rs:35-50)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ d3 local_normal(const DevScene& s, uint4 info, d3 p, double u, double v) {
  switch (info.x) {
    case RT_SHAPE_SPHERE: return sub3(p, mk3(0., 0., 0.));  // sphere.rs:10-12
    case RT_SHAPE_PLANE: return mk3(0., 1., 0.);            // plane.rs:10-12
    case RT_SHAPE_CUBE:
    case RT_SHAPE_BBOX: {  // cube.rs:10-23 (x wins ties, then y)
      double ax = fabs(p.x), ay = fabs(p.y), az = fabs(p.z);
      double m = fmax(fmax(ax, ay), az);
      if (m == ax) return mk3(p.x, 0., 0.);
      if (m == ay) return mk3(0., p.y, 0.);
      return mk3(0., 0., p.z);
    }
    case RT_SHAPE_CYLINDER: {  // cylinder.rs:42-53
      const double* q = s.quadric + (size_t)info.w * 2;
      double dist = p.x * p.x + p.z * p.z;
      bool within = dist < 1.;
      if (within && p.y >= q[1] - RT_EPS) return mk3(0., 1., 0.);
      if (within && p.y <= q[0] + RT_EPS) return mk3(0., -1., 0.);
      return mk3(p.x, 0., p.z);
    }
    case RT_SHAPE_CONE: {  // cone.rs:38-55
      const double* q = s.quadric + (size_t)info.w * 2;
      double dist = p.x * p.x + p.z * p.z;
      if (dist < fabs(q[1]) && p.y >= q[1] - RT_EPS) return mk3(0., 1., 0.);
      if (dist < fabs(q[0]) && p.y <= q[0] + RT_EPS) return mk3(0., -1., 0.);
      double y = sqrt(p.x * p.x + p.z * p.z);
      if (p.y > 0.) y = -y;
      return mk3(p.x, y, p.z);
    }
    case RT_SHAPE_TRIANGLE: {  // triangle.rs:101-103
      const double* nn = s.tri_norm + (size_t)info.w * 9;
      return mk3(nn[0], nn[1], nn[2]);
    }
    default: {  // smooth_triangle.rs:56-59: n2*u + n3*v + n1*(1-u-v)
      const double* nn = s.tri_norm + (size_t)info.w * 9;
      d3 n1 = mk3(nn[0], nn[1], nn[2]), n2 = mk3(nn[3], nn[4], nn[5]), n3 = mk3(nn[6], nn[7], nn[8]);
      return add3(add3(scale3(n2, u), scale3(n3, v)), scale3(n1, 1. - u - v));
    }
  }
}

__device__ __forceinline__ d3 normal_at(const DevScene& s, uint4 info, d3 scene_point, double u, double v) {
  d3 n;
  if (info.z == RT_NONE) {
    n = local_normal(s, info, scene_point, u, v);
  } else {
    const double* inv = s.xform_inv + (size_t)info.z * 16;
    d3 object_point = mat_point(inv, scene_point);
    d3 object_normal = local_normal(s, info, object_point, u, v);
    n = mat_transposed_vector(inv, object_normal);
  }
  return normalize3(n);
}

// Pattern::color_at_object (pattern.rs:113-128) + color_at (:78-111)
__device__ __forceinline__ d3 pattern_color(const DevScene& s, uint4 info, d3 point) {
  uint32_t m = info.y;
  uint32_t kind = s.pattern_kind[m];
  const double* pc1 = s.pattern_c1 + (size_t)m * 3;
  d3 c1 = mk3(pc1[0], pc1[1], pc1[2]);
  if (kind == RT_PATTERN_CONST) return c1;
  const double* pc2 = s.pattern_c2 + (size_t)m * 3;
  d3 c2 = mk3(pc2[0], pc2[1], pc2[2]);
  d3 object_point;
  if (info.z == RT_NONE) {
    // transformation_inverse() is the identity (object.rs:409-411) but the product is still formed
    object_point = mk3(1. * point.x + 0. * point.y + 0. * point.z + 0. * 1., 0. * point.x + 1. * point.y + 0. * point.z + 0. * 1.,
                       0. * point.x + 0. * point.y + 1. * point.z + 0. * 1.);
  } else {
    object_point = mat_point(s.xform_inv + (size_t)info.z * 16, point);
  }
  d3 p = mat_point(s.pattern_inv + (size_t)m * 16, object_point);
  switch (kind) {
    case RT_PATTERN_STRIPE: return approx_eq(fmod(floor(p.x), 2.), 0.) ? c1 : c2;
    case RT_PATTERN_GRADIENT: {
      double f = p.x - floor(p.x);
      return add3(c1, mk3(c2.x * f, c2.y * f, c2.z * f));
    }
    case RT_PATTERN_RING: {
      double val = floor(sqrt(p.x * p.x + p.z * p.z));
      return approx_eq(fmod(val, 2.), 0.) ? c1 : c2;
    }
    case RT_PATTERN_CHECKERS: {
      double x = floor(p.x), y = floor(p.y), z = floor(p.z);
      return approx_eq(fmod(x + y + z, 2.), 0.) ? c1 : c2;
    }
    default: return p;  // TestPattern
  }
}

// PointLightSource::color_of_illuminated_point (point_light.rs:38-76)
__device__ __forceinline__ d3 lighting(const DevScene& s, uint32_t light, uint4 info, const double* __restrict__ mat, d3 point,
                                       d3 eye, d3 normal, double shadow) {
  const double* li = s.light_intensity + (size_t)light * 3;
  const double* lp = s.light_pos + (size_t)light * 3;
  d3 intensity = mk3(li[0], li[1], li[2]);
  d3 effective = cmul3(pattern_color(s, info, point), intensity);
  d3 light_v = normalize3(sub3(mk3(lp[0], lp[1], lp[2]), point));
  double ka = mat[RT_MAT_AMBIENT];
  d3 ambient = mk3(effective.x * ka, effective.y * ka, effective.z * ka);
  double ldn = dot3(light_v, normal);
  if (approx_eq(shadow, 1.) || ldn < 0.) {  // ambient + black + black
    return mk3(ambient.x + 0. + 0., ambient.y + 0. + 0., ambient.z + 0. + 0.);
  }
  double kd = mat[RT_MAT_DIFFUSE];
  d3 diffuse = mk3(effective.x * kd * ldn, effective.y * kd * ldn, effective.z * kd * ldn);
  d3 reflect_v = reflect3(neg3(light_v), normal);
  double rde = dot3(reflect_v, eye);
  d3 specular = mk3(0., 0., 0.);
  if (sign_positive(rde)) {
    double factor = pow(rde, mat[RT_MAT_SHININESS]);
    double ks = mat[RT_MAT_SPECULAR];
    specular = mk3(intensity.x * ks * factor, intensity.y * ks * factor, intensity.z * ks * factor);
  }
  double k = 1. - shadow;
  return mk3(ambient.x + (diffuse.x + specular.x) * k, ambient.y + (diffuse.y + specular.y) * k,
             ambient.z + (diffuse.z + specular.z) * k);
}

// schlick_reflectance (light.rs:5-27); powi(5) = x * (x^2)^2
__device__ __forceinline__ double schlick(d3 eye, d3 normal, double n1, double n2) {
  double cosv = dot3(eye, normal);
  if (n1 > n2) {
    double ratio = n1 / n2;
    double sin2_t = (ratio * ratio) * (1. - cosv * cosv);
    if (sin2_t > 1.) return 1.;
    cosv = sqrt(1. - sin2_t);
  }
  double q = (n1 - n2) / (n1 + n2);
  double r0 = q * q;
  double x = 1. - cosv;
  double x2 = x * x;
  double x5 = x * (x2 * x2);
  return r0 + (1. - r0) * x5;
}

// ---------------------------------------------------------------------------------------------
// Integrator::color_at (integrator.rs:59-70) with the recursion of shade_hit_at_depth (:199-228) unrolled.
// Frame states: a frame first traces its ray; if the hit needs a reflected ray the child frame runs and returns
// into AFTER_REFLECT, likewise AFTER_REFRACT; then the light fold runs with both subtree colours in hand (the
// reference recomputes them per light; they are pure functions of the hit, so once is bit-identical).
// ---------------------------------------------------------------------------------------------
struct Frame {
  double ox, oy, oz, dx, dy, dz;  // the ray of this frame
  double t, u, v;                 // its hit
  double n1, n2;                  // refractive indices (intersection.rs:258-302)
  double rr, rg, rb;              // reflected colour, already times reflectivity
  uint32_t prim;
  uint32_t state;
};
enum { ST_ENTER = 0, ST_AFTER_REFLECT = 1, ST_AFTER_REFRACT = 2 };

struct HitGeom {  // IntersecComputations::new (intersection.rs:217-256)
  d3 point, eye, normal, over, under, reflect_v;
};
__device__ __forceinline__ HitGeom hit_geom(const DevScene& s, uint4 info, d3 o, d3 d, double t, double u, double v) {
  HitGeom g;
  g.point = add3(o, scale3(d, t));  // Ray::position (ray.rs:36-38)
  g.eye = neg3(d);
  d3 n = normal_at(s, info, g.point, u, v);
  if (dot3(n, g.eye) < 0.) n = neg3(n);
  g.normal = n;
  d3 off = scale3(n, RT_EPS);
  g.over = add3(g.point, off);
  g.under = sub3(g.point, off);
  g.reflect_v = reflect3(d, n);
  return g;
}

template <bool STATS>
__device__ d3 trace(const DevScene& s, const DevOptions& opt, d3 ro, d3 rd, Counters& cnt) {
  Frame frames[RT_FRAMES];
  int depth = 0;
  frames[0].ox = ro.x; frames[0].oy = ro.y; frames[0].oz = ro.z;
  frames[0].dx = rd.x; frames[0].dy = rd.y; frames[0].dz = rd.z;
  frames[0].state = ST_ENTER;
  d3 ret = mk3(0., 0., 0.);
  for (;;) {
    Frame& F = frames[depth];
    d3 o = mk3(F.ox, F.oy, F.oz), d = mk3(F.dx, F.dy, F.dz);
    uint32_t state = F.state;
    bool finished = false;
    uint4 info;
    const double* mat = nullptr;
    if (state == ST_ENTER) {
      Coll c;
      coll_init_closest(c);
      traverse<MODE_CLOSEST, STATS>(s, o, d, c, cnt);
      if (c.hit_prim == RT_NONE) {
        ret = mk3(0., 0., 0.);
        finished = true;
      } else {
        F.t = c.limit;
        F.u = c.hu;
        F.v = c.hv;
        F.prim = c.hit_prim;
        info = s.prim_info[c.hit_prim];
        mat = s.material + (size_t)info.y * RT_MAT_STRIDE;
        double n1 = 1.0, n2 = 1.0;  // AIR_REFRACTIVE_INDEX (material.rs:7)
        if (!approx_eq(mat[RT_MAT_TRANSPARENCY], 0.)) {  // Intersection::compute_refractive (:184-190)
          if (c.pa != RT_NONE) n1 = s.material[(size_t)s.prim_info[c.pa].y * RT_MAT_STRIDE + RT_MAT_REFRACTIVE_INDEX];
          if (c.hit_neg_odd) {  // the hit object leaves the container list
            uint32_t last = (c.pa == c.hit_prim) ? c.pb : c.pa;
            if (last != RT_NONE) n2 = s.material[(size_t)s.prim_info[last].y * RT_MAT_STRIDE + RT_MAT_REFRACTIVE_INDEX];
          } else {  // the hit object becomes the innermost container
            n2 = mat[RT_MAT_REFRACTIVE_INDEX];
          }
        }
        F.n1 = n1;
        F.n2 = n2;
        F.rr = F.rg = F.rb = 0.;
        if (STATS) {  // algorithmic bytes of shading this hit
          cnt.hits++;
          uint32_t b = 64u + 48u * s.n_lights;
          b += info.x == RT_SHAPE_SMOOTH_TRIANGLE ? 72u : (info.x == RT_SHAPE_TRIANGLE ? 24u : (info.z != RT_NONE ? 96u : 0u));
          if (s.pattern_kind[info.y] != RT_PATTERN_CONST) b += 144u;
          cnt.hit_bytes += b;
        }
        // reflected_color_at_depth (integrator.rs:154-168)
        if ((uint32_t)depth < opt.max_depth && !approx_eq(mat[RT_MAT_REFLECTIVITY], 0.)) {
          HitGeom g = hit_geom(s, info, o, d, F.t, F.u, F.v);
          F.state = ST_AFTER_REFLECT;
          Frame& C = frames[depth + 1];
          C.ox = g.over.x; C.oy = g.over.y; C.oz = g.over.z;
          C.dx = g.reflect_v.x; C.dy = g.reflect_v.y; C.dz = g.reflect_v.z;
          C.state = ST_ENTER;
          if (STATS) cnt.rays_reflect++;
          depth++;
          continue;
        }
        state = ST_AFTER_REFLECT;  // no reflected ray: reflected = black
      }
    } else {
      info = s.prim_info[F.prim];
      mat = s.material + (size_t)info.y * RT_MAT_STRIDE;
      if (state == ST_AFTER_REFLECT) {  // color * reflectivity (integrator.rs:167)
        double kr = mat[RT_MAT_REFLECTIVITY];
        F.rr = ret.x * kr;
        F.rg = ret.y * kr;
        F.rb = ret.z * kr;
      }
    }
    if (!finished) {
      HitGeom g = hit_geom(s, info, o, d, F.t, F.u, F.v);
      d3 refracted = mk3(0., 0., 0.);
      if (state == ST_AFTER_REFLECT) {
        // refracted_color_at_depth (integrator.rs:170-197)
        if ((uint32_t)depth < opt.max_depth && !approx_eq(mat[RT_MAT_TRANSPARENCY], 0.)) {
          double ratio = F.n1 / F.n2;
          double cos_i = dot3(g.eye, g.normal);
          double sin2_t = (ratio * ratio) * (1. - cos_i * cos_i);
          if (!(sin2_t > 1.)) {
            double cos_t = sqrt(1. - sin2_t);
            d3 dir = sub3(scale3(g.normal, ratio * cos_i - cos_t), scale3(g.eye, ratio));
            F.state = ST_AFTER_REFRACT;
            Frame& C = frames[depth + 1];
            C.ox = g.under.x; C.oy = g.under.y; C.oz = g.under.z;
            C.dx = dir.x; C.dy = dir.y; C.dz = dir.z;
            C.state = ST_ENTER;
            if (STATS) cnt.rays_refract++;
            depth++;
            continue;
          }
        }
      } else {  // ST_AFTER_REFRACT: color * transparency (integrator.rs:196)
        double kt = mat[RT_MAT_TRANSPARENCY];
        refracted = mk3(ret.x * kt, ret.y * kt, ret.z * kt);
      }
      // shade_hit_at_depth (integrator.rs:199-228)
      d3 reflected = mk3(F.rr, F.rg, F.rb);
      double kr = mat[RT_MAT_REFLECTIVITY], kt = mat[RT_MAT_TRANSPARENCY];
      bool use_schlick = kr > 0. && kt > 0. && !approx_eq(kr, 0.) && !approx_eq(kt, 0.);
      d3 rr;
      if (use_schlick) {
        double reflectance = schlick(g.eye, g.normal, F.n1, F.n2);
        double k2 = 1. - reflectance;
        rr = mk3(reflected.x * reflectance + refracted.x * k2, reflected.y * reflectance + refracted.y * k2,
                 reflected.z * reflectance + refracted.z * k2);
      } else {
        rr = add3(reflected, refracted);
      }
      d3 acc = mk3(0., 0., 0.);
      for (uint32_t l = 0; l < s.n_lights; l++) {
        // point_shadow_intensity_comps (integrator.rs:136-152) from over_point
        const double* lp = s.light_pos + (size_t)l * 3;
        d3 vv = sub3(mk3(lp[0], lp[1], lp[2]), g.over);
        double dist = mag3(vv);
        d3 sdir = normalize3(vv);
        Coll sc;
        coll_init_shadow(sc, dist);
        if (STATS) cnt.rays_shadow++;
        traverse<MODE_SHADOW, STATS>(s, g.over, sdir, sc, cnt);
        double shadow = fmin(sc.intensity, 1.);
        if (!opt.shadow_intensity) shadow = approx_eq(shadow, 0.) ? 0. : 1.;
        d3 surface = lighting(s, l, info, mat, g.over, g.eye, g.normal, shadow);
        acc = add3(add3(acc, surface), rr);  // acc + surface + reflected_refracted
      }
      ret = acc;
    }
    if (depth == 0) return ret;
    depth--;
  }
}

// Camera::ray_for_pixel (camera.rs:189-201)
__device__ __forceinline__ void camera_ray(const rt_camera& cam, double x, double y, d3& o, d3& d) {
  double xo = (x + 0.5) * cam.pixel_size;
  double yo = (y + 0.5) * cam.pixel_size;
  double sx = cam.half_width - xo;
  double sy = cam.half_height - yo;
  d3 pixel = mat_point(cam.inv_view, mk3(sx, sy, -1.));
  o = mat_point(cam.inv_view, mk3(0., 0., 0.));
  d = normalize3(sub3(pixel, o));
}
