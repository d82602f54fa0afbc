// rt_oracle.cpp — CPU ORACLE for the per-pixel render core of karolzwolak/raytracer.
//
// THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load it; the product path (raytracer_amd/csrc, librtcore.so)
// never links, calls or falls back to anything in this directory.
//
// What it is: a plain C++ restatement of the reference's algorithm for the hot path, function by
// function, recursion and per-ray Vec bookkeeping included, in the reference's operation order
// (compile with -ffp-contract=off: rustc never fuses a*b+c).  It consumes the same flattened tables
// as the device library (include/rtcore.h).  Each function cites the reference lines it follows.
//
// How it is pinned: the Rust reference cannot be compiled here (no rustc/cargo in the
// image), so the oracle is checked against (1) the reference's 12 committed golden PNGs
// (tests-reference-renders/images, 800x800, --depth 4 --supersampling-level 1), (2) its 4 showcase PNGs
// (depth 5, ss 2) and (3) the known-answer values of its in-file unit tests, transcribed as data in
// tests/test_oracle_known_answers.py.  See DESIGN.md "Oracle".
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rtcore.h"

namespace {

constexpr double EPSILON = 1.0e-8;  // src/math/approx_eq.rs:1
inline bool approx_eq(double a, double b) { return a == b || std::fabs(a - b) < EPSILON; }  // approx_eq.rs:16-20
inline bool sign_positive(double x) { return !std::signbit(x); }

struct V3 {
  double x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 mulf(V3 a, double f) { return {f * a.x, f * a.y, f * a.z}; }   // vector.rs:136-146
inline V3 cmulf(V3 a, double f) { return {a.x * f, a.y * f, a.z * f}; }  // color.rs:125-135
inline V3 cmul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; } // color.rs:114-124
inline V3 cdivf(V3 a, double f) { return {a.x / f, a.y / f, a.z / f}; }  // color.rs:136-146
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double magnitude(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // vector.rs:54-56
inline V3 normalize(V3 a) {                                                           // vector.rs:57-67
  double len = magnitude(a);
  if (len == 0.) return {0., 0., 0.};
  return {a.x / len, a.y / len, a.z / len};
}
inline V3 reflect(V3 v, V3 n) { return v - mulf(mulf(n, 2.), dot(v, n)); }  // vector.rs:81-83: self - normal*2.*dot

// matrix.rs:316-337 (Matrix * Tuple), w = 1 for points, 0 for vectors
inline V3 mat_point(const double* m, V3 p) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 1., m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 1.,
          m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 1.};
}
inline V3 mat_vector(const double* m, V3 p) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 0., m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 0.,
          m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 0.};
}
// matrix.rs:77-92 (mul_transposed) on a vector (w = 0)
inline V3 mat_transposed_vector(const double* m, V3 p) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12] * 0., m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13] * 0.,
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] * 0.};
}

struct Ray {  // src/render/ray.rs:8-14
  V3 o, d, dinv;
};
inline Ray make_ray(V3 o, V3 d) { return {o, d, {1. / d.x, 1. / d.y, 1. / d.z}}; }  // ray.rs:27-34
inline Ray transform_ray(const Ray& r, const double* m) { return make_ray(mat_point(m, r.o), mat_vector(m, r.d)); }  // ray.rs:21-23
inline V3 position(const Ray& r, double t) { return r.o + mulf(r.d, t); }  // ray.rs:36-38

struct Inter {  // intersection.rs:137-143
  double t;
  uint32_t prim;
  double u, v;
};

struct Counters {
  uint64_t rays_primary = 0, rays_shadow = 0, rays_reflect = 0, rays_refract = 0;
  uint64_t node_visits = 0, tri_tests = 0, prim_tests = 0, hits = 0, hit_bytes = 0;
  void add(const Counters& o) {
    hits += o.hits; hit_bytes += o.hit_bytes;
    rays_primary += o.rays_primary; rays_shadow += o.rays_shadow; rays_reflect += o.rays_reflect;
    rays_refract += o.rays_refract; node_visits += o.node_visits; tri_tests += o.tri_tests; prim_tests += o.prim_tests;
  }
};

struct Collector {  // intersection.rs:17-23
  std::vector<Inter> vec;
  bool has_hit = false;
  Inter hit{0, 0, 0, 0};
  bool shadow = false;
  double shadow_intensity = 0.;
  bool skip_redundant = true;
  uint32_t next_prim = RT_NONE;
  double hit_time() const { return has_hit ? hit.t : INFINITY; }  // :115-117
};

struct Ctx {
  const rt_scene_desc* s;
  rt_options opt;
  bool work_equivalent;  // re-trace reflect/refract subtrees once per light like integrator.rs:199-228 does
  Counters* c;
};

inline double transparency_of(const Ctx& cx, uint32_t prim) {
  return cx.s->material[(size_t)cx.s->prim_material[prim] * RT_MAT_STRIDE + RT_MAT_TRANSPARENCY];
}

// IntersectionCollector::skip_intersection (intersection.rs:65-89)
bool skip_intersection(const Ctx& cx, Collector& col, double time, const Inter& inter) {
  double dest_time = col.hit_time();
  if (!col.shadow) {
    bool is_positive = sign_positive(time);
    if (is_positive && time < col.hit_time()) {
      col.hit = inter;
      col.has_hit = true;
      return false;
    }
    return is_positive && col.skip_redundant;
  }
  if (sign_positive(time) && !approx_eq(time, dest_time) && time < dest_time && col.shadow_intensity < 1.) {
    col.shadow_intensity += 1. - transparency_of(cx, col.next_prim);
  }
  return col.skip_redundant;
}
void col_push(const Ctx& cx, Collector& col, const Inter& inter) {  // :90-95
  col.next_prim = inter.prim;
  if (!skip_intersection(cx, col, inter.t, inter)) col.vec.push_back(inter);
}
void col_add(const Ctx& cx, Collector& col, double t) {  // :97-102
  col_push(cx, col, Inter{t, col.next_prim, -INFINITY, -INFINITY});
}
void col_add_uv(const Ctx& cx, Collector& col, double t, double u, double v) {  // :104-109
  col_push(cx, col, Inter{t, col.next_prim, u, v});
}

// ---------------------------------------------------------------------------------------------
// Shapes (src/scene/object/primitive/*.rs)
// ---------------------------------------------------------------------------------------------
void sphere_intersect(const Ctx& cx, const Ray& r, Collector& col) {  // sphere.rs:20-35
  V3 sr = r.o - V3{0., 0., 0.};
  double a = dot(r.d, r.d);
  double b = 2. * dot(r.d, sr);
  double c = dot(sr, sr) - 1.;
  double disc = b * b - 4. * a * c;
  if (disc < 0. || a == 0.) return;
  double ds = std::sqrt(disc);
  col_add(cx, col, (-b - ds) / (2. * a));
  col_add(cx, col, (-b + ds) / (2. * a));
}

void plane_intersect(const Ctx& cx, const Ray& r, Collector& col) {  // plane.rs:13-19
  if (approx_eq(r.d.y, 0.)) return;
  col_add(cx, col, -r.o.y / r.d.y);
}

inline void cube_axis(double origin, double dir_inv, double& tmin, double& tmax) {  // cube.rs:51-63
  double tmin_n = -1. - origin;
  double tmax_n = 1. - origin;
  double a = tmin_n * dir_inv;
  double b = tmax_n * dir_inv;
  if (a < b) { tmin = a; tmax = b; } else { tmin = b; tmax = a; }
}
void cube_intersect(const Ctx& cx, const Ray& r, Collector& col) {  // cube.rs:25-42
  double xmin, xmax, ymin, ymax, zmin, zmax;
  cube_axis(r.o.x, r.dinv.x, xmin, xmax);
  cube_axis(r.o.y, r.dinv.y, ymin, ymax);
  cube_axis(r.o.z, r.dinv.z, zmin, zmax);
  double tmin = std::fmax(std::fmax(xmin, ymin), zmin);
  double tmax = std::fmin(std::fmin(xmax, ymax), zmax);
  if (tmin > tmax) return;
  col_add(cx, col, tmin);
  col_add(cx, col, tmax);
}
V3 cube_normal(V3 p) {  // cube.rs:10-23
  double ax = std::fabs(p.x), ay = std::fabs(p.y), az = std::fabs(p.z);
  double m = std::fmax(std::fmax(ax, ay), az);
  if (m == ax) return {p.x, 0., 0.};
  if (m == ay) return {0., p.y, 0.};
  return {0., 0., p.z};
}

void cylinder_intersect(const Ctx& cx, const Ray& r, Collector& col, double y_min, double y_max, bool closed) {  // cylinder.rs:54-94
  if (approx_eq(y_min, y_max)) return;
  // intersect_cyl_caps :107-120
  if (closed && !approx_eq(r.d.y, 0.)) {
    double tmin = (y_min - r.o.y) / r.d.y;
    double tmax = (y_max - r.o.y) / r.d.y;
    auto within = [&](double t) {  // :101-106
      double x = r.o.x + t * r.d.x;
      double z = r.o.z + t * r.d.z;
      return x * x + z * z <= 1.;
    };
    if (within(tmin)) col_add(cx, col, tmin);
    if (within(tmax)) col_add(cx, col, tmax);
  }
  double a = r.d.x * r.d.x + r.d.z * r.d.z;
  if (approx_eq(a, 0.)) return;
  double b = 2. * r.o.x * r.d.x + 2. * r.o.z * r.d.z;
  double c = r.o.x * r.o.x + r.o.z * r.o.z - 1.;
  double disc = b * b - 4. * a * c;
  if (disc < 0.) return;
  double ds = std::sqrt(disc);
  double t0 = (-b - ds) / (2. * a);
  double t1 = (-b + ds) / (2. * a);
  double y0 = r.o.y + t0 * r.d.y;
  if (y_min < y0 && y0 < y_max) col_add(cx, col, t0);
  double y1 = r.o.y + t1 * r.d.y;
  if (y_min < y1 && y1 < y_max) col_add(cx, col, t1);
}
V3 cylinder_normal(V3 p, double y_min, double y_max) {  // cylinder.rs:42-53
  double dist = p.x * p.x + p.z * p.z;
  bool within = dist < 1.;
  if (within && p.y >= y_max - EPSILON) return {0., 1., 0.};
  if (within && p.y <= y_min + EPSILON) return {0., -1., 0.};
  return {p.x, 0., p.z};
}

void cone_intersect(const Ctx& cx, const Ray& r, Collector& col, double y_min, double y_max, bool closed) {  // cone.rs:56-95
  if (approx_eq(y_min, y_max)) return;
  // intersect_cone_caps :114-127 (radius test compares a squared distance with |y|, sic)
  if (closed && !approx_eq(r.d.y, 0.)) {
    double tmin = (y_min - r.o.y) / r.d.y;
    double tmax = (y_max - r.o.y) / r.d.y;
    auto within = [&](double t, double radius) {  // :106-112
      double x = r.o.x + t * r.d.x;
      double z = r.o.z + t * r.d.z;
      return x * x + z * z <= radius;
    };
    if (within(tmin, std::fabs(y_min))) col_add(cx, col, tmin);
    if (within(tmax, std::fabs(y_max))) col_add(cx, col, tmax);
  }
  V3 d = r.d, o = r.o;
  double a = d.x * d.x - d.y * d.y + d.z * d.z;
  double b = 2. * (o.x * d.x - o.y * d.y + o.z * d.z);
  double c = o.x * o.x - o.y * o.y + o.z * o.z;
  if (approx_eq(a, 0.)) {
    if (approx_eq(b, 0.)) return;
    col_add(cx, col, -c / (2. * b));  // no y-range check, as in the reference
  } else {
    double disc = b * b - 4. * a * c;
    if (disc < 0.) return;
    double ds = std::sqrt(disc);
    double t0 = (-b - ds) / (2. * a);
    double t1 = (-b + ds) / (2. * a);
    double y0 = o.y + t0 * d.y;
    if (y_min < y0 && y0 < y_max) col_add(cx, col, t0);
    double y1 = o.y + t1 * d.y;
    if (y_min < y1 && y1 < y_max) col_add(cx, col, t1);
  }
}
V3 cone_normal(V3 p, double y_min, double y_max) {  // cone.rs:38-55
  double dist = p.x * p.x + p.z * p.z;
  double min_radius = std::fabs(y_min), max_radius = std::fabs(y_max);
  if (dist < max_radius && p.y >= y_max - EPSILON) return {0., 1., 0.};
  if (dist < min_radius && p.y <= y_min + EPSILON) return {0., -1., 0.};
  double y = std::sqrt(p.x * p.x + p.z * p.z);
  if (p.y > 0.) y = -y;
  return {p.x, y, p.z};
}

inline V3 ld3(const double* a, size_t i) { return {a[3 * i], a[3 * i + 1], a[3 * i + 2]}; }

void triangle_intersect(const Ctx& cx, const Ray& r, Collector& col, uint32_t tri, bool smooth) {  // triangle.rs:48-71, smooth_triangle.rs:60-83
  const rt_scene_desc* s = cx.s;
  V3 p1 = ld3(s->tri_p1, tri), e1 = ld3(s->tri_e1, tri), e2 = ld3(s->tri_e2, tri);
  V3 dir_cross_e2 = cross(r.d, e2);
  double det = dot(e1, dir_cross_e2);
  if (approx_eq(det, 0.)) return;
  double f = 1. / det;
  V3 p1_to_origin = r.o - p1;
  double u = f * dot(p1_to_origin, dir_cross_e2);
  if (!(u >= 0. && u <= 1.)) return;
  V3 origin_cross_e1 = cross(p1_to_origin, e1);
  double v = f * dot(r.d, origin_cross_e1);
  if (v < 0. || u + v > 1.) return;
  double t = f * dot(e2, origin_cross_e1);
  if (smooth) col_add_uv(cx, col, t, u, v);
  else col_add(cx, col, t);
}

// PrimitiveObject::intersect (object.rs:432-439) + Shape::local_intersect (shape.rs:51-62)
void prim_intersect(const Ctx& cx, uint32_t prim, const Ray& scene_ray, Collector& col) {
  const rt_scene_desc* s = cx.s;
  col.next_prim = prim;  // Object::intersect: collector.set_next_object(self) (object.rs:196)
  uint8_t shape = s->prim_shape[prim];
  if (shape == RT_SHAPE_TRIANGLE || shape == RT_SHAPE_SMOOTH_TRIANGLE) cx.c->tri_tests++;
  else cx.c->prim_tests++;
  Ray local = scene_ray;
  uint32_t xf = s->prim_xform[prim];
  if (xf != RT_NONE) local = transform_ray(scene_ray, s->xform_inv + (size_t)xf * 16);
  switch (shape) {
    case RT_SHAPE_SPHERE: sphere_intersect(cx, local, col); break;
    case RT_SHAPE_PLANE: plane_intersect(cx, local, col); break;
    case RT_SHAPE_CUBE:
    case RT_SHAPE_BBOX: cube_intersect(cx, local, col); break;
    case RT_SHAPE_CYLINDER: {
      uint32_t q = s->prim_geom[prim];
      cylinder_intersect(cx, local, col, s->quadric_y_min[q], s->quadric_y_max[q], s->quadric_closed[q] != 0);
      break;
    }
    case RT_SHAPE_CONE: {
      uint32_t q = s->prim_geom[prim];
      cone_intersect(cx, local, col, s->quadric_y_min[q], s->quadric_y_max[q], s->quadric_closed[q] != 0);
      break;
    }
    case RT_SHAPE_TRIANGLE: triangle_intersect(cx, local, col, s->prim_geom[prim], false); break;
    case RT_SHAPE_SMOOTH_TRIANGLE: triangle_intersect(cx, local, col, s->prim_geom[prim], true); break;
  }
}

// BoundingBox::intersection_time (bounding_box.rs:192-225) + axis_intersection_times (:141-153)
inline void axis_times(double origin, double dir_inv, double mn, double mx, double& tmin, double& tmax) {
  double a = (mn - origin) * dir_inv;
  double b = (mx - origin) * dir_inv;
  if (a < b) { tmin = a; tmax = b; } else { tmin = b; tmax = a; }
}
bool bbox_intersection_time(const double* bb, const Ray& r, double& out) {
  double xmin, xmax, ymin, ymax, zmin, zmax;
  axis_times(r.o.x, r.dinv.x, bb[0], bb[3], xmin, xmax);
  axis_times(r.o.y, r.dinv.y, bb[1], bb[4], ymin, ymax);
  double tmin = std::fmax(xmin, ymin);
  double tmax = std::fmin(xmax, ymax);
  if (tmin > tmax) return false;
  axis_times(r.o.z, r.dinv.z, bb[2], bb[5], zmin, zmax);
  tmin = std::fmax(tmin, zmin);
  tmax = std::fmin(tmax, zmax);
  if (tmin > tmax) return false;
  out = tmin;
  return true;
}

void ref_intersect(const Ctx& cx, uint32_t ref, const Ray& ray, Collector& col);

// ObjectGroup::intersect_iter (group.rs:171-201)
void group_intersect(const Ctx& cx, uint32_t root, const Ray& ray, Collector& col) {
  const rt_scene_desc* s = cx.s;
  std::vector<std::pair<uint32_t, double>> stack;
  double t;
  cx.c->node_visits++;
  if (!bbox_intersection_time(s->node_bbox + (size_t)root * 6, ray, t)) return;
  stack.emplace_back(root, t);
  while (!stack.empty()) {
    auto [node, time] = stack.back();
    stack.pop_back();
    if (time >= col.hit_time()) continue;
    uint32_t b = s->node_child_begin[node], e = s->node_child_begin[node + 1];
    for (uint32_t i = e; i-- > b;) {  // children.iter().rev()
      uint32_t ref = s->child_ref[i];
      if ((ref >> RT_REF_KIND_SHIFT) == RT_REF_NODE) {
        uint32_t child = ref & RT_REF_INDEX_MASK;
        double ct;
        cx.c->node_visits++;
        if (bbox_intersection_time(s->node_bbox + (size_t)child * 6, ray, ct)) stack.emplace_back(child, ct);
      } else {
        ref_intersect(cx, ref, ray, col);
      }
    }
  }
}

// CsgObject::intersect (csg.rs:112-122), filter_intersections (:87-111), is_intersection_allowed (:50-65)
void csg_intersect(const Ctx& cx, uint32_t csg, const Ray& ray, Collector& col) {
  const rt_scene_desc* s = cx.s;
  Collector temp;
  temp.skip_redundant = false;  // IntersectionCollector::new(None, false)
  ref_intersect(cx, s->csg_left[csg], ray, temp);
  ref_intersect(cx, s->csg_right[csg], ray, temp);
  // collect_sorted (intersection.rs:110-114); std::stable_sort == Rust's insertion sort for short slices
  std::stable_sort(temp.vec.begin(), temp.vec.end(), [](const Inter& a, const Inter& b) { return a.t < b.t; });
  bool inside_left = false, inside_right = false;
  uint8_t op = s->csg_op[csg];
  uint8_t depth = s->csg_depth[csg];
  for (const Inter& i : temp.vec) {
    bool hit_left = (s->prim_csg_left[i.prim] >> depth) & 1u;  // self.left.includes(i.object())
    if (hit_left) inside_left = !inside_left;
    else inside_right = !inside_right;
    bool allowed;
    if (op == RT_CSG_UNION) allowed = hit_left ? !inside_right : !inside_left;
    else if (op == RT_CSG_INTERSECTION) allowed = hit_left ? inside_right : inside_left;
    else allowed = hit_left ? !inside_right : inside_left;
    if (allowed) col_push(cx, col, i);
  }
}

// Object::intersect (object.rs:193-202)
void ref_intersect(const Ctx& cx, uint32_t ref, const Ray& ray, Collector& col) {
  uint32_t kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
  if (kind == RT_REF_PRIM) prim_intersect(cx, idx, ray, col);
  else if (kind == RT_REF_NODE) group_intersect(cx, idx, ray, col);
  else csg_intersect(cx, idx, ray, col);
}

// PrimitiveObject::normal_vector_at_with_intersection (object.rs:415-430) + Shape::local_normal_at (shape.rs:35-50)
V3 local_normal_at(const Ctx& cx, uint32_t prim, V3 p, double u, double v) {
  const rt_scene_desc* s = cx.s;
  switch (s->prim_shape[prim]) {
    case RT_SHAPE_SPHERE: return p - V3{0., 0., 0.};  // sphere.rs:10-12
    case RT_SHAPE_PLANE: return {0., 1., 0.};         // plane.rs:10-12
    case RT_SHAPE_CUBE:
    case RT_SHAPE_BBOX: return cube_normal(p);
    case RT_SHAPE_CYLINDER: {
      uint32_t q = s->prim_geom[prim];
      return cylinder_normal(p, s->quadric_y_min[q], s->quadric_y_max[q]);
    }
    case RT_SHAPE_CONE: {
      uint32_t q = s->prim_geom[prim];
      return cone_normal(p, s->quadric_y_min[q], s->quadric_y_max[q]);
    }
    case RT_SHAPE_TRIANGLE: return ld3(s->tri_n1, s->prim_geom[prim]);  // triangle.rs:101-103
    default: {                                                          // smooth_triangle.rs:56-59
      uint32_t t = s->prim_geom[prim];
      V3 n1 = ld3(s->tri_n1, t), n2 = ld3(s->tri_n2, t), n3 = ld3(s->tri_n3, t);
      return mulf(n2, u) + mulf(n3, v) + mulf(n1, 1. - u - v);
    }
  }
}
V3 normal_at(const Ctx& cx, uint32_t prim, V3 scene_point, double u, double v) {
  const rt_scene_desc* s = cx.s;
  uint32_t xf = s->prim_xform[prim];
  V3 n;
  if (xf == RT_NONE) {
    n = local_normal_at(cx, prim, scene_point, u, v);
  } else {
    const double* inv = s->xform_inv + (size_t)xf * 16;
    V3 object_point = mat_point(inv, scene_point);
    V3 object_normal = local_normal_at(cx, prim, object_point, u, v);
    n = mat_transposed_vector(inv, object_normal);
  }
  return normalize(n);
}

struct Comps {  // IntersecComputations (intersection.rs:201-214)
  double t;
  uint32_t prim;
  V3 point, under, over, eye, normal, reflect_v;
  bool inside;
  double n1, n2;
};

// IntersecComputations::new (intersection.rs:217-256)
Comps make_comps(const Ctx& cx, const Inter& i, const Ray& ray, double n1, double n2) {
  Comps c;
  c.t = i.t;
  c.prim = i.prim;
  c.point = position(ray, i.t);
  c.eye = neg(ray.d);
  V3 n = normal_at(cx, i.prim, c.point, i.u, i.v);
  c.inside = dot(n, c.eye) < 0.;
  if (c.inside) n = neg(n);
  c.normal = n;
  V3 off = mulf(n, EPSILON);
  c.over = c.point + off;
  c.under = c.point - off;
  c.reflect_v = reflect(ray.d, n);
  c.n1 = n1;
  c.n2 = n2;
  return c;
}

inline const double* material_of(const Ctx& cx, uint32_t prim) {
  return cx.s->material + (size_t)cx.s->prim_material[prim] * RT_MAT_STRIDE;
}

// IntersecComputations::get_refractive (intersection.rs:258-302) on the collector's vec + hit
void get_refractive(const Ctx& cx, std::vector<Inter>& vec, const Inter& hit, double& n1, double& n2) {
  n1 = 1.0;  // AIR_REFRACTIVE_INDEX material.rs:7
  n2 = 1.0;
  if (approx_eq(material_of(cx, hit.prim)[RT_MAT_TRANSPARENCY], 0.)) return;  // compute_refractive :184-190
  // IntersectionCollection::new sorts when the hit is transparent (:398-415)
  std::stable_sort(vec.begin(), vec.end(), [](const Inter& a, const Inter& b) { return a.t < b.t; });
  std::vector<uint32_t> containers;
  for (const Inter& inter : vec) {
    bool is_hit = inter.t == hit.t && inter.prim == hit.prim;  // PartialEq :149-153
    if (is_hit && !containers.empty()) n1 = material_of(cx, containers.back())[RT_MAT_REFRACTIVE_INDEX];
    auto it = std::find(containers.begin(), containers.end(), inter.prim);
    if (it != containers.end()) containers.erase(it);
    else containers.push_back(inter.prim);
    if (is_hit) {
      if (!containers.empty()) n2 = material_of(cx, containers.back())[RT_MAT_REFRACTIVE_INDEX];
      break;
    }
  }
}

// Pattern::color_at_object (pattern.rs:113-128) + color_at (:78-111)
V3 pattern_color_at_object(const Ctx& cx, uint32_t prim, V3 point) {
  const rt_scene_desc* s = cx.s;
  uint32_t m = s->prim_material[prim];
  uint8_t kind = s->pattern_kind[m];
  V3 c1 = ld3(s->pattern_c1, m), c2 = ld3(s->pattern_c2, m);
  if (kind == RT_PATTERN_CONST) return c1;
  V3 object_point = point;
  uint32_t xf = s->prim_xform[prim];
  static const double IDENT[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  // object.transformation_inverse() is identity when None (object.rs:409-411); the product is still formed
  object_point = mat_point(xf == RT_NONE ? IDENT : s->xform_inv + (size_t)xf * 16, point);
  V3 p = mat_point(s->pattern_inv + (size_t)m * 16, object_point);
  switch (kind) {
    case RT_PATTERN_STRIPE: return approx_eq(std::fmod(std::floor(p.x), 2.), 0.) ? c1 : c2;
    case RT_PATTERN_GRADIENT: return c1 + cmulf(c2, p.x - std::floor(p.x));
    case RT_PATTERN_RING: {
      double val = std::floor(std::sqrt(p.x * p.x + p.z * p.z));
      return approx_eq(std::fmod(val, 2.), 0.) ? c1 : c2;
    }
    case RT_PATTERN_CHECKERS: {
      double x = std::floor(p.x), y = std::floor(p.y), z = std::floor(p.z);
      return approx_eq(std::fmod(x + y + z, 2.), 0.) ? c1 : c2;
    }
    default: return {p.x, p.y, p.z};  // TestPattern
  }
}

// PointLightSource::color_of_illuminated_point (point_light.rs:38-76)
V3 lighting(const Ctx& cx, uint32_t light, uint32_t prim, V3 point, V3 eye, V3 normal, double shadow) {
  const rt_scene_desc* s = cx.s;
  const double* mat = material_of(cx, prim);
  V3 intensity = ld3(s->light_intensity, light);
  V3 effective = cmul(pattern_color_at_object(cx, prim, point), intensity);
  V3 light_v = normalize(ld3(s->light_pos, light) - point);
  V3 ambient = cmulf(effective, mat[RT_MAT_AMBIENT]);
  double ldn = dot(light_v, normal);
  V3 black{0., 0., 0.};
  if (approx_eq(shadow, 1.) || ldn < 0.) return ambient + black + black;
  V3 diffuse = cmulf(cmulf(effective, mat[RT_MAT_DIFFUSE]), ldn);
  V3 reflect_v = reflect(neg(light_v), normal);
  double rde = dot(reflect_v, eye);
  V3 specular = black;
  if (sign_positive(rde)) {
    double factor = std::pow(rde, mat[RT_MAT_SHININESS]);
    specular = cmulf(cmulf(intensity, mat[RT_MAT_SPECULAR]), factor);
  }
  return ambient + cmulf(diffuse + specular, 1. - shadow);
}

// schlick_reflectance (light.rs:5-27)
double schlick(V3 eye, V3 normal, double n1, double n2) {
  double cosv = dot(eye, normal);
  if (n1 > n2) {
    double ratio = n1 / n2;
    double sin2_t = (ratio * ratio) * (1. - cosv * cosv);
    if (sin2_t > 1.) return 1.;
    cosv = std::sqrt(1. - sin2_t);
  }
  double q = (n1 - n2) / (n1 + n2);
  double r0 = q * q;
  double x = 1. - cosv;
  double x2 = x * x;
  double x5 = x * (x2 * x2);  // powi(5) as LLVM/compiler-rt expand it
  return r0 + (1. - r0) * x5;
}

V3 color_at_depth(const Ctx& cx, const Ray& ray, uint32_t depth);

// Integrator::point_shadow_intensity_comps (integrator.rs:136-152) + get_point_shadow_dist_ray (:110-123)
double shadow_intensity_at(const Ctx& cx, uint32_t light, V3 point, uint32_t dest_prim) {
  V3 v = ld3(cx.s->light_pos, light) - point;
  double dist = magnitude(v);
  Ray ray = make_ray(point, normalize(v));
  Collector col;  // with_dest_obj_shadow_intensity (intersection.rs:49-57)
  col.has_hit = true;
  col.hit = Inter{dist, dest_prim, -INFINITY, -INFINITY};
  col.shadow = true;
  col.shadow_intensity = 0.;
  col.skip_redundant = true;
  cx.c->rays_shadow++;
  group_intersect(cx, 0, ray, col);
  double intensity = std::fmin(col.shadow_intensity, 1.);  // :124-126
  if (!cx.opt.shadow_intensity) return approx_eq(intensity, 0.) ? 0. : 1.;
  return intensity;
}

V3 reflected_color(const Ctx& cx, const Comps& h, uint32_t depth) {  // integrator.rs:154-168
  const double* mat = material_of(cx, h.prim);
  if (depth >= cx.opt.max_depth || approx_eq(mat[RT_MAT_REFLECTIVITY], 0.)) return {0., 0., 0.};
  cx.c->rays_reflect++;
  V3 c = color_at_depth(cx, make_ray(h.over, h.reflect_v), depth + 1);
  return cmulf(c, mat[RT_MAT_REFLECTIVITY]);
}

V3 refracted_color(const Ctx& cx, const Comps& h, uint32_t depth) {  // integrator.rs:170-197
  const double* mat = material_of(cx, h.prim);
  if (depth >= cx.opt.max_depth || approx_eq(mat[RT_MAT_TRANSPARENCY], 0.)) return {0., 0., 0.};
  double ratio = h.n1 / h.n2;
  double cos_i = dot(h.eye, h.normal);
  double sin2_t = (ratio * ratio) * (1. - cos_i * cos_i);
  if (sin2_t > 1.) return {0., 0., 0.};
  double cos_t = std::sqrt(1. - sin2_t);
  V3 dir = mulf(h.normal, ratio * cos_i - cos_t) - mulf(h.eye, ratio);
  cx.c->rays_refract++;
  V3 c = color_at_depth(cx, make_ray(h.under, dir), depth + 1);
  return cmulf(c, mat[RT_MAT_TRANSPARENCY]);
}

// Integrator::shade_hit_at_depth (integrator.rs:199-228)
V3 shade_hit(const Ctx& cx, const Comps& h, uint32_t depth) {
  const double* mat = material_of(cx, h.prim);
  V3 acc{0., 0., 0.};
  bool have_rr = false;
  V3 reflected{0, 0, 0}, refracted{0, 0, 0};
  for (uint32_t l = 0; l < cx.s->n_lights; l++) {
    V3 surface = lighting(cx, l, h.prim, h.over, h.eye, h.normal, shadow_intensity_at(cx, l, h.over, h.prim));
    if (cx.work_equivalent || !have_rr) {  // the subtrees are pure functions of `h`: identical for every light
      reflected = reflected_color(cx, h, depth);
      refracted = refracted_color(cx, h, depth);
      have_rr = true;
    }
    bool use_schlick = mat[RT_MAT_REFLECTIVITY] > 0. && mat[RT_MAT_TRANSPARENCY] > 0. &&
                       !approx_eq(mat[RT_MAT_REFLECTIVITY], 0.) && !approx_eq(mat[RT_MAT_TRANSPARENCY], 0.);
    V3 rr;
    if (use_schlick) {
      double reflectance = schlick(h.eye, h.normal, h.n1, h.n2);
      rr = cmulf(reflected, reflectance) + cmulf(refracted, 1. - reflectance);
    } else {
      rr = reflected + refracted;
    }
    acc = acc + surface + rr;
  }
  return acc;
}

bool hit_comps(const Ctx& cx, const Ray& ray, Comps& out) {
  Collector col;  // IntersectionCollection::from_group (intersection.rs:439-444)
  group_intersect(cx, 0, ray, col);
  if (!col.has_hit) return false;
  double n1, n2;
  get_refractive(cx, col.vec, col.hit, n1, n2);
  out = make_comps(cx, col.hit, ray, n1, n2);
  {  // algorithmic bytes of shading this hit
    const rt_scene_desc* s = cx.s;
    uint32_t p = col.hit.prim;
    uint64_t b = 64u + 48u * s->n_lights;
    uint8_t sh = s->prim_shape[p];
    b += sh == RT_SHAPE_SMOOTH_TRIANGLE ? 72u : (sh == RT_SHAPE_TRIANGLE ? 24u : (s->prim_xform[p] != RT_NONE ? 96u : 0u));
    if (s->pattern_kind[s->prim_material[p]] != RT_PATTERN_CONST) b += 144u;
    cx.c->hits++;
    cx.c->hit_bytes += b;
  }
  return true;
}

V3 color_at_depth(const Ctx& cx, const Ray& ray, uint32_t depth) {  // integrator.rs:59-66
  Comps h;
  if (!hit_comps(cx, ray, h)) return {0., 0., 0.};
  return shade_hit(cx, h, depth);
}

Ray ray_for_pixel(const rt_camera& cam, double x, double y) {  // camera.rs:189-201
  double xo = (x + 0.5) * cam.pixel_size;
  double yo = (y + 0.5) * cam.pixel_size;
  double sx = cam.half_width - xo;
  double sy = cam.half_height - yo;
  V3 pixel = mat_point(cam.inv_view, {sx, sy, -1.});
  V3 origin = mat_point(cam.inv_view, {0., 0., 0.});
  return make_ray(origin, normalize(pixel - origin));
}

V3 color_at_pixel(const Ctx& cx, const rt_camera& cam, uint32_t px, uint32_t py) {  // renderer.rs:50-66
  double x = (double)px, y = (double)py;
  V3 color{0., 0., 0.};
  for (uint32_t i = 0; i < cx.opt.n_offsets; i++)
    for (uint32_t j = 0; j < cx.opt.n_offsets; j++) {
      cx.c->rays_primary++;
      color = color + color_at_depth(cx, ray_for_pixel(cam, x + cx.opt.offsets[i], y + cx.opt.offsets[j]), 0);
    }
  uint32_t n = cx.opt.n_offsets;
  return cdivf(color, (double)(n * n));
}

}  // namespace

extern "C" {

// Renders rows [y0, y1) x columns [x0, x1) of the image into out_rgb (full-image row-major indexing),
// on n_threads host threads (chunks of rows handed out dynamically, like rayon's par_iter_mut, image.rs:73-81).
// work_equivalent != 0 re-traces reflect/refract subtrees once per light exactly as the reference does.
int rto_render_region(const rt_scene_desc* scene, const rt_camera* cam, const rt_options* opt, uint32_t x0, uint32_t y0,
                      uint32_t x1, uint32_t y1, int n_threads, int work_equivalent, double* out_rgb, rt_stats* stats) {
  if (!scene || !cam || !opt || !out_rgb || opt->n_offsets < 1 || opt->n_offsets > 5) return RT_ERR_INVALID_ARGUMENT;
  if (x1 > cam->width || y1 > cam->height || x0 > x1 || y0 > y1) return RT_ERR_INVALID_ARGUMENT;
  if (n_threads < 1) n_threads = 1;
  std::atomic<uint32_t> next_row{y0};
  std::vector<Counters> counters(n_threads);
  auto worker = [&](int tid) {
    Ctx cx{scene, *opt, work_equivalent != 0, &counters[tid]};
    while (true) {
      uint32_t y = next_row.fetch_add(1);
      if (y >= y1) break;
      for (uint32_t x = x0; x < x1; x++) {
        V3 c = color_at_pixel(cx, *cam, x, y);
        double* o = out_rgb + ((size_t)y * cam->width + x) * 3;
        o[0] = c.x;
        o[1] = c.y;
        o[2] = c.z;
      }
    }
  };
  std::vector<std::thread> threads;
  for (int t = 1; t < n_threads; t++) threads.emplace_back(worker, t);
  worker(0);
  for (auto& t : threads) t.join();
  if (stats) {
    Counters total;
    for (auto& c : counters) total.add(c);
    std::memset(stats, 0, sizeof(*stats));
    stats->rays_primary = total.rays_primary;
    stats->rays_shadow = total.rays_shadow;
    stats->rays_reflect = total.rays_reflect;
    stats->rays_refract = total.rays_refract;
    stats->node_visits = total.node_visits;
    stats->tri_tests = total.tri_tests;
    stats->prim_tests = total.prim_tests;
    stats->hits = total.hits;
    stats->hit_bytes = total.hit_bytes;
  }
  return RT_OK;
}

int rto_render(const rt_scene_desc* scene, const rt_camera* cam, const rt_options* opt, int n_threads, int work_equivalent,
               double* out_rgb, rt_stats* stats) {
  if (!cam) return RT_ERR_INVALID_ARGUMENT;
  return rto_render_region(scene, cam, opt, 0, 0, cam->width, cam->height, n_threads, work_equivalent, out_rgb, stats);
}

// ---- hooks used by the known-answer tests (the reference's #[cfg(test)] helpers) -------------------

// Integrator::color_at (integrator.rs:68)
void rto_color_at(const rt_scene_desc* scene, const rt_options* opt, const double* origin, const double* dir, double* out_rgb) {
  Counters c;
  Ctx cx{scene, *opt, false, &c};
  V3 col = color_at_depth(cx, make_ray({origin[0], origin[1], origin[2]}, {dir[0], dir[1], dir[2]}), 0);
  out_rgb[0] = col.x; out_rgb[1] = col.y; out_rgb[2] = col.z;
}

// Integrator::intersect_testing (integrator.rs:256-260): keep-all collector, sorted. Returns the count.
// direct_ref == RT_NONE intersects the scene root group; otherwise that object alone, without the root's
// bounding-box test (Object::intersect_to_collection, object.rs:204-208).
int rto_intersect_all(const rt_scene_desc* scene, const double* origin, const double* dir, uint32_t direct_ref, int max_out,
                      double* out_t, uint32_t* out_prim) {
  Counters c;
  rt_options opt{};
  Ctx cx{scene, opt, false, &c};
  Collector col;
  col.skip_redundant = false;
  Ray ray = make_ray({origin[0], origin[1], origin[2]}, {dir[0], dir[1], dir[2]});
  if (direct_ref == RT_NONE) group_intersect(cx, 0, ray, col);
  else ref_intersect(cx, direct_ref, ray, col);
  std::stable_sort(col.vec.begin(), col.vec.end(), [](const Inter& a, const Inter& b) { return a.t < b.t; });
  int n = 0;
  for (const Inter& i : col.vec) {
    if (n < max_out) {
      out_t[n] = i.t;
      out_prim[n] = i.prim;
    }
    n++;
  }
  return n;
}

// IntersectionCollection::from_group(...).hit_computations() (intersection.rs:439-444,476-478).
// out[0..25): t, prim, inside, n1, n2, point(3), over(3), under(3), eye(3), normal(3), reflect(3). Returns 1 on hit.
int rto_hit_computations(const rt_scene_desc* scene, const double* origin, const double* dir, double* out) {
  Counters c;
  rt_options opt{};
  Ctx cx{scene, opt, false, &c};
  Comps h;
  if (!hit_comps(cx, make_ray({origin[0], origin[1], origin[2]}, {dir[0], dir[1], dir[2]}), h)) return 0;
  double vals[23] = {h.t, (double)h.prim, h.inside ? 1. : 0., h.n1, h.n2, h.point.x, h.point.y, h.point.z,
                     h.over.x, h.over.y, h.over.z, h.under.x, h.under.y, h.under.z, h.eye.x, h.eye.y, h.eye.z,
                     h.normal.x, h.normal.y, h.normal.z, h.reflect_v.x, h.reflect_v.y, h.reflect_v.z};
  std::memcpy(out, vals, sizeof(vals));
  return 1;
}

// Integrator::point_shadow_intensity_comps via over_point (integrator.rs:136-152)
double rto_shadow_intensity(const rt_scene_desc* scene, const rt_options* opt, uint32_t light, const double* point) {
  Counters c;
  Ctx cx{scene, *opt, false, &c};
  return shadow_intensity_at(cx, light, {point[0], point[1], point[2]}, RT_NONE);
}

// PointLightSource::color_of_illuminated_point (point_light.rs:38-76)
void rto_lighting(const rt_scene_desc* scene, uint32_t light, uint32_t prim, const double* point, const double* eye,
                  const double* normal, double shadow, double* out_rgb) {
  Counters c;
  rt_options opt{};
  Ctx cx{scene, opt, false, &c};
  V3 col = lighting(cx, light, prim, {point[0], point[1], point[2]}, {eye[0], eye[1], eye[2]},
                    {normal[0], normal[1], normal[2]}, shadow);
  out_rgb[0] = col.x; out_rgb[1] = col.y; out_rgb[2] = col.z;
}

double rto_schlick(const double* eye, const double* normal, double n1, double n2) {
  return schlick({eye[0], eye[1], eye[2]}, {normal[0], normal[1], normal[2]}, n1, n2);
}

// Object::normal_vector_at (object.rs:164-166)
void rto_normal_at(const rt_scene_desc* scene, uint32_t prim, const double* point, double u, double v, double* out) {
  Counters c;
  rt_options opt{};
  Ctx cx{scene, opt, false, &c};
  V3 n = normal_at(cx, prim, {point[0], point[1], point[2]}, u, v);
  out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

// Pattern::color_at_object (pattern.rs:113-128)
void rto_pattern_at(const rt_scene_desc* scene, uint32_t prim, const double* point, double* out) {
  Counters c;
  rt_options opt{};
  Ctx cx{scene, opt, false, &c};
  V3 col = pattern_color_at_object(cx, prim, {point[0], point[1], point[2]});
  out[0] = col.x; out[1] = col.y; out[2] = col.z;
}

// Camera::ray_for_pixel (camera.rs:189-201): out = origin(3), direction(3)
void rto_ray_for_pixel(const rt_camera* cam, double x, double y, double* out6) {
  Ray r = ray_for_pixel(*cam, x, y);
  out6[0] = r.o.x; out6[1] = r.o.y; out6[2] = r.o.z; out6[3] = r.d.x; out6[4] = r.d.y; out6[5] = r.d.z;
}

// Color::as_scaled_values (color.rs:55-66): round-half-away(clamp(v,0,1)*255) as u8; NaN -> 0
void rto_quantize(const double* rgb, uint64_t n, uint8_t* out) {
  for (uint64_t i = 0; i < n; i++) {
    double v = rgb[i];
    if (v != v) { out[i] = 0; continue; }
    v = v < 0. ? 0. : (v > 1. ? 1. : v);
    out[i] = (uint8_t)std::round(v * 255.);
  }
}

}  // extern "C"
