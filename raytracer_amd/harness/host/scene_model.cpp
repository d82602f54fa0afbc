// scene_model.cpp — see scene_model.hpp.  Every function cites the reference lines it restates.
#include "scene_model.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <unordered_map>

#include "../../../include/rtcore.h"

namespace rts {

// ------------------------------------------------------------------------------------------------
// Matrix  (src/math/matrix.rs)
// ------------------------------------------------------------------------------------------------
Mat4 mat_identity() {
  Mat4 m;
  for (int i = 0; i < 16; i++) m.d[i] = (i % 5 == 0) ? 1. : 0.;
  return m;
}

Mat4 mat_mul(const Mat4& a, const Mat4& b) {  // matrix.rs:288-302
  Mat4 o;
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++)
      o.at(r, c) = a.at(r, 0) * b.at(0, c) + a.at(r, 1) * b.at(1, c) + a.at(r, 2) * b.at(2, c) + a.at(r, 3) * b.at(3, c);
  return o;
}

bool mat_inverse(const Mat4& m, Mat4& out) {  // matrix.rs:93-144 (Gauss-Jordan, first non-~0 pivot)
  Mat4 res = mat_identity();
  Mat4 copy = m;
  int row_of_one_for_col[4] = {0, 1, 2, 3};
  for (int i = 0; i < 4; i++) {
    for (int row = i; row < 4; row++) {
      if (!approx_eq(copy.at(row, i), 0.)) {
        row_of_one_for_col[i] = row;
        break;
      }
    }
    int actual_row = row_of_one_for_col[i];
    if (i != actual_row) {
      for (int col = 0; col < 4; col++) {
        std::swap(copy.at(actual_row, col), copy.at(i, col));
        std::swap(res.at(actual_row, col), res.at(i, col));
      }
    }
    double factor_to_1 = copy.at(i, i);
    if (approx_eq(factor_to_1, 0.)) return false;
    for (int col = 0; col < 4; col++) {
      copy.at(i, col) /= factor_to_1;
      res.at(i, col) /= factor_to_1;
    }
    for (int inner_row = 0; inner_row < 4; inner_row++) {
      double row_factor = copy.at(inner_row, i);
      for (int inner_col = 0; inner_col < 4; inner_col++) {
        if (i == inner_row) continue;
        copy.at(inner_row, inner_col) -= copy.at(i, inner_col) * row_factor;
        res.at(inner_row, inner_col) -= res.at(i, inner_col) * row_factor;
      }
    }
  }
  out = res;
  return true;
}

bool mat_approx_eq(const Mat4& a, const Mat4& b) {
  for (int i = 0; i < 16; i++)
    if (!approx_eq(a.d[i], b.d[i])) return false;
  return true;
}

V3 mat_mul_point(const Mat4& m, V3 p) {  // matrix.rs:316-337, w() == 1 (point.rs:33)
  return {m.d[0] * p.x + m.d[1] * p.y + m.d[2] * p.z + m.d[3] * 1., m.d[4] * p.x + m.d[5] * p.y + m.d[6] * p.z + m.d[7] * 1.,
          m.d[8] * p.x + m.d[9] * p.y + m.d[10] * p.z + m.d[11] * 1.};
}
V3 mat_mul_vector(const Mat4& m, V3 p) {  // matrix.rs:316-337, w() == 0 (vector.rs:33)
  return {m.d[0] * p.x + m.d[1] * p.y + m.d[2] * p.z + m.d[3] * 0., m.d[4] * p.x + m.d[5] * p.y + m.d[6] * p.z + m.d[7] * 0.,
          m.d[8] * p.x + m.d[9] * p.y + m.d[10] * p.z + m.d[11] * 0.};
}

Mat4 mat_translation(double x, double y, double z) {  // matrix.rs:148-155
  Mat4 m = mat_identity();
  m.d[3] = x;
  m.d[7] = y;
  m.d[11] = z;
  return m;
}
Mat4 mat_scaling(double x, double y, double z) {  // matrix.rs:158-169 (0 -> EPSILON)
  if (approx_eq(x, 0.)) x = EPSILON;
  if (approx_eq(y, 0.)) y = EPSILON;
  if (approx_eq(z, 0.)) z = EPSILON;
  Mat4 m = mat_identity();
  m.d[0] = x;
  m.d[5] = y;
  m.d[10] = z;
  return m;
}
Mat4 mat_rotation(int axis, double r) {  // matrix.rs:177-210
  double s = std::sin(r), c = std::cos(r);
  Mat4 m = mat_identity();
  if (axis == 0) {
    m.d[5] = c; m.d[6] = -s; m.d[9] = s; m.d[10] = c;
  } else if (axis == 1) {
    m.d[0] = c; m.d[2] = s; m.d[8] = -s; m.d[10] = c;
  } else {
    m.d[0] = c; m.d[1] = -s; m.d[4] = s; m.d[5] = c;
  }
  return m;
}
Mat4 mat_shearing(double xy, double xz, double yx, double yz, double zx, double zy) {  // matrix.rs:213-227
  Mat4 m = mat_identity();
  m.d[1] = xy; m.d[2] = xz; m.d[4] = yx; m.d[6] = yz; m.d[8] = zx; m.d[9] = zy;
  return m;
}
Mat4 mat_view(V3 from, V3 to, V3 up) {  // matrix.rs:229-245
  V3 up_v = normalize(up);
  V3 forward = normalize(to - from);
  V3 left = cross(forward, up_v);
  V3 true_up = cross(left, forward);
  Mat4 o = mat_identity();
  o.d[0] = left.x; o.d[1] = left.y; o.d[2] = left.z;
  o.d[4] = true_up.x; o.d[5] = true_up.y; o.d[6] = true_up.z;
  o.d[8] = -forward.x; o.d[9] = -forward.y; o.d[10] = -forward.z;
  return mat_mul(o, mat_translation(-from.x, -from.y, -from.z));
}

// ------------------------------------------------------------------------------------------------
// Transformation  (src/math/transform.rs)
// ------------------------------------------------------------------------------------------------
Mat4 transformation_matrix(const Transformation& t) {  // transform.rs:24-40
  switch (t.kind) {
    case T_SCALING: return mat_scaling(t.v[0], t.v[1], t.v[2]);
    case T_TRANSLATION: return mat_translation(t.v[0], t.v[1], t.v[2]);
    case T_ROTATION: return mat_rotation(t.axis, t.v[0]);
    case T_SHEARING: return mat_shearing(t.v[0], t.v[1], t.v[2], t.v[3], t.v[4], t.v[5]);
    default: return mat_identity();
  }
}

// `t.interpolated(at)` = base + (t - base) * at, with  a - b := a + (b * -1.)   (animation.rs:307-341,
// transform.rs:42-52 Base, :132-151 Mul<f64>, :153-183 Add, :185-191 Sub).
Transformation transformation_interpolated(const Transformation& t, double at) {
  Transformation r = t;
  int n = 0;
  double base = 0.;
  switch (t.kind) {
    case T_SCALING: n = 3; base = 1.; break;
    case T_TRANSLATION: n = 3; break;
    case T_ROTATION: n = 1; break;
    case T_SHEARING: n = 6; break;
    default: return r;
  }
  for (int i = 0; i < n; i++) {
    double neg_base = base * -1.;
    double diff = t.v[i] + neg_base;
    r.v[i] = base + diff * at;
  }
  return r;
}

Mat4 transformations_matrix(const std::vector<Transformation>& ts) {  // transform.rs:118-124
  Mat4 acc = mat_identity();
  for (const auto& t : ts) acc = mat_mul(transformation_matrix(t), acc);  // acc.transform_new(m) = m * acc
  return acc;
}

// ------------------------------------------------------------------------------------------------
// BoundingBox  (src/scene/object/bounding_box.rs)
// ------------------------------------------------------------------------------------------------
bool BBox::is_empty() const {  // :91-98
  return min.x == INFINITY && min.y == INFINITY && min.z == INFINITY && max.x == -INFINITY && max.y == -INFINITY &&
         max.z == -INFINITY;
}
void BBox::add_point(V3 p) {  // :107-110 with Tuple::min/max (tuple.rs:26-40)
  min = {std::fmin(p.x, min.x), std::fmin(p.y, min.y), std::fmin(p.z, min.z)};
  max = {std::fmax(p.x, max.x), std::fmax(p.y, max.y), std::fmax(p.z, max.z)};
}
void BBox::add_bbox(const BBox& o) {  // :111-117
  if (o.is_empty()) return;
  add_point(o.min);
  add_point(o.max);
}
BBox BBox::transformed(const Mat4& m) const {  // :47-63 (8 corners in this order)
  BBox nb;
  V3 corners[8] = {min,
                   {min.x, min.y, max.z},
                   {min.x, max.y, min.z},
                   {min.x, max.y, max.z},
                   {max.x, min.y, min.z},
                   {max.x, min.y, max.z},
                   {max.x, max.y, min.z},
                   max};
  for (auto& c : corners) nb.add_point(mat_mul_point(m, c));
  return nb;
}
V3 BBox::center() const { return {(min.x + max.x) / 2., (min.y + max.y) / 2., (min.z + max.z) / 2.}; }  // :307-313
double BBox::half_area() const {  // :328-336
  if (is_empty()) return 0.;
  double xl = max.x - min.x, yl = max.y - min.y, zl = max.z - min.z;
  return xl * yl + yl * zl + zl * xl;
}
void BBox::clamp_dimensions() {  // :154-157, MAX_DIM :77
  const double MAX_DIM = 10.0e6;
  min = {std::fmax(min.x, -MAX_DIM), std::fmax(min.y, -MAX_DIM), std::fmax(min.z, -MAX_DIM)};
  max = {std::fmin(max.x, MAX_DIM), std::fmin(max.y, MAX_DIM), std::fmin(max.z, MAX_DIM)};
}
Mat4 BBox::as_cube_transformation() const {  // :295-302
  V3 c = center();
  V3 scale = mulf(max - min, 0.5);
  return mat_mul(mat_translation(c.x, c.y, c.z), mat_scaling(scale.x, scale.y, scale.z));
}
bool bbox_eq(const BBox& a, const BBox& b) { return v3_approx_eq(a.min, b.min) && v3_approx_eq(a.max, b.max); }

// ------------------------------------------------------------------------------------------------
// LocalTransformation  (src/math/transform/local_transform.rs)
// ------------------------------------------------------------------------------------------------
static Transformation translation(double x, double y, double z) {
  Transformation t;
  t.kind = T_TRANSLATION;
  t.v[0] = x; t.v[1] = y; t.v[2] = z;
  return t;
}
static Transformation scaling(double x, double y, double z) {
  Transformation t;
  t.kind = T_SCALING;
  t.v[0] = x; t.v[1] = y; t.v[2] = z;
  return t;
}

static std::vector<Transformation> local_around_center(const BBox& bbox, const Transformation& t) {  // :56-66
  V3 c = bbox.center();
  std::vector<Transformation> res{translation(-c.x, -c.y, -c.z)};
  Transformation put_back = transformation_interpolated(res[0], -1.);
  res.push_back(t);
  res.push_back(put_back);
  return res;
}

std::vector<Transformation> local_into_transformations(const LocalTransformation& lt, const BBox& bbox) {  // :69-124
  switch (lt.kind) {
    case L_CENTER: {
      V3 c = bbox.center();
      return {translation(-c.x, -c.y, -c.z)};
    }
    case L_TRANSLATE_ABOVE:
      if (lt.axis == 0) return {translation(-bbox.min.x, 0., 0.)};
      if (lt.axis == 1) return {translation(0., -bbox.min.y, 0.)};
      return {translation(0., 0., -bbox.min.z)};
    case L_TRANSLATE_BELOW:
      if (lt.axis == 0) return {translation(-bbox.max.x, 0., 0.)};
      if (lt.axis == 1) return {translation(0., -bbox.max.y, 0.)};
      return {translation(0., 0., -bbox.max.z)};
    case L_NORMALIZE_TO_AXIS: {
      V3 size = bbox.max - bbox.min;
      double len = lt.axis == 0 ? size.x : (lt.axis == 1 ? size.y : size.z);
      double f = 1. / len;
      return local_around_center(bbox, scaling(f, f, f));
    }
    case L_NORMALIZE_TO_LONGEST_AXIS: {
      V3 size = bbox.max - bbox.min;
      double len = std::fmax(std::fmax(size.x, size.y), size.z);  // bounding_box.rs:236-238
      double f = 1. / len;
      return local_around_center(bbox, scaling(f, f, f));
    }
    case L_NORMALIZE_ALL_AXES: {
      V3 size = bbox.max - bbox.min;
      return local_around_center(bbox, scaling(1. / size.x, 1. / size.y, 1. / size.z));
    }
    case L_LOCAL: return local_around_center(bbox, lt.t);
    default: return {lt.t};
  }
}

std::vector<Transformation> locals_interpolated_with(const std::vector<LocalTransformation>& ls, const BBox& obj_bbox,
                                                     double at) {  // :206-235
  BBox bbox = obj_bbox;
  Mat4 buf = mat_identity();
  std::vector<Transformation> out;
  for (const auto& lt : ls) {
    if (lt.kind != L_TRANSFORMATION) {
      bbox = bbox.transformed(buf);
      buf = mat_identity();
    }
    std::vector<Transformation> res = local_into_transformations(lt, bbox);  // :48-51
    for (auto& t : res) t = transformation_interpolated(t, at);
    Mat4 m = transformations_matrix(res);
    buf = mat_mul(m, buf);  // buf.transform(&m)
    out.insert(out.end(), res.begin(), res.end());
  }
  return out;
}

// ------------------------------------------------------------------------------------------------
// Material equality (derive(PartialEq): f64 fields exact, Color/Matrix approximate)
// ------------------------------------------------------------------------------------------------
bool material_eq(const Material& a, const Material& b) {
  if (a.pattern.kind != b.pattern.kind) return false;
  if (a.pattern.kind == RT_PATTERN_CONST) {
    if (!v3_approx_eq(a.pattern.c1, b.pattern.c1)) return false;
  } else if (a.pattern.kind == RT_PATTERN_TEST) {
    if (!mat_approx_eq(a.pattern.inv, b.pattern.inv)) return false;
  } else {
    if (!v3_approx_eq(a.pattern.c1, b.pattern.c1) || !v3_approx_eq(a.pattern.c2, b.pattern.c2) ||
        !mat_approx_eq(a.pattern.inv, b.pattern.inv))
      return false;
  }
  return a.ambient == b.ambient && a.diffuse == b.diffuse && a.specular == b.specular && a.shininess == b.shininess &&
         a.reflectivity == b.reflectivity && a.transparency == b.transparency && a.refractive_index == b.refractive_index;
}

// ------------------------------------------------------------------------------------------------
// Animation  (src/scene/animation.rs)
// ------------------------------------------------------------------------------------------------
static double powi2(double x) { return x * x; }
static double powi3(double x) { return x * (x * x); }
static double powi4(double x) { double s = x * x; return s * s; }
static double powi5(double x) { double s = x * x; return x * (s * s); }

double Animation::val_at(double time) const {  // :187-202
  time = time - delay;
  if (time < 0.) return 0.;
  double q = time / duration;
  uint32_t curr_count;  // `as u32` saturates, NaN -> 0
  if (!(q == q)) curr_count = 0;
  else if (q <= 0.) curr_count = 0;
  else if (q >= 4294967295.) curr_count = 4294967295u;
  else curr_count = (uint32_t)q;
  if (repeat >= 0 && !(curr_count < (uint64_t)repeat)) return 1.;
  double curr_time = time - (double)curr_count * duration;
  double p = curr_time / duration;
  double fraction;
  switch (timing) {  // :19-27
    case 0: fraction = 6. * powi5(p) - 15. * powi4(p) + 10. * powi3(p); break;
    case 1: fraction = powi3(p); break;
    case 2: fraction = 1. - powi3(1. - p); break;
    case 3: fraction = 3. * powi2(p) - 2. * powi3(p); break;
    default: fraction = p;
  }
  switch (direction) {  // :74-93
    case 0: return fraction;
    case 1: return 1. - fraction;
    case 2: return (curr_count % 2 == 0) ? fraction : 1. - fraction;
    default: return (curr_count % 2 == 0) ? 1. - fraction : fraction;
  }
}

double Animation::effective_period() const {  // :161-185
  if (duration <= 0.) return 0.;
  uint32_t count = repeat < 0 ? 1u : (uint32_t)repeat;
  bool alternates = direction == 2 || direction == 3;
  uint32_t repeats = (alternates && count % 2 == 1) ? count * 2 : count;
  return delay + (double)repeats * duration;
}

static uint32_t gcd_u32(uint32_t a, uint32_t b) { return b == 0 ? a : gcd_u32(b, a % b); }
static uint32_t animation_cycle(uint32_t a, uint32_t b) {  // :249-265
  if (a == 0) return b;
  if (b == 0) return a;
  uint32_t g = gcd_u32(a, b);
  return g == 0 ? 0 : a * b / g;
}

// ------------------------------------------------------------------------------------------------
// Objects
// ------------------------------------------------------------------------------------------------
ObjectPtr Object::clone() const {
  ObjectPtr o(new Object());
  o->kind = kind;
  o->animations = animations;
  o->shape = shape;
  o->y_min = y_min; o->y_max = y_max; o->closed = closed;
  o->p1 = p1; o->p2 = p2; o->p3 = p3; o->e1 = e1; o->e2 = e2; o->n1 = n1; o->n2 = n2; o->n3 = n3;
  o->material = material;
  o->has_inv = has_inv;
  o->inv = inv;
  o->bbox = bbox;
  o->primitive_count_ = primitive_count_;
  o->csg_op = csg_op;
  o->mesh_id = mesh_id;
  o->mesh_posed = mesh_posed;
  o->mesh_pending = mesh_pending;
  for (const auto& c : children) o->children.push_back(c->clone());
  if (left) o->left = left->clone();
  if (right) o->right = right->clone();
  return o;
}

size_t Object::primitive_count() const {  // object.rs:245-251
  return kind == OBJ_GROUP ? primitive_count_ : 1;
}

static BBox shape_bbox(const Object& o) {  // shape.rs:63-76
  BBox b;
  switch (o.shape) {
    case RT_SHAPE_SPHERE:
    case RT_SHAPE_CUBE:
    case RT_SHAPE_BBOX:
      b.min = {-1., -1., -1.};
      b.max = {1., 1., 1.};
      break;
    case RT_SHAPE_PLANE:  // plane.rs:20-25
      b.min = {-INFINITY, 0., -INFINITY};
      b.max = {INFINITY, 0., INFINITY};
      break;
    case RT_SHAPE_CYLINDER:  // cylinder.rs:95-100
      b.min = {-1., o.y_min, -1.};
      b.max = {1., o.y_max, 1.};
      break;
    case RT_SHAPE_CONE: {  // cone.rs:96-104
      double r = std::fmax(std::fabs(o.y_min), std::fabs(o.y_max));
      b.min = {-r, o.y_min, -r};
      b.max = {r, o.y_max, r};
      break;
    }
    default:  // triangle.rs:73-79
      b.add_point(o.p1);
      b.add_point(o.p2);
      b.add_point(o.p3);
  }
  b.clamp_dimensions();
  return b;
}

static void triangle_derive(Object& o) {  // triangle.rs:34-47 / smooth_triangle.rs:42-55
  o.e1 = o.p2 - o.p1;
  o.e2 = o.p3 - o.p1;
  if (o.shape == RT_SHAPE_TRIANGLE) {
    o.n1 = normalize(cross(o.e2, o.e1));
    o.n2 = {0, 0, 0};
    o.n3 = {0, 0, 0};
  }
}

thread_local MeshPoseHook mesh_pose_hook = nullptr;
thread_local void* mesh_pose_user = nullptr;

void Object::recalculate_bbox() {
  if (kind == OBJ_GROUP && mesh_id >= 0) {  // the children are on the device: their union comes from there
    if (mesh_pose_hook) mesh_pose_hook(*this, mesh_pose_user);
    return;
  }
  if (kind == OBJ_GROUP) {  // group.rs:230-235
    bbox = BBox();
    for (const auto& c : children) bbox.add_bbox(c->bbox);
  } else if (kind == OBJ_CSG) {  // csg.rs:131-143
    if (csg_op == RT_CSG_DIFFERENCE) {
      bbox = left->bbox;  // bounding_box.rs:119-121
    } else if (csg_op == RT_CSG_UNION) {
      bbox = left->bbox;
      bbox.add_bbox(right->bbox);
    } else {  // bounding_box.rs:123-133
      BBox r;
      const BBox &a = left->bbox, &b = right->bbox;
      r.min = {std::fmax(b.min.x, a.min.x), std::fmax(b.min.y, a.min.y), std::fmax(b.min.z, a.min.z)};
      r.max = {std::fmin(b.max.x, a.max.x), std::fmin(b.max.y, a.max.y), std::fmin(b.max.z, a.max.z)};
      V3 size = r.max - r.min;
      bool valid = r.is_empty() || (size.x >= 0. && size.y >= 0. && size.z >= 0.);
      bbox = valid ? r : BBox();
    }
  }
}

void Object::transform(const Mat4& m) {
  if (kind == OBJ_PRIMITIVE) {  // object.rs:454-472
    bbox = bbox.transformed(m);
    if (shape == RT_SHAPE_TRIANGLE) {  // triangle.rs:24-30
      p1 = mat_mul_point(m, p1);
      p2 = mat_mul_point(m, p2);
      p3 = mat_mul_point(m, p3);
      triangle_derive(*this);
    } else if (shape == RT_SHAPE_SMOOTH_TRIANGLE) {  // smooth_triangle.rs:29-38 (normals by the forward matrix)
      p1 = mat_mul_point(m, p1);
      p2 = mat_mul_point(m, p2);
      p3 = mat_mul_point(m, p3);
      n1 = mat_mul_vector(m, n1);
      n2 = mat_mul_vector(m, n2);
      n3 = mat_mul_vector(m, n3);
      triangle_derive(*this);
    } else if (shape == RT_SHAPE_BBOX) {
      Mat4 i;
      mat_inverse(bbox.as_cube_transformation(), i);
      inv = i;
      has_inv = true;
    } else {
      Mat4 mi;
      mat_inverse(m, mi);
      if (has_inv) {
        inv = mat_mul(inv, mi);  // matrix.inverse().transform_new(inv) = inv * matrix^-1
      } else {
        inv = mi;
        has_inv = true;
      }
    }
  } else if (kind == OBJ_GROUP) {  // group.rs:263-270
    for (auto& c : children) c->transform(m);
    if (mesh_id >= 0) mesh_pending.push_back(m);  // (the children of a mesh proxy are transformed on the device)
    bbox = bbox.transformed(m);
  } else {  // csg.rs:157-163
    left->transform(m);
    right->transform(m);
    recalculate_bbox();
  }
}

void Object::local_transform(const std::vector<LocalTransformation>& ls) {  // local_transform.rs:15-20,156-158
  Mat4 m = transformations_matrix(locals_interpolated_with(ls, bbox, 1.));
  transform(m);
}

void Object::set_material(const Material& m) {  // object.rs:233-239, group.rs:71-75, csg.rs:124-127
  if (kind == OBJ_PRIMITIVE) {
    material = m;
  } else if (kind == OBJ_GROUP) {
    for (auto& c : children) c->set_material(m);
  } else {
    left->set_material(m);
    right->set_material(m);
  }
}

void Object::add_child(ObjectPtr child) {  // group.rs:76-80
  bbox.add_bbox(child->bbox);
  primitive_count_ += child->primitive_count();
  children.push_back(std::move(child));
}

static double split_cost(const Object& g, int axis, double pos) {  // group.rs:109-129
  BBox lb, rb;
  size_t lc = 0, rc = 0;
  for (const auto& c : g.children) {
    V3 ce = c->bbox.center();
    double cv = axis == 0 ? ce.x : (axis == 1 ? ce.y : ce.z);
    if (cv < pos) {
      lb.add_bbox(c->bbox);
      lc += c->primitive_count();
    } else {
      rb.add_bbox(c->bbox);
      rc += c->primitive_count();
    }
  }
  if (lc == 0 || rc == 0) return INFINITY;
  return lb.half_area() * (double)lc + rb.half_area() * (double)rc;
}

// The reference's recursion written the way the reference writes it: 15 passes over the children per group.  Kept as the
// cross-check of build_bvh() below (rts_flatten build_bvh = 2; tests/test_host_scene.py).
void Object::build_bvh_plain() {  // group.rs:144-167 (+ choose_split :87-108, split :131-143)
  if (kind != OBJ_GROUP) return;
  if (children.size() <= 2) {
    for (auto& c : children)
      if (c->kind == OBJ_GROUP) c->build_bvh_plain();
    return;
  }
  int best_axis = 0;
  double best_pos = 0.0, best_cost = INFINITY;
  V3 len = bbox.max - bbox.min;
  const int CANDIDATE_POS_NUMBER = 5;
  for (int i = 0; i < CANDIDATE_POS_NUMBER; i++) {
    double coef = ((double)i + 1.) / ((double)CANDIDATE_POS_NUMBER + 1.);
    V3 pos = bbox.min + mulf(len, coef);
    for (int axis = 0; axis < 3; axis++) {
      double p = axis == 0 ? pos.x : (axis == 1 ? pos.y : pos.z);
      double cost = split_cost(*this, axis, p);
      if (cost < best_cost) {
        best_cost = cost;
        best_axis = axis;
        best_pos = p;
      }
    }
  }
  double sah = bbox.half_area() * (double)primitive_count_;  // group.rs:220-222
  if (best_cost >= sah) return;
  ObjectPtr l(new Object()), r(new Object());
  l->kind = OBJ_GROUP;
  r->kind = OBJ_GROUP;
  std::vector<ObjectPtr> old;
  old.swap(children);
  for (auto& c : old) {
    V3 ce = c->bbox.center();
    double cv = best_axis == 0 ? ce.x : (best_axis == 1 ? ce.y : ce.z);
    if (cv < best_pos) l->add_child(std::move(c));
    else r->add_child(std::move(c));
  }
  if (l->primitive_count() > 0) {
    l->build_bvh_plain();
    children.push_back(std::move(l));
  }
  if (r->primitive_count() > 0) {
    r->build_bvh_plain();
    children.push_back(std::move(r));
  }
}


// ------------------------------------------------------------------------------------------------
// build_bvh: the same tree as build_bvh_plain, built over a compact array.
//
// The plain form walks `children` (pointers to ~700-byte objects) 15 times per group: on a 140 000-triangle scene that
// is ~10^8 cache misses, 2.8 s per frame of an animation.  Here every child of the group is first summarised as
// 88 bytes (centre, box, primitive count), the 15 candidate planes are scored from ONE pass over those (per axis a child
// falls into one of 2^5 classes, by which of the 5 planes its centre lies below; a candidate's two sides are unions
// of classes), and split() becomes a stable partition of the array.  min/max unions and integer counts do not depend on
// the order they are accumulated in, so costs, choices and therefore the tree are the reference's (the cross-check
// asserts identical flattened tables).  Classes rather than ordered bins so that nothing is assumed about the
// candidate positions: an unbounded box (a plane) makes them NaN, and then every child is on the right, as in
// calculate_split_cost.
// ------------------------------------------------------------------------------------------------
namespace {
struct BuildItem {
  double c[3];
  BBox bb;
  size_t count;
  ObjectPtr obj;
};
// Sums of the scoring pass.  Same arithmetic as BBox::add_bbox / half_area, with the NaN-ignoring min/max written inline
// (std::fmin is a libm call at -O2); same results, see min_ignoring_nan).
struct SideSum {
  double mn[3], mx[3];
  size_t count;
};
// f64::min/max drop a NaN operand (tuple.rs:26-40).  The running value starts at +-inf and therefore never becomes NaN, so
// "take the new value only if it compares below/above" is the same function, and compiles to minsd/maxsd.
inline double min_ignoring_nan(double v, double running) { return v < running ? v : running; }
inline double max_ignoring_nan(double v, double running) { return v > running ? v : running; }
inline void side_reset(SideSum& s) {
  for (int d = 0; d < 3; d++) {
    s.mn[d] = INFINITY;
    s.mx[d] = -INFINITY;
  }
  s.count = 0;
}
inline bool side_empty(const double* mn, const double* mx) {  // bounding_box.rs:91-98
  return mn[0] == INFINITY && mn[1] == INFINITY && mn[2] == INFINITY && mx[0] == -INFINITY && mx[1] == -INFINITY &&
         mx[2] == -INFINITY;
}
inline void side_add(SideSum& s, const double* mn, const double* mx, size_t count) {  // bounding_box.rs:107-117
  s.count += count;
  if (side_empty(mn, mx)) return;
  for (int d = 0; d < 3; d++) {
    s.mn[d] = min_ignoring_nan(mn[d], s.mn[d]);
    s.mx[d] = max_ignoring_nan(mn[d], s.mx[d]);
  }
  for (int d = 0; d < 3; d++) {
    s.mn[d] = min_ignoring_nan(mx[d], s.mn[d]);
    s.mx[d] = max_ignoring_nan(mx[d], s.mx[d]);
  }
}
inline double side_half_area(const SideSum& s) {  // bounding_box.rs:328-336
  if (side_empty(s.mn, s.mx)) return 0.;
  double xl = s.mx[0] - s.mn[0], yl = s.mx[1] - s.mn[1], zl = s.mx[2] - s.mn[2];
  return xl * yl + yl * zl + zl * xl;
}

void build_items(Object& g, std::vector<BuildItem>& items, size_t lo, size_t hi, std::vector<BuildItem>& tmp) {
  const size_t n = hi - lo;
  auto adopt = [&]() {
    g.children.reserve(n);
    for (size_t i = lo; i < hi; i++) g.children.push_back(std::move(items[i].obj));
  };
  if (n <= 2) {  // group.rs:145-152
    adopt();
    for (auto& c : g.children)
      if (c->kind == OBJ_GROUP) c->build_bvh();
    return;
  }
  const int CANDIDATE_POS_NUMBER = 5;
  double pos[CANDIDATE_POS_NUMBER][3];
  V3 len = g.bbox.max - g.bbox.min;
  for (int i = 0; i < CANDIDATE_POS_NUMBER; i++) {  // group.rs:93-95
    double coef = ((double)i + 1.) / ((double)CANDIDATE_POS_NUMBER + 1.);
    V3 p = g.bbox.min + mulf(len, coef);
    pos[i][0] = p.x; pos[i][1] = p.y; pos[i][2] = p.z;
  }
  SideSum cls[3][1 << CANDIDATE_POS_NUMBER];  // initialised on first use: `used` says which
  uint32_t used[3] = {0, 0, 0};
  for (size_t k = lo; k < hi; k++) {
    const BuildItem& it = items[k];
    for (int axis = 0; axis < 3; axis++) {
      unsigned m = 0;
      for (int i = 0; i < CANDIDATE_POS_NUMBER; i++) m |= (unsigned)(it.c[axis] < pos[i][axis]) << i;
      if (!((used[axis] >> m) & 1u)) {
        used[axis] |= 1u << m;
        side_reset(cls[axis][m]);
      }
      side_add(cls[axis][m], &it.bb.min.x, &it.bb.max.x, it.count);
    }
  }
  // Both sides of every candidate.  Ordered candidate positions (the normal case) leave only the classes "below none",
  // "below the last", "below the last two", ...: candidate i's left side is then a prefix of those and its right side
  // the matching suffix (5 + 5 unions per axis); any other class (NaN positions) takes the general path.
  SideSum left[CANDIDATE_POS_NUMBER][3], right[CANDIDATE_POS_NUMBER][3];
  for (int axis = 0; axis < 3; axis++) {
    constexpr uint32_t ORDERED = (1u << 0) | (1u << 16) | (1u << 24) | (1u << 28) | (1u << 30) | (1u << 31);
    if ((used[axis] & ~ORDERED) == 0) {
      // class k (k = 0..5) = centres below the candidates k..4 only: mask (0b11111 << k) & 31
      SideSum run;
      side_reset(run);
      for (int i = 0; i < CANDIDATE_POS_NUMBER; i++) {  // left of candidate i: classes 0..i
        unsigned m = (0x1Fu << i) & 0x1Fu;
        if ((used[axis] >> m) & 1u) side_add(run, cls[axis][m].mn, cls[axis][m].mx, cls[axis][m].count);
        left[i][axis] = run;
      }
      side_reset(run);
      for (int i = CANDIDATE_POS_NUMBER - 1; i >= 0; i--) {  // right of candidate i: classes i+1..5
        unsigned m = (0x1Fu << (i + 1)) & 0x1Fu;
        if ((used[axis] >> m) & 1u) side_add(run, cls[axis][m].mn, cls[axis][m].mx, cls[axis][m].count);
        right[i][axis] = run;
      }
    } else {
      for (int i = 0; i < CANDIDATE_POS_NUMBER; i++) {
        side_reset(left[i][axis]);
        side_reset(right[i][axis]);
        for (uint32_t rest = used[axis]; rest; rest &= rest - 1) {
          unsigned m = (unsigned)__builtin_ctz(rest);
          const SideSum& c = cls[axis][m];
          side_add(((m >> i) & 1u) ? left[i][axis] : right[i][axis], c.mn, c.mx, c.count);
        }
      }
    }
  }
  int best_axis = 0;
  double best_pos = 0.0, best_cost = INFINITY;
  for (int i = 0; i < CANDIDATE_POS_NUMBER; i++)
    for (int axis = 0; axis < 3; axis++) {  // group.rs:96-105, :109-129
      const SideSum &l = left[i][axis], &r = right[i][axis];
      double cost = (l.count == 0 || r.count == 0) ? INFINITY
                                                   : side_half_area(l) * (double)l.count + side_half_area(r) * (double)r.count;
      if (cost < best_cost) {
        best_cost = cost;
        best_axis = axis;
        best_pos = pos[i][axis];
      }
    }
  double sah = g.bbox.half_area() * (double)g.primitive_count_;  // group.rs:220-222
  if (best_cost >= sah) {  // group.rs:154-156: the children stay as they are, nested groups are NOT visited
    adopt();
    return;
  }
  // split (group.rs:131-143): a stable partition, left = centre below the plane
  SideSum lsum, rsum;
  side_reset(lsum);
  side_reset(rsum);
  size_t nl = 0, nr = 0;
  for (size_t k = lo; k < hi; k++) {
    BuildItem& it = items[k];
    if (it.c[best_axis] < best_pos) {
      side_add(lsum, &it.bb.min.x, &it.bb.max.x, it.count);
      if (lo + nl != k) items[lo + nl] = std::move(it);
      nl++;
    } else {
      side_add(rsum, &it.bb.min.x, &it.bb.max.x, it.count);
      tmp[nr++] = std::move(it);
    }
  }
  for (size_t k = 0; k < nr; k++) items[lo + nl + k] = std::move(tmp[k]);
  auto new_group = [](const SideSum& s) {  // ObjectGroup::empty + add_child per child (group.rs:62-80)
    ObjectPtr o(new Object());
    o->kind = OBJ_GROUP;
    o->bbox.min = {s.mn[0], s.mn[1], s.mn[2]};
    o->bbox.max = {s.mx[0], s.mx[1], s.mx[2]};
    o->primitive_count_ = s.count;
    return o;
  };
  // `tmp` is free again below: each recursion uses it only between its own partition and this point
  if (lsum.count > 0) {  // group.rs:159-166
    ObjectPtr l = new_group(lsum);
    build_items(*l, items, lo, lo + nl, tmp);
    g.children.push_back(std::move(l));
  }
  if (rsum.count > 0) {
    ObjectPtr r = new_group(rsum);
    build_items(*r, items, lo + nl, hi, tmp);
    g.children.push_back(std::move(r));
  }
}
}  // namespace

void Object::build_bvh() {  // group.rs:144-167
  if (kind != OBJ_GROUP) return;
  std::vector<BuildItem> items(children.size()), tmp(children.size());
  for (size_t k = 0; k < children.size(); k++) {
    BuildItem& it = items[k];
    V3 ce = children[k]->bbox.center();
    it.c[0] = ce.x; it.c[1] = ce.y; it.c[2] = ce.z;
    it.bb = children[k]->bbox;
    it.count = children[k]->primitive_count();
    it.obj = std::move(children[k]);
  }
  children.clear();
  build_items(*this, items, 0, items.size(), tmp);
}

ObjectPtr make_primitive(int shape, double y_min, double y_max, bool closed, const double* tri, const Material& material,
                         const Mat4& transformation) {  // object.rs:372-383
  ObjectPtr o(new Object());
  o->kind = OBJ_PRIMITIVE;
  o->shape = shape;
  o->y_min = y_min;
  o->y_max = y_max;
  o->closed = closed;
  if (tri) {
    o->p1 = {tri[0], tri[1], tri[2]};
    o->p2 = {tri[3], tri[4], tri[5]};
    o->p3 = {tri[6], tri[7], tri[8]};
    if (shape == RT_SHAPE_SMOOTH_TRIANGLE) {
      o->n1 = {tri[9], tri[10], tri[11]};
      o->n2 = {tri[12], tri[13], tri[14]};
      o->n3 = {tri[15], tri[16], tri[17]};
    }
    triangle_derive(*o);
  }
  o->material = material;
  o->bbox = shape_bbox(*o);
  if (!mat_approx_eq(transformation, mat_identity())) o->transform(transformation);
  return o;
}

ObjectPtr make_group(std::vector<ObjectPtr> children) {  // group.rs:37-43
  ObjectPtr g(new Object());
  g->kind = OBJ_GROUP;
  for (auto& c : children) g->add_child(std::move(c));
  return g;
}

ObjectPtr make_csg(int op, ObjectPtr left, ObjectPtr right) {  // csg.rs:77-86
  ObjectPtr c(new Object());
  c->kind = OBJ_CSG;
  c->csg_op = op;
  c->left = std::move(left);
  c->right = std::move(right);
  c->recalculate_bbox();
  return c;
}

void Object::into_group_with_bbox(const Material& m) {  // object.rs:169-179, bounding_box.rs:304-306
  ObjectPtr bbox_obj = make_primitive(RT_SHAPE_BBOX, 0, 0, false, nullptr, m, bbox.as_cube_transformation());
  ObjectPtr inner(new Object());
  // move this object's kind payload into `inner`; animations stay on the outer object
  inner->kind = kind;
  inner->shape = shape;
  inner->y_min = y_min; inner->y_max = y_max; inner->closed = closed;
  inner->p1 = p1; inner->p2 = p2; inner->p3 = p3; inner->e1 = e1; inner->e2 = e2;
  inner->n1 = n1; inner->n2 = n2; inner->n3 = n3;
  inner->material = material;
  inner->has_inv = has_inv;
  inner->inv = inv;
  inner->bbox = bbox;
  inner->children = std::move(children);
  inner->primitive_count_ = primitive_count_;
  inner->csg_op = csg_op;
  inner->left = std::move(left);
  inner->right = std::move(right);

  kind = OBJ_GROUP;
  children.clear();
  bbox = BBox();
  primitive_count_ = 0;
  add_child(std::move(inner));
  add_child(std::move(bbox_obj));
}

static Mat4 animations_matrix(const Object& o, double time) {  // animation.rs:211-243
  Mat4 acc = mat_identity();
  for (const auto& a : o.animations) {
    double at = a.animation.val_at(time);
    Mat4 m = approx_eq(at, 0.) ? mat_identity()
                               : transformations_matrix(locals_interpolated_with(a.transformations, o.bbox, at));
    acc = mat_mul(m, acc);
  }
  return acc;
}

void Object::group_animate(double time) {  // group.rs:236-241
  for (auto& c : children) c->animate(time);
  recalculate_bbox();
}

void Object::animate(double time) {  // object.rs:108-121
  Mat4 t = animations_matrix(*this, time);
  bool is_identity = mat_approx_eq(t, mat_identity());
  if (kind == OBJ_PRIMITIVE) {
    if (!is_identity) transform(t);
  } else if (kind == OBJ_GROUP) {  // group.rs:242-247
    if (!is_identity) transform(t);
    group_animate(time);
  } else {  // csg.rs:144-154
    if (!is_identity) transform(t);
    left->animate(time);
    right->animate(time);
    recalculate_bbox();
  }
}

uint32_t Object::animation_cycle_duration_ms() const {  // object.rs:60-74, animation.rs:245-247,284-289
  uint32_t dur = 0;
  for (const auto& a : animations) {
    double ms = std::round(a.animation.effective_period() * 1000.0);
    uint32_t v = ms <= 0. ? 0u : (ms >= 4294967295. ? 4294967295u : (uint32_t)ms);
    dur = animation_cycle(dur, v);
  }
  if (kind == OBJ_GROUP) {
    uint32_t g = 0;
    for (const auto& c : children) g = animation_cycle(g, c->animation_cycle_duration_ms());
    return animation_cycle(dur, g);
  }
  if (kind == OBJ_CSG) {
    dur = animation_cycle(dur, left->animation_cycle_duration_ms());
    dur = animation_cycle(dur, right->animation_cycle_duration_ms());
  }
  return dur;
}

static bool transformation_eq(const Transformation& a, const Transformation& b) {
  if (a.kind != b.kind) return false;
  if (a.kind == T_ROTATION && a.axis != b.axis) return false;
  for (int i = 0; i < 6; i++)
    if (a.v[i] != b.v[i]) return false;
  return true;
}
static bool animations_eq(const std::vector<TransformAnimation>& a, const std::vector<TransformAnimation>& b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); i++) {
    const Animation &x = a[i].animation, &y = b[i].animation;
    if (x.delay != y.delay || x.duration != y.duration || x.direction != y.direction || x.timing != y.timing ||
        x.repeat != y.repeat)
      return false;
    if (a[i].transformations.size() != b[i].transformations.size()) return false;
    for (size_t j = 0; j < a[i].transformations.size(); j++) {
      const auto &p = a[i].transformations[j], &q = b[i].transformations[j];
      if (p.kind != q.kind) return false;
      if ((p.kind >= L_TRANSLATE_ABOVE && p.kind <= L_NORMALIZE_TO_AXIS) && p.axis != q.axis) return false;
      if ((p.kind == L_LOCAL || p.kind == L_TRANSFORMATION) && !transformation_eq(p.t, q.t)) return false;
    }
  }
  return true;
}

// derive(PartialEq) over Object / ObjectKind / PrimitiveObject / ObjectGroup / CsgObject
bool object_eq(const Object& a, const Object& b) {
  if (a.kind != b.kind) return false;
  if (!animations_eq(a.animations, b.animations)) return false;
  if (a.kind == OBJ_PRIMITIVE) {
    if (a.shape != b.shape) return false;
    if (a.shape == RT_SHAPE_CYLINDER || a.shape == RT_SHAPE_CONE) {
      if (a.y_min != b.y_min || a.y_max != b.y_max || a.closed != b.closed) return false;
    }
    if (a.shape == RT_SHAPE_TRIANGLE || a.shape == RT_SHAPE_SMOOTH_TRIANGLE) {
      if (!v3_approx_eq(a.p1, b.p1) || !v3_approx_eq(a.p2, b.p2) || !v3_approx_eq(a.p3, b.p3) ||
          !v3_approx_eq(a.e1, b.e1) || !v3_approx_eq(a.e2, b.e2) || !v3_approx_eq(a.n1, b.n1))
        return false;
      if (a.shape == RT_SHAPE_SMOOTH_TRIANGLE && (!v3_approx_eq(a.n2, b.n2) || !v3_approx_eq(a.n3, b.n3))) return false;
    }
    if (!material_eq(a.material, b.material)) return false;
    if (a.has_inv != b.has_inv) return false;
    if (a.has_inv && !mat_approx_eq(a.inv, b.inv)) return false;
    return bbox_eq(a.bbox, b.bbox);
  }
  if (a.kind == OBJ_GROUP) {
    if (a.children.size() != b.children.size()) return false;
    for (size_t i = 0; i < a.children.size(); i++)
      if (!object_eq(*a.children[i], *b.children[i])) return false;
    return bbox_eq(a.bbox, b.bbox) && a.primitive_count_ == b.primitive_count_;
  }
  return a.csg_op == b.csg_op && object_eq(*a.left, *b.left) && object_eq(*a.right, *b.right) && bbox_eq(a.bbox, b.bbox);
}

bool Object::includes(const Object& other) const {  // object.rs:98-104, group.rs:224-226
  if (object_eq(*this, other)) return true;
  if (kind == OBJ_GROUP) {
    for (const auto& c : children)
      if (c->includes(other)) return true;
  }
  return false;
}

// ------------------------------------------------------------------------------------------------
// Wavefront OBJ  (src/scene/io/obj_model.rs)
// ------------------------------------------------------------------------------------------------
ObjectPtr parse_obj_model(const std::string& source, std::string& err) {
  std::vector<V3> vertices, normals;
  ObjectPtr main_group = make_group({});
  std::vector<std::pair<std::string, ObjectPtr>> groups;  // insertion order (the reference's HashMap order is unspecified)
  int curr = -1;
  bool has_curr_name = false;
  std::string curr_name;
  Material default_material;
  Mat4 identity = mat_identity();

  std::istringstream in(source);
  std::string line;
  while (std::getline(in, line)) {
    // line.trim()
    size_t b = line.find_first_not_of(" \t\r\n\f\v");
    if (b == std::string::npos) continue;
    size_t e = line.find_last_not_of(" \t\r\n\f\v");
    line = line.substr(b, e - b + 1);
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ls(line);
    std::string tag;
    ls >> tag;
    if (tag == "v" || tag == "vn") {
      std::string sx, sy, sz;
      if (!(ls >> sx >> sy >> sz)) {
        err = "bad tuple: " + line;
        return nullptr;
      }
      V3 v{std::strtod(sx.c_str(), nullptr), std::strtod(sy.c_str(), nullptr), std::strtod(sz.c_str(), nullptr)};
      (tag == "v" ? vertices : normals).push_back(v);
    } else if (tag == "f") {  // :94-116
      std::vector<size_t> vi, ni;
      std::string tok;
      while (ls >> tok) {
        std::vector<std::string> parts;
        size_t start = 0;
        while (true) {
          size_t p = tok.find('/', start);
          if (p == std::string::npos) {
            parts.push_back(tok.substr(start));
            break;
          }
          parts.push_back(tok.substr(start, p - start));
          start = p + 1;
        }
        vi.push_back((size_t)std::strtoull(parts[0].c_str(), nullptr, 10) - 1);
        if (parts.size() >= 3) ni.push_back((size_t)std::strtoull(parts[2].c_str(), nullptr, 10) - 1);
      }
      if (vi.empty()) continue;
      for (size_t i : vi)
        if (i >= vertices.size()) {
          err = "vertex index out of range: " + line;
          return nullptr;
        }
      if (has_curr_name) {  // curr_group_mut(): groups.entry(name).or_default()  (:37-42)
        curr = -1;
        for (size_t i = 0; i < groups.size(); i++)
          if (groups[i].first == curr_name) curr = (int)i;
        if (curr < 0) {
          groups.emplace_back(curr_name, make_group({}));
          curr = (int)groups.size() - 1;
        }
      }
      Object* target = curr < 0 ? main_group.get() : groups[curr].second.get();
      V3 v = vertices[vi[0]];
      if (ni.empty()) {  // :51-62 fan of flat triangles
        for (size_t k = 2; k < vi.size(); k++) {
          double tri[9] = {v.x, v.y, v.z, vertices[vi[k - 1]].x, vertices[vi[k - 1]].y, vertices[vi[k - 1]].z,
                           vertices[vi[k]].x, vertices[vi[k]].y, vertices[vi[k]].z};
          target->add_child(make_primitive(RT_SHAPE_TRIANGLE, 0, 0, false, tri, default_material, identity));
        }
      } else {  // :63-89 fan of smooth triangles
        if (ni.size() != vi.size()) {
          err = "face mixes vertices with and without normals: " + line;
          return nullptr;
        }
        for (size_t i : ni)
          if (i >= normals.size()) {
            err = "normal index out of range: " + line;
            return nullptr;
          }
        V3 n = normals[ni[0]];
        for (size_t k = 2; k < vi.size(); k++) {
          V3 a = vertices[vi[k - 1]], c = vertices[vi[k]], na = normals[ni[k - 1]], nc = normals[ni[k]];
          double tri[18] = {v.x, v.y, v.z, a.x, a.y, a.z, c.x, c.y, c.z, n.x, n.y, n.z, na.x, na.y, na.z, nc.x, nc.y, nc.z};
          target->add_child(make_primitive(RT_SHAPE_SMOOTH_TRIANGLE, 0, 0, false, tri, default_material, identity));
        }
      }
    } else if (tag == "g") {  // :118-120,154-156
      ls >> curr_name;
      has_curr_name = true;
    }
  }
  for (auto& g : groups) main_group->add_child(std::move(g.second));  // :176-181
  return main_group;
}

}  // namespace rts
