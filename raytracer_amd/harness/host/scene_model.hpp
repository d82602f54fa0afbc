// scene_model.hpp — host-side scene model: the part of the reference that stays on the CPU and feeds
// the flattener.  It mirrors, operation for operation, the f64 arithmetic of
//   src/math/matrix.rs, src/math/transform.rs, src/math/transform/local_transform.rs,
//   src/scene/object.rs, src/scene/object/{group,bounding_box,csg}.rs, src/scene/object/primitive/*.rs,
//   src/scene/animation.rs, src/scene/io/obj_model.rs, src/scene/camera.rs:160-187
// so that the flattened tables (include/rtcore.h) are bit-identical to what the Rust FFI crate would emit.
// Build with -ffp-contract=off: the reference (rustc) never fuses a*b+c.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace rts {

constexpr double EPSILON = 1.0e-8;  // src/math/approx_eq.rs:1
inline bool approx_eq(double a, double b) { return a == b || std::fabs(a - b) < EPSILON; }  // approx_eq.rs:16-20

struct V3 {
  double x = 0, y = 0, z = 0;
};
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 mulf(V3 a, double f) { return {f * a.x, f * a.y, f * a.z}; }  // vector.rs:136-146 (rhs * self.x)
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double magnitude(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 normalize(V3 a) {  // vector.rs:57-67
  double len = magnitude(a);
  if (len == 0.) return {0, 0, 0};
  return {a.x / len, a.y / len, a.z / len};
}
inline bool v3_approx_eq(V3 a, V3 b) { return approx_eq(a.x, b.x) && approx_eq(a.y, b.y) && approx_eq(a.z, b.z); }

struct Mat4 {
  double d[16];
  double& at(int r, int c) { return d[r * 4 + c]; }
  double at(int r, int c) const { return d[r * 4 + c]; }
};
Mat4 mat_identity();
Mat4 mat_mul(const Mat4& a, const Mat4& b);       // matrix.rs:288-302
bool mat_inverse(const Mat4& m, Mat4& out);       // matrix.rs:93-144
bool mat_approx_eq(const Mat4& a, const Mat4& b); // matrix.rs:248-255
V3 mat_mul_point(const Mat4& m, V3 p);            // matrix.rs:316-337 with w = 1
V3 mat_mul_vector(const Mat4& m, V3 v);           // matrix.rs:316-337 with w = 0
Mat4 mat_translation(double x, double y, double z);
Mat4 mat_scaling(double x, double y, double z);
Mat4 mat_rotation(int axis, double r);
Mat4 mat_shearing(double a, double b, double c, double d, double e, double f);
Mat4 mat_view(V3 from, V3 to, V3 up);             // matrix.rs:229-245

// src/math/transform.rs:10-16
enum TKind : int { T_SCALING = 0, T_TRANSLATION = 1, T_ROTATION = 2, T_SHEARING = 3, T_IDENTITY = 4 };
struct Transformation {
  int kind = T_IDENTITY;
  int axis = 0;      // rotation only
  double v[6] = {0, 0, 0, 0, 0, 0};
};
Mat4 transformation_matrix(const Transformation& t);              // transform.rs:24-40
Transformation transformation_interpolated(const Transformation& t, double at);  // animation.rs:307-341 + transform.rs:42-52,132-191
Mat4 transformations_matrix(const std::vector<Transformation>& ts);  // transform.rs:118-124

// src/math/transform/local_transform.rs:25-42
enum LKind : int {
  L_CENTER = 0,
  L_TRANSLATE_ABOVE = 1,
  L_TRANSLATE_BELOW = 2,
  L_NORMALIZE_TO_AXIS = 3,
  L_NORMALIZE_ALL_AXES = 4,
  L_NORMALIZE_TO_LONGEST_AXIS = 5,
  L_LOCAL = 6,
  L_TRANSFORMATION = 7
};
struct LocalTransformation {
  int kind = L_TRANSFORMATION;
  int axis = 0;
  Transformation t;
};

struct BBox {  // bounding_box.rs:25-28
  V3 min{INFINITY, INFINITY, INFINITY};
  V3 max{-INFINITY, -INFINITY, -INFINITY};
  bool is_empty() const;
  void add_point(V3 p);
  void add_bbox(const BBox& o);
  BBox transformed(const Mat4& m) const;
  V3 center() const;
  double half_area() const;
  void clamp_dimensions();
  Mat4 as_cube_transformation() const;
};
bool bbox_eq(const BBox& a, const BBox& b);

std::vector<Transformation> local_into_transformations(const LocalTransformation& lt, const BBox& bbox);  // local_transform.rs:69-124
std::vector<Transformation> locals_interpolated_with(const std::vector<LocalTransformation>& ls, const BBox& obj_bbox,
                                                     double at);  // local_transform.rs:206-235

struct Pattern {  // pattern.rs:7-37
  int kind = 0;   // enum rt_pattern
  V3 c1{1, 1, 1}, c2{0, 0, 0};
  Mat4 inv = mat_identity();
};
struct Material {  // material.rs:10-20, defaults :56-69
  Pattern pattern;
  double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200., reflectivity = 0., transparency = 0.,
         refractive_index = 1.0;
};
bool material_eq(const Material& a, const Material& b);

// src/scene/animation.rs
struct Animation {
  double delay = 0., duration = 1.;
  int direction = 0;  // 0 normal 1 reverse 2 alternate 3 alternate-reverse
  int timing = 0;     // 0 ease 1 ease-in 2 ease-out 3 ease-in-out 4 linear
  int64_t repeat = -1;  // -1 infinite, else count
  double val_at(double time) const;
  double effective_period() const;
};
struct TransformAnimation {
  Animation animation;
  std::vector<LocalTransformation> transformations;
};

enum ObjKind : int { OBJ_PRIMITIVE = 0, OBJ_GROUP = 1, OBJ_CSG = 2 };

struct Object;
using ObjectPtr = std::unique_ptr<Object>;

struct Object {
  int kind = OBJ_PRIMITIVE;
  std::vector<TransformAnimation> animations;

  // --- primitive (object.rs:357-363)
  int shape = 0;  // enum rt_shape
  double y_min = 0, y_max = 0;
  bool closed = false;
  V3 p1, p2, p3, e1, e2, n1, n2, n3;  // flat triangle: normal in n1
  Material material;
  bool has_inv = false;
  Mat4 inv = mat_identity();
  BBox bbox;  // primitive bbox / group bounding_box / csg bounding_box

  // --- group (group.rs:16-20)
  std::vector<ObjectPtr> children;
  size_t primitive_count_ = 0;

  // --- csg (csg.rs:68-74)
  int csg_op = 0;
  ObjectPtr left, right;

  // --- proxy of a device-resident mesh (include/rtcore.h rt_mesh): a group whose children (all triangles) live on the GPU.
  // It keeps the group's box and primitive count, so everything above it — animate, build_bvh, flatten — treats it as the
  // group it stands for; the matrices it receives are collected and applied on the device (mesh_pose_hook).
  int mesh_id = -1;
  bool mesh_posed = false;
  std::vector<Mat4> mesh_pending;  // Transform::transform calls since the base pose, in order (group.rs:263-270)

  ObjectPtr clone() const;
  size_t primitive_count() const;
  void transform(const Mat4& m);                               // Transform for Object (object.rs:77-85)
  void local_transform(const std::vector<LocalTransformation>& ls);  // local_transform.rs:15-20
  void set_material(const Material& m);                        // object.rs:233-239
  void add_child(ObjectPtr child);                             // group.rs:76-80
  void build_bvh();                                            // group.rs:144-167 (compact-array form)
  void build_bvh_plain();                                      // the same, as the reference writes it (cross-check)
  void into_group_with_bbox(const Material& m);                // object.rs:169-179
  void animate(double time);                                   // object.rs:108-121
  void group_animate(double time);                             // group.rs:236-241
  void recalculate_bbox();                                     // group.rs:230-235 / csg.rs:131-143
  uint32_t animation_cycle_duration_ms() const;                // object.rs:60-74
  bool includes(const Object& other) const;                    // object.rs:98-104
};
bool object_eq(const Object& a, const Object& b);

ObjectPtr make_primitive(int shape, double y_min, double y_max, bool closed, const double* tri /*18 or null*/,
                         const Material& material, const Mat4& transformation);  // PrimitiveObject::new object.rs:372-383
ObjectPtr make_group(std::vector<ObjectPtr> children);                          // ObjectGroup::new group.rs:37-43
ObjectPtr make_csg(int op, ObjectPtr left, ObjectPtr right);                     // CsgObject::new csg.rs:77-86
ObjectPtr parse_obj_model(const std::string& source, std::string& err);          // obj_model.rs:171-181

struct Light {
  V3 position, intensity;
};

// Poses the mesh behind a proxy group on the device with proxy.mesh_pending and stores the group's recalculate_bbox
// (group.rs:230-235) in proxy.bbox.  Set per thread by the C-ABI layer (scene_capi.cpp) around animate / flatten.
using MeshPoseHook = bool (*)(Object& proxy, void* user);
extern thread_local MeshPoseHook mesh_pose_hook;
extern thread_local void* mesh_pose_user;

}  // namespace rts
