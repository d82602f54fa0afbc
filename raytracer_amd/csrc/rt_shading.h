// rt_shading.h — normals, patterns, Phong, Schlick, hit geometry and the camera ray, f64, reference operation order.
// Shared by the fast traversal (rt_fast.h) and the reference-order traversal (rt_reforder.h).
#pragma once
#include "rt_shapes.h"

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
// `surface_color` = pattern_color(s, info, point): the same for every light of a point, so a caller that folds over the lights
// evaluates it once.
__device__ __forceinline__ d3 lighting_of(const DevScene& s, uint32_t light, d3 surface_color, const double* __restrict__ mat, d3 point,
                                          d3 eye, d3 normal, double shadow) {
  const double* li = s.light_intensity + (size_t)light * 3;
  const double* lp = s.light_pos + (size_t)light * 3;
  d3 intensity = mk3(li[0], li[1], li[2]);
  d3 effective = cmul3(surface_color, intensity);
  d3 light_v = normalize3(sub3(mk3(lp[0], lp[1], lp[2]), point));
  double ka = mat[RT_MAT_AMBIENT];
  d3 ambient = mk3(effective.x * ka, effective.y * ka, effective.z * ka);
  double ldn = dot3(light_v, normal);
  // (testing the shadow value, or stream::lit_bits' verdict, before the light vector is normalised was measured SLOWER in the
  // combine kernel, 2.10 -> 2.48 ms: the normalisation runs while the shadow value is still on its way)
  if (approx_eq(shadow, 1.) || ldn < 0.) {  // ambient + black + black
    return mk3(ambient.x + 0. + 0., ambient.y + 0. + 0., ambient.z + 0. + 0.);
  }
  double kd = mat[RT_MAT_DIFFUSE];
  d3 diffuse = mk3(effective.x * kd * ldn, effective.y * kd * ldn, effective.z * kd * ldn);
  d3 reflect_v = reflect3(neg3(light_v), normal);
  double rde = dot3(reflect_v, eye);
  d3 specular = mk3(0., 0., 0.);
  if (sign_positive(rde)) {
    double ks = mat[RT_MAT_SPECULAR], shininess = mat[RT_MAT_SHININESS];
    // A material without specular reflection (ks = 0) needs no pow: for rde in (0, 1] and shininess >= 0 the factor is a
    // finite number in [0, 1], and (intensity * 0) * factor is the same signed zero whatever it is.  (rde > 1 by rounding, a
    // negative or NaN shininess: the factor can be inf or NaN, and 0 * that is NaN — those take the pow.)
    double factor = (ks == 0. && rde <= 1. && shininess >= 0.) ? 1. : pow(rde, shininess);
    specular = mk3(intensity.x * ks * factor, intensity.y * ks * factor, intensity.z * ks * factor);
  }
  double k = 1. - shadow;
  return mk3(ambient.x + (diffuse.x + specular.x) * k, ambient.y + (diffuse.y + specular.y) * k,
             ambient.z + (diffuse.z + specular.z) * k);
}
__device__ __forceinline__ d3 lighting(const DevScene& s, uint32_t light, uint4 info, const double* __restrict__ mat, d3 point,
                                       d3 eye, d3 normal, double shadow) {
  return lighting_of(s, light, pattern_color(s, info, point), mat, point, eye, normal, shadow);
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
