// rt_reforder.h — the render core in the REFERENCE'S traversal order (validation / counting mode).
//
// BVH traversal visits nodes exactly in ObjectGroup::intersect_iter order (src/scene/object/group.rs:171-201):
// explicit stack of (group, t_enter), children in reverse, no near-first ordering, boxes behind the ray are
// entered.  The node/triangle/primitive counts of this mode are the ALGORITHMIC work of the reference
//  and its image is the in-device cross-check of the fast path (rt_fast.h) at full frame sizes.
// Selected with rt_options.reference_order = 1; not the path that is benchmarked.
#pragma once
#include "rt_shading.h"

namespace reforder {

#define RT_STACK_MAX 64      // BVH traversal stack entries per lane (validated against the scene at create time)
#define RT_FRAMES (RT_MAX_DEPTH + 1)

enum { MODE_CLOSEST = 0, MODE_SHADOW = 1 };

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

// BIG (a scene with rt_scene::general_csg): the general evaluation and its longer list (rt_shapes.h).
template <int MODE, bool STATS, bool BIG = false>
__device__ __forceinline__ void csg_intersect(const DevScene& s, uint32_t csg_root, d3 o, d3 d, Coll& c, Counters& cnt) {
  CsgHit list[BIG ? RT_CSG_BIG_LIST : RT_CSG_LIST_MAX];
  int n = BIG ? csg_survivors_general<STATS>(s, csg_root, o, d, list, cnt) : csg_survivors<STATS>(s, csg_root, o, d, list, cnt);
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
// `bound` (closest mode, the re-trace of rt_fast.h's coincident-surface rays): groups entered later than this are not
// visited.  The caller knows the closest intersection's t up to the tie window (bound = near_limit(that t)); every
// primitive of a group lies inside the group's box, so a group entered after `bound` holds only intersections beyond it:
// none can be the hit, and none can make the reference skip a group that is entered before `bound` (group.rs:182 skips
// when t_enter >= hit_time, and a hit_time such a primitive sets is > bound).  The groups entered up to `bound` are
// therefore visited, skipped and intersected exactly as in the unbounded walk, and the hit is the same; what is saved is
// the rest of the scene behind it, which the reference walks in tree order, not front to back.
template <int MODE, bool STATS, bool BIG = false>
__device__ __forceinline__ void traverse(const DevScene& s, d3 o, d3 d, Coll& c, Counters& cnt, double bound = INFINITY) {
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
    if (!bbox_time(bb, o, dinv, t) || t > bound) return;
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
        if (bbox_time(bb, o, dinv, t) && !(t > bound)) {
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
        int n = prim_roots<STATS>(s, info, o, d, ts, u, v, cnt);
        if (n) coll_add_roots<MODE>(s, c, idx, info.y, n, ts, u, v);
      } else {
        csg_intersect<MODE, STATS, BIG>(s, idx, o, d, c, cnt);
      }
    }
    if (MODE == MODE_SHADOW && c.intensity >= 1.) return;  // min(intensity, 1) is already 1 (intersection.rs:124-126)
  }
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

template <bool STATS, bool BIG = false>
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
      traverse<MODE_CLOSEST, STATS, BIG>(s, o, d, c, cnt);
#ifdef RT_EMU_TRACE
      printf("ref   depth %d hit prim %u t %.17g o %.17g %.17g %.17g d %.17g %.17g %.17g | pa %u ta %.17g pb %u tb %.17g odd %d\n", depth, c.hit_prim, c.limit, o.x, o.y, o.z, d.x, d.y, d.z, c.pa, c.ta, c.pb, c.tb, (int)c.hit_neg_odd);
#endif
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
        traverse<MODE_SHADOW, STATS, BIG>(s, g.over, sdir, sc, cnt);
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

}  // namespace reforder
