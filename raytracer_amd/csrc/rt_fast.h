// rt_fast.h — BVH traversal (fast::trav_run, what every production kernel traces rays with) and the per-pixel render core
// as ONE kernel (the single-kernel pipeline: comparison path, overflow fallback of the streamed pipeline, rt_stream.h).
//
// Single kernel: one wavefront (64 lanes) renders one 8x8 pixel tile; each lane owns one pixel and runs the whole
// Renderer::color_at_pixel (src/render/renderer.rs:50-66) for it.
//
//  * The reflect/refract recursion of Integrator::shade_hit_at_depth (src/shading/integrator.rs:199-228) is a
//    per-lane state machine over an explicit frame stack (post-order, so every sum is formed in the reference's
//    order).  EVERY ray of the machine — primary, reflected, refracted, shadow, and the refractive-container walk —
//    goes through ONE traversal loop, so lanes that are at different points of the recursion still traverse together.
//  * Traversal is a while-while loop over 128-B four-wide nodes (the boxes of up to four children per fetch, f32, rounded
//    outward), tested conservatively in f32 (box_test_f32), nearest child first (shadow rays: slot order), with the
//    per-lane stack in LDS (RT_LDS_STACK entries, deeper entries spill to scratch).  The other children are pushed with a
//    lower bound of their entry time and dropped at pop time once the closest hit is nearer.
//  * What makes a different visiting order legal: the closest hit of the reference is "smallest t, first visited wins
//    ties" (intersection.rs:65-77, strict <).  The visit order of ObjectGroup::intersect_iter (group.rs:171-201) is a
//    fixed pre-order of the tree, so "first visited" = smallest prim_rank, which is what ties are resolved with here.
//    Boxes are culled only when no accepted intersection can lie inside them (t_enter > closest t, or box entirely
//    behind the origin), so the set of candidate intersections is the reference's.  Negative-t intersections, which
//    the reference collects on the same walk for the refractive-index containers (intersection.rs:258-302), are
//    collected by a second walk over the boxes behind the origin, and only when the hit is transparent.
#pragma once
#include "rt_reforder.h"

namespace fast {

#ifndef RT_LDS_STACK
#define RT_LDS_STACK 16                                   // per-lane stack entries in LDS (8 KB per wavefront)
#endif
#ifndef RT_OVF_STACK
#define RT_OVF_STACK 48                                   // further entries in scratch
#endif
#define RT_FAST_STACK_MAX (RT_LDS_STACK + RT_OVF_STACK)  // validated against the tree depth at scene creation
#define RT_FRAMES (RT_MAX_DEPTH + 1)

enum { M_CLOSEST = 0, M_SHADOW = 1, M_CONTAINERS = 2 };

// The set of modes a kernel's lanes can be in (bit m = mode m), so that a kernel which only ever traces shadow rays, or
// never does, carries no code or registers for the other collectors.  MS_ALL: the mode is a per-lane run-time value.
enum { MS_ALL = 7, MS_CLOSEST_CONTAINERS = 5, MS_SHADOW = 2, MS_CLOSEST = 1 };

// The ray being traversed and its collector (IntersectionCollector, intersection.rs:17-135, without the Vec).
struct Trav {
  d3 o, d;
  double limit;   // closest: t of the best hit so far; shadow: distance to the light; containers: 0 (only boxes entered at t <= 0)
  double t2;      // closest: smallest t of an intersection with any OTHER primitive than the hit (the runner-up)
  double a, b;    // closest: u, v of the hit; shadow: a = accumulated intensity; containers: largest negative t of the two innermost containers
  uint32_t p, q;  // closest: p = hit primitive; containers: the two innermost containers
  uint32_t h;     // containers: the primitive that was hit (closest pass)
  uint32_t flag;  // containers: the hit primitive has an odd number of negative-t intersections
  uint32_t mode;
  uint32_t odd;   // 1: a component of dinv is not finite (an axis-parallel ray): 0 * inf = NaN can occur in the slab test,
                  // whose outcome then depends on the reference's operand order (see box_times)
  uint32_t cur;   // reference being visited, RT_FREF_NONE when nothing is left to visit
  int sp;
  // The ray in f32, for the conservative slab test of the node step (box_keys_f32): t_f = fma(b, fdi, fno) with
  // fdi ~ 1 / d and fno ~ -o / d, per axis, both computed in f32; ferr bounds what their errors can move t_f by (absolute
  // part; the relative part is RT_SLAB_REL); limf = an f32 no less than near_limit(limit).  The exact dir_inv of Ray::new
  // (ray.rs:27-34) is not kept: only the slab test of axis-parallel rays and the cube read it, and they divide themselves.
  float fdx, fdy, fdz, fnx, fny, fnz, ferr, limf;
};

// Where the reference can differ from "smallest t, ties to the smallest rank": it skips a popped group when
// t_enter >= closest t so far (group.rs:182).  A primitive lying ON a face of its group's box has t within rounding of
// that t_enter, so when a SECOND surface coincides with it (dragons.yml: the glass cubes stand on the pedestal caps) the
// reference may never test the nearer of the two.  Such a ray is recognised by a runner-up intersection within
// RT_TIE_WINDOW of the hit and is re-traced in the reference's own order (reforder::traverse); for every other ray the
// candidate sets are identical and so is the hit.  Boxes are therefore kept while they are entered within the window.
#define RT_TIE_WINDOW 1.0e-10
__device__ __forceinline__ double near_limit(double limit) { return limit + RT_TIE_WINDOW * fmax(1., limit); }

#ifdef RT_HOST_EMU
__device__ __forceinline__ float rcp_f32(float x) { return 1.0f / x; }
#else
__device__ __forceinline__ float rcp_f32(float x) { return __builtin_amdgcn_rcpf(x); }  // v_rcp_f32: 1 ulp
#endif

// An f32 no less than x (x > 0 or +inf): the nearest f32 moved up by at least its last bit.
__device__ __forceinline__ float f32_above(double x) {
  float f = (float)x;
  return __builtin_fmaf(fabsf(f), 1.1920929e-07f, f);
}

// Relative and absolute slack of the f32 slab test (see box_key_f32).  With u = 2^-24 (f32 unit roundoff): fdi = rcp((float)d)
// is within 3u of 1 / d (one rounding of d, v_rcp_f32's 1 ulp <= 2u), fno = -(float)o * fdi within 5u of -o / d, and the fma
// rounds once more, so t_f = fma(b, fdi, fno) differs from the exact (b - o) / d by at most 4u |t| + 8u |o / d| (using
// |b / d| <= |t| + |o / d|).  Both parts are taken twice over, which also covers the roundings of the slack arithmetic
// itself.
#define RT_SLAB_REL 4.7683716e-07f  // 8u
#define RT_SLAB_ABS 9.5367432e-07f  // 16u, times max |o / d|
#define RT_SLAB_ABS_MIN 1.0e-30f    // floor of the absolute part (o = 0 makes it 0 otherwise)

__device__ __forceinline__ void trav_start(Trav& T, d3 o, d3 d, uint32_t mode, double limit) {
  T.o = o;
  T.d = d;
  T.limit = limit;
  T.mode = mode;
  {
    T.fdx = rcp_f32((float)d.x);
    T.fdy = rcp_f32((float)d.y);
    T.fdz = rcp_f32((float)d.z);
    T.fnx = -(float)o.x * T.fdx;
    T.fny = -(float)o.y * T.fdy;
    T.fnz = -(float)o.z * T.fdz;
    float m = fmaxf(fmaxf(fabsf(T.fnx), fabsf(T.fny)), fabsf(T.fnz));
    T.ferr = fmaxf(m * RT_SLAB_ABS, RT_SLAB_ABS_MIN);
    T.limf = f32_above(near_limit(limit));
    // the f32 test needs all six finite (an axis-parallel ray has 1 / d = inf, and 0 * inf = NaN there is decided by the
    // reference's operand order: such rays take the exact f64 test).  (A component of d below f32's range counts as 0.)
    // (below 1e18 rather than merely finite: a grid step of up to 1e16 times fdi must not overflow, box coordinates being
    // at most 1e18, rt_scene_create)
    const float big = 1.0e18f;
    bool fin = fabsf(T.fdx) < big && fabsf(T.fdy) < big && fabsf(T.fdz) < big && m < big;
    T.odd = fin ? 0u : 1u;
  }
  T.cur = RT_MAKE_FREF(RT_FREF_INNER, 0, 0);
  T.sp = 0;
  if (mode == M_CLOSEST) {
    T.p = RT_NONE;
    T.a = T.b = -INFINITY;
    T.t2 = INFINITY;
  } else if (mode == M_SHADOW) {
    T.a = 0.;
  } else {
    T.a = T.b = -INFINITY;
    T.p = T.q = RT_NONE;
    T.flag = 0;
  }
}

template <int MS>
__device__ __forceinline__ bool in_mode(const Trav& T, uint32_t m) {
  if (!((MS >> m) & 1)) return false;
  if (MS == (1 << m)) return true;
  return T.mode == m;
}

// All roots of ONE primitive, in the order the reference add()s them (intersection.rs:65-109).
template <bool STATS, int MS = MS_ALL>
__device__ __forceinline__ void add_roots(const DevScene& s, Trav& T, uint32_t prim, uint32_t material, int n, const double* ts,
                                          double u, double v) {
  if (in_mode<MS>(T, M_CLOSEST)) {
    for (int i = 0; i < n; i++) {
      double t = ts[i];
      if (!sign_positive(t)) continue;
      bool take = t < T.limit;
      if (!take && t == T.limit && T.p != RT_NONE && T.p != prim) take = s.prim_rank[prim] < s.prim_rank[T.p];
      if (T.p != prim) T.t2 = fmin(T.t2, take ? T.limit : t);  // the dethroned hit, or this rejected one
      if (take) {
        T.limit = t;
        T.limf = f32_above(near_limit(t));
        T.p = prim;
        T.a = u;
        T.b = v;
      }
    }
  } else if (in_mode<MS>(T, M_SHADOW)) {
    double transparency = 0.;
    bool loaded = false;
    for (int i = 0; i < n; i++) {
      double t = ts[i];
      if (sign_positive(t) && !approx_eq(t, T.limit) && t < T.limit && T.a < 1.) {  // intersection.rs:78-87
        if (!loaded) {
          // (an opaque material is known from the leaf's own record: an occluder's hit costs no dependent fetches — the
          // prim_info -> material chain stalled the whole wavefront twice per hit: 15.35 -> 14.92 ms for the 4K frame's shadow rays.
          // Carrying a transparent material's value in the record as well, and the closest hit's "is glass" bit, gained nothing)
          if (material != RT_MAT_OPAQUE) transparency = s.material[(size_t)material * RT_MAT_STRIDE + RT_MAT_TRANSPARENCY];
          loaded = true;
        }
        T.a += 1. - transparency;
      }
    }
  } else {
    int neg = 0;
    double maxneg = -INFINITY;
    for (int i = 0; i < n; i++) {
      double t = ts[i];
      if (!sign_positive(t)) {
        neg++;
        maxneg = fmax(maxneg, t);
      }
    }
    if (neg & 1) {
      // innermost containers = odd-count objects with the largest negative t; among equal t the one the reference
      // visits later sorts later (stable sort of the collector's vec), i.e. the larger rank
      uint32_t r = s.prim_rank[prim];
      bool over_a = maxneg > T.a || (maxneg == T.a && (T.p == RT_NONE || r >= s.prim_rank[T.p]));
      if (over_a) {
        T.b = T.a;
        T.q = T.p;
        T.a = maxneg;
        T.p = prim;
      } else {
        bool over_b = maxneg > T.b || (maxneg == T.b && (T.q == RT_NONE || r >= s.prim_rank[T.q]));
        if (over_b) {
          T.b = maxneg;
          T.q = prim;
        }
      }
    }
    if (prim == T.h) T.flag = (uint32_t)(neg & 1);
  }
}

// CsgObject::intersect (csg.rs:112-122): survivors go to the collector in ascending t.
template <bool STATS, int MS = MS_ALL>
__device__ __forceinline__ void add_csg(const DevScene& s, Trav& T, uint32_t csg_root, Counters& cnt) {
  CsgHit list[RT_CSG_LIST_MAX];
  int n = csg_survivors<STATS>(s, csg_root, T.o, T.d, list, cnt);
  if (in_mode<MS>(T, M_CONTAINERS)) {
    // negative-t survivors, aggregated per leaf object (they all come from this one call)
    for (int i = 0; i < n; i++) {
      if (sign_positive(list[i].t)) continue;
      bool first = true;
      for (int j = 0; j < i; j++)
        if (!sign_positive(list[j].t) && list[j].prim == list[i].prim) first = false;
      if (!first) continue;
      double ts[RT_CSG_LIST_MAX];
      int k = 0;
      for (int j = i; j < n; j++)
        if (!sign_positive(list[j].t) && list[j].prim == list[i].prim) ts[k++] = list[j].t;
      add_roots<STATS, MS>(s, T, list[i].prim, 0, k, ts, 0., 0.);
    }
  } else {
    for (int i = 0; i < n; i++) {
      uint32_t prim = list[i].prim;
      uint32_t material = in_mode<MS>(T, M_SHADOW) ? s.prim_info[prim].y : 0u;
      add_roots<STATS, MS>(s, T, prim, material, 1, &list[i].t, list[i].u, list[i].v);
    }
  }
}

// The same for a CSG object of two analytic primitives with at most two roots each (DevCsg2Rec): both operands' roots from
// ONE record, the list of at most four in registers.  The reference collects the left operand's roots, then the right's,
// sorts by t (stable) and filters (csg.rs:87-122): here every root's place in that order is counted directly (the roots
// before it: smaller t, or equal t and collected earlier).  A NaN root has no place in an order: such a ray takes add_csg.
template <bool STATS, int MS = MS_ALL>
__device__ __forceinline__ void add_csg2(const DevScene& s, Trav& T, uint32_t csg_root, uint32_t rec, Counters& cnt) {
  const DevCsg2Rec* r = s.csg2_rec + rec;
  const uint32_t lflags = r->left.flags, rflags = r->right.flags, op = lflags >> RT_CSG2_OP_SHIFT;
  double tl[4], tr[4], ul, vl, ur, vr;
  const int nl = analytic_roots<STATS>(r->left.shape, lflags, r->left.inv, r->left.y_min, r->left.y_max, T.o, T.d, tl, ul, vl, cnt);
  const int nr = analytic_roots<STATS>(r->right.shape, rflags, r->right.inv, r->right.y_min, r->right.y_max, T.o, T.d, tr, ur, vr, cnt);
  // the collection order: left roots, then right roots
  const double t[4] = {tl[0], tl[1], tr[0], tr[1]};
  const bool valid[4] = {nl > 0, nl > 1, nr > 0, nr > 1};
  bool nan = false;
  for (int i = 0; i < 4; i++) nan = nan || (valid[i] && t[i] != t[i]);
  if (nan || nl > 2 || nr > 2) {  // (more than two roots cannot happen: rt_scene_create only records such operands)
    add_csg<STATS, MS>(s, T, csg_root, cnt);
    return;
  }
  int place[4];
  for (int i = 0; i < 4; i++) {
    place[i] = 0;
    for (int j = 0; j < 4; j++)
      if (j != i && valid[j] && (t[j] < t[i] || (t[j] == t[i] && j < i))) place[i]++;
  }
  const int n = nl + nr;
  // the sorted list, filtered (csg.rs:87-111): position p holds the root whose place is p
  double st[4];
  bool sleft[4], keep[4];
  bool inside_left = false, inside_right = false;
  for (int p = 0; p < 4; p++) {
    st[p] = 0.;
    sleft[p] = false;
    for (int i = 0; i < 4; i++)
      if (valid[i] && place[i] == p) {
        st[p] = t[i];
        sleft[p] = i < 2;
      }
    keep[p] = false;
    if (p < n) {
      const bool hit_left = sleft[p];
      if (hit_left) inside_left = !inside_left;
      else inside_right = !inside_right;
      bool allowed;
      if (op == RT_CSG_UNION) allowed = hit_left ? !inside_right : !inside_left;
      else if (op == RT_CSG_INTERSECTION) allowed = hit_left ? inside_right : inside_left;
      else allowed = hit_left ? !inside_right : inside_left;
      keep[p] = allowed;
    }
  }
  const uint32_t lprim = r->left.ref, rprim = r->right.ref;
  if (in_mode<MS>(T, M_CONTAINERS)) {
    // negative-t survivors, aggregated per leaf object, the object whose first one comes first in the list first (add_csg)
    double nl_t[2] = {0., 0.}, nr_t[2] = {0., 0.};
    int kl = 0, kr = 0;
    bool left_first = false, any = false;
    for (int p = 0; p < 4; p++) {
      if (!keep[p] || sign_positive(st[p])) continue;
      if (!any) left_first = sleft[p];
      any = true;
      if (sleft[p]) {
        if (kl == 0) nl_t[0] = st[p];
        else nl_t[1] = st[p];
        kl++;
      } else {
        if (kr == 0) nr_t[0] = st[p];
        else nr_t[1] = st[p];
        kr++;
      }
    }
    if (left_first) {
      if (kl) add_roots<STATS, MS>(s, T, lprim, 0, kl, nl_t, 0., 0.);
      if (kr) add_roots<STATS, MS>(s, T, rprim, 0, kr, nr_t, 0., 0.);
    } else {
      if (kr) add_roots<STATS, MS>(s, T, rprim, 0, kr, nr_t, 0., 0.);
      if (kl) add_roots<STATS, MS>(s, T, lprim, 0, kl, nl_t, 0., 0.);
    }
  } else {
    const uint32_t lmat = in_mode<MS>(T, M_SHADOW) ? ((lflags & RT_LEAFREC_OPAQUE) ? RT_MAT_OPAQUE : r->left.material) : 0u;
    const uint32_t rmat = in_mode<MS>(T, M_SHADOW) ? ((rflags & RT_LEAFREC_OPAQUE) ? RT_MAT_OPAQUE : r->right.material) : 0u;
    for (int p = 0; p < 4; p++) {
      if (!keep[p]) continue;
      const bool L = sleft[p];
      add_roots<STATS, MS>(s, T, L ? lprim : rprim, L ? lmat : rmat, 1, &st[p], L ? ul : ur, L ? vl : vr);
    }
  }
}

// BoundingBox::intersection_time (bounding_box.rs:192-225) + the exit time.  NaN (0 * inf) is dropped by fmax/fmin like
// f64::max/min; the reference's early return after x,y is implied by the final test.
__device__ __forceinline__ bool box_times(const double* __restrict__ bb, d3 o, d3 dinv, double& tmin, double& tmax) {
  double xmin, xmax, ymin, ymax, zmin, zmax;
  axis_times(o.x, dinv.x, bb[0], bb[3], xmin, xmax);
  axis_times(o.y, dinv.y, bb[1], bb[4], ymin, ymax);
  axis_times(o.z, dinv.z, bb[2], bb[5], zmin, zmax);
  tmin = fmax(fmax(xmin, ymin), zmin);
  tmax = fmin(fmin(xmax, ymax), zmax);
  return !(tmin > tmax);
}

#ifdef RT_HOST_EMU
__device__ __forceinline__ double min_f64(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ double max_f64(double a, double b) { return fmax(a, b); }
#else
// v_min_f64 / v_max_f64 as they are (the compiler's fmin/fmax first quiets each operand with an extra v_max_f64 x, x)
__device__ __forceinline__ double min_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double max_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#endif

// Keeps the compiler from sinking these loads below a later branch: they are all issued, and waited for, here.
__device__ __forceinline__ void pin_loaded(double2& a, double2& b, double2& c, double2& d, double2& e) {
#ifndef RT_HOST_EMU
  asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y), "+v"(c.x), "+v"(c.y), "+v"(d.x), "+v"(d.y), "+v"(e.x), "+v"(e.y));
#endif
}

// An f32 no greater than t (and finite for every t a visited box can have): the nearest f32 moved down by its last bit.
__device__ __forceinline__ float key_below(double t) {
  float k = (float)t;
  return fminf(fmaf(-fabsf(k), 1.1920929e-07f, k), 3.4028234e38f);
}

// ---------------------------------------------------------------------------------------------
// The node step's box test.
//
// A box only ever CULLS: which intersection wins is decided by the exact f64 primitive tests, the visit ranks and the tie
// re-trace, so the test may keep any box the reference's f64 slab test (bounding_box.rs:192-225) keeps and more.  The
// node's boxes are already the reference's boxes rounded outward to f32; for a ray without an infinite dinv component
// the slab test itself is done in f32 on INTERVALS: per axis t_f = fma(b, fdi, fno) for both planes, entry = max over the
// axes of the smaller, exit = min of the larger (no rounding in min / max), and then
//     lo = entry - RT_SLAB_REL |entry| - ferr   <=  the exact entry time   (Trav::ferr, RT_SLAB_REL: error bound of t_f)
//     hi = exit  + RT_SLAB_REL |exit|  + ferr   >=  the exact exit time
// (x -> x -+ rel |x| -+ err is monotone, so it commutes with the min / max over axes and planes).  The box is visited
// unless max(lo, floor) > min(hi, limf): floor = 0 for closest-hit and shadow rays, which also culls the boxes entirely
// behind the origin (hi < 0) — their intersections have t < 0 —, and -inf for the container walk, which wants exactly
// those.  Every comparison is written so that a NaN (inf - inf on absurd coordinates) keeps the box.
// What the f32 test costs: 14 VALU instructions per box instead of 30 (no f32 -> f64 conversions, fma instead of sub +
// mul, v_min3 / v_max3), and its key is already the f32 lower bound the stack keeps.
// key: >= RT_KEY_MIN (so that (key, ref) read as a u64 / f64 is a normal, ordered number), +inf = do not visit.
// ---------------------------------------------------------------------------------------------
#define RT_KEY_MIN 1.17549435e-38f
#define RT_KEY_NONE 0x7F800000u  // bits of +inf

// (v_pk_fma_f32 for the six fmas — two planes of an axis per instruction — was measured slower: 36.5 against 35.4 ms for the
// 4K frame; the packed instruction saves no issue time on this part and its register pairs cost the shadow kernel spills)
__device__ __forceinline__ bool box_test_f32(const Trav& T, const uint32_t* __restrict__ raw, int c, float floorv, uint32_t ref, float& lo) {
  const float b0 = __uint_as_float(raw[6 * c + 0]), b1 = __uint_as_float(raw[6 * c + 1]), b2 = __uint_as_float(raw[6 * c + 2]);
  const float b3 = __uint_as_float(raw[6 * c + 3]), b4 = __uint_as_float(raw[6 * c + 4]), b5 = __uint_as_float(raw[6 * c + 5]);
  const float x0 = __builtin_fmaf(b0, T.fdx, T.fnx), x1 = __builtin_fmaf(b3, T.fdx, T.fnx);
  const float y0 = __builtin_fmaf(b1, T.fdy, T.fny), y1 = __builtin_fmaf(b4, T.fdy, T.fny);
  const float z0 = __builtin_fmaf(b2, T.fdz, T.fnz), z1 = __builtin_fmaf(b5, T.fdz, T.fnz);
  const float entry = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
  const float exit_ = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
  lo = __builtin_fmaf(-fabsf(entry), RT_SLAB_REL, entry) - T.ferr;
  const float hi = __builtin_fmaf(fabsf(exit_), RT_SLAB_REL, exit_) + T.ferr;
  return !(fmaxf(lo, floorv) > fminf(hi, T.limf)) & (ref != RT_FREF_NONE);
}
__device__ __forceinline__ uint32_t box_key_f32(const Trav& T, const uint32_t* __restrict__ raw, int c, float floorv, uint32_t ref) {
  float lo;
  const bool visit = box_test_f32(T, raw, c, floorv, ref, lo);
  return visit ? __float_as_uint(fmaxf(lo, RT_KEY_MIN)) : RT_KEY_NONE;
}

// The exact test (the reference's own arithmetic on the node's boxes) for wavefronts holding an axis-parallel ray.
template <int MS>
__device__ __forceinline__ uint32_t box_key_f64(const Trav& T, const uint32_t* __restrict__ raw, int c, bool all_t, uint32_t ref) {
  double bb[6], tmin, tmax;
  for (int k = 0; k < 6; k++) bb[k] = (double)__uint_as_float(raw[6 * c + k]);
  bool hit = box_times(bb, T.o, mk3(1. / T.d.x, 1. / T.d.y, 1. / T.d.z), tmin, tmax);  // dir_inv of Ray::new (ray.rs:27-34)
  // a box can hold an accepted intersection only if it is entered no later than the limit (plus the window in which a
  // runner-up must still be seen, see near_limit) and, for t >= 0 intersections, is not entirely behind the origin
  bool h = hit & !(tmin > near_limit(T.limit)) & (all_t | !(tmax < 0.)) & (ref != RT_FREF_NONE);
  return h ? __float_as_uint(fmaxf(key_below(tmin), RT_KEY_MIN)) : RT_KEY_NONE;
}

#if RT_QNODE
// The same two tests on a quantised node (rt_device.h): box coordinate = org + q * step per axis, so
//     t_f = fma(q, A, B)   with   A = step * fdi (exact: step is a power of two),   B = fma(org, fdi, fno)   per node and axis.
// B carries the roundings of fdi and fno scaled by the origin plane's distance: 3u |org / d| + 5u |o / d| + u |B|, and
// |org / d| <= |B| + |o / d|; with the 3u (|t| + |B|) of q A and the fma's own u |t| that is 4u |t| + 7u |B| + 8u |o / d|
// in all: the relative and the per-ray slack as before, plus RT_SLAB_NODE max |B| per node (all taken twice over).
#define RT_SLAB_NODE 9.5367432e-07f  // 16u
struct QNodeF32 {
  float ax, ay, az, bx, by, bz, err;
};
__device__ __forceinline__ float grid_step(uint32_t exps, int axis) { return __uint_as_float(((exps >> (8 * axis)) & 0xFFu) << 23); }
__device__ __forceinline__ QNodeF32 qnode_f32(const Trav& T, uint4 head) {
  QNodeF32 n;
  n.ax = grid_step(head.w, 0) * T.fdx;
  n.ay = grid_step(head.w, 1) * T.fdy;
  n.az = grid_step(head.w, 2) * T.fdz;
  n.bx = __builtin_fmaf(__uint_as_float(head.x), T.fdx, T.fnx);
  n.by = __builtin_fmaf(__uint_as_float(head.y), T.fdy, T.fny);
  n.bz = __builtin_fmaf(__uint_as_float(head.z), T.fdz, T.fnz);
  n.err = __builtin_fmaf(fmaxf(fmaxf(fabsf(n.bx), fabsf(n.by)), fabsf(n.bz)), RT_SLAB_NODE, T.ferr);
  return n;
}
__device__ __forceinline__ float qcoord(uint32_t word, int c) { return (float)((word >> (8 * c)) & 0xFFu); }  // v_cvt_f32_ubyteN
__device__ __forceinline__ uint32_t qbox_key_f32(const Trav& T, const QNodeF32& n, const uint32_t* __restrict__ qc, int c, float floorv,
                                                 uint32_t ref) {
  const float x0 = __builtin_fmaf(qcoord(qc[0], c), n.ax, n.bx), x1 = __builtin_fmaf(qcoord(qc[3], c), n.ax, n.bx);
  const float y0 = __builtin_fmaf(qcoord(qc[1], c), n.ay, n.by), y1 = __builtin_fmaf(qcoord(qc[4], c), n.ay, n.by);
  const float z0 = __builtin_fmaf(qcoord(qc[2], c), n.az, n.bz), z1 = __builtin_fmaf(qcoord(qc[5], c), n.az, n.bz);
  const float entry = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
  const float exit_ = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
  const float lo = __builtin_fmaf(-fabsf(entry), RT_SLAB_REL, entry) - n.err;
  const float hi = __builtin_fmaf(fabsf(exit_), RT_SLAB_REL, exit_) + n.err;
  const bool visit = !(fmaxf(lo, floorv) > fminf(hi, T.limf)) & (ref != RT_FREF_NONE);
  return visit ? __float_as_uint(fmaxf(lo, RT_KEY_MIN)) : RT_KEY_NONE;
}
template <int MS>
__device__ __forceinline__ uint32_t qbox_key_f64(const Trav& T, uint4 head, const uint32_t* __restrict__ qc, int c, bool all_t, uint32_t ref) {
  double bb[6], tmin, tmax;
  for (int k = 0; k < 6; k++)  // org + q * step: exact in f64 (quantize_node)
    bb[k] = (double)__uint_as_float(k % 3 == 0 ? head.x : (k % 3 == 1 ? head.y : head.z)) +
            (double)((qc[k] >> (8 * c)) & 0xFFu) * (double)grid_step(head.w, k % 3);
  bool hit = box_times(bb, T.o, mk3(1. / T.d.x, 1. / T.d.y, 1. / T.d.z), tmin, tmax);  // dir_inv of Ray::new (ray.rs:27-34)
  bool h = hit & !(tmin > near_limit(T.limit)) & (all_t | !(tmax < 0.)) & (ref != RT_FREF_NONE);
  return h ? __float_as_uint(fmaxf(key_below(tmin), RT_KEY_MIN)) : RT_KEY_NONE;
}
#endif

// A stack entry: reference in the low word, key (f32 bits, a lower bound of the entry time) in the high word.
__device__ __forceinline__ double pack_entry(uint32_t ref, uint32_t key_bits) {
  return __longlong_as_double((long long)(((unsigned long long)key_bits << 32) | ref));
}
__device__ __forceinline__ uint32_t entry_ref(double e) { return (uint32_t)(unsigned long long)__double_as_longlong(e); }
__device__ __forceinline__ uint32_t entry_key(double e) { return (uint32_t)((unsigned long long)__double_as_longlong(e) >> 32); }

__device__ __forceinline__ void stack_push(Trav& T, uint2* lds, uint2* ovf, double entry) {
  // (the two branches store differently shaped values on purpose: identical stores get merged into one FLAT store through
  // a selected pointer, which takes the slow path through both the LDS and the memory pipeline — on every push)
  if (T.sp < RT_LDS_STACK) {
    *reinterpret_cast<double*>(&lds[T.sp * 64]) = entry;
  } else {
    int i = min(T.sp - RT_LDS_STACK, RT_OVF_STACK - 1);  // cannot run over: rt_scene_create rejects a tree that needs more
    ovf[i].x = entry_ref(entry);                         // than RT_FAST_STACK_MAX entries (the clamp only keeps a bug inside
    ovf[i].y = entry_key(entry);                         // the array)
  }
  T.sp++;
}

// The push without a branch: the entry is written to the slot above the top either way and only counted when `valid`.
// The caller has checked (for the whole wavefront) that the slot is one of the LDS part.
#ifndef RT_PUSH_BRANCHLESS
#define RT_PUSH_BRANCHLESS 1
#endif
__device__ __forceinline__ void stack_push_if(Trav& T, uint2* lds, double entry, bool valid) {
  *reinterpret_cast<double*>(&lds[T.sp * 64]) = entry;
  T.sp += valid ? 1 : 0;
}

__device__ __forceinline__ uint32_t stack_pop(Trav& T, const uint2* lds, const uint2* ovf) {
  while (T.sp > 0) {
    T.sp--;
    uint2 e = lds[(T.sp < RT_LDS_STACK ? T.sp : RT_LDS_STACK - 1) * 64];
    if (T.sp >= RT_LDS_STACK) {  // volatile: keeps this a scratch load (otherwise the two loads are merged into one flat load
      const volatile uint2* v = ovf;  // through a selected pointer, which waits on both memory counters)
      e.x = v[T.sp - RT_LDS_STACK].x;
      e.y = v[T.sp - RT_LDS_STACK].y;
    }
    if (!(__uint_as_float(e.y) > T.limf)) return e.x;  // still able to hold an accepted intersection
  }
  return RT_FREF_NONE;
}

__device__ __forceinline__ bool first_active_lane() {
  return (uint32_t)(__ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) == (threadIdx.x & 63u);
}
__device__ __forceinline__ void count_step(Counters& cnt) {  // statistics builds only
  unsigned long long active = __builtin_amdgcn_ballot_w64(true);
  cnt.lane_steps++;
  if ((uint32_t)(__ffsll((long long)active) - 1) == (threadIdx.x & 63u)) cnt.wave_steps++;
}

__device__ __forceinline__ bool trav_busy(const Trav& T) { return T.cur != RT_FREF_NONE; }

// Shadow rays do not need the nearest child first (every intersection before the light counts, in any order);
// RT_SHADOW_SORT = 0 visits a node's children in slot order instead.
#ifndef RT_SHADOW_SORT
#define RT_SHADOW_SORT 0  // measured: 18.90 -> 18.40 ms for the 4K frame's shadow rays without the sort
#endif

// The node loop is left early when fewer than RT_NODE_BREAK lanes are still descending while other lanes of the wavefront
// already hold a leaf: those lanes are served first and the descent resumes afterwards (0: every lane descends to a leaf
// before any leaf is done).  Measured on the 4K frame: 8-16 equally good (38.95 -> 37.0 ms), 20 less so, 28 none.
#ifndef RT_NODE_BREAK
#define RT_NODE_BREAK 16
#endif
#ifndef RT_COUNT_PHASE
#define RT_COUNT_PHASE 0  // statistics builds: what wave_steps / lane_steps count.  0: node steps and leaf visits (the documented
#endif                    // meaning); dev builds: 1 node steps only, 2 triangle tests only, 3 analytic-primitive tests only
#ifndef RT_TRI_PREFETCH
#define RT_TRI_PREFETCH 1  // fetch the next triangle of a leaf's run while this one is tested (20 more live registers)
#endif

// Runs the lane's traversal to completion (lanes that are not trav_busy have nothing to do).
// min_active > 0 (the streamed kernels, rt_stream.h): the loop is left, with the unfinished lanes' state intact, as soon as
// fewer than min_active lanes of the wavefront are still traversing, so that the idle ones can fetch their next ray.
// CSG = false: the kernel is only ever launched on scenes without CSG objects and carries no code for them (their
// evaluation, csg_survivors, costs the traversal loops registers: the 4K frame of a CSG-free scene 39.7 -> 36.6 ms).
// top (RT_LDS_NODES > 0, the streamed trace kernels): the first RT_LDS_NODES nodes of the tree — its top levels, numbered
// breadth-first by fold_nodes — as the workgroup's copy in LDS; a node step on one of them reads that copy.
// EXPERIMENT, off by default.  Measured (4K frame / one-dragon 800x800 frame): 35.07 / 3.17 ms without, 35.59 / 3.22 ms with
// 5 nodes (two levels), 35.45 / 3.23 ms with 16 (2 KB more LDS per wavefront: exactly the 160 KB of a CU at 16 wavefronts).
// The top nodes are the hottest lines of the vector L1 already; the copy saves a fraction of a short latency and pays a
// divergent branch in every node step.
#ifndef RT_LDS_NODES
#define RT_LDS_NODES 0
#endif
template <bool STATS, int MS = MS_ALL, bool CSG = true>
__device__ __forceinline__ void trav_run(const DevScene& s, Trav& T, uint2* lds, uint2* ovf, Counters& cnt, uint32_t min_active = 0,
                                         const uint4* top = nullptr) {
  while (trav_busy(T)) {
    if (min_active && (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) < min_active) break;
    unsigned long long tc0 = (STATS && min_active) ? (unsigned long long)clock64() : 0ull;
    // ---- inner nodes: descend until this lane holds a leaf (or runs out of work) ----
#if RT_NODE_BREAK
    const uint32_t busy0 = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true));
#endif
    for (;;) {
      if ((T.cur >> RT_FREF_KIND_SHIFT) != RT_FREF_INNER) break;
#if RT_NODE_BREAK
      if (min_active) {  // few lanes still descending while others hold leaves: do the leaves first
        uint32_t act = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true));
        if (act < (uint32_t)RT_NODE_BREAK && act < busy0) break;
      }
#endif
      if (STATS && (RT_COUNT_PHASE == 0 || RT_COUNT_PHASE == 1)) count_step(cnt);
      // The whole node is fetched first (16-B loads, all in flight together), then tested: written out by hand because
      // the compiler otherwise fetches and waits box by box.
      const bool all_t = in_mode<MS>(T, M_CONTAINERS);
      double ent[4];
#if RT_QNODE
      const uint4* np = reinterpret_cast<const uint4*>(s.nodes + (T.cur & RT_FREF_INDEX_MASK));
      const uint4 head = np[0], refs = np[1], qa = np[2];
      const uint2 qb = *reinterpret_cast<const uint2*>(np + 3);
      const uint32_t ref4[4] = {refs.x, refs.y, refs.z, refs.w};
      const uint32_t qc[6] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y};
      if (__builtin_amdgcn_ballot_w64(T.odd != 0u) == 0ull) {  // wave-uniform: a real branch
        const float floorv = all_t ? -INFINITY : 0.f;
        const QNodeF32 qn = qnode_f32(T, head);
        for (int c = 0; c < 4; c++) {
          if (STATS && ref4[c] != RT_FREF_NONE) cnt.node_visits++;
          ent[c] = pack_entry(ref4[c], qbox_key_f32(T, qn, qc, c, floorv, ref4[c]));
        }
      } else {  // some lane's ray is parallel to an axis: the reference's operand order decides what 0 * inf does
        for (int c = 0; c < 4; c++) {
          if (STATS && ref4[c] != RT_FREF_NONE) cnt.node_visits++;
          ent[c] = pack_entry(ref4[c], qbox_key_f64<MS>(T, head, qc, c, all_t, ref4[c]));
        }
      }
#else
      const uint4* np = reinterpret_cast<const uint4*>(s.nodes + (T.cur & RT_FREF_INDEX_MASK));
      constexpr int NQ = 7;
      uint4 q[NQ];
#if RT_LDS_NODES
      const uint32_t ni = T.cur & RT_FREF_INDEX_MASK;
      if (top && ni < (uint32_t)RT_LDS_NODES) {
        for (int k = 0; k < NQ; k++) q[k] = top[ni * 8u + k];
      } else
#endif
      {
        for (int k = 0; k < NQ; k++) q[k] = np[k];
      }
      uint32_t raw[NQ * 4];
      for (int k = 0; k < NQ; k++) {
        raw[4 * k] = q[k].x;
        raw[4 * k + 1] = q[k].y;
        raw[4 * k + 2] = q[k].z;
        raw[4 * k + 3] = q[k].w;
      }
      const bool plain = __builtin_amdgcn_ballot_w64(T.odd != 0u) == 0ull;  // wave-uniform: a real branch
      if (MS == MS_SHADOW && !RT_SHADOW_SORT && RT_PUSH_BRANCHLESS && plain &&
          __builtin_amdgcn_ballot_w64(T.sp > RT_LDS_STACK - 3) == 0ull) {
        // Shadow rays, slot order, no branches: children 3, 2, 1 go to the slot above the top (counted only when they are to be
        // visited); child 0, or else the top of what was just pushed, is next.  Keys are the plain lower bounds.
        float lo[4];
        bool v[4];
        for (int c = 0; c < 4; c++) {
          if (STATS && raw[24 + c] != RT_FREF_NONE) cnt.node_visits++;
          v[c] = box_test_f32(T, raw, c, 0.f, raw[24 + c], lo[c]);
        }
        for (int c = 3; c > 0; c--) {
          uint32_t* slot = reinterpret_cast<uint32_t*>(&lds[T.sp * 64]);
          slot[0] = raw[24 + c];
          slot[1] = __float_as_uint(lo[c]);
          T.sp += v[c] ? 1 : 0;
        }
        const uint32_t top = v[1] ? raw[25] : (v[2] ? raw[26] : (v[3] ? raw[27] : RT_FREF_NONE));
        T.sp -= (!v[0] & (v[1] | v[2] | v[3])) ? 1 : 0;
        T.cur = v[0] ? raw[24] : top;
        if (T.cur == RT_FREF_NONE) T.cur = stack_pop(T, lds, ovf);
        continue;
      }
      if (plain) {
        const float floorv = all_t ? -INFINITY : 0.f;
        for (int c = 0; c < 4; c++) {
          uint32_t ref = raw[24 + c];
          if (STATS && ref != RT_FREF_NONE) cnt.node_visits++;
          ent[c] = pack_entry(ref, box_key_f32(T, raw, c, floorv, ref));
        }
      } else {  // some lane's ray is parallel to an axis: the reference's operand order decides what 0 * inf does
        for (int c = 0; c < 4; c++) {
          uint32_t ref = raw[24 + c];
          if (STATS && ref != RT_FREF_NONE) cnt.node_visits++;
          ent[c] = pack_entry(ref, box_key_f64<MS>(T, raw, c, all_t, ref));
        }
      }
#endif
      if (MS != MS_SHADOW || RT_SHADOW_SORT) {
        // nearest child first: sort the four entries ascending by key (5 compare-exchanges).  An entry read as an f64 is a
        // positive normal number whose order is the order of (key, ref) as a u64, so an exchange is v_min_f64 + v_max_f64.
#define RT_CSWAP(a, b)                            \
  {                                               \
    double lo_ = min_f64(ent[a], ent[b]), hi_ = max_f64(ent[a], ent[b]); \
    ent[a] = lo_;                                 \
    ent[b] = hi_;                                 \
  }
        RT_CSWAP(0, 1)
        RT_CSWAP(2, 3)
        RT_CSWAP(0, 2)
        RT_CSWAP(1, 3)
        RT_CSWAP(1, 2)
#undef RT_CSWAP
        if (RT_PUSH_BRANCHLESS && __builtin_amdgcn_ballot_w64(T.sp > RT_LDS_STACK - 3) == 0ull) {
          for (int c = 3; c > 0; c--) stack_push_if(T, lds, ent[c], entry_key(ent[c]) < RT_KEY_NONE);  // farthest deepest
          T.cur = entry_key(ent[0]) < RT_KEY_NONE ? entry_ref(ent[0]) : RT_FREF_NONE;
          if (T.cur == RT_FREF_NONE) T.cur = stack_pop(T, lds, ovf);
        } else if (entry_key(ent[0]) < RT_KEY_NONE) {
          for (int c = 3; c > 0; c--)  // farthest deepest
            if (entry_key(ent[c]) < RT_KEY_NONE) stack_push(T, lds, ovf, ent[c]);
          T.cur = entry_ref(ent[0]);
        } else {
          T.cur = stack_pop(T, lds, ovf);
        }
      } else if (RT_PUSH_BRANCHLESS && __builtin_amdgcn_ballot_w64(T.sp > RT_LDS_STACK - 3) == 0ull) {
        // slot order, without branches: children 3, 2, 1 are pushed; child 0, or else the top of what was just pushed, is next
        const bool v0 = entry_key(ent[0]) < RT_KEY_NONE, v1 = entry_key(ent[1]) < RT_KEY_NONE, v2 = entry_key(ent[2]) < RT_KEY_NONE,
                   v3 = entry_key(ent[3]) < RT_KEY_NONE;
        stack_push_if(T, lds, ent[3], v3);
        stack_push_if(T, lds, ent[2], v2);
        stack_push_if(T, lds, ent[1], v1);
        const uint32_t top = v1 ? entry_ref(ent[1]) : (v2 ? entry_ref(ent[2]) : (v3 ? entry_ref(ent[3]) : RT_FREF_NONE));
        T.sp -= (!v0 & (v1 | v2 | v3)) ? 1 : 0;
        T.cur = v0 ? entry_ref(ent[0]) : top;
        if (T.cur == RT_FREF_NONE) T.cur = stack_pop(T, lds, ovf);
      } else {
        // slot order: the first child to visit becomes current, the others are pushed
        double nxt = pack_entry(RT_FREF_NONE, RT_KEY_NONE);
        for (int c = 3; c >= 0; c--) {
          if (entry_key(ent[c]) < RT_KEY_NONE) {
            if (entry_key(nxt) < RT_KEY_NONE) stack_push(T, lds, ovf, nxt);
            nxt = ent[c];
          }
        }
        T.cur = entry_key(nxt) < RT_KEY_NONE ? entry_ref(nxt) : stack_pop(T, lds, ovf);
      }
    }
    unsigned long long tc1 = (STATS && min_active) ? (unsigned long long)clock64() : 0ull;
    if (STATS && min_active && first_active_lane()) cnt.cyc_trav += (uint32_t)((tc1 - tc0) >> 6);
    const uint32_t leaf = T.cur;
    if (leaf == RT_FREF_NONE) break;
#if RT_NODE_BREAK
    if ((leaf >> RT_FREF_KIND_SHIFT) == RT_FREF_INNER) continue;
#endif
    // ---- leaf ----
    if (STATS && RT_COUNT_PHASE == 0) count_step(cnt);
    uint32_t kind = leaf >> RT_FREF_KIND_SHIFT;
    uint32_t count = (leaf >> RT_FREF_COUNT_SHIFT) & RT_FREF_COUNT_MASK, first = leaf & RT_FREF_INDEX_MASK;
    if (kind == RT_FREF_TRIS) {
      // A triangle's whole 80-B record is fetched in one go (the compiler would fetch p1 only after the determinant test: a
      // second round trip), and the NEXT triangle of the run is fetched before this one is tested, so that a run of k
      // triangles costs one exposed memory round trip instead of k.
      const double2* tp = reinterpret_cast<const double2*>(s.tri_isect + (size_t)first * RT_TRI_STRIDE);
      double2 r0 = tp[0], r1 = tp[1], r2 = tp[2], r3 = tp[3], r4 = tp[4];
      for (uint32_t j = 0; j < count; j++) {
        double2 n0 = r0, n1 = r1, n2 = r2, n3 = r3, n4 = r4;
        if (RT_TRI_PREFETCH && j + 1 < count) {
          tp += 5;
          n0 = tp[0];
          n1 = tp[1];
          n2 = tp[2];
          n3 = tp[3];
          n4 = tp[4];
        }
#ifndef RT_HOST_EMU
        asm volatile("" ::: "memory");  // the loads above are issued here, before the test below
#endif
        pin_loaded(r0, r1, r2, r3, r4);
        const double tri[RT_TRI_STRIDE] = {r0.x, r0.y, r1.x, r1.y, r2.x, r2.y, r3.x, r3.y, r4.x, r4.y};
        double t, u, v;
        if (STATS && RT_COUNT_PHASE == 2) count_step(cnt);
        if (STATS) cnt.tri_tests++;
        if (triangle_roots(tri, T.o, T.d, &t, u, v)) {
          const unsigned long long last = (unsigned long long)__double_as_longlong(tri[9]);
          uint32_t prim = (uint32_t)last, material = 0u;
          if (in_mode<MS>(T, M_SHADOW)) material = ((last >> RT_TRI_OPAQUE_BIT) & 1ull) ? RT_MAT_OPAQUE : s.prim_info[prim].y;
          add_roots<STATS, MS>(s, T, prim, material, 1, &t, u, v);
        }
        if (RT_TRI_PREFETCH) {
          r0 = n0;
          r1 = n1;
          r2 = n2;
          r3 = n3;
          r4 = n4;
        } else if (j + 1 < count) {
          tp += 5;
          r0 = tp[0];
          r1 = tp[1];
          r2 = tp[2];
          r3 = tp[3];
          r4 = tp[4];
        }
      }
    } else {  // RT_FREF_PRIMS
      for (uint32_t j = 0; j < count; j++) {
        // the whole 128-B entry in one go
        if (STATS && RT_COUNT_PHASE == 3) count_step(cnt);
        const uint4* rp = reinterpret_cast<const uint4*>(s.leaf_rec + (first + j));
        uint4 h0 = rp[0];
        double2 m0 = reinterpret_cast<const double2*>(rp)[1], m1 = reinterpret_cast<const double2*>(rp)[2],
                m2 = reinterpret_cast<const double2*>(rp)[3], m3 = reinterpret_cast<const double2*>(rp)[4],
                m4 = reinterpret_cast<const double2*>(rp)[5], m5 = reinterpret_cast<const double2*>(rp)[6],
                yr = reinterpret_cast<const double2*>(rp)[7];
#ifndef RT_HOST_EMU
        // (all eight loads are issued here, before the flags are looked at: the compiler otherwise fetches the header,
        // waits, branches and only then fetches the matrix — two memory round trips per primitive instead of one)
        asm volatile("" : "+v"(h0.x), "+v"(h0.y), "+v"(h0.z), "+v"(h0.w), "+v"(m0.x), "+v"(m0.y), "+v"(m1.x), "+v"(m1.y), "+v"(m2.x),
                          "+v"(m2.y), "+v"(m3.x), "+v"(m3.y), "+v"(m4.x), "+v"(m4.y), "+v"(m5.x), "+v"(m5.y), "+v"(yr.x), "+v"(yr.y));
#endif
        uint32_t ref = h0.x;
        uint32_t idx = ref & RT_REF_INDEX_MASK;
        if (h0.w & RT_LEAFREC_INLINE) {
          const double inv[12] = {m0.x, m0.y, m1.x, m1.y, m2.x, m2.y, m3.x, m3.y, m4.x, m4.y, m5.x, m5.y};
          double ts[4], u, v;
          int n = analytic_roots<STATS>(h0.y, h0.w, inv, yr.x, yr.y, T.o, T.d, ts, u, v, cnt);
          if (n) add_roots<STATS, MS>(s, T, idx, (h0.w & RT_LEAFREC_OPAQUE) ? RT_MAT_OPAQUE : h0.z, n, ts, u, v);
        } else if ((ref >> RT_REF_KIND_SHIFT) == RT_REF_PRIM) {
          uint4 info = s.prim_info[idx];
          double ts[4], u, v;
          int n = prim_roots<STATS>(s, info, T.o, T.d, ts, u, v, cnt);
          if (n) add_roots<STATS, MS>(s, T, idx, info.y, n, ts, u, v);
        } else if (CSG) {
          if (h0.w & RT_LEAFREC_CSG2) add_csg2<STATS, MS>(s, T, idx, h0.z, cnt);
          else add_csg<STATS, MS>(s, T, idx, cnt);
        }
      }
    }
    if (STATS && min_active && first_active_lane()) cnt.cyc_adv += (uint32_t)(((unsigned long long)clock64() - tc1) >> 6);
    if (in_mode<MS>(T, M_SHADOW) && T.a >= 1.) {  // min(intensity, 1) is already 1 (intersection.rs:124-126)
      T.cur = RT_FREF_NONE;
      T.sp = 0;
    } else {
      T.cur = stack_pop(T, lds, ovf);
    }
  }
}

// get_refractive (intersection.rs:258-302) from the result of the container walk for the hit primitive `hprim`.
__device__ __forceinline__ void refractive_indices(const DevScene& s, const Trav& T, uint32_t hprim, double& n1, double& n2) {
  n1 = 1.0;  // AIR_REFRACTIVE_INDEX (material.rs:7)
  n2 = 1.0;
  if (T.p != RT_NONE) n1 = s.material[(size_t)s.prim_info[T.p].y * RT_MAT_STRIDE + RT_MAT_REFRACTIVE_INDEX];
  if (T.flag) {  // the hit object leaves the container list
    uint32_t last = (T.p == hprim) ? T.q : T.p;
    if (last != RT_NONE) n2 = s.material[(size_t)s.prim_info[last].y * RT_MAT_STRIDE + RT_MAT_REFRACTIVE_INDEX];
  } else {  // the hit object becomes the innermost container
    n2 = s.material[(size_t)s.prim_info[hprim].y * RT_MAT_STRIDE + RT_MAT_REFRACTIVE_INDEX];
  }
}

// ---------------------------------------------------------------------------------------------
// Integrator::color_at (integrator.rs:59-70) with the recursion of shade_hit_at_depth (:199-228) as a state machine.
// ---------------------------------------------------------------------------------------------
struct Frame {  // one per recursion level, in scratch
  double t, u, v;  // the hit, kept across the container walk
  double over[3], under[3], eye[3], normal[3];  // IntersecComputations (intersection.rs:201-215)
  double n1, n2;
  double refl[3];  // reflected colour x reflectivity (integrator.rs:154-168)
  double rr[3];    // reflected + refracted, or their Schlick mix (integrator.rs:213-225)
  double acc[3];   // the fold over the lights (integrator.rs:201-227)
  uint32_t prim, pc, light, pad;
};

enum {
  PC_TRACE = 0,        // a traversal is pending: leave the machine
  PC_DONE,             // the lane has no more samples
  PC_AFTER_CLOSEST,
  PC_AFTER_CONTAINERS,
  PC_HIT_READY,
  PC_AFTER_REFLECT,
  PC_REFRACT,
  PC_AFTER_REFRACT,
  PC_LIGHTS,
  PC_AFTER_SHADOW,
  PC_RETURN
};

__device__ __forceinline__ void st3(double* p, d3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ d3 ld3(const double* p) { return mk3(p[0], p[1], p[2]); }

// Source of primary rays for one lane.  Camera: the supersampling grid of one pixel (renderer.rs:50-66).
struct PixelRays {
  const rt_camera* cam;
  const DevOptions* opt;
  double x, y;
  uint32_t k, n;  // next sample, samples per pixel
  d3 color;
  __device__ __forceinline__ bool next(d3& o, d3& d) {
    if (k >= n) return false;
    uint32_t i = k / opt->n_offsets, j = k % opt->n_offsets;  // dx outer, dy inner (renderer.rs:57-64)
    camera_ray(*cam, x + opt->offsets[i], y + opt->offsets[j], o, d);
    k++;
    return true;
  }
  __device__ __forceinline__ void add(d3 c) { color = add3(color, c); }
};

// Source of one given ray (rt_trace_rays: Integrator::color_at for arbitrary rays).
struct OneRay {
  d3 o, d;
  bool pending;
  d3 color;
  __device__ __forceinline__ bool next(d3& oo, d3& dd) {
    if (!pending) return false;
    pending = false;
    oo = o;
    dd = d;
    return true;
  }
  __device__ __forceinline__ void add(d3 c) { color = c; }
};

__device__ __forceinline__ unsigned long long wave_clock() { return (unsigned long long)clock64(); }  // statistics builds only

template <bool STATS, class Rays>
__device__ __forceinline__ void run_machine(const DevScene& s, const DevOptions& opt, Rays& rays, bool lane_has_work, uint2* lds,
                                            Counters& cnt) {
  Frame frames[RT_FRAMES];
  Trav T;
  T.cur = RT_FREF_NONE;
  T.sp = 0;
  T.mode = M_CLOSEST;
  T.limit = 0.;
  int depth = 0;
  bool finished = !lane_has_work;
  if (!finished) {
    d3 o, d;
    if (rays.next(o, d)) {
      if (STATS) cnt.rays_primary++;
      trav_start(T, o, d, M_CLOSEST, INFINITY);
      frames[0].pc = PC_AFTER_CLOSEST;
    } else {
      finished = true;
    }
  }
  uint2 ovf[RT_OVF_STACK];
  while (__builtin_amdgcn_ballot_w64(!finished) != 0ull) {
    // Lock step: every lane finishes its traversal before any lane advances.  (Letting idle lanes fetch their next ray
    // while stragglers are still traversing was measured 1.8-2.3x SLOWER on the dragons frame: the state machine then
    // runs many more times with few lanes each, and the fresh rays diverge from the stragglers.)
    unsigned long long c0 = STATS ? wave_clock() : 0ull;
    trav_run<STATS>(s, T, lds, ovf, cnt);
    unsigned long long c1 = STATS ? wave_clock() : 0ull;
    if (STATS && !finished) cnt.cyc_trav += (uint32_t)((c1 - c0) >> 6);
    if (finished) continue;
    // ---- the lane's traversal is complete: advance its state machine to the next ray ----
    uint32_t pc = frames[depth].pc;
    d3 ret = mk3(0., 0., 0.);        // colour returned by a finished frame
    d3 refracted = mk3(0., 0., 0.);
    double ht = 0., hu = 0., hv = 0., n1 = 1., n2 = 1.;
    uint32_t hprim = RT_NONE;
    do {
      Frame& F = frames[depth];
      if (pc == PC_AFTER_CLOSEST) {
#ifdef RT_EMU_TRACE
        printf("fast  depth %d hit prim %u t %.17g o %.17g %.17g %.17g d %.17g %.17g %.17g\n", depth, T.p, T.limit, T.o.x, T.o.y, T.o.z, T.d.x, T.d.y, T.d.z);
#endif
        if (T.p != RT_NONE && T.t2 <= near_limit(T.limit)) {  // coincident surfaces: settle the hit in the reference's order
          reforder::Coll c;
          reforder::coll_init_closest(c);
          reforder::traverse<reforder::MODE_CLOSEST, STATS>(s, T.o, T.d, c, cnt, near_limit(T.limit));
          T.p = c.hit_prim;
          T.limit = c.limit;
          T.a = c.hu;
          T.b = c.hv;
          if (STATS) cnt.retraced++;
        }
        if (T.p == RT_NONE) {  // miss: black (integrator.rs:62-65)
          ret = mk3(0., 0., 0.);
          pc = PC_RETURN;
        } else {
          hprim = T.p;
          ht = T.limit;
          hu = T.a;
          hv = T.b;
          n1 = 1.0;  // AIR_REFRACTIVE_INDEX (material.rs:7)
          n2 = 1.0;
          uint32_t m = s.prim_info[hprim].y;
          if (!approx_eq(s.material[(size_t)m * RT_MAT_STRIDE + RT_MAT_TRANSPARENCY], 0.)) {
            // Intersection::compute_refractive (intersection.rs:184-190) needs the negative-t intersections of this ray
            F.t = ht;
            F.u = hu;
            F.v = hv;
            F.prim = hprim;
            F.pc = PC_AFTER_CONTAINERS;
            d3 o = T.o, d = T.d;
            trav_start(T, o, d, M_CONTAINERS, 0.);
            T.h = hprim;
            pc = PC_TRACE;
          } else {
            pc = PC_HIT_READY;
          }
        }
      }
#ifdef RT_EMU_TRACE
      if (pc == PC_AFTER_CONTAINERS) printf("fast  containers pa %u ta %.17g pb %u tb %.17g flag %u\n", T.p, T.a, T.q, T.b, T.flag);
#endif
      if (pc == PC_AFTER_CONTAINERS) {  // get_refractive (intersection.rs:258-302)
        hprim = F.prim;
        ht = F.t;
        hu = F.u;
        hv = F.v;
        refractive_indices(s, T, hprim, n1, n2);
        pc = PC_HIT_READY;
      }
      if (pc == PC_HIT_READY) {  // IntersecComputations::new (intersection.rs:217-256)
        uint4 info = s.prim_info[hprim];
        const double* mat = s.material + (size_t)info.y * RT_MAT_STRIDE;
        HitGeom g = hit_geom(s, info, T.o, T.d, ht, hu, hv);
        F.prim = hprim;
        F.n1 = n1;
        F.n2 = n2;
        st3(F.over, g.over);
        st3(F.under, g.under);
        st3(F.eye, g.eye);
        st3(F.normal, g.normal);
        st3(F.refl, mk3(0., 0., 0.));
        if (STATS) {  // algorithmic bytes of shading this hit
          cnt.hits++;
          uint32_t b = 64u + 48u * s.n_lights;
          b += info.x == RT_SHAPE_SMOOTH_TRIANGLE ? 72u : (info.x == RT_SHAPE_TRIANGLE ? 24u : (info.z != RT_NONE ? 96u : 0u));
          if (s.pattern_kind[info.y] != RT_PATTERN_CONST) b += 144u;
          cnt.hit_bytes += b;
        }
        // reflected_color_at_depth (integrator.rs:154-168)
        if ((uint32_t)depth < opt.max_depth && !approx_eq(mat[RT_MAT_REFLECTIVITY], 0.)) {
          F.pc = PC_AFTER_REFLECT;
          depth++;
          frames[depth].pc = PC_AFTER_CLOSEST;
          if (STATS) cnt.rays_reflect++;
          trav_start(T, g.over, g.reflect_v, M_CLOSEST, INFINITY);
          pc = PC_TRACE;
        } else {
          pc = PC_REFRACT;
        }
      }
      if (pc == PC_AFTER_REFLECT) {  // color * reflectivity (integrator.rs:167)
        double kr = s.material[(size_t)s.prim_info[F.prim].y * RT_MAT_STRIDE + RT_MAT_REFLECTIVITY];
        st3(F.refl, mk3(ret.x * kr, ret.y * kr, ret.z * kr));
        pc = PC_REFRACT;
      }
      if (pc == PC_REFRACT) {  // refracted_color_at_depth (integrator.rs:170-197)
        const double* mat = s.material + (size_t)s.prim_info[F.prim].y * RT_MAT_STRIDE;
        refracted = mk3(0., 0., 0.);
        pc = PC_LIGHTS;
        if ((uint32_t)depth < opt.max_depth && !approx_eq(mat[RT_MAT_TRANSPARENCY], 0.)) {
          d3 eye = ld3(F.eye), normal = ld3(F.normal);
          double ratio = F.n1 / F.n2;
          double cos_i = dot3(eye, normal);
          double sin2_t = (ratio * ratio) * (1. - cos_i * cos_i);
          if (!(sin2_t > 1.)) {
            double cos_t = sqrt(1. - sin2_t);
            d3 dir = sub3(scale3(normal, ratio * cos_i - cos_t), scale3(eye, ratio));
            F.pc = PC_AFTER_REFRACT;
            d3 under = ld3(F.under);
            depth++;
            frames[depth].pc = PC_AFTER_CLOSEST;
            if (STATS) cnt.rays_refract++;
            trav_start(T, under, dir, M_CLOSEST, INFINITY);
            pc = PC_TRACE;
          }
        }
      }
      if (pc == PC_AFTER_REFRACT) {  // color * transparency (integrator.rs:196)
        double kt = s.material[(size_t)s.prim_info[F.prim].y * RT_MAT_STRIDE + RT_MAT_TRANSPARENCY];
        refracted = mk3(ret.x * kt, ret.y * kt, ret.z * kt);
        pc = PC_LIGHTS;
      }
      if (pc == PC_LIGHTS) {  // shade_hit_at_depth (integrator.rs:199-228): the terms that do not depend on the light
        const double* mat = s.material + (size_t)s.prim_info[F.prim].y * RT_MAT_STRIDE;
        d3 reflected = ld3(F.refl);
        double kr = mat[RT_MAT_REFLECTIVITY], kt = mat[RT_MAT_TRANSPARENCY];
        bool use_schlick = kr > 0. && kt > 0. && !approx_eq(kr, 0.) && !approx_eq(kt, 0.);
        d3 rr;
        if (use_schlick) {
          double reflectance = schlick(ld3(F.eye), ld3(F.normal), F.n1, F.n2);
          double k2 = 1. - reflectance;
          rr = mk3(reflected.x * reflectance + refracted.x * k2, reflected.y * reflectance + refracted.y * k2,
                   reflected.z * reflectance + refracted.z * k2);
        } else {
          rr = add3(reflected, refracted);
        }
        st3(F.rr, rr);
        st3(F.acc, mk3(0., 0., 0.));
        if (s.n_lights == 0) {
          ret = mk3(0., 0., 0.);
          pc = PC_RETURN;
        } else {
          pc = PC_AFTER_SHADOW;  // falls into the shadow-ray set-up below with light 0
          F.light = RT_NONE;
        }
      }
      if (pc == PC_AFTER_SHADOW) {
        uint32_t l = F.light;
        d3 over = ld3(F.over);
        if (l != RT_NONE) {  // a shadow traversal for light l has just finished
          uint4 info = s.prim_info[F.prim];
          const double* mat = s.material + (size_t)info.y * RT_MAT_STRIDE;
          double shadow = fmin(T.a, 1.);  // IntersectionCollector::shadow_intensity (intersection.rs:124-126)
          if (!opt.shadow_intensity) shadow = approx_eq(shadow, 0.) ? 0. : 1.;
          d3 surface = lighting(s, l, info, mat, over, ld3(F.eye), ld3(F.normal), shadow);
          d3 acc = add3(add3(ld3(F.acc), surface), ld3(F.rr));  // acc + surface + reflected_refracted
          st3(F.acc, acc);
          l++;
        } else {
          l = 0;
        }
        if (l < s.n_lights) {  // point_shadow_intensity_comps (integrator.rs:136-152) from over_point
          const double* lp = s.light_pos + (size_t)l * 3;
          d3 vv = sub3(mk3(lp[0], lp[1], lp[2]), over);
          double dist = mag3(vv);
          d3 sdir = normalize3(vv);
          F.light = l;
          F.pc = PC_AFTER_SHADOW;
          if (STATS) cnt.rays_shadow++;
          trav_start(T, over, sdir, M_SHADOW, dist);
          pc = PC_TRACE;
        } else {
          ret = ld3(F.acc);
          pc = PC_RETURN;
        }
      }
      if (pc == PC_RETURN) {
        if (depth == 0) {  // Renderer::color_at_pixel: next sample (renderer.rs:57-64)
          rays.add(ret);
          d3 o, d;
          if (rays.next(o, d)) {
            if (STATS) cnt.rays_primary++;
            trav_start(T, o, d, M_CLOSEST, INFINITY);
            frames[0].pc = PC_AFTER_CLOSEST;
            pc = PC_TRACE;
          } else {
            finished = true;
            T.cur = RT_FREF_NONE;
            pc = PC_DONE;
          }
        } else {
          depth--;
          pc = frames[depth].pc;  // PC_AFTER_REFLECT or PC_AFTER_REFRACT of the parent
        }
      }
    } while (pc != PC_TRACE && pc != PC_DONE);
    if (STATS) cnt.cyc_adv += (uint32_t)((wave_clock() - c1) >> 6);
  }
}

}  // namespace fast
