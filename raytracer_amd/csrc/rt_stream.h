// rt_stream.h — the production render core: Renderer::color_at_pixel (src/render/renderer.rs:50-66) as a STREAM of
// rays through small gfx950 kernels instead of one kernel that holds a pixel's whole recursion.
//
// Why: BVH traversal is bound by dependent-load latency, so it wants many resident wavefronts and every lane busy.  A
// kernel that also carries the shading state machine needs ~240 VGPRs (2 waves per SIMD) and idles the lanes whose ray
// has finished.  Here the traversal kernels hold nothing but the ray and its collector (4+ waves per SIMD), are
// persistent, and a lane whose ray is finished takes the next ray from the queue at once.
//
// One batch of samples (a run of 8x8 tiles x all supersamples) flows through, level by level of the
// reflect/refract recursion (integrator.rs:154-197):
//
//   level l = 0..max_depth:
//     trace_closest   closest hit of every ray of the level (level 0: camera rays, generated in the kernel; deeper: the
//                     queue written by the level above), then — same lane, same ray — the container walk when the hit
//                     is transparent (intersection.rs:258-302).  Rays whose hit has a runner-up within the tie window
//                     (coincident surfaces, see rt_fast.h) go to the retrace list instead.
//     retrace         those few rays, settled in the reference's own visiting order (rt_reforder.h).
//     shade           IntersecComputations::new for every hit (intersection.rs:217-256): one NODE record per hit
//                     (over point, eye, normal, n1, n2), plus the reflected / refracted rays of level l+1.
//     trace_shadow    one shadow ray per (node, light) for the nodes of the level (integrator.rs:136-152) — on a SECOND
//                     stream, beside the next levels' trace / shade kernels: the deeper levels of a frame hold few rays and
//                     leave most of the GPU idle (each launch still costs a dependent chain of ~150 node steps), and the
//                     shadow rays, which nothing but the final combine waits for, fill it.
//   level l = max_depth..0:
//     combine         shade_hit_at_depth's fold over the lights (integrator.rs:199-228) for every node of the level, in
//                     the reference's operation order; the colour goes to the parent node's reflected / refracted slot,
//                     or to the sample when l = 0.
//   resolve           per pixel: the samples summed left to right, divided by n^2 (renderer.rs:57-65).
//
// Every arithmetic step is the same device function the single-kernel path (rt_fast.h) uses, applied to the same
// operands in the same order, so the two are bit-identical; only WHEN a step runs differs.  Queue and node capacities
// are bounded: a sample whose ray tree does not fit is flagged and re-rendered by the single-kernel state machine
// (fallback_kernel), so capacity affects speed, never pixels.
#pragma once
#include "rt_fast.h"

namespace stream {

#ifndef RT_STREAM_NT
#define RT_STREAM_NT 1
#endif
#define RT_RAY_STRIDE 8  // doubles per queued ray: origin, direction, (parent, root) as two u32, one spare
#define RT_HIT_FIELDS 5  // doubles per hit: t, u, v, n1, n2 (the primitive is kept apart: a miss writes and reads 4 bytes)
#define RT_LEVELS (RT_MAX_DEPTH + 1)

// Node tables are struct-of-arrays: field f of node i at [f * cap_nodes + i], so that a wavefront working on consecutive
// nodes reads and writes whole cache lines.
enum { NF_OVER = 0, NF_EYE = 3, NF_NORMAL = 6, NF_N1 = 9, NF_N2 = 10, NF_REFL = 11, NF_REFR = 14, NF_COUNT = 17 };
enum { NU_PRIM = 0, NU_PARENT = 1, NU_COUNT = 2 };
#define RT_NODE_REFLECTED 0x80000000u  // NU_PRIM: a reflected ray was traced for this hit (integrator.rs:154-168)
#define RT_NODE_REFRACTED 0x40000000u  // NU_PRIM: a refracted ray was traced (integrator.rs:170-197)
#define RT_NODE_PRIM_MASK 0x3FFFFFFFu
#define RT_PARENT_REFRACT 0x80000000u  // NU_PARENT / ray link: the colour is the parent's REFRACTED one

struct Ctrl {  // device-resident control block of one batch; zeroed when the batch starts
  uint32_t n_rays[RT_LEVELS + 1];  // rays queued for each level (level 0: unused, its rays are the batch's samples)
  uint32_t fetch[RT_LEVELS][9];    // trace_closest: next ray of the level in each XCD's share; [8]: used-up shares (xcd_take)
  uint32_t n_retrace[RT_LEVELS];
  uint32_t level_end[RT_LEVELS];   // nodes [level_end[l-1], level_end[l]) belong to level l
  uint32_t n_nodes;                // nodes allocated (may run past cap_nodes; clamped in level_end)
  uint32_t n_flagged;              // samples handed to the fallback
  uint32_t shadow_fetch[RT_LEVELS][9];  // trace_shadow of each level: next (light, node) item in each XCD's share; [8]: used-up shares
};

struct Ws {  // workspace of one batch (device pointers; owned by rt_scene)
  Ctrl* ctrl;
  double* rays[2];       // [cap_rays][RT_RAY_STRIDE]; level l reads rays[l & 1], writes rays[(l + 1) & 1]
  double* hit_f;         // [RT_HIT_FIELDS][cap_rays] t, u, v, n1, n2 of the hits of the level being processed
  uint32_t* hit_prim[2]; // primitive hit by each ray (RT_NONE: miss): [0] level 0, [n_samples], kept for resolve; [1] deeper levels, [cap_rays]
  uint32_t* retrace;     // [max(n_samples, cap_rays)]
  double* node_f;        // [NF_COUNT][cap_nodes]
  uint32_t* node_u;      // [NU_COUNT][cap_nodes]
  double* shadow;        // [n_lights][cap_nodes] accumulated shadow intensity (IntersectionCollector, intersection.rs:78-87)
  uint8_t* node_lit;     // [cap_nodes] bit l: light l is on the node's side of its surface, i.e. its shadow ray is traced (lit_bits)
  double* sample_rgb;    // [n_samples][3]
  uint8_t* sample_flag;  // [n_samples] 1: re-render this sample in the fallback kernel
  uint32_t* tile_xy;     // [n_samples / 64] Batch::tile_xy
  uint32_t cap_rays, cap_nodes;
};

struct Batch {
  uint32_t part, n_parts, tiles_x, n_owned;  // the tile partition, as rt_render_tiles_device defines it
  uint32_t tile0, n_tiles;                   // this batch: positions [tile0, tile0 + n_tiles) of the part's tile order
  uint32_t n2;                               // samples per pixel
  uint32_t n_samples;                        // n_tiles * 64 * n2 (camera batch) or the number of given rays
  uint32_t packed;
  const double* given_o;                     // rt_trace_rays: level 0 is these rays ([n_samples][3] each) instead of the camera's
  const double* given_d;
  const uint32_t* tile_xy;                   // [n_tiles] tile x | tile y << 16 of each tile of the batch (tile_table_kernel)
  unsigned long long n2_magic;               // floor(2^32 / n2) + 1: r / n2 == (r * n2_magic) >> 32 for r < 2^32 / n2
};

// ---------------------------------------------------------------------------------------------
// wavefront helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rank_in(unsigned long long m) {  // number of set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// The lanes with `want` each get a distinct index from *counter (one atomic per wavefront); consecutive lanes get
// consecutive indices.
__device__ __forceinline__ uint32_t wave_take(uint32_t* counter, bool want) {
  unsigned long long m = __builtin_amdgcn_ballot_w64(want);
  uint32_t id = 0;
  if (want) {
    uint32_t r = rank_in(m), base = 0;
    if (r == 0) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    id = base + r;
  }
  return id;
}
// XCD-aware work distribution for the persistent trace kernels.  Workgroup ids are dealt round-robin to the 8 XCDs, each
// with its own 4 MB L2, so the items (rays in queue order = image order) are cut into chunks of RT_XCD_CHUNK and chunk c
// belongs to XCD c % 8: the rays in flight on one XCD then cover a compact patch of the image and walk the same BVH
// subtrees, which its L2 can hold.  Each XCD has its own counter over its share; a wavefront whose XCD's share is used
// up takes from the next XCD's (`victim`, only ever advances).
// A wavefront does not go to the counter for every refill: returning atomics on one address are served one after the
// other (~90 ns each), which at 64 rays per atomic bounded the trace kernels at ~0.7 G rays/s per counter.  It takes
// up to RT_XCD_GRAB items at a time and hands them to its lanes from that private range.  All cursor fields are wave-uniform.
#ifndef RT_XCD_CHUNK
#define RT_XCD_CHUNK 65536u
#endif
#ifndef RT_XCD_GRAB
#define RT_XCD_GRAB 1024u
#endif
#ifndef RT_XCD_GUIDE
#define RT_XCD_GUIDE 1u  // guided grabs: the next grab is 2 / RT_XCD_GUIDE of what is left of the share per wavefront of the XCD
#endif
// What a grab costs (round 3, profiles/r03f_drain_instrumentation.txt): a returning device-scope atomic is a ~2 us round trip
// during which the whole wavefront — the lanes still traversing included — stands still, and the guided end phase of every
// launch was ~3000 of them per counter, 64 rays at a time: 10-40 % of a wavefront's life in a launch of 1-16 M rays.  Hence
// the floor of 256 and 4 turns instead of 8 (floor 0 / 128 / 192 / 256 / 384 / 512 / 1024 at 8 turns, then 2 / 4 turns:
// part of 8 of the 4K frame 5.00 / 4.95 / - / 4.90 / - / 4.94 / 5.19 -> 4.82 ms at 256 and 4; csg-lens 800x800 8.07 -> 7.62 ms).
#ifndef RT_XCD_MIN_GRAB
#define RT_XCD_MIN_GRAB 256u  // floor of the guided grab (a thin launch's is the wavefront's lanes_in_use)
#endif
#ifndef RT_XCD_TURNS
#define RT_XCD_TURNS 4u  // a wavefront's grab is sized so that it gets at least this many
#endif
struct XcdCursor {
  uint32_t xcd, victim, next, end;  // [next, end): items of XCD (xcd + victim) & 7's share this wavefront has taken
  uint32_t grab;                    // items per atomic: 64 .. RT_XCD_GRAB, so that every wavefront still gets several turns
  uint32_t done;                    // bit x: XCD x's share is known to be used up
  uint32_t lanes;                   // lanes_in_use: the smallest grab
  uint32_t flat;                    // 1: [next, end) are item ids themselves (a launch without counters, below)
  uint32_t first;                   // RT_XCD_STATIC_FIRST: the first grab's size (every wavefront's first range is its own by position)
};
// A launch with at most one group of `lanes` items per wavefront uses NO counters: wavefront i owns items [i * lanes, (i + 1) * lanes).
// (Measured, profiles/r03f_drain_instrumentation.txt: in such a launch a wavefront spent 50-97 % of its life in the hand-out, and
// one that found nothing lived 37 us walking the eight counters — returning device-scope atomics are ~2 us round trips.)
#ifndef RT_XCD_FLAT
#define RT_XCD_FLAT 1
#endif
#ifndef RT_XCD_STATIC_FIRST
#define RT_XCD_STATIC_FIRST 1
#endif
__device__ __forceinline__ uint32_t xcd_share(uint32_t n_items, uint32_t xx);
__device__ __forceinline__ XcdCursor xcd_cursor(uint32_t n_items, uint32_t lanes = 64u) {
  if (RT_XCD_FLAT && (unsigned long long)gridDim.x * lanes >= n_items) {
    const uint32_t begin = (uint32_t)min((unsigned long long)blockIdx.x * lanes, (unsigned long long)n_items), end = min(begin + lanes, n_items);
    return XcdCursor{0u, begin == end ? 8u : 0u, begin, end, 0u, 0xFFu, lanes, 1u, 0u};
  }
  uint32_t grab = lanes;
  while (grab >= 64u && grab < RT_XCD_GRAB && (unsigned long long)grab * gridDim.x * RT_XCD_TURNS < n_items) grab <<= 1;
#if RT_XCD_STATIC_FIRST
  // the first range of a wavefront is its own by position — wavefront j of XCD x owns positions [j * grab, (j + 1) * grab) of
  // x's share — and the counters hand out what lies behind those (xcd_static): no launch starts with every wavefront's atomic
  // on eight addresses
  const uint32_t xx = blockIdx.x & 7u, j = blockIdx.x >> 3, share = xcd_share(n_items, xx);
  const uint32_t begin = j < gridDim.x / 8u ? (uint32_t)min((unsigned long long)j * grab, (unsigned long long)share) : share;
  return XcdCursor{xx, 0u, begin, min(begin + grab, share), grab, 0u, lanes, 0u, grab};
#else
  return XcdCursor{blockIdx.x & 7u, 0u, 0u, 0u, grab, 0u, lanes, 0u, 0u};
#endif
}

// A launch with few items (the deep levels of a frame, a small image, one GPU's part of a frame split eight ways) is bound by
// its slowest wavefront, and a wavefront's time grows with the number of DIFFERENT rays it interleaves: such a launch hands
// each wavefront fewer rays — lanes at and above the returned number never take any — so that all of the GPU's wavefronts
// share the work (64 items per wavefront would leave most of them without).
#ifndef RT_THIN_WAVES
#define RT_THIN_WAVES 1
#endif
__device__ __forceinline__ uint32_t lanes_in_use(uint32_t n_items) {
#ifdef RT_HOST_EMU
  return 64u;
#else
  if (!RT_THIN_WAVES) return 64u;
  uint32_t per_wave = (n_items + gridDim.x - 1u) / gridDim.x, lanes = 8u;
  while (lanes < 64u && lanes < per_wave) lanes <<= 1;
  return lanes;
#endif
}

// Number of items in XCD xx's share: q in [0, share) <-> item ((q / CHUNK) * 8 + xx) * CHUNK + q % CHUNK < n_items.
__device__ __forceinline__ uint32_t xcd_share(uint32_t n_items, uint32_t xx) {
  uint32_t n_chunks = (n_items + RT_XCD_CHUNK - 1u) / RT_XCD_CHUNK;
  if (n_chunks <= xx) return 0u;
  uint32_t mine = (n_chunks - xx + 7u) / 8u, share = mine * RT_XCD_CHUNK;
  if ((n_chunks - 1u) % 8u == xx) share -= n_chunks * RT_XCD_CHUNK - n_items;  // the ragged last chunk is this XCD's
  return share;
}

// Positions at the start of XCD xx's share that its wavefronts own by position (xcd_cursor); the counter counts from there.
__device__ __forceinline__ uint32_t xcd_static(uint32_t share, uint32_t first) {
  return (uint32_t)min((unsigned long long)(gridDim.x / 8u) * first, (unsigned long long)share);
}

// A wavefront's private range is up to RT_XCD_GRAB = 1024 CONSECUTIVE positions, and consecutive rays cost alike: 1024 camera rays
// are the sixteen sample planes of one 8x8 tile, all on a dragon or all in the sky (50x apart).  One wavefront working through an
// expensive tile alone was the floor of a level-0 launch whatever the number of rays (1.1 ms of the 1.39 ms that an eighth of
// the 4K frame takes, against 4.2 ms for the whole frame), and smaller grabs pay in same-address atomics instead (cap 256: the
// whole frame 4.2 -> 6.6 ms).  So the positions of a (full) chunk are dealt out in groups of 64 with a stride of 64 groups: a
// private range of 1024 positions is sixteen sample planes of sixteen DIFFERENT tiles four tiles apart, a fair sample of the
// chunk, while each group of 64 — what a wavefront's lanes hold together — stays one sample plane of one tile.
#ifndef RT_XCD_SPREAD
#define RT_XCD_SPREAD 1
#endif
static_assert(RT_XCD_CHUNK == 65536u || !RT_XCD_SPREAD, "chunk_spread permutes the 1024 groups of a 64 Ki chunk");
__device__ __forceinline__ uint32_t chunk_spread(uint32_t r) {
  const uint32_t g = r >> 6;
  return ((((g & 15u) << 6) | (g >> 4)) << 6) | (r & 63u);
}

// Every lane with `want` gets a distinct item id < n_items, or RT_NONE when nothing is left.
__device__ __forceinline__ uint32_t xcd_take(uint32_t* counters, uint32_t n_items, XcdCursor& c, bool want) {
  uint32_t id = RT_NONE;
  while (c.victim < 8u) {
    bool w = want && id == RT_NONE;
    unsigned long long m = __builtin_amdgcn_ballot_w64(w);
    if (m == 0ull) break;
    uint32_t xx = (c.xcd + c.victim) & 7u;
    if (c.next == c.end) {  // take the next c.grab items of that XCD's share
      uint32_t share = xcd_share(n_items, xx);
      if (share == 0u || ((c.done >> xx) & 1u)) {  // nothing there, or known to be used up
        c.victim++;
        continue;
      }
      uint32_t q = 0;
      if (w && rank_in(m) == 0) q = atomicAdd(&counters[xx], c.grab);
      q = __builtin_amdgcn_readfirstlane(__shfl(q, __ffsll((long long)m) - 1, 64));
      q = (uint32_t)min((unsigned long long)q + xcd_static(share, c.first), (unsigned long long)share);
      if (q >= share) {
        // Used up.  Word 8 of the counters collects the used-up shares: when a kernel drains, every wavefront would
        // otherwise try all 8 counters, and 3072 wavefronts x 8 returning atomics on 8 addresses took ~250 us per
        // launch, work or no work.  (Round 3: reading all eight counters in one go after a failed grab instead — lanes 0..7, one
        // load each — was SLOWER everywhere: part of 8 5.46 -> 5.66 ms, dragon_800 2.70 -> 2.99 ms.)
        uint32_t seen = __hip_atomic_load(&counters[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w && rank_in(m) == 0 && !((seen >> xx) & 1u)) atomicOr(&counters[8], 1u << xx);
        c.done |= __builtin_amdgcn_readfirstlane(__shfl(seen, __ffsll((long long)m) - 1, 64)) | (1u << xx);
        c.victim++;
        continue;
      }
      c.next = q;
      c.end = min(q + c.grab, share);
      // guided: the nearer the share's end, the smaller the next grab, so that no wavefront sits on a long private range
      // while the others have run dry
      uint32_t left = share - c.end;
      c.grab = min(max((left / max(gridDim.x * RT_XCD_GUIDE / 4u, 1u)) & ~63u, max(c.lanes, (uint32_t)RT_XCD_MIN_GRAB)), (uint32_t)RT_XCD_GRAB);
    }
    uint32_t avail = c.end - c.next, r = rank_in(m);
    if (w && r < avail) {
      uint32_t q = c.next + r;
      const uint32_t chunk0 = ((q / RT_XCD_CHUNK) * 8u + xx) * RT_XCD_CHUNK;
      id = chunk0 + ((RT_XCD_SPREAD && n_items - chunk0 >= RT_XCD_CHUNK) ? chunk_spread(q % RT_XCD_CHUNK) : q % RT_XCD_CHUNK);
      if (c.flat) id = q;
    }
    c.next += min(avail, (uint32_t)__builtin_popcountll(m));
  }
  return id;
}

// Streamed once, never re-read by this kernel: loads and stores of the ray queues and of the hit / node / shadow records carry
// the non-temporal hint, so that they do not push the scene's nodes and triangles (what the traversal re-reads) out of L2.
#if RT_STREAM_NT && !defined(RT_HOST_EMU)
template <class T> __device__ __forceinline__ T ld_stream(const T* p) { return __builtin_nontemporal_load(p); }
template <class T> __device__ __forceinline__ void st_stream(T* p, T v) { __builtin_nontemporal_store(v, p); }
#else
template <class T> __device__ __forceinline__ T ld_stream(const T* p) { return *p; }
template <class T> __device__ __forceinline__ void st_stream(T* p, T v) { *p = v; }
#endif
// Makes [c.next, c.end) non-empty — the next items of this wavefront's XCD's share, or of the next XCD's that has any left —
// and names the XCD they belong to; false when nothing is left.  Called by all 64 lanes.
__device__ __forceinline__ bool xcd_range(uint32_t* counters, uint32_t n_items, XcdCursor& c, uint32_t& xx_out) {
#ifdef RT_HOST_EMU
  const bool first = true;  // (a wavefront is one lane)
#else
  const bool first = (threadIdx.x & 63u) == 0u;
#endif
  while (c.victim < 8u) {
    const uint32_t xx = (c.xcd + c.victim) & 7u;
    xx_out = xx;
    if (c.next != c.end) return true;
    const uint32_t share = xcd_share(n_items, xx);
    if (share == 0u || ((c.done >> xx) & 1u)) {
      c.victim++;
      continue;
    }
    uint32_t q = 0;
    if (first) q = atomicAdd(&counters[xx], c.grab);
    q = __builtin_amdgcn_readfirstlane(q);
    q = (uint32_t)min((unsigned long long)q + xcd_static(share, c.first), (unsigned long long)share);
    if (q >= share) {  // used up (see xcd_take)
      uint32_t seen = __hip_atomic_load(&counters[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (first && !((seen >> xx) & 1u)) atomicOr(&counters[8], 1u << xx);
      c.done |= __builtin_amdgcn_readfirstlane(seen) | (1u << xx);
      c.victim++;
      continue;
    }
    c.next = q;
    c.end = min(q + c.grab, share);
    const uint32_t left = share - c.end;
    c.grab = min(max((left / max(gridDim.x * RT_XCD_GUIDE / 4u, 1u)) & ~63u, max(c.lanes, (uint32_t)RT_XCD_MIN_GRAB)), (uint32_t)RT_XCD_GRAB);
    return true;
  }
  return false;
}

// trace_shadow_kernel's items are (light, node) pairs, light-major; 28 % of the bench frame's have the light behind the
// surface and get no ray.  Handing those to lanes would leave the lanes idle until the next refill (and the refill is the
// expensive part of a short ray), so a wavefront looks at its items 64 at a time — one byte of node_lit each —, writes the
// zero of the unlit ones and hands only the lit ones out: the k-th lane that wants a ray takes the k-th pending item of the
// window.  All fields wave-uniform.
struct LitWindow {
  uint32_t id0;                 // item id of the window's lane 0 (the window's items are consecutive)
  unsigned long long pending;   // bit i: item id0 + i is lit and not handed out yet
};
template <bool STATS>
__device__ __forceinline__ uint32_t lit_take(const Ws& w, uint32_t* counters, uint32_t n_items, uint32_t n_nodes, uint32_t node0,
                                             bool all_lit, XcdCursor& c, LitWindow& win, bool want, Counters& cnt) {
#ifdef RT_HOST_EMU
  const uint32_t lane = 0u, wave = 1u;  // (a wavefront is one lane)
#else
  const uint32_t lane = threadIdx.x & 63u, wave = 64u;
#endif
  uint32_t id = RT_NONE;
  for (;;) {
    const bool wq = want && id == RT_NONE;
    const unsigned long long wm = __builtin_amdgcn_ballot_w64(wq);
    if (wm == 0ull) break;
    if (win.pending == 0ull) {  // the next window: up to 64 consecutive items of the private range
      uint32_t xx = 0;
      if (!xcd_range(counters, n_items, c, xx)) break;
      const uint32_t q = c.next, n_win = min(wave, min(c.end - q, RT_XCD_CHUNK - q % RT_XCD_CHUNK));
      // (not chunk_spread: a wavefront's successive windows are neighbouring surface points facing the same light, and dealing
      // them apart cost the 4K frame's shadow rays 12.7 -> 14.1 ms)
      win.id0 = c.flat ? q : ((q / RT_XCD_CHUNK) * 8u + xx) * RT_XCD_CHUNK + q % RT_XCD_CHUNK;
      c.next += n_win;
      bool lit = false;
      if (lane < n_win) {
        const uint32_t item = win.id0 + lane, light = item / n_nodes, node = node0 + (item - light * n_nodes);
        lit = all_lit || ((w.node_lit[node] >> light) & 1u) != 0u;  // (all_lit: more lights than node_lit has bits, lit_bits)
        if (!lit) {
          // (combine_kernel reads the slot whatever the light's side: telling it apart there — one more byte per node and a
          // select per light — made that kernel slower, 2.06 -> 2.27 ms, for 0.4 GB less written here)
          st_stream(&w.shadow[(size_t)light * w.cap_nodes + node], 0.);
          if (STATS) cnt.shadow_culled++;
        }
      }
      win.pending = __builtin_amdgcn_ballot_w64(lit);
      continue;
    }
#ifdef RT_HOST_EMU  // (a wavefront is one lane)
    id = win.id0;
    win.pending = 0ull;
#else
    // rank r of this lane's item among the pending ones; a permutation sends every lane's index to slot r (pending) or
    // behind all pending ones (not pending), so that slot k holds the position of the k-th pending item
    const uint32_t have = (uint32_t)__builtin_popcountll(win.pending), r = rank_in(win.pending);
    const bool mine = ((win.pending >> lane) & 1ull) != 0ull;
    const uint32_t slot = mine ? r : have + (lane - r);
    const uint32_t pos_of = (uint32_t)__builtin_amdgcn_ds_permute((int)(slot << 2), (int)lane);
    const uint32_t k = rank_in(wm);
    const uint32_t pos = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((k & 63u) << 2), (int)pos_of);
    if (wq && k < have) id = win.id0 + pos;
    const uint32_t handed = min(have, (uint32_t)__builtin_popcountll(wm));
    win.pending = __builtin_amdgcn_ballot_w64(mine && r >= handed);
#endif
  }
  return id;
}

__device__ __forceinline__ uint64_t dbits(double d) { return (uint64_t)__double_as_longlong(d); }
__device__ __forceinline__ double bitsd(uint64_t u) { return __longlong_as_double((long long)u); }

// Which lights get a shadow ray from a surface point: bit l is clear when light l is BEHIND the surface, where
// color_of_illuminated_point returns the ambient term without looking at the shadow value (point_light.rs:50-55).  The
// predicate is the one trace_shadow_kernel's shadow_start applies, dot(normalize(light - over), normal) < 0, decided
// without the square root and the three divisions whenever dot(light - over, normal) is clear of zero by 1e-9 of its
// terms' magnitudes (normalising divides the three components by one positive number, each quotient rounded by 2^-53, and
// the dot product rounds three more times: the sign cannot change beyond ~1e-15 of those magnitudes), and exactly
// otherwise.  More than RT_LIT_LIGHTS lights: every bit set (shadow_start culls).
#define RT_LIT_LIGHTS 8u
__device__ __forceinline__ uint32_t lit_bits(const DevScene& s, d3 over, d3 normal) {
  if (s.n_lights > RT_LIT_LIGHTS) return 0xFFu;
  uint32_t bits = 0;
  for (uint32_t l = 0; l < s.n_lights; l++) {
    const double* lp = s.light_pos + (size_t)l * 3;
    d3 vv = sub3(mk3(lp[0], lp[1], lp[2]), over);
    double tx = vv.x * normal.x, ty = vv.y * normal.y, tz = vv.z * normal.z;
    double c = tx + ty + tz, bound = 1.0e-9 * (fabs(tx) + fabs(ty) + fabs(tz));
    bool lit;
    if (c > bound) lit = true;
    else if (c < -bound) lit = false;
    else lit = !(dot3(normalize3(vv), normal) < 0.);
    bits |= lit ? (1u << l) : 0u;
  }
  return bits;
}

// Pixel of lane-slot p of the tile at position kpos of this part's tile order (the order rt_render_tiles_device packs).
__device__ __forceinline__ void tile_pixel(const Batch& b, uint32_t kpos, uint32_t p, uint32_t& px, uint32_t& py) {
  uint32_t k = kpos;
  if (b.tiles_x % b.n_parts == 0) {  // this part's tiles form a regular grid: walk it block by block (L2 locality)
    uint32_t gx = b.tiles_x / b.n_parts;
    k = blocked_order(k, gx, b.n_owned / gx);
  }
  uint32_t tile = k * b.n_parts + b.part;
  px = (tile % b.tiles_x) * RT_TILE_DIM + (p & 7u);
  py = (tile / b.tiles_x) * RT_TILE_DIM + (p >> 3);
}
// Position in the PACKED output of that tile (rt_render_tiles_device: the k-th owned tile in row-major tile order).
__device__ __forceinline__ uint32_t tile_packed_index(const Batch& b, uint32_t kpos) {
  if (b.tiles_x % b.n_parts == 0) {
    uint32_t gx = b.tiles_x / b.n_parts;
    return blocked_order(kpos, gx, b.n_owned / gx);
  }
  return kpos;
}

// The tiles of a batch, looked up by the kernels instead of re-deriving the block order per ray.
__global__ __launch_bounds__(256) void tile_table_kernel(Batch b, uint32_t* __restrict__ tile_xy) {
  for (uint32_t tl = blockIdx.x * blockDim.x + threadIdx.x; tl < b.n_tiles; tl += gridDim.x * blockDim.x) {
    uint32_t px, py;
    tile_pixel(b, b.tile0 + tl, 0, px, py);
    tile_xy[tl] = (px / RT_TILE_DIM) | ((py / RT_TILE_DIM) << 16);
  }
}

// Sample id -> camera ray.  id = ((tile of the batch * n2) + sample k) * 64 + pixel slot, so 64 consecutive ids are one
// sample of one 8x8 tile.  Camera::ray_for_pixel (camera.rs:189-201) at the offsets of renderer.rs:57-64 (dx outer).
__device__ __forceinline__ bool camera_sample(const rt_camera& cam, const DevOptions& opt, const Batch& b, uint32_t id, d3& o, d3& d) {
  uint32_t p = id & 63u, r = id >> 6, tl = (uint32_t)((r * b.n2_magic) >> 32), k = r - tl * b.n2;  // no integer division here
  uint32_t xy = b.tile_xy[tl];
  uint32_t px = (xy & 0xFFFFu) * RT_TILE_DIM + (p & 7u), py = (xy >> 16) * RT_TILE_DIM + (p >> 3);
  if (px >= cam.width || py >= cam.height) return false;  // ragged edge: no such pixel
  uint32_t n = opt.n_offsets;
  uint32_t i = (k >= n ? 1u : 0u) + (k >= 2 * n ? 1u : 0u) + (k >= 3 * n ? 1u : 0u) + (k >= 4 * n ? 1u : 0u), j = k - i * n;
  double ox = opt.offsets[0], oy = opt.offsets[0];
  for (uint32_t q = 1; q < 5; q++) {  // select, not index: the options live in scalar registers
    if (i == q) ox = opt.offsets[q];
    if (j == q) oy = opt.offsets[q];
  }
  camera_ray(cam, (double)px + ox, (double)py + oy, o, d);
  return true;
}

// The level-0 ray of sample `id`: the camera's, or the id-th given ray (rt_trace_rays).
template <bool CAMERA>
__device__ __forceinline__ bool first_ray(const rt_camera& cam, const DevOptions& opt, const Batch& b, uint32_t id, d3& o, d3& d) {
  if (CAMERA) return camera_sample(cam, opt, b, id, o, d);
  const double* po = b.given_o + (size_t)id * 3;
  const double* pd = b.given_d + (size_t)id * 3;
  o = mk3(po[0], po[1], po[2]);
  d = mk3(pd[0], pd[1], pd[2]);
  return true;
}

__device__ __forceinline__ void load_ray(const double* __restrict__ q, uint32_t id, d3& o, d3& d, uint32_t& parent, uint32_t& root) {
  const double* r = q + (size_t)id * RT_RAY_STRIDE;
  o = mk3(ld_stream(r + 0), ld_stream(r + 1), ld_stream(r + 2));
  d = mk3(ld_stream(r + 3), ld_stream(r + 4), ld_stream(r + 5));
  uint64_t link = dbits(ld_stream(r + 6));
  parent = (uint32_t)link;
  root = (uint32_t)(link >> 32);
}
__device__ __forceinline__ void store_ray(double* __restrict__ q, uint32_t id, d3 o, d3 d, uint32_t parent, uint32_t root) {
  double* r = q + (size_t)id * RT_RAY_STRIDE;
  r[0] = o.x;
  r[1] = o.y;
  r[2] = o.z;
  r[3] = d.x;
  r[4] = d.y;
  r[5] = d.z;
  r[6] = bitsd((uint64_t)parent | ((uint64_t)root << 32));
}

// with_n = false (a scene without container walks, DevScene::has_walks): n1 = n2 = 1.0 for every hit, neither stored here nor read
// by the shade kernel (16 of a hit's 44 bytes, each way).
__device__ __forceinline__ void store_hit(const Ws& w, uint32_t level, uint32_t id, uint32_t prim, double t, double u, double v,
                                          double n1, double n2, bool with_n = true) {
  w.hit_prim[level ? 1 : 0][id] = prim;  // (plain stores: the shade kernel reads the hits right after this kernel)
  if (prim != RT_NONE) {
    double* h = w.hit_f + id;
    const size_t cap = w.cap_rays;
    h[0] = t;
    h[cap] = u;
    h[2 * cap] = v;
    if (with_n) {
      h[3 * cap] = n1;
      h[4 * cap] = n2;
    }
  }
}

__device__ __forceinline__ void flag_sample(const Ws& w, uint32_t root) {
  w.sample_flag[root] = 1;
  atomicAdd(&w.ctrl->n_flagged, 1u);
}

// ---------------------------------------------------------------------------------------------
// trace_closest: persistent wavefronts; Integrator::intersect (integrator.rs:55-58) for every ray of one level, plus
// the container walk of transparent hits.  A lane whose ray is finished fetches the next one as soon as fewer than
// RT_REFILL_BELOW lanes of its wavefront are still traversing.
// ---------------------------------------------------------------------------------------------
#ifndef RT_TRACE_WAVES_CSG
#define RT_TRACE_WAVES_CSG 3  // the shadow kernel with the code for CSG objects
#endif
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 4  // waves per SIMD the trace kernels are compiled for (<= 128 VGPRs)
#endif
#ifdef RT_HOST_EMU  // debugging harness (tools/hostemu): a wavefront is one lane
#define RT_REFILL_BELOW 1
#endif
#ifndef RT_REFILL_BELOW
#define RT_REFILL_BELOW 24
#endif

// EXPERIMENT (RT_LDS_NODES > 0): the workgroup's copy of the tree's first nodes in LDS (rt_scene_create pads the node table to
// at least that many).
#if RT_LDS_NODES
#define RT_LOAD_TOP_NODES(s)                                                                                       \
  __shared__ uint4 lds_top_nodes[RT_LDS_NODES * 8];                                                                  \
  for (uint32_t i_ = threadIdx.x & 63u; i_ < RT_LDS_NODES * 8u; i_ += 64u)                                           \
    lds_top_nodes[i_] = reinterpret_cast<const uint4*>(s.nodes)[i_];                                                 \
  __syncthreads();                                                                                                 \
  const uint4* lds_top = lds_top_nodes;
#else
#define RT_LOAD_TOP_NODES(s) const uint4* lds_top = nullptr;
#endif

#ifndef RT_CLOSEST_WAVES_NO_WALK
#define RT_CLOSEST_WAVES_NO_WALK 4  // the deeper levels' closest-hit kernel without the container walk, big batches: 124 VGPRs (level 0:
                                    // 138, 3 waves; at 4 it spills 16: 10.05 -> 10.5 ms).  4K frame 10.3 -> 10.1 ms; an 800x800 frame is
                                    // slower with more, thinner wavefronts
#endif
#ifndef RT_CLOSEST_WAVES
#define RT_CLOSEST_WAVES 3  // the closest-hit kernel also carries the container walk: 150 VGPRs (level 0) / 143 (deeper levels), no spills;
                            // 4 waves for the deeper levels (128 VGPRs, 12 spilled): 11.50 -> 11.63 ms.  Container walks as a kernel of
                            // their own (glass hits listed, walked after the retrace; closest-hit kernel 139 / 125 VGPRs, 4 waves for the
                            // deeper levels; the walk pruned by the second-innermost container found so far) was SLOWER: closest hits
                            // 10.7 ms + walks 1.85 ms against 11.75 ms together — a walk is short (few boxes lie behind a ray's origin),
                            // pruning changes nothing, and what it costs is a ray's set-up, which the same lane doing both pays once
#endif
// WALK = false: for scenes in which no hit is followed by the container walk (DevScene::prim_flags all clear: no transparent
// material, or every refractive index 1.0): the kernel carries no code or registers for it.
// EXPERIMENT of round 3, removed again (it is commit 35943d7 of this repository; numbers in profiles/r03c_ray_fifo_experiment.txt
// and DESIGN.md §5): a wavefront kept up to 64 PREPARED rays in LDS — set up by all lanes at once, for rays nobody needed yet —
// so that an idle lane's refill was 7 LDS reads and trav_start, cheap enough to do as soon as 52 (40, 24) lanes were left
// traversing instead of below RT_REFILL_BELOW = 24.  Shadow rays 12.75 -> 17.6 (16.0, 15.2) ms, closest hits 9.9 -> 11.4 ms, and
// the lanes active per traversal step went DOWN (0.68 -> 0.63): rays that start together stay in step — all at inner nodes,
// then all at leaves —, and a lane that joins whenever it is free is in the node loop while its neighbours test triangles.
// The batch refill is what keeps the phases of a wavefront aligned; the lanes idle before it are the price of that.
template <bool CAMERA, bool STATS, bool CSG, bool FIRST, bool WALK = true, int WAVES = RT_CLOSEST_WAVES>
__global__ __launch_bounds__(64, WAVES) void trace_closest_kernel(DevScene s, rt_camera cam, DevOptions opt, Ws w, Batch b,
                                                                          uint32_t level, unsigned long long* stats,
                                                                          uint32_t* overflow_flag) {
  __shared__ uint2 lds_stack[RT_LDS_STACK * 64];
  uint2* lds = lds_stack + (threadIdx.x & 63u);
  RT_LOAD_TOP_NODES(s)
  uint2 ovf[RT_OVF_STACK];
  Counters cnt = {};
  const unsigned long long clk0 = STATS ? (unsigned long long)clock64() : 0ull;
  // FIRST: level 0, whose rays are generated here (the camera's, or the batch's given ones); the deeper levels read their
  // queue and carry no camera code (162 VGPRs for one kernel doing both; 150 / 143 apart: 11.74 -> 11.50 ms on the 4K frame)
  const uint32_t n_rays = FIRST ? b.n_samples : w.ctrl->n_rays[level];
  const double* __restrict__ queue = w.rays[level & 1];
  fast::Trav T;
  T.cur = RT_FREF_NONE;
  T.sp = 0;
  T.mode = fast::M_CLOSEST;
  T.limit = 0.;
  uint32_t my = RT_NONE;  // the ray this lane is working on
  double ht = 0., hu = 0., hv = 0.;  // the hit, kept across the container walk
  const uint32_t lanes = lanes_in_use(n_rays);
  bool exhausted = (threadIdx.x & 63u) >= lanes;
  XcdCursor cursor = xcd_cursor(n_rays, lanes);
  for (;;) {
    // ---- idle lanes take the next rays of the level ----
    bool want = my == RT_NONE && !exhausted;
    if (__builtin_amdgcn_ballot_w64(want) != 0ull) {
      uint32_t id = xcd_take(w.ctrl->fetch[level], n_rays, cursor, want);
      if (want) {
        if (id != RT_NONE) {
          d3 o, d;
          bool ok = true;
          if (FIRST) {
            ok = first_ray<CAMERA>(cam, opt, b, id, o, d);
          } else {
            uint32_t parent, root;
            load_ray(queue, id, o, d, parent, root);
          }
          if (ok) {
            if (STATS && FIRST) cnt.rays_primary++;
            my = id;
            fast::trav_start(T, o, d, fast::M_CLOSEST, INFINITY);
          } else {
            store_hit(w, level, id, RT_NONE, 0., 0., 0., 0., 0.);
          }
        } else {
          exhausted = true;
        }
      }
    }
    if (__builtin_amdgcn_ballot_w64(fast::trav_busy(T)) == 0ull) {
      if (__builtin_amdgcn_ballot_w64(!exhausted) == 0ull) break;
      continue;
    }
    // ---- traverse; once the queue is dry there is nothing to refill with, so run to completion ----
    uint32_t min_active = __builtin_amdgcn_ballot_w64(exhausted && (threadIdx.x & 63u) < lanes) != 0ull ? 0u : (uint32_t)RT_REFILL_BELOW * lanes / 64u;
    fast::trav_run<STATS, WALK ? fast::MS_CLOSEST_CONTAINERS : fast::MS_CLOSEST, CSG>(s, T, lds, ovf, cnt, min_active, lds_top);
    // ---- lanes whose traversal is complete ----
    if (my != RT_NONE && !fast::trav_busy(T)) {
      if (!WALK || T.mode == fast::M_CLOSEST) {
        if (T.p == RT_NONE) {  // miss
          store_hit(w, level, my, RT_NONE, 0., 0., 0., 0., 0.);
          my = RT_NONE;
        } else if (T.t2 <= fast::near_limit(T.limit)) {  // coincident surfaces: settled by the retrace kernel
          uint32_t k = atomicAdd(&w.ctrl->n_retrace[level], 1u);
          w.retrace[k] = my;
          w.hit_f[my] = T.limit;  // (the ray's t slot: how far the retrace has to look, reforder::traverse's bound)
          my = RT_NONE;
        } else {
          if (WALK && (s.prim_flags[T.p] & 1u)) {  // a transparent material (one byte instead of the prim_info -> material chain)
            // Intersection::compute_refractive (intersection.rs:184-190): the negative-t intersections of this ray
            ht = T.limit;
            hu = T.a;
            hv = T.b;
            uint32_t hprim = T.p;
            d3 o = T.o, d = T.d;
            fast::trav_start(T, o, d, fast::M_CONTAINERS, 0.);
            T.h = hprim;
          } else {
            store_hit(w, level, my, T.p, T.limit, T.a, T.b, 1.0, 1.0, s.has_walks != 0u);
            my = RT_NONE;
          }
        }
      } else {  // the container walk is complete
        double n1, n2;
        fast::refractive_indices(s, T, T.h, n1, n2);
        store_hit(w, level, my, T.h, ht, hu, hv, n1, n2);
        T.mode = fast::M_CLOSEST;
        my = RT_NONE;
      }
    }
  }
  if (STATS && (threadIdx.x & 63u) == 0) cnt.cyc_total += (uint32_t)(((unsigned long long)clock64() - clk0) >> 6);
  flush_counters<STATS>(cnt, stats, overflow_flag);
}

// The rays of the retrace list: closest hit in the reference's visiting order (ObjectGroup::intersect_iter,
// group.rs:171-201), then the container walk, exactly as the single-kernel state machine settles them.
#ifndef RT_RETRACE_WAVES
#define RT_RETRACE_WAVES 4  // 128 registers, 59-140 spilled (162 and 3 waves without the bound): the walks are chains of dependent fetches, more of them in flight pay more than the spills cost (4K frame: retrace 0.97 -> 0.83 ms with 32 blocks per CU)
#endif
template <bool CAMERA, bool STATS>
__global__ __launch_bounds__(64, RT_RETRACE_WAVES) void retrace_kernel(DevScene s, rt_camera cam, DevOptions opt, Ws w, Batch b, uint32_t level,
                                                        unsigned long long* stats, uint32_t* overflow_flag) {
  __shared__ uint2 lds_stack[RT_LDS_STACK * 64];
  uint2* lds = lds_stack + (threadIdx.x & 63u);
  uint2 ovf[RT_OVF_STACK];
  Counters cnt = {};
  const uint32_t n = w.ctrl->n_retrace[level];
  const double* __restrict__ queue = w.rays[level & 1];
  for (uint32_t i = blockIdx.x * 64u + (threadIdx.x & 63u); i < n; i += gridDim.x * 64u) {
    uint32_t id = w.retrace[i];
    d3 o, d;
    if (level == 0) {
      first_ray<CAMERA>(cam, opt, b, id, o, d);
    } else {
      uint32_t parent, root;
      load_ray(queue, id, o, d, parent, root);
    }
    reforder::Coll c;
    reforder::coll_init_closest(c);
    reforder::traverse<reforder::MODE_CLOSEST, STATS>(s, o, d, c, cnt, fast::near_limit(w.hit_f[id]));
    if (STATS) cnt.retraced++;
    double n1 = 1.0, n2 = 1.0;
    if (c.hit_prim != RT_NONE) {
      if (s.prim_flags[c.hit_prim] & 1u) {  // a transparent material, in a scene where the walk's result is not known beforehand
        fast::Trav T;
        T.sp = 0;
        fast::trav_start(T, o, d, fast::M_CONTAINERS, 0.);
        T.h = c.hit_prim;
        fast::trav_run<STATS, fast::MS_ALL>(s, T, lds, ovf, cnt);
        fast::refractive_indices(s, T, c.hit_prim, n1, n2);
      }
    }
    store_hit(w, level, id, c.hit_prim, c.limit, c.hu, c.hv, n1, n2, s.has_walks != 0u);
  }
  flush_counters<STATS>(cnt, stats, overflow_flag);
}

// ---------------------------------------------------------------------------------------------
// shade: IntersecComputations::new (intersection.rs:217-256) for every hit of the level -> node records, and the
// reflected / refracted rays of the next level (integrator.rs:154-197).
// ---------------------------------------------------------------------------------------------
// One hit -> its node record and its child rays' origin/direction (shared by both loop shapes below).
// EXPERIMENT, off by default.  The eye vector of a level-0 node is minus the direction of its sample's camera ray
// (IntersecComputations::new, intersection.rs:225): the combine kernel can compute it again from the sample id — the same
// operations on the same operands, so the same bits — instead of the shade kernel storing and the combine kernel loading 24
// bytes per node (33 M nodes on the 4K frame).  Measured: shade 3.20 -> 2.99 ms, combine 2.05 -> 2.36 ms: the combine kernel
// (128 registers, 4 spilled, ~1000 instructions per node) pays more for ~150 more instructions than for 24 bytes.
#ifndef RT_EYE_FROM_CAMERA
#define RT_EYE_FROM_CAMERA 0
#endif
struct ShadeOut {
  bool want_reflect, want_refract;
  d3 child_o[2], child_d[2];
};
template <bool CAMERA, bool STATS>
__device__ __forceinline__ void shade_hit(const DevScene& s, const rt_camera& cam, const DevOptions& opt, const Ws& w, const Batch& b,
                                          uint32_t level, const double* __restrict__ queue, uint32_t id, uint32_t prim, uint32_t node,
                                          uint32_t& root, ShadeOut& out, Counters& cnt) {
  const size_t cap = w.cap_nodes;
  d3 o, d;
  uint32_t parent = id;  // level 0: the sample
  root = id;
  if (level == 0) first_ray<CAMERA>(cam, opt, b, id, o, d);
  else load_ray(queue, id, o, d, parent, root);
  const double* h = w.hit_f + id;
  const size_t hcap = w.cap_rays;
  double t = h[0], u = h[hcap], v = h[2 * hcap], n1 = 1.0, n2 = 1.0;  // (AIR_REFRACTIVE_INDEX, material.rs:7)
  if (s.has_walks) {
    n1 = h[3 * hcap];
    n2 = h[4 * hcap];
  }
  uint4 info = s.prim_info[prim];
  const double* mat = s.material + (size_t)info.y * RT_MAT_STRIDE;
  HitGeom g = hit_geom(s, info, o, d, t, u, v);
  if (STATS) {  // algorithmic bytes of shading this hit
    cnt.hits++;
    uint32_t bytes = 64u + 48u * s.n_lights;
    bytes += info.x == RT_SHAPE_SMOOTH_TRIANGLE ? 72u : (info.x == RT_SHAPE_TRIANGLE ? 24u : (info.z != RT_NONE ? 96u : 0u));
    if (s.pattern_kind[info.y] != RT_PATTERN_CONST) bytes += 144u;
    cnt.hit_bytes += bytes;
  }
  // reflected_color_at_depth (integrator.rs:154-168)
  if (level < opt.max_depth && !approx_eq(mat[RT_MAT_REFLECTIVITY], 0.)) {
    out.want_reflect = true;
    out.child_o[0] = g.over;
    out.child_d[0] = g.reflect_v;
  }
  // refracted_color_at_depth (integrator.rs:170-197)
  if (level < opt.max_depth && !approx_eq(mat[RT_MAT_TRANSPARENCY], 0.)) {
    double ratio = n1 / n2;
    double cos_i = dot3(g.eye, g.normal);
    double sin2_t = (ratio * ratio) * (1. - cos_i * cos_i);
    if (!(sin2_t > 1.)) {
      double cos_t = sqrt(1. - sin2_t);
      out.want_refract = true;
      out.child_o[1] = g.under;
      out.child_d[1] = sub3(scale3(g.normal, ratio * cos_i - cos_t), scale3(g.eye, ratio));
    }
  }
  double* nf = w.node_f + node;
  nf[(NF_OVER + 0) * cap] = g.over.x;
  nf[(NF_OVER + 1) * cap] = g.over.y;
  nf[(NF_OVER + 2) * cap] = g.over.z;
  if (!RT_EYE_FROM_CAMERA || level != 0) {  // (level 0: combine_kernel has it from the sample's camera ray again)
    nf[(NF_EYE + 0) * cap] = g.eye.x;
    nf[(NF_EYE + 1) * cap] = g.eye.y;
    nf[(NF_EYE + 2) * cap] = g.eye.z;
  }
  nf[(NF_NORMAL + 0) * cap] = g.normal.x;
  nf[(NF_NORMAL + 1) * cap] = g.normal.y;
  nf[(NF_NORMAL + 2) * cap] = g.normal.z;
  if (s.has_walks) {
    nf[NF_N1 * cap] = n1;
    nf[NF_N2 * cap] = n2;
  }
  // a child ray that misses returns black; the slots of children that are not traced are never read (combine_kernel)
  if (out.want_reflect)
    for (int k = 0; k < 3; k++) nf[(NF_REFL + k) * cap] = 0.;
  if (out.want_refract)
    for (int k = 0; k < 3; k++) nf[(NF_REFR + k) * cap] = 0.;
  w.node_lit[node] = (uint8_t)lit_bits(s, g.over, g.normal);
  uint32_t* nu = w.node_u + node;
  nu[NU_PRIM * cap] = prim | (out.want_reflect ? RT_NODE_REFLECTED : 0u) | (out.want_refract ? RT_NODE_REFRACTED : 0u);
  nu[NU_PARENT * cap] = parent;
}

// The rays of level + 1 of one wavefront: its reflected rays from queue slot `base_reflect` on, its refracted rays from
// `base_refract` on (called by all its lanes; m1 / m2 = the wavefront's lanes that want each).
template <bool STATS>
__device__ __forceinline__ void store_children(const Ws& w, double* __restrict__ next, uint32_t base_reflect, uint32_t base_refract,
                                               unsigned long long m1, unsigned long long m2, uint32_t node, uint32_t root,
                                               const ShadeOut& out, Counters& cnt) {
  if (out.want_reflect) {
    uint32_t k = base_reflect + rank_in(m1);
    if (k < w.cap_rays) {
      store_ray(next, k, out.child_o[0], out.child_d[0], node, root);
      if (STATS) cnt.rays_reflect++;
    } else {
      flag_sample(w, root);
    }
  }
  if (out.want_refract) {
    uint32_t k = base_refract + rank_in(m2);
    if (k < w.cap_rays) {
      store_ray(next, k, out.child_o[1], out.child_d[1], node | RT_PARENT_REFRACT, root);
      if (STATS) cnt.rays_refract++;
    } else {
      flag_sample(w, root);
    }
  }
}

// A block looks at SPAN rays per iteration (RT_SHADE_SPAN, or RT_SHADE_SPAN_BIG for big batches), packs the ones that hit (in ray order) into a list in LDS, takes
// their node records with ONE atomic, and then shades 256 hits at a time, again with one atomic per pass for the queue
// slots of their child rays.  Atomics on one address complete one after the other (about 9 ns each here): with one per
// wavefront for the nodes and one for the children they were more than half of the kernel's time.  Node records of a
// span are consecutive and in ray order, which keeps neighbouring shadow rays in one wavefront later.
#ifndef RT_SHADE_SPAN
#define RT_SHADE_SPAN 256u  // small batches: their shadow rays are faster behind 256 (the node records then interleave image
#endif                      // regions finely, which balances the shadow kernel's grabs: 1080p ss 4 5.4 against 5.8 ms behind 1024)
#ifndef RT_SHADE_SPAN_BIG
#define RT_SHADE_SPAN_BIG 1024u  // batches of RT_SHADE_BIG_SAMPLES samples and more: a quarter of the same-address atomics, and
#endif                           // longer runs of neighbouring node records (4K frame: shade 3.8 -> 3.5 ms, shadow rays 13.1 -> 12.9)
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 4
#endif
template <bool CAMERA, bool STATS, uint32_t SPAN>
__global__ __launch_bounds__(256, RT_SHADE_WAVES) void shade_kernel(DevScene s, rt_camera cam, DevOptions opt, Ws w, Batch b, uint32_t level,
                                                    unsigned long long* stats, uint32_t* overflow_flag) {
  Counters cnt = {};
  const uint32_t n_rays = level == 0 ? b.n_samples : w.ctrl->n_rays[level];
  const double* __restrict__ queue = w.rays[level & 1];
  double* __restrict__ next = w.rays[(level + 1) & 1];
  const uint32_t* __restrict__ prims = w.hit_prim[level ? 1 : 0];
#ifdef RT_HOST_EMU
  // (one lane at a time: no block to cooperate with)
  for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < n_rays; id += gridDim.x * blockDim.x) {
    uint32_t prim = prims[id];
    if (prim == RT_NONE) continue;
    uint32_t node = wave_take(&w.ctrl->n_nodes, true), root = id;
    if (node >= w.cap_nodes) {  // no room for this hit: the whole sample goes to the fallback
      if (level != 0) {
        d3 o, d;
        uint32_t parent;
        load_ray(queue, id, o, d, parent, root);
      }
      flag_sample(w, root);
      continue;
    }
    ShadeOut out = {};
    shade_hit<CAMERA, STATS>(s, cam, opt, w, b, level, queue, id, prim, node, root, out, cnt);
    uint32_t n_children = (out.want_reflect ? 1u : 0u) + (out.want_refract ? 1u : 0u);
    if (n_children) {
      const uint32_t cbase = atomicAdd(&w.ctrl->n_rays[level + 1], n_children);
      store_children<STATS>(w, next, cbase, cbase + (out.want_reflect ? 1u : 0u), out.want_reflect ? 1ull : 0ull,
                            out.want_refract ? 1ull : 0ull, node, root, out, cnt);
    }
  }
#else
  constexpr uint32_t PER = SPAN / 256u;  // rays a thread looks at per iteration
  __shared__ uint32_t sh_ids[SPAN];     // the span's hits: ray index within the span, in ray order
  __shared__ uint32_t sh_count[PER * 4];          // hits of (sub-span k, wavefront wv) at [k * 4 + wv]
  __shared__ uint32_t sh_base;
  __shared__ uint32_t sh_children[2][2][4], sh_child_base[2];  // [pass parity][reflected / refracted][wavefront]
  uint32_t parity = 0;
  const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  // (the loop is uniform over the BLOCK: its wavefronts meet at barriers)
  // (what each ray of a span hit is fetched one iteration ahead: the fetch was the first link of an iteration's chain of
  // memory round trips — hit list, node records, the hit's tables, child rays' slots — which is what this kernel waits on)
  uint32_t ahead[PER];
  for (uint32_t k = 0; k < PER; k++) {
    uint32_t id = blockIdx.x * SPAN + k * 256u + threadIdx.x;
    ahead[k] = id < n_rays ? prims[id] : RT_NONE;
  }
  for (uint32_t bbase = blockIdx.x * SPAN; bbase < n_rays; bbase += gridDim.x * SPAN) {
    unsigned long long hm[PER];
    for (uint32_t k = 0; k < PER; k++) {
      bool hit = ahead[k] != RT_NONE;
      hm[k] = __builtin_amdgcn_ballot_w64(hit);
      if (lane == 0) sh_count[k * 4 + wv] = (uint32_t)__builtin_popcountll(hm[k]);
      unsigned long long id = (unsigned long long)bbase + (unsigned long long)gridDim.x * SPAN + k * 256u + threadIdx.x;
      ahead[k] = id < n_rays ? prims[id] : RT_NONE;
    }
    __syncthreads();
    uint32_t total = 0, before[PER];
    for (uint32_t k = 0; k < PER; k++)
      for (uint32_t q = 0; q < 4; q++) {
        if (q == wv) before[k] = total;
        total += sh_count[k * 4 + q];
      }
    for (uint32_t k = 0; k < PER; k++)
      if ((hm[k] >> lane) & 1ull) sh_ids[before[k] + rank_in(hm[k])] = k * 256u + threadIdx.x;
    if (threadIdx.x == 0) sh_base = total ? atomicAdd(&w.ctrl->n_nodes, total) : 0u;
    __syncthreads();
    const uint32_t base = sh_base;
    for (uint32_t j0 = 0; j0 < total; j0 += 256u) {  // uniform over the block
      uint32_t j = j0 + threadIdx.x;
      bool active = j < total;
      uint32_t node = base + j, root = 0;
      ShadeOut out = {};
      if (active) {
        uint32_t id = bbase + sh_ids[j];
        if (node >= w.cap_nodes) {  // no room for this hit: the whole sample goes to the fallback
          root = id;
          if (level != 0) {
            d3 o, d;
            uint32_t parent;
            load_ray(queue, id, o, d, parent, root);
          }
          flag_sample(w, root);
        } else {
          shade_hit<CAMERA, STATS>(s, cam, opt, w, b, level, queue, id, prims[id], node, root, out, cnt);
        }
      }
      // queue slots of the block's child rays: one atomic per 256 hits (one per wavefront was a third more shade time:
      // every one of them goes to the same address).  The pass's reflected rays come first, wavefront after wavefront, then
      // its refracted rays: a wavefront of the next level's trace kernel then mostly holds rays of one kind, which travel
      // alike (reflected and refracted rays of the same hits leave in opposite directions).
      unsigned long long m1 = __builtin_amdgcn_ballot_w64(out.want_reflect), m2 = __builtin_amdgcn_ballot_w64(out.want_refract);
      if (lane == 0) {
        sh_children[parity][0][wv] = (uint32_t)__builtin_popcountll(m1);
        sh_children[parity][1][wv] = (uint32_t)__builtin_popcountll(m2);
      }
      __syncthreads();
      uint32_t all_reflect = 0, all_refract = 0, before_reflect = 0, before_refract = 0;
      for (uint32_t q = 0; q < 4; q++) {
        if (q == wv) {
          before_reflect = all_reflect;
          before_refract = all_refract;
        }
        all_reflect += sh_children[parity][0][q];
        all_refract += sh_children[parity][1][q];
      }
      if (threadIdx.x == 0) sh_child_base[parity] = all_reflect + all_refract ? atomicAdd(&w.ctrl->n_rays[level + 1], all_reflect + all_refract) : 0u;
      __syncthreads();
      const uint32_t cbase = sh_child_base[parity];
      parity ^= 1u;
      store_children<STATS>(w, next, cbase + before_reflect, cbase + all_reflect + before_refract, m1, m2, node, root, out, cnt);
    }
    __syncthreads();  // sh_ids / sh_count / sh_base are rewritten by the next iteration
  }
#endif
  flush_counters<STATS>(cnt, stats, overflow_flag);
}

// Closes a level: clamps the counters the shade kernel may have run past, and records where the level's nodes end.
__global__ void close_level_kernel(Ws w, uint32_t level) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Ctrl* c = w.ctrl;
  if (c->n_nodes > w.cap_nodes) c->n_nodes = w.cap_nodes;
  c->level_end[level] = c->n_nodes;
  if (c->n_rays[level + 1] > w.cap_rays) c->n_rays[level + 1] = w.cap_rays;
}

// ---------------------------------------------------------------------------------------------
// trace_shadow: Integrator::point_shadow_intensity_comps (integrator.rs:136-152) for every (node, light) pair.
// Item id = light * n_nodes + node: a wavefront's lanes hold neighbouring nodes and the same light (a lane that walked
// through its node's lights in turn was measured 10 % slower: the lanes of a wavefront then face different lights).
// ---------------------------------------------------------------------------------------------
// Sets up the shadow ray of (over_point, light l) — unless the light is behind the surface.  There
// color_of_illuminated_point returns the ambient term without looking at the shadow value (point_light.rs:50-55:
// `approx_eq(shadow, 1) || light_v.dot(normal) < 0`), and light_v = normalize(light - over_point) is exactly the shadow
// ray's direction (integrator.rs:110-123), so the ray's result cannot reach a pixel and it is not traced.
__device__ __forceinline__ bool shadow_start(const DevScene& s, fast::Trav& T, d3 over, d3 normal, uint32_t l) {
  const double* lp = s.light_pos + (size_t)l * 3;
  d3 vv = sub3(mk3(lp[0], lp[1], lp[2]), over);  // get_point_shadow_dist_ray (integrator.rs:110-123)
  double dist = mag3(vv);
  d3 sdir = normalize3(vv);
  if (dot3(sdir, normal) < 0.) return false;
  fast::trav_start(T, over, sdir, fast::M_SHADOW, dist);
  return true;
}
// The same for an item lit_bits has already found lit (the decision is exact there): the normal is not read again.
__device__ __forceinline__ void shadow_start_lit(const DevScene& s, fast::Trav& T, d3 over, uint32_t l) {
  const double* lp = s.light_pos + (size_t)l * 3;
  d3 vv = sub3(mk3(lp[0], lp[1], lp[2]), over);
  double dist = mag3(vv);
  fast::trav_start(T, over, normalize3(vv), fast::M_SHADOW, dist);
}

template <bool STATS, bool CSG>
__global__ __launch_bounds__(64, (CSG && !STATS) ? RT_TRACE_WAVES_CSG : RT_TRACE_WAVES) void trace_shadow_kernel(DevScene s, Ws w, uint32_t level, uint32_t last_level,
                                                                         unsigned long long* stats, uint32_t* overflow_flag) {
  __shared__ uint2 lds_stack[RT_LDS_STACK * 64];
  uint2* lds = lds_stack + (threadIdx.x & 63u);
  RT_LOAD_TOP_NODES(s)
  uint2 ovf[RT_OVF_STACK];
  Counters cnt = {};
  const unsigned long long clk0 = STATS ? (unsigned long long)clock64() : 0ull;
  // the nodes of levels level..last_level (close_level_kernel has recorded where each level's nodes end)
  const uint32_t node0 = level == 0 ? 0u : w.ctrl->level_end[level - 1];
  const uint32_t n_nodes = w.ctrl->level_end[last_level] - node0, n_items = n_nodes * s.n_lights;  // (< 2^32: ensure_workspace)
  const size_t cap = w.cap_nodes;
  const bool all_lit = s.n_lights > RT_LIT_LIGHTS;  // more lights than node_lit has bits: shadow_start decides
  fast::Trav T;
  T.cur = RT_FREF_NONE;
  T.sp = 0;
  T.mode = fast::M_SHADOW;
  T.limit = 0.;
  T.a = 0.;
  uint32_t my = RT_NONE, light = 0;  // the node this lane is working on, and the light of its current ray
  const uint32_t lanes = lanes_in_use(n_items);
  bool exhausted = (threadIdx.x & 63u) >= lanes;
  XcdCursor cursor = xcd_cursor(n_items, lanes);
  LitWindow window = {0u, 0ull};
  for (;;) {
    bool want = my == RT_NONE && !exhausted;
    if (__builtin_amdgcn_ballot_w64(want) != 0ull) {
      uint32_t id = lit_take<STATS>(w, w.ctrl->shadow_fetch[level], n_items, n_nodes, node0, all_lit, cursor, window, want, cnt);
      if (want) {
        if (id != RT_NONE) {
          light = id / n_nodes;
          my = node0 + (id - light * n_nodes);
          const double* nf = w.node_f + my;
          d3 over = mk3(ld_stream(nf + (NF_OVER + 0) * cap), ld_stream(nf + (NF_OVER + 1) * cap), ld_stream(nf + (NF_OVER + 2) * cap));
          if (!all_lit) {  // lit_take hands out lit items only
            shadow_start_lit(s, T, over, light);
            if (STATS) cnt.rays_shadow++;
          } else {
            d3 normal = mk3(ld_stream(nf + (NF_NORMAL + 0) * cap), ld_stream(nf + (NF_NORMAL + 1) * cap), ld_stream(nf + (NF_NORMAL + 2) * cap));
            if (shadow_start(s, T, over, normal, light)) {
              if (STATS) cnt.rays_shadow++;
            } else {  // the light is behind the surface: no ray
              if (STATS) cnt.shadow_culled++;
              st_stream(&w.shadow[(size_t)light * cap + my], 0.);
              my = RT_NONE;
            }
          }
        } else {
          exhausted = true;
        }
      }
    }
    if (__builtin_amdgcn_ballot_w64(fast::trav_busy(T)) == 0ull) {
      if (__builtin_amdgcn_ballot_w64(!exhausted) == 0ull) break;
      continue;
    }
    uint32_t min_active = __builtin_amdgcn_ballot_w64(exhausted && (threadIdx.x & 63u) < lanes) != 0ull ? 0u : (uint32_t)RT_REFILL_BELOW * lanes / 64u;
    fast::trav_run<STATS, fast::MS_SHADOW, CSG>(s, T, lds, ovf, cnt, min_active, lds_top);
    if (my != RT_NONE && !fast::trav_busy(T)) {
      st_stream(&w.shadow[(size_t)light * cap + my], T.a);
      my = RT_NONE;
    }
  }
  if (STATS && (threadIdx.x & 63u) == 0) cnt.cyc_total += (uint32_t)(((unsigned long long)clock64() - clk0) >> 6);
  if (STATS) flush_shadow_counters(cnt, stats);
  flush_counters<STATS>(cnt, stats, overflow_flag);
}

// ---------------------------------------------------------------------------------------------
// combine: Integrator::shade_hit_at_depth (integrator.rs:199-228) for every node of one level, deepest level first.
// ---------------------------------------------------------------------------------------------
#ifndef RT_COMBINE_WAVES
#define RT_COMBINE_WAVES 4  // 128 registers, 4 spilled: 2.34 -> 2.10 ms on the 4K frame against 3 (138 registers, none spilled)
#endif
template <bool CAMERA>
__global__ __launch_bounds__(256, RT_COMBINE_WAVES) void combine_kernel(DevScene s, rt_camera cam, DevOptions opt, Ws w, Batch b, uint32_t level) {
  const uint32_t begin = level == 0 ? 0u : w.ctrl->level_end[level - 1], end = w.ctrl->level_end[level];
  const size_t cap = w.cap_nodes;
  for (uint32_t node = begin + blockIdx.x * blockDim.x + threadIdx.x; node < end; node += gridDim.x * blockDim.x) {
    const double* nf = w.node_f + node;
    const uint32_t* nu = w.node_u + node;
    uint32_t pf = nu[NU_PRIM * cap], parent = nu[NU_PARENT * cap];
    uint4 info = s.prim_info[pf & RT_NODE_PRIM_MASK];
    const double* mat = s.material + (size_t)info.y * RT_MAT_STRIDE;
    d3 over = mk3(nf[(NF_OVER + 0) * cap], nf[(NF_OVER + 1) * cap], nf[(NF_OVER + 2) * cap]);
    d3 eye;
    if (RT_EYE_FROM_CAMERA && level == 0) {
      d3 o, d;
      first_ray<CAMERA>(cam, opt, b, parent, o, d);  // (parent: the sample)
      eye = neg3(d);
    } else {
      eye = mk3(nf[(NF_EYE + 0) * cap], nf[(NF_EYE + 1) * cap], nf[(NF_EYE + 2) * cap]);
    }
    d3 normal = mk3(nf[(NF_NORMAL + 0) * cap], nf[(NF_NORMAL + 1) * cap], nf[(NF_NORMAL + 2) * cap]);
    double kr = mat[RT_MAT_REFLECTIVITY], kt = mat[RT_MAT_TRANSPARENCY];
    d3 reflected = mk3(0., 0., 0.), refracted = mk3(0., 0., 0.);
    if (pf & RT_NODE_REFLECTED) {  // color * reflectivity (integrator.rs:167)
      reflected = mk3(nf[(NF_REFL + 0) * cap] * kr, nf[(NF_REFL + 1) * cap] * kr, nf[(NF_REFL + 2) * cap] * kr);
    }
    if (pf & RT_NODE_REFRACTED) {  // color * transparency (integrator.rs:196)
      refracted = mk3(nf[(NF_REFR + 0) * cap] * kt, nf[(NF_REFR + 1) * cap] * kt, nf[(NF_REFR + 2) * cap] * kt);
    }
    bool use_schlick = kr > 0. && kt > 0. && !approx_eq(kr, 0.) && !approx_eq(kt, 0.);
    d3 rr;
    if (use_schlick) {
      double reflectance = s.has_walks ? schlick(eye, normal, nf[NF_N1 * cap], nf[NF_N2 * cap]) : schlick(eye, normal, 1.0, 1.0);
      double k2 = 1. - reflectance;
      rr = mk3(reflected.x * reflectance + refracted.x * k2, reflected.y * reflectance + refracted.y * k2,
               reflected.z * reflectance + refracted.z * k2);
    } else {
      rr = add3(reflected, refracted);
    }
    d3 acc = mk3(0., 0., 0.);
    const d3 surface_color = pattern_color(s, info, over);  // (the same for every light)
    for (uint32_t l = 0; l < s.n_lights; l++) {
      double shadow = fmin(w.shadow[(size_t)l * cap + node], 1.);  // IntersectionCollector::shadow_intensity (intersection.rs:124-126)
      if (!opt.shadow_intensity) shadow = approx_eq(shadow, 0.) ? 0. : 1.;
      d3 surface = lighting_of(s, l, surface_color, mat, over, eye, normal, shadow);
      acc = add3(add3(acc, surface), rr);  // acc + surface + reflected_refracted
    }
    double* dst;
    if (level == 0) dst = w.sample_rgb + (size_t)parent * 3;
    else dst = w.node_f + ((parent & RT_PARENT_REFRACT) ? NF_REFR : NF_REFL) * cap + (parent & ~RT_PARENT_REFRACT);
    size_t stride = level == 0 ? 1 : cap;
    dst[0] = acc.x;
    dst[stride] = acc.y;
    dst[2 * stride] = acc.z;
  }
}

// Samples whose ray tree did not fit the batch's queues: rendered by the single-kernel state machine (rt_fast.h).
template <bool CAMERA>
__global__ __launch_bounds__(64, 2) void fallback_kernel(DevScene s, rt_camera cam, DevOptions opt, Ws w, Batch b,
                                                         unsigned long long* stats, uint32_t* summary,
                                                         uint32_t* overflow_flag) {
  __shared__ uint2 lds_stack[RT_LDS_STACK * 64];
  // summary[1]: node records this batch used per sample, in sixteenths, rounded up (largest batch of the frame): what the
  // host sizes the following frames' batches by
  if (blockIdx.x == 0 && threadIdx.x == 0 && b.n_samples) {
    atomicMax(&summary[1], (uint32_t)(((unsigned long long)w.ctrl->n_nodes * 16ull + b.n_samples - 1u) / b.n_samples));
    uint32_t most_rays = 0;  // summary[2]: the same for the longest ray queue of a deeper level
    for (uint32_t l = 1; l <= RT_LEVELS; l++) most_rays = max(most_rays, w.ctrl->n_rays[l]);
    atomicMax(&summary[2], (uint32_t)(((unsigned long long)most_rays * 16ull + b.n_samples - 1u) / b.n_samples));
  }
  if (w.ctrl->n_flagged == 0) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    atomicAdd(summary, w.ctrl->n_flagged);
    if (stats) atomicAdd(&stats[RT_COUNTER_FALLBACK], (unsigned long long)w.ctrl->n_flagged);
  }
  Counters cnt = {};
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t wbase = blockIdx.x * 64u; wbase < b.n_samples; wbase += gridDim.x * 64u) {
    uint32_t id = wbase + lane;
    bool work = id < b.n_samples && w.sample_flag[id] != 0;
    if (__builtin_amdgcn_ballot_w64(work) == 0ull) continue;
    fast::OneRay rays;
    rays.o = mk3(0., 0., 0.);
    rays.d = mk3(0., 0., 1.);
    rays.pending = true;
    rays.color = mk3(0., 0., 0.);
    if (work) work = first_ray<CAMERA>(cam, opt, b, id, rays.o, rays.d);
    fast::run_machine<false>(s, opt, rays, work, lds_stack + lane, cnt);
    if (work) {
      double* c = w.sample_rgb + (size_t)id * 3;
      c[0] = rays.color.x;
      c[1] = rays.color.y;
      c[2] = rays.color.z;
    }
  }
  if (cnt.overflow) atomicOr(overflow_flag, 1u);
}

// resolve: Renderer::color_at_pixel's sum over the samples, left to right from black, then / n^2 (renderer.rs:57-65).
__global__ __launch_bounds__(256) void resolve_kernel(rt_camera cam, Ws w, Batch b, double* __restrict__ out) {
  const uint32_t n_pixels = b.n_tiles * 64u;
  const double nn = (double)b.n2;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += gridDim.x * blockDim.x) {
    uint32_t tl = i >> 6, p = i & 63u;
    uint32_t xy = b.tile_xy[tl];
    uint32_t px = (xy & 0xFFFFu) * RT_TILE_DIM + (p & 7u), py = (xy >> 16) * RT_TILE_DIM + (p >> 3);
    bool inside = px < cam.width && py < cam.height;
    d3 color = mk3(0., 0., 0.);
    if (inside) {
      const double* src = w.sample_rgb + ((size_t)tl * b.n2 * 64u + p) * 3;
      const uint32_t* hp = w.hit_prim[0] + ((size_t)tl * b.n2 * 64u + p);
      for (uint32_t k = 0; k < b.n2; k++, src += 64 * 3, hp += 64) {
        d3 c = mk3(0., 0., 0.);  // a sample whose ray hit nothing is black (integrator.rs:62-65); nothing was stored for it
        if (*hp != RT_NONE) c = mk3(src[0], src[1], src[2]);
        color = add3(color, c);
      }
    }
    color = mk3(color.x / nn, color.y / nn, color.z / nn);  // renderer.rs:65
    if (b.packed) {
      double* dst = out + ((size_t)tile_packed_index(b, b.tile0 + tl) * 64 + p) * 3;
      dst[0] = color.x;
      dst[1] = color.y;
      dst[2] = color.z;
    } else if (inside) {
      double* dst = out + ((size_t)py * cam.width + px) * 3;
      dst[0] = color.x;
      dst[1] = color.y;
      dst[2] = color.z;
    }
  }
}

}  // namespace stream
