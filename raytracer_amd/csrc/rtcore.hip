// rtcore.hip — kernels + C-ABI of librtcore.so (include/rtcore.h), gfx950 only.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared rtcore.hip -o librtcore.so
// (-ffp-contract=off: the reference never fuses a*b+c; see DESIGN.md "Numerics").
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <thread>
#include <cstddef>
#include <exception>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "rt_fast.h"

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
#define RT_N_COUNTERS 26
#define RT_COUNTER_SHADOW 17  // streamed pipeline: trace_shadow_kernel's own box / triangle / primitive tests, wave / lane steps, node / leaf / total clocks (8 slots)
#define RT_COUNTER_FALLBACK 15  // streamed pipeline: flag events of samples handed to the fallback kernel
#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 2  // waves per SIMD the register allocator must leave room for (256 VGPRs: no spills; measured best)
#endif

template <bool STATS>
__device__ __forceinline__ void flush_counters(const Counters& c, unsigned long long* stats, uint32_t* overflow_flag) {
  if (c.overflow) atomicOr(overflow_flag, 1u);
  if (!STATS) return;
  uint32_t v[16] = {c.rays_primary, c.rays_shadow, c.rays_reflect, c.rays_refract, c.node_visits, c.tri_tests,
                               c.prim_tests,   c.hits,        c.hit_bytes,    c.overflow,     c.wave_steps,  c.lane_steps,
                               c.retraced,     c.cyc_trav,    c.cyc_adv,     c.cyc_total};
  for (int k = 0; k < 16; k++) {
    unsigned long long x = v[k];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0 && x) atomicAdd(&stats[k == 15 ? 16 : k], x);
  }
}

// trace_shadow_kernel's share of the traversal counters, kept apart for its roofline (bench.py).
__device__ __forceinline__ void flush_shadow_counters(const Counters& c, unsigned long long* stats) {
  uint32_t v[9] = {c.node_visits, c.tri_tests, c.prim_tests, c.wave_steps, c.lane_steps, c.cyc_trav, c.cyc_adv, c.cyc_total, c.shadow_culled};
  for (int k = 0; k < 9; k++) {
    unsigned long long x = v[k];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0 && x) atomicAdd(&stats[RT_COUNTER_SHADOW + k], x);
  }
}

// XCD-aware tile order.  Consecutive workgroup ids are dealt round-robin over the 8 XCDs (each with its own L2), so
// workgroup id -> position in the tile order is chosen such that every XCD receives whole groups of RT_XCD_GROUP
// consecutive positions (one 16x16-tile block of the image: neighbouring tiles walk the same BVH subtrees, so they should
// share an L2), and the groups are dealt to the XCDs round-robin.  (Giving each XCD one contiguous eighth of the frame
// instead left the XCDs that own empty sky idle: 1.45 resident waves per SIMD on average on the dragons frame.)
// The grid is rounded up to a multiple of 8 * RT_XCD_GROUP; positions >= n are idle workgroups.
#define RT_XCD_GROUP 256u
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid) {
  uint32_t xcd = bid % 8, k = bid / 8;
  return ((k / RT_XCD_GROUP) * 8 + xcd) * RT_XCD_GROUP + k % RT_XCD_GROUP;
}

// Order in which an XCD walks its run of tiles: blocks of RT_BLOCK x RT_BLOCK tiles, row-major inside a block, so that the
// tiles in flight on one XCD (32 CUs x a few waves) cover a compact patch of the image and share BVH subtrees in its L2.
// Maps the j-th position of that order to the row-major cell index of a gx x gy grid (bijective, ragged edges included).
#define RT_BLOCK 16u
__device__ __forceinline__ uint32_t blocked_order(uint32_t j, uint32_t gx, uint32_t gy) {
  uint32_t r = j / (RT_BLOCK * gx);             // block row
  uint32_t hr = min(RT_BLOCK, gy - r * RT_BLOCK);  // its height
  uint32_t jj = j - r * RT_BLOCK * gx;
  uint32_t full = gx / RT_BLOCK;                // full-width blocks in a block row
  uint32_t bc = jj / (hr * RT_BLOCK);
  uint32_t cx, cy;
  if (bc < full) {
    uint32_t w = jj - bc * hr * RT_BLOCK;
    cy = w / RT_BLOCK;
    cx = bc * RT_BLOCK + w % RT_BLOCK;
  } else {
    uint32_t wl = gx - full * RT_BLOCK;  // width of the ragged last block (> 0 here)
    uint32_t w = jj - full * hr * RT_BLOCK;
    cy = w / wl;
    cx = full * RT_BLOCK + w % wl;
  }
  return (r * RT_BLOCK + cy) * gx + cx;
}

#include "rt_stream.h"  // the production pipeline (uses flush_counters and blocked_order above)
#include "rt_mesh.h"    // device-side per-frame scene preparation (rt_mesh_*; rt_scene_create with RT_REF_MESH children)

// Renderer::color_at_pixel (renderer.rs:50-66) for one 8x8 tile per wavefront, the whole recursion in one kernel.
// REFORDER = false: the single-kernel core (rt_fast.h; rt_options.pipeline = 1, and the fallback of the streamed
// pipeline).  REFORDER = true: reference visiting order (rt_reforder.h).
// BIGCSG (with REFORDER): the general CSG evaluation (rt_shapes.h csg_survivors_general) for scenes the bounded one refuses.
template <bool STATS, bool REFORDER, bool BIGCSG = false>
__global__ __launch_bounds__(64, REFORDER ? 2 : RT_WAVES_PER_EU) void render_tiles_kernel(DevScene s, rt_camera cam, DevOptions opt, uint32_t part,
                                                                            uint32_t n_parts, uint32_t tiles_x, uint32_t n_owned,
                                                                            int packed, double* __restrict__ out,
                                                                            unsigned long long* stats, uint32_t* overflow_flag) {
  __shared__ uint2 lds_stack[RT_LDS_STACK * 64];
  uint32_t k = xcd_remap(blockIdx.x);
  if (k >= n_owned) return;
  if (tiles_x % n_parts == 0) {  // this part's tiles form a regular (tiles_x / n_parts)-wide grid: walk it block by block
    uint32_t gx = tiles_x / n_parts;
    k = blocked_order(k, gx, n_owned / gx);
  }
  uint32_t tile = k * n_parts + part;
  uint32_t lane = threadIdx.x;
  uint32_t px = (tile % tiles_x) * RT_TILE_DIM + (lane & 7u);
  uint32_t py = (tile / tiles_x) * RT_TILE_DIM + (lane >> 3);
  Counters cnt = {};
  bool inside = px < cam.width && py < cam.height;
  d3 color = mk3(0., 0., 0.);
  if (REFORDER) {
    if (inside) {
      double x = (double)px, y = (double)py;
      for (uint32_t i = 0; i < opt.n_offsets; i++)
        for (uint32_t j = 0; j < opt.n_offsets; j++) {  // dx outer, dy inner (renderer.rs:57-64)
          d3 o, d;
          camera_ray(cam, x + opt.offsets[i], y + opt.offsets[j], o, d);
          if (STATS) cnt.rays_primary++;
          color = add3(color, reforder::trace<STATS, BIGCSG>(s, opt, o, d, cnt));
        }
    }
  } else {
    fast::PixelRays rays;
    rays.cam = &cam;
    rays.opt = &opt;
    rays.x = (double)px;
    rays.y = (double)py;
    rays.k = 0;
    rays.n = opt.n_offsets * opt.n_offsets;
    rays.color = mk3(0., 0., 0.);
    fast::run_machine<STATS>(s, opt, rays, inside, lds_stack + lane, cnt);
    color = rays.color;
  }
  double nn = (double)(opt.n_offsets * opt.n_offsets);
  color = mk3(color.x / nn, color.y / nn, color.z / nn);  // renderer.rs:65
  if (packed) {
    double* dst = out + ((size_t)k * 64 + lane) * 3;
    dst[0] = color.x;
    dst[1] = color.y;
    dst[2] = color.z;
  } else if (inside) {
    double* dst = out + ((size_t)py * cam.width + px) * 3;
    dst[0] = color.x;
    dst[1] = color.y;
    dst[2] = color.z;
  }
  flush_counters<STATS>(cnt, stats, overflow_flag);
}

// Packed tiles of all parts (part-major, as an all-gather delivers them) -> row-major image.
template <class T>  // f64 pixels, or u8 ones (rt_quantize_device of the packed tiles: an eighth of the bytes through the gather)
__global__ __launch_bounds__(256) void assemble_tiles_kernel(uint32_t width, uint32_t height, uint32_t n_parts, uint32_t part_stride,
                                                             uint32_t tiles_x, uint32_t n_tiles, const T* __restrict__ tiles,
                                                             T* __restrict__ out) {
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (tile, lane)
  if (gid >= (size_t)n_tiles * 64) return;
  uint32_t tile = (uint32_t)(gid / 64), lane = (uint32_t)(gid % 64);
  uint32_t part = tile % n_parts, k = tile / n_parts;
  // offset of `part`'s block: sum of owned-tile counts of the parts before it
  size_t base = (size_t)part * part_stride;
  if (part_stride == 0)
    for (uint32_t p = 0; p < part; p++) base += (n_tiles > p) ? (n_tiles - p + n_parts - 1) / n_parts : 0;
  uint32_t px = (tile % tiles_x) * RT_TILE_DIM + (lane & 7u);
  uint32_t py = (tile / tiles_x) * RT_TILE_DIM + (lane >> 3);
  if (px >= width || py >= height) return;
  const T* src = tiles + ((base + k) * 64 + lane) * 3;
  T* dst = out + ((size_t)py * width + px) * 3;
  dst[0] = src[0];
  dst[1] = src[1];
  dst[2] = src[2];
}

// Color::as_scaled_values (color.rs:55-66): clamp -> *255 -> round half away from zero -> u8 (NaN -> 0)
__global__ __launch_bounds__(256) void quantize_kernel(const double* __restrict__ rgb, uint64_t n, uint8_t* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = rgb[i];
  uint8_t r = 0;
  if (v == v) {
    v = v < 0. ? 0. : (v > 1. ? 1. : v);
    r = (uint8_t)round(v * 255.);
  }
  out[i] = r;
}

// Integrator::color_at for arbitrary rays (one lane per ray)
template <bool REFORDER, bool BIGCSG = false>
__global__ __launch_bounds__(64, REFORDER ? 2 : RT_WAVES_PER_EU) void trace_rays_kernel(DevScene s, DevOptions opt, uint32_t n,
                                                                          const double* __restrict__ origin,
                                                                          const double* __restrict__ dir, double* __restrict__ out,
                                                                          uint32_t* overflow_flag) {
  __shared__ uint2 lds_stack[RT_LDS_STACK * 64];
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  Counters cnt = {};
  bool valid = i < n;
  d3 o = mk3(0., 0., 0.), d = mk3(0., 0., 1.), c = mk3(0., 0., 0.);
  if (valid) {
    o = mk3(origin[3 * i], origin[3 * i + 1], origin[3 * i + 2]);
    d = mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
  }
  if (REFORDER) {
    if (valid) c = reforder::trace<false, BIGCSG>(s, opt, o, d, cnt);
  } else {
    fast::OneRay rays;
    rays.o = o;
    rays.d = d;
    rays.pending = true;
    rays.color = mk3(0., 0., 0.);
    fast::run_machine<false>(s, opt, rays, valid, lds_stack + threadIdx.x, cnt);
    c = rays.color;
  }
  if (valid) {
    out[3 * i] = c.x;
    out[3 * i + 1] = c.y;
    out[3 * i + 2] = c.z;
  }
  if (cnt.overflow) atomicOr(overflow_flag, 1u);
}

// rt_debug_box_tests: the node step's conservative f32 slab test (fast::box_test_f32) beside the exact f64 test it stands in
// for (fast::box_key_f64: the reference's BoundingBox::intersection_time arithmetic, bounding_box.rs:192-225, on the node's
// f32 box), one (ray, box, limit) triple per thread.  out: bit 0 the f32 test keeps the box, bit 1 the exact test keeps it,
// bit 2 the ray takes the exact test in production (an infinite or huge 1 / d component: Trav::odd).
__global__ __launch_bounds__(256) void debug_box_test_kernel(uint32_t n, const double* __restrict__ origin, const double* __restrict__ dir,
                                                             const float* __restrict__ boxes, const double* __restrict__ limit, int all_t,
                                                             uint8_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fast::Trav T;
  const d3 o = mk3(origin[3 * i], origin[3 * i + 1], origin[3 * i + 2]), d = mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
  fast::trav_start(T, o, d, all_t ? fast::M_CONTAINERS : fast::M_CLOSEST, all_t ? 0. : limit[i]);
  if (!all_t) {  // (as add_roots leaves them after a hit at `limit`)
    T.limit = limit[i];
    T.limf = fast::f32_above(fast::near_limit(limit[i]));
  }
  uint32_t raw[6];
  for (int k = 0; k < 6; k++) raw[k] = __float_as_uint(boxes[6 * (size_t)i + k]);
  float lo;
  const bool keep32 = fast::box_test_f32(T, raw, 0, all_t ? -INFINITY : 0.f, 0u, lo);
  const bool keep64 = (all_t ? fast::box_key_f64<fast::MS_ALL>(T, raw, 0, true, 0u) : fast::box_key_f64<fast::MS_CLOSEST>(T, raw, 0, false, 0u)) != RT_KEY_NONE;
  out[i] = (keep32 ? 1u : 0u) | (keep64 ? 2u : 0u) | (T.odd ? 4u : 0u);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_err;

// rt_render_device_noalloc's workspace (include/rtcore.h): ONE array per GPU for the life of the process, shared by every
// call on that GPU.  In host code `scratch` is only a handle for hipGetSymbolAddress: the address kernels and
// hipMemsetAsync are given is the per-device one kept in g_scratch_ptr (resolved by rt_scene_create, with the scene's
// device current).
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
#define RT_MAX_DEVICES 64
std::mutex g_scratch_mu;
void* g_scratch_ptr[RT_MAX_DEVICES] = {};
bool g_scratch_tried[RT_MAX_DEVICES] = {};

// The address of `scratch` on the CURRENT device `device` (NULL when it cannot be had: the entry point then says so).
void* scratch_on_device(int device) {
  if (device < 0 || device >= RT_MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_tried[device]) {
    g_scratch_tried[device] = true;
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) == hipSuccess) g_scratch_ptr[device] = ptr;
    else (void)hipGetLastError();
  }
  return g_scratch_ptr[device];
}

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

// No exception leaves the library: the callers are C and Rust, where an unwinding C++ exception is
// undefined behaviour.  Every extern "C" entry point that can allocate on the host or start a thread is a function-try-block
// ending in RT_NO_UNWIND: std::bad_alloc (a scene's tables) and std::system_error (no thread to be had) become RT_ERR_HOST.
int fail_host(const char* what) noexcept {
  try {
    g_err = std::string("host resources exhausted: ") + what;
  } catch (...) {
    g_err.clear();  // (noexcept)
  }
  return RT_ERR_HOST;
}
#define RT_NO_UNWIND                                                          \
  catch (const std::exception& e) { return fail_host(e.what()); }            \
  catch (...) { return fail_host("unknown exception"); }

// Host threads that are joined on every path out of their scope (a std::thread destroyed while joinable terminates).
struct JoinAll {
  std::vector<std::thread>& v;
  ~JoinAll() {
    for (std::thread& t : v)
      if (t.joinable()) t.join();
  }
};

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(RT_ERR_DEVICE, std::string(#expr) + " failed: " + hipGetErrorString(_e));        \
  } while (0)

// hipEvent_t that is destroyed on every path out of its scope.
struct Event {
  hipEvent_t e = nullptr;
  Event() = default;
  Event(const Event&) = delete;
  Event& operator=(const Event&) = delete;
  ~Event() {
    if (e) (void)hipEventDestroy(e);
  }
  hipError_t create() { return hipEventCreate(&e); }
};
// Device allocation that is freed on every path out of its scope.
struct DeviceBuffer {
  void* p = nullptr;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  ~DeviceBuffer() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};

// Device memory that scenes hand back when they are destroyed, kept for the next scene on the same device: an
// animation creates and destroys one rt_scene per frame (a scene table arena and a render workspace each), and
// hipMalloc / hipFree cost milliseconds and synchronise the device.  Bounded: at most 24 blocks and 96 GiB
// (RT_POOL_MAX_BYTES); the oldest blocks go first.  rt_release_cached_memory() empties it.
struct DevicePool {
  struct Block {
    void* p;
    size_t cap;
    int device;
  };
  std::mutex mu;
  std::vector<Block> blocks;  // oldest first
  std::atomic<size_t> cached{0};  // (read without the mutex by ensure_workspace)

  // A block of at least `need` bytes on `device`: a cached one that is not more than twice too large, else a new one.
  void* take(size_t need, int device, size_t* cap) {
    {
      std::lock_guard<std::mutex> lock(mu);
      int best = -1;
      for (size_t i = 0; i < blocks.size(); i++) {
        const Block& b = blocks[i];
        if (b.device != device || b.cap < need || b.cap > 2 * need + (1u << 20)) continue;
        if (best < 0 || b.cap < blocks[(size_t)best].cap) best = (int)i;
      }
      if (best >= 0) {
        Block b = blocks[(size_t)best];
        blocks.erase(blocks.begin() + best);
        cached -= b.cap;
        *cap = b.cap;
        return b.p;
      }
    }
    void* p = nullptr;
    if (hipMalloc(&p, need) != hipSuccess) {
      (void)hipGetLastError();
      clear();  // make room and try once more
      if (hipMalloc(&p, need) != hipSuccess) return nullptr;
    }
    *cap = need;
    return p;
  }
  // The caller has made sure that no GPU work still uses the block.
  void give(void* p, size_t cap, int device) {
    if (!p) return;
    size_t limit = 96ull << 30;
    if (const char* e = std::getenv("RT_POOL_MAX_BYTES")) limit = std::strtoull(e, nullptr, 10);
    std::vector<Block> evict;
    {
      std::lock_guard<std::mutex> lock(mu);
      blocks.push_back({p, cap, device});
      cached += cap;
      while (!blocks.empty() && (blocks.size() > 24 || cached > limit)) {
        evict.push_back(blocks.front());
        cached -= blocks.front().cap;
        blocks.erase(blocks.begin());
      }
    }
    for (const Block& b : evict) (void)hipFree(b.p);
  }
  void clear() {
    std::vector<Block> all;
    {
      std::lock_guard<std::mutex> lock(mu);
      all.swap(blocks);
      cached = 0;
    }
    for (const Block& b : all) (void)hipFree(b.p);
  }
};
DevicePool g_device_pool;

// 16-byte slots of pinned host memory (the per-frame fallback count lands in one): pinning is slow, so slots are cut from
// 4-KiB chunks and reused; the chunks live as long as the process.
struct PinnedSlots {
  std::mutex mu;
  std::vector<uint32_t*> free_slots;
  uint32_t* take() {
    std::lock_guard<std::mutex> lock(mu);
    if (free_slots.empty()) {
      void* chunk = nullptr;
      if (hipHostMalloc(&chunk, 4096, hipHostMallocPortable) != hipSuccess) return nullptr;
      chunks.push_back(chunk);
      for (size_t off = 0; off < 4096; off += 16) free_slots.push_back((uint32_t*)((char*)chunk + off));
    }
    uint32_t* s = free_slots.back();
    free_slots.pop_back();
    return s;
  }
  void give(uint32_t* s) {
    if (!s) return;
    std::lock_guard<std::mutex> lock(mu);
    free_slots.push_back(s);
  }
  std::vector<void*> chunks;
  // rt_shutdown: frees the chunks when no slot is out (a live scene still holds one: then they stay)
  void release() {
    std::lock_guard<std::mutex> lock(mu);
    if (free_slots.size() != chunks.size() * (4096 / 16)) return;
    for (void* c : chunks) (void)hipHostFree(c);
    chunks.clear();
    free_slots.clear();
  }
};
PinnedSlots g_pinned_slots;

}  // namespace

// A growable byte buffer that does not zero what it grows by (unlike std::vector<char>::resize).
struct StageBuffer {
  char* p = nullptr;
  size_t size = 0, cap = 0;
  char* resize(size_t n) {
    if (n > cap) {
      size_t want = std::max(n, cap + cap / 2);
      char* q = (char*)std::realloc(p, want);
      if (!q) {
        std::free(p);
        p = nullptr;
        size = cap = 0;
        return nullptr;
      }
      p = q;
      cap = want;
    }
    size = n;
    return p;
  }
  // Room for n bytes without moving: tables that are filled IN PLACE (stage_inplace) hold pointers into the buffer.
  bool reserve(size_t n) {
    if (n <= cap) return true;
    const size_t keep = size;
    if (!resize(n)) return false;
    size = keep;
    return true;
  }
  ~StageBuffer() { std::free(p); }
};

// What frames of one kind of scene taught about the batch layout (nodes_per_sample, ratio_floor), kept per device and scene
// shape for the scenes to come: an animation (or the one-shot rt_render, which is what the reference-side hook calls,
// renderer.rs:68-98) makes a new rt_scene per frame, and every new scene would otherwise start from 3 node records per
// sample again (the 4K bench frame: two batches instead of one).  A wrong guess costs time, never pixels (fallback_kernel).
struct LearnedLayout {
  struct Entry {
    int device;
    uint64_t shape;
    uint32_t nodes_per_sample, ratio_floor;
  };
  std::mutex mu;
  std::vector<Entry> entries;  // most recent last, at most 16
  bool lookup(int device, uint64_t shape, uint32_t* nodes_per_sample, uint32_t* ratio_floor) {
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry& e : entries)
      if (e.device == device && e.shape == shape) {
        *nodes_per_sample = e.nodes_per_sample;
        *ratio_floor = e.ratio_floor;
        return true;
      }
    return false;
  }
  void remember(int device, uint64_t shape, uint32_t nodes_per_sample, uint32_t ratio_floor) {
    std::lock_guard<std::mutex> lock(mu);
    for (size_t i = 0; i < entries.size(); i++)
      if (entries[i].device == device && entries[i].shape == shape) {
        entries.erase(entries.begin() + (long)i);
        break;
      }
    if (entries.size() >= 16) entries.erase(entries.begin());
    entries.push_back({device, shape, nodes_per_sample, ratio_floor});
  }
};
LearnedLayout g_learned;

// A group of triangle primitives resident on the device (include/rtcore.h "rt_mesh"; kernels in rt_mesh.h).
struct rt_mesh {
  int device = 0;
  uint32_t n = 0, n_palette = 0;
  void* block = nullptr;  // base pose, frame pose and build state: one allocation
  size_t block_bytes = 0;
  mesh::MeshDev dev{};
  hipStream_t stream = nullptr;
  unsigned long long* h_box = nullptr;  // pinned: the posed group's box as keys, then a copy of the build's Ctrl
  mesh::Ctrl* h_ctrl = nullptr;
  bool posed = false;
  bool acc_clean = false;     // every accumulator is empty (true after a complete build)
  uint32_t last_levels = 0;   // levels of the previous frame's tree: how many to launch before looking
  uint32_t cur = 0;           // which order buffer holds the finished order
  double bbox[6] = {};
  ~rt_mesh() {
    if (stream) {
      (void)hipStreamSynchronize(stream);
      (void)hipStreamDestroy(stream);
    }
    if (block) (void)hipFree(block);
    if (h_box) (void)hipHostFree(h_box);
  }
};

struct rt_scene {
  DevScene dev;
  // every scene table, the statistics counters and the summary word live in ONE device block (rt_scene_create)
  void* arena = nullptr;
  size_t arena_cap = 0;
  struct StageBuffer* stage = nullptr;             // while rt_scene_create runs: the arena's contents on the host
  std::vector<std::pair<size_t, void*>> fixups;    // (offset in the arena, address of the device pointer to set)
  std::vector<std::pair<size_t, size_t>> reserved; // [begin, end) of the arena image that the DEVICE fills (mesh parts of tables): not uploaded
  const uint32_t* mesh_mat_map = nullptr;          // the meshes' palettes as this scene's material indices / flags (rt_mesh.h MeshOut)
  const uint8_t* mesh_mat_flags = nullptr;
  const char* stage_base = nullptr;                // the staging buffer while tables are filled in place: it must not move
  size_t table_bytes[7] = {};                      // rt_scene_debug_table: tri_isect, tri_norm, prim_rank, prim_info, entries, nodes, prim_flags
  int device = 0;
  void* noalloc_scratch = nullptr;          // this device's copy of the `scratch` array (rt_render_device_noalloc)
  mutable hipEvent_t done_event = nullptr;  // recorded after every render: what the destructor waits for
  // the shadow rays of a level run on a stream of the scene's own, beside the following levels (rt_stream.h)
  hipStream_t side_stream = nullptr;
  hipEvent_t level_event[RT_MAX_DEPTH + 1] = {};  // level l's node records are complete (recorded on the render's stream)
  hipEvent_t side_event = nullptr;                // the side stream's shadow kernels of a batch have all been enqueued and done
  mutable bool rendered = false;
  unsigned long long* d_stats = nullptr;  // RT_N_COUNTERS counters
  uint32_t* d_overflow = nullptr;
  size_t bytes = 0;
  uint32_t stack_need = 0;
  uint32_t n_lights = 0;
  bool general_csg = false;  // a CSG object of the scene is beyond what the production kernels evaluate (more than RT_CSG_LIST_MAX
                             // intersections in the worst case, nesting beyond RT_CSG_NEST_MAX, a CSG object inside a group that
                             // is a CSG operand; or a CSG operand that is not a closed surface, is_solid) — or the scene holds a
                             // primitive that sticks out of its own bounding box (a cone's cap at |y| < 1, validate) —: every render of the scene runs the reference-order kernel with the general
                             // evaluation (rt_shapes.h csg_survivors_general) — several times slower, never refused
  bool has_csg = false;  // the scene holds CSG objects: the trace kernels built with their evaluation are launched
  bool has_walks = true; // some primitive's closest hit is followed by the container walk (DevScene::prim_flags)
  bool one_kernel = false;  // no material reflects or transmits (every ray tree is a primary ray and its shadow rays) and the
                            // scene holds no triangles: RT_PIPELINE_STREAM renders it with the single-kernel core, whose one
                            // launch beats the streamed pipeline's six (README scene 800x800: 0.10 against 0.37 ms)
  int n_cus = 256;
  // workspace of the streamed pipeline: allocated at the first render, kept for the scene's lifetime, grown on demand
  mutable void* ws_base = nullptr;
  mutable size_t ws_bytes = 0;
  mutable uint32_t ws_samples = 0;  // batch capacity in samples
  mutable stream::Ws ws{};
  // node records per sample the batches are laid out for.  A sample whose ray tree needs more is re-rendered by the
  // fallback kernel (same pixels, slower); the count of such events comes back with every frame and doubles this.
  mutable uint32_t nodes_per_sample = 3;
  mutable uint32_t ratio_floor = 1;  // smallest ratio worth trying again (one more than the last that overflowed)
  mutable bool layout_learned = false;  // a frame of this scene has reported back (absorb_summary)
  uint64_t shape = 0;                   // what g_learned files this scene under
  mutable uint32_t* d_summary = nullptr;  // [0] fallback flags of the frame in flight, [1] / [2] node records / rays of a level used per sample, x 16
  mutable uint32_t* h_summary = nullptr;  // pinned copy, valid once summary_event has completed
  mutable hipEvent_t summary_event = nullptr;
  mutable bool summary_pending = false;
  // rt_options.timing: HIP events around every launch of the streamed pipeline, by kernel class
  mutable bool timing_on = false;
  mutable uint32_t timing_classes = ~0u;  // bit c: launches of class c are timed (RT_TIMING_ONLY)
  mutable std::vector<hipEvent_t> event_pool;
  mutable std::vector<int> timed_class;  // class of the i-th timed launch (events 2i, 2i+1 of the pool)
  ~rt_scene() {
    // the blocks go back to the pool, for other scenes to use at once: first wait for this scene's own GPU work
    if (rendered && done_event) (void)hipEventSynchronize(done_event);
    if (side_stream) (void)hipStreamSynchronize(side_stream);  // (a batch that failed half-way may have left a shadow kernel there)
    if (summary_pending && summary_event) (void)hipEventSynchronize(summary_event);
    if (layout_learned) g_learned.remember(device, shape, nodes_per_sample, ratio_floor);
    g_device_pool.give(ws_base, ws_bytes, device);
    g_device_pool.give(arena, arena_cap, device);
    g_pinned_slots.give(h_summary);
    if (summary_event) (void)hipEventDestroy(summary_event);
    if (done_event) (void)hipEventDestroy(done_event);
    if (side_event) {
      (void)hipEventSynchronize(side_event);
      (void)hipEventDestroy(side_event);
    }
    for (hipEvent_t e : level_event)
      if (e) (void)hipEventDestroy(e);
    if (side_stream) (void)hipStreamDestroy(side_stream);
    for (hipEvent_t e : event_pool) (void)hipEventDestroy(e);
  }
};

namespace {

// Adds a table to the scene's arena image; the device pointer is set by commit_arena.
// (the image is built in a buffer that the thread keeps: a fresh 50-MB vector costs more in page faults, ~20 ms, than
// filling and copying it)
thread_local StageBuffer t_stage;
// tail: bytes behind the table that the device fills (the mesh parts, rt_mesh.h): part of the table, not of the upload.
size_t stage_bytes(rt_scene* sc, const void* data, size_t bytes, void* pointer_to_set, size_t tail = 0) {
  StageBuffer& st = *sc->stage;
  size_t off = (st.size + 255) & ~(size_t)255, len = std::max<size_t>(bytes, 16);
  char* p = st.resize(off + std::max(len, bytes + tail));
  if (!p) return 0;  // commit_arena reports it
  if (bytes && data) std::memcpy(p + off, data, bytes);
  else std::memset(p + off, 0, len);  // no data: zeros
  if (tail) sc->reserved.push_back({off + bytes, off + bytes + tail});  // (element `bytes / sizeof(T)` of the table is the tail's first)
  sc->fixups.push_back({off, pointer_to_set});
  sc->bytes += bytes + tail;
  return off;
}
// A table that is filled where it will be uploaded from (no vector of its own: a fresh 20-MB vector costs its page faults,
// and then a copy).  rt_scene_create has reserved the buffer's capacity, so the pointer stays valid until commit_arena.
template <class T>
T* stage_inplace(rt_scene* sc, size_t count, const T** out, size_t tail_elements = 0) {
  StageBuffer& st = *sc->stage;
  const size_t bytes = count * sizeof(T), tail = tail_elements * sizeof(T);
  size_t off = (st.size + 255) & ~(size_t)255, len = std::max<size_t>(bytes, 16);
  char* p = st.resize(off + std::max(len, bytes + tail));
  if (!p) return nullptr;
  if (!bytes) std::memset(p + off, 0, len);
  if (tail) sc->reserved.push_back({off + bytes, off + bytes + tail});
  sc->fixups.push_back({off, (void*)out});
  sc->bytes += bytes + tail;
  return (T*)(p + off);
}
template <class T>
int upload(rt_scene* sc, const std::vector<T>& host, const T** out, size_t tail_elements = 0) {
  stage_bytes(sc, host.data(), host.size() * sizeof(T), (void*)out, tail_elements * sizeof(T));
  return RT_OK;
}
// One device block for everything staged, one copy (on a stream of its own: a copy on the null stream would wait for, and
// hold up, renders of other scenes that other threads have in flight).
int commit_arena(rt_scene* sc) {
  StageBuffer& st = *sc->stage;
  if (!st.p) return fail(RT_ERR_DEVICE, "out of host memory while staging the scene tables");
  if (sc->stage_base && sc->stage_base != st.p) return fail(RT_ERR_HOST, "internal error: the staging buffer moved under tables filled in place");
  size_t need = (st.size + 255) & ~(size_t)255;
  sc->arena = g_device_pool.take(need, sc->device, &sc->arena_cap);
  if (!sc->arena) return fail(RT_ERR_DEVICE, "hipMalloc of the scene tables (" + std::to_string(need >> 20) + " MiB) failed");
  hipStream_t up = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
  auto t0 = std::chrono::steady_clock::now();
  // everything but the ranges the device fills itself
  hipError_t e = hipSuccess;
  size_t at = 0;
  for (size_t r = 0; r <= sc->reserved.size() && e == hipSuccess; r++) {
    const size_t end = r < sc->reserved.size() ? sc->reserved[r].first : st.size;
    if (end > at) e = hipMemcpyWithStream((char*)sc->arena + at, st.p + at, end - at, hipMemcpyHostToDevice, up);
    if (r < sc->reserved.size()) at = sc->reserved[r].second;
  }
  if (e == hipSuccess) e = hipStreamSynchronize(up);
  if (std::getenv("RT_DEBUG_TIMING"))
    fprintf(stderr, "[rtcore] arena copy %zu bytes: %.2f ms\n", st.size,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  (void)hipStreamDestroy(up);
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("upload of the scene tables failed: ") + hipGetErrorString(e));
  for (auto& f : sc->fixups) {
    void* p = (char*)sc->arena + f.first;
    std::memcpy(f.second, &p, sizeof(p));
  }
  st.size = 0;  // capacity kept
  sc->stage = nullptr;
  sc->fixups.clear();
  sc->reserved.clear();
  return RT_OK;
}

bool ref_ok(const rt_scene_desc* d, uint32_t ref) {
  uint32_t kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
  if (kind == RT_REF_NODE) return idx < d->n_nodes;
  if (kind == RT_REF_PRIM) return idx < d->n_prims;
  if (kind == RT_REF_CSG) return idx < d->n_csgs;
  if (kind == RT_REF_MESH) return idx < d->n_meshes;
  return false;
}

// Stack entries of the reference-order walk (rt_reforder.h) a mesh's own tree can hold, by mesh: known once the device has
// built it (0 before: rt_scene_create checks the rest of the scene first and the whole again afterwards).
struct MeshNeeds {
  std::vector<uint32_t> sub;  // [n_meshes]
};

// Worst-case traversal stack (group.rs:176-199): while child i of a node is being processed, its later
// group siblings are still on the stack.
uint32_t stack_need(const rt_scene_desc* d, uint32_t node, std::vector<int64_t>& memo, int guard, const MeshNeeds* meshes = nullptr) {
  if (memo[node] >= 0) return (uint32_t)memo[node];
  if (guard > 100000) return 1u << 30;
  uint32_t b = d->node_child_begin[node], e = d->node_child_begin[node + 1];
  std::vector<uint32_t> groups;  // (child references: groups and meshes, which the walk enters like groups)
  for (uint32_t i = b; i < e; i++) {
    const uint32_t kind = d->child_ref[i] >> RT_REF_KIND_SHIFT;
    if (kind == RT_REF_NODE || kind == RT_REF_MESH) groups.push_back(d->child_ref[i]);
  }
  uint32_t k = (uint32_t)groups.size(), need = k;
  for (uint32_t i = 0; i < k; i++) {
    uint32_t sub = (groups[i] >> RT_REF_KIND_SHIFT) == RT_REF_MESH
                       ? (meshes ? meshes->sub[groups[i] & RT_REF_INDEX_MASK] : 0u)
                       : stack_need(d, groups[i] & RT_REF_INDEX_MASK, memo, guard + 1, meshes);
    uint32_t v = (k - 1 - i) + sub;
    if (v > need) need = v;
  }
  memo[node] = need;
  return need;
}

// ---------------------------------------------------------------------------------------------
// Device layout of the fast path (rt_device.h): binary 128-B nodes, leaf runs, visit ranks.
// ---------------------------------------------------------------------------------------------
struct BinNode {  // the binary tree before the fold: exact f64 boxes of both children
  double bb[2][6];
  uint32_t ref[2];
};

struct FastLayout {
  std::vector<BinNode> bin;    // node 0 is a synthetic root whose child 0 is the scene's root group
  std::vector<DevNode> nodes;  // the 4-wide nodes the device walks (fold_nodes)
  uint32_t stack_need = 0;     // most entries the device traversal stack can hold at once
  std::vector<uint32_t> leaf_ref;
  std::vector<uint32_t> prim_rank;
  std::vector<uint32_t> prim_tri;  // triangle primitives: slot in the device triangle tables (leaf order), else RT_NONE
  uint32_t n_dev_tris = 0;
  std::vector<uint32_t> mesh_rank0;  // [n_meshes] visit rank of each mesh's first-visited triangle (assign_ranks)
  std::vector<uint32_t> mesh_held;   // [n_meshes] stack entries held when the mesh's root node is visited (fold_nodes)
  uint32_t depth = 0;
  uint32_t leaf_tris = 2;      // all-triangle subtrees of at most this many triangles are one leaf (<= 2: the reference's own
                               // leaves).  4 was best while a node step cost 30 instructions per box (r01: dragons 4K 54.5 ->
                               // 53.0 ms); with the f32 slab test 2 is: 34.6 -> 34.0 ms (3: 34.2, 6: 35.4, 8: 36.5; the one-dragon
                               // 800x800 frame 3.20 -> 3.27 ms).  RT_LEAF_TRIS in the environment overrides.
  std::string error;
};

#if RT_QNODE
// Quantises the boxes of a wide node's children onto the node's grid (rt_device.h): every quantised box contains the
// reference's f64 box (checked in exact arithmetic: origin + q * step is exact in f64 by the choice of the step).
// false: a coordinate is not finite or beyond 1e18 (the reference clamps infinite shapes to 1e7, bounding_box.rs:154-157).
bool quantize_node(DevNode& n, const double (*bb)[6], int count) {
  n.exps = 0;
  for (int k = 0; k < 6; k++) n.q[k] = 0;
  for (int a = 0; a < 3; a++) {
    double lo = INFINITY, hi = -INFINITY;
    for (int c = 0; c < count; c++) {
      if (!(std::fabs(bb[c][a]) <= 1e18) || !(std::fabs(bb[c][a + 3]) <= 1e18)) return false;
      lo = std::min(lo, bb[c][a]);
      hi = std::max(hi, bb[c][a + 3]);
    }
    if (count == 0) lo = hi = 0.;
    float org = (float)lo;
    if ((double)org > lo) org = std::nextafterf(org, -INFINITY);
    // step = 2^e: 255 steps reach the largest maximum; never finer than 2^-40 of the origin (so that origin + q * step is
    // exact in f64) nor than 2^-100
    int e = -100;
    if (org != 0.f) e = std::max(e, std::ilogb((double)org) - 40);
    double extent = hi - (double)org;
    if (extent > 0.) e = std::max(e, (int)std::ceil(std::log2(extent / 255.)) - 1);
    for (;; e++) {
      const double step = std::ldexp(1., e);
      bool ok = true;
      uint32_t ql[RT_NODE_WIDTH] = {}, qh[RT_NODE_WIDTH] = {};
      for (int c = 0; c < count && ok; c++) {
        double l = std::floor((bb[c][a] - (double)org) / step), h = std::ceil((bb[c][a + 3] - (double)org) / step);
        l = std::min(std::max(l, 0.), 255.);
        h = std::max(h, 0.);
        while (l > 0. && (double)org + l * step > bb[c][a]) l -= 1.;
        while (h <= 255. && (double)org + h * step < bb[c][a + 3]) h += 1.;
        if (h > 255. || (double)org + l * step > bb[c][a]) ok = false;
        ql[c] = (uint32_t)l;
        qh[c] = (uint32_t)h;
      }
      if (!ok) continue;
      if (e + 127 < 1 || e + 127 > 254) return false;
      n.org[a] = org;
      n.exps |= (uint32_t)(e + 127) << (8 * a);
      for (int c = 0; c < count; c++) {
        n.q[a] |= ql[c] << (8 * c);
        n.q[a + 3] |= qh[c] << (8 * c);
      }
      break;
    }
  }
  return true;
}
#else
// The reference's f64 box, rounded outward to the node's f32 fields (min down, max up): a superset of the box.
void set_node_box(DevNode& n, int child, const double* bb) {
  for (int k = 0; k < 6; k++) {
    float f = (float)bb[k];
    if (k < 3) {
      if ((double)f > bb[k]) f = std::nextafterf(f, -INFINITY);
    } else {
      if ((double)f < bb[k]) f = std::nextafterf(f, INFINITY);
    }
    n.bb[child][k] = f;
  }
}
#endif

// In the layout passes a mesh child stands as this reference (kind EMPTY, never RT_FREF_NONE); rt_scene_create replaces it by
// the mesh's root node once the node table's size is known.
#define RT_MESH_MARKER(k) RT_MAKE_FREF(RT_FREF_EMPTY, 1u, (k))
#define RT_IS_MESH_MARKER(r) ((r) != RT_FREF_NONE && ((r) >> RT_FREF_KIND_SHIFT) == RT_FREF_EMPTY)

struct FastItem {  // how a parent refers to one child: reference + the box tested before entering it
  uint32_t ref;
  double bb[6];
};

bool is_triangle_prim(const rt_scene_desc* d, uint32_t prim) {
  uint8_t sh = d->prim_shape[prim];
  return (sh == RT_SHAPE_TRIANGLE || sh == RT_SHAPE_SMOOTH_TRIANGLE) && d->prim_xform[prim] == RT_NONE;
}

// Visit order of ObjectGroup::intersect_iter (group.rs:171-201): a popped group handles its children in reverse,
// primitives and CSG objects at once, sub-groups pushed (so they pop in forward order).  Every leaf primitive of a
// CSG object shares the CSG's rank (its survivors reach the collector in one push sequence, csg.rs:119-121).
// A mesh (RT_REF_MESH) is a group: it is pushed like one, and when it pops its triangles take the next n ranks (in which
// order among themselves: rt_mesh.h finalize_kernel).
uint32_t mesh_tris(const rt_scene_desc* d, uint32_t k);
void assign_ranks(const rt_scene_desc* d, std::vector<uint32_t>& rank, std::vector<uint32_t>& mesh_rank0) {
  rank.assign(d->n_prims, 0xFFFFFFFFu);
  mesh_rank0.assign(d->n_meshes, 0u);
  uint32_t next = 0;
  std::vector<uint32_t> stack{0};  // group indices; bit 31: a mesh
  std::vector<uint32_t> csg_stack;
  while (!stack.empty()) {
    uint32_t g = stack.back();
    stack.pop_back();
    if (g & 0x80000000u) {
      const uint32_t k = g & 0x7FFFFFFFu;
      mesh_rank0[k] = next;
      next += mesh_tris(d, k);
      continue;
    }
    uint32_t b = d->node_child_begin[g], e = d->node_child_begin[g + 1];
    for (uint32_t i = e; i-- > b;) {
      uint32_t ref = d->child_ref[i], kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
      if (kind == RT_REF_NODE) {
        stack.push_back(idx);
      } else if (kind == RT_REF_MESH) {
        stack.push_back(idx | 0x80000000u);
      } else if (kind == RT_REF_PRIM) {
        rank[idx] = next++;
      } else {
        uint32_t r = next++;
        csg_stack.assign(1, idx);
        while (!csg_stack.empty()) {
          uint32_t c = csg_stack.back();
          csg_stack.pop_back();
          for (uint32_t cr : {d->csg_left[c], d->csg_right[c]}) {
            uint32_t ck = cr >> RT_REF_KIND_SHIFT, ci = cr & RT_REF_INDEX_MASK;
            if (ck == RT_REF_PRIM) rank[ci] = r;
            else if (ck == RT_REF_CSG) csg_stack.push_back(ci);
            else {  // a group operand: every primitive below it (check_graph: no CSG objects there)
              std::vector<uint32_t> groups{ci};
              while (!groups.empty()) {
                uint32_t gg = groups.back();
                groups.pop_back();
                for (uint32_t k = d->node_child_begin[gg]; k < d->node_child_begin[gg + 1]; k++) {
                  uint32_t gr = d->child_ref[k], gk = gr >> RT_REF_KIND_SHIFT, gi = gr & RT_REF_INDEX_MASK;
                  if (gk == RT_REF_PRIM) rank[gi] = r;
                  else if (gk == RT_REF_NODE) groups.push_back(gi);
                }
              }
            }
          }
        }
      }
    }
  }
}

// A vector that lives on the stack until it outgrows N elements: rt_scene_create walks ~2 groups per triangle, and three
// heap vectors per group were a third of a scene's creation time.
template <class T, int N>
struct SmallVec {
  T inl[N];
  std::vector<T> big;
  size_t n = 0;
  void push_back(const T& v) {
    if (n < (size_t)N) {
      inl[n++] = v;
      return;
    }
    if (n == (size_t)N) big.assign(inl, inl + N);
    big.push_back(v);
    n++;
  }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  const T* data() const { return n <= (size_t)N ? inl : big.data(); }
  const T& operator[](size_t i) const { return data()[i]; }
};
typedef SmallVec<uint32_t, 8> RefList;
typedef SmallVec<FastItem, 8> ItemList;

// Emits the reference(s) for a run of direct primitive/CSG children of one group, tested behind `bb` (the group's box).
// More than RT_FREF_LEAF_MAX of them become several leaves.
template <class Refs, class Items>
void make_leaves(const rt_scene_desc* d, FastLayout& L, const Refs& refs, const double* bb, Items& out) {
  bool all_tris = !refs.empty();
  for (size_t ri = 0; ri < refs.size(); ri++) {
    const uint32_t r = refs[ri];
    if ((r >> RT_REF_KIND_SHIFT) != RT_REF_PRIM || !is_triangle_prim(d, r & RT_REF_INDEX_MASK) || L.prim_tri[r & RT_REF_INDEX_MASK] != RT_NONE)
      all_tris = false;
  }
  for (size_t at = 0; at < refs.size(); at += RT_FREF_LEAF_MAX) {
    uint32_t n = (uint32_t)std::min<size_t>(RT_FREF_LEAF_MAX, refs.size() - at);
    FastItem it;
    std::memcpy(it.bb, bb, sizeof(it.bb));
    if (all_tris) {
      it.ref = RT_MAKE_FREF(RT_FREF_TRIS, n, L.n_dev_tris);
      for (uint32_t k = 0; k < n; k++) L.prim_tri[refs[at + k] & RT_REF_INDEX_MASK] = L.n_dev_tris++;
    } else {
      it.ref = RT_MAKE_FREF(RT_FREF_PRIMS, n, (uint32_t)L.leaf_ref.size());
      for (uint32_t k = 0; k < n; k++) L.leaf_ref.push_back(refs[at + k]);
    }
    out.push_back(it);
  }
}

// Folds items into binary nodes; returns the reference to the (sub)tree holding all of them.  Synthetic inner
// references are tested behind `bb` again (a box the ray is already known to hit).
template <class Items>
uint32_t fold_items(FastLayout& L, const Items& items, size_t from, const double* bb) {
  uint32_t id = (uint32_t)L.bin.size();
  L.bin.emplace_back();
  BinNode n;
  std::memset(&n, 0, sizeof(n));
  std::memcpy(n.bb[0], items[from].bb, sizeof(n.bb[0]));
  n.ref[0] = items[from].ref;
  size_t left = items.size() - from;
  if (left == 1) {
    n.ref[1] = RT_FREF_NONE;
  } else if (left == 2) {
    std::memcpy(n.bb[1], items[from + 1].bb, sizeof(n.bb[1]));
    n.ref[1] = items[from + 1].ref;
  } else {
    std::memcpy(n.bb[1], bb, sizeof(n.bb[1]));
    n.ref[1] = fold_items(L, items, from + 1, bb);
  }
  L.bin[id] = n;
  return RT_MAKE_FREF(RT_FREF_INNER, 0, id);
}

// Collects the primitives below group `g` (in the reference's child order) if they are at most `limit` plain triangles.
bool small_triangle_subtree(const rt_scene_desc* d, const FastLayout& L, uint32_t g, uint32_t limit, std::vector<uint32_t>& out) {
  uint32_t b = d->node_child_begin[g], e = d->node_child_begin[g + 1];
  for (uint32_t i = b; i < e; i++) {
    uint32_t ref = d->child_ref[i], kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
    if (kind == RT_REF_NODE) {
      if (!small_triangle_subtree(d, L, idx, limit, out)) return false;
    } else if (kind == RT_REF_PRIM && is_triangle_prim(d, idx) && L.prim_tri[idx] == RT_NONE) {
      out.push_back(ref);
      if (out.size() > limit) return false;
    } else {
      return false;
    }
  }
  return true;
}

// How the parent of group `g` refers to it (the box is g's own, tested by the parent).
uint32_t build_group(const rt_scene_desc* d, FastLayout& L, uint32_t g, uint32_t depth) {
  if (depth > L.depth) L.depth = depth;
  if (depth > 4096) {
    L.error = "group nesting deeper than 4096";
    return RT_FREF_NONE;
  }
  const double* bb = d->node_bbox + (size_t)g * 6;
  uint32_t b = d->node_child_begin[g], e = d->node_child_begin[g + 1];
  if (L.leaf_tris > 2) {  // a small all-triangle subtree becomes ONE leaf behind the group's box
    std::vector<uint32_t> all;
    if (small_triangle_subtree(d, L, g, L.leaf_tris, all) && !all.empty()) {
      std::vector<FastItem> one;
      make_leaves(d, L, all, bb, one);
      if (one.size() == 1) return one[0].ref;
    }
  }
  RefList direct, groups;
  for (uint32_t i = b; i < e; i++) {
    uint32_t ref = d->child_ref[i], kind = ref >> RT_REF_KIND_SHIFT;
    if (kind == RT_REF_NODE || kind == RT_REF_MESH) groups.push_back(ref);
    else direct.push_back(ref);
  }
  ItemList items;
  make_leaves(d, L, direct, bb, items);
  if (groups.empty() && items.size() == 1) return items[0].ref;  // a plain leaf group: no node of its own
  for (size_t gi = 0; gi < groups.size(); gi++) {
    const uint32_t c = groups[gi] & RT_REF_INDEX_MASK;
    FastItem it;
    if ((groups[gi] >> RT_REF_KIND_SHIFT) == RT_REF_MESH) {  // its subtree is built on the device: a marker, resolved after the fold
      std::memcpy(it.bb, d->mesh_bbox + (size_t)c * 6, sizeof(it.bb));
      it.ref = RT_MESH_MARKER(c);
      items.push_back(it);
      continue;
    }
    std::memcpy(it.bb, d->node_bbox + (size_t)c * 6, sizeof(it.bb));
    it.ref = build_group(d, L, c, depth + 1);
    if (!L.error.empty()) return RT_FREF_NONE;
    items.push_back(it);
  }
  if (items.empty()) return RT_FREF_NONE;  // empty group
  if (items.size() > 2 && depth + items.size() > L.depth) L.depth = depth + (uint32_t)items.size();
  return fold_items(L, items, 0, bb);
}

// Folds the binary tree into RT_NODE_WIDTH-wide nodes: the slots of a wide node start as the two children of a binary
// node; while a slot is free, the inner child with the largest box is replaced by ITS two children (its own box is then
// never tested: the children's boxes are tested directly).  Also finds the deepest the traversal stack can get: a
// visit of a node with k children pushes at most k - 1 of them.
struct FoldTodo {
  uint32_t bin, wide, held;  // binary node to fold, its wide node (index into `nodes`), stack entries held while inside it
};
// Folds the subtrees on `todo` into `nodes` until none is left — or, limit > 0, until at least `limit` subtrees wait (which the
// caller then hands to threads).
void fold_some(FastLayout& L, std::vector<DevNode>& nodes, std::vector<FoldTodo>& todo, uint32_t& stack_need, size_t limit) {
  struct Slot {
    double bb[6];
    uint32_t ref;
  };
  auto inner = [](uint32_t r) { return r != RT_FREF_NONE && (r >> RT_FREF_KIND_SHIFT) == RT_FREF_INNER; };
  auto area = [](const double* b) {
    double dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    double a = dx * dy + dy * dz + dz * dx;
    return a == a ? a : INFINITY;
  };
  std::vector<Slot> slots;  // (one allocation for all nodes)
  slots.reserve(2 * RT_NODE_WIDTH);
  while (!todo.empty()) {
    if (limit && todo.size() >= limit) return;
    // (RT_LDS_NODES: the first nodes are numbered breadth-first — the top levels of the tree —, the rest depth-first)
    const bool breadth_first = nodes.size() < (size_t)RT_LDS_NODES;
    FoldTodo t = breadth_first ? todo.front() : todo.back();
    if (breadth_first) todo.erase(todo.begin());
    else todo.pop_back();
    slots.clear();
    auto add_children = [&](uint32_t b) {
      for (int c = 0; c < 2; c++) {
        if (L.bin[b].ref[c] == RT_FREF_NONE) continue;
        Slot sl;
        std::memcpy(sl.bb, L.bin[b].bb[c], sizeof(sl.bb));
        sl.ref = L.bin[b].ref[c];
        slots.push_back(sl);
      }
    };
    add_children(t.bin);
    for (;;) {
      int best = -1;
      for (size_t i = 0; i < slots.size(); i++) {
        if (!inner(slots[i].ref)) continue;
        const BinNode& bn = L.bin[slots[i].ref & RT_FREF_INDEX_MASK];
        size_t kids = (bn.ref[0] != RT_FREF_NONE) + (bn.ref[1] != RT_FREF_NONE);
        if (slots.size() - 1 + kids > RT_NODE_WIDTH) continue;
        if (best < 0 || area(slots[i].bb) > area(slots[best].bb)) best = (int)i;
      }
      if (best < 0) break;
      uint32_t b = slots[best].ref & RT_FREF_INDEX_MASK;
      slots.erase(slots.begin() + best);
      add_children(b);
    }
    DevNode n;
    std::memset(&n, 0, sizeof(n));
    uint32_t held = t.held + (slots.empty() ? 0u : (uint32_t)slots.size() - 1u);
    if (held > stack_need) stack_need = held;
#if RT_QNODE
    {
      double boxes[RT_NODE_WIDTH][6];
      for (size_t i = 0; i < slots.size(); i++) std::memcpy(boxes[i], slots[i].bb, sizeof(boxes[i]));
      if (!quantize_node(n, boxes, (int)slots.size())) {
        L.error = "a bounding box coordinate is not finite or beyond 1e18";
        return;
      }
    }
#endif
    for (int i = 0; i < RT_NODE_WIDTH; i++) {
      if ((size_t)i >= slots.size()) {
        n.ref[i] = RT_FREF_NONE;
        continue;
      }
#if !RT_QNODE
      set_node_box(n, i, slots[i].bb);
#endif
      if (inner(slots[i].ref)) {
        uint32_t wide = (uint32_t)nodes.size();
        nodes.emplace_back();
        n.ref[i] = RT_MAKE_FREF(RT_FREF_INNER, 0, wide);
        todo.push_back({slots[i].ref & RT_FREF_INDEX_MASK, wide, held});
      } else {
        n.ref[i] = slots[i].ref;
        if (RT_IS_MESH_MARKER(slots[i].ref)) L.mesh_held[slots[i].ref & RT_FREF_INDEX_MASK] = held;
      }
    }
    nodes[t.wide] = n;
  }
}

// The whole tree: the top of it on this thread until a handful of subtrees wait, then those on up to four threads, each into
// a table of its own (its node 0 = the subtree's root, whose place in L.nodes is taken already), appended to L.nodes
// afterwards with their references moved (6.2 -> 2 ms for dragons.yml's 40 000 wide nodes).
void fold_nodes(FastLayout& L) {
  L.nodes.clear();
  L.nodes.emplace_back();
  L.nodes.reserve(L.bin.size() / 2 + 16);
  std::vector<FoldTodo> todo{{0u, 0u, 0u}};
  L.stack_need = 0;
  const bool threaded = L.bin.size() >= 16384 && RT_LDS_NODES == 0 && !RT_QNODE;
  fold_some(L, L.nodes, todo, L.stack_need, threaded ? 8 : 0);
  if (todo.empty() || !L.error.empty()) return;
  const size_t n_sub = todo.size();
  std::vector<std::vector<DevNode>> local(n_sub);
  std::vector<uint32_t> need(n_sub, 0u);
  std::atomic<size_t> next{0};
  auto work = [&] {
    for (size_t k = next.fetch_add(1); k < n_sub; k = next.fetch_add(1)) {
      local[k].emplace_back();
      std::vector<FoldTodo> mine{{todo[k].bin, 0u, todo[k].held}};
      fold_some(L, local[k], mine, need[k], 0);
    }
  };
  {
    std::vector<std::thread> workers;
    JoinAll join{workers};
    for (int t = 1; t < 4; t++) workers.emplace_back(work);
    work();
    for (std::thread& t : workers) t.join();
  }
  for (size_t k = 0; k < n_sub; k++) {
    const uint32_t base = (uint32_t)L.nodes.size();  // local node i >= 1 -> base + i - 1
    for (DevNode& n : local[k])
      for (int c = 0; c < RT_NODE_WIDTH; c++)
        if (n.ref[c] != RT_FREF_NONE && (n.ref[c] >> RT_FREF_KIND_SHIFT) == RT_FREF_INNER)
          n.ref[c] = RT_MAKE_FREF(RT_FREF_INNER, 0, base + (n.ref[c] & RT_FREF_INDEX_MASK) - 1u);
    L.nodes[todo[k].wide] = local[k][0];
    L.nodes.insert(L.nodes.end(), local[k].begin() + 1, local[k].end());
    L.stack_need = std::max(L.stack_need, need[k]);
  }
}

uint32_t mesh_tris(const rt_scene_desc* d, uint32_t k) { return d->mesh[k]->n; }

bool build_fast_layout(const rt_scene_desc* d, FastLayout& L) {
  // (L.prim_rank / L.mesh_rank0 — assign_ranks — are filled by rt_scene_create's own thread, beside this one)
  L.mesh_held.assign(d->n_meshes, 0u);
  L.prim_tri.assign(d->n_prims, RT_NONE);
  if (const char* e = getenv("RT_LEAF_TRIS")) L.leaf_tris = (uint32_t)std::min(std::max(atoi(e), 1), (int)RT_FREF_LEAF_MAX);
  L.bin.reserve((size_t)d->n_nodes + 2);
  L.bin.emplace_back();  // node 0: the synthetic root
  const bool debug_timing = std::getenv("RT_DEBUG_TIMING") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  uint32_t root = build_group(d, L, 0, 1);
  if (!L.error.empty()) return false;
  if (debug_timing)
    fprintf(stderr, "[rtcore]   (layout thread) binary tree of %zu nodes: %.2f ms\n", L.bin.size(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  BinNode n;
  std::memset(&n, 0, sizeof(n));
  std::memcpy(n.bb[0], d->node_bbox, sizeof(n.bb[0]));
  n.ref[0] = root;
  n.ref[1] = RT_FREF_NONE;
  L.bin[0] = n;
  fold_nodes(L);
  if (!L.error.empty()) return false;
  if (debug_timing)
    fprintf(stderr, "[rtcore]   (layout thread) + fold into %zu wide nodes: %.2f ms\n", L.nodes.size(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  // triangles outside triangle-only leaves (mixed leaves, CSG operands) still need a slot in the triangle tables
  for (uint32_t i = 0; i < d->n_prims; i++) {
    uint8_t sh = d->prim_shape[i];
    if ((sh == RT_SHAPE_TRIANGLE || sh == RT_SHAPE_SMOOTH_TRIANGLE) && L.prim_tri[i] == RT_NONE) L.prim_tri[i] = L.n_dev_tris++;
  }
  if (L.nodes.size() > RT_FREF_INDEX_MASK || L.n_dev_tris > RT_FREF_INDEX_MASK || L.leaf_ref.size() > RT_FREF_INDEX_MASK) {
    L.error = "scene too large for 24-bit node/triangle indices";
    return false;
  }
  return true;
}


// Most roots one primitive can add to a collector (sphere.rs:27-34, plane.rs:13-19, cube.rs:25-42, cylinder.rs:54-120,
// cone.rs:56-127, triangle.rs:48-71): what bounds a CSG object's hit list.
uint32_t max_roots(const rt_scene_desc* d, uint32_t prim) {
  switch (d->prim_shape[prim]) {
    case RT_SHAPE_PLANE:
    case RT_SHAPE_TRIANGLE:
    case RT_SHAPE_SMOOTH_TRIANGLE: return 1;
    case RT_SHAPE_CYLINDER:
    case RT_SHAPE_CONE: return d->quadric_closed[d->prim_geom[prim]] ? 4 : 2;
    default: return 2;
  }
}

// A shape that a line crosses an even number of times.  CsgObject::filter_intersections (csg.rs:87-110) takes "inside the left
// / right operand" from the PARITY of the operand's intersections so far, and a CSG object's bounding box is put together from
// its operands' boxes (csg.rs:131-143: the left one's for a difference, the boxes' intersection for an intersection).  With an
// operand that is not a closed surface — a plane, a triangle, a cylinder or cone without caps — the parity says "inside" where
// the ray is far outside the operand, so the object reports intersections OUTSIDE its own box (found by tools/fuzz_csg.py,
// seed 6384: a cone without caps minus a group).  The reference then still finds them whenever the LINE of the ray crosses the
// enclosing group's box, also behind the origin (BoundingBox::intersection_time, bounding_box.rs:192-225, has no "behind"
// test), and which of them it finds depends on its visiting order (group.rs:182 skips by entry time).  The production
// traversal relies on geometry lying inside its boxes (it passes over boxes behind the origin and visits nearest first): a
// scene with such an operand is rendered by the reference-order kernels (rt_scene::general_csg), which walk exactly as the
// reference does.
bool is_solid(const rt_scene_desc* d, uint32_t prim) {
  switch (d->prim_shape[prim]) {
    case RT_SHAPE_SPHERE:
    case RT_SHAPE_CUBE:
    case RT_SHAPE_BBOX: return true;
    case RT_SHAPE_CYLINDER:
    case RT_SHAPE_CONE: return d->quadric_closed[d->prim_geom[prim]] != 0;
    default: return false;
  }
}

// The object graph reachable from the root must be a tree (the reference's objects own their children): no reference
// back to an ancestor — the layout passes below recurse over it —, nothing referenced twice.  Also checked here, because
// they need the same walk: csg_depth is the real CSG nesting; what the device's CSG evaluation is bounded by
// (RT_CSG_LIST_MAX roots per outermost CSG object in the worst case, so a render can never overflow its hit list; group
// operands of at most RT_CSG_GROUP_STACK pending sub-groups, with no CSG object inside them).  Iterative: the input may be
// arbitrarily deep, and nothing may overflow the host stack on behalf of a malformed scene.
int check_graph(const rt_scene_desc* d, bool* general) {
  const uint32_t N = d->n_nodes, C = d->n_csgs;
  std::vector<uint8_t> seen(N + C, 0);  // 0 not visited, 1 on the current path, 2 done
  std::vector<uint8_t> mesh_seen(d->n_meshes, 0);
  struct Frame {
    uint32_t v, next, csg_depth, group_nest;  // group_nest: groups that are CSG operands on the path (one walk frame each)
    bool in_csg, in_csg_group;
    uint64_t roots;  // worst-case roots of the primitives below (kept inside CSG objects only)
  };
  std::vector<Frame> path;
  auto n_children = [&](uint32_t v) { return v < N ? d->node_child_begin[v + 1] - d->node_child_begin[v] : 2u; };
  auto child = [&](uint32_t v, uint32_t k) { return v < N ? d->child_ref[d->node_child_begin[v] + k] : (k == 0 ? d->csg_left[v - N] : d->csg_right[v - N]); };
  seen[0] = 1;
  path.push_back({0u, 0u, 0u, 0u, false, false, 0ull});
  while (!path.empty()) {
    Frame& f = path.back();
    if (f.next == n_children(f.v)) {
      Frame done = f;
      seen[done.v] = 2;
      path.pop_back();
      // (more than the production kernels' list holds: the general evaluation, whose longer list is checked while it runs)
      if (done.v >= N && done.csg_depth == 0 && done.roots > RT_CSG_LIST_MAX) *general = true;
      if (!path.empty() && path.back().in_csg) path.back().roots += done.roots;
      continue;
    }
    uint32_t ref = child(f.v, f.next++), kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
    if (kind == RT_REF_PRIM) {
      if (f.in_csg) {
        f.roots += max_roots(d, idx);
        if (!is_solid(d, idx)) *general = true;
      }
      continue;
    }
    if (kind == RT_REF_MESH) {
      if (f.in_csg) return fail(RT_ERR_UNSUPPORTED, "a device-resident mesh inside a CSG object is not supported");
      if (mesh_seen[idx]++) return fail(RT_ERR_INVALID_ARGUMENT, "a mesh is referenced more than once");
      continue;
    }
    uint32_t u = kind == RT_REF_NODE ? idx : N + idx;
    if (seen[u] == 1) return fail(RT_ERR_INVALID_ARGUMENT, "the object graph has a cycle (a group or CSG object contains itself)");
    if (seen[u] == 2) return fail(RT_ERR_INVALID_ARGUMENT, "a group or CSG object is referenced more than once");
    Frame g = {u, 0u, f.csg_depth, f.group_nest, f.in_csg, f.in_csg_group, 0ull};
    if (kind == RT_REF_CSG) {
      if (f.in_csg_group) *general = true;  // (csg.rs:113-115 inside group.rs:171-201 inside csg.rs: the general evaluation)
      if (d->csg_depth[idx] != (f.in_csg ? f.csg_depth + 1 : 0u))
        return fail(RT_ERR_INVALID_ARGUMENT, "csg_depth does not match the nesting of the CSG objects");
      g.csg_depth = d->csg_depth[idx];
      g.in_csg = true;
      g.in_csg_group = false;
    } else if (f.in_csg) {
      g.in_csg_group = true;
      if (f.v >= N && ++g.group_nest > RT_CSG_GROUP_NEST)  // (the group is an operand itself: a walk of its own)
        return fail(RT_ERR_UNSUPPORTED, "groups as CSG operands nested deeper than RT_CSG_GROUP_NEST (4) are not supported on the device");
    }
    seen[u] = 1;
    path.push_back(g);  // (f is dangling from here on)
  }
  for (uint32_t k = 0; k < d->n_meshes; k++)
    if (!mesh_seen[k]) return fail(RT_ERR_INVALID_ARGUMENT, "a mesh of rt_scene_desc::mesh is not referenced from the scene's tree");
  return RT_OK;
}

int validate(const rt_scene_desc* d, bool* general) {
  *general = false;
  if (!d) return fail(RT_ERR_INVALID_ARGUMENT, "scene is NULL");
  if (d->abi_version != RT_ABI_VERSION) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_desc.abi_version mismatch");
  if (d->n_nodes == 0 || !d->node_bbox || !d->node_child_begin)
    return fail(RT_ERR_INVALID_ARGUMENT, "scene needs at least the root node (node 0)");
  if (d->n_child_refs && !d->child_ref) return fail(RT_ERR_INVALID_ARGUMENT, "child_ref is NULL");
  if (d->node_child_begin[0] != 0 || d->node_child_begin[d->n_nodes] != d->n_child_refs)
    return fail(RT_ERR_INVALID_ARGUMENT, "node_child_begin must start at 0 and end at n_child_refs");
  for (uint32_t i = 0; i < d->n_nodes; i++)
    if (d->node_child_begin[i] > d->node_child_begin[i + 1]) return fail(RT_ERR_INVALID_ARGUMENT, "node_child_begin not monotonic");
  for (uint32_t i = 0; i < d->n_child_refs; i++)
    if (!ref_ok(d, d->child_ref[i])) return fail(RT_ERR_INVALID_ARGUMENT, "child_ref out of range");
  if (d->n_prims && (!d->prim_shape || !d->prim_material || !d->prim_xform || !d->prim_geom || !d->prim_csg_left))
    return fail(RT_ERR_INVALID_ARGUMENT, "primitive tables missing");
  if (d->n_materials == 0 && (d->n_prims || d->n_meshes)) return fail(RT_ERR_INVALID_ARGUMENT, "no materials");
  for (uint32_t i = 0; i < d->n_prims; i++) {
    uint8_t sh = d->prim_shape[i];
    if (sh > RT_SHAPE_BBOX) return fail(RT_ERR_INVALID_ARGUMENT, "unknown shape tag");
    if (d->prim_material[i] >= d->n_materials) return fail(RT_ERR_INVALID_ARGUMENT, "prim_material out of range");
    if (d->prim_xform[i] != RT_NONE && d->prim_xform[i] >= d->n_xforms) return fail(RT_ERR_INVALID_ARGUMENT, "prim_xform out of range");
    if (sh == RT_SHAPE_TRIANGLE || sh == RT_SHAPE_SMOOTH_TRIANGLE) {
      if (d->prim_geom[i] >= d->n_tris) return fail(RT_ERR_INVALID_ARGUMENT, "triangle index out of range");
    } else if (sh == RT_SHAPE_CYLINDER || sh == RT_SHAPE_CONE) {
      if (d->prim_geom[i] >= d->n_quadrics) return fail(RT_ERR_INVALID_ARGUMENT, "quadric index out of range");
    }
  }
  if (d->n_tris && (!d->tri_p1 || !d->tri_e1 || !d->tri_e2 || !d->tri_n1 || !d->tri_n2 || !d->tri_n3))
    return fail(RT_ERR_INVALID_ARGUMENT, "triangle tables missing");
  if (d->n_xforms && !d->xform_inv) return fail(RT_ERR_INVALID_ARGUMENT, "xform_inv is NULL");
  if (d->n_quadrics && (!d->quadric_y_min || !d->quadric_y_max || !d->quadric_closed))
    return fail(RT_ERR_INVALID_ARGUMENT, "quadric tables missing");
  if (d->n_materials && (!d->material || !d->pattern_kind || !d->pattern_c1 || !d->pattern_c2 || !d->pattern_inv))
    return fail(RT_ERR_INVALID_ARGUMENT, "material tables missing");
  for (uint32_t i = 0; i < d->n_materials; i++)
    if (d->pattern_kind[i] > RT_PATTERN_TEST) return fail(RT_ERR_INVALID_ARGUMENT, "unknown pattern tag");
  if (d->n_lights && (!d->light_pos || !d->light_intensity)) return fail(RT_ERR_INVALID_ARGUMENT, "light tables missing");
  if (d->n_csgs && (!d->csg_op || !d->csg_left || !d->csg_right || !d->csg_depth))
    return fail(RT_ERR_INVALID_ARGUMENT, "csg tables missing");
  if (d->n_meshes) {
    if (!d->mesh || !d->mesh_bbox || !d->mesh_built || !d->mesh_material_begin || !d->mesh_material) return fail(RT_ERR_INVALID_ARGUMENT, "mesh tables missing");
    int device = 0;
    (void)hipGetDevice(&device);
    if (d->mesh_material_begin[0] != 0) return fail(RT_ERR_INVALID_ARGUMENT, "mesh_material_begin must start at 0");
    uint64_t total = 0;
    for (uint32_t k = 0; k < d->n_meshes; k++) {
      const rt_mesh* m = d->mesh[k];
      if (!m) return fail(RT_ERR_INVALID_ARGUMENT, "a mesh is NULL");
      if (!m->posed) return fail(RT_ERR_INVALID_ARGUMENT, "a mesh has not been posed (rt_mesh_pose) for this scene");
      if (m->device != device) return fail(RT_ERR_INVALID_ARGUMENT, "a mesh lives on another device than the current one");
      if (d->mesh_material_begin[k + 1] < d->mesh_material_begin[k] || d->mesh_material_begin[k + 1] - d->mesh_material_begin[k] != m->n_palette)
        return fail(RT_ERR_INVALID_ARGUMENT, "mesh_material must hold one scene material per palette entry of each mesh");
      for (uint32_t q = d->mesh_material_begin[k]; q < d->mesh_material_begin[k + 1]; q++)
        if (d->mesh_material[q] >= d->n_materials) return fail(RT_ERR_INVALID_ARGUMENT, "mesh_material out of range");
      total += m->n;
    }
    if (total + d->n_prims > RT_REF_INDEX_MASK) return fail(RT_ERR_UNSUPPORTED, "too many primitives");
  }
  // A cone's cap test is x^2 + z^2 <= |y| (cone.rs:106-112: the radius, not its square), so a cap at |y| < 1 is a disk of
  // radius sqrt|y| > |y|, while the cone's bounding box takes max(|y_min|, |y_max|) for the radius (cone.rs:96-104): such a cap
  // sticks out of the cone's box, and out of every group's box above it.  As with the CSG operands of is_solid: the reference
  // finds those intersections whenever the ray's LINE crosses the group's box and its visiting order lets it, which the
  // production traversal cannot reproduce (found by tools/fuzz_scenes.py: 7 of 1500 random scenes differed in 1-3 pixels,
  // every one of them with such a cone).  The scene takes the reference-order kernels.  (The reference's own cones — min -1
  // max 0, min 0 max 1 — are not of this kind.)
  for (uint32_t i = 0; i < d->n_prims; i++) {
    if (d->prim_shape[i] != RT_SHAPE_CONE || !d->quadric_closed[d->prim_geom[i]]) continue;
    const double a = std::fabs(d->quadric_y_min[d->prim_geom[i]]), b = std::fabs(d->quadric_y_max[d->prim_geom[i]]), r = std::max(a, b);
    if (std::sqrt(a) > r || std::sqrt(b) > r) *general = true;
  }
  for (uint32_t i = 0; i < d->n_csgs; i++) {
    if ((d->csg_left[i] >> RT_REF_KIND_SHIFT) == RT_REF_MESH || (d->csg_right[i] >> RT_REF_KIND_SHIFT) == RT_REF_MESH)
      return fail(RT_ERR_UNSUPPORTED, "a device-resident mesh as a CSG operand is not supported");
    if (d->csg_op[i] > RT_CSG_DIFFERENCE) return fail(RT_ERR_INVALID_ARGUMENT, "unknown csg op");
    if (!ref_ok(d, d->csg_left[i]) || !ref_ok(d, d->csg_right[i])) return fail(RT_ERR_INVALID_ARGUMENT, "csg child out of range");
    if (d->csg_depth[i] >= RT_CSG_GEN_NEST_MAX)
      return fail(RT_ERR_UNSUPPORTED, "CSG objects nested deeper than RT_CSG_GEN_NEST_MAX (12) are not supported on the device");
    if (d->csg_depth[i] >= RT_CSG_NEST_MAX) *general = true;
  }
  return check_graph(d, general);
}

// ---------------------------------------------------------------------------------------------
// Device-side scene preparation (rt_mesh.h): the host side of rt_mesh_* and of the build rt_scene_create launches
// ---------------------------------------------------------------------------------------------
std::string mesh_status_text(uint32_t status) {
  std::string t;
  if (status & mesh::MESH_STATUS_NOT_FINITE) t += "a posed triangle's box has a coordinate that is not finite or beyond 1e18; ";
  if (status & mesh::MESH_STATUS_UNORDERED) t += "the candidate planes of a group are not ordered (NaN coordinates); ";
  if (status & mesh::MESH_STATUS_TOO_DEEP) t += "the tree is deeper than RT_MESH_MAX_LEVELS (64) levels; ";
  return t + "prepare this frame on the host instead";
}

// ObjectGroup::build_bvh (group.rs:144-167) for the posed mesh, then its part of the scene's tables — in two halves, so that the
// builds of a scene's meshes run side by side on their streams (rt_scene_create: begin for every mesh, then finish for every
// mesh; dragons.yml's six: rt_scene_create 4.85 -> 2.4 ms).  finish returns when the tables are complete.  levels_out: levels of the tree (a mesh
// that keeps its flat child list: 1); max_leaf_out: the most triangles a group that does not split holds.
struct MeshBuild {
  uint32_t cur = 0, level = 0, target = 0;
  bool built = false;
};
// Levels [b.level, b.target), the tables (written optimistically: when the tree turns out to be unfinished they are simply
// written again) and the control block back to the host.
int mesh_build_enqueue(rt_mesh* m, const mesh::MeshOut& out, MeshBuild& b) {
  const mesh::MeshDev& M = m->dev;
  hipStream_t st = m->stream;
  const uint32_t grid = (m->n + RT_MESH_BLOCK - 1u) / RT_MESH_BLOCK;
  for (; b.level < b.target; b.level++) {
    hipLaunchKernelGGL(mesh::bin_kernel, dim3(grid), dim3(RT_MESH_BLOCK), 0, st, M, b.cur, b.level);
    hipLaunchKernelGGL(mesh::choose_kernel, dim3((m->n + RT_MESH_CHOOSE_POS - 1u) / RT_MESH_CHOOSE_POS), dim3(RT_MESH_BLOCK), 0, st, M, b.cur, b.level);
    hipLaunchKernelGGL(mesh::count_kernel, dim3(grid), dim3(RT_MESH_BLOCK), 0, st, M, b.cur, b.level);
    hipLaunchKernelGGL(mesh::scatter_kernel, dim3(grid), dim3(RT_MESH_BLOCK), 0, st, M, b.cur, b.level);
    b.cur ^= 1u;
  }
  hipLaunchKernelGGL(mesh::fold_kernel, dim3(grid), dim3(RT_MESH_BLOCK), 0, st, M, out);
  hipLaunchKernelGGL(mesh::finalize_kernel, dim3(grid), dim3(RT_MESH_BLOCK), 0, st, M, out, b.cur);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(m->h_ctrl, M.ctrl, sizeof(mesh::Ctrl), hipMemcpyDeviceToHost, st));
  return RT_OK;
}
int mesh_build_begin(rt_mesh* m, const mesh::MeshOut& out, bool built, MeshBuild& b) {
  const mesh::MeshDev& M = m->dev;
  hipStream_t st = m->stream;
  const uint32_t grid = (m->n + RT_MESH_BLOCK - 1u) / RT_MESH_BLOCK;
  HIP_TRY(hipMemsetAsync(M.dec, 0xFF, (size_t)m->n * sizeof(mesh::Dec), st));
  HIP_TRY(hipMemsetAsync(M.bins, 0, (size_t)m->n * sizeof(mesh::Bin), st));
  if (!m->acc_clean) {
    const size_t items = (size_t)m->n * RT_MESH_ACC_PER_SEG;
    hipLaunchKernelGGL(mesh::acc_reset_kernel, dim3((unsigned)((items + RT_MESH_BLOCK - 1u) / RT_MESH_BLOCK)), dim3(RT_MESH_BLOCK), 0, st, M);
  }
  m->acc_clean = false;
  hipLaunchKernelGGL(mesh::begin_kernel, dim3(grid), dim3(RT_MESH_BLOCK), 0, st, M, out, built ? 1u : 0u);
  b = MeshBuild{};
  b.built = built;
  // levels to launch before looking: what the previous frame's tree needed and two more (the first frame: 2 log2 n)
  if (built) {
    if (m->last_levels) b.target = m->last_levels + 2u;
    else
      for (uint32_t v = m->n; v; v >>= 1) b.target += 2u;
    b.target = std::min<uint32_t>(std::max(b.target, 4u), RT_MESH_MAX_LEVELS);
  }
  return mesh_build_enqueue(m, out, b);
}
int mesh_build_finish(rt_mesh* m, const mesh::MeshOut& out, MeshBuild& b, uint32_t* levels_out, uint32_t* max_leaf_out) {
  for (;;) {
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->h_ctrl->status) return fail(RT_ERR_UNSUPPORTED, "device-side scene preparation: " + mesh_status_text(m->h_ctrl->status));
    if (!(b.built && b.level < RT_MESH_MAX_LEVELS && m->h_ctrl->active[b.level] != 0u)) break;
    b.target = std::min<uint32_t>(b.level + 8u, RT_MESH_MAX_LEVELS);  // groups are still open: eight more levels
    int rc = mesh_build_enqueue(m, out, b);
    if (rc) return rc;
  }
  m->acc_clean = true;
  m->cur = b.cur;
  uint32_t max_level = 0;  // the deepest level that held an open group
  for (uint32_t l = 0; l < RT_MESH_MAX_LEVELS; l++)
    if (m->h_ctrl->active[l]) max_level = l;
  m->last_levels = max_level + 2u;  // (levels 0..max_level split or stopped; their children are one level further)
  *levels_out = m->last_levels;
  *max_leaf_out = m->h_ctrl->max_leaf;
  return RT_OK;
}

DevOptions dev_options(const rt_options* o) {
  DevOptions d;
  d.n_offsets = o->n_offsets;
  for (int i = 0; i < 5; i++) d.offsets[i] = o->offsets[i];
  d.max_depth = o->max_depth;
  d.shadow_intensity = o->shadow_intensity;
  return d;
}

int check_camera_options(const rt_camera* cam, const rt_options* opt);
int check_render_args(const rt_scene* scene, const rt_camera* cam, const rt_options* opt) {
  if (!scene) return fail(RT_ERR_INVALID_ARGUMENT, "scene, camera and options are required");
  return check_camera_options(cam, opt);
}
int check_camera_options(const rt_camera* cam, const rt_options* opt) {
  if (!cam || !opt) return fail(RT_ERR_INVALID_ARGUMENT, "scene, camera and options are required");
  if (cam->width == 0 || cam->height == 0) return fail(RT_ERR_INVALID_ARGUMENT, "camera width/height must be > 0");
  if (cam->width > (1u << 19) || cam->height > (1u << 19)) return fail(RT_ERR_UNSUPPORTED, "camera width/height above 524288");
  if (opt->n_offsets < 1 || opt->n_offsets > 5) return fail(RT_ERR_INVALID_ARGUMENT, "n_offsets must be 1..5");
  if (opt->max_depth > RT_MAX_DEPTH)
    return fail(RT_ERR_UNSUPPORTED, "max_depth exceeds the device limit RT_MAX_DEPTH (16)");
  if (opt->pipeline > RT_PIPELINE_STREAM_ALWAYS) return fail(RT_ERR_INVALID_ARGUMENT, "unknown rt_options.pipeline");
  return RT_OK;
}

uint32_t tiles_total(uint32_t w, uint32_t h, uint32_t* tiles_x) {
  uint32_t tx = (w + RT_TILE_DIM - 1) / RT_TILE_DIM, ty = (h + RT_TILE_DIM - 1) / RT_TILE_DIM;
  if (tiles_x) *tiles_x = tx;
  return tx * ty;
}

// ---------------------------------------------------------------------------------------------
// The streamed pipeline (rt_stream.h): workspace and launch sequence
// ---------------------------------------------------------------------------------------------
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
#define RT_STREAM_RATIO_MAX 64u

// Capacities of a batch of n samples laid out for `ratio` node records per sample: rays per level, nodes.
void batch_caps(uint64_t n, uint32_t ratio, uint64_t* cap_rays, uint64_t* cap_nodes) {
  *cap_nodes = n * ratio;
  *cap_rays = n * std::max(1u, ratio / 2);  // a level's rays are at most twice the nodes of the level above
}

// Carves the workspace for batches of n samples; returns the bytes needed (and sets the pointers when base != NULL).
size_t carve_workspace(stream::Ws* w, char* base, uint64_t n, uint32_t ratio, uint32_t n_lights) {
  const size_t A = 256;
  uint64_t cap_rays, cap_nodes;
  batch_caps(n, ratio, &cap_rays, &cap_nodes);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t at = off;
    off = align_up(off + bytes, A);
    return at;
  };
  size_t o_ctrl = take(sizeof(stream::Ctrl));
  size_t o_rays0 = take((size_t)cap_rays * RT_RAY_STRIDE * 8);
  size_t o_rays1 = take((size_t)cap_rays * RT_RAY_STRIDE * 8);
  size_t o_hits = take((size_t)cap_rays * RT_HIT_FIELDS * 8);
  size_t o_hp0 = take((size_t)n * 4);
  size_t o_hp1 = take((size_t)cap_rays * 4);
  size_t o_retrace = take((size_t)cap_rays * 4);
  size_t o_nf = take((size_t)cap_nodes * stream::NF_COUNT * 8);
  size_t o_nu = take((size_t)cap_nodes * stream::NU_COUNT * 4);
  size_t o_shadow = take((size_t)cap_nodes * std::max<uint32_t>(n_lights, 1u) * 8);
  size_t o_lit = take((size_t)cap_nodes);
  size_t o_rgb = take((size_t)n * 24);
  size_t o_flag = take((size_t)n);
  size_t o_tiles = take((size_t)n / 64 * 4 + 4);
  if (base) {
    w->ctrl = (stream::Ctrl*)(base + o_ctrl);
    w->rays[0] = (double*)(base + o_rays0);
    w->rays[1] = (double*)(base + o_rays1);
    w->hit_f = (double*)(base + o_hits);
    w->hit_prim[0] = (uint32_t*)(base + o_hp0);
    w->hit_prim[1] = (uint32_t*)(base + o_hp1);
    w->retrace = (uint32_t*)(base + o_retrace);
    w->node_f = (double*)(base + o_nf);
    w->node_u = (uint32_t*)(base + o_nu);
    w->shadow = (double*)(base + o_shadow);
    w->node_lit = (uint8_t*)(base + o_lit);
    w->sample_rgb = (double*)(base + o_rgb);
    w->sample_flag = (uint8_t*)(base + o_flag);
    w->tile_xy = (uint32_t*)(base + o_tiles);
    w->cap_rays = (uint32_t)cap_rays;
    w->cap_nodes = (uint32_t)cap_nodes;
  }
  return off;
}

// Takes in the fallback count of the previous frame, if it has arrived: batches that overflowed get twice the node records.
void absorb_summary(const rt_scene* sc) {
  if (!sc->summary_pending || hipEventQuery(sc->summary_event) != hipSuccess) return;
  sc->summary_pending = false;
  sc->layout_learned = true;
  if (sc->h_summary[0] > 0) {
    sc->ratio_floor = std::min<uint32_t>(sc->nodes_per_sample + 1u, RT_STREAM_RATIO_MAX);  // this ratio was too small once
    if (sc->nodes_per_sample < RT_STREAM_RATIO_MAX) sc->nodes_per_sample *= 2;
  } else if (sc->h_summary[1] > 0) {
    // nothing overflowed: lay the next frames out for what this one used plus a quarter (never more than now, never a
    // ratio that has overflowed before).  Fewer records per sample = more samples per batch, and every batch has fixed
    // costs (a 4K frame at 16 spp uses 0.35 node records per sample: one batch instead of two, 46.8 -> 43.1 ms).
    // The ray queues hold max(1, ratio / 2) rays per sample (batch_caps).
    uint32_t want = std::max<uint32_t>(1u, (sc->h_summary[1] * 5u / 4u + 15u) / 16u);
    uint32_t rays = (sc->h_summary[2] * 5u / 4u + 15u) / 16u;
    if (rays > 1u) want = std::max(want, 2u * rays);
    want = std::max(want, sc->ratio_floor);
    if (want < sc->nodes_per_sample) sc->nodes_per_sample = want;
  }
}

// Lays the scene's workspace out for batches of up to min(want, limit) samples at the scene's current nodes-per-sample
// ratio, (re)allocating when it is too small.  limit: RT_STREAM_BATCH_SAMPLES (default 256 Mi samples: every launch of a
// persistent kernel has a fixed cost, 2 batches per 4K frame measured 5 % faster than 4), and the workspace
// never takes more than a third of the device's free memory.
int ensure_workspace(const rt_scene* sc, uint64_t want_samples, uint32_t* batch_samples) {
  if (!sc->h_summary) {  // (d_summary is part of the scene's arena, zeroed)
    sc->h_summary = g_pinned_slots.take();
    if (!sc->h_summary || hipEventCreate(&sc->summary_event) != hipSuccess)
      return fail(RT_ERR_DEVICE, "allocation of the frame summary failed");
    sc->h_summary[0] = sc->h_summary[1] = sc->h_summary[2] = 0;
  }
  absorb_summary(sc);
  const uint32_t ratio = sc->nodes_per_sample;
  uint64_t limit = 256ull << 20;
  if (const char* e = std::getenv("RT_STREAM_BATCH_SAMPLES")) {
    unsigned long long v = std::strtoull(e, nullptr, 10);
    if (v >= 64) limit = v;
  }
  // node ids carry two flag bits, and (node, light) items are counted in 32 bits
  limit = std::min<uint64_t>(limit, std::min<uint64_t>((1ull << 30) / ratio, ((1ull << 32) - 1) / ((uint64_t)ratio * std::max<uint32_t>(sc->n_lights, 1u))));
  limit = std::max<uint64_t>(limit & ~63ull, 64);  // (rounding n UP to 64 below must not carry it past the limit)
  uint64_t n = std::max<uint64_t>(align_up(std::min<uint64_t>(want_samples, limit), 64), 64);
  size_t need = carve_workspace(nullptr, nullptr, n, ratio, sc->n_lights);
  if (need > sc->ws_bytes) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) {
      // Blocks that destroyed scenes left in the pool can be had back (DevicePool::take reuses one that fits, and empties the
      // pool when an allocation fails), so they count as free here.  (Counting them as used and emptying the pool first made
      // every one-shot render of a large frame free and re-allocate its 50-GB workspace once free memory had dropped to
      // three times the need: 1.4 s per call.)
      size_t budget = (free_b + g_device_pool.cached) / 3 + sc->ws_bytes;
      // (a host that shares the device with others caps it: more, smaller batches, the same pixels)
      if (const char* e = std::getenv("RT_WORKSPACE_MAX_BYTES")) {
        const unsigned long long cap = std::strtoull(e, nullptr, 10);
        if (cap >= (64ull << 20)) budget = std::min<size_t>(budget, (size_t)cap);
      }
      while (need > budget && n > 4096) {
        n = align_up(n / 2, 64);
        need = carve_workspace(nullptr, nullptr, n, ratio, sc->n_lights);
      }
      if (need > budget) return fail(RT_ERR_DEVICE, "not enough free device memory for the render workspace");
    }
  }
  if (need > sc->ws_bytes) {
    if (sc->ws_base) {
      if (sc->rendered) (void)hipEventSynchronize(sc->done_event);  // a frame may still be using it
      g_device_pool.give(sc->ws_base, sc->ws_bytes, sc->device);
      sc->ws_base = nullptr;
      sc->ws_bytes = 0;
    }
    size_t cap = 0;
    void* base = g_device_pool.take(need, sc->device, &cap);
    if (!base) return fail(RT_ERR_DEVICE, "hipMalloc of the render workspace (" + std::to_string(need >> 20) + " MiB) failed");
    sc->ws_base = base;
    sc->ws_bytes = cap;
  }
  carve_workspace(&sc->ws, (char*)sc->ws_base, n, ratio, sc->n_lights);
  sc->ws_samples = (uint32_t)n;
  *batch_samples = (uint32_t)n;
  return RT_OK;
}

// After the last batch of a frame: its fallback count travels to the host without stalling anything.
int post_summary(const rt_scene* sc, hipStream_t st) {
  if (sc->summary_pending) return RT_OK;  // the previous one has not been read yet; this frame's count is dropped
  HIP_TRY(hipMemcpyAsync(sc->h_summary, sc->d_summary, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemsetAsync(sc->d_summary, 0, 3 * sizeof(uint32_t), st));
  HIP_TRY(hipEventRecord(sc->summary_event, st));
  sc->summary_pending = true;
  return RT_OK;
}

enum { TC_TRACE_CLOSEST = 0, TC_RETRACE, TC_SHADE, TC_TRACE_SHADOW, TC_COMBINE, TC_OTHER, TC_COUNT };

struct TimedLaunch {  // records an event before and after one launch when the scene is in timing mode
  const rt_scene* sc;
  hipStream_t st;
  bool on;
  TimedLaunch(const rt_scene* scene, hipStream_t stream, int cls)
      : sc(scene), st(stream), on(scene->timing_on && ((scene->timing_classes >> cls) & 1u)) {
    if (!on) return;
    size_t i = sc->timed_class.size();
    while (sc->event_pool.size() < 2 * i + 2) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) {
        on = false;
        return;
      }
      sc->event_pool.push_back(e);
    }
    sc->timed_class.push_back(cls);
    (void)hipEventRecord(sc->event_pool[2 * i], st);
  }
  ~TimedLaunch() {
    if (on) (void)hipEventRecord(sc->event_pool[2 * sc->timed_class.size() - 1], st);
  }
};

// One batch through every level (see rt_stream.h).  CAMERA: level 0 is generated from the camera; otherwise it is the
// batch's given rays (rt_trace_rays).
#ifndef RT_SHADOW_ON_SIDE_STREAM
#define RT_SHADOW_ON_SIDE_STREAM 1  // 0: every kernel of the pipeline on the caller's stream, one after the other
#endif
#ifndef RT_SHADOW_SIDE_SMALL
#define RT_SHADOW_SIDE_SMALL (12u << 20)
#endif
#ifndef RT_SHADOW_SIDE_MID
#define RT_SHADOW_SIDE_MID (14u << 20)
#endif
#ifndef RT_SHADOW_SIDE_BIG
#define RT_SHADOW_SIDE_BIG (64u << 20)
#endif
#ifndef RT_SHADOW_SIDE_LEVELS
#define RT_SHADOW_SIDE_LEVELS 1u  // levels 0 .. this - 1: a shadow launch per level on the side stream; the rest: one launch at the end
#endif
#ifndef RT_CLOSEST_BIG_SAMPLES
#define RT_CLOSEST_BIG_SAMPLES (64u << 20)  // (rt_stream.h, RT_CLOSEST_WAVES_NO_WALK; a 1080p ss 4 batch of 33 M samples is no faster with it)
#endif
#ifndef RT_SHADE_BIG_SAMPLES
#define RT_SHADE_BIG_SAMPLES (64u << 20)  // (rt_stream.h, RT_SHADE_SPAN_BIG)
#endif
#ifndef RT_RETRACE_BLOCKS_PER_CU
#define RT_RETRACE_BLOCKS_PER_CU 32u  // few rays, long dependent walks: 4 -> 16 blocks per CU took 1.85 -> 1.35 ms off the 4K frame (round 1); 16 -> 32 at 4 waves per SIMD 0.97 -> 0.83 ms (round 3)
#endif
template <bool CAMERA, bool STATS>
int run_stream_batch(const rt_scene* sc, const stream::Ws& w, const rt_camera& cam, const DevOptions& dopt, const stream::Batch& b,
                     hipStream_t st, bool allow_side = true) {
  const uint32_t trace_grid = (uint32_t)sc->n_cus * 4u * RT_TRACE_WAVES, closest_grid = (uint32_t)sc->n_cus * 4u * RT_CLOSEST_WAVES;
  const uint32_t wide_grid = (uint32_t)sc->n_cus * 8u, no_walk_grid = (uint32_t)sc->n_cus * 4u * RT_CLOSEST_WAVES_NO_WALK;
  uint32_t need = (b.n_samples + 255u) / 256u;
  const uint32_t first_grid = std::max(1u, std::min(wide_grid, need));
  // The shadow rays of the first levels run on the scene's side stream, level by level, beside the next levels' trace / shade
  // kernels; the shadow rays of the other levels are one launch after the last level.  How many levels (k):
  //  * small batches: k = 1 (round 2: an 800x800 frame 3.50 -> 3.23 ms; k = 2 is better for dragons.yml at 800x800 but worse for
  //    the one-dragon room, the CSG lens and refractions.yml, whose levels are all of one size);
  //  * from RT_SHADOW_SIDE_MID samples (14 Mi, so that an eighth of the 4K frame at 16 samples per pixel, 15.8 Mi, is in): k = 2
  //    (1080p ss 4 9.07 -> 8.89 ms; the eight eighths of the 4K frame, each alone: slowest 5.95 -> 5.55 ms, mean 5.25 -> 5.13 —
  //    level 1's shadow rays then run beside the thin levels 2 and 3, where a part can hold ONE ray that walks for 0.5 ms);
  //  * from RT_SHADOW_SIDE_BIG samples (64 Mi): k = 3 (the 4K frame 28.2 -> 27.4 ms, half of it 15.4 -> 14.9).
  // Until the end of round 3 big batches used no side stream at all (+1 % then: every kernel of the 4K frame fills the GPU by
  // itself): that was the work hand-out's doing — a launch beside another doubled the traffic on the same eight counters,
  // and every returning atomic is a round trip the whole wavefront waits for (rt_stream.h, RT_XCD_MIN_GRAB).
  // RT_SHADOW_SIDE_LEVELS in the environment fixes k for every size (development), RT_SHADOW_SIDE_SAMPLES bounds the batches
  // that use the side stream at all.
  const uint64_t side_below = [] {  // (read per batch: bench.py switches it between steps to time every kernel alone)
    const char* e = std::getenv("RT_SHADOW_SIDE_SAMPLES");
    return e ? std::strtoull(e, nullptr, 10) : ~0ull;
  }();
  const bool use_side = RT_SHADOW_ON_SIDE_STREAM && allow_side && dopt.max_depth > 0 && b.n_samples < side_below;
  const uint32_t side_levels_env = [] {
    const char* e = std::getenv("RT_SHADOW_SIDE_LEVELS");
    return e ? (uint32_t)std::strtoul(e, nullptr, 10) : 0u;
  }();
  // (a small batch of a scene whose frames have reported at most one node record per sample — most camera rays hit nothing, the
  // levels shrink fast: dragons.yml — also takes k = 2: 800x800 1.93 -> 1.72 ms; the one-dragon room, the CSG lens and
  // refractions.yml, whose every camera ray hits something, report 2 and more and lose 1-6 % with k = 2)
  const bool sparse = sc->layout_learned && sc->nodes_per_sample == 1;
  const uint32_t side_levels = side_levels_env ? side_levels_env
                                               : (b.n_samples >= RT_SHADOW_SIDE_BIG ? 3u : ((b.n_samples >= RT_SHADOW_SIDE_MID || sparse) ? 2u : (uint32_t)RT_SHADOW_SIDE_LEVELS));
  hipStream_t side = use_side ? sc->side_stream : st;
  auto launch_shadow = [&](uint32_t first, uint32_t last, hipStream_t on) {
    // A side-stream launch of a small batch leaves wavefront slots to the main stream's kernels: half of them below
    // RT_SHADOW_SIDE_SMALL samples (12 Mi), a quarter up to RT_SHADOW_SIDE_MID.  The persistent shadow wavefronts give up
    // their slots only when the launch's work has run out, and the main stream's next kernels — the thin deeper levels,
    // which cannot fill the GPU by themselves — otherwise wait for that (an eighth of the 4K frame: level 1's retrace launch
    // 0.05 -> 0.66 ms) and then run alone.  800x800: dragons.yml 2.02 -> 1.93 ms, one dragon 2.12 -> 2.08, the CSG lens 7.60 ->
    // 7.53; 1600x1600 3.50 -> 3.44; an eighth of the 4K frame 4.90 -> 4.85.  Bigger batches keep the full grid: what the
    // overlap buys there is the filling of each other's tails, not two kernels side by side (7/8, 6/8 of the slots: 1080p
    // ss 4 8.90 -> 8.94 / 8.98 ms, a quarter of the 4K frame 8.75 -> 8.75 / 8.88; the 4K frame with 7/8, 6/8, 4/8: 27.1 ->
    // 27.8 / 28.4 / 31.8 ms).  RT_SIDE_GRID_EIGHTHS in the environment fixes the fraction for small batches (development).
    static const uint32_t side_eighths_env = [] {
      const char* e = std::getenv("RT_SIDE_GRID_EIGHTHS");
      return e ? std::min<uint32_t>(8u, (uint32_t)std::strtoul(e, nullptr, 10)) : 0u;
    }();
    const uint32_t side_eighths = side_eighths_env ? side_eighths_env : (b.n_samples < RT_SHADOW_SIDE_SMALL ? 4u : 6u);
    const uint32_t grid = (on != st && b.n_samples < RT_SHADOW_SIDE_MID) ? std::max(8u, trace_grid / 64u * side_eighths * 8u) : trace_grid;
    if (STATS || sc->has_csg)
      hipLaunchKernelGGL((stream::trace_shadow_kernel<STATS, true>), dim3(grid), dim3(64), 0, on, sc->dev, w, first, last, sc->d_stats, sc->d_overflow);
    else
      hipLaunchKernelGGL((stream::trace_shadow_kernel<false, false>), dim3(grid), dim3(64), 0, on, sc->dev, w, first, last, sc->d_stats, sc->d_overflow);
  };
  HIP_TRY(hipMemsetAsync(w.ctrl, 0, sizeof(stream::Ctrl), st));
  HIP_TRY(hipMemsetAsync(w.sample_flag, 0, b.n_samples, st));
  for (uint32_t level = 0; level <= dopt.max_depth; level++) {
    {
      TimedLaunch t(sc, st, TC_TRACE_CLOSEST);
      // (the kernels without counters come in two builds: with and without the code for CSG objects)
#define RT_LAUNCH_CLOSEST(ST, CSG, FIRST, WALK)                                                                                                \
  hipLaunchKernelGGL((stream::trace_closest_kernel<CAMERA, ST, CSG, FIRST, WALK>), dim3(closest_grid), dim3(64), 0, st, sc->dev, cam, dopt, w, b, \
                     level, sc->d_stats, sc->d_overflow)
      if (STATS || sc->has_csg) {
        if (level == 0) RT_LAUNCH_CLOSEST(STATS, true, true, true);
        else RT_LAUNCH_CLOSEST(STATS, true, false, true);
      } else if (sc->has_walks) {
        if (level == 0) RT_LAUNCH_CLOSEST(false, false, true, true);
        else RT_LAUNCH_CLOSEST(false, false, false, true);
      } else {
        if (level == 0) RT_LAUNCH_CLOSEST(false, false, true, false);
        else if (b.n_samples >= RT_CLOSEST_BIG_SAMPLES)
          hipLaunchKernelGGL((stream::trace_closest_kernel<CAMERA, false, false, false, false, RT_CLOSEST_WAVES_NO_WALK>), dim3(no_walk_grid), dim3(64), 0, st,
                             sc->dev, cam, dopt, w, b, level, sc->d_stats, sc->d_overflow);
        else RT_LAUNCH_CLOSEST(false, false, false, false);
      }
#undef RT_LAUNCH_CLOSEST
    }
    {
      TimedLaunch t(sc, st, TC_RETRACE);
      hipLaunchKernelGGL((stream::retrace_kernel<CAMERA, STATS>), dim3((uint32_t)sc->n_cus * RT_RETRACE_BLOCKS_PER_CU), dim3(64), 0, st, sc->dev, cam, dopt, w,
                         b, level, sc->d_stats, sc->d_overflow);
    }
    {
      TimedLaunch t(sc, st, TC_SHADE);
      if (b.n_samples >= RT_SHADE_BIG_SAMPLES)
        hipLaunchKernelGGL((stream::shade_kernel<CAMERA, STATS, RT_SHADE_SPAN_BIG>), dim3(level == 0 ? first_grid : wide_grid), dim3(256), 0, st,
                           sc->dev, cam, dopt, w, b, level, sc->d_stats, sc->d_overflow);
      else
        hipLaunchKernelGGL((stream::shade_kernel<CAMERA, STATS, RT_SHADE_SPAN>), dim3(level == 0 ? first_grid : wide_grid), dim3(256), 0, st,
                           sc->dev, cam, dopt, w, b, level, sc->d_stats, sc->d_overflow);
    }
    {
      TimedLaunch t(sc, st, TC_OTHER);
      hipLaunchKernelGGL(stream::close_level_kernel, dim3(1), dim3(64), 0, st, w, level);
    }
    // the shadow rays of the first levels: on the side stream, level by level, beside the next levels' kernels
    if (use_side && level < side_levels) {
      HIP_TRY(hipEventRecord(sc->level_event[level], st));
      HIP_TRY(hipStreamWaitEvent(side, sc->level_event[level], 0));
      TimedLaunch t(sc, side, TC_TRACE_SHADOW);
      launch_shadow(level, level, side);
    }
  }
  const uint32_t rest = use_side ? std::min<uint32_t>(side_levels, dopt.max_depth + 1u) : 0u;
  if (rest <= dopt.max_depth) {  // the shadow rays of the other levels, in one launch
    TimedLaunch t(sc, st, TC_TRACE_SHADOW);
    launch_shadow(rest, dopt.max_depth, st);
  }
  if (use_side) {  // the side stream's shadow kernel is what the fold over the lights waits for
    HIP_TRY(hipEventRecord(sc->side_event, side));
    HIP_TRY(hipStreamWaitEvent(st, sc->side_event, 0));
  }
  for (uint32_t level = dopt.max_depth + 1; level-- > 0;) {
    TimedLaunch t(sc, st, TC_COMBINE);
    hipLaunchKernelGGL((stream::combine_kernel<CAMERA>), dim3(wide_grid), dim3(256), 0, st, sc->dev, cam, dopt, w, b, level);
  }
  {
    TimedLaunch t(sc, st, TC_OTHER);
    hipLaunchKernelGGL((stream::fallback_kernel<CAMERA>), dim3((uint32_t)sc->n_cus * 8u), dim3(64), 0, st, sc->dev, cam, dopt, w, b,
                       STATS ? sc->d_stats : nullptr, sc->d_summary, sc->d_overflow);
  }
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

// rt_options.progress: pixels finished, reported from the stream (a host function behind the batch's last kernel).
struct Progress {
  rt_progress_fn fn = nullptr;
  void* user = nullptr;
  uint64_t left = 0;  // pixels of this render not reported yet
};
struct ProgressCall {
  rt_progress_fn fn;
  void* user;
  uint64_t pixels;
};
std::mutex g_progress_mu;  // n_devices > 1: one stream per device, so several runtime threads; the callback sees one call at a time
void progress_trampoline(void* p) {
  ProgressCall* c = (ProgressCall*)p;
  {
    std::lock_guard<std::mutex> lock(g_progress_mu);
    c->fn(c->pixels, c->user);
  }
  delete c;
}
// Pixels (inside the image) of the tiles that part `part` of `n_parts` owns.
uint64_t part_pixels(const rt_camera* cam, uint32_t part, uint32_t n_parts) {
  uint32_t tiles_x;
  uint32_t n_tiles = tiles_total(cam->width, cam->height, &tiles_x);
  if (n_parts == 1) return (uint64_t)cam->width * cam->height;
  uint64_t n = 0;
  for (uint32_t t = part; t < n_tiles; t += n_parts) {
    uint32_t x0 = (t % tiles_x) * RT_TILE_DIM, y0 = (t / tiles_x) * RT_TILE_DIM;
    n += (uint64_t)std::min(RT_TILE_DIM, cam->width - x0) * std::min(RT_TILE_DIM, cam->height - y0);
  }
  return n;
}
Progress make_progress(const rt_options* opt, const rt_camera* cam, uint32_t part, uint32_t n_parts) {
  Progress p;
  if (opt->progress) {
    p.fn = opt->progress;
    p.user = opt->user;
    p.left = part_pixels(cam, part, n_parts);
  }
  return p;
}
// Reports up to `pixels` more pixels (all that is left when last) once the work queued on `st` so far has finished.
int post_progress(Progress* p, hipStream_t st, uint64_t pixels, bool last) {
  if (!p || !p->fn) return RT_OK;
  uint64_t n = last ? p->left : std::min(pixels, p->left);
  if (n == 0) return RT_OK;
  p->left -= n;
  ProgressCall* c = new (std::nothrow) ProgressCall{p->fn, p->user, n};
  if (!c) return fail(RT_ERR_DEVICE, "out of host memory");
  hipError_t e = hipLaunchHostFunc(st, progress_trampoline, c);
  if (e != hipSuccess) {
    delete c;
    return fail(RT_ERR_DEVICE, std::string("hipLaunchHostFunc failed: ") + hipGetErrorString(e));
  }
  return RT_OK;
}

// The frame's part in batches of at most batch_samples samples through workspace w (laid out for that many).
template <bool STATS>
int stream_render_with(const rt_scene* sc, const stream::Ws& w, uint32_t batch_samples, const rt_camera* cam, const DevOptions& dopt,
                       uint32_t part, uint32_t n_parts, uint32_t tiles_x, uint32_t n_owned, int packed, double* d_out, hipStream_t st,
                       Progress* progress = nullptr, bool allow_side = true) {
  const uint32_t n2 = dopt.n_offsets * dopt.n_offsets, per_tile = 64u * n2;
  const uint32_t tiles_per_batch = std::max(1u, batch_samples / per_tile);
  if (tiles_per_batch * per_tile > batch_samples) return fail(RT_ERR_DEVICE, "render workspace smaller than one tile");
  for (uint32_t tile0 = 0; tile0 < n_owned; tile0 += tiles_per_batch) {
    stream::Batch b;
    b.part = part;
    b.n_parts = n_parts;
    b.tiles_x = tiles_x;
    b.n_owned = n_owned;
    b.tile0 = tile0;
    b.n_tiles = std::min(tiles_per_batch, n_owned - tile0);
    b.n2 = n2;
    b.n_samples = b.n_tiles * per_tile;
    b.packed = (uint32_t)packed;
    b.given_o = b.given_d = nullptr;
    b.tile_xy = w.tile_xy;
    b.n2_magic = (1ull << 32) / n2 + 1;
    hipLaunchKernelGGL(stream::tile_table_kernel, dim3(std::max(1u, std::min(1024u, (b.n_tiles + 255u) / 256u))), dim3(256), 0, st, b,
                       w.tile_xy);
    int rc = run_stream_batch<true, STATS>(sc, w, *cam, dopt, b, st, allow_side);
    if (rc) return rc;
    uint32_t grid = std::max(1u, std::min((uint32_t)sc->n_cus * 8u, (b.n_tiles * 64u + 255u) / 256u));
    {
      TimedLaunch t(sc, st, TC_OTHER);
      hipLaunchKernelGGL(stream::resolve_kernel, dim3(grid), dim3(256), 0, st, *cam, w, b, d_out);
    }
    rc = post_progress(progress, st, (uint64_t)b.n_tiles * 64u, tile0 + tiles_per_batch >= n_owned);
    if (rc) return rc;
  }
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

template <bool STATS>
int launch_stream_render(const rt_scene* sc, const rt_camera* cam, const DevOptions& dopt, uint32_t part, uint32_t n_parts,
                         uint32_t tiles_x, uint32_t n_owned, int packed, double* d_out, hipStream_t st, Progress* progress) {
  const uint32_t per_tile = 64u * dopt.n_offsets * dopt.n_offsets;
  uint32_t batch_samples = 0;
  int rc = ensure_workspace(sc, (uint64_t)n_owned * per_tile, &batch_samples);
  if (rc) return rc;
  rc = stream_render_with<STATS>(sc, sc->ws, batch_samples, cam, dopt, part, n_parts, tiles_x, n_owned, packed, d_out, st, progress);
  if (rc) return rc;
  return post_summary(sc, st);
}

int launch_render_once(const rt_scene* scene, const rt_camera* cam, const rt_options* opt, uint32_t part, uint32_t n_parts, int packed,
                       double* d_out, hipStream_t stream, rt_stats* stats) {
  int rc = check_render_args(scene, cam, opt);
  if (rc) return rc;
  if (!d_out) return fail(RT_ERR_INVALID_ARGUMENT, "output buffer is NULL");
  if (n_parts == 0 || part >= n_parts) return fail(RT_ERR_INVALID_ARGUMENT, "part must be < n_parts");
  uint32_t tiles_x;
  uint32_t n_tiles = tiles_total(cam->width, cam->height, &tiles_x);
  uint32_t n_owned = n_tiles > part ? (n_tiles - part + n_parts - 1) / n_parts : 0;
  HIP_TRY(hipMemsetAsync(scene->d_overflow, 0, sizeof(uint32_t), stream));
  Event e0, e1;
  if (stats) {
    HIP_TRY(hipMemsetAsync(scene->d_stats, 0, RT_N_COUNTERS * sizeof(unsigned long long), stream));
    HIP_TRY(e0.create());
    HIP_TRY(e1.create());
    HIP_TRY(hipEventRecord(e0.e, stream));
  }
  if (n_owned) {
    DevOptions dopt = dev_options(opt);
    Progress progress = make_progress(opt, cam, part, n_parts);
    uint32_t grid = (n_owned + 8 * RT_XCD_GROUP - 1) / (8 * RT_XCD_GROUP) * (8 * RT_XCD_GROUP);
#define RT_LAUNCH_TILES(STATS, REFORDER)                                                                                     \
  hipLaunchKernelGGL((render_tiles_kernel<STATS, REFORDER>), dim3(grid), dim3(64), 0, stream, scene->dev, *cam, dopt, part, \
                     n_parts, tiles_x, n_owned, packed, d_out, scene->d_stats, scene->d_overflow)
    if (scene->general_csg) {
      if (stats)
        hipLaunchKernelGGL((render_tiles_kernel<true, true, true>), dim3(grid), dim3(64), 0, stream, scene->dev, *cam, dopt, part, n_parts,
                           tiles_x, n_owned, packed, d_out, scene->d_stats, scene->d_overflow);
      else
        hipLaunchKernelGGL((render_tiles_kernel<false, true, true>), dim3(grid), dim3(64), 0, stream, scene->dev, *cam, dopt, part, n_parts,
                           tiles_x, n_owned, packed, d_out, scene->d_stats, scene->d_overflow);
    } else if (opt->reference_order) {
      if (stats) RT_LAUNCH_TILES(true, true);
      else RT_LAUNCH_TILES(false, true);
    } else if (opt->pipeline == RT_PIPELINE_SINGLE_KERNEL || (opt->pipeline == RT_PIPELINE_STREAM && scene->one_kernel && !opt->timing)) {
      if (stats) RT_LAUNCH_TILES(true, false);
      else RT_LAUNCH_TILES(false, false);
    } else {
      // rt_options.timing: the production kernels (no counters) with events around every launch
      scene->timing_on = stats && opt->timing;
      scene->timing_classes = (opt->timing & RT_TIMING_ONLY) ? (opt->timing & 0xFFu) : ~0u;
      scene->timed_class.clear();
      rc = (stats && !opt->timing)
               ? launch_stream_render<true>(scene, cam, dopt, part, n_parts, tiles_x, n_owned, packed, d_out, stream, &progress)
               : launch_stream_render<false>(scene, cam, dopt, part, n_parts, tiles_x, n_owned, packed, d_out, stream, &progress);
      scene->timing_on = false;
      if (rc) {
        (void)hipStreamSynchronize(stream);  // whatever was launched is over before the caller may destroy the scene
        (void)hipStreamSynchronize(scene->side_stream);
        return rc;
      }
    }
#undef RT_LAUNCH_TILES
    HIP_TRY(hipGetLastError());
    rc = post_progress(&progress, stream, 0, true);  // (whatever the one-kernel modes have not reported: everything)
    if (rc) return rc;
    scene->rendered = true;
    HIP_TRY(hipEventRecord(scene->done_event, stream));
  }
  if (stats) {
    HIP_TRY(hipEventRecord(e1.e, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    unsigned long long h[RT_N_COUNTERS];
    uint32_t ovf = 0;
    HIP_TRY(hipMemcpy(h, scene->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&ovf, scene->d_overflow, sizeof(ovf), hipMemcpyDeviceToHost));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0.e, e1.e));
    std::memset(stats, 0, sizeof(*stats));
    stats->rays_primary = h[0];
    stats->rays_shadow = h[1];
    stats->rays_reflect = h[2];
    stats->rays_refract = h[3];
    stats->node_visits = h[4];
    stats->tri_tests = h[5];
    stats->prim_tests = h[6];
    stats->hits = h[7];
    stats->hit_bytes = h[8];
    stats->overflow_pixels = h[9];
    stats->wave_steps = h[10];
    stats->lane_steps = h[11];
    stats->rays_retraced = h[12];
    stats->clocks_traversal = h[13] * 64;
    stats->clocks_machine = h[14] * 64;
    stats->fallback_flags = h[15];
    stats->clocks_trace_total = h[16] * 64;
    stats->shadow_node_visits = h[RT_COUNTER_SHADOW];
    stats->shadow_tri_tests = h[RT_COUNTER_SHADOW + 1];
    stats->shadow_prim_tests = h[RT_COUNTER_SHADOW + 2];
    stats->shadow_wave_steps = h[RT_COUNTER_SHADOW + 3];
    stats->shadow_lane_steps = h[RT_COUNTER_SHADOW + 4];
    stats->shadow_clocks_node = h[RT_COUNTER_SHADOW + 5] * 64;
    stats->shadow_clocks_leaf = h[RT_COUNTER_SHADOW + 6] * 64;
    stats->shadow_clocks_total = h[RT_COUNTER_SHADOW + 7] * 64;
    stats->rays_shadow_culled = h[RT_COUNTER_SHADOW + 8];
    double* ms_class[TC_COUNT] = {&stats->ms_trace_closest, &stats->ms_retrace, &stats->ms_shade, &stats->ms_trace_shadow,
                                  &stats->ms_combine,       &stats->ms_other};
    for (size_t i = 0; i < scene->timed_class.size(); i++) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, scene->event_pool[2 * i], scene->event_pool[2 * i + 1]) != hipSuccess) continue;
      int c = scene->timed_class[i];
      *ms_class[c] += t;
      if (c == TC_TRACE_CLOSEST) stats->launches_trace_closest++;
      if (c == TC_TRACE_SHADOW) stats->launches_trace_shadow++;
    }
    scene->timed_class.clear();
    stats->ms_kernel = ms;
    if (ovf) return fail(RT_ERR_OVERFLOW, "a per-ray bounded list (BVH stack or CSG hit list) overflowed on the device");
  }
  return RT_OK;
}

// A statistics launch of the streamed pipeline must count every ray: the work of samples handed to the fallback kernel is
// not counted, so the frame is rendered again with twice the node records per sample until none is.
int launch_render(const rt_scene* scene, const rt_camera* cam, const rt_options* opt, uint32_t part, uint32_t n_parts, int packed,
                  double* d_out, hipStream_t stream, rt_stats* stats) {
  rt_options again = *opt;
  again.progress = nullptr;  // (the pixels are reported once, by the first pass)
  for (int pass = 0;; pass++) {
    int rc = launch_render_once(scene, cam, pass == 0 ? opt : &again, part, n_parts, packed, d_out, stream, stats);
    if (rc || !stats || stats->fallback_flags == 0 || scene->nodes_per_sample >= RT_STREAM_RATIO_MAX) return rc;
    scene->summary_pending = false;  // the stream is idle here; the same count must not double the ratio again
    scene->nodes_per_sample *= 2;
  }
}

// rt_options.n_devices -> the devices a one-shot frame is dealt to (the first one gathers).  RT_DEVICE_LIST (development:
// "0,0,0" runs the multi-device path on one GPU) overrides the list when more than one device is asked for.
std::vector<int> device_list(const rt_options* opt) {
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (!opt || opt->n_devices == 0 || opt->n_devices == 1) return {cur};
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return {cur};
  if (const char* e = std::getenv("RT_DEVICE_LIST")) {  // (ids outside [0, count) are dropped)
    std::vector<int> v;
    for (const char* p = e; *p;) {
      char* end = nullptr;
      long d = std::strtol(p, &end, 10);
      if (end == p) break;
      if (d >= 0 && d < count) v.push_back((int)d);
      p = *end == ',' ? end + 1 : end;
    }
    if (!v.empty()) return v;
  }
  int n = opt->n_devices < 0 ? count : std::min(opt->n_devices, count);
  std::vector<int> v;
  for (int d = 0; d < n; d++) v.push_back(d);
  return v;
}

// The image's way back to the caller's (pageable, usually freshly allocated) buffer.  One hipMemcpy does it at ~9 GB/s, and
// several of them on slices from several threads are no faster (measured).  Instead: one copy into a pinned staging block the
// process keeps (PCIe speed), then a few threads move its slices into the caller's pages, whose first-touch faults they
// take side by side.
#ifndef RT_DOWNLOAD_THREADS
#define RT_DOWNLOAD_THREADS 6
#endif
struct PinnedStage {
  std::mutex m;
  void* p = nullptr;
  size_t cap = 0;
  void release() {
    std::lock_guard<std::mutex> g(m);
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
} g_download_stage;
hipError_t download_image(void* dst, const void* d_src, size_t bytes) {
  const size_t slice_min = 8u << 20;
  const size_t n = std::min<size_t>(RT_DOWNLOAD_THREADS, bytes / slice_min);
  if (n < 2) return hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost);
  std::lock_guard<std::mutex> g(g_download_stage.m);  // (one download at a time through the block)
  if (g_download_stage.cap < bytes) {
    if (g_download_stage.p) (void)hipHostFree(g_download_stage.p);
    g_download_stage.p = nullptr;
    g_download_stage.cap = 0;
    if (hipHostMalloc(&g_download_stage.p, bytes, hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      g_download_stage.p = nullptr;
      return hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost);  // no pinned memory to be had: the plain way
    }
    g_download_stage.cap = bytes;
  }
  hipError_t e = hipMemcpy(g_download_stage.p, d_src, bytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return e;
  const size_t per = (bytes / n + 4095) & ~(size_t)4095;
  std::vector<std::thread> workers;
  JoinAll join{workers};
  for (size_t k = 1; k < n; k++)
    workers.emplace_back([&, k] {
      const size_t at = k * per;
      if (at < bytes) std::memcpy((char*)dst + at, (const char*)g_download_stage.p + at, std::min(per, bytes - at));
    });
  std::memcpy(dst, g_download_stage.p, std::min(per, bytes));
  for (std::thread& t : workers) t.join();
  return hipSuccess;
}

void add_stats(rt_stats* total, const rt_stats& s) {
  // (rt_stats is counters first, then times: sum the former, take the slowest device's of the latter)
  const size_t n_counters = offsetof(rt_stats, ms_upload) / sizeof(uint64_t);
  uint64_t* t = (uint64_t*)total;
  const uint64_t* a = (const uint64_t*)&s;
  for (size_t i = 0; i < n_counters; i++) t[i] += a[i];
  double* td = &total->ms_upload;
  const double* ad = &s.ms_upload;
  const size_t n_times = (sizeof(rt_stats) - offsetof(rt_stats, ms_upload)) / sizeof(double);
  for (size_t i = 0; i < n_times; i++) td[i] = std::max(td[i], ad[i]);
}

// One frame over several devices from ONE host process (rt_options.n_devices; SURVEY.md §8e for a host that is a single
// process, animation_renderer.rs:109-122): a host thread per device creates the scene there, renders the device's cyclic
// share of the 8x8 tiles into a packed buffer and copies it into the gather buffer on the first device; that device
// un-interleaves (assemble_tiles_kernel) and the image is downloaded from it.
int assemble_tiles(uint32_t width, uint32_t height, uint32_t n_parts, uint32_t part_stride_tiles, const void* d_tiles, void* d_out, bool u8,
                   hipStream_t st);
// u8: every device quantises its packed tiles (Color::as_scaled_values, color.rs:55-66) before they travel: 3 bytes per pixel
// instead of 24 through the peer copies, the assembly and the download; out is then width * height * 3 bytes.
int render_on_devices(const std::vector<int>& devs, const rt_scene_desc* desc, const rt_camera* cam, const rt_options* opt, void* out_rgb,
                      bool u8, rt_stats* stats) {
  const uint32_t n = (uint32_t)devs.size();
  int rc = check_camera_options(cam, opt);
  if (rc) return rc;
  const uint32_t n_max = rt_tile_count(cam->width, cam->height, 0, n);
  const size_t px_bytes = u8 ? 1 : sizeof(double);
  const size_t part_values = (size_t)n_max * 64 * 3, render_bytes = part_values * sizeof(double);
  const size_t part_bytes = part_values * px_bytes, image_bytes = (size_t)cam->width * cam->height * 3 * px_bytes;
  struct PoolBlock {
    void* p = nullptr;
    size_t cap = 0;
    int device = 0;
    ~PoolBlock() { g_device_pool.give(p, cap, device); }
  };
  int before = 0;
  (void)hipGetDevice(&before);
  struct Restore {
    int d;
    ~Restore() { (void)hipSetDevice(d); }
  } restore{before};
  HIP_TRY(hipSetDevice(devs[0]));
  PoolBlock gather, image;
  gather.device = image.device = devs[0];
  gather.p = g_device_pool.take(part_bytes * n, devs[0], &gather.cap);
  image.p = g_device_pool.take(image_bytes ? image_bytes : 16, devs[0], &image.cap);
  if (!gather.p || !image.p) return fail(RT_ERR_DEVICE, "hipMalloc of the gather buffers failed");
  std::vector<int> rcs(n, RT_OK);
  std::vector<std::string> errs(n);
  std::vector<rt_stats> part_stats(n);
  std::vector<std::thread> workers;
  JoinAll join{workers};
  for (uint32_t k = 0; k < n; k++)
    workers.emplace_back([&, k]() {
      auto run = [&]() -> int {
        HIP_TRY(hipSetDevice(devs[k]));
        if (devs[k] != devs[0]) {
          hipError_t e = hipDeviceEnablePeerAccess(devs[0], 0);  // (already enabled, or not possible: the copy below works either way)
          (void)e;
          (void)hipGetLastError();
        }
        rt_scene* sc = nullptr;
        int r = rt_scene_create(desc, &sc);
        if (r) return r;
        struct SceneGuard {
          rt_scene* s;
          ~SceneGuard() { rt_scene_destroy(s); }
        } guard{sc};
        PoolBlock local;
        local.device = devs[k];
        local.p = g_device_pool.take((render_bytes ? render_bytes : 16) + (u8 ? part_values : 0), devs[k], &local.cap);
        if (!local.p) return fail(RT_ERR_DEVICE, "hipMalloc of a device's tile buffer failed");
        void* send = u8 ? (void*)((char*)local.p + render_bytes) : local.p;  // (the u8 copy of the tiles sits behind the f64 ones)
        hipStream_t st = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        struct StreamGuard {
          hipStream_t s;
          ~StreamGuard() { (void)hipStreamDestroy(s); }
        } sguard{st};
        std::memset(&part_stats[k], 0, sizeof(rt_stats));
        r = launch_render(sc, cam, opt, k, n, 1, (double*)local.p, st, stats ? &part_stats[k] : nullptr);
        if ((r == RT_OK || r == RT_ERR_OVERFLOW) && u8 && part_values) {
          hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)((part_values + 255) / 256)), dim3(256), 0, st, (const double*)local.p, (uint64_t)part_values,
                             (uint8_t*)send);
          if (hipGetLastError() != hipSuccess) r = fail(RT_ERR_DEVICE, "quantize launch failed");
        }
        if (r == RT_OK || r == RT_ERR_OVERFLOW) {
          char* dst = (char*)gather.p + (size_t)k * part_bytes;
          hipError_t e = devs[k] == devs[0] ? hipMemcpyAsync(dst, send, part_bytes, hipMemcpyDeviceToDevice, st)
                                            : hipMemcpyPeerAsync(dst, devs[0], send, devs[k], part_bytes, st);
          if (e != hipSuccess) r = fail(RT_ERR_DEVICE, std::string("copy of a device's tiles to the first device failed: ") + hipGetErrorString(e));
        }
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess && r == RT_OK) r = fail(RT_ERR_DEVICE, std::string("render on a device failed: ") + hipGetErrorString(e));
        absorb_summary(sc);
        return r;
      };
      try {
        rcs[k] = run();
      } catch (const std::exception& e) {  // (nothing may unwind out of a thread either)
        rcs[k] = fail_host(e.what());
      }
      if (rcs[k]) errs[k] = g_err;  // (g_err is per thread)
    });
  for (std::thread& t : workers) t.join();
  for (uint32_t k = 0; k < n; k++)
    if (rcs[k] && rcs[k] != RT_ERR_OVERFLOW) return fail(rcs[k], "device " + std::to_string(devs[k]) + ": " + errs[k]);
  HIP_TRY(hipSetDevice(devs[0]));
  rc = assemble_tiles(cam->width, cam->height, n, n_max, gather.p, image.p, u8, nullptr);
  if (rc) return rc;
  if (download_image(out_rgb, image.p, image_bytes) != hipSuccess) return fail(RT_ERR_DEVICE, "download of the image failed");
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    for (uint32_t k = 0; k < n; k++) add_stats(stats, part_stats[k]);
  }
  for (uint32_t k = 0; k < n; k++)
    if (rcs[k] == RT_ERR_OVERFLOW) return fail(RT_ERR_OVERFLOW, "device " + std::to_string(devs[k]) + ": " + errs[k]);
  return RT_OK;
}

int assemble_tiles(uint32_t width, uint32_t height, uint32_t n_parts, uint32_t part_stride_tiles, const void* d_tiles, void* d_out, bool u8,
                   hipStream_t st) {
  if (!d_tiles || !d_out || n_parts == 0 || width == 0 || height == 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments to rt_assemble_tiles_device");
  uint32_t tiles_x;
  uint32_t n_tiles = tiles_total(width, height, &tiles_x);
  size_t threads = (size_t)n_tiles * 64;
  if (u8)
    hipLaunchKernelGGL(assemble_tiles_kernel<uint8_t>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, width, height, n_parts,
                       part_stride_tiles, tiles_x, n_tiles, (const uint8_t*)d_tiles, (uint8_t*)d_out);
  else
    hipLaunchKernelGGL(assemble_tiles_kernel<double>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, width, height, n_parts,
                       part_stride_tiles, tiles_x, n_tiles, (const double*)d_tiles, (double*)d_out);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

int render_one_shot(const rt_scene_desc* scene_desc, const rt_camera* camera, const rt_options* options, void* out, bool u8, rt_stats* stats);

}  // namespace

extern "C" {

uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }
const char* rt_last_error(void) { return g_err.c_str(); }

int rt_device_count(void) try {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_err = std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e);
    return RT_ERR_DEVICE;
  }
  return n;
} RT_NO_UNWIND

int rt_scene_create(const rt_scene_desc* d, rt_scene** out) try {
  auto t_create = std::chrono::steady_clock::now();
  const bool debug_timing = std::getenv("RT_DEBUG_TIMING") != nullptr;
  auto lap = [&](const char* what) {  // RT_DEBUG_TIMING: where the host side of a scene's creation spends its time
    if (debug_timing)
      fprintf(stderr, "[rtcore]   %-10s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count());
  };
  if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  bool general_csg = false;
  int rc = validate(d, &general_csg);
  if (rc) return rc;
  lap("validated");

  // The production tree is built on a second thread while this one lays out the reference-order records (both only read
  // the description; a third of a large scene's creation time each).  Small scenes are not worth a thread.
  FastLayout L;
  bool layout_ok = true, layout_threw = false;  // (declared before the thread that writes them: destroyed after it is joined)
  std::thread layout_thread;
  const bool two_threads = d->n_prims >= 4096;
  struct JoinOnExit {  // (every return below waits for it)
    std::thread& t;
    ~JoinOnExit() {
      if (t.joinable()) t.join();
    }
  } join_on_exit{layout_thread};
  if (two_threads)
    layout_thread = std::thread([&] {
      try {
        layout_ok = build_fast_layout(d, L);
      } catch (...) {  // (nothing may unwind out of a thread: reported below as RT_ERR_HOST)
        layout_threw = true;
      }
    });

  std::vector<int64_t> memo(d->n_nodes, -1);
  uint32_t need = stack_need(d, 0, memo, 0);
  if (need < 1) need = 1;
  if (need > RT_STACK_MAX)
    return fail(RT_ERR_UNSUPPORTED, "the BVH needs a traversal stack of " + std::to_string(need) + " entries; the device limit is " +
                                        std::to_string(RT_STACK_MAX));

  std::unique_ptr<rt_scene> owner(new rt_scene());  // (destroyed on every path but the last)
  rt_scene* sc = owner.get();
  t_stage.size = 0;
  sc->stage = &t_stage;
  (void)hipGetDevice(&sc->device);
  uint64_t mesh_tris_total = 0;
  for (uint32_t k = 0; k < d->n_meshes; k++) mesh_tris_total += d->mesh[k]->n;
  {  // an upper bound of the arena image: the big tables are filled in place, so the buffer must not move (stage_inplace)
    const size_t P = d->n_prims, R = d->n_child_refs, T = (size_t)mesh_tris_total;
    size_t cap = (R + d->n_nodes + 16) * sizeof(DevNode) + R * (4 + sizeof(DevLeafRec)) + ((size_t)d->n_csgs + 1) * sizeof(DevCsg2Rec) +
                 (R + 2 + 2 * (size_t)d->n_csgs) * sizeof(DevEntry) + P * (4 + 16 + 4 + 1) + (size_t)d->n_xforms * 128 +
                 P * (RT_TRI_STRIDE + 9) * sizeof(double) + (size_t)d->n_quadrics * 17 + (size_t)d->n_csgs * 10 +
                 (size_t)d->n_materials * (64 + 1 + 24 + 24 + 128 + 8) + (size_t)d->n_lights * 48 +
                 T * (2 * sizeof(DevNode) + (RT_TRI_STRIDE + 9) * sizeof(double) + 4 + 16 + 4 + 1 + 3 * sizeof(DevEntry)) + 64 * 1024;
    if (!t_stage.reserve(cap)) return fail_host("staging the scene tables");
    sc->stage_base = t_stage.p;
  }

  // child records: record 0 = root, then node i's children at 1 + node_child_begin[i]; filled in place (20 MB for dragons.yml)
  size_t n_entries = (size_t)d->n_child_refs + 1;
  for (uint32_t i = 0; i < d->n_csgs; i++)  // + one record per group that is a CSG operand (below)
    n_entries += ((d->csg_left[i] >> RT_REF_KIND_SHIFT) == RT_REF_NODE) + ((d->csg_right[i] >> RT_REF_KIND_SHIFT) == RT_REF_NODE);
  DevEntry* entries = stage_inplace<DevEntry>(sc, n_entries, &sc->dev.entries, 3 * (size_t)mesh_tris_total);
  if (!entries) return fail_host("staging the scene tables");
  size_t entries_size = (size_t)d->n_child_refs + 1;
  auto fill_node_fields = [&](DevEntry& e, uint32_t node) {
    std::memcpy(e.bb, d->node_bbox + (size_t)node * 6, sizeof(double) * 6);
    e.begin = 1 + d->node_child_begin[node];
    e.count = d->node_child_begin[node + 1] - d->node_child_begin[node];
  };
  std::memset(&entries[0], 0, sizeof(DevEntry));
  entries[0].ref = RT_MAKE_REF(RT_REF_NODE, 0);
  fill_node_fields(entries[0], 0);
  std::vector<uint32_t> mesh_root_entry(d->n_meshes, 0u);
  for (uint32_t i = 0; i < d->n_child_refs; i++) {
    DevEntry& e = entries[(size_t)i + 1];
    std::memset(&e, 0, sizeof(e));
    e.ref = d->child_ref[i];
    if ((e.ref >> RT_REF_KIND_SHIFT) == RT_REF_NODE) fill_node_fields(e, e.ref & RT_REF_INDEX_MASK);
    if ((e.ref >> RT_REF_KIND_SHIFT) == RT_REF_MESH) {  // a group for the walk; its child range is written by the device build
      const uint32_t k = e.ref & RT_REF_INDEX_MASK;
      std::memcpy(e.bb, d->mesh_bbox + (size_t)k * 6, sizeof(double) * 6);
      e.ref = RT_MAKE_REF(RT_REF_NODE, 0);
      mesh_root_entry[k] = i + 1;
    }
  }
  // a group that is a CSG operand (csg.rs:113-115) is walked from a record of its own, appended here; the device's CSG
  // tables refer to it by RT_MAKE_REF(RT_REF_NODE, record index)
  std::vector<uint32_t> dev_csg_left(d->csg_left, d->csg_left + d->n_csgs), dev_csg_right(d->csg_right, d->csg_right + d->n_csgs);
  for (uint32_t i = 0; i < d->n_csgs; i++)
    for (std::vector<uint32_t>* side : {&dev_csg_left, &dev_csg_right}) {
      uint32_t ref = (*side)[i];
      if ((ref >> RT_REF_KIND_SHIFT) != RT_REF_NODE) continue;
      uint32_t g = ref & RT_REF_INDEX_MASK;
      const uint32_t operand_need = std::max(1u, stack_need(d, g, memo, 0));
      if (operand_need > RT_CSG_BIG_GROUP_STACK)
        return fail(RT_ERR_UNSUPPORTED, "a group that is a CSG operand nests deeper than the device walks it (RT_CSG_BIG_GROUP_STACK)");
      if (operand_need > RT_CSG_GROUP_STACK) general_csg = true;  // (deeper than the production kernels walk an operand)
      DevEntry e;
      std::memset(&e, 0, sizeof(e));
      e.ref = ref;
      fill_node_fields(e, g);
      (*side)[i] = RT_MAKE_REF(RT_REF_NODE, (uint32_t)entries_size);
      entries[entries_size++] = e;
    }
  assign_ranks(d, L.prim_rank, L.mesh_rank0);  // (the layout thread does not touch these two)
  lap("entries");
  if (two_threads) layout_thread.join();
  else layout_ok = build_fast_layout(d, L);
  if (layout_threw) return fail_host("laying out the scene's tree");
  if (!layout_ok) return fail(RT_ERR_UNSUPPORTED, L.error);
  lap("tree");
  if (L.stack_need > RT_FAST_STACK_MAX)
    return fail(RT_ERR_UNSUPPORTED, "the BVH needs a device traversal stack of " + std::to_string(L.stack_need) + " entries; it holds " +
                                        std::to_string(RT_FAST_STACK_MAX));
  // The meshes' ranges in the tables (filled by the device, rt_mesh.h): behind the scene's own parts.
  struct MeshRange {
    uint32_t node0, tri0, prim0, entry_groups0, entry_prims0, mat0;
  };
  std::vector<MeshRange> mesh_range(d->n_meshes);
  uint64_t mesh_nodes_total = 0, mesh_entries_total = 0;
  mesh_tris_total = 0;
  {
    for (uint32_t k = 0; k < d->n_meshes; k++) {
      const uint32_t n = d->mesh[k]->n;
      MeshRange& r = mesh_range[k];
      r.node0 = (uint32_t)(std::max<size_t>(L.nodes.size(), (size_t)RT_LDS_NODES) + mesh_nodes_total);
      r.tri0 = (uint32_t)(L.n_dev_tris + mesh_tris_total);
      r.prim0 = (uint32_t)(d->n_prims + mesh_tris_total);
      r.entry_groups0 = (uint32_t)(entries_size + mesh_entries_total);
      r.entry_prims0 = r.entry_groups0 + 2u * n;
      r.mat0 = d->mesh_material_begin[k];
      mesh_tris_total += n;
      mesh_nodes_total += 2ull * n;
      mesh_entries_total += 3ull * n;
    }
    if (L.nodes.size() + mesh_nodes_total > RT_FREF_INDEX_MASK || L.n_dev_tris + mesh_tris_total > RT_FREF_INDEX_MASK)
      return fail(RT_ERR_UNSUPPORTED, "scene too large for 24-bit node/triangle indices");
    if (entries_size + mesh_entries_total > 0xFFFFFFFFull) return fail(RT_ERR_UNSUPPORTED, "scene too large for 32-bit child records");
    for (DevNode& n : L.nodes)
      for (int c = 0; c < RT_NODE_WIDTH; c++)
        if (RT_IS_MESH_MARKER(n.ref[c])) n.ref[c] = RT_MAKE_FREF(RT_FREF_INNER, 0, mesh_range[n.ref[c] & RT_FREF_INDEX_MASK].node0);
  }
  uint4* prim_info = stage_inplace<uint4>(sc, d->n_prims, &sc->dev.prim_info, (size_t)mesh_tris_total);
  if (!prim_info) return fail_host("staging the scene tables");
  for (uint32_t i = 0; i < d->n_prims; i++) {
    bool tri = d->prim_shape[i] == RT_SHAPE_TRIANGLE || d->prim_shape[i] == RT_SHAPE_SMOOTH_TRIANGLE;
    prim_info[i] = make_uint4(d->prim_shape[i], d->prim_material[i], d->prim_xform[i], tri ? L.prim_tri[i] : d->prim_geom[i]);
  }
  // triangle tables in leaf order: [slot][RT_TRI_STRIDE] p1, e1, e2, prim;  [slot][9] n1, n2, n3
  double* tri_isect = stage_inplace<double>(sc, (size_t)L.n_dev_tris * RT_TRI_STRIDE, &sc->dev.tri_isect, (size_t)mesh_tris_total * RT_TRI_STRIDE);
  double* tri_norm = stage_inplace<double>(sc, (size_t)L.n_dev_tris * 9, &sc->dev.tri_norm, (size_t)mesh_tris_total * 9);
  if (!tri_isect || !tri_norm) return fail_host("staging the scene tables");
  auto fill_tris = [&](uint32_t from, uint32_t to) {
  for (uint32_t i = from; i < to; i++) {
    uint32_t slot = L.prim_tri[i];
    if (slot == RT_NONE) continue;
    size_t src = d->prim_geom[i];
    double* ti = tri_isect + (size_t)slot * RT_TRI_STRIDE;
    double* tn = tri_norm + (size_t)slot * 9;
    for (int k = 0; k < 3; k++) {
      ti[k] = d->tri_p1[src * 3 + k];
      ti[3 + k] = d->tri_e1[src * 3 + k];
      ti[6 + k] = d->tri_e2[src * 3 + k];
      tn[k] = d->tri_n1[src * 3 + k];
      tn[3 + k] = d->tri_n2[src * 3 + k];
      tn[6 + k] = d->tri_n3[src * 3 + k];
    }
    uint64_t bits = i;
    if (d->material[(size_t)d->prim_material[i] * RT_MAT_STRIDE + RT_MAT_TRANSPARENCY] == 0.) bits |= 1ull << RT_TRI_OPAQUE_BIT;
    std::memcpy(ti + 9, &bits, sizeof(bits));
  }
  };
  if (d->n_prims >= 32768) {  // (every primitive writes its own slot: four threads, 5 -> 1.5 ms for 140 000 triangles)
    std::vector<std::thread> fillers;
    JoinAll join_fillers{fillers};
    const uint32_t per = (d->n_prims + 3) / 4;
    for (uint32_t k = 1; k < 4; k++) fillers.emplace_back(fill_tris, std::min(k * per, d->n_prims), std::min((k + 1) * per, d->n_prims));
    fill_tris(0, std::min(per, d->n_prims));
    for (std::thread& t : fillers) t.join();
  } else {
    fill_tris(0, d->n_prims);
  }
  std::vector<double> quadric((size_t)d->n_quadrics * 2);
  for (size_t i = 0; i < d->n_quadrics; i++) {
    quadric[i * 2] = d->quadric_y_min[i];
    quadric[i * 2 + 1] = d->quadric_y_max[i];
  }

  if (std::getenv("RT_DEBUG_TIMING"))
    fprintf(stderr, "[rtcore] layout: %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count());
  sc->noalloc_scratch = scratch_on_device(sc->device);
  sc->stack_need = need;
#define UP(vec, field)                                 \
  do {                                                 \
    rc = upload(sc, vec, &sc->dev.field);              \
    if (rc) return rc;                                 \
  } while (0)
#define UPA(T, ptr, n, field)                          \
  do {                                                 \
    std::vector<T> _v((ptr), (ptr) + (size_t)(n));     \
    UP(_v, field);                                     \
  } while (0)
  if (L.nodes.size() < (size_t)RT_LDS_NODES) L.nodes.resize(RT_LDS_NODES);  // (the trace kernels copy that many to LDS)
#define UPT(vec, field, tail)                          \
  do {                                                 \
    rc = upload(sc, vec, &sc->dev.field, (size_t)(tail)); \
    if (rc) return rc;                                 \
  } while (0)
  UPT(L.nodes, nodes, mesh_nodes_total);
  UP(L.leaf_ref, leaf_ref);
  {
    std::vector<DevLeafRec> recs(L.leaf_ref.size());
    std::vector<DevCsg2Rec> csg2;
    // everything an analytic primitive's test reads; false for triangles (table path, prim_roots)
    auto fill_inline = [&](DevLeafRec& r, uint32_t p) {
      uint8_t sh = d->prim_shape[p];
      if (sh == RT_SHAPE_TRIANGLE || sh == RT_SHAPE_SMOOTH_TRIANGLE) return false;
      r.shape = sh;
      r.material = d->prim_material[p];
      r.flags = RT_LEAFREC_INLINE;
      if (d->material[(size_t)r.material * RT_MAT_STRIDE + RT_MAT_TRANSPARENCY] == 0.) r.flags |= RT_LEAFREC_OPAQUE;
      if (d->prim_xform[p] != RT_NONE) {
        r.flags |= RT_LEAFREC_XFORM;
        std::memcpy(r.inv, d->xform_inv + (size_t)d->prim_xform[p] * 16, sizeof(r.inv));
      }
      if (sh == RT_SHAPE_CYLINDER || sh == RT_SHAPE_CONE) {
        r.y_min = d->quadric_y_min[d->prim_geom[p]];
        r.y_max = d->quadric_y_max[d->prim_geom[p]];
        if (d->quadric_closed[d->prim_geom[p]]) r.flags |= RT_LEAFREC_CLOSED;
      }
      return true;
    };
    for (size_t i = 0; i < recs.size(); i++) {
      DevLeafRec& r = recs[i];
      std::memset(&r, 0, sizeof(r));
      r.ref = L.leaf_ref[i];
      const uint32_t kind = r.ref >> RT_REF_KIND_SHIFT, idx = r.ref & RT_REF_INDEX_MASK;
      if (kind == RT_REF_PRIM) {
        fill_inline(r, idx);
      } else if (kind == RT_REF_CSG) {  // two analytic primitives of at most two roots each: DevCsg2Rec
        const uint32_t lr = d->csg_left[idx], rr = d->csg_right[idx];
        if ((lr >> RT_REF_KIND_SHIFT) != RT_REF_PRIM || (rr >> RT_REF_KIND_SHIFT) != RT_REF_PRIM) continue;
        const uint32_t lp = lr & RT_REF_INDEX_MASK, rp = rr & RT_REF_INDEX_MASK;
        if (max_roots(d, lp) > 2 || max_roots(d, rp) > 2) continue;
        DevCsg2Rec c;
        std::memset(&c, 0, sizeof(c));
        if (!fill_inline(c.left, lp) || !fill_inline(c.right, rp)) continue;
        c.left.ref = lp;
        c.right.ref = rp;
        c.left.flags |= (uint32_t)d->csg_op[idx] << RT_CSG2_OP_SHIFT;
        r.flags = RT_LEAFREC_CSG2;
        r.material = (uint32_t)csg2.size();
        csg2.push_back(c);
      }
    }
    if (csg2.empty()) csg2.emplace_back();  // (never read; keeps the table's pointer valid)
    UP(csg2, csg2_rec);
    UP(recs, leaf_rec);
  }
  UPT(L.prim_rank, prim_rank, mesh_tris_total);
  // (entries, prim_info and the triangle tables are in place already: stage_inplace)
  {
    std::vector<uint32_t> v(d->prim_csg_left, d->prim_csg_left + d->n_prims);
    UPT(v, prim_csg_left, mesh_tris_total);
  }
  {
    // bit 0: a closest hit on this primitive is followed by the container walk (get_refractive, intersection.rs:258-302).
    // The walk only ever COPIES refractive indices out of the material table (or the constant 1.0 of air): when every
    // material's index is exactly 1.0 its result is (1.0, 1.0) whatever the ray meets, and it is not made (dragons.yml: the
    // glass cases around the dragons have index 1; closest hits of the 4K frame 11.5 -> 10.x ms).
    bool walk_matters = false;
    for (uint32_t m = 0; m < d->n_materials; m++)
      if (!(d->material[(size_t)m * RT_MAT_STRIDE + RT_MAT_REFRACTIVE_INDEX] == 1.0)) walk_matters = true;
    std::vector<uint8_t> flags(d->n_prims, 0);
    for (uint32_t i = 0; i < d->n_prims && walk_matters; i++) {
      double tr = d->material[(size_t)d->prim_material[i] * RT_MAT_STRIDE + RT_MAT_TRANSPARENCY];
      if (!(tr == 0. || std::fabs(tr - 0.) < 1.0e-8)) flags[i] = 1;  // !approx_eq(transparency, 0) (approx_eq.rs:16-20)
    }
    sc->has_walks = std::find(flags.begin(), flags.end(), (uint8_t)1) != flags.end();
    // reflected_color_at_depth / refracted_color_at_depth (integrator.rs:154-197) return black for such materials
    sc->one_kernel = L.n_dev_tris + mesh_tris_total == 0;
    for (uint32_t m = 0; m < d->n_materials && sc->one_kernel; m++) {
      const double kr = d->material[(size_t)m * RT_MAT_STRIDE + RT_MAT_REFLECTIVITY], kt = d->material[(size_t)m * RT_MAT_STRIDE + RT_MAT_TRANSPARENCY];
      if (!(std::fabs(kr) < 1.0e-8) || !(std::fabs(kt) < 1.0e-8)) sc->one_kernel = false;
    }
    UPT(flags, prim_flags, mesh_tris_total);
    // the meshes' palettes in terms of this scene's materials: index, "opaque" (rt_device.h RT_TRI_OPAQUE_BIT), "walk follows"
    std::vector<uint32_t> mat_map(d->mesh_material, d->mesh_material + (d->n_meshes ? d->mesh_material_begin[d->n_meshes] : 0u));
    std::vector<uint8_t> mat_flags(mat_map.size(), 0);
    for (size_t q = 0; q < mat_map.size(); q++) {
      const double tr = d->material[(size_t)mat_map[q] * RT_MAT_STRIDE + RT_MAT_TRANSPARENCY];
      if (tr == 0.) mat_flags[q] |= 1u;
      if (walk_matters && !(tr == 0. || std::fabs(tr - 0.) < 1.0e-8)) {
        mat_flags[q] |= 2u;
        sc->has_walks = true;
      }
    }
    stage_bytes(sc, mat_map.data(), mat_map.size() * sizeof(uint32_t), (void*)&sc->mesh_mat_map);
    stage_bytes(sc, mat_flags.data(), mat_flags.size(), (void*)&sc->mesh_mat_flags);
  }
  UPA(double, d->xform_inv, (size_t)d->n_xforms * 16, xform_inv);
  sc->table_bytes[0] = ((size_t)L.n_dev_tris + mesh_tris_total) * RT_TRI_STRIDE * sizeof(double);
  sc->table_bytes[1] = ((size_t)L.n_dev_tris + mesh_tris_total) * 9 * sizeof(double);
  sc->table_bytes[2] = (d->n_prims + mesh_tris_total) * sizeof(uint32_t);
  sc->table_bytes[3] = (d->n_prims + mesh_tris_total) * sizeof(uint4);
  sc->table_bytes[4] = (entries_size + mesh_entries_total) * sizeof(DevEntry);
  sc->table_bytes[5] = (L.nodes.size() + mesh_nodes_total) * sizeof(DevNode);
  sc->table_bytes[6] = d->n_prims + mesh_tris_total;
  UP(quadric, quadric);
  UPA(uint8_t, d->quadric_closed, d->n_quadrics, quadric_closed);
  UPA(uint8_t, d->csg_op, d->n_csgs, csg_op);
  UPA(uint8_t, d->csg_depth, d->n_csgs, csg_depth);
  UP(dev_csg_left, csg_left);
  UP(dev_csg_right, csg_right);
  UPA(double, d->material, (size_t)d->n_materials * RT_MAT_STRIDE, material);
  UPA(uint8_t, d->pattern_kind, d->n_materials, pattern_kind);
  UPA(double, d->pattern_c1, (size_t)d->n_materials * 3, pattern_c1);
  UPA(double, d->pattern_c2, (size_t)d->n_materials * 3, pattern_c2);
  UPA(double, d->pattern_inv, (size_t)d->n_materials * 16, pattern_inv);
  UPA(double, d->light_pos, (size_t)d->n_lights * 3, light_pos);
  UPA(double, d->light_intensity, (size_t)d->n_lights * 3, light_intensity);
#undef UP
#undef UPA
#undef UPT
  sc->dev.n_lights = d->n_lights;
  sc->dev.has_walks = sc->has_walks ? 1u : 0u;
  sc->dev.n_prims = (uint32_t)(d->n_prims + mesh_tris_total);
  sc->n_lights = d->n_lights;
  sc->has_csg = d->n_csgs > 0;
  sc->general_csg = general_csg;
  sc->shape = ((uint64_t)(d->n_prims + mesh_tris_total) << 40) ^ ((uint64_t)d->n_nodes << 20) ^ ((uint64_t)d->n_lights << 8) ^ (uint64_t)d->n_materials;
  (void)g_learned.lookup(sc->device, sc->shape, &sc->nodes_per_sample, &sc->ratio_floor);
  if (const char* e = std::getenv("RT_STREAM_NODES_PER_SAMPLE")) {  // where the adaptive ratio starts (tests start it too low)
    unsigned long v = std::strtoul(e, nullptr, 10);
    if (v >= 1 && v <= RT_STREAM_RATIO_MAX) sc->nodes_per_sample = (uint32_t)v;
  }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, sc->device) == hipSuccess && cus > 0) sc->n_cus = cus;
  }
  stage_bytes(sc, nullptr, RT_N_COUNTERS * sizeof(unsigned long long), (void*)&sc->d_stats);  // zero-filled
  stage_bytes(sc, nullptr, 16, (void*)&sc->d_overflow);
  stage_bytes(sc, nullptr, 16, (void*)&sc->d_summary);
  if (std::getenv("RT_DEBUG_TIMING"))
    fprintf(stderr, "[rtcore] staged: %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count());
  rc = commit_arena(sc);
  if (rc == RT_OK && hipEventCreateWithFlags(&sc->done_event, hipEventDisableTiming) != hipSuccess)
    rc = fail(RT_ERR_DEVICE, "hipEventCreate failed");
  if (rc == RT_OK && (hipStreamCreateWithFlags(&sc->side_stream, hipStreamNonBlocking) != hipSuccess ||
                      hipEventCreateWithFlags(&sc->side_event, hipEventDisableTiming) != hipSuccess))
    rc = fail(RT_ERR_DEVICE, "creation of the scene's side stream failed");
  for (uint32_t l = 0; rc == RT_OK && l <= RT_MAX_DEPTH; l++)
    if (hipEventCreateWithFlags(&sc->level_event[l], hipEventDisableTiming) != hipSuccess) rc = fail(RT_ERR_DEVICE, "hipEventCreate failed");
  if (rc) return rc;
  // The meshes' subtrees and their parts of the tables, on the device (the scene's own parts are in place: commit_arena has
  // waited for the upload).
  if (d->n_meshes) {
    MeshNeeds needs;
    needs.sub.assign(d->n_meshes, 0u);
    uint32_t fast_need = L.stack_need;
    std::vector<MeshBuild> builds(d->n_meshes);
    std::vector<mesh::MeshOut> outs(d->n_meshes);
    // every mesh's build is enqueued on the mesh's own stream ...
    for (uint32_t k = 0; k < d->n_meshes; k++) {
      rt_mesh* m = d->mesh[k];
      mesh::MeshOut& o = outs[k];
      o.nodes = const_cast<DevNode*>(sc->dev.nodes);
      o.tri_isect = const_cast<double*>(sc->dev.tri_isect);
      o.tri_norm = const_cast<double*>(sc->dev.tri_norm);
      o.prim_rank = const_cast<uint32_t*>(sc->dev.prim_rank);
      o.prim_info = const_cast<uint4*>(sc->dev.prim_info);
      o.prim_csg_left = const_cast<uint32_t*>(sc->dev.prim_csg_left);
      o.prim_flags = const_cast<uint8_t*>(sc->dev.prim_flags);
      o.entries = const_cast<DevEntry*>(sc->dev.entries);
      o.mat_map = sc->mesh_mat_map + mesh_range[k].mat0;
      o.mat_flags = sc->mesh_mat_flags + mesh_range[k].mat0;
      o.node0 = mesh_range[k].node0;
      o.tri0 = mesh_range[k].tri0;
      o.prim0 = mesh_range[k].prim0;
      o.rank0 = L.mesh_rank0[k];
      o.entry_groups0 = mesh_range[k].entry_groups0;
      o.entry_prims0 = mesh_range[k].entry_prims0;
      o.root_entry = mesh_root_entry[k];
      o.built = d->mesh_built[k] ? 1u : 0u;
      std::memcpy(o.group_bb, d->mesh_bbox + (size_t)k * 6, sizeof(o.group_bb));
      rc = mesh_build_begin(m, o, d->mesh_built[k] != 0, builds[k]);
      if (rc) {  // (nothing may still be writing the scene's tables when the scene is destroyed)
        for (uint32_t q = 0; q <= k; q++) (void)hipStreamSynchronize(d->mesh[q]->stream);
        return rc;
      }
    }
    // ... and then waited for
    int first_rc = RT_OK;
    std::string first_error;
    for (uint32_t k = 0; k < d->n_meshes; k++) {
      rt_mesh* m = d->mesh[k];
      uint32_t levels = 0, max_leaf = 0;
      rc = mesh_build_finish(m, outs[k], builds[k], &levels, &max_leaf);
      if (rc) {
        if (first_rc == RT_OK) {
          first_rc = rc;
          first_error = rt_last_error();
        }
        (void)hipStreamSynchronize(m->stream);
        continue;
      }
      m->posed = false;  // (the scene has its own copy of everything: the mesh is free for the next frame's pose)
      // Stack entries the mesh's tree can add.  Reference-order walk: a binary tree of `levels` levels holds at most one
      // sibling per level.  Fast path: a 4-wide node per two levels, at most 3 entries held per node; a leaf group of more
      // than one run is a chain of nodes, 3 runs each.
      needs.sub[k] = levels + 1u;
      uint32_t fast = 3u * ((levels + 1u) / 2u + 1u);
      if (max_leaf > RT_FREF_LEAF_MAX) fast += 3u * ((max_leaf + 3u * RT_FREF_LEAF_MAX - 1u) / (3u * RT_FREF_LEAF_MAX));
      fast_need = std::max(fast_need, L.mesh_held[k] + fast);
    }
    if (first_rc) return fail(first_rc, first_error);
    std::fill(memo.begin(), memo.end(), (int64_t)-1);
    const uint32_t need_all = stack_need(d, 0, memo, 0, &needs);
    if (need_all > RT_STACK_MAX)
      return fail(RT_ERR_UNSUPPORTED, "the BVH (with the device-built subtrees) needs a traversal stack of " + std::to_string(need_all) +
                                          " entries; the device limit is " + std::to_string(RT_STACK_MAX));
    if (fast_need > RT_FAST_STACK_MAX)
      return fail(RT_ERR_UNSUPPORTED, "the BVH (with the device-built subtrees) needs a device traversal stack of " + std::to_string(fast_need) +
                                          " entries; it holds " + std::to_string(RT_FAST_STACK_MAX));
  }
  *out = owner.release();
  return RT_OK;
} RT_NO_UNWIND

int rt_mesh_create(const rt_mesh_desc* d, rt_mesh** out) try {
  if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (!d || d->n_tris == 0 || !d->shape || !d->p1 || !d->p2 || !d->p3 || !d->bbox || !d->material || d->n_palette == 0)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_mesh_desc: triangles, boxes and materials are required");
  if (d->n_tris > RT_MESH_MAX_TRIS) return fail(RT_ERR_UNSUPPORTED, "a mesh holds at most " + std::to_string(RT_MESH_MAX_TRIS) + " triangles");
#ifdef RT_HOST_EMU
  return fail(RT_ERR_UNSUPPORTED, "host emulation runs one lane at a time; the mesh kernels cooperate across a block");
#endif
  const uint32_t n = d->n_tris;
  bool any_smooth = false;
  for (uint32_t i = 0; i < n; i++) {
    if (d->shape[i] != RT_SHAPE_TRIANGLE && d->shape[i] != RT_SHAPE_SMOOTH_TRIANGLE) return fail(RT_ERR_INVALID_ARGUMENT, "a mesh holds triangles only");
    if (d->material[i] >= d->n_palette) return fail(RT_ERR_INVALID_ARGUMENT, "rt_mesh_desc.material out of range");
    any_smooth = any_smooth || d->shape[i] == RT_SHAPE_SMOOTH_TRIANGLE;
  }
  if (any_smooth && (!d->n1 || !d->n2 || !d->n3)) return fail(RT_ERR_INVALID_ARGUMENT, "smooth triangles need n1, n2, n3");
  std::unique_ptr<rt_mesh> owner(new rt_mesh());
  rt_mesh* m = owner.get();
  m->n = n;
  m->n_palette = d->n_palette;
  (void)hipGetDevice(&m->device);
  // one block: [base pose | frame pose | build state]; the base pose is staged on the host and copied in one go
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t at = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return at;
  };
  const size_t v3 = (size_t)n * 3 * sizeof(double);
  const size_t o_p1 = take(v3), o_p2 = take(v3), o_p3 = take(v3), o_n1 = take(v3), o_n2 = take(v3), o_n3 = take(v3);
  const size_t o_bbox = take(2 * v3), o_mat = take((size_t)n * 4), o_shape = take(n);
  const size_t base_bytes = off;
  const size_t o_q1 = take(v3), o_e1 = take(v3), o_e2 = take(v3), o_m1 = take(v3), o_m2 = take(v3), o_m3 = take(v3);
  const size_t o_pbb = take(2 * v3), o_ctr = take(v3), o_box = take(6 * sizeof(unsigned long long));
  const size_t o_ord0 = take((size_t)n * 4), o_ord1 = take((size_t)n * 4), o_seg0 = take((size_t)n * 4), o_seg1 = take((size_t)n * 4);
  const size_t o_segs = take((size_t)n * sizeof(mesh::Seg)), o_dec = take((size_t)n * sizeof(mesh::Dec));
  const size_t o_acc = take((size_t)n * RT_MESH_ACC_PER_SEG * sizeof(mesh::Acc)), o_bins = take((size_t)n * sizeof(mesh::Bin));
  const size_t o_bsum = take((size_t)RT_MESH_MAX_BLOCKS * 4), o_tail = take((size_t)n * 4), o_ctrl = take(sizeof(mesh::Ctrl));
  m->block_bytes = off;
  if (hipMalloc(&m->block, m->block_bytes) != hipSuccess) {
    (void)hipGetLastError();
    m->block = nullptr;
    return fail(RT_ERR_DEVICE, "hipMalloc of a mesh (" + std::to_string(m->block_bytes >> 20) + " MiB) failed");
  }
  {
    std::vector<char> stage(base_bytes, 0);
    std::memcpy(stage.data() + o_p1, d->p1, v3);
    std::memcpy(stage.data() + o_p2, d->p2, v3);
    std::memcpy(stage.data() + o_p3, d->p3, v3);
    if (d->n1) std::memcpy(stage.data() + o_n1, d->n1, v3);
    if (d->n2) std::memcpy(stage.data() + o_n2, d->n2, v3);
    if (d->n3) std::memcpy(stage.data() + o_n3, d->n3, v3);
    std::memcpy(stage.data() + o_bbox, d->bbox, 2 * v3);
    std::memcpy(stage.data() + o_mat, d->material, (size_t)n * 4);
    std::memcpy(stage.data() + o_shape, d->shape, n);
    HIP_TRY(hipMemcpy(m->block, stage.data(), base_bytes, hipMemcpyHostToDevice));
  }
  char* b = (char*)m->block;
  mesh::MeshDev& M = m->dev;
  M.n = n;
  M.shape = (const uint8_t*)(b + o_shape);
  M.p1 = (const double*)(b + o_p1);
  M.p2 = (const double*)(b + o_p2);
  M.p3 = (const double*)(b + o_p3);
  M.n1 = (const double*)(b + o_n1);
  M.n2 = (const double*)(b + o_n2);
  M.n3 = (const double*)(b + o_n3);
  M.bbox = (const double*)(b + o_bbox);
  M.mat = (const uint32_t*)(b + o_mat);
  M.q1 = (double*)(b + o_q1);
  M.e1 = (double*)(b + o_e1);
  M.e2 = (double*)(b + o_e2);
  M.m1 = (double*)(b + o_m1);
  M.m2 = (double*)(b + o_m2);
  M.m3 = (double*)(b + o_m3);
  M.pbb = (double*)(b + o_pbb);
  M.ctr = (double*)(b + o_ctr);
  M.pose_box = (unsigned long long*)(b + o_box);
  M.order[0] = (uint32_t*)(b + o_ord0);
  M.order[1] = (uint32_t*)(b + o_ord1);
  M.seg_of[0] = (uint32_t*)(b + o_seg0);
  M.seg_of[1] = (uint32_t*)(b + o_seg1);
  M.segs = (mesh::Seg*)(b + o_segs);
  M.dec = (mesh::Dec*)(b + o_dec);
  M.acc = (mesh::Acc*)(b + o_acc);
  M.bins = (mesh::Bin*)(b + o_bins);
  M.blocksum = (uint32_t*)(b + o_bsum);
  M.segtail = (uint32_t*)(b + o_tail);
  M.ctrl = (mesh::Ctrl*)(b + o_ctrl);
  HIP_TRY(hipMemset(M.ctrl, 0, sizeof(mesh::Ctrl)));
  HIP_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  void* pinned = nullptr;
  HIP_TRY(hipHostMalloc(&pinned, 64 + sizeof(mesh::Ctrl), hipHostMallocPortable));
  m->h_box = (unsigned long long*)pinned;
  m->h_ctrl = (mesh::Ctrl*)((char*)pinned + 64);
  *out = owner.release();
  return RT_OK;
} RT_NO_UNWIND

void rt_mesh_destroy(rt_mesh* mesh) { delete mesh; }

int rt_mesh_pose(rt_mesh* m, uint32_t n_matrices, const double* matrices, double* out_bbox6) try {
  if (!m || (n_matrices && !matrices)) return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments to rt_mesh_pose");
  if (n_matrices > RT_MESH_MAX_MATRICES)
    return fail(RT_ERR_UNSUPPORTED, "rt_mesh_pose applies at most RT_MESH_MAX_MATRICES (8) matrices per frame");
  int device = 0;
  (void)hipGetDevice(&device);
  if (device != m->device) return fail(RT_ERR_INVALID_ARGUMENT, "the mesh lives on another device than the current one");
  mesh::PoseArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n_mats = n_matrices;
  if (n_matrices) std::memcpy(a.m, matrices, (size_t)n_matrices * 16 * sizeof(double));
  const uint32_t grid = (m->n + RT_MESH_BLOCK - 1u) / RT_MESH_BLOCK;
  m->posed = false;
  HIP_TRY(hipMemsetAsync(&m->dev.ctrl->status, 0, sizeof(uint32_t), m->stream));
  hipLaunchKernelGGL(mesh::pose_init_kernel, dim3(1), dim3(RT_MESH_BLOCK), 0, m->stream, m->dev);
  hipLaunchKernelGGL(mesh::pose_kernel, dim3(grid), dim3(RT_MESH_BLOCK), 0, m->stream, m->dev, a);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(m->h_box, m->dev.pose_box, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipMemcpyAsync(m->h_ctrl, m->dev.ctrl, sizeof(mesh::Ctrl), hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  if (m->h_ctrl->status) return fail(RT_ERR_UNSUPPORTED, "rt_mesh_pose: " + mesh_status_text(m->h_ctrl->status));
  for (int k = 0; k < 6; k++) {  // (mesh::dunkey on the host)
    const unsigned long long key = m->h_box[k], u = (key >> 63) ? (key & 0x7FFFFFFFFFFFFFFFull) : ~key;
    std::memcpy(&m->bbox[k], &u, sizeof(double));
  }
  if (out_bbox6) std::memcpy(out_bbox6, m->bbox, sizeof(m->bbox));
  m->posed = true;
  return RT_OK;
} RT_NO_UNWIND

int rt_debug_box_tests(uint32_t n, const double* origin, const double* direction, const float* boxes, const double* limit, int all_t,
                       uint8_t* out) try {
  if (!origin || !direction || !boxes || !limit || !out) return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments to rt_debug_box_tests");
  if (n == 0) return RT_OK;
  DeviceBuffer b_o, b_d, b_b, b_l, b_out;
  HIP_TRY(b_o.alloc((size_t)n * 24));
  HIP_TRY(b_d.alloc((size_t)n * 24));
  HIP_TRY(b_b.alloc((size_t)n * 24));
  HIP_TRY(b_l.alloc((size_t)n * 8));
  HIP_TRY(b_out.alloc(n));
  HIP_TRY(hipMemcpy(b_o.p, origin, (size_t)n * 24, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b_d.p, direction, (size_t)n * 24, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b_b.p, boxes, (size_t)n * 24, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b_l.p, limit, (size_t)n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(debug_box_test_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, (const double*)b_o.p, (const double*)b_d.p,
                     (const float*)b_b.p, (const double*)b_l.p, all_t, (uint8_t*)b_out.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, b_out.p, n, hipMemcpyDeviceToHost));
  return RT_OK;
} RT_NO_UNWIND

long long rt_scene_debug_table(const rt_scene* scene, int which, void* out, unsigned long long capacity) try {
  if (!scene || which < 0 || which > 6 || (!out && capacity)) return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments to rt_scene_debug_table");
  const void* src[7] = {scene->dev.tri_isect, scene->dev.tri_norm, scene->dev.prim_rank, scene->dev.prim_info,
                        scene->dev.entries,   scene->dev.nodes,    scene->dev.prim_flags};
  const size_t bytes = scene->table_bytes[which], n = (size_t)std::min<unsigned long long>(bytes, capacity);
  if (n) HIP_TRY(hipMemcpy(out, src[which], n, hipMemcpyDeviceToHost));
  return (long long)bytes;
} RT_NO_UNWIND

void rt_scene_destroy(rt_scene* scene) { delete scene; }

void rt_release_cached_memory(void) {
  g_device_pool.clear();
  g_download_stage.release();
}

void rt_shutdown(void) {
  // (scenes are the caller's to destroy; what the library keeps for itself are the pools)
  (void)hipDeviceSynchronize();
  rt_release_cached_memory();
  g_pinned_slots.release();
}

uint32_t rt_tile_count(uint32_t width, uint32_t height, uint32_t part, uint32_t n_parts) {
  if (n_parts == 0 || part >= n_parts) return 0;
  uint32_t n = tiles_total(width, height, nullptr);
  return n > part ? (n - part + n_parts - 1) / n_parts : 0;
}

int rt_render_tiles_device(const rt_scene* scene, const rt_camera* camera, const rt_options* options, uint32_t part,
                           uint32_t n_parts, double* d_out_tiles, void* hip_stream, rt_stats* stats) try {
  return launch_render(scene, camera, options, part, n_parts, 1, d_out_tiles, (hipStream_t)hip_stream, stats);
} RT_NO_UNWIND

int rt_render_device(const rt_scene* scene, const rt_camera* camera, const rt_options* options, double* d_out_rgb, void* hip_stream,
                     rt_stats* stats) try {
  return launch_render(scene, camera, options, 0, 1, 0, d_out_rgb, (hipStream_t)hip_stream, stats);
} RT_NO_UNWIND

int rt_render_device_noalloc(const rt_scene* scene, const rt_camera* camera, const rt_options* options, double* d_out_rgb, void* hip_stream) try {
  int rc = check_render_args(scene, camera, options);
  if (rc) return rc;
  if (!d_out_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "output buffer is NULL");
  if (options->reference_order || (options->pipeline != RT_PIPELINE_STREAM && options->pipeline != RT_PIPELINE_STREAM_ALWAYS))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_device_noalloc renders with the streamed production pipeline only");
  if (scene->general_csg)
    return fail(RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: this scene takes the reference-order kernels (a CSG object beyond the production kernels' bounds or with an operand that is not a closed surface, or a cone whose cap sticks out of its box), which only the allocating entry points run");
  char* base = (char*)scene->noalloc_scratch;
  if (!base) return fail(RT_ERR_DEVICE, "the static scratch array could not be located on this device (hipGetSymbolAddress)");
  hipStream_t st = (hipStream_t)hip_stream;
  uint32_t tiles_x;
  const uint32_t n_tiles = tiles_total(camera->width, camera->height, &tiles_x);
  DevOptions dopt = dev_options(options);
  const uint32_t per_tile = 64u * dopt.n_offsets * dopt.n_offsets;
  const uint32_t ratio = scene->nodes_per_sample;
  // the largest whole number of tiles whose batch fits the array
  stream::Ws w{};
  uint64_t lo = 0, hi = std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, RT_NOALLOC_SCRATCH_BYTES / per_tile));
  while (lo < hi) {
    uint64_t mid = (lo + hi + 1) / 2;
    if (carve_workspace(nullptr, nullptr, mid * per_tile, ratio, scene->n_lights) <= RT_NOALLOC_SCRATCH_BYTES) lo = mid;
    else hi = mid - 1;
  }
  if (lo == 0)
    return fail(RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: one 8x8 tile of this scene and these options does not fit the " +
                                        std::to_string(RT_NOALLOC_SCRATCH_BYTES >> 20) + " MiB scratch array");
  const uint64_t n_batches = (n_tiles + lo - 1) / lo;
  if (n_batches > RT_NOALLOC_MAX_BATCHES)
    return fail(RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: the frame has " + std::to_string((uint64_t)n_tiles * per_tile) +
                                        " samples; at most " + std::to_string(lo * per_tile * RT_NOALLOC_MAX_BATCHES) +
                                        " (RT_NOALLOC_MAX_BATCHES batches of the scratch array) are rendered for this scene and these options");
  const uint32_t batch_samples = (uint32_t)(lo * per_tile);
  carve_workspace(&w, base, batch_samples, ratio, scene->n_lights);
  // Under stream capture nothing may synchronise and an event record would become a node of the graph: then the scene's
  // done_event is left alone (the caller keeps the scene alive until the graph's launches have finished, include/rtcore.h).
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &capture) != hipSuccess) {
    (void)hipGetLastError();
    capture = hipStreamCaptureStatusNone;
  }
  const bool capturing = capture != hipStreamCaptureStatusNone;
  HIP_TRY(hipMemsetAsync(scene->d_overflow, 0, sizeof(uint32_t), st));
  // every kernel on the caller's stream (no side stream, no cross-stream events): allow_side = false
  rc = stream_render_with<false>(scene, w, batch_samples, camera, dopt, 0, 1, tiles_x, n_tiles, 0, d_out_rgb, st, nullptr, false);
  if (rc) {
    if (!capturing) (void)hipStreamSynchronize(st);
    return rc;
  }
  // (the fallback kernel's frame summary is for the allocating entry points' layout heuristics: not carried over)
  HIP_TRY(hipMemsetAsync(scene->d_summary, 0, 3 * sizeof(uint32_t), st));
  if (!capturing) {
    scene->rendered = true;
    HIP_TRY(hipEventRecord(scene->done_event, st));
  }
  return RT_OK;
} RT_NO_UNWIND

int rt_assemble_tiles_device(uint32_t width, uint32_t height, uint32_t n_parts, uint32_t part_stride_tiles,
                             const double* d_tiles_all_parts, double* d_out_rgb, void* hip_stream) try {
  return assemble_tiles(width, height, n_parts, part_stride_tiles, d_tiles_all_parts, d_out_rgb, false, (hipStream_t)hip_stream);
} RT_NO_UNWIND

int rt_assemble_tiles_u8_device(uint32_t width, uint32_t height, uint32_t n_parts, uint32_t part_stride_tiles,
                                const uint8_t* d_tiles_all_parts, uint8_t* d_out_rgb8, void* hip_stream) try {
  return assemble_tiles(width, height, n_parts, part_stride_tiles, d_tiles_all_parts, d_out_rgb8, true, (hipStream_t)hip_stream);
} RT_NO_UNWIND

int rt_quantize_device(const double* d_rgb, uint64_t n_values, uint8_t* d_out_u8, void* hip_stream) try {
  if (!d_rgb || !d_out_u8) return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments to rt_quantize_device");
  if (n_values == 0) return RT_OK;
  hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)((n_values + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, d_rgb,
                     n_values, d_out_u8);
  HIP_TRY(hipGetLastError());
  return RT_OK;
} RT_NO_UNWIND

int rt_trace_rays(const rt_scene* scene, const rt_options* options, uint32_t n, const double* origin, const double* direction,
                  double* out_rgb) try {
  if (!scene || !options || !origin || !direction || !out_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments to rt_trace_rays");
  if (options->max_depth > RT_MAX_DEPTH) return fail(RT_ERR_UNSUPPORTED, "max_depth exceeds RT_MAX_DEPTH");
  if (n == 0) return RT_OK;
  size_t bytes = (size_t)n * 3 * sizeof(double);
  DeviceBuffer b_o, b_d, b_c;
  HIP_TRY(b_o.alloc(bytes));
  HIP_TRY(b_d.alloc(bytes));
  HIP_TRY(b_c.alloc(bytes));
  double *d_o = (double*)b_o.p, *d_d = (double*)b_d.p, *d_c = (double*)b_c.p;
  int rc = RT_OK;
  do {
    if (hipMemcpy(d_o, origin, bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_d, direction, bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(scene->d_overflow, 0, sizeof(uint32_t)) != hipSuccess) {
      rc = fail(RT_ERR_DEVICE, "upload of rays failed");
      break;
    }
    if (scene->general_csg) {
      hipLaunchKernelGGL((trace_rays_kernel<true, true>), dim3((n + 63) / 64), dim3(64), 0, 0, scene->dev, dev_options(options), n, d_o, d_d,
                         d_c, scene->d_overflow);
    } else if (options->reference_order) {
      hipLaunchKernelGGL(trace_rays_kernel<true>, dim3((n + 63) / 64), dim3(64), 0, 0, scene->dev, dev_options(options), n, d_o, d_d,
                         d_c, scene->d_overflow);
    } else if (options->pipeline == RT_PIPELINE_SINGLE_KERNEL) {
      hipLaunchKernelGGL(trace_rays_kernel<false>, dim3((n + 63) / 64), dim3(64), 0, 0, scene->dev, dev_options(options), n, d_o, d_d,
                         d_c, scene->d_overflow);
    } else {  // the streamed pipeline with the given rays as its level-0 queue
      uint32_t batch = 0;
      rc = ensure_workspace(scene, n, &batch);
      if (rc) break;
      rt_camera no_cam;
      std::memset(&no_cam, 0, sizeof(no_cam));
      DevOptions dopt = dev_options(options);
      for (uint32_t at = 0; at < n && rc == RT_OK; at += batch) {
        stream::Batch b;
        std::memset(&b, 0, sizeof(b));
        b.n_parts = 1;
        b.n2 = 1;
        b.n_samples = std::min(batch, n - at);
        b.given_o = d_o + (size_t)at * 3;
        b.given_d = d_d + (size_t)at * 3;
        // a ray that hits nothing stores no colour: black (integrator.rs:62-65)
        if (hipMemsetAsync(scene->ws.sample_rgb, 0, (size_t)b.n_samples * 24, 0) != hipSuccess) {
          rc = fail(RT_ERR_DEVICE, "hipMemsetAsync(colours) failed");
          break;
        }
        rc = run_stream_batch<false, false>(scene, scene->ws, no_cam, dopt, b, 0);
        if (rc == RT_OK && hipMemcpyAsync(d_c + (size_t)at * 3, scene->ws.sample_rgb, (size_t)b.n_samples * 24, hipMemcpyDeviceToDevice, 0) != hipSuccess)
          rc = fail(RT_ERR_DEVICE, "copy of the traced colours failed");
      }
      if (rc) break;
    }
    uint32_t ovf = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out_rgb, d_c, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&ovf, scene->d_overflow, sizeof(ovf), hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(RT_ERR_DEVICE, std::string("trace_rays kernel failed: ") + hipGetErrorString(hipGetLastError()));
      break;
    }
    if (ovf) rc = fail(RT_ERR_OVERFLOW, "a per-ray bounded list overflowed on the device");
  } while (0);
  return rc;
} RT_NO_UNWIND

int rt_render(const rt_scene_desc* scene_desc, const rt_camera* camera, const rt_options* options, double* out_rgb, rt_stats* stats) try {
  return render_one_shot(scene_desc, camera, options, out_rgb, false, stats);
} RT_NO_UNWIND

int rt_render_u8(const rt_scene_desc* scene_desc, const rt_camera* camera, const rt_options* options, uint8_t* out_rgb8, rt_stats* stats) try {
  return render_one_shot(scene_desc, camera, options, out_rgb8, true, stats);
} RT_NO_UNWIND

}  // extern "C"

namespace {
int render_one_shot(const rt_scene_desc* scene_desc, const rt_camera* camera, const rt_options* options, void* out_rgb, bool u8, rt_stats* stats) {
  if (!out_rgb || !camera) return fail(RT_ERR_INVALID_ARGUMENT, "out_rgb and camera are required");
  if (!scene_desc || !options) return fail(RT_ERR_INVALID_ARGUMENT, "scene and options are required");
  {
    std::vector<int> devs = device_list(options);
    if (devs.size() > 1) return render_on_devices(devs, scene_desc, camera, options, out_rgb, u8, stats);
  }
  Event e0, e1, e2, e3;  // (destroyed on every path)
  if (stats) {
    HIP_TRY(e0.create());
    HIP_TRY(e1.create());
    HIP_TRY(e2.create());
    HIP_TRY(e3.create());
    HIP_TRY(hipEventRecord(e0.e, 0));
  }
  rt_scene* sc = nullptr;
  int rc = rt_scene_create(scene_desc, &sc);
  if (rc) return rc;
  struct SceneGuard {
    rt_scene* s;
    ~SceneGuard() { rt_scene_destroy(s); }
  } guard{sc};
  if (stats) HIP_TRY(hipEventRecord(e1.e, 0));
  // the image: a block of the device pool (an animation through this entry point renders frame after frame); its u8 copy behind it
  const size_t n_values = (size_t)camera->width * camera->height * 3;
  size_t bytes = n_values * sizeof(double);
  struct PoolBlock {
    void* p = nullptr;
    size_t cap = 0;
    int device = 0;
    ~PoolBlock() { g_device_pool.give(p, cap, device); }
  } image;
  image.device = sc->device;
  image.p = g_device_pool.take((bytes ? bytes : 16) + (u8 ? n_values : 0), sc->device, &image.cap);
  if (!image.p) return fail(RT_ERR_DEVICE, "hipMalloc(image) failed");
  uint8_t* image_u8 = (uint8_t*)image.p + bytes;
  // stats == NULL (what the reference-side hook passes): the production kernels, nothing synchronises before the download
  rt_stats local;
  rc = launch_render(sc, camera, options, 0, 1, 0, (double*)image.p, 0, stats ? &local : nullptr);
  if (rc == RT_OK || rc == RT_ERR_OVERFLOW) {
    if (stats) (void)hipEventRecord(e2.e, 0);
    if (u8 && n_values) {  // Color::as_scaled_values on the device: 3 bytes per pixel come back instead of 24
      hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)((n_values + 255) / 256)), dim3(256), 0, 0, (const double*)image.p, (uint64_t)n_values, image_u8);
      if (hipGetLastError() != hipSuccess) rc = fail(RT_ERR_DEVICE, "quantize launch failed");
    }
    if (download_image(out_rgb, u8 ? (const void*)image_u8 : image.p, u8 ? n_values : bytes) != hipSuccess)
      rc = fail(RT_ERR_DEVICE, "download of the image failed");
    if (!stats && rc == RT_OK) {  // (with stats launch_render has looked already; the download above has waited for the frame)
      uint32_t ovf = 0;
      if (hipMemcpy(&ovf, sc->d_overflow, sizeof(ovf), hipMemcpyDeviceToHost) == hipSuccess && ovf)
        rc = fail(RT_ERR_OVERFLOW, "a per-ray bounded list (BVH stack or CSG hit list) overflowed on the device");
    }
    if (stats) {
      (void)hipEventRecord(e3.e, 0);
      (void)hipEventSynchronize(e3.e);
      float up = 0.f, down = 0.f;
      (void)hipEventElapsedTime(&up, e0.e, e1.e);
      (void)hipEventElapsedTime(&down, e2.e, e3.e);
      local.ms_upload = up;
      local.ms_download = down;
      *stats = local;
    }
  } else {
    (void)hipStreamSynchronize(0);
  }
  // (the download has waited for the frame: its summary is in; ~rt_scene remembers what it says about this kind of scene)
  absorb_summary(sc);
  return rc;
}
}  // namespace
