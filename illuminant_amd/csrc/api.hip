// api.hip -- implementation of the C ABI declared in include/illuminant_hip.h.
// Owns device memory behind opaque handles; validates arguments the way the
// reference's host code does (same limits, same failure conditions) and turns
// them into return codes that the C# wrapper rethrows as exceptions.
#include <atomic>
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "internal.hpp"

namespace {

using namespace ilm;

constexpr int kPlanePad = 1088;    // floats (a multiple of 64: planes stay 256-byte aligned)

thread_local char g_last_error[512] = "";

int32_t fail(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail((int32_t)_e, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// Inside a create function, after the object has been registered: a failing HIP call destroys the half-built object (its
// destroy entry point releases whatever was allocated so far; the error text of the failing call is kept) instead of leaking it.
#define HIP_TRY_OR_DESTROY(expr, destroy_call)                                                 \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            const int32_t _rc = fail((int32_t)_e, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            (void)(destroy_call);                                                              \
            return _rc;                                                                        \
        }                                                                                      \
    } while (0)

enum : uint32_t {
    kMagicCtx = 0x494C4D43u, kMagicEngine = 0x494C4D45u, kMagicSystem = 0x494C4D53u,
    kMagicSdf = 0x494C4D44u, kMagicGBuffer = 0x494C4D47u, kMagicLightmap = 0x494C4D4Cu
};

struct Ctx {
    uint32_t magic = kMagicCtx;
    int device = 0;
    // Sibling contexts (ilm_ctx_create_sibling, r05): contexts of one device that keep several frames in flight -- own stream, own scratch,
    // own lightmaps -- and may READ each other's distance fields and G-buffers in the light passes (Shared below orders those reads
    // against the owner's writes).  `family` = a process-wide serial number the first sibling call hands out; 0 = no siblings.
    uintptr_t family = 0;
    int children = 0;            // live engines / distance fields / G-buffers / lightmaps: the context cannot be destroyed under them
    std::vector<struct Lightmap*> mirrored;      // this context's lightmaps with an armed store-mode table (lightmap_set_mirrors), for their aliases
    // Two streams.  Everything is ordered on `stream_`; ilm_system_step may put the second half of a large step's chunk range on `aux`
    // (chunks never interact, ParticleSystem.cs:743-745: one half's launch tail is covered by the other half's launch, run_step).
    // The halves are only joined when something else needs them: every entry point takes its stream from main(), which makes
    // stream_ wait for what aux has queued and notes that aux must wait for stream_ before its next launch.  Back-to-back steps
    // therefore cost no event at all (a record / wait pair is a ~7 us bubble on this runtime).
    hipStream_t stream_ = nullptr, aux = nullptr;
    hipEvent_t ev_main = nullptr, ev_aux = nullptr;
    bool aux_pending = false;    // aux holds launches stream_ has not waited for
    bool main_pending = true;    // stream_ holds work aux has not waited for
    bool exported = false;       // ilm_ctx_stream handed stream_ to the caller, who may queue readers of the particle planes on it: no split
    hipStream_t main() {
        if (aux_pending) {
            (void)hipEventRecord(ev_aux, aux);
            (void)hipStreamWaitEvent(stream_, ev_aux, 0);
            aux_pending = false;
        }
        main_pending = true;
        return stream_;
    }
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // device staging for AoS <-> SoA conversion
    void* staging = nullptr; size_t staging_bytes = 0;
    // pinned ring for small asynchronous parameter uploads (light arrays)
    static constexpr int kRing = 4;
    void* pinned[kRing] = {}; size_t pinned_bytes[kRing] = {}; hipEvent_t pinned_ev[kRing] = {}; int ring_pos = 0;
    IlmLightVertex* d_lights = nullptr; void* d_recs = nullptr; int light_cap = 0;
    unsigned long long* d_stats = nullptr;
    // light split (plan_light_split): the tiles' per-part sums and their tickets
    int last_light_blocks = 0, last_light_split = 1, last_light_macro = 0;       // the last tile-kernel launch (ilm_debug_last_light_launch)
    float4* d_light_partials = nullptr; size_t light_partials_tiles = 0; uint32_t* d_light_tickets = nullptr; size_t light_tickets_cap = 0;
    int light_split = 0;                  // ilm_ctx_set_light_split: 0 = chosen per launch, else 1 / 2 / 4 / 8
    uint16_t* d_group_order = nullptr; int group_order_cap = 0; uint64_t group_order_key = 0; int group_order_groups = 0;
    // particle lights: records compacted on the device + their count, block counts, per-chunk quad counts
    void* d_pl_recs = nullptr; int pl_cap = 0; int32_t* d_pl_count = nullptr; int32_t* d_pl_blocks = nullptr; int pl_blocks_cap = 0;
    int32_t* d_pl_quads = nullptr; int pl_quads_cap = 0;
    float4* d_light_ramp = nullptr; int light_ramp_w = 0, light_ramp_h = 0;    // RampTexture of the light group being rendered
    int lightmap_blend = 0;               // ILM_BLEND_FP32_ACCUMULATE / ILM_BLEND_FP16_PER_LIGHT (ilm_ctx_set_lightmap_blend)
    RasterScratch raster;                                         // particle rasteriser buffers (raster.hip)
    int32_t* d_raster_quads = nullptr; int raster_quads_cap = 0;
    // particle read-back: draw-call records, total, block counts, per-chunk element counts
    IlmReadbackDrawCall* d_rb = nullptr; int rb_cap = 0; int32_t* d_rb_count = nullptr; int32_t* d_rb_blocks = nullptr; int rb_blocks_cap = 0;
    int32_t* d_rb_elems = nullptr; int rb_elems_cap = 0;
    IlmReadbackDrawCall* h_rb = nullptr; size_t h_rb_cap = 0;    // pinned host buffer the read-back records land in
    // light probes
    float4* d_probe_pairs = nullptr; size_t probe_pairs_cap = 0;   // one contribution per (light, probe)
    // parameter block of the distance-field generation pass (slice list, obstruction records, volumes, polygon vertices)
    void* d_field_params = nullptr; size_t field_params_bytes = 0;
};

struct Engine {
    uint32_t magic = kMagicEngine;
    Ctx* ctx = nullptr;
    int children = 0;            // live systems: the engine cannot be destroyed under them
    int chunk_size = 0, slots = 0;
    int64_t stride = 0;     // floats between the component planes of a chunk: span + kPlanePad
    int32_t span = 0;       // slots rounded up to kSlotsPerBlock
    float4* rnd = nullptr; int rw = 0, rh = 0;
    std::vector<float4> h_rnd;   // host copy: the uniform noise deltas are evaluated on the host (fill_noise_fast)
    uint2* rnd_lp = nullptr;   // LowPrecisionRandomnessTexture: the Rgba64 copy (ParticleEngine.cs:508-540)
    // Chunk pool (r05).  The reference's engine keeps its released buffer sets for reuse (AvailableBuffers / DiscardedBuffers, at most
    // SpareBufferCount = 20 spares: ParticleEngine.cs:53-58,145-170,402-419) because creating a render target in the middle of a frame is
    // expensive; so is hipMalloc: a Spawner that fills a 256^2 chunk every 60 steps paid ~70 us for the 61st (tools/host_cost_probe_cfg2.py).
    // Chunks are carved out of slabs of up to 8 (64 MB at most; one for the large chunk sizes), zero-filled on the context stream when a
    // slab is allocated and again when a chunk comes back; `spare` holds the ready ones.  Beyond kSpareChunks spares a slab whose chunks are
    // ALL spare is freed (release_chunk); the rest live until the engine is destroyed.
    static constexpr int kSpareChunks = 20;
    struct Slab { float* base; int chunks; };
    std::vector<Slab> slabs;
    std::vector<float*> spare;
    size_t chunk_bytes() const { return sizeof(float) * (size_t)kComponents * (size_t)stride; }
};

// A distance field or G-buffer that sibling contexts read (light passes on another context's stream).  Every WRITE to it happens on its
// owner's stream (uploads, generation, the field's cell rebuild); a foreign read is ordered behind them by an event recorded on the
// owner's stream when the read is queued, and the owner's later writes wait for the foreign reads' events.  Nothing here costs an
// object that only its owner uses a call.
struct Shared {
    hipEvent_t owner_ev = nullptr;                                   // "everything the owner has queued so far"
    std::vector<std::pair<Ctx*, hipEvent_t>> readers;                // per foreign context: its last queued read
    std::vector<bool> pending;
    void release() {
        if (owner_ev) (void)hipEventDestroy(owner_ev);
        for (auto& r : readers) (void)hipEventDestroy(r.second);
        owner_ev = nullptr; readers.clear(); pending.clear();
    }
};
// the owner is about to write: its stream waits for the sibling contexts' queued reads
inline hipError_t shared_before_write(Shared& sh, Ctx* owner) {
#ifdef ILM_EXP_NO_SHARED_ORDER      // EXPERIMENT (negative control of tests/test_frames_in_flight_gpu.py: without the ordering the test must fail)
    return hipSuccess;
#endif
    for (size_t i = 0; i < sh.readers.size(); i++)
        if (sh.pending[i]) {
            const hipError_t e = hipStreamWaitEvent(owner->main(), sh.readers[i].second, 0);
            if (e != hipSuccess) return e;
            sh.pending[i] = false;
        }
    return hipSuccess;
}
// `reader` (a sibling of the owner) is about to queue a read on its own stream: behind whatever the owner has queued
inline hipError_t shared_before_read(Shared& sh, Ctx* owner, Ctx* reader) {
#ifdef ILM_EXP_NO_SHARED_ORDER
    return hipSuccess;
#endif
    if (!sh.owner_ev) { const hipError_t e = hipEventCreateWithFlags(&sh.owner_ev, hipEventDisableTiming); if (e != hipSuccess) return e; }
    hipError_t e = hipEventRecord(sh.owner_ev, owner->main());
    if (e != hipSuccess) return e;
    return hipStreamWaitEvent(reader->main(), sh.owner_ev, 0);
}
// ... and has queued it
inline hipError_t shared_after_read(Shared& sh, Ctx* reader) {
    size_t i = 0;
    while (i < sh.readers.size() && sh.readers[i].first != reader) i++;
    if (i == sh.readers.size()) {
        hipEvent_t ev = nullptr;
        const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
        sh.readers.emplace_back(reader, ev); sh.pending.push_back(false);
    }
    const hipError_t e = hipEventRecord(sh.readers[i].second, reader->main());
    if (e == hipSuccess) sh.pending[i] = true;
    return e;
}
inline bool siblings(const Ctx* a, const Ctx* b) { return a == b || (a->family != 0 && a->family == b->family && a->device == b->device); }

struct Sdf {
    uint32_t magic = kMagicSdf;
    Ctx* ctx = nullptr;
    Shared shared;
    uint2* texels = nullptr; int width = 0, height = 0, format = 0;
    // The cone trace's view of the field (hlsl_math.hpp, SdfView::cells): built from the atlas on demand by ensure_sdf_cells and kept
    // until the atlas changes (`version` is bumped by every entry point that writes it) or is described by other uniforms (the layout
    // key).  A field whose device pointer was handed out (ilm_sdf_device_ptr) may change behind the library's back: its cells are
    // rebuilt before every use.
    void* cells = nullptr; size_t cells_bytes = 0;
    uint64_t version = 1, cells_version = 0;
    // The particle path's cells (SdfView::cells0, r06): channel r of the four taps of every slice-0 lookup, built on demand by
    // ensure_slice0_cells for the lean collision step and kept until the atlas changes (same versioning as `cells`).
    void* cells0 = nullptr; size_t cells0_bytes = 0; uint64_t cells0_version = 0;
    int cells_slices = 0, cells_columns = 0, cells_sw = 0, cells_sh = 0;
    bool escaped = false;
    // Which virtual slices of the atlas have been written since the cells were built (bit v of dirty[v / 64]; kMaxTableSlices = 256 bits;
    // slices past that have no cells anyway).  The reference regenerates MaximumFieldUpdatesPerFrame = 1 slice triplet per frame
    // (LightingRenderer.Configuration.cs:91, LightingRenderer.DistanceField.cs:415-464): the cells of slice v hold the channel pairs
    // (v, v + 1), so a triplet [s, s + 3) invalidates the cells of slices s - 1 .. s + 2 -- 4 of cfg5's 33, not all 138 MB.
    static constexpr int kDirtyWords = (kMaxTableSlices + 63) / 64;     // (sized by the build's table limit: -DILM_MAX_TABLE_SLICES may raise it)
    uint64_t dirty[kDirtyWords];
    Sdf() { for (uint64_t& w : dirty) w = ~0ull; }      // nothing has cells yet
    void mark_all_dirty() { for (uint64_t& w : dirty) w = ~0ull; version++; }
    void mark_slices_dirty(int first, int count) {
        // (64-bit bounds: first + count of a caller's ilm_sdf_mark_dirty may not fit an int)
        const int64_t lo = std::max<int64_t>(0, (int64_t)first - 1), hi = std::min<int64_t>((int64_t)first + (int64_t)count, (int64_t)kDirtyWords * 64);
        for (int64_t v = lo; v < hi; v++) dirty[v >> 6] |= 1ull << (v & 63);
        version++;
    }
    // what the last light pass over this field found / did (ilm_sdf_trace_info)
    uint64_t cell_rebuilds = 0, cell_slices_rebuilt = 0;
    int last_rebuilt_slices = 0, last_table_slices = 0;
};

struct GBuffer {
    uint32_t magic = kMagicGBuffer;
    Ctx* ctx = nullptr;
    Shared shared;
    void* texels = nullptr; int width = 0, height = 0, format = 0;
};

struct Lightmap {
    uint32_t magic = kMagicLightmap;
    Ctx* ctx = nullptr;
    void* texels = nullptr; int width = 0, height = 0, format = 0; bool external = false;
    // store-mode exchange of a group lightmap (lightmap_set_mirrors, group.hip): device array of the other members' buffers
    void** d_mirrors = nullptr; int mirror_count = 0;
};

struct System {
    uint32_t magic = kMagicSystem;
    Engine* engine = nullptr;
    std::vector<float*> chunks;
    // High-water mark per chunk: slots >= used[i] have never been written since the chunk was allocated (zero-filled) or
    // erased, so every plane of theirs is zero and a step leaves them exactly as they are -- the launch skips their units.
    // Raised by uploads and spawn ranges; a chunk whose device pointer was handed out is treated as fully used.
    std::vector<int32_t> used;
    float** d_table = nullptr; int table_cap = 0; bool table_dirty = true;
    // Five counter regions of 64-bit words.  Regions 0-3 belong to the step kernels, kCountLines lines per chunk (internal.hpp),
    // index = parity * 2 + half: a counting launch of half h accumulates into (parity, h) and zeroes (parity ^ 1, h) for the next
    // counting step -- no memset launch, and the two halves of a split step (run_step), which run on different streams, never touch
    // each other's lines; a launch that is not split uses half 0 and zeroes both halves of the other parity (they are adjacent).
    // Region 4, one line per chunk, belongs to ilm_system_live_counts.
    unsigned long long* d_counts = nullptr; int counts_cap = 0; int count_parity = 0;
    unsigned long long* counts_region(int r) const { return d_counts + (size_t)r * (size_t)counts_cap * kCountLines * kCountStride; }
    static size_t counts_bytes(int cap) { return sizeof(unsigned long long) * (size_t)cap * kCountStride * (4 * kCountLines + 1); }
    // chunks [.., split_at) of a split step run on the context stream, [split_at, ..) on the second one; moved only after a join
    int split_at = -1;
    IlmHandle sdf_handle = 0;   // bound distance field: resolved through the handle table at every use (it may have been destroyed)
    float4* ramp = nullptr; int ramp_w = 0, ramp_h = 0;
    uint32_t* d_slots = nullptr; int slots_cap = 0; uint32_t* d_slot_count = nullptr;
    // the Spawner's PositionBuffer per spawn record slot (ParticleSpawner.cs:301-353)
    float4* spawn_positions[ILM_MAX_SPAWNS] = {}; int spawn_position_count[ILM_MAX_SPAWNS] = {}; int spawn_position_cap[ILM_MAX_SPAWNS] = {};
    float4* bitmap = nullptr; int bitmap_w = 0, bitmap_h = 0;     // Appearance.Texture for the textured rasterise techniques
    // the PatternSpawner's texture per spawn record slot, mip levels back to back (SpecialSpawners.cs:19-22)
    float4* spawn_pattern[ILM_MAX_SPAWNS] = {}; int pattern_w[ILM_MAX_SPAWNS] = {}, pattern_h[ILM_MAX_SPAWNS] = {}, pattern_levels[ILM_MAX_SPAWNS] = {};
    // The fused live counts arrive in page-locked host memory, written by the step kernel itself (internal.hpp, StepLaunch::host_counts):
    // one word per chunk, sequence number << 32 | count.  A counting step over chunks [counts_first, counts_first + counts_span) is
    // complete when each of their words carries its sequence number; chunks outside the range count zero.
    unsigned long long* h_counts = nullptr; unsigned long long* h_counts_dev = nullptr; int h_counts_cap = 0;
    uint32_t count_seq = 0; int counts_first = 0, counts_span = 0;
    int counts_n = 0; bool counts_pending = false;
    bool counts_valid = false;   // h_counts holds (or is about to receive) the counts of the last counting step
};

SdfView make_sdf_view(const Sdf* f, const IlmDistanceFieldUniforms* df) {
    SdfView v;
    v.texels = f ? f->texels : nullptr;
    v.width = f ? f->width : 0;
    v.height = f ? f->height : 0;
    v.format = f ? f->format : ILM_SDF_UNORM16;
    v.wf = (float)v.width;
    v.hf = (float)v.height;
    v.inv_wf = v.width > 0 ? 1.0f / v.wf : 0.0f;
    // The sampler's single-step U wrap needs every tap column below 2^22 (hlsl_math.hpp); the largest column these uniforms can produce
    // is (floor(maxSlice) / 3 * sliceU + extentX * texelU) * width.  Anything at or above 2^20 (or not finite) keeps the two-fold wrap.
    v.wrap_half = 0.0f;
    v.cells0 = nullptr;        // (run_step binds the slice-0 cells for the lean collision step)
    // The cone trace's in-volume sampler (hlsl_math.hpp, sample_inside_table) needs the uniforms to describe exactly this atlas as
    // columns x rows whole slices with the reference's texel sizes (Uniforms.cs:90-110): then the U WRAP fold of a tap is decided by
    // its slice alone.  Its box: every tap of a sample at least a sixteenth of a texel inside its slice in x and y (tap x0 >= 0,
    // x0 + 1 <= sliceW - 1, likewise y), z between the offset and the last valid / tabulated slice.
    v.table_slices = 0; v.columns = 1;
    v.box_x0 = v.box_y0 = v.box_z0 = 1.0f; v.box_x1 = v.box_y1 = v.box_z1 = 0.0f;      // an empty box
    static const bool table_off = [] { const char* e = getenv("ILM_SDF_TABLE"); return e && e[0] == '0'; }();     // A/B switch
    if (df && v.width > 0 && !table_off) {
        const double cols = df->TextureSliceCount.x, rows = df->TextureSliceCount.y, slices = df->TextureSliceCount.w;
        const double isx = df->ConeAndMisc.w, isy = df->StepAndMisc2.w, ex = df->Extent.x, ey = df->Extent.y, ez = df->Extent.z;
        // ConeAndMisc.w is float(VirtualWidth / SliceWidth) (Uniforms.cs:108-109): for a resolution that is not a dyadic ratio the
        // quotient misses the integer slice size by the float's rounding; the atlas size below decides whether the nearest integer is meant
        auto nearest = [](double x) { const double r = std::floor(x + 0.5); return (std::fabs(x - r) <= 1e-6 * std::fmax(1.0, r)) ? r : x; };   // (a float's rounding: 6e-8; the box's 1/16-texel margin covers 1e-6 x 8192 texels)
        const double sw = nearest(ex / isx), sh = nearest(ey / isy);
        auto whole = [](double x) { return std::isfinite(x) && x >= 1.0 && x == std::floor(x); };
        const bool consistent =
            whole(cols) && whole(rows) && whole(slices) && whole(sw) && whole(sh) && slices <= kMaxTableSlices && cols <= 1024 &&
            cols * sw == (double)v.width && rows * sh == (double)v.height && std::ceil(slices / 3.0) <= cols * rows &&
            std::fabs((double)df->TextureSliceAndTexelSize.x * cols - 1.0) < 1e-6 && std::fabs((double)df->TextureSliceAndTexelSize.y * rows - 1.0) < 1e-6 &&
            std::fabs((double)df->TextureSliceAndTexelSize.z * ex * cols - 1.0) < 1e-6 && std::fabs((double)df->TextureSliceAndTexelSize.w * ey * rows - 1.0) < 1e-6 &&
            ez > 0 && std::isfinite(ez) && std::isfinite((double)df->ConeAndMisc.y) && df->Packed1.y > 0.0f && std::isfinite((double)df->Packed1.y) &&
            df->Packed1.z > 0.0f && sw >= 4 && sh >= 4;
        // the reference's float row index floor(vslice * Packed1.x) must name the atlas row the slice really lies in, for every slice
        bool rows_agree = consistent;
        for (int vi = 0; rows_agree && vi < (int)slices; vi++)
            rows_agree = floorf((float)vi * df->Packed1.x) == (float)((vi / 3) / (int)cols);
        if (consistent && rows_agree) {
            v.table_slices = (int)slices;
            v.columns = (int)cols;
            const double tx = 0.5625 * isx, ty = 0.5625 * isy;       // half a texel + 1/16, in world units
            v.box_x0 = (float)tx; v.box_x1 = (float)(ex - tx);
            v.box_y0 = (float)ty; v.box_y1 = (float)(ey - ty);
            // slice_position = (z - zOffset) * Packed1.y must stay below the table's end and within [0, validZ]
            const double zmax = std::min({ (double)df->Packed1.z, ez, (slices - 1e-3) / (double)df->Packed1.y });
            v.box_z0 = df->ConeAndMisc.y; v.box_z1 = (float)((double)df->ConeAndMisc.y + zmax * (1.0 - 1e-6));
        }
    }
    if (df && v.width > 0) {
        const double third_max = std::floor(std::fabs((double)df->Packed1.z * (double)df->Packed1.y)) / 3.0 + 1.0;
        const double u_max = third_max * std::fabs((double)df->TextureSliceAndTexelSize.x) +
                             std::fabs((double)df->Extent.x * (double)df->TextureSliceAndTexelSize.z) + 1.0;
        const double x_max = u_max * (double)v.width;
        if (x_max == x_max && x_max < 1048576.0)
            v.wrap_half = 0.5f * v.inv_wf;
    }
    return v;
}

// The uniforms of a light / probe pass must describe the atlas that is bound: columns x slice width = atlas width, rows x slice height
// = atlas height (DistanceField ctor, SDF/DistanceField.cs:91-109; slice size = virtual size / InvScaleFactor, Uniforms.cs:108-109).  The
// sampler wraps / clamps its taps into the real atlas whatever the uniforms say, but a frame traced through mismatched uniforms is
// garbage, so it is refused.  Uniforms that do not carry the information (zero / non-finite members, as the particle path leaves
// them) are not judged.
const char* field_uniforms_mismatch(const Sdf* f, const IlmDistanceFieldUniforms* df, char* text, size_t n) {
    if (!f || !df) return nullptr;
    const double cols = df->TextureSliceCount.x, rows = df->TextureSliceCount.y;
    const double sw = (double)df->Extent.x / (double)df->ConeAndMisc.w, sh = (double)df->Extent.y / (double)df->StepAndMisc2.w;
    if (!(cols >= 1 && rows >= 1 && sw >= 1 && sh >= 1) || !std::isfinite(cols * sw) || !std::isfinite(rows * sh)) return nullptr;
    if (std::fabs(cols * sw - (double)f->width) <= 0.5 && std::fabs(rows * sh - (double)f->height) <= 0.5) return nullptr;
    snprintf(text, n, "the distance-field uniforms describe a %.0f x %.0f atlas (%g x %g slices of %g x %g) but the bound field is %d x %d",
             cols * sw, rows * sh, cols, rows, sw, sh, f->width, f->height);
    return text;
}

// The view the cone trace launches with: make_sdf_view + the field's cell array, (re)built on `stream` when the atlas has changed since
// the last build or is described by another layout.  Without cells (ILM_SDF_CELLS=0, a field past the 2 GiB the 32-bit cell offsets
// reach, allocation failure) the view has no table and the trace uses the general sampler -- same results, slower.
hipError_t make_trace_view(Sdf* f, const IlmDistanceFieldUniforms* df, hipStream_t stream, TraceSdfView* out) {
    TraceSdfView v;
    static_cast<SdfView&>(v) = make_sdf_view(f, df);
    v.cells = nullptr; v.cells_bytes = 0; v.slice_w = 0; v.slice_h = 0;
    if (v.table_slices > 0) {       // (make_sdf_view has checked that the uniforms tile the atlas in whole slices)
        v.slice_w = v.width / v.columns;
        v.slice_h = (int)std::floor((double)df->Extent.y / (double)df->StepAndMisc2.w + 0.5);
    }
    *out = v;
    if (!f || v.table_slices <= 0) return hipSuccess;
    static const bool cells_off = [] { const char* e = getenv("ILM_SDF_CELLS"); return e && e[0] == '0'; }();     // A/B switch
    const size_t bytes = (size_t)v.table_slices * (size_t)v.slice_w * (size_t)v.slice_h * 16u;
    auto no_table = [&]() { out->table_slices = 0; out->box_x0 = out->box_y0 = out->box_z0 = 1.0f; out->box_x1 = out->box_y1 = out->box_z1 = 0.0f; };
    if (cells_off || bytes >= ((size_t)1 << 31)) { no_table(); return hipSuccess; }
    const bool same_layout = f->cells && f->cells_slices == v.table_slices && f->cells_columns == v.columns && f->cells_sw == v.slice_w && f->cells_sh == v.slice_h;
    f->last_rebuilt_slices = 0; f->last_table_slices = v.table_slices;
    if (!same_layout || f->escaped || f->cells_version != f->version) {
        // (the rebuild WRITES the cells: on the owner's stream -- `stream` is the owner's whoever asks, see borrowed_trace_view -- and
        // behind the sibling contexts' queued reads)
        { const hipError_t e = shared_before_write(f->shared, f->ctx); if (e != hipSuccess) return e; }
        if (bytes > f->cells_bytes) {
            if (f->cells) {
                (void)hipStreamSynchronize(stream);
                for (auto& r : f->shared.readers) (void)hipEventSynchronize(r.second);       // sibling contexts still tracing through the old array
                (void)hipFree(f->cells); f->cells = nullptr; f->cells_bytes = 0;
            }
            if (hipMalloc(&f->cells, bytes) != hipSuccess) {
                (void)hipGetLastError(); f->cells = nullptr; f->last_table_slices = 0;
                static std::atomic<bool> said{false};
                if (!said.exchange(true)) fprintf(stderr, "illuminant_hip: no memory for the %zu-byte cell array of a distance field: its light passes use the general sampler (slower)\n", bytes);
                no_table(); return hipSuccess;
            }
            f->cells_bytes = bytes;
            f->mark_all_dirty();
        }
        if (!same_layout || f->escaped) f->mark_all_dirty();
        // the runs of slices written since the last build (everything, the first time and for an escaped atlas)
        int v0 = 0;
        while (v0 < v.table_slices) {
            if (!((f->dirty[v0 >> 6] >> (v0 & 63)) & 1ull)) { v0++; continue; }
            int v1 = v0;
            while (v1 < v.table_slices && ((f->dirty[v1 >> 6] >> (v1 & 63)) & 1ull)) v1++;
            const hipError_t e = launch_build_sdf_cells(v, f->cells, v0, v1 - v0, stream);
            if (e != hipSuccess) return e;
            f->last_rebuilt_slices += v1 - v0;
            v0 = v1;
        }
        for (uint64_t& w : f->dirty) w = 0;
        f->cell_rebuilds++; f->cell_slices_rebuilt += (uint64_t)f->last_rebuilt_slices;
        f->cells_slices = v.table_slices; f->cells_columns = v.columns; f->cells_sw = v.slice_w; f->cells_sh = v.slice_h;
        f->cells_version = f->version;
    }
    out->cells = f->cells;
    out->cells_bytes = (uint32_t)bytes;
    return hipSuccess;
}

// The field (and G-buffer) of a light pass of context `c`: its own, or a sibling context's (ilm_ctx_create_sibling).  A borrowed field's
// cells are (re)built on ITS OWNER's stream, then c's stream is ordered behind everything the owner has queued; after the pass has been
// queued, light_pass_queued() leaves the marks the owner's next write waits for.
hipError_t borrowed_trace_view(Ctx* c, Sdf* f, GBuffer* g, const IlmDistanceFieldUniforms* df, TraceSdfView* out) {
    hipError_t e = make_trace_view(f, df, f ? f->ctx->main() : c->main(), out);
    if (e != hipSuccess) return e;
    if (f && f->ctx != c) { e = shared_before_read(f->shared, f->ctx, c); if (e != hipSuccess) return e; }
    if (g && g->ctx != c) { e = shared_before_read(g->shared, g->ctx, c); if (e != hipSuccess) return e; }
    return hipSuccess;
}
hipError_t light_pass_queued(Ctx* c, Sdf* f, GBuffer* g) {
    if (f && f->ctx != c) { const hipError_t e = shared_after_read(f->shared, c); if (e != hipSuccess) return e; }
    if (g && g->ctx != c) { const hipError_t e = shared_after_read(g->shared, c); if (e != hipSuccess) return e; }
    return hipSuccess;
}

// Handles are the object addresses, but an address is only trusted after it has been found in this table: a stale, foreign or
// garbage handle is answered with ILM_ERR_INVALID_HANDLE instead of being dereferenced (the C# side owns handle lifetimes and can
// hand in anything).  One mutex-protected lookup per entry point: ~50 ns against microseconds of launch cost.
struct HandleRegistry {
    std::mutex mutex;
    std::unordered_map<uintptr_t, uint32_t> live;     // address -> magic of the object type
};
static HandleRegistry& handle_registry() {
    static HandleRegistry* r = new HandleRegistry();  // never destroyed: entry points may run during process teardown
    return *r;
}
template <typename T>
T* from_handle(IlmHandle h, uint32_t magic) {
    HandleRegistry& r = handle_registry();
    std::lock_guard<std::mutex> lock(r.mutex);
    const auto it = r.live.find(static_cast<uintptr_t>(h));
    if (it == r.live.end() || it->second != magic)
        return nullptr;
    return reinterpret_cast<T*>(static_cast<uintptr_t>(h));
}
template <typename T>
IlmHandle to_handle(T* p) {
    HandleRegistry& r = handle_registry();
    std::lock_guard<std::mutex> lock(r.mutex);
    r.live[reinterpret_cast<uintptr_t>(p)] = p->magic;
    return static_cast<IlmHandle>(reinterpret_cast<uintptr_t>(p));
}
static void retire_handle(const void* p) {
    HandleRegistry& r = handle_registry();
    std::lock_guard<std::mutex> lock(r.mutex);
    r.live.erase(reinterpret_cast<uintptr_t>(p));
}

size_t lightmap_texel_bytes(int format) {
    return format == ILM_LIGHTMAP_FLOAT4 ? 16 : (format == ILM_LIGHTMAP_HALF4 ? 8 : 4);
}

int32_t ensure_staging(Ctx* c, size_t bytes) {
    if (bytes <= c->staging_bytes)
        return ILM_OK;
    if (c->staging) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        HIP_TRY(hipFree(c->staging));
        c->staging = nullptr; c->staging_bytes = 0;
    }
    size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes;
    HIP_TRY(hipMalloc(&c->staging, cap));
    c->staging_bytes = cap;
    return ILM_OK;
}

// the next slot of the pinned ring, free and at least `bytes` large: the caller fills *host and calls upload_small_commit
int32_t upload_small_begin(Ctx* c, size_t bytes, void** host, int* slot_out) {
    const int slot = c->ring_pos;
    c->ring_pos = (c->ring_pos + 1) % Ctx::kRing;
    if (c->pinned_ev[slot] == nullptr)
        HIP_TRY(hipEventCreateWithFlags(&c->pinned_ev[slot], hipEventDisableTiming));
    else
        HIP_TRY(hipEventSynchronize(c->pinned_ev[slot]));
    if (c->pinned_bytes[slot] < bytes) {
        if (c->pinned[slot]) HIP_TRY(hipHostFree(c->pinned[slot]));
        c->pinned[slot] = nullptr; c->pinned_bytes[slot] = 0;
        size_t cap = bytes < 65536 ? 65536 : bytes;
        HIP_TRY(hipHostMalloc(&c->pinned[slot], cap, hipHostMallocDefault));
        c->pinned_bytes[slot] = cap;
    }
    *host = c->pinned[slot];
    *slot_out = slot;
    return ILM_OK;
}
// Between 4 KB and 2 MB a kernel that reads the pinned slot and writes device memory gets the block there two to three times sooner
// than the copy engine does (tools/ubench/upload.hip: 32 KB in front of a dependent kernel 18.7 us as hipMemcpyAsync, 6.3 us as a
// kernel; 262 KB 22.9 | 9.3; equal below 4 KB; the engine wins from 4 MB on); the host's cost is the same launch.
__global__ __launch_bounds__(256) void copy_from_pinned_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t words, size_t bytes) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) {
        const size_t at = (bytes & ~(size_t)15) + threadIdx.x;
        reinterpret_cast<unsigned char*>(dst)[at] = reinterpret_cast<const unsigned char*>(src)[at];
    }
}
int32_t upload_small_commit(Ctx* c, void* dst, int slot, size_t bytes) {
    static const int by_kernel = [] { const char* e = getenv("ILM_UPLOAD_BY_KERNEL"); return e ? atoi(e) : 1; }();
    if (by_kernel && bytes > 4096 && bytes <= ((size_t)2 << 20) && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        void* dv = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dv, c->pinned[slot], 0));
        const size_t words = bytes / 16;
        const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 1024);
        hipLaunchKernelGGL(copy_from_pinned_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, c->main(), static_cast<const uint4*>(dv), static_cast<uint4*>(dst), words, bytes);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(hipMemcpyAsync(dst, c->pinned[slot], bytes, hipMemcpyHostToDevice, c->main()));
    }
    HIP_TRY(hipEventRecord(c->pinned_ev[slot], c->main()));
    return ILM_OK;
}

// copy a small host block to the device through the pinned ring (asynchronous)
int32_t upload_small(Ctx* c, void* dst, const void* src, size_t bytes) {
    void* host = nullptr;
    int slot = -1;
    const int32_t rc = upload_small_begin(c, bytes, &host, &slot);
    if (rc != ILM_OK) return rc;
    memcpy(host, src, bytes);
    return upload_small_commit(c, dst, slot, bytes);
}

// The same ring, read in place: `src` is copied into a pinned slot and the slot's device-visible address is returned; the caller queues the
// kernel that reads it on the context stream and then calls staged_small_done.  One device operation (and one dependent-launch gap)
// less than upload_small + kernel: the lights of a frame are read once, by the one kernel that digests them.
int32_t stage_small(Ctx* c, const void* src, size_t bytes, const void** device_visible, int* slot_out) {
    const int slot = c->ring_pos;
    c->ring_pos = (c->ring_pos + 1) % Ctx::kRing;
    if (c->pinned_ev[slot] == nullptr)
        HIP_TRY(hipEventCreateWithFlags(&c->pinned_ev[slot], hipEventDisableTiming));
    else
        HIP_TRY(hipEventSynchronize(c->pinned_ev[slot]));
    if (c->pinned_bytes[slot] < bytes) {
        if (c->pinned[slot]) HIP_TRY(hipHostFree(c->pinned[slot]));
        c->pinned[slot] = nullptr; c->pinned_bytes[slot] = 0;
        size_t cap = bytes < 65536 ? 65536 : bytes;
        HIP_TRY(hipHostMalloc(&c->pinned[slot], cap, hipHostMallocDefault));
        c->pinned_bytes[slot] = cap;
    }
    memcpy(c->pinned[slot], src, bytes);
    void* dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dev, c->pinned[slot], 0));
    *device_visible = dev;
    *slot_out = slot;
    return ILM_OK;
}
int32_t staged_small_done(Ctx* c, int slot) {
    HIP_TRY(hipEventRecord(c->pinned_ev[slot], c->main()));
    return ILM_OK;
}

int32_t refresh_table(System* s) {
    Ctx* c = s->engine->ctx;
    const int n = (int)s->chunks.size();
    if (n > s->table_cap) {
        if (s->d_table) { HIP_TRY(hipStreamSynchronize(c->main())); HIP_TRY(hipFree(s->d_table)); s->d_table = nullptr; }
        int cap = n < 64 ? 64 : n * 2;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_table), sizeof(float*) * (size_t)cap));
        s->table_cap = cap;
        s->table_dirty = true;
    }
    if (n > s->counts_cap) {
        if (s->d_counts) { HIP_TRY(hipStreamSynchronize(c->main())); HIP_TRY(hipFree(s->d_counts)); s->d_counts = nullptr; }
        int cap = n < 64 ? 64 : n * 2;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_counts), System::counts_bytes(cap)));
        HIP_TRY(hipMemsetAsync(s->d_counts, 0, System::counts_bytes(cap), c->main()));
        // (the stream is idle here: no kernel can still publish into the old host table)
        unsigned long long* old = s->h_counts;
        unsigned long long* fresh = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&fresh), sizeof(unsigned long long) * (size_t)cap, hipHostMallocMapped));
        for (int i = 0; i < cap; i++) fresh[i] = (i < s->h_counts_cap && old) ? old[i] : 0ull;
        if (old) HIP_TRY(hipHostFree(old));
        s->h_counts = fresh;
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&s->h_counts_dev), fresh, 0));
        s->h_counts_cap = cap;
        s->counts_cap = cap;
    }
    if (s->table_dirty && n > 0) {
        int32_t rc = upload_small(c, s->d_table, s->chunks.data(), sizeof(float*) * (size_t)n);
        if (rc != ILM_OK) return rc;
    }
    s->table_dirty = false;
    return ILM_OK;
}

int32_t validate_step(const System* s, const IlmStepDesc* d, int* first, int* count) {
    const int n = (int)s->chunks.size();
    if (d->OpCount < 0 || d->OpCount > ILM_MAX_OPS)
        return fail(ILM_ERR_TOO_MANY, "OpCount %d outside [0, %d]", d->OpCount, ILM_MAX_OPS);
    if (d->SpawnCount < 0 || d->SpawnCount > ILM_MAX_SPAWNS)
        return fail(ILM_ERR_TOO_MANY, "SpawnCount %d outside [0, %d]", d->SpawnCount, ILM_MAX_SPAWNS);
    if (d->UpdateMode < ILM_UPDATE_NONE || d->UpdateMode > ILM_UPDATE_ERASE)
        return fail(ILM_ERR_INVALID_ARGUMENT, "unknown UpdateMode %d", d->UpdateMode);
    for (int o = 0; o < d->OpCount; o++) {
        const IlmTransformOp& op = d->Ops[o];
        if (op.Type == ILM_OP_GRAVITY) {
            // Transforms.cs:348-349: "Maximum number of attractors per instance is 16"
            if (op.u.Gravity.AttractorCount > ILM_MAX_ATTRACTORS || op.u.Gravity.AttractorCount < 0)
                return fail(ILM_ERR_TOO_MANY, "Maximum number of attractors per instance is %d", ILM_MAX_ATTRACTORS);
        } else if (op.Type != ILM_OP_NOISE && op.Type != ILM_OP_FMA && op.Type != ILM_OP_MATRIX_MULTIPLY && op.Type != ILM_OP_SPATIAL_NOISE) {
            return fail(ILM_ERR_INVALID_ARGUMENT, "unknown transform type %d", op.Type);
        }
    }
    for (int k = 0; k < d->SpawnCount; k++) {
        const IlmSpawnRecord& r = d->Spawns[k];
        if (r.ChunkIndex < 0 || r.ChunkIndex >= n)
            return fail(ILM_ERR_OUT_OF_RANGE, "spawn target chunk %d outside [0, %d)", r.ChunkIndex, n);
        const float cs = r.Params.ChunkSizeAndIndices[0], first_i = r.Params.ChunkSizeAndIndices[1], last_i = r.Params.ChunkSizeAndIndices[2];
        if ((int)cs != s->engine->chunk_size)
            return fail(ILM_ERR_INVALID_ARGUMENT, "ChunkSizeAndIndices.x %g != engine chunk size %d", (double)cs, s->engine->chunk_size);
        if (first_i < 0 || last_i >= (float)s->engine->slots)
            return fail(ILM_ERR_OUT_OF_RANGE, "spawn range [%g, %g] outside the chunk", (double)first_i, (double)last_i);
        if (r.Kind == ILM_SPAWN_POSITION_BUFFER) {
            if (s->spawn_position_count[k] < 1 || r.Params.PositionConstantCount != (float)s->spawn_position_count[k])
                return fail(ILM_ERR_STATE, "spawn record %d: PositionConstantCount %g but %d positions bound (ilm_system_set_spawn_positions)",
                            k, (double)r.Params.PositionConstantCount, s->spawn_position_count[k]);
        } else if (r.Kind == ILM_SPAWN_FEEDBACK) {
            const System* src = from_handle<System>(r.Feedback.SourceSystem, kMagicSystem);
            if (!src) return fail(ILM_ERR_INVALID_HANDLE, "spawn record %d: feedback source is not a system handle", k);
            // "FIXME: Support using the same system as a feedback input?" -- the reference refuses it (SpecialSpawners.cs:347-349)
            if (src == s) return fail(ILM_ERR_INVALID_ARGUMENT, "spawn record %d: a system cannot feed back into itself", k);
            if (src->engine != s->engine) return fail(ILM_ERR_INVALID_ARGUMENT, "spawn record %d: feedback source belongs to another engine", k);
            if (r.Feedback.SourceChunkIndex < 0 || r.Feedback.SourceChunkIndex >= (int)src->chunks.size())
                return fail(ILM_ERR_OUT_OF_RANGE, "spawn record %d: source chunk %d outside [0, %d)", k, r.Feedback.SourceChunkIndex, (int)src->chunks.size());
            if (!(r.Feedback.InstanceMultiplier >= 1.0f))
                return fail(ILM_ERR_INVALID_ARGUMENT, "spawn record %d: InstanceMultiplier %g < 1", k, (double)r.Feedback.InstanceMultiplier);
        } else if (r.Kind == ILM_SPAWN_PATTERN) {
            if (s->pattern_levels[k] < 1)
                return fail(ILM_ERR_STATE, "spawn record %d: no pattern texture bound (ilm_system_set_spawn_pattern)", k);
            if (!(r.Pattern.StepWidthAndSizeScale[1] >= 1.0f))
                return fail(ILM_ERR_INVALID_ARGUMENT, "spawn record %d: ParticlesPerRow %g < 1", k, (double)r.Pattern.StepWidthAndSizeScale[1]);
        } else if (r.Kind != ILM_SPAWN_INLINE) {
            return fail(ILM_ERR_INVALID_ARGUMENT, "spawn record %d: unknown kind %d", k, r.Kind);
        }
        if (r.Kind != ILM_SPAWN_POSITION_BUFFER &&
            (r.Params.PositionConstantCount < 1.0f || r.Params.PositionConstantCount > (float)ILM_MAX_INLINE_POSITION_CONSTANTS))
            return fail(ILM_ERR_OUT_OF_RANGE, "PositionConstantCount %g outside [1, %d]", (double)r.Params.PositionConstantCount,
                        ILM_MAX_INLINE_POSITION_CONSTANTS);
    }
    if (d->UpdateMode == ILM_UPDATE_WITH_DISTANCE_FIELD && from_handle<Sdf>(s->sdf_handle, kMagicSdf) == nullptr)
        // ParticleSystem.cs:835-836
        return fail(ILM_ERR_STATE, "UpdateWithDistanceField requires a distance field (ilm_system_set_distance_field)");
    int f = d->FirstChunk, c = d->ChunkCount;
    if (c < 0) { f = 0; c = n; }
    if (f < 0 || f + c > n)
        return fail(ILM_ERR_OUT_OF_RANGE, "chunk range [%d, %d) outside [0, %d)", f, f + c, n);
    *first = f; *count = c;
    return ILM_OK;
}

// The coordinate -> texel map of one axis of Noise's table lookups (randomCustom with rate = texel, RandomCommon.fxh:27-30,
// Noise.fx:49-52), evaluated with the operations of particles.hip random_custom in the same order (this file is compiled with
// -ffp-contract=off, like that function's fenced body).
static float noise_axis_texel(float coordinate, float texel, float offset, int size) {
    const float u = ((coordinate * texel) + offset) * texel;
    return floorf(u * (float)size);
}

// Steps of one axis of the map over coordinates [0, last]: at most two (false otherwise, or when the texel values are too large
// for exact integer arithmetic).  run k covers [flips[k-1], flips[k]) and reads the WRAPped texel index texels[k].
static bool noise_axis_runs(float texel, float offset, int size, int last, int32_t flips[2], int32_t texels[3]) {
    flips[0] = flips[1] = INT32_MAX;
    texels[0] = texels[1] = texels[2] = 0;
    int begin = 0;
    for (int run = 0; run < 3; run++) {
        const float value = noise_axis_texel((float)begin, texel, offset, size);
        if (!(std::fabs(value) < 4194304.0f)) return false;      // also rejects NaN / infinity
        const long long v = (long long)value;
        texels[run] = (int32_t)(((v % size) + size) % size);    // WRAP addressing
        if (noise_axis_texel((float)last, texel, offset, size) == value) return true;
        if (run == 2) return false;                               // a third step
        // smallest coordinate in (begin, last] whose texel differs: the map is monotone, so bisect
        int lo = begin, hi = last;
        while (hi - lo > 1) {
            const int mid = (lo + hi) / 2;
            if (noise_axis_texel((float)mid, texel, offset, size) == value) lo = mid; else hi = mid;
        }
        flips[run] = hi;
        begin = hi;
    }
    return true;
}

static int run_of(const int32_t flips[2], int coordinate) { return (coordinate >= flips[0] ? 1 : 0) + (coordinate >= flips[1] ? 1 : 0); }

static float host_lerp(float a, float b, float t) { return a + (b - a) * t; }
// noise_shape of particles.hip (Noise.fx:53-60): sign(d) * max(|d|, minimum) * scale with HLSL's sign(0) = 0
static float host_noise_shape(float r, float offset, float minimum, float scale) {
    const float d = r + offset;
    const float m = std::fmax(std::fabs(d), minimum);
    return ((d == 0.0f) ? 0.0f * m : std::copysign(m, d)) * scale;
}

// StepDerived::NoiseFast for one Noise op (see internal.hpp).  Coordinates a chunk can ask for: x in [0, chunk_size + 1],
// y in [0, chunk_size] (the second sample pair sits at (x + 2, y + 1)).  Up to 3 classes per axis the delta tables sit in the
// NoiseFast block; up to 5 they take the place of the spawn records in `desc` (only when the launch has none).  false when the fast
// path does not apply: a wave would span several rows, the runs of the two sample sets combine into more classes, or no room.
static bool fill_noise_fast(StepDerived& dv, IlmStepDesc& desc, const IlmNoiseParams& p, int chunk_size, const std::vector<float4>& table, int rw, int rh) {
    StepDerived::NoiseFast& nf = dv.noise;
    if (chunk_size % 64 != 0 || chunk_size / 64 > 16 || table.empty()) return false;
    int32_t xf[2][2], yf[2][2], tx[2][3], ty[2][3];
    for (int s = 0; s < 2; s++) {
        const float* off = (s == 0) ? p.RandomnessOffset : p.NextRandomnessOffset;
        if (!noise_axis_runs(dv.inv_rw, off[0], rw, chunk_size + 1, xf[s], tx[s])) return false;
        if (!noise_axis_runs(dv.inv_rh, off[1], rh, chunk_size, yf[s], ty[s])) return false;
    }
    // classes along an axis = the intervals between the steps of either sample set (at most 4 steps)
    auto merge = [](const int32_t a[2], const int32_t b[2], int32_t out[4]) {
        int32_t all[4] = { a[0], a[1], b[0], b[1] };
        std::sort(all, all + 4);
        const int n = (int)(std::unique(all, all + 4) - all);
        int m = 0;
        for (int i = 0; i < n; i++)
            if (all[i] != INT32_MAX) out[m++] = all[i];
        const int steps = m;
        for (; m < 4; m++) out[m] = INT32_MAX;
        return steps;
    };
    auto class_of = [](const int32_t b[4], int coordinate) {
        int c = 0;
        for (int k = 0; k < 4; k++) c += (coordinate >= b[k]) ? 1 : 0;
        return c;
    };
    int32_t xb[4];
    const int xsteps = merge(xf[0], xf[1], xb), ysteps = merge(yf[0], yf[1], nf.yb);
    const bool small = (xsteps <= 2) && (ysteps <= 2);
    if (!small && desc.SpawnCount != 0) return false;        // the big tables need the spawn records' bytes
    nf.classes = small ? 3 : kNoiseBigClasses;
    for (int w = 0; w < 16; w++) {
        nf.wcode[w] = 0;
        if (w * 64 >= chunk_size) continue;
        const int x0 = w * 64;
        bool usable = true;
        for (int k = 0; k < 4; k++)
            if (xb[k] != INT32_MAX && ((xb[k] > x0 && xb[k] <= x0 + 63) || (xb[k] > x0 + 2 && xb[k] <= x0 + 65))) usable = false;
        nf.wcode[w] = (usable ? 64u : 0u) | (uint32_t)class_of(xb, x0) | ((uint32_t)class_of(xb, x0 + 2) << 3);
    }
    IlmFloat4* big = reinterpret_cast<IlmFloat4*>(&desc.Spawns[0]);
    const int n = nf.classes;
    // one representative coordinate per class: its first
    for (int yc = 0; yc < n; yc++)
        for (int xc = 0; xc < n; xc++) {
            const int x = (xc == 0) ? 0 : xb[xc - 1], y = (yc == 0) ? 0 : nf.yb[yc - 1];
            float4 pd = make_float4(0, 0, 0, 0), vd = pd;
            if (x != INT32_MAX && y != INT32_MAX) {
                const float4 a = table[(size_t)ty[0][run_of(yf[0], y)] * (size_t)rw + (size_t)tx[0][run_of(xf[0], x)]];
                const float4 b = table[(size_t)ty[1][run_of(yf[1], y)] * (size_t)rw + (size_t)tx[1][run_of(xf[1], x)]];
                const float f = p.FrequencyLerp;
                // class (yc, xc) read as the (x, y) pair gives positionDelta, read as the (x + 2, y + 1) pair velocityDelta
                pd = make_float4(host_noise_shape(host_lerp(a.x, b.x, f), p.PositionOffset.x, p.PositionMinimum.x, p.PositionScale.x),
                                 host_noise_shape(host_lerp(a.y, b.y, f), p.PositionOffset.y, p.PositionMinimum.y, p.PositionScale.y),
                                 host_noise_shape(host_lerp(a.z, b.z, f), p.PositionOffset.z, p.PositionMinimum.z, p.PositionScale.z),
                                 host_noise_shape(host_lerp(a.w, b.w, f), p.PositionOffset.w, p.PositionMinimum.w, p.PositionScale.w));
                vd = make_float4(host_noise_shape(host_lerp(a.x, b.x, f), p.VelocityOffset.x, p.VelocityMinimum.x, p.VelocityScale.x),
                                 host_noise_shape(host_lerp(a.y, b.y, f), p.VelocityOffset.y, p.VelocityMinimum.y, p.VelocityScale.y),
                                 host_noise_shape(host_lerp(a.z, b.z, f), p.VelocityOffset.z, p.VelocityMinimum.z, p.VelocityScale.z),
                                 host_noise_shape(host_lerp(a.w, b.w, f), p.VelocityOffset.w, p.VelocityMinimum.w, p.VelocityScale.w));
            }
            const IlmFloat4 pdv = { pd.x, pd.y, pd.z, pd.w }, vdv = { vd.x, vd.y, vd.z, vd.w };
            if (small) { nf.position[yc][xc] = pdv; nf.velocity[yc][xc] = vdv; }
            else { big[yc * n + xc] = pdv; big[n * n + yc * n + xc] = vdv; }
        }
    return true;
}

// Lower bound of an obstruction's distance function for the culling test of fields.hip.  p = rotateLocalPosition(world - centre, q)
// has |p| = |q|^2 * e (e = |world - centre|).  Box, cylinder (radius |size.xy|, half height size.z), spheroid and octagon are exact
// signed distances of a shape inside the ball of radius R around the centre, so f >= |p| - R everywhere (outside: the shape is no
// nearer than its bounding ball; inside: the boundary is at most R - |p| away).  sdEllipsoid_improvedV2 = k0 (k0 - 1) / k1 (or
// (k0 - 1) rmin inside) with k0 / k1 >= rmin and k0 >= |p| / rmax gives f >= (rmin / rmax) (|p| - rmax).  The margins (0.1 % on the
// scale and on |q|^2, 0.05 units + 0.01 % on the radius) dwarf the rounding of the float evaluation; anything irregular -- a
// non-unit quaternion, a non-positive or non-finite size -- switches the test off.
static void fill_cull_bound(FieldObstruction& r) {
    r.cull_inv_scale = 0.0f;
    r.cull_radius = INFINITY;
    r._pad2[0] = r._pad2[1] = 0.0f;
    const double qn = (double)r.qx * r.qx + (double)r.qy * r.qy + (double)r.qz * r.qz + (double)r.qw * r.qw;
    const double sx = r.sx, sy = r.sy, sz = r.sz;
    if (!(std::fabs(qn - 1.0) <= 1e-3) || !(sx > 0 && sy > 0 && sz > 0) || !std::isfinite(sx + sy + sz)) return;
    if (!std::isfinite((double)r.cx + r.cy + r.cz)) return;
    double scale = 1.0, radius;
    if (r.type == ILM_OBSTRUCTION_ELLIPSOID) {
        const double rmax = std::max(sx, std::max(sy, sz)), rmin = std::min(sx, std::min(sy, sz));
        scale = rmin / rmax;
        radius = rmax;
    } else if (r.type == ILM_OBSTRUCTION_OCTAGON) {
        const double k = 1.0823922003;      // circumradius / apothem of a regular octagon
        radius = std::sqrt(k * sx * k * sx + k * sy * k * sy + sz * sz);
    } else {
        radius = std::sqrt(sx * sx + sy * sy + sz * sz);
    }
    const double shrink = 1.0 - 1e-3;       // |q|^2 >= 1 - 1e-3
    r.cull_inv_scale = (float)(1.0 / (scale * (1.0 - 1e-3) * shrink) * (1.0 + 1e-6));
    r.cull_radius = (float)((radius * (1.0 + 1e-4) + 0.05) / shrink * (1.0 + 1e-6));
}

// (unsigned)fabsf(x) as the device converts (v_cvt_u32_f32 saturates: NaN and negatives give 0, 2^32 and above give 0xFFFFFFFF)
static uint32_t device_float_to_uint(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
// The uniform decisions of tForScaledBezier / evaluateBezier1 / evaluateBezier4 (Bezier.fxh:21-177), taken here with the comparisons
// the shader makes on RangeAndCount; layout in bezier.hpp.
static uint32_t bezier_code(const IlmFloat4& rc) {
    const float count = rc.z;
    uint32_t cls = 3u;
    if (count <= 1.5f) cls = 0u;
    else if (count <= 2.5f) cls = 1u;
    else if (count <= 3.5f) cls = 2u;
    const uint32_t mode = device_float_to_uint(std::fabs(rc.w));
    const uint32_t range = (mode > 511u) ? 2u : ((mode > 255u) ? 1u : 0u);
    const uint32_t neg = (rc.y < 0.0f) ? 1u : 0u;
    const uint32_t m = mode % 256u;
    const uint32_t shaping = (m == 1u) ? 1u : ((m == 2u) ? 2u : 0u);
    return cls | (range << 2) | (neg << 4) | (shaping << 5);
}

// (process-wide diagnostic switch; atomic because two host threads may each step a context of their own.  A CONTEXT itself is
// single-threaded: Ctx::main()'s stream hand-over state is not synchronised -- one thread per context at a time, include/illuminant_hip.h)
static std::atomic<int> g_step_streams{-1};       // -1: not decided yet (ILM_STEP_STREAMS), else 1 or 2
static int step_streams() {
    if (g_step_streams < 0) {
        const char* v = getenv("ILM_STEP_STREAMS");
        g_step_streams = (v && atoi(v) == 1) ? 1 : 2;
    }
    return g_step_streams;
}
int set_step_streams_impl(int n) {
    const int before = step_streams();
    g_step_streams = (n == 1) ? 1 : 2;
    return before;
}
constexpr int64_t kSplitMinUnits = 8192;      // half a million slots: below it the second launch costs more than the overlap returns

// The slice-0 cells of a UNORM16 field for the lean collision step (hlsl_math.hpp SdfView::cells0): (re)built on the context stream when
// the atlas has changed since the last build -- or before every use for an atlas whose device pointer was handed out.  16 us for the
// demo's 960 x 540 atlas; a field generated once costs it once.  Returns nullptr (the four-tap form) when there is no memory for them.
static const void* ensure_slice0_cells(Sdf* f, Ctx* c) {
    const size_t bytes = sizeof(uint2) * (size_t)f->width * (size_t)(f->height + 1);
    if (f->width <= 0 || f->height <= 0 || ((uint64_t)f->width * (uint64_t)(f->height + 1)) >= ((uint64_t)1 << 28)) return nullptr;     // 32-bit byte offsets
    if (f->cells0 && f->cells0_bytes == bytes && f->cells0_version == f->version && !f->escaped) return f->cells0;
    // (only a rebuild needs the context stream, joined with the second stepping stream: both halves of a split step see the new cells.
    // Taking it for every step would join the two streams every step -- 14 us per cfg2 step, measured)
    const hipStream_t stream = c->main();
    if (f->cells0_bytes != bytes) {
        if (f->cells0) { (void)hipStreamSynchronize(stream); (void)hipFree(f->cells0); f->cells0 = nullptr; f->cells0_bytes = 0; }
        if (hipMalloc(&f->cells0, bytes) != hipSuccess) { (void)hipGetLastError(); f->cells0 = nullptr; return nullptr; }
        f->cells0_bytes = bytes;
    }
    if (launch_build_slice0_cells(f->texels, f->width, f->height, f->cells0, stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    f->cells0_version = f->version;
    return f->cells0;
}

int32_t run_step(System* s, const IlmStepDesc* d) {
    int first = 0, count = 0;
    int32_t rc = validate_step(s, d, &first, &count);
    if (rc != ILM_OK) return rc;
    Engine* e = s->engine;
    Ctx* c = e->ctx;
    HIP_TRY(hipSetDevice(c->device));
    rc = refresh_table(s);
    if (rc != ILM_OK) return rc;
    if (count == 0) {
        // nothing to launch; a counting step over no chunks has produced its (empty) counts
        if (d->Flags & ILM_STEP_COUNT_LIVE) { s->counts_n = 0; s->counts_pending = true; s->counts_valid = true; }
        return ILM_OK;
    }
    const bool counting = (d->Flags & ILM_STEP_COUNT_LIVE) != 0;
    const int region = s->count_parity;

    StepLaunch a;
    memcpy(&a.desc, d, sizeof(IlmStepDesc));
    a.chunk_bases = s->d_table;
    a.stride = e->stride;
    a.span = e->span;
    a.chunk_size = e->chunk_size;
    a.slots = e->slots;
    a.first_chunk = first;
    a.chunk_count = count;
    a.op_mask = 0;
    for (int o = 0; o < d->OpCount; o++) a.op_mask |= 1u << d->Ops[o].Type;
    // 20 planes x stride x 4 B per chunk against the 256 MB Infinity Cache (MI355X_MICROARCH.md): above it every plane streams
    // through once per step and the non-temporal variant wins; below it the planes stay resident from one step to the next
    a.streaming = ((size_t)count * (size_t)e->stride * kComponents * sizeof(float) > ((size_t)256 << 20)) ? 1 : 0;
    if (const char* v = getenv("ILM_STEP_STREAMING")) a.streaming = atoi(v) != 0 ? 1 : 0;      // experiment / test switch, read per step
    for (int k = 0; k < d->SpawnCount; k++) {
        const IlmSpawnRecord& r = d->Spawns[k];
        if (r.Params.ChunkSizeAndIndices[2] >= r.Params.ChunkSizeAndIndices[1])
            s->used[(size_t)r.ChunkIndex] = std::max(s->used[(size_t)r.ChunkIndex], (int32_t)r.Params.ChunkSizeAndIndices[2] + 1);
    }
    a.rnd = e->rnd; a.rw = e->rw; a.rh = e->rh;
    a.rnd_lp = e->rnd_lp;
    for (int k = 0; k < ILM_MAX_SPAWNS; k++) {
        a.spawn_positions[k] = s->spawn_positions[k];
        a.spawn_position_count[k] = s->spawn_position_count[k];
        a.source_base[k] = nullptr;
        a.spawn_pattern[k] = s->spawn_pattern[k];
        a.pattern_w[k] = s->pattern_w[k]; a.pattern_h[k] = s->pattern_h[k]; a.pattern_levels[k] = s->pattern_levels[k];
        if (k < d->SpawnCount && d->Spawns[k].Kind == ILM_SPAWN_FEEDBACK)
            a.source_base[k] = from_handle<System>(d->Spawns[k].Feedback.SourceSystem, kMagicSystem)->chunks[(size_t)d->Spawns[k].Feedback.SourceChunkIndex];
    }
    a.ramp = s->ramp; a.ramp_w = s->ramp_w; a.ramp_h = s->ramp_h;
    {
        Sdf* field = from_handle<Sdf>(s->sdf_handle, kMagicSdf);
        a.sdf = make_sdf_view(field, &d->DistanceField);
        if (field && step_wants_slice0_cells(*d, field->format)) a.sdf.cells0 = ensure_slice0_cells(field, c);
    }
    a.live_counts = counting ? s->counts_region(region * 2) : nullptr;
    a.zero_counts = counting ? s->counts_region((region ^ 1) * 2) : nullptr;
    a.zero_n = counting ? 2 * (int32_t)s->counts_cap * kCountLines : 0;   // every line of both halves, so chunk-table growth after a shrink never meets stale counts
    if (counting && ++s->count_seq == 0u) s->count_seq = 1u;      // 0 is the table's initial content
    a.host_counts = counting ? s->h_counts_dev : nullptr;
    a.count_seq = s->count_seq;
    {   // StepDerived: same float operations, same order, as the device code they replace
        StepDerived& dv = a.derived;
        std::memset(&dv, 0, sizeof(dv));
        const float dt_ms = d->System.GlobalSettings.x;
        dv.dt_s = dt_ms / kVelocityConstantScale;
        dv.inv_rw = 1.0f / (float)e->rw;
        dv.inv_rh = 1.0f / (float)e->rh;
        dv.cs_shift = -1;
        for (int b = 0; b < 31; b++)
            if ((1 << b) == e->chunk_size) dv.cs_shift = b;
        dv.bezier_codes = bezier_code(d->Update.ColorFromLife.RangeAndCount) | (bezier_code(d->Update.ColorFromVelocity.RangeAndCount) << 8) |
                          (bezier_code(d->Update.SizeFromLife.RangeAndCount) << 16) | (bezier_code(d->Update.SizeFromVelocity.RangeAndCount) << 24);
        dv.update_bits = ((d->Update.LifeRampSettings.x != 0.0f) ? 1u : 0u) | ((d->System.AnimationRateAndRotationAndZToY.z == 0.0f) ? 2u : 0u) |
                         ((d->Update.LifeRampSettings.x < 0.0f) ? 4u : 0u);
        dv.noise.op = -1;
        for (int o = 0; o < d->OpCount && dv.noise.op < 0; o++)
            if (d->Ops[o].Type == ILM_OP_NOISE) {
                if (fill_noise_fast(dv, a.desc, d->Ops[o].u.Noise, e->chunk_size, e->h_rnd, e->rw, e->rh)) dv.noise.op = o;
                else break;     // only the first Noise op is considered
            }
        for (int o = 0; o < d->OpCount; o++) {
            const IlmTransformOp& op = d->Ops[o];
            if (op.Type == ILM_OP_GRAVITY) {
                dv.op[o].max_accel = op.u.Gravity.MaximumAcceleration * dt_ms / kVelocityConstantScale;
            } else if (op.Type == ILM_OP_NOISE || op.Type == ILM_OP_FMA) {
                const IlmAreaParams& ar = (op.Type == ILM_OP_NOISE) ? op.u.Noise.Area : op.u.FMA.Area;
                const float divisor = (op.Type == ILM_OP_NOISE) ? op.u.Noise.TimeDivisor : op.u.FMA.TimeDivisor;
                const int t = ar.AreaType < 0 ? -ar.AreaType : ar.AreaType;
                dv.op[o].area_none = (t < 1 || t > 5) ? 1 : 0;
                dv.op[o].t = ar.Strength * dt_ms / divisor;
                if (op.Type == ILM_OP_NOISE) {
                    // Noise is the one transform without a life check (Noise.fx:40): a dead slot goes through
                    // newLife = lerp(life, life + delta.w, t).  With PositionScale.w == 0 and every factor finite,
                    // delta.w is +-0 and the lerp returns life unchanged, so the slot stays dead and the update pass
                    // discards it whatever else Noise wrote: its arithmetic can be skipped.
                    const IlmNoiseParams& np = op.u.Noise;
                    const bool inert = (np.PositionScale.w == 0.0f) && std::isfinite(np.PositionOffset.w) && std::isfinite(np.PositionMinimum.w) &&
                                       std::isfinite(ar.Strength) && std::isfinite(dt_ms) && std::isfinite(divisor) && (divisor != 0.0f) &&
                                       std::isfinite(ar.AreaFalloff);
                    if (!inert) dv.noise_may_revive = 1;
                }
            } else if (op.Type == ILM_OP_SPATIAL_NOISE) {
                // same argument as Noise (no life check, Noise.fx:86): life' = lerp(life, life + (r.w + offset.w) * scale.w, t)
                const IlmNoiseParams& np = op.u.SpatialNoise.Noise;
                const bool inert = (np.PositionScale.w == 0.0f) && std::isfinite(np.PositionOffset.w) && std::isfinite(np.Area.Strength) &&
                                   std::isfinite(dt_ms) && std::isfinite(np.TimeDivisor) && (np.TimeDivisor != 0.0f) && std::isfinite(np.Area.AreaFalloff);
                if (!inert) dv.noise_may_revive = 1;
            }
        }
    }
    // Units past a chunk's high-water mark are skipped -- unless a Noise-type op can write dead slots in this launch
    // (no life check, Noise.fx:40 / :86: it moves even never-spawned slots when its result survives)
    a.partial_count = 0;
    {
        const bool noise_op = (a.op_mask & ((1u << ILM_OP_NOISE) | (1u << ILM_OP_SPATIAL_NOISE))) != 0u;
        const bool noise_touches_dead = noise_op && ((d->UpdateMode == ILM_UPDATE_NONE) || (a.derived.noise_may_revive != 0));
        // ... and such a launch may have brought never-written slots to life: from now on every slot of its chunks counts as used
        // (a later step without the Noise op must age, update and count them like the reference does)
        for (int ci = first; noise_touches_dead && ci < first + count; ci++) s->used[(size_t)ci] = e->slots;
        for (int ci = first; !noise_touches_dead && ci < first + count && a.partial_count < kMaxPartialChunks; ci++) {
            const int used_units = (s->used[(size_t)ci] + 63) / 64;
            if (used_units * 64 < e->slots) {
                a.partial_chunk[a.partial_count] = ci;
                a.partial_units[a.partial_count] = used_units;
                a.partial_count++;
            }
        }
    }
    // Two streams for a large step.  A launch of this size is a few wave generations long, and back-to-back launches on one stream
    // pay its ramp and its tail (the last waves run alone) plus the dispatch gap every time: ~7.7 us of the ~23 us a cfg2 step takes
    // (tools/two_stream_probe.py: two 8-chunk launches in a row 31.2 us, one 16-chunk launch 23.5 us).  Chunks never interact, so
    // the second half of the range goes to the context's second stream, where its launches overlap the first half's boundaries
    // (22.4 us).  Nothing joins the streams between steps; Ctx::main() does when any other entry point needs the result.  Feedback
    // spawners read other chunks, exported streams may carry readers this library cannot see: those steps stay on one stream.
    int mid = -1;
    if (step_streams() >= 2 && !c->exported && count >= 2 && (int64_t)count * (e->span / 64) >= kSplitMinUnits) {
        bool reads_other_chunks = false;
        for (int k = 0; k < d->SpawnCount; k++) reads_other_chunks = reads_other_chunks || (d->Spawns[k].Kind == ILM_SPAWN_FEEDBACK);
        if (!reads_other_chunks) {
            // balance the units that carry particles (a spawn-target chunk is mostly untouched)
            int64_t total = 0, left = 0;
            for (int ci = first; ci < first + count; ci++) total += (s->used[(size_t)ci] + 63) / 64 + 1;
            mid = first + 1;
            for (int ci = first; ci < first + count - 1; ci++) {
                left += (s->used[(size_t)ci] + 63) / 64 + 1;
                mid = ci + 1;
                if (2 * left >= total) break;
            }
        }
    }
    if (mid < 0) {
        HIP_TRY(launch_step(a, c->main()));
    } else {
        if (s->split_at != mid) { (void)c->main(); s->split_at = mid; }      // a chunk changes streams: join first
        if (c->main_pending) {
            HIP_TRY(hipEventRecord(c->ev_main, c->stream_));
            HIP_TRY(hipStreamWaitEvent(c->aux, c->ev_main, 0));
            c->main_pending = false;
        }
        StepLaunch b = a;
        a.chunk_count = mid - first;
        b.first_chunk = mid; b.chunk_count = first + count - mid;
        if (counting) {
            a.zero_n = b.zero_n = (int32_t)s->counts_cap * kCountLines;
            b.live_counts = s->counts_region(region * 2 + 1);
            b.zero_counts = s->counts_region((region ^ 1) * 2 + 1);
        }
        // the half that holds the spawn ranges first: its waves are the long ones
        bool spawn_in_b = false;
        for (int k = 0; k < d->SpawnCount; k++) spawn_in_b = spawn_in_b || (d->Spawns[k].ChunkIndex >= mid);
        if (spawn_in_b) {
            HIP_TRY(launch_step(b, c->aux));
            c->aux_pending = true;
            HIP_TRY(launch_step(a, c->stream_));
        } else {
            HIP_TRY(launch_step(a, c->stream_));
            HIP_TRY(launch_step(b, c->aux));
            c->aux_pending = true;
        }
    }
    if (d->UpdateMode == ILM_UPDATE_ERASE)
        for (int ci = first; ci < first + count; ci++) s->used[(size_t)ci] = 0;   // position, velocity and render planes are zero again
    if (d->Flags & ILM_STEP_COUNT_LIVE) {
        // the kernel publishes every chunk of its range itself; ilm_system_poll_counts / ilm_system_step_counts read the host table
        s->counts_n = (int)s->chunks.size();
        s->counts_first = first; s->counts_span = count;
        s->counts_pending = true;
        s->counts_valid = true;
        s->count_parity ^= 1;
    }
    return ILM_OK;
}

int32_t copy_counts(System* s, const unsigned long long* d_region64, uint32_t* out, int32_t capacity, int32_t saturate16) {
    const uint32_t* d_region = reinterpret_cast<const uint32_t*>(d_region64);
    Ctx* c = s->engine->ctx;
    const int n = (int)s->chunks.size();
    if (capacity < n)
        return fail(ILM_ERR_OUT_OF_RANGE, "capacity %d < chunk count %d", capacity, n);
    if (n == 0) return ILM_OK;
    // the counts come to the host through a pinned ring slot a copy kernel writes (no copy-engine latency in front of the synchronisation)
    const size_t bytes = sizeof(uint32_t) * (size_t)n * kCountStride;
    std::vector<uint32_t> fallback;
    const uint32_t* tmp = nullptr;
    if ((reinterpret_cast<uintptr_t>(d_region) & 15) == 0 && bytes <= ((size_t)2 << 20)) {
        unsigned char* block = nullptr;
        int slot = -1;
        const int32_t rc = upload_small_begin(c, bytes, reinterpret_cast<void**>(&block), &slot);
        if (rc != ILM_OK) return rc;
        void* dv = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dv, block, 0));
        const size_t words = bytes / 16;
        hipLaunchKernelGGL(copy_from_pinned_kernel, dim3((unsigned)std::max<size_t>(1, std::min<size_t>((words + 255) / 256, 1024))), dim3(256), 0, c->main(),
                           reinterpret_cast<const uint4*>(d_region), static_cast<uint4*>(dv), words, bytes);
        HIP_TRY(hipGetLastError());
        const int32_t rc2 = staged_small_done(c, slot);
        if (rc2 != ILM_OK) return rc2;
        HIP_TRY(hipStreamSynchronize(c->main()));
        tmp = reinterpret_cast<const uint32_t*>(block);
    } else {
        fallback.resize((size_t)n * kCountStride);
        HIP_TRY(hipMemcpyAsync(fallback.data(), d_region, bytes, hipMemcpyDeviceToHost, c->main()));
        HIP_TRY(hipStreamSynchronize(c->main()));
        tmp = fallback.data();
    }
    for (int i = 0; i < n; i++) {
        const uint32_t v = tmp[(size_t)i * kCountStride];
        out[i] = (saturate16 && v > ref::kLiveCountSaturation) ? ref::kLiveCountSaturation : v;   // 16-bit additive target, CountLiveParticles.fx:38 + ParticleEngine.cs:244-247
    }
    return ILM_OK;
}

}  // namespace

namespace ilm {
int32_t api_fail(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}
IlmHandle handle_register(const void* object, uint32_t magic) {
    HandleRegistry& r = handle_registry();
    std::lock_guard<std::mutex> lock(r.mutex);
    r.live[reinterpret_cast<uintptr_t>(object)] = magic;
    return static_cast<IlmHandle>(reinterpret_cast<uintptr_t>(object));
}
bool handle_is_live(IlmHandle h, uint32_t magic) {
    HandleRegistry& r = handle_registry();
    std::lock_guard<std::mutex> lock(r.mutex);
    const auto it = r.live.find(static_cast<uintptr_t>(h));
    return it != r.live.end() && it->second == magic;
}
void handle_retire(const void* object) { retire_handle(object); }
int ctx_child_count(IlmHandle h) {
    const Ctx* c = from_handle<Ctx>(h, kMagicCtx);
    return c ? c->children : -1;
}
int set_step_streams(int n) { return set_step_streams_impl(n); }
// ilm_group_gather_chunks (group.hip): one chunk of a system as the exchange sees it -- base of component 0, the stride between the
// component planes, the chunk's slot count and the context it lives on.  `written`: the caller is about to overwrite planes of the
// chunk, so every slot counts as used from now on (a later step must not skip units as never-written).
int32_t system_chunk_view(IlmHandle system, int chunk, bool written, float** out_base, int64_t* out_stride, int32_t* out_chunk_size, IlmHandle* out_ctx) {
    System* s = from_handle<System>(system, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (chunk < 0 || chunk >= (int)s->chunks.size()) return fail(ILM_ERR_OUT_OF_RANGE, "chunk %d outside [0, %d)", chunk, (int)s->chunks.size());
    if (out_base) *out_base = s->chunks[(size_t)chunk];
    if (out_stride) *out_stride = s->engine->stride;
    if (out_chunk_size) *out_chunk_size = s->engine->chunk_size;
    if (out_ctx) *out_ctx = to_handle(s->engine->ctx);
    if (written) s->used[(size_t)chunk] = s->engine->slots;
    return ILM_OK;
}
hipStream_t ctx_stream_joined(IlmHandle h) {
    Ctx* c = from_handle<Ctx>(h, kMagicCtx);
    return c ? c->main() : nullptr;
}
const TraceApi& trace_api() {
    static const TraceApi api = [] {
        TraceApi t;
        const char* e = getenv("ILM_TRACE");
        if (!e || atoi(e) == 0) return t;
        for (const char* name : { "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so" }) {
            void* lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!lib) continue;
            t.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
            t.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
            if (t.push && t.pop) { t.on = true; break; }
        }
        if (!t.on) fprintf(stderr, "libilluminant_hip: ILM_TRACE is set but no roctx library could be bound: no ranges\n");
        return t;
    }();
    return api;
}
// every lightmap of the process that holds a store-mode table, whatever its context (mirror_table_of: aliases from OTHER contexts)
static std::mutex g_armed_mutex;
static std::vector<Lightmap*> g_armed;
static std::atomic<int> g_armed_count{0};
static void forget_armed(const Lightmap* m) {
    std::lock_guard<std::mutex> lock(g_armed_mutex);
    g_armed.erase(std::remove(g_armed.begin(), g_armed.end(), m), g_armed.end());
    g_armed_count.store((int)g_armed.size(), std::memory_order_release);
}

// Store-mode exchange (ILM_GATHER_STORE): every light pass into this lightmap also stores its texels at the same offsets of `count`
// other buffers of the same size -- the other members' copies of a group's frame, addressable from this lightmap's device (peer access
// / the same device).  count == 0 ends it.  Synchronises the lightmap's stream (the table may be in use by a queued launch).
int32_t lightmap_set_mirrors(IlmHandle h, void* const* buffers, int count) {
    Lightmap* m = from_handle<Lightmap>(h, kMagicLightmap);
    if (!m) return fail(ILM_ERR_INVALID_HANDLE, "not a lightmap handle");
    if (count < 0 || count > 64 || (count > 0 && !buffers)) return fail(ILM_ERR_INVALID_ARGUMENT, "bad mirror list");
    HIP_TRY(hipSetDevice(m->ctx->device));
    HIP_TRY(hipStreamSynchronize(m->ctx->main()));
    std::vector<Lightmap*>& armed = m->ctx->mirrored;
    armed.erase(std::remove(armed.begin(), armed.end(), m), armed.end());
    forget_armed(m);
    if (count == 0) {
        if (m->d_mirrors) HIP_TRY(hipFree(m->d_mirrors));
        m->d_mirrors = nullptr; m->mirror_count = 0;
        return ILM_OK;
    }
    if (!m->d_mirrors) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_mirrors), sizeof(void*) * 64));
    HIP_TRY(hipMemcpy(m->d_mirrors, buffers, sizeof(void*) * (size_t)count, hipMemcpyHostToDevice));
    m->mirror_count = count;
    armed.push_back(m);
    {
        std::lock_guard<std::mutex> lock(g_armed_mutex);
        g_armed.push_back(m);
        g_armed_count.store((int)g_armed.size(), std::memory_order_release);
    }
    return ILM_OK;
}

// The store-mode table a light pass into `m` uses.  The table belongs to the BUFFER, not to the handle: a lightmap object that aliases an
// armed lightmap's texels on the same context (ilm_lightmap_create with external_device_ptr = ilm_lightmap_device_ptr of a group
// lightmap's member: what a host that wraps the group's buffer in its own renderer does, the host mirror among them) renders the same
// frame on the same stream, and a strip rendered through it that stayed at home would leave every other member with a hole.
// `*foreign` = true: `m` aliases the texels of a lightmap ANOTHER context holds armed (a sibling's renderer around a group member's
// buffer, say).  Its passes run on a stream the group's fence does not cover and would not be mirrored: the callers refuse.
static const Lightmap* mirror_table_of(const Lightmap* m, bool* foreign) {
    *foreign = false;
    if (m->d_mirrors && m->mirror_count > 0) return m;
    if (!m->external || g_armed_count.load(std::memory_order_acquire) == 0) return nullptr;
    for (const Lightmap* e : m->ctx->mirrored)
        // (a member's object is slot_rows * world rows tall, the frame and its alias may be shorter: same base, same pitch)
        if (e->texels == m->texels && e->width == m->width && m->height <= e->height && e->format == m->format && e->d_mirrors && e->mirror_count > 0) return e;
    std::lock_guard<std::mutex> lock(g_armed_mutex);
    for (const Lightmap* e : g_armed)
        if (e->texels == m->texels && e->ctx != m->ctx) *foreign = true;
    return nullptr;
}
// (both light-pass entry points)
static int32_t set_launch_mirrors(const Lightmap* m, LightLaunch* a) {
    bool foreign = false;
    const Lightmap* t = mirror_table_of(m, &foreign);
    if (foreign)
        return fail(ILM_ERR_STATE, "the lightmap aliases the buffer of a group lightmap member that ANOTHER context holds in store mode: a pass from here "
                                   "would not be mirrored into the other members' frames (render through the member's context, or disarm the mode)");
    a->mirrors = t ? t->d_mirrors : nullptr; a->mirror_count = t ? t->mirror_count : 0;
    return ILM_OK;
}
}  // namespace ilm

extern "C" {

int32_t ilm_abi_version(void) { return ILM_ABI_VERSION; }

int32_t ilm_debug_reference_constant(const char* key, double* out_value) {
    if (!key || !out_value) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    for (int i = 0; i < ref::kTableSize; i++)
        if (std::strcmp(ref::kTable[i].key, key) == 0) { *out_value = ref::kTable[i].value; return ILM_OK; }
    return fail(ILM_ERR_OUT_OF_RANGE, "the kernels take no constant named '%s' from the reference", key);
}
int32_t ilm_debug_reference_constant_count(void) { return ref::kTableSize; }
const char* ilm_debug_reference_constant_key(int32_t index) { return (index >= 0 && index < ref::kTableSize) ? ref::kTable[index].key : ""; }
const char* ilm_last_error(void) { return g_last_error; }

int32_t ilm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// ---- context ------------------------------------------------------------------------------------

int32_t ilm_ctx_create(int32_t device_id, IlmHandle* out_ctx) {
    if (!out_ctx) return fail(ILM_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
    *out_ctx = 0;
    const int n = ilm_device_count();
    if (n <= 0)
        return fail(ILM_ERR_NO_DEVICE, "no HIP device visible: libilluminant_hip has no CPU fallback");
    if (device_id < 0 || device_id >= n)
        return fail(ILM_ERR_OUT_OF_RANGE, "device %d outside [0, %d)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    Ctx* c = new (std::nothrow) Ctx();
    if (!c) return fail(ILM_ERR_INVALID_ARGUMENT, "out of host memory");
    c->device = device_id;
    const IlmHandle h = to_handle(c);
    HIP_TRY_OR_DESTROY(hipStreamCreateWithFlags(&c->stream_, hipStreamNonBlocking), ilm_ctx_destroy(h));
    HIP_TRY_OR_DESTROY(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking), ilm_ctx_destroy(h));
    HIP_TRY_OR_DESTROY(hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming), ilm_ctx_destroy(h));
    HIP_TRY_OR_DESTROY(hipEventCreateWithFlags(&c->ev_aux, hipEventDisableTiming), ilm_ctx_destroy(h));
    HIP_TRY_OR_DESTROY(hipEventCreate(&c->t0), ilm_ctx_destroy(h));
    HIP_TRY_OR_DESTROY(hipEventCreate(&c->t1), ilm_ctx_destroy(h));
    HIP_TRY_OR_DESTROY(hipMalloc(reinterpret_cast<void**>(&c->d_stats), 3 * sizeof(unsigned long long)), ilm_ctx_destroy(h));
    *out_ctx = h;
    return ILM_OK;
}

int32_t ilm_ctx_create_sibling(IlmHandle hctx, IlmHandle* out_ctx) {
    if (!out_ctx) return fail(ILM_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
    *out_ctx = 0;
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    IlmHandle h = 0;
    const int32_t rc = ilm_ctx_create(c->device, &h);
    if (rc != ILM_OK) return rc;
    // (a process-wide serial, not the first context's address: the heap may hand a destroyed context's address to an unrelated one, whose
    // siblings would then have joined the survivors of the old family -- ADVICE r05)
    static std::atomic<uintptr_t> next_family{1};
    if (c->family == 0) c->family = next_family.fetch_add(1);
    from_handle<Ctx>(h, kMagicCtx)->family = c->family;
    *out_ctx = h;
    return ILM_OK;
}

int32_t ilm_ctx_destroy(IlmHandle h) {
    Ctx* c = from_handle<Ctx>(h, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    if (c->children > 0)
        return fail(ILM_ERR_STATE, "%d object(s) of this context are still alive: destroy engines, fields, G-buffers and lightmaps first", c->children);
    (void)hipSetDevice(c->device);
    if (c->aux) (void)hipStreamSynchronize(c->aux);
    if (c->stream_) (void)hipStreamSynchronize(c->stream_);
    if (c->staging) (void)hipFree(c->staging);
    for (int i = 0; i < Ctx::kRing; i++) {
        if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
        if (c->pinned_ev[i]) (void)hipEventDestroy(c->pinned_ev[i]);
    }
    if (c->d_lights) (void)hipFree(c->d_lights);
    if (c->d_recs) (void)hipFree(c->d_recs);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->d_light_partials) (void)hipFree(c->d_light_partials);
    if (c->d_light_tickets) (void)hipFree(c->d_light_tickets);
    if (c->d_group_order) (void)hipFree(c->d_group_order);
    if (c->d_field_params) (void)hipFree(c->d_field_params);
    if (c->d_pl_recs) (void)hipFree(c->d_pl_recs);
    if (c->d_pl_count) (void)hipFree(c->d_pl_count);
    if (c->d_pl_blocks) (void)hipFree(c->d_pl_blocks);
    if (c->d_pl_quads) (void)hipFree(c->d_pl_quads);
    free_raster_scratch(c->raster);
    if (c->d_light_ramp) (void)hipFree(c->d_light_ramp);
    if (c->d_raster_quads) (void)hipFree(c->d_raster_quads);
    if (c->d_probe_pairs) (void)hipFree(c->d_probe_pairs);
    if (c->d_rb) (void)hipFree(c->d_rb);
    if (c->d_rb_count) (void)hipFree(c->d_rb_count);
    if (c->d_rb_blocks) (void)hipFree(c->d_rb_blocks);
    if (c->d_rb_elems) (void)hipFree(c->d_rb_elems);
    if (c->h_rb) (void)hipHostFree(c->h_rb);
    // (a context whose creation failed half-way has null members here)
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->ev_main) (void)hipEventDestroy(c->ev_main);
    if (c->ev_aux) (void)hipEventDestroy(c->ev_aux);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->stream_) (void)hipStreamDestroy(c->stream_);
    retire_handle(c);
    delete c;
    return ILM_OK;
}

int32_t ilm_ctx_sync(IlmHandle h) {
    Ctx* c = from_handle<Ctx>(h, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    HIP_TRY(hipStreamSynchronize(c->main()));
    return ILM_OK;
}

int32_t ilm_ctx_stream(IlmHandle h, void** out_stream) {
    Ctx* c = from_handle<Ctx>(h, kMagicCtx);
    if (!c || !out_stream) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    c->exported = true;      // the caller may queue readers of the particle planes on it: from now on every step stays on this one stream
    *out_stream = reinterpret_cast<void*>(c->main());
    return ILM_OK;
}

int32_t ilm_timer_start(IlmHandle h) {
    Ctx* c = from_handle<Ctx>(h, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    HIP_TRY(hipEventRecord(c->t0, c->main()));
    return ILM_OK;
}

int32_t ilm_timer_stop(IlmHandle h, float* out_ms) {
    Ctx* c = from_handle<Ctx>(h, kMagicCtx);
    if (!c || !out_ms) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    HIP_TRY(hipEventRecord(c->t1, c->main()));
    HIP_TRY(hipEventSynchronize(c->t1));
    HIP_TRY(hipEventElapsedTime(out_ms, c->t0, c->t1));
    return ILM_OK;
}

// ---- engine / system ------------------------------------------------------------------------------

int32_t ilm_engine_create(IlmHandle hctx, int32_t chunk_size, const IlmFloat4* randomness, int32_t rw, int32_t rh, IlmHandle* out) {
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    if (!out || !randomness) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = 0;
    if (chunk_size < 4 || chunk_size > 4096)
        return fail(ILM_ERR_OUT_OF_RANGE, "chunk_size %d outside [4, 4096]", chunk_size);
    if (rw <= 0 || rh <= 0) return fail(ILM_ERR_INVALID_ARGUMENT, "bad randomness table size");
    HIP_TRY(hipSetDevice(c->device));
    Engine* e = new (std::nothrow) Engine();
    if (!e) return fail(ILM_ERR_INVALID_ARGUMENT, "out of host memory");
    e->ctx = c;
    c->children++;
    e->chunk_size = chunk_size;
    e->slots = chunk_size * chunk_size;
    e->span = (int32_t)(((int64_t)e->slots + kSlotsPerBlock - 1) / kSlotsPerBlock * kSlotsPerBlock);
    // Component planes a power of two apart land on the same memory channels: every wave of the step touches the same 256 bytes of
    // 20 planes, and a bare copy of cfg2's planes runs 16 % faster once consecutive planes are offset by a kilobyte
    // (tools/ubench/stream <slots> 0 <pad>: 1 M slots 18.9 -> 15.8 us).
    static const int pad = [] { const char* v = getenv("ILM_PLANE_PAD"); return v ? atoi(v) : kPlanePad; }();
    e->stride = (int64_t)e->span + (pad / 64) * 64;
    e->rw = rw; e->rh = rh;
    const size_t bytes = sizeof(float4) * (size_t)rw * (size_t)rh;
    const IlmHandle h = to_handle(e);
    HIP_TRY_OR_DESTROY(hipMalloc(reinterpret_cast<void**>(&e->rnd), bytes), ilm_engine_destroy(h));
    HIP_TRY_OR_DESTROY(hipMemcpy(e->rnd, randomness, bytes, hipMemcpyHostToDevice), ilm_engine_destroy(h));
    e->h_rnd.assign(reinterpret_cast<const float4*>(randomness), reinterpret_cast<const float4*>(randomness) + (size_t)rw * (size_t)rh);
    {   // new Rgba64(Vector4) per texel (ParticleEngine.cs:536-538): round-half-even of clamp(v, 0, 1) * 65535 per channel
        std::vector<uint16_t> lp((size_t)rw * (size_t)rh * 4);
        const float* f = reinterpret_cast<const float*>(randomness);
        for (size_t i = 0; i < lp.size(); i++)
            lp[i] = (uint16_t)nearbyintf(fminf(fmaxf(f[i], 0.0f), 1.0f) * 65535.0f);
        HIP_TRY_OR_DESTROY(hipMalloc(reinterpret_cast<void**>(&e->rnd_lp), sizeof(uint2) * (size_t)rw * (size_t)rh), ilm_engine_destroy(h));
        HIP_TRY_OR_DESTROY(hipMemcpy(e->rnd_lp, lp.data(), sizeof(uint2) * (size_t)rw * (size_t)rh, hipMemcpyHostToDevice), ilm_engine_destroy(h));
    }
    *out = h;
    return ILM_OK;
}

// a zero-filled chunk of the engine's pool (zeroing is stream-ordered on the context stream, as its first use will be)
static int32_t acquire_chunk(Engine* e, float** out) {
    if (e->spare.empty()) {
        const size_t bytes = e->chunk_bytes();
        const int k = (int)std::min<size_t>(8, std::max<size_t>(1, ((size_t)64 << 20) / bytes));
        float* base = nullptr;
        hipError_t err = hipMalloc(reinterpret_cast<void**>(&base), bytes * (size_t)k);
        int got = k;
        if (err != hipSuccess && k > 1) { (void)hipGetLastError(); got = 1; err = hipMalloc(reinterpret_cast<void**>(&base), bytes); }
        if (err != hipSuccess) return fail((int32_t)err, "a particle chunk of %zu bytes: %s", bytes, hipGetErrorString(err));
        HIP_TRY(hipMemsetAsync(base, 0, bytes * (size_t)got, e->ctx->main()));
        e->slabs.push_back(Engine::Slab{ base, got });
        for (int i = got - 1; i >= 0; i--) e->spare.push_back(base + (size_t)i * (bytes / sizeof(float)));
    }
    *out = e->spare.back();
    e->spare.pop_back();
    return ILM_OK;
}
// A chunk nobody reads any more (the caller has drained the context stream) goes back, zeroed for its next owner.  Beyond kSpareChunks
// spares the pool gives memory back, as the reference discards buffers beyond SpareBufferCount (ParticleEngine.cs:402-419): a slab ALL
// of whose chunks are spare is freed, so a destroyed 64-chunk system of 1024^2 (5 GB) does not stay pinned behind a live engine
// (r05 only ever freed one-chunk slabs).
static void release_chunk(Engine* e, float* chunk) {
    (void)hipMemsetAsync(chunk, 0, e->chunk_bytes(), e->ctx->main());
    e->spare.push_back(chunk);
    const size_t chunk_floats = e->chunk_bytes() / sizeof(float);
    while ((int)e->spare.size() > Engine::kSpareChunks) {
        size_t victim = e->slabs.size();
        for (size_t i = 0; i < e->slabs.size() && victim == e->slabs.size(); i++) {
            const float* lo = e->slabs[i].base; const float* hi = lo + (size_t)e->slabs[i].chunks * chunk_floats;
            int spare_here = 0;
            for (const float* p : e->spare) spare_here += (p >= lo && p < hi) ? 1 : 0;
            if (spare_here == e->slabs[i].chunks) victim = i;
        }
        if (victim == e->slabs.size()) break;               // every slab still has a chunk in use: nothing can go yet
        const float* lo = e->slabs[victim].base; const float* hi = lo + (size_t)e->slabs[victim].chunks * chunk_floats;
        e->spare.erase(std::remove_if(e->spare.begin(), e->spare.end(), [&](const float* p) { return p >= lo && p < hi; }), e->spare.end());
        (void)hipStreamSynchronize(e->ctx->main());        // (the zero-fills queued on its chunks)
        (void)hipFree(e->slabs[victim].base);
        e->slabs.erase(e->slabs.begin() + (long)victim);
    }
}

int32_t ilm_engine_destroy(IlmHandle h) {
    Engine* e = from_handle<Engine>(h, kMagicEngine);
    if (!e) return fail(ILM_ERR_INVALID_HANDLE, "not an engine handle");
    if (e->children > 0) return fail(ILM_ERR_STATE, "%d system(s) of this engine are still alive", e->children);
    e->ctx->children--;
    (void)hipSetDevice(e->ctx->device);
    (void)hipStreamSynchronize(e->ctx->main());
    if (e->rnd) (void)hipFree(e->rnd);
    if (e->rnd_lp) (void)hipFree(e->rnd_lp);
    for (const Engine::Slab& slab : e->slabs) (void)hipFree(slab.base);
    retire_handle(e);
    delete e;
    return ILM_OK;
}

int32_t ilm_system_create(IlmHandle hengine, IlmHandle* out) {
    Engine* e = from_handle<Engine>(hengine, kMagicEngine);
    if (!e) return fail(ILM_ERR_INVALID_HANDLE, "not an engine handle");
    if (!out) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    System* s = new (std::nothrow) System();
    if (!s) return fail(ILM_ERR_INVALID_ARGUMENT, "out of host memory");
    s->engine = e;
    e->children++;
    *out = to_handle(s);
    return ILM_OK;
}

int32_t ilm_system_destroy(IlmHandle h) {
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    Ctx* c = s->engine->ctx;
    s->engine->children--;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->main());
    for (float* p : s->chunks) release_chunk(s->engine, p);
    if (s->d_table) (void)hipFree(s->d_table);
    if (s->d_counts) (void)hipFree(s->d_counts);
    if (s->ramp) (void)hipFree(s->ramp);
    if (s->d_slots) (void)hipFree(s->d_slots);
    if (s->d_slot_count) (void)hipFree(s->d_slot_count);
    for (int k = 0; k < ILM_MAX_SPAWNS; k++)
        if (s->spawn_positions[k]) (void)hipFree(s->spawn_positions[k]);
    for (int k = 0; k < ILM_MAX_SPAWNS; k++)
        if (s->spawn_pattern[k]) (void)hipFree(s->spawn_pattern[k]);
    if (s->bitmap) (void)hipFree(s->bitmap);
    if (s->h_counts) (void)hipHostFree(s->h_counts);
    retire_handle(s);
    delete s;
    return ILM_OK;
}

int32_t ilm_system_add_chunk(IlmHandle h, int32_t* out_index) {
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    Engine* e = s->engine;
    HIP_TRY(hipSetDevice(e->ctx->device));
    float* base = nullptr;
    { const int32_t rc = acquire_chunk(e, &base); if (rc != ILM_OK) return rc; }
    s->chunks.push_back(base);
    s->used.push_back(0);
    s->table_dirty = true;
    if (out_index) *out_index = (int32_t)s->chunks.size() - 1;
    return ILM_OK;
}

int32_t ilm_system_remove_chunk(IlmHandle h, int32_t index) {
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (index < 0 || index >= (int)s->chunks.size())
        return fail(ILM_ERR_OUT_OF_RANGE, "chunk %d outside [0, %d)", index, (int)s->chunks.size());
    Ctx* c = s->engine->ctx;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->main()));
    release_chunk(s->engine, s->chunks[(size_t)index]);
    s->chunks.erase(s->chunks.begin() + index);
    s->used.erase(s->used.begin() + index);
    s->table_dirty = true;
    return ILM_OK;
}

int32_t ilm_system_chunk_count(IlmHandle h, int32_t* out_count) {
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s || !out_count) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    *out_count = (int32_t)s->chunks.size();
    return ILM_OK;
}

static int32_t check_plane_range(System* s, int32_t chunk, int32_t plane, int32_t first_slot, int32_t count) {
    if (chunk < 0 || chunk >= (int)s->chunks.size())
        return fail(ILM_ERR_OUT_OF_RANGE, "chunk %d outside [0, %d)", chunk, (int)s->chunks.size());
    if (plane < ILM_PLANE_POSITION || plane > ILM_PLANE_RENDER_DATA)
        return fail(ILM_ERR_INVALID_ARGUMENT, "unknown plane %d", plane);
    if (first_slot < 0 || count < 0 || first_slot + count > s->engine->slots)
        return fail(ILM_ERR_OUT_OF_RANGE, "slot range [%d, %d) outside [0, %d)", first_slot, first_slot + count, s->engine->slots);
    return ILM_OK;
}

int32_t ilm_chunk_upload(IlmHandle h, int32_t chunk, int32_t plane, const IlmFloat4* src, int32_t first_slot, int32_t count) {
    ILM_TRACE_RANGE("ilm_chunk_upload");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!src) return fail(ILM_ERR_INVALID_ARGUMENT, "src is NULL");
    int32_t rc = check_plane_range(s, chunk, plane, first_slot, count);
    if (rc != ILM_OK || count == 0) return rc;
    Engine* e = s->engine; Ctx* c = e->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = sizeof(float4) * (size_t)count;
    rc = ensure_staging(c, bytes);
    if (rc != ILM_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c->staging, src, bytes, hipMemcpyHostToDevice, c->main()));
    s->used[(size_t)chunk] = std::max(s->used[(size_t)chunk], first_slot + count);
    HIP_TRY(launch_aos_to_soa(reinterpret_cast<const float4*>(c->staging), s->chunks[(size_t)chunk] + (int64_t)plane * 4 * e->stride,
                              e->stride, first_slot, count, c->main()));
    HIP_TRY(hipStreamSynchronize(c->main()));   // staging is reused by the next call
    return ILM_OK;
}

int32_t ilm_chunk_download(IlmHandle h, int32_t chunk, int32_t plane, IlmFloat4* dst, int32_t first_slot, int32_t count) {
    ILM_TRACE_RANGE("ilm_chunk_download");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!dst) return fail(ILM_ERR_INVALID_ARGUMENT, "dst is NULL");
    int32_t rc = check_plane_range(s, chunk, plane, first_slot, count);
    if (rc != ILM_OK || count == 0) return rc;
    Engine* e = s->engine; Ctx* c = e->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = sizeof(float4) * (size_t)count;
    rc = ensure_staging(c, bytes);
    if (rc != ILM_OK) return rc;
    HIP_TRY(launch_soa_to_aos(s->chunks[(size_t)chunk] + (int64_t)plane * 4 * e->stride, e->stride,
                              reinterpret_cast<float4*>(c->staging), first_slot, count, c->main()));
    HIP_TRY(hipMemcpyAsync(dst, c->staging, bytes, hipMemcpyDeviceToHost, c->main()));
    HIP_TRY(hipStreamSynchronize(c->main()));
    return ILM_OK;
}

int32_t ilm_chunk_device_ptr(IlmHandle h, int32_t chunk, int32_t component, void** out_ptr, int64_t* out_stride) {
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (chunk < 0 || chunk >= (int)s->chunks.size() || component < 0 || component >= kComponents)
        return fail(ILM_ERR_OUT_OF_RANGE, "chunk/component out of range");
    if (out_ptr) *out_ptr = s->chunks[(size_t)chunk] + (int64_t)component * s->engine->stride;
    s->used[(size_t)chunk] = s->engine->slots;     // the caller may write through the pointer
    if (out_stride) *out_stride = s->engine->stride;
    return ILM_OK;
}

int32_t ilm_system_set_distance_field(IlmHandle h, IlmHandle hsdf) {
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (hsdf == 0) { s->sdf_handle = 0; return ILM_OK; }
    Sdf* f = from_handle<Sdf>(hsdf, kMagicSdf);
    if (!f) return fail(ILM_ERR_INVALID_HANDLE, "not a distance field handle");
    if (f->ctx != s->engine->ctx) return fail(ILM_ERR_INVALID_ARGUMENT, "distance field belongs to another context");
    s->sdf_handle = hsdf;
    return ILM_OK;
}

int32_t ilm_system_set_life_ramp(IlmHandle h, const IlmFloat4* texels, int32_t width, int32_t height) {
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    Ctx* c = s->engine->ctx;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->main()));
    if (s->ramp) { HIP_TRY(hipFree(s->ramp)); s->ramp = nullptr; s->ramp_w = s->ramp_h = 0; }
    if (!texels || width <= 0 || height <= 0) return ILM_OK;
    const size_t bytes = sizeof(float4) * (size_t)width * (size_t)height;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->ramp), bytes));
    HIP_TRY(hipMemcpy(s->ramp, texels, bytes, hipMemcpyHostToDevice));
    s->ramp_w = width; s->ramp_h = height;
    return ILM_OK;
}

int32_t ilm_system_set_spawn_positions(IlmHandle h, int32_t slot, const IlmFloat4* positions, int32_t count) {
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (slot < 0 || slot >= ILM_MAX_SPAWNS) return fail(ILM_ERR_OUT_OF_RANGE, "spawn slot %d outside [0, %d)", slot, ILM_MAX_SPAWNS);
    if (count < 0 || (count > 0 && !positions)) return fail(ILM_ERR_INVALID_ARGUMENT, "bad position list");
    if (count > (1 << 20)) return fail(ILM_ERR_TOO_MANY, "at most %d positions", 1 << 20);
    Ctx* c = s->engine->ctx;
    HIP_TRY(hipSetDevice(c->device));
    if (count > s->spawn_position_cap[slot]) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (s->spawn_positions[slot]) HIP_TRY(hipFree(s->spawn_positions[slot]));
        s->spawn_positions[slot] = nullptr; s->spawn_position_cap[slot] = 0;
        const int cap = (count + 127) / 128 * 128;     // EnsurePositionBufferExists, ParticleSpawner.cs:307
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->spawn_positions[slot]), sizeof(float4) * (size_t)cap));
        s->spawn_position_cap[slot] = cap;
    }
    if (count > 0) {
        int32_t rc = upload_small(c, s->spawn_positions[slot], positions, sizeof(float4) * (size_t)count);
        if (rc != ILM_OK) return rc;
    }
    s->spawn_position_count[slot] = count;
    return ILM_OK;
}

int32_t ilm_system_set_spawn_pattern(IlmHandle h, int32_t slot, const IlmFloat4* texels, int32_t width, int32_t height, int32_t levels) {
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (slot < 0 || slot >= ILM_MAX_SPAWNS) return fail(ILM_ERR_OUT_OF_RANGE, "spawn slot %d outside [0, %d)", slot, ILM_MAX_SPAWNS);
    if (levels < 0 || levels > 16) return fail(ILM_ERR_OUT_OF_RANGE, "%d mip levels outside [0, 16]", levels);
    if (levels > 0 && (!texels || width < 1 || height < 1 || width > 16384 || height > 16384))
        return fail(ILM_ERR_INVALID_ARGUMENT, "bad pattern texture (%d x %d)", width, height);
    Ctx* c = s->engine->ctx;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->main()));     // an earlier step may still read the old texture
    if (s->spawn_pattern[slot]) HIP_TRY(hipFree(s->spawn_pattern[slot]));
    s->spawn_pattern[slot] = nullptr;
    s->pattern_w[slot] = s->pattern_h[slot] = s->pattern_levels[slot] = 0;
    if (levels == 0) return ILM_OK;
    size_t total = 0;
    for (int l = 0, lw = width, lh = height; l < levels; l++) {
        total += (size_t)lw * (size_t)lh;
        lw = std::max(1, lw >> 1); lh = std::max(1, lh >> 1);
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->spawn_pattern[slot]), sizeof(float4) * total));
    HIP_TRY(hipMemcpy(s->spawn_pattern[slot], texels, sizeof(float4) * total, hipMemcpyHostToDevice));
    s->pattern_w[slot] = width; s->pattern_h[slot] = height; s->pattern_levels[slot] = levels;
    return ILM_OK;
}

int32_t ilm_system_step(IlmHandle h, const IlmStepDesc* desc) {
    ILM_TRACE_RANGE("ilm_system_step");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!desc) return fail(ILM_ERR_INVALID_ARGUMENT, "desc is NULL");
    return run_step(s, desc);
}

static void init_single_pass(IlmStepDesc* d, int32_t chunk_index, const IlmParticleSystemUniforms* sys) {
    memset(d, 0, sizeof(*d));
    if (chunk_index < 0) { d->FirstChunk = 0; d->ChunkCount = -1; }
    else { d->FirstChunk = chunk_index; d->ChunkCount = 1; }
    if (sys) d->System = *sys;
}

int32_t ilm_spawn(IlmHandle h, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmSpawnParams* p) {
    ILM_TRACE_RANGE("ilm_spawn");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!p || chunk_index < 0) return fail(ILM_ERR_INVALID_ARGUMENT, "spawn needs parameters and a target chunk");
    IlmStepDesc d;
    init_single_pass(&d, chunk_index, sys);
    d.SpawnCount = 1;
    d.Spawns[0].ChunkIndex = chunk_index;
    d.Spawns[0].Params = *p;
    return run_step(s, &d);
}

int32_t ilm_gravity(IlmHandle h, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmGravityParams* p) {
    ILM_TRACE_RANGE("ilm_gravity");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!p || !sys) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    IlmStepDesc d;
    init_single_pass(&d, chunk_index, sys);
    d.OpCount = 1; d.Ops[0].Type = ILM_OP_GRAVITY; d.Ops[0].u.Gravity = *p;
    return run_step(s, &d);
}

int32_t ilm_noise(IlmHandle h, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmNoiseParams* p) {
    ILM_TRACE_RANGE("ilm_noise");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!p || !sys) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    IlmStepDesc d;
    init_single_pass(&d, chunk_index, sys);
    d.OpCount = 1; d.Ops[0].Type = ILM_OP_NOISE; d.Ops[0].u.Noise = *p;
    return run_step(s, &d);
}

int32_t ilm_fma(IlmHandle h, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmFMAParams* p) {
    ILM_TRACE_RANGE("ilm_fma");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!p || !sys) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    IlmStepDesc d;
    init_single_pass(&d, chunk_index, sys);
    d.OpCount = 1; d.Ops[0].Type = ILM_OP_FMA; d.Ops[0].u.FMA = *p;
    return run_step(s, &d);
}

int32_t ilm_matrix_multiply(IlmHandle h, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmMatrixMultiplyParams* p) {
    ILM_TRACE_RANGE("ilm_matrix_multiply");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!p || !sys) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    IlmStepDesc d;
    init_single_pass(&d, chunk_index, sys);
    d.OpCount = 1; d.Ops[0].Type = ILM_OP_MATRIX_MULTIPLY; d.Ops[0].u.MatrixMultiply = *p;
    return run_step(s, &d);
}

int32_t ilm_spatial_noise(IlmHandle h, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmSpatialNoiseParams* p) {
    ILM_TRACE_RANGE("ilm_spatial_noise");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!p || !sys) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    IlmStepDesc d;
    init_single_pass(&d, chunk_index, sys);
    d.OpCount = 1; d.Ops[0].Type = ILM_OP_SPATIAL_NOISE; d.Ops[0].u.SpatialNoise = *p;
    return run_step(s, &d);
}

int32_t ilm_update(IlmHandle h, int32_t chunk_index, const IlmParticleSystemUniforms* sys, const IlmUpdateParams* p,
                   const IlmDistanceFieldUniforms* df) {
    ILM_TRACE_RANGE("ilm_update");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!p || !sys) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    IlmStepDesc d;
    init_single_pass(&d, chunk_index, sys);
    d.Update = *p;
    if (df) { d.DistanceField = *df; d.UpdateMode = ILM_UPDATE_WITH_DISTANCE_FIELD; }
    else d.UpdateMode = ILM_UPDATE_POSITIONS;
    return run_step(s, &d);
}

int32_t ilm_erase(IlmHandle h, int32_t chunk_index) {
    ILM_TRACE_RANGE("ilm_erase");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    IlmStepDesc d;
    init_single_pass(&d, chunk_index, nullptr);
    d.UpdateMode = ILM_UPDATE_ERASE;
    return run_step(s, &d);
}

// Count of chunk i from the last counting step: chunks outside the step's range were not counted (zero); *ready goes false when the
// chunk's word does not carry the step's sequence number yet.
static uint32_t published_count(const System* s, int i, bool* ready) {
    if (i < s->counts_first || i >= s->counts_first + s->counts_span) return 0u;
    const unsigned long long w = __atomic_load_n(&s->h_counts[i], __ATOMIC_ACQUIRE);
    if ((uint32_t)(w >> 32) != s->count_seq) { if (ready) *ready = false; return 0u; }
    return (uint32_t)(w & 0xFFFFFFFFull);
}

int32_t ilm_system_live_counts(IlmHandle h, uint32_t* out_counts, int32_t capacity, int32_t saturate16) {
    ILM_TRACE_RANGE("ilm_system_live_counts");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!out_counts) return fail(ILM_ERR_INVALID_ARGUMENT, "out_counts is NULL");
    Ctx* c = s->engine->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int32_t rc = refresh_table(s);
    if (rc != ILM_OK) return rc;
    const int n = (int)s->chunks.size();
    if (n > 0) {
        HIP_TRY(hipMemsetAsync(s->counts_region(4), 0, sizeof(uint32_t) * (size_t)n * kCountStride, c->main()));
        HIP_TRY(launch_count_live(s->d_table, s->engine->stride, s->engine->span, s->engine->slots, n, reinterpret_cast<uint32_t*>(s->counts_region(4)), c->main()));
    }
    return copy_counts(s, s->counts_region(4), out_counts, capacity, saturate16);
}

int32_t ilm_system_step_counts(IlmHandle h, uint32_t* out_counts, int32_t capacity, int32_t saturate16) {
    ILM_TRACE_RANGE("ilm_system_step_counts");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!out_counts) return fail(ILM_ERR_INVALID_ARGUMENT, "out_counts is NULL");
    HIP_TRY(hipSetDevice(s->engine->ctx->device));
    if (!s->counts_valid)
        return fail(ILM_ERR_STATE, "no step with ILM_STEP_COUNT_LIVE has run");
    if (capacity < s->counts_n) return fail(ILM_ERR_OUT_OF_RANGE, "capacity %d < %d", capacity, s->counts_n);
    if (s->counts_n > 0)
        HIP_TRY(hipStreamSynchronize(s->engine->ctx->main()));   // the counting step has run: every chunk of its range is published
    bool ready = true;
    for (int i = 0; i < s->counts_n; i++) (void)published_count(s, i, &ready);
    if (!ready) return fail(ILM_ERR_STATE, "the counting step finished without publishing every chunk's count");
    for (int i = 0; i < s->counts_n; i++) {
        const uint32_t v = published_count(s, i, nullptr);
        out_counts[i] = (saturate16 && v > ref::kLiveCountSaturation) ? ref::kLiveCountSaturation : v;
    }
    return ILM_OK;
}

int32_t ilm_system_poll_counts(IlmHandle h, uint32_t* out_counts, int32_t capacity, int32_t saturate16, int32_t* out_ready) {
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!out_counts || !out_ready) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_ready = 0;
    if (!s->counts_pending) return fail(ILM_ERR_STATE, "no step with ILM_STEP_COUNT_LIVE is outstanding");
    if (capacity < s->counts_n) return fail(ILM_ERR_OUT_OF_RANGE, "capacity %d < %d", capacity, s->counts_n);
    bool ready = true;
    for (int i = 0; i < s->counts_n && ready; i++) (void)published_count(s, i, &ready);
    if (!ready) return ILM_OK;      // like the reference's deferred readback: never stall
    for (int i = 0; i < s->counts_n; i++) {
        const uint32_t v = published_count(s, i, nullptr);
        out_counts[i] = (saturate16 && v > ref::kLiveCountSaturation) ? ref::kLiveCountSaturation : v;
    }
    s->counts_pending = false;
    *out_ready = 1;
    return ILM_OK;
}

int32_t ilm_chunk_live_slots(IlmHandle h, int32_t chunk, uint32_t* out_slots, int32_t capacity, int32_t* out_count) {
    ILM_TRACE_RANGE("ilm_chunk_live_slots");
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (chunk < 0 || chunk >= (int)s->chunks.size())
        return fail(ILM_ERR_OUT_OF_RANGE, "chunk %d outside [0, %d)", chunk, (int)s->chunks.size());
    if (!out_count || capacity < 0 || (capacity > 0 && !out_slots)) return fail(ILM_ERR_INVALID_ARGUMENT, "bad output arguments");
    Engine* e = s->engine; Ctx* c = e->ctx;
    HIP_TRY(hipSetDevice(c->device));
    if (!s->d_slot_count)      // [0]: the list's length (the copying path); [1 ...]: live slots per block of 1024
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_slot_count), sizeof(uint32_t) * (size_t)(1 + (e->slots + 1023) / 1024)));
    if ((size_t)e->slots * sizeof(uint32_t) <= ((size_t)8 << 20)) {
        // the list and its length are written by the kernel straight into a pinned ring slot: one synchronisation, no copy commands
        const size_t list_bytes = sizeof(uint32_t) * (size_t)e->slots;
        unsigned char* block = nullptr;
        int slot = -1;
        const int32_t rc = upload_small_begin(c, 64 + list_bytes, reinterpret_cast<void**>(&block), &slot);
        if (rc != ILM_OK) return rc;
        void* dv = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dv, block, 0));
        uint32_t* p_count = static_cast<uint32_t*>(dv);
        uint32_t* p_slots = reinterpret_cast<uint32_t*>(static_cast<char*>(dv) + 64);
        HIP_TRY(launch_live_slots(s->chunks[(size_t)chunk] + 3 * e->stride, e->slots, p_slots, (uint32_t)e->slots, p_count, s->d_slot_count + 1, c->main()));
        const int32_t rc2 = staged_small_done(c, slot);
        if (rc2 != ILM_OK) return rc2;
        HIP_TRY(hipStreamSynchronize(c->main()));
        const uint32_t n = *reinterpret_cast<const uint32_t*>(block);
        *out_count = (int32_t)n;
        const uint32_t m = n < (uint32_t)capacity ? n : (uint32_t)capacity;
        if (m > 0) memcpy(out_slots, block + 64, sizeof(uint32_t) * (size_t)m);
        return ILM_OK;
    }
    if (s->slots_cap < e->slots) {
        if (s->d_slots) { HIP_TRY(hipStreamSynchronize(c->main())); HIP_TRY(hipFree(s->d_slots)); s->d_slots = nullptr; }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_slots), sizeof(uint32_t) * (size_t)e->slots));
        s->slots_cap = e->slots;
    }
    HIP_TRY(launch_live_slots(s->chunks[(size_t)chunk] + 3 * e->stride, e->slots, s->d_slots, (uint32_t)e->slots, s->d_slot_count, s->d_slot_count + 1, c->main()));
    uint32_t n = 0;
    HIP_TRY(hipMemcpyAsync(&n, s->d_slot_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->main()));
    HIP_TRY(hipStreamSynchronize(c->main()));
    *out_count = (int32_t)n;
    const uint32_t m = n < (uint32_t)capacity ? n : (uint32_t)capacity;
    if (m > 0) {
        HIP_TRY(hipMemcpyAsync(out_slots, s->d_slots, sizeof(uint32_t) * (size_t)m, hipMemcpyDeviceToHost, c->main()));
        HIP_TRY(hipStreamSynchronize(c->main()));
    }
    return ILM_OK;
}

// ---- lighting resources -----------------------------------------------------------------------------

int32_t ilm_sdf_create(IlmHandle hctx, int32_t w, int32_t ht, int32_t format, IlmHandle* out) {
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    if (!out) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = 0;
    // DistanceField.MaxSurfaceSize, SDF/DistanceField.cs:19
    if (w <= 0 || ht <= 0 || w > 8192 || ht > 8192) return fail(ILM_ERR_OUT_OF_RANGE, "atlas %dx%d outside (0, 8192]", w, ht);
    if (format != ILM_SDF_UNORM16 && format != ILM_SDF_FP16) return fail(ILM_ERR_INVALID_ARGUMENT, "unknown SDF format %d", format);
    HIP_TRY(hipSetDevice(c->device));
    Sdf* f = new (std::nothrow) Sdf();
    if (!f) return fail(ILM_ERR_INVALID_ARGUMENT, "out of host memory");
    f->ctx = c; f->width = w; f->height = ht; f->format = format;
    c->children++;
    const IlmHandle h = to_handle(f);
    HIP_TRY_OR_DESTROY(hipMalloc(reinterpret_cast<void**>(&f->texels), sizeof(uint2) * (size_t)w * (size_t)ht), ilm_sdf_destroy(h));
    HIP_TRY_OR_DESTROY(hipMemsetAsync(f->texels, 0, sizeof(uint2) * (size_t)w * (size_t)ht, c->main()), ilm_sdf_destroy(h));
    *out = h;
    return ILM_OK;
}

int32_t ilm_sdf_upload(IlmHandle h, const uint16_t* texels) {
    ILM_TRACE_RANGE("ilm_sdf_upload");
    Sdf* f = from_handle<Sdf>(h, kMagicSdf);
    if (!f) return fail(ILM_ERR_INVALID_HANDLE, "not a distance field handle");
    if (!texels) return fail(ILM_ERR_INVALID_ARGUMENT, "texels is NULL");
    HIP_TRY(hipSetDevice(f->ctx->device));
    HIP_TRY(shared_before_write(f->shared, f->ctx));      // sibling contexts' queued reads first
    HIP_TRY(hipMemcpyAsync(f->texels, texels, sizeof(uint2) * (size_t)f->width * (size_t)f->height, hipMemcpyHostToDevice, f->ctx->main()));
    f->mark_all_dirty();
    HIP_TRY(hipStreamSynchronize(f->ctx->main()));
    return ILM_OK;
}

int32_t ilm_sdf_sample(IlmHandle h, const IlmDistanceFieldUniforms* df, const float* positions, int32_t count, float* out_distances) {
    Sdf* f = from_handle<Sdf>(h, kMagicSdf);
    if (!f) return fail(ILM_ERR_INVALID_HANDLE, "not a distance field handle");
    if (!df || count < 0 || (count > 0 && (!positions || !out_distances))) return fail(ILM_ERR_INVALID_ARGUMENT, "bad arguments");
    if (count == 0) return ILM_OK;
    Ctx* c = f->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const size_t in_bytes = sizeof(float) * 3 * (size_t)count, out_bytes = sizeof(float) * (size_t)count;
    if (in_bytes + out_bytes <= ((size_t)4 << 20)) {
        // a point query of ordinary size: positions and distances through one pinned block the kernel reads and writes in place (every
        // word once) -- no copy command in front of or behind the one kernel (r04: 1 000 positions 35 us through the copy engine)
        unsigned char* block = nullptr;
        int slot = -1;
        int32_t rc = upload_small_begin(c, in_bytes + out_bytes, reinterpret_cast<void**>(&block), &slot);
        if (rc != ILM_OK) return rc;
        memcpy(block, positions, in_bytes);
        void* dv = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dv, block, 0));
        float* d_in = static_cast<float*>(dv);
        HIP_TRY(launch_sdf_sample(make_sdf_view(f, df), *df, d_in, count, d_in + 3 * (size_t)count, c->main()));
        rc = staged_small_done(c, slot);
        if (rc != ILM_OK) return rc;
        HIP_TRY(hipStreamSynchronize(c->main()));
        memcpy(out_distances, block + in_bytes, out_bytes);
        return ILM_OK;
    }
    int32_t rc = ensure_staging(c, in_bytes + out_bytes);
    if (rc != ILM_OK) return rc;
    float* d_in = static_cast<float*>(c->staging);
    float* d_out = d_in + 3 * (size_t)count;
    HIP_TRY(hipMemcpyAsync(d_in, positions, in_bytes, hipMemcpyHostToDevice, c->main()));
    HIP_TRY(launch_sdf_sample(make_sdf_view(f, df), *df, d_in, count, d_out, c->main()));
    HIP_TRY(hipMemcpyAsync(out_distances, d_out, out_bytes, hipMemcpyDeviceToHost, c->main()));
    HIP_TRY(hipStreamSynchronize(c->main()));
    return ILM_OK;
}

int32_t ilm_debug_sdf_sample_inside(IlmHandle h, const IlmDistanceFieldUniforms* df, const float* positions, int32_t count, float* out_distances,
                                    int32_t* out_used_table) {
    Sdf* f = from_handle<Sdf>(h, kMagicSdf);
    if (!f) return fail(ILM_ERR_INVALID_HANDLE, "not a distance field handle");
    if (!df || count < 0 || (count > 0 && (!positions || !out_distances || !out_used_table))) return fail(ILM_ERR_INVALID_ARGUMENT, "bad arguments");
    if (count == 0) return ILM_OK;
    Ctx* c = f->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const size_t in_bytes = sizeof(float) * 3 * (size_t)count, out_bytes = sizeof(float) * (size_t)count;
    int32_t rc = ensure_staging(c, in_bytes + 2 * out_bytes);
    if (rc != ILM_OK) return rc;
    float* d_in = static_cast<float*>(c->staging);
    float* d_out = d_in + 3 * (size_t)count;
    int32_t* d_used = reinterpret_cast<int32_t*>(d_out + (size_t)count);
    HIP_TRY(hipMemcpyAsync(d_in, positions, in_bytes, hipMemcpyHostToDevice, c->main()));
    TraceSdfView view;
    HIP_TRY(make_trace_view(f, df, c->main(), &view));
    HIP_TRY(launch_sdf_sample_inside(view, *df, d_in, count, d_out, d_used, c->main()));
    HIP_TRY(hipMemcpyAsync(out_distances, d_out, out_bytes, hipMemcpyDeviceToHost, c->main()));
    HIP_TRY(hipMemcpyAsync(out_used_table, d_used, out_bytes, hipMemcpyDeviceToHost, c->main()));
    HIP_TRY(hipStreamSynchronize(c->main()));
    return ILM_OK;
}

int32_t ilm_debug_divide_by_constants(IlmHandle hctx, float* out_divisors, uint64_t* out_mismatches, int32_t capacity, int32_t* out_count) {
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    if (!out_divisors || !out_mismatches || !out_count || capacity < 2) return fail(ILM_ERR_INVALID_ARGUMENT, "bad arguments");   // out_mismatches: 2 per divisor
    HIP_TRY(hipSetDevice(c->device));
    float pairs[2][2];
    light_constant_divisors(pairs);
    for (int i = 0; i < 2; i++) {
        HIP_TRY(hipMemsetAsync(c->d_stats, 0, 2 * sizeof(unsigned long long), c->main()));
        HIP_TRY(launch_divide_by_constant(pairs[i][0], pairs[i][1], c->d_stats, c->main()));
        unsigned long long bad[2] = { 0, 0 };
        HIP_TRY(hipMemcpyAsync(bad, c->d_stats, sizeof(bad), hipMemcpyDeviceToHost, c->main()));
        HIP_TRY(hipStreamSynchronize(c->main()));
        out_divisors[i] = pairs[i][0];
        out_mismatches[2 * i] = bad[0];
        out_mismatches[2 * i + 1] = bad[1];
    }
    *out_count = 2;
    return ILM_OK;
}

int32_t ilm_debug_step_interpreter(int32_t interpreter) { return (int32_t)set_step_interpreter(interpreter); }
int32_t ilm_debug_step_streams(int32_t streams) { return (int32_t)set_step_streams(streams); }
int32_t ilm_debug_last_light_launch(IlmHandle hctx, int32_t* out_workgroups, int32_t* out_split, int32_t* out_tile_macro) {
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    if (out_workgroups) *out_workgroups = c->last_light_blocks;
    if (out_split) *out_split = c->last_light_split;
    if (out_tile_macro) *out_tile_macro = c->last_light_macro;
    return ILM_OK;
}

int32_t ilm_debug_step_sdf_samples(IlmHandle hctx, int32_t enable, uint64_t* out_samples) {
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->main()));        // main() first joins the second stream
    unsigned long long value = 0;
    HIP_TRY(step_sdf_sample_counter(enable, &value));
    if (out_samples) *out_samples = (uint64_t)value;
    return ILM_OK;
}

int32_t ilm_debug_divide(IlmHandle hctx, const float* numerators, const float* denominators, int32_t count, float* out_fast, float* out_ieee) {
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    if (count < 0 || (count > 0 && (!numerators || !denominators || !out_fast || !out_ieee))) return fail(ILM_ERR_INVALID_ARGUMENT, "bad arguments");
    if (count == 0) return ILM_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = sizeof(float) * (size_t)count;
    int32_t rc = ensure_staging(c, 4 * bytes);
    if (rc != ILM_OK) return rc;
    float* d_n = static_cast<float*>(c->staging);
    float* d_d = d_n + count; float* d_f = d_d + count; float* d_i = d_f + count;
    HIP_TRY(hipMemcpyAsync(d_n, numerators, bytes, hipMemcpyHostToDevice, c->main()));
    HIP_TRY(hipMemcpyAsync(d_d, denominators, bytes, hipMemcpyHostToDevice, c->main()));
    HIP_TRY(launch_divide_probe(d_n, d_d, count, d_f, d_i, c->main()));
    HIP_TRY(hipMemcpyAsync(out_fast, d_f, bytes, hipMemcpyDeviceToHost, c->main()));
    HIP_TRY(hipMemcpyAsync(out_ieee, d_i, bytes, hipMemcpyDeviceToHost, c->main()));
    HIP_TRY(hipStreamSynchronize(c->main()));
    return ILM_OK;
}

int32_t ilm_sdf_destroy(IlmHandle h) {
    Sdf* f = from_handle<Sdf>(h, kMagicSdf);
    if (!f) return fail(ILM_ERR_INVALID_HANDLE, "not a distance field handle");
    f->ctx->children--;
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->ctx->main());
    for (auto& r : f->shared.readers) (void)hipEventSynchronize(r.second);      // sibling contexts' light passes still reading it
    f->shared.release();
    if (f->texels) (void)hipFree(f->texels);
    if (f->cells) (void)hipFree(f->cells);
    if (f->cells0) (void)hipFree(f->cells0);
    retire_handle(f);
    delete f;
    return ILM_OK;
}

int32_t ilm_sdf_download(IlmHandle h, uint16_t* texels) {
    ILM_TRACE_RANGE("ilm_sdf_download");
    Sdf* f = from_handle<Sdf>(h, kMagicSdf);
    if (!f) return fail(ILM_ERR_INVALID_HANDLE, "not a distance field handle");
    if (!texels) return fail(ILM_ERR_INVALID_ARGUMENT, "texels is NULL");
    HIP_TRY(hipSetDevice(f->ctx->device));
    HIP_TRY(hipMemcpyAsync(texels, f->texels, sizeof(uint2) * (size_t)f->width * (size_t)f->height, hipMemcpyDeviceToHost, f->ctx->main()));
    HIP_TRY(hipStreamSynchronize(f->ctx->main()));
    return ILM_OK;
}

int32_t ilm_sdf_device_ptr(IlmHandle h, void** out_ptr) {
    Sdf* f = from_handle<Sdf>(h, kMagicSdf);
    if (!f || !out_ptr) return fail(ILM_ERR_INVALID_HANDLE, "not a distance field handle");
    *out_ptr = f->texels;
    f->escaped = true;      // the caller may write the atlas: the trace's cell array is rebuilt before every use from now on (until ilm_sdf_mark_dirty)
    return ILM_OK;
}

int32_t ilm_sdf_mark_dirty(IlmHandle h, int32_t first_virtual_slice, int32_t slice_count) {
    Sdf* f = from_handle<Sdf>(h, kMagicSdf);
    if (!f) return fail(ILM_ERR_INVALID_HANDLE, "not a distance field handle");
    if (slice_count < 0 || first_virtual_slice < 0) return fail(ILM_ERR_OUT_OF_RANGE, "slices [%d, %d + %d)", first_virtual_slice, first_virtual_slice, slice_count);
    if (slice_count == 0) f->mark_all_dirty(); else f->mark_slices_dirty(first_virtual_slice, slice_count);
    f->escaped = false;     // the caller reports its writes from now on
    return ILM_OK;
}

int32_t ilm_sdf_trace_info(IlmHandle h, IlmSdfTraceInfo* out) {
    Sdf* f = from_handle<Sdf>(h, kMagicSdf);
    if (!f || !out) return fail(ILM_ERR_INVALID_HANDLE, "not a distance field handle");
    out->CellBytes = (uint64_t)f->cells_bytes;
    out->CellRebuilds = f->cell_rebuilds; out->CellSlicesRebuilt = f->cell_slices_rebuilt;
    out->LastRebuiltSlices = f->last_rebuilt_slices; out->TableSlices = f->last_table_slices;
    out->RebuiltEveryFrame = f->escaped ? 1 : 0; out->Reserved = 0;
    return ILM_OK;
}

int32_t ilm_sdf_render_slices(IlmHandle h, IlmHandle hclear, const IlmDistanceFieldRenderDesc* d,
                              const int32_t* first_virtual_slices, int32_t triplet_count,
                              const IlmObstruction* obstructions, int32_t obstruction_count,
                              const IlmHeightVolume* volumes, int32_t volume_count,
                              const float* polygon_xy, int32_t polygon_vertex_count) {
    ILM_TRACE_RANGE("ilm_sdf_render_slices");
    Sdf* f = from_handle<Sdf>(h, kMagicSdf);
    if (!f) return fail(ILM_ERR_INVALID_HANDLE, "not a distance field handle");
    Sdf* clr = nullptr;
    if (hclear) {
        clr = from_handle<Sdf>(hclear, kMagicSdf);
        if (!clr) return fail(ILM_ERR_INVALID_HANDLE, "clear source is not a distance field handle");
        if (clr->ctx != f->ctx || clr->width != f->width || clr->height != f->height || clr->format != f->format)
            return fail(ILM_ERR_INVALID_ARGUMENT, "clear source must match the distance field (context, size, format)");
        if (clr == f) return fail(ILM_ERR_INVALID_ARGUMENT, "clear source is the target itself");
    }
    if (!d) return fail(ILM_ERR_INVALID_ARGUMENT, "desc is NULL");
    if (triplet_count < 0 || obstruction_count < 0 || volume_count < 0 || polygon_vertex_count < 0 ||
        (triplet_count > 0 && !first_virtual_slices) || (obstruction_count > 0 && !obstructions) ||
        (volume_count > 0 && (!volumes || !polygon_xy)))
        return fail(ILM_ERR_INVALID_ARGUMENT, "bad array argument");
    if (d->SliceWidth <= 0 || d->SliceHeight <= 0 || d->ColumnCount <= 0 || d->RowCount <= 0 || d->SliceCount <= 0 ||
        d->VirtualWidth <= 0 || d->VirtualHeight <= 0 || !(d->MaximumEncodedDistance > 0.0f))
        return fail(ILM_ERR_INVALID_ARGUMENT, "bad distance field layout");
    if (d->SliceWidth * d->ColumnCount != f->width || d->SliceHeight * d->RowCount != f->height)
        return fail(ILM_ERR_INVALID_ARGUMENT, "layout %dx%d slices of %dx%d does not match the %dx%d atlas",
                    d->ColumnCount, d->RowCount, d->SliceWidth, d->SliceHeight, f->width, f->height);
    if (obstruction_count > 65535) return fail(ILM_ERR_TOO_MANY, "at most 65535 obstructions per call");
    const int physical_count = d->ColumnCount * d->RowCount;
    for (int i = 0; i < triplet_count; i++) {
        const int s = first_virtual_slices[i];
        if (s < 0 || (s % 3) != 0 || s / 3 >= physical_count)
            return fail(ILM_ERR_OUT_OF_RANGE, "first virtual slice %d is not a triplet start inside the atlas", s);
    }
    for (int i = 0; i < obstruction_count; i++)
        if (obstructions[i].Type < ILM_OBSTRUCTION_ELLIPSOID || obstructions[i].Type > ILM_OBSTRUCTION_OCTAGON)
            return fail(ILM_ERR_INVALID_ARGUMENT, "obstruction %d has unknown type %d", i, obstructions[i].Type);
    for (int i = 0; i < volume_count; i++)
        if (volumes[i].FirstVertex < 0 || volumes[i].VertexCount < 0 || volumes[i].FirstVertex + volumes[i].VertexCount > polygon_vertex_count)
            return fail(ILM_ERR_OUT_OF_RANGE, "height volume %d vertex range outside the polygon array", i);
    if (triplet_count == 0) return ILM_OK;
    Ctx* c = f->ctx;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(shared_before_write(f->shared, c));           // sibling contexts' queued reads of the atlas / cells first

    // the orthographic view transform maps [0, VirtualWidth * ColumnCount] onto the atlas width
    // (RenderDistanceFieldSliceTriplet, LightingRenderer.DistanceField.cs:97-102): virtual units -> slice pixels
    const float px_per_unit_x = (float)d->SliceWidth / (float)d->VirtualWidth;
    const float px_per_unit_y = (float)d->SliceHeight / (float)d->VirtualHeight;
    const bool filter = d->DynamicFlagFilter >= 0, want_dynamic = d->DynamicFlagFilter != 0;

    std::vector<FieldObstruction> recs;
    recs.reserve((size_t)obstruction_count);
    for (int i = 0; i < obstruction_count; i++) {
        const IlmObstruction& o = obstructions[i];
        if (filter && ((o.IsDynamic != 0) != want_dynamic)) continue;   // BuildDistanceFieldDistanceFunctionBuffer, :321-322
        FieldObstruction r;
        r.cx = o.Center[0]; r.cy = o.Center[1]; r.cz = o.Center[2]; r.type = o.Type;
        r.sx = o.Size[0]; r.sy = o.Size[1]; r.sz = o.Size[2]; r._pad = 0;
        r.qx = o.Orientation[0]; r.qy = o.Orientation[1]; r.qz = o.Orientation[2]; r.qw = o.Orientation[3];
        // The identity quaternion (what an unrotated LightObstruction carries) turns rotateLocalPosition into two quaternion products
        // of zeros and ones: 56 operations per evaluation whose result is the input, up to the sign of a zero component -- which no
        // shape function can see (they take |p|, p * p, p / size or sgn(p) * ...).  Flagged here, skipped in fields.hip; a centre that
        // is not finite keeps the products (inf * 0 is a NaN there).
        r._pad = (r.qx == 0.0f && r.qy == 0.0f && r.qz == 0.0f && r.qw == 1.0f && std::isfinite(r.cx) && std::isfinite(r.cy) && std::isfinite(r.cz)) ? 1 : 0;
        // bit 1: sizes and centre of ordinary magnitude (fields.hip: divisions by the sizes may share a refined reciprocal)
        auto ordinary_size = [](float v) { return v >= 0x1p-10f && v <= 0x1p20f; };
        if (ordinary_size(r.sx) && ordinary_size(r.sy) && ordinary_size(r.sz) && std::fabs(r.cx) <= 0x1p20f && std::fabs(r.cy) <= 0x1p20f && std::fabs(r.cz) <= 0x1p20f)
            r._pad |= 2;
        // DistanceFunctionVertexShader, DistanceFunction.fx:16-26
        const float msize = fmaxf(fmaxf(fabsf(o.Size[0]), fabsf(o.Size[1])), fabsf(o.Size[2])) + d->MaximumEncodedDistance + 4.0f;
        r.x0 = (o.Center[0] - msize) * px_per_unit_x; r.x1 = (o.Center[0] + msize) * px_per_unit_x;
        r.y0 = (o.Center[1] - msize) * px_per_unit_y; r.y1 = (o.Center[1] + msize) * px_per_unit_y;
        fill_cull_bound(r);
        recs.push_back(r);
    }
    std::vector<FieldVolume> vols;
    vols.reserve((size_t)volume_count);
    for (int i = 0; i < volume_count; i++) {
        const IlmHeightVolume& hv = volumes[i];
        if (filter && ((hv.IsDynamic != 0) != want_dynamic)) continue;   // :205-206
        if (hv.VertexCount < 1) continue;
        const float* P = polygon_xy + 2 * (size_t)hv.FirstVertex;
        float bx0 = P[0], bx1 = P[0], by0 = P[1], by1 = P[1];
        for (int e = 1; e < hv.VertexCount; e++) {
            bx0 = fminf(bx0, P[2 * e]); bx1 = fmaxf(bx1, P[2 * e]);
            by0 = fminf(by0, P[2 * e + 1]); by1 = fmaxf(by1, P[2 * e + 1]);
        }
        FieldVolume v;
        v.first_vertex = hv.FirstVertex; v.vertex_count = hv.VertexCount;
        v.z0 = hv.ZBase; v.z1 = hv.ZBase + hv.Height;
        // hv.Bounds.Expand(DistanceLimit, DistanceLimit), :216
        v.x0 = (bx0 - ILM_DISTANCE_LIMIT) * px_per_unit_x; v.x1 = (bx1 + ILM_DISTANCE_LIMIT) * px_per_unit_x;
        v.y0 = (by0 - ILM_DISTANCE_LIMIT) * px_per_unit_y; v.y1 = (by1 + ILM_DISTANCE_LIMIT) * px_per_unit_y;
        // a circle around the polygon (centre of its bounds, the farthest vertex, rounded up): outside it a texel's distance to the
        // polygon is at least its distance to the circle
        const double ccx = 0.5 * ((double)bx0 + (double)bx1), ccy = 0.5 * ((double)by0 + (double)by1);
        double rr = 0.0;
        for (int e = 0; e < hv.VertexCount; e++) rr = std::max(rr, std::hypot((double)P[2 * e] - ccx, (double)P[2 * e + 1] - ccy));
        v.cx = (float)ccx; v.cy = (float)ccy;
        v.radius = (float)(rr * (1.0 + 1e-6) + 1e-3 + std::fabs(ccx - (double)(float)ccx) + std::fabs(ccy - (double)(float)ccy));
        v.radius = std::nextafter(v.radius, INFINITY);          // (a NaN vertex leaves a NaN radius: the kernel never culls on it)
        v._pad = 0.0f;
        vols.push_back(v);
    }

    // one parameter block: [slices | obstruction records | volumes | polygon vertices], 64-byte aligned sections
    auto align64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t off_slices = 0;
    const size_t off_recs = align64(off_slices + sizeof(int32_t) * (size_t)triplet_count);
    const size_t off_vols = align64(off_recs + sizeof(FieldObstruction) * recs.size());
    const size_t off_poly = align64(off_vols + sizeof(FieldVolume) * vols.size());
    const size_t total = align64(off_poly + sizeof(float) * 2 * (size_t)(vols.empty() ? 0 : polygon_vertex_count));
    if (total > c->field_params_bytes) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_field_params) HIP_TRY(hipFree(c->d_field_params));
        c->d_field_params = nullptr; c->field_params_bytes = 0;
        const size_t cap = total < 65536 ? 65536 : total * 2;
        HIP_TRY(hipMalloc(&c->d_field_params, cap));
        c->field_params_bytes = cap;
    }
    std::vector<unsigned char> block(total, 0);
    memcpy(block.data() + off_slices, first_virtual_slices, sizeof(int32_t) * (size_t)triplet_count);
    if (!recs.empty()) memcpy(block.data() + off_recs, recs.data(), sizeof(FieldObstruction) * recs.size());
    if (!vols.empty()) {
        memcpy(block.data() + off_vols, vols.data(), sizeof(FieldVolume) * vols.size());
        memcpy(block.data() + off_poly, polygon_xy, sizeof(float) * 2 * (size_t)polygon_vertex_count);
    }
    int32_t rc = upload_small(c, c->d_field_params, block.data(), total);
    if (rc != ILM_OK) return rc;

    FieldLaunch a;
    char* base = static_cast<char*>(c->d_field_params);
    a.atlas = f->texels; a.clear_source = clr ? clr->texels : nullptr; a.atlas_w = f->width;
    a.first_slices = reinterpret_cast<const int32_t*>(base + off_slices); a.triplet_count = triplet_count;
    a.obstructions = reinterpret_cast<const FieldObstruction*>(base + off_recs); a.obstruction_count = (int32_t)recs.size();
    a.volumes = reinterpret_cast<const FieldVolume*>(base + off_vols); a.volume_count = (int32_t)vols.size();
    a.polygon_xy = reinterpret_cast<const float2*>(base + off_poly);
    a.slice_w = d->SliceWidth; a.slice_h = d->SliceHeight; a.columns = d->ColumnCount;
    a.virtual_w = d->VirtualWidth; a.virtual_h = d->VirtualHeight;
    a.slice_count_f = (float)d->SliceCount < 1.0f ? 1.0f : (float)d->SliceCount;   // Math.Max(1, (float)SliceCount), :33
    a.virtual_depth = d->VirtualDepth; a.z_offset = d->ZOffset; a.max_encoded = d->MaximumEncodedDistance;
    a.inv_scale_x = d->InvScaleFactorX; a.inv_scale_y = d->InvScaleFactorY;
    HIP_TRY(launch_render_slices(a, f->format, c->main()));
    for (int32_t i = 0; i < triplet_count; i++) f->mark_slices_dirty(first_virtual_slices[i], 3);
    return ILM_OK;
}

int32_t ilm_gbuffer_create(IlmHandle hctx, int32_t w, int32_t ht, int32_t format, IlmHandle* out) {
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    if (!out) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = 0;
    if (w <= 0 || ht <= 0) return fail(ILM_ERR_OUT_OF_RANGE, "bad G-buffer size %dx%d", w, ht);
    if (format != ILM_GBUFFER_FLOAT4 && format != ILM_GBUFFER_HALF4) return fail(ILM_ERR_INVALID_ARGUMENT, "unknown G-buffer format %d", format);
    HIP_TRY(hipSetDevice(c->device));
    GBuffer* g = new (std::nothrow) GBuffer();
    if (!g) return fail(ILM_ERR_INVALID_ARGUMENT, "out of host memory");
    g->ctx = c; g->width = w; g->height = ht; g->format = format;
    c->children++;
    const size_t bytes = (format == ILM_GBUFFER_FLOAT4 ? 16u : 8u) * (size_t)w * (size_t)ht;
    const IlmHandle h = to_handle(g);
    HIP_TRY_OR_DESTROY(hipMalloc(&g->texels, bytes), ilm_gbuffer_destroy(h));
    HIP_TRY_OR_DESTROY(hipMemsetAsync(g->texels, 0, bytes, c->main()), ilm_gbuffer_destroy(h));
    *out = h;
    return ILM_OK;
}

int32_t ilm_gbuffer_upload(IlmHandle h, const void* texels) {
    ILM_TRACE_RANGE("ilm_gbuffer_upload");
    GBuffer* g = from_handle<GBuffer>(h, kMagicGBuffer);
    if (!g) return fail(ILM_ERR_INVALID_HANDLE, "not a G-buffer handle");
    if (!texels) return fail(ILM_ERR_INVALID_ARGUMENT, "texels is NULL");
    HIP_TRY(hipSetDevice(g->ctx->device));
    HIP_TRY(shared_before_write(g->shared, g->ctx));
    const size_t bytes = (g->format == ILM_GBUFFER_FLOAT4 ? 16u : 8u) * (size_t)g->width * (size_t)g->height;
    HIP_TRY(hipMemcpyAsync(g->texels, texels, bytes, hipMemcpyHostToDevice, g->ctx->main()));
    HIP_TRY(hipStreamSynchronize(g->ctx->main()));
    return ILM_OK;
}

int32_t ilm_gbuffer_download(IlmHandle h, void* texels) {
    GBuffer* g = from_handle<GBuffer>(h, kMagicGBuffer);
    if (!g) return fail(ILM_ERR_INVALID_HANDLE, "not a G-buffer handle");
    if (!texels) return fail(ILM_ERR_INVALID_ARGUMENT, "texels is NULL");
    HIP_TRY(hipSetDevice(g->ctx->device));
    const size_t bytes = (g->format == ILM_GBUFFER_FLOAT4 ? 16u : 8u) * (size_t)g->width * (size_t)g->height;
    HIP_TRY(hipMemcpyAsync(texels, g->texels, bytes, hipMemcpyDeviceToHost, g->ctx->main()));
    HIP_TRY(hipStreamSynchronize(g->ctx->main()));
    return ILM_OK;
}

int32_t ilm_gbuffer_render(IlmHandle h, const IlmGBufferRenderDesc* d, const IlmHeightVolume* volumes, int32_t volume_count,
                           const float* polygon_xy, int32_t polygon_vertex_count) {
    ILM_TRACE_RANGE("ilm_gbuffer_render");
    GBuffer* g = from_handle<GBuffer>(h, kMagicGBuffer);
    if (!g) return fail(ILM_ERR_INVALID_HANDLE, "not a G-buffer handle");
    if (!d) return fail(ILM_ERR_INVALID_ARGUMENT, "desc is NULL");
    if (volume_count < 0 || polygon_vertex_count < 0 || (volume_count > 0 && (!volumes || !polygon_xy)))
        return fail(ILM_ERR_INVALID_ARGUMENT, "bad array argument");
    if (!(d->ViewportScale[0] > 0.0f) || !(d->ViewportScale[1] > 0.0f)) return fail(ILM_ERR_INVALID_ARGUMENT, "ViewportScale must be positive");
    for (int i = 0; i < volume_count; i++)
        if (volumes[i].FirstVertex < 0 || volumes[i].VertexCount < 0 || volumes[i].FirstVertex + volumes[i].VertexCount > polygon_vertex_count)
            return fail(ILM_ERR_OUT_OF_RANGE, "height volume %d vertex range outside the polygon array", i);
    Ctx* c = g->ctx;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(shared_before_write(g->shared, c));
    // RenderGBufferVolumes: OrderBy(hv => hv.ZBase + hv.Height) (stable), LightingRenderer.GBuffer.cs:210
    std::vector<int> order;
    for (int i = 0; i < volume_count; i++)
        if (volumes[i].VertexCount >= 3) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
        return (volumes[x].ZBase + volumes[x].Height) < (volumes[y].ZBase + volumes[y].Height); });
    std::vector<GBufferVolume> vols;
    for (int i : order) {
        const IlmHeightVolume& hv = volumes[i];
        const float* P = polygon_xy + 2 * (size_t)hv.FirstVertex;
        GBufferVolume v;
        v.first_vertex = hv.FirstVertex; v.vertex_count = hv.VertexCount;
        v.top = hv.ZBase + hv.Height; v.enable_shadows = hv.TopFaceEnableShadows;
        v.x0 = v.x1 = P[0]; v.y0 = v.y1 = P[1];
        for (int e = 1; e < hv.VertexCount; e++) {
            v.x0 = fminf(v.x0, P[2 * e]); v.x1 = fmaxf(v.x1, P[2 * e]);
            v.y0 = fminf(v.y0, P[2 * e + 1]); v.y1 = fmaxf(v.y1, P[2 * e + 1]);
        }
        vols.push_back(v);
    }
    auto align64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t off_poly = align64(sizeof(GBufferVolume) * vols.size());
    const size_t total = align64(off_poly + sizeof(float) * 2 * (size_t)(vols.empty() ? 0 : polygon_vertex_count)) + 64;
    if (total > c->field_params_bytes) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_field_params) HIP_TRY(hipFree(c->d_field_params));
        c->d_field_params = nullptr; c->field_params_bytes = 0;
        const size_t cap = total < 65536 ? 65536 : total * 2;
        HIP_TRY(hipMalloc(&c->d_field_params, cap));
        c->field_params_bytes = cap;
    }
    if (!vols.empty()) {
        std::vector<unsigned char> block(total, 0);
        memcpy(block.data(), vols.data(), sizeof(GBufferVolume) * vols.size());
        memcpy(block.data() + off_poly, polygon_xy, sizeof(float) * 2 * (size_t)polygon_vertex_count);
        int32_t rc = upload_small(c, c->d_field_params, block.data(), total);
        if (rc != ILM_OK) return rc;
    }
    GBufferLaunch a;
    a.texels = g->texels; a.width = g->width; a.height = g->height; a.format = g->format;
    a.desc = *d;
    a.volumes = reinterpret_cast<const GBufferVolume*>(c->d_field_params); a.volume_count = (int32_t)vols.size();
    a.polygon_xy = reinterpret_cast<const float2*>(static_cast<char*>(c->d_field_params) + off_poly);
    HIP_TRY(launch_render_gbuffer(a, c->main()));
    return ILM_OK;
}

int32_t ilm_gbuffer_render_meshes(IlmHandle h, const IlmGBufferMeshDesc* d,
                                  const IlmHeightVolumeVertex* top_vertices, int32_t top_vertex_count,
                                  const IlmHeightVolumeVertex* front_vertices, int32_t front_vertex_count,
                                  const IlmBillboardVertex* billboard_vertices, int32_t billboard_vertex_count,
                                  const IlmBillboardRun* runs, int32_t run_count) {
    ILM_TRACE_RANGE("ilm_gbuffer_render_meshes");
    GBuffer* g = from_handle<GBuffer>(h, kMagicGBuffer);
    if (!g) return fail(ILM_ERR_INVALID_HANDLE, "not a G-buffer handle");
    if (!d) return fail(ILM_ERR_INVALID_ARGUMENT, "desc is NULL");
    if (top_vertex_count < 0 || front_vertex_count < 0 || billboard_vertex_count < 0 || run_count < 0 ||
        (top_vertex_count > 0 && !top_vertices) || (front_vertex_count > 0 && !front_vertices) ||
        (billboard_vertex_count > 0 && !billboard_vertices) || (run_count > 0 && !runs))
        return fail(ILM_ERR_INVALID_ARGUMENT, "bad array argument");
    if ((top_vertex_count % 3) != 0 || (front_vertex_count % 3) != 0 || (billboard_vertex_count % 4) != 0)
        return fail(ILM_ERR_INVALID_ARGUMENT, "triangle lists hold 3 vertices per triangle, billboards 4 per quad");
    if (!(d->ViewportScale[0] > 0.0f) || !(d->ViewportScale[1] > 0.0f)) return fail(ILM_ERR_INVALID_ARGUMENT, "ViewportScale must be positive");
    if (!d->TwoPointFiveD && front_vertex_count > 0)
        return fail(ILM_ERR_INVALID_ARGUMENT, "front faces are only drawn with TwoPointFiveD (LightingRenderer.GBuffer.cs:160-167)");
    if (d->TwoPointFiveD && (top_vertex_count > 0 || front_vertex_count > 0) && !(d->DistanceFieldExtentZ > 0.0f))
        return fail(ILM_ERR_INVALID_ARGUMENT, "DistanceFieldExtentZ must be positive: the depth of a 2.5D vertex is z / DistanceFieldExtent.z");
    Ctx* c = g->ctx;
    // billboard quads in draw order: the mask batch is one layer below the g-data batch (:371-392)
    std::vector<int4> quads;
    std::vector<GBufferTex> textures((size_t)run_count);
    for (int r = 0; r < run_count; r++) {
        const IlmBillboardRun& run = runs[r];
        if (run.Type != ILM_BILLBOARD_MASK && run.Type != ILM_BILLBOARD_GBUFFER_DATA)
            return fail(ILM_ERR_INVALID_ARGUMENT, "billboard run %d: unknown type %d", r, run.Type);
        if (run.FirstQuad < 0 || run.QuadCount < 0 || (int64_t)run.FirstQuad + run.QuadCount > billboard_vertex_count / 4)
            return fail(ILM_ERR_OUT_OF_RANGE, "billboard run %d outside the vertex array", r);
        GBufferTex t = { nullptr, 0, 0, 0, 0 };
        if (run.Texture != 0) {
            Lightmap* m = from_handle<Lightmap>(run.Texture, kMagicLightmap);
            if (!m) return fail(ILM_ERR_INVALID_HANDLE, "billboard run %d: not a texture handle", r);
            if (m->ctx != c) return fail(ILM_ERR_INVALID_ARGUMENT, "billboard run %d: the texture belongs to another context", r);
            t.texels = m->texels; t.width = m->width; t.height = m->height; t.format = m->format;
        }
        textures[(size_t)r] = t;
    }
    for (int type = ILM_BILLBOARD_MASK; type <= ILM_BILLBOARD_GBUFFER_DATA; type++)
        for (int r = 0; r < run_count; r++)
            if (runs[r].Type == type)
                for (int q = runs[r].FirstQuad; q < runs[r].FirstQuad + runs[r].QuadCount; q++)
                    quads.push_back(make_int4(q, r, type == ILM_BILLBOARD_MASK ? 3 : 4, 0));
    const int64_t prim_count = 2 + (int64_t)top_vertex_count / 3 + front_vertex_count / 3 + 2 * (int64_t)quads.size();
    if (prim_count > (1 << 24)) return fail(ILM_ERR_TOO_MANY, "%lld triangles in one G-buffer frame", (long long)prim_count);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(shared_before_write(g->shared, c));
    auto align64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t off_front = align64(sizeof(IlmHeightVolumeVertex) * (size_t)top_vertex_count);
    const size_t off_bb = align64(off_front + sizeof(IlmHeightVolumeVertex) * (size_t)front_vertex_count);
    const size_t off_quads = align64(off_bb + sizeof(IlmBillboardVertex) * (size_t)billboard_vertex_count);
    const size_t off_tex = align64(off_quads + sizeof(int4) * quads.size());
    const size_t inputs = align64(off_tex + sizeof(GBufferTex) * textures.size()) + 64;
    const size_t off_bounds = inputs + align64(sizeof(GBufferPrim) * (size_t)prim_count);
    // coarse bins: 64 x 64 pixel blocks, doubled until the blocks' lists (one slot per triangle each) fit the budget
    int block_shift = 6;
    auto blocks_at = [&](int shift) { return (size_t)((g->width + (1 << shift) - 1) >> shift) * (size_t)((g->height + (1 << shift) - 1) >> shift); };
    size_t budget = kGBufferBlockListBudget;
    if (const char* e = getenv("ILM_GBUFFER_BLOCK_LIST_BYTES")) budget = (size_t)strtoull(e, nullptr, 10);      // (tests: the larger blocks without a million triangles)
    while (blocks_at(block_shift) > 1 && blocks_at(block_shift) * (size_t)prim_count * sizeof(int32_t) > budget)
        block_shift++;
    const size_t block_count = blocks_at(block_shift);
    const size_t off_verts = align64(off_bounds + sizeof(int4) * (size_t)prim_count);
    const size_t off_block_count = align64(off_verts + 2 * sizeof(int4) * (size_t)prim_count);
    const size_t off_block_list = align64(off_block_count + sizeof(int32_t) * block_count);
    const size_t total = off_block_list + sizeof(int32_t) * block_count * (size_t)prim_count;
    if (total > c->field_params_bytes) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_field_params) HIP_TRY(hipFree(c->d_field_params));
        c->d_field_params = nullptr; c->field_params_bytes = 0;
        const size_t cap = total < 65536 ? 65536 : total * 2;
        HIP_TRY(hipMalloc(&c->d_field_params, cap));
        c->field_params_bytes = cap;
    }
    // the frame's inputs are gathered straight into a pinned slot of the ring (one host copy) and go to the device as one block
    unsigned char* block = nullptr;
    int slot = -1;
    int32_t rc = upload_small_begin(c, inputs, reinterpret_cast<void**>(&block), &slot);
    if (rc != ILM_OK) return rc;
    if (top_vertex_count) memcpy(block, top_vertices, sizeof(IlmHeightVolumeVertex) * (size_t)top_vertex_count);
    if (front_vertex_count) memcpy(block + off_front, front_vertices, sizeof(IlmHeightVolumeVertex) * (size_t)front_vertex_count);
    if (billboard_vertex_count) memcpy(block + off_bb, billboard_vertices, sizeof(IlmBillboardVertex) * (size_t)billboard_vertex_count);
    if (!quads.empty()) memcpy(block + off_quads, quads.data(), sizeof(int4) * quads.size());
    if (!textures.empty()) memcpy(block + off_tex, textures.data(), sizeof(GBufferTex) * textures.size());
    // ... or is read by the setup kernel where it lies (the default: one device operation and one dependent-launch gap less; the
    // vertices are read once, by the one kernel that digests them; ILM_GBUFFER_IN_PLACE=0 copies)
    static const int in_place = [] { const char* e = getenv("ILM_GBUFFER_IN_PLACE"); return e ? atoi(e) : 1; }();
    char* base = static_cast<char*>(c->d_field_params);
    const char* in = base;
    if (in_place) {
        void* dv = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dv, block, 0));
        in = static_cast<const char*>(dv);
    } else {
        rc = upload_small_commit(c, c->d_field_params, slot, inputs);
        if (rc != ILM_OK) return rc;
    }
    GBufferMeshLaunch a;
    a.texels = g->texels; a.width = g->width; a.height = g->height; a.format = g->format;
    a.desc = *d;
    a.top = reinterpret_cast<const IlmHeightVolumeVertex*>(in); a.top_triangles = top_vertex_count / 3;
    a.front = reinterpret_cast<const IlmHeightVolumeVertex*>(in + off_front); a.front_triangles = front_vertex_count / 3;
    a.billboards = reinterpret_cast<const IlmBillboardVertex*>(in + off_bb);
    a.quads = reinterpret_cast<const int4*>(in + off_quads);
    a.textures = reinterpret_cast<const GBufferTex*>(base + off_tex);
    a.textures_in = in_place ? reinterpret_cast<const GBufferTex*>(in + off_tex) : nullptr; a.texture_count = (int32_t)textures.size();
    a.prims = reinterpret_cast<GBufferPrim*>(base + inputs); a.prim_count = (int32_t)prim_count;
    a.bounds = reinterpret_cast<int4*>(base + off_bounds);
    a.verts = reinterpret_cast<int4*>(base + off_verts);
    a.block_shift = block_shift;
    a.block_cols = (g->width + (1 << block_shift) - 1) >> block_shift; a.block_rows = (g->height + (1 << block_shift) - 1) >> block_shift;
    a.block_count = reinterpret_cast<int32_t*>(base + off_block_count);
    a.block_list = reinterpret_cast<int32_t*>(base + off_block_list);
    HIP_TRY(launch_gbuffer_meshes(a, c->main()));
    if (in_place) return staged_small_done(c, slot);           // the slot is free again once the setup kernel has run
    return ILM_OK;
}

int32_t ilm_gbuffer_destroy(IlmHandle h) {
    GBuffer* g = from_handle<GBuffer>(h, kMagicGBuffer);
    if (!g) return fail(ILM_ERR_INVALID_HANDLE, "not a G-buffer handle");
    g->ctx->children--;
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->main());
    for (auto& r : g->shared.readers) (void)hipEventSynchronize(r.second);
    g->shared.release();
    if (g->texels) (void)hipFree(g->texels);
    retire_handle(g);
    delete g;
    return ILM_OK;
}

int32_t ilm_lightmap_create(IlmHandle hctx, int32_t w, int32_t ht, int32_t format, void* external, IlmHandle* out) {
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    if (!out) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = 0;
    if (w <= 0 || ht <= 0) return fail(ILM_ERR_OUT_OF_RANGE, "bad lightmap size %dx%d", w, ht);
    if (format < ILM_LIGHTMAP_FLOAT4 || format > ILM_LIGHTMAP_RGBA8) return fail(ILM_ERR_INVALID_ARGUMENT, "unknown lightmap format %d", format);
    HIP_TRY(hipSetDevice(c->device));
    Lightmap* m = new (std::nothrow) Lightmap();
    if (!m) return fail(ILM_ERR_INVALID_ARGUMENT, "out of host memory");
    m->ctx = c; m->width = w; m->height = ht; m->format = format;
    c->children++;
    const IlmHandle h = to_handle(m);
    if (external) {
        m->texels = external;
        m->external = true;
    } else {
        const size_t bytes = lightmap_texel_bytes(format) * (size_t)w * (size_t)ht;
        HIP_TRY_OR_DESTROY(hipMalloc(&m->texels, bytes), ilm_lightmap_destroy(h));
        HIP_TRY_OR_DESTROY(hipMemsetAsync(m->texels, 0, bytes, c->main()), ilm_lightmap_destroy(h));
    }
    *out = h;
    return ILM_OK;
}

int32_t ilm_lightmap_download(IlmHandle h, void* dst, int32_t first_row, int32_t row_count) {
    ILM_TRACE_RANGE("ilm_lightmap_download");
    Lightmap* m = from_handle<Lightmap>(h, kMagicLightmap);
    if (!m) return fail(ILM_ERR_INVALID_HANDLE, "not a lightmap handle");
    if (!dst) return fail(ILM_ERR_INVALID_ARGUMENT, "dst is NULL");
    if (first_row < 0 || row_count < 0 || first_row + row_count > m->height)
        return fail(ILM_ERR_OUT_OF_RANGE, "rows [%d, %d) outside [0, %d)", first_row, first_row + row_count, m->height);
    HIP_TRY(hipSetDevice(m->ctx->device));
    const size_t row_bytes = lightmap_texel_bytes(m->format) * (size_t)m->width;
    HIP_TRY(hipMemcpyAsync(dst, static_cast<const char*>(m->texels) + row_bytes * (size_t)first_row, row_bytes * (size_t)row_count,
                           hipMemcpyDeviceToHost, m->ctx->main()));
    HIP_TRY(hipStreamSynchronize(m->ctx->main()));
    return ILM_OK;
}

int32_t ilm_lightmap_upload(IlmHandle h, const void* src, int32_t first_row, int32_t row_count) {
    ILM_TRACE_RANGE("ilm_lightmap_upload");
    Lightmap* m = from_handle<Lightmap>(h, kMagicLightmap);
    if (!m) return fail(ILM_ERR_INVALID_HANDLE, "not a lightmap handle");
    if (!src) return fail(ILM_ERR_INVALID_ARGUMENT, "src is NULL");
    if (first_row < 0 || row_count < 0 || first_row + row_count > m->height)
        return fail(ILM_ERR_OUT_OF_RANGE, "rows [%d, %d) outside [0, %d)", first_row, first_row + row_count, m->height);
    HIP_TRY(hipSetDevice(m->ctx->device));
    const size_t row_bytes = lightmap_texel_bytes(m->format) * (size_t)m->width;
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(m->texels) + row_bytes * (size_t)first_row, src, row_bytes * (size_t)row_count,
                           hipMemcpyHostToDevice, m->ctx->main()));
    HIP_TRY(hipStreamSynchronize(m->ctx->main()));
    return ILM_OK;
}

int32_t ilm_lightmap_device_ptr(IlmHandle h, void** out_ptr) {
    Lightmap* m = from_handle<Lightmap>(h, kMagicLightmap);
    if (!m || !out_ptr) return fail(ILM_ERR_INVALID_HANDLE, "not a lightmap handle");
    *out_ptr = m->texels;
    return ILM_OK;
}

int32_t ilm_lightmap_destroy(IlmHandle h) {
    Lightmap* m = from_handle<Lightmap>(h, kMagicLightmap);
    if (!m) return fail(ILM_ERR_INVALID_HANDLE, "not a lightmap handle");
    m->ctx->children--;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->main());
    if (m->texels && !m->external) (void)hipFree(m->texels);
    if (m->d_mirrors) (void)hipFree(m->d_mirrors);
    m->ctx->mirrored.erase(std::remove(m->ctx->mirrored.begin(), m->ctx->mirrored.end(), m), m->ctx->mirrored.end());
    forget_armed(m);
    retire_handle(m);
    delete m;
    return ILM_OK;
}

namespace {
// Block -> tile mapping of the light kernel (the dispatcher places block b on XCD b % 8; every XCD has its own L2).  Measured on MI355X,
// bench.py lighting, ms / frame cfg3 | cfg5.  r01: (0) one contiguous band of tiles per XCD 2.02 | 29.0 -- each L2 stays on one part of
// the field, but the lights are not spread evenly over the screen and whole XCDs idle while the busiest band finishes; (1) tile rows
// round-robin over the XCDs 1.65 | 22.8; (2) identity, consecutive tiles on consecutive XCDs, 1.59 | 23.1 -- perfectly balanced, and an
// XCD's tiles are every eighth of a row: no two of them are neighbours.  r03: (4) square groups of M x M tiles dealt round-robin to the
// XCDs, each XCD walking its groups in turn: the tiles an XCD runs side by side lie side by side, their rays towards a light cross the
// same cells and read the same light records, and with ~100 groups per XCD the balance holds.  Identity 0.623 | 9.41; M = 2 0.624 | 9.11,
// 3 0.618 | 9.00, 4 0.632 | 9.06, 5 0.649 | 8.84, **6 0.618 | 8.78**, 7 0.630 | 8.88, 8 0.642 | 8.98, 10 0.664 | 9.19, 12 0.625 | 8.98,
// 16 0.808 | 9.14 (tools/ab_tilemap.sh; small frames lose balance as the groups grow, large ones gain locality until the groups get
// few).  Dealing a group row's groups out with the start rotated by the row (diagonal stripes) 0.636 | 8.93; whole group columns per
// XCD, walked top to bottom, 0.70 | 8.72 (cfg5's 40 columns divide by 8, cfg3's 20 do not); the eight groups in flight as a 4 x 2 block
// of groups 0.618 | 9.00 against 0.611 | 8.74 beside it; groups dealt out heaviest first (one-workgroup cost + bitonic sort) 0.618 | 8.96
// against 0.613 | 8.73; single tiles sorted heaviest first 0.70 | 9.96.  Every form of "balance first" lost to "neighbours together".
// On the eight-wave build the XCDs of a cfg3 frame end 12 % apart (tools/light_trace_probe.py), so the groups were also handed out
// dynamically -- a workgroup draws a ticket on the XCD it runs on (XCC_ID), every run of M * M tickets opens the next group off a global
// counter, idle XCDs complete the others' last runs; exact for any dispatch order, 88 light tests green -- 0.601 | 8.56 against
// 0.600 | 8.53: nothing, and taken out again.
// Default: 4 with M = 6.
// (function-local statics initialised by a lambda: thread-safe, contexts may be driven from different threads)
int light_tile_map() {
    static const int v = [] { const char* e = getenv("ILM_LIGHT_TILE_MAP"); return e ? atoi(e) : 4; }();
    return v;
}
int light_tile_macro() {
    static const int v = [] { const char* e = getenv("ILM_LIGHT_TILE_MACRO"); const int m = e ? atoi(e) : 6; return m < 1 ? 1 : m; }();
    return v;
}
int light_split_target_waves();
// ... and per launch: SHORT launches (at most ILM_LIGHT_SPLIT_WAVES tile-kernel waves: the ones that end tapered) are dealt in groups of
// 4 x 4 tiles instead of 6 x 6 -- a strip of a 4K frame is 120 groups of 6 x 6, fifteen per XCD, and the XCDs end several per cent
// apart; with 4 x 4 groups (264 of them) cfg5's eight cost-balanced strips take 10.0 ms summed instead of 10.3, 1.31 instead of 1.36
// at most (3 x 3: 10.0 | 1.33, 2 x 2: 10.0 | 1.33; r04, tools/strip_probe.py).  Whole frames keep 6 (4: 8.86 against 8.69 ms).
// ILM_LIGHT_TILE_MACRO, when set, is used for every launch.
int light_tile_macro_for(int width, int rows) {
    static const bool forced = getenv("ILM_LIGHT_TILE_MACRO") != nullptr;
    if (forced || rows <= 0 || width <= 0) return light_tile_macro();
    const int64_t waves = (int64_t)((width + kLightTile - 1) / kLightTile) * (int64_t)((rows + kLightTile - 1) / kLightTile) * (kLightTileThreads / 64);
    return (waves <= (int64_t)light_split_target_waves()) ? 4 : light_tile_macro();
}
// Light split: how many workgroups serve a tile of this launch (LightLaunch::split / taper, lighting.hip).
// A wave of the light pass lives as long as its pixels' lights take, one after the other (~0.5 ms on cfg5, ~0.15 on cfg3), and the chip
// holds 8 192 of them: a launch of only a generation or two -- one rank's strip of a frame split over 8 GPUs is 16 320 waves on cfg5, 4 080
// on cfg3 -- ends in a drain as long as a wave's life, during which the device empties (r03: the eight strips of cfg5 summed to 11.9 ms
// for an 8.55 ms frame).  With K workgroups per tile the same work is K times as many waves of 1/K the life, and the result's bits do
// not depend on K (kLightParts).  What a member costs (r04, tools/light_overhead_probe.py, tools/pmc_dispatches.sh): ~300 vector
// instructions (+1.5 % of cfg5's at K = 4) -- and a slot that issues nothing while it is launched (the dispatcher starts ~0.5 G waves/s:
// a frame of lights that touch nothing takes 0.13 ms at K = 1 and 0.5 ms at K = 8 with NO work in it), reads its arguments, bins, and
// meets the others at the ticket: whole cfg5 frames 8.8 ms at K = 1, 9.1 at 2, 10.6 at 4, 11.5 at 8.  So the split is TAPERED: the
// tiles an XCD starts first are served whole, later ones by 2, then 4, the last by 8 workgroups -- the drain is made of the shortest
// waves, the overhead is paid on the part of the launch that needs it.  On cfg5's cost-balanced strips (one GPU standing in for each
// of 8 ranks, tools/strip_probe.py): at most 1.65 ms and 12.0 ms summed untouched, 1.44 | 10.8 at K = 2 throughout, 1.36 | 10.3 tapered, 1.31 | 10.0 tapered in 4 x 4 groups.
//   chosen per launch: launches of at most one device fill split every tile (2 | 4 | 8 by halves), up to ILM_LIGHT_SPLIT_WAVES waves
//   (default 24 576, three fills) the second half tapers 2 | 4 | 8, longer ones (whole frames) are not split.
//   ILM_LIGHT_SPLIT = 1 / 2 / 4 / 8 or ilm_ctx_set_light_split force one K for every tile; ILM_LIGHT_TAPER = f1,f2,f3 forces the taper.
int light_split_env() {
    static const int v = [] { const char* e = getenv("ILM_LIGHT_SPLIT"); return e ? atoi(e) : 0; }();
    return v;
}
int light_split_target_waves() {
    static const int v = [] { const char* e = getenv("ILM_LIGHT_SPLIT_WAVES"); return e ? atoi(e) : 24576; }();
    return v;
}
// ILM_LIGHT_TAPER = "f1,f2,f3": the fractions of an XCD's tiles from which on a tile is served by 2, 4 and 8 workgroups (1 = never)
void light_taper_env(double f[3], bool* set) {
    static double v[3]; static bool have = false;
    static const bool once = [] {
        const char* e = getenv("ILM_LIGHT_TAPER");
        if (e && sscanf(e, "%lf,%lf,%lf", &v[0], &v[1], &v[2]) == 3) have = true;
        return true;
    }();
    (void)once;
    *set = have; f[0] = v[0]; f[1] = v[1]; f[2] = v[2];
}
int32_t plan_light_split(Ctx* c, LightLaunch* a) {
    a->split = 1; a->partials = nullptr; a->tickets = nullptr;
    const int rows = a->row_end - a->row_begin;
    if (rows <= 0 || a->width <= 0) return ILM_OK;
    const int slots = light_block_slots(*a);
    a->taper[0] = a->taper[1] = a->taper[2] = slots; a->taper_slots = slots;
    const int64_t tiles = (int64_t)((a->width + kLightTile - 1) / kLightTile) * (int64_t)((rows + kLightTile - 1) / kLightTile);
    // one list per tile (<= 1 024 lights; device-side counts are particle lights: thousands), and no per-light fp16 rounding chain
    if (a->blend_fp16 || a->light_count_ptr != nullptr || a->light_count > 1024 || a->light_count < 16) return ILM_OK;
    int k = c->light_split ? c->light_split : light_split_env();
    double f[3] = { 1.0, 1.0, 1.0 };
    bool taper_set = false;
    light_taper_env(f, &taper_set);
    if (k == 1) return ILM_OK;
    if (k == 2 || k == 4 || k == 8) {
        // every tile of the launch by k workgroups
        f[0] = 0.0; f[1] = (k >= 4) ? 0.0 : 1.0; f[2] = (k == 8) ? 0.0 : 1.0;
    } else if (!taper_set) {
        // chosen per launch: short launches end tapered, whole frames are left alone
        const int64_t waves = tiles * (kLightTileThreads / 64);
        if (waves > (int64_t)light_split_target_waves()) return ILM_OK;
        if (waves <= 8192) { f[0] = 0.0; f[1] = 0.5; f[2] = 0.75; }     // everything starts at once: the longest wave is the launch
        else { f[0] = 0.5; f[1] = 0.75; f[2] = 0.875; }
    }
    int t[3];
    for (int i = 0; i < 3; i++) {
        double v = f[i] < 0.0 ? 0.0 : (f[i] > 1.0 ? 1.0 : f[i]);
        t[i] = (int)(v * (double)slots + 0.5);
        if (i > 0 && t[i] < t[i - 1]) t[i] = t[i - 1];
    }
    if (t[0] >= slots) return ILM_OK;
    a->taper[0] = t[0]; a->taper[1] = t[1]; a->taper[2] = t[2];
    a->split = (t[2] < slots) ? 8 : (t[1] < slots) ? 4 : 2;
    // partial sums: one record of kLightParts x 256 float4 (32 KB) per SPLIT block slot of the launch -- (slots - taper[0]) x 8 XCDs -- not
    // per tile (a forced split on a 4K frame asked for 1 GB).  When the device cannot give it the launch is not split: the
    // frame is the same bits either way.
    const size_t split_slots = (size_t)(slots - t[0]) * 8u;
    if (split_slots > c->light_partials_tiles) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_light_partials) HIP_TRY(hipFree(c->d_light_partials));
        c->d_light_partials = nullptr; c->light_partials_tiles = 0;
        if (hipMalloc(reinterpret_cast<void**>(&c->d_light_partials), split_slots * (size_t)kLightParts * (size_t)kLightTileThreads * sizeof(float4)) != hipSuccess) {
            (void)hipGetLastError();
            c->d_light_partials = nullptr;
            a->split = 1; a->taper[0] = a->taper[1] = a->taper[2] = slots;
            return ILM_OK;
        }
        c->light_partials_tiles = split_slots;
    }
    if ((size_t)tiles > c->light_tickets_cap) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_light_tickets) HIP_TRY(hipFree(c->d_light_tickets));
        c->d_light_tickets = nullptr; c->light_tickets_cap = 0;
        // (one ticket per tile and wave: the quadrants of a tile are delivered independently)
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_light_tickets), (size_t)tiles * (kLightTileThreads / 64) * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(c->d_light_tickets, 0, (size_t)tiles * (kLightTileThreads / 64) * sizeof(uint32_t), c->main()));
        c->light_tickets_cap = (size_t)tiles;
    }
    a->partials = c->d_light_partials; a->tickets = c->d_light_tickets;
    return ILM_OK;
}

// The groups of tiles dealt out heaviest first (r04).  Cost of a group = summed area of the lights' footprint boxes inside it (host
// arithmetic on the frame's light vertices, redone only when the lights, the view or the rows change).  A whole cfg5 frame (920 groups):
// 8.85 -> 8.69 ms, three A/B pairs on one box (-1.8 %); cfg3's 240 groups and the strips of either frame do not respond, so launches
// of fewer than ILM_LIGHT_GROUP_ORDER_MIN groups (default 512) keep the row-major deal.  ILM_LIGHT_GROUP_ORDER=0 switches it off, =1 on
// for every launch.  (r03 had measured "heaviest first" as a loss: that was single TILES sorted, which tears neighbours apart; whole
// 6 x 6 groups keep the locality and only change which group an XCD takes next.)
int light_group_order_env() {
    static const int v = [] { const char* e = getenv("ILM_LIGHT_GROUP_ORDER"); return e ? atoi(e) : -1; }();
    return v;
}
int light_group_order_min() {
    static const int v = [] { const char* e = getenv("ILM_LIGHT_GROUP_ORDER_MIN"); return e ? atoi(e) : 512; }();
    return v;
}
int32_t plan_group_order(Ctx* c, LightLaunch* a, const IlmLightVertex* lights, int light_count, int group_edge_px) {
    a->group_order = nullptr;
    uint64_t pending_key = 0;
    const int mode = light_group_order_env();
    if (mode == 0 || a->tile_map != 4 || light_count <= 0) return ILM_OK;
    const int rows = a->row_end - a->row_begin;
    const int gx = (a->width + group_edge_px - 1) / group_edge_px, gy = (rows + group_edge_px - 1) / group_edge_px;
    const int groups = gx * gy;
    if (groups < 2 || groups > 65535 || (mode < 0 && groups < light_group_order_min())) return ILM_OK;
    // the same lights over the same rows through the same view: the table on the device is still right
    {
        uint64_t key = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t i = 0; i < n; i++) { key ^= b[i]; key *= 1099511628211ull; } };
        mix(lights, sizeof(IlmLightVertex) * (size_t)light_count);
        mix(&a->env, sizeof(a->env));
        const int32_t geom[5] = { a->width, a->row_begin, a->row_end, group_edge_px, light_count };
        mix(geom, sizeof(geom));
        if (c->d_group_order && c->group_order_key == key && c->group_order_groups == groups) { a->group_order = c->d_group_order; return ILM_OK; }
        pending_key = key;       // committed below, once the table IS on the device (a failed upload left the key of a table that was not there)
        c->group_order_groups = 0;
    }
    std::vector<double> cost((size_t)groups, 0.0);
    const float sx = a->env.GBufferTexelSizeAndMisc.z * a->env.ZAndScale.z, sy = a->env.GBufferTexelSizeAndMisc.w * a->env.ZAndScale.w;
    for (int i = 0; i < light_count; i++) {
        const IlmLightVertex& L = lights[i];
        const double reach = (double)L.LightProperties.x + (double)L.LightProperties.y + 1.0;
        const double x0 = ((double)L.LightPosition1.x - reach - a->env.ViewportPosition[0]) * sx, x1 = ((double)L.LightPosition1.x + reach - a->env.ViewportPosition[0]) * sx;
        const double y0 = ((double)L.LightPosition1.y - reach - a->env.ViewportPosition[1]) * sy - a->row_begin, y1 = ((double)L.LightPosition1.y + reach - a->env.ViewportPosition[1]) * sy - a->row_begin;
        if (!(x1 > x0) || !(y1 > y0)) continue;
        // (clamped to the launch's rectangle in double BEFORE the conversion: a huge or infinite radius must not reach the int cast)
        const double cx0 = std::min(std::max(x0, 0.0), (double)a->width), cx1 = std::min(std::max(x1, 0.0), (double)a->width);
        const double cy0 = std::min(std::max(y0, 0.0), (double)rows), cy1 = std::min(std::max(y1, 0.0), (double)rows);
        if (!(cx1 > cx0) || !(cy1 > cy0)) continue;
        const int ga = std::max(0, (int)std::floor(cx0 / group_edge_px)), gb = std::min(gx - 1, (int)std::floor(cx1 / group_edge_px));
        const int gc = std::max(0, (int)std::floor(cy0 / group_edge_px)), gd = std::min(gy - 1, (int)std::floor(cy1 / group_edge_px));
        for (int yy = gc; yy <= gd; yy++)
            for (int xx = ga; xx <= gb; xx++) {
                const double w = std::min(x1, (double)std::min((xx + 1) * group_edge_px, a->width)) - std::max(x0, (double)(xx * group_edge_px));
                const double h = std::min(y1, (double)std::min((yy + 1) * group_edge_px, rows)) - std::max(y0, (double)(yy * group_edge_px));
                if (w > 0 && h > 0) cost[(size_t)(yy * gx + xx)] += w * h;
            }
    }
    std::vector<uint16_t> order((size_t)groups);
    for (int g = 0; g < groups; g++) order[(size_t)g] = (uint16_t)g;
    std::stable_sort(order.begin(), order.end(), [&](uint16_t l, uint16_t r) { return cost[l] > cost[r]; });
    // dealt to the XCDs in a snake (0..7, 7..0): position p of the dealing order is read by XCD p % 8 as its (p / 8)-th group
    std::vector<uint16_t> dealt((size_t)groups);
    for (int p = 0; p < groups; p++) {
        const int round = p / 8, slot = p % 8;
        const int src = round * 8 + ((round & 1) ? 7 - slot : slot);
        dealt[(size_t)p] = order[(size_t)std::min(src, groups - 1)];
    }
    // (the last, partial round of a snake may name a group twice and skip another: repair by a plain copy of that round)
    { const int tail = (groups / 8) * 8; for (int p = tail; p < groups; p++) dealt[(size_t)p] = order[(size_t)p]; }
    if (groups > c->group_order_cap) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_group_order) HIP_TRY(hipFree(c->d_group_order));
        c->d_group_order = nullptr; c->group_order_cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_group_order), sizeof(uint16_t) * (size_t)groups * 2));
        c->group_order_cap = groups * 2;
    }
    const int32_t rc = upload_small(c, c->d_group_order, dealt.data(), sizeof(uint16_t) * (size_t)groups);
    if (rc != ILM_OK) return rc;
    c->group_order_key = pending_key; c->group_order_groups = groups;
    a->group_order = c->d_group_order;
    return ILM_OK;
}

// shared by the three light passes: resource checks + the launch descriptor
int32_t fill_light_launch(Ctx* c, const IlmEnvironment* env, const IlmDistanceFieldUniforms* df, IlmHandle hgbuffer, IlmHandle hsdf,
                          IlmHandle hlightmap, int32_t row_begin, int32_t row_end, LightLaunch* a) {
    Lightmap* m = from_handle<Lightmap>(hlightmap, kMagicLightmap);
    if (!m) return fail(ILM_ERR_INVALID_HANDLE, "not a lightmap handle");
    GBuffer* g = nullptr; Sdf* f = nullptr;
    if (hgbuffer) { g = from_handle<GBuffer>(hgbuffer, kMagicGBuffer); if (!g) return fail(ILM_ERR_INVALID_HANDLE, "not a G-buffer handle"); }
    if (hsdf) { f = from_handle<Sdf>(hsdf, kMagicSdf); if (!f) return fail(ILM_ERR_INVALID_HANDLE, "not a distance field handle"); }
    if (!env || !df) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    // (the lightmap is written: the context's own; the field and the G-buffer are read: the context's or a sibling's)
    if (m->ctx != c || (g && !siblings(g->ctx, c)) || (f && !siblings(f->ctx, c))) return fail(ILM_ERR_INVALID_ARGUMENT, "resources belong to another context");
    if (row_begin < 0 || row_end > m->height || row_begin > row_end)
        return fail(ILM_ERR_OUT_OF_RANGE, "rows [%d, %d) outside [0, %d]", row_begin, row_end, m->height);
    { char why[256]; if (field_uniforms_mismatch(f, df, why, sizeof(why))) return fail(ILM_ERR_INVALID_ARGUMENT, "%s", why); }
    a->lights = nullptr; a->light_count = 0;
    a->env = *env; a->df = *df;
    a->gbuffer.texels = g ? g->texels : nullptr;
    a->gbuffer.width = g ? g->width : 0; a->gbuffer.height = g ? g->height : 0; a->gbuffer.format = g ? g->format : 0;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(borrowed_trace_view(c, f, g, df, &a->sdf));
    for (int i = 0; i < 4; i++) a->ambient[i] = 0.0f;
    a->lightmap = m->texels; a->width = m->width; a->height = m->height; a->format = m->format;
    a->row_begin = row_begin; a->row_end = row_end;
    a->stats = nullptr; a->light_count_ptr = nullptr; a->accumulate = 0;
    a->ramp = RampView{ nullptr, 0, 0 };      // particle lights have no ramp technique (LightingRenderer.cs:176-178)
    a->blend_fp16 = (c->lightmap_blend == ILM_BLEND_FP16_PER_LIGHT) ? 1 : 0;
    a->tile_map = light_tile_map();
    a->tile_macro = light_tile_macro_for(m->width, row_end - row_begin);
    a->split = 1; a->partials = nullptr; a->tickets = nullptr; a->group_order = nullptr;
    { const int32_t rc = set_launch_mirrors(m, a); if (rc != ILM_OK) return rc; }
    return ILM_OK;
}
}  // namespace


int32_t ilm_render_particle_lights(IlmHandle hctx, IlmHandle hsystem, const int32_t* quad_counts, int32_t chunk_count,
                                   const IlmParticleLightParams* params, const IlmEnvironment* env, const IlmDistanceFieldUniforms* df,
                                   IlmHandle hgbuffer, IlmHandle hsdf, IlmHandle hlightmap, int32_t row_begin, int32_t row_end,
                                   IlmRenderStats* stats) {
    ILM_TRACE_RANGE("ilm_render_particle_lights");
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    System* s = from_handle<System>(hsystem, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (s->engine->ctx != c) return fail(ILM_ERR_INVALID_ARGUMENT, "the particle system belongs to another context");
    if (!params) return fail(ILM_ERR_INVALID_ARGUMENT, "params is NULL");
    // StippleReject (Fracture DitherCommon.fxh) is not in the reference tree: only "reject nothing" is defined here
    if (!(params->StippleFactor >= 1.0f))
        return fail(ILM_ERR_INVALID_ARGUMENT, "StippleFactor %g < 1 needs Fracture's StippleReject, which is outside the reference tree", (double)params->StippleFactor);
    const int n = (int)s->chunks.size();
    if (chunk_count < 0 || chunk_count > n) return fail(ILM_ERR_OUT_OF_RANGE, "chunk_count %d outside [0, %d]", chunk_count, n);
    LightLaunch a = {};
    int32_t rc = fill_light_launch(c, env, df, hgbuffer, hsdf, hlightmap, row_begin, row_end, &a);
    if (rc != ILM_OK) return rc;
    if (chunk_count == 0) return ILM_OK;
    Engine* e = s->engine;
    HIP_TRY(hipSetDevice(c->device));
    rc = refresh_table(s);
    if (rc != ILM_OK) return rc;

    int64_t total = 0;
    for (int i = 0; i < chunk_count; i++) {
        const int q = quad_counts ? quad_counts[i] : e->slots;
        if (q < 0 || q > e->slots) return fail(ILM_ERR_OUT_OF_RANGE, "quad_counts[%d] = %d outside [0, %d]", i, q, e->slots);
        total += q;
    }
    if (total > (1 << 22)) return fail(ILM_ERR_TOO_MANY, "%lld particle lights in one call (limit %d)", (long long)total, 1 << 22);
    if (total == 0) return ILM_OK;
    const int blocks_per_chunk = (e->slots + 1023) / 1024;
    const int blocks = chunk_count * blocks_per_chunk;
    if ((int)total > c->pl_cap) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_pl_recs) HIP_TRY(hipFree(c->d_pl_recs));
        c->d_pl_recs = nullptr; c->pl_cap = 0;
        const int cap = (int)total < 4096 ? 4096 : (int)total;
        HIP_TRY(hipMalloc(&c->d_pl_recs, kLightRecBytes * (size_t)cap));
        c->pl_cap = cap;
    }
    if (!c->d_pl_count) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_pl_count), sizeof(int32_t)));
    if (blocks > c->pl_blocks_cap) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_pl_blocks) HIP_TRY(hipFree(c->d_pl_blocks));
        c->d_pl_blocks = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_pl_blocks), sizeof(int32_t) * (size_t)blocks * 2));
        c->pl_blocks_cap = blocks * 2;
    }
    if (chunk_count > c->pl_quads_cap) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_pl_quads) HIP_TRY(hipFree(c->d_pl_quads));
        c->d_pl_quads = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_pl_quads), sizeof(int32_t) * (size_t)chunk_count * 2));
        c->pl_quads_cap = chunk_count * 2;
    }
    if (quad_counts) {
        rc = upload_small(c, c->d_pl_quads, quad_counts, sizeof(int32_t) * (size_t)chunk_count);
        if (rc != ILM_OK) return rc;
    }
    ParticleLightLaunch pl;
    pl.chunk_bases = s->d_table; pl.stride = e->stride; pl.chunk_count = chunk_count; pl.slots = e->slots;
    pl.quad_counts = quad_counts ? c->d_pl_quads : nullptr;
    pl.params = *params; pl.env = *env; pl.max_cone_radius = df->ConeAndMisc.x;
    pl.gate = make_trace_gate(*df, a.sdf);
    pl.block_counts = c->d_pl_blocks; pl.recs = c->d_pl_recs; pl.capacity = c->pl_cap; pl.out_count = c->d_pl_count;
    HIP_TRY(launch_prepare_particle_lights(pl, c->main()));

    a.light_count = 0;
    a.light_count_ptr = c->d_pl_count;
    a.accumulate = 1;
    if (stats) {
        HIP_TRY(hipMemsetAsync(c->d_stats, 0, 3 * sizeof(unsigned long long), c->main()));
        a.stats = c->d_stats;
    }
    c->last_light_blocks = light_launch_blocks(a); c->last_light_split = a.split; c->last_light_macro = (a.tile_map == 4) ? a.tile_macro : 0;
    HIP_TRY(launch_sphere_lights_prepared(a, c->d_pl_recs, c->main()));
    HIP_TRY(light_pass_queued(c, hsdf ? from_handle<Sdf>(hsdf, kMagicSdf) : nullptr, hgbuffer ? from_handle<GBuffer>(hgbuffer, kMagicGBuffer) : nullptr));
    if (stats) {
        unsigned long long host[3] = { 0, 0, 0 };
        HIP_TRY(hipMemcpyAsync(host, c->d_stats, sizeof(host), hipMemcpyDeviceToHost, c->main()));
        HIP_TRY(hipStreamSynchronize(c->main()));
        stats->SdfSamples = host[0]; stats->PixelLightPairs = host[1]; stats->TracedPairs = host[2];
    }
    return ILM_OK;
}

int32_t ilm_render_light_probes(IlmHandle hctx, const IlmLightVertex* lights, int32_t light_count,
                                const IlmFloat4* probe_positions, const IlmFloat4* probe_normals, int32_t probe_count,
                                const IlmEnvironment* env, const IlmDistanceFieldUniforms* df, IlmHandle hsdf, IlmFloat4* out_values) {
    ILM_TRACE_RANGE("ilm_render_light_probes");
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    Sdf* f = nullptr;
    if (hsdf) { f = from_handle<Sdf>(hsdf, kMagicSdf); if (!f) return fail(ILM_ERR_INVALID_HANDLE, "not a distance field handle"); }
    if (f && f->ctx != c) return fail(ILM_ERR_INVALID_ARGUMENT, "resources belong to another context");
    if (!env || !df) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    if (light_count < 0 || (light_count > 0 && !lights)) return fail(ILM_ERR_INVALID_ARGUMENT, "bad light array");
    if (probe_count < 0 || (probe_count > 0 && (!probe_positions || !probe_normals || !out_values))) return fail(ILM_ERR_INVALID_ARGUMENT, "bad probe arrays");
    { char why[256]; if (field_uniforms_mismatch(f, df, why, sizeof(why))) return fail(ILM_ERR_INVALID_ARGUMENT, "%s", why); }
    if (probe_count == 0) return ILM_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (light_count > c->light_cap) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_lights) HIP_TRY(hipFree(c->d_lights));
        if (c->d_recs) HIP_TRY(hipFree(c->d_recs));
        c->d_lights = nullptr; c->d_recs = nullptr;
        int cap = light_count < 256 ? 256 : light_count * 2;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_lights), sizeof(IlmLightVertex) * (size_t)cap));
        HIP_TRY(hipMalloc(&c->d_recs, kLightRecBytes * (size_t)cap));
        c->light_cap = cap;
    }
    // One pinned block [lights | positions | normals | values]: the kernels read their inputs where it lies (each word once) and the
    // last one writes the values into it -- no copy command on either side of the two kernels (r04; three uploads and a read-back before)
    auto align64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t off_pos = align64(sizeof(IlmLightVertex) * (size_t)light_count);
    const size_t off_nrm = off_pos + align64(sizeof(float4) * (size_t)probe_count);
    const size_t off_val = off_nrm + align64(sizeof(float4) * (size_t)probe_count);
    const size_t block_bytes = off_val + sizeof(float4) * (size_t)probe_count;
    unsigned char* block = nullptr;
    int slot = -1;
    int32_t rc = upload_small_begin(c, block_bytes, reinterpret_cast<void**>(&block), &slot);
    if (rc != ILM_OK) return rc;
    if (light_count > 0) memcpy(block, lights, sizeof(IlmLightVertex) * (size_t)light_count);
    memcpy(block + off_pos, probe_positions, sizeof(float4) * (size_t)probe_count);
    memcpy(block + off_nrm, probe_normals, sizeof(float4) * (size_t)probe_count);
    void* dv = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dv, block, 0));
    const char* in = static_cast<const char*>(dv);
    if (light_count > 0)
        HIP_TRY(launch_prepare_lights(reinterpret_cast<const IlmLightVertex*>(in), light_count, *env, *df, make_sdf_view(f, df), c->d_recs, c->main()));
    const float4* d_pos = reinterpret_cast<const float4*>(in + off_pos);
    const float4* d_nrm = reinterpret_cast<const float4*>(in + off_nrm);
    float4* d_val = reinterpret_cast<float4*>(const_cast<char*>(in) + off_val);
    // scratch for one contribution per (light, probe): the pairs are shaded side by side and added per probe in light order
    float4* d_pairs = nullptr;
    const size_t pair_count = (size_t)probe_count * (size_t)(light_count > 0 ? light_count : 0);
    if (light_count > 1 && pair_count * sizeof(float4) <= ((size_t)256 << 20)) {
        if (pair_count > c->probe_pairs_cap) {
            HIP_TRY(hipStreamSynchronize(c->main()));
            if (c->d_probe_pairs) HIP_TRY(hipFree(c->d_probe_pairs));
            c->d_probe_pairs = nullptr; c->probe_pairs_cap = 0;
            const size_t cap = pair_count < 65536 ? 65536 : pair_count + pair_count / 2;
            if (hipMalloc(reinterpret_cast<void**>(&c->d_probe_pairs), sizeof(float4) * cap) == hipSuccess) c->probe_pairs_cap = cap;
            else { (void)hipGetLastError(); c->d_probe_pairs = nullptr; }      // (the one-kernel form needs no scratch)
        }
        d_pairs = c->d_probe_pairs;
    }
    HIP_TRY(launch_light_probes(c->d_recs, light_count, d_pos, d_nrm, probe_count, *env, *df, make_sdf_view(f, df),
                                RampView{ c->d_light_ramp, c->light_ramp_w, c->light_ramp_h }, d_val, d_pairs, c->main()));
    rc = staged_small_done(c, slot);
    if (rc != ILM_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->main()));
    memcpy(out_values, block + off_val, sizeof(float4) * (size_t)probe_count);
    return ILM_OK;
}

// Compacts the live particles of the first chunk_count chunks into draw-call records in c->d_rb (device) and copies the first
// min(total, capacity) of them into the context's pinned host buffer; *out_total = live particles found.
static int32_t readback_to_pinned(System* s, const int32_t* element_counts, int32_t chunk_count, const IlmReadbackParams* params,
                                  int32_t capacity, int32_t* out_total) {
    Engine* e = s->engine;
    Ctx* c = e->ctx;
    for (int i = 0; i < chunk_count; i++)
        if (element_counts && (element_counts[i] < 0 || element_counts[i] > e->slots))
            return fail(ILM_ERR_OUT_OF_RANGE, "element_counts[%d] = %d outside [0, %d]", i, element_counts[i], e->slots);
    HIP_TRY(hipSetDevice(c->device));
    int32_t rc = refresh_table(s);
    if (rc != ILM_OK) return rc;
    const int blocks = chunk_count * ((e->slots + 1023) / 1024);
    if (capacity > c->rb_cap) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_rb) HIP_TRY(hipFree(c->d_rb));
        c->d_rb = nullptr; c->rb_cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_rb), sizeof(IlmReadbackDrawCall) * (size_t)capacity));
        c->rb_cap = capacity;
    }
    if (!c->d_rb_count) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_rb_count), sizeof(int32_t)));
    if (blocks > c->rb_blocks_cap) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_rb_blocks) HIP_TRY(hipFree(c->d_rb_blocks));
        c->d_rb_blocks = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_rb_blocks), sizeof(int32_t) * (size_t)blocks * 2));
        c->rb_blocks_cap = blocks * 2;
    }
    if (chunk_count > c->rb_elems_cap) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_rb_elems) HIP_TRY(hipFree(c->d_rb_elems));
        c->d_rb_elems = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_rb_elems), sizeof(int32_t) * (size_t)chunk_count * 2));
        c->rb_elems_cap = chunk_count * 2;
    }
    if (element_counts) {
        rc = upload_small(c, c->d_rb_elems, element_counts, sizeof(int32_t) * (size_t)chunk_count);
        if (rc != ILM_OK) return rc;
    }
    ReadbackLaunch a;
    a.chunk_bases = s->d_table; a.stride = e->stride; a.chunk_count = chunk_count; a.slots = e->slots;
    a.element_counts = element_counts ? c->d_rb_elems : nullptr;
    a.params = *params;
    // ParticleReadback.cs:100-112
    a.region_w = params->TextureRegion[2] - params->TextureRegion[0];
    a.region_h = params->TextureRegion[3] - params->TextureRegion[1];
    a.frame_count_x = std::max((int)(1.0f / a.region_w), 1);
    a.frame_count_y = std::max((int)(1.0f / a.region_h), 1);
    a.max_angle_x = (2 * 3.14159265358979323846) / a.frame_count_x;
    a.max_angle_y = (2 * 3.14159265358979323846) / a.frame_count_y;
    a.block_counts = c->d_rb_blocks; a.out = c->d_rb; a.capacity = capacity; a.out_count = c->d_rb_count;
    HIP_TRY(launch_readback(a, c->main()));
    int32_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, c->d_rb_count, sizeof(int32_t), hipMemcpyDeviceToHost, c->main()));
    HIP_TRY(hipStreamSynchronize(c->main()));
    *out_total = total;
    const int n_copy = total < capacity ? total : capacity;
    if (n_copy > 0) {
        // pinned destination: the records cross PCIe at DMA rate (a pageable destination goes through a bounce buffer and faults
        // its pages in on first touch: 62 MB of records took 30 ms that way)
        if ((size_t)n_copy > c->h_rb_cap) {
            if (c->h_rb) HIP_TRY(hipHostFree(c->h_rb));
            c->h_rb = nullptr; c->h_rb_cap = 0;
            const size_t cap = (size_t)n_copy + (size_t)n_copy / 4 + 1024;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_rb), sizeof(IlmReadbackDrawCall) * cap, hipHostMallocDefault));
            c->h_rb_cap = cap;
        }
        HIP_TRY(hipMemcpyAsync(c->h_rb, c->d_rb, sizeof(IlmReadbackDrawCall) * (size_t)n_copy, hipMemcpyDeviceToHost, c->main()));
        HIP_TRY(hipStreamSynchronize(c->main()));
    }
    return ILM_OK;
}

static int32_t readback_capacity(const System* s, const int32_t* element_counts, int32_t chunk_count) {
    long long total = 0;
    for (int i = 0; i < chunk_count; i++) total += element_counts ? (long long)std::max(element_counts[i], 0) : (long long)s->engine->slots;
    return (int32_t)std::min<long long>(std::max<long long>(total, 1), INT32_MAX);
}

int32_t ilm_system_readback_view(IlmHandle hsystem, const int32_t* element_counts, int32_t chunk_count, const IlmReadbackParams* params,
                                 const IlmReadbackDrawCall** out_records, int32_t* out_count) {
    ILM_TRACE_RANGE("ilm_system_readback_view");
    System* s = from_handle<System>(hsystem, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!params || !out_count || !out_records) return fail(ILM_ERR_INVALID_ARGUMENT, "bad arguments");
    *out_count = 0; *out_records = nullptr;
    const int n = (int)s->chunks.size();
    if (chunk_count < 0 || chunk_count > n) return fail(ILM_ERR_OUT_OF_RANGE, "chunk_count %d outside [0, %d]", chunk_count, n);
    if (chunk_count == 0) return ILM_OK;
    int32_t total = 0;
    const int32_t rc = readback_to_pinned(s, element_counts, chunk_count, params, readback_capacity(s, element_counts, chunk_count), &total);
    if (rc != ILM_OK) return rc;
    *out_count = total;
    *out_records = (total > 0) ? s->engine->ctx->h_rb : nullptr;
    return ILM_OK;
}

int32_t ilm_system_readback(IlmHandle hsystem, const int32_t* element_counts, int32_t chunk_count, const IlmReadbackParams* params,
                            IlmReadbackDrawCall* out, int32_t capacity, int32_t* out_count) {
    ILM_TRACE_RANGE("ilm_system_readback");
    System* s = from_handle<System>(hsystem, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (!params || !out_count || capacity < 0 || (capacity > 0 && !out)) return fail(ILM_ERR_INVALID_ARGUMENT, "bad arguments");
    *out_count = 0;
    const int n = (int)s->chunks.size();
    if (chunk_count < 0 || chunk_count > n) return fail(ILM_ERR_OUT_OF_RANGE, "chunk_count %d outside [0, %d]", chunk_count, n);
    if (chunk_count == 0) return ILM_OK;
    int32_t total = 0;
    const int32_t rc = readback_to_pinned(s, element_counts, chunk_count, params, capacity, &total);
    if (rc != ILM_OK) return rc;
    *out_count = total;
    const int n_copy = total < capacity ? total : capacity;
    if (n_copy > 0)
        std::memcpy(out, s->engine->ctx->h_rb, sizeof(IlmReadbackDrawCall) * (size_t)n_copy);
    return ILM_OK;
}

int32_t ilm_system_set_bitmap(IlmHandle h, const IlmFloat4* texels, int32_t width, int32_t height) {
    System* s = from_handle<System>(h, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    if (width < 0 || height < 0 || width > 16384 || height > 16384 || ((width > 0) != (height > 0)) || (width > 0 && !texels))
        return fail(ILM_ERR_INVALID_ARGUMENT, "bad bitmap (%d x %d)", width, height);
    Ctx* c = s->engine->ctx;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->main()));     // an earlier render may still read the old bitmap
    if (s->bitmap) HIP_TRY(hipFree(s->bitmap));
    s->bitmap = nullptr; s->bitmap_w = s->bitmap_h = 0;
    if (width == 0) return ILM_OK;
    const size_t bytes = sizeof(float4) * (size_t)width * (size_t)height;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->bitmap), bytes));
    HIP_TRY(hipMemcpy(s->bitmap, texels, bytes, hipMemcpyHostToDevice));
    s->bitmap_w = width; s->bitmap_h = height;
    return ILM_OK;
}

int32_t ilm_ctx_set_lightmap_blend(IlmHandle hctx, int32_t mode) {
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    if (mode != ILM_BLEND_FP32_ACCUMULATE && mode != ILM_BLEND_FP16_PER_LIGHT) return fail(ILM_ERR_INVALID_ARGUMENT, "unknown lightmap blend mode %d", mode);
    c->lightmap_blend = mode;
    return ILM_OK;
}

int32_t ilm_ctx_set_light_split(IlmHandle hctx, int32_t workgroups) {
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    if (workgroups != 0 && workgroups != 1 && workgroups != 2 && workgroups != 4 && workgroups != 8)
        return fail(ILM_ERR_INVALID_ARGUMENT, "light split %d is not 0 (automatic), 1, 2, 4 or 8", workgroups);
    c->light_split = workgroups;
    return ILM_OK;
}

int32_t ilm_ctx_set_light_ramp(IlmHandle hctx, const IlmFloat4* texels, int32_t width, int32_t height) {
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    if (width < 0 || height < 0 || width > 16384 || height > 16384 || ((width > 0) != (height > 0)) || (width > 0 && !texels))
        return fail(ILM_ERR_INVALID_ARGUMENT, "bad ramp texture (%d x %d)", width, height);
    const bool none = (width == 0) || (width == 1 && height == 1);
    if (none && !c->d_light_ramp) return ILM_OK;  // nothing bound, nothing to unbind: no synchronisation
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->main()));     // an earlier pass may still read the old ramp
    if (c->d_light_ramp) HIP_TRY(hipFree(c->d_light_ramp));
    c->d_light_ramp = nullptr; c->light_ramp_w = c->light_ramp_h = 0;
    // a 1 x 1 ramp is no ramp (LightingRenderer.cs:822-827)
    if (width == 0 || (width == 1 && height == 1)) return ILM_OK;
    const size_t bytes = sizeof(float4) * (size_t)width * (size_t)height;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_light_ramp), bytes));
    HIP_TRY(hipMemcpy(c->d_light_ramp, texels, bytes, hipMemcpyHostToDevice));
    c->light_ramp_w = width; c->light_ramp_h = height;
    return ILM_OK;
}

int32_t ilm_lightmap_clear(IlmHandle h, const float rgba[4]) {
    ILM_TRACE_RANGE("ilm_lightmap_clear");
    Lightmap* m = from_handle<Lightmap>(h, kMagicLightmap);
    if (!m) return fail(ILM_ERR_INVALID_HANDLE, "not a lightmap handle");
    if (!rgba) return fail(ILM_ERR_INVALID_ARGUMENT, "rgba is NULL");
    HIP_TRY(hipSetDevice(m->ctx->device));
    HIP_TRY(launch_clear_target(m->texels, m->format, (size_t)m->width * (size_t)m->height, make_float4(rgba[0], rgba[1], rgba[2], rgba[3]), m->ctx->main()));
    return ILM_OK;
}

int32_t ilm_render_particles(IlmHandle hsystem, const int32_t* quad_counts, int32_t chunk_count, const IlmRasterizeParams* params,
                             IlmHandle htarget, uint64_t* out_stats) {
    ILM_TRACE_RANGE("ilm_render_particles");
    System* s = from_handle<System>(hsystem, kMagicSystem);
    if (!s) return fail(ILM_ERR_INVALID_HANDLE, "not a system handle");
    Lightmap* m = from_handle<Lightmap>(htarget, kMagicLightmap);
    if (!m) return fail(ILM_ERR_INVALID_HANDLE, "target is not a lightmap handle");
    if (!params) return fail(ILM_ERR_INVALID_ARGUMENT, "params is NULL");
    Engine* e = s->engine;
    Ctx* c = e->ctx;
    if (m->ctx != c) return fail(ILM_ERR_INVALID_ARGUMENT, "system and target belong to different contexts");
    // Fracture code outside the reference tree (DitherCommon.fxh): not guessed
    if (!(params->StippleFactor >= 1.0f))
        return fail(ILM_ERR_INVALID_ARGUMENT, "StippleFactor %g < 1 needs Fracture's StippleReject, which is not part of the reference tree", (double)params->StippleFactor);
    if (params->BlendMode != ILM_BLEND_ALPHA && params->BlendMode != ILM_BLEND_ADDITIVE)
        return fail(ILM_ERR_INVALID_ARGUMENT, "unknown blend mode %d", params->BlendMode);
    if (params->BitmapFilter < ILM_BITMAP_NONE || params->BitmapFilter > ILM_BITMAP_LINEAR)
        return fail(ILM_ERR_INVALID_ARGUMENT, "unknown bitmap filter %d", params->BitmapFilter);
    if (params->BitmapFilter != ILM_BITMAP_NONE) {
        if (!s->bitmap) return fail(ILM_ERR_STATE, "textured technique without a bitmap (ilm_system_set_bitmap)");
        const float rw = params->BitmapTextureRegion.z - params->BitmapTextureRegion.x, rh = params->BitmapTextureRegion.w - params->BitmapTextureRegion.y;
        if (!(rw > 0.0f) || !(rh > 0.0f)) return fail(ILM_ERR_INVALID_ARGUMENT, "empty BitmapTextureRegion");
    }
    const int n = (int)s->chunks.size();
    if (chunk_count < 0 || chunk_count > n) return fail(ILM_ERR_OUT_OF_RANGE, "chunk_count %d outside [0, %d]", chunk_count, n);
    if (out_stats) out_stats[0] = out_stats[1] = out_stats[2] = 0;
    if (chunk_count == 0) return ILM_OK;
    for (int i = 0; i < chunk_count; i++) {
        const int q = quad_counts ? quad_counts[i] : e->slots;
        if (q < 0 || q > e->slots) return fail(ILM_ERR_OUT_OF_RANGE, "quad_counts[%d] = %d outside [0, %d]", i, q, e->slots);
    }
    if ((long long)chunk_count * (long long)e->slots > (long long)INT32_MAX)
        return fail(ILM_ERR_TOO_MANY, "%d chunks of %d slots exceed the 32-bit slot index of the sort key", chunk_count, e->slots);
    if ((m->width + 15) / 16 > 65535 || (m->height + 15) / 16 > 65535) return fail(ILM_ERR_OUT_OF_RANGE, "target too large");
    HIP_TRY(hipSetDevice(c->device));
    int32_t rc = refresh_table(s);
    if (rc != ILM_OK) return rc;
    if (quad_counts) {
        if (chunk_count > c->raster_quads_cap) {
            HIP_TRY(hipStreamSynchronize(c->main()));
            if (c->d_raster_quads) HIP_TRY(hipFree(c->d_raster_quads));
            c->d_raster_quads = nullptr; c->raster_quads_cap = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_raster_quads), sizeof(int32_t) * (size_t)chunk_count * 2));
            c->raster_quads_cap = chunk_count * 2;
        }
        rc = upload_small(c, c->d_raster_quads, quad_counts, sizeof(int32_t) * (size_t)chunk_count);
        if (rc != ILM_OK) return rc;
    }
    RasterLaunch a;
    std::memset(&a, 0, sizeof(a));
    a.chunk_bases = s->d_table; a.stride = e->stride; a.chunk_count = chunk_count; a.slots = e->slots;
    a.total_slots = chunk_count * e->slots;
    a.quad_counts = quad_counts ? c->d_raster_quads : nullptr;
    a.params = *params;
    a.target = m->texels; a.format = m->format; a.width = m->width; a.height = m->height;
    a.tiles_x = (m->width + 15) / 16; a.tiles_y = (m->height + 15) / 16;
    a.count_shaded = out_stats ? 1 : 0;
    a.bitmap = s->bitmap; a.bitmap_w = s->bitmap_w; a.bitmap_h = s->bitmap_h;
    unsigned long long stats[3] = { 0, 0, 0 };
    bool too_many = false;
    HIP_TRY(render_particles(a, c->raster, c->main(), out_stats ? stats : nullptr, &too_many));
    if (too_many) return fail(ILM_ERR_TOO_MANY, "more than 2^28 (quad, tile) pairs: the quads are too large for this path");
    if (out_stats) { out_stats[0] = stats[0]; out_stats[1] = stats[1]; out_stats[2] = stats[2]; }
    return ILM_OK;
}

int32_t ilm_resolve_lighting(IlmHandle hsrc, IlmHandle hdst, const IlmHDRConfiguration* hdr, int32_t row_begin, int32_t row_end) {
    ILM_TRACE_RANGE("ilm_resolve_lighting");
    return ilm_resolve_lighting_with_albedo(hsrc, 0, hdst, hdr, row_begin, row_end);
}

int32_t ilm_resolve_lighting_with_albedo(IlmHandle hsrc, IlmHandle halbedo, IlmHandle hdst, const IlmHDRConfiguration* hdr, int32_t row_begin, int32_t row_end) {
    ILM_TRACE_RANGE("ilm_resolve_lighting_with_albedo");
    Lightmap* src = from_handle<Lightmap>(hsrc, kMagicLightmap);
    Lightmap* dst = from_handle<Lightmap>(hdst, kMagicLightmap);
    if (!src || !dst) return fail(ILM_ERR_INVALID_HANDLE, "not a lightmap handle");
    Lightmap* albedo = nullptr;
    if (halbedo) {
        albedo = from_handle<Lightmap>(halbedo, kMagicLightmap);
        if (!albedo) return fail(ILM_ERR_INVALID_HANDLE, "albedo is not a texture (lightmap) handle");
        if (albedo->ctx != src->ctx || albedo->width != src->width)
            return fail(ILM_ERR_INVALID_ARGUMENT, "the albedo texture must share context and width with the lightmap (the resolve is texel for texel)");
        if (hdr && hdr->AlbedoIsSRGB != 0)
            return fail(ILM_ERR_INVALID_ARGUMENT, "AlbedoIsSRGB needs Fracture's pSRGBToPLinear (sRGBCommon.fxh), which is outside the reference tree");
    }
    if (!hdr) return fail(ILM_ERR_INVALID_ARGUMENT, "hdr is NULL");
    // Heights may differ: a group member's lightmap carries padding rows below the frame (world x slot rows, group.hip), the host's back
    // buffer and albedo are frame-sized.  The rows resolved must exist in all of them.
    if (src->ctx != dst->ctx || src->width != dst->width)
        return fail(ILM_ERR_INVALID_ARGUMENT, "source and destination must share context and width");
    int common_height = src->height < dst->height ? src->height : dst->height;
    if (albedo && albedo->height < common_height) common_height = albedo->height;
    if (hdr->Mode < ILM_HDR_NONE || hdr->Mode > ILM_HDR_TONE_MAP) return fail(ILM_ERR_INVALID_ARGUMENT, "unknown HDR mode %d", hdr->Mode);
    if (hdr->ResolveToSRGB != 0)
        return fail(ILM_ERR_INVALID_ARGUMENT, "ResolveToSRGB needs Fracture's pLinearToPSRGB (sRGBCommon.fxh), which is outside the reference tree");
    if (hdr->DitheringStrength != 0)
        return fail(ILM_ERR_INVALID_ARGUMENT, "dithering needs Fracture's ApplyDither (DitherCommon.fxh), which is outside the reference tree");
    if (row_begin < 0 || row_end > common_height || row_begin > row_end)
        return fail(ILM_ERR_OUT_OF_RANGE, "rows [%d, %d) outside [0, %d] (the smallest of the textures' heights)", row_begin, row_end, common_height);
    Ctx* c = src->ctx;
    HIP_TRY(hipSetDevice(c->device));
    // clamps of SetGammaCompressionParameters / SetToneMappingParameters, IlluminantMaterials.cs:81-137
    const float min_v = 1.0f / 256.0f, max_v = 99999.0f;
    auto clamp = [](float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); };
    ResolveLaunch a;
    a.src = src->texels; a.src_format = src->format; a.dst = dst->texels; a.dst_format = dst->format;
    a.albedo = albedo ? albedo->texels : nullptr; a.albedo_format = albedo ? albedo->format : 0;
    a.width = src->width; a.row_begin = row_begin; a.row_end = row_end; a.mode = hdr->Mode;
    a.inverse_scale = (hdr->InverseScaleFactor != 0.0f) ? hdr->InverseScaleFactor : 1.0f;     // LightingRenderer.cs:1468-1472
    a.offset = hdr->Offset;
    a.exposure_minus_one = clamp(hdr->Exposure, min_v, max_v) - 1.0f;
    a.gamma_minus_one = clamp(hdr->Gamma, 0.1f, 4.0f) - 1.0f;
    const float white_point = clamp(hdr->Mode == ILM_HDR_TONE_MAP ? hdr->WhitePoint : 1.0f, min_v, max_v);
    {   // Uncharted2Tonemap1(WhitePoint), HDR.fxh:30-36
        const float kA = 0.15f, kB = 0.50f, kC = 0.10f, kD = 0.20f, kE = 0.02f, kF = 0.30f;
        const float w = ((white_point * (kA * white_point + kC * kB) + kD * kE) / (white_point * (kA * white_point + kB) + kD * kF)) - kE / kF;
        a.inv_white = 1.0f / w;
    }
    a.middle_gray = clamp(hdr->MiddleGray, 0.0f, max_v);
    a.inv_average_luminance = 1.0f / clamp(hdr->AverageLuminance, min_v, max_v);
    const float maximum_luminance = clamp(hdr->MaximumLuminance, min_v, max_v);
    a.inv_maximum_luminance_squared = 1.0f / (maximum_luminance * maximum_luminance);
    HIP_TRY(launch_resolve(a, c->main()));
    return ILM_OK;
}

int32_t ilm_render_sphere_lights(IlmHandle hctx, const IlmLightVertex* lights, int32_t light_count, const IlmEnvironment* env,
                                 const IlmDistanceFieldUniforms* df, IlmHandle hgbuffer, IlmHandle hsdf, const float ambient[4],
                                 IlmHandle hlightmap, int32_t row_begin, int32_t row_end, IlmRenderStats* stats) {
    ILM_TRACE_RANGE("ilm_render_sphere_lights");
    Ctx* c = from_handle<Ctx>(hctx, kMagicCtx);
    if (!c) return fail(ILM_ERR_INVALID_HANDLE, "not a context handle");
    Lightmap* m = from_handle<Lightmap>(hlightmap, kMagicLightmap);
    if (!m) return fail(ILM_ERR_INVALID_HANDLE, "not a lightmap handle");
    GBuffer* g = nullptr; Sdf* f = nullptr;
    if (hgbuffer) { g = from_handle<GBuffer>(hgbuffer, kMagicGBuffer); if (!g) return fail(ILM_ERR_INVALID_HANDLE, "not a G-buffer handle"); }
    if (hsdf) { f = from_handle<Sdf>(hsdf, kMagicSdf); if (!f) return fail(ILM_ERR_INVALID_HANDLE, "not a distance field handle"); }
    if (!env || !df) return fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    if (light_count < 0 || (light_count > 0 && !lights)) return fail(ILM_ERR_INVALID_ARGUMENT, "bad light array");
    if (light_count > 65535) return fail(ILM_ERR_TOO_MANY, "at most 65535 lights per call");
    // (the lightmap is written: the context's own; the field and the G-buffer are read: the context's or a sibling's, ilm_ctx_create_sibling)
    if (m->ctx != c || (g && !siblings(g->ctx, c)) || (f && !siblings(f->ctx, c))) return fail(ILM_ERR_INVALID_ARGUMENT, "resources belong to another context");
    { char why[256]; if (field_uniforms_mismatch(f, df, why, sizeof(why))) return fail(ILM_ERR_INVALID_ARGUMENT, "%s", why); }
    if (row_begin < 0 || row_end > m->height || row_begin > row_end)
        return fail(ILM_ERR_OUT_OF_RANGE, "rows [%d, %d) outside [0, %d]", row_begin, row_end, m->height);
    HIP_TRY(hipSetDevice(c->device));
    TraceSdfView trace_view;
    HIP_TRY(borrowed_trace_view(c, f, g, df, &trace_view));

    if (light_count > c->light_cap) {
        HIP_TRY(hipStreamSynchronize(c->main()));
        if (c->d_lights) HIP_TRY(hipFree(c->d_lights));
        if (c->d_recs) HIP_TRY(hipFree(c->d_recs));
        c->d_lights = nullptr; c->d_recs = nullptr;
        int cap = light_count < 256 ? 256 : light_count * 2;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_lights), sizeof(IlmLightVertex) * (size_t)cap));
        HIP_TRY(hipMalloc(&c->d_recs, kLightRecBytes * (size_t)cap));
        c->light_cap = cap;
    }
    if (light_count > 0) {
        static const int in_place = [] { const char* e = getenv("ILM_LIGHTS_IN_PLACE"); return e ? atoi(e) : 1; }();
        if (in_place) {
            // the vertices are read where the host left them (pinned ring): prepare_lights_kernel is their only reader
            const void* staged = nullptr; int slot = 0;
            int32_t rc = stage_small(c, lights, sizeof(IlmLightVertex) * (size_t)light_count, &staged, &slot);
            if (rc != ILM_OK) return rc;
            HIP_TRY(launch_prepare_lights(static_cast<const IlmLightVertex*>(staged), light_count, *env, *df, trace_view, c->d_recs, c->main()));
            rc = staged_small_done(c, slot);
            if (rc != ILM_OK) return rc;
        } else {
            int32_t rc = upload_small(c, c->d_lights, lights, sizeof(IlmLightVertex) * (size_t)light_count);
            if (rc != ILM_OK) return rc;
            HIP_TRY(launch_prepare_lights(c->d_lights, light_count, *env, *df, trace_view, c->d_recs, c->main()));
        }
    }

    LightLaunch a = {};
    a.lights = c->d_lights;
    a.light_count = light_count;
    a.env = *env;
    a.df = *df;
    a.gbuffer.texels = g ? g->texels : nullptr;
    a.gbuffer.width = g ? g->width : 0; a.gbuffer.height = g ? g->height : 0; a.gbuffer.format = g ? g->format : 0;
    a.sdf = trace_view;
    for (int i = 0; i < 4; i++) a.ambient[i] = ambient ? ambient[i] : 0.0f;
    a.lightmap = m->texels; a.width = m->width; a.height = m->height; a.format = m->format;
    a.row_begin = row_begin; a.row_end = row_end;
    a.stats = nullptr; a.light_count_ptr = nullptr;
    a.accumulate = ambient ? 0 : 1;          // a further light group of the frame: added to what the lightmap holds
    a.ramp = RampView{ c->d_light_ramp, c->light_ramp_w, c->light_ramp_h };
    a.blend_fp16 = (c->lightmap_blend == ILM_BLEND_FP16_PER_LIGHT) ? 1 : 0;
    a.tile_map = light_tile_map();
    a.tile_macro = light_tile_macro_for(m->width, row_end - row_begin);
    if (stats) {
        HIP_TRY(hipMemsetAsync(c->d_stats, 0, 3 * sizeof(unsigned long long), c->main()));
        a.stats = c->d_stats;
    }
    a.split = 1; a.partials = nullptr; a.tickets = nullptr; a.group_order = nullptr;
    { const int32_t rc = set_launch_mirrors(m, &a); if (rc != ILM_OK) return rc; }      // store-mode exchange of a group lightmap
    { const int32_t rc = plan_light_split(c, &a); if (rc != ILM_OK) return rc; }
    { const int32_t rc = plan_group_order(c, &a, lights, light_count, 16 * a.tile_macro); if (rc != ILM_OK) return rc; }
    c->last_light_blocks = light_launch_blocks(a); c->last_light_split = a.split; c->last_light_macro = (a.tile_map == 4) ? a.tile_macro : 0;
    HIP_TRY(launch_sphere_lights_prepared(a, c->d_recs, c->main()));
    HIP_TRY(light_pass_queued(c, f, g));
    if (stats) {
        unsigned long long host[3] = { 0, 0, 0 };
        HIP_TRY(hipMemcpyAsync(host, c->d_stats, sizeof(host), hipMemcpyDeviceToHost, c->main()));
        HIP_TRY(hipStreamSynchronize(c->main()));
        stats->SdfSamples = host[0]; stats->PixelLightPairs = host[1]; stats->TracedPairs = host[2];
    }
    return ILM_OK;
}

}  // extern "C"
