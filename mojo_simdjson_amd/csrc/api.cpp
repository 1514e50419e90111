// api.cpp -- extern "C" entry points declared in include/msj_stage1.h.
//
// Host-side counterpart of DomParserImplementation.stage1 / allocate
// (src/mojo_simdjson/include/generic/dom_parser_implementation.mojo:59-69,85-89):
// argument checks, workspace management and kernel launches.  There is no CPU
// implementation behind this ABI: without a usable HIP device every entry
// point returns MSJ_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "../../include/msj_stage1.h"
#include "stage1_kernel.h"
#include "tokens_launch.h"

// small-input path of msj_stage1: input, result and len + 3 indices fit the pinned staging buffer, which the kernel
// reads and writes itself over PCIe (no DMA calls; measured against staged copies: 77 B 24 -> 22 us, 13 KB 33 -> 23,
// 62 KB 45 -> 30, 258 KB 76 -> 42, 1 MB 136 -> 104; 4 MB 242 -> 371, hence the limit)
constexpr uint64_t kSmallInput = 1u << 20;
// from here on msj_stage1 stages through pinned rings in chunks (host_pipeline)
constexpr uint64_t kPipelineMinDefault = 64u << 20;  // (below that plain staging is as fast or faster; test hook: msj_debug_set_pipeline_min_bytes)
constexpr uint64_t kPinBytes = kSmallInput + 64 + (kSmallInput + 3) * sizeof(uint32_t) + 64;

static_assert(sizeof(msj_carry) == 64, "the small-input staging layout assumes a 64-byte carry");

namespace {
struct HostPipe;  // pinned rings, streams and copy workers of the host-pointer entry point (below)
}

struct msj_ctx {
    int device = 0;
    HostPipe *pipe = nullptr;     // created by the first large msj_stage1 call
    // Two workspace buffers (tickets + descriptors) used alternately.  A launch needs its
    // buffer zeroed; instead of a memset in front of every launch, each launch zeroes the
    // OTHER buffer word for word when that one was dirtied with the same layout (same ntiles).
    uint64_t *ws = nullptr;
    uint64_t ws_words = 0;        // words per buffer
    uint32_t ws_toggle = 0;
    uint32_t ws_dirty[2] = {0, 0}; // ntiles of the launch that last used the buffer; 0 = clean, ~0 = all of it
    msj_carry *carries = nullptr; // [0] = zero carry, [1..] chained segment carries
    uint32_t n_carries = 0;
    // staging for the host-pointer entry points
    uint8_t *d_in = nullptr;
    uint64_t d_in_bytes = 0;
    uint32_t *d_idx = nullptr;
    uint64_t d_idx_words = 0;
    msj_carry *d_result = nullptr;
    uint8_t *h_pin = nullptr;     // pinned host staging of the small-input path of msj_stage1 (kPinBytes)
    uint8_t *d_small = nullptr;   // ... and the msj_carry of that path (device memory: the kernel updates it with atomics)
    uint32_t grid = 0;            // persistent workgroups per launch (CUs x resident blocks per CU)
    uint32_t wait_ticks = msj::kWaitTicksDefault;  // bound of the kernel's waits (10 ns ticks)
    uint64_t seg_bytes = msj::kSegmentBytes;  // longest segment of one launch (test hook: msj_debug_set_segment_bytes)
    uint64_t pipeline_min = kPipelineMinDefault;  // host-pointer inputs from this size on take the chunked pipeline
    bool pipe_unavailable = false;  // the pipeline's pinned memory / streams could not be had: plain staging from then on
    bool pipe_fail_setup = false;   // test hook: msj_debug_fail_pipeline_setup
    uint64_t *tp = nullptr;       // workspace of the two-pass path (2 words per tile), allocated on first use
    uint64_t tp_words = 0;
    uint64_t fallbacks = 0;       // calls re-issued through the two-pass path after an expired wait
    // the last shard call, so that msj_carry_fetch can re-issue it (one in-flight call per context)
    struct {
        bool valid = false;
        const uint8_t *d_buf; uint64_t len; uint32_t *d_idx; uint64_t idx_capacity;
        const msj_carry *d_carry_in; msj_carry *d_carry_out; msj_segment *d_segments; uint32_t max_segments;
        bool has_prefix, is_final, no_emit; uint64_t trailer_len; hipStream_t stream; uint32_t flags;
        bool by_value; uint32_t carry_bits;  // msj_stage1_shard_device_cv
    } last;
    struct HostRange { const uint8_t *base; uint64_t bytes; };
    std::vector<HostRange> pinned;  // msj_host_register: caller-owned host ranges the DMA engines can reach directly
    bool is_pinned(const void *p, uint64_t n) const {
        const uint8_t *q = static_cast<const uint8_t *>(p);
        for (const HostRange &r : pinned)
            if (q >= r.base && n <= r.bytes && (uint64_t)(q - r.base) <= r.bytes - n) return true;
        return false;
    }
    uint32_t *span_fix = nullptr; // work list of the span kernel's fix-up pass (tokens_kernel.hip), zeroed once
    int32_t *tok_ws = nullptr;    // block aggregates of the token pre-pass
    uint64_t tok_ws_bytes = 0;
    uint64_t tok_doc_n = ~0ull;   // the token count whose document aggregates tok_ws holds (~0: none)
    uint32_t *seg_idx = nullptr;  // msj_stage2_prep_segments: 16-byte aligned copy of a segment's index slice that is not
    uint64_t seg_idx_words = 0;
    uint8_t *types_out = nullptr; // msj_stage1_types_device (prototype): where the launch being enqueued writes the type bytes
    uint32_t *resid = nullptr;    // msj_stage2_prep_segments with d_match: MSJ_RESID_WORDS per segment (the brackets a segment could not pair)
    uint64_t resid_words = 0;
    msj_token_opts tok_opts;      // test hooks of the token calls (msj_debug_set_span_limits / _span_mode): per context
    void *doc_ws = nullptr;       // block counts of the document split
    uint64_t doc_ws_bytes = 0;
};

namespace {

uint64_t *g_stamps = nullptr;  // diagnostic builds (-DMSJ_STAMPS) only: msj_debug_set_stamps
constexpr uint32_t kMaxChain = 64;  // segments per shard call (64 x ~4 GiB)

bool hip_ok(hipError_t e) { return e == hipSuccess; }

constexpr uint32_t kAllDirty = 0xFFFFFFFFu;

int32_t ensure_workspace(msj_ctx *ctx, uint32_t ntiles) {
    const uint64_t need = msj::workspace_words(ntiles);
    if (need <= ctx->ws_words) return MSJ_SUCCESS;
    if (ctx->ws) {
        (void)hipDeviceSynchronize();  // nothing of ours may still be using the old buffers
        (void)hipFree(ctx->ws);
    }
    ctx->ws = nullptr;
    ctx->ws_words = 0;
    // grow with head-room so repeated calls of similar size do not re-allocate
    const uint64_t words = (need + need / 4 + 64 + 511) & ~511ull;
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&ctx->ws), 2 * words * sizeof(uint64_t))))
        return MSJ_MEMALLOC;
    ctx->ws_words = words;
    ctx->ws_toggle = 0;
    ctx->ws_dirty[0] = ctx->ws_dirty[1] = kAllDirty;  // fresh memory: zeroed before first use
    return MSJ_SUCCESS;
}

// Zero what a previous launch (or nothing known) left in workspace buffer b.
bool scrub(msj_ctx *ctx, uint32_t b, hipStream_t stream) {
    const uint32_t d = ctx->ws_dirty[b];
    if (d == 0) return true;
    const uint64_t words = (d == kAllDirty) ? ctx->ws_words : msj::workspace_words(d);
    if (!hip_ok(hipMemsetAsync(ctx->ws + (uint64_t)b * ctx->ws_words, 0, words * sizeof(uint64_t), stream)))
        return false;
    ctx->ws_dirty[b] = 0;
    return true;
}

// Enqueue the kernels for one shard: a chain of <= kSegmentBytes launches whose
// carry structs stay in device memory (no host synchronisation in between).
int32_t enqueue_shard(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint32_t *d_idx,
                      uint64_t idx_capacity, const msj_carry *d_carry_in, msj_carry *d_carry_out,
                      msj_segment *d_segments, uint32_t max_segments, uint32_t *n_segments_out,
                      bool has_prefix, bool is_final, bool no_emit, uint64_t trailer_len,
                      hipStream_t stream, uint32_t flags, uint32_t index_bias = 0, const uint32_t *carry_bits = nullptr) {
    // carry_bits: the state at the shard's first byte by value (bit 0 in_string, 1 next_is_escaped, 2 prev_scalar;
    // counts and sticky flags zero) instead of d_carry_in
    if (!ctx || !d_buf || (!d_carry_in && !carry_bits) || !d_carry_out || len == 0) return MSJ_ERR_BAD_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(d_buf) & 15u) != 0) return MSJ_ERR_BAD_ARGUMENT;
    if (!no_emit && !d_idx) return MSJ_ERR_BAD_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(d_idx) & 15u) != 0) return MSJ_ERR_BAD_ARGUMENT;  // 16-B stores
    const uint64_t seg_bytes = ctx->seg_bytes;
    const uint64_t nseg = (len + seg_bytes - 1) / seg_bytes;
    if (nseg > kMaxChain) return MSJ_CAPACITY;
    if (d_segments && nseg > max_segments) return MSJ_CAPACITY;
    if (!hip_ok(hipSetDevice(ctx->device))) return MSJ_ERR_HIP;
    ctx->last.valid = true;
    ctx->last.d_buf = d_buf; ctx->last.len = len; ctx->last.d_idx = d_idx; ctx->last.idx_capacity = idx_capacity;
    ctx->last.d_carry_in = d_carry_in; ctx->last.d_carry_out = d_carry_out; ctx->last.d_segments = d_segments;
    ctx->last.max_segments = max_segments; ctx->last.has_prefix = has_prefix; ctx->last.is_final = is_final;
    ctx->last.no_emit = no_emit; ctx->last.trailer_len = trailer_len; ctx->last.stream = stream; ctx->last.flags = flags;
    ctx->last.by_value = carry_bits != nullptr; ctx->last.carry_bits = carry_bits ? *carry_bits : 0u;

    const uint64_t first_len = len < seg_bytes ? len : seg_bytes;
    const uint32_t max_tiles = (uint32_t)((first_len + msj::kTileBytes - 1) / msj::kTileBytes);
    int32_t rc = ensure_workspace(ctx, max_tiles);
    if (rc != MSJ_SUCCESS) return rc;

    for (uint64_t s = 0; s < nseg; s++) {
        const uint64_t base = s * seg_bytes;
        const uint64_t seg_len = (len - base) < seg_bytes ? (len - base) : seg_bytes;
        msj::KernelArgs a;
        a.buf = d_buf + base;
        a.len = seg_len;
        a.idx = d_idx;
        a.capacity = idx_capacity;
        const uint32_t wb = ctx->ws_toggle, wo = wb ^ 1u;
        a.ws = ctx->ws + (uint64_t)wb * ctx->ws_words;
        a.carry_in = (s == 0) ? d_carry_in : &ctx->carries[s];
        a.carry_bits = 0;
        a.reserved0 = 0;
        a.carry_out = (s + 1 == nseg) ? d_carry_out : &ctx->carries[s + 1];
        a.segment = d_segments ? &d_segments[s] : nullptr;
        a.segment_byte_base = base;
        a.trailer_len = trailer_len;
        a.ntiles = (uint32_t)((seg_len + msj::kTileBytes - 1) / msj::kTileBytes);
        a.flags = flags & (msj::kFlagStrictUtf8 | msj::kFlagNoUtf8);
        if (is_final && s + 1 == nseg) a.flags |= msj::kFlagFinal;
        if (has_prefix || s > 0) a.flags |= msj::kFlagHasPrefix;
        if (no_emit) a.flags |= msj::kFlagNoEmit;
        if (s == 0) a.flags |= flags & (15u << msj::kFlagSkipShift);
        if (s > 0) a.flags |= msj::kFlagEchoThrough;
        if (s == 0 && carry_bits) {
            a.flags |= msj::kFlagCarryByValue;
            a.carry_bits = *carry_bits & 7u;
            a.carry_in = nullptr;
        }
        a.stamps = g_stamps;
        a.types = ctx->types_out;
        if (a.types) a.flags |= msj::kFlagEmitTypes;
        a.wait_ticks = ctx->wait_ticks;
        // without a segment table nothing tells the caller where a later segment's offsets start: they stay
        // relative to the call's buffer (wrapping like the reference's UInt32 would, json_structural_indexer.mojo:138)
        a.index_bias = index_bias + (d_segments ? 0u : (uint32_t)base);
        a.tp = nullptr;
        if (flags & MSJ_FLAG_DEBUG_STALL) a.flags |= msj::kFlagDebugStall;
        if (flags & MSJ_FLAG_TWO_PASS) {
            // three plain kernels, no inter-workgroup waiting, own workspace (need not be zeroed)
            const uint64_t need = 2ull * max_tiles;
            if (need > ctx->tp_words) {
                if (ctx->tp) {
                    (void)hipDeviceSynchronize();
                    (void)hipFree(ctx->tp);
                }
                ctx->tp = nullptr;
                ctx->tp_words = 0;
                if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&ctx->tp), need * sizeof(uint64_t)))) return MSJ_MEMALLOC;
                ctx->tp_words = need;
            }
            a.tp = ctx->tp;
            a.ws = nullptr;
            a.ws_clean = nullptr;
            if (msj_launch_stage1_twopass(&a, stream) != 0) return MSJ_ERR_HIP;
            continue;
        }
        // ticket + descriptors must read as "not ready" at launch: this launch's buffer is clean
        // already in the steady state; the other one is cleaned by this launch if its dirt has
        // this launch's layout, by a memset otherwise
        a.ws_clean = nullptr;
        if (ctx->ws_dirty[wo] == a.ntiles) a.ws_clean = ctx->ws + (uint64_t)wo * ctx->ws_words;
        if (!scrub(ctx, wb, stream) || (!a.ws_clean && !scrub(ctx, wo, stream))) {
            ctx->ws_dirty[0] = ctx->ws_dirty[1] = kAllDirty;
            return MSJ_ERR_HIP;
        }
        // (Tried in round 3: a grid sized so that every workgroup gets the same number of ranges -- 993 x 33 instead of
        // 1 023 workers with 31 stragglers in a 33rd round at 1 GiB -- loses 3-4 %: 30 empty workgroup slots cost more
        // than the stragglers' ~5 us.  The grid fills every CU.)
        if (msj_launch_stage1(&a, stream, ctx->grid) != 0) {
            ctx->ws_dirty[0] = ctx->ws_dirty[1] = kAllDirty;
            return MSJ_ERR_HIP;
        }
        ctx->ws_dirty[wb] = a.ntiles;
        ctx->ws_dirty[wo] = 0;
        ctx->ws_toggle = wo;
    }
    if (n_segments_out) *n_segments_out = (uint32_t)nseg;
    return MSJ_SUCCESS;
}

// ---- host-pointer path for large inputs: library-owned pinned rings, chunked and overlapped -----------------
// DomParserImplementation.stage1 (include/generic/dom_parser_implementation.mojo:65-69) hands over pageable host
// memory.  Pageable hipMemcpy is synchronous and its two directions do not overlap on this platform (measured:
// 31 GB/s of JSON for 268 MB up + 208 MB down), while pinned memory moves 57 GB/s each way at once
// (scripts/ubench/pcie_probe.cpp).  So: the input goes up in chunks through a ring of pinned buffers, filled by a
// few copy threads (one thread copies 32 GB/s, four 96 GB/s on the box's host); every chunk is one shard launch
// with the carry chained in device memory (msj_stage1_shard_device); a second host thread follows the chunks'
// counts and brings the finished part of the index array down through a second pinned ring while later chunks
// are still on their way up -- both PCIe directions and the kernel run at the same time.
// ---- where the host side of the pipeline lives (round 5: the PCIe-inclusive rate differed by 37 % between two boxes of
// the pool with nothing in the record to say why).  The GPU hangs off ONE NUMA node's root complex: staging copies that
// run on the other socket, or pinned rings whose pages lie there, cross the inter-socket link twice.  The copy workers
// are therefore bound to the CPUs of the GPU's node (those of them the process may use: a cgroup / taskset limit is
// respected; no such CPU -> no binding), the rings are allocated by a thread bound the same way (first touch), and
// msj_host_placement reports all of it.  Linux sysfs / syscalls only, no libnuma; anything unreadable reads as -1.
struct GpuHostLocality {
    char pci[32] = "";
    int node = -1;          // NUMA node of the GPU's PCIe root complex (-1: unknown / single node)
    cpu_set_t cpus;         // CPUs of that node that this process may run on
    int n_cpus = 0;
    char link_speed[32] = "", link_width[16] = "";
};
static bool read_line(const char *path, char *out, size_t cap) {
    FILE *f = std::fopen(path, "r");
    if (!f) return false;
    const bool ok = std::fgets(out, (int)cap, f) != nullptr;
    std::fclose(f);
    if (ok) out[std::strcspn(out, "\n")] = 0;
    return ok;
}
static GpuHostLocality gpu_locality(int device) {
    GpuHostLocality g;
    CPU_ZERO(&g.cpus);
    char bus[32] = "";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) {
        (void)hipGetLastError();
        return g;
    }
    for (char *c = bus; *c; c++)
        if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');  // sysfs spells the address in lower case
    std::snprintf(g.pci, sizeof g.pci, "%s", bus);
    char path[128], line[4096];
    std::snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    if (read_line(path, line, sizeof line)) g.node = std::atoi(line);
    std::snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/current_link_speed", bus);
    (void)read_line(path, g.link_speed, sizeof g.link_speed);
    std::snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/current_link_width", bus);
    (void)read_line(path, g.link_width, sizeof g.link_width);
    std::snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bus);
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (read_line(path, line, sizeof line) && sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
        for (char *tok = std::strtok(line, ","); tok; tok = std::strtok(nullptr, ",")) {  // "0-31,64-95"
            int lo = 0, hi = 0;
            const int k = std::sscanf(tok, "%d-%d", &lo, &hi);
            if (k == 1) hi = lo;
            for (int c = lo; k >= 1 && c <= hi && c < CPU_SETSIZE; c++)
                if (CPU_ISSET(c, &allowed)) {
                    CPU_SET(c, &g.cpus);
                    g.n_cpus++;
                }
        }
    }
    return g;
}
// NUMA node a mapped page lies on (get_mempolicy(MPOL_F_NODE | MPOL_F_ADDR)); -1 where the kernel will not say
static int numa_node_of(const void *p) {
#ifdef SYS_get_mempolicy
    int node = -1;
    if (p && syscall(SYS_get_mempolicy, &node, nullptr, 0UL, const_cast<void *>(p), 3UL /* MPOL_F_NODE | MPOL_F_ADDR */) == 0) return node;
#endif
    (void)p;
    return -1;
}

struct CopyPool {
    const cpu_set_t *bind = nullptr;  // the GPU's CPUs (HostPipe): every worker runs there
    int bound = 0;                    // workers whose affinity call succeeded
    std::vector<std::thread> threads;
    std::deque<std::function<void()>> tasks;
    std::mutex m;
    std::condition_variable cv;
    bool stop = false;
    explicit CopyPool(int n, const cpu_set_t *cpus = nullptr) : bind(cpus) {
        std::atomic<int> ok{0};
        for (int i = 0; i < n; i++)
            threads.emplace_back([this, &ok] {
                if (bind && pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), bind) == 0) ok.fetch_add(1);
                ok.fetch_add(1 << 16);  // this worker has started
                for (;;) {
                    std::function<void()> f;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv.wait(lk, [this] { return stop || !tasks.empty(); });
                        if (stop && tasks.empty()) return;
                        f = std::move(tasks.front());
                        tasks.pop_front();
                    }
                    f();
                }
            });
        while ((ok.load() >> 16) < n) std::this_thread::yield();  // (`ok` lives on this frame)
        bound = ok.load() & 0xFFFF;
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : threads) t.join();
    }
    // memcpy split over `parts` workers, not waited for: *pending counts the slices still to do
    void copy_async(void *dst, const void *src, uint64_t n, int parts, std::atomic<int> *pending) {
        if (parts < 1) parts = 1;
        pending->store(parts, std::memory_order_relaxed);
        const uint64_t step = ((n / parts) + 63) & ~63ull;
        for (int i = 0; i < parts; i++) {
            const uint64_t lo = step * i < n ? step * i : n, hi = (i + 1 == parts || step * (i + 1) > n) ? n : step * (i + 1);
            {
                std::lock_guard<std::mutex> lk(m);
                tasks.emplace_back([=] {
                    if (hi > lo) std::memcpy(static_cast<char *>(dst) + lo, static_cast<const char *>(src) + lo, hi - lo);
                    pending->fetch_sub(1, std::memory_order_release);
                });
            }
            cv.notify_one();
        }
    }
    static void wait(std::atomic<int> *pending) {
        while (pending->load(std::memory_order_acquire) != 0) std::this_thread::yield();
    }
    // memcpy split over `parts` workers; returns when all of it is done
    void copy(void *dst, const void *src, uint64_t n, int parts) {
        if (n < (1u << 20) || parts <= 1) {
            std::memcpy(dst, src, n);
            return;
        }
        std::mutex dm;
        std::condition_variable dcv;
        int left = parts;
        const uint64_t step = ((n / parts) + 63) & ~63ull;
        for (int i = 0; i < parts; i++) {
            const uint64_t lo = step * i < n ? step * i : n, hi = (i + 1 == parts || step * (i + 1) > n) ? n : step * (i + 1);
            {
                std::lock_guard<std::mutex> lk(m);
                tasks.emplace_back([=, &dm, &dcv, &left] {
                    if (hi > lo) std::memcpy(static_cast<char *>(dst) + lo, static_cast<const char *>(src) + lo, hi - lo);
                    std::lock_guard<std::mutex> g(dm);
                    if (--left == 0) dcv.notify_one();
                });
            }
            cv.notify_one();
        }
        std::unique_lock<std::mutex> lk(dm);
        dcv.wait(lk, [&] { return left == 0; });
    }
};

// Tuning knobs of the host pipeline exist in the measurement build only (make -C csrc knobs: -DMSJ_DEBUG_KNOBS,
// scripts/libmsj_stage1_knobs.so); the product library has the measured defaults compiled in and reads no
// environment variable at all.
#ifdef MSJ_DEBUG_KNOBS
static int knob_int(const char *name, int dflt, int lo) {
    const char *v = std::getenv(name);
    const int x = v && *v ? std::atoi(v) : dflt;
    return x < lo ? lo : x;  // a pool without workers would block its callers for ever
}
static bool knob_set(const char *name) { return std::getenv(name) != nullptr; }
#else
static int knob_int(const char *, int dflt, int) { return dflt; }
static bool knob_set(const char *) { return false; }
#endif

struct HostPipe {
    static constexpr int kInSlots = 3, kOutSlots = 2;
    // copy workers and slices per staging copy (defaults measured on the MI355X box's host)
    const int kCopyThreads = knob_int("MSJ_PIPE_THREADS", 8, 1), kParts = knob_int("MSJ_PIPE_PARTS", 4, 1);
    const bool direct_upload = knob_int("MSJ_PIPE_DIRECT_UPLOAD", 0, 0) != 0;
    static constexpr uint64_t kChunk = 16ull << 20;  // input bytes per chunk (a multiple of the tile)
    static constexpr uint64_t kPiece = 16ull << 20;  // index bytes per download piece
    uint8_t *pin_in[kInSlots] = {nullptr, nullptr, nullptr};
    uint8_t *pin_out[kOutSlots] = {nullptr, nullptr};
    msj_carry *h_carries = nullptr;  // pinned: the carry after every chunk
    msj_carry *d_carries = nullptr;
    uint64_t n_carries = 0;
    hipStream_t s_up = nullptr, s_k = nullptr, s_down = nullptr;
    hipEvent_t ev_in[kInSlots] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_out[kOutSlots] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev_chunk;
    GpuHostLocality where;            // the GPU's NUMA node and the CPUs of it this process may use
    CopyPool pool;
    bool ok = false;

    explicit HostPipe(int device) : where(gpu_locality(device)), pool(kCopyThreads, where.n_cpus > 0 ? &where.cpus : nullptr) {
        ok = true;
        // the rings: allocated (and touched) by a thread that runs on the GPU's node, so that first-touch placement puts
        // their pages there; the caller's thread keeps its own affinity
        std::thread([&] {
            (void)hipSetDevice(device);
            if (where.n_cpus > 0) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &where.cpus);
            for (auto &p : pin_in) {
                ok = ok && hip_ok(hipHostMalloc(reinterpret_cast<void **>(&p), kChunk, hipHostMallocDefault));
                if (ok) std::memset(p, 0, kChunk);
            }
            for (auto &p : pin_out) {
                ok = ok && hip_ok(hipHostMalloc(reinterpret_cast<void **>(&p), kPiece, hipHostMallocDefault));
                if (ok) std::memset(p, 0, kPiece);
            }
        }).join();
        ok = ok && hip_ok(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking)) &&
             hip_ok(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking)) &&
             hip_ok(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
        for (auto &e : ev_in) ok = ok && hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : ev_out) ok = ok && hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    ~HostPipe() {
        for (auto p : pin_in)
            if (p) (void)hipHostFree(p);
        for (auto p : pin_out)
            if (p) (void)hipHostFree(p);
        if (h_carries) (void)hipHostFree(h_carries);
        if (d_carries) (void)hipFree(d_carries);
        for (auto e : ev_in)
            if (e) (void)hipEventDestroy(e);
        for (auto e : ev_out)
            if (e) (void)hipEventDestroy(e);
        for (auto e : ev_chunk) (void)hipEventDestroy(e);
        if (s_up) (void)hipStreamDestroy(s_up);
        if (s_k) (void)hipStreamDestroy(s_k);
        if (s_down) (void)hipStreamDestroy(s_down);
    }
    bool reserve(uint64_t nchunks) {
        if (nchunks + 1 > n_carries) {
            if (h_carries) (void)hipHostFree(h_carries);
            if (d_carries) (void)hipFree(d_carries);
            h_carries = nullptr;
            d_carries = nullptr;
            n_carries = 0;
            if (!hip_ok(hipHostMalloc(reinterpret_cast<void **>(&h_carries), (nchunks + 1) * sizeof(msj_carry), hipHostMallocDefault)) ||
                !hip_ok(hipMalloc(reinterpret_cast<void **>(&d_carries), (nchunks + 1) * sizeof(msj_carry))))
                return false;
            n_carries = nchunks + 1;
        }
        while (ev_chunk.size() < nchunks) {
            hipEvent_t e;
            if (!hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming))) return false;
            ev_chunk.push_back(e);
        }
        return true;
    }
};

// The pipelined form of msj_stage1_ctx's device staging.  Returns kPipeUnavailable when the machinery cannot be
// set up (pinned memory, streams, events: e.g. a memlock limit in a container) -- nothing has been enqueued then
// and the caller takes the plain path, for this call and every later one; any other failure is the call's
// result.  Otherwise fills *res with the final carry.
constexpr int32_t kPipeUnavailable = -100;
int32_t host_pipeline(msj_ctx *ctx, const uint8_t *buf, uint64_t len, uint32_t *idx_out, uint64_t dev_cap, uint32_t flags,
                      msj_carry *res) {
    if (ctx->pipe_fail_setup) return kPipeUnavailable;  // test hook (msj_debug_fail_pipeline_setup)
    if (!ctx->pipe) {
        ctx->pipe = new (std::nothrow) HostPipe(ctx->device);
        if (!ctx->pipe) return kPipeUnavailable;
    }
    HostPipe &P = *ctx->pipe;
    if (!P.ok) return kPipeUnavailable;
    const uint64_t chunk = HostPipe::kChunk;
    const uint64_t nchunks = (len + chunk - 1) / chunk;
    if (!P.reserve(nchunks)) return kPipeUnavailable;
    if (!hip_ok(hipMemsetAsync(&P.d_carries[0], 0, sizeof(msj_carry), P.s_k))) return kPipeUnavailable;

    static const bool trace = knob_set("MSJ_PIPE_TRACE");  // measurement build: where the call's time goes
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    std::vector<double> t_chunk_done(nchunks, 0.0), t_piece;
    // msj_host_register: a side whose caller memory is pinned needs no staging
    const bool in_pinned = P.direct_upload || ctx->is_pinned(buf, len);
    const bool out_pinned = ctx->is_pinned(idx_out, dev_cap * sizeof(uint32_t));
    // ---- the downloader: follows the chunks' cumulative counts, brings finished indices down in pieces
    int32_t down_rc = MSJ_SUCCESS;
    std::atomic<bool> abort{false};
    std::atomic<uint64_t> recorded{0};  // chunks whose event the uploader has recorded (an unrecorded event reads as done)
    std::thread down([&] {
        (void)hipSetDevice(ctx->device);
        uint64_t sent = 0, pieces = 0;
        std::atomic<int> copying[HostPipe::kOutSlots];
        for (auto &c : copying) c.store(0);
        for (uint64_t k = 0; k < nchunks; k++) {
            while (recorded.load(std::memory_order_acquire) <= k && !abort.load()) std::this_thread::yield();
            if (abort.load()) break;
            if (!hip_ok(hipEventSynchronize(P.ev_chunk[k]))) { down_rc = MSJ_ERR_HIP; break; }
            if (abort.load()) break;
            if (trace) t_chunk_done[k] = now() - t_begin;
            const msj_carry &c = P.h_carries[k + 1];
            const bool last = k + 1 == nchunks;
            uint64_t avail = c.count;
            if (last && (c.code == MSJ_SUCCESS || c.code == MSJ_EMPTY || c.code == MSJ_UTF8_ERROR)) avail += 3;  // the trailer
            if (avail > dev_cap) avail = dev_cap;
            if (out_pinned) {  // the caller's array is pinned: one DMA per chunk straight into it, nothing to wait for here
                if (avail > sent &&
                    !hip_ok(hipMemcpyAsync(idx_out + sent, ctx->d_idx + sent, (avail - sent) * sizeof(uint32_t), hipMemcpyDeviceToHost, P.s_down))) {
                    down_rc = MSJ_ERR_HIP;
                    break;
                }
                sent = avail;
                if (trace) t_piece.push_back(now() - t_begin);
                continue;
            }
            const uint64_t piece = HostPipe::kPiece / sizeof(uint32_t);
            while (sent < avail) {  // whatever this chunk added, in pieces of at most one slot
                // source and slot keep the same offset inside a 256-byte line: a DMA between differently aligned
                // ends runs at half the rate
                const uint64_t mis = sent & 63u;
                const uint64_t n = avail - sent < piece - mis ? avail - sent : piece - mis;
                const int slot = (int)(pieces % HostPipe::kOutSlots);
                CopyPool::wait(&copying[slot]);  // the piece that used this slot has been copied out
                // the DMA into the pinned slot (this call returns when it is done), then the copy into the
                // caller's memory by the pool while the next piece's DMA runs
                if (!hip_ok(hipMemcpyAsync(P.pin_out[slot], ctx->d_idx + (sent - mis), (n + mis) * sizeof(uint32_t), hipMemcpyDeviceToHost, P.s_down)) ||
                    !hip_ok(hipStreamSynchronize(P.s_down))) {
                    down_rc = MSJ_ERR_HIP;
                    break;
                }
                P.pool.copy_async(idx_out + sent, P.pin_out[slot] + mis * sizeof(uint32_t), n * sizeof(uint32_t), P.kParts, &copying[slot]);
                sent += n;
                pieces++;
                if (trace) t_piece.push_back(now() - t_begin);
            }
            if (down_rc != MSJ_SUCCESS) break;
        }
        for (auto &c : copying) CopyPool::wait(&c);
        if (out_pinned && !hip_ok(hipStreamSynchronize(P.s_down))) down_rc = MSJ_ERR_HIP;
    });

    // ---- the uploader (this thread): pinned staging, H2D, one shard launch per chunk
    double t_copy = 0, t_wait = 0;
    int32_t rc = MSJ_SUCCESS;
    for (uint64_t k = 0; k < nchunks && rc == MSJ_SUCCESS; k++) {
        const uint64_t off = k * chunk, n = len - off < chunk ? len - off : chunk;
        const int slot = (int)(k % HostPipe::kInSlots);
        double t0 = now();
        if (in_pinned) {
            // the caller's pages are pinned (or MSJ_PIPE_DIRECT_UPLOAD: the runtime pins them in flight): no staging copy of ours
            if (!hip_ok(hipMemcpyAsync(ctx->d_in + off, buf + off, n, hipMemcpyHostToDevice, P.s_up))) rc = MSJ_ERR_HIP;
            t_copy += now() - t0;
        } else {
            if (k >= (uint64_t)HostPipe::kInSlots && !hip_ok(hipEventSynchronize(P.ev_in[slot]))) rc = MSJ_ERR_HIP;
            double t1 = now();
            if (rc == MSJ_SUCCESS) P.pool.copy(P.pin_in[slot], buf + off, n, P.kParts);
            t_wait += t1 - t0;
            t_copy += now() - t1;
            if (rc == MSJ_SUCCESS && !hip_ok(hipMemcpyAsync(ctx->d_in + off, P.pin_in[slot], n, hipMemcpyHostToDevice, P.s_up)))
                rc = MSJ_ERR_HIP;
        }
        if (rc == MSJ_SUCCESS && (!hip_ok(hipEventRecord(P.ev_in[slot], P.s_up)) || !hip_ok(hipStreamWaitEvent(P.s_k, P.ev_in[slot], 0))))
            rc = MSJ_ERR_HIP;
        if (rc == MSJ_SUCCESS)
            rc = enqueue_shard(ctx, ctx->d_in + off, n, ctx->d_idx, dev_cap, &P.d_carries[k], &P.d_carries[k + 1], nullptr, 0, nullptr,
                               k > 0, k + 1 == nchunks, false, len, P.s_k, flags, (uint32_t)off);
        if (rc == MSJ_SUCCESS &&
            (!hip_ok(hipMemcpyAsync(&P.h_carries[k + 1], &P.d_carries[k + 1], sizeof(msj_carry), hipMemcpyDeviceToHost, P.s_k)) ||
             !hip_ok(hipEventRecord(P.ev_chunk[k], P.s_k))))
            rc = MSJ_ERR_HIP;
        if (rc == MSJ_SUCCESS) recorded.store(k + 1, std::memory_order_release);
        if (rc != MSJ_SUCCESS) {
            // the downloader waits on every chunk's event: record the rest so that it can leave
            abort.store(true);
            for (uint64_t j = k; j < nchunks; j++) (void)hipEventRecord(P.ev_chunk[j], P.s_k);
        }
    }
    const double t_up = now();
    down.join();
    (void)hipStreamSynchronize(P.s_k);
    if (trace)
        std::fprintf(stderr, "msj host pipeline: %llu chunks, upload loop %.2f ms (slot waits %.2f, staging copies %.2f), "
                             "then %.2f ms until the last index was down\n",
                     (unsigned long long)nchunks, t_up - t_begin, t_wait, t_copy, now() - t_up);
    if (trace) {
        std::fprintf(stderr, "  chunk results seen at (ms):");
        for (double t : t_chunk_done) std::fprintf(stderr, " %.2f", t);
        std::fprintf(stderr, "\n  download pieces issued+previous copied out at (ms):");
        for (double t : t_piece) std::fprintf(stderr, " %.2f", t);
        std::fprintf(stderr, "\n");
    }
    ctx->last.valid = false;  // the chunk launches are not one call that msj_carry_fetch could re-issue
    if (rc != MSJ_SUCCESS) return rc;
    if (down_rc != MSJ_SUCCESS) return down_rc;
    *res = P.h_carries[nchunks];
    return MSJ_SUCCESS;
}


std::mutex g_default_mutex;
msj_ctx *g_default_ctx = nullptr;

}  // namespace

extern "C" {

#ifndef MSJ_SOURCE_HASH
#define MSJ_SOURCE_HASH "unknown"
#endif
// "... src:<hash>": the first 12 hex digits of the SHA-256 of the stage-1 kernel's sources (csrc/Makefile), so that
// measurements kept beside the code (profiles/traffic.json) can say which kernel they were taken with
const char *msj_version(void) { return "mojo-simdjson_amd stage1 0.4 (gfx950) src:" MSJ_SOURCE_HASH; }

uint32_t msj_tile_bytes(void) { return msj::kTileBytes; }

int32_t msj_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t msj_ctx_create(int32_t device, msj_ctx **out) {
    if (!out) return MSJ_ERR_BAD_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MSJ_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return MSJ_ERR_BAD_ARGUMENT;
    if (!hip_ok(hipSetDevice(device))) return MSJ_ERR_HIP;
    msj_ctx *ctx = new (std::nothrow) msj_ctx();
    if (!ctx) return MSJ_MEMALLOC;
    ctx->device = device;
    {
        // persistent grid: fill every CU to the kernel's occupancy
        hipDeviceProp_t prop;
        int per_cu = 0;
        if (hip_ok(hipGetDeviceProperties(&prop, device)) && msj_stage1_occupancy(&per_cu) == 0 &&
            per_cu > 0)
            ctx->grid = (uint32_t)prop.multiProcessorCount * (uint32_t)per_cu;
        else
            ctx->grid = 1024;
    }
    ctx->n_carries = kMaxChain + 2;
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&ctx->carries),
                          ctx->n_carries * sizeof(msj_carry))) ||
        !hip_ok(hipMemset(ctx->carries, 0, ctx->n_carries * sizeof(msj_carry))) ||
        !hip_ok(hipMalloc(reinterpret_cast<void **>(&ctx->d_result), sizeof(msj_carry)))) {
        msj_ctx_destroy(ctx);
        return MSJ_MEMALLOC;
    }
    *out = ctx;
    return MSJ_SUCCESS;
}

int32_t msj_ctx_device(const msj_ctx *ctx) { return ctx ? ctx->device : -1; }

void msj_ctx_destroy(msj_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->tp) (void)hipFree(ctx->tp);
    if (ctx->tok_ws) (void)hipFree(ctx->tok_ws);
    if (ctx->seg_idx) (void)hipFree(ctx->seg_idx);
    if (ctx->resid) (void)hipFree(ctx->resid);
    if (ctx->span_fix) (void)hipFree(ctx->span_fix);
    if (ctx->doc_ws) (void)hipFree(ctx->doc_ws);
    if (ctx->carries) (void)hipFree(ctx->carries);
    if (ctx->d_in) (void)hipFree(ctx->d_in);
    if (ctx->d_idx) (void)hipFree(ctx->d_idx);
    if (ctx->d_result) (void)hipFree(ctx->d_result);
    delete ctx->pipe;
    for (const msj_ctx::HostRange &r : ctx->pinned) (void)hipHostUnregister(const_cast<uint8_t *>(r.base));
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->d_small) (void)hipFree(ctx->d_small);
    delete ctx;
}

int32_t msj_stage1_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint32_t *d_idx,
                          uint64_t idx_capacity, msj_carry *d_result, void *stream,
                          uint32_t flags) {
    if (!ctx || !d_result) return MSJ_ERR_BAD_ARGUMENT;
    if (len == 0) return MSJ_EMPTY;  // json_structural_indexer.mojo:91-92
    if (len > MSJ_MAX_SEGMENT_BYTES) return MSJ_CAPACITY;  // include/base.mojo:2
    // carries[0] is the all-zero state at the start of a document (:74-79)
    return enqueue_shard(ctx, d_buf, len, d_idx, idx_capacity, &ctx->carries[0], d_result, nullptr,
                         0, nullptr, false, true, false, len, static_cast<hipStream_t>(stream),
                         flags);
}

static bool ensure_tok_ws(msj_ctx *ctx, uint64_t need);
extern "C" int msj_launch_depth_from_types(const uint8_t *d_type, uint64_t n, int32_t *d_depth, uint32_t *d_match, msj_tokens_result *d_result,
                                           int32_t *d_ws, void *stream, const msj_token_opts &o);

int32_t msj_stage1_types_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint32_t *d_idx, uint64_t idx_capacity,
                                uint8_t *d_types, msj_carry *d_result, void *stream, uint32_t flags) {
    if (!ctx || !d_result || !d_types || (reinterpret_cast<uintptr_t>(d_types) & 3u)) return MSJ_ERR_BAD_ARGUMENT;
    if (len == 0) return MSJ_EMPTY;
    if (len > ctx->seg_bytes || (flags & MSJ_FLAG_TWO_PASS)) return MSJ_CAPACITY;  // one single-pass launch (prototype)
    ctx->types_out = d_types;
    const int32_t rc = enqueue_shard(ctx, d_buf, len, d_idx, idx_capacity, &ctx->carries[0], d_result, nullptr, 0, nullptr, false, true, false,
                                     len, static_cast<hipStream_t>(stream), flags);
    ctx->types_out = nullptr;
    ctx->last.valid = false;  // (no two-pass fallback for this form: a poisoned launch stays poisoned)
    return rc;
}

int32_t msj_depth_from_types_device(msj_ctx *ctx, const uint8_t *d_type, uint64_t n, int32_t *d_depth, uint32_t *d_match,
                                    msj_tokens_result *d_result, const msj_tokens_result *d_prev, void *stream) {
    if (!ctx || !d_result || d_prev == d_result) return MSJ_ERR_BAD_ARGUMENT;
    if (n > 0 && (!d_type || !d_depth)) return MSJ_ERR_BAD_ARGUMENT;
    if (n >= (1ull << 31)) return MSJ_CAPACITY;
    if ((reinterpret_cast<uintptr_t>(d_depth) & 15u) || (reinterpret_cast<uintptr_t>(d_type) & 7u) || (reinterpret_cast<uintptr_t>(d_match) & 15u))
        return MSJ_ERR_BAD_ARGUMENT;
    if (!hip_ok(hipSetDevice(ctx->device))) return MSJ_ERR_HIP;
    if (!ensure_tok_ws(ctx, msj_stage2_prep_workspace_bytes(n, 0, d_match != nullptr))) return MSJ_MEMALLOC;
    ctx->tok_doc_n = ~0ull;
    msj_token_opts o = ctx->tok_opts;
    o.d_prev = d_prev;
    if (msj_launch_depth_from_types(d_type, n, d_depth, d_match, d_result, ctx->tok_ws, stream, o) != 0) return MSJ_ERR_HIP;
    ctx->tok_doc_n = n;
    return MSJ_SUCCESS;
}

int32_t msj_stage1_shard_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint32_t *d_idx,
                                uint64_t idx_capacity, const msj_carry *d_carry_in,
                                msj_carry *d_carry_out, msj_segment *d_segments,
                                uint32_t max_segments, uint32_t *n_segments_out,
                                int32_t has_prefix, int32_t is_final, int32_t no_emit,
                                uint64_t trailer_len, void *stream, uint32_t flags) {
    return enqueue_shard(ctx, d_buf, len, d_idx, idx_capacity, d_carry_in, d_carry_out, d_segments,
                         max_segments, n_segments_out, has_prefix != 0, is_final != 0, no_emit != 0,
                         trailer_len, static_cast<hipStream_t>(stream), flags);
}

int32_t msj_stage1_shard_device_cv(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint32_t *d_idx,
                                   uint64_t idx_capacity, uint32_t carry_bits, msj_carry *d_carry_out,
                                   msj_segment *d_segments, uint32_t max_segments, uint32_t *n_segments_out,
                                   int32_t has_prefix, int32_t is_final, int32_t no_emit, uint64_t trailer_len,
                                   void *stream, uint32_t flags) {
    if (carry_bits > 7u) return MSJ_ERR_BAD_ARGUMENT;
    return enqueue_shard(ctx, d_buf, len, d_idx, idx_capacity, nullptr, d_carry_out, d_segments, max_segments,
                         n_segments_out, has_prefix != 0, is_final != 0, no_emit != 0, trailer_len,
                         static_cast<hipStream_t>(stream), flags, 0, &carry_bits);
}


// the workspace of the token calls (block aggregates, chunk aggregates, group table): grown when a call needs more
static bool ensure_tok_ws(msj_ctx *ctx, uint64_t need) {
    if (need <= ctx->tok_ws_bytes) return true;
    if (ctx->tok_ws) {
        (void)hipDeviceSynchronize();
        (void)hipFree(ctx->tok_ws);
    }
    ctx->tok_ws = nullptr;
    ctx->tok_ws_bytes = 0;
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&ctx->tok_ws), need + need / 4))) return false;
    ctx->tok_ws_bytes = need + need / 4;
    return true;
}

int32_t msj_tokens_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n,
                          uint8_t *d_type, int32_t *d_depth, uint32_t *d_match, msj_tokens_result *d_result,
                          void *stream) {
    return msj_tokens_chain_device(ctx, d_buf, len, d_idx, n, d_type, d_depth, d_match, d_result, nullptr, stream);
}

static int32_t tokens_chain_impl(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n, uint8_t *d_type,
                                 int32_t *d_depth, uint32_t *d_match, msj_tokens_result *d_result, const msj_tokens_result *d_prev,
                                 void *stream, msj_bracket_pair *d_pairs);
int32_t msj_tokens_chain_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n,
                                uint8_t *d_type, int32_t *d_depth, uint32_t *d_match, msj_tokens_result *d_result,
                                const msj_tokens_result *d_prev, void *stream) {
    return tokens_chain_impl(ctx, d_buf, len, d_idx, n, d_type, d_depth, d_match, d_result, d_prev, stream, nullptr);
}
int32_t msj_tokens_pairs_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n, uint8_t *d_type,
                                int32_t *d_depth, msj_bracket_pair *d_pairs, msj_tokens_result *d_result,
                                const msj_tokens_result *d_prev, void *stream) {
    if (!d_pairs || (reinterpret_cast<uintptr_t>(d_pairs) & 7u)) return MSJ_ERR_BAD_ARGUMENT;
    return tokens_chain_impl(ctx, d_buf, len, d_idx, n, d_type, d_depth, nullptr, d_result, d_prev, stream, d_pairs);
}
static int32_t tokens_chain_impl(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n, uint8_t *d_type,
                                 int32_t *d_depth, uint32_t *d_match, msj_tokens_result *d_result, const msj_tokens_result *d_prev,
                                 void *stream, msj_bracket_pair *d_pairs) {
    if (!ctx || !d_result || d_prev == d_result) return MSJ_ERR_BAD_ARGUMENT;
    if (n > 0 && (!d_buf || !d_idx || !d_type || !d_depth)) return MSJ_ERR_BAD_ARGUMENT;
    if (len > MSJ_MAX_SEGMENT_BYTES || n >= (1ull << 31)) return MSJ_CAPACITY;
    if ((reinterpret_cast<uintptr_t>(d_idx) & 15u) || (reinterpret_cast<uintptr_t>(d_depth) & 15u) ||
        (reinterpret_cast<uintptr_t>(d_type) & 7u) || (reinterpret_cast<uintptr_t>(d_match) & 15u))  // match[] leaves as 16-byte stores
        return MSJ_ERR_BAD_ARGUMENT;
    if (!hip_ok(hipSetDevice(ctx->device))) return MSJ_ERR_HIP;
    const uint64_t need = msj_stage2_prep_workspace_bytes(n, len, d_pairs ? 2 : (d_match != nullptr ? 1 : 0));  // incl. the fused kernel's chunk aggregates and table
    if (!ensure_tok_ws(ctx, need)) return MSJ_MEMALLOC;
    ctx->tok_doc_n = ~0ull;
    msj_token_opts o = ctx->tok_opts;
    o.d_prev = d_prev;
    o.d_pairs = d_pairs;
    if (msj_launch_tokens(d_buf, len, d_idx, n, d_type, d_depth, d_match, d_result, ctx->tok_ws, stream, o) != 0) return MSJ_ERR_HIP;
    ctx->tok_doc_n = n;
    return MSJ_SUCCESS;
}


static bool ensure_span_fix(msj_ctx *ctx) {
    if (ctx->span_fix) return true;
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&ctx->span_fix), msj_span_fix_bytes()))) return false;
    if (!hip_ok(hipMemset(ctx->span_fix, 0, msj_span_fix_bytes()))) {
        (void)hipFree(ctx->span_fix);
        ctx->span_fix = nullptr;
        return false;
    }
    return true;
}

int32_t msj_token_spans_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n,
                               uint32_t *d_end, uint8_t *d_flags, void *stream) {
    if (!ctx) return MSJ_ERR_BAD_ARGUMENT;
    if (n > 0 && (!d_buf || !d_idx || !d_end || !d_flags)) return MSJ_ERR_BAD_ARGUMENT;
    if (len > MSJ_MAX_SEGMENT_BYTES || n >= (1ull << 31)) return MSJ_CAPACITY;
    if (!hip_ok(hipSetDevice(ctx->device))) return MSJ_ERR_HIP;
    if (!ensure_span_fix(ctx)) return MSJ_MEMALLOC;
    if (!ensure_tok_ws(ctx, msj_stage2_prep_workspace_bytes(n, len, 0))) return MSJ_MEMALLOC;  // the group table lives there
    ctx->tok_doc_n = ~0ull;
    return msj_launch_token_spans(d_buf, len, d_idx, n, d_end, d_flags, ctx->tok_ws, ctx->span_fix, stream, ctx->tok_opts) == 0 ? MSJ_SUCCESS
                                                                                                                              : MSJ_ERR_HIP;
}


int32_t msj_stage2_prep_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n,
                               uint8_t *d_type, int32_t *d_depth, uint32_t *d_match, uint32_t *d_end, uint8_t *d_flags,
                               msj_tokens_result *d_result, void *stream) {
    return msj_stage2_prep_chain_device(ctx, d_buf, len, d_idx, n, d_type, d_depth, d_match, d_end, d_flags, d_result, nullptr, stream);
}

static int32_t prep_chain_impl(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n, uint8_t *d_type,
                               int32_t *d_depth, uint32_t *d_match, uint32_t *d_end, uint8_t *d_flags, msj_tokens_result *d_result,
                               const msj_tokens_result *d_prev, void *stream, uint32_t match_bias, uint32_t *d_resid,
                               msj_bracket_pair *d_pairs = nullptr);

int32_t msj_stage2_prep_pairs_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n,
                                     uint8_t *d_type, int32_t *d_depth, msj_bracket_pair *d_pairs, uint32_t *d_end, uint8_t *d_flags,
                                     msj_tokens_result *d_result, const msj_tokens_result *d_prev, void *stream) {
    if (!d_pairs || (reinterpret_cast<uintptr_t>(d_pairs) & 7u)) return MSJ_ERR_BAD_ARGUMENT;
    return prep_chain_impl(ctx, d_buf, len, d_idx, n, d_type, d_depth, nullptr, d_end, d_flags, d_result, d_prev, stream, 0u, nullptr, d_pairs);
}

int32_t msj_stage2_prep_chain_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n,
                                     uint8_t *d_type, int32_t *d_depth, uint32_t *d_match, uint32_t *d_end, uint8_t *d_flags,
                                     msj_tokens_result *d_result, const msj_tokens_result *d_prev, void *stream) {
    return prep_chain_impl(ctx, d_buf, len, d_idx, n, d_type, d_depth, d_match, d_end, d_flags, d_result, d_prev, stream, 0u, nullptr);
}

// match_bias / d_resid: msj_stage2_prep_segments (partners as positions in the shard's arrays, the call's unpaired brackets kept)
static int32_t prep_chain_impl(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n, uint8_t *d_type,
                               int32_t *d_depth, uint32_t *d_match, uint32_t *d_end, uint8_t *d_flags, msj_tokens_result *d_result,
                               const msj_tokens_result *d_prev, void *stream, uint32_t match_bias, uint32_t *d_resid,
                               msj_bracket_pair *d_pairs) {
    if (!ctx || !d_result || d_prev == d_result) return MSJ_ERR_BAD_ARGUMENT;
    if (n > 0 && (!d_buf || !d_idx || !d_type || !d_depth || !d_end || !d_flags)) return MSJ_ERR_BAD_ARGUMENT;
    if (len > MSJ_MAX_SEGMENT_BYTES || n >= (1ull << 31)) return MSJ_CAPACITY;
    if ((reinterpret_cast<uintptr_t>(d_idx) & 15u) || (reinterpret_cast<uintptr_t>(d_depth) & 15u) ||
        (reinterpret_cast<uintptr_t>(d_type) & 7u) || (reinterpret_cast<uintptr_t>(d_match) & 15u))  // match[] leaves as 16-byte stores
        return MSJ_ERR_BAD_ARGUMENT;
    if (!hip_ok(hipSetDevice(ctx->device))) return MSJ_ERR_HIP;
    const uint64_t need = msj_stage2_prep_workspace_bytes(n, len, d_pairs ? 2 : (d_match != nullptr ? 1 : 0));
    if (!ensure_tok_ws(ctx, need)) return MSJ_MEMALLOC;
    ctx->tok_doc_n = ~0ull;
    if (!ensure_span_fix(ctx)) return MSJ_MEMALLOC;
    msj_token_opts o = ctx->tok_opts;
    o.d_prev = d_prev;
    o.match_bias = match_bias;
    o.d_resid = d_match ? d_resid : nullptr;
    o.d_pairs = d_pairs;
    if (msj_launch_stage2_prep(d_buf, len, d_idx, n, d_type, d_depth, d_match, d_end, d_flags, d_result, ctx->tok_ws, ctx->span_fix, stream, o) != 0)
        return MSJ_ERR_HIP;
    ctx->tok_doc_n = n;
    return MSJ_SUCCESS;
}

int32_t msj_stage2_prep_segments(msj_ctx *ctx, const uint8_t *d_buf, const msj_segment *segments, uint32_t n_segments,
                                 const uint32_t *d_idx, uint8_t *d_type, int32_t *d_depth, uint32_t *d_match, uint32_t *d_end,
                                 uint8_t *d_flags, msj_tokens_result *d_results, const msj_tokens_result *d_prev,
                                 uint64_t *offsets_out, void *stream) {
    if (!ctx || !segments || n_segments == 0 || !d_results || !d_buf) return MSJ_ERR_BAD_ARGUMENT;
    if (!hip_ok(hipSetDevice(ctx->device))) return MSJ_ERR_HIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint64_t begin0 = segments[0].index_begin, base0 = segments[0].byte_base;
    // the whole table is checked before the first launch: a bad entry k must not leave segments 0 .. k-1 on the stream
    for (uint32_t s = 0; s < n_segments; s++) {
        const msj_segment &sg = segments[s];
        if (sg.byte_len == 0 || sg.byte_len > MSJ_MAX_SEGMENT_BYTES || sg.count >= (1ull << 31)) return MSJ_CAPACITY;
        if (s > 0) {  // segments of one shard follow each other without gaps, in bytes and in indices
            const msj_segment &pv = segments[s - 1];
            if (sg.byte_base != pv.byte_base + pv.byte_len || sg.index_begin != pv.index_begin + pv.count) return MSJ_ERR_BAD_ARGUMENT;
        }
    }
    // bracket partners over the whole shard: match[] holds positions in the shard's output arrays (uint32), every
    // segment leaves its unpaired brackets in a residual list of the context's, a stitch pairs them at the end
    msj_stitch_args st_args;
    st_args.n_segments = n_segments;
    if (d_match) {
        if (n_segments > MSJ_STITCH_MAX_SEGMENTS) return MSJ_CAPACITY;
        uint64_t total = 0;
        for (uint32_t s = 0; s < n_segments; s++) total = (((total + segments[s].count + 3u) & ~3ull) + 7u) & ~7ull;
        if (total >= 0xFFFFFFFFull) return MSJ_CAPACITY;  // (0xFFFFFFFF is "no partner")
        const uint64_t need = (uint64_t)n_segments * MSJ_RESID_WORDS;
        if (need > ctx->resid_words) {
            if (ctx->resid) {
                (void)hipDeviceSynchronize();
                (void)hipFree(ctx->resid);
            }
            ctx->resid = nullptr;
            ctx->resid_words = 0;
            if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&ctx->resid), need * sizeof(uint32_t)))) return MSJ_MEMALLOC;
            ctx->resid_words = need;
        }
        if (!hip_ok(hipMemsetAsync(ctx->resid, 0, need * sizeof(uint32_t), st))) return MSJ_ERR_HIP;
    }
    uint64_t off = 0;
    for (uint32_t s = 0; s < n_segments; s++) {
        const msj_segment &sg = segments[s];
        const uint64_t n = sg.count;
        const uint32_t *idx = d_idx + (sg.index_begin - begin0);
        if (n && (reinterpret_cast<uintptr_t>(idx) & 15u)) {
            // stage 1 writes a shard's indices densely, so a later segment's slice starts wherever the one in front
            // ended: the token kernels read index quads, so it is copied to an aligned buffer first (4 bytes per token
            // each way, on the stream; the first segment of a shard never needs it)
            if (n > ctx->seg_idx_words) {
                if (ctx->seg_idx) {
                    (void)hipDeviceSynchronize();
                    (void)hipFree(ctx->seg_idx);
                }
                ctx->seg_idx = nullptr;
                ctx->seg_idx_words = 0;
                if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&ctx->seg_idx), (n + 4) * sizeof(uint32_t)))) return MSJ_MEMALLOC;
                ctx->seg_idx_words = n + 4;
            }
            if (!hip_ok(hipMemcpyAsync(ctx->seg_idx, idx, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st))) return MSJ_ERR_HIP;
            idx = ctx->seg_idx;
        }
        if (offsets_out) offsets_out[s] = off;
        uint32_t *resid = d_match ? ctx->resid + (uint64_t)s * MSJ_RESID_WORDS : nullptr;
        if (d_match) {
            st_args.offsets[s] = (uint32_t)off;
            st_args.resid[s] = resid;
        }
        const int32_t rc = prep_chain_impl(ctx, d_buf + (sg.byte_base - base0), sg.byte_len, idx, n, d_type ? d_type + off : nullptr,
                                           d_depth ? d_depth + off : nullptr, d_match ? d_match + off : nullptr,
                                           d_end ? d_end + off : nullptr, d_flags ? d_flags + off : nullptr, &d_results[s],
                                           s == 0 ? d_prev : &d_results[s - 1], stream, (uint32_t)off, resid);
        if (rc != MSJ_SUCCESS) return rc;
        off += (n + 3u) & ~3ull;  // every segment's slices start 16-byte aligned (8 for the byte arrays: n rounded to 4 ... 8 below)
        off = (off + 7u) & ~7ull;
    }
    if (d_match && msj_launch_stitch_partners(st_args, d_match, d_results, d_prev, stream) != 0) return MSJ_ERR_HIP;
    return MSJ_SUCCESS;
}

extern "C" uint64_t msj_documents_workspace_bytes(uint64_t n);
extern "C" int msj_launch_documents(const uint8_t *d_buf, uint64_t len, int is_final, const uint32_t *d_idx, uint64_t n,
                                    const uint8_t *d_type, const int32_t *d_depth, const msj_carry *d_carry, uint32_t *d_doc_first, uint64_t capacity,
                                    msj_documents_result *d_result, void *d_ws, const void *d_block_agg, void *stream);

int32_t msj_documents_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, int32_t is_final, const uint32_t *d_idx,
                             uint64_t n, const uint8_t *d_type, const int32_t *d_depth, const msj_carry *d_carry,
                             uint32_t *d_doc_first, uint64_t capacity, msj_documents_result *d_result, void *stream) {
    if (!ctx || !d_result) return MSJ_ERR_BAD_ARGUMENT;
    if (n > 0 && (!d_buf || !d_idx || !d_type || !d_depth)) return MSJ_ERR_BAD_ARGUMENT;
    if (capacity > 0 && !d_doc_first) return MSJ_ERR_BAD_ARGUMENT;
    if (len > MSJ_MAX_SEGMENT_BYTES || n >= (1ull << 31)) return MSJ_CAPACITY;
    if ((reinterpret_cast<uintptr_t>(d_depth) & 15u) || (reinterpret_cast<uintptr_t>(d_type) & 7u)) return MSJ_ERR_BAD_ARGUMENT;
    if (!hip_ok(hipSetDevice(ctx->device))) return MSJ_ERR_HIP;
    const uint64_t need = msj_documents_workspace_bytes(n);
    if (need > ctx->doc_ws_bytes) {
        if (ctx->doc_ws) {
            (void)hipDeviceSynchronize();
            (void)hipFree(ctx->doc_ws);
        }
        ctx->doc_ws = nullptr;
        ctx->doc_ws_bytes = 0;
        if (!hip_ok(hipMalloc(&ctx->doc_ws, need + need / 4))) return MSJ_MEMALLOC;
        ctx->doc_ws_bytes = need + need / 4;
    }
    // MSJ_DOCS_AFTER_TOKENS: the block aggregates the token pre-pass left in its workspace are for these arrays
    const void *pre = ((is_final & MSJ_DOCS_AFTER_TOKENS) && ctx->tok_ws && ctx->tok_doc_n == n && n > 0)
                          ? msj_tokens_doc_aggregates(ctx->tok_ws, n)
                          : nullptr;
    return msj_launch_documents(d_buf, len, is_final & MSJ_DOCS_FINAL, d_idx, n, d_type, d_depth, d_carry, d_doc_first, capacity, d_result, ctx->doc_ws,
                                pre, stream) == 0
               ? MSJ_SUCCESS
               : MSJ_ERR_HIP;
}

int32_t msj_device_alloc(msj_ctx *ctx, uint64_t bytes, void **d_out) {
    if (!ctx || !d_out) return MSJ_ERR_BAD_ARGUMENT;
    *d_out = nullptr;
    if (!hip_ok(hipSetDevice(ctx->device))) return MSJ_ERR_HIP;
    return hip_ok(hipMalloc(d_out, bytes ? bytes : 1)) ? MSJ_SUCCESS : MSJ_MEMALLOC;
}

int32_t msj_device_free(msj_ctx *ctx, void *d_ptr) {
    if (!ctx) return MSJ_ERR_BAD_ARGUMENT;
    if (!d_ptr) return MSJ_SUCCESS;
    if (!hip_ok(hipSetDevice(ctx->device))) return MSJ_ERR_HIP;
    return hip_ok(hipFree(d_ptr)) ? MSJ_SUCCESS : MSJ_ERR_HIP;
}

static int32_t blocking_copy(msj_ctx *ctx, void *dst, const void *src, uint64_t bytes, void *stream, hipMemcpyKind kind) {
    if (!ctx || (bytes && (!dst || !src))) return MSJ_ERR_BAD_ARGUMENT;
    if (bytes == 0) return MSJ_SUCCESS;
    if (!hip_ok(hipSetDevice(ctx->device))) return MSJ_ERR_HIP;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!hip_ok(hipMemcpyAsync(dst, src, bytes, kind, s))) return MSJ_ERR_HIP;
    return hip_ok(hipStreamSynchronize(s)) ? MSJ_SUCCESS : MSJ_ERR_HIP;
}

int32_t msj_copy_to_device(msj_ctx *ctx, void *d_dst, const void *src, uint64_t bytes, void *stream) {
    return blocking_copy(ctx, d_dst, src, bytes, stream, hipMemcpyHostToDevice);
}

int32_t msj_copy_to_host(msj_ctx *ctx, void *dst, const void *d_src, uint64_t bytes, void *stream) {
    return blocking_copy(ctx, dst, d_src, bytes, stream, hipMemcpyDeviceToHost);
}

int32_t msj_carry_fetch(msj_ctx *ctx, const msj_carry *d_carry, msj_carry *host_out, void *stream) {
    if (!ctx || !d_carry || !host_out) return MSJ_ERR_BAD_ARGUMENT;
    if (!hip_ok(hipSetDevice(ctx->device))) return MSJ_ERR_HIP;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!hip_ok(hipMemcpyAsync(host_out, d_carry, sizeof(msj_carry), hipMemcpyDeviceToHost, s)))
        return MSJ_ERR_HIP;
    if (!hip_ok(hipStreamSynchronize(s))) return MSJ_ERR_HIP;
    if (host_out->internal_error && ctx->last.valid && ctx->last.d_carry_out == d_carry &&
        !(ctx->last.flags & MSJ_FLAG_TWO_PASS)) {
        // a wait inside the single-pass kernel ran into its bound (a starved GPU, a stalled resolver): the
        // result is poisoned.  Run the call again through the two-pass kernels, which wait for nothing.
        const auto L = ctx->last;
        ctx->fallbacks++;
        ctx->ws_dirty[0] = ctx->ws_dirty[1] = kAllDirty;  // the poisoned launch may have left anything behind
        const int32_t rc = enqueue_shard(ctx, L.d_buf, L.len, L.d_idx, L.idx_capacity, L.d_carry_in, L.d_carry_out,
                                         L.d_segments, L.max_segments, nullptr, L.has_prefix, L.is_final, L.no_emit,
                                         L.trailer_len, L.stream, L.flags | MSJ_FLAG_TWO_PASS, 0,
                                         L.by_value ? &L.carry_bits : nullptr);
        if (rc != MSJ_SUCCESS) return rc;
        if (!hip_ok(hipStreamSynchronize(L.stream))) return MSJ_ERR_HIP;
        if (!hip_ok(hipMemcpyAsync(host_out, d_carry, sizeof(msj_carry), hipMemcpyDeviceToHost, s)) ||
            !hip_ok(hipStreamSynchronize(s)))
            return MSJ_ERR_HIP;
    }
    return MSJ_SUCCESS;
}

static msj_ctx *default_ctx_locked();
int32_t msj_host_placement(msj_ctx *ctx, char *out, uint64_t capacity) {
    if (!out || capacity == 0) return MSJ_ERR_BAD_ARGUMENT;
    std::unique_lock<std::mutex> lock(g_default_mutex, std::defer_lock);
    if (!ctx) {
        lock.lock();
        ctx = default_ctx_locked();
        if (!ctx) return MSJ_ERR_NO_DEVICE;
    }
    const GpuHostLocality g = ctx->pipe ? ctx->pipe->where : gpu_locality(ctx->device);
    const int ring_in = ctx->pipe ? numa_node_of(ctx->pipe->pin_in[0]) : -1, ring_out = ctx->pipe ? numa_node_of(ctx->pipe->pin_out[0]) : -1;
    int nodes = 0;
    for (;; nodes++) {
        char path[64];
        std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d", nodes);
        if (access(path, F_OK) != 0) break;
    }
    const int n = std::snprintf(out, (size_t)capacity,
                                "{\"gpu_pci\": \"%s\", \"gpu_numa_node\": %d, \"numa_nodes\": %d, \"gpu_local_cpus_usable\": %d, "
                                "\"pipeline_created\": %s, \"copy_threads\": %d, \"copy_threads_bound_to_gpu_node\": %d, "
                                "\"ring_in_node\": %d, \"ring_out_node\": %d, \"pcie_link_speed\": \"%s\", \"pcie_link_width\": \"%s\"}",
                                g.pci, g.node, nodes, g.n_cpus, ctx->pipe ? "true" : "false", ctx->pipe ? ctx->pipe->kCopyThreads : 0,
                                ctx->pipe ? ctx->pipe->pool.bound : 0, ring_in, ring_out, g.link_speed, g.link_width);
    return (n < 0 || (uint64_t)n >= capacity) ? MSJ_CAPACITY : MSJ_SUCCESS;
}

int32_t msj_debug_numa_node_of(const void *host_ptr) { return numa_node_of(host_ptr); }

int32_t msj_debug_set_wait_ticks(msj_ctx *ctx, uint32_t ticks) {
    if (!ctx) return MSJ_ERR_BAD_ARGUMENT;
    ctx->wait_ticks = ticks;
    return MSJ_SUCCESS;
}

static msj_ctx *default_ctx_locked() {  // g_default_mutex held
    if (!g_default_ctx && msj_ctx_create(0, &g_default_ctx) != MSJ_SUCCESS) return nullptr;
    return g_default_ctx;
}

int32_t msj_host_register(msj_ctx *ctx, void *ptr, uint64_t bytes) {
    if (!ptr || bytes == 0) return MSJ_ERR_BAD_ARGUMENT;
    std::unique_lock<std::mutex> lock(g_default_mutex, std::defer_lock);
    if (!ctx) {
        lock.lock();
        ctx = default_ctx_locked();
        if (!ctx) return MSJ_ERR_NO_DEVICE;
    }
    if (!hip_ok(hipSetDevice(ctx->device))) return MSJ_ERR_HIP;
    if (!hip_ok(hipHostRegister(ptr, bytes, hipHostRegisterDefault))) {
        (void)hipGetLastError();
        return MSJ_ERR_HIP;
    }
    ctx->pinned.push_back({static_cast<const uint8_t *>(ptr), bytes});
    return MSJ_SUCCESS;
}

int32_t msj_host_unregister(msj_ctx *ctx, void *ptr) {
    if (!ptr) return MSJ_ERR_BAD_ARGUMENT;
    std::unique_lock<std::mutex> lock(g_default_mutex, std::defer_lock);
    if (!ctx) {
        lock.lock();
        ctx = g_default_ctx;
        if (!ctx) return MSJ_ERR_BAD_ARGUMENT;
    }
    for (size_t i = 0; i < ctx->pinned.size(); i++)
        if (ctx->pinned[i].base == ptr) {
            ctx->pinned.erase(ctx->pinned.begin() + (long)i);
            (void)hipSetDevice(ctx->device);
            (void)hipDeviceSynchronize();  // nothing of ours may still be moving bytes of the range
            return hip_ok(hipHostUnregister(ptr)) ? MSJ_SUCCESS : MSJ_ERR_HIP;
        }
    return MSJ_ERR_BAD_ARGUMENT;
}

uint64_t msj_fallback_count(const msj_ctx *ctx) { return ctx ? ctx->fallbacks : 0; }

int32_t msj_debug_set_pipeline_min_bytes(msj_ctx *ctx, uint64_t bytes) {
    std::unique_lock<std::mutex> lock(g_default_mutex, std::defer_lock);
    if (!ctx) {
        lock.lock();
        ctx = default_ctx_locked();
        if (!ctx) return MSJ_ERR_NO_DEVICE;
    }
    ctx->pipeline_min = bytes ? bytes : kPipelineMinDefault;
    return MSJ_SUCCESS;
}

int32_t msj_debug_fail_pipeline_setup(msj_ctx *ctx, int32_t on) {
    std::unique_lock<std::mutex> lock(g_default_mutex, std::defer_lock);
    if (!ctx) {
        lock.lock();
        ctx = default_ctx_locked();
        if (!ctx) return MSJ_ERR_NO_DEVICE;
    }
    ctx->pipe_fail_setup = on != 0;
    if (!on) ctx->pipe_unavailable = false;
    return ctx->pipe_unavailable ? 1 : 0;
}

int32_t msj_debug_set_span_limits(msj_ctx *ctx, uint32_t lds_limit_bytes, uint32_t fix_capacity) {
    if (!ctx) return MSJ_ERR_BAD_ARGUMENT;
    ctx->tok_opts.lds_limit = lds_limit_bytes;
    ctx->tok_opts.fix_cap = fix_capacity;
    return MSJ_SUCCESS;
}
int32_t msj_debug_set_span_mode(msj_ctx *ctx, uint32_t mode) {
    if (!ctx || mode > 2u) return MSJ_ERR_BAD_ARGUMENT;
    ctx->tok_opts.span_mode = mode;
    return MSJ_SUCCESS;
}

int32_t msj_debug_set_segment_bytes(msj_ctx *ctx, uint64_t bytes) {
    if (!ctx || bytes == 0 || bytes % msj::kTileBytes != 0 || bytes > msj::kSegmentBytes) return MSJ_ERR_BAD_ARGUMENT;
    ctx->seg_bytes = bytes;
    return MSJ_SUCCESS;
}

int32_t msj_stage1_ctx(msj_ctx *ctx, const uint8_t *buf, uint64_t len, uint32_t *idx_out,
                       uint64_t idx_capacity, uint64_t *n_out, int32_t *utf8_verdict_out,
                       uint32_t flags) {
    if (!ctx) return MSJ_ERR_BAD_ARGUMENT;
    if (len == 0) return MSJ_EMPTY;  // json_structural_indexer.mojo:91-92
    if (!buf || !idx_out || !n_out) return MSJ_ERR_BAD_ARGUMENT;
    if (len > MSJ_MAX_SEGMENT_BYTES) return MSJ_CAPACITY;
    if (!hip_ok(hipSetDevice(ctx->device))) return MSJ_ERR_HIP;

    // Small inputs: latency, not bandwidth.  Through pinned staging everything is enqueued at once -- input up,
    // kernel, result and ALL len + 3 index slots down in one copy (n is not known yet; 4 bytes per input byte
    // is cheap at this size) -- and the host waits once instead of three times (pageable copies are synchronous).
    if (len <= kSmallInput && idx_capacity >= len + 3) {
        if (!ctx->h_pin && !hip_ok(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_pin), kPinBytes, hipHostMallocDefault))) ctx->h_pin = nullptr;
        if (!ctx->d_small && !hip_ok(hipMalloc(reinterpret_cast<void **>(&ctx->d_small), sizeof(msj_carry)))) ctx->d_small = nullptr;
    }
    if (len <= kSmallInput && idx_capacity >= len + 3 && ctx->h_pin && ctx->d_small) {
        // pinned: [0, kSmallInput): input | [kSmallInput, +64): msj_carry | then len + 3 indices.  The kernel reads the
        // input from there and writes the indices there; the 64-byte result lives in device memory (the kernel
        // updates it with atomics) and comes down by the one copy of the call
        std::memcpy(ctx->h_pin, buf, len);
        msj_carry *d_res = reinterpret_cast<msj_carry *>(ctx->d_small);
        uint32_t *h_ix = reinterpret_cast<uint32_t *>(ctx->h_pin + kSmallInput + 64);
        int32_t rc = msj_stage1_device(ctx, ctx->h_pin, len, h_ix, len + 3, d_res, nullptr, flags);
        if (rc != MSJ_SUCCESS) return rc;
        if (!hip_ok(hipMemcpyAsync(ctx->h_pin + kSmallInput, d_res, sizeof(msj_carry), hipMemcpyDeviceToHost, nullptr)) ||
            !hip_ok(hipStreamSynchronize(nullptr)))
            return MSJ_ERR_HIP;
        const msj_carry res = *reinterpret_cast<const msj_carry *>(ctx->h_pin + kSmallInput);
        if (res.internal_error && !(flags & MSJ_FLAG_TWO_PASS)) {
            // an expired wait in the single-pass kernel: once more through the two-pass kernels
            ctx->fallbacks++;
            ctx->ws_dirty[0] = ctx->ws_dirty[1] = kAllDirty;
            return msj_stage1_ctx(ctx, buf, len, idx_out, idx_capacity, n_out, utf8_verdict_out, flags | MSJ_FLAG_TWO_PASS);
        }
        if (utf8_verdict_out) *utf8_verdict_out = res.utf8_error ? MSJ_UTF8_ERROR : MSJ_SUCCESS;
        if (res.code == MSJ_UNCLOSED_STRING || res.code == MSJ_UNESCAPED_CHARS || res.code == MSJ_UNEXPECTED_ERROR ||
            res.code == MSJ_CAPACITY)
            return res.code;  // the reference returns before it sets n or the trailer (:151-158)
        std::memcpy(idx_out, ctx->h_pin + kSmallInput + 64, (res.count + 3) * sizeof(uint32_t));
        *n_out = res.count;
        return res.code;
    }
    // device staging (the reference's allocate(len), dom_parser_implementation.mojo:85-89)
    const uint64_t in_bytes = (len + 63u) & ~63ull;
    if (in_bytes > ctx->d_in_bytes) {
        if (ctx->d_in) (void)hipFree(ctx->d_in);
        ctx->d_in = nullptr;
        ctx->d_in_bytes = 0;
        if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&ctx->d_in), in_bytes))) return MSJ_MEMALLOC;
        ctx->d_in_bytes = in_bytes;
    }
    const uint64_t dev_cap = idx_capacity < len + 3 ? idx_capacity : len + 3;
    if (dev_cap > ctx->d_idx_words) {
        if (ctx->d_idx) (void)hipFree(ctx->d_idx);
        ctx->d_idx = nullptr;
        ctx->d_idx_words = 0;
        if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&ctx->d_idx), dev_cap * sizeof(uint32_t))))
            return MSJ_MEMALLOC;
        ctx->d_idx_words = dev_cap;
    }
    msj_carry res;
    int32_t rc;
    static const bool pipe_off = knob_set("MSJ_PIPE_DISABLE");  // measurement build: the plain staging path
    bool piped = len >= ctx->pipeline_min && !(flags & MSJ_FLAG_TWO_PASS) && !pipe_off && !ctx->pipe_unavailable;
    if (piped) {
        // large inputs: pinned rings, chunks, both PCIe directions and the kernel at once
        try {
            rc = host_pipeline(ctx, buf, len, idx_out, dev_cap, flags, &res);
        } catch (...) {  // std::thread / std::vector could not get what they need: nothing of ours crosses the C boundary
            rc = MSJ_MEMALLOC;
        }
        if (rc == kPipeUnavailable) {
            // the machinery cannot be had on this host: the plain staging path below, now and from now on
            (void)hipGetLastError();
            ctx->pipe_unavailable = true;
            piped = false;
        } else if (rc != MSJ_SUCCESS) {
            return rc;
        }
    }
    if (piped) {
        if (res.internal_error) {  // an expired wait somewhere in the chain: once more, two-pass, plain staging
            ctx->fallbacks++;
            ctx->ws_dirty[0] = ctx->ws_dirty[1] = kAllDirty;
            return msj_stage1_ctx(ctx, buf, len, idx_out, idx_capacity, n_out, utf8_verdict_out, flags | MSJ_FLAG_TWO_PASS);
        }
    } else {
        if (!hip_ok(hipMemcpy(ctx->d_in, buf, len, hipMemcpyHostToDevice))) return MSJ_ERR_HIP;
        rc = msj_stage1_device(ctx, ctx->d_in, len, ctx->d_idx, dev_cap, ctx->d_result, nullptr, flags);
        if (rc != MSJ_SUCCESS) return rc;
        rc = msj_carry_fetch(ctx, ctx->d_result, &res, nullptr);
        if (rc != MSJ_SUCCESS) return rc;
    }
    if (utf8_verdict_out) *utf8_verdict_out = res.utf8_error ? MSJ_UTF8_ERROR : MSJ_SUCCESS;
    // On UNCLOSED_STRING / UNESCAPED_CHARS the reference returns before it sets
    // n_structural_indexes or the trailer (json_structural_indexer.mojo:151-158).
    if (res.code == MSJ_UNCLOSED_STRING || res.code == MSJ_UNESCAPED_CHARS ||
        res.code == MSJ_UNEXPECTED_ERROR || res.code == MSJ_CAPACITY)
        return res.code;
    const uint64_t n = res.count;
    if (!piped && !hip_ok(hipMemcpy(idx_out, ctx->d_idx, (n + 3) * sizeof(uint32_t), hipMemcpyDeviceToHost)))
        return MSJ_ERR_HIP;  // (the pipeline has brought the indices and the trailer down already)
    *n_out = n;
    return res.code;
}

#ifdef MSJ_STAMPS
void msj_debug_set_stamps(uint64_t *d_stamps) { g_stamps = d_stamps; }
#endif

int32_t msj_stage1(const uint8_t *buf, uint64_t len, uint32_t *idx_out, uint64_t idx_capacity,
                   uint64_t *n_out, int32_t *utf8_verdict_out, uint32_t flags) {
    std::lock_guard<std::mutex> lock(g_default_mutex);
    if (!g_default_ctx) {
        const int32_t rc = msj_ctx_create(0, &g_default_ctx);
        if (rc != MSJ_SUCCESS) return rc;
    }
    return msj_stage1_ctx(g_default_ctx, buf, len, idx_out, idx_capacity, n_out, utf8_verdict_out,
                          flags);
}

}  // extern "C"
