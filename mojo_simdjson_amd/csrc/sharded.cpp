// sharded.cpp -- the N-GPU form of the stage-1 path (SURVEY.md section 8b `msj_stage1_sharded`, 8e):
// one contiguous byte-range shard of one stream per rank (one process per GPU).
//
// The reference has no parallelism of any kind; the only state its stage 1 carries from block to
// block is three bits and a count (json_escape_scanner.mojo:13, json_string_scanner.mojo:49,
// json_scanner.mojo:57, json_structural_indexer.mojo:34).  A shard therefore needs the state at
// its first byte, and the stream needs ONE tiny exchange:
//   1. a rank derives (next_is_escaped, prev_scalar) from the 64 stream bytes in front of its shard
//      (pure byte inspection) and SPECULATES in_string from the context of the first unescaped quote
//      (msj_shard_speculate);
//   2. one single-pass kernel launch per shard with that carry-in;
//   3. ONE all-gather of a 128-byte report (carry used | carry out) per rank -- ncclAllGather over
//      RCCL/xGMI in production (msj_exchange_rccl), any callback in tests: every rank replays the chain
//      (msj_shard_verify), which proves or refutes every speculation at once; only the refuted ranks
//      index again, with the now exact carry, the others contribute their cached report to the next
//      all-gather.  The result is always exact; the guess only decides how often step 3 repeats.
// No bulk data crosses xGMI; index arrays stay shard-local.
//
// Everything device-side goes through msj_sharded_ops, so that the protocol itself (submit / result /
// re-run loop) also runs on a CPU-only box under test (tests/test_sharded_cpu.py: a fake shard runner
// that follows the serial spec, gloo as the exchange).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <new>

#include "../../include/msj_stage1.h"

namespace {

constexpr uint32_t kDepth = 3;  // submissions that may be in flight per rank
constexpr uint64_t kLook = 64 + 4096;  // the first look at a shard the caller gave no carry for: halo + head

bool is_nonscalar(uint8_t c) {
    // op | ws, haswell.mojo:22-74 (effective sets): {09,0A,0D,20} and {0C,1A,2C,3A,5B,5D,7B,7D}
    switch (c) {
        case 0x09: case 0x0A: case 0x0D: case 0x20: case 0x0C: case 0x1A:
        case 0x2C: case 0x3A: case 0x5B: case 0x5D: case 0x7B: case 0x7D:
            return true;
        default:
            return false;
    }
}

// a byte a valid document can hold outside of strings: blanks, operators, number characters, the letters of the
// three literals
bool plausible_outside(uint8_t c) {
    switch (c) {
        case 0x09: case 0x0A: case 0x0D: case 0x20: case ',': case ':': case '[': case ']': case '{': case '}':
        case '0': case '1': case '2': case '3': case '4': case '5': case '6': case '7': case '8': case '9':
        case '-': case '+': case '.': case 'e': case 'E':
        case 't': case 'r': case 'u': case 'f': case 'a': case 'l': case 's': case 'n':
            return true;
        default:
            return false;
    }
}

// length of the backslash run ending just before position pos of p[0..), -1 if it reaches p[0]
int64_t run_before(const uint8_t *p, uint64_t pos) {
    int64_t n = 0;
    while (pos > 0 && p[pos - 1] == 0x5C) {
        pos--;
        n++;
    }
    return pos > 0 ? n : -1;
}

// ---- default device operations: HIP
int32_t hip_alloc(void *, uint64_t bytes, int pinned_host, void **out) {
    const hipError_t e = pinned_host ? hipHostMalloc(out, bytes, hipHostMallocDefault) : hipMalloc(out, bytes);
    if (e != hipSuccess) return MSJ_MEMALLOC;
    // zeroed: a report whose `used` half no launch ever wrote (the carry goes by value) must not hold stale bytes
    if (pinned_host) {
        std::memset(*out, 0, bytes);
    } else if (hipMemset(*out, 0, bytes) != hipSuccess) {
        (void)hipFree(*out);
        *out = nullptr;
        return MSJ_ERR_HIP;
    }
    return MSJ_SUCCESS;
}
void hip_free(void *, void *p, int pinned_host) {
    if (!p) return;
    if (pinned_host)
        (void)hipHostFree(p);
    else
        (void)hipFree(p);
}
int32_t hip_copy(void *, void *dst, const void *src, uint64_t bytes, int to_host, void *stream) {
    return hipMemcpyAsync(dst, src, bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice,
                          static_cast<hipStream_t>(stream)) == hipSuccess
               ? MSJ_SUCCESS
               : MSJ_ERR_HIP;
}
int32_t hip_sync(void *, void *stream) {
    return hipStreamSynchronize(static_cast<hipStream_t>(stream)) == hipSuccess ? MSJ_SUCCESS : MSJ_ERR_HIP;
}
// events (optional part of msj_sharded_ops): what lets a result wait for ITS submission only
int32_t hip_event_create(void *, void **out) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return MSJ_ERR_HIP;
    *out = e;
    return MSJ_SUCCESS;
}
void hip_event_destroy(void *, void *e) {
    if (e) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
}
int32_t hip_event_record(void *, void *e, void *stream) {
    return hipEventRecord(static_cast<hipEvent_t>(e), static_cast<hipStream_t>(stream)) == hipSuccess ? MSJ_SUCCESS : MSJ_ERR_HIP;
}
int32_t hip_event_wait(void *, void *e) {
    return hipEventSynchronize(static_cast<hipEvent_t>(e)) == hipSuccess ? MSJ_SUCCESS : MSJ_ERR_HIP;
}
int32_t hip_stream_wait(void *, void *stream, void *e) {
    return hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(e), 0) == hipSuccess ? MSJ_SUCCESS
                                                                                                            : MSJ_ERR_HIP;
}
int32_t hip_event_query(void *, void *e) {
    const hipError_t r = hipEventQuery(static_cast<hipEvent_t>(e));
    if (r == hipSuccess) return 1;
    if (r == hipErrorNotReady) {
        (void)hipGetLastError();  // "not ready" is an answer, not a sticky error
        return 0;
    }
    return MSJ_ERR_HIP;
}
int32_t hip_event_elapsed_ns(void *, void *a, void *b, uint64_t *out) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, static_cast<hipEvent_t>(a), static_cast<hipEvent_t>(b)) != hipSuccess) return MSJ_ERR_HIP;
    *out = ms > 0.f ? (uint64_t)((double)ms * 1e6) : 0;
    return MSJ_SUCCESS;
}
int32_t hip_run_shard(void *user, const uint8_t *d_shard, uint64_t len, uint32_t *d_idx, uint64_t cap,
                      const msj_carry *d_in, msj_carry *d_out, msj_segment *d_segments, uint32_t max_segments,
                      int32_t has_prefix, int32_t is_final, uint64_t trailer_len, void *stream, uint32_t flags) {
    return msj_stage1_shard_device(static_cast<msj_ctx *>(user), d_shard, len, d_idx, cap, d_in, d_out, d_segments,
                                   max_segments, nullptr, has_prefix, is_final, 0, trailer_len, stream, flags);
}

// ---- RCCL exchange: ncclAllGather resolved at run time (the library does not link librccl)
typedef int (*nccl_allgather_t)(const void *, void *, size_t, int, void *, void *);
struct RcclExchange {
    void *comm;
    nccl_allgather_t allgather;
};
int32_t rccl_allgather(void *comm, const void *d_send, void *d_recv, uint64_t bytes_per_rank, void *stream) {
    RcclExchange *x = static_cast<RcclExchange *>(comm);
    // ncclUint8 = 1 (rccl.h ncclDataType_t); enqueued on the caller's stream, behind the kernel
    return x->allgather(d_send, d_recv, (size_t)bytes_per_rank, 1, x->comm, stream) == 0 ? MSJ_SUCCESS : MSJ_ERR_HIP;
}

struct Slot {
    msj_shard_report *d_mine = nullptr;      // device: this rank's report (used | out)
    msj_shard_report *d_gathered = nullptr;  // device: world reports
    msj_shard_report *h_gathered = nullptr;  // pinned host copy of d_gathered
    msj_carry *h_spec = nullptr;             // pinned host: the carry this rank's launch used
    bool busy = false;
    // with event operations (the default HIP ones have them): the start and the end of the round's kernel on the
    // submission's stream, the arrival of the gathered reports in pinned memory on the exchange's stream.  A result
    // waits for ev_stitch of ITS slot and for nothing else.
    void *ev_start = nullptr, *ev_kernel = nullptr, *ev_stitch = nullptr;
    bool kernel_ran = false;  // this round launched a kernel (a round in which only the other ranks index again does not)
    uint64_t looked = 0;      // bytes of `look` the submission read to guess its carry (0: the caller gave one)
    uint8_t look[kLook];
    // the call, for re-runs
    const uint8_t *d_shard;
    uint64_t shard_len, idx_capacity, total_len;
    uint32_t *d_idx;
    msj_segment *d_segments;
    uint32_t max_segments, flags;
    int32_t has_prefix;
    void *stream;
};

}  // namespace

struct msj_sharded {
    msj_sharded_ops ops;
    msj_exchange x;
    Slot slots[kDepth];
    uint32_t next = 0;
    uint64_t reruns = 0, rounds = 0, results = 0, reruns_behind_queue = 0;
    uint64_t stitch_device_ns = 0, result_wait_ns = 0, kernel_device_ns = 0;
    uint64_t last_kernel_ns = 0, last_stitch_ns = 0;
    bool hip_default = false; // the default HIP operations: the carry goes into the launch by value, the report's `used`
                              // comes back as the echo in carry_out.reserved[0] -- no copy in front of the kernel
    bool events = false;      // ops has the event operations: per-slot waits, exchange beside the next kernel
    void *side = nullptr;     // the stream the exchange and the read-back are enqueued on (NULL: the submission's own)
    bool owns_side = false;   // created here (default HIP operations)
    RcclExchange *owned_rccl = nullptr;
    // What the last verified result proved about a shard the library had to look at itself (no `speculation`), kept
    // with the bytes it looked at first (64-byte halo + 4 KiB head): when the same placed shard comes again, those
    // bytes are unchanged and they decide nothing by themselves, the proven carry is the guess and the longer looks
    // (64 KiB, 1 MiB) are not repeated.  Only ever a guess: the chain check decides.
    struct {
        const uint8_t *d_shard = nullptr;
        uint64_t shard_len = 0, n = 0;
        msj_carry carry;
        uint8_t bytes[kLook];
    } proven;
};

extern "C" {

// (next_is_escaped, prev_scalar) after the last byte of halo[0..halo_len), from those bytes alone
// (json_escape_scanner.mojo:18-45, json_scanner.mojo:64-79: a byte is escaped iff an odd run of
// backslashes precedes it; prev_scalar = the last byte is a scalar character that is not a real quote).
// Returns 0 and sets *decided = 0 when the bytes cannot decide (a backslash run reaches halo[0]).
static void halo_state(const uint8_t *halo, uint64_t n, uint32_t *e, uint32_t *ps, int *decided) {
    *e = 0;
    *ps = 0;
    *decided = 1;
    if (n == 0) return;
    const int64_t r = run_before(halo, n);
    if (r < 0) {
        *decided = 0;
        return;
    }
    if (r >= 1) {
        *e = (uint32_t)(r & 1);
        *ps = 1;
        return;
    }
    const uint8_t c = halo[n - 1];
    if (is_nonscalar(c)) return;
    if (c != 0x22) {
        *ps = 1;
        return;
    }
    const int64_t r2 = run_before(halo, n - 1);
    if (r2 < 0) {
        *decided = 0;
        return;
    }
    *ps = (uint32_t)(r2 & 1);  // an escaped quote is a non-quote scalar
}

int32_t msj_shard_speculate(const uint8_t *halo, uint64_t halo_len, const uint8_t *head, uint64_t head_len,
                            msj_carry *out) {
    return msj_shard_speculate_ex(halo, halo_len, head, head_len, out, nullptr);
}

int32_t msj_shard_speculate_ex(const uint8_t *halo, uint64_t halo_len, const uint8_t *head, uint64_t head_len,
                               msj_carry *out, int32_t *decided_out) {
    if (!out || (halo_len && !halo) || (head_len && !head)) return MSJ_ERR_BAD_ARGUMENT;
    std::memset(out, 0, sizeof *out);
    if (decided_out) *decided_out = 1;
    if (halo_len == 0) return MSJ_SUCCESS;  // start of the stream: the all-zero state
    uint32_t e, ps;
    int decided;
    halo_state(halo, halo_len, &e, &ps, &decided);
    if (!decided) {  // >= 64 backslashes in front of the shard: any guess; the chain check settles it
        e = 0;
        ps = 1;
    }
    out->next_is_escaped = e;
    out->prev_scalar = ps;
    // in_string, first try: follow the head under both hypotheses at once (they are each other's complement at
    // every byte) until one of them meets a byte it cannot hold -- a control character inside a string, or outside
    // strings anything but blanks, operators, number characters and the letters of true / false / null.  A valid
    // document never contradicts the true hypothesis; text, keys and UTF-8 contradict the false one within a few
    // bytes.  Only a guess all the same (both or neither may fail): the chain check decides.
    {
        uint32_t esc = e, in0 = 0;  // in0: inside a string if the shard starts outside of one
        for (uint64_t i = 0; i < head_len; i++) {
            const uint8_t c = head[i];
            const uint32_t escaped = esc;
            esc = (!escaped && c == 0x5C) ? 1u : 0u;
            if (escaped) continue;  // whatever it is, it is taken literally
            if (c == 0x22) {
                in0 ^= 1u;
                continue;
            }
            const bool bad_in = c < 0x20;
            const bool bad_out = !plausible_outside(c);
            const bool h0_bad = in0 ? bad_in : bad_out, h1_bad = in0 ? bad_out : bad_in;
            if (h0_bad != h1_bad) {
                out->in_string = h0_bad ? 1u : 0u;
                return MSJ_SUCCESS;
            }
            if (h0_bad) break;  // neither holds: not a valid document, any guess
        }
    }
    // second try: the first unescaped quote of the shard opens a string if one of `: , [ {` precedes it,
    // closes one if one of `: , ] }` follows it (blanks skipped).  Nothing in the head contradicted either
    // hypothesis: the caller may want to look at more bytes before it relies on this.
    if (decided_out) *decided_out = 0;
    uint32_t esc = e;
    int64_t q = -1;
    for (uint64_t i = 0; i < head_len; i++) {
        const uint32_t escaped = esc;
        if (escaped)
            esc = 0;
        else if (head[i] == 0x5C)
            esc = 1;
        if (head[i] == 0x22 && !escaped) {
            q = (int64_t)i;
            break;
        }
    }
    if (q < 0) return MSJ_SUCCESS;
    auto at = [&](int64_t k) -> int {  // byte k of halo + head, k relative to the shard start
        if (k < 0) return -k <= (int64_t)halo_len ? halo[(int64_t)halo_len + k] : -1;
        return k < (int64_t)head_len ? head[k] : -1;
    };
    auto blank = [](int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; };
    int64_t k = q - 1;
    while (blank(at(k))) k--;
    int c = at(k);
    if (c == ':' || c == ',' || c == '[' || c == '{') return MSJ_SUCCESS;  // opens: outside
    k = q + 1;
    while (blank(at(k))) k++;
    c = at(k);
    if (c == ':' || c == ',' || c == ']' || c == '}') out->in_string = 1;  // closes: inside
    return MSJ_SUCCESS;
}

int32_t msj_shard_verify(const msj_shard_report *reports, uint32_t world, msj_carry *exact_in, uint64_t *rerun_mask) {
    if (!reports || !exact_in || !rerun_mask || world == 0 || world > 64) return MSJ_ERR_BAD_ARGUMENT;
    // A shard's quote parity is out.in_string ^ used.in_string whatever used.in_string was, and its
    // next_is_escaped / prev_scalar out do not depend on the string state at all: the chain can be
    // replayed through ranks that guessed in_string wrong (they only have to index again).  It stops at a
    // rank whose escape carries were wrong or whose launch was poisoned: what that rank reports is not
    // reliable, so nothing behind it can be judged yet.
    uint64_t mask = 0;
    uint32_t s = 0, e = 0, ps = 0;
    uint64_t count = 0, bytes = 0;
    uint32_t g = 0;
    for (; g < world; g++) {
        const msj_carry &u = reports[g].used, &o = reports[g].out;
        std::memset(&exact_in[g], 0, sizeof(msj_carry));
        exact_in[g].in_string = s;
        exact_in[g].next_is_escaped = e;
        exact_in[g].prev_scalar = ps;
        // the stitched offsets: where shard g's first index and first byte sit in the
        // stream.  Index arrays themselves stay shard-local -- a launch counts from the carry it is given, and
        // the sharded entry points give it 0 -- so these are what a consumer adds (msj_shard_placement).
        exact_in[g].count = count;
        exact_in[g].bytes = bytes;
        count += o.count - u.count;
        bytes += o.bytes - u.bytes;
        if ((u.next_is_escaped & 1u) != e || (u.prev_scalar & 1u) != ps || o.internal_error) {
            mask |= 1ull << g;
            g++;
            break;
        }
        if ((u.in_string & 1u) != s) mask |= 1ull << g;
        s ^= (o.in_string ^ u.in_string) & 1u;
        e = o.next_is_escaped & 1u;
        ps = o.prev_scalar & 1u;
    }
    *rerun_mask = mask;
    // returns the number of ranks whose exact carry-in is known (world when the chain was replayed to the end)
    return (int32_t)g;
}

int32_t msj_shard_global_code(const msj_shard_report *reports, uint32_t world, uint32_t flags, uint64_t *total_count) {
    if (!reports || world == 0) return MSJ_ERR_BAD_ARGUMENT;
    uint64_t total = 0;
    uint32_t unescaped = 0, utf8 = 0, internal = 0, clipped = 0;
    for (uint32_t g = 0; g < world; g++) {
        total += reports[g].out.count - reports[g].used.count;
        unescaped |= reports[g].out.unescaped_error;
        utf8 |= reports[g].out.utf8_error;
        internal |= reports[g].out.internal_error;
        clipped |= reports[g].out.capacity_error;
    }
    if (total_count) *total_count = total;
    // finish(), json_structural_indexer.mojo:147-186: 15, then 14, then 13, then (strict) 11; CAPACITY where
    // the single-GPU path has it (stage1_kernel.hip finish_launch): a rank whose index buffer was too small
    // clipped its writes, wherever in the stream it sits
    if (internal) return MSJ_UNEXPECTED_ERROR;
    if (reports[world - 1].out.in_string) return MSJ_UNCLOSED_STRING;
    if (unescaped) return MSJ_UNESCAPED_CHARS;
    if (clipped) return MSJ_CAPACITY;
    if (total == 0) return MSJ_EMPTY;
    if ((flags & MSJ_FLAG_STRICT_UTF8) && utf8) return MSJ_UTF8_ERROR;
    return MSJ_SUCCESS;
}

int32_t msj_exchange_rccl(void *nccl_comm, uint32_t rank, uint32_t world, const char *librccl_path, msj_exchange *out) {
    if (!nccl_comm || !out || world == 0 || rank >= world) return MSJ_ERR_BAD_ARGUMENT;
    void *h = dlopen(librccl_path && *librccl_path ? librccl_path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return MSJ_ERR_NO_DEVICE;
    nccl_allgather_t fn = reinterpret_cast<nccl_allgather_t>(dlsym(h, "ncclAllGather"));
    if (!fn) return MSJ_ERR_NO_DEVICE;
    RcclExchange *x = new (std::nothrow) RcclExchange{nccl_comm, fn};
    if (!x) return MSJ_MEMALLOC;
    out->comm = x;  // owned by the msj_sharded created from this exchange (msj_sharded_destroy frees it)
    out->allgather = rccl_allgather;
    out->rank = rank;
    out->world = world;
    out->owns_comm = 1;
    return MSJ_SUCCESS;
}

void msj_exchange_release(msj_exchange *x) {
    if (!x || !x->owns_comm) return;
    delete static_cast<RcclExchange *>(x->comm);
    x->comm = nullptr;
    x->allgather = nullptr;
    x->owns_comm = 0;
}

int32_t msj_sharded_create(msj_ctx *ctx, const msj_exchange *xchg, const msj_sharded_ops *ops, msj_sharded **out) {
    if (!xchg || !out || !xchg->allgather || xchg->world == 0 || xchg->world > 64 || xchg->rank >= xchg->world)
        return MSJ_ERR_BAD_ARGUMENT;
    if (!ops && !ctx) return MSJ_ERR_BAD_ARGUMENT;
    *out = nullptr;
    msj_sharded *sh = new (std::nothrow) msj_sharded();
    if (!sh) return MSJ_MEMALLOC;
    if (ops) {
        sh->ops = *ops;
        sh->side = ops->side_stream;
    } else {
        sh->ops = msj_sharded_ops{};
        sh->ops.user = ctx;
        sh->ops.alloc = hip_alloc;
        sh->ops.free = hip_free;
        sh->ops.copy = hip_copy;
        sh->ops.sync = hip_sync;
        sh->ops.run_shard = hip_run_shard;
        sh->ops.event_create = hip_event_create;
        sh->ops.event_destroy = hip_event_destroy;
        sh->ops.event_record = hip_event_record;
        sh->ops.event_wait = hip_event_wait;
        sh->ops.stream_wait = hip_stream_wait;
        sh->ops.event_query = hip_event_query;
        sh->ops.event_elapsed_ns = hip_event_elapsed_ns;
        sh->hip_default = true;
        // the exchange's own stream, at the highest priority the device has: the all-gather and the read-back of
        // submission k become ready at the same moment as the kernel of submission k + 1 (both wait for kernel k)
        // and must not queue behind it.  Without it (creation failed) they share the submission's stream.
        int least = 0, greatest = 0;
        hipStream_t side = nullptr;
        if (hipSetDevice(msj_ctx_device(ctx)) == hipSuccess &&
            hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
            hipStreamCreateWithPriority(&side, hipStreamNonBlocking, greatest) == hipSuccess) {
            sh->side = side;
            sh->owns_side = true;
        } else {
            (void)hipGetLastError();
        }
    }
    const msj_sharded_ops &o = sh->ops;
    sh->events = o.event_create && o.event_destroy && o.event_record && o.event_wait && o.stream_wait;
    if (!sh->events) sh->side = nullptr;  // nothing could order a second stream behind the kernel
    sh->x = *xchg;
    if (xchg->owns_comm) sh->owned_rccl = static_cast<RcclExchange *>(xchg->comm);
    const uint64_t wb = (uint64_t)xchg->world * sizeof(msj_shard_report);
    for (Slot &sl : sh->slots) {
        void *p = nullptr;
        bool ok = o.alloc(o.user, sizeof(msj_shard_report), 0, &p) == MSJ_SUCCESS;
        sl.d_mine = static_cast<msj_shard_report *>(p);
        ok = ok && o.alloc(o.user, wb, 0, &p) == MSJ_SUCCESS;
        if (ok) sl.d_gathered = static_cast<msj_shard_report *>(p);
        ok = ok && o.alloc(o.user, wb, 1, &p) == MSJ_SUCCESS;
        if (ok) sl.h_gathered = static_cast<msj_shard_report *>(p);
        ok = ok && o.alloc(o.user, sizeof(msj_carry), 1, &p) == MSJ_SUCCESS;
        if (ok) sl.h_spec = static_cast<msj_carry *>(p);
        if (ok && sh->events)
            ok = o.event_create(o.user, &sl.ev_start) == MSJ_SUCCESS && o.event_create(o.user, &sl.ev_kernel) == MSJ_SUCCESS &&
                 o.event_create(o.user, &sl.ev_stitch) == MSJ_SUCCESS;
        if (!ok) {
            msj_sharded_destroy(sh);
            return MSJ_MEMALLOC;
        }
    }
    *out = sh;
    return MSJ_SUCCESS;
}

// best effort: nothing of this slot is in flight any more (error paths and destruction)
static void drain_slot(msj_sharded *sh, Slot &sl) {
    const msj_sharded_ops &o = sh->ops;
    (void)o.sync(o.user, sl.stream);
    if (sh->side) (void)o.sync(o.user, sh->side);
}

void msj_sharded_destroy(msj_sharded *sh) {
    if (!sh) return;
    const msj_sharded_ops &o = sh->ops;
    for (Slot &sl : sh->slots) {
        if (sl.busy) drain_slot(sh, sl);  // a submission nobody asked the result of
        o.free(o.user, sl.d_mine, 0);
        o.free(o.user, sl.d_gathered, 0);
        o.free(o.user, sl.h_gathered, 1);
        o.free(o.user, sl.h_spec, 1);
        if (sh->events) {
            if (sl.ev_start) o.event_destroy(o.user, sl.ev_start);
            if (sl.ev_kernel) o.event_destroy(o.user, sl.ev_kernel);
            if (sl.ev_stitch) o.event_destroy(o.user, sl.ev_stitch);
        }
    }
    if (sh->owns_side) (void)hipStreamDestroy(static_cast<hipStream_t>(sh->side));
    delete sh->owned_rccl;
    delete sh;
}

uint64_t msj_sharded_reruns(const msj_sharded *sh) { return sh ? sh->reruns : 0; }
uint64_t msj_sharded_rounds(const msj_sharded *sh) { return sh ? sh->rounds : 0; }
int32_t msj_sharded_get_stats(const msj_sharded *sh, msj_sharded_stats *out) {
    if (!sh || !out) return MSJ_ERR_BAD_ARGUMENT;
    std::memset(out, 0, sizeof *out);
    out->results = sh->results;
    out->rounds = sh->rounds;
    out->reruns = sh->reruns;
    out->stitch_device_ns = sh->stitch_device_ns;
    out->result_wait_ns = sh->result_wait_ns;
    out->kernel_device_ns = sh->kernel_device_ns;
    out->last_kernel_ns = sh->last_kernel_ns;
    out->last_stitch_ns = sh->last_stitch_ns;
    out->reruns_behind_queue = sh->reruns_behind_queue;
    return MSJ_SUCCESS;
}

int32_t msj_sharded_ticket_state(msj_sharded *sh, uint32_t ticket) {
    if (!sh || ticket >= kDepth || !sh->slots[ticket].busy) return MSJ_ERR_BAD_ARGUMENT;
    const msj_sharded_ops &o = sh->ops;
    if (!sh->events || !o.event_query) return MSJ_SHARDED_KERNEL_DONE | MSJ_SHARDED_REPORTS_IN;  // synchronous operations
    const Slot &sl = sh->slots[ticket];
    const int32_t k = o.event_query(o.user, sl.ev_kernel), r = o.event_query(o.user, sl.ev_stitch);
    if (k < 0 || r < 0) return MSJ_ERR_HIP;
    return (k ? MSJ_SHARDED_KERNEL_DONE : 0) | (r ? MSJ_SHARDED_REPORTS_IN : 0);
}

// kernel (with the carry in sl.h_spec) on the submission's stream -> all-gather of the reports -> pinned host copy,
// both on the exchange's stream behind the kernel's event; nothing waits here
static int32_t launch_round(msj_sharded *sh, Slot &sl, bool run_kernel, uint32_t extra_flags) {
    const msj_sharded_ops &o = sh->ops;
    int32_t rc;
    sl.kernel_ran = run_kernel;
    if (sh->events) {
        rc = o.event_record(o.user, sl.ev_start, sl.stream);
        if (rc != MSJ_SUCCESS) return rc;
    }
    if (run_kernel) {
        const int32_t last = sh->x.rank + 1 == sh->x.world;
        if (sh->hip_default) {
            // the three bits travel in the kernel's arguments and come back in carry_out.reserved[0] (the report's
            // `used` half is rebuilt from that echo on the host): a 64-byte upload in front of every launch cost the
            // stream ~9 us per step (profiles/r04/stitch_overlap.txt)
            const uint32_t bits = (sl.h_spec->in_string & 1u) | ((sl.h_spec->next_is_escaped & 1u) << 1) |
                                  ((sl.h_spec->prev_scalar & 1u) << 2);
            rc = msj_stage1_shard_device_cv(static_cast<msj_ctx *>(o.user), sl.d_shard, sl.shard_len, sl.d_idx, sl.idx_capacity,
                                            bits, &sl.d_mine->out, sl.d_segments, sl.max_segments, nullptr, sl.has_prefix,
                                            last, 0, sl.total_len, sl.stream, sl.flags | extra_flags);
        } else {
            rc = o.copy(o.user, &sl.d_mine->used, sl.h_spec, sizeof(msj_carry), 0, sl.stream);
            if (rc != MSJ_SUCCESS) return rc;
            rc = o.run_shard(o.user, sl.d_shard, sl.shard_len, sl.d_idx, sl.idx_capacity, &sl.d_mine->used, &sl.d_mine->out,
                             sl.d_segments, sl.max_segments, sl.has_prefix, last, sl.total_len, sl.stream,
                             sl.flags | extra_flags);
        }
        if (rc != MSJ_SUCCESS) return rc;
    }
    void *xs = sl.stream;  // the stream of the exchange
    if (sh->events) {
        rc = o.event_record(o.user, sl.ev_kernel, sl.stream);
        if (rc != MSJ_SUCCESS) return rc;
        if (sh->side) {
            xs = sh->side;
            rc = o.stream_wait(o.user, xs, sl.ev_kernel);
            if (rc != MSJ_SUCCESS) return rc;
        }
    }
    rc = sh->x.allgather(sh->x.comm, sl.d_mine, sl.d_gathered, sizeof(msj_shard_report), xs);
    if (rc != MSJ_SUCCESS) return rc;
    sh->rounds++;
    rc = o.copy(o.user, sl.h_gathered, sl.d_gathered, (uint64_t)sh->x.world * sizeof(msj_shard_report), 1, xs);
    if (rc == MSJ_SUCCESS && sh->events) rc = o.event_record(o.user, sl.ev_stitch, xs);
    return rc;
}

int32_t msj_stage1_sharded_submit(msj_sharded *sh, const uint8_t *d_shard, uint64_t shard_len, uint32_t *d_idx,
                                  uint64_t idx_capacity, uint64_t total_len, int32_t has_prefix,
                                  const msj_carry *speculation, msj_segment *d_segments, uint32_t max_segments,
                                  void *stream, uint32_t flags, uint32_t *ticket_out) {
    if (!sh || !d_shard || shard_len == 0 || !ticket_out) return MSJ_ERR_BAD_ARGUMENT;
    Slot &sl = sh->slots[sh->next];
    if (sl.busy) return MSJ_CAPACITY;  // more than kDepth submissions without a result
    const msj_sharded_ops &o = sh->ops;
    uint64_t looked = 0;
    if (speculation) {
        // only the three carry bits are the caller's to assume: counts, bytes and sticky flags start at zero
        std::memset(sl.h_spec, 0, sizeof(msj_carry));
        sl.h_spec->in_string = speculation->in_string & 1u;
        sl.h_spec->next_is_escaped = speculation->next_is_escaped & 1u;
        sl.h_spec->prev_scalar = speculation->prev_scalar & 1u;
    } else if (!has_prefix) {
        std::memset(sl.h_spec, 0, sizeof(msj_carry));
    } else {
        // from the shard's own bytes: the 64 stream bytes in front of it and its first 4 KiB -- or, while those
        // contradict neither hypothesis (strings made of digits, blanks or the letters of the literals), its first
        // 64 KiB, then its first MiB: a refuted guess costs the whole shard a second launch, a longer look one more
        // small blocking read.  A shard whose first look is what it was when its last result was verified takes
        // that result's carry instead of the longer looks.  A caller that resubmits an unchanged shard should pass
        // the carry its last result reported as used (`speculation`): then nothing is read here at all.
        static const uint64_t kHeads[] = {4096, 65536, 1048576};
        uint8_t *big = nullptr;
        int32_t rc = MSJ_SUCCESS;
        for (const uint64_t want : kHeads) {
            const uint64_t head = shard_len < want ? shard_len : want;
            uint8_t *ctx_bytes = sl.look;
            if (head > 4096) {
                if (!big) big = new (std::nothrow) uint8_t[64 + kHeads[2]];
                if (!big) break;  // keep the guess from the shorter head
                ctx_bytes = big;
            }
            rc = o.copy(o.user, ctx_bytes, d_shard - 64, 64 + head, 1, stream);
            if (rc == MSJ_SUCCESS) rc = o.sync(o.user, stream);
            int32_t decided = 0;
            if (rc == MSJ_SUCCESS) rc = msj_shard_speculate_ex(ctx_bytes, 64, ctx_bytes + 64, head, sl.h_spec, &decided);
            if (rc != MSJ_SUCCESS || decided || head == shard_len) break;
            if (head <= 4096) {
                looked = 64 + head;
                if (sh->proven.d_shard == d_shard && sh->proven.shard_len == shard_len && sh->proven.n == looked &&
                    std::memcmp(sh->proven.bytes, sl.look, looked) == 0) {
                    std::memset(sl.h_spec, 0, sizeof(msj_carry));
                    sl.h_spec->in_string = sh->proven.carry.in_string & 1u;
                    sl.h_spec->next_is_escaped = sh->proven.carry.next_is_escaped & 1u;
                    sl.h_spec->prev_scalar = sh->proven.carry.prev_scalar & 1u;
                    break;
                }
            }
        }
        delete[] big;
        if (rc != MSJ_SUCCESS) return rc;
    }
    sl.looked = looked;
    sl.d_shard = d_shard;
    sl.shard_len = shard_len;
    sl.d_idx = d_idx;
    sl.idx_capacity = idx_capacity;
    sl.total_len = total_len;
    sl.has_prefix = has_prefix;
    sl.d_segments = d_segments;
    sl.max_segments = max_segments;
    sl.stream = stream;
    sl.flags = flags;
    const int32_t rc = launch_round(sh, sl, true, 0);
    if (rc != MSJ_SUCCESS) {
        // part of the round may be queued (the carry's copy, the kernel, an exchange that failed behind them) and
        // reads or writes this slot's buffers: nothing of it may be pending when the slot is handed out again
        drain_slot(sh, sl);
        return rc;
    }
    sl.busy = true;
    *ticket_out = sh->next;
    sh->next = (sh->next + 1) % kDepth;
    return MSJ_SUCCESS;
}

int32_t msj_stage1_sharded_result(msj_sharded *sh, uint32_t ticket, int32_t *code_out, uint64_t *total_count_out,
                                  msj_carry *local_out, msj_carry *used_out, msj_shard_placement *placement_out) {
    if (!sh || ticket >= kDepth || !sh->slots[ticket].busy) return MSJ_ERR_BAD_ARGUMENT;
    Slot &sl = sh->slots[ticket];
    const msj_sharded_ops &o = sh->ops;
    const uint32_t world = sh->x.world, rank = sh->x.rank;
    msj_carry exact[64];
    // whatever goes wrong below, the ticket is free again afterwards (the submission is lost, not the slot) -- and
    // nothing of it is still queued: an error return leaves with the slot's streams drained, so that the next
    // submission that is given this slot cannot overwrite h_spec / h_gathered under a pending copy
    struct Release {
        msj_sharded *sh;
        Slot &sl;
        bool failed = true;
        ~Release() {
            if (failed) drain_slot(sh, sl);
            sl.busy = false;
        }
    } release{sh, sl};
    for (;;) {
        const auto t0 = std::chrono::steady_clock::now();
        // THIS submission's reports only: later submissions on the same stream keep running (with synchronous
        // operations -- no events -- the stream is drained as before)
        int32_t rc = sh->events ? o.event_wait(o.user, sl.ev_stitch) : o.sync(o.user, sl.stream);
        sh->result_wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        if (rc != MSJ_SUCCESS) return rc;
        if (sh->events && o.event_elapsed_ns) {
            uint64_t ns = 0;
            if (sl.kernel_ran && o.event_elapsed_ns(o.user, sl.ev_start, sl.ev_kernel, &ns) == MSJ_SUCCESS) {
                sh->kernel_device_ns += ns;
                sh->last_kernel_ns = ns;
            }
            if (o.event_elapsed_ns(o.user, sl.ev_kernel, sl.ev_stitch, &ns) == MSJ_SUCCESS) {
                sh->stitch_device_ns += ns;
                sh->last_stitch_ns = ns;
            }
        }
        // a launch that echoes the carry it started from (carry_out.reserved[0], every launch of this library does)
        // is taken at its word: its report's `used` half may never have been written
        for (uint32_t g = 0; g < world; g++) {
            const uint32_t echo = sl.h_gathered[g].out.reserved[0];
            if (echo & MSJ_CARRY_ECHO_VALID) {
                std::memset(&sl.h_gathered[g].used, 0, sizeof(msj_carry));
                sl.h_gathered[g].used.in_string = echo & 1u;
                sl.h_gathered[g].used.next_is_escaped = (echo >> 1) & 1u;
                sl.h_gathered[g].used.prev_scalar = (echo >> 2) & 1u;
            } else if (sh->hip_default) {
                // with the default operations nobody writes `used` on the device (the carry goes by value): a report
                // without the echo -- a launch that never reached finish(), a library in the group that predates the
                // echo -- has no carry to verify against.  An error, not a guess.
                return MSJ_ERR_HIP;
            }
        }
        uint64_t mask = 0;
        const int32_t known = msj_shard_verify(sl.h_gathered, world, exact, &mask);
        if (known < 0) return known;
        if (mask == 0 && (uint32_t)known == world) break;
        // every rank sees the same reports, so all agree on who indexes again; everybody else re-contributes
        // the report it has (no kernel), because the all-gather is collective
        const bool mine = (mask >> rank) & 1u;
        uint32_t extra = 0;
        if (mine) {
            // poisoned launch (an expired wait): through the two-pass kernels this time
            if (sl.h_gathered[rank].out.internal_error) extra = MSJ_FLAG_TWO_PASS;
            *sl.h_spec = exact[rank];
            sl.h_spec->count = 0;  // index arrays stay shard-local: every shard counts from 0 ...
            sl.h_spec->bytes = 0;  // ... its own bytes (the stitched offsets come back as msj_shard_placement)
            sh->reruns++;
            // the second launch goes to the END of the submission's stream: behind the kernels of the submissions
            // made since (one context = one workspace: launches of a context are stream-ordered).  Counted, so that
            // a caller sees it; a caller that passes the carry its last result proved never gets here twice.
            for (const Slot &other : sh->slots)
                if (&other != &sl && other.busy && other.stream == sl.stream) {
                    sh->reruns_behind_queue++;
                    break;
                }
        }
        rc = launch_round(sh, sl, mine, extra);
        if (rc != MSJ_SUCCESS) return rc;
    }
    release.failed = false;
    sh->results++;
    if (sl.looked) {  // an undecided first look: remember what the chain proved for these bytes
        sh->proven.d_shard = sl.d_shard;
        sh->proven.shard_len = sl.shard_len;
        sh->proven.n = sl.looked;
        sh->proven.carry = sl.h_gathered[rank].used;
        std::memcpy(sh->proven.bytes, sl.look, sl.looked);
    }
    uint64_t total = 0;
    const int32_t code = msj_shard_global_code(sl.h_gathered, world, sl.flags, &total);
    if (code_out) *code_out = code;
    if (total_count_out) *total_count_out = total;
    if (local_out) *local_out = sl.h_gathered[rank].out;
    if (used_out) *used_out = sl.h_gathered[rank].used;
    if (placement_out) {
        // every report stands: the exclusive sums of the replay are final (json_structural_indexer.mojo:34-37,
        // 160-165: BitIndexer.tail / n_structural_indexes of the one stream the shards are cut from)
        placement_out->index_begin = exact[rank].count;
        placement_out->byte_base = exact[rank].bytes;
        placement_out->count = sl.h_gathered[rank].out.count - sl.h_gathered[rank].used.count;
        placement_out->bytes = sl.h_gathered[rank].out.bytes - sl.h_gathered[rank].used.bytes;
    }
    return MSJ_SUCCESS;
}

}  // extern "C"
