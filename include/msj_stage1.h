/*
 * msj_stage1.h -- C ABI of the MI355X-native stage-1 JSON structural indexer.
 *
 * This is the drop-in boundary for ONE path of gabrieldemarmiesse/mojo-simdjson:
 * the call
 *     JsonStructuralIndexer.index[128](buffer, self)
 * made by DomParserImplementation.stage1(Span[UInt8])
 * (reference: src/mojo_simdjson/include/generic/dom_parser_implementation.mojo:65-69,
 *  callee src/mojo_simdjson/generic/stage1/json_structural_indexer.mojo:81-108,147-186).
 * The reference has no FFI of its own (it is a plain Mojo static-method call);
 * the entry points below are what a Mojo `sys.ffi.DLHandle` binding for that
 * call site binds (INTEGRATION.md shows the shim).  Plain pointers and sizes
 * only; no torch / HIP types appear in any signature (`stream` is an opaque
 * hipStream_t passed as void*).
 *
 * Contract left behind by a successful call (what stage 2 reads,
 * generic/stage2/json_iterator.mojo:28-38,256-288, and what the reference's
 * own test asserts, tests/test_stage_1.mojo:43-82):
 *   idx[0..n)   strictly increasing uint32 byte offsets of structural starts
 *   idx[n]   = (uint32) len      \
 *   idx[n+1] = (uint32) len       } json_structural_indexer.mojo:167-173
 *   idx[n+2] = 0                 /
 *   *n_out   = n                   (parser.n_structural_indexes, :160-165)
 * Return value: the reference's integer error code (errors.mojo:2-36):
 *   0 SUCCESS, 1 CAPACITY, 13 EMPTY, 14 UNESCAPED_CHARS, 15 UNCLOSED_STRING,
 *   24 UNEXPECTED_ERROR; 11 UTF8_ERROR only when MSJ_FLAG_STRICT_UTF8 is set
 *   (the reference's UTF-8 checker is an empty stub that always succeeds,
 *   json_structural_indexer.mojo:16-30, so reference parity ignores UTF-8).
 * Error precedence follows finish() (:147-186): 15, then 14, then (trailer
 * written) 13, then 11.  On 14/15 the reference returns before writing n and
 * the trailer; the host-pointer entry points do the same (n_out untouched).
 *
 * Capacity: the reference allocates exactly `len` slots (allocate(len),
 * dom_parser_implementation.mojo:85-89) but writes up to n+3 <= len+3 words.
 * Callers of this ABI must provide idx_capacity >= n + 3; len + 3 is always
 * enough.  A smaller buffer is accepted: writes are clipped and CAPACITY (1)
 * is returned if n + 3 > idx_capacity.
 *
 * HIP backend only: every entry point fails with MSJ_ERR_NO_DEVICE (-2) when no
 * gfx950 device / HIP runtime is usable.  There is no CPU fallback.
 */
#ifndef MSJ_STAGE1_H
#define MSJ_STAGE1_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference error codes, src/mojo_simdjson/errors.mojo:2-26 */
#define MSJ_SUCCESS 0
#define MSJ_CAPACITY 1
#define MSJ_MEMALLOC 2
#define MSJ_UTF8_ERROR 11
#define MSJ_EMPTY 13
#define MSJ_UNESCAPED_CHARS 14
#define MSJ_UNCLOSED_STRING 15
#define MSJ_UNEXPECTED_ERROR 24
/* library-level failures (negative: never collide with reference codes) */
#define MSJ_ERR_BAD_ARGUMENT (-1)
#define MSJ_ERR_NO_DEVICE (-2)
#define MSJ_ERR_HIP (-3)

/* flags */
#define MSJ_FLAG_STRICT_UTF8 1u /* return 11 when the input is not valid UTF-8 */
#define MSJ_FLAG_NO_UTF8 2u     /* skip UTF-8 validation entirely (verdict = 0) */
/* Index through the two-pass kernels (summary, scan, emission: no workgroup ever waits for another one)
 * instead of the single-pass kernel.  Slower; what the library itself falls back to when a single-pass
 * launch reports internal_error (an inter-workgroup wait ran into its 2 s bound), so that a valid document
 * never comes back as UNEXPECTED_ERROR (24).  Same results bit for bit. */
#define MSJ_FLAG_TWO_PASS 0x100u
/* Test hook: the single-pass kernel's resolver idles ~2 ms before it starts (with msj_debug_set_wait_ticks
 * this forces the wait-expiry path). */
#define MSJ_FLAG_DEBUG_STALL 0x200u
/* The first n (0..15) bytes of the buffer read as blanks: a window of a document stream starts at
 * a document, its 16-byte aligned base a few bytes earlier (msj_documents_device, resume_offset). */
#define MSJ_FLAG_SKIP(n) (((uint32_t)(n) & 15u) << 24)

/* SIMDJSON_MAXSIZE_BYTES, src/mojo_simdjson/include/base.mojo:2: indices are
 * uint32, so one segment of input is limited to this many bytes. */
#define MSJ_MAX_SEGMENT_BYTES 0xFFFFFFFFull

/*
 * Carry state at a byte boundary of the input stream: everything the
 * reference's scanners carry from one 64-byte block to the next
 * (JsonEscapeScanner.next_is_escaped json_escape_scanner.mojo:13,
 *  JsonStringScanner.prev_in_string json_string_scanner.mojo:49,
 *  JsonScanner.prev_scalar json_scanner.mojo:57,
 *  JsonStructuralIndexer.unescaped_chars_error json_structural_indexer.mojo:72,
 *  BitIndexer.tail :34) plus the UTF-8 verdict so far.  Lives in device
 * memory; 64 bytes.
 */
typedef struct msj_carry {
    uint64_t count;           /* structurals emitted so far (BitIndexer.tail - base) */
    uint64_t bytes;           /* input bytes consumed so far */
    uint32_t in_string;       /* 1 = inside a string (prev_in_string != 0) */
    uint32_t next_is_escaped; /* 1 = next byte is escaped */
    uint32_t prev_scalar;     /* 1 = previous byte was a non-quote scalar */
    uint32_t unescaped_error; /* sticky: control char seen inside a string */
    uint32_t utf8_error;      /* sticky: invalid UTF-8 seen */
    uint32_t internal_error;  /* sticky: a wait inside the single-pass kernel ran into its bound */
    int32_t code;             /* reference return code, valid after a FINAL segment */
    uint32_t capacity_error;  /* sticky: the index buffer could not hold every index so far (+ the 3 trailer words
                                 on a FINAL segment); writes were clipped.  A FINAL segment also reports it as
                                 code = MSJ_CAPACITY, a non-final shard only here (msj_shard_global_code reads it) */
    uint32_t reserved[4];     /* [0], written by every shard call into its carry_out: bit 31 set, bits 0..2 the in_string /
                                 next_is_escaped / prev_scalar the call STARTED from (handed down the chain of a shard of
                                 several segments): what a rank of a sharded stream reports as the carry it used; [1..3] 0 */
} msj_carry;
#define MSJ_CARRY_ECHO_VALID 0x80000000u

/* One <= 4 GiB piece of a larger input: offsets in
 * idx[index_begin .. index_begin+count) are relative to byte_base. */
typedef struct msj_segment {
    uint64_t byte_base;
    uint64_t byte_len;
    uint64_t index_begin;
    uint64_t count;
} msj_segment;

typedef struct msj_ctx msj_ctx;

/* Library / device probes.  msj_device_count() returns the number of HIP
 * devices (0 when none or no runtime); never initialises a context. */
int32_t msj_device_count(void);
/* "mojo-simdjson_amd stage1 <version> (gfx950) src:<12 hex digits>": the digits are a hash of the stage-1 kernel's
 * sources; measurements kept beside the code (profiles/traffic.json) name the kernel they were taken with. */
const char *msj_version(void);

/* Context: owns the per-device workspace (tile descriptors, carry structs,
 * staging buffers).  Not re-entrant: one in-flight call per context, matching
 * the reference (one parser = one synchronous call). */
int32_t msj_ctx_create(int32_t device, msj_ctx **out);
void msj_ctx_destroy(msj_ctx *ctx);
int32_t msj_ctx_device(const msj_ctx *ctx); /* the HIP device the context was created on (-1: NULL) */

/*
 * msj_stage1 -- host-pointer form; replaces
 *   JsonStructuralIndexer.index[128](buffer, self)   (json_structural_indexer.mojo:81-108)
 * `buf`/`idx_out` are host memory; the library copies the input to the device,
 * runs the HIP kernels, and copies back idx[0..n+3).  Uses a process-wide
 * default context on device 0 (created on first use).
 * utf8_verdict_out (optional): 0 valid, 11 invalid -- reported separately from
 * the return code unless MSJ_FLAG_STRICT_UTF8.
 */
int32_t msj_stage1(const uint8_t *buf, uint64_t len, uint32_t *idx_out, uint64_t idx_capacity,
                   uint64_t *n_out, int32_t *utf8_verdict_out, uint32_t flags);

/* Same, on an explicit context. */
int32_t msj_stage1_ctx(msj_ctx *ctx, const uint8_t *buf, uint64_t len, uint32_t *idx_out,
                       uint64_t idx_capacity, uint64_t *n_out, int32_t *utf8_verdict_out,
                       uint32_t flags);

/*
 * Optional: pin a caller-owned host range once (hipHostRegister) and remember it.  msj_stage1 / msj_stage1_ctx
 * calls whose input and / or index array lie inside a registered range move that side by DMA straight from / into
 * the caller's memory instead of staging it through the library's pinned rings (two host copies less per byte).
 * For buffers the host reuses: the reference allocates structural_indexes once per parser, in allocate()
 * (include/generic/dom_parser_implementation.mojo:85-89) -- the shim registers it there and unregisters it where the
 * parser is destroyed.  Pinning costs ~50 us per MiB, once.  The range must stay allocated until
 * msj_host_unregister (msj_ctx_destroy unregisters what is left).  ctx NULL: the default context of msj_stage1.
 * Returns MSJ_SUCCESS, MSJ_ERR_BAD_ARGUMENT (null / empty / ptr not the start of a registered range), MSJ_ERR_HIP.
 */
int32_t msj_host_register(msj_ctx *ctx, void *ptr, uint64_t bytes);
/*
 * msj_host_placement -- where the host side of msj_stage1's pipeline lives, as one line of JSON in `out` (MSJ_CAPACITY if it
 * does not fit): the GPU's PCI address and NUMA node, how many CPUs of that node the process may use, whether the
 * pipeline exists yet (the first large call creates it), how many of its copy workers are bound to the GPU's node, the
 * node its pinned rings were placed on, the PCIe link's speed and width (sysfs; -1 / "" = the kernel does not say).
 * ctx NULL = the default context of msj_stage1.  A measurement aid: bench.py records it beside `end_to_end`.
 */
int32_t msj_host_placement(msj_ctx *ctx, char *out, uint64_t capacity);
/* the NUMA node the page under a host address lies on (-1: unknown); measurement aid like the above */
int32_t msj_debug_numa_node_of(const void *host_ptr);
int32_t msj_host_unregister(msj_ctx *ctx, void *ptr);

/*
 * msj_stage1_device -- device-resident form (what bench.py times).
 * d_buf: device pointer, 16-byte aligned, len < 2^32 bytes.
 * d_idx: device pointer (16-byte aligned) to idx_capacity uint32 slots.
 * d_result: device pointer to one msj_carry; after the stream drains it holds
 *   count (= n), code (reference return code), utf8_error, ...
 * Enqueues on `stream` (hipStream_t as void*, NULL = default stream) and
 * returns immediately; the return value only reports argument / launch errors.
 */
int32_t msj_stage1_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint32_t *d_idx,
                          uint64_t idx_capacity, msj_carry *d_result, void *stream,
                          uint32_t flags);

/* Blocking read-back of a device msj_carry (synchronises `stream`).  If it is the result of the context's last
 * msj_stage1_device / msj_stage1_shard_device call and reports internal_error (a wait inside the single-pass
 * kernel expired), that call is issued again through the two-pass kernels first (MSJ_FLAG_TWO_PASS), so the
 * caller gets the document's real result. */
int32_t msj_carry_fetch(msj_ctx *ctx, const msj_carry *d_carry, msj_carry *host_out, void *stream);

/* Test hook: bound of every inter-workgroup wait of the single-pass kernel, in 10 ns ticks (default 2 s). */
int32_t msj_debug_set_wait_ticks(msj_ctx *ctx, uint32_t ticks);
/* Number of times this context fell back to the two-pass kernels after an expired wait. */
uint64_t msj_fallback_count(const msj_ctx *ctx);

/*
 * msj_stage1_shard_device -- one byte-range shard of a larger stream
 * (multi-GPU sharding and > 4 GiB inputs).
 * The shard is cut into <= MSJ_MAX_SEGMENT_BYTES segments; indices are written
 * densely to d_idx (relative to each segment's byte_base, see msj_segment) and
 * the segment table to d_segments (device, max_segments entries; count is
 * filled in on the device).
 *   d_carry_in : device msj_carry with the exact state at the shard's first
 *                byte (zeroed for the start of the document).
 *   d_carry_out: device msj_carry receiving the state after the last byte (a shard may end
 *                anywhere, also inside a multi-byte character or right after a backslash; only its
 *                base must be 16-byte aligned).
 *   has_prefix : non-zero when d_buf[-64..0) is readable and holds the 64
 *                stream bytes preceding the shard (used for the UTF-8
 *                continuation check across the shard boundary).
 *   is_final   : non-zero for the last shard of the stream: writes the trailer
 *                (with trailer_len) and the reference return code into
 *                d_carry_out->code.
 *   no_emit    : non-zero = summary pass only (no index writes): used to get
 *                the shard's quote parity before the RCCL stitch.
 * *n_segments_out receives the number of segments used.
 */
int32_t msj_stage1_shard_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint32_t *d_idx,
                                uint64_t idx_capacity, const msj_carry *d_carry_in,
                                msj_carry *d_carry_out, msj_segment *d_segments,
                                uint32_t max_segments, uint32_t *n_segments_out,
                                int32_t has_prefix, int32_t is_final, int32_t no_emit,
                                uint64_t trailer_len, void *stream, uint32_t flags);

/* The same with the state at the shard's first byte given BY VALUE -- carry_bits: bit 0 in_string, bit 1
 * next_is_escaped, bit 2 prev_scalar; structurals, bytes and the sticky flags start at zero -- instead of a device
 * msj_carry: the start of a shard whose carries the host knows (or assumes: msj_shard_speculate).  Nothing has to be
 * copied to the device in front of the launch. */
int32_t msj_stage1_shard_device_cv(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint32_t *d_idx,
                                   uint64_t idx_capacity, uint32_t carry_bits, msj_carry *d_carry_out,
                                   msj_segment *d_segments, uint32_t max_segments, uint32_t *n_segments_out,
                                   int32_t has_prefix, int32_t is_final, int32_t no_emit, uint64_t trailer_len,
                                   void *stream, uint32_t flags);

/*
 * ---- N-GPU form: one contiguous byte-range shard of one stream per rank (SURVEY.md section 8b
 * `msj_stage1_sharded`, section 8e; one process per GPU) ----------------------------------------
 * The reference is single-threaded (no parallelism of any kind); what a Mojo host
 * calling DomParserImplementation.stage1 (include/generic/dom_parser_implementation.mojo:65-69) on a stream
 * that is spread over the GPUs of a node binds is the pair msj_stage1_sharded_submit / _result below.
 * Protocol (csrc/sharded.cpp): every rank assumes the carries at its shard's first byte from its own bytes
 * (msj_shard_speculate), runs the single-pass kernel once, and ONE all-gather of a 128-byte report per rank
 * lets every rank replay the chain (msj_shard_verify); only ranks whose assumption was refuted index again.
 * Index arrays stay shard-local (offsets relative to the shard, or to its segments: msj_segment); the
 * trailer is written by the last rank with `total_len`.
 */
typedef struct msj_shard_report {
    msj_carry used; /* the carry this rank's launch assumed at its first byte */
    msj_carry out;  /* the state after its last byte, its count and sticky errors */
} msj_shard_report;

/* Carries a shard may assume from its own bytes: halo = the <= 64 stream bytes in front of it (halo_len 0 at the
 * start of the stream), head = its first <= 4096 bytes.  next_is_escaped / prev_scalar are exact unless a run of
 * backslashes reaches halo[0]; in_string is a guess: the head is followed under both hypotheses until one meets a byte
 * it cannot hold (a control character inside a string; outside of strings anything but blanks, operators, number
 * characters and the letters of true / false / null), else the neighbours of the first unescaped quote decide. */
int32_t msj_shard_speculate(const uint8_t *halo, uint64_t halo_len, const uint8_t *head, uint64_t head_len,
                            msj_carry *out);
/* The same; *decided_out (optional) = 1 when one hypothesis was contradicted by the head (or there is no halo: the
 * start of the stream), 0 when the in_string guess rests on the quote-neighbour rule or on nothing at all -- a caller
 * with more bytes at hand then asks again with a longer head (msj_stage1_sharded_submit does: 4 KiB, 64 KiB, 1 MiB). */
int32_t msj_shard_speculate_ex(const uint8_t *halo, uint64_t halo_len, const uint8_t *head, uint64_t head_len,
                               msj_carry *out, int32_t *decided_out);
/* Replays the chain of all ranks' reports.  exact_in[g] (world entries) receives the exact carry at the start of
 * shard g for every g < return value -- in_string / next_is_escaped / prev_scalar, and the STITCHED OFFSETS:
 * exact_in[g].count = structurals of the stream in front of shard g (the position of its first index in the
 * stream-wide array that the reference's BitIndexer.tail / n_structural_indexes define,
 * json_structural_indexer.mojo:34-37,160-165), exact_in[g].bytes = stream bytes in front of it.  The counts of
 * ranks named in *rerun_mask are not final yet (they index again), so the offsets are final when the mask is 0.
 * *rerun_mask gets bit g set for every rank that has to index again with exact_in[g] (wrong in_string guess,
 * wrong escape carries, or a poisoned launch -- the chain cannot be followed past the latter two).  Returns world
 * and mask 0 when every report stands; < 0 on bad arguments. */
int32_t msj_shard_verify(const msj_shard_report *reports, uint32_t world, msj_carry *exact_in, uint64_t *rerun_mask);
/* The reference's return code for the whole stream (finish(), json_structural_indexer.mojo:147-186) and the
 * total structural count, from reports that msj_shard_verify accepted.  MSJ_CAPACITY when any rank's index
 * buffer was too small for its shard (out.capacity_error), in the place the single-GPU path gives it (after
 * 15 and 14, before 13 and 11). */
int32_t msj_shard_global_code(const msj_shard_report *reports, uint32_t world, uint32_t flags, uint64_t *total_count);

/* The one collective: all-gather of `bytes_per_rank` bytes per rank, device memory, enqueued on `stream`
 * (or completed before returning).  Returns MSJ_SUCCESS or a negative library error. */
typedef int32_t (*msj_allgather_fn)(void *comm, const void *d_send, void *d_recv, uint64_t bytes_per_rank, void *stream);
typedef struct msj_exchange {
    void *comm;                 /* passed to allgather as is */
    msj_allgather_fn allgather;
    uint32_t rank, world;       /* world <= 64 */
    uint32_t owns_comm;         /* set by msj_exchange_rccl: `comm` is freed by msj_sharded_destroy */
    uint32_t reserved;
} msj_exchange;
/* RCCL over xGMI: fills *out with an all-gather that calls ncclAllGather(..., ncclUint8, nccl_comm, stream).
 * nccl_comm is the caller's ncclComm_t (passed as void*; it stays the caller's: create it with
 * ncclCommInitRank, destroy it with ncclCommDestroy after msj_sharded_destroy).  The library does not link RCCL:
 * the symbol is taken from `librccl_path` (NULL: "librccl.so") at run time -- pass the RCCL the communicator
 * was created with. */
int32_t msj_exchange_rccl(void *nccl_comm, uint32_t rank, uint32_t world, const char *librccl_path, msj_exchange *out);

/* Device operations behind the protocol.  NULL in msj_sharded_create = HIP on the context's device; tests
 * substitute host memory and a CPU shard runner to run the protocol without a GPU. */
typedef struct msj_sharded_ops {
    void *user;
    int32_t (*alloc)(void *user, uint64_t bytes, int pinned_host, void **out);
    void (*free)(void *user, void *p, int pinned_host);
    int32_t (*copy)(void *user, void *dst, const void *src, uint64_t bytes, int to_host, void *stream);
    int32_t (*sync)(void *user, void *stream);
    int32_t (*run_shard)(void *user, const uint8_t *d_shard, uint64_t len, uint32_t *d_idx, uint64_t idx_capacity,
                         const msj_carry *d_carry_in, msj_carry *d_carry_out, msj_segment *d_segments,
                         uint32_t max_segments, int32_t has_prefix, int32_t is_final, uint64_t trailer_len,
                         void *stream, uint32_t flags);
    /* Optional -- the first five together or not at all (all NULL: operations that complete before they return, the
     * CPU tests' kind; msj_stage1_sharded_result then drains the submission's stream with `sync`).  With them a
     * result waits for the event behind ITS submission's read-back only, so that later submissions keep the GPU
     * busy meanwhile, and the exchange + read-back go to `side_stream` behind an event at the end of the kernel,
     * so that the next kernel starts behind the kernel, not behind the collective.  The default HIP operations have
     * all of them (hipEvent_t, a non-blocking stream of the highest priority). */
    int32_t (*event_create)(void *user, void **event_out);
    void (*event_destroy)(void *user, void *event);
    int32_t (*event_record)(void *user, void *event, void *stream);
    int32_t (*event_wait)(void *user, void *event);                /* the host blocks until the event has happened */
    int32_t (*stream_wait)(void *user, void *stream, void *event); /* what is enqueued on `stream` from now on waits */
    int32_t (*event_query)(void *user, void *event);               /* optional: 1 happened, 0 not yet, < 0 error */
    int32_t (*event_elapsed_ns)(void *user, void *from, void *to, uint64_t *ns_out); /* optional: statistics only */
    void *side_stream; /* custom operations: the stream of the exchange and the read-back (NULL: the submission's) */
} msj_sharded_ops;

typedef struct msj_sharded msj_sharded;
int32_t msj_sharded_create(msj_ctx *ctx, const msj_exchange *xchg, const msj_sharded_ops *ops, msj_sharded **out);
void msj_sharded_destroy(msj_sharded *sh);
uint64_t msj_sharded_reruns(const msj_sharded *sh); /* shard launches repeated by this rank (refuted guesses) */
uint64_t msj_sharded_rounds(const msj_sharded *sh); /* all-gathers so far */
/* Where the time of the stitch goes (cumulative since msj_sharded_create; device figures only with the default HIP
 * operations, 0 otherwise): results = msj_stage1_sharded_result calls completed; stitch_device_ns = HIP-event time
 * from the end of a round's kernel to the arrival of the gathered reports in pinned host memory (the all-gather
 * and the read-back, per round); result_wait_ns = host time spent blocked inside msj_stage1_sharded_result. */
typedef struct msj_sharded_stats {
    uint64_t results, rounds, reruns;
    uint64_t stitch_device_ns, result_wait_ns;
    uint64_t kernel_device_ns;    /* cumulative event time of this rank's shard launches alone (the carry's 64-byte upload
                                     + the kernel chain), no exchange in it: the rank's kernel-only time */
    uint64_t last_kernel_ns;      /* ... of the last completed result's launch */
    uint64_t last_stitch_ns;      /* end of that launch -> gathered reports in pinned memory.  The rank whose kernel ends
                                     last sees the exchange's bare latency here, every other rank that + its lead: the
                                     spread of this figure over the ranks of one step is the ranks' skew */
    uint64_t reruns_behind_queue; /* second launches (refuted guesses) that were enqueued behind the kernels of LATER
                                     submissions on the same stream (launches of one context are stream-ordered) */
    uint64_t reserved[3];
} msj_sharded_stats;
int32_t msj_sharded_get_stats(const msj_sharded *sh, msj_sharded_stats *out);
/* Where a submission is, without waiting: a set of MSJ_SHARDED_* bits (both set with operations that have no events:
 * everything completed inside submit), < 0 on error / a ticket that is not in flight. */
#define MSJ_SHARDED_KERNEL_DONE 1 /* the round's kernel (and what was in front of it on its stream) has finished */
#define MSJ_SHARDED_REPORTS_IN 2  /* the gathered reports have arrived: msj_stage1_sharded_result will not block */
int32_t msj_sharded_ticket_state(msj_sharded *sh, uint32_t ticket);
/* Gives back what msj_exchange_rccl allocated when the exchange is NOT handed to msj_sharded_create after all. */
void msj_exchange_release(msj_exchange *x);

/* Enqueue this rank's shard: the kernel on `stream`; the all-gather of the reports and the pinned read-back on the
 * library's own high-priority stream behind an event at the kernel's end (so the kernel of the NEXT submission on
 * `stream` runs beside them); returns at once with a ticket (up to 3 submissions may be in flight, all launches of
 * one msj_sharded on ONE stream: a context's launches are stream-ordered).  d_shard: 16-byte aligned device pointer;
 * with has_prefix the 64 bytes in front of it must be readable stream bytes.  speculation: the carry to assume (what
 * msj_shard_speculate gave for host copies of the bytes, or -- resubmitting an unchanged shard -- the carry its last
 * result reported as used: then nothing is read here).  NULL = derived here from the device bytes: one blocking
 * 4 KiB read of `stream` per call, plus a 64 KiB and a 1 MiB read while the bytes decide nothing (strings of digits
 * or literals) unless the shard's last verified result already settled it for the same bytes.
 * d_segments / max_segments as in msj_stage1_shard_device. */
int32_t msj_stage1_sharded_submit(msj_sharded *sh, const uint8_t *d_shard, uint64_t shard_len, uint32_t *d_idx,
                                  uint64_t idx_capacity, uint64_t total_len, int32_t has_prefix,
                                  const msj_carry *speculation, msj_segment *d_segments, uint32_t max_segments,
                                  void *stream, uint32_t flags, uint32_t *ticket_out);
/* Where a shard's results sit in the stream: the stitch's offsets (every rank folds the
 * ranks below it).  Local index k of this shard is index (index_begin + k) of the stream-wide array; a local
 * offset (plus its segment's byte_base, msj_segment) + byte_base is the offset in the stream. */
typedef struct msj_shard_placement {
    uint64_t index_begin; /* structurals of the stream in front of this shard: BitIndexer.tail - base at its first byte
                             (json_structural_indexer.mojo:34-37); the exclusive sum of the lower ranks' counts */
    uint64_t byte_base;   /* stream bytes in front of this shard */
    uint64_t count;       /* this shard's structurals */
    uint64_t bytes;       /* this shard's bytes */
} msj_shard_placement;
/* Wait for a submission -- for ITS gathered reports only: later submissions keep running on the GPU meanwhile
 * (msj_sharded_ticket_state tells) -- ; collective (every rank calls it for its matching ticket).  *code_out: the reference's
 * return code for the whole stream; *total_count_out: structurals of the whole stream (n_structural_indexes,
 * json_structural_indexer.mojo:160-165); *local_out: this shard's msj_carry (count = its own structurals);
 * *used_out: the exact carry at its first byte; *placement_out: the stitched offsets.  Any out pointer may be NULL.
 * A failure of the exchange or of a launch while ranks index again is returned as is and frees the ticket; the
 * collective is then broken for that submission on every rank (they fail or time out in their own exchange). */
int32_t msj_stage1_sharded_result(msj_sharded *sh, uint32_t ticket, int32_t *code_out, uint64_t *total_count_out,
                                  msj_carry *local_out, msj_carry *used_out, msj_shard_placement *placement_out);
/* Test hook: host-pointer inputs of at least this many bytes go through the chunked pinned pipeline of msj_stage1
 * (default 64 MiB; 0 restores it).  ctx NULL: the default context. */
int32_t msj_debug_set_pipeline_min_bytes(msj_ctx *ctx, uint64_t bytes);
/* Test hook: on != 0 makes the pipeline's set-up fail as it does on a host that cannot give it pinned memory; the
 * call (and every later one) then goes through the plain staging path.  on == 0 clears that state.  Returns 1 while
 * the context has given the pipeline up, 0 otherwise, < 0 on error.  ctx NULL: the default context. */
int32_t msj_debug_fail_pipeline_setup(msj_ctx *ctx, int32_t on);
/* Test hook: longest segment (bytes, multiple of 4096) one launch indexes; default MSJ_MAX_SEGMENT_BYTES rounded
 * down to the tile. */
int32_t msj_debug_set_segment_bytes(msj_ctx *ctx, uint64_t bytes);

/*
 * ---- token stream for stage 2 (DERIVED, see below) -------------
 * msj_tokens_device -- from the structural indices of one segment, two coalesced arrays:
 *   d_type[i]  = buf[idx[i]]: the byte JsonIterator.advance / peek / last_structural dereference
 *                one structural at a time (generic/stage2/json_iterator.mojo:256-288);
 *   d_depth[i] = nesting depth of token i, the running count walk_document keeps by hand
 *                (+1 at '{' '[', -1 at '}' ']', json_iterator.mojo:84-90,173-180): a bracket
 *                carries the depth of the container it sits in, so an opening bracket and its
 *                closing bracket have the same value and everything between them is deeper.
 *   d_match[i] (optional, may be NULL) = for a bracket, the index of the other end of its
 *                container -- what start_container / end_container keep on a stack
 *                (generic/stage2/tape_builder.mojo:235-272); 0xFFFFFFFF for every other token and
 *                for a bracket without a partner.
 * d_result: n, the final / minimum / maximum running depth (after each token): final != 0 is an
 * unclosed document, minimum < 0 a closing bracket without an opening one, maximum is what the
 * reference compares with max_depth (DEPTH_ERROR).
 * These are derived quantities: the reference has no such arrays and no fixture for them, so
 * the CPU statement the tests compare with is a definition, not a pin.
 * d_buf / d_idx as produced by msj_stage1_device (offsets < len < 2^32, n < 2^31); d_idx,
 * d_depth and (when given) d_match 16-byte aligned, d_type 8-byte aligned: anything else is
 * MSJ_ERR_BAD_ARGUMENT (the arrays leave as 16-byte stores).  Asynchronous on `stream`.
 */
typedef struct msj_tokens_result {
    uint64_t n;
    int32_t final_depth;
    int32_t min_depth;
    int32_t max_depth;
    uint32_t reserved; /* number of opening brackets */
} msj_tokens_result;

int32_t msj_tokens_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n,
                          uint8_t *d_type, int32_t *d_depth, uint32_t *d_match, msj_tokens_result *d_result,
                          void *stream);
/* The same for tokens that are NOT the start of their stream -- the next uint32 segment of a shard (msj_segment: a
 * 64 GiB stream is 8 GiB per GPU, two segments), the next window of a document stream: d_prev (device, or NULL =
 * msj_tokens_device) is the msj_tokens_result of the call that covered the tokens in front.  The running depth
 * walk_document keeps (generic/stage2/json_iterator.mojo:84-90,173-180) goes on from d_prev->final_depth -- that
 * int32 is the whole carry -- and d_result's final / min / max are those of the stream so far (n: this call's).
 * Read on the device, in stream order: calls chain without a host round trip.  d_match stays LOCAL to the call: a
 * container that closes in a later call keeps 0xFFFFFFFF at both ends (its opening bracket is found again from the
 * depths: the first later token at its depth). */
int32_t msj_tokens_chain_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n,
                                uint8_t *d_type, int32_t *d_depth, uint32_t *d_match, msj_tokens_result *d_result,
                                const msj_tokens_result *d_prev, void *stream);

/*
 * msj_token_spans_device -- per structural (DERIVED like the token
 * stream): for a string token the offset of its closing quote and whether the body holds a backslash
 * (the scan parse_string does first, generic/stage2/string_parsing.mojo:334-386); for a number token the
 * offset one past its last character and whether it is written as a float (number_parsing.mojo:22-80).
 *   d_end[i]:   string: offset of the closing quote (len if never closed); number: the end parse_number's scan finds
 *               (number_parsing.mojo:41-59: '-'? digits, then . e E makes it a float that ends at the first
 *               structural or blank byte; bytes past the buffer read as blanks); 0 if longer than 1024 characters;
 *               0 for every other token
 *   d_flags[i]: MSJ_SPAN_* bits
 */
#define MSJ_SPAN_STRING 1u   /* the token opens a string */
#define MSJ_SPAN_ESCAPED 2u  /* ... whose body holds at least one backslash */
#define MSJ_SPAN_NUMBER 4u   /* the token starts a number */
#define MSJ_SPAN_FLOAT 8u    /* ... written with '.', 'e' or 'E' */
#define MSJ_SPAN_OPEN 16u    /* string not closed before the end of the buffer (d_end = len) */
#define MSJ_SPAN_BAD 32u     /* number: the byte behind its digits is neither . e E nor structural / blank: the
                                reference's parse_number returns NUMBER_ERROR here (number_parsing.mojo:56-57) */
#define MSJ_SPAN_LONG 128u   /* number of more than 1024 characters: not scanned (d_end = 0).  Never set on a string: the
                                closing quote and MSJ_SPAN_ESCAPED are exact at any body length (round 5: bodies over 1024
                                bytes are scanned by a wave each behind the span kernel, over 1 MiB by the whole grid) */
int32_t msj_token_spans_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n,
                               uint32_t *d_end, uint8_t *d_flags, void *stream);

/*
 * Bracket partners as a COMPACT LIST (round 5): d_pairs[k] = {token index of the k-th opening bracket of the call, token
 * index of the bracket that closes its container, 0xFFFFFFFF if it is never closed inside the call}, k in the order of
 * the opening brackets; d_result->reserved is their number, the array needs room for as many (n at most).  Eight bytes per
 * CONTAINER instead of the four bytes per TOKEN of d_match -- brackets are 12 % of the minified workload's tokens -- for
 * the consumer that walks the tokens in order and takes one record at every opening bracket, the way the reference's
 * stage 2 pushes in start_container and pops in end_container (generic/stage2/tape_builder.mojo:235-272).  Same arguments
 * and results otherwise as msj_tokens_chain_device / msj_stage2_prep_chain_device without d_match; d_pairs 8-byte aligned.
 * The faster of the two partner forms (1 GiB minified: 1.00 ms per call against 1.17 with d_match): the call keeps the
 * brackets of its tokens as a compact list in the context's workspace (8 more bytes per token of capacity) and pairs them
 * there.
 */
typedef struct msj_bracket_pair {
    uint32_t open, close;
} msj_bracket_pair;
int32_t msj_tokens_pairs_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n, uint8_t *d_type,
                                int32_t *d_depth, msj_bracket_pair *d_pairs, msj_tokens_result *d_result,
                                const msj_tokens_result *d_prev, void *stream);
int32_t msj_stage2_prep_pairs_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n,
                                     uint8_t *d_type, int32_t *d_depth, msj_bracket_pair *d_pairs, uint32_t *d_end, uint8_t *d_flags,
                                     msj_tokens_result *d_result, const msj_tokens_result *d_prev, void *stream);

/*
 * PROTOTYPE (round 5; SURVEY.md section 8 row f1 from ONE pass over the bytes; measured and decided in DESIGN.md section 5b):
 * msj_stage1_types_device -- msj_stage1_device that also writes d_types[k] = d_buf[d_idx[k]], the type byte stage 2's
 *   JsonIterator.advance dereferences (generic/stage2/json_iterator.mojo:256-262), beside every index from the same
 *   emission (one uint32 segment, single-pass kernel only: MSJ_CAPACITY otherwise; d_types 4-byte aligned, same
 *   capacity as d_idx);
 * msj_depth_from_types_device -- depth (and, d_match given, bracket partners) of every token from such type bytes:
 *   msj_tokens_chain_device without the pass over the buffer.
 */
int32_t msj_stage1_types_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint32_t *d_idx, uint64_t idx_capacity,
                                uint8_t *d_types, msj_carry *d_result, void *stream, uint32_t flags);
int32_t msj_depth_from_types_device(msj_ctx *ctx, const uint8_t *d_type, uint64_t n, int32_t *d_depth, uint32_t *d_match,
                                    msj_tokens_result *d_result, const msj_tokens_result *d_prev, void *stream);

/* Test hook (per context, like msj_debug_set_segment_bytes): stretches of more than lds_limit_bytes take the span
 * kernels' global-memory path, the fix-up list holds fix_capacity entries; 0xFFFFFFFF = the built-in value of either. */
int32_t msj_debug_set_span_limits(msj_ctx *ctx, uint32_t lds_limit_bytes, uint32_t fix_capacity);
/* Test hook (per context): which of their two kernels the token calls run -- 0 (default) by the density of the index
 * (the kernel organised by tiles of the buffer from one structural per 11 bytes on, the one organised by tokens below
 * that), 1 = by tokens, 2 = by tiles whatever the density.  Identical results; the tests run both. */
int32_t msj_debug_set_span_mode(msj_ctx *ctx, uint32_t mode);
/* Test hook: bytes of the buffer per workgroup of the kernel organised by tiles (which = 0) and of its halo (which = 1):
 * what the tests move their tokens across. */
uint32_t msj_debug_tile_group(int32_t which);

/*
 * msj_stage2_prep_device -- msj_tokens_device and msj_token_spans_device in one go (rows f1 + f2 + f4), with
 * identical results: the span kernel holds every token's first byte already, so it writes the type bytes and
 * the depth aggregates too and the buffer is read once instead of twice.  Same arguments, alignment and limits
 * as the two calls (d_match optional).  Asynchronous on `stream`.
 */
int32_t msj_stage2_prep_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n,
                               uint8_t *d_type, int32_t *d_depth, uint32_t *d_match, uint32_t *d_end, uint8_t *d_flags,
                               msj_tokens_result *d_result, void *stream);
/* ... continuing a stream (d_prev as in msj_tokens_chain_device). */
int32_t msj_stage2_prep_chain_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n,
                                     uint8_t *d_type, int32_t *d_depth, uint32_t *d_match, uint32_t *d_end, uint8_t *d_flags,
                                     msj_tokens_result *d_result, const msj_tokens_result *d_prev, void *stream);
/*
 * msj_stage2_prep_segments -- rows f1 + f2 + f4 for a whole SHARD of several uint32 segments (what
 * msj_stage1_shard_device leaves behind for more than MSJ_MAX_SEGMENT_BYTES: BASELINE config 5's 8 GiB per GPU), in
 * one call: one msj_stage2_prep_chain_device per segment, the depth handed from segment to segment on the device.
 *   segments: HOST copy of the shard's msj_segment table (n_segments entries; their counts size the launches)
 *   d_buf: the shard's first byte; d_idx and all output arrays: the shard's token arrays, segment s at
 *          [index_begin_s - index_begin_0, + count_s) of d_idx (dense, as stage 1 wrote them; a slice that does not
 *          start on the 16-byte grid is copied to an aligned buffer of the library's before the kernels read it).  The
 *          OUTPUT arrays are not dense: see offsets below; pass them with 8 elements of slack per segment.
 *   The table is checked as a whole before anything is launched: segments must follow each other without gaps
 *   (byte_base_s = byte_base_{s-1} + byte_len_{s-1}, index_begin likewise: MSJ_ERR_BAD_ARGUMENT), 0 < byte_len <=
 *   MSJ_MAX_SEGMENT_BYTES and count < 2^31 (MSJ_CAPACITY).
 *   d_results: n_segments msj_tokens_result (device); the last one describes the shard
 *   d_prev: the result in front of the shard, or NULL
 * Outputs of segment s start at element offsets[s] = the sum of the counts in front of it, each rounded up to a multiple of 8
 * elements (so that every slice keeps the single calls' alignment); offsets_out (host, n_segments entries, may be NULL) receives them.
 * d_end is relative to the SEGMENT's bytes (like the indices).
 * d_match (round 5) is valid over the WHOLE SHARD: for a bracket, the position IN THE SHARD'S OUTPUT ARRAYS (offsets[s] +
 * the token's index inside its segment: the element at which that token's type / depth / end / flags / match are
 * stored) of the other end of its container -- also when the container is opened in one segment and closed in a later
 * one (the stack of start_container / end_container, generic/stage2/tape_builder.mojo:235-272, has no such border):
 * every segment leaves the brackets it could not pair in a residual list of the context's (their depth is their place
 * in it), and a stitch behind the last segment pairs them.  0xFFFFFFFF: not a bracket, or no partner inside the shard.
 * Limits with d_match: at most 32 segments and fewer than 2^32 - 1 output elements (MSJ_CAPACITY otherwise); at a segment
 * border, partners are stitched for nesting up to 65 536 containers deep -- beyond that the brackets keep 0xFFFFFFFF and
 * bit 31 of d_results[n_segments - 1].reserved is set.  The single calls (msj_stage2_prep_chain_device,
 * msj_tokens_chain_device) write into arrays of the caller's for each call and keep their partners local to the call.
 */
int32_t msj_stage2_prep_segments(msj_ctx *ctx, const uint8_t *d_buf, const msj_segment *segments, uint32_t n_segments,
                                 const uint32_t *d_idx, uint8_t *d_type, int32_t *d_depth, uint32_t *d_match, uint32_t *d_end,
                                 uint8_t *d_flags, msj_tokens_result *d_results, const msj_tokens_result *d_prev,
                                 uint64_t *offsets_out, void *stream);

/*
 * ---- multi-document mode (DERIVED) -----------------------------
 * The reference left upstream simdjson's streaming modes out
 * (generic/stage1/json_structural_indexer.mojo:153,169; generic/stage2/tape_builder.mojo:25 "TODO: add
 * streaming").  msj_documents_device splits the token stream of one window of a stream of concatenated
 * documents (NDJSON, or no separator at all) into documents: a document starts at every token that sits
 * at depth 0 and is not a closing bracket.
 *   d_doc_first[k] = token index (into d_idx / d_type / d_depth) of the first token of document k,
 *                    ascending; at most `capacity` are stored, n_documents counts all of them
 *   d_result:  n_documents      documents that START in the window
 *              n_complete       ... of which complete: all, or all but the last.  The last one is
 *                               complete if it is a container closed before the window ends; a closed
 *                               string (d_carry->in_string tells); any other scalar when is_final, or
 *                               when the window ends in a blank -- a number or literal that touches
 *                               the end of a window may go on in the next one
 *              tokens_complete  tokens covered by the complete documents: n, or the index of the first
 *                               token of the cut document (what upstream's find_next_document_index
 *                               returns for the window)
 *              resume_offset    byte offset (relative to the window) of that token = where the next
 *                               window has to start; len when nothing is cut
 * d_buf / len: the window; is_final: MSJ_DOCS_* bits -- MSJ_DOCS_FINAL: the window is the end of the stream;
 * MSJ_DOCS_AFTER_TOKENS: d_type / d_depth are exactly what the LAST msj_tokens_device / msj_stage2_prep_device call
 * on this context wrote (same n, same stream order) and have not been changed since: the per-block counts that call
 * left in the context's workspace are used instead of a pass over the two arrays.  d_type / d_depth as written by
 * msj_tokens_device for the same d_idx; d_carry (optional, may be NULL):
 * the carry_out of the window's msj_stage1_shard_device call, read on the device.  A window of a
 * stream is indexed with msj_stage1_shard_device(..., is_final = 0): nothing is an error yet at its end.
 * Asynchronous on `stream`.
 */
#define MSJ_DOCS_FINAL 1
#define MSJ_DOCS_AFTER_TOKENS 2
typedef struct msj_documents_result {
    uint64_t n_documents;
    uint64_t n_complete;
    uint64_t tokens_complete;
    uint64_t resume_offset;
} msj_documents_result;

int32_t msj_documents_device(msj_ctx *ctx, const uint8_t *d_buf, uint64_t len, int32_t is_final, const uint32_t *d_idx,
                             uint64_t n, const uint8_t *d_type, const int32_t *d_depth, const msj_carry *d_carry,
                             uint32_t *d_doc_first, uint64_t capacity, msj_documents_result *d_result, void *stream);

/*
 * Device memory for hosts that have no HIP binding of their own (a Mojo DLHandle, plain C, the C++ mirrors
 * under include/): allocation on the context's device and blocking copies.  Plumbing, not part of the path.
 */
int32_t msj_device_alloc(msj_ctx *ctx, uint64_t bytes, void **d_out);
int32_t msj_device_free(msj_ctx *ctx, void *d_ptr);
int32_t msj_copy_to_device(msj_ctx *ctx, void *d_dst, const void *src, uint64_t bytes, void *stream);
int32_t msj_copy_to_host(msj_ctx *ctx, void *dst, const void *d_src, uint64_t bytes, void *stream);

/* Tile geometry (for roofline bookkeeping and tests). */
uint32_t msj_tile_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* MSJ_STAGE1_H */
