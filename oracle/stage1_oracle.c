/*
 * stage1_oracle.c -- CPU ORACLE for the stage-1 structural indexer.
 *
 * ===========================================================================
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load this library, and only as the checker / the reported CPU baseline.
 * The product path (mojo_simdjson_amd/, libmsj_stage1.so) never links, loads
 * or calls it.
 * ===========================================================================
 *
 * What this is: a plain-C restatement of the *algorithm* of the reference's
 * stage 1 (gabrieldemarmiesse/mojo-simdjson @ 2025-02-17, pure Mojo, cannot be
 * compiled here: no Mojo toolchain exists in this image).  Each function cites
 * the reference file:line it follows (paths relative to
 * /root/reference/src/mojo_simdjson/).
 *
 * Parity pin: the restatement is checked in tests/test_oracle.py against every
 * golden vector the reference's own tests hold for this path
 * (the 14 files of tests/jsons_for_test/valid/, copied as data into
 * tests/golden/jsons_for_test/), including the three trailer words the
 * reference test asserts (tests/test_stage_1.mojo:70-82), and against the
 * independent byte-serial formulation below (msj_oracle_stage1_serial) under
 * fuzzing.
 *
 * Three entry points:
 *   msj_oracle_stage1         block-for-block restatement (128-byte step, two
 *                             64-byte sub-blocks, delayed index write, 0x20
 *                             tail padding, same error order).  This is also
 *                             the "reference-faithful CPU" baseline bench.py
 *                             times (kind = "port").
 *   msj_oracle_stage1_serial  independent one-byte-at-a-time state machine
 *                             (SURVEY.md section 8c "serial spec").
 *   msj_oracle_utf8           strict UTF-8 validity (Unicode Table 3-7).  The
 *                             reference's checker is an empty stub
 *                             (generic/stage1/json_structural_indexer.mojo:16-30)
 *                             so this verdict is NOT part of reference parity;
 *                             tests pin it against CPython's strict decoder.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

/* errors.mojo:2-26 */
enum {
    MSJ_SUCCESS = 0,
    MSJ_CAPACITY = 1,
    MSJ_UTF8_ERROR = 11,
    MSJ_EMPTY = 13,
    MSJ_UNESCAPED_CHARS = 14,
    MSJ_UNCLOSED_STRING = 15,
    MSJ_UNEXPECTED_ERROR = 24
};

#define STEP_SIZE 128 /* include/generic/dom_parser_implementation.mojo:69 index[128] */

static int g_trace = 0; /* globals.mojo:3 TRACING_ENABLED (compile-time there, runtime here) */

void msj_oracle_set_trace(int on) { g_trace = on; }

/* debug.mojo:4-10 bin_display_reverse: LSB first, zeros blanked */
static void bin_display_reverse(uint64_t v, const char *info) {
    if (!g_trace) return;
    char line[65];
    for (int i = 0; i < 64; i++) line[i] = ((v >> i) & 1) ? '1' : ' ';
    line[64] = 0;
    printf("%s   %s\n", line, info);
}

/* memory.unsafe.pack_bits on a 64-lane bool vector: bit i <-> lane i */
/* stuff.mojo:6-9 eq[c] */
static uint64_t eq_mask(const uint8_t *in, uint8_t c) {
    uint64_t m = 0;
    for (int i = 0; i < 64; i++) m |= (uint64_t)(in[i] == c) << i;
    return m;
}

/* stuff.mojo:21-28 prefix_xor: inclusive prefix parity, literally the
 * 64-iteration shift + popcount loop (`bits << 64 - i - 1` is `<< (63 - i)`). */
static uint64_t prefix_xor(uint64_t bits) {
    uint64_t result = 0;
    for (int i = 0; i < 64; i++) {
        uint64_t b = (uint64_t)__builtin_popcountll(bits << (63 - i)) % 2;
        result |= b << i;
    }
    return result;
}

/* haswell.mojo:22-74 classify.  Two 16-entry tables repeated to 32 entries
 * (repeat_until[32], :6-19), looked up with SIMD._dynamic_shuffle(in_).
 * The lookup is restated as table[in & 31] (index modulo the table width).
 * Under x86 pshufb semantics (index = in & 15, result 0 when in >= 0x80) the
 * outcome is identical for all 256 byte values: every table value is < 0x80
 * and (in | 0x20) is never 0 -- tests/test_oracle.py checks both readings
 * exhaustively against the plain set definition below. */
static const uint8_t WS_TABLE16[16] = {' ', 100, 100, 100, 17, 100, 113, 2,
                                       100, '\t', '\n', 112, 100, '\r', 100, 100};
static const uint8_t OP_TABLE16[16] = {0, 0, 0, 0, 0, 0, 0, 0,
                                       0, 0, ':', '{', ',', '}', 0, 0};

int msj_oracle_classify_byte(uint8_t b, int pshufb_semantics) {
    /* returns bit0 = whitespace, bit1 = op for one byte (exposed for tests) */
    uint8_t ws_v, op_v;
    if (pshufb_semantics) {
        ws_v = (b & 0x80) ? 0 : WS_TABLE16[b & 15];
        op_v = (b & 0x80) ? 0 : OP_TABLE16[b & 15];
    } else {
        ws_v = WS_TABLE16[(b & 31) & 15]; /* 32-wide table = 16-wide repeated */
        op_v = OP_TABLE16[(b & 31) & 15];
    }
    int ws = (b == ws_v);                     /* haswell.mojo:65 */
    int op = ((uint8_t)(b | 0x20) == op_v);   /* haswell.mojo:67-69 */
    return ws | (op << 1);
}

typedef struct {
    uint64_t whitespace, op;
} JsonCharacterBlock; /* generic/json_character_block.mojo:4-23 */

static JsonCharacterBlock classify(const uint8_t *in) {
    JsonCharacterBlock c = {0, 0};
    for (int i = 0; i < 64; i++) {
        int r = msj_oracle_classify_byte(in[i], 0);
        c.whitespace |= (uint64_t)(r & 1) << i;
        c.op |= (uint64_t)((r >> 1) & 1) << i;
    }
    bin_display_reverse(c.whitespace, "whitespace");
    bin_display_reverse(c.op, "op");
    return c;
}
/* json_character_block.mojo:21-23 */
static uint64_t scalar_of(JsonCharacterBlock c) { return ~(c.op | c.whitespace); }

/* generic/stage1/json_escape_scanner.mojo:3,12-45 */
#define ODD_BITS 0xAAAAAAAAAAAAAAAAULL
typedef struct {
    uint64_t next_is_escaped;
} JsonEscapeScanner;

static uint64_t next_escape_and_terminal_code(uint64_t potential_escape) { /* :39-45 */
    uint64_t maybe_escaped = potential_escape << 1;
    uint64_t maybe_escaped_and_odd_bits = maybe_escaped | ODD_BITS;
    uint64_t even_series_codes_and_odd_bits = maybe_escaped_and_odd_bits - potential_escape;
    return even_series_codes_and_odd_bits ^ ODD_BITS;
}

/* :18-32.  SIMDJSON_SKIP_BACKSLASH_SHORT_CIRCUIT is True (globals.mojo:4) so
 * the no-backslash early-out is compiled out in the reference; same here. */
static uint64_t escape_scanner_next(JsonEscapeScanner *s, uint64_t backslash) {
    uint64_t escape_and_terminal_code =
        next_escape_and_terminal_code(backslash & ~s->next_is_escaped);
    uint64_t escaped = escape_and_terminal_code ^ (backslash | s->next_is_escaped);
    uint64_t escape = escape_and_terminal_code & backslash;
    s->next_is_escaped = escape >> 63;
    return escaped;
}

/* generic/stage1/json_string_scanner.mojo:9-74 */
typedef struct {
    uint64_t escaped, quote, in_string;
} JsonStringBlock;
typedef struct {
    JsonEscapeScanner escape_scanner;
    uint64_t prev_in_string;
} JsonStringScanner;

static JsonStringBlock string_scanner_next(JsonStringScanner *s, const uint8_t *in) { /* :55-69 */
    uint64_t backslash = eq_mask(in, '\\');
    uint64_t escaped = escape_scanner_next(&s->escape_scanner, backslash);
    uint64_t quote = eq_mask(in, '"') & ~escaped;
    uint64_t in_string = prefix_xor(quote) ^ s->prev_in_string;
    s->prev_in_string = (uint64_t)((int64_t)in_string >> 63); /* :60-62 sign extension */
    bin_display_reverse(escaped, "escaped");
    bin_display_reverse(quote, "quote");
    bin_display_reverse(in_string, "in_string");
    JsonStringBlock b = {escaped, quote, in_string};
    return b;
}

/* generic/stage1/json_scanner.mojo:7-79 */
typedef struct {
    JsonStringBlock string;
    JsonCharacterBlock characters;
    uint64_t follows_potential_nonquote_scalar;
} JsonBlock;
typedef struct {
    uint64_t prev_scalar;
    JsonStringScanner string_scanner;
} JsonScanner;

static uint64_t follows(uint64_t match, uint64_t *overflow) { /* :76-79 */
    uint64_t result = (match << 1) | *overflow;
    *overflow = match >> 63;
    return result;
}

static JsonBlock scanner_next(JsonScanner *s, const uint8_t *in) { /* :64-70 */
    JsonBlock b;
    b.string = string_scanner_next(&s->string_scanner, in);
    b.characters = classify(in);
    uint64_t nonquote_scalar = scalar_of(b.characters) & ~b.string.quote;
    b.follows_potential_nonquote_scalar = follows(nonquote_scalar, &s->prev_scalar);
    return b;
}

static uint64_t structural_start(const JsonBlock *b) { /* :24-26, :40-49 */
    uint64_t potential_scalar_start =
        scalar_of(b->characters) & ~b->follows_potential_nonquote_scalar;
    uint64_t potential_structural_start = b->characters.op | potential_scalar_start;
    bin_display_reverse(potential_structural_start, "potential_structural_start");
    uint64_t string_tail = b->string.in_string ^ b->string.quote; /* json_string_scanner.mojo:40-44 */
    bin_display_reverse(string_tail, "string_tail");
    return potential_structural_start & ~string_tail;
}

/* generic/stage1/json_structural_indexer.mojo:33-58 BitIndexer */
typedef struct {
    uint32_t *tail;
} BitIndexer;

static void bit_indexer_write(BitIndexer *ix, uint32_t idx, uint64_t bits) { /* :46-58 */
    if (bits == 0) return;
    int count = __builtin_popcountll(bits);
    for (int i = 0; i < count; i++) { /* :39-44 write_index */
        ix->tail[i] = idx + (uint32_t)__builtin_ctzll(bits);
        bits = bits & (bits - 1);
    }
    ix->tail += count;
}

/* generic/stage1/json_structural_indexer.mojo:67-186 JsonStructuralIndexer */
typedef struct {
    JsonScanner scanner;
    BitIndexer indexer;
    uint64_t prev_structurals;
    uint64_t unescaped_chars_error;
} JsonStructuralIndexer;

static void indexer_next(JsonStructuralIndexer *s, const uint8_t *in, const JsonBlock *jb,
                         int64_t index) { /* :129-145 */
    uint64_t unescaped = 0;
    for (int i = 0; i < 64; i++) unescaped |= (uint64_t)(in[i] <= 0x1F) << i; /* :135 */
    /* :137 checker.check_next_input is a no-op stub (:16-30) */
    bit_indexer_write(&s->indexer, (uint32_t)(index - 64), s->prev_structurals); /* :138 */
    s->prev_structurals = structural_start(jb);                                   /* :140 */
    bin_display_reverse(s->prev_structurals, "structural_start");
    s->unescaped_chars_error |= unescaped & jb->string.in_string; /* :143-145 */
}

static void indexer_step(JsonStructuralIndexer *s, const uint8_t *block, int64_t *reader_idx) {
    /* :110-127: two 64-byte sub-blocks per 128-byte step, then reader.advance() */
    for (int start = 0; start < STEP_SIZE; start += 64) {
        const uint8_t *in = block + start;
        if (g_trace) {
            printf("----------------------------------------------------------------\n");
            fwrite(in, 1, 64, stdout);
            printf("\n");
        }
        JsonBlock jb = scanner_next(&s->scanner, in);
        indexer_next(s, in, &jb, *reader_idx + start);
    }
    *reader_idx += STEP_SIZE; /* buf_block_reader.mojo:38-39 */
}

/*
 * DomParserImplementation.stage1(Span[UInt8]) + JsonStructuralIndexer.index[128]
 * (include/generic/dom_parser_implementation.mojo:65-69,85-89;
 *  generic/stage1/json_structural_indexer.mojo:81-108,147-186).
 *
 * idx/idx_capacity: the caller's structural_indexes buffer.  The reference
 * sizes it to exactly `len` slots (allocate(len), :85-89) and then writes the
 * three trailer words at n..n+2 (:167-173), which overruns by up to 3 slots
 * when every byte is structural; this oracle therefore requires
 * idx_capacity >= len + 3 and reports CAPACITY otherwise.
 *
 * On the two early error returns (UNCLOSED_STRING, UNESCAPED_CHARS) the
 * reference leaves n_structural_indexes and the trailer untouched
 * (:151-158); *n_out is likewise left untouched here.
 */
int32_t msj_oracle_stage1(const uint8_t *buf, uint64_t len, uint32_t *idx, uint64_t idx_capacity,
                          uint64_t *n_out) {
    if (len + 3 > idx_capacity) return MSJ_CAPACITY; /* :87-89 (see note above) */
    if (len == 0) return MSJ_EMPTY;                  /* :91-92 */

    /* buf_block_reader.mojo:10-17: len_minus_step = len - step (may be negative) */
    int64_t len_minus_step = (int64_t)len - STEP_SIZE;
    int64_t reader_idx = 0;

    JsonStructuralIndexer s;
    memset(&s, 0, sizeof s); /* :74-79 all carries zero */
    s.indexer.tail = idx;

    while (reader_idx < len_minus_step) { /* buf_block_reader.mojo:22-23, strict < */
        indexer_step(&s, buf + reader_idx, &reader_idx);
    }
    /* :103-107 last (always executed) block, padded with 0x20 */
    uint8_t block[STEP_SIZE];
    memset(block, 0x20, STEP_SIZE);
    int64_t number_of_chars = (int64_t)len - reader_idx; /* buf_block_reader.mojo:28-36 */
    if (number_of_chars == 0) return MSJ_UNEXPECTED_ERROR;
    memcpy(block, buf + reader_idx, (size_t)number_of_chars);
    indexer_step(&s, block, &reader_idx);

    /* finish(), :147-186 */
    bit_indexer_write(&s.indexer, (uint32_t)(reader_idx - 64), s.prev_structurals); /* :150 */
    if (s.scanner.string_scanner.prev_in_string) return MSJ_UNCLOSED_STRING; /* :151-155 */
    if (s.unescaped_chars_error) return MSJ_UNESCAPED_CHARS;                  /* :157-158 */
    uint64_t n = (uint64_t)(s.indexer.tail - idx);                            /* :160-165 */
    idx[n] = (uint32_t)len;                                                   /* :167-169 */
    idx[n + 1] = (uint32_t)len;                                               /* :170-172 */
    idx[n + 2] = 0;                                                           /* :173 */
    *n_out = n; /* parser.n_structural_indexes; next_structural_index = 0 (:174) */
    if (n == 0) return MSJ_EMPTY;                                             /* :176-177 */
    if (idx[n - 1] > (uint32_t)len) return MSJ_UNEXPECTED_ERROR;              /* :179-183 */
    return MSJ_SUCCESS; /* :185-186 utf8 checker stub always SUCCESS */
}

/*
 * Independent byte-serial formulation.  Deliberately
 * shares no code with the block version above: plain set membership, one byte
 * at a time, four booleans of state.
 */
int32_t msj_oracle_stage1_serial(const uint8_t *buf, uint64_t len, uint32_t *idx,
                                 uint64_t idx_capacity, uint64_t *n_out) {
    if (len + 3 > idx_capacity) return MSJ_CAPACITY;
    if (len == 0) return MSJ_EMPTY;
    int esc = 0, instr = 0, pnq = 0, bad = 0;
    uint64_t n = 0;
    for (uint64_t i = 0; i < len; i++) {
        uint8_t c = buf[i];
        int escaped = esc;
        if (escaped)
            esc = 0;
        else if (c == '\\')
            esc = 1;
        int quote = (c == '"') && !escaped;
        int before = instr;
        instr ^= quote;
        int ws = (c == 0x09 || c == 0x0A || c == 0x0D || c == 0x20);
        int op = (c == 0x0C || c == 0x1A || c == 0x2C || c == 0x3A || c == 0x5B || c == 0x5D ||
                  c == 0x7B || c == 0x7D);
        int scalar = !(ws || op);
        int pot = op || (scalar && !pnq);
        if (pot && !before) idx[n++] = (uint32_t)i;
        bad |= (c <= 0x1F) && instr;
        pnq = scalar && !quote;
    }
    if (instr) return MSJ_UNCLOSED_STRING;
    if (bad) return MSJ_UNESCAPED_CHARS;
    idx[n] = (uint32_t)len;
    idx[n + 1] = (uint32_t)len;
    idx[n + 2] = 0;
    *n_out = n;
    if (n == 0) return MSJ_EMPTY;
    return MSJ_SUCCESS;
}

/*
 * Strict UTF-8 validity, Unicode 15 Table 3-7 (well-formed byte sequences):
 * no overlongs (C0, C1, E0 80..9F, F0 80..8F), no surrogates (ED A0..BF),
 * nothing above U+10FFFF (F4 90.., F5..FF), no truncated sequence at EOF.
 * Returns 0 (valid) or MSJ_UTF8_ERROR (errors.mojo:13).
 */
int32_t msj_oracle_utf8(const uint8_t *buf, uint64_t len) {
    uint64_t i = 0;
    while (i < len) {
        uint8_t c = buf[i];
        if (c < 0x80) {
            i++;
            continue;
        }
        int need;
        uint8_t lo = 0x80, hi = 0xBF;
        if (c >= 0xC2 && c <= 0xDF)
            need = 1;
        else if (c == 0xE0) {
            need = 2;
            lo = 0xA0;
        } else if (c >= 0xE1 && c <= 0xEC)
            need = 2;
        else if (c == 0xED) {
            need = 2;
            hi = 0x9F;
        } else if (c >= 0xEE && c <= 0xEF)
            need = 2;
        else if (c == 0xF0) {
            need = 3;
            lo = 0x90;
        } else if (c >= 0xF1 && c <= 0xF3)
            need = 3;
        else if (c == 0xF4) {
            need = 3;
            hi = 0x8F;
        } else
            return MSJ_UTF8_ERROR;
        if (i + (uint64_t)need >= len) return MSJ_UTF8_ERROR; /* truncated at EOF */
        if (buf[i + 1] < lo || buf[i + 1] > hi) return MSJ_UTF8_ERROR;
        for (int k = 2; k <= need; k++)
            if (buf[i + k] < 0x80 || buf[i + k] > 0xBF) return MSJ_UTF8_ERROR;
        i += (uint64_t)need + 1;
    }
    return MSJ_SUCCESS;
}

/* Per-block trace of one input, the counterpart of building the reference with
 * -D SIMDJSON_TRACING_ENABLED (globals.mojo:3): prints every intermediate mask
 * LSB-first with zeros blanked (debug.mojo:4-10). */
int32_t msj_oracle_stage1_trace(const uint8_t *buf, uint64_t len, uint32_t *idx,
                                uint64_t idx_capacity, uint64_t *n_out) {
    g_trace = 1;
    int32_t rc = msj_oracle_stage1(buf, len, idx, idx_capacity, n_out);
    g_trace = 0;
    fflush(stdout);
    return rc;
}

/* Timing helper for bench.py's cpu_baseline leg: runs the block restatement
 * `reps` times over the same buffer, returns the last return code. */
int32_t msj_oracle_stage1_repeat(const uint8_t *buf, uint64_t len, uint32_t *idx,
                                 uint64_t idx_capacity, uint64_t *n_out, int reps) {
    int32_t rc = 0;
    for (int r = 0; r < reps; r++) rc = msj_oracle_stage1(buf, len, idx, idx_capacity, n_out);
    return rc;
}
