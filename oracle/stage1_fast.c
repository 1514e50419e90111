/*
 * stage1_fast.c -- "best-case CPU" figures for the stage-1 path.  TEST / MEASUREMENT
 * INFRASTRUCTURE ONLY, like everything under oracle/: never linked into or called by the
 * product (mojo_simdjson_amd/), only by tests/ and by bench.py's cpu_baseline leg.
 *
 * NOT the reference.  SURVEY.md section 8d asks for two CPU numbers beside the
 * reference-faithful port (stage1_oracle.c), clearly labelled as not the reference:
 *   (i)  the same algorithm with the hot spots the reference leaves on the table fixed:
 *        prefix_xor as one carry-less multiply (the reference's stuff.mojo:21-28 is a
 *        64-iteration popcount loop), eq / classify as AVX2 compares + pshufb nibble
 *        tables (the reference's haswell.mojo:22-74 degenerates to per-lane extracts),
 *        the unescaped-character mask as one unsigned compare;
 *   (ii) that code on all cores: the stream is cut into chunks, pass 1 computes each
 *        chunk's (quote parity, structural count for both incoming in-string states,
 *        error bits for both), a serial prefix over the chunks fixes every chunk's state
 *        and output offset, pass 2 writes the indices.  The escape and prev_scalar carries
 *        into a chunk are derived from the bytes in front of it.
 * Results are identical to stage1_oracle.c (tests/test_oracle_fast.py): same effective
 * character sets including the reference's quirks (0x0C and 0x1A are operators), same
 * error precedence, same trailer.
 */
#include <immintrin.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MSJ_SUCCESS 0
#define MSJ_CAPACITY 1
#define MSJ_EMPTY 13
#define MSJ_UNESCAPED_CHARS 14
#define MSJ_UNCLOSED_STRING 15
#define ODD_BITS 0xAAAAAAAAAAAAAAAAULL

typedef struct {
    uint64_t backslash, quote, op, ws, ctrl;
} Masks;

static inline uint64_t movemask64(__m256i lo, __m256i hi) {
    return (uint64_t)(uint32_t)_mm256_movemask_epi8(lo) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(hi) << 32);
}

static inline uint64_t eq64(__m256i lo, __m256i hi, char c) {
    const __m256i v = _mm256_set1_epi8(c);
    return movemask64(_mm256_cmpeq_epi8(lo, v), _mm256_cmpeq_epi8(hi, v));
}

static inline Masks classify64(const uint8_t *in) {
    const __m256i lo = _mm256_loadu_si256((const __m256i *)in);
    const __m256i hi = _mm256_loadu_si256((const __m256i *)(in + 32));
    /* upstream simdjson's nibble tables: whitespace {09,0A,0D,20}, operators {2C,3A,5B,5D,7B,7D};
     * the reference's 32-entry variant additionally makes 0x0C and 0x1A operators */
    const __m256i ws_tbl = _mm256_setr_epi8(' ', 100, 100, 100, 17, 100, 113, 2, 100, '\t', '\n', 112, 100, '\r', 100, 100,
                                            ' ', 100, 100, 100, 17, 100, 113, 2, 100, '\t', '\n', 112, 100, '\r', 100, 100);
    const __m256i op_tbl = _mm256_setr_epi8(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, ':', '{', ',', '}', 0, 0,
                                            0, 0, 0, 0, 0, 0, 0, 0, 0, 0, ':', '{', ',', '}', 0, 0);
    const __m256i x20 = _mm256_set1_epi8(0x20);
    Masks m;
    m.ws = movemask64(_mm256_cmpeq_epi8(lo, _mm256_shuffle_epi8(ws_tbl, lo)),
                      _mm256_cmpeq_epi8(hi, _mm256_shuffle_epi8(ws_tbl, hi)));
    m.op = movemask64(_mm256_cmpeq_epi8(_mm256_or_si256(lo, x20), _mm256_shuffle_epi8(op_tbl, lo)),
                      _mm256_cmpeq_epi8(_mm256_or_si256(hi, x20), _mm256_shuffle_epi8(op_tbl, hi)));
    m.op |= eq64(lo, hi, 0x0C) | eq64(lo, hi, 0x1A);
    m.backslash = eq64(lo, hi, '\\');
    m.quote = eq64(lo, hi, '"');
    const __m256i x1f = _mm256_set1_epi8(0x1F);
    m.ctrl = movemask64(_mm256_cmpeq_epi8(_mm256_max_epu8(lo, x1f), x1f), _mm256_cmpeq_epi8(_mm256_max_epu8(hi, x1f), x1f));
    return m;
}

static inline uint64_t prefix_xor(uint64_t x) {
    const __m128i r = _mm_clmulepi64_si128(_mm_set_epi64x(0, (long long)x), _mm_set1_epi8((char)0xFF), 0);
    return (uint64_t)_mm_cvtsi128_si64(r);
}

typedef struct {
    uint64_t next_is_escaped, prev_in_string, prev_scalar, unescaped_error;
} Carry;

/* one 64-byte block: structural_start for the block's actual state, carries updated */
static inline uint64_t block_structurals(const Masks *m, Carry *c) {
    const uint64_t pe = m->backslash & ~c->next_is_escaped;
    const uint64_t t = (((pe << 1) | ODD_BITS) - pe) ^ ODD_BITS;
    const uint64_t escaped = t ^ (m->backslash | c->next_is_escaped);
    c->next_is_escaped = (t & m->backslash) >> 63;
    const uint64_t quote = m->quote & ~escaped;
    const uint64_t in_string = prefix_xor(quote) ^ c->prev_in_string;
    c->prev_in_string = (uint64_t)((int64_t)in_string >> 63);
    const uint64_t scalar = ~(m->op | m->ws);
    const uint64_t nqs = scalar & ~quote;
    const uint64_t follows = (nqs << 1) | c->prev_scalar;
    c->prev_scalar = nqs >> 63;
    c->unescaped_error |= m->ctrl & in_string;
    return (m->op | (scalar & ~follows)) & ~(in_string ^ quote);
}

static inline uint32_t *write_indices(uint32_t *tail, uint32_t base, uint64_t bits) {
    while (bits) {
        *tail++ = base + (uint32_t)__builtin_ctzll(bits);
        bits &= bits - 1;
    }
    return tail;
}

/* blocks [first, first + nblocks) of the stream; the last block of the stream is padded with
 * spaces (json_structural_indexer.mojo:103-107).  tail == NULL: count only. */
static uint64_t run_blocks(const uint8_t *buf, uint64_t len, uint64_t first, uint64_t nblocks, Carry *c, uint32_t *tail) {
    uint64_t count = 0;
    for (uint64_t b = first; b < first + nblocks; b++) {
        const uint64_t off = b * 64;
        uint8_t pad[64];
        const uint8_t *in = buf + off;
        if (off + 64 > len) {
            memset(pad, 0x20, 64);
            memcpy(pad, buf + off, (size_t)(len - off));
            in = pad;
        }
        const Masks m = classify64(in);
        const uint64_t s = block_structurals(&m, c);
        if (tail)
            tail = write_indices(tail, (uint32_t)off, s);
        count += (uint64_t)__builtin_popcountll(s);
    }
    return count;
}

/* pass 1 of the chunked run: one sweep that counts for BOTH incoming in-string states.  Entering
 * inside a string complements in_string for the whole chunk: string_tail = in_string ^ quote
 * flips, so state 1 keeps exactly the potential starts state 0 drops (json_scanner.mojo:24-26). */
static void run_blocks_both(const uint8_t *buf, uint64_t len, uint64_t first, uint64_t nblocks, Carry *c,
                            uint64_t count[2], uint64_t err[2]) {
    count[0] = count[1] = 0;
    err[0] = err[1] = 0;
    for (uint64_t b = first; b < first + nblocks; b++) {
        const uint64_t off = b * 64;
        uint8_t pad[64];
        const uint8_t *in = buf + off;
        if (off + 64 > len) {
            memset(pad, 0x20, 64);
            memcpy(pad, buf + off, (size_t)(len - off));
            in = pad;
        }
        const Masks m = classify64(in);
        const uint64_t pe = m.backslash & ~c->next_is_escaped;
        const uint64_t t = (((pe << 1) | ODD_BITS) - pe) ^ ODD_BITS;
        const uint64_t escaped = t ^ (m.backslash | c->next_is_escaped);
        c->next_is_escaped = (t & m.backslash) >> 63;
        const uint64_t quote = m.quote & ~escaped;
        const uint64_t in_string0 = prefix_xor(quote) ^ c->prev_in_string; /* chunk entered outside a string */
        c->prev_in_string = (uint64_t)((int64_t)in_string0 >> 63);
        const uint64_t scalar = ~(m.op | m.ws);
        const uint64_t nqs = scalar & ~quote;
        const uint64_t follows = (nqs << 1) | c->prev_scalar;
        c->prev_scalar = nqs >> 63;
        const uint64_t potential = m.op | (scalar & ~follows);
        const uint64_t tail0 = in_string0 ^ quote;
        count[0] += (uint64_t)__builtin_popcountll(potential & ~tail0);
        count[1] += (uint64_t)__builtin_popcountll(potential & tail0);
        err[0] |= m.ctrl & in_string0;
        err[1] |= m.ctrl & ~in_string0;
    }
    err[0] = err[0] != 0;
    err[1] = err[1] != 0;
}

static int32_t finish(const Carry *c, uint64_t n, uint64_t len, uint32_t *idx, uint64_t *n_out) {
    if (c->prev_in_string) return MSJ_UNCLOSED_STRING;
    if (c->unescaped_error) return MSJ_UNESCAPED_CHARS;
    idx[n] = (uint32_t)len;
    idx[n + 1] = (uint32_t)len;
    idx[n + 2] = 0;
    *n_out = n;
    return n == 0 ? MSJ_EMPTY : MSJ_SUCCESS;
}

/* (i) one thread */
int32_t msj_fast_stage1(const uint8_t *buf, uint64_t len, uint32_t *idx, uint64_t idx_capacity, uint64_t *n_out) {
    if (len + 3 > idx_capacity) return MSJ_CAPACITY;
    if (len == 0) return MSJ_EMPTY;
    Carry c = {0, 0, 0, 0};
    const uint64_t n = run_blocks(buf, len, 0, (len + 63) / 64, &c, idx);
    return finish(&c, n, len, idx, n_out);
}

/* ---- (ii) all cores ---------------------------------------------------------------- */
typedef struct {
    const uint8_t *buf;
    uint64_t len, first, nblocks;
    /* pass 1 out: per incoming in-string state q */
    uint64_t count[2], parity, err[2];
    /* pass 2 in */
    uint64_t in_string;
    uint32_t *tail;
    int pass;
} Chunk;

/* next_is_escaped / prev_scalar at a block boundary, from the bytes in front of it */
static void boundary_carry(const uint8_t *buf, uint64_t pos, Carry *c) {
    c->next_is_escaped = 0;
    c->prev_scalar = 0;
    if (pos == 0) return;
    uint64_t run = 0;
    while (run < pos && buf[pos - 1 - run] == '\\') run++;
    c->next_is_escaped = run & 1;
    const uint8_t b = buf[pos - 1];
    if (run >= 1) {
        c->prev_scalar = 1; /* a backslash is a non-quote scalar */
        return;
    }
    const int nonscalar = b == 0x20 || b == 0x09 || b == 0x0A || b == 0x0D || b == 0x0C || b == 0x1A || b == 0x2C ||
                          b == 0x3A || b == 0x5B || b == 0x5D || b == 0x7B || b == 0x7D;
    if (nonscalar) return;
    if (b != '"') {
        c->prev_scalar = 1;
        return;
    }
    /* a quote: a real one (not a scalar for `follows`) unless escaped by an odd run before it */
    uint64_t r2 = 0;
    while (r2 + 1 < pos && buf[pos - 2 - r2] == '\\') r2++;
    c->prev_scalar = r2 & 1;
}

static void *chunk_main(void *arg) {
    Chunk *k = (Chunk *)arg;
    if (k->pass == 1) {
        Carry c;
        boundary_carry(k->buf, k->first * 64, &c);
        c.prev_in_string = 0;
        c.unescaped_error = 0;
        run_blocks_both(k->buf, k->len, k->first, k->nblocks, &c, k->count, k->err);
        k->parity = c.prev_in_string & 1;
    } else {
        Carry c;
        boundary_carry(k->buf, k->first * 64, &c);
        c.prev_in_string = k->in_string ? ~0ULL : 0ULL;
        c.unescaped_error = 0;
        run_blocks(k->buf, k->len, k->first, k->nblocks, &c, k->tail);
    }
    return NULL;
}

int32_t msj_fast_stage1_mt(const uint8_t *buf, uint64_t len, uint32_t *idx, uint64_t idx_capacity, uint64_t *n_out,
                           int32_t nthreads) {
    if (len + 3 > idx_capacity) return MSJ_CAPACITY;
    if (len == 0) return MSJ_EMPTY;
    const uint64_t nblocks = (len + 63) / 64;
    if (nthreads < 1) nthreads = 1;
    if ((uint64_t)nthreads > nblocks) nthreads = (int32_t)nblocks;
    Chunk *ck = (Chunk *)calloc((size_t)nthreads, sizeof(Chunk));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    if (!ck || !th) {
        free(ck);
        free(th);
        return 2; /* MEMALLOC */
    }
    const uint64_t per = (nblocks + (uint64_t)nthreads - 1) / (uint64_t)nthreads;
    int used = 0;
    for (int t = 0; t < nthreads; t++) {
        const uint64_t first = (uint64_t)t * per;
        if (first >= nblocks) break;
        ck[t].buf = buf;
        ck[t].len = len;
        ck[t].first = first;
        ck[t].nblocks = (first + per <= nblocks) ? per : nblocks - first;
        used++;
    }
    for (int pass = 1; pass <= 2; pass++) {
        if (pass == 2) { /* serial prefix over the chunks */
            uint64_t s = 0, total = 0;
            for (int t = 0; t < used; t++) {
                ck[t].in_string = s;
                ck[t].tail = idx + total;
                total += ck[t].count[s];
                s ^= ck[t].parity;
            }
        }
        for (int t = 0; t < used; t++) {
            ck[t].pass = pass;
            pthread_create(&th[t], NULL, chunk_main, &ck[t]);
        }
        for (int t = 0; t < used; t++) pthread_join(th[t], NULL);
    }
    Carry c = {0, 0, 0, 0};
    uint64_t s = 0, total = 0, err = 0;
    for (int t = 0; t < used; t++) {
        total += ck[t].count[s];
        err |= ck[t].err[s];
        s ^= ck[t].parity;
    }
    c.prev_in_string = s ? ~0ULL : 0ULL;
    c.unescaped_error = err;
    free(ck);
    free(th);
    return finish(&c, total, len, idx, n_out);
}
