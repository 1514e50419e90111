/*
 * tokens_oracle.c -- CPU definition of the token-stream pre-pass.
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/.
 *
 * PARITY UNPINNED: the reference has no array of type bytes or depths and no fixture for them.
 * What it has is the code that recomputes both one structural at a time:
 *   type[i]  = buf[structural_indexes[i]]  -- JsonIterator.peek / advance / last_structural
 *              (src/mojo_simdjson/generic/stage2/json_iterator.mojo:256-288);
 *   depth    -- walk_document's running counter, +1 when a container is entered and -1 at
 *              scope_end (json_iterator.mojo:84-90,173-180).
 * This file states the quantity the HIP kernels (csrc/tokens_kernel.hip) compute: the plain
 * bracket nesting depth of every token (the reference skips the counter for empty containers and
 * stops at the first grammar error; a pre-pass cannot know either, stage 2 still does those
 * checks).  A bracket has the depth of the container it sits in.
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct {
    uint64_t n;
    int32_t final_depth, min_depth, max_depth;
    uint32_t reserved;
} msj_tokens_result;

void msj_oracle_tokens(const uint8_t *buf, const uint32_t *idx, uint64_t n, uint8_t *type, int32_t *depth,
                       msj_tokens_result *res) {
    int32_t d = 0, mn = 0, mx = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t c = buf[idx[i]];
        type[i] = c;
        if (c == '{' || c == '[') {
            depth[i] = d;
            d += 1;
        } else if (c == '}' || c == ']') {
            d -= 1;
            depth[i] = d;
        } else {
            depth[i] = d;
        }
        if (i == 0 || d < mn) mn = d;
        if (i == 0 || d > mx) mx = d;
    }
    res->n = n;
    res->final_depth = d;
    res->min_depth = mn;
    res->max_depth = mx;
    res->reserved = 0;
}

/* match[i]: for a bracket, the index of the other end of its container -- the stack start_container /
 * end_container keep (generic/stage2/tape_builder.mojo:235-272); 0xFFFFFFFF for every other token and
 * for a bracket without a partner (a closing bracket on an empty stack, an opening one never closed).
 * Like the depth, bracket KINDS are not compared here (stage 2's state machine does that). */
int msj_oracle_match(const uint8_t *type, uint64_t n, uint32_t *match) {
    uint32_t *stack = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    if (!stack) return -1;
    uint64_t top = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t c = type[i];
        match[i] = 0xFFFFFFFFu;
        if (c == '{' || c == '[') {
            stack[top++] = (uint32_t)i;
        } else if ((c == '}' || c == ']') && top > 0) {
            const uint32_t o = stack[--top];
            match[i] = o;
            match[o] = (uint32_t)i;
        }
    }
    free(stack);
    return 0;
}

/* Token spans (rows f2 / f4), the plain statement: scan forward from the opening quote like
 * parse_string does (generic/stage2/string_parsing.mojo:334-386) -- the first quote that is not behind a
 * backslash closes the string, any backslash on the way sets the flag; walk over the characters
 * parse_number's scan takes (include/generic/number_parsing.mojo:41-59, restated in the number branch
 * below and, without the cap, in msj_ref_parse_number_scan).  (The HIP kernel finds the closing
 * quote from the NEXT structural instead; the tests check that this is the same thing.)
 * Strings: exact at any length (round 5).  Numbers over 1024 characters: LONG, end 0. */
/* structural_or_whitespace, internal/jsoncharutils_tables.mojo:5-16: 09 0A 0D 20 , : [ ] { } */
static int msj_ref_structural_or_whitespace(uint8_t c) {
    return c == 0x09 || c == 0x0A || c == 0x0D || c == 0x20 || c == ',' || c == ':' || c == '[' || c == ']' || c == '{' || c == '}';
}

/*
 * parse_number (include/generic/number_parsing.mojo:22-80) up to the point where it hands the text to the
 * standard library: returns 9 (NUMBER_ERROR, :56-57) or 0, *end = offset one past the number's text, *is_float.
 * Literal: no cap, reads buf[len] as the caller's padding byte `pad` (the reference reads the String's NUL).
 */
int msj_ref_parse_number_scan(const uint8_t *buf, uint64_t len, uint64_t start, uint8_t pad, uint64_t *end, int *is_float) {
#define AT(k) ((k) < len ? buf[(k)] : pad)
    const int has_minus_sign = AT(start) == '-';           /* :42 */
    uint64_t p = start + (uint64_t)has_minus_sign;          /* :43 */
    while (AT(p) >= '0' && AT(p) <= '9') p++;               /* :45-46 */
    if (AT(p) == '.' || AT(p) == 'e' || AT(p) == 'E') {     /* :49 */
        *is_float = 1;
        while (!msj_ref_structural_or_whitespace(AT(p)) && p < len + 1) p++;  /* :53-54 (bounded by the padding byte) */
    } else if (!msj_ref_structural_or_whitespace(AT(p))) {  /* :56 */
        *end = p;
        *is_float = 0;
        return 9;                                           /* :57 NUMBER_ERROR */
    } else {
        *is_float = 0;                                      /* :59 */
    }
    *end = p;
    return 0;
#undef AT
}

/*
 * parse_string's terminator search (generic/stage2/string_parsing.mojo:334-386) over BackslashAndQuote windows of
 * 8 bytes (include/haswell/stringparsing_defs.mojo:27-48: the load is 8 wide): `start` = offset of the first body
 * byte.  Returns the offset of the closing quote, or -1 where the reference returns a null pointer (a bogus escape,
 * :372-375, or a \u escape whose four hex digits are not hex -- handle_unicode_codepoint's first check; surrogate
 * pairing is not restated: it consumes 6 or 12 bytes of valid hex and cannot hide a quote).  *escaped: a backslash
 * was met on the way.  Bytes past the buffer read as `pad`.
 * bytes_processed: how far a window without quote or backslash advances.  The reference advances by
 * BackslashAndQuote.BYTES_PROCESSED = 32 (stringparsing_defs.mojo:10) although copy_and_find loads and examines 8
 * bytes (:42 `src.load[width=8]()`): with 32 it skips 24 unexamined bytes, so on a string whose first 8 body bytes
 * hold neither a quote nor a backslash it can run past the closing quote (upstream simdjson loads 32 and advances
 * 32).  Literal restatement: pass 32.  What the port evidently means -- and what stage 1's own in-string logic and the
 * HIP kernels compute -- is the advance by the width examined: pass 8.  tests/test_tokens.py checks the kernels against
 * 8 and records where the literal 32 differs.
 */
int64_t msj_ref_parse_string_end(const uint8_t *buf, uint64_t len, uint64_t start, uint8_t pad, int bytes_processed, int *escaped) {
#define AT(k) ((k) < len ? buf[(k)] : pad)
    static const char ok_escapes[] = "\"\\/bfnrt";         /* escape_map's non-zero entries */
    uint64_t src = start;
    *escaped = 0;
    for (;;) {
        if (src > len) return -1;                           /* ran off an unterminated string (the reference would read on) */
        uint32_t bs_bits = 0, quote_bits = 0;               /* copy_and_find: 8 bytes */
        for (int k = 0; k < 8; k++) {
            if (AT(src + k) == '\\') bs_bits |= 1u << k;
            if (AT(src + k) == '"') quote_bits |= 1u << k;
        }
        if (((bs_bits - 1u) & quote_bits) != 0) return (int64_t)(src + (uint64_t)__builtin_ctz(quote_bits));  /* has_quote_first :358-360 */
        if (((quote_bits - 1u) & bs_bits) != 0) {            /* has_backslash :361 */
            const uint64_t bs_dist = (uint64_t)__builtin_ctz(bs_bits);
            const uint8_t escape_char = AT(src + bs_dist + 1);
            *escaped = 1;
            if (escape_char == 'u') {                        /* :366-375 */
                for (int k = 2; k < 6; k++) {
                    const uint8_t h = AT(src + bs_dist + k);
                    if (!((h >= '0' && h <= '9') || (h >= 'a' && h <= 'f') || (h >= 'A' && h <= 'F'))) return -1;
                }
                src += bs_dist + 6;
            } else {
                int ok = 0;
                for (const char *e = ok_escapes; *e; e++) ok |= (uint8_t)*e == escape_char;
                if (!ok) return -1;                          /* bogus escape value :379-382 */
                src += bs_dist + 2;                          /* :384 */
            }
        } else {
            src += (uint64_t)bytes_processed;                /* neither :385-386 */
        }
    }
#undef AT
}

void msj_oracle_token_spans(const uint8_t *buf, uint64_t len, const uint32_t *idx, uint64_t n, uint32_t *end, uint8_t *flags) {
    const uint64_t cap = 1024;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t start = idx[i];
        const uint8_t c = buf[start];
        uint32_t e = 0, f = 0;
        if (c == '"') {
            f = 1;
            uint64_t j = start + 1;
            int closed = 0, esc = 0;
            while (j < len) {
                if (buf[j] == '\\') { esc = 1; j += 2; continue; }
                if (buf[j] == '"') { closed = 1; break; }
                j++;
            }
            if (!closed) { e = (uint32_t)len; f |= 16; }
            else {
                e = (uint32_t)j;
                if (esc) f |= 2;  /* at ANY body length (round 5: the cap of 1024 bytes now binds numbers only) */
            }
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            /* parse_number's scan restated (include/generic/number_parsing.mojo:41-59): has_minus_sign,
             * `while is_digit(p[0])`, then  p[0] in . e E -> is_float, `while is_not_structural_or_whitespace(p[0])`;
             * elif is_not_structural_or_whitespace(p[0]) -> NUMBER_ERROR (flag 32); else an integer that ends here.
             * Bytes past the buffer read as blanks (the kernels' convention). */
            f = 4;
            const uint64_t stop = (start + 1 + cap < len) ? start + 1 + cap : len;
            uint64_t j = start + (c == '-' ? 1 : 0);
            while (j < stop && buf[j] >= '0' && buf[j] <= '9') j++;
            if (j < stop || j == len) {
                const uint8_t ch = j < len ? buf[j] : ' ';
                if (ch == '.' || ch == 'e' || ch == 'E') {
                    f |= 8;
                    while (j < stop && !msj_ref_structural_or_whitespace(buf[j])) j++;
                } else if (!msj_ref_structural_or_whitespace(ch)) {
                    f |= 32;
                }
            }
            if (j == stop && stop < len) { f = (f & ~32u) | 128; e = 0; }
            else e = (uint32_t)j;
        }
        end[i] = e;
        flags[i] = (uint8_t)f;
    }
}

/*
 * ---- multi-document mode -------------------------------------------
 * PARITY UNPINNED: the reference has no streaming mode (json_structural_indexer.mojo:153,169 and
 * tape_builder.mojo:25 "TODO: add streaming" mark where upstream simdjson's was left out).
 *
 * msj_oracle_documents is the DEFINITION of what csrc/documents_kernel.hip computes, as one forward
 * walk: a document starts at every token at depth 0 that is not a closing bracket; the last one is
 * complete if it is a container whose closing bracket -- a closing bracket at depth 0 -- comes after
 * it, a closed string, or another scalar that cannot go on in the next window (the window is the end
 * of the stream, or it ends in a blank).
 *
 * msj_oracle_find_next_document_index restates the published algorithm of upstream simdjson (the
 * C++ library the reference ports; src/generic/stage1/find_next_document_index.h, with the
 * streaming_partial step of json_structural_indexer::finish in front of it): a backward walk over
 * the structurals that counts brackets until it meets the start of the last document.  The CPU
 * tests check that both agree on well-formed streams cut at arbitrary points (with is_final = 1:
 * upstream counts a number that touches the end of a window as complete); on malformed input
 * they differ by design (upstream stops at the first plausible boundary, the definition above
 * follows the depth).
 */
typedef struct {
    uint64_t n_documents, n_complete, tokens_complete, resume_offset;
} msj_documents_result;

void msj_oracle_documents(const uint8_t *buf, uint64_t len, int is_final, const uint32_t *idx, uint64_t n,
                          const uint8_t *type, const int32_t *depth, int open_string, uint32_t *doc_first, uint64_t capacity, msj_documents_result *res) {
    uint64_t docs = 0, last_start = 0, last_close = 0;
    int have_close = 0;
    for (uint64_t i = 0; i < n; i++) {
        const int closing = type[i] == '}' || type[i] == ']';
        if (depth[i] != 0) continue;
        if (closing) {
            last_close = i;
            have_close = 1;
        } else {
            if (docs < capacity) doc_first[docs] = (uint32_t)i;
            docs++;
            last_start = i;
        }
    }
    res->n_documents = docs;
    if (docs == 0) {
        res->n_complete = 0;
        res->tokens_complete = n;
        res->resume_offset = len;
        return;
    }
    int complete;
    if (type[last_start] == '{' || type[last_start] == '[')
        complete = have_close && last_close > last_start;
    else if (last_start + 1 < n)
        complete = 1; /* something follows the scalar */
    else if (open_string)
        complete = 0; /* the window ends inside the string this token opens */
    else if (type[last_start] == '"' || is_final)
        complete = 1;
    else /* a number or a literal that touches the end of the window may go on in the next one */
        complete = buf[len - 1] == ' ' || buf[len - 1] == '\n' || buf[len - 1] == '\r' || buf[len - 1] == '\t';
    res->n_complete = docs - (complete ? 0 : 1);
    res->tokens_complete = complete ? n : last_start;
    res->resume_offset = complete ? len : idx[last_start];
}

/* returns the number of structurals that belong to complete documents (upstream: the new
 * n_structural_indexes of a streaming_partial window); *error = 1 where upstream returns an error
 * (nothing left after dropping the unclosed string's quote) */
uint64_t msj_oracle_find_next_document_index(const uint8_t *buf, const uint32_t *idx, uint64_t n, int open_string, int *error) {
    *error = 0;
    if (open_string) {  /* the quote that opens the unclosed string is not part of this window */
        if (n == 0) {
            *error = 1;
            return 0;
        }
        n--;
    }
    if (n == 0) {
        *error = 1;
        return 0;
    }
    int64_t arr_cnt = 0, obj_cnt = 0;
    for (uint64_t i = n - 1; i > 0; i--) {
        const uint8_t b = buf[idx[i]];
        if (b == ':' || b == ',') continue;
        if (b == '}') {
            obj_cnt--;
            continue;
        }
        if (b == ']') {
            arr_cnt--;
            continue;
        }
        if (b == '{') obj_cnt++;
        if (b == '[') arr_cnt++;
        const uint8_t a = buf[idx[i - 1]];
        if (a == '{' || a == '[' || a == ':' || a == ',') continue;
        /* token i starts a document */
        if (arr_cnt == 0 && obj_cnt == 0) return n; /* and that document is complete */
        return i;
    }
    const uint8_t f = buf[idx[0]];
    if (f == '}') obj_cnt--;
    if (f == ']') arr_cnt--;
    if (f == '{') obj_cnt++;
    if (f == '[') arr_cnt++;
    if (arr_cnt == 0 && obj_cnt == 0) return n;
    return 0;
}
