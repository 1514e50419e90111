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
 * parse_number accepts (include/generic/number_parsing.mojo:22-80).  (The HIP kernel finds the closing
 * quote from the NEXT structural instead; the tests check that this is the same thing.)
 * Bodies over 1024 bytes: end still reported, backslash flag not (LONG); numbers over 1024: LONG, end 0. */
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
                if (j - (start + 1) > cap) f |= 128;
                else if (esc) f |= 2;
            }
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            f = 4;
            uint64_t j = start + 1;
            const uint64_t stop = (j + cap < len) ? j + cap : len;
            for (; j < stop; j++) {
                const uint8_t b = buf[j];
                if (b == '.' || b == 'e' || b == 'E') f |= 8;
                else if (!((b >= '0' && b <= '9') || b == '+' || b == '-')) break;
            }
            if (j == stop && stop < len) f |= 128;
            else e = (uint32_t)j;
        }
        end[i] = e;
        flags[i] = (uint8_t)f;
    }
}
