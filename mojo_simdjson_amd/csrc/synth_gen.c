/*
 * synth_gen.c -- deterministic synthetic JSON workloads for bench.py and tests
 * (BASELINE.json configs 2-4).  Host-only C, no GPU.
 *
 * Every generated unit is one complete, valid JSON document
 *   {"statuses":[ <tweet-like record>, ... ],"search_metadata":{...,"pad":"xx"}}
 * whose byte length is forced to == 77 (mod 128), so that when the unit is
 * repeated to fill 1 GiB the 64-byte block / 128-byte step / 16 KiB tile
 * boundaries fall at every phase of the text.
 *
 *   mode 0  minified ASCII "twitter-like"       (config 2)
 *   mode 1  UTF-8-heavy string bodies + escapes (config 3)
 *   indent  0 = minified, >0 = pretty-printed with that many spaces per level
 *           (config 4); newline is "\n", or "\r\n" when crlf != 0, tabs when
 *           indent < 0.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef struct {
    uint8_t *out;
    uint64_t cap, pos;
    uint64_t rng;
    int mode, indent, crlf, depth;
    int overflow;
} Gen;

static uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint32_t rnd(Gen *g, uint32_t n) { return (uint32_t)(splitmix64(&g->rng) % n); }

static void put(Gen *g, const void *p, uint64_t n) {
    if (g->pos + n > g->cap) {
        g->overflow = 1;
        return;
    }
    memcpy(g->out + g->pos, p, n);
    g->pos += n;
}
static void putc_(Gen *g, char c) { put(g, &c, 1); }
static void puts_(Gen *g, const char *s) { put(g, s, strlen(s)); }

static void newline(Gen *g) {
    if (g->indent == 0) return;
    if (g->crlf) putc_(g, '\r');
    putc_(g, '\n');
    if (g->indent < 0) {
        for (int i = 0; i < g->depth; i++) putc_(g, '\t');
    } else {
        for (int i = 0; i < g->depth * g->indent; i++) putc_(g, ' ');
    }
}

static void utf8_put(Gen *g, uint32_t cp) {
    uint8_t b[4];
    if (cp < 0x80) {
        b[0] = (uint8_t)cp;
        put(g, b, 1);
    } else if (cp < 0x800) {
        b[0] = 0xC0 | (cp >> 6);
        b[1] = 0x80 | (cp & 0x3F);
        put(g, b, 2);
    } else if (cp < 0x10000) {
        b[0] = 0xE0 | (cp >> 12);
        b[1] = 0x80 | ((cp >> 6) & 0x3F);
        b[2] = 0x80 | (cp & 0x3F);
        put(g, b, 3);
    } else {
        b[0] = 0xF0 | (cp >> 18);
        b[1] = 0x80 | ((cp >> 12) & 0x3F);
        b[2] = 0x80 | ((cp >> 6) & 0x3F);
        b[3] = 0x80 | (cp & 0x3F);
        put(g, b, 4);
    }
}

static const char ASCII_ALPHA[] =
    "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789    __..::#@--";

/* string body of ~len characters; escape_pct % of strings carry escapes */
static void gen_string(Gen *g, uint32_t len, int allow_escapes) {
    if (g->mode == 1) len = len * 3 + 8; /* UTF-8-heavy: string bodies dominate the document */
    putc_(g, '"');
    int with_esc = allow_escapes && (rnd(g, 100) < (g->mode == 1 ? 10u : 2u));
    for (uint32_t i = 0; i < len; i++) {
        if (with_esc && rnd(g, 12) == 0) {
            switch (rnd(g, g->mode == 1 ? 7 : 5)) {
                case 0: puts_(g, "\\\""); break;
                case 1: puts_(g, "\\\\"); break;
                case 2: puts_(g, "\\/"); break;
                case 3: puts_(g, "\\n"); break;
                case 4: puts_(g, "\\\\\\\""); break; /* backslash then escaped quote */
                case 5: { /* \uXXXX BMP */
                    char t[8];
                    snprintf(t, sizeof t, "\\u%04x", 0x00A0 + rnd(g, 0x2000));
                    puts_(g, t);
                    break;
                }
                default: { /* surrogate pair */
                    char t[16];
                    snprintf(t, sizeof t, "\\ud83d\\ude%02x", rnd(g, 0x50));
                    puts_(g, t);
                    break;
                }
            }
            continue;
        }
        if (g->mode == 1) {
            uint32_t r = rnd(g, 100);
            if (r < 40)
                utf8_put(g, 0x80 + rnd(g, 0x800 - 0x80));
            else if (r < 75)
                utf8_put(g, 0x4E00 + rnd(g, 0x9FFF - 0x4E00));
            else if (r < 85)
                utf8_put(g, 0x1F300 + rnd(g, 0x1FAFF - 0x1F300));
            else
                putc_(g, ASCII_ALPHA[rnd(g, sizeof ASCII_ALPHA - 1)]);
        } else {
            putc_(g, ASCII_ALPHA[rnd(g, sizeof ASCII_ALPHA - 1)]);
        }
    }
    putc_(g, '"');
}

/* log-uniform length in [2,140] */
static uint32_t text_len(Gen *g) {
    uint32_t bits = 1 + rnd(g, 7); /* 1..7 */
    uint32_t v = 2 + rnd(g, 1u << bits);
    return v > 140 ? 140 : v;
}

static void key(Gen *g, const char *k, int *first) {
    if (!*first) putc_(g, ',');
    *first = 0;
    newline(g);
    putc_(g, '"');
    puts_(g, k);
    putc_(g, '"');
    putc_(g, ':');
    if (g->indent) putc_(g, ' ');
}
static void open_(Gen *g, char c) {
    putc_(g, c);
    g->depth++;
}
static void close_(Gen *g, char c, int empty) {
    g->depth--;
    if (!empty) newline(g);
    putc_(g, c);
}
static void put_u64(Gen *g, uint64_t v) {
    char t[32];
    snprintf(t, sizeof t, "%llu", (unsigned long long)v);
    puts_(g, t);
}
static void put_id(Gen *g) { put_u64(g, 100000000000000ull + splitmix64(&g->rng) % 899999999999999999ull); }

static void gen_user(Gen *g) {
    int f = 1;
    open_(g, '{');
    key(g, "id", &f); put_u64(g, 1000 + rnd(g, 2000000000u));
    key(g, "name", &f); gen_string(g, 3 + rnd(g, 18), 1);
    key(g, "screen_name", &f); gen_string(g, 3 + rnd(g, 12), 0);
    key(g, "location", &f); gen_string(g, rnd(g, 24), 1);
    key(g, "description", &f); gen_string(g, text_len(g), 1);
    key(g, "url", &f);
    if (rnd(g, 3) == 0) puts_(g, "null"); else puts_(g, "\"http:\\/\\/t.co\\/AbCdEf123\"");
    key(g, "followers_count", &f); put_u64(g, rnd(g, 100000));
    key(g, "friends_count", &f); put_u64(g, rnd(g, 5000));
    key(g, "verified", &f); puts_(g, rnd(g, 10) == 0 ? "true" : "false");
    key(g, "utc_offset", &f);
    if (rnd(g, 4) == 0) puts_(g, "null"); else { if (rnd(g, 2)) putc_(g, '-'); put_u64(g, rnd(g, 43200)); }
    key(g, "lang", &f); puts_(g, "\"en\"");
    close_(g, '}', 0);
}

static void gen_record(Gen *g) {
    int f = 1;
    open_(g, '{');
    key(g, "created_at", &f); puts_(g, "\"Mon Sep 24 03:35:21 +0000 2012\"");
    key(g, "id", &f); put_id(g);
    key(g, "id_str", &f); putc_(g, '"'); put_id(g); putc_(g, '"');
    key(g, "text", &f); gen_string(g, text_len(g), 1);
    key(g, "source", &f);
    puts_(g, "\"<a href=\\\"http:\\/\\/twitter.com\\\" rel=\\\"nofollow\\\">web<\\/a>\"");
    key(g, "truncated", &f); puts_(g, "false");
    key(g, "in_reply_to_status_id", &f);
    if (rnd(g, 3)) puts_(g, "null"); else put_id(g);
    key(g, "user", &f); gen_user(g);
    key(g, "geo", &f); puts_(g, "null");
    key(g, "coordinates", &f);
    if (rnd(g, 8)) puts_(g, "null");
    else {
        char t[48];
        snprintf(t, sizeof t, "[%d.%04u,-%d.%05u]", (int)rnd(g, 90), rnd(g, 10000), (int)rnd(g, 180), rnd(g, 100000));
        puts_(g, t);
    }
    key(g, "retweet_count", &f); put_u64(g, rnd(g, 1000));
    key(g, "favorited", &f); puts_(g, rnd(g, 2) ? "true" : "false");
    key(g, "entities", &f);
    {
        int e = 1;
        open_(g, '{');
        key(g, "hashtags", &e);
        uint32_t nh = rnd(g, 3);
        open_(g, '[');
        for (uint32_t i = 0; i < nh; i++) {
            if (i) putc_(g, ',');
            newline(g);
            int h = 1;
            open_(g, '{');
            key(g, "text", &h); gen_string(g, 3 + rnd(g, 10), 0);
            key(g, "indices", &h);
            open_(g, '[');
            newline(g); put_u64(g, rnd(g, 100)); putc_(g, ',');
            newline(g); put_u64(g, 100 + rnd(g, 40));
            close_(g, ']', 0);
            close_(g, '}', 0);
        }
        close_(g, ']', nh == 0);
        key(g, "urls", &e); puts_(g, "[]");
        key(g, "user_mentions", &e); puts_(g, "[]");
        close_(g, '}', 0);
    }
    key(g, "score", &f);
    {
        char t[40];
        snprintf(t, sizeof t, "%u.%03ue-%u", rnd(g, 10), rnd(g, 1000), rnd(g, 9));
        puts_(g, t);
    }
    key(g, "lang", &f); puts_(g, "\"en\"");
    close_(g, '}', 0);
}

/* returns the unit length, 0 on overflow of `cap` */
uint64_t msj_gen_unit(uint8_t *out, uint64_t cap, uint64_t target_bytes, uint64_t seed, int mode,
                      int indent, int crlf) {
    Gen g;
    memset(&g, 0, sizeof g);
    g.out = out;
    g.cap = cap;
    g.rng = seed;
    g.mode = mode;
    g.indent = indent;
    g.crlf = crlf;
    int f = 1;
    open_(&g, '{');
    key(&g, "statuses", &f);
    open_(&g, '[');
    int first = 1;
    while (g.pos + 4096 < target_bytes && !g.overflow) {
        if (!first) putc_(&g, ',');
        first = 0;
        newline(&g);
        gen_record(&g);
    }
    close_(&g, ']', first);
    key(&g, "search_metadata", &f);
    {
        int m = 1;
        open_(&g, '{');
        key(&g, "completed_in", &m); puts_(&g, "0.035");
        key(&g, "count", &m); put_u64(&g, 4);
        key(&g, "pad", &m);
        putc_(&g, '"');
        /* bytes still to come after the pad body: closing quote, close of
         * search_metadata, close of the root object (with pretty newlines) */
        uint64_t tail = 1;
        {
            Gen t = g;
            uint8_t scratch[512];
            t.out = scratch;
            t.cap = sizeof scratch;
            t.pos = 0;
            close_(&t, '}', 0);
            close_(&t, '}', 0);
            tail += t.pos;
        }
        uint64_t total = g.pos + tail;
        uint64_t want = 77;
        uint64_t padn = (want + 128 - (total % 128)) % 128;
        for (uint64_t i = 0; i < padn; i++) putc_(&g, 'x');
        putc_(&g, '"');
        close_(&g, '}', 0);
    }
    close_(&g, '}', 0);
    if (g.overflow) return 0;
    return g.pos;
}

/* Synthetic density extremes for the config-4 sweep.
 * kind 0: "[[[[...]]]]"  (every byte structural, d = 1.0; `n` is rounded to even)
 * kind 1: "[1,1,1,...,1]" (d ~ 1.0 too: every byte a structural start) -> use "[10,10,...]" d = 2/3
 * kind 2: one giant string "\"aaaa...\"" (d ~ 0)
 * kind 3: spaces then a single scalar (d ~ 0, whitespace)
 */
uint64_t msj_gen_extreme(uint8_t *out, uint64_t n, int kind) {
    if (n < 8) return 0;
    switch (kind) {
        case 0: {
            uint64_t h = n / 2;
            memset(out, '[', h);
            memset(out + h, ']', h);
            return 2 * h;
        }
        case 1: {
            uint64_t p = 0;
            out[p++] = '[';
            while (p + 4 < n) {
                out[p++] = '1';
                out[p++] = '0';
                out[p++] = ',';
            }
            out[p++] = '7';
            out[p++] = ']';
            return p;
        }
        case 4: {  /* [123,123,...,7] : one scalar start + one comma per 4 bytes, d = 0.5 */
            uint64_t p = 0;
            out[p++] = '[';
            while (p + 6 < n) {
                out[p++] = '1';
                out[p++] = '2';
                out[p++] = '3';
                out[p++] = ',';
            }
            out[p++] = '7';
            out[p++] = ']';
            return p;
        }
        case 5: {  /* [1234,1234,...,7] : d = 0.4 (1 638 structurals per 4 KiB tile) */
            uint64_t p = 0;
            out[p++] = '[';
            while (p + 7 < n) {
                out[p++] = '1';
                out[p++] = '2';
                out[p++] = '3';
                out[p++] = '4';
                out[p++] = ',';
            }
            out[p++] = '7';
            out[p++] = ']';
            return p;
        }
        case 6:    /* [123,1234,123,1234,...,7] : d = 4/9 = 0.444, NOT a whole number of indices per 64-byte block */
        case 7: {  /* [12,123,12,123,...,7]     : d = 4/7 = 0.571 (2 341 structurals per tile: the block-wise emission) */
            const char *pat = kind == 6 ? "123,1234," : "12,123,";
            const uint64_t pl = strlen(pat);
            uint64_t p = 0;
            out[p++] = '[';
            while (p + pl + 2 < n) {
                memcpy(out + p, pat, pl);
                p += pl;
            }
            out[p++] = '7';
            out[p++] = ']';
            return p;
        }
        case 2:
            out[0] = '"';
            memset(out + 1, 'a', n - 2);
            out[n - 1] = '"';
            return n;
        default:
            memset(out, ' ', n);
            out[n - 1] = '1';
            return n;
    }
}
