// tokens_launch.h -- launch interface between api.cpp and tokens_kernel.hip (rows f1 / f2 / f4 of SURVEY.md section 8).
#pragma once
#include <stdint.h>

#include "../../include/msj_stage1.h"

// What a token call takes besides its arrays.  The first three are TEST HOOKS kept per context
// (msj_debug_set_span_limits / msj_debug_set_span_mode; 0xFFFFFFFF / 0 = the built-in behaviour): a hook left set by
// one test or tool cannot change the data path of another context.
struct msj_token_opts {
    uint32_t span_mode = 0;             // 0 by the density of the index, 1 the kernel organised by tokens, 2 by tiles
    uint32_t lds_limit = 0xFFFFFFFFu;   // stretches over this many bytes take the span kernels' global-memory path
    uint32_t fix_cap = 0xFFFFFFFFu;     // entries of the fix-up list
    // the msj_tokens_result (device) of the call that covered the tokens IN FRONT of this call's, or null at the start of
    // a stream: the running depth walk_document keeps (json_iterator.mojo:84-90,173-180) goes on from its final_depth,
    // and this call's min / max / final are those of the stream so far
    const msj_tokens_result *d_prev = nullptr;
    // msj_stage2_prep_segments (bracket partners over a whole shard): match[] values are positions in the SHARD's output
    // arrays -- this call's token index + match_bias -- and the brackets this call could not pair (their container is cut
    // by the call's border) are left in d_resid for the stitch behind the last segment (tokens_kernel.hip, "residuals")
    uint32_t match_bias = 0;
    uint32_t *d_resid = nullptr;
    // msj_*_pairs_device: the containers as {open, close} records in the order of their opening brackets (msj_bracket_pair),
    // instead of a partner index per token
    msj_bracket_pair *d_pairs = nullptr;
};

// residual brackets of one call (device, uint32 words): [0] unclosed opening brackets, [1] closing brackets without a
// partner, [2..3] spare; then MSJ_RESID_CAP positions of the former -- entry j = the one at depth final_depth - 1 - j --
// and MSJ_RESID_CAP of the latter -- entry k = the one at depth start_depth - 1 - k (token indices local to the call)
#define MSJ_RESID_CAP 65536u
#define MSJ_RESID_WORDS (4u + 2u * MSJ_RESID_CAP)
#define MSJ_STITCH_MAX_SEGMENTS 32u
struct msj_stitch_args {
    uint32_t n_segments;
    uint32_t offsets[MSJ_STITCH_MAX_SEGMENTS];       // element offset of every segment's slices in the shard's output arrays
    const uint32_t *resid[MSJ_STITCH_MAX_SEGMENTS];  // its residual brackets
};
int msj_launch_stitch_partners(const msj_stitch_args &a, uint32_t *d_match, msj_tokens_result *d_results, const msj_tokens_result *d_prev,
                               void *stream);

extern "C" uint64_t msj_tokens_workspace_bytes(uint64_t n, int with_match);
extern "C" uint64_t msj_stage2_prep_workspace_bytes(uint64_t n, uint64_t len, int with_match);
extern "C" uint64_t msj_span_fix_bytes(void);
extern "C" void *msj_tokens_doc_aggregates(int32_t *d_ws, uint64_t n);
int msj_launch_tokens(const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n, uint8_t *d_type, int32_t *d_depth,
                      uint32_t *d_match, msj_tokens_result *d_result, int32_t *d_ws, void *stream, const msj_token_opts &o);
int msj_launch_token_spans(const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n, uint32_t *d_end, uint8_t *d_flags,
                           int32_t *d_ws, uint32_t *d_fix, void *stream, const msj_token_opts &o);
int msj_launch_stage2_prep(const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n, uint8_t *d_type, int32_t *d_depth,
                           uint32_t *d_match, uint32_t *d_end, uint8_t *d_flags, msj_tokens_result *d_result, int32_t *d_ws,
                           uint32_t *d_fix, void *stream, const msj_token_opts &o);
