/* lfx_testhooks.h — host-only test hooks exported by liblfx.so.
 *
 * NOT part of the drop-in boundary (include/lfx.h is).  These four entry points run, on the HOST, the very sources
 * the device kernels compile (lfx_huff.h: code-length builder, canonical codes, block header; lfx_plan.h: the write
 * schedule → chunk/block planner) so that `pytest -m "not gpu"` can check that logic (against the CPU restatement under tests) on a machine
 * without a GPU.  No compression or decompression of data can be reached through them; the product entry points in
 * lfx.h never call them.  tests/test_host_pipeline.py is their only user; tests/test_abi.py checks that the library
 * exports exactly lfx.h ∪ lfx_testhooks.h.
 */
#ifndef LFX_TESTHOOKS_H
#define LFX_TESTHOOKS_H

#include "lfx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One block's Huffman build, as huffman_kernel runs it (src/huffman.rs:192-362, src/deflate/symbol.rs:486-540):
 * hist320 = 288 literal/length + 32 distance counts; type 1 = fixed, 2 = dynamic.  Outputs: lit288/dist32 packed
 * (code << 8 | width), hdr160 = header bit words, *hdr_bits, *body_bits. */
int lfx_debug_huff_block(const uint32_t *hist320, uint32_t type, uint32_t *lit288, uint32_t *dist32, uint32_t *hdr160,
                         uint32_t *hdr_bits, uint64_t *body_bits);

/* The planner on a schedule over n input bytes (libflate_lz77/src/default.rs:60-70, src/deflate/encode.rs:266-316):
 * chunk_out rows {in_off, len, block, flags}, block_out rows {type, final, first_chunk, n_chunks, in_off, in_len}. */
int lfx_debug_plan(int format, const lfx_encode_opts *o, const lfx_schedule *s, uint64_t n, uint64_t *chunk_out,
                   size_t max_chunks, size_t *n_chunks, uint64_t *block_out, size_t max_blocks, size_t *n_blocks);

/* The same plan collected the way the stream encoder does it: closed blocks are taken out after every take_every-th
 * schedule event and the remainder rebased; must equal lfx_debug_plan's answer. */
int lfx_debug_plan_incremental(int format, const lfx_encode_opts *o, const lfx_schedule *s, uint64_t n,
                               uint32_t take_every, uint64_t *chunk_out, size_t max_chunks, size_t *n_chunks,
                               uint64_t *block_out, size_t max_blocks, size_t *n_blocks);

/* Symbol::code / extra bits / distance closed forms (src/deflate/symbol.rs:343-386):
 * out6 = {len symbol, extra width, extra value, dist symbol, extra width, extra value}. */
void lfx_debug_symbols(uint32_t length, uint32_t distance, uint32_t *out6);

#ifdef __cplusplus
}
#endif
#endif
