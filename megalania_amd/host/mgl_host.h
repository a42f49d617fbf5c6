/*
 * mgl_host.h -- host-side (C) mirror of the reference's emission path, over the kept
 * EncoderInterface / OutputInterface vtables (include/megalania_interfaces.h).
 *
 * Name map to the reference (same argument order and meaning, `mgl_` prefixed so that the
 * library can be linked next to the reference's own objects without symbol clashes):
 *
 *   mgl_range_encoder_new / _free     range_encoder.c:83-101   (_free flushes 5 bytes, then frees)
 *   mgl_perplexity_encoder_new        perplexity_encoder.c:19-24 (borrows the caller's uint64_t)
 *   mgl_lzma_state_init               lzma_state.c:16-27
 *   mgl_lzma_encode_packet            lzma_packet_encoder.c:169-194
 *   mgl_lzma_encode_header            lzma_header_encoder.c:11-21
 *   mgl_file_output_new               file_output.c:9-13
 *   mgl_memory_output_new             (no reference equivalent: OutputInterface over a buffer)
 *   mgl_emit_stream                   main.c:110-119 as one call
 *
 * Error conventions follow the reference: constructors that allocate return false/NULL and
 * print to stderr; a failed OutputInterface.write is only logged (range_encoder.c:29-31).
 */
#ifndef MGL_HOST_H
#define MGL_HOST_H

#include <stdio.h>
#include "../../include/megalania_interfaces.h"
#include "../../include/megalania_hip.h"
#include "../csrc/mgl_model.h"

#ifdef __cplusplus
extern "C" {
#endif

/* lzma_state.h:60-74.  Probabilities are heap-allocated because lc/lp change their number. */
typedef struct {
	const uint8_t* data;
	size_t data_size;
	mgl_properties properties;
	mgl_layout layout;
	mgl_wstate walk; /* ctx_state, rep distances, position */
	Prob* probs;
} mgl_lzma_state;

bool mgl_lzma_state_init(mgl_lzma_state* st, const uint8_t* data, size_t data_size, mgl_properties props);
void mgl_lzma_state_free(mgl_lzma_state* st);
void mgl_lzma_encode_packet(mgl_lzma_state* st, EncoderInterface* enc, mgl_packet packet);
void mgl_lzma_encode_header(const mgl_lzma_state* st, OutputInterface* output);

bool mgl_range_encoder_new(EncoderInterface* enc, OutputInterface* output);
void mgl_range_encoder_free(EncoderInterface* enc);
void mgl_perplexity_encoder_new(EncoderInterface* enc, uint64_t* perplexity);

void mgl_file_output_new(OutputInterface* output, FILE* file);
typedef struct { uint8_t* buf; size_t cap; size_t len; } mgl_memory_sink;
void mgl_memory_output_new(OutputInterface* output, mgl_memory_sink* sink);

/* header + every packet on the slab's walk through a fresh range coder; false on bad input */
bool mgl_emit_stream(const uint8_t* data, size_t n, mgl_properties props, const mgl_packet* slab, OutputInterface* output);

#ifdef __cplusplus
}
#endif
#endif
