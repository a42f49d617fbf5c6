/*
 * megalania_interfaces.h -- the two plugin seams of blackle/Megalania, kept ABI-identical.
 *
 * The reference routes every coded bit through `EncoderInterface` (encoder_interface.h:7-13;
 * called only from probability_model.c:7,19) and every output byte through `OutputInterface`
 * (output_interface.h:8-13; called from range_encoder.c:29 and lzma_header_encoder.c:14-20).
 * The GPU path does not change them: the final packet slab comes back from mgl_sa_best() in
 * the reference's LZMAPacket layout and is emitted on the host through these two vtables, by
 * the reference's own range_encoder.c / lzma_header_encoder.c when linked into the
 * reference, or by the equivalents in megalania_amd/host/ when used standalone.
 *
 * Same struct tags, same member order, same callback signatures: an object built by either
 * side can be handed to the other.
 */
#ifndef MEGALANIA_INTERFACES_H
#define MEGALANIA_INTERFACES_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

/* probability.h:6-8 -- 11-bit probability that the next bit is 0, in a uint16_t */
#ifndef Prob
#define Prob uint16_t
#endif
#define MGL_NUM_BIT_MODEL_TOTAL_BITS 11
#define MGL_PROB_INIT_VAL (1 << (MGL_NUM_BIT_MODEL_TOTAL_BITS - 1))

/* encoder_interface.h:7-13 */
typedef struct EncoderInterface_struct EncoderInterface;
struct EncoderInterface_struct {
	void (*encode_bit)(EncoderInterface* enc, bool bit, Prob prob);
	void (*encode_direct_bits)(EncoderInterface* enc, unsigned bits, unsigned num_bits);
	void* private_data;
};

/* output_interface.h:8-13 */
typedef struct OutputInterface_struct OutputInterface;
struct OutputInterface_struct {
	bool (*write)(OutputInterface* out, const void* data, size_t data_size);
	void* private_data;
};

#endif /* MEGALANIA_INTERFACES_H */
