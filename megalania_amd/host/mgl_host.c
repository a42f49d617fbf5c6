/*
 * mgl_host.c -- host emission path in C (see mgl_host.h for the map to the reference).
 * The bit model is shared with the GPU kernels through csrc/mgl_model.h: a packet is
 * planned once and its events are visited in slot order, which is coding order.
 */
#include "mgl_host.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ LZMA state / packets */

bool mgl_lzma_state_init(mgl_lzma_state* st, const uint8_t* data, size_t data_size, mgl_properties props)
{
	st->data = data;
	st->data_size = data_size;
	st->properties = props;
	st->layout = mgl_make_layout(props.lc, props.lp, props.pb);
	memset(&st->walk, 0, sizeof st->walk);
	st->probs = (Prob*)malloc(sizeof(Prob) * st->layout.total);
	if (st->probs == NULL) {
		fprintf(stderr, "Error: could not allocate the probability model in mgl_lzma_state_init\n");
		return false;
	}
	for (uint32_t i = 0; i < st->layout.total; i++) st->probs[i] = MGL_PROB_INIT_VAL;
	return true;
}

void mgl_lzma_state_free(mgl_lzma_state* st)
{
	free(st->probs);
	st->probs = NULL;
}

void mgl_lzma_encode_packet(mgl_lzma_state* st, EncoderInterface* enc, mgl_packet packet)
{
	const uint32_t pos = st->walk.pos;
	const uint32_t byte = st->data[pos];
	uint32_t match_byte = 0, prev_byte = 0;
	if (packet.type == MGL_LITERAL) {
		if (st->walk.ctx_state >= 7) match_byte = st->data[pos - st->walk.dists[0] - 1];
		if (pos > 0) prev_byte = st->data[pos - 1];
	}
	mgl_plan plan;
	mgl_plan_packet(&st->layout, &st->walk, packet.type, packet.dist, packet.len, byte, match_byte, prev_byte, &plan);
	for (uint32_t slot = 0; slot < plan.nev; slot++) {
		if (plan.ndirect && slot == plan.direct_after) {
			(*enc->encode_direct_bits)(enc, plan.direct_val, plan.ndirect);
		}
		uint32_t ctx, bit;
		mgl_plan_event(&plan, slot, &ctx, &bit);
		Prob* prob = &st->probs[ctx];
		(*enc->encode_bit)(enc, bit != 0, *prob); /* sink sees the pre-update value, probability_model.c:7 */
		*prob = (Prob)mgl_prob_update(*prob, bit);
	}
	mgl_advance(&st->walk, packet.type, packet.dist, packet.len);
}

void mgl_lzma_encode_header(const mgl_lzma_state* st, OutputInterface* output)
{
	uint8_t hdr[13];
	const mgl_properties* p = &st->properties;
	hdr[0] = (uint8_t)((p->pb * 5 + p->lp) * 9 + p->lc);
	const uint32_t dict = 0x400000; /* lzma_header_encoder.c:16 */
	for (int i = 0; i < 4; i++) hdr[1 + i] = (uint8_t)(dict >> (8 * i));
	const uint64_t size = (uint32_t)st->data_size; /* :19 goes through htole32 */
	for (int i = 0; i < 8; i++) hdr[5 + i] = (uint8_t)(size >> (8 * i));
	(*output->write)(output, &hdr[0], 1);
	(*output->write)(output, &hdr[1], 4);
	(*output->write)(output, &hdr[5], 8);
}

/* ------------------------------------------------------------------ range coder */

typedef struct {
	OutputInterface* output;
	uint64_t low;
	uint32_t range;
	uint8_t cache;
	uint64_t pending; /* cache byte + the 0xFF run behind it */
	/* range_encoder.c:18-38 hands every byte to OutputInterface.write on its own; here they are
	 * collected and handed over 64 KiB at a time through the same vtable (same bytes, same order) */
	size_t fill;
	bool write_failed;
	uint8_t buf[65536];
} mgl_rc;

#define MGL_RC_TOP 0x01000000u

static void rc_flush(mgl_rc* rc)
{
	if (rc->fill && !(*rc->output->write)(rc->output, rc->buf, rc->fill) && !rc->write_failed) {
		fprintf(stderr, "could not write %zu bytes\n", rc->fill); /* logged only, like range_encoder.c:29-31 */
		rc->write_failed = true;
	}
	rc->fill = 0;
}
static void rc_emit(mgl_rc* rc, uint8_t byte)
{
	rc->buf[rc->fill++] = byte;
	if (rc->fill == sizeof rc->buf) rc_flush(rc);
}

/* move the top byte of `low` out, resolving a possible carry into the bytes held back */
static void rc_shift_low(mgl_rc* rc)
{
	const uint32_t carry = (uint32_t)(rc->low >> 32);
	const uint32_t low32 = (uint32_t)rc->low;
	if (low32 < 0xFF000000u || carry) {
		uint8_t head = rc->cache;
		do {
			rc_emit(rc, (uint8_t)(head + carry));
			head = 0xFF;
		} while (--rc->pending);
		rc->cache = (uint8_t)(low32 >> 24);
	}
	rc->pending++;
	rc->low = (uint64_t)(low32 & 0x00FFFFFFu) << 8;
}

static void rc_encode_bit(EncoderInterface* enc, bool bit, Prob prob)
{
	mgl_rc* rc = (mgl_rc*)enc->private_data;
	const uint32_t bound = (rc->range >> MGL_NUM_BIT_MODEL_TOTAL_BITS) * prob;
	if (bit) {
		rc->low += bound;
		rc->range -= bound;
	} else {
		rc->range = bound;
	}
	while (rc->range < MGL_RC_TOP) {
		rc->range <<= 8;
		rc_shift_low(rc);
	}
}

static void rc_encode_direct_bits(EncoderInterface* enc, unsigned bits, unsigned num_bits)
{
	mgl_rc* rc = (mgl_rc*)enc->private_data;
	while (num_bits--) {
		rc->range >>= 1;
		if ((bits >> num_bits) & 1u) rc->low += rc->range;
		if (rc->range < MGL_RC_TOP) {
			rc->range <<= 8;
			rc_shift_low(rc);
		}
	}
}

bool mgl_range_encoder_new(EncoderInterface* enc, OutputInterface* output)
{
	mgl_rc* rc = (mgl_rc*)malloc(sizeof *rc);
	if (rc == NULL) {
		fprintf(stderr, "Error: could not allocate memory in mgl_range_encoder_new\n");
		return false;
	}
	rc->output = output;
	rc->low = 0;
	rc->range = 0xFFFFFFFFu;
	rc->cache = 0;
	rc->pending = 1;
	rc->fill = 0;
	rc->write_failed = false;
	enc->encode_bit = rc_encode_bit;
	enc->encode_direct_bits = rc_encode_direct_bits;
	enc->private_data = rc;
	return true;
}

void mgl_range_encoder_free(EncoderInterface* enc)
{
	mgl_rc* rc = (mgl_rc*)enc->private_data;
	for (int i = 0; i < 5; i++) rc_shift_low(rc);
	rc_flush(rc);
	free(rc);
	enc->private_data = NULL;
}

/* ------------------------------------------------------------------ perplexity backend */

static const uint16_t k_bit_cost[2048] = {
#include "../csrc/mgl_cost_table.inc"
};

static void perp_encode_bit(EncoderInterface* enc, bool bit, Prob prob)
{
	*(uint64_t*)enc->private_data += k_bit_cost[bit ? 2048 - prob : prob];
}
static void perp_encode_direct_bits(EncoderInterface* enc, unsigned bits, unsigned num_bits)
{
	(void)bits;
	*(uint64_t*)enc->private_data += (uint64_t)num_bits << MGL_NUM_BIT_MODEL_TOTAL_BITS;
}
void mgl_perplexity_encoder_new(EncoderInterface* enc, uint64_t* perplexity)
{
	enc->encode_bit = perp_encode_bit;
	enc->encode_direct_bits = perp_encode_direct_bits;
	enc->private_data = perplexity;
}

/* ------------------------------------------------------------------ outputs */

static bool file_sink_write(OutputInterface* output, const void* data, size_t data_size)
{
	return fwrite(data, data_size, 1, (FILE*)output->private_data) == 1;
}
void mgl_file_output_new(OutputInterface* output, FILE* file)
{
	output->write = file_sink_write;
	output->private_data = file;
}
static bool memory_sink_write(OutputInterface* output, const void* data, size_t data_size)
{
	mgl_memory_sink* s = (mgl_memory_sink*)output->private_data;
	const bool fits = s->len + data_size <= s->cap;
	if (fits) memcpy(s->buf + s->len, data, data_size);
	s->len += data_size; /* keeps counting so the caller learns the needed size */
	return fits;
}
void mgl_memory_output_new(OutputInterface* output, mgl_memory_sink* sink)
{
	output->write = memory_sink_write;
	output->private_data = sink;
}

/* ------------------------------------------------------------------ main.c:110-119 */

bool mgl_emit_stream(const uint8_t* data, size_t n, mgl_properties props, const mgl_packet* slab, OutputInterface* output)
{
	mgl_lzma_state st;
	if (!mgl_lzma_state_init(&st, data, n, props)) return false;
	/* the reference's emitter codes whatever it is given (main.c:116-118) and reads
	 * data[pos - dists[0] - 1] unguarded; here a slab is refused unless its walk is a valid parse of
	 * the input that the header's 4 MiB dictionary can decode: packet types and lengths, MATCH
	 * distances inside the window and the prefix, LONG_REP indices, and the copied bytes themselves */
	{
		mgl_wstate w;
		memset(&w, 0, sizeof w);
		const char* why = NULL;
		while (w.pos < n && !why) {
			const mgl_packet* p = &slab[w.pos];
			const size_t pos = w.pos;
			uint32_t src_dist = 0;
			if (p->type < MGL_LITERAL || p->type > MGL_LONG_REP || p->len == 0 || pos + p->len > n) why = "not a packet";
			else if (p->type == MGL_LITERAL) { if (p->len != 1) why = "literal longer than one byte"; }
			else if (p->type == MGL_SHORT_REP) {
				if (p->len != 1) why = "short rep longer than one byte";
				else if (w.dists[0] >= pos || data[pos] != data[pos - w.dists[0] - 1]) why = "short rep does not reproduce the input";
			} else {
				if (p->len < MGL_MIN_MATCH || p->len > MGL_MAX_MATCH) why = "match length outside 2..273";
				else if (p->type == MGL_LONG_REP && p->dist > 3) why = "rep index above 3";
				else {
					src_dist = p->type == MGL_MATCH ? p->dist : mgl_dist_at(&w, p->dist);
					if (src_dist >= pos) why = "distance reaches before the start of the input";
					else if (src_dist >= 0x400000u) why = "distance outside the 4 MiB dictionary of the header";
					else {
						/* overlapping copies are legal: compare byte by byte */
						const uint8_t* a = data + pos - src_dist - 1;
						const uint8_t* b = data + pos;
						for (uint32_t i = 0; i < p->len; i++) if (a[i] != b[i]) { why = "match does not reproduce the input"; break; }
					}
				}
			}
			if (why) {
				fprintf(stderr, "Error: slab entry at %zu: %s\n", pos, why);
				mgl_lzma_state_free(&st);
				return false;
			}
			mgl_advance(&w, p->type, p->dist, p->len);
		}
	}
	mgl_lzma_encode_header(&st, output);
	EncoderInterface enc;
	if (!mgl_range_encoder_new(&enc, output)) { mgl_lzma_state_free(&st); return false; }
	while (st.walk.pos < n) mgl_lzma_encode_packet(&st, &enc, slab[st.walk.pos]);
	mgl_range_encoder_free(&enc);
	mgl_lzma_state_free(&st);
	return true;
}
