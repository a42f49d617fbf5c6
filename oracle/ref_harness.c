/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (never shipped, never linked into the product).
 *
 * A thin flat-C facade over the *reference's own objects*.  It is compiled together
 * with the reference sources where they lie (/root/reference/src/*.c minus main.c,
 * exactly the file set of the reference's test target, Makefile:6) into
 * oracle/_ref/libmegalania_ref.so by oracle/Makefile.  Nothing from the reference is
 * copied into this repository: this file only *calls* the reference's exported
 * functions through the reference's own headers (found via -I/root/reference/src).
 *
 * Used by tests/ (in the build container only -- /root/reference does not exist on the
 * GPU box, the prebuilt .so travels) to
 *   - validate oracle/mgl_oracle.c against the real thing, and
 *   - generate the committed fixtures under tests/golden/ (tools/make_golden.py).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "lzma_state.h"
#include "lzma_packet.h"
#include "lzma_packet_encoder.h"
#include "lzma_header_encoder.h"
#include "perplexity_encoder.h"
#include "range_encoder.h"
#include "substring_enumerator.h"
#include "packet_enumerator.h"
#include "top_k_packet_finder.h"
#include "packet_slab.h"
#include "packet_slab_neighbour.h"

size_t ref_sizeof_packet(void) { return sizeof(LZMAPacket); }
size_t ref_sizeof_state(void) { return sizeof(LZMAState); }
size_t ref_num_probs(void) { return sizeof(LZMAProbabilityModel) / sizeof(Prob); }

/* Cost a whole position-indexed slab with the reference's perplexity backend
 * (the loop of main.c:116-118 with perplexity_encoder instead of range_encoder).
 * cum[k] = cumulative perplexity after the k-th walked packet. */
uint64_t ref_cost_slab(const uint8_t* data, size_t n, const LZMAPacket* slab,
                       uint64_t* cum, size_t* npackets,
                       uint16_t* probs_out, uint8_t* ctx_state_out, uint32_t* dists_out)
{
	LZMAState st;
	LZMAProperties props = { 0, 0, 0 };
	lzma_state_init(&st, data, n, props);
	uint64_t perp = 0;
	EncoderInterface enc;
	perplexity_encoder_new(&enc, &perp);
	size_t k = 0;
	while (st.position < st.data_size) {
		lzma_encode_packet(&st, &enc, slab[st.position]);
		if (cum) cum[k] = perp;
		k++;
	}
	if (npackets) *npackets = k;
	if (probs_out) memcpy(probs_out, &st.probs, sizeof(st.probs));
	if (ctx_state_out) *ctx_state_out = st.ctx_state;
	if (dists_out) memcpy(dists_out, st.dists, sizeof(st.dists));
	return perp;
}

/* Top-K at `position` (which must lie on the slab's walk) from the adapted state.
 * Output in the reference's pop order (worst first, best last); costs[i] is the
 * integer perplexity/length the reference turns into its float key. */
size_t ref_top_k(const uint8_t* data, size_t n, LZMAPacket* slab, size_t position, size_t k,
                 LZMAPacket* out, uint64_t* costs)
{
	LZMAState st;
	LZMAProperties props = { 0, 0, 0 };
	lzma_state_init(&st, data, n, props);
	uint64_t perp = 0;
	EncoderInterface enc;
	perplexity_encoder_new(&enc, &perp);
	while (st.position < position) {
		lzma_encode_packet(&st, &enc, slab[st.position]);
	}
	if (st.position != position) return (size_t)-1;

	PacketEnumerator* en = packet_enumerator_new(data, n);
	TopKPacketFinder* finder = top_k_packet_finder_new(k, en);
	top_k_packet_finder_find(finder, &st, slab);
	size_t count = 0;
	LZMAPacket p;
	while (top_k_packet_finder_pop(finder, &p)) {
		LZMAState tmp = st;
		uint64_t c = 0;
		EncoderInterface e2;
		perplexity_encoder_new(&e2, &c);
		lzma_encode_packet(&tmp, &e2, p);
		out[count] = p;
		costs[count] = c / (tmp.position - position);
		count++;
	}
	top_k_packet_finder_free(finder);
	packet_enumerator_free(en);
	return count;
}

typedef struct { uint32_t* offs; uint32_t* lens; size_t cap; size_t count; } SubCollect;
static void sub_collect(void* ud, size_t offset, size_t length)
{
	SubCollect* c = (SubCollect*)ud;
	if (c->count < c->cap) { c->offs[c->count] = (uint32_t)offset; c->lens[c->count] = (uint32_t)length; }
	c->count++;
}
size_t ref_substrings(const uint8_t* data, size_t n, size_t pos, size_t max_len,
                      uint32_t* offs, uint32_t* lens, size_t cap)
{
	SubstringEnumerator* em = substring_enumerator_new(data, n, 2, max_len);
	SubCollect c = { offs, lens, cap, 0 };
	substring_enumerator_for_each(em, pos, sub_collect, &c);
	substring_enumerator_free(em);
	return c.count;
}

/* The hot loop of main.c:78-102 for iterations [i_begin, i_end) of one epoch, calling the
 * reference's own neighbour generator.  The caller seeds glibc rand() (ref_srand) once, as
 * main.c:68 does.  trace[2*it] = neighbour perplexity, trace[2*it+1] = accepted flag, per
 * *successful* generate (failed generates are retried like main.c:81-84 and not traced). */
void ref_srand(unsigned seed) { srand(seed); }
int ref_rand(void) { return rand(); }

int ref_sa_iters(const uint8_t* data, size_t n, LZMAPacket* slab_io, LZMAPacket* best_io,
                 uint64_t* cur_io, uint64_t* best_cost_io, unsigned step, int num_iters,
                 int i_begin, int i_end, uint64_t* trace, uint64_t* undo_total)
{
	LZMAState init_state;
	LZMAProperties props = { 0, 0, 0 };
	lzma_state_init(&init_state, data, n, props);
	PacketEnumerator* en = packet_enumerator_new(data, n);
	TopKPacketFinder* finder = top_k_packet_finder_new(20, en);
	PacketSlab* slab = packet_slab_new(n);
	LZMAPacket* packets = packet_slab_packets(slab);
	memcpy(packets, slab_io, sizeof(LZMAPacket) * n);
	uint64_t current_perplexity = *cur_io, best_perplexity = *best_cost_io;
	size_t t = 0;
	uint64_t undos = 0;
	PacketSlabNeighbour neighbour;
	for (int i = i_begin; i < i_end; i++) {
		packet_slab_neighbour_new(&neighbour, slab, init_state);
		bool success = packet_slab_neighbour_generate(&neighbour, finder);
		if (!success) { i--; continue; }
		undos += packet_slab_neighbour_undo_count(&neighbour);
		bool transition = rand() % (i*i+1+step*num_iters/2) < sqrt(num_iters);
		int accepted = 0;
		if (current_perplexity == 0 || neighbour.perplexity < current_perplexity || transition) {
			current_perplexity = neighbour.perplexity;
			accepted = 1;
			if (best_perplexity == 0 || current_perplexity < best_perplexity) {
				best_perplexity = current_perplexity;
				memcpy(best_io, packets, sizeof(LZMAPacket) * n);
			}
		} else {
			packet_slab_neighbour_undo(&neighbour);
		}
		if (trace) { trace[2*t] = neighbour.perplexity; trace[2*t+1] = (uint64_t)accepted; }
		t++;
		packet_slab_neighbour_free(&neighbour);
	}
	memcpy(slab_io, packets, sizeof(LZMAPacket) * n);
	*cur_io = current_perplexity;
	*best_cost_io = best_perplexity;
	if (undo_total) *undo_total = undos;
	packet_slab_free(slab);
	top_k_packet_finder_free(finder);
	packet_enumerator_free(en);
	return (int)t;
}

/* Emission, main.c:110-119, into a memory buffer through the reference's OutputInterface. */
typedef struct { uint8_t* buf; size_t cap; size_t len; } MemOut;
static bool mem_write(OutputInterface* o, const void* d, size_t sz)
{
	MemOut* m = (MemOut*)o->private_data;
	if (m->len + sz <= m->cap) memcpy(m->buf + m->len, d, sz);
	m->len += sz;
	return true;
}
size_t ref_emit(const uint8_t* data, size_t n, const LZMAPacket* slab, uint8_t* out, size_t cap)
{
	LZMAState st;
	LZMAProperties props = { 0, 0, 0 };
	lzma_state_init(&st, data, n, props);
	MemOut m = { out, cap, 0 };
	OutputInterface output = { mem_write, &m };
	lzma_encode_header(&st, &output);
	EncoderInterface enc;
	range_encoder_new(&enc, &output);
	while (st.position < st.data_size) {
		lzma_encode_packet(&st, &enc, slab[st.position]);
	}
	range_encoder_free(&enc);
	return m.len;
}
