/*
 * dropin_main.c -- TEST INFRASTRUCTURE: the drop-in proof.
 *
 * This is what the reference's src/main.c looks like with its SA loop (main.c:48-51, 64-105) replaced by
 * calls into libmegalania_hip.so, and nothing else touched: input mapping (main.c:34-46) and emission
 * (main.c:107-121) go through the reference's OWN memory_mapper.c, packet_slab.c, lzma_state.c,
 * lzma_header_encoder.c, range_encoder.c, lzma_packet_encoder.c, probability_model.c, file_output.c --
 * compiled in place from /root/reference/src by oracle/Makefile (build container only; the binary,
 * oracle/_ref/megalania_dropin, travels to the GPU box like the other oracle/_ref outputs).  Our file; it
 * only calls the reference's exported functions and the C ABI.  tests/test_gpu_dropin.py runs it and
 * compares the stream with the standalone host emitter's and with xz / liblzma.
 *
 *   megalania_dropin <file> [steps] [neighbours]        stream on stdout, progress on stderr
 */
#include <stdio.h>
#include <stdlib.h>

#include "file_output.h"
#include "lzma_header_encoder.h"
#include "lzma_packet_encoder.h"
#include "lzma_state.h"
#include "memory_mapper.h"
#include "packet_slab.h"
#include "range_encoder.h"

/* the C ABI takes the reference's own records: same layout (lzma_packet.h:13-17, lzma_state.h:53-57) */
#define MGL_NO_PACKET_TYPES
typedef LZMAPacket mgl_packet;
typedef LZMAProperties mgl_properties;
#include "megalania_hip.h"

int main(int argc, char** argv)
{
	if (argc < 2) {
		fprintf(stderr, "usage: %s filename [steps] [neighbours]\n", argv[0]);
		return -1;
	}
	const uint8_t* file_data;
	size_t file_size;
	if (map_file(argv[1], &file_data, &file_size) < 0) return -1; /* main.c:36 */
	if (file_size == 0) return 0;

	LZMAState init_state;
	LZMAProperties properties = { .lc = 0, .lp = 0, .pb = 0 }; /* main.c:45 */
	lzma_state_init(&init_state, file_data, file_size, properties);

	/* main.c:48-51, 64-105 replaced by: */
	mgl_sa_config cfg = { .seed = 1673551, /* main.c:68 */
		                  .neighbours_per_step = argc > 3 ? (uint32_t)strtoul(argv[3], NULL, 0) : 4096,
		                  .top_k = 20, /* main.c:49 */
		                  .iters_per_epoch = file_size }; /* main.c:67 */
	mgl_sa* sa = mgl_sa_create(file_data, file_size, properties, &cfg);
	if (sa == NULL) { fprintf(stderr, "%s\n", mgl_last_error()); return -1; }
	const unsigned long long steps = argc > 2 ? strtoull(argv[2], NULL, 0) : (file_size + cfg.neighbours_per_step - 1) / cfg.neighbours_per_step;
	mgl_sa_stats st;
	if (mgl_sa_run(sa, steps, &st) != MGL_OK) { fprintf(stderr, "%s\n", mgl_last_error()); return -1; }
	fprintf(stderr, "current file size: %f\tsteps: %llu\n", 18 + st.current_cost / 16384.f, steps); /* main.c:97-99 */

	PacketSlab* packet_slab_best = packet_slab_new(file_size); /* main.c:50-51 */
	LZMAPacket* packets_best = packet_slab_packets(packet_slab_best);
	uint64_t best_perplexity = 0;
	if (mgl_sa_best(sa, packets_best, &best_perplexity) != MGL_OK) { fprintf(stderr, "%s\n", mgl_last_error()); return -1; }
	mgl_sa_destroy(sa);
	fprintf(stderr, "best perplexity: %llu\n", (unsigned long long)best_perplexity);

	/* main.c:110-119, unchanged */
	OutputInterface output;
	file_output_new(&output, stdout);
	LZMAState state = init_state;
	lzma_encode_header(&state, &output);
	EncoderInterface enc;
	range_encoder_new(&enc, &output);
	while (state.position < state.data_size) {
		lzma_encode_packet(&state, &enc, packets_best[state.position]);
	}
	range_encoder_free(&enc);

	packet_slab_free(packet_slab_best);
	if (unmap(file_data, file_size) < 0) return -1;
	return 0;
}
