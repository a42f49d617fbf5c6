/*
 * mgl_oracle.h -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's SA hot path (blackle/Megalania, src/):
 * the LZMA bit model, the perplexity cost backend, the bigram match index, the packet
 * enumerator, the top-K finder, and the neighbour generator.  Each function cites the
 * reference file:line it follows.  It exists so that tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg have something to check / time the HIP path against.
 * The product (megalania_amd/, include/) never includes, links or calls anything here.
 *
 * Pinning: validated against the reference itself (oracle/_ref, built from
 * /root/reference by oracle/Makefile) in tests/test_oracle_vs_ref.py, and against the
 * committed fixtures in tests/golden/ (generated from that same reference build by
 * tools/make_golden.py) in tests/test_oracle_golden.py.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* lzma_packet.h:5-17 -- same 12-byte record, same type codes. */
enum { ORC_INVALID = 0, ORC_LITERAL = 1, ORC_MATCH = 2, ORC_SHORT_REP = 3, ORC_LONG_REP = 4 };
typedef struct {
	uint8_t type;
	uint32_t dist; /* MATCH: distance-1; LONG_REP: rep index 0..3 */
	uint16_t len;  /* LITERAL/SHORT_REP: 1; else 2..273 */
} orc_packet;

typedef struct {
	uint32_t position;
	orc_packet old_packet;
	orc_packet new_packet;
} orc_diff;

typedef struct orc_ctx orc_ctx;

/* top-K selection semantics */
enum {
	ORC_TOPK_REF = 0,   /* sequential 20-slot max-heap with '<=' replacement, top_k_packet_finder.c:72-93 + max_heap.c */
	ORC_TOPK_CANON = 1  /* order-independent: best = (cost asc, enumeration order desc); what the HIP path implements */
};

orc_ctx* orc_new(const uint8_t* data, size_t n, int lc, int lp, int pb, uint32_t dict_limit);
void orc_free(orc_ctx* c);
size_t orc_num_probs(const orc_ctx* c);

/* The 2048-entry cost table, perplexity_table.h:4 / generate_table.py:7-10, recomputed. */
const uint16_t* orc_cost_table(void);

/* Walk a position-indexed slab from byte 0 with the perplexity backend
 * (lzma_packet_encoder.c:169-194 under perplexity_encoder.c:6-17). */
uint64_t orc_cost_slab(orc_ctx* c, const orc_packet* slab, uint64_t* cum, size_t* npackets,
                       uint16_t* probs_out, uint8_t* ctx_state_out, uint32_t* dists_out);

/* every coded bit of the walk (context index in reference struct order, bit, probability
 * before the update, position of its packet) + the walk state before each packet */
size_t orc_trace_events(orc_ctx* c, const orc_packet* slab, uint32_t* ev_ctx, uint8_t* ev_bit, uint16_t* ev_prob,
                        uint32_t* ev_pos, size_t cap, uint32_t* pk_pos, uint32_t* pk_state, size_t pk_cap, size_t* npk);

/* bigram match index query, substring_enumerator.c:85-105 */
size_t orc_substrings(orc_ctx* c, size_t pos, size_t max_len, uint32_t* offs, uint32_t* lens, size_t cap);

/* top_k_packet_finder.c:120-138 at `position` on the slab's walk; pop order (worst first). */
size_t orc_top_k(orc_ctx* c, const orc_packet* slab, size_t position, int mode, size_t k,
                 orc_packet* out, uint64_t* costs);

/* main.c:78-102 replayed with glibc rand() (caller seeds with orc_srand). */
void orc_srand(unsigned seed);
int orc_sa_iters(orc_ctx* c, orc_packet* slab_io, orc_packet* best_io, uint64_t* cur_io,
                 uint64_t* best_cost_io, unsigned step, int num_iters, int i_begin, int i_end,
                 uint64_t* trace, uint64_t* undo_total);

/* The batched ("device") SA semantics -- counter-based RNG, canonical top-K, target chosen
 * by position rejection sampling.  See DESIGN.md section 4; mirrored by the HIP kernels. */
uint32_t orc_draw(uint64_t seed, uint64_t step, uint32_t j, uint32_t n);
/* Generate neighbour j of `slab` at `step`; returns 1 on success.  The slab is left
 * mutated when keep != 0, restored otherwise.  diffs = (position, old, new) journal,
 * de-duplicated by position and sorted ascending. */
int orc_neighbour(orc_ctx* c, orc_packet* slab, uint64_t seed, uint64_t step, uint32_t j,
                  int keep, uint64_t* cost, orc_diff* diffs, size_t* ndiffs, size_t cap);
/* the same with the neighbour's status (1 ok, 0 no candidate, -1 dropped by the 64-entry journal
 * capacity of the device) and its window, 4 words: [0] target position, [1] first position from which
 * neighbour and base are coded identically again (n if never; ~0 when not ok), [2] the soft end (the first
 * meeting point in the rep-free tail of the base parse, else = [1]), [3] 1 when a rep packet of the neighbour
 * inside the window reads a rep distance from before the window */
int orc_neighbour_ex(orc_ctx* c, orc_packet* slab, uint64_t seed, uint64_t step, uint32_t j,
                     int keep, uint64_t* cost, orc_diff* diffs, size_t* ndiffs, size_t cap, uint32_t* window);
/* Run batched steps [step_begin, step_end) of K neighbours; mirrors mgl_sa_run.  iter0 = evaluations
 * already made in this epoch (the reference's i); modes (nullable = all 0): per step 0 = take the
 * best acceptable neighbour, 1 = take every acceptable neighbour that is the best of its window.
 * trace (nullable) gets 4 u64 per step: smallest acceptable cost (or ~0), neighbours accepted,
 * acceptable neighbours, current cost after the step. */
uint64_t orc_bulk_overlaps(void);  /* slab entries written by two taken journals of one bulk step (must stay 0) */
uint64_t orc_bulk_rollbacks(void); /* bulk steps taken back by the validity check since the library was loaded */
int orc_sa_batched(orc_ctx* c, orc_packet* slab_io, orc_packet* best_io, uint64_t* cur_io,
                   uint64_t* best_cost_io, uint64_t seed, uint32_t K, unsigned phase,
                   uint64_t iters_per_epoch, uint64_t iter0, uint64_t step_begin, uint64_t step_end,
                   const uint8_t* modes, uint64_t* trace, uint64_t* valid_evals, uint64_t* dropped);
/* batched mode's target rule: K != 0 = stratified by packet ordinal over K neighbours per step (the device's default;
 * orc_sa_batched takes its own K whenever this is non-zero), 0 = position draws (mgl_sa_config.flags & MGL_F_POSITION_TARGETS) */
void orc_set_strata(orc_ctx* c, uint32_t K);
/* cap on the match-index hits a top-K query enumerates (nearest first); mirrors mgl_sa_config.max_bucket_scan */
void orc_set_max_bucket_scan(orc_ctx* c, uint32_t m);

/* Opt-in Metropolis rule for orc_sa_batched (0 = the reference's rule); mirrors mgl_sa_set_temperature. */
void orc_set_temperature(orc_ctx* c, uint64_t temperature);

/* Emission (header + range coder), lzma_header_encoder.c:5-21 + range_encoder.c:18-101. */
size_t orc_emit(orc_ctx* c, const orc_packet* slab, uint8_t* out, size_t cap);

#ifdef __cplusplus
}
#endif
