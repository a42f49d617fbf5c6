/*
 * megalania_hip.h -- C ABI of the MI355X SA hot path (libmegalania_hip.so).
 *
 * Drop-in boundary for blackle/Megalania's simulated-annealing loop.  Everything here is
 * plain C: pointers, sizes, integer status codes.  The library owns all device memory; the
 * caller owns every host buffer it passes in.  One host thread per mgl_sa handle (the
 * reference is single-threaded: global rand(), stateful TopKPacketFinder).
 *
 * What each entry point replaces in the reference (paths relative to its src/):
 *
 *   mgl_sa_create     main.c:44-51   lzma_state_init + packet_enumerator_new (match index,
 *                                    substring_enumerator.c:26-47) + top_k_packet_finder_new(20)
 *                                    + packet_slab_new (all-literal, packet_slab.c:15-35)
 *   mgl_sa_begin_epoch main.c:69-77  fresh slab per epoch (all-literal, or a copy of best)
 *   mgl_sa_run        main.c:78-102  the hot loop: packet_slab_neighbour_generate
 *                                    (packet_slab_neighbour.c:154-173: prefix cost, mutate,
 *                                    top-K pick, repair, total perplexity), accept / undo
 *   mgl_sa_best       main.c:91      packets_best, handed back in the reference's own
 *                                    12-byte LZMAPacket layout (lzma_packet.h:13-17) so that
 *                                    main.c:110-119 (header + range coder) emits it unchanged
 *   mgl_cost_slab     the loop of main.c:116-118 run with perplexity_encoder
 *                                    (perplexity_encoder.c:6-17) instead of range_encoder
 *   mgl_top_k         top_k_packet_finder_find/pop (top_k_packet_finder.c:120-138)
 *   mgl_substrings    substring_enumerator_for_each (substring_enumerator.c:85-105)
 *   mgl_sa_destroy    main.c:107-108,121 the matching frees
 *
 * Error convention: the reference returns NULL / -1 / false and prints to stderr
 * (packet_slab.c:18-27, memory_mapper.c:12-31); here constructors return NULL and
 * operations return 0 or a negative MGL_E* code; mgl_last_error() has the text.
 * Nothing aborts; HIP errors are translated.
 */
#ifndef MEGALANIA_HIP_H
#define MEGALANIA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGL_OK 0
#define MGL_EINVAL (-1)   /* bad argument */
#define MGL_ENOMEM (-2)   /* host or device allocation failed */
#define MGL_EDEVICE (-3)  /* HIP runtime error (no GPU, launch failure, ...) */
#define MGL_ERANGE (-4)   /* position not on the slab's walk / buffer too small */

/* lzma_packet.h:5-17 -- identical layout (type @0, dist @4, len @8, sizeof == 12). */
#ifndef MGL_NO_PACKET_TYPES
typedef struct {
	uint8_t type;   /* 1 LITERAL, 2 MATCH, 3 SHORT_REP, 4 LONG_REP */
	uint32_t dist;  /* MATCH: distance-1; LONG_REP: rep index 0..3 */
	uint16_t len;   /* LITERAL/SHORT_REP: 1; else 2..273 */
} mgl_packet;
/* lzma_state.h:53-57 */
typedef struct {
	uint8_t lc, lp, pb;
} mgl_properties;
#endif

typedef struct {
	uint64_t seed;                /* main.c:68 uses 1673551 */
	uint32_t neighbours_per_step; /* K: candidate neighbours costed per SA step */
	uint32_t top_k;               /* main.c:49 uses 20; at most 32 */
	uint32_t dict_limit;          /* candidates need dist < dict_limit; 0 = 0x400000, the
	                                 dictionary size lzma_header_encoder.c:16 writes */
	uint32_t max_bucket_scan;     /* cap on match-index hits scanned per top-K query, nearest
	                                 first; 0 = unlimited (the reference's behaviour) */
	uint64_t iters_per_epoch;     /* main.c:67 num_iters (defaults to the input size) */
	int32_t device;               /* HIP device ordinal */
	uint32_t flags;               /* MGL_F_* */
} mgl_sa_config;

#define MGL_F_TIMING 1u   /* record HIP events around every kernel launch of mgl_sa_run */
#define MGL_F_FULLWALK 2u /* cost every neighbour by walking the slab to the end (the simple,
                             slow engine) instead of incrementally against the base chains;
                             both engines return identical numbers */

#define MGL_F_PROFILE 4u  /* diagnostic: per-phase cycle counters in the neighbour kernels */
#define MGL_F_SERIAL_BUILD 16u /* derive the base structures with the one-wavefront builder only (diagnostic) */
#define MGL_F_POSITION_TARGETS 32u /* neighbour j's target from up to 32 uniform position draws (the first that is a packet start)
                              * instead of the default: a packet of the j-th of K equal slices of the walk's packets, by ordinal.
                              * Both make every packet equally likely (packet_slab_neighbour.c:162-163 draws a packet ordinal);
                              * the default keeps the K targets of one step apart, so that fewer improving neighbours of a step
                              * overlap and a bulk step takes 95 % of them instead of 80 % */
#define MGL_F_NO_SNAPSHOTS 8u /* do not keep device copies of the all-literal / best base structures:
                              * mgl_sa_begin_epoch then re-derives them from the slab (less memory, slower) */

typedef struct {
	uint64_t steps;          /* SA steps executed by this call */
	uint64_t evaluations;    /* neighbour evaluations that produced a cost (successful generates) */
	uint64_t failed;         /* generates that found no candidate (main.c:81-84 retries those) */
	uint64_t accepted;       /* steps that moved to a neighbour */
	uint64_t improved;       /* steps that set a new best */
	uint64_t current_cost;   /* perplexity of the current slab, 1/2048-bit units; 0 = none yet */
	uint64_t best_cost;
	uint64_t packets;        /* packets on the current slab's walk */
	uint64_t packets_evaluated; /* sum over evaluations of the packets costed (for B_eval) */
	double gpu_ms_total;     /* first launch -> last launch, HIP events on the library's stream */
	double gpu_ms_neighbours;/* sum over the timed steps of the neighbour kernels' span (MGL_F_TIMING: every step of a run of at most 16 steps,
	                          * every fourth step of a longer one, at most 512; `neighbour_launches` says how many were timed) */
	double gpu_ms_rebuild;   /* the same steps: decision / selection and accept (or rebuild) */
	uint64_t neighbour_launches;
	uint64_t full_rebuilds;       /* accepted steps whose base update fell back to a full rebuild */
	uint64_t fallback_neighbours; /* neighbours costed by the full-walk kernel instead of incrementally */
	uint64_t second_pass_neighbours; /* neighbours redone incrementally with their lists in global memory */
	uint64_t bulk_steps;          /* steps that took every window-best acceptable neighbour (mgl_sa_set_accept_mode) */
	uint64_t dropped_neighbours;  /* generates dropped because their journal outgrew 64 slab entries (counted in `failed`;
	                               * the reference's undo stack grows without bound, packet_slab_undo_stack.c:70-77) */
	uint64_t improving_neighbours;/* evaluations that cost less than the slab they were made from */
	uint64_t bulk_rollbacks;      /* bulk steps whose combined parse failed the after-the-fact validation and were taken back
	                               * as a whole (the safety net of the soft window ends; never seen) */
	uint64_t bulk_double_writes;  /* slab entries that two journals taken by one bulk step both wanted to change (detected entry
	                               * by entry with a compare-and-swap; such a step is taken back, so it also counts above) */
	double gpu_ms_sim;            /* sum over launches of the chain re-simulation kernel k_sim, the path's dominant kernel, each
	                               * bracketed by HIP events on the stream it runs on (MGL_F_TIMING; split launch form only) */
	uint64_t sim_launches;        /* the launches summed in gpu_ms_sim */
	uint64_t sim_bytes_counted;   /* bytes of chain data (positions, events) and change lists k_sim's loads asked for, counted by
	                               * the kernel itself while mgl_debug_set key 4 is on (its algorithmic bytes; 0 otherwise) */
} mgl_sa_stats;

/* How a step of K costed neighbours moves the chain (the reference decides after every single
 * evaluation, main.c:86-96; see DESIGN.md section 4 for the batched rule both modes share):
 *   MGL_ACCEPT_SINGLE  the best acceptable neighbour of the step (base structures updated in place);
 *   MGL_ACCEPT_BULK    every acceptable neighbour that is the best of its own window -- hundreds to
 *                      thousands of moves per step while the slab is young -- followed by a parallel
 *                      rebuild of the base structures, which also yields the new slab's exact cost;
 *   MGL_ACCEPT_AUTO    (default) bulk steps while a step offers at least `bulk_threshold` improving
 *                      neighbours on average, single steps otherwise; decided per block of steps from
 *                      device counters only, so a run is reproducible.  bulk_threshold 0 = default. */
#define MGL_ACCEPT_AUTO 0
#define MGL_ACCEPT_SINGLE 1
#define MGL_ACCEPT_BULK 2

typedef struct {
	uint32_t position;
	mgl_packet old_packet;
	mgl_packet new_packet;
} mgl_diff;

typedef struct mgl_sa mgl_sa;

/* library / device */
const char* mgl_version(void);
const char* mgl_last_error(void);
int mgl_device_count(void);

/* lifecycle.  `data` is copied to the device; it is not retained. */
mgl_sa* mgl_sa_create(const uint8_t* data, size_t n, mgl_properties props, const mgl_sa_config* cfg);
void mgl_sa_destroy(mgl_sa* sa);

/* Start an epoch (main.c:71-77): phase = the reference's `step` (0..2); from_best != 0 copies
 * the best slab into the current one, else the current slab becomes all-literal.  Resets the
 * within-epoch iteration counter and the current cost. */
int mgl_sa_begin_epoch(mgl_sa* sa, unsigned phase, int from_best);
/* Replace the current slab (n entries, position-indexed).  Must be a valid parse. */
int mgl_sa_set_slab(mgl_sa* sa, const mgl_packet* packets);
/* Replace the current slab by a greedy LZ parse made on the device (not in the reference, whose
 * search always starts from the all-literal slab, main.c:71; SURVEY 8f-3 "greedy seeding"): every
 * position independently receives the longest match among the `candidates` nearest earlier
 * occurrences of its two and of its four leading bytes inside the dictionary window (nearest
 * among equals; len 2 only up to distance 128, len 3 up to 2^14; dropped when the next position
 * would take a longer one), or a literal; the slab walk picks the parse out of them.  Same effect on the handle as mgl_sa_set_slab. */
int mgl_sa_seed_greedy(mgl_sa* sa, uint32_t candidates);
/* Opt-in Metropolis accept rule (not in the reference, whose rule ignores the cost difference,
 * main.c:86).  temperature = 0 (default): the reference's rule.  temperature > 0, in
 * cost units (16384 per output byte, main.c:97): when the step's best neighbour does not improve,
 * the randomly drawn neighbour is accepted iff u < exp(-delta / t_eff) with u uniform, evaluated in
 * integers through the reference's log table (perplexity_table.h:4): delta * 2048 <=
 * t_eff * T[u], u in 1..2047, t_eff = temperature * (iters_per_epoch - i) / iters_per_epoch
 * (linear cooling inside the epoch).  Must be below 2^40. */
int mgl_sa_set_temperature(mgl_sa* sa, uint64_t temperature);
int mgl_sa_set_accept_mode(mgl_sa* sa, int mode, uint32_t bulk_threshold);
/* The modes of the steps of the last mgl_sa_run (0 single, 1 bulk), for replaying a run elsewhere; of a call of more
 * than 2^20 steps only the first 2^20 are kept.  In MGL_ACCEPT_AUTO the mode is chosen per block of 16 single (4 bulk)
 * steps; a block carries over from one mgl_sa_run call to the next, so the sequence of modes -- and with it the
 * trajectory -- does not depend on how a run is cut into calls; mgl_sa_begin_epoch, mgl_sa_set_slab, mgl_sa_seed_greedy
 * and mgl_sa_set_accept_mode start a fresh block whatever came before (bulk steps first, except single steps first
 * in an epoch that starts from the best slab). */
int mgl_sa_step_modes(mgl_sa* sa, uint8_t* modes_out, size_t cap, size_t* count);
/* Adopt a best slab found elsewhere (another chain / GPU): replaces best slab and best cost.
 * `perplexity` must be the slab's exact cost (it is re-derived on the device and checked). */
int mgl_sa_set_best(mgl_sa* sa, const mgl_packet* packets, uint64_t perplexity);
/* ---- chains on several GPUs (no reference counterpart: the reference is one process; main.c:75-77 is what
 * an exchange feeds).  One chain per GPU / process; mgl_comm wraps one RCCL communicator (librccl is loaded on
 * first use).  Rank 0 calls mgl_comm_unique_id and hands the 128 bytes to the others by any means (a file, a
 * launcher's store); every rank then calls mgl_comm_init on its device. */
typedef struct mgl_comm mgl_comm;
int mgl_comm_unique_id(uint8_t id_out[128]);
int mgl_comm_init(mgl_comm** comm_out, const uint8_t id[128], int rank, int world, int device);
/* The same handle over a host shared-memory file instead of RCCL (no reference counterpart either): keys and the packed
 * slab are staged through `path` (a file on a memory-backed file system, e.g. under /dev/shm).  For chains that share one
 * GPU -- RCCL refuses two ranks per device -- and for boxes without RCCL; mgl_sa_exchange_best runs the same protocol over
 * it.  Rank 0 creates the file (replacing any left by an earlier run) and removes it in mgl_comm_destroy; the others wait
 * for a file that carries this run's `nonce` and `world` (any value all ranks of one run agree on, different from run to
 * run).  Waits -- here and in every exchange -- give up with MGL_EDEVICE after MGL_COMM_TIMEOUT_S seconds (default 600). */
int mgl_comm_init_shm(mgl_comm** comm_out, const char* path, uint64_t nonce, int rank, int world, int device);
/* host transport only, no device involved: the minimum over the ranks of one word each (the first half of an exchange) */
int mgl_comm_min_u64(mgl_comm* comm, uint64_t mine, uint64_t* min_out);
void mgl_comm_destroy(mgl_comm* comm);
int mgl_comm_rank(const mgl_comm* comm);
int mgl_comm_world(const mgl_comm* comm);
/* One exchange: a single 8-byte ncclAllReduce(min) of (best_cost << 8 | rank), then ncclBroadcast of the
 * winner's best slab in its packed 8-byte device form, HBM to HBM over xGMI (the host transport stages both
 * through its file); chains whose own best is worse
 * adopt it as packets_best (mgl_sa_begin_epoch(.., from_best) continues from it and verifies it first).
 * winner_rank / winner_cost (nullable) receive the outcome; cost 0 = no chain has a best slab yet.
 * Collective: every rank of the communicator must call it. */
int mgl_sa_exchange_best(mgl_sa* sa, mgl_comm* comm, int* winner_rank, uint64_t* winner_cost);
/* The same hand-over through host memory in the packed device form (dist | len << 32 | type << 48, 8 bytes
 * per position), for transports other than RCCL.  Adopting does not verify; see mgl_sa_exchange_best. */
int mgl_sa_best_packed(mgl_sa* sa, uint64_t* packed_out, uint64_t* perplexity_out);
int mgl_sa_adopt_best_packed(mgl_sa* sa, const uint64_t* packed, uint64_t perplexity);

/* Run `steps` SA steps, each costing cfg.neighbours_per_step neighbours.  Entirely
 * device-resident; the call returns after the last kernel has completed. */
int mgl_sa_run(mgl_sa* sa, uint64_t steps, mgl_sa_stats* stats);
/* Current / best slab, n entries each, reference layout. */
int mgl_sa_current(mgl_sa* sa, mgl_packet* packets_out, uint64_t* perplexity_out);
int mgl_sa_best(mgl_sa* sa, mgl_packet* packets_out, uint64_t* perplexity_out);

/* Parity hooks (each leaves the SA state untouched). */
/* Cost a whole slab from byte 0.  per_packet_cumulative (nullable) receives one running
 * total per walked packet; npackets (nullable) their number. */
int mgl_cost_slab(mgl_sa* sa, const mgl_packet* packets, uint64_t* total,
                  uint64_t* per_packet_cumulative, size_t* npackets);
/* Final model state after costing `packets`: probabilities in the reference's struct order
 * (lzma_state.h:47-53: lit | len | rep_len | dist | ctx_state), ctx_state, rep distances. */
int mgl_final_state(mgl_sa* sa, const mgl_packet* packets, uint16_t* probs_out, size_t probs_cap,
                    uint8_t* ctx_state_out, uint32_t dists_out[4]);
/* Best cfg.top_k next packets at `position` (must be on the walk of `packets`), in the
 * reference's pop order: worst first, best last.  costs[i] = perplexity/length (integer). */
int mgl_top_k(mgl_sa* sa, const mgl_packet* packets, size_t position, mgl_packet* out,
              uint64_t* costs, size_t* count);
/* Match-index query at `pos`: (offset, length) pairs in the reference's callback order. */
int mgl_substrings(mgl_sa* sa, size_t pos, size_t max_len, uint32_t* offsets, uint32_t* lengths,
                   size_t cap, size_t* count);
/* Generate and cost the K neighbours the *next* mgl_sa_run step would look at (or those of
 * an explicit global step number), without deciding.  costs[j] = UINT64_MAX for a failed
 * generate.  diffs (nullable): diff_cap entries per neighbour, ndiffs[j] valid ones. */
int mgl_neighbours(mgl_sa* sa, uint64_t global_step, uint64_t* costs, mgl_diff* diffs,
                   uint32_t* ndiffs, size_t diff_cap);
/* Test / diagnostic hooks (no reference counterpart).  mgl_debug_dump: raw copy of one of the
 * incremental engine's device structures; *bytes receives the size even when the buffer is too
 * small.  Selectors: 0 chain offsets, 1 chain lengths, 2 chain positions, 3 chain events,
 * 4 on-walk bitmap, 5 special bitmap, 6 special-state records, 7 dense checkpoints, 8 chain
 * capacities, 9 phase-cycle counters (MGL_F_PROFILE), 10 per-step overflow / repair counters,
 * 11 parallel-builder totals, 12 / 13 match index (bucket offsets / positions), 14 accept-path
 * counters, 15 pick records, 16 the control block, 21 the windows (target, end) of the last costed neighbours, 22 their soft ends | dep << 31,
 * 30-35 / 40-45 / 50-55 / 60-65 positions / ranks / run starts / next byte of the exact-length orders D = 2..7,
 * 70-73 and 74-77 positions, ranks, run starts, next eight bytes of the 8- and 16-byte orders, 80 two u64 host counters: bulk
 * steps whose moves were patched into the base structures at once (batch accept), and those that began so and fell back to the rebuild.
 * mgl_debug_set: key 0 = stop the neighbour kernels after a phase (tools/phase_cost.py), 50 =
 * stage timing in the accept path; key 1 = make the parallel builder redo every chain segment
 * serially (exercises its fallback); key 2 = shrink the first-pass change lists (a multiple of 8,
 * at most the allocated size) so that neighbours overflow into the second pass (exercises it); key 3 = treat the
 * next so many bulk steps that took moves as failed validations (exercises the rollback); key 4 = 1 / 0: the
 * re-simulation kernel adds up the bytes it reads (mgl_sa_stats.sim_bytes_counted; a few percent slower); key 5 = n: the next
 * n batch accepts (bulk steps that patch few moves into the base structures) give up after they have written their journals
 * and bitmaps (exercises the fallback to the rebuild from there). */
int mgl_debug_dump(mgl_sa* sa, uint32_t what, void* out, size_t cap_bytes, size_t* bytes);
int mgl_debug_set(mgl_sa* sa, uint32_t key, uint64_t value);
/* draw n of neighbour j at global step `step` (31-bit, like rand()); j = 0xFFFFFFFF is the
 * step's own stream (accept decision). */
uint32_t mgl_rng_draw_at(uint64_t seed, uint64_t step, uint32_t j, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* MEGALANIA_HIP_H */
