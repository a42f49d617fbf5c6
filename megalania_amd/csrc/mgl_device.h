/*
 * mgl_device.h -- device-side building blocks shared by the kernels in mgl_kernels.hip.
 * gfx950 only: 64-lane wavefronts, probabilities and the bit-cost table in LDS.
 *
 * Conventions
 *   - one wavefront works on one slab walk.  Control flow is wave-uniform; walk state
 *     (position, ctx_state, rep distances) is kept in scalar registers via readfirstlane.
 *   - lane e of the wave evaluates event slot e of the current packet (mgl_model.h), so a
 *     packet costs one LDS read-modify-write round trip; per-lane u64 partial sums are
 *     reduced once at the end (u64 addition commutes, so the total is bit-exact).
 *   - the walk reads the slab and the input through a 64-position register window
 *     (one coalesced load per 64 bytes of input) and fetches entries with v_readlane.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mgl_model.h"

#define MGL_WAVE 64
#define MGL_CKPT_SHIFT 10u          /* one prefix checkpoint per 1024 input bytes */
#define MGL_MAX_DIFFS 64u           /* journal capacity per neighbour */
#define MGL_MAX_REPAIR_PICKS 8u /* top-K picks one neighbour's repair may need before it is given up and dropped (every one is a model
                                * reconstruction + a top-K query in the middle of the walk); the oracle counts the same picks */
#define MGL_MAX_WALK 2048u   /* neighbour packets the two-pointer walk visits (skipped literal runs do not count) before the neighbour is
                            * given up and dropped: rep distances that no match ever flushes (long runs coded as rep matches) keep two
                            * walks apart for kilobytes; the oracle counts the same packets */
#define MGL_MAX_TOPK 32u
#define MGL_PRICE_WORDS 1072u         /* top-K price tables per wavefront (u32): 2x272 lengths, 4x64 slots, 128 tails, 16 align, 64 slot bounds, 64 suffix minima */
#define MGL_SEQ_MASK ((1ull << 44) - 1ull)
#define MGL_INVALID_COST (~0ull)

struct DevCtx {
	const uint8_t* data;        /* n bytes (+64 bytes of zero padding) */
	uint32_t n;
	const uint32_t* bucket_off; /* 65537 */
	const uint32_t* bucket_pos; /* n-1 positions, ascending inside each bigram bucket */
	const uint16_t* bucket_nx;  /* per entry of bucket_pos: data[p + 2] | data[p + 3] << 8 */
	const uint32_t* quad_pos;   /* the same positions ordered by (data[p..p+3], p): inside a bigram bucket, runs of equal next-two-bytes */
	const uint16_t* quad_nx;    /* per entry of quad_pos: data[p + 2] << 8 | data[p + 3] (ascending inside a bucket) */
	/* deeper orders of the same positions (mgl_index.hip): sorted by the first 4 / 8 / 16 bytes, then by position.
	 * Per order: rank[p] = where position p sits; run[i] = first entry of the run of equal prefixes entry i belongs
	 * to (so the earlier positions that share p's prefix are exactly entries [run[rank[p]], rank[p])); and the
	 * bytes that follow the prefix, so that a hit's match length is known up to the next order's prefix without
	 * touching the input */
	/* the exact-length sources, D = 2..7 (index D - 2): positions ordered by their first D bytes then by position
	 * (D = 2: the bigram bucket itself, D = 4: quad_pos), rank / run as above (unused for D = 2: the bucket bounds
	 * play that part), and byte D of every entry: an entry whose byte D equals the target's belongs to the next
	 * source, all others match exactly D bytes */
	const uint32_t* xpos[6]; const uint32_t* xrank[6]; const uint32_t* xrun[6]; const uint8_t* xnxb[6];
	const uint32_t* oct_pos; const uint32_t* oct_rank; const uint32_t* oct_run; const uint64_t* oct_nx8;    /* bytes 8..15 */
	const uint32_t* hex_pos; const uint32_t* hex_rank; const uint32_t* hex_run; const uint64_t* hex_nx8;    /* bytes 16..23 */
	const uint16_t* cost_tbl;   /* 2048 x u16 */
	mgl_layout L;
	uint32_t dict_limit;
	uint32_t max_scan;
	uint32_t top_k;
	uint32_t diag_stop;         /* diagnostic: neighbour kernel returns after phase N (0 = run normally) */
	/* stratified targets (nullptr: MGL_F_POSITION_TARGETS, position draws): packets on the walk before every block of 4 096 positions
	 * (k_rank_blocks / k_rank_scan, once per step), so that neighbour j can take the packet of a given ordinal */
	const uint32_t* strat_pre;
	uint32_t strat_nblk;
	const uint32_t* strat_tgt;  /* K: the step's targets, made from strat_pre by k_targets before the neighbour kernels run */
};

struct CkptHdr {
	uint32_t pos, ctx_state;
	uint32_t dists[4];
	uint32_t ordinal, pad;
	uint64_t cum;
};

struct BaseView {
	mgl_pk* slab;          /* n packed entries, position-indexed */
	uint64_t* onwalk;      /* bitmap: bit (p&63) of word p>>6 set iff a packet starts at p */
	uint16_t* ckpt_probs;  /* nckpt x ckpt_elems */
	CkptHdr* ckpt_hdr;
	uint32_t nckpt;
	uint32_t ckpt_elems;   /* probabilities per checkpoint, rounded up to a multiple of 8 */
};

struct Control {
	uint64_t cur_cost, best_cost;
	uint64_t gstep, iter;
	uint64_t evals, failed, accepted, improved, packets_eval;
	uint64_t packets, rebuild_cost;
	uint32_t phase;
	uint32_t accepted_flag, copy_best_flag, dirty_pos, winner;
	uint32_t apply_failed; /* the incremental accept did not fit: rebuild from the slab */
	uint32_t best_is_current; /* the base structures are the best slab's: no device copy of them exists yet */
	uint32_t nbr_single;    /* the next step evaluates neighbours with the one-kernel form (repairs are frequent) */
	uint32_t mode_steps;    /* steps since the form last changed */
	uint32_t probing;       /* steps left of a short trial of the other form (keeps its timing fresh) */
	uint32_t ema_clean;     /* smoothed step duration, 10 ns ticks: split form, second pass empty ... */
	uint32_t ema_dirty;     /* ... split form, second pass taken ... */
	uint32_t ema_single;    /* ... one-kernel form */
	uint32_t p_dirty;       /* smoothed probability (x 65536) that a step needs a repair pick / second pass */
	unsigned long long t_last; /* wall clock at the end of the previous step */
	uint32_t full_rebuilds;  /* accepts that went through k_build */
	uint64_t fallback_nbrs;  /* neighbours costed by the full-walk kernel (did not fit the LDS lists) */
	uint64_t big_nbrs;       /* neighbours redone by the second (global-scratch) pass */
	uint32_t final_ctx_state;
	uint32_t final_dists[4];
	uint32_t error_flags;
	/* batched decision (mgl_kernels4.hip) */
	uint64_t imp_cands;      /* neighbours that cost less than the current slab, summed over steps */
	uint64_t dropped;        /* neighbours dropped because their journal outgrew MGL_MAX_DIFFS */
	uint64_t bulk_steps;     /* steps that took every window-best acceptable neighbour */
	uint32_t taken;          /* neighbours the last step took */
	uint32_t bulk_was_best;  /* bulk step: the base held the best slab's structures when the step began */
	uint32_t bulk_need_undo; /* bulk step: the best slab has to be restored from the new one + the undo log */
	uint32_t la_lo, la_end;  /* window of the move the last step accepted (la_lo = MGL_POS_INF: none) */
	uint32_t mod_lo, mod_hi; /* the last accepted move changed some context's probability before positions in (mod_lo, mod_hi] (MGL_POS_INF: to the end) */
	uint64_t bulk_overlaps;  /* slab entries two taken journals of one bulk step both wanted to change (k_bulk_round; such a step is taken back) */
};
#define MGL_ERR_REBUILD_MISMATCH 1u
#define MGL_ERR_WALK_OVERRUN 2u
#define MGL_ERR_BAD_PACKET 4u /* a packet that does not reproduce the input (k_validate) */

#define MGL_WIN_NONE 0xFFFFFFFFu    /* no cost: no candidate at the target / handed to a later pass */
#define MGL_WIN_DROPPED 0xFFFFFFFEu /* no cost: the journal outgrew MGL_MAX_DIFFS */
struct NbrOut {
	uint64_t* cost;    /* K */
	uint32_t* ndiffs;  /* K */
	uint32_t* walked;  /* K: packets costed by the neighbour (from its checkpoint on) */
	uint32_t* win;     /* 2 K: target position; first position from which neighbour and base are coded
	                    * identically again (n if never), or MGL_WIN_NONE / MGL_WIN_DROPPED with an invalid cost */
	uint32_t* win2;    /* K: soft end of the window (first meeting point inside the base's rep-free tail, else the end)
	                    * | bit 31: a rep packet inside the window reads a rep distance from before it */
	uint32_t* dpos;    /* K x MGL_MAX_DIFFS */
	mgl_pk* dold;
	mgl_pk* dnew;
};

__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint32_t rdlane(uint32_t x, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)l); }
__device__ __forceinline__ uint64_t rdlane64(uint64_t x, uint32_t l)
{
	return (uint64_t)rdlane((uint32_t)x, l) | ((uint64_t)rdlane((uint32_t)(x >> 32), l) << 32);
}
__device__ __forceinline__ uint64_t uni64(uint64_t x)
{
	return (uint64_t)uni((uint32_t)x) | ((uint64_t)uni((uint32_t)(x >> 32)) << 32);
}
__device__ __forceinline__ uint64_t shfl64(uint64_t x, int src)
{
	uint32_t lo = (uint32_t)__shfl((int)(uint32_t)x, src, 64);
	uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(x >> 32), src, 64);
	return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ uint64_t shfl_up64(uint64_t x, int delta)
{
	uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, delta, 64);
	uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), delta, 64);
	return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v)
{
	for (int o = 32; o > 0; o >>= 1) {
		uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
		uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64);
		v += (uint64_t)lo | ((uint64_t)hi << 32);
	}
	return v;
}
/* Stratified target: neighbour j of a step of K takes a packet of the j-th of K equal slices of the
 * walk's P packets (by ordinal), uniformly inside the slice (draw 0 of its stream).  Every packet is as likely a target as
 * under independent uniform draws (packet_slab_neighbour.c:162-163 draws an ordinal too), but the K targets of a step are
 * distinct and spread over the file, so that far fewer improving neighbours of one step overlap.  `pre[b]` = packets that
 * start before block b of 4 096 positions (k_rank_blocks / k_rank_scan, once per step).  Returns the target position. */
__device__ __forceinline__ uint32_t stratified_target(const uint64_t* onwalk, uint32_t nw0, const uint32_t* pre, uint32_t nblk,
                                                      uint32_t P, uint32_t K, uint32_t j, uint32_t u, uint32_t lane)
{
	const uint32_t lo_o = (uint32_t)((uint64_t)j * P / K), hi_o = (uint32_t)((uint64_t)(j + 1u) * P / K);
	uint32_t ord = lo_o + (hi_o > lo_o ? u % (hi_o - lo_o) : 0u);
	if (ord >= P) ord = P ? P - 1u : 0u;
	uint32_t lo = 0, hi = nblk; /* pre[lo] <= ord < pre[hi] */
	while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= ord) lo = mid; else hi = mid; }
	const uint32_t rem = ord - pre[lo];
	const uint32_t w = lo * 64u + lane;
	const uint64_t word = w < nw0 ? onwalk[w] : 0ull;
	const uint32_t cnt = (uint32_t)__popcll(word);
	uint32_t incl = cnt;
	for (int o = 1; o < 64; o <<= 1) { const uint32_t t2 = (uint32_t)__shfl_up((int)incl, o, 64); if ((int)lane >= o) incl += t2; }
	const unsigned long long has = __ballot(incl > rem);
	uint32_t tpos = 0;
	if (has) {
		const uint32_t l = (uint32_t)__ffsll((long long)has) - 1u;
		const uint32_t before = rdlane(incl - cnt, l);
		uint64_t wv = rdlane64(word, l);
		for (uint32_t q = rem - before; q > 0; q--) wv &= wv - 1ull;
		tpos = ((lo * 64u + l) << 6) + (uint32_t)__ffsll((long long)wv) - 1u;
	}
	return uni(tpos);
}

/* LDS written by some lanes of this wave is read by others: keep the compiler from moving
 * accesses across (the hardware executes a wave's LDS operations in order). */
__device__ __forceinline__ void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

/* ------------------------------------------------------------------ the slab walk */
struct Walk {
	mgl_wstate st;   /* uniform */
	uint64_t acc;    /* per-lane partial perplexity */
	uint32_t wbase;  /* window: positions [wbase, wbase+64) */
	mgl_pk wpk;
	uint32_t wbyte;
	uint32_t packets;
};

__device__ __forceinline__ void walk_reset(Walk& w)
{
	w.st.pos = 0; w.st.ctx_state = 0;
	w.st.dists[0] = w.st.dists[1] = w.st.dists[2] = w.st.dists[3] = 0;
	w.acc = 0; w.wbase = 0xFFFFFFFFu; w.wpk = 0; w.wbyte = 0; w.packets = 0;
}

/* make the window cover st.pos */
__device__ __forceinline__ void walk_window(Walk& w, const DevCtx& c, const mgl_pk* slab, uint32_t lane)
{
	uint32_t base = w.st.pos & ~63u;
	if (base != w.wbase) {
		uint32_t p = base + lane;
		w.wpk = p < c.n ? slab[p] : 0;
		w.wbyte = c.data[p < c.n ? p : c.n]; /* data has >= 64 bytes of zero padding */
		w.wbase = base;
	}
}
__device__ __forceinline__ mgl_pk walk_slab_at(const Walk& w, uint32_t pos) { return rdlane64(w.wpk, pos - w.wbase); }
__device__ __forceinline__ uint32_t walk_byte_at(const Walk& w, uint32_t pos) { return rdlane(w.wbyte, pos - w.wbase); }

/* Cost one packet at the walk's position against the adaptive model in LDS and advance.
 * UPDATE=false leaves probabilities untouched (candidate costing) and does not advance. */
template <bool UPDATE>
__device__ __forceinline__ void walk_packet(Walk& w, const DevCtx& c, uint16_t* probs, const uint16_t* T,
                                            uint32_t type, uint32_t dist, uint32_t len, uint32_t lane)
{
	uint32_t pos = w.st.pos;
	uint32_t byte = walk_byte_at(w, pos);
	uint32_t match_byte = 0, prev_byte = 0;
	if (type == MGL_LITERAL) {
		if (w.st.ctx_state >= 7 && w.st.dists[0] < pos) match_byte = c.data[pos - w.st.dists[0] - 1];
		if (c.L.lc > 0 && pos > 0) prev_byte = c.data[pos - 1];
	}
	mgl_plan pl;
	mgl_plan_packet(&c.L, &w.st, type, dist, len, byte, match_byte, prev_byte, &pl);
	if (lane < pl.nev) {
		uint32_t ctx, bit;
		mgl_plan_event(&pl, lane, &ctx, &bit);
		uint32_t p = probs[ctx];
		w.acc += T[bit ? 2048u - p : p];
		if (UPDATE) probs[ctx] = (uint16_t)mgl_prob_update(p, bit);
	}
	if (lane == 0) w.acc += (uint64_t)pl.ndirect << 11;
	if (UPDATE) {
		mgl_advance(&w.st, type, dist, len);
		w.packets++;
	}
}
