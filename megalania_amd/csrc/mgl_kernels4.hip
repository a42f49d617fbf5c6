/*
 * mgl_kernels4.hip -- the step's decision: which of the K costed neighbours
 * the chain moves to.
 *
 * Every evaluation keeps the reference's own rule (main.c:86-87): neighbour j of a step is the
 * epoch's iteration i = Control::iter + j and is *acceptable* when it costs less than the current
 * slab or when its transition draw  draw % (i*i + 1 + phase*N/2) < sqrt(N)  says so (opt-in, not in the
 * reference: with a temperature, when u < exp(-delta / t_eff) instead, in integers through the
 * reference's log table).  Acceptable neighbours are ranked by key = (improving ? 0 : 1, cost, j).
 *
 *   k_decide        single mode: the acceptable neighbour with the smallest key wins; the accept
 *                   path (mgl_kernels3.hip) folds it into the base structures incrementally.
 *   k_bulk_prep / k_bulk_round / k_bulk_end / k_bulk_finish / k_bulk_keep_*
 *                   bulk mode: every acceptable neighbour whose window [target, end) overlaps no
 *                   acceptable neighbour of smaller key is taken in the same step.  Windows are
 *                   where a neighbour's walk differs from the base's, so disjoint windows give a
 *                   valid parse; the journals go into the slab, the base structures are re-derived
 *                   by the parallel builder (mgl_pbuild.hip), and the new total is its exact cost.
 *
 * Both are mirrored by oracle/mgl_oracle.c:orc_sa_batched, step by step, bit for bit.
 */
#include "mgl_device.h"

struct DecideArgs {
	uint32_t K;
	uint64_t seed;
	uint64_t iters_per_epoch;
	uint64_t sqrt_thresh;
	uint64_t temperature;
};

/* one neighbour's verdict; key = ~0 when it is not acceptable */
struct Verdict {
	uint64_t key;
	uint32_t valid, improving, dropped, walked;
};
__device__ __forceinline__ Verdict judge(const DevCtx& c, const Control* ctl, const NbrOut& out, const DecideArgs& a, uint32_t j,
                                         uint64_t base_cost, uint64_t gstep, uint64_t iter)
{
	Verdict v;
	v.key = MGL_INVALID_COST; v.valid = 0; v.improving = 0; v.dropped = 0; v.walked = 0;
	const uint64_t cst = out.cost[j];
	const uint32_t wk = out.walked[j], we = out.win[2u * j + 1u];
	if (cst == MGL_INVALID_COST) { v.dropped = we == MGL_WIN_DROPPED ? 1u : 0u; return v; }
	v.valid = 1; v.walked = wk;
	bool ok = cst < base_cost;
	v.improving = ok ? 1u : 0u;
	if (!ok) {
		uint64_t i = iter + j;
		if (i > 0x7FFFFFFFull) i = 0x7FFFFFFFull;
		const uint32_t draw = mgl_rng_draw(mgl_rng_key(a.seed, gstep, 0xFFFFFFFFu), 2u + j);
		if (a.temperature) {
			const uint32_t u = draw % 2047u + 1u;
			const uint64_t ic = i < a.iters_per_epoch ? i : a.iters_per_epoch;
			const uint64_t t_eff = a.temperature * (a.iters_per_epoch - ic) / a.iters_per_epoch;
			ok = (cst - base_cost) * 2048u <= t_eff * (uint64_t)c.cost_tbl[u];
		} else {
			const uint64_t m = i * i + 1ull + (uint64_t)ctl->phase * a.iters_per_epoch / 2ull;
			ok = (m > 0x7FFFFFFFull ? (uint64_t)draw : (uint64_t)draw % m) < a.sqrt_thresh; /* (draws are 31 bits: beyond i = 46 341 the division changes nothing) */
		}
	}
	if (ok) v.key = ((v.improving ? 0ull : 1ull) << 63) | (cst << 20) | j;
	return v;
}

/* ================================================================== k_decide (single mode) */
__global__ void __launch_bounds__(1024) k_decide(DevCtx c, BaseView b, Control* ctl, NbrOut out, DecideArgs a, int apply_journal)
{
	__shared__ uint64_t s_key[16];
	__shared__ uint64_t s_cnt[16 * 4];
	__shared__ uint32_t s_winner;
	const uint32_t tid = threadIdx.x;
	const uint64_t gstep = ctl->gstep, iter = ctl->iter;
	const uint64_t base_cost = ctl->cur_cost ? ctl->cur_cost : ctl->rebuild_cost;
	uint64_t best = MGL_INVALID_COST, valid = 0, walked = 0, imp = 0, dropped = 0;
	for (uint32_t j = tid; j < a.K; j += blockDim.x) {
		const Verdict v = judge(c, ctl, out, a, j, base_cost, gstep, iter);
		valid += v.valid; walked += v.walked; imp += v.improving; dropped += v.dropped;
		best = v.key < best ? v.key : best;
	}
	for (int o = 32; o > 0; o >>= 1) {
		const uint64_t ob = (uint64_t)__shfl_xor((unsigned long long)best, o, 64);
		best = ob < best ? ob : best;
		valid += (uint64_t)__shfl_xor((unsigned long long)valid, o, 64);
		walked += (uint64_t)__shfl_xor((unsigned long long)walked, o, 64);
		imp += (uint64_t)__shfl_xor((unsigned long long)imp, o, 64);
		dropped += (uint64_t)__shfl_xor((unsigned long long)dropped, o, 64);
	}
	if ((tid & 63u) == 0) {
		s_key[tid >> 6] = best;
		s_cnt[tid >> 6] = valid; s_cnt[16 + (tid >> 6)] = walked; s_cnt[32 + (tid >> 6)] = imp; s_cnt[48 + (tid >> 6)] = dropped;
	}
	__syncthreads();
	if (tid == 0) {
		for (uint32_t w = 1; w < (blockDim.x >> 6); w++) {
			s_key[0] = s_key[w] < s_key[0] ? s_key[w] : s_key[0];
			s_cnt[0] += s_cnt[w]; s_cnt[16] += s_cnt[16 + w]; s_cnt[32] += s_cnt[32 + w]; s_cnt[48] += s_cnt[48 + w];
		}
		const uint64_t bkey = s_key[0];
		const uint32_t winner = bkey == MGL_INVALID_COST ? ~0u : (uint32_t)(bkey & 0xFFFFFu);
		ctl->evals += s_cnt[0];
		ctl->failed += a.K - s_cnt[0];
		ctl->packets_eval += s_cnt[16];
		ctl->imp_cands += s_cnt[32];
		ctl->dropped += s_cnt[48];
		ctl->gstep = gstep + 1;
		ctl->iter = iter + a.K; /* the reference's i counts evaluations (main.c:78) */
		ctl->winner = winner;
		ctl->accepted_flag = winner != ~0u;
		ctl->copy_best_flag = 0;
		ctl->taken = winner != ~0u ? 1u : 0u;
		ctl->la_lo = winner != ~0u ? out.win[2u * winner] : MGL_POS_INF; /* for the look-ahead's check of the next step */
		ctl->la_end = winner != ~0u ? out.win[2u * winner + 1u] : 0u;
		ctl->cur_cost = base_cost; /* an epoch's current cost starts as the exact cost of its first slab */
		if (winner != ~0u) {
			ctl->accepted++;
			ctl->cur_cost = out.cost[winner];
			ctl->dirty_pos = out.dpos[(size_t)winner * MGL_MAX_DIFFS];
			if (ctl->best_cost == 0 || ctl->cur_cost < ctl->best_cost) {
				ctl->best_cost = ctl->cur_cost;
				ctl->copy_best_flag = 1;
				ctl->improved++;
			}
		}
		s_winner = winner;
	}
	__syncthreads();
	const uint32_t winner = s_winner;
	if (winner == ~0u || !apply_journal) return; /* incremental engine: k_apply_walk writes the journal */
	const uint32_t nd = out.ndiffs[winner];
	if (tid < nd) b.slab[out.dpos[(size_t)winner * MGL_MAX_DIFFS + tid]] = out.dnew[(size_t)winner * MGL_MAX_DIFFS + tid];
}

/* ================================================================== bulk mode */
struct BulkBuf {
	uint64_t* ckey;     /* K: keys of the acceptable neighbours (compacted, any order) */
	uint4* cwin;        /* K: their windows: target, end, soft end, dep */
	uint32_t* taken;    /* K: neighbour indices taken this step */
	uint8_t* cstate;    /* 2 x K: selection state per candidate (0 undecided, 1 taken, 2 rejected), one array per round parity */
	uint32_t* cflags;   /* K: what k_bulk_pairs found for each candidate in the current round (zero between rounds) */
	unsigned long long* hdr; /* [0] acceptable [1] taken [2] valid [3] walked [4] improving [5] dropped [6] smallest taken key */
};

__global__ void __launch_bounds__(256) k_bulk_prep(DevCtx c, Control* ctl, NbrOut out, DecideArgs a, BulkBuf bb)
{
	__shared__ unsigned long long s_cnt[4];
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
	if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	const uint64_t base_cost = ctl->cur_cost ? ctl->cur_cost : ctl->rebuild_cost;
	Verdict v;
	v.key = MGL_INVALID_COST; v.valid = v.improving = v.dropped = v.walked = 0;
	if (j < a.K) v = judge(c, ctl, out, a, j, base_cost, ctl->gstep, ctl->iter);
	uint64_t valid = v.valid, walked = v.walked, imp = v.improving, dropped = v.dropped;
	for (int o = 32; o > 0; o >>= 1) {
		valid += (uint64_t)__shfl_xor((unsigned long long)valid, o, 64);
		walked += (uint64_t)__shfl_xor((unsigned long long)walked, o, 64);
		imp += (uint64_t)__shfl_xor((unsigned long long)imp, o, 64);
		dropped += (uint64_t)__shfl_xor((unsigned long long)dropped, o, 64);
	}
	if (lane == 0) { atomicAdd(&s_cnt[0], valid); atomicAdd(&s_cnt[1], walked); atomicAdd(&s_cnt[2], imp); atomicAdd(&s_cnt[3], dropped); }
	/* compact the acceptable ones: one atomic per wavefront */
	const bool acc = v.key != MGL_INVALID_COST;
	const unsigned long long m = __ballot(acc);
	if (m) {
		unsigned long long base = 0;
		if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(&bb.hdr[0], (unsigned long long)__popcll(m));
		base = (unsigned long long)shfl64(base, __ffsll((long long)m) - 1);
		if (acc) {
			const uint32_t at = (uint32_t)base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
			bb.ckey[at] = v.key;
			const uint32_t w2 = out.win2[j];
			bb.cwin[at] = make_uint4(out.win[2u * j], out.win[2u * j + 1u], w2 & 0x7FFFFFFFu, w2 >> 31);
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		atomicAdd(&bb.hdr[2], s_cnt[0]); atomicAdd(&bb.hdr[3], s_cnt[1]); atomicAdd(&bb.hdr[4], s_cnt[2]); atomicAdd(&bb.hdr[5], s_cnt[3]);
	}
}

/* Two neighbours cannot both be taken -- with A the one that starts first -- unless B starts at or after A's soft
 * end (from there on A's walk is the base's, packet for packet, and nothing reads the rep distances that may still
 * differ) and either B is self-contained (no rep packet of B reads a distance from before its window) or B starts
 * at or after A's end (where the rep distances agree again).  windows = (target, end, soft end, dep). */
__device__ __forceinline__ bool windows_conflict(const uint4& x, const uint4& y)
{
	const bool xf = x.x <= y.x;
	const uint4 a = xf ? x : y, b = xf ? y : x;
	if (a.x == b.x) return true;
	return !(a.z <= b.x && (b.w == 0u || a.y <= b.x));
}
/* Selection = the greedy independent set in key order (a candidate is taken iff no TAKEN candidate of smaller key
 * conflicts with it), computed in MGL_BULK_ROUNDS synchronous rounds: in a round an undecided candidate is rejected
 * if a taken smaller-key candidate conflicts with it, taken if every conflicting smaller-key candidate is already
 * rejected, and stays undecided otherwise; states are read from the previous round's array and written to the next
 * (so the result does not depend on scheduling), and what is still undecided after the last round is rejected.
 * The taken ones write their journals into the slab at once (taken windows are pairwise compatible, so no two of
 * them touch one entry). */
/* one round, first half: workgroup (x, y) holds the candidates of tile x against the candidates of tile y (256 each) and
 * ORs what it finds into the candidates' flags -- bit 0: a taken smaller-key candidate conflicts, bit 1: an undecided one
 * does.  (One workgroup per candidate tile walking all tiles took 3.5 ms per step with 5 000 candidates on 20 CUs.) */
__global__ void __launch_bounds__(256) k_bulk_pairs(BulkBuf bb, uint32_t K, uint32_t round)
{
	__shared__ uint64_t s_key[256];
	__shared__ uint4 s_win[256];
	__shared__ uint8_t s_st[256];
	const uint32_t n = (uint32_t)bb.hdr[0];
	if (blockIdx.x * 256u >= n || blockIdx.y * 256u >= n) return;
	const uint8_t* st_in = bb.cstate + (size_t)(round & 1u) * K;
	const uint32_t a = blockIdx.x * 256u + threadIdx.x;
	const bool mine = a < n;
	const uint8_t my = (mine && round) ? st_in[a] : (uint8_t)0;
	if (!__syncthreads_or(mine && my == 0)) return;
	const uint32_t t0 = blockIdx.y * 256u;
	if (t0 + threadIdx.x < n) {
		s_key[threadIdx.x] = bb.ckey[t0 + threadIdx.x]; s_win[threadIdx.x] = bb.cwin[t0 + threadIdx.x];
		s_st[threadIdx.x] = round ? st_in[t0 + threadIdx.x] : (uint8_t)0;
	}
	__syncthreads();
	if (!mine || my != 0) return;
	const uint64_t key = bb.ckey[a];
	const uint4 w = bb.cwin[a];
	const uint32_t cnt = (n - t0) < 256u ? (n - t0) : 256u;
	uint32_t f = 0;
	for (uint32_t i = 0; i < cnt; i++)
		if (s_st[i] != 2 && s_key[i] < key && windows_conflict(s_win[i], w)) f |= s_st[i] == 1 ? 1u : 2u;
	if (f) atomicOr(&bb.cflags[a], f);
}
/* second half: the verdicts of the round */
__global__ void __launch_bounds__(256) k_bulk_round(Control* ctl, NbrOut out, BulkBuf bb, mgl_pk* slab, uint32_t K, uint32_t round)
{
	const uint32_t n = (uint32_t)bb.hdr[0];
	const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= n) return;
	const uint8_t* st_in = bb.cstate + (size_t)(round & 1u) * K;
	uint8_t* st_out = bb.cstate + (size_t)((round + 1u) & 1u) * K;
	const uint8_t my = round ? st_in[a] : (uint8_t)0;
	const uint32_t f = bb.cflags[a];
	bb.cflags[a] = 0;
	const uint8_t now = my ? my : ((f & 1u) ? (uint8_t)2 : ((f & 2u) ? (uint8_t)0 : (uint8_t)1));
	st_out[a] = now;
	if (my != 0 || now != 1) return;
	const uint64_t key = bb.ckey[a];
	const uint32_t j = (uint32_t)(key & 0xFFFFFu);
	const unsigned long long at = atomicAdd(&bb.hdr[1], 1ull);
	bb.taken[at] = j;
	atomicMin(&bb.hdr[6], (unsigned long long)key);
	(void)ctl; (void)out; (void)slab;
}
/* The journals of the taken neighbours into the slab, for the rebuild (a batch accept writes them itself, mgl_kernels5.hip).
 * They are written in parallel and must touch disjoint entries.  The window rule is meant to guarantee that; it is checked
 * here, entry by entry, instead of trusted: an entry is only replaced while it still holds the value this neighbour was
 * evaluated against (64-bit compare-and-swap).  One that another taken journal has changed already -- to a different
 * packet -- would make the result depend on which of the two writes last: the step is flagged like an invalid parse
 * (MGL_ERR_BAD_PACKET), taken back as a whole (k_bulk_rollback restores the old values, the same for both writers) and
 * counted (Control::bulk_overlaps, mgl_sa_stats.bulk_double_writes; the oracle counts the same thing). */
__global__ void __launch_bounds__(256) k_bulk_write(Control* ctl, NbrOut out, BulkBuf bb, mgl_pk* slab)
{
	const uint32_t taken = (uint32_t)bb.hdr[1];
	for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < taken; t += gridDim.x * blockDim.x) {
		const uint32_t j = bb.taken[t], nd = out.ndiffs[j];
		for (uint32_t e = 0; e < nd; e++) {
			const size_t k = (size_t)j * MGL_MAX_DIFFS + e;
			const mgl_pk was = out.dold[k], neu = out.dnew[k];
			const mgl_pk seen = (mgl_pk)atomicCAS((unsigned long long*)&slab[out.dpos[k]], (unsigned long long)was, (unsigned long long)neu);
			if (seen != was && seen != neu) {
				atomicOr(&ctl->error_flags, MGL_ERR_BAD_PACKET);
				atomicAdd((unsigned long long*)&ctl->bulk_overlaps, 1ull);
			}
		}
	}
}

/* The same selection -- the same MGL_BULK_ROUNDS synchronous rounds, the same verdicts -- by one workgroup in one launch, for
 * steps with few acceptable neighbours (the sixteen launches of the tiled form cost 75 us whatever they find to do; the host
 * picks this form when the previous step had at most 512 candidates).  Any number of candidates is handled (in tiles of
 * 1024 through LDS), only slowly beyond a few thousand. */
__global__ void __launch_bounds__(1024) k_bulk_select_small(Control* ctl, NbrOut out, BulkBuf bb, uint32_t K)
{
	__shared__ uint64_t s_key[1024];
	__shared__ uint4 s_win[1024];
	__shared__ uint8_t s_st[1024];
	const uint32_t n = (uint32_t)bb.hdr[0];
	const uint32_t tid = threadIdx.x;
	(void)ctl; (void)out;
	/* states live in cstate[0 .. K) (previous round) and cstate[K .. 2K) (next round), as in the tiled form */
	for (uint32_t a = tid; a < n; a += blockDim.x) bb.cstate[a] = 0;
	__syncthreads();
	for (uint32_t round = 0; round < MGL_BULK_ROUNDS; round++) {
		const uint8_t* st_in = bb.cstate + (size_t)(round & 1u) * K;
		uint8_t* st_out = bb.cstate + (size_t)((round + 1u) & 1u) * K;
		for (uint32_t a0 = 0; a0 < n; a0 += blockDim.x) {
			const uint32_t a = a0 + tid;
			const bool mine = a < n;
			const uint8_t my = mine ? st_in[a] : (uint8_t)2;
			uint64_t key = 0; uint4 w = make_uint4(0, 0, 0, 0);
			if (mine) { key = bb.ckey[a]; w = bb.cwin[a]; }
			uint32_t f = 0;
			for (uint32_t t0 = 0; t0 < n; t0 += 1024u) {
				__syncthreads();
				if (t0 + tid < n) { s_key[tid] = bb.ckey[t0 + tid]; s_win[tid] = bb.cwin[t0 + tid]; s_st[tid] = st_in[t0 + tid]; }
				__syncthreads();
				if (mine && my == 0) {
					const uint32_t cnt = (n - t0) < 1024u ? (n - t0) : 1024u;
					for (uint32_t i = 0; i < cnt; i++)
						if (s_st[i] != 2 && s_key[i] < key && windows_conflict(s_win[i], w)) f |= s_st[i] == 1 ? 1u : 2u;
				}
			}
			if (mine) {
				const uint8_t now = my ? my : ((f & 1u) ? (uint8_t)2 : ((f & 2u) ? (uint8_t)0 : (uint8_t)1));
				st_out[a] = now;
				if (my == 0 && now == 1) {
					const unsigned long long at = atomicAdd(&bb.hdr[1], 1ull);
					bb.taken[at] = (uint32_t)(key & 0xFFFFFu);
					atomicMin(&bb.hdr[6], (unsigned long long)key);
				}
			}
		}
		__syncthreads();
		__threadfence_block();
	}
}

/* bookkeeping between the selection and the rebuild */
__global__ void k_bulk_end(Control* ctl, BulkBuf bb, DecideArgs a)
{
	if (threadIdx.x || blockIdx.x) return;
	const uint64_t base_cost = ctl->cur_cost ? ctl->cur_cost : ctl->rebuild_cost;
	const uint32_t taken = (uint32_t)bb.hdr[1];
	ctl->evals += bb.hdr[2];
	ctl->failed += a.K - bb.hdr[2];
	ctl->packets_eval += bb.hdr[3];
	ctl->imp_cands += bb.hdr[4];
	ctl->dropped += bb.hdr[5];
	ctl->gstep += 1;
	ctl->iter += a.K;
	ctl->accepted += taken;
	ctl->taken = taken;
	ctl->bulk_steps += 1;
	ctl->cur_cost = base_cost;
	ctl->winner = ~0u;
	ctl->accepted_flag = 0; ctl->copy_best_flag = 0; ctl->apply_failed = 0;
	ctl->bulk_was_best = ctl->best_is_current;
	ctl->bulk_need_undo = 0;
}

/* after the parallel builder: the new slab's exact cost is the current cost; best-slab tracking
 * (main.c:88-92).  lazy_best: the best slab's structures live in the base while best_is_current;
 * when a bulk step leaves the best slab (rare: it took moves and their sum did not improve) the
 * best slab itself is restored from the new one and the undo log (k_bulk_keep_*), and its
 * structures are re-derived when an epoch next starts from it. */
/* The kernels that close a bulk step take a gate: the batch accept's status word (mgl_kernels5.hip).  They are queued
 * behind the batch accept before the host has read that word, so that they run while it does: they go ahead when the
 * moves are in (3) or the step took none (0); a step that goes to the rebuild instead queues them again, ungated. */
__device__ __forceinline__ bool gate_open(const uint32_t* gate) { return gate == nullptr || gate[0] == 3u || gate[0] == 0u; }
__global__ void k_bulk_finish(Control* ctl, BulkBuf bb, int lazy_best, uint32_t* snap_best_valid, const uint32_t* gate)
{
	if (threadIdx.x || blockIdx.x || !gate_open(gate)) return;
	const uint32_t taken = ctl->taken;
	if (taken) {
		ctl->cur_cost = ctl->rebuild_cost;
		if (ctl->best_cost == 0 || ctl->cur_cost < ctl->best_cost) {
			ctl->best_cost = ctl->cur_cost;
			ctl->copy_best_flag = 1;
			ctl->improved++;
		}
		if (lazy_best) {
			if (ctl->copy_best_flag) ctl->best_is_current = 1;
			else if (ctl->bulk_was_best) {
				ctl->best_is_current = 0;
				ctl->bulk_need_undo = 1;
				if (snap_best_valid) *snap_best_valid = 0u; /* SnapMeta::valid of the best snapshot */
			}
		}
	}
	ctl->t_last = 0; /* a bulk step is not a sample for the split / one-kernel choice */
	bb.hdr[0] = 0; bb.hdr[1] = taken; /* k_bulk_keep_undo still needs the count; cleared there */
	bb.hdr[2] = bb.hdr[3] = bb.hdr[4] = bb.hdr[5] = 0; bb.hdr[6] = ~0ull;
}
/* best slab := the slab before this step = the new slab ... */
__global__ void __launch_bounds__(256) k_bulk_keep_copy(const Control* ctl, const mgl_pk* slab, mgl_pk* best, uint32_t n, const uint32_t* gate)
{
	if (!gate_open(gate) || !ctl->bulk_need_undo) return;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) best[i] = slab[i];
}
/* ... with the taken journals undone */
__global__ void __launch_bounds__(256) k_bulk_keep_undo(const Control* ctl, NbrOut out, BulkBuf bb, mgl_pk* best, const uint32_t* gate)
{
	if (!gate_open(gate)) return;
	const uint32_t taken = (uint32_t)bb.hdr[1];
	if (ctl->bulk_need_undo) {
		for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < taken; t += gridDim.x * blockDim.x) {
			const uint32_t j = bb.taken[t], nd = out.ndiffs[j];
			for (uint32_t e = 0; e < nd; e++) best[out.dpos[(size_t)j * MGL_MAX_DIFFS + e]] = out.dold[(size_t)j * MGL_MAX_DIFFS + e];
		}
	}
}
/* the safety net of the soft window ends: the combined parse did not validate (k_validate after the rebuild): every
 * taken journal is taken back */
__global__ void __launch_bounds__(256) k_bulk_rollback(NbrOut out, BulkBuf bb, mgl_pk* slab)
{
	const uint32_t taken = (uint32_t)bb.hdr[1];
	for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < taken; t += gridDim.x * blockDim.x) {
		const uint32_t j = bb.taken[t], nd = out.ndiffs[j];
		for (uint32_t e = 0; e < nd; e++) slab[out.dpos[(size_t)j * MGL_MAX_DIFFS + e]] = out.dold[(size_t)j * MGL_MAX_DIFFS + e];
	}
}
__global__ void k_bulk_reset(BulkBuf bb, const uint32_t* gate)
{
	if (threadIdx.x == 0 && blockIdx.x == 0 && gate_open(gate)) bb.hdr[1] = 0;
}
/* the slab copy of a new best, for handles without snapshots (k_copy_best behind the gate) */
__global__ void __launch_bounds__(256) k_bulk_copy_best(const Control* ctl, const mgl_pk* slab, mgl_pk* best, uint32_t n, const uint32_t* gate)
{
	if (!gate_open(gate) || !ctl->copy_best_flag) return;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) best[i] = slab[i];
}
__global__ void k_bulk_step_end(Control* ctl, uint32_t* counts, int form_single, const uint32_t* gate)
{
	if (gate_open(gate)) step_end_body(ctl, 0, counts, 0, form_single);
}

/* ================================================================== look-ahead
 *
 * While a step's re-simulations, second pass, decision and accept run (few wavefronts, one long dependency chain), the
 * first two thirds of the NEXT step's evaluation -- pick and window walk, which do not price anything -- run beside
 * them on the base as it is before the accept.  After the accept k_la_check keeps every such result that the accepted
 * move cannot have touched and lists the others for a fresh evaluation:
 *   - the target draw is repeated on the new on-walk bitmap (same draws; a draw inside the accepted window may land
 *     differently now);
 *   - a neighbour wholly in front of the accepted window (its walk met the base again at or before the window's first
 *     position) saw only unchanged packets, and the model at its target is the old one;
 *   - a neighbour whose target lies at or behind the window's end starts from the same walk state (that is what the end
 *     means) and reads unchanged packets; its pick is kept iff no context's probability before its target changed:
 *     target outside (mod_lo, mod_hi], the span k_apply_chains reports (first change .. last re-coupling over all contexts);
 *   - anything else is evaluated again.
 * What is kept is bit for bit what a launch after the accept would have produced: the walk and the change lists hold
 * positions, packets and contexts only; costs come from k_sim, which runs after the accept either way. */
__global__ void __launch_bounds__(256) k_la_check(DevCtx c, Base2 b, const Control* ctl, NbrOut out, const uint4* pickstate, uint4* sim_hdr,
                                                  const uint32_t* todo_count, uint8_t* mark, uint32_t* list, uint32_t* hdr, uint64_t seed, uint32_t K)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
	if (j == 0) hdr[1] = *todo_count; /* second-pass entries made by the speculative launch: [0, this) */
	bool redo = false;
	const uint32_t a_lo = ctl->la_lo, a_end = ctl->la_end;
	if (j < K && a_lo != MGL_POS_INF) {
		const uint4 s0 = pickstate[2u * j];
		const uint32_t target = s0.x, rn = s0.y;
		/* the draw of mgl_kernels2.hip:nbr2_one, one thread instead of 32 lanes */
		const uint64_t key = mgl_rng_key(seed, ctl->gstep, j);
		uint32_t t2 = 0, rn2 = 32, last = 0;
		bool hit = false;
		for (uint32_t i = 0; i < 32 && !hit; i++) {
			last = mgl_rng_draw(key, i) % c.n;
			if ((b.onwalk[last >> 6] >> (last & 63u)) & 1ull) { hit = true; t2 = last; rn2 = i + 1u; }
		}
		if (!hit) {
			uint32_t wd = last >> 6;
			uint64_t bits = b.onwalk[wd] & (~0ull << (last & 63u));
			while (!bits && ++wd < b.nw0) bits = b.onwalk[wd];
			t2 = bits ? (wd << 6) + ctz64(bits) : 0u;
		}
		if (t2 != target || rn2 != rn) redo = true;
		else if (target < a_lo) {
			const uint32_t wend = out.win[2u * j + 1u];
			if (target > ctl->mod_lo) redo = true;                          /* (a full rebuild reports everything as changed) */
			else if (wend == MGL_WIN_DROPPED) redo = true;                  /* given up on the way: how far it got is not recorded */
			else if (wend != MGL_WIN_NONE && wend > a_lo) redo = true;      /* walked into the accepted window */
		} else {
			const uint32_t m_lo = ctl->mod_lo, m_hi = ctl->mod_hi;
			if (target < a_end) redo = true;
			else if (target > m_lo && target <= m_hi) redo = true;
		}
	}
	if (j < K) mark[j] = redo ? 1u : 0u;
	const unsigned long long m = __ballot(redo);
	if (m) {
		unsigned long long base = 0;
		const int first = __ffsll((long long)m) - 1;
		if ((int)lane == first) base = atomicAdd(&hdr[0], (uint32_t)__popcll(m));
		base = (unsigned long long)shfl64(base, first);
		if (redo) {
			list[(uint32_t)base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = j;
			sim_hdr[j].x = 0xFFFFFFFFu; /* the regular re-simulation launch passes it by */
		}
	}
}
