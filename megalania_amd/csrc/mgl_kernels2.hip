/*
 * mgl_kernels2.hip -- the incremental neighbour path.
 *
 *   k_build        one wavefront walks the base slab twice: (1) bitmaps, per-special state
 *                  records, per-context event counts -> chain offsets; (2) fills the chains
 *                  (position, bit, probability before) and the dense checkpoints, totals the cost
 *   k_neighbours2  one wavefront per neighbour.  Work is proportional to the *changed window*,
 *                  not to the file: walk state at the target from the special records, mutate
 *                  (top-K needs the model at the target: dense checkpoint + <= 64 bytes of replay),
 *                  then a two-pointer walk over the neighbour's and the base's packets that stops
 *                  when position, ctx_state and rep distances agree again; identical packets
 *                  cancel, runs of plain literals are skipped through the special bitmap.  The
 *                  inserted / removed events are then priced per probability context against
 *                  the base chains until each perturbed probability re-joins its base trajectory.
 *                  The result is the neighbour's exact total (u64) -- the same number
 *                  packet_slab_neighbour_generate (packet_slab_neighbour.c:154-173) would return.
 */
#include "mgl_base2.h"

/* ================================================================== k_build */

struct BitAcc {
	uint32_t word;
	uint64_t bits;
};
__device__ __forceinline__ void bitacc_move(BitAcc& a, uint64_t* arr, uint32_t new_word, uint32_t lane)
{
	if (new_word == a.word) return;
	if (lane == 0) arr[a.word] = a.bits;
	for (uint32_t z = a.word + 1 + lane; z < new_word; z += 64) arr[z] = 0;
	a.word = new_word; a.bits = 0;
}

__device__ void build_body(const DevCtx& c, const Base2& b, Control* ctl, int check_cost)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint16_t* T = (uint16_t*)smem;
	uint16_t* probs = (uint16_t*)(smem + 4096);
	uint32_t* cnt = (uint32_t*)(smem + 4096 + (size_t)b.ck_elems * 2);
	uint32_t* off = cnt + b.ck_elems;
	const uint32_t lane = threadIdx.x;
	const uint32_t total = c.L.total;
	/* check_cost: 0 = unconditional build; 1 = per-step rebuild after an accept;
	 * 2 = only when the incremental accept (k_apply_*) gave up */
	if (check_cost && !ctl->accepted_flag) return;
	if (check_cost == 2 && !ctl->apply_failed) return;
	for (uint32_t i = lane; i < 2048; i += 64) T[i] = c.cost_tbl[i];
	for (uint32_t i = lane; i < b.ck_elems; i += 64) { cnt[i] = 0; off[i] = 0; }
	wave_sync();

	/* ---- pass 1: bitmaps, special-state records, event counts */
	Walk w;
	walk_reset(w);
	BitAcc on = { 0, 0 }, sp = { 0, 0 };
	uint32_t guard = 0;
	while (w.st.pos < c.n) {
		const uint32_t pos = w.st.pos;
		if (++guard > c.n) { if (lane == 0) atomicOr(&ctl->error_flags, MGL_ERR_WALK_OVERRUN); break; }
		bitacc_move(on, b.onwalk, pos >> 6, lane);
		bitacc_move(sp, b.sp0, pos >> 6, lane);
		on.bits |= 1ull << (pos & 63u);
		walk_window(w, c, b.slab, lane);
		const mgl_pk pk = walk_slab_at(w, pos);
		uint32_t type = mgl_pk_type(pk), len = mgl_pk_len(pk), dist = mgl_pk_dist(pk);
		if (type < MGL_LITERAL || type > MGL_LONG_REP || len == 0 || pos + len > c.n) {
			type = MGL_LITERAL; len = 1; dist = 0;
			if (lane == 0) atomicOr(&ctl->error_flags, MGL_ERR_WALK_OVERRUN);
		}
		if (type != MGL_LITERAL) {
			sp.bits |= 1ull << (pos & 63u);
			if (lane < 8) {
				const uint32_t v = lane == 0 ? w.st.ctx_state : lane == 1 ? w.st.dists[0] : lane == 2 ? w.st.dists[1]
				                 : lane == 3 ? w.st.dists[2] : lane == 4 ? w.st.dists[3] : 0u;
				b.sp_state[(size_t)pos * 8 + lane] = v;
			}
		}
		const uint32_t byte = walk_byte_at(w, pos);
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
			cnt[ctx]++; /* contexts of one packet are distinct: no conflict */
		}
		mgl_advance(&w.st, type, dist, len);
		w.packets++;
	}
	{
		const uint32_t nw0 = b.nw0;
		bitacc_move(on, b.onwalk, nw0, lane);
		bitacc_move(sp, b.sp0, nw0, lane);
	}
	const uint32_t npackets = w.packets;
	__threadfence();
	wave_sync();
	/* summary levels of the special bitmap */
	for (uint32_t u = 0; u < b.nw1; u++) {
		const uint32_t wd = u * 64 + lane;
		const bool nz = wd < b.nw0 && b.sp0[wd] != 0;
		const unsigned long long m = __ballot(nz);
		if (lane == 0) b.sp1[u] = m;
	}
	__threadfence();
	wave_sync();
	for (uint32_t v = 0; v < b.nw2; v++) {
		const uint32_t u = v * 64 + lane;
		const bool nz = u < b.nw1 && b.sp1[u] != 0;
		const unsigned long long m = __ballot(nz);
		if (lane == 0) b.sp2[v] = m;
	}
	/* chain offsets: capacity = 2 len + 257 rounded up to 8 (incl. the sentinel): room to grow before a
	 * chain has to move (k_apply_chains); lanes own contiguous context ranges */
	{
		const uint32_t per = (total + 63u) / 64u;
		const uint32_t lo = lane * per, hi = (lo + per) < total ? (lo + per) : total;
		uint32_t sum = 0;
		for (uint32_t i = lo; i < hi; i++) sum += (2u * cnt[i] + 257u + 7u) & ~7u;
		uint32_t incl = sum;
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
			if ((int)lane >= o) incl += t;
		}
		uint32_t run = incl - sum;
		for (uint32_t i = lo; i < hi; i++) {
			const uint32_t cap = (2u * cnt[i] + 257u + 7u) & ~7u; /* multiple of 8: 16-byte aligned chains */
			off[i] = run;
			b.ch_off[i] = run; b.ch_len[i] = cnt[i]; b.ch_cap[i] = cap;
			run += cap;
		}
		const uint32_t all = (uint32_t)__shfl((int)incl, 63, 64);
		if (lane == 0) {
			*b.pool_top = all;
			if (all > b.pool_cap) atomicOr(&ctl->error_flags, MGL_ERR_WALK_OVERRUN);
		}
		if (all > b.pool_cap) return;
	}
	wave_sync();
	for (uint32_t i = lane; i < b.ck_elems; i += 64) { cnt[i] = 0; probs[i] = MGL_PROB_INIT; }
	wave_sync();

	/* ---- pass 2: chains + dense checkpoints + chain index + cost */
	walk_reset(w);
	uint32_t next_ck = 0, next_sb = 0;
	guard = 0;
	while (w.st.pos < c.n) {
		const uint32_t pos = w.st.pos;
		if (++guard > c.n) break;
		while (next_sb <= (pos >> b.sb_shift)) { /* entries per context before this block */
			for (uint32_t i = lane; i < total; i += 64) b.ch_sb[(size_t)i * b.sb_stride + next_sb] = cnt[i];
			next_sb++;
		}
		while (next_ck < b.nck && (next_ck << MGL_CK2_SHIFT) <= pos) {
			uint32_t* dst = (uint32_t*)(b.ck_probs + (size_t)next_ck * b.ck_elems);
			const uint32_t* src = (const uint32_t*)probs;
			for (uint32_t i = lane; i < b.ck_elems / 2; i += 64) dst[i] = src[i];
			next_ck++;
		}
		walk_window(w, c, b.slab, lane);
		const mgl_pk pk = walk_slab_at(w, pos);
		uint32_t type = mgl_pk_type(pk), len = mgl_pk_len(pk), dist = mgl_pk_dist(pk);
		if (type < MGL_LITERAL || type > MGL_LONG_REP || len == 0 || pos + len > c.n) { type = MGL_LITERAL; len = 1; dist = 0; }
		const uint32_t byte = walk_byte_at(w, pos);
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
			const uint32_t p = probs[ctx];
			const uint32_t k = off[ctx] + cnt[ctx]++;
			b.ch_pos[k] = pos;
			b.ch_ev[k] = (uint16_t)((bit << 15) | p);
			w.acc += T[bit ? 2048u - p : p];
			probs[ctx] = (uint16_t)mgl_prob_update(p, bit);
		}
		if (lane == 0) w.acc += (uint64_t)pl.ndirect << 11;
		mgl_advance(&w.st, type, dist, len);
		w.packets++;
	}
	wave_sync();
	/* checkpoints whose boundary lies behind the last packet start hold the final model, so
	 * that every checkpoint is defined (k_apply_chains relies on that) */
	while (next_ck < b.nck) {
		uint32_t* dst = (uint32_t*)(b.ck_probs + (size_t)next_ck * b.ck_elems);
		const uint32_t* src = (const uint32_t*)probs;
		for (uint32_t i = lane; i < b.ck_elems / 2; i += 64) dst[i] = src[i];
		next_ck++;
	}
	for (; next_sb <= b.nsb; next_sb++)
		for (uint32_t i = lane; i < total; i += 64) b.ch_sb[(size_t)i * b.sb_stride + next_sb] = cnt[i];
	/* sentinels: position = infinity, probability = the context's final value */
	for (uint32_t i = lane; i < total; i += 64) {
		const uint32_t k = off[i] + cnt[i];
		b.ch_pos[k] = MGL_POS_INF;
		b.ch_ev[k] = probs[i];
	}
	const uint64_t cost = wave_sum64(w.acc);
	if (lane == 0) {
		ctl->packets = npackets;
		ctl->rebuild_cost = cost;
		ctl->final_ctx_state = w.st.ctx_state;
		ctl->final_dists[0] = w.st.dists[0]; ctl->final_dists[1] = w.st.dists[1];
		ctl->final_dists[2] = w.st.dists[2]; ctl->final_dists[3] = w.st.dists[3];
		if (check_cost) {
			if (ctl->cur_cost != cost) atomicOr(&ctl->error_flags, MGL_ERR_REBUILD_MISMATCH);
			ctl->accepted_flag = 0;
			ctl->full_rebuilds++;
			ctl->mod_lo = 0u; ctl->mod_hi = MGL_POS_INF; /* look-ahead: nothing evaluated ahead of this rebuild is kept */
		}
	}
}

__global__ void __launch_bounds__(64) k_build(DevCtx c, Base2 b, Control* ctl, int check_cost)
{
	build_body(c, b, ctl, check_cost);
}
/* the tail of an incremental step in one launch: the fallback rebuild (only when k_apply_* gave
 * up) and then the step's bookkeeping (mgl_kernels3.hip) */
__device__ void step_end_body(Control* ctl, int lazy_best, uint32_t* counts, int adaptive, int form_single);
__global__ void __launch_bounds__(64) k_build_end(DevCtx c, Base2 b, Control* ctl, int lazy_best, uint32_t* counts, int adaptive, int form_single)
{
	build_body(c, b, ctl, 2);
	__threadfence();
	wave_sync();
	step_end_body(ctl, lazy_best, counts, adaptive, form_single);
}

/* ================================================================== change lists */

struct Changes {
	uint16_t* ins_key; /* ctx | bit << 15 */
	uint32_t* ins_pos;
	uint16_t* rem_key; /* ctx */
	uint32_t* rem_pos;
	uint16_t* uctx;     /* distinct touched contexts (scratch of chain_sim) */
	uint32_t* ctxbits;  /* LDS bitmap over the probability contexts (scratch of chain_sim) */
	uint32_t cap;       /* capacity of each list */
	uint32_t uctx_cap;
	uint32_t nbitwords;
	uint32_t n_ins, n_rem; /* uniform */
	int64_t direct;        /* (inserted - removed) direct-bit cost, uniform */
	unsigned long long* dbg; /* diagnostic counters (MGL_F_PROFILE), else nullptr */
	uint32_t diag;
	bool overflow;
	bool list_full;        /* overflow because a list reached its capacity (as opposed to any other reason) */
};

/* append the events of one planned packet to the inserted (INS) or removed list */
template <bool INS>
__device__ __forceinline__ void changes_add(Changes& ch, const mgl_plan& pl, uint32_t pos, uint32_t lane)
{
	uint32_t& n = INS ? ch.n_ins : ch.n_rem;
	if (n + pl.nev > ch.cap) { ch.overflow = true; ch.list_full = true; return; }
	if (lane < pl.nev) {
		uint32_t ctx, bit;
		mgl_plan_event(&pl, lane, &ctx, &bit);
		if (INS) { ch.ins_key[n + lane] = (uint16_t)(ctx | (bit << 15)); ch.ins_pos[n + lane] = pos; }
		else { ch.rem_key[n + lane] = (uint16_t)ctx; ch.rem_pos[n + lane] = pos; }
	}
	n += pl.nev;
	const int64_t d = (int64_t)((uint64_t)pl.ndirect << 11);
	ch.direct += INS ? d : -d;
}

/* first index >= start (and < n) whose key, masked, equals cx; n if none.  Sixteen 16-bit keys
 * per trip: two independent 16-byte reads (the lists are 16-byte aligned), so a scan waits for
 * half as many LDS round trips. */
__device__ __forceinline__ uint32_t next_with_ctx(const uint16_t* keys, uint32_t start, uint32_t n, uint32_t cx, uint32_t mask)
{
	uint32_t base = start & ~7u;
	while (base < n) {
		const uint4 q = *reinterpret_cast<const uint4*>(keys + base);
		uint4 r = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); /* no context has this key */
		if (base + 8u < n) r = *reinterpret_cast<const uint4*>(keys + base + 8u);
		const uint32_t w[8] = { q.x, q.y, q.z, q.w, r.x, r.y, r.z, r.w };
		uint32_t hit = 0;
#pragma unroll
		for (uint32_t e = 0; e < 16; e++) {
			const uint32_t key = (w[e >> 1] >> ((e & 1u) * 16u)) & mask;
			hit |= (key == cx ? 1u : 0u) << e;
		}
		if (base < start) hit &= ~0u << (start - base);
		if (hit) {
			const uint32_t idx = base + (uint32_t)__ffs((int)hit) - 1u;
			return idx < n ? idx : n;
		}
		base += 16;
	}
	return n;
}

/* Price the change lists against the base chains.  Every distinct touched context is
 * re-simulated by one lane from its first change, merging the base chain (minus removed
 * events) with the inserted events by position, until no change is pending and the running
 * probability equals the base's (re-coupled), the chain ends, or `limit` is reached.
 * Returns sum(new cost - base cost) over the simulated spans (all lanes get the value).
 * If overlay != nullptr the probability each touched context has at `limit` is written into
 * it (contexts that re-coupled before `limit` keep the base value already there). */
/* part A (one wavefront): the distinct touched contexts, ascending, into ch.uctx; returns how many */
__device__ __forceinline__ uint32_t chain_list(Changes& ch, uint32_t lane, bool* too_many)
{
	const uint32_t m = ch.n_ins + ch.n_rem;
	wave_sync();
	/* distinct contexts through a bitmap in LDS, then listed in ascending order */
	for (uint32_t i = lane; i < ch.nbitwords; i += 64) ch.ctxbits[i] = 0;
	wave_sync();
	for (uint32_t e = lane; e < m; e += 64) {
		const uint32_t key = e < ch.n_ins ? (ch.ins_key[e] & 0x7FFFu) : ch.rem_key[e - ch.n_ins];
		atomicOr(&ch.ctxbits[key >> 5], 1u << (key & 31u));
	}
	wave_sync();
	uint32_t nu = 0;
	for (uint32_t wbase = 0; wbase < ch.nbitwords; wbase += 64) {
		const uint32_t word = wbase + lane < ch.nbitwords ? ch.ctxbits[wbase + lane] : 0u;
		const uint32_t cntl = (uint32_t)__popc(word);
		uint32_t incl = cntl;
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t tmp = (uint32_t)__shfl_up((int)incl, o, 64);
			if ((int)lane >= o) incl += tmp;
		}
		const uint32_t chunk = (uint32_t)__shfl((int)incl, 63, 64);
		if (nu + chunk > ch.uctx_cap) { *too_many = true; return 0; }
		uint32_t at = nu + incl - cntl;
		uint32_t wv = word;
		while (wv) {
			ch.uctx[at++] = (uint16_t)(((wbase + lane) << 5) + ((uint32_t)__ffs((int)wv) - 1u));
			wv &= wv - 1u;
		}
		nu += chunk;
	}
	wave_sync();
	return nu;
}
/* part B: this lane's share of the contexts -- uctx[first + lane], then every `stride` further on
 * (64 for a wavefront on its own; k_sim puts several wavefronts on one neighbour).  Returns the
 * lane's own sum. */
/* traffic (nullable): the bytes of chain data this lane asked the memory system for are added to it -- 4 per position
 * probed by a search, 48 per eight-entry chunk (positions + events), 32 per sixteen events of the re-coupling stretch:
 * the re-simulation's algorithmic bytes, counted where they are read (bench.py's roofline block) */
__device__ __forceinline__ int64_t chain_sim_contexts(const Base2& b, Changes& ch, const uint16_t* T, uint32_t limit, uint16_t* overlay,
                                                      uint32_t lane, uint32_t first, uint32_t stride, uint32_t nu, uint32_t* traffic = nullptr)
{
	int64_t delta = 0;
	for (uint32_t base = first; base < nu; base += stride) {
		if (base + lane >= nu) continue;
		const uint32_t cx = ch.uctx[base + lane];
		uint32_t ii = next_with_ctx(ch.ins_key, 0, ch.n_ins, cx, 0x7FFFu);
		uint32_t ri = next_with_ctx(ch.rem_key, 0, ch.n_rem, cx, 0xFFFFu);
		const uint32_t ipos0 = ii < ch.n_ins ? ch.ins_pos[ii] : MGL_POS_INF;
		const uint32_t rpos0 = ri < ch.n_rem ? ch.rem_pos[ri] : MGL_POS_INF;
		const uint32_t x0 = ipos0 < rpos0 ? ipos0 : rpos0;
		const uint32_t* cpos = b.ch_pos + b.ch_off[cx];
		const uint16_t* cev = b.ch_ev + b.ch_off[cx];
		const uint32_t clen = b.ch_len[cx];
		uint32_t probes = 0;
		uint32_t blo, bhi; /* the chain index narrows the search to the entries of x0's block */
		chain_block(b, cx, x0, &blo, &bhi);
		uint32_t k = chain_lower_bound(cpos, bhi, x0, traffic ? &probes : nullptr, blo);
		if (traffic) *traffic += 20u; /* chain offset and length of this context, two index entries */
		/* eight chain entries (positions + events) per round trip, kept in registers */
		uint4 c_pa = make_uint4(0, 0, 0, 0), c_pb = c_pa, c_ev = c_pa;
		uint32_t c_base = 0xFFFFFFFFu;
		auto chunk = [&](uint32_t kk) {
			if ((kk & ~7u) != c_base) {
				c_base = kk & ~7u; /* chains start 32-byte aligned; the pool is over-allocated past its end */
				c_pa = *reinterpret_cast<const uint4*>(cpos + c_base);
				c_pb = *reinterpret_cast<const uint4*>(cpos + c_base + 4);
				c_ev = *reinterpret_cast<const uint4*>(cev + c_base);
				if (traffic) *traffic += 48u;
			}
		};
		auto pos_at = [&](uint32_t kk) -> uint32_t {
			const uint32_t e = kk & 7u;
			return e == 0 ? c_pa.x : e == 1 ? c_pa.y : e == 2 ? c_pa.z : e == 3 ? c_pa.w : e == 4 ? c_pb.x : e == 5 ? c_pb.y : e == 6 ? c_pb.z : c_pb.w;
		};
		auto ev_at = [&](uint32_t kk) -> uint32_t {
			const uint32_t e = kk & 7u;
			const uint32_t wd = e < 2 ? c_ev.x : e < 4 ? c_ev.y : e < 6 ? c_ev.z : c_ev.w;
			return (wd >> ((e & 1u) * 16u)) & 0xFFFFu;
		};
		chunk(k);
		uint32_t p = ev_at(k) & 0x7FFu;
		bool at_limit = false, ended = false;
		if (ch.diag == 42) { delta += p; continue; }
		uint32_t iters = 0;
		/* ---- part 1: while this context still has changes ahead (or a limit applies):
		 * merge base entries and inserted events by position; the position of the next change of
		 * either list is kept in a register */
		uint32_t ipos = ipos0, rpos = rpos0;
		for (;;) {
			const bool pending = ipos != MGL_POS_INF || rpos != MGL_POS_INF;
			if (!pending && limit == MGL_POS_INF) break; /* -> part 2 */
			iters++;
			chunk(k);
			const uint32_t bpos = pos_at(k);
			if (ipos < bpos) {
				if (ipos >= limit) { at_limit = true; ended = true; break; }
				const uint32_t bit = ch.ins_key[ii] >> 15;
				delta += T[bit ? 2048u - p : p];
				p = mgl_prob_update(p, bit);
				ii = next_with_ctx(ch.ins_key, ii + 1, ch.n_ins, cx, 0x7FFFu);
				ipos = ii < ch.n_ins ? ch.ins_pos[ii] : MGL_POS_INF;
				continue;
			}
			if (bpos == MGL_POS_INF) { ended = true; break; }          /* chain exhausted */
			if (bpos >= limit) { at_limit = true; ended = true; break; }
			const uint32_t ev = ev_at(k);
			const uint32_t bp = ev & 0x7FFu, bb = ev >> 15;
			if (p == bp) {
				if (!pending) { ended = true; break; }   /* re-coupled, nothing ahead (limit mode) */
				/* re-coupled, but this context changes again further on: everything up to
				 * that position is coded exactly as in the base, so jump there */
				const uint32_t nxt = ipos < rpos ? ipos : rpos;
				if (nxt > bpos) {
					if (nxt >= limit) { ended = true; break; } /* the base value holds at the limit */
					chain_block(b, cx, nxt, &blo, &bhi);
					k = chain_lower_bound(cpos, bhi, nxt, traffic ? &probes : nullptr, blo > k ? blo : k);
					if (traffic) *traffic += 8u;
					chunk(k);
					p = ev_at(k) & 0x7FFu;
					continue;
				}
			}
			delta -= T[bb ? 2048u - bp : bp];
			if (rpos == bpos) {
				ri = next_with_ctx(ch.rem_key, ri + 1, ch.n_rem, cx, 0xFFFFu);
				rpos = ri < ch.n_rem ? ch.rem_pos[ri] : MGL_POS_INF;
			} else {
				delta += T[bb ? 2048u - p : p];
				p = mgl_prob_update(p, bb);
			}
			k++;
		}
		if (ch.diag == 43) { delta += p; continue; }
		/* ---- part 2: no change ahead: follow the base chain until the probability re-joins
		 * the base trajectory.  Only (bit, base probability) is needed: 16 entries per round
		 * trip (two 16-byte loads), the next 16 already in flight while these are processed. */
		if (!ended) {
			uint32_t kb = k & ~15u;
			uint4 a0 = *reinterpret_cast<const uint4*>(cev + kb);
			uint4 a1 = *reinterpret_cast<const uint4*>(cev + kb + 8);
			int32_t d32 = 0; /* 16 events x 22 528 at most per round: widened once per round */
			while (!ended) {
				/* may read up to 64 bytes past the sentinel: the pool is over-allocated for that */
				const uint4 n0 = *reinterpret_cast<const uint4*>(cev + kb + 16);
				const uint4 n1 = *reinterpret_cast<const uint4*>(cev + kb + 24);
				if (traffic) *traffic += 32u;
#pragma unroll
				for (uint32_t e = 0; e < 16; e++) {
					const uint32_t idx = kb + e;
					if (ended || idx < k) continue;
					if (idx >= clen) { ended = true; continue; }               /* chain exhausted */
					const uint4 q = e < 8 ? a0 : a1;
					const uint32_t h = e & 7u;
					const uint32_t wd = h < 2 ? q.x : h < 4 ? q.y : h < 6 ? q.z : q.w;
					const uint32_t ev = (wd >> ((h & 1u) * 16u)) & 0xFFFFu;
					const uint32_t bp = ev & 0x7FFu, bb = ev >> 15;
					if (p == bp) { ended = true; continue; }                   /* re-coupled */
					d32 += (int32_t)T[bb ? 2048u - p : p] - (int32_t)T[bb ? 2048u - bp : bp];
					p = mgl_prob_update(p, bb);
				}
				delta += d32; d32 = 0;
				a0 = n0; a1 = n1; kb += 16;
			}
			k = kb;
		}
		if (ch.dbg) atomicMax(&ch.dbg[21], (unsigned long long)iters);
		if (traffic) *traffic += 4u * probes;
		if (overlay && (at_limit || (limit != MGL_POS_INF && cpos[k] == MGL_POS_INF))) overlay[cx] = (uint16_t)p;
		(void)c_base;
	}
	return delta;
}
__device__ int64_t chain_sim(const Base2& b, Changes& ch, const uint16_t* T, uint32_t limit, uint16_t* overlay,
                             uint32_t lane, bool* too_many)
{
	const uint32_t nu = chain_list(ch, lane, too_many);
	if (*too_many) return 0;
	if (ch.diag == 41) return (int64_t)nu;
	const int64_t delta = chain_sim_contexts(b, ch, T, limit, overlay, lane, 0u, 64u, nu);
	/* signed wave sum */
	uint64_t u = (uint64_t)delta;
	u = wave_sum64(u);
	wave_sync();
	return (int64_t)u;
}

/* ================================================================== k_neighbours2 */

/* optional per-phase cycle accounting (diagnostic: prof == nullptr in normal runs) */
struct Prof {
	unsigned long long* acc;
	unsigned long long t;
};
__device__ __forceinline__ void prof_start(Prof& p, unsigned long long* acc) { p.acc = acc; if (acc) p.t = __builtin_readcyclecounter(); }
__device__ __forceinline__ void prof_mark(Prof& p, int phase, uint32_t lane)
{
	if (!p.acc) return;
	const unsigned long long now = __builtin_readcyclecounter();
	if (lane == 0) { atomicAdd(&p.acc[phase], now - p.t); atomicAdd(&p.acc[8 + phase], 1ull); atomicMax(&p.acc[16 + phase], now - p.t); }
	p.t = __builtin_readcyclecounter();
}

struct Win {
	uint32_t base;
	mgl_pk pk;
	uint32_t byte;
};
__device__ __forceinline__ void win_cover(Win& w, const DevCtx& c, const mgl_pk* slab, uint32_t pos, uint32_t lane)
{
	const uint32_t base = pos & ~63u;
	if (base != w.base) {
		const uint32_t p = base + lane;
		w.pk = p < c.n ? slab[p] : 0;
		w.byte = c.data[p < c.n ? p : c.n];
		w.base = base;
	}
}
__device__ __forceinline__ mgl_pk win_pk(const Win& w, uint32_t pos) { return rdlane64(w.pk, pos - w.base); }
__device__ __forceinline__ uint32_t win_byte(const Win& w, uint32_t pos) { return rdlane(w.byte, pos - w.base); }

__device__ __forceinline__ void plan_at(const DevCtx& c, const mgl_wstate& st, uint32_t type, uint32_t dist, uint32_t len,
                                        uint32_t byte, mgl_plan& pl)
{
	uint32_t match_byte = 0, prev_byte = 0;
	if (type == MGL_LITERAL) {
		if (st.ctx_state >= 7 && st.dists[0] < st.pos) match_byte = c.data[st.pos - st.dists[0] - 1];
		if (c.L.lc > 0 && st.pos > 0) prev_byte = c.data[st.pos - 1];
	}
	mgl_plan_packet(&c.L, &st, type, dist, len, byte, match_byte, prev_byte, &pl);
}

/* Base packets the neighbour's walk has passed over disappear, and under a freshly picked match
 * they are mostly a run of plain literals (ctx_state < 7, so no match byte): up to seven of them
 * are planned at once, nine lanes each (is_match + the eight literal bits), instead of one per
 * turn of the walk loop.  Uses only the slab / input window in registers.  Returns the number of
 * packets removed (0: not a plain-literal run of at least two -- take the packet-at-a-time path). */
template <bool INS>
__device__ __forceinline__ uint32_t literal_run_events(const DevCtx& c, Changes& ch, const Win& win, mgl_wstate& st, uint32_t limit, uint32_t lane)
{
	if (st.ctx_state >= 7u) return 0;
	const uint32_t o = st.pos - win.base;
	const unsigned long long lit = __ballot(mgl_pk_type(win.pk) == MGL_LITERAL) >> o; /* window entries from st.pos on */
	uint32_t run = (uint32_t)__ffsll((long long)~lit) - 1u; /* consecutive literals: each one's successor is on the walk */
	if (~lit == 0ull) run = 64u;
	run = run < 64u - o ? run : 64u - o;
	run = run < limit ? run : limit;
	if (run < 2u) return 0;
	const uint32_t take = run < 7u ? run : 7u;
	uint32_t& n = INS ? ch.n_ins : ch.n_rem;
	if (n + 9u * take > ch.cap) { ch.overflow = true; ch.list_full = true; return take; }
	const uint32_t i = lane / 9u, slot = lane - i * 9u;
	const uint32_t p = st.pos + i;
	const bool active = i < take;
	const uint32_t byte = (uint32_t)__shfl((int)win.byte, (int)((p - win.base) & 63u), 64);
	uint32_t prev_byte = 0;
	if (c.L.lc > 0) {
		const uint32_t wprev = (uint32_t)__shfl((int)win.byte, (int)((p - 1u - win.base) & 63u), 64);
		prev_byte = p == 0 ? 0u : (p - 1u >= win.base ? wprev : (uint32_t)c.data[p - 1u]);
	}
	mgl_wstate sv = st;
	sv.pos = p;
	sv.ctx_state = lit_steps(st.ctx_state, i);
	mgl_plan pl;
	mgl_plan_packet(&c.L, &sv, MGL_LITERAL, 0, 1, byte, 0, prev_byte, &pl);
	if (__ballot(active && pl.nev != 9u)) return 0; /* not the 1 + 8 events assumed above: one at a time */
	if (active) {
		uint32_t ctx, bit;
		mgl_plan_event(&pl, slot, &ctx, &bit);
		const uint32_t at = n + i * 9u + slot;
		if (INS) { ch.ins_key[at] = (uint16_t)(ctx | (bit << 15)); ch.ins_pos[at] = p; }
		else { ch.rem_key[at] = (uint16_t)ctx; ch.rem_pos[at] = p; }
	}
	n += 9u * take;
	st.pos += take;
	st.ctx_state = lit_steps(st.ctx_state, take);
	return take;
}

/* The base's adaptive model before the first base packet at or after y, into LDS: dense
 * checkpoint + replay of < 2^MGL_CK2_SHIFT bytes of base packets. */
__device__ void model_load(const DevCtx& c, const Base2& b, uint16_t* probs, const uint16_t* T, uint32_t y, uint32_t lane)
{
	const uint32_t ck = y >> MGL_CK2_SHIFT;
	/* 16 bytes per lane and four loads in flight per trip (rows are 16-byte multiples, 16-byte aligned on both sides):
	 * a 5 KiB model is two round trips instead of twenty-one */
	const uint4* src = (const uint4*)(b.ck_probs + (size_t)ck * b.ck_elems);
	uint4* dst = (uint4*)probs;
	const uint32_t n16 = b.ck_elems / 8u;
	for (uint32_t base = 0; base < n16; base += 256u) {
		uint4 r[4];
#pragma unroll
		for (uint32_t u = 0; u < 4; u++) { const uint32_t i = base + u * 64u + lane; if (i < n16) r[u] = src[i]; }
#pragma unroll
		for (uint32_t u = 0; u < 4; u++) { const uint32_t i = base + u * 64u + lane; if (i < n16) dst[i] = r[u]; }
	}
	/* first base packet start in [ck << shift, y): one bitmap word holds the whole checkpoint block */
	const uint32_t first = ck << MGL_CK2_SHIFT;
	const uint64_t bits = b.onwalk[first >> 6] & ((y & 63u) ? ((1ull << (y & 63u)) - 1ull) : 0ull) & (~0ull << (first & 63u));
	wave_sync();
	if (bits) {
		const uint32_t start = uni(((first >> 6) << 6) + ctz64(bits));
		Walk w;
		walk_reset(w);
		w.st = uni_state(base_state_at(b, start));
		while (w.st.pos < y) {
			walk_window(w, c, b.slab, lane);
			if (w.st.ctx_state < 7u) {
				/* a run of plain literals: up to seven are planned at once, nine lanes each; their
				 * probability updates are then applied packet by packet (the packets share contexts:
				 * is_match and the top of the literal tree), which is all that has to stay in order */
				const uint32_t o = w.st.pos - w.wbase;
				const unsigned long long lit = __ballot(mgl_pk_type(w.wpk) == MGL_LITERAL) >> o;
				uint32_t run = ~lit == 0ull ? 64u : (uint32_t)__ffsll((long long)~lit) - 1u;
				run = run < 64u - o ? run : 64u - o;
				run = run < y - w.st.pos ? run : y - w.st.pos;
				if (run >= 2u) {
					const uint32_t take = run < 7u ? run : 7u;
					const uint32_t i = lane / 9u, slot = lane - i * 9u, p = w.st.pos + i;
					const bool active = i < take;
					const uint32_t byte = (uint32_t)__shfl((int)w.wbyte, (int)((p - w.wbase) & 63u), 64);
					uint32_t prev_byte = 0;
					if (c.L.lc > 0) {
						const uint32_t wprev = (uint32_t)__shfl((int)w.wbyte, (int)((p - 1u - w.wbase) & 63u), 64);
						prev_byte = p == 0 ? 0u : (p - 1u >= w.wbase ? wprev : (uint32_t)c.data[p - 1u]);
					}
					mgl_wstate sv = w.st;
					sv.pos = p; sv.ctx_state = lit_steps(w.st.ctx_state, i);
					mgl_plan pl;
					mgl_plan_packet(&c.L, &sv, MGL_LITERAL, 0, 1, byte, 0, prev_byte, &pl);
					if (!__ballot(active && pl.nev != 9u)) {
						uint32_t ctx = 0, bit = 0;
						if (active) mgl_plan_event(&pl, slot, &ctx, &bit);
						for (uint32_t r = 0; r < take; r++) {
							if (active && i == r) probs[ctx] = (uint16_t)mgl_prob_update(probs[ctx], bit);
							wave_sync();
						}
						w.st.pos += take; w.st.ctx_state = lit_steps(w.st.ctx_state, take);
						w.packets += take;
						continue;
					}
				}
			}
			const mgl_pk pk = walk_slab_at(w, w.st.pos);
			walk_packet<true>(w, c, probs, T, mgl_pk_type(pk), mgl_pk_dist(pk), mgl_pk_len(pk), lane);
		}
		wave_sync();
	}
}

/* the Walk-based helpers of the full-walk path read the input through Walk's window */
__device__ __forceinline__ void walk_from_state(Walk& w, const mgl_wstate& st)
{
	walk_reset(w);
	w.st = st;
}

/* BIG = false: the regular launch, change lists in LDS (MGL_CHG_CAP events each).
 * BIG = true: second chance for the few neighbours whose lists overflowed LDS: same code, lists
 * in a per-neighbour global scratch area (big.cap events each); what overflows even that goes
 * to the full-walk kernel. */
struct BigScratch {
	uint16_t* ins_key; uint32_t* ins_pos; uint16_t* rem_key; uint32_t* rem_pos; uint16_t* uctx;
	uint32_t cap, uctx_cap, slots;
	const uint32_t* todo_in; const uint32_t* todo_in_count;
	uint32_t* spill_ctr; /* slots handed out to first-pass wavefronts that had to spill their lists */
	/* re-simulation handed to k_sim (several wavefronts per neighbour): per neighbour a header
	 * {n_ins, n_rem, direct lo, direct hi} (n_ins = ~0: nothing to do) and the two change lists */
	uint32_t chg_cap; /* events per first-pass list (MGL_CHG_CAP, more when a step has few neighbours and LDS to spare) */
	uint4* sim_hdr;
	uint4* sim_hdr2; /* the same for the second pass's neighbours (k_sim's second launch); nullptr: the second pass re-simulates inline */
	uint32_t* sim_slot2; /* per neighbour: the scratch slot its lists sit in */
	uint32_t lds_cache; /* second pass: 12 * MGL_BIG_CAP bytes of LDS behind the wavefront's area hold a copy of the lists during a re-simulation */
	uint16_t* sim_keys; /* per neighbour: ins_key[chg_cap] | rem_key[chg_cap] */
	uint32_t* sim_pos;  /* per neighbour: ins_pos[chg_cap] | rem_pos[chg_cap] */
	/* look-ahead (mgl_kernels4.hip:k_la_check): the regular launch over a list of neighbours instead of a slice, and the
	 * second pass skipping list entries of the speculative launch that the check took back */
	const uint32_t* la_list; const uint32_t* la_count;
	const uint8_t* la_mark; const uint32_t* la_spec_count;
	const uint32_t* todo_first; /* second pass over the list's tail only: entries from *todo_first on (nullptr: all) */
	/* continuation records, one per scratch slot (MGL_CONT_WORDS u32 each): a second-half wavefront whose repair needs a top-K pick
	 * saves its walk here (state, journal; its change lists go to the slot's list area) and the second pass resumes it at the
	 * pick instead of evaluating the neighbour again from its target.  nullptr: no continuations (restart, as before) */
	uint32_t* cont;
};
#define MGL_CONT_WORDS 368u       /* 48 header words | 64 journal positions | 64 old packets | 64 new packets */
#define MGL_CONT_MAGIC 0x434F4E54u
/* MODE: the regular launch is split in two so that each half needs fewer registers and more
 * wavefronts fit a SIMD (top-K is the register hog):
 *   MGL_NBR_PICK  everything up to the mutated packet: target, walk state, and -- unless the
 *                 mutation is a grow/shrink -- the model at the target and the top-K pick; the
 *                 picked packet and the RNG position go to `pickrec`;
 *   MGL_NBR_REST  recomputes the cheap part, takes the pick from `pickrec`, then window walk and
 *                 chain re-simulation.  A repair that needs another top-K pick (rare) hands the
 *                 neighbour to the next pass;
 *   MGL_NBR_FULL  the whole thing in one kernel (the BIG second pass, which starts from scratch). */
#ifndef MGL_PICK_T_GLOBAL
#define MGL_PICK_T_GLOBAL 1 /* the pick half reads the 4 KiB cost table through the vector cache instead of keeping a copy in LDS: 16 instead of 11 wavefronts per CU at 10 MB (c3 neighbour kernels - 5 %), no change at 100 KB */
#endif
#ifndef MGL_REST_WAVES_PER_SIMD
#define MGL_REST_WAVES_PER_SIMD 4
#endif
#define MGL_SIM2_CAP 4096u /* events per list k_sim's second launch takes (the second pass's neighbours) */
#define MGL_NBR_FULL 0
#define MGL_NBR_PICK 1
#define MGL_NBR_REST 2
/* one neighbour, by the wavefront `wid` of its workgroup; `unit` = the neighbour's index in this launch's slice
 * (regular launch) or its slot in the second pass's list (BIG) */
template <bool BIG, int MODE, bool LIST>
__device__ __forceinline__ void nbr2_one(const DevCtx& c, const Base2& b, Control* ctl, uint64_t seed,
                                         uint64_t step_override, uint32_t K, const NbrOut& out, uint32_t per_wave_bytes,
                                         uint32_t* todo, uint32_t* todo_count, unsigned long long* prof_acc,
                                         const BigScratch& big, uint4* pickrec, uint32_t j_base, uint32_t j_end, uint4* pickstate,
                                         unsigned char* smem, const uint16_t* T, uint32_t unit, uint32_t lane, uint32_t wid)
{
	uint32_t j = j_base + unit; /* [j_base, j_end): the slice of the step this launch covers */
	uint32_t slot = 0;
	const unsigned long long t_begin = prof_acc ? __builtin_readcyclecounter() : 0ull;
	if (!BIG && LIST) j = uni(big.la_list[unit]); /* the launch covers a list (look-ahead: the neighbours evaluated again) */
	if (BIG) {
		slot = unit + ((LIST && big.todo_first != nullptr) ? *big.todo_first : 0u);
		const uint32_t nflag = *big.todo_in_count;
		if (slot >= nflag) return;
		j = uni(big.todo_in[slot]);
		/* an entry the speculative launch made for a neighbour that was evaluated again since: that evaluation speaks for it */
		if (LIST && big.la_mark != nullptr && slot < *big.la_spec_count && j < K && big.la_mark[j]) return;
		if (lane == 0) atomicAdd((unsigned long long*)&ctl->big_nbrs, 1ull); /* counted where the second pass takes it up */
	}
	if (j >= K || (!BIG && j >= j_end)) return;
	/* LDS per wavefront: [journal | context bitmap | union].  The union holds EITHER the model +
	 * top-K price tables (while the mutation is chosen) OR the change lists (afterwards): the
	 * lists only start to fill once the mutated packet is known.  A repair that needs another
	 * top-K pick while the lists are live is handed to the BIG pass, whose lists are in global
	 * memory.  This keeps 11 instead of 8 wavefronts per CU. */
	unsigned char* mine = smem + ((MODE == MGL_NBR_REST || (MODE == MGL_NBR_PICK && MGL_PICK_T_GLOBAL)) ? 0u : 4096u) + (size_t)wid * per_wave_bytes;
	Journal jn;
	jn.old = (mgl_pk*)mine;
	jn.neu = jn.old + MGL_MAX_DIFFS;
	jn.pos = (uint32_t*)(jn.neu + MGL_MAX_DIFFS);
	jn.count = 0; jn.overflow = false;
	Changes ch;
	ch.nbitwords = (c.L.total + 31u) >> 5;
	ch.ctxbits = jn.pos + MGL_MAX_DIFFS;
	/* the pick half never reads the journal or the bitmap: its wave area is the model + price tables alone */
	unsigned char* uni_base = MODE == MGL_NBR_PICK ? mine : (unsigned char*)(ch.ctxbits + ((ch.nbitwords + 3u) & ~3u));
	uint16_t* probs = (uint16_t*)uni_base;
	uint32_t* lencost = (uint32_t*)(uni_base + (size_t)b.ck_elems * 2);
	ch.ins_pos = (uint32_t*)uni_base;
	ch.rem_pos = ch.ins_pos + big.chg_cap;
	ch.ins_key = (uint16_t*)(ch.rem_pos + big.chg_cap);
	ch.rem_key = ch.ins_key + big.chg_cap;
	ch.uctx = ch.rem_key + big.chg_cap;
	ch.cap = big.chg_cap; ch.uctx_cap = 2 * big.chg_cap;
	if (BIG) {
		if (slot >= big.slots) { /* more flagged neighbours than scratch slots: full walk */
			if (lane == 0) { const uint32_t s2 = atomicAdd(todo_count, 1u); todo[s2] = j; }
			return;
		}
		ch.ins_key = big.ins_key + (size_t)slot * big.cap; ch.ins_pos = big.ins_pos + (size_t)slot * big.cap;
		ch.rem_key = big.rem_key + (size_t)slot * big.cap; ch.rem_pos = big.rem_pos + (size_t)slot * big.cap;
		ch.uctx = big.uctx + (size_t)slot * big.uctx_cap;
		ch.cap = big.cap; ch.uctx_cap = big.uctx_cap;
	}
	ch.n_ins = ch.n_rem = 0; ch.direct = 0; ch.overflow = false; ch.list_full = false;
	bool too_many = false;
	bool spilled = BIG;
	const bool spilled_lds = false; /* BIG: the lists live in this neighbour's global scratch slot from the start */

	const uint64_t gstep = step_override != ~0ull ? step_override : ctl->gstep;
	NbrRng rng; rng.key = mgl_rng_key(seed, gstep, j); rng.n = 0;
	/* second pass: a walk the second half saved at its repair pick (same neighbour, same slot) is taken up there */
	uint32_t* const crec = (BIG && big.cont != nullptr) ? big.cont + (size_t)slot * MGL_CONT_WORDS : nullptr;
	bool resumed = false;
	if (BIG && crec != nullptr) resumed = uni(crec[0]) == MGL_CONT_MAGIC && uni(crec[1]) == j;

	/* target (same rule as the full-walk path) and the walk state there; the second half of a split
	 * launch takes both from the first half's record instead of drawing and searching again */
	uint32_t target;
	mgl_wstate nb; /* neighbour's walk state */
	if (MODE == MGL_NBR_REST && lane == 0) big.sim_hdr[j] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
	if (BIG && big.sim_hdr2 != nullptr && lane == 0) big.sim_hdr2[j] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
	if (MODE == MGL_NBR_REST) {
		const uint4 s0 = pickstate[2u * j], s1 = pickstate[2u * j + 1u];
		target = s0.x; rng.n = s0.y;
		nb.pos = target; nb.ctx_state = s0.z;
		nb.dists[0] = s0.w; nb.dists[1] = s1.x; nb.dists[2] = s1.y; nb.dists[3] = s1.z;
	} else if (BIG && resumed) {
		target = uni(crec[2]);
		nb.pos = uni(crec[3]); nb.ctx_state = uni(crec[4]);
		nb.dists[0] = uni(crec[5]); nb.dists[1] = uni(crec[6]); nb.dists[2] = uni(crec[7]); nb.dists[3] = uni(crec[8]);
	} else if (c.strat_pre != nullptr) {
		target = uni(c.strat_tgt[j]); /* stratified_target(), worked out for the whole step by k_targets */
		rng.n = 1;
		nb = uni_state(base_state_at(b, target));
		if (MODE == MGL_NBR_PICK && lane == 0) {
			pickstate[2u * j] = make_uint4(target, rng.n, nb.ctx_state, nb.dists[0]);
			pickstate[2u * j + 1u] = make_uint4(nb.dists[1], nb.dists[2], nb.dists[3], 0u);
		}
	} else {
		uint32_t mydraw = lane < 32 ? mgl_rng_draw(rng.key, lane) % c.n : 0;
		bool on = lane < 32 && ((b.onwalk[mydraw >> 6] >> (mydraw & 63u)) & 1ull);
		unsigned long long m = __ballot(on);
		if (m) {
			int f = __ffsll((long long)m) - 1;
			target = rdlane(mydraw, (uint32_t)f);
			rng.n = (uint32_t)f + 1;
		} else {
			rng.n = 32;
			uint32_t p = rdlane(mydraw, 31);
			uint32_t wd = p >> 6;
			uint64_t bits = b.onwalk[wd] & (~0ull << (p & 63u));
			while (!bits && ++wd < b.nw0) bits = b.onwalk[wd];
			target = uni(bits ? (wd << 6) + ctz64(bits) : 0u);
		}
		nb = uni_state(base_state_at(b, target));
		if (MODE == MGL_NBR_PICK && lane == 0) {
			pickstate[2u * j] = make_uint4(target, rng.n, nb.ctx_state, nb.dists[0]);
			pickstate[2u * j + 1u] = make_uint4(nb.dists[1], nb.dists[2], nb.dists[3], 0u);
		}
	}
	Prof prof;
	prof_start(prof, prof_acc);
	ch.dbg = prof_acc;
	ch.diag = c.diag_stop;
	const uint32_t pos = target;
	mgl_wstate bs = nb;                               /* base's walk state */
	Win win; win.base = 0xFFFFFFFFu; win.pk = 0; win.byte = 0;
	Walk tw; /* scratch Walk for top-K (its window is the input window at the query position) */

	/* ---- mutate, packet_slab_neighbour.c:119-152 */
	win_cover(win, c, b.slab, pos, lane);
	const mgl_pk first = win_pk(win, pos);
	mgl_pk m_first = first, m_second = 0;
	bool second_set = false, mutated = false;
	prof_mark(prof, 0, lane); /* state at target */
	if (c.diag_stop == 1) { if (lane == 0) { out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = first; } return; }
	if (!(BIG && resumed) && pos + 1 < c.n && (nbr_draw(rng) % 2u) == 0) {
		const mgl_pk second = uni64(b.slab[pos + 1]);
		const uint32_t ft = mgl_pk_type(first), flen = mgl_pk_len(first);
		const uint32_t st = mgl_pk_type(second), slen = mgl_pk_len(second), sdist = mgl_pk_dist(second);
		if ((ft == MGL_LONG_REP || ft == MGL_MATCH) && flen > 2) {
			m_second = mgl_pack(ft, mgl_pk_dist(first), flen - 1);
			m_first = MGL_PK_LITERAL;
			if (MODE != MGL_NBR_PICK) { journal_set(jn, pos, first, m_first, lane); journal_set(jn, pos + 1, second, m_second, lane); }
			second_set = true; mutated = true;
		} else if ((ft == MGL_LITERAL || ft == MGL_SHORT_REP) && (st == MGL_MATCH || st == MGL_LONG_REP)) {
			uint32_t rep_start = pos - (st == MGL_LONG_REP ? mgl_dist_at(&nb, sdist) : sdist);
			if (slen < MGL_MAX_MATCH && rep_start > 0 && rep_start <= pos && win_byte(win, pos) == c.data[rep_start - 1]) {
				m_first = mgl_pack(st, sdist, slen + 1);
				if (MODE != MGL_NBR_PICK) journal_set(jn, pos, first, m_first, lane);
				mutated = true;
			}
		}
	}
	/* ---- the rest runs as a small phase machine so that each big piece of code (model load,
	 * chain re-simulation, top-K) exists exactly once in the kernel: the mutation's top-K pick and
	 * a repair's top-K pick share one site, and so do the overlay re-simulation (model at a repair
	 * position) and the final one.  Halves the code size and the register pressure. */
	enum { P_MODEL, P_SIM, P_TOPK, P_WALK, P_OUT };
	uint32_t phase = mutated ? P_WALK : P_MODEL;
	if (MODE == MGL_NBR_PICK && mutated) return; /* nothing to pick: the second half redoes the grow/shrink itself */
	if (MODE == MGL_NBR_REST && c.diag_stop != 0 && c.diag_stop != 4 && c.diag_stop < 40) return; /* diagnostic stops of the first half */
	/* pending top-K request */
	bool pick_is_mutation = !mutated;
	uint32_t pick_pos = pos;
	mgl_pk pick_inc = first;
	bool pick_best = false;
	/* its result, handed back to the walk */
	bool have_pick = false;
	mgl_pk picked_pk = 0, resume_old = 0;
	/* re-simulation request */
	uint32_t sim_limit = MGL_POS_INF;
	bool sim_overlay = false;
	int64_t delta = 0;
	bool generate_failed = false, sim_deferred = false, walk_long = false;
	uint32_t npicks = 0; /* repair picks of this neighbour */
	uint32_t count = 0;   /* repair packet counter of packet_slab_neighbour.c:84-86, saturating */
	uint32_t walked = 0;
	uint32_t wend = 0; /* where the two walks meet again: the end of this neighbour's window */
	/* for the bulk step's selection: the first meeting point -- same byte, same ctx_state -- behind
	 * the last rep packet the walk meets, and whether a rep packet of the neighbour reads a distance from before the window */
	uint32_t wsoft = 0xFFFFFFFFu, taint = 0xFu, dep = 0u;
	bool first_packet = true;
	uint32_t guard = 0;
	/* the second pass, too, takes the mutation's pick from the first half when there was one */
	if (BIG && resumed) {
		/* the saved walk: everything the loop below carries from one packet to the next */
		bs.pos = uni(crec[9]); bs.ctx_state = uni(crec[10]);
		bs.dists[0] = uni(crec[11]); bs.dists[1] = uni(crec[12]); bs.dists[2] = uni(crec[13]); bs.dists[3] = uni(crec[14]);
		count = uni(crec[15]); walked = uni(crec[16]); npicks = uni(crec[17]); wsoft = uni(crec[18]); taint = uni(crec[19]); dep = uni(crec[20]);
		rng.n = uni(crec[21]); guard = uni(crec[22]); jn.count = uni(crec[23]); ch.n_ins = uni(crec[24]); ch.n_rem = uni(crec[25]);
		const uint32_t fl = uni(crec[26]);
		pick_best = (fl & 1u) != 0; second_set = (fl & 2u) != 0;
		pick_inc = (mgl_pk)uni(crec[28]) | ((mgl_pk)uni(crec[29]) << 32);
		resume_old = (mgl_pk)uni(crec[30]) | ((mgl_pk)uni(crec[31]) << 32);
		m_second = (mgl_pk)uni(crec[32]) | ((mgl_pk)uni(crec[33]) << 32);
		ch.direct = (int64_t)((uint64_t)uni(crec[34]) | ((uint64_t)uni(crec[35]) << 32));
		if (lane < jn.count) {
			jn.pos[lane] = crec[48u + lane];
			jn.old[lane] = reinterpret_cast<const mgl_pk*>(crec + 112u)[lane];
			jn.neu[lane] = reinterpret_cast<const mgl_pk*>(crec + 240u)[lane];
		}
		wave_sync();
		if (lane == 0) crec[0] = 0u; /* taken */
		pick_pos = nb.pos;
		mutated = true; pick_is_mutation = false; first_packet = false;
		phase = P_MODEL;
	}
	if ((MODE == MGL_NBR_REST || (BIG && pickrec != nullptr)) && !mutated) { /* the host passes the pick records only when the split form ran */
		const uint4 rec = pickrec[j];
		if (!(rec.w & 1u)) { generate_failed = true; phase = P_OUT; }
		else {
			m_first = (mgl_pk)rec.x | ((mgl_pk)rec.y << 32);
			journal_set(jn, pos, first, m_first, lane);
			rng.n = rec.z;
			pick_is_mutation = false;
			phase = P_WALK;
		}
	}

	for (;;) {
		if (MODE == MGL_NBR_REST && (phase == P_MODEL || phase == P_TOPK)) { ch.overflow = true; phase = P_OUT; continue; } /* repair pick without a continuation record: the next pass evaluates the neighbour again */
		if (MODE != MGL_NBR_REST && phase == P_MODEL) {
			/* the adaptive model the neighbour has at pick_pos: base model before the first base
			 * packet at or after it (dense checkpoint + replay of < 64 bytes of base packets) ... */
			if ((ch.n_ins + ch.n_rem) != 0 && !spilled) {
				/* the model is about to overwrite the live lists (they share LDS): move the lists to a
				 * global scratch slot and carry on from there */
				/* look-ahead's fresh evaluations run beside the second pass, which owns the scratch slots by list index: no spill, next pass */
				if (!BIG && LIST) { ch.overflow = true; phase = P_OUT; continue; }
				uint32_t sl = 0;
				if (lane == 0) sl = atomicAdd(big.spill_ctr, 1u);
				sl = uni(sl);
				if (sl >= big.slots) { ch.overflow = true; phase = P_OUT; continue; }
				uint16_t* gik = big.ins_key + (size_t)sl * big.cap; uint32_t* gip = big.ins_pos + (size_t)sl * big.cap;
				uint16_t* grk = big.rem_key + (size_t)sl * big.cap; uint32_t* grp = big.rem_pos + (size_t)sl * big.cap;
				for (uint32_t i = lane; i < ch.n_ins; i += 64) { gik[i] = ch.ins_key[i]; gip[i] = ch.ins_pos[i]; }
				for (uint32_t i = lane; i < ch.n_rem; i += 64) { grk[i] = ch.rem_key[i]; grp[i] = ch.rem_pos[i]; }
				__threadfence_block();
				wave_sync();
				ch.ins_key = gik; ch.ins_pos = gip; ch.rem_key = grk; ch.rem_pos = grp;
				ch.uctx = big.uctx + (size_t)sl * big.uctx_cap;
				ch.cap = big.cap; ch.uctx_cap = big.uctx_cap;
				spilled = true;
			}
			model_load(c, b, probs, T, pick_pos, lane);
			if (!pick_is_mutation) prof_mark(prof, 5, lane); /* repair: base model at the pick position */
			if (pick_is_mutation) {
				prof_mark(prof, 1, lane); /* model at target */
				if (c.diag_stop == 2) { if (lane == 0) { out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = probs[lane]; } return; }
			}
			/* ... with the touched contexts overridden by their re-simulated values */
			if ((ch.n_ins + ch.n_rem) != 0) { sim_limit = pick_pos; sim_overlay = true; phase = P_SIM; }
			else phase = P_TOPK;
		}
		if (BIG && big.sim_hdr2 != nullptr && !sim_overlay && !spilled_lds && ch.n_ins <= MGL_SIM2_CAP && ch.n_rem <= MGL_SIM2_CAP && phase == P_SIM) {
			/* the second pass hands over its FINAL re-simulation too: k_sim's second launch reads the lists straight from this
			 * neighbour's scratch slot into LDS and puts two wavefronts on them (here they sit in global memory in front of one
			 * wavefront: the long-list neighbours were the slowest of the pass) */
			if (lane == 0) {
				big.sim_slot2[j] = slot;
				big.sim_hdr2[j] = make_uint4(ch.n_ins, ch.n_rem, (uint32_t)(uint64_t)ch.direct, (uint32_t)((uint64_t)ch.direct >> 32));
			}
			sim_deferred = true;
			phase = P_OUT;
			continue;
		}
		if (MODE == MGL_NBR_REST && phase == P_SIM) {
			/* the second half ends here (always: it holds no re-simulation code of its own): the lists go to k_sim, which puts several wavefronts on the
			 * contexts of one neighbour and writes the cost; journal and counters are written below */
			uint16_t* gk = big.sim_keys + (size_t)j * (2u * big.chg_cap);
			uint32_t* gp = big.sim_pos + (size_t)j * (2u * big.chg_cap);
			for (uint32_t e = lane; e < ch.n_ins; e += 64) { gk[e] = ch.ins_key[e]; gp[e] = ch.ins_pos[e]; }
			for (uint32_t e = lane; e < ch.n_rem; e += 64) { gk[big.chg_cap + e] = ch.rem_key[e]; gp[big.chg_cap + e] = ch.rem_pos[e]; }
			/* no fence: the consumer is a later launch on a stream that waits for this one */
			if (lane == 0) big.sim_hdr[j] = make_uint4(ch.n_ins, ch.n_rem, (uint32_t)(uint64_t)ch.direct, (uint32_t)((uint64_t)ch.direct >> 32));
			sim_deferred = true;
			phase = P_OUT;
			continue;
		}
		if (MODE == MGL_NBR_FULL && phase == P_SIM) {
			int64_t r;
			if (BIG && big.lds_cache && ch.n_ins <= MGL_BIG_CAP && ch.n_rem <= MGL_BIG_CAP) {
				/* the second pass keeps its lists in global memory; a re-simulation scans them once per touched context,
				 * so it works on a copy in LDS (behind this wavefront's regular area: the second pass's launch reserves it) */
				uint32_t* cp = (uint32_t*)(mine + per_wave_bytes);
				Changes cl = ch;
				cl.ins_pos = cp; cl.rem_pos = cp + MGL_BIG_CAP;
				cl.ins_key = (uint16_t*)(cp + 2u * MGL_BIG_CAP); cl.rem_key = cl.ins_key + MGL_BIG_CAP;
				for (uint32_t e = lane; e < ch.n_ins; e += 64) { cl.ins_pos[e] = ch.ins_pos[e]; cl.ins_key[e] = ch.ins_key[e]; }
				for (uint32_t e = lane; e < ch.n_rem; e += 64) { cl.rem_pos[e] = ch.rem_pos[e]; cl.rem_key[e] = ch.rem_key[e]; }
				wave_sync();
				r = chain_sim(b, cl, T, sim_limit, sim_overlay ? probs : nullptr, lane, &too_many);
			} else {
				r = chain_sim(b, ch, T, sim_limit, (MODE != MGL_NBR_REST && sim_overlay) ? probs : nullptr, lane, &too_many);
			}
			wave_sync();
			if (too_many) { phase = P_OUT; continue; }
			if (sim_overlay) { prof_mark(prof, 6, lane); phase = P_TOPK; } /* repair: model overridden by the re-simulated values */
			else { delta = r; prof_mark(prof, 4, lane); phase = P_OUT; }
		}
		if (MODE != MGL_NBR_REST && phase == P_TOPK) {
			walk_from_state(tw, nb);
			walk_window(tw, c, b.slab, lane);
			mgl_pk picked;
			const bool ok = pick_from_top_k<(MODE == MGL_NBR_PICK ? 4 : 1)>(c, tw, probs, T, lencost, pick_inc, pick_best, rng, lane, &picked);
			if (MODE == MGL_NBR_PICK) {
				if (lane == 0) pickrec[j] = make_uint4((uint32_t)picked, (uint32_t)(picked >> 32), rng.n, ok ? 1u : 0u);
				return;
			}
			if (pick_is_mutation) {
				/* evaluated again by the one-kernel form (look-ahead): the second pass, should this neighbour reach it, takes the
				 * mutation's pick from the record like every other neighbour of a split step */
				if (!BIG && LIST && pickrec != nullptr && lane == 0)
					pickrec[j] = make_uint4((uint32_t)picked, (uint32_t)(picked >> 32), rng.n, ok ? 1u : 0u);
				if (!ok) { generate_failed = true; phase = P_OUT; continue; }
				m_first = picked;
				journal_set(jn, pos, first, m_first, lane);
				pick_is_mutation = false;
				prof_mark(prof, 2, lane); /* top-K */
				if (c.diag_stop == 3 || (c.diag_stop >= 31 && c.diag_stop <= 39)) { if (lane == 0) { out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = (uint32_t)m_first; } return; }
			} else {
				picked_pk = ok ? picked : pick_inc;
				have_pick = true;
				prof_mark(prof, 7, lane); /* repair: top-K */
			}
			win.base = 0xFFFFFFFFu;
			phase = P_WALK;
		}
		if (MODE != MGL_NBR_PICK && phase == P_WALK) {
			/* ---- two-pointer walk over neighbour packets (nb) and base packets (bs) */
			bool request_pick = false;
			while (nb.pos < c.n || bs.pos < c.n) {
				if (ch.overflow || jn.overflow || too_many || ++guard > (BIG ? (1u << 20) : 4096u)) { ch.overflow = true; break; }
				if (!first_packet && !have_pick && nb.pos == bs.pos && count >= 3) {
					const bool same_ctx = nb.ctx_state == bs.ctx_state;
					const bool same_d = nb.dists[0] == bs.dists[0] && nb.dists[1] == bs.dists[1] && nb.dists[2] == bs.dists[2] &&
					                    nb.dists[3] == bs.dists[3];
					if (same_ctx && wsoft == 0xFFFFFFFFu) wsoft = nb.pos;
					if (same_ctx && same_d) break; /* the rest of the file is coded identically */
					if (same_ctx && nb.ctx_state < 7) {
						/* plain literals up to the next special packet code identically: skip them */
						uint32_t sx = uni(sp_find_next(b, nb.pos));
						if (sx == MGL_POS_INF || sx > c.n) sx = c.n;
						if (sx > nb.pos) {
							const uint32_t cs = lit_steps(nb.ctx_state, sx - nb.pos);
							nb.pos = bs.pos = sx; nb.ctx_state = bs.ctx_state = cs;
							count = 8;
							continue;
						}
					}
				}
				if (walked > MGL_MAX_WALK && !have_pick) { walk_long = true; break; }
				if ((have_pick || nb.pos <= bs.pos) && nb.pos < c.n) {
					/* ---- next neighbour packet: repair rules of packet_slab_neighbour.c:82-117 */
					const uint32_t p = nb.pos;
					win_cover(win, c, b.slab, p, lane);
					mgl_pk pk, old = 0;
					if (have_pick) {
						pk = picked_pk; old = resume_old; have_pick = false; /* back from the top-K pick for this packet */
					} else if (first_packet) {
						pk = m_first; /* :169 the mutated packet is coded as is */
					} else {
						if (count < 8) count++;
						old = (second_set && p == pos + 1) ? m_second : win_pk(win, p);
						pk = old;
						uint32_t type = mgl_pk_type(pk);
						if (type == MGL_SHORT_REP || (type == MGL_LITERAL && count < 4)) {
							const bool same = nb.dists[0] < p && win_byte(win, p) == c.data[p - nb.dists[0] - 1];
							if (same) { if (count < 4) pk = MGL_PK_SHORT_REP; }
							else pk = MGL_PK_LITERAL;
						}
						type = mgl_pk_type(pk);
						if (type == MGL_LONG_REP) {
							const uint32_t len = mgl_pk_len(pk);
							uint32_t idx = mgl_pk_dist(pk);
							walk_from_state(tw, nb);
							bool ok = long_rep_ok(c, tw, idx, len, lane);
							for (uint32_t i = 0; i < 4 && !ok; i++) { idx = i; ok = long_rep_ok(c, tw, idx, len, lane); }
							pk = mgl_pack(MGL_LONG_REP, idx, len);
							if (!ok) {
								if (++npicks > MGL_MAX_REPAIR_PICKS) { walk_long = true; break; }
								pick_best = (nbr_draw(rng) % 4u) == 0;
								/* the model at p needs every base packet that starts before p priced in */
								while (bs.pos < p && !ch.overflow) {
									win_cover(win, c, b.slab, bs.pos, lane);
									const mgl_pk bpk = win_pk(win, bs.pos);
									mgl_plan bpl;
									plan_at(c, bs, mgl_pk_type(bpk), mgl_pk_dist(bpk), mgl_pk_len(bpk), win_byte(win, bs.pos), bpl);
									changes_add<false>(ch, bpl, bs.pos, lane);
									mgl_advance(&bs, mgl_pk_type(bpk), mgl_pk_dist(bpk), mgl_pk_len(bpk));
								}
								if (ch.overflow) break;
								if (MODE == MGL_NBR_REST && big.cont != nullptr) {
									/* the second half holds no top-K code: the walk as it stands goes to a continuation record (state, journal;
									 * the change lists to the slot's list area) and the second pass takes it up at this pick.  Saved here, where
									 * everything is live anyway, so that the walk loop carries no value for it */
									uint32_t slot2 = 0;
									if (lane == 0) {
										atomicAdd(big.spill_ctr + 1, 1u); /* repair picks this step */
										slot2 = atomicAdd(todo_count, 1u);
										todo[slot2] = j;
										out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = 0; out.win[2u * j] = pos; out.win[2u * j + 1u] = MGL_WIN_NONE;
									}
									slot2 = uni(slot2);
									if (slot2 < big.slots) { /* else: no scratch slot, the second pass sends it on to the full walk */
										uint16_t* gik = big.ins_key + (size_t)slot2 * big.cap; uint32_t* gip = big.ins_pos + (size_t)slot2 * big.cap;
										uint16_t* grk = big.rem_key + (size_t)slot2 * big.cap; uint32_t* grp = big.rem_pos + (size_t)slot2 * big.cap;
										for (uint32_t i = lane; i < ch.n_ins; i += 64) { gik[i] = ch.ins_key[i]; gip[i] = ch.ins_pos[i]; }
										for (uint32_t i = lane; i < ch.n_rem; i += 64) { grk[i] = ch.rem_key[i]; grp[i] = ch.rem_pos[i]; }
										uint32_t* rec = big.cont + (size_t)slot2 * MGL_CONT_WORDS;
										if (lane < jn.count) {
											rec[48u + lane] = jn.pos[lane];
											reinterpret_cast<mgl_pk*>(rec + 112u)[lane] = jn.old[lane];
											reinterpret_cast<mgl_pk*>(rec + 240u)[lane] = jn.neu[lane];
										}
										if (lane == 0) {
											rec[1] = j; rec[2] = pos;
											rec[3] = nb.pos; rec[4] = nb.ctx_state; rec[5] = nb.dists[0]; rec[6] = nb.dists[1]; rec[7] = nb.dists[2]; rec[8] = nb.dists[3];
											rec[9] = bs.pos; rec[10] = bs.ctx_state; rec[11] = bs.dists[0]; rec[12] = bs.dists[1]; rec[13] = bs.dists[2]; rec[14] = bs.dists[3];
											rec[15] = count; rec[16] = walked; rec[17] = npicks; rec[18] = wsoft; rec[19] = taint; rec[20] = dep;
											rec[21] = rng.n; rec[22] = guard; rec[23] = jn.count; rec[24] = ch.n_ins; rec[25] = ch.n_rem;
											rec[26] = (pick_best ? 1u : 0u) | (second_set ? 2u : 0u);
											rec[28] = (uint32_t)pk; rec[29] = (uint32_t)(pk >> 32);       /* the incumbent of the pick */
											rec[30] = (uint32_t)old; rec[31] = (uint32_t)(old >> 32);     /* the packet the repair found at p */
											rec[32] = (uint32_t)m_second; rec[33] = (uint32_t)(m_second >> 32);
											rec[34] = (uint32_t)(uint64_t)ch.direct; rec[35] = (uint32_t)((uint64_t)ch.direct >> 32);
											rec[0] = MGL_CONT_MAGIC; /* the reader is a later launch */
										}
									}
									return;
								}
								resume_old = old; pick_pos = p; pick_inc = pk;
								request_pick = true;
								if (!BIG && lane == 0) atomicAdd(big.spill_ctr + 1, 1u); /* repair picks this step */
								break; /* -> P_MODEL, then back here with have_pick */
							}
						}
					}
					if (!first_packet && pk != old) {
						const mgl_pk base_old = (second_set && p == pos + 1) ? uni64(b.slab[p]) : old;
						journal_set(jn, p, base_old, pk, lane);
						wsoft = 0xFFFFFFFFu; /* a changed packet (a SHORT_REP the repair turned into a literal, say): the soft window reaches behind it */
					}
					first_packet = false;
					walked++;
					const uint32_t ntype = mgl_pk_type(pk), ndist = mgl_pk_dist(pk), nlen = mgl_pk_len(pk);
					/* which rep distances this packet reads / pushes: bit k of taint = distance k comes from before the window */
					if (ntype == MGL_SHORT_REP || ntype == MGL_LONG_REP) {
						/* a rep packet: the soft window reaches at least to behind it -- unless it is the base's own packet at this
						 * position reading a slot that holds the same distance in both walks (the move has no part in what it codes) */
						const uint32_t slot = ntype == MGL_SHORT_REP ? 0u : ndist;
						const bool same_read = bs.pos == p && win_pk(win, p) == pk && mgl_dist_at(&nb, slot) == mgl_dist_at(&bs, slot);
						if (!same_read) wsoft = 0xFFFFFFFFu;
						dep |= ntype == MGL_SHORT_REP ? (taint & 1u) : ((taint >> ndist) & 1u);
					}
					if (ntype == MGL_MATCH) taint = (taint << 1) & 0xFu;
					else if (ntype == MGL_LONG_REP) taint = (taint & ~((2u << ndist) - 1u)) | ((taint & ((1u << ndist) - 1u)) << 1) | ((taint >> ndist) & 1u);
					/* the base packet at the same position: identical coding cancels.  Which events a
					 * packet produces depends on the packet, ctx_state, the position and -- for a literal
					 * after a match -- the byte at rep distance 0; decided before anything is planned */
					bool cancelled = false;
					if (bs.pos == p && win_pk(win, p) == pk && nb.ctx_state == bs.ctx_state) {
						cancelled = true;
						if (ntype == MGL_LITERAL && nb.ctx_state >= 7) {
							const uint32_t mn = nb.dists[0] < p ? c.data[p - nb.dists[0] - 1] : 0u;
							const uint32_t mb = bs.dists[0] < p ? c.data[p - bs.dists[0] - 1] : 0u;
							cancelled = mn == mb;
						}
					}
					if (cancelled) {
						mgl_advance(&bs, ntype, ndist, nlen);
					} else {
						if (bs.pos == p) {
							const mgl_pk bpk = win_pk(win, p);
							const uint32_t btype = mgl_pk_type(bpk), bdist = mgl_pk_dist(bpk), blen = mgl_pk_len(bpk);
							mgl_plan bpl;
							plan_at(c, bs, btype, bdist, blen, win_byte(win, p), bpl);
							changes_add<false>(ch, bpl, p, lane);
							mgl_advance(&bs, btype, bdist, blen);
						}
						mgl_plan npl;
						plan_at(c, nb, ntype, ndist, nlen, win_byte(win, p), npl);
						changes_add<true>(ch, npl, p, lane);
					}
					mgl_advance(&nb, ntype, ndist, nlen);
				} else {
					/* ---- a base packet the neighbour has already passed over: its events go away */
					win_cover(win, c, b.slab, bs.pos, lane);
					if (literal_run_events<false>(c, ch, win, bs, (nb.pos < c.n ? nb.pos : c.n) - bs.pos, lane)) continue;
					const mgl_pk bpk = win_pk(win, bs.pos);
					mgl_plan bpl;
					plan_at(c, bs, mgl_pk_type(bpk), mgl_pk_dist(bpk), mgl_pk_len(bpk), win_byte(win, bs.pos), bpl);
					changes_add<false>(ch, bpl, bs.pos, lane);
					mgl_advance(&bs, mgl_pk_type(bpk), mgl_pk_dist(bpk), mgl_pk_len(bpk));
				}
			}
			if (walk_long) { phase = P_OUT; continue; }
			if (request_pick) { prof_mark(prof, 3, lane); phase = P_MODEL; continue; }
			prof_mark(prof, 3, lane); /* window walk */
			if (c.diag_stop == 4) { if (lane == 0) { out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = ch.n_ins + ch.n_rem; } return; }
			if (ch.overflow || jn.overflow || too_many) { phase = P_OUT; continue; }
			wend = nb.pos;
			sim_limit = MGL_POS_INF; sim_overlay = false; phase = P_SIM;
			continue;
		}
		if (phase == P_OUT) break;
	}

	if (generate_failed) { /* no candidate at the target (main.c:81-84 retries those) */
		if (lane == 0) { out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = 0; out.win[2u * j] = pos; out.win[2u * j + 1u] = MGL_WIN_NONE; }
		return;
	}
	if (jn.overflow) { /* same rule as the full-walk path and the oracle: dropped */
		if (lane == 0) { out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = walked; out.win[2u * j] = pos; out.win[2u * j + 1u] = MGL_WIN_DROPPED; }
		return;
	}
	if (walk_long || (ch.list_full && ch.cap == big.cap && !jn.overflow)) {
		/* more inserted or removed events than the second pass's lists hold (MGL_BIG_CAP), or more than MGL_MAX_WALK packets
		 * visited before the walks met again: dropped, like a neighbour with too long a journal; the oracle counts the same */
		if (lane == 0) { out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = walked; out.win[2u * j] = pos; out.win[2u * j + 1u] = MGL_WIN_DROPPED; }
		return;
	}
	if (ch.overflow || too_many) {
		/* does not fit the change lists: hand it to the next pass (BIG, then the full-walk kernel) */
		if (lane == 0) {
			const uint32_t slot2 = atomicAdd(todo_count, 1u);
			todo[slot2] = j;
			if (BIG) atomicAdd((unsigned long long*)&ctl->fallback_nbrs, 1ull);
			out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = 0; out.win[2u * j] = pos; out.win[2u * j + 1u] = MGL_WIN_NONE;
		}
		return;
	}
	const uint64_t total = (uint64_t)((int64_t)ctl->rebuild_cost + delta + ch.direct); /* rebuild_cost: exact cost of the base */
	uint32_t nd = 0;
	for (uint32_t i = 0; i < jn.count; i++) {
		if (jn.old[i] == jn.neu[i]) continue;
		if (lane == 0) {
			out.dpos[(size_t)j * MGL_MAX_DIFFS + nd] = jn.pos[i];
			out.dold[(size_t)j * MGL_MAX_DIFFS + nd] = jn.old[i];
			out.dnew[(size_t)j * MGL_MAX_DIFFS + nd] = jn.neu[i];
		}
		nd++;
	}
	if (lane == 0) { if (!sim_deferred) out.cost[j] = total; out.ndiffs[j] = nd; out.walked[j] = walked; out.win[2u * j] = pos; out.win[2u * j + 1u] = wend; out.win2[j] = (wsoft < wend ? wsoft : wend) | (dep << 31); }
	if (prof_acc && lane == 0) /* diagnostic: wave lifetime (40 bits) | change events (12 bits) | packets walked (12 bits) */
		prof_acc[32 + j] = ((__builtin_readcyclecounter() - t_begin) & 0xFFFFFFFFFFull) |
		                   ((unsigned long long)((ch.n_ins + ch.n_rem) & 0xFFFu) << 40) | ((unsigned long long)(walked & 0xFFFu) << 52);
}

template <bool BIG, int MODE, bool LIST = false>
__global__ void __launch_bounds__((MODE == MGL_NBR_PICK ? 512 : 64), (MODE == MGL_NBR_FULL ? (BIG ? 1 : MGL_NBR_WAVES_PER_SIMD) : (MODE == MGL_NBR_REST ? MGL_REST_WAVES_PER_SIMD : 4))) k_neighbours2(DevCtx c, Base2 b, Control* ctl, uint64_t seed,
                                                     uint64_t step_override, uint32_t K, NbrOut out, uint32_t per_wave_bytes,
                                                     uint32_t* todo, uint32_t* todo_count, unsigned long long* prof_acc,
                                                     BigScratch big, uint4* pickrec, uint32_t j_base, uint32_t j_end, uint4* pickstate)
{
	/* Which form of the regular launch runs (split: pick + rest + k_sim, or the one-kernel form) is the host's
	 * choice per block of steps (mgl_sa_run reads the device's recommendation, Control::nbr_single): only the
	 * chosen form is launched.  The second pass is launched with a small grid whatever its list holds -- the
	 * host does not know the count -- and strides over it; with an empty list a workgroup costs one load. */
	const uint32_t waves = blockDim.x >> 6;
	if (BIG && blockIdx.x * waves + ((LIST && big.todo_first != nullptr) ? *big.todo_first : 0u) >= *big.todo_in_count) return;
	if (!BIG && LIST && blockIdx.x * waves >= *big.la_count) return;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint16_t* T = (uint16_t*)smem;
	/* 4 KiB as 256 16-byte units (the table is hipMalloc-aligned, T sits at the start of the LDS block); the second
	 * half of the split form prices nothing (its re-simulation is k_sim's) and has no table: 4 KiB less per workgroup */
	if (MODE == MGL_NBR_PICK && MGL_PICK_T_GLOBAL) {
		T = const_cast<uint16_t*>(c.cost_tbl); /* read through the vector cache: 4 KiB less LDS per pick workgroup */
	} else if (MODE != MGL_NBR_REST) {
		for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) reinterpret_cast<uint4*>(T)[i] = reinterpret_cast<const uint4*>(c.cost_tbl)[i];
		__syncthreads();
	}
	const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
	if (BIG) {
		const uint32_t first = (LIST && big.todo_first != nullptr) ? *big.todo_first : 0u;
		const uint32_t n = *big.todo_in_count > first ? *big.todo_in_count - first : 0u;
		for (uint32_t unit = blockIdx.x * waves + wid; unit < n; unit += gridDim.x * waves)
			nbr2_one<BIG, MODE, LIST>(c, b, ctl, seed, step_override, K, out, per_wave_bytes, todo, todo_count, prof_acc, big, pickrec, j_base, j_end, pickstate, smem, T, unit, lane, wid);
	} else if (LIST) {
		const uint32_t n = *big.la_count;
		for (uint32_t unit = blockIdx.x * waves + wid; unit < n; unit += gridDim.x * waves)
			nbr2_one<BIG, MODE, true>(c, b, ctl, seed, step_override, K, out, per_wave_bytes, todo, todo_count, prof_acc, big, pickrec, 0u, K, pickstate, smem, T, unit, lane, wid);
	} else {
		nbr2_one<BIG, MODE, false>(c, b, ctl, seed, step_override, K, out, per_wave_bytes, todo, todo_count, prof_acc, big, pickrec, j_base, j_end, pickstate, smem, T,
		                    blockIdx.x * waves + wid, lane, wid);
	}
}


/* ================================================================== k_sim
 *
 * The re-simulation of the split launch's second half, as its own launch: MGL_SIM_WAVES wavefronts
 * per neighbour share its touched contexts (one context per lane, as in chain_sim), so a neighbour
 * with 150 contexts is one trip instead of three.  The second half's duration is its slowest
 * wavefront (the mean one lives a third of that), and the slowest are the ones with many contexts.
 * Small kernel: no walk state, no journal -- 64 VGPRs less than the second half. */
#ifndef MGL_SIM_WAVES
#define MGL_SIM_WAVES 2u   /* the regular launch: thousands of neighbours, a hundred contexts each */
#endif
#ifndef MGL_SIM_WAVES_LIST
#define MGL_SIM_WAVES_LIST 8u /* the second pass's few neighbours (long lists, hundreds of contexts): one trip over the contexts */
#endif
#define MGL_SIM_WAVES_MAX 8u
struct SimShared {
	uint16_t* T;
	uint32_t* dyn;
	unsigned long long* sum;
	uint32_t* nu_many; /* [0] distinct contexts, [1] too many */
};
template <bool FROM_BIG, bool COUNT>
__device__ __forceinline__ void sim_one(const DevCtx& c, const Base2& b, Control* ctl, const NbrOut& out, const BigScratch& big, const uint4* hdrs,
                                        uint32_t j, uint32_t* todo, uint32_t* todo_count, const SimShared& sh, unsigned long long* traffic_ctr)
{
	const uint4 hdr = hdrs[j];
	if (hdr.x == 0xFFFFFFFFu) return; /* failed, dropped or handed on: its cost is written (uniform over the workgroup) */
	const uint32_t cap = FROM_BIG ? MGL_SIM2_CAP : big.chg_cap;
	const uint32_t slot = FROM_BIG ? big.sim_slot2[j] : 0u;
	uint32_t* s_bits = sh.dyn;
	uint32_t* s_pos = sh.dyn + ((((c.L.total + 31u) >> 5) + 3u) & ~3u);
	uint16_t* s_key = (uint16_t*)(s_pos + 2u * cap);
	uint16_t* s_uctx = s_key + 2u * cap;
	const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
	Changes ch;
	ch.n_ins = hdr.x; ch.n_rem = hdr.y;
	ch.ins_pos = s_pos; ch.rem_pos = s_pos + cap;
	ch.ins_key = s_key; ch.rem_key = s_key + cap;
	ch.uctx = s_uctx; ch.ctxbits = s_bits;
	ch.cap = cap; ch.uctx_cap = 2 * cap;
	ch.nbitwords = (c.L.total + 31u) >> 5;
	ch.direct = 0; ch.dbg = nullptr; ch.diag = c.diag_stop; ch.overflow = false; ch.list_full = false;
	__syncthreads(); /* the previous neighbour of a striding workgroup is done with the LDS */
	if (FROM_BIG) {
		const uint16_t* ik = big.ins_key + (size_t)slot * big.cap; const uint32_t* ip = big.ins_pos + (size_t)slot * big.cap;
		const uint16_t* rk = big.rem_key + (size_t)slot * big.cap; const uint32_t* rp = big.rem_pos + (size_t)slot * big.cap;
		for (uint32_t e = threadIdx.x; e < ch.n_ins; e += blockDim.x) { s_key[e] = ik[e]; s_pos[e] = ip[e]; }
		for (uint32_t e = threadIdx.x; e < ch.n_rem; e += blockDim.x) { s_key[cap + e] = rk[e]; s_pos[cap + e] = rp[e]; }
	} else {
		const uint16_t* gk = big.sim_keys + (size_t)j * (2u * cap);
		const uint32_t* gp = big.sim_pos + (size_t)j * (2u * cap);
		for (uint32_t e = threadIdx.x; e < ch.n_ins; e += blockDim.x) { s_key[e] = gk[e]; s_pos[e] = gp[e]; }
		for (uint32_t e = threadIdx.x; e < ch.n_rem; e += blockDim.x) { s_key[cap + e] = gk[cap + e]; s_pos[cap + e] = gp[cap + e]; }
	}
	__syncthreads();
	if (wid == 0) {
		bool too_many = false;
		const uint32_t nu = chain_list(ch, lane, &too_many);
		if (lane == 0) { sh.nu_many[0] = nu; sh.nu_many[1] = too_many ? 1u : 0u; }
	}
	__syncthreads();
	if (sh.nu_many[1]) { /* more distinct contexts than the list holds: the late second pass re-simulates it inline */
		if (threadIdx.x == 0) {
			const uint32_t slot2 = atomicAdd(todo_count, 1u);
			todo[slot2] = j;
			out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = 0; out.win[2u * j + 1u] = MGL_WIN_NONE;
		}
		return;
	}
	if (c.diag_stop == 41) { if (threadIdx.x == 0) out.cost[j] = sh.nu_many[0]; return; } /* diagnostic: listing only */
	uint32_t traffic = 0;
	const int64_t mine = chain_sim_contexts(b, ch, sh.T, MGL_POS_INF, nullptr, lane, wid * 64u, blockDim.x, sh.nu_many[0], COUNT ? &traffic : nullptr);
	const uint64_t u = wave_sum64((uint64_t)mine);
	if (lane == 0) sh.sum[wid] = u;
	if (COUNT) {
		/* + the change lists this workgroup brought into LDS (6 bytes per event), once per neighbour */
		const uint64_t tb = wave_sum64((uint64_t)traffic) + (wid == 0 ? 6ull * (ch.n_ins + ch.n_rem) + 16ull : 0ull);
		if (lane == 0) atomicAdd(traffic_ctr, (unsigned long long)tb);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint64_t d = 0;
		for (uint32_t w = 0; w < (blockDim.x >> 6); w++) d += sh.sum[w];
		const int64_t direct = (int64_t)((uint64_t)hdr.z | ((uint64_t)hdr.w << 32));
		out.cost[j] = (uint64_t)((int64_t)ctl->rebuild_cost + (int64_t)d + direct);
	}
}
/* list == nullptr: the regular launch, workgroup x = neighbour j_base + x with header sim_hdr.  list != nullptr: the
 * second pass's neighbours (headers in sim_hdr2), a small grid striding over the list. */
/* COUNT: the same kernel adding up the bytes of chain data and change lists it reads (traffic_ctr[0]) and its launches that had
 * work (traffic_ctr[1]); bench.py runs a few steps with it after its timed region */
template <bool COUNT>
__global__ void __launch_bounds__(64 * MGL_SIM_WAVES_MAX, 8) k_sim(DevCtx c, Base2 b, Control* ctl, NbrOut out, BigScratch big, uint32_t j_base, uint32_t j_end,
                                                           uint32_t* todo, uint32_t* todo_count, const uint32_t* list, const uint32_t* list_count,
                                                           unsigned long long* traffic_ctr)
{
	if (list ? blockIdx.x >= *list_count : j_base + blockIdx.x >= j_end) return;
	__shared__ __attribute__((aligned(16))) uint16_t T[2048];
	/* sized by the launch: one bit per context, then the two lists (positions, keys) and the context list */
	extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
	__shared__ unsigned long long s_sum[MGL_SIM_WAVES_MAX];
	__shared__ uint32_t s_nm[2];
	if (!list && big.sim_hdr[j_base + blockIdx.x].x == 0xFFFFFFFFu) return; /* before the table load: most launches of a bulk-free step's tail */
	for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) reinterpret_cast<uint4*>(T)[i] = reinterpret_cast<const uint4*>(c.cost_tbl)[i];
	SimShared sh;
	sh.T = T; sh.dyn = s_dyn; sh.sum = s_sum; sh.nu_many = s_nm;
	if (list) {
		const uint32_t n = *list_count;
		for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
			/* look-ahead: an entry of the speculative launch whose neighbour was evaluated again -- the second pass skipped it, its header is stale */
			if (big.la_mark != nullptr && i < *big.la_spec_count && big.la_mark[list[i]]) continue;
			sim_one<true, COUNT>(c, b, ctl, out, big, big.sim_hdr2, list[i], todo, todo_count, sh, traffic_ctr);
		}
	} else {
		sim_one<false, COUNT>(c, b, ctl, out, big, big.sim_hdr, j_base + blockIdx.x, todo, todo_count, sh, traffic_ctr);
	}
}


/* ================================================================== packets before every block of 4 096 positions (stratified targets) */
__global__ void __launch_bounds__(64) k_rank_blocks(const uint64_t* onwalk, uint32_t nw0, uint32_t* pre, uint32_t nblk)
{
	const uint32_t blk = blockIdx.x, lane = threadIdx.x;
	if (blk >= nblk) return;
	const uint32_t w = blk * 64u + lane;
	unsigned long long v = w < nw0 ? (unsigned long long)__popcll(onwalk[w]) : 0ull;
	v = wave_sum64(v);
	if (lane == 0) pre[blk + 1u] = (uint32_t)v;
}
__global__ void __launch_bounds__(1024) k_rank_scan(uint32_t* pre, uint32_t nblk)
{
	__shared__ uint32_t s_w[16], s_base;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
	if (tid == 0) { s_base = 0; pre[0] = 0; }
	__syncthreads();
	for (uint32_t base = 0; base < nblk; base += blockDim.x) {
		const uint32_t i = base + tid;
		const uint32_t v = i < nblk ? pre[i + 1u] : 0u;
		uint32_t incl = v;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if ((int)lane >= o) incl += t; }
		if (lane == 63) s_w[wid] = incl;
		__syncthreads();
		uint32_t bsum = s_base;
		for (uint32_t q = 0; q < wid; q++) bsum += s_w[q];
		if (i < nblk) pre[i + 1u] = bsum + incl;
		__syncthreads();
		if (tid == blockDim.x - 1u) s_base = bsum + incl;
		__syncthreads();
	}
}

/* the step's targets: one wavefront per neighbour (mgl_device.h:stratified_target; draw 0 of the neighbour's stream picks the
 * packet inside its slice).  A kernel of its own: in the pick kernel, sixteen wavefronts per CU, the same search was a
 * dozen dependent loads at the head of every wavefront's critical path */
__global__ void __launch_bounds__(256) k_targets(DevCtx c, const uint64_t* onwalk, uint32_t nw0, const Control* ctl, uint64_t seed, uint64_t step_override,
                                                 uint32_t K, uint32_t* tgt)
{
	const uint32_t j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (j >= K) return;
	const uint64_t gstep = step_override != ~0ull ? step_override : ctl->gstep;
	/* packets on the walk: the last prefix sum (Control::packets says the same once the accept is through; this kernel may run beside it) */
	const uint32_t t = stratified_target(onwalk, nw0, c.strat_pre, c.strat_nblk, c.strat_pre[c.strat_nblk], K, j, mgl_rng_draw(mgl_rng_key(seed, gstep, j), 0), lane);
	if (lane == 0) tgt[j] = t;
}
