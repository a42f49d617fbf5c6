/*
 * mgl_kernels3.hip -- accepting a neighbour without rebuilding the base.
 *
 *   k_apply_walk    one wavefront re-runs the two-pointer walk of the winning neighbour (its
 *                   packets are known from the journal: no RNG, no top-K) against the old base:
 *                   updates the on-walk / special bitmaps and the state records of every special
 *                   packet it passes, collects the inserted / removed events, lists the touched
 *                   probability contexts, then writes the journal into the slab.
 *   k_apply_chains  one workgroup per touched context: re-simulates the context from its first
 *                   change, splices the new entries into its chain (the tail moves by the net
 *                   insert/remove count), and patches that context's value in every dense
 *                   checkpoint between the change and the point where it re-joins the old
 *                   trajectory.
 * Anything that does not fit (event lists, span buffers, chain capacity) raises
 * Control::apply_failed and the same step falls back to k_build, which rebuilds everything
 * from the slab; both routes produce identical structures (tests/test_gpu_incremental.py).
 */
#include "mgl_base2.h"

#define MGL_APPLY_CAP 8192u   /* inserted / removed events per accepted neighbour */
#define MGL_SUB_CAP 1024u     /* per context */
#define MGL_SPAN_CAP 4096u    /* rewritten chain entries per context */
#define MGL_PIECE_CAP 160u
#define MGL_APPLY_THREADS 1024u
#define MGL_APPLY_WIN 2048u

struct ApplyBuf {
	uint32_t* hdr;      /* [0] n_ins [1] n_rem [2] n_tctx [3] first journal position */
	uint16_t* ins_key;  /* ctx | bit << 15 */
	uint32_t* ins_pos;
	uint16_t* rem_key;
	uint32_t* rem_pos;
	uint16_t* tctx;
	/* chain rewrites are planned per context (k_apply_chains) and carried out as flat lists of copy
	 * jobs by the whole device (k_apply_jobs): pass B saves the old entries, pass C writes the new */
	uint32_t* scratch_pos; /* space 1: old chain entries saved by pass B (scratch_cap entries) */
	uint16_t* scratch_ev;
	uint32_t scratch_cap;
	uint32_t* span_pos;    /* space 2: re-simulated entries (span_cap entries) */
	uint16_t* span_ev;
	uint32_t span_cap;
	uint4* jobs_b;         /* { src offset, dst offset, count, src space | dst space << 8 } */
	uint4* jobs_c;
	uint32_t job_cap;
};
/* hdr: [0] n_ins [1] n_rem [2] n_tctx [3] first journal position [4] jobs B [5] jobs C [6] span entries used
 * [7] scratch entries used */
#define MGL_JOB_CHUNK 4096u
#define MGL_SPACE_CHAIN 0u
#define MGL_SPACE_SCRATCH 1u
#define MGL_SPACE_SPAN 2u
#define MGL_SPACE_SHIFT 3u /* job kind: in-place shift of one chunk of a chain's tail */

__device__ __forceinline__ void bit_write(uint64_t* arr, uint32_t pos, bool on)
{
	const uint64_t m = 1ull << (pos & 63u);
	uint64_t v = arr[pos >> 6];
	v = on ? (v | m) : (v & ~m);
	arr[pos >> 6] = v;
}
/* special bitmap with its two summary levels (lane 0 only) */
__device__ __forceinline__ void special_write(const Base2& b, uint32_t pos, bool on)
{
	const uint32_t w = pos >> 6;
	uint64_t v = b.sp0[w];
	const uint64_t m = 1ull << (pos & 63u);
	const uint64_t nv = on ? (v | m) : (v & ~m);
	if (nv == v) return;
	b.sp0[w] = nv;
	if ((v != 0) == (nv != 0)) return;
	const uint32_t u = w >> 6;
	uint64_t v1 = b.sp1[u];
	const uint64_t m1 = 1ull << (w & 63u);
	const uint64_t nv1 = nv ? (v1 | m1) : (v1 & ~m1);
	b.sp1[u] = nv1;
	if ((v1 != 0) == (nv1 != 0)) return;
	const uint32_t x = u >> 6;
	const uint64_t m2 = 1ull << (u & 63u);
	uint64_t v2 = b.sp2[x];
	b.sp2[x] = nv1 ? (v2 | m2) : (v2 & ~m2);
}

__global__ void __launch_bounds__(64) k_apply_walk(DevCtx c, Base2 b, Control* ctl, NbrOut out, ApplyBuf ab)
{
	__shared__ uint32_t s_jpos[MGL_MAX_DIFFS];
	__shared__ mgl_pk s_jnew[MGL_MAX_DIFFS];
	__shared__ uint32_t s_ctxbits[512]; /* up to 16384 contexts */
	const uint32_t lane = threadIdx.x;
	if (lane == 0) { ctl->mod_lo = MGL_POS_INF; ctl->mod_hi = 0u; }
	if (!ctl->accepted_flag) return;
	const uint32_t winner = ctl->winner;
	const uint32_t nd = out.ndiffs[winner];
	if (lane < nd) {
		s_jpos[lane] = out.dpos[(size_t)winner * MGL_MAX_DIFFS + lane];
		s_jnew[lane] = out.dnew[(size_t)winner * MGL_MAX_DIFFS + lane];
	}
	for (uint32_t i = lane; i < 512; i += 64) s_ctxbits[i] = 0;
	wave_sync();
	const uint32_t t = s_jpos[0], last_j = s_jpos[nd - 1];

	const bool prof = c.diag_stop == 50u;
	const unsigned long long tw0 = prof ? __builtin_readcyclecounter() : 0ull;
	mgl_wstate nb = uni_state(base_state_at(b, t));
	mgl_wstate bs = nb;
	const unsigned long long tw1 = prof ? __builtin_readcyclecounter() : 0ull;
	Win win; win.base = 0xFFFFFFFFu; win.pk = 0; win.byte = 0;
	uint32_t n_ins = 0, n_rem = 0;
	int32_t dpackets = 0;
	bool failed = false;
	uint32_t ji = 0; /* next journal entry (positions ascending) */
	uint32_t guard = 0;
	while (nb.pos < c.n || bs.pos < c.n) {
		if (++guard > (1u << 20)) { failed = true; break; }
		if (nb.pos == bs.pos) {
			const bool same_ctx = nb.ctx_state == bs.ctx_state;
			const bool same_d = nb.dists[0] == bs.dists[0] && nb.dists[1] == bs.dists[1] && nb.dists[2] == bs.dists[2] &&
			                    nb.dists[3] == bs.dists[3];
			if (same_ctx && same_d && nb.pos > last_j) break;
			if (same_ctx && nb.ctx_state < 7) {
				/* plain literals code identically: skip to the next special or journal position */
				uint32_t s = uni(sp_find_next(b, nb.pos));
				if (s == MGL_POS_INF || s > c.n) s = c.n;
				while (ji < nd && s_jpos[ji] < nb.pos) ji++;
				if (ji < nd && s_jpos[ji] < s) s = s_jpos[ji];
				if (s > nb.pos) {
					const uint32_t cs = lit_steps(nb.ctx_state, s - nb.pos);
					nb.pos = bs.pos = s; nb.ctx_state = bs.ctx_state = cs;
					continue;
				}
			}
		}
		if (nb.pos <= bs.pos && nb.pos < c.n) {
			const uint32_t p = nb.pos;
			win_cover(win, c, b.slab, p, lane);
			const mgl_pk old_at_p = win_pk(win, p);
			while (ji < nd && s_jpos[ji] < p) ji++;
			const mgl_pk pk = (ji < nd && s_jpos[ji] == p) ? s_jnew[ji] : old_at_p;
			const uint32_t ntype = mgl_pk_type(pk), ndist = mgl_pk_dist(pk), nlen = mgl_pk_len(pk);
			const bool paired = bs.pos == p;
			/* identical coding cancels; decided before anything is planned (as in the neighbour kernel) */
			bool cancelled = false;
			if (paired && old_at_p == pk && nb.ctx_state == bs.ctx_state) {
				cancelled = true;
				if (ntype == MGL_LITERAL && nb.ctx_state >= 7) {
					const uint32_t mn = nb.dists[0] < p ? c.data[p - nb.dists[0] - 1] : 0u;
					const uint32_t mb = bs.dists[0] < p ? c.data[p - bs.dists[0] - 1] : 0u;
					cancelled = mn == mb;
				}
			}
			mgl_plan npl, bpl;
			uint32_t btype = 0, bdist = 0, blen = 0;
			if (paired) { btype = mgl_pk_type(old_at_p); bdist = mgl_pk_dist(old_at_p); blen = mgl_pk_len(old_at_p); }
			if (!cancelled) {
				plan_at(c, nb, ntype, ndist, nlen, win_byte(win, p), npl);
				if (paired) plan_at(c, bs, btype, bdist, blen, win_byte(win, p), bpl);
			}
			/* base structures at p: on the new walk; special iff not a literal */
			if (lane == 0) {
				if (!paired) bit_write(b.onwalk, p, true);
				special_write(b, p, ntype != MGL_LITERAL);
			}
			if (ntype != MGL_LITERAL && lane < 8) {
				const uint32_t v = lane == 0 ? nb.ctx_state : lane == 1 ? nb.dists[0] : lane == 2 ? nb.dists[1]
				                 : lane == 3 ? nb.dists[2] : lane == 4 ? nb.dists[3] : 0u;
				b.sp_state[(size_t)p * 8 + lane] = v;
			}
			if (!cancelled) {
				if (n_ins + npl.nev > MGL_APPLY_CAP || (paired && n_rem + bpl.nev > MGL_APPLY_CAP)) { failed = true; break; }
				if (lane < npl.nev) {
					uint32_t ctx, bit;
					mgl_plan_event(&npl, lane, &ctx, &bit);
					ab.ins_key[n_ins + lane] = (uint16_t)(ctx | (bit << 15));
					ab.ins_pos[n_ins + lane] = p;
					atomicOr(&s_ctxbits[ctx >> 5], 1u << (ctx & 31u));
				}
				n_ins += npl.nev;
				if (paired) {
					if (lane < bpl.nev) {
						uint32_t ctx, bit;
						mgl_plan_event(&bpl, lane, &ctx, &bit);
						ab.rem_key[n_rem + lane] = (uint16_t)ctx;
						ab.rem_pos[n_rem + lane] = p;
						atomicOr(&s_ctxbits[ctx >> 5], 1u << (ctx & 31u));
					}
					n_rem += bpl.nev;
				}
			}
			dpackets += paired ? 0 : 1;
			if (paired) mgl_advance(&bs, btype, bdist, blen);
			mgl_advance(&nb, ntype, ndist, nlen);
		} else {
			/* an old packet that is not on the new walk any more */
			const uint32_t q = bs.pos;
			win_cover(win, c, b.slab, q, lane);
			if (bs.ctx_state < 7u) {
				/* a run of plain literals goes seven at a time, nine lanes each (as in the neighbour kernel) */
				const uint32_t o = q - win.base;
				const unsigned long long lit = __ballot(mgl_pk_type(win.pk) == MGL_LITERAL) >> o;
				uint32_t run = ~lit == 0ull ? 64u : (uint32_t)__ffsll((long long)~lit) - 1u;
				const uint32_t limit = (nb.pos < c.n ? nb.pos : c.n) - q;
				run = run < 64u - o ? run : 64u - o;
				run = run < limit ? run : limit;
				if (run >= 2u) {
					const uint32_t take = run < 7u ? run : 7u;
					if (n_rem + 9u * take > MGL_APPLY_CAP) { failed = true; break; }
					const uint32_t i = lane / 9u, slot = lane - i * 9u, p = q + i;
					const bool active = i < take;
					const uint32_t byte = (uint32_t)__shfl((int)win.byte, (int)((p - win.base) & 63u), 64);
					uint32_t prev_byte = 0;
					if (c.L.lc > 0) {
						const uint32_t wprev = (uint32_t)__shfl((int)win.byte, (int)((p - 1u - win.base) & 63u), 64);
						prev_byte = p == 0 ? 0u : (p - 1u >= win.base ? wprev : (uint32_t)c.data[p - 1u]);
					}
					mgl_wstate sv = bs;
					sv.pos = p; sv.ctx_state = lit_steps(bs.ctx_state, i);
					mgl_plan pl;
					mgl_plan_packet(&c.L, &sv, MGL_LITERAL, 0, 1, byte, 0, prev_byte, &pl);
					if (!__ballot(active && pl.nev != 9u)) {
						if (active) {
							uint32_t ctx, bit;
							mgl_plan_event(&pl, slot, &ctx, &bit);
							ab.rem_key[n_rem + i * 9u + slot] = (uint16_t)ctx;
							ab.rem_pos[n_rem + i * 9u + slot] = p;
							atomicOr(&s_ctxbits[ctx >> 5], 1u << (ctx & 31u));
						}
						n_rem += 9u * take;
						if (lane == 0) b.onwalk[q >> 6] &= ~(((take >= 64u ? ~0ull : ((1ull << take) - 1ull))) << (q & 63u)); /* literals are never special */
						dpackets -= (int32_t)take;
						bs.pos += take; bs.ctx_state = lit_steps(bs.ctx_state, take);
						continue;
					}
				}
			}
			const mgl_pk bpk = win_pk(win, q);
			const uint32_t btype = mgl_pk_type(bpk), bdist = mgl_pk_dist(bpk), blen = mgl_pk_len(bpk);
			mgl_plan bpl;
			plan_at(c, bs, btype, bdist, blen, win_byte(win, q), bpl);
			if (n_rem + bpl.nev > MGL_APPLY_CAP) { failed = true; break; }
			if (lane < bpl.nev) {
				uint32_t ctx, bit;
				mgl_plan_event(&bpl, lane, &ctx, &bit);
				ab.rem_key[n_rem + lane] = (uint16_t)ctx;
				ab.rem_pos[n_rem + lane] = q;
				atomicOr(&s_ctxbits[ctx >> 5], 1u << (ctx & 31u));
			}
			n_rem += bpl.nev;
			if (lane == 0) { bit_write(b.onwalk, q, false); special_write(b, q, false); }
			dpackets -= 1;
			mgl_advance(&bs, btype, bdist, blen);
		}
	}
	const unsigned long long tw2 = prof ? __builtin_readcyclecounter() : 0ull;
	wave_sync();
	/* the journal goes into the slab (main.c keeps the mutated slab on accept) */
	if (lane < nd) b.slab[s_jpos[lane]] = s_jnew[lane];
	/* touched contexts, ascending */
	uint32_t nt = 0;
	for (uint32_t wbase = 0; wbase < 512; wbase += 64) {
		const uint32_t word = s_ctxbits[wbase + lane];
		const uint32_t cntl = (uint32_t)__popc(word);
		uint32_t incl = cntl;
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t tmp = (uint32_t)__shfl_up((int)incl, o, 64);
			if ((int)lane >= o) incl += tmp;
		}
		uint32_t at = nt + incl - cntl;
		uint32_t wv = word;
		while (wv) {
			const uint32_t bit = (uint32_t)__ffs((int)wv) - 1u;
			ab.tctx[at++] = (uint16_t)(((wbase + lane) << 5) + bit);
			wv &= wv - 1u;
		}
		nt += (uint32_t)__shfl((int)incl, 63, 64);
	}
	if (lane == 0) {
		ab.hdr[0] = n_ins; ab.hdr[1] = n_rem; ab.hdr[2] = failed ? 0u : nt; ab.hdr[3] = t;
		ab.hdr[4] = 0; ab.hdr[5] = 0; ab.hdr[6] = 0; ab.hdr[7] = 0;
		if (prof) { /* diagnostic: cycles of [state lookup, walk, listing] and walk iterations of this accept */
			const unsigned long long tw3 = __builtin_readcyclecounter();
			ab.hdr[13] = (uint32_t)(tw1 - tw0); ab.hdr[14] = (uint32_t)(tw2 - tw1);
			ab.hdr[15] = ((uint32_t)(tw3 - tw2) & 0xFFFFFu) | (guard << 20); /* hdr[8..12] belong to k_apply_chains' stages */
		}
		ctl->packets = (uint64_t)((int64_t)ctl->packets + dpackets);
		ctl->rebuild_cost = ctl->cur_cost; /* exact cost of the new base */
		if (failed) ctl->apply_failed = 1;
	}
}

/* first packet start at or after checkpoint ck's boundary (new walk), or MGL_POS_INF */
__device__ __forceinline__ uint32_t ckpt_boundary(const Base2& b, uint32_t ck)
{
	const uint32_t first = ck << MGL_CK2_SHIFT;
	uint32_t w = first >> 6;
	uint64_t keep = ~0ull << (first & 63u); /* the boundary may lie inside a bitmap word */
	while (w < b.nw0) {
		const uint64_t v = b.onwalk[w] & keep;
		if (v) return (w << 6) + ctz64(v);
		w++; keep = ~0ull;
	}
	return MGL_POS_INF;
}

struct Piece {
	uint32_t dst;   /* index in the chain, relative to k0 */
	uint32_t src;   /* index in span[] or in the scratch copy (relative to k0) */
	uint32_t count;
	uint32_t from_span;
};

__global__ void __launch_bounds__(1024) k_apply_chains(DevCtx c, Base2 b, Control* ctl, ApplyBuf ab)
{
	__shared__ uint32_t s_ipos[MGL_SUB_CAP];
	__shared__ uint16_t s_ibit[MGL_SUB_CAP];
	__shared__ uint32_t s_rpos[MGL_SUB_CAP];
	__shared__ uint32_t s_span_pos[MGL_SPAN_CAP];
	__shared__ uint16_t s_span_ev[MGL_SPAN_CAP];
	__shared__ Piece s_piece[MGL_PIECE_CAP];
	/* checkpoint patch segments: positions (lo, hi] take their value from span[first..last) */
	__shared__ uint32_t s_seg_lo[MGL_PIECE_CAP], s_seg_hi[MGL_PIECE_CAP], s_seg_first[MGL_PIECE_CAP], s_seg_last[MGL_PIECE_CAP];
	__shared__ uint16_t s_seg_endp[MGL_PIECE_CAP];
	__shared__ uint32_t s_wcount[16];
	/* the chain entries the re-simulation is about to walk, fetched by the whole workgroup */
	__shared__ uint32_t s_win_pos[MGL_APPLY_WIN];
	__shared__ uint16_t s_win_ev[MGL_APPLY_WIN];
	__shared__ uint32_t s_lo, s_hi, s_k, s_ns, s_job_b, s_job_c, s_span_base, s_scr_base;
	__shared__ uint32_t s_ni, s_nr, s_npiece, s_nseg, s_k0, s_newtail, s_oldlen, s_fail, s_newlen, s_newoff, s_newcap;
	if (!ctl->accepted_flag || ctl->apply_failed) return;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
	const uint32_t n_ins = ab.hdr[0], n_rem = ab.hdr[1], nt = ab.hdr[2];

	const bool prof = c.diag_stop == 50u; /* diagnostic: hdr[8..13] = slowest workgroup's cycles per stage */
	for (uint32_t ti = blockIdx.x; ti < nt; ti += gridDim.x) {
		const uint32_t cx = ab.tctx[ti];
		unsigned long long t0 = prof ? __builtin_readcyclecounter() : 0ull, t1;
#define APPLY_STAGE(i_) if (prof && tid == 0) { t1 = __builtin_readcyclecounter(); atomicMax(&ab.hdr[8 + (i_)], (uint32_t)(t1 - t0)); t0 = t1; }
		__syncthreads();
		if (tid == 0) { s_ni = 0; s_nr = 0; s_fail = 0; }
		__syncthreads();
		/* ---- 1. this context's inserted / removed events, order preserved */
		for (int pass = 0; pass < 2; pass++) {
			const uint32_t m = pass == 0 ? n_ins : n_rem;
			for (uint32_t base = 0; base < m; base += MGL_APPLY_THREADS) {
				const uint32_t e = base + tid;
				bool hit = false;
				uint32_t key = 0, pos = 0;
				if (e < m) {
					key = pass == 0 ? ab.ins_key[e] : ab.rem_key[e];
					pos = pass == 0 ? ab.ins_pos[e] : ab.rem_pos[e];
					hit = (key & 0x7FFFu) == cx;
				}
				const unsigned long long mask = __ballot(hit);
				if (lane == 0) s_wcount[wid] = (uint32_t)__popcll(mask);
				__syncthreads();
				uint32_t before = pass == 0 ? s_ni : s_nr;
				for (uint32_t w = 0; w < wid; w++) before += s_wcount[w];
				const uint32_t idx = before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
				if (hit) {
					if (idx < MGL_SUB_CAP) {
						if (pass == 0) { s_ipos[idx] = pos; s_ibit[idx] = (uint16_t)(key >> 15); }
						else s_rpos[idx] = pos;
					} else s_fail = 1;
				}
				__syncthreads();
				if (tid == 0) {
					uint32_t tot = 0;
					for (uint32_t w = 0; w < MGL_APPLY_THREADS / 64; w++) tot += s_wcount[w];
					if (pass == 0) s_ni += tot; else s_nr += tot;
				}
				__syncthreads();
			}
		}
		if (s_fail) { if (tid == 0) ctl->apply_failed = 1; continue; }
		APPLY_STAGE(0)

		const uint32_t off = b.ch_off[cx], len = b.ch_len[cx], cap = b.ch_cap[cx];
		uint32_t* cpos = b.ch_pos + off;
		uint16_t* cev = b.ch_ev + off;
		/* ---- 1b. the chain index: every block behind the first change has (inserted - removed events below it) more entries
		 * before it (the lists are in position order) */
		{
			const uint32_t ni = s_ni, nr = s_nr;
			const uint32_t first = ((ni ? s_ipos[0] : MGL_POS_INF) < (nr ? s_rpos[0] : MGL_POS_INF)) ? s_ipos[0] : s_rpos[0];
			uint32_t* row = b.ch_sb + (size_t)cx * b.sb_stride;
			for (uint32_t blk = (first >> b.sb_shift) + 1u + tid; blk <= b.nsb; blk += MGL_APPLY_THREADS) {
				const uint32_t bound = blk == b.nsb ? MGL_POS_INF : blk << b.sb_shift;
				uint32_t a = 0, z = ni;
				while (a < z) { const uint32_t m = (a + z) >> 1; if (s_ipos[m] < bound) a = m + 1; else z = m; }
				const uint32_t di = a;
				a = 0; z = nr;
				while (a < z) { const uint32_t m = (a + z) >> 1; if (s_rpos[m] < bound) a = m + 1; else z = m; }
				if (di != a) row[blk] += di - a;
			}
		}
		/* ---- 2. re-simulate: new entries, pieces, checkpoint segments.  First the whole workgroup
		 * finds the chain entry of the first change (1024-ary search) and stages the entries from
		 * there on in LDS; then one thread walks them. */
		const uint32_t x0 = ((s_ni ? s_ipos[0] : MGL_POS_INF) < (s_nr ? s_rpos[0] : MGL_POS_INF)) ? s_ipos[0] : s_rpos[0];
		if (tid == 0) { s_lo = 0; s_hi = len; }
		__syncthreads();
		for (;;) { /* invariant: entries before s_lo are < x0, entry s_hi is >= x0 (the sentinel is) */
			const uint32_t lo = s_lo, hi = s_hi;
			if (hi <= lo) break;
			const uint32_t step = (hi - lo + MGL_APPLY_THREADS - 1u) / MGL_APPLY_THREADS;
			const uint32_t idx = lo + tid * step;
			const bool lt = idx < hi && cpos[idx] < x0;
			const bool next_lt = (idx + step) < hi && cpos[idx + step] < x0;
			__syncthreads();
			if (tid == 0 && !lt) s_hi = lo; /* not even the first probe is below x0 */
			__syncthreads();
			if (lt && !next_lt) { s_lo = idx + 1u; s_hi = (idx + step) < hi ? (idx + step) : hi; }
			__syncthreads();
		}
		const uint32_t k0c = s_lo;
		for (uint32_t i = tid; i < MGL_APPLY_WIN; i += MGL_APPLY_THREADS) {
			const bool in = k0c + i <= len;
			s_win_pos[i] = in ? cpos[k0c + i] : MGL_POS_INF;
			s_win_ev[i] = in ? cev[k0c + i] : (uint16_t)0;
		}
		__syncthreads();
		APPLY_STAGE(1)
		if (wid == 0) { /* the first wavefront, every lane with the same state (shared-memory writes are the same value from every lane);
		                 * the tail -- old entries re-priced until the probability re-joins -- sixty-four entries at a time */
			const uint32_t ni = s_ni, nr = s_nr;
			uint32_t ii = 0, ri = 0, ns = 0, np = 0, nseg = 0;
			const uint32_t k0 = k0c;
#define CPOS(k_) (((k_) - k0) < MGL_APPLY_WIN ? s_win_pos[(k_) - k0] : cpos[(k_)])
#define CEV(k_) (((k_) - k0) < MGL_APPLY_WIN ? (uint32_t)s_win_ev[(k_) - k0] : (uint32_t)cev[(k_)])
			uint32_t k = k0, dst = 0;
			bool fail = false;
			uint32_t seg_first = 0, seg_lo = x0;
			bool in_seg = true; /* a segment = a stretch where the new trajectory differs / entries change */
			/* everything the loop compares lives in registers and is reloaded only when its cursor
			 * moves; the next old entry is fetched one iteration ahead (one thread, LDS latency bound) */
			uint32_t bpos = CPOS(k), ev = CEV(k);
			uint32_t nbpos = bpos != MGL_POS_INF ? CPOS(k + 1) : MGL_POS_INF, nev = bpos != MGL_POS_INF ? CEV(k + 1) : 0u;
			uint32_t ipos = ni ? s_ipos[0] : MGL_POS_INF, ibit = ni ? s_ibit[0] : 0u;
			uint32_t rposn = nr ? s_rpos[0] : MGL_POS_INF;
			uint32_t p = ev & 0x7FFu;
			for (;;) {
				const bool pending = ii < ni || ri < nr;
				if (!pending && in_seg) {
					/* nothing left to insert or remove: follow the old entries until the probability
					 * re-joins them, four staged entries per round of LDS reads */
					bool moved_on = false;
					for (;;) {
						const uint32_t rel = k - k0;
						if (rel >= MGL_APPLY_WIN || ns >= MGL_SPAN_CAP) break;
						uint32_t avail = MGL_APPLY_WIN - rel;
						avail = avail < MGL_SPAN_CAP - ns ? avail : MGL_SPAN_CAP - ns;
						avail = avail < 64u ? avail : 64u;
						const uint32_t my_pos = lane < avail ? s_win_pos[rel + lane] : MGL_POS_INF;
						const uint32_t my_ev = lane < avail ? (uint32_t)s_win_ev[rel + lane] : 0u;
						const unsigned long long okm = __ballot(my_pos != MGL_POS_INF); /* entries in front of the sentinel */
						const uint32_t cnt = okm == ~0ull ? 64u : (uint32_t)__ffsll((long long)~okm) - 1u;
						if (cnt == 0u) break; /* the sentinel (or the end of what is staged): the general code below closes up */
						const unsigned long long bits = __ballot((my_ev >> 15) != 0u);
						uint32_t q = p, mine = 0;
						for (uint32_t e = 0; e < cnt; e++) { /* the recurrence alone; every lane keeps the value in front of its entry */
							mine = lane == e ? q : mine;
							q = mgl_prob_update(q, (uint32_t)((bits >> e) & 1ull));
						}
						const unsigned long long eqm = __ballot(lane < cnt && mine == (my_ev & 0x7FFu));
						uint32_t take = cnt;
						if (eqm != 0ull) { /* re-coupled at that entry (which stays) */
							const uint32_t f = (uint32_t)__ffsll((long long)eqm) - 1u;
							take = f; q = rdlane(mine, f);
						}
						if (lane < take) { s_span_pos[ns + lane] = my_pos; s_span_ev[ns + lane] = (uint16_t)((my_ev & 0x8000u) | mine); }
						ns += take; k += take; p = q;
						moved_on = moved_on || take != 0u;
						if (eqm != 0ull || cnt < avail) break;
					}
					if (moved_on) {
						bpos = CPOS(k); ev = CEV(k);
						nbpos = bpos != MGL_POS_INF ? CPOS(k + 1) : MGL_POS_INF; nev = bpos != MGL_POS_INF ? CEV(k + 1) : 0u;
					}
				}
				if (ipos < bpos) {
					if (ns >= MGL_SPAN_CAP) { fail = true; break; }
					s_span_pos[ns] = ipos; s_span_ev[ns] = (uint16_t)((ibit << 15) | p); ns++;
					p = mgl_prob_update(p, ibit);
					ii++;
					ipos = ii < ni ? s_ipos[ii] : MGL_POS_INF; ibit = ii < ni ? s_ibit[ii] : 0u;
					continue;
				}
				if (bpos == MGL_POS_INF) break; /* reached the sentinel */
				const uint32_t bp = ev & 0x7FFu, bb = ev >> 15;
				const uint32_t nxt = ipos < rposn ? ipos : rposn;
				if (p == bp && (!pending || nxt > bpos)) {
					/* re-coupled at old entry k (which itself stays): close the segment */
					if (nseg >= MGL_PIECE_CAP || np + 2 > MGL_PIECE_CAP) { fail = true; break; }
					s_seg_lo[nseg] = seg_lo; s_seg_hi[nseg] = bpos; s_seg_first[nseg] = seg_first; s_seg_last[nseg] = ns;
					s_seg_endp[nseg] = (uint16_t)p; nseg++;
					s_piece[np].dst = dst; s_piece[np].src = seg_first; s_piece[np].count = ns - seg_first; s_piece[np].from_span = 1; np++;
					dst += ns - seg_first;
					in_seg = false;
					if (!pending) break;
					/* unchanged stretch up to the next change of this context */
					const uint32_t k2 = chain_lower_bound(cpos, len, nxt, nullptr, k);
					s_piece[np].dst = dst; s_piece[np].src = k - k0; s_piece[np].count = k2 - k; s_piece[np].from_span = 0; np++;
					dst += k2 - k;
					k = k2;
					bpos = CPOS(k); ev = CEV(k);
					nbpos = bpos != MGL_POS_INF ? CPOS(k + 1) : MGL_POS_INF; nev = bpos != MGL_POS_INF ? CEV(k + 1) : 0u;
					p = ev & 0x7FFu;
					seg_first = ns; seg_lo = nxt; in_seg = true;
					continue;
				}
				if (rposn == bpos) {
					ri++;
					rposn = ri < nr ? s_rpos[ri] : MGL_POS_INF;
				} else {
					if (ns >= MGL_SPAN_CAP) { fail = true; break; }
					s_span_pos[ns] = bpos; s_span_ev[ns] = (uint16_t)((bb << 15) | p); ns++;
					p = mgl_prob_update(p, bb);
				}
				k++;
				bpos = nbpos; ev = nev;
				if (bpos != MGL_POS_INF) { nbpos = CPOS(k + 1); nev = CEV(k + 1); }
			}
			if (!fail && in_seg) {
				/* ran into the sentinel un-coupled: the last segment reaches the end of the file */
				if (nseg >= MGL_PIECE_CAP || np + 1 > MGL_PIECE_CAP) fail = true;
				else {
					s_seg_lo[nseg] = seg_lo; s_seg_hi[nseg] = MGL_POS_INF; s_seg_first[nseg] = seg_first; s_seg_last[nseg] = ns;
					s_seg_endp[nseg] = (uint16_t)p; nseg++;
					s_piece[np].dst = dst; s_piece[np].src = seg_first; s_piece[np].count = ns - seg_first; s_piece[np].from_span = 1; np++;
					dst += ns - seg_first;
				}
			}
			/* k = first old entry that stays (re-coupled), or the sentinel */
			const uint32_t tail = len + 1 - k;   /* entries [k, len] incl. the sentinel */
			const uint32_t newlen = k0 + dst + (len - k);
			uint32_t newoff = off, newcap = cap;
			if (!fail && newlen + 1 > cap) {
				/* the chain outgrew its slot: move it to fresh space at the top of the pool */
				newcap = (2u * (newlen + 1u) + 256u + 7u) & ~7u;
				newoff = 0;
				if (lane == 0) newoff = atomicAdd(b.pool_top, newcap);
				newoff = uni(newoff);
				if (newoff + newcap > b.pool_cap) fail = true; /* pool exhausted: k_build compacts */
			}
			s_newoff = newoff; s_newcap = newcap;
			if (!fail) {
				s_piece[np].dst = dst; s_piece[np].src = k - k0; s_piece[np].count = tail; s_piece[np].from_span = 0; np++;
				/* an un-coupled end changes the context's final probability: the sentinel */
			}
			s_npiece = np; s_nseg = nseg; s_k0 = k0; s_oldlen = len; s_newlen = newlen; s_fail = fail ? 1u : 0u;
			s_k = k; s_ns = ns;
			s_newtail = (CPOS(k) == MGL_POS_INF && in_seg) ? (p | 0x10000u) : 0u; /* new sentinel value if un-coupled */
#undef CPOS
#undef CEV
		}
		__syncthreads();
		if (s_fail) { if (tid == 0) ctl->apply_failed = 1; continue; }
		APPLY_STAGE(2)
		const uint32_t k0 = s_k0, oldlen = s_oldlen, newlen = s_newlen, kk = s_k, ns = s_ns, np = s_npiece;
		/* ---- 3. the rewrite as copy jobs.  Pass B saves the old entries [k0, len] (only [k0, k) when the
		 * tail neither shifts nor moves) and, for a chain that moves, copies its prefix to the new slot;
		 * pass C writes every piece to its place: new entries from the span space, old ones from the
		 * saved copy.  The tail -- by far the largest piece -- is cut into chunks so that the whole
		 * device moves it. */
		const bool moved = s_newoff != off;
		const bool tail_stays = !moved && newlen == oldlen;
		const bool new_sentinel = (s_newtail & 0x10000u) != 0; /* then the tail is the old sentinel alone */
		const uint32_t tail_count = (tail_stays || new_sentinel) ? 0u : (oldlen + 1u - kk);
		/* Three ways to place the old entries:
		 *  - a chain that moves: everything is read from the old slot and written to the new one (no hazard);
		 *  - in place, tail not shifting: only the small region [k0, k) is saved and rewritten;
		 *  - in place, tail shifting by delta (|delta| <= MGL_SUB_CAP): the small region is saved, and the
		 *    tail moves in one pass of 4 096-entry chunks -- a workgroup loads its chunk into registers,
		 *    then stores it shifted.  The only entries another chunk's stores can reach before they are
		 *    loaded are a chunk's last (first) |delta| ones: pass B saves those slivers and the chunk takes
		 *    them from the save area. */
		const bool shift = !moved && !tail_stays && tail_count != 0;
		const int32_t delta = (int32_t)newlen - (int32_t)oldlen;
		const uint32_t ad = (uint32_t)(delta < 0 ? -delta : delta);
		const uint32_t small_count = moved ? 0u : (kk - k0);
		const uint32_t chunk_s = small_count / 512u > MGL_JOB_CHUNK ? (small_count + 511u) / 512u : MGL_JOB_CHUNK;
		const uint32_t chunk_t = shift ? MGL_JOB_CHUNK : (tail_count / 512u > MGL_JOB_CHUNK ? (tail_count + 511u) / 512u : MGL_JOB_CHUNK);
		const uint32_t chunk_p = k0 / 512u > MGL_JOB_CHUNK ? (k0 + 511u) / 512u : MGL_JOB_CHUNK;
		const uint32_t n_save = (small_count + chunk_s - 1u) / chunk_s;
		const uint32_t n_prefix = moved ? (k0 + chunk_p - 1u) / chunk_p : 0u;
		const uint32_t n_tail = (tail_count + chunk_t - 1u) / chunk_t;
		const uint32_t n_sliver = shift ? n_tail : 0u;
		const uint32_t n_small = np - 1u; /* every piece but the tail (always the last one) */
		const uint32_t scratch_need = small_count + n_sliver * ad;
		if (tid == 0) {
			s_job_b = atomicAdd(&ab.hdr[4], n_save + n_prefix + n_sliver);
			s_job_c = atomicAdd(&ab.hdr[5], n_small + n_tail + (new_sentinel ? 1u : 0u));
			s_span_base = atomicAdd(&ab.hdr[6], ns + 1u);
			s_scr_base = atomicAdd(&ab.hdr[7], scratch_need);
			if (s_job_b + n_save + n_prefix + n_sliver > ab.job_cap || s_job_c + n_small + n_tail + 1u > ab.job_cap ||
			    s_span_base + ns + 1u > ab.span_cap || s_scr_base + scratch_need > ab.scratch_cap || (shift && ad > MGL_SUB_CAP))
				s_fail = 1;
		}
		__syncthreads();
		if (s_fail) { if (tid == 0) ctl->apply_failed = 1; continue; }
		const uint32_t jb = s_job_b, jc = s_job_c, spb = s_span_base, scb = s_scr_base, noff = s_newoff;
		for (uint32_t i = tid; i < ns; i += MGL_APPLY_THREADS) { ab.span_pos[spb + i] = s_span_pos[i]; ab.span_ev[spb + i] = s_span_ev[i]; }
		if (tid == 0 && new_sentinel) { ab.span_pos[spb + ns] = MGL_POS_INF; ab.span_ev[spb + ns] = (uint16_t)(s_newtail & 0x7FFu); }
		/* pass B */
		for (uint32_t i = tid; i < n_save; i += MGL_APPLY_THREADS) {
			const uint32_t at = i * chunk_s, cnt = (small_count - at) < chunk_s ? (small_count - at) : chunk_s;
			ab.jobs_b[jb + i] = make_uint4(off + k0 + at, scb + at, cnt, MGL_SPACE_CHAIN | (MGL_SPACE_SCRATCH << 8));
		}
		for (uint32_t i = tid; i < n_prefix; i += MGL_APPLY_THREADS) {
			const uint32_t at = i * chunk_p, cnt = (k0 - at) < chunk_p ? (k0 - at) : chunk_p;
			ab.jobs_b[jb + n_save + i] = make_uint4(off + at, noff + at, cnt, MGL_SPACE_CHAIN | (MGL_SPACE_CHAIN << 8));
		}
		const Piece tp = s_piece[np - 1u]; /* the tail: { dst, src = k - k0, count = len + 1 - k, from old } */
		for (uint32_t i = tid; i < n_sliver; i += MGL_APPLY_THREADS) {
			const uint32_t at = i * chunk_t, cnt = (tail_count - at) < chunk_t ? (tail_count - at) : chunk_t;
			const uint32_t sl = ad < cnt ? ad : cnt; /* a last chunk shorter than |delta| is a sliver as a whole */
			const uint32_t from = delta < 0 ? (kk + at + cnt - sl) : (kk + at);
			ab.jobs_b[jb + n_save + n_prefix + i] = make_uint4(off + from, scb + small_count + i * ad, sl, MGL_SPACE_CHAIN | (MGL_SPACE_SCRATCH << 8));
		}
		/* pass C */
		for (uint32_t i = tid; i < n_small; i += MGL_APPLY_THREADS) {
			const Piece pc = s_piece[i];
			ab.jobs_c[jc + i] = pc.from_span ? make_uint4(spb + pc.src, noff + k0 + pc.dst, pc.count, MGL_SPACE_SPAN | (MGL_SPACE_CHAIN << 8))
			                    : moved      ? make_uint4(off + k0 + pc.src, noff + k0 + pc.dst, pc.count, MGL_SPACE_CHAIN | (MGL_SPACE_CHAIN << 8))
			                                 : make_uint4(scb + pc.src, noff + k0 + pc.dst, pc.count, MGL_SPACE_SCRATCH | (MGL_SPACE_CHAIN << 8));
		}
		for (uint32_t i = tid; i < n_tail; i += MGL_APPLY_THREADS) {
			const uint32_t at = i * chunk_t, cnt = (tail_count - at) < chunk_t ? (tail_count - at) : chunk_t;
			if (shift) /* { first source entry, sliver in the save area, count, kind | |delta| << 8 | sign << 20 } */
				ab.jobs_c[jc + n_small + i] = make_uint4(off + kk + at, scb + small_count + i * ad, cnt, MGL_SPACE_SHIFT | (ad << 8) | (delta < 0 ? 1u << 20 : 0u));
			else /* a chain that moves: straight from the old slot */
				ab.jobs_c[jc + n_small + i] = make_uint4(off + kk + at, noff + k0 + tp.dst + at, cnt, MGL_SPACE_CHAIN | (MGL_SPACE_CHAIN << 8));
		}
		if (tid == 0 && new_sentinel)
			ab.jobs_c[jc + n_small + n_tail] = make_uint4(spb + ns, noff + k0 + tp.dst, 1u, MGL_SPACE_SPAN | (MGL_SPACE_CHAIN << 8));
		if (tid == 0) {
			b.ch_len[cx] = newlen;
			if (moved) { b.ch_off[cx] = noff; b.ch_cap[cx] = s_newcap; }
		}
		APPLY_STAGE(3)
		/* ---- 4. dense checkpoints: this context's value wherever its trajectory changed */
		for (uint32_t sg = 0; sg < s_nseg; sg++) {
			const uint32_t lo = s_seg_lo[sg], hi = s_seg_hi[sg];
			if (tid == 0) { atomicMin(&ctl->mod_lo, lo); atomicMax(&ctl->mod_hi, hi); }
			const uint32_t first = s_seg_first[sg], last = s_seg_last[sg];
			/* a boundary can lie behind a packet that starts up to 272 bytes before lo */
			const uint32_t ck_lo = (lo > MGL_MAX_MATCH ? lo - MGL_MAX_MATCH : 0u) >> MGL_CK2_SHIFT;
			const uint32_t ck_hi = hi == MGL_POS_INF ? b.nck : ((hi >> MGL_CK2_SHIFT) + 1u < b.nck ? (hi >> MGL_CK2_SHIFT) + 1u : b.nck);
			for (uint32_t ck = ck_lo + tid; ck < ck_hi; ck += MGL_APPLY_THREADS) {
				const uint32_t P = ckpt_boundary(b, ck); /* MGL_POS_INF: behind the last packet = final model */
				if (P <= lo || P > hi) continue;     /* value there is the old one */
				/* probability before the first new event of this segment at or after P */
				uint32_t a = first, z = last;
				while (a < z) { const uint32_t mid = (a + z) >> 1; if (s_span_pos[mid] < P) a = mid + 1; else z = mid; }
				const uint16_t v = a < last ? (uint16_t)(s_span_ev[a] & 0x7FFu) : s_seg_endp[sg];
				b.ck_probs[(size_t)ck * b.ck_elems + cx] = v;
			}
		}
		__syncthreads();
		APPLY_STAGE(4)
#undef APPLY_STAGE
	}
}

/* decide-only variant of the step's tail: flags for k_build's conditional run are reset there */
__device__ void step_end_body(Control* ctl, int lazy_best, uint32_t* counts, int adaptive, int form_single)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		if (adaptive) {
			/* Which form of the regular neighbour launch runs the next step.  The split form has twice
			 * the wavefronts per SIMD, but a neighbour that needs a repair pick costs it a second pass
			 * at a lone wavefront's latency (~0.3 ms whatever the count).  Expected split step =
			 * clean + p (dirty - clean), p = how often a step has such a neighbour (seen in either
			 * form); the one-kernel form is used while its measured step is shorter than that. */
			const unsigned long long now = (unsigned long long)wall_clock64();
			const unsigned long long dt64 = now - ctl->t_last;
			const uint32_t cur = form_single ? 1u : 0u; /* the form the host launched this step */
			const bool dirty = counts[0] != 0 || counts[3] != 0;
			ctl->p_dirty = (uint32_t)((int32_t)ctl->p_dirty + (((dirty ? 65536 : 0) - (int32_t)ctl->p_dirty) >> 5));
			if (ctl->t_last != 0 && ctl->mode_steps >= 1u && dt64 < 0x7FFFFFFFull) {
				uint32_t* e = cur ? &ctl->ema_single : (dirty ? &ctl->ema_dirty : &ctl->ema_clean);
				const uint32_t dt = (uint32_t)dt64;
				if (*e == 0) *e = dt;
				else if (dt < 4u * *e) *e = (15u * *e + dt) >> 4; /* longer: a host-side gap, not a step */
			}
			ctl->t_last = now;
			ctl->mode_steps++;
			uint32_t want = cur;
			if (ctl->probing) {
				if (--ctl->probing == 0) want = 1u - cur; /* trial over: back, then decide below on fresh numbers */
			} else if (cur == 0u && counts[0] > 128u) {
				want = 1u; /* a burst of repairs: hundreds of neighbours would go through the second pass */
				ctl->p_dirty = ctl->p_dirty > 49152u ? ctl->p_dirty : 49152u;
			} else if ((cur == 0u ? ctl->ema_single == 0 : (ctl->ema_clean == 0 && ctl->ema_dirty == 0)) ? ctl->mode_steps >= 8u : ctl->mode_steps >= 2048u) { /* a trial lasts until the host's next look, a block of steps */
				want = 1u - cur; ctl->probing = 4; /* measure the other form for a few steps */
			} else if ((ctl->ema_clean != 0 || ctl->ema_dirty != 0) && ctl->ema_single != 0) {
				const uint32_t clean = ctl->ema_clean ? ctl->ema_clean : ctl->ema_dirty; /* no clean split step seen yet */
				const uint32_t dirty_cost = ctl->ema_dirty > clean ? ctl->ema_dirty : clean + 30000u;
				const unsigned long long exp_split = clean + (((unsigned long long)ctl->p_dirty * (dirty_cost - clean)) >> 16);
				/* 4 % hysteresis around the incumbent */
				if (cur == 0u) want = (unsigned long long)ctl->ema_single * 100ull < exp_split * 96ull ? 1u : 0u;
				else want = exp_split * 100ull < (unsigned long long)ctl->ema_single * 96ull ? 0u : 1u;
			}
			/* a recommendation: the host adopts it at its next look at the control block (mgl_sa_run) */
			if (want != (ctl->nbr_single ? 1u : 0u)) { ctl->nbr_single = want; ctl->mode_steps = 0; }
		}
		if (lazy_best) {
			/* a new best: the base now is the best slab's; any other accepted move: they part ways
			 * (k_snapshot kept a copy just before, see launch_apply) */
			if (ctl->copy_best_flag) ctl->best_is_current = 1;
			else if (ctl->accepted_flag) ctl->best_is_current = 0;
		}
		ctl->accepted_flag = 0; ctl->apply_failed = 0;
		/* the per-step counters ([0] second-pass list, [1] last-resort list, [2] spill slots, [3] repair picks, [4] late second-pass list):
		 * kept for diagnostics in [8..15], cleared for the next step (saves a memset launch) */
		for (int i = 0; i < 8; i++) { counts[8 + i] = counts[i]; counts[i] = 0; }
	}
}
__global__ void k_step_end(Control* ctl, int lazy_best, uint32_t* counts, int adaptive, int form_single)
{
	step_end_body(ctl, lazy_best, counts, adaptive, form_single);
}

/* ================================================================== base snapshots
 *
 * main.c:71-77 starts every epoch from the all-literal slab or from packets_best.  Rather than
 * re-deriving the base structures of those two slabs each time (k_build: a serial walk), a copy of
 * everything that hangs off the slab is kept for each: the all-literal one is taken once at
 * creation, the best one whenever k_decide raises copy_best_flag.  A restore is then a
 * device-to-device copy at HBM speed.  MI355X has 288 GB of HBM: three copies of the structures
 * of a 100 MB input are ~85 GB.
 */
#define MGL_SNAP_SEGS 14u
struct SnapSeg {
	const void* src;
	void* dst;
	unsigned long long bytes;
	uint32_t pool_elem; /* != 0: the segment is a chain pool array, copy only (top + 256) * pool_elem bytes */
};
struct SnapPlan {
	SnapSeg seg[MGL_SNAP_SEGS];
	uint32_t nseg;
	const uint32_t* src_pool_top;
};
struct SnapMeta {
	uint32_t valid, final_ctx_state, pad0, pad;
	uint32_t final_dists[4];
	unsigned long long packets, cost;
};

/* dir 0: base -> snapshot (meta taken from ctl); dir 1: snapshot -> base (ctl restored from meta).
 * cond 1: only when this step produced a new best; cond 2: only when this step is about to move
 * the base away from the best slab it still holds (the copy is taken lazily, at the last moment). */
__global__ void __launch_bounds__(256) k_snapshot(SnapPlan p, Control* ctl, SnapMeta* meta, int dir, int cond)
{
	if (cond == 1 && !ctl->copy_best_flag) return;
	if (cond == 2 && !(ctl->accepted_flag && !ctl->copy_best_flag && ctl->best_is_current)) return;
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		if (dir == 0) {
			meta->valid = 1u;
			meta->final_ctx_state = ctl->final_ctx_state;
			for (int i = 0; i < 4; i++) meta->final_dists[i] = ctl->final_dists[i];
			meta->packets = ctl->packets;
			meta->cost = ctl->rebuild_cost;
		} else {
			ctl->final_ctx_state = meta->final_ctx_state;
			for (int i = 0; i < 4; i++) ctl->final_dists[i] = meta->final_dists[i];
			ctl->packets = meta->packets;
			ctl->rebuild_cost = meta->cost;
		}
	}
	const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
	const unsigned long long top = (unsigned long long)(*p.src_pool_top) + 256ull;
	for (uint32_t s = 0; s < p.nseg; s++) {
		const SnapSeg g = p.seg[s];
		unsigned long long bytes = g.bytes;
		if (g.pool_elem) {
			const unsigned long long lim = top * g.pool_elem;
			bytes = lim < bytes ? lim : bytes;
		}
		const uint4* src = (const uint4*)g.src;
		uint4* dst = (uint4*)g.dst;
		const unsigned long long vecs = bytes >> 4;
		for (unsigned long long i = tid; i < vecs; i += nthreads) dst[i] = src[i];
		const unsigned long long done = vecs << 4;
		if (tid < bytes - done) ((uint8_t*)g.dst)[done + tid] = ((const uint8_t*)g.src)[done + tid];
	}
}


/* carry out one list of copy jobs: entries (position u32 + event u16) between the chain pool, the
 * save area and the span area */
#define MGL_SPACE_SHIFT2 4u /* job kind (batch accept, mgl_kernels5.hip): in-place shift of one chunk of a stretch, slivers saved at both ends */
/* batch_hdr: nullptr for a single accept (runs when this step accepted a move), else the batch accept's header (runs while
 * its status is 1) */
__global__ void __launch_bounds__(256) k_apply_jobs(Base2 b, const Control* ctl, ApplyBuf ab, int pass, const uint32_t* batch_hdr)
{
	if (batch_hdr ? (batch_hdr[0] != 1u || batch_hdr[4] != 0u || ctl->apply_failed) : (!ctl->accepted_flag || ctl->apply_failed)) return;
	const uint32_t njobs = ab.hdr[pass == 0 ? 4 : 5];
	const uint4* jobs = pass == 0 ? ab.jobs_b : ab.jobs_c;
	for (uint32_t j = blockIdx.x; j < njobs; j += gridDim.x) {
		const uint4 job = jobs[j];
		const uint32_t ss = job.w & 0xFFu, ds = (job.w >> 8) & 0xFFu;
		if (ss == MGL_SPACE_SHIFT) {
			/* entries [x, x + z) of the chain pool move by delta; the |delta| entries at the end another
			 * chunk's stores can reach come from the save area (pass B).  Load everything, then store. */
			const uint32_t ad = (job.w >> 8) & 0xFFFu, cnt = job.z;
			const bool left = (job.w >> 20) & 1u;
			const uint32_t sl = ad < cnt ? ad : cnt;
			uint32_t rp[MGL_JOB_CHUNK / 256u];
			uint16_t re[MGL_JOB_CHUNK / 256u];
#pragma unroll
			for (uint32_t u = 0; u < MGL_JOB_CHUNK / 256u; u++) {
				const uint32_t i = u * 256u + threadIdx.x;
				rp[u] = 0; re[u] = 0;
				if (i < cnt) {
					const bool in_sliver = left ? (i >= cnt - sl) : (i < sl);
					const uint32_t si = left ? (i - (cnt - sl)) : i;
					rp[u] = in_sliver ? ab.scratch_pos[job.y + si] : b.ch_pos[job.x + i];
					re[u] = in_sliver ? ab.scratch_ev[job.y + si] : b.ch_ev[job.x + i];
				}
			}
			__syncthreads();
			const uint32_t to = left ? job.x - ad : job.x + ad;
#pragma unroll
			for (uint32_t u = 0; u < MGL_JOB_CHUNK / 256u; u++) {
				const uint32_t i = u * 256u + threadIdx.x;
				if (i < cnt) { b.ch_pos[to + i] = rp[u]; b.ch_ev[to + i] = re[u]; }
			}
			__syncthreads();
			continue;
		}
		if (ss == MGL_SPACE_SHIFT2) {
			/* entries [x, x + z) of the chain pool move by delta; their first and last `sl` entries -- all that another piece's
			 * stores can reach before this chunk has loaded -- come from the copies pass B took (first at y, last at y + maxd) */
			const uint32_t ad = (job.w >> 8) & 0xFFFu, cnt = job.z, maxd = job.w >> 21;
			const bool left = (job.w >> 20) & 1u;
			const uint32_t sl = maxd < cnt ? maxd : cnt;
			uint32_t rp[MGL_JOB_CHUNK / 256u];
			uint16_t re[MGL_JOB_CHUNK / 256u];
#pragma unroll
			for (uint32_t u = 0; u < MGL_JOB_CHUNK / 256u; u++) {
				const uint32_t i = u * 256u + threadIdx.x;
				rp[u] = 0; re[u] = 0;
				if (i < cnt) {
					const bool head = i < sl, tail = i >= cnt - sl;
					const uint32_t si = head ? i : maxd + (i - (cnt - sl));
					rp[u] = (head || tail) ? ab.scratch_pos[job.y + si] : b.ch_pos[job.x + i];
					re[u] = (head || tail) ? ab.scratch_ev[job.y + si] : b.ch_ev[job.x + i];
				}
			}
			__syncthreads();
			const uint32_t to = left ? job.x - ad : job.x + ad;
#pragma unroll
			for (uint32_t u = 0; u < MGL_JOB_CHUNK / 256u; u++) {
				const uint32_t i = u * 256u + threadIdx.x;
				if (i < cnt) { b.ch_pos[to + i] = rp[u]; b.ch_ev[to + i] = re[u]; }
			}
			__syncthreads();
			continue;
		}
		const uint32_t* sp = ss == MGL_SPACE_CHAIN ? b.ch_pos : ss == MGL_SPACE_SCRATCH ? ab.scratch_pos : ab.span_pos;
		const uint16_t* se = ss == MGL_SPACE_CHAIN ? b.ch_ev : ss == MGL_SPACE_SCRATCH ? ab.scratch_ev : ab.span_ev;
		uint32_t* dp = ds == MGL_SPACE_CHAIN ? b.ch_pos : ab.scratch_pos;
		uint16_t* de = ds == MGL_SPACE_CHAIN ? b.ch_ev : ab.scratch_ev;
		sp += job.x; se += job.x; dp += job.y; de += job.y;
		for (uint32_t i = threadIdx.x; i < job.z; i += blockDim.x) { dp[i] = sp[i]; de[i] = se[i]; }
	}
}

/* ================================================================== slab validation
 *
 * The reference emits whatever slab it is given (main.c:116-118).  A slab that enters from outside
 * (mgl_sa_set_slab, mgl_sa_adopt_best / mgl_sa_set_best) is checked here, after its base structures
 * exist: one thread per on-walk position; the walk state before every non-literal packet is its
 * special-state record.  Any violation raises MGL_ERR_BAD_PACKET. */
__global__ void __launch_bounds__(256) k_validate(DevCtx c, Base2 b, Control* ctl)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= c.n) return;
	if (!((b.onwalk[p >> 6] >> (p & 63u)) & 1ull)) return;
	const mgl_pk pk = b.slab[p];
	const uint32_t type = mgl_pk_type(pk), len = mgl_pk_len(pk), dist = mgl_pk_dist(pk);
	bool bad = type < MGL_LITERAL || type > MGL_LONG_REP || len == 0 || p + len > c.n;
	if (!bad) {
		if (type == MGL_LITERAL) bad = len != 1;
		else {
			const uint32_t* r = b.sp_state + (size_t)p * 8;
			const uint32_t d0 = r[1], d1 = r[2], d2 = r[3], d3 = r[4];
			uint32_t src = 0;
			if (type == MGL_SHORT_REP) { bad = len != 1; src = d0; }
			else {
				bad = len < MGL_MIN_MATCH || len > MGL_MAX_MATCH || (type == MGL_LONG_REP && dist > 3u);
				src = type == MGL_MATCH ? dist : (dist == 0 ? d0 : dist == 1 ? d1 : dist == 2 ? d2 : d3);
			}
			if (!bad) bad = src >= p || src >= c.dict_limit;
			if (!bad) {
				const uint8_t* a = c.data + p - src - 1u;
				const uint8_t* z = c.data + p;
				for (uint32_t i = 0; i < len; i++) if (a[i] != z[i]) { bad = true; break; }
			}
		}
	}
	if (bad) atomicOr(&ctl->error_flags, MGL_ERR_BAD_PACKET);
}
