/*
 * mgl_kernels5.hip -- accepting SEVERAL neighbours of one step without rebuilding the base.
 *
 * A bulk step takes a set of mutually compatible neighbours (mgl_kernels4.hip).  Deriving every base structure again
 * from the slab (mgl_pbuild.hip) costs 3 ms at 10 MB whether the step took five moves or five thousand.  When it took
 * at most MGL_BATCH_MAX, the structures are patched instead, the way a single accept patches them (mgl_kernels3.hip),
 * for all the moves at once:
 *
 *   k_batch_clusters  the taken neighbours in position order; those whose windows reach into each other (a neighbour
 *                     may start behind another's soft end, before the rep distances have met again) form a cluster
 *                     with one merged journal.  Clusters are separated by points where walk and base agree entirely.
 *   k_batch_walk      one wavefront per cluster re-runs the two-pointer walk of the merged journal against the OLD base
 *                     (nothing is modified while any walk is running): what changes in the bitmaps and the special-state
 *                     records goes to an op list, the inserted / removed events to a staging area; every rep packet on
 *                     the new walk inside the cluster is checked against the input (the window rule of the selection
 *                     rests on an argument: here is where a wrong combination would show).
 *   k_batch_commit    ops into the bitmaps (atomics: clusters share words) and state records, journals into the slab,
 *                     staged events into one list in position order.  The summary levels of the special bitmap are
 *                     re-derived afterwards (pb_levels).
 *   k_batch_ctxlist   the touched contexts, ascending.
 *   k_batch_chains    one workgroup per touched context.  Its events fall into groups (one per cluster that touches
 *                     it); a thread per group re-simulates from the group's first change until the probability re-joins
 *                     the old trajectory -- through the following groups when it has not re-joined by their first
 *                     change, so a run's result is exact whenever its first group starts from the old trajectory, which
 *                     is true of the first group and then of every group that follows the end of a run: the runs that
 *                     count are found by chasing those ends.  New entries go to the span area; the chain is rewritten
 *                     as copy jobs (stretches between runs move by the runs' accumulated length change, in place, in
 *                     4 096-entry chunks whose first and last entries -- all that another chunk's stores can reach
 *                     before they are read -- are saved first); the context's dense checkpoints are patched along
 *                     every run; the exact change of the total cost is summed (new entries - replaced entries).
 *   k_apply_jobs      (mgl_kernels3.hip) carries out the job lists.
 *   k_batch_end       totals into Control.
 *
 * Anything that does not fit (more than MGL_BATCH_MAX moves, list / journal / span capacities, an invalid rep packet)
 * leaves the step to the full rebuild: both routes produce identical structures (tests/test_gpu_incremental.py).
 */
#include "mgl_base2.h"

#define MGL_BATCH_MAX 192u     /* taken neighbours a batch accept handles */
#define MGL_BATCH_JCAP 512u    /* journal entries per cluster */
#define MGL_BATCH_EVCAP 4096u  /* inserted / removed events staged per cluster */
#define MGL_BATCH_OPCAP 2048u  /* bitmap / state-record ops per cluster */
#define MGL_BATCH_SUB 1024u    /* events of one kind per context */
#define MGL_BATCH_RES 256u     /* entries of the span area every run gets to begin with ... */
#define MGL_BATCH_RES_FEW 1024u /* ... when the step has at most MGL_BATCH_FEW clusters (the evolved slab's steps: a run of the distance-align or slot-root
                                * contexts -- near-random bits -- takes several hundred events to re-join, and one that does not fit runs twice) */
#define MGL_BATCH_FEW 32u
#define MGL_BATCH_GAP 32u      /* events of a context further apart than this start a group of their own */
#define MGL_BATCH_ALLOC (MGL_BATCH_MAX * MGL_BATCH_EVCAP) /* entries of the combined event lists (ApplyBuf lists are allocated this long) */

struct BatchBuf {
	uint32_t* hdr;      /* [0] status: 0 nothing to do, 1 batch accept, 2 left to the rebuild; [1] clusters; [2] inserted events,
	                     * [3] removed events (combined lists); [4] failure seen by a kernel; [5] journal entries; [6] touched contexts */
	uint32_t* cl;       /* per cluster, 8 words: first journal entry, journal entries, staged inserted, staged removed, ops, first target, - , - */
	uint32_t* jpos;     /* merged journals, cluster after cluster */
	mgl_pk* jnew;
	mgl_pk* jold;
	uint16_t* st_ikey; uint32_t* st_ipos; uint16_t* st_rkey; uint32_t* st_rpos; /* per cluster MGL_BATCH_EVCAP */
	uint4* ops;         /* per cluster MGL_BATCH_OPCAP x 2: {pos, flags | count << 8, ctx_state, d0} {d1, d2, d3, 0} */
	uint8_t* ins_cl; uint8_t* rem_cl; /* cluster of every event of the combined lists */
	/* the combined lists again, bucketed by context (k_batch_scan / k_batch_fill): per context its inserted and removed events,
	 * in any order (k_batch_chains sorts its own few by position) */
	uint32_t* cnt_i; uint32_t* cnt_r;   /* per context: events (counted by k_batch_commit) */
	uint32_t* off_i; uint32_t* off_r;   /* exclusive scans */
	uint32_t* cur_i; uint32_t* cur_r;   /* fill cursors */
	uint32_t* touched;                  /* the contexts that have events (hdr[10] of them), listed by k_batch_scan */
	uint32_t* bk_ipos; uint16_t* bk_ibit; uint8_t* bk_icl; /* MGL_BATCH_ALLOC each */
	uint32_t* bk_rpos; uint8_t* bk_rcl;
	uint32_t nctx;      /* contexts (DevCtx::L.total) */
	uint4* runs;        /* the runs that count, two words of four each: {context, lo, hi, first span entry} {entries, end probability, -, -}:
	                     * k_batch_ckpt patches the dense checkpoints along them (hdr[8] = how many) */
	uint32_t runs_cap;
	long long* acc;     /* [0] cost change summed by k_batch_chains [1] direct-bit cost change [2] packets on the walk, change */
};
#define MGL_OP_ON 1u     /* position joins the walk */
#define MGL_OP_OFF 2u    /* `count` positions from pos on leave the walk (they are not special any more either) */
#define MGL_OP_SPEC 4u   /* position is a non-literal packet of the new walk: special bit on, state record written */
#define MGL_OP_PLAIN 8u  /* position is a literal of the new walk: special bit off */

/* ------------------------------------------------------------------ clusters */
__global__ void __launch_bounds__(1024) k_batch_clusters(DevCtx c, const Control* ctl, NbrOut out, BulkBuf bb, BatchBuf bt)
{
	__shared__ uint32_t s_j[MGL_BATCH_MAX], s_t[MGL_BATCH_MAX], s_e[MGL_BATCH_MAX], s_nd[MGL_BATCH_MAX], s_ord[MGL_BATCH_MAX];
	__shared__ uint32_t s_joff[MGL_BATCH_MAX + 1], s_cl[MGL_BATCH_MAX * 2];
	__shared__ uint32_t s_ncl, s_bad;
	const uint32_t tid = threadIdx.x;
	const uint32_t m = (uint32_t)bb.hdr[1];
	if (tid == 0) {
		s_bad = 0; s_ncl = 0;
		bt.hdr[0] = m == 0 ? 0u : (m > MGL_BATCH_MAX ? 2u : 1u);
		bt.hdr[1] = 0; bt.hdr[2] = 0; bt.hdr[3] = 0; bt.hdr[4] = 0; bt.hdr[5] = 0; bt.hdr[6] = 0;
		bt.hdr[8] = 0;
		bt.hdr[7] = (uint32_t)bb.hdr[0]; /* acceptable neighbours of this step: the host sizes the next step's selection by it */
		bt.acc[0] = 0; bt.acc[1] = 0; bt.acc[2] = 0;
	}
	if (m == 0 || m > MGL_BATCH_MAX) return;
	for (uint32_t i = tid; i < bt.nctx; i += blockDim.x) { bt.cnt_i[i] = 0; bt.cnt_r[i] = 0; bt.cur_i[i] = 0; bt.cur_r[i] = 0; }
	(void)ctl; (void)c;
	if (tid < m) {
		const uint32_t j = bb.taken[tid];
		s_j[tid] = j; s_t[tid] = out.win[2u * j]; s_e[tid] = out.win[2u * j + 1u]; s_nd[tid] = out.ndiffs[j];
	}
	__syncthreads();
	/* position order (targets are distinct: two neighbours with one target conflict) */
	if (tid < m) {
		uint32_t r = 0;
		for (uint32_t i = 0; i < m; i++) r += (s_t[i] < s_t[tid] || (s_t[i] == s_t[tid] && i < tid)) ? 1u : 0u;
		s_ord[r] = tid;
	}
	__syncthreads();
	if (tid == 0) {
		uint32_t ncl = 0, reach = 0, joff = 0;
		for (uint32_t r = 0; r < m; r++) {
			const uint32_t i = s_ord[r];
			s_joff[r] = joff;
			if (s_nd[i] == 0) continue; /* a taken neighbour that changes nothing (its mutation re-made the packet that was there) */
			if (ncl == 0 || s_t[i] >= reach) { /* walk and base agree entirely from `reach` on: a new cluster */
				if (ncl) s_cl[(ncl - 1u) * 2u + 1u] = joff - s_cl[(ncl - 1u) * 2u];
				s_cl[ncl * 2u] = joff;
				ncl++;
			}
			joff += s_nd[i];
			reach = s_e[i] > reach ? s_e[i] : reach;
		}
		s_joff[m] = joff;
		if (ncl) s_cl[(ncl - 1u) * 2u + 1u] = joff - s_cl[(ncl - 1u) * 2u];
		s_ncl = ncl;
		for (uint32_t q = 0; q < ncl; q++) if (s_cl[q * 2u + 1u] > MGL_BATCH_JCAP) s_bad = 1;
	}
	__syncthreads();
	if (tid < s_ncl) { bt.cl[tid * 8u] = s_cl[tid * 2u]; bt.cl[tid * 8u + 1u] = s_cl[tid * 2u + 1u]; }
	/* the merged journals: member after member (a member's entries all lie before the next member's target); one thread per
	 * (member, entry) */
	for (uint32_t x = tid; x < m * MGL_MAX_DIFFS; x += blockDim.x) {
		const uint32_t r = x / MGL_MAX_DIFFS, e = x - r * MGL_MAX_DIFFS;
		const uint32_t i = s_ord[r];
		if (e >= s_nd[i]) continue;
		const size_t k = (size_t)s_j[i] * MGL_MAX_DIFFS + e;
		const uint32_t at = s_joff[r] + e;
		bt.jpos[at] = out.dpos[k]; bt.jnew[at] = out.dnew[k]; bt.jold[at] = out.dold[k];
	}
	__syncthreads();
	/* ... which must come out strictly ascending: two taken journals on one entry (or out of order) are the rebuild's business */
	const uint32_t tot = s_joff[m];
	for (uint32_t x = tid; x < m * MGL_MAX_DIFFS; x += blockDim.x) {
		const uint32_t r = x / MGL_MAX_DIFFS, e = x - r * MGL_MAX_DIFFS;
		const uint32_t i = s_ord[r];
		if (e >= s_nd[i]) continue;
		const uint32_t at = s_joff[r] + e;
		if (at + 1u >= tot) continue;
		/* the next entry of the merged journal: this member's next one, or the first one of the next member that has any */
		const size_t k = (size_t)s_j[i] * MGL_MAX_DIFFS + e;
		uint32_t nxt;
		if (e + 1u < s_nd[i]) nxt = out.dpos[k + 1];
		else {
			uint32_t r2 = r + 1u;
			while (r2 < m && s_nd[s_ord[r2]] == 0) r2++;
			nxt = r2 < m ? out.dpos[(size_t)s_j[s_ord[r2]] * MGL_MAX_DIFFS] : 0xFFFFFFFFu;
		}
		if (out.dpos[k] >= nxt) s_bad = 1;
	}
	__syncthreads();
	if (tid == 0) {
		bt.hdr[1] = s_ncl; bt.hdr[5] = tot;
		if (s_bad) bt.hdr[0] = 2u;
	}
}

/* ------------------------------------------------------------------ the walks (read-only on the base) */
__device__ __forceinline__ void batch_op(const BatchBuf& bt, uint32_t cl, uint32_t& nops, bool& failed, uint32_t pos, uint32_t flags, uint32_t count,
                                         const mgl_wstate& st, uint32_t lane)
{
	if (nops >= MGL_BATCH_OPCAP) { failed = true; return; }
	if (lane == 0) {
		uint4* o = bt.ops + ((size_t)cl * MGL_BATCH_OPCAP + nops) * 2u;
		o[0] = make_uint4(pos, flags | (count << 8), st.ctx_state, st.dists[0]);
		o[1] = make_uint4(st.dists[1], st.dists[2], st.dists[3], 0u);
	}
	nops++;
}
/* every byte of a rep packet against its source (packet_slab_neighbour.c:74-80, as k_validate does for a whole slab) */
__device__ __forceinline__ bool batch_rep_ok(const DevCtx& c, uint32_t p, uint32_t rd, uint32_t len, uint32_t lane)
{
	if (rd >= p || rd >= c.dict_limit || p + len > c.n) return false;
	const uint32_t src = p - rd - 1u;
	bool bad = false;
	for (uint32_t i = lane; i < len; i += 64) bad |= c.data[src + i] != c.data[p + i];
	return __ballot(bad) == 0;
}

__global__ void __launch_bounds__(64) k_batch_walk(DevCtx c, Base2 b, BatchBuf bt)
{
	__shared__ uint32_t s_jpos[MGL_BATCH_JCAP];
	__shared__ mgl_pk s_jnew[MGL_BATCH_JCAP];
	if (bt.hdr[0] != 1u) return;
	const uint32_t cl = blockIdx.x, lane = threadIdx.x;
	if (cl >= bt.hdr[1]) return;
	const uint32_t j0 = bt.cl[cl * 8u], nd = bt.cl[cl * 8u + 1u];
	for (uint32_t i = lane; i < nd; i += 64) { s_jpos[i] = bt.jpos[j0 + i]; s_jnew[i] = bt.jnew[j0 + i]; }
	wave_sync();
	const uint32_t t = s_jpos[0], last_j = s_jpos[nd - 1];
	uint16_t* ikey = bt.st_ikey + (size_t)cl * MGL_BATCH_EVCAP; uint32_t* ipos = bt.st_ipos + (size_t)cl * MGL_BATCH_EVCAP;
	uint16_t* rkey = bt.st_rkey + (size_t)cl * MGL_BATCH_EVCAP; uint32_t* rpos = bt.st_rpos + (size_t)cl * MGL_BATCH_EVCAP;

	mgl_wstate nb = uni_state(base_state_at(b, t));
	mgl_wstate bs = nb;
	Win win; win.base = 0xFFFFFFFFu; win.pk = 0; win.byte = 0;
	uint32_t n_ins = 0, n_rem = 0, nops = 0;
	int32_t dpackets = 0;
	long long ddirect = 0;
	bool failed = false, invalid = false;
	uint32_t ji = 0, guard = 0;
	while (nb.pos < c.n || bs.pos < c.n) {
		if (failed || ++guard > (1u << 18)) { failed = true; break; }
		if (nb.pos == bs.pos) {
			const bool same_ctx = nb.ctx_state == bs.ctx_state;
			const bool same_d = nb.dists[0] == bs.dists[0] && nb.dists[1] == bs.dists[1] && nb.dists[2] == bs.dists[2] && nb.dists[3] == bs.dists[3];
			if (same_ctx && same_d && nb.pos > last_j) break;
			if (same_ctx && nb.ctx_state < 7) {
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
			if (ntype < MGL_LITERAL || ntype > MGL_LONG_REP || nlen == 0 || p + nlen > c.n) { invalid = true; break; }
			const bool paired = bs.pos == p;
			/* a rep packet codes the bytes its distance slot points at in THIS walk: checked, not assumed */
			if (ntype == MGL_SHORT_REP || ntype == MGL_LONG_REP) {
				if ((ntype == MGL_LONG_REP && ndist > 3u) || !batch_rep_ok(c, p, mgl_dist_at(&nb, ntype == MGL_SHORT_REP ? 0u : ndist), nlen, lane)) { invalid = true; break; }
			}
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
			/* p is on the new walk; special iff not a literal, with the state before it */
			batch_op(bt, cl, nops, failed, p, (paired ? 0u : MGL_OP_ON) | (ntype != MGL_LITERAL ? MGL_OP_SPEC : MGL_OP_PLAIN), 1u, nb, lane);
			if (!cancelled) {
				if (n_ins + npl.nev > MGL_BATCH_EVCAP || (paired && n_rem + bpl.nev > MGL_BATCH_EVCAP)) { failed = true; break; }
				if (lane < npl.nev) {
					uint32_t ctx, bit;
					mgl_plan_event(&npl, lane, &ctx, &bit);
					ikey[n_ins + lane] = (uint16_t)(ctx | (bit << 15));
					ipos[n_ins + lane] = p;
				}
				n_ins += npl.nev;
				ddirect += (long long)((uint64_t)npl.ndirect << 11);
				if (paired) {
					if (lane < bpl.nev) {
						uint32_t ctx, bit;
						mgl_plan_event(&bpl, lane, &ctx, &bit);
						rkey[n_rem + lane] = (uint16_t)ctx;
						rpos[n_rem + lane] = p;
						}
					n_rem += bpl.nev;
					ddirect -= (long long)((uint64_t)bpl.ndirect << 11);
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
				const uint32_t o = q - win.base;
				const unsigned long long lit = __ballot(mgl_pk_type(win.pk) == MGL_LITERAL) >> o;
				uint32_t run = ~lit == 0ull ? 64u : (uint32_t)__ffsll((long long)~lit) - 1u;
				const uint32_t limit = (nb.pos < c.n ? nb.pos : c.n) - q;
				run = run < 64u - o ? run : 64u - o;
				run = run < limit ? run : limit;
				if (run >= 2u) {
					const uint32_t take = run < 7u ? run : 7u;
					if (n_rem + 9u * take > MGL_BATCH_EVCAP) { failed = true; break; }
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
							rkey[n_rem + i * 9u + slot] = (uint16_t)ctx;
							rpos[n_rem + i * 9u + slot] = p;
								}
						n_rem += 9u * take;
						batch_op(bt, cl, nops, failed, q, MGL_OP_OFF, take, bs, lane);
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
			if (n_rem + bpl.nev > MGL_BATCH_EVCAP) { failed = true; break; }
			if (lane < bpl.nev) {
				uint32_t ctx, bit;
				mgl_plan_event(&bpl, lane, &ctx, &bit);
				rkey[n_rem + lane] = (uint16_t)ctx;
				rpos[n_rem + lane] = q;
			}
			n_rem += bpl.nev;
			ddirect -= (long long)((uint64_t)bpl.ndirect << 11);
			batch_op(bt, cl, nops, failed, q, MGL_OP_OFF, 1u, bs, lane);
			dpackets -= 1;
			mgl_advance(&bs, btype, bdist, blen);
		}
	}
	wave_sync();
	if (lane == 0) {
		bt.cl[cl * 8u + 2u] = n_ins; bt.cl[cl * 8u + 3u] = n_rem; bt.cl[cl * 8u + 4u] = nops;
		if (failed || invalid) atomicOr(&bt.hdr[4], invalid ? 2u : 1u);
		atomicAdd((unsigned long long*)&bt.acc[1], (unsigned long long)ddirect);
		atomicAdd((unsigned long long*)&bt.acc[2], (unsigned long long)(long long)dpackets);
	}
}

/* ------------------------------------------------------------------ commit: every walk has finished, nothing reads the old base any more */
__global__ void __launch_bounds__(256) k_batch_commit(DevCtx c, Base2 b, Control* ctl, BatchBuf bt, ApplyBuf ab)
{
	if (bt.hdr[0] != 1u) return;
	if (bt.hdr[4]) { if (blockIdx.x == 0 && threadIdx.x == 0) bt.hdr[0] = 2u; return; } /* a walk gave up: nothing has been touched, the rebuild takes the step */
	const uint32_t cl = blockIdx.x, tid = threadIdx.x, ncl = bt.hdr[1];
	if (cl >= ncl) return;
	(void)c; (void)ctl;
	/* this cluster's place in the combined lists: clusters are in position order */
	uint32_t ioff = 0, roff = 0;
	for (uint32_t q = 0; q < cl; q++) { ioff += bt.cl[q * 8u + 2u]; roff += bt.cl[q * 8u + 3u]; }
	const uint32_t n_ins = bt.cl[cl * 8u + 2u], n_rem = bt.cl[cl * 8u + 3u], nops = bt.cl[cl * 8u + 4u];
	if (cl == ncl - 1u && tid == 0) { bt.hdr[2] = ioff + n_ins; bt.hdr[3] = roff + n_rem; }
	const size_t sb = (size_t)cl * MGL_BATCH_EVCAP;
	for (uint32_t e = tid; e < n_ins; e += blockDim.x) {
		const uint16_t key = bt.st_ikey[sb + e];
		ab.ins_key[ioff + e] = key; ab.ins_pos[ioff + e] = bt.st_ipos[sb + e]; bt.ins_cl[ioff + e] = (uint8_t)cl;
		atomicAdd(&bt.cnt_i[key & 0x7FFFu], 1u);
	}
	for (uint32_t e = tid; e < n_rem; e += blockDim.x) {
		const uint16_t key = bt.st_rkey[sb + e];
		ab.rem_key[roff + e] = key; ab.rem_pos[roff + e] = bt.st_rpos[sb + e]; bt.rem_cl[roff + e] = (uint8_t)cl;
		atomicAdd(&bt.cnt_r[key], 1u);
	}
	/* bitmaps and state records */
	for (uint32_t o = tid; o < nops; o += blockDim.x) {
		const uint4 a = bt.ops[((size_t)cl * MGL_BATCH_OPCAP + o) * 2u], d = bt.ops[((size_t)cl * MGL_BATCH_OPCAP + o) * 2u + 1u];
		const uint32_t pos = a.x, flags = a.y & 0xFFu, count = a.y >> 8;
		const uint32_t w = pos >> 6;
		if (flags & MGL_OP_OFF) {
			/* `count` consecutive positions (at most seven, inside one window of 64) leave the walk */
			const unsigned long long mask = ((count >= 64u ? ~0ull : ((1ull << count) - 1ull))) << (pos & 63u);
			atomicAnd((unsigned long long*)&b.onwalk[w], ~mask);
			atomicAnd((unsigned long long*)&b.sp0[w], ~mask);
			continue;
		}
		const unsigned long long bit = 1ull << (pos & 63u);
		if (flags & MGL_OP_ON) atomicOr((unsigned long long*)&b.onwalk[w], bit);
		if (flags & MGL_OP_SPEC) {
			atomicOr((unsigned long long*)&b.sp0[w], bit);
			uint32_t* r = b.sp_state + (size_t)pos * 8;
			r[0] = a.z; r[1] = a.w; r[2] = d.x; r[3] = d.y; r[4] = d.z; r[5] = 0; r[6] = 0; r[7] = 0;
		} else if (flags & MGL_OP_PLAIN) {
			atomicAnd((unsigned long long*)&b.sp0[w], ~bit);
		}
	}
	/* the journal goes into the slab (main.c keeps the mutated slab on accept) */
	const uint32_t j0 = bt.cl[cl * 8u], nd = bt.cl[cl * 8u + 1u];
	for (uint32_t e = tid; e < nd; e += blockDim.x) b.slab[bt.jpos[j0 + e]] = bt.jnew[j0 + e];
}

/* per context: where its events go in the bucketed lists (one workgroup: two exclusive scans over the contexts) */
__global__ void __launch_bounds__(1024) k_batch_scan(BatchBuf bt, ApplyBuf ab)
{
	__shared__ uint32_t s_wi[16], s_wr[16], s_bi, s_br, s_nt;
	if (bt.hdr[0] != 1u) return;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
	if (tid == 0) { s_bi = 0; s_br = 0; s_nt = 0; ab.hdr[4] = 0; ab.hdr[5] = 0; ab.hdr[6] = 0; ab.hdr[7] = 0; } /* + the job / span / scratch cursors of k_apply_jobs' lists */
	__syncthreads();
	for (uint32_t base = 0; base < bt.nctx; base += blockDim.x) {
		const uint32_t cx = base + tid;
		const uint32_t vi = cx < bt.nctx ? bt.cnt_i[cx] : 0u, vr = cx < bt.nctx ? bt.cnt_r[cx] : 0u;
		uint32_t ii = vi, ir = vr;
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t ti = (uint32_t)__shfl_up((int)ii, o, 64), tr = (uint32_t)__shfl_up((int)ir, o, 64);
			if ((int)lane >= o) { ii += ti; ir += tr; }
		}
		if (lane == 63) { s_wi[wid] = ii; s_wr[wid] = ir; }
		__syncthreads();
		uint32_t bi = s_bi, br = s_br;
		for (uint32_t w = 0; w < wid; w++) { bi += s_wi[w]; br += s_wr[w]; }
		if (cx < bt.nctx) { bt.off_i[cx] = bi + ii - vi; bt.off_r[cx] = br + ir - vr; }
		if (cx < bt.nctx && (vi | vr)) bt.touched[atomicAdd(&s_nt, 1u)] = cx; /* (in any order: k_batch_chains takes a workgroup per entry) */
		__syncthreads();
		if (tid == blockDim.x - 1u) { s_bi = bi + ii; s_br = br + ir; }
		__syncthreads();
	}
	if (tid == 0) bt.hdr[10] = s_nt;
}
/* every event of the combined lists into its context's bucket */
__global__ void __launch_bounds__(256) k_batch_fill(BatchBuf bt, ApplyBuf ab)
{
	if (bt.hdr[0] != 1u) return;
	const uint32_t n_ins = bt.hdr[2], n_rem = bt.hdr[3];
	for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n_ins + n_rem; e += gridDim.x * blockDim.x) {
		if (e < n_ins) {
			const uint32_t key = ab.ins_key[e], cx = key & 0x7FFFu;
			const uint32_t at = bt.off_i[cx] + atomicAdd(&bt.cur_i[cx], 1u);
			bt.bk_ipos[at] = ab.ins_pos[e]; bt.bk_ibit[at] = (uint16_t)(key >> 15); bt.bk_icl[at] = bt.ins_cl[e];
		} else {
			const uint32_t r = e - n_ins, cx = ab.rem_key[r];
			const uint32_t at = bt.off_r[cx] + atomicAdd(&bt.cur_r[cx], 1u);
			bt.bk_rpos[at] = ab.rem_pos[r]; bt.bk_rcl[at] = bt.rem_cl[r];
		}
	}
}

/* ------------------------------------------------------------------ chains */
#define MGL_BATCH_THREADS 1024u /* one workgroup per touched context; a thread per group (at most MGL_BATCH_MAX of them), all of them for the parallel parts */
#define MGL_BATCH_GRID 2048u

struct RunOut {
	uint32_t k_start, k_end; /* old entries [k_start, k_end) are replaced */
	uint32_t ns;             /* by this many new entries */
	uint32_t last_group;     /* last group whose events the run consumed */
	uint32_t lo, hi;         /* the trajectory differs for positions in (lo, hi] (hi = MGL_POS_INF: to the end of the file) */
	uint32_t end_p;          /* probability after the run's last new entry */
	uint32_t uncoupled;      /* ran into the sentinel without re-joining: the sentinel's value changes */
	long long dcost;
};

/* One run: from group g's first change through every following change that comes before the probability has re-joined the
 * old trajectory.  WRITE = false: sizes and cost only.  WRITE = true: the new entries to span_pos / span_ev from `at` on. */
template <bool WRITE>
__device__ __forceinline__ RunOut batch_run(const uint32_t* cpos, const uint16_t* cev, uint32_t len, const uint16_t* T,
                                            const uint32_t* s_ipos, const uint16_t* s_ibit, const uint8_t* s_icl, uint32_t ni,
                                            const uint32_t* s_rpos, const uint8_t* s_rcl, uint32_t nr,
                                            uint32_t ii, uint32_t ri, const uint8_t* s_gcl, uint32_t g,
                                            uint32_t* span_pos, uint16_t* span_ev, uint32_t at, uint32_t wcap,
                                            const uint32_t* sb_row, uint32_t sb_row_info /* nsb | sb_shift << 24 */)
{
	RunOut r; /* (WRITE: only the first `wcap` new entries are written; the sizes are exact whatever it is) */
	uint32_t ipos = ii < ni ? s_ipos[ii] : MGL_POS_INF, rpos = ri < nr ? s_rpos[ri] : MGL_POS_INF;
	const uint32_t x0 = ipos < rpos ? ipos : rpos;
	uint32_t k;
	{
		uint32_t blo, bhi; /* the chain index (still the old chain's: patched at the end of the kernel) narrows the search */
		const uint32_t nsb = sb_row_info & 0xFFFFFFu, shift = sb_row_info >> 24;
		uint32_t blk = x0 >> shift;
		if (blk >= nsb) blk = nsb - 1u;
		blo = sb_row[blk]; bhi = sb_row[blk + 1u];
		k = chain_lower_bound(cpos, bhi, x0, nullptr, blo);
	}
	r.k_start = k; r.lo = x0; r.uncoupled = 0; r.hi = MGL_POS_INF;
	/* eight chain entries (positions + events) per round trip, kept in registers (as in chain_sim_contexts: chains start
	 * 32-byte aligned, the pool is over-allocated past its end) */
	uint4 c_pa = make_uint4(0, 0, 0, 0), c_pb = c_pa, c_ev = c_pa;
	uint32_t c_base = 0xFFFFFFFFu;
	auto chunk = [&](uint32_t kk) {
		if ((kk & ~7u) != c_base) {
			c_base = kk & ~7u;
			c_pa = *reinterpret_cast<const uint4*>(cpos + c_base);
			c_pb = *reinterpret_cast<const uint4*>(cpos + c_base + 4);
			c_ev = *reinterpret_cast<const uint4*>(cev + c_base);
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
	uint32_t last_cl = s_gcl[g];
	long long dc = 0;
	uint32_t ns = 0;
	for (;;) {
		chunk(k);
		{
			/* Most of a run is its tail: no change of this context anywhere near, the old entries are re-priced one after the
			 * other until the probability has re-joined theirs.  When the next change lies behind everything this chunk holds,
			 * its entries go through a loop unrolled over the eight registers (no merge logic, no register picked at run time). */
			const uint32_t nxt0 = ipos < rpos ? ipos : rpos;
			const uint32_t lastp = c_pb.w; /* positions ascend: the chunk's last one bounds them all (the sentinel's is infinity) */
			if (c_base + 8u <= len && nxt0 > lastp) { /* (a chunk that holds the sentinel, and whatever lies behind it, goes the slow way) */
				const uint32_t pp[8] = { c_pa.x, c_pa.y, c_pa.z, c_pa.w, c_pb.x, c_pb.y, c_pb.z, c_pb.w };
				const uint32_t ew[4] = { c_ev.x, c_ev.y, c_ev.z, c_ev.w };
				const uint32_t e0 = k & 7u;
				/* nothing the re-join test looks at changes inside the chunk: whether a re-join ends the run is decided once, and the
				 * eight entries go through without a branch (selects; the stores under their own mask) */
				const uint32_t nxt_cl = nxt0 == MGL_POS_INF ? 0xFFFFu : (ipos <= rpos ? (uint32_t)s_icl[ii] : (uint32_t)s_rcl[ri]);
				const bool join_ends = nxt_cl != last_cl;
				bool live = true;
				int32_t dc8 = 0;
#pragma unroll
				for (uint32_t e = 0; e < 8; e++) {
					const uint32_t ev = (ew[e >> 1] >> ((e & 1u) * 16u)) & 0xFFFFu;
					const uint32_t bp = ev & 0x7FFu, bb = ev >> 15;
					const bool in = live && e >= e0;
					const bool stop = in && p == bp && join_ends; /* re-joined: the run ends at old entry k */
					r.hi = stop ? pp[e] : r.hi;
					live = live && !stop;
					const bool take = in && !stop;
					if (WRITE) { /* (an entry that is not written goes to the spare entry behind the run's place: no branch) */
						const uint32_t wi = (take && ns < wcap) ? at + ns : at + wcap;
						span_pos[wi] = pp[e]; span_ev[wi] = (uint16_t)((bb << 15) | p);
					}
					const int32_t d = (int32_t)T[bb ? 2048u - p : p] - (int32_t)T[bb ? 2048u - bp : bp];
					dc8 += take ? d : 0;
					p = take ? mgl_prob_update(p, bb) : p;
					ns += take ? 1u : 0u;
					k += take ? 1u : 0u;
				}
				dc += dc8;
				const bool done = !live;
				if (done) break;
				continue;
			}
		}
		const uint32_t bpos = k > len ? MGL_POS_INF : pos_at(k); /* entry `len` is the sentinel (position = infinity) */
		if (ipos < bpos) { /* an inserted event comes first */
			const uint32_t bit = s_ibit[ii];
			last_cl = s_icl[ii];
			if (WRITE && ns < wcap) { span_pos[at + ns] = ipos; span_ev[at + ns] = (uint16_t)((bit << 15) | p); }
			ns++;
			dc += T[bit ? 2048u - p : p];
			p = mgl_prob_update(p, bit);
			ii++;
			ipos = ii < ni ? s_ipos[ii] : MGL_POS_INF;
			continue;
		}
		if (bpos == MGL_POS_INF) { r.uncoupled = 1; break; } /* the sentinel: the final probability changes */
		const uint32_t ev = ev_at(k);
		const uint32_t bp = ev & 0x7FFu, bb = ev >> 15;
		const uint32_t nxt = ipos < rpos ? ipos : rpos;
		if (p == bp && nxt > bpos) {
			/* re-joined at old entry k (which stays), nothing changes at it.  The run ends here -- unless the context's next
			 * change belongs to the group the run is in (a group's events can come in bunches a few bytes apart: a match at
			 * 246 behind literals re-priced up to 243), and a group has one run; the entries in between are written again as
			 * they are (same probability: no cost change) */
			const uint32_t nxt_cl = nxt == MGL_POS_INF ? 0xFFFFu : (ipos <= rpos ? (uint32_t)s_icl[ii] : (uint32_t)s_rcl[ri]);
			if (nxt_cl != last_cl) { r.hi = bpos; break; }
		}
		dc -= T[bb ? 2048u - bp : bp];
		if (rpos == bpos) { /* the old entry goes away */
			last_cl = s_rcl[ri];
			ri++;
			rpos = ri < nr ? s_rpos[ri] : MGL_POS_INF;
		} else { /* it stays, priced at the new probability */
			if (WRITE && ns < wcap) { span_pos[at + ns] = bpos; span_ev[at + ns] = (uint16_t)((bb << 15) | p); }
			ns++;
			dc += T[bb ? 2048u - p : p];
			p = mgl_prob_update(p, bb);
		}
		k++;
	}
	r.k_end = k; r.ns = ns; r.end_p = p; r.dcost = dc;
	/* the last group consumed: the group of the last event taken (groups are in cluster order, as the events are) */
	uint32_t lg = g;
	while (s_gcl[lg] != last_cl) lg++;
	r.last_group = lg;
	(void)c_base;
	return r;
}

/* The same run by a whole wavefront (k_batch_chains takes this form when a context has few groups: the evolved slab's steps).
 * Every lane carries the run's state; what a lone lane spends its time on -- the tail, old entries re-priced one after the
 * other -- goes sixty-four entries at a time: the probability recurrence alone runs through them (every lane keeping the value
 * in front of its own entry), prices, stores and the re-join test are one entry per lane.  Around the context's changes the
 * steps are the lone lane's, taken by all lanes alike.  Same results, entry for entry (RunOut, the span entries written). */
template <bool WRITE>
__device__ __forceinline__ RunOut batch_run_wave(const uint32_t* cpos, const uint16_t* cev, uint32_t len, const uint16_t* T,
                                                 const uint32_t* s_ipos, const uint16_t* s_ibit, const uint8_t* s_icl, uint32_t ni,
                                                 const uint32_t* s_rpos, const uint8_t* s_rcl, uint32_t nr,
                                                 uint32_t ii, uint32_t ri, const uint8_t* s_gcl, uint32_t g,
                                                 uint32_t* span_pos, uint16_t* span_ev, uint32_t at, uint32_t wcap,
                                                 const uint32_t* sb_row, uint32_t sb_row_info, uint32_t lane)
{
	RunOut r;
	uint32_t ipos = ii < ni ? s_ipos[ii] : MGL_POS_INF, rpos = ri < nr ? s_rpos[ri] : MGL_POS_INF;
	const uint32_t x0 = ipos < rpos ? ipos : rpos;
	uint32_t k;
	{
		const uint32_t nsb = sb_row_info & 0xFFFFFFu, shift = sb_row_info >> 24;
		uint32_t blk = x0 >> shift;
		if (blk >= nsb) blk = nsb - 1u;
		k = uni(chain_lower_bound(cpos, sb_row[blk + 1u], x0, nullptr, sb_row[blk]));
	}
	r.k_start = k; r.lo = x0; r.uncoupled = 0; r.hi = MGL_POS_INF;
	/* sixty-four chain entries, one per lane (entry `len` is the sentinel: position = infinity; nothing is read behind it) */
	uint32_t wbase = 0xFFFFFFFFu, my_pos = MGL_POS_INF, my_ev = 0;
	auto window = [&](uint32_t kk) {
		if (wbase != 0xFFFFFFFFu && kk >= wbase && kk - wbase < 64u) return;
		wbase = kk;
		const uint32_t idx = kk + lane;
		my_pos = idx <= len ? cpos[idx] : MGL_POS_INF;
		my_ev = idx <= len ? (uint32_t)cev[idx] : 0u;
	};
	window(k);
	uint32_t p = rdlane(my_ev, 0) & 0x7FFu;
	uint32_t last_cl = s_gcl[g];
	long long dc_lane = 0, dc_all = 0; /* per lane (the tail's entries) / the same on every lane (the steps around the changes) */
	uint32_t ns = 0;
	for (;;) {
		window(k);
		const uint32_t off = uni(k - wbase);
		const uint32_t nxt0 = ipos < rpos ? ipos : rpos;
		/* the entries from k on that lie in front of the context's next change and of the sentinel: the tail's next stretch */
		const unsigned long long okm = __ballot(lane >= off && wbase + lane < len && my_pos < nxt0) >> off;
		const uint32_t cnt = okm == ~0ull ? 64u - off : (uint32_t)__ffsll((long long)~okm) - 1u;
		if (cnt > 0u) {
			const uint32_t nxt_cl = nxt0 == MGL_POS_INF ? 0xFFFFu : (ipos <= rpos ? (uint32_t)s_icl[ii] : (uint32_t)s_rcl[ri]);
			const bool join_ends = nxt_cl != last_cl;
			const unsigned long long bits = __ballot((my_ev >> 15) != 0u) >> off;
			uint32_t q = p, mine = 0;
			for (uint32_t e = 0; e < cnt; e++) { /* the recurrence, alone */
				mine = lane == off + e ? q : mine;
				q = mgl_prob_update(q, (uint32_t)((bits >> e) & 1ull));
			}
			const uint32_t bp = my_ev & 0x7FFu, bb = my_ev >> 15;
			const bool in = lane >= off && lane < off + cnt;
			const unsigned long long eqm = __ballot(in && mine == bp);
			uint32_t take = cnt;
			bool done = false;
			if (eqm != 0ull && join_ends) { /* re-joined: the run ends at the first such entry (which stays as it is) */
				const uint32_t f = (uint32_t)__ffsll((long long)eqm) - 1u;
				take = f - off; r.hi = rdlane(my_pos, f); q = rdlane(mine, f); done = true;
			}
			if (lane >= off && lane < off + take) {
				const uint32_t e = lane - off;
				if (WRITE && ns + e < wcap) { span_pos[at + ns + e] = my_pos; span_ev[at + ns + e] = (uint16_t)((bb << 15) | mine); }
				dc_lane += (long long)T[bb ? 2048u - mine : mine] - (long long)T[bb ? 2048u - bp : bp];
			}
			ns += take; k += take; p = q;
			if (done) break;
			continue;
		}
		const uint32_t bpos = k > len ? MGL_POS_INF : rdlane(my_pos, off);
		if (ipos < bpos) { /* an inserted event comes first */
			const uint32_t bit = s_ibit[ii];
			last_cl = s_icl[ii];
			if (WRITE && lane == 0 && ns < wcap) { span_pos[at + ns] = ipos; span_ev[at + ns] = (uint16_t)((bit << 15) | p); }
			ns++;
			dc_all += T[bit ? 2048u - p : p];
			p = mgl_prob_update(p, bit);
			ii++;
			ipos = ii < ni ? s_ipos[ii] : MGL_POS_INF;
			continue;
		}
		if (bpos == MGL_POS_INF) { r.uncoupled = 1; break; } /* the sentinel: the final probability changes */
		const uint32_t ev = rdlane(my_ev, off);
		const uint32_t bp = ev & 0x7FFu, bb = ev >> 15;
		const uint32_t nxt = ipos < rpos ? ipos : rpos;
		if (p == bp && nxt > bpos) {
			const uint32_t nxt_cl = nxt == MGL_POS_INF ? 0xFFFFu : (ipos <= rpos ? (uint32_t)s_icl[ii] : (uint32_t)s_rcl[ri]);
			if (nxt_cl != last_cl) { r.hi = bpos; break; }
		}
		dc_all -= T[bb ? 2048u - bp : bp];
		if (rpos == bpos) { /* the old entry goes away */
			last_cl = s_rcl[ri];
			ri++;
			rpos = ri < nr ? s_rpos[ri] : MGL_POS_INF;
		} else { /* it stays, priced at the new probability */
			if (WRITE && lane == 0 && ns < wcap) { span_pos[at + ns] = bpos; span_ev[at + ns] = (uint16_t)((bb << 15) | p); }
			ns++;
			dc_all += T[bb ? 2048u - p : p];
			p = mgl_prob_update(p, bb);
		}
		k++;
	}
	r.k_end = k; r.ns = ns; r.end_p = p;
	r.dcost = (long long)wave_sum64((uint64_t)dc_lane) + dc_all;
	uint32_t lg = g;
	while (s_gcl[lg] != last_cl) lg++;
	r.last_group = lg;
	return r;
}

__global__ void __launch_bounds__(MGL_BATCH_THREADS) k_batch_chains(DevCtx c, Base2 b, Control* ctl, BatchBuf bt, ApplyBuf ab)
{
	__shared__ __attribute__((aligned(16))) uint16_t T[2048];
	__shared__ uint32_t s_ipos[MGL_BATCH_SUB], s_rpos[MGL_BATCH_SUB], s_upos[MGL_BATCH_SUB];
	__shared__ uint16_t s_ibit[MGL_BATCH_SUB];
	__shared__ uint8_t s_icl[MGL_BATCH_SUB], s_rcl[MGL_BATCH_SUB];
	/* groups (one per cluster that touches the context, ascending) and the runs that start at them */
	__shared__ uint8_t s_gcl[MGL_BATCH_MAX + 1];
	__shared__ uint32_t s_gi[MGL_BATCH_MAX + 1], s_gr[MGL_BATCH_MAX + 1]; /* first inserted / removed event of the group */
	__shared__ RunOut s_run[MGL_BATCH_MAX];
	__shared__ uint32_t s_hlist[MGL_BATCH_MAX], s_hspan[MGL_BATCH_MAX + 1], s_gat[MGL_BATCH_MAX], s_hjb[MGL_BATCH_MAX], s_hjc[MGL_BATCH_MAX], s_hscr[MGL_BATCH_MAX];
	__shared__ int32_t s_hdelta[MGL_BATCH_MAX + 1]; /* length change accumulated up to and including head h */
	__shared__ uint32_t s_ng, s_nh, s_fail, s_scr_base, s_job_b, s_job_c, s_newoff, s_newcap, s_newlen, s_maxd;
	if (bt.hdr[0] != 1u || bt.hdr[4]) return;
	if (bt.hdr[9]) { if (blockIdx.x == 0 && threadIdx.x == 0) ctl->apply_failed = 1; return; } /* test hook (mgl_debug_set key 5): give up behind the commit */
	/* one workgroup per touched context (k_batch_scan lists them: a tenth of the contexts on a step of fifteen moves), the
	 * grid strides over the list */
	const uint32_t ntouched = bt.hdr[10];
	if (blockIdx.x >= ntouched) return;
	const uint32_t tid = threadIdx.x;
	for (uint32_t i = tid; i < 256; i += blockDim.x) reinterpret_cast<uint4*>(T)[i] = reinterpret_cast<const uint4*>(c.cost_tbl)[i];
	for (uint32_t w = blockIdx.x; w < ntouched; w += gridDim.x) {
		const uint32_t cx = bt.touched[w];
		const uint32_t ni = bt.cnt_i[cx], nr = bt.cnt_r[cx];
		if (ni > MGL_BATCH_SUB || nr > MGL_BATCH_SUB) { if (tid == 0) ctl->apply_failed = 1; return; }
		long long my_cost = 0; /* thread 0 sums the context's runs */
		__syncthreads(); /* (the shared arrays are the previous context's until here) */
		if (tid == 0) s_fail = 0;
		/* ---- 1. this context's events, by position (its bucket holds them in any order; positions are distinct within a list) */
		for (int pass = 0; pass < 2; pass++) {
			const uint32_t m = pass == 0 ? ni : nr, o = pass == 0 ? bt.off_i[cx] : bt.off_r[cx];
			const uint32_t* gp = pass == 0 ? bt.bk_ipos + o : bt.bk_rpos + o;
			__syncthreads();
			for (uint32_t e = tid; e < m; e += blockDim.x) s_upos[e] = gp[e];
			__syncthreads();
			for (uint32_t e = tid; e < m; e += blockDim.x) {
				const uint32_t pos = s_upos[e];
				uint32_t r = 0;
				for (uint32_t x = 0; x < m; x++) r += s_upos[x] < pos ? 1u : 0u;
				if (pass == 0) { s_ipos[r] = pos; s_ibit[r] = bt.bk_ibit[o + e]; }
				else s_rpos[r] = pos;
			}
		}
		__syncthreads();
		/* ---- 2. groups: the context's events (both lists, by position) in bunches -- a new one wherever the next event lies
		 * more than MGL_BATCH_GAP positions on (at most MGL_BATCH_MAX of them: the last one takes the rest).  Every group
		 * starts a run; which partition is chosen does not matter for the result (runs that have not re-joined the old
		 * trajectory by the next group's first event go on through it), only for how much runs in parallel and how many
		 * unchanged entries between two bunches of one group are written again. */
		if (tid == 0) {
			uint32_t ng = 0, i = 0, r = 0, prev = 0;
			while (i < ni || r < nr) {
				const uint32_t pi = i < ni ? s_ipos[i] : MGL_POS_INF, pr = r < nr ? s_rpos[r] : MGL_POS_INF;
				const uint32_t pos = pi < pr ? pi : pr;
				if (ng == 0 || (pos - prev > MGL_BATCH_GAP && ng < MGL_BATCH_MAX)) { s_gcl[ng] = (uint8_t)ng; s_gi[ng] = i; s_gr[ng] = r; ng++; }
				if (pi == pos) { s_icl[i] = (uint8_t)(ng - 1u); i++; }
				if (pr == pos) { s_rcl[r] = (uint8_t)(ng - 1u); r++; }
				prev = pos;
			}
			s_gcl[ng] = 0xFF; s_gi[ng] = ni; s_gr[ng] = nr;
			s_ng = ng;
		}
		__syncthreads();
		const uint32_t ng = s_ng;
		const uint32_t off = b.ch_off[cx], len = b.ch_len[cx], cap = b.ch_cap[cx];
		const uint32_t* cpos = b.ch_pos + off;
		const uint16_t* cev = b.ch_ev + off;
		uint32_t* const sb_row = b.ch_sb + (size_t)cx * b.sb_stride;
		const uint32_t sb_info = b.nsb | (b.sb_shift << 24);
		/* ---- 2b. the chain index: every block behind the first change has (inserted - removed events below it) more entries
		 * before it.  (The runs below search the OLD chain with the index: a row is patched by its own workgroup only, after
		 * the runs have been planned ... so this happens at the very end, see step 8.) */
		/* ---- 3. every group starts a run, writing its new entries into MGL_BATCH_RES entries of the span area taken for it
		 * (most runs fit: a perturbed probability re-joins the old trajectory after about a hundred events; one that does not is
		 * run again in step 5, into a place of its size) ... */
		const uint32_t res = bt.hdr[1] <= MGL_BATCH_FEW ? MGL_BATCH_RES_FEW : MGL_BATCH_RES;
		/* few groups: a wavefront per run (batch_run_wave); many: a lane per run, all of them side by side */
		const uint32_t lane = tid & 63u, wid = tid >> 6, nwaves = blockDim.x >> 6;
		const bool by_wave = ng <= 3u * nwaves;
		if (by_wave) {
			for (uint32_t g = wid; g < ng; g += nwaves) {
				uint32_t at = 0;
				if (lane == 0) at = atomicAdd(&ab.hdr[6], res + 8u);
				at = uni(at);
				if (lane == 0) s_gat[g] = at;
				if (at + res + 8u > ab.span_cap) { s_fail = 1; continue; }
				const RunOut rr = batch_run_wave<true>(cpos, cev, len, T, s_ipos, s_ibit, s_icl, ni, s_rpos, s_rcl, nr, s_gi[g], s_gr[g], s_gcl, g, ab.span_pos, ab.span_ev, at, res, sb_row, sb_info, lane);
				if (lane == 0) s_run[g] = rr;
			}
		} else if (tid < ng) {
			const uint32_t at = atomicAdd(&ab.hdr[6], res + 8u); /* (+ the spare entry the tail loop writes to instead of branching) */
			s_gat[tid] = at;
			if (at + res + 8u > ab.span_cap) s_fail = 1;
			else s_run[tid] = batch_run<true>(cpos, cev, len, T, s_ipos, s_ibit, s_icl, ni, s_rpos, s_rcl, nr, s_gi[tid], s_gr[tid], s_gcl, tid, ab.span_pos, ab.span_ev, at, res, sb_row, sb_info);
		}
		__syncthreads();
		if (s_fail) { if (tid == 0) ctl->apply_failed = 1; return; }
		/* ---- 4. ... the ones that count begin where the previous one ended */
		if (tid == 0) {
			uint32_t nh = 0, maxd = 0;
			int32_t d = 0;
			for (uint32_t g = 0; g < ng; g = s_run[g].last_group + 1u) {
				const uint32_t spn = s_run[g].ns + (s_run[g].uncoupled ? 1u : 0u); /* + the new sentinel */
				s_hlist[nh] = g;
				s_hspan[nh] = spn <= res ? s_gat[g] : 0xFFFFFFFFu; /* where the run's entries are (absolute); 0xFFFFFFFF: to be written in step 5 */
				d += (int32_t)s_run[g].ns - (int32_t)(s_run[g].k_end - s_run[g].k_start);
				s_hdelta[nh] = d;
				const uint32_t adl = (uint32_t)(d < 0 ? -d : d);
				maxd = adl > maxd ? adl : maxd;
				my_cost += s_run[g].dcost;
				nh++;
			}
			s_nh = nh; s_maxd = maxd;
			const uint32_t newlen = (uint32_t)((int32_t)len + d); /* (a run that ends at the sentinel un-coupled replaces everything up to it, and writes a new one) */
			s_newlen = newlen;
			uint32_t newoff = off, newcap = cap;
			bool fail = maxd > 2047u;
			if (!fail && newlen + 1u > cap) { /* the chain outgrew its slot: fresh space at the top of the pool */
				newcap = (2u * (newlen + 1u) + 256u + 7u) & ~7u;
				newoff = atomicAdd(b.pool_top, newcap);
				if (newoff + newcap > b.pool_cap) fail = true;
			}
			s_newoff = newoff; s_newcap = newcap;
			s_fail = fail ? 1u : 0u;
		}
		__syncthreads();
		if (s_fail) { if (tid == 0) ctl->apply_failed = 1; return; }
		const uint32_t nh = s_nh, noff = s_newoff, maxd = s_maxd;
		const bool moved = noff != off;
		/* ---- 5. a run that counts and did not fit its place: again, into one of its size; the new sentinel behind a run that
		 * ran into the old one un-coupled */
		if (by_wave) {
			for (uint32_t h = wid; h < nh; h += nwaves) {
				const uint32_t g = s_hlist[h];
				const RunOut r0 = s_run[g];
				const uint32_t spn = r0.ns + (r0.uncoupled ? 1u : 0u);
				uint32_t at = s_hspan[h];
				if (at == 0xFFFFFFFFu) {
					at = 0;
					if (lane == 0) at = atomicAdd(&ab.hdr[6], spn + 1u);
					at = uni(at);
					if (at + spn + 1u > ab.span_cap) { s_fail = 1; continue; }
					if (lane == 0) s_hspan[h] = at;
					(void)batch_run_wave<true>(cpos, cev, len, T, s_ipos, s_ibit, s_icl, ni, s_rpos, s_rcl, nr, s_gi[g], s_gr[g], s_gcl, g, ab.span_pos, ab.span_ev, at, spn, sb_row, sb_info, lane);
				}
				if (lane == 0 && r0.uncoupled) { ab.span_pos[at + r0.ns] = MGL_POS_INF; ab.span_ev[at + r0.ns] = (uint16_t)r0.end_p; }
			}
		} else if (tid < nh) {
			const uint32_t g = s_hlist[tid];
			const RunOut& r0 = s_run[g];
			const uint32_t spn = r0.ns + (r0.uncoupled ? 1u : 0u);
			uint32_t at = s_hspan[tid];
			if (at == 0xFFFFFFFFu) {
				at = atomicAdd(&ab.hdr[6], spn + 1u);
				if (at + spn + 1u > ab.span_cap) s_fail = 1;
				else {
					s_hspan[tid] = at;
					(void)batch_run<true>(cpos, cev, len, T, s_ipos, s_ibit, s_icl, ni, s_rpos, s_rcl, nr, s_gi[g], s_gr[g], s_gcl, g, ab.span_pos, ab.span_ev, at, spn, sb_row, sb_info);
				}
			}
			if (!s_fail && r0.uncoupled) { ab.span_pos[at + r0.ns] = MGL_POS_INF; ab.span_ev[at + r0.ns] = (uint16_t)r0.end_p; }
		}
		__syncthreads();
		if (s_fail) { if (tid == 0) ctl->apply_failed = 1; return; }
		/* ---- 6. the rewrite as copy jobs.  Pieces in chain order: [prefix] run 0, stretch 0, run 1, stretch 1, ... ; stretch h
		 * = old entries [k_end(h), k_start(h + 1)) (the last one runs to the sentinel, included) and moves by the length change
		 * accumulated up to run h.  In place: a stretch that does not move is left alone; one that moves goes in chunks that
		 * are loaded whole, then stored (MGL_SPACE_SHIFT2), their first and last `maxd` entries -- all that the stores of
		 * another piece can reach before the chunk is loaded -- coming from copies taken in pass B.  A chain that moves to a
		 * new slot is simply copied piece by piece.  Thread 0 counts the jobs of every piece and takes their places in the
		 * lists; all threads write them (a chain of a million entries is hundreds of chunks). */
		if (tid == 0) {
			uint32_t jb = 0, jc = 0, scr = 0;
			if (moved) jb += (s_run[s_hlist[0]].k_start + MGL_JOB_CHUNK - 1u) / MGL_JOB_CHUNK; /* prefix to the new slot */
			for (uint32_t h = 0; h < nh; h++) {
				const RunOut& r = s_run[s_hlist[h]];
				const uint32_t spn = r.ns + (r.uncoupled ? 1u : 0u);
				s_hjb[h] = jb; s_hjc[h] = jc; s_hscr[h] = scr;
				jc += (spn + MGL_JOB_CHUNK - 1u) / MGL_JOB_CHUNK;
				if (r.uncoupled) continue; /* nothing behind it */
				const uint32_t s0 = r.k_end, s1 = h + 1u < nh ? s_run[s_hlist[h + 1u]].k_start : len + 1u;
				const uint32_t cnt = s1 - s0;
				if (cnt == 0 || (!moved && s_hdelta[h] == 0)) continue;
				const uint32_t nchunks = (cnt + MGL_JOB_CHUNK - 1u) / MGL_JOB_CHUNK;
				jc += nchunks;
				if (!moved) { jb += 2u * nchunks; scr += 2u * maxd * nchunks; }
			}
			s_job_b = atomicAdd(&ab.hdr[4], jb);
			s_job_c = atomicAdd(&ab.hdr[5], jc);
			s_scr_base = atomicAdd(&ab.hdr[7], scr);
			if (s_job_b + jb > ab.job_cap || s_job_c + jc > ab.job_cap || s_scr_base + scr > ab.scratch_cap) s_fail = 1;
		}
		__syncthreads();
		if (s_fail) { if (tid == 0) ctl->apply_failed = 1; return; }
		{
			const uint32_t jb0 = s_job_b, jc0 = s_job_c, scr0 = s_scr_base;
			const uint32_t pre = moved ? (s_run[s_hlist[0]].k_start + MGL_JOB_CHUNK - 1u) / MGL_JOB_CHUNK : 0u; /* the prefix's jobs lead pass B's list */
			if (moved) {
				const uint32_t k0 = s_run[s_hlist[0]].k_start;
				for (uint32_t ci = tid; ci < pre; ci += blockDim.x) {
					const uint32_t at = ci * MGL_JOB_CHUNK;
					ab.jobs_b[jb0 + ci] = make_uint4(off + at, noff + at, (k0 - at) < MGL_JOB_CHUNK ? (k0 - at) : MGL_JOB_CHUNK, MGL_SPACE_CHAIN | (MGL_SPACE_CHAIN << 8));
				}
			}
			(void)pre;
			for (uint32_t h = 0; h < nh; h++) {
				const RunOut& r = s_run[s_hlist[h]];
				const int32_t dprev = h ? s_hdelta[h - 1u] : 0;
				const uint32_t spn = r.ns + (r.uncoupled ? 1u : 0u);
				const uint32_t dst0 = (uint32_t)((int32_t)r.k_start + dprev);
				const uint32_t nsp = (spn + MGL_JOB_CHUNK - 1u) / MGL_JOB_CHUNK;
				for (uint32_t ci = tid; ci < nsp; ci += blockDim.x) {
					const uint32_t at = ci * MGL_JOB_CHUNK;
					ab.jobs_c[jc0 + s_hjc[h] + ci] = make_uint4(s_hspan[h] + at, noff + dst0 + at, (spn - at) < MGL_JOB_CHUNK ? (spn - at) : MGL_JOB_CHUNK, MGL_SPACE_SPAN | (MGL_SPACE_CHAIN << 8));
				}
				if (r.uncoupled) continue;
				const uint32_t s0 = r.k_end, s1 = h + 1u < nh ? s_run[s_hlist[h + 1u]].k_start : len + 1u;
				const uint32_t cnt = s1 - s0;
				const int32_t d = s_hdelta[h];
				if (cnt == 0 || (!moved && d == 0)) continue;
				const uint32_t nchunks = (cnt + MGL_JOB_CHUNK - 1u) / MGL_JOB_CHUNK;
				const uint32_t ad = (uint32_t)(d < 0 ? -d : d);
				for (uint32_t ci = tid; ci < nchunks; ci += blockDim.x) {
					const uint32_t at = ci * MGL_JOB_CHUNK;
					const uint32_t n1 = (cnt - at) < MGL_JOB_CHUNK ? (cnt - at) : MGL_JOB_CHUNK;
					if (moved) { ab.jobs_c[jc0 + s_hjc[h] + nsp + ci] = make_uint4(off + s0 + at, (uint32_t)((int32_t)(noff + s0 + at) + d), n1, MGL_SPACE_CHAIN | (MGL_SPACE_CHAIN << 8)); continue; }
					const uint32_t sl = maxd < n1 ? maxd : n1; /* a chunk shorter than maxd is saved whole (both slivers overlap) */
					const uint32_t scr = scr0 + s_hscr[h] + 2u * maxd * ci;
					ab.jobs_b[jb0 + s_hjb[h] + 2u * ci] = make_uint4(off + s0 + at, scr, sl, MGL_SPACE_CHAIN | (MGL_SPACE_SCRATCH << 8));
					ab.jobs_b[jb0 + s_hjb[h] + 2u * ci + 1u] = make_uint4(off + s0 + at + n1 - sl, scr + maxd, sl, MGL_SPACE_CHAIN | (MGL_SPACE_SCRATCH << 8));
					ab.jobs_c[jc0 + s_hjc[h] + nsp + ci] = make_uint4(off + s0 + at, scr, n1, MGL_SPACE_SHIFT2 | (ad << 8) | (d < 0 ? 1u << 20 : 0u) | (maxd << 21));
				}
			}
			if (tid == 0) {
				b.ch_len[cx] = s_newlen;
				if (moved) { b.ch_off[cx] = noff; b.ch_cap[cx] = s_newcap; }
			}
		}
		/* ---- 6b. the chain index of this context (nothing searches the old chain any more) */
		__syncthreads();
		{
			const uint32_t first = ((ni ? s_ipos[0] : MGL_POS_INF) < (nr ? s_rpos[0] : MGL_POS_INF)) ? s_ipos[0] : s_rpos[0];
			for (uint32_t blk = (first >> b.sb_shift) + 1u + tid; blk <= b.nsb; blk += blockDim.x) {
				const uint32_t bound = blk == b.nsb ? MGL_POS_INF : blk << b.sb_shift;
				uint32_t a = 0, z = ni;
				while (a < z) { const uint32_t m = (a + z) >> 1; if (s_ipos[m] < bound) a = m + 1; else z = m; }
				const uint32_t di = a;
				a = 0; z = nr;
				while (a < z) { const uint32_t m = (a + z) >> 1; if (s_rpos[m] < bound) a = m + 1; else z = m; }
				if (di != a) atomicAdd(&sb_row[blk], di - a); /* (no value comes back: the thread does not wait for the row) */
			}
		}
		/* ---- 7. the runs go on a list: k_batch_ckpt patches this context's value in the dense checkpoints along each of them
		 * (a run of a rare context reaches over hundreds of kilobytes, thousands of rows: a job for more than one wavefront) */
		__syncthreads();
		if (tid < nh) {
			const RunOut& r = s_run[s_hlist[tid]];
			const uint32_t at = atomicAdd(&bt.hdr[8], 1u);
			if (at < bt.runs_cap) {
				bt.runs[2u * at] = make_uint4(cx, r.lo, r.hi, s_hspan[tid]);
				bt.runs[2u * at + 1u] = make_uint4(r.ns, r.end_p, 0u, 0u);
			} else ctl->apply_failed = 1;
		}
		if (tid == 0 && my_cost) atomicAdd((unsigned long long*)&bt.acc[0], (unsigned long long)my_cost);
	}
}

/* dense checkpoints: along every run, the context's value wherever its trajectory changed (one workgroup per run) */
#define MGL_BATCH_CK_SPAN 2048u
__global__ void __launch_bounds__(256) k_batch_ckpt(Base2 b, const Control* ctl, BatchBuf bt, ApplyBuf ab)
{
	__shared__ uint32_t s_pos[MGL_BATCH_CK_SPAN];
	__shared__ uint16_t s_ev[MGL_BATCH_CK_SPAN];
	if (bt.hdr[0] != 1u || bt.hdr[4] || ctl->apply_failed) return;
	const uint32_t nruns = bt.hdr[8] < bt.runs_cap ? bt.hdr[8] : bt.runs_cap;
	for (uint32_t ri = blockIdx.x; ri < nruns; ri += gridDim.x) {
		const uint4 r0 = bt.runs[2u * ri], r1 = bt.runs[2u * ri + 1u];
		const uint32_t cx = r0.x, lo = r0.y, hi = r0.z, first = r0.w, ns = r1.x, end_p = r1.y;
		const bool in_lds = ns <= MGL_BATCH_CK_SPAN;
		__syncthreads();
		if (in_lds) for (uint32_t i = threadIdx.x; i < ns; i += blockDim.x) { s_pos[i] = ab.span_pos[first + i]; s_ev[i] = ab.span_ev[first + i]; }
		__syncthreads();
		/* a boundary can lie behind a packet that starts up to 272 bytes before lo */
		const uint32_t ck_lo = (lo > MGL_MAX_MATCH ? lo - MGL_MAX_MATCH : 0u) >> MGL_CK2_SHIFT;
		const uint32_t ck_hi = hi == MGL_POS_INF ? b.nck : ((hi >> MGL_CK2_SHIFT) + 1u < b.nck ? (hi >> MGL_CK2_SHIFT) + 1u : b.nck);
		/* blockIdx.y: the run's rows in gridDim.y slices (a run of a rare context reaches over thousands of rows) */
		const uint32_t per = (ck_hi - ck_lo + gridDim.y - 1u) / gridDim.y;
		const uint32_t my_lo = ck_lo + blockIdx.y * per, my_hi = (my_lo + per) < ck_hi ? (my_lo + per) : ck_hi;
		/* a run that never re-joined reaches to the end of the file: behind its last new entry every row -- whatever its
		 * boundary -- holds the run's final probability (most of such a run's rows: no boundary search for them) */
		const uint32_t last_pos = hi == MGL_POS_INF ? (ns ? ab.span_pos[first + ns - 1u] : lo) : MGL_POS_INF;
		for (uint32_t ck = my_lo + threadIdx.x; ck < my_hi; ck += blockDim.x) {
			if ((ck << MGL_CK2_SHIFT) > last_pos && last_pos != MGL_POS_INF) { b.ck_probs[(size_t)ck * b.ck_elems + cx] = (uint16_t)end_p; continue; }
			const uint32_t P = ckpt_boundary(b, ck); /* on the NEW walk (the bitmaps are committed); MGL_POS_INF: the final model */
			if (P <= lo || P > hi) continue;      /* the value there is the old one */
			/* probability before the first new entry of the run at or after P */
			uint32_t a = 0, z = ns;
			while (a < z) { const uint32_t mid = (a + z) >> 1; if ((in_lds ? s_pos[mid] : ab.span_pos[first + mid]) < P) a = mid + 1; else z = mid; }
			const uint16_t v = a < ns ? (uint16_t)((in_lds ? s_ev[a] : ab.span_ev[first + a]) & 0x7FFu) : (uint16_t)end_p;
			b.ck_probs[(size_t)ck * b.ck_elems + cx] = v;
		}
	}
}

/* totals of a batch accept into Control: the new base's exact cost, its packet count */
__global__ void k_batch_end(Control* ctl, BatchBuf bt)
{
	if (threadIdx.x || blockIdx.x) return;
	if (bt.hdr[0] != 1u || ctl->apply_failed) return;
	const uint64_t base_cost = ctl->cur_cost ? ctl->cur_cost : ctl->rebuild_cost;
	ctl->rebuild_cost = (uint64_t)((long long)base_cost + bt.acc[0] + bt.acc[1]);
	ctl->packets = (uint64_t)((long long)ctl->packets + bt.acc[2]);
	bt.hdr[0] = 3u; /* done */
}
