/*
 * mgl_base2.h -- device-resident description of the *base slab* used by the incremental
 * neighbour path.  Sized for HBM capacity rather than frugality:
 *
 *   slab        n x 8 B     packed packets, position-indexed (packet_slab.h:5)
 *   onwalk      n/8 B       bitmap of packet start positions
 *   sp0/1/2     n/8 B ...   3-level bitmap of "special" starts (on-walk, not LITERAL): the only
 *                           packets whose coding depends on / changes the rep distances
 *   sp_state    n x 32 B    walk state (ctx_state, rep distances) before each special packet;
 *                           the state at any other position follows from the previous special
 *   ck_probs    n/64 x ~5 KB  the adaptive model before the first packet at or after every
 *                           64th byte (dense checkpoints: top-K at a random position replays
 *                           at most 64 bytes of packets)
 *   chains      ~2 x 9n x 6 B  for every probability context, its coded events in walk order:
 *                           (position of the packet, bit, probability before the update),
 *                           closed by a sentinel holding the final probability
 *
 * With these a neighbour is costed as   base_total + sum over touched contexts of
 * (re-simulated chain segment - base chain segment), each segment ending where the
 * perturbed probability meets the base trajectory again (typically ~100 events).
 */
#pragma once
#include "mgl_device.h"

#define MGL_CK2_SHIFT 6u
#define MGL_POS_INF 0xFFFFFFFFu
#define MGL_CHG_CAP 256u   /* inserted / removed events per neighbour kept in LDS */
#define MGL_BIG_CAP 4096u   /* the same, per flagged neighbour, in the global scratch of the second pass: a neighbour that inserts
                            * or removes more events than this is dropped (the oracle counts the same events) */
#define MGL_BULK_ROUNDS 8u /* rounds of the bulk selection (mgl_kernels4.hip:k_bulk_round); mirrored by the oracle */
#ifndef MGL_NBR_WAVES_PER_SIMD
#define MGL_NBR_WAVES_PER_SIMD 2 /* register budget of the neighbour kernel: 2 -> 256 VGPRs, no scratch; 3 -> 168 VGPRs
                                    but 200 B/lane of scratch (52 MB of spill traffic per launch) for the same speed */
#endif

struct Base2 {
	mgl_pk* slab;
	uint64_t* onwalk;
	uint64_t* sp0;
	uint64_t* sp1;
	uint64_t* sp2;
	uint32_t nw0, nw1, nw2;
	uint32_t* sp_state;   /* 8 u32 per position */
	uint16_t* ck_probs;
	uint32_t nck, ck_elems;
	uint32_t* ch_off;
	uint32_t* ch_len;     /* entries without the sentinel */
	uint32_t* ch_cap;
	uint32_t* ch_pos;
	uint16_t* ch_ev;      /* bit << 15 | probability before the update */
	uint32_t pool_cap;
	uint32_t* pool_top;
	/* chain index (round 3): ch_sb[c * sb_stride + b] = entries of context c's chain with a position below b << sb_shift
	 * (b = 0 .. nsb; the last one is the chain's length): a search for a position starts inside one block's entries instead
	 * of over the whole chain -- two index loads and one or two probes instead of nine dependent round trips */
	uint32_t* ch_sb;
	uint32_t sb_shift, nsb, sb_stride;
};

__device__ __forceinline__ uint32_t ctz64(uint64_t v) { return (uint32_t)__ffsll((long long)v) - 1u; }
__device__ __forceinline__ uint32_t msb64(uint64_t v) { return 63u - (uint32_t)__clzll((long long)v); }

/* largest special position <= x, or MGL_POS_INF */
__device__ __forceinline__ uint32_t sp_find_prev(const Base2& b, uint32_t x)
{
	uint32_t w = x >> 6;
	uint64_t bits = b.sp0[w] & (~0ull >> (63u - (x & 63u)));
	if (bits) return (w << 6) + msb64(bits);
	if (w == 0) return MGL_POS_INF;
	uint32_t y = w - 1; /* largest non-empty level-0 word <= y */
	uint32_t u = y >> 6;
	uint64_t b1 = b.sp1[u] & (~0ull >> (63u - (y & 63u)));
	if (!b1) {
		if (u == 0) return MGL_POS_INF;
		uint32_t z = u - 1, v = z >> 6;
		uint64_t b2 = b.sp2[v] & (~0ull >> (63u - (z & 63u)));
		while (!b2) {
			if (v == 0) return MGL_POS_INF;
			v--;
			b2 = b.sp2[v];
		}
		u = (v << 6) + msb64(b2);
		b1 = b.sp1[u];
	}
	w = (u << 6) + msb64(b1);
	return (w << 6) + msb64(b.sp0[w]);
}
/* smallest special position >= x, or MGL_POS_INF */
__device__ __forceinline__ uint32_t sp_find_next(const Base2& b, uint32_t x)
{
	uint32_t w = x >> 6;
	if (w >= b.nw0) return MGL_POS_INF;
	uint64_t bits = b.sp0[w] & (~0ull << (x & 63u));
	if (bits) return (w << 6) + ctz64(bits);
	uint32_t y = w + 1;
	if (y >= b.nw0) return MGL_POS_INF;
	uint32_t u = y >> 6;
	uint64_t b1 = b.sp1[u] & (~0ull << (y & 63u));
	if (!b1) {
		uint32_t z = u + 1;
		if (z >= b.nw1) return MGL_POS_INF;
		uint32_t v = z >> 6;
		uint64_t b2 = b.sp2[v] & (~0ull << (z & 63u));
		while (!b2) {
			v++;
			if (v >= b.nw2) return MGL_POS_INF;
			b2 = b.sp2[v];
		}
		u = (v << 6) + ctz64(b2);
		b1 = b.sp1[u];
	}
	w = (u << 6) + ctz64(b1);
	return (w << 6) + ctz64(b.sp0[w]);
}

/* n literal transitions of the ctx_state automaton (lzma_state.c:33-41); 3 reach 0 from anywhere */
__device__ __forceinline__ uint32_t lit_steps(uint32_t s, uint32_t n)
{
	if (n > 3) n = 3;
	for (uint32_t i = 0; i < n; i++) s = mgl_next_ctx_state(s, MGL_LITERAL);
	return s;
}

/* Walk state before the base packet that starts at x (x must be on the base walk). */
__device__ __forceinline__ mgl_wstate base_state_at(const Base2& b, uint32_t x)
{
	mgl_wstate st;
	st.pos = x; st.ctx_state = 0;
	st.dists[0] = st.dists[1] = st.dists[2] = st.dists[3] = 0;
	if (x == 0) return st;
	uint32_t s = sp_find_prev(b, x - 1);
	if (s == MGL_POS_INF) return st; /* only literals so far: state 0, distances 0 */
	const uint32_t* r = b.sp_state + (size_t)s * 8;
	mgl_wstate t;
	t.pos = s; t.ctx_state = r[0];
	t.dists[0] = r[1]; t.dists[1] = r[2]; t.dists[2] = r[3]; t.dists[3] = r[4];
	const mgl_pk pk = b.slab[s];
	mgl_advance(&t, mgl_pk_type(pk), mgl_pk_dist(pk), mgl_pk_len(pk));
	st.ctx_state = lit_steps(t.ctx_state, x - t.pos);
	st.dists[0] = t.dists[0]; st.dists[1] = t.dists[1]; st.dists[2] = t.dists[2]; st.dists[3] = t.dists[3];
	return st;
}
__device__ __forceinline__ mgl_wstate uni_state(mgl_wstate s)
{
	s.pos = uni(s.pos); s.ctx_state = uni(s.ctx_state);
	s.dists[0] = uni(s.dists[0]); s.dists[1] = uni(s.dists[1]); s.dists[2] = uni(s.dists[2]); s.dists[3] = uni(s.dists[3]);
	return s;
}

/* the stretch of context c's chain that holds position x: entries before *lo are below x's block, entry *hi is behind it */
__device__ __forceinline__ void chain_block(const Base2& b, uint32_t c, uint32_t x, uint32_t* lo, uint32_t* hi)
{
	const uint32_t* row = b.ch_sb + (size_t)c * b.sb_stride;
	uint32_t blk = x >> b.sb_shift;
	if (blk >= b.nsb) blk = b.nsb - 1u;
	const uint2 v = make_uint2(row[blk], row[blk + 1u]);
	*lo = v.x; *hi = v.y;
}
/* first chain entry of context c whose packet position is >= x (the sentinel if none) */
#ifndef MGL_LB_WIDE_ABOVE
#define MGL_LB_WIDE_ABOVE 32768u
#endif
/* probes (nullable): the number of chain positions read is added to it (bench.py's self-counted traffic); lo0: entries
 * before it are known to be below x (a search that continues from a place in the chain: the pointer stays the chain's
 * 16-byte aligned start, which the last step relies on) */
__device__ __forceinline__ uint32_t chain_lower_bound(const uint32_t* pos, uint32_t len, uint32_t x, uint32_t* probes = nullptr, uint32_t lo0 = 0)
{
	uint32_t lo = lo0, hi = len; /* answer in [lo, hi] */
	/* the top of a long chain is probed by every neighbour that touches the context (cache-resident): 8-ary steps
	 * there, 7 independent probes per round trip.  Further down every probe is its own 64-byte fetch that nobody
	 * else will use: 4-ary steps (3 probes) move a third of the bytes per halving of the range. */
	while (hi - lo > MGL_LB_WIDE_ABOVE) {
		const uint32_t step = (hi - lo) >> 3;
		uint32_t nlo = lo, nhi = hi;
#pragma unroll
		for (uint32_t i = 1; i < 8; i++) {
			const uint32_t m = lo + i * step;
			const bool ge = pos[m] >= x;
			if (!ge) nlo = m + 1;
			else if (m < nhi) nhi = m;
		}
		lo = nlo; hi = nhi < nlo ? nlo : nhi;
		if (probes) *probes += 7u;
	}
	while (hi - lo > 8) {
		const uint32_t step = (hi - lo) >> 2;
		uint32_t nlo = lo, nhi = hi;
#pragma unroll
		for (uint32_t i = 1; i < 4; i++) {
			const uint32_t m = lo + i * step;
			const bool ge = pos[m] >= x;
			if (!ge) nlo = m + 1;
			else if (m < nhi) nhi = m;
		}
		lo = nlo; hi = nhi < nlo ? nlo : nhi;
		if (probes) *probes += 3u;
	}
	/* at most 8 entries left: all of them in one round trip; the ones below x are a prefix.  (Read as three aligned
	 * 16-byte units instead -- three load instructions for eight -- the re-simulation kernel spilled 14 more registers
	 * and was no faster: c3 k_sim 226 -> 229 us per launch) */
	if (probes) *probes += 8u;
	uint32_t below = 0;
#pragma unroll
	for (uint32_t i = 0; i < 8; i++) {
		const bool in = lo + i < hi;
		const uint32_t v = pos[in ? lo + i : lo];
		below += (in && v < x) ? 1u : 0u;
	}
	return lo + below;
}
