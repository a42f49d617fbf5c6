/*
 * mgl_kernels.hip -- CDNA4 (gfx950) kernels of the SA hot path.
 *
 *   k_rebuild      walk the base slab: on-walk bitmap, prefix checkpoints, total cost
 *                  (== the loop of main.c:116-118 under perplexity_encoder.c:6-17)
 *   k_neighbours   one wavefront per candidate neighbour: target pick, prefix from the
 *                  nearest checkpoint, mutate (packet_slab_neighbour.c:119-152), wave-wide
 *                  top-K (top_k_packet_finder.c:95-125 over packet_enumerator.c:57-74 and
 *                  substring_enumerator.c:85-105), repair + cost to the end (:82-117)
 *   (k_decide and the bulk decision live in mgl_kernels4.hip)
 *   k_copy_best    main.c:91
 *   k_topk_probe / k_substrings / k_import / k_export   parity hooks and layout conversion
 *
 * Integer table-lookup work: no MFMA.  Probabilities (u16) and the bit-cost table live in
 * LDS; the input, the slab and the match index stay in HBM/L2 and are read coalesced.
 */
#include "mgl_device.h"

/* ================================================================== top-K (wave-wide) */

struct TopK {
	uint64_t key;   /* lane i (< count) holds the i-th best key; others ~0 */
	uint32_t count; /* uniform */
	uint32_t k;
};

__device__ __forceinline__ uint64_t topk_make_key(uint32_t cost, uint64_t seq)
{
	return ((uint64_t)cost << 44) | (MGL_SEQ_MASK - seq);
}
__device__ __forceinline__ uint64_t topk_threshold(const TopK& t)
{
	return t.count < t.k ? MGL_INVALID_COST : rdlane64(t.key, t.k - 1u); /* count and k are uniform */
}
/* every lane may offer one candidate key (or ~0); all offers better than the current
 * K-th best are merged into the sorted list */
__device__ __forceinline__ void topk_offer(TopK& t, uint64_t cand, uint32_t lane, bool allow_batch = true)
{
	const uint64_t thr0 = topk_threshold(t);
	unsigned long long m = __ballot(cand < thr0);
	if (!m) return;
	if (allow_batch && __popcll(m) >= 4) {
		/* many at once: rank every element of (list + qualifying offers) among all of them -- keys
		 * are distinct -- and send each to the lane of its rank */
		uint32_t less_e = 0, less_c = 0, e_less = 0;
		for (unsigned long long mm = m; mm; mm &= mm - 1ull) {
			const uint32_t j = (uint32_t)__ffsll((long long)mm) - 1u;
			const uint64_t x = rdlane64(cand, j);
			less_e += x < t.key ? 1u : 0u;
			less_c += x < cand ? 1u : 0u;
			const uint32_t below = (uint32_t)__popcll(__ballot(t.key < x)); /* list keys below offer j */
			if (lane == j) e_less = below;
		}
		const bool mine = (m >> lane) & 1ull;
		uint32_t dest_e = lane < t.count ? lane + less_e : 63u;
		uint32_t dest_c = mine ? e_less + less_c : 63u;
		dest_e = dest_e < 63u ? dest_e : 63u; dest_c = dest_c < 63u ? dest_c : 63u;
		const uint64_t ve = lane < t.count ? t.key : 0ull, vc = mine ? cand : 0ull;
		const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest_e << 2), (int)(uint32_t)ve) |
		                    (uint32_t)__builtin_amdgcn_ds_permute((int)(dest_c << 2), (int)(uint32_t)vc);
		const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest_e << 2), (int)(uint32_t)(ve >> 32)) |
		                    (uint32_t)__builtin_amdgcn_ds_permute((int)(dest_c << 2), (int)(uint32_t)(vc >> 32));
		uint32_t nc = t.count + (uint32_t)__popcll(m);
		nc = nc < t.k ? nc : t.k;
		t.count = nc;
		t.key = lane < nc ? ((uint64_t)lo | ((uint64_t)hi << 32)) : MGL_INVALID_COST;
		return;
	}
	for (;;) {
		uint64_t thr = topk_threshold(t);
		m = __ballot(cand < thr);
		if (!m) break;
		int src = __ffsll((long long)m) - 1;
		uint64_t x = rdlane64(cand, (uint32_t)src);
		uint32_t r = (uint32_t)__popcll(__ballot(t.key < x));
		uint64_t up = shfl_up64(t.key, 1);
		if (lane == r) t.key = x;
		else if (lane > r) t.key = up;
		if (lane >= t.k) t.key = MGL_INVALID_COST;
		if (t.count < t.k) t.count++;
		if ((int)lane == src) cand = MGL_INVALID_COST;
	}
}

__device__ __forceinline__ uint32_t bit_cost(const uint16_t* probs, const uint16_t* T, uint32_t ctx, uint32_t bit)
{
	uint32_t p = probs[ctx];
	return T[bit ? 2048u - p : p];
}
__device__ __forceinline__ uint32_t tree_cost(const uint16_t* probs, const uint16_t* T, uint32_t base,
                                              uint32_t val, uint32_t nbits)
{
	uint32_t m = 1, c = 0;
	for (uint32_t i = nbits; i-- > 0;) {
		uint32_t b = (val >> i) & 1u;
		c += bit_cost(probs, T, base + m, b);
		m = (m << 1) | b;
	}
	return c;
}
__device__ __forceinline__ uint32_t rev_tree_cost(const uint16_t* probs, const uint16_t* T, uint32_t base,
                                                  uint32_t val, uint32_t nbits)
{
	uint32_t m = 1, c = 0;
	for (uint32_t i = 0; i < nbits; i++) {
		uint32_t b = val & 1u;
		val >>= 1;
		c += bit_cost(probs, T, base + m, b);
		m = (m << 1) | b;
	}
	return c;
}
/* lzma_packet_encoder.c:42-63 as a read-only cost */
__device__ __forceinline__ uint32_t length_cost(const uint16_t* probs, const uint16_t* T, uint32_t base,
                                                uint32_t len, uint32_t pos_state)
{
	uint32_t l = len - 2;
	if (l < 8) return bit_cost(probs, T, base, 0) + tree_cost(probs, T, base + MGL_LEN_LOW + pos_state * 8, l, 3);
	uint32_t c = bit_cost(probs, T, base, 1);
	if (l < 16) return c + bit_cost(probs, T, base + 1, 0) + tree_cost(probs, T, base + MGL_LEN_MID + pos_state * 8, l - 8, 3);
	return c + bit_cost(probs, T, base + 1, 1) + tree_cost(probs, T, base + MGL_LEN_HIGH, l - 16, 8);
}

/* first index in [a, b) whose bucket position is >= x: 64-ary search, one probe per lane */
__device__ __forceinline__ uint32_t bucket_lower_bound(const uint32_t* bp, uint32_t a, uint32_t b, uint32_t x, uint32_t lane)
{
	while (b > a) {
		uint32_t span = b - a;
		uint32_t step = (span + 63u) / 64u;
		uint32_t idx = a + lane * step;
		bool ge = idx >= b ? true : (bp[idx] >= x);
		unsigned long long m = __ballot(ge);
		/* the predicate is monotone in the lane; f = first lane that holds (64 if none).
		 * lane 0 probes a itself; the answer lies in (probe[f-1], probe[f]] */
		int f = m ? __ffsll((long long)m) - 1 : 64;
		if (f == 0) break; /* answer is a */
		uint32_t na = a + (uint32_t)(f - 1) * step + 1u;
		uint32_t nb = a + (uint32_t)f * step;
		if (nb > b) nb = b;
		a = na; b = nb;
	}
	return a;
}

/* the same search over the big-endian next-two-bytes column of the four-byte order */
__device__ __forceinline__ uint32_t nx_lower_bound(const uint16_t* a, uint32_t lo, uint32_t hi, uint32_t x, uint32_t lane)
{
	while (hi > lo) {
		const uint32_t span = hi - lo, step = (span + 63u) / 64u;
		const uint32_t idx = lo + lane * step;
		const bool ge = idx >= hi ? true : ((uint32_t)a[idx] >= x);
		const unsigned long long m = __ballot(ge);
		const int f = m ? __ffsll((long long)m) - 1 : 64;
		if (f == 0) break;
		const uint32_t na = lo + (uint32_t)(f - 1) * step + 1u;
		uint32_t nb = lo + (uint32_t)f * step;
		if (nb > hi) nb = hi;
		lo = na; hi = nb;
	}
	return lo;
}

/* top_k_packet_finder_find (top_k_packet_finder.c:120-125): enumerate every legal next
 * packet at the walk's state, cost each from the adapted model (cost = perplexity/length,
 * :115-116), keep the k best.  Order-independent ("canonical") selection: better = lower
 * cost, then later in the reference's enumeration order (the reference's `<=`, :89).
 * lencost: LDS scratch, 2 x 272 u32.
 * W: entries a lane takes per trip through the exact-length sources (4 where registers allow: the pick kernel; 1 in the
 * one-kernel form, which has none to spare). */
template <int W>
__device__ void topk_find(TopK& t, const DevCtx& c, const Walk& w, const uint16_t* probs, const uint16_t* T,
                          uint32_t* lencost, mgl_pk incumbent, uint32_t lane)
{
	const uint32_t pos = w.st.pos;
	const uint32_t state = w.st.ctx_state;
	const uint32_t pos_state = pos & ((1u << c.L.pb) - 1u);
	const uint32_t sp = (state << 4) + pos_state;
	t.key = MGL_INVALID_COST;
	t.count = 0;
	t.k = c.top_k;

	/* packet headers, lzma_packet_encoder.c:13-40 */
	const uint32_t c_m1 = bit_cost(probs, T, MGL_CS_IS_MATCH + sp, 1);
	const uint32_t c_r0 = bit_cost(probs, T, MGL_CS_IS_REP + state, 0);
	const uint32_t c_r1 = bit_cost(probs, T, MGL_CS_IS_REP + state, 1);
	const uint32_t g0_0 = bit_cost(probs, T, MGL_CS_G0 + state, 0), g0_1 = bit_cost(probs, T, MGL_CS_G0 + state, 1);
	const uint32_t g1_0 = bit_cost(probs, T, MGL_CS_G1 + state, 0), g1_1 = bit_cost(probs, T, MGL_CS_G1 + state, 1);
	const uint32_t g2_0 = bit_cost(probs, T, MGL_CS_G2 + state, 0), g2_1 = bit_cost(probs, T, MGL_CS_G2 + state, 1);
	const uint32_t l_0 = bit_cost(probs, T, MGL_CS_REP0_LONG + sp, 0), l_1 = bit_cost(probs, T, MGL_CS_REP0_LONG + sp, 1);
	const uint32_t hdr_match = c_m1 + c_r0;
	const uint32_t hdr_rep = c_m1 + c_r1;
	const uint32_t hdr_lr0 = hdr_rep + g0_0 + l_1, hdr_lr1 = hdr_rep + g0_1 + g1_0;
	const uint32_t hdr_lr2 = hdr_rep + g0_1 + g1_1 + g2_0, hdr_lr3 = hdr_rep + g0_1 + g1_1 + g2_1;

	/* price tables in LDS, the way LZMA encoders price candidates: lencost[0..271] (match) and
	 * [272..543] (rep) by length; slotcost[len_ctx][slot] at 544; the reverse-tree tail of every
	 * distance below 128 at 800; the four align bits at 928 (lzma_packet_encoder.c:42-104).
	 * Lengths >= 18 (the 8-bit high tree) are priced only when a match that long shows up. */
	uint32_t* slotcost = lencost + 544;
	uint32_t* disttail = lencost + 800;
	uint32_t* aligncost = lencost + 928;
	if (lane < 32) {
		const uint32_t l = 2 + (lane & 15u);
		const uint32_t base = lane < 16 ? MGL_OFF_LEN : MGL_OFF_REP_LEN;
		lencost[(lane < 16 ? 0 : 272) + l - 2] = length_cost(probs, T, base, l, pos_state);
	}
	for (uint32_t e = lane; e < 256; e += 64) slotcost[e] = tree_cost(probs, T, MGL_OFF_DIST + (e & ~63u), e & 63u, 6);
	for (uint32_t d = 4 + lane; d < 128; d += 64) {
		const uint32_t nlow = mgl_msb32(d) - 2, low = d & ((1u << nlow) - 1u), high = d >> nlow;
		disttail[d] = rev_tree_cost(probs, T, MGL_OFF_DIST + MGL_DIST_POS + (high << nlow) - (nlow * 2 + high), low, nlow);
	}
	if (lane < 16) aligncost[lane] = rev_tree_cost(probs, T, MGL_OFF_DIST + MGL_DIST_ALIGN, lane, 4);
	bool high_ready = false;
	wave_sync();
	/* per distance slot: the least a distance in that slot can cost in any length context (pruning bound) */
	uint32_t* lbslot = lencost + 944;
	uint32_t* sufmin = lencost + 1008;
	{
		const uint32_t slot = lane;
		uint32_t m = slotcost[slot];
		for (uint32_t k = 1; k < 4; k++) { const uint32_t v = slotcost[64u * k + slot]; m = v < m ? v : m; }
		uint32_t amin = lane < 16 ? aligncost[lane] : 0xFFFFFFFFu;
		for (int o = 8; o > 0; o >>= 1) { const uint32_t a2 = (uint32_t)__shfl_xor((int)amin, o, 64); amin = a2 < amin ? a2 : amin; }
		amin = uni(amin);
		/* direct bits + at least the cheapest align nibble; the reverse-tree tails of nearer slots count as 0 */
		if (slot >= 14) m += (((slot >> 1) - 5u) << 11) + amin;
		lbslot[slot] = m;
		/* sufmin[s] = the least of lbslot[s..63] */
		uint32_t sm = m;
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t tdn = (uint32_t)__shfl_down((int)sm, o, 64);
			if ((int)lane + o < 64) sm = tdn < sm ? tdn : sm;
		}
		sufmin[slot] = sm;
	}
	wave_sync();
	if (c.diag_stop == 31) return;
	/* cheapest length price of either coder over the lengths priced so far: lower bounds for pruning */
	uint32_t minlen_m, minlen_r;
	{
		uint32_t a = lane < 16 ? lencost[lane] : 0xFFFFFFFFu, b = lane < 16 ? lencost[272 + lane] : 0xFFFFFFFFu;
		for (int o = 8; o > 0; o >>= 1) {
			const uint32_t a2 = (uint32_t)__shfl_xor((int)a, o, 64), b2 = (uint32_t)__shfl_xor((int)b, o, 64);
			a = a2 < a ? a2 : a; b = b2 < b ? b2 : b;
		}
		minlen_m = uni(a); minlen_r = uni(b);
	}

	/* LITERAL and SHORT_REP, packet_enumerator.c:60-66 */
	{
		uint64_t cand = MGL_INVALID_COST;
		uint32_t byte = walk_byte_at(w, pos);
		uint32_t rep_byte = (pos > 0 && w.st.dists[0] < pos) ? c.data[pos - w.st.dists[0] - 1] : 0x100u;
		uint32_t prev_byte = (c.L.lc > 0 && pos > 0) ? c.data[pos - 1] : 0;
		mgl_plan pl;
		mgl_plan_packet(&c.L, &w.st, MGL_LITERAL, 0, 1, byte, rep_byte, prev_byte, &pl);
		uint32_t ec = 0;
		if (lane < pl.nev) {
			uint32_t ctx, bit;
			mgl_plan_event(&pl, lane, &ctx, &bit);
			ec = bit_cost(probs, T, ctx, bit);
		}
		uint32_t lit = (uint32_t)wave_sum64(ec);
		if (lane == 0 && incumbent != MGL_PK_LITERAL) cand = topk_make_key(lit, 0);
		if (lane == 1 && pos > 0 && byte == rep_byte && incumbent != MGL_PK_SHORT_REP)
			cand = topk_make_key(hdr_rep + g0_0 + l_0, 1);
		topk_offer(t, cand, lane);
	}

	if (c.diag_stop == 32) return;
	/* substring_enumerator.c:85-105: nothing at the first and the last byte */
	if (pos == 0 || pos >= c.n - 1) return;
	const uint32_t bigram = ((uint32_t)walk_byte_at(w, pos) << 8) | c.data[pos + 1];
	uint32_t lo = c.bucket_off[bigram];
	const uint32_t end = c.bucket_off[bigram + 1];
	const uint32_t hi = bucket_lower_bound(c.bucket_pos, lo, end, pos, lane); /* hits are < pos */
	if (pos > c.dict_limit) lo = bucket_lower_bound(c.bucket_pos, lo, hi, pos - c.dict_limit, lane);
	if (c.max_scan && hi - lo > c.max_scan) lo = hi - c.max_scan;
	const uint32_t nhits = hi - lo;
	if (c.diag_stop == 33) { t.count += nhits & 1u; return; }
	if (nhits == 0) return;
	const uint32_t wpos = c.bucket_pos[lo]; /* the scan window: hits at positions [wpos, pos) */
	const uint32_t maxlen = (c.n - pos) < MGL_MAX_MATCH ? (c.n - pos) : MGL_MAX_MATCH;
	const uint32_t inc_type = mgl_pk_type(incumbent), inc_len = mgl_pk_len(incumbent), inc_dist = mgl_pk_dist(incumbent);

	auto price_high_lengths = [&]() {
		/* lengths >= 18 (the 8-bit high tree) are priced only when a match that long shows up */
		for (uint32_t l = 18 + lane; l <= MGL_MAX_MATCH; l += 64) {
			lencost[l - 2] = length_cost(probs, T, MGL_OFF_LEN, l, pos_state);
			lencost[272 + l - 2] = length_cost(probs, T, MGL_OFF_REP_LEN, l, pos_state);
		}
		high_ready = true;
		wave_sync();
		uint32_t a = 0xFFFFFFFFu, b = 0xFFFFFFFFu;
		for (uint32_t l = 16 + lane; l < 272; l += 64) { a = lencost[l] < a ? lencost[l] : a; b = lencost[272 + l] < b ? lencost[272 + l] : b; }
		for (int o = 32; o > 0; o >>= 1) {
			const uint32_t a2 = (uint32_t)__shfl_xor((int)a, o, 64), b2 = (uint32_t)__shfl_xor((int)b, o, 64);
			a = a2 < a ? a2 : a; b = b2 < b ? b2 : b;
		}
		minlen_m = a < minlen_m ? uni(a) : minlen_m; minlen_r = b < minlen_r ? uni(b) : minlen_r;
	};

	if (c.diag_stop == 34) return;
	/* ---- LONG_REP candidates (packet_enumerator.c:48-54: a hit whose distance is rep distance i also offers
	 * LONG_REP i at every length).  At most four hits can: the positions the four rep distances point at, when they
	 * carry this bigram and lie inside the scan window.  They are priced here, on their own (a LONG_REP's price does
	 * not depend on the distance), so the distance-ordered scans below deal with MATCH candidates only and can stop
	 * on a distance bound even when a rep distance points into the bucket. */
	{
		const uint32_t b0 = bigram >> 8, b1 = bigram & 0xFFu;
#pragma unroll
		for (uint32_t k = 0; k < 4; k++) {
			const uint32_t dk = mgl_dist_at(&w.st, k);
			if (dk >= pos) continue;
			const uint32_t q = pos - dk - 1u;
			if (q < wpos || c.data[q] != b0 || c.data[q + 1] != b1) continue;
			/* match length, substring_enumerator.c:99-103: 64 bytes per trip */
			uint32_t L = maxlen;
			for (uint32_t base = 0; base < maxlen; base += 64) {
				const uint32_t i = base + lane;
				const bool stop = i >= maxlen || c.data[q + i] != c.data[pos + i];
				const unsigned long long m = __ballot(stop);
				if (m) { const uint32_t f = base + (uint32_t)__ffsll((long long)m) - 1u; L = f < maxlen ? f : maxlen; break; }
			}
			if (L >= 18 && !high_ready) price_high_lengths();
			const uint32_t hdr = k == 0 ? hdr_lr0 : k == 1 ? hdr_lr1 : k == 2 ? hdr_lr2 : hdr_lr3;
			for (uint32_t top = L; top >= 2;) {
				const uint64_t thr = topk_threshold(t);
				const bool nolim = thr == MGL_INVALID_COST;
				const uint32_t lim = nolim ? 0u : (uint32_t)(thr >> 44) + 1u;
				if (!nolim && hdr + minlen_r >= lim * top) break; /* nothing at this or any shorter length */
				uint64_t cand = MGL_INVALID_COST;
				if (lane + 2u <= top) {
					const uint32_t len = top - lane;
					const uint32_t perp = hdr + lencost[272 + len - 2];
					if ((nolim || perp < lim * len) && !(inc_type == MGL_LONG_REP && len == inc_len && inc_dist == k)) {
						const uint64_t key = topk_make_key(perp / len, ((uint64_t)(q + 1) << 12) | ((uint64_t)len << 3) | (1u + k));
						if (key < thr) cand = key;
					}
				}
				topk_offer(t, cand, lane);
				top = top > 64u + 1u ? top - 64u : 0u;
			}
		}
	}

	/* ---- MATCH candidates: four sources, longest matches first, nearest entries first inside each.
	 *   16+ bytes  the run of the sixteen-byte order that holds this position: entries [run start, own rank) are the
	 *              earlier positions sharing 16 bytes; eight more bytes sit beside each entry, the rest is compared
	 *              in the input;
	 *   8..15      the run of the eight-byte order, entries of the deeper run skipped (their next eight bytes equal
	 *              ours); the length comes from those eight bytes alone;
	 *   4..7       the same with the four-byte order and its next four bytes;
	 *   2..3       the bigram bucket, sized from the next two bytes.
	 * Inside a source the price only grows with the distance slot, and the lengths are capped by the next source's
	 * prefix: a scan stops at the first batch whose nearest entry cannot reach the K-th best any more even at the
	 * cheapest length up to that cap and the cheapest slot from there on (a true lower bound: the selection stays exact). */
	uint64_t x8, x16, x0;
	__builtin_memcpy(&x0, c.data + pos, 8); /* bytes 0..7: byte D is what separates source D from source D + 1 */
	__builtin_memcpy(&x8, c.data + pos + 8, 8);
	__builtin_memcpy(&x16, c.data + pos + 16, 8);
	const uint32_t r8 = c.oct_rank[pos], r16 = c.hex_rank[pos];
	const uint32_t l8 = bucket_lower_bound(c.oct_pos, c.oct_run[r8], r8, wpos, lane);
	const uint32_t l16 = bucket_lower_bound(c.hex_pos, c.hex_run[r16], r16, wpos, lane);
	if (c.diag_stop == 35) return; /* diagnostic stops: 34 = before the rep pass (below), 35 = after it, 36..38 = after source 0..2 */
	/* cheapest match-length price up to the eight-byte source's cap (lengths 2..17 are priced by now) */
	uint32_t min15;
	{
		uint32_t m15 = lane < 14 ? lencost[lane] : 0xFFFFFFFFu;
		for (int o = 8; o > 0; o >>= 1) { const uint32_t d2 = (uint32_t)__shfl_xor((int)m15, o, 64); m15 = d2 < m15 ? d2 : m15; }
		min15 = uni(m15);
	}
	uint32_t lim32;
	{ const uint64_t th = topk_threshold(t); lim32 = th == MGL_INVALID_COST ? 0xFFFFFFFFu : (uint32_t)(th >> 44) + 1u; }
	/* price the survivors of a batch: per lane one hit at position hq with match length hL (have = lane holds one) */
	auto price_hits = [&](uint32_t hq, uint32_t hL, bool have) {
		if (!high_ready && __ballot(have && hL >= 18)) price_high_lengths();
		/* one table read decides for most hits: even the cheapest conceivable price at the longest
		 * length this hit offers does not reach the current K-th best.  lim32 = (K-th best cost + 1),
		 * kept across batches and refreshed after offers; costs are < 2^20 and lengths <= 273, so the
		 * products fit 32 bits (0xFFFFFFFF = no K-th best yet) */
		const uint32_t d = pos - hq - 1u;
		uint32_t slot = d, lb = 0;
		if (have) {
			if (d >= 4) { const uint32_t nlow = mgl_msb32(d) - 2; slot = nlow * 2 + (d >> nlow); }
			lb = hdr_match + minlen_m + lbslot[slot];
			if (lim32 != 0xFFFFFFFFu && lb >= lim32 * hL) have = false;
		}
		if (!__ballot(have)) return;
		uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, tail = 0;
		if (have) {
			/* distance price per length context from the tables */
			if (d >= 4) {
				const uint32_t nlow = mgl_msb32(d) - 2;
				tail = d < 128 ? disttail[d] : ((nlow - 4) << 11) + aligncost[d & 15u];
			}
			s0 = slotcost[slot]; s1 = slotcost[64 + slot]; s2 = slotcost[128 + slot]; s3 = slotcost[192 + slot];
		}
		/* Candidates of a hit: the MATCH at every length 2..L (packet_enumerator.c:48-49).  The selection is
		 * order-independent, so each lane walks its hit from the longest length down (the cheapest per byte
		 * first, which tightens the threshold at once) and stops as soon as even a lower bound of the price
		 * cannot beat the current K-th best any more.  The exact perp/len division is only done for candidates
		 * that pass the multiply test. */
		uint32_t len = hL;
		while (__ballot(have)) {
			const uint64_t thr = topk_threshold(t);
			/* a candidate can only qualify if perp/len <= thr_cost, i.e. perp < (thr_cost+1)*len */
			const bool nolim = thr == MGL_INVALID_COST;
			const uint32_t lim = nolim ? 0u : (uint32_t)(thr >> 44) + 1u;
			uint64_t cand = MGL_INVALID_COST;
			while (have) {
				if (!nolim && lb >= lim * len) { have = false; break; } /* nothing at this or any shorter length */
				const uint32_t sc = len == 2 ? s0 : len == 3 ? s1 : len == 4 ? s2 : s3;
				const uint32_t perp = hdr_match + lencost[len - 2] + sc + tail;
				const uint32_t clen = len;
				len--;
				if (len < 2) have = false;
				if (!nolim && perp >= lim * clen) continue;
				if (inc_type == MGL_MATCH && clen == inc_len && d == inc_dist) continue; /* top_k_packet_finder.c:99-101 */
				const uint64_t key = topk_make_key(perp / clen, ((uint64_t)(hq + 1) << 12) | ((uint64_t)clen << 3));
				if (key < thr) { cand = key; break; }
			}
			topk_offer(t, cand, lane);
		}
		{ const uint64_t th = topk_threshold(t); lim32 = th == MGL_INVALID_COST ? 0xFFFFFFFFu : (uint32_t)(th >> 44) + 1u; }
	};

	/* ---- the two long sources: 64 entries per trip, the next trip's entries in flight */
	for (uint32_t ph = 0; ph < 2; ph++) {
		if (c.diag_stop >= 36 && c.diag_stop <= 38 && ph > c.diag_stop - 36u) break;
		const uint32_t* spos = ph == 0 ? c.hex_pos : c.oct_pos;
		const uint64_t* snx = ph == 0 ? c.hex_nx8 : c.oct_nx8;
		const uint32_t top = ph == 0 ? r16 : r8;
		const uint32_t cnt = top - (ph == 0 ? l16 : l8);
		const uint32_t lcap = ph == 0 ? maxlen : 15u;
		uint32_t q = 0, q1 = 0;
		uint64_t y = 0, y1 = 0;
		if (lane < cnt) { q = spos[top - 1u - lane]; y = snx[top - 1u - lane]; }
		for (uint32_t hb = 0; hb < cnt; hb += 64) {
			bool have = hb + lane < cnt;
			if (hb + 64u + lane < cnt) { q1 = spos[top - 65u - hb - lane]; y1 = snx[top - 65u - hb - lane]; }
			const uint32_t cq = q;
			const uint64_t cy = y;
			q = q1; y = y1;
			{
				/* lane 0 holds the nearest entry of this batch */
				const uint32_t dn = pos - rdlane(cq, 0) - 1u;
				uint32_t sn = dn;
				if (dn >= 4) { const uint32_t nl = mgl_msb32(dn) - 2; sn = nl * 2 + (dn >> nl); }
				const uint32_t ml = ph == 0 ? minlen_m : min15;
				if (lim32 != 0xFFFFFFFFu && hdr_match + ml + sufmin[sn] >= lim32 * lcap) break;
			}
			uint32_t L = 0;
			if (have) {
				if (ph == 0) {
					const uint64_t df = x16 ^ cy;
					if (df) L = 16u + (((uint32_t)__ffsll((long long)df) - 1u) >> 3);
					else {
						/* 24 bytes match: eight bytes per step from the input (zero padded past its end) */
						L = 24;
						while (L < maxlen) {
							uint64_t x, yy;
							__builtin_memcpy(&x, c.data + pos + L, 8);
							__builtin_memcpy(&yy, c.data + cq + L, 8);
							const uint64_t df2 = x ^ yy;
							if (df2) { L += ((uint32_t)__ffsll((long long)df2) - 1u) >> 3; break; }
							L += 8;
						}
					}
				} else {
					const uint64_t df = x8 ^ cy;
					if (!df) have = false; /* 16+ bytes: priced from the deeper run */
					else L = 8u + (((uint32_t)__ffsll((long long)df) - 1u) >> 3);
				}
				if (L > maxlen) L = maxlen;
			}
			price_hits(cq, L, have);
		}
	}
	if (c.diag_stop == 36 || c.diag_stop == 37) return;

	/* ---- the exact-length sources, D = 7 down to 2: an entry of the D-byte order's run whose byte D differs from
	 * ours matches exactly D bytes (the others belong to the next source up).  These runs hold the bulk of the
	 * entries (10^4 .. 10^5 per query in a 4 MiB window of text) and almost all of them are turned away; with one
	 * length per source the pruning bound is one distance threshold, recomputed only when the K-th best moves: an
	 * entry at distance >= dthr cannot qualify at length D or shorter (hdr + cheapest length price up to D +
	 * sufmin[slot] >= lim * D, a true lower bound), and the scan of the source ends at the first such entry.  The
	 * scan is memory latency bound, so a lane takes W consecutive entries per trip (W = 4: 256 per wavefront, one
	 * 16-byte and one 4-byte load) and filters them with a subtraction and two compares each. */
	for (uint32_t D = 7; D >= 2; D--) {
		if (c.diag_stop == 38 && D < 4) break;
		const uint32_t li = D - 2u;
		const uint32_t* spos = c.xpos[li];
		const uint8_t* snxb = c.xnxb[li];
		uint32_t top, low;
		if (D == 2) { top = hi; low = lo; }
		else {
			top = c.xrank[li][pos];
			low = bucket_lower_bound(spos, c.xrun[li][top], top, wpos, lane);
		}
		const uint32_t cnt = top - low;
		if (cnt == 0) continue;
		const uint32_t xb = (uint32_t)(x0 >> (8u * D)) & 0xFFu; /* our byte D (D = 7: byte 7 is the top byte of x0) */
		const uint32_t hitlen = D < maxlen ? D : maxlen;
		/* cheapest match-length price over lengths 2..D */
		uint32_t minup;
		{
			uint32_t v = lane < D - 1u ? lencost[lane] : 0xFFFFFFFFu;
			for (int o = 4; o > 0; o >>= 1) { const uint32_t a2 = (uint32_t)__shfl_xor((int)v, o, 64); v = a2 < v ? a2 : v; }
			minup = uni(v);
		}
		uint32_t dthr = 0xFFFFFFFFu, lim_seen = 0xFFFFFFFFu;
		auto refresh_threshold = [&]() {
			lim_seen = lim32;
			dthr = 0xFFFFFFFFu;
			if (lim32 == 0xFFFFFFFFu) return;
			const uint32_t fixed = hdr_match + minup, need = lim32 * D;
			const unsigned long long m = __ballot(need <= fixed || sufmin[lane] >= need - fixed);
			if (m) {
				const uint32_t sl = (uint32_t)__ffsll((long long)m) - 1u;
				dthr = sl < 4u ? sl : ((2u | (sl & 1u)) << ((sl >> 1) - 1u)); /* first distance of that slot */
			}
		};
		/* entries hb + W lane + r, r = 0..W-1 (r = 0 nearest): array indices top - 1 - e, i.e. the W words at
		 * top - W - hb - W lane, the nearest last */
		auto loadw = [&](uint32_t hb, uint32_t qv[W], uint32_t& yv) {
			const uint32_t e0 = hb + (uint32_t)W * lane;
			if (W == 4 && e0 + 3u < cnt) {
				const uint32_t base = top - 4u - e0;
				uint4 qq;
				__builtin_memcpy(&qq, spos + base, 16);
				qv[0] = qq.w; qv[1 % W] = qq.z; qv[2 % W] = qq.y; qv[3 % W] = qq.x;
				uint32_t yy;
				__builtin_memcpy(&yy, snxb + base, 4);
				yv = __builtin_bswap32(yy); /* byte r of yv = entry r */
			} else {
				yv = 0;
#pragma unroll
				for (uint32_t r = 0; r < (uint32_t)W; r++) {
					const bool in = e0 + r < cnt;
					const uint32_t idx = in ? top - 1u - (e0 + r) : top - 1u;
					qv[r] = in ? spos[idx] : pos; /* distance "-1": never passes the filter */
					yv |= (uint32_t)snxb[idx] << (8u * r);
				}
			}
		};
		refresh_threshold();
		uint32_t qa[W], qb[W], ya, yb = 0;
#pragma unroll
		for (uint32_t r = 0; r < (uint32_t)W; r++) qb[r] = 0;
		loadw(0, qa, ya);
		for (uint32_t hb = 0; hb < cnt; hb += 64u * W) {
			if (hb + 64u * W < cnt) loadw(hb + 64u * W, qb, yb); /* the next trip, in flight */
			if (lim_seen != lim32) refresh_threshold();
			/* lane 0's first entry is the nearest of this trip */
			if (pos - rdlane(qa[0], 0) - 1u >= dthr) break;
			uint32_t prmask = 0; /* bit r: entry r of this lane passes the filter */
#pragma unroll
			for (uint32_t r = 0; r < (uint32_t)W; r++) {
				const uint32_t d = pos - qa[r] - 1u; /* out-of-range slots hold q = pos: d = 0xFFFFFFFF */
				/* byte D equal: the match goes on: priced from the next source up */
				const bool p = ((ya >> (8u * r)) & 0xFFu) != xb && d < dthr && d != 0xFFFFFFFFu;
				prmask |= (p ? 1u : 0u) << r;
			}
			if (c.diag_stop == 39) prmask = 0; /* diagnostic: the filter alone */
			/* one copy of the pricing code: the (rare) survivors of the W entries take turns */
			for (uint32_t r = 0; r < (uint32_t)W && __ballot(prmask != 0); r++) {
				const bool p = (prmask >> r) & 1u;
				if (!__ballot(p)) continue;
				uint32_t q_r = qa[0];
#pragma unroll
				for (uint32_t k = 1; k < (uint32_t)W; k++) q_r = r == k ? qa[k] : q_r;
				price_hits(q_r, hitlen, p);
				prmask &= ~(1u << r);
			}
#pragma unroll
			for (uint32_t r = 0; r < (uint32_t)W; r++) qa[r] = qb[r];
			ya = yb;
		}
	}
}

/* decode a top-K key back into a packet */
__device__ __forceinline__ mgl_pk topk_packet(uint64_t key, uint32_t pos)
{
	uint64_t seq = MGL_SEQ_MASK - (key & MGL_SEQ_MASK);
	if (seq == 0) return MGL_PK_LITERAL;
	if (seq == 1) return MGL_PK_SHORT_REP;
	uint32_t q1 = (uint32_t)(seq >> 12), len = (uint32_t)(seq >> 3) & 0x1FFu, kind = (uint32_t)seq & 7u;
	if (kind == 0) return mgl_pack(MGL_MATCH, pos - q1, len);
	return mgl_pack(MGL_LONG_REP, kind - 1, len);
}

/* ================================================================== checkpoints */

__device__ __forceinline__ void ckpt_store(const BaseView& b, const DevCtx& c, uint32_t ci, const uint16_t* probs,
                                           const Walk& w, uint64_t cum, uint32_t lane)
{
	uint32_t* dst = (uint32_t*)(b.ckpt_probs + (size_t)ci * b.ckpt_elems);
	const uint32_t* src = (const uint32_t*)probs;
	for (uint32_t i = lane; i < b.ckpt_elems / 2; i += 64) dst[i] = src[i];
	if (lane == 0) {
		CkptHdr h;
		h.pos = w.st.pos; h.ctx_state = w.st.ctx_state;
		h.dists[0] = w.st.dists[0]; h.dists[1] = w.st.dists[1]; h.dists[2] = w.st.dists[2]; h.dists[3] = w.st.dists[3];
		h.ordinal = w.packets; h.pad = 0; h.cum = cum;
		b.ckpt_hdr[ci] = h;
	}
}
__device__ __forceinline__ uint64_t ckpt_load(const BaseView& b, const DevCtx& c, uint32_t ci, uint16_t* probs,
                                              Walk& w, uint32_t lane)
{
	const uint32_t* src = (const uint32_t*)(b.ckpt_probs + (size_t)ci * b.ckpt_elems);
	uint32_t* dst = (uint32_t*)probs;
	for (uint32_t i = lane; i < b.ckpt_elems / 2; i += 64) dst[i] = src[i];
	const CkptHdr* h = &b.ckpt_hdr[ci];
	walk_reset(w);
	w.st.pos = uni(h->pos); w.st.ctx_state = uni(h->ctx_state);
	w.st.dists[0] = uni(h->dists[0]); w.st.dists[1] = uni(h->dists[1]);
	w.st.dists[2] = uni(h->dists[2]); w.st.dists[3] = uni(h->dists[3]);
	w.packets = uni(h->ordinal);
	uint64_t cum = uni64(h->cum);
	wave_sync();
	return cum;
}

/* ================================================================== k_rebuild */
/* One wavefront walks the base slab from the checkpoint that precedes the first changed
 * position (or from byte 0) to the end: rewrites the on-walk bitmap and the prefix
 * checkpoints, counts packets and produces the slab's exact total cost.
 * cum_out (nullable): running total after every packet (parity hook). */
__global__ void __launch_bounds__(64) k_rebuild(DevCtx c, BaseView b, Control* ctl, int from_dirty, uint64_t* cum_out,
                                                uint16_t* final_probs)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint16_t* T = (uint16_t*)smem;
	uint16_t* probs = (uint16_t*)(smem + 4096);
	const uint32_t lane = threadIdx.x;
	for (uint32_t i = lane; i < 2048; i += 64) T[i] = c.cost_tbl[i];

	Walk w;
	uint64_t base_cum = 0;
	uint32_t ci = 0;
	if (from_dirty) {
		if (!ctl->accepted_flag) return;
		ci = uni(ctl->dirty_pos) >> MGL_CKPT_SHIFT;
	}
	if (ci == 0) {
		for (uint32_t i = lane; i < b.ckpt_elems; i += 64) probs[i] = MGL_PROB_INIT;
		walk_reset(w);
		wave_sync();
	} else {
		base_cum = ckpt_load(b, c, ci, probs, w, lane);
	}
	uint32_t next_ck = ci; /* next checkpoint index to (re)write: covers positions >= next_ck << SHIFT */
	/* on-walk bitmap: keep the bits below the restart position in its word */
	uint32_t word = w.st.pos >> 6;
	uint64_t bits = 0;
	if (w.st.pos & 63u) bits = b.onwalk[word] & ((1ull << (w.st.pos & 63u)) - 1ull);

	uint32_t guard = 0;
	while (w.st.pos < c.n) {
		const uint32_t pos = w.st.pos;
		if (++guard > c.n) { if (lane == 0) atomicOr(&ctl->error_flags, MGL_ERR_WALK_OVERRUN); break; }
		while (next_ck < b.nckpt && (next_ck << MGL_CKPT_SHIFT) <= pos) {
			uint64_t cum = base_cum + wave_sum64(w.acc);
			ckpt_store(b, c, next_ck, probs, w, cum, lane);
			next_ck++;
		}
		/* flush finished bitmap words */
		const uint32_t pw = pos >> 6;
		if (pw != word) {
			if (lane == 0) b.onwalk[word] = bits;
			for (uint32_t z = word + 1 + lane; z < pw; z += 64) b.onwalk[z] = 0;
			word = pw; bits = 0;
		}
		bits |= 1ull << (pos & 63u);
		walk_window(w, c, b.slab, lane);
		const mgl_pk pk = walk_slab_at(w, pos);
		uint32_t type = mgl_pk_type(pk), len = mgl_pk_len(pk), dist = mgl_pk_dist(pk);
		if (type < MGL_LITERAL || type > MGL_LONG_REP || len == 0 || pos + len > c.n) { /* corrupt entry: cost as literal */
			type = MGL_LITERAL; len = 1; dist = 0;
			if (lane == 0) atomicOr(&ctl->error_flags, MGL_ERR_WALK_OVERRUN);
		}
		walk_packet<true>(w, c, probs, T, type, dist, len, lane);
		if (cum_out) {
			uint64_t cum = base_cum + wave_sum64(w.acc);
			if (lane == 0) cum_out[w.packets - 1] = cum;
		}
	}
	if (lane == 0) b.onwalk[word] = bits;
	{
		const uint32_t nwords = (c.n + 63u) >> 6;
		for (uint32_t z = word + 1 + lane; z < nwords; z += 64) b.onwalk[z] = 0;
	}
	const uint64_t total = base_cum + wave_sum64(w.acc);
	wave_sync();
	if (final_probs) for (uint32_t i = lane; i < c.L.total; i += 64) final_probs[i] = probs[i];
	if (lane == 0) {
		ctl->packets = w.packets;
		ctl->rebuild_cost = total;
		ctl->final_ctx_state = w.st.ctx_state;
		ctl->final_dists[0] = w.st.dists[0]; ctl->final_dists[1] = w.st.dists[1];
		ctl->final_dists[2] = w.st.dists[2]; ctl->final_dists[3] = w.st.dists[3];
		if (from_dirty) {
			if (ctl->cur_cost != total) atomicOr(&ctl->error_flags, MGL_ERR_REBUILD_MISMATCH);
			ctl->accepted_flag = 0;
		}
	}
}

/* ================================================================== k_neighbours */

struct Journal {
	uint32_t* pos; /* LDS, MGL_MAX_DIFFS */
	mgl_pk* old;
	mgl_pk* neu;
	uint32_t count; /* uniform */
	bool overflow;
};
__device__ __forceinline__ void journal_set(Journal& jn, uint32_t pos, mgl_pk old, mgl_pk neu, uint32_t lane)
{
	/* the mutated pair (first two entries) may be touched again by the repair */
	for (uint32_t i = 0; i < jn.count && i < 2; i++) {
		if (jn.pos[i] == pos) { if (lane == 0) jn.neu[i] = neu; wave_sync(); return; }
	}
	if (jn.count >= MGL_MAX_DIFFS) { jn.overflow = true; return; }
	if (lane == 0) { jn.pos[jn.count] = pos; jn.old[jn.count] = old; jn.neu[jn.count] = neu; }
	jn.count++;
	wave_sync();
}

struct NbrRng { uint64_t key; uint32_t n; };
__device__ __forceinline__ uint32_t nbr_draw(NbrRng& r) { return mgl_rng_draw(r.key, r.n++); }

/* packet_slab_neighbour.c:56-72 with the canonical top-K order */
template <int W>
__device__ bool pick_from_top_k(const DevCtx& c, const Walk& w, const uint16_t* probs, const uint16_t* T, uint32_t* lencost,
                                mgl_pk incumbent, bool best, NbrRng& rng, uint32_t lane, mgl_pk* picked)
{
	TopK t;
	topk_find<W>(t, c, w, probs, T, lencost, incumbent, lane);
	const uint32_t count = t.count;
	if (count == 0) return false;
	uint32_t choice = nbr_draw(rng) % count; /* :48-54 max of 8 draws */
	for (int i = 0; i < 7; i++) { uint32_t x = nbr_draw(rng) % count; choice = x > choice ? x : choice; }
	if (nbr_draw(rng) % 8u == 0 || best) choice = count - 1;
	/* pop order is worst first: the (choice+1)-th pop is rank count-1-choice from the best */
	uint64_t key = shfl64(t.key, (int)(count - 1 - choice));
	*picked = topk_packet(key, w.st.pos);
	return true;
}

/* packet_slab_neighbour.c:74-80: memcmp over the rep source, all lanes compare */
__device__ __forceinline__ bool long_rep_ok(const DevCtx& c, const Walk& w, uint32_t idx, uint32_t len, uint32_t lane)
{
	const uint32_t rd = mgl_dist_at(&w.st, idx);
	if (rd >= w.st.pos || w.st.pos + len > c.n) return false;
	const uint32_t src = w.st.pos - rd - 1u;
	bool bad = false;
	for (uint32_t i = lane; i < len; i += 64) bad |= c.data[src + i] != c.data[w.st.pos + i];
	return __ballot(bad) == 0;
}

/* The neighbour's packet at p: its journal entry if it has one, else the base slab's. */
__device__ __forceinline__ mgl_pk journal_or_base(const Journal& jn, const mgl_pk* slab, uint32_t p, uint32_t lane)
{
	const bool hit = lane < jn.count && jn.pos[lane] == p; /* MGL_MAX_DIFFS == 64: one entry per lane */
	const unsigned long long m = __ballot(hit);
	if (m) return jn.neu[(uint32_t)__ffsll((long long)m) - 1u];
	return slab[p];
}
/* End of the neighbour's window: the first position at which its walk and the base's stand on the
 * same byte with the same ctx_state and rep distances, at least three repair packets after the
 * mutated one (the incremental kernel stops its two-pointer walk there).  The
 * full-walk engine has walked to the end of the file, so it finds the point afterwards from the
 * journal.  `st` = walk state at the target (uniform). */
/* n literal transitions of the ctx_state automaton (3 reach 0 from anywhere) */
__device__ __forceinline__ uint32_t lit_steps_(uint32_t s, uint32_t n)
{
	if (n > 3) n = 3;
	for (uint32_t i = 0; i < n; i++) s = mgl_next_ctx_state(s, MGL_LITERAL);
	return s;
}
struct WinInfo { uint32_t end, soft, n_ins, n_rem, walked; };
/* events of one packet (mgl_plan_packet's nev: independent of the bytes) */
__device__ __forceinline__ uint32_t packet_events(const DevCtx& c, const mgl_wstate& st, mgl_pk pk)
{
	mgl_plan pl;
	mgl_plan_packet(&c.L, &st, mgl_pk_type(pk), mgl_pk_dist(pk), mgl_pk_len(pk), 0u, 0u, 0u, &pl);
	return pl.nev;
}
__device__ WinInfo window_end_from_journal(const DevCtx& c, const mgl_pk* slab, const Journal& jn, mgl_wstate st, uint32_t lane)
{
	mgl_wstate nb = st, bs = st;
	uint32_t count = 0, wsoft = 0xFFFFFFFFu, taint = 0xFu, dep = 0u, wend = c.n, n_ins = 0, n_rem = 0, walked = 0;
	bool first = true;
	for (;;) {
		if (!first && nb.pos == bs.pos && count >= 3 && nb.ctx_state == bs.ctx_state) {
			if (wsoft == 0xFFFFFFFFu) wsoft = nb.pos;
			if (nb.dists[0] == bs.dists[0] && nb.dists[1] == bs.dists[1] && nb.dists[2] == bs.dists[2] && nb.dists[3] == bs.dists[3]) { wend = nb.pos; break; }
			if (nb.ctx_state < 7u && nb.pos < c.n) {
				/* plain literals up to the base's next non-literal packet: skipped, not visited (the incremental kernel's jump) */
				uint32_t sx = nb.pos;
				while (sx < c.n && mgl_pk_type(uni64(slab[sx])) == MGL_LITERAL) sx++;
				if (sx > nb.pos) {
					const uint32_t cs = lit_steps_(nb.ctx_state, sx - nb.pos);
					nb.pos = bs.pos = sx; nb.ctx_state = bs.ctx_state = cs;
					count = 8;
					continue;
				}
			}
		}
		if (nb.pos >= c.n && bs.pos >= c.n) { wend = c.n; break; }
		if (walked > MGL_MAX_WALK) break;
		if (nb.pos <= bs.pos && nb.pos < c.n) {
			if (!first && count < 8) count++;
			first = false;
			walked++;
			const uint32_t p = nb.pos;
			const mgl_pk pk = uni64(journal_or_base(jn, slab, p, lane));
			const uint32_t ntype = mgl_pk_type(pk), ndist = mgl_pk_dist(pk);
			if (count > 0 && pk != uni64(slab[p])) wsoft = 0xFFFFFFFFu; /* a changed packet: the soft window reaches behind it */
			if (ntype == MGL_SHORT_REP || ntype == MGL_LONG_REP) {
				/* not when it is the base's own packet reading a slot that holds the same distance in both walks */
				const uint32_t slot = ntype == MGL_SHORT_REP ? 0u : ndist;
				const bool same_read = bs.pos == p && uni64(slab[p]) == pk && mgl_dist_at(&nb, slot) == mgl_dist_at(&bs, slot);
				if (!same_read) wsoft = 0xFFFFFFFFu;
				dep |= ntype == MGL_SHORT_REP ? (taint & 1u) : ((taint >> ndist) & 1u);
			}
			if (ntype == MGL_MATCH) taint = (taint << 1) & 0xFu;
			else if (ntype == MGL_LONG_REP) taint = (taint & ~((2u << ndist) - 1u)) | ((taint & ((1u << ndist) - 1u)) << 1) | ((taint >> ndist) & 1u);
			/* identical coding of the base packet at the same position cancels (the rule of the incremental kernel) */
			const bool paired = bs.pos == p;
			mgl_pk bpk = pk;
			bool cancelled = false;
			if (paired) {
				bpk = uni64(slab[p]);
				cancelled = bpk == pk && nb.ctx_state == bs.ctx_state;
				if (cancelled && ntype == MGL_LITERAL && nb.ctx_state >= 7) {
					const uint32_t mn = nb.dists[0] < p ? c.data[p - nb.dists[0] - 1] : 0u;
					const uint32_t mb = bs.dists[0] < p ? c.data[p - bs.dists[0] - 1] : 0u;
					cancelled = mn == mb;
				}
			}
			if (!cancelled) {
				n_ins += packet_events(c, nb, pk);
				if (paired) n_rem += packet_events(c, bs, bpk);
			}
			if (paired) mgl_advance(&bs, mgl_pk_type(bpk), mgl_pk_dist(bpk), mgl_pk_len(bpk));
			mgl_advance(&nb, ntype, ndist, mgl_pk_len(pk));
		} else {
			const mgl_pk pk = uni64(slab[bs.pos]);
			n_rem += packet_events(c, bs, pk);
			mgl_advance(&bs, mgl_pk_type(pk), mgl_pk_dist(pk), mgl_pk_len(pk));
		}
	}
	WinInfo wi;
	wi.n_ins = n_ins; wi.n_rem = n_rem; wi.walked = walked;
	wi.end = wend;
	wi.soft = (wsoft < wend ? wsoft : wend) | (dep << 31);
	return wi;
}

/* One wavefront = one neighbour of the base slab (packet_slab_neighbour.c:154-173). */
/* todo != nullptr: only the neighbours listed there are evaluated (the ones the incremental
 * kernel could not fit), walking from byte 0 instead of from a prefix checkpoint. */
__device__ void nbr_fullwalk_one(const DevCtx& c, const BaseView& b, const Control* ctl, uint64_t seed, uint64_t step_override, uint32_t K,
                                 const NbrOut& out, uint32_t per_wave_bytes, bool from_zero, unsigned char* smem, const uint16_t* T, uint32_t j,
                                 uint32_t lane, uint32_t wid)
{
	if (j >= K) return;
	unsigned char* mine = smem + 4096 + (size_t)wid * per_wave_bytes;
	uint16_t* probs = (uint16_t*)mine;
	uint32_t* lencost = (uint32_t*)(mine + (size_t)b.ckpt_elems * 2);
	Journal jn;
	jn.old = (mgl_pk*)(lencost + MGL_PRICE_WORDS);
	jn.neu = jn.old + MGL_MAX_DIFFS;
	jn.pos = (uint32_t*)(jn.neu + MGL_MAX_DIFFS);
	jn.count = 0; jn.overflow = false;

	const uint64_t gstep = step_override != ~0ull ? step_override : ctl->gstep;
	NbrRng rng; rng.key = mgl_rng_key(seed, gstep, j); rng.n = 0;

	/* target: uniform over the packets of the walk by rejection sampling on positions
	 * (the reference draws a packet ordinal, packet_slab_neighbour.c:162-163) */
	uint32_t target;
	if (c.strat_pre != nullptr) {
		target = uni(c.strat_tgt[j]); /* stratified_target(), worked out for the whole step by k_targets */
		rng.n = 1;
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
			/* next on-walk position at or after p, else 0 */
			const uint32_t nwords = (c.n + 63u) >> 6;
			uint32_t wd = p >> 6;
			uint64_t bits = b.onwalk[wd] & (~0ull << (p & 63u));
			while (!bits && ++wd < nwords) bits = b.onwalk[wd];
			target = bits ? (wd << 6) + (uint32_t)__ffsll((long long)bits) - 1u : 0u;
			target = uni(target);
		}
	}

	/* prefix: nearest checkpoint, then the unchanged packets up to the target (:165) */
	Walk w;
	uint64_t base_cum = 0;
	if (from_zero) {
		for (uint32_t i = lane; i < b.ckpt_elems; i += 64) probs[i] = MGL_PROB_INIT;
		walk_reset(w);
		wave_sync();
	} else {
		base_cum = ckpt_load(b, c, target >> MGL_CKPT_SHIFT, probs, w, lane);
	}
	const uint32_t first_packet = w.packets;
	while (w.st.pos < target) {
		walk_window(w, c, b.slab, lane);
		const mgl_pk pk = walk_slab_at(w, w.st.pos);
		walk_packet<true>(w, c, probs, T, mgl_pk_type(pk), mgl_pk_dist(pk), mgl_pk_len(pk), lane);
	}
	if (w.st.pos != target) { /* cannot happen for an on-walk target; fail safe */
		if (lane == 0) { out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = 0; out.win[2u * j] = target; out.win[2u * j + 1u] = MGL_WIN_NONE; }
		return;
	}
	const mgl_wstate st_target = w.st;

	/* mutate, packet_slab_neighbour.c:119-152 */
	const uint32_t pos = target;
	walk_window(w, c, b.slab, lane);
	const mgl_pk first = walk_slab_at(w, pos);
	mgl_pk m_first = first, m_second = 0;
	bool second_set = false, mutated = false;
	if (pos + 1 < c.n && (nbr_draw(rng) % 2u) == 0) {
		const mgl_pk second = uni64(b.slab[pos + 1]);
		const uint32_t ft = mgl_pk_type(first), flen = mgl_pk_len(first);
		const uint32_t st = mgl_pk_type(second), slen = mgl_pk_len(second), sdist = mgl_pk_dist(second);
		if ((ft == MGL_LONG_REP || ft == MGL_MATCH) && flen > 2) {
			m_second = mgl_pack(ft, mgl_pk_dist(first), flen - 1);
			m_first = MGL_PK_LITERAL;
			journal_set(jn, pos, first, m_first, lane);
			journal_set(jn, pos + 1, second, m_second, lane);
			second_set = true; mutated = true;
		} else if ((ft == MGL_LITERAL || ft == MGL_SHORT_REP) && (st == MGL_MATCH || st == MGL_LONG_REP)) {
			uint32_t rep_start = pos - (st == MGL_LONG_REP ? mgl_dist_at(&w.st, sdist) : sdist);
			if (slen < MGL_MAX_MATCH && rep_start > 0 && rep_start <= pos &&
			    walk_byte_at(w, pos) == c.data[rep_start - 1]) {
				m_first = mgl_pack(st, sdist, slen + 1);
				journal_set(jn, pos, first, m_first, lane);
				mutated = true;
			}
		}
	}
	if (!mutated) {
		mgl_pk picked;
		if (!pick_from_top_k<1>(c, w, probs, T, lencost, first, false, rng, lane, &picked)) {
			if (lane == 0) { out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = 0; out.win[2u * j] = target; out.win[2u * j + 1u] = MGL_WIN_NONE; }
			return;
		}
		m_first = picked;
		journal_set(jn, pos, first, m_first, lane);
	}
	walk_packet<true>(w, c, probs, T, mgl_pk_type(m_first), mgl_pk_dist(m_first), mgl_pk_len(m_first), lane); /* :169 */

	/* repair_remaining_packets, packet_slab_neighbour.c:82-117 */
	uint32_t count = 0, guard = 0, npicks = 0;
	while (w.st.pos < c.n && !jn.overflow) {
		if (++guard > c.n) break;
		count++;
		const uint32_t p = w.st.pos;
		walk_window(w, c, b.slab, lane);
		const mgl_pk old = (second_set && p == pos + 1) ? m_second : walk_slab_at(w, p);
		mgl_pk pk = old;
		uint32_t type = mgl_pk_type(pk);
		if (type == MGL_SHORT_REP || (type == MGL_LITERAL && count < 4)) {
			/* for a LITERAL past the third packet the byte test cannot change anything (:91-97) */
			const bool same = w.st.dists[0] < p && walk_byte_at(w, p) == c.data[p - w.st.dists[0] - 1];
			if (same) { if (count < 4) pk = MGL_PK_SHORT_REP; }
			else pk = MGL_PK_LITERAL;
		}
		type = mgl_pk_type(pk);
		if (type == MGL_LONG_REP) {
			const uint32_t len = mgl_pk_len(pk);
			uint32_t idx = mgl_pk_dist(pk);
			bool ok = long_rep_ok(c, w, idx, len, lane);
			for (uint32_t i = 0; i < 4 && !ok; i++) { idx = i; ok = long_rep_ok(c, w, idx, len, lane); }
			pk = mgl_pack(MGL_LONG_REP, idx, len);
			if (!ok) {
				npicks++;
				const bool best = (nbr_draw(rng) % 4u) == 0;
				mgl_pk picked;
				if (pick_from_top_k<1>(c, w, probs, T, lencost, pk, best, rng, lane, &picked)) pk = picked;
			}
		}
		if (pk != old) {
			const mgl_pk base_old = (second_set && p == pos + 1) ? uni64(b.slab[p]) : old;
			journal_set(jn, p, base_old, pk, lane);
		}
		walk_packet<true>(w, c, probs, T, mgl_pk_type(pk), mgl_pk_dist(pk), mgl_pk_len(pk), lane);
	}

	const uint64_t total = base_cum + wave_sum64(w.acc);
	wave_sync();
	if (jn.overflow) {
		if (lane == 0) { out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = w.packets - first_packet; out.win[2u * j] = target; out.win[2u * j + 1u] = MGL_WIN_DROPPED; }
		return;
	}
	const WinInfo wi = window_end_from_journal(c, b.slab, jn, st_target, lane);
	const uint32_t wend = wi.end;
	if (wi.n_ins > 4096u || wi.n_rem > 4096u || wi.walked > MGL_MAX_WALK || npicks > MGL_MAX_REPAIR_PICKS) { /* MGL_BIG_CAP / MGL_MAX_WALK / MGL_MAX_REPAIR_PICKS: what the incremental engine's lists hold, its walk visits, its repair picks */
		if (lane == 0) { out.cost[j] = MGL_INVALID_COST; out.ndiffs[j] = 0; out.walked[j] = w.packets - first_packet; out.win[2u * j] = target; out.win[2u * j + 1u] = MGL_WIN_DROPPED; }
		return;
	}
	/* journal out: entries whose final value equals the base value are dropped */
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
	if (lane == 0) { out.cost[j] = total; out.ndiffs[j] = nd; out.walked[j] = w.packets - first_packet; out.win[2u * j] = target; out.win[2u * j + 1u] = wend; out.win2[j] = wi.soft; }
}

/* todo != nullptr: only the neighbours listed there are evaluated (the ones the incremental kernels could not
 * fit), walking from byte 0 instead of from a prefix checkpoint; the launch is a small grid that strides over
 * the list (the host does not know its length: empty in practice, and then a workgroup costs one load). */
__global__ void __launch_bounds__(256) k_neighbours(DevCtx c, BaseView b, const Control* ctl, uint64_t seed,
                                                    uint64_t step_override, uint32_t K, NbrOut out, uint32_t per_wave_bytes,
                                                    const uint32_t* todo, const uint32_t* todo_count)
{
	const uint32_t waves = blockDim.x >> 6;
	if (todo && blockIdx.x * waves >= *todo_count) return; /* nothing listed for this workgroup: before any load */
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint16_t* T = (uint16_t*)smem;
	for (uint32_t i = threadIdx.x; i < 2048; i += blockDim.x) T[i] = c.cost_tbl[i];
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
	if (todo) {
		const uint32_t n = *todo_count;
		for (uint32_t slot = blockIdx.x * waves + wid; slot < n; slot += gridDim.x * waves)
			nbr_fullwalk_one(c, b, ctl, seed, step_override, K, out, per_wave_bytes, true, smem, T, uni(todo[slot]), lane, wid);
	} else {
		nbr_fullwalk_one(c, b, ctl, seed, step_override, K, out, per_wave_bytes, false, smem, T, blockIdx.x * waves + wid, lane, wid);
	}
}

/* main.c:91 -- the new best slab (after the journal has been applied) */
__global__ void k_copy_best(const Control* ctl, const mgl_pk* slab, mgl_pk* best, uint32_t n)
{
	if (!ctl->copy_best_flag) return;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) best[i] = slab[i];
}

/* ================================================================== hooks & layout */

/* 12-byte reference records (lzma_packet.h:13-17) <-> packed 8-byte entries */
__global__ void k_import(const uint32_t* aos /* 3 dwords per record */, mgl_pk* slab, uint32_t n)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t t = aos[3 * i] & 0xFFu, d = aos[3 * i + 1], l = aos[3 * i + 2] & 0xFFFFu;
		slab[i] = mgl_pack(t, d, l);
	}
}
__global__ void k_export(const mgl_pk* slab, uint32_t* aos, uint32_t n)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const mgl_pk p = slab[i];
		aos[3 * i] = mgl_pk_type(p); aos[3 * i + 1] = mgl_pk_dist(p); aos[3 * i + 2] = mgl_pk_len(p);
	}
}
__global__ void k_fill_literal(mgl_pk* slab, uint32_t n)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) slab[i] = MGL_PK_LITERAL;
}

/* top-K at an arbitrary on-walk position of a (scratch) base; list written worst first */
__global__ void __launch_bounds__(64) k_topk_probe(DevCtx c, BaseView b, uint32_t position, mgl_pk* out_pk, uint64_t* out_cost,
                                                   uint32_t* out_count)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint16_t* T = (uint16_t*)smem;
	uint16_t* probs = (uint16_t*)(smem + 4096);
	uint32_t* lencost = (uint32_t*)(smem + 4096 + (size_t)b.ckpt_elems * 2);
	const uint32_t lane = threadIdx.x;
	for (uint32_t i = lane; i < 2048; i += 64) T[i] = c.cost_tbl[i];
	wave_sync();
	if (position >= c.n || !((b.onwalk[position >> 6] >> (position & 63u)) & 1ull)) {
		if (lane == 0) *out_count = ~0u;
		return;
	}
	Walk w;
	ckpt_load(b, c, position >> MGL_CKPT_SHIFT, probs, w, lane);
	while (w.st.pos < position) {
		walk_window(w, c, b.slab, lane);
		const mgl_pk pk = walk_slab_at(w, w.st.pos);
		walk_packet<true>(w, c, probs, T, mgl_pk_type(pk), mgl_pk_dist(pk), mgl_pk_len(pk), lane);
	}
	walk_window(w, c, b.slab, lane);
	TopK t;
	topk_find<4>(t, c, w, probs, T, lencost, walk_slab_at(w, position), lane);
	if (lane < t.count) {
		const uint32_t o = t.count - 1 - lane; /* worst first */
		out_pk[o] = topk_packet(t.key, position);
		out_cost[o] = t.key >> 44;
	}
	if (lane == 0) *out_count = t.count;
}

/* substring_enumerator_for_each, one thread, reference callback order (parity hook) */
__global__ void k_substrings(DevCtx c, uint32_t pos, uint32_t max_len, uint32_t* offs, uint32_t* lens, uint32_t cap,
                             uint32_t* count)
{
	uint32_t k = 0;
	if (pos != 0 && pos != c.n - 1 && pos < c.n) {
		const uint32_t bg = ((uint32_t)c.data[pos] << 8) | c.data[pos + 1];
		for (uint32_t i = c.bucket_off[bg]; i < c.bucket_off[bg + 1]; i++) {
			const uint32_t q = c.bucket_pos[i];
			if (q >= pos) break;
			if (pos - q - 1 >= c.dict_limit) continue;
			if (k < cap) { offs[k] = q; lens[k] = 2; }
			k++;
			for (uint32_t j = 2; j < max_len && j + pos < c.n; j++) {
				if (c.data[pos + j] != c.data[q + j]) break;
				if (k < cap) { offs[k] = q; lens[k] = j + 1; }
				k++;
			}
		}
	}
	*count = k;
}
