/*
 * mgl_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see mgl_oracle.h).
 *
 * Restates, in one file and with a flat probability array, the arithmetic of the
 * reference's hot path.  "ref:" comments give the reference file:line each block follows
 * (paths relative to /root/reference/src/).
 *
 * Parity status: PINNED -- bit-exact against the compiled reference (oracle/_ref) for
 * per-packet cumulative costs, final model state, top-K pop sequences, match-index
 * callbacks, whole SA trajectories under glibc rand(), and emitted .lzma bytes
 * (tests/test_oracle_vs_ref.py), and against tests/golden/ fixtures made from it.
 * lc/lp/pb != 0 has no reference implementation (lzma_packet_encoder.c:17,44,113 hard-code
 * 0): that extension is "parity unpinned" and is checked through xz/liblzma round trips.
 */
#include "mgl_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- cost table */
/* ref: generate_table.py:7-10 -- T[0]=0, T[i] = -int(log2(i/2048)*2048). */
static uint16_t g_cost[2048];
static int g_cost_ready = 0;
const uint16_t* orc_cost_table(void)
{
	if (!g_cost_ready) {
		g_cost[0] = 0;
		for (int i = 1; i < 2048; i++) {
			g_cost[i] = (uint16_t)(-(int)(log2(i / 2048.) * 2048));
		}
		g_cost_ready = 1;
	}
	return g_cost;
}

/* ---------------------------------------------------------------- model layout */
/* ref: lzma_state.h:15-58.  Same order as the reference's struct so a memcpy of its
 * LZMAProbabilityModel compares directly when lc=lp=0: lit | len | rep_len | dist | ctx. */
#define LEN_CHOICE1 0
#define LEN_CHOICE2 1
#define LEN_LOW 2      /* [16][8] */
#define LEN_MID 130    /* [16][8] */
#define LEN_HIGH 258   /* [256] */
#define LEN_SIZE 514
#define DIST_SLOT 0    /* [4][64] */
#define DIST_ALIGN 256 /* [16] */
#define DIST_POS 272   /* [115] */
#define DIST_SIZE 387
#define CS_IS_MATCH 0  /* [12<<4] */
#define CS_IS_REP 192
#define CS_G0 204
#define CS_G1 216
#define CS_G2 228
#define CS_REP0_LONG 240 /* [12<<4] */
#define CS_SIZE 432

typedef struct {
	uint32_t lit, len, rep_len, dist, cs, total;
} orc_layout;

typedef struct {
	uint16_t* p;
	uint8_t ctx_state;
	uint32_t dists[4];
	size_t pos;
} orc_state;

struct orc_ctx {
	const uint8_t* data;
	size_t n;
	int lc, lp, pb;
	uint32_t dict_limit;
	orc_layout L;
	/* match index, ref: substring_enumerator.c:9-47 */
	uint32_t* bucket_off; /* 65536+1 */
	uint32_t* bucket_pos; /* n-1 positions, ascending inside a bucket */
	uint64_t temperature; /* 0 = the reference's accept rule; else the opt-in Metropolis rule of orc_sa_batched */
	uint32_t max_bucket_scan; /* 0 = every hit (the reference); else only the nearest M hits of the window */
	uint32_t strata;          /* batched mode: 0 = the target comes from position draws; K = neighbour j takes a packet of the j-th
	                           * of K equal slices of the walk's packets (the device's default, mgl_device.h:stratified_target) */
};

void orc_set_temperature(orc_ctx* c, uint64_t temperature) { c->temperature = temperature; }
void orc_set_max_bucket_scan(orc_ctx* c, uint32_t m) { c->max_bucket_scan = m; }
void orc_set_strata(orc_ctx* c, uint32_t K) { c->strata = K; }

size_t orc_num_probs(const orc_ctx* c) { return c->L.total; }

/* ---------------------------------------------------------------- bit sinks */
/* ref: encoder_interface.h:9-13 -- two callbacks; here a tagged struct. */
typedef struct {
	int kind; /* 0 perplexity, 1 range coder */
	int readonly; /* cost only: leave the probabilities untouched */
	uint64_t perp;
	/* range coder, ref: range_encoder.c:10-16 */
	uint64_t low;
	uint32_t range;
	uint8_t cache;
	uint64_t cache_size;
	uint8_t* out;
	size_t out_cap, out_len;
	/* optional event trace (orc_trace_events): one record per coded bit */
	const uint16_t* tr_base;
	uint32_t* tr_ctx; uint8_t* tr_bit; uint16_t* tr_prob; uint32_t* tr_pos;
	size_t tr_cap, tr_len;
	uint32_t tr_cur_pos;
} orc_sink;

static void rc_put(orc_sink* s, uint8_t b)
{
	if (s->out_len < s->out_cap) s->out[s->out_len] = b;
	s->out_len++;
}
/* ref: range_encoder.c:18-38 */
static void rc_shift_low(orc_sink* s)
{
	uint32_t hi = (uint32_t)(s->low >> 32);
	uint32_t lo = (uint32_t)s->low;
	if (lo < 0xFF000000u || hi != 0) {
		uint8_t t = s->cache;
		do {
			rc_put(s, (uint8_t)(t + (hi & 0xFF)));
			t = 0xFF;
		} while (--s->cache_size != 0);
		s->cache = (uint8_t)(s->low >> 24);
	}
	s->cache_size++;
	s->low = (uint32_t)(lo << 8); /* 32-bit shift: the top byte was just emitted/cached */
}

/* ref: probability_model.c:5-15 (update) + perplexity_encoder.c:6-10 / range_encoder.c:47-64 (sink) */
static inline void put_bit(orc_sink* s, uint16_t* prob, unsigned bit)
{
	unsigned v = *prob;
	if (s->tr_base) {
		if (s->tr_len < s->tr_cap) {
			s->tr_ctx[s->tr_len] = (uint32_t)(prob - s->tr_base);
			s->tr_bit[s->tr_len] = (uint8_t)bit;
			s->tr_prob[s->tr_len] = (uint16_t)v;
			s->tr_pos[s->tr_len] = s->tr_cur_pos;
		}
		s->tr_len++;
	}
	if (s->kind == 0) {
		s->perp += g_cost[bit ? 2048 - v : v];
	} else {
		uint32_t bound = (s->range >> 11) * v;
		if (bit) { s->low += bound; s->range -= bound; } else { s->range = bound; }
		while ((s->range & 0xFF000000u) == 0) { s->range <<= 8; rc_shift_low(s); }
	}
	if (s->readonly) return;
	if (bit) v -= v >> 5; else v += (2048 - v) >> 5;
	*prob = (uint16_t)v;
}
/* ref: perplexity_encoder.c:12-17 / range_encoder.c:66-81 */
static void put_direct(orc_sink* s, unsigned bits, unsigned nbits)
{
	if (s->kind == 0) { s->perp += (uint64_t)nbits << 11; return; }
	do {
		unsigned bit = (bits >> (nbits - 1)) & 1;
		s->range >>= 1;
		if (bit) s->low += s->range;
		if ((s->range & 0xFF000000u) == 0) { s->range <<= 8; rc_shift_low(s); }
	} while (--nbits);
}
/* ref: probability_model.c:22-44 */
static void put_tree(orc_sink* s, uint16_t* probs, unsigned value, unsigned nbits)
{
	unsigned m = 1;
	for (unsigned i = nbits; i-- > 0;) {
		unsigned b = (value >> i) & 1;
		put_bit(s, &probs[m], b);
		m = (m << 1) | b;
	}
}
static void put_tree_rev(orc_sink* s, uint16_t* probs, unsigned value, unsigned nbits)
{
	unsigned m = 1;
	for (unsigned i = 0; i < nbits; i++) {
		unsigned b = value & 1;
		value >>= 1;
		put_bit(s, &probs[m], b);
		m = (m << 1) | b;
	}
}

/* ---------------------------------------------------------------- state */
static void state_init(const orc_ctx* c, orc_state* st, uint16_t* storage)
{
	st->p = storage;
	for (uint32_t i = 0; i < c->L.total; i++) storage[i] = 1024; /* ref: lzma_state.c:6-14 */
	st->ctx_state = 0;
	memset(st->dists, 0, sizeof st->dists);
	st->pos = 0;
}
/* ref: lzma_state.c:29-57 */
static uint8_t next_ctx_state(uint8_t s, unsigned type)
{
	switch (type) {
	case ORC_LITERAL: return s < 4 ? 0 : (s < 10 ? s - 3 : s - 6);
	case ORC_MATCH: return s < 7 ? 7 : 10;
	case ORC_SHORT_REP: return s < 7 ? 9 : 11;
	default: return s < 7 ? 8 : 11; /* LONG_REP */
	}
}

/* ref: lzma_packet_encoder.c:42-63.  pos_state is the pb extension (reference: always 0). */
static void put_length(orc_sink* s, uint16_t* lm, unsigned len, unsigned pos_state)
{
	len -= 2;
	if (len < 8) {
		put_bit(s, &lm[LEN_CHOICE1], 0);
		put_tree(s, &lm[LEN_LOW + pos_state * 8], len, 3);
	} else if (len < 16) {
		put_bit(s, &lm[LEN_CHOICE1], 1);
		put_bit(s, &lm[LEN_CHOICE2], 0);
		put_tree(s, &lm[LEN_MID + pos_state * 8], len - 8, 3);
	} else {
		put_bit(s, &lm[LEN_CHOICE1], 1);
		put_bit(s, &lm[LEN_CHOICE2], 1);
		put_tree(s, &lm[LEN_HIGH], len - 16, 8);
	}
}
/* ref: lzma_packet_encoder.c:71-104 */
static void put_distance(orc_sink* s, uint16_t* dm, uint32_t dist, unsigned len)
{
	unsigned len_ctx = len - 2 < 3 ? len - 2 : 3;
	uint16_t* slot_probs = &dm[DIST_SLOT + len_ctx * 64];
	if (dist < 4) { put_tree(s, slot_probs, dist, 6); return; }
	unsigned nlow = (32 - (unsigned)__builtin_clz(dist)) - 2;
	uint32_t low = dist & ((1u << nlow) - 1);
	uint32_t high = dist >> nlow; /* 2 or 3 */
	unsigned slot = nlow * 2 + high;
	put_tree(s, slot_probs, slot, 6);
	if (slot < 14) {
		put_tree_rev(s, &dm[DIST_POS + (high << nlow) - slot], low, nlow);
		return;
	}
	put_direct(s, low >> 4, nlow - 4);
	put_tree_rev(s, &dm[DIST_ALIGN], low & 15, 4);
}

/* ref: lzma_packet_encoder.c:169-194 with its statics :13-40 (header), :106-136 (literal),
 * :138-146 (match: pushes the distance before coding length+distance), :148-152, :154-167. */
static void encode_packet(const orc_ctx* c, orc_state* st, orc_sink* s, orc_packet pk)
{
	const orc_layout* L = &c->L;
	uint16_t* cs = st->p + L->cs;
	unsigned state = st->ctx_state;
	unsigned pos_state = (unsigned)(st->pos & ((1u << c->pb) - 1));
	unsigned sp = (state << 4) + pos_state;
	switch (pk.type) {
	case ORC_LITERAL: {
		put_bit(s, &cs[CS_IS_MATCH + sp], 0);
		unsigned prev = st->pos ? c->data[st->pos - 1] : 0;
		unsigned lit_ctx = (((unsigned)st->pos & ((1u << c->lp) - 1)) << c->lc) + (prev >> (8 - c->lc));
		if (c->lc == 0) lit_ctx = (unsigned)st->pos & ((1u << c->lp) - 1);
		uint16_t* lp = st->p + L->lit + 0x300 * lit_ctx;
		unsigned byte = c->data[st->pos];
		unsigned symbol = 1;
		int matched = state >= 7;
		unsigned match_byte = matched ? c->data[st->pos - st->dists[0] - 1] : 0;
		for (int i = 7; i >= 0; i--) {
			unsigned bit = (byte >> i) & 1;
			unsigned ctx = symbol;
			if (matched) {
				unsigned mb = (match_byte >> i) & 1;
				ctx += (1 + mb) << 8;
				matched = (mb == bit);
			}
			put_bit(s, &lp[ctx], bit);
			symbol = (symbol << 1) | bit;
		}
		break;
	}
	case ORC_MATCH:
		put_bit(s, &cs[CS_IS_MATCH + sp], 1);
		put_bit(s, &cs[CS_IS_REP + state], 0);
		st->dists[3] = st->dists[2]; st->dists[2] = st->dists[1]; st->dists[1] = st->dists[0];
		st->dists[0] = pk.dist; /* ref: lzma_state.c:59-65 */
		put_length(s, st->p + L->len, pk.len, pos_state);
		put_distance(s, st->p + L->dist, pk.dist, pk.len);
		break;
	case ORC_SHORT_REP:
		put_bit(s, &cs[CS_IS_MATCH + sp], 1);
		put_bit(s, &cs[CS_IS_REP + state], 1);
		put_bit(s, &cs[CS_G0 + state], 0);
		put_bit(s, &cs[CS_REP0_LONG + sp], 0);
		break;
	default: { /* LONG_REP */
		unsigned idx = pk.dist;
		put_bit(s, &cs[CS_IS_MATCH + sp], 1);
		put_bit(s, &cs[CS_IS_REP + state], 1);
		if (idx == 0) {
			put_bit(s, &cs[CS_G0 + state], 0);
			put_bit(s, &cs[CS_REP0_LONG + sp], 1);
		} else {
			put_bit(s, &cs[CS_G0 + state], 1);
			put_bit(s, &cs[CS_G1 + state], idx != 1);
			if (idx != 1) put_bit(s, &cs[CS_G2 + state], idx != 2);
		}
		/* ref: lzma_state.c:67-81 move-to-front */
		uint32_t d = st->dists[idx];
		for (unsigned k = idx; k > 0; k--) st->dists[k] = st->dists[k - 1];
		st->dists[0] = d;
		put_length(s, st->p + L->rep_len, pk.len, pos_state);
		break;
	}
	}
	st->ctx_state = next_ctx_state((uint8_t)state, pk.type);
	st->pos += pk.len;
}

/* ---------------------------------------------------------------- context */
/* ref: substring_enumerator.c:26-47 -- positions bucketed by leading bigram, ascending. */
static void build_index(orc_ctx* c)
{
	c->bucket_off = (uint32_t*)calloc(65537, sizeof(uint32_t));
	c->bucket_pos = (uint32_t*)malloc(sizeof(uint32_t) * (c->n ? c->n : 1));
	for (size_t i = 1; i < c->n; i++) c->bucket_off[(((unsigned)c->data[i - 1] << 8) | c->data[i]) + 1]++;
	for (unsigned b = 0; b < 65536; b++) c->bucket_off[b + 1] += c->bucket_off[b];
	uint32_t* fill = (uint32_t*)calloc(65536, sizeof(uint32_t));
	for (size_t i = 1; i < c->n; i++) {
		unsigned b = ((unsigned)c->data[i - 1] << 8) | c->data[i];
		c->bucket_pos[c->bucket_off[b] + fill[b]++] = (uint32_t)(i - 1);
	}
	free(fill);
}

orc_ctx* orc_new(const uint8_t* data, size_t n, int lc, int lp, int pb, uint32_t dict_limit)
{
	orc_cost_table();
	orc_ctx* c = (orc_ctx*)calloc(1, sizeof *c);
	if (!c) return NULL;
	c->data = data; c->n = n; c->lc = lc; c->lp = lp; c->pb = pb;
	c->dict_limit = dict_limit ? dict_limit : 0xFFFFFFFFu;
	c->L.lit = 0;
	c->L.len = 0x300u << (lc + lp);
	c->L.rep_len = c->L.len + LEN_SIZE;
	c->L.dist = c->L.rep_len + LEN_SIZE;
	c->L.cs = c->L.dist + DIST_SIZE;
	c->L.total = c->L.cs + CS_SIZE;
	build_index(c);
	return c;
}
void orc_free(orc_ctx* c)
{
	if (!c) return;
	free(c->bucket_off); free(c->bucket_pos); free(c);
}

uint64_t orc_cost_slab(orc_ctx* c, const orc_packet* slab, uint64_t* cum, size_t* npackets,
                       uint16_t* probs_out, uint8_t* ctx_state_out, uint32_t* dists_out)
{
	uint16_t* probs = (uint16_t*)malloc(sizeof(uint16_t) * c->L.total);
	orc_state st;
	state_init(c, &st, probs);
	orc_sink s = { 0 };
	size_t k = 0;
	while (st.pos < c->n) {
		encode_packet(c, &st, &s, slab[st.pos]);
		if (cum) cum[k] = s.perp;
		k++;
	}
	if (npackets) *npackets = k;
	if (probs_out) memcpy(probs_out, probs, sizeof(uint16_t) * c->L.total);
	if (ctx_state_out) *ctx_state_out = st.ctx_state;
	if (dists_out) memcpy(dists_out, st.dists, sizeof st.dists);
	free(probs);
	return s.perp;
}

/* Every coded bit of a slab walk, in coding order: probability index (reference struct order),
 * bit, probability before the update, position of the packet that coded it; plus the walk
 * state before every packet (pk_state: ctx_state, dists[4] per walked packet).  Used to
 * check the device's per-context event chains and state records. */
size_t orc_trace_events(orc_ctx* c, const orc_packet* slab, uint32_t* ev_ctx, uint8_t* ev_bit, uint16_t* ev_prob,
                        uint32_t* ev_pos, size_t cap, uint32_t* pk_pos, uint32_t* pk_state, size_t pk_cap, size_t* npk)
{
	uint16_t* probs = (uint16_t*)malloc(sizeof(uint16_t) * c->L.total);
	orc_state st;
	state_init(c, &st, probs);
	orc_sink s = { 0 };
	s.tr_base = probs; s.tr_ctx = ev_ctx; s.tr_bit = ev_bit; s.tr_prob = ev_prob; s.tr_pos = ev_pos; s.tr_cap = cap;
	size_t k = 0;
	while (st.pos < c->n) {
		if (k < pk_cap) {
			pk_pos[k] = (uint32_t)st.pos;
			pk_state[5 * k] = st.ctx_state;
			for (int i = 0; i < 4; i++) pk_state[5 * k + 1 + i] = st.dists[i];
		}
		k++;
		s.tr_cur_pos = (uint32_t)st.pos;
		encode_packet(c, &st, &s, slab[st.pos]);
	}
	if (npk) *npk = k;
	free(probs);
	return s.tr_len;
}

/* ---------------------------------------------------------------- enumeration */
typedef void (*cand_cb)(void* ud, orc_packet pk, uint64_t seq);

/* ref: substring_enumerator.c:85-105 (+ the dictionary window its :97 todo asks for, and the
 * optional fan-out cap of mgl_sa_config.max_bucket_scan: of the hits inside the window only the
 * nearest M -- the last M of the bucket before pos -- are enumerated; the reference scans them all). */
static void for_each_substring(const orc_ctx* c, size_t pos, size_t max_len,
                               void (*cb)(void*, size_t, size_t), void* ud)
{
	if (pos == 0 || pos == c->n - 1) return;
	unsigned b = ((unsigned)c->data[pos] << 8) | c->data[pos + 1];
	uint32_t lo = c->bucket_off[b], hi = lo;
	const uint32_t end = c->bucket_off[b + 1];
	while (hi < end && c->bucket_pos[hi] < pos) hi++;
	while (lo < hi && pos - c->bucket_pos[lo] - 1 >= c->dict_limit) lo++;
	if (c->max_bucket_scan && hi - lo > c->max_bucket_scan) lo = hi - c->max_bucket_scan;
	for (uint32_t i = lo; i < hi; i++) {
		size_t q = c->bucket_pos[i];
		cb(ud, q, 2);
		for (size_t j = 2; j < max_len && j + pos < c->n; j++) {
			if (c->data[pos + j] != c->data[q + j]) break;
			cb(ud, q, j + 1);
		}
	}
}

typedef struct { uint32_t* offs; uint32_t* lens; size_t cap, count; } sub_collect;
static void sub_cb(void* ud, size_t off, size_t len)
{
	sub_collect* s = (sub_collect*)ud;
	if (s->count < s->cap) { s->offs[s->count] = (uint32_t)off; s->lens[s->count] = (uint32_t)len; }
	s->count++;
}
size_t orc_substrings(orc_ctx* c, size_t pos, size_t max_len, uint32_t* offs, uint32_t* lens, size_t cap)
{
	sub_collect s = { offs, lens, cap, 0 };
	for_each_substring(c, pos, max_len, sub_cb, &s);
	return s.count;
}

/* ---------------------------------------------------------------- top-K */
#define ORC_MAX_K 64
typedef struct { orc_packet pk; uint32_t cost; uint64_t seq; } topk_entry;
typedef struct {
	const orc_ctx* c;
	const orc_state* st;
	orc_packet incumbent;
	int mode;
	size_t k;
	size_t count;
	topk_entry e[ORC_MAX_K];
	unsigned heap[ORC_MAX_K]; /* ref mode: index heap, max_heap.c */
	uint16_t* scratch;        /* copy of the model for costing one candidate */
} topk;

static int pk_eq(orc_packet a, orc_packet b) { return a.type == b.type && a.len == b.len && a.dist == b.dist; }

/* canonical order: a is better than b */
static int canon_better(const topk_entry* a, const topk_entry* b)
{
	return a->cost < b->cost || (a->cost == b->cost && a->seq > b->seq);
}

/* ref: max_heap.c:82-121 (sift-down / sift-up on an index heap; comparator = cost sign,
 * top_k_packet_finder.c:21-36) */
static void heap_down(topk* t, size_t parent)
{
	for (;;) {
		size_t l = 2 * parent + 1, r = l + 1;
		if (l >= t->count) break;
		size_t big = l;
		if (r < t->count && t->e[t->heap[r]].cost > t->e[t->heap[l]].cost) big = r;
		if (t->e[t->heap[big]].cost > t->e[t->heap[parent]].cost) {
			unsigned tmp = t->heap[big]; t->heap[big] = t->heap[parent]; t->heap[parent] = tmp;
			parent = big;
		} else break;
	}
}
static void heap_up(topk* t, size_t node)
{
	while (node > 0) {
		size_t parent = (node - 1) / 2;
		if (t->e[t->heap[node]].cost > t->e[t->heap[parent]].cost) {
			unsigned tmp = t->heap[node]; t->heap[node] = t->heap[parent]; t->heap[parent] = tmp;
			node = parent;
		} else break;
	}
}

/* ref: top_k_packet_finder.c:95-118 (cost one candidate from a copy of the state) and
 * :72-93 (insert).  cost = perplexity / length, integer (the reference then stores it
 * in a float; values stay < 2^24 so the float holds it exactly). */
static void topk_offer(void* ud, orc_packet pk, uint64_t seq)
{
	topk* t = (topk*)ud;
	if (pk_eq(pk, t->incumbent)) return; /* :99-101 */
	const orc_ctx* c = t->c;
	/* the reference copies the whole 5 280-byte state per candidate (:103); no context occurs
	 * twice inside one packet, so costing against the shared model read-only is identical */
	orc_state tmp = *t->st;
	orc_sink s = { 0 };
	s.readonly = 1;
	encode_packet(c, &tmp, &s, pk);
	topk_entry ent = { pk, (uint32_t)(s.perp / (tmp.pos - t->st->pos)), seq };
	if (t->mode == ORC_TOPK_REF) {
		if (t->count < t->k) {
			t->e[t->count] = ent;
			t->heap[t->count] = (unsigned)t->count;
			t->count++;
			heap_up(t, t->count - 1);
		} else if (ent.cost <= t->e[t->heap[0]].cost) {
			t->e[t->heap[0]] = ent;
			heap_down(t, 0);
		}
	} else {
		/* keep e[0..count) sorted worst -> best */
		if (t->count == t->k) {
			if (!canon_better(&ent, &t->e[0])) return;
			memmove(&t->e[0], &t->e[1], sizeof(topk_entry) * (t->count - 1));
			t->count--;
		}
		size_t i = t->count;
		while (i > 0 && canon_better(&t->e[i - 1], &ent)) i--;
		/* entries [i, count) are better than ent: they stay to the right */
		memmove(&t->e[i + 1], &t->e[i], sizeof(topk_entry) * (t->count - i));
		t->e[i] = ent;
		t->count++;
	}
}

/* candidate sequence number: strictly increasing along the reference's enumeration order */
static uint64_t seq_of(size_t hit_pos_plus1, unsigned len, unsigned kind)
{
	return ((uint64_t)hit_pos_plus1 << 12) | ((uint64_t)len << 3) | kind;
}

typedef struct { const orc_ctx* c; const orc_state* st; cand_cb cb; void* ud; } enum_ctx;
/* ref: packet_enumerator.c:41-55 */
static void enum_sub_cb(void* ud, size_t off, size_t len)
{
	enum_ctx* e = (enum_ctx*)ud;
	uint32_t dist = (uint32_t)(e->st->pos - off - 1);
	orc_packet m = { ORC_MATCH, dist, (uint16_t)len };
	e->cb(e->ud, m, seq_of(off + 1, (unsigned)len, 0));
	for (unsigned i = 0; i < 4; i++) {
		if (dist == e->st->dists[i]) {
			orc_packet r = { ORC_LONG_REP, i, (uint16_t)len };
			e->cb(e->ud, r, seq_of(off + 1, (unsigned)len, 1 + i));
		}
	}
}
/* ref: packet_enumerator.c:57-74 */
static void enumerate(const orc_ctx* c, const orc_state* st, cand_cb cb, void* ud)
{
	orc_packet lit = { ORC_LITERAL, 0, 1 };
	cb(ud, lit, 0);
	if (st->pos > 0 && c->data[st->pos] == c->data[st->pos - st->dists[0] - 1]) {
		orc_packet sr = { ORC_SHORT_REP, 0, 1 };
		cb(ud, sr, 1);
	}
	enum_ctx e = { c, st, cb, ud };
	for_each_substring(c, st->pos, 273, enum_sub_cb, &e);
}

/* ref: top_k_packet_finder.c:120-125 */
static void topk_find(topk* t, const orc_ctx* c, const orc_state* st, const orc_packet* slab,
                      int mode, size_t k, uint16_t* scratch)
{
	t->c = c; t->st = st; t->incumbent = slab[st->pos]; t->mode = mode;
	t->k = k > ORC_MAX_K ? ORC_MAX_K : k; t->count = 0; t->scratch = scratch;
	enumerate(c, st, topk_offer, t);
}
/* ref: top_k_packet_finder.c:127-138 + max_heap.c:146-156 -- pop the worst */
static int topk_pop(topk* t, topk_entry* out)
{
	if (t->count == 0) return 0;
	if (t->mode == ORC_TOPK_REF) {
		*out = t->e[t->heap[0]];
		t->heap[0] = t->heap[--t->count];
		heap_down(t, 0);
	} else {
		*out = t->e[0];
		memmove(&t->e[0], &t->e[1], sizeof(topk_entry) * (t->count - 1));
		t->count--;
	}
	return 1;
}

size_t orc_top_k(orc_ctx* c, const orc_packet* slab, size_t position, int mode, size_t k,
                 orc_packet* out, uint64_t* costs)
{
	uint16_t* probs = (uint16_t*)malloc(sizeof(uint16_t) * c->L.total * 2);
	orc_state st;
	state_init(c, &st, probs);
	orc_sink s = { 0 };
	while (st.pos < position) encode_packet(c, &st, &s, slab[st.pos]);
	if (st.pos != position) { free(probs); return (size_t)-1; }
	topk t;
	topk_find(&t, c, &st, slab, mode, k, probs + c->L.total);
	size_t count = 0;
	topk_entry ent;
	while (topk_pop(&t, &ent)) { out[count] = ent.pk; costs[count] = ent.cost; count++; }
	free(probs);
	return count;
}

/* ---------------------------------------------------------------- neighbour */
typedef uint32_t (*rng_fn)(void* ud);

typedef struct {
	orc_diff* d;
	size_t count, cap;
} journal;

/* ref: packet_slab_undo_stack.c:62-83 -- here a growable array; LIFO restore */
static void journal_push(journal* jn, uint32_t pos, orc_packet old)
{
	if (jn->count == jn->cap) {
		jn->cap = jn->cap ? jn->cap * 2 : 32;
		jn->d = (orc_diff*)realloc(jn->d, sizeof(orc_diff) * jn->cap);
	}
	jn->d[jn->count].position = pos;
	jn->d[jn->count].old_packet = old;
	jn->count++;
}
/* ref: packet_slab_undo_stack.c:85-100 */
static void journal_undo(journal* jn, orc_packet* slab)
{
	while (jn->count > 0) {
		jn->count--;
		slab[jn->d[jn->count].position] = jn->d[jn->count].old_packet;
	}
}

typedef struct {
	orc_ctx* c;
	rng_fn rng;
	void* rng_ud;
	int topk_mode;
	uint16_t* scratch;
	size_t repair_picks; /* LONG_REPs of the repair that no rep distance fitted: each costs a top-K query */
} gen_env;

/* ref: packet_slab_neighbour.c:48-54 */
static size_t rand_max_of(gen_env* g, size_t count, size_t num)
{
	size_t r = g->rng(g->rng_ud) % count;
	while (--num != 0) {
		size_t x = g->rng(g->rng_ud) % count;
		if (x > r) r = x;
	}
	return r;
}
/* ref: packet_slab_neighbour.c:56-72 */
static int pick_from_top_k(gen_env* g, const orc_state* st, orc_packet* slab, int best)
{
	topk t;
	topk_find(&t, g->c, st, slab, g->topk_mode, 20, g->scratch);
	size_t count = t.count;
	if (count == 0) return 0;
	size_t choice = rand_max_of(g, count, 8);
	if (g->rng(g->rng_ud) % 8 == 0 || best) choice = count - 1;
	topk_entry ent;
	while (topk_pop(&t, &ent)) {
		slab[st->pos] = ent.pk;
		if (choice-- == 0) return 1;
	}
	return 1;
}
/* ref: packet_slab_neighbour.c:74-80 */
static int long_rep_ok(const orc_ctx* c, const orc_state* st, orc_packet pk)
{
	return memcmp(c->data + st->pos - st->dists[pk.dist] - 1, c->data + st->pos, pk.len) == 0;
}

/* ref: packet_slab_neighbour.c:119-152 */
static int mutate(gen_env* g, const orc_state* st, orc_packet* slab, journal* jn)
{
	const orc_ctx* c = g->c;
	size_t pos = st->pos;
	orc_packet* first = &slab[pos];
	if (pos + 1 < c->n && g->rng(g->rng_ud) % 2 == 0) {
		orc_packet* second = &slab[pos + 1];
		if ((first->type == ORC_LONG_REP || first->type == ORC_MATCH) && first->len > 2) {
			journal_push(jn, (uint32_t)pos, *first);
			journal_push(jn, (uint32_t)pos + 1, *second);
			*second = *first;
			second->len--;
			first->type = ORC_LITERAL; first->dist = 0; first->len = 1;
			return 1;
		} else if (first->type == ORC_LITERAL || first->type == ORC_SHORT_REP) {
			if (second->type == ORC_MATCH || second->type == ORC_LONG_REP) {
				size_t rep_start = pos - second->dist;
				if (second->type == ORC_LONG_REP) rep_start = pos - st->dists[second->dist];
				if (second->len < 273 && rep_start > 0 && c->data[pos] == c->data[rep_start - 1]) {
					journal_push(jn, (uint32_t)pos, *first);
					*first = *second;
					first->len++;
					return 1;
				}
			}
		}
	}
	journal_push(jn, (uint32_t)pos, slab[pos]);
	return pick_from_top_k(g, st, slab, 0);
}

/* ref: packet_slab_neighbour.c:82-117 */
static void repair(gen_env* g, orc_state* st, orc_sink* s, orc_packet* slab, journal* jn)
{
	const orc_ctx* c = g->c;
	size_t count = 0;
	while (st->pos < c->n) {
		count++;
		orc_packet* pk = &slab[st->pos];
		orc_packet old = *pk;
		if (pk->type == ORC_SHORT_REP || pk->type == ORC_LITERAL) {
			if (c->data[st->pos] == c->data[st->pos - st->dists[0] - 1]) {
				if (count < 4) { pk->type = ORC_SHORT_REP; pk->dist = 0; pk->len = 1; }
			} else {
				pk->type = ORC_LITERAL; pk->dist = 0; pk->len = 1;
			}
		}
		if (pk->type == ORC_LONG_REP) {
			unsigned idx = 0;
			while (!long_rep_ok(c, st, *pk) && idx < 4) { pk->dist = idx; idx++; }
			if (!long_rep_ok(c, st, *pk)) {
				int best = (g->rng(g->rng_ud) % 4 == 0);
				g->repair_picks++;
				pick_from_top_k(g, st, slab, best);
			}
		}
		if (!pk_eq(old, *pk)) journal_push(jn, (uint32_t)st->pos, old);
		encode_packet(c, st, s, *pk);
	}
}

/* ref: packet_slab_neighbour.c:154-173, with the target supplied by the caller as a
 * byte position on the walk (the reference derives it from a packet ordinal, :162-165). */
static int generate_at(gen_env* g, orc_packet* slab, size_t target_pos, int by_ordinal,
                       size_t ordinal, journal* jn, uint64_t* cost, uint16_t* probs)
{
	orc_ctx* c = g->c;
	orc_state st;
	state_init(c, &st, probs);
	orc_sink s = { 0 };
	size_t k = 0;
	/* ref: packet_slab_neighbour.c:22-32 */
	while (st.pos < c->n) {
		if (by_ordinal ? (k == ordinal) : (st.pos >= target_pos)) break;
		k++;
		encode_packet(c, &st, &s, slab[st.pos]);
	}
	if (!mutate(g, &st, slab, jn)) return 0;
	encode_packet(c, &st, &s, slab[st.pos]);
	repair(g, &st, &s, slab, jn);
	*cost = s.perp;
	return 1;
}

/* ref: packet_slab.c:48-57 */
static size_t slab_count(const orc_packet* slab, size_t n)
{
	size_t pos = 0, count = 0;
	while (pos < n) { count++; pos += slab[pos].len; }
	return count;
}

static uint32_t libc_rng(void* ud) { (void)ud; return (uint32_t)rand(); }
void orc_srand(unsigned seed) { srand(seed); }

/* ref: main.c:78-102.  i*i etc. in int like the reference (callers keep i < 46341). */
int orc_sa_iters(orc_ctx* c, orc_packet* slab, orc_packet* best, uint64_t* cur_io,
                 uint64_t* best_cost_io, unsigned step, int num_iters, int i_begin, int i_end,
                 uint64_t* trace, uint64_t* undo_total)
{
	uint16_t* probs = (uint16_t*)malloc(sizeof(uint16_t) * c->L.total * 2);
	gen_env g = { c, libc_rng, NULL, ORC_TOPK_REF, probs + c->L.total, 0 };
	journal jn = { 0 };
	uint64_t cur = *cur_io, best_cost = *best_cost_io, undos = 0;
	size_t t = 0;
	for (int i = i_begin; i < i_end; i++) {
		jn.count = 0;
		size_t pcount = slab_count(slab, c->n);
		size_t target = (size_t)rand() % pcount;
		uint64_t cost = 0;
		if (!generate_at(&g, slab, 0, 1, target, &jn, &cost, probs)) { i--; continue; }
		undos += jn.count;
		int transition = rand() % (i * i + 1 + (int)(step * num_iters / 2)) < sqrt(num_iters);
		int accepted = 0;
		if (cur == 0 || cost < cur || transition) {
			cur = cost;
			accepted = 1;
			if (best_cost == 0 || cur < best_cost) {
				best_cost = cur;
				memcpy(best, slab, sizeof(orc_packet) * c->n);
			}
		} else {
			journal_undo(&jn, slab);
		}
		if (trace) { trace[2 * t] = cost; trace[2 * t + 1] = (uint64_t)accepted; }
		t++;
	}
	*cur_io = cur; *best_cost_io = best_cost;
	if (undo_total) *undo_total = undos;
	free(jn.d); free(probs);
	return (int)t;
}

/* ---------------------------------------------------------------- batched semantics */
#define ORC_MAX_JOURNAL 64
#define ORC_MAX_EVENTS 4096
#define ORC_BULK_ROUNDS 8u
#define ORC_MAX_REPAIR_PICKS 8 /* top-K picks the repair of one neighbour may need (invalid LONG_REPs, packet_slab_neighbour.c:99-109) */
#define ORC_MAX_WALK 2048 /* neighbour packets the device's two-pointer walk visits before it gives a neighbour up */
static uint64_t mix64(uint64_t z)
{
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
/* 31-bit draw number n of neighbour j at global step `step` */
uint32_t orc_draw(uint64_t seed, uint64_t step, uint32_t j, uint32_t n)
{
	uint64_t key = mix64(seed ^ mix64(step * 0x100000001B3ull + j));
	return (uint32_t)(mix64(key + n) >> 33);
}
typedef struct { uint64_t seed, step; uint32_t j, n; } ctr_rng;
static uint32_t ctr_next(void* ud)
{
	ctr_rng* r = (ctr_rng*)ud;
	return orc_draw(r->seed, r->step, r->j, r->n++);
}

static int diff_cmp(const void* a, const void* b)
{
	uint32_t pa = ((const orc_diff*)a)->position, pb = ((const orc_diff*)b)->position;
	return pa < pb ? -1 : pa > pb;
}

/* the walk state without the probabilities: ref lzma_state.c:29-81 */
typedef struct { size_t pos; uint8_t ctx_state; uint32_t dists[4]; } wstate;
static void wstate_advance(wstate* w, orc_packet pk)
{
	if (pk.type == ORC_MATCH) {
		w->dists[3] = w->dists[2]; w->dists[2] = w->dists[1]; w->dists[1] = w->dists[0];
		w->dists[0] = pk.dist;
	} else if (pk.type == ORC_LONG_REP) {
		uint32_t d = w->dists[pk.dist];
		for (unsigned k = pk.dist; k > 0; k--) w->dists[k] = w->dists[k - 1];
		w->dists[0] = d;
	}
	w->ctx_state = next_ctx_state(w->ctx_state, pk.type);
	w->pos += pk.len;
}
static int wstate_same(const wstate* a, const wstate* b)
{
	return a->ctx_state == b->ctx_state && memcmp(a->dists, b->dists, sizeof a->dists) == 0;
}
/* the base slab's entry at p while `slab` holds the neighbour: the first value journaled for p */
static orc_packet base_at(const orc_packet* slab, const journal* jn, size_t p)
{
	for (size_t i = 0; i < jn->count; i++) if (jn->d[i].position == p) return jn->d[i].old_packet;
	return slab[p];
}
/* The neighbour's window [target, end): `end` is the first position at which the neighbour's walk
 * and the base's walk stand on the same byte with the same ctx_state and rep distances, at least
 * three repair packets after the mutated one (the first three may still turn literals into short
 * reps, packet_slab_neighbour.c:90-98) -- from there on both parses are coded identically.  n when
 * they never meet again.  This is where the device's two-pointer walk stops.
 *
 * Two refinements for the bulk step's selection.  soft_end: the first such meeting
 * point -- same byte, same ctx_state, three repair packets done -- behind the last SHORT_REP / LONG_REP packet
 * the walk meets before `end`: the rep distances still differ in [soft_end, end), but no packet there reads
 * them, so a second neighbour may start there provided it is self-contained.  dep: the neighbour is NOT
 * self-contained -- some SHORT_REP / LONG_REP packet of its parse reads a rep distance that was pushed before
 * the window.  (A taken set is checked after the fact: orc_sa_batched validates the combined parse.) */
typedef struct { size_t end, soft_end; int dep; size_t n_ins, n_rem, walked; } window_info;
/* events (coded bits with a probability context) of one packet: 1 + 8 for a literal, 4 for a short rep, header +
 * length (+ slot + low distance bits through a reverse tree or the align tree) otherwise; direct bits are not events */
static size_t packet_events(const wstate* w, orc_packet pk)
{
	(void)w;
	if (pk.type == ORC_LITERAL) return 9;
	if (pk.type == ORC_SHORT_REP) return 4;
	const unsigned l = pk.len - 2u;
	const size_t lenbits = l < 8 ? 4 : (l < 16 ? 5 : 10);
	if (pk.type == ORC_LONG_REP) return (pk.dist < 2 ? 4 : 5) + lenbits;
	size_t tail = 0;
	if (pk.dist >= 4) {
		const unsigned nlow = (32u - (unsigned)__builtin_clz(pk.dist)) - 2u, slot = nlow * 2u + (pk.dist >> nlow);
		tail = slot < 14 ? nlow : 4;
	}
	return 2 + lenbits + 6 + tail;
}
static window_info window_end(const orc_ctx* c, const orc_packet* slab, const journal* jn, size_t target)
{
	window_info wi = { c->n, (size_t)-1, 0, 0, 0, 0 };
	wstate bs = { 0, 0, { 0, 0, 0, 0 } };
	while (bs.pos < target) wstate_advance(&bs, base_at(slab, jn, bs.pos));
	wstate nb = bs;
	unsigned count = 0, taint = 0xF; /* bit k: rep distance k was pushed before the window */
	int first = 1;
	for (;;) {
		if (!first && nb.pos == bs.pos && count >= 3 && nb.ctx_state == bs.ctx_state) {
			if (wi.soft_end == (size_t)-1) wi.soft_end = nb.pos;
			if (wstate_same(&nb, &bs)) { wi.end = nb.pos; break; }
			if (nb.ctx_state < 7 && nb.pos < c->n) {
				/* plain literals up to the base's next non-literal packet are coded identically in both walks: the device
				 * skips them in one jump (they do not count as visited) */
				size_t sx = nb.pos;
				while (sx < c->n && base_at(slab, jn, sx).type == ORC_LITERAL) sx++;
				if (sx > nb.pos) {
					const size_t k = sx - nb.pos;
					for (size_t i = 0; i < (k < 3 ? k : 3); i++) nb.ctx_state = next_ctx_state(nb.ctx_state, ORC_LITERAL);
					bs.ctx_state = nb.ctx_state;
					nb.pos = bs.pos = sx;
					count = 8;
					continue;
				}
			}
		}
		if (nb.pos >= c->n && bs.pos >= c->n) { wi.end = c->n; break; }
		if (wi.walked > ORC_MAX_WALK) break; /* given up: the caller drops it */
		if (nb.pos <= bs.pos && nb.pos < c->n) {
			if (!first && count < 8) count++;
			first = 0;
			wi.walked++;
			const size_t p = nb.pos;
			const orc_packet pk = slab[p];
			/* a packet the move changed (a SHORT_REP the repair turned into a literal, say): the soft window reaches behind it */
			if (count > 0 && !pk_eq(pk, base_at(slab, jn, p))) wi.soft_end = (size_t)-1;
			/* a rep packet reads a rep distance: the soft window reaches at least to behind it -- unless it is the base's own
			 * packet at this position reading a slot that holds the same distance in both walks (then the move has no part
			 * in what it codes) */
			if (pk.type == ORC_SHORT_REP || pk.type == ORC_LONG_REP) {
				const unsigned slot = pk.type == ORC_SHORT_REP ? 0u : pk.dist;
				const int same_read = bs.pos == p && pk_eq(pk, base_at(slab, jn, p)) && nb.dists[slot] == bs.dists[slot];
				if (!same_read) wi.soft_end = (size_t)-1;
				wi.dep |= pk.type == ORC_SHORT_REP ? (taint & 1u) : ((taint >> pk.dist) & 1u);
			}
			if (pk.type == ORC_MATCH) taint = (taint << 1) & 0xFu;
			else if (pk.type == ORC_LONG_REP) {
				const unsigned k = pk.dist, bit = (taint >> k) & 1u;
				taint = (taint & ~((2u << k) - 1u)) | ((taint & ((1u << k) - 1u)) << 1) | bit;
			}
			/* the base packet at the same position: identical coding cancels (same packet, same ctx_state and -- for a
			 * literal after a match -- the same byte at rep distance 0); otherwise its events go and the neighbour's come */
			const int paired = bs.pos == p;
			orc_packet bpk = pk;
			int cancelled = 0;
			if (paired) {
				bpk = base_at(slab, jn, p);
				cancelled = pk_eq(pk, bpk) && nb.ctx_state == bs.ctx_state;
				if (cancelled && pk.type == ORC_LITERAL && nb.ctx_state >= 7) {
					const unsigned mn = nb.dists[0] < p ? c->data[p - nb.dists[0] - 1] : 0u;
					const unsigned mb = bs.dists[0] < p ? c->data[p - bs.dists[0] - 1] : 0u;
					cancelled = mn == mb;
				}
			}
			if (!cancelled) {
				wi.n_ins += packet_events(&nb, pk);
				if (paired) wi.n_rem += packet_events(&bs, bpk);
			}
			if (paired) wstate_advance(&bs, bpk);
			wstate_advance(&nb, pk);
		} else {
			const orc_packet bpk = base_at(slab, jn, bs.pos);
			wi.n_rem += packet_events(&bs, bpk); /* a base packet the neighbour has passed over */
			wstate_advance(&bs, bpk);
		}
	}
	if (wi.soft_end == (size_t)-1 || wi.soft_end > wi.end) wi.soft_end = wi.end;
	return wi;
}
/* status: 1 = ok, 0 = no candidate at the target (main.c:81-84 retries those), -1 = dropped because
 * its journal needs more than ORC_MAX_JOURNAL distinct positions (the device's journal capacity) */
int orc_neighbour_ex(orc_ctx* c, orc_packet* slab, uint64_t seed, uint64_t step, uint32_t j,
                     int keep, uint64_t* cost, orc_diff* diffs, size_t* ndiffs, size_t cap, uint32_t* window)
{
	ctr_rng r = { seed, step, j, 0 };
	uint16_t* probs = (uint16_t*)malloc(sizeof(uint16_t) * c->L.total * 2);
	gen_env g = { c, ctr_next, &r, ORC_TOPK_CANON, probs + c->L.total, 0 };
	journal jn = { 0 };
	/* on-walk flags of the base slab */
	uint8_t* on = (uint8_t*)calloc(c->n, 1);
	for (size_t p = 0; p < c->n; p += slab[p].len) on[p] = 1;
	/* target: up to 32 uniform position draws, first one on the walk wins; otherwise the
	 * next on-walk position at or after the last draw (wrapping to 0) */
	size_t target = 0;
	if (c->strata) {
		/* stratified (the device's default): the packet whose ordinal is drawn uniformly from the j-th of K equal slices of
		 * the P packets on the walk (packet_slab_neighbour.c:162-163 draws an ordinal from all of them) */
		const uint32_t K = c->strata;
		uint64_t P = 0;
		for (size_t p = 0; p < c->n; p++) P += on[p];
		const uint64_t lo = (uint64_t)j * P / K, hi = (uint64_t)(j + 1u) * P / K;
		const uint32_t u = ctr_next(&r);
		uint64_t ord = lo + (hi > lo ? u % (hi - lo) : 0u);
		if (ord >= P) ord = P ? P - 1u : 0u;
		uint64_t k = 0;
		for (size_t p = 0; p < c->n; p++) if (on[p] && k++ == ord) { target = p; break; }
	} else {
		int found = 0;
		for (int t = 0; t < 32 && !found; t++) {
			target = ctr_next(&r) % c->n;
			found = on[target];
		}
		if (!found) {
			while (target < c->n && !on[target]) target++;
			if (target >= c->n) target = 0;
		}
	}
	free(on);
	uint64_t total = 0;
	int ok = generate_at(&g, slab, target, 0, 0, &jn, &total, probs);
	/* the device journal holds ORC_MAX_JOURNAL distinct positions; a neighbour that needs
	 * more is dropped as a failed generate */
	if (ok) {
		size_t distinct = 0;
		for (size_t i = 0; i < jn.count; i++) {
			int seen = 0;
			for (size_t q = 0; q < i && !seen; q++) seen = jn.d[q].position == jn.d[i].position;
			distinct += !seen;
		}
		if (distinct > ORC_MAX_JOURNAL) ok = -1;
		if (g.repair_picks > ORC_MAX_REPAIR_PICKS) ok = -1; /* the device gives such a neighbour up */
	}
	uint32_t wv[4] = { (uint32_t)target, 0xFFFFFFFFu, 0xFFFFFFFFu, 0 };
	if (ok == 1) {
		const window_info wi = window_end(c, slab, &jn, target);
		wv[1] = (uint32_t)wi.end; wv[2] = (uint32_t)wi.soft_end; wv[3] = (uint32_t)wi.dep;
		/* the device keeps at most ORC_MAX_EVENTS inserted and as many removed events per neighbour (its second
		 * pass's lists) and visits at most ORC_MAX_WALK of its packets: a neighbour that changes more of the coding than
		 * that, or whose walk stays apart from the base's for longer, is dropped, like one with too long a journal */
		if (wi.n_ins > ORC_MAX_EVENTS || wi.n_rem > ORC_MAX_EVENTS || wi.walked > ORC_MAX_WALK) { ok = -1; wv[1] = wv[2] = 0xFFFFFFFFu; wv[3] = 0; }
	}
	if (cost) *cost = ok == 1 ? total : ~0ull;
	if (window) memcpy(window, wv, sizeof wv);
	/* compact the journal: first old value per position + final value, drop no-ops */
	size_t nd = 0;
	if (ok == 1 && ndiffs) {
		orc_diff* tmp = (orc_diff*)malloc(sizeof(orc_diff) * (jn.count ? jn.count : 1));
		size_t m = 0;
		for (size_t i = 0; i < jn.count; i++) {
			int seen = 0;
			for (size_t q = 0; q < m; q++) if (tmp[q].position == jn.d[i].position) { seen = 1; break; }
			if (seen) continue;
			tmp[m] = jn.d[i];
			tmp[m].new_packet = slab[jn.d[i].position];
			m++;
		}
		qsort(tmp, m, sizeof(orc_diff), diff_cmp);
		for (size_t i = 0; i < m; i++) {
			if (pk_eq(tmp[i].old_packet, tmp[i].new_packet)) continue;
			if (nd < cap && diffs) diffs[nd] = tmp[i];
			nd++;
		}
		free(tmp);
	}
	if (ndiffs) *ndiffs = nd;
	if (ok != 1 || !keep) journal_undo(&jn, slab);
	free(jn.d); free(probs);
	return ok;
}
int orc_neighbour(orc_ctx* c, orc_packet* slab, uint64_t seed, uint64_t step, uint32_t j,
                  int keep, uint64_t* cost, orc_diff* diffs, size_t* ndiffs, size_t cap)
{
	return orc_neighbour_ex(c, slab, seed, step, j, keep, cost, diffs, ndiffs, cap, NULL) == 1;
}

static uint64_t ceil_sqrt_u64(uint64_t x)
{
	uint64_t r = (uint64_t)sqrt((double)x);
	while (r * r > x) r--;
	while ((r + 1) * (r + 1) <= x) r++;
	return r * r == x ? r : r + 1;
}

/* every packet on the walk reproduces the input (the device's k_validate) */
static int parse_is_valid(const orc_ctx* c, const orc_packet* slab)
{
	wstate w = { 0, 0, { 0, 0, 0, 0 } };
	while (w.pos < c->n) {
		const orc_packet pk = slab[w.pos];
		if (pk.type < ORC_LITERAL || pk.type > ORC_LONG_REP || pk.len == 0 || w.pos + pk.len > c->n) return 0;
		if (pk.type == ORC_LITERAL) { if (pk.len != 1) return 0; }
		else {
			uint32_t src;
			if (pk.type == ORC_SHORT_REP) { if (pk.len != 1) return 0; src = w.dists[0]; }
			else {
				if (pk.len < 2 || pk.len > 273 || (pk.type == ORC_LONG_REP && pk.dist > 3)) return 0;
				src = pk.type == ORC_MATCH ? pk.dist : w.dists[pk.dist];
			}
			if (src >= w.pos || src >= c->dict_limit) return 0;
			for (unsigned i = 0; i < pk.len; i++) if (c->data[w.pos - src - 1 + i] != c->data[w.pos + i]) return 0;
		}
		wstate_advance(&w, pk);
	}
	return 1;
}

/* Two neighbours (window = target, end, soft_end, dep) cannot both be taken: with A the one that starts first,
 * unless B starts at or after A's soft end, and either B is self-contained or B starts at or after A's end. */
/* bulk steps of orc_sa_batched whose combined parse failed the check and was taken back (tests: the window rule should never need this net) */
static uint64_t g_bulk_rollbacks;
uint64_t orc_bulk_rollbacks(void) { return g_bulk_rollbacks; }
/* slab entries two taken journals of one step both wrote (must stay 0: the selection's windows cover every changed packet) */
static uint64_t g_bulk_overlaps;
uint64_t orc_bulk_overlaps(void) { return g_bulk_overlaps; }

static int windows_conflict(const uint32_t* x, const uint32_t* y)
{
	const uint32_t* a = x[0] <= y[0] ? x : y;
	const uint32_t* b = x[0] <= y[0] ? y : x;
	if (a[0] == b[0]) return 1;
	return !(a[2] <= b[0] && (b[3] == 0 || a[1] <= b[0]));
}

/* one step = K neighbours of the same base slab, then one decision.
 *
 * Every evaluation keeps the reference's own rule (main.c:86-87): neighbour j of a step is the
 * epoch's iteration i = iter0 + (step - step_begin) * K + j and is *acceptable* when it costs less
 * than the current slab, or when its transition draw  draw % (i*i + 1 + phase*N/2) < sqrt(N)  says so
 * (with a temperature t > 0, opt-in and not in the reference: instead when u < exp(-delta / t_eff),
 * in integers through the reference's own log table, delta * 2048 <= t_eff * T[u], t_eff = t cooled
 * linearly over the epoch).  The current cost of an epoch that has not accepted anything yet is the
 * exact cost of its starting slab (the reference leaves it at 0 and so accepts its first neighbour
 * whatever it costs, main.c:74,87).
 * Acceptable neighbours are ranked by key = (improving ? 0 : 1, cost, j).  What a step takes:
 *   mode 0 (single): the acceptable neighbour with the smallest key;
 *   mode 1 (bulk):   every acceptable neighbour whose window [target, end) overlaps no acceptable
 *                    neighbour of smaller key -- their journals touch disjoint parts of the slab and
 *                    the walk state between them is the base's, so the result is a valid parse; its
 *                    exact cost is that of a fresh walk.
 * trace (nullable) gets 4 u64 per step: smallest acceptable cost (or ~0), neighbours accepted,
 * acceptable neighbours, current cost after the step. */
int orc_sa_batched(orc_ctx* c, orc_packet* slab, orc_packet* best, uint64_t* cur_io,
                   uint64_t* best_cost_io, uint64_t seed, uint32_t K, unsigned phase,
                   uint64_t iters_per_epoch, uint64_t iter0, uint64_t step_begin, uint64_t step_end,
                   const uint8_t* modes, uint64_t* trace, uint64_t* valid_evals, uint64_t* dropped_out)
{
	uint64_t cur = *cur_io, best_cost = *best_cost_io, valid = 0, dropped = 0;
	uint64_t* costs = (uint64_t*)malloc(sizeof(uint64_t) * K);
	uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * K);
	uint32_t* win = (uint32_t*)malloc(sizeof(uint32_t) * 4 * K);
	orc_diff* diffs = (orc_diff*)malloc(sizeof(orc_diff) * ORC_MAX_JOURNAL * (size_t)K);
	size_t* nd = (size_t*)malloc(sizeof(size_t) * K);
	uint8_t* take = (uint8_t*)malloc(K);
	const uint64_t thresh = ceil_sqrt_u64(iters_per_epoch);
	if (c->strata) c->strata = K; /* stratified targets: the slices are those of this run's K */
	for (uint64_t s = step_begin; s < step_end; s++) {
		const int bulk = modes ? modes[s - step_begin] : 0;
		if (cur == 0) cur = orc_cost_slab(c, slab, NULL, NULL, NULL, NULL, NULL);
		uint64_t minkey = ~0ull;
		uint32_t minj = ~0u;
		size_t nacceptable = 0;
		for (uint32_t j = 0; j < K; j++) {
			const int st = orc_neighbour_ex(c, slab, seed, s, j, 0, &costs[j], diffs + (size_t)j * ORC_MAX_JOURNAL, &nd[j],
			                                ORC_MAX_JOURNAL, win + 4 * (size_t)j);
			keys[j] = ~0ull;
			if (st == -1) dropped++;
			if (st != 1) continue;
			valid++;
			uint64_t i = iter0 + (s - step_begin) * K + j;
			if (i > 0x7FFFFFFFull) i = 0x7FFFFFFFull;
			const uint32_t draw = orc_draw(seed, s, 0xFFFFFFFFu, 2u + j);
			int ok = costs[j] < cur;
			if (!ok) {
				if (c->temperature) {
					const uint32_t u = draw % 2047u + 1u;
					const uint64_t ic = i < iters_per_epoch ? i : iters_per_epoch;
					const uint64_t t_eff = c->temperature * (iters_per_epoch - ic) / iters_per_epoch;
					ok = (costs[j] - cur) * 2048u <= t_eff * (uint64_t)orc_cost_table()[u];
				} else {
					const uint64_t m = i * i + 1 + (uint64_t)phase * iters_per_epoch / 2;
					ok = ((uint64_t)draw % m) < thresh;
				}
			}
			if (!ok) continue;
			nacceptable++;
			keys[j] = ((costs[j] < cur ? 0ull : 1ull) << 63) | (costs[j] << 20) | j;
			if (keys[j] < minkey) { minkey = keys[j]; minj = j; }
		}
		size_t ntaken = 0;
		memset(take, 0, K);
		if (!bulk) {
			if (minj != ~0u) { take[minj] = 1; ntaken = 1; }
		} else {
			/* the greedy independent set in key order, in ORC_BULK_ROUNDS synchronous rounds (states read from the
			 * previous round, written for the next; still undecided at the end = rejected) -- the device's
			 * k_bulk_round, round for round */
			uint8_t* st = (uint8_t*)calloc(2 * (size_t)K, 1);
			uint32_t* acc = (uint32_t*)malloc(sizeof(uint32_t) * K);
			uint32_t nacc = 0;
			for (uint32_t j = 0; j < K; j++) if (keys[j] != ~0ull) acc[nacc++] = j;
			for (uint32_t r = 0; r < ORC_BULK_ROUNDS; r++) {
				const uint8_t* in = st + (size_t)(r & 1u) * K;
				uint8_t* outp = st + (size_t)((r + 1u) & 1u) * K;
				for (uint32_t aj = 0; aj < nacc; aj++) {
					const uint32_t j = acc[aj];
					outp[j] = in[j];
					if (in[j]) continue;
					int lose = 0, blocked = 0;
					for (uint32_t ai = 0; ai < nacc && !lose; ai++) {
						const uint32_t i = acc[ai];
						if (in[i] == 2 || !(keys[i] < keys[j]) || !windows_conflict(win + 4 * (size_t)i, win + 4 * (size_t)j)) continue;
						if (in[i] == 1) lose = 1; else blocked = 1;
					}
					outp[j] = lose ? 2 : (blocked ? 0 : 1);
				}
			}
			free(acc);
			const uint8_t* fin = st + (size_t)(ORC_BULK_ROUNDS & 1u) * K;
			for (uint32_t j = 0; j < K; j++)
				if (keys[j] != ~0ull && fin[j] == 1) { take[j] = 1; ntaken++; }
			free(st);
		}
		for (uint32_t j = 0; j < K; j++) {
			if (!take[j]) continue;
			for (size_t e = 0; e < nd[j]; e++) {
				const orc_diff* d = &diffs[(size_t)j * ORC_MAX_JOURNAL + e];
				/* taken journals must touch disjoint entries (the device writes them in parallel): an entry that no longer
				 * holds what this journal replaced was written by another one */
				if (!pk_eq(slab[d->position], d->old_packet)) g_bulk_overlaps++;
				slab[d->position] = d->new_packet;
			}
		}
		if (ntaken && bulk && !parse_is_valid(c, slab)) {
			/* the safety net of the soft window ends: a combination that is not a valid parse is taken back as a whole
			 * (never seen to happen; the device checks the same thing with k_validate after its rebuild) */
			for (uint32_t j = 0; j < K; j++) {
				if (!take[j]) continue;
				for (size_t e = 0; e < nd[j]; e++) slab[diffs[(size_t)j * ORC_MAX_JOURNAL + e].position] = diffs[(size_t)j * ORC_MAX_JOURNAL + e].old_packet;
			}
			ntaken = 0;
			g_bulk_rollbacks++;
		}
		if (ntaken) {
			cur = (!bulk) ? costs[minj] : orc_cost_slab(c, slab, NULL, NULL, NULL, NULL, NULL);
			if (best_cost == 0 || cur < best_cost) {
				best_cost = cur;
				memcpy(best, slab, sizeof(orc_packet) * c->n);
			}
		}
		if (trace) {
			trace[4 * (s - step_begin) + 0] = minj == ~0u ? ~0ull : costs[minj];
			trace[4 * (s - step_begin) + 1] = ntaken;
			trace[4 * (s - step_begin) + 2] = nacceptable;
			trace[4 * (s - step_begin) + 3] = cur;
		}
	}
	*cur_io = cur; *best_cost_io = best_cost;
	if (valid_evals) *valid_evals = valid;
	if (dropped_out) *dropped_out = dropped;
	free(costs); free(keys); free(win); free(diffs); free(nd); free(take);
	return 0;
}

/* ---------------------------------------------------------------- emission */
/* ref: lzma_header_encoder.c:5-21, range_encoder.c:83-101, main.c:110-119 */
size_t orc_emit(orc_ctx* c, const orc_packet* slab, uint8_t* out, size_t cap)
{
	orc_sink s = { 0 };
	s.kind = 1; s.out = out; s.out_cap = cap;
	rc_put(&s, (uint8_t)((c->pb * 5 + c->lp) * 9 + c->lc));
	uint32_t dict = 0x400000;
	for (int i = 0; i < 4; i++) rc_put(&s, (uint8_t)(dict >> (8 * i)));
	uint64_t sz = (uint32_t)c->n; /* the reference truncates through htole32, :19 */
	for (int i = 0; i < 8; i++) rc_put(&s, (uint8_t)(sz >> (8 * i)));
	s.low = 0; s.range = 0xFFFFFFFFu; s.cache = 0; s.cache_size = 1;
	uint16_t* probs = (uint16_t*)malloc(sizeof(uint16_t) * c->L.total);
	orc_state st;
	state_init(c, &st, probs);
	while (st.pos < c->n) encode_packet(c, &st, &s, slab[st.pos]);
	for (int i = 0; i < 5; i++) rc_shift_low(&s);
	free(probs);
	return s.out_len;
}
