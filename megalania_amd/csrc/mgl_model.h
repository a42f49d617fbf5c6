/*
 * mgl_model.h -- the LZMA bit model of the hot path as a table of *event slots*.
 *
 * The reference codes a packet as a serial sequence of (probability context, bit) pairs
 * (lzma_packet_encoder.c:169-194 driving probability_model.c:5-44).  No context occurs
 * twice inside one packet, and which contexts/bits a packet uses depends only on the packet,
 * the input bytes and the small walk state (ctx_state, rep distances, position) -- never on
 * the adaptive probabilities.  So a packet is planned once (mgl_plan_packet) and every one
 * of its <= 26 events is then available in closed form by slot number (mgl_plan_event):
 * on the GPU lane e of a wavefront evaluates slot e, so a whole packet costs one LDS
 * round trip instead of ~9-23 dependent ones; on the host the emitter walks the slots in
 * order (slot order == coding order) and feeds EncoderInterface.
 *
 * Plain C99/C++ subset, no dependencies; MGL_HD marks host+device functions under hipcc.
 */
#ifndef MGL_MODEL_H
#define MGL_MODEL_H

#include <stdint.h>

#ifdef __HIPCC__
#define MGL_HD __host__ __device__ static inline
#else
#define MGL_HD static inline
#endif

/* lzma_packet.h:5-9 */
#define MGL_LITERAL 1u
#define MGL_MATCH 2u
#define MGL_SHORT_REP 3u
#define MGL_LONG_REP 4u

#define MGL_MIN_MATCH 2u
#define MGL_MAX_MATCH 273u /* packet_enumerator.c:6-7 */
#define MGL_PROB_INIT 1024u /* probability.h:7 */
#define MGL_MAX_EVENTS 26u

/* Position-indexed slab entry packed in 8 bytes: dist[31:0] len[47:32] type[55:48]. */
typedef uint64_t mgl_pk;
MGL_HD mgl_pk mgl_pack(uint32_t type, uint32_t dist, uint32_t len)
{
	return (uint64_t)dist | ((uint64_t)len << 32) | ((uint64_t)type << 48);
}
MGL_HD uint32_t mgl_pk_type(mgl_pk p) { return (uint32_t)(p >> 48) & 0xFFu; }
MGL_HD uint32_t mgl_pk_len(mgl_pk p) { return (uint32_t)(p >> 32) & 0xFFFFu; }
MGL_HD uint32_t mgl_pk_dist(mgl_pk p) { return (uint32_t)p; }
#define MGL_PK_LITERAL ((uint64_t)1 << 32 | (uint64_t)MGL_LITERAL << 48)
#define MGL_PK_SHORT_REP ((uint64_t)1 << 32 | (uint64_t)MGL_SHORT_REP << 48)

/* Probability array layout (u16 units).  Same contexts as lzma_state.h:15-58; the fixed-size
 * groups come first and the literal coder last so that lc/lp only change the tail. */
#define MGL_CS_IS_MATCH 0u    /* [12][16] */
#define MGL_CS_IS_REP 192u    /* [12] */
#define MGL_CS_G0 204u
#define MGL_CS_G1 216u
#define MGL_CS_G2 228u
#define MGL_CS_REP0_LONG 240u /* [12][16] */
#define MGL_OFF_LEN 432u      /* choice1, choice2, low[16][8], mid[16][8], high[256] */
#define MGL_OFF_REP_LEN 946u
#define MGL_LEN_LOW 2u
#define MGL_LEN_MID 130u
#define MGL_LEN_HIGH 258u
#define MGL_OFF_DIST 1460u    /* pos_slot[4][64], align[16], pos_coder[115] */
#define MGL_DIST_ALIGN 256u
#define MGL_DIST_POS 272u
#define MGL_OFF_LIT 1847u     /* 0x300 << (lc+lp) */

typedef struct {
	uint32_t lc, lp, pb;
	uint32_t total; /* number of probabilities */
} mgl_layout;

MGL_HD mgl_layout mgl_make_layout(uint32_t lc, uint32_t lp, uint32_t pb)
{
	mgl_layout L;
	L.lc = lc; L.lp = lp; L.pb = pb;
	L.total = MGL_OFF_LIT + (0x300u << (lc + lp));
	return L;
}

/* walk state, lzma_state.h:60-74 minus the probabilities */
typedef struct {
	uint32_t pos;
	uint32_t ctx_state;
	uint32_t dists[4];
} mgl_wstate;

/* rep distance by (runtime) index, without indexing the array dynamically: on the GPU that
 * would push the whole walk state into scratch memory */
MGL_HD uint32_t mgl_dist_at(const mgl_wstate* st, uint32_t i)
{
	return i == 0 ? st->dists[0] : i == 1 ? st->dists[1] : i == 2 ? st->dists[2] : st->dists[3];
}

/* lzma_state.c:29-57 */
MGL_HD uint32_t mgl_next_ctx_state(uint32_t s, uint32_t type)
{
	if (type == MGL_LITERAL) return s < 4 ? 0 : (s < 10 ? s - 3 : s - 6);
	if (type == MGL_MATCH) return s < 7 ? 7 : 10;
	if (type == MGL_SHORT_REP) return s < 7 ? 9 : 11;
	return s < 7 ? 8 : 11;
}

/* lzma_state.c:59-81 + the position advance of lzma_packet_encoder.c:193 */
MGL_HD void mgl_advance(mgl_wstate* st, uint32_t type, uint32_t dist, uint32_t len)
{
	if (type == MGL_MATCH) {
		st->dists[3] = st->dists[2]; st->dists[2] = st->dists[1]; st->dists[1] = st->dists[0];
		st->dists[0] = dist;
	} else if (type == MGL_LONG_REP) {
		uint32_t d = mgl_dist_at(st, dist);
		if (dist > 2) st->dists[3] = st->dists[2];
		if (dist > 1) st->dists[2] = st->dists[1];
		if (dist > 0) st->dists[1] = st->dists[0];
		st->dists[0] = d;
	}
	st->ctx_state = mgl_next_ctx_state(st->ctx_state, type);
	st->pos += len;
}

/* Everything about one packet that its event slots need (all wave-uniform). */
typedef struct {
	uint32_t type;
	uint32_t nev;      /* events in slots [0, nev) */
	uint32_t ndirect;  /* direct bits, coded between the slot tree and the align tree */
	uint32_t direct_after; /* slot index after which the direct bits go (host emitter) */
	uint32_t direct_val;
	/* header */
	uint32_t sp;       /* (ctx_state<<4) + pos_state */
	uint32_t state;
	uint32_t nhdr;
	uint32_t hdr_ctx0, hdr_ctx1, hdr_ctx2, hdr_ctx3, hdr_ctx4; /* named, not an array: lanes index them */
	uint32_t hdr_bits; /* bit i of event i */
	/* literal */
	uint32_t lit_base, byte, match_byte, matched;
	/* length (MATCH / LONG_REP) */
	uint32_t len_base;  /* MGL_OFF_LEN or MGL_OFF_REP_LEN */
	uint32_t len_nchoice; /* 1 or 2 choice bits */
	uint32_t len_choice_bits;
	uint32_t len_tree;  /* context base of the bit tree */
	uint32_t len_tbits; /* 3 or 8 */
	uint32_t len_val;
	/* distance (MATCH) */
	uint32_t slot_tree; /* context base of pos_slot[len_ctx] */
	uint32_t slot;
	uint32_t tail_tree; /* reverse-tree context base (pos_coder + off, or align) */
	uint32_t tail_bits; /* 0..5 */
	uint32_t tail_val;
} mgl_plan;

/* lzma_packet_encoder.c:42-63 */
MGL_HD void mgl_plan_length(mgl_plan* p, uint32_t base, uint32_t len, uint32_t pos_state)
{
	uint32_t l = len - MGL_MIN_MATCH;
	p->len_base = base;
	if (l < 8) {
		p->len_nchoice = 1; p->len_choice_bits = 0;
		p->len_tree = base + MGL_LEN_LOW + pos_state * 8; p->len_tbits = 3; p->len_val = l;
	} else if (l < 16) {
		p->len_nchoice = 2; p->len_choice_bits = 1; /* choice1=1, choice2=0 */
		p->len_tree = base + MGL_LEN_MID + pos_state * 8; p->len_tbits = 3; p->len_val = l - 8;
	} else {
		p->len_nchoice = 2; p->len_choice_bits = 3;
		p->len_tree = base + MGL_LEN_HIGH; p->len_tbits = 8; p->len_val = l - 16;
	}
}

MGL_HD uint32_t mgl_msb32(uint32_t v)
{
#ifdef __HIP_DEVICE_COMPILE__
	return 32u - (uint32_t)__clz((int)v);
#else
	return 32u - (uint32_t)__builtin_clz(v);
#endif
}

/* Plan a packet at walk state `st`.  `byte` = data[pos]; `match_byte` = data[pos-dists[0]-1]
 * (only read when the packet is a LITERAL and ctx_state >= 7, lzma_packet_encoder.c:117-122);
 * `prev_byte` = data[pos-1] (only for lc > 0; the reference has lc = 0). */
MGL_HD void mgl_plan_packet(const mgl_layout* L, const mgl_wstate* st, uint32_t type, uint32_t dist,
                            uint32_t len, uint32_t byte, uint32_t match_byte, uint32_t prev_byte, mgl_plan* p)
{
	uint32_t state = st->ctx_state;
	uint32_t pos_state = st->pos & ((1u << L->pb) - 1u);
	p->type = type;
	p->state = state;
	p->sp = (state << 4) + pos_state;
	p->ndirect = 0; p->direct_after = 0; p->direct_val = 0;
	p->tail_bits = 0; p->len_tbits = 0; p->len_nchoice = 0;
	p->hdr_ctx1 = p->hdr_ctx2 = p->hdr_ctx3 = p->hdr_ctx4 = 0;
	p->lit_base = 0; p->byte = 0; p->match_byte = 0; p->matched = 0;
	p->len_base = 0; p->len_choice_bits = 0; p->len_tree = 0; p->len_val = 0;
	p->slot_tree = 0; p->slot = 0; p->tail_tree = 0; p->tail_val = 0;
	p->hdr_ctx0 = MGL_CS_IS_MATCH + p->sp;
	if (type == MGL_LITERAL) {
		/* lzma_packet_encoder.c:106-136 */
		uint32_t lit_ctx = ((st->pos & ((1u << L->lp) - 1u)) << L->lc) + ((prev_byte & 0xFFu) >> (8u - L->lc));
		p->nhdr = 1; p->hdr_bits = 0;
		p->lit_base = MGL_OFF_LIT + 0x300u * lit_ctx;
		p->byte = byte & 0xFFu;
		p->matched = state >= 7;
		p->match_byte = match_byte & 0xFFu;
		p->nev = 9;
		return;
	}
	p->hdr_ctx1 = MGL_CS_IS_REP + state;
	if (type == MGL_MATCH) {
		/* lzma_packet_encoder.c:138-146, :71-104 */
		p->nhdr = 2; p->hdr_bits = 1; /* is_match=1, is_rep=0 */
		mgl_plan_length(p, MGL_OFF_LEN, len, pos_state);
		uint32_t len_ctx = len - 2 < 3 ? len - 2 : 3;
		p->slot_tree = MGL_OFF_DIST + len_ctx * 64;
		uint32_t nslot_ev = 6;
		if (dist < 4) {
			p->slot = dist;
		} else {
			uint32_t nlow = mgl_msb32(dist) - 2;
			uint32_t low = dist & ((1u << nlow) - 1u);
			uint32_t high = dist >> nlow;
			p->slot = nlow * 2 + high;
			if (p->slot < 14) {
				p->tail_tree = MGL_OFF_DIST + MGL_DIST_POS + (high << nlow) - p->slot;
				p->tail_bits = nlow; p->tail_val = low;
			} else {
				p->ndirect = nlow - 4; p->direct_val = low >> 4;
				p->tail_tree = MGL_OFF_DIST + MGL_DIST_ALIGN;
				p->tail_bits = 4; p->tail_val = low & 15u;
			}
		}
		p->nev = p->nhdr + p->len_nchoice + p->len_tbits + nslot_ev + p->tail_bits;
		p->direct_after = p->nhdr + p->len_nchoice + p->len_tbits + nslot_ev;
		return;
	}
	p->hdr_ctx2 = MGL_CS_G0 + state;
	if (type == MGL_SHORT_REP) {
		/* lzma_packet_encoder.c:148-152 */
		p->nhdr = 4; p->hdr_bits = 0x3; /* 1,1,0,0 */
		p->hdr_ctx3 = MGL_CS_REP0_LONG + p->sp;
		p->nev = 4;
		return;
	}
	/* LONG_REP, lzma_packet_encoder.c:154-167 + header :31-39 */
	{
		uint32_t idx = dist;
		if (idx == 0) {
			p->nhdr = 4; p->hdr_bits = 0x3 | (1u << 3); /* 1,1,0,1 */
			p->hdr_ctx3 = MGL_CS_REP0_LONG + p->sp;
		} else {
			p->hdr_ctx3 = MGL_CS_G1 + state;
			if (idx == 1) { p->nhdr = 4; p->hdr_bits = 0x7; /* 1,1,1,0 */ }
			else {
				p->nhdr = 5; p->hdr_ctx4 = MGL_CS_G2 + state;
				p->hdr_bits = 0xF | ((idx != 2 ? 1u : 0u) << 4); /* 1,1,1,1,(idx!=2) */
			}
		}
		mgl_plan_length(p, MGL_OFF_REP_LEN, len, pos_state);
		p->nev = p->nhdr + p->len_nchoice + p->len_tbits;
	}
}

MGL_HD uint32_t mgl_bitrev(uint32_t v, uint32_t nbits)
{
	uint32_t r = 0;
	for (uint32_t i = 0; i < nbits; i++) r |= ((v >> i) & 1u) << (nbits - 1 - i);
	return r;
}

/* Event number `slot` (< p->nev) of a planned packet: its probability context and bit.
 * Slot order is coding order. */
MGL_HD void mgl_plan_event(const mgl_plan* p, uint32_t slot, uint32_t* ctx, uint32_t* bit)
{
	if (slot < p->nhdr) {
		*ctx = slot == 0 ? p->hdr_ctx0 : slot == 1 ? p->hdr_ctx1 : slot == 2 ? p->hdr_ctx2
		     : slot == 3 ? p->hdr_ctx3 : p->hdr_ctx4;
		*bit = (p->hdr_bits >> slot) & 1u;
		return;
	}
	uint32_t e = slot - p->nhdr;
	if (p->type == MGL_LITERAL) {
		/* e = 0..7 codes bit i = 7-e; context = 1 followed by the higher bits already coded */
		uint32_t i = 7u - e;
		uint32_t b = (p->byte >> i) & 1u;
		uint32_t c = (1u << e) | (p->byte >> (i + 1u));
		if (p->matched && ((p->byte ^ p->match_byte) >> (i + 1u)) == 0u) {
			c += (1u + ((p->match_byte >> i) & 1u)) << 8;
		}
		*ctx = p->lit_base + c;
		*bit = b;
		return;
	}
	if (e < p->len_nchoice) {
		*ctx = p->len_base + e; /* choice_1, choice_2 */
		*bit = (p->len_choice_bits >> e) & 1u;
		return;
	}
	e -= p->len_nchoice;
	if (e < p->len_tbits) {
		uint32_t i = p->len_tbits - 1u - e;
		*ctx = p->len_tree + ((1u << e) | (p->len_val >> (i + 1u)));
		*bit = (p->len_val >> i) & 1u;
		return;
	}
	e -= p->len_tbits;
	if (e < 6u) {
		uint32_t i = 5u - e;
		*ctx = p->slot_tree + ((1u << e) | (p->slot >> (i + 1u)));
		*bit = (p->slot >> i) & 1u;
		return;
	}
	e -= 6u;
	/* reverse bit tree, probability_model.c:34-44: context = 1 followed by the low bits
	 * already coded, first-coded bit most significant */
	*ctx = p->tail_tree + ((1u << e) | mgl_bitrev(p->tail_val & ((1u << e) - 1u), e));
	*bit = (p->tail_val >> e) & 1u;
}

/* probability_model.c:5-15 */
MGL_HD uint32_t mgl_prob_update(uint32_t v, uint32_t bit)
{
	/* bit ? v - (v >> 5) : v + ((2048 - v) >> 5), written so that the chain through v is three
	 * operations deep (every re-simulation loop is a chain of these): -(v >> 5) == (31 - v) >> 5 with an
	 * arithmetic shift, for every v in 0..2048 (checked exhaustively in tests/test_host.py) */
	const int32_t c = bit ? 31 : 2048;
	return (uint32_t)((int32_t)v + ((c - (int32_t)v) >> 5));
}

/* counter-based RNG of the batched SA semantics */
MGL_HD uint64_t mgl_mix64(uint64_t z)
{
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
MGL_HD uint64_t mgl_rng_key(uint64_t seed, uint64_t step, uint32_t j)
{
	return mgl_mix64(seed ^ mgl_mix64(step * 0x100000001B3ull + j));
}
MGL_HD uint32_t mgl_rng_draw(uint64_t key, uint32_t n)
{
	return (uint32_t)(mgl_mix64(key + n) >> 33);
}

#endif /* MGL_MODEL_H */
