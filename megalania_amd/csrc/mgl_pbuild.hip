/*
 * mgl_pbuild.hip -- deriving the base structures of a slab in parallel.
 *
 * k_build (mgl_kernels2.hip) walks the slab with one wavefront: fine for a per-step fallback on
 * small inputs, 135 ms at 100 KB and 14 s at 10 MB.  The same structures, bit for bit, from a
 * pipeline of wide kernels over blocks of 2^shift (256..1024) positions:
 *
 *   pb_exits    per block, for each of the 273 offsets a walk can enter it at: where it leaves
 *   pb_entries  chase those maps block to block: the first on-walk position of every block
 *   pb_mark     one thread per block: on-walk and special bitmaps, packet count, and the block's
 *               effect on (ctx_state, rep distances) as a composable transform
 *   pb_scan     compose the transforms in block order: the walk state every block starts in
 *   pb_levels   summary levels of the special bitmap
 *   pb_walk     one wavefront per block: special-state records, events per context, every event staged with its rank
 *   pb_offsets  per context, exclusive scan of the per-block counts; pb_layout: chain offsets
 *   pb_scatter  one thread per staged event: to its slot of its context's chain
 *   pb_sim      per 2048-event chain segment: the probability before each event, the cost (+ pb_sim_fix)
 *   pb_ckpt     dense checkpoints read off the chains, written as 128-byte rows
 *   pb_finish   totals into Control
 *
 * Everything the walk needs at a block boundary is carried by the scans, and both scans (entry offsets, walk states) are
 * compositions of maps done in groups, so no kernel is serial in the file size.  Integer work throughout; results are
 * identical to k_build's (tests/test_gpu_incremental.py).
 */
#include "mgl_base2.h"

#define MGL_PB_MAX_SHIFT 10u  /* block = 2^shift positions, shift chosen per input size (8..10) */
#define MGL_PB_MAX_BLOCK (1u << MGL_PB_MAX_SHIFT)
#ifndef MGL_PB_SEG
#define MGL_PB_SEG 2048u      /* pb_sim: events per thread ... */
#endif
#ifndef MGL_PB_WARM
#define MGL_PB_WARM 1024u     /* ... after this many events of warm-up from both ends of the probability range */
#endif
#define MGL_PB_ENTRIES 273u /* a packet starting before a block boundary ends at most 272 bytes past it */
#define MGL_PB_CK_ROWS 64u

struct PBuild {
	uint32_t nblk, shift;
	uint16_t* exits;    /* nblk x MGL_PB_ENTRIES: offset past the block end a walk entering at offset e leaves at */
	uint32_t* entry;    /* nblk + 1: first on-walk position >= blk << shift */
	uint16_t* gexits;   /* ngrp x MGL_PB_ENTRIES: the exit maps of MGL_PB_GROUP consecutive blocks composed */
	uint16_t* gentry;   /* ngrp: the entry offset of every group's first block */
	uint32_t* gsum;     /* ceil(nblk / MGL_PB_OFF_ROWS) x ck_elems: pb_offsets' per-group sums, then their exclusive scan */
	uint64_t* ch_map;   /* per chunk of MGL_PB_SCAN_CHUNK blocks: their transforms composed (pb_scan_chunks) ... */
	uint32_t* ch_vs;    /* ... values[4], sources[4] ... */
	uint32_t* ch_pk;    /* ... packets */
	uint32_t* ch_state; /* per chunk x 8: the walk state its first block starts in (pb_scan) */
	uint32_t ngrp;
	uint64_t* tf_ctx;   /* nblk: ctx_state after the block for each of the 12 states before it, 4 bits each */
	uint32_t* tf_dist;  /* nblk x 8: value[4], source[4] (0..3 = rep distance i before the block, 4 = value) */
	uint32_t* tf_pk;    /* nblk: packets starting in the block */
	uint32_t* st_in;    /* nblk x 8: ctx_state, dists[4] at the block's first packet */
	uint32_t* hist;     /* nblk x ck_elems: events per context, then their exclusive scan over blocks */
	uint64_t* stage;    /* nblk x (16 << shift + 32): the events pb_walk found, for pb_scatter */
	uint32_t* stage_n;  /* nblk: how many */
	uint32_t* stage_over; /* one word: a block outgrew its staging area (cannot happen; checked) */
	unsigned long long* acc; /* [0] cost [1] packets [2] final ctx_state [3..6] final dists [7] segments pb_sim_fix redid [8] segments pb_sim left to it */
	uint32_t* seg_off;  /* total + 1: first pb_sim segment of each context */
	uint8_t* unres;     /* per segment: warm-up did not pin the probability, left to pb_sim_fix */
	uint32_t seg_cap;
	uint32_t force_fix; /* diagnostic: treat every warm-up as inconclusive (exercises pb_sim_fix) */
};

/* the validity rule of k_build: anything that is not a packet that fits is walked as a literal */
__device__ __forceinline__ bool pb_decode(mgl_pk pk, uint32_t pos, uint32_t n, uint32_t& type, uint32_t& dist, uint32_t& len)
{
	type = mgl_pk_type(pk); len = mgl_pk_len(pk); dist = mgl_pk_dist(pk);
	if (type < MGL_LITERAL || type > MGL_LONG_REP || len == 0 || pos + len > n) {
		type = MGL_LITERAL; len = 1; dist = 0;
		return false;
	}
	return true;
}

__global__ void __launch_bounds__(320) pb_exits(DevCtx c, Base2 b, PBuild pb)
{
	__shared__ uint16_t lens[MGL_PB_MAX_BLOCK];
	const uint32_t B = 1u << pb.shift;
	const uint32_t blk = blockIdx.x, base = blk << pb.shift;
	const uint32_t lim = (c.n - base) < B ? (c.n - base) : B;
	for (uint32_t i = threadIdx.x; i < B; i += blockDim.x) {
		uint32_t type, dist, len = 1;
		if (i < lim) pb_decode(b.slab[base + i], base + i, c.n, type, dist, len);
		lens[i] = (uint16_t)len;
	}
	__syncthreads();
	if (threadIdx.x < MGL_PB_ENTRIES) {
		uint32_t p = threadIdx.x;
		while (p < lim) p += lens[p];
		pb.exits[(size_t)blk * MGL_PB_ENTRIES + threadIdx.x] = (uint16_t)(p - lim);
	}
}

/* The chase from block to block is a composition of maps (offset a walk enters a block at -> offset it enters the next one
 * at), and composition is associative: MGL_PB_GROUP consecutive blocks are composed into one map per group by a workgroup
 * of its own (one thread per entry offset), one wavefront chases the group maps (n / 2^shift / 64 steps instead of
 * n / 2^shift: 156 instead of 9 953 at 10 MB), and every group then replays its own blocks from its known entry. */
#define MGL_PB_GROUP 64u
__global__ void __launch_bounds__(320) pb_entries_group(PBuild pb)
{
	extern __shared__ uint16_t g_rows[]; /* MGL_PB_GROUP x MGL_PB_ENTRIES */
	const uint32_t g = blockIdx.x, b0 = g * MGL_PB_GROUP;
	const uint32_t cnt = (pb.nblk - b0) < MGL_PB_GROUP ? (pb.nblk - b0) : MGL_PB_GROUP;
	for (uint32_t i = threadIdx.x; i < cnt * MGL_PB_ENTRIES; i += blockDim.x) g_rows[i] = pb.exits[(size_t)b0 * MGL_PB_ENTRIES + i];
	__syncthreads();
	if (threadIdx.x < MGL_PB_ENTRIES) {
		uint32_t e = threadIdx.x;
		for (uint32_t j = 0; j < cnt; j++) e = g_rows[j * MGL_PB_ENTRIES + e];
		pb.gexits[(size_t)g * MGL_PB_ENTRIES + threadIdx.x] = (uint16_t)e;
	}
}
#define MGL_PB_STAGE 16u
__global__ void __launch_bounds__(64) pb_entries(DevCtx c, PBuild pb)
{
	__shared__ uint16_t rows[MGL_PB_STAGE * MGL_PB_ENTRIES];
	const uint32_t lane = threadIdx.x;
	uint32_t e = 0;
	for (uint32_t g = 0; g < pb.ngrp; g += MGL_PB_STAGE) {
		const uint32_t cnt = (pb.ngrp - g) < MGL_PB_STAGE ? (pb.ngrp - g) : MGL_PB_STAGE;
		for (uint32_t i = lane; i < cnt * MGL_PB_ENTRIES; i += 64) rows[i] = pb.gexits[(size_t)g * MGL_PB_ENTRIES + i];
		wave_sync();
		if (lane == 0) {
			for (uint32_t j = 0; j < cnt; j++) {
				pb.gentry[g + j] = (uint16_t)e;
				e = rows[j * MGL_PB_ENTRIES + e];
			}
		}
		e = uni(e);
		wave_sync();
	}
	if (lane == 0) pb.entry[pb.nblk] = c.n;
}
__global__ void __launch_bounds__(64) pb_entries_fill(PBuild pb)
{
	const uint32_t g = blockIdx.x, b0 = g * MGL_PB_GROUP;
	const uint32_t cnt = (pb.nblk - b0) < MGL_PB_GROUP ? (pb.nblk - b0) : MGL_PB_GROUP;
	if (threadIdx.x != 0) return;
	uint32_t e = pb.gentry[g];
	for (uint32_t j = 0; j < cnt; j++) {
		pb.entry[b0 + j] = ((b0 + j) << pb.shift) + e;
		e = pb.exits[(size_t)(b0 + j) * MGL_PB_ENTRIES + e];
	}
}

/* ctx_state automaton (lzma_state.c:29-57) as one nibble per source state */
#define MGL_PB_MAP_ID  0x0000BA9876543210ull
#define MGL_PB_MAP_LIT 0x0000546543210000ull
#define MGL_PB_MAP_MATCH 0x0000AAAAA7777777ull
#define MGL_PB_MAP_SREP 0x0000BBBBB9999999ull
#define MGL_PB_MAP_LREP 0x0000BBBBB8888888ull

/* One wavefront per block: the block's slab entries come into LDS with coalesced loads, then the wavefront follows the packets
 * from the block's entry position there, every lane doing the same (uniform) bookkeeping and lanes 0..11 each following the
 * ctx_state automaton from one source state.  (A thread per block chasing the packets through global memory waited a memory
 * round trip per packet and composed a 12-nibble map per packet on its own: 0.47 ms at 10 MB.) */
__global__ void __launch_bounds__(64) pb_mark(DevCtx c, Base2 b, PBuild pb, Control* ctl)
{
	__shared__ mgl_pk s_slab[MGL_PB_MAX_BLOCK];
	const uint32_t blk = blockIdx.x;
	if (blk >= pb.nblk) return;
	const uint32_t base = blk << pb.shift;
	const uint32_t end = ((blk + 1) << pb.shift) < c.n ? ((blk + 1) << pb.shift) : c.n;
	for (uint32_t i = threadIdx.x; base + i < end; i += blockDim.x) s_slab[i] = b.slab[base + i];
	wave_sync();
	const uint32_t lane = threadIdx.x;
	uint32_t p = uni(pb.entry[blk]);
	const uint32_t w_first = blk << (pb.shift - 6), w_lim = (w_first + (1u << (pb.shift - 6))) < b.nw0 ? (w_first + (1u << (pb.shift - 6))) : b.nw0;
	uint32_t word = w_first;
	uint64_t on = 0, sp = 0;
	/* the ctx_state map, one source state per lane (lane i < 12 follows the automaton from state i: three instructions per
	 * packet for the whole map instead of a 12-nibble loop on one lane -- the walk is issue-bound) */
	uint32_t mst = lane < 12u ? lane : 0u;
	uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0, s0 = 0, s1 = 1, s2 = 2, s3 = 3, npk = 0;
	bool bad = false;
	while (p < end) {
		const uint32_t w = p >> 6;
		if (w != word) {
			if (lane == 0) {
				b.onwalk[word] = on; b.sp0[word] = sp;
				for (uint32_t z = word + 1; z < w; z++) { b.onwalk[z] = 0; b.sp0[z] = 0; }
			}
			word = w; on = 0; sp = 0;
		}
		const uint64_t bit = 1ull << (p & 63u);
		on |= bit;
		uint32_t type, dist, len;
		if (!pb_decode(s_slab[p - base], p, c.n, type, dist, len)) bad = true;
		uint64_t tbl = MGL_PB_MAP_LIT;
		if (type != MGL_LITERAL) {
			sp |= bit;
			tbl = type == MGL_MATCH ? MGL_PB_MAP_MATCH : type == MGL_SHORT_REP ? MGL_PB_MAP_SREP : MGL_PB_MAP_LREP;
			if (type == MGL_MATCH) {
				v3 = v2; s3 = s2; v2 = v1; s2 = s1; v1 = v0; s1 = s0;
				v0 = dist; s0 = 4;
			} else if (type == MGL_LONG_REP) { /* mgl_advance, including its reading of an index > 3 */
				const uint32_t tv = dist == 0 ? v0 : dist == 1 ? v1 : dist == 2 ? v2 : v3;
				const uint32_t ts = dist == 0 ? s0 : dist == 1 ? s1 : dist == 2 ? s2 : s3;
				if (dist > 2) { v3 = v2; s3 = s2; }
				if (dist > 1) { v2 = v1; s2 = s1; }
				if (dist > 0) { v1 = v0; s1 = s0; }
				v0 = tv; s0 = ts;
			}
		}
		mst = (uint32_t)(tbl >> (4u * mst)) & 15u;
		p += len;
		npk++;
	}
	uint64_t map = 0;
#pragma unroll
	for (uint32_t i = 0; i < 12; i++) map |= (uint64_t)rdlane(mst, i) << (4u * i);
	if (lane != 0) return;
	if (word < w_lim) {
		b.onwalk[word] = on; b.sp0[word] = sp;
		for (uint32_t z = word + 1; z < w_lim; z++) { b.onwalk[z] = 0; b.sp0[z] = 0; }
	}
	pb.tf_ctx[blk] = map;
	uint32_t* t = pb.tf_dist + (size_t)blk * 8;
	t[0] = v0; t[1] = v1; t[2] = v2; t[3] = v3; t[4] = s0; t[5] = s1; t[6] = s2; t[7] = s3;
	pb.tf_pk[blk] = npk;
	if (bad) atomicOr(&ctl->error_flags, MGL_ERR_WALK_OVERRUN);
}

/* A block's effect on the walk state: ctx_state map (12 nibbles) and where each rep distance comes from */
struct PbTf {
	uint64_t map;
	uint32_t v0, v1, v2, v3, s0, s1, s2, s3;
};
__device__ __forceinline__ PbTf pb_tf_load(const PBuild& pb, uint32_t blk)
{
	PbTf t;
	t.map = pb.tf_ctx[blk];
	const uint4 tv = ((const uint4*)pb.tf_dist)[(size_t)blk * 2], ts = ((const uint4*)pb.tf_dist)[(size_t)blk * 2 + 1];
	t.v0 = tv.x; t.v1 = tv.y; t.v2 = tv.z; t.v3 = tv.w; t.s0 = ts.x; t.s1 = ts.y; t.s2 = ts.z; t.s3 = ts.w;
	return t;
}
/* state after the block, given the state before it */
__device__ __forceinline__ void pb_tf_apply(const PbTf& t, uint32_t& cs, uint32_t& d0, uint32_t& d1, uint32_t& d2, uint32_t& d3)
{
#define PB_PICK(q, a) ((q) == 4u ? (a) : (q) == 0u ? d0 : (q) == 1u ? d1 : (q) == 2u ? d2 : d3)
	const uint32_t n0 = PB_PICK(t.s0, t.v0), n1 = PB_PICK(t.s1, t.v1), n2 = PB_PICK(t.s2, t.v2), n3 = PB_PICK(t.s3, t.v3);
#undef PB_PICK
	d0 = n0; d1 = n1; d2 = n2; d3 = n3;
	cs = (uint32_t)(t.map >> (4 * cs)) & 15u;
}
/* a := a followed by t */
__device__ __forceinline__ void pb_tf_then(PbTf& a, const PbTf& t)
{
	uint64_t r = 0;
#pragma unroll
	for (uint32_t i = 0; i < 12; i++) {
		const uint32_t s = (uint32_t)(a.map >> (4 * i)) & 15u;
		r |= ((t.map >> (4 * s)) & 15ull) << (4 * i);
	}
	a.map = r;
#define PB_SRC_V(q, cv) ((q) == 4u ? (cv) : (q) == 0u ? a.v0 : (q) == 1u ? a.v1 : (q) == 2u ? a.v2 : a.v3)
#define PB_SRC_S(q) ((q) == 4u ? 4u : (q) == 0u ? a.s0 : (q) == 1u ? a.s1 : (q) == 2u ? a.s2 : a.s3)
	const uint32_t nv0 = PB_SRC_V(t.s0, t.v0), nv1 = PB_SRC_V(t.s1, t.v1), nv2 = PB_SRC_V(t.s2, t.v2), nv3 = PB_SRC_V(t.s3, t.v3);
	const uint32_t ns0 = PB_SRC_S(t.s0), ns1 = PB_SRC_S(t.s1), ns2 = PB_SRC_S(t.s2), ns3 = PB_SRC_S(t.s3);
#undef PB_SRC_V
#undef PB_SRC_S
	a.v0 = nv0; a.v1 = nv1; a.v2 = nv2; a.v3 = nv3; a.s0 = ns0; a.s1 = ns1; a.s2 = ns2; a.s3 = ns3;
}

__device__ __forceinline__ PbTf pb_tf_shfl_up(const PbTf& a, int o)
{
	PbTf t;
	t.map = shfl_up64(a.map, o);
	t.v0 = (uint32_t)__shfl_up((int)a.v0, o, 64); t.v1 = (uint32_t)__shfl_up((int)a.v1, o, 64);
	t.v2 = (uint32_t)__shfl_up((int)a.v2, o, 64); t.v3 = (uint32_t)__shfl_up((int)a.v3, o, 64);
	t.s0 = (uint32_t)__shfl_up((int)a.s0, o, 64); t.s1 = (uint32_t)__shfl_up((int)a.s1, o, 64);
	t.s2 = (uint32_t)__shfl_up((int)a.s2, o, 64); t.s3 = (uint32_t)__shfl_up((int)a.s3, o, 64);
	return t;
}
/* The walk state every block starts in: the blocks' transforms composed in order.  Three kernels: every thread composes
 * MGL_PB_SCAN_CHUNK consecutive blocks into one transform (wide), one wavefront applies the chunk transforms in order, 64
 * per trip (n / 2^shift / 8 steps of register work: 1 244 at 10 MB), every thread replays its chunk from its own entry
 * state (wide).  (One wavefront doing all three with 64 blocks per lane took 0.38 ms at 10 MB.) */
#define MGL_PB_SCAN_CHUNK 8u
__global__ void __launch_bounds__(256) pb_scan_chunks(PBuild pb)
{
	const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t first = ch * MGL_PB_SCAN_CHUNK;
	if (first >= pb.nblk) return;
	const uint32_t last = (first + MGL_PB_SCAN_CHUNK) < pb.nblk ? (first + MGL_PB_SCAN_CHUNK) : pb.nblk;
	PbTf acc;
	acc.map = MGL_PB_MAP_ID; acc.v0 = acc.v1 = acc.v2 = acc.v3 = 0; acc.s0 = 0; acc.s1 = 1; acc.s2 = 2; acc.s3 = 3;
	uint32_t packets = 0;
	for (uint32_t blk = first; blk < last; blk++) {
		const PbTf t = pb_tf_load(pb, blk);
		pb_tf_then(acc, t);
		packets += pb.tf_pk[blk];
	}
	pb.ch_map[ch] = acc.map;
	((uint4*)pb.ch_vs)[(size_t)ch * 2] = make_uint4(acc.v0, acc.v1, acc.v2, acc.v3);
	((uint4*)pb.ch_vs)[(size_t)ch * 2 + 1] = make_uint4(acc.s0, acc.s1, acc.s2, acc.s3);
	pb.ch_pk[ch] = packets;
}
__global__ void __launch_bounds__(64) pb_scan(PBuild pb)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t nch = (pb.nblk + MGL_PB_SCAN_CHUNK - 1u) / MGL_PB_SCAN_CHUNK;
	uint32_t cs = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0; /* uniform running state */
	unsigned long long packets = 0;
	for (uint32_t g = 0; g < nch; g += 64) {
		const uint32_t ch = g + lane;
		PbTf acc;
		acc.map = MGL_PB_MAP_ID; acc.v0 = acc.v1 = acc.v2 = acc.v3 = 0; acc.s0 = 0; acc.s1 = 1; acc.s2 = 2; acc.s3 = 3;
		if (ch < nch) {
			acc.map = pb.ch_map[ch];
			const uint4 tv = ((const uint4*)pb.ch_vs)[(size_t)ch * 2], ts = ((const uint4*)pb.ch_vs)[(size_t)ch * 2 + 1];
			acc.v0 = tv.x; acc.v1 = tv.y; acc.v2 = tv.z; acc.v3 = tv.w; acc.s0 = ts.x; acc.s1 = ts.y; acc.s2 = ts.z; acc.s3 = ts.w;
			packets += pb.ch_pk[ch];
		}
		/* composition is associative: an inclusive scan over the lanes (six shuffle-and-compose steps) instead of 64
		 * transforms applied one after the other */
		for (int o = 1; o < 64; o <<= 1) {
			PbTf t = pb_tf_shfl_up(acc, o);
			if ((int)lane >= o) { pb_tf_then(t, acc); acc = t; } /* (lanes lane-2o+1 .. lane-o) then (lane-o+1 .. lane) */
		}
		/* this lane's chunk starts in the running state put through the chunks before it in this trip */
		PbTf ex = pb_tf_shfl_up(acc, 1);
		uint32_t my_cs = cs, m0 = d0, m1 = d1, m2 = d2, m3 = d3;
		if (lane != 0) pb_tf_apply(ex, my_cs, m0, m1, m2, m3);
		/* and the running state moves on through all 64 */
		{
			PbTf all;
			all.map = rdlane64(acc.map, 63);
			all.v0 = rdlane(acc.v0, 63); all.v1 = rdlane(acc.v1, 63); all.v2 = rdlane(acc.v2, 63); all.v3 = rdlane(acc.v3, 63);
			all.s0 = rdlane(acc.s0, 63); all.s1 = rdlane(acc.s1, 63); all.s2 = rdlane(acc.s2, 63); all.s3 = rdlane(acc.s3, 63);
			pb_tf_apply(all, cs, d0, d1, d2, d3);
		}
		if (ch < nch) {
			((uint4*)pb.ch_state)[(size_t)ch * 2] = make_uint4(my_cs, m0, m1, m2);
			((uint4*)pb.ch_state)[(size_t)ch * 2 + 1] = make_uint4(m3, 0, 0, 0);
		}
	}
	packets = wave_sum64(packets);
	if (lane == 0) {
		pb.acc[0] = 0;
		pb.acc[1] = packets;
		pb.acc[2] = cs; pb.acc[3] = d0; pb.acc[4] = d1; pb.acc[5] = d2; pb.acc[6] = d3; pb.acc[7] = 0; pb.acc[8] = 0;
	}
}
__global__ void __launch_bounds__(256) pb_scan_fill(PBuild pb)
{
	const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t first = ch * MGL_PB_SCAN_CHUNK;
	if (first >= pb.nblk) return;
	const uint32_t last = (first + MGL_PB_SCAN_CHUNK) < pb.nblk ? (first + MGL_PB_SCAN_CHUNK) : pb.nblk;
	const uint4 s0 = ((const uint4*)pb.ch_state)[(size_t)ch * 2], s1 = ((const uint4*)pb.ch_state)[(size_t)ch * 2 + 1];
	uint32_t my_cs = s0.x, m0 = s0.y, m1 = s0.z, m2 = s0.w, m3 = s1.x;
	for (uint32_t blk = first; blk < last; blk++) {
		uint4* o = (uint4*)(pb.st_in + (size_t)blk * 8);
		o[0] = make_uint4(my_cs, m0, m1, m2);
		o[1] = make_uint4(m3, 0, 0, 0);
		const PbTf t = pb_tf_load(pb, blk);
		pb_tf_apply(t, my_cs, m0, m1, m2, m3);
	}
}

/* level[w >> 6] bit (w & 63) = lower[w] != 0; the lower array is zero padded to a multiple of 64 words */
__global__ void __launch_bounds__(256) pb_levels(const uint64_t* lower, uint64_t* level, uint32_t nlower, uint32_t nlevel)
{
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned long long m = __ballot(w < nlower && lower[w] != 0);
	if ((threadIdx.x & 63u) == 0 && (w >> 6) < nlevel) level[w >> 6] = m;
}

/* One wavefront walks one block from its entry state.  SCATTER = false: special-state records +
 * events per context (pb.hist row) + direct-bit cost.  SCATTER = true: events into the chains. */
/* One wavefront walks one block from its entry state: special-state records, events per context (pb.hist row), direct-bit
 * cost -- and every event, with its rank among the block's events of its context, into the block's staging area
 * (pos | ctx << 32 | bit << 46 | rank << 47), so that nothing has to walk the block a second time: once the per-context
 * offsets are known, pb_scatter puts every staged event into its chain slot, one thread per event. */
#define MGL_PB_STAGE_PER_POS 16u /* staged events per position of a block: the densest parse (length-2 far matches, 27 events each) has 13.5 */
__device__ __forceinline__ uint64_t pb_stage_pack(uint32_t pos, uint32_t ctx, uint32_t bit, uint32_t rank)
{
	return (uint64_t)pos | ((uint64_t)ctx << 32) | ((uint64_t)bit << 46) | ((uint64_t)rank << 47);
}
__global__ void __launch_bounds__(64) pb_walk(DevCtx c, Base2 b, PBuild pb)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint32_t* cnt = (uint32_t*)smem;
	const uint32_t lane = threadIdx.x, blk = blockIdx.x;
	uint32_t* row = pb.hist + (size_t)blk * b.ck_elems;
	const uint32_t cap = (MGL_PB_STAGE_PER_POS << pb.shift) + 32u;
	uint64_t* stage = pb.stage + (size_t)blk * cap;
	uint32_t nst = 0; /* staged so far (uniform) */
	for (uint32_t i = lane; i < b.ck_elems; i += 64) cnt[i] = 0u;
	wave_sync();
	Walk w;
	walk_reset(w);
	{
		const uint32_t* s = pb.st_in + (size_t)blk * 8;
		w.st.pos = pb.entry[blk];
		w.st.ctx_state = s[0];
		w.st.dists[0] = s[1]; w.st.dists[1] = s[2]; w.st.dists[2] = s[3]; w.st.dists[3] = s[4];
	}
	const uint32_t end = ((blk + 1) << pb.shift) < c.n ? ((blk + 1) << pb.shift) : c.n;
	uint32_t ndirect = 0;
	while (w.st.pos < end) {
		const uint32_t pos = w.st.pos;
		walk_window(w, c, b.slab, lane);
		if (w.st.ctx_state < 7u) {
			/* a run of plain literals: up to seven are planned at once, nine lanes each (is_match + the eight tree nodes);
			 * their events take their ranks packet by packet, because the packets share contexts (is_match, the top
			 * of the literal tree) and a chain is in position order -- the model replay of mgl_kernels2.hip does the same */
			const uint32_t o = pos - w.wbase;
			uint32_t lt, ld, ll;
			const uint32_t myp = w.wbase + lane;
			pb_decode(w.wpk, myp, c.n, lt, ld, ll);
			const unsigned long long lit = __ballot(myp < c.n && lt == MGL_LITERAL) >> o;
			uint32_t run = ~lit == 0ull ? 64u : (uint32_t)__ffsll((long long)~lit) - 1u;
			run = run < 64u - o ? run : 64u - o;
			run = run < end - pos ? run : end - pos;
			if (run >= 2u) {
				const uint32_t take = run < 7u ? run : 7u;
				const uint32_t i = lane / 9u, slot = lane - i * 9u, p = pos + i;
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
					uint32_t ctx = 0, bit = 0, k = 0;
					if (active) mgl_plan_event(&pl, slot, &ctx, &bit);
					for (uint32_t r = 0; r < take; r++) {
						if (active && i == r) k = cnt[ctx]++;
						wave_sync();
					}
					if (active && nst + lane < cap) stage[nst + lane] = pb_stage_pack(p, ctx, bit, k);
					nst += 9u * take;
					w.st.pos += take; w.st.ctx_state = lit_steps(w.st.ctx_state, take);
					continue;
				}
			}
		}
		uint32_t type, dist, len;
		pb_decode(walk_slab_at(w, pos), pos, c.n, type, dist, len);
		if (type != MGL_LITERAL && lane < 8) {
			const uint32_t v = lane == 0 ? w.st.ctx_state : lane == 1 ? w.st.dists[0] : lane == 2 ? w.st.dists[1]
			                 : lane == 3 ? w.st.dists[2] : lane == 4 ? w.st.dists[3] : 0u;
			b.sp_state[(size_t)pos * 8 + lane] = v;
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
			const uint32_t k = cnt[ctx]++; /* contexts of one packet are distinct: no conflict */
			if (nst + lane < cap) stage[nst + lane] = pb_stage_pack(pos, ctx, bit, k);
		}
		nst += pl.nev;
		ndirect += pl.ndirect;
		mgl_advance(&w.st, type, dist, len);
	}
	wave_sync();
	for (uint32_t i = lane; i < b.ck_elems; i += 64) row[i] = cnt[i];
	if (lane == 0) {
		pb.stage_n[blk] = nst < cap ? nst : cap; /* (cap is never reached: see MGL_PB_STAGE_PER_POS) */
		if (nst > cap) atomicOr(pb.stage_over, 1u);
		if (ndirect) atomicAdd(&pb.acc[0], (unsigned long long)ndirect << 11);
	}
}
/* every staged event to its slot: chain offset of its context + events of that context in earlier blocks + its rank in the block */
__global__ void __launch_bounds__(256) pb_scatter(Base2 b, PBuild pb, uint32_t total)
{
	if (*b.pool_top > b.pool_cap) return;
	const uint32_t blk = blockIdx.x;
	const uint32_t cap = (MGL_PB_STAGE_PER_POS << pb.shift) + 32u;
	const uint64_t* stage = pb.stage + (size_t)blk * cap;
	const uint32_t* row = pb.hist + (size_t)blk * b.ck_elems;
	const uint32_t n = pb.stage_n[blk];
	extern __shared__ __attribute__((aligned(16))) uint32_t s_first[]; /* per context: the slot of this block's first event of it */
	for (uint32_t i = threadIdx.x; i < b.ck_elems; i += blockDim.x) s_first[i] = i < total ? b.ch_off[i] + row[i] : 0u;
	__syncthreads();
	for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
		const uint64_t v = stage[e];
		const uint32_t pos = (uint32_t)v, ctx = (uint32_t)(v >> 32) & 0x3FFFu, bit = (uint32_t)(v >> 46) & 1u, rank = (uint32_t)(v >> 47);
		const uint32_t k = s_first[ctx] + rank;
		b.ch_pos[k] = pos;
		b.ch_ev[k] = (uint16_t)(bit << 15);
	}
}

/* per context: counts per block -> offset of the block's first event in the context's chain.  A column scan over all
 * blocks per context is 2 616 threads walking 10⁴ rows each (0.6 ms at 10 MB): done in groups of MGL_PB_OFF_ROWS rows
 * instead -- sums per (context, group), a short scan over the groups, then every group rescans its own rows. */
#define MGL_PB_OFF_ROWS 128u
__global__ void __launch_bounds__(256) pb_offsets_sum(Base2 b, PBuild pb)
{
	const uint32_t ctx = blockIdx.x * blockDim.x + threadIdx.x;
	if (ctx >= b.ck_elems) return;
	const uint32_t r0 = blockIdx.y * MGL_PB_OFF_ROWS, r1 = (r0 + MGL_PB_OFF_ROWS) < pb.nblk ? (r0 + MGL_PB_OFF_ROWS) : pb.nblk;
	const uint32_t* col = pb.hist + ctx;
	const size_t E = b.ck_elems;
	uint32_t sum = 0;
	for (uint32_t blk = r0; blk < r1; blk++) sum += col[(size_t)blk * E];
	pb.gsum[(size_t)blockIdx.y * E + ctx] = sum;
}
__global__ void __launch_bounds__(256) pb_offsets_top(Base2 b, PBuild pb, uint32_t total, uint32_t ngroups)
{
	const uint32_t ctx = blockIdx.x * blockDim.x + threadIdx.x;
	if (ctx >= b.ck_elems) return;
	const size_t E = b.ck_elems;
	uint32_t run = 0;
	for (uint32_t g = 0; g < ngroups; g++) { const uint32_t t = pb.gsum[(size_t)g * E + ctx]; pb.gsum[(size_t)g * E + ctx] = run; run += t; }
	if (ctx < total) b.ch_len[ctx] = run;
}
__global__ void __launch_bounds__(256) pb_offsets(Base2 b, PBuild pb, uint32_t total)
{
	const uint32_t ctx = blockIdx.x * blockDim.x + threadIdx.x;
	if (ctx >= b.ck_elems) return;
	const uint32_t r0 = blockIdx.y * MGL_PB_OFF_ROWS, r1 = (r0 + MGL_PB_OFF_ROWS) < pb.nblk ? (r0 + MGL_PB_OFF_ROWS) : pb.nblk;
	const size_t E = b.ck_elems;
	uint32_t run = pb.gsum[(size_t)blockIdx.y * E + ctx];
	uint32_t* col = pb.hist + ctx;
	uint32_t blk = r0;
	for (; blk + 8 <= r1; blk += 8) {
		uint32_t t[8];
#pragma unroll
		for (int u = 0; u < 8; u++) t[u] = col[(size_t)(blk + u) * E];
#pragma unroll
		for (int u = 0; u < 8; u++) { col[(size_t)(blk + u) * E] = run; run += t[u]; }
	}
	for (; blk < r1; blk++) { const uint32_t t = col[(size_t)blk * E]; col[(size_t)blk * E] = run; run += t; }
	(void)total;
}

/* the chain index: pb.hist after pb_offsets is "events of the context in earlier blocks", row per block -- the index is its
 * transpose, row per context (one more column: the chain's length) */
__global__ void __launch_bounds__(256) pb_index(Base2 b, PBuild pb, uint32_t total)
{
	__shared__ uint32_t tile[32][33];
	const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5; /* 32 x 8 */
	const uint32_t c0 = blockIdx.x * 32u, b0 = blockIdx.y * 32u;
	const size_t E = b.ck_elems;
	for (uint32_t r = ty; r < 32u; r += 8u) {
		const uint32_t blk = b0 + r, ctx = c0 + tx;
		tile[r][tx] = (blk < pb.nblk && ctx < total) ? pb.hist[(size_t)blk * E + ctx] : 0u;
	}
	__syncthreads();
	for (uint32_t r = ty; r < 32u; r += 8u) {
		const uint32_t ctx = c0 + r, blk = b0 + tx;
		if (ctx < total && blk < pb.nblk) b.ch_sb[(size_t)ctx * b.sb_stride + blk] = tile[tx][r];
		if (ctx < total && blk == pb.nblk) b.ch_sb[(size_t)ctx * b.sb_stride + blk] = b.ch_len[ctx];
	}
}

/* chain offsets; capacity = 2 len + 257 rounded up to 8 entries, as in k_build */
__device__ __forceinline__ uint32_t pb_nseg(uint32_t len) { return len ? (len + MGL_PB_SEG - 1u) / MGL_PB_SEG : 1u; }
__global__ void __launch_bounds__(64) pb_layout(Base2 b, PBuild pb, Control* ctl, uint32_t total)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t per = (total + 63u) / 64u;
	const uint32_t lo = lane * per, hi = (lo + per) < total ? (lo + per) : total;
	uint32_t sum = 0, ssum = 0;
	for (uint32_t i = lo; i < hi; i++) { sum += (2u * b.ch_len[i] + 257u + 7u) & ~7u; ssum += pb_nseg(b.ch_len[i]); }
	uint32_t incl = sum, sincl = ssum;
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
		const uint32_t u = (uint32_t)__shfl_up((int)sincl, o, 64);
		if ((int)lane >= o) { incl += t; sincl += u; }
	}
	uint32_t run = incl - sum, srun = sincl - ssum;
	for (uint32_t i = lo; i < hi; i++) {
		const uint32_t cap = (2u * b.ch_len[i] + 257u + 7u) & ~7u;
		b.ch_off[i] = run; b.ch_cap[i] = cap;
		run += cap;
		pb.seg_off[i] = srun;
		srun += pb_nseg(b.ch_len[i]);
	}
	if (lane == 63) pb.seg_off[total] = sincl;
	const uint32_t all = (uint32_t)__shfl((int)incl, 63, 64);
	if (lane == 0) {
		*b.pool_top = all;
		if (all > b.pool_cap) atomicOr(&ctl->error_flags, MGL_ERR_WALK_OVERRUN);
	}
}

/* The probability before every event of every chain, the sentinels, the cost.  A chain is cut into
 * segments of MGL_PB_SEG events, one thread each.  The update v -= v >> 5 / v += (2048 - v) >> 5 is
 * monotone in v, so running both ends of the reachable range (31 and 2017) through the
 * MGL_PB_WARM events before a segment brackets the true value; when the two meet -- they nearly
 * always do within a few hundred events -- the value at the segment start is exact without the
 * chain's history.  Segments where they have not met are left to pb_sim_fix. */
__device__ __forceinline__ void pb_sim_range(uint16_t* ev, uint32_t i0, uint32_t i1, uint32_t& p, unsigned long long& cost, const uint16_t* T)
{
	for (uint32_t i = i0; i < i1; i += 8) { /* chains are 16-byte aligned, i0 is a multiple of 8 */
		uint4 v = *(const uint4*)(ev + i);
		uint32_t wds[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (uint32_t k = 0; k < 8; k++) {
			if (i + k < i1) {
				const uint32_t e = (wds[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
				const uint32_t bit = e >> 15;
				wds[k >> 1] = (wds[k >> 1] & ~(0xFFFFu << (16 * (k & 1)))) | (((bit << 15) | p) << (16 * (k & 1)));
				cost += T[bit ? 2048u - p : p];
				p = mgl_prob_update(p, bit);
			}
		}
		*(uint4*)(ev + i) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
	}
}
__global__ void __launch_bounds__(64) pb_sim(DevCtx c, Base2 b, PBuild pb)
{
	__shared__ uint16_t T[2048];
	for (uint32_t i = threadIdx.x; i < 2048; i += 64) T[i] = c.cost_tbl[i];
	__syncthreads();
	if (*b.pool_top > b.pool_cap) return;
	const uint32_t total = c.L.total;
	const uint32_t seg = blockIdx.x * 64 + threadIdx.x;
	unsigned long long cost = 0;
	if (seg < pb.seg_off[total] && seg < pb.seg_cap) {
		uint32_t lo = 0, hi = total; /* last context with seg_off <= seg */
		while (hi - lo > 1) {
			const uint32_t mid = (lo + hi) >> 1;
			if (pb.seg_off[mid] <= seg) lo = mid; else hi = mid;
		}
		const uint32_t ctx = lo, s = seg - pb.seg_off[ctx];
		const uint32_t off = b.ch_off[ctx], len = b.ch_len[ctx];
		uint16_t* ev = b.ch_ev + off;
		const uint32_t i0 = s * MGL_PB_SEG, i1 = (i0 + MGL_PB_SEG) < len ? (i0 + MGL_PB_SEG) : len;
		uint32_t p = MGL_PROB_INIT;
		bool ok = true;
		if (i0 != 0) {
			uint32_t plo = 31u, phi = 2017u;
			for (uint32_t i = i0 - MGL_PB_WARM; i < i0; i += 8) {
				const uint4 v = *(const uint4*)(ev + i);
				const uint32_t wds[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
				for (uint32_t k = 0; k < 8; k++) {
					const uint32_t bit = (wds[k >> 1] >> (16 * (k & 1) + 15)) & 1u;
					plo = mgl_prob_update(plo, bit);
					phi = mgl_prob_update(phi, bit);
				}
			}
			ok = plo == phi && !pb.force_fix;
			p = plo;
		}
		pb.unres[seg] = ok ? 0 : 1;
		if (!ok) atomicAdd(&pb.acc[8], 1ull); /* pb_sim_fix returns at once when nothing is left to it (as good as always) */
		if (ok) {
			pb_sim_range(ev, i0, i1, p, cost, T);
			if (i1 == len) { ev[len] = (uint16_t)p; b.ch_pos[off + len] = MGL_POS_INF; }
		}
	}
	cost = wave_sum64(cost);
	if (threadIdx.x == 0 && cost) atomicAdd(&pb.acc[0], cost);
}
/* the segments pb_sim could not start: in chain order, from the entry before them */
__global__ void __launch_bounds__(64) pb_sim_fix(DevCtx c, Base2 b, PBuild pb)
{
	if (pb.acc[8] == 0) return; /* every segment started exact */
	__shared__ uint16_t T[2048];
	for (uint32_t i = threadIdx.x; i < 2048; i += 64) T[i] = c.cost_tbl[i];
	__syncthreads();
	if (*b.pool_top > b.pool_cap) return;
	const uint32_t ctx = blockIdx.x * 64 + threadIdx.x;
	unsigned long long cost = 0, redone = 0;
	if (ctx < c.L.total) {
		const uint32_t s0 = pb.seg_off[ctx], s1 = pb.seg_off[ctx + 1];
		const uint32_t off = b.ch_off[ctx], len = b.ch_len[ctx];
		uint16_t* ev = b.ch_ev + off;
		for (uint32_t sg = s0 + 1; sg < s1 && sg < pb.seg_cap; sg++) {
			if (!pb.unres[sg]) continue;
			const uint32_t i0 = (sg - s0) * MGL_PB_SEG, i1 = (i0 + MGL_PB_SEG) < len ? (i0 + MGL_PB_SEG) : len;
			const uint32_t e = ev[i0 - 1];
			uint32_t p = mgl_prob_update(e & 0x7FFFu, e >> 15);
			pb_sim_range(ev, i0, i1, p, cost, T);
			if (i1 == len) { ev[len] = (uint16_t)p; b.ch_pos[off + len] = MGL_POS_INF; }
			redone++;
		}
	}
	cost = wave_sum64(cost);
	redone = wave_sum64(redone);
	if (threadIdx.x == 0 && cost) atomicAdd(&pb.acc[0], cost);
	if (threadIdx.x == 0 && redone) atomicAdd(&pb.acc[7], redone);
}

/* dense checkpoints: row k = every context's probability before the first packet at or after
 * 64 k = the value stored with the context's first chain entry at a position >= 64 k (the
 * sentinel holds the final value).  Lane = context; rows leave through LDS as 128-byte stores. */
__global__ void __launch_bounds__(64) pb_ckpt(DevCtx c, Base2 b)
{
	__shared__ __attribute__((aligned(16))) uint16_t tile[MGL_PB_CK_ROWS * 64];
	if (*b.pool_top > b.pool_cap) return;
	const uint32_t lane = threadIdx.x, ctx = blockIdx.x * 64 + lane;
	const uint32_t k0 = blockIdx.y * MGL_PB_CK_ROWS, k1 = (k0 + MGL_PB_CK_ROWS) < b.nck ? (k0 + MGL_PB_CK_ROWS) : b.nck;
	if (ctx < c.L.total) {
		const uint32_t off = b.ch_off[ctx], len = b.ch_len[ctx];
		const uint32_t* cp = b.ch_pos + off;
		const uint16_t* ce = b.ch_ev + off;
		/* first entry with position >= 64 k0 (len = the sentinel): 8-ary steps, seven probes per round trip */
		uint32_t idx = chain_lower_bound(cp, len, k0 << MGL_CK2_SHIFT);
		uint32_t pos = cp[idx];
		uint16_t val = ce[idx] & 0x7FFFu; /* re-read only when the entry changes: most contexts keep one entry for all 64 rows */
		for (uint32_t k = k0; k < k1; k++) {
			const uint32_t at = k << MGL_CK2_SHIFT;
			/* positions in a chain are strictly increasing and pos >= at - 64: the entry wanted is at most 64 further on
			 * (or the sentinel, whose position is MGL_POS_INF).  Eight entries per round trip instead of a six-step binary
			 * search: the busiest contexts have about nine entries per row, most have none (chains are 32-byte aligned
			 * and over-allocated past their sentinel, like everywhere else) */
			while (pos < at) {
				const uint32_t base = idx + 1u; /* entries base .. base + 7, clamped to the sentinel (never below `at`) */
				uint32_t v[8];
#pragma unroll
				for (uint32_t e = 0; e < 8; e++) v[e] = cp[base + e < len ? base + e : len];
				uint32_t below = 0; /* the entries below `at` are a prefix */
#pragma unroll
				for (uint32_t e = 0; e < 8; e++) below += v[e] < at ? 1u : 0u;
				if (below == 8u) { idx = base + 7u; pos = v[7]; continue; } /* all eight: carry on from the last of them */
				idx = base + below;
#pragma unroll
				for (uint32_t e = 0; e < 8; e++) if (e == below) pos = v[e];
				val = ce[idx] & 0x7FFFu;
			}
			tile[(k - k0) * 64 + lane] = val;
		}
	} else {
		for (uint32_t k = k0; k < k1; k++) tile[(k - k0) * 64 + lane] = MGL_PROB_INIT;
	}
	__syncthreads();
	/* rows leave as 16-byte stores: eight lanes cover a row's 64 contexts, a store instruction eight rows (rows are 16-byte
	 * multiples; 128 bytes per instruction -- one row, two bytes a lane -- took 64 instructions per tile) */
	const uint32_t seg = lane & 7u, c0 = blockIdx.x * 64u + seg * 8u;
	if (c0 < b.ck_elems)
		for (uint32_t k = k0 + (lane >> 3); k < k1; k += 8u)
			*reinterpret_cast<uint4*>(b.ck_probs + (size_t)k * b.ck_elems + c0) = *reinterpret_cast<const uint4*>(tile + (k - k0) * 64u + seg * 8u);
}

__global__ void pb_finish(PBuild pb, Control* ctl)
{
	if (threadIdx.x || blockIdx.x) return;
	if (*pb.stage_over) atomicOr(&ctl->error_flags, MGL_ERR_WALK_OVERRUN); /* a block outgrew its staging area: cannot happen */
	ctl->packets = pb.acc[1];
	ctl->rebuild_cost = pb.acc[0];
	ctl->final_ctx_state = (uint32_t)pb.acc[2];
	for (int i = 0; i < 4; i++) ctl->final_dists[i] = (uint32_t)pb.acc[3 + i];
}
