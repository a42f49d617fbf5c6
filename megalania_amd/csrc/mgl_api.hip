/*
 * mgl_api.hip -- the C ABI of include/megalania_hip.h over the kernels in mgl_kernels.hip.
 * Owns all device memory; one HIP stream per handle; no host thread is created.
 */
#include "mgl_kernels.hip"
#include "mgl_kernels2.hip"
#include "mgl_kernels3.hip"
#include "mgl_kernels4.hip"
#include "mgl_kernels5.hip"
#include "mgl_pbuild.hip"
#include "mgl_index.hip"
#include "../../include/megalania_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

static const uint16_t k_cost_table[2048] = {
#include "mgl_cost_table.inc"
};

#define MGL_MAX_INPUT 176000000ull
static thread_local std::string g_err;
static int fail(int code, const std::string& msg)
{
	g_err = msg;
	return code;
}
#define HIPCHK(expr)                                                                         \
	do {                                                                                     \
		hipError_t e_ = (expr);                                                              \
		if (e_ != hipSuccess)                                                                \
			return fail(MGL_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
	} while (0)

/* Look-ahead (opt-in, MGL_LOOKAHEAD=1) keeps five streams busy -- the step's own chain, the second slice, the re-simulations,
 * two speculative slices -- and streams that share a hardware queue run one behind the other: more queues than the runtime's
 * default of four, set before the runtime initialises (its first call) and only if the user has not set the variable. */
__attribute__((constructor)) static void mgl_more_hw_queues(void) { if (getenv("MGL_LOOKAHEAD")) setenv("GPU_MAX_HW_QUEUES", "8", 0); }

extern "C" const char* mgl_version(void) { return "megalania-hip 0.1 (gfx950)"; }
extern "C" const char* mgl_last_error(void) { return g_err.c_str(); }
extern "C" int mgl_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}
extern "C" uint32_t mgl_rng_draw_at(uint64_t seed, uint64_t step, uint32_t j, uint32_t n)
{
	return mgl_rng_draw(mgl_rng_key(seed, step, j), n);
}

struct BaseMem {
	BaseView v;
	Control* ctl;
};

struct mgl_sa {
	int device;
	hipStream_t stream;
	hipStream_t stream2;     /* second half of a step's neighbours: its kernels fill the other half's tails */
	hipEvent_t ev_fork, ev_join;
	hipEvent_t ev_tgt = nullptr, ev_tgt_go = nullptr; /* the next step's targets worked out beside the rest of this step's accept (launch_targets_ahead) */
	int tgt_ahead = 0;             /* 0 none, 1 queued and good for the next launch_neighbours, 2 queued but the base went another way: to be made again */
	hipStream_t stream3 = nullptr;   /* the re-simulation kernels of the split form: beside the second pass, which only needs the walks */
	hipEvent_t ev_rest[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr }, ev_sim = nullptr;
	uint32_t halves;
	uint32_t n;
	mgl_properties props;
	mgl_sa_config cfg;
	uint64_t temperature = 0; /* mgl_sa_set_temperature: 0 = the reference's accept rule */
	DevCtx ctx;
	uint8_t* d_data;
	uint32_t* d_bucket_off;
	uint32_t* d_bucket_pos;
	uint16_t* d_bucket_nx;
	uint32_t* d_quad_pos;
	uint16_t* d_quad_nx;
	uint32_t *d_quad_rank = nullptr, *d_quad_run = nullptr;
	uint32_t* d_xpos[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };  /* exact-length orders D = 2..7; [0] and [2] alias bucket_pos / quad_pos */
	uint32_t* d_xrank[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
	uint32_t* d_xrun[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
	uint8_t* d_xnxb[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
	uint32_t *d_oct_pos = nullptr, *d_oct_rank = nullptr, *d_oct_run = nullptr;
	uint32_t *d_hex_pos = nullptr, *d_hex_rank = nullptr, *d_hex_run = nullptr;
	uint64_t *d_oct_nx8 = nullptr, *d_hex_nx8 = nullptr;
	uint16_t* d_cost_tbl;
	BaseMem base, scratch;
	mgl_pk* d_best;
	NbrOut nbr;
	uint32_t* d_aos;
	uint64_t* d_cum;
	uint16_t* d_final_probs;
	mgl_pk* d_topk_pk;
	uint64_t* d_topk_cost;
	uint32_t* d_small; /* [0] count */
	uint32_t* d_sub_offs;
	uint32_t* d_sub_lens;
	uint32_t sub_cap;
	uint64_t sqrt_thresh;
	uint32_t per_wave_bytes, waves_per_block, nbr_lds, walk_lds;
	/* incremental path (mgl_kernels2.hip) */
	bool incremental;
	Base2 b2;
	unsigned long long* d_prof; /* 16 u64: per-phase cycles + counts, only with MGL_F_PROFILE */
	uint32_t* d_todo;       /* [0] = count, [1..K] = neighbour indices for the full-walk fallback */
	uint32_t per_wave2, waves_per_block2, nbr2_lds, build_lds;
	uint32_t chg_cap = MGL_CHG_CAP; /* events per first-pass list */
	uint32_t per_wave_pick, per_wave_rest, pick_waves; /* LDS per wavefront of the two halves of the split launch */
	size_t b2_bytes;
	BigScratch big;
	uint32_t* d_todo2;
	uint32_t* d_todo3 = nullptr; /* neighbours k_sim could not take (more touched contexts than its list holds): the late second pass */
	uint4* d_pickstate;     /* 2 K: target, RNG position and walk state at the target (first half -> second half) */
	uint4* d_pickrec;       /* K: picked packet, RNG position, ok flag (first half -> second half of the neighbour evaluation) */
	bool split_nbr, adaptive; /* adaptive: the device recommends the split or the one-kernel form, the host adopts it block by block */
	bool form_single = false; /* the form the regular launch runs as right now */
	uint32_t* d_counts;     /* [0] first-pass overflow count, [1] second-pass overflow count, [2] spill slots used */
	ApplyBuf ab;
	uint32_t apply_blocks;
	bool incremental_apply;
	/* copies of the base structures of the all-literal slab and of the best slab (main.c:71-77) */
	bool snapshots;
	Base2 snap_lit, snap_best;
	SnapMeta* d_snap_meta; /* [0] literal, [1] best */
	bool parallel_build;
	PBuild pb;
	hipEvent_t ev_begin, ev_end;
	std::vector<hipEvent_t> ev_pool;
	/* the step's decision: single (one winner, incremental accept) or bulk (every window-best acceptable
	 * neighbour, parallel rebuild); MGL_ACCEPT_AUTO switches between them block by block */
	int accept_mode = MGL_ACCEPT_AUTO;
	uint32_t bulk_threshold = 0;   /* improving neighbours per step above which a bulk step pays */
	bool bulk_now = true;          /* AUTO: what the next block of steps runs as */
	uint64_t bulk_hold = 0;        /* AUTO: single steps left before bulk steps are tried again (their windows were too long) */
	uint64_t blk_done = 0;         /* AUTO: steps of the current block already run (a block carries over from one mgl_sa_run call to the next) */
	uint64_t blk_imp0 = 0, blk_acc0 = 0; /* AUTO: the device's improving / accepted counters when the current block began */
	BulkBuf bulk;
	BatchBuf batch;               /* a bulk step that took few moves patches the base instead of rebuilding it (mgl_kernels5.hip) */
	bool batch_ok = false;
	uint32_t* d_strat_tgt = nullptr;
	uint32_t* d_strat_pre = nullptr; /* stratified targets: packets before every block of 4 096 positions */
	bool select_small = false;    /* the previous bulk step had few acceptable neighbours: this one's selection runs as one launch */
	uint32_t* h_bstat = nullptr;   /* pinned: the batch accept's status words, read back once per bulk step */
	hipEvent_t ev_bstat = nullptr; /* behind that read-back's copy */
	uint32_t force_batch_fail = 0; /* diagnostic (mgl_debug_set key 5): the next so many batch accepts give up behind their commit */
	uint64_t batch_accepts = 0, batch_fallbacks = 0; /* bulk steps whose moves were patched in / that went to the rebuild although a batch accept began */
	std::vector<uint8_t> mode_log; /* per step of the last mgl_sa_run: 0 single, 1 bulk */
	uint32_t force_rollbacks = 0;  /* diagnostic: treat the next so many bulk steps that took moves as failed validations */
	uint64_t bulk_rollbacks = 0;   /* bulk steps whose combined parse failed validation and was taken back (never seen) */
	bool best_unverified = false;  /* packets_best came from another chain: checked when an epoch starts from it */
	/* look-ahead (k_la_check): the next step's pick + window walk run beside this step's tail into the other buffer set */
	struct NbrSet { NbrOut nbr; uint4* pickrec; uint4* pickstate; uint4* sim_hdr; uint16_t* sim_keys; uint32_t* sim_pos; uint32_t* todo; uint32_t* counts; };
	NbrSet alt = {};
	uint32_t sim_waves = MGL_SIM_WAVES; /* wavefronts per neighbour in the regular re-simulation launch (MGL_SIMW: 1, 2, 4, 8) */
	/* the re-simulation kernel measured on its own (it is the path's dominant kernel): HIP events around its launches on the
	 * streams they run on (MGL_F_TIMING), and -- when switched on, mgl_debug_set key 4 -- the bytes of chain data it reads */
	std::vector<hipEvent_t> ev_sim_pool;
	int64_t time_sim_step = -1;     /* >= 0: the step whose k_sim launches are bracketed by events */
	bool count_traffic = false;
	unsigned long long* d_traffic = nullptr; /* [0] bytes [1] spare */
	bool la_enabled = false, la_ready = false;
	hipGraph_t nbr_graph = nullptr; hipGraphExec_t nbr_graph_exec = nullptr; uint64_t nbr_graph_key = 0; bool graph_ok = false; /* a step's neighbour launches, captured */
	uint32_t short_looks = 0;      /* blocks of at most four steps still to come after a switch of the launch form */
	uint8_t* d_la_mark = nullptr;
	uint32_t* d_la_list = nullptr;
	uint32_t* d_la_hdr = nullptr;  /* [0] neighbours to evaluate again, [1] second-pass entries of the speculative launch */
	hipStream_t stream5 = nullptr, stream6 = nullptr;
	hipEvent_t ev_val = nullptr; /* a bulk step's validation beside its build */
	hipEvent_t ev_la_check = nullptr, ev_la_zero = nullptr, ev_redo = nullptr, ev_spec[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
};

static uint64_t ceil_sqrt_u64(uint64_t x)
{
	uint64_t r = (uint64_t)sqrt((double)x);
	while (r * r > x) r--;
	while ((r + 1) * (r + 1) <= x) r++;
	return r * r == x ? r : r + 1;
}

static void dfree(void* p) { if (p) (void)hipFree(p); }
static void free_base(BaseMem& b)
{
	dfree(b.v.slab); dfree(b.v.onwalk); dfree(b.v.ckpt_probs); dfree(b.v.ckpt_hdr); dfree(b.ctl);
	memset(&b, 0, sizeof b);
}
static hipError_t alloc_base(BaseMem& b, uint32_t n, uint32_t ckpt_elems)
{
	memset(&b, 0, sizeof b);
	hipError_t e;
	b.v.nckpt = (n + (1u << MGL_CKPT_SHIFT) - 1u) >> MGL_CKPT_SHIFT;
	b.v.ckpt_elems = ckpt_elems;
	if ((e = hipMalloc(&b.v.slab, sizeof(mgl_pk) * (size_t)n)) != hipSuccess) return e;
	if ((e = hipMalloc(&b.v.onwalk, sizeof(uint64_t) * (((size_t)n + 63) / 64 + 1))) != hipSuccess) return e;
	if ((e = hipMalloc(&b.v.ckpt_probs, sizeof(uint16_t) * (size_t)b.v.nckpt * ckpt_elems)) != hipSuccess) return e;
	if ((e = hipMalloc(&b.v.ckpt_hdr, sizeof(CkptHdr) * (size_t)b.v.nckpt)) != hipSuccess) return e;
	if ((e = hipMalloc(&b.ctl, sizeof(Control))) != hipSuccess) return e;
	if ((e = hipMemset(b.v.onwalk, 0, sizeof(uint64_t) * (((size_t)n + 63) / 64 + 1))) != hipSuccess) return e;
	return hipMemset(b.ctl, 0, sizeof(Control));
}

static size_t b2_pool_entries(const Base2& b) { return (size_t)b.pool_cap + 256; }
/* device arrays of the incremental base; `own_slab`: also the slab and the on-walk bitmap */
static int alloc_b2(Base2& b, size_t n, bool own_slab, bool init, size_t* bytes_out)
{
	size_t bytes = 0;
	if (own_slab) {
		HIPCHK(hipMalloc(&b.slab, sizeof(mgl_pk) * n)); bytes += sizeof(mgl_pk) * n;
		HIPCHK(hipMalloc(&b.onwalk, sizeof(uint64_t) * ((n + 63) / 64 + 1)));
	}
	HIPCHK(hipMalloc(&b.sp0, sizeof(uint64_t) * (b.nw0 + 64))); bytes += sizeof(uint64_t) * (b.nw0 + 64);
	HIPCHK(hipMalloc(&b.sp1, sizeof(uint64_t) * (b.nw1 + 64)));
	HIPCHK(hipMalloc(&b.sp2, sizeof(uint64_t) * (b.nw2 + 64)));
	HIPCHK(hipMalloc(&b.sp_state, sizeof(uint32_t) * 8 * n)); bytes += sizeof(uint32_t) * 8 * n;
	HIPCHK(hipMalloc(&b.ck_probs, sizeof(uint16_t) * (size_t)b.nck * b.ck_elems)); bytes += sizeof(uint16_t) * (size_t)b.nck * b.ck_elems;
	HIPCHK(hipMalloc(&b.ch_off, sizeof(uint32_t) * b.ck_elems));
	HIPCHK(hipMalloc(&b.ch_len, sizeof(uint32_t) * b.ck_elems));
	HIPCHK(hipMalloc(&b.ch_cap, sizeof(uint32_t) * b.ck_elems));
	HIPCHK(hipMalloc(&b.ch_pos, sizeof(uint32_t) * b2_pool_entries(b))); bytes += sizeof(uint32_t) * (size_t)b.pool_cap;
	HIPCHK(hipMalloc(&b.ch_ev, sizeof(uint16_t) * b2_pool_entries(b))); bytes += sizeof(uint16_t) * (size_t)b.pool_cap;
	HIPCHK(hipMalloc(&b.pool_top, sizeof(uint32_t)));
	HIPCHK(hipMalloc(&b.ch_sb, sizeof(uint32_t) * (size_t)b.ck_elems * b.sb_stride)); bytes += sizeof(uint32_t) * (size_t)b.ck_elems * b.sb_stride;
	if (init) {
		HIPCHK(hipMemset(b.ch_sb, 0, sizeof(uint32_t) * (size_t)b.ck_elems * b.sb_stride));
		HIPCHK(hipMemset(b.sp0, 0, sizeof(uint64_t) * (b.nw0 + 64)));
		HIPCHK(hipMemset(b.sp1, 0, sizeof(uint64_t) * (b.nw1 + 64)));
		HIPCHK(hipMemset(b.sp2, 0, sizeof(uint64_t) * (b.nw2 + 64)));
		HIPCHK(hipMemset(b.ch_pos, 0xFF, sizeof(uint32_t) * b2_pool_entries(b)));
		HIPCHK(hipMemset(b.ch_ev, 0, sizeof(uint16_t) * b2_pool_entries(b)));
	}
	if (bytes_out) *bytes_out = bytes;
	return MGL_OK;
}
static void free_b2(Base2& b, bool own_slab)
{
	if (own_slab) { dfree(b.slab); dfree(b.onwalk); }
	dfree(b.sp0); dfree(b.sp1); dfree(b.sp2); dfree(b.sp_state); dfree(b.ck_probs);
	dfree(b.ch_off); dfree(b.ch_len); dfree(b.ch_cap); dfree(b.ch_pos); dfree(b.ch_ev); dfree(b.pool_top); dfree(b.ch_sb);
	memset(&b, 0, sizeof b);
}
/* dir 0: base -> snapshot, dir 1: snapshot -> base; cond: only if this step found a new best */
static int launch_snapshot(mgl_sa* sa, Base2& snap, uint32_t which, int dir, int cond)
{
	const Base2& from = dir == 0 ? sa->b2 : snap;
	const Base2& to = dir == 0 ? snap : sa->b2;
	const size_t n = sa->n;
	SnapPlan p;
	memset(&p, 0, sizeof p);
	uint32_t k = 0;
	auto seg = [&](const void* s, void* d, size_t bytes, uint32_t pool_elem) {
		p.seg[k].src = s; p.seg[k].dst = d; p.seg[k].bytes = bytes; p.seg[k].pool_elem = pool_elem; k++;
	};
	seg(from.slab, to.slab, sizeof(mgl_pk) * n, 0);
	seg(from.onwalk, to.onwalk, sizeof(uint64_t) * ((n + 63) / 64 + 1), 0);
	seg(from.sp0, to.sp0, sizeof(uint64_t) * (from.nw0 + 64), 0);
	seg(from.sp1, to.sp1, sizeof(uint64_t) * (from.nw1 + 64), 0);
	seg(from.sp2, to.sp2, sizeof(uint64_t) * (from.nw2 + 64), 0);
	seg(from.sp_state, to.sp_state, sizeof(uint32_t) * 8 * n, 0);
	seg(from.ck_probs, to.ck_probs, sizeof(uint16_t) * (size_t)from.nck * from.ck_elems, 0);
	seg(from.ch_off, to.ch_off, sizeof(uint32_t) * from.ck_elems, 0);
	seg(from.ch_len, to.ch_len, sizeof(uint32_t) * from.ck_elems, 0);
	seg(from.ch_cap, to.ch_cap, sizeof(uint32_t) * from.ck_elems, 0);
	seg(from.ch_pos, to.ch_pos, sizeof(uint32_t) * b2_pool_entries(from), 4);
	seg(from.ch_ev, to.ch_ev, sizeof(uint16_t) * b2_pool_entries(from), 2);
	seg(from.pool_top, to.pool_top, sizeof(uint32_t), 0);
	seg(from.ch_sb, to.ch_sb, sizeof(uint32_t) * (size_t)from.ck_elems * from.sb_stride, 0);
	p.nseg = k;
	p.src_pool_top = from.pool_top;
	hipLaunchKernelGGL(k_snapshot, dim3(2048), dim3(256), 0, sa->stream, p, sa->base.ctl, sa->d_snap_meta + which, dir, cond);
	HIPCHK(hipGetLastError());
	return MGL_OK;
}

static int launch_rebuild(mgl_sa* sa, BaseMem& b, int from_dirty, uint64_t* cum, uint16_t* final_probs)
{
	hipLaunchKernelGGL(k_rebuild, dim3(1), dim3(64), sa->walk_lds, sa->stream, sa->ctx, b.v, b.ctl, from_dirty, cum, final_probs);
	HIPCHK(hipGetLastError());
	return MGL_OK;
}
/* (re)derive everything that hangs off the current base slab */
/* the base structures of the slab from scratch, block-parallel (mgl_pbuild.hip) */
static int launch_pbuild(mgl_sa* sa, bool validate_beside = false)
{
	const DevCtx& c = sa->ctx;
	Base2& b = sa->b2;
	PBuild& pb = sa->pb;
	Control* ctl = sa->base.ctl;
	hipStream_t st = sa->stream;
	const uint32_t total = c.L.total;
	hipLaunchKernelGGL(pb_exits, dim3(pb.nblk), dim3(320), 0, st, c, b, pb);
	hipLaunchKernelGGL(pb_entries_group, dim3(pb.ngrp), dim3(320), MGL_PB_GROUP * MGL_PB_ENTRIES * 2u, st, pb);
	hipLaunchKernelGGL(pb_entries, dim3(1), dim3(64), 0, st, c, pb);
	hipLaunchKernelGGL(pb_entries_fill, dim3(pb.ngrp), dim3(64), 0, st, pb);
	hipLaunchKernelGGL(pb_mark, dim3(pb.nblk), dim3(64), 0, st, c, b, pb, ctl);
	{
		const uint32_t nch = (pb.nblk + MGL_PB_SCAN_CHUNK - 1u) / MGL_PB_SCAN_CHUNK;
		hipLaunchKernelGGL(pb_scan_chunks, dim3((nch + 255u) / 256u), dim3(256), 0, st, pb);
		hipLaunchKernelGGL(pb_scan, dim3(1), dim3(64), 0, st, pb);
		hipLaunchKernelGGL(pb_scan_fill, dim3((nch + 255u) / 256u), dim3(256), 0, st, pb);
	}
	hipLaunchKernelGGL(pb_levels, dim3((b.nw0 + 255) / 256), dim3(256), 0, st, (const uint64_t*)b.sp0, b.sp1, b.nw0, b.nw1);
	hipLaunchKernelGGL(pb_levels, dim3((b.nw1 + 255) / 256), dim3(256), 0, st, (const uint64_t*)b.sp1, b.sp2, b.nw1, b.nw2);
	hipLaunchKernelGGL(pb_walk, dim3(pb.nblk), dim3(64), b.ck_elems * 4u, st, c, b, pb);
	if (validate_beside) {
		/* a bulk step's check of the new parse (every packet against the input) needs the bitmaps and the special-state
		 * records, which exist from here on: it runs on the second stream beside the rest of the build */
		HIPCHK(hipEventRecord(sa->ev_fork, st));
		HIPCHK(hipStreamWaitEvent(sa->stream2, sa->ev_fork, 0));
		hipLaunchKernelGGL(k_validate, dim3((sa->ctx.n + 255u) / 256u), dim3(256), 0, sa->stream2, sa->ctx, sa->b2, sa->base.ctl);
		HIPCHK(hipEventRecord(sa->ev_val, sa->stream2));
	}
	{
		const uint32_t og = (pb.nblk + MGL_PB_OFF_ROWS - 1u) / MGL_PB_OFF_ROWS;
		hipLaunchKernelGGL(pb_offsets_sum, dim3((b.ck_elems + 255) / 256, og), dim3(256), 0, st, b, pb);
		hipLaunchKernelGGL(pb_offsets_top, dim3((b.ck_elems + 255) / 256), dim3(256), 0, st, b, pb, total, og);
		hipLaunchKernelGGL(pb_offsets, dim3((b.ck_elems + 255) / 256, og), dim3(256), 0, st, b, pb, total);
		hipLaunchKernelGGL(pb_index, dim3((total + 31u) / 32u, (pb.nblk + 1u + 31u) / 32u), dim3(256), 0, st, b, pb, total);
	}
	hipLaunchKernelGGL(pb_layout, dim3(1), dim3(64), 0, st, b, pb, ctl, total);
	hipLaunchKernelGGL(pb_scatter, dim3(pb.nblk), dim3(256), b.ck_elems * 4u, st, b, pb, total);
	hipLaunchKernelGGL(pb_sim, dim3((pb.seg_cap + 63) / 64), dim3(64), 0, st, c, b, pb);
	hipLaunchKernelGGL(pb_sim_fix, dim3((total + 63) / 64), dim3(64), 0, st, c, b, pb);
	hipLaunchKernelGGL(pb_ckpt, dim3((b.ck_elems + 63) / 64, (b.nck + MGL_PB_CK_ROWS - 1) / MGL_PB_CK_ROWS), dim3(64), 0, st, c, b);
	hipLaunchKernelGGL(pb_finish, dim3(1), dim3(64), 0, st, pb, ctl);
	if (validate_beside) HIPCHK(hipStreamWaitEvent(st, sa->ev_val, 0));
	HIPCHK(hipGetLastError());
	return MGL_OK;
}
static int rebuild_base(mgl_sa* sa, int after_accept)
{
	if (!sa->incremental) return launch_rebuild(sa, sa->base, after_accept, nullptr, nullptr);
	if (sa->parallel_build && !after_accept) return launch_pbuild(sa);
	hipLaunchKernelGGL(k_build, dim3(1), dim3(64), sa->build_lds, sa->stream, sa->ctx, sa->b2, sa->base.ctl, after_accept);
	HIPCHK(hipGetLastError());
	return MGL_OK;
}
/* every packet on the current base's walk reproduces the input (incremental engine: needs the special-state records) */
static int launch_validate(mgl_sa* sa)
{
	if (!sa->incremental) return MGL_OK;
	hipLaunchKernelGGL(k_validate, dim3((sa->ctx.n + 255u) / 256u), dim3(256), 0, sa->stream, sa->ctx, sa->b2, sa->base.ctl);
	HIPCHK(hipGetLastError());
	return MGL_OK;
}
/* incremental engine, after k_decide: fold the winner into the base structures */
static int launch_targets_ahead(mgl_sa* sa, uint64_t next_gstep);
static int launch_apply(mgl_sa* sa, uint64_t next_gstep = ~0ull)
{
	/* the base is about to leave the best slab it still holds: keep a copy first (lazy: while every
	 * accepted step is a new best no copy is ever taken) */
	if (sa->snapshots) {
		int rc = launch_snapshot(sa, sa->snap_best, 1, 0, 2);
		if (rc) return rc;
	}
	hipLaunchKernelGGL(k_apply_walk, dim3(1), dim3(64), 0, sa->stream, sa->ctx, sa->b2, sa->base.ctl, sa->nbr, sa->ab);
	if (next_gstep != ~0ull) { int rc = launch_targets_ahead(sa, next_gstep); if (rc) return rc; } /* the bitmap is final from here on */
	if (!sa->snapshots)
		hipLaunchKernelGGL(k_copy_best, dim3(256), dim3(256), 0, sa->stream, (const Control*)sa->base.ctl,
		                   (const mgl_pk*)sa->base.v.slab, sa->d_best, sa->ctx.n);
	hipLaunchKernelGGL(k_apply_chains, dim3(sa->apply_blocks), dim3(MGL_APPLY_THREADS), 0, sa->stream, sa->ctx, sa->b2, sa->base.ctl, sa->ab);
	hipLaunchKernelGGL(k_apply_jobs, dim3(1024), dim3(256), 0, sa->stream, sa->b2, (const Control*)sa->base.ctl, sa->ab, 0, (const uint32_t*)nullptr);
	hipLaunchKernelGGL(k_apply_jobs, dim3(1024), dim3(256), 0, sa->stream, sa->b2, (const Control*)sa->base.ctl, sa->ab, 1, (const uint32_t*)nullptr);
	hipLaunchKernelGGL(k_build_end, dim3(1), dim3(64), sa->build_lds, sa->stream, sa->ctx, sa->b2, sa->base.ctl, sa->snapshots ? 1 : 0, sa->d_counts, sa->adaptive ? 1 : 0, sa->form_single ? 1 : 0);
	HIPCHK(hipGetLastError());
	return MGL_OK;
}
/* diagnostic (MGL_TRACE=1): name every launch of the neighbour evaluation on stderr and wait for the device after it, so that a
 * faulting kernel is the last one named */
/* diagnostic switches of the launch path, read once */
static const bool g_trace = getenv("MGL_TRACE") != nullptr, g_prof_big = getenv("MGL_PROF_BIG") != nullptr, g_big_inline_sim = getenv("MGL_BIG_INLINE_SIM") != nullptr;
#define NBR_TRACE(name) do { if (g_trace) { fprintf(stderr, "[mgl] %s\n", name); hipError_t e_ = hipDeviceSynchronize(); if (e_ != hipSuccess) fprintf(stderr, "[mgl] %s -> %s\n", name, hipGetErrorString(e_)); } } while (0)
static hipEvent_t sim_event(mgl_sa* sa, size_t i)
{
	while (sa->ev_sim_pool.size() <= i) {
		hipEvent_t e = nullptr;
		if (hipEventCreate(&e) != hipSuccess) return nullptr;
		sa->ev_sim_pool.push_back(e);
	}
	return sa->ev_sim_pool[i];
}
static mgl_sa::NbrSet cur_set(const mgl_sa* sa)
{
	mgl_sa::NbrSet t = { sa->nbr, sa->d_pickrec, sa->d_pickstate, sa->big.sim_hdr, sa->big.sim_keys, sa->big.sim_pos, sa->d_todo, sa->d_counts };
	return t;
}
static void use_set(mgl_sa* sa, const mgl_sa::NbrSet& t)
{
	sa->nbr = t.nbr; sa->d_pickrec = t.pickrec; sa->d_pickstate = t.pickstate;
	sa->big.sim_hdr = t.sim_hdr; sa->big.sim_keys = t.sim_keys; sa->big.sim_pos = t.sim_pos;
	sa->d_todo = t.todo; sa->d_counts = t.counts;
	sa->big.todo_in = t.todo; sa->big.todo_in_count = t.counts; sa->big.spill_ctr = t.counts + 2;
}
static uint32_t nbr_slices(const mgl_sa* sa)
{
	const uint32_t K = sa->cfg.neighbours_per_step;
	return (sa->halves >= 2 && K >= 1024) ? sa->halves : 1u;
}
/* the first two thirds of the split form -- pick, then window walk -- of every slice of a step into the buffer set `t`;
 * slices alternate between the two streams, done[h] is recorded behind slice h's walk */
static int launch_pick_rest(mgl_sa* sa, const mgl_sa::NbrSet& t, uint64_t step_override, hipStream_t s_even, hipStream_t s_odd, hipEvent_t* done)
{
	const uint32_t K = sa->cfg.neighbours_per_step, slices = nbr_slices(sa);
	BigScratch g = sa->big;
	g.sim_hdr = t.sim_hdr; g.sim_keys = t.sim_keys; g.sim_pos = t.sim_pos;
	g.todo_in = t.todo; g.todo_in_count = t.counts; g.spill_ctr = t.counts + 2;
	for (uint32_t h = 0; h < slices; h++) {
		const uint32_t j0 = (uint32_t)((uint64_t)K * h / slices), j1 = (uint32_t)((uint64_t)K * (h + 1) / slices);
		hipStream_t st = (h & 1u) ? s_odd : s_even;
		hipLaunchKernelGGL((k_neighbours2<false, MGL_NBR_PICK>), dim3((j1 - j0 + sa->pick_waves - 1) / sa->pick_waves), dim3(64 * sa->pick_waves),
		                   (MGL_PICK_T_GLOBAL ? 0u : 4096u) + sa->pick_waves * sa->per_wave_pick, st, sa->ctx,
		                   sa->b2, sa->base.ctl, sa->cfg.seed, step_override, K, t.nbr, sa->per_wave_pick, t.todo, t.counts,
		                   (unsigned long long*)nullptr, g, t.pickrec, j0, j1, t.pickstate); NBR_TRACE("k_neighbours2<false, MGL_NBR_PICK>");
		hipLaunchKernelGGL((k_neighbours2<false, MGL_NBR_REST>), dim3(j1 - j0), dim3(64), sa->per_wave_rest, st, sa->ctx,
		                   sa->b2, sa->base.ctl, sa->cfg.seed, step_override, K, t.nbr, sa->per_wave_rest, t.todo, t.counts,
		                   g_prof_big ? (unsigned long long*)nullptr : sa->d_prof, g, t.pickrec, j0, j1, t.pickstate); NBR_TRACE("k_neighbours2<false, MGL_NBR_REST>");
		HIPCHK(hipEventRecord(done[h], st));
	}
	HIPCHK(hipGetLastError());
	return MGL_OK;
}
/* look-ahead: the NEXT step's pick + window walk, speculatively, into the other buffer set; they start once this step's own
 * walks are through (after_check: once k_la_check has run) and the accept waits for them (launch_apply's caller) */
static int launch_lookahead(mgl_sa* sa, uint64_t gstep_next, bool after_check)
{
	const uint32_t slices = nbr_slices(sa);
	if (after_check) HIPCHK(hipStreamWaitEvent(sa->stream5, sa->ev_la_check, 0));
	else for (uint32_t h = 0; h < slices; h++) HIPCHK(hipStreamWaitEvent(sa->stream5, sa->ev_rest[h], 0));
	HIPCHK(hipMemsetAsync(sa->alt.counts, 0, 8 * sizeof(uint32_t), sa->stream5));
	HIPCHK(hipEventRecord(sa->ev_la_zero, sa->stream5));
	HIPCHK(hipStreamWaitEvent(sa->stream6, sa->ev_la_zero, 0));
	int rc = launch_pick_rest(sa, sa->alt, gstep_next, sa->stream5, sa->stream6, sa->ev_spec);
	if (rc) return rc;
	sa->la_ready = true;
	return MGL_OK;
}
/* stratified targets of a step: packets before every block of 4 096 positions of the current walk, their
 * prefix sums, then one wavefront per neighbour turns its ordinal into a position */
static void launch_targets(mgl_sa* sa, hipStream_t st, uint64_t step_override)
{
	const uint32_t K = sa->cfg.neighbours_per_step;
	const uint64_t* onwalk = sa->incremental ? sa->b2.onwalk : sa->base.v.onwalk;
	const uint32_t nw0 = sa->incremental ? sa->b2.nw0 : (sa->ctx.n + 63u) >> 6;
	hipLaunchKernelGGL(k_rank_blocks, dim3(sa->ctx.strat_nblk), dim3(64), 0, st, onwalk, nw0, sa->d_strat_pre, sa->ctx.strat_nblk);
	hipLaunchKernelGGL(k_rank_scan, dim3(1), dim3(1024), 0, st, sa->d_strat_pre, sa->ctx.strat_nblk);
	hipLaunchKernelGGL(k_targets, dim3((K + 3u) / 4u), dim3(256), 0, st, sa->ctx, onwalk, nw0, (const Control*)sa->base.ctl, sa->cfg.seed, step_override, K, sa->d_strat_tgt);
}
/* The next step's targets need the on-walk bitmap of the new base and nothing else: an accept has written it long before it is
 * through with chains and checkpoints, so they are worked out on the second stream beside the rest of the accept (30 us off
 * every step).  Called right behind the launch that completes the bitmap; `next_gstep` is the step they are for. */
static int launch_targets_ahead(mgl_sa* sa, uint64_t next_gstep)
{
	if (!sa->d_strat_pre || !sa->incremental) return MGL_OK;
	HIPCHK(hipEventRecord(sa->ev_tgt_go, sa->stream));
	HIPCHK(hipStreamWaitEvent(sa->stream2, sa->ev_tgt_go, 0));
	launch_targets(sa, sa->stream2, next_gstep);
	HIPCHK(hipEventRecord(sa->ev_tgt, sa->stream2));
	HIPCHK(hipGetLastError());
	sa->tgt_ahead = 1;
	return MGL_OK;
}
static uint64_t graph_key(const mgl_sa* sa);
/* the launches of a step's neighbour evaluation on the incremental engine: pick / walk slices on two streams, re-simulations on a
 * third, second pass, last resort (what launch_neighbours queues behind its targets) */
static int launch_neighbours_body(mgl_sa* sa, uint64_t step_override, bool from_lookahead)
{
	const uint32_t K = sa->cfg.neighbours_per_step;
	const uint32_t blocks2 = (K + sa->waves_per_block2 - 1) / sa->waves_per_block2;
	const bool split_now = sa->split_nbr && !sa->form_single;
	const uint32_t sim_lds_regular = ((((sa->ctx.L.total + 31u) >> 5) + 3u) & ~3u) * 4u + sa->chg_cap * 16u;
	if (split_now) {
		/* the step's neighbours in slices on two streams: while the slowest wavefronts of one slice's kernel finish,
		 * the other slice's kernels keep the CUs busy.  The re-simulation kernels run on a third stream: the second
		 * pass below needs the walks, not the re-simulations, and its few long-running wavefronts overlap with them. */
		const uint32_t slices = nbr_slices(sa);
		if (!from_lookahead) {
			HIPCHK(hipEventRecord(sa->ev_fork, sa->stream));
			if (slices >= 2) HIPCHK(hipStreamWaitEvent(sa->stream2, sa->ev_fork, 0));
			HIPCHK(hipStreamWaitEvent(sa->stream3, sa->ev_fork, 0));
			int rc = launch_pick_rest(sa, cur_set(sa), step_override, sa->stream, sa->stream2, sa->ev_rest);
			if (rc) return rc;
		} else {
			/* pick + walk of this step ran ahead, beside the previous step's tail (launch_lookahead): keep what the accepted
			 * move cannot have touched, evaluate the others again in the one-kernel form (a list, usually short) */
			HIPCHK(hipMemsetAsync(sa->d_la_hdr, 0, 2 * sizeof(uint32_t), sa->stream));
			hipLaunchKernelGGL(k_la_check, dim3((K + 255u) / 256u), dim3(256), 0, sa->stream, sa->ctx, sa->b2, (const Control*)sa->base.ctl, sa->nbr,
			                   (const uint4*)sa->d_pickstate, sa->big.sim_hdr, (const uint32_t*)sa->d_counts, sa->d_la_mark, sa->d_la_list, sa->d_la_hdr, sa->cfg.seed, K); NBR_TRACE("k_la_check");
			HIPCHK(hipEventRecord(sa->ev_la_check, sa->stream));
			HIPCHK(hipStreamWaitEvent(sa->stream3, sa->ev_la_check, 0));
			HIPCHK(hipStreamWaitEvent(sa->stream2, sa->ev_la_check, 0));
		}
		for (uint32_t h = 0; h < slices; h++) {
			const uint32_t j0 = (uint32_t)((uint64_t)K * h / slices), j1 = (uint32_t)((uint64_t)K * (h + 1) / slices);
			/* the second half's re-simulation, several wavefronts per neighbour; a neighbour with more touched contexts
			 * than its list holds goes straight to the last resort's list (the second pass may be running by then) */
			if (!from_lookahead) HIPCHK(hipStreamWaitEvent(sa->stream3, sa->ev_rest[h], 0));
			if (sa->time_sim_step >= 0 && h < 2) HIPCHK(hipEventRecord(sim_event(sa, 6 * (size_t)sa->time_sim_step + 2 * h), sa->stream3));
			if (sa->count_traffic)
				hipLaunchKernelGGL(k_sim<true>, dim3(j1 - j0), dim3(64 * sa->sim_waves), sim_lds_regular, sa->stream3, sa->ctx, sa->b2, sa->base.ctl,
				                   sa->nbr, sa->big, j0, j1, sa->d_todo3, sa->d_counts + 4, (const uint32_t*)nullptr, (const uint32_t*)nullptr, sa->d_traffic);
			else
				hipLaunchKernelGGL(k_sim<false>, dim3(j1 - j0), dim3(64 * sa->sim_waves), sim_lds_regular, sa->stream3, sa->ctx, sa->b2, sa->base.ctl,
				                   sa->nbr, sa->big, j0, j1, sa->d_todo3, sa->d_counts + 4, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (unsigned long long*)nullptr);
			NBR_TRACE("k_sim");
			if (sa->time_sim_step >= 0 && h < 2) HIPCHK(hipEventRecord(sim_event(sa, 6 * (size_t)sa->time_sim_step + 2 * h + 1), sa->stream3));
		}
		HIPCHK(hipEventRecord(sa->ev_sim, sa->stream3));
		if (!from_lookahead) {
			for (uint32_t h = 1; h < slices; h += 2) HIPCHK(hipStreamWaitEvent(sa->stream, sa->ev_rest[h], 0)); /* the walks of the other stream's slices */
		} else {
			BigScratch g = sa->big;
			g.la_list = sa->d_la_list; g.la_count = sa->d_la_hdr;
			const uint32_t grid = blocks2 < 4096u ? blocks2 : 4096u; /* strides over the list */
			/* on the second stream, beside the second pass over the speculative launch's entries; its own entries get a second pass of their own below */
			hipLaunchKernelGGL((k_neighbours2<false, MGL_NBR_FULL, true>), dim3(grid), dim3(64 * sa->waves_per_block2), sa->nbr2_lds, sa->stream2, sa->ctx,
			                   sa->b2, sa->base.ctl, sa->cfg.seed, step_override, K, sa->nbr, sa->per_wave2, sa->d_todo, sa->d_counts,
			                   sa->d_prof, g, sa->d_pickrec, 0u, K, sa->d_pickstate); NBR_TRACE("k_neighbours2<false, MGL_NBR_FULL, true>");
			HIPCHK(hipEventRecord(sa->ev_redo, sa->stream2));
		}
	}
	if (!sa->split_nbr || sa->form_single) { /* the one-kernel form */
		hipLaunchKernelGGL((k_neighbours2<false, MGL_NBR_FULL>), dim3(blocks2), dim3(64 * sa->waves_per_block2), sa->nbr2_lds, sa->stream, sa->ctx,
		                   sa->b2, sa->base.ctl, sa->cfg.seed, step_override, K, sa->nbr, sa->per_wave2, sa->d_todo, sa->d_counts,
		                   sa->d_prof, sa->big, sa->d_pickrec, 0u, K, sa->d_pickstate); NBR_TRACE("k_neighbours2<false, MGL_NBR_FULL>");
	}
	/* the few whose change lists overflowed LDS (or that need a second top-K pick): the whole
	 * evaluation in one kernel, lists in global scratch */
	const uint32_t bigneed = (sa->big.slots + sa->waves_per_block2 - 1) / sa->waves_per_block2; /* a slot per neighbour: none is dropped */
	const uint32_t bigblocks = bigneed < 1024u ? bigneed : 1024u; /* the kernel strides over its list */
	BigScratch big_now = sa->big;
	big_now.lds_cache = 1u;
	const uint32_t big_lds = sa->nbr2_lds + 12u * MGL_BIG_CAP; /* + a copy of both lists for the re-simulations */
	if (!split_now || g_big_inline_sim) big_now.sim_hdr2 = nullptr; /* the one-kernel form has no k_sim launch to hand over to */
	const bool la_step = from_lookahead && split_now;
	if (la_step) {
		/* the list-mode instances dereference these unconditionally or behind a test of their partner pointer only (a launch
		 * of k_neighbours2<true, FULL, true> once faulted on a null address, cause never pinned down: DESIGN.md section 10) */
		if (!sa->d_la_mark || !sa->d_la_hdr || !sa->d_la_list || !sa->d_todo || !sa->d_todo2 || !sa->d_pickrec || !sa->d_pickstate || !sa->big.ins_key || !sa->big.sim_slot2)
			return fail(MGL_EDEVICE, "launch_neighbours: look-ahead step without its buffers");
		/* first the entries the speculative launch made (their count is fixed since k_la_check; the ones evaluated again are passed by) ... */
		big_now.la_mark = sa->d_la_mark; big_now.la_spec_count = sa->d_la_hdr + 1;
		big_now.todo_in_count = sa->d_la_hdr + 1;
	}
	if (la_step) {
		hipLaunchKernelGGL((k_neighbours2<true, MGL_NBR_FULL, true>), dim3(bigblocks), dim3(64 * sa->waves_per_block2), big_lds, sa->stream, sa->ctx,
		                   sa->b2, sa->base.ctl, sa->cfg.seed, step_override, K, sa->nbr, sa->per_wave2, sa->d_todo2, sa->d_counts + 1,
		                   (unsigned long long*)nullptr, big_now, sa->d_pickrec, 0u, K, sa->d_pickstate); NBR_TRACE("k_neighbours2<true, MGL_NBR_FULL, true>");
	} else {
		hipLaunchKernelGGL((k_neighbours2<true, MGL_NBR_FULL>), dim3(bigblocks), dim3(64 * sa->waves_per_block2), big_lds, sa->stream, sa->ctx,
		                   sa->b2, sa->base.ctl, sa->cfg.seed, step_override, K, sa->nbr, sa->per_wave2, sa->d_todo2, sa->d_counts + 1,
		                   g_prof_big ? sa->d_prof : (unsigned long long*)nullptr, big_now, (sa->split_nbr && !sa->form_single) ? sa->d_pickrec : (uint4*)nullptr, 0u, K, sa->d_pickstate); NBR_TRACE("k_neighbours2<true, MGL_NBR_FULL>");
	}
	if (la_step) {
		/* ... then, once the fresh evaluations are through, the entries they added */
		HIPCHK(hipStreamWaitEvent(sa->stream, sa->ev_redo, 0));
		big_now.todo_in_count = sa->d_counts;
		big_now.todo_first = sa->d_la_hdr + 1;
		hipLaunchKernelGGL((k_neighbours2<true, MGL_NBR_FULL, true>), dim3(bigblocks < 256u ? bigblocks : 256u), dim3(64 * sa->waves_per_block2), big_lds, sa->stream, sa->ctx,
		                   sa->b2, sa->base.ctl, sa->cfg.seed, step_override, K, sa->nbr, sa->per_wave2, sa->d_todo2, sa->d_counts + 1,
		                   (unsigned long long*)nullptr, big_now, sa->d_pickrec, 0u, K, sa->d_pickstate); NBR_TRACE("k_neighbours2<true, MGL_NBR_FULL, true>");
		big_now.todo_first = nullptr;
	}
	if (split_now) {
		/* the second pass handed its final re-simulations to k_sim as well (headers in sim_hdr2): a small grid over its list */
		const uint32_t sim_lds = ((((sa->ctx.L.total + 31u) >> 5) + 3u) & ~3u) * 4u + MGL_SIM2_CAP * 16u;
		if (sa->time_sim_step >= 0) HIPCHK(hipEventRecord(sim_event(sa, 6 * (size_t)sa->time_sim_step + 4), sa->stream));
		if (sa->count_traffic)
			hipLaunchKernelGGL(k_sim<true>, dim3(256), dim3(64 * MGL_SIM_WAVES_LIST), sim_lds, sa->stream, sa->ctx, sa->b2, sa->base.ctl, sa->nbr, big_now, 0u, K,
			                   sa->d_todo3, sa->d_counts + 4, (const uint32_t*)sa->d_todo, (const uint32_t*)sa->d_counts, sa->d_traffic);
		else
			hipLaunchKernelGGL(k_sim<false>, dim3(256), dim3(64 * MGL_SIM_WAVES_LIST), sim_lds, sa->stream, sa->ctx, sa->b2, sa->base.ctl, sa->nbr, big_now, 0u, K,
			                   sa->d_todo3, sa->d_counts + 4, (const uint32_t*)sa->d_todo, (const uint32_t*)sa->d_counts, (unsigned long long*)nullptr);
		NBR_TRACE("k_sim");
		if (sa->time_sim_step >= 0) HIPCHK(hipEventRecord(sim_event(sa, 6 * (size_t)sa->time_sim_step + 5), sa->stream));
		/* what k_sim (either launch) could not take: a late second pass that re-simulates inline */
		HIPCHK(hipStreamWaitEvent(sa->stream, sa->ev_sim, 0));
		BigScratch late = sa->big;
		late.todo_in = sa->d_todo3; late.todo_in_count = sa->d_counts + 4; late.sim_hdr2 = nullptr; late.lds_cache = 1u;
		late.cont = nullptr; /* its list is another one: the slots' saved walks belong to the second pass proper */
		hipLaunchKernelGGL((k_neighbours2<true, MGL_NBR_FULL>), dim3(bigblocks < 64u ? bigblocks : 64u), dim3(64 * sa->waves_per_block2), big_lds, sa->stream, sa->ctx,
		                   sa->b2, sa->base.ctl, sa->cfg.seed, step_override, K, sa->nbr, sa->per_wave2, sa->d_todo2, sa->d_counts + 1,
		                   (unsigned long long*)nullptr, late, sa->d_pickrec, 0u, K, sa->d_pickstate); NBR_TRACE("k_neighbours2<true, MGL_NBR_FULL>");
	}
	/* and whatever overflowed even that: exact full walk from byte 0 (a small grid strides over the list) */
	const uint32_t blocks = (K + sa->waves_per_block - 1) / sa->waves_per_block;
	hipLaunchKernelGGL(k_neighbours, dim3(blocks < 256u ? blocks : 256u), dim3(64 * sa->waves_per_block), sa->nbr_lds, sa->stream, sa->ctx, sa->base.v,
	                   (const Control*)sa->base.ctl, sa->cfg.seed, step_override, K, sa->nbr, sa->per_wave_bytes,
	                   (const uint32_t*)sa->d_todo2, (const uint32_t*)(sa->d_counts + 1)); NBR_TRACE("k_neighbours");
	HIPCHK(hipGetLastError());
	return MGL_OK;
}

static int launch_neighbours(mgl_sa* sa, uint64_t step_override, bool zero_counts = true, bool from_lookahead = false)
{
	const uint32_t K = sa->cfg.neighbours_per_step;
	if (sa->d_strat_pre) {
		if (sa->tgt_ahead) HIPCHK(hipStreamWaitEvent(sa->stream, sa->ev_tgt, 0)); /* made beside the previous step's accept (or at least out of the buffers' way) */
		if (sa->tgt_ahead != 1) launch_targets(sa, sa->stream, step_override);
		sa->tgt_ahead = 0;
	}
	if (!sa->incremental) {
		const uint32_t blocks = (K + sa->waves_per_block - 1) / sa->waves_per_block;
		hipLaunchKernelGGL(k_neighbours, dim3(blocks), dim3(64 * sa->waves_per_block), sa->nbr_lds, sa->stream, sa->ctx, sa->base.v,
		                   (const Control*)sa->base.ctl, sa->cfg.seed, step_override, K, sa->nbr, sa->per_wave_bytes,
		                   (const uint32_t*)nullptr, (const uint32_t*)nullptr); NBR_TRACE("k_neighbours");
		HIPCHK(hipGetLastError());
		return MGL_OK;
	}
	if (zero_counts) HIPCHK(hipMemsetAsync(sa->d_counts, 0, 8 * sizeof(uint32_t), sa->stream)); /* todo counts + spill slots; k_step_end clears them between steps */
	/* The launches of a step are the same from step to step (the kernels read the step's number from the control block): captured
	 * once as a graph -- three streams, their events -- and replayed, unless something about them changed (the launch form, a
	 * buffer, a diagnostic) or the step is a timed or traced one. */
	if (sa->graph_ok && step_override == ~0ull && !from_lookahead && !sa->la_enabled && sa->time_sim_step < 0 && !sa->count_traffic && !g_trace && !g_prof_big && sa->d_prof == nullptr) {
		const uint64_t key = graph_key(sa);
		if (sa->nbr_graph_exec == nullptr || key != sa->nbr_graph_key) {
			if (sa->nbr_graph_exec) { (void)hipGraphExecDestroy(sa->nbr_graph_exec); sa->nbr_graph_exec = nullptr; }
			if (sa->nbr_graph) { (void)hipGraphDestroy(sa->nbr_graph); sa->nbr_graph = nullptr; }
			HIPCHK(hipStreamBeginCapture(sa->stream, hipStreamCaptureModeRelaxed));
			const int rc = launch_neighbours_body(sa, ~0ull, false);
			const hipError_t ec = hipStreamEndCapture(sa->stream, &sa->nbr_graph);
			if (rc) return rc;
			if (ec != hipSuccess || sa->nbr_graph == nullptr) return fail(MGL_EDEVICE, "launch_neighbours: graph capture failed");
			HIPCHK(hipGraphInstantiate(&sa->nbr_graph_exec, sa->nbr_graph, nullptr, nullptr, 0));
			sa->nbr_graph_key = key;
		}
		HIPCHK(hipGraphLaunch(sa->nbr_graph_exec, sa->stream));
		return MGL_OK;
	}
	return launch_neighbours_body(sa, step_override, from_lookahead);
}

/* everything a step's neighbour launches are made of: a captured graph is good while this stays what it was */
static uint64_t graph_key(const mgl_sa* sa)
{
	uint64_t h = 1469598103934665603ull;
	auto mix = [&](const void* p, size_t n) { const unsigned char* c = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 1099511628211ull; } };
	mix(&sa->ctx, sizeof sa->ctx); mix(&sa->b2, sizeof sa->b2); mix(&sa->base, sizeof sa->base); mix(&sa->nbr, sizeof sa->nbr); mix(&sa->big, sizeof sa->big);
	const uint64_t w[] = { sa->cfg.neighbours_per_step, sa->cfg.seed, (uint64_t)sa->form_single, (uint64_t)sa->split_nbr, sa->halves, sa->waves_per_block, sa->waves_per_block2,
	                       sa->pick_waves, sa->per_wave_pick, sa->per_wave_rest, sa->per_wave2, sa->per_wave_bytes, sa->nbr_lds, sa->nbr2_lds, sa->chg_cap, sa->sim_waves,
	                       (uint64_t)(uintptr_t)sa->d_todo, (uint64_t)(uintptr_t)sa->d_todo2, (uint64_t)(uintptr_t)sa->d_todo3, (uint64_t)(uintptr_t)sa->d_counts,
	                       (uint64_t)(uintptr_t)sa->d_pickrec, (uint64_t)(uintptr_t)sa->d_pickstate, (uint64_t)g_big_inline_sim };
	mix(w, sizeof w);
	return h;
}

static int import_slab(mgl_sa* sa, const mgl_packet* packets, mgl_pk* d_slab)
{
	HIPCHK(hipMemcpyAsync(sa->d_aos, packets, sizeof(mgl_packet) * (size_t)sa->n, hipMemcpyHostToDevice, sa->stream));
	hipLaunchKernelGGL(k_import, dim3(1024), dim3(256), 0, sa->stream, (const uint32_t*)sa->d_aos, d_slab, sa->n);
	HIPCHK(hipGetLastError());
	return MGL_OK;
}
static int export_slab(mgl_sa* sa, const mgl_pk* d_slab, mgl_packet* packets)
{
	hipLaunchKernelGGL(k_export, dim3(1024), dim3(256), 0, sa->stream, d_slab, sa->d_aos, sa->n);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(packets, sa->d_aos, sizeof(mgl_packet) * (size_t)sa->n, hipMemcpyDeviceToHost, sa->stream));
	HIPCHK(hipStreamSynchronize(sa->stream));
	return MGL_OK;
}

extern "C" void mgl_sa_destroy(mgl_sa* sa)
{
	if (!sa) return;
	(void)hipSetDevice(sa->device);
	if (sa->stream) (void)hipStreamSynchronize(sa->stream);
	dfree(sa->d_data); dfree(sa->d_bucket_off); dfree(sa->d_bucket_pos); dfree(sa->d_bucket_nx); dfree(sa->d_quad_pos); dfree(sa->d_quad_nx); dfree(sa->d_cost_tbl);
	dfree(sa->d_quad_rank); dfree(sa->d_quad_run); dfree(sa->d_oct_pos);
	for (int i = 0; i < 6; i++) { if (i != 0 && i != 2) { dfree(sa->d_xpos[i]); dfree(sa->d_xrank[i]); dfree(sa->d_xrun[i]); } dfree(sa->d_xnxb[i]); } dfree(sa->d_oct_rank); dfree(sa->d_oct_run);
	dfree(sa->d_hex_pos); dfree(sa->d_hex_rank); dfree(sa->d_hex_run); dfree(sa->d_oct_nx8); dfree(sa->d_hex_nx8);
	free_base(sa->base); free_base(sa->scratch);
	if (!sa->snapshots) dfree(sa->d_best); /* otherwise it is the best snapshot's slab */
	dfree(sa->nbr.cost); dfree(sa->nbr.ndiffs); dfree(sa->nbr.walked); dfree(sa->nbr.win); dfree(sa->nbr.win2); dfree(sa->nbr.dpos);
	dfree(sa->bulk.ckey); dfree(sa->bulk.cwin); dfree(sa->bulk.taken); dfree(sa->bulk.cstate); dfree(sa->bulk.cflags); dfree(sa->bulk.hdr);
	dfree(sa->nbr.dold); dfree(sa->nbr.dnew);
	dfree(sa->d_aos); dfree(sa->d_cum); dfree(sa->d_final_probs);
	free_b2(sa->b2, false);
	free_b2(sa->snap_lit, true); free_b2(sa->snap_best, true); dfree(sa->d_snap_meta);
	dfree(sa->pb.exits); dfree(sa->pb.entry); dfree(sa->pb.gexits); dfree(sa->pb.gentry); dfree(sa->pb.gsum); dfree(sa->pb.ch_map); dfree(sa->pb.ch_vs); dfree(sa->pb.ch_pk); dfree(sa->pb.ch_state); dfree(sa->pb.tf_ctx); dfree(sa->pb.tf_dist); dfree(sa->pb.tf_pk);
	dfree(sa->pb.st_in); dfree(sa->pb.hist); dfree(sa->pb.stage); dfree(sa->pb.stage_n); dfree(sa->pb.stage_over); dfree(sa->pb.acc); dfree(sa->pb.seg_off); dfree(sa->pb.unres);
	{
		mgl_sa::NbrSet& t = sa->alt;
		dfree(t.nbr.cost); dfree(t.nbr.ndiffs); dfree(t.nbr.walked); dfree(t.nbr.win); dfree(t.nbr.win2); dfree(t.nbr.dpos); dfree(t.nbr.dold); dfree(t.nbr.dnew);
		dfree(t.pickrec); dfree(t.pickstate); dfree(t.sim_hdr); dfree(t.sim_keys); dfree(t.sim_pos); dfree(t.todo); dfree(t.counts);
		dfree(sa->d_la_mark); dfree(sa->d_la_list); dfree(sa->d_la_hdr);
		if (sa->stream5) (void)hipStreamDestroy(sa->stream5);
		if (sa->stream6) (void)hipStreamDestroy(sa->stream6);
		if (sa->ev_la_check) (void)hipEventDestroy(sa->ev_la_check);
		if (sa->ev_la_zero) (void)hipEventDestroy(sa->ev_la_zero);
		if (sa->ev_redo) (void)hipEventDestroy(sa->ev_redo);
		for (int i = 0; i < 8; i++) if (sa->ev_spec[i]) (void)hipEventDestroy(sa->ev_spec[i]);
	}
	dfree(sa->d_todo); dfree(sa->d_prof);
	dfree(sa->big.sim_hdr); dfree(sa->big.sim_keys); dfree(sa->big.sim_pos);
	dfree(sa->big.ins_key); dfree(sa->big.rem_key); dfree(sa->big.ins_pos); dfree(sa->big.rem_pos); dfree(sa->big.uctx);
	if (sa->h_bstat) (void)hipHostFree(sa->h_bstat);
	if (sa->ev_bstat) (void)hipEventDestroy(sa->ev_bstat);
	dfree(sa->big.cont); dfree(sa->d_traffic); dfree(sa->d_strat_pre); dfree(sa->d_strat_tgt);
	{
		BatchBuf& bt = sa->batch;
		dfree(bt.hdr); dfree(bt.cl); dfree(bt.jpos); dfree(bt.jnew); dfree(bt.jold); dfree(bt.st_ikey); dfree(bt.st_rkey); dfree(bt.st_ipos);
		dfree(bt.st_rpos); dfree(bt.ops); dfree(bt.ins_cl); dfree(bt.rem_cl); dfree(bt.acc); dfree(bt.runs);
		dfree(bt.cnt_i); dfree(bt.cnt_r); dfree(bt.off_i); dfree(bt.off_r); dfree(bt.cur_i); dfree(bt.cur_r); dfree(bt.touched);
		dfree(bt.bk_ipos); dfree(bt.bk_rpos); dfree(bt.bk_ibit); dfree(bt.bk_icl); dfree(bt.bk_rcl);
	}
	for (hipEvent_t e : sa->ev_sim_pool) if (e) (void)hipEventDestroy(e);
	dfree(sa->d_todo2); dfree(sa->d_todo3); dfree(sa->d_counts); dfree(sa->big.sim_hdr2); dfree(sa->big.sim_slot2); dfree(sa->d_pickrec); dfree(sa->d_pickstate);
	dfree(sa->ab.hdr); dfree(sa->ab.ins_key); dfree(sa->ab.rem_key); dfree(sa->ab.ins_pos); dfree(sa->ab.rem_pos);
	dfree(sa->ab.tctx); dfree(sa->ab.scratch_pos); dfree(sa->ab.scratch_ev);
	dfree(sa->ab.span_pos); dfree(sa->ab.span_ev); dfree(sa->ab.jobs_b); dfree(sa->ab.jobs_c);
	dfree(sa->d_topk_pk); dfree(sa->d_topk_cost); dfree(sa->d_small); dfree(sa->d_sub_offs); dfree(sa->d_sub_lens);
	for (hipEvent_t e : sa->ev_pool) (void)hipEventDestroy(e);
	if (sa->ev_begin) (void)hipEventDestroy(sa->ev_begin);
	if (sa->ev_end) (void)hipEventDestroy(sa->ev_end);
	if (sa->stream2) (void)hipStreamDestroy(sa->stream2);
	if (sa->stream3) (void)hipStreamDestroy(sa->stream3);
	for (auto& e : sa->ev_rest) if (e) (void)hipEventDestroy(e);
	if (sa->ev_sim) (void)hipEventDestroy(sa->ev_sim);
	if (sa->ev_val) (void)hipEventDestroy(sa->ev_val);
	if (sa->nbr_graph_exec) (void)hipGraphExecDestroy(sa->nbr_graph_exec);
	if (sa->nbr_graph) (void)hipGraphDestroy(sa->nbr_graph);
	if (sa->ev_fork) (void)hipEventDestroy(sa->ev_fork);
	if (sa->ev_tgt) (void)hipEventDestroy(sa->ev_tgt);
	if (sa->ev_tgt_go) (void)hipEventDestroy(sa->ev_tgt_go);
	if (sa->ev_join) (void)hipEventDestroy(sa->ev_join);
	if (sa->stream) (void)hipStreamDestroy(sa->stream);
	delete sa;
}

static int create_impl(mgl_sa* sa, const uint8_t* data, size_t n)
{
	HIPCHK(hipSetDevice(sa->device));
	HIPCHK(hipStreamCreate(&sa->stream));
	HIPCHK(hipStreamCreate(&sa->stream2));
	HIPCHK(hipEventCreateWithFlags(&sa->ev_fork, hipEventDisableTiming));
	HIPCHK(hipEventCreateWithFlags(&sa->ev_join, hipEventDisableTiming));
	HIPCHK(hipEventCreateWithFlags(&sa->ev_tgt, hipEventDisableTiming));
	HIPCHK(hipEventCreateWithFlags(&sa->ev_tgt_go, hipEventDisableTiming));
	HIPCHK(hipStreamCreate(&sa->stream3));
	for (auto& e : sa->ev_rest) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	HIPCHK(hipEventCreateWithFlags(&sa->ev_sim, hipEventDisableTiming));
	HIPCHK(hipEventCreateWithFlags(&sa->ev_val, hipEventDisableTiming));
	/* measured: + 8 % on the 10 MB input, nothing on the 100 KB one (its kernels are too short to overlap) */
	sa->graph_ok = getenv("MGL_GRAPH") != nullptr;
	sa->halves = getenv("MGL_HALVES") ? (uint32_t)atoi(getenv("MGL_HALVES")) : (n > (1u << 20) ? 2u : 1u);
	if (sa->halves < 1 || sa->halves > 8) sa->halves = 1;
	HIPCHK(hipEventCreate(&sa->ev_begin));
	HIPCHK(hipEventCreate(&sa->ev_end));
	const mgl_layout L = mgl_make_layout(sa->props.lc, sa->props.lp, sa->props.pb);
	const uint32_t ckpt_elems = (L.total + 7u) & ~7u;

	/* input, zero padded so that window loads past the end stay in bounds */
	HIPCHK(hipMalloc(&sa->d_data, n + 128));
	HIPCHK(hipMemset(sa->d_data, 0, n + 128));
	HIPCHK(hipMemcpy(sa->d_data, data, n, hipMemcpyHostToDevice));

	/* match index (substring_enumerator.c:26-47): positions bucketed by leading bigram, ascending
	 * inside a bucket -- a stable two-pass counting sort on the device (mgl_index.hip) */
	{
		const uint32_t m = (uint32_t)(n - 1); /* positions that start a bigram */
		HIPCHK(hipMalloc(&sa->d_bucket_off, sizeof(uint32_t) * 65537));
		HIPCHK(hipMalloc(&sa->d_bucket_pos, sizeof(uint32_t) * (n ? n : 1)));
		HIPCHK(hipMalloc(&sa->d_bucket_nx, sizeof(uint16_t) * (n ? n : 1)));
		HIPCHK(hipMalloc(&sa->d_quad_pos, sizeof(uint32_t) * (n ? n : 1)));
		HIPCHK(hipMalloc(&sa->d_quad_nx, sizeof(uint16_t) * (n ? n : 1)));
		for (int i = 0; i < 6; i++) {
			HIPCHK(hipMalloc(&sa->d_xnxb[i], n + 1));
			HIPCHK(hipMemset(sa->d_xnxb[i], 0, n + 1));
			if (i == 0 || i == 2) continue;
			for (uint32_t** a : { &sa->d_xpos[i], &sa->d_xrank[i], &sa->d_xrun[i] }) {
				HIPCHK(hipMalloc(a, sizeof(uint32_t) * (n + 1)));
				HIPCHK(hipMemset(*a, 0, sizeof(uint32_t) * (n + 1)));
			}
		}
		for (uint32_t** a : { &sa->d_quad_rank, &sa->d_quad_run, &sa->d_oct_pos, &sa->d_oct_rank, &sa->d_oct_run, &sa->d_hex_pos, &sa->d_hex_rank, &sa->d_hex_run }) {
			HIPCHK(hipMalloc(a, sizeof(uint32_t) * (n + 1)));
			HIPCHK(hipMemset(*a, 0, sizeof(uint32_t) * (n + 1)));
		}
		HIPCHK(hipMalloc(&sa->d_oct_nx8, sizeof(uint64_t) * (n + 1)));
		HIPCHK(hipMalloc(&sa->d_hex_nx8, sizeof(uint64_t) * (n + 1)));
		if (m == 0) HIPCHK(hipMemset(sa->d_bucket_off, 0, sizeof(uint32_t) * 65537));
		else {
			const uint32_t nblk = (m + MGL_IX_ITEMS - 1) / MGL_IX_ITEMS;
			uint32_t *tmp = nullptr, *matrix = nullptr;
			HIPCHK(hipMalloc(&tmp, sizeof(uint32_t) * m));
			HIPCHK(hipMalloc(&matrix, sizeof(uint32_t) * 256 * (size_t)nblk));
			/* bigram order: by data[p + 1], then data[p] */
			for (int pass = 1; pass >= 0; pass--) {
				const uint32_t* in = pass == 1 ? nullptr : tmp;
				uint32_t* out = pass == 1 ? tmp : sa->d_bucket_pos;
				hipLaunchKernelGGL(ix_count, dim3(nblk), dim3(64), 0, sa->stream, (const uint8_t*)sa->d_data, in, m, pass, matrix, nblk);
				hipLaunchKernelGGL(ix_scan, dim3(1), dim3(1024), 0, sa->stream, matrix, 256u * nblk);
				hipLaunchKernelGGL(ix_scatter, dim3(nblk), dim3(64), 0, sa->stream, (const uint8_t*)sa->d_data, in, out, m, pass,
				                   (const uint32_t*)matrix, nblk);
			}
			hipLaunchKernelGGL(ix_offsets, dim3(m / 256 + 1), dim3(256), 0, sa->stream, (const uint8_t*)sa->d_data,
			                   (const uint32_t*)sa->d_bucket_pos, m, sa->d_bucket_off);
			hipLaunchKernelGGL(ix_next2, dim3(m / 256 + 1), dim3(256), 0, sa->stream, (const uint8_t*)sa->d_data,
			                   (const uint32_t*)sa->d_bucket_pos, m, sa->d_bucket_nx, 0);
			/* four-byte order: by data[p + 3], data[p + 2], data[p + 1], data[p] (ping-pong tmp <-> quad_pos) */
			for (int pass = 3; pass >= 0; pass--) {
				const uint32_t* in = pass == 3 ? nullptr : ((pass & 1) ? (const uint32_t*)sa->d_quad_pos : (const uint32_t*)tmp);
				uint32_t* out = (pass & 1) ? tmp : sa->d_quad_pos;
				hipLaunchKernelGGL(ix_count, dim3(nblk), dim3(64), 0, sa->stream, (const uint8_t*)sa->d_data, in, m, pass, matrix, nblk);
				hipLaunchKernelGGL(ix_scan, dim3(1), dim3(1024), 0, sa->stream, matrix, 256u * nblk);
				hipLaunchKernelGGL(ix_scatter, dim3(nblk), dim3(64), 0, sa->stream, (const uint8_t*)sa->d_data, in, out, m, pass,
				                   (const uint32_t*)matrix, nblk);
			}
			hipLaunchKernelGGL(ix_next2, dim3(m / 256 + 1), dim3(256), 0, sa->stream, (const uint8_t*)sa->d_data,
			                   (const uint32_t*)sa->d_quad_pos, m, sa->d_quad_nx, 1);
			/* the eight- and sixteen-byte orders, and for all three deeper orders: rank, run starts, next bytes */
			sa->d_xpos[0] = sa->d_bucket_pos; sa->d_xpos[2] = sa->d_quad_pos;
			sa->d_xrank[2] = sa->d_quad_rank; sa->d_xrun[2] = sa->d_quad_run;
			for (uint32_t D : { 3u, 5u, 6u, 7u, 8u, 16u }) {
				uint32_t* dst = D == 8 ? sa->d_oct_pos : D == 16 ? sa->d_hex_pos : sa->d_xpos[D - 2];
				for (int pass = (int)D - 1; pass >= 0; pass--) {
					/* ping-pong so that the last pass (byte 0) lands in dst */
					const uint32_t* in = pass == (int)D - 1 ? nullptr : ((pass & 1) ? (const uint32_t*)dst : (const uint32_t*)tmp);
					uint32_t* out = (pass & 1) ? tmp : dst;
					hipLaunchKernelGGL(ix_count, dim3(nblk), dim3(64), 0, sa->stream, (const uint8_t*)sa->d_data, in, m, pass, matrix, nblk);
					hipLaunchKernelGGL(ix_scan, dim3(1), dim3(1024), 0, sa->stream, matrix, 256u * nblk);
					hipLaunchKernelGGL(ix_scatter, dim3(nblk), dim3(64), 0, sa->stream, (const uint8_t*)sa->d_data, in, out, m, pass,
					                   (const uint32_t*)matrix, nblk);
				}
			}
			struct { const uint32_t* pos; uint32_t* rank; uint32_t* run; void* nx; uint32_t D, nxbytes; } lv[7] = {
				{ sa->d_xpos[1], sa->d_xrank[1], sa->d_xrun[1], sa->d_xnxb[1], 3u, 1u },
				{ sa->d_quad_pos, sa->d_quad_rank, sa->d_quad_run, sa->d_xnxb[2], 4u, 1u },
				{ sa->d_xpos[3], sa->d_xrank[3], sa->d_xrun[3], sa->d_xnxb[3], 5u, 1u },
				{ sa->d_xpos[4], sa->d_xrank[4], sa->d_xrun[4], sa->d_xnxb[4], 6u, 1u },
				{ sa->d_xpos[5], sa->d_xrank[5], sa->d_xrun[5], sa->d_xnxb[5], 7u, 1u },
				{ sa->d_oct_pos, sa->d_oct_rank, sa->d_oct_run, sa->d_oct_nx8, 8u, 8u },
				{ sa->d_hex_pos, sa->d_hex_rank, sa->d_hex_run, sa->d_hex_nx8, 16u, 8u } };
			hipLaunchKernelGGL(ix_next_byte, dim3(m / 256 + 1), dim3(256), 0, sa->stream, (const uint8_t*)sa->d_data, (const uint32_t*)sa->d_bucket_pos, m, sa->d_xnxb[0], 2u);
			for (auto& l : lv) {
				hipLaunchKernelGGL(ix_rank, dim3(m / 256 + 1), dim3(256), 0, sa->stream, l.pos, m, l.rank);
				if (l.nxbytes == 1) hipLaunchKernelGGL(ix_next_byte, dim3(m / 256 + 1), dim3(256), 0, sa->stream, (const uint8_t*)sa->d_data, l.pos, m, (uint8_t*)l.nx, l.D);
				else hipLaunchKernelGGL(ix_next_bytes, dim3(m / 256 + 1), dim3(256), 0, sa->stream, (const uint8_t*)sa->d_data, l.pos, m, l.nx, l.D, l.nxbytes);
				hipLaunchKernelGGL(ix_heads, dim3(m / 256 + 1), dim3(256), 0, sa->stream, (const uint8_t*)sa->d_data, l.pos, m, l.D, l.run);
				hipLaunchKernelGGL(ix_maxscan, dim3(1), dim3(1024), 0, sa->stream, l.run, m);
			}
			hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(sa->stream);
			(void)hipFree(tmp); (void)hipFree(matrix);
			HIPCHK(e1); HIPCHK(e2);
		}
	}
	HIPCHK(hipMalloc(&sa->d_cost_tbl, sizeof(k_cost_table)));
	HIPCHK(hipMemcpy(sa->d_cost_tbl, k_cost_table, sizeof(k_cost_table), hipMemcpyHostToDevice));

	sa->ctx.data = sa->d_data; sa->ctx.n = (uint32_t)n;
	sa->ctx.bucket_off = sa->d_bucket_off; sa->ctx.bucket_pos = sa->d_bucket_pos; sa->ctx.bucket_nx = sa->d_bucket_nx; sa->ctx.quad_pos = sa->d_quad_pos; sa->ctx.quad_nx = sa->d_quad_nx;
	for (int i = 0; i < 6; i++) { sa->ctx.xpos[i] = sa->d_xpos[i]; sa->ctx.xrank[i] = sa->d_xrank[i]; sa->ctx.xrun[i] = sa->d_xrun[i]; sa->ctx.xnxb[i] = sa->d_xnxb[i]; }
	sa->ctx.oct_pos = sa->d_oct_pos; sa->ctx.oct_rank = sa->d_oct_rank; sa->ctx.oct_run = sa->d_oct_run; sa->ctx.oct_nx8 = sa->d_oct_nx8;
	sa->ctx.hex_pos = sa->d_hex_pos; sa->ctx.hex_rank = sa->d_hex_rank; sa->ctx.hex_run = sa->d_hex_run; sa->ctx.hex_nx8 = sa->d_hex_nx8;
	sa->ctx.cost_tbl = sa->d_cost_tbl; sa->ctx.L = L;
	sa->ctx.dict_limit = sa->cfg.dict_limit; sa->ctx.max_scan = sa->cfg.max_bucket_scan; sa->ctx.top_k = sa->cfg.top_k;

	HIPCHK(alloc_base(sa->base, (uint32_t)n, ckpt_elems));
	HIPCHK(alloc_base(sa->scratch, (uint32_t)n, ckpt_elems));

	const size_t K = sa->cfg.neighbours_per_step;
	HIPCHK(hipMalloc(&sa->nbr.cost, sizeof(uint64_t) * K));
	HIPCHK(hipMalloc(&sa->nbr.ndiffs, sizeof(uint32_t) * K));
	HIPCHK(hipMalloc(&sa->nbr.walked, sizeof(uint32_t) * K));
	HIPCHK(hipMalloc(&sa->nbr.win, sizeof(uint32_t) * 2 * K));
	HIPCHK(hipMemset(sa->nbr.win, 0xFF, sizeof(uint32_t) * 2 * K));
	HIPCHK(hipMalloc(&sa->nbr.win2, sizeof(uint32_t) * K));
	HIPCHK(hipMemset(sa->nbr.win2, 0xFF, sizeof(uint32_t) * K));
	memset(&sa->bulk, 0, sizeof sa->bulk);
	HIPCHK(hipMalloc(&sa->bulk.ckey, sizeof(uint64_t) * K));
	HIPCHK(hipMalloc(&sa->bulk.cwin, sizeof(uint4) * K));
	HIPCHK(hipMalloc(&sa->bulk.taken, sizeof(uint32_t) * K));
	HIPCHK(hipMalloc(&sa->bulk.cstate, 2u * (size_t)K));
	HIPCHK(hipMalloc(&sa->bulk.cflags, sizeof(uint32_t) * (size_t)K));
	HIPCHK(hipMemset(sa->bulk.cflags, 0, sizeof(uint32_t) * (size_t)K));
	HIPCHK(hipMalloc(&sa->bulk.hdr, sizeof(unsigned long long) * 8));
	{
		const unsigned long long h0[8] = { 0, 0, 0, 0, 0, 0, ~0ull, 0 };
		HIPCHK(hipMemcpy(sa->bulk.hdr, h0, sizeof h0, hipMemcpyHostToDevice));
	}
	HIPCHK(hipMalloc(&sa->nbr.dpos, sizeof(uint32_t) * K * MGL_MAX_DIFFS));
	HIPCHK(hipMalloc(&sa->nbr.dold, sizeof(mgl_pk) * K * MGL_MAX_DIFFS));
	HIPCHK(hipMalloc(&sa->nbr.dnew, sizeof(mgl_pk) * K * MGL_MAX_DIFFS));
	HIPCHK(hipMalloc(&sa->d_aos, sizeof(mgl_packet) * n));
	HIPCHK(hipMalloc(&sa->d_cum, sizeof(uint64_t) * n));
	HIPCHK(hipMalloc(&sa->d_final_probs, sizeof(uint16_t) * ckpt_elems));
	HIPCHK(hipMalloc(&sa->d_topk_pk, sizeof(mgl_pk) * 64));
	HIPCHK(hipMalloc(&sa->d_topk_cost, sizeof(uint64_t) * 64));
	HIPCHK(hipMalloc(&sa->d_small, sizeof(uint32_t) * 16));
	sa->sub_cap = 1u << 20;
	HIPCHK(hipMalloc(&sa->d_sub_offs, sizeof(uint32_t) * sa->sub_cap));
	HIPCHK(hipMalloc(&sa->d_sub_lens, sizeof(uint32_t) * sa->sub_cap));

	/* LDS budget: 4 KiB cost table + per wave {probabilities, 2x272 length prices, journal} */
	sa->per_wave_bytes = ckpt_elems * 2u + MGL_PRICE_WORDS * 4u + MGL_MAX_DIFFS * (8u + 8u + 4u);
	sa->waves_per_block = 4;
	while (sa->waves_per_block > 1 && 4096u + sa->waves_per_block * sa->per_wave_bytes > 160u * 1024u) sa->waves_per_block--;
	sa->nbr_lds = 4096u + sa->waves_per_block * sa->per_wave_bytes;
	sa->walk_lds = 4096u + ckpt_elems * 2u + MGL_PRICE_WORDS * 4u;
	HIPCHK(hipFuncSetAttribute((const void*)k_neighbours, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sa->nbr_lds));
	HIPCHK(hipFuncSetAttribute((const void*)k_rebuild, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sa->walk_lds));
	HIPCHK(hipFuncSetAttribute((const void*)k_topk_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sa->walk_lds));

	/* incremental path: bitmaps, special-state records, dense checkpoints, event chains */
	sa->incremental = !(sa->cfg.flags & MGL_F_FULLWALK);
	if (sa->incremental) {
		Base2& b = sa->b2;
		memset(&b, 0, sizeof b);
		b.slab = sa->base.v.slab;
		b.onwalk = sa->base.v.onwalk;
		b.nw0 = (uint32_t)(((size_t)n + 63) / 64);
		b.nw1 = (b.nw0 + 63) / 64;
		b.nw2 = (b.nw1 + 63) / 64;
		b.nck = (uint32_t)(((size_t)n + (1u << MGL_CK2_SHIFT) - 1) >> MGL_CK2_SHIFT);
		b.ck_elems = ckpt_elems;
		b.pool_cap = (uint32_t)(24 * n + (size_t)L.total * 272 + 4096);
		/* chain index: one entry per context and block of the parallel builder (the builder's per-block counts are its source) */
		b.sb_shift = n <= (1u << 20) ? 8u : n <= (1u << 23) ? 9u : MGL_PB_MAX_SHIFT;
		b.nsb = (uint32_t)((n + (1u << b.sb_shift) - 1) >> b.sb_shift);
		b.sb_stride = (b.nsb + 2u + 3u) & ~3u;
		size_t bytes = 0;
		{
			int rc = alloc_b2(b, n, false, true, &bytes);
			if (rc) return rc;
		}
		sa->parallel_build = !(sa->cfg.flags & MGL_F_SERIAL_BUILD);
		if (sa->parallel_build) {
			PBuild& pb = sa->pb;
			memset(&pb, 0, sizeof pb);
			pb.shift = b.sb_shift; /* 256 / 512 / 1 024 positions per block up to 1 MiB / 8 MiB / above */
			pb.nblk = (uint32_t)((n + (1u << pb.shift) - 1) >> pb.shift);
			HIPCHK(hipMalloc(&pb.exits, sizeof(uint16_t) * (size_t)pb.nblk * MGL_PB_ENTRIES));
			HIPCHK(hipMalloc(&pb.entry, sizeof(uint32_t) * ((size_t)pb.nblk + 1)));
			pb.ngrp = (pb.nblk + MGL_PB_GROUP - 1u) / MGL_PB_GROUP;
			HIPCHK(hipMalloc(&pb.gexits, sizeof(uint16_t) * (size_t)pb.ngrp * MGL_PB_ENTRIES));
			HIPCHK(hipMalloc(&pb.gentry, sizeof(uint16_t) * (size_t)pb.ngrp));
			{
				const size_t nch = ((size_t)pb.nblk + MGL_PB_SCAN_CHUNK - 1u) / MGL_PB_SCAN_CHUNK;
				HIPCHK(hipMalloc(&pb.ch_map, sizeof(uint64_t) * nch));
				HIPCHK(hipMalloc(&pb.ch_vs, sizeof(uint32_t) * 8 * nch));
				HIPCHK(hipMalloc(&pb.ch_pk, sizeof(uint32_t) * nch));
				HIPCHK(hipMalloc(&pb.ch_state, sizeof(uint32_t) * 8 * nch));
			}
			HIPCHK(hipMalloc(&pb.gsum, sizeof(uint32_t) * (size_t)((pb.nblk + MGL_PB_OFF_ROWS - 1u) / MGL_PB_OFF_ROWS) * ckpt_elems));
			HIPCHK(hipMalloc(&pb.tf_ctx, sizeof(uint64_t) * pb.nblk));
			HIPCHK(hipMalloc(&pb.tf_dist, sizeof(uint32_t) * 8 * (size_t)pb.nblk));
			HIPCHK(hipMalloc(&pb.tf_pk, sizeof(uint32_t) * pb.nblk));
			HIPCHK(hipMalloc(&pb.st_in, sizeof(uint32_t) * 8 * (size_t)pb.nblk));
			HIPCHK(hipMalloc(&pb.hist, sizeof(uint32_t) * (size_t)pb.nblk * ckpt_elems));
			HIPCHK(hipMalloc(&pb.stage, sizeof(uint64_t) * (size_t)pb.nblk * ((MGL_PB_STAGE_PER_POS << pb.shift) + 32u)));
			HIPCHK(hipMalloc(&pb.stage_n, sizeof(uint32_t) * (size_t)pb.nblk));
			HIPCHK(hipMalloc(&pb.stage_over, sizeof(uint32_t)));
			HIPCHK(hipMemset(pb.stage_over, 0, sizeof(uint32_t)));
			HIPCHK(hipMalloc(&pb.acc, sizeof(unsigned long long) * 16));
			HIPCHK(hipMemset(pb.acc, 0, sizeof(unsigned long long) * 16));
			pb.seg_cap = b.pool_cap / MGL_PB_SEG + ckpt_elems + 64u;
			HIPCHK(hipMalloc(&pb.seg_off, sizeof(uint32_t) * (ckpt_elems + 1)));
			HIPCHK(hipMalloc(&pb.unres, pb.seg_cap));
		}
		sa->snapshots = !(sa->cfg.flags & MGL_F_NO_SNAPSHOTS);
		if (sa->snapshots) {
			for (Base2* s : { &sa->snap_lit, &sa->snap_best }) {
				*s = b;
				s->slab = nullptr; s->onwalk = nullptr; s->sp0 = s->sp1 = s->sp2 = nullptr; s->sp_state = nullptr;
				s->ck_probs = nullptr; s->ch_off = s->ch_len = s->ch_cap = s->ch_pos = nullptr; s->ch_ev = nullptr;
				s->pool_top = nullptr;
				int rc = alloc_b2(*s, n, true, false, nullptr);
				if (rc) return rc;
			}
			HIPCHK(hipMalloc(&sa->d_snap_meta, sizeof(SnapMeta) * 2));
			HIPCHK(hipMemset(sa->d_snap_meta, 0, sizeof(SnapMeta) * 2));
			sa->d_best = sa->snap_best.slab; /* packets_best lives in the best snapshot */
		}
		if (sa->cfg.flags & MGL_F_PROFILE) {
			HIPCHK(hipMalloc(&sa->d_prof, sizeof(unsigned long long) * (32 + K)));
			HIPCHK(hipMemset(sa->d_prof, 0, sizeof(unsigned long long) * (32 + K)));
		}
		HIPCHK(hipMalloc(&sa->d_todo, sizeof(uint32_t) * (K + 1)));
		HIPCHK(hipMemset(sa->d_todo, 0, sizeof(uint32_t) * (K + 1)));
		sa->b2_bytes = bytes;
		sa->incremental_apply = getenv("MGL_NO_INCREMENTAL_APPLY") == nullptr;
		{
			ApplyBuf& ab = sa->ab;
			memset(&ab, 0, sizeof ab);
			/* one workgroup per touched context plans the rewrite; the copies run as job lists */
			sa->apply_blocks = 256;
			ab.scratch_cap = b.pool_cap + 256u;
			ab.span_cap = 1u << 22;
			ab.job_cap = 1u << 20;
			HIPCHK(hipMalloc(&ab.hdr, sizeof(uint32_t) * 16));
			HIPCHK(hipMemset(ab.hdr, 0, sizeof(uint32_t) * 16));
			HIPCHK(hipMalloc(&ab.ins_key, sizeof(uint16_t) * MGL_BATCH_ALLOC)); /* a single accept uses MGL_APPLY_CAP of it */
			HIPCHK(hipMalloc(&ab.rem_key, sizeof(uint16_t) * MGL_BATCH_ALLOC)); /* a single accept uses MGL_APPLY_CAP of it */
			HIPCHK(hipMalloc(&ab.ins_pos, sizeof(uint32_t) * MGL_BATCH_ALLOC)); /* a single accept uses MGL_APPLY_CAP of it */
			HIPCHK(hipMalloc(&ab.rem_pos, sizeof(uint32_t) * MGL_BATCH_ALLOC)); /* a single accept uses MGL_APPLY_CAP of it */
			HIPCHK(hipMalloc(&ab.tctx, sizeof(uint16_t) * 16384));
			HIPCHK(hipMalloc(&ab.scratch_pos, sizeof(uint32_t) * (size_t)ab.scratch_cap));
			HIPCHK(hipMalloc(&ab.scratch_ev, sizeof(uint16_t) * (size_t)ab.scratch_cap));
			HIPCHK(hipMalloc(&ab.span_pos, sizeof(uint32_t) * (size_t)ab.span_cap));
			HIPCHK(hipMalloc(&ab.span_ev, sizeof(uint16_t) * (size_t)ab.span_cap));
			HIPCHK(hipMalloc(&ab.jobs_b, sizeof(uint4) * (size_t)ab.job_cap));
			HIPCHK(hipMalloc(&ab.jobs_c, sizeof(uint4) * (size_t)ab.job_cap));
			BatchBuf& bt = sa->batch;
			memset(&bt, 0, sizeof bt);
			HIPCHK(hipMalloc(&bt.hdr, sizeof(uint32_t) * 16));
			HIPCHK(hipMemset(bt.hdr, 0, sizeof(uint32_t) * 16));
			HIPCHK(hipMalloc(&bt.cl, sizeof(uint32_t) * 8 * MGL_BATCH_MAX));
			HIPCHK(hipMalloc(&bt.jpos, sizeof(uint32_t) * MGL_BATCH_MAX * MGL_MAX_DIFFS));
			HIPCHK(hipMalloc(&bt.jnew, sizeof(mgl_pk) * MGL_BATCH_MAX * MGL_MAX_DIFFS));
			HIPCHK(hipMalloc(&bt.jold, sizeof(mgl_pk) * MGL_BATCH_MAX * MGL_MAX_DIFFS));
			HIPCHK(hipMalloc(&bt.st_ikey, sizeof(uint16_t) * MGL_BATCH_ALLOC));
			HIPCHK(hipMalloc(&bt.st_rkey, sizeof(uint16_t) * MGL_BATCH_ALLOC));
			HIPCHK(hipMalloc(&bt.st_ipos, sizeof(uint32_t) * MGL_BATCH_ALLOC));
			HIPCHK(hipMalloc(&bt.st_rpos, sizeof(uint32_t) * MGL_BATCH_ALLOC));
			HIPCHK(hipMalloc(&bt.ops, sizeof(uint4) * 2 * (size_t)MGL_BATCH_MAX * MGL_BATCH_OPCAP));
			HIPCHK(hipMalloc(&bt.ins_cl, MGL_BATCH_ALLOC));
			HIPCHK(hipMalloc(&bt.rem_cl, MGL_BATCH_ALLOC));
			bt.nctx = L.total;
			bt.runs_cap = 1u << 17;
			HIPCHK(hipMalloc(&bt.runs, sizeof(uint4) * 2 * (size_t)bt.runs_cap));
			for (uint32_t** p : { &bt.cnt_i, &bt.cnt_r, &bt.off_i, &bt.off_r, &bt.cur_i, &bt.cur_r, &bt.touched }) {
				HIPCHK(hipMalloc(p, sizeof(uint32_t) * (bt.nctx + 64u)));
				HIPCHK(hipMemset(*p, 0, sizeof(uint32_t) * (bt.nctx + 64u)));
			}
			HIPCHK(hipMalloc(&bt.bk_ipos, sizeof(uint32_t) * MGL_BATCH_ALLOC));
			HIPCHK(hipMalloc(&bt.bk_rpos, sizeof(uint32_t) * MGL_BATCH_ALLOC));
			HIPCHK(hipMalloc(&bt.bk_ibit, sizeof(uint16_t) * MGL_BATCH_ALLOC));
			HIPCHK(hipMalloc(&bt.bk_icl, MGL_BATCH_ALLOC));
			HIPCHK(hipMalloc(&bt.bk_rcl, MGL_BATCH_ALLOC));
			HIPCHK(hipMalloc(&bt.acc, sizeof(long long) * 4));
			HIPCHK(hipMemset(bt.acc, 0, sizeof(long long) * 4));
			HIPCHK(hipHostMalloc((void**)&sa->h_bstat, sizeof(uint32_t) * 16, hipHostMallocDefault));
			HIPCHK(hipEventCreateWithFlags(&sa->ev_bstat, hipEventDisableTiming));
			sa->batch_ok = sa->incremental_apply && getenv("MGL_NO_BATCH") == nullptr;

		}
		{
			/* journal + context bitmap + max(model + price tables, change lists + context list) */
			const uint32_t fixed = MGL_MAX_DIFFS * (8u + 8u + 4u) + (((((L.total + 31u) >> 5) + 3u) & ~3u) * 4u) ;
			const uint32_t model = ckpt_elems * 2u + MGL_PRICE_WORDS * 4u;
			/* first-pass list size: a step of few neighbours leaves most of the LDS idle, and on repetitive inputs
			 * (long matches everywhere) windows are long: give the lists the room */
			sa->chg_cap = K <= 1024u ? 4u * MGL_CHG_CAP : (K <= 2048u ? 2u * MGL_CHG_CAP : MGL_CHG_CAP);
			const uint32_t lists = sa->chg_cap * (4u + 4u + 2u + 2u) + 2u * sa->chg_cap * 2u;
			sa->per_wave2 = (fixed + (model > lists ? model : lists) + 15u) & ~15u;
			sa->per_wave_pick = (model + 15u) & ~15u; /* no journal / bitmap in the pick half */
			sa->per_wave_rest = (fixed + lists + 15u) & ~15u;
		}
		{
			BigScratch& g = sa->big;
			memset(&g, 0, sizeof g);
			g.chg_cap = sa->chg_cap;
			g.cap = MGL_BIG_CAP; g.uctx_cap = ckpt_elems; g.slots = (uint32_t)(K > 512 ? K : 512); /* every neighbour of a step may need one */
			HIPCHK(hipMalloc(&g.ins_key, sizeof(uint16_t) * (size_t)g.cap * g.slots));
			HIPCHK(hipMalloc(&g.rem_key, sizeof(uint16_t) * (size_t)g.cap * g.slots));
			HIPCHK(hipMalloc(&g.ins_pos, sizeof(uint32_t) * (size_t)g.cap * g.slots));
			HIPCHK(hipMalloc(&g.rem_pos, sizeof(uint32_t) * (size_t)g.cap * g.slots));
			HIPCHK(hipMalloc(&g.uctx, sizeof(uint16_t) * (size_t)g.uctx_cap * g.slots));
			HIPCHK(hipMalloc(&sa->d_todo2, sizeof(uint32_t) * (K + 1)));
			HIPCHK(hipMemset(sa->d_todo2, 0, sizeof(uint32_t) * (K + 1)));
		}
		/* waves per workgroup that packs the most waves into a CU's 160 KiB of LDS */
		{
			uint32_t best_w = 1, best_total = 0;
			for (uint32_t w = 1; w <= 8; w++) {
				const uint32_t bytes = 4096u + w * sa->per_wave2;
				if (bytes > 160u * 1024u) break;
				const uint32_t tot = (160u * 1024u / bytes) * w;
				if (tot > best_total) { best_total = tot; best_w = w; }
			}
			(void)best_w; /* measured: one wavefront per workgroup wins although it packs fewer waves (LDS is
			               * released per workgroup, and neighbour run times have a long tail) */
			sa->waves_per_block2 = 1;
		}
		sa->nbr2_lds = 4096u + sa->waves_per_block2 * sa->per_wave2;
		sa->build_lds = 4096u + ckpt_elems * 2u + ckpt_elems * 8u;
		HIPCHK(hipMalloc(&sa->d_counts, sizeof(uint32_t) * 16)); /* [0..7] live, [8..15] the last finished step's */
		HIPCHK(hipMemset(sa->d_counts, 0, sizeof(uint32_t) * 16));
		HIPCHK(hipMalloc(&sa->d_todo3, sizeof(uint32_t) * (K + 1)));
		HIPCHK(hipMemset(sa->d_todo3, 0, sizeof(uint32_t) * (K + 1)));
		sa->big.todo_in = sa->d_todo; sa->big.todo_in_count = sa->d_counts; sa->big.spill_ctr = sa->d_counts + 2;
		HIPCHK(hipFuncSetAttribute((const void*)k_neighbours2<false, MGL_NBR_FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sa->nbr2_lds));
		HIPCHK(hipFuncSetAttribute((const void*)k_neighbours2<false, MGL_NBR_FULL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sa->nbr2_lds));
		HIPCHK(hipFuncSetAttribute((const void*)k_neighbours2<false, MGL_NBR_PICK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIPCHK(hipFuncSetAttribute((const void*)k_neighbours2<false, MGL_NBR_REST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sa->nbr2_lds));
		HIPCHK(hipFuncSetAttribute((const void*)k_neighbours2<true, MGL_NBR_FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sa->nbr2_lds + 12u * MGL_BIG_CAP)));
		HIPCHK(hipFuncSetAttribute((const void*)k_neighbours2<true, MGL_NBR_FULL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sa->nbr2_lds + 12u * MGL_BIG_CAP)));
		HIPCHK(hipMalloc(&sa->d_pickrec, sizeof(uint4) * K));
		HIPCHK(hipMalloc(&sa->d_pickstate, sizeof(uint4) * 2 * K));
		sa->split_nbr = getenv("MGL_NO_SPLIT") == nullptr;
		if (sa->split_nbr) {
			/* the second half's re-simulation as its own launch (k_sim): lists and headers per neighbour */
			BigScratch& g = sa->big;
			HIPCHK(hipMalloc(&g.sim_hdr, sizeof(uint4) * (size_t)K));
			HIPCHK(hipMalloc(&g.sim_keys, sizeof(uint16_t) * 2u * sa->chg_cap * (size_t)K));
			HIPCHK(hipMalloc(&g.sim_pos, sizeof(uint32_t) * 2u * sa->chg_cap * (size_t)K));
			HIPCHK(hipMemset(g.sim_hdr, 0xFF, sizeof(uint4) * (size_t)K));
			HIPCHK(hipMalloc(&g.sim_hdr2, sizeof(uint4) * (size_t)K));
			HIPCHK(hipMalloc(&g.sim_slot2, sizeof(uint32_t) * (size_t)K));
			HIPCHK(hipMemset(g.sim_hdr2, 0xFF, sizeof(uint4) * (size_t)K));
			/* continuation records: the second half saves a walk that stops at a repair pick, the second pass resumes it
			 * (not beside the look-ahead, whose speculative launch would write the slots the running second pass reads) */
			if (getenv("MGL_NO_CONT") == nullptr && getenv("MGL_LOOKAHEAD") == nullptr) {
				HIPCHK(hipMalloc(&g.cont, sizeof(uint32_t) * MGL_CONT_WORDS * (size_t)g.slots));
				HIPCHK(hipMemset(g.cont, 0, sizeof(uint32_t) * MGL_CONT_WORDS * (size_t)g.slots));
			}
			/* look-ahead: a second set of everything pick + walk write, the check's list and marks, two more streams */
			/* Opt-in (MGL_LOOKAHEAD=1): exact (tests/test_gpu_parity.py runs it against the plain order), but slower on MI355X as
			 * measured (c3: 1.58 ms per step against 1.37, profiles/r02_lookahead_c3.txt): the pick kernel is bound by LDS capacity
			 * and the re-simulations by the memory system, so kernels running beside each other take their time from each other. */
			sa->la_enabled = sa->incremental_apply && getenv("MGL_LOOKAHEAD") != nullptr;
			if (sa->la_enabled) {
				mgl_sa::NbrSet& t = sa->alt;
				HIPCHK(hipMalloc(&t.nbr.cost, sizeof(uint64_t) * K));
				HIPCHK(hipMalloc(&t.nbr.ndiffs, sizeof(uint32_t) * K));
				HIPCHK(hipMalloc(&t.nbr.walked, sizeof(uint32_t) * K));
				HIPCHK(hipMalloc(&t.nbr.win, sizeof(uint32_t) * 2 * K));
				HIPCHK(hipMemset(t.nbr.win, 0xFF, sizeof(uint32_t) * 2 * K));
				HIPCHK(hipMalloc(&t.nbr.win2, sizeof(uint32_t) * K));
				HIPCHK(hipMemset(t.nbr.win2, 0xFF, sizeof(uint32_t) * K));
				HIPCHK(hipMalloc(&t.nbr.dpos, sizeof(uint32_t) * K * MGL_MAX_DIFFS));
				HIPCHK(hipMalloc(&t.nbr.dold, sizeof(mgl_pk) * K * MGL_MAX_DIFFS));
				HIPCHK(hipMalloc(&t.nbr.dnew, sizeof(mgl_pk) * K * MGL_MAX_DIFFS));
				HIPCHK(hipMalloc(&t.pickrec, sizeof(uint4) * K));
				HIPCHK(hipMalloc(&t.pickstate, sizeof(uint4) * 2 * K));
				HIPCHK(hipMalloc(&t.sim_hdr, sizeof(uint4) * (size_t)K));
				HIPCHK(hipMemset(t.sim_hdr, 0xFF, sizeof(uint4) * (size_t)K));
				HIPCHK(hipMalloc(&t.sim_keys, sizeof(uint16_t) * 2u * sa->chg_cap * (size_t)K));
				HIPCHK(hipMalloc(&t.sim_pos, sizeof(uint32_t) * 2u * sa->chg_cap * (size_t)K));
				HIPCHK(hipMalloc(&t.todo, sizeof(uint32_t) * (K + 1)));
				HIPCHK(hipMemset(t.todo, 0, sizeof(uint32_t) * (K + 1)));
				HIPCHK(hipMalloc(&t.counts, sizeof(uint32_t) * 16));
				HIPCHK(hipMemset(t.counts, 0, sizeof(uint32_t) * 16));
				HIPCHK(hipMalloc(&sa->d_la_mark, K));
				HIPCHK(hipMemset(sa->d_la_mark, 0, K));
				HIPCHK(hipMalloc(&sa->d_la_list, sizeof(uint32_t) * K));
				HIPCHK(hipMalloc(&sa->d_la_hdr, sizeof(uint32_t) * 2));
				HIPCHK(hipMemset(sa->d_la_hdr, 0, sizeof(uint32_t) * 2));
				int plo = 0, phi = 0;
				HIPCHK(hipDeviceGetStreamPriorityRange(&plo, &phi));
				HIPCHK(hipStreamCreateWithPriority(&sa->stream5, hipStreamDefault, plo));
				HIPCHK(hipStreamCreateWithPriority(&sa->stream6, hipStreamDefault, plo));
				HIPCHK(hipEventCreateWithFlags(&sa->ev_redo, hipEventDisableTiming));
				HIPCHK(hipEventCreateWithFlags(&sa->ev_la_check, hipEventDisableTiming));
				HIPCHK(hipEventCreateWithFlags(&sa->ev_la_zero, hipEventDisableTiming));
				for (int i = 0; i < 8; i++) HIPCHK(hipEventCreateWithFlags(&sa->ev_spec[i], hipEventDisableTiming));
			}
		}
		if (getenv("MGL_SIMW")) { const int w = atoi(getenv("MGL_SIMW")); if (w == 1 || w == 2 || w == 4 || w == 8) sa->sim_waves = (uint32_t)w; }
		sa->adaptive = sa->split_nbr && getenv("MGL_NO_ADAPT") == nullptr;
		sa->form_single = !sa->split_nbr; /* one-kernel form only, or the split form until the device recommends otherwise */
		/* eight pick wavefronts share a cost table per workgroup: 4 KiB + 8 x 9.3 KiB = 78 KiB, two
		 * workgroups = 16 wavefronts per CU, so the 4 096 neighbours of a c2 step are all resident at
		 * once (with two per workgroup 14 fit and the last 512 waited for a slot: 127 -> 108 us).
		 * Above 1 MiB top-K run times vary too much for wavefronts to hold each other's LDS: one per
		 * workgroup there (c3, first 200 steps: 3.04 ms per step against 3.26 with eight) */
		sa->pick_waves = getenv("MGL_PICK_WAVES") ? (uint32_t)atoi(getenv("MGL_PICK_WAVES")) : 0u;
		if (sa->pick_waves != 1 && sa->pick_waves != 2 && sa->pick_waves != 4 && sa->pick_waves != 8) sa->pick_waves = 0;
		{
			/* the workgroup size that puts most wavefronts on a CU's 160 KiB (larger models -- lc > 0 -- fit fewer) */
			uint32_t best_w = 1, best_resident = 0;
			for (uint32_t w = 1; w <= 8; w *= 2) {
				const uint32_t bytes = (MGL_PICK_T_GLOBAL ? 0u : 4096u) + w * sa->per_wave_pick;
				if (bytes > 160u * 1024u) break;
				const uint32_t resident = (160u * 1024u / ((bytes + 1279u) / 1280u * 1280u)) * w;
				if (resident > best_resident) { best_resident = resident; best_w = w; }
			}
			if (n > (1u << 20)) best_w = 1;
			if (sa->pick_waves == 0 || (MGL_PICK_T_GLOBAL ? 0u : 4096u) + sa->pick_waves * sa->per_wave_pick > 160u * 1024u) sa->pick_waves = best_w;
		}
		HIPCHK(hipFuncSetAttribute((const void*)k_sim<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
		HIPCHK(hipFuncSetAttribute((const void*)k_sim<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
		HIPCHK(hipMalloc(&sa->d_traffic, sizeof(unsigned long long) * 2));
		HIPCHK(hipMemset(sa->d_traffic, 0, sizeof(unsigned long long) * 2));
		HIPCHK(hipFuncSetAttribute((const void*)k_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sa->build_lds));
		HIPCHK(hipFuncSetAttribute((const void*)k_build_end, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sa->build_lds));
	}

	/* targets: stratified by packet ordinal (default), or position draws (MGL_F_POSITION_TARGETS; the look-ahead needs them:
	 * it keeps speculative results whose target draw lands on the same position after the accept) */
	if (!(sa->cfg.flags & MGL_F_POSITION_TARGETS) && !sa->la_enabled && getenv("MGL_POSITION_TARGETS") == nullptr) { /* (the environment switch: A/B runs) */
		sa->ctx.strat_nblk = (uint32_t)((n + 4095u) / 4096u);
		HIPCHK(hipMalloc(&sa->d_strat_pre, sizeof(uint32_t) * (sa->ctx.strat_nblk + 2u)));
		HIPCHK(hipMemset(sa->d_strat_pre, 0, sizeof(uint32_t) * (sa->ctx.strat_nblk + 2u)));
		sa->ctx.strat_pre = sa->d_strat_pre;
		HIPCHK(hipMalloc(&sa->d_strat_tgt, sizeof(uint32_t) * sa->cfg.neighbours_per_step));
		sa->ctx.strat_tgt = sa->d_strat_tgt;
	}
	sa->sqrt_thresh = ceil_sqrt_u64(sa->cfg.iters_per_epoch);
	/* a bulk step costs a parallel rebuild (about 3 single steps at 100 KB, 6 at 10 MB, 20 at 100 MB) */
	/* ... unless the step takes few moves: those are patched in at once (mgl_kernels5.hip) for about half a single step's time, so a bulk
	 * step pays as soon as it can be expected to take two moves */
	sa->bulk_threshold = getenv("MGL_BULK_THRESHOLD") ? (uint32_t)atoi(getenv("MGL_BULK_THRESHOLD"))
	                     : sa->batch_ok ? 2u : (n <= (1u << 20) ? 16u : n <= (1u << 24) ? 32u : 128u);
	if (sa->bulk_threshold == 0) sa->bulk_threshold = 1;

	/* packet_slab_new: all-literal current and best slabs */
	hipLaunchKernelGGL(k_fill_literal, dim3(1024), dim3(256), 0, sa->stream, sa->base.v.slab, sa->ctx.n);
	if (!sa->d_best) HIPCHK(hipMalloc(&sa->d_best, sizeof(mgl_pk) * n));
	hipLaunchKernelGGL(k_fill_literal, dim3(1024), dim3(256), 0, sa->stream, sa->d_best, sa->ctx.n);
	HIPCHK(hipGetLastError());
	int rc = rebuild_base(sa, 0);
	if (rc) return rc;
	if (sa->incremental && sa->snapshots && (rc = launch_snapshot(sa, sa->snap_lit, 0, 0, 0))) return rc;
	HIPCHK(hipStreamSynchronize(sa->stream));
	{
		/* an input the structures were not sized for shows here, not steps later inside mgl_sa_run */
		Control c0;
		HIPCHK(hipMemcpy(&c0, sa->base.ctl, sizeof c0, hipMemcpyDeviceToHost));
		if (c0.error_flags) return fail(MGL_EDEVICE, "mgl_sa_create: building the base structures of the all-literal slab failed");
	}
	return MGL_OK;
}

extern "C" mgl_sa* mgl_sa_create(const uint8_t* data, size_t n, mgl_properties props, const mgl_sa_config* cfg)
{
	if (!data || n == 0 || !cfg) { fail(MGL_EINVAL, "mgl_sa_create: bad data/size/config"); return nullptr; }
	/* 32-bit chain pool: 24 n + 272 contexts + slack must fit (and costs stay below 2^44: k_decide packs cost << 20 | j) */
	if (n > MGL_MAX_INPUT) { fail(MGL_EINVAL, "mgl_sa_create: input larger than 176 000 000 bytes is not supported"); return nullptr; }
	if (props.lc + props.lp > 4 || props.pb > 4 || props.lc > 8) { fail(MGL_EINVAL, "mgl_sa_create: unsupported lc/lp/pb"); return nullptr; }
	mgl_sa* sa = new (std::nothrow) mgl_sa();
	if (!sa) { fail(MGL_ENOMEM, "mgl_sa_create: out of host memory"); return nullptr; }
	sa->cfg = *cfg;
	sa->props = props;
	sa->n = (uint32_t)n;
	sa->device = cfg->device;
	if (sa->cfg.neighbours_per_step == 0) sa->cfg.neighbours_per_step = 1;
	if (sa->cfg.neighbours_per_step > (1u << 20)) { delete sa; fail(MGL_EINVAL, "neighbours_per_step > 2^20"); return nullptr; }
	if (sa->cfg.top_k == 0) sa->cfg.top_k = 20;
	if (sa->cfg.top_k > MGL_MAX_TOPK) { delete sa; fail(MGL_EINVAL, "top_k > 32"); return nullptr; }
	if (sa->cfg.dict_limit == 0) sa->cfg.dict_limit = 0x400000u;
	if (sa->cfg.iters_per_epoch == 0) sa->cfg.iters_per_epoch = n;
	int rc = create_impl(sa, data, n);
	if (rc != MGL_OK) {
		std::string keep = g_err;
		mgl_sa_destroy(sa);
		g_err = keep;
		return nullptr;
	}
	return sa;
}

static int read_ctl(mgl_sa* sa, BaseMem& b, Control* out)
{
	HIPCHK(hipMemcpyAsync(out, b.ctl, sizeof(Control), hipMemcpyDeviceToHost, sa->stream));
	HIPCHK(hipStreamSynchronize(sa->stream));
	return MGL_OK;
}
static int write_ctl(mgl_sa* sa, BaseMem& b, const Control* in)
{
	HIPCHK(hipMemcpyAsync(b.ctl, in, sizeof(Control), hipMemcpyHostToDevice, sa->stream));
	HIPCHK(hipStreamSynchronize(sa->stream));
	return MGL_OK;
}

/* the base still is the only holder of the best slab's structures and is about to be replaced */
static int keep_best_before_overwrite(mgl_sa* sa, Control& c)
{
	if (!(sa->incremental && sa->snapshots) || !c.best_is_current) return MGL_OK;
	int rc = launch_snapshot(sa, sa->snap_best, 1, 0, 0);
	if (rc) return rc;
	c.best_is_current = 0;
	return MGL_OK;
}

extern "C" int mgl_sa_begin_epoch(mgl_sa* sa, unsigned phase, int from_best)
{
	if (!sa) return fail(MGL_EINVAL, "null handle");
	HIPCHK(hipSetDevice(sa->device));
	Control c;
	int rc = read_ctl(sa, sa->base, &c);
	if (rc) return rc;
	if (from_best && c.best_cost == 0) from_best = 0; /* no best yet: packets_best is still the all-literal slab */
	const bool snaps = sa->incremental && sa->snapshots;
	const bool base_is_best = snaps && c.best_is_current;
	if (!from_best && (rc = keep_best_before_overwrite(sa, c))) return rc;
	c.iter = 0; c.cur_cost = 0; c.phase = phase; c.accepted_flag = 0; c.copy_best_flag = 0;
	if ((rc = write_ctl(sa, sa->base, &c))) return rc;
	/* MGL_ACCEPT_AUTO starts every epoch the same way, whatever the previous one left behind: a fresh block, bulk steps
	 * first from the all-literal slab (thousands of improving neighbours), single steps first from the best slab */
	sa->bulk_now = !from_best; sa->bulk_hold = 0; sa->blk_done = 0; sa->select_small = false;
	if (from_best && base_is_best) return MGL_OK; /* main.c:73-76 would copy packets_best over itself */
	if (snaps) {
		SnapMeta meta[2];
		HIPCHK(hipMemcpyAsync(meta, sa->d_snap_meta, sizeof meta, hipMemcpyDeviceToHost, sa->stream));
		HIPCHK(hipStreamSynchronize(sa->stream));
		const uint32_t which = from_best ? 1u : 0u;
		if (meta[which].valid) {
			if ((rc = launch_snapshot(sa, which ? sa->snap_best : sa->snap_lit, which, 1, 0))) return rc;
			HIPCHK(hipStreamSynchronize(sa->stream));
			return MGL_OK;
		}
	}
	if (from_best) HIPCHK(hipMemcpyAsync(sa->base.v.slab, sa->d_best, sizeof(mgl_pk) * (size_t)sa->n, hipMemcpyDeviceToDevice, sa->stream));
	else {
		hipLaunchKernelGGL(k_fill_literal, dim3(1024), dim3(256), 0, sa->stream, sa->base.v.slab, sa->ctx.n);
		HIPCHK(hipGetLastError());
	}
	if ((rc = rebuild_base(sa, 0))) return rc;
	if (from_best && sa->best_unverified) {
		/* a slab adopted from another chain is checked here, where it first matters: every packet against
		 * the input, and the total against the cost it came with */
		if ((rc = launch_validate(sa))) return rc;
		Control v;
		if ((rc = read_ctl(sa, sa->base, &v))) return rc;
		if (v.error_flags || v.rebuild_cost != v.best_cost) {
			v.error_flags = 0;
			(void)write_ctl(sa, sa->base, &v);
			return fail(MGL_EINVAL, "mgl_sa_begin_epoch: the adopted best slab is not a valid parse of the input at the cost it came with");
		}
		sa->best_unverified = false;
	}
	/* the base now is the best (or the literal) slab's: keep it for the next epoch */
	if (snaps && (rc = launch_snapshot(sa, from_best ? sa->snap_best : sa->snap_lit, from_best ? 1u : 0u, 0, 0))) return rc;
	HIPCHK(hipStreamSynchronize(sa->stream));
	return MGL_OK;
}

extern "C" int mgl_sa_set_slab(mgl_sa* sa, const mgl_packet* packets)
{
	if (!sa || !packets) return fail(MGL_EINVAL, "null argument");
	HIPCHK(hipSetDevice(sa->device));
	Control c;
	int rc = read_ctl(sa, sa->base, &c);
	if (rc) return rc;
	if ((rc = keep_best_before_overwrite(sa, c))) return rc;
	sa->bulk_now = true; sa->bulk_hold = 0; sa->blk_done = 0; sa->select_small = false; /* MGL_ACCEPT_AUTO starts over on a new slab */
	if ((rc = import_slab(sa, packets, sa->base.v.slab))) return rc;
	c.cur_cost = 0; c.accepted_flag = 0; c.copy_best_flag = 0; c.error_flags = 0;
	if ((rc = write_ctl(sa, sa->base, &c))) return rc;
	if ((rc = rebuild_base(sa, 0))) return rc;
	if ((rc = launch_validate(sa))) return rc;
	if ((rc = read_ctl(sa, sa->base, &c))) return rc;
	if (c.error_flags) return fail(MGL_EINVAL, "mgl_sa_set_slab: slab is not a valid parse of the input");
	return MGL_OK;
}

extern "C" int mgl_sa_set_temperature(mgl_sa* sa, uint64_t temperature)
{
	if (!sa) return fail(MGL_EINVAL, "null handle");
	if (temperature >> 40) return fail(MGL_EINVAL, "mgl_sa_set_temperature: temperature must be below 2^40 cost units");
	sa->temperature = temperature;
	return MGL_OK;
}

extern "C" int mgl_sa_seed_greedy(mgl_sa* sa, uint32_t candidates)
{
	if (!sa) return fail(MGL_EINVAL, "null handle");
	if (candidates == 0) return fail(MGL_EINVAL, "mgl_sa_seed_greedy: candidates must be > 0");
	HIPCHK(hipSetDevice(sa->device));
	Control c;
	int rc = read_ctl(sa, sa->base, &c);
	if (rc) return rc;
	if ((rc = keep_best_before_overwrite(sa, c))) return rc;
	sa->bulk_now = true; sa->bulk_hold = 0; sa->blk_done = 0; sa->select_small = false; /* MGL_ACCEPT_AUTO starts over on a new slab */
	hipLaunchKernelGGL(k_greedy_seed, dim3((sa->ctx.n + 255u) / 256u), dim3(256), 0, sa->stream, sa->ctx, sa->base.v.slab, candidates);
	HIPCHK(hipGetLastError());
	c.cur_cost = 0; c.accepted_flag = 0; c.copy_best_flag = 0; c.error_flags = 0;
	if ((rc = write_ctl(sa, sa->base, &c))) return rc;
	if ((rc = rebuild_base(sa, 0))) return rc;
	if ((rc = read_ctl(sa, sa->base, &c))) return rc;
	if (c.error_flags) return fail(MGL_EDEVICE, "mgl_sa_seed_greedy: the seeded slab failed the walk check");
	return MGL_OK;
}

static hipEvent_t pool_event(mgl_sa* sa, size_t i)
{
	while (sa->ev_pool.size() <= i) {
		hipEvent_t e = nullptr;
		if (hipEventCreate(&e) != hipSuccess) return nullptr;
		sa->ev_pool.push_back(e);
	}
	return sa->ev_pool[i];
}

/* MGL_ACCEPT_AUTO starts over: a fresh block, bulk steps first (a new slab says nothing about the old one's windows) */
static void auto_reset(mgl_sa* sa) { sa->bulk_now = true; sa->bulk_hold = 0; sa->blk_done = 0; sa->select_small = false; }
extern "C" int mgl_sa_set_accept_mode(mgl_sa* sa, int mode, uint32_t bulk_threshold)
{
	if (!sa) return fail(MGL_EINVAL, "null handle");
	if (mode != MGL_ACCEPT_AUTO && mode != MGL_ACCEPT_SINGLE && mode != MGL_ACCEPT_BULK) return fail(MGL_EINVAL, "mgl_sa_set_accept_mode: unknown mode");
	if (mode == MGL_ACCEPT_BULK && !(sa->incremental && sa->parallel_build)) return fail(MGL_EINVAL, "mgl_sa_set_accept_mode: bulk steps need the incremental engine and its parallel builder");
	sa->accept_mode = mode;
	if (bulk_threshold) sa->bulk_threshold = bulk_threshold;
	auto_reset(sa);
	return MGL_OK;
}
extern "C" int mgl_sa_step_modes(mgl_sa* sa, uint8_t* modes_out, size_t cap, size_t* count)
{
	if (!sa || !count) return fail(MGL_EINVAL, "null argument");
	*count = sa->mode_log.size();
	if (modes_out) memcpy(modes_out, sa->mode_log.data(), cap < sa->mode_log.size() ? cap : sa->mode_log.size());
	return MGL_OK;
}

/* MGL_ACCEPT_AUTO, after a block of steps, from device counters only (reproducible): bulk steps pay while a
 * step offers many improving neighbours AND their windows are short enough for several to be taken at once.
 * From the all-literal slab nothing resets the rep distances, so every window runs to the end of the file
 * and a bulk step takes one move like a single step, at three times the price: then single steps for a while. */
static void auto_decide(mgl_sa* sa, bool was_bulk, uint64_t block, uint64_t improving, uint64_t taken)
{
	const bool many = improving >= (uint64_t)sa->bulk_threshold * block;
	if (was_bulk) {
		/* (a bulk step that takes one move is a single step at a higher price: from the all-literal slab, where no window
		 * closes, or once improving neighbours have become rare) */
		if (sa->batch_ok ? 2u * taken < 3u * block : taken < 4u * block) { sa->bulk_now = false; sa->bulk_hold = 48; }
		else sa->bulk_now = many;
		return;
	}
	sa->bulk_hold = sa->bulk_hold > block ? sa->bulk_hold - block : 0;
	sa->bulk_now = many && sa->bulk_hold == 0;
}

static DecideArgs decide_args(const mgl_sa* sa)
{
	DecideArgs a;
	a.K = sa->cfg.neighbours_per_step; a.seed = sa->cfg.seed; a.iters_per_epoch = sa->cfg.iters_per_epoch;
	a.sqrt_thresh = sa->sqrt_thresh; a.temperature = sa->temperature;
	return a;
}
/* what closes a bulk step: best-slab tracking, the step's counters.  `gate` = the batch accept's status word when these
 * are queued before the host has read it (they then run beside the read-back), nullptr when the host knows the step is in */
static void launch_bulk_close(mgl_sa* sa, const uint32_t* gate)
{
	hipLaunchKernelGGL(k_bulk_finish, dim3(1), dim3(64), 0, sa->stream, sa->base.ctl, sa->bulk, sa->snapshots ? 1 : 0,
	                   sa->snapshots ? &sa->d_snap_meta[1].valid : (uint32_t*)nullptr, gate);
	if (sa->snapshots) {
		hipLaunchKernelGGL(k_bulk_keep_copy, dim3(1024), dim3(256), 0, sa->stream, (const Control*)sa->base.ctl, (const mgl_pk*)sa->base.v.slab, sa->d_best, sa->ctx.n, gate);
		hipLaunchKernelGGL(k_bulk_keep_undo, dim3(64), dim3(256), 0, sa->stream, (const Control*)sa->base.ctl, sa->nbr, sa->bulk, sa->d_best, gate);
	} else {
		hipLaunchKernelGGL(k_bulk_copy_best, dim3(256), dim3(256), 0, sa->stream, (const Control*)sa->base.ctl, (const mgl_pk*)sa->base.v.slab, sa->d_best, sa->ctx.n, gate);
	}
	hipLaunchKernelGGL(k_bulk_reset, dim3(1), dim3(64), 0, sa->stream, sa->bulk, gate);
	hipLaunchKernelGGL(k_bulk_step_end, dim3(1), dim3(64), 0, sa->stream, sa->base.ctl, sa->d_counts, sa->form_single ? 1 : 0, gate);
}
/* the tail of a bulk step: selection, journals into the slab, parallel rebuild, best-slab tracking */
static int launch_bulk_tail(mgl_sa* sa, uint64_t next_gstep)
{
	const uint32_t K = sa->cfg.neighbours_per_step;
	const DecideArgs a = decide_args(sa);
	const uint32_t blocks = (K + 255u) / 256u;
	hipLaunchKernelGGL(k_bulk_prep, dim3(blocks), dim3(256), 0, sa->stream, sa->ctx, sa->base.ctl, sa->nbr, a, sa->bulk);
	if (sa->select_small) {
		hipLaunchKernelGGL(k_bulk_select_small, dim3(1), dim3(1024), 0, sa->stream, sa->base.ctl, sa->nbr, sa->bulk, K);
	} else for (uint32_t r = 0; r < MGL_BULK_ROUNDS; r++) {
		hipLaunchKernelGGL(k_bulk_pairs, dim3(blocks, blocks), dim3(256), 0, sa->stream, sa->bulk, K, r);
		hipLaunchKernelGGL(k_bulk_round, dim3(blocks), dim3(256), 0, sa->stream, sa->base.ctl, sa->nbr, sa->bulk, sa->base.v.slab, K, r);
	}
	hipLaunchKernelGGL(k_bulk_end, dim3(1), dim3(64), 0, sa->stream, sa->base.ctl, sa->bulk, a);
	HIPCHK(hipGetLastError());
	int rc = MGL_OK;
	/* A step that took few moves patches the base structures for all of them at once (mgl_kernels5.hip); every kernel of
	 * that looks at the status word first, so a step that took none, or too many, passes through in a few microseconds. */
	uint32_t bstat[8] = { 2u, 0, 0, 0, 0, 0, 0, 0 }; /* without the batch path: everything is the rebuild's */
	bool closed = false; /* the closing kernels are queued (behind the batch accept's gate) and the gate is open */
	if (sa->batch_ok && !sa->force_rollbacks) { /* (the rollback net hangs under the rebuild: while a test forces it, steps go that way) */
		Base2& b = sa->b2;
		if (sa->force_batch_fail) { const uint32_t one = 1u; HIPCHK(hipMemcpyAsync(sa->batch.hdr + 9, &one, sizeof one, hipMemcpyHostToDevice, sa->stream)); }
		hipLaunchKernelGGL(k_batch_clusters, dim3(1), dim3(1024), 0, sa->stream, sa->ctx, (const Control*)sa->base.ctl, sa->nbr, sa->bulk, sa->batch);
		hipLaunchKernelGGL(k_batch_walk, dim3(MGL_BATCH_MAX), dim3(64), 0, sa->stream, sa->ctx, b, sa->batch);
		hipLaunchKernelGGL(k_batch_commit, dim3(MGL_BATCH_MAX), dim3(256), 0, sa->stream, sa->ctx, b, sa->base.ctl, sa->batch, sa->ab);
		hipLaunchKernelGGL(pb_levels, dim3((b.nw0 + 255) / 256), dim3(256), 0, sa->stream, (const uint64_t*)b.sp0, b.sp1, b.nw0, b.nw1);
		hipLaunchKernelGGL(pb_levels, dim3((b.nw1 + 255) / 256), dim3(256), 0, sa->stream, (const uint64_t*)b.sp1, b.sp2, b.nw1, b.nw2);
		/* the bitmaps are the new base's from here on -- if the batch accept goes through, which the status read below says */
		if (next_gstep != ~0ull && (rc = launch_targets_ahead(sa, next_gstep))) return rc;
		hipLaunchKernelGGL(k_batch_scan, dim3(1), dim3(1024), 0, sa->stream, sa->batch, sa->ab);
		hipLaunchKernelGGL(k_batch_fill, dim3(256), dim3(256), 0, sa->stream, sa->batch, sa->ab);
		hipLaunchKernelGGL(k_batch_chains, dim3(MGL_BATCH_GRID), dim3(MGL_BATCH_THREADS), 0, sa->stream, sa->ctx, b, sa->base.ctl, sa->batch, sa->ab);
		hipLaunchKernelGGL(k_batch_ckpt, dim3(1024, 8), dim3(256), 0, sa->stream, b, (const Control*)sa->base.ctl, sa->batch, sa->ab);
		hipLaunchKernelGGL(k_apply_jobs, dim3(1024), dim3(256), 0, sa->stream, b, (const Control*)sa->base.ctl, sa->ab, 0, (const uint32_t*)sa->batch.hdr);
		hipLaunchKernelGGL(k_apply_jobs, dim3(1024), dim3(256), 0, sa->stream, b, (const Control*)sa->base.ctl, sa->ab, 1, (const uint32_t*)sa->batch.hdr);
		hipLaunchKernelGGL(k_batch_end, dim3(1), dim3(64), 0, sa->stream, sa->base.ctl, sa->batch);
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(sa->h_bstat, sa->batch.hdr, sizeof bstat, hipMemcpyDeviceToHost, sa->stream)); /* pinned: no staging copy */
		HIPCHK(hipEventRecord(sa->ev_bstat, sa->stream));
		launch_bulk_close(sa, (const uint32_t*)sa->batch.hdr); /* behind the gate: they run while the host reads the status */
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventSynchronize(sa->ev_bstat));
		memcpy(bstat, sa->h_bstat, sizeof bstat);
		closed = bstat[0] == 3u || bstat[0] == 0u;
		if (!closed && sa->tgt_ahead) sa->tgt_ahead = 2; /* the step goes to the rebuild (or was never committed): the targets are made again */
		if (sa->force_batch_fail) {
			const uint32_t zero = 0u;
			HIPCHK(hipMemcpy(sa->batch.hdr + 9, &zero, sizeof zero, hipMemcpyHostToDevice));
			if (bstat[0] == 1u) sa->force_batch_fail--; /* (a step that took nothing, or went to the rebuild anyway, does not count) */
		}
	}
	if (sa->batch_ok) sa->select_small = bstat[7] <= 512u; /* this step's acceptable neighbours size the next step's selection */
	if (bstat[0] == 3u) sa->batch_accepts++; /* the moves are in: structures patched, exact cost in Control::rebuild_cost, every rep packet of the new walk checked */
	else if (bstat[0] != 0u) {
		/* the rebuild's: the journals into the slab (unless the batch accept got as far as writing them: status 1 = it gave
		 * up behind its commit), everything that hangs off the slab re-derived, the parse checked after the fact (soft window
		 * ends rest on an argument about rep distances, not on a proof for every coincidence of values): every packet of
		 * the new parse against the input; a parse that fails is taken back as a whole.  One small read-back per bulk
		 * step (a bulk step of this kind is a rebuild: milliseconds) */
		if (sa->batch_ok && bstat[0] == 1u) sa->batch_fallbacks++;
		if (bstat[0] != 1u)
			hipLaunchKernelGGL(k_bulk_write, dim3(64), dim3(256), 0, sa->stream, sa->base.ctl, sa->nbr, sa->bulk, sa->base.v.slab);
		HIPCHK(hipGetLastError());
		Control now;
		if (sa->batch_ok && bstat[0] == 1u) { /* apply_failed was raised: not the rebuild's concern */
			if ((rc = read_ctl(sa, sa->base, &now))) return rc;
			now.apply_failed = 0;
			if ((rc = write_ctl(sa, sa->base, &now))) return rc;
		}
		if ((rc = launch_pbuild(sa, sa->incremental))) return rc;
		if ((rc = read_ctl(sa, sa->base, &now))) return rc;
		if (sa->force_rollbacks && now.taken) { sa->force_rollbacks--; now.error_flags |= MGL_ERR_BAD_PACKET; } /* diagnostic (mgl_debug_set key 3): exercise the net */
		if (now.error_flags & MGL_ERR_BAD_PACKET) {
			hipLaunchKernelGGL(k_bulk_rollback, dim3(64), dim3(256), 0, sa->stream, sa->nbr, sa->bulk, sa->base.v.slab);
			HIPCHK(hipGetLastError());
			now.error_flags &= ~MGL_ERR_BAD_PACKET;
			now.accepted -= now.taken;
			now.taken = 0;
			sa->bulk_rollbacks++;
			if ((rc = write_ctl(sa, sa->base, &now))) return rc;
			if ((rc = launch_pbuild(sa))) return rc;
		}
	}
	if (!closed) launch_bulk_close(sa, nullptr);
	HIPCHK(hipGetLastError());
	return MGL_OK;
}

extern "C" int mgl_sa_run(mgl_sa* sa, uint64_t steps, mgl_sa_stats* stats)
{
	if (!sa) return fail(MGL_EINVAL, "null handle");
	HIPCHK(hipSetDevice(sa->device));
	Control before, after;
	int rc = read_ctl(sa, sa->base, &before);
	if (rc) return rc;
	const bool timing = (sa->cfg.flags & MGL_F_TIMING) != 0;
	/* MGL_F_TIMING brackets steps with events: every step of a short run, every fourth of a longer one (the ten event records of a
	 * timed step cost it 30 us at 10 MB) */
	const uint64_t t_stride = steps > 16 ? 4u : 1u;
	const uint64_t timed_steps = timing ? ((steps + t_stride - 1) / t_stride < 512 ? (steps + t_stride - 1) / t_stride : 512) : 0;
	const bool inc_apply = sa->incremental && sa->incremental_apply;
	const DecideArgs dargs = decide_args(sa);
	const int mode = (sa->incremental && sa->parallel_build) ? sa->accept_mode : MGL_ACCEPT_SINGLE; /* bulk steps rebuild with the parallel builder */
	sa->mode_log.clear();
	sa->la_ready = false;
	if (sa->tgt_ahead) sa->tgt_ahead = 2; /* (only behind a run that ended early: whatever was queued is not trusted) */
	const bool la_ok = sa->la_enabled && inc_apply && sa->split_nbr && sa->ctx.diag_stop == 0;
	const uint64_t rollbacks_before = sa->bulk_rollbacks;
	std::vector<uint8_t> sim_timed; /* per timed step: how many of the regular k_sim launches carry events (0: none, the one-kernel form ran) */
	unsigned long long traffic_before[2] = { 0, 0 };
	if (sa->count_traffic) HIPCHK(hipMemcpy(traffic_before, sa->d_traffic, sizeof traffic_before, hipMemcpyDeviceToHost));
	if (sa->blk_done == 0) { sa->blk_imp0 = before.imp_cands; sa->blk_acc0 = before.accepted; }
	HIPCHK(hipEventRecord(sa->ev_begin, sa->stream));
	for (uint64_t s = 0; s < steps;) {
		const bool bulk = mode == MGL_ACCEPT_BULK || (mode == MGL_ACCEPT_AUTO && sa->bulk_now);
		/* AUTO looks at the device counters between blocks of steps (one small read-back per block) */
		/* every block ends with one small read-back: AUTO needs the counters, and the form of the regular launch
		 * (split / one kernel) follows the device's recommendation from block to block */
		/* AUTO: a block is 16 single / 4 bulk steps wherever the calls' boundaries fall (blk_done steps of it are done) */
		const uint64_t full = mode == MGL_ACCEPT_AUTO ? (bulk ? 4u : 16u) : 64u;
		uint64_t block = mode == MGL_ACCEPT_AUTO ? full - sa->blk_done : full;
		if (sa->short_looks && block > 4u) block = 4u; /* a form was just switched (a trial, usually): look again soon, a trial of a form twice as slow should not last a whole block */
		if (block > steps - s) block = steps - s;
		for (uint64_t e = s + block; s < e; s++) {
			const bool t = s % t_stride == 0 && s / t_stride < timed_steps;
			const uint64_t ti = s / t_stride; /* the step's place among the timed ones */
			if (sa->mode_log.size() < (1u << 20)) sa->mode_log.push_back(bulk ? 1 : 0);
			if (t) HIPCHK(hipEventRecord(pool_event(sa, 4 * ti + 0), sa->stream));
			/* look-ahead: this step's pick + walk may have run beside the previous step's tail, into the other buffer set */
			const bool from_la = sa->la_ready;
			if (from_la) {
				const mgl_sa::NbrSet now = cur_set(sa);
				use_set(sa, sa->alt);
				sa->alt = now;
				sa->la_ready = false;
			}
			sa->time_sim_step = (t && sa->split_nbr && !sa->form_single && !from_la) ? (int64_t)ti : -1;
			if (t) sim_timed.push_back(sa->time_sim_step >= 0 ? (nbr_slices(sa) < 2u ? 1 : 2) : 0);
			rc = launch_neighbours(sa, ~0ull, (s == 0 && !from_la) || !inc_apply, from_la);
			sa->time_sim_step = -1;
			if (rc) return rc;
			if (la_ok && !bulk && !sa->form_single && s + 1 < e && (rc = launch_lookahead(sa, before.gstep + s + 1, from_la))) return rc;
			if (t) HIPCHK(hipEventRecord(pool_event(sa, 4 * ti + 1), sa->stream));
			if (bulk) {
				if (t) HIPCHK(hipEventRecord(pool_event(sa, 4 * ti + 2), sa->stream));
				if ((rc = launch_bulk_tail(sa, s + 1 < steps ? before.gstep + s + 1 : ~0ull))) return rc;
				if (t) HIPCHK(hipEventRecord(pool_event(sa, 4 * ti + 3), sa->stream));
				continue;
			}
			hipLaunchKernelGGL(k_decide, dim3(1), dim3(1024), 0, sa->stream, sa->ctx, sa->base.v, sa->base.ctl, sa->nbr, dargs, inc_apply ? 0 : 1);
			HIPCHK(hipGetLastError());
			if (t) HIPCHK(hipEventRecord(pool_event(sa, 4 * ti + 2), sa->stream));
			if (inc_apply) {
				/* the accept changes the base: the speculative pick + walk of the next step read it until they are through */
				if (sa->la_ready) for (uint32_t h = 0; h < nbr_slices(sa); h++) HIPCHK(hipStreamWaitEvent(sa->stream, sa->ev_spec[h], 0));
				if ((rc = launch_apply(sa, s + 1 < steps ? before.gstep + s + 1 : ~0ull))) return rc;
			} else {
				hipLaunchKernelGGL(k_copy_best, dim3(256), dim3(256), 0, sa->stream, (const Control*)sa->base.ctl,
				                   (const mgl_pk*)sa->base.v.slab, sa->d_best, sa->ctx.n);
				HIPCHK(hipGetLastError());
				if ((rc = rebuild_base(sa, 1))) return rc;
				if (sa->incremental && sa->snapshots && (rc = launch_snapshot(sa, sa->snap_best, 1, 0, 1))) return rc;
			}
			if (t) HIPCHK(hipEventRecord(pool_event(sa, 4 * ti + 3), sa->stream));
		}
		if (mode == MGL_ACCEPT_AUTO) sa->blk_done += block;
		const bool block_over = mode == MGL_ACCEPT_AUTO && sa->blk_done == full;
		if (block_over || (s < steps && sa->adaptive)) {
			Control now;
			if ((rc = read_ctl(sa, sa->base, &now))) return rc;
			if (now.error_flags) break;
			if (sa->short_looks) sa->short_looks--;
			if (sa->adaptive && sa->form_single != (now.nbr_single != 0)) { sa->form_single = now.nbr_single != 0; sa->short_looks = 3; }
			if (block_over) {
				auto_decide(sa, bulk, full, now.imp_cands - sa->blk_imp0, now.accepted - sa->blk_acc0);
				sa->blk_done = 0; sa->blk_imp0 = now.imp_cands; sa->blk_acc0 = now.accepted;
			}
		}
	}
	HIPCHK(hipEventRecord(sa->ev_end, sa->stream));
	HIPCHK(hipStreamSynchronize(sa->stream));
	if ((rc = read_ctl(sa, sa->base, &after))) return rc;
	if (sa->adaptive && sa->form_single != (after.nbr_single != 0)) { sa->form_single = after.nbr_single != 0; sa->short_looks = 3; }
	if (stats) {
		memset(stats, 0, sizeof *stats);
		stats->steps = after.gstep - before.gstep;
		stats->evaluations = after.evals - before.evals;
		stats->failed = after.failed - before.failed;
		stats->accepted = after.accepted - before.accepted;
		stats->improved = after.improved - before.improved;
		stats->packets_evaluated = after.packets_eval - before.packets_eval;
		stats->current_cost = after.cur_cost;
		stats->best_cost = after.best_cost;
		stats->packets = after.packets;
		float ms = 0;
		HIPCHK(hipEventElapsedTime(&ms, sa->ev_begin, sa->ev_end));
		stats->gpu_ms_total = ms;
		for (uint64_t s = 0; s < timed_steps; s++) {
			HIPCHK(hipEventElapsedTime(&ms, sa->ev_pool[4 * s + 0], sa->ev_pool[4 * s + 1]));
			stats->gpu_ms_neighbours += ms;
			HIPCHK(hipEventElapsedTime(&ms, sa->ev_pool[4 * s + 2], sa->ev_pool[4 * s + 3]));
			stats->gpu_ms_rebuild += ms;
		}
		stats->neighbour_launches = timed_steps;
		for (uint64_t s = 0; s < timed_steps && s < sim_timed.size(); s++) {
			if (!sim_timed[s]) continue;
			for (uint32_t h = 0; h < 3; h++) { /* the regular launches (one per slice, at most two timed) and the second pass's list */
				if (h < 2 && h >= sim_timed[s]) continue;
				HIPCHK(hipEventElapsedTime(&ms, sa->ev_sim_pool[6 * s + 2 * h], sa->ev_sim_pool[6 * s + 2 * h + 1]));
				stats->gpu_ms_sim += ms;
				stats->sim_launches++;
			}
		}
		if (sa->count_traffic) {
			unsigned long long now[2] = { 0, 0 };
			HIPCHK(hipMemcpy(now, sa->d_traffic, sizeof now, hipMemcpyDeviceToHost));
			stats->sim_bytes_counted = now[0] - traffic_before[0];
		}
		stats->full_rebuilds = after.full_rebuilds - before.full_rebuilds;
		stats->fallback_neighbours = after.fallback_nbrs - before.fallback_nbrs;
		stats->second_pass_neighbours = after.big_nbrs - before.big_nbrs;
		stats->bulk_steps = after.bulk_steps - before.bulk_steps;
		stats->dropped_neighbours = after.dropped - before.dropped;
		stats->improving_neighbours = after.imp_cands - before.imp_cands;
		stats->bulk_rollbacks = sa->bulk_rollbacks - rollbacks_before;
		stats->bulk_double_writes = after.bulk_overlaps - before.bulk_overlaps;
	}
	if (after.error_flags) {
		char buf[96];
		snprintf(buf, sizeof buf, "mgl_sa_run: device consistency check failed (flags 0x%x)", after.error_flags);
		return fail(MGL_EDEVICE, buf);
	}
	return MGL_OK;
}

extern "C" int mgl_sa_current(mgl_sa* sa, mgl_packet* packets_out, uint64_t* perplexity_out)
{
	if (!sa) return fail(MGL_EINVAL, "null handle");
	HIPCHK(hipSetDevice(sa->device));
	Control c;
	int rc = read_ctl(sa, sa->base, &c);
	if (rc) return rc;
	if (perplexity_out) *perplexity_out = c.rebuild_cost;
	if (packets_out) return export_slab(sa, sa->base.v.slab, packets_out);
	return MGL_OK;
}
extern "C" int mgl_sa_best(mgl_sa* sa, mgl_packet* packets_out, uint64_t* perplexity_out)
{
	if (!sa) return fail(MGL_EINVAL, "null handle");
	HIPCHK(hipSetDevice(sa->device));
	Control c;
	int rc = read_ctl(sa, sa->base, &c);
	if (rc) return rc;
	if (perplexity_out) *perplexity_out = c.best_cost;
	if (packets_out) return export_slab(sa, (sa->incremental && sa->snapshots && c.best_is_current) ? sa->base.v.slab : sa->d_best, packets_out);
	return MGL_OK;
}

/* run the walk kernel over `packets` in the scratch base */
static int scratch_walk(mgl_sa* sa, const mgl_packet* packets, bool want_cum, bool want_probs, Control* out)
{
	int rc = import_slab(sa, packets, sa->scratch.v.slab);
	if (rc) return rc;
	HIPCHK(hipMemsetAsync(sa->scratch.ctl, 0, sizeof(Control), sa->stream));
	if ((rc = launch_rebuild(sa, sa->scratch, 0, want_cum ? sa->d_cum : nullptr, want_probs ? sa->d_final_probs : nullptr))) return rc;
	if ((rc = read_ctl(sa, sa->scratch, out))) return rc;
	if (out->error_flags) return fail(MGL_EINVAL, "slab is not a valid parse of the input");
	return MGL_OK;
}

extern "C" int mgl_sa_set_best(mgl_sa* sa, const mgl_packet* packets, uint64_t perplexity)
{
	if (!sa || !packets) return fail(MGL_EINVAL, "null argument");
	HIPCHK(hipSetDevice(sa->device));
	Control c;
	int rc = scratch_walk(sa, packets, false, false, &c);
	if (rc) return rc;
	if (c.rebuild_cost != perplexity) return fail(MGL_EINVAL, "mgl_sa_set_best: perplexity does not match the slab");
	HIPCHK(hipMemcpyAsync(sa->d_best, sa->scratch.v.slab, sizeof(mgl_pk) * (size_t)sa->n, hipMemcpyDeviceToDevice, sa->stream));
	if (sa->d_snap_meta) HIPCHK(hipMemsetAsync(sa->d_snap_meta + 1, 0, sizeof(SnapMeta), sa->stream));
	if ((rc = read_ctl(sa, sa->base, &c))) return rc;
	c.best_cost = perplexity;
	c.best_is_current = 0;
	return write_ctl(sa, sa->base, &c);
}

extern "C" int mgl_cost_slab(mgl_sa* sa, const mgl_packet* packets, uint64_t* total, uint64_t* per_packet_cumulative,
                             size_t* npackets)
{
	if (!sa || !packets) return fail(MGL_EINVAL, "null argument");
	HIPCHK(hipSetDevice(sa->device));
	Control c;
	int rc = scratch_walk(sa, packets, per_packet_cumulative != nullptr, false, &c);
	if (rc) return rc;
	if (total) *total = c.rebuild_cost;
	if (npackets) *npackets = (size_t)c.packets;
	if (per_packet_cumulative)
		HIPCHK(hipMemcpy(per_packet_cumulative, sa->d_cum, sizeof(uint64_t) * (size_t)c.packets, hipMemcpyDeviceToHost));
	return MGL_OK;
}

extern "C" int mgl_final_state(mgl_sa* sa, const mgl_packet* packets, uint16_t* probs_out, size_t probs_cap,
                               uint8_t* ctx_state_out, uint32_t dists_out[4])
{
	if (!sa || !packets) return fail(MGL_EINVAL, "null argument");
	HIPCHK(hipSetDevice(sa->device));
	Control c;
	int rc = scratch_walk(sa, packets, false, true, &c);
	if (rc) return rc;
	if (ctx_state_out) *ctx_state_out = (uint8_t)c.final_ctx_state;
	if (dists_out) memcpy(dists_out, c.final_dists, sizeof c.final_dists);
	if (probs_out) {
		const uint32_t total = sa->ctx.L.total, lit = total - MGL_OFF_LIT;
		if (probs_cap < total) return fail(MGL_ERANGE, "probs_out too small");
		std::vector<uint16_t> mine(total);
		HIPCHK(hipMemcpy(mine.data(), sa->d_final_probs, sizeof(uint16_t) * total, hipMemcpyDeviceToHost));
		/* reference struct order (lzma_state.h:47-53): lit | len | rep_len | dist | ctx_state */
		uint16_t* o = probs_out;
		memcpy(o, &mine[MGL_OFF_LIT], 2 * lit); o += lit;
		memcpy(o, &mine[MGL_OFF_LEN], 2 * 514); o += 514;
		memcpy(o, &mine[MGL_OFF_REP_LEN], 2 * 514); o += 514;
		memcpy(o, &mine[MGL_OFF_DIST], 2 * 387); o += 387;
		memcpy(o, &mine[0], 2 * 432);
	}
	return MGL_OK;
}

extern "C" int mgl_top_k(mgl_sa* sa, const mgl_packet* packets, size_t position, mgl_packet* out, uint64_t* costs, size_t* count)
{
	if (!sa || !packets || !out || !costs || !count) return fail(MGL_EINVAL, "null argument");
	if (position >= sa->n) return fail(MGL_ERANGE, "position outside the input");
	HIPCHK(hipSetDevice(sa->device));
	Control c;
	int rc = scratch_walk(sa, packets, false, false, &c);
	if (rc) return rc;
	hipLaunchKernelGGL(k_topk_probe, dim3(1), dim3(64), sa->walk_lds, sa->stream, sa->ctx, sa->scratch.v, (uint32_t)position,
	                   sa->d_topk_pk, sa->d_topk_cost, sa->d_small);
	HIPCHK(hipGetLastError());
	uint32_t cnt = 0;
	mgl_pk pk[64];
	uint64_t cs[64];
	HIPCHK(hipMemcpyAsync(&cnt, sa->d_small, sizeof cnt, hipMemcpyDeviceToHost, sa->stream));
	HIPCHK(hipMemcpyAsync(pk, sa->d_topk_pk, sizeof pk, hipMemcpyDeviceToHost, sa->stream));
	HIPCHK(hipMemcpyAsync(cs, sa->d_topk_cost, sizeof cs, hipMemcpyDeviceToHost, sa->stream));
	HIPCHK(hipStreamSynchronize(sa->stream));
	if (cnt == ~0u) return fail(MGL_ERANGE, "position is not on the slab's walk");
	for (uint32_t i = 0; i < cnt; i++) {
		out[i].type = (uint8_t)mgl_pk_type(pk[i]); out[i].dist = mgl_pk_dist(pk[i]); out[i].len = (uint16_t)mgl_pk_len(pk[i]);
		costs[i] = cs[i];
	}
	*count = cnt;
	return MGL_OK;
}

extern "C" int mgl_substrings(mgl_sa* sa, size_t pos, size_t max_len, uint32_t* offsets, uint32_t* lengths, size_t cap, size_t* count)
{
	if (!sa || !count) return fail(MGL_EINVAL, "null argument");
	if (pos >= sa->n) return fail(MGL_ERANGE, "position outside the input");
	HIPCHK(hipSetDevice(sa->device));
	const uint32_t dcap = cap < sa->sub_cap ? (uint32_t)cap : sa->sub_cap;
	DevCtx c = sa->ctx;
	c.dict_limit = 0xFFFFFFFFu; /* the index query itself has no window (substring_enumerator.c:97 is a todo) */
	hipLaunchKernelGGL(k_substrings, dim3(1), dim3(1), 0, sa->stream, c, (uint32_t)pos, (uint32_t)max_len, sa->d_sub_offs,
	                   sa->d_sub_lens, dcap, sa->d_small);
	HIPCHK(hipGetLastError());
	uint32_t cnt = 0;
	HIPCHK(hipMemcpyAsync(&cnt, sa->d_small, sizeof cnt, hipMemcpyDeviceToHost, sa->stream));
	HIPCHK(hipStreamSynchronize(sa->stream));
	const uint32_t take = cnt < dcap ? cnt : dcap;
	if (offsets && take) HIPCHK(hipMemcpy(offsets, sa->d_sub_offs, sizeof(uint32_t) * take, hipMemcpyDeviceToHost));
	if (lengths && take) HIPCHK(hipMemcpy(lengths, sa->d_sub_lens, sizeof(uint32_t) * take, hipMemcpyDeviceToHost));
	*count = cnt;
	return MGL_OK;
}

extern "C" int mgl_neighbours(mgl_sa* sa, uint64_t global_step, uint64_t* costs, mgl_diff* diffs, uint32_t* ndiffs, size_t diff_cap)
{
	if (!sa || !costs) return fail(MGL_EINVAL, "null argument");
	HIPCHK(hipSetDevice(sa->device));
	int rc = launch_neighbours(sa, global_step);
	if (rc == MGL_OK && sa->d_counts) HIPCHK(hipMemcpyAsync(sa->d_counts + 8, sa->d_counts, sizeof(uint32_t) * 8, hipMemcpyDeviceToDevice, sa->stream));
	if (rc) return rc;
	const size_t K = sa->cfg.neighbours_per_step;
	HIPCHK(hipMemcpyAsync(costs, sa->nbr.cost, sizeof(uint64_t) * K, hipMemcpyDeviceToHost, sa->stream));
	std::vector<uint32_t> nd(K), dpos;
	std::vector<mgl_pk> dold, dnew;
	HIPCHK(hipMemcpyAsync(nd.data(), sa->nbr.ndiffs, sizeof(uint32_t) * K, hipMemcpyDeviceToHost, sa->stream));
	if (diffs) {
		dpos.resize(K * MGL_MAX_DIFFS); dold.resize(K * MGL_MAX_DIFFS); dnew.resize(K * MGL_MAX_DIFFS);
		HIPCHK(hipMemcpyAsync(dpos.data(), sa->nbr.dpos, sizeof(uint32_t) * dpos.size(), hipMemcpyDeviceToHost, sa->stream));
		HIPCHK(hipMemcpyAsync(dold.data(), sa->nbr.dold, sizeof(mgl_pk) * dold.size(), hipMemcpyDeviceToHost, sa->stream));
		HIPCHK(hipMemcpyAsync(dnew.data(), sa->nbr.dnew, sizeof(mgl_pk) * dnew.size(), hipMemcpyDeviceToHost, sa->stream));
	}
	HIPCHK(hipStreamSynchronize(sa->stream));
	for (size_t j = 0; j < K; j++) {
		if (ndiffs) ndiffs[j] = nd[j];
		if (!diffs) continue;
		for (size_t e = 0; e < nd[j] && e < diff_cap; e++) {
			mgl_diff* d = &diffs[j * diff_cap + e];
			const mgl_pk o = dold[j * MGL_MAX_DIFFS + e], w = dnew[j * MGL_MAX_DIFFS + e];
			d->position = dpos[j * MGL_MAX_DIFFS + e];
			d->old_packet.type = (uint8_t)mgl_pk_type(o); d->old_packet.dist = mgl_pk_dist(o); d->old_packet.len = (uint16_t)mgl_pk_len(o);
			d->new_packet.type = (uint8_t)mgl_pk_type(w); d->new_packet.dist = mgl_pk_dist(w); d->new_packet.len = (uint16_t)mgl_pk_len(w);
		}
	}
	return MGL_OK;
}

/* test hook: raw copies of the incremental path's base structures */
extern "C" int mgl_debug_dump(mgl_sa* sa, uint32_t what, void* out, size_t cap_bytes, size_t* bytes)
{
	if (!sa || !out || !bytes) return fail(MGL_EINVAL, "null argument");
	if (!sa->incremental && what != 21 && what != 22) return fail(MGL_EINVAL, "mgl_debug_dump: handle runs the full-walk engine");
	HIPCHK(hipSetDevice(sa->device));
	HIPCHK(hipStreamSynchronize(sa->stream));
	const Base2& b = sa->b2;
	const void* src = nullptr;
	size_t sz = 0;
	uint32_t top = 0;
	if (sa->incremental) HIPCHK(hipMemcpy(&top, b.pool_top, sizeof top, hipMemcpyDeviceToHost));
	switch (what) {
	case 0: src = b.ch_off; sz = sizeof(uint32_t) * sa->ctx.L.total; break;
	case 1: src = b.ch_len; sz = sizeof(uint32_t) * sa->ctx.L.total; break;
	case 2: src = b.ch_pos; sz = sizeof(uint32_t) * (size_t)top; break;
	case 3: src = b.ch_ev; sz = sizeof(uint16_t) * (size_t)top; break;
	case 4: src = b.onwalk; sz = sizeof(uint64_t) * b.nw0; break;
	case 5: src = b.sp0; sz = sizeof(uint64_t) * b.nw0; break;
	case 6: src = b.sp_state; sz = sizeof(uint32_t) * 8 * (size_t)sa->n; break;
	case 7: src = b.ck_probs; sz = sizeof(uint16_t) * (size_t)b.nck * b.ck_elems; break;
	case 8: src = b.ch_cap; sz = sizeof(uint32_t) * sa->ctx.L.total; break;
	case 9: src = sa->d_prof; sz = sa->d_prof ? sizeof(unsigned long long) * (32 + sa->cfg.neighbours_per_step) : 0; break;
	case 80: { /* host counters: batch accepts, batch accepts that fell back to the rebuild */
		const uint64_t v[2] = { sa->batch_accepts, sa->batch_fallbacks };
		*bytes = sizeof v;
		if (cap_bytes >= sizeof v) memcpy(out, v, sizeof v);
		return MGL_OK;
	}
	case 83: src = b.ch_sb; sz = sizeof(uint32_t) * (size_t)b.ck_elems * b.sb_stride; break; /* the chain index, row per context, sb_stride words each */
	case 81: src = sa->batch.hdr; sz = sizeof(uint32_t) * 16; break;
	case 82: src = sa->batch.acc; sz = sizeof(long long) * 4; break;
	case 10: src = sa->d_counts + 8; sz = sizeof(uint32_t) * 4; break; /* of the last finished step / mgl_neighbours call */
	case 14: src = sa->ab.hdr; sz = sa->ab.hdr ? sizeof(uint32_t) * 16 : 0; break; /* apply counters / stage cycles */
	case 16: src = sa->base.ctl; sz = sizeof(Control); break; /* raw control block */
	case 15: src = sa->d_pickrec; sz = sa->d_pickrec ? sizeof(uint4) * sa->cfg.neighbours_per_step : 0; break;
	case 18: src = sa->d_quad_pos; sz = sizeof(uint32_t) * (sa->n - 1); break;
	case 19: src = sa->d_quad_nx; sz = sizeof(uint16_t) * (sa->n - 1); break;
	case 12: src = sa->d_bucket_off; sz = sizeof(uint32_t) * 65537; break;
	case 13: src = sa->d_bucket_pos; sz = sizeof(uint32_t) * (sa->n - 1); break;
	case 30: case 31: case 32: case 33: case 34: case 35: src = sa->d_xpos[what - 30]; sz = sizeof(uint32_t) * (sa->n - 1); break; /* exact-length orders D = what - 28 */
	case 40: case 41: case 42: case 43: case 44: case 45: src = sa->d_xrank[what - 40]; sz = src ? sizeof(uint32_t) * (sa->n - 1) : 0; break;
	case 50: case 51: case 52: case 53: case 54: case 55: src = sa->d_xrun[what - 50]; sz = src ? sizeof(uint32_t) * (sa->n - 1) : 0; break;
	case 60: case 61: case 62: case 63: case 64: case 65: src = sa->d_xnxb[what - 60]; sz = sa->n - 1; break;
	case 70: src = sa->d_oct_pos; sz = sizeof(uint32_t) * (sa->n - 1); break;
	case 71: src = sa->d_oct_rank; sz = sizeof(uint32_t) * (sa->n - 1); break;
	case 72: src = sa->d_oct_run; sz = sizeof(uint32_t) * (sa->n - 1); break;
	case 73: src = sa->d_oct_nx8; sz = sizeof(uint64_t) * (sa->n - 1); break;
	case 74: src = sa->d_hex_pos; sz = sizeof(uint32_t) * (sa->n - 1); break;
	case 75: src = sa->d_hex_rank; sz = sizeof(uint32_t) * (sa->n - 1); break;
	case 76: src = sa->d_hex_run; sz = sizeof(uint32_t) * (sa->n - 1); break;
	case 77: src = sa->d_hex_nx8; sz = sizeof(uint64_t) * (sa->n - 1); break;
	case 22: src = sa->nbr.win2; sz = sizeof(uint32_t) * sa->cfg.neighbours_per_step; break; /* soft ends | dep << 31 */
	case 21: src = sa->nbr.win; sz = sizeof(uint32_t) * 2 * sa->cfg.neighbours_per_step; break; /* windows of the last costed neighbours */
	case 11: src = sa->pb.acc; sz = sa->pb.acc ? sizeof(unsigned long long) * 8 : 0; break; /* parallel builder totals */
	default: return fail(MGL_EINVAL, "mgl_debug_dump: unknown selector");
	}
	*bytes = sz;
	if (sz > cap_bytes) return fail(MGL_ERANGE, "mgl_debug_dump: buffer too small");
	HIPCHK(hipMemcpy(out, src, sz, hipMemcpyDeviceToHost));
	return MGL_OK;
}

/* diagnostic knobs (tools/, tests/): key 0 = stop the neighbour kernel after phase `value`; key 1 = see below */
extern "C" int mgl_debug_set(mgl_sa* sa, uint32_t key, uint64_t value)
{
	if (sa && key == 5) { sa->force_batch_fail = (uint32_t)value; return MGL_OK; } /* the next `value` batch accepts give up behind their commit (exercises the fallback to the rebuild) */
	if (sa && key == 4) { /* count the bytes of chain data the re-simulation kernel reads (mgl_sa_stats.sim_bytes_counted) */
		if (!sa->d_traffic) return fail(MGL_EINVAL, "mgl_debug_set: no split neighbour evaluation on this handle");
		sa->count_traffic = value != 0;
		return MGL_OK;
	}
	if (!sa) return fail(MGL_EINVAL, "null handle");
	if (key == 0) { sa->ctx.diag_stop = (uint32_t)value; return MGL_OK; }
	if (key == 1) { sa->pb.force_fix = (uint32_t)value; return MGL_OK; } /* parallel builder: redo every chain segment serially */
	if (key == 2) { /* first-pass list capacity, below what was allocated: pushes neighbours into the second pass */
		if (value < 8 || value > sa->chg_cap || (value & 7u)) return fail(MGL_EINVAL, "mgl_debug_set: list capacity must be a multiple of 8 within the allocated one");
		sa->big.chg_cap = (uint32_t)value;
		return MGL_OK;
	}
	if (key == 3) { sa->force_rollbacks = (uint32_t)value; return MGL_OK; } /* the next `value` bulk steps that take moves are taken back as if their parse had failed validation */
	return fail(MGL_EINVAL, "mgl_debug_set: unknown key");
}

#include "mgl_exchange.inc"
