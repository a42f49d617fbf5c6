/*
 * mgl_index.hip -- the match index built on the device.
 *
 * substring_enumerator.c:26-47 buckets every position by its leading bigram and keeps the
 * positions of a bucket ascending.  Here: a stable two-pass counting sort of the positions
 * 0 .. n-2, first by data[pos + 1], then by data[pos] (LSD radix, 8 bits per pass).  A pass is
 * three kernels over blocks of 4096 items: per-block digit histogram, one exclusive scan of the
 * digit-major histogram matrix, and a scatter in which a wavefront ranks its 64 items among equal
 * digits with ballots, so equal keys keep their position order.  bucket_off is then read off the
 * sorted keys.  The dictionary window (dict_limit) is applied at query time by a search inside
 * the bucket (bucket_lower_bound, mgl_kernels.hip), so the index itself is window-independent.
 * A second order of the same positions -- by the four bytes at the position, then by position
 * (two more passes in front) -- groups, inside every bigram bucket, the entries that agree on
 * the next two bytes as well: a top-K query reads its >= 4-byte matches as one contiguous,
 * position-ordered run of that array instead of finding them among all the bucket's entries.
 */
#include "mgl_device.h"

#define MGL_IX_ITEMS 4096u /* per wavefront */

/* `pass` = byte offset from the position that this counting pass sorts by; `in` = the order produced
 * by the previous pass (nullptr: positions in their natural order) */
__device__ __forceinline__ uint32_t ix_digit(const uint8_t* data, const uint32_t* in, uint32_t idx, int pass, uint32_t& pos)
{
	pos = in ? in[idx] : idx;
	return data[pos + (uint32_t)pass];
}

__global__ void __launch_bounds__(64) ix_count(const uint8_t* data, const uint32_t* in, uint32_t m, int pass, uint32_t* matrix, uint32_t nblk)
{
	__shared__ uint32_t cnt[256];
	const uint32_t lane = threadIdx.x, blk = blockIdx.x;
	for (uint32_t i = lane; i < 256; i += 64) cnt[i] = 0;
	__syncthreads();
	const uint32_t lo = blk * MGL_IX_ITEMS, hi = (lo + MGL_IX_ITEMS) < m ? (lo + MGL_IX_ITEMS) : m;
	for (uint32_t idx = lo + lane; idx < hi; idx += 64) {
		uint32_t pos;
		atomicAdd(&cnt[ix_digit(data, in, idx, pass, pos)], 1u);
	}
	__syncthreads();
	for (uint32_t i = lane; i < 256; i += 64) matrix[(size_t)i * nblk + blk] = cnt[i];
}

/* in-place exclusive scan of `count` words by one workgroup */
__global__ void __launch_bounds__(1024) ix_scan(uint32_t* a, uint32_t count)
{
	__shared__ uint32_t wsum[16];
	__shared__ uint32_t carry_s;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
	if (tid == 0) carry_s = 0;
	__syncthreads();
	for (uint32_t base = 0; base < count; base += 4096) {
		const uint32_t i0 = base + tid * 4;
		uint32_t v[4], s = 0;
#pragma unroll
		for (int k = 0; k < 4; k++) { v[k] = (i0 + k) < count ? a[i0 + k] : 0u; s += v[k]; }
		uint32_t incl = s;
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
			if ((int)lane >= o) incl += t;
		}
		if (lane == 63) wsum[wv] = incl;
		__syncthreads();
		uint32_t before = carry_s;
		for (uint32_t w = 0; w < wv; w++) before += wsum[w];
		uint32_t run = before + incl - s;
#pragma unroll
		for (int k = 0; k < 4; k++) { if ((i0 + k) < count) a[i0 + k] = run; run += v[k]; }
		__syncthreads();
		if (tid == 1023) carry_s = run;
		__syncthreads();
	}
}

__global__ void __launch_bounds__(64) ix_scatter(const uint8_t* data, const uint32_t* in, uint32_t* out, uint32_t m, int pass,
                                                 const uint32_t* matrix, uint32_t nblk)
{
	__shared__ uint32_t cnt[256];
	const uint32_t lane = threadIdx.x, blk = blockIdx.x;
	for (uint32_t i = lane; i < 256; i += 64) cnt[i] = matrix[(size_t)i * nblk + blk];
	wave_sync();
	const uint32_t lo = blk * MGL_IX_ITEMS;
	const unsigned long long below = (1ull << lane) - 1ull;
	for (uint32_t it = 0; it < MGL_IX_ITEMS / 64u; it++) {
		const uint32_t idx = lo + it * 64u + lane;
		const bool valid = idx < m;
		if (!__any(valid)) break;
		uint32_t pos = 0;
		const uint32_t digit = valid ? ix_digit(data, in, idx, pass, pos) : 0u;
		unsigned long long peers = __ballot(valid);
#pragma unroll
		for (uint32_t k = 0; k < 8; k++) {
			const bool bit = (digit >> k) & 1u;
			const unsigned long long bm = __ballot(bit);
			peers &= bit ? bm : ~bm;
		}
		const uint32_t rank = (uint32_t)__popcll(peers & below);
		const uint32_t base = valid ? cnt[digit] : 0u;
		wave_sync();
		if (valid && rank == 0) cnt[digit] = base + (uint32_t)__popcll(peers);
		wave_sync();
		if (valid) out[base + rank] = pos;
	}
}

/* bucket_off[key] = first sorted index whose key is >= key; bucket_off[65536] = m */
__global__ void __launch_bounds__(256) ix_offsets(const uint8_t* data, const uint32_t* sorted, uint32_t m, uint32_t* bucket_off)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j > m) return;
	const uint32_t key = j < m ? (((uint32_t)data[sorted[j]] << 8) | data[sorted[j] + 1]) : 65536u;
	const uint32_t prev = j == 0 ? 0u : ((((uint32_t)data[sorted[j - 1]] << 8) | data[sorted[j - 1] + 1]) + 1u);
	for (uint32_t k = prev; k <= key; k++) bucket_off[k] = j; /* empty buckets in between start here too */
}

/* the two bytes after each indexed bigram, in the order of `sorted` (the input is zero padded past its
 * end).  big_endian = 0: data[p+2] | data[p+3] << 8 (bucket_nx, only compared for equality);
 * big_endian = 1: data[p+2] << 8 | data[p+3] (quad_nx: ascending inside a bucket, searchable) */
__global__ void __launch_bounds__(256) ix_next2(const uint8_t* data, const uint32_t* sorted, uint32_t m, uint16_t* nx, int big_endian)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= m) return;
	const uint32_t p = sorted[j];
	const uint32_t a = data[p + 2], b = data[p + 3];
	nx[j] = (uint16_t)(big_endian ? ((a << 8) | b) : (a | (b << 8)));
}

/* ---- the deeper orders: rank of every position, start of its run of equal prefixes, the bytes behind the prefix */
__global__ void __launch_bounds__(256) ix_rank(const uint32_t* sorted, uint32_t m, uint32_t* rank)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < m) rank[sorted[j]] = j;
}
/* nx[j] = the `bytes` (4 or 8) input bytes at sorted[j] + off, little endian (the input is zero padded past its end) */
__global__ void __launch_bounds__(256) ix_next_bytes(const uint8_t* data, const uint32_t* sorted, uint32_t m, void* nx, uint32_t off, uint32_t bytes)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= m) return;
	const uint8_t* p = data + sorted[j] + off;
	if (bytes == 4) { uint32_t v; __builtin_memcpy(&v, p, 4); ((uint32_t*)nx)[j] = v; }
	else { uint64_t v; __builtin_memcpy(&v, p, 8); ((uint64_t*)nx)[j] = v; }
}
/* nxb[j] = input byte at sorted[j] + off */
__global__ void __launch_bounds__(256) ix_next_byte(const uint8_t* data, const uint32_t* sorted, uint32_t m, uint8_t* nxb, uint32_t off)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < m) nxb[j] = data[sorted[j] + off];
}
/* run[j] = j where entry j starts a run (its first D bytes differ from entry j - 1's), else 0 ... */
__global__ void __launch_bounds__(256) ix_heads(const uint8_t* data, const uint32_t* sorted, uint32_t m, uint32_t D, uint32_t* run)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= m) return;
	bool head = j == 0;
	if (!head) {
		const uint8_t *a = data + sorted[j], *b = data + sorted[j - 1];
		for (uint32_t i = 0; i < D; i++) head |= a[i] != b[i];
	}
	run[j] = head ? j : 0u;
}
/* ... and an inclusive running maximum turns that into "start of my run" (one workgroup, in place) */
__global__ void __launch_bounds__(1024) ix_maxscan(uint32_t* a, uint32_t count)
{
	__shared__ uint32_t wmax[16];
	__shared__ uint32_t carry_s;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
	if (tid == 0) carry_s = 0;
	__syncthreads();
	for (uint32_t base = 0; base < count; base += 4096) {
		const uint32_t i0 = base + tid * 4;
		uint32_t v[4], s = 0;
#pragma unroll
		for (int k = 0; k < 4; k++) { v[k] = (i0 + k) < count ? a[i0 + k] : 0u; s = v[k] > s ? v[k] : s; v[k] = s; }
		uint32_t incl = s;
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
			if ((int)lane >= o) incl = t > incl ? t : incl;
		}
		if (lane == 63) wmax[wv] = incl;
		__syncthreads();
		uint32_t before = carry_s;
		for (uint32_t w = 0; w < wv; w++) before = wmax[w] > before ? wmax[w] : before;
		const uint32_t prev = (uint32_t)__shfl_up((int)incl, 1, 64);
		if (lane > 0) before = prev > before ? prev : before;
#pragma unroll
		for (int k = 0; k < 4; k++) if ((i0 + k) < count) a[i0 + k] = v[k] > before ? v[k] : before;
		__syncthreads();
		if (tid == 1023) carry_s = incl > before ? incl : before;
		__syncthreads();
	}
}

/* ================================================================== greedy seed
 *
 * The reference starts every search from the all-literal slab (main.c:71).  A packet slab holds
 * a packet at EVERY byte position (packet_slab.h:5) and a parse is whatever the walk
 * pos += len meets, so a greedy LZ parse needs no serial pass at all: every position gets,
 * independently, the longest match the index offers there, and the walk from 0 picks the greedy
 * parse out of them (the entries it jumps over are the alternatives the search tries first when
 * the alignment shifts).  Not in the reference: an opt-in starting point, nothing on the costing
 * path changes.
 *
 * Rule (restated in tests/test_gpu_greedy.py): candidates at p (0 < p < n-1) are the `cand`
 * nearest earlier positions with the same two bytes and the `cand` nearest earlier positions with
 * the same four bytes, inside the dictionary window; the longest match wins, the nearest among
 * equally long ones; it is kept if len >= 4, or len == 3 and distance <= 2^14, or len == 2 and
 * distance <= 128; and dropped again if the next position would take a longer match (the lazy
 * step of LZ77 coders, decided per position from the same rule); otherwise the position holds a
 * literal. */
__device__ __forceinline__ uint32_t gs_lower_u32(const uint32_t* a, uint32_t lo, uint32_t hi, uint32_t x)
{
	while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (a[mid] < x) lo = mid + 1; else hi = mid; }
	return lo;
}
__device__ __forceinline__ uint32_t gs_lower_u16(const uint16_t* a, uint32_t lo, uint32_t hi, uint32_t x)
{
	while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (a[mid] < x) lo = mid + 1; else hi = mid; }
	return lo;
}
/* longest match at p among the candidates (0 if none); *q_out = its source position */
__device__ __forceinline__ uint32_t gs_best(const DevCtx& c, uint32_t p, uint32_t cand, uint32_t* q_out)
{
	const uint8_t* d = c.data;
	const uint32_t maxlen = (c.n - p) < MGL_MAX_MATCH ? (c.n - p) : MGL_MAX_MATCH;
	const uint32_t bigram = ((uint32_t)d[p] << 8) | d[p + 1];
	const uint32_t b_lo = c.bucket_off[bigram], b_end = c.bucket_off[bigram + 1];
	uint32_t best_len = 0, best_q = 0;
	for (int src = 0; src < 2; src++) {
		const uint32_t* pos_arr;
		uint32_t lo, hi, from;
		if (src == 0) {
			pos_arr = c.bucket_pos; lo = b_lo;
			hi = gs_lower_u32(pos_arr, b_lo, b_end, p);
			from = 2;
		} else {
			if (maxlen < 4) break;
			const uint32_t x2 = ((uint32_t)d[p + 2] << 8) | d[p + 3];
			const uint32_t qa = gs_lower_u16(c.quad_nx, b_lo, b_end, x2);
			const uint32_t qb = gs_lower_u16(c.quad_nx, qa, b_end, x2 + 1u);
			pos_arr = c.quad_pos; lo = qa;
			hi = gs_lower_u32(pos_arr, qa, qb, p);
			from = 4;
		}
		for (uint32_t i = hi, taken = 0; i > lo && taken < cand; taken++) {
			const uint32_t q = pos_arr[--i];
			if (p - q - 1u >= c.dict_limit) break; /* ascending positions: everything further is outside too */
			uint32_t len = from;
			while (len < maxlen && d[q + len] == d[p + len]) len++;
			if (len > best_len || (len == best_len && q > best_q)) { best_len = len; best_q = q; }
		}
	}
	const uint32_t dist = p - best_q; /* real distance */
	if (!(best_len >= 4 || (best_len == 3 && dist <= (1u << 14)) || (best_len == 2 && dist <= 128u))) best_len = 0;
	*q_out = best_q;
	return best_len;
}
__global__ void __launch_bounds__(256) k_greedy_seed(DevCtx c, mgl_pk* slab, uint32_t cand)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= c.n) return;
	mgl_pk pk = MGL_PK_LITERAL;
	if (p > 0 && p + 1 < c.n) {
		uint32_t q = 0, q1 = 0;
		const uint32_t len = gs_best(c, p, cand, &q);
		/* lazy step: a literal here if the next position starts a longer match */
		const uint32_t len1 = (len != 0 && p + 2 < c.n) ? gs_best(c, p + 1, cand, &q1) : 0u;
		if (len != 0 && len1 <= len) pk = mgl_pack(MGL_MATCH, p - q - 1u, len);
	}
	slab[p] = pk;
}
