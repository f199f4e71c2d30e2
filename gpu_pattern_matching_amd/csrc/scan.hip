// The scan pipeline: speculative chain walk -> boundary resolve -> exclusive
// scan of per-chain counts -> ordered scatter.  gfx950 only.
//
// Replaces ahomatch.cl:1-165 (+ its launch, ocl_aho_match.c:96-131) and the
// prefix-sum/compaction stage behind it (databuf.c:648-651), with the serial
// semantics of SURVEY App. B.1: one record per text position whose transition
// enters a final state, each position exactly once, in position order.
//
// How a serial DFA walk is made parallel without a per-lane warm-up:
//
//   The text is cut into chains of S bytes (S = 16..256, a power of two).
//
//   K1 spec_walk   every lane walks its chain(s) from the ROOT state.  The
//                  state reached after m bytes equals the true serial state
//                  as soon as the true state's depth is <= m (the state only
//                  remembers its last depth bytes); from then on the two
//                  walks coincide.  K1 records the end state e[j], the
//                  tentative hits (staged per wave), their number and the
//                  step of the first one.
//   K2 resolve     lane j rebuilds the true state at its chain start from
//                  e[] of the q = ceil(L/S) chains before it: continue from
//                  e[j-q], and in every following chain walk only until the
//                  depth test says the walk has merged with that chain's own
//                  root walk, then jump to its e[].  Then it walks the head
//                  of its own chain from the true state until the same test
//                  holds; hits found there are staged separately.  If a K1
//                  hit lies inside that unmerged head, K2 re-walks the whole
//                  chain and K1's records for it are dropped.
//                  Typical cost: a handful of steps per chain; worst case
//                  (q-1)*S + S, the price of a classic (L-1)-byte halo.
//   scan + scatter per-chain counts -> offsets -> records land in position
//                  order; pattern index = out[state] is looked up here, off
//                  the walk's critical path.
//
// Depth test: non-final dev ids are in BFS order, so depth(s) <= m  <=>
// s < depth_cum[m]; final states carry their depth in depth_final[].
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstring>

#include "acm_internal.h"
#include "device_dfa.h"

namespace {

constexpr int kBlock1 = 1024;          // K1 workgroup: 16 waves, one per CU (LDS bound)
constexpr int kWaves1 = kBlock1 / 64;
constexpr int kBlock2 = 256;
constexpr uint32_t kNoFirst = 0xFFFFu;

struct ScanArgs {
	const uint32_t *cold;
	const uint16_t *hot;
	const int32_t *out;
	const uint32_t *dev2ref;
	const uint32_t *depth_cum;
	const uint16_t *depth_final;
	const uint4 *text16;
	const uint8_t *text;
	uint32_t n;             // text bytes
	uint32_t S, logS;       // chain bytes
	uint32_t n_chains;
	uint32_t n_tiles;       // K1 wave tiles
	uint32_t H;             // hot rows
	uint32_t F;             // first final dev id
	uint32_t L;             // max pattern length
	uint32_t q;             // look-back chains
	uint32_t init_state;    // dev numbering
	uint32_t drop_before;   // records ending before this offset are context (halo), not output
	int32_t off_shift;      // added to every reported offset
	// workspace
	uint32_t *end_state;
	uint32_t *c1f;
	uint32_t *k2info;
	int32_t *cnt;
	int32_t *off;
	uint32_t *wave_cnt1;
	uint32_t *wave_cnt2;
	uint32_t *misc;         // [0] last state (dev), [1] total records
	uint2 *stage1;
	uint2 *stage2;
	// output
	int32_t *pat_plane;
	int32_t *off_plane;
	uint32_t plane_capacity;
};

__device__ __forceinline__ uint32_t mbcnt64(uint64_t m)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

template <int K>
__device__ __forceinline__ uint32_t byte_of(const uint4 &w)
{
	const uint32_t d = (K < 4) ? w.x : (K < 8) ? w.y : (K < 12) ? w.z : w.w;
	return (d >> (8 * (K & 3))) & 0xFFu;
}

// One DFA step for C independent chains, loads issued back to back so the
// chains hide each other's latency.  Cold lanes (state beyond the LDS rows,
// or an LDS cell that says "does not fit") read the HBM plane; the branch
// around that is wave-uniform.
template <int C, int K, bool GUARD>
__device__ __forceinline__ void step_all(const ScanArgs &a, const uint16_t *hot, const uint4 (&w)[C],
    uint32_t (&st)[C], uint32_t (&cnt)[C], uint32_t (&first)[C], const uint32_t (&base)[C],
    const uint32_t (&len)[C], uint32_t g, uint32_t &wcount, uint2 *stage)
{
	uint32_t idx[C], e[C];
	bool need[C];
	bool any_need = false;
#pragma unroll
	for (int c = 0; c < C; c++) {
		idx[c] = (st[c] << 8) | byte_of<K>(w[c]);
		e[c] = hot[st[c] < a.H ? idx[c] : 0u];
	}
#pragma unroll
	for (int c = 0; c < C; c++) {
		need[c] = (st[c] >= a.H) | (e[c] == acm::kHotSentinel);
		any_need |= need[c];
	}
	if (__builtin_amdgcn_ballot_w64(any_need)) {
		uint32_t v[C];
#pragma unroll
		for (int c = 0; c < C; c++)
			v[c] = a.cold[need[c] ? idx[c] : 0u];
#pragma unroll
		for (int c = 0; c < C; c++)
			e[c] = need[c] ? v[c] : e[c];
	}
	const uint32_t step = g * 16 + K + 1;  // 1-based step inside the chain
	bool hit[C];
	bool any_hit = false;
#pragma unroll
	for (int c = 0; c < C; c++) {
		if (GUARD && step > len[c])
			e[c] = st[c];  // past the end of the text: freeze
		hit[c] = (e[c] >= a.F) && (!GUARD || step <= len[c]);
		any_hit |= hit[c];
	}
	if (__builtin_amdgcn_ballot_w64(any_hit)) {
#pragma unroll
		for (int c = 0; c < C; c++) {
			hit[c] = hit[c] && (base[c] + step - 1 >= a.drop_before);
			const uint64_t m = __builtin_amdgcn_ballot_w64(hit[c]);
			if (m) {
				if (hit[c]) {
					stage[wcount + mbcnt64(m)] =
					    make_uint2(base[c] + step - 1, e[c] | (cnt[c] << 24));
					if (cnt[c] == 0)
						first[c] = step;
					cnt[c]++;
				}
				wcount += (uint32_t)__popcll(m);
			}
		}
	}
#pragma unroll
	for (int c = 0; c < C; c++)
		st[c] = e[c];
}

template <int C, bool GUARD>
__device__ __forceinline__ void walk_tile(const ScanArgs &a, const uint16_t *hot, uint32_t wt,
    uint32_t lane)
{
	uint32_t st[C], cnt[C], first[C], base[C], len[C], chain[C];
	uint32_t wcount = 0;
	uint2 *stage = a.stage1 + (((size_t)wt * C * 64) << a.logS);
#pragma unroll
	for (int c = 0; c < C; c++) {
		chain[c] = (wt * C + c) * 64 + lane;
		base[c] = chain[c] << a.logS;
		len[c] = GUARD ? (base[c] >= a.n ? 0u : min(a.S, a.n - base[c])) : a.S;
		st[c] = 0;
		cnt[c] = 0;
		first[c] = kNoFirst;
	}
	const uint32_t groups = a.S >> 4;
	for (uint32_t g = 0; g < groups; g++) {
		uint4 w[C];
#pragma unroll
		for (int c = 0; c < C; c++) {
			if (!GUARD || base[c] + g * 16 < a.n)
				w[c] = a.text16[(base[c] >> 4) + g];
			else
				w[c] = make_uint4(0, 0, 0, 0);
		}
#define ACM_STEP(K) step_all<C, K, GUARD>(a, hot, w, st, cnt, first, base, len, g, wcount, stage)
		ACM_STEP(0); ACM_STEP(1); ACM_STEP(2); ACM_STEP(3);
		ACM_STEP(4); ACM_STEP(5); ACM_STEP(6); ACM_STEP(7);
		ACM_STEP(8); ACM_STEP(9); ACM_STEP(10); ACM_STEP(11);
		ACM_STEP(12); ACM_STEP(13); ACM_STEP(14); ACM_STEP(15);
#undef ACM_STEP
	}
#pragma unroll
	for (int c = 0; c < C; c++) {
		if (chain[c] < a.n_chains) {
			a.end_state[chain[c]] = st[c];
			a.c1f[chain[c]] = cnt[c] | (first[c] << 16);
		}
	}
	if (lane == 0)
		a.wave_cnt1[wt] = wcount;
}

// K1: persistent workgroups (one per CU), the hot rows live in LDS for the
// whole launch, each wave takes wave tiles of C*64 chains round-robin.
template <int C>
__global__ __launch_bounds__(kBlock1) void k_spec_walk(ScanArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint16_t hot[];
	{
		const uint4 *src = (const uint4 *)a.hot;
		uint4 *dst = (uint4 *)hot;
		const uint32_t n16 = a.H * 32;  // 512 B per row
		for (uint32_t i = threadIdx.x; i < n16; i += kBlock1)
			dst[i] = src[i];
	}
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = blockIdx.x * kWaves1 + (threadIdx.x >> 6);
	const uint32_t nwaves = gridDim.x * kWaves1;
	const uint32_t tile_bytes = (C * 64u) << a.logS;
	for (uint32_t wt = wave; wt < a.n_tiles; wt += nwaves) {
		const bool full = (uint64_t)(wt + 1) * tile_bytes <= a.n;
		if (full)
			walk_tile<C, false>(a, hot, wt, lane);
		else
			walk_tile<C, true>(a, hot, wt, lane);
	}
}

__device__ __forceinline__ bool depth_le(const ScanArgs &a, uint32_t s, uint32_t m)
{
	if (s < a.F)
		return s < a.depth_cum[min(m, a.L + 1)];
	return a.depth_final[s - a.F] <= m;
}

// K2: one lane per chain.
__global__ __launch_bounds__(kBlock2) void k_resolve(ScanArgs a)
{
	__shared__ uint32_t wave_fill[kBlock2 / 64];
	const uint32_t j = blockIdx.x * kBlock2 + threadIdx.x;
	const uint32_t wv = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0)
		wave_fill[wv] = 0;
	__syncthreads();
	const uint32_t gw = j >> 6;
	uint2 *stage = a.stage2 + (((size_t)gw * 64) << a.logS);

	if (j < a.n_chains) {
		// ---- true state at the start of chain j -------------------------
		uint32_t state, c;
		if (j < a.q) {
			state = a.init_state;
			c = 0;
		} else {
			state = a.end_state[j - a.q];
			c = j - a.q + 1;
		}
		uint32_t m = 0;
		while (c < j) {
			if (m == 0 && state == 0) {  // root: already merged with chain c's own walk
				state = a.end_state[c];
				c++;
				continue;
			}
			m++;
			state = a.cold[((size_t)state << 8) | a.text[((size_t)c << a.logS) + m - 1]];
			if (depth_le(a, state, m)) {
				state = a.end_state[c];
				c++;
				m = 0;
			} else if (m == a.S) {
				c++;
				m = 0;
			}
		}
		// ---- head of the own chain ---------------------------------------
		const uint32_t base = j << a.logS;
		const uint32_t len = min(a.S, a.n - base);
		const uint32_t info = a.c1f[j];
		const uint32_t c1 = info & 0xFFFFu, f = info >> 16;
		uint32_t c2 = 0;
		bool killed = false, merged = (state == 0);
		if (!merged) {
			for (m = 1; m <= len; m++) {
				state = a.cold[((size_t)state << 8) | a.text[(size_t)base + m - 1]];
				if (!killed && depth_le(a, state, m)) {
					merged = true;
					break;
				}
				if (state >= a.F && base + m - 1 >= a.drop_before) {
					const uint32_t slot = atomicAdd(&wave_fill[wv], 1u);
					stage[slot] = make_uint2(base + m - 1, state | (c2 << 24));
					c2++;
				}
				if (!killed && m >= f)
					killed = true;  // a K1 hit sits in the unmerged head: take the chain over
			}
		}
		a.k2info[j] = c2 | (killed ? 0x10000u : 0u);
		a.cnt[j] = (int32_t)(c2 + (killed ? 0u : c1));
		if (j == a.n_chains - 1)
			a.misc[0] = merged ? a.end_state[j] : state;
	}
	__syncthreads();
	if ((threadIdx.x & 63) == 0 && gw * 64 < a.n_chains)
		a.wave_cnt2[gw] = wave_fill[wv];
}

// records of one staging region -> final planes, one wave per region
template <bool FROM_K1>
__global__ __launch_bounds__(256) void k_scatter(ScanArgs a, uint32_t regions, uint32_t chains_per_region)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= regions)
		return;
	const uint32_t nrec = FROM_K1 ? a.wave_cnt1[r] : a.wave_cnt2[r];
	const uint2 *stage = (FROM_K1 ? a.stage1 : a.stage2) + (((size_t)r * chains_per_region) << a.logS);
	for (uint32_t i = lane; i < nrec; i += 64) {
		const uint2 rec = stage[i];
		const uint32_t pos = rec.x, st = rec.y & 0xFFFFFFu, seq = rec.y >> 24;
		const uint32_t j = pos >> a.logS;
		const uint32_t info = a.k2info[j];
		uint32_t d = (uint32_t)a.off[j] + seq;
		if (FROM_K1) {
			if (info & 0x10000u)
				continue;
			d += info & 0xFFFFu;
		}
		if (d + 2 < a.plane_capacity) {
			a.pat_plane[1 + d] = a.out[st];
			a.off_plane[1 + d] = (int32_t)pos + a.off_shift;
		}
	}
}

// header and trailer cells of the compact planes (compactarray.cl:49-55)
__global__ void k_finalize(ScanArgs a, int have_chains)
{
	if (threadIdx.x != 0 || blockIdx.x != 0)
		return;
	const uint32_t total = have_chains ? a.misc[1] : 0u;
	const uint32_t last = have_chains ? a.misc[0] : a.init_state;
	const int32_t last_ref = (int32_t)a.dev2ref[last];
	uint32_t tail = total + 1;
	if (tail > a.plane_capacity - 1)
		tail = a.plane_capacity - 1;
	a.pat_plane[0] = (int32_t)total;
	a.off_plane[0] = (int32_t)total;
	a.pat_plane[tail] = last_ref;
	a.off_plane[tail] = last_ref;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Layout {
	size_t end_state, c1f, k2info, cnt, off, wave_cnt1, wave_cnt2, misc, stage1, stage2, scan_ws;
	size_t scan_ws_bytes;
	size_t total;
};

// sized for the smallest chain length (16 B) so any geometry fits
Layout layout_for(size_t max_text)
{
	Layout l;
	const size_t chains = max_text / 16 + 64 * 4 + 64;
	const size_t waves = chains / 64 + 2;
	const size_t stage_recs = max_text + (size_t)4 * 64 * 256;
	size_t o = 0;
	auto take = [&](size_t bytes) {
		size_t at = o;
		o = align_up(o + bytes, 256);
		return at;
	};
	l.end_state = take(chains * 4);
	l.c1f = take(chains * 4);
	l.k2info = take(chains * 4);
	l.cnt = take(chains * 4);
	l.off = take(chains * 4);
	l.wave_cnt1 = take(waves * 4);
	l.wave_cnt2 = take(waves * 4);
	l.misc = take(64);
	l.stage1 = take(stage_recs * 8);
	l.stage2 = take(stage_recs * 8);
	l.scan_ws_bytes = acm_exclusive_scan_workspace_bytes(chains);
	l.scan_ws = take(l.scan_ws_bytes);
	l.total = o;
	return l;
}

template <int C>
int launch_spec_walk(const ScanArgs &a, int num_cus, hipStream_t s)
{
	const size_t lds = (size_t)a.H * 512;
	if (lds > 48 * 1024)  // per device, so not cached in a static
		ACM_HIP_TRY(hipFuncSetAttribute((const void *)k_spec_walk<C>,
		    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(acm::kHotRowsMax * 512)));
	uint32_t blocks = (a.n_tiles + kWaves1 - 1) / kWaves1;
	if (blocks > (uint32_t)num_cus)
		blocks = (uint32_t)num_cus;
	hipLaunchKernelGGL(k_spec_walk<C>, dim3(blocks), dim3(kBlock1), lds, s, a);
	ACM_HIP_TRY(hipGetLastError());
	return ACM_OK;
}

}  // namespace

extern "C" size_t acm_scan_workspace_bytes(const acm_dfa *, size_t max_text)
{
	return layout_for(max_text).total;
}

extern "C" int acm_scan_set_chain_bytes(acm_dfa *d, int chain_bytes)
{
	if (!d)
		return 0;
	if (chain_bytes == 0 || (chain_bytes >= 16 && chain_bytes <= 256 &&
	    (chain_bytes & (chain_bytes - 1)) == 0))
		d->chain_bytes = chain_bytes;
	return d->chain_bytes;
}

extern "C" int acm_scan_kernel_count(void) { return 8; }

extern "C" int acm_scan_async(const acm_dfa *d, const void *d_text, size_t n, long init_state,
    void *d_workspace, size_t workspace_bytes, int32_t *d_pat_plane, int32_t *d_off_plane,
    size_t plane_capacity, void *stream)
{
	return acm_scan_shard_async(d, d_text, n, 0, 0, init_state, d_workspace, workspace_bytes,
	    d_pat_plane, d_off_plane, plane_capacity, stream);
}

extern "C" int acm_scan_shard_async(const acm_dfa *d, const void *d_text, size_t n, size_t halo,
    long offset_shift, long init_state, void *d_workspace, size_t workspace_bytes,
    int32_t *d_pat_plane, int32_t *d_off_plane, size_t plane_capacity, void *stream)
{
	if (!d || !d_pat_plane || !d_off_plane || plane_capacity < 2 || (n && !d_text))
		return acm::fail(ACM_ERR_ARG, "acm_scan_async: bad arguments");
	if (n > 0x7FFFFFEFul)
		return acm::fail(ACM_ERR_LIMIT, "acm_scan_async: %zu bytes exceed the 2 GiB buffer limit", n);
	if (((uintptr_t)d_text & 15) != 0)
		return acm::fail(ACM_ERR_ARG, "acm_scan_async: text must be 16-byte aligned");
	if (halo > n || offset_shift < INT32_MIN || offset_shift > INT32_MAX ||
	    (long)n + offset_shift > (long)INT32_MAX)
		return acm::fail(ACM_ERR_ARG, "acm_scan_shard_async: halo/offset_shift out of range");
	if (init_state < 0 || (uint64_t)init_state >= d->num_states)
		return acm::fail(ACM_ERR_ARG, "acm_scan_async: init_state %ld is not a state", init_state);
	if (plane_capacity > 0xFFFFFFFFul)
		plane_capacity = 0xFFFFFFFFul;
	const Layout l = layout_for(n);
	if (!d_workspace || workspace_bytes < l.total)
		return acm::fail(ACM_ERR_ARG, "acm_scan_async: workspace %zu B < required %zu B",
		    workspace_bytes, l.total);
	hipStream_t s = (hipStream_t)stream;
	ACM_HIP_TRY(hipSetDevice(d->device));

	// geometry: enough chains to give every lane of every CU work, chains
	// as long as that allows (longer chains = fewer look-back steps)
	constexpr int C = 2;
	uint32_t S = (uint32_t)d->chain_bytes;
	if (S == 0) {
		const size_t lanes = (size_t)d->num_cus * kBlock1 * C;
		S = 16;
		while (S < 256 && (size_t)S * 2 * lanes <= n)
			S *= 2;
	}
	uint32_t logS = 0;
	while ((1u << logS) < S)
		logS++;

	char *ws = (char *)d_workspace;
	ScanArgs a;
	memset(&a, 0, sizeof(a));
	a.cold = d->d_cold;
	a.hot = d->d_hot;
	a.out = d->d_out;
	a.dev2ref = d->d_dev2ref;
	a.depth_cum = d->d_depth_cum;
	a.depth_final = d->d_depth_final;
	a.text16 = (const uint4 *)d_text;
	a.text = (const uint8_t *)d_text;
	a.n = (uint32_t)n;
	a.S = S;
	a.logS = logS;
	a.n_chains = (uint32_t)((n + S - 1) >> logS);
	a.n_tiles = (a.n_chains + C * 64 - 1) / (C * 64);
	a.H = d->hot_rows;
	a.F = d->first_final;
	a.L = d->max_pattern_len;
	a.q = (a.L + S - 1) / S;
	if (a.q == 0)
		a.q = 1;
	a.init_state = d->ref2dev[(size_t)init_state];
	a.drop_before = (uint32_t)halo;
	a.off_shift = (int32_t)offset_shift;
	a.end_state = (uint32_t *)(ws + l.end_state);
	a.c1f = (uint32_t *)(ws + l.c1f);
	a.k2info = (uint32_t *)(ws + l.k2info);
	a.cnt = (int32_t *)(ws + l.cnt);
	a.off = (int32_t *)(ws + l.off);
	a.wave_cnt1 = (uint32_t *)(ws + l.wave_cnt1);
	a.wave_cnt2 = (uint32_t *)(ws + l.wave_cnt2);
	a.misc = (uint32_t *)(ws + l.misc);
	a.stage1 = (uint2 *)(ws + l.stage1);
	a.stage2 = (uint2 *)(ws + l.stage2);
	a.pat_plane = d_pat_plane;
	a.off_plane = d_off_plane;
	a.plane_capacity = (uint32_t)plane_capacity;

	if (a.n_chains == 0) {
		hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, s, a, 0);
		ACM_HIP_TRY(hipGetLastError());
		return ACM_OK;
	}

	hipEvent_t ev[3] = { nullptr, nullptr, nullptr };
	if (d->profile) {
		for (auto &e : ev) {
			if (!d->profile_pool.empty()) {  // recycled: no create/destroy in a timed loop
				e = (hipEvent_t)d->profile_pool.back();
				d->profile_pool.pop_back();
			} else {
				ACM_HIP_TRY(hipEventCreate(&e));
			}
		}
		ACM_HIP_TRY(hipEventRecord(ev[0], s));
	}
	int rc = launch_spec_walk<C>(a, d->num_cus, s);
	if (rc != ACM_OK)
		return rc;
	if (d->profile)
		ACM_HIP_TRY(hipEventRecord(ev[1], s));
	hipLaunchKernelGGL(k_resolve, dim3((a.n_chains + kBlock2 - 1) / kBlock2), dim3(kBlock2), 0, s, a);
	ACM_HIP_TRY(hipGetLastError());
	rc = acm_exclusive_scan_i32(a.cnt, a.off, a.n_chains, (int32_t *)(a.misc + 1), ws + l.scan_ws,
	    l.scan_ws_bytes, s);
	if (rc != ACM_OK)
		return rc;
	const uint32_t waves2 = (a.n_chains + 63) / 64;
	hipLaunchKernelGGL(k_scatter<true>, dim3((a.n_tiles + 3) / 4), dim3(256), 0, s, a, a.n_tiles,
	    (uint32_t)(C * 64));
	hipLaunchKernelGGL(k_scatter<false>, dim3((waves2 + 3) / 4), dim3(256), 0, s, a, waves2, 64u);
	hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, s, a, 1);
	ACM_HIP_TRY(hipGetLastError());
	if (d->profile) {
		ACM_HIP_TRY(hipEventRecord(ev[2], s));
		for (auto e : ev)
			d->profile_events.push_back((void *)e);
	}
	return ACM_OK;
}

extern "C" int acm_scan_profile_enable(acm_dfa *d, int enable)
{
	if (!d)
		return acm::fail(ACM_ERR_ARG, "acm_scan_profile_enable: null dfa");
	d->profile = enable != 0;
	return ACM_OK;
}

extern "C" int acm_scan_profile_read(acm_dfa *d, double *walk_ms, double *pipeline_ms, int *launches)
{
	if (!d)
		return acm::fail(ACM_ERR_ARG, "acm_scan_profile_read: null dfa");
	double walk = 0, pipe = 0;
	int n = 0;
	for (size_t i = 0; i + 2 < d->profile_events.size(); i += 3) {
		hipEvent_t e0 = (hipEvent_t)d->profile_events[i], e1 = (hipEvent_t)d->profile_events[i + 1],
			   e2 = (hipEvent_t)d->profile_events[i + 2];
		float a = 0, b = 0;
		ACM_HIP_TRY(hipEventSynchronize(e2));
		ACM_HIP_TRY(hipEventElapsedTime(&a, e0, e1));
		ACM_HIP_TRY(hipEventElapsedTime(&b, e0, e2));
		walk += a;
		pipe += b;
		n++;
		d->profile_pool.push_back((void *)e0);
		d->profile_pool.push_back((void *)e1);
		d->profile_pool.push_back((void *)e2);
	}
	d->profile_events.clear();
	if (walk_ms) *walk_ms = walk;
	if (pipeline_ms) *pipeline_ms = pipe;
	if (launches) *launches = n;
	return ACM_OK;
}
