// The scan pipeline: speculative chain walk -> boundary resolve -> block scan
// of per-chain counts -> ordered scatter.  gfx950 only.
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
//   K1 k_spec_walk   every lane walks its chains from the ROOT state.  The
//                    state reached after m bytes equals the true serial state
//                    as soon as the true state's depth is <= m (a state only
//                    remembers its last depth bytes); from then on the two
//                    walks coincide.  K1 records the end state e[j], the
//                    tentative hits (staged per wave), their number and the
//                    step of the first one.  Its tile epilogue already settles
//                    the common case of the next two stages (see below).
//   K2a k_probe      lane c walks chain c from e[c-1] -- the state chain c
//                    starts in whenever chain c-1 has merged -- until the
//                    depth test says it has merged with chain c's own root
//                    walk: typically after one byte.
//   K2b k_resolve    lane j: if every chain of its look-back window
//                    (q = ceil(L/S) chains) merged in its probe, the true
//                    start state is known without walking and K1's records
//                    stand.  Otherwise it rebuilds the start state by walking
//                    (re-using probe results wherever the walk enters a chain
//                    in the state the probe assumed), re-walks the head of
//                    its own chain from the true state, stages the true hits
//                    and, if a K1 hit lies in the unmerged head, takes the
//                    whole chain over (K1's records for it are dropped).
//                    Worst case (q-1)*S + S steps: the price of a classic
//                    (L-1)-byte halo, paid only where the text really is deep
//                    inside a pattern.
//   k_scan_top + k_scatter_all   per-chain counts -> offsets -> records land
//                    in position order; pattern index = out[state] is looked
//                    up here, off the walk's critical path.
//
// Deep walks (K2) get everything about the state they enter with the state
// itself: deep[idx] = state | depth << 32 | run << 48.  depth feeds the merge
// test; run is the length of the unary trie path ahead, along which states
// are consecutive ids and the walk only compares text with in_byte[] --
// 16 bytes per load level instead of one table lookup per byte.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <atomic>
#include <cstring>
#include <mutex>

#include "acm_internal.h"
#include "deep_walk.h"
#include "device_dfa.h"
#include "lds_walk.h"
#include "sparse.h"

namespace {

using acm_dev::ChainText;
using acm_dev::Deep;
using acm_dev::deep_step;
using acm_dev::fast_forward;

constexpr int kBlock1 = 1024;          // K1 workgroup: 16 waves, one per CU (LDS bound)
constexpr int kWaves1 = kBlock1 / 64;
constexpr int kBlock2 = 256;
constexpr uint32_t kNoFirst = 0xFFFFu;
constexpr uint8_t kProbeTodo = 0xFF;

struct ScanArgs {
	const uint32_t *cold;   // [states][1 << ls] next state, a cell per byte class
	const uint64_t *deep;   // [states][1 << ls] next | depth(next) << 32 | run(next) << 48
	const uint16_t *hot;    // [H][1 << ls]
	const uint8_t *cls;     // [256] byte -> class (the identity when ls == 8)
	uint32_t ls;            // log2 of the cells per row
	uint32_t halo_bytes;    // halo mode (below, walk_tile): bytes every chain walks in front of its own; else 0
	uint32_t halo_mode;
	uint32_t halo_pre;      // halo mode: the chains' text is loaded up front (k_halo_walk)
	const int32_t *out;
	const uint32_t *dev2ref;
	const uint8_t *in_byte;
	const uint4 *text16;
	const uint8_t *text;
	uint32_t n;             // text bytes
	uint32_t n_pad;         // n rounded up to 16 (readable)
	uint32_t S, logS;       // chain bytes
	uint32_t n_chains;
	uint32_t n_tiles;       // K1 wave tiles
	uint32_t H;             // hot rows
	uint32_t hot_depth1;    // hot ids below this have depth <= 1
	uint32_t F;             // first final dev id
	uint32_t L;             // max pattern length
	uint32_t q;             // look-back chains
	uint32_t init_state;    // dev numbering
	const uint32_t *init_ptr;   // not null: the state to start in is THERE (handed over on the device, k_carry_init), init_state is not
	uint32_t drop_before;   // records ending before this offset are context (halo), not output
	int32_t off_shift;      // added to every reported offset
	// workspace
	uint32_t *end_state;    // [chains] e[j]: end state of chain j's root walk
	uint32_t *c1f;          // [chains] K1 hit count | first-hit step << 16
	uint32_t *k2info;       // [chains] K2 hit count | K1 records dropped << 16
	uint32_t *wend;         // [chains] end state of the probe walk of chain c from e[c-1]
	uint8_t *probe;         // [chains] bit0 probe merged, bit1 head needs an emission walk
	uint8_t *rflag;         // [chains] 1: resolved by the walk kernel's epilogue
	int32_t *cnt;           // [chains] final record count of the chain
	int32_t *off;           // [blocks] per-256-chain totals, scanned in place
	uint32_t *wave_cnt1;
	uint32_t *wave_cnt2;
	uint32_t *misc;         // [0] last state (dev), [1] total records, [2] which pipeline produced the planes
	uint2 *stage1;
	uint2 *stage2;
	// output
	int32_t *pat_plane;
	int32_t *off_plane;
	uint32_t plane_capacity;
	uint32_t fold_blocks;      // > 0: off[] holds the raw totals of this many blocks; the scatter adds them up itself
};

// the state the text starts in (wave-uniform: one scalar load where it was handed over on the device)
__device__ __forceinline__ uint32_t init_of(const ScanArgs &a)
{
	return a.init_ptr ? *a.init_ptr : a.init_state;
}

__device__ __forceinline__ uint32_t mbcnt64(uint64_t m)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

__device__ __forceinline__ uint32_t cold_next(const ScanArgs &a, uint32_t idx)
{
	return a.cold[idx];
}

template <int K>
__device__ __forceinline__ uint32_t byte_of(const uint4 &w)
{
	const uint32_t d = (K < 4) ? w.x : (K < 8) ? w.y : (K < 12) ? w.z : w.w;
	return (d >> (8 * (K & 3))) & 0xFFu;
}

// ---------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------

// One DFA step for C independent chains, loads issued back to back so the
// chains hide each other's latency.  Cold lanes (state beyond the LDS rows,
// or an LDS cell that says "look in HBM") gather from the cold plane; the
// branch around that is wave-uniform.  (Issuing the HBM gather before the LDS
// result is known -- the plane a lane needs only depends on its state --
// measured 35% SLOWER on MI355X, so the gather waits for the LDS cell.)
template <int C, int K, bool GUARD, bool CLS, bool HALO>
__device__ __forceinline__ void step_all(const ScanArgs &a, const uint16_t *hot, const uint8_t *clsmap, uint32_t ls,
    const uint4 (&w)[C],
    uint32_t (&st)[C], uint32_t (&cnt)[C], uint32_t (&first)[C], const uint32_t (&base)[C],
    const uint32_t (&len)[C], const uint32_t (&lead)[C], uint32_t hb, uint32_t g, uint32_t &wcount, uint2 *stage)
{
	uint32_t idx[C], e[C];
	bool need[C];
	bool any_need = false;
#pragma unroll
	for (int c = 0; c < C; c++) {
		const uint32_t b = byte_of<K>(w[c]);
		idx[c] = CLS ? (st[c] << ls) | clsmap[b] : (st[c] << 8) | b;   // (the class of a byte: one more LDS read, off the chain)
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
			v[c] = cold_next(a, need[c] ? idx[c] : 0u);
#pragma unroll
		for (int c = 0; c < C; c++)
			e[c] = need[c] ? v[c] : e[c];
	}
	// 1-based step inside the chain; in halo mode the first hb bytes of the walk lie in front of
	// the chain (steps <= 0): they only bring the state up to date
	const int32_t step = (int32_t)(g * 16 + K + 1) - (HALO ? (int32_t)hb : 0);
	bool hit[C];
	bool any_hit = false;
#pragma unroll
	for (int c = 0; c < C; c++) {
		if (GUARD && (step > (int32_t)len[c] || (HALO && step <= -(int32_t)lead[c])))
			e[c] = st[c];  // past the end of the text, or in front of where this chain's walk starts: freeze
		hit[c] = (e[c] >= a.F) && (!HALO || step >= 1) && (!GUARD || step <= (int32_t)len[c]);
		any_hit |= hit[c];
	}
	if (__builtin_amdgcn_ballot_w64(any_hit)) {
#pragma unroll
		for (int c = 0; c < C; c++) {
			hit[c] = hit[c] && (base[c] + (uint32_t)step - 1 >= a.drop_before);
			const uint64_t m = __builtin_amdgcn_ballot_w64(hit[c]);
			if (m) {
				if (hit[c]) {
					stage[wcount + mbcnt64(m)] =
					    make_uint2(base[c] + (uint32_t)step - 1, e[c] | (cnt[c] << 24));
					if (cnt[c] == 0)
						first[c] = (uint32_t)step;
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

// End of a wave tile: the common case of the boundary resolve, done where the
// data already is.  Lane l holds e[chain], its left neighbour holds e[chain-1]
// (a shuffle away), the first text byte of the chain is still in a register
// and the hot rows are in LDS -- so the probe's first step (does the walk from
// e[chain-1] merge with this chain's own root walk after one byte?) costs one
// LDS lookup.  A chain whose probe merges that way, and whose look-back window
// is made of such chains, is final: count = K1 count, nothing to re-walk.
// Everything else is left to k_probe / k_resolve through the flag arrays.
template <int C>
__device__ __forceinline__ void tile_epilogue(const ScanArgs &a, const uint16_t *hot, uint32_t ls, uint32_t wt,
    uint32_t lane, const uint32_t (&st)[C], const uint32_t (&cnt)[C], const uint32_t (&chain)[C],
    const uint32_t (&fb)[C])
{
	bool decided[C];
	uint64_t dmask[C];
#pragma unroll
	for (int c = 0; c < C; c++) {
		// both shuffles are executed by the whole wave (a shuffle inside a divergent
		// branch cannot read lanes that did not take the branch)
		uint32_t pe = __shfl_up(st[c], 1, 64);
		const uint32_t prev_slot_last = __shfl(st[c > 0 ? c - 1 : 0], 63, 64);
		bool known = true;
		if (lane == 0) {
			if (c > 0)
				pe = prev_slot_last;
			else if (chain[c] == 0)
				pe = init_of(a);
			else
				known = false;   // previous chain belongs to another wave
		}
		bool ok = false;
		if (known && chain[c] < a.n_chains) {
			if (pe == 0) {
				ok = true;
			} else if (pe < a.H) {
				const uint32_t t = hot[(pe << ls) | fb[c]];   // fb: the class of the chain's first byte
				ok = t < a.hot_depth1;   // non-final, depth <= 1 (a sentinel cell is never below)
			}
		}
		decided[c] = ok;
		dmask[c] = __builtin_amdgcn_ballot_w64(ok);
	}
	// chain j is settled when its own probe merged at step <= 1 and so did the
	// probes of the q-1 chains before it (then the true start state is e[j-1])
	const uint32_t window = a.q - 1;
#pragma unroll
	for (int c = 0; c < C; c++) {
		if (chain[c] >= a.n_chains)
			continue;
		bool final_here = decided[c] && window <= 32;
		const int p = c * 64 + (int)lane;   // position inside the tile
		for (uint32_t k = 1; final_here && k <= window; k++) {
			const int pos = p - (int)k;
			if (pos < 0) {
				final_here = (wt == 0);   // before the first chain of the text: nothing to check
				break;
			}
			uint64_t mword = dmask[0];
#pragma unroll
			for (int cc = 1; cc < C; cc++)
				mword = (pos >> 6) == cc ? dmask[cc] : mword;
			final_here = (mword >> (pos & 63)) & 1ull;
		}
		a.probe[chain[c]] = decided[c] ? 1 : kProbeTodo;
		if (decided[c])
			a.wend[chain[c]] = st[c];
		a.rflag[chain[c]] = final_here ? 1 : 0;
		if (final_here) {
			a.k2info[chain[c]] = 0;
			a.cnt[chain[c]] = (int32_t)cnt[c];
		}
	}
}

// Halo mode (HALO): when the longest pattern is short against the chain (L - 1 <= S, a set of
// words), a chain does not start in the root at its first byte and leave the truth to the probe
// and resolve kernels -- it starts hb = L - 1 bytes (rounded up to 16) EARLIER: the DFA state only
// remembers the last L - 1 bytes, so from its own first byte on the walk is the serial one.  Hits
// in front of the chain are not counted; the chain's count is final, the tile's total goes to the
// scatter kernel directly, k_probe and k_resolve are not launched.  A chain less than hb bytes
// into the text starts at byte 0 in the carried-in state.
// PRE (halo mode, chains of up to five 16-byte groups): all text of the lane's chains is loaded up
// front, a chain's five loads back to back.  Taken one group per trip through the loop, a 64-byte
// line of the text is touched by four or five loads many microseconds apart -- lane-per-chain is a
// gather, 16 bytes per lane at a stride of one chain -- and with 8 MB of such lines in flight per
// XCD against 4 MB of L2 they are fetched again and again (135 MB counted for a 32 MiB text).
constexpr int kPreGroups = 5;
template <int C, bool GUARD, bool CLS, bool HALO, bool PRE = false>
__device__ __forceinline__ void walk_tile(const ScanArgs &a, const uint16_t *hot, const uint8_t *clsmap, uint32_t wt,
    uint32_t lane)
{
	const uint32_t ls = CLS ? a.ls : 8u;
	const uint32_t hb = HALO ? a.halo_bytes : 0u;
	uint32_t st[C], cnt[C], first[C], base[C], len[C], chain[C], fb[C], lead[C];
	uint32_t wcount = 0;
	uint2 *stage = a.stage1 + (((size_t)wt * C * 64) << a.logS);
#pragma unroll
	for (int c = 0; c < C; c++) {
		chain[c] = (wt * C + c) * 64 + lane;
		base[c] = chain[c] << a.logS;
		len[c] = GUARD ? (base[c] >= a.n ? 0u : min(a.S, a.n - base[c])) : a.S;
		lead[c] = min(hb, base[c]);
		st[c] = (HALO && base[c] <= hb) ? init_of(a) : 0u;
		cnt[c] = 0;
		first[c] = kNoFirst;
		fb[c] = 0;
	}
	const uint32_t groups = (a.S + hb) >> 4;
	// (five named sets, picked by a branch on the wave-uniform g below: an array of them indexed by g
	// would live in scratch)
	uint4 p0[C], p1[C], p2[C], p3[C], p4[C];
	static_assert(kPreGroups == 5, "five register sets");
	if (PRE) {
		auto fetch = [&](int c, uint32_t gi) -> uint4 {
			const int64_t at = (int64_t)base[c] - hb + (int64_t)gi * 16;
			if (gi < groups && (!GUARD || (at >= 0 && at < (int64_t)a.n)))
				return a.text16[at >> 4];
			return make_uint4(0, 0, 0, 0);
		};
#pragma unroll
		for (int c = 0; c < C; c++) {
			p0[c] = fetch(c, 0);
			p1[c] = fetch(c, 1);
			p2[c] = fetch(c, 2);
			p3[c] = fetch(c, 3);
			p4[c] = fetch(c, 4);
			if (!HALO)   // speculative mode: the class of the chain's first byte, for the tile epilogue
				fb[c] = CLS ? (uint32_t)clsmap[p0[c].x & 0xFFu] : p0[c].x & 0xFFu;
		}
	}
	for (uint32_t g = 0; g < groups; g++) {
		// (prefetching the next group does not help: vector loads return in order, so
		// the first cold gather of this group would wait for the prefetch anyway)
		uint4 w[C];
		if (PRE) {   // (g is wave-uniform: scalar branches, no indexed registers)
			const uint32_t gu = __builtin_amdgcn_readfirstlane(g);
#pragma unroll
			for (int c = 0; c < C; c++) {
				if (gu == 0) w[c] = p0[c];
				else if (gu == 1) w[c] = p1[c];
				else if (gu == 2) w[c] = p2[c];
				else if (gu == 3) w[c] = p3[c];
				else w[c] = p4[c];
			}
		}
#pragma unroll
		for (int c = 0; c < C && !PRE; c++) {
			const int64_t at = (int64_t)base[c] - hb + (int64_t)g * 16;   // first byte of the group
			if (!GUARD || (at >= 0 && at < (int64_t)a.n))
				w[c] = a.text16[at >> 4];
			else
				w[c] = make_uint4(0, 0, 0, 0);
			if (!HALO && g == 0)
				fb[c] = CLS ? (uint32_t)clsmap[w[c].x & 0xFFu] : w[c].x & 0xFFu;
		}
#define ACM_STEP(K) step_all<C, K, GUARD, CLS, HALO>(a, hot, clsmap, ls, w, st, cnt, first, base, len, lead, hb, g, wcount, stage)
		ACM_STEP(0); ACM_STEP(1); ACM_STEP(2); ACM_STEP(3);
		ACM_STEP(4); ACM_STEP(5); ACM_STEP(6); ACM_STEP(7);
		ACM_STEP(8); ACM_STEP(9); ACM_STEP(10); ACM_STEP(11);
		ACM_STEP(12); ACM_STEP(13); ACM_STEP(14); ACM_STEP(15);
#undef ACM_STEP
	}
	if (HALO) {
		static_assert(!HALO || C * 64 == kBlock2, "halo mode: a wave tile is a scatter block");
		uint32_t total = 0;
#pragma unroll
		for (int c = 0; c < C; c++) {
			if (chain[c] < a.n_chains) {
				a.cnt[chain[c]] = (int32_t)cnt[c];
				total += cnt[c];
				if (chain[c] == a.n_chains - 1)
					a.misc[0] = st[c];   // the state after the last byte
			}
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1)
			total += __shfl_xor(total, o, 64);
		if (lane == 0) {
			a.wave_cnt1[wt] = wcount;
			a.off[wt] = (int32_t)total;   // the scatter block's total (k_resolve's job otherwise)
		}
		if (lane < kBlock2 / 64)
			a.wave_cnt2[wt * (kBlock2 / 64) + lane] = 0;   // nothing staged by a resolve kernel
		return;
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
	tile_epilogue<C>(a, hot, ls, wt, lane, st, cnt, chain, fb);
}

// K1: persistent workgroups (one per CU), the hot rows live in LDS for the
// whole launch, each wave takes wave tiles of C*64 chains round-robin.
template <int C, bool CLS, bool HALO>
__global__ __launch_bounds__(kBlock1) void k_spec_walk(ScanArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint16_t hot[];
	uint8_t *clsmap = (uint8_t *)hot + acm::kHotBytes;   // behind the rows: the byte classes (CLS only)
	if (CLS && threadIdx.x < 64)
		((uint32_t *)clsmap)[threadIdx.x] = ((const uint32_t *)a.cls)[threadIdx.x];
	{
		// every workgroup copies the same table: start each one at a different
		// offset so the CUs do not all ask the same L2 channel at the same time
		const uint4 *src = (const uint4 *)a.hot;
		uint4 *dst = (uint4 *)hot;
		const uint32_t n16 = ((a.H << a.ls) + 7) >> 3;  // uint4s: a row has 1 << ls cells of 2 bytes (the table is padded to 16)
		const uint32_t rot = n16 ? (blockIdx.x * 1021u) % n16 : 0u;
		for (uint32_t i = threadIdx.x; i < n16; i += kBlock1) {
			uint32_t j = i + rot;
			j = j >= n16 ? j - n16 : j;
			dst[j] = src[j];
		}
	}
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = blockIdx.x * kWaves1 + (threadIdx.x >> 6);
	const uint32_t nwaves = gridDim.x * kWaves1;
	const uint32_t tile_bytes = (C * 64u) << a.logS;
	for (uint32_t wt = wave; wt < a.n_tiles; wt += nwaves) {
		// (the unguarded walk: every chain of the tile whole, and in halo mode with its whole halo)
		const bool full = (uint64_t)(wt + 1) * tile_bytes <= a.n && (!HALO || (uint64_t)wt * tile_bytes >= a.halo_bytes);
		if (full)
			walk_tile<C, false, CLS, HALO>(a, hot, clsmap, wt, lane);
		else
			walk_tile<C, true, CLS, HALO>(a, hot, clsmap, wt, lane);
	}
}

// Halo mode with the text loaded up front (walk_tile, PRE): the registers that takes are there for
// workgroups of 8 waves; with chains of 64 bytes a 32 MiB text has a tile per wave of them anyway.
constexpr int kBlockPre = 768;
// HALO = false: the speculative mode's chains loaded up front the same way (chains of up to 80 bytes; with the
// 64-byte chains long patterns get, a lane's four 16-byte loads per chain at a stride of one chain fetched every
// line of the text several times: 182 MB counted for a 32 MiB text).
// C = 2, BLOCK = 1024 (speculative mode): 16 waves of two chains per lane -- 512 Ki chains a round, which is what a
// 32 MiB text has at 64 bytes a chain: every lane busy, where 12 waves x 4 chains leave a third of them without work.
template <bool CLS, bool HALO = true, int C = 4, int BLOCK = kBlockPre>
__global__ __launch_bounds__(BLOCK) void k_halo_walk(ScanArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint16_t hot[];
	uint8_t *clsmap = (uint8_t *)hot + acm::kHotBytes;
	if (CLS && threadIdx.x < 64)
		((uint32_t *)clsmap)[threadIdx.x] = ((const uint32_t *)a.cls)[threadIdx.x];
	{
		const uint4 *src = (const uint4 *)a.hot;
		uint4 *dst = (uint4 *)hot;
		const uint32_t n16 = ((a.H << a.ls) + 7) >> 3;
		const uint32_t rot = n16 ? (blockIdx.x * 1021u) % n16 : 0u;
		for (uint32_t i = threadIdx.x; i < n16; i += BLOCK) {
			uint32_t j = i + rot;
			j = j >= n16 ? j - n16 : j;
			dst[j] = src[j];
		}
	}
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
	const uint32_t nwaves = gridDim.x * (BLOCK / 64);
	const uint32_t tile_bytes = ((uint32_t)C * 64u) << a.logS;
	for (uint32_t wt = wave; wt < a.n_tiles; wt += nwaves) {
		const bool full = (uint64_t)(wt + 1) * tile_bytes <= a.n && (uint64_t)wt * tile_bytes >= a.halo_bytes;
		if (full)
			walk_tile<C, false, CLS, HALO, true>(a, hot, clsmap, wt, lane);
		else
			walk_tile<C, true, CLS, HALO, true>(a, hot, clsmap, wt, lane);
	}
}

// ---------------------------------------------------------------------------
// K2: deep walks
// ---------------------------------------------------------------------------

// K2a probe: lane c walks chain c from e[c-1] (the state chain c would start
// in if chain c-1 has merged -- the overwhelmingly common case) until the
// depth test says it has merged with chain c's own root walk.  It records
// whether it merged, the state at the end of the chain under that
// assumption (wend), and whether anything in the unmerged head needs an
// emission walk (a true hit there, or a K1 hit that must be dropped).
// Chains the walk kernel's epilogue already probed are skipped.
__global__ __launch_bounds__(kBlock2) void k_probe(ScanArgs a)
{
	const uint32_t c = blockIdx.x * kBlock2 + threadIdx.x;
	if (c >= a.n_chains)
		return;
	if (a.probe[c] != kProbeTodo)
		return;
	Deep d;
	d.s = c == 0 ? init_of(a) : a.end_state[c - 1];
	d.depth = 0;
	d.run = 0;
	const uint32_t base = c << a.logS;
	const uint32_t len = min(a.S, a.n - base);
	const uint32_t f = a.c1f[c] >> 16;
	bool merged = (d.s == 0), work = false;
	if (!merged) {
		ChainText txt(a, base);
		for (uint32_t m = 1; m <= len; m++) {
			d = deep_step(a, d.s, txt.at(m));
			if (d.depth <= m) {
				merged = true;
				break;
			}
			work |= (d.s >= a.F) | (m >= f);
			if (d.run != 0 && d.s < a.F) {
				m += fast_forward(a, d, base + m, len - m);
				work |= (m >= f);
			}
		}
	}
	a.wend[c] = merged ? a.end_state[c] : d.s;
	a.probe[c] = (uint8_t)((merged ? 1u : 0u) | (work ? 2u : 0u));
}

// K2b resolve: one lane per chain (see the file header).
__global__ __launch_bounds__(kBlock2) void k_resolve(ScanArgs a)
{
	__shared__ uint32_t wave_fill[kBlock2 / 64];
	__shared__ uint32_t wave_sum[kBlock2 / 64];
	const uint32_t j = blockIdx.x * kBlock2 + threadIdx.x;
	const uint32_t wv = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0)
		wave_fill[wv] = 0;
	__syncthreads();
	const uint32_t gw = j >> 6;
	uint2 *stage = a.stage2 + (((size_t)gw * 64) << a.logS);
	uint32_t my_cnt = 0;

	if (j < a.n_chains && a.rflag[j]) {
		my_cnt = (uint32_t)a.cnt[j];   // resolved by the walk kernel's epilogue
		if (j == a.n_chains - 1)
			a.misc[0] = a.wend[j];
	} else if (j < a.n_chains) {
		const uint32_t info = a.c1f[j];
		const uint32_t c1 = info & 0xFFFFu, f = info >> 16;
		const uint32_t first = j < a.q ? 0u : j - a.q + 1;   // first chain the look-back walks
		// ---- fast path -----------------------------------------------------
		bool window_ok = true;
		for (uint32_t c = first; c + 1 < j; c++)            // chains first .. j-2 must have merged
			window_ok &= (a.probe[c] & 1u) != 0;
		const uint32_t pj = a.probe[j];
		const uint32_t assumed = j == 0 ? init_of(a) : a.end_state[j - 1];
		// q == 1 (S >= L): the window is empty and e[j-1] is already the true state
		uint32_t state = (j == 0 || first == j) ? assumed : a.wend[j - 1];
		if (window_ok && state == assumed && !(pj & 2u)) {
			a.k2info[j] = 0;
			my_cnt = c1;
			if (j == a.n_chains - 1)
				a.misc[0] = a.wend[j];
		} else {
			uint32_t m = 0;
			if (!window_ok) {
				// ---- general look-back: true state at the start of chain j ----
				uint32_t c = first;
				state = j < a.q ? init_of(a) : a.end_state[j - a.q];
				ChainText txt(a, first << a.logS);
				while (c < j) {
					if (m == 0 && state == 0) {  // root: merged with chain c's own walk
						state = a.end_state[c];
						c++;
						continue;
					}
					if (m == 0 && state == (c == 0 ? init_of(a) : a.end_state[c - 1])) {
						state = a.wend[c];       // exactly the walk the probe of chain c did
						c++;
						continue;
					}
					m++;
					// chains are adjacent: step m of chain c is byte (c - first) * S + m - 1
					Deep d = deep_step(a, state, txt.at(((c - first) << a.logS) + m));
					if (d.depth <= m) {
						state = a.end_state[c];
						c++;
						m = 0;
						continue;
					}
					if (d.run != 0 && d.s < a.F)
						m += fast_forward(a, d, (c << a.logS) + m, a.S - m);
					state = d.s;
					if (m == a.S) {
						c++;
						m = 0;
					}
				}
			}
			// ---- head of the own chain, from the true state --------------------
			const uint32_t base = j << a.logS;
			const uint32_t len = min(a.S, a.n - base);
			uint32_t c2 = 0;
			bool killed = false, merged = (state == 0);
			if (!merged) {
				ChainText txt(a, base);
				for (m = 1; m <= len; m++) {
					Deep d = deep_step(a, state, txt.at(m));
					state = d.s;
					if (!killed && d.depth <= m) {
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
					if (d.run != 0 && state < a.F) {
						m += fast_forward(a, d, base + m, len - m);
						state = d.s;
						if (!killed && m >= f)
							killed = true;
					}
				}
			}
			a.k2info[j] = c2 | (killed ? 0x10000u : 0u);
			my_cnt = c2 + (killed ? 0u : c1);
			if (j == a.n_chains - 1)
				a.misc[0] = merged ? a.end_state[j] : state;
		}
		a.cnt[j] = (int32_t)my_cnt;
	}
	// per-block total of the chain counts, scanned by k_scan_top
	uint32_t wsum = my_cnt;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		wsum += __shfl_down(wsum, o, 64);
	if ((threadIdx.x & 63) == 0)
		wave_sum[wv] = wsum;
	__syncthreads();
	if ((threadIdx.x & 63) == 0 && gw * 64 < a.n_chains)
		a.wave_cnt2[gw] = wave_fill[wv];
	if (threadIdx.x == 0) {
		uint32_t t = 0;
		for (int w = 0; w < kBlock2 / 64; w++)
			t += wave_sum[w];
		a.off[blockIdx.x] = (int32_t)t;
	}
}

// ---------------------------------------------------------------------------
// scan of the block totals + ordered scatter
// ---------------------------------------------------------------------------

// exclusive scan, in place, of the per-block totals by ONE workgroup: each
// thread owns a contiguous slice, the 1024 slice sums go through a Blelloch
// up-sweep / down-sweep in LDS.  Also publishes the grand total.
constexpr int kTopThreads = 1024;
constexpr uint32_t kTopMax = kTopThreads * 64;
constexpr uint32_t kFoldMax = 8192;   // block totals a scatter block still adds up by itself (64 MiB at S = 32)

__global__ __launch_bounds__(kTopThreads) void k_scan_top(ScanArgs a, uint32_t nb)
{
	__shared__ int32_t tree[kTopThreads + (kTopThreads >> 5) + 1];
	const int tid = threadIdx.x;
	const uint32_t per = (nb + kTopThreads - 1) / kTopThreads;
	const uint32_t lo = min(nb, (uint32_t)tid * per), hi = min(nb, lo + per);
	int32_t sum = 0;
	for (uint32_t i = lo; i < hi; i++)
		sum += a.off[i];
	auto pad = [](int i) { return i + (i >> 5); };
	tree[pad(tid)] = sum;
	int offset = 1;
	for (int d = kTopThreads >> 1; d > 0; d >>= 1) {
		__syncthreads();
		if (tid < d)
			tree[pad(offset * (2 * tid + 2) - 1)] += tree[pad(offset * (2 * tid + 1) - 1)];
		offset <<= 1;
	}
	__syncthreads();
	if (tid == 0) {
		a.misc[1] = (uint32_t)tree[pad(kTopThreads - 1)];
		tree[pad(kTopThreads - 1)] = 0;
	}
	for (int d = 1; d < kTopThreads; d <<= 1) {
		offset >>= 1;
		__syncthreads();
		if (tid < d) {
			const int ai = pad(offset * (2 * tid + 1) - 1), bi = pad(offset * (2 * tid + 2) - 1);
			const int32_t t = tree[ai];
			tree[ai] = tree[bi];
			tree[bi] += t;
		}
	}
	__syncthreads();
	int32_t run = tree[pad(tid)];
	for (uint32_t i = lo; i < hi; i++) {
		const int32_t v = a.off[i];
		a.off[i] = run;
		run += v;
	}
}

// Ordered scatter of both staging areas + header/trailer cells, one workgroup
// per 256 chains (the K2b block): scan the block's 256 counts in LDS on top
// of the block base, then every wave drains one K2 staging region and, while
// there are any, one K1 wave tile of this block.
template <int C>
__global__ __launch_bounds__(kBlock2) void k_scatter_all(ScanArgs a)
{
	__shared__ uint32_t off[kBlock2];
	__shared__ uint32_t wtot[kBlock2 / 64];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t j0 = blockIdx.x * kBlock2, j = j0 + tid;
	const uint32_t c = j < a.n_chains ? (uint32_t)a.cnt[j] : 0u;
	// in-wave inclusive scan by shuffles, then add the preceding waves
	uint32_t inc = c;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(inc, o, 64);
		if (lane >= (uint32_t)o)
			inc += t;
	}
	if (lane == 63)
		wtot[wv] = inc;
	__syncthreads();
	uint32_t before, total_records = 0;
	if (a.fold_blocks) {
		// no separate scan launch over the block totals: every block adds up the ones in front of
		// it (a few thousand L2-resident words), block 0 also all of them for the header cell
		__shared__ uint32_t part[kBlock2 / 64], part_all[kBlock2 / 64];
		uint32_t mine = 0, all = 0;
		const uint32_t upto = blockIdx.x == 0 ? a.fold_blocks : blockIdx.x;
		for (uint32_t i = tid; i < upto; i += kBlock2) {
			const uint32_t v = (uint32_t)a.off[i];
			all += v;
			mine += i < blockIdx.x ? v : 0u;
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			mine += __shfl_xor(mine, o, 64);
			all += __shfl_xor(all, o, 64);
		}
		if (lane == 0) {
			part[wv] = mine;
			part_all[wv] = all;
		}
		__syncthreads();
		before = 0;
		for (uint32_t w = 0; w < kBlock2 / 64; w++) {
			before += part[w];
			total_records += part_all[w];
		}
	} else {
		before = (uint32_t)a.off[blockIdx.x];
		if (blockIdx.x == 0 && tid == 0)
			total_records = a.misc[1];
	}
	for (uint32_t w = 0; w < wv; w++)
		before += wtot[w];
	off[tid] = before + inc - c;
	__syncthreads();

	auto drain = [&](const uint2 *stage, uint32_t nrec, bool from_k1) {
		for (uint32_t i = lane; i < nrec; i += 64) {
			const uint2 rec = stage[i];
			const uint32_t pos = rec.x, st = rec.y & 0xFFFFFFu, seq = rec.y >> 24;
			const uint32_t jj = pos >> a.logS;
			uint32_t d = off[jj - j0] + seq;
			if (from_k1 && !a.halo_mode) {   // (halo mode: nothing is ever dropped or staged by a resolve kernel)
				const uint32_t info = a.k2info[jj];
				if (info & 0x10000u)
					continue;
				d += info & 0xFFFFu;
			}
			if (d + 2 < a.plane_capacity) {
				a.pat_plane[1 + d] = a.out[st];
				a.off_plane[1 + d] = (int32_t)pos + a.off_shift;
			}
		}
	};
	const uint32_t gw = blockIdx.x * (kBlock2 / 64) + wv;            // K2 staging region
	if (gw * 64 < a.n_chains)
		drain(a.stage2 + (((size_t)gw * 64) << a.logS), a.wave_cnt2[gw], false);
	constexpr uint32_t kTilesPerBlock = kBlock2 / (C * 64);          // K1 wave tiles in this block
	if (wv < kTilesPerBlock) {
		const uint32_t wt = blockIdx.x * kTilesPerBlock + wv;
		if (wt < a.n_tiles)
			drain(a.stage1 + (((size_t)wt * C * 64) << a.logS), a.wave_cnt1[wt], true);
	}
	if (blockIdx.x == 0 && tid == 0) {   // header and trailer cells (compactarray.cl:49-55)
		const uint32_t total = total_records;
		const int32_t last_ref = (int32_t)a.dev2ref[a.misc[0]];
		a.misc[2] = (uint32_t)ACM_SCAN_MODE_CHAIN;   // acm_scan_path_taken
		uint32_t tail = total + 1;
		if (tail > a.plane_capacity - 1)
			tail = a.plane_capacity - 1;
		a.pat_plane[0] = (int32_t)total;
		a.off_plane[0] = (int32_t)total;
		a.pat_plane[tail] = last_ref;
		a.off_plane[tail] = last_ref;
	}
}

// A scan that continues another on the device (acm_scan_batch.d_init_plane): the final state of that scan is in the
// trailer cell of its pattern plane, behind the header cell that says where (compactarray.cl:49-55 layout;
// the reference hands it from round to round through the host, databuf.c:622).  One thread translates it
// into this device's numberings and leaves it in the workspace, where the kernels of the scan look for it.
__global__ void k_carry_init(const int32_t *plane, uint32_t capacity, const uint32_t *ref2dev, const uint32_t *ref2code,
    uint32_t states, uint32_t *dev_out, uint32_t *code_out)
{
	if (threadIdx.x != 0 || blockIdx.x != 0)
		return;
	const int32_t m = plane[0];
	uint32_t tail = (uint32_t)(m < 0 ? 0 : m) + 1;
	if (tail > capacity - 1)
		tail = capacity - 1;
	uint32_t ref = (uint32_t)plane[tail];
	if (ref >= states)
		ref = 0;   // (not a state: cannot happen with planes of this library)
	*dev_out = ref2dev[ref];
	*code_out = ref2code ? ref2code[ref] : 0u;
}

// empty text: header and trailer only
__global__ void k_finalize_empty(ScanArgs a)
{
	if (threadIdx.x != 0 || blockIdx.x != 0)
		return;
	const int32_t last_ref = (int32_t)a.dev2ref[init_of(a)];
	a.misc[2] = (uint32_t)ACM_SCAN_MODE_CHAIN;
	a.pat_plane[0] = 0;
	a.off_plane[0] = 0;
	a.pat_plane[1] = last_ref;
	a.off_plane[1] = last_ref;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Layout {
	size_t end_state, c1f, k2info, wend, probe, rflag, cnt, off, wave_cnt1, wave_cnt2, misc, stage1,
	    stage2, scan_ws, sparse;
	size_t scan_ws_bytes;
	size_t total;
};

// sized for the smallest chain length (16 B) so any geometry fits
Layout layout_for(const acm_dfa *d, size_t max_text)
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
	l.wend = take(chains * 4);
	l.probe = take(chains);
	l.rflag = take(chains);
	l.cnt = take(chains * 4);
	l.off = take(chains * 4);
	l.wave_cnt1 = take(waves * 4);
	l.wave_cnt2 = take(waves * 4);
	l.misc = take(64);
	l.stage1 = take(stage_recs * 8);
	l.stage2 = take(stage_recs * 8);
	l.scan_ws_bytes = acm_exclusive_scan_workspace_bytes(chains / kBlock2 + 2);
	l.scan_ws = take(l.scan_ws_bytes);
	l.sparse = take(d ? acm::sparse_workspace_bytes(d, max_text) : 0);
	l.total = o;
	return l;
}

template <int C>
int launch_spec_walk(const ScanArgs &a, int num_cus, hipStream_t s)
{
	const size_t lds = acm::kHotBytes + 256;   // rows + class map; above 48 KiB: allowed by acm::scan_prepare
	uint32_t blocks = (a.n_tiles + kWaves1 - 1) / kWaves1;
	if (blocks > (uint32_t)num_cus)
		blocks = (uint32_t)num_cus;
	if (C == 2 && a.halo_pre && !a.halo_mode) {   // (speculative mode, 64-byte chains: 16 waves of two chains per lane)
		uint32_t pblocks = (a.n_tiles + 1024 / 64 - 1) / (1024 / 64);
		if (pblocks > (uint32_t)num_cus)
			pblocks = (uint32_t)num_cus;
		if (a.ls == 8)
			hipLaunchKernelGGL((k_halo_walk<false, false, 2, 1024>), dim3(pblocks), dim3(1024), lds, s, a);
		else
			hipLaunchKernelGGL((k_halo_walk<true, false, 2, 1024>), dim3(pblocks), dim3(1024), lds, s, a);
	} else if (C == 4 && a.halo_pre) {
		uint32_t pblocks = (a.n_tiles + kBlockPre / 64 - 1) / (kBlockPre / 64);
		if (pblocks > (uint32_t)num_cus)
			pblocks = (uint32_t)num_cus;
		if (a.halo_mode) {
			if (a.ls == 8)
				hipLaunchKernelGGL((k_halo_walk<false, true>), dim3(pblocks), dim3(kBlockPre), lds, s, a);
			else
				hipLaunchKernelGGL((k_halo_walk<true, true>), dim3(pblocks), dim3(kBlockPre), lds, s, a);
		} else {
			if (a.ls == 8)
				hipLaunchKernelGGL((k_halo_walk<false, false>), dim3(pblocks), dim3(kBlockPre), lds, s, a);
			else
				hipLaunchKernelGGL((k_halo_walk<true, false>), dim3(pblocks), dim3(kBlockPre), lds, s, a);
		}
	} else if (a.halo_mode && C == 4) {
		if (a.ls == 8)
			hipLaunchKernelGGL((k_spec_walk<4, false, true>), dim3(blocks), dim3(kBlock1), lds, s, a);
		else
			hipLaunchKernelGGL((k_spec_walk<4, true, true>), dim3(blocks), dim3(kBlock1), lds, s, a);
	} else if (a.ls == 8) {
		hipLaunchKernelGGL((k_spec_walk<C, false, false>), dim3(blocks), dim3(kBlock1), lds, s, a);
	} else {
		hipLaunchKernelGGL((k_spec_walk<C, true, false>), dim3(blocks), dim3(kBlock1), lds, s, a);
	}
	ACM_HIP_TRY(hipGetLastError());
	return ACM_OK;
}

}  // namespace

extern "C" size_t acm_scan_workspace_bytes(const acm_dfa *d, size_t max_text)
{
	return layout_for(d, max_text).total;
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

extern "C" int acm_scan_set_chains_per_lane(acm_dfa *d, int chains)
{
	if (!d)
		return 0;
	if (chains == 2 || chains == 4)
		d->chains_per_lane = chains;
	return d->chains_per_lane;
}

extern "C" int acm_scan_kernel_count(void) { return 4; }

namespace {
// Which pipeline the next batch gets.  Tiny texts are not worth the sieve's tables.  In AUTO
// mode the choice adapts: the sparse pipeline is exact on any text but slow on one that is dense
// in matches or in flagged samples (its emit kernel counts such batches, sparse.hip), so when
// half of the last 16 sparse batches were dense the next 64 go to the chain pipeline, then the
// sparse one is tried again -- for 4 batches; if half of those are dense again the chain
// pipeline gets four times as many batches as last time (up to 4096), and so on until a look
// finds the text quiet.
bool pick_sparse(const acm_dfa *d, size_t n)
{
	if (!d->sparse_ok || d->scan_mode == ACM_SCAN_MODE_CHAIN || n < 64)
		return false;
	if (d->scan_mode != ACM_SCAN_MODE_AUTO || !d->h_giveups)
		return true;
	uint32_t hold = d->chain_hold.load(std::memory_order_relaxed);
	while (hold > 0)
		if (d->chain_hold.compare_exchange_weak(hold, hold - 1, std::memory_order_relaxed))
			return false;
	const uint32_t count = d->sparse_batches.fetch_add(1, std::memory_order_relaxed) + 1;
	const uint32_t window = d->auto_window.load(std::memory_order_relaxed);
	if (count >= window) {
		d->sparse_batches.store(0, std::memory_order_relaxed);
		const uint32_t seen = *(volatile uint32_t *)d->h_giveups;   // written by k_sieve_emit, may lag
		const uint32_t before = d->giveups_seen.exchange(seen, std::memory_order_relaxed);
		if (seen - before >= window / 2) {
			const uint32_t stay = d->auto_next_hold.load(std::memory_order_relaxed);
			d->chain_hold.store(stay, std::memory_order_relaxed);
			d->auto_next_hold.store(std::min<uint32_t>(stay * 4, 4096u), std::memory_order_relaxed);
			d->auto_window.store(4, std::memory_order_relaxed);
		} else {
			d->auto_next_hold.store(64, std::memory_order_relaxed);
			d->auto_window.store(16, std::memory_order_relaxed);
		}
	}
	return true;
}
}  // namespace

extern "C" int acm_scan_set_mode(acm_dfa *d, int mode)
{
	if (!d)
		return ACM_SCAN_MODE_CHAIN;
	if (mode == ACM_SCAN_MODE_AUTO || mode == ACM_SCAN_MODE_CHAIN || mode == ACM_SCAN_MODE_SPARSE)
		d->scan_mode = mode;
	return d->scan_mode;
}

extern "C" int acm_scan_sparse_eligible(const acm_dfa *d) { return d && d->sparse_ok ? 1 : 0; }
extern "C" int acm_scan_lds_resident(const acm_dfa *d) { return d && d->lds_ok && d->use_halo ? 1 : 0; }
extern "C" int acm_scan_group_capable(const acm_dfa *d)
{
	if (!d || d->use_graphs || d->max_group <= 1)
		return 0;
	const bool sparse = d->sparse_ok && d->scan_mode != ACM_SCAN_MODE_CHAIN;
	return (sparse || (d->lds_ok && d->use_halo)) ? 1 : 0;
}

extern "C" int acm_scan_path_taken(const acm_dfa *d, const void *d_workspace, size_t n, void *stream)
{
	if (!d || !d_workspace)
		return acm::fail(ACM_ERR_ARG, "acm_scan_path_taken: bad arguments");
	ACM_HIP_TRY(hipSetDevice(d->device));
	ACM_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
	if (n == 0)
		return ACM_SCAN_MODE_CHAIN;
	const Layout l = layout_for(d, n);
	uint32_t marker = 0;   // misc[2]: written by whichever pipeline produced the planes
	ACM_HIP_TRY(hipMemcpy(&marker, (const char *)d_workspace + l.misc + 8, 4, hipMemcpyDeviceToHost));
	return (int)marker;
}

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
	acm_scan_batch b;
	memset(&b, 0, sizeof(b));
	b.d_text = d_text;
	b.n = n;
	b.halo = halo;
	b.offset_shift = offset_shift;
	b.init_state = init_state;
	b.d_workspace = d_workspace;
	b.workspace_bytes = workspace_bytes;
	b.d_pat_plane = d_pat_plane;
	b.d_off_plane = d_off_plane;
	b.plane_capacity = plane_capacity;
	b.stream = stream;
	return acm_scan_batch_async(d, &b);
}

namespace {
bool pick_sparse(const acm_dfa *d, size_t n);
// what enqueue_batch leaves to the caller when it defers the launches of a batch that joins a group
struct Deferred {
	acm::SieveJob sieve;
	acm::LdsJob lds;
};
int enqueue_batch(const acm_dfa *d, const acm_scan_batch *batch, bool sparse, Deferred *defer);
// the chain pipeline's LDS-resident form takes launch groups too (lds_walk.hip)
bool lds_path(const acm_dfa *d, bool sparse, size_t n) { return !sparse && d->lds_ok && d->use_halo && n > 0; }

// consecutive sparse batches of one size on one stream, each with its own workspace and planes,
// that wait for nothing and are not timed: one group for the sparse kernels
bool groupable(const acm_dfa *d, const acm_scan_batch &b)
{
	return !d->profile && !b.wait_before_walk && !b.record_after_walk && !b.d_init_plane && b.n > 0;
}
bool joins(const acm_scan_batch *const *group, uint32_t m, const acm_scan_batch &b)
{
	if (b.stream != group[0]->stream || b.n != group[0]->n || (b.profile != 0) != (group[0]->profile != 0))
		return false;   // (a group is timed as a whole or not at all)
	for (uint32_t i = 0; i < m; i++)
		if (b.d_workspace == group[i]->d_workspace || b.d_pat_plane == group[i]->d_pat_plane ||
		    b.d_off_plane == group[i]->d_off_plane)
			return false;
	return true;
}
}  // namespace

extern "C" int acm_scan_batches_async(const acm_dfa *d, const acm_scan_batch *batches, size_t count)
{
	if (!batches && count)
		return acm::fail(ACM_ERR_ARG, "acm_scan_batches_async: null batches");
	if (!d || d->use_graphs || d->max_group <= 1) {
		for (size_t i = 0; i < count; i++) {
			const int rc = acm_scan_batch_async(d, &batches[i]);
			if (rc != ACM_OK)
				return rc;
		}
		return ACM_OK;
	}
	const uint32_t cap = std::min<uint32_t>((uint32_t)d->max_group, std::min(acm::sparse_max_group(), acm::lds_walk_max_group()));
	const acm_scan_batch *group[32];
	uint32_t m = 0;
	bool group_sparse = true;   // the pipeline of the group being collected
	// Failure contract: "stops at the first batch that fails; the batches before it stay enqueued" --
	// also inside a group: the members in front of the one that failed validation are launched (as a
	// shorter group), and events taken from the pool go back to it on every error path.
	auto flush = [&]() -> int {
		int rc = ACM_OK;
		const uint32_t members = m;
		m = 0;
		if (members == 1) {
			rc = enqueue_batch(d, group[0], group_sparse, nullptr);
		} else if (members > 1) {
			Deferred jobs[32];
			uint32_t good = 0;
			int first_bad = ACM_OK;
			for (; good < members; good++) {
				first_bad = enqueue_batch(d, group[good], group_sparse, &jobs[good]);
				if (first_bad != ACM_OK)
					break;
			}
			hipStream_t gs = (hipStream_t)group[0]->stream;
			hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
			auto give_back = [&]() {
				std::lock_guard<std::mutex> lock(d->profile_mutex);
				for (auto &e : ev)
					if (e) {
						d->profile_pool.push_back((void *)e);
						e = nullptr;
					}
			};
			if (good > 0 && group[0]->profile) {   // the group's kernels, timed like a single batch's
				std::lock_guard<std::mutex> lock(d->profile_mutex);
				for (auto &e : ev) {
					if (!d->profile_pool.empty()) {
						e = (hipEvent_t)d->profile_pool.back();
						d->profile_pool.pop_back();
					} else if (hipEventCreate(&e) != hipSuccess) {
						e = nullptr;
						rc = acm::fail(ACM_ERR_HIP, "acm_scan_batches_async: hipEventCreate failed");
						break;
					}
				}
			}
			if (rc != ACM_OK) {
				give_back();
				return rc;
			}
			if (good > 0) {
				if (ev[0] && hipEventRecord(ev[0], gs) != hipSuccess)
					rc = acm::fail(ACM_ERR_HIP, "acm_scan_batches_async: hipEventRecord failed");
				if (rc == ACM_OK && group_sparse) {
					acm::SieveJob sj[32];
					for (uint32_t i = 0; i < good; i++)
						sj[i] = jobs[i].sieve;
					rc = acm::sparse_group_enqueue(d, sj, good, gs, ev[1], ev[2]);
				} else if (rc == ACM_OK) {
					acm::LdsJob lj[32];
					for (uint32_t i = 0; i < good; i++)
						lj[i] = jobs[i].lds;
					rc = acm::lds_walk_enqueue(d, lj, good, gs, ev[1], nullptr);
					if (rc == ACM_OK && ev[2] && hipEventRecord(ev[2], gs) != hipSuccess)   // (the walk is the first stage, there is no second)
						rc = acm::fail(ACM_ERR_HIP, "acm_scan_batches_async: hipEventRecord failed");
				}
				if (rc == ACM_OK && ev[0] && hipEventRecord(ev[3], gs) != hipSuccess)
					rc = acm::fail(ACM_ERR_HIP, "acm_scan_batches_async: hipEventRecord failed");
				if (rc == ACM_OK && ev[0]) {
					std::lock_guard<std::mutex> lock(d->profile_mutex);
					for (auto &e : ev) {
						d->profile_events.push_back((void *)e);
						e = nullptr;
					}
				}
				give_back();   // (only what an error left behind)
			}
			if (rc == ACM_OK)
				rc = first_bad;
		}
		return rc;
	};
	for (size_t i = 0; i < count; i++) {
		const acm_scan_batch &b = batches[i];
		const bool sparse = pick_sparse(d, b.n);   // (counts the batch: once per batch)
		if ((sparse || lds_path(d, sparse, b.n)) && groupable(d, b)) {
			if (m && (m >= cap || sparse != group_sparse || !joins(group, m, b))) {
				const int rc = flush();
				if (rc != ACM_OK)
					return rc;
			}
			group_sparse = sparse;
			group[m++] = &b;
			continue;
		}
		int rc = flush();
		if (rc == ACM_OK)
			rc = enqueue_batch(d, &b, sparse, nullptr);
		if (rc != ACM_OK)
			return rc;
	}
	return flush();
}

namespace acm {
int scan_prepare(const acm_dfa *)
{
	const void *walks[] = { (const void *)k_spec_walk<4, false, false>, (const void *)k_spec_walk<4, true, false>,
		(const void *)k_spec_walk<2, false, false>, (const void *)k_spec_walk<2, true, false>,
		(const void *)k_spec_walk<4, false, true>, (const void *)k_spec_walk<4, true, true>,
		(const void *)k_halo_walk<false, true>, (const void *)k_halo_walk<true, true>,
		(const void *)k_halo_walk<false, false>, (const void *)k_halo_walk<true, false>,
		(const void *)k_halo_walk<false, false, 2, 1024>, (const void *)k_halo_walk<true, false, 2, 1024> };
	for (const void *k : walks)
		ACM_HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(acm::kHotBytes + 256)));
	return ACM_OK;
}
}  // namespace acm

namespace {
int enqueue_batch(const acm_dfa *d, const acm_scan_batch *batch, bool sparse, Deferred *defer = nullptr);

// what a cached graph was captured for: every input of enqueue_batch except the stream
acm_dfa::GraphKey graph_key(const acm_dfa *d, const acm_scan_batch *b, bool sparse)
{
	acm_dfa::GraphKey k;
	memset(&k, 0, sizeof(k));
	k.text = b->d_text;
	k.n = b->n;
	k.halo = b->halo;
	k.offset_shift = b->offset_shift;
	k.init_state = b->init_state;
	k.workspace = b->d_workspace;
	k.workspace_bytes = b->workspace_bytes;
	k.pat_plane = b->d_pat_plane;
	k.off_plane = b->d_off_plane;
	k.plane_capacity = b->plane_capacity;
	k.report = b->report;
	k.init_plane = b->d_init_plane;
	k.init_plane_capacity = b->init_plane_capacity;
	k.mode = sparse ? ACM_SCAN_MODE_SPARSE : ACM_SCAN_MODE_CHAIN;
	k.chain_bytes = d->chain_bytes;
	k.chains_per_lane = d->chains_per_lane;
	return k;
}
}  // namespace

// The kernels of one scan are short and many; a host that scans with the same
// buffers over and over (a worker with its staging buffers, as the reference's
// workers do) pays more for launching them than the GPU for running them.  So
// the enqueue of a batch that repeats is captured once into a HIP graph and
// replayed with one hipGraphLaunch.
extern "C" int acm_scan_batch_async(const acm_dfa *d, const acm_scan_batch *batch)
{
	if (!batch)
		return acm::fail(ACM_ERR_ARG, "acm_scan_batch_async: null batch");
	const bool sparse = d && pick_sparse(d, batch->n);
	if (!d || !d->use_graphs || d->profile || batch->profile || !batch->stream || batch->wait_before_walk ||
	    batch->record_after_walk || batch->n == 0)
		return enqueue_batch(d, batch, sparse);
	hipStream_t s = (hipStream_t)batch->stream;
	const acm_dfa::GraphKey key = graph_key(d, batch, sparse);
	hipGraphExec_t exec = nullptr;
	bool capture = false;
	{
		std::lock_guard<std::mutex> lock(d->graph_mutex);
		acm_dfa::GraphEntry *e = nullptr;
		for (auto &g : d->graphs)
			if (!memcmp(&g.key, &key, sizeof(key)))
				e = &g;
		if (!e) {   // first sighting: remember it, enqueue the plain way
			if (d->graphs.size() >= acm_dfa::kMaxGraphs) {
				size_t oldest = 0;
				for (size_t i = 1; i < d->graphs.size(); i++)
					if (d->graphs[i].last_use < d->graphs[oldest].last_use)
						oldest = i;
				// its last launch may still be running: an exec is only ever destroyed by
				// acm_dfa_release; an evicted one is parked until then
				if (d->graphs[oldest].exec)
					d->parked_graphs.push_back(d->graphs[oldest].exec);
				d->graphs.erase(d->graphs.begin() + (long)oldest);
			}
			acm_dfa::GraphEntry fresh;
			fresh.key = key;
			fresh.exec = nullptr;
			fresh.last_use = ++d->graph_tick;
			d->graphs.push_back(fresh);
		} else {
			e->last_use = ++d->graph_tick;
			exec = (hipGraphExec_t)e->exec;
			capture = !exec;
		}
	}
	if (exec) {
		ACM_HIP_TRY(hipSetDevice(d->device));
		ACM_HIP_TRY(hipGraphLaunch(exec, s));
		return ACM_OK;
	}
	if (!capture)
		return enqueue_batch(d, batch, sparse);
	// second sighting: capture.  Argument errors surface here exactly as in the plain path.
	ACM_HIP_TRY(hipSetDevice(d->device));
	if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
		(void)hipGetLastError();
		d->use_graphs = false;   // e.g. the caller is capturing this stream itself
		return enqueue_batch(d, batch, sparse);
	}
	const int rc = enqueue_batch(d, batch, sparse);
	hipGraph_t graph = nullptr;
	const hipError_t end = hipStreamEndCapture(s, &graph);
	if (rc != ACM_OK || end != hipSuccess || !graph) {
		if (graph)
			hipGraphDestroy(graph);
		(void)hipGetLastError();
		d->use_graphs = false;
		return rc != ACM_OK ? rc : enqueue_batch(d, batch, sparse);
	}
	const hipError_t inst = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
	hipGraphDestroy(graph);
	if (inst != hipSuccess || !exec) {
		(void)hipGetLastError();
		d->use_graphs = false;
		return enqueue_batch(d, batch, sparse);
	}
	{
		std::lock_guard<std::mutex> lock(d->graph_mutex);
		bool stored = false;
		for (auto &g : d->graphs)
			if (!memcmp(&g.key, &key, sizeof(key)) && !g.exec) {
				g.exec = (void *)exec;
				stored = true;
			}
		if (!stored) {   // evicted, or another thread was quicker
			ACM_HIP_TRY(hipGraphLaunch(exec, s));
			d->parked_graphs.push_back((void *)exec);   // cannot be destroyed while in flight
			return ACM_OK;
		}
	}
	ACM_HIP_TRY(hipGraphLaunch(exec, s));
	return ACM_OK;
}

extern "C" int acm_scan_set_max_group(acm_dfa *d, int batches)
{
	if (!d)
		return 1;
	if (batches >= 1)
		d->max_group = std::min<int>(batches, (int)acm::sparse_max_group());
	return d->max_group;
}

extern "C" int acm_scan_set_graphs(acm_dfa *d, int enable)
{
	if (!d)
		return 0;
	if (enable >= 0)
		d->use_graphs = enable != 0;
	return d->use_graphs ? 1 : 0;
}

namespace {
// defer: (sparse pipeline, LDS walk) check and lay out as always, but leave the launches to the caller,
// who enqueues a group of such batches with one set of kernels (acm_scan_batches_async)
int enqueue_batch(const acm_dfa *d, const acm_scan_batch *batch, bool sparse, Deferred *defer)
{
	const void *d_text = batch->d_text;
	const size_t n = batch->n, halo = batch->halo;
	const long offset_shift = batch->offset_shift, init_state = batch->init_state;
	void *d_workspace = batch->d_workspace;
	const size_t workspace_bytes = batch->workspace_bytes;
	int32_t *d_pat_plane = batch->d_pat_plane, *d_off_plane = batch->d_off_plane;
	size_t plane_capacity = batch->plane_capacity;
	void *stream = batch->stream;
	if (!d || !d_pat_plane || !d_off_plane || plane_capacity < 2 || (n && !d_text))
		return acm::fail(ACM_ERR_ARG, "acm_scan_async: bad arguments");
	if (n > 0x7FFFFFEFul)
		return acm::fail(ACM_ERR_LIMIT, "acm_scan_async: %zu bytes exceed the 2 GiB buffer limit", n);
	if (((uintptr_t)d_text & 15) != 0)
		return acm::fail(ACM_ERR_ARG, "acm_scan_async: text must be 16-byte aligned");
	if (batch->report != ACM_REPORT_HEAD && batch->report != ACM_REPORT_STATE)
		return acm::fail(ACM_ERR_ARG, "acm_scan_batch_async: report %d is not an ACM_REPORT_* value", batch->report);
	if (halo > n || offset_shift < INT32_MIN || offset_shift > INT32_MAX ||
	    (long)n + offset_shift > (long)INT32_MAX)
		return acm::fail(ACM_ERR_ARG, "acm_scan_shard_async: halo/offset_shift out of range");
	if (batch->d_init_plane && batch->init_plane_capacity < 2)
		return acm::fail(ACM_ERR_ARG, "acm_scan_batch_async: d_init_plane needs the capacity its scan was given");
	if (init_state < 0 || (uint64_t)init_state >= d->num_states)
		return acm::fail(ACM_ERR_ARG, "acm_scan_async: init_state %ld is not a state", init_state);
	if (plane_capacity > 0xFFFFFFFFul)
		plane_capacity = 0xFFFFFFFFul;
	const Layout l = layout_for(d, n);
	if (!d_workspace || workspace_bytes < l.total)
		return acm::fail(ACM_ERR_ARG, "acm_scan_async: workspace %zu B < required %zu B",
		    workspace_bytes, l.total);
	hipStream_t s = (hipStream_t)stream;
	ACM_HIP_TRY(hipSetDevice(d->device));

	// geometry: enough chains to give every lane of every CU work, chains
	// as long as that allows (longer chains = fewer look-back steps)
	int C = d->chains_per_lane == 2 ? 2 : 4;
	uint32_t S = (uint32_t)d->chain_bytes;
	bool wide_pre = false;   // speculative mode with 64-byte chains: two chains per lane, 16 waves per CU, text loaded up front
	if (S == 0) {
		const size_t lanes = (size_t)d->num_cus * kBlock1 * C;
		S = 16;
		while (S < 256 && (size_t)S * 2 * lanes <= n)
			S *= 2;
		// halo mode walks L - 1 bytes per chain twice: with half the waves of every CU still busy a
		// chain of four halos is the better deal when batches are in flight side by side (sentiment
		// set, 32 MiB: 650 instead of 574 GB/s; alone the walk takes 89 instead of 56 us)
		const uint32_t hb = ((d->max_pattern_len > 1 ? d->max_pattern_len - 1 : 0u) + 15u) & ~15u;
		uint32_t want = 16;
		while (want < 4 * hb)
			want *= 2;
		if (d->use_halo && C == 4 && hb > 0 && hb <= S && want > S && want <= 256 && (size_t)want * lanes <= 2 * n)
			S = want;
		// speculative mode (patterns longer than a chain): a chain's look-back window is ceil(L / S) chains, and
		// it is the probe and resolve kernels' slowest lanes -- chains of dependent loads through that window --
		// that those kernels take as long as.  Chains of a third of the longest pattern: half the chains of the
		// same text at 32 MiB, i.e. half of every CU's lanes idle in the walk kernel -- which batches in flight
		// side by side fill (ClamAV signatures, 3 streams: 635 instead of 489 GB/s; a batch alone 111 instead of
		// 98 us).  Large texts only: a small one is short of chains as it is.
		if (!(d->use_halo && hb <= S) && n >= ((size_t)16 << 20)) {
			while (S < 256 && (d->max_pattern_len + S - 1) / S > 3)
				S *= 2;
			static const bool no_wide = getenv("ACM_SCAN_NO_WIDE_PRE") != nullptr;   // debugging aid
			if (S == 64 && C == 4 && d->use_preload && !no_wide) {
				wide_pre = true;
				C = 2;
			}
		}
	}
	uint32_t logS = 0;
	while ((1u << logS) < S)
		logS++;

	char *ws = (char *)d_workspace;
	ScanArgs a;
	memset(&a, 0, sizeof(a));
	a.cold = d->d_cold;
	a.deep = d->d_deep;
	a.hot = d->d_hot;
	a.cls = d->d_class;
	a.ls = d->log_stride;
	// the plane value of a record is a per-state table lookup: the head pattern, or the state's
	// reference id when the caller wants to expand the whole match list afterwards
	a.out = batch->report == ACM_REPORT_STATE ? (const int32_t *)d->d_dev2ref : d->d_out;
	a.dev2ref = d->d_dev2ref;
	a.in_byte = d->d_in_byte;
	a.text16 = (const uint4 *)d_text;
	a.text = (const uint8_t *)d_text;
	a.n = (uint32_t)n;
	a.n_pad = (uint32_t)((n + 15) & ~(size_t)15);
	a.S = S;
	a.logS = logS;
	a.n_chains = (uint32_t)((n + S - 1) >> logS);
	a.n_tiles = (a.n_chains + C * 64 - 1) / (C * 64);
	a.H = d->hot_rows;
	a.hot_depth1 = d->hot_depth1;
	a.F = d->first_final;
	a.L = d->max_pattern_len;
	a.q = (a.L + S - 1) / S;
	if (a.q == 0)
		a.q = 1;
	a.init_state = d->ref2dev[(size_t)init_state];
	{
		// halo mode (walk_tile): the longest pattern fits the chain -- no speculation to resolve
		const uint32_t hb = ((a.L > 1 ? a.L - 1 : 0u) + 15u) & ~15u;
		a.halo_mode = (d->use_halo && C == 4 && hb <= S) ? 1u : 0u;
		a.halo_bytes = a.halo_mode ? hb : 0u;
		a.halo_pre = (a.halo_mode && ((S + hb) >> 4) <= (uint32_t)kPreGroups && d->use_preload) ? 1u : 0u;
		if (!a.halo_mode && S == 64 && d->use_preload && (C == 4 || wide_pre))
			a.halo_pre = 1u;   // (speculative mode, chains of 64 bytes: k_halo_walk<CLS, false, ...>)
	}
	a.drop_before = (uint32_t)halo;
	a.off_shift = (int32_t)offset_shift;
	a.end_state = (uint32_t *)(ws + l.end_state);
	a.c1f = (uint32_t *)(ws + l.c1f);
	a.k2info = (uint32_t *)(ws + l.k2info);
	a.wend = (uint32_t *)(ws + l.wend);
	a.probe = (uint8_t *)(ws + l.probe);
	a.rflag = (uint8_t *)(ws + l.rflag);
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

	const bool empty = a.n_chains == 0;   // (header and trailer only: below, once the state to start in is known)

	hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
	const bool profile = !defer && (d->profile || batch->profile);   // (a deferred batch is timed with its group)
	if (profile) {
		std::lock_guard<std::mutex> lock(d->profile_mutex);
		for (auto &e : ev) {
			if (!d->profile_pool.empty()) {  // recycled: no create/destroy in a timed loop
				e = (hipEvent_t)d->profile_pool.back();
				d->profile_pool.pop_back();
			} else {
				ACM_HIP_TRY(hipEventCreate(&e));
			}
		}
	}
	if (batch->wait_before_walk)
		ACM_HIP_TRY(hipStreamWaitEvent(s, (hipEvent_t)batch->wait_before_walk, 0));
	if (profile)
		ACM_HIP_TRY(hipEventRecord(ev[0], s));
	int rc;
	const uint32_t *init_dev_ptr = nullptr, *init_code_ptr = nullptr;
	if (batch->d_init_plane) {   // the state to start in comes from another scan's planes, on the device
		if (defer)
			return acm::fail(ACM_ERR_ARG, "acm_scan_batches_async: a batch with d_init_plane cannot join a launch group");
		hipLaunchKernelGGL(k_carry_init, dim3(1), dim3(64), 0, s, batch->d_init_plane,
		    (uint32_t)std::min<size_t>(batch->init_plane_capacity, 0xFFFFFFFFul), (const uint32_t *)d->d_ref2dev,
		    (const uint32_t *)(d->lds_ok ? d->d_lds_ref2code : nullptr), d->num_states, a.misc + 4, a.misc + 5);
		ACM_HIP_TRY(hipGetLastError());
		init_dev_ptr = a.misc + 4;
		init_code_ptr = a.misc + 5;
		a.init_ptr = init_dev_ptr;
	}
	if (empty) {
		hipLaunchKernelGGL(k_finalize_empty, dim3(1), dim3(64), 0, s, a);
		ACM_HIP_TRY(hipGetLastError());
		if (profile) {
			for (int k = 1; k < 4; k++)
				ACM_HIP_TRY(hipEventRecord(ev[k], s));
			std::lock_guard<std::mutex> lock(d->profile_mutex);
			for (auto e : ev)
				d->profile_events.push_back((void *)e);
		}
		return ACM_OK;
	}
	if (sparse && defer) {
		defer->sieve.batch = batch;
		defer->sieve.init_dev = a.init_state;
		defer->sieve.init_ptr = nullptr;
		defer->sieve.sparse_ws = ws + l.sparse;
		defer->sieve.path_marker = a.misc + 2;
		return ACM_OK;
	}
	if (lds_path(d, sparse, n)) {   // the automaton fits the LDS whole: walk + scatter of lds_walk.hip
		acm::LdsJob job;
		job.batch = batch;
		job.stage = (uint32_t *)(ws + l.stage1);
		job.cnt = (uint8_t *)(ws + l.cnt);
		job.tile_total = (uint32_t *)(ws + l.off);
		job.misc = a.misc;
		job.init_ptr = init_code_ptr;
		size_t stage_words, cnt_bytes, tile_words;
		acm::lds_walk_needs(d, n, &stage_words, &cnt_bytes, &tile_words);
		if (stage_words * 4 > l.stage2 - l.stage1 || cnt_bytes > l.off - l.cnt || tile_words * 4 > l.wave_cnt1 - l.off)
			return acm::fail(ACM_ERR_ARG, "acm_scan_async: workspace layout too small for the LDS walk");
		if (defer) {
			defer->lds = job;
			return ACM_OK;
		}
		rc = acm::lds_walk_enqueue(d, &job, 1, s, profile ? ev[1] : nullptr, (hipEvent_t)batch->record_after_walk);
		if (rc != ACM_OK)
			return rc;
		if (profile) {
			ACM_HIP_TRY(hipEventRecord(ev[2], s));
			ACM_HIP_TRY(hipEventRecord(ev[3], s));
			std::lock_guard<std::mutex> lock(d->profile_mutex);
			for (auto e : ev)
				d->profile_events.push_back((void *)e);
		}
		return ACM_OK;
	}
	if (sparse) {   // three kernels of its own; it always produces the planes
		rc = acm::sparse_scan_enqueue(d, batch, a.init_state, init_dev_ptr, ws + l.sparse, a.misc + 2, s, ev[1], ev[2]);
		if (rc != ACM_OK)
			return rc;
		if (batch->record_after_walk)
			ACM_HIP_TRY(hipEventRecord((hipEvent_t)batch->record_after_walk, s));
		if (profile) {
			ACM_HIP_TRY(hipEventRecord(ev[3], s));
			std::lock_guard<std::mutex> lock(d->profile_mutex);
			for (auto e : ev)
				d->profile_events.push_back((void *)e);
		}
		return ACM_OK;
	}
	rc = C == 4 ? launch_spec_walk<4>(a, d->num_cus, s) : launch_spec_walk<2>(a, d->num_cus, s);
	if (rc != ACM_OK)
		return rc;
	if (batch->record_after_walk)
		ACM_HIP_TRY(hipEventRecord((hipEvent_t)batch->record_after_walk, s));
	if (profile) {   // the walk is the first stage, there is no second
		ACM_HIP_TRY(hipEventRecord(ev[1], s));
		ACM_HIP_TRY(hipEventRecord(ev[2], s));
	}
	const uint32_t nb = (a.n_chains + kBlock2 - 1) / kBlock2;   // K2 blocks == scatter blocks
	if (!a.halo_mode) {
		hipLaunchKernelGGL(k_probe, dim3(nb), dim3(kBlock2), 0, s, a);
		hipLaunchKernelGGL(k_resolve, dim3(nb), dim3(kBlock2), 0, s, a);
		ACM_HIP_TRY(hipGetLastError());
	}
	if (nb <= kFoldMax) {
		a.fold_blocks = nb;   // the scatter kernel sums the block totals itself: one launch less
	} else if (nb <= kTopMax) {
		hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kTopThreads), 0, s, a, nb);
	} else {  // > 16M chains: generic multi-level scan of the block totals
		rc = acm_exclusive_scan_i32(a.off, a.off, nb, (int32_t *)(a.misc + 1), ws + l.scan_ws,
		    l.scan_ws_bytes, s);
		if (rc != ACM_OK)
			return rc;
	}
	static_assert(kBlock2 % (4 * 64) == 0, "a scatter block must hold whole K1 wave tiles");
	if (C == 4)
		hipLaunchKernelGGL(k_scatter_all<4>, dim3(nb), dim3(kBlock2), 0, s, a);
	else
		hipLaunchKernelGGL(k_scatter_all<2>, dim3(nb), dim3(kBlock2), 0, s, a);
	ACM_HIP_TRY(hipGetLastError());
	if (profile) {
		ACM_HIP_TRY(hipEventRecord(ev[3], s));
		std::lock_guard<std::mutex> lock(d->profile_mutex);
		for (auto e : ev)
			d->profile_events.push_back((void *)e);
	}
	return ACM_OK;
}

}  // namespace

extern "C" int acm_scan_profile_enable(acm_dfa *d, int enable)
{
	if (!d)
		return acm::fail(ACM_ERR_ARG, "acm_scan_profile_enable: null dfa");
	d->profile = enable != 0;
	return ACM_OK;
}

extern "C" int acm_scan_profile_read(acm_dfa *d, double *first_ms, double *second_ms, double *pipeline_ms,
    int *launches)
{
	if (!d)
		return acm::fail(ACM_ERR_ARG, "acm_scan_profile_read: null dfa");
	double first = 0, second = 0, pipe = 0;
	int n = 0;
	std::lock_guard<std::mutex> lock(d->profile_mutex);
	for (size_t i = 0; i + 3 < d->profile_events.size(); i += 4) {
		hipEvent_t e[4];
		for (int k = 0; k < 4; k++)
			e[k] = (hipEvent_t)d->profile_events[i + k];
		float a = 0, b = 0, c = 0;
		ACM_HIP_TRY(hipEventSynchronize(e[3]));
		ACM_HIP_TRY(hipEventElapsedTime(&a, e[0], e[1]));
		ACM_HIP_TRY(hipEventElapsedTime(&b, e[1], e[2]));
		ACM_HIP_TRY(hipEventElapsedTime(&c, e[0], e[3]));
		first += a;
		second += b;
		pipe += c;
		n++;
		for (int k = 0; k < 4; k++)
			d->profile_pool.push_back((void *)e[k]);
	}
	d->profile_events.clear();
	if (first_ms) *first_ms = first;
	if (second_ms) *second_ms = second;
	if (pipeline_ms) *pipeline_ms = pipe;
	if (launches) *launches = n;
	return ACM_OK;
}
