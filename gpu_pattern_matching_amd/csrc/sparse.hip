// The sparse scan pipeline: trigram filter -> candidate walks with in-kernel
// dedup, count and ordered scatter.  gfx950 only.  Same contract and output as
// the chain pipeline of scan.hip (it is the second implementation of
// acm_scan_*_async, chosen for pattern sets whose shortest pattern has at
// least 3 bytes); scan.hip stays the general path and the in-enqueue fallback.
//
// Why it is exact.  The serial DFA state at text position k is the longest
// suffix of the text that is a trie node.  While that depth is <= 2 the state
// is a pure function of the last two bytes (T2).  Depth can only grow by one
// per byte, so every maximal stretch of positions with depth >= 3 (a "deep
// run") starts at a position whose last three bytes ARE a depth-3 trie node.
// With patterns of >= 3 bytes every final state has depth >= 3: all records
// lie inside deep runs.  So:
//
//   SF  k_sparse_filter   every position tests its trigram against a Bloom
//                         filter of the depth-3 nodes held in LDS (no false
//                         negatives).  No state, no dependent load, fully
//                         coalesced text reads: this is the bulk pass.
//                         Output: one candidate bit per text byte.
//   SW  k_sparse_walk     one walker per run of consecutive candidate bits:
//                         start in T2[bytes a-2, a-1] at position a and walk
//                         the DFA exactly (deep plane, fast-forward)
//                         while the state is deep or the next position is a
//                         candidate.  A walker that started while an earlier
//                         deep run was still alive walks suffix-states of the
//                         true ones until that run ends, and exact states
//                         after it.  Hence the rule: walker j keeps its hits
//                         at positions > M_j, the largest deep extent of the
//                         walkers before it in position order -- exactly the
//                         part of its walk no earlier walker covers.  M_j is
//                         an exclusive prefix max; the part of it inside the
//                         workgroup (shuffles within a wave, LDS across
//                         waves) is applied here.  Hits stay in LDS while the
//                         walkers run (on gfx9 a store in flight holds up the
//                         next dependent load); at the end the workgroup
//                         writes its surviving hits in position order and its
//                         largest deep extent.
//   SE  k_sparse_emit     one workgroup: prefix max of the workgroup extents,
//                         drops the hits an earlier workgroup's walker covers
//                         (a prefix of each sorted list), prefix sum of what
//                         is left, ordered scatter into the planes, header
//                         and trailer cells.  A few thousand values: latency,
//                         not bandwidth.
//
// The state carried into the buffer (init_state) is handled by a walker at
// position 0 that starts from it.  Work is capped -- table steps per walker,
// walker rounds per 8192 positions, hits per 32768 positions; texts beyond the caps
// (one endless deep run, a match at every byte) raise a device flag and the
// chain pipeline, enqueued right behind and otherwise a row of early-exit
// launches, produces the planes instead.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "acm_internal.h"
#include "deep_walk.h"
#include "device_dfa.h"
#include "sparse.h"

namespace {

using acm_dev::ChainText;
using acm_dev::Deep;
using acm_dev::deep_step;
using acm_dev::fast_forward;

constexpr int kFilterBlock = 1024;
constexpr int kWalkBlock = 256;            // threads of a SW workgroup
constexpr int kWalkWaves = kWalkBlock / 64;
constexpr uint32_t kLaneWords = 2;         // words (of 64 text positions) per lane (1, 2, 4 measured: 2 is best) ...
constexpr uint32_t kWaveWords = 64 * kLaneWords;             // ... per wave: 8192 positions
constexpr uint32_t kBlockWords = kWalkWaves * kWaveWords;    // ... per workgroup: 32768 positions
constexpr uint32_t kIterCap = 256;         // table steps one walker may take ...
constexpr uint32_t kRunCap = 1024;         // ... or skip this many repeats of one byte in a self-looping state
constexpr uint32_t kForwardCap = 4096;     // ... each followed by this many fast-forwarded bytes at most
constexpr uint32_t kMaxWalkWave = 2048;    // walkers per wave (8192 positions): 32 rounds of 64
constexpr uint32_t kMaxHits = 256;         // staged hits per workgroup (32768 positions)

struct SparseArgs {
	const uint64_t *deep;     // [states][256] next | depth(next) << 32 | run(next) << 48
	const int32_t *out;
	const uint32_t *dev2ref;
	const uint8_t *in_byte;
	const uint32_t *bloom;    // [bloom_words] blocked Bloom filter of the depth-3 trigrams
	uint32_t bloom_words, bloom_log_words;
	const uint32_t *t2g;      // [65536] state after bytes (p, c) from the root, index p | c << 8
	const uint4 *text16;
	const uint8_t *text;
	uint32_t n, n_pad;
	uint32_t F;
	uint32_t init_state;
	uint32_t drop_before;
	int32_t off_shift;
	uint32_t nwords, nblocks;
	// workspace
	uint16_t *mask;                // [n_pad / 16 + 8] candidate bits
	uint32_t *block_extent;        // [nblocks] largest deep extent + 1 of the workgroup's walkers
	uint32_t *block_hits;          // [nblocks] hits the workgroup staged
	uint2 *hit_list;               // [nblocks][kMaxHits] {position, pattern}, ascending positions
	uint32_t *flags;               // [0] gave up -> chain pipeline runs
	uint32_t *path_marker;         // acm_scan_path_taken: set to SPARSE here, overwritten by the chain kernels
	uint32_t *giveups;             // host-visible count of batches given up on (adaptive AUTO mode), or null
	unsigned long long *keeper;    // start position << 32 | state of the first walker that reached the end
	// output
	int32_t *pat_plane, *off_plane;
	uint32_t plane_capacity;
};

// ------------------------------------------------------------------ SF ---

// bit 0 of the result: both filter bits of the trigram (low 24 bits of tri) are set
__device__ __forceinline__ uint32_t bloom_test(const uint32_t *bloom, uint32_t tri, uint32_t word_shift)
{
	const uint32_t p1 = __umul24(tri, acm::kBloomMul1), p2 = __umul24(tri, acm::kBloomMul2);
	const uint32_t w = bloom[p1 >> word_shift];
	return (w >> (p2 >> 27)) & (w >> ((p2 >> 22) & 31));
}

// shifts the candidate bit of position K of the 16-byte group into m from the top
template <int K>
__device__ __forceinline__ uint32_t probe(const uint32_t *bloom, uint32_t word_shift, const uint32_t (&x)[5],
    uint32_t m)
{
	// bytes (K-2, K-1, K) of the group; x[0] is the dword in front of it
	constexpr int lo = (K + 2) / 4, sh = (K + 2) % 4;
	const uint32_t v = sh == 0 ? x[lo] : __builtin_amdgcn_alignbyte(x[lo + 1 > 4 ? 4 : lo + 1], x[lo], sh);
	return __builtin_amdgcn_alignbit(bloom_test(bloom, v, word_shift), m, 1);
}

// persistent workgroups keep the filter in LDS; a wave-iteration reads 1 KiB of
// text with one coalesced 16 B/lane load, issued one iteration ahead of its use
__global__ __launch_bounds__(kFilterBlock) void k_sparse_filter(SparseArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t bloom[];
	{
		const uint4 *src = (const uint4 *)a.bloom;
		uint4 *dst = (uint4 *)bloom;
		const uint32_t n16 = a.bloom_words / 4;
		const uint32_t rot = (blockIdx.x * 1021u) % n16;
		for (uint32_t i = threadIdx.x; i < n16; i += kFilterBlock) {
			uint32_t j = i + rot;
			j = j >= n16 ? j - n16 : j;
			dst[j] = src[j];
		}
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) {   // first kernel of the pipeline: reset what the others accumulate into
		a.flags[0] = 0;
		*a.keeper = ~0ull;
		*a.path_marker = (uint32_t)ACM_SCAN_MODE_SPARSE;
	}
	__syncthreads();
	const uint32_t word_shift = 32 - a.bloom_log_words;
	const uint32_t n16 = a.n_pad >> 4;
	const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * (kFilterBlock / 64) + (threadIdx.x >> 6);
	const uint32_t nw = gridDim.x * (kFilterBlock / 64);
	const uint32_t *text32 = (const uint32_t *)a.text16;
	const uint32_t tiles = (n16 + 8 + 63) / 64;   // wave-iterations, including the 8 padding cells
	uint4 w = make_uint4(0, 0, 0, 0);
	uint32_t prev = 0;
	if (wave < tiles && wave * 64 + lane < n16) {
		w = a.text16[wave * 64 + lane];
		prev = wave * 64 + lane ? text32[(wave * 64 + lane) * 4 - 1] : 0u;
	}
	for (uint32_t t = wave; t < tiles; t += nw) {
		const uint32_t i16 = t * 64 + lane;
		const uint32_t x[5] = { prev, w.x, w.y, w.z, w.w };
		const uint32_t i16n = (t + nw) * 64 + lane;   // next iteration's group: load now, use then
		if (t + nw < tiles && i16n < n16) {
			w = a.text16[i16n];
			prev = text32[i16n * 4 - 1];
		}
		if (i16 >= n16) {
			if (i16 < n16 + 8)
				a.mask[i16] = 0;   // padding the walkers may read
			continue;
		}
		uint32_t m = 0;
		m = probe<0>(bloom, word_shift, x, m); m = probe<1>(bloom, word_shift, x, m);
		m = probe<2>(bloom, word_shift, x, m); m = probe<3>(bloom, word_shift, x, m);
		m = probe<4>(bloom, word_shift, x, m); m = probe<5>(bloom, word_shift, x, m);
		m = probe<6>(bloom, word_shift, x, m); m = probe<7>(bloom, word_shift, x, m);
		m = probe<8>(bloom, word_shift, x, m); m = probe<9>(bloom, word_shift, x, m);
		m = probe<10>(bloom, word_shift, x, m); m = probe<11>(bloom, word_shift, x, m);
		m = probe<12>(bloom, word_shift, x, m); m = probe<13>(bloom, word_shift, x, m);
		m = probe<14>(bloom, word_shift, x, m); m = probe<15>(bloom, word_shift, x, m);
		m >>= 16;
		if (i16 == 0)
			m &= ~3u;                        // no full trigram yet: the position-0 walker covers these
		const uint32_t pos0 = i16 << 4;
		if (pos0 + 16 > a.n)                 // bytes past the end of the text
			m &= (1u << (a.n - pos0)) - 1u;
		a.mask[i16] = (uint16_t)m;
	}
}

// ------------------------------------------------------------------ SE ---

// hits of one workgroup, in LDS until they are written to the planes
struct HitList {
	uint32_t pos[kMaxHits];
	uint32_t state[kMaxHits];
	uint32_t owner[kMaxHits];   // wave << 16 | round << 8 | lane of the walker; bit 31: dropped
	int32_t covered[kMaxHits];  // largest deep extent of the wave's walkers in front of that walker
	uint32_t count;             // may run past kMaxHits: then the workgroup gave up
};

// One walker: from the state before position p0 over p0, p0 + 1, ... while the
// state is deep or the next position is a candidate.  end_out: the last deep
// position (-1: none).  Returns false when a cap was hit.
__device__ __forceinline__ bool walker(const SparseArgs &a, HitList &hits, const uint64_t *lmask, uint32_t word0,
    uint32_t owner, uint32_t p0, int32_t &end_out)
{
	uint32_t state, min_steps = 0;
	if (p0 == 0) {
		state = a.init_state;
		min_steps = 2;   // positions 0 and 1 have no candidate bit of their own
	} else {
		state = a.t2g[(uint32_t)a.text[p0 - 2] | ((uint32_t)a.text[p0 - 1] << 8)];
	}
	int32_t end = -1;
	uint32_t p = p0, iters = 0;
	const uint32_t tbase = p0 & ~15u;
	ChainText txt(a, tbase);
	bool reached_end = false, ok = true;
	for (;;) {
		if (p >= a.n) {
			reached_end = true;
			break;
		}
		const uint32_t byte = txt.at(p - tbase + 1);
		Deep d = deep_step(a, state, byte);
		if (d.s == state && d.depth >= 3 && state < a.F) {
			// A deep, non-final state that maps to itself: the text repeats one byte (a zero page
			// under a signature that starts with zeros) and nothing changes until it stops doing
			// so.  Skip the repeats 16 per load level; past kRunCap bytes the run is the chain
			// pipeline's job (it does not care how long a run is).
			const uint64_t splat = 0x0101010101010101ull * byte;
			uint32_t skipped = 0;
			while (p + 17 <= a.n && skipped < kRunCap) {
				const acm_dev::Unaligned16 *t = (const acm_dev::Unaligned16 *)(a.text + p + 1);
				const uint64_t x0 = t->lo ^ splat, x1 = t->hi ^ splat;
				const uint32_t same = x0 ? (uint32_t)(__ffsll((long long)x0) - 1) >> 3
							 : 8u + (x1 ? (uint32_t)(__ffsll((long long)x1) - 1) >> 3 : 8u);
				p += same;
				skipped += same;
				if (same < 16)
					break;
			}
			if (skipped >= kRunCap) {
				ok = false;
				break;
			}
		}
		state = d.s;
		const bool deep = d.depth >= 3;
		if (deep)
			end = (int32_t)p;
		if (state >= a.F && p >= a.drop_before) {
			const uint32_t slot = atomicAdd(&hits.count, 1u);
			if (slot >= kMaxHits) {
				ok = false;
				break;
			}
			hits.pos[slot] = p;
			hits.state[slot] = state;
			hits.owner[slot] = owner;
		}
		if (deep && d.run != 0 && state < a.F) {
			// k more deep, non-final positions along a unary trie path, 16 per load level
			const uint32_t k = fast_forward(a, d, p + 1, min(a.n - p - 1, kForwardCap));
			p += k;
			state = d.s;
			end = (int32_t)p;
		}
		p++;
		if (!deep && p - p0 >= min_steps) {
			if (p >= a.n) {
				reached_end = true;
				break;
			}
			// candidate bit of position p: in LDS for the wave's own positions
			const uint32_t wi = (p >> 6) - word0;
			const uint64_t mbits = wi < kWaveWords ? lmask[wi] : *(const uint64_t *)(a.mask + (size_t)(p >> 6) * 4);
			if (!((mbits >> (p & 63)) & 1ull))
				break;
		}
		if (++iters >= kIterCap) {
			ok = false;
			break;
		}
	}
	end_out = end;
	if (reached_end)   // the earliest such walker is exact at the last byte: it carries last_state
		atomicMin(a.keeper, ((unsigned long long)p0 << 32) | state);
	return ok;
}

__global__ __launch_bounds__(kWalkBlock, 8) void k_sparse_walk(SparseArgs a)
{
	__shared__ HitList hits;
	__shared__ uint64_t lmask[kWalkWaves][kWaveWords];       // candidate bits of the wave's words
	__shared__ int32_t wave_max[kWalkWaves];
	__shared__ uint32_t s_gave_up, s_survivors;
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	static_assert(kMaxHits == kWalkBlock, "one thread clears one slot");
	hits.owner[threadIdx.x] = 0xFFFFFFFFu;   // a reserved but not yet written slot belongs to nobody
	if (threadIdx.x == 0) {
		hits.count = 0;
		s_gave_up = 0;
		s_survivors = 0;
	}
	__syncthreads();
	// a lane owns kLaneWords consecutive words, a wave kWaveWords
	const uint32_t word0 = (blockIdx.x * kWalkWaves + wv) * kWaveWords;   // the wave's first word
	const uint32_t w = word0 + lane * kLaneWords;                          // the lane's first word

	// the walkers of the wave, in position order
	uint64_t starts[kLaneWords];
	uint32_t cnt = 0;
	{
		uint64_t prev = (w && w < a.nwords) ? (uint64_t)(a.mask[(size_t)w * 4 - 1] >> 15) : 0ull;
#pragma unroll
		for (uint32_t j = 0; j < kLaneWords; j++) {
			const uint64_t m = w + j < a.nwords ? *(const uint64_t *)(a.mask + (size_t)(w + j) * 4) : 0ull;
			starts[j] = m & ~((m << 1) | prev);   // first bit of every run of candidate bits
			prev = m >> 63;
			lmask[wv][lane * kLaneWords + j] = m;
		}
		if (w == 0)
			starts[0] |= 1ull;   // the walker that carries init_state (bits 0, 1 are never candidates)
#pragma unroll
		for (uint32_t j = 0; j < kLaneWords; j++)
			cnt += (uint32_t)__popcll(starts[j]);
	}
	uint32_t inc = cnt;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(inc, o, 64);
		if (lane >= (uint32_t)o)
			inc += t;
	}
	const uint32_t total = __shfl(inc, 63, 64), base = inc - cnt;
	bool ok = total <= kMaxWalkWave;   // wave-uniform
	int32_t carry = -1;                // largest deep extent of the wave's walkers so far
	uint32_t settled = 0;              // hits below this index have their 'covered' value
	const uint32_t rounds = ok ? (total + 63) / 64 : 0;
	for (uint32_t r = 0; r < rounds; r++) {   // walker k runs on lane k % 64, whichever word it starts in
		if (__any(*(volatile uint32_t *)a.flags != 0))
			break;   // some walker of the batch hit a cap: the chain pipeline will redo it all
		const uint32_t k = r * 64 + lane;
		const bool valid = k < total;
		uint32_t owner_lane = 0;               // the last lane whose first walker index is <= k
#pragma unroll
		for (int step = 32; step > 0; step >>= 1) {
			const uint32_t cand = owner_lane + step;
			const uint32_t b = __shfl(base, cand & 63, 64);
			if (cand < 64 && b <= k)
				owner_lane = cand;
		}
		uint32_t skip = k - __shfl(base, owner_lane, 64);   // walkers of the owner lane in front of k
		uint64_t word_starts = 0;                            // start bits of the word walker k is in
		uint32_t word_index = 0;
		bool found = false;
#pragma unroll
		for (uint32_t j = 0; j < kLaneWords; j++) {
			const uint32_t lo = __shfl((uint32_t)starts[j], owner_lane, 64);
			const uint32_t hi = __shfl((uint32_t)(starts[j] >> 32), owner_lane, 64);
			const uint64_t sj = ((uint64_t)hi << 32) | lo;
			const uint32_t c = (uint32_t)__popcll(sj);
			if (!found) {
				if (skip < c) {
					word_starts = sj;
					word_index = j;
					found = true;
				} else {
					skip -= c;
				}
			}
		}
		int32_t end = -1;
		if (valid) {
			for (uint32_t i = 0; i < skip; i++)
				word_starts &= word_starts - 1;
			const uint32_t p0 = ((word0 + owner_lane * kLaneWords + word_index) << 6) +
					    (uint32_t)__ffsll((long long)word_starts) - 1;
			ok &= walker(a, hits, lmask[wv], word0, (wv << 16) | (r << 8) | lane, p0, end);
			if (!ok)
				a.flags[0] = 1;   // at once: every wave of the batch stops at its next round
		}
		int32_t incm = end;   // inclusive prefix max over the round, then exclusive + earlier rounds
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const int32_t t = __shfl_up(incm, o, 64);
			if (lane >= (uint32_t)o)
				incm = max(incm, t);
		}
		int32_t excl = __shfl_up(incm, 1, 64);
		if (lane == 0)
			excl = -1;
		// the hits this round's walkers staged learn what the walkers in front of theirs cover
		// (the value sits in the owner's lane: a shuffle, in a loop every lane runs)
		const int32_t before = max(carry, excl);
		const uint32_t staged = min(*(volatile uint32_t *)&hits.count, kMaxHits);
		for (uint32_t i = settled; i < staged; i++) {
			const uint32_t o = *(volatile uint32_t *)&hits.owner[i];
			const int32_t v = __shfl(before, o & 63u, 64);
			if (lane == 0 && (o >> 8) == ((wv << 8) | r))
				hits.covered[i] = v;
		}
		settled = staged;
		carry = max(carry, __shfl(incm, 63, 64));
	}
	if (!ok)
		s_gave_up = 1;
	if (lane == 0)
		wave_max[wv] = carry;
	__syncthreads();

	// drop the hits an earlier walker of this workgroup covers
	const uint32_t nh = min(hits.count, kMaxHits);
	for (uint32_t i = threadIdx.x; i < nh; i += kWalkBlock) {
		const uint32_t o = hits.owner[i], ow = o >> 16;
		int32_t m = hits.covered[i];
		for (uint32_t v = 0; v < ow; v++)
			m = max(m, wave_max[v]);
		if ((int32_t)hits.pos[i] <= m)
			hits.owner[i] = o | 0x80000000u;
	}
	__syncthreads();
	// the others have distinct positions: rank = number of survivors in front
	uint32_t survivors = 0;
	for (uint32_t i = threadIdx.x; i < nh; i += kWalkBlock) {
		if (hits.owner[i] & 0x80000000u)
			continue;
		const uint32_t pos = hits.pos[i];
		uint32_t rank = 0;
		for (uint32_t j = 0; j < nh; j++)
			rank += (!(hits.owner[j] & 0x80000000u) && hits.pos[j] < pos) ? 1u : 0u;
		a.hit_list[(size_t)blockIdx.x * kMaxHits + rank] = make_uint2(pos, (uint32_t)a.out[hits.state[i]]);
		survivors++;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		survivors += __shfl_xor(survivors, o, 64);
	if (lane == 0 && survivors)
		atomicAdd(&s_survivors, survivors);
	__syncthreads();
	if (threadIdx.x == 0) {
		int32_t bm = -1;
		for (int i = 0; i < kWalkWaves; i++)
			bm = max(bm, wave_max[i]);
		a.block_extent[blockIdx.x] = (uint32_t)(bm + 1);
		a.block_hits[blockIdx.x] = s_survivors;
		if (s_gave_up)
			a.flags[0] = 1;
	}
}

constexpr int kEmitBlock = 1024;

// exclusive scan (max or sum) of one value per thread over the workgroup; *total = the full reduction
template <bool IS_MAX>
__device__ __forceinline__ uint32_t block_exclusive(uint32_t x, uint32_t *lds, uint32_t *total)
{
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	uint32_t incl = x;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(incl, o, 64);
		if (lane >= (uint32_t)o)
			incl = IS_MAX ? max(incl, t) : incl + t;
	}
	uint32_t excl = __shfl_up(incl, 1, 64);
	if (lane == 0)
		excl = 0;
	if (lane == 63)
		lds[wv] = incl;
	__syncthreads();
	uint32_t before = 0, all = 0;
	for (uint32_t i = 0; i < kEmitBlock / 64; i++) {
		const uint32_t v = lds[i];
		if (i < wv)
			before = IS_MAX ? max(before, v) : before + v;
		all = IS_MAX ? max(all, v) : all + v;
	}
	__syncthreads();
	*total = all;
	return IS_MAX ? max(before, excl) : before + excl;
}

__global__ __launch_bounds__(kEmitBlock) void k_sparse_emit(SparseArgs a)
{
	__shared__ uint32_t lds[kEmitBlock / 64];
	__shared__ uint32_t first_cell[kEmitBlock + 1];   // output cell of each workgroup's first kept hit (this pass)
	__shared__ uint32_t first_kept[kEmitBlock];       // index of that hit in the workgroup's list
	if (a.flags[0]) {   // a cap was hit: the chain pipeline behind this kernel produces the planes
		if (threadIdx.x == 0 && a.giveups)
			__hip_atomic_fetch_add(a.giveups, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		return;
	}
	uint32_t extent_before = 0, cells_before = 0;   // over the workgroups of earlier passes
	for (uint32_t first = 0; first < a.nblocks; first += kEmitBlock) {
		const uint32_t b = first + threadIdx.x;
		const bool live = b < a.nblocks;
		const uint32_t extent = live ? a.block_extent[b] : 0u;
		const uint32_t staged = live ? a.block_hits[b] : 0u;
		const uint2 *list = a.hit_list + (size_t)b * kMaxHits;
		// the list is sorted: what an earlier workgroup's walker covers is a prefix of it.  Its first
		// entries are fetched before 'covered' is known (one load level instead of a dependent loop)
		uint2 head[4];
#pragma unroll
		for (uint32_t i = 0; i < 4; i++)
			head[i] = i < staged ? list[i] : make_uint2(0xFFFFFFFFu, 0u);
		uint32_t pass_extent, pass_cells;
		const uint32_t covered = max(extent_before, block_exclusive<true>(extent, lds, &pass_extent));
		uint32_t dropped = 0;
#pragma unroll
		for (uint32_t i = 0; i < 4; i++)
			dropped += (i < staged && head[i].x < covered) ? 1u : 0u;
		if (dropped == 4)
			while (dropped < staged && list[dropped].x < covered)
				dropped++;
		const uint32_t kept = staged - dropped;
		const uint32_t cell = block_exclusive<false>(kept, lds, &pass_cells);
		first_cell[threadIdx.x] = cell;
		first_kept[threadIdx.x] = dropped;
		if (threadIdx.x == 0)
			first_cell[kEmitBlock] = pass_cells;
		__syncthreads();
		// one thread per output cell: find the workgroup it belongs to, copy the record
		for (uint32_t c = threadIdx.x; c < pass_cells; c += kEmitBlock) {
			uint32_t lo = 0;   // last workgroup whose first cell is <= c (those without hits share a cell with the next)
#pragma unroll
			for (uint32_t step = kEmitBlock / 2; step > 0; step >>= 1)
				if (first_cell[lo + step] <= c)
					lo += step;
			const uint2 rec = a.hit_list[(size_t)(first + lo) * kMaxHits + first_kept[lo] + (c - first_cell[lo])];
			const uint32_t d = cells_before + c;
			if (d + 2 < a.plane_capacity) {
				a.pat_plane[1 + d] = (int32_t)rec.y;
				a.off_plane[1 + d] = (int32_t)rec.x + a.off_shift;
			}
		}
		__syncthreads();
		extent_before = max(extent_before, pass_extent);
		cells_before += pass_cells;
	}
	if (threadIdx.x == 0) {   // header and trailer cells
		const uint32_t total_hits = cells_before;
		uint32_t last;
		const unsigned long long k = *a.keeper;
		if (k != ~0ull)
			last = (uint32_t)k;   // a walker was still going at the last byte
		else                      // depth <= 2 at the end: the state is a function of the last two bytes
			last = a.t2g[(uint32_t)a.text[a.n - 2] | ((uint32_t)a.text[a.n - 1] << 8)];
		const int32_t last_ref = (int32_t)a.dev2ref[last];
		uint32_t tail = total_hits + 1;
		if (tail > a.plane_capacity - 1)
			tail = a.plane_capacity - 1;
		a.pat_plane[0] = (int32_t)total_hits;
		a.off_plane[0] = (int32_t)total_hits;
		a.pat_plane[tail] = last_ref;
		a.off_plane[tail] = last_ref;
	}
}

size_t align_up(size_t v, size_t al) { return (v + al - 1) / al * al; }

}  // namespace

namespace acm {

size_t sparse_workspace_bytes(size_t max_text)
{
	const size_t blocks = (max_text / 64 + 2) / kBlockWords + 2;
	size_t o = 0;
	o += align_up((max_text / 16 + 16) * 2, 256);   // mask
	o += align_up(blocks * 4, 256) * 2;              // per-workgroup extent, hit count
	o += align_up(blocks * kMaxHits * 8, 256);       // per-workgroup hit lists
	o += 256;                                        // flags + keeper
	return o;
}

int sparse_prepare(const acm_dfa *)
{
	ACM_HIP_TRY(hipFuncSetAttribute((const void *)k_sparse_filter, hipFuncAttributeMaxDynamicSharedMemorySize,
	    (int)(acm::kBloomMaxWords * 4)));
	return ACM_OK;
}

int sparse_scan_enqueue(const acm_dfa *d, const acm_scan_batch *b, uint32_t init_dev, void *sparse_ws,
    uint32_t *path_marker, hipStream_t s, const uint32_t **gate, hipEvent_t after_filter, hipEvent_t after_walk)
{
	const size_t n = b->n;
	SparseArgs a;
	memset(&a, 0, sizeof(a));
	a.deep = d->d_deep;
	a.out = b->report == ACM_REPORT_STATE ? (const int32_t *)d->d_dev2ref : d->d_out;
	a.dev2ref = d->d_dev2ref;
	a.in_byte = d->d_in_byte;
	a.bloom = d->d_bloom;
	a.bloom_log_words = d->bloom_log_words;
	a.bloom_words = 1u << d->bloom_log_words;
	a.t2g = d->d_t2g;
	a.text16 = (const uint4 *)b->d_text;
	a.text = (const uint8_t *)b->d_text;
	a.n = (uint32_t)n;
	a.n_pad = (uint32_t)((n + 15) & ~(size_t)15);
	a.F = d->first_final;
	a.init_state = init_dev;
	a.drop_before = (uint32_t)b->halo;
	a.off_shift = (int32_t)b->offset_shift;
	a.nwords = (uint32_t)((n + 63) / 64);
	a.nblocks = (a.nwords + kBlockWords - 1) / kBlockWords;
	char *ws = (char *)sparse_ws;
	size_t o = 0;
	auto take = [&](size_t bytes) {
		char *p = ws + o;
		o += align_up(bytes, 256);
		return p;
	};
	const size_t blocks = (n / 64 + 2) / kBlockWords + 2;
	a.mask = (uint16_t *)take((n / 16 + 16) * 2);
	a.block_extent = (uint32_t *)take(blocks * 4);
	a.block_hits = (uint32_t *)take(blocks * 4);
	a.hit_list = (uint2 *)take(blocks * kMaxHits * 8);
	a.flags = (uint32_t *)take(256);
	a.keeper = (unsigned long long *)(a.flags + 8);
	a.pat_plane = b->d_pat_plane;
	a.off_plane = b->d_off_plane;
	a.plane_capacity = (uint32_t)(b->plane_capacity > 0xFFFFFFFFul ? 0xFFFFFFFFul : b->plane_capacity);
	a.path_marker = path_marker;
	a.giveups = d->d_giveups;
	*gate = a.flags;

	// a small filter leaves room for two workgroups per CU (and for the other kernels' LDS)
	const size_t lds = (size_t)a.bloom_words * 4;
	const uint32_t per_cu = lds <= 64 * 1024 ? 2 : 1;
	const uint32_t wave_iters = (a.n_pad / 16 + 8 + 63) / 64;
	uint32_t fblocks = (wave_iters + kFilterBlock / 64 - 1) / (kFilterBlock / 64);
	if (fblocks > (uint32_t)d->num_cus * per_cu)
		fblocks = (uint32_t)d->num_cus * per_cu;
	hipLaunchKernelGGL(k_sparse_filter, dim3(fblocks), dim3(kFilterBlock), lds, s, a);
	if (after_filter)
		ACM_HIP_TRY(hipEventRecord(after_filter, s));
	hipLaunchKernelGGL(k_sparse_walk, dim3(a.nblocks), dim3(kWalkBlock), 0, s, a);
	if (after_walk)
		ACM_HIP_TRY(hipEventRecord(after_walk, s));
	hipLaunchKernelGGL(k_sparse_emit, dim3(1), dim3(kEmitBlock), 0, s, a);
	ACM_HIP_TRY(hipGetLastError());
	return ACM_OK;
}

}  // namespace acm
