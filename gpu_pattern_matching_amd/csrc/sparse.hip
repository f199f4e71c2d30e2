// The sparse scan pipeline: trigram filter -> candidate walks -> prefix max ->
// count -> ordered scatter.  gfx950 only.  Same contract and output as the
// chain pipeline of scan.hip (it is the second implementation of
// acm_scan_*_async, chosen at upload for pattern sets whose shortest pattern
// has at least 3 bytes); scan.hip stays the general path and the in-launch
// fallback.
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
//   SW1 k_sparse_walk     one walker per run of consecutive candidate bits:
//                         start in T2[bytes a-2, a-1] at position a and walk
//                         the DFA exactly (cold/meta planes, fast-forward)
//                         while the state is deep or the next position is a
//                         candidate.  Hits and the walker's deep extent
//                         [a, end] are staged per 64-position word.  A walker
//                         that started while an earlier deep run was still
//                         alive walks suffix-states of the true ones until
//                         that run ends, and exact states after it.
//   max-scan + SW2/SS     walker j keeps its hits at positions > M_j, the
//                         largest 'end' of the walkers before it (exclusive
//                         prefix max in position order): exactly the part of
//                         its walk no earlier walker covers.  Counts are
//                         scanned and the kept hits land in position order.
//
// The state carried into the buffer (init_state) is handled by a walker at
// position 0 that starts from it.  A walker is capped (kIterCap table steps, 64
// staged records per word); texts that exceed the caps -- very long deep
// runs, e.g. a page of the byte a signature starts with three times -- raise
// a device flag and the chain pipeline, enqueued right behind and otherwise
// a row of early-exit launches, produces the planes instead.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>
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
constexpr int kWordBlock = 256;            // words (of 64 text positions) per SW block
constexpr uint32_t kRecPerWord = 64;       // staged records per word (aliases the chain staging area)
constexpr uint32_t kIterCap = 256;         // table steps one walker may take ...
constexpr uint32_t kForwardCap = 4096;     // ... each followed by this many fast-forwarded bytes at most
constexpr uint32_t kHdr = 0x80000000u;

struct SparseArgs {
	const uint32_t *cold, *meta;
	const int32_t *out;
	const uint32_t *dev2ref;
	const uint8_t *in_byte;
	const uint32_t *bloom;    // [kBloomWords] blocked Bloom filter of the depth-3 trigrams
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
	uint16_t *mask;           // [n_pad / 16 + 8] candidate bits
	int32_t *maxend;          // [nwords] largest deep extent of the word's walkers, -1 if none
	uint32_t *nrec;           // [nwords] staged records (headers + hits)
	uint32_t *cnt;            // [nwords] kept hits
	int32_t *bmax;            // [nblocks] per-block max of maxend, then exclusive prefix max
	int32_t *boff;            // [nblocks] per-block kept hits, then exclusive prefix sum
	uint2 *stage;             // [nwords][kRecPerWord]
	uint32_t *flags;          // [0] overflow -> fall back, [1] total
	unsigned long long *keeper;   // start position << 32 | state of the first walker that reached the end
	// output
	int32_t *pat_plane, *off_plane;
	uint32_t plane_capacity;
};

// ------------------------------------------------------------------ SF ---

__device__ __forceinline__ uint32_t bloom_test(const uint32_t *bloom, uint32_t tri)
{
	const uint32_t p1 = __umul24(tri, acm::kBloomMul1), p2 = __umul24(tri, acm::kBloomMul2);
	const uint32_t w = bloom[p1 >> (32 - acm::kBloomLogWords)];
	return (w >> (p2 >> 27)) & (w >> ((p2 >> 22) & 31)) & 1u;
}

template <int K>
__device__ __forceinline__ uint32_t probe(const uint32_t *bloom, const uint32_t (&x)[5])
{
	// bytes (K-2, K-1, K) of the 16-byte group; x[0] is the dword in front of it
	constexpr int lo = (K + 2) / 4, sh = (K + 2) % 4;
	const uint32_t v = sh == 0 ? x[lo] : __builtin_amdgcn_alignbyte(x[lo + 1 > 4 ? 4 : lo + 1], x[lo], sh);
	return bloom_test(bloom, v & 0xFFFFFFu) << K;
}

// persistent: one workgroup per CU keeps the filter in LDS; a wave-iteration
// reads 1 KiB of text with one coalesced 16 B/lane load
__global__ __launch_bounds__(kFilterBlock) void k_sparse_filter(SparseArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t bloom[];
	{
		const uint4 *src = (const uint4 *)a.bloom;
		uint4 *dst = (uint4 *)bloom;
		constexpr uint32_t n16 = acm::kBloomWords / 4;
		const uint32_t rot = (blockIdx.x * 1021u) % n16;
		for (uint32_t i = threadIdx.x; i < n16; i += kFilterBlock) {
			uint32_t j = i + rot;
			j = j >= n16 ? j - n16 : j;
			dst[j] = src[j];
		}
	}
	__syncthreads();
	const uint32_t n16 = a.n_pad >> 4;
	const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * (kFilterBlock / 64) + (threadIdx.x >> 6);
	const uint32_t nw = gridDim.x * (kFilterBlock / 64);
	const uint32_t *text32 = (const uint32_t *)a.text16;
	for (uint32_t t = wave; t * 64 < n16 + 8; t += nw) {
		const uint32_t i16 = t * 64 + lane;
		if (i16 >= n16) {
			if (i16 < n16 + 8)
				a.mask[i16] = 0;   // padding the walkers may read
			continue;
		}
		const uint4 w = a.text16[i16];
		const uint32_t prev = i16 ? text32[i16 * 4 - 1] : 0u;
		const uint32_t x[5] = { prev, w.x, w.y, w.z, w.w };
		uint32_t m = 0;
		m |= probe<0>(bloom, x); m |= probe<1>(bloom, x); m |= probe<2>(bloom, x); m |= probe<3>(bloom, x);
		m |= probe<4>(bloom, x); m |= probe<5>(bloom, x); m |= probe<6>(bloom, x); m |= probe<7>(bloom, x);
		m |= probe<8>(bloom, x); m |= probe<9>(bloom, x); m |= probe<10>(bloom, x); m |= probe<11>(bloom, x);
		m |= probe<12>(bloom, x); m |= probe<13>(bloom, x); m |= probe<14>(bloom, x); m |= probe<15>(bloom, x);
		if (i16 == 0)
			m &= ~3u;                        // no full trigram yet: the position-0 walker covers these
		const uint32_t pos0 = i16 << 4;
		if (pos0 + 16 > a.n)                 // bytes past the end of the text
			m &= (1u << (a.n - pos0)) - 1u;
		a.mask[i16] = (uint16_t)m;
	}
}

// ----------------------------------------------------------------- SW1 ---

struct WordOut {
	uint2 *region;
	uint32_t nrec;
	int32_t maxend;
	bool overflow;
};

__device__ __forceinline__ bool mask_bit(const SparseArgs &a, uint32_t p)
{
	return (a.mask[p >> 4] >> (p & 15)) & 1u;
}

// walk from 'state' (the state before position p0) over positions p0, p0+1, ...
__device__ __forceinline__ void walker(const SparseArgs &a, WordOut &o, uint32_t start_tag, uint32_t state,
    uint32_t p0, uint32_t min_steps)
{
	if (o.nrec >= kRecPerWord) {
		o.overflow = true;
		return;
	}
	const uint32_t hdr = o.nrec++;
	int32_t end = -1;
	uint32_t p = p0, iters = 0;
	const uint32_t tbase = p0 & ~15u;
	ChainText txt(a, tbase);
	bool reached_end = false;
	for (;;) {
		if (p >= a.n) {
			reached_end = true;
			break;
		}
		Deep d = deep_step(a, state, txt.at(p - tbase + 1));
		state = d.s;
		const bool deep = d.depth >= 3;
		if (deep)
			end = (int32_t)p;
		if (state >= a.F && p >= a.drop_before) {
			if (o.nrec >= kRecPerWord) {
				o.overflow = true;
				break;
			}
			o.region[o.nrec++] = make_uint2(p, state);
		}
		if (deep && d.run != 0 && state < a.F) {
			// k more deep, non-final positions along a unary trie path, 16 per load level
			const uint32_t k = fast_forward(a, d, p + 1, min(a.n - p - 1, kForwardCap));
			p += k;
			state = d.s;
			end = (int32_t)p;
		}
		p++;
		if (!deep && p - p0 >= min_steps && (p >= a.n || !mask_bit(a, p))) {
			reached_end = p >= a.n;
			break;
		}
		if (++iters >= kIterCap) {
			o.overflow = true;
			break;
		}
	}
	o.region[hdr] = make_uint2(start_tag, kHdr | (uint32_t)(end + 1));
	if (end > o.maxend)
		o.maxend = end;
	if (reached_end)   // the earliest such walker is exact at the last byte: it carries last_state
		atomicMin(a.keeper, ((unsigned long long)start_tag << 32) | state);
}

__global__ __launch_bounds__(kWordBlock) void k_sparse_walk(SparseArgs a)
{
	__shared__ int32_t wmax[kWordBlock / 64];
	const uint32_t w = blockIdx.x * kWordBlock + threadIdx.x;
	int32_t mymax = -1;
	if (w < a.nwords) {
		const uint64_t m = *(const uint64_t *)(a.mask + (size_t)w * 4);
		const uint64_t prev = w ? (uint64_t)(a.mask[(size_t)w * 4 - 1] >> 15) : 0ull;
		uint64_t starts = m & ~((m << 1) | prev);
		WordOut o;
		o.region = a.stage + (size_t)w * kRecPerWord;
		o.nrec = 0;
		o.maxend = -1;
		o.overflow = false;
		if (w == 0)   // the state carried into the buffer
			walker(a, o, 0u, a.init_state, 0u, 2u);
		while (starts) {
			const uint32_t b = (uint32_t)__ffsll((long long)starts) - 1;
			starts &= starts - 1;
			const uint32_t pos = (w << 6) + b;   // >= 2: bits 0 and 1 of the text are never set
			const uint32_t st = a.t2g[(uint32_t)a.text[pos - 2] | ((uint32_t)a.text[pos - 1] << 8)];
			walker(a, o, pos, st, pos, 0u);
			if (o.overflow)
				break;
		}
		a.nrec[w] = o.nrec;
		a.maxend[w] = o.maxend;
		mymax = o.maxend;
		if (o.overflow)
			a.flags[0] = 1;
	}
#pragma unroll
	for (int s = 32; s > 0; s >>= 1)
		mymax = max(mymax, __shfl_down(mymax, s, 64));
	if ((threadIdx.x & 63) == 0)
		wmax[threadIdx.x >> 6] = mymax;
	__syncthreads();
	if (threadIdx.x == 0) {
		int32_t t = -1;
		for (int i = 0; i < kWordBlock / 64; i++)
			t = max(t, wmax[i]);
		a.bmax[blockIdx.x] = t;
	}
}

// ------------------------------------------------------- top-level scans ---

constexpr int kTopThreads = 1024;

// exclusive prefix max (IS_MAX) or sum of up to 64K block values by one workgroup, in place
template <bool IS_MAX>
__global__ __launch_bounds__(kTopThreads) void k_sparse_top(int32_t *v, uint32_t nb, uint32_t *total_out)
{
	__shared__ int32_t part[kTopThreads];
	const int tid = threadIdx.x;
	const uint32_t per = (nb + kTopThreads - 1) / kTopThreads;
	const uint32_t lo = min(nb, (uint32_t)tid * per), hi = min(nb, lo + per);
	const int32_t ident = IS_MAX ? -1 : 0;
	int32_t acc = ident;
	for (uint32_t i = lo; i < hi; i++)
		acc = IS_MAX ? max(acc, v[i]) : acc + v[i];
	part[tid] = acc;
	__syncthreads();
	// Hillis-Steele inclusive scan over the 1024 partials
	for (int o = 1; o < kTopThreads; o <<= 1) {
		int32_t t = ident;
		if (tid >= o)
			t = part[tid - o];
		__syncthreads();
		if (tid >= o)
			part[tid] = IS_MAX ? max(part[tid], t) : part[tid] + t;
		__syncthreads();
	}
	int32_t run = tid ? part[tid - 1] : ident;
	if (!IS_MAX && total_out && tid == kTopThreads - 1)
		*total_out = (uint32_t)part[tid];
	for (uint32_t i = lo; i < hi; i++) {
		const int32_t x = v[i];
		v[i] = run;
		run = IS_MAX ? max(run, x) : run + x;
	}
}

// ------------------------------------------------------------ SW2 / SS ---

// exclusive prefix of 'mine' over the block's words, on top of 'base'
template <bool IS_MAX>
__device__ __forceinline__ int32_t block_exclusive(int32_t mine, int32_t base, int32_t *lds)
{
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int32_t ident = IS_MAX ? -1 : 0;
	int32_t inc = mine;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int32_t t = __shfl_up(inc, o, 64);
		if (lane >= (uint32_t)o)
			inc = IS_MAX ? max(inc, t) : inc + t;
	}
	int32_t excl = __shfl_up(inc, 1, 64);
	if (lane == 0)
		excl = ident;
	if (lane == 63)
		lds[wv] = inc;
	__syncthreads();
	int32_t before = base;
	for (uint32_t i = 0; i < wv; i++)
		before = IS_MAX ? max(before, lds[i]) : before + lds[i];
	__syncthreads();
	return IS_MAX ? max(before, excl) : before + excl;
}

// SW2: how many of the word's staged hits survive (position > what earlier walkers cover)
__global__ __launch_bounds__(kWordBlock) void k_sparse_count(SparseArgs a)
{
	__shared__ int32_t lds[kWordBlock / 64];
	__shared__ uint32_t wsum[kWordBlock / 64];
	if (a.flags[0])
		return;   // capped: the chain pipeline behind us produces the result
	const uint32_t w = blockIdx.x * kWordBlock + threadIdx.x;
	const int32_t mine = w < a.nwords ? a.maxend[w] : -1;
	int32_t cur = block_exclusive<true>(mine, a.bmax[blockIdx.x], lds);
	uint32_t kept = 0;
	if (w < a.nwords) {
		const uint2 *region = a.stage + (size_t)w * kRecPerWord;
		const uint32_t nrec = a.nrec[w];
		int32_t thresh = cur;
		for (uint32_t r = 0; r < nrec; r++) {
			const uint2 rec = region[r];
			if (rec.y & kHdr) {
				thresh = cur;
				cur = max(cur, (int32_t)(rec.y & ~kHdr) - 1);
			} else if ((int32_t)rec.x > thresh) {
				kept++;
			}
		}
		a.cnt[w] = kept;
	}
	uint32_t s = kept;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		s += __shfl_down(s, o, 64);
	if ((threadIdx.x & 63) == 0)
		wsum[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t t = 0;
		for (int i = 0; i < kWordBlock / 64; i++)
			t += wsum[i];
		a.boff[blockIdx.x] = (int32_t)t;
	}
}

// SS: write the surviving hits in position order + header/trailer cells
__global__ __launch_bounds__(kWordBlock) void k_sparse_scatter(SparseArgs a)
{
	__shared__ int32_t lds[kWordBlock / 64];
	if (a.flags[0])
		return;
	const uint32_t w = blockIdx.x * kWordBlock + threadIdx.x;
	const int32_t mine = w < a.nwords ? a.maxend[w] : -1;
	int32_t cur = block_exclusive<true>(mine, a.bmax[blockIdx.x], lds);
	const int32_t kept = w < a.nwords ? (int32_t)a.cnt[w] : 0;
	uint32_t d = (uint32_t)block_exclusive<false>(kept, a.boff[blockIdx.x], lds);
	if (w < a.nwords && kept) {
		const uint2 *region = a.stage + (size_t)w * kRecPerWord;
		const uint32_t nrec = a.nrec[w];
		int32_t thresh = cur;
		for (uint32_t r = 0; r < nrec; r++) {
			const uint2 rec = region[r];
			if (rec.y & kHdr) {
				thresh = cur;
				cur = max(cur, (int32_t)(rec.y & ~kHdr) - 1);
			} else if ((int32_t)rec.x > thresh) {
				if (d + 2 < a.plane_capacity) {
					a.pat_plane[1 + d] = a.out[rec.y];
					a.off_plane[1 + d] = (int32_t)rec.x + a.off_shift;
				}
				d++;
			}
		}
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		const uint32_t total = a.flags[1];
		uint32_t last;
		const unsigned long long k = *a.keeper;
		if (k != ~0ull)
			last = (uint32_t)k;   // a walker was still going at the last byte
		else                      // depth <= 2 at the end: the state is a function of the last two bytes
			last = a.t2g[(uint32_t)a.text[a.n - 2] | ((uint32_t)a.text[a.n - 1] << 8)];
		const int32_t last_ref = (int32_t)a.dev2ref[last];
		uint32_t tail = total + 1;
		if (tail > a.plane_capacity - 1)
			tail = a.plane_capacity - 1;
		a.pat_plane[0] = (int32_t)total;
		a.off_plane[0] = (int32_t)total;
		a.pat_plane[tail] = last_ref;
		a.off_plane[tail] = last_ref;
	}
}

size_t align_up(size_t v, size_t al) { return (v + al - 1) / al * al; }

}  // namespace

namespace acm {

size_t sparse_workspace_bytes(size_t max_text)
{
	const size_t words = max_text / 64 + 2, blocks = words / kWordBlock + 2;
	size_t o = 0;
	o += align_up((max_text / 16 + 16) * 2, 256);   // mask
	o += align_up(words * 4, 256) * 3;               // maxend, nrec, cnt
	o += align_up(blocks * 4, 256) * 2;              // bmax, boff
	o += 256;                                        // flags + keeper
	return o;
}

// enqueue the sparse pipeline; *gate receives the device address of the flag
// the chain pipeline must test (non-zero -> run)
int sparse_scan_enqueue(const acm_dfa *d, const acm_scan_batch *b, uint32_t init_dev, void *sparse_ws,
    void *stage_area, hipStream_t s, const uint32_t **gate)
{
	const size_t n = b->n;
	SparseArgs a;
	memset(&a, 0, sizeof(a));
	a.cold = d->d_cold;
	a.meta = d->d_meta;
	a.out = d->d_out;
	a.dev2ref = d->d_dev2ref;
	a.in_byte = d->d_in_byte;
	a.bloom = d->d_bloom;
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
	a.nblocks = (a.nwords + kWordBlock - 1) / kWordBlock;
	char *ws = (char *)sparse_ws;
	size_t o = 0;
	auto take = [&](size_t bytes) {
		char *p = ws + o;
		o += align_up(bytes, 256);
		return p;
	};
	const size_t words = n / 64 + 2, blocks = words / kWordBlock + 2;
	a.mask = (uint16_t *)take((n / 16 + 16) * 2);
	a.maxend = (int32_t *)take(words * 4);
	a.nrec = (uint32_t *)take(words * 4);
	a.cnt = (uint32_t *)take(words * 4);
	a.bmax = (int32_t *)take(blocks * 4);
	a.boff = (int32_t *)take(blocks * 4);
	a.flags = (uint32_t *)take(256);
	a.keeper = (unsigned long long *)(a.flags + 8);
	a.stage = (uint2 *)stage_area;
	a.pat_plane = b->d_pat_plane;
	a.off_plane = b->d_off_plane;
	a.plane_capacity = (uint32_t)(b->plane_capacity > 0xFFFFFFFFul ? 0xFFFFFFFFul : b->plane_capacity);
	*gate = a.flags;

	ACM_HIP_TRY(hipMemsetAsync(a.flags, 0, 32, s));
	ACM_HIP_TRY(hipMemsetAsync(a.keeper, 0xFF, 8, s));
	const size_t lds = (size_t)kBloomWords * 4;
	ACM_HIP_TRY(hipFuncSetAttribute((const void *)k_sparse_filter, hipFuncAttributeMaxDynamicSharedMemorySize,
	    (int)lds));
	const uint32_t wave_iters = (a.n_pad / 16 + 8 + 63) / 64;
	uint32_t fblocks = (wave_iters + kFilterBlock / 64 - 1) / (kFilterBlock / 64);
	if (fblocks > (uint32_t)d->num_cus)
		fblocks = (uint32_t)d->num_cus;
	hipLaunchKernelGGL(k_sparse_filter, dim3(fblocks), dim3(kFilterBlock), lds, s, a);
	hipLaunchKernelGGL(k_sparse_walk, dim3(a.nblocks), dim3(kWordBlock), 0, s, a);
	hipLaunchKernelGGL(k_sparse_top<true>, dim3(1), dim3(kTopThreads), 0, s, a.bmax, a.nblocks, (uint32_t *)nullptr);
	hipLaunchKernelGGL(k_sparse_count, dim3(a.nblocks), dim3(kWordBlock), 0, s, a);
	hipLaunchKernelGGL(k_sparse_top<false>, dim3(1), dim3(kTopThreads), 0, s, a.boff, a.nblocks, a.flags + 1);
	hipLaunchKernelGGL(k_sparse_scatter, dim3(a.nblocks), dim3(kWordBlock), 0, s, a);
	ACM_HIP_TRY(hipGetLastError());
	return ACM_OK;
}

}  // namespace acm
