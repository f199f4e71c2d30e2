// The sparse ("sieve") scan pipeline: strided 3-gram filter -> exact prefix check ->
// trie-path followers -> ordered emit.  gfx950 only.  Same contract and output as
// the chain pipeline of scan.hip (it is the second implementation of
// acm_scan_*_async, for pattern sets whose shortest pattern has at least 3
// bytes); scan.hip stays the general path.
//
// Why it is exact.  The serial DFA state at text position e is the longest suffix
// of the text that is a trie node; call its first byte the START of e.  Let m be
// the length of the shortest pattern, W the stride (sieve_tables.h: W + 2 <= m)
// and D = min(m, 10).  Every final state has depth >= m >= D, so a record at e
// lies on a trie path text[s..e] of at least D bytes, s = start(e).
//
//  1. Sieve.  The path's first W + 2 bytes are the first W + 2 bytes of a pattern.
//     The one sample position p = 0 mod W in [s, s + W) sees the 3-gram
//     P[p-s .. p-s+2], which is in G = { P[o..o+2] : o < W }.  Testing only the
//     samples against G (a Bloom filter in LDS, no false negatives) therefore
//     misses no start: one LDS probe per W text bytes, everything else about the
//     bulk pass is the coalesced 16 B/lane read of the text -- HBM roofline.
//  2. Check.  A flagged sample asks the exact gram table for the offsets o its
//     3-gram occurs at, and the prefix table whether text[p-o .. p-o+D) is a trie
//     path: what survives is a FOLLOWER (s, depth-D node).
//  3. Follow.  A follower walks its own path only: compare the text with the
//     single outgoing edge (64 bytes per load level along unary runs) or pick the child
//     from a short edge list; the first mismatch ends it.  No fail links: the
//     suffix the DFA would fall back to starts later and has its own follower.
//     extent(s) = last position its path reaches.  It notes the final nodes it
//     enters as hits.
//  4. Shadow.  Followers alive at e are nested suffixes; the state at e is the one
//     with the smallest s.  So follower s keeps its hits at e > M(s), the largest
//     extent of the followers with a smaller start -- an exclusive prefix max in
//     start order.  (If the follower with the smallest start alive at e is not on
//     a final node, no later one is: a final suffix makes the longer node final,
//     acsmx.c:417-429.)  Samples are taken in position order and the offsets of one
//     sample in descending order, so followers come out in start order by
//     construction; the prefix max runs inside the wave that owns the tile, and
//     across tiles in the emit kernel.
//
// The state carried into the buffer (init_state) is a path that started before
// byte 0: one lane walks the real DFA from it (cold plane + depth table) while its
// start stays < 0; it precedes every follower in start order.  last_state: the
// follower with the smallest start that is alive at the last byte, else the DFA
// state after the last D-1 bytes from the root (depth < D there).
//
// Three launches, no host round trip between them:
//   k_sieve        the bulk pass.  Workgroups of 8 waves (one per CU and launch) with the Bloom filter
//                  in LDS; a wave owns a tile of the text at a time: reads it 16 B per
//                  lane, probes one 3-gram per W bytes, appends {position, 3-gram} of what
//                  the filter flags to the tile's sample list (ranks from ballots).
//                  Nothing in it waits on a dependent load: HBM roofline.
//   k_sieve_check  the exact part.  A wave takes the sample lists of eight tiles: gram and
//                  prefix lookups for 64 samples at a time (stage 1), then 64 followers
//                  at a time (stage 2), shadow inside the wave, hits appended to the
//                  wave's row in position order, a 32-byte row summary.  Latency-bound,
//                  five hundred waves of dependent loads; it runs beside the next batch's
//                  bulk pass.  Its last workgroup walks the carried state instead.
//   k_sieve_emit   exclusive prefix max / prefix sum over the row summaries, drops the
//                  shadowed head of each row, copies the records to the planes.
// Nothing is capped and nothing falls back: a tile's sample list has room for every
// sample of the tile and a row's hit list for one hit per position its followers can
// reach (geometry_for); the lists are address space, only what is written is touched.
// A text that is dense in matches, or in 3-grams of the pattern set that are no pattern
// prefixes (real binaries), is merely slow here -- the emit kernel counts such batches and
// AUTO mode moves to the chain pipeline of scan.hip (pick_sparse).
// A launch may carry up to sixteen batches of one size (SieveGroup, acm_scan_batches_async):
// the kernels' fixed costs are then paid once for all of them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdio>
#include <vector>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "acm_internal.h"
#include "deep_walk.h"
#include "device_dfa.h"
#include "sieve_tables.h"
#include "sparse.h"

namespace {

using acm_dev::agree16;

constexpr int kBlock = 512;              // threads of a k_sieve workgroup
constexpr int kWaves = kBlock / 64;
constexpr uint32_t kTilesPerChecker = 8;    // tiles whose flagged samples one wave of k_sieve_check takes when a batch is launched alone ...
constexpr uint32_t kTilesPerCheckerWide = 16;   // ... and in a launch group (see k_sieve_check)
constexpr int kCheckBlock = 64;            // threads of a k_sieve_check workgroup
constexpr uint32_t kQ2Cap = 64 + 64 * 8;   // followers a checker queues: what a round leaves + 64 samples x 8 offsets
constexpr uint32_t kMinTile = 1024;        // bytes; one 16-byte group per lane
constexpr uint32_t kMaxTiles = 4096;       // the tile size doubles until the text has at most this many
constexpr uint32_t kSummaryWords = 8;      // per tile: extent + 1, hits, first, last position, alive key, alive node
constexpr uint32_t kGaveUp = 0xFFFFFFFFu;  // hit count of a tile whose list was full (cannot happen: geometry_for)
constexpr uint32_t kMarkerFailed = 0xDEADu; // path marker of a scan that did not produce planes
constexpr uint32_t kDenseDivisor = 128;    // more than a record per this many bytes: a dense batch
constexpr uint32_t kHeavyDivisor = 512;    // more than a flagged sample per this many bytes: the next launches get helper waves (k_sieve_check)
constexpr uint32_t kBusyDivisor = 48;      // ... or a flagged sample per this many: the check kernel's time, not the bulk kernel's, is the batch's
                                           // (with helper waves the sparse pipeline keeps up with the chain pipeline's 70..95 us to about 700 k samples
                                           // per 32 MiB: 62 us at 447 k, 91 at 650 k, 106 at 983 k -- tools/real_data_probe.py)
constexpr int kEmitBlock = 1024;
#ifdef ACM_SIEVE_CHECK_STAMPS   // debugging aid (with ACM_SIEVE_STAMPS=1 at run time): clock stamps and level counts of the check
constexpr bool kCheckStamps = true;    // kernel's waves -- compiled in on request only, they cost the kernel registers
#else
constexpr bool kCheckStamps = false;
#endif
constexpr uint32_t kMaxRows = kMaxTiles / kTilesPerChecker + 1;   // static rows: the carried-state walker + a block of eight tiles each
// A block whose tiles hold more flagged samples than this (real binaries: the 3-grams of code and tables
// cluster) is cut into SUB-ROWS of at most this many samples, in position order; the block's own wave
// takes the first, the others are numbered through the whole batch in block order and dealt to the
// helper waves by that number.  No list, no counter, nobody waits: every helper works the numbering
// out for itself from the tiles' sample counts, as the emit kernel does from the rows' summaries.  To
// the emit kernel a sub-row is a row like any other.
constexpr uint32_t kSubRow = 256;
constexpr uint32_t kSubRowMin = 128;       // the smallest ACM_SIEVE_SUBROW (debugging aid) may ask for: the workspace is laid out for it
constexpr uint32_t kHelperWaves = 4096;    // waves per batch that do nothing but sub-rows, launched for sample-heavy streams only (512: 212 us for the
                                           // worst 32 MiB of a real binary x 15000 signatures, 1024: 147, 2048: 106, 4096: 102 and the second-worst piece 91 -> 76, 8192: the same; sub-rows of 128 samples: no better)

struct SieveArgs {
	// tables
	const uint32_t *bloom;
	uint32_t bloom_words, bloom_log_words;
	const uint4 *gram;
	uint32_t gram_log_buckets, gram_probes;
	const uint4 *prefix;
	uint32_t prefix_log_slots, prefix_probes;
	const uint4 *rec, *edges;
	uint32_t report_state;        // planes get the final state's reference id instead of its pattern
	const uint8_t *in_byte;
	const uint32_t *cold;         // [states][1 << ls]: the real DFA, a cell per byte class
	const uint8_t *cls;           // [256] byte -> class
	uint32_t ls;
	const uint16_t *depth;
	const int32_t *out;
	const uint32_t *dev2ref;
	uint32_t F, D;
	uint32_t nt_loads;            // bulk kernel: the text with the streaming hint
	uint32_t tail_walk;           // bytes side_walks() walks from the root at the end of the text: D - 1, or W + 4 with 6-byte filter keys
	uint32_t run_ok[8];           // bit b: D copies of byte b are a trie path (a run of b can start a pattern)
	// text
	const uint4 *text16;
	const uint8_t *text;
	uint32_t n, n_pad;
	uint32_t init_state, drop_before;
	const uint32_t *init_ptr;     // not null: the state to start in is there (handed over on the device), init_state is not
	int32_t off_shift;
	// geometry
	uint32_t tile_bytes, ntiles, cap, nrows;   // nrows: static rows = blocks of eight tiles + 1 (the walker's)
	uint32_t max_extra, sub_k;                 // sub-rows a text can have at most; list room a sub-row needs on top of its span
	uint32_t subrow;                           // samples of a sub-row (kSubRow)
	// workspace
	uint2 *shead, *lhead;   // [ntiles][kSampleHead], [nrows][kHitHead]: the first entries of the lists below
	uint2 *samples;      // [ntiles][scap] {position, 3-gram} of the samples the filter flagged, ascending
	uint32_t *scount;    // [ntiles] how many (kGaveUp: more than scap)
	uint32_t scap;
	uint32_t *summary;   // [nrows][kSummaryWords]; row 0: the carried-state walker, row 1 + w: checker wave w
	uint2 *lists;        // [nrows][cap] {position, plane value}, ascending
	uint32_t *misc;      // [0] state after the last D-1 bytes from the root
	uint32_t *path_marker, *giveups;
	unsigned long long *stamps;   // debugging aid (ACM_SIEVE_STAMPS): [wave][8] clock readings, or null
	// output
	int32_t *pat_plane, *off_plane;
	uint32_t plane_capacity;
};

// A launch takes up to kMaxGroup (16) batches of the same size (acm_scan_batches_async): the bulk
// kernel's waves go through the tiles of one batch after the other, the check and emit kernels
// get a share of workgroups per batch.  The kernels' fixed costs -- launch, filter fill, the
// chains of dependent loads of the check kernel -- are paid once per group instead of once
// per batch.
constexpr uint32_t kMaxGroup = 16;
// What travels in the argument buffer: the tables and the geometry once (a group's batches have
// one size), 64 bytes per batch; a kernel puts the SieveArgs of its batch together from both (the
// fields it does not use cost nothing).  An argument buffer of 4 x SieveArgs -- 1.2 KB -- made
// every launch measurably slower.
struct SieveBatch {
	const uint8_t *text;
	char *ws;                 // the sparse part of the batch's workspace
	int32_t *pat_plane, *off_plane;
	uint32_t *path_marker;
	uint32_t init_state, drop_before;
	int32_t off_shift;
	uint32_t plane_capacity, report_state;
	const uint32_t *init_ptr;
};
struct SieveGroup {
	SieveArgs common;         // everything but the fields of SieveBatch and the workspace pointers
	size_t o_summary, o_lists, o_samples, o_shead, o_lhead, o_scount, o_misc;   // byte offsets into ws
	SieveBatch b[kMaxGroup];
	uint32_t count;
};
__device__ __forceinline__ SieveArgs batch_view(const SieveGroup &g, uint32_t bi)
{
	SieveArgs a = g.common;
	const SieveBatch &b = g.b[bi];
	a.text = b.text;
	a.text16 = (const uint4 *)b.text;
	a.init_state = b.init_state;
	a.init_ptr = b.init_ptr;
	a.drop_before = b.drop_before;
	a.off_shift = b.off_shift;
	a.report_state = b.report_state;
	a.out = b.report_state ? (const int32_t *)g.common.dev2ref : g.common.out;
	a.summary = (uint32_t *)(b.ws + g.o_summary);
	a.lists = (uint2 *)(b.ws + g.o_lists);
	a.samples = (uint2 *)(b.ws + g.o_samples);
	a.shead = (uint2 *)(b.ws + g.o_shead);
	a.lhead = (uint2 *)(b.ws + g.o_lhead);
	a.scount = (uint32_t *)(b.ws + g.o_scount);
	a.misc = (uint32_t *)(b.ws + g.o_misc);
	a.pat_plane = b.pat_plane;
	a.off_plane = b.off_plane;
	a.plane_capacity = b.plane_capacity;
	a.path_marker = b.path_marker;
	return a;
}

// Both kinds of list are far longer than what they usually hold (a few samples per tile, a few
// hits per row), and a row per page would cost every writer and reader an address translation:
// the first kHead entries of each list live side by side in a small dense array, the rest (rare)
// in the list proper.
constexpr uint32_t kSampleHead = 32;   // a check wave reads a tile's samples one per lane
constexpr uint32_t kHitHead = 8;       // the emit kernel reads a row's first hits 64 bytes per thread: rows side by side
// where a row keeps what: its summary and the dense head of its hit list under its slot, the rest of
// the list from entry `lbase` of the lists on (a block's sub-rows share the block's region)
struct RowRef {
	uint32_t slot, lbase;
};
__device__ __forceinline__ uint2 *hit_slot(const SieveArgs &a, RowRef row, uint32_t idx)
{
	return idx < kHitHead ? a.lhead + (size_t)row.slot * kHitHead + idx : a.lists + (size_t)row.lbase + idx;
}
__device__ __forceinline__ uint2 *sample_slot(const SieveArgs &a, uint32_t tile, uint32_t idx)
{
	return idx < kSampleHead ? a.shead + (size_t)tile * kSampleHead + idx : a.samples + (size_t)tile * a.scap + idx;
}

struct __attribute__((packed)) Unaligned8 {
	uint64_t v;
};

// bytes sh .. sh+7 of the 16-byte string lo:hi
__device__ __forceinline__ uint64_t funnel(uint64_t lo, uint64_t hi, uint32_t sh)
{
	const uint32_t bits = sh * 8;
	return bits == 0 ? lo : bits >= 64 ? hi : (lo >> bits) | (hi << (64 - bits));
}

// the filter in two halves, so that the reads of many samples can be in flight before the first is
// looked at: the key's 64-bit block (one ds_read_b64), then its four bits
__device__ __forceinline__ uint64_t bloom_block(const uint32_t *bloom, uint32_t gram, uint32_t more, uint32_t block_shift)
{
	return ((const uint64_t *)bloom)[(acm::mul24(gram, acm::kSieveMulA) + acm::mul24(more, acm::kSieveMulE)) >> block_shift];
}
__device__ __forceinline__ uint32_t bloom_test(uint64_t w, uint32_t gram, uint32_t more)
{
	const uint32_t p = acm::mul24(gram, acm::kSieveMulB) + acm::mul24(more, acm::kSieveMulF);
	const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
	// (a 32-bit shift takes the low five bits of its count: sieve_bloom_bits' fields need no masks)
	return ((lo >> ((p >> 27) & 31)) & (lo >> ((p >> 22) & 31)) & (hi >> ((p >> 17) & 31)) & (hi >> ((p >> 12) & 31))) & 1u;
}

__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t v, uint32_t lane, uint32_t &total)
{
	uint32_t incl = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(incl, o, 64);
		if (lane >= (uint32_t)o)
			incl += t;
	}
	total = __shfl(incl, 63, 64);
	return incl - v;
}

__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v, uint32_t lane)
{
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(v, o, 64);
		if (lane >= (uint32_t)o)
			v = max(v, t);
	}
	return v;
}

// Hits of one lane in one round, newest first: a hit is pushed in front, the others move
// one slot back (no slot is picked by a run-time index: such a struct ends up in scratch).
struct LaneHits {
	uint32_t n;          // hits noted; more than 4: the round gives up
	uint32_t e0, v0, e1, v1, e2, v2, e3, v3;   // position, plane value
	uint32_t levels;     // debugging aid: dependent load levels of the lane's follower
};

__device__ __forceinline__ void note_hit(LaneHits &h, uint32_t x, uint32_t value, bool take)
{
	h.e3 = take ? h.e2 : h.e3;
	h.v3 = take ? h.v2 : h.v3;
	h.e2 = take ? h.e1 : h.e2;
	h.v2 = take ? h.v1 : h.v2;
	h.e1 = take ? h.e0 : h.e1;
	h.v1 = take ? h.v0 : h.v1;
	h.e0 = take ? x : h.e0;
	h.v0 = take ? value : h.v0;
	h.n += take ? 1u : 0u;
}

// bytes the windows of CH * 16 bytes at p and q agree on before the first difference (CH * 16:
// all).  All the loads are unconditional and in flight together: one latency for the whole window
// (a load inside a branch gets its own wait: a round trip each instead of one).
template <int CH>
__device__ __forceinline__ uint32_t agree(const uint8_t *p, const uint8_t *q)
{
	uint64_t lo[CH], hi[CH];
#pragma unroll
	for (int k = 0; k < CH; k++)
		acm_dev::diff_bytes16(p + k * 16, q + k * 16, lo[k], hi[k]);
	uint32_t same = CH * 16;
#pragma unroll
	for (int k = CH - 1; k >= 0; k--) {   // the earliest difference is written last
		same = hi[k] ? (uint32_t)k * 16 + 8 + (((uint32_t)__ffsll((long long)hi[k]) - 1u) >> 3) : same;
		same = lo[k] ? (uint32_t)k * 16 + (((uint32_t)__ffsll((long long)lo[k]) - 1u) >> 3) : same;
	}
	return same;
}
constexpr uint32_t kLongLevel = 64;    // bytes a follower compares per load level along a unary run (192: fewer levels, but each slower -- measured)

// One follower: from 'node' (depth D, last matched byte at x) along its trie path.
// Returns extent + 1; at_end: the path was still alive at the last byte (on 'node').
// What happens to the final nodes it enters depends on 'mode':
//   kKeep   the first pass of a round: up to four go to registers (h), all are counted (h.n)
//   kCount  a follower with more than four: count those at positions >= floor (h.n), note
//           the first and the last of them (h.e3, h.e0)
//   kWrite  ... and write those to the row's list from index 'at' on, ascending
enum { kKeep = 0, kCount = 1, kWrite = 2 };
struct Follow {
	uint32_t extent1, node;
	bool at_end;
};

__device__ __forceinline__ void final_node(const SieveArgs &a, LaneHits &h, uint32_t x, uint32_t value, bool take,
    int mode, uint32_t floor, RowRef row, uint32_t at)
{
	if (mode == kKeep) {
		note_hit(h, x, value, take);
	} else if (take && x >= floor) {
		if (mode == kWrite)
			*hit_slot(a, row, at + h.n) = make_uint2(x, value);
		if (h.n == 0)
			h.e3 = x;
		h.e0 = x;
		h.n++;
	}
}

__device__ __forceinline__ Follow follow(const SieveArgs &a, LaneHits &h, uint32_t node, uint32_t run, uint32_t x, int mode,
    uint32_t floor, RowRef row, uint32_t at)
{
	Follow f;
	if (node >= a.F)   // a pattern of exactly D bytes, or one that ends a longer suffix
		final_node(a, h, x, (uint32_t)a.out[node], x >= a.drop_before, mode, floor, row, at);
	for (;;) {
		if (x + 1 >= a.n)
			break;
		h.levels++;
		uint4 rec;
		uint32_t c;
		if (run) {   // unary, non-final path ahead: node+1, node+2, ... as long as the text agrees
			// (a signature is one unary run from its prefix to its leaf more often than not: the whole of
			// it in one level where the text has the room, 64 bytes near its end)
			const bool roomy = x + 1 + kLongLevel <= a.n_pad;
			const uint32_t step = min(run, roomy ? kLongLevel : 64u), want = min(step, a.n - 1 - x);
			// with the compare, what the node at the end of the stretch needs if the stretch is the
			// whole run: its record and the byte behind it -- one level instead of two
			c = a.text[min(x + step + 1, a.n_pad - 1)];
			rec = a.rec[node + step];
			uint32_t same;
			if (roomy) {
				same = agree<kLongLevel / 16>(a.in_byte + node + 1, a.text + x + 1);
			} else if (x + 65 <= a.n_pad) {
				same = agree<4>(a.in_byte + node + 1, a.text + x + 1);
			} else {
				same = 0;
				while (same < want && a.in_byte[node + 1 + same] == a.text[x + 1 + same])
					same++;
			}
			const uint32_t k = min(same, want);
			node += k;
			x += k;
			run -= k;
			if (k < step)
				break;   // a mismatch, or the end of the text
			if (run)
				continue;
			if (x + 1 >= a.n)
				break;
		} else {
			c = a.text[x + 1];   // both loads before anything looks at either
			rec = a.rec[node];
		}
		__asm__ volatile("" : "+v"(c));   // (keeps the byte's load from sinking into the branch that uses it)
		const uint32_t nchild = rec.y & 0x1FFu;
		uint32_t value;
		bool leaf;
		if (nchild == 0)
			break;
		if (nchild == 1) {
			if ((rec.x >> 24) != c)
				break;
			node = rec.x & 0xFFFFFFu;
			run = rec.y >> 16;
			leaf = (rec.y >> 9) & 1u;
			value = a.report_state ? rec.w : rec.z;
		} else {
			const uint4 *e = a.edges + (rec.x & 0xFFFFFFu);
			bool hit = false, past = false;
			uint4 v = make_uint4(0, 0, 0, 0);
			for (uint32_t i = 0; i < nchild && !hit && !past; i += 4) {   // sorted by byte; four edges a level
				uint4 q[4];
#pragma unroll
				for (uint32_t j = 0; j < 4; j++)
					q[j] = e[min(i + j, nchild - 1)];
#pragma unroll
				for (uint32_t j = 0; j < 4; j++) {
					const uint32_t b = q[j].x & 0xFFu;
					if (!hit && !past && i + j < nchild) {
						if (b == c) {
							hit = true;
							v = q[j];
						} else if (b > c) {
							past = true;
						}
					}
				}
			}
			if (!hit)
				break;
			node = v.x >> 8;
			run = v.y & 0xFFFFu;
			leaf = (v.y >> 16) & 1u;
			value = a.report_state ? v.w : v.z;
		}
		x++;
		final_node(a, h, x, value, node >= a.F && x >= a.drop_before, mode, floor, row, at);
		if (leaf)
			break;
	}
	f.at_end = x + 1 >= a.n;   // however the loop ended: the path reached the last byte
	f.extent1 = x + 1;
	f.node = node;
	return f;
}

// What a checker wave keeps in LDS between its two stages: the followers stage 1 found,
// in start order, until stage 2 has a full round of them.
struct FollowerQueue {
	uint32_t *start, *node, *run;   // [kQ2Cap] each
	uint32_t count;
};

// Stage 1, one flagged sample per lane (ascending positions): exact gram lookup, then the
// prefix lookups of all the offsets the gram occurs at -- every load of a step in flight
// together: two dependent steps whatever the number of offsets.  What is a trie path goes
// to the follower queue, the lanes' followers one behind the other, a lane's own in
// descending offset = ascending start order.
template <int W>
__device__ __forceinline__ void stage1_round(const SieveArgs &a, FollowerQueue &fq, uint32_t p, uint32_t gram, bool act,
    uint32_t lane)
{
	// the 24 bytes around the sample: A = [p-8, p), B = [p, p+8), C = [p+8, p+16)
	uint64_t A = 0, B = 0, C = 0;
	uint32_t mask = 0;
	if (act) {
		// four loads and nothing between them: the window at clamped addresses (the first and the last
		// bytes of the text are put right afterwards) and the gram's bucket
		const uint32_t last8 = a.n_pad - 8;
		const uint32_t pb = min(p, last8), pc = min(p + 8, last8);
		const uint64_t va = ((const Unaligned8 *)(a.text + (p >= 8 ? p - 8 : 0)))->v;
		const uint64_t vb = ((const Unaligned8 *)(a.text + pb))->v;
		const uint64_t vc = ((const Unaligned8 *)(a.text + pc))->v;
		A = p >= 8 ? va : p ? va << (8 * (8 - p)) : 0ull;
		B = p <= last8 ? vb : p >= a.n_pad ? 0ull : vb >> (8 * (p - last8));
		C = p + 8 <= last8 ? vc : p + 8 >= a.n_pad ? 0ull : vc >> (8 * (p + 8 - last8));
		const uint32_t bmask = (1u << a.gram_log_buckets) - 1u;
		uint32_t b = acm::sieve_gram_bucket(gram, a.gram_log_buckets);
		for (uint32_t probe = 0; probe < a.gram_probes; probe++, b = (b + 1) & bmask) {
			const uint4 e = a.gram[b];
			if (e.x && (e.x & 0xFFFFFFu) == gram) mask = e.x >> 24;
			else if (e.y && (e.y & 0xFFFFFFu) == gram) mask = e.y >> 24;
			else if (e.z && (e.z & 0xFFFFFFu) == gram) mask = e.z >> 24;
			else if (e.w && (e.w & 0xFFFFFFu) == gram) mask = e.w >> 24;
			if (mask || !e.x || !e.y || !e.z || !e.w)
				break;   // found, or a bucket with room: the gram is not in the table
		}
	}
	if (!__ballot(mask != 0))
		return;
	// bytes of the key beyond D are zero in the table
	const uint64_t klo = a.D >= 8 ? ~0ull : (1ull << (8 * a.D)) - 1ull;
	const uint32_t khi = a.D >= 10 ? 0xFFFFu : a.D <= 8 ? 0u : 0xFFu;
	const uint32_t smask = (1u << a.prefix_log_slots) - 1u;
	// (keys are recomputed where they are needed instead of kept: registers)
	auto key_of = [&](uint32_t o, uint32_t &k0, uint32_t &k1, uint32_t &k2) {
		const uint64_t lo = funnel(A, B, 8 - o) & klo;
		k2 = (uint32_t)funnel(B, C, 8 - o) & khi;
		k0 = (uint32_t)lo;
		k1 = (uint32_t)(lo >> 32);
	};
	uint4 ent[W];
#pragma unroll
	for (int o = W - 1; o >= 0; o--) {
		ent[o] = make_uint4(0, 0, 0, 0);
		// a start before byte 0 is the carried-state walker's path; one too close to the end cannot be final
		if (((mask >> o) & 1u) && (uint32_t)o <= p && p - (uint32_t)o + a.D <= a.n) {
			uint32_t k0, k1, k2;
			key_of((uint32_t)o, k0, k1, k2);
			ent[o] = a.prefix[acm::sieve_prefix_slot(k0, k1, k2, a.prefix_log_slots)];
		}
	}
	uint32_t vmask = 0;
#pragma unroll
	for (int o = W - 1; o >= 0; o--) {
		if (!ent[o].w)
			continue;
		uint32_t k0, k1, k2;
		key_of((uint32_t)o, k0, k1, k2);
		uint32_t at = acm::sieve_prefix_slot(k0, k1, k2, a.prefix_log_slots);
		for (uint32_t probe = 1;; probe++) {
			if (ent[o].x == k0 && ent[o].y == k1 && (ent[o].z & 0xFFFFu) == k2) {
				vmask |= 1u << o;
				break;
			}
			if (probe >= a.prefix_probes)
				break;
			at = (at + 1) & smask;   // another key's slot: the next one (an eighth of them are taken)
			ent[o] = a.prefix[at];
			if (!ent[o].w)
				break;
		}
	}
	uint32_t total;
	uint32_t idx = fq.count + wave_excl_sum((uint32_t)__popc(vmask), lane, total);
	if (!total)
		return;
#pragma unroll
	for (int o = W - 1; o >= 0; o--)
		if ((vmask >> o) & 1u) {
			fq.start[idx] = p - (uint32_t)o;
			fq.node[idx] = ent[o].w;
			fq.run[idx] = ent[o].z >> 16;
			idx++;
		}
	fq.count += total;
	__builtin_amdgcn_wave_barrier();
}

// What a checker wave has found so far: one row of the emit kernel's input (wave-uniform values).
struct Row {
	uint32_t carry;      // largest extent + 1 of the followers so far
	uint32_t count;      // hits staged in the row's list
	uint32_t first, last;
	uint32_t akey, anode;
	uint32_t gave_up;
	uint32_t samples;    // flagged samples the row looked at (for AUTO mode: the emit kernel adds them up)
	uint32_t room;       // entries its list has room for
};

// word 7 of a static row: sub-rows of its block beyond its own; of a sub-row: where its list starts
__device__ __forceinline__ void write_summary(const SieveArgs &a, const Row &t, uint32_t slot, uint32_t word7)
{
	uint32_t *s = a.summary + (size_t)slot * kSummaryWords;
	*(uint4 *)s = make_uint4(t.carry, t.gave_up ? kGaveUp : t.count, t.first, t.last);
	*(uint4 *)(s + 4) = make_uint4(t.akey, t.anode, t.samples, word7);
}

// Stage 2, one follower per lane (ascending starts): follow, shadow across the lanes, append the
// surviving hits to the row's list.
__device__ __forceinline__ void stage2_round(const SieveArgs &a, Row &t, RowRef row, uint32_t s, uint32_t node,
    uint32_t run, bool act, uint32_t lane, uint32_t &dbg_levels)
{
	LaneHits h;
	uint32_t E = 0, akey = 0, anode = 0, M = 0, mine = 0, keep = 0;
	bool big = false;
	// A lane whose follower enters more than four final nodes (nested patterns) goes through the
	// loop twice: keep/count all, then count what the shadow leaves; a third pass writes that.
#pragma nounroll
	for (int mode = kKeep; mode <= kCount; mode++) {
		if (mode == kKeep || big) {
			h.n = 0;
			h.e0 = h.v0 = h.e1 = h.v1 = h.e2 = h.v2 = h.e3 = h.v3 = 0;
			h.levels = 0;
		}
		if (act && (mode == kKeep || big)) {
			const Follow f = follow(a, h, node, run, s + a.D - 1, mode, M, RowRef{ 0, 0 }, 0);
			E = f.extent1;
			if (f.at_end) {
				akey = s + 2;
				anode = f.node;
			}
		}
		if (mode == kKeep) {
			if (kCheckStamps && a.stamps) {
				uint32_t lv = h.levels;
#pragma unroll
				for (int o = 32; o > 0; o >>= 1)
					lv = max(lv, __shfl_xor(lv, o, 64));
				dbg_levels += lv;
			}
			big = h.n > 4;
			if (!__ballot(big))
				break;
			// the count pass needs the shadow bound: below
		}
		if (mode == kKeep) {
			const uint32_t incl0 = wave_incl_max(E, lane);
			uint32_t excl0 = __shfl_up(incl0, 1, 64);
			M = max(t.carry, lane == 0 ? 0u : excl0);
		} else {
			mine = big ? h.n : 0u;
		}
	}
	// shadow: a follower keeps its hits behind the extents of the followers in front of it -- of
	// earlier rounds (the row's carry) and of this round
	const uint32_t incl = wave_incl_max(E, lane);
	uint32_t excl = __shfl_up(incl, 1, 64);
	if (lane == 0)
		excl = 0;
	M = max(t.carry, excl);
	if (!big) {
		const uint32_t nh = min(h.n, 4u);   // slot 3 holds the oldest (smallest position) of four
		if (nh > 3 && h.e3 >= M) keep |= 1u;
		if (nh > 2 && h.e2 >= M) keep |= 2u;
		if (nh > 1 && h.e1 >= M) keep |= 4u;
		if (nh > 0 && h.e0 >= M) keep |= 8u;
		mine = (uint32_t)__popc(keep);
	}
	uint32_t total;
	const uint32_t base = wave_excl_sum(mine, lane, total);
	const uint32_t he[4] = { h.e3, h.e2, h.e1, h.e0 }, hv[4] = { h.v3, h.v2, h.v1, h.v0 };
	uint32_t firstpos, lastpos;
	if (big) {   // the count pass left the first and the last position in e3 and e0
		firstpos = h.e3;
		lastpos = h.e0;
	} else {
		firstpos = (keep & 1u) ? he[0] : (keep & 2u) ? he[1] : (keep & 4u) ? he[2] : he[3];
		lastpos = (keep & 8u) ? he[3] : (keep & 4u) ? he[2] : (keep & 2u) ? he[1] : he[0];
	}
	if (total) {
		if (t.count + total > a.cap) {   // (cannot happen: a list holds a hit per position the row can reach)
			t.gave_up = 1;
		} else {
			const uint32_t at = t.count + base;
			if (!big) {
				uint32_t k = 0;
#pragma unroll
				for (uint32_t i = 0; i < 4; i++)
					if (keep & (1u << i))
						*hit_slot(a, row, at + k++) = make_uint2(he[i], hv[i]);
			}
			if (__ballot(big && mine)) {   // third pass: the big lanes write what the shadow leaves them
				h.n = 0;
				if (big && mine)
					(void)follow(a, h, node, run, s + a.D - 1, kWrite, M, row, at);
			}
			const unsigned long long keepers = __ballot(mine != 0);
			const uint32_t fp = (uint32_t)__builtin_amdgcn_readlane((int)firstpos, (int)((uint32_t)__ffsll((long long)keepers) - 1u));
			const uint32_t lp = (uint32_t)__builtin_amdgcn_readlane((int)lastpos, (int)(63u - (uint32_t)__clzll((long long)keepers)));
			if (t.count == 0)
				t.first = fp;
			t.last = lp;
			t.count += total;
		}
	}
	t.carry = max(t.carry, (uint32_t)__builtin_amdgcn_readlane((int)incl, 63));
	const unsigned long long alive = __ballot(akey != 0);
	if (alive && !t.akey) {
		const int al = (int)((uint32_t)__ffsll((long long)alive) - 1u);
		t.akey = (uint32_t)__builtin_amdgcn_readlane((int)akey, al);
		t.anode = (uint32_t)__builtin_amdgcn_readlane((int)anode, al);
	}
}

// The carried state (a path that started before byte 0) and the state the last D-1
// bytes lead to from the root: two short serial walks of the real DFA, one lane.
__device__ void side_walks(const SieveArgs &a)
{
	{
		// exact wherever the state at the last byte is no deeper than the bytes walked.  A deeper state
		// has a follower -- if its sample could be looked at: with 6-byte filter keys the sample of a path
		// that starts less than W + 5 bytes before the end has its key cut off by the end of the text and
		// is never flagged, so the walk covers W + 4 bytes then (tail_walk), not just D - 1
		uint32_t st = 0;
		for (uint32_t x = a.n > a.tail_walk ? a.n - a.tail_walk : 0u; x < a.n; x++)
			st = a.cold[((size_t)st << a.ls) | a.cls[a.text[x]]];
		a.misc[0] = st;
	}
	Row t;
	t.carry = t.count = t.first = t.last = t.akey = t.anode = t.gave_up = t.samples = 0;
	t.room = a.cap;
	uint32_t state = a.init_ptr ? *a.init_ptr : a.init_state, x = 0;   // x: next byte to consume
	uint32_t run = 0;                        // known unary, non-final path ahead of 'state'
	while (state != 0) {
		if (x >= a.n) {
			t.akey = 1;
			t.anode = state;
			break;
		}
		if (run && x + 16 <= a.n_pad) {
			const uint32_t want = min(min(run, 16u), a.n - x);
			const uint32_t k = min(agree16(a.in_byte + state + 1, a.text + x), want);
			state += k;   // depth grows with every byte: the start stays where it is
			x += k;
			run -= k;
			t.carry = x;
			if (k == want)
				continue;
			run = 0;
		}
		const uint32_t c = a.text[x];
		const uint4 rec = a.rec[state];
		const uint32_t next = a.cold[((size_t)state << a.ls) | a.cls[c]];
		if ((uint32_t)a.depth[next] < x + 2)
			break;   // the state after byte x starts at or behind byte 0: a follower's business
		run = ((rec.y & 0x1FFu) == 1 && (rec.x & 0xFFFFFFu) == next) ? rec.y >> 16 : 0u;
		state = next;
		t.carry = x + 1;
		if (state >= a.F && x >= a.drop_before) {
			if (t.count >= a.cap) {
				t.gave_up = 1;
				break;
			}
			*hit_slot(a, RowRef{ 0, 0 }, t.count) = make_uint2(x, (uint32_t)a.out[state]);
			if (t.count == 0)
				t.first = x;
			t.last = x;
			t.count++;
		}
		x++;
	}
	write_summary(a, t, 0, 0);
}

// Text loads the compiler does not know of, with the waits placed by hand (a wait names the
// register it is for, so nothing that reads the register is scheduled above it).  The compiler's
// own bookkeeping cannot wait for one group of a sub-block at a time here: it merges what is
// pending around the loop (stores of the previous sub-block, which complete out of order with
// loads) and falls back to waiting for everything.  vmcnt(N) is safe with stores pending: of the
// operations that must have completed for N to be reached, at most the stores are not loads.
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load16(v4u &dst, const v4u *p)
{
	asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));
}
__device__ __forceinline__ void load16_nt(v4u &dst, const v4u *p)   // the same with the streaming hint
{
	asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p));
}
__device__ __forceinline__ void load4(uint32_t &dst, const uint32_t *p)
{
	asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p));
}
template <int N>
__device__ __forceinline__ void wait_loads(v4u &x, uint32_t &y)
{
	asm volatile("s_waitcnt vmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N));
}

// ------------------------------------------------------------------- K1 ---
// The bulk pass.  Persistent workgroups keep the Bloom filter in LDS; a wave owns a tile
// at a time, reads it 16 bytes per lane, tests one 3-gram per W bytes and writes the
// samples the filter flags -- position and 3-gram -- to the tile's list in position order
// (ranks from ballots: no scan, no LDS queue).  Nothing here waits on a dependent load.
// What the bulk kernel needs of a batch, read from the argument buffer once per batch and pinned
// in scalar registers (left to itself the compiler re-reads a field where it is used -- a scalar
// load and an lgkmcnt(0) between the LDS probes, which then wait for each other).
struct BulkBatch {
	const uint4 *text16;
	uint2 *shead, *samples;
	uint32_t *scount;
	uint32_t n, n_pad, tile_bytes, ntiles, scap;
	__device__ __forceinline__ explicit BulkBatch(const SieveArgs &a)
	    : text16(a.text16), shead(a.shead), samples(a.samples), scount(a.scount), n(a.n), n_pad(a.n_pad),
	      tile_bytes(a.tile_bytes), ntiles(a.ntiles), scap(a.scap)
	{
		asm volatile("" : "+s"(text16), "+s"(shead), "+s"(samples), "+s"(scount));
		asm volatile("" : "+s"(n), "+s"(n_pad), "+s"(tile_bytes), "+s"(ntiles), "+s"(scap));
	}
	__device__ __forceinline__ uint2 *slot(uint32_t tile, uint32_t idx) const
	{
		return idx < kSampleHead ? shead + (size_t)tile * kSampleHead + idx : samples + (size_t)tile * scap + idx;
	}
};

template <int W, bool DBG, int LG>
__global__ __launch_bounds__(kBlock) void k_sieve(SieveGroup g)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t bloom[];
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	constexpr uint32_t S = 16 / W;                              // samples per 16-byte group
	constexpr uint32_t LOADS = W >= 4 ? 8 : W == 2 ? 4 : 2;     // groups per lane per sub-block: S * LOADS <= 32 flag bits
	constexpr uint32_t SUB = LOADS * 1024;                      // bytes of a sub-block
	constexpr uint32_t SMASK = (1u << S) - 1u;
	const uint32_t nwaves = gridDim.x * kWaves;

	v4u w[LOADS];
	uint32_t nx[LOADS];
	uint32_t present = 0;   // bit j: group j of the sub-block exists for this lane
	constexpr bool NEXT = W <= 2 || (LG == 6 && W == 4);   // a sample's key reaches into the next group: 4 more bytes per lane
	constexpr int PER = NEXT ? 2 : 1;   // load instructions per group
	// the groups of the sub-block at 'off' of 'tile'.  Always exactly LOADS * PER load instructions,
	// whatever exists of the tile (what does not exist is read at the start of the text and masked
	// out): the waits count them.  With the streaming hint (nt): a read-only stream of this shape reaches
	// 6.0 TB/s with plain loads and 6.9 with the hint on this part (tools/micro/readbw.hip), and this kernel 5.4
	// against 5.7 (check and emit kernels off), the job 3.83 against 4.04 TB/s.  (Round 2 found plain loads
	// 7 % better -- on 320 MiB of texts, which the Infinity Cache then kept in part for the check kernel's
	// look at the bytes around every flagged sample; with the texts truly streamed from HBM there is
	// nothing to keep.  ACM_SIEVE_NT=0 brings the plain loads back.)
	const bool nt_loads = __builtin_amdgcn_readfirstlane((int)g.common.nt_loads) != 0;
	auto issue = [&](const BulkBatch &a, uint32_t tile, uint32_t off) {
		const uint32_t n16 = a.n_pad >> 4;
		const uint32_t *text32 = (const uint32_t *)a.text16;
		present = 0;
#pragma unroll
		for (uint32_t j = 0; j < LOADS; j++) {
			const uint32_t rel = off + j * 1024;
			const uint32_t g16 = ((tile * a.tile_bytes + rel) >> 4) + lane;   // wraps only for tiles that do not exist
			const bool ok = tile < a.ntiles && rel < a.tile_bytes && g16 < n16;
			present |= (ok ? 1u : 0u) << j;
			if (nt_loads)
				load16_nt(w[j], (const v4u *)a.text16 + (ok ? g16 : 0u));
			else
				load16(w[j], (const v4u *)a.text16 + (ok ? g16 : 0u));
			if (NEXT)
				load4(nx[j], text32 + (ok && g16 + 1 < n16 ? (size_t)g16 * 4 + 4 : 0));
			else
				nx[j] = 0;
		}
	};
	// group j has arrived (and everything issued before it)
	auto arrived = [&](uint32_t j) {
		if (j == 0) wait_loads<(LOADS - 1) * PER>(w[0], nx[0]);
		if (LOADS > 1 && j == 1) wait_loads<(LOADS > 1 ? LOADS - 2 : 0) * PER>(w[LOADS > 1 ? 1 : 0], nx[LOADS > 1 ? 1 : 0]);
		if (LOADS > 2 && j == 2) wait_loads<(LOADS > 2 ? LOADS - 3 : 0) * PER>(w[LOADS > 2 ? 2 : 0], nx[LOADS > 2 ? 2 : 0]);
		if (LOADS > 3 && j == 3) wait_loads<(LOADS > 3 ? LOADS - 4 : 0) * PER>(w[LOADS > 3 ? 3 : 0], nx[LOADS > 3 ? 3 : 0]);
		if (LOADS > 4 && j == 4) wait_loads<(LOADS > 4 ? LOADS - 5 : 0) * PER>(w[LOADS > 4 ? 4 : 0], nx[LOADS > 4 ? 4 : 0]);
		if (LOADS > 5 && j == 5) wait_loads<(LOADS > 5 ? LOADS - 6 : 0) * PER>(w[LOADS > 5 ? 5 : 0], nx[LOADS > 5 ? 5 : 0]);
		if (LOADS > 6 && j == 6) wait_loads<(LOADS > 6 ? LOADS - 7 : 0) * PER>(w[LOADS > 6 ? 6 : 0], nx[LOADS > 6 ? 6 : 0]);
		if (LOADS > 7 && j == 7) wait_loads<0>(w[LOADS > 7 ? 7 : 0], nx[LOADS > 7 ? 7 : 0]);
	};
	// the 3-gram of sample k of group j (registers)
	auto gram_of = [&](uint32_t j, uint32_t k) -> uint32_t {
		const uint32_t x[5] = { w[j].x, w[j].y, w[j].z, w[j].w, nx[j] };
		const uint32_t b = k * W, i = b / 4, sh = b % 4;
		const uint32_t v = sh == 0 ? x[i] : __builtin_amdgcn_alignbyte(x[i + 1 > 4 ? 4 : i + 1], x[i], sh);
		return v & 0xFFFFFFu;
	};
	// the 3 bytes behind the 3-gram (6-byte keys), else 0
	auto more_of = [&](uint32_t j, uint32_t k) -> uint32_t {
		if (LG != 6)
			return 0u;
		const uint32_t x[5] = { w[j].x, w[j].y, w[j].z, w[j].w, nx[j] };
		const uint32_t b = k * W + 3, i = b / 4, sh = b % 4;
		const uint32_t v = sh == 0 ? x[i > 4 ? 4 : i] : __builtin_amdgcn_alignbyte(x[i + 1 > 4 ? 4 : i + 1], x[i > 4 ? 4 : i], sh);
		return v & 0xFFFFFFu;
	};
	const uint32_t tile_first = blockIdx.x * kWaves + wv;
	const BulkBatch b0(batch_view(g, 0));   // the first batch's scalars now, with the filter still on its way
	unsigned long long *stamp = DBG && g.common.stamps ? g.common.stamps + (size_t)tile_first * 8 : nullptr;
	if (DBG && stamp && lane == 0)
		stamp[0] = __builtin_amdgcn_s_memrealtime();
	{
		// The filter goes to LDS by DMA (no registers, nothing waits yet), 1 KiB pieces dealt over the
		// waves; then the text loads; then a wait for the filter only -- the text keeps arriving
		// while the first groups are probed.
		const SieveArgs &a = g.common;
		const uint32_t pieces = a.bloom_words / 256;
		const uint32_t rot = (blockIdx.x * 7u) % pieces;   // the workgroups do not start on the same L2 channel
		for (uint32_t c = wv; c < pieces; c += kWaves) {
			uint32_t piece = c + rot;
			piece = piece >= pieces ? piece - pieces : piece;
			// (inline asm, not the builtin: the compiler makes every LDS read after a DMA it knows of wait
			// for ALL outstanding loads, the text included; the wait and the barrier below are what
			// orders the filter's arrival before its first use)
			const uint32_t lds_at = __builtin_amdgcn_readfirstlane(
			    (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(bloom + piece * 256));
			const uint4 *from = (const uint4 *)a.bloom + piece * 64 + lane;
			uint32_t keep_m0;
			asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
			    : "=&s"(keep_m0) : "s"(lds_at), "v"(from) : "memory");
		}
		issue(b0, tile_first, 0);
		asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS * PER) : "memory");
	}
	__builtin_amdgcn_s_barrier();   // (not __syncthreads: its fence would wait for the text as well)
	if (DBG && stamp && lane == 0)
		stamp[1] = __builtin_amdgcn_s_memrealtime();
	const uint32_t word_shift = 33 - g.common.bloom_log_words;   // 64-bit blocks
	const unsigned long long lt = (1ull << lane) - 1ull;   // the lanes in front of this one

	bool first = true;
	for (uint32_t bi = 0; bi < g.count; bi++) {
		const BulkBatch a(batch_view(g, bi));   // (a scalar-cache hit per batch: the workgroup's waves read the same lines)
		uint32_t tile = tile_first, off = 0;
		uint32_t qn = 0;   // samples of the current tile written so far
		while (tile < a.ntiles) {
			// A sub-block's loads are issued here and probed in order as they arrive (the very first
			// one's were issued above, in front of the filter's wait); the flagged samples are stored
			// once all groups are probed.  Other waves of the CU cover the latency.
			if (!first)
				issue(a, tile, off);
			first = false;
			const uint32_t base = tile * a.tile_bytes + off;
			uint32_t f = 0;
			// samples of group j whose 3-gram lies inside the text (none for groups that do not exist)
			auto room_of = [&](uint32_t j) -> uint32_t {
				const uint32_t pos0 = base + j * 1024 + lane * 16;
				const bool loaded = ((present >> j) & 1u) && pos0 + 2 < a.n;
				return loaded ? (a.n - 2 - pos0 + W - 1) / W : 0u;
			};
			// The filter blocks of two samples (a unit) are read together, as their group arrives
			// and without a branch around the reads, and looked at one unit later: the LDS reads of two
			// units are in flight instead of one sample's.  (Pinned below: left alone the compiler reads every
			// block of the sub-block first and tests afterwards, 122 registers instead of 80.)
			constexpr uint32_t U = 2, UPG = S / U, NU = LOADS * UPG;
			uint64_t blk[2][U];
#pragma unroll
			for (uint32_t u = 0; u <= NU; u++) {
				if (u < NU) {
					const uint32_t j = u / UPG, k0 = (u % UPG) * U;
					if (k0 == 0)
						arrived(j);
#pragma unroll
					for (uint32_t i = 0; i < U; i++)
						blk[u & 1][i] = bloom_block(bloom, gram_of(j, k0 + i), more_of(j, k0 + i), word_shift);
				}
				if (u > 0) {
					const uint32_t v = u - 1, j = v / UPG, k0 = (v % UPG) * U;
					const uint32_t room = room_of(j);
#pragma unroll
					for (uint32_t i = 0; i < U; i++)
						f |= (bloom_test(blk[v & 1][i], gram_of(j, k0 + i), more_of(j, k0 + i)) & (k0 + i < room ? 1u : 0u)) << (j * S + k0 + i);
				}
				// (an accumulated flag word nobody looks at before the end would let the compiler sink all the
				// bit tests behind the last read, the blocks of the whole sub-block live until then)
				asm volatile("" : "+v"(f));
				__builtin_amdgcn_sched_barrier(0);
			}
			if (DBG && stamp && lane == 0 && stamp[2] == 0)
				stamp[2] = __builtin_amdgcn_s_memrealtime();
			// the flagged samples go to the tile's list in position order: ranks from ballots
			if (__ballot(f != 0)) {
#pragma unroll
				for (uint32_t j = 0; j < LOADS; j++) {
					uint32_t fj = (f >> (j * S)) & SMASK;
					if (!__ballot(fj != 0))
						continue;
					{
						// Runs (zero pages, padding, zero-filled fields): every window a candidate of sample p
						// could start in lies in [p - W + 1, p + D), D <= 10.  If all of that is one byte b -- the
						// lane's own 16 bytes, and where the range leaves them the tail of the lane in front or the
						// head of the lane behind -- then unless D copies of b are a trie path none of the windows
						// is a pattern's prefix, and the check kernel is spared the sample.
						const uint32_t x = w[j].x;
						if (__ballot(fj != 0 && w[j].y == x)) {   // (one compare decides it on any text without runs)
							const bool same = w[j].y == x && w[j].z == x && w[j].w == x && __builtin_amdgcn_alignbyte(x, x, 1) == x;
							const uint32_t pz = __shfl_up(w[j].z, 1, 64), pw = __shfl_up(w[j].w, 1, 64);
							const uint32_t nx0 = __shfl_down(w[j].x, 1, 64), nx1 = __shfl_down(w[j].y, 1, 64);
							const bool left = lane > 0 && pz == x && pw == x;          // the 8 bytes in front of the lane's
							const bool right4 = lane < 63 && nx0 == x, right8 = right4 && nx1 == x;   // the 4 / 8 bytes behind them
							const uint32_t b = x & 0xFFu, wi = b >> 5;
							uint32_t word = g.common.run_ok[0];
#pragma unroll
							for (uint32_t i = 1; i < 8; i++)
								word = wi == i ? g.common.run_ok[i] : word;
							if (same && !((word >> (b & 31u)) & 1u)) {
#pragma unroll
								for (uint32_t k = 0; k < S; k++) {
									const bool lok = k * W >= W - 1 || left;
									constexpr uint32_t kD = acm::kSieveMaxPrefix;   // (the largest D there is: a compile-time bound -- with the run-time D in here the kernel took 106 registers instead of 94, too many for a check wave to fit next to two bulk workgroups)
									static_assert(kD <= 10, "the neighbours' tails looked at here reach 8 bytes beyond the lane's own 16");
									const uint32_t reach = k * W + kD;   // the candidate range ends here at the latest, counted from the lane's first byte
									const bool rok = reach <= 16 || (reach <= 20 ? right4 : reach <= 24 ? right8 : false);
									if (lok && rok)
										fj &= ~(1u << k);
								}
							}
						}
					}
					if (!__ballot(fj != 0))
						continue;
					const uint32_t pos0 = base + j * 1024 + lane * 16;
					uint32_t before = 0, all = 0;
#pragma unroll
					for (uint32_t k = 0; k < S; k++) {
						const unsigned long long bk = __ballot((fj >> k) & 1u);
						before += (uint32_t)__popcll(bk & lt);
						all += (uint32_t)__popcll(bk);
					}
#pragma unroll
					for (uint32_t k = 0; k < S; k++)
						if ((fj >> k) & 1u) {
							const uint32_t at = qn + before + (uint32_t)__popc(fj & ((1u << k) - 1u));
							*a.slot(tile, at) = make_uint2(pos0 + k * W, gram_of(j, k));
						}
					qn += all;
				}
			}
			const uint32_t cur = tile;
			off += SUB;
			if (off >= a.tile_bytes) {
				off = 0;
				tile += nwaves;
			}
			if (tile != cur) {
				if (lane == 0)
					a.scount[cur] = qn;
				qn = 0;
			}
		}
	}
	if (DBG && stamp && lane == 0)
		stamp[4] = __builtin_amdgcn_s_memrealtime();
}

// ------------------------------------------------------------------- K2 ---
// The checks.  A block of kTilesPerChecker consecutive tiles is a ROW of the emit kernel's input: its
// flagged samples, one list behind the other, are looked at 64 a round (stage 1); what stage 1 finds
// to be trie paths is followed 64 at a time (stage 2).  A few hundred waves, each a chain of dependent
// loads: latency-bound, light on everything else -- it runs beside the next batch's bulk pass.
// On real binaries the samples cluster (a block can hold thousands where most hold a dozen), so a
// block with more than kSubRow of them is cut into sub-rows of kSubRow samples: the block's own wave
// announces them on the batch's work list and takes the first; every wave that has done its own work
// -- and kHelperWaves that have none of their own -- takes sub-rows off the list until it is empty.
// Nobody waits for anybody: a block's wave empties the list itself if no one else does.  One more
// workgroup per batch walks the carried state and the tail.
template <uint32_t TPC>
struct BlockCounts {
	uint32_t cum[TPC + 1];   // samples in front of each of the block's tiles (wave-uniform)
	uint32_t mycount;        // of the tile this lane looks at in the speculative round
	uint2 spec;              // that tile's sample for this lane
	uint2 spec2;             // (16 tiles: four lanes per tile) the sample four further on: the lanes hold the tiles' first eight
};
static_assert(64 % kTilesPerChecker == 0 && 64 % kTilesPerCheckerWide == 0 && kSampleHead >= 64 / kTilesPerChecker,
    "the speculative first round deals the lanes evenly");

// the counts of a block's tiles and, in the same load level, the first samples of each: when no tile has
// more than its share of the lanes (with eight tiles: nearly always) that is the first and only stage-1 round
template <uint32_t TPC>
__device__ __forceinline__ void load_counts(const SieveArgs &a, uint32_t blk, uint32_t lane, BlockCounts<TPC> &bc)
{
	constexpr uint32_t kSpecLanes = 64 / TPC;   // lanes (= samples) per tile in the speculative round
	const uint32_t tile0 = blk * TPC;
	const uint32_t my_tile = tile0 + lane / kSpecLanes, my_idx = lane % kSpecLanes;
	const bool tile_ok = my_tile < a.ntiles;
	const uint32_t tile_c = tile_ok ? my_tile : tile0;
	bc.mycount = tile_ok ? a.scount[tile_c] : 0u;
	bc.spec = a.shead[(size_t)tile_c * kSampleHead + my_idx];
	bc.spec2 = 2 * kSpecLanes <= kSampleHead ? a.shead[(size_t)tile_c * kSampleHead + my_idx + kSpecLanes] : make_uint2(0, 0);
	bc.cum[0] = 0;
#pragma unroll
	for (uint32_t k = 0; k < TPC; k++)
		bc.cum[k + 1] = bc.cum[k] + (uint32_t)__builtin_amdgcn_readlane((int)bc.mycount, (int)(k * kSpecLanes));
}

// Sub-row j of block blk: its samples [j * kSubRow, (j + 1) * kSubRow) through both stages, its summary
// under `slot`.  word7: what a static row's summary says about its block's sub-rows.
// Returns the number of sub-rows the block has beyond its first.  (The counts are loaded in here, not handed in:
// carried around the caller's loop they cost the kernel thirty registers.)
// WHOLE: no sub-rows, the block is one row (no helper waves launched: cutting it up would only be bookkeeping).
template <int W, bool WHOLE, uint32_t TPC>
__device__ __forceinline__ uint32_t check_subrow(const SieveArgs &a, uint32_t (*q2)[kQ2Cap], uint32_t lane, uint32_t blk, uint32_t j,
    uint32_t slot, unsigned long long *stamp)
{
	constexpr uint32_t kSpecLanes = 64 / TPC;
	BlockCounts<TPC> bc;
	load_counts<TPC>(a, blk, lane, bc);
	const uint32_t tile0 = blk * TPC;
	uint32_t dbg_rounds = 0, dbg_cands = 0, dbg_levels = 0;
	Row t;
	t.carry = t.count = t.first = t.last = t.akey = t.anode = t.gave_up = 0;
	const uint32_t ns = bc.cum[TPC];
	const uint32_t extras = !WHOLE && ns > a.subrow ? (ns + a.subrow - 1) / a.subrow - 1 : 0u;
	const uint32_t lo = WHOLE ? 0u : j * a.subrow, hi = WHOLE ? ns : min(ns, lo + a.subrow);
	t.samples = j == 0 ? ns : 0u;   // (in the row's summary: an atomic on one counter would be 512 waves' loads waiting for it)
	// the block's list region; a sub-row's part of it starts where its first sample lies, plus the room
	// the sub-rows in front need beyond their spans (set below, once that sample is known)
	RowRef row;
	row.slot = slot;
	row.lbase = (blk + 1) * a.cap;
	t.room = a.cap;   // (a sub-row behind the first: what is left of the block's region, set with lbase below)
	FollowerQueue fq;
	fq.start = q2[0];
	fq.node = q2[1];
	fq.run = q2[2];
	fq.count = 0;
	uint32_t r0 = lo;
	bool placed = j == 0;
	if (j == 0 && ns && !__ballot(bc.mycount > kSpecLanes)) {
		dbg_rounds++;
		dbg_cands += ns;
		r0 = ns;
		stage1_round<W>(a, fq, bc.spec.x, bc.spec.y, lane % kSpecLanes < bc.mycount, lane);
	}
	for (;;) {
		if (fq.count >= 64 || (r0 >= hi && fq.count > 0)) {   // stage 2: a round of followers
			const uint32_t cnt = min(fq.count, 64u);
			const bool act = lane < cnt;
			const uint32_t s0 = act ? fq.start[lane] : 0u, nd = act ? fq.node[lane] : 0u;
			const uint32_t rn = act ? fq.run[lane] : 0u;
			// what is left moves to the front of the queue
			const uint32_t left = fq.count - cnt;
			uint32_t m0 = 0, m1 = 0, m2 = 0;
			if (lane < left) {
				m0 = fq.start[cnt + lane];
				m1 = fq.node[cnt + lane];
				m2 = fq.run[cnt + lane];
			}
			__builtin_amdgcn_wave_barrier();
			if (lane < left) {
				fq.start[lane] = m0;
				fq.node[lane] = m1;
				fq.run[lane] = m2;
			}
			for (uint32_t i = 64 + lane; i < left; i += 64) {   // more than 64 left: ascending copy is safe (i < cnt + i)
				fq.start[i] = fq.start[cnt + i];
				fq.node[i] = fq.node[cnt + i];
				fq.run[i] = fq.run[cnt + i];
				__builtin_amdgcn_wave_barrier();
			}
			fq.count = left;
			stage2_round(a, t, row, s0, nd, rn, act, lane, dbg_levels);
			continue;
		}
		if (r0 >= hi)
			break;
		const uint32_t idx = r0 + lane;
		const bool act = idx < hi;
		uint32_t tslot = 0, before = 0;
#pragma unroll
		for (uint32_t k = 1; k < TPC; k++) {
			tslot += idx >= bc.cum[k] ? 1u : 0u;
			before = idx >= bc.cum[k] ? bc.cum[k] : before;
		}
		// the first eight samples of every tile came with the counts (load_counts): from the lane that holds them,
		// no load level; only what lies deeper in a tile's list is fetched
		const uint32_t kk = idx - before;
		const int src = (int)(tslot * kSpecLanes + kk % kSpecLanes);
		uint2 it = make_uint2((uint32_t)__shfl((int)bc.spec.x, src, 64), (uint32_t)__shfl((int)bc.spec.y, src, 64));
		if (2 * kSpecLanes <= kSampleHead) {
			const uint2 it2 = make_uint2((uint32_t)__shfl((int)bc.spec2.x, src, 64), (uint32_t)__shfl((int)bc.spec2.y, src, 64));
			if (kk >= kSpecLanes)
				it = it2;
		}
		constexpr uint32_t kHeld = 2 * kSpecLanes <= kSampleHead ? 2 * kSpecLanes : kSpecLanes;
		if (!act)
			it = make_uint2(0, 0);
		else if (kk >= kHeld)
			it = *sample_slot(a, tile0 + tslot, kk);
		if (!placed) {
			// (hits of this sub-row lie at or behind its first sample's position less W, one per position:
			// the span of its samples + the longest pattern + W entries are enough, sub_k allows for that)
			const uint32_t p_first = (uint32_t)__builtin_amdgcn_readlane((int)it.x, 0);
			row.lbase += (p_first - tile0 * a.tile_bytes) + j * a.sub_k;
			t.room = (blk + 2) * a.cap - row.lbase;
			placed = true;
		}
		dbg_rounds++;
		dbg_cands += min(hi - r0, 64u);
		r0 += 64;
		stage1_round<W>(a, fq, it.x, it.y, act, lane);
	}
	if (lane == 0)
		write_summary(a, t, slot, j == 0 ? extras : row.lbase);
	if (stamp && lane == 0) {
		stamp[4] = __builtin_amdgcn_s_memrealtime();
		stamp[5] = ((unsigned long long)dbg_rounds << 32) | dbg_cands;
		stamp[6] = ((unsigned long long)dbg_levels << 32);
	}
	return extras;
}

// lane 0 asks, every lane gets the answer
#define ACM_LANE0(expr)                                                    \
	([&]() -> uint32_t {                                               \
		uint32_t v_ = 0;                                           \
		if (lane == 0)                                             \
			v_ = (expr);                                       \
		return (uint32_t)__builtin_amdgcn_readfirstlane((int)v_);  \
	}())

// HELPED: helper waves were launched (the last batches were sample-heavy).  Without them (the common case) every
// block is one row and the kernel is a straight line: the loop of the helped kind costs 40 registers, and next to
// two bulk workgroups a SIMD has 128 left for the checks of the other streams' batches.
// TPC: tiles per block = per row.  8 when a batch is launched alone: the first stage-1 round then has the samples of
// nearly every block in one load level, 36 us for a batch; 16 in a launch group of four or more: half the waves,
// each with twice the lanes busy in the followers' levels -- the kernel is paced by waves x levels in the shadow
// of the other streams' bulk kernels: 4.09 -> 4.36 TB/s (40 us for a batch alone; 32 tiles: 4.24 and 53).
template <int W, bool HELPED, uint32_t TPC>
__global__ __launch_bounds__(kCheckBlock) void k_sieve_check(SieveGroup g, uint32_t helpers)
{
	__shared__ uint32_t q2[3][kQ2Cap];
	const uint32_t lane = threadIdx.x;
	// The workgroups of a launch are dealt over the group's batches, a multiple of 8 each: workgroup ids
	// go round the 8 XCDs, and block r of EVERY batch should land on the XCD whose bulk-kernel workgroup
	// r read the block's tiles a moment ago -- its L2 still has them.  Per batch: the blocks, the
	// workgroup of the serial walks, the helpers.
	const uint32_t nblocks = g.common.nrows - 1;
	const uint32_t nhelp = HELPED ? helpers : 0u;
	const uint32_t per = (nblocks + 1 + nhelp + 7u) & ~7u, bi = blockIdx.x / per, blk = blockIdx.x - bi * per;
	if (blk > nblocks + nhelp)
		return;
	const SieveArgs a = batch_view(g, bi);
	if (blk == nblocks) {
		if (threadIdx.x == 0)
			side_walks(a);
		return;
	}
	unsigned long long *stamp = kCheckStamps && a.stamps && blk < nblocks ? a.stamps + (size_t)(blk + 8192) * 8 : nullptr;
	if (stamp && lane == 0)
		stamp[0] = __builtin_amdgcn_s_memrealtime();
	if constexpr (!HELPED) {
		(void)check_subrow<W, true, TPC>(a, q2, lane, blk, 0, blk + 1, stamp);
	} else {
		// The sub-rows behind the first of every block are numbered through the batch in block order and dealt to
		// the helper waves.  A lane takes eight neighbouring blocks (sixty-four tiles: the text has at most 4096),
		// the wave's prefix sum gives each block the number of its first sub-row.  (The per-lane numbers live in
		// LDS, not in sixteen registers carried through both stages.)
		__shared__ uint32_t s_extra[8][kCheckBlock], s_first[8][kCheckBlock];
		bool own = blk < nblocks;   // a block's wave does the block's own row = sub-row 0, nothing else
		uint32_t base = 0, all = 0;
		if (!own) {
			const uint4 *sc = (const uint4 *)a.scount;
			uint32_t mine = 0;
#pragma unroll
			for (uint32_t i = 0; i < 8; i++) {
				const uint32_t t0 = (lane * 8 + i) * TPC;
				uint32_t ns = 0;
#pragma unroll
				for (uint32_t q = 0; q < TPC / 4; q++) {
					uint4 v = make_uint4(0, 0, 0, 0);
					if (t0 + 4 * q < a.ntiles)   // (the counts behind the last tile are whatever they are: masked below)
						v = sc[t0 / 4 + q];
					const uint32_t c[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
					for (uint32_t k = 0; k < 4; k++)
						ns += t0 + 4 * q + k < a.ntiles ? c[k] : 0u;
				}
				const uint32_t ex = ns > a.subrow ? (ns + a.subrow - 1) / a.subrow - 1 : 0u;
				s_extra[i][lane] = ex;
				s_first[i][lane] = mine;
				mine += ex;
			}
			base = wave_excl_sum(mine, lane, all);
		}
		// one loop for everything a wave does, so that the two stages are in the kernel once
		for (uint32_t s0 = blk - nblocks - 1;; s0 += nhelp) {
			uint32_t tb = blk, tj = 0, tslot = blk + 1;
			if (!own) {
				if (s0 >= all)
					return;
				// sub-row number s0 -> its block and index: the lane whose blocks hold it says so
				uint32_t fb = 0, fk = 0;
				bool found = false;
#pragma unroll
				for (uint32_t i = 0; i < 8; i++) {
					const uint32_t lo = base + s_first[i][lane];
					if (s0 >= lo && s0 < lo + s_extra[i][lane]) {
						found = true;
						fb = lane * 8 + i;
						fk = s0 - lo + 1;
					}
				}
				const unsigned long long who = __ballot(found);
				if (!who)
					return;   // (cannot happen: every number below `all` belongs to a block)
				const int src = (int)__builtin_ctzll(who);
				tb = (uint32_t)__builtin_amdgcn_readlane((int)fb, src);
				tj = (uint32_t)__builtin_amdgcn_readlane((int)fk, src);
				tslot = a.nrows + s0;
			}
			(void)check_subrow<W, false, TPC>(a, q2, lane, tb, tj, tslot, own ? stamp : nullptr);
			if (own)
				return;
		}
	}
}

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

// ------------------------------------------------------------------- K3 ---
// Every workgroup works out the whole prefix (the 32-byte summaries, L2 resident: exclusive prefix
// max of the extents, exclusive prefix sum of what the shadow leaves of each list) and copies its
// share of the records: a thread per output cell finds the cell's row by binary search in LDS and
// moves one record, so that the stores of a wave lie side by side.  (The owner of a row copying the
// row's records itself, one predicated store per possible record, took three times as long: it is
// the instruction count of 1024 threads that matters here.)
// Rows in position order: the walker's, then block by block the block's own row and its sub-rows
// (k_sieve_check), kEmitBlock rows a round -- one round unless the text is sample-heavy.
// PLAIN: the check kernel ran without helper waves, i.e. no block has sub-rows: rows = static rows = threads.
// (Known at compile time, the kernel needs 40 registers instead of 75.)
template <bool PLAIN>
__global__ __launch_bounds__(kEmitBlock) void k_sieve_emit(SieveGroup g)
{
	const uint32_t per = gridDim.x / g.count, bi = blockIdx.x / per, blk = blockIdx.x - bi * per;
	const SieveArgs a = batch_view(g, bi);
	__shared__ uint32_t lds[kEmitBlock / 64];
	__shared__ uint32_t s_base[kEmitBlock + 1];   // first output cell of each row of the round
	__shared__ uint32_t s_drop[kEmitBlock];       // shadowed head of each row's list
	__shared__ uint32_t s_slot[kEmitBlock], s_lbase[kEmitBlock];   // where each row keeps its hits (RowRef)
	__shared__ uint32_t s_ord[kMaxRows + 1];      // static row -> its place in the order of all rows
	__shared__ uint32_t s_first[kMaxRows];        // static row -> work-list index of its first sub-row
	__shared__ unsigned long long s_alive;
	__shared__ uint32_t s_samples;                // flagged samples of the batch
	const uint32_t statics = a.nrows, tid = threadIdx.x;
	unsigned long long *stamp = a.stamps && threadIdx.x == 0 ? a.stamps + (size_t)(9000 + blk) * 8 : nullptr;
	if (stamp)
		stamp[0] = __builtin_amdgcn_s_memrealtime();
	// every kernel argument the kernel will need, fetched now, together (the compiler would fetch each
	// where it is first used: a fetch from the argument buffer in front of every phase)
	asm volatile("" :: "s"(a.pat_plane), "s"(a.off_plane), "s"(a.plane_capacity), "s"(a.off_shift), "s"(a.lhead),
	    "s"(a.lists), "s"(a.cap), "s"(a.misc), "s"(a.dev2ref), "s"(a.path_marker), "s"(a.giveups), "s"(a.n));
	if (tid == 0) {
		s_alive = ~0ull;
		s_samples = 0;
	}
	static_assert(kMaxRows <= kEmitBlock, "a thread per static row");
	// the static rows' summaries, a thread each (loads without a branch around them: in flight together)
	uint4 s0, al0;
	{
		const uint32_t rc = min(tid, statics - 1);
		s0 = *(const uint4 *)(a.summary + (size_t)rc * kSummaryWords);
		al0 = *(const uint4 *)(a.summary + (size_t)rc * kSummaryWords + 4);   // alive key, node, samples, sub-rows
		if (tid >= statics) {
			s0 = make_uint4(0, 0, 0, 0);
			al0 = make_uint4(0, 0, 0, 0);
		}
	}
	uint32_t my_samples = al0.z, rows_all;
	const uint32_t extra = PLAIN || tid == 0 ? 0u : al0.w & 0xFFFFu;   // (row 0 is the walker's: its word says nothing)
	if constexpr (PLAIN) {
		rows_all = statics;   // (row = thread: no order to work out)
	} else {
		const uint32_t ord = block_exclusive<false>(tid < statics ? 1u + extra : 0u, lds, &rows_all);
		if (tid < statics) {
			s_ord[tid] = ord;
			s_first[tid] = ord - tid;   // sub-rows of the blocks in front: the number of this block's first (k_sieve_check)
		}
		if (tid == 0)
			s_ord[statics] = rows_all;
	}
	if (blk == 0) {   // (a wave at a time: five hundred atomics on one LDS word would take longer than the copy below)
#pragma unroll
		for (int o = 32; o > 0; o >>= 1)
			my_samples += __shfl_xor(my_samples, o, 64);
		if ((tid & 63) == 0 && my_samples)
			atomicAdd(&s_samples, my_samples);
	}
	__syncthreads();
	if (stamp)
		stamp[1] = __builtin_amdgcn_s_memrealtime();
	const bool plain = PLAIN || rows_all == statics;   // no sub-rows: row = thread, everything is loaded already
	uint32_t carry_in = 0, cell_in = 0;
	unsigned long long alive = ~0ull;
	bool gave_any = false;
	for (uint32_t o0 = 0; o0 < rows_all; o0 += kEmitBlock) {
		const uint32_t o = o0 + tid;
		const bool have = o < rows_all;
		uint4 su = s0, sa = al0;
		RowRef rr;
		rr.slot = tid;
		rr.lbase = tid * a.cap;
		if (!PLAIN && !plain) {
			// the static row this row belongs to: the last one whose place is not behind o
			uint32_t lo = 0, hi = statics;
			while (hi - lo > 1) {
				const uint32_t mid = (lo + hi) >> 1;
				if (s_ord[mid] <= o)
					lo = mid;
				else
					hi = mid;
			}
			const uint32_t k = o - s_ord[lo];
			rr.slot = k == 0 ? lo : statics + s_first[lo] + k - 1;
			rr.lbase = lo * a.cap;
			su = make_uint4(0, 0, 0, 0);
			sa = make_uint4(0, 0, 0, 0);
			if (have) {
				su = *(const uint4 *)(a.summary + (size_t)rr.slot * kSummaryWords);
				sa = *(const uint4 *)(a.summary + (size_t)rr.slot * kSummaryWords + 4);
				if (k)
					rr.lbase = sa.w;
			}
		}
		const uint32_t E = su.x, cnt = su.y == kGaveUp ? 0u : su.y, first = su.z, last = su.w;
		gave_any |= su.y == kGaveUp;
		if (have && sa.x)
			alive = min(alive, ((unsigned long long)sa.x << 32) | sa.y);
		uint32_t all_max, all_cells;
		const uint32_t carry = max(carry_in, block_exclusive<true>(have ? E : 0u, lds, &all_max));
		uint32_t d = 0;
		if (have && cnt && first < carry) {
			if (last < carry) {
				d = cnt;
			} else {
				while (d < cnt && hit_slot(a, rr, d)->x < carry)
					d++;
			}
		}
		const uint32_t kept = have ? cnt - d : 0u;
		const uint32_t cell = cell_in + block_exclusive<false>(kept, lds, &all_cells);
		s_base[tid] = cell;
		s_drop[tid] = d;
		if constexpr (!PLAIN) {
			s_slot[tid] = rr.slot;
			s_lbase[tid] = rr.lbase;
		}
		if (tid == 0)
			s_base[kEmitBlock] = cell_in + all_cells;
		__syncthreads();
		// one thread per output cell of the round: find the row it belongs to, copy the record
		const uint32_t rows_here = min(rows_all - o0, (uint32_t)kEmitBlock), cell_end = cell_in + all_cells;
		for (uint32_t c = cell_in + blk * kEmitBlock + tid; c < cell_end; c += per * kEmitBlock) {
			uint32_t lo = 0, hi = rows_here;   // the row r with s_base[r] <= c < s_base[r + 1]
			while (hi - lo > 1) {
				const uint32_t mid = (lo + hi) >> 1;
				if (s_base[mid] <= c)
					lo = mid;
				else
					hi = mid;
			}
			RowRef from;
			from.slot = PLAIN ? lo : s_slot[lo];
			from.lbase = PLAIN ? lo * a.cap : s_lbase[lo];
			const uint2 rec = *hit_slot(a, from, s_drop[lo] + (c - s_base[lo]));
			if (c + 2 < a.plane_capacity) {
				a.pat_plane[1 + c] = (int32_t)rec.y;
				a.off_plane[1 + c] = (int32_t)rec.x + a.off_shift;
			}
		}
		carry_in = max(carry_in, all_max);
		cell_in = cell_end;
		__syncthreads();
	}
	if (__syncthreads_or(gave_any ? 1 : 0)) {   // cannot happen (geometry_for): say so instead of leaving planes that look complete
		if (blk == 0 && tid == 0) {
			*a.path_marker = kMarkerFailed;
			a.pat_plane[0] = a.off_plane[0] = 0;
		}
		return;
	}
	if (alive != ~0ull)
		atomicMin(&s_alive, alive);
	__syncthreads();
	if (stamp)
		stamp[3] = __builtin_amdgcn_s_memrealtime();
	if (blk == 0 && tid == 0) {   // header and trailer cells
		const uint32_t all_cells = cell_in;
		const unsigned long long k = s_alive;
		const uint32_t last_dev = k != ~0ull ? (uint32_t)k : a.misc[0];
		const int32_t last_ref = (int32_t)a.dev2ref[last_dev];
		uint32_t tail = all_cells + 1;
		if (tail > a.plane_capacity - 1)
			tail = a.plane_capacity - 1;
		a.pat_plane[0] = (int32_t)all_cells;
		a.off_plane[0] = (int32_t)all_cells;
		a.pat_plane[tail] = last_ref;
		a.off_plane[tail] = last_ref;
		*a.path_marker = (uint32_t)ACM_SCAN_MODE_SPARSE;
		// a batch this dense in matches, or with this many samples for the check kernel to look at (real
		// binaries: common 3-grams of code), is the chain pipeline's: AUTO mode counts them (scan.hip,
		// pick_sparse)
		if (a.giveups)   // what the host sizes the next check launches by (sparse_group_enqueue)
			__hip_atomic_store(a.giveups + 1, s_samples, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		if (a.giveups && (all_cells > a.n / kDenseDivisor || s_samples > a.n / kBusyDivisor))
			__hip_atomic_fetch_add(a.giveups, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	if (stamp)
		stamp[4] = __builtin_amdgcn_s_memrealtime();
}

size_t align_up(size_t v, size_t al) { return (v + al - 1) / al * al; }

struct Geometry {
	uint32_t tile_bytes, ntiles, cap, scap, nrows, sub_k, max_extra;
};

// Nothing overflows: a tile's sample list has room for every sample of the tile, a row's hit list
// for one hit per text position the followers of its tiles can reach (the shadow rule leaves at
// most one per position).
Geometry geometry_for(const acm_dfa *d, size_t n, uint32_t tpc = kTilesPerChecker)
{
	Geometry g;
	g.tile_bytes = kMinTile;
	while ((n + g.tile_bytes - 1) / g.tile_bytes > kMaxTiles)
		g.tile_bytes *= 2;
	g.ntiles = (uint32_t)((n + g.tile_bytes - 1) / g.tile_bytes);
	g.nrows = (g.ntiles + tpc - 1) / tpc + 1;
	g.scap = g.tile_bytes / (d->sv_stride ? d->sv_stride : 1);
	// sub-rows (k_sieve_check): a block has at most maxsub, each needs sub_k entries beyond its span
	g.sub_k = d->max_pattern_len + (d->sv_stride ? d->sv_stride : 1) + 8;
	const uint32_t maxsub = (tpc * g.scap + kSubRowMin - 1) / kSubRowMin;
	g.cap = tpc * g.tile_bytes + maxsub * g.sub_k + 8;
	g.max_extra = (uint32_t)(((size_t)g.ntiles * g.scap + kSubRowMin - 1) / kSubRowMin) + 1;
	return g;
}

}  // namespace

namespace acm {

namespace {
size_t workspace_for_tpc(const acm_dfa *d, size_t n, uint32_t tpc)
{
	const Geometry g = geometry_for(d, n, tpc);   // tile size and cap grow with the text, the tile count is bounded
	size_t o = 0;
	const size_t rows_max = (size_t)g.nrows + g.max_extra;
	o += align_up(rows_max * kSummaryWords * 4, 256);
	o += align_up((size_t)g.nrows * g.cap * 8, 256);
	o += align_up((size_t)g.ntiles * g.scap * 8 + 64, 256);
	o += align_up((size_t)kMaxTiles * kSampleHead * 8, 256);
	o += align_up((rows_max + 4) * kHitHead * 8, 256);
	o += align_up((size_t)kMaxTiles * 4, 256);
	o += 256;
	return o;
}
size_t workspace_for_exactly(const acm_dfa *d, size_t n)   // (either block size: which one a scan uses depends on how it is launched)
{
	return std::max(workspace_for_tpc(d, n, kTilesPerChecker), workspace_for_tpc(d, n, kTilesPerCheckerWide));
}
}  // namespace

// Enough for every text of up to max_text bytes.  The need is not monotonic in n: where the tile
// size doubles the number of rows halves while every row keeps its per-row terms, so the largest
// text of each smaller tile size is looked at too (a scan lays the workspace out for its own n).
size_t sparse_workspace_bytes(const acm_dfa *d, size_t max_text)
{
	if (!d->sparse_ok)
		return 0;
	size_t need = workspace_for_exactly(d, max_text);
	for (size_t tile = kMinTile; tile * kMaxTiles < max_text; tile *= 2)
		need = std::max(need, workspace_for_exactly(d, tile * kMaxTiles));
	return need;
}

int sparse_prepare(const acm_dfa *)
{
	const int lds = (int)((1u << acm::kSieveMaxLogWords) * 4);
	const void *kernels[] = { (const void *)k_sieve<8, false, 3>, (const void *)k_sieve<4, false, 3>, (const void *)k_sieve<2, false, 3>,
		(const void *)k_sieve<1, false, 3>, (const void *)k_sieve<8, true, 3>, (const void *)k_sieve<4, true, 3>,
		(const void *)k_sieve<2, true, 3>, (const void *)k_sieve<1, true, 3>, (const void *)k_sieve<8, false, 6>,
		(const void *)k_sieve<4, false, 6>, (const void *)k_sieve<8, true, 6>, (const void *)k_sieve<4, true, 6> };
	for (const void *k : kernels)
		ACM_HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
	return ACM_OK;
}

namespace {
// the shared part of a group's kernel arguments: tables, geometry for texts of n bytes, workspace layout
void fill_common(const acm_dfa *d, size_t n, SieveGroup &grp, Geometry &g, uint32_t tpc)
{
	g = geometry_for(d, n, tpc);
	SieveArgs &a = grp.common;
	a.bloom = d->d_sv_bloom;
	a.bloom_log_words = d->sv_bloom_log_words;
	a.bloom_words = 1u << d->sv_bloom_log_words;
	a.gram = (const uint4 *)d->d_sv_gram;
	a.gram_log_buckets = d->sv_gram_log_buckets;
	a.gram_probes = d->sv_gram_probes;
	a.prefix = (const uint4 *)d->d_sv_prefix;
	a.prefix_log_slots = d->sv_prefix_log_slots;
	a.prefix_probes = d->sv_prefix_probes;
	a.rec = (const uint4 *)d->d_sv_rec;
	a.edges = (const uint4 *)d->d_sv_edges;
	a.in_byte = d->d_in_byte;
	a.cold = d->d_cold;
	a.cls = d->d_class;
	a.ls = d->log_stride;
	a.depth = d->d_depth;
	a.out = d->d_out;
	a.dev2ref = d->d_dev2ref;
	a.F = d->first_final;
	a.D = d->sv_prefix_len;
	static const uint32_t nt = getenv("ACM_SIEVE_NT") ? (uint32_t)atoi(getenv("ACM_SIEVE_NT")) : 1u;   // debugging aid: 0 = plain loads
	a.nt_loads = nt;
	a.tail_walk = a.D - 1;
	if (d->sv_gram_len == 6)
		a.tail_walk = std::max(a.tail_walk, d->sv_stride + 4);
	memcpy(a.run_ok, d->sv_run_ok, sizeof(a.run_ok));
	a.n = (uint32_t)n;
	a.n_pad = (uint32_t)((n + 15) & ~(size_t)15);
	a.tile_bytes = g.tile_bytes;
	a.ntiles = g.ntiles;
	a.cap = g.cap;
	a.nrows = g.nrows;
	a.max_extra = g.max_extra;
	a.sub_k = g.sub_k;
	static const uint32_t subrow = getenv("ACM_SIEVE_SUBROW") ? std::max<uint32_t>(kSubRowMin, (uint32_t)atoi(getenv("ACM_SIEVE_SUBROW"))) : kSubRow;
	a.subrow = subrow;
	a.scap = g.scap;
	a.giveups = d->d_giveups;
	size_t o = 0;
	auto take = [&](size_t bytes) {
		const size_t at = o;
		o += align_up(bytes, 256);
		return at;
	};
	const size_t rows_max = (size_t)g.nrows + g.max_extra;
	grp.o_summary = take(rows_max * kSummaryWords * 4);
	grp.o_lists = take((size_t)g.nrows * g.cap * 8);
	grp.o_samples = take((size_t)g.ntiles * g.scap * 8);
	grp.o_shead = take((size_t)g.ntiles * kSampleHead * 8);
	grp.o_lhead = take((rows_max + 4) * kHitHead * 8);
	grp.o_scount = take((size_t)g.ntiles * 4);
	grp.o_misc = take(256);
}

void fill_batch(const SieveJob &job, SieveBatch &b)
{
	const acm_scan_batch *in = job.batch;
	b.text = (const uint8_t *)in->d_text;
	b.ws = (char *)job.sparse_ws;
	b.pat_plane = in->d_pat_plane;
	b.off_plane = in->d_off_plane;
	b.path_marker = job.path_marker;
	b.init_state = job.init_dev;
	b.init_ptr = job.init_ptr;
	b.drop_before = (uint32_t)in->halo;
	b.off_shift = (int32_t)in->offset_shift;
	b.plane_capacity = (uint32_t)(in->plane_capacity > 0xFFFFFFFFul ? 0xFFFFFFFFul : in->plane_capacity);
	b.report_state = in->report == ACM_REPORT_STATE ? 1 : 0;
}
}  // namespace

int sparse_group_enqueue(const acm_dfa *d, const SieveJob *jobs, uint32_t count, hipStream_t s, hipEvent_t after_sieve,
    hipEvent_t after_emit)
{
	if (count == 0 || count > kMaxGroup)
		return acm::fail(ACM_ERR_ARG, "sparse_group_enqueue: %u batches", count);
	for (uint32_t i = 1; i < count; i++)
		if (jobs[i].batch->n != jobs[0].batch->n)
			return acm::fail(ACM_ERR_ARG, "sparse_group_enqueue: the batches of a group must have one size");
	const size_t n = jobs[0].batch->n;
	SieveGroup grp;
	memset(&grp, 0, sizeof(grp));
	grp.count = count;
	// Helper waves for the sub-rows of sample-heavy blocks only where the text is like that: the emit
	// kernel leaves the batch's sample count in pinned host memory, and a launch whose stream's last batch
	// had more than a flagged sample per 512 bytes gets them (a block is one row whatever it holds
	// otherwise -- the first heavy batch after quiet ones is slow).  On quiet text thousands more
	// workgroups per batch, each reading the batch's 4096 tile counts, cost a launch group of sixteen 20 us.
	static const char *force = getenv("ACM_SIEVE_HELPERS");   // debugging aid: "0" never, anything else always
	const uint32_t seen = d->h_giveups ? ((volatile uint32_t *)d->h_giveups)[1] : 0u;
	const bool heavy = force ? force[0] != '0' : seen > n / kHeavyDivisor;
	static const uint32_t helper_waves = getenv("ACM_SIEVE_HELPER_WAVES") ? (uint32_t)atoi(getenv("ACM_SIEVE_HELPER_WAVES")) : kHelperWaves;   // debugging aid
	const uint32_t helpers = heavy ? helper_waves : 0u;
	// blocks of 16 tiles for a launch group without helpers, of 8 otherwise (k_sieve_check)
	static const char *force_tpc = getenv("ACM_SIEVE_TPC");   // debugging aid: 8 or 16 always
	const uint32_t tpc = force_tpc ? (atoi(force_tpc) == 16 && !helpers ? kTilesPerCheckerWide : kTilesPerChecker)
	                               : (!helpers && count >= 4 ? kTilesPerCheckerWide : kTilesPerChecker);
	Geometry g;
	fill_common(d, n, grp, g, tpc);
	for (uint32_t i = 0; i < count; i++)
		fill_batch(jobs[i], grp.b[i]);
	SieveArgs &a = grp.common;
	static unsigned long long *d_stamps = nullptr;
	const bool want_stamps = count == 1 && getenv("ACM_SIEVE_STAMPS") != nullptr;
	const size_t stamp_words = (size_t)(9000 + 128) * 8;
	if (want_stamps) {
		if (!d_stamps)
			ACM_HIP_TRY(hipMalloc((void **)&d_stamps, stamp_words * 8));
		ACM_HIP_TRY(hipMemsetAsync(d_stamps, 0, stamp_words * 8, s));
		a.stamps = d_stamps;
	}

	// K1: persistent workgroups of 8 waves, a tile per wave at a time, at most TWO workgroups per CU
	// (it asks for more than a third of the LDS whatever the filter's size): 8 waves with 8 KiB in
	// flight each keep the memory system as busy as 16 do (measured: 16 are 1 us slower per launch,
	// 4 are 3 us slower; a second register set prefetching the wave's next tile gains nothing
	// either).  One launch fills a CU's first slot; the second slot lets the next stream's bulk
	// kernel ramp up while this one drains (4.0 instead of 3.8 TB/s), and half of the CU's wave
	// slots and registers still stay free for the check and emit kernels of the batches in flight
	// on other streams -- with bulk workgroups resident everywhere those would wait for a whole
	// bulk workgroup to drain, every time.
	static const size_t lds_kb = getenv("ACM_SIEVE_LDS_KB") ? (size_t)atoi(getenv("ACM_SIEVE_LDS_KB")) : 56;   // debugging aid
	const size_t lds = std::max((size_t)a.bloom_words * 4, lds_kb * 1024);
	uint32_t blocks = (g.ntiles + kWaves - 1) / kWaves;
	if (blocks > (uint32_t)d->num_cus)
		blocks = (uint32_t)d->num_cus;
	// K2: a wave per kTilesPerChecker tiles, and one workgroup for the serial walks
	const uint32_t cwaves = g.nrows - 1;
	const uint32_t cblocks = (cwaves + 1 + helpers + 7u) & ~7u;   // (a multiple of 8 per batch: k_sieve_check)
	uint32_t eblocks = 2;   // a power of two; each works out the whole prefix, more of them only for the copies
	while (eblocks < 64 && ((size_t)eblocks << 22) < n)   // (a CU issues scattered 4-byte stores one a clock)
		eblocks *= 2;
	const bool long_keys = d->sv_gram_len == 6;
	if (!want_stamps) {
		switch (d->sv_stride) {
		case 8:
			if (long_keys) hipLaunchKernelGGL((k_sieve<8, false, 6>), dim3(blocks), dim3(kBlock), lds, s, grp);
			else hipLaunchKernelGGL((k_sieve<8, false, 3>), dim3(blocks), dim3(kBlock), lds, s, grp);
			break;
		case 4:
			if (long_keys) hipLaunchKernelGGL((k_sieve<4, false, 6>), dim3(blocks), dim3(kBlock), lds, s, grp);
			else hipLaunchKernelGGL((k_sieve<4, false, 3>), dim3(blocks), dim3(kBlock), lds, s, grp);
			break;
		case 2: hipLaunchKernelGGL((k_sieve<2, false, 3>), dim3(blocks), dim3(kBlock), lds, s, grp); break;
		default: hipLaunchKernelGGL((k_sieve<1, false, 3>), dim3(blocks), dim3(kBlock), lds, s, grp); break;
		}
	} else {   // debugging aid: the same kernel with clock stamps
		switch (d->sv_stride) {
		case 8:
			if (long_keys) hipLaunchKernelGGL((k_sieve<8, true, 6>), dim3(blocks), dim3(kBlock), lds, s, grp);
			else hipLaunchKernelGGL((k_sieve<8, true, 3>), dim3(blocks), dim3(kBlock), lds, s, grp);
			break;
		case 4:
			if (long_keys) hipLaunchKernelGGL((k_sieve<4, true, 6>), dim3(blocks), dim3(kBlock), lds, s, grp);
			else hipLaunchKernelGGL((k_sieve<4, true, 3>), dim3(blocks), dim3(kBlock), lds, s, grp);
			break;
		case 2: hipLaunchKernelGGL((k_sieve<2, true, 3>), dim3(blocks), dim3(kBlock), lds, s, grp); break;
		default: hipLaunchKernelGGL((k_sieve<1, true, 3>), dim3(blocks), dim3(kBlock), lds, s, grp); break;
		}
	}
	if (after_sieve)
		ACM_HIP_TRY(hipEventRecord(after_sieve, s));
	static const char *skip = getenv("ACM_SIEVE_SKIP");   // experiment (results are void): "c" no check kernel, "e" no emit kernel
	const bool skip_check = skip && strchr(skip, 'c'), skip_emit = skip && strchr(skip, 'e');
#define ACM_CHECK(W)                                                                                                                        \
	if (helpers) hipLaunchKernelGGL((k_sieve_check<W, true, kTilesPerChecker>), dim3(cblocks * count), dim3(kCheckBlock), 0, s, grp, helpers); \
	else if (tpc == kTilesPerCheckerWide)                                                                                                       \
		hipLaunchKernelGGL((k_sieve_check<W, false, kTilesPerCheckerWide>), dim3(cblocks * count), dim3(kCheckBlock), 0, s, grp, helpers);     \
	else hipLaunchKernelGGL((k_sieve_check<W, false, kTilesPerChecker>), dim3(cblocks * count), dim3(kCheckBlock), 0, s, grp, helpers)
	if (!skip_check)
	switch (d->sv_stride) {
	case 8: ACM_CHECK(8); break;
	case 4: ACM_CHECK(4); break;
	case 2: ACM_CHECK(2); break;
	default: ACM_CHECK(1); break;
	}
#undef ACM_CHECK
	if (!skip_emit) {
		if (helpers)
			hipLaunchKernelGGL(k_sieve_emit<false>, dim3(eblocks * count), dim3(kEmitBlock), 0, s, grp);
		else
			hipLaunchKernelGGL(k_sieve_emit<true>, dim3(eblocks * count), dim3(kEmitBlock), 0, s, grp);
	}
	if (after_emit)
		ACM_HIP_TRY(hipEventRecord(after_emit, s));
	if (want_stamps) {   // debugging aid: where the waves spend their time (100 MHz clock)
		std::vector<unsigned long long> h(stamp_words);
		ACM_HIP_TRY(hipStreamSynchronize(s));
		ACM_HIP_TRY(hipMemcpy(h.data(), d_stamps, stamp_words * 8, hipMemcpyDeviceToHost));
		unsigned long long t0 = ~0ull;
		for (size_t i = 0; i < stamp_words / 8; i++)
			if (h[i * 8])
				t0 = std::min(t0, h[i * 8]);
		auto report = [&](const char *what, size_t first, size_t count, const char *const *names) {
			std::vector<double> col[5];
			unsigned long long rounds = 0, cands = 0;
			for (size_t i = first; i < first + count; i++) {
				if (!h[i * 8])
					continue;
				for (int k = 0; k < 5; k++)
					if (h[i * 8 + k])
						col[k].push_back((double)(h[i * 8 + k] - t0) / 100.0);
				rounds += h[i * 8 + 5] >> 32;
				cands += h[i * 8 + 5] & 0xFFFFFFFFull;
			}
			for (int k = 0; k < 5; k++) {
				std::sort(col[k].begin(), col[k].end());
				if (col[k].empty())
					continue;
				fprintf(stderr, "[%s] %-10s min %7.2f  p50 %7.2f  p99 %7.2f  max %7.2f us (%zu waves)\n", what, names[k],
				    col[k].front(), col[k][col[k].size() / 2], col[k][col[k].size() * 99 / 100], col[k].back(),
				    col[k].size());
			}
			if (rounds)
				fprintf(stderr, "[%s] stage-1 rounds %llu, samples %llu\n", what, rounds, cands);
		};
		const char *n1[5] = { "start", "filled", "probed", "", "end" };
		const char *n2[5] = { "start", "", "", "", "end" };
		report("sieve", 0, (size_t)blocks * kWaves, n1);
		report("check", 8192, cwaves, n2);
		const char *n3[5] = { "start", "loaded", "scanned", "copied", "end" };
		report("emit", 9000, eblocks, n3);
		for (uint32_t bl = 0; bl < eblocks; bl++)
			(void)bl;
		{
			std::vector<double> by[40];
			for (size_t i = 8192; i < 8192 + cwaves; i++) {
				if (!h[i * 8] || !h[i * 8 + 4])
					continue;
				const unsigned lv = (unsigned)(h[i * 8 + 6] >> 32);
				by[lv < 39 ? lv : 39].push_back((double)(h[i * 8 + 4] - h[i * 8]) / 100.0);
			}
			for (int k = 0; k < 40; k++) {
				if (by[k].empty())
					continue;
				std::sort(by[k].begin(), by[k].end());
				fprintf(stderr, "[check] follow levels %2d: %5zu waves, start->end p50 %6.2f max %6.2f us\n", k, by[k].size(),
				    by[k][by[k].size() / 2], by[k].back());
			}
		}
		fprintf(stderr, "[sieve] tiles %u, tile bytes %u, bloom words %u\n", g.ntiles, g.tile_bytes, a.bloom_words);
	}
	ACM_HIP_TRY(hipGetLastError());
	return ACM_OK;
}

int sparse_scan_enqueue(const acm_dfa *d, const acm_scan_batch *b, uint32_t init_dev, const uint32_t *init_ptr, void *sparse_ws,
    uint32_t *path_marker, hipStream_t s, hipEvent_t after_sieve, hipEvent_t after_emit)
{
	SieveJob job;
	job.batch = b;
	job.init_dev = init_dev;
	job.init_ptr = init_ptr;
	job.sparse_ws = sparse_ws;
	job.path_marker = path_marker;
	return sparse_group_enqueue(d, &job, 1, s, after_sieve, after_emit);
}

uint32_t sparse_max_group() { return kMaxGroup; }

}  // namespace acm
