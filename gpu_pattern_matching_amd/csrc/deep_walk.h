// Device helpers shared by the chain pipeline (scan.hip) and the sparse
// pipeline (sparse.hip): exact DFA walking over the deep plane with
// fast-forward along unary trie paths.  'A' is any kernel-argument struct
// with members deep, ls, cls, in_byte, text, text16, n_pad.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace acm_dev {

// byte m-1 (m = 1-based step) of the text starting at 16-byte aligned 'base';
// a 16-byte group is loaded when the walk enters it, off the dependent chain
struct ChainText {
	const uint4 *p;
	uint64_t lo, hi;   // the 16-byte group as two words (a uint4 member ends up as a private array in LDS)
	uint32_t group;
	template <class A>
	__device__ __forceinline__ ChainText(const A &a, uint32_t base)
	    : p(a.text16 + (base >> 4)), lo(0), hi(0), group(0xFFFFFFFFu)
	{
	}
	__device__ __forceinline__ uint32_t at(uint32_t m)
	{
		const uint32_t g = (m - 1) >> 4, k = (m - 1) & 15;
		if (g != group) {
			const uint4 w = p[g];
			lo = (uint64_t)w.x | ((uint64_t)w.y << 32);
			hi = (uint64_t)w.z | ((uint64_t)w.w << 32);
			group = g;
		}
		const uint64_t half = k < 8 ? lo : hi;
		return (uint32_t)(half >> (8 * (k & 7))) & 0xFFu;
	}
};

// the state a deep walk is in, with what the cell that produced it said
struct Deep {
	uint32_t s;       // dev id
	uint32_t depth;   // trie depth of s
	uint32_t run;     // unary, non-final trie path ahead: s+1, s+2, ... s+run
};

template <class A>
__device__ __forceinline__ Deep deep_step(const A &a, uint32_t state, uint32_t byte)
{
	// (rows have one cell per byte class; the class map is 256 bytes that never leave the L1)
	const uint32_t col = a.ls == 8 ? byte : (uint32_t)a.cls[byte];
	const size_t idx = ((size_t)state << a.ls) | col;
	const uint64_t cell = a.deep[idx];   // target | depth << 32 | run << 48: one load, one TLB entry
	const uint32_t next = (uint32_t)cell, m = (uint32_t)(cell >> 32);
	Deep d;
	d.s = next;
	d.depth = m & 0xFFFFu;
	d.run = m >> 16;
	return d;
}

// XOR of the 16 bytes at two arbitrarily aligned addresses, as two 64-bit
// words.  gfx950 under HSA runs with unaligned access enabled: a packed
// 16-byte load is ONE global_load_dwordx4 whatever the address.
struct __attribute__((packed)) Unaligned16 {
	uint64_t lo, hi;
};

__device__ __forceinline__ void diff_bytes16(const uint8_t *p, const uint8_t *q, uint64_t &lo, uint64_t &hi)
{
	const Unaligned16 *x = (const Unaligned16 *)p, *y = (const Unaligned16 *)q;
	lo = x->lo ^ y->lo;
	hi = x->hi ^ y->hi;
}

// bytes two 16-byte windows agree on before the first difference (16: all)
__device__ __forceinline__ uint32_t agree16(const uint8_t *p, const uint8_t *q)
{
	uint64_t x0, x1;
	diff_bytes16(p, q, x0, x1);
	return x0 ? (uint32_t)(__ffsll((long long)x0) - 1) >> 3
		  : 8u + (x1 ? (uint32_t)(__ffsll((long long)x1) - 1) >> 3 : 8u);
}

// Fast-forward along the unary path ahead of d.  Text byte 'pos' is the next
// one to consume, at most 'limit' bytes may be consumed.  While the text
// agrees with the single outgoing edge of each state, the walk goes
// s -> s+1 -> ...; none of the states entered is final and depth grows in
// step with the bytes consumed (an unmerged walk stays unmerged).  One load
// level moves the walk up to 16 bytes.  (64 bytes per level -- four compares
// in flight, always or only after a first full 16 -- was measured slower on
// every percentile of the walk times, not just for short runs.)  Returns the
// bytes consumed.  Stops at the last full 16 bytes of the padded text: the
// caller single-steps there.
template <class A>
__device__ __forceinline__ uint32_t fast_forward(const A &a, Deep &d, uint32_t pos, uint32_t limit)
{
	uint32_t total = 0;
	while (d.run != 0 && total < limit && pos + total + 16 <= a.n_pad) {
		const uint32_t want = min(min(d.run, limit - total), 16u);
		const uint32_t same = min(agree16(a.in_byte + d.s + 1, a.text + pos + total), want);
		d.s += same;
		d.depth += same;
		d.run -= same;
		total += same;
		if (same < want || same < 16)
			break;   // mismatch, or the run / the limit ended inside this round
	}
	return total;
}

}  // namespace acm_dev
