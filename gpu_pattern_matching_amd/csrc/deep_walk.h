// Device helpers shared by the chain pipeline (scan.hip) and the sparse
// pipeline (sparse.hip): exact DFA walking over the cold/meta planes with
// fast-forward along unary trie paths.  'A' is any kernel-argument struct
// with members cold, meta, in_byte, text, text16, n_pad.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace acm_dev {

// byte m-1 (m = 1-based step) of the text starting at 16-byte aligned 'base';
// a 16-byte group is loaded when the walk enters it, off the dependent chain
struct ChainText {
	const uint4 *p;
	uint4 w;
	uint32_t group;
	template <class A>
	__device__ __forceinline__ ChainText(const A &a, uint32_t base)
	    : p(a.text16 + (base >> 4)), group(0xFFFFFFFFu)
	{
		w = make_uint4(0, 0, 0, 0);
	}
	__device__ __forceinline__ uint32_t at(uint32_t m)
	{
		const uint32_t g = (m - 1) >> 4, k = (m - 1) & 15;
		if (g != group) {
			w = p[g];
			group = g;
		}
		const uint32_t d = k < 8 ? (k < 4 ? w.x : w.y) : (k < 12 ? w.z : w.w);
		return (d >> (8 * (k & 3))) & 0xFFu;
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
	const size_t idx = ((size_t)state << 8) | byte;
	const uint32_t next = a.cold[idx], m = a.meta[idx];   // two independent loads, one level
	Deep d;
	d.s = next;
	d.depth = m & 0xFFFFu;
	d.run = m >> 16;
	return d;
}

// Fast-forward along the unary path ahead of d.  Text byte 'pos' is the next
// one to consume, at most 'limit' bytes may be consumed.  While the text
// agrees with the single outgoing edge of each state, the walk goes
// s -> s+1 -> ...; none of the states entered is final and depth grows in
// step with the bytes consumed (an unmerged walk stays unmerged).  One load
// level moves the walk up to 16 bytes.  Returns the bytes consumed.
template <class A>
__device__ __forceinline__ uint32_t fast_forward(const A &a, Deep &d, uint32_t pos, uint32_t limit)
{
	uint32_t total = 0;
	while (d.run != 0 && total < limit && pos + total + 16 <= a.n_pad) {
		const uint32_t want = min(min(d.run, limit - total), 16u);
		uint64_t e0, e1, t0, t1;
		__builtin_memcpy(&e0, a.in_byte + d.s + 1, 8);
		__builtin_memcpy(&e1, a.in_byte + d.s + 9, 8);
		__builtin_memcpy(&t0, a.text + pos + total, 8);
		__builtin_memcpy(&t1, a.text + pos + total + 8, 8);
		const uint64_t x0 = e0 ^ t0, x1 = e1 ^ t1;
		uint32_t same = x0 ? (uint32_t)(__ffsll((long long)x0) - 1) >> 3
				   : 8u + (x1 ? (uint32_t)(__ffsll((long long)x1) - 1) >> 3 : 8u);
		same = min(same, want);
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
