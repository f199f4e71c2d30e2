// Tables of the sparse ("sieve") scan pipeline, and the hash functions the host
// builder (device_dfa.hip) and the kernels (sparse.hip) must agree on.
//
// Idea (details and the proof sketch are in sparse.hip).  With m = length of the
// shortest pattern, every final state has trie depth >= m.  Sample the text every
// W bytes (W = 8 when m >= 10) and test the 3 bytes at the sample against the set
//   G = { P[o .. o+2] : P a pattern, 0 <= o < W }
// Any trie path of >= W + 2 bytes that starts at s has a sample p in [s, s + W)
// whose 3-gram is P[p - s ..], i.e. in G: no start is missed, and only one text
// position in W is ever hashed.  A flagged sample is then checked exactly:
//   gram table    3-gram -> bit mask of the offsets o it occurs at (bucketed hash)
//   prefix table  first D = min(m, 10) bytes of a pattern -> the depth-D trie node
// and what survives follows its own trie path (node records below): no fail
// links, no dense rows -- 17 bytes per state instead of 2 KiB.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define ACM_HD __host__ __device__ __forceinline__
#else
#define ACM_HD inline
#endif

namespace acm {

constexpr uint32_t kSieveMaxPrefix = 10;      // D <= 10: key (10 B) + run (2 B) + node (4 B) = one 16-byte slot
constexpr uint32_t kSieveMinLogWords = 8;     // Bloom filter: 2^8 .. 2^15 words of 32 bits (1 .. 128 KiB of LDS)
constexpr uint32_t kSieveMaxLogWords = 15;

// stride for a shortest pattern of m >= 3 bytes: the largest power of two W <= 8 with W + 2 <= m
ACM_HD uint32_t sieve_stride(uint32_t m) { return m >= 10 ? 8u : m >= 6 ? 4u : m >= 4 ? 2u : 1u; }

// ---- Bloom filter in LDS: the four bits of a key in one 64-bit block ---------
// (one ds_read_b64 per sample; a 32-bit block with two bits per key flags 0.8 % of random
// samples against 16 000 keys in 64 KiB and 5 % against 120 000 in 128 KiB, this one 0.02 %
// and 2 %)
constexpr uint32_t kSieveMulA = 0x9E3779u, kSieveMulB = 0x85EBCAu, kSieveMulC = 0xC2B2AFu;   // 24-bit odd
ACM_HD uint32_t mul24(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __umul24(a, b);
#else
	return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (b & 0xFFFFFFu));
#endif
}
// The key of a sample is its 3-gram, or -- where the shortest pattern has W + 5 bytes, i.e. where
// the 6 bytes at every sampled offset are pattern bytes -- the 3-gram and the 3 bytes behind it
// ('more' = those three, 0 otherwise): 48 bits instead of 24 are what keeps the common 3-grams of
// real binaries out of the check kernel.
constexpr uint32_t kSieveMulE = 0xB5297Bu, kSieveMulF = 0x68E31Du;
// block index: log_words counts 32-bit words, a block is two of them
ACM_HD uint32_t sieve_bloom_block(uint32_t gram, uint32_t more, uint32_t log_words)
{
	return (mul24(gram, kSieveMulA) + mul24(more, kSieveMulE)) >> (33 - log_words);
}
// the key's four bits: two in each 32-bit half of its block, so that the bulk kernel tests them with four 32-bit
// shifts (whose count field is five bits wide: no masks) instead of four 64-bit ones
ACM_HD uint64_t sieve_bloom_bits(uint32_t gram, uint32_t more)
{
	const uint32_t p = mul24(gram, kSieveMulB) + mul24(more, kSieveMulF);
	const uint32_t lo = (1u << (p >> 27)) | (1u << ((p >> 22) & 31));
	const uint32_t hi = (1u << ((p >> 17) & 31)) | (1u << ((p >> 12) & 31));
	return (uint64_t)lo | ((uint64_t)hi << 32);
}

// ---- gram table: buckets of four {gram | offset mask << 24}, 0 = empty -------
ACM_HD uint32_t sieve_gram_bucket(uint32_t gram, uint32_t log_buckets)
{
	return log_buckets ? mul24(gram, kSieveMulC) >> (32 - log_buckets) : 0u;
}

// ---- prefix table: open addressing over 16-byte slots ------------------------
// slot = { key bytes 0-3, key bytes 4-7, key bytes 8-9 | run << 16, node }; node 0 = empty
ACM_HD uint32_t sieve_prefix_slot(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t log_slots)
{
	uint32_t h = k0 * 0x9E3779B1u;
	h = (h ^ (h >> 15) ^ k1) * 0x85EBCA6Bu;
	h = (h ^ (h >> 13) ^ k2) * 0xC2B2AE35u;
	return log_slots ? h >> (32 - log_slots) : 0u;
}

// ---- node record (one per state, device numbering), 16 bytes ----------------
// What a path follower needs when it stands on a node whose unary run is used up:
//   w0  bits  0..23  the only child (dev id), or the index of the first of its edges
//       bits 24..31  byte on the edge to the only child
//   w1  bits  0..8   number of children (0 = leaf)
//       bit   9      the only child is a leaf
//       bits 16..31  run of the only child: how many steps child -> child+1 -> ... are
//                    unary and enter non-final states (acm_automaton::dev_run)
//   w2  pattern index the only child reports when it is final (head of its match list)
//   w3  reference id of the only child (what ACM_REPORT_STATE scans report)
// Edge (nodes with >= 2 children; sorted by byte), 16 bytes:
//   w0  bits 0..7 byte, 8..31 child;  w1 bits 0..15 run of the child, bit 16 child is a leaf;
//   w2, w3 as above
struct SieveRec {
	uint32_t w0, w1, w2, w3;
};
ACM_HD SieveRec sieve_rec(uint32_t child_or_edge, uint32_t byte, uint32_t nchild, bool leaf, uint32_t child_run,
    uint32_t out, uint32_t ref)
{
	SieveRec r;
	r.w0 = (child_or_edge & 0xFFFFFFu) | ((byte & 0xFFu) << 24);
	r.w1 = (nchild & 0x1FFu) | ((leaf ? 1u : 0u) << 9) | ((child_run & 0xFFFFu) << 16);
	r.w2 = out;
	r.w3 = ref;
	return r;
}
ACM_HD SieveRec sieve_edge(uint32_t byte, uint32_t child, bool leaf, uint32_t child_run, uint32_t out, uint32_t ref)
{
	SieveRec r;
	r.w0 = (byte & 0xFFu) | ((child & 0xFFFFFFu) << 8);
	r.w1 = (child_run & 0xFFFFu) | ((leaf ? 1u : 0u) << 16);
	r.w2 = out;
	r.w3 = ref;
	return r;
}

}  // namespace acm
