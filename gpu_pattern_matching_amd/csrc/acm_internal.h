// Internal declarations shared by the translation units of libacmatch.so.
// Nothing here is part of the ABI (see include/acmatch.h for that).
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "acmatch.h"

namespace acm {

// ---- error plumbing -------------------------------------------------------
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

#define ACM_HIP_TRY(expr)                                                      \
	do {                                                                   \
		hipError_t e_ = (expr);                                        \
		if (e_ != hipSuccess)                                          \
			return acm::fail(ACM_ERR_HIP, "%s: %s (%s:%d)", #expr, \
			    hipGetErrorString(e_), __FILE__, __LINE__);        \
	} while (0)

// ---- design limits --------------------------------------------------------
constexpr uint32_t kMaxStates = 1u << 24;     // state id packed in 24 bits of a staged record
constexpr int kMaxPatternLine = 4096;         // utils.h:14 MAX_PAT_SIZE
constexpr uint32_t kHotRowsMax = 256;         // rows of 256 cells staged in LDS (256 * 512 B = 128 KiB); more with byte classes
constexpr uint32_t kHotBytes = kHotRowsMax * 512;
constexpr uint32_t kHotSentinel = 0xFFFFu;    // hot cell value meaning "look in the cold plane"
constexpr uint32_t kNoPattern = 0xFFFFFFFFu;

}  // namespace acm

// Host automaton.  States carry two numberings:
//   ref id : the reference's creation order (acsmx.c:339-344), used at the
//            API boundary (last_state, acsm_get_states, table export)
//   dev id : [0, hot_count)        the first non-final states in BFS order
//                                  (root, depth 1, ...): the rows the walk
//                                  kernel keeps in LDS;
//            [hot_count, first_final)  every other non-final state, in ref
//                                  order -- states created by one pattern
//                                  insertion are consecutive, so a unary trie
//                                  path is a run of consecutive dev ids;
//            [first_final, n)      final states, in ref order.
//            "final" == id >= first_final.  Root is dev id 0.
struct acm_automaton {
	struct Pattern {
		std::vector<unsigned char> bytes;
		int iid;
	};
	std::vector<Pattern> patterns;
	int max_pattern_len = 0;
	bool compiled = false;

	uint32_t num_states = 0;               // highest ref id + 1
	std::vector<uint32_t> parent;          // [ref]
	std::vector<uint8_t> in_byte;          // [ref] byte on the edge from parent
	std::vector<uint16_t> depth;           // [ref]
	std::vector<uint32_t> fail;            // [ref] -> ref
	std::vector<uint32_t> child_begin;     // [ref+1] CSR into child_list
	struct Edge {
		uint8_t byte;
		uint32_t to;
	};
	std::vector<Edge> child_list;          // children sorted by byte
	std::vector<uint32_t> bfs_order;       // ref ids in BFS order (byte order per node)

	// match lists (pattern indices, head first) for states that have one
	std::vector<int32_t> list_begin;       // [ref] offset into list_pool or -1
	std::vector<int32_t> list_len;         // [ref]
	std::vector<int32_t> list_pool;

	// Byte classes: bytes that occur in no pattern all lead to the root from every state, so
	// they share one column of the DFA (class 0); every other byte has a class of its own.  A
	// set over a small alphabet (26 letters -> 27 classes) has rows of 32 cells instead of 256:
	// eight times as many states fit the LDS rows, and the planes of the chain pipeline shrink
	// to L2 size.  log_stride = 8: no compression (every byte its own class, identity map).
	uint8_t byte_class[256];
	uint8_t class_byte[256];               // a byte of each class
	uint32_t num_classes = 256, log_stride = 8;

	std::vector<uint32_t> ref2dev, dev2ref;
	uint32_t hot_count = 0;                // dev ids below this are the LDS rows
	uint32_t hot_depth1 = 1;               // hot dev ids below this have depth <= 1
	uint32_t first_final = 0;              // dev ids >= this are final
	std::vector<int32_t> next_chained;     // [pattern] acsm_get_patterns_table chain
	// dev_run[d]: how many steps d -> d+1 -> ... follow the trie: d+1 is the
	// ONLY child of d and is not final.  Along such a run the walk only has
	// to compare text with in_byte of the states ahead -- 16 bytes per load.
	std::vector<uint16_t> dev_run;         // [dev]

	// dense DFA, dev numbering, [num_states][256] cells; built on first use.
	// cell = target dev id | depth(target) << 32 | dev_run[target] << 48
	mutable std::vector<uint64_t> dense;
	const std::vector<uint64_t> &dense_rows() const;
	void drop_dense() const { std::vector<uint64_t>().swap(dense); }

	int head_of(uint32_t ref) const
	{
		return list_begin[ref] < 0 ? -1 : list_pool[list_begin[ref]];
	}
	bool is_final_ref(uint32_t ref) const { return ref != 0 && list_begin[ref] >= 0; }
};
