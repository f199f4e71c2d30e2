// The LDS-resident form of a small automaton (host side: construction + a reference step).
//
// A set of words over a small alphabet (the sentiment set: 15 704 states, 27 byte classes) has a
// full DFA of 1 MB -- six times the LDS -- yet its rows are nearly all copies: the row of a trie
// node is the row of its fail state with the node's own children written over it.  So only the
// shallow states (and the few that branch) keep a full row; every other state is a 4-byte record
// "one override, else look in THAT row", and the whole automaton -- every state the walk can be
// in, final ones included -- fits the 160 KiB of a CU.  The walk kernel (lds_walk.hip) then never
// leaves the LDS: no cold plane, no wave-uniform gather branch.
//
//   state code   e = final << 15 | cid << 1     (16 bits; cid = compact state id < 16384; "final" is
//                                                 the sign of the code read as an int16)
//   cls    u8 [256]      byte -> class, at LDS address 0
//   rows   u16 [R][NC]   full rows of R chosen states, cells are state codes; a row is named by its
//                        first cell, counted in cells from LDS address 0
//   rec    u64 [n]       per state: up to two explicit transitions and where to look otherwise:
//                          tgt1 | cls1 << 16 | cls2 << 24,  tgt2 | next16 << 16     (class 0xFF: none)
//                          next16 <  0xC000: next = cell [next16 + class]   (a row)
//                          next16 >= 0xC000: ask the record of state next16 - 0xC000 (the fail state's)
// Which states keep a row: the root, every state with more than two transitions that neither its
// nearest row nor its fail state's record explains, and -- shallow first, while the LDS has room --
// states that would otherwise defer to their fail state's record (a second hop for the walk).
// Compact ids: every state in trie preorder (root = 0), so that a state's first child is cid + 1 and
// its code one small delta away.  A row is found through its state's record, not through the id.
#pragma once

#include <cstdint>
#include <vector>

struct acm_automaton;

namespace acm {

constexpr uint32_t kCompactMaxStates = 16384;
constexpr uint32_t kCompactSideBase = 0xC000;
constexpr uint32_t kCompactNoClass = 0xFF;
constexpr uint32_t kCompactLdsBytes = 160 * 1024 - 2048;   // what the image may take of a CU's LDS (the rest: the few words of a scatter workgroup running next to the walk)

struct CompactTables {
	bool ok = false;
	uint32_t nc = 0;           // byte classes
	uint32_t n = 0;            // states
	uint32_t rows = 0;         // R
	uint32_t nside = 0;
	uint32_t off_rows = 0, off_rec = 0, off_side = 0, off_cls = 0, image_bytes = 0;
	std::vector<uint8_t> image;               // what the kernel copies to LDS, 16-byte granules
	std::vector<uint32_t> ref2cid, cid2ref;
	std::vector<uint8_t> is_final;            // [cid]
	// statistics of the construction
	uint32_t promoted = 0, simple = 0, side_row = 0, side_link = 0;

	uint32_t code_of_ref(uint32_t ref) const { return (ref2cid[ref] << 1) | ((uint32_t)is_final[ref2cid[ref]] << 15); }
	static uint32_t cid_of_code(uint32_t e) { return (e & 0x7FFEu) >> 1; }
};

// Builds the tables; t.ok stays false when the set does not qualify (too many states or classes,
// or the rows that must be full do not fit lds_bytes).
void build_compact(const acm_automaton &a, CompactTables &t, uint32_t lds_bytes = kCompactLdsBytes);

// One step of the walk exactly as the kernel takes it, on the host image.
uint32_t compact_step(const CompactTables &t, uint32_t e, uint8_t byte, uint32_t *hops = nullptr);   // hops: records it deferred through

}  // namespace acm
