// The LDS-resident form of a small automaton (host side: construction + a reference step).
//
// A set of words over a small alphabet (the sentiment set: 15 704 states, 27 byte classes) has a
// full DFA of 1 MB -- six times the LDS -- yet its rows are nearly all copies: the row of a trie
// node is the row of its fail state with the node's own children written over it.  So only the
// shallow states (and the few that branch) keep a full row; every state has an 8-byte record
// "two overrides, else look in THAT row", and the whole automaton -- every state the walk can be
// in, final ones included -- fits the 160 KiB of a CU.  The walk kernel (lds_walk.hip) then never
// leaves the LDS: no cold plane, no wave-uniform gather branch.  (160 KiB less 2 KiB: the few words of LDS a
// scatter workgroup needs to run beside the walk.)
//
//   state code   e = (byte address of the state's record) / 8 = off_rec / 8 + cid     (15 bits; the walk shifts it
//                    left by three and has the address.  cid = compact state id < 16384, non-final states
//                    first: "final" is e >= final_code)
//   cls    u8 [256]      byte -> 2 * class, at LDS address 0
//   rows   u16 [R][NC]   full rows of R chosen states, cells are state codes; a row is named by the BYTE address
//                        of its first cell (even, below 0xC000: the rows end 48 KiB into the image at the latest)
//   rec    u64 [n]       per state: up to two explicit transitions and where to look otherwise:
//                          tgt1 | 2*cls1 << 16 | 2*cls2 << 24,  tgt2 | next16 << 16     (class byte 0xFF: none)
//                          next16 <  0xC000: next = the cell at byte address next16 + 2*class   (a row)
//                          next16 >= 0xC000: ask the record of state cid = next16 - 0xC000 (the fail state's)
// Which states keep a row: the root, every state with more than two transitions that neither its
// nearest row nor its fail state's record explains, and -- shallow first, while the LDS has room --
// states that would otherwise defer to their fail state's record (a second hop for the walk).
// Compact ids: the non-final states in breadth-first order (root = 0), then the final ones; states that other
// records defer to on even ids (compact_tables.cpp says why).  A row is found through its state's record, not
// through the id.  (Every field is sized so that a step of the walk is nine vector
// instructions: address = code << 3; cell address = next16 + 2*class in one add; targets are whole codes.)
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

	uint32_t first_final = 0;                 // cid of the first final state (n: none)

	uint32_t code_of_cid(uint32_t cid) const { return off_rec / 8 + cid; }
	uint32_t code_of_ref(uint32_t ref) const { return code_of_cid(ref2cid[ref]); }
	uint32_t cid_of_code(uint32_t e) const { return (e & 0xFFFFu) - off_rec / 8; }
	uint32_t root_code() const { return code_of_cid(0); }
	uint32_t final_code() const { return code_of_cid(first_final); }   // codes from here on are final states
};

// Builds the tables; t.ok stays false when the set does not qualify (too many states or classes,
// or the rows that must be full do not fit lds_bytes).
void build_compact(const acm_automaton &a, CompactTables &t, uint32_t lds_bytes = kCompactLdsBytes);

// One step of the walk exactly as the kernel takes it, on the host image.
uint32_t compact_step(const CompactTables &t, uint32_t e, uint8_t byte, uint32_t *hops = nullptr);   // hops: records it deferred through

}  // namespace acm
