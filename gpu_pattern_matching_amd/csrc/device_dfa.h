// acm_dfa: the device-resident automaton (layout described in device_dfa.hip).
#pragma once

#include <cstdint>
#include <vector>

struct acm_dfa {
	int device = 0;
	int num_cus = 256;
	uint32_t num_states = 0;
	uint32_t first_final = 0;
	uint32_t hot_rows = 0;
	uint32_t max_pattern_len = 0;

	uint32_t *d_cold = nullptr;
	uint16_t *d_hot = nullptr;
	int32_t *d_out = nullptr;
	uint32_t *d_dev2ref = nullptr;
	uint32_t *d_depth_cum = nullptr;
	uint16_t *d_depth_final = nullptr;
	uint32_t *d_ffinfo = nullptr;        // [dev] ref id | unary run << 24
	uint32_t *d_ref2dev = nullptr;       // [ref]
	uint8_t *d_in_byte = nullptr;        // [ref + 32] byte on the edge into ref state
	uint8_t *d_ff_run = nullptr;         // [ref + 32] unary run length, by ref id
	uint16_t *d_t2 = nullptr;            // [256 + 65536] bigram table image for LDS
	uint8_t *d_bloom = nullptr;          // [16384] trigram filter image for LDS
	uint32_t d2lo = 0, d2hi = 0;         // non-final depth-2 ids
	uint32_t cum1 = 1;                   // depth_cum[1]
	bool use_bigram = false, bigram_default = false;
	size_t device_bytes = 0;

	std::vector<uint32_t> ref2dev;       // host copies for init_state / last_state
	std::vector<uint32_t> dev2ref_host;

	int chain_bytes = 0;                 // 0 = pick automatically
	int chains_per_lane = 4;             // 2 or 4 independent chains per lane in the walk

	// optional in-line timing (acm_scan_profile_*): event triples
	// {before walk, after walk, after last kernel} per recorded launch
	mutable bool profile = false;
	mutable std::vector<void *> profile_events;
	mutable std::vector<void *> profile_pool;   // idle events, reused
};
