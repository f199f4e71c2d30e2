// acm_dfa: the device-resident automaton (layout described in device_dfa.hip).
#pragma once

#include <atomic>
#include <cstdint>
#include <mutex>
#include <vector>

#include "acmatch.h"

struct acm_dfa {
	int device = 0;
	int num_cus = 256;
	uint32_t num_states = 0;
	uint32_t first_final = 0;
	uint32_t hot_rows = 0;
	uint32_t hot_depth1 = 1;             // hot ids below this have depth <= 1
	uint32_t max_pattern_len = 0;

	uint32_t log_stride = 8;             // cells per row of the planes below = 1 << log_stride (byte classes; 8: bytes)
	uint8_t *d_class = nullptr;          // [256] byte -> class (the identity when log_stride == 8)
	uint32_t *d_cold = nullptr;          // [states][stride] target id
	uint64_t *d_deep = nullptr;          // [states][stride] target id | depth(target) << 32 | run(target) << 48
	uint16_t *d_hot = nullptr;           // [hot_rows][stride]
	int32_t *d_out = nullptr;            // [states] reported pattern index
	uint32_t *d_dev2ref = nullptr;       // [states]
	uint32_t *d_ref2dev = nullptr;       // [states] the other way: a final state handed over on the device (acm_scan_batch.d_init_plane)
	uint8_t *d_in_byte = nullptr;        // [states + 224] byte on the edge into dev state
	uint16_t *d_depth = nullptr;         // [states] trie depth of each state (carried-state walker of the sparse pipeline)

	// sparse ("sieve") pipeline tables, sieve_tables.h
	uint32_t sv_stride = 0;              // W: one text position in W is sampled
	uint32_t sv_gram_len = 3;            // bytes of a sample the filter looks at: 3, or 6 where every pattern has W + 5 bytes
	uint32_t sv_run_ok[8] = { 0 };       // bit b: D copies of byte b are a trie path
	uint32_t sv_prefix_len = 0;          // D: bytes of a pattern checked exactly before a follower starts
	uint32_t *d_sv_bloom = nullptr;      // [1 << sv_bloom_log_words] Bloom filter over the 3-grams at offsets < W, copied to LDS
	uint32_t sv_bloom_log_words = 0;
	uint32_t *d_sv_gram = nullptr;       // [4 << sv_gram_log_buckets] gram | offset mask << 24, four per bucket
	uint32_t sv_gram_log_buckets = 0, sv_gram_probes = 1;
	uint32_t *d_sv_prefix = nullptr;     // [4 << sv_prefix_log_slots] 16-byte slots: key, run, node
	uint32_t sv_prefix_log_slots = 0, sv_prefix_probes = 1;
	void *d_sv_rec = nullptr;            // [states] node records (acm::SieveRec, 16 bytes)
	void *d_sv_edges = nullptr;          // edges of the nodes with two or more children
	uint32_t *d_list_begin = nullptr;    // [states, reference numbering] offset of the state's match list in d_list_pool
	uint32_t *d_list_len = nullptr;      // [states] its length (0: not final)
	int32_t *d_list_pool = nullptr;      // pattern indices, list order
	size_t device_bytes = 0;
	void *arena = nullptr;               // one allocation for the small tables (device_dfa.hip, upload_small)
	size_t arena_bytes = 0, arena_used = 0;

	// adaptive AUTO mode (scan.hip, pick_sparse): sparse batches that turned out dense in matches,
	// counted by the device in pinned host memory
	uint32_t *h_giveups = nullptr, *d_giveups = nullptr;
	mutable std::atomic<uint32_t> sparse_batches{0}, giveups_seen{0}, chain_hold{0};
	mutable std::atomic<uint32_t> auto_window{16}, auto_next_hold{64};   // batches per look at the counter; chain batches after a bad look

	// the automaton whole in LDS (lds_walk.hip, compact_tables.h): small alphabets, <= 16384 states
	bool lds_ok = false;
	uint8_t *d_lds_image = nullptr;      // what the walk kernel copies to LDS
	int32_t *d_lds_out = nullptr;        // [compact id] reported pattern index
	uint32_t *d_lds_cid2ref = nullptr;   // [compact id] reference id
	uint32_t *d_lds_ref2code = nullptr;  // [reference id] state code (a state handed over on the device)
	uint32_t lds_image_bytes = 0, lds_off_rec = 0, lds_halo = 0, lds_rows = 0, lds_root_code = 0, lds_final_code = 0;
	std::vector<uint16_t> lds_ref2code;  // host: state code of a reference id (init_state)

	bool sparse_ok = false;              // every pattern has >= 3 bytes: the sparse pipeline applies
	int scan_mode = ACM_SCAN_MODE_AUTO;

	std::vector<uint32_t> ref2dev;       // host copy for init_state

	int chain_bytes = 0;                 // 0 = pick automatically
	int chains_per_lane = 4;             // 2 or 4 independent chains per lane in the walk

	// HIP graphs of batches that repeat (scan.hip, acm_scan_batch_async)
	struct GraphKey {
		const void *text;
		size_t n, halo;
		long offset_shift, init_state;
		void *workspace;
		size_t workspace_bytes;
		void *pat_plane, *off_plane;
		size_t plane_capacity;
		int report, mode, chain_bytes, chains_per_lane;
		const void *init_plane;
		size_t init_plane_capacity;
	};
	struct GraphEntry {
		GraphKey key;
		void *exec;          // hipGraphExec_t, null until the key has been seen twice
		uint64_t last_use;
	};
	static constexpr size_t kMaxGraphs = 32;
	bool use_preload = true;             // halo mode: k_halo_walk (text loaded up front) where the chains are short enough
	bool use_halo = true;                // chain pipeline: halo mode where the longest pattern fits a chain (scan.hip)
	int max_group = 16;                   // batches acm_scan_batches_async puts into one set of sparse launches
	mutable bool use_graphs = false;     // opt-in: measured neutral on this stack
	mutable std::vector<GraphEntry> graphs;
	mutable std::vector<void *> parked_graphs;   // evicted execs, destroyed by acm_dfa_release
	mutable std::mutex graph_mutex;
	mutable uint64_t graph_tick = 0;

	// optional in-line timing (acm_scan_profile_*): event triples
	// {before walk, after walk, after last kernel} per recorded launch
	mutable bool profile = false;
	mutable std::vector<void *> profile_events;
	mutable std::vector<void *> profile_pool;   // idle events, reused
	mutable std::mutex profile_mutex;
};
