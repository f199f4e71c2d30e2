// Interface between scan.hip (dispatch, workspace layout) and lds_walk.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "acmatch.h"

struct acm_dfa;
struct acm_automaton;

namespace acm {

// once per device DFA: builds the LDS form of the automaton (compact_tables.h) and uploads it;
// d->lds_ok says whether the set qualifies (small alphabet, <= 16384 states, patterns of <= 33 bytes)
int lds_walk_prepare(const acm_automaton *a, acm_dfa *d);
void lds_walk_release(acm_dfa *d);

// One batch of a launch group: the batch and the areas of its workspace the two kernels use.
struct LdsJob {
	const acm_scan_batch *batch;
	uint32_t *stage;        // lds_walk_needs: stage_words
	uint8_t *cnt;           // cnt_bytes
	uint32_t *tile_total;   // tile_words
	uint32_t *misc;         // [0] last state code, [2] path marker
	const uint32_t *init_ptr;   // null, or where the code of the state to start in is (handed over on the device)
};
void lds_walk_needs(const acm_dfa *d, size_t n, size_t *stage_words, size_t *cnt_bytes, size_t *tile_words);
uint32_t lds_walk_max_group();
// walk + scatter for up to lds_walk_max_group() batches of one size on one stream
// (after_walk, after_walk2: events to record between the two kernels, or null)
int lds_walk_enqueue(const acm_dfa *d, const LdsJob *jobs, uint32_t count, hipStream_t s, hipEvent_t after_walk,
    hipEvent_t after_walk2);

}  // namespace acm
