// Interface between scan.hip (dispatch, chain pipeline) and sparse.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "acmatch.h"

struct acm_dfa;

namespace acm {

// bytes of private workspace the sparse pipeline needs for texts up to max_text
size_t sparse_workspace_bytes(const acm_dfa *d, size_t max_text);

// once per device DFA: kernel attributes (scan.hip / sparse.hip)
int scan_prepare(const acm_dfa *d);
int sparse_prepare(const acm_dfa *d);

// Enqueue the sparse pipeline for 'b' on stream s (three kernels; it always produces the planes).
// after_sieve / after_emit: events to record behind the two kernels, or null.
// path_marker: device word that receives ACM_SCAN_MODE_SPARSE.
int sparse_scan_enqueue(const acm_dfa *d, const acm_scan_batch *b, uint32_t init_dev, const uint32_t *init_ptr, void *sparse_ws,
    uint32_t *path_marker, hipStream_t s, hipEvent_t after_sieve, hipEvent_t after_emit);

// The same for up to sparse_max_group() batches of one size on one stream, with three launches
// for all of them: every batch needs its own workspace and planes (they are in flight together).
struct SieveJob {
	const acm_scan_batch *batch;
	uint32_t init_dev;       // device id of the batch's init_state
	const uint32_t *init_ptr;   // null, or where the device id of the state to start in is (handed over on the device)
	void *sparse_ws;         // the sparse part of the batch's workspace
	uint32_t *path_marker;
};
uint32_t sparse_max_group();
int sparse_group_enqueue(const acm_dfa *d, const SieveJob *jobs, uint32_t count, hipStream_t s, hipEvent_t after_sieve,
    hipEvent_t after_emit);

}  // namespace acm
