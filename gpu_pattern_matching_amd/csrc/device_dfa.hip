// Device-resident DFA: HBM/LDS layout + upload.
//
// Replaces the d_trans half of acsm_gen_state_table (acsmx.c:618-666).  The
// reference ships one int32 [state][2][256] table (2 KiB per state, final
// transitions stored negated, pattern index in a second plane).  Here, in the
// device numbering of acm_internal.h (hot rows, other non-finals in reference
// order, finals last):
//
//   cold   u32 [states][256]    1 KiB rows, next state; "the transition is
//                               final" is  next >= first_final  -- no flag
//                               bits.  The walk kernel reads only this plane.
//   deep   u64 [states][256]    next | depth(next) << 32 | run(next) << 48: what
//                               the exact walks (boundary resolve, sparse
//                               walkers) need about the state they enter
//                               (merge test, fast-forward) comes back in the
//                               same load as the state -- one load, one TLB
//                               entry per step on a latency-bound path.  Costs
//                               2 KiB per state next to the 1 KiB cold row;
//                               HBM is not what this path is short of.
//   hot    u16 [H][256]         rows of the first H (<= 256) non-final states
//                               in BFS order (root, depth 1, ...): copied to
//                               LDS by the walk kernel.  A cell holds the next
//                               state, or 0xFFFF when it does not fit or is
//                               final (the lane then reads the cold plane).
//   out    i32 [states]         pattern index a final state reports (the head
//                               of its match list, acsmx.c:650), -1 otherwise
//   dev2ref u32 [states]        back to the reference's numbering (last_state)
//   in_byte u8 [states + 96]    byte on the trie edge into each state; along
//                               a unary path the states ahead are d+1, d+2, ...
//                               so 16 expected bytes are one contiguous load
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>

#include "acm_internal.h"
#include "device_dfa.h"
#include "sparse.h"

extern "C" int acm_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

namespace {

template <typename T>
int upload(T **dptr, const T *src, size_t count, size_t *total)
{
	size_t bytes = (count ? count : 1) * sizeof(T);
	ACM_HIP_TRY(hipMalloc((void **)dptr, bytes));
	if (count)
		ACM_HIP_TRY(hipMemcpy(*dptr, src, count * sizeof(T), hipMemcpyHostToDevice));
	*total += bytes;
	return ACM_OK;
}

}  // namespace

extern "C" int acm_dfa_upload(const acm_automaton *a, int device, acm_dfa **out)
{
	if (!a || !a->compiled || !out)
		return acm::fail(ACM_ERR_ARG, "acm_dfa_upload: automaton not compiled");
	int ndev = acm_device_count();
	if (ndev <= 0)
		return acm::fail(ACM_ERR_NODEV, "acm_dfa_upload: no HIP device visible");
	if (device < 0 || device >= ndev)
		return acm::fail(ACM_ERR_NODEV, "acm_dfa_upload: device %d not in [0,%d)", device, ndev);
	ACM_HIP_TRY(hipSetDevice(device));

	acm_dfa *d = new (std::nothrow) acm_dfa();
	if (!d)
		return acm::fail(ACM_ERR_NOMEM, "acm_dfa_upload: out of host memory");
	d->device = device;
	d->num_states = a->num_states;
	d->first_final = a->first_final;
	d->hot_rows = a->hot_count;
	d->hot_depth1 = a->hot_depth1;
	d->max_pattern_len = (uint32_t)a->max_pattern_len;
	d->ref2dev = a->ref2dev;

	int rc = ACM_OK;
	try {
		const std::vector<uint64_t> &rows = a->dense_rows();
		const uint32_t n = a->num_states, H = a->hot_count, F = a->first_final;

		std::vector<uint16_t> hot((size_t)H * 256);
		for (size_t i = 0; i < hot.size(); i++) {
			const uint32_t t = (uint32_t)rows[i];
			hot[i] = (uint16_t)((t < acm::kHotSentinel && t < F) ? t : acm::kHotSentinel);
		}
		std::vector<int32_t> outp(n);
		std::vector<uint8_t> inb((size_t)n + 96, 0);
		for (uint32_t s = 0; s < n; s++) {
			const uint32_t r = a->dev2ref[s];
			outp[s] = a->is_final_ref(r) ? a->head_of(r) : -1;
			inb[s] = a->in_byte[r];
		}
		{
			std::vector<uint32_t> plane(rows.size());
			for (size_t i = 0; i < rows.size(); i++)
				plane[i] = (uint32_t)rows[i];
			rc = upload(&d->d_cold, plane.data(), plane.size(), &d->device_bytes);
			if (rc == ACM_OK)
				rc = upload(&d->d_deep, rows.data(), rows.size(), &d->device_bytes);
		}
		if (rc == ACM_OK) rc = upload(&d->d_hot, hot.data(), hot.size(), &d->device_bytes);
		if (rc == ACM_OK) rc = upload(&d->d_out, outp.data(), outp.size(), &d->device_bytes);
		if (rc == ACM_OK)
			rc = upload(&d->d_dev2ref, a->dev2ref.data(), a->dev2ref.size(), &d->device_bytes);
		if (rc == ACM_OK) rc = upload(&d->d_in_byte, inb.data(), inb.size(), &d->device_bytes);

		// sparse pipeline: Bloom filter over the byte triples that lead to a
		// depth-3 state, and the depth <= 2 part of the DFA as a flat table
		std::vector<uint32_t> tris, t2g(65536);
		for (uint32_t s = 0; s < n; s++) {
			const uint32_t r2 = a->dev2ref[s];
			if (a->depth[r2] != 3)
				continue;
			const uint32_t r1 = a->parent[r2], r0 = a->parent[r1];
			tris.push_back((uint32_t)a->in_byte[r0] | ((uint32_t)a->in_byte[r1] << 8) |
			    ((uint32_t)a->in_byte[r2] << 16));
		}
		d->bloom_log_words = acm::kBloomMinLogWords;
		while (d->bloom_log_words < acm::kBloomMaxLogWords && ((size_t)1 << d->bloom_log_words) < 4 * tris.size())
			d->bloom_log_words++;
		if (const char *e = getenv("ACM_BLOOM_LOG_WORDS")) {   // debugging aid
			const int v = atoi(e);
			if (v >= (int)acm::kBloomMinLogWords && v <= (int)acm::kBloomMaxLogWords)
				d->bloom_log_words = (uint32_t)v;
		}
		std::vector<uint32_t> bloom((size_t)1 << d->bloom_log_words, 0);
		for (uint32_t tri : tris)
			bloom[acm::bloom_word(tri, d->bloom_log_words)] |= acm::bloom_bits(tri);
		for (uint32_t p = 0; p < 256; p++) {
			const uint32_t s1 = (uint32_t)rows[p];
			for (uint32_t c = 0; c < 256; c++)
				t2g[p | (c << 8)] = (uint32_t)rows[(size_t)s1 * 256 + c];
		}
		// match lists, for all-patterns reporting (post.hip, acm_expand_matches_async)
		std::vector<uint32_t> lbegin(n, 0), llen(n, 0);
		for (uint32_t r = 0; r < n; r++)
			if (a->is_final_ref(r)) {
				lbegin[r] = (uint32_t)a->list_begin[r];
				llen[r] = (uint32_t)a->list_len[r];
			}
		if (rc == ACM_OK) rc = upload(&d->d_list_begin, lbegin.data(), lbegin.size(), &d->device_bytes);
		if (rc == ACM_OK) rc = upload(&d->d_list_len, llen.data(), llen.size(), &d->device_bytes);
		if (rc == ACM_OK) rc = upload(&d->d_list_pool, a->list_pool.data(), a->list_pool.size(), &d->device_bytes);
		if (rc == ACM_OK) rc = upload(&d->d_bloom, bloom.data(), bloom.size(), &d->device_bytes);
		if (rc == ACM_OK) rc = upload(&d->d_t2g, t2g.data(), t2g.size(), &d->device_bytes);
		d->sparse_ok = !a->patterns.empty();
		for (const auto &p : a->patterns)
			if (p.bytes.size() < 3)
				d->sparse_ok = false;
	} catch (const std::bad_alloc &) {
		rc = acm::fail(ACM_ERR_NOMEM, "acm_dfa_upload: out of host memory");
	}
	if (rc != ACM_OK) {
		acm_dfa_release(d);
		return rc;
	}
	if (acm::scan_prepare(d) != ACM_OK || (d->sparse_ok && acm::sparse_prepare(d) != ACM_OK)) {
		acm_dfa_release(d);
		return ACM_ERR_HIP;
	}
	if (d->sparse_ok && hipHostMalloc((void **)&d->h_giveups, 64, hipHostMallocMapped) == hipSuccess) {
		*d->h_giveups = 0;
		if (hipHostGetDevicePointer((void **)&d->d_giveups, d->h_giveups, 0) != hipSuccess)
			d->d_giveups = nullptr;
	}
	if (!d->d_giveups && d->h_giveups) {   // no adaptive mode without the counter
		hipHostFree(d->h_giveups);
		d->h_giveups = nullptr;
	}
	(void)hipGetLastError();
	if (const char *m = getenv("ACM_SCAN_MODE")) {   // debugging aid: same as acm_scan_set_mode
		if (!strcmp(m, "chain")) d->scan_mode = ACM_SCAN_MODE_CHAIN;
		else if (!strcmp(m, "sparse")) d->scan_mode = ACM_SCAN_MODE_SPARSE;
	}
	if (const char *g = getenv("ACM_SCAN_GRAPHS"))
		d->use_graphs = atoi(g) != 0;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess)
		d->num_cus = prop.multiProcessorCount;
	*out = d;
	return ACM_OK;
}

extern "C" void acm_dfa_release(acm_dfa *d)
{
	if (!d)
		return;
	if (hipSetDevice(d->device) == hipSuccess) {
		hipFree(d->d_cold);
		hipFree(d->d_deep);
		hipFree(d->d_hot);
		hipFree(d->d_out);
		hipFree(d->d_dev2ref);
		hipFree(d->d_in_byte);
		hipFree(d->d_list_begin);
		hipFree(d->d_list_len);
		hipFree(d->d_list_pool);
		hipFree(d->d_bloom);
		hipFree(d->d_t2g);
		if (d->h_giveups)
			hipHostFree(d->h_giveups);
		for (auto &g : d->graphs)
			if (g.exec)
				hipGraphExecDestroy((hipGraphExec_t)g.exec);
		for (void *e : d->profile_events)
			hipEventDestroy((hipEvent_t)e);
		for (void *e : d->profile_pool)
			hipEventDestroy((hipEvent_t)e);
	}
	delete d;
}

extern "C" size_t acm_dfa_device_bytes(const acm_dfa *d) { return d ? d->device_bytes : 0; }
extern "C" int acm_dfa_hot_rows(const acm_dfa *d) { return d ? (int)d->hot_rows : 0; }
extern "C" int acm_dfa_device(const acm_dfa *d) { return d ? d->device : -1; }
